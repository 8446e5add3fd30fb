// Training-mode pieces of the path (first slice): one convolution layer of the encoder /
// Postnet stacks as the reference runs it in .train() mode - conv1d + BatchNorm1d with BATCH statistics + activation +
// dropout (models/tts/tacotron2.py:149-199, :207-220, :234-235) - forward and backward, and the backward of the criterion
// (Tacotron2Loss, :598-615).  Weights come in the reference's own parameter layout (training updates them in place: there
// is no packed blob on this side); activations cross the C ABI in the reference's [B, C, T] layout.
//
// Every contraction runs on the exact-fp32 MFMA GEMM of the forward path (gemm_f32.hip):
//   forward  z[(b,t)][co]  = sum_{j,ci} xcl[b][t + j][ci] * Wk[co][j][ci] + bias        implicit GEMM on the halo-padded input
//   dgrad    dx[(b,t)][ci] = sum_{j,co} dzh[b][t + j][co] * W2[ci][j][co],  W2[ci][j][co] = W[co][ci][k-1-j]   the same, flipped taps
//   wgrad    dW[co][(j,ci)] = sum_r dz^T[co][r] * X^T[(j,ci)][r]                        both operands transposed to row-contiguous
// BatchNorm statistics and the reductions of its backward are column sums in double precision; everything else is
// elementwise.  Correctness first: these kernels are not tuned (the slice exists to pin the training semantics).
#include "../../include/genvox_amd.h"
#include "gvx_kernels.h"

#include <cstdio>

namespace gvx {
namespace {

constexpr float BN_EPS_F = 1e-5f;
inline size_t up256(size_t x) { return (x + 255) / 256 * 256; }

// [Cout][Cin][k] -> Wk[Cout][k][Cin] (forward)  and  W2[Cin][k][Cout] with flipped taps (dgrad)
__global__ void repack_conv_kernel(const float* w, float* wk, float* w2, int Cout, int Cin, int k) {
    const long n = (long)Cout * Cin * k;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i % k), ci = (int)((i / k) % Cin), co = (int)(i / ((long)k * Cin));
        const float v = w[i];
        if (wk) wk[((long)co * k + j) * Cin + ci] = v;
        if (w2) w2[((long)ci * k + (k - 1 - j)) * Cout + co] = v;
    }
}

// column sums over the rows of X [rows][C] (and of X * Y when Y != nullptr), double accumulation, fixed order
__global__ __launch_bounds__(256) void col_reduce_kernel(const float* X, const float* Y, long rows, int C, float* sum_x, float* sum_xy) {
    __shared__ double sx[8][32], sxy[8][32];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    double a = 0.0, b = 0.0;
    if (c < C)
        for (long r = rl; r < rows; r += 8) {
            const double x = X[r * C + c];
            a += x;
            if (Y) b += x * (double)Y[r * C + c];
        }
    sx[rl][cl] = a; sxy[rl][cl] = b;
    __syncthreads();
    if (rl == 0 && c < C) {
        double ta = 0.0, tb = 0.0;
        for (int i = 0; i < 8; ++i) { ta += sx[i][cl]; tb += sxy[i][cl]; }
        sum_x[c] = (float)ta;
        if (Y && sum_xy) sum_xy[c] = (float)tb;
    }
}

// biased batch variance in double from the centred values (two passes keep it exact enough for invstd); also the running
// statistics update of torch.nn.BatchNorm1d (momentum 0.1, unbiased variance)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* Z, long rows, int C, float* mean, float* invstd, float* running_mean,
                                                       float* running_var, float momentum) {
    __shared__ double s1[8][32], s2[8][32];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    double a = 0.0;
    if (c < C)
        for (long r = rl; r < rows; r += 8) a += (double)Z[r * C + c];
    s1[rl][cl] = a;
    __syncthreads();
    double m = 0.0;
    for (int i = 0; i < 8; ++i) m += s1[i][cl];
    m /= (double)rows;
    double v = 0.0;
    if (c < C)
        for (long r = rl; r < rows; r += 8) { const double d = (double)Z[r * C + c] - m; v += d * d; }
    s2[rl][cl] = v;
    __syncthreads();
    if (rl == 0 && c < C) {
        double var = 0.0;
        for (int i = 0; i < 8; ++i) var += s2[i][cl];
        var /= (double)rows;
        mean[c] = (float)m;
        invstd[c] = (float)(1.0 / sqrt(var + (double)BN_EPS_F));
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * (double)rows / (double)(rows > 1 ? rows - 1 : 1));
    }
}

__device__ __forceinline__ float act_fwd(float u, int act) { return act == ACT_TANH ? tanhf(u) : (act == ACT_RELU ? fmaxf(u, 0.f) : u); }

// z [(b,t)][c] -> xhat, a (channels-last, saved) and y[b][c][t] = a * keep / (1 - p)
__global__ void bn_act_drop_fwd_kernel(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                       const uint8_t* keep, float scale, int act, int B, int C, int T, float* xhat, float* a, float* y) {
    const long n = (long)B * T * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long bt = i / C;
        const int t = (int)(bt % T), b = (int)(bt / T);
        const float xh = (z[i] - mean[c]) * invstd[c];
        const float av = act_fwd(xh * gamma[c] + beta[c], act);
        xhat[i] = xh; a[i] = av;
        const long o = ((long)b * C + c) * T + t;
        y[o] = keep ? (keep[o] ? av * scale : 0.f) : av;
    }
}

// du[(b,t)][c] = dy[b][c][t] * keep / (1 - p) * act'(a)
__global__ void act_drop_bwd_kernel(const float* dy, const uint8_t* keep, float scale, int act, const float* a, int B, int C, int T, float* du) {
    const long n = (long)B * T * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long bt = i / C;
        const int t = (int)(bt % T), b = (int)(bt / T);
        const long o = ((long)b * C + c) * T + t;
        float g = dy[o];
        if (keep) g = keep[o] ? g * scale : 0.f;
        const float av = a[i];
        if (act == ACT_TANH) g *= 1.f - av * av;
        else if (act == ACT_RELU) g = av > 0.f ? g : 0.f;
        du[i] = g;
    }
}

// dz = gamma * invstd * (du - dbeta / n - xhat * dgamma / n), written compact [(b,t)][c] and halo-padded [b][t + pad][c]
__global__ void bn_bwd_kernel(const float* du, const float* xhat, const float* gamma, const float* invstd, const float* dbeta,
                              const float* dgamma, int B, int C, int T, int pad, float* dz, float* dzh) {
    const long n = (long)B * T * C;
    const float inv_n = 1.f / (float)((long)B * T);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long bt = i / C;
        const int t = (int)(bt % T), b = (int)(bt / T);
        const float v = gamma[c] * invstd[c] * (du[i] - dbeta[c] * inv_n - xhat[i] * dgamma[c] * inv_n);
        dz[i] = v;
        dzh[((long)b * (T + 2 * pad) + pad + t) * C + c] = v;
    }
}

// dst[c][r] = src[r][c]  for r < rows; columns of dst are padded with zeros up to rows_p
__global__ void transpose_pad_kernel(const float* src, float* dst, long rows, int C, long rows_p) {
    const long n = (long)C * rows_p;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i % rows_p;
        const int c = (int)(i / rows_p);
        dst[i] = r < rows ? src[r * C + c] : 0.f;
    }
}
// XT[(j*Cin + ci)][b*T + t] = xcl[b][t + j][ci]   (xcl halo-padded channels-last), rows padded with zeros up to rows_p
__global__ void im2col_t_kernel(const float* xcl, float* xt, int B, int Cin, int T, int k, long rows_p) {
    const int pad = (k - 1) / 2;
    const long n = (long)k * Cin * rows_p;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i % rows_p;
        const long jc = i / rows_p;
        const int ci = (int)(jc % Cin), j = (int)(jc / Cin);
        float v = 0.f;
        if (r < (long)B * T) {
            const int t = (int)(r % T), b = (int)(r / T);
            v = xcl[((long)b * (T + 2 * pad) + t + j) * Cin + ci];
        }
        xt[i] = v;
    }
}
// dwk [Cout][k][Cin] -> dw [Cout][Cin][k]
__global__ void unpack_dw_kernel(const float* dwk, float* dw, int Cout, int Cin, int k) {
    const long n = (long)Cout * Cin * k;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i % k), ci = (int)((i / k) % Cin), co = (int)(i / ((long)k * Cin));
        dw[i] = dwk[((long)co * k + j) * Cin + ci];
    }
}
// x [(b,t)][c] -> y [b][c][t]
__global__ void to_channels_first_kernel(const float* x, float* y, int B, int C, int T) {
    const long n = (long)B * T * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long bt = i / C;
        const int t = (int)(bt % T), b = (int)(bt / T);
        y[((long)b * C + c) * T + t] = x[i];
    }
}

__global__ void loss_backward_kernel(const float* mel, const float* post, const float* gate, const float* mel_t, const float* gate_t,
                                     long n_mel, long n_gate, float* dmel, float* dpost, float* dgate) {
    const float cm = 2.f / (float)n_mel, cg = 1.f / (float)n_gate;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_mel; i += (long)gridDim.x * blockDim.x) {
        dmel[i] = cm * (mel[i] - mel_t[i]);
        dpost[i] = cm * (post[i] - mel_t[i]);
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_gate; i += (long)gridDim.x * blockDim.x)
        dgate[i] = cg * (1.f / (1.f + expf(-gate[i])) - gate_t[i]);
}

inline int blocks_for(long n) { long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }

// layout of the saved-for-backward buffer and of the scratch of one layer (byte offsets)
struct ConvTrainPlan {
    size_t xcl, xhat, a, mean, invstd, saved_total;                       // saved
    size_t wk, w2, z, du, dz, dzh, dzt, xt, dwk, dxcl, xcl2, ws_total;   // workspace
};
ConvTrainPlan conv_train_plan(int B, int Cin, int Cout, int T, int k) {
    const int pad = (k - 1) / 2;
    const long rows = (long)B * T, rows_p = (rows + 3) / 4 * 4;
    ConvTrainPlan p{};
    size_t o = 0;
    auto take = [&](size_t floats) { size_t r = o; o = up256(o + floats * sizeof(float)); return r; };
    p.xcl = take((size_t)B * (T + 2 * pad) * Cin);
    p.xhat = take((size_t)rows * Cout);
    p.a = take((size_t)rows * Cout);
    p.mean = take(Cout);
    p.invstd = take(Cout);
    p.saved_total = o;
    o = 0;
    p.wk = take((size_t)Cout * k * Cin);
    p.w2 = take((size_t)Cin * k * Cout);
    p.z = take((size_t)rows * Cout);
    p.du = take((size_t)rows * Cout);
    p.dz = take((size_t)rows * Cout);
    p.dzh = take((size_t)B * (T + 2 * pad) * Cout);
    p.dzt = take((size_t)Cout * rows_p);
    p.xt = take((size_t)k * Cin * rows_p);
    p.dwk = take((size_t)Cout * k * Cin);
    p.dxcl = take((size_t)rows * Cin);
    p.xcl2 = take((size_t)B * (T + 2 * pad) * Cin);
    p.ws_total = o;
    return p;
}

thread_local char g_train_err[256];
int tfail(int code, const char* msg) { return set_error(code, msg); }

#define TR_TRY(expr)                                                                       \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            snprintf(g_train_err, sizeof g_train_err, "%s failed: %s", #expr, hipGetErrorString(_e)); \
            return set_error(GVX_ERR_HIP, g_train_err);                                    \
        }                                                                                  \
    } while (0)

template <typename T>
T* at(void* base, size_t off) { return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + off); }
template <typename T>
const T* at(const void* base, size_t off) { return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + off); }

int check_conv_args(int B, int Cin, int Cout, int T, int k) {
    if (B < 1 || T < 1 || Cin < 8 || Cout < 8 || (Cin % 8) || (Cout % 8) || k < 1 || !(k & 1))
        return tfail(GVX_ERR_UNSUPPORTED, "conv training op: channels must be positive multiples of 8, kernel size odd");
    if ((long)B * T > (1L << 30)) return tfail(GVX_ERR_UNSUPPORTED, "conv training op: B * T exceeds the GEMM row index range");
    return GVX_OK;
}

}  // namespace
}  // namespace gvx

using namespace gvx;

extern "C" {

size_t gvx_conv_train_saved_bytes(int B, int Cin, int Cout, int T, int k) {
    if (check_conv_args(B, Cin, Cout, T, k) != GVX_OK) return 0;
    return conv_train_plan(B, Cin, Cout, T, k).saved_total;
}
size_t gvx_conv_train_workspace_bytes(int B, int Cin, int Cout, int T, int k) {
    if (check_conv_args(B, Cin, Cout, T, k) != GVX_OK) return 0;
    return conv_train_plan(B, Cin, Cout, T, k).ws_total;
}

int gvx_conv_bn_act_train_forward(const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, int B, int Cin, int Cout, int T, int k, int act,
                                  const uint8_t* keep, float p_drop, float* y, void* saved, size_t saved_bytes, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    int rc = check_conv_args(B, Cin, Cout, T, k);
    if (rc != GVX_OK) return rc;
    if (!x || !w || !bias || !gamma || !beta || !y || !saved || !workspace) return tfail(GVX_ERR_INVALID_ARG, "null argument");
    if (act != ACT_NONE && act != ACT_RELU && act != ACT_TANH) return tfail(GVX_ERR_INVALID_ARG, "activation must be 0 (none), 1 (relu) or 2 (tanh)");
    if (keep && !(p_drop >= 0.f && p_drop < 1.f)) return tfail(GVX_ERR_INVALID_ARG, "dropout probability must be in [0, 1)");
    const ConvTrainPlan pl = conv_train_plan(B, Cin, Cout, T, k);
    if (saved_bytes < pl.saved_total || workspace_bytes < pl.ws_total) return tfail(GVX_ERR_WORKSPACE, "saved / workspace buffer too small");
    if ((reinterpret_cast<uintptr_t>(saved) | reinterpret_cast<uintptr_t>(workspace)) & 255) return tfail(GVX_ERR_WORKSPACE, "buffers must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int pad = (k - 1) / 2;
    const long rows = (long)B * T;
    float* xcl = at<float>(saved, pl.xcl);
    TR_TRY(launch_to_channels_last(x, xcl, B, Cin, T, pad, nullptr, s));
    float* wk = at<float>(workspace, pl.wk);
    hipLaunchKernelGGL(repack_conv_kernel, dim3(blocks_for((long)Cout * Cin * k)), dim3(256), 0, s, w, wk, (float*)nullptr, Cout, Cin, k);
    float* z = at<float>(workspace, pl.z);
    GemmParams g{};
    g.A = xcl; g.amap = RowMap{T, (long)(T + 2 * pad) * Cin, (long)Cin};
    g.W = wk; g.ldw = (long)k * Cin;
    g.C = z; g.cmap = RowMap{(int)rows, 0, (long)Cout};
    g.bias = bias; g.M = (int)rows; g.N = Cout; g.K = k * Cin; g.act = ACT_NONE;
    TR_TRY(launch_gemm(g, s));
    float* mean = at<float>(saved, pl.mean);
    float* invstd = at<float>(saved, pl.invstd);
    hipLaunchKernelGGL(bn_stats_kernel, dim3((Cout + 31) / 32), dim3(256), 0, s, z, rows, Cout, mean, invstd, running_mean, running_var, 0.1f);
    hipLaunchKernelGGL(bn_act_drop_fwd_kernel, dim3(blocks_for(rows * Cout)), dim3(256), 0, s, z, mean, invstd, gamma, beta, keep,
                       keep ? 1.f / (1.f - p_drop) : 1.f, act, B, Cout, T, at<float>(saved, pl.xhat), at<float>(saved, pl.a), y);
    TR_TRY(hipGetLastError());
    return GVX_OK;
}

int gvx_conv_bn_act_train_backward(const float* dy, const void* saved, size_t saved_bytes, const float* w, const float* gamma,
                                   const float* x_wgrad, int B, int Cin, int Cout, int T, int k, int act, const uint8_t* keep,
                                   float p_drop, float* dx, float* dw, float* dbias, float* dgamma, float* dbeta, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    int rc = check_conv_args(B, Cin, Cout, T, k);
    if (rc != GVX_OK) return rc;
    if (!dy || !saved || !w || !gamma || !dw || !dbias || !dgamma || !dbeta || !workspace) return tfail(GVX_ERR_INVALID_ARG, "null argument");
    const ConvTrainPlan pl = conv_train_plan(B, Cin, Cout, T, k);
    if (saved_bytes < pl.saved_total || workspace_bytes < pl.ws_total) return tfail(GVX_ERR_WORKSPACE, "saved / workspace buffer too small");
    hipStream_t s = (hipStream_t)stream;
    const int pad = (k - 1) / 2;
    const long rows = (long)B * T, rows_p = (rows + 3) / 4 * 4;
    const float* xhat = at<float>(saved, pl.xhat);
    const float* a = at<float>(saved, pl.a);
    const float* invstd = at<float>(saved, pl.invstd);
    float* du = at<float>(workspace, pl.du);
    hipLaunchKernelGGL(act_drop_bwd_kernel, dim3(blocks_for(rows * Cout)), dim3(256), 0, s, dy, keep, keep ? 1.f / (1.f - p_drop) : 1.f, act, a,
                       B, Cout, T, du);
    // dbeta = sum du, dgamma = sum du * xhat
    hipLaunchKernelGGL(col_reduce_kernel, dim3((Cout + 31) / 32), dim3(256), 0, s, du, xhat, rows, Cout, dbeta, dgamma);
    float* dz = at<float>(workspace, pl.dz);
    float* dzh = at<float>(workspace, pl.dzh);
    TR_TRY(hipMemsetAsync(dzh, 0, (size_t)B * (T + 2 * pad) * Cout * sizeof(float), s));
    hipLaunchKernelGGL(bn_bwd_kernel, dim3(blocks_for(rows * Cout)), dim3(256), 0, s, du, xhat, gamma, invstd, dbeta, dgamma, B, Cout, T, pad, dz, dzh);
    hipLaunchKernelGGL(col_reduce_kernel, dim3((Cout + 31) / 32), dim3(256), 0, s, dz, (const float*)nullptr, rows, Cout, dbias, (float*)nullptr);
    // weight gradient: dzT [Cout][rows_p] x XT [(j, ci)][rows_p]
    const float* xcl = at<float>(saved, pl.xcl);
    const float* xcl_w = xcl;
    if (x_wgrad) {   // (the reference masks the Postnet's input in place after its forward - outside autograd, so the first
                     // layer's weight gradient sees the MASKED input: models/tts/tacotron2.py:463, :466-473)
        float* x2 = at<float>(workspace, pl.xcl2);
        TR_TRY(launch_to_channels_last(x_wgrad, x2, B, Cin, T, pad, nullptr, s));
        xcl_w = x2;
    }
    float* dzt = at<float>(workspace, pl.dzt);
    float* xt = at<float>(workspace, pl.xt);
    hipLaunchKernelGGL(transpose_pad_kernel, dim3(blocks_for((long)Cout * rows_p)), dim3(256), 0, s, dz, dzt, rows, Cout, rows_p);
    hipLaunchKernelGGL(im2col_t_kernel, dim3(blocks_for((long)k * Cin * rows_p)), dim3(256), 0, s, xcl_w, xt, B, Cin, T, k, rows_p);
    float* dwk = at<float>(workspace, pl.dwk);
    {
        GemmParams g{};
        g.A = dzt; g.amap = RowMap{Cout, 0, rows_p};
        g.W = xt; g.ldw = rows_p;
        g.C = dwk; g.cmap = RowMap{Cout, 0, (long)k * Cin};
        g.M = Cout; g.N = k * Cin; g.K = (int)rows_p; g.act = ACT_NONE;
        TR_TRY(launch_gemm(g, s));
    }
    hipLaunchKernelGGL(unpack_dw_kernel, dim3(blocks_for((long)Cout * Cin * k)), dim3(256), 0, s, dwk, dw, Cout, Cin, k);
    if (dx) {   // data gradient: flipped-tap implicit GEMM on the halo-padded dz
        float* w2 = at<float>(workspace, pl.w2);
        hipLaunchKernelGGL(repack_conv_kernel, dim3(blocks_for((long)Cout * Cin * k)), dim3(256), 0, s, w, (float*)nullptr, w2, Cout, Cin, k);
        float* dxcl = at<float>(workspace, pl.dxcl);
        GemmParams g{};
        g.A = dzh; g.amap = RowMap{T, (long)(T + 2 * pad) * Cout, (long)Cout};
        g.W = w2; g.ldw = (long)k * Cout;
        g.C = dxcl; g.cmap = RowMap{(int)rows, 0, (long)Cin};
        g.M = (int)rows; g.N = Cin; g.K = k * Cout; g.act = ACT_NONE;
        TR_TRY(launch_gemm(g, s));
        hipLaunchKernelGGL(to_channels_first_kernel, dim3(blocks_for(rows * Cin)), dim3(256), 0, s, dxcl, dx, B, Cin, T);
    }
    TR_TRY(hipGetLastError());
    return GVX_OK;
}

int gvx_tacotron2_loss_backward(const float* mel_out, const float* mel_post_out, const float* gate_out, const float* mel_target,
                                const float* gate_target, int B, int n_mels, int T, float* dmel_out, float* dmel_post_out,
                                float* dgate_out, void* stream) {
    if (!mel_out || !mel_post_out || !gate_out || !mel_target || !gate_target || !dmel_out || !dmel_post_out || !dgate_out)
        return tfail(GVX_ERR_INVALID_ARG, "null argument");
    if (B < 1 || n_mels < 1 || T < 1) return tfail(GVX_ERR_INVALID_ARG, "B, n_mels and T must be >= 1");
    const long n_mel = (long)B * n_mels * T, n_gate = (long)B * T;
    hipLaunchKernelGGL(loss_backward_kernel, dim3(blocks_for(n_mel)), dim3(256), 0, (hipStream_t)stream, mel_out, mel_post_out, gate_out,
                       mel_target, gate_target, n_mel, n_gate, dmel_out, dmel_post_out, dgate_out);
    TR_TRY(hipGetLastError());
    return GVX_OK;
}

}  // extern "C"

// =====================================================================================================================
// Back-propagation through time: generic primitives for the host side (genvox_amd/training.py), which walks the decoder
// loop and the encoder BiLSTM backwards exactly as oracle/train_ref.py states it.  All tensors row-major fp32 with an
// explicit leading dimension where slices are taken.  Correctness first (one fused kernel per formula group; the dense
// products go through the exact-fp32 MFMA GEMM).
// =====================================================================================================================
namespace gvx {
namespace {

// ---- LSTM cell backward (torch gate order i, f, g, o in blocks of H along the row) -----------------------------------------
// dh = dh_a[b][j] (+ dh_b[b][j]); h' = o tanh(c) * keep * scale.  pre = gate pre-activations [B][4H]; c_prev [B][H].
// active (may be null): rows with active[b] == 0 pass (dh, dc) through: dgates = 0, dc_prev = dc_next, dh_pass = dh.
__global__ void lstm_cell_bwd_kernel(const float* dh_a, long ld_a, const float* dh_b, long ld_b, const float* dc_next, const float* pre,
                                     const float* c_prev, const uint8_t* keep, float scale, const uint8_t* active, int B, int H,
                                     float* dgates, float* dc_prev, float* dh_pass) {
    const long n = (long)B * H;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i % H), b = (int)(i / H);
        float dh = dh_a[(long)b * ld_a + j];
        if (dh_b) dh += dh_b[(long)b * ld_b + j];
        const float dcn = dc_next[i];
        float* dg = dgates + (long)b * 4 * H;
        if (active && !active[b]) {
            dg[j] = dg[H + j] = dg[2 * H + j] = dg[3 * H + j] = 0.f;
            dc_prev[i] = dcn;
            if (dh_pass) dh_pass[i] = dh;
            continue;
        }
        const float* p = pre + (long)b * 4 * H;
        const float ig = 1.f / (1.f + expf(-p[j])), fg = 1.f / (1.f + expf(-p[H + j])), gg = tanhf(p[2 * H + j]), og = 1.f / (1.f + expf(-p[3 * H + j]));
        const float cp = c_prev[i];
        const float c = fg * cp + ig * gg, tc = tanhf(c);
        if (keep) dh = keep[i] ? dh * scale : 0.f;
        const float d_o = dh * tc;
        const float dc = dh * og * (1.f - tc * tc) + dcn;
        dg[j] = dc * gg * ig * (1.f - ig);
        dg[H + j] = dc * cp * fg * (1.f - fg);
        dg[2 * H + j] = dc * ig * (1.f - gg * gg);
        dg[3 * H + j] = d_o * og * (1.f - og);
        dc_prev[i] = dc * fg;
        if (dh_pass) dh_pass[i] = 0.f;
    }
}

// ---- attention step backward, part 1: one workgroup per batch row --------------------------------------------------------
// dw[l] = dw_next[l] + G[l] + sum_e dctx[e] memory[l][e];  dmemory[l][:] += w[l] dctx;  de[l] = w[l] (dw[l] - sum_l' w dw)
__global__ __launch_bounds__(256) void attn_bwd_weights_kernel(const float* dctx_a, long ld_a, const float* dctx_b, long ld_b, const float* dctx_c,
                                                               long ld_c, const float* dw_next, const float* G, const float* memory,
                                                               const float* w, int L, int E, float* dmemory, float* de, float* dctx_sum) {
    extern __shared__ float sm[];   // dctx [E], dw [L], red [256]
    float* dc = sm; float* dw = sm + E; float* red = dw + L;
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int e = tid; e < E; e += 256) {
        float v = dctx_a[(long)b * ld_a + e];
        if (dctx_b) v += dctx_b[(long)b * ld_b + e];
        if (dctx_c) v += dctx_c[(long)b * ld_c + e];
        dc[e] = v;
        if (dctx_sum) dctx_sum[(long)b * E + e] = v;
    }
    __syncthreads();
    const float* mb = memory + (long)b * L * E;
    float* dmb = dmemory + (long)b * L * E;
    const float* wb = w + (long)b * L;
    for (int l = tid >> 5; l < L; l += 8) {   // 8 groups of 32 lanes, one position each
        float acc = 0.f;
        const float wl = wb[l];
        for (int e = tid & 31; e < E; e += 32) {
            acc += dc[e] * mb[(long)l * E + e];
            dmb[(long)l * E + e] += wl * dc[e];
        }
        for (int o = 16; o > 0; o >>= 1) acc += __shfl_down(acc, o, 32);
        if ((tid & 31) == 0) dw[l] = acc + dw_next[(long)b * L + l] + G[(long)b * L + l];
    }
    __syncthreads();
    float part = 0.f;
    for (int l = tid; l < L; l += 256) part += wb[l] * dw[l];
    red[tid] = part;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float s = red[0];
    for (int l = tid; l < L; l += 256) de[(long)b * L + l] = wb[l] * (dw[l] - s);
}

// location convolution forward, channels-last: locf[(b,l)][f] = sum_{c,k} in_c[b][l + k - pad] lw[f][c][k]   (in_0 = w_prev, in_1 = w_cum_prev)
__global__ void loc_conv_fwd_kernel(const float* w_prev, const float* w_cum, const float* lw, int B, int L, int F, int k, float* locf) {
    const int pad = (k - 1) / 2;
    const long n = (long)B * L * F;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int f = (int)(i % F);
        const long bl = i / F;
        const int l = (int)(bl % L), b = (int)(bl / L);
        float acc = 0.f;
        for (int c = 0; c < 2; ++c) {
            const float* x = (c == 0 ? w_prev : w_cum) + (long)b * L;
            for (int j = 0; j < k; ++j) {
                const int p = l + j - pad;
                if (p >= 0 && p < L) acc += x[p] * lw[((long)f * 2 + c) * k + j];
            }
        }
        locf[i] = acc;
    }
}

// u = q[b] + locd[(b,l)] + pm[(b,l)]; th = tanh(u); du = de v (1 - th^2); dpm += du; per (b): dq[a] = sum_l du, dv_acc[b][a] += sum_l de th
__global__ __launch_bounds__(256) void attn_bwd_energy_kernel(const float* q, const float* locd, const float* pm, const float* v, const float* de,
                                                              int L, int a, float* du, float* dpm, float* dq, float* dv_acc) {
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int d = tid; d < a; d += 256) {
        const float qd = q[(long)b * a + d], vd = v[d];
        float sq = 0.f, sv = 0.f;
        for (int l = 0; l < L; ++l) {
            const long o = ((long)b * L + l) * a + d;
            const float th = tanhf(qd + locd[o] + pm[o]);
            const float e = de[(long)b * L + l];
            const float g = e * vd * (1.f - th * th);
            du[o] = g;
            dpm[o] += g;
            sq += g; sv += e * th;
        }
        dq[(long)b * a + d] = sq;
        dv_acc[(long)b * a + d] += sv;
    }
}

// location convolution backward: dloc_in[b][c][l] = sum_{f,j} dlocf[(b, l - j + pad)][f] lw[f][c][j];
// dlw_acc[b][f][c][j] += sum_l dlocf[(b,l)][f] in_c[b][l + j - pad]     (per-row accumulators: fixed summation order)
__global__ __launch_bounds__(256) void loc_conv_bwd_kernel(const float* dlocf, const float* w_prev, const float* w_cum, const float* lw, int L, int F,
                                                           int k, float* dw_prev_out, float* G, float* dlw_acc) {
    const int b = blockIdx.x, tid = threadIdx.x, pad = (k - 1) / 2;
    const float* dl = dlocf + (long)b * L * F;
    for (int i = tid; i < 2 * L; i += 256) {
        const int c = i / L, l = i - c * L;
        float acc = 0.f;
        for (int j = 0; j < k; ++j) {
            const int p = l - j + pad;
            if (p < 0 || p >= L) continue;
            for (int f = 0; f < F; ++f) acc += dl[(long)p * F + f] * lw[((long)f * 2 + c) * k + j];
        }
        if (c == 0) dw_prev_out[(long)b * L + l] = acc;
        else G[(long)b * L + l] += acc;
    }
    for (int i = tid; i < F * 2 * k; i += 256) {
        const int j = i % k, c = (i / k) % 2, f = i / (2 * k);
        const float* x = (c == 0 ? w_prev : w_cum) + (long)b * L;
        float acc = 0.f;
        for (int l = 0; l < L; ++l) {
            const int p = l + j - pad;
            if (p >= 0 && p < L) acc += dl[(long)l * F + f] * x[p];
        }
        dlw_acc[(long)b * F * 2 * k + i] += acc;
    }
}

// generic elementwise: y[r][c] = alpha * a[r][c] + beta * b[r][c]   (b may be null), each with its own leading dimension
__global__ void axpby_kernel(const float* a, long lda, float alpha, const float* b, long ldb, float beta, float* y, long ldy, long rows, int cols) {
    const long n = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cols);
        const long r = i / cols;
        float v = alpha * a[r * lda + c];
        if (b) v += beta * b[r * ldb + c];
        y[r * ldy + c] = v;
    }
}
// dz = dy * keep * scale * (act_out > 0)      (Prenet: relu then dropout)
__global__ void relu_drop_bwd_kernel(const float* dy, const float* act_out, const uint8_t* keep, float scale, long n, float* dz) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dz[i] = (keep[i] && act_out[i] > 0.f) ? dy[i] * scale : 0.f;
}
// k-group-blocked vector [K/8][B][8] -> row-major [B][K]  (slots: n_slots consecutive vectors)
__global__ void unblock_kernel(const float* src, float* dst, long n_slots, int B, int K) {
    const long n = n_slots * B * K;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const long sb = i / K;
        const int b = (int)(sb % B);
        const long s = sb / B;
        dst[i] = src[s * B * K + (long)(k >> 3) * B * 8 + b * 8 + (k & 7)];
    }
}
__global__ void embedding_bwd_kernel(const int64_t* tokens, const float* dx, long n_tok, int E, int n_rows, float* demb) {
    const long n = n_tok * E;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long t = i / E;
        const int e = (int)(i % E);
        const int64_t id = tokens[t];
        if (id >= 0 && id < n_rows) atomicAdd(demb + id * E + e, dx[i]);
    }
}
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* x, long n, double* acc) {
    __shared__ double red[256];
    double s = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += (double)x[i] * (double)x[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) atomicAdd(acc, red[0]);
}
// torch.optim.Adam (L2 weight decay folded into the gradient, bias-corrected), gradient pre-scaled by gscale (clipping)
__global__ void adam_kernel(float* p, const float* g, float* m, float* v, long n, float gscale, float lr, float wd, float b1, float b2, float eps,
                            float bc1, float bc2_sqrt) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale + wd * p[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}

}  // namespace
}  // namespace gvx

extern "C" {

// C[m][n] = sum_k A[m*lda + k] * W[n*ldw + k] (+ bias[n]);  K % 4 == 0
int gvx_train_gemm_nt(const float* A, long lda, const float* W, long ldw, float* C, long ldc, int M, int N, int K, const float* bias,
                      float* scratch, size_t scratch_bytes, void* stream) {
    if (!A || !W || !C || M < 1 || N < 1 || K < 4 || (K & 3)) return tfail(GVX_ERR_INVALID_ARG, "gemm_nt: null argument or K not a positive multiple of 4");
    GemmParams g{};
    g.A = A; g.amap = RowMap{M, 0, lda};
    g.W = W; g.ldw = ldw;
    g.C = C; g.cmap = RowMap{M, 0, ldc};
    g.bias = bias; g.M = M; g.N = N; g.K = K; g.act = ACT_NONE;
    // few output tiles and a long K (the per-step products of the backward: 32 rows x thousands of columns): split K over
    // enough workgroups to fill the chip, partial tiles in the caller's scratch, added in split order (deterministic)
    const long tiles = (long)((M + 63) / 64) * ((N + 127) / 128);
    int splitk = 1;
    if (scratch && tiles < 128 && K >= 512) {
        splitk = (int)((256 + tiles - 1) / tiles);
        if (splitk > K / 128) splitk = K / 128;
        while (splitk > 1 && (size_t)splitk * M * N * sizeof(float) > scratch_bytes) --splitk;
    }
    TR_TRY(launch_gemm_splitk(g, splitk, scratch, (hipStream_t)stream));
    return GVX_OK;
}
// dst[c][r] = src[r * ld_src + c] for r < rows (0 for rows <= r < rows_p);  dst rows are rows_p long
int gvx_train_transpose(const float* src, long ld_src, float* dst, long rows, int cols, long rows_p, void* stream) {
    if (!src || !dst || rows < 1 || cols < 1 || rows_p < rows) return tfail(GVX_ERR_INVALID_ARG, "transpose: bad argument");
    if (ld_src != cols) return tfail(GVX_ERR_UNSUPPORTED, "transpose: source must be dense (ld == cols)");
    hipLaunchKernelGGL(transpose_pad_kernel, dim3(blocks_for((long)cols * rows_p)), dim3(256), 0, (hipStream_t)stream, src, dst, rows, cols, rows_p);
    TR_TRY(hipGetLastError());
    return GVX_OK;
}
int gvx_train_colsum(const float* X, long rows, int C, float* out, void* stream) {
    if (!X || !out || rows < 1 || C < 1) return tfail(GVX_ERR_INVALID_ARG, "colsum: bad argument");
    hipLaunchKernelGGL(col_reduce_kernel, dim3((C + 31) / 32), dim3(256), 0, (hipStream_t)stream, X, (const float*)nullptr, rows, C, out, (float*)nullptr);
    TR_TRY(hipGetLastError());
    return GVX_OK;
}
int gvx_train_axpby(const float* a, long lda, float alpha, const float* b, long ldb, float beta, float* y, long ldy, long rows, int cols, void* stream) {
    if (!a || !y || rows < 1 || cols < 1) return tfail(GVX_ERR_INVALID_ARG, "axpby: bad argument");
    hipLaunchKernelGGL(axpby_kernel, dim3(blocks_for(rows * cols)), dim3(256), 0, (hipStream_t)stream, a, lda, alpha, b, ldb, beta, y, ldy, rows, cols);
    TR_TRY(hipGetLastError());
    return GVX_OK;
}
int gvx_train_lstm_cell_backward(const float* dh_a, long ld_a, const float* dh_b, long ld_b, const float* dc_next, const float* pre,
                                 const float* c_prev, const uint8_t* keep, float scale, const uint8_t* active, int B, int H, float* dgates,
                                 float* dc_prev, float* dh_pass, void* stream) {
    if (!dh_a || !dc_next || !pre || !c_prev || !dgates || !dc_prev || B < 1 || H < 1) return tfail(GVX_ERR_INVALID_ARG, "lstm_cell_backward: bad argument");
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(blocks_for((long)B * H)), dim3(256), 0, (hipStream_t)stream, dh_a, ld_a, dh_b, ld_b, dc_next, pre,
                       c_prev, keep, scale, active, B, H, dgates, dc_prev, dh_pass);
    TR_TRY(hipGetLastError());
    return GVX_OK;
}
int gvx_train_attention_weights_backward(const float* dctx_a, long ld_a, const float* dctx_b, long ld_b, const float* dctx_c, long ld_c,
                                         const float* dw_next, const float* G, const float* memory, const float* w, int B, int L, int E,
                                         float* dmemory, float* de, float* dctx_sum, void* stream) {
    if (!dctx_a || !dw_next || !G || !memory || !w || !dmemory || !de || B < 1 || L < 1 || E < 1) return tfail(GVX_ERR_INVALID_ARG, "attention_weights_backward: bad argument");
    const size_t lds = (size_t)(E + L + 256) * sizeof(float);
    if (lds > 64 * 1024) return tfail(GVX_ERR_UNSUPPORTED, "attention_weights_backward: L + E too large for one workgroup's LDS");
    hipLaunchKernelGGL(attn_bwd_weights_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, dctx_a, ld_a, dctx_b, ld_b, dctx_c, ld_c, dw_next, G,
                       memory, w, L, E, dmemory, de, dctx_sum);
    TR_TRY(hipGetLastError());
    return GVX_OK;
}
int gvx_train_location_conv_forward(const float* w_prev, const float* w_cum, const float* lw, int B, int L, int F, int k, float* locf, void* stream) {
    if (!w_prev || !w_cum || !lw || !locf || B < 1 || L < 1 || F < 1 || k < 1 || !(k & 1)) return tfail(GVX_ERR_INVALID_ARG, "location_conv_forward: bad argument");
    hipLaunchKernelGGL(loc_conv_fwd_kernel, dim3(blocks_for((long)B * L * F)), dim3(256), 0, (hipStream_t)stream, w_prev, w_cum, lw, B, L, F, k, locf);
    TR_TRY(hipGetLastError());
    return GVX_OK;
}
int gvx_train_attention_energy_backward(const float* q, const float* locd, const float* pm, const float* v, const float* de, int B, int L, int a,
                                        float* du, float* dpm, float* dq, float* dv_acc, void* stream) {
    if (!q || !locd || !pm || !v || !de || !du || !dpm || !dq || !dv_acc || B < 1 || L < 1 || a < 1) return tfail(GVX_ERR_INVALID_ARG, "attention_energy_backward: bad argument");
    hipLaunchKernelGGL(attn_bwd_energy_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, q, locd, pm, v, de, L, a, du, dpm, dq, dv_acc);
    TR_TRY(hipGetLastError());
    return GVX_OK;
}
int gvx_train_location_conv_backward(const float* dlocf, const float* w_prev, const float* w_cum, const float* lw, int B, int L, int F, int k,
                                     float* dw_prev_out, float* G, float* dlw_acc, void* stream) {
    if (!dlocf || !w_prev || !w_cum || !lw || !dw_prev_out || !G || !dlw_acc || B < 1) return tfail(GVX_ERR_INVALID_ARG, "location_conv_backward: bad argument");
    hipLaunchKernelGGL(loc_conv_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dlocf, w_prev, w_cum, lw, L, F, k, dw_prev_out, G, dlw_acc);
    TR_TRY(hipGetLastError());
    return GVX_OK;
}
int gvx_train_relu_dropout_backward(const float* dy, const float* act_out, const uint8_t* keep, float scale, long n, float* dz, void* stream) {
    if (!dy || !act_out || !keep || !dz || n < 1) return tfail(GVX_ERR_INVALID_ARG, "relu_dropout_backward: bad argument");
    hipLaunchKernelGGL(relu_drop_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, dy, act_out, keep, scale, n, dz);
    TR_TRY(hipGetLastError());
    return GVX_OK;
}
int gvx_train_unblock(const float* blocked, float* rows_out, long n_slots, int B, int K, void* stream) {
    if (!blocked || !rows_out || n_slots < 1 || B < 1 || K < 8 || (K & 7)) return tfail(GVX_ERR_INVALID_ARG, "unblock: bad argument");
    hipLaunchKernelGGL(unblock_kernel, dim3(blocks_for(n_slots * B * K)), dim3(256), 0, (hipStream_t)stream, blocked, rows_out, n_slots, B, K);
    TR_TRY(hipGetLastError());
    return GVX_OK;
}
int gvx_train_embedding_backward(const int64_t* tokens, const float* dx, long n_tokens_in_batch, int E, int n_rows, float* demb, void* stream) {
    if (!tokens || !dx || !demb || n_tokens_in_batch < 1 || E < 1 || n_rows < 1) return tfail(GVX_ERR_INVALID_ARG, "embedding_backward: bad argument");
    TR_TRY(hipMemsetAsync(demb, 0, (size_t)n_rows * E * sizeof(float), (hipStream_t)stream));
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(blocks_for(n_tokens_in_batch * E)), dim3(256), 0, (hipStream_t)stream, tokens, dx, n_tokens_in_batch, E, n_rows, demb);
    TR_TRY(hipGetLastError());
    return GVX_OK;
}
int gvx_train_sqnorm_accumulate(const float* x, long n, double* acc, void* stream) {
    if (!x || !acc || n < 1) return tfail(GVX_ERR_INVALID_ARG, "sqnorm: bad argument");
    hipLaunchKernelGGL(sqnorm_kernel, dim3(blocks_for(n) > 256 ? 256 : blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, n, acc);
    TR_TRY(hipGetLastError());
    return GVX_OK;
}
int gvx_train_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float grad_scale, float lr, float weight_decay,
                        float beta1, float beta2, float eps, int step, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 1 || step < 1) return tfail(GVX_ERR_INVALID_ARG, "adam_step: bad argument");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, grad_scale, lr, weight_decay,
                       beta1, beta2, eps, bc1, bc2s);
    TR_TRY(hipGetLastError());
    return GVX_OK;
}

}  // extern "C"
