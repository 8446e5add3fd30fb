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
