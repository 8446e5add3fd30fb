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
#include <cmath>
#include <cstdlib>

#include <cstdio>
#include <cstring>

namespace gvx {
namespace {

constexpr float BN_EPS_F = 1e-5f;
inline size_t up256(size_t x) { return (x + 255) / 256 * 256; }

// Split-K factor for a product with `tiles` output tiles of 64 x 128 and a long K (weight gradients: few outputs, thousands of
// rows to sum over): the tiles run in rounds of 256 (one per CU), so the time goes like ceil(tiles * s / 256) / s - the
// smallest s <= 8 that minimises it, pieces of at least 256 k.  1 = no split.
inline int choose_splitk(long tiles, int K) {
    if (tiles >= 256 || K < 512) return 1;
    int best = 1;
    double best_t = 1.0;   // (tiles < 256: one round at s = 1)
    for (int sp = 2; sp <= 8 && K / sp >= 256; ++sp) {
        const double t = (double)((tiles * sp + 255) / 256) / sp;
        if (t < best_t - 1e-9) { best_t = t; best = sp; }
    }
    return best;
}

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

// column sums over the rows of X [rows][C] (and of X * Y when Y != nullptr), double accumulation, fixed order.
// Workgroup = 32 columns x 32 row lanes; a lane walks its rows four at a time with independent partial sums, so that the
// loads of a pass are in flight together (the first version walked 8 lanes x rows / 8 dependent iterations: 290 us for the
// 6 400 x 4 096 gate-gradient matrices of a 32 x 200 step).
constexpr int CR_LANES = 32;
__global__ __launch_bounds__(1024) void col_reduce_kernel(const float* X, const float* Y, long rows, int C, float* sum_x, float* sum_xy) {
    __shared__ double sx[CR_LANES][33], sxy[CR_LANES][33];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
    if (c < C) {
        long r = rl;
        for (; r + 3 * CR_LANES < rows; r += 4 * CR_LANES) {
            const float x0 = X[r * C + c], x1 = X[(r + CR_LANES) * C + c], x2 = X[(r + 2 * CR_LANES) * C + c], x3 = X[(r + 3 * CR_LANES) * C + c];
            a0 += x0; a1 += x1; a2 += x2; a3 += x3;
            if (Y) {
                b0 += (double)x0 * (double)Y[r * C + c]; b1 += (double)x1 * (double)Y[(r + CR_LANES) * C + c];
                b2 += (double)x2 * (double)Y[(r + 2 * CR_LANES) * C + c]; b3 += (double)x3 * (double)Y[(r + 3 * CR_LANES) * C + c];
            }
        }
        for (; r < rows; r += CR_LANES) {
            const double x = X[r * C + c];
            a0 += x;
            if (Y) b0 += x * (double)Y[r * C + c];
        }
    }
    sx[rl][cl] = (a0 + a1) + (a2 + a3); sxy[rl][cl] = (b0 + b1) + (b2 + b3);
    __syncthreads();
    if (rl == 0 && c < C) {
        double ta = 0.0, tb = 0.0;
        for (int i = 0; i < CR_LANES; ++i) { ta += sx[i][cl]; tb += sxy[i][cl]; }
        sum_x[c] = (float)ta;
        if (Y && sum_xy) sum_xy[c] = (float)tb;
    }
}

// biased batch variance in double from the centred values (two passes keep it exact enough for invstd); also the running
// statistics update of torch.nn.BatchNorm1d (momentum 0.1, unbiased variance).  Same 32 x 32 walk as col_reduce_kernel.
__global__ __launch_bounds__(1024) void bn_stats_kernel(const float* Z, long rows, int C, float* mean, float* invstd, float* running_mean,
                                                        float* running_var, float momentum) {
    __shared__ double s1[CR_LANES][33];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (c < C) {
        long r = rl;
        for (; r + 3 * CR_LANES < rows; r += 4 * CR_LANES) {
            a0 += (double)Z[r * C + c]; a1 += (double)Z[(r + CR_LANES) * C + c];
            a2 += (double)Z[(r + 2 * CR_LANES) * C + c]; a3 += (double)Z[(r + 3 * CR_LANES) * C + c];
        }
        for (; r < rows; r += CR_LANES) a0 += (double)Z[r * C + c];
    }
    s1[rl][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    double m = 0.0;
    for (int i = 0; i < CR_LANES; ++i) m += s1[i][cl];
    m /= (double)rows;
    __syncthreads();
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    if (c < C) {
        long r = rl;
        for (; r + 3 * CR_LANES < rows; r += 4 * CR_LANES) {
            const double d0 = (double)Z[r * C + c] - m, d1 = (double)Z[(r + CR_LANES) * C + c] - m;
            const double d2 = (double)Z[(r + 2 * CR_LANES) * C + c] - m, d3 = (double)Z[(r + 3 * CR_LANES) * C + c] - m;
            v0 += d0 * d0; v1 += d1 * d1; v2 += d2 * d2; v3 += d3 * d3;
        }
        for (; r < rows; r += CR_LANES) { const double d = (double)Z[r * C + c] - m; v0 += d * d; }
    }
    s1[rl][cl] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    if (rl == 0 && c < C) {
        double var = 0.0;
        for (int i = 0; i < CR_LANES; ++i) var += s1[i][cl];
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

// dst[c][r] = src[r][c]  for r < rows; columns of dst are padded with zeros up to rows_p.  32 x 32 tiles through LDS: both the
// reads and the writes are row-contiguous.  grid (ceil(rows_p / 32), ceil(C / 32)), 256 threads
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float* src, float* dst, long rows, int C, long rows_p) {
    __shared__ float tile[32][33];
    const long r0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long r = r0 + ty + 8 * i;
        const int c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < rows && c < C) ? src[r * C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i;
        const long r = r0 + tx;
        if (c < C && r < rows_p) dst[(long)c * rows_p + r] = tile[tx][ty + 8 * i];
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
    size_t wk, w2, z, du, dz, dzh, dwk, dxcl, xcl2, dwk_part, ws_total;   // workspace
};
ConvTrainPlan conv_train_plan(int B, int Cin, int Cout, int T, int k) {
    const int pad = (k - 1) / 2;
    const long rows = (long)B * T;
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
    p.dwk = take((size_t)Cout * k * Cin);
    p.dxcl = take((size_t)rows * Cin);
    p.xcl2 = take((size_t)B * (T + 2 * pad) * Cin);
    p.dwk_part = take((size_t)8 * Cout * k * Cin);   // split-K partial tiles of the weight gradient (at most 8 splits)
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
    hipLaunchKernelGGL(bn_stats_kernel, dim3((Cout + 31) / 32), dim3(1024), 0, s, z, rows, Cout, mean, invstd, running_mean, running_var, 0.1f);
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
    const long rows = (long)B * T;
    const float* xhat = at<float>(saved, pl.xhat);
    const float* a = at<float>(saved, pl.a);
    const float* invstd = at<float>(saved, pl.invstd);
    float* du = at<float>(workspace, pl.du);
    hipLaunchKernelGGL(act_drop_bwd_kernel, dim3(blocks_for(rows * Cout)), dim3(256), 0, s, dy, keep, keep ? 1.f / (1.f - p_drop) : 1.f, act, a,
                       B, Cout, T, du);
    // dbeta = sum du, dgamma = sum du * xhat
    hipLaunchKernelGGL(col_reduce_kernel, dim3((Cout + 31) / 32), dim3(1024), 0, s, du, xhat, rows, Cout, dbeta, dgamma);
    float* dz = at<float>(workspace, pl.dz);
    float* dzh = at<float>(workspace, pl.dzh);
    TR_TRY(hipMemsetAsync(dzh, 0, (size_t)B * (T + 2 * pad) * Cout * sizeof(float), s));
    hipLaunchKernelGGL(bn_bwd_kernel, dim3(blocks_for(rows * Cout)), dim3(256), 0, s, du, xhat, gamma, invstd, dbeta, dgamma, B, Cout, T, pad, dz, dzh);
    hipLaunchKernelGGL(col_reduce_kernel, dim3((Cout + 31) / 32), dim3(1024), 0, s, dz, (const float*)nullptr, rows, Cout, dbias, (float*)nullptr);
    const float* xcl = at<float>(saved, pl.xcl);
    const float* xcl_w = xcl;
    if (x_wgrad) {   // (the reference masks the Postnet's input in place after its forward - outside autograd, so the first
                     // layer's weight gradient sees the MASKED input: models/tts/tacotron2.py:463, :466-473)
        float* x2 = at<float>(workspace, pl.xcl2);
        TR_TRY(launch_to_channels_last(x_wgrad, x2, B, Cin, T, pad, nullptr, s));
        xcl_w = x2;
    }
    float* dwk = at<float>(workspace, pl.dwk);
    {   // dwk[co][(j, ci)] = sum over the rows r = (b, t) of dz[r][co] * xcl[b][t + j][ci]: both operands K-major as they lie in
        // memory (the im2col row of r is the k * Cin contiguous floats at padded row t) - no transposed copies
        GemmParams g{};
        g.kmajor = true;
        g.A = dz; g.amap = RowMap{(int)rows, 0, (long)Cout};
        g.W = xcl_w; g.wmap = RowMap{T, (long)(T + 2 * pad) * Cin, (long)Cin};
        g.C = dwk; g.cmap = RowMap{Cout, 0, (long)k * Cin};
        g.M = Cout; g.N = k * Cin; g.K = (int)rows; g.act = ACT_NONE;
        // few output tiles, thousands of rows to sum over: K split over enough workgroups to fill the chip (the Postnet's 512 x 2560
        // gradients are 160 tiles, its first layer's 32: 290 / 330 us each as one round)
        const long tiles = (long)((Cout + 63) / 64) * ((k * Cin + 127) / 128);
        TR_TRY(launch_gemm_splitk(g, choose_splitk(tiles, (int)rows), at<float>(workspace, pl.dwk_part), s));
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
// Whole-sequence pieces of the backward pass: generic primitives for the host side (genvox_amd/training.py) - dense products
// on the exact-fp32 MFMA GEMM, transposes, column sums, elementwise updates, Adam.  All tensors row-major fp32 with an
// explicit leading dimension where slices are taken.  The two recurrences are at the end of this file.
// =====================================================================================================================
namespace gvx {
namespace {

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
// d embedding[row][e] = sum over the batch positions that hold token `row`, added in position order (no atomics: the result
// does not depend on the launch's scheduling).  One workgroup per table row, threads over the channels.  The positions that hold
// the row's token are first compacted, in order, into LDS (chunks of EMB_CHUNK tokens: every thread looks at a contiguous
// segment, an exclusive scan over the threads places its matches) - walking all positions one by one, as the first version
// did, took 0.32 ms for 4096 positions; the sums themselves then run over ~1 % of them with independent loads.
constexpr int EMB_CHUNK = 8192;
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int64_t* tokens, const float* dx, long n_tok, int E, int n_rows, float* demb) {
    __shared__ int list[EMB_CHUNK];
    __shared__ int cnt[257];
    const int row = blockIdx.x, tid = threadIdx.x;
    for (int pass = 0; pass * 1024 < E; ++pass) {   // (E <= 1024: one pass; every thread takes part in the barriers of every pass)
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const int e0 = pass * 1024 + tid, e1 = e0 + 256, e2 = e0 + 512, e3 = e0 + 768;
        for (long base = 0; base < n_tok; base += EMB_CHUNK) {
            const int n = (int)((n_tok - base) < EMB_CHUNK ? (n_tok - base) : EMB_CHUNK), seg = (n + 255) / 256;
            const int lo = tid * seg, hi = lo + seg < n ? lo + seg : n;
            int mine = 0;
            for (int t = lo; t < hi; ++t) mine += tokens[base + t] == row;
            __syncthreads();   // (the list of the previous chunk / pass has been consumed)
            cnt[tid + 1] = mine;
            if (tid == 0) cnt[0] = 0;
            __syncthreads();
            if (tid == 0) for (int i = 1; i <= 256; ++i) cnt[i] += cnt[i - 1];
            __syncthreads();
            int at = cnt[tid];
            for (int t = lo; t < hi; ++t) if (tokens[base + t] == row) list[at++] = t;
            __syncthreads();
            const int m = cnt[256];
            for (int k = 0; k < m; ++k) {   // position order; the loads of later positions do not wait for the adds
                const float* r = dx + (base + list[k]) * E;
                if (e0 < E) a0 += r[e0];
                if (e1 < E) a1 += r[e1];
                if (e2 < E) a2 += r[e2];
                if (e3 < E) a3 += r[e3];
            }
        }
        if (e0 < E) demb[(long)row * E + e0] = a0;
        if (e1 < E) demb[(long)row * E + e1] = a1;
        if (e2 < E) demb[(long)row * E + e2] = a2;
        if (e3 < E) demb[(long)row * E + e3] = a3;
    }
}
// sum of squares of MANY tensors in one launch: workgroup (x, tensor) writes its partial (double) to partials[tensor][x]; a
// second tiny launch adds all partials in index order - the total does not depend on the launch's scheduling (torch's
// clip_grad_norm_ on the reference side is a tree of its own; this one is at least reproducible)
constexpr int SQN_BLOCKS = 64;
__global__ __launch_bounds__(256) void sqnorm_many_kernel(const gvx_tensor_ref* refs, double* partials) {
    __shared__ double red[256];
    const gvx_tensor_ref r = refs[blockIdx.y];
    double s0 = 0.0, s1 = 0.0;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)SQN_BLOCKS * 256;
    for (; i + stride < r.numel; i += 2 * stride) {
        const double a = r.data[i], b = r.data[i + stride];
        s0 += a * a; s1 += b * b;
    }
    if (i < r.numel) { const double a = r.data[i]; s0 += a * a; }
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) partials[(long)blockIdx.y * SQN_BLOCKS + blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void sqnorm_finish_kernel(const double* partials, int n, double* out) {
    // 256 strided sums, then a tree: a fixed order (one thread walking all ~6 000 partials took 0.2 ms)
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = red[0];
}
// torch.optim.Adam (L2 weight decay folded into the gradient, bias-corrected), gradient pre-scaled by gscale (clipping),
// for MANY tensors in one launch: workgroup (x, tensor) walks its share of the tensor
__global__ void adam_many_kernel(const gvx_adam_ref* refs, float gscale, float lr, float wd, float b1, float b2, float eps, float bc1,
                                 float bc2_sqrt) {
    const gvx_adam_ref r = refs[blockIdx.y];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < r.numel; i += (long)gridDim.x * blockDim.x) {
        const float gi = r.grad[i] * gscale + wd * r.param[i];
        const float mi = b1 * r.exp_avg[i] + (1.f - b1) * gi;
        const float vi = b2 * r.exp_avg_sq[i] + (1.f - b2) * gi * gi;
        r.exp_avg[i] = mi; r.exp_avg_sq[i] = vi;
        r.param[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
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
    int splitk = scratch ? choose_splitk(tiles, K) : 1;
    while (splitk > 1 && (size_t)splitk * M * N * sizeof(float) > scratch_bytes) --splitk;
    TR_TRY(launch_gemm_splitk(g, splitk, scratch, (hipStream_t)stream));
    return GVX_OK;
}
// C[m][n] = sum_r A[r * lda + m] * Bm[r * ldb + n]: the weight-gradient form, both operands as they lie in memory
int gvx_train_gemm_tn(const float* A, long lda, const float* Bm, long ldb, float* C, long ldc, int M, int N, long rows, float* scratch,
                      size_t scratch_bytes, void* stream) {
    if (!A || !Bm || !C || M < 1 || N < 1 || rows < 1 || rows > (1L << 30)) return tfail(GVX_ERR_INVALID_ARG, "gemm_tn: bad argument");
    GemmParams g{};
    g.kmajor = true;
    g.A = A; g.amap = RowMap{(int)rows, 0, lda};
    g.W = Bm; g.wmap = RowMap{(int)rows, 0, ldb};
    g.C = C; g.cmap = RowMap{M, 0, ldc};
    g.M = M; g.N = N; g.K = (int)rows; g.act = ACT_NONE;
    const long tiles = (long)((M + 63) / 64) * ((N + 127) / 128);
    int splitk = scratch ? choose_splitk(tiles, (int)rows) : 1;
    while (splitk > 1 && (size_t)splitk * M * N * sizeof(float) > scratch_bytes) --splitk;
    TR_TRY(launch_gemm_splitk(g, splitk, scratch, (hipStream_t)stream));
    return GVX_OK;
}
// dst[c][r] = src[r * ld_src + c] for r < rows (0 for rows <= r < rows_p);  dst rows are rows_p long
int gvx_train_transpose(const float* src, long ld_src, float* dst, long rows, int cols, long rows_p, void* stream) {
    if (!src || !dst || rows < 1 || cols < 1 || rows_p < rows) return tfail(GVX_ERR_INVALID_ARG, "transpose: bad argument");
    if (ld_src != cols) return tfail(GVX_ERR_UNSUPPORTED, "transpose: source must be dense (ld == cols)");
    hipLaunchKernelGGL(transpose_pad_kernel, dim3((unsigned)((rows_p + 31) / 32), (cols + 31) / 32), dim3(256), 0, (hipStream_t)stream, src, dst, rows, cols, rows_p);
    TR_TRY(hipGetLastError());
    return GVX_OK;
}
int gvx_train_colsum(const float* X, long rows, int C, float* out, void* stream) {
    if (!X || !out || rows < 1 || C < 1) return tfail(GVX_ERR_INVALID_ARG, "colsum: bad argument");
    hipLaunchKernelGGL(col_reduce_kernel, dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, X, (const float*)nullptr, rows, C, out, (float*)nullptr);
    TR_TRY(hipGetLastError());
    return GVX_OK;
}
int gvx_train_axpby(const float* a, long lda, float alpha, const float* b, long ldb, float beta, float* y, long ldy, long rows, int cols, void* stream) {
    if (!a || !y || rows < 1 || cols < 1) return tfail(GVX_ERR_INVALID_ARG, "axpby: bad argument");
    hipLaunchKernelGGL(axpby_kernel, dim3(blocks_for(rows * cols)), dim3(256), 0, (hipStream_t)stream, a, lda, alpha, b, ldb, beta, y, ldy, rows, cols);
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
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, tokens, dx, n_tokens_in_batch, E, n_rows, demb);
    TR_TRY(hipGetLastError());
    return GVX_OK;
}
int gvx_train_sqnorm_many(const gvx_tensor_ref* refs_device, int n_tensors, double* scratch, double* sumsq_out, void* stream) {
    if (!refs_device || !scratch || !sumsq_out || n_tensors < 1) return tfail(GVX_ERR_INVALID_ARG, "sqnorm_many: bad argument");
    hipLaunchKernelGGL(sqnorm_many_kernel, dim3(SQN_BLOCKS, n_tensors), dim3(256), 0, (hipStream_t)stream, refs_device, scratch);
    hipLaunchKernelGGL(sqnorm_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, SQN_BLOCKS * n_tensors, sumsq_out);
    TR_TRY(hipGetLastError());
    return GVX_OK;
}
size_t gvx_train_sqnorm_scratch_bytes(int n_tensors) { return n_tensors < 1 ? 0 : (size_t)SQN_BLOCKS * n_tensors * sizeof(double); }
int gvx_train_adam_step_many(const gvx_adam_ref* refs_device, int n_tensors, float grad_scale, float lr, float weight_decay, float beta1,
                             float beta2, float eps, int step, void* stream) {
    if (!refs_device || n_tensors < 1 || step < 1) return tfail(GVX_ERR_INVALID_ARG, "adam_step_many: bad argument");
    // (bias corrections in double, as torch.optim.Adam computes them in Python: 1 - 0.999f in fp32 is 1.3e-5 off at step 1)
    const float bc1 = (float)(1.0 - std::pow((double)beta1, (double)step)), bc2s = (float)std::sqrt(1.0 - std::pow((double)beta2, (double)step));
    hipLaunchKernelGGL(adam_many_kernel, dim3(128, n_tensors), dim3(256), 0, (hipStream_t)stream, refs_device, grad_scale, lr, weight_decay, beta1, beta2,
                       eps, bc1, bc2s);
    TR_TRY(hipGetLastError());
    return GVX_OK;
}

}  // extern "C"

// =====================================================================================================================
// Back-propagation through the decoder loop in ONE call (gvx_train_decoder_bptt): three launches per step, issued from
// here - not ~20 primitives per step strung together by the host mirror (round 3, first version: 2.2 ms per step of
// launch overhead and small-kernel time).  Step t, walking down from T - 1:
//   A  bptt_attention_kernel  (G position chunks x B rows): gradient of the context -> attention weights -> energies ->
//      query / processed memory / location layers, and through the location convolution into the previous and cumulative
//      weights.  Everything of a row is local to its positions except a few sums over positions, which leave the kernel as
//      per-chunk partials that their consumers add in chunk order (deterministic, no atomics).
//   B  bptt_cells_kernel: attention-LSTM cell of step t and decoder-LSTM cell of step t - 1 backwards (the decoder cell of
//      step t - 1 only needs the decoder cell of step t: the two recurrences overlap exactly like in the forward loop).
//   C  the two products dgates x [W_ih | W_hh] on the weight-streaming skinny kernel of the forward path (skinny.hip,
//      mode 2) with the matrices transposed and packed in MFMA-fragment order on the device at the start of the call; K
//      is cut in two so that the default layer sizes give 256 equal tiles (48 + 80 row tiles x 2 K halves, 256 KB each).
// The softmax term sum_l w_l dw_l is taken as  w . (dw_next + G) + dctx . ctx(t)  (ctx(t) = sum_l w_l memory_l is on the
// tape): a chunk does not need the other chunks' dw.  The context path of the memory gradient, sum_t w_t (x) dctx_t, is one
// kernel after the loop.  What is not on the recurrence - the Prenet columns of the attention LSTM, all weight gradients -
// stays with the host mirror as whole-sequence GEMMs.
// =====================================================================================================================
namespace gvx {
namespace {

constexpr int BP_THREADS = 256;
#define TR_STAMP(flag, k, i) do { if (flag) GVX_STAMP(k, i); } while (0)
constexpr int BP_GMAX = 8;   // position chunks per batch row

inline int bptt_chunks(int L) { int g = (L + 3) / 4; return g < 1 ? 1 : (g > BP_GMAX ? BP_GMAX : g); }
inline int round32(int x) { return (x + 31) / 32 * 32; }

struct BpttAttn {
    int B, L, E, a, F, kl, G;
    const float* dhc_ctx; long dhc_ld;        // d loss / d ctx(t) through the projection: row b at dhc_ctx + b * dhc_ld
    const float* yd0; const float* yd1; int yd_ld, yd_ctx;   // decoder-cell products of step t (two K halves): context columns at yd_ctx
    const float* ya0; const float* ya1; int ya_ld;           // attention-cell products of step t + 1 (nullptr at t = T - 1): context columns at 0
    const float* ctx; long ctx_bs;            // ctx(t), row b at ctx + b * ctx_bs
    const float* w; const float* w_prev; const float* wcum;   // alignments of step t, t - 1 (nullptr at t = 0), cumulative before t: [B][L]
    const float* q;                           // [B][a] query of step t
    const float* memory; const float* pm; const float* v; const float* lw; const float* ld;
    const float* dw_in; const float* gc_in;   // [B][G][L] partials written by step t + 1
    float* dw_out; float* gc_out;             // [B][G][L] partials of this step
    float* dq_part;                           // [B][G][a]
    float* dctx_out;                          // [B][E] (dctx_all[t])
    float* dpm;                               // [B][L][a]  accumulated
    float* dv_acc; float* dld_acc; float* dlw_acc;   // [B][G][a], [B][G][a][F], [B][G][F * 2 * kl]  accumulated
    int stamp;                                // stamps build: this launch records its phase times
};

// LDS rows of the location filters are FS = 32 floats whatever F is (zeros past F): every loop over filters is a compile-time
// 32-iteration loop, its operands in registers / consecutive LDS words
constexpr int BA_THREADS = 512;
constexpr int BA_FS = 32;
inline size_t bptt_attn_lds_floats(int L, int E, int a, int F, int kl, int G) {
    const int CH = (L + G - 1) / G;
    return (size_t)E + 2 * L + BA_THREADS + 2 * (CH + kl - 1) + (size_t)2 * kl * (BA_FS + 1) + (size_t)a * (BA_FS + 1) + (size_t)CH * BA_FS + CH +
           (size_t)2 * CH * a + (size_t)CH * BA_FS + (size_t)CH * 2 * kl + 8;
}

__device__ __forceinline__ float fast_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f); }

__global__ __launch_bounds__(BA_THREADS) void bptt_attention_kernel(BpttAttn p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int L = p.L, E = p.E, a = p.a, F = p.F, kl = p.kl, G = p.G, pad = (kl - 1) / 2;
    constexpr int FS = BA_FS, LDF = BA_FS + 1;
    const int CH = (L + G - 1) / G, l0 = g * CH;
    const int n = max(0, min(L, l0 + CH) - l0);   // positions of this chunk (0: the chunk only passes its partial buffers on)
    const int LW = CH + kl - 1;
    float* locf = sm;                  // [CH][FS]        zeros past F          (16-byte aligned rows: read as float4)
    float* dlocf = locf + CH * FS;     // [CH][FS]
    float* dc = dlocf + CH * FS;       // [E]
    float* dwg = dc + E;               // [L]   dw_next + G of the whole row
    float* wrow = dwg + L;             // [L]   alignment of step t
    float* red = wrow + L;             // [BA_THREADS]
    float* win = red + BA_THREADS;     // [2][LW] previous / cumulative weights at positions l0 - pad ...
    float* lws = win + 2 * LW;         // [2 kl][FS + 1]  lw[f][c][j] at (c kl + j, f), zeros past F
    float* lds_ = lws + 2 * kl * LDF;  // [a][FS + 1]     zeros past F
    float* des = lds_ + a * LDF;       // [CH]
    float* du = des + CH;              // [CH][a]
    float* dvt = du + CH * a;          // [CH][a]
    float* t1 = dvt + CH * a;          // [CH][2][kl]
    TR_STAMP(p.stamp, 0, 0);

    // ---- every global operand that does not depend on this launch's arithmetic is requested up front (a dependent round trip
    // to memory the previous launch wrote costs ~1 us; the first version of this kernel had a dozen of them in a row)
    constexpr int NBE = 8, NBD = 2, NBC = 4;   // items per thread and batch: energies, dense-gradient groups, conv-gradient items
    const int n_en = n * a, n_dg = a * (FS / 8), n_cv = F * 2 * kl;
    float e_pm[NBE], e_dpm[NBE];
#pragma unroll
    for (int u = 0; u < NBE; ++u) {
        const int i = tid + u * BA_THREADS;
        e_pm[u] = e_dpm[u] = 0.f;
        if (i < n_en) { const int li = i / a, d = i - li * a; const long o = ((long)b * L + l0 + li) * a + d; e_pm[u] = p.pm[o]; e_dpm[u] = p.dpm[o]; }
    }
    float* dldb = p.dld_acc + ((long)b * G + g) * a * F;
    float d_old[NBD][8];
#pragma unroll
    for (int u = 0; u < NBD; ++u) {
        const int grp = tid + u * BA_THREADS, d = grp >> 2, f0 = (grp & 3) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) d_old[u][k] = (grp < n_dg && f0 + k < F) ? dldb[(long)d * F + f0 + k] : 0.f;
    }
    float* dlwb = p.dlw_acc + ((long)b * G + g) * F * 2 * kl;
    float c_old[NBC];
#pragma unroll
    for (int u = 0; u < NBC; ++u) { const int i = tid + u * BA_THREADS; c_old[u] = i < n_cv ? dlwb[i] : 0.f; }
    const long part_o = ((long)b * G + g) * L;
    float dv_old = 0.f, gq_ = 0.f, gv_ = 0.f;
    if (tid < a) { dv_old = p.dv_acc[((long)b * G + g) * a + tid]; }
    if (BA_THREADS % a == 0) { gq_ = p.q[(long)b * a + tid % a]; gv_ = p.v[tid % a]; }
    for (int e = tid; e < E; e += BA_THREADS) {
        float v = p.dhc_ctx[(long)b * p.dhc_ld + e] + p.yd0[(long)b * p.yd_ld + p.yd_ctx + e] + p.yd1[(long)b * p.yd_ld + p.yd_ctx + e];
        if (p.ya0) v += p.ya0[(long)b * p.ya_ld + e] + p.ya1[(long)b * p.ya_ld + e];
        dc[e] = v;
        if (g == 0) p.dctx_out[(long)b * E + e] = v;
    }
    for (int l = tid; l < L; l += BA_THREADS) {
        float pd[BP_GMAX], pg[BP_GMAX];
#pragma unroll
        for (int gg = 0; gg < BP_GMAX; ++gg) {
            pd[gg] = gg < G ? p.dw_in[((long)b * G + gg) * L + l] : 0.f;
            pg[gg] = gg < G ? p.gc_in[((long)b * G + gg) * L + l] : 0.f;
        }
        float s = 0.f;
#pragma unroll
        for (int gg = 0; gg < BP_GMAX; ++gg) s += pd[gg] + pg[gg];
        dwg[l] = s;
        wrow[l] = p.w[(long)b * L + l];
    }
    for (int i = tid; i < 2 * LW; i += BA_THREADS) {
        const int c = i / LW, ii = i - c * LW, l = l0 + ii - pad;
        float v = 0.f;
        if (l >= 0 && l < L) v = c == 0 ? (p.w_prev ? p.w_prev[(long)b * L + l] : 0.f) : p.wcum[(long)b * L + l];
        win[i] = v;
    }
    for (int i = tid; i < 2 * kl * FS; i += BA_THREADS) {   // lw [F][2][kl] -> rows (c, j), filters along the row
        const int f = i & (FS - 1), cj = i >> 5;
        lws[cj * LDF + f] = f < F ? p.lw[(long)f * 2 * kl + cj] : 0.f;
    }
    for (int i = tid; i < a * FS; i += BA_THREADS) {
        const int f = i & (FS - 1), d = i >> 5;
        lds_[d * LDF + f] = f < F ? p.ld[(long)d * F + f] : 0.f;
    }
    __syncthreads();
    TR_STAMP(p.stamp, 0, 1);
    // s = sum_l w_l dw_l = w . (dw_next + G) + dctx . ctx(t)
    {
        float part = 0.f;
        for (int e = tid; e < E; e += BA_THREADS) part += dc[e] * p.ctx[(long)b * p.ctx_bs + e];
        for (int l = tid; l < L; l += BA_THREADS) part += wrow[l] * dwg[l];
        red[tid] = part;
        __syncthreads();
        for (int o = BA_THREADS / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    }
    const float ssum = red[0];
    TR_STAMP(p.stamp, 0, 2);
    // dw and de of the chunk's positions: one wave per position, lanes over the memory channels
    {
        const int wave = tid >> 6, lane = tid & 63;
        for (int li = wave; li < n; li += BA_THREADS / 64) {
            const float* mrow = p.memory + ((long)b * L + l0 + li) * E;
            float acc = 0.f;
#pragma unroll 8
            for (int e = lane; e < E; e += 64) acc += dc[e] * mrow[e];
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
            if (lane == 0) des[li] = wrow[l0 + li] * (acc + dwg[l0 + li] - ssum);
        }
    }
    TR_STAMP(p.stamp, 0, 3);
    // location features of the chunk (recomputed): locf[l][f] = sum_{c,j} in_c[l + j - pad] lw[f][c][j]
    for (int i = tid; i < n * FS; i += BA_THREADS) {
        const int li = i >> 5, f = i & (FS - 1);
        float acc = 0.f;
        for (int c = 0; c < 2; ++c) {
#pragma unroll 8
            for (int j = 0; j < kl; ++j) acc += win[c * LW + li + j] * lws[(c * kl + j) * LDF + f];
        }
        locf[i] = acc;   // (filters past F: zero weights -> 0)
    }
    __syncthreads();
    TR_STAMP(p.stamp, 0, 4);
    // energies backwards: u = q + locf ld^T + pm, th = tanh(u), du = de v (1 - th^2).  A thread keeps the dense row of its
    // attention dim in registers while that dim does not change (a | 512: never)
    {
        float ldr[FS];
        int d_have = -1;
        float qd = gq_, vd = gv_;
        for (int i0 = tid; i0 < n_en; i0 += NBE * BA_THREADS) {
            if (i0 != tid) {   // later batches (more than 8 items per thread): their operands are requested here
#pragma unroll
                for (int u = 0; u < NBE; ++u) {
                    const int i = i0 + u * BA_THREADS;
                    if (i < n_en) { const int li = i / a, d = i - li * a; const long o = ((long)b * L + l0 + li) * a + d; e_pm[u] = p.pm[o]; e_dpm[u] = p.dpm[o]; }
                }
            }
#pragma unroll
            for (int u = 0; u < NBE; ++u) {
                const int i = i0 + u * BA_THREADS;
                if (i < n_en) {
                    const int li = i / a, d = i - li * a;
                    if (d != d_have) {
#pragma unroll
                        for (int f = 0; f < FS; ++f) ldr[f] = lds_[d * LDF + f];
                        if (BA_THREADS % a != 0) { qd = p.q[(long)b * a + d]; vd = p.v[d]; }
                        d_have = d;
                    }
                    float locd = 0.f;
#pragma unroll
                    for (int f = 0; f < FS; ++f) locd += locf[li * FS + f] * ldr[f];
                    const float th = fast_tanh(qd + locd + e_pm[u]);
                    const float e_ = des[li];
                    const float gq = e_ * vd * (1.f - th * th);
                    du[i] = gq;
                    dvt[i] = e_ * th;
                    p.dpm[((long)b * L + l0 + li) * a + d] = e_dpm[u] + gq;
                }
            }
        }
    }
    __syncthreads();
    TR_STAMP(p.stamp, 0, 5);
    for (int d = tid; d < a; d += BA_THREADS) {
        if (d != tid) dv_old = p.dv_acc[((long)b * G + g) * a + d];
        float sq = 0.f, sv = 0.f;
#pragma unroll 4
        for (int li = 0; li < n; ++li) { sq += du[li * a + d]; sv += dvt[li * a + d]; }
        p.dq_part[((long)b * G + g) * a + d] = sq;
        p.dv_acc[((long)b * G + g) * a + d] = dv_old + sv;
    }
    TR_STAMP(p.stamp, 0, 6);
    // d location_dense[d][f] += sum_l du[l][d] locf[l][f]: a thread owns 8 consecutive filters of one attention dim
    for (int g0 = tid; g0 < n_dg; g0 += NBD * BA_THREADS) {
#pragma unroll
        for (int u = 0; u < NBD; ++u) {
            const int grp = g0 + u * BA_THREADS;
            if (grp < n_dg) {
                const int d = grp >> 2, f0 = (grp & 3) * 8;
                if (g0 != tid) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) d_old[u][k] = f0 + k < F ? dldb[(long)d * F + f0 + k] : 0.f;
                }
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
                for (int li = 0; li < n; ++li) {
                    const float dv_ = du[li * a + d];
                    const float4 x0 = *reinterpret_cast<const float4*>(locf + li * FS + f0), x1 = *reinterpret_cast<const float4*>(locf + li * FS + f0 + 4);
                    acc[0] += dv_ * x0.x; acc[1] += dv_ * x0.y; acc[2] += dv_ * x0.z; acc[3] += dv_ * x0.w;
                    acc[4] += dv_ * x1.x; acc[5] += dv_ * x1.y; acc[6] += dv_ * x1.z; acc[7] += dv_ * x1.w;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (f0 + k < F) dldb[(long)d * F + f0 + k] = d_old[u][k] + acc[k];
            }
        }
    }
    TR_STAMP(p.stamp, 0, 7);
    for (int i = tid; i < n * FS; i += BA_THREADS) {   // dlocf[l][f] = sum_d du[l][d] ld[d][f]
        const int li = i >> 5, f = i & (FS - 1);
        float acc = 0.f;
#pragma unroll 8
        for (int d = 0; d < a; ++d) acc += du[li * a + d] * lds_[d * LDF + f];
        dlocf[i] = acc;
    }
    __syncthreads();
    TR_STAMP(p.stamp, 0, 8);
    for (int i = tid; i < n * 2 * kl; i += BA_THREADS) {   // t1[l][c][j] = sum_f dlocf[l][f] lw[f][c][j]
        const int li = i / (2 * kl), cj = i - li * 2 * kl;
        float acc = 0.f;
#pragma unroll
        for (int f = 0; f < FS; ++f) acc += dlocf[li * FS + f] * lws[cj * LDF + f];
        t1[i] = acc;
    }
    TR_STAMP(p.stamp, 0, 9);
    // d location_conv[f][c][j] += sum_l dlocf[l][f] in_c[l + j - pad]
    for (int i0 = tid; i0 < n_cv; i0 += NBC * BA_THREADS) {
#pragma unroll
        for (int u = 0; u < NBC; ++u) {
            const int i = i0 + u * BA_THREADS;
            if (i < n_cv) {
                if (i0 != tid) c_old[u] = dlwb[i];
                const int f = i / (2 * kl), cj = i - f * 2 * kl, c = cj / kl, j = cj - c * kl;
                float acc = 0.f;
#pragma unroll 4
                for (int li = 0; li < n; ++li) acc += dlocf[li * FS + f] * win[c * LW + li + j];
                dlwb[i] = c_old[u] + acc;
            }
        }
    }
    __syncthreads();
    TR_STAMP(p.stamp, 0, 10);
    // d in_c[l'] = sum over the chunk's l of t1[l][c][l' - l + pad]: c = 0 -> previous weights (next step's dw_next),
    // c = 1 -> cumulative weights (added to G for all earlier steps)
    for (int i = tid; i < 2 * L; i += BA_THREADS) {
        const int c = i / L, lt = i - c * L;
        const float old = c == 0 ? 0.f : p.gc_in[part_o + lt];
        float acc = 0.f;
        const int li_lo = max(0, lt + pad - (kl - 1) - l0), li_hi = min(n, lt + pad - l0 + 1);   // 0 <= lt - (l0 + li) + pad < kl
        for (int li = li_lo; li < li_hi; ++li) acc += t1[(li * 2 + c) * kl + lt - (l0 + li) + pad];
        if (c == 0) p.dw_out[part_o + lt] = acc;
        else p.gc_out[part_o + lt] = old + acc;
    }
    TR_STAMP(p.stamp, 0, 11);
}

struct BpttCells {
    int B, A, D, E, a, G;
    // attention cell of step t (att == 0: skipped)
    int att;
    const float* yd0; const float* yd1; int yd_ld;      // decoder-cell products of step t: h_a columns at 0, h_d columns at A + E
    const float* ya0; const float* ya1; int ya_ld;      // attention-cell products of step t + 1 (nullptr at t = T - 1): h_a columns at E
    const float* dq_part; const float* wq;              // [B][G][a], [a][A]
    const float* pre_a; const float* c_a; const uint8_t* keep_a; float scale_a;   // step t: [B][A][4], [B][A], [B][A]
    float* dc_a;                                        // [B][A] state
    float* dga; float* xa_blk; float* dq_out;           // dga_all[t] [B][4A], blocked copy, dq_all[t] [B][a]
    // decoder cell of step t - 1 (dec == 0: skipped)
    int dec; int have_yd;                               // have_yd == 0: no later step (t - 1 = T - 1)
    const float* dhc_hd; long dhc_ld;                   // dhc_all[t - 1], h_d columns at 0
    const float* pre_d; const float* c_d; const uint8_t* keep_d; float scale_d;
    float* dc_d;
    float* dgd; float* xd_blk;
};

__device__ __forceinline__ void lstm_cell_bwd_one(float dh, float dcn, float pi, float pf, float pg, float po, float cp, float& gi, float& gf,
                                                  float& gg_, float& go, float& dc_prev) {
    const float ig = 1.f / (1.f + expf(-pi)), fg = 1.f / (1.f + expf(-pf)), gg = tanhf(pg), og = 1.f / (1.f + expf(-po));
    const float c = fg * cp + ig * gg, tc = tanhf(c);
    const float d_o = dh * tc;
    const float dc = dh * og * (1.f - tc * tc) + dcn;
    gi = dc * gg * ig * (1.f - ig);
    gf = dc * cp * fg * (1.f - fg);
    gg_ = dc * ig * (1.f - gg * gg);
    go = d_o * og * (1.f - og);
    dc_prev = dc * fg;
}

__global__ __launch_bounds__(BP_THREADS) void bptt_cells_kernel(BpttCells p) {
    __shared__ float dq[256];
    const int b = blockIdx.y, tid = threadIdx.x, j = blockIdx.x * BP_THREADS + tid;
    const int A = p.A, D = p.D, B = p.B;
    const bool att_block = p.att && (int)blockIdx.x * BP_THREADS < A;   // block-uniform
    if (att_block) {
        for (int d = tid; d < p.a; d += BP_THREADS) {
            float s = 0.f;
            for (int g = 0; g < p.G; ++g) s += p.dq_part[((long)b * p.G + g) * p.a + d];
            dq[d] = s;
            if (blockIdx.x == 0) p.dq_out[(long)b * p.a + d] = s;
        }
        __syncthreads();
    }
    if (j < A) {
        if (!p.att) return;
        float dh = p.yd0[(long)b * p.yd_ld + j] + p.yd1[(long)b * p.yd_ld + j];
        if (p.ya0) dh += p.ya0[(long)b * p.ya_ld + p.E + j] + p.ya1[(long)b * p.ya_ld + p.E + j];
        float hq = 0.f;
#pragma unroll 16
        for (int d = 0; d < p.a; ++d) hq += dq[d] * p.wq[(long)d * A + j];
        dh += hq;
        dh = p.keep_a[(long)b * A + j] ? dh * p.scale_a : 0.f;
        const float4 pr = *reinterpret_cast<const float4*>(p.pre_a + ((long)b * A + j) * 4);
        float gi, gf, gg, go, dcp;
        lstm_cell_bwd_one(dh, p.dc_a[(long)b * A + j], pr.x, pr.y, pr.z, pr.w, p.c_a[(long)b * A + j], gi, gf, gg, go, dcp);
        p.dc_a[(long)b * A + j] = dcp;
        const float gv[4] = {gi, gf, gg, go};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = q * A + j;
            p.dga[(long)b * 4 * A + k] = gv[q];
            p.xa_blk[(long)(k >> 3) * B * 8 + b * 8 + (k & 7)] = gv[q];
        }
    } else if (j < A + D) {
        if (!p.dec) return;
        const int jd = j - A;
        float dh = p.dhc_hd[(long)b * p.dhc_ld + jd];
        if (p.have_yd) dh += p.yd0[(long)b * p.yd_ld + A + p.E + jd] + p.yd1[(long)b * p.yd_ld + A + p.E + jd];
        dh = p.keep_d[(long)b * D + jd] ? dh * p.scale_d : 0.f;
        const float4 pr = *reinterpret_cast<const float4*>(p.pre_d + ((long)b * D + jd) * 4);
        float gi, gf, gg, go, dcp;
        lstm_cell_bwd_one(dh, p.dc_d[(long)b * D + jd], pr.x, pr.y, pr.z, pr.w, p.c_d[(long)b * D + jd], gi, gf, gg, go, dcp);
        p.dc_d[(long)b * D + jd] = dcp;
        const float gv[4] = {gi, gf, gg, go};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = q * D + jd;
            p.dgd[(long)b * 4 * D + k] = gv[q];
            p.xd_blk[(long)(k >> 3) * B * 8 + b * 8 + (k & 7)] = gv[q];
        }
    }
}

// Transposed recurrent matrix in MFMA-fragment order: logical row n (< N, zero rows up to Np) = column col0 + n of
// [W_ih | W_hh] ([K][Kin], [K][H]), logical column k = gate row k (torch order).  out [Np/32][K/8][64][4]
__global__ void pack_transposed_frag_kernel(const float* w_ih, int Kin, const float* w_hh, int H, int col0, int N, int Np, int K, float* out) {
    const long total = (long)(Np / 32) * (K / 8) * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const long tk = i >> 6;
        const int kg = (int)(tk % (K / 8)), tile = (int)(tk / (K / 8));
        const int n = tile * 32 + (lane & 31), k0 = 8 * kg + 4 * (lane >> 5);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < N) {
            const int col = col0 + n;
            const float* src = col < Kin ? w_ih + col : w_hh + (col - Kin);
            const long ld = col < Kin ? Kin : H;
            v.x = src[(long)(k0 + 0) * ld]; v.y = src[(long)(k0 + 1) * ld]; v.z = src[(long)(k0 + 2) * ld]; v.w = src[(long)(k0 + 3) * ld];
        }
        reinterpret_cast<float4*>(out)[i] = v;
    }
}

// wcum[t][b][l] = sum_{s < t} w[s][b][l], added in ascending order like the forward loop does
__global__ void cumulative_weights_kernel(const float* w, int T, long BL, float* wcum) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < BL; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) { wcum[(long)t * BL + i] = s; s += w[(long)t * BL + i]; }
    }
}

// dmemory[b][l][e] = sum_t w[t][b][l] dctx[t][b][e]   (8 positions per workgroup)
__global__ __launch_bounds__(BP_THREADS) void memory_context_grad_kernel(const float* w, const float* dctx, int T, int B, int L, int E, float* dmemory) {
    const int b = blockIdx.y, l0 = blockIdx.x * 8, tid = threadIdx.x;
    for (int e = tid; e < E; e += BP_THREADS) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < T; ++t) {
            const float dv = dctx[((long)t * B + b) * E + e];
            const float* wr = w + ((long)t * B + b) * L;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += wr[min(l0 + i, L - 1)] * dv;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (l0 + i < L) dmemory[((long)b * L + l0 + i) * E + e] = acc[i];
    }
}

struct BpttPlan {   // float offsets into the workspace
    size_t wa_t, wd_t, xa, xd, ya, yd, dc_a, dc_d, dwp, gcp, dqp, wcum, dv_acc, dld_acc, dlw_acc, total;
    int Na, Nd, G;
};
BpttPlan bptt_plan(const gvx_bptt_decoder_args& a) {
    BpttPlan p{};
    p.Na = round32(a.E + a.A); p.Nd = round32(a.A + a.E + a.D); p.G = bptt_chunks(a.L);
    size_t o = 0;
    auto take = [&](size_t floats) { size_t r = o; o += (floats + 63) / 64 * 64; return r; };
    p.wa_t = take((size_t)p.Na * 4 * a.A);
    p.wd_t = take((size_t)p.Nd * 4 * a.D);
    p.xa = take((size_t)a.B * 4 * a.A);
    p.xd = take((size_t)a.B * 4 * a.D);
    p.ya = take((size_t)2 * a.B * p.Na);
    p.yd = take((size_t)2 * a.B * p.Nd);
    p.dc_a = take((size_t)a.B * a.A);
    p.dc_d = take((size_t)a.B * a.D);
    p.dwp = take((size_t)2 * a.B * p.G * a.L);
    p.gcp = take((size_t)2 * a.B * p.G * a.L);
    p.dqp = take((size_t)a.B * p.G * a.a);
    p.wcum = take((size_t)a.T * a.B * a.L);
    p.dv_acc = take((size_t)a.B * p.G * a.a);
    p.dld_acc = take((size_t)a.B * p.G * a.a * a.F);
    p.dlw_acc = take((size_t)a.B * p.G * a.F * 2 * a.kl);
    p.total = o;
    return p;
}

int check_bptt_args(const gvx_bptt_decoder_args* a) {
    if (!a) return tfail(GVX_ERR_INVALID_ARG, "decoder_bptt: null argument block");
    if (a->B < 1 || a->B > 32 || a->L < 1 || a->T < 1) return tfail(GVX_ERR_UNSUPPORTED, "decoder_bptt: 1 <= B <= 32, L >= 1, T >= 1");
    if (a->A < 8 || a->D < 8 || (a->A % 8) || (a->D % 8) || (a->E % 8) || (a->P % 4) || a->a < 1 || a->a > 256 || a->F < 1 || a->F > 32 || a->kl < 1 || !(a->kl & 1))
        return tfail(GVX_ERR_UNSUPPORTED, "decoder_bptt: unsupported layer sizes");
    const void* need[] = {a->dhc_all, a->pre_a, a->pre_d, a->c_a_all, a->c_d_all, a->att_keep, a->dec_keep, a->q_all, a->ctx_all, a->w_all, a->memory, a->pm,
                          a->w_ih_a, a->w_hh_a, a->w_ih_d, a->w_hh_d, a->wq, a->v, a->loc_conv, a->loc_dense, a->dga_all, a->dgd_all, a->dq_all,
                          a->dctx_all, a->dpm, a->dmemory, a->dv, a->dloc_dense, a->dloc_conv};
    for (const void* q : need)
        if (!q) return tfail(GVX_ERR_INVALID_ARG, "decoder_bptt: null pointer in the argument block");
    const size_t lds = bptt_attn_lds_floats(a->L, a->E, a->a, a->F, a->kl, bptt_chunks(a->L)) * sizeof(float);
    if (lds > 160 * 1024) return tfail(GVX_ERR_UNSUPPORTED, "decoder_bptt: a row's attention chunk does not fit the LDS (L or E too large)");
    return GVX_OK;
}

}  // namespace
}  // namespace gvx

extern "C" {

size_t gvx_train_decoder_bptt_workspace_bytes(const gvx_bptt_decoder_args* a) {
    if (check_bptt_args(a) != GVX_OK) return 0;
    return bptt_plan(*a).total * sizeof(float);
}

int gvx_train_decoder_bptt(const gvx_bptt_decoder_args* ap, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_bptt_args(ap);
    if (rc != GVX_OK) return rc;
    const gvx_bptt_decoder_args& a = *ap;
    const BpttPlan pl = bptt_plan(a);
    if (!workspace || workspace_bytes < pl.total * sizeof(float)) return tfail(GVX_ERR_WORKSPACE, "decoder_bptt: workspace too small");
    if (reinterpret_cast<uintptr_t>(workspace) & 255) return tfail(GVX_ERR_WORKSPACE, "decoder_bptt: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    float* ws = reinterpret_cast<float*>(workspace);
    const int B = a.B, L = a.L, T = a.T, A = a.A, D = a.D, E = a.E, P = a.P, G = pl.G, Na = pl.Na, Nd = pl.Nd;
    const int Ka = 4 * A, Kd = 4 * D;
    // (per call, not once per process: the attribute belongs to the current device)
    TR_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(bptt_attention_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t lds_attn = bptt_attn_lds_floats(L, E, a.a, a.F, a.kl, G) * sizeof(float);

    // ---- before the loop: transposed matrices in fragment order, cumulative weights, cleared state and accumulators
    float* wa_t = ws + pl.wa_t; float* wd_t = ws + pl.wd_t;
    hipLaunchKernelGGL(pack_transposed_frag_kernel, dim3(blocks_for((long)(Na / 32) * (Ka / 8) * 64)), dim3(256), 0, s, a.w_ih_a, P + E, a.w_hh_a, A, P,
                       E + A, Na, Ka, wa_t);
    hipLaunchKernelGGL(pack_transposed_frag_kernel, dim3(blocks_for((long)(Nd / 32) * (Kd / 8) * 64)), dim3(256), 0, s, a.w_ih_d, A + E, a.w_hh_d, D, 0,
                       A + E + D, Nd, Kd, wd_t);
    float* wcum = ws + pl.wcum;
    hipLaunchKernelGGL(cumulative_weights_kernel, dim3(blocks_for((long)B * L)), dim3(256), 0, s, a.w_all, T, (long)B * L, wcum);
    TR_TRY(hipMemsetAsync(ws + pl.dc_a, 0, (pl.dqp - pl.dc_a) * sizeof(float), s));               // dc_a, dc_d, dw / G partials (both parities)
    TR_TRY(hipMemsetAsync(ws + pl.dv_acc, 0, (pl.total - pl.dv_acc) * sizeof(float), s));         // dv, dld, dlw accumulators
    TR_TRY(hipMemsetAsync(a.dpm, 0, (size_t)B * L * a.a * sizeof(float), s));
    float* ya0 = ws + pl.ya; float* ya1 = ya0 + (size_t)B * Na;
    float* yd0 = ws + pl.yd; float* yd1 = yd0 + (size_t)B * Nd;
    float* xa = ws + pl.xa; float* xd = ws + pl.xd;
    const size_t part = (size_t)B * G * L;

    // ---- slot t = T ... 0: attention chain of step t (t < T) and decoder cell of step t - 1 (t > 0)
    for (int t = T; t >= 0; --t) {
        const bool att = t < T, dec = t > 0;
        const int par = t & 1;   // partial buffers: step t reads parity (t + 1) & 1, writes parity t & 1
        if (att) {
            BpttAttn q{};
            q.B = B; q.L = L; q.E = E; q.a = a.a; q.F = a.F; q.kl = a.kl; q.G = G;
            q.dhc_ctx = a.dhc_all + (size_t)t * B * (D + E) + D; q.dhc_ld = D + E;
            q.yd0 = yd0; q.yd1 = yd1; q.yd_ld = Nd; q.yd_ctx = A;
            q.ya0 = t < T - 1 ? ya0 : nullptr; q.ya1 = ya1; q.ya_ld = Na;
            q.ctx = a.ctx_all + (long)t * (long)a.ctx_ts; q.ctx_bs = (long)a.ctx_bs;
            q.w = a.w_all + (size_t)t * B * L;
            q.w_prev = t > 0 ? a.w_all + (size_t)(t - 1) * B * L : nullptr;
            q.wcum = wcum + (size_t)t * B * L;
            q.q = a.q_all + (size_t)t * B * a.a;
            q.memory = a.memory; q.pm = a.pm; q.v = a.v; q.lw = a.loc_conv; q.ld = a.loc_dense;
            q.dw_in = ws + pl.dwp + (size_t)(par ^ 1) * part; q.gc_in = ws + pl.gcp + (size_t)(par ^ 1) * part;
            q.dw_out = ws + pl.dwp + (size_t)par * part; q.gc_out = ws + pl.gcp + (size_t)par * part;
            q.dq_part = ws + pl.dqp;
            q.dctx_out = a.dctx_all + (size_t)t * B * E;
            q.dpm = a.dpm; q.dv_acc = ws + pl.dv_acc; q.dld_acc = ws + pl.dld_acc; q.dlw_acc = ws + pl.dlw_acc;
            q.stamp = t == T / 2;
            hipLaunchKernelGGL(bptt_attention_kernel, dim3(G, B), dim3(BA_THREADS), lds_attn, s, q);
        }
        {
            BpttCells c{};
            c.B = B; c.A = A; c.D = D; c.E = E; c.a = a.a; c.G = G;
            c.att = att ? 1 : 0; c.dec = dec ? 1 : 0;
            c.yd0 = yd0; c.yd1 = yd1; c.yd_ld = Nd;
            c.ya0 = t < T - 1 ? ya0 : nullptr; c.ya1 = ya1; c.ya_ld = Na;
            c.dq_part = ws + pl.dqp; c.wq = a.wq;
            c.dc_a = ws + pl.dc_a; c.xa_blk = xa; c.scale_a = a.att_scale;
            if (att) {
                c.pre_a = a.pre_a + (size_t)t * B * Ka; c.c_a = a.c_a_all + (size_t)t * B * A; c.keep_a = a.att_keep + (size_t)t * B * A;
                c.dga = a.dga_all + (size_t)t * B * Ka; c.dq_out = a.dq_all + (size_t)t * B * a.a;
            }
            c.have_yd = t < T ? 1 : 0;
            c.dc_d = ws + pl.dc_d; c.xd_blk = xd; c.scale_d = a.dec_scale;
            if (dec) {
                c.dhc_hd = a.dhc_all + (size_t)(t - 1) * B * (D + E); c.dhc_ld = D + E;
                c.pre_d = a.pre_d + (size_t)(t - 1) * B * Kd; c.c_d = a.c_d_all + (size_t)(t - 1) * B * D; c.keep_d = a.dec_keep + (size_t)(t - 1) * B * D;
                c.dgd = a.dgd_all + (size_t)(t - 1) * B * Kd;
            }
            hipLaunchKernelGGL(bptt_cells_kernel, dim3((A + D + BP_THREADS - 1) / BP_THREADS, B), dim3(BP_THREADS), 0, s, c);
        }
        if (t == 0) break;   // the products of step 0's attention cell feed nothing on the recurrence
        {
            SkinnyJob jobs[4];
            std::memset(jobs, 0, sizeof jobs);
            int nj = 0;
            auto add = [&](const float* wt, const float* x, int K, int N, float* y0, float* y1) {
                const int nkg = K / 8, h0 = (nkg / 2 + 0), h1 = nkg - h0;
                const int kg0[2] = {0, h0}, nk[2] = {h0, h1};
                float* ys[2] = {y0, y1};
                for (int hh = 0; hh < 2; ++hh) {
                    SkinnyJob& J = jobs[nj++];
                    J.Wp = wt; J.N = N; J.nkg = nk[hh]; J.kg0 = kg0[hh]; J.nkg_w = nkg; J.mode = 2; J.B = B;
                    J.x[0] = XSeg{x + (size_t)kg0[hh] * B * 8, nk[hh] * 8};
                    J.y = ys[hh];
                }
            };
            if (att) add(wa_t, xa, Ka, Na, ya0, ya1);
            add(wd_t, xd, Kd, Nd, yd0, yd1);
            TR_TRY(launch_skinny(jobs, nj, SK_TRAIN, s));
        }
    }
    TR_TRY(hipGetLastError());
    // ---- after the loop: per-chunk accumulators summed in (row, chunk) order; context path of the memory gradient
    hipLaunchKernelGGL(col_reduce_kernel, dim3((a.a + 31) / 32), dim3(1024), 0, s, ws + pl.dv_acc, (const float*)nullptr, (long)B * G, a.a, a.dv, (float*)nullptr);
    hipLaunchKernelGGL(col_reduce_kernel, dim3((a.a * a.F + 31) / 32), dim3(1024), 0, s, ws + pl.dld_acc, (const float*)nullptr, (long)B * G, a.a * a.F,
                       a.dloc_dense, (float*)nullptr);
    hipLaunchKernelGGL(col_reduce_kernel, dim3((a.F * 2 * a.kl + 31) / 32), dim3(1024), 0, s, ws + pl.dlw_acc, (const float*)nullptr, (long)B * G,
                       a.F * 2 * a.kl, a.dloc_conv, (float*)nullptr);
    hipLaunchKernelGGL(memory_context_grad_kernel, dim3((L + 7) / 8, B), dim3(BP_THREADS), 0, s, a.w_all, a.dctx_all, T, B, L, E, a.dmemory);
    TR_TRY(hipGetLastError());
    return GVX_OK;
}

}  // extern "C"

// =====================================================================================================================
// Back-propagation through the encoder BiLSTM in one call (gvx_train_encoder_lstm_bptt: one launch per time step for both
// directions; gvx_train_encoder_lstm_bptt_resident: the same walk as ONE resident launch), packed-sequence semantics as in the forward (a row takes part in step s while s < its length; the reverse
// direction walks each row from its own last token).  Workgroup = (4 hidden units, direction); thread = (batch row, lane
// r of 8) - the 8 lanes of a row split the K of every dot product and combine with DPP-free shuffles.  Launch s first
// finishes dh(s) = dgates(s + 1) W_hh + pass-through for its units (the previous launch wrote dgates(s + 1)), recomputes
// the gate pre-activations from the tape (x-projection + h_prev W_hh^T) and runs the cell backwards.
// =====================================================================================================================
namespace gvx {
namespace {

constexpr int EB_UJ = 4;   // hidden units per workgroup

struct EncBptt {
    int B, L, H, s;
    const float* xg;         // [2][B][L][4H]  W_ih x + b_ih + b_hh per direction, torch gate order
    const float* memory;     // [B][L][2H]     BiLSTM outputs (forward direction in channels [0, H))
    const float* c_enc;      // [B][L][2H]     cell states
    const float* dmemory;    // [B][L][2H]     d loss / d memory
    const float* w_hh;       // [2][4H][H]
    const float* w_hh_t;     // [2][H][4H]
    const int32_t* lengths;  // [B]
    const float* dg_in; const float* dpass_in;   // [2][B][4H], [2][B][H] written by step s + 1
    float* dg_out; float* dpass_out;
    float* dc;               // [2][B][H] state
    float* dg_pos;           // [2][B][L][4H]  gate gradients filed under the position they belong to (zeros elsewhere)
    float* hprev_pos;        // [2][B][L][H]   the previous hidden state of that position
    int stamp;
};

// staging of a workgroup's weights (EB_UJ columns and 4 EB_UJ rows of W_hh) into LDS
__device__ __forceinline__ void enc_bptt_stage(const EncBptt& p, float* sm, int dir, int j0, int tid) {
    const int H = p.H, H4 = 4 * H;
    const float* whh = p.w_hh + (size_t)dir * H4 * H;
    const float* whht = p.w_hh_t + (size_t)dir * H * H4;
    // staging: rows of 4H / H floats are contiguous on both sides -> 16-byte pieces, all requested before the first LDS store
    {
        constexpr int NV = 8;
        const int nc4 = EB_UJ * H4 / 4, nr4 = 4 * EB_UJ * H / 4;
        for (int i0 = tid; i0 < nc4 + nr4; i0 += NV * 256) {
            float4 v[NV];
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int i = i0 + u * 256;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < nc4) {
                    const int jl = i / (H4 / 4), n4 = i - jl * (H4 / 4);
                    if (j0 + jl < H) v[u] = reinterpret_cast<const float4*>(whht + (size_t)(j0 + jl) * H4)[n4];
                } else if (i < nc4 + nr4) {
                    const int ii = i - nc4, qj = ii / (H / 4), k4 = ii - qj * (H / 4), jl = qj % EB_UJ, q = qj / EB_UJ;
                    if (j0 + jl < H) v[u] = reinterpret_cast<const float4*>(whh + ((size_t)q * H + j0 + jl) * H)[k4];
                }
            }
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int i = i0 + u * 256;
                if (i < nc4 + nr4) reinterpret_cast<float4*>(sm)[i] = v[u];   // (wrow starts right behind wcol)
            }
        }
    }
    __syncthreads();
}

// One time step of the walk for this workgroup's units.  RES: the step runs inside the resident kernel - the vectors other
// workgroups wrote in the previous step of the same launch (dg_in) and this workgroup's own state words are read and written
// write-through / past the L1 (sc1), as every handed-off byte of the resident loops is.
template <bool RES>
__device__ __forceinline__ void enc_bptt_step(const EncBptt& p, const float* sm, int dir, int j0, int tid) {
    const int H = p.H, H4 = 4 * H, L = p.L, B = p.B;
    const int r = tid & 7, b0 = tid >> 3;
    const float* wcol = sm;                    // [UJ][4H]  column j of W_hh = row j of its transpose
    const float* wrow = sm + EB_UJ * H4;       // [4][UJ][H] rows (q H + j) of W_hh
    const __amdgpu_buffer_rsrc_t r_dgi = make_rsrc(p.dg_in), r_dgo = make_rsrc(p.dg_out), r_pi = make_rsrc(p.dpass_in),
                                 r_po = make_rsrc(p.dpass_out), r_dc = make_rsrc(p.dc);
    // thread = (batch row, lane r of 8).  The 8 lanes of a row split K in float4 pieces: lane r takes the floats
    // 32 i + 4 r ... + 3, so that a row's 8 lanes read 128 contiguous bytes per instruction (global and LDS alike)
    const bool vec_h = (H & 31) == 0;
    for (int b = b0; b < B; b += 32) {
        const int len = p.lengths[b];
        const bool active = p.s < len;
        const int t_idx = dir == 0 ? p.s : max(len - 1 - p.s, 0);
        const int p_idx = dir == 0 ? t_idx - 1 : t_idx + 1;
        const bool has_prev = active && p.s > 0;
        const float* hp = p.memory + ((size_t)b * L + min(max(p_idx, 0), L - 1)) * 2 * H + dir * H;
        const float* dgi = p.dg_in + ((size_t)dir * B + b) * H4;
        const unsigned dgi_off = (unsigned)(((size_t)dir * B + b) * H4 * sizeof(float));
        // operands of the cell (lanes r < UJ own unit j0 + r): requested now, used after the dot products
        const int j = j0 + r;
        const bool own = r < EB_UJ && j < H;
        const size_t sj = ((size_t)dir * B + b) * H + (own ? j : 0);
        float o_pass = 0.f, o_dmem = 0.f, o_dc = 0.f, o_cp = 0.f, o_hp = 0.f, o_x0 = 0.f, o_x1 = 0.f, o_x2 = 0.f, o_x3 = 0.f;
        if (own) {
            if (RES) { o_pass = load_sc1_f32(r_pi, (unsigned)(sj * 4)); o_dc = load_sc1_f32(r_dc, (unsigned)(sj * 4)); }
            else { o_pass = p.dpass_in[sj]; o_dc = p.dc[sj]; }
            if (active) {
                o_dmem = p.dmemory[((size_t)b * L + t_idx) * 2 * H + dir * H + j];
                const float* xg = p.xg + (((size_t)dir * B + b) * L + t_idx) * H4;
                o_x0 = xg[j]; o_x1 = xg[H + j]; o_x2 = xg[2 * H + j]; o_x3 = xg[3 * H + j];
                if (has_prev) { o_cp = p.c_enc[((size_t)b * L + p_idx) * 2 * H + dir * H + j]; o_hp = hp[j]; }
            }
        }
        float sdh[EB_UJ], sp[4][EB_UJ];
#pragma unroll
        for (int jl = 0; jl < EB_UJ; ++jl) { sdh[jl] = 0.f; sp[0][jl] = sp[1][jl] = sp[2][jl] = sp[3][jl] = 0.f; }
        constexpr int NX = 16;   // float4 pieces of the row requested together
        for (int ib = 0; ib < H4 / 32; ib += NX) {
            float4 x[NX];
#pragma unroll
            for (int u = 0; u < NX; ++u)
                x[u] = ib + u >= H4 / 32 ? make_float4(0.f, 0.f, 0.f, 0.f)
                       : (RES ? load_sc1(r_dgi, dgi_off + (unsigned)(32 * (ib + u) + 4 * r) * 4u) : *reinterpret_cast<const float4*>(dgi + 32 * (ib + u) + 4 * r));
#pragma unroll
            for (int u = 0; u < NX; ++u) {
                if (ib + u < H4 / 32) {
#pragma unroll
                    for (int jl = 0; jl < EB_UJ; ++jl) {
                        const float4 w = *reinterpret_cast<const float4*>(wcol + jl * H4 + 32 * (ib + u) + 4 * r);
                        sdh[jl] += x[u].x * w.x + x[u].y * w.y + x[u].z * w.z + x[u].w * w.w;
                    }
                }
            }
        }
        TR_STAMP(p.stamp, 1, 2);
        if (has_prev) {
            if (vec_h) {
                constexpr int NHX = 8;
                for (int ib = 0; ib < H / 32; ib += NHX) {
                    float4 x[NHX];
#pragma unroll
                    for (int u = 0; u < NHX; ++u) x[u] = ib + u < H / 32 ? *reinterpret_cast<const float4*>(hp + 32 * (ib + u) + 4 * r) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int u = 0; u < NHX; ++u) {
                        if (ib + u < H / 32) {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
#pragma unroll
                                for (int jl = 0; jl < EB_UJ; ++jl) {
                                    const float4 w = *reinterpret_cast<const float4*>(wrow + (q * EB_UJ + jl) * H + 32 * (ib + u) + 4 * r);
                                    sp[q][jl] += x[u].x * w.x + x[u].y * w.y + x[u].z * w.z + x[u].w * w.w;
                                }
                        }
                    }
                }
            } else {
                for (int k = r; k < H; k += 8) {
                    const float hv = hp[k];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int jl = 0; jl < EB_UJ; ++jl) sp[q][jl] += hv * wrow[(q * EB_UJ + jl) * H + k];
                }
            }
        }
        TR_STAMP(p.stamp, 1, 3);
        float my_dh = 0.f, my_pre[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jl = 0; jl < EB_UJ; ++jl) {
            float v0 = sdh[jl], v1 = sp[0][jl], v2 = sp[1][jl], v3 = sp[2][jl], v4 = sp[3][jl];
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                v0 += __shfl_xor(v0, o, 64); v1 += __shfl_xor(v1, o, 64); v2 += __shfl_xor(v2, o, 64);
                v3 += __shfl_xor(v3, o, 64); v4 += __shfl_xor(v4, o, 64);
            }
            if (r == jl) { my_dh = v0; my_pre[0] = v1; my_pre[1] = v2; my_pre[2] = v3; my_pre[3] = v4; }
        }
        TR_STAMP(p.stamp, 1, 4);
        if (own) {
            const float dh = my_dh + o_pass + o_dmem;
            float* dgo = p.dg_out + ((size_t)dir * B + b) * H4;
            const unsigned dgo_off = (unsigned)((((size_t)dir * B + b) * H4 + j) * sizeof(float));
            auto put = [&](float* ptr, __amdgpu_buffer_rsrc_t rs, unsigned off, float v) {
                if (RES) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (int)off, 0, 16); else *ptr = v;
            };
            if (!active) {
                for (int q = 0; q < 4; ++q) put(dgo + q * H + j, r_dgo, dgo_off + (unsigned)(q * H) * 4u, 0.f);
                put(p.dpass_out + sj, r_po, (unsigned)(sj * 4), dh);   // (dc stays)
            } else {
                float gi, gf, gg, go, dcp;
                lstm_cell_bwd_one(dh, o_dc, o_x0 + my_pre[0], o_x1 + my_pre[1], o_x2 + my_pre[2], o_x3 + my_pre[3], o_cp, gi, gf, gg, go, dcp);
                put(p.dc + sj, r_dc, (unsigned)(sj * 4), dcp);
                put(p.dpass_out + sj, r_po, (unsigned)(sj * 4), 0.f);
                put(dgo + j, r_dgo, dgo_off, gi); put(dgo + H + j, r_dgo, dgo_off + (unsigned)H * 4u, gf);
                put(dgo + 2 * H + j, r_dgo, dgo_off + (unsigned)(2 * H) * 4u, gg); put(dgo + 3 * H + j, r_dgo, dgo_off + (unsigned)(3 * H) * 4u, go);
                float* dgp = p.dg_pos + (((size_t)dir * B + b) * L + t_idx) * H4;
                dgp[j] = gi; dgp[H + j] = gf; dgp[2 * H + j] = gg; dgp[3 * H + j] = go;
                p.hprev_pos[(((size_t)dir * B + b) * L + t_idx) * H + j] = o_hp;
            }
        }
    }
}

__global__ __launch_bounds__(256) void encoder_bptt_step_kernel(EncBptt p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int dir = blockIdx.y, j0 = blockIdx.x * EB_UJ, tid = threadIdx.x;
    TR_STAMP(p.stamp, 1, 0);
    enc_bptt_stage(p, sm, dir, j0, tid);
    TR_STAMP(p.stamp, 1, 1);
    enc_bptt_step<false>(p, sm, dir, j0, tid);
    TR_STAMP(p.stamp, 1, 5);
}

// The whole walk as ONE resident launch (the launch per time step: 128 x 22 us for ~2 us of work each, and at the end of a
// training step it runs alone on the GPU).  Same grid, same arithmetic in the same order; the weights are staged once; step s
// starts when every workgroup of the direction has published step s + 1 (one flag word per workgroup, each on a 128-byte line
// of its own, value = steps published; stores drained and a barrier in front of the flag, cdna_hip_programming.md guideline
// 16).  Parity buffers as in the launch-per-step walk: step s writes what step s + 1's readers have left - they all published
// s + 1 before anybody could start s.  Every wait is bounded: after a time-out all waits return at once, the grid drains and
// the caller's last launch overwrites the outputs with NaN and raises the status word of the workspace.
struct EncBpttRes {
    EncBptt q;               // (s, dg_in / dg_out, dpass_in / dpass_out are set per step inside)
    float* dg; float* dpass; // [2 parities][2][B][4H] / [2 parities][2][B][H]
    unsigned* flags;         // [2][nwg] x 32 words
    unsigned* tmo;           // the call's time-out word
    unsigned spin_limit;
    int nwg;                 // workgroups per direction
    int debug_skip_block;    // tests: this workgroup leaves at once
};
__global__ __launch_bounds__(256) void encoder_bptt_resident_kernel(EncBpttRes a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int dir = blockIdx.y, j0 = blockIdx.x * EB_UJ, tid = threadIdx.x;
    if ((int)(blockIdx.y * gridDim.x + blockIdx.x) == a.debug_skip_block) return;   // (uniform per workgroup)
    EncBptt p = a.q;
    enc_bptt_stage(p, sm, dir, j0, tid);
    const int B = p.B, H = p.H, L = p.L;
    const unsigned* fl = a.flags + (size_t)dir * a.nwg * 32;
    const unsigned limit = (a.spin_limit ? a.spin_limit : HANDOFF_SPIN_LIMIT) * 16u;
    for (int st = L - 1; st >= 0; --st) {
        const int par = st & 1;
        p.s = st;
        p.dg_in = a.dg + (size_t)(par ^ 1) * 2 * B * 4 * H; p.dg_out = a.dg + (size_t)par * 2 * B * 4 * H;
        p.dpass_in = a.dpass + (size_t)(par ^ 1) * 2 * B * H; p.dpass_out = a.dpass + (size_t)par * 2 * B * H;
        if (st < L - 1) {   // everybody has published step st + 1 = (L - 1 - st) steps
            if (tid < 64) {
                const unsigned want = (unsigned)(L - 1 - st);
                unsigned spins = 0;
                while (true) {
                    bool ok = true;
                    for (int w = tid; w < a.nwg; w += 64) ok = ok && __hip_atomic_load(fl + (size_t)w * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
                    if (__all(ok)) break;
                    if ((++spins & 127u) == 1u) {   // (after a time-out every wait gives up at its first look)
                        if (__hip_atomic_load(a.tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                        if (spins > limit) { if (tid == 0) __hip_atomic_store(a.tmo, 0x600u + (unsigned)dir, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
        }
        enc_bptt_step<true>(p, sm, dir, j0, tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores have left
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.flags + ((size_t)dir * a.nwg + blockIdx.x) * 32, (unsigned)(L - st), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// dst[d][c][r] = src[d][r][c]
__global__ void transpose_batched_kernel(const float* src, float* dst, int n, int rows, int cols) {
    const long per = (long)rows * cols, total = per * n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long d = i / per, rc = i - d * per;
        const int c = (int)(rc / rows), r = (int)(rc - (long)c * rows);
        dst[i] = src[d * per + (long)r * cols + c];
    }
}

struct EncBpttPlan { size_t wt, dg, dpass, dc, sync, total; };
constexpr int EB_SYNC_STATUS = 0, EB_SYNC_TMO = 32, EB_SYNC_FLAGS = 64;   // words inside the sync region (a 128-byte line each)
EncBpttPlan enc_bptt_plan(int B, int H) {
    EncBpttPlan p{};
    size_t o = 0;
    auto take = [&](size_t floats) { size_t r = o; o += (floats + 63) / 64 * 64; return r; };
    p.wt = take((size_t)2 * H * 4 * H);
    p.dg = take((size_t)2 * 2 * B * 4 * H);      // two parities
    p.dpass = take((size_t)2 * 2 * B * H);
    p.dc = take((size_t)2 * B * H);
    p.sync = take((size_t)EB_SYNC_FLAGS + (size_t)2 * ((H + EB_UJ - 1) / EB_UJ) * 32);   // status word, time-out word, a flag line per workgroup
    p.total = o;
    return p;
}

}  // namespace
}  // namespace gvx

extern "C" {

size_t gvx_train_encoder_lstm_bptt_workspace_bytes(int B, int H) {
    if (B < 1 || H < 1) return 0;
    return enc_bptt_plan(B, H).total * sizeof(float);
}

static int encoder_lstm_bptt_impl(const float* xg, const float* memory, const float* cell_states, const float* dmemory, const float* w_hh,
                                  const int32_t* lengths, int B, int L, int H, float* dg_pos, float* hprev_pos, void* workspace,
                                  size_t workspace_bytes, void* stream, bool resident) {
    if (!xg || !memory || !cell_states || !dmemory || !w_hh || !lengths || !dg_pos || !hprev_pos || !workspace)
        return tfail(GVX_ERR_INVALID_ARG, "encoder_lstm_bptt: null argument");
    if (B < 1 || L < 1 || H < 8 || (H % 8)) return tfail(GVX_ERR_UNSUPPORTED, "encoder_lstm_bptt: B, L >= 1, H a positive multiple of 8");
    const EncBpttPlan pl = enc_bptt_plan(B, H);
    if (workspace_bytes < pl.total * sizeof(float)) return tfail(GVX_ERR_WORKSPACE, "encoder_lstm_bptt: workspace too small");
    const size_t lds = (size_t)(EB_UJ * 4 * H * 2) * sizeof(float);
    if (lds > 160 * 1024) return tfail(GVX_ERR_UNSUPPORTED, "encoder_lstm_bptt: H too large for the LDS");
    hipStream_t s = (hipStream_t)stream;
    float* ws = reinterpret_cast<float*>(workspace);
    TR_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(encoder_bptt_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(transpose_batched_kernel, dim3(blocks_for((long)2 * 4 * H * H)), dim3(256), 0, s, w_hh, ws + pl.wt, 2, 4 * H, H);
    TR_TRY(hipMemsetAsync(ws + pl.dg, 0, (pl.total - pl.dg) * sizeof(float), s));
    TR_TRY(hipMemsetAsync(dg_pos, 0, (size_t)2 * B * L * 4 * H * sizeof(float), s));
    TR_TRY(hipMemsetAsync(hprev_pos, 0, (size_t)2 * B * L * H * sizeof(float), s));
    // one resident launch for the whole walk where all its workgroups fit on the GPU at once (default layer size: 128 of 256 CUs)
    const int nwg = (H + EB_UJ - 1) / EB_UJ;
    if (resident && 2 * nwg <= 192 && lds <= 64 * 1024) {
        TR_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(encoder_bptt_resident_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        unsigned* sync = reinterpret_cast<unsigned*>(ws + pl.sync);
        EncBpttRes a{};
        a.q.B = B; a.q.L = L; a.q.H = H;
        a.q.xg = xg; a.q.memory = memory; a.q.c_enc = cell_states; a.q.dmemory = dmemory; a.q.w_hh = w_hh; a.q.w_hh_t = ws + pl.wt; a.q.lengths = lengths;
        a.q.dc = ws + pl.dc; a.q.dg_pos = dg_pos; a.q.hprev_pos = hprev_pos;
        a.dg = ws + pl.dg; a.dpass = ws + pl.dpass; a.flags = sync + EB_SYNC_FLAGS; a.tmo = sync + EB_SYNC_TMO; a.nwg = nwg;
        { const char* e = std::getenv("GVX_HANDOFF_SPIN_LIMIT"); a.spin_limit = e ? (unsigned)std::strtoul(e, nullptr, 10) : 0u; }
        { const char* e = std::getenv("GVX_DEBUG_ENC_BPTT_SKIP_BLOCK"); a.debug_skip_block = e ? std::atoi(e) : -1; }   // (tests: forced time-out)
        hipLaunchKernelGGL(encoder_bptt_resident_kernel, dim3(nwg, 2), dim3(256), lds, s, a);
        // a hand-off that timed out must not look like a result: NaN over both outputs, the code into the workspace's status word
        float* outs[2] = {dg_pos, hprev_pos};
        const size_t counts[2] = {(size_t)2 * B * L * 4 * H, (size_t)2 * B * L * H};
        TR_TRY(launch_poison_on_timeout(sync + EB_SYNC_TMO, reinterpret_cast<int*>(sync + EB_SYNC_STATUS), outs, counts, 2, s));
        TR_TRY(hipGetLastError());
        return GVX_OK;
    }
    for (int st = L - 1; st >= 0; --st) {
        EncBptt q{};
        q.B = B; q.L = L; q.H = H; q.s = st;
        q.xg = xg; q.memory = memory; q.c_enc = cell_states; q.dmemory = dmemory; q.w_hh = w_hh; q.w_hh_t = ws + pl.wt; q.lengths = lengths;
        const int par = st & 1;
        q.dg_in = ws + pl.dg + (size_t)(par ^ 1) * 2 * B * 4 * H; q.dg_out = ws + pl.dg + (size_t)par * 2 * B * 4 * H;
        q.dpass_in = ws + pl.dpass + (size_t)(par ^ 1) * 2 * B * H; q.dpass_out = ws + pl.dpass + (size_t)par * 2 * B * H;
        q.dc = ws + pl.dc; q.dg_pos = dg_pos; q.hprev_pos = hprev_pos;
        q.stamp = st == L / 2;
        hipLaunchKernelGGL(encoder_bptt_step_kernel, dim3((H + EB_UJ - 1) / EB_UJ, 2), dim3(256), lds, s, q);
    }
    TR_TRY(hipGetLastError());
    return GVX_OK;
}

int gvx_train_encoder_lstm_bptt(const float* xg, const float* memory, const float* cell_states, const float* dmemory, const float* w_hh,
                                const int32_t* lengths, int B, int L, int H, float* dg_pos, float* hprev_pos, void* workspace,
                                size_t workspace_bytes, void* stream) {
    return encoder_lstm_bptt_impl(xg, memory, cell_states, dmemory, w_hh, lengths, B, L, H, dg_pos, hprev_pos, workspace, workspace_bytes, stream, false);
}
int gvx_train_encoder_lstm_bptt_resident(const float* xg, const float* memory, const float* cell_states, const float* dmemory, const float* w_hh,
                                         const int32_t* lengths, int B, int L, int H, float* dg_pos, float* hprev_pos, void* workspace,
                                         size_t workspace_bytes, void* stream) {
    return encoder_lstm_bptt_impl(xg, memory, cell_states, dmemory, w_hh, lengths, B, L, H, dg_pos, hprev_pos, workspace, workspace_bytes, stream, true);
}
int gvx_train_encoder_lstm_bptt_status(const void* workspace, size_t workspace_bytes, int B, int H, int* code_out, void* stream) {
    if (!workspace || !code_out || B < 1 || H < 8) return tfail(GVX_ERR_INVALID_ARG, "encoder_lstm_bptt_status: bad argument");
    const EncBpttPlan pl = enc_bptt_plan(B, H);
    if (workspace_bytes < pl.total * sizeof(float)) return tfail(GVX_ERR_WORKSPACE, "encoder_lstm_bptt_status: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    TR_TRY(hipMemcpyAsync(code_out, reinterpret_cast<const float*>(workspace) + pl.sync + EB_SYNC_STATUS, sizeof(int), hipMemcpyDeviceToHost, s));
    TR_TRY(hipStreamSynchronize(s));
    return GVX_OK;
}

}  // extern "C"

#ifdef GVX_STAMPS
// diagnostic build only: phase timestamps of the flagged BPTT launches (tools/stamps_train.py)
extern "C" int gvx_debug_read_stamps_train(unsigned long long* host96) {
    return hipMemcpyFromSymbol(host96, HIP_SYMBOL(gvx::gvx_stamps), sizeof(unsigned long long) * 96) == hipSuccess ? 0 : 1;
}
#endif
