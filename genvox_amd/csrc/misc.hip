// Data-movement kernels around the GEMM / recurrent kernels: embedding gather, layout changes between
// the reference's channels-first API tensors ([B, C, T]) and the channels-last halo-padded layout the
// implicit-GEMM convolutions use, output padding mask, Prenet keep-mask generator and the
// autoregressive stop bookkeeping.  All of them are pure HBM-bound byte movers: coalesced 128-byte
// rows in, LDS tile transpose, coalesced rows out.
#include "gvx_kernels.h"

namespace gvx {

// ---- embedding gather (reference: nn.Embedding, models/tts/tacotron2.py:459/:486) -------------------
// Blocks past the B*L token rows clear the halo rows (the convolutions' zero padding) of x and of the stack's second buffer x2:
// nobody else ever writes them, and clearing the two whole buffers cost two 8 MB fills per call.
__global__ void embed_kernel(const int64_t* tokens, const float* emb, int n_tokens, float* x, float* x2, int B, int L, int E, int halo, int* err_flag) {
    const int row = blockIdx.x;  // b*L + l
    if (row >= B * L) {
        const int k = row - B * L, b = k / (2 * halo), r = k - b * 2 * halo;
        const long at = ((long)b * (L + 2 * halo) + (r < halo ? r : L + r)) * E;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = threadIdx.x; i < E / 4; i += blockDim.x) {
            reinterpret_cast<float4*>(x + at)[i] = z;
            if (x2) reinterpret_cast<float4*>(x2 + at)[i] = z;
        }
        return;
    }
    const int b = row / L, l = row - b * L;
    long tok = tokens[row];
    bool ok = tok >= 0 && tok < n_tokens;
    if (!ok && threadIdx.x == 0) atomicExch(err_flag, 1);
    const float4* src = reinterpret_cast<const float4*>(emb + (ok ? tok : 0) * E);
    float4* dst = reinterpret_cast<float4*>(x + ((long)b * (L + 2 * halo) + halo + l) * E);
    for (int i = threadIdx.x; i < E / 4; i += blockDim.x) dst[i] = ok ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
}

hipError_t launch_embed(const int64_t* tokens, const float* emb, int n_tokens, float* x, float* x2, int B, int L, int E, int halo,
                        int* err_flag, hipStream_t s) {
    hipLaunchKernelGGL(embed_kernel, dim3(B * L + B * 2 * halo), dim3(128), 0, s, tokens, emb, n_tokens, x, x2, B, L, E, halo, err_flag);
    return hipGetLastError();
}

// ---- batched 2-D transpose with strides:  dst[b][j][i] = src[b][i][j] (+ add[b][j][i]) -------------
// src element (b,i,j) at src + b*sb + i*si + j ; dst element at dst + b*db + j*dj + i.
// Column j == extra_col (if extra != nullptr) is diverted to extra[b*extra_bs + i] instead.
struct Tr3 {
    const float* src; long sb, si; int ni, nj;
    float* dst; long db, dj;
    const float* add;
    float* extra; long extra_bs; int extra_col;
    int nj_main;  // columns [0, nj_main) go to dst
    // optional per-sequence lengths along the TIME axis (len_axis: 0 = i is time, 1 = j is time): elements at or past
    // lens[b] are written as zeros; halo > 0 (len_axis 1 only): dst rows j in [-halo, 0) and [nj, nj + halo) are zeroed
    const int32_t* lens; int len_axis; int halo;
};

__global__ void transpose3_kernel(Tr3 p) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, j = j0 + tx;
        tile[r][tx] = (i < p.ni && j < p.nj) ? p.src[(long)b * p.sb + (long)i * p.si + j] : 0.f;
    }
    __syncthreads();
    const int len = p.lens ? p.lens[b] : 0x7fffffff;
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + r, i = i0 + tx;
        if (i >= p.ni || j >= p.nj) continue;
        const float v = tile[tx][r];
        if (j < p.nj_main) {
            const long o = (long)b * p.db + (long)j * p.dj + i;
            const bool live = (p.len_axis ? j : i) < len;
            p.dst[o] = live ? (p.add ? v + p.add[o] : v) : 0.f;
        } else if (p.extra && j == p.extra_col) {
            p.extra[(long)b * p.extra_bs + i] = v;
        }
    }
    if (p.halo > 0 && blockIdx.x == 0 && blockIdx.y == 0) {   // one block per sequence also clears the halo rows
        const int tid = ty * 32 + tx, n = p.halo * p.ni;
        float* front = p.dst + (long)b * p.db - (long)p.halo * p.dj;
        float* back = p.dst + (long)b * p.db + (long)p.nj * p.dj;
        for (int e = tid; e < n; e += 256) {
            const int hr = e / p.ni, c = e - hr * p.ni;
            front[(long)hr * p.dj + c] = 0.f;
            back[(long)hr * p.dj + c] = 0.f;
        }
    }
}

static hipError_t launch_tr3(const Tr3& p, int B, hipStream_t s) {
    dim3 grid((p.nj + 31) / 32, (p.ni + 31) / 32, B);
    hipLaunchKernelGGL(transpose3_kernel, grid, dim3(32, 8), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_frames_from_mel(const float* mel_in, float* frames, int B, int M, int T, hipStream_t s) {
    // frames[(t+1)*B + b][m] = mel_in[b][m][t]; the go-frame rows (t = 0) are zeroed by the caller
    Tr3 p{};
    p.src = mel_in; p.sb = (long)M * T; p.si = T; p.ni = M; p.nj = T;
    p.dst = frames + (long)B * M; p.db = M; p.dj = (long)B * M;
    p.nj_main = T;
    return launch_tr3(p, B, s);
}

hipError_t launch_split_projection(const float* proj, float* mel_out, float* gate_out, int B, int M, int T, hipStream_t s) {
    // proj [B][T][PS] (PS = padded M+1) -> mel_out [B][M][T], gate_out [B][T]
    const int PS = (M + 1 + 3) & ~3;
    Tr3 p{};
    p.src = proj; p.sb = (long)T * PS; p.si = PS; p.ni = T; p.nj = M + 1;
    p.dst = mel_out; p.db = (long)M * T; p.dj = T;
    p.nj_main = M; p.extra = gate_out; p.extra_bs = T; p.extra_col = M;
    return launch_tr3(p, B, s);
}

hipError_t launch_to_channels_last(const float* src, float* dst, int B, int M, int T, int halo, const int32_t* lens, hipStream_t s) {
    Tr3 p{};
    p.src = src; p.sb = (long)M * T; p.si = T; p.ni = M; p.nj = T;
    p.dst = dst + (long)halo * M; p.db = (long)(T + 2 * halo) * M; p.dj = M;
    p.nj_main = T;
    p.lens = lens; p.len_axis = 1; p.halo = halo;
    return launch_tr3(p, B, s);
}

hipError_t launch_residual_to_channels_first(const float* mel, const float* y, float* mel_post, int B, int M, int T,
                                             const int32_t* lens, hipStream_t s) {
    Tr3 p{};
    p.src = y; p.sb = (long)T * M; p.si = M; p.ni = T; p.nj = M;
    p.dst = mel_post; p.db = (long)M * T; p.dj = T; p.add = mel;
    p.nj_main = M;
    p.lens = lens; p.len_axis = 0;
    return launch_tr3(p, B, s);
}

// ---- [T][B][n] -> [B][T][n] row permutation (alignments out of the time-major step-loop buffer) ---------------
__global__ void permute01_kernel(const float* src, float* dst, int T, int B, int n) {
    const long rows = (long)T * B;
    for (long r = blockIdx.x; r < rows; r += gridDim.x) {
        const int t = (int)(r / B), b = (int)(r - (long)t * B);
        const float* sp = src + r * n;
        float* dp = dst + ((long)b * T + t) * n;
        for (int i = threadIdx.x; i < n; i += blockDim.x) dp[i] = sp[i];
    }
}

hipError_t launch_permute01(const float* src, float* dst, int T, int B, int n, hipStream_t s) {
    const long rows = (long)T * B;
    const int grid = (int)(rows < 4096 ? rows : 4096);
    hipLaunchKernelGGL(permute01_kernel, dim3(grid), dim3(128), 0, s, src, dst, T, B, n);
    return hipGetLastError();
}

// ---- zero halo rows of a channels-last buffer ---------------------------------------------------------
__global__ void zero_halo_kernel(float* buf, int T, int halo, int C) {
    const int b = blockIdx.x;
    float* base = buf + (long)b * (T + 2 * halo) * C;
    const int n = halo * C;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        base[i] = 0.f;
        base[(long)(halo + T) * C + i] = 0.f;
    }
}

hipError_t launch_zero_halo(float* buf, int B, int T, int halo, int C, hipStream_t s) {
    if (halo <= 0) return hipSuccess;
    hipLaunchKernelGGL(zero_halo_kernel, dim3(B), dim3(256), 0, s, buf, T, halo, C);
    return hipGetLastError();
}

// ---- output padding mask (models/tts/tacotron2.py:466-473) -----------------------------------------
__global__ void mask_padding_kernel(float* mel, float* mel_post, float* gate, const int32_t* mel_lengths, int M, int T) {
    const int b = blockIdx.y;
    const int len = mel_lengths[b];
    const long n = (long)M * T;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)(i % T);
        if (t >= len) {
            if (mel) mel[(long)b * n + i] = 0.f;
            if (mel_post) mel_post[(long)b * n + i] = 0.f;
            if (gate && i < T) gate[(long)b * T + t] = 1e3f;
        }
    }
}

hipError_t launch_mask_padding(float* mel, float* mel_post, float* gate, const int32_t* mel_lengths, int B, int M, int T,
                               hipStream_t s) {
    const long n = (long)M * T;
    int gx = (int)((n + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(mask_padding_kernel, dim3(gx, B), dim3(256), 0, s, mel, mel_post, gate, mel_lengths, M, T);
    return hipGetLastError();
}

// ---- several small buffers cleared by ONE launch (every hipMemsetAsync is a launch of its own: ~5 us on the stream) ----
struct ZeroArgs { uint32_t* p[8]; size_t n[8]; };   // n: 32-bit words
__global__ void zero_many_kernel(ZeroArgs a) {
    uint32_t* p = a.p[blockIdx.y];
    const size_t n = a.n[blockIdx.y];
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) p[k] = 0u;
}
hipError_t launch_zero_many(void* const* ptrs, const size_t* bytes, int n_arrays, hipStream_t s) {
    if (n_arrays < 1 || n_arrays > 8) return hipErrorInvalidValue;
    ZeroArgs a{};
    size_t most = 0;
    for (int i = 0; i < n_arrays; ++i) {
        if ((reinterpret_cast<uintptr_t>(ptrs[i]) & 3) || (bytes[i] & 3)) return hipErrorInvalidValue;
        a.p[i] = reinterpret_cast<uint32_t*>(ptrs[i]); a.n[i] = bytes[i] / 4;
        if (a.n[i] > most) most = a.n[i];
    }
    int gx = (int)((most + 1023) / 1024);   // (four words per thread at the widest array, at most 64 workgroups each)
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(zero_many_kernel, dim3(gx, n_arrays), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---- hand-off time-out made loud: NaN over every output of the call + a sticky status word -------------
// Last launch of a teacher-forced call: when the call's hand-off time-out word (zeroed at the start of the call, raised by
// a bounded in-launch wait that gave up) is set, the loop drained with wrong results - they must not look like results.
struct PoisonArgs { float* p[4]; size_t n[4]; };
__global__ void poison_on_timeout_kernel(const unsigned* tmo, int* sticky, PoisonArgs a) {
    const unsigned code = __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (code == 0u) return;   // the normal case: one word read per workgroup
    if (blockIdx.x == 0 && threadIdx.x == 0) *sticky = (int)code;
    const float nan = __int_as_float(0x7fc00000);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < a.n[i]; k += (size_t)gridDim.x * blockDim.x) a.p[i][k] = nan;
}
hipError_t launch_poison_on_timeout(const unsigned* tmo, int* sticky, float* const* ptrs, const size_t* counts, int n_arrays, hipStream_t s) {
    PoisonArgs a{};
    for (int i = 0; i < 4; ++i) { a.p[i] = i < n_arrays ? ptrs[i] : nullptr; a.n[i] = i < n_arrays && ptrs[i] ? counts[i] : 0; }
    hipLaunchKernelGGL(poison_on_timeout_kernel, dim3(64), dim3(256), 0, s, tmo, sticky, a);
    return hipGetLastError();
}

// ---- Prenet keep masks: Bernoulli(0.5) bytes from a splitmix64 counter hash --------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void mask_gen_kernel(uint8_t* out, size_t n, uint64_t seed) {
    // one 64-bit hash gives 64 keep bits -> 64 output bytes (8 per 8-byte store)
    const size_t words = (n + 7) / 8;
    for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (size_t)gridDim.x * blockDim.x) {
        const uint64_t bits = splitmix64(splitmix64(seed) ^ (w >> 3));
        const unsigned byte = (unsigned)(bits >> (8 * (w & 7))) & 0xFFu;
        uint64_t v = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) v |= (uint64_t)((byte >> i) & 1u) << (8 * i);
        if (8 * w + 8 <= n) {
            reinterpret_cast<uint64_t*>(out)[w] = v;
        } else {
            for (size_t i = 8 * w; i < n; ++i) out[i] = (uint8_t)((v >> (8 * (i - 8 * w))) & 1u);
        }
    }
}

hipError_t launch_mask_gen(uint8_t* out, size_t n, uint64_t seed, hipStream_t s) {
    if (n == 0) return hipSuccess;
    if (reinterpret_cast<uintptr_t>(out) & 7) return hipErrorInvalidValue;
    const size_t words = (n + 7) / 8;
    int grid = (int)((words + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(mask_gen_kernel, dim3(grid), dim3(256), 0, s, out, n, seed);
    return hipGetLastError();
}

// ---- autoregressive bookkeeping (models/tts/tacotron2.py:405-409, per row) ----------------------------
// proj_t is a blocked vector [ceil((M+1)/8)][B][8]: element (b, n) at (n>>3)*B*8 + b*8 + (n&7)
__global__ void ar_stop_kernel(const float* proj_t, int gate_col, float threshold, int t, int B, int32_t* n_frames, int32_t* n_done) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (n_frames[b] != 0) return;  // already finished
    const float g = proj_t[(long)(gate_col >> 3) * B * 8 + b * 8 + (gate_col & 7)];
    const float sg = 1.f / (1.f + expf(-g));
    if (sg > threshold) {
        n_frames[b] = t + 1;
        atomicAdd(n_done, 1);
    }
}

hipError_t launch_ar_stop(const float* proj_t, int gate_col, float threshold, int t, int B, int32_t* n_frames, int32_t* n_done,
                          hipStream_t s) {
    hipLaunchKernelGGL(ar_stop_kernel, dim3((B + 63) / 64), dim3(64), 0, s, proj_t, gate_col, threshold, t, B, n_frames, n_done);
    return hipGetLastError();
}

// Autoregressive step tail, grid (S, B), 256 threads: workgroup (s, b) handles batch row b and the s-th slice of the Prenet
// output columns.
//   phase 1 (every workgroup of the row, redundantly - 45 KB of partials): proj[b][n] = sum of the partial slabs the
//            decoder-LSTM tiles emitted (two halves of the slab range summed by two thread groups, lower half added first)
//            + p_ctx; slice 0 stores it and runs the per-row stop test (models/tts/tacotron2.py:405-409)
//   phase 2 (redundantly): Prenet layer 1 of the NEXT step on the fresh mel frame (tacotron2.py:398, :140-144):
//            h1[j] = 2 * keep0 * relu(sum_n W0[j][n] * mel[n])
//   phase 3: this workgroup's columns of Prenet layer 2: out[o] = 2 * keep1 * relu(sum_k W1[o][k] * h1[k]), K split over
//            four thread groups and added in a fixed order; written as the blocked vector the attention-LSTM tiles read.
// Both Prenet matrices are read transposed ([k][out]) so that a wave's loads are contiguous.
constexpr int ARP_KQ = 4;
__global__ __launch_bounds__(256) void ar_project_kernel(const float* __restrict__ p_slab, int n_slabs, const float* __restrict__ p_ctx,
                                                         float* __restrict__ proj_t, int M, int PSB, float threshold, int t, int B,
                                                         int32_t* n_frames, int32_t* n_done, const float* __restrict__ w0t,
                                                         const float* __restrict__ w1t, int P, const uint8_t* __restrict__ keep0,
                                                         const uint8_t* __restrict__ keep1, float* __restrict__ prenet_out) {
    extern __shared__ float arp_smem[];
    __shared__ float part[2][128];
    __shared__ float mel[128];
    float* h1 = arp_smem;                 // [P]
    float* red = arp_smem + P;            // [ARP_KQ][OS]
    const int sl = blockIdx.x, S = gridDim.x, b = blockIdx.y, tid = threadIdx.x;
    const int half = tid / PSB, n = tid - half * PSB;   // PSB <= 128
    if (half < 2 && n <= M) {
        const int s_begin = half == 0 ? 0 : n_slabs / 2, s_end = half == 0 ? n_slabs / 2 : n_slabs;
        const float* sp = p_slab + (long)b * PSB + n;
        const long sstride = (long)B * PSB;
        float acc = 0.f;
        int q0 = s_begin;
        for (; q0 + 16 <= s_end; q0 += 16) {   // 16 loads in flight, added in ascending slab order
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = sp[(long)(q0 + q) * sstride];
#pragma unroll
            for (int q = 0; q < 16; ++q) acc += v[q];
        }
        for (; q0 < s_end; ++q0) acc += sp[(long)q0 * sstride];
        part[half][n] = acc;
    }
    __syncthreads();
    if (half == 0 && n <= M) {
        const long blocked = (long)(n >> 3) * B * 8 + b * 8 + (n & 7);
        const float v = (part[0][n] + part[1][n]) + p_ctx[blocked];
        if (n < M) mel[n] = v;
        if (sl == 0) {
            proj_t[blocked] = v;
            if (n == M && n_frames[b] == 0) {
                const float sg = 1.f / (1.f + expf(-v));
                if (sg > threshold) {
                    n_frames[b] = t + 1;
                    atomicAdd(n_done, 1);
                }
            }
        }
    }
    if (!keep0) return;   // last step: nobody consumes a next Prenet input (uniform)
    __syncthreads();
    for (int j = tid; j < P; j += 256) {
        float acc = 0.f;
        for (int k = 0; k < M; ++k) acc = fmaf(w0t[(long)k * P + j], mel[k], acc);
        acc = fmaxf(acc, 0.f);
        h1[j] = keep0[(long)b * P + j] ? 2.f * acc : 0.f;
    }
    __syncthreads();
    const int OS = (P + S - 1) / S, o_begin = sl * OS, o_cnt = min(OS, P - o_begin);
    if (o_cnt <= 0) return;
    const int KC = (P + ARP_KQ - 1) / ARP_KQ;
    for (int idx = tid; idx < o_cnt * ARP_KQ; idx += 256) {
        const int kq = idx / o_cnt, ol = idx - kq * o_cnt;
        const int k_end = min(P, (kq + 1) * KC);
        const float* wcol = w1t + o_begin + ol;
        float acc = 0.f;
        for (int k = kq * KC; k < k_end; ++k) acc = fmaf(wcol[(long)k * P], h1[k], acc);
        red[kq * OS + ol] = acc;
    }
    __syncthreads();
    for (int ol = tid; ol < o_cnt; ol += 256) {
        float acc = red[ol];
#pragma unroll
        for (int kq = 1; kq < ARP_KQ; ++kq) acc += red[kq * OS + ol];
        const int o = o_begin + ol;
        acc = fmaxf(acc, 0.f);
        acc = keep1[(long)b * P + o] ? 2.f * acc : 0.f;
        prenet_out[(long)(o >> 3) * B * 8 + b * 8 + (o & 7)] = acc;
    }
}

// The same step tail for the common sizes (P <= 256, <= 128 slabs), written for latency: the kernel sits on the
// autoregressive critical chain between the decoder LSTM and the next attention LSTM, and small dependent kernels like this
// one are bound by instruction issue and by the number of memory round trips, not by bytes.  1024 threads; every global
// access is a 16-byte load; the only loads that depend on the previous launch (the partial slabs) are issued first and
// behind them - before anything waits - the Prenet weights each thread needs, so one round trip covers everything.
//   slabs    thread (sg, n4): slabs 4 sg .. 4 sg + 3 of float4 column n4            -> sp4[32][PSB/4] -> 88 threads add 32
//   layer 1  thread (kq, j4): k in [KPT kq, KPT kq + KPT) of output float4 j4       -> l1p[16][P/4]   -> P threads add 16
//   layer 2  thread (kq, o4): k in [4 kq, 4 kq + 4) of this slice's output float4   -> l2p[64][16] -> l2q[16][16] -> add 16
// (fixed summation orders everywhere: bitwise reproducible).
constexpr int ARP_THREADS = 1024;
template <int KPT>   // layer-1 k values per thread: n_mels <= 16 * KPT
__global__ __launch_bounds__(ARP_THREADS) void ar_project_fast_kernel(const float* __restrict__ p_slab, int n_slabs,
                                                                      const float* __restrict__ p_ctx, float* __restrict__ proj_t, int M,
                                                                      int PSB, float threshold, int t, int B, int32_t* n_frames,
                                                                      int32_t* n_done, const float* __restrict__ w0t,
                                                                      const float* __restrict__ w1t, int P,
                                                                      const uint8_t* __restrict__ keep0, const uint8_t* __restrict__ keep1,
                                                                      float* __restrict__ prenet_out) {
    __shared__ float4 sp4[32][32];
    __shared__ float4 l1p[16][64];
    __shared__ float4 l2p[64][16];
    __shared__ float4 l2q[16][16];
    __shared__ __attribute__((aligned(16))) float mel[128];
    __shared__ __attribute__((aligned(16))) float h1[256];
    const int sl = blockIdx.x, S = gridDim.x, b = blockIdx.y;
    const unsigned tid = threadIdx.x;
    const bool more = keep0 != nullptr;   // uniform: false for the last step (nobody consumes a next Prenet input)
    // ---- dependent loads first: partial slabs
    const unsigned n4c = (unsigned)PSB >> 2;                  // float4 per slab row (<= 32)
    const unsigned sg = tid / n4c, n4 = tid - sg * n4c;
    const bool slab_thread = sg < 32;
    float4 sv[4];
    {
        const float4* sp = reinterpret_cast<const float4*>(p_slab) + (unsigned)b * n4c + n4;
        const unsigned sstride = (unsigned)B * n4c;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned sidx = min(sg * 4u + q, (unsigned)n_slabs - 1u);
            sv[q] = slab_thread ? sp[sidx * sstride] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (sg * 4u + q >= (unsigned)n_slabs) sv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const unsigned nn = min(tid, (unsigned)PSB - 1u);
    const long blocked = (long)(nn >> 3) * B * 8 + b * 8 + (nn & 7);
    const float pc = p_ctx[blocked];
    // ---- Prenet weights into registers (independent of the previous launch; L2-resident, shared by all rows)
    const unsigned j4c = (unsigned)P >> 2;                    // float4 per Prenet row (<= 64)
    const unsigned kq1 = tid / j4c, j4 = tid - kq1 * j4c;    // layer 1: 16 k slices
    const unsigned OS = (unsigned)P / S, o4c = OS >> 2, o_begin = sl * OS;   // host guarantees P % (4 S) == 0
    const unsigned kq2 = tid >> 4, o4 = tid & 15;             // layer 2: 64 k slices of 4
    float4 w0v[KPT], w1v[4];
    unsigned char k0 = 0, k1 = 0;
    if (more) {
        const float4* w0 = reinterpret_cast<const float4*>(w0t) + j4;
#pragma unroll
        for (int i = 0; i < KPT; ++i) w0v[i] = w0[min(min(kq1, 15u) * KPT + i, (unsigned)M - 1u) * j4c];
        const float4* w1 = reinterpret_cast<const float4*>(w1t) + (o_begin >> 2) + min(o4, o4c - 1u);
#pragma unroll
        for (int i = 0; i < 4; ++i) w1v[i] = w1[min(kq2 * 4u + i, (unsigned)P - 1u) * j4c];
        if (tid < (unsigned)P) k0 = keep0[(long)b * P + tid];
        if (tid < OS) k1 = keep1[(long)b * P + o_begin + tid];
    }
    // ---- phase 1: slab reduction
    if (slab_thread) {
        float4 acc = sv[0];
        acc.x += sv[1].x; acc.y += sv[1].y; acc.z += sv[1].z; acc.w += sv[1].w;
        acc.x += sv[2].x; acc.y += sv[2].y; acc.z += sv[2].z; acc.w += sv[2].w;
        acc.x += sv[3].x; acc.y += sv[3].y; acc.z += sv[3].z; acc.w += sv[3].w;
        sp4[sg][n4] = acc;
    }
    __syncthreads();
    if (tid < 128) {
        float v = 0.f;
        if (tid <= (unsigned)M) {
            const float* col = reinterpret_cast<const float*>(sp4) + tid;
            float acc = col[0];
#pragma unroll
            for (int q = 1; q < 32; ++q) acc += col[q * 128];
            v = acc + pc;
            if (sl == 0) {
                proj_t[blocked] = v;
                if (tid == (unsigned)M && n_frames[b] == 0) {
                    const float sgm = 1.f / (1.f + expf(-v));
                    if (sgm > threshold) {
                        n_frames[b] = t + 1;
                        atomicAdd(n_done, 1);
                    }
                }
            }
        }
        mel[tid] = tid < (unsigned)M ? v : 0.f;   // zeros past the mel bins: clamped weight loads contribute nothing
    }
    if (!more) return;
    __syncthreads();
    // ---- phase 2: Prenet layer 1
    if (kq1 < 16 && j4 < j4c) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const float x = mel[kq1 * KPT + i];
            acc.x = fmaf(w0v[i].x, x, acc.x); acc.y = fmaf(w0v[i].y, x, acc.y);
            acc.z = fmaf(w0v[i].z, x, acc.z); acc.w = fmaf(w0v[i].w, x, acc.w);
        }
        l1p[kq1][j4] = acc;
    }
    __syncthreads();
    if (tid < 256) {
        float r = 0.f;
        if (tid < (unsigned)P) {
            const float* col = reinterpret_cast<const float*>(l1p) + tid;
            float acc = col[0];
#pragma unroll
            for (int q = 1; q < 16; ++q) acc += col[q * 256];
            acc = fmaxf(acc, 0.f);
            r = k0 ? 2.f * acc : 0.f;
        }
        h1[tid] = r;   // zeros past P
    }
    __syncthreads();
    // ---- phase 3: this workgroup's layer-2 columns
    {
        const float4 h4 = reinterpret_cast<const float4*>(h1)[kq2];
        float4 acc;
        acc.x = w1v[0].x * h4.x; acc.y = w1v[0].y * h4.x; acc.z = w1v[0].z * h4.x; acc.w = w1v[0].w * h4.x;
        acc.x = fmaf(w1v[1].x, h4.y, acc.x); acc.y = fmaf(w1v[1].y, h4.y, acc.y); acc.z = fmaf(w1v[1].z, h4.y, acc.z); acc.w = fmaf(w1v[1].w, h4.y, acc.w);
        acc.x = fmaf(w1v[2].x, h4.z, acc.x); acc.y = fmaf(w1v[2].y, h4.z, acc.y); acc.z = fmaf(w1v[2].z, h4.z, acc.z); acc.w = fmaf(w1v[2].w, h4.z, acc.w);
        acc.x = fmaf(w1v[3].x, h4.w, acc.x); acc.y = fmaf(w1v[3].y, h4.w, acc.y); acc.z = fmaf(w1v[3].z, h4.w, acc.z); acc.w = fmaf(w1v[3].w, h4.w, acc.w);
        l2p[kq2][o4] = acc;
    }
    __syncthreads();
    if (tid < 256) {   // (r, o4): add k slices 4 r .. 4 r + 3
        const unsigned r = tid >> 4;
        float4 acc = l2p[4 * r][o4];
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const float4 v = l2p[4 * r + q][o4];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        l2q[r][o4] = acc;
    }
    __syncthreads();
    if (tid < OS) {
        const float* col = reinterpret_cast<const float*>(l2q) + tid;
        float acc = col[0];
#pragma unroll
        for (int q = 1; q < 16; ++q) acc += col[q * 64];
        const unsigned o = o_begin + tid;
        acc = fmaxf(acc, 0.f);
        acc = k1 ? 2.f * acc : 0.f;
        prenet_out[(long)(o >> 3) * B * 8 + b * 8 + (o & 7)] = acc;
    }
}

hipError_t launch_ar_project(const float* p_slab, int n_slabs, const float* p_ctx, float* proj_t, int M, float threshold, int t, int B,
                             int32_t* n_frames, int32_t* n_done, const float* w0t, const float* w1t, int P, const uint8_t* keep0,
                             const uint8_t* keep1, float* prenet_out, hipStream_t s) {
    const int PSB = (M + 1 + 7) & ~7;
    if (PSB > 128 || M > 128) return hipErrorInvalidValue;
    // column slices per row: enough workgroups to spread the 4*P*P bytes of layer 2, whole float4 groups per slice
    int S = 4;
    while (S > 1 && (P % (4 * S) != 0 || P / S < 8 || S * B > 512)) S >>= 1;
    if (P <= 256 && P % (4 * S) == 0 && n_slabs >= 1 && n_slabs <= 128) {
        const dim3 grid(S, B), block(ARP_THREADS);
#define GVX_ARP(K) ar_project_fast_kernel<K><<<grid, block, 0, s>>>(p_slab, n_slabs, p_ctx, proj_t, M, PSB, threshold, t, B, n_frames, \
                                                                    n_done, w0t, w1t, P, keep0, keep1, prenet_out)
        if (M <= 32) GVX_ARP(2);
        else if (M <= 80) GVX_ARP(5);
        else GVX_ARP(8);
#undef GVX_ARP
        return hipGetLastError();
    }
    const int OS = (P + S - 1) / S;
    const size_t lds = (size_t)(P + ARP_KQ * OS) * sizeof(float);
    hipLaunchKernelGGL(ar_project_kernel, dim3(S, B), dim3(256), lds, s, p_slab, n_slabs, p_ctx, proj_t, M, PSB, threshold, t, B,
                       n_frames, n_done, w0t, w1t, P, keep0, keep1, prenet_out);
    return hipGetLastError();
}

__global__ void ar_emit_all_kernel(const float* proj, float* mel_out, float* gate_out, int B, int M, int Tmax, int steps, int PSB,
                                   const int32_t* n_frames) {
    // thread = (b, m); loops over time so that writes along t are contiguous per thread row
    const int b = blockIdx.y;
    const int nf = n_frames ? n_frames[b] : steps;
    for (int m = blockIdx.x * blockDim.y + threadIdx.y; m <= M; m += gridDim.x * blockDim.y) {
        const long src = (long)(m >> 3) * B * 8 + b * 8 + (m & 7);
        for (int t = threadIdx.x; t < steps; t += blockDim.x) {
            const float v = proj[(long)t * B * PSB + src];
            if (m < M) mel_out[((long)b * M + m) * Tmax + t] = t < nf ? v : 0.f;
            else gate_out[(long)b * Tmax + t] = t < nf ? v : 1e3f;
        }
    }
}

hipError_t launch_ar_emit_all(const float* proj, float* mel_out, float* gate_out, int B, int M, int Tmax, int steps,
                              const int32_t* n_frames, hipStream_t s) {
    if (steps <= 0) return hipSuccess;
    const int PSB = (M + 1 + 7) & ~7;
    hipLaunchKernelGGL(ar_emit_all_kernel, dim3((M + 1 + 3) / 4, B), dim3(64, 4), 0, s, proj, mel_out, gate_out, B, M, Tmax, steps, PSB,
                       n_frames);
    return hipGetLastError();
}

__global__ void permute01_partial_kernel(const float* src, float* dst, int steps, int Tdst, int B, int n, const int32_t* n_frames) {
    const long rows = (long)steps * B;
    for (long r = blockIdx.x; r < rows; r += gridDim.x) {
        const int t = (int)(r / B), b = (int)(r - (long)t * B);
        const bool live = !n_frames || t < n_frames[b];
        const float* sp = src + r * n;
        float* dp = dst + ((long)b * Tdst + t) * n;
        for (int i = threadIdx.x; i < n; i += blockDim.x) dp[i] = live ? sp[i] : 0.f;
    }
}

hipError_t launch_permute01_partial(const float* src, float* dst, int steps, int Tdst, int B, int n, const int32_t* n_frames,
                                    hipStream_t s) {
    const long rows = (long)steps * B;
    if (rows <= 0) return hipSuccess;
    const int grid = (int)(rows < 4096 ? rows : 4096);
    hipLaunchKernelGGL(permute01_partial_kernel, dim3(grid), dim3(128), 0, s, src, dst, steps, Tdst, B, n, n_frames);
    return hipGetLastError();
}

// ---- criterion of the evaluation step (Tacotron2Loss, models/tts/tacotron2.py:598-615) ------------------------------
// partial[block] = {sum (mel - target)^2, sum (mel_post - target)^2, sum BCE-with-logits(gate, gate target)} in float64;
// LOSS_BLOCKS fixed blocks with a fixed element assignment, then one thread adds the partials in block order: bitwise
// reproducible, and accurate enough that the means match a float64 evaluation to float32 rounding.
constexpr int LOSS_BLOCKS = 256;

__global__ __launch_bounds__(256) void loss_partial_kernel(const float* __restrict__ mel, const float* __restrict__ mel_post,
                                                           const float* __restrict__ gate, const float* __restrict__ mel_t,
                                                           const float* __restrict__ gate_t, long n_mel, long n_gate, double* partial) {
    __shared__ double red[3][256];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    const long stride = (long)LOSS_BLOCKS * 256, first = (long)blockIdx.x * 256 + threadIdx.x;
    for (long i = first; i < n_mel; i += stride) {
        const float t = mel_t[i], a = mel[i] - t, b = mel_post[i] - t;
        s0 += (double)(a * a);
        s1 += (double)(b * b);
    }
    for (long i = first; i < n_gate; i += stride) {
        const float x = gate[i], y = gate_t[i];
        s2 += (double)(fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x))));   // numerically stable BCE with logits
    }
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int q = 0; q < 3; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x < 3) partial[(long)blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ void loss_final_kernel(const double* partial, long n_mel, long n_gate, float* out3) {
    double s[3] = {0.0, 0.0, 0.0};
    for (int b = 0; b < LOSS_BLOCKS; ++b)
        for (int q = 0; q < 3; ++q) s[q] += partial[b * 3 + q];
    const float mse0 = (float)(s[0] / (double)n_mel), mse1 = (float)(s[1] / (double)n_mel);
    const float mel_loss = mse0 + mse1;                      // float32 sums, like the reference's tensor arithmetic
    const float gate_loss = (float)(s[2] / (double)n_gate);
    out3[0] = mel_loss + gate_loss;
    out3[1] = mel_loss;
    out3[2] = gate_loss;
}

size_t loss_scratch_bytes() { return (size_t)LOSS_BLOCKS * 3 * sizeof(double); }

hipError_t launch_tacotron2_loss(const float* mel, const float* mel_post, const float* gate, const float* mel_t, const float* gate_t,
                                 long n_mel, long n_gate, double* scratch, float* out3, hipStream_t s) {
    loss_partial_kernel<<<dim3(LOSS_BLOCKS), dim3(256), 0, s>>>(mel, mel_post, gate, mel_t, gate_t, n_mel, n_gate, scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    loss_final_kernel<<<dim3(1), dim3(1), 0, s>>>(scratch, n_mel, n_gate, out3);
    return hipGetLastError();
}

}  // namespace gvx
