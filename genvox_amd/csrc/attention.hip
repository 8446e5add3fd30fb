// Location-sensitive attention, one decoder step.  Replaces Attention.forward / get_alignment_energies /
// LocationLayer.forward of the reference (models/tts/tacotron2.py:89-129, :48-53) plus the cumulative-weights
// update (:353).
//
//   q      = sum of the per-tile partial products written by the attention-LSTM kernel (skinny.hip)
//   f[l,c] = conv1d_k31( [w_prev ; w_cum] )                       (2 -> F filters)
//   e[l]   = v . tanh(q + pm[l,:] + Wd f[l,:]),  -inf for l >= len
//   w      = softmax_l(e);  w_cum += w;  ctx = sum_l w[l] * memory[l,:]
//
// A batch row needs ~0.8 MFLOP of fp32 VALU work and reads L*(a+E) floats (320 KiB at L=128) per step.  One CU
// moves only ~30-70 GB/s from L2 / Infinity Cache (and the weight stream of the LSTM kernel evicts these rows from
// the 4-MiB L2 between steps), so a row is split over G workgroups:
//   attn_energy_kernel  (grid G x B): positions chunk g  -> energies[b][l]
//   attn_context_kernel (grid G x B): softmax over the whole row (512 B, recomputed per workgroup), then the
//                                     context columns slice g; slice 0 also emits the alignment row and w_cum.
// The only cross-workgroup dependency (softmax normaliser) is carried by the kernel boundary: no atomics, no
// in-kernel hand-off, bitwise reproducible.
#include "gvx_kernels.h"
#include "attn_step_body.h"

namespace gvx {

constexpr int AT_THREADS = 256;              // context kernel
constexpr int EN_THREADS = 512;              // energy kernel: 8 waves (conv / energy phases are VALU+LDS work that scales with waves)
constexpr int EN_WAVES = EN_THREADS / 64;
constexpr int EN_LC = 32;    // positions per pass of the energy kernel
constexpr int AT_FP = 32;    // max location filters (register/LDS row width)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

__host__ __device__ inline int chunk_len(int L, int G) { return (L + G - 1) / G; }

struct EnergyLds { int q_off, part_off, total; };
__host__ __device__ inline EnergyLds energy_lds_layout(int a) {
    EnergyLds o;
    auto al = [](int x) { return (x + 3) & ~3; };
    int off = 0;
    o.q_off = off; off += al(EN_QG * a);
    o.part_off = off; off += EN_LC * 65;
    o.total = off;
    return o;
}

// energies[b][l] = v . tanh(q + pm[l,:] + loc[l,:])  for the workgroup's chunk of positions.  The location features
// loc were produced by extra workgroups of the preceding LSTM launch (skinny.hip, loc_body); q is the sum of that
// launch's per-tile partial slabs.  What is left here is one round trip of coalesced loads, a+L tanh per position and
// a 64-lane reduction.
template <int DPL>
__global__ __launch_bounds__(EN_THREADS) void attn_energy_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = p.L, a = p.a;
    const int Lg = chunk_len(L, p.G);
    const EnergyLds lo = energy_lds_layout(a);
    float* qs = smem + lo.q_off;
    float* part = smem + lo.part_off;

    const int g = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l_begin = g * Lg, l_end = min(L, l_begin + Lg);
    if (l_begin >= l_end) return;  // uniform per workgroup
    const int len = p.lengths ? p.lengths[b] : L;
    GVX_STAMP(1, 0);

    // ---- issue all loads of the first pass up front (each dependent round trip to fresh data costs ~1 us)
    const int a4 = a >> 2;
    const int qg = min(EN_QG, EN_THREADS / a4);
    const int grp = tid / a4, d4 = tid - grp * a4;
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int QV = 8;
    float4 qld[QV];
    const bool q_fast = grp < qg && p.n_slabs == QV * qg;   // default dims: 128 slabs = 8 x 16
    {
        const float4* base = reinterpret_cast<const float4*>(p.q_slab + (long)b * a) + d4;
        const long tstride = (long)p.B * a4;
        if (q_fast) {
#pragma unroll
            for (int i = 0; i < QV; ++i) qld[i] = base[(long)(grp + i * qg) * tstride];
        } else if (grp < qg) {
            for (int t = grp; t < p.n_slabs; t += qg) {
                const float4 v = base[(long)t * tstride];
                s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
            }
        }
    }
    float vv[DPL];
    float pmv[EN_LC / EN_WAVES][DPL], lcv[EN_LC / EN_WAVES][DPL];
#pragma unroll
    for (int i = 0; i < DPL; ++i) vv[i] = (lane + 64 * i) < a ? p.v[min(lane + 64 * i, a - 1)] : 0.f;
#pragma unroll
    for (int j = 0; j < EN_LC / EN_WAVES; ++j) {
        const long row = ((long)b * L + min(l_begin + wave + j * EN_WAVES, l_end - 1)) * a;
#pragma unroll
        for (int i = 0; i < DPL; ++i) {
            const int d = min(lane + 64 * i, a - 1);
            pmv[j][i] = p.pm[row + d];
            lcv[j][i] = p.loc[row + d];
        }
    }
    GVX_STAMP(1, 1);
    if (q_fast) {
#pragma unroll
        for (int i = 0; i < QV; ++i) { s4.x += qld[i].x; s4.y += qld[i].y; s4.z += qld[i].z; s4.w += qld[i].w; }
    }
    if (grp < qg) reinterpret_cast<float4*>(qs + grp * a)[d4] = s4;
    __syncthreads();
    GVX_STAMP(1, 2);
    float qv[DPL];
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
        const int d = min(lane + 64 * i, a - 1);
        float qsum = 0.f;
        for (int gq = 0; gq < qg; ++gq) qsum += qs[gq * a + d];
        qv[i] = qsum;
    }

    for (int l0 = l_begin; l0 < l_end; l0 += EN_LC) {
        const int lc = min(EN_LC, l_end - l0);
        if (l0 != l_begin) {  // later passes (long chunks)
#pragma unroll
            for (int j = 0; j < EN_LC / EN_WAVES; ++j) {
                const long row = ((long)b * L + min(l0 + wave + j * EN_WAVES, l_end - 1)) * a;
#pragma unroll
                for (int i = 0; i < DPL; ++i) {
                    const int d = min(lane + 64 * i, a - 1);
                    pmv[j][i] = p.pm[row + d];
                    lcv[j][i] = p.loc[row + d];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < EN_LC / EN_WAVES; ++j) {
            const int ll = wave + j * EN_WAVES;
            float pe = 0.f;
#pragma unroll
            for (int i = 0; i < DPL; ++i) pe = fmaf(vv[i], fast_tanh((qv[i] + lcv[j][i]) + pmv[j][i]), pe);
            part[ll * 65 + lane] = pe;
        }
        __syncthreads();
        GVX_STAMP(1, 3);
        {   // thread = (position, sixteenth of the lanes): 4 partials each, then 4 shuffles
            const int ll = tid >> 4, sg = tid & 15;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) s += part[ll * 65 + sg * 4 + i];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            s += __shfl_xor(s, 8);
            const int l = l0 + ll;
            if (sg == 0 && ll < lc) p.energies[(long)b * L + l] = l < len ? s : -INFINITY;
        }
        __syncthreads();
        GVX_STAMP(1, 4);
    }
}

struct ContextLds { int e_off, red_off, total; };
__host__ __device__ inline ContextLds context_lds_layout(int L) {
    ContextLds o;
    int off = 0;
    o.e_off = off; off += (L + 3) & ~3;
    o.red_off = off; off += 8 * 32 * 4;   // [8 position groups][32 float4 columns]
    o.total = off;
    return o;
}

constexpr int CX_EV = 4;    // energies per lane held in registers (rows up to 256 positions; longer rows re-read)
constexpr int CX_MV = 16;   // memory float4 loads in flight per thread

__global__ __launch_bounds__(AT_THREADS) void attn_context_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = p.L, E = p.E, B = p.B;
    const ContextLds lo = context_lds_layout(L);
    float* ws = smem + lo.e_off;
    float4* red = reinterpret_cast<float4*>(smem + lo.red_off);

    const int g = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    GVX_STAMP(2, 0);

    // ---- issue everything first: the row's energies, and the first CX_MV positions of this thread's memory column.
    // The memory loads do not depend on the softmax, so both round trips overlap.
    const float* erow = p.energies + (long)b * L;
    float ev[CX_EV];
#pragma unroll
    for (int i = 0; i < CX_EV; ++i) {
        const int l = lane + 64 * i;
        ev[i] = l < L ? erow[l] : -INFINITY;
    }
    const int e4n = E >> 2;
    const int cols = (e4n + p.G - 1) / p.G;       // float4 columns per workgroup
    const int c_begin = g * cols, c_end = min(e4n, c_begin + cols);
    const int cc = tid & 31, pg = tid >> 5;        // thread = (float4 column, position residue mod 8)
    const float4* mbase = reinterpret_cast<const float4*>(p.memory + (long)b * L * E);
    float4 mv[CX_MV];
    {
        const int e4 = min(c_begin + cc, e4n - 1);
#pragma unroll
        for (int i = 0; i < CX_MV; ++i) mv[i] = mbase[(long)min(pg + 8 * i, L - 1) * e4n + e4];
    }

    // ---- masked softmax over the whole row; every wave computes the normaliser, wave 0 publishes the weights
    {
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < CX_EV; ++i) m = fmaxf(m, ev[i]);
        for (int l = lane + 64 * CX_EV; l < L; l += 64) m = fmaxf(m, erow[l]);
        m = wave_max(m);
        float s = 0.f;
        float ex[CX_EV];
#pragma unroll
        for (int i = 0; i < CX_EV; ++i) { ex[i] = expf(ev[i] - m); s += ex[i]; }
        for (int l = lane + 64 * CX_EV; l < L; l += 64) s += expf(erow[l] - m);
        s = wave_sum(s);
        const float inv = 1.f / s;
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < CX_EV; ++i) {
                const int l = lane + 64 * i;
                if (l < L) {
                    const float w = ex[i] * inv;
                    ws[l] = w;
                    if (g == 0) {
                        p.w_out[(long)b * p.w_out_bs + l] = w;
                        p.w_cum[(long)b * L + l] += w;
                    }
                }
            }
            for (int l = lane + 64 * CX_EV; l < L; l += 64) {
                const float w = expf(erow[l] - m) * inv;
                ws[l] = w;
                if (g == 0) {
                    p.w_out[(long)b * p.w_out_bs + l] = w;
                    p.w_cum[(long)b * L + l] += w;
                }
            }
        }
    }
    __syncthreads();
    GVX_STAMP(2, 1);

    // ---- context columns slice (weights past the row's length are exactly 0, so clamped loads are harmless)
    for (int c0 = c_begin; c0 < c_end; c0 += 32) {
        const int e4 = c0 + cc;
        const int e4c = min(e4, e4n - 1);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int lb = 0; lb < L; lb += 8 * CX_MV) {
            if (c0 != c_begin || lb != 0) {
#pragma unroll
                for (int i = 0; i < CX_MV; ++i) mv[i] = mbase[(long)min(lb + pg + 8 * i, L - 1) * e4n + e4c];
            }
#pragma unroll
            for (int i = 0; i < CX_MV; ++i) {
                const int l = lb + pg + 8 * i;
                const float w = l < L ? ws[l] : 0.f;
                acc.x = fmaf(w, mv[i].x, acc.x); acc.y = fmaf(w, mv[i].y, acc.y);
                acc.z = fmaf(w, mv[i].z, acc.z); acc.w = fmaf(w, mv[i].w, acc.w);
            }
        }
        red[pg * 32 + cc] = acc;
        GVX_STAMP(2, 2);
        __syncthreads();
        if (tid < 32 && e4 < c_end) {
            float4 o = red[tid];
#pragma unroll
            for (int i = 1; i < 8; ++i) {
                const float4 t = red[i * 32 + tid];
                o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
            }
            const int e = 4 * e4;  // blocked context vector [E/8][B][8]
            *reinterpret_cast<float4*>(p.ctx_out + (long)(e >> 3) * B * 8 + b * 8 + (e & 7)) = o;
        }
        __syncthreads();
        GVX_STAMP(2, 3);
    }
}

// The one-launch attention step (attn_step_body) lives in attn_step_body.h: the autoregressive loop also runs it as extra
// workgroups of a weight-streaming launch (skinny.hip, ar_attn_tiles_kernel).
template <int NJ>
__global__ __launch_bounds__(MA_THREADS) void attn_step_kernel(AttnParams p) { attn_step_body<NJ>(p, (int)blockIdx.x); }

// context-column slices per row for the one-launch step: aim at one workgroup per CU (256), at least 8 float4 columns each
int attention_slices(int B, int E) {
    int S = 16;
    while (S > 1 && (S * B > 256 || (E >> 2) / S < 8)) S >>= 1;
    return S;
}

int attention_groups(int B, int L) {
    int G = 8;
    while (G > 1 && (G * B > 128 || G * 8 > L)) G >>= 1;
    return G;
}

bool attention_supported(int L, int a, int F, int kl, int E) {
    if (F > AT_FP || a > 256 || (E & 7) || (a & 3) || a < 4) return false;
    const size_t lds_c = (size_t)context_lds_layout(L).total * sizeof(float);
    const size_t lds_s = (size_t)step_lds_layout(a, L).total * sizeof(float);
    return lds_c <= 160 * 1024 && lds_s <= 160 * 1024 && (size_t)(2 * (L + kl) + 2 * kl * 32 + 32 * 32 + 8) * sizeof(float) <= 160 * 1024;
}

hipError_t attention_init() {
    hipError_t e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_energy_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_energy_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_energy_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_step_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_step_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_step_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(attn_context_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t launch_attention_step(const AttnParams& p, hipStream_t s) {
    if (!attention_supported(p.L, p.a, p.F, p.kl, p.E) || p.G < 1) return hipErrorInvalidValue;
    const size_t lds = (size_t)step_lds_layout(p.a, p.L).total * sizeof(float);
    const dim3 grid(8 * ((p.B + 7) / 8) * p.G), block(MA_THREADS);
    if (p.a <= 32) attn_step_kernel<1><<<grid, block, lds, s>>>(p);
    else if (p.a <= 128) attn_step_kernel<4><<<grid, block, lds, s>>>(p);
    else attn_step_kernel<8><<<grid, block, lds, s>>>(p);
    return hipGetLastError();
}

hipError_t launch_attention(const AttnParams& p, hipStream_t s) {
    if (!attention_supported(p.L, p.a, p.F, p.kl, p.E) || p.G < 1) return hipErrorInvalidValue;
    const size_t lds_e = (size_t)energy_lds_layout(p.a).total * sizeof(float);
    const size_t lds_c = (size_t)context_lds_layout(p.L).total * sizeof(float);
    const dim3 grid(p.G, p.B), block(AT_THREADS), eblock(EN_THREADS);
    const int dpl = (p.a + 63) / 64;
    if (dpl == 1) attn_energy_kernel<1><<<grid, eblock, lds_e, s>>>(p);
    else if (dpl == 2) attn_energy_kernel<2><<<grid, eblock, lds_e, s>>>(p);
    else attn_energy_kernel<4><<<grid, eblock, lds_e, s>>>(p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    attn_context_kernel<<<grid, block, lds_c, s>>>(p);
    return hipGetLastError();
}

#ifdef GVX_STAMPS
hipError_t read_stamps_attention(unsigned long long* host96) {
    return hipMemcpyFromSymbol(host96, HIP_SYMBOL(gvx_stamps), sizeof(unsigned long long) * 96);
}
#endif

}  // namespace gvx
