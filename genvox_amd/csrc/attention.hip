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

namespace gvx {

constexpr int AT_THREADS = 256;
constexpr int AT_WAVES = AT_THREADS / 64;
constexpr int EN_LC = 32;    // positions per pass of the energy kernel
constexpr int AT_FP = 32;    // max location filters (register/LDS row width)
constexpr int EN_QG = 8;     // slab rows summed in parallel for the query

__device__ __forceinline__ float fast_tanh(float x) {
    // 1 - 2/(exp(2x)+1): v_exp_f32 + v_rcp_f32, abs error ~1e-7; saturates correctly at +-inf
    const float e = __expf(2.f * x);
    return 1.f - __fdividef(2.f, e + 1.f);
}

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

struct EnergyLds { int q_off, v_off, wc_off, cw_off, f_off, part_off, total; };
__host__ __device__ inline EnergyLds energy_lds_layout(int Lg, int a, int kl) {
    EnergyLds o;
    auto al = [](int x) { return (x + 3) & ~3; };
    int off = 0;
    o.q_off = off; off += al(EN_QG * a);
    o.v_off = off; off += al(a);
    o.wc_off = off; off += al(2 * (Lg + kl - 1));   // [2][chunk + halo]
    o.cw_off = off; off += al(2 * kl * AT_FP);      // conv weights [2][kl][AT_FP]
    o.f_off = off; off += EN_LC * AT_FP;
    o.part_off = off; off += EN_LC * 65;
    o.total = off;
    return o;
}

template <int DPL>
__global__ __launch_bounds__(AT_THREADS) void attn_energy_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = p.L, a = p.a, F = p.F, kl = p.kl;
    const int Lg = chunk_len(L, p.G);
    const EnergyLds lo = energy_lds_layout(Lg, a, kl);
    float* qs = smem + lo.q_off;
    float* vs = smem + lo.v_off;
    float* wc = smem + lo.wc_off;
    float* cw = smem + lo.cw_off;
    float* fb = smem + lo.f_off;
    float* part = smem + lo.part_off;

    const int g = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l_begin = g * Lg, l_end = min(L, l_begin + Lg);
    if (l_begin >= l_end) return;  // uniform per workgroup
    const int pad = (kl - 1) / 2, LW = Lg + kl - 1;
    const int len = p.lengths ? p.lengths[b] : L;

    // ---- query: sum of the LSTM kernel's per-tile partial slabs, EN_QG slab rows in parallel, float4 per thread
    const int a4 = a >> 2;
    const int qg = min(EN_QG, AT_THREADS / a4);  // slab rows summed in parallel
    {
        const int grp = tid / a4, d4 = tid - grp * a4;
        if (grp < qg) {
            float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4* base = reinterpret_cast<const float4*>(p.q_slab + (long)b * a) + d4;
            const long tstride = (long)p.B * a4;
            int t = grp;
            for (; t + 7 * qg < p.n_slabs; t += 8 * qg) {
                float4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = base[(long)(t + i * qg) * tstride];
#pragma unroll
                for (int i = 0; i < 8; ++i) { s4.x += v[i].x; s4.y += v[i].y; s4.z += v[i].z; s4.w += v[i].w; }
            }
            for (; t < p.n_slabs; t += qg) {
                const float4 v = base[(long)t * tstride];
                s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
            }
            reinterpret_cast<float4*>(qs + grp * a)[d4] = s4;
        }
    }
    for (int idx = tid; idx < a; idx += AT_THREADS) vs[idx] = p.v[idx];
    // previous / cumulative weights of the chunk with a zero-filled halo of (kl-1)/2 positions
    for (int idx = tid; idx < 2 * LW; idx += AT_THREADS) {
        const int ch = idx / LW, i = idx - ch * LW, l = l_begin + i - pad;
        float val = 0.f;
        if (l >= 0 && l < L) val = ch == 0 ? (p.w_prev ? p.w_prev[(long)b * p.w_prev_bs + l] : 0.f) : p.w_cum[(long)b * L + l];
        wc[idx] = val;
    }
    for (int idx = tid; idx < 2 * kl * AT_FP; idx += AT_THREADS) {
        const int c = idx % AT_FP, ck = idx / AT_FP;  // ck = ch*kl + k
        cw[idx] = c < F ? p.loc_conv[(long)c * 2 * kl + ck] : 0.f;
    }
    // dense location weights: lane owns attention dims d = lane + 64*i (kept in registers for the whole kernel)
    float wd[DPL][AT_FP];
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
        const int d = lane + 64 * i;
#pragma unroll
        for (int c = 0; c < AT_FP; ++c) wd[i][c] = (d < a && c < F) ? p.loc_dense[(long)d * F + c] : 0.f;
    }
    __syncthreads();
    float qv[DPL], vv[DPL];
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
        const int d = lane + 64 * i;
        float qsum = 0.f;
        if (d < a)
            for (int gq = 0; gq < qg; ++gq) qsum += qs[gq * a + d];
        qv[i] = qsum;
        vv[i] = d < a ? vs[d] : 0.f;
    }

    for (int l0 = l_begin; l0 < l_end; l0 += EN_LC) {
        const int lc = min(EN_LC, l_end - l0);
        // processed-memory values of this wave's positions: issued now, consumed after the conv (latency hidden)
        float pmv[EN_LC / AT_WAVES][DPL];
#pragma unroll
        for (int j = 0; j < EN_LC / AT_WAVES; ++j) {
            const int ll = wave + j * AT_WAVES;
            const float* pmrow = p.pm + ((long)b * L + l0 + min(ll, lc - 1)) * a;
#pragma unroll
            for (int i = 0; i < DPL; ++i) {
                const int d = lane + 64 * i;
                pmv[j][i] = d < a ? pmrow[d] : 0.f;
            }
        }
        {   // location conv: thread = (position, group of 4 filters)
            const int ll = tid & (EN_LC - 1), fg = tid >> 5;  // 8 groups of 4 filters
            float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
            if (ll < lc) {
                for (int ch = 0; ch < 2; ++ch) {
                    const float* xrow = wc + ch * LW + (l0 - l_begin) + ll;
                    const float* wrow = cw + (ch * kl) * AT_FP + fg * 4;
                    for (int k = 0; k < kl; ++k) {
                        const float x = xrow[k];
                        const float4 w = *reinterpret_cast<const float4*>(wrow + k * AT_FP);
                        acc0 = fmaf(w.x, x, acc0); acc1 = fmaf(w.y, x, acc1);
                        acc2 = fmaf(w.z, x, acc2); acc3 = fmaf(w.w, x, acc3);
                    }
                }
            }
            *reinterpret_cast<float4*>(fb + ll * AT_FP + fg * 4) = make_float4(acc0, acc1, acc2, acc3);
        }
        __syncthreads();
        // energies: a wave takes positions wave, wave+4, ...; lane = attention dim(s); per-lane partials go to LDS
        // and are reduced over the 64 lanes in one batched pass (instead of 6 cross-lane shuffles per position)
#pragma unroll
        for (int j = 0; j < EN_LC / AT_WAVES; ++j) {
            const int ll = wave + j * AT_WAVES;
            float pe = 0.f;
            const float* frow = fb + ll * AT_FP;
#pragma unroll
            for (int i = 0; i < DPL; ++i) {
                float s = qv[i] + pmv[j][i];
#pragma unroll
                for (int c4 = 0; c4 < AT_FP / 4; ++c4) {
                    const float4 fv = *reinterpret_cast<const float4*>(frow + 4 * c4);
                    s = fmaf(wd[i][4 * c4 + 0], fv.x, s);
                    s = fmaf(wd[i][4 * c4 + 1], fv.y, s);
                    s = fmaf(wd[i][4 * c4 + 2], fv.z, s);
                    s = fmaf(wd[i][4 * c4 + 3], fv.w, s);
                }
                pe = fmaf(vv[i], fast_tanh(s), pe);
            }
            part[ll * 65 + lane] = pe;
        }
        __syncthreads();
        {   // thread = (position, eighth of the lanes): 8 partials each, then 3 shuffles
            const int ll = tid >> 3, sg = tid & 7;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += part[ll * 65 + sg * 8 + i];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            const int l = l0 + ll;
            if (sg == 0 && ll < lc) p.energies[(long)b * L + l] = l < len ? s : -INFINITY;
        }
        __syncthreads();
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

__global__ __launch_bounds__(AT_THREADS) void attn_context_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = p.L, E = p.E, B = p.B;
    const ContextLds lo = context_lds_layout(L);
    float* ws = smem + lo.e_off;
    float4* red = reinterpret_cast<float4*>(smem + lo.red_off);

    const int g = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int len = p.lengths ? p.lengths[b] : L;

    // ---- masked softmax over the whole row; every wave computes the normaliser, wave 0 publishes the weights
    {
        const float* erow = p.energies + (long)b * L;
        float m = -INFINITY;
        for (int l = lane; l < L; l += 64) m = fmaxf(m, erow[l]);
        m = wave_max(m);
        float s = 0.f;
        for (int l = lane; l < L; l += 64) s += expf(erow[l] - m);
        s = wave_sum(s);
        const float inv = 1.f / s;
        if (wave == 0) {
            for (int l = lane; l < L; l += 64) {
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

    // ---- context columns slice: thread = (float4 column, position residue mod 8)
    const int e4n = E >> 2;
    const int cols = (e4n + p.G - 1) / p.G;       // float4 columns per workgroup
    const int c_begin = g * cols, c_end = min(e4n, c_begin + cols);
    const int cc = tid & 31, pg = tid >> 5;
    for (int c0 = c_begin; c0 < c_end; c0 += 32) {
        const int e4 = c0 + cc;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e4 < c_end) {
            const float4* mrow = reinterpret_cast<const float4*>(p.memory + (long)b * L * E) + e4;
#pragma unroll 8
            for (int l = pg; l < len; l += 8) {
                const float w = ws[l];
                const float4 mv = mrow[(long)l * e4n];
                acc.x = fmaf(w, mv.x, acc.x); acc.y = fmaf(w, mv.y, acc.y);
                acc.z = fmaf(w, mv.z, acc.z); acc.w = fmaf(w, mv.w, acc.w);
            }
        }
        red[pg * 32 + cc] = acc;
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
    }
}

int attention_groups(int B, int L) {
    int G = 8;
    while (G > 1 && (G * B > 128 || G * 8 > L)) G >>= 1;
    return G;
}

bool attention_supported(int L, int a, int F, int kl, int E) {
    if (F > AT_FP || a > 256 || (E & 7) || (a & 3) || a < 4) return false;
    const size_t lds_e = (size_t)energy_lds_layout(chunk_len(L, 1), a, kl).total * sizeof(float);
    const size_t lds_c = (size_t)context_lds_layout(L).total * sizeof(float);
    return lds_e <= 160 * 1024 && lds_c <= 160 * 1024;
}

hipError_t attention_init() {
    hipError_t e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_energy_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_energy_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_energy_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(attn_context_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t launch_attention(const AttnParams& p, hipStream_t s) {
    if (!attention_supported(p.L, p.a, p.F, p.kl, p.E) || p.G < 1) return hipErrorInvalidValue;
    const size_t lds_e = (size_t)energy_lds_layout(chunk_len(p.L, p.G), p.a, p.kl).total * sizeof(float);
    const size_t lds_c = (size_t)context_lds_layout(p.L).total * sizeof(float);
    const dim3 grid(p.G, p.B), block(AT_THREADS);
    const int dpl = (p.a + 63) / 64;
    if (dpl == 1) attn_energy_kernel<1><<<grid, block, lds_e, s>>>(p);
    else if (dpl == 2) attn_energy_kernel<2><<<grid, block, lds_e, s>>>(p);
    else attn_energy_kernel<4><<<grid, block, lds_e, s>>>(p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    attn_context_kernel<<<grid, block, lds_c, s>>>(p);
    return hipGetLastError();
}

}  // namespace gvx
