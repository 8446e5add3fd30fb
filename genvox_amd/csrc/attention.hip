// Location-sensitive attention, one decoder step, one 512-thread workgroup per batch row.
// Replaces Attention.forward / get_alignment_energies / LocationLayer.forward of the reference
// (models/tts/tacotron2.py:89-129, :48-53) plus the cumulative-weights update (:353).
//
//   q      = sum of the per-tile partial products written by the attention-LSTM kernel (skinny.hip)
//   f[l,c] = conv1d_k31( [w_prev ; w_cum] )                       (2 -> F filters)
//   e[l]   = v . tanh(q + pm[l,:] + Wd f[l,:]),  -inf for l >= len
//   w      = softmax_l(e);  w_cum += w;  ctx = sum_l w[l] * memory[l,:]
//
// Everything between the inputs and (w, ctx) stays in LDS/registers; per step a row reads its
// processed memory (L*a floats) and memory (L*E floats) once from L2 with coalesced 16-byte loads.
// The dense location projection keeps the [a x F] matrix in registers (two attention dims per lane),
// a wave processes one position at a time and reduces over the attention dim with wave shuffles.
#include "gvx_kernels.h"

namespace gvx {

constexpr int AT_THREADS = 512;
constexpr int AT_WAVES = AT_THREADS / 64;
constexpr int AT_LC = 128;   // positions per chunk for the location features
constexpr int AT_FP = 32;    // max location filters (register/LDS row width)

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

struct AttnLds {
    int q_off, v_off, wc_off, cw_off, f_off, e_off, red_off, part_off, total;  // float offsets
};
__host__ __device__ inline AttnLds attn_lds_layout(int L, int a, int F, int kl) {
    AttnLds o;
    auto al = [](int x) { return (x + 3) & ~3; };
    int off = 0;
    o.q_off = off; off += al(a * 16);                   // q partial sums [<=16][a]
    o.v_off = off; off += al(a);
    o.wc_off = off; off += al(2 * (L + kl - 1));        // [2][L + kl - 1] with zero halo
    o.cw_off = off; off += al(2 * kl * AT_FP);          // conv weights [2][kl][AT_FP]
    o.f_off = off; off += AT_LC * AT_FP;                // location features of the current chunk
    o.e_off = off; off += al(L);                        // energies, then weights
    o.red_off = off; off += 4 * 512 + 32;               // context partials [4][E<=512] + scalars
    o.part_off = off; off += AT_LC * 65;                // per-lane energy partials [AT_LC][64 (+1 pad)]
    o.total = off;
    return o;
}
size_t attention_lds_bytes(int L, int a, int F, int kl) { return (size_t)attn_lds_layout(L, a, F, kl).total * sizeof(float); }

template <int DPL>
__global__ __launch_bounds__(AT_THREADS) void attention_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const AttnLds lo = attn_lds_layout(p.L, p.a, p.F, p.kl);
    float* qs = smem + lo.q_off;
    float* vs = smem + lo.v_off;
    float* wc = smem + lo.wc_off;
    float* cw = smem + lo.cw_off;
    float* fb = smem + lo.f_off;
    float* es = smem + lo.e_off;
    float* red = smem + lo.red_off;
    float* part = smem + lo.part_off;

    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = p.L, a = p.a, F = p.F, kl = p.kl, E = p.E;
    const int pad = (kl - 1) / 2, LW = L + kl - 1;
    const int len = p.lengths ? p.lengths[b] : L;

    // ---- stage: attention query = sum of the LSTM kernel's per-tile partial products.  Rows of a/4 float4;
    // AT_THREADS/(a/4) slab rows are summed in parallel (<= 16 groups), 8 independent loads in flight per thread.
    const int a4 = a >> 2;
    const int qgroups = min(16, AT_THREADS / a4);
    {
        const int grp = tid / a4, d4 = tid - grp * a4;
        if (grp < qgroups) {
            float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4* base = reinterpret_cast<const float4*>(p.q_slab + (long)b * a) + d4;
            const long tstride = (long)p.B * a4;
            int t = grp;
            for (; t + 7 * qgroups < p.n_slabs; t += 8 * qgroups) {
                float4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = base[(long)(t + i * qgroups) * tstride];
#pragma unroll
                for (int i = 0; i < 8; ++i) { s4.x += v[i].x; s4.y += v[i].y; s4.z += v[i].z; s4.w += v[i].w; }
            }
            for (; t < p.n_slabs; t += qgroups) {
                const float4 v = base[(long)t * tstride];
                s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
            }
            reinterpret_cast<float4*>(qs + grp * a)[d4] = s4;
        }
    }
    for (int idx = tid; idx < a; idx += AT_THREADS) vs[idx] = p.v[idx];
    for (int idx = tid; idx < 2 * LW; idx += AT_THREADS) {
        const int ch = idx / LW, i = idx - ch * LW, l = i - pad;
        float val = 0.f;
        if (l >= 0 && l < L) val = ch == 0 ? (p.w_prev ? p.w_prev[(long)b * p.w_prev_bs + l] : 0.f) : p.w_cum[(long)b * L + l];
        wc[idx] = val;
    }
    for (int idx = tid; idx < 2 * kl * AT_FP; idx += AT_THREADS) {
        const int c = idx % AT_FP, ck = idx / AT_FP;  // ck = ch*kl + k
        cw[idx] = c < F ? p.loc_conv[(long)c * 2 * kl + ck] : 0.f;
    }
    // dense location weights: lane owns attention dims d = lane + 64*i
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
        if (d < a) for (int g = 0; g < qgroups; ++g) qsum += qs[g * a + d];
        qv[i] = qsum;
        vv[i] = d < a ? vs[d] : 0.f;
    }

    // ---- location features + energies, AT_LC positions at a time
    for (int l0 = 0; l0 < L; l0 += AT_LC) {
        const int lc = min(AT_LC, L - l0);
        // processed-memory values of this wave's positions: issued now, consumed after the conv (latency hidden)
        float pmv[AT_LC / AT_WAVES][DPL];
#pragma unroll
        for (int j = 0; j < AT_LC / AT_WAVES; ++j) {
            const int ll = wave + j * AT_WAVES;
            const float* pmrow = p.pm + ((long)b * L + l0 + min(ll, lc - 1)) * a;
#pragma unroll
            for (int i = 0; i < DPL; ++i) {
                const int d = lane + 64 * i;
                pmv[j][i] = d < a ? pmrow[d] : 0.f;
            }
        }
        {   // conv: thread = (position, quarter of the filters)
            const int ll = tid & (AT_LC - 1), cq = tid >> 7;  // 4 quarters of 8 filters
            float acc[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = 0.f;
            if (ll < lc) {
                for (int ch = 0; ch < 2; ++ch) {
                    const float* xrow = wc + ch * LW + l0 + ll;
                    const float* wrow = cw + (ch * kl) * AT_FP + cq * 8;
                    for (int k = 0; k < kl; ++k) {
                        const float x = xrow[k];
                        const float4 w0 = *reinterpret_cast<const float4*>(wrow + k * AT_FP);
                        const float4 w1 = *reinterpret_cast<const float4*>(wrow + k * AT_FP + 4);
                        acc[0] = fmaf(w0.x, x, acc[0]); acc[1] = fmaf(w0.y, x, acc[1]);
                        acc[2] = fmaf(w0.z, x, acc[2]); acc[3] = fmaf(w0.w, x, acc[3]);
                        acc[4] = fmaf(w1.x, x, acc[4]); acc[5] = fmaf(w1.y, x, acc[5]);
                        acc[6] = fmaf(w1.z, x, acc[6]); acc[7] = fmaf(w1.w, x, acc[7]);
                    }
                }
            }
            float* frow = fb + ll * AT_FP + cq * 8;
            *reinterpret_cast<float4*>(frow) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *reinterpret_cast<float4*>(frow + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
        __syncthreads();
        // energies: a wave takes positions wave, wave+8, ...; lane = attention dim(s); per-lane partials go to LDS
        // and are reduced over the 64 lanes in one batched pass (instead of 6 cross-lane shuffles per position)
#pragma unroll
        for (int j = 0; j < AT_LC / AT_WAVES; ++j) {
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
        {   // thread = (position, quarter of the lanes): 16 partials each, then 2 shuffles
            const int ll = tid >> 2, qd = tid & 3;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) s += part[ll * 65 + qd * 16 + i];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            const int l = l0 + ll;
            if (qd == 0 && ll < lc) es[l] = l < len ? s : -INFINITY;
        }
        __syncthreads();
    }

    // ---- masked softmax over positions (wave 0), new weights -> LDS + global, cumulative update
    if (wave == 0) {
        float m = -INFINITY;
        for (int l = lane; l < L; l += 64) m = fmaxf(m, es[l]);
        m = wave_max(m);
        float s = 0.f;
        for (int l = lane; l < L; l += 64) {
            const float ex = expf(es[l] - m);
            es[l] = ex;
            s += ex;
        }
        s = wave_sum(s);
        const float inv = 1.f / s;
        for (int l = lane; l < L; l += 64) {
            const float w = es[l] * inv;
            es[l] = w;
            p.w_out[(long)b * p.w_out_bs + l] = w;
            p.w_cum[(long)b * L + l] = wc[LW + pad + l] + w;
        }
    }
    __syncthreads();

    // ---- context: thread = (float4 column group, position residue mod 4)
    const int e4n = E >> 2;  // E % 4 == 0
    for (int c0 = 0; c0 < e4n; c0 += 128) {
        const int e4 = c0 + (tid & 127), lp = tid >> 7;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e4 < e4n) {
            const float4* mrow = reinterpret_cast<const float4*>(p.memory + (long)b * L * E) + e4;
#pragma unroll 8
            for (int l = lp; l < len; l += 4) {
                const float w = es[l];
                const float4 mv = mrow[(long)l * e4n];
                acc.x = fmaf(w, mv.x, acc.x); acc.y = fmaf(w, mv.y, acc.y);
                acc.z = fmaf(w, mv.z, acc.z); acc.w = fmaf(w, mv.w, acc.w);
            }
        }
        float4* r4 = reinterpret_cast<float4*>(red);
        r4[lp * 128 + (tid & 127)] = acc;
        __syncthreads();
        if (tid < 128 && e4 < e4n) {
            const float4 a0 = r4[tid], a1 = r4[128 + tid], a2 = r4[256 + tid], a3 = r4[384 + tid];
            float4 o;
            o.x = (a0.x + a1.x) + (a2.x + a3.x); o.y = (a0.y + a1.y) + (a2.y + a3.y);
            o.z = (a0.z + a1.z) + (a2.z + a3.z); o.w = (a0.w + a1.w) + (a2.w + a3.w);
            reinterpret_cast<float4*>(p.ctx_out + (long)b * p.ctx_bs)[e4] = o;
        }
        __syncthreads();
    }
}

hipError_t attention_init() {
    // dynamic LDS above the 64 KiB default is requested per launch size; allow the maximum once
    hipError_t e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t launch_attention(const AttnParams& p, hipStream_t s) {
    if (p.F > AT_FP || p.a > 256 || (p.E & 3) || (p.a & 3)) return hipErrorInvalidValue;
    const size_t lds = attention_lds_bytes(p.L, p.a, p.F, p.kl);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const int dpl = (p.a + 63) / 64;
    if (dpl == 1) hipLaunchKernelGGL(attention_kernel<1>, dim3(p.B), dim3(AT_THREADS), lds, s, p);
    else if (dpl == 2) hipLaunchKernelGGL(attention_kernel<2>, dim3(p.B), dim3(AT_THREADS), lds, s, p);
    else hipLaunchKernelGGL(attention_kernel<4>, dim3(p.B), dim3(AT_THREADS), lds, s, p);
    return hipGetLastError();
}

}  // namespace gvx
