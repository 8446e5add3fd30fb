// One-launch attention step of the decoder loops: the workgroup body, shared by attn_step_kernel (attention.hip) and by the
// autoregressive launch that runs it beside weight-streaming tiles (skinny.hip).  See attention.hip for the algorithm.
#pragma once
#include "gvx_kernels.h"

namespace gvx {

constexpr int EN_QG = 16;    // slab rows summed in parallel for the query

__device__ __forceinline__ float fast_tanh(float x) {
    // 1 - 2/(exp(2x)+1): v_exp_f32 + v_rcp_f32, abs error ~1e-7; saturates correctly at +-inf
    // (__fdividef compiles to the full IEEE division sequence - div_scale / rcp / 4 fma / div_fmas / div_fixup - under
    // the flags this library is built with; the reciprocal instruction is asked for by name)
    const float e = __expf(2.f * x);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}

// ---------------------------------------------------------------------------------------------------------------------
// One launch per decoder step (replaces the energy + context pair, i.e. one dependent kernel boundary, one launch ramp and
// one first-byte round trip per step): workgroup (g, b) computes ALL energies of row b (redundantly in the G workgroups of
// the row - the row's pm + location features are 64 KB, its query slabs 64 KB), the softmax, and the g-th slice of the
// context columns.  The only cross-workgroup dependency left is the one the kernel boundary carries (query slabs / location
// features of the LSTM launch), so there is still no atomic, no in-kernel hand-off, and every bit is reproducible.
//   `ploc` = pm + location features, summed by the location workgroups of the LSTM launch (skinny.hip, loc_body).
// The kernel is a latency chain on the step's critical path and is bound by instruction ISSUE, not by bytes (measured:
// in-kernel time tracked the straight-line code size at ~0.5 us per KiB): every global load is issued before the first
// wait, loads are 16 B wide with 32-bit offsets from uniform bases, cross-lane sums use DPP (VALU) instead of LDS permutes,
// and the softmax weights are recomputed by the threads that need them instead of being published through LDS.
constexpr int MA_THREADS = 512;
constexpr int MA_WAVES = MA_THREADS / 64;
constexpr int MA_PG = 2;     // groups of 8 positions per wave held in registers (first pass: 8 * MA_WAVES * MA_PG positions)
constexpr int MA_MV = 8;     // memory float4 loads per thread held in registers

template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {   // v of the lane selected by the DPP control (row = 16 lanes)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum8(float v) {      // all lanes of each aligned group of 8 end up with the group's sum
    v += dpp_get<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_get<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_get<0x141>(v);   // row_half_mirror
    return v;
}
__device__ __forceinline__ float lane_bcast(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ float wave_sum_dpp(float v) {   // uniform result
    v = sum8(v);
    v += dpp_get<0x140>(v);   // row_mirror: 16-lane rows
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = fmaxf(v, dpp_get<0xB1>(v));
    v = fmaxf(v, dpp_get<0x4E>(v));
    v = fmaxf(v, dpp_get<0x141>(v));
    v = fmaxf(v, dpp_get<0x140>(v));
    return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
}

struct StepLds { int qp_off, q_off, e_off, red_off, total; };
__host__ __device__ inline StepLds step_lds_layout(int a, int L) {
    StepLds o;
    auto al = [](int x) { return (x + 3) & ~3; };
    int off = 0;
    o.qp_off = off; off += al(EN_QG * a);
    o.q_off = off; off += al(a);
    o.e_off = off; off += al(L);
    o.red_off = off; off += MA_WAVES * 32 * 4;
    o.total = off;
    return o;
}

template <int NJ, int MV = MA_MV>   // NJ: float4 groups of the attention dim per lane (8 lanes share a position): a <= 32 * NJ;
                                    // MV: memory float4 loads per thread in flight (the launch shared with tiles keeps fewer registers)
__device__ __forceinline__ void attn_step_body(const AttnParams& p, const int block_id) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = p.L, a = p.a, E = p.E, B = p.B;
    const StepLds lo = step_lds_layout(a, L);
    float* qp = smem + lo.qp_off;
    float* qs = smem + lo.q_off;
    float* es = smem + lo.e_off;
    float4* red = reinterpret_cast<float4*>(smem + lo.red_off);

    // Block -> (row, slice) mapping: workgroups are dealt round-robin over the 8 XCDs (blocks i and i + 8 share an L2), and
    // the G slices of a row all read the same 128 KB (query slabs + ploc).  Slices of one row therefore take block ids that
    // are congruent mod 8: the row's bytes are fetched into that XCD's L2 once.  (Speed only - nothing depends on placement.)
    // prefetch for the next launch (AttnParams.pf_*): this wave's first four k-groups of tile `block_id`, as that launch will read them
    float4 pfv0 = make_float4(0.f, 0.f, 0.f, 0.f), pfv1 = pfv0;
    const bool pf = p.pf_w[0] != nullptr && block_id < p.pf_tiles;
    if (pf) {
        const int j = block_id >= p.pf_tiles0, t = block_id - (j ? p.pf_tiles0 : 0);
        const int nkg = p.pf_nkg[j], per = (nkg + 7) >> 3;
        const int w_ = (int)threadIdx.x >> 6, kb = min(w_ * per, nkg - 1);
        const float4* wp = reinterpret_cast<const float4*>(p.pf_w[j]) + ((long)t * nkg + kb) * 64 + (threadIdx.x & 63);
        pfv0 = wp[0];
        pfv1 = wp[kb + 1 < nkg ? 64 : 0];
        const float4 c2 = wp[kb + 2 < nkg ? 128 : 0], c3 = wp[kb + 3 < nkg ? 192 : 0];
        pfv0.y += c2.x; pfv1.y += c3.x;
    }
    const int xcd = block_id & 7, jb = block_id >> 3;
    const int b = (jb / p.G) * 8 + xcd, g = jb % p.G;
    if (b >= B) return;   // uniform per workgroup (padding blocks when B is not a multiple of 8)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int len = p.lengths ? p.lengths[b] : L;
    GVX_STAMP(1, 0);

    // ---- loads: query slabs.  thread = (grp, d4): slabs grp, grp + qg, ... of float4 column d4
    const unsigned a4 = (unsigned)a >> 2;
    const int qg = min(EN_QG, MA_THREADS / (int)a4);
    const unsigned grp = (unsigned)tid / a4, d4 = (unsigned)tid - grp * a4;
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int QV = 8;
    float4 qld[QV];
    const bool q_fast = (int)grp < qg && p.n_slabs == QV * qg;   // default dims: 128 slabs = 8 x 16
    {
        const float4* base = reinterpret_cast<const float4*>(p.q_slab) + (unsigned)b * a4;
        const unsigned tstride = (unsigned)B * a4;
        if (q_fast) {
#pragma unroll
            for (int i = 0; i < QV; ++i) qld[i] = base[(grp + (unsigned)(i * qg)) * tstride + d4];
        } else if ((int)grp < qg) {
            for (unsigned t = grp; t < (unsigned)p.n_slabs; t += qg) {
                const float4 v = base[t * tstride + d4];
                s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
            }
        }
    }
    // ---- loads: ploc rows.  lane = (a8 = lane & 7, p8 = lane >> 3): 8 lanes share a position and hold the float4 groups
    // a8, a8 + 8, ... of its row (a wave instruction reads 8 rows x 128 contiguous bytes); the wave's i-th group of 8
    // positions is 8 * (wave + MA_WAVES * i) + p8
    const unsigned a8 = lane & 7, p8 = lane >> 3;
    const float4* prow = reinterpret_cast<const float4*>(p.loc) + (unsigned)b * (unsigned)L * a4;
    float4 pv[MA_PG][NJ];
#pragma unroll
    for (int i = 0; i < MA_PG; ++i) {
        const unsigned l = min(8u * (unsigned)(wave + MA_WAVES * i) + p8, (unsigned)L - 1u);
#pragma unroll
        for (int j = 0; j < NJ; ++j) pv[i][j] = prow[l * a4 + min(a8 + 8u * j, a4 - 1u)];
    }
    // ---- loads: this workgroup's slice of the memory columns.  thread = (cc = float4 column, pg = position group); a wave
    // covers CC columns x 64/CC position groups, so its loads are rows of CC * 16 contiguous bytes
    const unsigned e4n = (unsigned)E >> 2;
    const int cols = ((int)e4n + p.G - 1) / p.G;
    const int c_begin = g * cols, c_end = min((int)e4n, c_begin + cols);
    const int cshift = cols <= 8 ? 3 : (cols <= 16 ? 4 : 5);      // column block of 8 / 16 / 32 float4
    const int CC = 1 << cshift, npg = MA_THREADS >> cshift;
    const int cc = tid & (CC - 1), pg = tid >> cshift;
    const float4* mbase = reinterpret_cast<const float4*>(p.memory) + (unsigned)b * (unsigned)L * e4n;
    float4 mv[MV];
    {
        const unsigned e4 = min((unsigned)(c_begin + cc), e4n - 1u);
#pragma unroll
        for (int i = 0; i < MV; ++i) mv[i] = mbase[min((unsigned)(pg + npg * i), (unsigned)L - 1u) * e4n + e4];
    }
    float4 vv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        vv[j] = reinterpret_cast<const float4*>(p.v)[min(a8 + 8u * j, a4 - 1u)];
        if (a8 + 8u * j >= a4) vv[j] = make_float4(0.f, 0.f, 0.f, 0.f);   // lanes past the attention dim contribute nothing
    }
    GVX_STAMP(1, 1);

    // ---- q = sum of the slabs (fixed order): per-thread partial -> 16 partial rows in LDS -> one thread per dim adds them
    if (q_fast) {
#pragma unroll
        for (int i = 0; i < QV; ++i) { s4.x += qld[i].x; s4.y += qld[i].y; s4.z += qld[i].z; s4.w += qld[i].w; }
    }
    if ((int)grp < qg) reinterpret_cast<float4*>(qp + grp * a)[d4] = s4;
    __syncthreads();
    if (tid < a) {
        float acc = qp[tid];
        for (int gq = 1; gq < qg; ++gq) acc += qp[gq * a + tid];
        qs[tid] = acc;
    }
    __syncthreads();
    float4 qv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) qv[j] = reinterpret_cast<const float4*>(qs)[min(a8 + 8u * j, a4 - 1u)];
    GVX_STAMP(1, 2);

    // ---- energies of the whole row
    const int pass = 8 * MA_WAVES * MA_PG;
    for (int l0 = 0; l0 < L; l0 += pass) {
        if (l0) {
#pragma unroll
            for (int i = 0; i < MA_PG; ++i) {
                const unsigned l = min((unsigned)l0 + 8u * (unsigned)(wave + MA_WAVES * i) + p8, (unsigned)L - 1u);
#pragma unroll
                for (int j = 0; j < NJ; ++j) pv[i][j] = prow[l * a4 + min(a8 + 8u * j, a4 - 1u)];
            }
        }
#pragma unroll
        for (int i = 0; i < MA_PG; ++i) {
            float pe = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                pe = fmaf(vv[j].x, fast_tanh(qv[j].x + pv[i][j].x), pe);
                pe = fmaf(vv[j].y, fast_tanh(qv[j].y + pv[i][j].y), pe);
                pe = fmaf(vv[j].z, fast_tanh(qv[j].z + pv[i][j].z), pe);
                pe = fmaf(vv[j].w, fast_tanh(qv[j].w + pv[i][j].w), pe);
            }
            pe = sum8(pe);
            const int l = l0 + 8 * (wave + MA_WAVES * i) + (int)p8;
            if (a8 == 0 && l < L) es[l] = l < len ? pe : -INFINITY;
        }
    }
    __syncthreads();
    GVX_STAMP(1, 3);

    // ---- masked softmax over the row: every wave computes the normaliser (uniform), every thread the weights it needs
    float mx = -INFINITY;
    for (int l = lane; l < L; l += 64) mx = fmaxf(mx, es[l]);
    mx = wave_max_dpp(mx);
    float sum = 0.f;
    for (int l = lane; l < L; l += 64) sum += __expf(es[l] - mx);
    sum = wave_sum_dpp(sum);
    const float inv = 1.f / sum;
    if (g == 0 && wave == 0) {
        for (int l = lane; l < L; l += 64) {
            const float w = __expf(es[l] - mx) * inv;
            p.w_out[(long)b * p.w_out_bs + l] = w;
            p.w_cum[(long)b * L + l] += w;
        }
    }
    GVX_STAMP(1, 4);

    // ---- context columns slice (weights past the row's length are exactly 0: exp(-inf) = 0, so clamped loads are harmless)
    for (int c0 = c_begin; c0 < c_end; c0 += CC) {
        const int e4 = c0 + cc;
        const unsigned e4c = min((unsigned)e4, e4n - 1u);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int lb = 0; lb < L; lb += npg * MV) {
            if (c0 != c_begin || lb != 0) {
#pragma unroll
                for (int i = 0; i < MV; ++i) mv[i] = mbase[min((unsigned)(lb + pg + npg * i), (unsigned)L - 1u) * e4n + e4c];
            }
#pragma unroll
            for (int i = 0; i < MV; ++i) {
                const int l = lb + pg + npg * i;
                if (l < L) {   // uniform for all but the last pass
                    const float w = __expf(es[l] - mx) * inv;
                    acc.x = fmaf(w, mv[i].x, acc.x); acc.y = fmaf(w, mv[i].y, acc.y);
                    acc.z = fmaf(w, mv[i].z, acc.z); acc.w = fmaf(w, mv[i].w, acc.w);
                }
            }
        }
        // position groups: 64 / CC of them sit in this wave's lanes (lane bits above cshift), the rest in the other waves
        for (int off = CC; off < 64; off <<= 1) {
            acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off);
            acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
        }
        if (lane < CC) red[wave * 32 + lane] = acc;
        __syncthreads();
        if (tid < CC && e4 < c_end) {
            float4 o = red[tid];
#pragma unroll
            for (int i = 1; i < MA_WAVES; ++i) {
                const float4 t = red[i * 32 + tid];
                o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
            }
            const int e = 4 * e4;  // blocked context vector [E/8][B][8]
            *reinterpret_cast<float4*>(p.ctx_out + (long)(e >> 3) * B * 8 + b * 8 + (e & 7)) = o;
        }
        if (c0 + CC < c_end) __syncthreads();
    }
    if (pf) asm volatile("" :: "v"(pfv0.x), "v"(pfv0.w), "v"(pfv1.x), "v"(pfv1.w));   // (the loads only have to have happened)
    GVX_STAMP(1, 5);
}

}  // namespace gvx
