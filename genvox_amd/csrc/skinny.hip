// Skinny recurrent GEMM for the sequential part of the path (batch rows <= 64):
//   decoder attention-LSTM and decoder-LSTM cells (one launch per decoder step covers both),
//   encoder BiLSTM recurrence (one launch per time step covers both directions),
//   the column-slice partial sums and the context projection of the autoregressive step (gvx_api.hip).
//
// Roofline: weight-streaming kernels. Every step re-reads the full recurrent matrices (71.3 MB fp32 for the
// two decoder cells) while each weight is used only B times, so the kernel is bound by HBM / Infinity-Cache
// bandwidth at B = 32 and is balanced against the fp32 MFMA rate at B = 64.
//
// Mapping (one workgroup of 8 waves per 32 output rows, normally one per CU):
//   * the weight matrix is pre-packed in MFMA-fragment order [tile][k-group][lane][4], so each wave
//     instruction reads 1 KiB contiguous - perfectly coalesced, streamed exactly once per step;
//   * K is split over the 8 waves (no barrier in the main loop, each wave streams its own slice straight
//     to VGPRs - an LDS round trip would be pure overhead for an operand nobody else reuses);
//   * x (the concatenated [input ; context ; hidden] vectors, <= 3 segments) is kept in the k-group-blocked
//     layout [K/8][B][8], so the x fragment of a k-group (32 rows x 8 k) is also ONE contiguous 1-KiB load
//     (row-major x costs 32 cache lines per load instruction and thrashes the 32-KiB L1: 4x over-fetch);
//   * v_mfma_f32_32x32x2_f32 with A = W (rows = outputs), B = x^T (cols = batch rows): lane (b, half)
//     ends up holding the four gate pre-activations i,f,g,o of hidden units 2g+half in accumulator
//     registers 4g..4g+3, because gate rows are packed as row = 4*j + gate.  The LSTM cell update is
//     therefore lane-local after the cross-wave K reduction (through LDS, 4 KiB per wave).
//   * attention-LSTM tiles also emit the partial products of the attention query projection for their 8
//     hidden units (slab[tile][b][:]), so the attention kernel never has to re-read the 512 KiB query matrix
//     per batch row.
#include "gvx_kernels.h"

namespace gvx {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// Prefetch depth (k-groups in flight per wave): swept on the MI355X in interleaved A/B runs - 4 (one batch tile) and
// 3 (two batch tiles) beat 8 / 6 by 6-8 %: the per-wave K slices (28 / 40 k-groups) then stay in the branch-free
// steady-state loop almost to the end, and 8 waves x 3 KiB per CU already cover the loaded-memory latency.
#ifndef SK_DEPTH1
#define SK_DEPTH1 4
#endif
#ifndef SK_DEPTH2
#define SK_DEPTH2 3
#endif
#ifndef SK_NWAVES
#define SK_NWAVES 8
#endif
constexpr int SK_WAVES = SK_NWAVES;
constexpr int SK_THREADS = SK_WAVES * 64;

struct SkinnyJobs {
    SkinnyJob job[3];
    int njobs;
    int tiles0;   // tiles of job 0
    int tiles1;   // tiles of job 1
    int tiles;    // tiles of all jobs; blocks >= tiles are location-feature workgroups
    LocJob loc;
};

constexpr int LOC_LC = 32;   // positions per pass
constexpr int LOC_FP = 32;   // location filters (padded)
__host__ __device__ inline int loc_chunk_len(int L, int G) { return (L + G - 1) / G; }
__host__ __device__ inline int loc_lds_floats(int L, int G, int kl) {
    return 2 * (loc_chunk_len(L, G) + kl - 1) + 4 + 2 * kl * LOC_FP + LOC_LC * LOC_FP;
}

// One workgroup = one (row b, chunk of positions): conv 2 -> 32 filters (k taps) then dense 32 -> a, written to
// loc_out[b][l][:].  512 threads; the dense weights of a lane's attention dims sit in registers.
__device__ __forceinline__ void loc_body(const LocJob& Q, int wg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = Q.L, a = Q.a, kl = Q.kl;
    const int g = wg % Q.G, b = wg / Q.G;
    const int Lg = loc_chunk_len(L, Q.G);
    const int l_begin = g * Lg, l_end = min(L, l_begin + Lg);
    if (l_begin >= l_end) return;
    const int pad = (kl - 1) / 2, LW = Lg + kl - 1;
    float* wc = smem;                                   // [2][LW]
    float* cw = smem + ((2 * LW + 3) & ~3);             // [2][kl][32]
    float* fb = cw + 2 * kl * LOC_FP;                   // [LOC_LC][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int idx = tid; idx < 2 * LW; idx += SK_THREADS) {
        const int ch = idx / LW, ii = idx - ch * LW, l = l_begin + ii - pad;
        float val = 0.f;
        if (l >= 0 && l < L) val = ch == 0 ? (Q.w_prev ? Q.w_prev[(long)b * Q.w_prev_bs + l] : 0.f) : Q.w_cum[(long)b * L + l];
        wc[idx] = val;
    }
    const int cw4 = (2 * kl * LOC_FP) >> 2;
    for (int idx = tid; idx < cw4; idx += SK_THREADS)
        reinterpret_cast<float4*>(cw)[idx] = reinterpret_cast<const float4*>(Q.loc_conv_t)[idx];
    __syncthreads();
    for (int l0 = l_begin; l0 < l_end; l0 += LOC_LC) {
        const int lc = min(LOC_LC, l_end - l0);
        {   // conv: thread = (position, pair of filters)
            const int ll = tid & (LOC_LC - 1), fg = tid >> 5;
            float acc0 = 0.f, acc1 = 0.f;
            if (ll < lc) {
                for (int ch = 0; ch < 2; ++ch) {
                    const float* xrow = wc + ch * LW + (l0 - l_begin) + ll;
                    const float* wrow = cw + (ch * kl) * LOC_FP + fg * 2;
#pragma unroll 8
                    for (int k = 0; k < kl; ++k) {
                        const float x = xrow[k];
                        const float2 w = *reinterpret_cast<const float2*>(wrow + k * LOC_FP);
                        acc0 = fmaf(w.x, x, acc0); acc1 = fmaf(w.y, x, acc1);
                    }
                }
            }
            *reinterpret_cast<float2*>(fb + ll * LOC_FP + fg * 2) = make_float2(acc0, acc1);
        }
        __syncthreads();
        // dense: wave -> positions wave, wave+8, ...; lane -> attention dims lane, lane+64, ...
        // (register budget: this body shares a kernel with the LSTM tiles and must stay under 128 VGPRs, or its workgroups
        // can no longer be co-resident with them - the loop over d0 stays rolled)
        constexpr int PM_L = LOC_LC / SK_WAVES;   // positions per wave and pass
        for (int d0 = 0; d0 < a; d0 += 64) {
            const int d = min(d0 + lane, a - 1);
            // processed-memory addends of this pass (one-launch attention step): issued together with the dense weights, so
            // they share that round trip instead of adding one per stored value
            float pmr[PM_L];
            if (Q.pm) {
#pragma unroll
                for (int li = 0; li < PM_L; ++li)
                    pmr[li] = Q.pm[((long)b * L + min(l0 + wave + li * SK_WAVES, l_end - 1)) * a + d];
            }
            float wd[LOC_FP];
#pragma unroll
            for (int c4 = 0; c4 < LOC_FP / 4; ++c4) {
                const float4 w4 = reinterpret_cast<const float4*>(Q.loc_dense_t)[(long)c4 * a + d];
                wd[4 * c4 + 0] = w4.x; wd[4 * c4 + 1] = w4.y; wd[4 * c4 + 2] = w4.z; wd[4 * c4 + 3] = w4.w;
            }
#pragma unroll
            for (int li = 0; li < PM_L; ++li) {
                const int ll = wave + li * SK_WAVES;
                if (ll < lc) {
                    const float* frow = fb + ll * LOC_FP;
                    float s = 0.f;   // one k-ordered chain: same summation order as a plain dot product over the filters
#pragma unroll
                    for (int c4 = 0; c4 < LOC_FP / 4; ++c4) {
                        const float4 fv = *reinterpret_cast<const float4*>(frow + 4 * c4);
                        s = fmaf(wd[4 * c4 + 0], fv.x, s);
                        s = fmaf(wd[4 * c4 + 1], fv.y, s);
                        s = fmaf(wd[4 * c4 + 2], fv.z, s);
                        s = fmaf(wd[4 * c4 + 3], fv.w, s);
                    }
                    if (d0 + lane < a) Q.loc_out[((long)b * L + l0 + ll) * a + d0 + lane] = Q.pm ? s + pmr[li] : s;
                }
            }
        }
        __syncthreads();
    }
}


// v_exp_f32 / v_rcp_f32 forms (abs error ~1e-7): the cell update sits on the per-step critical path after the
// workgroup barrier, where the libm expf/tanhf sequences cost ~1 us per step.
// (v_rcp_f32 by name: __fdividef compiles to the 10-instruction IEEE division sequence under this library's flags)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f); }

// Weight loads use the DEFAULT cache policy on purpose: the 71 MB of recurrent weights are re-read every step and stay
// resident in the 256-MiB Infinity Cache; non-temporal loads bypass it and were measured 16 % slower (round 1: 20.2 us vs
// 17.4 us per decoder step).

template <int MT, int DEPTH>
__device__ __forceinline__ void skinny_body(const SkinnyJobs& jobs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* red = smem;                                   // [SK_WAVES][MT][16][64]
    float* hs = smem + SK_WAVES * MT * 16 * 64;          // [MT*32][8] h' of this tile (LSTM + q slabs)

    int jsel = 0, tile = (int)blockIdx.x;
    if (jobs.njobs > 1 && tile >= jobs.tiles0) {
        jsel = 1; tile -= jobs.tiles0;
        if (jobs.njobs > 2 && tile >= jobs.tiles1) { jsel = 2; tile -= jobs.tiles1; }
    }
    const SkinnyJob& J = jobs.job[jsel];
    GVX_STAMP(0, 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar branches, counted waits
    const int bl = lane & 31, h = lane >> 5;
    const int B = J.B;
    const long blk = (long)B * 8;  // floats per k-group of a blocked vector

    // epilogue operands of this wave's unit (cell state, bias): fetched now so their latency hides under the main loop
    float c_pref = 0.f;
    float4 bias_pref = make_float4(0.f, 0.f, 0.f, 0.f);
    if (J.mode == 0 && wave < 4 * MT) {
        const int mt_ = wave >> 2, g_ = wave & 3;
        const int b_ = mt_ * 32 + bl, j_ = tile * 8 + 2 * g_ + h;
        if (b_ < B) c_pref = J.c[(long)b_ * (J.N >> 2) + j_];
        if (J.bias) bias_pref = *reinterpret_cast<const float4*>(J.bias + tile * 32 + 8 * g_ + 4 * h);
        // decoder cells finished from partial sums (autoregressive launches): the addend is known at launch, fetch it now
        if (J.addend && !J.seq_out && b_ < B) {
            const float4 ad = *reinterpret_cast<const float4*>(J.addend + (long)b_ * J.add_bs + tile * 32 + 8 * g_ + 4 * h);
            bias_pref.x += ad.x; bias_pref.y += ad.y; bias_pref.z += ad.z; bias_pref.w += ad.w;
        }
    }

    // ---- main loop: this wave's K slice, software pipelined DEPTH k-groups deep.
    // Per k-group a wave issues one 1-KiB weight load (HBM / Infinity Cache) and MT 1-KiB x-loads (L2); with DEPTH
    // groups in flight per wave and 8 waves per CU about DEPTH*8 KiB of weights are outstanding per CU, which is
    // what it takes to cover the ~2 us loaded-memory latency at ~30 GB/s per CU.  No load is conditional (indices
    // are clamped instead) so the compiler can retire them with counted s_waitcnt vmcnt(N).
    // K split evenly over the waves (static, so the summation order - and therefore every output bit - is reproducible).
    // Measured (tools/stamps.py): the four younger waves of a workgroup (the second wave on each SIMD) finish their slice
    // ~3 us after the four older ones; uneven static splits (36:28, 38:26) and s_setprio for the younger half did not
    // change the launch time - the CU's memory pipeline, not the split, sets when the last byte lands.
    const int per = (J.nkg + SK_WAVES - 1) / SK_WAVES;
    const int kg_begin = wave * per;
    const int kg_end = min(J.nkg, kg_begin + per);
    const int g0 = J.x[0].len >> 3, g1 = g0 + (J.x[1].len >> 3);  // k-group boundaries of the segments

    const float* xb0[MT]; const float* xb1[MT]; const float* xb2[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int b = mt * 32 + bl;
        const long bb = b < B ? b : 0;  // rows past B read row 0; their results are never stored
        xb0[mt] = J.x[0].p + bb * 8 + 4 * h;
        xb1[mt] = (J.x[1].p ? J.x[1].p : J.x[0].p) + bb * 8 + 4 * h - g0 * blk;
        xb2[mt] = (J.x[2].p ? J.x[2].p : J.x[0].p) + bb * 8 + 4 * h - g1 * blk;
    }
    // the job may cover only the k-groups [kg0, kg0 + nkg) of a matrix packed with nkg_w k-groups per tile (column slices
    // of the recurrent matrices: see the autoregressive step in gvx_api.hip)
    const int nkg_w = J.nkg_w > 0 ? J.nkg_w : J.nkg;
    const float4* wp = reinterpret_cast<const float4*>(J.Wp) + ((long)tile * nkg_w + J.kg0) * 64 + lane;

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[mt][q] = 0.f;

    if (kg_begin < kg_end) {
        float4 wv[DEPTH], xv[MT][DEPTH];
        const int g_last = kg_end - 1;
#define SK_LOAD(slot, gg)                                                                         \
        {                                                                                             \
            const int g_ = min((gg), g_last);                                                         \
            wv[slot] = wp[(long)g_ * 64];                                                            \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                       \
                const float* base_ = g_ < g0 ? xb0[mt] : (g_ < g1 ? xb1[mt] : xb2[mt]);               \
                xv[mt][slot] = *reinterpret_cast<const float4*>(base_ + (long)g_ * blk);              \
            }                                                                                         \
        }
#define SK_MFMA(slot)                                                                                  \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                               \
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[slot].x, xv[mt][slot].x, acc[mt], 0, 0, 0); \
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[slot].y, xv[mt][slot].y, acc[mt], 0, 0, 0); \
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[slot].z, xv[mt][slot].z, acc[mt], 0, 0, 0); \
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[slot].w, xv[mt][slot].w, acc[mt], 0, 0, 0); \
        }
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) SK_LOAD(u, kg_begin + u)
        int base = kg_begin;
        // steady state: every slot is valid and so is its refill -> branch-free body, counted waits
        for (; base + 2 * DEPTH <= kg_end; base += DEPTH) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                SK_MFMA(u)
                SK_LOAD(u, base + u + DEPTH)
                // keep the refill right behind the MFMAs that freed its registers: left alone, the scheduler sinks
                // all refills to the end of the pass and the wave drains to vmcnt(0) every DEPTH groups
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // drain.  The slots hold the next min(remaining, DEPTH) groups and remaining < 2*DEPTH.  Only when more than
        // DEPTH groups are left is there anything to refill: that pass is branch free (every slot valid, refills are
        // clamped loads; the few surplus ones re-read the last group and are never consumed).  The final pass issues
        // no loads and only guards the MFMAs (wave-uniform).
        int rem = kg_end - base;
        if (rem > DEPTH) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                SK_MFMA(u)
                SK_LOAD(u, base + u + DEPTH)
                __builtin_amdgcn_sched_barrier(0);
            }
            rem -= DEPTH;
        }
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            if (u < rem) SK_MFMA(u)
        }
#undef SK_MFMA
#undef SK_LOAD
    }

    GVX_STAMP(0, 1);
#ifdef GVX_STAMPS
    // per-wave end-of-main-loop times of one attention-LSTM tile (block 0) and one decoder-LSTM tile (block tiles0)
    if (lane == 0 && (blockIdx.x == 0 || (int)blockIdx.x == jobs.tiles0))
        gvx::gvx_stamps[blockIdx.x == 0 ? 1 : 2][8 + wave] = wall_clock64();
    if (threadIdx.x == 0 && (int)blockIdx.x == jobs.tiles0) gvx::gvx_stamps[2][7] = gvx::gvx_stamps[0][0];
#endif
    // ---- cross-wave K reduction through LDS
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < 16; ++q) red[((wave * MT + mt) * 16 + q) * 64 + lane] = acc[mt][q];
    __syncthreads();
    GVX_STAMP(0, 2);

    // unit u = (mt, g): register group g (4 registers) of batch tile mt; one unit per wave
    for (int u = wave; u < 4 * MT; u += SK_WAVES) {
        const int mt = u >> 2, g = u & 3;
        float s[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < SK_WAVES; ++w) t += red[((w * MT + mt) * 16 + 4 * g + qq) * 64 + lane];
            s[qq] = t;
        }
        const int b = mt * 32 + bl;
        const int nloc = 8 * g + 4 * h;          // first of the lane's 4 consecutive packed rows in the tile
        const int n = tile * 32 + nloc;
        if (J.mode == 0) {
            // LSTM cell: rows n..n+3 are gates i,f,g,o of hidden unit j = tile*8 + jloc
            const int jloc = 2 * g + h;
            const int j = tile * 8 + jloc;
            const int H = J.N >> 2;
            float hval = 0.f;
            if (b < B) {
                bool active = true;
                int tb = 0;
                if (J.seq_out) {
                    const int len = J.lengths ? J.lengths[b] : J.seq_len;
                    active = J.step < len;
                    tb = J.reverse ? (len - 1 - J.step) : J.step;
                }
                const long hoff = (long)tile * blk + b * 8 + jloc;  // blocked: k-group = tile, k & 7 = jloc
                if (active) {
                    float pre[4] = {s[0] + bias_pref.x, s[1] + bias_pref.y, s[2] + bias_pref.z, s[3] + bias_pref.w};
                    if (J.addend && J.seq_out) {   // encoder: the row's own time index picks the addend (decoder: prefetched above)
                        const float4 ad = *reinterpret_cast<const float4*>(J.addend + (long)b * J.add_bs + (long)tb * J.add_ts + n);
                        pre[0] += ad.x; pre[1] += ad.y; pre[2] += ad.z; pre[3] += ad.w;
                    }
                    const float c_old = c_pref;
                    const float c_new = sigmoidf_(pre[1]) * c_old + sigmoidf_(pre[0]) * tanhf_(pre[2]);
                    hval = sigmoidf_(pre[3]) * tanhf_(c_new);
                    J.c[(long)b * H + j] = c_new;
                    if (J.seq_out) J.seq_out[(long)b * J.seq_bs + (long)tb * J.seq_ts + j] = hval;
                } else {
                    hval = J.h_prev[hoff];
                }
                J.h_out[hoff] = hval;
            }
            if (J.q_slab) hs[b * 8 + jloc] = hval;
        } else if (J.mode == 2) {
            // partial pre-activations of a column slice: raw sums, batch-major [B][N] (the layout `addend` is read in)
            if (b < B) *reinterpret_cast<float4*>(J.y + (long)b * J.N + n) = make_float4(s[0], s[1], s[2], s[3]);
        } else {
            if (b < B) {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int nn = n + qq;
                    if (nn < J.N) {
                        float v = s[qq];
                        if (J.bias) v += J.bias[nn];
                        if (J.act == ACT_RELU) v = fmaxf(v, 0.f);
                        else if (J.act == ACT_TANH) v = tanhf(v);
                        if (J.keep) v = J.keep[(long)b * J.keep_stride + nn] ? 2.f * v : 0.f;
                        J.y[(long)(nn >> 3) * blk + b * 8 + (nn & 7)] = v;  // blocked output
                    }
                }
            }
        }
    }

    GVX_STAMP(0, 3);
    // ---- attention query partial products for this tile's 8 hidden units
    if (J.mode == 0 && J.q_slab) {
        __syncthreads();
        const int a = J.att_dim;
        const float* wq = J.Wq_t + (long)tile * a * 8;
        for (int idx = tid; idx < B * a; idx += SK_THREADS) {
            const int b = idx / a, d = idx - b * a;
            const float4 w0 = *reinterpret_cast<const float4*>(wq + d * 8);
            const float4 w1 = *reinterpret_cast<const float4*>(wq + d * 8 + 4);
            const float4 h0 = *reinterpret_cast<const float4*>(hs + b * 8);
            const float4 h1 = *reinterpret_cast<const float4*>(hs + b * 8 + 4);
            float v = w0.x * h0.x;
            v = fmaf(w0.y, h0.y, v); v = fmaf(w0.z, h0.z, v); v = fmaf(w0.w, h0.w, v);
            v = fmaf(w1.x, h1.x, v); v = fmaf(w1.y, h1.y, v); v = fmaf(w1.z, h1.z, v); v = fmaf(w1.w, h1.w, v);
            J.q_slab[((long)tile * B + b) * a + d] = v;
        }
    }
    GVX_STAMP(0, 4);
}

// Same body under three kernel names so that profiles separate the teacher-forced decoder step (the dominant kernel of
// the path) from the autoregressive step launches and the encoder recurrence.
#ifdef GVX_STAMPS
// diagnostic build: begin / end time of every workgroup of the last multi-job decoder launch (tools/stamps.py)
namespace { __device__ unsigned long long gvx_wg_span[2][512]; }
#define GVX_WG_BEGIN() do { if (threadIdx.x == 0 && jobs.njobs >= 2 && blockIdx.x < 512) gvx_wg_span[0][blockIdx.x] = wall_clock64(); } while (0)
#define GVX_WG_END() do { if (threadIdx.x == 0 && jobs.njobs >= 2 && blockIdx.x < 512) gvx_wg_span[1][blockIdx.x] = wall_clock64(); } while (0)
#else
#define GVX_WG_BEGIN() do { } while (0)
#define GVX_WG_END() do { } while (0)
#endif
template <int MT> __global__ __launch_bounds__(SK_THREADS) void decoder_lstm_step_kernel(SkinnyJobs jobs) {
    GVX_WG_BEGIN();
    if ((int)blockIdx.x >= jobs.tiles) { loc_body(jobs.loc, (int)blockIdx.x - jobs.tiles); GVX_WG_END(); return; }   // uniform per workgroup
    skinny_body<MT, (MT == 1 ? SK_DEPTH1 : SK_DEPTH2)>(jobs);
    GVX_WG_END();
}
template <int MT> __global__ __launch_bounds__(SK_THREADS) void ar_lstm_step_kernel(SkinnyJobs jobs) {   // autoregressive launches A / C
    if ((int)blockIdx.x >= jobs.tiles) { loc_body(jobs.loc, (int)blockIdx.x - jobs.tiles); return; }
    skinny_body<MT, (MT == 1 ? SK_DEPTH1 : SK_DEPTH2)>(jobs);
}
template <int MT> __global__ __launch_bounds__(SK_THREADS) void encoder_lstm_step_kernel(SkinnyJobs jobs) { skinny_body<MT, (MT == 1 ? SK_DEPTH1 : SK_DEPTH2)>(jobs); }

static size_t skinny_lds(int MT) { return (size_t)(SK_WAVES * MT * 16 * 64 + MT * 32 * 8) * sizeof(float); }

template <typename K>
static hipError_t set_lds(K kern, int MT) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)skinny_lds(MT));
}

hipError_t skinny_init() {
    hipError_t e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_lstm_step_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_lstm_step_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(ar_lstm_step_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(ar_lstm_step_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = set_lds(encoder_lstm_step_kernel<1>, 1)) != hipSuccess) return e;
    return set_lds(encoder_lstm_step_kernel<2>, 2);
}

hipError_t launch_skinny(const SkinnyJob* jobs, int njobs, int kind, hipStream_t s, const LocJob* loc) {
    if (njobs < 1 || njobs > 3) return hipErrorInvalidValue;
    SkinnyJobs js;
    js.njobs = njobs;
    for (int i = 0; i < 3; ++i) js.job[i] = jobs[i < njobs ? i : njobs - 1];
    js.tiles0 = (jobs[0].N + 31) / 32;
    js.tiles1 = njobs > 1 ? (jobs[1].N + 31) / 32 : 0;
    js.tiles = js.tiles0 + js.tiles1 + (njobs > 2 ? (jobs[2].N + 31) / 32 : 0);
    const int B = jobs[0].B;
    for (int i = 1; i < njobs; ++i)
        if (jobs[i].B != B) return hipErrorInvalidValue;
    for (int i = 0; i < njobs; ++i)
        if (jobs[i].mode == 2 && (jobs[i].N & 31)) return hipErrorInvalidValue;   // partial tiles store whole float4 rows
    if (B < 1 || B > 64) return hipErrorInvalidValue;
    int extra = 0;
    js.loc = LocJob{};
    if (loc && loc->G > 0) {
        if (kind == SK_ENCODER) return hipErrorInvalidValue;
        js.loc = *loc;
        extra = loc->B * loc->G;
    }
    const int MT = B > 32 ? 2 : 1;
    size_t lds = skinny_lds(MT);
    if (extra) {
        const size_t need = (size_t)loc_lds_floats(loc->L, loc->G, loc->kl) * sizeof(float);
        if (need > lds) lds = need;
        if (lds > 160 * 1024) return hipErrorInvalidValue;
    }
    const dim3 grid(js.tiles + extra), block(SK_THREADS);
    if (MT == 2) {
        if (kind == SK_DECODER) decoder_lstm_step_kernel<2><<<grid, block, lds, s>>>(js);
        else if (kind == SK_ENCODER) encoder_lstm_step_kernel<2><<<grid, block, lds, s>>>(js);
        else ar_lstm_step_kernel<2><<<grid, block, lds, s>>>(js);
    } else {
        if (kind == SK_DECODER) decoder_lstm_step_kernel<1><<<grid, block, lds, s>>>(js);
        else if (kind == SK_ENCODER) encoder_lstm_step_kernel<1><<<grid, block, lds, s>>>(js);
        else ar_lstm_step_kernel<1><<<grid, block, lds, s>>>(js);
    }
    return hipGetLastError();
}

#ifdef GVX_STAMPS
hipError_t read_stamps_skinny(unsigned long long* host96) {
    return hipMemcpyFromSymbol(host96, HIP_SYMBOL(gvx_stamps), sizeof(unsigned long long) * 96);
}
hipError_t read_wg_spans(unsigned long long* host1024) {
    return hipMemcpyFromSymbol(host1024, HIP_SYMBOL(gvx_wg_span), sizeof(unsigned long long) * 1024);
}
#endif

}  // namespace gvx
