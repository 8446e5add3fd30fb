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
#include "attn_step_body.h"

#include <cstdlib>

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
    SkinnyJob job[4];
    int njobs;
    int tiles0;   // tiles of job 0
    int tiles1;   // tiles of job 1
    int tiles2;   // tiles of job 2 (only read when njobs == 4)
    int tiles;    // tiles of all jobs; blocks >= tiles are location-feature workgroups
    LocJob loc;
    int pa_layout;   // teacher-forced step beside the persistent attention kernel: 224 (96) workgroups, see skinny_body
    int block0 = 0;  // first block of the tiles (launch shared with attention workgroups: ar_attn_tiles_kernel); 0 otherwise
    int rot;         // > 0: every workgroup walks its waves' k-group slices from a start rotated by (tile * rot) - thousands of
                     // waves otherwise read the same 1-KiB x fragments (L2 lines) at the same time
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

// XH: the workgroup may carry an extra HALF tile (16 of another tile's 32 packed rows) through the same pass over x.
// Layout of the teacher-forced launch beside the persistent attention kernel, which holds 32 CUs: a CU streams ~25 KB/us
// whatever shares it, so the launch must have one workgroup per remaining CU and equal bytes per workgroup (measured with
// tools/micro/partition_bench.hip: 256 tiles on 224 CUs 26 us, this deal 18.0 us, 256 tiles on 256 CUs 17.5 us):
//   blocks [0, 64):   attention-LSTM tile + half of a neighbour  (48 packed rows, 336 KB of weights, one pass over x)
//   blocks [64, 96):  attention-LSTM tiles 96 .. 127             (224 KB)
//   blocks [96, 224): decoder-LSTM tiles                         (320 KB)
// block 2m carries tile 3m and rows 0..15 of tile 3m + 1, block 2m + 1 tile 3m + 2 and rows 16..31 of tile 3m + 1.
//
// RT = 2: the workgroup carries TWO consecutive row tiles (tile, tile + 1: 64 packed rows) through one pass over x - two
// weight fragments per x fragment.  Layout of the teacher-forced launch beside a resident attention kernel that holds 64
// CUs (two workgroups per batch row, 128 < L <= 256: attn_persist.hip), pa_layout 2, 192 workgroups:
//   blocks [0, 64):   attention-LSTM tiles 2 m, 2 m + 1   (448 KB of weights + 224 KB of x)
//   blocks [64, 192): decoder-LSTM tiles                   (320 KB of weights + 320 KB of x)
// - about equal bytes through every CU's load path.  One batch tile only (MT = 1), no half tiles.
// ARX: the autoregressive loop's extras - a second addend for the cells, addends on partial-sum jobs, tiles that start at
// block `block0` of a launch shared with attention workgroups.  A template parameter so that the teacher-forced kernels keep
// the instruction stream they were tuned with: as run-time branches these moved the dominant launch by +0.25 us (the kernel's
// time follows the compiler's schedule, not its instruction count - compiling the training / encoder branches OUT of the
// inference kernel made it 0.2-0.3 us slower, measured the same way; A/B of whole libraries on one box, GVX_LIB).
template <int MT, int DEPTH, bool XH = false, bool DEFER = false, int RT = 1, bool ARX = false>
__device__ __forceinline__ void skinny_body(const SkinnyJobs& jobs) {
    static_assert(RT == 1 || (RT == 2 && MT == 1 && !XH), "two row tiles: one batch tile, no half tile");
    constexpr int NT = MT * RT;   // accumulator tiles per wave; t = rt (RT = 2) or mt (MT = 2)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* red = smem;                                   // [SK_WAVES][NT][16][64]
    float* hs = smem + SK_WAVES * NT * 16 * 64;          // [MT*32][RT*8] h' of this workgroup's hidden units (LSTM + q slabs)
    float* red2 = hs + NT * 32 * 8;                      // XH: [SK_WAVES][8][64] of the extra half tile, then its h' [32][4]

    int jsel = 0, tile = (int)blockIdx.x - (ARX ? jobs.block0 : 0);
    int xt = -1, xhalf = 0;    // extra half tile: packed rows 16 xhalf .. 16 xhalf + 15 of tile xt
    if (jobs.pa_layout == 2) {
        const int bid = (int)blockIdx.x;
        if (RT == 2) tile = 2 * bid;
        else { jsel = 1; tile = bid - 64; }
    } else if (XH && jobs.pa_layout) {
        const int bid = (int)blockIdx.x;
        if (bid < 64) { const int m = bid >> 1, odd = bid & 1; tile = 3 * m + 2 * odd; xt = 3 * m + 1; xhalf = odd; }
        else if (bid < 96) tile = 96 + (bid - 64);
        else { jsel = 1; tile = bid - 96; }
    } else if (jobs.njobs > 1 && tile >= jobs.tiles0) {
        jsel = 1; tile -= jobs.tiles0;
        if (jobs.njobs > 2 && tile >= jobs.tiles1) {
            jsel = 2; tile -= jobs.tiles1;
            if (jobs.njobs > 3 && tile >= jobs.tiles2) { jsel = 3; tile -= jobs.tiles2; }
        }
    }
    const bool has_x = XH && xt >= 0;   // workgroup-uniform
    const SkinnyJob& J = jobs.job[jsel];
    if (DEFER && J.start_cnt && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_fetch_add(J.start_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (DEFER && J.ctx_cnt) {   // workgroup-local "context has arrived" word (see the deferred segment below)
        if (threadIdx.x == 0) reinterpret_cast<volatile int*>(hs)[NT * 32 * 8 - 1] = 0;
        __syncthreads();
    }
    if (jobs.njobs >= 2) GVX_STAMP(0, 0);   // (stamps build: the single-job drain launch must not overwrite a step's stamps)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar branches, counted waits
    const int bl = lane & 31, h = lane >> 5;
    const int B = J.B;
    const long blk = (long)B * 8;  // floats per k-group of a blocked vector

    // epilogue operands of this wave's unit (cell state, bias): fetched now so their latency hides under the main loop
    float c_pref = 0.f;
    float4 bias_pref = make_float4(0.f, 0.f, 0.f, 0.f);
    if (J.mode == 0 && wave < 4 * NT) {
        const int t_ = wave >> 2, g_ = wave & 3;
        const int mt_ = RT > 1 ? 0 : t_, tile_ = tile + (RT > 1 ? t_ : 0);
        const int b_ = mt_ * 32 + bl, j_ = tile_ * 8 + 2 * g_ + h;
        if (b_ < B) c_pref = J.c[(long)b_ * (J.N >> 2) + j_];
        if (J.bias) bias_pref = *reinterpret_cast<const float4*>(J.bias + tile_ * 32 + 8 * g_ + 4 * h);
        // decoder cells finished from partial sums (autoregressive launches): the addend is known at launch, fetch it now
        if (J.addend && !J.seq_out && b_ < B) {
            const float4 ad = *reinterpret_cast<const float4*>(J.addend + (long)b_ * J.add_bs + tile_ * 32 + 8 * g_ + 4 * h);
            bias_pref.x += ad.x; bias_pref.y += ad.y; bias_pref.z += ad.z; bias_pref.w += ad.w;
            if (ARX && J.addend2) {
                const float4 a2 = *reinterpret_cast<const float4*>(J.addend2 + (long)b_ * J.add_bs + tile_ * 32 + 8 * g_ + 4 * h);
                bias_pref.x += a2.x; bias_pref.y += a2.y; bias_pref.z += a2.z; bias_pref.w += a2.w;
            }
        }
    }
    // partial-sum jobs that continue somebody else's sums: the addends of the lane's four rows, fetched now as well
    if (ARX && J.mode == 2 && J.addend && wave < 4 * NT) {
        const int t_ = wave >> 2, g_ = wave & 3;
        const int mt_ = RT > 1 ? 0 : t_, tile_ = tile + (RT > 1 ? t_ : 0);
        const int b_ = mt_ * 32 + bl;
        if (b_ < B) {
            bias_pref = *reinterpret_cast<const float4*>(J.addend + (long)b_ * J.add_bs + tile_ * 32 + 8 * g_ + 4 * h);
            if (J.addend2) {
                const float4 a2 = *reinterpret_cast<const float4*>(J.addend2 + (long)b_ * J.add_bs + tile_ * 32 + 8 * g_ + 4 * h);
                bias_pref.x += a2.x; bias_pref.y += a2.y; bias_pref.z += a2.z; bias_pref.w += a2.w;
            }
        }
    }

    if (XH && has_x && J.mode == 0 && (wave == 4 || wave == 5)) {
        // extra half tile: lane (j = lane & 15, g = lane >> 4) finishes hidden unit 4 xhalf + g of tile xt for batch row
        // j (wave 4) or 16 + j (wave 5)
        const int b_ = (lane & 15) + 16 * (wave - 4), jl_ = 4 * xhalf + (lane >> 4);
        if (b_ < B) c_pref = J.c[(long)b_ * (J.N >> 2) + xt * 8 + jl_];
        if (J.bias) bias_pref = *reinterpret_cast<const float4*>(J.bias + xt * 32 + 4 * jl_);
        if (J.addend && b_ < B) {
            const float4 ad = *reinterpret_cast<const float4*>(J.addend + (long)b_ * J.add_bs + xt * 32 + 4 * jl_);
            bias_pref.x += ad.x; bias_pref.y += ad.y; bias_pref.z += ad.z; bias_pref.w += ad.w;
        }
    }

    // Query-slab weights of this wave's 32 attention dims (MFMA form of the slab phase, one batch tile only): fetched now
    // as well - loaded after the cells they were a dependent global round trip at the end of the launch's slowest workgroups
    const int a_tiles = (J.att_dim + 31) >> 5;   // 32-wide slices of the slab's columns, one per wave
#ifdef SK_NO_MFMA_SLAB   // A/B builds only
    const bool slab_mfma = false;
#else
    const bool slab_mfma = MT == 1 && J.mode == 0 && J.q_slab && a_tiles <= SK_WAVES;
#endif
    float4 wq_a = make_float4(0.f, 0.f, 0.f, 0.f), wq_b = wq_a, wq_c = wq_a, wq_x = wq_a;
    const bool slab_wave = slab_mfma && wave < a_tiles;
    if (slab_wave && 32 * wave + bl < J.att_dim) {   // (columns past att_dim keep zero weights and are never stored)
        const float* wq_l = J.Wq_t + ((long)tile * J.att_dim + 32 * wave + bl) * 8;
        wq_a = *reinterpret_cast<const float4*>(wq_l);
        wq_b = *reinterpret_cast<const float4*>(wq_l + 4);
        if (XH && has_x) wq_c = *reinterpret_cast<const float4*>(J.Wq_t + ((long)xt * J.att_dim + 32 * wave + bl) * 8 + 4 * xhalf);
        if (RT == 2) {   // second row tile: its 8 hidden units follow the first tile's in the slab's sum
            const float* wq_l2 = J.Wq_t + ((long)(tile + 1) * J.att_dim + 32 * wave + bl) * 8;
            wq_c = *reinterpret_cast<const float4*>(wq_l2);
            wq_x = *reinterpret_cast<const float4*>(wq_l2 + 4);
        }
        if (RT == 1 && J.xw) wq_x = *reinterpret_cast<const float4*>(J.xw + ((long)tile * J.att_dim + 32 * wave + bl) * 4);
    }
    // extra slab terms (J.xw): the lane's two values xsrc[b][4 tile + kh], xsrc[b][4 tile + 2 + kh].  Read with sc1 loads - the
    // vector may have been published inside this launch by another kernel -, at the start of the launch or, when it is the
    // deferred segment, right after its arrival (their latency then hides under the deferred k-groups)
    float xe_a = 0.f, xe_b = 0.f;
    auto load_extra = [&]() {
        if (slab_wave && J.xw) {
            const __amdgpu_buffer_rsrc_t rxs = make_rsrc(J.xsrc);
            const unsigned k0 = 4u * (unsigned)tile;
            const unsigned off = ((k0 >> 3) * (unsigned)blk + (unsigned)(bl < B ? bl : 0) * 8u + (k0 & 7u) + (unsigned)h) * 4u;
            xe_a = load_sc1_f32(rxs, off);
            xe_b = load_sc1_f32(rxs, off + 8u);
        }
    };
    if (!(DEFER && J.defer_seg && J.ctx_cnt)) load_extra();

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
    const float4* wp_b = wp + (long)nkg_w * 64;   // RT = 2: the same k-groups of tile + 1
    // extra half tile: the lanes of the other half read their partner's address (same bytes, no extra traffic): the copies
    // feed the second batch-row block of the 16x16x1 MFMAs below
    const bool x_mine = ((lane >> 4) & 1) == xhalf;
    const float4* wp2 = reinterpret_cast<const float4*>(J.Wp) + ((long)(has_x ? xt : tile) * nkg_w + J.kg0) * 64 + (x_mine ? lane : (lane ^ 16));

    f32x16 acc[NT];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[mt][q] = 0.f;
    f32x16 acc2;   // XH only
#pragma unroll
    for (int q = 0; q < 16; ++q) acc2[q] = 0.f;

    // Streams the k-groups MAP(s), s in [S_BEGIN, S_END), of this wave through the MFMAs, DEPTH groups in flight.
    // XLOAD(mt, kg) loads the x fragment of k-group kg.
#define SK_STREAM(S_BEGIN, S_END, MAP, XLOAD)                                                            \
    if ((S_BEGIN) < (S_END)) {                                                                            \
        float4 wv[DEPTH], xv[MT][DEPTH], wv2[(XH || RT == 2) ? DEPTH : 1];                                \
        const int s_last = (S_END) - 1;                                                                   \
        _Pragma("unroll") for (int u = 0; u < DEPTH; ++u) SK_LOAD(u, (S_BEGIN) + u, MAP, XLOAD)           \
        int base = (S_BEGIN);                                                                             \
        /* steady state: every slot is valid and so is its refill -> branch-free body, counted waits */   \
        for (; base + 2 * DEPTH <= (S_END); base += DEPTH) {                                              \
            _Pragma("unroll") for (int u = 0; u < DEPTH; ++u) {                                           \
                SK_MFMA(u)                                                                                \
                SK_LOAD(u, base + u + DEPTH, MAP, XLOAD)                                                  \
                /* keep the refill right behind the MFMAs that freed its registers: left alone, the   */  \
                /* scheduler sinks all refills to the end of the pass and the wave drains to vmcnt(0) */  \
                __builtin_amdgcn_sched_barrier(0);                                                        \
            }                                                                                             \
        }                                                                                                 \
        /* drain.  The slots hold the next min(remaining, DEPTH) groups and remaining < 2*DEPTH.  Only */ \
        /* when more than DEPTH groups are left is there anything to refill: that pass is branch free  */ \
        /* (every slot valid, refills are clamped loads; the few surplus ones re-read the last group   */ \
        /* and are never consumed).  The final pass issues no loads and only guards the MFMAs.         */ \
        int rem = (S_END) - base;                                                                         \
        if (rem > DEPTH) {                                                                                \
            _Pragma("unroll") for (int u = 0; u < DEPTH; ++u) {                                           \
                SK_MFMA(u)                                                                                \
                SK_LOAD(u, base + u + DEPTH, MAP, XLOAD)                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                        \
            }                                                                                             \
            rem -= DEPTH;                                                                                 \
        }                                                                                                 \
        _Pragma("unroll") for (int u = 0; u < DEPTH; ++u) {                                               \
            if (u < rem) SK_MFMA(u)                                                                       \
        }                                                                                                 \
    }
#define SK_LOAD(slot, ss, MAP, XLOAD)                                                             \
        {                                                                                             \
            const int s_ = min((ss), s_last);                                                         \
            const int g_ = MAP(s_);                                                                   \
            wv[slot] = wp[(long)g_ * 64];                                                            \
            if (RT == 2) wv2[slot] = wp_b[(long)g_ * 64];                                            \
            if (XH && has_x) wv2[slot] = wp2[(long)g_ * 64];                                         \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) xv[mt][slot] = XLOAD(mt, g_);           \
        }
/* ONE statement (braces): the drain pass guards it with `if (u < rem)`.  As two statements the guard covered only the    */
/* 32x32 loop and the half tile's MFMAs ran on stale slots whenever rem < DEPTH - never at depth 4 with the K slices of */
/* the default layer sizes (16 and 8 k-groups per wave), always at depth 5 / 6: round 2's "depth 5 / 6 gives wrong      */
/* results in the 48-row workgroups" (tests/test_parity_gpu.py::test_persistent_attention_depth6)   */
#define SK_MFMA(slot) {                                                                                \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                               \
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[slot].x, xv[mt][slot].x, acc[mt], 0, 0, 0); \
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[slot].y, xv[mt][slot].y, acc[mt], 0, 0, 0); \
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[slot].z, xv[mt][slot].z, acc[mt], 0, 0, 0); \
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[slot].w, xv[mt][slot].w, acc[mt], 0, 0, 0); \
        }                                                                                                 \
        if (RT == 2) {                                                                                    \
            acc[NT - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv2[slot].x, xv[0][slot].x, acc[NT - 1], 0, 0, 0); \
            acc[NT - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv2[slot].y, xv[0][slot].y, acc[NT - 1], 0, 0, 0); \
            acc[NT - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv2[slot].z, xv[0][slot].z, acc[NT - 1], 0, 0, 0); \
            acc[NT - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv2[slot].w, xv[0][slot].w, acc[NT - 1], 0, 0, 0); \
        }                                                                                                 \
        if (XH && has_x) {                                                                                \
            /* extra half tile on v_mfma_f32_16x16x1_4b_f32: four independent 16 x 16 blocks, block = lane >> 4.  With */ \
            /* the packed fragment in the half's lanes AND copied into their partners (lane ^ 16), block 0 is         */ \
            /* (rows 0-15 of the half) x (batch rows 0-15) at k = c, block 1 the same rows x batch rows 16-31,        */ \
            /* blocks 2 / 3 the same at k = 4 + c: no lane is wasted (the 32x32x2 form with half its rows zeroed      */ \
            /* made these workgroups MFMA bound: 26-28 us per launch)                                                 */ \
            const float4 w2_ = wv2[slot];                                                                 \
            acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(w2_.x, xv[0][slot].x, acc2, 0, 0, 0);             \
            acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(w2_.y, xv[0][slot].y, acc2, 0, 0, 0);             \
            acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(w2_.z, xv[0][slot].z, acc2, 0, 0, 0);             \
            acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(w2_.w, xv[0][slot].w, acc2, 0, 0, 0);             \
        } }
#define SK_X_PLAIN(mt, g) (*reinterpret_cast<const float4*>(((g) < g0 ? xb0[mt] : ((g) < g1 ? xb1[mt] : xb2[mt])) + (long)(g) * blk))
    // rotated walk over a slice [b_, e_): stream index s -> b_ + (s - b_ + r_) mod (e_ - b_)   (scalar arithmetic)
    const int rot_t = jobs.rot > 0 ? tile * jobs.rot : 0;
#define SK_ROT(s, b_, e_, r_) ((s) + (r_) < (e_) ? (s) + (r_) : (s) + (r_) - ((e_) - (b_)))
    const int rot_id = kg_end > kg_begin ? rot_t % (kg_end - kg_begin) : 0;
#define SK_MAP_ID(s) SK_ROT(s, kg_begin, kg_end, rot_id)
    if (!DEFER || !J.defer_seg) {   // (DEFER is a template parameter so that the ordinary kernels do not carry the code below)
        SK_STREAM(kg_begin, kg_end, SK_MAP_ID, SK_X_PLAIN)
    } else {
        // Deferred segment (teacher-forced decoder: x[1] is the attention context of the previous step, which the persistent
        // attention kernel publishes while this launch is already streaming): every wave first takes its share of the
        // k-groups of x[0] and x[2], then - once the context counter has reached its target - its share of x[1]'s.
        const int nc = g1 - g0, nn = J.nkg - nc;
        const int per_n = (nn + SK_WAVES - 1) / SK_WAVES, per_c = (nc + SK_WAVES - 1) / SK_WAVES;
        const int n_begin = min(nn, wave * per_n), n_end = min(nn, n_begin + per_n);
        const int c_begin = min(nc, wave * per_c), c_end = min(nc, c_begin + per_c);
        const int rot_n = n_end > n_begin ? rot_t % (n_end - n_begin) : 0, rot_c = c_end > c_begin ? rot_t % (c_end - c_begin) : 0;
#define SK_MAP_N0(s) ((s) < g0 ? (s) : (s) + nc)
#define SK_MAP_N(s) SK_MAP_N0(SK_ROT(s, n_begin, n_end, rot_n))
#define SK_MAP_C(s) (SK_ROT(s, c_begin, c_end, rot_c) + g0)
        SK_STREAM(n_begin, n_end, SK_MAP_N, SK_X_PLAIN)
        if (J.ctx_cnt) {
            // ONE wave per workgroup polls the counter (1 800 waves polling one word queue in front of the producer's own
            // add to it); the others watch a word in LDS that the polling wave sets.  The context is then read with sc1
            // loads only: the bytes were stored write-through by another kernel, a plain load could hit a stale L1 / L2 line
            volatile int* seen = reinterpret_cast<volatile int*>(hs) + NT * 32 * 8 - 1;   // last word of hs (free until the epilogue)
            if (wave == 0) {
                handoff_wait(J.ctx_cnt, J.ctx_target, J.tmo, 0x200u, J.spin_limit);
                *seen = 1;
            } else {
                // no bound of its own: wave 0 sets the word after ITS bounded wait in every case (with a bound here a wave
                // could give up before wave 0 does and read a context that has not arrived, unreported)
                while (*seen == 0) __builtin_amdgcn_s_sleep(4);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no instruction: keeps the compiler from moving the loads above the poll
            load_extra();
            const __amdgpu_buffer_rsrc_t rx = make_rsrc(J.x[1].p);
            unsigned xo[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int b = mt * 32 + bl;
                xo[mt] = (unsigned)((b < B ? b : 0) * 8 + 4 * h) * 4u;
            }
#define SK_X_SC1(mt, g) load_sc1(rx, xo[mt] + (unsigned)((g) - g0) * (unsigned)(blk * 4))
            SK_STREAM(c_begin, c_end, SK_MAP_C, SK_X_SC1)
#undef SK_X_SC1
        } else {
            SK_STREAM(c_begin, c_end, SK_MAP_C, SK_X_PLAIN)
        }
#undef SK_MAP_N
#undef SK_MAP_N0
#undef SK_MAP_C
    }
#undef SK_MAP_ID
#undef SK_ROT
#undef SK_X_PLAIN
#undef SK_MFMA
#undef SK_LOAD
#undef SK_STREAM

    if (jobs.njobs >= 2) GVX_STAMP(0, 1);   // (stamps build: the single-job drain launch must not overwrite a step's stamps)
#ifdef GVX_STAMPS
    // per-wave end-of-main-loop times of one attention-LSTM tile (block 0) and one decoder-LSTM tile (block tiles0)
    if (lane == 0 && (blockIdx.x == 0 || (int)blockIdx.x == jobs.tiles0))
        gvx::gvx_stamps[blockIdx.x == 0 ? 1 : 2][8 + wave] = wall_clock64();
    if (threadIdx.x == 0 && (int)blockIdx.x == jobs.tiles0) gvx::gvx_stamps[2][7] = gvx::gvx_stamps[0][0];
#endif
    // ---- cross-wave K reduction through LDS
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int q = 0; q < 16; ++q) red[((wave * NT + mt) * 16 + q) * 64 + lane] = acc[mt][q];
    if (XH && has_x) {   // the two k-half blocks of a batch-row block are added here, before the cross-wave sum
#pragma unroll
        for (int q = 0; q < 8; ++q) red2[(wave * 8 + q) * 64 + lane] = acc2[q] + acc2[8 + q];
    }
    __syncthreads();
    if (jobs.njobs >= 2) GVX_STAMP(0, 2);   // (stamps build: the single-job drain launch must not overwrite a step's stamps)
    float* hs2 = red2 + SK_WAVES * 8 * 64;    // [32][4] h' of the extra half tile's 4 hidden units
    if (XH && has_x && J.mode == 0 && (wave == 4 || wave == 5)) {
        // LSTM cell of the extra half tile's units (decoder semantics only: no packed sequences, no addend).  Accumulator
        // 4 blk + r of lane (j, g) is gate r of unit g for batch row j (blocks 0, 2: k halves) or 16 + j (blocks 1, 3)
        const int rb = wave - 4;
        float s2[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < SK_WAVES; ++w) t += red2[(w * 8 + 4 * rb + qq) * 64 + lane];
            s2[qq] = t;
        }
        const int b2 = (lane & 15) + 16 * rb, g2 = lane >> 4;
        const int jloc = 4 * xhalf + g2, j = xt * 8 + jloc, H = J.N >> 2;
        float hval = 0.f;
        if (b2 < B) {
            const float p0 = s2[0] + bias_pref.x, p1 = s2[1] + bias_pref.y, p2 = s2[2] + bias_pref.z, p3 = s2[3] + bias_pref.w;
            if (J.pre_out) *reinterpret_cast<float4*>(J.pre_out + ((long)b2 * H + j) * 4) = make_float4(p0, p1, p2, p3);   // (training tape)
            const float c_new = sigmoidf_(p1) * c_pref + sigmoidf_(p0) * tanhf_(p2);
            hval = sigmoidf_(p3) * tanhf_(c_new);
            if (J.h_keep) hval = J.h_keep[(long)b2 * H + j] ? hval * J.h_scale : 0.f;
            (J.c_out ? J.c_out : J.c)[(long)b2 * H + j] = c_new;
            J.h_out[(long)xt * blk + b2 * 8 + jloc] = hval;
        }
        hs2[b2 * 4 + g2] = hval;
    }

    // unit u = (t, g): register group g (4 registers) of accumulator tile t (batch tile mt or row tile rt); one unit per wave
    const int tile0 = tile;
    for (int u = wave; u < 4 * NT; u += SK_WAVES) {
        const int tt = u >> 2, g = u & 3;
        const int mt = RT > 1 ? 0 : tt, rt = RT > 1 ? tt : 0;
        const int tile = tile0 + rt;   // (shadows the workgroup's first tile inside the unit)
        float s[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < SK_WAVES; ++w) t += red[((w * NT + tt) * 16 + 4 * g + qq) * 64 + lane];
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
                    if (J.pre_out) *reinterpret_cast<float4*>(J.pre_out + ((long)b * H + j) * 4) = make_float4(pre[0], pre[1], pre[2], pre[3]);
                    const float c_old = c_pref;
                    const float c_new = sigmoidf_(pre[1]) * c_old + sigmoidf_(pre[0]) * tanhf_(pre[2]);
                    hval = sigmoidf_(pre[3]) * tanhf_(c_new);
                    if (J.h_keep) hval = J.h_keep[(long)b * H + j] ? hval * J.h_scale : 0.f;
                    (J.c_out ? J.c_out : J.c)[(long)b * H + j] = c_new;
                    if (J.seq_out) J.seq_out[(long)b * J.seq_bs + (long)tb * J.seq_ts + j] = hval;
                    if (J.c_seq_out) J.c_seq_out[(long)b * J.seq_bs + (long)tb * J.seq_ts + j] = c_new;
                } else {
                    hval = J.h_prev[hoff];
                }
                J.h_out[hoff] = hval;
            }
            if (J.q_slab) hs[(b * RT + rt) * 8 + jloc] = hval;
        } else if (J.mode == 2) {
            // partial pre-activations of a column slice: raw sums, batch-major [B][N] (the layout `addend` is read in)
            if (b < B) *reinterpret_cast<float4*>(J.y + (long)b * J.N + n) =
                ARX ? make_float4(s[0] + bias_pref.x, s[1] + bias_pref.y, s[2] + bias_pref.z, s[3] + bias_pref.w)   // (zeros without addends)
                    : make_float4(s[0], s[1], s[2], s[3]);
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

    if (jobs.njobs >= 2) GVX_STAMP(0, 3);   // (stamps build: the single-job drain launch must not overwrite a step's stamps)
    // ---- attention query partial products for this workgroup's 8 (12) hidden units
    if (J.mode == 0 && J.q_slab) {
        __syncthreads();
        const int a = J.att_dim;
        const float* wq = J.Wq_t + (long)tile * a * 8;
        const float* wq2 = J.Wq_t + (long)(has_x ? xt : tile) * a * 8 + 4 * xhalf;
        const int slab = jobs.pa_layout ? (int)blockIdx.x : tile;   // one slab per workgroup
        if (slab_mfma) {
            // slab[b][d] = sum_j h'[b][j] Wq[d][j] over the workgroup's 8 (12) hidden units on the matrix pipe: wave w takes
            // the attention dims 32 w .. 32 w + 31.  v_mfma_f32_32x32x2_f32 with A = h' (lane (b, kh): h'[b][2 s + kh]) and
            // B = Wq^T (lane (d, kh): Wq[d][2 s + kh]); lane (d, hh) receives D[b = 8 g + 4 hh + r][d] in register 4 g + r,
            // so every store instruction writes two rows of 32 consecutive floats.  The sum runs over j in index order,
            // the chain the VALU form used (which took ~2 us of dependent FMAs and LDS reads here).
            if (slab_wave) {
                f32x16 qa;
#pragma unroll
                for (int q = 0; q < 16; ++q) qa[q] = 0.f;
                const float* hrow = hs + bl * RT * 8 + h;
                qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow[0], h ? wq_a.y : wq_a.x, qa, 0, 0, 0);
                qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow[2], h ? wq_a.w : wq_a.z, qa, 0, 0, 0);
                qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow[4], h ? wq_b.y : wq_b.x, qa, 0, 0, 0);
                qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow[6], h ? wq_b.w : wq_b.z, qa, 0, 0, 0);
                if (XH && has_x) {
                    const float* hrow2 = hs2 + bl * 4 + h;
                    qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow2[0], h ? wq_c.y : wq_c.x, qa, 0, 0, 0);
                    qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow2[2], h ? wq_c.w : wq_c.z, qa, 0, 0, 0);
                }
                if (RT == 2) {   // hidden units of tile + 1
                    qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow[8], h ? wq_c.y : wq_c.x, qa, 0, 0, 0);
                    qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow[10], h ? wq_c.w : wq_c.z, qa, 0, 0, 0);
                    qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow[12], h ? wq_x.y : wq_x.x, qa, 0, 0, 0);
                    qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow[14], h ? wq_x.w : wq_x.z, qa, 0, 0, 0);
                }
                if (RT == 1 && J.xw) {
                    qa = __builtin_amdgcn_mfma_f32_32x32x2f32(xe_a, h ? wq_x.y : wq_x.x, qa, 0, 0, 0);
                    qa = __builtin_amdgcn_mfma_f32_32x32x2f32(xe_b, h ? wq_x.w : wq_x.z, qa, 0, 0, 0);
                }
                if (32 * wave + bl < a) {
                    float* qout = J.q_slab + (long)slab * B * a + 32 * wave + bl;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int b = 8 * g + 4 * h + r;
                            if (b < B) qout[(long)b * a] = qa[4 * g + r];
                        }
                }
            }
            if (J.ready_cnt && tid == 0) handoff_wait(J.ready_cnt, J.ready_target, J.tmo, 0x300u, J.spin_limit);
            if (jobs.njobs >= 2) GVX_STAMP(0, 4);
            return;
        }
        // a thread keeps its attention dim d over the passes when the thread count is a multiple of a: the weights are loaded
        // once (they were re-loaded per value, one dependent round trip each: 3 us in the 12-unit workgroups)
        const bool fixed_d = (SK_THREADS % a) == 0;
        const int d_f = tid % a;
        float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0, w2 = w0;
        if (fixed_d) {
            w0 = *reinterpret_cast<const float4*>(wq + d_f * 8);
            w1 = *reinterpret_cast<const float4*>(wq + d_f * 8 + 4);
            if (XH && has_x) w2 = *reinterpret_cast<const float4*>(wq2 + d_f * 8);
        }
        auto slab_row = [&](int b, int d) {
            const float4 h0 = *reinterpret_cast<const float4*>(hs + b * 8);
            const float4 h1 = *reinterpret_cast<const float4*>(hs + b * 8 + 4);
            float v = w0.x * h0.x;
            v = fmaf(w0.y, h0.y, v); v = fmaf(w0.z, h0.z, v); v = fmaf(w0.w, h0.w, v);
            v = fmaf(w1.x, h1.x, v); v = fmaf(w1.y, h1.y, v); v = fmaf(w1.z, h1.z, v); v = fmaf(w1.w, h1.w, v);
            if (XH && has_x) {   // the slab of this workgroup covers its 12 hidden units
                const float4 h2 = *reinterpret_cast<const float4*>(hs2 + b * 4);
                v = fmaf(w2.x, h2.x, v); v = fmaf(w2.y, h2.y, v); v = fmaf(w2.z, h2.z, v); v = fmaf(w2.w, h2.w, v);
            }
            J.q_slab[((long)slab * B + b) * a + d] = v;
        };
        if (fixed_d) {
            // batch rows tid / a + (SK_THREADS / a) i: unrolled by 8 so that the passes' LDS reads and dependent FMA chains
            // interleave (rolled, each pass waited for its own: 2 us at the end of the launch's slowest workgroups)
            const int rstep = SK_THREADS / a, b_first = tid / a;
            for (int b0 = b_first; b0 < B; b0 += 8 * rstep) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (b0 + i * rstep < B) slab_row(b0 + i * rstep, d_f);
            }
        } else {
            for (int idx = tid; idx < B * a; idx += SK_THREADS) {
                const int b = idx / a, d = idx - b * a;
                w0 = *reinterpret_cast<const float4*>(wq + d * 8);
                w1 = *reinterpret_cast<const float4*>(wq + d * 8 + 4);
                if (XH && has_x) w2 = *reinterpret_cast<const float4*>(wq2 + d * 8);
                slab_row(b, d);
            }
        }
        // first launch beside the persistent attention kernel: do not end before that kernel is resident (afterwards this
        // launch's successors would fill every CU)
        if (J.ready_cnt && tid == 0) handoff_wait(J.ready_cnt, J.ready_target, J.tmo, 0x300u, J.spin_limit);
    }
    if (jobs.njobs >= 2) GVX_STAMP(0, 4);   // (stamps build: the single-job drain launch must not overwrite a step's stamps)
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
// teacher-forced step beside the persistent attention kernel (224 / 96 workgroups, no location workgroups)
// DEPTH: k-groups in flight per wave.  4 ships; 6 is kept selectable (GVX_PA_DEPTH=6) because it is the configuration that
// exposed the drain-pass bug of the half tiles (see SK_MFMA) and the parity suite runs it.
template <int DEPTH> __global__ __launch_bounds__(SK_THREADS) void decoder_lstm_step_pa_kernel(SkinnyJobs jobs) {
    GVX_WG_BEGIN();
    skinny_body<1, DEPTH, true, true>(jobs);
    GVX_WG_END();
}
// teacher-forced step beside the 64-CU resident attention kernel (128 < L <= 256): 192 (64) workgroups, attention-LSTM tiles in pairs
__global__ __launch_bounds__(SK_THREADS) void decoder_lstm_step_pa192_kernel(SkinnyJobs jobs) {
    GVX_WG_BEGIN();
    if (blockIdx.x < 64) skinny_body<1, SK_DEPTH1, false, true, 2>(jobs);   // uniform per workgroup
    else skinny_body<1, SK_DEPTH1, false, true, 1>(jobs);
    GVX_WG_END();
}
// teacher-forced step of 33 .. 64 rows beside the 64-CU resident attention kernel: two batch tiles per workgroup, two workgroups
// per CU (4 waves per SIMD: at most 128 VGPRs)
__global__ __launch_bounds__(SK_THREADS, 4) void decoder_lstm_step_pa64_kernel(SkinnyJobs jobs) {
    GVX_WG_BEGIN();
    skinny_body<2, SK_DEPTH2, false, true>(jobs);
    GVX_WG_END();
}
// the launch that drains the loop (decoder-LSTM of the last step): ordinary layout, context handed over in-launch
__global__ __launch_bounds__(SK_THREADS) void decoder_lstm_drain_pa_kernel(SkinnyJobs jobs) { skinny_body<1, SK_DEPTH1, false, true>(jobs); }
template <int MT> __global__ __launch_bounds__(SK_THREADS) void ar_lstm_step_kernel(SkinnyJobs jobs) {   // autoregressive launches A / C
    if ((int)blockIdx.x >= jobs.tiles) { loc_body(jobs.loc, (int)blockIdx.x - jobs.tiles); return; }
    skinny_body<MT, (MT == 1 ? SK_DEPTH1 : SK_DEPTH2), false, false, 1, true>(jobs);
}
// ---- Encoder BiLSTM recurrence, resident for the whole sequence (EncPersistParams, gvx_kernels.h).
// The launch-per-step loop pays a dispatch, a first-byte round trip for 32 KB of L2-resident weights and a kernel-end write-back
// per position (7.3 us of kernel + 1.5 us of gap for ~1 us of work).  Here workgroup (direction, tile) holds its weight
// fragments (4 k-groups per wave = 16 VGPRs), its cell states and its previous hidden values in registers; per position every
// wave reads its 4 KB of the direction's hidden vector with sc1 loads, runs 16 MFMAs, the K slices are summed through LDS, the
// cell waves finish the cells of the tile's 8 hidden units and store h(t) write-through.
// Hand-off WITHOUT flags: |h| <= 1, so bit 30 of its float pattern is always 0 - the store flips it to the generation bit of
// the position (((t >> 1) & 1) ^ 1: the two exchange buffers alternate, so what a buffer held before - zeros at the start,
// h(t-2) later - carries the other value), and a reader simply loads its share again until every dword carries the bit it
// expects, then flips it back.  Every dword validates itself (a dword store is atomic), nothing waits for a write
// acknowledgement and no flag line is shared by 32 pollers (the flag version - one word per cell wave, wait + barrier + loads,
// `s_waitcnt vmcnt(0)` in front of the flag store - took 4.5 us per position).  A non-finite h never validates: the wait
// times out and the output is poisoned - NaN comes out either way, with the status word up.
// Buffer parity: h(t) goes to buffer (t+1) & 1, which position t-1's readers have left - a tile stores h(t-1) only behind its
// own reads of position t-1, and nobody passes the reads of position t before all of h(t-1) is there.  Every wait is bounded:
// after a time-out all waits return at their first look, the grid drains, and the caller's poison launch overwrites the
// output with NaN.  Positions past a row's length get zeros from here (the caller does not clear the output).
__global__ __launch_bounds__(SK_THREADS) void encoder_lstm_persistent_kernel(EncPersistParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // two sets of partial sums [2][SK_WAVES][16][64]: one barrier per position
    const int tiles = p.H >> 3;                       // 32-row tiles per direction (4H / 32)
    // (one direction per XCD - blocks i % 8 == d of a grid of 8 x 32, the rest leaving at once - was tried: 1.97 instead of
    // 1.78 ms for the encoder stage; the exchange goes through memory either way and 32 workgroups then share one XCD's L2)
    const int dir = (int)blockIdx.x / tiles, tile = (int)blockIdx.x - dir * tiles;
    if ((int)blockIdx.x == p.debug_skip_block) return;   // (tests of the time-out path; uniform per workgroup)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bl = lane & 31, h = lane >> 5;
    const int B = p.B, L = p.L, H = p.H, nkg = H >> 3;
    const unsigned blk = (unsigned)B * 8u;
    constexpr int KPW = 4;                            // k-groups per wave (H = 256)
    constexpr unsigned GEN = 0x40000000u;             // the generation bit
    float4 wv[KPW];
    {
        const float4* wp = reinterpret_cast<const float4*>(p.Wp[dir]) + ((long)tile * nkg + wave * KPW) * 64 + lane;
#pragma unroll
        for (int i = 0; i < KPW; ++i) wv[i] = wp[i * 64];
    }
    const bool cell_wave = wave < 4;
    const int g = wave & 3, jloc = 2 * g + h, j = tile * 8 + jloc;
    const bool row = bl < B;
    const int len = row ? (p.lengths ? min(max(p.lengths[bl], 0), L) : L) : 0;   // (clamped: a bad length must not index past the row)
    float c_state = 0.f, h_state = 0.f;
    unsigned* tmo = p.sync + HANDOFF_TIMEOUT;
    const unsigned limit = (p.spin_limit ? p.spin_limit : HANDOFF_SPIN_LIMIT) * 16u;
    const long E2 = 2L * H;
    const float* xg_row = p.xg + (long)(row ? bl : 0) * L * 4 * E2 + (long)dir * 4 * H + tile * 32 + 8 * g + 4 * h;
    float* out_row = p.seq_out + (long)(row ? bl : 0) * L * E2 + (long)dir * H + j;
    float* cs_row = p.c_seq_out ? p.c_seq_out + (long)(row ? bl : 0) * L * E2 + (long)dir * H + j : nullptr;
    const unsigned x_off = ((unsigned)(row ? bl : 0) * 8u + 4u * (unsigned)h) * 4u + (unsigned)(wave * KPW) * blk * 4u;
    bool gave_up = false;   // (wave-uniform)
    for (int step = 0; step < L; ++step) {
        float* red = smem + (step & 1) * (SK_WAVES * 16 * 64);
        // the position's input projection (known since before the launch): fetched before the wait
        const bool active = cell_wave && row && step < len;
        const int tb = dir ? (len - 1 - step) : step;
        float4 ad = make_float4(0.f, 0.f, 0.f, 0.f);
        if (active) ad = *reinterpret_cast<const float4*>(xg_row + (long)tb * 4 * E2);
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        if (step > 0) {   // h(-1) = 0: nothing to multiply at the first position
            const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.hx + ((long)dir * 2 + (step & 1)) * H * B);
            const unsigned gen = ((((unsigned)(step - 1) >> 1) & 1u) ^ 1u) * GEN;   // what position step-1 stored with
            float4 xv[KPW];
            unsigned spins = 0;
            while (true) {
#pragma unroll
                for (int i = 0; i < KPW; ++i) xv[i] = load_sc1(rx, x_off + (unsigned)i * blk * 4u);
                unsigned bad = 0u;
#pragma unroll
                for (int i = 0; i < KPW; ++i) {
                    const unsigned a = __float_as_uint(xv[i].x) ^ gen, b = __float_as_uint(xv[i].y) ^ gen;
                    const unsigned c = __float_as_uint(xv[i].z) ^ gen, d = __float_as_uint(xv[i].w) ^ gen;
                    bad |= a | b | c | d;
                    xv[i] = make_float4(__uint_as_float(a), __uint_as_float(b), __uint_as_float(c), __uint_as_float(d));
                }
                if (gave_up || __all((bad & GEN) == 0u)) break;
                if ((++spins & 127u) == 1u) {   // (after a time-out every wait gives up at its first look)
                    if (__hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { gave_up = true; break; }
                    if (spins > limit) {
                        if (lane == 0) __hip_atomic_store(tmo, 0x400u + (unsigned)dir, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        gave_up = true;
                        break;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < KPW; ++i) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[i].x, xv[i].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[i].y, xv[i].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[i].z, xv[i].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[i].w, xv[i].w, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) red[(wave * 16 + q) * 64 + lane] = acc[q];
        __syncthreads();   // (the only barrier of a position: the other set of sums is written behind the next one)
        if (cell_wave) {
            float s[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < SK_WAVES; ++w) t += red[(w * 16 + 4 * g + qq) * 64 + lane];
                s[qq] = t;
            }
            if (active) {
                const float p0 = s[0] + ad.x, p1 = s[1] + ad.y, p2 = s[2] + ad.z, p3 = s[3] + ad.w;
                c_state = sigmoidf_(p1) * c_state + sigmoidf_(p0) * tanhf_(p2);
                h_state = sigmoidf_(p3) * tanhf_(c_state);
            }
            if (row) {   // inactive rows carry their state (packed-sequence semantics); rows past B are never read
                const __amdgpu_buffer_rsrc_t rh = make_rsrc(p.hx + ((long)dir * 2 + ((step + 1) & 1)) * H * B);
                const unsigned gen = ((((unsigned)step >> 1) & 1u) ^ 1u) * GEN;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h_state) ^ gen, rh, (int)(((unsigned)tile * blk + (unsigned)bl * 8u + (unsigned)jloc) * 4u), 0, 16);
                // the sequence output: the cell of an active position, zeros past the row's length (position `step` then, for
                // both directions: the reverse one walks len-1 .. 0 first)
                const long pos = active ? tb : step;
                out_row[pos * E2] = active ? h_state : 0.f;
                if (cs_row) cs_row[pos * E2] = active ? c_state : 0.f;
            }
        }
    }
}

// autoregressive launch B: the attention step of the row slices (attn_step_body.h) in the first `n_attn` workgroups, behind them
// the partial sums that only need h_a(t) - they stream while the attention's latency chain runs
__global__ __launch_bounds__(SK_THREADS, 4) void ar_attn_tiles_kernel(SkinnyJobs jobs, AttnParams ap) {   // 4 waves per SIMD: two workgroups per CU
    if ((int)blockIdx.x < jobs.block0) { attn_step_body<4, 4>(ap, (int)blockIdx.x); return; }   // uniform per workgroup
    skinny_body<1, SK_DEPTH1, false, false, 1, true>(jobs);
}
// autoregressive launch C beside the resident attention kernel: the context of the step arrives inside the launch (deferred segment)
__global__ __launch_bounds__(SK_THREADS) void ar_lstm_defer_kernel(SkinnyJobs jobs) { skinny_body<1, SK_DEPTH1, false, true>(jobs); }
// training step, back-propagation through the decoder loop: dgates x transposed recurrent matrices as partial sums (mode 2 jobs, train.hip)
__global__ __launch_bounds__(SK_THREADS) void train_bptt_products_kernel(SkinnyJobs jobs) { skinny_body<1, SK_DEPTH1>(jobs); }
template <int MT> __global__ __launch_bounds__(SK_THREADS) void encoder_lstm_step_kernel(SkinnyJobs jobs) { skinny_body<MT, (MT == 1 ? SK_DEPTH1 : SK_DEPTH2)>(jobs); }

static size_t skinny_lds(int MT) { return (size_t)(SK_WAVES * MT * 16 * 64 + MT * 32 * 8) * sizeof(float); }
static size_t skinny_pa_lds() { return skinny_lds(1) + (size_t)(SK_WAVES * 8 * 64 + 32 * 4) * sizeof(float); }

template <typename K>
static hipError_t set_lds(K kern, int MT) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)skinny_lds(MT));
}

static int sk_rot() {   // GVX_SK_ROT=<stride>: rotated k-group walk (A/B knob; 0 = off)
    static const int r = [] { const char* e = std::getenv("GVX_SK_ROT"); return e ? std::atoi(e) : 0; }();
    return r;
}

hipError_t skinny_init() {
    hipError_t e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_lstm_step_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_lstm_step_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(ar_lstm_step_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(ar_lstm_step_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_lstm_step_pa_kernel<SK_DEPTH1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)skinny_pa_lds())) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_lstm_step_pa_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)skinny_pa_lds())) != hipSuccess) return e;
    if ((e = set_lds(decoder_lstm_drain_pa_kernel, 1)) != hipSuccess) return e;
    if ((e = set_lds(decoder_lstm_step_pa192_kernel, 2)) != hipSuccess) return e;
    if ((e = set_lds(decoder_lstm_step_pa64_kernel, 2)) != hipSuccess) return e;
    if ((e = set_lds(ar_lstm_defer_kernel, 1)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(ar_attn_tiles_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = set_lds(train_bptt_products_kernel, 1)) != hipSuccess) return e;
    if ((e = set_lds(encoder_lstm_step_kernel<1>, 1)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(encoder_lstm_persistent_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SK_WAVES * 16 * 64 * (int)sizeof(float))) != hipSuccess) return e;
    return set_lds(encoder_lstm_step_kernel<2>, 2);
}

hipError_t launch_skinny_pa(const SkinnyJob& att, const SkinnyJob* dec, hipStream_t s, int depth, int layout) {
    // the layouts are built for the default layer sizes: 128 tiles per cell, batch rows in one MFMA tile
    if (att.N != 4096 || att.B < 1 || att.B > 32 || att.mode != 0 || !att.q_slab || (dec && (dec->N != 4096 || dec->B != att.B || dec->mode != 0)))
        return hipErrorInvalidValue;
    if (layout == 2) {   // attention-LSTM tiles in pairs: the slab phase exists in its MFMA form only
        if ((att.att_dim + 31) / 32 > SK_WAVES) return hipErrorInvalidValue;
        SkinnyJobs js;
        js.njobs = dec ? 2 : 1;
        js.job[0] = att; js.job[1] = dec ? *dec : att; js.job[2] = js.job[1]; js.job[3] = js.job[1];
        js.tiles0 = 128; js.tiles1 = dec ? 128 : 0; js.tiles2 = 0; js.tiles = js.tiles0 + js.tiles1;
        js.loc = LocJob{};
        js.pa_layout = 2; js.rot = sk_rot();
        decoder_lstm_step_pa192_kernel<<<dim3(dec ? 192 : 64), dim3(SK_THREADS), skinny_lds(2), s>>>(js);
        return hipGetLastError();
    }
    SkinnyJobs js;
    js.njobs = dec ? 2 : 1;
    js.job[0] = att; js.job[1] = dec ? *dec : att; js.job[2] = js.job[1]; js.job[3] = js.job[1];
    js.tiles0 = 128; js.tiles1 = dec ? 128 : 0; js.tiles2 = 0; js.tiles = js.tiles0 + js.tiles1;
    js.loc = LocJob{};
    js.pa_layout = 1; js.rot = sk_rot();
    if (depth == 6) decoder_lstm_step_pa_kernel<6><<<dim3(dec ? 224 : 96), dim3(SK_THREADS), skinny_pa_lds(), s>>>(js);
    else decoder_lstm_step_pa_kernel<SK_DEPTH1><<<dim3(dec ? 224 : 96), dim3(SK_THREADS), skinny_pa_lds(), s>>>(js);
    return hipGetLastError();
}

hipError_t launch_skinny_pa64(const SkinnyJob* jobs, int njobs, hipStream_t s) {
    if (njobs < 1 || njobs > 3) return hipErrorInvalidValue;
    SkinnyJobs js;
    js.njobs = njobs;
    js.pa_layout = 0;   // ordinary block -> (job, tile) mapping; one slab per attention-LSTM tile
    js.rot = sk_rot();
    js.loc = LocJob{};
    for (int i = 0; i < 4; ++i) js.job[i] = jobs[i < njobs ? i : njobs - 1];
    for (int i = 0; i < njobs; ++i)
        if (jobs[i].N != 4096 || jobs[i].B != jobs[0].B || jobs[i].B <= 32 || jobs[i].B > 64) return hipErrorInvalidValue;
    js.tiles0 = 128; js.tiles1 = njobs > 1 ? 128 : 0; js.tiles2 = 0; js.tiles = 128 * njobs;
    decoder_lstm_step_pa64_kernel<<<dim3(js.tiles), dim3(SK_THREADS), skinny_lds(2), s>>>(js);
    return hipGetLastError();
}

hipError_t launch_skinny(const SkinnyJob* jobs, int njobs, int kind, hipStream_t s, const LocJob* loc) {
    if (njobs < 1 || njobs > 4) return hipErrorInvalidValue;
    SkinnyJobs js;
    js.njobs = njobs;
    js.pa_layout = 0; js.rot = sk_rot();
    for (int i = 0; i < 4; ++i) js.job[i] = jobs[i < njobs ? i : njobs - 1];
    js.tiles0 = (jobs[0].N + 31) / 32;
    js.tiles1 = njobs > 1 ? (jobs[1].N + 31) / 32 : 0;
    js.tiles2 = njobs > 2 ? (jobs[2].N + 31) / 32 : 0;
    js.tiles = js.tiles0 + js.tiles1 + js.tiles2 + (njobs > 3 ? (jobs[3].N + 31) / 32 : 0);
    const int B = jobs[0].B;
    for (int i = 1; i < njobs; ++i)
        if (jobs[i].B != B) return hipErrorInvalidValue;
    for (int i = 0; i < njobs; ++i)
        if (jobs[i].mode == 2 && (jobs[i].N & 31)) return hipErrorInvalidValue;   // partial tiles store whole float4 rows
    if (B < 1 || B > 64) return hipErrorInvalidValue;
    for (int i = 0; i < njobs; ++i)
        if ((jobs[i].defer_seg || jobs[i].xw) && B > 32) return hipErrorInvalidValue;   // one batch tile only
    int extra = 0;
    js.loc = LocJob{};
    if (loc && loc->G > 0) {
        if (kind == SK_ENCODER) return hipErrorInvalidValue;
        js.loc = *loc;
        extra = loc->B * loc->G;
    }
    const int MT = B > 32 ? 2 : 1;
    if (kind == SK_TRAIN && MT != 1) return hipErrorInvalidValue;
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
        if (kind == SK_DECODER && jobs[0].defer_seg) {
            if (B > 32 || extra) return hipErrorInvalidValue;
            decoder_lstm_drain_pa_kernel<<<grid, block, lds, s>>>(js);
        } else if (kind == SK_TRAIN) {
            if (extra) return hipErrorInvalidValue;
            train_bptt_products_kernel<<<grid, block, lds, s>>>(js);
        } else if (kind == SK_DECODER) decoder_lstm_step_kernel<1><<<grid, block, lds, s>>>(js);
        else if (kind == SK_ENCODER) encoder_lstm_step_kernel<1><<<grid, block, lds, s>>>(js);
        else if (jobs[0].defer_seg) {
            if (extra) return hipErrorInvalidValue;
            ar_lstm_defer_kernel<<<grid, block, lds, s>>>(js);
        } else ar_lstm_step_kernel<1><<<grid, block, lds, s>>>(js);
    }
    return hipGetLastError();
}

bool encoder_persistent_supported(int B, int H) { return B >= 1 && B <= 32 && H == 256; }

hipError_t launch_encoder_persistent(const EncPersistParams& p, hipStream_t s) {
    if (!encoder_persistent_supported(p.B, p.H) || p.L < 1 || !p.Wp[0] || !p.Wp[1] || !p.xg || !p.hx || !p.seq_out || !p.sync) return hipErrorInvalidValue;
    const int tiles = p.H / 8;   // per direction
    encoder_lstm_persistent_kernel<<<dim3(2 * tiles), dim3(SK_THREADS), (size_t)2 * SK_WAVES * 16 * 64 * sizeof(float), s>>>(p);
    return hipGetLastError();
}

hipError_t launch_skinny_attn(const SkinnyJob* jobs, int njobs, const AttnParams& ap, hipStream_t s) {
    static_assert(MA_THREADS == SK_THREADS, "the attention step and the tiles share a launch: same workgroup size");
    if (njobs < 1 || njobs > 4 || ap.a > 128 || ap.a <= 32 || ap.G < 1) return hipErrorInvalidValue;
    if (!attention_supported(ap.L, ap.a, ap.F, ap.kl, ap.E)) return hipErrorInvalidValue;
    SkinnyJobs js;
    js.njobs = njobs;
    js.pa_layout = 0; js.rot = sk_rot();
    js.loc = LocJob{};
    for (int i = 0; i < 4; ++i) js.job[i] = jobs[i < njobs ? i : njobs - 1];
    js.tiles0 = (jobs[0].N + 31) / 32;
    js.tiles1 = njobs > 1 ? (jobs[1].N + 31) / 32 : 0;
    js.tiles2 = njobs > 2 ? (jobs[2].N + 31) / 32 : 0;
    js.tiles = js.tiles0 + js.tiles1 + js.tiles2 + (njobs > 3 ? (jobs[3].N + 31) / 32 : 0);
    const int B = jobs[0].B;
    if (B < 1 || B > 32 || B != ap.B) return hipErrorInvalidValue;   // one batch tile
    for (int i = 0; i < njobs; ++i)
        if (jobs[i].B != B || jobs[i].mode != 2 || (jobs[i].N & 31) || jobs[i].defer_seg || jobs[i].q_slab) return hipErrorInvalidValue;
    js.block0 = 8 * ((ap.B + 7) / 8) * ap.G;   // the attention step's own grid (attention.hip, launch_attention_step)
    size_t lds = skinny_lds(1);
    const size_t lds_a = (size_t)step_lds_layout(ap.a, ap.L).total * sizeof(float);
    if (lds_a > lds) lds = lds_a;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    ar_attn_tiles_kernel<<<dim3(js.block0 + js.tiles), dim3(SK_THREADS), lds, s>>>(js, ap);
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
