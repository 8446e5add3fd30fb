// Dense fp32 GEMM on the gfx950 f32 matrix instruction (v_mfma_f32_32x32x2_f32): exact fp32,
// bit-for-bit a k-ordered fmaf chain per output, which is what the 1e-3 parity budget needs
// (bf16/f16 MFMA inputs would not fit it; there is no xf32 on gfx950).
//
// Used for every "dense over all positions" contraction on the path:
//   encoder convs, encoder LSTM input projection, memory projection, Prenet, mel/gate projection
//   (teacher-forced: hoisted out of the step loop), Postnet convs.
// conv1d(k, pad (k-1)/2) is an implicit GEMM: activations are kept channels-last with a zero halo,
// [B][T+2p][C], so the im2col row of (b,t) is simply the 5*C contiguous floats starting at row t
// of the padded buffer - no gather, no materialised im2col.  The conv weight is repacked at load
// time to [Cout][k][Cin] with eval-mode BatchNorm folded in.
//
// Tile: one wave per (WR, WC) cell of the shape, BK = 32, LDS double buffered through registers.
//   <4,2,1,2>: 128x128 tile on eight waves of 32x64                 - general case (row-major operands)
//   <2,2,2,2>: 128x128 tile, each of four waves 64x64               - its predecessor (GVX_GEMM_8W=0) and the K-major form
//   <2,2,1,2> / <2,2,1,1>: 64x128 / 64x64 tiles of four waves       - few tiles (encoder convolutions), remainders
//   <4,1,1,3>: 128x96 tile, each wave 32x96 (1x3 MFMA tiles)        - N <= 96 (the last Postnet conv, the mel / gate projection)
//   <2,3,1,1>: 64x96 tile of six waves, each 32x32                  - N <= 96 and few rows (a wave's MFMA chain is the launch)
//   <4,1,1,1>: 128x32                                              - N <= 32
// Eight waves instead of four on the big tile: two waves per SIMD inside ONE workgroup - a workgroup's epilogue (64 KB of
// stores) and its barrier waits run beside another wave's products (the attention LSTM's Prenet columns, K = 256 and 419 MB
// of output: encoder stage 1.29 -> 1.19 ms; Postnet 2.15 -> 2.11 ms).  Tried for N <= 96 and dropped: 64x96 tiles of six
// waves with one MFMA tile each (MFMA busy 0.40 vs 0.44), k-tiles of 64 (0.46, no better once the loads sat right).
// LDS rows are 36 floats (144 B): with that stride the ds_read_b128 fragment reads of the
// sixteen-lane groups fall on distinct 16-byte bank slots (conflict free).
// MFMA operand trick: one float4 per lane feeds four k-steps (k = 8*kg + 4*(lane>>5) + s), so both
// operands are read as 16-byte vectors from K-contiguous rows; the k order inside a group of 8 is
// permuted, which only reorders the fp32 summation.
#include "gvx_kernels.h"

#include <cstdlib>

namespace gvx {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BK32 = 32;
// (Rounds 1-3 carried a second version of these products on the bf16 matrix pipe - every fp32 operand split exactly into three bf16
// pieces, six v_mfma_f32_32x32x16_bf16 per 16 k - which was 19 % faster on the Postnet and parity-green.  Retired in round 4: the
// double-rate 16-bit MFMA forms of gfx950 corrupt fp32 kernels that run on the SAME OR A NEIGHBOURING CU at the same time
// (tools/micro/mfma_bf16_neighbour.hip, profiles/r04_mfma_bf16_neighbour_cumask.txt: 0 wrong words on disjoint halves of the CU
// mask, thousands on shared or interleaved CUs), and a library cannot choose its neighbours on the chip.)

__device__ __forceinline__ long row_off(const RowMap& m, int row) {
    return (long)(row / m.R) * m.s1 + (long)(row % m.R) * m.s0;
}

// KMAJ: both operands are K-major - element (m, k) of A at A + amap(k) + m, element (n, k) of W at W + wmap(k) + n - the
// form of a weight gradient (sum over the rows of two activation matrices) without transposed copies.
template <int WR, int WC, int TM, int TN, bool KMAJ = false, int BK = BK32>
__device__ __forceinline__ void gemm_f32_body(const GemmParams& p, const int block_x, const int tid, float* smem) {
    constexpr int LDS_LD = BK + 4;   // 36 / 68 floats: see the header
    static_assert(BK == 32 || (BK == 64 && !KMAJ), "k-tiles of 32, or 64 for the row-major form");
    constexpr int BM = WR * TM * 32;
    constexpr int BN = WC * TN * 32;
    constexpr int NT = WR * WC * 64;  // threads: 256 for the four-wave shapes, 384 for <2,3,1,1>
    constexpr int TPR = BK / 4;       // threads per tile row (16 bytes each)
    constexpr int RPP = NT / TPR;     // tile rows one pass of the loaders covers
    constexpr int A_V4 = (BM + RPP - 1) / RPP;  // float4 loads per thread for the A tile (the last pass may be partial)
    constexpr int B_V4 = (BN + RPP - 1) / RPP;
    static_assert(!KMAJ || NT == 256, "the K-major loaders are written for 256 threads");
    float* As = smem;                         // [2][BM][LDS_LD]
    float* Bs = smem + 2 * BM * LDS_LD;       // [2][BN][LDS_LD]

    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int r = lane & 31, h = lane >> 5;
    // consecutive blocks share the W panel (same n-tile) -> neighbouring M tiles stay in one L2
    const int n_tiles = (p.N + BN - 1) / BN;
    const int mt = block_x / n_tiles, nt = block_x % n_tiles;
    const int m0 = p.m_begin + mt * BM, n0 = nt * BN;
    // split-K: this workgroup's k range and partial output (the whole K and C itself when splitk == 1)
    const int k_begin = p.splitk > 1 ? (int)blockIdx.y * p.kchunk : 0;
    const int k_end = p.splitk > 1 ? min(p.K, k_begin + p.kchunk) : p.K;
    float* const Cout = p.C + (p.splitk > 1 ? (long)blockIdx.y * p.c_split : 0);

    // global -> register staging assignments
    const int ld_row = tid / TPR, ld_c4 = tid % TPR;
    const float* a_ptr[A_V4]; bool a_ok[A_V4];
    const float* b_ptr[B_V4]; bool b_ok[B_V4];
    // K-major operands: a thread owns 4 consecutive rows of the tile (m or n) and one quad of k per pass: four 16-byte loads
    // (one per k, coalesced across the lanes along m), a 4 x 4 transpose in registers, four 16-byte LDS stores (one per m) in
    // the same [row][k] layout as the row-major form.  Needs M, N and the row strides to be multiples of 4.
    constexpr int A_MQ = BM / 4, B_NQ = BN / 4;             // threads along m / n
    constexpr int A_P = (8 * A_MQ + 255) / 256, B_P = (8 * B_NQ + 255) / 256;   // passes over the tile's 8 k-quads
    const int amq = tid % A_MQ, akq = tid / A_MQ, bnq = tid % B_NQ, bkq = tid / B_NQ;
    const int a_col = min(m0 + 4 * amq, p.M - 4), b_col = min(n0 + 4 * bnq, p.N - 4);   // (rows past M / N are never stored)
    // (group, index) of the first k of the thread's quads under the two-level row maps, kept up to date from k-tile to k-tile
    // by increments (a division per load would make the K-major form VALU bound)
    int ak_g[A_P], ak_i[A_P], bk_g[B_P], bk_i[B_P];
    long a_safe = 0, b_safe = 0;
    float4 ka_reg[A_P][4], kb_reg[B_P][4];
    if (KMAJ) {
#pragma unroll
        for (int i = 0; i < A_P; ++i) { const int k = k_begin + 4 * (akq + (256 / A_MQ) * i); ak_g[i] = k / p.amap.R; ak_i[i] = k - ak_g[i] * p.amap.R; }
#pragma unroll
        for (int i = 0; i < B_P; ++i) { const int k = k_begin + 4 * (bkq + (256 / B_NQ) * i); bk_g[i] = k / p.wmap.R; bk_i[i] = k - bk_g[i] * p.wmap.R; }
        a_safe = row_off(p.amap, k_begin); b_safe = row_off(p.wmap, k_begin);
    }
    if (!KMAJ) {
#pragma unroll
        for (int i = 0; i < A_V4; ++i) {
            int m = m0 + ld_row + RPP * i;
            a_ok[i] = m < p.M;
            a_ptr[i] = p.A + (a_ok[i] ? row_off(p.amap, m) : 0);
        }
#pragma unroll
        for (int i = 0; i < B_V4; ++i) {
            int n = n0 + ld_row + RPP * i;
            b_ok[i] = n < p.N;
            b_ptr[i] = p.W + (b_ok[i] ? (long)n * p.ldw : 0);
        }
    }
    float4 a_reg[A_V4], b_reg[B_V4];

    // global -> registers for the k-tile starting at K0.  Rows past M / N read row 0 (in bounds): their products land in
    // accumulator rows / columns the epilogue never stores; the k tail reads in-bounds addresses too and is zeroed at the LDS
    // stores (GEMM_STORE_TILE), so that nothing depends on the loaded values while the products of the current tile run.
#define GEMM_KMAJ_LOAD(NP, QPP, KQ, KG, KI, MAP, SAFE, PTR, COL, REG)                                        \
        _Pragma("unroll") for (int i = 0; i < NP; ++i) {                                                    \
            const int c_ = KQ + QPP * i;                                                                    \
            if (c_ < BK / 4) {                                                                              \
                const int kq_ = (K0_) + 4 * c_;                                                             \
                int g_ = KG[i], x_ = KI[i];                                                                 \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                             \
                    const bool ok_ = kq_ + j < k_end;                                                       \
                    REG[i][j] = *reinterpret_cast<const float4*>(PTR + (ok_ ? (long)g_ * MAP.s1 + (long)x_ * MAP.s0 : SAFE) + COL); \
                    if (++x_ == MAP.R) { x_ = 0; ++g_; }                                                    \
                }                                                                                           \
                x_ = KI[i] + BK; g_ = KG[i];                                                                \
                while (x_ >= MAP.R) { x_ -= MAP.R; ++g_; }                                                  \
                KI[i] = x_; KG[i] = g_;                                                                     \
            }                                                                                               \
        }
#define GEMM_LOAD_TILE(K0)                                                                                  \
    if (KMAJ) {                                                                                             \
        const int K0_ = (K0);                                                                               \
        GEMM_KMAJ_LOAD(A_P, (256 / A_MQ), akq, ak_g, ak_i, p.amap, a_safe, p.A, a_col, ka_reg)              \
        GEMM_KMAJ_LOAD(B_P, (256 / B_NQ), bkq, bk_g, bk_i, p.wmap, b_safe, p.W, b_col, kb_reg)              \
    } else                                                                                                  \
    {                                                                                                       \
        const int k_ = (K0) + 4 * ld_c4;                                                                    \
        const bool k_ok_ = k_ < k_end;                                                                      \
        const long a_koff_ = (long)(k_ >> 3) * p.a_kblk + (k_ & 7);                                         \
        _Pragma("unroll") for (int i = 0; i < A_V4; ++i) {                                                  \
            if (BM % RPP != 0 && ld_row + RPP * i >= BM) continue;                                          \
            a_reg[i] = *reinterpret_cast<const float4*>(a_ptr[i] + (k_ok_ ? a_koff_ : 0));                  \
        }                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < B_V4; ++i) {                                                  \
            if (BN % RPP != 0 && ld_row + RPP * i >= BN) continue;                                          \
            b_reg[i] = *reinterpret_cast<const float4*>(b_ptr[i] + (k_ok_ ? k_ : 0));                       \
        }                                                                                                   \
    }
#define GEMM_KMAJ_STORE(NP, QPP, KQ, SM, ROWS, RQ, REG, BUF)                                                 \
        _Pragma("unroll") for (int i = 0; i < NP; ++i) {                                                    \
            const int c_ = KQ + QPP * i;                                                                    \
            if (c_ < BK / 4) {                                                                              \
                float* d_ = &SM[((BUF) * ROWS + 4 * RQ) * LDS_LD + 4 * c_];                                 \
                const int kq_ = (KS_) + 4 * c_;   /* the k tail: zeros, here and not at the loads (see GEMM_LOAD_TILE) */ \
                const bool o0_ = kq_ < k_end, o1_ = kq_ + 1 < k_end, o2_ = kq_ + 2 < k_end, o3_ = kq_ + 3 < k_end;   \
                *reinterpret_cast<float4*>(d_) = make_float4(o0_ ? REG[i][0].x : 0.f, o1_ ? REG[i][1].x : 0.f, o2_ ? REG[i][2].x : 0.f, o3_ ? REG[i][3].x : 0.f);              \
                *reinterpret_cast<float4*>(d_ + LDS_LD) = make_float4(o0_ ? REG[i][0].y : 0.f, o1_ ? REG[i][1].y : 0.f, o2_ ? REG[i][2].y : 0.f, o3_ ? REG[i][3].y : 0.f);     \
                *reinterpret_cast<float4*>(d_ + 2 * LDS_LD) = make_float4(o0_ ? REG[i][0].z : 0.f, o1_ ? REG[i][1].z : 0.f, o2_ ? REG[i][2].z : 0.f, o3_ ? REG[i][3].z : 0.f); \
                *reinterpret_cast<float4*>(d_ + 3 * LDS_LD) = make_float4(o0_ ? REG[i][0].w : 0.f, o1_ ? REG[i][1].w : 0.f, o2_ ? REG[i][2].w : 0.f, o3_ ? REG[i][3].w : 0.f); \
            }                                                                                               \
        }
#define GEMM_STORE_TILE(BUF, K0)                                                                            \
    if (KMAJ) {                                                                                             \
        const int KS_ = (K0);                                                                               \
        GEMM_KMAJ_STORE(A_P, (256 / A_MQ), akq, As, BM, amq, ka_reg, BUF)                                   \
        GEMM_KMAJ_STORE(B_P, (256 / B_NQ), bkq, Bs, BN, bnq, kb_reg, BUF)                                   \
    } else                                                                                                  \
    {                                                                                                       \
        const bool s_ok_ = (K0) + 4 * ld_c4 < k_end;   /* the k tail: zeros (per component: a float4 select makes LLVM build a scratch lookup table) */ \
        _Pragma("unroll") for (int i = 0; i < A_V4; ++i)                                                    \
            if (BM % RPP == 0 || ld_row + RPP * i < BM)                                                     \
                *reinterpret_cast<float4*>(&As[((BUF) * BM + ld_row + RPP * i) * LDS_LD + 4 * ld_c4]) =     \
                    make_float4(s_ok_ ? a_reg[i].x : 0.f, s_ok_ ? a_reg[i].y : 0.f, s_ok_ ? a_reg[i].z : 0.f, s_ok_ ? a_reg[i].w : 0.f); \
        _Pragma("unroll") for (int i = 0; i < B_V4; ++i)                                                    \
            if (BN % RPP == 0 || ld_row + RPP * i < BN)                                                     \
                *reinterpret_cast<float4*>(&Bs[((BUF) * BN + ld_row + RPP * i) * LDS_LD + 4 * ld_c4]) =     \
                    make_float4(s_ok_ ? b_reg[i].x : 0.f, s_ok_ ? b_reg[i].y : 0.f, s_ok_ ? b_reg[i].z : 0.f, s_ok_ ? b_reg[i].w : 0.f); \
    }
#define GEMM_COMPUTE_TILE(BUF) GEMM_COMPUTE_KGS(BUF, 0, BK / 8)
#define GEMM_COMPUTE_KGS(BUF, KG0, KG1)   /* k-groups KG0 .. KG1 - 1 of the tile in buffer BUF */           \
    {                                                                                                       \
        const float* a_base = &As[((BUF) * BM + wr * TM * 32 + r) * LDS_LD + 4 * h];                        \
        const float* b_base = &Bs[((BUF) * BN + wc * TN * 32 + r) * LDS_LD + 4 * h];                        \
        _Pragma("unroll") for (int kg = (KG0); kg < (KG1); ++kg) {                                          \
            float4 af[TM], bf[TN];                                                                          \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                  \
                af[i] = *reinterpret_cast<const float4*>(a_base + i * 32 * LDS_LD + 8 * kg);                \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                  \
                bf[j] = *reinterpret_cast<const float4*>(b_base + j * 32 * LDS_LD + 8 * kg);                \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                  \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                            \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0); \
                }                                                                                           \
        }                                                                                                   \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    // rows that do not exist read row 0 / weight row 0 (in bounds) and are zeroed by the select above
    const int nkt = (k_end - k_begin + BK - 1) / BK;
    GEMM_LOAD_TILE(k_begin)
    GEMM_STORE_TILE(0, k_begin)
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nkt; ++kt) {            // steady state: prefetch tile kt+1 while computing tile kt (no branches inside)
        const int cur = kt & 1;
        GEMM_LOAD_TILE(k_begin + (kt + 1) * BK)
        // The loads stay in front of the products, and what they return is first touched (k-tail zeros, LDS stores) beside the
        // products of the tile's LAST k-group, in the shadow of its MFMAs.  Left to itself the compiler put loads, selects and stores
        // together - three quarters through the MFMA sequence (a whole memory latency exposed per k-tile), or right behind the
        // loads once the selects had moved to the stores.
        __builtin_amdgcn_sched_barrier(0);
        GEMM_COMPUTE_KGS(cur, 0, BK / 8 - 1)
        __builtin_amdgcn_sched_barrier(0);
        GEMM_COMPUTE_KGS(cur, BK / 8 - 1, BK / 8)
        GEMM_STORE_TILE(cur ^ 1, k_begin + (kt + 1) * BK)
        __syncthreads();
    }
    GEMM_COMPUTE_TILE(kt & 1)               // last tile
#undef GEMM_LOAD_TILE
#undef GEMM_KMAJ_LOAD
#undef GEMM_KMAJ_STORE
#undef GEMM_STORE_TILE
#undef GEMM_COMPUTE_TILE
#undef GEMM_COMPUTE_KGS

    // epilogue: D[row][col] with col = lane&31, row = (q&3) + 8*(q>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m = m0 + (wr * TM + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
            if (m >= p.M) continue;
            const long c_off = row_off(p.cmap, m);
            const int grp = m / p.cmap.R, idx = m - grp * p.cmap.R;
            const bool live = !p.row_len || idx < p.row_len[grp];
            // conv output with halo rows: the first / last c_halo interior rows of a sequence also zero one halo row each
            const bool halo_front = p.c_halo > 0 && idx < p.c_halo;
            const bool halo_back = p.c_halo > 0 && idx >= p.cmap.R - p.c_halo;
            const long halo_step = (long)p.c_halo * p.cmap.s0;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + (wc * TN + j) * 32 + r;
                if (n >= p.N) continue;
                float v = acc[i][j][q];
                if (p.bias) v += p.bias[n];
                if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
                else if (p.act == ACT_TANH) v = tanhf(v);
                if (p.keep) v = p.keep[(long)m * p.keep_ld + n] ? 2.f * v : 0.f;
                const long col = (long)(n >> 3) * p.c_nblk + (n & 7);
                Cout[c_off + col] = live ? v : 0.f;
                if (halo_front) Cout[c_off - halo_step + col] = 0.f;
                if (halo_back) Cout[c_off + halo_step + col] = 0.f;
            }
        }
    }
}

template <int WR, int WC, int TM, int TN, bool KMAJ = false, int BK = BK32>
__global__ __launch_bounds__(WR * WC * 64) void gemm_f32_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_f32_body<WR, WC, TM, TN, KMAJ, BK>(p, (int)blockIdx.x, (int)threadIdx.x, smem);
}

template <int WR, int WC, int TM, int TN, int BK = BK32>
static size_t lds_bytes_cfg() { return (size_t)2 * (WR * TM * 32 + WC * TN * 32) * (BK + 4) * sizeof(float); }

template <int WR, int WC, int TM, int TN, bool KMAJ = false, int BK = BK32>
static hipError_t init_cfg() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<WR, WC, TM, TN, KMAJ, BK>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes_cfg<WR, WC, TM, TN, BK>());
}

hipError_t gemm_init() {
    hipError_t e = init_cfg<2, 2, 2, 2>();
    if (e != hipSuccess) return e;
    e = init_cfg<2, 2, 1, 2>();
    if (e != hipSuccess) return e;
    e = init_cfg<4, 1, 1, 3>();
    if (e != hipSuccess) return e;
    e = init_cfg<2, 3, 1, 1>();
    if (e != hipSuccess) return e;
    e = init_cfg<4, 2, 1, 2>();
    if (e != hipSuccess) return e;
    e = init_cfg<2, 2, 1, 1>();
    if (e != hipSuccess) return e;
    e = init_cfg<2, 2, 2, 2, true>();
    if (e != hipSuccess) return e;
    e = init_cfg<2, 2, 1, 2, true>();
    if (e != hipSuccess) return e;
    return init_cfg<4, 1, 1, 1>();
}

template <int WR, int WC, int TM, int TN, bool KMAJ = false, int BK = BK32>
static hipError_t launch_cfg(const GemmParams& p, hipStream_t s) {
    constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
    const int grid = ((p.M - p.m_begin + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    gemm_f32_kernel<WR, WC, TM, TN, KMAJ, BK><<<dim3(grid, p.splitk > 1 ? p.splitk : 1), dim3(WR * WC * 64), lds_bytes_cfg<WR, WC, TM, TN, BK>(), s>>>(p);
    return hipGetLastError();
}

// GVX_GEMM_8W=0 (A/B): the 128 x 128 tile on four waves of 64 x 64 instead of eight of 32 x 64
static bool four_waves() {
    static const bool on = [] { const char* e = std::getenv("GVX_GEMM_8W"); return e && e[0] == '0'; }();
    return on;
}

hipError_t launch_gemm(const GemmParams& p, hipStream_t s) {
    if (p.M <= 0 || p.N <= 0) return hipSuccess;
    if (p.kmajor) {   // K-major operands (weight gradients): the two square-ish tile shapes only, any K; 16-byte pieces along m / n
        if ((p.M & 3) || (p.N & 3) || p.M < 4 || p.N < 4 || ((p.amap.s0 | p.amap.s1 | p.wmap.s0 | p.wmap.s1) & 3)) return hipErrorInvalidValue;
        const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
        return t128 < 384 ? launch_cfg<2, 2, 1, 2, true>(p, s) : launch_cfg<2, 2, 2, 2, true>(p, s);
    }
    if (p.K & 3) return hipErrorInvalidValue;
    if (p.N <= 32) return launch_cfg<4, 1, 1, 1>(p, s);
    if (p.N <= 96) {
        // few rows (a single utterance: 568 frames = 5 tiles of 128 rows): the launch lasts as long as ONE wave's chain of dependent
        // MFMAs over K - three accumulators per wave in the 128 x 96 tile (151 us for the last Postnet convolution on 800 frames),
        // one in the 64 x 96 tile of six waves.  Many rows: the six-wave tile is no faster (MFMA busy 0.40 vs 0.44, round 4)
        if (p.M - p.m_begin <= 4096) return launch_cfg<2, 3, 1, 1>(p, s);
        return launch_cfg<4, 1, 1, 3>(p, s);
    }
    // fewer than ~1.5 workgroups per CU with 128 x 128 tiles (encoder convolutions: 4096 x 512): halve the tile height so
    // that all 256 CUs get work
    const long n_tiles = (p.N + 127) / 128, tiles128 = (long)((p.M + 127) / 128) * n_tiles;
    // and below ~0.75 per CU (the encoder convolutions themselves: 128 tiles) 64 x 64 tiles, two workgroups per CU: one
    // 64 x 128 workgroup of four waves per CU left the matrix pipe waiting on its own loads (155 -> 115 us per convolution)
    if (tiles128 < 192) return launch_cfg<2, 2, 1, 1>(p, s);
    if (tiles128 < 384) return launch_cfg<2, 2, 1, 2>(p, s);
    // Wave quantisation: a last round of 128 x 128 tiles with few tiles leaves most of the chip idle for a whole tile time (the
    // Postnet at 32 x 800 frames: 800 tiles = 3 per CU + 32).  When that remainder is at most a quarter of a round, the rows of
    // the full rounds get the big tiles and the remaining rows 64 x 64 tiles (four times as many workgroups, a quarter of the
    // time each) in a second launch.  Every output element still sums its K products in the same order: results do not depend
    // on the tile shape.  (Rounds 2-4 ran both shapes in ONE launch, the small tiles filling the CUs as the last big round
    // drained: worth 0.2 ms per Postnet with four-wave tiles, but 2.18 vs 2.12 ms - slower - with the eight-wave ones.)
    const long rem = tiles128 % 256;
    if (p.splitk <= 1 && p.m_begin == 0 && rem > 0 && rem <= 64) {
        const long rows_big = ((tiles128 - rem) / n_tiles) * 128;   // whole row tiles of the full rounds
        if (rows_big > 0 && rows_big < p.M) {
            GemmParams a = p, b = p;
            a.M = (int)rows_big;
            b.m_begin = (int)rows_big;
            const hipError_t e = four_waves() ? launch_cfg<2, 2, 2, 2>(a, s) : launch_cfg<4, 2, 1, 2>(a, s);
            if (e != hipSuccess) return e;
            return launch_cfg<2, 2, 1, 1>(b, s);
        }
    }
    return four_waves() ? launch_cfg<2, 2, 2, 2>(p, s) : launch_cfg<4, 2, 1, 2>(p, s);
}

// out[m][n] = sum_s part[s][m][n] (+ bias[n]), splits added in index order
__global__ void splitk_reduce_kernel(const float* part, int splitk, long MN, int N, const float* bias, float* out, long ldc) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < MN; i += (long)gridDim.x * blockDim.x) {
        float v = part[i];
        for (int s = 1; s < splitk; ++s) v += part[(long)s * MN + i];
        const int n = (int)(i % N);
        if (bias) v += bias[n];
        out[(i / N) * ldc + n] = v;
    }
}

hipError_t launch_gemm_splitk(const GemmParams& p0, int splitk, float* scratch, hipStream_t s) {
    if (splitk <= 1) return launch_gemm(p0, s);
    if (p0.act != ACT_NONE || p0.keep || p0.row_len || p0.c_halo || p0.c_nblk != 8 || p0.cmap.s1 != 0 || p0.cmap.R < p0.M || !scratch)
        return hipErrorInvalidValue;   // plain row-major outputs only
    GemmParams p = p0;
    const int kc = ((p.K + splitk - 1) / splitk + 63) / 64 * 64;   // (whole k-tiles of either size)
    p.splitk = (p.K + kc - 1) / kc;
    p.kchunk = kc;
    p.C = scratch; p.cmap = RowMap{p.M, 0, (long)p.N};
    p.c_split = (long)p.M * p.N;
    p.bias = nullptr;
    hipError_t e = launch_gemm(p, s);
    if (e != hipSuccess) return e;
    const long MN = (long)p.M * p.N;
    long blocks = (MN + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    splitk_reduce_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(scratch, p.splitk, MN, p.N, p0.bias, p0.C, p0.cmap.s0);
    return hipGetLastError();
}

}  // namespace gvx
