// Dense GEMM with fp32 operands on the bf16 matrix pipe: every fp32 value is split EXACTLY into three bf16 pieces
//   x = p1 + p2 + p3,  p1 = bf16(x), p2 = bf16(x - p1), p3 = bf16(x - p1 - p2)      (round to nearest at each level)
// and a product a * b is taken as the six piece products a1 b3 + a3 b1 + a2 b2 + a1 b2 + a2 b1 + a1 b1 (the three dropped
// ones are below 2^-26 |a b|), each exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16.  Six bf16 MFMAs of 32 cycles
// replace eight fp32 MFMAs of 64 cycles per 16 k: 2.67x less time on the matrix pipe for the same 1e-3 parity budget (the fp32
// MFMA of gfx950 runs at the vector rate; there is no xf32).  Round 1 measured this split on the weight-streaming LSTM step
// (-4 %: that kernel is bandwidth bound); the dense products - convolutions, Prenet, projections, data gradients - are matrix-
// pipe bound, which is where it pays.
//
// Same interface, tiling and epilogue as gemm_f32_kernel (gemm_f32.hip): 256 threads = 4 waves, 2 x 2 (or 1 x 2 / 1 x 1) MFMA
// tiles of 32 x 32 per wave, operands staged global -> registers -> LDS (double buffered), BK = 16.  The split happens ONCE
// per element on the way to LDS; a row of a tile in LDS is [p1: 16 k][p2: 16 k][p3: 16 k] bf16 + 16 bytes of padding =
// 112 bytes (28 words: the 16-byte fragment reads of 8 consecutive rows fall on distinct bank groups).
// MFMA operands: lane (row = lane & 31, half = lane >> 5) holds the 8 consecutive k = 8 half ... 8 half + 7 of its row.
//
// OPT-IN (GVX_GEMM_BX3=1), not the default.  Measured (round 3): Postnet on 256 x 800 frames 16.0 -> 13.0 ms, the 32 x 800
// forward 19.4 -> 18.7 ms, whole parity suite green - but while waves of this kernel are on the chip, OTHER kernels of other
// streams (the autoregressive step kernels of a second lane, of another model) stop being bit-reproducible.  Traced to the
// instruction, not to this code: a self-contained probe (tools/micro/mfma_bf16_neighbour.hip, results in
// profiles/r03_mfma_bf16_neighbour_probe.txt) shows a race-free fp32 VALU + LDS kernel returning different numbers whenever a
// kernel full of v_mfma_f32_32x32x16_bf16 (or v_mfma_f32_16x16x32_bf16 - the double-rate bf16 forms new in gfx950) runs
// beside it, and never beside v_mfma_f32_32x32x8_bf16_1k, v_mfma_f32_32x32x2_f32 or plain VALU / LDS work with the same
// launch shape (DESIGN.md section 4, round-3 experiments: "bf16 matrix instructions and their neighbours").  The split only pays at the double
// rate (at the 32x32x8 rate six piece products cost more than the fp32 instruction), so the default stays on fp32 MFMA.
#include "gvx_kernels.h"

namespace gvx {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int XBK = 16;
constexpr int XROW = 112;   // bytes per LDS row: 3 pieces x 32 bytes + 16

__device__ __forceinline__ long row_off(const RowMap& m, int row) {
    return (long)(row / m.R) * m.s1 + (long)(row % m.R) * m.s0;
}

// four floats -> their three bf16 pieces, 4 bf16 (8 bytes) each
__device__ __forceinline__ void split4(const float4& v, uint2& q1, uint2& q2, uint2& q3) {
    const f32x2_t a = {v.x, v.y}, b = {v.z, v.w};
    const bf16x2_t a1 = __builtin_convertvector(a, bf16x2_t), b1 = __builtin_convertvector(b, bf16x2_t);
    const f32x2_t ra = a - __builtin_convertvector(a1, f32x2_t), rb = b - __builtin_convertvector(b1, f32x2_t);
    const bf16x2_t a2 = __builtin_convertvector(ra, bf16x2_t), b2 = __builtin_convertvector(rb, bf16x2_t);
    const f32x2_t sa = ra - __builtin_convertvector(a2, f32x2_t), sb = rb - __builtin_convertvector(b2, f32x2_t);
    const bf16x2_t a3 = __builtin_convertvector(sa, bf16x2_t), b3 = __builtin_convertvector(sb, bf16x2_t);
    q1 = make_uint2(__builtin_bit_cast(unsigned, a1), __builtin_bit_cast(unsigned, b1));
    q2 = make_uint2(__builtin_bit_cast(unsigned, a2), __builtin_bit_cast(unsigned, b2));
    q3 = make_uint2(__builtin_bit_cast(unsigned, a3), __builtin_bit_cast(unsigned, b3));
}

template <int WR, int WC, int TM, int TN>
__global__ __launch_bounds__(256) void gemm_bx3_kernel(GemmParams p) {
    constexpr int BM = WR * TM * 32;
    constexpr int BN = WC * TN * 32;
    constexpr int A_V4 = (BM + 63) / 64;   // passes of 64 rows (256 threads = 64 rows x 4 float4 of k)
    constexpr int B_V4 = (BN + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_x[];
    unsigned char* As = smem_x;                          // [2][BM][XROW]
    unsigned char* Bs = smem_x + 2 * BM * XROW;          // [2][BN][XROW]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int r = lane & 31, h = lane >> 5;
    const int n_tiles = (p.N + BN - 1) / BN;
    const int mt = blockIdx.x / n_tiles, nt = blockIdx.x % n_tiles;
    const int m0 = p.m_begin + mt * BM, n0 = nt * BN;
    const int k_begin = p.splitk > 1 ? (int)blockIdx.y * p.kchunk : 0;
    const int k_end = p.splitk > 1 ? min(p.K, k_begin + p.kchunk) : p.K;
    float* const Cout = p.C + (p.splitk > 1 ? (long)blockIdx.y * p.c_split : 0);

    const int ld_row = tid >> 2, ld_c4 = tid & 3;
    const float* a_ptr[A_V4]; bool a_in[A_V4];
    const float* b_ptr[B_V4]; bool b_in[B_V4];
#pragma unroll
    for (int i = 0; i < A_V4; ++i) {
        const int lr = ld_row + 64 * i;
        const int m = m0 + lr;
        a_in[i] = lr < BM;
        a_ptr[i] = p.A + (a_in[i] && m < p.M ? row_off(p.amap, m) : 0);
    }
#pragma unroll
    for (int i = 0; i < B_V4; ++i) {
        const int lr = ld_row + 64 * i;
        const int n = n0 + lr;
        b_in[i] = lr < BN;
        b_ptr[i] = p.W + (b_in[i] && n < p.N ? (long)n * p.ldw : 0);
    }
    float4 a_reg[A_V4], b_reg[B_V4];

#define BX3_LOAD_TILE(K0)                                                                                   \
    {                                                                                                       \
        const int k_ = (K0) + 4 * ld_c4;                                                                    \
        const bool k_ok_ = k_ < k_end;                                                                      \
        const long a_koff_ = (long)(k_ >> 3) * p.a_kblk + (k_ & 7);                                         \
        _Pragma("unroll") for (int i = 0; i < A_V4; ++i) {                                                  \
            const float4 v_ = *reinterpret_cast<const float4*>(a_ptr[i] + (k_ok_ ? a_koff_ : 0));           \
            a_reg[i].x = k_ok_ ? v_.x : 0.f; a_reg[i].y = k_ok_ ? v_.y : 0.f;                               \
            a_reg[i].z = k_ok_ ? v_.z : 0.f; a_reg[i].w = k_ok_ ? v_.w : 0.f;                               \
        }                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < B_V4; ++i) {                                                  \
            const float4 v_ = *reinterpret_cast<const float4*>(b_ptr[i] + (k_ok_ ? k_ : 0));               \
            b_reg[i].x = k_ok_ ? v_.x : 0.f; b_reg[i].y = k_ok_ ? v_.y : 0.f;                               \
            b_reg[i].z = k_ok_ ? v_.z : 0.f; b_reg[i].w = k_ok_ ? v_.w : 0.f;                               \
        }                                                                                                   \
    }
#define BX3_STORE_TILE(BUF)                                                                                 \
    {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < A_V4; ++i) {                                                  \
            if (a_in[i]) {                                                                                  \
                uint2 q1_, q2_, q3_;                                                                        \
                split4(a_reg[i], q1_, q2_, q3_);                                                            \
                unsigned char* d_ = As + ((BUF) * BM + ld_row + 64 * i) * XROW + 8 * ld_c4;                 \
                *reinterpret_cast<uint2*>(d_) = q1_;                                                        \
                *reinterpret_cast<uint2*>(d_ + 32) = q2_;                                                   \
                *reinterpret_cast<uint2*>(d_ + 64) = q3_;                                                   \
            }                                                                                               \
        }                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < B_V4; ++i) {                                                  \
            if (b_in[i]) {                                                                                  \
                uint2 q1_, q2_, q3_;                                                                        \
                split4(b_reg[i], q1_, q2_, q3_);                                                            \
                unsigned char* d_ = Bs + ((BUF) * BN + ld_row + 64 * i) * XROW + 8 * ld_c4;                 \
                *reinterpret_cast<uint2*>(d_) = q1_;                                                        \
                *reinterpret_cast<uint2*>(d_ + 32) = q2_;                                                   \
                *reinterpret_cast<uint2*>(d_ + 64) = q3_;                                                   \
            }                                                                                               \
        }                                                                                                   \
    }
#define BX3_COMPUTE_TILE(BUF)                                                                               \
    {                                                                                                       \
        const unsigned char* a_base = As + ((BUF) * BM + wr * TM * 32 + r) * XROW + 16 * h;                 \
        const unsigned char* b_base = Bs + ((BUF) * BN + wc * TN * 32 + r) * XROW + 16 * h;                 \
        bf16x8_t af[TM][3], bf[TN][3];                                                                      \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                      \
            _Pragma("unroll") for (int q = 0; q < 3; ++q)                                                   \
                af[i][q] = *reinterpret_cast<const bf16x8_t*>(a_base + i * 32 * XROW + 32 * q);             \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                      \
            _Pragma("unroll") for (int q = 0; q < 3; ++q)                                                   \
                bf[j][q] = *reinterpret_cast<const bf16x8_t*>(b_base + j * 32 * XROW + 32 * q);             \
        /* smallest terms first */                                                                          \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                      \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0); \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0); \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0); \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0); \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0); \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0); \
            }                                                                                               \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    const int nkt = (k_end - k_begin + XBK - 1) / XBK;
    BX3_LOAD_TILE(k_begin)
    BX3_STORE_TILE(0)
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nkt; ++kt) {
        const int cur = kt & 1;
        BX3_LOAD_TILE(k_begin + (kt + 1) * XBK)
        BX3_COMPUTE_TILE(cur)
        BX3_STORE_TILE(cur ^ 1)
        __syncthreads();
    }
    BX3_COMPUTE_TILE(kt & 1)
#undef BX3_LOAD_TILE
#undef BX3_STORE_TILE
#undef BX3_COMPUTE_TILE

    // epilogue: as gemm_f32_kernel (D[row][col] with col = lane & 31, row = (q & 3) + 8 (q >> 2) + 4 (lane >> 5))
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m = m0 + (wr * TM + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
            if (m >= p.M) continue;
            const long c_off = row_off(p.cmap, m);
            const int grp = m / p.cmap.R, idx = m - grp * p.cmap.R;
            const bool live = !p.row_len || idx < p.row_len[grp];
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

template <int WR, int WC, int TM, int TN>
size_t lds_bytes() { return (size_t)2 * (WR * TM * 32 + WC * TN * 32) * XROW; }

template <int WR, int WC, int TM, int TN>
hipError_t launch_one(const GemmParams& p, hipStream_t s) {
    constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
    const int grid = ((p.M - p.m_begin + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    gemm_bx3_kernel<WR, WC, TM, TN><<<dim3(grid, p.splitk > 1 ? p.splitk : 1), dim3(256), lds_bytes<WR, WC, TM, TN>(), s>>>(p);
    return hipGetLastError();
}

}  // namespace

// Row-major operands only (A rows through amap, W rows n * ldw); shapes the fp32 kernel's narrow configurations serve
// (N <= 96) stay there.  Same tile-shape policy as launch_gemm, incl. the two-shape cover of a nearly empty last round.
bool gemm_bx3_serves(const GemmParams& p) { return !p.kmajor && p.N > 96; }

hipError_t launch_gemm_bx3(const GemmParams& p, hipStream_t s) {
    if (p.M <= 0 || p.N <= 0) return hipSuccess;
    if (p.K & 3) return hipErrorInvalidValue;
    const long n_tiles = (p.N + 127) / 128, tiles128 = (long)((p.M - p.m_begin + 127) / 128) * n_tiles;
    if (tiles128 < 384) return launch_one<2, 2, 1, 2>(p, s);
    const long rem = tiles128 % 256;
    if (p.splitk <= 1 && p.m_begin == 0 && rem > 0 && rem <= 64) {
        const long rows_big = ((tiles128 - rem) / n_tiles) * 128;
        if (rows_big > 0 && rows_big < p.M) {
            GemmParams a = p, b = p;
            a.M = (int)rows_big;
            b.m_begin = (int)rows_big;
            const hipError_t e = launch_one<2, 2, 2, 2>(a, s);
            if (e != hipSuccess) return e;
            return launch_one<2, 2, 1, 1>(b, s);
        }
    }
    return launch_one<2, 2, 2, 2>(p, s);
}

hipError_t gemm_bx3_init() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bx3_kernel<2, 2, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bx3_kernel<2, 2, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bx3_kernel<2, 2, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace gvx
