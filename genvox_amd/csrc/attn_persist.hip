// Location-sensitive attention of the teacher-forced decoder loop as ONE kernel that lives for the whole loop
// (models/tts/tacotron2.py:243-262 Attention.forward, :317-348 Decoder.decode).
//
// Why: per step the attention reads, for one batch row, the encoder memory (L x E, 256 KB), the processed memory
// (L x a, 64 KB), the location features (64 KB) and the query partial sums (64 KB).  All but the last are either constant
// over the decode or produced by the attention itself, yet a per-step launch has to fetch them again through a fabric that
// the LSTM weight stream saturates (measured: 7-8 us per launch on the step's critical chain, 11-14 us when run beside
// the stream).  Here one workgroup per batch row keeps the row's memory (64 VGPRs x 1024 threads) and processed memory
// (16 VGPRs) in registers and its location features in LDS for all T steps (conv + dense on the MFMA units); per step it
// only receives the 48 KB of query partial sums from the 96 attention-LSTM workgroups and publishes 2 KB of context.  It
// runs BESIDE the LSTM launches (on CUs of its own, 32 of 256 - a kernel that shares a CU with a weight-streaming tile gets
// half its issue slots and a full memory queue: measured 12.8 instead of 5.8 us for this chain), so the chain
// slabs(t) -> energies -> softmax -> context(t)  overlaps the next launch's streaming of the weight columns that do not
// depend on the context; that launch waits for context(t) just before its context columns (skinny.hip, deferred segment)
// and is dealt to 224 workgroups of equal weight, because a CU streams ~25 KB/us whatever shares it.
//
// Hand-offs between the two kernels follow cdna_hip_programming.md section 6 guideline 16 / MI355X_MICROARCH.md
// "inter-workgroup visibility" (per-XCD L2s are not coherent):
//   * the context is stored write-through (`sc1` 16-byte buffer stores), the storing waves drain (`s_waitcnt vmcnt(0)`), the
//     workgroup meets at a barrier, ONE lane adds to an agent-scope counter; the LSTM waves poll that counter with `sc1`
//     loads (one wave per workgroup, bounded spin with back-off) and read the context with `sc1` buffer loads only;
//   * the query slabs need no in-kernel publication: they are plain stores of launch t, and block 0 of launch t + 1 adds 1 to
//     a second counter when it STARTS - by then launch t has completed and its stores have been written back (a
//     write-through store + drain at the end of every attention-LSTM workgroup cost the launch 6 us).  They are read with
//     `sc1` buffer loads (never plain loads: L1 / L2 may hold last step's lines).
// Every spin is bounded: on a timeout the waiter raises the workspace's hand-off status word and every later wait returns
// at once, so both kernels drain (with wrong results, which gvx_workspace_status reports) instead of hanging the GPU.
#include "gvx_kernels.h"

namespace gvx {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int PA_THREADS = 1024;
constexpr int PA_A = 128;      // attention dim
constexpr int PA_E = 512;      // encoder embedding dim
constexpr int PA_L = 128;      // positions held per row
constexpr int PA_FB_S = 33;    // LDS row stride of the conv output [l][32 filters] (odd: the dense product's lanes read one filter of 32 positions, no bank conflicts)
constexpr int PA_KL_MAX = 31;
constexpr int PA_SLABS = 96;   // workgroups of the attention LSTM in the teacher-forced launch's layout (skinny.hip): one query slab each
constexpr int PA_SLABS_AR = 128;   // autoregressive launches: one slab per attention-LSTM tile
constexpr int PA_WC_S = PA_L + 32;   // stride of the two halo-padded weight rows (prev, cum); taps are padded to 32
constexpr int PA_XCH = PA_E + 64;    // floats of one exchange buffer of the split kernel: c[512], (m, s, -, -), edge energies[16]

// LDS layout (floats)
constexpr int PA_OFF_LOC = 0;                                  // [128][128] location features, float4 groups swizzled (pa_loc_swz)
constexpr int PA_OFF_FB = PA_OFF_LOC + PA_L * PA_A;            // [128][36]
constexpr int PA_OFF_CW = PA_OFF_FB + PA_L * PA_FB_S;          // [2][32][32] conv weights, taps zero-padded to 32
constexpr int PA_OFF_WD = PA_OFF_CW + 2 * 32 * 32;             // [32 filters][128] dense weights (filter-major: lanes read 32 consecutive dims)
constexpr int PA_OFF_WC = PA_OFF_WD + 32 * PA_A;               // [2][PA_WC_S]
constexpr int PA_OFF_QP = PA_OFF_WC + 2 * PA_WC_S;             // [32][128] query partial sums
constexpr int PA_OFF_QS = PA_OFF_QP + 32 * PA_A;               // [128] query
constexpr int PA_OFF_ES = PA_OFF_QS + PA_A;                    // [128] energies
constexpr int PA_OFF_V = PA_OFF_ES + PA_L;                     // [128] v
constexpr int PA_OFF_CP = PA_OFF_V + PA_A;                     // [8][512] context partial sums
constexpr int PA_OFF_FLAG = PA_OFF_CP + 8 * PA_E;              // [4] "leave the loop" word of the step's wait
constexpr int PA_OFF_H1 = PA_OFF_FLAG + 4;                     // autoregressive role: [128] projection bias, [1] "this row's stop token has fired"
constexpr int PA_OFF_MEML = PA_OFF_H1 + 256;                   // autoregressive role: [1024] float4, the 16th memory vector of every thread (below)
constexpr int PA_LDS_FLOATS = PA_OFF_MEML + 4 * PA_THREADS;
constexpr int PA_P = 256;      // Prenet width (autoregressive role)
constexpr int PA_KPT = 5;      // layer-1 k values per thread: n_mels <= 16 PA_KPT
constexpr int PA_SP_S = 25;    // float4 row stride of the slab partial sums [32][PSB / 4 <= 24]: odd - their column reads are conflict free (with 32,
                               // a half wave's 32 rows of a column sat in the same banks: 2 us of the row's second part)
static_assert(PA_LDS_FLOATS * 4 <= 160 * 1024, "persistent attention LDS");
static_assert((PA_OFF_FB % 4) == 0 && (PA_OFF_CW % 4) == 0 && (PA_OFF_WD % 4) == 0 && (PA_OFF_QP % 4) == 0 && (PA_OFF_QS % 4) == 0 &&
              (PA_OFF_V % 4) == 0 && (PA_OFF_CP % 4) == 0 && (PA_OFF_H1 % 4) == 0 && (PA_OFF_MEML % 4) == 0, "float4 alignment");

// float4 group g of position l of the location features sits at group g ^ pa_loc_swz(l) of its 128-float row: bit 3 by the position's
// parity (the energies phase: 8 lanes read a row's groups dg + 8 j, neighbouring positions in the other half of the banks), bit 2
// by bit 2 of the position (the dense product's writes: a half wave writes 16 dims of two positions 4 apart)
__device__ __forceinline__ int pa_loc_swz(int l) { return ((l & 1) << 3) | (((l >> 2) & 1) << 2); }
__device__ __forceinline__ float fast_tanh(float x) {   // as attention.hip
    const float e = __expf(2.f * x);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum8(float v) {
    v += dpp_get<0xB1>(v);
    v += dpp_get<0x4E>(v);
    v += dpp_get<0x141>(v);
    return v;
}
__device__ __forceinline__ float lane_bcast(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = sum8(v);
    v += dpp_get<0x140>(v);
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = fmaxf(v, dpp_get<0xB1>(v));
    v = fmaxf(v, dpp_get<0x4E>(v));
    v = fmaxf(v, dpp_get<0x141>(v));
    v = fmaxf(v, dpp_get<0x140>(v));
    return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
}

// One WAVE waits until every one of the n (<= 128) flag words (one per 128-byte line) reads >= target: two words per lane per look.  Bounded like
// handoff_wait<true>; returns false (wave-uniform) when the wait was given up, by a time-out here or anywhere else, or by the
// host's stop word.  (Two looks in flight half a round trip apart were measured: no gain - 22.07 vs 21.68 us per autoregressive step.)
__device__ __forceinline__ bool flags_wait(const unsigned* flags, int n, unsigned target, unsigned* tmo, unsigned code, unsigned limit,
                                           const unsigned* stop, bool nosleep) {
    if (limit == 0u) limit = HANDOFF_SPIN_LIMIT;
    const int lane = threadIdx.x & 63;
    const int i0 = (lane < n ? lane : 0) * 32, i1 = (lane + 64 < n ? lane + 64 : 0) * 32;   // (a flag per 128-byte line)
    unsigned spins = 0;
    while (true) {
        const unsigned v0 = __hip_atomic_load(flags + i0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned v1 = __hip_atomic_load(flags + i1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(v0 >= target && v1 >= target)) return true;
        if ((++spins & 127u) == 1u) {
            if (__hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (stop && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (spins > 16u * limit) {
                if (lane == 0) __hip_atomic_store(tmo, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        if (!nosleep) __builtin_amdgcn_s_sleep(2);
    }
}

}  // namespace

// SPG: query slabs per group of 32 threads (32 groups): 3 = the 96 slabs of the teacher-forced launch, 4 = the 128 of the
// autoregressive launches, 2 = the 64 of the launch beside the split kernel.
// SPLIT (128 < L <= 256): TWO workgroups per batch row, each with one half of the positions resident (a row of 256 positions
// is 512 KB of encoder memory: two CUs' worth of registers).  Per step each half computes its energies, its local maximum m,
// the local sum s of exp(e - m) and the unnormalised context c = sum exp(e - m) memory[l], and hands (m, s, c, the 16
// energies next to the cut) to its partner through global memory (write-through stores, one flag word per half, double
// buffered by step parity: guideline 16 again); both then form M = max(m0, m1), f_h = exp(m_h - M), S = s0 f0 + s1 f1 and
// the context (c0 f0 + c1 f1) / S with the operands in half order - bit-identical in both -, and normalise their own
// weights and those of the partner's 15 positions next to the cut (the halo of the location convolution) as
// exp(e - M) / S.  Half 0 publishes the context.  The softmax equals the reference's (models/tts/tacotron2.py:126) up to
// rounding (1e-7 relative).
#ifdef GVX_STAMPS
namespace { __device__ unsigned long long pa_row_stamps[64][8]; __device__ unsigned long long pa_row_stamps_ar[64][4]; __device__ unsigned long long pa_loc_stamps[4][8]; }
#define PA_LSTAMP(i) do { if (loc_stamp_on && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) % 5 == 0) pa_loc_stamps[(threadIdx.x >> 6) / 5][i] = wall_clock64(); } while (0)
#else
#define PA_LSTAMP(i) do { } while (0)   // per resident workgroup: the phase stamps of step 20
#endif
// AR (beside decoder_ar_resident_kernel, dec_resident.hip): after its context the row also finishes the step - it sums the 128
// projection slabs of the decoder-LSTM tiles into the frame + gate of the step (Decoder.decode's linear projection and gate
// layer, models/tts/tacotron2.py:360-362), runs the stop test (:401-406) and Prenet layer 1 on the frame (:176-179) and hands
// that to the 8 workgroups of layer 2.  Same arithmetic and summation orders as ar_project_fast_kernel (misc.hip).
template <int SPG, bool SPLIT, bool AR = false>
__global__ __launch_bounds__(PA_THREADS) void attn_persistent_kernel(AttnPersistParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* locf = smem + PA_OFF_LOC;
    float* fb = smem + PA_OFF_FB;
    float* cw = smem + PA_OFF_CW;
    float* wdl = smem + PA_OFF_WD;
    float* wc = smem + PA_OFF_WC;
    float* qp = smem + PA_OFF_QP;
    float* qs = smem + PA_OFF_QS;
    float* es = smem + PA_OFF_ES;
    float* vl = smem + PA_OFF_V;
    volatile int* leave = reinterpret_cast<volatile int*>(smem + PA_OFF_FLAG);

    const int b = SPLIT ? (int)(blockIdx.x >> 1) : (int)blockIdx.x, hf = SPLIT ? (int)(blockIdx.x & 1) : 0;
    const int B = p.B, kl = p.kl, pad = (kl - 1) / 2;
    const int l_base = PA_L * hf;                    // first position of this workgroup
    const int L = min(PA_L, p.L - l_base);           // positions held here (>= 1: the split kernel runs for p.L > PA_L only)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int len = (p.lengths ? p.lengths[b] : p.L) - l_base;   // valid positions among them (may be <= 0 in half 1)
    unsigned* const cnt_q = p.sync + HANDOFF_CNT_Q;
    unsigned* const cnt_ctx = p.sync + HANDOFF_CNT_CTX;
    unsigned* const tmo = p.sync + HANDOFF_TIMEOUT;
    const unsigned* const stop = p.sync + HANDOFF_STOP;
    if (tid == 0) *leave = 0;
    if (tid == 0) __hip_atomic_fetch_add(p.sync + HANDOFF_READY, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // resident

    // ---- resident operands
    // memory: thread (e4 = float4 column, lg = group of 16 positions) holds memory[b][16 lg + i][e4], i < 16.  The 8 position
    // groups of a column sit in 8 neighbouring lanes: the context's sum over them is three DPP adds, no LDS, no barrier
    // (autoregressive variants: the kernel sits at its 128-register limit and the allocator kept spilling one of these vectors - a
    // scratch round trip per step on the chain; the 16th one lives in LDS there, 16 KB the shrunken tables left free, read once per step)
    constexpr int NMR = AR ? 15 : 16;
    float4 mem[NMR];
    {
        const int e4 = tid >> 3, lg = tid & 7;
        const float4* mb = reinterpret_cast<const float4*>(p.memory) + ((long)b * p.L + l_base) * (PA_E / 4);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float4 v = mb[(long)min(16 * lg + i, L - 1) * (PA_E / 4) + e4];
            if (i < NMR) mem[i < NMR ? i : 0] = v;
            else reinterpret_cast<float4*>(smem + PA_OFF_MEML)[tid] = v;
        }
    }
    // processed memory: thread (l = position, dg) holds the float4 groups dg + 8 j of pm[b][l]
    float4 pmr[4];
    {
        const int el = tid >> 3, dg = tid & 7;
        const float4* pb = reinterpret_cast<const float4*>(p.pm) + ((long)b * p.L + l_base + min(el, L - 1)) * (PA_A / 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) pmr[j] = pb[dg + 8 * j];
    }
    for (int i = tid; i < 2 * 32 * 32; i += PA_THREADS) {   // conv weights [2][kl][32] -> [2][32][32], missing taps are zero
        const int ch = i >> 10, k = (i >> 5) & 31, f = i & 31;
        cw[i] = k < kl ? p.loc_conv_t[(ch * kl + k) * 32 + f] : 0.f;
    }
    for (int i = tid; i < 32 * PA_A; i += PA_THREADS) {   // global [32/4][a][4] -> [32][a]
        const int f = i / PA_A, dd = i - f * PA_A;
        wdl[i] = p.loc_dense_t[((f >> 2) * PA_A + dd) * 4 + (f & 3)];
    }
    if (tid < PA_A) vl[tid] = p.v[tid];
    if (AR) {   // projection bias and the row's stop state: read on the chain every step
        if (tid < 128) smem[PA_OFF_H1 + tid] = tid <= p.n_mels ? p.proj_b[tid] : 0.f;
        if (tid == 128) reinterpret_cast<int*>(smem + PA_OFF_H1)[128] = p.n_frames[b] != 0;
    }
    for (int i = tid; i < 2 * PA_WC_S; i += PA_THREADS) wc[i] = 0.f;
    __syncthreads();

    // location features of the next step from the previous / cumulative weights in wc, on the MFMA units:
    //   conv   C[f][l] = sum_kk Wc[f][kk] X[kk][l],  kk = 32 ch + k, X[kk][l] = wc[ch][l + k]   (2 x 8 tiles of 16 x 16, K = 64)
    //   dense  D[l][d] = sum_f  C[f][l] Wd[d][f]                                                (8 x 8 tiles of 16 x 16, K = 32)
    // both on v_mfma_f32_16x16x4_f32 over all 16 waves (lane mappings at the loops).  History of the phase: 9.5 us per step as VALU
    // loops, 6.9 on 32 x 32 x 2 tiles (the convolution on 4 waves), 4.0 now - see EXPERIMENTS.md, round 4.
#ifdef GVX_STAMPS
    bool loc_stamp_on = false;   // (stamps build: waves 0 / 5 / 10 / 15 of row 0 stamp the phases of the location features of step 20)
#endif
    auto location_features = [&]() {
        // (per-thread indices are recomputed from an opaque copy of the thread id in every phase of the step loop: hoisted out
        // of the loop they would sit in registers the resident operands need)
        int tq = tid;
        asm volatile("" : "+v"(tq));
        {   // conv on all 16 waves: wave -> (filter tile of 16 = wave & 1, position tile of 16 = wave >> 1), v_mfma_f32_16x16x4_f32:
            // A lane (i = lane & 15, kq = lane >> 4) gives A[i][kq], B lane (j, kq) gives B[kq][j], lane (j, g) receives D[4 g + r][j] in
            // register r.  (As four waves with 32 x 32 tiles the loop was a chain of LDS round trips - the kernel has no registers
            // to fetch ahead - and took 2.4 us; four waves per SIMD hide each other's.)
            using f32x4 = __attribute__((ext_vector_type(4))) float;
            const int li = tq & 15, kq = (tq >> 4) & 3;
            const int f0 = 16 * (wave & 1), l0 = 16 * (wave >> 1);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            // (operands of four products fetched together: left to itself the compiler waits for each LDS read right before its product;
            // two in the autoregressive variants, which have no registers for four - a resident memory vector was spilled, a scratch
            // round trip on the chain - and 10 us of slack for this phase)
            constexpr int CB = AR ? 2 : 4;
#pragma unroll
            for (int s0 = 0; s0 < 16; s0 += CB) {
                float av[CB], bv[CB];
#pragma unroll
                for (int u = 0; u < CB; ++u) {
                    const int kk = 4 * (s0 + u) + kq, ch = kk >> 5, k = kk & 31;
                    av[u] = cw[(ch * 32 + k) * 32 + f0 + li];
                    bv[u] = wc[ch * PA_WC_S + l0 + li + k];
                }
#pragma unroll
                for (int u = 0; u < CB; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) fb[(l0 + li) * PA_FB_S + f0 + 4 * kq + r] = acc[r];
        }
        PA_LSTAMP(0);
        __syncthreads();
        PA_LSTAMP(1);
        {   // dense: wave -> (32 positions l0 .., 32 dims d0 ..) as FOUR 16 x 16 tiles (v_mfma_f32_16x16x4_f32, 8 k-steps; a k-step's
            // 4 products share 2 + 2 operands).  POSITIONS are the tile's rows and dims its columns: lane (j, g) receives
            // D[position 4 g + r][dim j] in register r, so that 16 lanes write 16 consecutive dims of a position - with dims as rows
            // (and one 32 x 32 tile per wave) the float4 writes of 16 / 32 lanes went to the SAME banks of rows 128 floats apart:
            // 1.3 us of the phase's 3.6 were bank conflicts of these writes (stamps without them: profiles/r04_stamps_resident_final.txt)
            using f32x4 = __attribute__((ext_vector_type(4))) float;
            const int li = tq & 15, kq = (tq >> 4) & 3;
            const int d0 = 32 * (wave & 3), l0 = 32 * (wave >> 2);
            auto put = [&](const f32x4& v, int lt, int dt) {
                const int dim = d0 + 16 * dt + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int l = l0 + 16 * lt + 4 * kq + r;
                    locf[l * PA_A + 4 * ((dim >> 2) ^ pa_loc_swz(l)) + (dim & 3)] = v[r];
                }
            };
            if (!AR) {
                f32x4 a00 = {0.f, 0.f, 0.f, 0.f}, a01 = a00, a10 = a00, a11 = a00;   // [position tile][dim tile]
#pragma unroll 2
                for (int s = 0; s < 8; ++s) {
                    const int f = 4 * s + kq;
                    const float av0 = fb[(l0 + li) * PA_FB_S + f], av1 = fb[(l0 + 16 + li) * PA_FB_S + f];
                    const float bv0 = wdl[f * PA_A + d0 + li], bv1 = wdl[f * PA_A + d0 + 16 + li];
                    a00 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0, bv0, a00, 0, 0, 0);
                    a01 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0, bv1, a01, 0, 0, 0);
                    a10 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1, bv0, a10, 0, 0, 0);
                    a11 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1, bv1, a11, 0, 0, 0);
                }
                put(a00, 0, 0); put(a01, 0, 1); put(a10, 1, 0); put(a11, 1, 1);
            } else {
                // (autoregressive variants, as in the convolution above: two passes of two tiles each)
#pragma unroll 1
                for (int dt = 0; dt < 2; ++dt) {
                    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;   // [position tile]
#pragma unroll 1
                    for (int s = 0; s < 8; ++s) {
                        const int f = 4 * s + kq;
                        const float av0 = fb[(l0 + li) * PA_FB_S + f], av1 = fb[(l0 + 16 + li) * PA_FB_S + f];
                        const float bv = wdl[f * PA_A + d0 + 16 * dt + li];
                        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0, bv, a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1, bv, a1, 0, 0, 0);
                    }
                    put(a0, 0, dt); put(a1, 1, dt);
                }
            }
        }
        PA_LSTAMP(2);
        __syncthreads();
        PA_LSTAMP(3);
    };
    location_features();   // step 0: zero weights

    const __amdgpu_buffer_rsrc_t rq = make_rsrc(p.q_slab);
#ifdef GVX_STAMPS
#define PA_STAMP(i) do { if (t == 20) { GVX_STAMP(0, i); if (tid == 0 && (i) < 8) pa_row_stamps[blockIdx.x & 63][i] = wall_clock64(); } } while (0)
#define PA_ARSTAMP(i) do { if (t == 20 && tid == 0) pa_row_stamps_ar[blockIdx.x & 63][i] = wall_clock64(); } while (0)
#else
#define PA_STAMP(i) do { } while (0)
#define PA_ARSTAMP(i) do { } while (0)
#endif
    for (int t = 0; t < p.T; ++t) {
        PA_STAMP(0);
        // ---- query: the slabs of step t are plain stores of launch t; launch t + 1 (or the drain launch) adds 1 to the counter
        // when it STARTS, i.e. after launch t has completed and its stores have been written back (count t + 2: launch 0
        // adds too).  Then sum the partial slabs (fixed order)
        // A wait that returns without its word (the host has ended the loop early - autoregressive decode, every row has
        // stopped - or some wait has timed out) ends the kernel: nothing it could still compute would be used
        if (!p.q_flags) {
            if (tid == 0 && !handoff_wait<true>(cnt_q, (unsigned)t + p.q_first, tmo, 0x100u + (unsigned)b, p.spin_limit, stop)) *leave = 1;
            __syncthreads();
            if (*leave) break;
            PA_STAMP(1);
        }
        int tq = tid;
        asm volatile("" : "+v"(tq));
        const int lane = tq & 63, e4 = tq >> 3, lg = tq & 7, el = tq >> 3, dg = tq & 7;
        const bool cl = lg == 0;   // the lane of its column that stores / exchanges the context
        {
            const int d4 = tq & 31, sg = tq >> 5;   // slabs SPG sg .. SPG sg + SPG - 1 (one slab per workgroup / tile of the attention LSTM)
            // beside the resident decoder kernel: one flag per producing workgroup (value = steps published).  Every wave waits for the
            // producers of ITS slabs only (2 SPG of them) and fetches those at once: a row's slabs are 48 KB, 2 us of a CU's load path,
            // and the producers finish over 1.5 us - most of the bytes are in before the last flag goes up
            // (the wait's parameters from the kernel-argument segment, like the autoregressive role's below)
            typedef const __attribute__((address_space(4))) AttnPersistParams* KargPtr;
            KargPtr kq = (KargPtr)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(kq));
            if (p.q_flags && !flags_wait(kq->q_flags + ((b % RS_REP1) * kq->n_q_flags + 2 * SPG * wave) * 32, 2 * SPG, (unsigned)t + 1u, tmo, 0x100u + (unsigned)b,
                                         kq->spin_limit, stop, (kq->debug & 32) == 0) && (tid & 63) == 0) *leave = 1;
            if (p.q_flags) PA_STAMP(1);   // (wave 0's slabs seen)
            float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 ql[SPG];
#pragma unroll
            for (int i = 0; i < SPG; ++i) ql[i] = load_sc1(rq, (unsigned)(((SPG * sg + i) * B + b) * PA_A + 4 * d4) * 4u);
#pragma unroll
            for (int i = 0; i < SPG; ++i) { s4.x += ql[i].x; s4.y += ql[i].y; s4.z += ql[i].z; s4.w += ql[i].w; }
            // the wave's two slab groups (lanes l, l ^ 32) first: 16 partial rows instead of 32 for the second stage
            if (!SPLIT) { s4.x += __shfl_xor(s4.x, 32, 64); s4.y += __shfl_xor(s4.y, 32, 64);
            s4.z += __shfl_xor(s4.z, 32, 64); s4.w += __shfl_xor(s4.w, 32, 64);
            if ((tq & 32) == 0) reinterpret_cast<float4*>(qp)[(sg >> 1) * 32 + d4] = s4; } else reinterpret_cast<float4*>(qp)[sg * 32 + d4] = s4;
        }
        __syncthreads();
        if (*leave) break;   // (a wait that was given up: time-out, or the loop has ended)
        if (tq < PA_A) {
            float acc = qp[tq];
#pragma unroll
            for (int g = 1; g < (SPLIT ? 32 : 16); ++g) acc += qp[g * PA_A + tq];
            qs[tq] = acc;
        }
        __syncthreads();
        PA_STAMP(2);
        // ---- energies: e[l] = v . tanh(q + (loc[l] + pm[l]))   (8 lanes per position, DPP sum)
        {
            // Two elements per instruction wherever the ISA has a packed fp32 form (v_pk_add / mul / fma_f32): the phase was bound by
            // instruction issue - 8 full-rate and 2 quarter-rate (v_exp, v_rcp) instructions per element, 16 elements per thread,
            // 4 waves per SIMD: 1.64 us.  tanh(z) = 1 - 2 / (exp2(z 2 log2 e) + 1) as in fast_tanh, the same values bit for bit
            // (2 z and 2 r are exact); the two halves of a pair accumulate separately (the sum's order differs from the per-step kernel's).
            typedef float v2f __attribute__((ext_vector_type(2)));
            const v2f k2l = {2.885390043f, 2.885390043f};   // 2 log2(e)
            v2f pe2 = {0.f, 0.f};
            auto pair = [&](float qa, float qb, float la, float lb, float pa, float pb, float va, float vb) {
                v2f z = (v2f){qa, qb} + ((v2f){la, lb} + (v2f){pa, pb});
                z = z * k2l;
                v2f d = (v2f){__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)} + (v2f){1.f, 1.f};
                const v2f r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
                const v2f th = __builtin_elementwise_fma((v2f){-2.f, -2.f}, r, (v2f){1.f, 1.f});
                pe2 = __builtin_elementwise_fma((v2f){va, vb}, th, pe2);
            };
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int g4 = dg + 8 * j;
                const float4 qv = reinterpret_cast<const float4*>(qs)[g4];
                const float4 vv = reinterpret_cast<const float4*>(vl)[g4];
                const float4 lv = reinterpret_cast<const float4*>(locf)[el * (PA_A / 4) + (g4 ^ pa_loc_swz(el))];
                pair(qv.x, qv.y, lv.x, lv.y, pmr[j].x, pmr[j].y, vv.x, vv.y);
                pair(qv.z, qv.w, lv.z, lv.w, pmr[j].z, pmr[j].w, vv.z, vv.w);
            }
            float pe = pe2.x + pe2.y;
            pe = sum8(pe);
            if (dg == 0) es[el] = el < len ? pe : -INFINITY;
        }
        __syncthreads();
        PA_STAMP(3);
        // ---- masked softmax (every wave computes the normaliser) and this thread's share of the context
        float mx = -INFINITY;
        for (int l = lane; l < L; l += 64) mx = fmaxf(mx, es[l]);
        mx = wave_max_dpp(mx);
        // SPLIT: a half whose positions are all masked has mx = -inf; its local terms use reference 0 and come out as zeros
        const float mref = (SPLIT && mx == -INFINITY) ? 0.f : mx;
        // exp(e - m) of every position, once per wave: summed for the normaliser and left in a row of the wave's own in the query
        // partial-sum region (free since the query was summed), where the context loop below reads its 16 - as broadcasts, in place of
        // 16 more v_exp_f32 per thread (a quarter-rate instruction: 0.35 us of the phase)
        float* wex = qp + wave * PA_L;
        float sum = 0.f;
        for (int l = lane; l < PA_L; l += 64) {
            const float ex = l < L ? __expf(es[l] - mref) : 0.f;
            wex[l] = ex;
            sum += ex;
        }
        sum = wave_sum_dpp(sum);
        float inv = SPLIT ? 1.f : 1.f / sum;   // SPLIT: the context partials stay unnormalised until the halves have met
        float4 o_ctx;
        {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int l = 16 * lg + i;
                const float w = wex[l] * inv;   // exactly 0 past the row's length (exp(-inf)) and past the positions held
                const float4 mv = i < NMR ? mem[i < NMR ? i : 0] : reinterpret_cast<const float4*>(smem + PA_OFF_MEML)[tq];
                acc.x = fmaf(w, mv.x, acc.x); acc.y = fmaf(w, mv.y, acc.y);
                acc.z = fmaf(w, mv.z, acc.z); acc.w = fmaf(w, mv.w, acc.w);
                // (the split variant has no registers to spare: without this the scheduler computes all 16 weights first and
                // spills three of the resident memory vectors to make room - three scratch round trips per step on the chain)
            }
            acc.x = sum8(acc.x); acc.y = sum8(acc.y); acc.z = sum8(acc.z); acc.w = sum8(acc.w);   // over the column's 8 position groups
            o_ctx = acc;
        }
        if (!SPLIT && wave == 0) {   // alignment row out; previous / cumulative weights for the next location features
            for (int l = lane; l < L; l += 64) {
                const float w = __expf(es[l] - mx) * inv;
                p.w_out[(long)t * p.w_out_ts + (long)b * p.w_out_bs + l] = w;
                wc[pad + l] = w;
                wc[PA_WC_S + pad + l] += w;
            }
        }
        PA_STAMP(4);
        float4 o = o_ctx;
        if (SPLIT) {
            // (indices re-derived from a fresh opaque copy of the thread id: kept live from the top of the step they cost this
            // variant - 80 resident VGPRs - its last free registers)
            int tx = tid;
            asm volatile("" : "+v"(tx));
            const int e4x = tx >> 3;
            const bool clx = (tx & 7) == 0;
            // ---- hand (m, s, c, edge energies) to the partner half and take its; buffers alternate with the step's parity
            // (a half can be at most one publication ahead of what its partner has read)
            float* const xb_own = p.xchg + (((long)b * 2 + hf) * 2 + (t & 1)) * PA_XCH;
            const float* const xb_par = p.xchg + (((long)b * 2 + (hf ^ 1)) * 2 + (t & 1)) * PA_XCH;
            unsigned* const flag_own = p.sync + HANDOFF_PAIR + (b * 2 + hf) * 32;
            const unsigned* const flag_par = p.sync + HANDOFF_PAIR + (b * 2 + (hf ^ 1)) * 32;
            {
                const __amdgpu_buffer_rsrc_t ro = make_rsrc(xb_own);
                if (clx) store_sc1(ro, (unsigned)e4x * 16u, o);
                else if (tx == 129) store_sc1(ro, 512u * 4u, make_float4(mx, sum, 0.f, 0.f));
                else if (tx >= 132 && tx < 136) {   // the 16 energies next to the cut: the last 16 positions of half 0, the first 16 of half 1
                    const int i4 = tx - 132, e0 = hf == 0 ? PA_L - 16 : 0;
                    store_sc1(ro, (516u + 4u * (unsigned)i4) * 4u, *reinterpret_cast<const float4*>(es + e0 + 4 * i4));
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains before the flag goes up
            }
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_store(flag_own, (unsigned)t + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!handoff_wait<true>(flag_par, (unsigned)t + 1u, tmo, 0x400u + (unsigned)blockIdx.x, p.spin_limit, stop)) *leave = 1;
            }
            __syncthreads();
            if (*leave) break;
            const __amdgpu_buffer_rsrc_t rp = make_rsrc(xb_par);
            const float4 hd = load_sc1(rp, 512u * 4u);                          // partner's (m, s)
            const float4 cpar = clx ? load_sc1(rp, (unsigned)e4x * 16u) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float m0 = hf == 0 ? mx : hd.x, m1 = hf == 0 ? hd.x : mx;     // operands in half order: both halves compute the same bits
            const float s0 = hf == 0 ? sum : hd.y, s1 = hf == 0 ? hd.y : sum;
            const float M = fmaxf(m0, m1);                                      // finite: position 0 of half 0 is never masked
            const float f0 = m0 == -INFINITY ? 0.f : __expf(m0 - M), f1 = m1 == -INFINITY ? 0.f : __expf(m1 - M);
            inv = 1.f / (s0 * f0 + s1 * f1);
            if (clx) {
                const float4 c0 = hf == 0 ? o : cpar, c1 = hf == 0 ? cpar : o;
                o.x = (c0.x * f0 + c1.x * f1) * inv; o.y = (c0.y * f0 + c1.y * f1) * inv;
                o.z = (c0.z * f0 + c1.z * f1) * inv; o.w = (c0.w * f0 + c1.w * f1) * inv;
            }
            if (wave == 0) {   // alignment row out; previous / cumulative weights (own positions + the partner's `pad` next to the cut)
                for (int l = lane; l < L; l += 64) {
                    const float w = __expf(es[l] - M) * inv;
                    p.w_out[(long)t * p.w_out_ts + (long)b * p.w_out_bs + l_base + l] = w;
                    wc[pad + l] = w;
                    wc[PA_WC_S + pad + l] += w;
                }
                if (lane < pad) {
                    // half 0: halo = local positions 128 .. 128 + pad - 1 = the partner's first `pad` energies (edge[i]);
                    // half 1: halo = local positions -pad .. -1 = the partner's last `pad` (edge[16 - pad + i])
                    const float e = load_sc1_f32(rp, (516u + (unsigned)(hf == 0 ? lane : 16 - pad + lane)) * 4u);
                    const float w = __expf(e - M) * inv;
                    const int idx = hf == 0 ? pad + PA_L + lane : lane;
                    wc[idx] = w;
                    wc[PA_WC_S + idx] += w;
                }
            }
        }
        if (cl && hf == 0) {
            // blocked context vector [E/8][B][8] of step t, write-through
            const __amdgpu_buffer_rsrc_t rc = make_rsrc(p.ctx_base + (long)t * p.ctx_ts);
            store_sc1(rc, (unsigned)((e4 >> 1) * B * 8 + b * 8 + 4 * (e4 & 1)) * 4u, o);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (p.ctx_flags) {   // one wave instruction: lane r stores the row's flag into replica r
            int tf = tid;   // (address from an opaque copy of the thread id: hoisted out of the loop it was spilled and reloaded here)
            asm volatile("" : "+v"(tf));
            if (tf < RS_REP1 && hf == 0) __hip_atomic_store(p.ctx_flags + (tf * 32 + b) * 32, (unsigned)t + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (tid == 0 && hf == 0) __hip_atomic_fetch_add(cnt_ctx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        PA_STAMP(5);
#ifdef GVX_STAMPS
        if (t == 20 && tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pa_row_stamps[blockIdx.x & 63][7] = wall_clock64(); }   // flag store acknowledged
#endif
        // ---- off the chain: location features of step t + 1 (autoregressive role: after the row's second part of the step, which IS on
        // the chain - the features have until the next query arrives, two hand-offs and a cell later)
#ifdef GVX_STAMPS
        loc_stamp_on = t == 20 && blockIdx.x == 0;
#endif
        if (!AR && t + 1 < p.T) location_features();
        PA_STAMP(6);
        if (AR && hf == 0) {   // (a row of two halves: half 0, which has published the context, finishes the step)
            int ta = tid;
            asm volatile("" : "+v"(ta));
            // (the role's parameters are read from the kernel-argument segment every step: kept in scalar registers for the whole
            // loop they pushed other scalars into lanes of vector registers, and one of the resident memory vectors into scratch)
            typedef const __attribute__((address_space(4))) AttnPersistParams* KargPtr;
            KargPtr kp = (KargPtr)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(kp));
            const bool more = t + 1 < p.T;   // (the last step: nobody consumes a next Prenet input)
            const int M = kp->n_mels, PSB = kp->PSB;
            float4* sp4 = reinterpret_cast<float4*>(qp);                  // [32][PA_SP_S]
            float4* l1p = reinterpret_cast<float4*>(smem + PA_OFF_CP);    // [16][64]
            float* mel = qs;                                              // [128]
            const float* pbl = smem + PA_OFF_H1;                          // [128] projection bias
            int* fired = reinterpret_cast<int*>(smem + PA_OFF_H1) + 128;
            // next step's keep byte of layer 1 (one register across the wait; its round trip would sit on the chain)
            unsigned char k0 = 0;
            if (more && ta < PA_P) k0 = kp->keep0[((long)(t + 1) * B + b) * PA_P + ta];
            // ---- the slabs of the step: wave w takes those of decoder-LSTM workgroups 8 w .. 8 w + 7 as soon as THEIR flags are up (45 KB per
            // row: the same reasoning as for the query slabs), lanes (g, n4): slabs 8 w + 4 g .. + 3, float4 column n4
            if (!flags_wait(kp->p_flags + ((b % RS_REP_P) * 128 + 8 * wave) * 32, 8, (unsigned)t + 1u, tmo, 0x600u + (unsigned)b, kp->spin_limit, stop, (kp->debug & 32) == 0) && (tid & 63) == 0) *leave = 1;
            PA_ARSTAMP(0);   // projection slabs of wave 0 seen
            const int n4c = PSB >> 2;                 // float4 per slab row (<= 24)
            {
                const int la = ta & 63, g = la >= n4c ? 1 : 0, n4 = la - g * n4c;
                if (la < 2 * n4c) {
                    const __amdgpu_buffer_rsrc_t rs = make_rsrc(kp->p_slab);
                    float4 sv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) sv[q] = load_sc1(rs, (unsigned)(((8 * wave + 4 * g + q) * B + b) * PSB + 4 * n4) * 4u);
                    float4 acc = sv[0];
                    acc.x += sv[1].x; acc.y += sv[1].y; acc.z += sv[1].z; acc.w += sv[1].w;
                    acc.x += sv[2].x; acc.y += sv[2].y; acc.z += sv[2].z; acc.w += sv[2].w;
                    acc.x += sv[3].x; acc.y += sv[3].y; acc.z += sv[3].z; acc.w += sv[3].w;
                    sp4[(2 * wave + g) * PA_SP_S + n4] = acc;
                }
            }
            __syncthreads();
            if (*leave) break;
            // this thread's layer-1 weights (L2 hits, the same for every row and step): requested here, their round trip overlaps the
            // column sums (the kernel has no 20 registers to hold them across the wait)
            const int kq1 = ta >> 6, j4 = ta & 63;    // layer 1: 16 k slices x 64 float4 columns
            float4 w0v[PA_KPT];
            {
                const float4* w0 = reinterpret_cast<const float4*>(kp->pre_w0_t) + j4;
#pragma unroll
                for (int i = 0; i < PA_KPT; ++i) w0v[i] = w0[min(kq1 * PA_KPT + i, M - 1) * (PA_P / 4)];
            }
            {   // column c4 (4 projection outputs) summed over the 32 slab groups by the 32 lanes of a half wave: no serial chain of LDS reads
                const int c4 = ta >> 5, sgl = ta & 31;
                if (c4 < n4c) {   // (uniform per half wave)
                    float4 v4 = sp4[sgl * PA_SP_S + c4];   // (odd row stride: the half wave's 32 rows of one column in different banks)
#pragma unroll
                    for (int o = 16; o >= 1; o >>= 1) {
                        v4.x += __shfl_xor(v4.x, o, 64); v4.y += __shfl_xor(v4.y, o, 64);
                        v4.z += __shfl_xor(v4.z, o, 64); v4.w += __shfl_xor(v4.w, o, 64);
                    }
                    if (sgl == 0) {
                        float* vp = &v4.x;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int n = 4 * c4 + i;
                            vp[i] = n <= M ? vp[i] + pbl[n] : 0.f;
                        }
                        if (c4 == (M >> 2) && *fired == 0) {   // stop token (models/tts/tacotron2.py:401-406)
                            const int gi = M & 3;
                            const float gate = gi == 0 ? v4.x : (gi == 1 ? v4.y : (gi == 2 ? v4.z : v4.w));
                            const float sgm = 1.f / (1.f + expf(-gate));
                            if (sgm > kp->gate_threshold) {
                                *fired = 1;
                                kp->n_frames[b] = t + 1;
                                atomicAdd(kp->n_done, 1);
                            }
                        }
                        // blocked vector [PSB/8][B][8] of the step
                        *reinterpret_cast<float4*>(kp->proj_out + (long)t * B * PSB + (long)(c4 >> 1) * B * 8 + b * 8 + 4 * (c4 & 1)) = v4;
#pragma unroll
                        for (int i = 0; i < 4; ++i) mel[4 * c4 + i] = 4 * c4 + i < M ? vp[i] : 0.f;   // zeros past the mel bins: clamped weight loads contribute nothing
                    }
                } else if (ta < 32 * n4c + 128 - 4 * n4c) {
                    mel[4 * n4c + (ta - 32 * n4c)] = 0.f;   // (columns past PSB of the 128-float frame buffer)
                }
            }
            __syncthreads();
            PA_ARSTAMP(1);   // frame summed
            if (more) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < PA_KPT; ++i) {
                    const float x = mel[kq1 * PA_KPT + i];
                    acc.x = fmaf(w0v[i].x, x, acc.x); acc.y = fmaf(w0v[i].y, x, acc.y);
                    acc.z = fmaf(w0v[i].z, x, acc.z); acc.w = fmaf(w0v[i].w, x, acc.w);
                }
                l1p[kq1 * 64 + j4] = acc;
            }
            __syncthreads();
            if (ta < PA_P) {   // (waves 0 - 3)
                float r = 0.f;
                if (more) {
                    const float* col = reinterpret_cast<const float*>(l1p) + ta;
                    float acc = col[0];
#pragma unroll
                    for (int q = 1; q < 16; ++q) acc += col[q * 256];
                    acc = fmaxf(acc, 0.f);
                    r = k0 ? 2.f * acc : 0.f;   // dropout p = 0.5 also at inference time (models/tts/tacotron2.py:178)
                }
                // four neighbours' values -> one 16-byte write-through piece of the blocked vector [P/8][B][8]
                const int l0 = (ta & 63) & ~3;
                const float4 o = make_float4(__shfl(r, l0, 64), __shfl(r, l0 + 1, 64), __shfl(r, l0 + 2, 64), __shfl(r, l0 + 3, 64));
                if (more && (ta & 3) == 0) {
                    const __amdgpu_buffer_rsrc_t ry = make_rsrc(kp->y1);
                    store_sc1(ry, (unsigned)((ta >> 3) * B * 8 + b * 8 + (ta & 7)) * 4u, o);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // (the wave that ran the stop test: its counter update has arrived)
            if (wave == (M >> 2) / 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            PA_ARSTAMP(2);   // layer 1 stored
            if (ta == 0) __hip_atomic_store(kp->y1_flags + b * 32, (unsigned)t + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (AR && t + 1 < p.T) location_features();
    }
}

// The resident kernel may hold 64 of the 256 CUs at most:
//   layout 1: B <= 32, L <= 128        one workgroup per row (<= 32 CUs), launches of 224 workgroups (skinny.hip)
//   layout 2: B <= 32, 128 < L <= 256  two workgroups per row (<= 64 CUs), launches of 192 workgroups
//   layout 3: 32 < B <= 64, L <= 128   one workgroup per row (<= 64 CUs), launches of 384 workgroups with two batch tiles
//                                      each, two per CU (gvx_api.hip, decoder_tf_impl)
int attention_persistent_layout(int B, int L) {
    if (B < 1 || L < 1) return 0;
    if (B <= 32) return L <= PA_L ? 1 : (L <= 2 * PA_L ? 2 : 0);
    return B <= 64 && L <= PA_L ? 3 : 0;
}
bool attention_persistent_supported(int B, int L, int a, int F, int kl, int E, int att_rnn_dim, int dec_rnn_dim) {
    // default layer sizes only: the launch layouts deal 128 + 128 tiles of the two cells
    return attention_persistent_layout(B, L) != 0 && a == PA_A && E == PA_E && F >= 1 && F <= 32 && kl >= 1 && kl <= PA_KL_MAX &&
           (kl & 1) && att_rnn_dim == 1024 && dec_rnn_dim == 1024;
}
int attention_persistent_slabs(int layout) { return layout == 2 ? 64 : (layout == 3 ? PA_SLABS_AR : PA_SLABS); }
int attention_persistent_workgroups(int B, int L) { return attention_persistent_layout(B, L) == 2 ? 2 * B : B; }
size_t attention_persistent_xchg_floats(int B) { return (size_t)B * 2 * 2 * PA_XCH; }

template <typename K>
static hipError_t pa_set_lds(K kern) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, PA_LDS_FLOATS * (int)sizeof(float));
}
hipError_t attention_persistent_init() {
    hipError_t e;
    if ((e = pa_set_lds(attn_persistent_kernel<3, false>)) != hipSuccess) return e;
    if ((e = pa_set_lds(attn_persistent_kernel<4, false>)) != hipSuccess) return e;
    if ((e = pa_set_lds(attn_persistent_kernel<3, false, true>)) != hipSuccess) return e;
    if ((e = pa_set_lds(attn_persistent_kernel<3, true, true>)) != hipSuccess) return e;
    if ((e = pa_set_lds(attn_persistent_kernel<3, true>)) != hipSuccess) return e;
    return pa_set_lds(attn_persistent_kernel<2, true>);
}

__global__ void handoff_set_kernel(unsigned* word) { __hip_atomic_store(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
hipError_t launch_handoff_set(unsigned* word, hipStream_t s) {
    handoff_set_kernel<<<dim3(1), dim3(1), 0, s>>>(word);
    return hipGetLastError();
}

#ifdef GVX_STAMPS
hipError_t read_loc_stamps_persist(unsigned long long* host32) {
    return hipMemcpyFromSymbol(host32, HIP_SYMBOL(pa_loc_stamps), sizeof(unsigned long long) * 32);
}
hipError_t read_row_stamps_persist_ar(unsigned long long* host256) {
    return hipMemcpyFromSymbol(host256, HIP_SYMBOL(pa_row_stamps_ar), sizeof(unsigned long long) * 256);
}
hipError_t read_row_stamps_persist(unsigned long long* host512) {
    return hipMemcpyFromSymbol(host512, HIP_SYMBOL(pa_row_stamps), sizeof(unsigned long long) * 512);
}
hipError_t read_stamps_persist(unsigned long long* host96) {
    return hipMemcpyFromSymbol(host96, HIP_SYMBOL(gvx_stamps), sizeof(unsigned long long) * 96);
}
#endif

hipError_t launch_attention_persistent(const AttnPersistParams& p, hipStream_t s) {
    if (!attention_persistent_supported(p.B, p.L, PA_A, 32, p.kl, PA_E, 1024, 1024) || p.T < 1) return hipErrorInvalidValue;
    const size_t lds = PA_LDS_FLOATS * sizeof(float);
    if (p.p_slab) {   // autoregressive role beside decoder_ar_resident_kernel (224-workgroup deal: 96 slabs; rows of 129-256 tokens: two
                      // workgroups per row, which leaves room for 16 rows beside the 224)
        if (p.n_slabs != PA_SLABS || !p.q_flags || !p.ctx_flags || !p.p_flags || !p.y1_flags || !p.proj_b || !p.proj_out || !p.pre_w0_t || !p.keep0 ||
            !p.y1 || !p.n_frames || !p.n_done || p.n_mels < 1 || p.n_mels > 16 * PA_KPT || p.PSB < p.n_mels + 1 || p.PSB > 96 || (p.PSB & 3))
            return hipErrorInvalidValue;
        if (p.L > PA_L) {
            if (!p.xchg || p.B > 16) return hipErrorInvalidValue;
            attn_persistent_kernel<3, true, true><<<dim3(2 * p.B), dim3(PA_THREADS), lds, s>>>(p);
        } else attn_persistent_kernel<3, false, true><<<dim3(p.B), dim3(PA_THREADS), lds, s>>>(p);
    } else if (p.L > PA_L) {
        if (!p.xchg) return hipErrorInvalidValue;
        if (p.n_slabs == 64) attn_persistent_kernel<2, true><<<dim3(2 * p.B), dim3(PA_THREADS), lds, s>>>(p);
        else if (p.n_slabs == PA_SLABS && p.B <= 16) attn_persistent_kernel<3, true><<<dim3(2 * p.B), dim3(PA_THREADS), lds, s>>>(p);   // (beside the 224-workgroup tile kernel)
        else return hipErrorInvalidValue;
    } else if (p.n_slabs == PA_SLABS) attn_persistent_kernel<3, false><<<dim3(p.B), dim3(PA_THREADS), lds, s>>>(p);
    else if (p.n_slabs == PA_SLABS_AR) attn_persistent_kernel<4, false><<<dim3(p.B), dim3(PA_THREADS), lds, s>>>(p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace gvx
