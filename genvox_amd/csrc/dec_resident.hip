// The decoder loops as ONE resident, WEIGHT-STATIONARY kernel each, beside the resident attention kernel (attn_persist.hip):
//   decoder_resident_kernel     teacher-forced (models/tts/tacotron2.py:365-388 Decoder.forward, :333-363 Decoder.decode): both LSTM
//                               cells of every step; 224-workgroup deal (below) or 192 (pairs of attention-LSTM tiles, rows of 129-256 tokens)
//   decoder_ar_resident_kernel  autoregressive (:390-413 Decoder.inference): the same engine with the frame feedback - the attention
//                               LSTM's Prenet columns, the projection as slabs of the decoder-LSTM tiles, Prenet layer 2; comment at the kernel
// Both run batches of one or two rows on the vector ALUs instead of the matrix units (RS_MUL32; one body, two instantiations).
//
// Why: the two cells' recurrent matrices are 65.5 MB of fp32 and every decoder step multiplies all of them with the 32 batch
// rows' vectors.  Streamed per step (skinny.hip, one launch per step) that is 18.3 us per step: a CU's load path moves ~25 KB/us
// whatever shares it (13 us for the 292 KB a CU gets) and every launch pays ~5 us of first-byte latency and tail.  Streaming the
// same bytes from a resident kernel (LDS-DMA loader ring, first version of this file, round 4) did not help: the stream cap then
// sits ON the step's dependency chain (24 us per step, profiles/r04_stamps_resident_ring.txt).  But 224 CUs hold 35 MB of LDS and
// 114 MB of vector registers: the matrices FIT ON CHIP.  Here every workgroup loads its tile's weight fragments ONCE per call -
// the context columns and the first k-groups of the other columns into registers (up to 120 VGPRs per lane), the rest into LDS
// (up to 120 KB) - and then runs all T steps out of them: no weight byte moves after the prologue, a step is its MFMAs (7.3 us of
// fp32 matrix pipe per CU at 32 rows), the x fragments out of L2 and the hand-offs.
//
// Same deal, tiles, K slices per wave and summation order as the launch-per-step kernel (skinny.hip, 224-workgroup layout), so the
// results are bit-identical to it:
//   blocks [0, 64):   attention-LSTM tile + half of a neighbour's rows (288 KB of weights: 120 VGPRs + 108 KB of LDS)
//   blocks [64, 96):  attention-LSTM tiles 96 .. 127                   (192 KB: 96 VGPRs)
//   blocks [96, 224): decoder-LSTM tiles                                (320 KB: 100 VGPRs + 120 KB of LDS)
// The decoder cell is off the step's chain (only the projection after the loop reads h_d): its workgroups free-run behind the
// attention-LSTM workgroups, gated by the same flags, at most RS_HA_SLOTS - 1 steps behind (the h_a ring; its writers check at the
// top of a step that the slot's last reader has finished).
//
// Hand-offs (cdna_hip_programming.md guideline 16, MI355X_MICROARCH.md "Valid forms", first table row): every handed-off byte is
// stored `sc1` in 16-byte pieces, every storing wave drains (s_waitcnt vmcnt(0)), the waves meet at a barrier, ONE lane stores the
// workgroup's flag (`sc1`, value = steps published); a reader polls every producer's flag (`sc1` loads, ONE wave per workgroup:
// whichever wave blocks first takes an LDS lock and polls - its own class and those other waves have left in the want mask, in one
// round trip), keeps "steps everybody has published" per producer class in LDS words behind the matched poll, and every load of
// the bytes is an `sc1` buffer load issued behind such a word.  Flags exist in replicas on lines of their own (gvx_kernels.h).
// Every wait is bounded: polls without progress beyond the limit raise the call's status word and the workgroup's abort word, every
// wait then returns at once, the grid drains and the caller's poison launch overwrites the outputs (gvx_api.hip).
#include "gvx_kernels.h"

namespace gvx {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int RS_WAVES = 8;
constexpr int RS_THREADS = RS_WAVES * 64;
constexpr int RS_A = 1024, RS_D = 1024, RS_E = 512, RS_P = 256, RS_ATT = 128;   // default layer sizes (attention_persistent_supported)
constexpr int RS_NKGW_ATT = (RS_P + RS_E + RS_A) / 8;   // k-groups per tile of the packed attention-LSTM matrix (224)
constexpr int RS_NKGW_DEC = (RS_A + RS_E + RS_D) / 8;   // ... of the decoder-LSTM matrix (320)
constexpr int RS_KG0_ATT = RS_P / 8;                    // the Prenet columns are applied before the loop (pre_gate)

// Per kind of workgroup (0: attention-LSTM tile + half tile, 1: attention-LSTM tile, 2: decoder-LSTM tile): k-groups per wave and
// step in the part whose x is known early (NN: h_a / h_d columns) and in the context part (NC), how many of the former live in
// registers (NRN; the context part always does), and the LDS layout (bytes)
template <int KIND> struct RsCfg {
    static constexpr bool XH = KIND == 0, ATT = KIND != 2;
    static constexpr int RT = KIND == 3 ? 2 : 1;                           // row tiles per workgroup (KIND 3: a pair of attention-LSTM tiles)
    static constexpr int NN = ATT ? 16 : 32, NC = 8;
    static constexpr int NRN = KIND == 0 ? 7 : (KIND == 1 ? 16 : (KIND == 2 ? 17 : 9));
    static constexpr int NLN = NN - NRN;                                   // k-groups per wave in LDS
    static constexpr int WAVE_W = NLN * (XH ? 1536 : 1024 * RT);          // LDS bytes of a wave's fragments (+ 512 per half fragment)
    static constexpr int OFF_RED = RS_WAVES * WAVE_W;                      // [8 waves][16][64] floats (KIND 3: one tile after the other)
    static constexpr int OFF_RED2 = OFF_RED + RS_WAVES * 16 * 64 * 4;      // [8][8][64] floats (half tile)
    static constexpr int OFF_HS = OFF_RED2 + (XH ? RS_WAVES * 8 * 64 * 4 : 0);   // [32][8 RT] h' of the workgroup's units
    static constexpr int OFF_HS2 = OFF_HS + 32 * 8 * 4 * RT;               // [32][4] h' of the half tile's units
    static constexpr int OFF_CTRL = OFF_HS2 + 32 * 4 * 4;                  // control words
    static constexpr int LDS_BYTES = OFF_CTRL + 256;
};
constexpr int RS_LDS_BYTES = RsCfg<0>::LDS_BYTES > RsCfg<2>::LDS_BYTES ? RsCfg<0>::LDS_BYTES : RsCfg<2>::LDS_BYTES;
static_assert(RS_LDS_BYTES <= 160 * 1024 && RsCfg<1>::LDS_BYTES <= RS_LDS_BYTES && RsCfg<3>::LDS_BYTES <= RS_LDS_BYTES, "resident decoder LDS");
// control words (int index)
constexpr int RC_HA = 0;           // steps whose h_a + slabs every attention-LSTM workgroup has published
constexpr int RC_CTX = 1;          // steps whose context every attention row has published
constexpr int RC_HD = 2;           // steps whose h_d every decoder-LSTM workgroup has published
constexpr int RC_ABORT = 3;        // != 0: a wait of this workgroup has given up - every wait returns at once, the workgroup drains
constexpr int RC_LOCK = 4;         // the wave that holds it polls the global flags for the workgroup
constexpr int RC_IDLE = 5;         // polls since the last one that moved a word (the bound of the waits)
constexpr int RC_PRE = 6;          // autoregressive loop: steps whose Prenet output every Prenet workgroup has published (value t: input of step t)
constexpr int RC_Y1 = 7;           // ... whose Prenet layer 1 every attention row has published
constexpr int RC_EXIT = 8;         // [2] by step parity: "the workgroup leaves after this step" (written before the step's last barrier)
constexpr int RC_WANT = 10;        // classes (bit = control word) that waves without the lock are waiting for

// Everything the two kernels of this file take (the launchers fill it from DecResidentParams / ArResidentParams)
struct RsArgs {
    const float* att_frag; const float* att_bias; const float* wq_t; const float* dec_frag; const float* dec_bias;
    const float* pre_gate;                                          // teacher-forced loop only
    const float* proj_hd_t; const float* proj_ctx_t; const float* pre_w1; const uint8_t* keep1;   // autoregressive loop only ...
    float* prenet; const float* y1; float* p_slab; const int32_t* n_done; int PSB;
    float* h_a; float* hc; float* q_slab; float* c_a; float* c_d;
    // training-mode tape of the teacher-forced loop (TR instantiations only): keep masks of the dropout on both cells' outputs
    // [T][B][H], cell states [T+1][B][H], gate pre-activations [T][B][H][4]; h_a then points at the tape [T+1][A/8][B][8]
    const uint8_t* tr_keep_a; const uint8_t* tr_keep_d; float* tr_c_a; float* tr_c_d; float* tr_pre_a; float* tr_pre_d;
    float tr_scale_a, tr_scale_d;
    unsigned* sync;
    unsigned att_frag_bytes;
    int B, T;
    unsigned spin_limit;
    int debug, layout;
};

#ifdef GVX_STAMPS
// diagnostic build (tools/stamps_resident.py): wall-clock stamps (10 ns) of decoder step RS_STAMP_T in one workgroup of each kind
// (blocks 0 / 64 / 96): [kind][wave 0-7][event]; per workgroup (wave 0): step begins, gate 1, gate 2, flag stored
constexpr int RS_STAMP_T = 20;
__device__ unsigned long long rs_stamps[4][10][16];
__device__ unsigned long long rs_wg_stamps[224][4];
__device__ unsigned long long rs_wg_stamps_ar[224][4];   // autoregressive loop: y1 seen, Prenet flag stored, Prenet seen, Prenet columns done
#define RS_STAMP(ev) do { if (stamp_wg && t == RS_STAMP_T && lane == 0) rs_stamps[KIND][wave][ev] = wall_clock64(); } while (0)
#define RS_WGSTAMP(i) do { if (t == RS_STAMP_T && tid == 0) rs_wg_stamps[bid][i] = wall_clock64(); } while (0)
#define RS_ARSTAMP(i) do { if (t == RS_STAMP_T && tid == 0) rs_wg_stamps_ar[bid][i] = wall_clock64(); } while (0)
#else
#define RS_STAMP(ev) do { } while (0)
#define RS_WGSTAMP(i) do { } while (0)
#define RS_ARSTAMP(i) do { } while (0)
#endif

__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f); }

struct RsPoll {   // what a poll needs: the flag words, the status word, sizes
    const unsigned* f_att; const unsigned* f_dec; const unsigned* f_ctx; unsigned* tmo;
    const unsigned* f_pre; const unsigned* f_y1; const unsigned* stop;   // autoregressive loop (nullptr otherwise)
    unsigned limit; int B, T, bid;
    int n_att;   // attention-LSTM workgroups (96 in the 224-workgroup deal, 64 pairs in the 192-workgroup one)
    int sleep;   // s_sleep units between two looks of the polling wave (GVX_RS_DEBUG experiments)
};

// One look at the flags of the producer classes in `mask` (bit = control word: the class the caller waits for and those other waves
// of the workgroup have asked for - all in ONE round trip) by the wave that holds the lock; "steps everybody has published" per
// class by ballots (a count only moves up).  Every 16th fruitless look also reads the call's status word.  Gives up - status word,
// abort word - after `limit` looks in a row that moved nothing, or when anybody else has.
__device__ __forceinline__ bool rs_count(int* ctrl, int word, unsigned v, unsigned Tu, int lane) {
    const unsigned s0 = (unsigned)__hip_atomic_load(ctrl + word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    unsigned n = s0;
    while (n < Tu && __all(v > n)) ++n;
    if (lane == 0 && n != s0) __hip_atomic_store(ctrl + word, (int)n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    return n != s0;
}
__device__ __forceinline__ void rs_poll_once(int* ctrl, const RsPoll& q, int lane, int mask) {
    constexpr unsigned none = 0xffffffffu;
    unsigned va = none, vd = none, vc = none, vp = none, vy = none;
    if (mask & (1 << RC_HA)) {
        const unsigned a0 = __hip_atomic_load(q.f_att + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned a1 = lane + 64 < q.n_att ? __hip_atomic_load(q.f_att + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : none;
        va = min(a0, a1);
    }
    if (mask & (1 << RC_HD)) {
        const unsigned d0 = __hip_atomic_load(q.f_dec + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned d1 = __hip_atomic_load(q.f_dec + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        vd = min(d0, d1);
    }
    if (mask & (1 << RC_CTX)) vc = lane < q.B ? __hip_atomic_load(q.f_ctx + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : none;   // (a flag per 128-byte line)
    if (q.f_pre) {   // (autoregressive loop)
        if (mask & (1 << RC_PRE)) vp = lane < 8 ? __hip_atomic_load(q.f_pre + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : none;
        if (mask & (1 << RC_Y1)) vy = lane < q.B ? __hip_atomic_load(q.f_y1 + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : none;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no instruction: nothing below moves above the loads
    const unsigned Tu = (unsigned)q.T;
    bool moved = false;
    if (mask & (1 << RC_HA)) moved |= rs_count(ctrl, RC_HA, va, Tu, lane);
    if (mask & (1 << RC_HD)) moved |= rs_count(ctrl, RC_HD, vd, Tu, lane);
    if (mask & (1 << RC_CTX)) moved |= rs_count(ctrl, RC_CTX, vc, Tu, lane);
    if (q.f_pre) {
        if (mask & (1 << RC_PRE)) moved |= rs_count(ctrl, RC_PRE, vp, Tu, lane);
        if (mask & (1 << RC_Y1)) moved |= rs_count(ctrl, RC_Y1, vy, Tu, lane);
    }
    int idle = 0;
    if (lane == 0) {
        idle = moved ? 0 : __hip_atomic_load(ctrl + RC_IDLE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1;
        __hip_atomic_store(ctrl + RC_IDLE, idle, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    idle = __builtin_amdgcn_readfirstlane(idle);
    if (idle == 0 || (idle & 15) != 0) return;
    bool give_up = __hip_atomic_load(q.tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;   // somebody else has timed out (wave-uniform address)
    // (autoregressive loop: every row has fired its stop token - the loop is over, the waits end like those of a time-out, but without one)
    if (q.stop && __hip_atomic_load(q.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) give_up = true;
    if ((unsigned)idle > q.limit) {
        if (lane == 0) __hip_atomic_store(q.tmo, 0x500u + (unsigned)(q.bid & 0xff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        give_up = true;
    }
    if (give_up && lane == 0) __hip_atomic_store(ctrl + RC_ABORT, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Gate of a wave's x loads: wait until the LDS word reads >= need.  A wave that has to wait polls for the whole workgroup if nobody
// else is doing so already; otherwise it leaves the class it waits for in the workgroup's want mask, which the polling wave
// serves with its own (waves of a decoder-LSTM workgroup wait for two classes at once: without the mask the wave that polled for
// the contexts kept the lock and the h_a flags went unread for microseconds).  Returns at once when the abort word is up.
__device__ __forceinline__ void rs_gate(int* ctrl, int word, int need, const RsPoll& q, int lane) {
    while (__hip_atomic_load(ctrl + word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
        if (__hip_atomic_load(ctrl + RC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return;
        int got = 0;
        if (lane == 0) {
            int expect = 0;
            got = __hip_atomic_compare_exchange_strong(ctrl + RC_LOCK, &expect, 1, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1 : 0;
            if (got) got = __hip_atomic_exchange(ctrl + RC_WANT, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) | (1 << word);
            else __hip_atomic_fetch_or(ctrl + RC_WANT, 1 << word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        got = __builtin_amdgcn_readfirstlane(got);
        if (got) {
            rs_poll_once(ctrl, q, lane, got);
            if (lane == 0) __hip_atomic_store(ctrl + RC_LOCK, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (q.sleep == 8) __builtin_amdgcn_s_sleep(8);
            else if (q.sleep == 32) __builtin_amdgcn_s_sleep(32);
        } else {
            __builtin_amdgcn_s_sleep(1);
        }
    }
}
// Barrier of the workgroup's 8 waves.  All of them reach every barrier of every step also after an abort (waits return at once,
// nobody leaves the loop early), so the hardware barrier is safe; LDS traffic of the wave is retired first (the barrier itself
// waits for no counter), vector memory is left alone (an earlier version - a counter in LDS polled with s_sleep - cost
// 0.1-0.35 us per barrier, three per step on the chain).
__device__ __forceinline__ void rs_cbar() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// one LDS-DMA piece: 64 lanes x 16 bytes from rsrc + voff + soff to LDS lds_addr + 16 lane (M0: written in the same statement,
// restored afterwards - the compiler owns it; s_nop 0: M0 write -> LDS-DMA read)
typedef __attribute__((ext_vector_type(4))) unsigned rs_u32x4;
__device__ __forceinline__ rs_u32x4 rs_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    rs_u32x4 r;
    r.x = (unsigned)a; r.y = (unsigned)(a >> 32) & 0xffffu; r.z = bytes; r.w = 0x00020000u;
    return r;
}
__device__ __forceinline__ void rs_glds(rs_u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

#define RS_MFMA32(W, X)                                                   \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((W).x, (X).x, acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((W).y, (X).y, acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((W).z, (X).z, acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((W).w, (X).w, acc, 0, 0, 0);
// (second row tile of a pair: the same x fragment, its own accumulator - acc2, which the half tiles use in the other kinds)
#define RS_MFMA32B(W, X)                                                    \
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32((W).x, (X).x, acc2, 0, 0, 0); \
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32((W).y, (X).y, acc2, 0, 0, 0); \
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32((W).z, (X).z, acc2, 0, 0, 0); \
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32((W).w, (X).w, acc2, 0, 0, 0);
#define RS_MFMA16(W, X)                                                     \
    acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32((W).x, (X).x, acc2, 0, 0, 0); \
    acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32((W).y, (X).y, acc2, 0, 0, 0); \
    acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32((W).z, (X).z, acc2, 0, 0, 0); \
    acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32((W).w, (X).w, acc2, 0, 0, 0);

// Batches of one or two rows: the same products on the vector ALUs.  A 32 x 32 MFMA tile costs 64 cycles per 2 k whatever the batch
// is; with one row every lane already holds x[0][8 kg + 4 h ..] (rows past B read row 0), so a k-group is 4 FMAs per lane into ONE
// accumulator (lane (n, h): row n of the tile, k half h) - 16 cycles instead of 256.
// (as inline assembly: left to the compiler, the updates of two accumulators were packed into v_pk_fma_f32, whose operands - PAIRS of
// resident weight registers that are not neighbours - it then copied into 64 more registers, hoisted out of the step loop: spills)
#define RS_VFMAC(A, W, X) asm("v_fmac_f32 %0, %1, %2" : "+v"(A) : "v"(W), "v"(X));
#define RS_VDOT(A, W, X) RS_VFMAC(A, (W).x, (X).x) RS_VFMAC(A, (W).y, (X).y) RS_VFMAC(A, (W).z, (X).z) RS_VFMAC(A, (W).w, (X).w)
// (two rows: even lanes load row 0's fragment, odd lanes row 1's; a lane gets the other row's from its neighbour - DPP quad_perm
// [1, 0, 3, 2], no LDS - and keeps one accumulator for its OWN row and one for the OTHER; the two swap roles in odd lanes at the end)
#define RS_DPPSWAP(v) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true))
#define RS_XROW1(X) make_float4(RS_DPPSWAP((X).x), RS_DPPSWAP((X).y), RS_DPPSWAP((X).z), RS_DPPSWAP((X).w))
#define RS_MUL32(W, X) if (V) { RS_VDOT(av[0], W, X) if (two_rows) { const float4 x1_ = RS_XROW1(X); RS_VDOT(av[1], W, x1_) } } else { RS_MFMA32(W, X) }
#define RS_MUL32B(W, X) if (V) { RS_VDOT(av2[0], W, X) if (two_rows) { const float4 x1_ = RS_XROW1(X); RS_VDOT(av2[1], W, x1_) } } else { RS_MFMA32B(W, X) }
#define RS_MUL16(W, X) if (V) { RS_VDOT(av2[0], W, X) if (two_rows) { const float4 x1_ = RS_XROW1(X); RS_VDOT(av2[1], W, x1_) } } else { RS_MFMA16(W, X) }
template <bool B> struct RsBool { static constexpr bool value = B; };

}  // namespace

typedef const __attribute__((address_space(4))) RsArgs* RsKarg;   // the kernel-argument segment (the kernels' only parameter)

template <int KIND, bool AR, bool TR = false, bool KA = AR>
__device__ __forceinline__ void rs_body(const RsArgs& p, char* smem, const int bid) {
    static_assert(!(AR && TR), "the tape belongs to the teacher-forced loop");
    using Cfg = RsCfg<KIND>;
    constexpr bool XH = Cfg::XH, ATT = Cfg::ATT;
    constexpr int NN = Cfg::NN, NC = Cfg::NC, NRN = Cfg::NRN, NLN = Cfg::NLN;
    constexpr int NKGW = ATT ? RS_NKGW_ATT : RS_NKGW_DEC;
    constexpr int H = ATT ? RS_A : RS_D;
    constexpr int NRH = XH ? NRN : 1, NCH = XH ? NC : 1;   // half-tile fragments in registers
    constexpr int RT = Cfg::RT;
    constexpr int NR2 = RT == 2 ? NRN : 1, NC2 = RT == 2 ? NC : 1;   // second row tile's fragments in registers

    // block -> tile(s): the 224-workgroup deal (one attention workgroup per row: L <= 128) or the 192-workgroup one (two per row)
    const int dec0 = p.layout == 2 ? 64 : 96;   // first decoder-LSTM workgroup
    int tile, xt = 0, xhalf = 0;
    if (KIND == 0) { const int m = bid >> 1, odd = bid & 1; tile = 3 * m + 2 * odd; xt = 3 * m + 1; xhalf = odd; }
    else if (KIND == 1) tile = 96 + (bid - 64);
    else if (KIND == 3) tile = 2 * bid;
    else tile = bid - dec0;

    int* ctrl = reinterpret_cast<int*>(smem + Cfg::OFF_CTRL);
    float* red = reinterpret_cast<float*>(smem + Cfg::OFF_RED);
    float* red2 = reinterpret_cast<float*>(smem + Cfg::OFF_RED2);
    float* hs = reinterpret_cast<float*>(smem + Cfg::OFF_HS);
    float* hs2 = reinterpret_cast<float*>(smem + Cfg::OFF_HS2);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int B = p.B, T = p.T;
    const unsigned blkb = (unsigned)B * 32u;   // bytes per k-group of a blocked vector
    unsigned* const tmo_w = p.sync + HANDOFF_TIMEOUT;
    const int rep = bid % RS_REP;   // the flag replica this workgroup reads
    const RsPoll poll{p.sync + RS_FLAG_ATT + rep * 128, p.sync + RS_FLAG_DEC + rep * 128, p.sync + RS_FLAG_CTX + (bid % RS_REP1) * 32 * 32, tmo_w,
                      AR ? p.sync + RS_FLAG_PRE + (bid % RS_REP_PRE) * 8 * 32 : nullptr, AR ? p.sync + RS_FLAG_Y1 : nullptr, AR ? p.sync + HANDOFF_STOP : nullptr,
                      (p.spin_limit ? p.spin_limit : HANDOFF_SPIN_LIMIT) * 16u, B, T, bid, dec0,
                      (!ATT && (p.debug & 16)) ? 32 : ((p.debug & 8) ? 8 : 0)};
    const int bl = lane & 31, h = lane >> 5;
    const bool x_mine = ((lane >> 4) & 1) == xhalf;
    const int mlane = x_mine ? lane : (lane ^ 16);   // the lane whose half-tile fragment this lane multiplies with
    const bool vmode = B <= 2 && !(p.debug & 64), two_rows = B == 2;   // products on the vector ALUs (RS_MUL32; GVX_RS_DEBUG & 64: never)
    // rows past B read row 0; their results are never stored (vector-ALU mode: row lane % 2 in every lane)
    const unsigned x_lane = (unsigned)((vmode ? (two_rows ? (bl & 1) : 0) : (bl < B ? bl : 0)) * 8 + 4 * h) * 4u;

    // ---- prologue: this wave's weight fragments, once per call.  Wave w owns, in the order it multiplies them (skinny.hip):
    //   attention LSTM: k-groups KG0 + 64 + 16 w + i (h_a columns, i < 16), then KG0 + 8 w + i (context columns, i < 8)
    //   decoder LSTM:   k-groups 32 w + i (waves 0-3: h_a columns) / 32 w + 64 + i (waves 4-7: h_d columns), i < 32, then 128 + 8 w + i
    const float4* wsrc = reinterpret_cast<const float4*>(ATT ? p.att_frag : p.dec_frag);
    constexpr bool ARD = AR && KIND == 2;   // decoder LSTM of the autoregressive loop: its own K deal (below)
    // the Prenet columns of the attention LSTM (k-groups [0, 32) of its matrix) multiplied in the kernel: the autoregressive loop only
    // (prenet(t) is made by the loop).  The teacher-forced loop adds the products of one GEMM over all steps (pre_gate): measured with
    // the part in the kernel, the tile + half workgroups - whose matrix pipes are 66 % busy and bound the step - lost 0.7-0.9 us
    // per step wherever the part was placed, more than the 0.8 ms GEMM (which overlaps the encoder) costs (round 4, EXPERIMENTS.md)
    constexpr bool PP = AR && ATT;
    constexpr int NSUB = ARD ? 2 : 1;
    const int kn0 = ATT ? RS_KG0_ATT + 64 + 16 * wave : (wave < 4 ? 32 * wave : 32 * wave + 64);
    // k-group of the wave's i-th "early" fragment inside the packed matrix (ARD: 16 h_d k-groups 192 + 16 w + i, then 16 h_a k-groups 16 w + i)
    auto kni = [&](int i) -> int { return ARD ? (i < 16 ? 192 + 16 * wave + i : 16 * wave + (i - 16)) : kn0 + i; };
    const int kc0 = ATT ? RS_KG0_ATT + 8 * wave : 128 + 8 * wave;
    const float4* wt = wsrc + (long)tile * NKGW * 64 + lane;
    const float4* wx = wsrc + (long)xt * NKGW * 64 + mlane;
    float4 wn[NRN], wc[NC], hn[NRH], hc[NCH], wn2[NR2], wc2[NC2];
    const float4* wt2 = wt + (long)NKGW * 64;   // the pair's second tile
#pragma unroll
    for (int i = 0; i < NRN; ++i) wn[i] = wt[(long)kni(i) * 64];
#pragma unroll
    for (int i = 0; i < NC; ++i) wc[i] = wt[(long)(kc0 + i) * 64];
#pragma unroll
    for (int i = 0; i < NR2; ++i) wn2[i] = RT == 2 ? wt2[(long)kni(i) * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NC2; ++i) wc2[i] = RT == 2 ? wt2[(long)(kc0 + i) * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NRH; ++i) hn[i] = XH ? wx[(long)kni(i) * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NCH; ++i) hc[i] = XH ? wx[(long)(kc0 + i) * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
    // the rest in LDS, private to the wave: [NLN][64 lanes] tile fragments, then [NLN][32] half fragments (written by the lanes
    // that own them, read by their partners too)
    float4* lw = reinterpret_cast<float4*>(smem + wave * Cfg::WAVE_W);   // [NLN][RT][64]
    float4* lh = lw + NLN * 64;
    const int hidx = (mlane & 15) + 16 * (mlane >> 5);
#pragma unroll
    for (int i = 0; i < NLN; ++i) {
        lw[i * RT * 64 + lane] = wt[(long)kni(NRN + i) * 64];
        if (RT == 2) lw[(i * RT + 1) * 64 + lane] = wt2[(long)kni(NRN + i) * 64];
        if (XH && x_mine) lh[i * 32 + hidx] = wx[(long)kni(NRN + i) * 64];
    }

    // ---- autoregressive loop: the Prenet columns (k-groups [0, 32) of the attention-LSTM matrix, 4 per wave) cannot be applied
    // before the loop.  Single tiles keep them in 16 more registers; a tile + half workgroup has neither registers nor LDS left
    // (48 KB) and stages them every step by LDS-DMA into the partial-sum regions `red` / `red2`, which are free during the
    // step's products: 6 KB per wave = 4 tile fragments + 4 half fragments, L2 hits, no register passes through
    constexpr int NP = (PP && KIND == 1) ? 4 : 1;
    float4 wp[NP], w1f[NP];
    const bool l2_wg = AR && KIND == 1 && bid - 64 < 8;   // the 8 workgroups that also run Prenet layer 2 (32 output units each)
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        wp[i] = (PP && KIND == 1) ? wt[(long)(4 * wave + i) * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
        // layer-2 fragment: lane (n, h) holds W1[32 i2 + n][8 kg + 4 h .. + 3], kg = 4 wave + i
        w1f[i] = l2_wg ? *reinterpret_cast<const float4*>(p.pre_w1 + (long)(32 * (bid - 64) + bl) * RS_P + 8 * (4 * wave + i) + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const rs_u32x4 rw_att = rs_rsrc(p.att_frag, p.att_frag_bytes);
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const unsigned stage_t = lds_base + (unsigned)Cfg::OFF_RED + (unsigned)wave * 4096u;    // [4][64] float4
    const unsigned stage_h = lds_base + (unsigned)Cfg::OFF_RED2 + (unsigned)wave * 2048u;   // [4][32] float4
    // half tile: lanes 0-31 fetch the 32 lanes of the fragment that hold rows 16 xhalf .. + 15 of a k-group, lanes 32-63 those of the next
    const unsigned v_half = (((unsigned)bl < 16u ? 16u * (unsigned)xhalf + (unsigned)bl : 32u + 16u * (unsigned)xhalf + ((unsigned)bl - 16u)) * 16u) + (unsigned)h * 1024u;

    // per-lane constants of the epilogue: cell state (registers for the whole loop), bias, query-slab weights
    const bool cell_wave = wave < 4 * RT;
    const bool cell2_wave = XH && (wave == 4 || wave == 5);
    const int ctile = tile + (RT == 2 ? (wave >> 2) : 0);   // the row tile whose unit this wave finishes
    const int g = wave & 3, jloc = 2 * g + h, j = ctile * 8 + jloc;
    const int b2 = (lane & 15) + 16 * (wave - 4), g2 = lane >> 4, jloc2 = 4 * xhalf + g2, j2 = xt * 8 + jloc2;
    float* c_mem = ATT ? p.c_a : p.c_d;
    const float* bias = ATT ? p.att_bias : p.dec_bias;
    float c_state = 0.f;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cell_wave) {
        if (bl < B) c_state = c_mem[(long)bl * H + j];
        bias4 = *reinterpret_cast<const float4*>(bias + ctile * 32 + 8 * g + 4 * h);
    } else if (cell2_wave) {
        if (b2 < B) c_state = c_mem[(long)b2 * H + j2];
        bias4 = *reinterpret_cast<const float4*>(bias + xt * 32 + 4 * jloc2);
    }
    const bool slab_wave = ATT ? wave < RS_ATT / 32 : (AR && wave < 3);   // (autoregressive loop: the decoder-LSTM tiles emit projection slabs, PSB <= 96 dims)

#ifdef GVX_STAMPS
    const bool stamp_wg = bid == 0 || bid == 64 || bid == dec0;
#endif
    for (int t = 0; t < T; ++t) {
        RS_STAMP(0);
        RS_WGSTAMP(0);
        // (the autoregressive loop's own arguments are read from the kernel-argument segment where they are used: as arguments they
        // sat in scalar registers for the whole loop and were spilled into vector lanes around the step's products)
        RsKarg ka = nullptr;
        if (KA) { ka = (RsKarg)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(ka)); }
        // ---- the step's addend (attention LSTM: the Prenet columns, applied to all steps before the loop): its round trip hides
        // under the products
        // (teacher-forced loop) the waves that will store h_a(t) into ring slot (t + 1) % RS make sure NOW that the slot's last reader,
        // decoder LSTM (t - RS_HA_SLOTS), has finished: here the wait - a look at the decoder-LSTM flags every few steps - hides under
        // the wait for h_a(t-1); in the epilogue it sat on the step's chain (0.9 us in the steps that needed a look).  The
        // autoregressive loop needs no check: prenet(t) exists because that cell's step t - 1 has finished
        if (ATT && !AR && !TR && t >= RS_HA_SLOTS && (wave == 6 || ((XH || RT == 2) && wave == 7))) rs_gate(ctrl, RC_HD, t + 1 - RS_HA_SLOTS, poll, lane);
        float4 add4 = bias4;
        if (PP && XH) {   // this step's Prenet-column fragments on their way into LDS (every reader of `red` / `red2` has passed the barrier that ended the last step)
            const unsigned kg = (unsigned)(4 * wave);
#pragma unroll
            for (int i = 0; i < 4; ++i) rs_glds(rw_att, (unsigned)lane * 16u, ((unsigned)tile * NKGW + kg + (unsigned)i) * 1024u, stage_t + (unsigned)i * 1024u);
#pragma unroll
            for (int i = 0; i < 2; ++i) rs_glds(rw_att, v_half, ((unsigned)xt * NKGW + kg + 2u * (unsigned)i) * 1024u, stage_h + (unsigned)i * 1024u);
        }
        if (ATT && !PP) {
            if (cell_wave && bl < B) {
                const float4 ad = *reinterpret_cast<const float4*>(p.pre_gate + ((long)t * B + bl) * 4 * RS_A + ctile * 32 + 8 * g + 4 * h);
                add4.x += ad.x; add4.y += ad.y; add4.z += ad.z; add4.w += ad.w;
            } else if (cell2_wave && b2 < B) {
                const float4 ad = *reinterpret_cast<const float4*>(p.pre_gate + ((long)t * B + b2) * 4 * RS_A + xt * 32 + 4 * jloc2);
                add4.x += ad.x; add4.y += ad.y; add4.z += ad.z; add4.w += ad.w;
            }
        }
        f32x16 acc2x;                    // pairs of tiles: the second tile's sums, kept until the second phase of the epilogue
        float av2x[2] = {0.f, 0.f};      // ... in vector-ALU mode
        // ---- the step's products and their way into `red`, once with the products on the matrix units and once on the vector ALUs
        // (B <= 2): two instantiations of one body, so that each has only ITS accumulators live beside the resident weights
        auto body = [&](auto vc) __attribute__((always_inline)) {
        constexpr bool V = decltype(vc)::value;
        f32x16 acc, acc2;
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc[q] = 0.f; acc2[q] = 0.f; }
        float av[2] = {0.f, 0.f}, av2[2] = {0.f, 0.f};   // vector-ALU mode: row b of the tile / of the half or second tile
        // ---- the Prenet columns (autoregressive loop): prenet(t) behind RC_PRE >= t, the one part of this cell on the chain - called
        // after the others
        auto prenet_part = [&]() __attribute__((always_inline)) {
            rs_gate(ctrl, RC_PRE, t, poll, lane);
            RS_ARSTAMP(2);
            const __amdgpu_buffer_rsrc_t rx = make_rsrc(ka->prenet + (long)(4 * wave) * B * 8);
            float4 xp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) xp[u] = load_sc1(rx, x_lane + (unsigned)u * blkb);
            if (XH) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the staged fragments have landed (and the x fragments with them)
                const float4* st = reinterpret_cast<const float4*>(red + wave * 16 * 64);
                const float4* sh = reinterpret_cast<const float4*>(red2 + wave * 8 * 64);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 wl = st[u * 64 + lane];
                    const float4 hl = sh[u * 32 + hidx];
                    RS_MUL32(wl, xp[u])
                    RS_MUL16(hl, xp[u])
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) { RS_MUL32(wp[u < NP ? u : 0], xp[u]) }
            }
        };

        // ---- part 1: the columns whose x is known early
        //   attention LSTM (t):  h_a(t-1) [ring slot t % RS]  behind RC_HA >= t
        //   decoder LSTM (t):    waves 0-3 h_a(t) [slot (t+1) % RS] behind RC_HA >= t + 1, waves 4-7 h_d(t-1) [hc slot t] behind RC_HD >= t
        //   decoder LSTM (t) of the autoregressive loop (on the chain): EVERY wave first its sixteenth of the h_d(t-1) columns, then
        //   of the h_a(t) columns - when h_a(t) arrives all four SIMDs multiply it (1.7 us instead of 3.4 on two), done before the context
        {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const float* xsrc;
            int word, need;
            // (training mode: h_a is the tape, slot t + 1 = after step t - no ring, nothing is overwritten)
            if (ATT) { xsrc = p.h_a + (long)(TR ? t : t % RS_HA_SLOTS) * RS_A * B + (long)(16 * wave) * B * 8; word = RC_HA; need = t; }
            else if (ARD && sub == 0) { xsrc = p.hc + (long)t * B * (RS_D + RS_E) + (long)(16 * wave) * B * 8; word = RC_HD; need = t; }
            else if (ARD) { xsrc = p.h_a + (long)((t + 1) % RS_HA_SLOTS) * RS_A * B + (long)(16 * wave) * B * 8; word = RC_HA; need = t + 1; }
            else if (wave < 4) { xsrc = p.h_a + (long)(TR ? t + 1 : (t + 1) % RS_HA_SLOTS) * RS_A * B + (long)(32 * wave) * B * 8; word = RC_HA; need = t + 1; }
            else { xsrc = p.hc + (long)t * B * (RS_D + RS_E) + (long)(32 * (wave - 4)) * B * 8; word = RC_HD; need = t; }
            rs_gate(ctrl, word, need, poll, lane);
            if (sub == NSUB - 1) { RS_STAMP(1); RS_WGSTAMP(1); }
            const __amdgpu_buffer_rsrc_t rx = make_rsrc(xsrc);
            constexpr int XG = (XH || RT == 2) ? 2 : 4;   // k-groups per x batch: later batches load while this one multiplies
            // batches in flight ahead of the one being multiplied.  One is enough where the cell is off the chain; the decoder LSTM of
            // the autoregressive loop multiplies h_a(t) between its arrival and the context's, and a batch's round trip is longer than
            // its products (0.43 us)
            constexpr int XD = ARD ? 3 : 2;
            constexpr int NNS = NN / NSUB;   // k-groups of this sub-part
            float4 xb[XD][XG];
#pragma unroll
            for (int d = 0; d < XD - 1; ++d)
#pragma unroll
                for (int u = 0; u < XG; ++u) xb[d][u] = load_sc1(rx, x_lane + (unsigned)(XG * d + u) * blkb);
#pragma unroll
            for (int gi = 0; gi < NNS / XG; ++gi) {
                if (gi + XD - 1 < NNS / XG) {
#pragma unroll
                    for (int u = 0; u < XG; ++u) xb[(gi + XD - 1) % XD][u] = load_sc1(rx, x_lane + (unsigned)(XG * (gi + XD - 1) + u) * blkb);
                }
#pragma unroll
                for (int u = 0; u < XG; ++u) {
                    const int i = sub * NNS + XG * gi + u;   // compile-time after unrolling: registers / LDS by index
                    const float4 xv = xb[gi % XD][u];
                    if (i < NRN) {
                        RS_MUL32(wn[i], xv)
                        if (XH) { RS_MUL16(hn[i < NRH ? i : 0], xv) }
                        if (RT == 2) { RS_MUL32B(wn2[i < NR2 ? i : 0], xv) }
                    } else {
                        const float4 wl = lw[(i - NRN) * RT * 64 + lane];
                        RS_MUL32(wl, xv)
                        if (XH) { const float4 hl = lh[(i - NRN) * 32 + hidx]; RS_MUL16(hl, xv) }
                        if (RT == 2) { const float4 wl2 = lw[((i - NRN) * RT + 1) * 64 + lane]; RS_MUL32B(wl2, xv) }
                    }
                }
            }
        }
        }
        RS_STAMP(2);
        // ---- part 2: the context columns (registers): ctx(t-1) [hc slot t] behind RC_CTX >= t (attention LSTM), ctx(t) [hc slot t + 1]
        // behind RC_CTX >= t + 1 (decoder LSTM)
        {
            const float* xsrc = p.hc + (long)(ATT ? t : t + 1) * B * (RS_D + RS_E) + (long)RS_D * B + (long)(8 * wave) * B * 8;
            // (decoder LSTM: "every attention-LSTM workgroup has published step t + 1" implies ctx(t) - they consumed it - and keeps
            // its 128 workgroups off the context flags' line, which the step's chain waits on; the last step has no successor)
            if (ATT) rs_gate(ctrl, RC_CTX, t, poll, lane);
            else if (AR) rs_gate(ctrl, RC_CTX, t + 1, poll, lane);   // (autoregressive loop: this cell is on the chain - no shortcut)
            else if (t + 2 <= T) rs_gate(ctrl, RC_HA, t + 2, poll, lane);
            else rs_gate(ctrl, RC_CTX, t + 1, poll, lane);
            RS_STAMP(3);
            RS_WGSTAMP(2);
            const __amdgpu_buffer_rsrc_t rx = make_rsrc(xsrc);
            constexpr int XG = 4;
            float4 xc[XG], xn[XG];
#pragma unroll
            for (int u = 0; u < XG; ++u) xc[u] = load_sc1(rx, x_lane + (unsigned)u * blkb);
#pragma unroll
            for (int gi = 0; gi < NC / XG; ++gi) {
                if (gi + 1 < NC / XG) {
#pragma unroll
                    for (int u = 0; u < XG; ++u) xn[u] = load_sc1(rx, x_lane + (unsigned)(XG * (gi + 1) + u) * blkb);
                }
#pragma unroll
                for (int u = 0; u < XG; ++u) {
                    const int i = XG * gi + u;
                    RS_MUL32(wc[i], xc[u])
                    if (XH) { RS_MUL16(hc[i < NCH ? i : 0], xc[u]) }
                    if (RT == 2) { RS_MUL32B(wc2[i < NC2 ? i : 0], xc[u]) }
                }
#pragma unroll
                for (int u = 0; u < XG; ++u) xc[u] = xn[u];
            }
        }
        if (AR && KIND == 1) {
            // ---- Prenet layer 2 of this step's input (8 workgroups, 32 output units each, K = 256 split over the 8 waves): y1 comes from
            // the attention rows (layer 1 on the frame of step t - 1), the result goes to all 96 attention-LSTM workgroups
            if (l2_wg && t >= 1) {
                // (this lane's keep bytes of layer 2's dropout: requested before the wait, their round trip is off the chain)
                unsigned km = 0;
                if (wave < 4 && bl < B) km = *reinterpret_cast<const unsigned*>(ka->keep1 + ((long)t * B + bl) * RS_P + 32 * (bid - 64) + 8 * wave + 4 * h);
                rs_gate(ctrl, RC_Y1, t, poll, lane);
                RS_ARSTAMP(0);
                const __amdgpu_buffer_rsrc_t ry = make_rsrc(ka->y1 + (long)(4 * wave) * B * 8);
                float4 xy[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) xy[u] = load_sc1(ry, x_lane + (unsigned)u * blkb);
                // every row has run the stop test of step t - 1: when all have fired, the loop is over (models/tts/tacotron2.py:401-406) -
                // nothing is published any more, every wait of every workgroup ends at its next look at the stop word
                if (bid == 64 && tid == 0 && __hip_atomic_load(ka->n_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= B) {
                    __hip_atomic_store(p.sync + HANDOFF_STOP, (unsigned)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(ctrl + RC_ABORT, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (V) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) { RS_MUL32B(w1f[u < NP ? u : 0], xy[u]) }
                    // lane (n, hh) holds its k half of unit n: both halves, then into the MFMA result layout (lane (b, hh'), register
                    // 4 g + r = unit 8 g + 4 hh' + r) for rows b < B - the other columns of `red` are never stored
                    if (two_rows && (lane & 1)) { const float s2 = av2[0]; av2[0] = av2[1]; av2[1] = s2; }
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        av2[b] += __shfl_xor(av2[b], 32, 64);
                        if (h == 0 && b < B) red[(wave * 16 + 4 * (bl >> 3) + (bl & 3)) * 64 + b + 32 * ((bl >> 2) & 1)] = av2[b];
                        av2[b] = 0.f;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) { RS_MFMA32B(w1f[u < NP ? u : 0], xy[u]) }
#pragma unroll
                    for (int q = 0; q < 16; ++q) { red[(wave * 16 + q) * 64 + lane] = acc2[q]; acc2[q] = 0.f; }
                }
                rs_cbar();
                if (wave < 4) {   // lane (b, hh) of wave g: units 32 i2 + 8 g + 4 hh .. + 3 of row b
                    float4 o;
                    float* op = &o.x;
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        float tt = 0.f;
#pragma unroll
                        for (int w = 0; w < RS_WAVES; ++w) tt += red[(w * 16 + 4 * wave + qq) * 64 + lane];
                        op[qq] = tt;
                    }
                    if (bl < B) {
                        // dropout p = 0.5 at inference time too (models/tts/tacotron2.py:178): relu, then keep * 2
                        o.x = (km & 0xffu) ? 2.f * fmaxf(o.x, 0.f) : 0.f;
                        o.y = (km & 0xff00u) ? 2.f * fmaxf(o.y, 0.f) : 0.f;
                        o.z = (km & 0xff0000u) ? 2.f * fmaxf(o.z, 0.f) : 0.f;
                        o.w = (km & 0xff000000u) ? 2.f * fmaxf(o.w, 0.f) : 0.f;
                        const __amdgpu_buffer_rsrc_t rp = make_rsrc(ka->prenet + (long)(4 * (bid - 64) + wave) * B * 8);
                        store_sc1(rp, (unsigned)(bl * 8 + 4 * h) * 4u, o);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                rs_cbar();
                if (tid < RS_REP_PRE && __hip_atomic_load(ctrl + RC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0)
                    __hip_atomic_store(p.sync + RS_FLAG_PRE + (tid * 8 + (bid - 64)) * 32, (unsigned)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                RS_ARSTAMP(1);
            }
        }
        if (AR && PP) prenet_part();
        RS_STAMP(4);
        if (AR) RS_ARSTAMP(3);
        // ---- cross-wave K reduction through LDS (same order as skinny.hip): this wave's sums into its rows of `red`
        {
        int tr = tid;
        asm volatile("" : "+v"(tr));
        const int el = tr & 63;
        if (V) {
            // vector-ALU mode: lane (n, hh) holds its k half of row n of the tile for batch rows 0 (1): both halves, then into the MFMA
            // result layout the cell waves read (tile: lane (b, hh'), register 4 g + r = row 8 g + 4 hh' + r; half tile: lane
            // (b % 16, unit), register 4 (b / 16) + gate).  Columns of rows >= B keep old bytes: their cells are never stored
            const int en = el & 31;
            if (two_rows && (el & 1)) {   // odd lanes: "own" was row 1
                const float s0 = av[0], s2 = av2[0];
                av[0] = av[1]; av[1] = s0; av2[0] = av2[1]; av2[1] = s2;
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                av[b] += __shfl_xor(av[b], 32, 64);
                if (XH || RT == 2) av2[b] += __shfl_xor(av2[b], 32, 64);
                if (el < 32 && b < B) red[(wave * 16 + 4 * (en >> 3) + (en & 3)) * 64 + b + 32 * ((en >> 2) & 1)] = av[b];
                // (half tile: the lanes that hold their own fragment, rows 16 xhalf + (lane & 15) of the neighbour's tile)
                if (XH && el < 32 && ((el >> 4) & 1) == xhalf && b < B) red2[(wave * 8 + (el & 3)) * 64 + b + 16 * ((el & 15) >> 2)] = av2[b];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) red[(wave * 16 + q) * 64 + el] = acc[q];
            if (XH) {
#pragma unroll
                for (int q = 0; q < 8; ++q) red2[(wave * 8 + q) * 64 + el] = acc2[q] + acc2[8 + q];
            }
        }
        if (RT == 2) {
            if (V) { av2x[0] = av2[0]; av2x[1] = av2[1]; } else acc2x = acc2;
        }
        }
        };
        if (vmode) body(RsBool<true>{}); else body(RsBool<false>{});

        // ---- the epilogue's per-lane indices are recomputed from an opaque copy of the thread id: hoisted out of the step loop they
        // would sit in registers the resident weights need
        int te = tid;
        asm volatile("" : "+v"(te));
        const int el = te & 63, ebl = el & 31, eh = el >> 5;
        // query-slab weights of this wave's 32 attention dims: requested now, used after the cells (L2 hits)
        float4 wq_a = make_float4(0.f, 0.f, 0.f, 0.f), wq_b = wq_a, wq_c = wq_a, wq_d = wq_a;
        float4 xe = wq_a;
        if (AR && !ATT && slab_wave) {
            // projection slab of this tile's 8 hidden units, output dims 32 wave .. + 31 (rows past PSB: clamped, never stored), and the
            // four context columns that ride on it: weights (L2 hits) and ctx(t)[b][4 tile .. + 3]
            const int dd = min(32 * wave + ebl, ka->PSB - 1);
            const float* wp_l = ka->proj_hd_t + ((long)tile * ka->PSB + dd) * 8;
            wq_a = *reinterpret_cast<const float4*>(wp_l);
            wq_b = *reinterpret_cast<const float4*>(wp_l + 4);
            wq_c = *reinterpret_cast<const float4*>(ka->proj_ctx_t + ((long)tile * ka->PSB + dd) * 4);
            const __amdgpu_buffer_rsrc_t rc = make_rsrc(p.hc + (long)(t + 1) * B * (RS_D + RS_E) + (long)RS_D * B + (long)(tile >> 1) * B * 8);
            xe = load_sc1(rc, (unsigned)((ebl < B ? ebl : 0) * 8 + 4 * (tile & 1)) * 4u);
        }
        if (ATT && slab_wave) {
            const float* wq_l = p.wq_t + ((long)tile * RS_ATT + 32 * wave + ebl) * 8;
            wq_a = *reinterpret_cast<const float4*>(wq_l);
            wq_b = *reinterpret_cast<const float4*>(wq_l + 4);
            if (XH) wq_c = *reinterpret_cast<const float4*>(p.wq_t + ((long)xt * RS_ATT + 32 * wave + ebl) * 8 + 4 * xhalf);
            if (RT == 2) {   // the pair's second tile: its 8 hidden units follow the first tile's in the slab's sum
                const float* wq_l2 = p.wq_t + ((long)(tile + 1) * RS_ATT + 32 * wave + ebl) * 8;
                wq_c = *reinterpret_cast<const float4*>(wq_l2);
                wq_d = *reinterpret_cast<const float4*>(wq_l2 + 4);
            }
        }
        rs_cbar();
        RS_STAMP(5);
        if (cell2_wave) {
            const int rb = wave - 4;
            const int eb2 = (el & 15) + 16 * rb, eg2 = el >> 4;
            float s2[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                float tt = 0.f;
#pragma unroll
                for (int w = 0; w < RS_WAVES; ++w) tt += red2[(w * 8 + 4 * rb + qq) * 64 + el];
                s2[qq] = tt;
            }
            float hval = 0.f;
            if (eb2 < B) {
                const float p0 = s2[0] + add4.x, p1 = s2[1] + add4.y, p2 = s2[2] + add4.z, p3 = s2[3] + add4.w;
                c_state = sigmoidf_(p1) * c_state + sigmoidf_(p0) * tanhf_(p2);
                hval = sigmoidf_(p3) * tanhf_(c_state);
                if (TR) {   // the tape (skinny.hip's cell epilogue): pre-activations, the new cell state, dropout on the output
                    // (the pointers from the kernel-argument segment at the point of use - as arguments they sat in scalar registers
                    // for the whole loop and pushed others out; per-step bases are uniform, the lane's part is a 32-bit index)
                    RsKarg kq = (RsKarg)__builtin_amdgcn_kernarg_segment_ptr();
                    asm volatile("" : "+s"(kq));
                    const unsigned u = (unsigned)eb2 * (unsigned)H + (unsigned)(xt * 8 + 4 * xhalf + eg2);
                    const long tb = (long)t * B * H;
                    reinterpret_cast<float4*>(kq->tr_pre_a + tb * 4)[u] = make_float4(p0, p1, p2, p3);
                    (kq->tr_c_a + tb + (long)B * H)[u] = c_state;
                    hval = (kq->tr_keep_a + tb)[u] ? hval * kq->tr_scale_a : 0.f;
                }
            }
            hs2[eb2 * 4 + eg2] = hval;
        }
        // (a pair of tiles: `red` holds one tile's partial sums at a time - waves 0-3 finish the first tile's units, then every wave
        // writes the second tile's sums and waves 4-7 finish those; 64 KB of `red` would push two k-groups per wave out of LDS)
#pragma unroll
        for (int ph = 0; ph < RT; ++ph) {
            if (ph == 1) {
                rs_cbar();
                if (vmode) {
                    const int en = el & 31;
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        if (el < 32 && b < B) red[(wave * 16 + 4 * (en >> 3) + (en & 3)) * 64 + b + 32 * ((en >> 2) & 1)] = av2x[b];
                } else {
#pragma unroll
                    for (int q = 0; q < 16; ++q) red[(wave * 16 + q) * 64 + el] = acc2x[q];
                }
                rs_cbar();
            }
            if (cell_wave && (wave >> 2) == ph) {
                float s[4];
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    float tt = 0.f;
#pragma unroll
                    for (int w = 0; w < RS_WAVES; ++w) tt += red[(w * 16 + 4 * g + qq) * 64 + el];
                    s[qq] = tt;
                }
                float hval = 0.f;
                if (ebl < B) {
                    const float p0 = s[0] + add4.x, p1 = s[1] + add4.y, p2 = s[2] + add4.z, p3 = s[3] + add4.w;
                    c_state = sigmoidf_(p1) * c_state + sigmoidf_(p0) * tanhf_(p2);
                    hval = sigmoidf_(p3) * tanhf_(c_state);
                    if (TR) {
                        RsKarg kq = (RsKarg)__builtin_amdgcn_kernarg_segment_ptr();
                        asm volatile("" : "+s"(kq));
                        const unsigned u = (unsigned)ebl * (unsigned)H + (unsigned)((tile + (RT == 2 ? ph : 0)) * 8 + 2 * g + eh);
                        const long tb = (long)t * B * H;
                        reinterpret_cast<float4*>((ATT ? kq->tr_pre_a : kq->tr_pre_d) + tb * 4)[u] = make_float4(p0, p1, p2, p3);
                        ((ATT ? kq->tr_c_a : kq->tr_c_d) + tb + (long)B * H)[u] = c_state;
                        hval = ((ATT ? kq->tr_keep_a : kq->tr_keep_d) + tb)[u] ? hval * (ATT ? kq->tr_scale_a : kq->tr_scale_d) : 0.f;
                    }
                }
                hs[(ebl * RT + ph) * 8 + 2 * g + eh] = hval;
            }
        }
        rs_cbar();
        RS_STAMP(6);

        // ---- publication: h' as 16-byte write-through pieces, the query slab (attention LSTM), then the workgroup's flag
        float* hdst = ATT ? p.h_a + (long)(TR ? t + 1 : (t + 1) % RS_HA_SLOTS) * RS_A * B : p.hc + (long)(t + 1) * B * (RS_D + RS_E);
        if (slab_wave) {
            // slab[b][d] = sum_j h'[b][j] Wq[d][j] over the workgroup's 8 (12) hidden units, attention dims 32 wave .. + 31 (skinny.hip)
            f32x16 qa;
#pragma unroll
            for (int q = 0; q < 16; ++q) qa[q] = 0.f;
            const float* hrow = hs + ebl * 8 * RT + eh;
            qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow[0], eh ? wq_a.y : wq_a.x, qa, 0, 0, 0);
            qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow[2], eh ? wq_a.w : wq_a.z, qa, 0, 0, 0);
            qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow[4], eh ? wq_b.y : wq_b.x, qa, 0, 0, 0);
            qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow[6], eh ? wq_b.w : wq_b.z, qa, 0, 0, 0);
            if (RT == 2) {
                qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow[8], eh ? wq_c.y : wq_c.x, qa, 0, 0, 0);
                qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow[10], eh ? wq_c.w : wq_c.z, qa, 0, 0, 0);
                qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow[12], eh ? wq_d.y : wq_d.x, qa, 0, 0, 0);
                qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow[14], eh ? wq_d.w : wq_d.z, qa, 0, 0, 0);
            }
            if (XH) {
                const float* hrow2 = hs2 + ebl * 4 + eh;
                qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow2[0], eh ? wq_c.y : wq_c.x, qa, 0, 0, 0);
                qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow2[2], eh ? wq_c.w : wq_c.z, qa, 0, 0, 0);
            }
            if (AR && !ATT) {   // + sum_j Wp_ctx[d][4 tile + j] ctx(t)[b][4 tile + j]
                qa = __builtin_amdgcn_mfma_f32_32x32x2f32(eh ? xe.y : xe.x, eh ? wq_c.y : wq_c.x, qa, 0, 0, 0);
                qa = __builtin_amdgcn_mfma_f32_32x32x2f32(eh ? xe.w : xe.z, eh ? wq_c.w : wq_c.z, qa, 0, 0, 0);
            }
            // lane (d, hh) holds D[b = 8 gg + 4 hh + rr][d] in register 4 gg + rr: through the wave's own 4 KiB of `red` (free since the
            // barrier above) into rows of 32 floats, stored as 16-byte pieces - 8 whole 128-byte lines per instruction
            float* tq = red + wave * 16 * 64;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) tq[(8 * gg + 4 * eh + rr) * 32 + ebl] = qa[4 * gg + rr];
            const int sdim = ATT ? RS_ATT : ka->PSB;   // floats per slab row
            const __amdgpu_buffer_rsrc_t rq = make_rsrc(ATT ? p.q_slab + (long)bid * B * RS_ATT + 32 * wave : ka->p_slab + (long)tile * B * sdim + 32 * wave);
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = 8 * ps + (el >> 3), c4 = el & 7;
                const float4 v = *reinterpret_cast<const float4*>(tq + row * 32 + 4 * c4);
                if (row < B && (ATT || 32 * wave + 4 * c4 < sdim)) store_sc1(rq, (unsigned)(row * sdim + 4 * c4) * 4u, v);
            }
        } else if (wave == 6) {
            const __amdgpu_buffer_rsrc_t rh = make_rsrc(hdst + (long)tile * B * 8);
            // (lane l: row l / 2, units 4 (l % 2) .. + 3 of the first tile; hs rows hold 8 RT units)
            if (4 * el < 8 * B) store_sc1(rh, (unsigned)el * 16u, *reinterpret_cast<const float4*>(hs + (el >> 1) * 8 * RT + 4 * (el & 1)));
        } else if (RT == 2 && wave == 7) {
            const __amdgpu_buffer_rsrc_t rh = make_rsrc(hdst + (long)(tile + 1) * B * 8);
            if (4 * el < 8 * B) store_sc1(rh, (unsigned)el * 16u, *reinterpret_cast<const float4*>(hs + (el >> 1) * 16 + 8 + 4 * (el & 1)));
        } else if (XH && wave == 7) {
            const __amdgpu_buffer_rsrc_t rh = make_rsrc(hdst + (long)xt * B * 8);
            if (el < B) store_sc1(rh, (unsigned)(el * 8 + 4 * xhalf) * 4u, *reinterpret_cast<const float4*>(hs2 + 4 * el));
        }
        RS_STAMP(7);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains before the flag goes up
        RS_STAMP(8);
        // (autoregressive loop: a workgroup whose waits have been ended - every row has stopped, or a time-out - publishes nothing more
        // and leaves; the decision is one thread's, taken before the barrier, so that all waves leave after the same step)
        if (AR && tid == 0) ctrl[RC_EXIT + (t & 1)] = __hip_atomic_load(ctrl + RC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        rs_cbar();
        RS_STAMP(9);
        const bool leave = AR && __builtin_amdgcn_readfirstlane(ctrl[RC_EXIT + (t & 1)]) != 0;
        if (leave) break;
        // one wave instruction: lane r stores the workgroup's flag into replica r (addresses from an opaque copy of the thread id:
        // hoisted out of the loop they were spilled and reloaded here, on the chain)
        int tf = tid;
        asm volatile("" : "+v"(tf));
        if (tf < RS_REP) __hip_atomic_store(p.sync + (ATT ? RS_FLAG_ATT + bid : RS_FLAG_DEC + (bid - dec0)) + tf * 128, (unsigned)t + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (ATT && tf < RS_REP + RS_REP1)   // ... and lanes 32 .. 39 the slab flags the attention rows watch (a line each)
            __hip_atomic_store(p.sync + RS_FLAG_Q + ((tf - RS_REP) * dec0 + bid) * 32, (unsigned)t + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (AR && !ATT && tf < RS_REP + RS_REP_P)   // (autoregressive loop: the attention rows watch the projection slabs' flags)
            __hip_atomic_store(p.sync + RS_FLAG_P + ((tf - RS_REP) * 128 + (bid - dec0)) * 32, (unsigned)t + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        RS_WGSTAMP(3);
    }
    // final cell states (the launch-per-step loop keeps them in memory; callers that continue a sequence read them there)
    if (cell_wave && bl < B) c_mem[(long)bl * H + j] = c_state;
    if (cell2_wave && b2 < B) c_mem[(long)b2 * H + j2] = c_state;
}

// One kernel per deal: with all four workgroup kinds behind one run-time branch the register allocation of the kernel was
// the union of their needs (49 scalar registers spilled into vector lanes in the 224-workgroup deal's kinds, which the
// 192-workgroup deal's pairs of tiles had pushed out).
template <bool TR>
__device__ __forceinline__ void rs_kernel_224(const RsArgs& p, char* rs_smem) {
    const int bid = (int)blockIdx.x;
    const int kind = bid < 64 ? 0 : (bid < 96 ? 1 : 2);   // (uniform per workgroup)
    const int off_ctrl = kind == 0 ? RsCfg<0>::OFF_CTRL : (kind == 1 ? RsCfg<1>::OFF_CTRL : RsCfg<2>::OFF_CTRL);
    if (threadIdx.x < 64) reinterpret_cast<int*>(rs_smem + off_ctrl)[threadIdx.x] = 0;
    __syncthreads();
    if (kind == 0) rs_body<0, false, TR, !TR>(p, rs_smem, bid);
    else if (kind == 1) rs_body<1, false, TR, !TR>(p, rs_smem, bid);
    else rs_body<2, false, TR, !TR>(p, rs_smem, bid);
}
template <bool TR>
__device__ __forceinline__ void rs_kernel_192(const RsArgs& p, char* rs_smem) {
    const int bid = (int)blockIdx.x;
    const int kind = bid < 64 ? 3 : 2;
    const int off_ctrl = kind == 3 ? RsCfg<3>::OFF_CTRL : RsCfg<2>::OFF_CTRL;
    if (threadIdx.x < 64) reinterpret_cast<int*>(rs_smem + off_ctrl)[threadIdx.x] = 0;
    __syncthreads();
    if (kind == 3) rs_body<3, false, TR>(p, rs_smem, bid);
    else rs_body<2, false, TR>(p, rs_smem, bid);
}
__global__ __launch_bounds__(RS_THREADS) void decoder_resident_kernel(RsArgs p) {   // the 224-workgroup deal (p.layout == 1)
    extern __shared__ __attribute__((aligned(16))) char rs_smem[];
    rs_kernel_224<false>(p, rs_smem);
}
__global__ __launch_bounds__(RS_THREADS) void decoder_resident_pairs_kernel(RsArgs p) {   // the 192-workgroup deal (p.layout == 2)
    extern __shared__ __attribute__((aligned(16))) char rs_smem[];
    rs_kernel_192<false>(p, rs_smem);
}

// Training-mode forward of the same loop (models/tts/tacotron2.py:341, :358 under .train()): dropout on both cells' outputs with the
// caller's keep masks, and the tape of back-propagation through time written by the cell epilogues (RsArgs::tr_*).
__global__ __launch_bounds__(RS_THREADS) void decoder_resident_train_kernel(RsArgs p) {
    extern __shared__ __attribute__((aligned(16))) char rs_smem[];
    rs_kernel_224<true>(p, rs_smem);
}
__global__ __launch_bounds__(RS_THREADS) void decoder_resident_train_pairs_kernel(RsArgs p) {
    extern __shared__ __attribute__((aligned(16))) char rs_smem[];
    rs_kernel_192<true>(p, rs_smem);
}

// Autoregressive decode (models/tts/tacotron2.py:390-413 Decoder.inference): the same engine in the 224-workgroup deal, ONE launch
// for the whole decode.  The frame of step t feeds step t + 1, so both cells are on the step's chain:
//   attention LSTM (t): h_a(t-1) and ctx(t-1) columns early, the Prenet columns when prenet(t) arrives     -> h_a(t), query slabs
//   attention rows (t) (attn_persist.hip): energies, softmax, context                                        -> ctx(t)
//   decoder LSTM (t): h_d(t-1), h_a(t) columns early, the context columns when ctx(t) arrives                -> h_d(t), projection slabs
//   attention rows again: slab sum = frame + gate of step t, stop test, Prenet layer 1 on the frame          -> y1(t+1)
//   8 of the attention-LSTM workgroups: Prenet layer 2                                                       -> prenet(t+1)
// When every row's stop token has fired, the first Prenet workgroup raises the stop word instead of publishing: every waiter
// finds it at its next look and leaves, the host reads the number of steps from the word.
__global__ __launch_bounds__(RS_THREADS) void decoder_ar_resident_kernel(RsArgs p) {
    extern __shared__ __attribute__((aligned(16))) char rs_smem[];
    const int bid = (int)blockIdx.x;
    const int kind = bid < 64 ? 0 : (bid < 96 ? 1 : 2);
    const int off_ctrl = kind == 0 ? RsCfg<0>::OFF_CTRL : (kind == 1 ? RsCfg<1>::OFF_CTRL : RsCfg<2>::OFF_CTRL);
    if (threadIdx.x < 64) reinterpret_cast<int*>(rs_smem + off_ctrl)[threadIdx.x] = 0;
    __syncthreads();
    if (kind == 0) rs_body<0, true>(p, rs_smem, bid);
    else if (kind == 1) rs_body<1, true>(p, rs_smem, bid);
    else rs_body<2, true>(p, rs_smem, bid);
}

#ifdef GVX_STAMPS
hipError_t read_stamps_resident(unsigned long long* host480) {
    return hipMemcpyFromSymbol(host480, HIP_SYMBOL(rs_stamps), sizeof(unsigned long long) * 480);   // (kinds 0 - 2; kind 3's rows follow)
}
hipError_t read_wg_stamps_resident_ar(unsigned long long* host896) {
    return hipMemcpyFromSymbol(host896, HIP_SYMBOL(rs_wg_stamps_ar), sizeof(unsigned long long) * 896);
}
hipError_t read_wg_stamps_resident(unsigned long long* host896) {
    return hipMemcpyFromSymbol(host896, HIP_SYMBOL(rs_wg_stamps), sizeof(unsigned long long) * 896);
}
#endif

hipError_t decoder_resident_init() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_resident_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS_BYTES);
    if (e != hipSuccess) return e;
    for (const void* k : {reinterpret_cast<const void*>(decoder_resident_train_kernel), reinterpret_cast<const void*>(decoder_resident_pairs_kernel),
                          reinterpret_cast<const void*>(decoder_resident_train_pairs_kernel)})
        if ((e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS_BYTES)) != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_ar_resident_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS_BYTES);
}

bool decoder_resident_supported(int B, int L) { const int lay = attention_persistent_layout(B, L); return lay == 1 || lay == 2; }

hipError_t launch_decoder_resident(const DecResidentParams& p, hipStream_t s) {
    if (p.B < 1 || p.B > 32 || p.T < 1 || !p.att_frag || !p.dec_frag || !p.att_bias || !p.dec_bias || !p.wq_t || !p.pre_gate || !p.h_a ||
        !p.hc || !p.q_slab || !p.c_a || !p.c_d || !p.sync)
        return hipErrorInvalidValue;
    if (p.layout != 1 && p.layout != 2) return hipErrorInvalidValue;
    RsArgs a{};
    a.att_frag = p.att_frag; a.att_bias = p.att_bias; a.wq_t = p.wq_t; a.dec_frag = p.dec_frag; a.dec_bias = p.dec_bias;
    a.pre_gate = p.pre_gate; a.h_a = p.h_a; a.hc = p.hc; a.q_slab = p.q_slab; a.c_a = p.c_a; a.c_d = p.c_d; a.sync = p.sync;
    a.att_frag_bytes = p.att_frag_bytes; a.B = p.B; a.T = p.T; a.spin_limit = p.spin_limit; a.debug = p.debug; a.layout = p.layout;
    const dim3 grid(p.layout == 2 ? 192 : 224);
    if (p.tr_keep_a) {   // training mode: every tape pointer, or none
        if (!p.tr_keep_d || !p.tr_c_a || !p.tr_c_d || !p.tr_pre_a || !p.tr_pre_d) return hipErrorInvalidValue;
        a.tr_keep_a = p.tr_keep_a; a.tr_keep_d = p.tr_keep_d; a.tr_c_a = p.tr_c_a; a.tr_c_d = p.tr_c_d; a.tr_pre_a = p.tr_pre_a; a.tr_pre_d = p.tr_pre_d;
        a.tr_scale_a = p.tr_scale_a; a.tr_scale_d = p.tr_scale_d;
        if (p.layout == 2) decoder_resident_train_pairs_kernel<<<grid, dim3(RS_THREADS), RS_LDS_BYTES, s>>>(a);
        else decoder_resident_train_kernel<<<grid, dim3(RS_THREADS), RS_LDS_BYTES, s>>>(a);
    } else if (p.layout == 2) {
        decoder_resident_pairs_kernel<<<grid, dim3(RS_THREADS), RS_LDS_BYTES, s>>>(a);
    } else {
        decoder_resident_kernel<<<grid, dim3(RS_THREADS), RS_LDS_BYTES, s>>>(a);
    }
    return hipGetLastError();
}

hipError_t launch_decoder_ar_resident(const ArResidentParams& p, hipStream_t s) {
    if (p.B < 1 || p.B > 32 || p.T < 1 || !p.att_frag || !p.dec_frag || !p.att_bias || !p.dec_bias || !p.wq_t || !p.proj_hd_t || !p.proj_ctx_t ||
        !p.pre_w1 || !p.keep1 || !p.prenet || !p.y1 || !p.h_a || !p.hc || !p.q_slab || !p.p_slab || !p.c_a || !p.c_d || !p.n_done || !p.sync)
        return hipErrorInvalidValue;
    if (p.PSB < 8 || p.PSB > 96 || (p.PSB & 3) || p.att_frag_bytes == 0) return hipErrorInvalidValue;   // three slab waves of 32 dims
    RsArgs a{};
    a.att_frag = p.att_frag; a.att_bias = p.att_bias; a.wq_t = p.wq_t; a.dec_frag = p.dec_frag; a.dec_bias = p.dec_bias;
    a.proj_hd_t = p.proj_hd_t; a.proj_ctx_t = p.proj_ctx_t; a.pre_w1 = p.pre_w1; a.keep1 = p.keep1; a.prenet = p.prenet; a.y1 = p.y1;
    a.p_slab = p.p_slab; a.n_done = p.n_done; a.PSB = p.PSB;
    a.h_a = p.h_a; a.hc = p.hc; a.q_slab = p.q_slab; a.c_a = p.c_a; a.c_d = p.c_d; a.sync = p.sync;
    a.att_frag_bytes = p.att_frag_bytes; a.B = p.B; a.T = p.T; a.spin_limit = p.spin_limit; a.debug = p.debug; a.layout = 1;
    decoder_ar_resident_kernel<<<dim3(224), dim3(RS_THREADS), RS_LDS_BYTES, s>>>(a);
    return hipGetLastError();
}

}  // namespace gvx
