// Teacher-forced decoder loop as ONE resident weight-streaming kernel (models/tts/tacotron2.py:365-388 Decoder.forward,
// :333-363 Decoder.decode): both LSTM cells of every step, beside the resident attention kernel (attn_persist.hip).
//
// Why: as a launch per step (skinny.hip, decoder_lstm_step_pa_kernel) every step paid ~2 us from dispatch to its first
// weight bytes and ~3 us of tail (reduction barrier skew, cells, slabs, kernel-end write-back) during which the CU's load
// path idles - 4 of the loop's 14.7 ms at 32 x 800.  The weights do not depend on the step, so a workgroup that stays
// resident can keep its weight stream running THROUGH those bubbles: here a loader wave per workgroup pulls the
// workgroup's weight fragments into a ring in LDS by LDS-DMA (buffer_load ... lds, 1 KiB per instruction, no registers),
// always as far ahead of the consumers as the ring allows, while eight consumer waves multiply out of the ring
// (v_mfma_f32_32x32x2_f32, same tiles / K slices / summation order as the launch-per-step kernel: the results are
// bit-identical) and a poller wave watches the hand-off flags of the other workgroups.
//
// Roles inside a workgroup of 10 waves (one workgroup per CU, 224 of them; the other 32 CUs run the attention kernel):
//   waves 0-7  consumers: K split over the waves as in skinny.hip; per k-group one 16-byte LDS read of the weight fragment,
//              one 16-byte `sc1` load of the x fragment (h_a / context / h_d: written by OTHER workgroups of this launch),
//              4 (+4 for the half tile) MFMAs; then the cross-wave sum through LDS, the cells (cell state in registers
//              for the whole loop), the query slabs, the write-through publication of h and the slabs, one flag store.
//   wave 8     loader: walks the same piece sequence every step (the weights repeat), two k-groups ("a pair") per
//              consumer wave and round; throttled by the consumers' `consumed` words, publishes `landed` rounds behind a
//              counted s_waitcnt vmcnt.  Its queue is always full - which is why neither polls nor publications may
//              share a wave with it (vmcnt retires in issue order: a poll behind 48 KB of DMA returns 2 us late).
//   wave 9     poller: reads the flag words of the attention-LSTM workgroups (h_a, slabs), of the decoder-LSTM
//              workgroups (h_d) and of the attention rows (context) with `sc1` loads - its queue is otherwise empty -
//              and keeps their minima in LDS words the consumers gate their x loads on.  The only bounded spin of the
//              workgroup: on a time-out it raises the call's status word and opens every gate, so the grid drains and
//              the caller's poison launch overwrites the outputs (gvx_api.hip).
//
// Deal (the 224-workgroup layout of skinny.hip): blocks [0, 64) attention-LSTM tile + half of a neighbour's rows,
// [64, 96) attention-LSTM tiles 96 .. 127, [96, 224) decoder-LSTM tiles.  The decoder cell is off the step's chain
// (only the projection after the loop reads h_d): its workgroups free-run behind the attention-LSTM workgroups, gated
// by the same flags, at most RS_HA_SLOTS - 1 steps behind (the h_a ring; checked by the writers).
//
// Hand-offs (cdna_hip_programming.md guideline 16, MI355X_MICROARCH.md "Valid forms", first table row): every handed-off
// byte is stored `sc1` in 16-byte pieces, every storing wave drains (s_waitcnt vmcnt(0)), the consumer waves meet at a
// barrier, ONE lane stores the workgroup's flag (`sc1`); readers poll every producer's flag (`sc1` loads, one wave), set
// an LDS word behind the matched poll, and every load of the bytes is an `sc1` buffer load issued behind that word.
#include "gvx_kernels.h"

namespace gvx {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __attribute__((ext_vector_type(4))) unsigned rs_u32x4;
typedef __attribute__((ext_vector_type(4))) int rs_i32x4;

constexpr int RS_CONS = 8;                       // consumer waves
constexpr int RS_THREADS = (RS_CONS + 2) * 64;   // + loader + poller
constexpr int RS_A = 1024, RS_D = 1024, RS_E = 512, RS_P = 256, RS_ATT = 128;   // default layer sizes (attention_persistent_supported)
constexpr int RS_NKGW_ATT = (RS_P + RS_E + RS_A) / 8;   // k-groups per tile of the packed attention-LSTM matrix (224)
constexpr int RS_NKGW_DEC = (RS_A + RS_E + RS_D) / 8;   // ... of the decoder-LSTM matrix (320)
constexpr int RS_KG0_ATT = RS_P / 8;                    // the Prenet columns are applied before the loop (pre_gate)

// LDS layout (bytes)
constexpr int RS_RING_W = 12288;                              // ring of one consumer wave: 6 pairs of 2 KiB / 4 pairs of 3 KiB
constexpr int RS_OFF_RED = RS_CONS * RS_RING_W;               // [8 waves][16][64] floats
constexpr int RS_OFF_RED2 = RS_OFF_RED + RS_CONS * 16 * 64 * 4;   // [8][8][64] floats (half tile)
constexpr int RS_OFF_HS = RS_OFF_RED2 + RS_CONS * 8 * 64 * 4;      // [32][8] h' of the tile's units
constexpr int RS_OFF_HS2 = RS_OFF_HS + 32 * 8 * 4;                 // [32][4] h' of the half tile's units
constexpr int RS_OFF_CTRL = RS_OFF_HS2 + 32 * 4 * 4;               // control words
constexpr int RS_LDS_BYTES = RS_OFF_CTRL + 256;
static_assert(RS_LDS_BYTES <= 160 * 1024, "resident decoder LDS");
// control words (int index)
constexpr int RC_LANDED = 0;       // rounds whose pieces are in LDS (loader)
constexpr int RC_CONSUMED = 8;     // [8] pairs read by consumer wave w (32-byte aligned: the loader reads them as two int4)
constexpr int RC_HA = 16;          // steps whose h_a + slabs every attention-LSTM workgroup has published (poller)
constexpr int RC_CTX = 17;         // steps whose context every attention row has published
constexpr int RC_HD = 18;          // steps whose h_d every decoder-LSTM workgroup has published
constexpr int RC_ABORT = 19;       // != 0: some wait of this workgroup has given up - every wait returns at once, the workgroup drains
constexpr int RC_BAR = 24;         // consumer barrier counter
// Watchdog of the waits on LDS words (they depend on the poller's bounded wait and on each other only, so it never fires unless
// the kernel itself is wrong): polls before a wait raises the call's time-out word and the workgroup's abort word
constexpr unsigned RS_WATCHDOG = 1u << 24;

#ifdef GVX_STAMPS
// diagnostic build (tools/stamps_resident.py): wall-clock stamps (10 ns) of decoder step RS_STAMP_T in one workgroup of each kind
// (blocks 0 / 64 / 96): [kind][wave 0-7][event]; row [kind][8]: the loader's totals (flow-control wait, counted-vmcnt wait, loop)
constexpr int RS_STAMP_T = 20;
__device__ unsigned long long rs_stamps[3][10][16];
__device__ unsigned long long rs_wg_stamps[224][4];   // per workgroup, wave 0: step begins, gate 1, gate 2, flag stored (step RS_STAMP_T)
#define RS_WGSTAMP(i) do { if (t == RS_STAMP_T && tid == 0) rs_wg_stamps[bid][i] = wall_clock64(); } while (0)
#define RS_STAMP(ev) do { if (stamp_wg && t == RS_STAMP_T && lane == 0) rs_stamps[KIND][wave][ev] = wall_clock64(); } while (0)
#else
#define RS_STAMP(ev) do { } while (0)
#define RS_WGSTAMP(i) do { } while (0)
#endif

__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f); }

__device__ __forceinline__ rs_u32x4 rs_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    rs_u32x4 r;
    r.x = (unsigned)a; r.y = (unsigned)(a >> 32) & 0xffffu; r.z = bytes; r.w = 0x00020000u;
    return r;
}
// one LDS-DMA piece: 64 lanes x 16 bytes from rsrc + voff + soff to LDS lds_addr + 16 lane (M0: written in the same statement,
// restored afterwards - the compiler owns it; s_nop 0: M0 write -> LDS-DMA read)
__device__ __forceinline__ void rs_glds(rs_u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// barrier of the 8 consumer waves (s_barrier would include the loader and the poller): a counter in LDS, target = 8 x the
// number of barriers so far.  Workgroup-scope release / acquire: LDS traffic of a wave is processed in issue order.
// Wait until the LDS word reads >= need.  Returns at once when the workgroup's abort word is up.
__device__ __forceinline__ void rs_wait_ge(int* ctrl, int word, int need, unsigned* tmo, unsigned code) {
    unsigned n = 0;
    while (__hip_atomic_load(ctrl + word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
        if (__hip_atomic_load(ctrl + RC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return;
        if (++n > RS_WATCHDOG) {
            __hip_atomic_store(ctrl + RC_ABORT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(tmo, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
// barrier of the 8 consumer waves (s_barrier would include the loader and the poller): a counter in LDS, target = 8 x the
// number of barriers so far.  Workgroup-scope release / acquire: LDS traffic of a wave is processed in issue order.
__device__ __forceinline__ void rs_cbar(int* ctrl, int& nbar, int lane, unsigned* tmo) {
    ++nbar;
    if (lane == 0) __hip_atomic_fetch_add(ctrl + RC_BAR, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    rs_wait_ge(ctrl, RC_BAR, RS_CONS * nbar, tmo, 0x600u);
}
// gate of a consumer's x loads: the poller's word (it opens every gate after ITS bounded wait in every case)
__device__ __forceinline__ void rs_gate(int* ctrl, int word, int need, unsigned* tmo) { rs_wait_ge(ctrl, word, need, tmo, 0x610u); }

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = min(v, (unsigned)__shfl_xor((int)v, m, 64));
    return v;
}

}  // namespace

// KIND 0: attention-LSTM tile + half tile, 1: attention-LSTM tile, 2: decoder-LSTM tile
template <int KIND>
__device__ __forceinline__ void rs_body(const DecResidentParams& p, char* smem, const int bid) {
    constexpr bool XH = KIND == 0, ATT = KIND < 2;
    constexpr int PAIRB = XH ? 3072 : 2048;            // bytes of a pair slot: two tile fragments (+ two half fragments)
    constexpr int SP = RS_RING_W / PAIRB;              // pair slots per consumer ring
    constexpr int NP_N = ATT ? 8 : 16, NP_C = 4;       // pairs per wave and step: columns known early / context columns
    constexpr int NPAIRS = NP_N + NP_C;
    constexpr int NKGW = ATT ? RS_NKGW_ATT : RS_NKGW_DEC;
    constexpr int H = ATT ? RS_A : RS_D;

    int tile, xt = 0, xhalf = 0;
    if (KIND == 0) { const int m = bid >> 1, odd = bid & 1; tile = 3 * m + 2 * odd; xt = 3 * m + 1; xhalf = odd; }
    else if (KIND == 1) tile = 96 + (bid - 64);
    else tile = bid - 96;

    int* ctrl = reinterpret_cast<int*>(smem + RS_OFF_CTRL);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int B = p.B, T = p.T;
    const unsigned blkb = (unsigned)B * 32u;   // bytes per k-group of a blocked vector
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    unsigned* const tmo_w = p.sync + HANDOFF_TIMEOUT;

    if (wave == RS_CONS) {
        // ================================================================= loader
        const rs_u32x4 rw = rs_rsrc(ATT ? p.att_frag : p.dec_frag, ATT ? p.att_frag_bytes : p.dec_frag_bytes);
        const unsigned v_tile = (unsigned)lane * 16u;
        // half tile: lanes 0-31 fetch the 32 lanes of the fragment that hold rows 16 xhalf .. 16 xhalf + 15 for the pair's first
        // k-group, lanes 32-63 for its second one (the next 1 KiB of the packed matrix)
        const unsigned ml = (unsigned)(lane & 31);
        const unsigned v_half = ((ml < 16u ? 16u * (unsigned)xhalf + ml : 32u + 16u * (unsigned)xhalf + (ml - 16u)) * 16u) + (unsigned)(lane >> 5) * 1024u;
        const unsigned tile_base = (unsigned)tile * NKGW * 1024u, xt_base = (unsigned)xt * NKGW * 1024u;
        constexpr int IPR = (XH ? 3 : 2) * RS_CONS;   // DMA instructions per round
        constexpr int KEEP = XH ? 1 : 2;              // rounds left in flight behind the counted wait (<= 48 instructions outstanding)
        const int total = T * NPAIRS;
        int j = 0, slot = 0, min_consumed = 0;
        bool aborted = false;
#ifdef GVX_STAMPS
        unsigned long long st_fc = 0, st_vm = 0;
        const unsigned long long st_begin = wall_clock64();
#endif
        for (int r = 0; r < total; ++r) {
            // flow control: the slot of round r is free once every consumer has read its pair r - SP
            const int need = r - SP + 1;
#ifdef GVX_STAMPS
            const unsigned long long st0 = wall_clock64();
#endif
            if (need > min_consumed) {
                unsigned n = 0;
                while (true) {
                    const rs_i32x4 c0 = *reinterpret_cast<volatile rs_i32x4*>(ctrl + RC_CONSUMED);
                    const rs_i32x4 c1 = *reinterpret_cast<volatile rs_i32x4*>(ctrl + RC_CONSUMED + 4);
                    // (readfirstlane: the loop's control flow - and with it the scalar operands of the DMA statements - stays provably uniform)
                    min_consumed = __builtin_amdgcn_readfirstlane(min(min(min(c0.x, c0.y), min(c0.z, c0.w)), min(min(c1.x, c1.y), min(c1.z, c1.w))));
                    if (min_consumed >= need) break;
                    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(ctrl + RC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0) { aborted = true; break; }
                    if (++n > RS_WATCHDOG) {
                        __hip_atomic_store(ctrl + RC_ABORT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(tmo_w, 0x620u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        aborted = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (aborted) break;   // (the consumers no longer wait for anything)
            }
#ifdef GVX_STAMPS
            st_fc += wall_clock64() - st0;
#endif
            const unsigned slot_off = lds_base + (unsigned)slot * PAIRB;
#pragma unroll
            for (int w = 0; w < RS_CONS; ++w) {
                int widx;   // first k-group of the pair inside the tile's packed row of k-groups
                if (ATT) widx = j < NP_N ? RS_KG0_ATT + 64 + 16 * w + 2 * j : RS_KG0_ATT + 8 * w + 2 * (j - NP_N);
                else widx = j < NP_N ? (w < 4 ? 32 * w : 32 * w + 64) + 2 * j : 128 + 8 * w + 2 * (j - NP_N);
                const unsigned dst = slot_off + (unsigned)w * RS_RING_W;
                if (p.debug & 4) continue;   // (timing experiments: the protocol without the DMA)
                rs_glds(rw, v_tile, tile_base + (unsigned)widx * 1024u, dst);
                rs_glds(rw, v_tile, tile_base + (unsigned)widx * 1024u + 1024u, dst + 1024u);
                if (XH) rs_glds(rw, v_half, xt_base + (unsigned)widx * 1024u, dst + 2048u);
            }
            if (++j == NPAIRS) j = 0;
            if (++slot == SP) slot = 0;
#ifdef GVX_STAMPS
            const unsigned long long st1 = wall_clock64();
#endif
            if (KEEP == 1) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
#ifdef GVX_STAMPS
            st_vm += wall_clock64() - st1;
#endif
            static_assert(IPR * KEEP == (XH ? 24 : 32), "counted wait of the loader");
            if (r >= KEEP && lane == 0) __hip_atomic_store(ctrl + RC_LANDED, r - KEEP + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#ifdef GVX_STAMPS
        if ((bid == 0 || bid == 64 || bid == 96) && lane == 0) {
            rs_stamps[KIND][8][0] = st_fc; rs_stamps[KIND][8][1] = st_vm; rs_stamps[KIND][8][2] = wall_clock64() - st_begin;
        }
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(ctrl + RC_LANDED, total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }

    if (wave == RS_CONS + 1) {
        // ================================================================= poller
        const unsigned* f_att = p.sync + RS_FLAG_ATT;
        const unsigned* f_dec = p.sync + RS_FLAG_DEC;
        const unsigned* f_ctx = p.sync + RS_FLAG_CTX;
        unsigned* tmo = p.sync + HANDOFF_TIMEOUT;
        const unsigned limit = (p.spin_limit ? p.spin_limit : HANDOFF_SPIN_LIMIT) * 8u;
        // every look: the five flag lines + the status word, one round trip; then "how many steps has EVERYBODY published" per
        // class by ballots (a count only moves up, and by at most a few steps per look)
        unsigned spins = 0, seen_a = 0, seen_c = 0, seen_d = 0;
        const unsigned Tu = (unsigned)T;
        while (true) {
            const unsigned a0 = __hip_atomic_load(f_att + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned a1 = lane < 32 ? __hip_atomic_load(f_att + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
            const unsigned d0 = __hip_atomic_load(f_dec + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned d1 = __hip_atomic_load(f_dec + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned c0 = lane < B ? __hip_atomic_load(f_ctx + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
            const unsigned t0 = __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned va = min(a0, a1), vd = min(d0, d1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            unsigned na = seen_a, nc = seen_c, nd = seen_d;
            while (na < Tu && __all(va > na)) ++na;
            while (nc < Tu && __all(c0 > nc)) ++nc;
            while (nd < Tu && __all(vd > nd)) ++nd;
            const bool progress = na != seen_a || nc != seen_c || nd != seen_d;
            if (lane == 0) {
                if (nc != seen_c) __hip_atomic_store(ctrl + RC_CTX, (int)nc, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (na != seen_a) __hip_atomic_store(ctrl + RC_HA, (int)na, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (nd != seen_d) __hip_atomic_store(ctrl + RC_HD, (int)nd, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            seen_a = na; seen_c = nc; seen_d = nd;
            if (na >= Tu && nc >= Tu && nd >= Tu) return;   // every hand-off of the loop has happened
            if (progress) spins = 0;
            bool give_up = __builtin_amdgcn_readfirstlane(t0) != 0u;   // somebody else (a workgroup of this kernel or the attention kernel) has timed out
            if (++spins > limit) {
                if (lane == 0) __hip_atomic_store(tmo, 0x500u + (unsigned)(bid & 0xff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                give_up = true;
            }
            if (give_up) {   // open every gate: the consumers run to the end on whatever is in memory, the call is poisoned afterwards
                if (lane == 0) {
                    __hip_atomic_store(ctrl + RC_ABORT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(ctrl + RC_HA, 0x7fffffff, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(ctrl + RC_CTX, 0x7fffffff, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(ctrl + RC_HD, 0x7fffffff, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                return;
            }
        }
    }

    // ===================================================================== consumers
    float* red = reinterpret_cast<float*>(smem + RS_OFF_RED);
    float* red2 = reinterpret_cast<float*>(smem + RS_OFF_RED2);
    float* hs = reinterpret_cast<float*>(smem + RS_OFF_HS);
    float* hs2 = reinterpret_cast<float*>(smem + RS_OFF_HS2);
    const char* ring = smem + wave * RS_RING_W;
    const int bl = lane & 31, h = lane >> 5;
    const bool x_mine = ((lane >> 4) & 1) == xhalf;
    const int mlane = x_mine ? lane : (lane ^ 16);                          // the lane whose half-tile fragment this lane multiplies with
    const unsigned half_off = 2048u + (unsigned)((mlane & 15) + 16 * (mlane >> 5)) * 16u;
    const unsigned x_lane = (unsigned)((bl < B ? bl : 0) * 8 + 4 * h) * 4u;  // rows past B read row 0; their results are never stored

    // per-lane constants of the epilogue: cell state (registers for the whole loop), bias, query-slab weights
    const bool cell_wave = wave < 4;
    const bool cell2_wave = XH && (wave == 4 || wave == 5);
    const int g = wave & 3, jloc = 2 * g + h, j = tile * 8 + jloc;
    const int b2 = (lane & 15) + 16 * (wave - 4), g2 = lane >> 4, jloc2 = 4 * xhalf + g2, j2 = xt * 8 + jloc2;
    float* c_mem = ATT ? p.c_a : p.c_d;
    const float* bias = ATT ? p.att_bias : p.dec_bias;
    float c_state = 0.f;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cell_wave) {
        if (bl < B) c_state = c_mem[(long)bl * H + j];
        bias4 = *reinterpret_cast<const float4*>(bias + tile * 32 + 8 * g + 4 * h);
    } else if (cell2_wave) {
        if (b2 < B) c_state = c_mem[(long)b2 * H + j2];
        bias4 = *reinterpret_cast<const float4*>(bias + xt * 32 + 4 * jloc2);
    }
    float4 wq_a = make_float4(0.f, 0.f, 0.f, 0.f), wq_b = wq_a, wq_c = wq_a;
    const bool slab_wave = ATT && wave < RS_ATT / 32;
    if (slab_wave) {
        const float* wq_l = p.wq_t + ((long)tile * RS_ATT + 32 * wave + bl) * 8;
        wq_a = *reinterpret_cast<const float4*>(wq_l);
        wq_b = *reinterpret_cast<const float4*>(wq_l + 4);
        if (XH) wq_c = *reinterpret_cast<const float4*>(p.wq_t + ((long)xt * RS_ATT + 32 * wave + bl) * 8 + 4 * xhalf);
    }

    int nbar = 0, landed_seen = 0, r = 0, slot = 0;
#ifdef GVX_STAMPS
    const bool stamp_wg = bid == 0 || bid == 64 || bid == 96;
#endif
    for (int t = 0; t < T; ++t) {
        // ---- the step's addend (attention LSTM: the Prenet columns, applied to all steps before the loop): its round trip hides
        // under the stream
        RS_STAMP(0);
        RS_WGSTAMP(0);
        float4 add4 = bias4;
        if (ATT) {
            if (cell_wave && bl < B) {
                const float4 ad = *reinterpret_cast<const float4*>(p.pre_gate + ((long)t * B + bl) * 4 * RS_A + tile * 32 + 8 * g + 4 * h);
                add4.x += ad.x; add4.y += ad.y; add4.z += ad.z; add4.w += ad.w;
            } else if (cell2_wave && b2 < B) {
                const float4 ad = *reinterpret_cast<const float4*>(p.pre_gate + ((long)t * B + b2) * 4 * RS_A + xt * 32 + 4 * jloc2);
                add4.x += ad.x; add4.y += ad.y; add4.z += ad.z; add4.w += ad.w;
            }
        }
        f32x16 acc, acc2;
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc[q] = 0.f; acc2[q] = 0.f; }

        // ---- the two column parts of the step: x source, gate word, gate value, pairs
        //   attention LSTM (t):  h_a(t-1) [slot t % RS]  behind RC_HA >= t;   ctx(t-1) [hc slot t]  behind RC_CTX >= t
        //   decoder LSTM (t):    waves 0-3 h_a(t) [slot (t+1) % RS] behind RC_HA >= t + 1, waves 4-7 h_d(t-1) [hc slot t] behind
        //                        RC_HD >= t;   ctx(t) [hc slot t + 1] behind RC_CTX >= t + 1
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const float* xsrc;
            int word, need, npairs;
            if (part == 0) {
                npairs = NP_N;
                if (ATT) { xsrc = p.h_a + (long)(t % RS_HA_SLOTS) * RS_A * B + (long)(16 * wave) * B * 8; word = RC_HA; need = t; }
                else if (wave < 4) { xsrc = p.h_a + (long)((t + 1) % RS_HA_SLOTS) * RS_A * B + (long)(32 * wave) * B * 8; word = RC_HA; need = t + 1; }
                else { xsrc = p.hc + (long)t * B * (RS_D + RS_E) + (long)(32 * (wave - 4)) * B * 8; word = RC_HD; need = t; }
            } else {
                npairs = NP_C;
                xsrc = p.hc + (long)(ATT ? t : t + 1) * B * (RS_D + RS_E) + (long)RS_D * B + (long)(8 * wave) * B * 8;
                word = RC_CTX; need = ATT ? t : t + 1;
            }
            if (!(p.debug & 1)) rs_gate(ctrl, word, need, tmo_w);
            RS_STAMP(1 + 2 * part);
            RS_WGSTAMP(1 + part);
            const __amdgpu_buffer_rsrc_t rx = make_rsrc(xsrc);
            const int ngroups = npairs >> 1;   // groups of 4 k-groups: the x fragments of the next group load while this one multiplies
            float4 xc[4], xn[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) xc[u] = load_sc1(rx, x_lane + (unsigned)u * blkb);
            for (int gi = 0; gi < ngroups; ++gi) {
                const int gn = min(gi + 1, ngroups - 1);   // (the last group re-reads itself: no conditional loads)
#pragma unroll
                for (int u = 0; u < 4; ++u) xn[u] = load_sc1(rx, x_lane + (unsigned)(4 * gn + u) * blkb);
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    if (landed_seen <= r && !(p.debug & 2)) {
                        rs_wait_ge(ctrl, RC_LANDED, r + 1, tmo_w, 0x630u);
                        landed_seen = __hip_atomic_load(ctrl + RC_LANDED, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    const char* sp = ring + slot * PAIRB;
                    const float4 wa = *reinterpret_cast<const float4*>(sp + lane * 16);
                    const float4 wb = *reinterpret_cast<const float4*>(sp + 1024 + lane * 16);
                    float4 ha = make_float4(0.f, 0.f, 0.f, 0.f), hb = ha;
                    if (XH) {
                        ha = *reinterpret_cast<const float4*>(sp + half_off);
                        hb = *reinterpret_cast<const float4*>(sp + half_off + 512);
                    }
                    const float4 xa = xc[2 * pr], xb = xc[2 * pr + 1];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.x, xa.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.y, xa.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.z, xa.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.w, xa.w, acc, 0, 0, 0);
                    if (XH) {
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(ha.x, xa.x, acc2, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(ha.y, xa.y, acc2, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(ha.z, xa.z, acc2, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(ha.w, xa.w, acc2, 0, 0, 0);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wb.x, xb.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wb.y, xb.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wb.z, xb.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wb.w, xb.w, acc, 0, 0, 0);
                    if (XH) {
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(hb.x, xb.x, acc2, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(hb.y, xb.y, acc2, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(hb.z, xb.z, acc2, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(hb.w, xb.w, acc2, 0, 0, 0);
                    }
                    ++r;
                    if (++slot == SP) slot = 0;
                    // the pair's fragments are in registers (the LDS reads above are older than this store in the wave's LDS queue)
                    if (lane == 0) __hip_atomic_store(ctrl + RC_CONSUMED + wave, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) xc[u] = xn[u];
            }
            RS_STAMP(2 + 2 * part);
        }

        // ---- cross-wave K reduction through LDS (same order as skinny.hip)
#pragma unroll
        for (int q = 0; q < 16; ++q) red[(wave * 16 + q) * 64 + lane] = acc[q];
        if (XH) {
#pragma unroll
            for (int q = 0; q < 8; ++q) red2[(wave * 8 + q) * 64 + lane] = acc2[q] + acc2[8 + q];
        }
        rs_cbar(ctrl, nbar, lane, tmo_w);
        RS_STAMP(5);
        if (cell2_wave) {
            const int rb = wave - 4;
            float s2[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                float tt = 0.f;
#pragma unroll
                for (int w = 0; w < RS_CONS; ++w) tt += red2[(w * 8 + 4 * rb + qq) * 64 + lane];
                s2[qq] = tt;
            }
            float hval = 0.f;
            if (b2 < B) {
                const float p0 = s2[0] + add4.x, p1 = s2[1] + add4.y, p2 = s2[2] + add4.z, p3 = s2[3] + add4.w;
                c_state = sigmoidf_(p1) * c_state + sigmoidf_(p0) * tanhf_(p2);
                hval = sigmoidf_(p3) * tanhf_(c_state);
            }
            hs2[b2 * 4 + g2] = hval;
        }
        if (cell_wave) {
            float s[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                float tt = 0.f;
#pragma unroll
                for (int w = 0; w < RS_CONS; ++w) tt += red[(w * 16 + 4 * g + qq) * 64 + lane];
                s[qq] = tt;
            }
            float hval = 0.f;
            if (bl < B) {
                const float p0 = s[0] + add4.x, p1 = s[1] + add4.y, p2 = s[2] + add4.z, p3 = s[3] + add4.w;
                c_state = sigmoidf_(p1) * c_state + sigmoidf_(p0) * tanhf_(p2);
                hval = sigmoidf_(p3) * tanhf_(c_state);
            }
            hs[bl * 8 + jloc] = hval;
        }
        rs_cbar(ctrl, nbar, lane, tmo_w);
        RS_STAMP(6);

        // ---- publication: h' as 16-byte write-through pieces, the query slab (attention LSTM), then the workgroup's flag
        float* hdst = ATT ? p.h_a + (long)((t + 1) % RS_HA_SLOTS) * RS_A * B : p.hc + (long)(t + 1) * B * (RS_D + RS_E);
        if (slab_wave) {
            // slab[b][d] = sum_j h'[b][j] Wq[d][j] over the workgroup's 8 (12) hidden units, attention dims 32 wave .. + 31 (skinny.hip)
            f32x16 qa;
#pragma unroll
            for (int q = 0; q < 16; ++q) qa[q] = 0.f;
            const float* hrow = hs + bl * 8 + h;
            qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow[0], h ? wq_a.y : wq_a.x, qa, 0, 0, 0);
            qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow[2], h ? wq_a.w : wq_a.z, qa, 0, 0, 0);
            qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow[4], h ? wq_b.y : wq_b.x, qa, 0, 0, 0);
            qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow[6], h ? wq_b.w : wq_b.z, qa, 0, 0, 0);
            if (XH) {
                const float* hrow2 = hs2 + bl * 4 + h;
                qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow2[0], h ? wq_c.y : wq_c.x, qa, 0, 0, 0);
                qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow2[2], h ? wq_c.w : wq_c.z, qa, 0, 0, 0);
            }
            // lane (d, hh) holds D[b = 8 gg + 4 hh + rr][d] in register 4 gg + rr: through the wave's own 4 KiB of `red` (free since the
            // barrier above) into rows of 32 floats, stored as 16-byte pieces - 8 whole 128-byte lines per instruction
            float* tq = red + wave * 16 * 64;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) tq[(8 * gg + 4 * h + rr) * 32 + bl] = qa[4 * gg + rr];
            const __amdgpu_buffer_rsrc_t rq = make_rsrc(p.q_slab + (long)bid * B * RS_ATT + 32 * wave);
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = 8 * ps + (lane >> 3), c4 = lane & 7;
                const float4 v = *reinterpret_cast<const float4*>(tq + row * 32 + 4 * c4);
                if (row < B) store_sc1(rq, (unsigned)(row * RS_ATT + 4 * c4) * 4u, v);
            }
        } else if (wave == 6) {
            if (ATT && t >= RS_HA_SLOTS) rs_gate(ctrl, RC_HD, t + 1 - RS_HA_SLOTS, tmo_w);   // the slot's last reader, decoder LSTM (t - RS_HA_SLOTS), has finished
            const __amdgpu_buffer_rsrc_t rh = make_rsrc(hdst + (long)tile * B * 8);
            if (4 * lane < 8 * B) store_sc1(rh, (unsigned)lane * 16u, *reinterpret_cast<const float4*>(hs + 4 * lane));
        } else if (XH && wave == 7) {
            if (t >= RS_HA_SLOTS) rs_gate(ctrl, RC_HD, t + 1 - RS_HA_SLOTS, tmo_w);
            const __amdgpu_buffer_rsrc_t rh = make_rsrc(hdst + (long)xt * B * 8);
            if (lane < B) store_sc1(rh, (unsigned)(lane * 8 + 4 * xhalf) * 4u, *reinterpret_cast<const float4*>(hs2 + 4 * lane));
        }
        RS_STAMP(7);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains before the flag goes up
        RS_STAMP(8);
        rs_cbar(ctrl, nbar, lane, tmo_w);
        RS_STAMP(9);
        if (tid == 0) __hip_atomic_store(p.sync + (ATT ? RS_FLAG_ATT + bid : RS_FLAG_DEC + (bid - 96)), (unsigned)t + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        RS_WGSTAMP(3);
    }
    // final cell states (the launch-per-step loop keeps them in memory; callers that continue a sequence read them there)
    if (cell_wave && bl < B) c_mem[(long)bl * H + j] = c_state;
    if (cell2_wave && b2 < B) c_mem[(long)b2 * H + j2] = c_state;
}

__global__ __launch_bounds__(RS_THREADS) void decoder_resident_kernel(DecResidentParams p) {
    extern __shared__ __attribute__((aligned(16))) char rs_smem[];
    const int bid = (int)blockIdx.x;
    if (threadIdx.x < 64) reinterpret_cast<int*>(rs_smem + RS_OFF_CTRL)[threadIdx.x] = 0;
    __syncthreads();   // the only s_barrier of the kernel: all ten waves, before their roles part
    if (bid < 64) rs_body<0>(p, rs_smem, bid);        // uniform per workgroup
    else if (bid < 96) rs_body<1>(p, rs_smem, bid);
    else rs_body<2>(p, rs_smem, bid);
}

#ifdef GVX_STAMPS
hipError_t read_stamps_resident(unsigned long long* host480) {
    return hipMemcpyFromSymbol(host480, HIP_SYMBOL(rs_stamps), sizeof(unsigned long long) * 480);
}
hipError_t read_wg_stamps_resident(unsigned long long* host896) {
    return hipMemcpyFromSymbol(host896, HIP_SYMBOL(rs_wg_stamps), sizeof(unsigned long long) * 896);
}
#endif

hipError_t decoder_resident_init() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_resident_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS_BYTES);
}

bool decoder_resident_supported(int B, int L) { return attention_persistent_layout(B, L) == 1; }

hipError_t launch_decoder_resident(const DecResidentParams& p, hipStream_t s) {
    if (p.B < 1 || p.B > 32 || p.T < 1 || !p.att_frag || !p.dec_frag || !p.att_bias || !p.dec_bias || !p.wq_t || !p.pre_gate || !p.h_a ||
        !p.hc || !p.q_slab || !p.c_a || !p.c_d || !p.sync)
        return hipErrorInvalidValue;
    // the loader's descriptors bound every DMA read to the packed matrices
    if (p.att_frag_bytes != (unsigned)(128u * RS_NKGW_ATT * 1024u) || p.dec_frag_bytes != (unsigned)(128u * RS_NKGW_DEC * 1024u)) return hipErrorInvalidValue;
    decoder_resident_kernel<<<dim3(224), dim3(RS_THREADS), RS_LDS_BYTES, s>>>(p);
    return hipGetLastError();
}

}  // namespace gvx
