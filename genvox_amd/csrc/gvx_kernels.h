// Internal launcher declarations shared by the kernel translation units and the C-ABI (gvx_api.hip).
// gfx950 only. All device data is fp32 unless noted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Diagnostic build only (-DGVX_STAMPS, python -m genvox_amd.build --stamps): workgroup (0,0), thread 0 records the
// 100-MHz wall clock at phase boundaries into a device array that tools/stamps.py reads back.  Never defined in the
// shipped library; no stamp executes in the measured kernels.
#ifdef GVX_STAMPS
// each translation unit that stamps owns a file-local array (no relocatable device code needed)
namespace gvx { namespace { __device__ unsigned long long gvx_stamps[3][32]; } }
#define GVX_STAMP(k, i)                                                                         \
    do {                                                                                        \
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) gvx::gvx_stamps[k][i] = wall_clock64(); \
    } while (0)
#else
#define GVX_STAMP(k, i) do { } while (0)
#endif

namespace gvx {

// error reporting shared by every translation unit of the C ABI (thread-local message, returns `code`)
int set_error(int code, const char* msg);

enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };

// ---------------------------------------------------------------------------------------------
// Dense GEMM  C[m][n] = epi(sum_k A(m,k) * W[n][k])  on v_mfma_f32_32x32x2_f32 (exact fp32).
// Rows of A and C are addressed through a two-level affine map so that the same kernel runs
//   * plain row-major matrices                      (R = M, s0 = ld)
//   * conv1d as implicit GEMM on channels-last, halo-padded activations [B][T+2p][C]:
//     row (b,t) starts at b*(T+2p)*C + t*C and is k*C floats long (overlapping rows), (R = T)
//   * time-major -> batch-major scatter of the output (R = B).
// ---------------------------------------------------------------------------------------------
struct RowMap {
    int R;         // rows per outer group
    long s1, s0;   // offset(m) = (m / R) * s1 + (m % R) * s0   (in floats)
};

// Recurrent vectors (LSTM inputs/outputs, context, Prenet output) live in a k-group-blocked layout
//   v[b][k]  at  base + (k >> 3) * B * 8 + b * 8 + (k & 7)
// so that one wave-wide 16-byte-per-lane load of an MFMA x-fragment (32 rows x 8 k) is 1 KiB contiguous.
// The GEMM reads/writes that layout through a_kblk / c_nblk (= B*8); 8 means plain row-major.
struct GemmParams {
    const float* A; RowMap amap; long a_kblk = 8;   // element (m,k) at A + amap(m) + (k>>3)*a_kblk + (k&7)
    const float* W; long ldw;          // W row n at W + n*ldw (K contiguous)
    float* C; RowMap cmap; long c_nblk = 8;         // element (m,n) at C + cmap(m) + (n>>3)*c_nblk + (n&7)
    const float* bias;                 // [N] or nullptr
    const uint8_t* keep; long keep_ld; // Prenet keep mask [M][N] {0,1} or nullptr; kept values are doubled
    int M, N, K;                       // K % 4 == 0
    int act;
    // conv-output extras (rows grouped by cmap.R = frames per sequence):
    // split-K (few output tiles, long K: the small-batch products of the training backward): grid.y = splitk workgroups per
    // tile, split s sums k in [s * kchunk, (s + 1) * kchunk) and writes its partial tile at C + s * c_split; no bias /
    // activation / mask then (launch_gemm_splitk adds the partials in split order and the bias)
    int splitk = 1; int kchunk = 0; long c_split = 0;
    bool kmajor = false; RowMap wmap{1, 0, 0};   // K-major operands: A(m, k) at A + amap(k) + m, W(n, k) at W + wmap(k) + n (ldw, a_kblk unused)
    int m_begin = 0;                   // first row of this launch (launch_gemm may cover the rows of one product with two tile shapes)
    const int32_t* row_len = nullptr;  // [M / cmap.R] valid rows per group: rows at or past it are written as zeros
    int c_halo = 0;                    // > 0: C has c_halo halo rows before and after each group's cmap.R rows (C points at
                                       // the first interior row); the tile that owns an edge row also zeroes "its" halo row
                                       // (requires cmap.R >= c_halo)
};
hipError_t launch_gemm(const GemmParams& p, hipStream_t s);
// plain row-major C [M][N] (ldc) = A W^T (+ bias) with K split over `splitk` workgroups per tile; partials [splitk][M][N] in scratch
hipError_t launch_gemm_splitk(const GemmParams& p, int splitk, float* scratch, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Skinny recurrent GEMM (batch rows <= 64): one workgroup = 32 packed output rows x all batch rows,
// K split over the workgroup's waves, weights streamed once from HBM in MFMA-fragment order.
// Epilogue LSTM: fused cell update (i,f,g,o), optional attention-query partial slabs.
// Epilogue LINEAR: bias + activation + keep mask.
// ---------------------------------------------------------------------------------------------
struct XSeg { const float* p; int len; };   // blocked vector: x[b][k] = p[(k>>3)*B*8 + b*8 + (k&7)], len % 8 == 0

struct SkinnyJob {
    const float* Wp;        // packed fragments [ntiles][nkg][64 lanes][4]
    const float* bias;      // [N] in packed row order, or nullptr
    XSeg x[3];              // K = x[0].len + x[1].len + x[2].len
    int N;                  // output rows (LSTM: 4H in packed order row = 4*j + gate)
    int nkg;                // K / 8 of this job (the x segments)
    int kg0, nkg_w;         // the job's first k-group inside the packed matrix and the matrix's k-groups per tile
                            // (nkg_w = 0: the job covers the whole matrix, nkg_w = nkg)
    int mode;               // 0 = LSTM cell, 1 = linear, 2 = partial sums y[b][n] (row-major [B][N], N % 32 == 0)
    // --- LSTM epilogue
    float* c;               // [B][H] cell state (row-major, private to the owning workgroup), updated in place
    float* h_out;           // h' as a blocked vector [H/8][B][8]
    // encoder extras (all nullptr/0 for the decoder)
    const float* addend; long add_bs, add_ts;  // pre-activation addend[b][t_b][n] (encoder: x-projection incl. bias;
                                               // autoregressive decoder: partial sums of the other column slice)
    const float* addend2;                      // decoder cells / partial-sum jobs: a second addend [b][n] with the stride add_bs
                                               // (autoregressive loop: the h_a columns, summed beside the attention step); a mode-2
                                               // job adds `addend` and `addend2` to the sums it stores
    const int32_t* lengths; int step; int reverse; int seq_len;  // packed-sequence semantics
    float* seq_out; long seq_bs, seq_ts;       // seq_out[b][t_b][j] (row-major encoder output)
    const float* h_prev;                       // blocked; carried over for inactive rows
    // training mode: dropout on the cell's hidden output (models/tts/tacotron2.py:341, :358): h' = keep ? h' * h_scale : 0
    // before anything consumes it (h_out, sequence output, query slab); keep [B][H] uint8, nullptr = no dropout
    const uint8_t* h_keep; float h_scale;
    // training mode: where to write the new cell state (nullptr: in place, into c) and, for sequence jobs, the cell state of
    // every position like seq_out (c_seq_out[b][t_b][j], same strides) - what back-propagation through time reads
    float* c_out; float* c_seq_out;
    // training mode, decoder cells: the gate pre-activations (i, f, g, o of unit j at pre_out[(b * H + j) * 4 ...]) for the backward pass
    float* pre_out;
    // attention query partial products: slab[tile][b][a] = sum_{j in tile} Wq[a][j] * h'[b][j]
    const float* Wq_t;      // [H/8][att_dim][8] (tile-major repack of query_layer.weight) or nullptr
    float* q_slab; int att_dim;
    // extra terms of the slab (autoregressive decoder LSTM: the context columns of the mel / gate projection ride on the
    // projection slabs of the tiles): slab[tile][b][d] += sum_{j < 4} xw[(tile * att_dim + d) * 4 + j] * xsrc[b][4 tile + j],
    // xsrc a blocked vector; one batch tile only (B <= 32)
    const float* xw; const float* xsrc;
    // teacher-forced decoder with the persistent attention kernel (attn_persist.hip); all 0 / nullptr otherwise
    int defer_seg;                  // 1: every wave streams its share of x[1]'s k-groups (the context) after its share of the others
    const unsigned* ctx_cnt; unsigned ctx_target;   // wait for *ctx_cnt >= ctx_target before x[1]; x[1] is then read with sc1 loads
    unsigned* start_cnt;            // block 0 adds 1 here when the launch starts: everything the previous launch of the stream
                                    // stored (its query slabs) has been written back by then
    const unsigned* ready_cnt; unsigned ready_target;   // a tile does not end before *ready_cnt >= ready_target (first launch)
    unsigned* tmo;                  // hand-off status word
    unsigned spin_limit;            // polls before a wait gives up (0 = HANDOFF_SPIN_LIMIT)
    // --- linear epilogue
    float* y;               // blocked output [ceil(N/8)][B][8]
    const uint8_t* keep; long keep_stride;  // keep[b*stride + n]
    int act;
    int B;
};
// Location features of the NEXT attention step, computed by extra workgroups of the decoder LSTM launch (which is
// memory bound and leaves VALU/LDS idle): loc[b][l][:] = Wd * conv1d_k([w_prev ; w_cum])[l].  They depend only on the
// previous step's attention output, not on this launch's h_a, so they come off the critical chain of the step.
struct LocJob {
    const float* w_prev; long w_prev_bs;    // previous alignment row b at w_prev + b*bs, or nullptr (step 0)
    const float* w_cum;                     // [B][L]
    const float* loc_conv_t;                // [2][kl][32]
    const float* loc_dense_t;               // [32/4][a][4]
    const float* pm;                        // [B][L][a] processed memory added to the features (one-launch attention step), or nullptr
    float* loc_out;                         // [B][L][a]
    int B, L, a, kl, G;                     // G position chunks per row (0 = no location job in this launch)
};
enum SkinnyKind : int { SK_DECODER = 0, SK_ENCODER = 1, SK_AR = 2, SK_TRAIN = 3 };  // kernel name only; same code (up to 4 jobs)
hipError_t launch_skinny(const SkinnyJob* jobs, int njobs, int kind, hipStream_t s, const LocJob* loc = nullptr);
struct AttnParams;
// autoregressive loop: the one-launch attention step (attention.hip) and partial-sum tiles that only need h_a(t) in ONE launch -
// the attention workgroups first, then the tiles of `jobs` (B <= 32, attention dim <= 128)
hipError_t launch_skinny_attn(const SkinnyJob* jobs, int njobs, const AttnParams& ap, hipStream_t s);
// Encoder BiLSTM recurrence as ONE resident launch (skinny.hip, encoder_lstm_persistent_kernel): 2 directions x 32 tiles, every
// workgroup keeps its 32 KB of W_hh in registers for the whole sequence and its cell states in registers; the hidden state goes
// round through a double-buffered blocked vector (write-through stores, sc1 loads) whose words carry the hand-off's generation
// bit themselves (no flags).  B <= 32, H = 256 (the default layer size); packed-sequence semantics as the launch-per-step
// loop; every position of seq_out / c_seq_out is written (zeros past a row's length).
struct EncPersistParams {
    const float* Wp[2];        // packed fragments of W_hh per direction [H/8 tiles][H/8 k-groups][64][4]
    const float* xg;           // input projections incl. biases [B][L][2 * 4H] (direction-major inside a position)
    const int32_t* lengths;    // [B] or nullptr (all rows L)
    float* hx;                 // exchange buffers [2 directions][2][H/8][B][8], zeroed by the caller (bit 30 of every word is the
                               // hand-off's generation bit)
    float* seq_out;            // [B][L][2H]: direction d writes columns d*H ..
    float* c_seq_out;          // training tape: the cell state of every position, same layout, or nullptr
    unsigned* sync;            // HANDOFF_WORDS words; only the time-out word (HANDOFF_TIMEOUT) is used, zeroed by the caller
    unsigned spin_limit;
    int B, L, H;
    int debug_skip_block;      // tests: this workgroup leaves at once (its flags never go up: every wait on them times out); -1 = none
};
bool encoder_persistent_supported(int B, int H);
hipError_t launch_encoder_persistent(const EncPersistParams& p, hipStream_t s);
// teacher-forced step beside the persistent attention kernel: attention-LSTM (+ decoder-LSTM of the previous step) dealt to
// 96 (224) workgroups of equal weight (skinny.hip); default layer sizes, B <= 32
// layout 1: 224 (96) workgroups beside a 32-CU resident kernel; layout 2: 192 (64) workgroups beside a 64-CU one (skinny.hip)
hipError_t launch_skinny_pa(const SkinnyJob& att, const SkinnyJob* dec, hipStream_t s, int depth = 4, int layout = 1);
// layout 3 (33 .. 64 rows): up to three jobs of 128 tiles each, two batch tiles per workgroup, deferred context segments
hipError_t launch_skinny_pa64(const SkinnyJob* jobs, int njobs, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Location-sensitive attention, one decoder step, split over G workgroups per batch row:
//   attn_energy  (grid G x B): query (sum of LSTM partial slabs), location conv + dense, tanh, v-dot
//                              -> energies[b][l] for the workgroup's chunk of positions
//   attn_context (grid G x B): masked softmax over the row (recomputed per workgroup, it is tiny),
//                              context columns slice; slice 0 also emits the alignment row and w_cum.
// ---------------------------------------------------------------------------------------------
struct AttnParams {
    const float* q_slab; int n_slabs;       // [n_slabs][B][a]
    float* w_cum;                           // [B][L], updated in place by the context kernel
    const float* loc;                       // [B][L][a] location features of this step (written by the LSTM launch)
    const float* v;                         // [a]
    const float* pm;                        // [B][L][a]
    const float* memory;                    // [B][L][E]
    const int32_t* lengths;                 // [B] or nullptr
    float* energies;                        // [B][L] scratch between the two kernels
    float* w_out; long w_out_bs;            // new alignment row b -> w_out + b*bs   (length L)
    float* ctx_out;                         // blocked context vector [E/8][B][8]
    int B, L, a, F, kl, E, G;
    // (teacher-forced launch-per-step loop; GVX_ATTN_PREFETCH=0: off) first k-groups of the NEXT weight-streaming launch, pulled into this XCD's L2 by the
    // block with the same index (blocks of both launches are dealt round-robin over the XCDs): fragment-packed matrices of that
    // launch's jobs 0 / 1, their k-groups per tile, tiles of job 0, tiles in all
    const float* pf_w[2]; int pf_nkg[2]; int pf_tiles0, pf_tiles;
};
hipError_t launch_attention(const AttnParams& p, hipStream_t s);   // energy + context kernels (p.loc = location features)
// one launch: energies of the whole row (redundantly per slice), softmax, context slice; p.loc = pm + location features
// (LocJob.pm set), p.G = attention_slices(B, E)
hipError_t launch_attention_step(const AttnParams& p, hipStream_t s);
int attention_slices(int B, int E);
int attention_groups(int B, int L);                                // G for a batch / length
bool attention_supported(int L, int a, int F, int kl, int E);

// ---------------------------------------------------------------------------------------------
// Persistent attention of the teacher-forced loop (attn_persist.hip): one workgroup per batch row lives for all T steps
// beside the LSTM launches; hand-off words live in a 16-KB block of the workspace (HANDOFF_WORDS words) that is zeroed before every loop.
// ---------------------------------------------------------------------------------------------
// word indices: 4 KB apart, so that the pollers of one word do not queue in front of another word's traffic at the same channel
constexpr int HANDOFF_CNT_Q = 0, HANDOFF_CNT_CTX = 1024, HANDOFF_READY = 2048, HANDOFF_TIMEOUT = 3072;
constexpr int HANDOFF_STOP = 3072 + 512;   // the host's "the loop has ended early" word (autoregressive decode): the resident kernel leaves
constexpr int HANDOFF_PAIR = 4096;   // split resident kernel: flag word of half hf of row b at HANDOFF_PAIR + (2 b + hf) * 32
constexpr int HANDOFF_WORDS = 69632;   // (from 8192 on: flag replicas of the resident decoder loops, RS_FLAG_* below)
constexpr unsigned HANDOFF_SPIN_LIMIT = 200000u;   // polls with ~2 us of s_sleep between them (a few hundred ms), then the wait gives up
struct AttnPersistParams {
    const float* q_slab; int n_slabs;       // [n_slabs][B][a], rewritten (sc1) by the attention-LSTM tiles every step
    const float* v; const float* pm; const float* memory; const int32_t* lengths;
    const float* loc_conv_t; const float* loc_dense_t;
    float* w_out; long w_out_bs, w_out_ts;  // alignment row of (step t, row b) at w_out + t*ts + b*bs
    float* ctx_base; long ctx_ts;           // blocked context vector [E/8][B][8] of step t at ctx_base + t*ctx_ts
    unsigned* sync;                         // HANDOFF_WORDS words
    int B, L, T, kl;
    unsigned spin_limit;                    // polls before a wait gives up (0 = HANDOFF_SPIN_LIMIT)
    float* xchg;                            // split kernel (L > 128): exchange buffers of the row halves, attention_persistent_xchg_floats(B)
    unsigned q_first;                       // step t waits for the query counter to reach t + q_first (teacher-forced loop: 2 -
                                            // launch 0 signals too; autoregressive loop: 1 - one signalling launch per step)
    // beside the resident decoder kernel (dec_resident.hip): the slabs of step t are there when every one of the n_q_flags
    // producer flags reads >= t + 1 (in place of the query counter), and row b announces its context of step t by storing
    // t + 1 into ctx_flags[b] (in place of the context counter); nullptr otherwise
    const unsigned* q_flags; int n_q_flags;   // producer i's flag for the rows of replica r at q_flags[(r * n_q_flags + i) * 32], r < RS_REP1
    unsigned* ctx_flags;                      // replica r of row b's flag at ctx_flags[(r * 32 + b) * 32], r < RS_REP1
    int debug;   // timing experiments (GVX_RS_DEBUG & 32: s_sleep between looks at the flags)
    // autoregressive role (beside decoder_ar_resident_kernel; p_slab == nullptr otherwise): after publishing its context, row b
    // waits for the 128 projection slabs of the step (p_flags), sums them into the step's frame + gate (proj_out), runs the stop
    // test and Prenet layer 1 on the frame and hands y1 to the Prenet workgroups (y1_flags)
    const float* p_slab; int PSB, n_mels;     // [128][B][PSB]
    const float* proj_b;                      // [n_mels + 1]
    float* proj_out;                          // [T][PSB/8][B][8] blocked projection vector of every step
    const float* pre_w0_t;                    // [n_mels][P] Prenet layer 1 transposed
    const uint8_t* keep0;                     // [T][B][P] keep mask of layer 1's dropout for the input of step t
    float* y1;                                // blocked [P/8][B][8]
    int32_t* n_frames; int32_t* n_done; float gate_threshold;
    const unsigned* p_flags;                  // decoder-LSTM workgroup i's flag for the rows of replica r at p_flags[(r * 128 + i) * 32], r < RS_REP_P
    unsigned* y1_flags;                       // row b's flag at y1_flags[b * 32]
};
// ---------------------------------------------------------------------------------------------
// Teacher-forced decoder loop as ONE resident weight-streaming kernel beside the resident attention kernel (dec_resident.hip):
// 224 workgroups keep both LSTM cells' weight streams running through the steps' hand-offs (LDS-DMA loader ring per
// workgroup); default layer sizes, B <= 32, L <= 128.  Flag words (values = completed steps) live in the hand-off block:
// ---------------------------------------------------------------------------------------------
// Every flag exists in RS_REP replicas on lines of their own: operations on one line are served one after the other at the memory
// side (~15 ns each), so 224 workgroups polling the line that the producers are storing into took 2.5 us per look (round 4,
// profiles/r04_stamps_resident_stationary_v1.txt).  A producer stores its flag into every replica with ONE wave instruction (lane r
// -> replica r); a reader polls the replica of its index modulo RS_REP: 7 readers per line.
constexpr int RS_REP = 32;
constexpr int RS_FLAG_ATT = 8192;                        // [RS_REP][128] attention-LSTM workgroup i (< 96 / 64) has published h_a and its query slab of steps < value
constexpr int RS_FLAG_DEC = RS_FLAG_ATT + RS_REP * 128;  // [RS_REP][128] decoder-LSTM workgroup i has published h_d of steps < value
// The two hand-offs ON the step's chain - contexts to the attention-LSTM workgroups, query slabs to the attention rows - use flags
// on lines of their own (one writer per line: 32 stores into one line took up to 1 us to be acknowledged), in RS_REP1 replicas:
constexpr int RS_REP1 = 8;
constexpr int RS_FLAG_CTX = RS_FLAG_DEC + RS_REP * 128;             // [RS_REP1][32 rows][32 words]: attention row b has published its context of steps < value
constexpr int RS_FLAG_Q = RS_FLAG_CTX + RS_REP1 * 32 * 32;          // [RS_REP1][96][32 words]: attention-LSTM workgroup i has published its query slab of steps < value
// Autoregressive resident loop (dec_resident.hip, decoder_ar_resident_kernel): three more hand-offs on the step's chain, all with one
// writer per line:
constexpr int RS_FLAG_Y1 = RS_FLAG_Q + RS_REP1 * 96 * 32;           // [32 rows][32 words]: attention row b has published Prenet layer 1 of steps <= value (8 readers)
constexpr int RS_REP_PRE = 8;
constexpr int RS_FLAG_PRE = RS_FLAG_Y1 + 32 * 32;                   // [RS_REP_PRE][8][32 words]: Prenet workgroup i has published its slice of layer 2 of steps <= value
constexpr int RS_REP_P = 4;
constexpr int RS_FLAG_P = RS_FLAG_PRE + RS_REP_PRE * 8 * 32;        // [RS_REP_P][128][32 words]: decoder-LSTM workgroup i has published h_d and its projection slab of steps < value
static_assert(RS_FLAG_P + RS_REP_P * 128 * 32 <= HANDOFF_WORDS, "flag replicas inside the hand-off block");
constexpr int RS_HA_SLOTS = 8;            // ring of h_a vectors: h_a(t) in slot (t + 1) % RS_HA_SLOTS, slot 0 = the zero state (the decoder-LSTM workgroups of the teacher-forced loop may run up to 7 steps behind)
struct DecResidentParams {
    const float* att_frag; const float* att_bias; const float* wq_t;   // packed [128][224][64][4], [4A] packed row order, [A/8][a][8]
    const float* dec_frag; const float* dec_bias;                      // packed [128][320][64][4], [4D]
    const float* pre_gate;          // [T][B][4A] Prenet columns of the attention LSTM applied to all steps (bias not included)
    float* h_a;                     // [RS_HA_SLOTS][A/8][B][8]
    float* hc;                      // [T+1][(D+E)/8][B][8]: slot t + 1 = [h_d(t) ; ctx(t)]
    float* q_slab;                  // [96][B][a]
    float* c_a; float* c_d;         // [B][A], [B][D] cell states: read at the start, written at the end
    unsigned* sync;                 // HANDOFF_WORDS words, zeroed by the caller
    unsigned att_frag_bytes, dec_frag_bytes;   // bounds of the loader's buffer descriptors
    int B, T;
    unsigned spin_limit;            // polls without progress before the poller gives up (0 = HANDOFF_SPIN_LIMIT)
    int debug;                      // timing experiments only (GVX_RS_DEBUG: sleeps between polls)
    int layout;                     // 1: 224 workgroups beside one attention workgroup per row (L <= 128); 2: 192 workgroups - 64 pairs of
                                    // attention-LSTM tiles + 128 decoder-LSTM tiles - beside two per row (128 < L <= 256)
    // training mode (all or none): keep masks of the dropout on both cells' outputs [T][B][H] and their scales 1 / (1 - p), the
    // tape - cell states [T+1][B][H] (slot 0 given), gate pre-activations [T][B][H][4]; h_a is then the tape of the (dropped)
    // hidden states [T+1][A/8][B][8] (slot 0 given) instead of the ring
    const uint8_t* tr_keep_a; const uint8_t* tr_keep_d; float tr_scale_a, tr_scale_d;
    float* tr_c_a; float* tr_c_d; float* tr_pre_a; float* tr_pre_d;
};
// Autoregressive decode as ONE launch of the same weight-stationary engine (224-workgroup deal) beside the resident attention
// kernel, which also reduces the projection slabs, tests the stop condition and runs Prenet layer 1 (attn_persist.hip, AR role).
struct ArResidentParams {
    const float* att_frag; const float* att_bias; const float* wq_t;   // as DecResidentParams (the Prenet columns, k-groups [0, P/8), are used here)
    const float* dec_frag; const float* dec_bias;
    const float* proj_hd_t;         // [D/8][PSB][8] projection weights of the tiles' hidden units (slab operand)
    const float* proj_ctx_t;        // [E/4][PSB][4] projection weights of the context columns, four per decoder-LSTM tile
    const float* pre_w1;            // [P][P] Prenet layer 2, row-major [out][in]
    const uint8_t* keep1;           // [T][B][P] keep mask of layer 2's dropout for the input of step t
    float* prenet;                  // blocked [P/8][B][8]: Prenet output = input of the step about to run (zero for step 0)
    const float* y1;                // blocked [P/8][B][8]: Prenet layer 1 of the next step (written by the attention rows)
    float* h_a; float* hc; float* q_slab;   // as DecResidentParams
    float* p_slab;                  // [128][B][PSB]
    float* c_a; float* c_d;
    const int32_t* n_done;          // rows whose stop token has fired
    unsigned* sync;
    unsigned att_frag_bytes;
    int B, T, PSB;
    unsigned spin_limit;
    int debug;
};
bool decoder_resident_supported(int B, int L);
hipError_t launch_decoder_ar_resident(const ArResidentParams& p, hipStream_t s);
hipError_t decoder_resident_init();
hipError_t launch_decoder_resident(const DecResidentParams& p, hipStream_t s);

bool attention_persistent_supported(int B, int L, int a, int F, int kl, int E, int att_rnn_dim, int dec_rnn_dim);
int attention_persistent_layout(int B, int L);      // 0: not served; 1 / 2 / 3: see attn_persist.hip
int attention_persistent_slabs(int layout);         // query slabs of the launch layout
int attention_persistent_workgroups(int B, int L);  // resident workgroups (= CUs held)
size_t attention_persistent_xchg_floats(int B);
hipError_t attention_persistent_init();
hipError_t launch_attention_persistent(const AttnPersistParams& p, hipStream_t s);
// *word = 1 with an agent-scope store (the form every polled hand-off word is written in)
hipError_t launch_handoff_set(unsigned* word, hipStream_t s);

#if defined(__HIPCC__)
// write-through / L1-bypassing 16-byte accesses of handed-off bytes (aux 16 = sc1)
typedef __attribute__((ext_vector_type(4))) unsigned gvx_u32x4;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float4 load_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const gvx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float load_sc1_f32(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 16));
}
__device__ __forceinline__ void store_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float4 v) {
    gvx_u32x4 u;
    u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)byte_off, 0, 16);
}
// One lane waits until *cnt >= target.  Bounded; a timeout (or one raised by anybody else) makes every wait return at once.
template <bool FEW_WAITERS = false>
__device__ __forceinline__ bool handoff_wait(const unsigned* cnt, unsigned target, unsigned* tmo, unsigned code, unsigned limit = 0u,
                                             const unsigned* stop = nullptr) {
    if (limit == 0u) limit = HANDOFF_SPIN_LIMIT;
    // Normally the word is there at the first look.  A waiter that is early backs off (up to ~2 us between polls): a launch has
    // ~220 waves that may wait on the same word, and their polls queue in front of the producer's own traffic.  FEW_WAITERS
    // (the 32 workgroups of the persistent attention kernel, which wait every step): poll every ~0.1 us instead - what they
    // wait for starts the step's critical chain.
    unsigned spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        ++spins;
        if ((spins & (FEW_WAITERS ? 127u : 7u)) == 1u) {   // (after a timeout every wait gives up at its first look)
            if (__hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (stop && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;   // not an error
            if (spins > (FEW_WAITERS ? 16u : 1u) * limit) {
                __hip_atomic_store(tmo, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        if (FEW_WAITERS || spins < 8u) __builtin_amdgcn_s_sleep(FEW_WAITERS ? 2 : 8);
        else __builtin_amdgcn_s_sleep(64);
    }
    return true;
}
#endif

// ---------------------------------------------------------------------------------------------
// Small data-movement kernels.
// ---------------------------------------------------------------------------------------------
// x[b][p + l][:] = emb[tokens[b][l]][:]   into a halo-padded channels-last buffer (halo rows pre-zeroed)
// (x2: a second halo buffer of the same shape whose halo rows are cleared as well, or nullptr)
hipError_t launch_embed(const int64_t* tokens, const float* emb, int n_tokens, float* x, float* x2, int B, int L, int E, int halo,
                        int* err_flag, hipStream_t s);
// frames[(t)*B + b][m] = (t == 0) ? 0 : mel_in[b][m][t-1]     t in [0, T]
hipError_t launch_frames_from_mel(const float* mel_in, float* frames, int B, int M, int T, hipStream_t s);
// proj [B][T][M+1] (batch-major)  ->  mel_out [B][M][T], gate_out [B][T]
hipError_t launch_split_projection(const float* proj, float* mel_out, float* gate_out, int B, int M, int T, hipStream_t s);
// [B][M][T] -> halo-padded channels-last [B][T+2p][M]; halo rows and frames t >= lens[b] (lens may be nullptr) are zeroed
hipError_t launch_to_channels_last(const float* src, float* dst, int B, int M, int T, int halo, const int32_t* lens, hipStream_t s);
// mel_post[b][m][t] = mel[b][m][t] + y[b][t][m]   (0 for t >= lens[b] when lens != nullptr)
hipError_t launch_residual_to_channels_first(const float* mel, const float* y, float* mel_post, int B, int M, int T,
                                             const int32_t* lens, hipStream_t s);
// zero the 2*halo halo rows of every sequence of a channels-last buffer [B][T+2*halo][C]
hipError_t launch_zero_halo(float* buf, int B, int T, int halo, int C, hipStream_t s);
hipError_t launch_mask_padding(float* mel, float* mel_post, float* gate, const int32_t* mel_lengths, int B, int M, int T,
                               hipStream_t s);
// if *tmo != 0 (a bounded in-launch wait of this call gave up): NaN over the n_arrays (<= 4) output arrays and *sticky = *tmo
// up to 8 arrays (4-byte aligned, sizes in bytes, multiples of 4) cleared by one launch
hipError_t launch_zero_many(void* const* ptrs, const size_t* bytes, int n_arrays, hipStream_t s);
hipError_t launch_poison_on_timeout(const unsigned* tmo, int* sticky, float* const* ptrs, const size_t* counts, int n_arrays, hipStream_t s);
// dst[b][t][:] = src[t][b][:]  (rows of n floats, n % 4 == 0 not required)
hipError_t launch_permute01(const float* src, float* dst, int T, int B, int n, hipStream_t s);
// Tacotron2Loss: out3 = {loss, mel_loss, gate_loss}; scratch = loss_scratch_bytes() of device memory (8-byte aligned)
size_t loss_scratch_bytes();
hipError_t launch_tacotron2_loss(const float* mel, const float* mel_post, const float* gate, const float* mel_t, const float* gate_t,
                                 long n_mel, long n_gate, double* scratch, float* out3, hipStream_t s);
hipError_t launch_mask_gen(uint8_t* out, size_t n, uint64_t seed, hipStream_t s);
// AR: gate logits of step t (blocked projection vector) -> per-row finished flags / frame counts / all-finished counter
hipError_t launch_ar_stop(const float* proj_t, int gate_col, float threshold, int t, int B,
                          int32_t* n_frames, int32_t* n_done, hipStream_t s);
// AR step tail: proj_t (blocked [PSB/8][B][8]) = sum over the n_slabs partial slabs [slab][B][PSB] (fixed order) + p_ctx
// (blocked, bias included), rows n <= M; the gate row also runs the per-row stop test of launch_ar_stop; then, unless
// keep0 == nullptr, the whole Prenet of the next step on the fresh frame: prenet_out (blocked [P/8][B][8]) =
// 2 keep1 relu(W1 (2 keep0 relu(W0 mel))), w0t = W0 transposed [M][P], w1t = W1 transposed [P][P]
hipError_t launch_ar_project(const float* p_slab, int n_slabs, const float* p_ctx, float* proj_t, int M, float threshold, int t, int B,
                             int32_t* n_frames, int32_t* n_done, const float* w0t, const float* w1t, int P, const uint8_t* keep0,
                             const uint8_t* keep1, float* prenet_out, hipStream_t s);
// AR: scatter the blocked per-step projections proj[t][PSB/8][B][8], t < steps, into mel_out [B][M][Tmax], gate_out [B][Tmax]
// frames t >= n_frames[b] get the padding values of the reference's mask_padding: mel 0, gate 1e3
hipError_t launch_ar_emit_all(const float* proj, float* mel_out, float* gate_out, int B, int M, int Tmax, int steps,
                              const int32_t* n_frames, hipStream_t s);
// dst[b][t][:] = src[t][b][:] for t < steps (0 for t >= n_frames[b]), dst rows have Tdst time slots
hipError_t launch_permute01_partial(const float* src, float* dst, int steps, int Tdst, int B, int n, const int32_t* n_frames,
                                    hipStream_t s);

}  // namespace gvx
