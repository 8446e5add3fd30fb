// C-ABI of the MI355X Tacotron2 forward path (include/genvox_amd.h): model handle, host-side weight
// packing (BatchNorm folding, LSTM gate-row permutation, MFMA-fragment layout), workspace planning and
// the launch sequences of the encoder, the teacher-forced / autoregressive decoder and the Postnet.
//
// Data layout in HBM (all fp32):
//   activations are channels-last; conv inputs carry a zero halo of (k-1)/2 rows per sequence
//       encoder   x[B][L+2p][E]         memory[B][L][E]        pm[B][L][a]        xg[B][L][2*4H]
//       decoder   frames[(T+1)*B][M]    prenet[(T+1)*B][P]     (time-major: one step's rows are contiguous)
//                 hc[T+1][B][D+E]       slot s holds [h_d ; ctx] after step s-1 (slot 0 = zeros); it is at
//                                       once the LSTM input of the next step and the A operand of the hoisted
//                                       mel/gate projection GEMM over all T*B rows
//                 h_a[2][B][A] (ping-pong), c_a[B][A], c_d[B][D], w_cum[B][L], q_slab[A/8][B][a]
//       postnet   y[B][T+2p][C]
//   weights live in one packed blob (see pack_weights) so that multi-GPU start-up is a single broadcast.
#include "../../include/genvox_amd.h"
#include "gvx_kernels.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace gvx {
#ifdef GVX_STAMPS
hipError_t read_stamps_skinny(unsigned long long* host96);
hipError_t read_stamps_attention(unsigned long long* host96);
hipError_t read_stamps_persist(unsigned long long* host96);
hipError_t read_wg_spans(unsigned long long* host1024);
hipError_t read_stamps_resident(unsigned long long* host480);
hipError_t read_wg_stamps_resident(unsigned long long* host896);
hipError_t read_wg_stamps_resident_ar(unsigned long long* host896);
hipError_t read_row_stamps_persist_ar(unsigned long long* host256);
hipError_t read_loc_stamps_persist(unsigned long long* host32);
hipError_t read_row_stamps_persist(unsigned long long* host512);
#endif
hipError_t skinny_init();
hipError_t gemm_init();
hipError_t attention_init();
hipError_t attention_persistent_init();
}  // namespace gvx

using namespace gvx;

namespace {
thread_local std::string g_err;
std::mutex g_capture_mutex;
}
namespace gvx {
int set_error(int code, const char* msg) { g_err = msg; return code; }
}

namespace {

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) return fail(GVX_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

constexpr int MAX_CONV = 8;
constexpr double BN_EPS = 1e-5;

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Blob {  // offsets in floats into the packed weight blob
    size_t emb;
    size_t enc_w[MAX_CONV], enc_b[MAX_CONV];
    size_t enc_wih, enc_bih, enc_whh_frag[2];
    size_t pre_w0, pre_w1, pre_w0_t, pre_w1_t;
    size_t att_frag, att_bias, att_wpre, wq_t, wmem, v, loc_conv, loc_dense;
    size_t dec_frag, dec_bias;
    size_t proj_w, proj_b, proj_frag, proj_hd_t, proj_ctx_frag, proj_ctx_t;   // last three: autoregressive split of the projection (see gvx_decoder_autoregressive)
    size_t post_w[MAX_CONV], post_b[MAX_CONV];
    size_t total;
};

inline size_t frag_floats(int N, int K) { return (size_t)((N + 31) / 32) * (K / 8) * 64 * 4; }

struct WsPlan {  // byte offsets into the caller's workspace
    size_t xa, xb, xg, enc_h, enc_c, flags, sync, memory;
    size_t pm, frames, pre1, prenet, h_a, c_a, c_d, hc, w_cum, q_slab, proj, energies, align_tm, len_copy, loc, ar_masks, p_slab, p_ctx;
    size_t att_part, dec_part, pre_gate, xchg;
    size_t ya, yb;
    size_t total;
};

}  // namespace

struct gvx_model {
    gvx_dims d;
    Blob blob;
    const float* dev_blob = nullptr;
    bool timing = false;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool ev_valid = false;
    int last_decoder_launches = 0;
    // hipGraph caches of the step loops.  A key holds every pointer / size the captured launches bake in - including the
    // weight blob: re-binding weights (load_state_dict -> new blob) must never replay launches that read the old one.
    struct LoopKey {
        const void* ws; const void* memory; const void* blob; int B, L, T; bool has_len;
        float threshold = 0.f;   // autoregressive graphs only
        int variant = 0;         // teacher-forced loop: 1 = with the persistent attention kernel
        bool operator==(const LoopKey& o) const {
            return ws == o.ws && memory == o.memory && blob == o.blob && B == o.B && L == o.L && T == o.T &&
                   has_len == o.has_len && threshold == o.threshold && variant == o.variant;
        }
    };
    // One entry per key: the graphs of its chunks (one for the encoder / teacher-forced loop, one per 16-step chunk of the
    // autoregressive loop).  Policy: the first call with a key launches eagerly and only remembers the key; capture starts at
    // the second sighting (a serving process sees a new (B, L) per request - instantiating ~60 graphs of ~100 nodes for a
    // shape that never comes back costs more than the launches it saves); at most GRAPH_SETS keys per cache, LRU eviction.
    struct GraphSet {
        LoopKey key;
        int sightings = 0;
        uint64_t last_use = 0;
        std::vector<hipGraphExec_t> execs;
    };
    static constexpr size_t GRAPH_SETS = 4;
    std::vector<GraphSet> ar_graphs, loop_graphs, enc_graphs;
    uint64_t use_clock = 0;
    bool capture_first = false;   // GVX_GRAPH_FIRST=1: capture at the first sighting (tests of the replay path)
    bool attn_one_launch = true;  // GVX_ATTN_SPLIT=1: energy + context as two launches (the round-1 step, kept for A/B runs)
    // teacher-forced loop: attention as one kernel that lives beside the LSTM launches (attn_persist.hip) when the shape
    // allows it; GVX_ATTN_PERSISTENT=0 keeps the launch per step
    bool attn_persistent = true;
    // GVX_AR_RESIDENT=1: the autoregressive loop runs beside the resident attention kernel too.  Off by default: measured
    // (round 3, 200-step decodes) 49 vs 47 us per step at batch 1 and no gain at 2 x 32 rows - launch C then has 256 equal
    // tiles for 256 - B free CUs, so one CU streams two of them
    bool ar_resident = false;
    bool enc_persistent = true;   // encoder BiLSTM recurrence as one resident launch (B <= 32, H = 256); GVX_ENC_PERSISTENT=0: launch per position
    bool ar_split_h = true;       // autoregressive step: the h_a(t) columns of both cells as partial sums beside the attention step
                                  // (GVX_AR_SPLIT_H=0: the round-2 schedule, attention as a launch of its own)
    // GVX_TF_ROWS64=1: batches of 33 .. 64 rows run as ONE call beside a 64-CU resident kernel (layout 3).  Off by default:
    // with two batch tiles per workgroup the fp32 matrix pipe sets the launch length (37 us per 64-row step, MFMA pipe 54 %
    // busy on the 192 CUs, round 3) and two 32-row lanes on two streams are faster (40.1 vs 44.4 us per 64-row step)
    bool tf_rows64 = false;
    // teacher-forced loop as ONE resident weight-streaming kernel beside the resident attention kernel (dec_resident.hip):
    // B <= 32, L <= 128, inference mode; GVX_TF_RESIDENT=0 keeps the launch per step
    bool tf_resident = true;
    bool ar_resident_loop = true;   // autoregressive decode as two resident kernels (GVX_AR_RESIDENT_LOOP=0: launches per step)
    bool tf_long_rows_224 = true;   // teacher-forced rows of 129-256 tokens, <= 16 rows: the 224-workgroup deal (GVX_TF_LONG_224=0: 192)
    int pa_depth = 4;                  // GVX_PA_DEPTH=6: prefetch depth of the launch beside the resident kernel (tests, A/B runs)
    unsigned spin_limit = 0;           // GVX_HANDOFF_SPIN_LIMIT: polls before an in-launch wait gives up (0 = the built-in limit)
    bool debug_skip_resident = false;  // GVX_DEBUG_SKIP_RESIDENT=1: never launch the resident attention kernel, so that every
                                       // wait of the loop runs into its limit (test of the time-out reporting only)
    hipStream_t pa_stream = nullptr;
    hipEvent_t pa_fork = nullptr, pa_join = nullptr, enc_mid = nullptr;
    bool train_resident_loop = true;   // GVX_TRAIN_RESIDENT_LOOP=0: the training forward's decoder loop as a launch per step
    int enc_fork_after = 1;      // GVX_ENC_FORK_AFTER=<n>: the caller's Prenet products start behind n encoder convolutions
    // autoregressive loop: the all-rows-finished counter of chunk k is read (pinned slot k & 1, event k & 1) while chunk k + 1 runs
    int32_t* ar_done_host = nullptr;
    hipEvent_t ar_ev[2] = {nullptr, nullptr};
    // device-side re-packing (gvx_model_pack_weights_device): where every float of the blob comes from, built once per
    // state_dict layout by running the HOST packer over index-coded stand-ins of the tensors
    std::vector<std::string> gather_names;
    std::vector<int64_t> gather_numel;
    int32_t* gather_off = nullptr;     // [blob.total] offset inside the source tensor (device)
    uint8_t* gather_tid = nullptr;     // [blob.total] source tensor + 1, 0 = constant zero (device)
    void drop_graphs() {
        for (auto* c : {&ar_graphs, &loop_graphs, &enc_graphs}) {
            for (auto& gs : *c)
                for (auto e : gs.execs)
                    if (e) (void)hipGraphExecDestroy(e);
            c->clear();
        }
    }
    hipStream_t cap_stream = nullptr;  // private stream used only to record captures (the caller's may be the null stream)
    bool use_graph = true;
    // per-launch timing of the decoder step kernels (measurement only)
    bool ktiming = false;
    std::vector<hipEvent_t> kev;
    int n_lstm_ev = 0, n_attn_ev = 0;
    int reserve_events(size_t n) {
        while (kev.size() < n) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return GVX_ERR_HIP;
            kev.push_back(e);
        }
        return GVX_OK;
    }
    // derived
    int H() const { return d.embed_dim / 2; }
    int PS() const { return (d.n_mels + 1 + 3) & ~3; }  // padded row stride of the mel+gate projection (row-major)
    int PSB() const { return (d.n_mels + 1 + 7) & ~7; } // floats per row of the blocked per-step projection vector
};

namespace {

Blob make_blob_layout(const gvx_dims& d) {
    Blob b{};
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off = align_up(off + n, 64); return o; };
    const int E = d.embed_dim, H = E / 2, M = d.n_mels, P = d.prenet_dim, A = d.att_rnn_dim, D = d.dec_rnn_dim;
    b.emb = take((size_t)d.n_tokens * E);
    for (int i = 0; i < d.enc_n_conv; ++i) { b.enc_w[i] = take((size_t)E * d.enc_kernel * E); b.enc_b[i] = take(E); }
    b.enc_wih = take((size_t)8 * H * E);
    b.enc_bih = take((size_t)8 * H);
    for (int dir = 0; dir < 2; ++dir) b.enc_whh_frag[dir] = take(frag_floats(4 * H, H));
    b.pre_w0 = take((size_t)P * M); b.pre_w1 = take((size_t)P * P);
    b.pre_w0_t = take((size_t)M * P);   // both Prenet matrices transposed ([in][out]) for the autoregressive step tail
    b.pre_w1_t = take((size_t)P * P);   // (ar_project_kernel)
    b.att_frag = take(frag_floats(4 * A, P + E + A)); b.att_bias = take((size_t)4 * A);
    b.att_wpre = take((size_t)4 * A * P);   // the Prenet columns of the attention LSTM again, plain [4A packed rows][P]: one GEMM per
                                            // teacher-forced loop applies them to all steps at once (persistent-attention path)
    b.wq_t = take((size_t)A * d.att_dim);
    b.wmem = take((size_t)d.att_dim * E); b.v = take(d.att_dim);
    b.loc_conv = take((size_t)2 * d.att_loc_kernel * 32);   // transposed [2][kl][32]
    b.loc_dense = take((size_t)32 * d.att_dim);             // transposed [32/4][a][4]
    b.dec_frag = take(frag_floats(4 * D, A + E + D)); b.dec_bias = take((size_t)4 * D);
    b.proj_w = take((size_t)(M + 1) * (D + E)); b.proj_b = take(M + 1);
    b.proj_frag = take(frag_floats(M + 1, D + E));
    b.proj_hd_t = take((size_t)D * ((M + 1 + 7) & ~7));
    b.proj_ctx_frag = take(frag_floats(M + 1, E));
    b.proj_ctx_t = take((size_t)(E / 4) * ((M + 1 + 7) & ~7) * 4);
    for (int i = 0; i < d.postnet_n_conv; ++i) {
        const int cin = i == 0 ? M : d.postnet_dim, cout = i == d.postnet_n_conv - 1 ? M : d.postnet_dim;
        b.post_w[i] = take((size_t)cout * d.postnet_kernel * cin);
        b.post_b[i] = take(cout);
    }
    b.total = off;
    return b;
}

bool persistent_path(const gvx_model* m, int B, int L);

enum WsMode : int { WS_TEACHER_FORCED = 0, WS_AUTOREGRESSIVE = 1 };

// status words at the front of every workspace (int32 indices into `flags`)
constexpr int FLAG_TOKEN = 0;      // sticky: a token id was outside the embedding table
constexpr int FLAG_AR_DONE = 1;    // autoregressive loop: rows finished
constexpr int FLAG_TIMEOUT = 2;    // sticky: a teacher-forced call ended with its hand-off time-out word set
constexpr int FLAG_AR_FRAMES = 64; // autoregressive loop: frame counts [B <= 64]

WsPlan make_ws_plan(const gvx_model* m, int B, int L, int T, int mode = WS_TEACHER_FORCED) {
    const gvx_dims& d = m->d;
    const int E = d.embed_dim, H = E / 2, M = d.n_mels, P = d.prenet_dim, A = d.att_rnn_dim, D = d.dec_rnn_dim;
    const int pe = (d.enc_kernel - 1) / 2, pp = (d.postnet_kernel - 1) / 2;
    WsPlan w{};
    size_t off = 0;
    auto take = [&](size_t floats) { size_t o = off; off = align_up(off + floats * sizeof(float), 256); return o; };
    // status words first, at a shape-independent offset (gvx_workspace_status)
    w.flags = take(128);  // FLAG_* words; the sticky ones are only cleared by gvx_workspace_status
    w.sync = take(HANDOFF_WORDS);   // hand-off words of the persistent attention kernel (zeroed before every decoder loop)
    w.xa = take((size_t)B * (L + 2 * pe) * E);
    w.xb = take((size_t)B * (L + 2 * pe) * E);
    w.xg = take((size_t)B * L * 8 * H);
    w.enc_h = take((size_t)2 * 2 * B * H);
    w.enc_c = take((size_t)2 * B * H);
    w.memory = take((size_t)B * L * E);  // encoder output of the fused forward
    w.len_copy = take((size_t)B);        // token lengths copied next to the graphs' other operands (offset independent of T)
    w.pm = take((size_t)B * L * d.att_dim);
    w.frames = take((size_t)(T + 1) * B * M);
    w.pre1 = take((size_t)(T + 1) * B * P);
    w.prenet = take((size_t)(T + 1) * B * P);
    w.h_a = take((size_t)RS_HA_SLOTS * B * A);   // ping-pong of the launch-per-step loops; ring of the resident loop (dec_resident.hip)
    w.c_a = take((size_t)B * A);
    w.c_d = take((size_t)B * D);
    w.hc = take((size_t)(T + 1) * B * (D + E));
    w.w_cum = take((size_t)B * L);
    w.q_slab = take((size_t)(A / 8) * B * d.att_dim);
    w.proj = take((size_t)B * T * m->PSB());
    w.energies = take((size_t)B * L);
    w.align_tm = take((size_t)T * B * L);   // alignments of the step loop, time-major [T][B][L]
    w.loc = take((size_t)B * L * d.att_dim);  // location features of the current step
    w.p_slab = take((size_t)(D / 8) * B * m->PSB());   // autoregressive mode: projection partials of the decoder-LSTM tiles
    w.p_ctx = take((size_t)B * m->PSB());            //   and of the context columns (blocked vector)
    w.att_part = take((size_t)2 * B * 4 * A);        // autoregressive mode: partial gate pre-activations [B][4A] / [B][4D] of the
    w.dec_part = take((size_t)2 * B * 4 * D);        //   column slices that are known one launch early (two buffers: the 64-row
                                                     //   teacher-forced loop finishes a decoder cell one launch after its partial)
    // teacher-forced loop beside the persistent attention kernel: Prenet contribution to the attention LSTM's gates, all steps
    // (only where that loop can run: 0.5 GB at B = 32, T = 1000 that the autoregressive / launch-per-step paths never touch)
    w.pre_gate = take(mode == WS_TEACHER_FORCED && persistent_path(m, B, L) ? (size_t)T * B * 4 * A : 0);
    w.xchg = take(attention_persistent_xchg_floats(B));   // split resident kernel (128 < L <= 256): exchange buffers of the row halves
    w.ar_masks = take(((size_t)2 * T * B * P + 3) / 4);  // autoregressive mode: keep masks copied next to the graphs' operands (bytes)
    const int cmax = d.postnet_dim > M ? d.postnet_dim : M;
    w.ya = take((size_t)B * (T + 2 * pp) * cmax);
    w.yb = take((size_t)B * (T + 2 * pp) * cmax);
    w.total = off;
    return w;
}

template <typename T>
T* ws_ptr(void* ws, size_t off) { return reinterpret_cast<T*>(reinterpret_cast<char*>(ws) + off); }

int check_dims(const gvx_dims& d) {
    if (d.n_tokens < 1) return fail(GVX_ERR_INVALID_ARG, "n_tokens must be >= 1");
    const int dims8[] = {d.embed_dim, d.prenet_dim, d.att_rnn_dim, d.dec_rnn_dim, d.att_dim, d.postnet_dim, d.n_mels};
    const char* names[] = {"embed_dim", "prenet_dim", "att_rnn_dim", "dec_rnn_dim", "att_dim", "postnet_dim", "n_mels"};
    for (int i = 0; i < 7; ++i)
        if (dims8[i] < 8 || dims8[i] % 8) return fail(GVX_ERR_UNSUPPORTED, "%s = %d must be a positive multiple of 8", names[i], dims8[i]);
    if (d.embed_dim % 16) return fail(GVX_ERR_UNSUPPORTED, "embed_dim = %d must be a multiple of 16 (BiLSTM halves are multiples of 8)", d.embed_dim);
    if (d.att_dim > 256) return fail(GVX_ERR_UNSUPPORTED, "att_dim = %d > 256 is not supported", d.att_dim);
    if (d.att_loc_filters < 1 || d.att_loc_filters > 32) return fail(GVX_ERR_UNSUPPORTED, "att_loc_filters = %d must be in [1, 32]", d.att_loc_filters);
    const int ks[] = {d.enc_kernel, d.att_loc_kernel, d.postnet_kernel};
    for (int k : ks)
        if (k < 1 || k % 2 == 0) return fail(GVX_ERR_UNSUPPORTED, "kernel size %d must be odd (the reference pads (k-1)/2 on both sides)", k);
    if (d.enc_n_conv < 1 || d.enc_n_conv > MAX_CONV || d.postnet_n_conv < 1 || d.postnet_n_conv > MAX_CONV)
        return fail(GVX_ERR_UNSUPPORTED, "number of convolutions must be in [1, %d]", MAX_CONV);
    return GVX_OK;
}

// W: N x K row-major -> [tile][k-group][lane][4]; lane (n = lane&31, half = lane>>5) holds k = 8*kg + 4*half + 0..3
void pack_frag(const std::vector<float>& W, int N, int K, float* out) {
    const int ntiles = (N + 31) / 32, nkg = K / 8;
    for (int t = 0; t < ntiles; ++t)
        for (int kg = 0; kg < nkg; ++kg)
            for (int lane = 0; lane < 64; ++lane) {
                const int n = t * 32 + (lane & 31), k = 8 * kg + 4 * (lane >> 5);
                float* o = out + (((size_t)t * nkg + kg) * 64 + lane) * 4;
                for (int s = 0; s < 4; ++s) o[s] = n < N ? W[(size_t)n * K + k + s] : 0.f;
            }
}

struct WeightTable {
    std::unordered_map<std::string, const gvx_weight_desc*> map;
    const float* get(const std::string& name, int64_t numel, int* rc) const {
        auto it = map.find(name);
        if (it == map.end()) { *rc = fail(GVX_ERR_MISSING_WEIGHT, "missing weight '%s'", name.c_str()); return nullptr; }
        if (it->second->numel != numel) {
            *rc = fail(GVX_ERR_SHAPE, "weight '%s' has %lld elements, expected %lld", name.c_str(), (long long)it->second->numel, (long long)numel);
            return nullptr;
        }
        return it->second->data;
    }
};

// conv (+ eval BatchNorm) -> [Cout][k][Cin] with the BN scale folded in, bias' = (b - mean) * scale + beta
int pack_conv(const WeightTable& wt, const std::string& prefix, int cout, int cin, int k, float* w_out, float* b_out) {
    int rc = GVX_OK;
    const float* w = wt.get(prefix + ".0.conv.weight", (int64_t)cout * cin * k, &rc); if (!w) return rc;
    const float* b = wt.get(prefix + ".0.conv.bias", cout, &rc); if (!b) return rc;
    const float* g = wt.get(prefix + ".1.weight", cout, &rc); if (!g) return rc;
    const float* beta = wt.get(prefix + ".1.bias", cout, &rc); if (!beta) return rc;
    const float* mu = wt.get(prefix + ".1.running_mean", cout, &rc); if (!mu) return rc;
    const float* var = wt.get(prefix + ".1.running_var", cout, &rc); if (!var) return rc;
    for (int co = 0; co < cout; ++co) {
        const double scale = (double)g[co] / std::sqrt((double)var[co] + BN_EPS);
        for (int kk = 0; kk < k; ++kk)
            for (int ci = 0; ci < cin; ++ci)
                w_out[((size_t)co * k + kk) * cin + ci] = (float)((double)w[((size_t)co * cin + ci) * k + kk] * scale);
        b_out[co] = (float)(((double)b[co] - (double)mu[co]) * scale + (double)beta[co]);
    }
    return GVX_OK;
}

// LSTM: rows permuted to row' = 4*j + gate, columns = [W_ih | W_hh], bias = b_ih + b_hh
int pack_lstm(const WeightTable& wt, const std::string& wih_name, const std::string& whh_name, const std::string& bih_name,
              const std::string& bhh_name, int Hd, int Kin, std::vector<float>* wcat, float* bias_out) {
    int rc = GVX_OK;
    const float* wih = wt.get(wih_name, (int64_t)4 * Hd * Kin, &rc); if (!wih) return rc;
    const float* whh = wt.get(whh_name, (int64_t)4 * Hd * Hd, &rc); if (!whh) return rc;
    const float* bih = wt.get(bih_name, 4 * Hd, &rc); if (!bih) return rc;
    const float* bhh = wt.get(bhh_name, 4 * Hd, &rc); if (!bhh) return rc;
    const int K = Kin + Hd;
    wcat->assign((size_t)4 * Hd * K, 0.f);
    for (int j = 0; j < Hd; ++j)
        for (int q = 0; q < 4; ++q) {
            const int src = q * Hd + j, dst = 4 * j + q;
            std::memcpy(&(*wcat)[(size_t)dst * K], wih + (size_t)src * Kin, sizeof(float) * Kin);
            std::memcpy(&(*wcat)[(size_t)dst * K + Kin], whh + (size_t)src * Hd, sizeof(float) * Hd);
            bias_out[dst] = bih[src] + bhh[src];
        }
    return GVX_OK;
}

hipError_t zero_async(void* p, size_t bytes, hipStream_t s) { return hipMemsetAsync(p, 0, bytes, s); }

// Find (or create, evicting the least recently used) the graph set of `key` and count the sighting.
gvx_model::GraphSet* touch_graph_set(gvx_model* m, std::vector<gvx_model::GraphSet>& cache, const gvx_model::LoopKey& key) {
    gvx_model::GraphSet* hit = nullptr;
    for (auto& gs : cache)
        if (gs.key == key) hit = &gs;
    if (!hit) {
        if (cache.size() >= gvx_model::GRAPH_SETS) {
            size_t lru = 0;
            for (size_t i = 1; i < cache.size(); ++i)
                if (cache[i].last_use < cache[lru].last_use) lru = i;
            for (auto e : cache[lru].execs)
                if (e) (void)hipGraphExecDestroy(e);
            cache.erase(cache.begin() + lru);
        }
        cache.emplace_back();
        hit = &cache.back();
        hit->key = key;
    }
    ++hit->sightings;
    hit->last_use = ++m->use_clock;
    return hit;
}

// Run the launches `enqueue(stream)` issues as chunk `chunk` of graph set `gs`: eagerly at the key's first sighting,
// afterwards from a hipGraph (captured on the model's private stream: the caller's may be the null stream, which
// cannot be captured).
template <class F>
int run_chunk(gvx_model* m, gvx_model::GraphSet* gs, size_t chunk, hipStream_t s, F&& enqueue) {
    if (!m->use_graph || !gs || (gs->sightings < 2 && !m->capture_first)) return enqueue(s);
    if (gs->execs.size() <= chunk) gs->execs.resize(chunk + 1, nullptr);
    hipGraphExec_t exec = gs->execs[chunk];
    if (!exec) {
        // Captures are serialised across handles: the host mirror drives two handles from two threads (chunk lanes), and
        // although each records on its own stream in thread-local mode, concurrent capture / instantiate is not something
        // to lean on in the runtime.  A one-time cost per graph; launches of existing graphs are not serialised.
        std::lock_guard<std::mutex> lock(g_capture_mutex);
        hipGraph_t graph = nullptr;
        if (!m->cap_stream) HIP_TRY(hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking));
        HIP_TRY(hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeThreadLocal));
        const int rc = enqueue(m->cap_stream);
        const hipError_t ce = hipStreamEndCapture(m->cap_stream, &graph);
        if (rc != GVX_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        HIP_TRY(ce);
        HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        HIP_TRY(hipGraphDestroy(graph));
        gs->execs[chunk] = exec;
    }
    HIP_TRY(hipGraphLaunch(exec, s));
    return GVX_OK;
}

}  // namespace

// =====================================================================================================
extern "C" {

const char* gvx_last_error(void) { return g_err.c_str(); }
int gvx_version(void) { return 1; }

int gvx_model_create(const gvx_dims* dims, gvx_model** out) {
    if (!dims || !out) return fail(GVX_ERR_INVALID_ARG, "null argument");
    int rc = check_dims(*dims);
    if (rc != GVX_OK) return rc;
    gvx_model* m = new gvx_model();
    m->d = *dims;
    m->blob = make_blob_layout(*dims);
    if (const char* e = std::getenv("GVX_NO_GRAPH")) m->use_graph = !(e[0] == '1');
    if (const char* e = std::getenv("GVX_GRAPH_FIRST")) m->capture_first = e[0] == '1';
    if (const char* e = std::getenv("GVX_ATTN_SPLIT")) m->attn_one_launch = !(e[0] == '1');
    if (const char* e = std::getenv("GVX_ATTN_PERSISTENT")) m->attn_persistent = !(e[0] == '0');
    // the resident attention kernel and the launches it feeds must run at the same time: under kernel serialisation every
    // hand-off would run into its limit
    for (const char* name : {"AMD_SERIALIZE_KERNEL", "HIP_LAUNCH_BLOCKING"})
        if (const char* e = std::getenv(name))
            if (e[0] != '\0' && e[0] != '0') m->attn_persistent = false;
    if (const char* e = std::getenv("GVX_AR_RESIDENT")) m->ar_resident = e[0] == '1';
    if (const char* e = std::getenv("GVX_AR_SPLIT_H")) m->ar_split_h = e[0] != '0';
    if (const char* e = std::getenv("GVX_ENC_PERSISTENT")) m->enc_persistent = e[0] != '0';
    if (const char* e = std::getenv("GVX_ENC_FORK_AFTER")) m->enc_fork_after = std::atoi(e);
    if (const char* e = std::getenv("GVX_TRAIN_RESIDENT_LOOP")) m->train_resident_loop = e[0] != '0';
    if (const char* e = std::getenv("GVX_TF_ROWS64")) m->tf_rows64 = e[0] == '1';
    if (const char* e = std::getenv("GVX_TF_RESIDENT")) m->tf_resident = e[0] != '0';
    if (const char* e = std::getenv("GVX_AR_RESIDENT_LOOP")) m->ar_resident_loop = e[0] != '0';
    if (const char* e = std::getenv("GVX_TF_LONG_224")) m->tf_long_rows_224 = e[0] != '0';
    if (const char* e = std::getenv("GVX_PA_DEPTH")) m->pa_depth = std::atoi(e) == 6 ? 6 : 4;
    if (const char* e = std::getenv("GVX_HANDOFF_SPIN_LIMIT")) m->spin_limit = (unsigned)std::strtoul(e, nullptr, 10);
    if (const char* e = std::getenv("GVX_DEBUG_SKIP_RESIDENT")) m->debug_skip_resident = e[0] == '1';
    *out = m;
    return GVX_OK;
}

void gvx_model_destroy(gvx_model* m) {
    if (m) {
        if (m->gather_off) (void)hipFree(m->gather_off);
        if (m->gather_tid) (void)hipFree(m->gather_tid);
    }
    if (!m) return;
    if (m->ev_valid)
        for (auto& e : m->ev) (void)hipEventDestroy(e);
    for (auto& e : m->kev) (void)hipEventDestroy(e);
    m->drop_graphs();
    if (m->cap_stream) (void)hipStreamDestroy(m->cap_stream);
    // (pa_stream belongs to the process-wide side-stream pool)
    if (m->pa_fork) (void)hipEventDestroy(m->pa_fork);
    if (m->pa_join) (void)hipEventDestroy(m->pa_join);
    if (m->enc_mid) (void)hipEventDestroy(m->enc_mid);
    if (m->ar_done_host) (void)hipHostFree(m->ar_done_host);
    for (hipEvent_t e : m->ar_ev)
        if (e) (void)hipEventDestroy(e);
    delete m;
}

size_t gvx_model_blob_bytes(const gvx_model* m) { return m ? m->blob.total * sizeof(float) : 0; }

int gvx_model_pack_weights(gvx_model* m, const gvx_weight_desc* table, int n, void* host_blob) {
    if (!m || !table || !host_blob) return fail(GVX_ERR_INVALID_ARG, "null argument");
    WeightTable wt;
    for (int i = 0; i < n; ++i) wt.map[table[i].name] = &table[i];
    const gvx_dims& d = m->d;
    const Blob& bl = m->blob;
    float* out = reinterpret_cast<float*>(host_blob);
    std::memset(out, 0, bl.total * sizeof(float));
    const int E = d.embed_dim, H = E / 2, M = d.n_mels, P = d.prenet_dim, A = d.att_rnn_dim, D = d.dec_rnn_dim, a = d.att_dim;
    int rc = GVX_OK;
    const float* src;

    if (!(src = wt.get("embedding.weight", (int64_t)d.n_tokens * E, &rc))) return rc;
    std::memcpy(out + bl.emb, src, sizeof(float) * d.n_tokens * E);

    for (int i = 0; i < d.enc_n_conv; ++i) {
        rc = pack_conv(wt, "encoder.convolutions." + std::to_string(i), E, E, d.enc_kernel, out + bl.enc_w[i], out + bl.enc_b[i]);
        if (rc != GVX_OK) return rc;
    }
    {   // encoder BiLSTM: input projection rows [dir][4*j+gate], recurrent part in fragment order
        const char* sfx[2] = {"", "_reverse"};
        for (int dir = 0; dir < 2; ++dir) {
            std::vector<float> wcat;
            std::vector<float> bias(4 * H);
            rc = pack_lstm(wt, std::string("encoder.lstm.weight_ih_l0") + sfx[dir], std::string("encoder.lstm.weight_hh_l0") + sfx[dir],
                           std::string("encoder.lstm.bias_ih_l0") + sfx[dir], std::string("encoder.lstm.bias_hh_l0") + sfx[dir], H, E, &wcat, bias.data());
            if (rc != GVX_OK) return rc;
            std::vector<float> whh((size_t)4 * H * H);
            for (int r = 0; r < 4 * H; ++r) {
                std::memcpy(out + bl.enc_wih + ((size_t)dir * 4 * H + r) * E, &wcat[(size_t)r * (E + H)], sizeof(float) * E);
                std::memcpy(&whh[(size_t)r * H], &wcat[(size_t)r * (E + H) + E], sizeof(float) * H);
            }
            std::memcpy(out + bl.enc_bih + (size_t)dir * 4 * H, bias.data(), sizeof(float) * 4 * H);
            pack_frag(whh, 4 * H, H, out + bl.enc_whh_frag[dir]);
        }
    }
    {   // Prenet (no bias)
        if (!(src = wt.get("decoder.prenet.layers.0.linear_layer.weight", (int64_t)P * M, &rc))) return rc;
        std::memcpy(out + bl.pre_w0, src, sizeof(float) * P * M);
        for (int j = 0; j < P; ++j)
            for (int k = 0; k < M; ++k) out[bl.pre_w0_t + (size_t)k * P + j] = src[(size_t)j * M + k];
        if (!(src = wt.get("decoder.prenet.layers.1.linear_layer.weight", (int64_t)P * P, &rc))) return rc;
        std::memcpy(out + bl.pre_w1, src, sizeof(float) * P * P);
        for (int j = 0; j < P; ++j)
            for (int k = 0; k < P; ++k) out[bl.pre_w1_t + (size_t)k * P + j] = src[(size_t)j * P + k];
    }
    {   // attention LSTM: x = [prenet ; context ; h_a]
        std::vector<float> wcat;
        rc = pack_lstm(wt, "decoder.attention_rnn.weight_ih", "decoder.attention_rnn.weight_hh", "decoder.attention_rnn.bias_ih",
                       "decoder.attention_rnn.bias_hh", A, P + E, &wcat, out + bl.att_bias);
        if (rc != GVX_OK) return rc;
        pack_frag(wcat, 4 * A, P + E + A, out + bl.att_frag);
        for (int n = 0; n < 4 * A; ++n) std::memcpy(out + bl.att_wpre + (size_t)n * P, &wcat[(size_t)n * (P + E + A)], sizeof(float) * P);
    }
    {   // attention layer
        const std::string att = "decoder.attention_layer.";
        if (!(src = wt.get(att + "query_layer.linear_layer.weight", (int64_t)a * A, &rc))) return rc;
        for (int t = 0; t < A / 8; ++t)
            for (int dd = 0; dd < a; ++dd)
                for (int jj = 0; jj < 8; ++jj) out[bl.wq_t + ((size_t)t * a + dd) * 8 + jj] = src[(size_t)dd * A + t * 8 + jj];
        if (!(src = wt.get(att + "memory_layer.linear_layer.weight", (int64_t)a * E, &rc))) return rc;
        std::memcpy(out + bl.wmem, src, sizeof(float) * a * E);
        if (!(src = wt.get(att + "v.linear_layer.weight", a, &rc))) return rc;
        std::memcpy(out + bl.v, src, sizeof(float) * a);
        const int64_t nconv = (int64_t)d.att_loc_filters * 2 * d.att_loc_kernel;
        if (!(src = wt.get(att + "location_layer.location_conv.conv.weight", nconv, &rc))) return rc;
        for (int c = 0; c < d.att_loc_filters; ++c)
            for (int ck = 0; ck < 2 * d.att_loc_kernel; ++ck) out[bl.loc_conv + (size_t)ck * 32 + c] = src[(size_t)c * 2 * d.att_loc_kernel + ck];
        if (!(src = wt.get(att + "location_layer.location_dense.linear_layer.weight", (int64_t)a * d.att_loc_filters, &rc))) return rc;
        for (int dd = 0; dd < a; ++dd)
            for (int c = 0; c < d.att_loc_filters; ++c) out[bl.loc_dense + ((size_t)(c >> 2) * a + dd) * 4 + (c & 3)] = src[(size_t)dd * d.att_loc_filters + c];
    }
    {   // decoder LSTM: x = [h_a ; context ; h_d]
        std::vector<float> wcat;
        rc = pack_lstm(wt, "decoder.decoder_rnn.weight_ih", "decoder.decoder_rnn.weight_hh", "decoder.decoder_rnn.bias_ih",
                       "decoder.decoder_rnn.bias_hh", D, A + E, &wcat, out + bl.dec_bias);
        if (rc != GVX_OK) return rc;
        pack_frag(wcat, 4 * D, A + E + D, out + bl.dec_frag);
    }
    {   // mel + gate projection, rows 0..M-1 = linear_projection, row M = gate_layer; x = [h_d ; context]
        const int K = D + E;
        std::vector<float> w((size_t)(M + 1) * K);
        if (!(src = wt.get("decoder.linear_projection.linear_layer.weight", (int64_t)M * K, &rc))) return rc;
        std::memcpy(w.data(), src, sizeof(float) * M * K);
        if (!(src = wt.get("decoder.gate_layer.linear_layer.weight", K, &rc))) return rc;
        std::memcpy(w.data() + (size_t)M * K, src, sizeof(float) * K);
        std::memcpy(out + bl.proj_w, w.data(), sizeof(float) * w.size());
        pack_frag(w, M + 1, K, out + bl.proj_frag);
        // autoregressive mode: the h_d columns tile-major [D/8][PSB][8] (rows past M are zero) for the partial products the
        // decoder-LSTM tiles emit, the context columns as their own fragment matrix
        const int PSBp = (M + 1 + 7) & ~7;
        for (int t = 0; t < D / 8; ++t)
            for (int n = 0; n < PSBp; ++n)
                for (int jj = 0; jj < 8; ++jj)
                    out[bl.proj_hd_t + ((size_t)t * PSBp + n) * 8 + jj] = n <= M ? w[(size_t)n * K + t * 8 + jj] : 0.f;
        std::vector<float> wc((size_t)(M + 1) * E);
        for (int n = 0; n <= M; ++n) std::memcpy(&wc[(size_t)n * E], &w[(size_t)n * K + D], sizeof(float) * E);
        pack_frag(wc, M + 1, E, out + bl.proj_ctx_frag);
        // ... and once more tile-major [E/4][PSB][4]: four context columns ride on the projection slab of each decoder-LSTM
        // tile (skinny.hip, extra slab terms) when the tile counts match (E / 4 == D / 8)
        for (int t = 0; t < E / 4; ++t)
            for (int n = 0; n < PSBp; ++n)
                for (int jj = 0; jj < 4; ++jj)
                    out[bl.proj_ctx_t + ((size_t)t * PSBp + n) * 4 + jj] = n <= M ? w[(size_t)n * K + D + t * 4 + jj] : 0.f;
        if (!(src = wt.get("decoder.linear_projection.linear_layer.bias", M, &rc))) return rc;
        std::memcpy(out + bl.proj_b, src, sizeof(float) * M);
        if (!(src = wt.get("decoder.gate_layer.linear_layer.bias", 1, &rc))) return rc;
        out[bl.proj_b + M] = src[0];
    }
    for (int i = 0; i < d.postnet_n_conv; ++i) {
        const int cin = i == 0 ? M : d.postnet_dim, cout = i == d.postnet_n_conv - 1 ? M : d.postnet_dim;
        rc = pack_conv(wt, "postnet.convolutions." + std::to_string(i), cout, cin, d.postnet_kernel, out + bl.post_w[i], out + bl.post_b[i]);
        if (rc != GVX_OK) return rc;
    }
    return GVX_OK;
}

}  // extern "C"

namespace {

constexpr int PACK_MAX_TENSORS = 192;
struct PackSources { const float* p[PACK_MAX_TENSORS]; };

__global__ void pack_gather_kernel(PackSources src, const int32_t* off, const uint8_t* tid, long n, float* blob) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int t = tid[i];
        if (t == 255) continue;   // written by the fold / bias kernels below
        blob[i] = t ? src.p[t - 1][off[i]] : 0.f;
    }
}
// pack_conv on the device: [Cout][Cin][k] -> [Cout][k][Cin] with the eval-mode BatchNorm scale folded in (double, like the host)
__global__ void pack_conv_fold_kernel(const float* w, const float* b, const float* g, const float* beta, const float* mu, const float* var,
                                      int cout, int cin, int k, float* w_out, float* b_out) {
    const long n = (long)cout * cin * k;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cin), kk = (int)((i / cin) % k), co = (int)(i / ((long)cin * k));
        const double scale = (double)g[co] / sqrt((double)var[co] + BN_EPS);
        w_out[i] = (float)((double)w[((long)co * cin + ci) * k + kk] * scale);
        if (ci == 0 && kk == 0) b_out[co] = (float)(((double)b[co] - (double)mu[co]) * scale + (double)beta[co]);
    }
}
// bias of an LSTM in packed row order: out[4 j + q] = b_ih[q H + j] + b_hh[q H + j]
__global__ void pack_lstm_bias_kernel(const float* bih, const float* bhh, int Hd, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4 * Hd) { const int j = i >> 2, q = i & 3; out[i] = bih[q * Hd + j] + bhh[q * Hd + j]; }
}

struct DevTable {
    std::unordered_map<std::string, int> idx;
    const gvx_weight_desc* t;
    const float* get(const std::string& name, int64_t numel, int* rc) const {
        auto it = idx.find(name);
        if (it == idx.end()) { *rc = fail(GVX_ERR_MISSING_WEIGHT, "missing weight '%s'", name.c_str()); return nullptr; }
        if (t[it->second].numel != numel) { *rc = fail(GVX_ERR_SHAPE, "weight '%s' has %lld elements, expected %lld", name.c_str(), (long long)t[it->second].numel, (long long)numel); return nullptr; }
        return t[it->second].data;
    }
};

// Build (or re-use) the gather map of `table`'s layout.  Two runs of the host packer over stand-ins whose floats carry the
// low / high 12 bits of their own index tell where each blob float comes from; the regions the packer COMPUTES (BatchNorm
// folds, bias sums) are marked 255 and written by their own kernels.
int ensure_gather_map(gvx_model* m, const gvx_weight_desc* table, int n) {
    bool same = m->gather_off && (int)m->gather_names.size() == n;
    for (int i = 0; same && i < n; ++i) same = m->gather_names[i] == table[i].name && m->gather_numel[i] == table[i].numel;
    if (same) return GVX_OK;
    if (n > PACK_MAX_TENSORS || n > 254) return fail(GVX_ERR_UNSUPPORTED, "pack_weights_device: more than %d tensors", 254);
    const size_t total = m->blob.total;
    std::vector<std::vector<float>> lo(n), hi(n);
    std::vector<gvx_weight_desc> tl(n), th(n);
    for (int i = 0; i < n; ++i) {
        if (table[i].numel < 0 || table[i].numel >= (int64_t)1 << 31) return fail(GVX_ERR_UNSUPPORTED, "pack_weights_device: tensor too large");
        lo[i].resize((size_t)table[i].numel); hi[i].resize((size_t)table[i].numel);
        for (int64_t e = 0; e < table[i].numel; ++e) { lo[i][e] = (float)((e & 4095) + 1); hi[i][e] = (float)((e >> 12) * 256 + i + 1); }
        tl[i] = gvx_weight_desc{table[i].name, lo[i].data(), table[i].numel};
        th[i] = gvx_weight_desc{table[i].name, hi[i].data(), table[i].numel};
    }
    std::vector<float> bl(total), bh(total);
    int rc = gvx_model_pack_weights(m, tl.data(), n, bl.data());
    if (rc != GVX_OK) return rc;
    rc = gvx_model_pack_weights(m, th.data(), n, bh.data());
    if (rc != GVX_OK) return rc;
    std::vector<int32_t> off(total);
    std::vector<uint8_t> tid(total);
    for (size_t i = 0; i < total; ++i) {
        if (bl[i] == 0.f && bh[i] == 0.f) { off[i] = 0; tid[i] = 0; continue; }
        const long h = (long)bh[i] - 1, l = (long)bl[i] - 1;
        const int t = (int)(h % 256);
        off[i] = (int32_t)((h / 256) * 4096 + l);
        tid[i] = (uint8_t)(t + 1);
    }
    // computed regions
    const gvx_dims& d = m->d;
    const Blob& b = m->blob;
    auto mark = [&](size_t o, size_t cnt) { std::fill(tid.begin() + o, tid.begin() + o + cnt, (uint8_t)255); };
    const int E = d.embed_dim, H = E / 2, M = d.n_mels;
    for (int i = 0; i < d.enc_n_conv; ++i) { mark(b.enc_w[i], (size_t)E * d.enc_kernel * E); mark(b.enc_b[i], E); }
    for (int i = 0; i < d.postnet_n_conv; ++i) {
        const int cin = i == 0 ? M : d.postnet_dim, cout = i == d.postnet_n_conv - 1 ? M : d.postnet_dim;
        mark(b.post_w[i], (size_t)cout * d.postnet_kernel * cin); mark(b.post_b[i], cout);
    }
    mark(b.enc_bih, (size_t)8 * H); mark(b.att_bias, (size_t)4 * d.att_rnn_dim); mark(b.dec_bias, (size_t)4 * d.dec_rnn_dim);
    // sanity: every gathered float points inside its tensor
    for (size_t i = 0; i < total; ++i)
        if (tid[i] && tid[i] != 255 && (tid[i] > n || off[i] < 0 || off[i] >= table[tid[i] - 1].numel))
            return fail(GVX_ERR_UNSUPPORTED, "pack_weights_device: gather map is inconsistent at blob float %zu", i);
    if (m->gather_off) { (void)hipFree(m->gather_off); m->gather_off = nullptr; }
    if (m->gather_tid) { (void)hipFree(m->gather_tid); m->gather_tid = nullptr; }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->gather_off), total * sizeof(int32_t)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->gather_tid), total));
    HIP_TRY(hipMemcpy(m->gather_off, off.data(), total * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m->gather_tid, tid.data(), total, hipMemcpyHostToDevice));
    m->gather_names.clear(); m->gather_numel.clear();
    for (int i = 0; i < n; ++i) { m->gather_names.push_back(table[i].name); m->gather_numel.push_back(table[i].numel); }
    return GVX_OK;
}

}  // namespace

extern "C" {

int gvx_model_pack_weights_device(gvx_model* m, const gvx_weight_desc* table, int n, void* device_blob, void* stream) {
    if (!m || !table || !device_blob || n < 1) return fail(GVX_ERR_INVALID_ARG, "null argument");
    if (reinterpret_cast<uintptr_t>(device_blob) & 255) return fail(GVX_ERR_INVALID_ARG, "blob must be 256-byte aligned");
    int rc = ensure_gather_map(m, table, n);
    if (rc != GVX_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    float* out = reinterpret_cast<float*>(device_blob);
    PackSources src{};
    for (int i = 0; i < n; ++i) src.p[i] = table[i].data;
    const long total = (long)m->blob.total;
    hipLaunchKernelGGL(pack_gather_kernel, dim3(4096), dim3(256), 0, s, src, m->gather_off, m->gather_tid, total, out);
    DevTable dt; dt.t = table;
    for (int i = 0; i < n; ++i) dt.idx[table[i].name] = i;
    const gvx_dims& d = m->d;
    const Blob& bl = m->blob;
    const int E = d.embed_dim, H = E / 2, M = d.n_mels;
    auto conv = [&](const std::string& prefix, int cout, int cin, int k, size_t w_off, size_t b_off) -> int {
        int r = GVX_OK;
        const float* w = dt.get(prefix + ".0.conv.weight", (int64_t)cout * cin * k, &r); if (!w) return r;
        const float* b = dt.get(prefix + ".0.conv.bias", cout, &r); if (!b) return r;
        const float* g = dt.get(prefix + ".1.weight", cout, &r); if (!g) return r;
        const float* beta = dt.get(prefix + ".1.bias", cout, &r); if (!beta) return r;
        const float* mu = dt.get(prefix + ".1.running_mean", cout, &r); if (!mu) return r;
        const float* var = dt.get(prefix + ".1.running_var", cout, &r); if (!var) return r;
        const long cnt = (long)cout * cin * k;
        hipLaunchKernelGGL(pack_conv_fold_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, w, b, g, beta, mu, var, cout, cin, k, out + w_off, out + b_off);
        return GVX_OK;
    };
    for (int i = 0; i < d.enc_n_conv; ++i)
        if ((rc = conv("encoder.convolutions." + std::to_string(i), E, E, d.enc_kernel, bl.enc_w[i], bl.enc_b[i])) != GVX_OK) return rc;
    for (int i = 0; i < d.postnet_n_conv; ++i) {
        const int cin = i == 0 ? M : d.postnet_dim, cout = i == d.postnet_n_conv - 1 ? M : d.postnet_dim;
        if ((rc = conv("postnet.convolutions." + std::to_string(i), cout, cin, d.postnet_kernel, bl.post_w[i], bl.post_b[i])) != GVX_OK) return rc;
    }
    auto bias = [&](const std::string& bih_n, const std::string& bhh_n, int Hd, size_t o) -> int {
        int r = GVX_OK;
        const float* bih = dt.get(bih_n, 4 * Hd, &r); if (!bih) return r;
        const float* bhh = dt.get(bhh_n, 4 * Hd, &r); if (!bhh) return r;
        hipLaunchKernelGGL(pack_lstm_bias_kernel, dim3((4 * Hd + 255) / 256), dim3(256), 0, s, bih, bhh, Hd, out + o);
        return GVX_OK;
    };
    if ((rc = bias("encoder.lstm.bias_ih_l0", "encoder.lstm.bias_hh_l0", H, bl.enc_bih)) != GVX_OK) return rc;
    if ((rc = bias("encoder.lstm.bias_ih_l0_reverse", "encoder.lstm.bias_hh_l0_reverse", H, bl.enc_bih + (size_t)4 * H)) != GVX_OK) return rc;
    if ((rc = bias("decoder.attention_rnn.bias_ih", "decoder.attention_rnn.bias_hh", d.att_rnn_dim, bl.att_bias)) != GVX_OK) return rc;
    if ((rc = bias("decoder.decoder_rnn.bias_ih", "decoder.decoder_rnn.bias_hh", d.dec_rnn_dim, bl.dec_bias)) != GVX_OK) return rc;
    HIP_TRY(hipGetLastError());
    return GVX_OK;
}

int gvx_model_bind_blob(gvx_model* m, const void* device_blob) {
    if (!m || !device_blob) return fail(GVX_ERR_INVALID_ARG, "null argument");
    if (reinterpret_cast<uintptr_t>(device_blob) & 255) return fail(GVX_ERR_INVALID_ARG, "blob must be 256-byte aligned");
    // captured step loops bake blob addresses into their kernel nodes: a new blob invalidates every cached graph (the
    // key carries the blob pointer as well, so a stale graph could not be selected even if one survived)
    if (m->dev_blob != device_blob) m->drop_graphs();
    m->dev_blob = reinterpret_cast<const float*>(device_blob);
    HIP_TRY(gemm_init());
    HIP_TRY(skinny_init());
    HIP_TRY(attention_init());
    HIP_TRY(attention_persistent_init());
    HIP_TRY(decoder_resident_init());
    return GVX_OK;
}

size_t gvx_workspace_bytes(const gvx_model* m, int B, int L, int T) {
    if (!m || B < 1 || L < 1 || T < 1) return 0;
    return make_ws_plan(m, B, L, T).total;
}

size_t gvx_workspace_bytes_autoregressive(const gvx_model* m, int B, int L, int max_steps) {
    if (!m || B < 1 || L < 1 || max_steps < 1) return 0;
    return make_ws_plan(m, B, L, max_steps, WS_AUTOREGRESSIVE).total;
}

}  // extern "C"

// =====================================================================================================
namespace {

int check_common(const gvx_model* m, int B, int L, int T, void* ws, size_t ws_bytes, int mode = WS_TEACHER_FORCED) {
    if (!m) return fail(GVX_ERR_INVALID_ARG, "null model");
    if (!m->dev_blob) return fail(GVX_ERR_STATE, "weights not bound (call gvx_model_bind_blob)");
    if (B < 1 || B > 64) return fail(GVX_ERR_UNSUPPORTED, "batch %d not in [1, 64] (shard larger batches across calls / GPUs)", B);
    if (L < 1 || T < 1) return fail(GVX_ERR_INVALID_ARG, "L and T must be >= 1");
    if (!ws) return fail(GVX_ERR_WORKSPACE, "null workspace");
    if (reinterpret_cast<uintptr_t>(ws) & 255) return fail(GVX_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    const size_t need = make_ws_plan(m, B, L, T, mode).total;
    if (ws_bytes < need) return fail(GVX_ERR_WORKSPACE, "workspace too small: %zu < %zu bytes", ws_bytes, need);
    if (!attention_supported(L, m->d.att_dim, m->d.att_loc_filters, m->d.att_loc_kernel, m->d.embed_dim))
        return fail(GVX_ERR_UNSUPPORTED, "L = %d is too long for the attention kernels' LDS budget", L);
    return GVX_OK;
}

// conv stack on channels-last halo buffers: in -> (ping/pong) ; returns pointer of the final output buffer
int conv_layer(const gvx_model* m, const float* in, float* out, int B, int T, int cin, int cout, int k, size_t w_off, size_t b_off,
               int act, int out_halo, hipStream_t s) {
    const int p = (k - 1) / 2;
    GemmParams g{};
    g.A = in; g.amap = RowMap{T, (long)(T + 2 * p) * cin, (long)cin};
    g.W = m->dev_blob + w_off; g.ldw = (long)k * cin;
    g.C = out + (long)out_halo * cout; g.cmap = RowMap{T, (long)(T + 2 * out_halo) * cout, (long)cout};
    g.bias = m->dev_blob + b_off;
    g.M = B * T; g.N = cout; g.K = k * cin; g.act = act;
    HIP_TRY(launch_gemm(g, s));
    return GVX_OK;
}

// conv_out (training mode only): the output of the convolution stack, [B, E, L] in the reference's layout, computed by the
// caller with batch statistics and dropout (gvx_conv_bn_act_train_forward); the embedding and the folded-BatchNorm
// convolutions are then skipped and only the BiLSTM part runs
int encoder_impl(gvx_model* m, const int64_t* tokens, const int32_t* lengths, int B, int L, float* memory_out, void* ws,
                 const WsPlan& wp, hipStream_t s, const float* conv_out = nullptr, float* c_seq_out = nullptr, float* xg_out = nullptr,
                 hipEvent_t dense_done = nullptr) {   // dense_done: recorded on s behind the convolutions (see below)
    const gvx_dims& d = m->d;
    const int E = d.embed_dim, H = E / 2, pe = (d.enc_kernel - 1) / 2;
    float* xa = ws_ptr<float>(ws, wp.xa);
    float* xb = ws_ptr<float>(ws, wp.xb);
    float* xg = ws_ptr<float>(ws, wp.xg);
    float* enc_h = ws_ptr<float>(ws, wp.enc_h);
    float* enc_c = ws_ptr<float>(ws, wp.enc_c);
    int* flags = ws_ptr<int>(ws, wp.flags);
    float* cur = xa;
    float* nxt = xb;
    if (conv_out) {
        HIP_TRY(launch_to_channels_last(conv_out, xa, B, E, L, pe, nullptr, s));
    } else {
        // (the token-error word is sticky: raised here, cleared only by gvx_workspace_status - a later chunk on the same
        // workspace must not wipe an earlier chunk's error)
        HIP_TRY(launch_embed(tokens, m->dev_blob + m->blob.emb, d.n_tokens, xa, xb, B, L, E, pe, flags + FLAG_TOKEN, s));   // (+ the halo rows of xa and xb)
        for (int i = 0; i < d.enc_n_conv; ++i) {
            if (dense_done && i == m->enc_fork_after) { HIP_TRY(hipEventRecord(dense_done, s)); dense_done = nullptr; }
            int rc = conv_layer(m, cur, nxt, B, L, E, E, d.enc_kernel, m->blob.enc_w[i], m->blob.enc_b[i], ACT_RELU, pe, s);
            if (rc != GVX_OK) return rc;
            float* t = cur; cur = nxt; nxt = t;
        }
    }
    // (the fork event of the caller's other dense products, if the convolution loop has not recorded it: training mode, or
    // GVX_ENC_FORK_AFTER >= the number of convolutions)
    if (dense_done) HIP_TRY(hipEventRecord(dense_done, s));
    {   // LSTM input projection for both directions: xg[b][l][dir*4H + 4j+gate]
        GemmParams g{};
        g.A = cur + (long)pe * E; g.amap = RowMap{L, (long)(L + 2 * pe) * E, (long)E};
        g.W = m->dev_blob + m->blob.enc_wih; g.ldw = E;
        g.C = xg; g.cmap = RowMap{B * L, 0, (long)8 * H};
        g.bias = m->dev_blob + m->blob.enc_bih;
        g.M = B * L; g.N = 8 * H; g.K = E; g.act = ACT_NONE;
        HIP_TRY(launch_gemm(g, s));
    }
    const bool resident = m->enc_persistent && encoder_persistent_supported(B, H);
    if (!resident) HIP_TRY(zero_async(enc_h, (size_t)4 * B * H * sizeof(float), s));
    if (!resident) HIP_TRY(zero_async(enc_c, (size_t)2 * B * H * sizeof(float), s));   // (the resident kernel keeps the cells in registers)
    // The L recurrence launches only reference workspace operands: lengths are copied next to them and the sequence output
    // goes to the workspace-resident memory buffer (copied to the caller's tensor afterwards when that is a different
    // one), so the key of the cached hipGraph does not depend on a freshly allocated output tensor.
    float* mem_ws = ws_ptr<float>(ws, wp.memory);
    if (!resident) HIP_TRY(zero_async(mem_ws, (size_t)B * L * E * sizeof(float), s));   // (the resident kernel writes every position)
    const int32_t* len_ws = nullptr;
    if (lengths) {
        int32_t* lc = ws_ptr<int32_t>(ws, wp.len_copy);
        HIP_TRY(hipMemcpyAsync(lc, lengths, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        len_ws = lc;
    }
    auto enqueue = [&](hipStream_t st) -> int {
        for (int step = 0; step < L; ++step) {
            SkinnyJob jobs[2];
            for (int dir = 0; dir < 2; ++dir) {
                SkinnyJob& J = jobs[dir];
                std::memset(&J, 0, sizeof J);
                float* h_cur = enc_h + ((size_t)dir * 2 + (step & 1)) * B * H;
                float* h_nxt = enc_h + ((size_t)dir * 2 + ((step + 1) & 1)) * B * H;
                J.Wp = m->dev_blob + m->blob.enc_whh_frag[dir];
                J.x[0] = XSeg{h_cur, H};
                J.N = 4 * H; J.nkg = H / 8; J.mode = 0; J.B = B;
                J.c = enc_c + (size_t)dir * B * H;
                J.h_out = h_nxt;
                J.addend = xg + (size_t)dir * 4 * H; J.add_bs = (long)L * 8 * H; J.add_ts = 8 * H;
                J.lengths = len_ws; J.step = step; J.reverse = dir; J.seq_len = L;
                J.seq_out = mem_ws + (size_t)dir * H; J.seq_bs = (long)L * E; J.seq_ts = E;
                if (c_seq_out) J.c_seq_out = c_seq_out + (size_t)dir * H;   // (training tape; the loop is then not replayed from a graph)
                J.h_prev = h_cur;
            }
            HIP_TRY(launch_skinny(jobs, 2, SK_ENCODER, st));
        }
        return GVX_OK;
    };
    if (resident) {
        // one resident launch for the whole recurrence (skinny.hip, encoder_lstm_persistent_kernel); a hand-off that times out
        // leaves NaN in the encoder output and raises the sticky status word, like the resident decoder loops
        unsigned* sync = ws_ptr<unsigned>(ws, wp.sync);
        void* const zp[2] = {enc_h, sync + HANDOFF_TIMEOUT};   // the exchange buffers (generation bits 0), the call's time-out word
        const size_t zb[2] = {(size_t)4 * B * H * sizeof(float), sizeof(unsigned)};
        HIP_TRY(launch_zero_many(zp, zb, 2, s));
        EncPersistParams ep{};
        ep.Wp[0] = m->dev_blob + m->blob.enc_whh_frag[0]; ep.Wp[1] = m->dev_blob + m->blob.enc_whh_frag[1];
        ep.xg = xg; ep.lengths = len_ws; ep.hx = enc_h; ep.seq_out = mem_ws; ep.c_seq_out = c_seq_out;
        ep.sync = sync; ep.spin_limit = m->spin_limit; ep.B = B; ep.L = L; ep.H = H;
        { const char* e = std::getenv("GVX_DEBUG_ENC_SKIP_BLOCK"); ep.debug_skip_block = e ? std::atoi(e) : -1; }   // (tests: forced time-out)
        HIP_TRY(launch_encoder_persistent(ep, s));
        float* outs[2] = {mem_ws, c_seq_out};
        const size_t counts[2] = {(size_t)B * L * E, (size_t)B * L * E};
        HIP_TRY(launch_poison_on_timeout(sync + HANDOFF_TIMEOUT, flags + FLAG_TIMEOUT, outs, counts, c_seq_out ? 2 : 1, s));
    } else {
        const gvx_model::LoopKey key{ws, mem_ws, m->dev_blob, B, L, 0, lengths != nullptr};
        const int rc = run_chunk(m, m->use_graph && !c_seq_out ? touch_graph_set(m, m->enc_graphs, key) : nullptr, 0, s, enqueue);
        if (rc != GVX_OK) return rc;
    }
    if (xg_out) HIP_TRY(hipMemcpyAsync(xg_out, xg, (size_t)B * L * 8 * H * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (memory_out != mem_ws)
        HIP_TRY(hipMemcpyAsync(memory_out, mem_ws, (size_t)B * L * E * sizeof(float), hipMemcpyDeviceToDevice, s));
    return GVX_OK;
}

struct DecoderBuffers {
    float *pm, *frames, *pre1, *prenet, *h_a, *c_a, *c_d, *hc, *w_cum, *q_slab, *proj, *energies, *align_tm, *loc, *p_slab, *p_ctx;
    float *att_part, *dec_part, *pre_gate;
    int32_t* len_copy;
};

DecoderBuffers decoder_buffers(void* ws, const WsPlan& wp) {
    DecoderBuffers b;
    b.pm = ws_ptr<float>(ws, wp.pm); b.frames = ws_ptr<float>(ws, wp.frames); b.pre1 = ws_ptr<float>(ws, wp.pre1);
    b.prenet = ws_ptr<float>(ws, wp.prenet); b.h_a = ws_ptr<float>(ws, wp.h_a); b.c_a = ws_ptr<float>(ws, wp.c_a);
    b.c_d = ws_ptr<float>(ws, wp.c_d); b.hc = ws_ptr<float>(ws, wp.hc); b.w_cum = ws_ptr<float>(ws, wp.w_cum);
    b.q_slab = ws_ptr<float>(ws, wp.q_slab); b.proj = ws_ptr<float>(ws, wp.proj); b.energies = ws_ptr<float>(ws, wp.energies);
    b.align_tm = ws_ptr<float>(ws, wp.align_tm); b.len_copy = ws_ptr<int32_t>(ws, wp.len_copy);
    b.loc = ws_ptr<float>(ws, wp.loc);
    b.p_slab = ws_ptr<float>(ws, wp.p_slab); b.p_ctx = ws_ptr<float>(ws, wp.p_ctx);
    b.att_part = ws_ptr<float>(ws, wp.att_part); b.dec_part = ws_ptr<float>(ws, wp.dec_part);
    b.pre_gate = ws_ptr<float>(ws, wp.pre_gate);
    return b;
}

// Decoder.initialize_decoder_states (models/tts/tacotron2.py:303-315): zero states + memory projection
int decoder_init_states(gvx_model* m, const float* memory, int B, int L, const DecoderBuffers& db, hipStream_t s) {
    const gvx_dims& d = m->d;
    const int E = d.embed_dim, A = d.att_rnn_dim, D = d.dec_rnn_dim;
    void* const zp[5] = {db.h_a, db.c_a, db.c_d, db.hc /* slot 0 */, db.w_cum};
    const size_t zb[5] = {(size_t)RS_HA_SLOTS * B * A * sizeof(float), (size_t)B * A * sizeof(float), (size_t)B * D * sizeof(float),
                          (size_t)B * (D + E) * sizeof(float), (size_t)B * L * sizeof(float)};
    HIP_TRY(launch_zero_many(zp, zb, 5, s));
    GemmParams g{};
    g.A = memory; g.amap = RowMap{B * L, 0, (long)E};
    g.W = m->dev_blob + m->blob.wmem; g.ldw = E;
    g.C = db.pm; g.cmap = RowMap{B * L, 0, (long)d.att_dim};
    g.M = B * L; g.N = d.att_dim; g.K = E; g.act = ACT_NONE;
    HIP_TRY(launch_gemm(g, s));
    return GVX_OK;
}

// attention LSTM of step t: x = [prenet(t) ; ctx(t-1) ; h_a(t-1)] (all blocked vectors)
void fill_att_job(const gvx_model* m, SkinnyJob& J, const float* prenet_t, int t, int B, const DecoderBuffers& db) {
    const gvx_dims& d = m->d;
    const int E = d.embed_dim, P = d.prenet_dim, A = d.att_rnn_dim, D = d.dec_rnn_dim;
    std::memset(&J, 0, sizeof J);
    const float* hc_t = db.hc + (size_t)t * B * (D + E);
    J.Wp = m->dev_blob + m->blob.att_frag; J.bias = m->dev_blob + m->blob.att_bias;
    J.x[0] = XSeg{prenet_t, P};
    J.x[1] = XSeg{hc_t + (size_t)D * B, E};                   // context part of slot t: k-groups D/8 ...
    J.x[2] = XSeg{db.h_a + (size_t)(t & 1) * B * A, A};       // h_a of step t-1
    J.N = 4 * A; J.nkg = (P + E + A) / 8; J.mode = 0; J.B = B;
    J.c = db.c_a;
    J.h_out = db.h_a + (size_t)((t + 1) & 1) * B * A;
    J.Wq_t = m->dev_blob + m->blob.wq_t; J.q_slab = db.q_slab; J.att_dim = d.att_dim;
}

// decoder LSTM of step t: x = [h_a(t) ; ctx(t) ; h_d(t-1)], writes h_d(t) into hc slot t+1
void fill_dec_job(const gvx_model* m, SkinnyJob& J, int t, int B, const DecoderBuffers& db) {
    const gvx_dims& d = m->d;
    const int E = d.embed_dim, A = d.att_rnn_dim, D = d.dec_rnn_dim;
    std::memset(&J, 0, sizeof J);
    const float* hc_t = db.hc + (size_t)t * B * (D + E);
    float* hc_n = db.hc + (size_t)(t + 1) * B * (D + E);
    J.Wp = m->dev_blob + m->blob.dec_frag; J.bias = m->dev_blob + m->blob.dec_bias;
    J.x[0] = XSeg{db.h_a + (size_t)((t + 1) & 1) * B * A, A};
    J.x[1] = XSeg{hc_n + (size_t)D * B, E};
    J.x[2] = XSeg{hc_t, D};
    J.N = 4 * D; J.nkg = (A + E + D) / 8; J.mode = 0; J.B = B;
    J.c = db.c_d;
    J.h_out = hc_n;
}

// location features for step t's attention, computed inside the LSTM launch of step t from attention(t-1)'s outputs
void fill_loc(const gvx_model* m, LocJob& q, int t, int B, int L, const float* align_base, long align_bs, long align_ts,
              const DecoderBuffers& db) {
    const gvx_dims& d = m->d;
    q.w_prev = t > 0 ? align_base + (size_t)(t - 1) * align_ts : nullptr; q.w_prev_bs = align_bs;
    q.w_cum = db.w_cum;
    q.loc_conv_t = m->dev_blob + m->blob.loc_conv; q.loc_dense_t = m->dev_blob + m->blob.loc_dense;
    q.loc_out = db.loc;
    q.pm = m->attn_one_launch ? db.pm : nullptr;
    q.B = B; q.L = L; q.a = d.att_dim; q.kl = d.att_loc_kernel; q.G = attention_groups(B, L);
}

void fill_attn(const gvx_model* m, AttnParams& p, const float* memory, const int32_t* lengths, int t, int B, int L,
               float* align_out, long align_bs, long align_ts, const DecoderBuffers& db) {
    const gvx_dims& d = m->d;
    const int E = d.embed_dim, D = d.dec_rnn_dim;
    std::memset(&p, 0, sizeof p);
    p.q_slab = db.q_slab; p.n_slabs = d.att_rnn_dim / 8;
    p.w_cum = db.w_cum;
    p.loc = db.loc; p.v = m->dev_blob + m->blob.v;
    p.pm = db.pm; p.memory = memory; p.lengths = lengths;
    p.w_out = align_out + (size_t)t * align_ts; p.w_out_bs = align_bs;
    p.ctx_out = db.hc + (size_t)(t + 1) * B * (D + E) + (size_t)D * B;
    p.energies = db.energies;
    p.B = B; p.L = L; p.a = d.att_dim; p.F = d.att_loc_filters; p.kl = d.att_loc_kernel; p.E = E;
    p.G = m->attn_one_launch ? attention_slices(B, E) : attention_groups(B, L);
}

hipError_t launch_attn(const gvx_model* m, const AttnParams& p, hipStream_t s) {
    return m->attn_one_launch ? launch_attention_step(p, s) : launch_attention(p, s);
}

// whether the teacher-forced loop of this shape runs beside the persistent attention kernel
bool persistent_path(const gvx_model* m, int B, int L) {
    const gvx_dims& d = m->d;
    if (B > 32 && !m->tf_rows64) return false;
    return m->attn_persistent && m->attn_one_launch &&
           attention_persistent_supported(B, L, d.att_dim, d.att_loc_filters, d.att_loc_kernel, d.embed_dim, d.att_rnn_dim, d.dec_rnn_dim);
}

// Side stream of the resident attention kernels: ONE per device, shared by every handle of the process (GVX_SIDE_POOL=2: two,
// dealt round-robin per call, for concurrent resident loops - the opt-in autoregressive lanes).
// Highest priority: HIP keeps separate hardware queues per priority, so this stream can never be dealt the queue of a
// (normal-priority) stream an LSTM chain runs on - the attention kernel would then sit in front of the launches it waits
// for until its spin limit (observed in a process that had created a dozen streams before).  Shared instead of one per
// handle because a process has only ~4 hardware queues, dealt in order of first use: with the null stream and the host
// mirror's two lane streams in use, a second side stream landed on the queue of a stream that feeds it and every other
// forward took 17 instead of 6.7 ms (tools/queue_probe.py, round 3).  Calls that share the stream only serialise their
// resident kernels (the later one starts when the earlier loop has ended, well inside the spin limit).
struct SidePool { hipStream_t s[2] = {nullptr, nullptr}; unsigned next = 0; };
std::mutex g_pool_mutex;
std::unordered_map<int, SidePool> g_side_pools;

// Resident loops take turns on a device.  A loop whose kernels wait for each other (the resident attention kernel beside LSTM
// launches, or beside the resident decoder kernel: 32 + 224 workgroups that must ALL be on the chip) cannot share the chip with a
// second one: dispatched at the same time from two streams, each could get half of its workgroups a CU and both would spin
// until their limits.  So every such loop is enqueued under this mutex, behind an event the previous one recorded at its join -
// ordering on the device, no host wait - and its kernels reach the shared side stream in turn order.
struct ResidentTurn { hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr}; unsigned n = 0; };
std::mutex g_turn_mutex;
std::unordered_map<int, ResidentTurn> g_turns;

int turn_begin(hipStream_t s) {   // caller holds g_turn_mutex
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    ResidentTurn& t = g_turns[dev];
    if (!t.ev[0])
        for (auto& e : t.ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (t.n > 0) HIP_TRY(hipStreamWaitEvent(s, t.ev[(t.n - 1) & 3], 0));
    return GVX_OK;
}
int turn_end(hipStream_t s) {     // caller holds g_turn_mutex
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    ResidentTurn& t = g_turns[dev];
    HIP_TRY(hipEventRecord(t.ev[t.n & 3], s));
    ++t.n;
    return GVX_OK;
}

// the autoregressive decode as two resident kernels (beside attention_persistent_supported for the shape): default layer sizes, a
// handle that has the chip to itself
bool ar_resident_loop_ok(const gvx_model* m, int B, int L) {
    const gvx_dims& d = m->d;
    const int lay = attention_persistent_layout(B, L);
    // (rows of 129-256 tokens take two attention workgroups each: 16 rows of them fit beside the 224 workgroups of the tile kernel)
    return B <= 32 && d.embed_dim / 4 == d.dec_rnn_dim / 8 && m->attn_persistent && m->tf_resident && m->ar_resident_loop &&
           decoder_resident_supported(B, L) && (lay == 1 || (lay == 2 && B <= 16)) && d.prenet_dim == 256 && d.n_mels <= 80 &&
           m->PSB() <= 96 && d.att_dim == 128;
}

int ensure_side_stream(gvx_model* m) {
    if (!m->pa_fork) {
        HIP_TRY(hipEventCreateWithFlags(&m->pa_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&m->pa_join, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&m->enc_mid, hipEventDisableTiming));
    }
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    SidePool& pool = g_side_pools[dev];
    if (!pool.s[0]) {
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        for (auto& st : pool.s) HIP_TRY(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, greatest));
    }
    static const unsigned pool_size = [] { const char* e = std::getenv("GVX_SIDE_POOL"); return e && e[0] == '2' ? 2u : 1u; }();
    m->pa_stream = pool.s[pool.next++ % pool_size];
    return GVX_OK;
}

// Prenet over all T+1 frames at once (models/tts/tacotron2.py:370-373) and, for the persistent-attention loop, the Prenet
// columns of the attention LSTM applied to all steps.  Depends on the mel input and the weights only: the fused forward runs
// it on the side stream while the encoder's (latency-bound) recurrence has the chip to itself.
int decoder_prenet_part(gvx_model* m, int B, int L, const float* mel_in, int T, const uint8_t* keep_masks, void* ws, const WsPlan& wp,
                        hipStream_t s) {
    const gvx_dims& d = m->d;
    const int M = d.n_mels, P = d.prenet_dim;
    const DecoderBuffers db = decoder_buffers(ws, wp);
    HIP_TRY(zero_async(db.frames, (size_t)B * M * sizeof(float), s));  // go-frame
    HIP_TRY(launch_frames_from_mel(mel_in, db.frames, B, M, T, s));
    const int rows = (T + 1) * B;
    GemmParams g{};
    g.A = db.frames; g.amap = RowMap{rows, 0, (long)M};
    g.W = m->dev_blob + m->blob.pre_w0; g.ldw = M;
    g.C = db.pre1; g.cmap = RowMap{rows, 0, (long)P};
    g.keep = keep_masks; g.keep_ld = P;
    g.M = rows; g.N = P; g.K = M; g.act = ACT_RELU;
    HIP_TRY(launch_gemm(g, s));
    g.A = db.pre1; g.amap = RowMap{rows, 0, (long)P};
    g.W = m->dev_blob + m->blob.pre_w1; g.ldw = P;
    g.C = db.prenet; g.cmap = RowMap{B, (long)B * P, 8}; g.c_nblk = (long)B * 8;  // step t: blocked [P/8][B][8]
    g.keep = keep_masks + (size_t)rows * P;
    g.K = P;
    HIP_TRY(launch_gemm(g, s));
    if (persistent_path(m, B, L)) {   // pre_gate[t][b][:] = W_ih[:, :P] prenet(t)[b]  for all T steps: 4A x P weights read once
        GemmParams h{};
        h.A = db.prenet; h.amap = RowMap{B, (long)B * P, 8}; h.a_kblk = (long)B * 8;
        h.W = m->dev_blob + m->blob.att_wpre; h.ldw = P;
        h.C = db.pre_gate; h.cmap = RowMap{T * B, 0, (long)4 * d.att_rnn_dim};
        h.M = T * B; h.N = 4 * d.att_rnn_dim; h.K = P; h.act = ACT_NONE;
        HIP_TRY(launch_gemm(h, s));
    }
    return GVX_OK;
}

// Training mode (models/tts/tacotron2.py:341, :358): the outputs of both LSTM cells go through dropout before anything uses
// them (next step's recurrence, the attention query, the other cell, the projection).  Explicit keep masks, as for the Prenet.
struct LstmDropout {
    const uint8_t* att_keep; const uint8_t* dec_keep; float att_scale, dec_scale;   // [T][B][A], [T][B][D]
    // tape for back-propagation through time (all may be nullptr): the attention LSTM's (dropped) hidden state of every step as
    // blocked vectors [T+1][A/8][B][8] (slot t + 1 = after step t, slot 0 = zeros) and both cells' states [T+1][B][H] row-major
    float* h_a_all; float* c_a_all; float* c_d_all;
    float* pre_a_all; float* pre_d_all;   // gate pre-activations of every step [T][B][H][4] (gates of a unit together)
};

int decoder_tf_impl(gvx_model* m, const float* memory, const int32_t* lengths, int B, int L, const float* mel_in, int T,
                    const uint8_t* keep_masks, float* mel_out, float* gate_out, float* align_out, void* ws, const WsPlan& wp,
                    hipStream_t s, bool prenet_done = false, const LstmDropout* train = nullptr) {
    const gvx_dims& d = m->d;
    const int E = d.embed_dim, M = d.n_mels, P = d.prenet_dim, D = d.dec_rnn_dim;
    const DecoderBuffers db = decoder_buffers(ws, wp);
    HIP_TRY(zero_async(ws_ptr<unsigned>(ws, wp.sync), HANDOFF_WORDS * sizeof(unsigned), s));   // hand-off status of THIS call
    const bool timed = m->timing && m->ev_valid;
    int rc = GVX_OK;
    if (!prenet_done) {
        rc = decoder_prenet_part(m, B, L, mel_in, T, keep_masks, ws, wp, s);
        if (rc != GVX_OK) return rc;
    }
    if (!prenet_done) {   // (the fused forward has run this behind its encoder, on the side stream, beside the Prenet products)
        rc = decoder_init_states(m, memory, B, L, db, s);
        if (rc != GVX_OK) return rc;
    }
    if (timed) HIP_TRY(hipEventRecord(m->ev[2], s));
    // ---- T decoder steps.  Launch 1 of step t: attention-LSTM(t) together with decoder-LSTM(t-1), which is off
    // the critical chain (only the next step's projection needs it).  Launch 2: the attention step (energies, softmax, context;
    // GVX_ATTN_SPLIT=1: the round-1 energy + context pair).
    // The loop only touches workspace operands (alignments go to a time-major workspace buffer, the lengths are
    // copied in), so its 2T+1 launches are captured once per (workspace, weight blob, shape) into a hipGraph and replayed.
    const int32_t* len_ws = nullptr;
    if (lengths) {
        HIP_TRY(hipMemcpyAsync(db.len_copy, lengths, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        len_ws = db.len_copy;
    }
    int launches = 0;
    const bool kt = m->ktiming;
    if (kt) {
        rc = m->reserve_events(4);
        if (rc != GVX_OK) return rc;
        m->n_lstm_ev = m->n_attn_ev = 0;
    }
    // Persistent attention (attn_persist.hip): the loop is then T + 1 LSTM launches on `st` and ONE attention kernel on a
    // forked stream; the LSTM tiles stream the k-groups of the context last and wait for it in the launch.
    // (training mode takes the same loop since round 3: the tape - dropped hidden states, cell states, gate pre-activations -
    // is written by the cell epilogues of both launch layouts; GVX_TRAIN_RESIDENT=0 keeps the launch per attention step)
    static const bool train_resident = [] { const char* e = std::getenv("GVX_TRAIN_RESIDENT"); return !(e && e[0] == '0'); }();
    const bool pa = (!train || train_resident) && persistent_path(m, B, L);
    const int pa_layout = attention_persistent_layout(B, L);   // 1: L <= 128 (32 CUs, 224 workgroups); 2: L <= 256 (64 CUs, 192
                                                               // workgroups); 3: 33 .. 64 rows (64 CUs, 384 workgroups, two per CU)
    unsigned* sync = ws_ptr<unsigned>(ws, wp.sync);
    // ... and the LSTM launches as ONE resident kernel too (dec_resident.hip): inference mode, one batch tile, L <= 128
    // (training mode: the same kernel with the tape in its cell epilogues, when the caller asks for the whole tape;
    // GVX_TRAIN_RESIDENT_LOOP=0 keeps the launch per step)
    const bool train_resident_loop = m->train_resident_loop;
    const bool tape_whole = train && train->h_a_all && train->c_a_all && train->c_d_all && train->pre_a_all && train->pre_d_all;
    const bool resident = pa && (pa_layout == 1 || pa_layout == 2) && m->tf_resident && (!train || (train_resident_loop && tape_whole)) &&
                          decoder_resident_supported(B, L);
    // the resident tile kernel's deal: 224 workgroups beside <= 32 attention workgroups - rows of 129-256 tokens take two each, so
    // up to 16 such rows keep the 224-workgroup deal (its 48-row workgroups are lighter than the pairs of the 192-workgroup one:
    // 15.4 vs 17.3 us per step at 16 x L = 190; not at B <= 2, where the products run on the vector ALUs and the 64 slabs of the
    // 192-workgroup deal win: 13.9 vs 14.2)
    const int tile_layout = resident && pa_layout == 2 && B > 2 && B <= 16 && m->tf_long_rows_224 ? 1 : pa_layout;
    if (pa && !prenet_done) {   // (the fused forward has taken a side stream for this call already: the encoder ran on it)
        rc = ensure_side_stream(m);
        if (rc != GVX_OK) return rc;
    }
    auto defer = [&](SkinnyJob& J, int t_ctx, bool first) {   // t_ctx: the step whose context the job's x[1] is
        if (!pa) return;      // (the deferred k order costs the launch ~0.9 us: only where the context arrives in-launch)
        J.defer_seg = 1;
        if (J.q_slab) {
            // attention LSTM: its Prenet columns were applied to all steps by one GEMM before the loop (pre_gate); the job
            // streams the k-groups of [context ; h_a] only and takes the rest as an addend (step t_ctx + 1)
            const int A = d.att_rnn_dim;
            J.x[0] = XSeg{J.x[1].p, 0};
            J.kg0 = P / 8; J.nkg_w = (P + E + A) / 8; J.nkg = (E + A) / 8;
            J.addend = db.pre_gate + (size_t)(t_ctx + 1) * B * 4 * A; J.add_bs = 4 * A;
        }
        J.tmo = sync + HANDOFF_TIMEOUT;
        J.spin_limit = m->spin_limit;
        if (t_ctx >= 0) { J.ctx_cnt = sync + HANDOFF_CNT_CTX; J.ctx_target = (unsigned)B * (unsigned)(t_ctx + 1); }
        J.start_cnt = sync + HANDOFF_CNT_Q;   // every launch of the loop (and the drain launch) announces its start
        if (first) { J.ready_cnt = sync + HANDOFF_READY; J.ready_target = (unsigned)attention_persistent_workgroups(B, L); }
    };
    // The resident kernel is launched eagerly on the handle's side stream, ordered behind everything already queued on `s`;
    // only the LSTM chain is replayed from a graph (a graph that contains both may run its branches one after the other -
    // observed: the attention node first, waiting for slabs of launches queued behind it until its spin limit)
    auto pa_begin = [&](hipStream_t st) -> int {
        HIP_TRY(hipEventRecord(m->pa_fork, st));
        HIP_TRY(hipStreamWaitEvent(m->pa_stream, m->pa_fork, 0));
        AttnPersistParams pp{};
        pp.q_slab = db.q_slab; pp.n_slabs = attention_persistent_slabs(resident ? tile_layout : pa_layout);
        pp.xchg = ws_ptr<float>(ws, wp.xchg);
        pp.v = m->dev_blob + m->blob.v; pp.pm = db.pm; pp.memory = memory; pp.lengths = len_ws;
        pp.loc_conv_t = m->dev_blob + m->blob.loc_conv; pp.loc_dense_t = m->dev_blob + m->blob.loc_dense;
        pp.w_out = db.align_tm; pp.w_out_bs = (long)L; pp.w_out_ts = (long)B * L;
        pp.ctx_base = db.hc + (size_t)B * (D + E) + (size_t)D * B; pp.ctx_ts = (long)B * (D + E);   // slot t + 1
        pp.sync = sync; pp.B = B; pp.L = L; pp.T = T; pp.kl = d.att_loc_kernel;
        pp.spin_limit = m->spin_limit; pp.q_first = 2;   // (launch 0 announces its start too)
        if (resident) {   // beside the resident decoder kernel: flags per producer instead of the two counters
            pp.q_flags = sync + RS_FLAG_Q; pp.n_q_flags = pp.n_slabs;   // (row b polls replica b % RS_REP1: attn_persist.hip)
            pp.ctx_flags = sync + RS_FLAG_CTX;
            { static const int dbg = [] { const char* e = std::getenv("GVX_RS_DEBUG"); return e ? std::atoi(e) : 0; }(); pp.debug = dbg; }
        }
        if (!m->debug_skip_resident) HIP_TRY(launch_attention_persistent(pp, m->pa_stream));
        HIP_TRY(hipEventRecord(m->pa_join, m->pa_stream));
        return GVX_OK;
    };
    auto enqueue_loop = [&](hipStream_t st) -> int {
        for (int t = 0; t < T; ++t) {
            SkinnyJob jobs[2];
            fill_att_job(m, jobs[0], db.prenet + (size_t)t * B * P, t, B, db);
            defer(jobs[0], t - 1, t == 0);
            if (t > 0) {
                fill_dec_job(m, jobs[1], t - 1, B, db);
                defer(jobs[1], t - 1, false);
            }
            if (train) {
                const size_t BA = (size_t)B * d.att_rnn_dim, BD = (size_t)B * D;
                jobs[0].h_keep = train->att_keep + (size_t)t * BA; jobs[0].h_scale = train->att_scale;
                if (train->h_a_all) {
                    jobs[0].x[2].p = train->h_a_all + (size_t)t * BA; jobs[0].h_out = train->h_a_all + (size_t)(t + 1) * BA;
                }
                if (train->c_a_all) { jobs[0].c = train->c_a_all + (size_t)t * BA; jobs[0].c_out = train->c_a_all + (size_t)(t + 1) * BA; }
                if (train->pre_a_all) jobs[0].pre_out = train->pre_a_all + (size_t)t * 4 * BA;
                if (t > 0) {
                    if (train->pre_d_all) jobs[1].pre_out = train->pre_d_all + (size_t)(t - 1) * 4 * BD;
                    jobs[1].h_keep = train->dec_keep + (size_t)(t - 1) * BD; jobs[1].h_scale = train->dec_scale;
                    if (train->h_a_all) jobs[1].x[0].p = train->h_a_all + (size_t)t * BA;   // h_a(t-1)
                    if (train->c_d_all) { jobs[1].c = train->c_d_all + (size_t)(t - 1) * BD; jobs[1].c_out = train->c_d_all + (size_t)t * BD; }
                }
            }
            if (pa) {
                HIP_TRY(launch_skinny_pa(jobs[0], t > 0 ? &jobs[1] : nullptr, st, m->pa_depth, pa_layout));
                ++launches;
                continue;
            }
            LocJob lq;
            fill_loc(m, lq, t, B, L, db.align_tm, (long)L, (long)B * L, db);
            HIP_TRY(launch_skinny(jobs, t > 0 ? 2 : 1, SK_DECODER, st, &lq));
            AttnParams ap;
            fill_attn(m, ap, memory, len_ws, t, B, L, db.align_tm, (long)L, (long)B * L, db);
            {   // the attention launch has the chip to itself: block i pulls the first k-groups of tile i of the NEXT launch into
                // its XCD's L2 (both grids are dealt round-robin over the XCDs) - loop 20.27 -> 20.06 ms at 32 x 800
                // (GVX_ATTN_PREFETCH=0: off); the rotated K walk (GVX_SK_ROT, an A/B knob) is not followed
                static const bool prefetch = [] { const char* e = std::getenv("GVX_ATTN_PREFETCH"); return !e || e[0] != '0'; }();
                if (prefetch && m->attn_one_launch && B <= 32) {
                    ap.pf_w[0] = m->dev_blob + m->blob.att_frag; ap.pf_w[1] = m->dev_blob + m->blob.dec_frag;
                    ap.pf_nkg[0] = (d.prenet_dim + E + d.att_rnn_dim) / 8; ap.pf_nkg[1] = (d.att_rnn_dim + E + D) / 8;
                    ap.pf_tiles0 = 4 * d.att_rnn_dim / 32; ap.pf_tiles = ap.pf_tiles0 + 4 * D / 32;
                } }
            HIP_TRY(launch_attn(m, ap, st));
            launches += m->attn_one_launch ? 2 : 3;
        }
        SkinnyJob job;
        fill_dec_job(m, job, T - 1, B, db);
        defer(job, T - 1, false);
        if (train) {
            const size_t BA = (size_t)B * d.att_rnn_dim, BD = (size_t)B * D;
            job.h_keep = train->dec_keep + (size_t)(T - 1) * BD; job.h_scale = train->dec_scale;
            if (train->h_a_all) job.x[0].p = train->h_a_all + (size_t)T * BA;
            if (train->c_d_all) { job.c = train->c_d_all + (size_t)(T - 1) * BD; job.c_out = train->c_d_all + (size_t)T * BD; }
            if (train->pre_d_all) job.pre_out = train->pre_d_all + (size_t)(T - 1) * 4 * BD;
        }
        HIP_TRY(launch_skinny(&job, 1, SK_DECODER, st));
        ++launches;
        return GVX_OK;
    };
    // Layout 3 (33 .. 64 rows): every launch has two batch tiles per workgroup, so the matrix pipe, not the weight stream,
    // sets its length - and a decoder-LSTM tile (320 k-groups) would take 1.7x an attention-LSTM tile (192).  The decoder cell
    // is therefore cut in two along K and finished one launch later:
    //   launch t:  att-LSTM(t)            [ctx(t-1) deferred ; h_a(t-1)]            128 tiles x 192 k-groups
    //              dec-LSTM(t-1) partial  [h_a(t-1) ; ctx(t-1) deferred] -> sums    128 tiles x 192 k-groups   (mode 2)
    //              dec-LSTM(t-2) final    [h_d(t-3)] + those sums of launch t-1     128 tiles x 128 k-groups
    // 384 workgroups on the 192 CUs the resident kernel leaves, two per CU; two drain launches end the loop.
    float* dec_part2[2] = {db.dec_part, db.dec_part + (size_t)B * 4 * D};
    auto jobs64 = [&](int t, SkinnyJob* jobs) -> int {
        const int A = d.att_rnn_dim;
        int n = 0;
        if (t < T) {
            fill_att_job(m, jobs[n], db.prenet + (size_t)t * B * P, t, B, db);
            defer(jobs[n], t - 1, t == 0);
            ++n;
        }
        if (t >= 1 && t - 1 < T) {
            SkinnyJob& J = jobs[n];
            fill_dec_job(m, J, t - 1, B, db);
            J.x[2] = XSeg{nullptr, 0};
            J.nkg = (A + E) / 8; J.kg0 = 0; J.nkg_w = (A + E + D) / 8;
            J.mode = 2; J.bias = nullptr; J.c = nullptr; J.h_out = nullptr;
            J.y = dec_part2[t & 1];
            defer(J, t - 1, false);
            ++n;
        }
        if (t >= 2 && t - 2 < T) {
            SkinnyJob& J = jobs[n];
            std::memset(&J, 0, sizeof J);
            const float* hc_t = db.hc + (size_t)(t - 2) * B * (D + E);
            float* hc_n = db.hc + (size_t)(t - 1) * B * (D + E);
            J.Wp = m->dev_blob + m->blob.dec_frag; J.bias = m->dev_blob + m->blob.dec_bias;
            J.x[0] = XSeg{hc_t, D};
            J.N = 4 * D; J.nkg = D / 8; J.kg0 = (A + E) / 8; J.nkg_w = (A + E + D) / 8; J.mode = 0; J.B = B;
            J.c = db.c_d; J.h_out = hc_n;
            J.addend = dec_part2[(t - 1) & 1]; J.add_bs = 4 * D;
            J.start_cnt = sync + HANDOFF_CNT_Q;   // (only counts when this job owns block 0: never, a partial job precedes it)
            ++n;
        }
        return n;
    };
    auto enqueue_loop64 = [&](hipStream_t st) -> int {
        for (int t = 0; t < T + 2; ++t) {
            SkinnyJob jobs[3];
            const int n = jobs64(t, jobs);
            HIP_TRY(launch_skinny_pa64(jobs, n, st));
            ++launches;
        }
        return GVX_OK;
    };
    std::unique_lock<std::mutex> turn;   // held while the loop is enqueued (released on every return path)
    if (pa) {
        turn = std::unique_lock<std::mutex>(g_turn_mutex);
        rc = turn_begin(s);
        if (rc != GVX_OK) return rc;
        rc = pa_begin(s);   // (the hand-off words were zeroed at the top of this call)
        if (rc != GVX_OK) return rc;
    }
    if (resident) {
        // one launch for the whole loop: attention LSTM (t) and decoder LSTM (t) of every step, hand-offs by flags
        DecResidentParams rp{};
        rp.att_frag = m->dev_blob + m->blob.att_frag; rp.att_bias = m->dev_blob + m->blob.att_bias; rp.wq_t = m->dev_blob + m->blob.wq_t;
        rp.dec_frag = m->dev_blob + m->blob.dec_frag; rp.dec_bias = m->dev_blob + m->blob.dec_bias;
        rp.pre_gate = db.pre_gate; rp.h_a = db.h_a; rp.hc = db.hc; rp.q_slab = db.q_slab; rp.c_a = db.c_a; rp.c_d = db.c_d;
        if (train) {
            rp.h_a = train->h_a_all;
            rp.tr_keep_a = train->att_keep; rp.tr_keep_d = train->dec_keep; rp.tr_scale_a = train->att_scale; rp.tr_scale_d = train->dec_scale;
            rp.tr_c_a = train->c_a_all; rp.tr_c_d = train->c_d_all; rp.tr_pre_a = train->pre_a_all; rp.tr_pre_d = train->pre_d_all;
        }
        rp.sync = sync;
        rp.att_frag_bytes = (unsigned)(frag_floats(4 * d.att_rnn_dim, P + E + d.att_rnn_dim) * sizeof(float));
        rp.dec_frag_bytes = (unsigned)(frag_floats(4 * D, d.att_rnn_dim + E + D) * sizeof(float));
        rp.B = B; rp.T = T; rp.spin_limit = m->spin_limit; rp.layout = tile_layout;
        { static const int dbg = [] { const char* e = std::getenv("GVX_RS_DEBUG"); return e ? std::atoi(e) : 0; }(); rp.debug = dbg; }
        if (kt) HIP_TRY(hipEventRecord(m->kev[0], s));
        HIP_TRY(launch_decoder_resident(rp, s));
        if (kt) {
            HIP_TRY(hipEventRecord(m->kev[1], s));
            HIP_TRY(hipEventRecord(m->kev[2], s));
            m->n_lstm_ev = T; m->n_attn_ev = 0;   // (the kernel's duration over its T steps)
        }
        launches = 1;
    } else if (pa && pa_layout == 3) {
        if (m->use_graph && !kt) {
            const gvx_model::LoopKey key{ws, memory, m->dev_blob, B, L, T, lengths != nullptr, 0.f, 48};
            rc = run_chunk(m, touch_graph_set(m, m->loop_graphs, key), 0, s, enqueue_loop64);
            launches = T + 2;
        } else rc = enqueue_loop64(s);
        if (rc != GVX_OK) return rc;
    } else if (m->use_graph && !kt && !train) {
        const gvx_model::LoopKey key{ws, memory, m->dev_blob, B, L, T, lengths != nullptr, 0.f, pa ? m->pa_depth + 16 * pa_layout : 0};
        rc = run_chunk(m, touch_graph_set(m, m->loop_graphs, key), 0, s, enqueue_loop);
        if (rc != GVX_OK) return rc;
        launches = pa ? T + 1 : (m->attn_one_launch ? 2 : 3) * T + 1;
    } else {
        rc = enqueue_loop(s);
        if (rc != GVX_OK) return rc;
    }
    if (pa) {
        HIP_TRY(hipStreamWaitEvent(s, m->pa_join, 0));
        ++launches;
        rc = turn_end(s);
        if (rc != GVX_OK) return rc;
        turn.unlock();
    }
    if (kt && !resident) {
        // Per-kernel duration for the roofline figure: the launch of a mid-sequence step replayed back to back between
        // two events on this stream (bracketing every launch of the real loop with events measures launch gaps, not
        // the kernel).  The replays scribble over the recurrent state, which nobody reads after this point of an
        // instrumented pass except the projection of the already finished outputs' copies below.
        const int REPS = 64, tm = T > 1 ? T / 2 : 0;
        SkinnyJob jobs[2];
        fill_att_job(m, jobs[0], db.prenet + (size_t)tm * B * P, tm, B, db);
        if (tm > 0) fill_dec_job(m, jobs[1], tm - 1, B, db);
        if (pa && pa_layout == 3) {
            HIP_TRY(hipStreamWaitEvent(s, m->pa_join, 0));
            SkinnyJob j3[3];
            const int n = jobs64(tm > 2 ? tm : 2, j3);
            for (int i = 0; i < n; ++i) j3[i].start_cnt = nullptr;
            HIP_TRY(hipEventRecord(m->kev[0], s));
            for (int i = 0; i < REPS; ++i) HIP_TRY(launch_skinny_pa64(j3, n, s));
            HIP_TRY(hipEventRecord(m->kev[1], s));
            HIP_TRY(hipEventRecord(m->kev[2], s));
            m->n_lstm_ev = REPS;
            m->n_attn_ev = 0;
        } else if (pa) {
            // the launch of the loop as it ran: deferred context columns read with sc1 loads; the context counter already
            // stands at its final value, so no replay waits (the attention runs in its own kernel: nothing to time per step)
            HIP_TRY(hipStreamWaitEvent(s, m->pa_join, 0));
            defer(jobs[0], tm - 1, false);
            if (tm > 0) defer(jobs[1], tm - 1, false);
            jobs[0].start_cnt = jobs[1].start_cnt = nullptr;
            HIP_TRY(hipEventRecord(m->kev[0], s));
            for (int i = 0; i < REPS; ++i) HIP_TRY(launch_skinny_pa(jobs[0], tm > 0 ? &jobs[1] : nullptr, s, m->pa_depth, pa_layout));
            HIP_TRY(hipEventRecord(m->kev[1], s));
            HIP_TRY(hipEventRecord(m->kev[2], s));
            m->n_lstm_ev = REPS;
            m->n_attn_ev = 0;
        } else {
            LocJob lq;
            fill_loc(m, lq, tm, B, L, db.align_tm, (long)L, (long)B * L, db);
            AttnParams ap;
            fill_attn(m, ap, memory, len_ws, tm, B, L, db.align_tm, (long)L, (long)B * L, db);
            ap.w_out = db.energies;  // do not disturb the real alignments / cumulative weights (db.loc is an INPUT of the
            ap.w_cum = db.energies;  // one-launch step: it must not be scribbled on)
            HIP_TRY(hipEventRecord(m->kev[0], s));
            for (int i = 0; i < REPS; ++i) HIP_TRY(launch_skinny(jobs, tm > 0 ? 2 : 1, SK_DECODER, s, &lq));
            HIP_TRY(hipEventRecord(m->kev[1], s));
            for (int i = 0; i < REPS; ++i) HIP_TRY(launch_attn(m, ap, s));
            HIP_TRY(hipEventRecord(m->kev[2], s));
            m->n_lstm_ev = m->n_attn_ev = REPS;
        }
    }
    // alignments: time-major workspace [T][B][L] -> caller's [B][T][L]
    HIP_TRY(launch_permute01(db.align_tm, align_out, T, B, L, s));
    m->last_decoder_launches = launches;
    if (timed) HIP_TRY(hipEventRecord(m->ev[3], s));
    // ---- mel + gate projection hoisted out of the loop: one GEMM over all T*B rows of hc[1..T]
    {
        const int PS = m->PS();
        GemmParams g{};
        g.A = db.hc + (size_t)B * (D + E); g.amap = RowMap{B, (long)B * (D + E), 8}; g.a_kblk = (long)B * 8;  // slots 1..T, blocked
        g.W = m->dev_blob + m->blob.proj_w; g.ldw = D + E;
        g.C = db.proj; g.cmap = RowMap{B, (long)PS, (long)T * PS};  // row (t,b) -> proj[b][t][:]
        g.bias = m->dev_blob + m->blob.proj_b;
        g.M = T * B; g.N = M + 1; g.K = D + E; g.act = ACT_NONE;
        HIP_TRY(launch_gemm(g, s));
        HIP_TRY(launch_split_projection(db.proj, mel_out, gate_out, B, M, T, s));
    }
    return GVX_OK;
}

// Postnet + residual on channels-last halo buffers ya / yb (each B * (T + 2p) * max(postnet_dim, n_mels) floats).
// mel_lengths (optional): row b is treated as a sequence of mel_lengths[b] frames - the input and every layer's output
// are zero from that frame on, exactly what the convolutions of a batch-1 run see as padding at the sequence end.
int postnet_impl(gvx_model* m, const float* mel_in, const int32_t* mel_lengths, int B, int T, float* mel_post_out, float* ya,
                 float* yb, hipStream_t s) {
    const gvx_dims& d = m->d;
    const int M = d.n_mels, pp = (d.postnet_kernel - 1) / 2, n = d.postnet_n_conv;
    // every conv input needs zero halo rows in ITS channel layout: the producer of a buffer writes them (the transpose for
    // the first one, the GEMM epilogue of layer i for layer i + 1; a separate launch only when T < halo)
    HIP_TRY(launch_to_channels_last(mel_in, ya, B, M, T, pp, mel_lengths, s));
    float* cur = ya;
    float* nxt = yb;
    for (int i = 0; i < n; ++i) {
        const int cin = i == 0 ? M : d.postnet_dim, cout = i == n - 1 ? M : d.postnet_dim;
        const bool last = i == n - 1;
        const int out_halo = last ? 0 : pp;
        const bool fused_halo = out_halo > 0 && T >= out_halo;
        if (out_halo > 0 && !fused_halo) HIP_TRY(launch_zero_halo(nxt, B, T, pp, cout, s));
        GemmParams g{};
        g.A = cur; g.amap = RowMap{T, (long)(T + 2 * pp) * cin, (long)cin};
        g.W = m->dev_blob + m->blob.post_w[i]; g.ldw = (long)d.postnet_kernel * cin;
        g.C = nxt + (long)out_halo * cout; g.cmap = RowMap{T, (long)(T + 2 * out_halo) * cout, (long)cout};
        g.bias = m->dev_blob + m->blob.post_b[i];
        g.M = B * T; g.N = cout; g.K = d.postnet_kernel * cin; g.act = last ? ACT_NONE : ACT_TANH;
        g.row_len = last ? nullptr : mel_lengths;   // the last layer's padding is zeroed by the residual kernel
        g.c_halo = fused_halo ? out_halo : 0;
        HIP_TRY(launch_gemm(g, s));
        float* t = cur; cur = nxt; nxt = t;
    }
    HIP_TRY(launch_residual_to_channels_first(mel_in, cur, mel_post_out, B, M, T, mel_lengths, s));
    return GVX_OK;
}

// A hand-off time-out of the resident-attention loop must not return numbers that look like results: the call's last
// launch writes NaN over every output and raises the workspace's sticky status word when the time-out word is set
// (no host synchronisation; a no-op of one word read per workgroup otherwise).  gvx_workspace_status reports it.
int poison_if_timed_out(const gvx_model* m, int B, int L, void* ws, const WsPlan& wp, float* const* outs, const size_t* counts, int n,
                        hipStream_t s) {
    if (!persistent_path(m, B, L)) return GVX_OK;   // no in-launch hand-off on the other paths
    HIP_TRY(launch_poison_on_timeout(ws_ptr<unsigned>(ws, wp.sync) + HANDOFF_TIMEOUT, ws_ptr<int>(ws, wp.flags) + FLAG_TIMEOUT, outs, counts, n, s));
    return GVX_OK;
}

struct PostnetPlan { size_t ya, yb, total; };
PostnetPlan make_postnet_plan(const gvx_model* m, int B, int T) {
    const gvx_dims& d = m->d;
    const int pp = (d.postnet_kernel - 1) / 2, cmax = d.postnet_dim > d.n_mels ? d.postnet_dim : d.n_mels;
    const size_t buf = align_up((size_t)B * (T + 2 * pp) * cmax * sizeof(float), 256);
    PostnetPlan p;
    p.ya = align_up((128 + HANDOFF_WORDS) * sizeof(float), 256);   // behind the status and hand-off words of the full plan
    p.yb = p.ya + buf;
    p.total = p.yb + buf;
    return p;
}

}  // namespace

// =====================================================================================================
extern "C" {

int gvx_teacher_forced_rows_per_call(const gvx_model* m, int L) {
    if (!m) return 0;
    return persistent_path(m, 64, L) ? 64 : 32;
}

int gvx_teacher_forced_resident(const gvx_model* m, int B, int L) {
    if (!m || B < 1 || L < 1) return 0;
    return persistent_path(m, B, L) ? 1 : 0;
}

int gvx_teacher_forced_loop_kind(const gvx_model* m, int B, int L) {
    if (!m || B < 1 || L < 1) return 0;
    if (!persistent_path(m, B, L)) return 0;
    return m->tf_resident && decoder_resident_supported(B, L) ? 2 : 1;
}

int gvx_autoregressive_loop_kind(const gvx_model* m, int B, int L) {
    if (!m || B < 1 || L < 1) return 0;
    const gvx_dims& d = m->d;
    const bool pa_ok = m->attn_one_launch && attention_persistent_layout(B, L) == 1 &&
                       attention_persistent_supported(B, L, d.att_dim, d.att_loc_filters, d.att_loc_kernel, d.embed_dim, d.att_rnn_dim, d.dec_rnn_dim);
    const bool pa_any = m->attn_one_launch &&
                        attention_persistent_supported(B, L, d.att_dim, d.att_loc_filters, d.att_loc_kernel, d.embed_dim, d.att_rnn_dim, d.dec_rnn_dim);
    if (pa_any && ar_resident_loop_ok(m, B, L)) return 2;
    return pa_ok && m->ar_resident ? 1 : 0;
}

int gvx_model_set_persistent_attention(gvx_model* m, int enable) {
    if (!m) return fail(GVX_ERR_INVALID_ARG, "null argument");
    m->attn_persistent = enable != 0;
    return GVX_OK;
}

int gvx_model_set_resident_kernels(gvx_model* m, int enable) {
    if (!m) return fail(GVX_ERR_INVALID_ARG, "null argument");
    m->attn_persistent = m->enc_persistent = m->tf_resident = enable != 0;
    if (!enable) m->ar_resident = false;
    return GVX_OK;
}

int gvx_workspace_status(const gvx_model* m, void* ws, size_t ws_bytes, void* stream, int32_t* host_out) {
    if (!m || !ws || !host_out) return fail(GVX_ERR_INVALID_ARG, "null argument");
    const WsPlan wp = make_ws_plan(m, 1, 1, 1, WS_AUTOREGRESSIVE);   // the status words sit in front of every shape-dependent region
    if (ws_bytes < wp.flags + 4 * sizeof(int32_t)) return fail(GVX_ERR_WORKSPACE, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    int32_t h[4] = {0, 0, 0, 0}, tmo = 0;
    int32_t* flags = ws_ptr<int32_t>(ws, wp.flags);
    HIP_TRY(hipMemcpyAsync(h, flags, sizeof h, hipMemcpyDeviceToHost, s));
    if (ws_bytes >= wp.sync + HANDOFF_WORDS * sizeof(unsigned))   // (a Postnet-only workspace ends before the hand-off words)
        HIP_TRY(hipMemcpyAsync(&tmo, reinterpret_cast<const char*>(ws) + wp.sync + HANDOFF_TIMEOUT * sizeof(unsigned), sizeof tmo,
                               hipMemcpyDeviceToHost, s));
    // the sticky words accumulate over every call since the last look: reading them clears them
    HIP_TRY(zero_async(flags + FLAG_TOKEN, sizeof(int32_t), s));
    HIP_TRY(zero_async(flags + FLAG_TIMEOUT, sizeof(int32_t), s));
    if (ws_bytes >= wp.sync + HANDOFF_WORDS * sizeof(unsigned))
        HIP_TRY(zero_async(ws_ptr<unsigned>(ws, wp.sync) + HANDOFF_TIMEOUT, sizeof(unsigned), s));
    HIP_TRY(hipStreamSynchronize(s));
    host_out[0] = h[FLAG_TOKEN];
    host_out[1] = h[FLAG_TIMEOUT] ? h[FLAG_TIMEOUT] : tmo;
    return GVX_OK;
}

int gvx_encoder_forward(gvx_model* m, const int64_t* tokens, const int32_t* lengths, int B, int L, float* memory_out,
                        void* ws, size_t ws_bytes, void* stream) {
    int rc = check_common(m, B, L, 1, ws, ws_bytes, WS_AUTOREGRESSIVE);   // (the encoder's buffers precede every mode-dependent one)
    if (rc != GVX_OK) return rc;
    if (!tokens || !memory_out) return fail(GVX_ERR_INVALID_ARG, "null argument");
    return encoder_impl(m, tokens, lengths, B, L, memory_out, ws, make_ws_plan(m, B, L, 1, WS_AUTOREGRESSIVE), (hipStream_t)stream);
}

int gvx_decoder_teacher_forced(gvx_model* m, const float* memory, const int32_t* lengths, int B, int L, const float* mel_in, int T,
                               const uint8_t* keep_masks, float* mel_out, float* gate_out, float* align_out, void* ws,
                               size_t ws_bytes, void* stream) {
    int rc = check_common(m, B, L, T, ws, ws_bytes);
    if (rc != GVX_OK) return rc;
    if (!memory || !mel_in || !keep_masks || !mel_out || !gate_out || !align_out) return fail(GVX_ERR_INVALID_ARG, "null argument");
    const WsPlan wp = make_ws_plan(m, B, L, T);
    rc = decoder_tf_impl(m, memory, lengths, B, L, mel_in, T, keep_masks, mel_out, gate_out, align_out, ws, wp, (hipStream_t)stream);
    if (rc != GVX_OK) return rc;
    float* outs[3] = {mel_out, gate_out, align_out};
    const size_t counts[3] = {(size_t)B * m->d.n_mels * T, (size_t)B * T, (size_t)B * T * L};
    return poison_if_timed_out(m, B, L, ws, wp, outs, counts, 3, (hipStream_t)stream);
}

int gvx_encoder_lstm_forward(gvx_model* m, const float* conv_out, const int32_t* lengths, int B, int L, float* memory_out,
                             float* cell_states_out, float* input_preact_out, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_common(m, B, L, 1, ws, ws_bytes, WS_AUTOREGRESSIVE);
    if (rc != GVX_OK) return rc;
    if (!conv_out || !memory_out) return fail(GVX_ERR_INVALID_ARG, "null argument");
    if (cell_states_out) HIP_TRY(zero_async(cell_states_out, (size_t)B * L * m->d.embed_dim * sizeof(float), (hipStream_t)stream));
    return encoder_impl(m, nullptr, lengths, B, L, memory_out, ws, make_ws_plan(m, B, L, 1, WS_AUTOREGRESSIVE), (hipStream_t)stream, conv_out,
                        cell_states_out, input_preact_out);
}

int gvx_decoder_teacher_forced_train(gvx_model* m, const float* memory, const int32_t* lengths, int B, int L, const float* mel_in, int T,
                                     const uint8_t* keep_masks, const uint8_t* att_keep, const uint8_t* dec_keep, float p_att, float p_dec,
                                     float* mel_out, float* gate_out, float* align_out, float* att_hidden_all, float* att_cell_all,
                                     float* dec_cell_all, float* dec_hidden_context_all, float* att_preact_all, float* dec_preact_all,
                                     void* ws, size_t ws_bytes, void* stream) {
    int rc = check_common(m, B, L, T, ws, ws_bytes);
    if (rc != GVX_OK) return rc;
    if (!memory || !mel_in || !keep_masks || !att_keep || !dec_keep || !mel_out || !gate_out || !align_out) return fail(GVX_ERR_INVALID_ARG, "null argument");
    if (!(p_att >= 0.f && p_att < 1.f && p_dec >= 0.f && p_dec < 1.f)) return fail(GVX_ERR_INVALID_ARG, "dropout probabilities must be in [0, 1)");
    hipStream_t s = (hipStream_t)stream;
    const int A = m->d.att_rnn_dim, D = m->d.dec_rnn_dim, E = m->d.embed_dim;
    const LstmDropout tr{att_keep, dec_keep, 1.f / (1.f - p_att), 1.f / (1.f - p_dec), att_hidden_all, att_cell_all, dec_cell_all, att_preact_all, dec_preact_all};
    if (att_hidden_all) HIP_TRY(zero_async(att_hidden_all, (size_t)B * A * sizeof(float), s));   // slot 0: the initial (zero) states
    if (att_cell_all) HIP_TRY(zero_async(att_cell_all, (size_t)B * A * sizeof(float), s));
    if (dec_cell_all) HIP_TRY(zero_async(dec_cell_all, (size_t)B * D * sizeof(float), s));
    const WsPlan wp = make_ws_plan(m, B, L, T);
    rc = decoder_tf_impl(m, memory, lengths, B, L, mel_in, T, keep_masks, mel_out, gate_out, align_out, ws, wp, s, false, &tr);
    if (rc != GVX_OK) return rc;
    {   // a hand-off of the resident-attention loop that timed out must not look like a result (NaN outputs + sticky status)
        float* outs[3] = {mel_out, gate_out, align_out};
        const size_t counts[3] = {(size_t)B * m->d.n_mels * T, (size_t)B * T, (size_t)B * T * L};
        rc = poison_if_timed_out(m, B, L, ws, wp, outs, counts, 3, s);
        if (rc != GVX_OK) return rc;
    }
    if (dec_hidden_context_all)   // [T+1] slots of blocked [h_d ; ctx] vectors: slot t + 1 = after step t
        HIP_TRY(hipMemcpyAsync(dec_hidden_context_all, ws_ptr<float>(ws, wp.hc), (size_t)(T + 1) * B * (D + E) * sizeof(float),
                               hipMemcpyDeviceToDevice, s));
    return GVX_OK;
}

int gvx_train_export(const gvx_model* m, const void* ws_c, size_t ws_bytes, int B, int L, int T, int what, float* dst, void* stream) {
    void* ws = const_cast<void*>(ws_c);
    if (!m || !ws || !dst) return fail(GVX_ERR_INVALID_ARG, "null argument");
    const WsPlan wp = make_ws_plan(m, B, L, T);
    if (ws_bytes < wp.total) return fail(GVX_ERR_WORKSPACE, "workspace too small");
    const gvx_dims& d = m->d;
    hipStream_t s = (hipStream_t)stream;
    const size_t rows = (size_t)(T + 1) * B;
    switch (what) {
        case 0: HIP_TRY(hipMemcpyAsync(dst, ws_ptr<float>(ws, wp.frames), rows * d.n_mels * sizeof(float), hipMemcpyDeviceToDevice, s)); break;
        case 1: HIP_TRY(hipMemcpyAsync(dst, ws_ptr<float>(ws, wp.pre1), rows * d.prenet_dim * sizeof(float), hipMemcpyDeviceToDevice, s)); break;
        case 2: return gvx_train_unblock(ws_ptr<float>(ws, wp.prenet), dst, T + 1, B, d.prenet_dim, stream);
        case 3: HIP_TRY(hipMemcpyAsync(dst, ws_ptr<float>(ws, wp.pm), (size_t)B * L * d.att_dim * sizeof(float), hipMemcpyDeviceToDevice, s)); break;
        default: return fail(GVX_ERR_INVALID_ARG, "gvx_train_export: unknown buffer %d", what);
    }
    return GVX_OK;
}

size_t gvx_postnet_workspace_bytes(const gvx_model* m, int B, int T) {
    if (!m || B < 1 || T < 1) return 0;
    return make_postnet_plan(m, B, T).total;
}

int gvx_postnet_forward(gvx_model* m, const float* mel_in, const int32_t* mel_lengths, int B, int T, float* mel_post_out, void* ws,
                        size_t ws_bytes, void* stream) {
    if (!m) return fail(GVX_ERR_INVALID_ARG, "null model");
    if (!m->dev_blob) return fail(GVX_ERR_STATE, "weights not bound (call gvx_model_bind_blob)");
    if (B < 1 || T < 1) return fail(GVX_ERR_INVALID_ARG, "B and T must be >= 1");
    if ((long)B * T > (1L << 30)) return fail(GVX_ERR_UNSUPPORTED, "B * T = %ld frames exceed the GEMM row index range", (long)B * T);
    if (!ws) return fail(GVX_ERR_WORKSPACE, "null workspace");
    if (reinterpret_cast<uintptr_t>(ws) & 255) return fail(GVX_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    const PostnetPlan pp = make_postnet_plan(m, B, T);
    if (ws_bytes < pp.total) return fail(GVX_ERR_WORKSPACE, "workspace too small: %zu < %zu bytes", ws_bytes, pp.total);
    if (!mel_in || !mel_post_out) return fail(GVX_ERR_INVALID_ARG, "null argument");
    return postnet_impl(m, mel_in, mel_lengths, B, T, mel_post_out, ws_ptr<float>(ws, pp.ya), ws_ptr<float>(ws, pp.yb), (hipStream_t)stream);
}

int gvx_mask_padding(float* mel, float* mel_post, float* gate, const int32_t* mel_lengths, int B, int n_mels, int T, void* stream) {
    if (!mel_lengths || B < 1 || T < 1 || n_mels < 1) return fail(GVX_ERR_INVALID_ARG, "bad argument");
    HIP_TRY(launch_mask_padding(mel, mel_post, gate, mel_lengths, B, n_mels, T, (hipStream_t)stream));
    return GVX_OK;
}

int gvx_tacotron2_forward(gvx_model* m, const int64_t* tokens, const int32_t* token_lengths, int B, int L, const float* mel_in,
                          const int32_t* mel_lengths, int T, const uint8_t* keep_masks, float* mel_out, float* mel_post_out,
                          float* gate_out, float* align_out, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_common(m, B, L, T, ws, ws_bytes);
    if (rc != GVX_OK) return rc;
    if (!tokens || !mel_in || !keep_masks || !mel_out || !gate_out || !align_out)   // (mel_post_out may be null: no Postnet, no padding mask)
        return fail(GVX_ERR_INVALID_ARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    const WsPlan wp = make_ws_plan(m, B, L, T);
    const bool timed = m->timing && m->ev_valid;
    float* memory = ws_ptr<float>(ws, wp.memory);
    if (timed) HIP_TRY(hipEventRecord(m->ev[0], s));
    // The Prenet part of the decoder needs the mel input only.  On the persistent-attention path (which owns a high-priority
    // side stream) the ENCODER runs on that stream - its BiLSTM recurrence is 128 small latency-bound launches that leave the
    // chip idle, and at the higher priority they are dispatched ahead of the GEMM workgroups - while the Prenet GEMMs fill
    // the chip from the caller's stream (the other way round the encoder took 2.0 instead of 1.4 ms).
    const bool overlap = persistent_path(m, B, L);
    if (overlap) {
        rc = ensure_side_stream(m);
        if (rc != GVX_OK) return rc;
        HIP_TRY(hipEventRecord(m->pa_fork, s));
        HIP_TRY(hipStreamWaitEvent(m->pa_stream, m->pa_fork, 0));
        // The Prenet products start behind the FIRST encoder convolution (enc_fork_after): side by side from the start the two sets
        // of GEMMs slow each other and leave the second half of the recurrence - a quarter of the chip, latency bound - alone on
        // an idle GPU; behind all three convolutions the `pre_gate` GEMM outlasts the recurrence (encoder stage 1.335 / 1.31 / 1.28 /
        // 1.335 ms for the fork behind 3 / 2 / 1 / 0 convolutions with four-wave GEMM tiles; 1.14 / 1.13 / 1.15 / 1.22 with eight-wave ones:
        // tools/r4_enc2.sh)
        rc = encoder_impl(m, tokens, token_lengths, B, L, memory, ws, wp, m->pa_stream, nullptr, nullptr, nullptr, m->enc_mid);
        if (rc != GVX_OK) return rc;
        // the decoder's zero states and the memory projection (needs the encoder output) right behind the encoder on its stream:
        // they end under the tail of the Prenet products instead of between the join and the first step (~60 us)
        rc = decoder_init_states(m, memory, B, L, decoder_buffers(ws, wp), m->pa_stream);
        if (rc != GVX_OK) return rc;
        HIP_TRY(hipEventRecord(m->pa_join, m->pa_stream));
        HIP_TRY(hipStreamWaitEvent(s, m->enc_mid, 0));
        rc = decoder_prenet_part(m, B, L, mel_in, T, keep_masks, ws, wp, s);
        if (rc != GVX_OK) return rc;
        HIP_TRY(hipStreamWaitEvent(s, m->pa_join, 0));
    } else {
        rc = encoder_impl(m, tokens, token_lengths, B, L, memory, ws, wp, s);
        if (rc != GVX_OK) return rc;
    }
    if (timed) HIP_TRY(hipEventRecord(m->ev[1], s));
    rc = decoder_tf_impl(m, memory, token_lengths, B, L, mel_in, T, keep_masks, mel_out, gate_out, align_out, ws, wp, s, overlap);
    if (rc != GVX_OK) return rc;
    if (timed) HIP_TRY(hipEventRecord(m->ev[4], s));
    // (mel_post_out == nullptr: the caller runs the Postnet and the padding mask itself - over all chunks of a larger batch in
    // one call, whose GEMMs fill the chip better than a chunk's; a timed-out call's NaN in mel_out reaches them through the Postnet)
    if (mel_post_out) {
        rc = postnet_impl(m, mel_out, nullptr, B, T, mel_post_out, ws_ptr<float>(ws, wp.ya), ws_ptr<float>(ws, wp.yb), s);
        if (rc != GVX_OK) return rc;
        if (mel_lengths) HIP_TRY(launch_mask_padding(mel_out, mel_post_out, gate_out, mel_lengths, B, m->d.n_mels, T, s));
    }
    float* outs[4] = {mel_out, gate_out, align_out, mel_post_out};
    const size_t nm = (size_t)B * m->d.n_mels * T;
    const size_t counts[4] = {nm, (size_t)B * T, (size_t)B * T * L, nm};
    rc = poison_if_timed_out(m, B, L, ws, wp, outs, counts, mel_post_out ? 4 : 3, s);
    if (rc != GVX_OK) return rc;
    if (timed) HIP_TRY(hipEventRecord(m->ev[5], s));
    return GVX_OK;
}

int gvx_tacotron2_loss(const float* mel_out, const float* mel_post_out, const float* gate_out, const float* mel_target,
                       const float* gate_target, int B, int n_mels, int T, float* loss_out, void* scratch, size_t scratch_bytes,
                       void* stream) {
    if (!mel_out || !mel_post_out || !gate_out || !mel_target || !gate_target || !loss_out || !scratch)
        return fail(GVX_ERR_INVALID_ARG, "null argument");
    if (B < 1 || n_mels < 1 || T < 1) return fail(GVX_ERR_INVALID_ARG, "B, n_mels and T must be >= 1");
    if (scratch_bytes < loss_scratch_bytes() || (reinterpret_cast<uintptr_t>(scratch) & 7))
        return fail(GVX_ERR_WORKSPACE, "scratch too small (%zu bytes needed) or not 8-byte aligned", loss_scratch_bytes());
    HIP_TRY(launch_tacotron2_loss(mel_out, mel_post_out, gate_out, mel_target, gate_target, (long)B * n_mels * T, (long)B * T,
                                  reinterpret_cast<double*>(scratch), loss_out, (hipStream_t)stream));
    return GVX_OK;
}

int gvx_prenet_masks_generate(uint8_t* masks_out, size_t n, uint64_t seed, void* stream) {
    if (!masks_out) return fail(GVX_ERR_INVALID_ARG, "null argument");
    HIP_TRY(launch_mask_gen(masks_out, n, seed, (hipStream_t)stream));
    return GVX_OK;
}

int gvx_stage_timing_enable(gvx_model* m, int enable) {
    if (!m) return fail(GVX_ERR_INVALID_ARG, "null model");
    if (enable && !m->ev_valid) {
        for (auto& e : m->ev) HIP_TRY(hipEventCreate(&e));
        m->ev_valid = true;
    }
    m->timing = enable != 0;
    return GVX_OK;
}

int gvx_kernel_timing_enable(gvx_model* m, int enable) {
    if (!m) return fail(GVX_ERR_INVALID_ARG, "null model");
    m->ktiming = enable != 0;
    return GVX_OK;
}

int gvx_kernel_times_ms(gvx_model* m, float* lstm_avg_ms, float* attn_avg_ms, int* n_steps) {
    if (!m || !lstm_avg_ms || !attn_avg_ms) return fail(GVX_ERR_INVALID_ARG, "null argument");
    if (m->n_lstm_ev < 1) return fail(GVX_ERR_STATE, "no timed decoder loop has run");
    HIP_TRY(hipEventSynchronize(m->kev[2]));
    float a = 0, b = 0;
    HIP_TRY(hipEventElapsedTime(&a, m->kev[0], m->kev[1]));
    HIP_TRY(hipEventElapsedTime(&b, m->kev[1], m->kev[2]));
    *lstm_avg_ms = a / m->n_lstm_ev;
    *attn_avg_ms = m->n_attn_ev > 0 ? b / m->n_attn_ev : 0.f;   // 0: the attention ran as one kernel beside the loop
    if (n_steps) *n_steps = m->n_lstm_ev;
    return GVX_OK;
}

int gvx_stage_times_ms(gvx_model* m, float* t5, int* launches) {
    if (!m || !t5) return fail(GVX_ERR_INVALID_ARG, "null argument");
    if (!m->ev_valid) return fail(GVX_ERR_STATE, "stage timing was not enabled");
    HIP_TRY(hipEventSynchronize(m->ev[5]));
    // ev: 0 start, 1 encoder done, 2 prenet+init done, 3 decoder loop done, 4 projection done, 5 postnet done
    for (int i = 0; i < 5; ++i) HIP_TRY(hipEventElapsedTime(&t5[i], m->ev[i], m->ev[i + 1]));
    if (launches) *launches = m->last_decoder_launches;
    return GVX_OK;
}

int gvx_decoder_autoregressive(gvx_model* m, const float* memory, const int32_t* lengths, int B, int L, int max_steps,
                               float gate_threshold, const uint8_t* keep_masks, float* mel_out, float* gate_out, float* align_out,
                               int32_t* n_frames_out, int* steps_run_out, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_common(m, B, L, max_steps, ws, ws_bytes, WS_AUTOREGRESSIVE);
    if (rc != GVX_OK) return rc;
    if (!memory || !keep_masks || !mel_out || !gate_out || !align_out || !n_frames_out)
        return fail(GVX_ERR_INVALID_ARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    const gvx_dims& d = m->d;
    const int E = d.embed_dim, M = d.n_mels, P = d.prenet_dim, A = d.att_rnn_dim, D = d.dec_rnn_dim, T = max_steps;
    const WsPlan wp = make_ws_plan(m, B, L, T, WS_AUTOREGRESSIVE);
    const DecoderBuffers db = decoder_buffers(ws, wp);
    if (std::getenv("GVX_DEBUG_PLAN")) {   // (diagnostics, tools/ar_ws_diff.py: byte offsets of the workspace buffers)
        static bool printed = false;
        if (!printed) {
            printed = true;
#define GVX_PL(f) std::fprintf(stderr, "wsplan %s %zu\n", #f, wp.f);
            GVX_PL(xa) GVX_PL(xb) GVX_PL(xg) GVX_PL(enc_h) GVX_PL(enc_c) GVX_PL(flags) GVX_PL(sync) GVX_PL(memory) GVX_PL(pm) GVX_PL(frames) GVX_PL(pre1) GVX_PL(prenet) GVX_PL(h_a) GVX_PL(c_a) GVX_PL(c_d)
            GVX_PL(hc) GVX_PL(w_cum) GVX_PL(q_slab) GVX_PL(proj) GVX_PL(energies) GVX_PL(align_tm) GVX_PL(len_copy) GVX_PL(loc) GVX_PL(ar_masks)
            GVX_PL(p_slab) GVX_PL(p_ctx) GVX_PL(att_part) GVX_PL(dec_part) GVX_PL(pre_gate) GVX_PL(xchg) GVX_PL(ya) GVX_PL(yb) GVX_PL(total)
#undef GVX_PL
        }
    }
    HIP_TRY(zero_async(ws_ptr<unsigned>(ws, wp.sync), HANDOFF_WORDS * sizeof(unsigned), s));   // hand-off status of THIS call
    const int PSB = m->PSB();
    int32_t* flags = ws_ptr<int32_t>(ws, wp.flags);
    int32_t* n_done = flags + FLAG_AR_DONE;
    int32_t* n_frames_ws = flags + FLAG_AR_FRAMES;
    // Everything a step touches is moved next to the workspace so that the step launches only bake workspace addresses:
    // encoder output, token lengths and keep masks are copied in; alignments / per-step projections stay in workspace
    // buffers and are scattered to the caller's tensors once, after the loop.
    float* memory_ws = ws_ptr<float>(ws, wp.memory);
    if (memory != memory_ws)
        HIP_TRY(hipMemcpyAsync(memory_ws, memory, (size_t)B * L * E * sizeof(float), hipMemcpyDeviceToDevice, s));
    uint8_t* masks_ws = ws_ptr<uint8_t>(ws, wp.ar_masks);
    HIP_TRY(hipMemcpyAsync(masks_ws, keep_masks, (size_t)2 * T * B * P, hipMemcpyDeviceToDevice, s));
    const int32_t* len_ws = nullptr;
    if (lengths) {
        HIP_TRY(hipMemcpyAsync(db.len_copy, lengths, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        len_ws = db.len_copy;
    }
    rc = decoder_init_states(m, memory_ws, B, L, db, s);
    if (rc != GVX_OK) return rc;
    // Resident attention (attn_persist.hip) when the shape allows it: ONE attention kernel lives beside the step launches for
    // the whole decode, the context of a step arrives inside launch C (deferred segment) and the attention launch leaves the
    // step's chain.  The projection's context columns then ride on the decoder-LSTM tiles' projection slabs (`fold`), so that
    // launch C is exactly 256 tiles.
    const bool pa_any = m->attn_one_launch &&
                        attention_persistent_supported(B, L, d.att_dim, d.att_loc_filters, d.att_loc_kernel, d.embed_dim, d.att_rnn_dim, d.dec_rnn_dim);
    const bool pa_ok = pa_any && attention_persistent_layout(B, L) == 1;
    const bool fold = B <= 32 && E / 4 == D / 8;
    // ... and when the layer sizes are the default ones, the LSTM cells, the projection and the Prenet live in a second resident
    // kernel as well (dec_resident.hip, decoder_ar_resident_kernel): the whole decode is two launches.  Not on handles that share
    // the chip with other calls (gvx_model_set_persistent_attention(model, 0): the two kernels need all 256 CUs)
    const bool ar_res = pa_any && ar_resident_loop_ok(m, B, L);
    const bool pa = (pa_ok && m->ar_resident) || ar_res;
    // h_a(t) exists when launch A ends, the context only after the attention step: the h_a columns of both cells (two thirds of
    // what launch C used to stream) are summed by tiles that share the attention step's launch - the step's latency chain
    // hides under 33 MB of weight stream - and launch C is left with the context columns
    // (not on handles that share the chip with other calls - gvx_model_set_persistent_attention(model, 0), the lanes of a
    // batch above 32 rows: two such launches of 512 workgroups each queue behind one another, measured 66 vs 61 us per step)
    const bool split_h = !pa && m->ar_split_h && m->attn_persistent && m->attn_one_launch && B <= 32 && d.att_dim > 32 && d.att_dim <= 128;
    float* att_part2 = db.att_part + (size_t)B * 4 * A;
    float* dec_part2 = db.dec_part + (size_t)B * 4 * D;
    unsigned* sync = ws_ptr<unsigned>(ws, wp.sync);
    if (pa) {
        rc = ensure_side_stream(m);
        if (rc != GVX_OK) return rc;
    }
    HIP_TRY(zero_async(db.prenet, (size_t)B * P * sizeof(float), s));       // Prenet of the go-frame: no biases, relu(W 0) = 0
    HIP_TRY(zero_async(db.att_part, (size_t)B * 4 * A * sizeof(float), s)); // ctx(-1) = h_a(-1) = 0
    HIP_TRY(zero_async(n_done, sizeof(int32_t), s));                         // (the sticky status words in between stay)
    HIP_TRY(zero_async(n_frames_ws, (size_t)64 * sizeof(int32_t), s));

    // One step = 5 launches.  In autoregressive mode BOTH cells are on the critical chain (the frame feeds back), and a cell
    // alone is only 128 tiles - half the chip.  But most of a cell's input is known one launch early: the attention LSTM's
    // [ctx(t-1) ; h_a(t-1)] columns (1536 of 1792) exist before the decoder LSTM of step t-1 runs, the decoder LSTM's h_d(t-1)
    // columns (1024 of 2560) before the attention LSTM of step t.  So every LSTM launch runs 128 tiles that FINISH one cell
    // (remaining columns + partial sums of the others through `addend`) next to 128 tiles that pre-compute the other cell's
    // early columns (mode 2, partial sums to att_part / dec_part): all 256 CUs stream weights in both launches.
    //   A(t): attention LSTM final [prenet(t)]            + decoder LSTM partial [h_d(t-1)]      + location features
    //   energies(t), context(t)
    //   C(t): decoder LSTM final [h_a(t) ; ctx(t)] (+ mel/gate projection partials of its 8 hidden units)
    //         + attention LSTM partial for step t+1 [ctx(t) ; h_a(t)] + 3 tiles projecting the context
    //   D(t): projection reduction, per-row stop test, whole Prenet of step t+1 on the fresh frame
    // db.proj holds one blocked projection vector [PSB/8][B][8] per step.
    const int kgP = P / 8, kgE = E / 8, kgA = A / 8, kgD = D / 8;
    auto enqueue_steps = [&](hipStream_t st, int t0, int t1) -> int {
        for (int t = t0; t < t1; ++t) {
            float* proj_t = db.proj + (size_t)t * B * PSB;
            const float* hc_t = db.hc + (size_t)t * B * (D + E);
            float* hc_n = db.hc + (size_t)(t + 1) * B * (D + E);
            float* ha_prev = db.h_a + (size_t)(t & 1) * B * A;
            float* ha_new = db.h_a + (size_t)((t + 1) & 1) * B * A;
            SkinnyJob ja[2];
            std::memset(ja, 0, sizeof ja);
            {   // attention LSTM of step t: final tiles over the Prenet columns
                SkinnyJob& J = ja[0];
                J.Wp = m->dev_blob + m->blob.att_frag; J.bias = m->dev_blob + m->blob.att_bias;
                J.x[0] = XSeg{db.prenet, P};
                J.N = 4 * A; J.nkg = kgP; J.kg0 = 0; J.nkg_w = kgP + kgE + kgA; J.mode = 0; J.B = B;
                J.addend = db.att_part; J.add_bs = 4 * A; J.add_ts = 0;
                J.c = db.c_a; J.h_out = ha_new;
                J.Wq_t = m->dev_blob + m->blob.wq_t; J.q_slab = db.q_slab; J.att_dim = d.att_dim;
            }
            {   // decoder LSTM of step t: partial sums over the h_d(t-1) columns
                SkinnyJob& J = ja[1];
                J.Wp = m->dev_blob + m->blob.dec_frag;
                J.x[0] = XSeg{hc_t, D};
                J.N = 4 * D; J.nkg = kgD; J.kg0 = kgA + kgE; J.nkg_w = kgA + kgE + kgD; J.mode = 2; J.B = B;
                J.y = db.dec_part;
            }
            if (pa) {
                if (t == 0) {   // the first launch does not end before the resident kernel is resident: launch C waits for it
                    ja[0].ready_cnt = sync + HANDOFF_READY; ja[0].ready_target = (unsigned)B;
                    ja[0].tmo = sync + HANDOFF_TIMEOUT; ja[0].spin_limit = m->spin_limit;
                }
                HIP_TRY(launch_skinny(ja, 2, SK_AR, st));
            } else {
                LocJob lq;
                fill_loc(m, lq, t, B, L, db.align_tm, (long)L, (long)B * L, db);
                HIP_TRY(launch_skinny(ja, 2, SK_AR, st, &lq));
                AttnParams ap;
                fill_attn(m, ap, memory_ws, len_ws, t, B, L, db.align_tm, (long)L, (long)B * L, db);
                if (split_h) {
                    SkinnyJob jb[2];
                    std::memset(jb, 0, sizeof jb);
                    {   // decoder LSTM of step t: partial sums over the h_a(t) columns
                        SkinnyJob& J = jb[0];
                        J.Wp = m->dev_blob + m->blob.dec_frag;
                        J.x[0] = XSeg{ha_new, A};
                        J.N = 4 * D; J.nkg = kgA; J.kg0 = 0; J.nkg_w = kgA + kgE + kgD; J.mode = 2; J.B = B;
                        J.y = dec_part2;
                    }
                    {   // attention LSTM of step t+1: partial sums over the h_a(t) columns
                        SkinnyJob& J = jb[1];
                        J.Wp = m->dev_blob + m->blob.att_frag;
                        J.x[0] = XSeg{ha_new, A};
                        J.N = 4 * A; J.nkg = kgA; J.kg0 = kgP + kgE; J.nkg_w = kgP + kgE + kgA; J.mode = 2; J.B = B;
                        J.y = att_part2;
                    }
                    HIP_TRY(launch_skinny_attn(jb, 2, ap, st));
                } else {
                    HIP_TRY(launch_attn(m, ap, st));
                }
            }
            SkinnyJob jc[3];
            std::memset(jc, 0, sizeof jc);
            {   // decoder LSTM of step t: final tiles over [h_a(t) ; ctx(t)]; every tile also emits the mel/gate projection
                // partial products of its 8 hidden units (the attention-query slab mechanism with the projection's h_d columns)
                SkinnyJob& J = jc[0];
                J.Wp = m->dev_blob + m->blob.dec_frag; J.bias = m->dev_blob + m->blob.dec_bias;
                J.x[0] = XSeg{ha_new, A};
                J.x[1] = XSeg{hc_n + (size_t)D * B, E};
                J.N = 4 * D; J.nkg = kgA + kgE; J.kg0 = 0; J.nkg_w = kgA + kgE + kgD; J.mode = 0; J.B = B;
                J.addend = db.dec_part; J.add_bs = 4 * D; J.add_ts = 0;
                J.c = db.c_d; J.h_out = hc_n;
                J.Wq_t = m->dev_blob + m->blob.proj_hd_t; J.q_slab = db.p_slab; J.att_dim = PSB;
                if (fold) { J.xw = m->dev_blob + m->blob.proj_ctx_t; J.xsrc = hc_n + (size_t)D * B; }
            }
            {   // attention LSTM of step t+1: partial sums over [ctx(t) ; h_a(t)]  (x[0] is an empty segment so that the
                // context is x[1], the segment the deferred order streams last)
                SkinnyJob& J = jc[1];
                J.Wp = m->dev_blob + m->blob.att_frag;
                J.x[0] = XSeg{hc_n + (size_t)D * B, 0};
                J.x[1] = XSeg{hc_n + (size_t)D * B, E};
                J.x[2] = XSeg{ha_new, A};
                J.N = 4 * A; J.nkg = kgE + kgA; J.kg0 = kgP; J.nkg_w = kgP + kgE + kgA; J.mode = 2; J.B = B;
                J.y = db.att_part;
            }
            if (split_h) {   // launch C streams the context columns only; the h_a columns arrive as sums
                SkinnyJob& Jd = jc[0];
                Jd.x[0] = XSeg{hc_n + (size_t)D * B, E}; Jd.x[1] = XSeg{nullptr, 0};
                Jd.nkg = kgE; Jd.kg0 = kgA;
                Jd.addend2 = dec_part2;
                SkinnyJob& Ja = jc[1];
                Ja.x[0] = XSeg{hc_n + (size_t)D * B, E}; Ja.x[1] = XSeg{nullptr, 0}; Ja.x[2] = XSeg{nullptr, 0};
                Ja.nkg = kgE; Ja.kg0 = kgP;
                Ja.addend = att_part2; Ja.add_bs = 4 * A;
            }
            if (pa)
                for (int i = 0; i < 2; ++i) {   // the context of step t is published by the resident kernel while this launch streams
                    SkinnyJob& J = jc[i];
                    J.defer_seg = 1;
                    J.ctx_cnt = sync + HANDOFF_CNT_CTX; J.ctx_target = (unsigned)B * (unsigned)(t + 1);
                    J.tmo = sync + HANDOFF_TIMEOUT; J.spin_limit = m->spin_limit;
                    if (i == 0) J.start_cnt = sync + HANDOFF_CNT_Q;   // "launch A of this step has completed: its query slabs are in memory"
                }
            if (!fold) {   // context columns of the mel/gate projection (known before the launch)
                SkinnyJob& J = jc[2];
                J.Wp = m->dev_blob + m->blob.proj_ctx_frag; J.bias = m->dev_blob + m->blob.proj_b;
                J.x[0] = XSeg{hc_n + (size_t)D * B, E};
                J.N = M + 1; J.nkg = kgE; J.mode = 1; J.B = B; J.act = ACT_NONE;
                J.y = db.p_ctx;
            }
            (void)ha_prev;
            HIP_TRY(launch_skinny(jc, fold ? 2 : 3, SK_AR, st));
            const bool more = t + 1 < T;
            HIP_TRY(launch_ar_project(db.p_slab, D / 8, db.p_ctx, proj_t, M, gate_threshold, t, B, n_frames_ws, n_done,
                                      m->dev_blob + m->blob.pre_w0_t, m->dev_blob + m->blob.pre_w1_t, P,
                                      more ? masks_ws + (size_t)(t + 1) * B * P : nullptr,
                                      more ? masks_ws + ((size_t)T + t + 1) * B * P : nullptr, db.prenet, st));
        }
        return GVX_OK;
    };
    if (fold && !ar_res) {   // p_ctx = the projection's bias, once: the linear job on the all-zero context of slot 0
        SkinnyJob J;
        std::memset(&J, 0, sizeof J);
        J.Wp = m->dev_blob + m->blob.proj_ctx_frag; J.bias = m->dev_blob + m->blob.proj_b;
        J.x[0] = XSeg{db.hc + (size_t)D * B, E};
        J.N = M + 1; J.nkg = kgE; J.mode = 1; J.B = B; J.act = ACT_NONE;
        J.y = db.p_ctx;
        HIP_TRY(launch_skinny(&J, 1, SK_AR, s));
    }
    if (pa && !ar_res) {   // the resident kernel: launched eagerly on the handle's side stream, behind everything queued on `s` so far
        HIP_TRY(hipEventRecord(m->pa_fork, s));
        HIP_TRY(hipStreamWaitEvent(m->pa_stream, m->pa_fork, 0));
        AttnPersistParams pp{};
        pp.q_slab = db.q_slab; pp.n_slabs = A / 8;
        pp.v = m->dev_blob + m->blob.v; pp.pm = db.pm; pp.memory = memory_ws; pp.lengths = len_ws;
        pp.loc_conv_t = m->dev_blob + m->blob.loc_conv; pp.loc_dense_t = m->dev_blob + m->blob.loc_dense;
        pp.w_out = db.align_tm; pp.w_out_bs = (long)L; pp.w_out_ts = (long)B * L;
        pp.ctx_base = db.hc + (size_t)B * (D + E) + (size_t)D * B; pp.ctx_ts = (long)B * (D + E);   // slot t + 1
        pp.sync = sync; pp.B = B; pp.L = L; pp.T = T; pp.kl = d.att_loc_kernel;
        pp.spin_limit = m->spin_limit; pp.q_first = 1;   // one signalling launch (C) per step
        if (!m->debug_skip_resident) HIP_TRY(launch_attention_persistent(pp, m->pa_stream));
        HIP_TRY(hipEventRecord(m->pa_join, m->pa_stream));
    }
    const int CHUNK = 16;  // steps per graph = steps between host checks of the all-rows-finished counter
    int t = 0;
    if (ar_res) {
        // ---- the whole decode as TWO resident kernels (dec_resident.hip decoder_ar_resident_kernel + attn_persist.hip, AR role): no
        // launch per step, no host check - the kernels find the end of the loop themselves (every row's stop token has fired: the
        // stop word holds the number of steps that ran) or run into max_steps
        std::unique_lock<std::mutex> turn(g_turn_mutex);   // resident loops take turns on the device
        rc = turn_begin(s);
        if (rc != GVX_OK) return rc;
        HIP_TRY(hipEventRecord(m->pa_fork, s));
        HIP_TRY(hipStreamWaitEvent(m->pa_stream, m->pa_fork, 0));
        static const int dbg = [] { const char* e = std::getenv("GVX_RS_DEBUG"); return e ? std::atoi(e) : 0; }();
        AttnPersistParams pp{};
        pp.q_slab = db.q_slab; pp.n_slabs = attention_persistent_slabs(1);
        pp.xchg = ws_ptr<float>(ws, wp.xchg);   // (rows of 129-256 tokens: the halves' exchange buffers)
        pp.v = m->dev_blob + m->blob.v; pp.pm = db.pm; pp.memory = memory_ws; pp.lengths = len_ws;
        pp.loc_conv_t = m->dev_blob + m->blob.loc_conv; pp.loc_dense_t = m->dev_blob + m->blob.loc_dense;
        pp.w_out = db.align_tm; pp.w_out_bs = (long)L; pp.w_out_ts = (long)B * L;
        pp.ctx_base = db.hc + (size_t)B * (D + E) + (size_t)D * B; pp.ctx_ts = (long)B * (D + E);   // slot t + 1
        pp.sync = sync; pp.B = B; pp.L = L; pp.T = T; pp.kl = d.att_loc_kernel;
        pp.spin_limit = m->spin_limit; pp.q_first = 1;
        pp.q_flags = sync + RS_FLAG_Q; pp.n_q_flags = pp.n_slabs; pp.ctx_flags = sync + RS_FLAG_CTX; pp.debug = dbg;
        pp.p_slab = db.p_slab; pp.PSB = PSB; pp.n_mels = M; pp.proj_b = m->dev_blob + m->blob.proj_b; pp.proj_out = db.proj;
        pp.pre_w0_t = m->dev_blob + m->blob.pre_w0_t; pp.keep0 = masks_ws; pp.y1 = db.pre1;
        pp.n_frames = n_frames_ws; pp.n_done = n_done; pp.gate_threshold = gate_threshold;
        pp.p_flags = sync + RS_FLAG_P; pp.y1_flags = sync + RS_FLAG_Y1;
        if (!m->debug_skip_resident) HIP_TRY(launch_attention_persistent(pp, m->pa_stream));
        HIP_TRY(hipEventRecord(m->pa_join, m->pa_stream));
        ArResidentParams rp{};
        rp.att_frag = m->dev_blob + m->blob.att_frag; rp.att_bias = m->dev_blob + m->blob.att_bias; rp.wq_t = m->dev_blob + m->blob.wq_t;
        rp.dec_frag = m->dev_blob + m->blob.dec_frag; rp.dec_bias = m->dev_blob + m->blob.dec_bias;
        rp.proj_hd_t = m->dev_blob + m->blob.proj_hd_t; rp.proj_ctx_t = m->dev_blob + m->blob.proj_ctx_t;
        rp.pre_w1 = m->dev_blob + m->blob.pre_w1; rp.keep1 = masks_ws + (size_t)T * B * P;
        rp.prenet = db.prenet; rp.y1 = db.pre1;
        rp.h_a = db.h_a; rp.hc = db.hc; rp.q_slab = db.q_slab; rp.p_slab = db.p_slab; rp.c_a = db.c_a; rp.c_d = db.c_d;
        rp.n_done = n_done; rp.sync = sync;
        rp.att_frag_bytes = (unsigned)(frag_floats(4 * A, P + E + A) * sizeof(float));
        rp.B = B; rp.T = T; rp.PSB = PSB; rp.spin_limit = m->spin_limit; rp.debug = dbg;
        HIP_TRY(launch_decoder_ar_resident(rp, s));
        HIP_TRY(hipStreamWaitEvent(s, m->pa_join, 0));
        rc = turn_end(s);
        if (rc != GVX_OK) return rc;
        turn.unlock();
        if (!m->ar_done_host) {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&m->ar_done_host), 2 * sizeof(int32_t), hipHostMallocDefault));
            for (auto& e : m->ar_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        HIP_TRY(hipMemcpyAsync(m->ar_done_host, sync + HANDOFF_STOP, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        t = m->ar_done_host[0] > 0 && m->ar_done_host[0] < T ? m->ar_done_host[0] : T;
    }
    gvx_model::GraphSet* gset = nullptr;
    if (m->use_graph && !ar_res) {
        gvx_model::LoopKey key{ws, memory_ws, m->dev_blob, B, L, T, lengths != nullptr};
        key.threshold = gate_threshold;
        key.variant = pa ? 1 : (split_h ? 2 : 0);
        gset = touch_graph_set(m, m->ar_graphs, key);
    }
    // One chunk of look-ahead: chunk k + 1 is enqueued BEFORE the host reads chunk k's all-rows-finished counter (pinned slot,
    // event), so the GPU never idles for the round trip of the check (~63 of them in a 1000-step decode: 30-40 us each).  When
    // chunk k turns out to have finished every row, the chunk already in flight runs 16 more steps that nobody reads: rows that
    // have fired keep their frame counts, and only the steps up to the end of chunk k are emitted below.
    if (!m->ar_done_host) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&m->ar_done_host), 2 * sizeof(int32_t), hipHostMallocDefault));
        for (auto& e : m->ar_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    auto enqueue_chunk = [&](int t0, int slot) -> int {
        const int t1 = t0 + CHUNK < T ? t0 + CHUNK : T;
        const int r = run_chunk(m, gset, (size_t)(t0 / CHUNK), s, [&](hipStream_t st) { return enqueue_steps(st, t0, t1); });
        if (r != GVX_OK) return r;
        HIP_TRY(hipMemcpyAsync(m->ar_done_host + slot, n_done, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(m->ar_ev[slot], s));
        return GVX_OK;
    };
    int t_enq = 0, slot = 0;   // steps enqueued so far; slot of the chunk the host looks at next
    if (!ar_res) {
        rc = enqueue_chunk(0, 0);
        if (rc != GVX_OK) return rc;
        t_enq = CHUNK < T ? CHUNK : T;
    }
    while (!ar_res) {
        const int t_chunk_end = t_enq;   // end of the chunk whose counter is read next
        const bool more = t_enq < T;
        if (more) {
            rc = enqueue_chunk(t_enq, slot ^ 1);
            if (rc != GVX_OK) return rc;
            t_enq = t_enq + CHUNK < T ? t_enq + CHUNK : T;
        }
        HIP_TRY(hipEventSynchronize(m->ar_ev[slot]));
        t = t_chunk_end;
        if (m->ar_done_host[slot] >= B || !more) break;
        slot ^= 1;
    }
    if (pa && !ar_res) {   // the loop may have ended early: tell the resident kernel (it leaves at its next look), then wait for it
        HIP_TRY(launch_handoff_set(sync + HANDOFF_STOP, s));
        HIP_TRY(hipStreamWaitEvent(s, m->pa_join, 0));
    }
    // rows that never fired ran into the cap ("Warning! Reached max decoder steps", models/tts/tacotron2.py:407-409)
    HIP_TRY(launch_ar_stop(db.proj, M, -1.f, t - 1, B, n_frames_ws, n_done, s));
    HIP_TRY(hipMemcpyAsync(n_frames_out, n_frames_ws, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    // rows that stopped early kept decoding until the last row finished: their frames past n_frames get the reference's
    // padding values (mel 0, gate 1e3, alignment 0 - mask_padding, models/tts/tacotron2.py:466-473)
    HIP_TRY(launch_ar_emit_all(db.proj, mel_out, gate_out, B, M, T, t, n_frames_ws, s));
    HIP_TRY(launch_permute01_partial(db.align_tm, align_out, t, T, B, L, n_frames_ws, s));
    if (pa) {   // a hand-off that timed out must not leave numbers that look like results
        float* outs[3] = {mel_out, gate_out, align_out};
        const size_t counts[3] = {(size_t)B * M * T, (size_t)B * T, (size_t)B * T * L};
        HIP_TRY(launch_poison_on_timeout(sync + HANDOFF_TIMEOUT, flags + FLAG_TIMEOUT, outs, counts, 3, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    if (steps_run_out) *steps_run_out = t;
    return GVX_OK;
}

#ifdef GVX_STAMPS
int gvx_debug_read_stamps_skinny(unsigned long long* host96) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(gvx::read_stamps_skinny(host96));
    return GVX_OK;
}
int gvx_debug_read_stamps_persist(unsigned long long* host96) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(gvx::read_stamps_persist(host96));
    return GVX_OK;
}
int gvx_debug_read_wg_spans(unsigned long long* host1024) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(gvx::read_wg_spans(host1024));
    return GVX_OK;
}
int gvx_debug_read_stamps_resident(unsigned long long* host480) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(gvx::read_stamps_resident(host480));
    return GVX_OK;
}
int gvx_debug_read_wg_stamps_resident(unsigned long long* host896, unsigned long long* rows512) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(gvx::read_wg_stamps_resident(host896));
    HIP_TRY(gvx::read_row_stamps_persist(rows512));
    return GVX_OK;
}
int gvx_debug_read_loc_stamps(unsigned long long* host32) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(gvx::read_loc_stamps_persist(host32));
    return GVX_OK;
}
int gvx_debug_read_stamps_ar(unsigned long long* host896, unsigned long long* rows256) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(gvx::read_wg_stamps_resident_ar(host896));
    HIP_TRY(gvx::read_row_stamps_persist_ar(rows256));
    return GVX_OK;
}
int gvx_debug_read_stamps(unsigned long long* host96) {
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long tmp[96];
    HIP_TRY(gvx::read_stamps_skinny(host96));       // row 0 is the LSTM kernel's
    HIP_TRY(gvx::read_stamps_attention(tmp));        // rows 1, 2 are the attention kernels'
    for (int i = 32; i < 96; ++i) host96[i] = tmp[i];
    return GVX_OK;
}
#endif

}  // extern "C"
