// Internal launcher declarations shared by the kernel translation units and the C-ABI (gvx_api.hip).
// gfx950 only. All device data is fp32 unless noted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gvx {

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

struct GemmParams {
    const float* A; RowMap amap;
    const float* W; long ldw;          // W row n at W + n*ldw (K contiguous)
    float* C; RowMap cmap;
    const float* bias;                 // [N] or nullptr
    const uint8_t* keep; long keep_ld; // Prenet keep mask [M][N] {0,1} or nullptr; kept values are doubled
    int M, N, K;                       // K % 4 == 0
    int act;
};
hipError_t launch_gemm(const GemmParams& p, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Skinny recurrent GEMM (batch rows <= 64): one workgroup = 32 packed output rows x all batch rows,
// K split over the workgroup's waves, weights streamed once from HBM in MFMA-fragment order.
// Epilogue LSTM: fused cell update (i,f,g,o), optional attention-query partial slabs.
// Epilogue LINEAR: bias + activation + keep mask.
// ---------------------------------------------------------------------------------------------
struct XSeg { const float* p; long stride; int len; };   // x[b][k] = p[b*stride + k], len % 8 == 0

struct SkinnyJob {
    const float* Wp;        // packed fragments [ntiles][nkg][64 lanes][4]
    const float* bias;      // [N] in packed row order, or nullptr
    XSeg x[3];              // K = x[0].len + x[1].len + x[2].len
    int N;                  // output rows (LSTM: 4H in packed order row = 4*j + gate)
    int nkg;                // K / 8
    int mode;               // 0 = LSTM cell, 1 = linear
    // --- LSTM epilogue
    float* c;               // [B][H] cell state, updated in place
    float* h_out; long h_out_stride;     // h'[b][j] -> h_out[b*stride + j]
    float* h_out2; long h_out2_stride;   // optional second copy
    // encoder extras (all nullptr/0 for the decoder)
    const float* addend; long add_bs, add_ts;  // pre-activation addend[b][t_b][n] (x-projection incl. bias)
    const int32_t* lengths; int step; int reverse; int seq_len;  // packed-sequence semantics
    float* seq_out; long seq_bs, seq_ts;       // seq_out[b][t_b][j]
    const float* h_prev; long h_prev_stride;   // carried over for inactive rows
    // attention query partial products: slab[tile][b][a] = sum_{j in tile} Wq[a][j] * h'[b][j]
    const float* Wq_t;      // [H/8][att_dim][8] (tile-major repack of query_layer.weight) or nullptr
    float* q_slab; int att_dim;
    // --- linear epilogue
    float* y; long y_stride;            // y[b*stride + n]
    const uint8_t* keep; long keep_stride;
    int act;
    int B;
};
enum SkinnyKind : int { SK_DECODER = 0, SK_ENCODER = 1, SK_LINEAR = 2 };  // kernel name only; same code
hipError_t launch_skinny(const SkinnyJob* jobs, int njobs, int kind, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Location-sensitive attention, one step, one workgroup per batch row.
// ---------------------------------------------------------------------------------------------
struct AttnParams {
    const float* q_slab; int n_slabs;       // [n_slabs][B][a]
    const float* w_prev; long w_prev_bs;    // previous alignment row b at w_prev + b*bs, or nullptr (step 0)
    float* w_cum;                           // [B][L], updated in place
    const float* loc_conv;                  // [F][2][kl]
    const float* loc_dense;                 // [a][F]
    const float* v;                         // [a]
    const float* pm;                        // [B][L][a]
    const float* memory;                    // [B][L][E]
    const int32_t* lengths;                 // [B] or nullptr
    float* w_out; long w_out_bs;            // new alignment row b -> w_out + b*bs   (length L)
    float* ctx_out; long ctx_bs;            // context row b -> ctx_out + b*bs       (length E)
    int B, L, a, F, kl, E;
};
hipError_t launch_attention(const AttnParams& p, hipStream_t s);
size_t attention_lds_bytes(int L, int a, int F, int kl);

// ---------------------------------------------------------------------------------------------
// Small data-movement kernels.
// ---------------------------------------------------------------------------------------------
// x[b][p + l][:] = emb[tokens[b][l]][:]   into a halo-padded channels-last buffer (halo rows pre-zeroed)
hipError_t launch_embed(const int64_t* tokens, const float* emb, int n_tokens, float* x, int B, int L, int E, int halo,
                        int* err_flag, hipStream_t s);
// frames[(t)*B + b][m] = (t == 0) ? 0 : mel_in[b][m][t-1]     t in [0, T]
hipError_t launch_frames_from_mel(const float* mel_in, float* frames, int B, int M, int T, hipStream_t s);
// proj [B][T][M+1] (batch-major)  ->  mel_out [B][M][T], gate_out [B][T]
hipError_t launch_split_projection(const float* proj, float* mel_out, float* gate_out, int B, int M, int T, hipStream_t s);
// [B][M][T] -> halo-padded channels-last [B][T+2p][M]
hipError_t launch_to_channels_last(const float* src, float* dst, int B, int M, int T, int halo, hipStream_t s);
// mel_post[b][m][t] = mel[b][m][t] + y[b][t][m]
hipError_t launch_residual_to_channels_first(const float* mel, const float* y, float* mel_post, int B, int M, int T, hipStream_t s);
// zero the 2*halo halo rows of every sequence of a channels-last buffer [B][T+2*halo][C]
hipError_t launch_zero_halo(float* buf, int B, int T, int halo, int C, hipStream_t s);
hipError_t launch_mask_padding(float* mel, float* mel_post, float* gate, const int32_t* mel_lengths, int B, int M, int T,
                               hipStream_t s);
hipError_t launch_mask_gen(uint8_t* out, size_t n, uint64_t seed, hipStream_t s);
// AR: gate logits of step t -> per-row finished flags / frame counts / all-finished counter
hipError_t launch_ar_stop(const float* proj_t, long proj_stride, int gate_col, float threshold, int t, int B,
                          int32_t* n_frames, int32_t* n_done, hipStream_t s);
// AR: scatter step-t projection [B][M+1] into mel_out [B][M][Tmax], gate_out [B][Tmax]
hipError_t launch_ar_emit(const float* proj_t, long proj_stride, float* mel_out, float* gate_out, int B, int M, int Tmax, int t,
                          hipStream_t s);

}  // namespace gvx
