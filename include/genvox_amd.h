/*
 * genvox_amd.h — C ABI of the MI355X (gfx950) Tacotron2 text->mel forward path and the
 * batched Griffin-Lim vocoder.
 *
 * The reference (saiakarsh193/GenVox) has no FFI/operator layer: its boundary for this
 * path is the Python class surface.  This header is the boundary a
 * native replacement exposes underneath that surface; every entry point names the
 * reference function it replaces (paths relative to the reference root).  The Python
 * mirror in genvox_amd/ binds these with ctypes (see INTEGRATION.md for the stub).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch types.
 *  - Unless a parameter says "host", every pointer is a DEVICE pointer (e.g. tensor.data_ptr()
 *    of a PyTorch-ROCm tensor).  Outputs and workspaces are caller-allocated; nothing is
 *    retained after a call returns except the blob pointer given to gvx_model_bind_blob and
 *    cached hipGraphs of the step loops, which reference the workspace and the blob by address
 *    (keyed by both; re-binding a blob drops them).
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *    Calls enqueue work and return; they do not synchronise unless stated.
 *  - Every int-returning call returns GVX_OK (0) or a negative gvx_status;
 *    gvx_last_error() gives the message for the calling thread.
 *  - A gvx_model handle is not thread-safe; distinct handles may be used concurrently.
 *  - All floating point data is fp32 (the reference's dtype).  Channel dims must be
 *    multiples of 8 (GVX_ERR_UNSUPPORTED otherwise).
 */
#ifndef GENVOX_AMD_H
#define GENVOX_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gvx_status {
    GVX_OK = 0,
    GVX_ERR_INVALID_ARG = -1,
    GVX_ERR_UNSUPPORTED = -2,
    GVX_ERR_MISSING_WEIGHT = -3,
    GVX_ERR_SHAPE = -4,
    GVX_ERR_WORKSPACE = -5,
    GVX_ERR_HIP = -6,
    GVX_ERR_STATE = -7
} gvx_status;

/* Hyper-parameters that fix tensor shapes.  Mirrors configs/models.py:10-26 (Tacotron2Config),
 * configs/__init__.py:136 (AudioConfig.n_mels) and TextConfig.n_tokens (configs/__init__.py:81). */
typedef struct gvx_dims {
    int32_t n_tokens;          /* embedding rows                                   */
    int32_t embed_dim;         /* symbols_embedding_dim == encoder_embedding_dim   */
    int32_t enc_kernel;        /* encoder_kernel_size (odd)                        */
    int32_t enc_n_conv;        /* encoder_n_convolutions                           */
    int32_t prenet_dim;
    int32_t att_rnn_dim;
    int32_t dec_rnn_dim;
    int32_t att_dim;
    int32_t att_loc_filters;   /* attention_location_n_filters                     */
    int32_t att_loc_kernel;    /* attention_location_kernel_size (odd)             */
    int32_t postnet_dim;       /* postnet_embedding_dim                            */
    int32_t postnet_kernel;    /* postnet_kernel_size (odd)                        */
    int32_t postnet_n_conv;
    int32_t n_mels;
} gvx_dims;

/* One named fp32 tensor of the reference's state_dict (models/tts/tacotron2.py:574-584). */
typedef struct gvx_weight_desc {
    const char* name;      /* e.g. "decoder.attention_rnn.weight_ih"                      */
    const float* data;     /* HOST pointer (DEVICE for gvx_model_pack_weights_device), contiguous row-major */
    int64_t numel;
} gvx_weight_desc;

typedef struct gvx_model gvx_model;

const char* gvx_last_error(void);
int gvx_version(void);

/* ---- model handle + weights: replaces Tacotron2.__init__ / load_state_dict
 *      (models/tts/tacotron2.py:417-448, :581-584).
 * gvx_model_pack_weights folds eval-mode BatchNorm into the conv weights, reorders LSTM gate
 * rows and lays the recurrent matrices out in MFMA-fragment order, writing one contiguous
 * blob of gvx_model_blob_bytes() bytes to HOST memory.  The caller copies the blob to the
 * device (or receives it over RCCL, see gvx_model_blob_bytes) and binds it. */
int gvx_model_create(const gvx_dims* dims, gvx_model** out);
void gvx_model_destroy(gvx_model* model);
size_t gvx_model_blob_bytes(const gvx_model* model);
int gvx_model_pack_weights(gvx_model* model, const gvx_weight_desc* table, int n, void* host_blob);
int gvx_model_bind_blob(gvx_model* model, const void* device_blob);
/* The same packing on the device: `table` holds DEVICE pointers (the parameters where training updates them in place),
 * device_blob receives exactly the bytes gvx_model_pack_weights would produce.  The first call with a given list of
 * (name, numel) builds a gather map - by running the host packer over index-coded stand-ins - and keeps it in device
 * memory owned by the handle (5 bytes per blob float); later calls are one gather launch plus the BatchNorm folds and
 * bias sums.  This is what a training loop calls after every optimizer step (the host packer takes 80 ms per call). */
int gvx_model_pack_weights_device(gvx_model* model, const gvx_weight_desc* table, int n, void* device_blob, void* stream);

/* Bytes of scratch the calls below need for batch B, L tokens and up to T frames.  gvx_workspace_bytes covers every call;
 * gvx_workspace_bytes_autoregressive is the (smaller) amount gvx_encoder_forward + gvx_decoder_autoregressive +
 * gvx_postnet_forward need for max_steps frames (it leaves out the per-step buffers only the teacher-forced loop uses). */
size_t gvx_workspace_bytes(const gvx_model* model, int B, int L, int T);
size_t gvx_workspace_bytes_autoregressive(const gvx_model* model, int B, int L, int max_steps);

/* Device-side status words the kernels raise in the workspace, copied to host_out[2] after synchronising `stream`.  Both
 * are STICKY: they accumulate over every call made with this workspace since the last gvx_workspace_status, which clears them.
 *   [0] != 0: a token id was outside [0, n_tokens) (the reference's nn.Embedding raises IndexError there,
 *             models/tts/tacotron2.py:459; the row is embedded as zeros here) in some gvx_encoder_forward /
 *             gvx_tacotron2_forward call;
 *   [1] != 0: a bounded in-launch wait of a teacher-forced decoder loop gave up (the attention kernel that runs
 *             beside the LSTM launches and those launches hand the query / context over through counters in the
 *             workspace; a wait that is not served within a few hundred ms raises the call's time-out word and every
 *             kernel drains).  The outputs of such a call are NOT results and do not look like results: the call's
 *             last launch overwrites all of them (mel, mel_post, gate, alignments) with NaN.  It cannot happen while
 *             the two kernels run concurrently; the library switches the resident kernel off when the process runs
 *             under AMD_SERIALIZE_KERNEL / HIP_LAUNCH_BLOCKING.
 * Costs a stream synchronisation: meant for tests and for one check after a batch of calls, not for every call. */
int gvx_workspace_status(const gvx_model* model, void* workspace, size_t workspace_bytes, void* stream, int32_t* host_out);

/* Teacher-forced decoder loop: run the attention as ONE kernel that lives beside the step launches (default, used when the
 * shape allows it: B <= 32, L <= 128, default layer sizes) or as a launch per step (enable = 0).  Callers that drive one
 * handle from two streams at once (the host mirror does that for batches above 32 rows) must switch it off: the
 * resident kernel owns the handle's side stream for the whole loop.  Results are identical either way.
 * enable = 0 also marks the handle as one that shares the GPU with concurrent calls: the autoregressive loop then keeps its
 * attention step a launch of its own instead of running it beside 256 partial-sum tiles (two such launches at once queue
 * behind one another); results equal to fp32 rounding (the h_a columns are added in a different order). */
int gvx_model_set_persistent_attention(gvx_model* model, int enable);

/* enable = 0: no kernel of this handle waits for another one inside a launch any more - the encoder recurrence becomes a launch
 * per position, the teacher-forced decoder loop a launch pair per step (no resident attention / decoder kernel).  Same results
 * (to fp32 rounding), slower.  What a caller switches to when a call came back with the hand-off time-out status (NaN outputs,
 * gvx_workspace_status): the resident kernels could not run at the same time on this GPU (CU masking, a serialising profiler, a
 * co-tenant holding CUs) - the host mirror does exactly this and runs the call again (genvox_amd/tacotron2.py, forward(strict=True)).
 * enable = 1 restores the defaults (environment knobs are not re-read). */
int gvx_model_set_resident_kernels(gvx_model* model, int enable);

/* Batch rows gvx_tacotron2_forward / gvx_decoder_teacher_forced serve best per call for rows of L tokens: 64 where the 64-row
 * loop beside the resident attention kernel applies (default layer sizes, L <= 128, resident attention enabled: one pass over
 * the recurrent weights per step for all 64 rows), else 32 (callers with more rows run 32-row chunks - in turn where gvx_teacher_forced_resident says 1, else on two streams with a
 * handle each, as the host mirror does).  Any B in [1, 64] is accepted by every call regardless. */
int gvx_teacher_forced_rows_per_call(const gvx_model* model, int L);

/* 1 if a teacher-forced call with B rows of L tokens runs its decoder loop beside the resident attention kernel (the fast path:
 * such chunks are best run one after the other on one stream - 2 x 18.7 ms for 64 x 800 frames against 39.3 ms as two
 * concurrent lanes of launch-per-step loops), 0 if it takes a launch per attention step (chunks then gain from two streams). */
int gvx_teacher_forced_resident(const gvx_model* model, int B, int L);

/* How the decoder loop of a teacher-forced inference call with B rows of L tokens runs (models/tts/tacotron2.py:365-388):
 * 2 = ONE resident weight-stationary kernel for all T steps beside the resident attention kernel (dec_resident.hip: default layer
 *     sizes, B <= 32, L <= 128; the LSTM matrices stay in registers and LDS, hand-offs by flags),
 * 1 = one weight-streaming launch per step beside the resident attention kernel, 0 = a launch pair per step. */
int gvx_teacher_forced_loop_kind(const gvx_model* model, int B, int L);

/* How an autoregressive call with B rows of L tokens runs its decode (models/tts/tacotron2.py:390-413 Decoder.inference):
 * 2 = TWO resident kernels for the whole decode (dec_resident.hip decoder_ar_resident_kernel beside attn_persist.hip's rows, which
 *     also sum the frame, test the stop token and run Prenet layer 1; default layer sizes, B <= 32, L <= 128, the handle does not
 *     share the chip): no launch per step, the kernels end the loop themselves,
 * 1 = launches per step beside the resident attention kernel (opt-in, GVX_AR_RESIDENT=1), 0 = launches per step. */
int gvx_autoregressive_loop_kind(const gvx_model* model, int B, int L);

/* ---- Encoder: embedding + conv/BN/relu stack + BiLSTM with packed-sequence semantics.
 * Replaces nn.Embedding + Encoder.forward / Encoder.inference (models/tts/tacotron2.py:459,
 * :231-246, :248-256).  tokens: int64 [B, L]; lengths: int32 [B] or NULL (= all L);
 * memory_out: [B, L, embed_dim], zero past each row's length.
 * For B <= 32 and the default layer sizes the recurrence (:239-245) is ONE resident launch whose 64 workgroups hand the hidden
 * state round through the workspace (bounded waits: a time-out leaves NaN in memory_out and raises status word [1] of
 * gvx_workspace_status); other shapes run a launch per token position.  Same results to fp32 rounding. */
int gvx_encoder_forward(gvx_model* model, const int64_t* tokens, const int32_t* lengths, int B, int L,
                        float* memory_out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- Teacher-forced decoder: Prenet over all frames, T x (attention LSTM, location-sensitive
 * attention, decoder LSTM), mel/gate projection.  Replaces Decoder.forward
 * (models/tts/tacotron2.py:365-388; decode :333-363; Attention :89-129; Prenet :140-144).
 * memory: [B, L, embed_dim]; lengths: int32 [B] or NULL; mel_in: [B, n_mels, T];
 * keep_masks: uint8 {0,1} [2, (T+1)*B, prenet_dim], row = t*B + b (the two Prenet dropout keep masks);
 * mel_out: [B, n_mels, T]; gate_out: [B, T] (logits); align_out: [B, T, L]. */
int gvx_decoder_teacher_forced(gvx_model* model, const float* memory, const int32_t* lengths, int B, int L,
                               const float* mel_in, int T, const uint8_t* keep_masks,
                               float* mel_out, float* gate_out, float* align_out,
                               void* workspace, size_t workspace_bytes, void* stream);

/* ---- Autoregressive decoder, batched.  Replaces Decoder.inference (models/tts/tacotron2.py:390-414),
 * which is batch-1 only; here every row stops on its own: n_frames_out[b] = index of the first step
 * whose sigmoid(gate) > gate_threshold, plus one (or max_steps).  Frames past n_frames_out[b] carry the
 * reference's padding values (mel 0, gate 1e3, alignment 0; mask_padding, :466-473).  keep_masks: uint8 [2, max_steps, B, prenet_dim].  Outputs are sized for max_steps:
 * mel_out [B, n_mels, max_steps], gate_out [B, max_steps], align_out [B, max_steps, L].
 * This call synchronises the stream (it polls the all-rows-finished flag between step chunks).
 * steps_run_out (HOST int) receives the number of steps actually executed. */
int gvx_decoder_autoregressive(gvx_model* model, const float* memory, const int32_t* lengths, int B, int L,
                               int max_steps, float gate_threshold, const uint8_t* keep_masks,
                               float* mel_out, float* gate_out, float* align_out, int32_t* n_frames_out,
                               int* steps_run_out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- Postnet + residual: mel_post_out = mel_in + Postnet(mel_in).  Replaces Postnet.forward and the
 * residual add (models/tts/tacotron2.py:194-200, :464/:491).  Tensors are [B, n_mels, T]; any B (GEMM-only
 * path, no per-call batch limit).  mel_lengths: int32 [B] or NULL.  With lengths, row b is processed as a
 * sequence of mel_lengths[b] frames (the convolutions see zeros from that frame on, as a batch-1 run of the
 * reference sees its zero padding) and mel_post_out is 0 there.  The workspace needs
 * gvx_postnet_workspace_bytes(B, T) bytes (a gvx_workspace_bytes(B, L, T) workspace is always large enough). */
size_t gvx_postnet_workspace_bytes(const gvx_model* model, int B, int T);
int gvx_postnet_forward(gvx_model* model, const float* mel_in, const int32_t* mel_lengths, int B, int T, float* mel_post_out,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ---- Output padding mask (models/tts/tacotron2.py:466-473): frames >= mel_lengths[b] get
 * mel = 0, mel_post = 0, gate = 1e3.  mel_lengths: int32 [B]. In place. */
int gvx_mask_padding(float* mel, float* mel_post, float* gate, const int32_t* mel_lengths,
                     int B, int n_mels, int T, void* stream);

/* ---- Whole teacher-forced forward in one call (what Tacotron2.forward does, models/tts/tacotron2.py:450-481).
 * mel_lengths may be NULL (no padding mask).  mel_post_out may be NULL: the call then ends behind the mel / gate projection
 * (mel_out, gate_out, align_out unmasked) and the caller runs gvx_postnet_forward + gvx_mask_padding itself - once over all
 * 32-row chunks of a larger batch, for instance. */
int gvx_tacotron2_forward(gvx_model* model, const int64_t* tokens, const int32_t* token_lengths, int B, int L,
                          const float* mel_in, const int32_t* mel_lengths, int T, const uint8_t* keep_masks,
                          float* mel_out, float* mel_post_out, float* gate_out, float* align_out,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ---- Criterion of the evaluation step.  Replaces Tacotron2Loss (models/tts/tacotron2.py:598-615) as called by
 * Tacotron2.eval_step (:524-529): loss_out[3] (device) = {loss, mel_loss, gate_loss} with
 *   mel_loss = mean((mel_out - mel_target)^2) + mean((mel_post_out - mel_target)^2)   over all B*n_mels*T elements,
 *   gate_loss = mean(BCE-with-logits(gate_out, gate_target))                          over all B*T elements,
 *   loss = mel_loss + gate_loss.
 * All tensors are the [B, n_mels, T] / [B, T] device arrays of the forward call; forward only (no gradients).
 * scratch: >= 6144 bytes of device memory, 8-byte aligned (float64 partial sums of the two-stage reduction). */
int gvx_tacotron2_loss(const float* mel_out, const float* mel_post_out, const float* gate_out, const float* mel_target,
                       const float* gate_target, int B, int n_mels, int T, float* loss_out, void* scratch, size_t scratch_bytes,
                       void* stream);

/* =====================================================================================================
 * Training: the convolution layers of the encoder and Postnet stacks as the
 * reference runs them under .train() - nn.Conv1d + nn.BatchNorm1d with BATCH statistics + activation + F.dropout
 * (models/tts/tacotron2.py:149-199, :207-220, :234-235) - forward and backward, and the backward of Tacotron2Loss
 * (:598-615).  Parameters are taken in the reference's own layout (conv weight [Cout, Cin, k]); activations in its
 * [B, C, T] layout.  act: 0 none, 1 relu, 2 tanh.  keep: uint8 {0,1} [B, Cout, T] (the dropout's keep mask; NULL = no
 * dropout); kept values are scaled by 1 / (1 - p).  Channels must be multiples of 8, k odd.
 * `saved` carries what the backward needs (gvx_conv_train_saved_bytes); both buffers 256-byte aligned.
 * Together with the BPTT primitives further down these make up Tacotron2.train_step of the host mirror
 * (genvox_amd/tacotron2.py, genvox_amd/training.py), which is pinned to the reference's own train_step.
 * ===================================================================================================== */
/* The rest of the training-mode FORWARD (models/tts/tacotron2.py:231-246, :333-363 under .train()): the encoder's BiLSTM on
 * the output of its training-mode convolution stack (conv_out [B, embed_dim, L], reference layout), and the teacher-forced
 * decoder with dropout on the hidden outputs of both LSTM cells (att_keep uint8 [T, B, att_rnn_dim], dec_keep uint8
 * [T, B, dec_rnn_dim]; kept values scaled by 1 / (1 - p); the dropped state is what the next step, the attention query, the
 * other cell and the projection see).  Both read the LSTM / attention / Prenet / projection weights from the bound blob. */
/* Tape outputs (each may be NULL) are what back-propagation through time reads:
 *   cell_states_out [B, L, embed_dim]   the BiLSTM's cell state at every position (forward direction in the first half of
 *                                       the channels, like memory_out), zeros past a row's length;
 *   input_preact_out [B, L, 2 * 4H]     W_ih x + b_ih + b_hh of both directions, gate rows in the library's packed order
 *                                       (row 4 j + gate);
 *   att_hidden_all [T+1][A/8][B][8]     the attention LSTM's dropped hidden state after every step (slot t + 1; slot 0 zeros)
 *                                       as k-group-blocked vectors: element (b, k) at (k / 8) * B * 8 + b * 8 + k % 8;
 *   att_cell_all, dec_cell_all [T+1][B][H]   both cells' states (slot t + 1 = after step t);
 *   dec_hidden_context_all [T+1][(D+E)/8][B][8]   [h_d ; ctx] after every step, blocked as above;
 *   att_preact_all [T][B][A][4], dec_preact_all [T][B][D][4]   gate pre-activations i, f, g, o of every unit and step. */
int gvx_encoder_lstm_forward(gvx_model* model, const float* conv_out, const int32_t* lengths, int B, int L, float* memory_out,
                             float* cell_states_out, float* input_preact_out, void* workspace, size_t workspace_bytes, void* stream);
int gvx_decoder_teacher_forced_train(gvx_model* model, const float* memory, const int32_t* lengths, int B, int L, const float* mel_in,
                                     int T, const uint8_t* keep_masks, const uint8_t* att_keep, const uint8_t* dec_keep, float p_att,
                                     float p_dec, float* mel_out, float* gate_out, float* align_out, float* att_hidden_all,
                                     float* att_cell_all, float* dec_cell_all, float* dec_hidden_context_all, float* att_preact_all,
                                     float* dec_preact_all, void* workspace, size_t workspace_bytes, void* stream);
/* Copy one of the decoder's per-call buffers out of the workspace of the last gvx_decoder_teacher_forced(_train) call with the
 * same (B, L, T), row-major: what 0 = decoder input frames [(T+1) B, n_mels] (row t B + b; frame 0 = zeros), 1 = Prenet layer-1
 * output [(T+1) B, prenet_dim], 2 = Prenet output [(T+1) B, prenet_dim], 3 = processed memory [B, L, att_dim]. */
int gvx_train_export(const gvx_model* model, const void* workspace, size_t workspace_bytes, int B, int L, int T, int what, float* dst, void* stream);
size_t gvx_conv_train_saved_bytes(int B, int Cin, int Cout, int T, int k);
size_t gvx_conv_train_workspace_bytes(int B, int Cin, int Cout, int T, int k);
/* y = dropout(act(BatchNorm_train(conv1d(x, w, bias, padding (k-1)/2)))).  running_mean / running_var (may be NULL) get the
 * momentum-0.1 update of torch.nn.BatchNorm1d (unbiased variance), in place. */
int gvx_conv_bn_act_train_forward(const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, int B, int Cin, int Cout, int T, int k, int act,
                                  const uint8_t* keep, float p_drop, float* y, void* saved, size_t saved_bytes, void* workspace,
                                  size_t workspace_bytes, void* stream);
/* Gradients of one layer given dy = d loss / d y: dx [B, Cin, T] (may be NULL), dw [Cout, Cin, k], dbias, dgamma, dbeta [Cout].
 * x_wgrad (may be NULL): input to use for the weight gradient instead of the saved one - the reference masks the Postnet's
 * input in place after its forward, outside autograd, so its first layer's weight gradient sees the masked tensor. */
int gvx_conv_bn_act_train_backward(const float* dy, const void* saved, size_t saved_bytes, const float* w, const float* gamma,
                                   const float* x_wgrad, int B, int Cin, int Cout, int T, int k, int act, const uint8_t* keep,
                                   float p_drop, float* dx, float* dw, float* dbias, float* dgamma, float* dbeta, void* workspace,
                                   size_t workspace_bytes, void* stream);
/* d loss / d (mel_out [its own MSE term], mel_post_out, gate_out) of Tacotron2Loss on the (masked) outputs of the forward. */
int gvx_tacotron2_loss_backward(const float* mel_out, const float* mel_post_out, const float* gate_out, const float* mel_target,
                                const float* gate_target, int B, int n_mels, int T, float* dmel_out, float* dmel_post_out,
                                float* dgate_out, void* stream);

/* ---- The training step's whole-sequence pieces: primitives the host mirror (genvox_amd/training.py) strings together as
 * oracle/train_ref.py states them (the reference: loss.backward(), clip_grad_norm_, Adam.step, models/tts/tacotron2.py:515-522);
 * the two recurrences are single calls (gvx_train_decoder_bptt, gvx_train_encoder_lstm_bptt below).  Row-major fp32 with
 * explicit leading dimensions; LSTM gates in torch order i, f, g, o. */
/* C[m][n] = sum_k A[m * lda + k] W[n * ldw + k] (+ bias[n]), K % 4 == 0.  scratch (may be NULL): device scratch for split-K
 * partial tiles, used when the product has few output tiles and a long K. */
int gvx_train_gemm_nt(const float* A, long lda, const float* W, long ldw, float* C, long ldc, int M, int N, int K, const float* bias,
                      float* scratch, size_t scratch_bytes, void* stream);
/* C[m][n] = sum_r A[r * lda + m] Bm[r * ldb + n] (a weight gradient: the sum over the rows of two activation matrices, no
 * transposed copies). */
int gvx_train_gemm_tn(const float* A, long lda, const float* Bm, long ldb, float* C, long ldc, int M, int N, long rows, float* scratch,
                      size_t scratch_bytes, void* stream);
int gvx_train_transpose(const float* src, long ld_src, float* dst, long rows, int cols, long rows_padded, void* stream);
int gvx_train_colsum(const float* X, long rows, int C, float* out, void* stream);
int gvx_train_axpby(const float* a, long lda, float alpha, const float* b, long ldb, float beta, float* y, long ldy, long rows, int cols, void* stream);
int gvx_train_relu_dropout_backward(const float* dy, const float* act_out, const uint8_t* keep, float scale, long n, float* dz, void* stream);
int gvx_train_unblock(const float* blocked, float* rows_out, long n_slots, int B, int K, void* stream);
int gvx_train_embedding_backward(const int64_t* tokens, const float* dx, long n_tokens_in_batch, int E, int n_rows, float* demb, void* stream);
/* Many tensors per launch (device arrays of references, built by the caller once per step): sum of squares of all of them -
 * the square of clip_grad_norm_'s total norm, added in a fixed order (scratch: gvx_train_sqnorm_scratch_bytes) - and the Adam
 * update of all of them. */
typedef struct gvx_tensor_ref { const float* data; int64_t numel; } gvx_tensor_ref;
typedef struct gvx_adam_ref { float* param; const float* grad; float* exp_avg; float* exp_avg_sq; int64_t numel; } gvx_adam_ref;
size_t gvx_train_sqnorm_scratch_bytes(int n_tensors);
int gvx_train_sqnorm_many(const gvx_tensor_ref* refs_device, int n_tensors, double* scratch, double* sumsq_out, void* stream);
int gvx_train_adam_step_many(const gvx_adam_ref* refs_device, int n_tensors, float grad_scale, float lr, float weight_decay, float beta1,
                             float beta2, float eps, int step, void* stream);

/* ---- Back-propagation through the decoder loop in one call (three launches per step issued by the library instead of ~20
 * primitives per step strung together by the host): d loss / d of both LSTM cells' gates at every step, of the attention
 * queries, of the processed memory, of the memory through the contexts, and of v / location_dense / location_conv
 * (Decoder.forward backwards, models/tts/tacotron2.py:365-388 with :333-363 and Attention :89-129).  What is not on the
 * recurrence (weight gradients as whole-sequence products, the Prenet columns of the attention LSTM) stays with the caller.
 * All pointers device, fp32 row-major, gates in torch order i, f, g, o. */
typedef struct gvx_bptt_decoder_args {
    int32_t B, L, T;                    /* B <= 32 */
    int32_t A, D, E, P, a, F, kl;       /* att_rnn_dim, dec_rnn_dim, embed_dim, prenet_dim, att_dim, location filters / kernel size */
    float att_scale, dec_scale;         /* 1 / (1 - p) of the dropout on each cell's hidden output */
    const float* dhc_all;               /* [T][B][D+E]  d loss / d [h_d(t) ; ctx(t)] through the mel / gate projection */
    const float* pre_a;                 /* [T][B][A][4] gate pre-activations of the attention LSTM as the forward's tape holds them */
    const float* pre_d;                 /* [T][B][D][4] ... of the decoder LSTM */
    const float* c_a_all;               /* [T+1][B][A]  cell states, slot t = before step t */
    const float* c_d_all;               /* [T+1][B][D] */
    const uint8_t* att_keep;            /* [T][B][A]    keep masks of the hidden-output dropouts */
    const uint8_t* dec_keep;            /* [T][B][D] */
    const float* q_all;                 /* [T][B][a]    attention queries W_q h_a(t) */
    const float* ctx_all;               /* context of (step t, row b) at ctx_all + t * ctx_ts + b * ctx_bs, E floats */
    int64_t ctx_ts, ctx_bs;
    const float* w_all;                 /* [T][B][L]    alignments, time-major */
    const float* memory;                /* [B][L][E] */
    const float* pm;                    /* [B][L][a]    processed memory */
    const float* w_ih_a; const float* w_hh_a;   /* attention_rnn.weight_ih [4A][P+E], weight_hh [4A][A] */
    const float* w_ih_d; const float* w_hh_d;   /* decoder_rnn.weight_ih [4D][A+E], weight_hh [4D][D] */
    const float* wq;                    /* query_layer weight [a][A] */
    const float* v;                     /* [a] */
    const float* loc_conv;              /* location_conv weight [F][2][kl] */
    const float* loc_dense;             /* location_dense weight [a][F] */
    float* dga_all;                     /* out [T][B][4A]  d loss / d gate pre-activations of the attention LSTM */
    float* dgd_all;                     /* out [T][B][4D] */
    float* dq_all;                      /* out [T][B][a] */
    float* dctx_all;                    /* out [T][B][E]   total d loss / d ctx(t) */
    float* dpm;                         /* out [B][L][a] */
    float* dmemory;                     /* out [B][L][E]   context path only: sum_t w_t (x) dctx_t */
    float* dv;                          /* out [a] */
    float* dloc_dense;                  /* out [a][F] */
    float* dloc_conv;                   /* out [F][2][kl] */
} gvx_bptt_decoder_args;
size_t gvx_train_decoder_bptt_workspace_bytes(const gvx_bptt_decoder_args* args);
int gvx_train_decoder_bptt(const gvx_bptt_decoder_args* args, void* workspace, size_t workspace_bytes, void* stream);
/* Back-propagation through the encoder BiLSTM (Encoder.forward, models/tts/tacotron2.py:239-245, packed-sequence semantics),
 * one launch per time step for both directions.  xg [2][B][L][4H] = W_ih x + b_ih + b_hh per direction; memory / cell_states /
 * dmemory [B][L][2H] (forward direction in the first H channels); w_hh [2][4H][H].  Outputs, filed under the POSITION a step
 * belongs to (zeros past a row's length): dg_pos [2][B][L][4H] gate gradients, hprev_pos [2][B][L][H] the step's previous
 * hidden state - the operands of the weight gradients and of d loss / d x. */
size_t gvx_train_encoder_lstm_bptt_workspace_bytes(int B, int H);
int gvx_train_encoder_lstm_bptt(const float* xg, const float* memory, const float* cell_states, const float* dmemory, const float* w_hh,
                                const int32_t* lengths, int B, int L, int H, float* dg_pos, float* hprev_pos, void* workspace,
                                size_t workspace_bytes, void* stream);
/* The same walk as ONE resident launch (its workgroups hand the gate gradients of a step round through the workspace; shapes
 * whose workgroups do not all fit on the GPU at once take the launch per step).  Bounded waits: after a time-out - the workgroups
 * could not run at the same time - both outputs are NaN and the workspace's status word holds the code, which
 * gvx_train_encoder_lstm_bptt_status reads (one stream synchronisation; 0 = fine). */
int gvx_train_encoder_lstm_bptt_resident(const float* xg, const float* memory, const float* cell_states, const float* dmemory,
                                         const float* w_hh, const int32_t* lengths, int B, int L, int H, float* dg_pos,
                                         float* hprev_pos, void* workspace, size_t workspace_bytes, void* stream);
int gvx_train_encoder_lstm_bptt_status(const void* workspace, size_t workspace_bytes, int B, int H, int* code_out, void* stream);

/* ---- Prenet keep-mask generator for callers that do not supply masks (the reference draws them from
 * torch's RNG inside F.dropout, models/tts/tacotron2.py:143).  Writes n bytes of Bernoulli(0.5) {0,1}. */
int gvx_prenet_masks_generate(uint8_t* masks_out, size_t n, uint64_t seed, void* stream);

/* ---- Stage timing (measurement only): when enabled, the whole-forward call records HIP events around
 * each stage on `stream`; gvx_stage_times_ms synchronises and returns encoder, prenet, decoder loop,
 * projection, postnet milliseconds of the last call and the number of decoder-step kernel launches. */
int gvx_stage_timing_enable(gvx_model* model, int enable);
int gvx_stage_times_ms(gvx_model* model, float* times5_out, int* decoder_launches_out);

/* =====================================================================================================
 * Vocoder: mel (dB) -> waveform.  Replaces the mel->wav half of utils/audio/base.py and
 * AudioProcessor.convert_mel2wav (core/processors.py:81-96), batched over utterances.
 * Spectrograms use the reference's layout [B][bins][frames] (bins = n_fft/2 + 1); complex values are
 * interleaved (re, im) floats.  `window` is the float32 periodic Hann window [n_fft] (device).
 * A gvx_gl_plan owns the rocFFT plans (one pair per distinct B*T, created on first use).
 * ===================================================================================================== */
typedef struct gvx_gl_plan gvx_gl_plan;
int gvx_gl_plan_create(int n_fft, int hop, gvx_gl_plan** out);
void gvx_gl_plan_destroy(gvx_gl_plan* plan);
size_t gvx_gl_workspace_bytes(gvx_gl_plan* plan, int B, int T, int n_mels);

/* stft (utils/audio/base.py:58-69): signal [B][n_samples] -> spec_out complex [B][bins][T], T = (n_samples-n_fft)/hop+1 */
int gvx_stft(gvx_gl_plan* plan, const float* signal, const float* window, int B, long n_samples, float* spec_out,
             void* workspace, size_t workspace_bytes, void* stream);
/* istft (utils/audio/base.py:71-88): spec complex [B][bins][T] -> signal_out [B][n_fft + (T-1)*hop] */
int gvx_istft(gvx_gl_plan* plan, const float* spec, const float* window, int B, int T, float* signal_out,
              void* workspace, size_t workspace_bytes, void* stream);
/* db_to_amplitude + mel2fft (utils/audio/base.py:38-52 with power=False/scale=1, :143-145):
 * mel_db [B][n_mels][T], inv_basis [bins][n_mels] -> mag_out [B][bins][T].  log10_kind: 0 = np.log, 1 = np.log10. */
int gvx_mel_to_magnitude(gvx_gl_plan* plan, const float* mel_db, const float* inv_basis, int B, int n_mels, int T, int log10_kind,
                         float ref, float* mag_out, void* workspace, size_t workspace_bytes, void* stream);
/* fast Griffin-Lim (utils/audio/base.py:147-162) + final synthesis istft(mag * exp(i phase)) (core/processors.py:89-90).
 * mag [B][bins][T]; phase_out [B][bins][T] or NULL; wav_out [B][n_fft + (T-1)*hop] or NULL. */
int gvx_griffin_lim(gvx_gl_plan* plan, const float* mag, const float* window, int B, int T, int n_iter, float momentum,
                    float* phase_out, float* wav_out, void* workspace, size_t workspace_bytes, void* stream);
/* wav -> mel features (AudioProcessor.convert_wav2mel, core/processors.py:70-79: stft, |.|, fft2mel, amplitude_to_db with
 * power=False/scale=1; utils/audio/base.py:24-36, :58-69, :139-141).  signal [B][n_samples] (already normalised),
 * mel_basis [n_mels][bins] -> mel_db_out [B][n_mels][T], T = (n_samples - n_fft)/hop + 1. */
int gvx_wav_to_mel(gvx_gl_plan* plan, const float* signal, const float* window, const float* mel_basis, int B, long n_samples,
                   int n_mels, int log10_kind, float ref, float* mel_db_out, void* workspace, size_t workspace_bytes, void* stream);
/* tail of convert_mel2wav (core/processors.py:91-95): samples with |y| > 1 -> 0, drop `trim` samples at both ends,
 * divide by the peak (float32), IIR filter b/a (HOST doubles, order+1 each; scipy.signal.lfilter semantics, float64).
 * out: float64 [B][n_samples - 2*trim]; scratch_B: B uint32 of device scratch. */
int gvx_wav_finalize(const float* wav, int B, long n_samples, int trim, const double* b_coef, const double* a_coef, int order,
                     double* out, unsigned int* scratch_B, void* stream);

/* ---- Per-kernel timing of the decoder step (measurement only): when enabled, a teacher-forced call replays the
 * mid-sequence LSTM-step launch and the attention launches 64 times each, back to back, between HIP events on
 * `stream` (after its loop; the call's outputs are not valid afterwards); gvx_kernel_times_ms synchronises and
 * returns the average duration of each (ms) and the number of replays. */
int gvx_kernel_timing_enable(gvx_model* model, int enable);
int gvx_kernel_times_ms(gvx_model* model, float* lstm_avg_ms_out, float* attn_avg_ms_out, int* n_steps_out);

#ifdef __cplusplus
}
#endif
#endif /* GENVOX_AMD_H */
