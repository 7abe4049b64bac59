/*
 * speech_amd.h -- C ABI of libspeech_amd.so: the MI355X (gfx950) implementation of the awni/speech CTC hot path.
 *
 * Plain C: pointers, sizes, a stream handle.  No torch types, no C++ in the signatures.  Every pointer named
 * d_* / acts / grads / workspace is DEVICE memory owned by the caller.  All work is enqueued on the caller's stream (a
 * hipStream_t passed as void*; NULL = the default stream); only the entry points that return HOST results (marked
 * SYNC) wait for it.  Every function returns a ctcStatus_t; nothing throws.
 * State: the entry points allocate nothing and keep no state between calls -- they are re-entrant -- with ONE stated
 * exception, the GRU stack entry points (sa_gru_stack_*, sa_gru_health_flag, sa_gru_persist_*).  Those own, PER
 * DEVICE (the device current at the call; 16 at most) and created at first use: a two-word sticky health word in
 * device memory with a ring of 8 pinned host words and events behind it, one non-blocking side stream with 64 events
 * (weight-gradient / projection GEMMs beside a recurrence), and a per-device mutex that serialises the stack calls of
 * several host threads on one device (held while launches are enqueued: microseconds).  Different devices share
 * nothing.  The opt-in profiler (sa_gru_profile_*) is the one piece of process-wide state and serves one device.
 * The library never reads the process environment; what can be steered is the table of named options at the end of this
 * file (sa_set_option), read when an entry point is called.
 *
 * Each entry point names the reference interface it replaces (paths relative to /root/reference).
 */
#ifndef SPEECH_AMD_H
#define SPEECH_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------------------------
 * 1. The warp-ctc C API.  The reference binds it through cffi: `import functions.ctc as ctc`
 *    (speech/models/ctc_model.py:9) -> ctc.CTCLoss()(out, y, x_lens, y_lens) (ctc_model.py:38-39); the library is
 *    cloned un-pinned by Makefile:4-7 and is NOT in the reference tree, so the shapes below restate upstream
 *    warp-ctc's public ctc.h from memory (SURVEY.md 8b "[recalled]").  Exported under the same names so that the
 *    binding's `gpu_ctc` can link against this library unchanged.
 * ----------------------------------------------------------------------------------------------------------------*/
typedef enum {
    CTC_STATUS_SUCCESS = 0,
    CTC_STATUS_MEMOPS_FAILED = 1,
    CTC_STATUS_INVALID_VALUE = 2,
    CTC_STATUS_EXECUTION_FAILED = 3,
    CTC_STATUS_UNKNOWN_ERROR = 4
} ctcStatus_t;

typedef enum { CTC_CPU = 0, CTC_GPU = 1 } ctcComputeLocation;

typedef struct ctcOptions {
    ctcComputeLocation loc; /* must be CTC_GPU: this library has no CPU path (CTC_CPU -> CTC_STATUS_EXECUTION_FAILED) */
    union {
        unsigned int num_threads; /* unused */
        void* stream;             /* hipStream_t */
    };
    int blank_label; /* the reference model uses blank = alphabet_size - 1 (ctc_model.py:18) */
} ctcOptions;

int get_warpctc_version(void);
const char* ctcGetStatusString(ctcStatus_t status);

/* activations: DEVICE, (T, B, alphabet_size) row-major, un-normalised logits (softmax is applied inside).
 * gradients:   DEVICE, same shape, or NULL for score-only.   flat_labels / label_lengths / input_lengths / costs: HOST.
 * costs[b] = -log p(labels_b | acts_b); infeasible alignment -> +inf with zero gradient; rows t >= input_lengths[b]
 * get zero gradient.  SYNC (costs are returned on the host). */
ctcStatus_t compute_ctc_loss(const float* activations, float* gradients, const int* flat_labels,
                             const int* label_lengths, const int* input_lengths, int alphabet_size, int minibatch,
                             float* costs, void* workspace, ctcOptions options);

ctcStatus_t get_workspace_size(const int* label_lengths, const int* input_lengths, int alphabet_size, int minibatch,
                               ctcOptions options, size_t* size_bytes);

/* ------------------------------------------------------------------------------------------------------------------
 * 2. Stream-ordered, layout-agnostic CTC loss -- what speech_amd.ctc.CTCLoss (the replacement for
 *    functions.ctc.CTCLoss, ctc_model.py:38-39) calls.  The model hands over batch-first logits (B, T', V+1)
 *    (ctc_model.py:29-32,36); strides let both layouts run without a transpose copy:
 *        acts[b * stride_b + t * stride_t + k],  grads likewise.
 *    All pointers DEVICE.  d_costs[b] per-utterance cost.  No host synchronisation.
 *    Below 512 utterances per call the alpha / beta chains run in the probability domain (no exp2 / log2 on the T-step
 *    dependent chain), certified at run time -- range kept per batch of steps, flow conservation of every lattice row --
 *    and an utterance that fails a check is recomputed by the log-domain kernels launched behind (csrc/ctc_loss.hip,
 *    ctc_chain_p).  Option "ctc.prob" = 0 selects the log-domain kernels alone.
 * ----------------------------------------------------------------------------------------------------------------*/
size_t sa_ctc_workspace_bytes(int max_T, int max_L, int alphabet_size, int minibatch);

/* Diagnostic: byte offset, inside that workspace, of int flags[minibatch] left by the last sa_ctc_loss call with gradients
 * (0 = the probability-domain result stands, else the utterance came from the log-domain kernels). */
size_t sa_ctc_flags_offset(int max_T, int max_L, int alphabet_size, int minibatch);

ctcStatus_t sa_ctc_loss(const float* acts, float* grads /* or NULL */, long stride_t, long stride_b,
                        const int* d_flat_labels, const int* d_label_lengths, const int* d_input_lengths,
                        int alphabet_size, int minibatch, int max_T, int max_L, int blank_label, float* d_costs,
                        void* workspace, size_t workspace_bytes, void* stream);

/* The same call with the reduction the model's loss applies folded in (CTC.loss, ctc_model.py:34-40 -> train.py:29-33:
 * one scalar per batch): d_loss[0] = scale * sum_b d_costs[b] (fixed summation order) and every gradient element is
 * multiplied by `scale` as it is written (scale = 1 / batch size for size_average; a data-parallel rank passes
 * 1 / GLOBAL batch size).  No separate reduction / scaling pass over the gradient exists. */
ctcStatus_t sa_ctc_loss_reduced(const float* acts, float* grads /* or NULL */, long stride_t, long stride_b,
                                const int* d_flat_labels, const int* d_label_lengths, const int* d_input_lengths,
                                int alphabet_size, int minibatch, int max_T, int max_L, int blank_label, float scale,
                                float* d_costs, float* d_loss, void* workspace, size_t workspace_bytes, void* stream);

/* y[i] *= *d_factor (a DEVICE scalar, e.g. the gradient autograd hands the loss), skipped when the factor is exactly 1. */
ctcStatus_t sa_scale_by_device_scalar(float* y, size_t n, const float* d_factor, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * 3. CTC decoding.
 *    sa_ctc_beam_decode  replaces speech/models/ctc_decoder.py:38-113 decode(probs, beam_size, blank) as called per
 *        utterance by CTC.infer (ctc_model.py:55-60, beam_size = 1, blank = V) -- same float32/double arithmetic,
 *        same first-touch tie order, same stable descending sort.  input_is_logits != 0 fuses the softmax of
 *        ctc_model.py:30-31 (probs = softmax(logits), then log, as the reference does).
 *    sa_ctc_greedy_decode replaces np.argmax + CTC.max_decode (ctc_model.py:62-70).
 *    in[b * stride_b + t * stride_t + s]; d_out_labels (B, max_T) int32; d_out_lens (B); d_out_nll (B) float or NULL.
 * ----------------------------------------------------------------------------------------------------------------*/
size_t sa_ctc_beam_workspace_bytes(int max_T, int alphabet_size, int minibatch, int beam_size);

ctcStatus_t sa_ctc_beam_decode(const float* in, long stride_t, long stride_b, const int* d_input_lengths,
                               int alphabet_size, int minibatch, int max_T, int beam_size, int blank_label,
                               int input_is_logits, int* d_out_labels, int* d_out_lens, float* d_out_nll,
                               void* workspace, size_t workspace_bytes, void* stream);

ctcStatus_t sa_ctc_greedy_decode(const float* in, long stride_t, long stride_b, const int* d_input_lengths,
                                 int alphabet_size, int minibatch, int max_T, int blank_label, int* d_out_labels,
                                 int* d_out_lens, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * 4. Encoder building blocks (speech/models/model.py:60-79 Model.encode and ctc_model.py:19,29 fc).
 *    fp32 storage and fp32 accumulation throughout.  Arithmetic of the matrix products:
 *      - recurrences, convolutions, the decoder's small products and every GEMM below 8 GFLOP: the f32-input MFMA
 *        (v_mfma_f32_*_f32: exact fp32 products);
 *      - GEMMs of at least 8 GFLOP with K >= 256 and M, N >= 64 (sa_gemm_is_split_bf16 -- a function of the shape ONLY):
 *        each fp32 operand is split exactly into three bf16 pieces (a = a1 + a2 + a3) and six of the nine piece products
 *        run on the bf16 MFMA with fp32 accumulation; the dropped terms are below 2^-26 of each product, the result is
 *        within the fp32 accumulation bound 2^-24 (2 + sqrt(K)) |A||B| and measures at or below the f32-input kernel's
 *        error against fp64 (tests/test_gpu_blocks.py::test_split_bf16_gemm_error_budget); integer data stays bit-exact.
 *        Option "gemm.exact" = 1 selects the f32-input kernel for every product, = 0 the split path for every
 *        product (tests); nothing else -- in particular not the workspace size -- changes the arithmetic.
 * ----------------------------------------------------------------------------------------------------------------*/

/* C[M,N] = alpha * op(A) * op(B) (+ bias[N]) (+ beta * C),  row-major, leading dimensions in elements.
 *   trans_a == 0: A is (M,K), lda >= K;  trans_a != 0: A is stored (K,M), lda >= M.
 *   trans_b == 0: B is (K,N), ldb >= N;  trans_b != 0: B is stored (N,K), ldb >= K   (nn.Linear / GRU weight layout).
 * Replaces the cuBLAS calls under nn.Linear (model.py:126-133) and nn.GRU's input projections (model.py:35-39).
 * Products with few output tiles and a long K (the weight gradients, K = B*T') are split along K into `workspace`
 * and reduced in a fixed order.  workspace >= sa_gemm_workspace_bytes(M, N, K).  A product that runs the split-bf16
 * path (sa_gemm_is_split_bf16) NEEDS that workspace for its packed operands: with less the call returns
 * CTC_STATUS_INVALID_VALUE (it never changes arithmetic silently).  An exact-path product accepts NULL / a smaller
 * buffer: K is then not split -- another summation order of the same exact products. */
size_t sa_gemm_workspace_bytes(int M, int N, int K);
int sa_gemm_is_split_bf16(int M, int N, int K);
ctcStatus_t sa_gemm_f32(int trans_a, int trans_b, int M, int N, int K, float alpha, const float* A, long lda,
                        const float* B, long ldb, float beta, float* C, long ldc, const float* bias /* or NULL */,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Conv2d(in_c, out_c, (kh, kw), stride (s, s), padding 0) + ReLU, model.py:19-29,61-62.
 * x (B, in_c, T, F) NCHW contiguous; w (out_c, in_c, kh, kw); y written with caller-given strides so the last conv
 * can emit the GRU-ready (B, T', out_c * F') layout of model.py:66-71 directly:
 *     y[b * ys_b + c * ys_c + t * ys_t + f].
 * workspace >= sa_conv2d_fwd_workspace_bytes() holds the im2col matrix; if keep_cols != NULL the im2col matrix
 * ((B*T'*F') x (in_c*kh*kw) floats) is written there instead, for sa_conv2d_relu_bwd(fwd_cols) to reuse. */
/* 1 when this shape runs the direct kernels (one input channel, <= 32 output channels, even kh * kw <= 224: the first
 * conv of every shipped config): no im2col matrix exists, keep_cols / fwd_cols are ignored and may be NULL. */
int sa_conv2d_is_direct(int in_c, int F, int out_c, int kh, int kw, int s);
size_t sa_conv2d_fwd_workspace_bytes(int B, int in_c, int T, int F, int out_c, int kh, int kw, int s);
ctcStatus_t sa_conv2d_relu_fwd(const float* x, const float* w, const float* bias, float* y, int B, int in_c, int T,
                               int F, int out_c, int kh, int kw, int s, long ys_b, long ys_c, long ys_t,
                               float* keep_cols /* or NULL */, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of the above: dy (same strides as y), y (to mask the ReLU) -> dw (+=0: overwritten), dbias, and dx
 * (NCHW, overwritten) unless dx == NULL.  workspace >= sa_conv2d_bwd_workspace_bytes().
 * The input gradient of a direct conv at stride 2 (kw a multiple of 4) runs as four stride-1 phases in one launch -- the
 * rows / columns of one parity against the taps of that parity -- instead of a transposed conv over a zero-stuffed dy;
 * the results are bit-identical (option "conv.dx_phases" = 0 selects the zero-stuffed form). */
size_t sa_conv2d_bwd_workspace_bytes(int B, int in_c, int T, int F, int out_c, int kh, int kw, int s);
ctcStatus_t sa_conv2d_relu_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw,
                               float* dbias, int B, int in_c, int T, int F, int out_c, int kh, int kw, int s,
                               long ys_b, long ys_c, long ys_t, const float* fwd_cols /* or NULL */, void* workspace,
                               size_t workspace_bytes, void* stream);

/* Dropout (config["dropout"] != 0, training mode): nn.Dropout(p) behind every conv ReLU (model.py:25-27) and
 * nn.GRU(dropout=p) between the layers of the stack (model.py:35-39); every shipped config trains with p = 0.2 .. 0.5
 * (examples/timit/ctc_config.json:20).  The masks are generated INSIDE the kernels that write / route the masked
 * elements: element `idx` of masked tensor `mask_stream` of a pass keyed by `seed` is kept iff word (idx & 3) of
 * Philox4x32-10(counter = {idx >> 2 (64 bit), mask_stream, 0}, key = seed) >= floor(p * 2^32), kept elements are
 * scaled by 1 / (1 - p).  0 <= p < 1; p == 0 is exactly the entry point without dropout.
 *   conv:  idx = the element's NCHW offset ((b O + c) T' + t') F' + f' (whatever strides y is written with);
 *   GRU :  layer l < L-1, mask_stream0 + l, idx = the element's offset in the (T, B, D*H) output.
 * sa_dropout_mask_f32 writes the factors (0 or 1 / (1 - p)) of elements idx0 .. idx0 + n - 1 (tests hand the oracle the
 * SAME mask); sa_dropout_apply_f32: out[i] = in[i] * factor(idx0 + i), in == out allowed. */
ctcStatus_t sa_dropout_mask_f32(float* d_out, size_t n, size_t idx0, float p, unsigned long long seed,
                                unsigned int mask_stream, void* stream);
ctcStatus_t sa_dropout_apply_f32(const float* d_in, float* d_out, size_t n, size_t idx0, float p,
                                 unsigned long long seed, unsigned int mask_stream, void* stream);
/* sa_conv2d_relu_fwd with the mask applied in the epilogue (y = the DROPPED output), and its backward pass: y must
 * be that dropped output -- [y > 0] is then "ReLU passed and kept", so the backward needs p only for the scale. */
ctcStatus_t sa_conv2d_relu_dropout_fwd(const float* x, const float* w, const float* bias, float* y, int B, int in_c,
                                       int T, int F, int out_c, int kh, int kw, int s, long ys_b, long ys_c, long ys_t,
                                       float* keep_cols /* or NULL */, void* workspace, size_t workspace_bytes, float p,
                                       unsigned long long seed, unsigned int mask_stream, void* stream);
ctcStatus_t sa_conv2d_relu_dropout_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx,
                                       float* dw, float* dbias, int B, int in_c, int T, int F, int out_c, int kh, int kw,
                                       int s, long ys_b, long ys_c, long ys_t, const float* fwd_cols /* or NULL */,
                                       void* workspace, size_t workspace_bytes, float p, void* stream);

/* One direction of one nn.GRU layer (model.py:35-39,73; equations SURVEY.md App. B; gate order [r; z; n]).
 *   ai   (B, T, 3H): input pre-activations x W_ih^T + b_ih (from sa_gemm_f32), batch-first.
 *   w_hh (3H, H), b_hh (3H).   h_out[b * hs_b + t * hs_t + j], j < H  (strides allow writing one half of a
 *   bidirectional (B, T, 2H) output).  reverse != 0 runs t = T-1 .. 0.
 *   stash (B, T, 5H) or NULL: r, z, n, q = W_hn h + b_hn, and h_{t-1}, saved for the backward pass.
 * All utterances run the full T steps (the reference passes the padded length for every utterance, ctc_model.py:43-45). */
ctcStatus_t sa_gru_fwd(const float* ai, const float* w_hh, const float* b_hh, float* h_out, long hs_b, long hs_t,
                       float* stash, int B, int T, int H, int reverse, void* stream);

/* Backward through time of sa_gru_fwd.
 *   dh_out: gradient wrt h_out (same strides);  outputs: dai (B,T,3H) = grad wrt ai, dah (B,T,3H) = grad wrt the
 *   h2h pre-activations (for dW_hh = dah^T h_prev and db_hh via sa_gemm_f32 / sa_colsum).  workspace holds the
 *   running dh (2 * B * H floats). */
size_t sa_gru_bwd_workspace_bytes(int B, int T, int H);
ctcStatus_t sa_gru_bwd(const float* dh_out, long hs_b, long hs_t, const float* h_out, const float* stash,
                       const float* w_hh, float* dai, float* dah, int B, int T, int H, int reverse, void* workspace,
                       size_t workspace_bytes, void* stream);

/* The whole GRU stack of Model.encode (model.py:35-39,73), TIME-MAJOR arrays: x (T, B, I0); h_out[l] (T, B, D*H);
 * stash[l*D+d] (T, B, 5H) or stash == NULL (inference).  Parameter arrays hold L*D device pointers, index l*D + d
 * (d = 1: the reverse direction), in nn.GRU's layouts.  Includes the input projections (MFMA GEMMs).
 * Unidirectional stacks run as a chunked layer wavefront (`chunk` time steps per chunk, <= 0: default): layer l
 * processes chunk c while layer l+1 processes chunk c-1, up to L layer-steps per launch.  Bidirectional stacks run
 * layer by layer with both directions sharing each launch.  L <= 8.
 * aux_streams (n_aux >= 0 extra hipStream_t handles, may be NULL): a step launch is a latency chain that leaves the
 * chip idle, so the batch is cut into 1 + n_aux independent groups of rows whose step kernels run concurrently, one
 * group per stream; the groups rejoin `stream` (event wait) once per wavefront wave and at the end of the call.
 * Arithmetic of the recurrence (fp32 storage and accumulation everywhere):
 *   - on a 256-CU device, H in {128, 256, 384, 512} and (layers or directions) x ceil(B / 16) groups that fit its 8 XCDs: the
 *     persistent kernels on producer-written bf16 planes (csrc/gru.hip, gru_fwd_planes_kernel / gru_fwd_chunk_planes_kernel):
 *     h is split exactly into three bf16 pieces by the thread that computes it, W_hh / W_ih once per launch, and six of the nine
 *     piece products run on the bf16 MFMA -- the split-bf16 arithmetic of the GEMMs above (dropped terms < 2^-26 of a product);
 *     sigmoid / tanh on v_exp_f32 / v_rcp_f32 (relative error < 3e-7).  Against an fp64 evaluation its error is the error of the
 *     exact-fp32 kernels (tests/test_gpu_blocks.py::test_planes_forward_error_budget); deterministic run to run;
 *   - option "gru.fwd_planes" = 0, or any other shape: the f32-input MFMA kernels (exact fp32 products, expf / tanhf), whose
 *     recurrence is bit-identical to one launch per time step.
 * Which of the two a call runs is a function of its shape, the device and that option only. */
size_t sa_gru_stack_fwd_workspace_bytes(int L, int D, int B, int T, int H, int I0);
ctcStatus_t sa_gru_stack_fwd(const float* x, int I0, const float* const* w_ih, const float* const* b_ih,
                             const float* const* w_hh, const float* const* b_hh, float* const* h_out,
                             float* const* stash, int L, int D, int B, int T, int H, int chunk, void* workspace,
                             size_t workspace_bytes, void* stream, void* const* aux_streams, int n_aux);

/* Backward through the stack.  dh_top (T, B, D*H): gradient wrt the top layer's output.  Fills dai / dah [l*D+d]
 * (T, B, 3H) (gradients wrt the i2h / h2h pre-activations of every layer and direction: the weight and bias gradients
 * are dai^T x_l, dah^T h_prev, column sums -- sa_gemm_f32 / sa_colsum_f32) and dx (T, B, I0), the gradient wrt the
 * stack input (may be NULL). */
size_t sa_gru_stack_bwd_workspace_bytes(int L, int D, int B, int T, int H, int I0);
ctcStatus_t sa_gru_stack_bwd(const float* dh_top, const float* const* stash, const float* const* w_ih,
                             const float* const* w_hh, float* const* dai, float* const* dah, float* dx, int I0, int L,
                             int D, int B, int T, int H, int chunk, void* workspace, size_t workspace_bytes,
                             void* stream, void* const* aux_streams, int n_aux);
/* The same backward pass, plus the stack's PARAMETER gradients (what autograd derives for nn.GRU, train.py:30):
 *   dw_ih[l*D+d] (3H, I_l) = dai^T in_l (in_0 = x (T, B, I0), in_l = h_out[l-1] (T, B, D*H)),  db_ih = column sums of dai,
 *   dw_hh[l*D+d] (3H, H)   = dah^T h_prev (the stash's fifth block),                            db_hh = column sums of dah.
 * The library schedules them (csrc/gru.hip, WGradIssuer).  Unidirectional stacks: on `stream` behind the recurrence, on
 * split-bf16 operands packed once per layer -- with the one-launch backward kernel (B a multiple of 16, H a multiple of
 * 128) the kernel writes the packed gate gradients and the bias sums itself.  Bidirectional stacks: a finished layer's
 * products go to a library-owned side stream as XCD-filtered launches that run BESIDE the next layer's recurrence, on the
 * XCDs it leaves idle; the call joins the side stream before it returns (stream-ordered:
 * everything is complete when `stream` reaches the end of the call's work).  Deterministic: fixed accumulation order.
 * dai / dah are SCRATCH for these entry points: when the recurrence kernel packs the gate gradients itself it writes no
 * row-major copy (except dai of the bottom layer of a unidirectional stack, which the d x product reads); callers that
 * want dai / dah use sa_gru_stack_bwd. */
ctcStatus_t sa_gru_stack_bwd_wgrad(const float* dh_top, const float* const* stash, const float* const* w_ih,
                                   const float* const* w_hh, float* const* dai, float* const* dah, float* dx, int I0,
                                   int L, int D, int B, int T, int H, int chunk, const float* x,
                                   const float* const* h_out, float* const* dw_ih, float* const* dw_hh,
                                   float* const* db_ih, float* const* db_hh, void* workspace, size_t workspace_bytes,
                                   void* stream);

/* The stack with nn.GRU's inter-layer dropout (model.py:38; see "Dropout" above).  h_drop[l], l < L-1: caller-owned
 * (T, B, D*H) buffers that receive h_out[l] * mask -- what layer l+1 reads, and what its dW_ih product reads in the
 * backward pass (h_out[l] itself, undropped, stays the layer's own recurrence state).  The one-launch fused kernels of
 * eligible unidirectional stacks write / apply the mask themselves (the stack stays ONE launch per direction of time);
 * other paths add one element-wise launch per layer (bidirectional) or per wavefront chunk.  The backward entry point
 * takes the same p / seed / mask_stream0 and the h_drop buffers the forward call filled. */
ctcStatus_t sa_gru_stack_fwd_dropout(const float* x, int I0, const float* const* w_ih, const float* const* b_ih,
                                     const float* const* w_hh, const float* const* b_hh, float* const* h_out,
                                     float* const* h_drop, float* const* stash, int L, int D, int B, int T, int H,
                                     int chunk, void* workspace, size_t workspace_bytes, float p,
                                     unsigned long long seed, unsigned int mask_stream0, void* stream);
ctcStatus_t sa_gru_stack_bwd_wgrad_dropout(const float* dh_top, const float* const* stash, const float* const* w_ih,
                                           const float* const* w_hh, float* const* dai, float* const* dah, float* dx,
                                           int I0, int L, int D, int B, int T, int H, int chunk, const float* x,
                                           const float* const* h_out, float* const* h_drop, float* const* dw_ih,
                                           float* const* dw_hh, float* const* db_ih, float* const* db_hh,
                                           void* workspace, size_t workspace_bytes, float p, unsigned long long seed,
                                           unsigned int mask_stream0, void* stream);

/* ---- run-time options ------------------------------------------------------------------------------------------------
 * The library never reads the process environment (rounds 1-4 did, on every call: VERDICT r04).  What the parity tests and the
 * measurement tools need to steer -- which CTC domain runs, whether the persistent recurrence kernels are used, fault
 * injection, in-kernel phase clocks ... -- is a table of named integer options: process-wide atomics, read by an entry point
 * when it is called (setting one from another thread never tears a value; a call already running keeps what it read).  Set an
 * option BEFORE the calls it should affect; -1 means "the library's own rule decides".  Names and meanings:
 * sa_option_name(i) / sa_option_help(i) for i in [0, sa_option_count()).  Every option's default is the measured-best path; a
 * caller that sets nothing gets exactly what bench.py measures.  (speech_amd/_lib.py applies SA_<NAME> environment variables
 * when it loads the library: SA_GRU_FUSED=0 sets "gru.fused" to 0 -- a convenience of the Python host, not of the ABI.)
 * Unknown name: CTC_STATUS_INVALID_VALUE. */
ctcStatus_t sa_set_option(const char* name, long value);
ctcStatus_t sa_get_option(const char* name, long* value);
void sa_reset_options(void);
int sa_option_count(void);
const char* sa_option_name(int i);
const char* sa_option_help(int i);
long sa_option_default(int i); /* the value sa_reset_options() gives option i (0 for an index out of range) */

/* Opt-in profiler of the CTC loss's serial part (bench.py's `ctc_chain_*` fields): after sa_ctc_profile_configure(1) every
 * latency-regime call (minibatch < 512: one workgroup per utterance) stamps the device-clock span of its alpha / beta
 * kernel -- earliest workgroup entry to latest workgroup exit -- into one of 256 slots; sa_ctc_profile_read(i, &us, &T, &B)
 * copies slot i back (SYNC).  Process-wide state, one device.  Returns 0 / -1. */
int sa_ctc_profile_configure(int enable);
int sa_ctc_profile_count(void);
int sa_ctc_profile_read(int i, double* chain_us, int* T, int* B);

/* Opt-in launch profiler for the stack entry points (bench.py): after sa_gru_profile_configure(1), block 0 of every
 * step launch stamps the 100 MHz wall clock at entry and exit into a device ring (16 K launches).
 * sa_gru_profile_read(kind, &interval_us, &kernel_us) (kind 0 = forward, 1 = backward step kernel; SYNC) returns the
 * number of full-width launches averaged, their mean entry-to-entry interval and mean entry-to-exit time, and resets.
 * configure(0) switches it off (the default).  The library's only process-global state. */
void sa_gru_profile_configure(int enable);
int sa_gru_profile_read(int kind, float* avg_interval_us, float* avg_kernel_us);
/* time steps covered by each launch averaged by the last sa_gru_profile_read(kind): 1 for the step kernels, the chunk
 * length for the persistent chunk kernels (whose launches are separated by GEMMs: their avg_interval_us is 0). */
int sa_gru_profile_steps_per_launch(int kind);

/* Unidirectional stacks with H = 512 (32 unit tiles), layers x ceil(B/16) <= 8, on a 256-CU device run the recurrence
 * as PERSISTENT chunk kernels whose sync groups are XCD-local (one (layer, batch tile) group per XCD, hand-off through
 * that XCD's L2; option "gru.persist" = 0 switches them off).  A workgroup derives its
 * group from the XCC id it actually runs on, so a dispatcher that does not spread 32 workgroups per XCD -- or a
 * hand-off that times out -- cannot hang or silently corrupt: it ORs its code into the library's STICKY device error
 * word (nothing but sa_gru_persist_reset() clears it).  Three things hang off that word:
 *   - sa_gru_health_flag() writes 1.0f / 0.0f for word != 0 / == 0 to a device float, stream-ordered: the caller
 *     appends that float to its gradient message and hands it to sa_clip_sgd_step() as d_skip_flag, so the optimiser
 *     skips the update ON THE DEVICE (no host round trip between a failure and the update it must stop; summed by the
 *     data-parallel all-reduce, one rank's failure stops every rank's update);
 *   - the failing workgroup also ORs its code into a word of mapped host memory (no copy is queued: a healthy call costs
 *     the host nothing); every sa_gru_stack_* call looks at that word, and once one finds it non-zero the persistent
 *     path is off for the process (every stack call from then on runs the step kernels).  The
 *     calls themselves keep returning success -- data-parallel ranks must stay in lock-step, so the failure travels
 *     through the gate above, not through one rank's return code; forward-only users call sa_gru_persist_status();
 *   - sa_gru_persist_status() waits for the device to drain and returns the OR of the codes seen (0 = fine;
 *     1 = a hand-off timed out, 2 = more than 32 workgroups landed on one XCD, 4 = an XCD-filtered side-stream GEMM
 *     launch -- the weight gradients / input projections that run beside a bidirectional layer's recurrence on the XCDs
 *     it leaves idle -- did not draw all of its tiles because the dispatcher placed too few of its blocks there);
 *     sa_gru_persist_reset() does the same, then clears the device word and the host state (the path stays off: the
 *     caller re-runs the lost step on the step kernels).
 * Tests: option "gru.fault" = 1 makes one workgroup of every persistent launch leave before its first step,
 * "gru.spin_limit" = n shortens the hand-off timeout (default 2^20 polls). */
int sa_gru_persist_status(void);
int sa_gru_persist_reset(void);
ctcStatus_t sa_gru_health_flag(float* d_flag, void* stream);

/* out[n] (+)= sum_m a[m * lda + n]  -- bias gradients; two deterministic stages through `workspace`. */
size_t sa_colsum_workspace_bytes(int M, int N);
ctcStatus_t sa_colsum_f32(const float* a, long lda, int M, int N, float* out, int accumulate, void* workspace,
                          size_t workspace_bytes, void* stream);

/* Weight and bias gradient of a linear layer in one call (LinearND, model.py:118-133, under loss.backward(), train.py:30):
 * C (M, N) = A^T B for A stored (K, M) -- the gradient of the layer's output, rows x classes -- and B (K, N) the layer's
 * input; colsum (M) = the column sums of A.  Both are OVERWRITTEN.  With M <= 32 (the 29 classes of the CTC model) one pass
 * over the operands forms both (thin_tn_kernel<CS>); other shapes take the tiled kernel with its column-sum epilogue.
 * Deterministic (fixed summation order).  workspace: sa_gemm_workspace_bytes(M, N, K). */
ctcStatus_t sa_gemm_tn_colsum_f32(int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C,
                                  long ldc, float* colsum, void* workspace, size_t workspace_bytes, void* stream);

/* y[i] = a[i] + b[i] (bidirectional sum, model.py:75-77, and gradient fan-in); strided rows. */
ctcStatus_t sa_add_rows_f32(const float* a, long lda, const float* b, long ldb, float* y, long ldy, int rows,
                            int cols, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * 5. Optimiser step on a flat fp32 parameter / gradient buffer (train.py:32 clip_grad_norm(params, 200) and
 *    train.py:35,95-97 SGD(lr, momentum)).  d_norm_out receives the pre-clip global L2 norm (device float).
 *    grad_scale multiplies the gradient first (1/world_size after the RCCL all-reduce).
 *    d_skip_flag (or NULL): device float; non-zero = leave params / momentum untouched and write MINUS the norm to
 *    d_norm_out (the health gate of sa_gru_health_flag above).
 * ----------------------------------------------------------------------------------------------------------------*/
size_t sa_sgd_workspace_bytes(size_t n);
ctcStatus_t sa_clip_sgd_step(float* params, float* grads, float* momentum_buf /* or NULL */, size_t n, float lr,
                             float momentum, float max_norm, float grad_scale, float* d_norm_out,
                             const float* d_skip_flag /* or NULL */, void* workspace, size_t workspace_bytes,
                             void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * 6. Featuriser: the log power spectrogram of speech/loader.py:156-166
 *    (scipy.signal.spectrogram, periodic Hann window of nperseg samples, hop = nperseg - noverlap, no detrend,
 *    one-sided 'density' scaling, then log(x + eps)) and the per-bin normalisation (x - mean) / std of loader.py:67.
 *    d_audio: int16 samples on the device; d_dft: the (nperseg, 2 * (nperseg/2+1)) windowed DFT matrix filled once by
 *    sa_specgram_build_dft; d_mean / d_std: per-bin statistics or both NULL; out: (frames, nperseg/2+1) fp32.
 * ----------------------------------------------------------------------------------------------------------------*/
int sa_specgram_frames(int n, int nperseg, int hop);
ctcStatus_t sa_specgram_build_dft(float* d_dft, int nperseg, void* stream);
size_t sa_log_specgram_workspace_bytes(int n, int nperseg, int hop);
ctcStatus_t sa_log_specgram(const short* d_audio, int n, int sample_rate, int nperseg, int hop, const float* d_dft,
                            const float* d_mean, const float* d_std, float eps, float* out, void* workspace,
                            size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * 7. RNN-Transducer.  The reference takes the loss and the decoder from the un-vendored,
 *    un-pinned package github.com/awni/transducer (Makefile:11): transducer.functions.transducer.TransducerLoss
 *    (speech/models/transducer_model.py:10-11,50-51) and transducer.decoders.decode_static (:98).  Its C interface
 *    is not on disk, so the binding contract is the Python one; these are the entry points speech_amd.transducer
 *    (and the `transducer` import shim) call.
 *
 *    sa_transducer_loss: log_probs = the LOG-softmax lattice (B, max_T, max_U1, K) contiguous, as the model builds it
 *    (transducer_model.py:76-77); labels flat int32, U_b = label_lengths[b] <= max_U1 - 1, T_b = input_lengths[b];
 *    blank_label = K - 1 at the reference's call site (:32).  d_costs[b] = -log p(y_b | lattice_b); grads (same shape
 *    as log_probs, may be NULL) = d costs[b] / d log_probs (zero outside the utterance's (T_b, U_b + 1) window and for
 *    every class other than blank and the cell's label).  All pointers DEVICE; stream-ordered; no host sync.
 * ----------------------------------------------------------------------------------------------------------------*/
size_t sa_transducer_workspace_bytes(int max_T, int max_U1, int minibatch);
ctcStatus_t sa_transducer_loss(const float* log_probs, float* grads /* or NULL */, const int* d_flat_labels,
                               const int* d_label_lengths, const int* d_input_lengths, int alphabet_size, int minibatch,
                               int max_T, int max_U1, int blank_label, float* d_costs, void* workspace,
                               size_t workspace_bytes, void* stream);

/* Static beam search of transducer.decoders.decode_static (transducer_model.py:98) for every utterance of a batch:
 * utterance b searches frames [0, d_T[b]) and lattice rows [0, d_U1[b]) of log_probs (B, max_T, max_U1, K); the search
 * never re-runs the prediction network, looks for hypotheses of exactly d_U1[b] - 1 labels' worth of rows, merges equal
 * hypotheses by log-sum-exp and keeps `beam_size` (<= 16) per step by a stable descending sort; K <= 64.
 * d_out_labels (B, max_U1) / d_out_lens (B): the best hypothesis; d_out_scores (B, double): its log-probability
 * including the final blank.  All pointers DEVICE. */
size_t sa_transducer_decode_workspace_bytes(int max_T, int max_U1, int minibatch, int beam_size);
ctcStatus_t sa_transducer_decode_static(const float* log_probs, const int* d_T, const int* d_U1, int alphabet_size,
                                        int minibatch, int max_T, int max_U1, int beam_size, int blank_label,
                                        int* d_out_labels, int* d_out_lens, double* d_out_scores, void* workspace,
                                        size_t workspace_bytes, void* stream);

/* Prediction / joint network pieces of Transducer.decode (transducer_model.py:54-78); the products around them are
 * sa_gemm_f32, the prediction GRU is sa_gru_stack_*.
 *   embedding: out[i,:] = table[idx[i],:] (idx int64, as torch.LongTensor);  bwd: dtable[v,:] = sum_{idx[i]=v} dout[i,:]
 *   joint:     z[b,t,u,:] = relu(xa[b,t,:] + ya[b,u,:])  with xa = fc1(x) (B,T,H), ya = fc1(y) (B,U1,H)  (:72-74)
 *              bwd: dxa[b,t,:] = sum_u dz * [xa+ya>0],  dya[b,u,:] = sum_t dz * [xa+ya>0]
 *   log_softmax over the last axis of (rows, K)  (:76) and its gradient dx = dy - exp(y) * sum(dy). */
ctcStatus_t sa_embedding_fwd(const float* table, const long long* idx, float* out, int n, int E, void* stream);
ctcStatus_t sa_embedding_bwd(const float* dout, const long long* idx, float* dtable, int n, int E, int V, void* stream);
ctcStatus_t sa_joint_relu_fwd(const float* xa, const float* ya, float* z, int B, int T, int U1, int H, void* stream);
ctcStatus_t sa_joint_relu_bwd(const float* dz, const float* xa, const float* ya, float* dxa, float* dya, int B, int T,
                              int U1, int H, void* stream);
ctcStatus_t sa_log_softmax_fwd(const float* x, float* y, long rows, int K, void* stream);
ctcStatus_t sa_log_softmax_bwd(const float* dy, const float* y, float* dx, long rows, int K, void* stream);

/* The joint of Transducer.decode as ONE operator (transducer_model.py:72-76: relu(xa + ya) -> fc2 -> log_softmax) that
 * never materialises the (B, T, U1, H) joint tensor: the MFMA operand is built from xa / ya rows on the fly.
 *   fwd: logp[b,t,u,:] = log_softmax(W2 relu(xa[b,t,:] + ya[b,u,:]) + b2)          W2 (K, H), b2 (K), logp (B,T,U1,K)
 *   bwd: given glp = dLoss/dlogp (B,T,U1,K) and the saved logp: dxa (B,T,H), dya (B,U1,H), dW2 (K,H), db2 (K), all
 *        WRITTEN (not accumulated); partial sums are folded in a fixed order (deterministic).
 * Supported shapes: H % 64 == 0, H <= 512, K <= 32; sa_joint_fused_workspace_bytes returns 0 for any other shape (use
 * sa_joint_relu_* + sa_gemm_f32 + sa_log_softmax_*), otherwise the backward workspace size in bytes. */
size_t sa_joint_fused_workspace_bytes(int B, int T, int U1, int H, int K);
ctcStatus_t sa_joint_fused_fwd(const float* xa, const float* ya, const float* w2, const float* b2, float* logp, int B,
                               int T, int U1, int H, int K, void* stream);
ctcStatus_t sa_joint_fused_bwd(const float* glp, const float* logp, const float* xa, const float* ya, const float* w2,
                               float* dxa, float* dya, float* dw2, float* db2, int B, int T, int U1, int H, int K,
                               void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * 8. Seq2Seq attention decoder (speech/models/seq2seq.py).  Per output token the decoder runs an
 *    nn.GRUCell (:20-21,97), NNAttention (:331-360) and, over all tokens, a linear layer and a summed cross-entropy
 *    (:59-63).  The projections are sa_gemm_f32; these are the pieces in between.
 *
 *    grucell gates: gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh (both (B, 3H), gate order r|z|n as torch);
 *        h' = (1-z) n + z h,  r = sig(gi_r+gh_r), z = sig(gi_z+gh_z), n = tanh(gi_n + r gh_n).
 *        stash (B, 4H) = r|z|n|gh_n (may be NULL forward-only).  bwd: dgi, dgh (B, 3H), dh_prev = dh * z.
 *    attention: eh (B,T,H) encoder states, ox (B,H) decoder state, ax_prev (B,T) previous alignment or NULL (then the
 *        location term is absent, :343), conv_w (H,KS) / conv_b (H) = Conv1d(1,H,KS,padding=(KS-1)/2) (KS odd, <= 15),
 *        nn_w (H) / nn_b (1) = Linear(H,1), scale = log(T) if log_t else 1.
 *        score_t = scale * (nn_b + sum_h nn_w[h] relu(eh[t,h] + ox[h] + conv_b[h] + sum_k conv_w[h,k] ax_prev[t+k-(KS-1)/2]))
 *        ax = softmax_t(score), sx = sum_t ax[t] eh[t,:].
 *        bwd: given d_sx (B,H) and d_ax_next (B,T) or NULL: d_eh += , d_ox =, d_ax_prev =, and per-utterance
 *        parameter-gradient partials g_* (+=; the caller sums them over B once per batch).
 *    softmax_xent: loss_rows[i] = lse(logits_i) - logits_i[target_i]; dlogits = (softmax - onehot) * scale (or NULL).
 *    argmax_rows: first index of each row's maximum (int64 out).
 * ----------------------------------------------------------------------------------------------------------------*/
ctcStatus_t sa_grucell_gates_fwd(const float* gi, const float* gh, const float* h_prev, float* h_out, float* stash,
                                 int B, int H, void* stream);
ctcStatus_t sa_grucell_gates_bwd(const float* dh, const float* stash, const float* h_prev, float* dgi, float* dgh,
                                 float* dh_prev, int B, int H, void* stream);
size_t sa_attention_workspace_bytes(int B, int T, int H, int KS);
ctcStatus_t sa_attention_fwd(const float* eh, const float* ox, const float* ax_prev, const float* conv_w,
                             const float* conv_b, const float* nn_w, const float* nn_b, float scale, float* ax,
                             float* sx, int B, int T, int H, int KS, void* workspace, size_t workspace_bytes,
                             void* stream);
ctcStatus_t sa_attention_bwd(const float* eh, const float* ox, const float* ax_prev, const float* conv_w,
                             const float* conv_b, const float* nn_w, const float* nn_b, float scale, const float* ax,
                             const float* d_sx, const float* d_ax_next, float* d_eh, float* d_ox, float* d_ax_prev,
                             float* g_conv_w, float* g_conv_b, float* g_nn_w, float* g_nn_b, int B, int T, int H,
                             int KS, void* workspace, size_t workspace_bytes, void* stream);
/* The whole decoder loop (seq2seq.py:77-112) and its backward through the tokens, enqueued by ONE call each.
 *   params / grads: HOST arrays of 11 DEVICE pointers in the order embedding.weight (V,E), dec_rnn.weight_ih (3H,E),
 *   weight_hh (3H,H), bias_ih, bias_hh, attend.conv.weight (H,KS), attend.conv.bias, attend.nn.1.fc.weight (H),
 *   attend.nn.1.fc.bias (1), fc.fc.weight (K,H), fc.fc.bias (K).  E == H (the context is added to the embedding, :95).
 *   y (B,U) int64 labels with start / end tokens; token t consumes y[:, t] -- or, where the HOST flag sample[t] is set
 *   (t >= 1; NULL = teacher forcing), the argmax of token t-1's logits (scheduled sampling, :91-96).
 *   Time-major outputs / stashes for the backward: out (U-1,B,K) logits, IDX (U-1,B) int64 inputs actually used,
 *   IX (U-1,B,E), ST (U-1,B,4H), HX (U-1,B,H), AX (U-1,B,T) alignments, OIN (U-1,B,H) = state + context.
 *   bwd: d_out (U-1,B,K) -> d_eh (B,T,H) and every parameter gradient (grads[i] written, not accumulated).
 *   Kernel forms (option "s2s.kernels", bits; default 3; H <= 256): bit 0 -- the attention backward of a token as ONE launch in
 *   the MFMA accumulator layout (the softmax's sum over the utterance from the stashed context: OIN - HX) instead of the
 *   round-4 stages (two launches); bit 1 -- the forward's score network in that layout and the context on four workgroups per
 *   utterance (also in sa_attention_fwd, sa_s2s_decoder_step, sa_s2s_beam_search, sa_s2s_greedy_decode).  fp32 throughout:
 *   the location term is an exact fp32 FMA chain in tap order on v_mfma_f32_16x16x4_f32 in every form. */
size_t sa_s2s_decoder_workspace_bytes(int B, int T, int U1, int H, int E, int KS, int K);
ctcStatus_t sa_s2s_decoder_fwd(const float* eh, const long long* y, const unsigned char* sample,
                               const float* const* params, int B, int T, int U, int H, int E, int KS, int K, float scale,
                               float* out, long long* IDX, float* IX, float* ST, float* HX, float* AX, float* OIN,
                               void* workspace, size_t workspace_bytes, void* stream);
ctcStatus_t sa_s2s_decoder_bwd(const float* eh, const float* const* params, const float* d_out, const long long* IDX,
                               const float* IX, const float* ST, const float* HX, const float* AX, const float* OIN,
                               int B, int T, int U, int H, int E, int KS, int K, int V, float scale, float* d_eh,
                               float* const* grads, void* workspace, size_t workspace_bytes, void* stream);
/* One decoder token without the training stashes (Seq2Seq.decode_step, seq2seq.py:114-138) -- the greedy / beam-search
 * step.  idx (B) int64 tokens; hprev (B,H), ax_prev (B,T), sx_prev (B,H) the previous state, all three NULL for the
 * first token.  Writes the new state hx, ax, sx and the logits out (B,K).  Same arithmetic as one iteration of
 * sa_s2s_decoder_fwd; workspace: sa_s2s_decoder_workspace_bytes(B, T, 1, H, E, KS, K). */
ctcStatus_t sa_s2s_decoder_step(const float* eh, const long long* idx, const float* hprev, const float* ax_prev,
                                const float* sx_prev, const float* const* params, int B, int T, int H, int E, int KS,
                                int K, float scale, float* hx, float* ax, float* sx, float* out, void* workspace,
                                size_t workspace_bytes, void* stream);
/* Beam search of the attention decoder, entirely on the device (Seq2Seq.beam_search, seq2seq.py:180-227; BASELINE config 4
 * names beam = 8).  ONE utterance: eh (T,H) encoder states, shared by every hypothesis.  The live hypotheses are the rows
 * of one batched decoder step (the arithmetic of sa_s2s_decoder_step); between two steps ONE kernel does what the
 * reference does in python: log_softmax rows, candidate scores (DOUBLE sums of float32 log-probabilities, as python
 * float + float32), the stable descending sort's first beam_size entries in the reference's (hypothesis rank, class)
 * tie order, end-token candidates inside them -> `complete`, the first beam_size non-end candidates of the whole list ->
 * the next beam, the stopping rule (:216-223), and the survivors' states / next inputs gathered by parent.
 *   start_tok / end_tok: first / last label of the reference's collate (:182-183); end_tok < K, start_tok < V.
 *   beam_size <= 32, beam_size * K <= 8192.  max_len: search steps at most (:180).
 *   check_every > 0: the call synchronises `stream` every check_every tokens to read ONE word (has the search stopped?)
 *   and stops enqueueing when it has; 0: never synchronises -- all max_len tokens are enqueued and a finished search
 *   ignores the rest.
 *   Outputs (DEVICE): d_hyp (max_len + 1) int64 = the winning hypothesis incl. the start token (and the end token when
 *   it completed), d_len its length, d_score its score (double), d_info[2] = {search steps run, hypotheses completed}.
 *   params as for sa_s2s_decoder_fwd. */
size_t sa_s2s_beam_workspace_bytes(int T, int H, int E, int KS, int K, int beam_size, int max_len);
ctcStatus_t sa_s2s_beam_search(const float* eh, const float* const* params, int T, int H, int E, int KS, int K,
                               float scale, int start_tok, int end_tok, int beam_size, int max_len, int check_every,
                               long long* d_hyp, int* d_len, double* d_score, int* d_info, void* workspace,
                               size_t workspace_bytes, void* stream);
/* Greedy decode of a batch as ONE call (Seq2Seq.infer / infer_decode, seq2seq.py:140-178): per token the decoder step above
 * on ping-pong state buffers and one kernel that takes each row's arg-max (first maximum), appends it to d_tokens, feeds it
 * back, and stops the decode when EVERY row emitted end_tok in the same step (:155-156) or after max_len steps.
 *   d_tokens (B, max_len + 1) int64 DEVICE: column 0 holds the start tokens on entry; d_steps[0] (DEVICE int) = steps run --
 *   the reference's result is the first d_steps[0] + 1 columns.  check_every as for sa_s2s_beam_search. */
size_t sa_s2s_greedy_workspace_bytes(int B, int T, int H, int E, int KS, int K, int max_len);
ctcStatus_t sa_s2s_greedy_decode(const float* eh, const float* const* params, int B, int T, int H, int E, int KS, int K,
                                 float scale, int end_tok, int max_len, int check_every, long long* d_tokens, int* d_steps,
                                 void* workspace, size_t workspace_bytes, void* stream);
ctcStatus_t sa_softmax_xent(const float* logits, const long long* targets, float scale, float* loss_rows,
                            float* dlogits, long rows, int K, void* stream);
ctcStatus_t sa_argmax_rows(const float* x, long long* out, long rows, int K, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPEECH_AMD_H */
