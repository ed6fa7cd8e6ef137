/* C ABI of libzeggs_hip.so -- the MI355X (gfx950) engine for the ZeroEGGS hot path.
 *
 * The reference (ubisoft/ubisoft-laforge-ZeroEGGS) has no FFI: its seam is the Python
 * nn.Module level.  Each entry point below replaces the ATen op sequence of one reference
 * interface (file:line relative to the reference repo) and is what a binding for that
 * interface would call (ctypes stub: INTEGRATION.md; shipped binding: zeggs/ops.py).
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all tensors contiguous row-major fp32 DEVICE
 *     memory unless stated; quaternions (w,x,y,z); int64 indices where stated.
 *   - the caller owns all memory.  The library never allocates or frees device memory and does
 *     not synchronise the device (one documented exception: the FIRST use of each persistent
 *     kernel on a process is validated with a stream synchronisation, never inside a capture);
 *     it only enqueues kernels on `stream` (a hipStream_t passed as void*), so every call is
 *     hipGraph-capturable.
 *   - no per-call state is global: what a call needs beyond its arguments travels in the
 *     ZeggsDecCall struct of the *_ex entry points (prepared workspaces, the stream of deferred
 *     weight-gradient GEMMs, the caller-owned status words).  zeggs_set_option holds process-wide
 *     TUNING switches only (kernel selection, measurement hooks); the only other process-wide
 *     facts are "this persistent kernel was validated / disabled on this process"
 *     (zeggs_persistent_state) and the thread-local last-error string.  Two engines with their own
 *     workspaces and streams can therefore interleave calls in one process.
 *   - scratch + activations saved for backward live in a caller-provided workspace whose
 *     size is returned by the matching *_workspace_bytes(); fwd and bwd of one module must be
 *     given the SAME workspace (bwd reads what fwd saved).
 *   - return 0 on success, -1 on error; zeggs_last_error() returns a thread-local message.
 *   - gradient outputs are OVERWRITTEN (not accumulated).
 */
#ifndef ZEGGS_HIP_H
#define ZEGGS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int zeggs_version(void);
const char* zeggs_last_error(void);
/* process-wide TUNING switches: "decoder_fast" 1 (default) = fragment-packed stage kernels, 0 = generic GEMM path;
 * "persistent" / "train_persistent" / "bwd_persistent" 0/1 = the three weight-stationary persistent kernels;
 * "persistent_spin" = bound of their device-side waits (polls; 0: the first unsatisfied wait gives up -- test hook);
 * "timing" 1 = HIP events on the caller's stream around the decoder's steady-state stage sweeps;
 * round 5: "gemm_direct" 0 / 1 (default) / 2 / 3 / 5 = the barrier-free, LDS-free stream-K form of the TN (weight-gradient)
 * products: off / on / 128x64 / 64x64 wave tiles / only the small and batch-reduce products, "gemm_direct_wgs",
 * "gemm_direct_depth" (4 / 6 / 8 k-pairs in flight), "gemm_direct_shield" 0 (default) / 1 / 2 = the variant that allocates the
 * whole register file of its SIMDs, so that no wave of another stream becomes resident beside it (2: big products only) -- what a
 * caller that runs several streams beside each other should pick (zeggs.engine.TrainEngine: 1, depth 8), "gemm_direct_reserve" n =
 * CUs that variant's grids leave out (for a collective's resident workgroups in data-parallel runs); "ln_bwd4" 0 / 1 = the 16-byte-lane LayerNorm-backward pass; "mel_exact_log" 1 = the literal log / pow chain of
 * data_pipeline.py:62-63 instead of the affine map;
 * round 6 (A/B switches and experiments, each measured in profiles/r06_*): "tp_dual" 0 (default) / 1 = the forward sweep as two 16-row
 * dependency chains in one launch; "gemm_split_bf16" 0 (default) / 6 / 9 = the TN products on the bf16 matrix cores through an fp32-exact
 * three-plane split; "attn_bwd_one_launch" 1 (default) / 0 = the attention backward's two passes as one grid / two launches; "loss_lds"
 * 1 (default) / 0 = the loss tree walk's level messages through LDS / through the tables; "wgrad_order"; "tp_prologue" 1 (default) / 0 = the
 * training rollout's prologue (frame 0, CellStateEncoder, conditioning columns, step-1 products) in five launches / in ten; "chain" =
 * refused (-1) unless the library was built with -DZEGGS_CHAIN (measurement builds) */
int zeggs_set_option(const char* name, int value);
/* elapsed ms of the last recorded stage sweep: which = 0 forward (T-1 steps x 3 launches), 1 backward; blocks on
 * the end event.  Measurement hook of bench.py (roofline figures); there is no reference counterpart. */
int zeggs_timing_ms(int which, float* ms);

/* ---------------------------------------------------------------- generic GEMM (tests / tools)
 * C(m,n) = act(alpha * sum_k A(m,k) B(k,n) + beta * C(m,n) + bias[n]), element strides, batched. */
int zeggs_gemm(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
               long sam, long sak, long sbk, long sbn, long scm, long scn, int nbatch, long bsA,
               long bsB, long bsC, float alpha, float beta, int act, void* stream);
/* batch-reduce form (the weight gradient of a batched convolution, ZEGGS/modules.py:346-420 through autograd):
 * C(m,n) = beta * C(m,n) + sum_b sum_k A_b(m,k) B_b(k,n), A_b = A + b kbsA, B_b = B + b kbsB; beta 0 or 1. */
/* a weight gradient and its bias gradient in one call: dW[N, K] (+)= dy[M, N]^T x[M, K], db[N] (+)= column sums of dy; beta = 1:
 * both accumulate -- then the sums come out of the product kernel's own operand fragments (GemmArgs.asum), else a separate launch.
 * (What the decoder's deferred weight-gradient products call; exported for tests/test_gpu_parity.py.) */
int zeggs_gemm_tn_bias(const float* dy, long lddy, const float* x, long ldx, float* dW, long lddw, int M_contract, int N, int K,
                       float beta, float* db, void* stream);
/* The direct (LDS-free, barrier-free) TN kernel behind zeggs_gemm_tn / zeggs_gemm_tn_bias issues its operand loads as inline asm with
 * hand-counted waits; every variant (wave tile 64x64 / 128x64, 4 / 6 / 8 operand pairs in flight, plain / "shield") is therefore
 * CHECKED on its first use per process against a float64 host sum (ragged 293 x 155 x 346 product with row sums; synchronises; never
 * inside a stream capture) and the kernel is disabled for the process -- the LDS-tiled stream-K kernel takes its products -- if the
 * check fails.  This entry point runs that check for one variant on demand: 1 = agrees, 0 = does not (tests, toolchain bumps). */
int zeggs_gemm_direct_selftest(int big, int depth, int shield);
/* Routing of the TN products launched BY THE CALLING THREAD from now on: the values of the options "gemm_direct", "gemm_direct_shield",
 * "gemm_direct_depth", "gemm_direct_reserve" for this thread's launches, -1 = the process-wide option (zeggs_set_option).  The way
 * a caller with preferences of its own (zeggs/engine.py: TrainEngine wants the shield variant for its three-queue tail) states them
 * without changing what other engines or plain callers in the process get: per call, like ZeggsDecCall, not per process. */
int zeggs_gemm_route(int direct, int shield, int depth, int reserve);
int zeggs_gemm_route_get(int* out4);      /* the EFFECTIVE values for the calling thread (route, else process-wide option); host only */
int zeggs_gemm_kbatch(const float* A, const float* B, float* C, int M, int N, int K, long sam, long sak, long sbk, long sbn,
                      long scm, long scn, int kbatch, long kbsA, long kbsB, float beta, void* stream);

/* ---------------------------------------------------------------- SpeechEncoder
 * replaces SpeechEncoder.forward, ZEGGS/modules.py:265-272 (+ autograd backward).
 * x [B,T,F] normalised audio features -> out [B,T,O]. dropout_p = 0.2 in training, 0 in eval. */
typedef struct {
  int B, T, F, H, O, KW;
  float dropout_p;
  uint64_t seed;
} ZeggsSpeechDims;
typedef struct {
  const float *w0, *b0; /* layer0 Conv1d(F->H,k=1): [H,F,1],[H] */
  const float *w1, *b1; /* layer1 Conv1d(H->O,k=KW,replicate): [O,H,KW],[O] */
  const float *w2, *b2; /* layer2 Linear(O->O) */
} ZeggsSpeechParams;
typedef struct {
  float *w0, *b0, *w1, *b1, *w2, *b2;
} ZeggsSpeechGrads;
size_t zeggs_speech_encoder_workspace_bytes(const ZeggsSpeechDims*);
int zeggs_speech_encoder_fwd(const ZeggsSpeechDims*, const ZeggsSpeechParams*, const float* x, float* out,
                             void* ws, size_t ws_bytes, void* stream);
int zeggs_speech_encoder_bwd(const ZeggsSpeechDims*, const ZeggsSpeechParams*, const float* x,
                             const float* out, const float* dout, const ZeggsSpeechGrads*, void* ws,
                             size_t ws_bytes, void* stream);
/* grads_zeroed != 0: every gradient output is zero on entry (see ZeggsDecCall.grads_zeroed) */
int zeggs_speech_encoder_bwd_ex(const ZeggsSpeechDims*, const ZeggsSpeechParams*, const float* x,
                                const float* out, const float* dout, const ZeggsSpeechGrads*, void* ws,
                                size_t ws_bytes, void* stream, int grads_zeroed);

/* ---------------------------------------------------------------- StyleEncoderAttn trunk
 * replaces StyleEncoderAttn.forward, ZEGGS/modules.py:391-420 (convs :359-384, FFTBlock :484-612).
 * x [B,L,C] normalised exemplar features, pos [>=L,E] sinusoidal table -> out [B,E] (E = 2*S with VAE). */
typedef struct {
  int B, L, C, H, E, NH;
  int dropout; /* 1: training-mode dropout (.2/.2/.1/.1/.1), 0: eval */
  uint64_t seed;
} ZeggsStyleDims;
typedef struct {
  const float *c0_w, *c0_b;   /* encoder.convs.0.conv  [H,C,3] */
  const float *ln0_g, *ln0_b; /* encoder.convs.2       [H] */
  const float *c4_w, *c4_b;   /* encoder.convs.4.conv  [E,H,3] */
  const float *ln1_g, *ln1_b; /* encoder.convs.6       [E] */
  const float *in_w, *in_b;   /* attention in_proj     [3E,E] */
  const float *out_w, *out_b; /* attention out_proj    [E,E] */
  const float *lna_g, *lna_b; /* attention.layer_norm */
  const float *ff0_w, *ff0_b; /* feed_forward.convs.0.conv [E,E,3] */
  const float *ff2_w, *ff2_b; /* feed_forward.convs.2.conv [E,E,3] */
  const float *lnf_g, *lnf_b; /* feed_forward.layer_norm */
} ZeggsStyleParams;
typedef struct {
  float *c0_w, *c0_b, *ln0_g, *ln0_b, *c4_w, *c4_b, *ln1_g, *ln1_b, *in_w, *in_b, *out_w, *out_b, *lna_g,
      *lna_b, *ff0_w, *ff0_b, *ff2_w, *ff2_b, *lnf_g, *lnf_b;
} ZeggsStyleGrads;
size_t zeggs_style_encoder_workspace_bytes(const ZeggsStyleDims*);
int zeggs_style_encoder_fwd(const ZeggsStyleDims*, const ZeggsStyleParams*, const float* x, const float* pos,
                            float* out, void* ws, size_t ws_bytes, void* stream);
/* the same in two calls: part 1 = the head (weight packs, input padding, the first convolution -- one chip-filling product),
 * part 2 = the rest, part 3 = both.  For callers that run other streams beside the encoder and want to release them between the
 * two (zeggs/engine.py); same workspace, same stream, part 1 before part 2. */
int zeggs_style_encoder_fwd_part(const ZeggsStyleDims*, const ZeggsStyleParams*, const float* x, const float* pos, float* out,
                                 void* ws, size_t ws_bytes, void* stream, int part);
/* `part | 4`: the padded input of the first convolution ([B][L + 2][C], one zero row either side of every batch entry) already
 * lies in the workspace, at byte offset zeggs_style_encoder_input_offset() -- a caller that gathers the style example itself
 * (zeggs_gather_example; reference ZEGGS/dataset.py:176-204 + the normalisation of train.py:239) writes it there and the encoder's
 * own padding copy (2 x 56 MB at the head of the longest chain before the decoder sweep) is skipped; `x` is not read then. */
size_t zeggs_style_encoder_input_offset(const ZeggsStyleDims*);
int zeggs_style_encoder_bwd(const ZeggsStyleDims*, const ZeggsStyleParams*, const float* dout,
                            const ZeggsStyleGrads*, void* ws, size_t ws_bytes, void* stream);
int zeggs_style_encoder_bwd_ex(const ZeggsStyleDims*, const ZeggsStyleParams*, const float* dout,
                               const ZeggsStyleGrads*, void* ws, size_t ws_bytes, void* stream, int grads_zeroed);
/* part 1 = the chain of the backward (everything but the six weight-gradient products), 2 = those products (on `stream`; they read what
 * part 1 left in the workspace, not `dout`), 3 = both interleaved (= zeggs_style_encoder_bwd_ex).  For callers with a second queue: part 1 on
 * the caller's stream, part 2 on the other one behind it, join before the optimizer (round 6; zeggs/engine.py: defer_style_wgrads). */
int zeggs_style_encoder_bwd_part(const ZeggsStyleDims*, const ZeggsStyleParams*, const float* dout, const ZeggsStyleGrads*, void* ws,
                                 size_t ws_bytes, void* stream, int grads_zeroed, int part);
/* style_encoder.type "gru": replaces StyleEncoderGRU.forward, ZEGGS/modules.py:307-343 (conv3+ReLU x2, one
 * bidirectional GRU layer, projection of the last time step) and its backward.  x [B,L,C] -> out [B,O]. */
typedef struct {
  int B, L, C, H, O;
} ZeggsStyleGruDims;
typedef struct {
  const float *c0_w, *c0_b;                   /* encoder.convs.0.conv [H,C,3] */
  const float *c2_w, *c2_b;                   /* encoder.convs.2.conv [H,H,3] */
  const float *w_ih, *w_hh, *b_ih, *b_hh;     /* encoder.rnn_layer *_l0          [3H,H] (gates r,z,n) */
  const float *w_ih_r, *w_hh_r, *b_ih_r, *b_hh_r; /* encoder.rnn_layer *_l0_reverse */
  const float *p_w, *p_b;                     /* encoder.projection_layer.linear_layer [O,2H] */
} ZeggsStyleGruParams;
typedef struct {
  float *c0_w, *c0_b, *c2_w, *c2_b, *w_ih, *w_hh, *b_ih, *b_hh, *w_ih_r, *w_hh_r, *b_ih_r, *b_hh_r, *p_w, *p_b;
} ZeggsStyleGruGrads;
size_t zeggs_style_encoder_gru_workspace_bytes(const ZeggsStyleGruDims*);
int zeggs_style_encoder_gru_fwd(const ZeggsStyleGruDims*, const ZeggsStyleGruParams*, const float* x, float* out,
                                void* ws, size_t ws_bytes, void* stream);
int zeggs_style_encoder_gru_bwd(const ZeggsStyleGruDims*, const ZeggsStyleGruParams*, const float* dout,
                                const ZeggsStyleGruGrads*, void* ws, size_t ws_bytes, void* stream);
/* VAE re-parameterisation, StyleEncoder.forward ZEGGS/modules.py:291-302:
 * enc [B,2S] -> z = mu + eps*exp(.5 logvar)/temperature ; bwd gives denc from dz, dmu, dlogvar */
int zeggs_vae_reparam_fwd(const float* enc, const float* eps, float* z, float* mu_out /* [B,S] or NULL */,
                          float* logvar_out /* [B,S] or NULL */, int B, int S, float temperature, void* stream);
int zeggs_vae_reparam_bwd(const float* enc, const float* eps, const float* dz, const float* dmu,
                          const float* dlogvar, float* denc, int B, int S, float temperature, void* stream);

/* ---------------------------------------------------------------- Decoder
 * replaces Decoder.forward, ZEGGS/modules.py:47-162 (CellStateEncoder :230-243, RecurrentDecoderNormal
 * :165-185, vectorize_input :677-713, devectorize_output :716-742) and its autograd backward (BPTT).
 * Pose rows are the reference's output vector layout [root_vel3, root_vrt3, lpos 3J, ltxy 6J, lvel 3J,
 * lvrt 3J] (PO = 6+15J), de-normalised.  Frame 0 of every output is the given first pose. */
typedef struct {
  int B, T, PI, PO, SP, ST, H;
  float dt;
  int film; /* 0: RecurrentDecoderNormal (rnn_cond "normal"); 1: RecurrentDecoderFiLM, ZEGGS/modules.py:188-227 */
} ZeggsDecDims;
/* rnn_cond "film": the style leaves the step input (x = [pose, speech], XS = PI+SP) and modulates the two hidden
 * layers instead: hid = ELU(layer0 x) * (1 + gamma[:H]) + beta[:H]; h2 = ELU(layer2 h1) * (1 + gamma[H:]) + beta[H:];
 * y = layer3 h2, with gamma = gammas_predictor(style), beta = betas_predictor(style). */
typedef struct {
  const float *l0_w, *l0_b;                     /* recurrent_decoder.layer0 [H, PI+SP+ST]          (film: [H, PI+SP]) */
  const float *w_ih0, *w_hh0, *b_ih0, *b_hh0;   /* layer1 GRU l0: [3H, H+PI+SP+ST], [3H,H] (gates r,z,n)  (film: [3H, H+PI+SP]) */
  const float *w_ih1, *w_hh1, *b_ih1, *b_hh1;   /* layer1 GRU l1: [3H,H],[3H,H] */
  const float *l2_w, *l2_b;                     /* layer2 [PO,H]                                   (film: [H,H]) */
  const float *c0_w, *c0_b, *c1_w, *c1_b, *c2_w, *c2_b; /* cell_state_encoder [H,PI+ST],[H,H],[2H,H] */
  const float *l3_w, *l3_b;                     /* film only: layer3 [PO,H] */
  const float *g_w, *g_b, *be_w, *be_b;         /* film only: gammas_predictor / betas_predictor [2H,ST],[2H] */
} ZeggsDecParams;
typedef struct {
  float *l0_w, *l0_b, *w_ih0, *w_hh0, *b_ih0, *b_hh0, *w_ih1, *w_hh1, *b_ih1, *b_hh1, *l2_w, *l2_b, *c0_w,
      *c0_b, *c1_w, *c1_b, *c2_w, *c2_b, *l3_w, *l3_b, *g_w, *g_b, *be_w, *be_b;
} ZeggsDecGrads;
typedef struct {
  const float *in_mean, *in_std;   /* [PI] */
  const float *out_mean, *out_std; /* [PO] */
} ZeggsDecStats;
size_t zeggs_decoder_workspace_bytes(const ZeggsDecDims*, int training);
/* pose0 [B,PO], rpos0 [B,3], rrot0 [B,4], gaze [B,T,3], speech [B,T,SP], style [B,T,ST]
 * -> pose [B,T,PO], rpos [B,T,3], rrot [B,T,4].  training=1 saves activations for bwd. */
int zeggs_decoder_fwd(const ZeggsDecDims*, const ZeggsDecParams*, const ZeggsDecStats*, const float* pose0,
                      const float* rpos0, const float* rrot0, const float* gaze, const float* speech,
                      const float* style, float* pose, float* rpos, float* rrot, int training, void* ws,
                      size_t ws_bytes, void* stream);
/* Chunked (streaming) decode, inference only: the same rollout resumed from a given state.  Frame 0 of the chunk is
 * the last frame already produced (pose0 / rpos0 / rrot0 = its outputs; index 0 of gaze / speech / style belongs to
 * it), h_in [2,B,H] = GRU state after that frame (NULL: first chunk -> CellStateEncoder as in zeggs_decoder_fwd),
 * h_out [2,B,H] (may be NULL) = state after the chunk's last frame.  Workspace: zeggs_decoder_workspace_bytes(d, 0). */
int zeggs_decoder_fwd_state(const ZeggsDecDims*, const ZeggsDecParams*, const ZeggsDecStats*, const float* pose0,
                            const float* rpos0, const float* rrot0, const float* gaze, const float* speech,
                            const float* style, float* pose, float* rpos, float* rrot, const float* h_in,
                            float* h_out, void* ws, size_t ws_bytes, void* stream);
/* Inference rollouts with B = 1 (generate.py's regime) run as ONE persistent launch by default (option "persistent",
 * csrc/decode_persistent.hip): every CU keeps its slice of the weights in registers for the whole rollout and the CUs
 * exchange the 4-9 KB step vectors as data-tagged granules; the first use on a process is validated (bounded waits,
 * error word, automatic fall-back to the stage launches).  zeggs_set_option("persistent", 0) forces the stage launches.
 *
 * "chain" option (zeggs_set_option("chain", 1), stage-launch rollouts with B <= 2; measured SLOWER, kept for the record): consecutive stage launches alternate
 * between the caller's stream and a library-owned second stream and hand over through device-side arrival counters, so
 * each launch fetches its weights while its predecessor still runs.  Every device-side wait is bounded; this returns the
 * error word of the last rollout on `ws` (0 = all hand-offs completed).  Synchronises the device. */
int zeggs_persistent_state(int which /* 0: B=1 decode kernel, 1: training-forward kernel, 2: BPTT sweep */);   /* 1 ok, 0 disabled, -1 unused */
/* measurement builds (-DZEGGS_TPTIME) only: phase stamps of the last 4 steps of the persistent training rollout */
int zeggs_tp_stamps(const ZeggsDecDims*, void* ws, size_t ws_bytes, unsigned long long* out /* host [4][2][32] */);
/* -DZEGGS_TPSTAT builds: 100 MHz ticks every workgroup spent polling for the hand-off into phase 1..3 of the rollout */
int zeggs_tp_waits(const ZeggsDecDims*, void* ws, size_t ws_bytes, unsigned long long* out /* host [256][4] */);
/* Batch 17..32: the rollout runs as TWO independent 16-row dependency chains in one launch by default (option "tp_dual",
 * csrc/train_dual.hip: one chain's epilogue and grid hand-off under the other chain's matrix products; same packs, operands and
 * canonical saves; replaces the same loop, ZEGGS/modules.py:100-151).  -DZEGGS_DCTIME builds: slot stamps of workgroups 0 / 255. */
int zeggs_tp_dual_stamps(const ZeggsDecDims*, void* ws, size_t ws_bytes, unsigned long long* out /* host [2][8][6][6] */);
/* Training backward with batch <= 64 (33..64: two sweeps of 32 rows): the BPTT sweep of zeggs_decoder_bwd (replaces the per-step autograd of
 * ZEGGS/train.py:425 through modules.py:100-151) runs as ONE persistent launch by default (option "bwd_persistent",
 * csrc/train_bwd_persistent.hip: transposed weights resident as 4-row v_mfma_f32_4x4x1 tiles, four grid hand-offs per step).
 * Same validation / fall-back protocol as the other persistent kernels.  -DZEGGS_BPTIME builds: phase stamps of steps 3..1 */
int zeggs_bp_stamps(const ZeggsDecDims*, void* ws, size_t ws_bytes, unsigned long long* out /* host [3][2][32] */);
/* -DZEGGS_BPSTAT builds: 100 MHz ticks every workgroup spent polling for the hand-off into phase 1..4, summed over the sweep */
int zeggs_bp_waits(const ZeggsDecDims*, void* ws, size_t ws_bytes, unsigned long long* out /* host [4][256][4]: polling, products-done -> arrived, P4 epilogue split of thread 0 */);
int zeggs_decoder_chain_errors(const ZeggsDecDims*, int training, void* ws, size_t ws_bytes, int* out);
/* measurement builds (-DZEGGS_CHTIME) only: 100 MHz wall-clock stamps of the phases of the last 16 chained launches */
int zeggs_decoder_chain_stamps(const ZeggsDecDims*, int training, void* ws, size_t ws_bytes,
                               unsigned long long* out /* host [16][2][16] */);
/* dpose [B,T,PO], drpos [B,T,3], drrot [B,T,4] (frame 0 ignored) -> parameter grads, dspeech [B,T,SP],
 * dstyle [B,T,ST] */
int zeggs_decoder_bwd(const ZeggsDecDims*, const ZeggsDecParams*, const ZeggsDecStats*, const float* gaze,
                      const float* pose, const float* rpos, const float* rrot, const float* dpose,
                      const float* drpos, const float* drrot, const ZeggsDecGrads*, float* dspeech,
                      float* dstyle, void* ws, size_t ws_bytes, void* stream);
/* Per-call controls of zeggs_decoder_fwd_ex / zeggs_decoder_bwd_ex (NULL or all-zero = the plain calls above).
 *   prepared      bit 0 (fwd) / bit 1 (bwd): zeggs_decoder_prepare has run on THIS workspace with THESE weights and returned
 *                 that bit -> the call skips the weight-only packs / the zero state it starts from.
 *   defer_wgrads  bwd only.  1: the weight-gradient GEMMs of the recurrent layers (layer0, the GRU, layer2 of
 *                 ZEGGS/modules.py:165-185: seven GEMMs with K = B (T-1) that read only what the sweep saved in `ws`) and their
 *                 bias sums are enqueued on `wgrad_stream` right after the sweep (ordered behind it by an event) and the call
 *                 returns WITHOUT joining: they run beside the CellStateEncoder backward and whatever the caller enqueues next
 *                 (the encoders' backward).  The caller makes every consumer of the decoder's weight gradients (all-reduce,
 *                 optimizer) wait for that stream and keeps `ws` alive until then (zeggs/ops.py, zeggs/engine.py).  The
 *                 reference has no counterpart: its autograd runs every backward op on one stream (ZEGGS/train.py:425).
 *                 2 (data-parallel runs): only the GEMMs of layer2 and GRU layer 1 go there -- with the bias sums and the
 *                 CellStateEncoder gradients that is the SECOND half of the decoder's parameters in module order; the caller
 *                 starts its all-reduce and lets zeggs_decoder_wgrads(what = 4) compute the first half underneath it.
 *   wgrad_stream  hipStream_t of the deferred GEMMs (required when defer_wgrads != 0; e.g. zeggs_side_stream).
 *   status        caller-owned DEVICE words [ZEGGS_STATUS_WORDS], zeroed by the caller once, or NULL.  A persistent kernel
 *                 whose bounded wait gives up after its validated first use (e.g. a co-tenant holds CUs) ORs its
 *                 ZEGGS_GAVE_UP_* bit into status[0] (sticky) and writes NaN into what its consumers read first;
 *                 zeggs_radam_step_guarded turns the optimizer step of such an iteration into a counted no-op (status[1]),
 *                 so nothing invalid reaches the weights before the host has looked (zeggs/engine.py re-runs the lost steps
 *                 on the stage kernels; zeggs/ops.py re-runs an inference rollout). */
#define ZEGGS_STATUS_WORDS 4
#define ZEGGS_GAVE_UP_DECODE 1u    /* decode_persistent_k   (B = 1 inference rollout) */
#define ZEGGS_GAVE_UP_TRAIN_FWD 2u /* train_fwd_persistent_k (training rollout) */
#define ZEGGS_GAVE_UP_BPTT 4u      /* train_bwd_persistent_k (BPTT sweep) */
typedef struct {
  int prepared;
  int defer_wgrads;
  void* wgrad_stream;
  unsigned* status;
  int grads_zeroed;     /* bwd: every gradient output is zero on entry (a loop that zeroes its flat gradient buffer once per step):
                           weight / bias gradients are accumulated by the split GEMMs / column sums, no zero-fill launch each */
} ZeggsDecCall;
int zeggs_decoder_fwd_ex(const ZeggsDecDims*, const ZeggsDecParams*, const ZeggsDecStats*, const float* pose0,
                         const float* rpos0, const float* rrot0, const float* gaze, const float* speech,
                         const float* style, float* pose, float* rpos, float* rrot, int training, void* ws,
                         size_t ws_bytes, void* stream, const ZeggsDecCall* call);
int zeggs_decoder_bwd_ex(const ZeggsDecDims*, const ZeggsDecParams*, const ZeggsDecStats*, const float* gaze,
                         const float* pose, const float* rpos, const float* rrot, const float* dpose,
                         const float* drpos, const float* drrot, const ZeggsDecGrads*, float* dspeech,
                         float* dstyle, void* ws, size_t ws_bytes, void* stream, const ZeggsDecCall* call);
/* zeggs_decoder_fwd_state + ZeggsDecCall (only `status` is used): a caller that feeds the returned state into its NEXT chunk
 * must be able to tell a rollout that gave up -- without a status word the only trace is NaN in the chunk's LAST frame, which
 * such a caller would carry forward for ever.  zeggs/stream.py owns one word per stream, looks at it after every chunk and
 * redoes the chunk on the stage launches.  (The plain entry points without a ZeggsDecCall -- zeggs_decoder_fwd,
 * zeggs_decoder_fwd_state, zeggs_decoder_bwd -- have nowhere to report a give-up of a validated persistent kernel: callers
 * that cannot rule out a co-tenant on the GPU use the *_ex forms or switch the persistent kernels off.) */
int zeggs_decoder_fwd_state_ex(const ZeggsDecDims*, const ZeggsDecParams*, const ZeggsDecStats*, const float* pose0,
                               const float* rpos0, const float* rrot0, const float* gaze, const float* speech,
                               const float* style, float* pose, float* rpos, float* rrot, const float* h_in,
                               float* h_out, void* ws, size_t ws_bytes, void* stream, const ZeggsDecCall* call);
/* second half of the deferred weight gradients (ZeggsDecCall.defer_wgrads = 2): what = 4 -> GRU layer 0 and layer0;
 * what | 8: the gradient outputs are zero on entry (as ZeggsDecCall.grads_zeroed of the backward it completes) */
int zeggs_decoder_wgrads(const ZeggsDecDims*, const ZeggsDecGrads*, void* ws, size_t ws_bytes, int what, void* stream);
/* The weight-only preparation of a training step on `ws` (merged / folded matrices and the fragment packs of the two
 * persistent sweeps; the reference has no counterpart, its GEMMs read nn.Parameter storage directly): may run on a second
 * stream beside the encoders' forward.  Returns the bit mask for ZeggsDecCall.prepared (1: forward packs ready, 2: backward
 * packs and zero state ready; 0: nothing done, these dimensions take another path); < 0: error. */
int zeggs_decoder_prepare(const ZeggsDecDims*, const ZeggsDecParams*, const ZeggsDecStats*, void* ws, size_t ws_bytes,
                          void* stream);
/* a low-priority stream per device, created on first use and never destroyed (a convenience for ZeggsDecCall.wgrad_stream;
 * any stream of the caller's will do) */
int zeggs_side_stream(void** out /* hipStream_t */);

/* ---------------------------------------------------------------- training loss
 * replaces the inline loss of ZEGGS/train.py:276-421 (xform_orthogonalize_from_xy, xform_fk_vel of
 * ZEGGS/anim/txform.py:10-34, 17 weighted L1 terms + KL, sum/18) and its backward.
 * O = prediction (pose/rpos/rrot as produced by zeggs_decoder_fwd), W = ground truth in the same layout.
 * terms[18] (weighted, as logged by the reference), loss = sum(terms)/18 in terms[18].
 * kl_weight <= 0 or mu == NULL disables the KL term.  Gradients are scaled by `gscale` (1/world_size). */
typedef struct {
  int B, T, J, S;
  float dt;
} ZeggsLossDims;
size_t zeggs_loss_workspace_bytes(const ZeggsLossDims*);
int zeggs_loss_fwd_bwd(const ZeggsLossDims*, const int* parents, const float* o_pose, const float* o_rpos,
                       const float* o_rrot, const float* w_pose, const float* w_rpos, const float* w_rrot,
                       const float* gaze, const float* mu, const float* logvar, float kl_weight, float* terms,
                       float* dpose, float* drpos, float* drrot, float* dmu, float* dlogvar, float gscale,
                       void* ws, size_t ws_bytes, void* stream);

/* The ground-truth half of the loss's feature pass (transposed truth rows + forward kinematics of the truth side) depends on
 * the batch only: zeggs_loss_prepare_truth runs it ahead of the step (any stream, e.g. with the batch prefetch) into the
 * workspace that the loss call of that batch is then given with truth_prepared = 1 (zeggs_loss_fwd_bwd = truth_prepared 0). */
int zeggs_loss_prepare_truth(const ZeggsLossDims*, const int* parents, const float* w_pose, const float* w_rpos,
                             const float* w_rrot, const float* gaze, void* ws, size_t ws_bytes, void* stream);
int zeggs_loss_fwd_bwd_ex(const ZeggsLossDims*, const int* parents, const float* o_pose, const float* o_rpos,
                          const float* o_rrot, const float* w_pose, const float* w_rpos, const float* w_rrot,
                          const float* gaze, const float* mu, const float* logvar, float kl_weight, float* terms,
                          float* dpose, float* drpos, float* drrot, float* dmu, float* dlogvar, float gscale,
                          void* ws, size_t ws_bytes, void* stream, int truth_prepared);

/* ---------------------------------------------------------------- RAdam
 * replaces RAdam.step, ZEGGS/optimizers.py:31-99, fused over one flat fp32 buffer.
 * rectified != 0: p -= step_scale * m / (sqrt(v) + eps), else p -= step_scale * m (host scalars of
 * optimizers.py:64-84 computed by the caller). */
int zeggs_radam_step(float* p, const float* g, float* m, float* v, long n, float beta1, float beta2, float eps,
                     float step_scale, int rectified, void* stream);
/* the same step, skipped on the DEVICE (p, m, v untouched, status[1] += 1) when status[0] != 0 (a persistent sweep of this
 * rank gave up: ZeggsDecCall.status) or gflag != NULL and gflag[0] != 0 (the flag of all ranks, summed by the gradient
 * all-reduce: zeggs_status_flag writes this rank's 0 / 1 into the float that travels with the gradients).
 * CALLER'S OBLIGATION (the flush protocol).  A skipped step is invisible in the weights -- they simply stay what they were -- so a
 * caller of the guarded step that READS THE WEIGHTS FOR KEEPS (a checkpoint, rendered samples, the end of training) must first
 * (1) synchronise the stream, (2) read status[1]: if it is > 0, that many optimizer steps did not happen; (3) switch the persistent
 * sweeps off (zeggs_set_option), zero the status words, rewind its step count by status[1] (the RAdam bias corrections) and re-run
 * those iterations -- same batches, same noise seeds -- before it saves anything; (4) optionally switch the sweeps back on after a
 * probation.  zeggs/engine.py: TrainEngine.flush() is exactly this (and _check_status() the lagged, stall-free version of it that
 * runs every iteration; re-arming: TrainEngine._maybe_rearm); zeggs.train() calls flush() before every checkpoint and at the end.
 * Nothing in the library can enforce it: the status words are the caller's, and so are the weights. */
int zeggs_radam_step_guarded(float* p, const float* g, float* m, float* v, long n, float beta1, float beta2, float eps,
                             float step_scale, int rectified, unsigned* status, const float* gflag, void* stream);
/* a step applied in PIECES (slices of the flat buffers, each as soon as its gradients are final -- zeggs/engine.py runs the
 * decoder's slice on the weight-gradient stream underneath the encoders' backward): every piece is guarded alike, exactly one of
 * them passes count_skip != 0 so that status[1] still counts skipped STEPS */
int zeggs_radam_step_guarded_part(float* p, const float* g, float* m, float* v, long n, float beta1, float beta2, float eps,
                                  float step_scale, int rectified, unsigned* status, const float* gflag, int count_skip,
                                  void* stream);
/* the general form: the step above with RAdam's weight_decay (ZEGGS/optimizers.py:88-95: p += -weight_decay * lr * p before the
 * update, in the two branches that apply a step) -- decay = weight_decay * lr, 0 when the step is not applied; status may be NULL
 * (unguarded).  The entry points above are this one with decay = 0. */
int zeggs_radam_step_wd(float* p, const float* g, float* m, float* v, long n, float beta1, float beta2, float eps,
                        float step_scale, int rectified, float decay, unsigned* status, const float* gflag, int count_skip,
                        void* stream);
int zeggs_status_flag(const unsigned* status, float* dst /* device float */, void* stream);
/* measurement / test hook (no reference counterpart): `workgroups` x `threads` resident for `ms` milliseconds of wall clock on
 * `stream`, touching `scratch[0..n)` lightly (may be NULL) -- the stand-in for a collective's resident workgroups that
 * tools/cotenant_probe.py puts beside the iteration's tail and tests/test_gpu_giveup.py across a sweep boundary */
int zeggs_test_cotenant(int workgroups, int threads, float ms, float* scratch, long n, void* stream);

/* ---------------------------------------------------------------- batch gather
 * replaces SGDataset.__getitem__/get_example + default collate, ZEGGS/dataset.py:110-204, reading
 * HBM-resident dataset arrays.  frames[f] rows of width `width`; window starts int64 [B]. */
int zeggs_gather_windows(const float* frames, int width, const int64_t* starts, int B, int T, float* out,
                         void* stream);
/* style example rows: src_rows int64 [B, L] (precomputed by the host rule of dataset.py:180-203) */
int zeggs_gather_rows(const float* frames, int width, const int64_t* rows, long nrows, float* out, int out_ld,
                      void* stream);

/* ---------------------------------------------------------------- audio front-end
 * replaces preprocess_audio, ZEGGS/data_pipeline.py:33-84 (mel + energy at the animation rate), i.e.
 * extract_mel_spectrogram_for_tts / extract_spectrogram / linear_to_mel / amplitude_to_db of
 * ZEGGS/audio/spectrograms.py:8-269 in float64, for the shipped conf (centered, real amplitude, Slaney
 * mel, range-normalised, no pre-emphasis).  Loudness normalisation (pyloudnorm) is applied by the caller.
 * wav float32 [n] (device) -> out float32 [n_frames, n_mels+1]; filterbank float64 [n_mels, n_fft/2+1]. */
typedef struct {
  int n_fft, hop, n_mels, fs;
  float fps, min_clip;
  double pre_emph;      /* audio_conf.pre_emphasis ? audio_conf.pre_emph_coeff : 0 (spectrograms.py:35; round 6: the struct grew by this
                           field and the next -- zeggs_version() >= 101) */
  int flags;            /* bit 0: audio_conf.centered is FALSE (spectrograms.py:237-239); bit 1: audio_conf.normalize_range is FALSE
                           (spectrograms.py:123-129); bits 2-3: audio_conf.resample_method (data_pipeline.py:65-79) -- 0 "linear",
                           4 "nearest", 8 "cubic" (not-a-knot spline over the whole signal: zeggs_mel_features only, the streaming
                           range call refuses it); 0 = the shipped configuration */
} ZeggsMelDims;
long zeggs_mel_stft_frames(const ZeggsMelDims*, long n_samples); /* integer rule of spectrograms.py:242-245 */
size_t zeggs_mel_workspace_bytes(const ZeggsMelDims*, long n_samples);
int zeggs_mel_features(const ZeggsMelDims*, const float* wav, long n_samples, const double* filterbank,
                       int n_frames, float* out, void* ws, size_t ws_bytes, void* stream);
/* streaming form: rows [k0, k1) of the same feature table from the samples received so far.  final == 0: the signal
 * continues, k1 must be <= zeggs_mel_frames_ready(d, n_samples) (no frame may need samples not yet received);
 * final != 0: n_samples is the whole signal -> identical to rows k0..k1-1 of zeggs_mel_features. */
long zeggs_mel_frames_ready(const ZeggsMelDims*, long n_samples);
size_t zeggs_mel_range_workspace_bytes(const ZeggsMelDims*, long k0, long k1);
int zeggs_mel_features_range(const ZeggsMelDims*, const float* wav, long n_samples, int final, const double* filterbank,
                             long k0, long k1, float* out, void* ws, size_t ws_bytes, void* stream);

/* Loudness normalisation pre-pass of preprocess_audio, ZEGGS/data_pipeline.py:34-39 (pyloudnorm 0.1.0: Meter(rate)
 * .integrated_loudness + normalize.loudness to `target` LUFS; algorithm restated in oracle/loudness.py), mono, on the device:
 * the two K-weighting biquads as chunk-parallel float64 recurrences, 400 ms gating-block energies, absolute / relative
 * gates, gain.  Host-prepared inputs: coef[10] (both biquads: b0 b1 b2 a1 a2), trans[8] (their `chunk`-sample homogeneous
 * state transitions), blk_lo / blk_hi [nblocks] (device; pyloudnorm's truncated block bounds).  f32_stages != 0 stores every
 * filter stage's output through float32, as pyloudnorm does on float32 input.  result (device): [0] LUFS, [1] gain;
 * gain32 (device float) feeds zeggs_scale_copy(dev_scale) to apply the gain without a host round trip. */
size_t zeggs_loudness_workspace_bytes(long n_samples, int nblocks, long chunk);
int zeggs_loudness_gain(const float* wav, long n_samples, int rate, double target, const double* coef, const double* trans,
                        long chunk, const long* blk_lo, const long* blk_hi, int nblocks, int f32_stages, double* result,
                        float* gain32, void* ws, size_t ws_bytes, void* stream);

/* the style example of a batch in one pass (replaces SGDataset.get_example, ZEGGS/dataset.py:176-204, + the input normalisation
 * of ZEGGS/train.py:239): out [B][pad + L + pad][out_width], out[b][pad + l][c] = ((c < width ? frames[rows[b L + l]][c] : 0) -
 * mean[c]) / stdv[c] (the gaze slot of the example is zero BEFORE the normalisation, dataset.py:194), the 2 pad edge rows of every
 * batch entry zero.  mean / stdv: [out_width].  With pad = 1 and `out` = workspace + zeggs_style_encoder_input_offset() this is the
 * padded input of the style encoder's first convolution. */
int zeggs_gather_example(const float* frames, int width, const int64_t* rows, int B, int L, const float* mean, const float* stdv,
                         float* out, int out_width, int pad, void* stream);
/* in-place feature normalisation (x - mean) / std of ZEGGS/train.py:232-234,239 ; stdv == NULL -> scalar std */
int zeggs_normalize_rows(float* x, long rows, int width, long ld, const float* mean, const float* stdv,
                         float std_scalar, void* stream);

/* ---------------------------------------------------------------- plumbing kernels of the training step
 * Elementwise work that the reference leaves to ATen inside train() (ZEGGS/train.py:215-432): zeroing the gradient
 * buffer (optimizer.zero_grad, :428), scaling by the upstream scalar of loss.backward() (:423), the VAE noise
 * eps = randn_like(std) (ZEGGS/modules.py:772-775; counter-hash stream, not torch's Philox), and
 * style_encoding.unsqueeze(1).repeat_interleave(T, 1) with its adjoint (ZEGGS/train.py:256). */
int zeggs_fill(float* dst, long n, float value, void* stream);
int zeggs_scale_copy(float* dst, const float* src, long n, const float* dev_scale /* device scalar or NULL */,
                     float alpha, void* stream);
int zeggs_randn(float* out, long n, uint64_t seed, void* stream);
/* x[i] *= mask(seed, i) / (1 - p): the counter-hash dropout mask every encoder kernel applies (F.dropout /
 * nn.Dropout of ZEGGS/modules.py:258-272, 361-388); element i depends on (seed, i) only, so forward and backward
 * regenerate the same mask without storing it */
int zeggs_dropout(float* x, long n, float p, uint64_t seed, void* stream);
int zeggs_broadcast_time(float* out /* [B,T,S] */, const float* z /* [B,S] */, int B, int T, int S, void* stream);
int zeggs_sum_time(float* dz /* [B,S] */, const float* dout /* [B,T,S] */, int B, int T, int S, void* stream);

/* ---------------------------------------------------------------- free functions of the Networks layer
 * The module-level functions ZEGGS/train.py:20-24 and ZEGGS/generate.py import next to the classes
 * (`from modules import compute_KL_div, normalize`) and the two pose (de)vectorisers Decoder.forward is built from, forward and
 * backward, so that the reference's OWN training loop (its inline ATen loss differentiates through them) runs against the
 * drop-in `modules` (INTEGRATION.md route 2).  Inside zeggs_decoder_* / zeggs_loss_* the same arithmetic is fused.
 *
 * normalize, ZEGGS/modules.py:673-675: y = x / (||x||_2 + eps) over the last dimension of [rows, width]. */
int zeggs_normalize_vec_fwd(const float* x, float* y, long rows, int width, float eps, void* stream);
int zeggs_normalize_vec_bwd(const float* x, const float* dy, float* dx, long rows, int width, float eps, void* stream);
/* vectorize_input, ZEGGS/modules.py:677-713: out [B, 9+15J] = (cat[root_vel, root_vrt, lpos, ltxy, lvel, lvrt,
 * quat_inv_mul_vec(root_rot, gaze_pos - root_pos)] - in_mean) / in_std; inputs [B,3] [B,4] [B,3] [B,3] [B,J,3] [B,J,2,3]
 * [B,J,3] [B,J,3] [B,3].  The backward writes every input gradient (shapes of the inputs). */
int zeggs_vectorize_input_fwd(int B, int J, const float* root_pos, const float* root_rot, const float* root_vel,
                              const float* root_vrt, const float* lpos, const float* ltxy, const float* lvel,
                              const float* lvrt, const float* gaze_pos, const float* in_mean, const float* in_std,
                              float* out, void* stream);
int zeggs_vectorize_input_bwd(int B, int J, const float* root_pos, const float* root_rot, const float* gaze_pos,
                              const float* in_std, const float* dout, float* d_root_pos, float* d_root_rot,
                              float* d_root_vel, float* d_root_vrt, float* d_lpos, float* d_ltxy, float* d_lvel,
                              float* d_lvrt, float* d_gaze_pos, void* stream);
/* devectorize_output, ZEGGS/modules.py:716-742: pose [B, 6+15J] = predicted * out_std + out_mean (its slices 0:3, 3:6, lpos,
 * ltxy, lvel, lvrt are the six pose outputs); new_root_pos = quat_mul_vec(root_rot, vel dt) + root_pos; new_root_rot =
 * quat_mul(quat_from_helical(quat_mul_vec(root_rot, vrt dt)), root_rot).  Backward: any of the three upstream gradients may be
 * NULL (= zero). */
int zeggs_devectorize_output_fwd(int B, int J, float dt, const float* predicted, const float* root_pos,
                                 const float* root_rot, const float* out_mean, const float* out_std, float* pose,
                                 float* new_root_pos, float* new_root_rot, void* stream);
int zeggs_devectorize_output_bwd(int B, int J, float dt, const float* predicted, const float* root_pos,
                                 const float* root_rot, const float* out_mean, const float* out_std, const float* d_pose,
                                 const float* d_new_root_pos, const float* d_new_root_rot, float* d_predicted,
                                 float* d_root_pos, float* d_root_rot, void* stream);
/* compute_KL_div, ZEGGS/modules.py:764-789 (the divergence; its annealing weight is host arithmetic):
 * out[0] = mean_b(-0.5 mean_s(1 + logvar - mu^2 - exp(logvar))), mu / logvar [B,S]; dout = device scalar. */
int zeggs_kl_div_fwd(const float* mu, const float* logvar, int B, int S, float* out, void* stream);
int zeggs_kl_div_bwd(const float* mu, const float* logvar, int B, int S, const float* dout, float* dmu, float* dlogvar,
                     void* stream);
/* get_mask_from_lengths, ZEGGS/modules.py:802-813: mask[b][i] = i < lengths[b] (bytes 0/1), lengths int64 [B] */
int zeggs_mask_from_lengths(const int64_t* lengths, int B, int max_len, uint8_t* mask, void* stream);

/* ---------------------------------------------------------------- animation pre-/post-processing (float64)
 * zeggs_anim_features replaces preprocess_animation, ZEGGS/data_pipeline.py:90-228 (with quat.py from_euler /
 * unroll / fk / between / to_helical / fk_vel): BVH channels of one clip -> the feature arrays consumed by
 * generate_gesture (ZEGGS/generate.py:229-248, 339-362) and by the dataset builder.
 *   euler_deg [N,J,3] channel order "zyx" (degrees), positions [N,J,3], parents int32 [J] (device, -1 = root);
 *   hips / spine2 / head = joint indices of "Hips" / "Spine2" / "Head"; N >= 4.
 * Outputs (device, caller-owned): float64 except the two-axis encodings ltxy / ctxy (float32, as the reference). */
typedef struct {
  int N, J, hips, spine2, head;
  double dt;
  int order;       /* channel order of euler_deg, packed: axis of channel i (0 x, 1 y, 2 z) in bits 2i..2i+1; 0 = "zyx" (round 6: any
                      order, as quat.from_euler, ZEGGS/anim/quat.py:154-163; the struct grew by this field: zeggs_version() >= 101) */
} ZeggsAnimDims;
typedef struct {
  double *root_pos, *root_rot, *root_vel, *root_vrt; /* [N,3] [N,4] [N,3] [N,3] */
  double *lpos, *lrot, *lvel, *lvrt;                 /* [N,J,3] [N,J,4] [N,J,3] [N,J,3] */
  float* ltxy;                                       /* [N,J,2,3] */
  double *cpos, *crot, *cvel, *cvrt;                 /* character space (root-relative FK) */
  float* ctxy;
  double *gaze_pos, *gaze_dir;                       /* [N,3] */
} ZeggsAnimOut;
size_t zeggs_anim_features_workspace_bytes(const ZeggsAnimDims*);
int zeggs_anim_features(const ZeggsAnimDims*, const int* parents, const double* euler_deg, const double* positions,
                        const ZeggsAnimOut* out, void* ws, size_t ws_bytes, void* stream);

/* decoder output -> BVH channels: replaces quat.from_xform(xform_orthogonalize_from_xy(V_ltxy)) of
 * ZEGGS/generate.py:389 and write_bvh, ZEGGS/utils.py:47-87 (root re-basing, root folded into joint 0,
 * quat.to_euler order "zyx" in degrees).  Inputs float32 [T,3] [T,4] [T,J,3] [T,J,2,3]; outputs float64 [T,J,3]. */
typedef struct {
  int T, J, rebase;            /* rebase != 0: start_position / start_rotation given */
  double start_pos[3], start_rot[4];
  int order;                   /* channel order of the euler output, packed as ZeggsAnimDims.order; 0 = "zyx"; "zyx" and "xzy" are what
                                  quat.to_euler (ZEGGS/anim/quat.py:111-127) implements (round 6; zeggs_version() >= 101) */
} ZeggsBvhDims;
int zeggs_pose_to_bvh(const ZeggsBvhDims*, const float* root_pos, const float* root_rot, const float* lpos,
                      const float* ltxy, double* positions, double* euler_deg, void* stream);

/* The same conversion for a CHUNK of a longer clip, written straight into the rows of the BVH motion block: table [T, 3 + 3J]
 * float64 = [root position | euler angles of joint seq[0], seq[1], ...] with seq int32 [J] (device) = the file's hierarchy
 * order (ZEGGS/anim/bvh.py save()); ref_root_pos [3] / ref_root_rot [4] (device, may be NULL = this chunk's frame 0) = frame 0
 * of the WHOLE clip, which the re-basing of utils.py:60-66 refers to.  generate_gesture() converts, downloads and formats a
 * chunk while the next one is being decoded. */
int zeggs_pose_to_bvh_table(const ZeggsBvhDims*, const float* root_pos, const float* root_rot, const float* lpos,
                            const float* ltxy, const float* ref_root_pos, const float* ref_root_rot, const int* seq,
                            double* table, void* stream);

/* HOST helper of write_bvh / bvh.save (ZEGGS/anim/bvh.py: one text row per frame, "%f" per channel, a space after every
 * number): appends (append != 0) or writes `rows` x `cols` HOST doubles to `path`.  No device work. */
int zeggs_write_table_text(const char* path, int append, const double* table /* host */, long rows, int cols);
/* HOST helper of bvh.load (ZEGGS/anim/bvh.py: the MOTION block): `text` [len] bytes, NUL-terminated behind them, holding `rows`
 * non-empty lines of `cols` whitespace-separated numbers each -> HOST doubles [rows, cols] (strtod: correctly rounded, as float()).
 * -1 when the text does not have that shape.  No device work. */
int zeggs_parse_table_text(const char* text, size_t len, double* table /* host */, long rows, int cols);
/* the formatting half alone, into the caller's buffer (single-threaded, re-entrant: called from several host threads on row
 * blocks, the caller writes the blocks in order); *written = bytes produced.  cap >= rows * (cols * 24 + 1) suffices. */
int zeggs_format_table_text(const double* table /* host */, long rows, int cols, char* out, size_t cap, size_t* written);

/* Launch sequences that are replayed as hipGraphs (the frame sweeps of the "gru" style encoder with option "sweep_graphs" = 1; default off):
 * how many were captured and how many replays hit the cache on this process.  No reference counterpart (diagnostics). */
int zeggs_sweep_graph_stats(long* captures, long* replays);

#ifdef __cplusplus
}
#endif
#endif
