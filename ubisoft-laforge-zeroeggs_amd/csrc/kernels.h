// Host-side launchers of the streaming (HBM-bound) kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// strided-batch view of a [R, C] row set: row r lives at p + (r / rpb) * bstride + (r % rpb) * C (rpb == 0: contiguous)
struct RowView {
  float* p;
  long rpb, bstride;
};
inline RowView rv(const float* p) { return RowView{(float*)p, 0, 0}; }
inline RowView rv(const float* p, long rpb, long bstride) { return RowView{(float*)p, rpb, bstride}; }
int k_layernorm_fwd_v(RowView y, RowView x, RowView res, const float* gamma, const float* beta, float* mean,
                      float* rstd, int R, int C, float eps, hipStream_t s);
int k_layernorm_bwd_v(RowView dx, RowView dy, RowView x, RowView res, const float* gamma, const float* mean,
                      const float* rstd, float* dgamma, float* dbeta, int R, int C, hipStream_t s);
int k_act_bwd_v(RowView dx, RowView dy, RowView y, long R, int C, int act, float y_scale, hipStream_t s);
// One pass over the rows of a backward chain around a LayerNorm (the style encoder's backward is four such chains):
//   dy   = (dyA + dyB) * mask(seed_pre)                     [dy_pool != null: the mean-pool backward dy_pool[row / pool_L] / pool_L]
//   dx   = LayerNorm backward of dy (gamma == null: dx = dy), dgamma / dbeta accumulated (atomics)
//   dx_raw (optional) = dx                                   (the residual branch's share)
//   out  = dx * mask(seed_post) * act'(ysave);  dbias[c] += sum_rows out   (atomics; the gradient of the bias that produced the
//          LayerNorm's input);  pad_L > 0: `out` is the interior of a [B, pad_L + 2, C] buffer whose edge rows are zeroed here
// Element index of the masks: row * C + c (k_dropout over the contiguous [R, C] array).  In-place use is fine row by row.
struct LnBwdFused {
  RowView dyA, dyB;
  const float* dy_pool;
  int pool_L;
  float p_pre, p_post;
  uint64_t seed_pre, seed_post;
  RowView x, res;
  const float *gamma, *mean, *rstd;
  RowView dx_raw, out, ysave;
  int act, pad_L;
  float *dgamma, *dbeta, *dbias;
  int R, C;
};
LnBwdFused ln_bwd_fused_args(int R, int C);
// ... and the forward twin: y = (LayerNorm(x * mask(seed_pre) + res) * gamma + beta) * mask(seed_post) + table[row % table_L]
// in one pass (the masked x is written back when p_pre > 0: the backward's LayerNorm input is the post-dropout tensor; with pre_bias
// x is act(x + pre_bias) before anything else, also written back);
// pad_L > 0: y is the interior of a [B, pad_L + 2, C] buffer whose edge rows are zeroed here.  C <= 512.
struct LnFwdFused {
  RowView x, res, y;
  const float* pre_bias;      // != null: x = act(x + pre_bias) first, written back (the bias / activation epilogue of the product that
  int pre_act;                //          made x, folded into this pass: that product can then split K freely)
  const float *gamma, *beta, *table;
  float *mean, *rstd;
  float p_pre, p_post, eps;
  uint64_t seed_pre, seed_post;
  int table_L, pad_L, R, C;
};
LnFwdFused ln_fwd_fused_args(int R, int C, float eps);
int k_ln_fwd_fused(const LnFwdFused& a, hipStream_t s);
int k_ln_bwd_fused(const LnBwdFused& a, hipStream_t s);
int k_colsum_v(float* out, RowView x, long R, int C, float beta, hipStream_t s);

// y = x (optional), fill
int k_fill(float* p, long n, float v, hipStream_t s);
int k_copy(float* dst, const float* src, long n, hipStream_t s);
int k_add_inplace(float* dst, const float* src, long n, hipStream_t s);            // dst += src
// dx = dy * act'(y) where y is the activation OUTPUT (times inv_keep if y went through dropout)
int k_act_bwd(float* dx, const float* dy, const float* y, long n, int act, float y_scale, hipStream_t s);
// x *= dropout_scale(seed, base+i, p)   (used for forward and for the backward mask)
int k_dropout(float* x, long n, float p, uint64_t seed, hipStream_t s);
// rows x cols views with a leading dimension (dropout on the interior of a padded buffer)
int k_dropout_rows(float* x, int rows, int cols, long ld, long batch_rows, long batch_stride, float p,
                   uint64_t seed, hipStream_t s);
// LayerNorm over the last dim C of [R, C]; x may have an added residual; saves mean/rstd
int k_layernorm_fwd(float* y, const float* x, const float* res, const float* gamma, const float* beta,
                    float* mean, float* rstd, int R, int C, float eps, hipStream_t s);
// dx (+)= LN backward; dgamma/dbeta accumulated with atomics (buffers must be zeroed by the caller)
int k_layernorm_bwd(float* dx, const float* dy, const float* x, const float* res, const float* gamma,
                    const float* mean, const float* rstd, float* dgamma, float* dbeta, int R, int C,
                    hipStream_t s);
// row softmax of [R, L]; optional dropout copy Pd = dropout(P)
int k_softmax_fwd(float* P, float* Pd, const float* S, long R, int L, float p, uint64_t seed, hipStream_t s);
// dS = P * (dP*mask - sum_j(dP_j*mask_j*P_j))
int k_softmax_bwd(float* dS, const float* dPd, const float* P, long R, int L, float p, uint64_t seed,
                  hipStream_t s);
// out[c] (+)= sum_r x[r*ld + c]
int k_colsum(float* out, const float* x, long R, int C, long ld, float beta, hipStream_t s);
// [B, T, C] -> [B, pl + T + pr, C]; mode 0 zero, 1 replicate
int k_pad_rows(float* dst, const float* src, int B, int T, int C, int pl, int pr, int mode, hipStream_t s);
// fill only the pad rows of an already-populated padded buffer (mode 0 zero / 1 replicate)
int k_pad_edges(float* buf, int B, int T, int C, int pl, int pr, int mode, hipStream_t s);
// backward of replicate padding: dx[b,t] = dpad[b,t+pl] (+ folded edge rows at t=0 / t=T-1)
int k_unpad_fold(float* dx, const float* dpad, int B, int T, int C, int pl, int pr, int mode, hipStream_t s);
// conv weight [Co, Ci, Kw] -> fwd-packed [(j,ci), co] and bwd-packed [(j',co), ci] (flipped taps)
int k_pack_conv_w(float* wf, float* wb, const float* w, int Co, int Ci, int Kw, hipStream_t s);
// up to four of them in ONE launch (the style encoder packs its four convolutions at the top of its forward: one dispatch on the
// serial chain instead of four)
struct PackConvW { float *wf, *wb; const float* w; int Co, Ci, Kw; };
int k_pack_conv_w_multi(const PackConvW* items, int n, hipStream_t s);
// dW[co,ci,j] = dWf[(j,ci), co]
int k_unpack_conv_dw(float* dw, const float* dwf, int Co, int Ci, int Kw, hipStream_t s);
// h[b,l,c] += table[l,c]
int k_add_rows_bcast(float* h, const float* table, int B, int L, int C, hipStream_t s);
// pooled[b,c] = sum_l f[b,l,c] / L ; bwd: df[b,l,c] = dpooled[b,c] / L
int k_meanpool_fwd(float* out, const float* f, int B, int L, int C, hipStream_t s);
int k_meanpool_bwd(float* df, const float* dout, int B, int L, int C, hipStream_t s);

// fused multi-head attention of the style encoder (attention.hip); head dimension 32 only (attn_fused_supported)
int attn_fused_supported(int E, int NH);
int k_attn_fwd(const float* qkv, float* O, float* lse, int B, int L, int E, int NH, float p, uint64_t seed, hipStream_t s);
// dbias != null: [3E] += column sums of dqkv (atomics onto a zeroed / accumulating bias gradient)
int k_attn_bwd(const float* qkv, const float* O, const float* lse, const float* dO, float* dqkv, float* dsum, float* dbias,
               int B, int L, int E, int NH, float p, uint64_t seed, hipStream_t s);

// style encoder "gru": forward-direction recurrence and BPTT on the decoder's stage kernels (decoder_fast.hip)
struct Arena;
struct SgFast {
  float *pw, *pb, *xf, *Hxf, *DIxf, *DHxf, *GT, *DHn;
  long xf_floats;
  int NB, nT5, nTH, KBH, KB3H;
};
int sg_fast_supported(int B, int H);
SgFast sg_fast_carve(int B, int H, int L, Arena& a);
int sg_fast_fwd(int B, int H, int L, const float* w_hh, const float* b_hh, const float* GI, float* Hs, const SgFast& f,
                hipStream_t s);
int sg_fast_bwd(int B, int H, int L, const float* w_hh, const float* Hs, float* DI, float* dhc, const SgFast& f, hipStream_t s);
