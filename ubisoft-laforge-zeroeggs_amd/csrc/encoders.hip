// Speech encoder and style encoder (attention VAE trunk): forward + backward
// orchestration over the MFMA GEMM (gemm.hip) and the streaming kernels
// (kernels.hip).  Everything runs on the caller's stream; all scratch and saved
// activations live in the caller-provided workspace (same carve in fwd and bwd).
//
// Reference semantics: ZEGGS/modules.py:249-272 (SpeechEncoder), :346-420
// (StyleEncoderAttn), :484-612 (FFT block).  Convolutions are evaluated as
// GEMMs over padded [T+2p, C] buffers (row stride C, K = taps*C).
#include "../../include/zeggs_hip.h"
#include "common.h"
#include "gemm.h"
#include "kernels.h"

namespace {

// ------------------------------------------------------------------ speech
struct SpeechWs {
  float *h0p, *h1, *wf1, *wb1;       // saved: padded conv input (post-dropout), conv output (post-dropout)
  float *d2, *dh1pp, *dh0p, *dh0, *dwf1;  // backward scratch
};

SpeechWs carve_speech(const ZeggsSpeechDims& d, Arena& a) {
  SpeechWs w;
  const long BT = (long)d.B * d.T, pad = d.KW - 1;
  w.h0p = a.f((long)d.B * (d.T + pad) * d.H);
  w.h1 = a.f(BT * d.O);
  w.wf1 = a.f((long)d.KW * d.H * d.O);
  w.wb1 = a.f((long)d.KW * d.O * d.H);
  w.d2 = a.f(BT * d.O);
  w.dh1pp = a.f((long)d.B * (d.T + 2 * pad) * d.O);
  w.dh0p = a.f((long)d.B * (d.T + pad) * d.H);
  w.dh0 = a.f(BT * d.H);
  w.dwf1 = a.f((long)d.KW * d.H * d.O);
  return w;
}

// batched conv-as-GEMM: y[b][t][co] = act(sum_k xp_b[t*C + k] * Wf[k][co] + bias)   for t < M
int conv_gemm(const float* xp, long xp_bstride, int C, const float* Wf, int Ktot, int Co, float* y, long ldy,
              long y_bstride, const float* bias, int B, int M, int act, hipStream_t s) {
  GemmArgs g = gemm_args(xp, Wf, y, M, Co, Ktot);
  g.sam = C; g.sak = 1; g.sbk = Co; g.sbn = 1; g.scm = ldy; g.scn = 1;
  g.bsA0 = xp_bstride; g.bsB0 = 0; g.bsC0 = y_bstride; g.nb1 = 1;
  g.bias = bias; g.act = act;
  return launch_gemm(g, B, s);
}
// the same convolution over ALL batches as one product: xp [B][LP][C] read as a [B LP - (taps - 1), taps C] matrix with
// overlapping rows (lda = C), y row r = b LP + t (the rows t >= LP - taps + 1 of a batch mix two batches: never read); no bias /
// activation, so launch_gemm may split K (stream-K when the output has few tiles)
int conv_flat(const float* xp, int C, const float* Wf, int Ktot, int Co, float* y, int Mrows, hipStream_t s) {
  GemmArgs g = gemm_args(xp, Wf, y, Mrows, Co, Ktot);
  g.sam = C; g.sak = 1; g.sbk = Co; g.sbn = 1; g.scm = Co; g.scn = 1; g.nb1 = 1;
  return launch_gemm(g, 1, s);
}
// dWf[k][co] = sum_b sum_t xp_b[t*C + k] * dy_b[t][co]
int conv_dw_gemm(const float* xp, long xp_bstride, int C, const float* dy, long lddy, long dy_bstride, float* dWf,
                 int Ktot, int Co, int B, int T, hipStream_t s) {
  GemmArgs g = gemm_args(xp, dy, dWf, Ktot, Co, T);
  g.sam = 1; g.sak = C; g.sbk = lddy; g.sbn = 1; g.scm = Co; g.scn = 1;
  g.kbatch = B; g.kbsA = xp_bstride; g.kbsB = dy_bstride;
  return launch_gemm(g, 1, s);
}

}  // namespace

extern "C" size_t zeggs_speech_encoder_workspace_bytes(const ZeggsSpeechDims* d) {
  Arena a(nullptr, 0);
  carve_speech(*d, a);
  return a.off + 256;
}

extern "C" int zeggs_speech_encoder_fwd(const ZeggsSpeechDims* dp, const ZeggsSpeechParams* P, const float* x,
                                        float* out, void* ws, size_t ws_bytes, void* stream) {
  const ZeggsSpeechDims& d = *dp;
  hipStream_t s = (hipStream_t)stream;
  ZCHECK(d.KW % 2 == 1, "speech encoder: even kernel width %d", d.KW);
  Arena a(ws, ws_bytes);
  SpeechWs w = carve_speech(d, a);
  ZCHECK(a.ok(), "speech encoder: workspace too small (%zu < %zu)", ws_bytes, a.off);
  const int B = d.B, T = d.T, half = (d.KW - 1) / 2, TP = T + d.KW - 1;
  const long BT = (long)B * T;
  // layer0: 1x1 conv == Linear over features, written into the interior of the padded buffer
  {
    GemmArgs g = gemm_args(x, P->w0, w.h0p + (long)half * d.H, T, d.H, d.F);
    g.sam = d.F; g.sak = 1; g.sbk = 1; g.sbn = d.F; g.scm = d.H; g.scn = 1;
    g.bsA0 = (long)T * d.F; g.bsC0 = (long)TP * d.H; g.bias = P->b0; g.act = ACT_ELU;
    ZTRY(launch_gemm(g, B, s));
  }
  ZTRY(k_dropout_rows(w.h0p + (long)half * d.H, (int)BT, d.H, d.H, T, (long)TP * d.H, d.dropout_p, d.seed + 1, s));
  ZTRY(k_pad_edges(w.h0p, B, T, d.H, half, half, 1, s));
  // layer1: k=31 replicate-padded conv
  ZTRY(k_pack_conv_w(w.wf1, w.wb1, P->w1, d.O, d.H, d.KW, s));
  ZTRY(conv_gemm(w.h0p, (long)TP * d.H, d.H, w.wf1, d.KW * d.H, d.O, w.h1, d.O, (long)T * d.O, P->b1, B, T,
                 ACT_ELU, s));
  ZTRY(k_dropout(w.h1, BT * d.O, d.dropout_p, d.seed + 2, s));
  // layer2: Linear + ELU
  ZTRY(gemm_nt(w.h1, d.O, P->w2, d.O, out, d.O, P->b2, (int)BT, d.O, d.O, ACT_ELU, 0.f, s));
  return 0;
}

extern "C" int zeggs_speech_encoder_bwd(const ZeggsSpeechDims* dp, const ZeggsSpeechParams* P, const float* x,
                                        const float* out, const float* dout, const ZeggsSpeechGrads* G, void* ws,
                                        size_t ws_bytes, void* stream) {
  return zeggs_speech_encoder_bwd_ex(dp, P, x, out, dout, G, ws, ws_bytes, stream, 0);
}
// grads_zeroed != 0: the caller vouches that every gradient output is zero on entry (a training loop that zeroes its flat gradient
// buffer once per step): weight / bias gradients are then ACCUMULATED by the split GEMMs / column sums without a zero-fill launch each
extern "C" int zeggs_speech_encoder_bwd_ex(const ZeggsSpeechDims* dp, const ZeggsSpeechParams* P, const float* x,
                                           const float* out, const float* dout, const ZeggsSpeechGrads* G, void* ws,
                                           size_t ws_bytes, void* stream, int grads_zeroed) {
  const ZeggsSpeechDims& d = *dp;
  const float gb = grads_zeroed ? 1.f : 0.f;
  hipStream_t s = (hipStream_t)stream;
  Arena a(ws, ws_bytes);
  SpeechWs w = carve_speech(d, a);
  ZCHECK(a.ok(), "speech encoder bwd: workspace too small");
  const int B = d.B, T = d.T, half = (d.KW - 1) / 2, pad = d.KW - 1, TP = T + pad, TPP = T + 2 * pad;
  const long BT = (long)B * T;
  const float keep = 1.f - d.dropout_p;
  // layer2
  ZTRY(k_act_bwd(w.d2, dout, out, BT * d.O, ACT_ELU, 1.f, s));
  ZTRY(gemm_tn(w.d2, d.O, w.h1, d.O, G->w2, d.O, (int)BT, d.O, d.O, gb, s));
  ZTRY(k_colsum(G->b2, w.d2, BT, d.O, d.O, gb, s));
  // dh1 (into the interior of the doubly zero-padded buffer), through dropout and ELU
  float* dh1 = w.dh1pp + (long)pad * d.O;
  {
    GemmArgs g = gemm_args(w.d2, P->w2, dh1, T, d.O, d.O);
    g.sam = d.O; g.sak = 1; g.sbk = d.O; g.sbn = 1; g.scm = d.O; g.scn = 1;
    g.bsA0 = (long)T * d.O; g.bsC0 = (long)TPP * d.O;
    ZTRY(launch_gemm(g, B, s));
  }
  // mask (same seed/idx as forward), then ELU' using the post-dropout saved output
  ZTRY(k_dropout_rows(dh1, (int)BT, d.O, d.O, T, (long)TPP * d.O, d.dropout_p, d.seed + 2, s));
  {
    RowView v = rv(dh1, T, (long)TPP * d.O);
    ZTRY(k_act_bwd_v(v, v, rv(w.h1), BT, d.O, ACT_ELU, keep, s));
  }
  ZTRY(k_pad_edges(w.dh1pp, B, T, d.O, pad, pad, 0, s));
  // bias / weight grads of the conv
  ZTRY(k_colsum_v(G->b1, rv(dh1, T, (long)TPP * d.O), BT, d.O, gb, s));
  ZTRY(conv_dw_gemm(w.h0p, (long)TP * d.H, d.H, dh1, d.O, (long)TPP * d.O, w.dwf1, d.KW * d.H, d.O, B, T, s));
  ZTRY(k_unpack_conv_dw(G->w1, w.dwf1, d.O, d.H, d.KW, s));
  // input grad w.r.t. the padded conv input: correlation of zero-padded dh1 with flipped taps
  ZTRY(conv_gemm(w.dh1pp, (long)TPP * d.O, d.O, w.wb1, d.KW * d.O, d.H, w.dh0p, d.H, (long)TP * d.H, nullptr, B, TP,
                 ACT_NONE, s));
  ZTRY(k_unpad_fold(w.dh0, w.dh0p, B, T, d.H, half, half, 1, s));
  // through dropout0 and ELU0 (saved h0p interior is post-dropout)
  ZTRY(k_dropout(w.dh0, BT * d.H, d.dropout_p, d.seed + 1, s));
  ZTRY(k_act_bwd_v(rv(w.dh0), rv(w.dh0), rv(w.h0p + (long)half * d.H, T, (long)TP * d.H), BT, d.H, ACT_ELU, keep, s));
  ZTRY(gemm_tn(w.dh0, d.H, x, d.F, G->w0, d.F, (int)BT, d.H, d.F, gb, s));
  ZTRY(k_colsum(G->b0, w.dh0, BT, d.H, d.H, gb, s));
  return 0;
}

// =================================================================== style
namespace {

struct StyleWs {
  // packed conv weights
  float *wf0, *wb0, *wf4, *wb4, *wff0, *wfb0, *wff2, *wfb2;
  // saved activations
  float *xp, *c1, *m1, *r1, *a1p, *c2, *m2, *r2, *h, *qkv, *P, *Pd, *O, *ao, *ma, *ra, *ap, *f1p, *f2, *mf, *rf, *f;
  // scratch (backward)
  float *S, *t0, *t1, *t2, *t3, *dqkv, *dwf;
  float *t0a, *t0b, *t1a;      // the weight-gradient products' own copies of df2 / df1 / dao (they may run after the chain: bwd_part)
  // fused attention (attention.hip): row log-sum-exp (saved) and rowsum(dO . O) (backward scratch) instead of S / P / Pd
  float *lse, *dsum;
  int fused;
};

StyleWs carve_style(const ZeggsStyleDims& d, Arena& a) {
  StyleWs w;
  const long B = d.B, L = d.L, BL = B * L, LP = L + 2;
  const int C = d.C, H = d.H, E = d.E, NH = d.NH;
  w.wf0 = a.f(3L * C * H); w.wb0 = nullptr;              // input grad of the first conv is never needed
  w.wf4 = a.f(3L * H * E); w.wb4 = a.f(3L * E * H);
  w.wff0 = a.f(3L * E * E); w.wfb0 = a.f(3L * E * E);
  w.wff2 = a.f(3L * E * E); w.wfb2 = a.f(3L * E * E);
  w.xp = a.f(B * LP * C);
  w.c1 = a.f(B * LP * H); w.m1 = a.f(BL); w.r1 = a.f(BL);      // c1, c2: rows b * LP + t (t < L valid), see conv_flat

  w.a1p = a.f(B * LP * H);
  w.c2 = a.f(B * LP * E); w.m2 = a.f(BL); w.r2 = a.f(BL);
  w.h = a.f(BL * E);
  w.qkv = a.f(BL * 3 * E);
  // (the choice depends on the dimensions and a process-wide tuning switch only: forward and backward of one call pair carve
  //  the same layout unless the switch is flipped in between)
  w.fused = attn_fused_supported(E, NH);
  w.lse = w.dsum = nullptr;
  if (w.fused) {
    w.P = w.Pd = nullptr;
    w.lse = a.f(B * NH * L); w.dsum = a.f(B * NH * L);
  } else {
    w.P = a.f(B * NH * L * L);
    w.Pd = d.dropout ? a.f(B * NH * L * L) : nullptr;
  }
  w.O = a.f(BL * E);
  w.ao = a.f(BL * E); w.ma = a.f(BL); w.ra = a.f(BL);
  w.ap = a.f(B * LP * E);
  w.f1p = a.f(B * LP * E);
  w.f2 = a.f(BL * E); w.mf = a.f(BL); w.rf = a.f(BL);
  w.f = a.f(BL * E);
  w.S = w.fused ? nullptr : a.f(B * NH * L * L);
  long big = BL * (long)(H > 3 * E ? H : 3 * E);
  // t0 / t3 hold padded rows of width H AND of width E (k_pad_rows(..., E)); dwf holds the packed weight gradient of
  // every conv in turn (3*C*H, 3*H*E, 3*E*E): size them for the widest user, whatever the option dictionary says
  const long HE = H > E ? H : E;
  long dw = (long)C * H;
  if ((long)H * E > dw) dw = (long)H * E;
  if ((long)E * E > dw) dw = (long)E * E;
  w.t0 = a.f(B * LP * HE); w.t1 = a.f(big); w.t2 = a.f(big); w.t3 = a.f(B * LP * HE);
  w.dqkv = a.f(BL * 3 * E);
  w.dwf = a.f(3L * dw);
  w.t0a = a.f(B * LP * E); w.t0b = a.f(B * LP * E); w.t1a = a.f(BL * E);
  return w;
}

}  // namespace

extern "C" size_t zeggs_style_encoder_workspace_bytes(const ZeggsStyleDims* d) {
  Arena a(nullptr, 0);
  carve_style(*d, a);
  return a.off + 256;
}

// x: [B, L, C] normalised exemplar features; pos: [>=L, E] sinusoidal table; out: [B, E]
extern "C" int zeggs_style_encoder_fwd(const ZeggsStyleDims* dp, const ZeggsStyleParams* P, const float* x,
                                       const float* pos, float* out, void* ws, size_t ws_bytes, void* stream) {
  return zeggs_style_encoder_fwd_part(dp, P, x, pos, out, ws, ws_bytes, stream, 3);
}
// part: 1 = the head (weight packs, input padding, the first convolution's product: 43 of the encoder's 52 forward GFLOP, one
// chip-filling launch), 2 = everything behind it, 3 = both.  A training loop that runs other queues beside the encoder calls the
// two parts separately and releases those queues BETWEEN them (zeggs/engine.py): the head then has the chip to itself, and the other
// queues' work runs under the chain of small dependent launches that follows, instead of under the one launch that could use it all.
// bit 4 of `part`: the padded input [B][L + 2][C] (one zero row either side of every batch entry) is already in the workspace at
// zeggs_style_encoder_input_offset(): `x` is not read
extern "C" size_t zeggs_style_encoder_input_offset(const ZeggsStyleDims* d) {
  char base[1];
  Arena a(base, ~(size_t)0);        // (a non-null base: pointers are base + offset; nothing is dereferenced)
  StyleWs w = carve_style(*d, a);
  return (size_t)((char*)w.xp - base);
}
extern "C" int zeggs_style_encoder_fwd_part(const ZeggsStyleDims* dp, const ZeggsStyleParams* P, const float* x, const float* pos,
                                            float* out, void* ws, size_t ws_bytes, void* stream, int part) {
  const ZeggsStyleDims& d = *dp;
  hipStream_t s = (hipStream_t)stream;
  Arena a(ws, ws_bytes);
  StyleWs w = carve_style(d, a);
  ZCHECK(a.ok(), "style encoder: workspace too small (%zu < %zu)", ws_bytes, a.off);
  ZCHECK(d.E % d.NH == 0, "style encoder: E %% heads != 0");
  const int B = d.B, L = d.L, C = d.C, H = d.H, E = d.E, NH = d.NH, HD = E / NH, LP = L + 2;
  const long BL = (long)B * L;
  const float p2 = d.dropout ? 0.2f : 0.f, p1 = d.dropout ? 0.1f : 0.f;
  const float eps = 1e-5f;
  const bool fuse0 = H <= 512, fuse = E <= 512;
  const RowView c1v = rv(w.c1, L, (long)LP * H), c2v = rv(w.c2, L, (long)LP * E);
  if (part & 1) {
  // the four convolutions' weights -> k-major packs, one launch
  {
    const PackConvW items[4] = {{w.wf0, nullptr, P->c0_w, H, C, 3}, {w.wf4, w.wb4, P->c4_w, E, H, 3},
                                {w.wff0, w.wfb0, P->ff0_w, E, E, 3}, {w.wff2, w.wfb2, P->ff2_w, E, E, 3}};
    ZTRY(k_pack_conv_w_multi(items, 4, s));
  }
  // conv stack.  Round 5: the first two convolutions run as ONE product over all batches (conv_flat: the padded input [B][LP][C] is
  // one [B LP, C] matrix whose rows overlap; output row b LP + t, the two rows per batch that straddle a boundary are never read)
  // WITHOUT bias / activation, so that the product may split K (stream-K: every CU the same share -- the batched form had 24
  // tiles per batch entry at full K, 1.5 rounds of tiles on half the CUs, and at N = 128 split + zero fill + bias pass); bias and
  // ReLU are folded into the LayerNorm pass that reads the result anyway (LnFwdFused.pre_bias, written back: the backward's
  // LayerNorm input and ReLU mask).
  // (part & 4: the caller's batch gather wrote the padded input where this workspace keeps it -- zeggs_gather_example into
  //  ws + zeggs_style_encoder_input_offset -- so the 2 x 56 MB padding copy at the head of the longest chain before the sweep is gone)
  if (!(part & 4)) ZTRY(k_pad_rows(w.xp, x, B, L, C, 1, 1, 0, s));
  if (fuse0) ZTRY(conv_flat(w.xp, C, w.wf0, 3 * C, H, w.c1, B * LP - 2, s));
  else ZTRY(conv_gemm(w.xp, (long)LP * C, C, w.wf0, 3 * C, H, w.c1, H, (long)LP * H, P->c0_b, B, L, ACT_RELU, s));
  }
  if (!(part & 2)) return 0;
  // LN1 -> interior of padded a1p, dropout, zero edges
  if (fuse0) {     // bias + ReLU + LayerNorm + dropout + the zero edge rows of the next conv's input in one row pass
    LnFwdFused q = ln_fwd_fused_args((int)BL, H, eps);
    q.x = c1v; q.pre_bias = P->c0_b; q.pre_act = ACT_RELU;
    q.y = rv(w.a1p + H, L, (long)LP * H); q.gamma = P->ln0_g; q.beta = P->ln0_b; q.mean = w.m1; q.rstd = w.r1;
    q.p_post = p2; q.seed_post = d.seed + 1; q.pad_L = L;
    ZTRY(k_ln_fwd_fused(q, s));
  } else {
  ZTRY(k_layernorm_fwd_v(rv(w.a1p + H, L, (long)LP * H), c1v, rv(nullptr), P->ln0_g, P->ln0_b, w.m1, w.r1,
                          (int)BL, H, eps, s));
  ZTRY(k_dropout_rows(w.a1p + H, (int)BL, H, H, L, (long)LP * H, p2, d.seed + 1, s));
  ZTRY(k_pad_edges(w.a1p, B, L, H, 1, 1, 0, s));
  }
  if (fuse) ZTRY(conv_flat(w.a1p, H, w.wf4, 3 * H, E, w.c2, B * LP - 2, s));
  else ZTRY(conv_gemm(w.a1p, (long)LP * H, H, w.wf4, 3 * H, E, w.c2, E, (long)LP * E, P->c4_b, B, L, ACT_RELU, s));
  if (fuse) {         // h = dropout(LayerNorm(ReLU(c2 + bias))) + positional table
    LnFwdFused q = ln_fwd_fused_args((int)BL, E, eps);
    q.x = c2v; q.pre_bias = P->c4_b; q.pre_act = ACT_RELU;
    q.y = rv(w.h); q.gamma = P->ln1_g; q.beta = P->ln1_b; q.mean = w.m2; q.rstd = w.r2;
    q.p_post = p2; q.seed_post = d.seed + 2; q.table = pos; q.table_L = L;
    ZTRY(k_ln_fwd_fused(q, s));
  } else {
  ZTRY(k_layernorm_fwd_v(rv(w.h), c2v, rv(nullptr), P->ln1_g, P->ln1_b, w.m2, w.r2, (int)BL, E, eps, s));
  ZTRY(k_dropout(w.h, BL * E, p2, d.seed + 2, s));
  ZTRY(k_add_rows_bcast(w.h, pos, B, L, E, s));
  }
  // multi-head self attention
  ZTRY(gemm_nt(w.h, E, P->in_w, E, w.qkv, 3 * E, P->in_b, (int)BL, 3 * E, E, ACT_NONE, 0.f, s));
  if (w.fused) {      // softmax(Q K^T / sqrt(hd)) -> dropout -> . V in one kernel (attention.hip)
    ZTRY(k_attn_fwd(w.qkv, w.O, w.lse, B, L, E, NH, p1, d.seed + 3, s));
  } else {
  {
    GemmArgs g = gemm_args(w.qkv, w.qkv + E, w.S, L, L, HD);            // S = Q K^T / sqrt(hd)
    g.sam = 3 * E; g.sak = 1; g.sbk = 1; g.sbn = 3 * E; g.scm = L; g.scn = 1;
    g.nb1 = NH; g.bsA0 = (long)L * 3 * E; g.bsA1 = HD; g.bsB0 = (long)L * 3 * E; g.bsB1 = HD;
    g.bsC0 = (long)NH * L * L; g.bsC1 = (long)L * L; g.alpha = 1.0f / sqrtf((float)HD);
    ZTRY(launch_gemm(g, B * NH, s));
  }
  ZTRY(k_softmax_fwd(w.P, w.Pd, w.S, (long)B * NH * L, L, p1, d.seed + 3, s));
  {
    const float* Pm = w.Pd ? w.Pd : w.P;
    GemmArgs g = gemm_args(Pm, w.qkv + 2 * E, w.O, L, HD, L);           // O = P V
    g.sam = L; g.sak = 1; g.sbk = 3 * E; g.sbn = 1; g.scm = E; g.scn = 1;
    g.nb1 = NH; g.bsA0 = (long)NH * L * L; g.bsA1 = (long)L * L; g.bsB0 = (long)L * 3 * E; g.bsB1 = HD;
    g.bsC0 = (long)L * E; g.bsC1 = HD;
    ZTRY(launch_gemm(g, B * NH, s));
  }
  }
  ZTRY(gemm_nt(w.O, E, P->out_w, E, w.ao, E, P->out_b, (int)BL, E, E, ACT_NONE, 0.f, s));
  if (fuse) {         // ao = dropout(ao) (kept: the backward's LayerNorm input), a = LN(ao + h) -> interior of padded ap, edge rows zero
    LnFwdFused q = ln_fwd_fused_args((int)BL, E, eps);
    q.x = rv(w.ao); q.res = rv(w.h); q.y = rv(w.ap + E, L, (long)LP * E); q.gamma = P->lna_g; q.beta = P->lna_b;
    q.mean = w.ma; q.rstd = w.ra; q.p_pre = p1; q.seed_pre = d.seed + 4; q.pad_L = L;
    ZTRY(k_ln_fwd_fused(q, s));
  } else {
  ZTRY(k_dropout(w.ao, BL * E, p1, d.seed + 4, s));
  // a = LN(ao + h) -> interior of padded ap
  ZTRY(k_layernorm_fwd_v(rv(w.ap + E, L, (long)LP * E), rv(w.ao), rv(w.h), P->lna_g, P->lna_b, w.ma, w.ra, (int)BL, E,
                          eps, s));
  ZTRY(k_pad_edges(w.ap, B, L, E, 1, 1, 0, s));
  }
  // position-wise conv feed-forward
  ZTRY(conv_gemm(w.ap, (long)LP * E, E, w.wff0, 3 * E, E, w.f1p + E, E, (long)LP * E, P->ff0_b, B, L, ACT_RELU, s));
  ZTRY(k_pad_edges(w.f1p, B, L, E, 1, 1, 0, s));
  // (ff2's bias: folded into the LayerNorm pass when that is the fused one -- the product then needs no bias pass behind its K split)
  ZTRY(conv_gemm(w.f1p, (long)LP * E, E, w.wff2, 3 * E, E, w.f2, E, (long)L * E, fuse ? nullptr : P->ff2_b, B, L, ACT_NONE, s));
  if (fuse) {         // f2 = dropout(f2 + bias), f = LN(f2 + a)
    LnFwdFused q = ln_fwd_fused_args((int)BL, E, eps);
    q.x = rv(w.f2); q.pre_bias = P->ff2_b; q.pre_act = ACT_NONE; q.res = rv(w.ap + E, L, (long)LP * E); q.y = rv(w.f); q.gamma = P->lnf_g; q.beta = P->lnf_b;
    q.mean = w.mf; q.rstd = w.rf; q.p_pre = p1; q.seed_pre = d.seed + 5;
    ZTRY(k_ln_fwd_fused(q, s));
  } else {
  ZTRY(k_dropout(w.f2, BL * E, p1, d.seed + 5, s));
  ZTRY(k_layernorm_fwd_v(rv(w.f), rv(w.f2), rv(w.ap + E, L, (long)LP * E), P->lnf_g, P->lnf_b, w.mf, w.rf, (int)BL, E,
                          eps, s));
  }
  ZTRY(k_meanpool_fwd(out, w.f, B, L, E, s));
  return 0;
}

extern "C" int zeggs_style_encoder_bwd(const ZeggsStyleDims* dp, const ZeggsStyleParams* P, const float* dout,
                                       const ZeggsStyleGrads* G, void* ws, size_t ws_bytes, void* stream) {
  return zeggs_style_encoder_bwd_ex(dp, P, dout, G, ws, ws_bytes, stream, 0);
}
extern "C" int zeggs_style_encoder_bwd_ex(const ZeggsStyleDims* dp, const ZeggsStyleParams* P, const float* dout,
                                          const ZeggsStyleGrads* G, void* ws, size_t ws_bytes, void* stream, int grads_zeroed) {
  return zeggs_style_encoder_bwd_part(dp, P, dout, G, ws, ws_bytes, stream, grads_zeroed, 3);
}
// part: 1 = the CHAIN (input gradients, LayerNorm / activation passes, attention, bias and LayerNorm-parameter gradients),
// 2 = the six WEIGHT-GRADIENT products (ff2, ff0, out-proj, in-proj, conv c4, conv c0) with their packs, 3 = both, each product right
// behind the pass that makes its operand (what zeggs_style_encoder_bwd_ex does).  The products are chip-filling matrix work nothing in the
// chain waits for, while the chain is ~25 small dependent launches: a training loop with a second queue calls part 1 on the caller's
// stream and part 2 on the other one BEHIND it (the operands of the products -- df2, df1, dao -- are kept in buffers of their own for
// that; part 2 reads `dout` not at all), and joins before the optimizer (zeggs/engine.py: defer_style_wgrads).
extern "C" int zeggs_style_encoder_bwd_part(const ZeggsStyleDims* dp, const ZeggsStyleParams* P, const float* dout,
                                            const ZeggsStyleGrads* G, void* ws, size_t ws_bytes, void* stream, int grads_zeroed,
                                            int part) {
  ZCHECK(part >= 1 && part <= 3, "style encoder bwd: part must be 1, 2 or 3");
  const bool chain = part & 1, inl = part == 3, only_w = part == 2;
  const ZeggsStyleDims& d = *dp;
  const float gb = grads_zeroed ? 1.f : 0.f;      // (see zeggs_speech_encoder_bwd_ex)
  hipStream_t s = (hipStream_t)stream;
  Arena a(ws, ws_bytes);
  StyleWs w = carve_style(d, a);
  ZCHECK(a.ok(), "style encoder bwd: workspace too small");
  const int B = d.B, L = d.L, C = d.C, H = d.H, E = d.E, NH = d.NH, HD = E / NH, LP = L + 2;
  const long BL = (long)B * L;
  const float p2 = d.dropout ? 0.2f : 0.f, p1 = d.dropout ? 0.1f : 0.f;
  float *t0 = w.t0, *t1 = w.t1, *t2 = w.t2, *t3 = w.t3;
  // The elementwise work between the products is four passes, each fused around a LayerNorm backward (k_ln_bwd_fused): the
  // chain is the critical path of the iteration's tail, beside the decoder's weight-gradient GEMMs on the other stream, and every
  // separate small pass (pool, copy, mask, ReLU', bias sum, pad) waited for CU slots there.
  float *t0a = w.t0a, *t0b = w.t0b, *t1a = w.t1a;
  // the six products (each with the pack of its result); `q` = the stream they are launched on
  auto dw_ff2 = [&](hipStream_t q) -> int {
    ZTRY(conv_dw_gemm(w.f1p, (long)LP * E, E, t0a + E, E, (long)LP * E, w.dwf, 3 * E, E, B, L, q));
    return k_unpack_conv_dw(G->ff2_w, w.dwf, E, E, 3, q);
  };
  auto dw_ff0 = [&](hipStream_t q) -> int {
    ZTRY(conv_dw_gemm(w.ap, (long)LP * E, E, t0b + E, E, (long)LP * E, w.dwf, 3 * E, E, B, L, q));
    return k_unpack_conv_dw(G->ff0_w, w.dwf, E, E, 3, q);
  };
  auto dw_out = [&](hipStream_t q) -> int { return gemm_tn(t1a, E, w.O, E, G->out_w, E, (int)BL, E, E, gb, q); };
  auto dw_in = [&](hipStream_t q) -> int { return gemm_tn(w.dqkv, 3 * E, w.h, E, G->in_w, E, (int)BL, 3 * E, E, gb, q); };
  auto dw_c4 = [&](hipStream_t q) -> int {
    ZTRY(conv_dw_gemm(w.a1p, (long)LP * H, H, t0 + E, E, (long)LP * E, w.dwf, 3 * H, E, B, L, q));
    return k_unpack_conv_dw(G->c4_w, w.dwf, E, H, 3, q);
  };
  auto dw_c0 = [&](hipStream_t q) -> int {
    ZTRY(conv_dw_gemm(w.xp, (long)LP * C, C, t2, H, (long)L * H, w.dwf, 3 * C, H, B, L, q));
    return k_unpack_conv_dw(G->c0_w, w.dwf, H, C, 3, q);
  };
  if (only_w) {
    ZTRY(dw_ff2(s)); ZTRY(dw_ff0(s)); ZTRY(dw_out(s)); ZTRY(dw_in(s)); ZTRY(dw_c4(s)); ZTRY(dw_c0(s));
    return 0;
  }
  (void)chain;
  if (!grads_zeroed) {
    ZTRY(k_fill(G->lnf_g, E, 0.f, s)); ZTRY(k_fill(G->lnf_b, E, 0.f, s)); ZTRY(k_fill(G->ff2_b, E, 0.f, s));
    ZTRY(k_fill(G->ff0_b, E, 0.f, s)); ZTRY(k_fill(G->lna_g, E, 0.f, s)); ZTRY(k_fill(G->lna_b, E, 0.f, s));
    ZTRY(k_fill(G->out_b, E, 0.f, s)); ZTRY(k_fill(G->ln1_g, E, 0.f, s)); ZTRY(k_fill(G->ln1_b, E, 0.f, s));
    ZTRY(k_fill(G->c4_b, E, 0.f, s)); ZTRY(k_fill(G->ln0_g, H, 0.f, s)); ZTRY(k_fill(G->ln0_b, H, 0.f, s));
    ZTRY(k_fill(G->c0_b, H, 0.f, s));
    if (w.fused) ZTRY(k_fill(G->in_b, 3 * E, 0.f, s));
  }
  const RowView t0in = rv(t0 + E, L, (long)LP * E);      // interior of the padded [B, LP, E] buffer the input-gradient convs read
  const RowView t0ain = rv(t0a + E, L, (long)LP * E), t0bin = rv(t0b + E, L, (long)LP * E);
  // ---- mean pool, final LN (f = LN(f2 + a)), the mask of ff2's output: t2 = d(f2 + a) (the residual branch keeps it),
  //      t0 interior = df2 = t2 * mask, ff2_b += column sums
  {
    LnBwdFused q = ln_bwd_fused_args((int)BL, E);
    q.dy_pool = dout; q.pool_L = L;
    q.x = rv(w.f2); q.res = rv(w.ap + E, L, (long)LP * E); q.gamma = P->lnf_g; q.mean = w.mf; q.rstd = w.rf;
    q.dgamma = G->lnf_g; q.dbeta = G->lnf_b;
    q.dx_raw = rv(t2); q.out = t0ain; q.pad_L = L; q.p_post = p1; q.seed_post = d.seed + 5; q.dbias = G->ff2_b;
    ZTRY(k_ln_bwd_fused(q, s));
  }
  if (inl) ZTRY(dw_ff2(s));
  // d f1 = conv_bwd(df2): the zero-padded df2 correlated with the flipped taps, then ReLU' (saved f1), ff0_b, padded again
  ZTRY(conv_gemm(t0a, (long)LP * E, E, w.wfb2, 3 * E, E, t1, E, (long)L * E, nullptr, B, L, ACT_NONE, s));
  {
    LnBwdFused q = ln_bwd_fused_args((int)BL, E);
    q.dyA = rv(t1); q.out = t0bin; q.pad_L = L; q.ysave = rv(w.f1p + E, L, (long)LP * E); q.act = ACT_RELU; q.dbias = G->ff0_b;
    ZTRY(k_ln_bwd_fused(q, s));
  }
  if (inl) ZTRY(dw_ff0(s));
  ZTRY(conv_gemm(t0b, (long)LP * E, E, w.wfb0, 3 * E, E, t1, E, (long)L * E, nullptr, B, L, ACT_NONE, s));
  // ---- attention LN: a = LN(ao + h), da = t2 (residual) + t1 (conv path): t3 = d(ao + h) (h's residual share),
  //      t1 = dao = t3 * mask, out_b += column sums
  {
    LnBwdFused q = ln_bwd_fused_args((int)BL, E);
    q.dyA = rv(t2); q.dyB = rv(t1);
    q.x = rv(w.ao); q.res = rv(w.h); q.gamma = P->lna_g; q.mean = w.ma; q.rstd = w.ra; q.dgamma = G->lna_g; q.dbeta = G->lna_b;
    q.dx_raw = rv(t3); q.out = rv(t1a); q.p_post = p1; q.seed_post = d.seed + 4; q.dbias = G->out_b;
    ZTRY(k_ln_bwd_fused(q, s));
  }
  if (inl) ZTRY(dw_out(s));
  ZTRY(gemm_nn(t1a, E, P->out_w, E, t2, E, (int)BL, E, E, 0.f, s));             // t2 = dO [BL,E]
  if (w.fused) {      // dQ, dK, dV with the probabilities recomputed from the saved row log-sum-exp (attention.hip)
    ZTRY(k_attn_bwd(w.qkv, w.O, w.lse, t2, w.dqkv, w.dsum, G->in_b, B, L, E, NH, p1, d.seed + 3, s));
  } else {
  const float* Pm = w.Pd ? w.Pd : w.P;
  {
    GemmArgs g = gemm_args(t2, w.qkv + 2 * E, w.S, L, L, HD);                   // dPd = dO V^T  -> S
    g.sam = E; g.sak = 1; g.sbk = 1; g.sbn = 3 * E; g.scm = L; g.scn = 1;
    g.nb1 = NH; g.bsA0 = (long)L * E; g.bsA1 = HD; g.bsB0 = (long)L * 3 * E; g.bsB1 = HD;
    g.bsC0 = (long)NH * L * L; g.bsC1 = (long)L * L;
    ZTRY(launch_gemm(g, B * NH, s));
  }
  {
    GemmArgs g = gemm_args(Pm, t2, w.dqkv + 2 * E, L, HD, L);                   // dV = Pd^T dO
    g.sam = 1; g.sak = L; g.sbk = E; g.sbn = 1; g.scm = 3 * E; g.scn = 1;
    g.nb1 = NH; g.bsA0 = (long)NH * L * L; g.bsA1 = (long)L * L; g.bsB0 = (long)L * E; g.bsB1 = HD;
    g.bsC0 = (long)L * 3 * E; g.bsC1 = HD;
    ZTRY(launch_gemm(g, B * NH, s));
  }
  ZTRY(k_softmax_bwd(w.S, w.S, w.P, (long)B * NH * L, L, p1, d.seed + 3, s));  // S = dS (in place)
  const float sc = 1.0f / sqrtf((float)HD);
  {
    GemmArgs g = gemm_args(w.S, w.qkv + E, w.dqkv, L, HD, L);                   // dQ = dS K * sc
    g.sam = L; g.sak = 1; g.sbk = 3 * E; g.sbn = 1; g.scm = 3 * E; g.scn = 1;
    g.nb1 = NH; g.bsA0 = (long)NH * L * L; g.bsA1 = (long)L * L; g.bsB0 = (long)L * 3 * E; g.bsB1 = HD;
    g.bsC0 = (long)L * 3 * E; g.bsC1 = HD; g.alpha = sc;
    ZTRY(launch_gemm(g, B * NH, s));
  }
  {
    GemmArgs g = gemm_args(w.S, w.qkv, w.dqkv + E, L, HD, L);                   // dK = dS^T Q * sc
    g.sam = 1; g.sak = L; g.sbk = 3 * E; g.sbn = 1; g.scm = 3 * E; g.scn = 1;
    g.nb1 = NH; g.bsA0 = (long)NH * L * L; g.bsA1 = (long)L * L; g.bsB0 = (long)L * 3 * E; g.bsB1 = HD;
    g.bsC0 = (long)L * 3 * E; g.bsC1 = HD; g.alpha = sc;
    ZTRY(launch_gemm(g, B * NH, s));
  }
  }
  if (inl) ZTRY(dw_in(s));
  if (!w.fused) ZTRY(k_colsum(G->in_b, w.dqkv, BL, 3 * E, 3 * E, gb, s));          // (fused attention: summed by its kernels)
  ZTRY(gemm_nn(w.dqkv, 3 * E, P->in_w, E, t1, E, (int)BL, 3 * E, E, 0.f, s));   // t1 = dh (attention part; pos table: no grad)
  // ---- conv stack: h = dropout(LN(c2)) + pos, c2 = ReLU(conv): dh = (t1 + t3) * mask -> LN backward -> ReLU' -> t0 interior
  //      (padded for the conv's input gradient), c4_b += column sums
  {
    LnBwdFused q = ln_bwd_fused_args((int)BL, E);
    q.dyA = rv(t1); q.dyB = rv(t3); q.p_pre = p2; q.seed_pre = d.seed + 2;
    q.x = rv(w.c2, L, (long)LP * E); q.gamma = P->ln1_g; q.mean = w.m2; q.rstd = w.r2; q.dgamma = G->ln1_g; q.dbeta = G->ln1_b;
    q.out = t0in; q.pad_L = L; q.ysave = rv(w.c2, L, (long)LP * E); q.act = ACT_RELU; q.dbias = G->c4_b;
    ZTRY(k_ln_bwd_fused(q, s));
  }
  if (inl) ZTRY(dw_c4(s));
  ZTRY(conv_gemm(t0, (long)LP * E, E, w.wb4, 3 * E, H, t1, H, (long)L * H, nullptr, B, L, ACT_NONE, s));  // t1 = da1 [BL,H]
  // ---- a1 = dropout(LN(c1)), c1 = ReLU(conv): the same pass at width H, t2 = dc1
  {
    LnBwdFused q = ln_bwd_fused_args((int)BL, H);
    q.dyA = rv(t1); q.p_pre = p2; q.seed_pre = d.seed + 1;
    q.x = rv(w.c1, L, (long)LP * H); q.gamma = P->ln0_g; q.mean = w.m1; q.rstd = w.r1; q.dgamma = G->ln0_g; q.dbeta = G->ln0_b;
    q.out = rv(t2); q.ysave = rv(w.c1, L, (long)LP * H); q.act = ACT_RELU; q.dbias = G->c0_b;
    ZTRY(k_ln_bwd_fused(q, s));
  }
  if (inl) ZTRY(dw_c0(s));
  return 0;
}
