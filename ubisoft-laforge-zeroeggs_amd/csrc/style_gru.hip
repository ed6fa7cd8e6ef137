// Style encoder, recurrent variant (style_encoder.type = "gru"): forward + backward.
//
// Reference: StyleEncoderGRU, ZEGGS/modules.py:307-343: conv(3)+ReLU, conv(3)+ReLU over the exemplar frames,
// ONE bidirectional GRU layer, projection of the last time step.  Only output[:, -1] is used, so
//   * the forward direction is the usual L-step recurrence (L dependent [B,H]x[H,3H] products; its input-side
//     products for all frames are one GEMM written time-major);
//   * the reverse direction contributes its FIRST step only (zero state, input x[L-1]): one gate evaluation.
// Convolutions are GEMMs over zero-padded [L+2, C] buffers as in encoders.hip.  All scratch and saved
// activations live in the caller's workspace (same carve in fwd and bwd).
#include "../../include/zeggs_hip.h"
#include "common.h"
#include "gemm.h"
#include "kernels.h"

namespace {

// nn.GRU cell gate math (gate order r, z, n); gi / gh = input / hidden pre-activations incl. biases
__global__ void sg_gate_fwd_k(const float* gi, const float* gh, const float* hprev, float* hout, float* R, float* Z,
                              float* N, float* NH, int B, int H) {
  long n = (long)B * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int u = (int)(i % H);
    long b = i / H;
    const float* gib = gi + b * 3 * H;
    const float* ghb = gh + b * 3 * H;
    float r = d_sigmoid(gib[u] + ghb[u]);
    float z = d_sigmoid(gib[H + u] + ghb[H + u]);
    float nh = ghb[2 * H + u];
    float nn = d_tanh(gib[2 * H + u] + r * nh);
    hout[i] = (1.f - z) * nn + z * hprev[i];
    R[i] = r; Z[i] = z; N[i] = nn; NH[i] = nh;
  }
}
// dh (ld = lddh) -> di, dhh [B,3H]; dhc = dh * z (direct path to h_prev)
__global__ void sg_gate_bwd_k(const float* dh, long lddh, const float* R, const float* Z, const float* N, const float* NH,
                              const float* hprev, float* di, float* dhh, float* dhc, int B, int H) {
  long n = (long)B * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int u = (int)(i % H);
    long b = i / H;
    float g = dh[b * lddh + u], r = R[i], z = Z[i], nn = N[i], nh = NH[i], hp = hprev[i];
    float dn = g * (1.f - z);
    float dz = g * (hp - nn);
    float dan = dn * (1.f - nn * nn);
    float dar = dan * nh * r * (1.f - r);
    float daz = dz * z * (1.f - z);
    float* dib = di + b * 3 * H;
    float* dhb = dhh + b * 3 * H;
    dib[u] = dar; dib[H + u] = daz; dib[2 * H + u] = dan;
    dhb[u] = dar; dhb[H + u] = daz; dhb[2 * H + u] = dan * r;
    if (dhc) dhc[i] = g * z;
  }
}
__global__ void sg_concat_k(float* cat, const float* a, const float* b, int B, int H) {
  long n = (long)B * 2 * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % (2 * H));
    long r = i / (2 * H);
    cat[i] = c < H ? a[r * H + c] : b[r * H + (c - H)];
  }
}
__global__ void sg_copy_cols_k(float* dst, const float* src, long lds, int w, int B) {
  long n = (long)B * w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dst[i] = src[(i / w) * lds + (i % w)];
}
inline dim3 sg1(long n) { long g = (n + 255) / 256; return dim3((unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g))); }

struct SgWs {
  float *wf0, *wf2, *wb2;                       // packed conv weights
  float *xp, *a1p, *c2, *GI, *gh, *Hs, *R, *Z, *N, *NH, *gir, *ghr, *Rr, *Zr, *Nr, *NHr, *hb, *cat;   // saved
  float *dcat, *DI, *DH, *dh, *dhc, *dir, *dhr, *dC, *t0, *t1, *dwf;                                  // backward
  bool use_fast;      // the forward-direction recurrence runs on the decoder's stage kernels (decoder_fast.hip: sg_fast_*)
  SgFast fast;
};
SgWs carve_sg(const ZeggsStyleGruDims& d, Arena& a) {
  SgWs w;
  const long B = d.B, L = d.L, LP = L + 2, C = d.C, H = d.H;
  w.wf0 = a.f(3 * C * H); w.wf2 = a.f(3 * H * H); w.wb2 = a.f(3 * H * H);
  w.xp = a.f(B * LP * C); w.a1p = a.f(B * LP * H); w.c2 = a.f(B * L * H);
  w.GI = a.f(L * B * 3 * H); w.gh = a.f(B * 3 * H); w.Hs = a.f((L + 1) * B * H);
  w.R = a.f(L * B * H); w.Z = a.f(L * B * H); w.N = a.f(L * B * H); w.NH = a.f(L * B * H);
  w.gir = a.f(B * 3 * H); w.ghr = a.f(B * 3 * H);
  w.Rr = a.f(B * H); w.Zr = a.f(B * H); w.Nr = a.f(B * H); w.NHr = a.f(B * H); w.hb = a.f(B * H); w.cat = a.f(B * 2 * H);
  w.dcat = a.f(B * 2 * H); w.DI = a.f(L * B * 3 * H); w.DH = a.f(L * B * 3 * H); w.dh = a.f(B * H); w.dhc = a.f(B * H);
  w.dir = a.f(B * 3 * H); w.dhr = a.f(B * 3 * H); w.dC = a.f(B * L * H);
  w.t0 = a.f(B * LP * H); w.t1 = a.f(B * L * H);
  w.dwf = a.f(3 * (C > H ? C : H) * H);
  w.use_fast = sg_fast_supported(d.B, d.H) != 0;
  if (w.use_fast) w.fast = sg_fast_carve(d.B, d.H, d.L, a);
  return w;
}
// conv-as-GEMM helpers (same contracts as encoders.hip)
int sg_conv(const float* xp, long xp_bstride, int C, const float* Wf, int Ktot, int Co, float* y, long ldy,
            long y_bstride, const float* bias, int B, int M, int act, hipStream_t s) {
  GemmArgs g = gemm_args(xp, Wf, y, M, Co, Ktot);
  g.sam = C; g.sak = 1; g.sbk = Co; g.sbn = 1; g.scm = ldy; g.scn = 1;
  g.bsA0 = xp_bstride; g.bsB0 = 0; g.bsC0 = y_bstride; g.nb1 = 1;
  g.bias = bias; g.act = act;
  return launch_gemm(g, B, s);
}
int sg_conv_dw(const float* xp, long xp_bstride, int C, const float* dy, long lddy, long dy_bstride, float* dWf, int Ktot,
               int Co, int B, int T, hipStream_t s) {
  GemmArgs g = gemm_args(xp, dy, dWf, Ktot, Co, T);
  g.sam = 1; g.sak = C; g.sbk = lddy; g.sbn = 1; g.scm = Co; g.scn = 1;
  g.kbatch = B; g.kbsA = xp_bstride; g.kbsB = dy_bstride;
  return launch_gemm(g, 1, s);
}

}  // namespace

extern "C" size_t zeggs_style_encoder_gru_workspace_bytes(const ZeggsStyleGruDims* d) {
  Arena a(nullptr, 0);
  carve_sg(*d, a);
  return a.off + 256;
}

// x [B, L, C] normalised exemplar features -> out [B, O]
extern "C" int zeggs_style_encoder_gru_fwd(const ZeggsStyleGruDims* dp, const ZeggsStyleGruParams* P, const float* x,
                                           float* out, void* ws, size_t ws_bytes, void* stream) {
  const ZeggsStyleGruDims& d = *dp;
  hipStream_t s = (hipStream_t)stream;
  ZCHECK(d.B >= 1 && d.L >= 1, "style encoder (gru): empty batch or sequence");
  Arena a(ws, ws_bytes);
  SgWs w = carve_sg(d, a);
  ZCHECK(a.ok(), "style encoder (gru): workspace too small (%zu < %zu)", ws_bytes, a.off);
  const int B = d.B, L = d.L, C = d.C, H = d.H, LP = L + 2;
  const long sH = (long)B * H, s3 = 3 * sH;
  ZTRY(k_pad_rows(w.xp, x, B, L, C, 1, 1, 0, s));
  ZTRY(k_pack_conv_w(w.wf0, nullptr, P->c0_w, H, C, 3, s));
  ZTRY(sg_conv(w.xp, (long)LP * C, C, w.wf0, 3 * C, H, w.a1p + H, H, (long)LP * H, P->c0_b, B, L, ACT_RELU, s));
  ZTRY(k_pad_edges(w.a1p, B, L, H, 1, 1, 0, s));
  ZTRY(k_pack_conv_w(w.wf2, w.wb2, P->c2_w, H, H, 3, s));
  ZTRY(sg_conv(w.a1p, (long)LP * H, H, w.wf2, 3 * H, H, w.c2, H, (long)L * H, P->c2_b, B, L, ACT_RELU, s));
  {   // input-side gate pre-activations of every frame, written time-major [L][B][3H]
    GemmArgs g = gemm_args(w.c2, P->w_ih, w.GI, L, 3 * H, H);
    g.sam = H; g.sak = 1; g.sbk = 1; g.sbn = H; g.scm = (long)B * 3 * H; g.scn = 1;
    g.bsA0 = (long)L * H; g.bsC0 = 3 * H; g.nb1 = 1; g.bias = P->b_ih;
    ZTRY(launch_gemm(g, B, s));
  }
  ZTRY(k_fill(w.Hs, sH, 0.f, s));
  if (w.use_fast) ZTRY(sg_fast_fwd(B, H, L, P->w_hh, P->b_hh, w.GI, w.Hs, w.fast, s));
  else
  for (int t = 0; t < L; ++t) {
    ZTRY(gemm_nt(w.Hs + t * sH, H, P->w_hh, H, w.gh, 3 * H, P->b_hh, B, 3 * H, H, ACT_NONE, 0.f, s));
    hipLaunchKernelGGL(sg_gate_fwd_k, sg1(sH), dim3(256), 0, s, w.GI + t * s3, w.gh, w.Hs + t * sH, w.Hs + (t + 1) * sH,
                       w.R + t * sH, w.Z + t * sH, w.N + t * sH, w.NH + t * sH, B, H);
  }
  // reverse direction at the last position = its first step (zero state)
  ZTRY(gemm_nt(w.c2 + (long)(L - 1) * H, (long)L * H, P->w_ih_r, H, w.gir, 3 * H, P->b_ih_r, B, 3 * H, H, ACT_NONE, 0.f, s));
  ZTRY(gemm_nt(w.Hs, H, P->w_hh_r, H, w.ghr, 3 * H, P->b_hh_r, B, 3 * H, H, ACT_NONE, 0.f, s));
  hipLaunchKernelGGL(sg_gate_fwd_k, sg1(sH), dim3(256), 0, s, w.gir, w.ghr, w.Hs, w.hb, w.Rr, w.Zr, w.Nr, w.NHr, B, H);
  hipLaunchKernelGGL(sg_concat_k, sg1(2 * sH), dim3(256), 0, s, w.cat, w.Hs + (long)L * sH, w.hb, B, H);
  ZLAUNCH_CHECK("style_gru_fwd");
  ZTRY(gemm_nt(w.cat, 2 * H, P->p_w, 2 * H, out, d.O, P->p_b, B, d.O, 2 * H, ACT_NONE, 0.f, s));
  return 0;
}

extern "C" int zeggs_style_encoder_gru_bwd(const ZeggsStyleGruDims* dp, const ZeggsStyleGruParams* P, const float* dout,
                                           const ZeggsStyleGruGrads* G, void* ws, size_t ws_bytes, void* stream) {
  const ZeggsStyleGruDims& d = *dp;
  hipStream_t s = (hipStream_t)stream;
  Arena a(ws, ws_bytes);
  SgWs w = carve_sg(d, a);
  ZCHECK(a.ok(), "style encoder (gru) bwd: workspace too small");
  const int B = d.B, L = d.L, C = d.C, H = d.H, LP = L + 2;
  const long sH = (long)B * H, s3 = 3 * sH, BL = (long)B * L;
  // projection
  ZTRY(gemm_nn(dout, d.O, P->p_w, 2 * H, w.dcat, 2 * H, B, d.O, 2 * H, 0.f, s));
  ZTRY(gemm_tn(dout, d.O, w.cat, 2 * H, G->p_w, 2 * H, B, d.O, 2 * H, 0.f, s));
  ZTRY(k_colsum(G->p_b, dout, B, d.O, d.O, 0.f, s));
  // reverse direction (one step from the zero state: no W_hh_r gradient)
  hipLaunchKernelGGL(sg_gate_bwd_k, sg1(sH), dim3(256), 0, s, w.dcat + H, (long)2 * H, w.Rr, w.Zr, w.Nr, w.NHr, w.Hs, w.dir,
                     w.dhr, (float*)nullptr, B, H);
  ZTRY(gemm_tn(w.dir, 3 * H, w.c2 + (long)(L - 1) * H, (long)L * H, G->w_ih_r, H, B, 3 * H, H, 0.f, s));
  ZTRY(k_colsum(G->b_ih_r, w.dir, B, 3 * H, 3 * H, 0.f, s));
  ZTRY(k_colsum(G->b_hh_r, w.dhr, B, 3 * H, 3 * H, 0.f, s));
  ZTRY(k_fill(G->w_hh_r, 3L * H * H, 0.f, s));
  // forward direction: BPTT over the L frames
  if (w.use_fast) {
    hipLaunchKernelGGL(sg_copy_cols_k, sg1(sH), dim3(256), 0, s, w.dhc, w.dcat, (long)2 * H, H, B);
    ZTRY(sg_fast_bwd(B, H, L, P->w_hh, w.Hs, w.DI, w.dhc, w.fast, s));
  } else
  for (int t = L - 1; t >= 0; --t) {
    const float* dh = (t == L - 1) ? w.dcat : w.dh;
    const long lddh = (t == L - 1) ? 2 * H : H;
    hipLaunchKernelGGL(sg_gate_bwd_k, sg1(sH), dim3(256), 0, s, dh, lddh, w.R + t * sH, w.Z + t * sH, w.N + t * sH,
                       w.NH + t * sH, w.Hs + t * sH, w.DI + t * s3, w.DH + t * s3, w.dhc, B, H);
    if (t > 0) {
      ZTRY(k_copy(w.dh, w.dhc, sH, s));
      ZTRY(gemm_nn(w.DH + t * s3, 3 * H, P->w_hh, H, w.dh, H, B, 3 * H, H, 1.f, s));
    }
  }
  ZLAUNCH_CHECK("style_gru_bwd");
  if (w.use_fast) {      // compact hidden-side gradients: r, z rows = those of DI, n rows in fast.DHn
    ZTRY(gemm_tn(w.DI, 3 * H, w.Hs, H, G->w_hh, H, (int)((long)L * B), 2 * H, H, 0.f, s));
    ZTRY(gemm_tn(w.fast.DHn, H, w.Hs, H, G->w_hh + 2L * H * H, H, (int)((long)L * B), H, H, 0.f, s));
    ZTRY(k_colsum(G->b_hh, w.DI, (long)L * B, 2 * H, 3 * H, 0.f, s));
    ZTRY(k_colsum(G->b_hh + 2 * H, w.fast.DHn, (long)L * B, H, H, 0.f, s));
  } else {
    ZTRY(gemm_tn(w.DH, 3 * H, w.Hs, H, G->w_hh, H, (int)((long)L * B), 3 * H, H, 0.f, s));
    ZTRY(k_colsum(G->b_hh, w.DH, (long)L * B, 3 * H, 3 * H, 0.f, s));
  }
  ZTRY(k_colsum(G->b_ih, w.DI, (long)L * B, 3 * H, 3 * H, 0.f, s));
  {   // dW_ih = sum_b DI_b^T c2_b   (DI time-major, c2 batch-major: batch-reduce over b)
    GemmArgs g = gemm_args(w.DI, w.c2, G->w_ih, 3 * H, H, L);
    g.sam = 1; g.sak = (long)B * 3 * H; g.sbk = H; g.sbn = 1; g.scm = H; g.scn = 1;
    g.kbatch = B; g.kbsA = 3 * H; g.kbsB = (long)L * H;
    ZTRY(launch_gemm(g, 1, s));
  }
  {   // dc2_b = DI_b W_ih  -> batch-major [B, L, H]
    GemmArgs g = gemm_args(w.DI, P->w_ih, w.dC, L, H, 3 * H);
    g.sam = (long)B * 3 * H; g.sak = 1; g.sbk = H; g.sbn = 1; g.scm = H; g.scn = 1;
    g.bsA0 = 3 * H; g.bsC0 = (long)L * H; g.nb1 = 1;
    ZTRY(launch_gemm(g, B, s));
  }
  ZTRY(gemm_nn(w.dir, 3 * H, P->w_ih_r, H, w.dC + (long)(L - 1) * H, (long)L * H, B, 3 * H, H, 1.f, s));
  // conv stack
  ZTRY(k_act_bwd(w.dC, w.dC, w.c2, BL * H, ACT_RELU, 1.f, s));
  ZTRY(k_colsum(G->c2_b, w.dC, BL, H, H, 0.f, s));
  ZTRY(sg_conv_dw(w.a1p, (long)LP * H, H, w.dC, H, (long)L * H, w.dwf, 3 * H, H, B, L, s));
  ZTRY(k_unpack_conv_dw(G->c2_w, w.dwf, H, H, 3, s));
  ZTRY(k_pad_rows(w.t0, w.dC, B, L, H, 1, 1, 0, s));
  ZTRY(sg_conv(w.t0, (long)LP * H, H, w.wb2, 3 * H, H, w.t1, H, (long)L * H, nullptr, B, L, ACT_NONE, s));
  ZTRY(k_act_bwd_v(rv(w.t1), rv(w.t1), rv(w.a1p + H, L, (long)LP * H), BL, H, ACT_RELU, 1.f, s));
  ZTRY(k_colsum(G->c0_b, w.t1, BL, H, H, 0.f, s));
  ZTRY(sg_conv_dw(w.xp, (long)LP * C, C, w.t1, H, (long)L * H, w.dwf, 3 * C, H, B, L, s));
  ZTRY(k_unpack_conv_dw(G->c0_w, w.dwf, H, C, 3, s));
  return 0;
}
