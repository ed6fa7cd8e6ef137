// Autoregressive gesture decoder: cell-state encoder + T-1 sequential steps of
//   vectorize_input -> Linear+ELU -> 2-layer GRU (seq len 1) -> Linear -> devectorize_output
// (reference ZEGGS/modules.py:47-243, 677-742) and the matching BPTT.
//
// Data layout (all time-major so that one step's rows are contiguous and the
// weight-gradient GEMMs contract over the flattened (t, b) axis):
//   Gin [T][B][GL]   GL = round4(H + XD): per step [hid(H) | x(XD)], x = [pose(PI) | speech | style]
//   H0/H1 [T][B][H]  hidden states, slot 0 = CellStateEncoder output
//   R*,Z*,N*,NH*     [T][B][H] saved gates (NH = W_hn h + b_hn) for both layers
//   D*               [T][B][.] gate/pre-activation gradients produced by the backward sweep
#include "../../include/zeggs_hip.h"
#include "common.h"
#include "gemm.h"
#include "kernels.h"
#include "dec_math.h"
#include "dec_prologue.h"
#include "decoder_ws.h"

int g_decoder_fast = 1;      // option "decoder_fast": 0 = generic per-step GEMM path everywhere (A/B reference of the stage kernels)
static int g_bwd_chunks = 1;   // BPTT sweep chunks whose weight-gradient GEMMs overlap the rest of the sweep (1: serial)
extern int g_stage_variant;
extern int g_gemm_wg_target;
extern int g_timing;
extern int g_chain;
extern int g_sweep_graphs;
extern int g_launch_window;
extern int g_persistent;
extern int g_train_persistent;
extern int g_bwd_persistent;
extern int g_fused_attention;
extern int g_gemm_streamk;
extern int g_gemm_skinny;
extern int g_tp_tiles4;
extern int g_tp_dual;
extern int g_loss_lds;
extern int g_gemm_split_bf16;
int g_tp_prologue = 1;      // zeggs_set_option("tp_prologue", 0 / 1): the training rollout's prologue in five launches instead of ten
int g_wgrad_order = 0;      // zeggs_set_option("wgrad_order", 0 / 1 / 2): see dec_recurrent_wgrads
void zeggs_gemm_set_dma(int on);
void zeggs_gemm_set_direct(int mode, int wgs);
void zeggs_gemm_set_direct_depth(int d);
void zeggs_gemm_set_direct_shield(int on);
void zeggs_gemm_set_direct_reserve(int n);
void zeggs_gemm_set_asum(int on);
extern int g_gemm_mid_split;
extern int g_gemm_streamk_wgs;
extern int g_mel_mfma;
extern int g_mel_fft;
extern int g_ln_bwd4;
extern int g_mel_exact_log;
extern int g_attn_bwd_one_launch;
extern "C" int zeggs_set_option(const char* name, int value) {
  if (strcmp(name, "attn_bwd_one_launch") == 0) { g_attn_bwd_one_launch = value != 0; return 0; }
  if (strcmp(name, "decoder_fast") == 0) { g_decoder_fast = value; return 0; }
  if (strcmp(name, "stage_variant") == 0) { g_stage_variant = value; return 0; }
  if (strcmp(name, "gemm_wg_target") == 0) { g_gemm_wg_target = value; return 0; }
  if (strcmp(name, "timing") == 0) { g_timing = value; return 0; }
  if (strcmp(name, "chain") == 0) {
#ifndef ZEGGS_CHAIN
    if (value) { zeggs_set_error("option chain: the chained (run-ahead) stage launches lost to the persistent decode kernel and are "
                                 "compiled in measurement builds only (-DZEGGS_CHAIN)"); return -1; }
#endif
    g_chain = value; return 0;
  }
  if (strcmp(name, "sweep_graphs") == 0) { g_sweep_graphs = value != 0; return 0; }
  if (strcmp(name, "launch_window") == 0) { g_launch_window = value < 0 ? 0 : value; return 0; }
  // (re-)enabling gives a kernel that was disabled after a failed validation another chance
  if (strcmp(name, "train_persistent") == 0) {
    g_train_persistent = value;
    if (value && dec_tp_state() == 0) dec_tp_set_state(-1);
    return 0;
  }
  if (strcmp(name, "bwd_persistent") == 0) {
    g_bwd_persistent = value;
    if (value && dec_bp_state() == 0) dec_bp_set_state(-1);
    return 0;
  }
  if (strcmp(name, "persistent") == 0) {
    g_persistent = value;
    if (value && dec_persistent_state() == 0) dec_persistent_set_state(-1);
    return 0;
  }
  if (strcmp(name, "mel_mfma") == 0) { g_mel_mfma = value != 0; return 0; }
  if (strcmp(name, "mel_fft") == 0) { g_mel_fft = value != 0; return 0; }
  if (strcmp(name, "gemm_streamk_wgs") == 0) { g_gemm_streamk_wgs = value; return 0; }
  if (strcmp(name, "gemm_mid_split") == 0) { g_gemm_mid_split = value != 0; return 0; }
  if (strcmp(name, "gemm_dma") == 0) { zeggs_gemm_set_dma(value != 0); return 0; }
  if (strcmp(name, "gemm_direct") == 0) { zeggs_gemm_set_direct(value, -1); return 0; }
  if (strcmp(name, "gemm_direct_wgs") == 0) { zeggs_gemm_set_direct(-1, value); return 0; }
  if (strcmp(name, "gemm_direct_depth") == 0) { zeggs_gemm_set_direct_depth(value); return 0; }
  if (strcmp(name, "gemm_direct_shield") == 0) { zeggs_gemm_set_direct_shield(value); return 0; }
  if (strcmp(name, "gemm_direct_reserve") == 0) { zeggs_gemm_set_direct_reserve(value); return 0; }
  if (strcmp(name, "gemm_asum") == 0) { zeggs_gemm_set_asum(value); return 0; }
  if (strcmp(name, "gemm_skinny") == 0) { g_gemm_skinny = value != 0; return 0; }
  if (strcmp(name, "gemm_streamk") == 0) { g_gemm_streamk = value != 0; return 0; }
  if (strcmp(name, "fused_attention") == 0) { g_fused_attention = value != 0; return 0; }
  if (strcmp(name, "bwd_chunks") == 0) { g_bwd_chunks = value < 1 ? 1 : value; return 0; }
  // bound of every device-side wait of the persistent kernels (polls); 0 makes the first unsatisfied wait give up: the
  // tests use it to drive the give-up path (tests/test_gpu_giveup.py)
  if (strcmp(name, "tp_tiles4") == 0) { g_tp_tiles4 = value != 0; return 0; }
  if (strcmp(name, "tp_dual") == 0) { g_tp_dual = value != 0; return 0; }
  if (strcmp(name, "tp_prologue") == 0) { g_tp_prologue = value != 0; return 0; }
  if (strcmp(name, "loss_lds") == 0) { g_loss_lds = value != 0; return 0; }
  if (strcmp(name, "wgrad_order") == 0) { g_wgrad_order = value; return 0; }
  if (strcmp(name, "gemm_split_bf16") == 0) { g_gemm_split_bf16 = (value == 3 || value == 6 || value == 9) ? value : 0; return 0; }
  if (strcmp(name, "poll_stagger") == 0) { g_poll_stagger = value < 0 ? 0 : value; return 0; }
  if (strcmp(name, "poll_sleep") == 0) { g_poll_sleep = value < 0 ? 0 : value; return 0; }
  if (strcmp(name, "persistent_spin") == 0) { g_persistent_spin = value < 0 ? 0 : value; return 0; }
  if (strcmp(name, "ln_bwd4") == 0) { g_ln_bwd4 = value; return 0; }
  if (strcmp(name, "mel_exact_log") == 0) { g_mel_exact_log = value; return 0; }
  zeggs_set_error("unknown option %s", name);
  return -1;
}

// What tells a caller that a persistent kernel gave up AFTER its first (validated) use -- e.g. a co-tenant took CUs, so not
// every workgroup was resident and a bounded wait ran out: (1) the kernel ORs its bit into the caller-owned sticky status word
// (ZeggsDecCall.status) that zeggs_radam_step_guarded reads on the device -- the optimizer step of that iteration is then a
// no-op, nothing invalid reaches the weights -- and that the caller inspects whenever it likes; (2) it writes NaN into what its
// consumers read first (last output frame / the carries of the CellStateEncoder backward).  The library keeps no per-process
// watch state of its own.
// 1: the persistent kernel was validated on this process, 0: it failed once and is disabled, -1: not used yet
extern "C" int zeggs_persistent_state(int which /* 0 decode (B=1), 1 training forward, 2 BPTT sweep */) {
  return which == 0 ? dec_persistent_state() : which == 1 ? dec_tp_state() : dec_bp_state();
}

namespace {

// ------------------------------------------------------------------ kernels
// init: frame-0 outputs, CellStateEncoder input (gaze of frame 0) and the pose part of x_1 (gaze of frame 1)
__global__ void dec_init_k(ZeggsDecDims d, ZeggsDecStats st, const float* pose0, const float* rp0, const float* rr0,
                           const float* gaze, const float* style, float* pose, float* rpos, float* rrot,
                           float* cse_in, float* gin1, int GL) {
  dec_init_body(blockIdx.x, d, st, pose0, rp0, rr0, gaze, style, pose, rpos, rrot, cse_in, gin1, GL);
}

// speech / style columns of x_t for one step (or all steps when nt > 1): Gin[t][b][H+PI ...]
__global__ void dec_fill_cond_k(ZeggsDecDims d, const float* speech, const float* style, float* gin, int GL, int t0,
                                int nt, long slot_stride, int ring) {
  dec_fill_cond_body(blockIdx.x, gridDim.x, d, speech, style, gin, GL, t0, nt, slot_stride, ring);
}

// GRU cell gate math (nn.GRU, gate order r,z,n)
__global__ void gru_gate_fwd_k(const float* gi, const float* gh, const float* hprev, float* hout, f4* GT, int B, int H) {
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
    float hp = hprev[i];
    hout[i] = (1.f - z) * nn + z * hp;
    if (GT) GT[i] = f4{r, z, nn, nh};
  }
}

// dh (total grad wrt h') -> di [B,3H] (grad wrt W_ih x + b_ih), dhh [B,3H] (grad wrt W_hh h + b_hh),
// dhc = dh * z (direct path to h_prev)
__global__ void gru_gate_bwd_k(const float* dh, const f4* GT, const float* hprev, float* di, float* dhh, float* dhc, int B,
                               int H) {
  long n = (long)B * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int u = (int)(i % H);
    long b = i / H;
    const f4 gt = GT[i];
    float g = dh[i], r = gt.x, z = gt.y, nn = gt.z, nh = gt.w, hp = hprev[i];
    float dn = g * (1.f - z);
    float dz = g * (hp - nn);
    float dan = dn * (1.f - nn * nn);
    float dar = dan * nh * r * (1.f - r);
    float daz = dz * z * (1.f - z);
    float* dib = di + b * 3 * H;
    float* dhb = dhh + b * 3 * H;
    dib[u] = dar; dib[H + u] = daz; dib[2 * H + u] = dan;
    dhb[u] = dar; dhb[H + u] = daz; dhb[2 * H + u] = dan * r;
    dhc[i] = g * z;
  }
}

// devectorize_output + next step's vectorize_input for one step t
__global__ void dec_devec_k(ZeggsDecDims d, ZeggsDecStats st, const float* y, int POL, const float* gaze, float* pose,
                            float* rpos, float* rrot, float* gin_next, int GL, int t) {
  const int b = blockIdx.x;
  const float* yb = y + (long)b * POL;
  float* pb = pose + ((long)b * d.T + t) * d.PO;
  for (int c = threadIdx.x; c < d.PO; c += blockDim.x) {
    float p = yb[c] * st.out_std[c] + st.out_mean[c];
    pb[c] = p;
    if (gin_next) gin_next[(long)b * GL + d.H + c] = (p - st.in_mean[c]) / st.in_std[c];
  }
  if (threadIdx.x == 0) {
    float p[6];
    for (int c = 0; c < 6; ++c) p[c] = yb[c] * st.out_std[c] + st.out_mean[c];
    const float* rq = rrot + ((long)b * d.T + t - 1) * 4;
    const float* rp = rpos + ((long)b * d.T + t - 1) * 3;
    Q4 q = Q4{rq[0], rq[1], rq[2], rq[3]};
    V3 pos = v3(rp[0], rp[1], rp[2]);
#if defined(ZEGGS_ROOT_FP64) && ZEGGS_ROOT_FP64
    // DIAGNOSTIC build (tools/drift_ab.py): the root integration of the generic path evaluated in float64 from the same fp32
    // inputs, rounded once at the end -- separates "rounding inside the root integration" from everything else
    V3 npos; Q4 nq;
    {
      auto qmv = [](const double* q4, const double* v, double* o) {
        const double tx = 2.0 * (q4[2] * v[2] - q4[3] * v[1]), ty = 2.0 * (q4[3] * v[0] - q4[1] * v[2]),
                     tz = 2.0 * (q4[1] * v[1] - q4[2] * v[0]);
        o[0] = v[0] + q4[0] * tx + (q4[2] * tz - q4[3] * ty);
        o[1] = v[1] + q4[0] * ty + (q4[3] * tx - q4[1] * tz);
        o[2] = v[2] + q4[0] * tz + (q4[1] * ty - q4[2] * tx);
      };
      const double qd[4] = {q.w, q.x, q.y, q.z}, dtd = (double)d.dt;
      const double v[3] = {dtd * p[0], dtd * p[1], dtd * p[2]}, w3[3] = {dtd * p[3], dtd * p[4], dtd * p[5]};
      double o[3], u3[3];
      qmv(qd, v, o);
      npos = v3((float)(o[0] + pos.x), (float)(o[1] + pos.y), (float)(o[2] + pos.z));
      qmv(qd, w3, u3);
      const double hx = 0.5 * u3[0], hy = 0.5 * u3[1], hz = 0.5 * u3[2], h = sqrt(hx * hx + hy * hy + hz * hz);
      double e[4];
      if (h < 1e-5) { const double n = sqrt(1.0 + h * h) + 1e-5; e[0] = 1.0 / n; e[1] = hx / n; e[2] = hy / n; e[3] = hz / n; }
      else { const double sc = sin(h) / h; e[0] = cos(h); e[1] = hx * sc; e[2] = hy * sc; e[3] = hz * sc; }
      // quat_mul(e, q)
      nq = Q4{(float)(qd[0] * e[0] - qd[1] * e[1] - qd[2] * e[2] - qd[3] * e[3]),
              (float)(qd[0] * e[1] + qd[1] * e[0] - qd[2] * e[3] + qd[3] * e[2]),
              (float)(qd[0] * e[2] + qd[1] * e[3] + qd[2] * e[0] - qd[3] * e[1]),
              (float)(qd[0] * e[3] - qd[1] * e[2] + qd[2] * e[1] + qd[3] * e[0])};
    }
    V3 u = v3(0.f, 0.f, 0.f); (void)u;
#else
    V3 npos = quat_mul_vec(q, d.dt * v3(p[0], p[1], p[2])) + pos;
    V3 u = quat_mul_vec(q, d.dt * v3(p[3], p[4], p[5]));
    Q4 nq = quat_exp_mul(0.5f * u, q);
#endif
    float* op = rpos + ((long)b * d.T + t) * 3; op[0] = npos.x; op[1] = npos.y; op[2] = npos.z;
    float* oq = rrot + ((long)b * d.T + t) * 4; oq[0] = nq.w; oq[1] = nq.x; oq[2] = nq.y; oq[3] = nq.z;
    if (gin_next) {
      const float* gz = gaze + ((long)b * d.T + t + 1) * 3;
      V3 gd = quat_mul_vec(quat_inv(nq), v3(gz[0], gz[1], gz[2]) - npos);
      float gv[3] = {gd.x, gd.y, gd.z};
      for (int k = 0; k < 3; ++k)
        gin_next[(long)b * GL + d.H + d.PO + k] = (gv[k] - st.in_mean[d.PO + k]) / st.in_std[d.PO + k];
    }
  }
}

// backward of dec_devec_k for step t.
//   dxn   [B, XD]  grad wrt x_{t+1} (pose part used), null at the last step
//   carry [B, 8]   grad wrt (rpos_t[3], rrot_t[4]) coming from steps > t (updated to frame t-1's)
//   dy    [B, POL] output: grad wrt the raw network output of step t
__global__ void dec_devec_bwd_k(ZeggsDecDims d, ZeggsDecStats st, const float* dpose, const float* drpos,
                                const float* drrot, const float* dxn, int XD, const float* gaze, const float* pose,
                                const float* rpos, const float* rrot, float* carry, float* dy, int POL, int t) {
  const int b = blockIdx.x;
  const float* dpb = dpose + ((long)b * d.T + t) * d.PO;
  const float* dxb = dxn ? dxn + (long)b * XD : nullptr;
  float* dyb = dy + (long)b * POL;
  for (int c = 6 + threadIdx.x; c < d.PO; c += blockDim.x) {
    float g = dpb[c] + (dxb ? dxb[c] / st.in_std[c] : 0.f);
    dyb[c] = g * st.out_std[c];
  }
  if (threadIdx.x == 0) {
    float g6[6];
    for (int c = 0; c < 6; ++c) g6[c] = dpb[c] + (dxb ? dxb[c] / st.in_std[c] : 0.f);
    float* cr = carry + b * 8;
    const float* a = drpos + ((long)b * d.T + t) * 3;
    const float* e = drrot + ((long)b * d.T + t) * 4;
    V3 g_rp = v3(cr[0] + a[0], cr[1] + a[1], cr[2] + a[2]);
    Q4 g_rr = Q4{cr[3] + e[0], cr[4] + e[1], cr[5] + e[2], cr[6] + e[3]};
    const float* rq = rrot + ((long)b * d.T + t) * 4;
    const float* rp = rpos + ((long)b * d.T + t) * 3;
    Q4 q_t = Q4{rq[0], rq[1], rq[2], rq[3]};
    V3 p_t = v3(rp[0], rp[1], rp[2]);
    if (dxb) {   // gaze direction of x_{t+1}
      const float* gz = gaze + ((long)b * d.T + t + 1) * 3;
      V3 dgd = v3(dxb[d.PO] / st.in_std[d.PO], dxb[d.PO + 1] / st.in_std[d.PO + 1],
                  dxb[d.PO + 2] / st.in_std[d.PO + 2]);
      Q4 dqi; V3 dv;
      qmv_bwd(quat_inv(q_t), v3(gz[0], gz[1], gz[2]) - p_t, dgd, dqi, dv);
      g_rr.w += dqi.w; g_rr.x -= dqi.x; g_rr.y -= dqi.y; g_rr.z -= dqi.z;
      g_rp = g_rp - dv;
    }
    const float* pq = rrot + ((long)b * d.T + t - 1) * 4;
    Q4 q_p = Q4{pq[0], pq[1], pq[2], pq[3]};
    const float* pt = pose + ((long)b * d.T + t) * d.PO;
    V3 vel = v3(pt[0], pt[1], pt[2]), vrt = v3(pt[3], pt[4], pt[5]);
    // rpos_t = qmv(q_p, vel dt) + rpos_{t-1}
    Q4 dq1; V3 dv1;
    qmv_bwd(q_p, d.dt * vel, g_rp, dq1, dv1);
    // rrot_t = qmul(E, q_p), E = qexp(u/2), u = qmv(q_p, vrt dt)
    V3 u = quat_mul_vec(q_p, d.dt * vrt);
    Q4 E = quat_exp(0.5f * u);
    Q4 dE, dqy;
    qmul_bwd(E, q_p, g_rr, dE, dqy);
    V3 du = 0.5f * qexp_bwd(0.5f * u, dE);
    Q4 dq2; V3 dv2;
    qmv_bwd(q_p, d.dt * vrt, du, dq2, dv2);
    g6[0] += d.dt * dv1.x; g6[1] += d.dt * dv1.y; g6[2] += d.dt * dv1.z;
    g6[3] += d.dt * dv2.x; g6[4] += d.dt * dv2.y; g6[5] += d.dt * dv2.z;
    cr[0] = g_rp.x; cr[1] = g_rp.y; cr[2] = g_rp.z;
    cr[3] = dq1.w + dqy.w + dq2.w; cr[4] = dq1.x + dqy.x + dq2.x;
    cr[5] = dq1.y + dqy.y + dq2.y; cr[6] = dq1.z + dqy.z + dq2.z;
    for (int c = 0; c < 6; ++c) dyb[c] = g6[c] * st.out_std[c];
  }
}

// d0[b][u] = dgin[b][u] * ELU'(hid[b][u]);  rows strided by GL
__global__ void elu_bwd_rows_k(float* d0, const float* dgin, const float* gin, int B, int H, int GL) {
  long n = (long)B * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int u = (int)(i % H);
    long b = i / H;
    d0[i] = dgin[b * GL + u] * d_elu_grad_from_out(gin[b * GL + u]);
  }
}

// copy rows: dst[b][0:w] = src[b][off : off+w]
__global__ void copy_cols_k(float* dst, long ldd, const float* src, long lds, int off, int w, int B) {
  long n = (long)B * w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % w);
    long b = i / w;
    dst[b * ldd + c] = src[b * lds + off + c];
  }
}

// scatter time-major dX [T][B][XD] speech/style columns into batch-major outputs
__global__ void dec_scatter_cond_grad_k(ZeggsDecDims d, const float* DX, int XD, float* dspeech, float* dstyle) {
  const int XC = d.SP + (d.film ? 0 : d.ST);
  long n = (long)d.T * d.B * XC;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % XC);
    long r = i / XC;
    int b = (int)(r % d.B);
    int t = (int)(r / d.B);
    float v = t == 0 ? 0.f : DX[((long)t * d.B + b) * XD + d.PI + c];
    if (c < d.SP) dspeech[((long)b * d.T + t) * d.SP + c] = v;
    else dstyle[((long)b * d.T + t) * d.ST + (c - d.SP)] = v;
  }
}
// ---- FiLM (reference modules.py:213-225): out = a * (1 + gamma) + beta; all operands row-strided views
__global__ void film_fwd_k(float* out, long ldo, const float* a, const float* gam, const float* bet, long ldg, int B,
                           int H) {
  long n = (long)B * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int u = (int)(i % H);
    long b = i / H;
    out[b * ldo + u] = a[i] * (1.f + gam[b * ldg + u]) + bet[b * ldg + u];
  }
}
// g = grad wrt the modulated value -> dpre = g (1+gamma) ELU'(a) (a = ELU output), dgamma = g a, dbeta = g
__global__ void film_bwd_k(const float* g, long ldgr, const float* a, const float* gam, long ldg, float* dpre,
                           float* dgam, float* dbet, int B, int H) {
  long n = (long)B * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int u = (int)(i % H);
    long b = i / H;
    const float gv = g[b * ldgr + u], av = a[i];
    dpre[i] = gv * (1.f + gam[b * ldg + u]) * d_elu_grad_from_out(av);
    dgam[b * ldg + u] = gv * av;
    dbet[b * ldg + u] = gv;
  }
}
// batch-major style [B,T,ST] <-> time-major [T,B,ST]
__global__ void style_time_major_k(ZeggsDecDims d, const float* style, float* stm) {
  long n = (long)d.T * d.B * d.ST;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % d.ST);
    long r = i / d.ST;
    int b = (int)(r % d.B), t = (int)(r / d.B);
    stm[i] = style[((long)b * d.T + t) * d.ST + c];
  }
}
__global__ void style_grad_from_time_major_k(ZeggsDecDims d, const float* dstm, float* dstyle) {
  long n = (long)d.T * d.B * d.ST;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % d.ST);
    long r = i / d.ST;
    int b = (int)(r % d.B), t = (int)(r / d.B);
    dstyle[((long)b * d.T + t) * d.ST + c] = t == 0 ? 0.f : dstm[i];
  }
}
__global__ void add_style0_grad_k(ZeggsDecDims d, const float* dcse_in, float* dstyle) {
  long n = (long)d.B * d.ST;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % d.ST);
    long b = i / d.ST;
    dstyle[(b * d.T) * d.ST + c] += dcse_in[b * (d.PI + d.ST) + d.PI + c];
  }
}

// weight gradients of the recurrent part for the steps t_lo..t_hi: contraction over the flattened (t, b) rows
// compact != 0 (fast path): DH0 / DH1 hold only the n rows of the hidden-side gate gradients, [T][B][H]; their r, z rows
// are the r, z rows of DI0 / DI1.
// what: 1 = weight gradients (GEMMs) of layer2 and GRU layer 1, 4 = of GRU layer 0 and layer0 (the two halves of the decoder's
// slice of a flat gradient buffer in parameter order: the caller can all-reduce the first while the second is computed),
// 2 = the bias gradients (column sums); 7 = everything
int dec_recurrent_wgrads(const ZeggsDecDims& d, const DecWs& w, const ZeggsDecGrads* G, int t_lo, int t_hi, float beta,
                         int compact, hipStream_t s, int what = 7) {
  const int B = d.B, H = d.H, GL = w.GL, XD = w.XD, POL = w.POL;
  const long sG = (long)B * GL, sH = (long)B * H, s3 = 3 * sH, sY = (long)B * POL;
  const int M = (t_hi - t_lo + 1) * B;
  const long o = t_lo;
  if (d.film) {
    const long sg = (long)B * 2 * H, sS = (long)B * d.ST;
    if (what & 1) ZTRY(gemm_tn(w.DY + o * sY, POL, w.F2 + o * sH, H, G->l3_w, H, M, d.PO, H, beta, s));
    if (what & 2) ZTRY(k_colsum(G->l3_b, w.DY + o * sY, M, d.PO, POL, beta, s));
    if (what & 1) ZTRY(gemm_tn(w.D2 + o * sH, H, w.H1 + o * sH, H, G->l2_w, H, M, H, H, beta, s));
    if (what & 2) ZTRY(k_colsum(G->l2_b, w.D2 + o * sH, M, H, H, beta, s));
    if (what & 1) ZTRY(gemm_tn(w.DGAM + o * sg, 2 * H, w.STm + o * sS, d.ST, G->g_w, d.ST, M, 2 * H, d.ST, beta, s));
    if (what & 2) ZTRY(k_colsum(G->g_b, w.DGAM + o * sg, M, 2 * H, 2 * H, beta, s));
    if (what & 1) ZTRY(gemm_tn(w.DBET + o * sg, 2 * H, w.STm + o * sS, d.ST, G->be_w, d.ST, M, 2 * H, d.ST, beta, s));
    if (what & 2) ZTRY(k_colsum(G->be_b, w.DBET + o * sg, M, 2 * H, 2 * H, beta, s));
  } else {
    if ((what & 3) == 3) ZTRY(gemm_tn_bias(w.DY + o * sY, POL, w.H1 + o * sH, H, G->l2_w, H, M, d.PO, H, beta, G->l2_b, s));
    else {
    if (what & 1) ZTRY(gemm_tn(w.DY + o * sY, POL, w.H1 + o * sH, H, G->l2_w, H, M, d.PO, H, beta, s));
    if (what & 2) ZTRY(k_colsum(G->l2_b, w.DY + o * sY, M, d.PO, POL, beta, s));
    }
  }
  // a weight-gradient product and the bias column sums of the same dy: ONE launch where both are asked for in this call and the
  // direct kernel takes the product (gemm_tn_bias: round 5, the sums were a dozen launches of 4-18 workgroups between the products)
  auto tn = [&](int gbit, const float* dy, long lddy, const float* x, long ldx, float* dW, long lddw, int N, int K, float* db) -> int {
    if ((what & gbit) && (what & 2)) return gemm_tn_bias(dy, lddy, x, ldx, dW, lddw, M, N, K, beta, db, s);
    if (what & gbit) ZTRY(gemm_tn(dy, lddy, x, ldx, dW, lddw, M, N, K, beta, s));
    if (what & 2) ZTRY(k_colsum(db, dy, M, N, lddy, beta, s));
    return 0;
  };
  // option "wgrad_order" (A/B, round 6): where the biggest product (dW_ih0, 0.8 ms of the second queue's 3.3) stands among the
  // others -- 0: parameter order (shipped), 1: first of its group, in front of layer 1's too when both groups are in this call,
  // 2: last of all
  auto ih0 = [&]() -> int { return tn(4, w.DI0 + o * s3, 3 * H, w.Gin + o * sG, GL, G->w_ih0, H + XD, 3 * H, H + XD, G->b_ih0); };
  if (g_wgrad_order == 1) ZTRY(ih0());
  ZTRY(tn(1, w.DI1 + o * s3, 3 * H, w.H0 + o * sH, H, G->w_ih1, H, 3 * H, H, G->b_ih1));
  if (compact) {
    ZTRY(tn(1, w.DI1 + o * s3, 3 * H, w.H1 + (o - 1) * sH, H, G->w_hh1, H, 2 * H, H, G->b_hh1));
    ZTRY(tn(1, w.DH1 + o * sH, H, w.H1 + (o - 1) * sH, H, G->w_hh1 + 2L * H * H, H, H, H, G->b_hh1 + 2 * H));
  } else {
    ZTRY(tn(1, w.DH1 + o * s3, 3 * H, w.H1 + (o - 1) * sH, H, G->w_hh1, H, 3 * H, H, G->b_hh1));
  }
  if (g_wgrad_order == 0) ZTRY(ih0());
  if (compact) {
    ZTRY(tn(4, w.DI0 + o * s3, 3 * H, w.H0 + (o - 1) * sH, H, G->w_hh0, H, 2 * H, H, G->b_hh0));
    ZTRY(tn(4, w.DH0 + o * sH, H, w.H0 + (o - 1) * sH, H, G->w_hh0 + 2L * H * H, H, H, H, G->b_hh0 + 2 * H));
  } else {
    ZTRY(tn(4, w.DH0 + o * s3, 3 * H, w.H0 + (o - 1) * sH, H, G->w_hh0, H, 3 * H, H, G->b_hh0));
  }
  ZTRY(tn(4, w.D0 + o * sH, H, w.Gin + o * sG + H, GL, G->l0_w, XD, H, XD, G->l0_b));
  if (g_wgrad_order == 2) ZTRY(ih0());
  return 0;
}

// Side stream (lowest priority) + events for the overlapped weight-gradient GEMMs, one set per device.
struct SideStream { hipStream_t s; hipEvent_t chunk, done; };
int side_stream(SideStream** out) {
  static SideStream pool[16];
  static bool ready[16] = {};
  int dev = 0;
  ZCHECK(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16, "decoder bwd: unsupported device index");
  if (!ready[dev]) {
    int lo = 0, hi = 0;
    ZCHECK(hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess, "hipDeviceGetStreamPriorityRange failed");
    ZCHECK(hipStreamCreateWithPriority(&pool[dev].s, hipStreamNonBlocking, lo) == hipSuccess, "side stream creation failed");
    ZCHECK(hipEventCreateWithFlags(&pool[dev].chunk, hipEventDisableTiming) == hipSuccess, "event creation failed");
    ZCHECK(hipEventCreateWithFlags(&pool[dev].done, hipEventDisableTiming) == hipSuccess, "event creation failed");
    ready[dev] = true;
  }
  *out = &pool[dev];
  return 0;
}

// fork event of a deferred-GEMM hand-over: stream-ordered use only (record on one stream, wait on another, both enqueued
// before this returns), so one event per device and hand-over point (`which`) is enough; events carry no data and are created once
int fork_event(hipEvent_t* out, int which = 0) {
  static hipEvent_t pool[16][2];
  static bool ready[16][2] = {};
  int dev = 0;
  ZCHECK(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16, "decoder bwd: unsupported device index");
  if (!ready[dev][which]) {
    ZCHECK(hipEventCreateWithFlags(&pool[dev][which], hipEventDisableTiming) == hipSuccess, "event creation failed");
    ready[dev][which] = true;
  }
  *out = pool[dev][which];
  return 0;
}

inline dim3 g1(long n) { long g = (n + 255) / 256; return dim3((unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g))); }

}  // namespace

extern "C" size_t zeggs_decoder_workspace_bytes(const ZeggsDecDims* d, int training) {
  Arena a(nullptr, 0);
  carve_dec(*d, training, a);
  return a.off + 256;
}

static int decoder_fwd_impl(const ZeggsDecDims* dp, const ZeggsDecParams* P, const ZeggsDecStats* st,
                            const float* pose0, const float* rpos0, const float* rrot0, const float* gaze,
                            const float* speech, const float* style, float* pose, float* rpos, float* rrot,
                            int training, const float* h_in, float* h_out, void* ws, size_t ws_bytes, void* stream,
                            const ZeggsDecCall* call);

extern "C" int zeggs_decoder_fwd(const ZeggsDecDims* dp, const ZeggsDecParams* P, const ZeggsDecStats* st,
                                 const float* pose0, const float* rpos0, const float* rrot0, const float* gaze,
                                 const float* speech, const float* style, float* pose, float* rpos, float* rrot,
                                 int training, void* ws, size_t ws_bytes, void* stream) {
  return decoder_fwd_impl(dp, P, st, pose0, rpos0, rrot0, gaze, speech, style, pose, rpos, rrot, training, nullptr,
                          nullptr, ws, ws_bytes, stream, nullptr);
}
extern "C" int zeggs_decoder_fwd_ex(const ZeggsDecDims* dp, const ZeggsDecParams* P, const ZeggsDecStats* st,
                                    const float* pose0, const float* rpos0, const float* rrot0, const float* gaze,
                                    const float* speech, const float* style, float* pose, float* rpos, float* rrot,
                                    int training, void* ws, size_t ws_bytes, void* stream, const ZeggsDecCall* call) {
  return decoder_fwd_impl(dp, P, st, pose0, rpos0, rrot0, gaze, speech, style, pose, rpos, rrot, training, nullptr,
                          nullptr, ws, ws_bytes, stream, call);
}

// Chunked (streaming) decode: frame 0 of the chunk is the last frame already produced (its pose / root state come in as
// pose0 / rpos0 / rrot0), h_in [2,B,H] is the GRU state after that frame (NULL: first chunk, CellStateEncoder), h_out
// receives the state after the chunk's last frame.  Inference only (2-slot rings).
extern "C" int zeggs_decoder_fwd_state(const ZeggsDecDims* dp, const ZeggsDecParams* P, const ZeggsDecStats* st,
                                       const float* pose0, const float* rpos0, const float* rrot0, const float* gaze,
                                       const float* speech, const float* style, float* pose, float* rpos, float* rrot,
                                       const float* h_in, float* h_out, void* ws, size_t ws_bytes, void* stream) {
  return decoder_fwd_impl(dp, P, st, pose0, rpos0, rrot0, gaze, speech, style, pose, rpos, rrot, 0, h_in, h_out, ws,
                          ws_bytes, stream, nullptr);
}
// the same with per-call controls: ZeggsDecCall.status receives the give-up bit of the B = 1 persistent kernel, so that a caller
// that feeds the returned state into the NEXT chunk (zeggs/stream.py) can notice a rollout that did not complete and redo it
extern "C" int zeggs_decoder_fwd_state_ex(const ZeggsDecDims* dp, const ZeggsDecParams* P, const ZeggsDecStats* st,
                                          const float* pose0, const float* rpos0, const float* rrot0, const float* gaze,
                                          const float* speech, const float* style, float* pose, float* rpos, float* rrot,
                                          const float* h_in, float* h_out, void* ws, size_t ws_bytes, void* stream,
                                          const ZeggsDecCall* call) {
  return decoder_fwd_impl(dp, P, st, pose0, rpos0, rrot0, gaze, speech, style, pose, rpos, rrot, 0, h_in, h_out, ws,
                          ws_bytes, stream, call);
}

static int decoder_fwd_impl(const ZeggsDecDims* dp, const ZeggsDecParams* P, const ZeggsDecStats* st,
                            const float* pose0, const float* rpos0, const float* rrot0, const float* gaze,
                            const float* speech, const float* style, float* pose, float* rpos, float* rrot,
                            int training, const float* h_in, float* h_out, void* ws, size_t ws_bytes, void* stream,
                            const ZeggsDecCall* call) {
  const ZeggsDecDims& d = *dp;
  hipStream_t s = (hipStream_t)stream;
  const bool fwd_prepared = call && (call->prepared & 1);
  unsigned* status = call ? call->status : nullptr;
  ZCHECK(d.PI == d.PO + 3, "decoder: pose_input_size must be pose_output_size + 3 (gaze)");
  ZCHECK(d.B >= 1 && d.T >= 1, "decoder: empty batch or sequence");
  Arena a(ws, ws_bytes);
  DecWs w = carve_dec(d, training, a);
  ZCHECK(a.ok(), "decoder: workspace too small (%zu < %zu)", ws_bytes, a.off);
  const int B = d.B, T = d.T, H = d.H, GL = w.GL, XD = w.XD, CI = d.PI + d.ST;
  const long sG = (long)B * GL, sH = (long)B * H;
  const int ring = training ? 0 : 1;
  auto slot = [&](int t) { return ring ? (t & 1) : t; };
  if (!training) ZTRY(k_fill(w.Gin, 2 * sG, 0.f, s));   // ring slots: pad columns must be finite (GEMV decode path)
  const bool fast = g_decoder_fast && dec_fast_supported(d);
  // ---- training, batch <= 64: will the forward rollout run as one persistent launch (train_persistent.hip)?  Then its whole prologue is
  // five launches instead of ten (round 6; option "tp_prologue", default on): [dec_init | dec_fill_cond | tp_cond] in one,
  // [CellStateEncoder layer 0 | hid_1 | the step-1 pose product] in one, CellStateEncoder layer 1, the two halves of its last layer
  // in one, and the rollout's own operand fragments (dec_tp_run) -- every one of them was launch latency on an idle chip
  bool tp_path = false;
  if (fast && training && g_train_persistent && dec_tp_state() != 0 && dec_tp_supported(d, w)) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) cap = hipStreamCaptureStatusActive;   // a failed query must not read as "not capturing"
    tp_path = cap == hipStreamCaptureStatusNone || dec_tp_state() == 1;
  }
  const bool tp_pro = tp_path && g_tp_prologue && !h_in && T > 1 && !d.film;
  if (tp_pro) {
    float* gin1 = w.Gin + sG;
    ZTRY(dec_tp_prologue(d, st, w, pose0, rpos0, rrot0, gaze, speech, style, pose, rpos, rrot, s, fwd_prepared));
    const GemmNtItem l1[3] = {{w.cse_in, CI, P->c0_w, CI, w.cse_a, H, P->c0_b, H, CI, ACT_ELU},
                              {gin1 + H, GL, P->l0_w, XD, gin1, GL, P->l0_b, H, XD, ACT_ELU},          // hid_1 = ELU(W0 x_1 + b0)
                              dec_tp_p1x_item(d, P, w)};
    ZTRY(gemm_nt_multi(l1, 3, B, s));
    ZTRY(gemm_nt(w.cse_a, H, P->c1_w, H, w.cse_b, H, P->c1_b, B, H, H, ACT_ELU, 0.f, s));
    const GemmNtItem l3[2] = {{w.cse_b, H, P->c2_w, H, w.H0, H, P->c2_b, H, H, ACT_NONE},
                              {w.cse_b, H, P->c2_w + (long)H * H, H, w.H1, H, P->c2_b + H, H, H, ACT_NONE}};
    ZTRY(gemm_nt_multi(l3, 2, B, s));
  } else {
  // frame 0 + CellStateEncoder
  hipLaunchKernelGGL(dec_init_k, dim3(B), dim3(256), 0, s, d, *st, pose0, rpos0, rrot0, gaze, style, pose, rpos, rrot,
                     w.cse_in, w.Gin + slot(1) * sG, GL);
  ZLAUNCH_CHECK("dec_init");
  if (h_in) {   // resumed rollout: the recurrent state is given
    ZTRY(k_copy(w.H0 + slot(0) * sH, h_in, sH, s));
    ZTRY(k_copy(w.H1 + slot(0) * sH, h_in + sH, sH, s));
  } else {
    ZTRY(gemm_nt(w.cse_in, CI, P->c0_w, CI, w.cse_a, H, P->c0_b, B, H, CI, ACT_ELU, 0.f, s));
    ZTRY(gemm_nt(w.cse_a, H, P->c1_w, H, w.cse_b, H, P->c1_b, B, H, H, ACT_ELU, 0.f, s));
    ZTRY(gemm_nt(w.cse_b, H, P->c2_w, H, w.H0 + slot(0) * sH, H, P->c2_b, B, H, H, ACT_NONE, 0.f, s));
    ZTRY(gemm_nt(w.cse_b, H, P->c2_w + (long)H * H, H, w.H1 + slot(0) * sH, H, P->c2_b + H, B, H, H, ACT_NONE, 0.f, s));
  }
  }
  auto save_state = [&]() -> int {
    if (h_out) {
      ZTRY(k_copy(h_out, w.H0 + slot(T - 1) * sH, sH, s));
      ZTRY(k_copy(h_out + sH, w.H1 + slot(T - 1) * sH, sH, s));
    }
    return 0;
  };
  if (training && T > 1) {
    if (!tp_pro) {
      hipLaunchKernelGGL(dec_fill_cond_k, g1((long)(T - 1) * B * (d.SP + d.ST)), dim3(256), 0, s, d, speech, style, w.Gin,
                         GL, 1, T - 1, sG, 0);
      ZLAUNCH_CHECK("dec_fill_cond");
    }
    if (d.film) {   // modulation vectors of every step in two GEMMs over the time-major style
      ZCHECK(P->l3_w && P->l3_b && P->g_w && P->g_b && P->be_w && P->be_b, "decoder: film parameters missing");
      hipLaunchKernelGGL(style_time_major_k, g1((long)T * B * d.ST), dim3(256), 0, s, d, style, w.STm);
      ZLAUNCH_CHECK("style_time_major");
      ZTRY(gemm_nt(w.STm, d.ST, P->g_w, d.ST, w.GAM, 2 * H, P->g_b, T * B, 2 * H, d.ST, ACT_NONE, 0.f, s));
      ZTRY(gemm_nt(w.STm, d.ST, P->be_w, d.ST, w.BET, 2 * H, P->be_b, T * B, 2 * H, d.ST, ACT_NONE, 0.f, s));
    }
  }
  // ---- batch-1 inference: the weight-stationary persistent kernel (one launch for all frames, decode_persistent.hip)
  if (fast && !training && g_persistent && dec_persistent_state() != 0 && dec_persistent_supported(d, w)) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) cap = hipStreamCaptureStatusActive;   // a failed query must not read as "not capturing"
    // the first use on a process is validated (device sync + error word); never inside a stream capture
    if (cap == hipStreamCaptureStatusNone || dec_persistent_state() == 1) {
      float* gin1 = w.Gin + slot(1) * sG;
      hipLaunchKernelGGL(dec_fill_cond_k, g1((long)B * (d.SP + d.ST)), dim3(256), 0, s, d, speech, style, w.Gin, GL, 1, 1,
                         sG, 1);
      ZLAUNCH_CHECK("dec_fill_cond");
      ZTRY(gemm_nt(gin1 + H, GL, P->l0_w, XD, gin1, GL, P->l0_b, B, H, XD, ACT_ELU, 0.f, s));   // hid_1 = ELU(W0 x_1 + b0)
      ZTRY(dec_fast_merge_prep(d, P, st, w, s));
      dec_timing_mark(0, s);
      ZTRY(dec_persistent_run(d, P, st, w, gaze, speech, style, pose, rpos, rrot, gin1, w.H0 + slot(0) * sH,
                              w.H1 + slot(0) * sH, w.H0 + slot(T - 1) * sH, w.H1 + slot(T - 1) * sH, s,
                              dec_persistent_state() == 1 ? status : nullptr));
      dec_timing_mark(1, s);
      if (dec_persistent_state() == 1) return save_state();
      unsigned perr = 1;
      ZCHECK(hipStreamSynchronize(s) == hipSuccess, "persistent decode: stream sync failed");
      ZTRY(dec_persistent_errors(w, &perr));
      dec_persistent_set_state(perr == 0 ? 1 : 0);
      if (perr == 0) return save_state();
      // a bounded sweep gave up (not every workgroup resident?): disabled for this process, the stage kernels redo the rollout
    }
  }
  // ---- training, batch <= 32: the forward rollout as one persistent launch (train_persistent.hip)
  if (tp_path) {
    {
      float* gin1 = w.Gin + sG;
      if (!tp_pro) ZTRY(gemm_nt(gin1 + H, GL, P->l0_w, XD, gin1, GL, P->l0_b, B, H, XD, ACT_ELU, 0.f, s));   // hid_1 = ELU(W0 x_1 + b0)
      if (!fwd_prepared) {
        ZTRY(dec_fast_merge_prep(d, P, st, w, s));
        ZTRY(dec_tp_pack(d, P, st, w, s));
      }
      dec_timing_mark(0, s);
      // (the first, validated use reports through the workspace's own error word: a give-up there is handled right below)
      ZTRY(dec_tp_run(d, P, st, w, gaze, speech, style, pose, rpos, rrot, s, fwd_prepared,
                      dec_tp_state() == 1 ? status : nullptr, tp_pro));
      dec_timing_mark(1, s);
      if (dec_tp_state() == 1) return save_state();
      unsigned perr = 1;
      ZCHECK(hipStreamSynchronize(s) == hipSuccess, "persistent training rollout: stream sync failed");
      ZTRY(dec_tp_errors(w, &perr));
      dec_tp_set_state(perr == 0 ? 1 : 0);
      if (perr == 0) return save_state();
    }
  }
  if (fast) {
    if (!training && T > 1) {
      hipLaunchKernelGGL(dec_fill_cond_k, g1((long)B * (d.SP + d.ST)), dim3(256), 0, s, d, speech, style, w.Gin, GL, 1,
                         1, sG, 1);
      ZLAUNCH_CHECK("dec_fill_cond");
    }
    ZTRY(dec_fast_pack_fwd(d, P, w, s));
    ZTRY(dec_fast_fwd_steps(d, P, st, w, gaze, speech, style, pose, rpos, rrot, training, s));
    return save_state();
  }
  for (int t = 1; t < T; ++t) {
    float* gin = w.Gin + slot(t) * sG;
    float* gin_next = (t + 1 < T) ? w.Gin + slot(t + 1) * sG : nullptr;
    const float *h0p = w.H0 + slot(t - 1) * sH, *h1p = w.H1 + slot(t - 1) * sH;
    float *h0 = w.H0 + slot(t) * sH, *h1 = w.H1 + slot(t) * sH;
    if (!training) {
      hipLaunchKernelGGL(dec_fill_cond_k, g1((long)B * (d.SP + d.ST)), dim3(256), 0, s, d, speech, style, w.Gin, GL, t,
                         1, sG, 1);
      ZLAUNCH_CHECK("dec_fill_cond");
    }
    // hid = ELU(layer0(x))   [film: modulated by the style]
    const float *gam = nullptr, *bet = nullptr;
    if (d.film) {
      if (!training) {   // ring path: this step's modulation vectors from style[:, t]
        ZCHECK(P->l3_w && P->l3_b && P->g_w && P->g_b && P->be_w && P->be_b, "decoder: film parameters missing");
        ZTRY(gemm_nt(style + (long)t * d.ST, (long)T * d.ST, P->g_w, d.ST, w.GAM, 2 * H, P->g_b, B, 2 * H, d.ST, ACT_NONE,
                     0.f, s));
        ZTRY(gemm_nt(style + (long)t * d.ST, (long)T * d.ST, P->be_w, d.ST, w.BET, 2 * H, P->be_b, B, 2 * H, d.ST, ACT_NONE,
                     0.f, s));
      }
      gam = w.GAM + (training ? (long)t * B * 2 * H : 0);
      bet = w.BET + (training ? (long)t * B * 2 * H : 0);
      float* a0 = w.A0 + slot(t) * sH;
      ZTRY(gemm_nt(gin + H, GL, P->l0_w, XD, a0, H, P->l0_b, B, H, XD, ACT_ELU, 0.f, s));
      hipLaunchKernelGGL(film_fwd_k, g1(sH), dim3(256), 0, s, gin, (long)GL, a0, gam, bet, (long)2 * H, B, H);
    } else {
      ZTRY(gemm_nt(gin + H, GL, P->l0_w, XD, gin, GL, P->l0_b, B, H, XD, ACT_ELU, 0.f, s));
    }
    // GRU layer 0
    ZTRY(gemm_nt(gin, GL, P->w_ih0, H + XD, w.gi, 3 * H, P->b_ih0, B, 3 * H, H + XD, ACT_NONE, 0.f, s));
    ZTRY(gemm_nt(h0p, H, P->w_hh0, H, w.gh, 3 * H, P->b_hh0, B, 3 * H, H, ACT_NONE, 0.f, s));
    const long o = (long)t * sH;
    hipLaunchKernelGGL(gru_gate_fwd_k, g1(sH), dim3(256), 0, s, w.gi, w.gh, h0p, h0,
                       training ? (f4*)w.GT0 + o : (f4*)nullptr, B, H);
    // GRU layer 1
    ZTRY(gemm_nt(h0, H, P->w_ih1, H, w.gi, 3 * H, P->b_ih1, B, 3 * H, H, ACT_NONE, 0.f, s));
    ZTRY(gemm_nt(h1p, H, P->w_hh1, H, w.gh, 3 * H, P->b_hh1, B, 3 * H, H, ACT_NONE, 0.f, s));
    hipLaunchKernelGGL(gru_gate_fwd_k, g1(sH), dim3(256), 0, s, w.gi, w.gh, h1p, h1,
                       training ? (f4*)w.GT1 + o : (f4*)nullptr, B, H);
    // output projection + pose integration
    if (d.film) {
      float *a2 = w.A2 + slot(t) * sH, *f2 = w.F2 + slot(t) * sH;
      ZTRY(gemm_nt(h1, H, P->l2_w, H, a2, H, P->l2_b, B, H, H, ACT_ELU, 0.f, s));
      hipLaunchKernelGGL(film_fwd_k, g1(sH), dim3(256), 0, s, f2, (long)H, a2, gam + H, bet + H, (long)2 * H, B, H);
      ZTRY(gemm_nt(f2, H, P->l3_w, H, w.Y, w.POL, P->l3_b, B, d.PO, H, ACT_NONE, 0.f, s));
    } else {
      ZTRY(gemm_nt(h1, H, P->l2_w, H, w.Y, w.POL, P->l2_b, B, d.PO, H, ACT_NONE, 0.f, s));
    }
    hipLaunchKernelGGL(dec_devec_k, dim3(B), dim3(256), 0, s, d, *st, w.Y, w.POL, gaze, pose, rpos, rrot, gin_next, GL,
                       t);
    ZLAUNCH_CHECK("dec_step");
  }
  return save_state();
}

// error word of the chained (run-ahead) stage launches of the last rollout that used `ws`: 0 = every hand-off wait was
// satisfied; non-zero = a bounded spin gave up (the results of that rollout are invalid).  Synchronises the device.
extern "C" int zeggs_decoder_chain_errors(const ZeggsDecDims* dp, int training, void* ws, size_t ws_bytes, int* out) {
  Arena a(ws, ws_bytes);
  DecWs w = carve_dec(*dp, training, a);
  ZCHECK(a.ok() && w.chain, "chain_errors: workspace too small or no fast path for these dims");
  unsigned v = 0;
  ZCHECK(hipMemcpy(&v, w.chain + 4 * 8 * 32, sizeof(v), hipMemcpyDeviceToHost) == hipSuccess, "chain_errors: copy failed");
  *out = (int)v;
  return 0;
}
// measurement builds (-DZEGGS_CHTIME): the phase stamps of the last 16 chained launches, 16 launches x {first, last
// workgroup} x 16 stamps (100 MHz wall clock)
extern "C" int zeggs_decoder_chain_stamps(const ZeggsDecDims* dp, int training, void* ws, size_t ws_bytes,
                                          unsigned long long* out /* [16][2][16] host */) {
  Arena a(ws, ws_bytes);
  DecWs w = carve_dec(*dp, training, a);
  ZCHECK(a.ok() && w.chain, "chain_stamps: workspace too small or no fast path for these dims");
  ZCHECK(hipMemcpy(out, w.chain + 4 * 8 * 32 + 32, 16 * 2 * 16 * 8, hipMemcpyDeviceToHost) == hipSuccess, "copy failed");
  return 0;
}

// Second half of the deferred weight gradients (option "defer_wgrads" = 2): the GEMMs of GRU layer 0 and layer0 (what = 4) on the
// caller's stream -- normally zeggs_side_stream again, behind the first half and the all-reduce the caller has started on it.
extern "C" int zeggs_decoder_wgrads(const ZeggsDecDims* dp, const ZeggsDecGrads* G, void* ws, size_t ws_bytes, int what,
                                    void* stream) {
  const ZeggsDecDims& d = *dp;
  Arena a(ws, ws_bytes);
  DecWs w = carve_dec(d, 1, a);
  ZCHECK(a.ok(), "decoder wgrads: workspace too small (was the forward run with training=1?)");
  ZCHECK(d.T > 1, "decoder wgrads: T must be > 1");
  const bool fast_path = g_decoder_fast && dec_fast_supported(d);
  // what & 8: the gradient outputs are zero on entry (ZeggsDecCall.grads_zeroed of the backward this call completes)
  return dec_recurrent_wgrads(d, w, G, 1, d.T - 1, (what & 8) ? 1.f : 0.f, fast_path ? 1 : 0, (hipStream_t)stream, what & 5);
}
// Everything the two persistent sweeps of a training step need that depends on the WEIGHTS only (the merged / folded
// matrices, the per-workgroup fragment packs of both kernels): the caller may run it on a second stream beside the encoders'
// forward and passes ZeggsDecCall.prepared to the zeggs_decoder_fwd_ex / _bwd_ex calls that follow on the same workspace.  Returns a bit mask
// (1: forward packs ready, 2: backward packs ready; 0: these dimensions take another path, nothing was done), < 0 on error.
extern "C" int zeggs_decoder_prepare(const ZeggsDecDims* dp, const ZeggsDecParams* P, const ZeggsDecStats* st, void* ws,
                                     size_t ws_bytes, void* stream) {
  const ZeggsDecDims& d = *dp;
  hipStream_t s = (hipStream_t)stream;
  Arena a(ws, ws_bytes);
  DecWs w = carve_dec(d, 1, a);
  ZCHECK(a.ok(), "decoder prepare: workspace too small");
  const bool fast = g_decoder_fast && dec_fast_supported(d);
  // only once the kernels have been validated on this process (the first use takes the ordinary path)
  if (!(fast && d.T > 1 && g_train_persistent && dec_tp_state() == 1 && dec_tp_supported(d, w))) return 0;
  ZTRY(dec_fast_merge_prep(d, P, st, w, s));
  ZTRY(dec_tp_pack(d, P, st, w, s));
  ZTRY(dec_tp_zero(d, w, s));
  int mask = 1;
  if (g_bwd_persistent && dec_bp_state() == 1 && dec_bp_supported(d, w)) {
    ZTRY(dec_bp_pack(d, P, w, s));
    // ... and the zero state the backward starts from (carries, frame-0 slot of DX, arrival slots + error word)
    ZTRY(k_fill(w.dH0c, (long)d.B * d.H, 0.f, s));
    ZTRY(k_fill(w.dH1c, (long)d.B * d.H, 0.f, s));
    ZTRY(k_fill(w.carry, (long)2 * d.B * 8, 0.f, s));
    ZTRY(k_fill(w.DX, (long)d.B * w.XD, 0.f, s));
    ZTRY(dec_bp_zero_slots(w, s));
    mask |= 2;
  }
  return mask;
}
// the library's low-priority second stream of the current device (also used by the chunked stage-launch sweep)
extern "C" int zeggs_side_stream(void** out) {
  SideStream* ss = nullptr;
  ZTRY(side_stream(&ss));
  *out = (void*)ss->s;
  return 0;
}

extern "C" int zeggs_decoder_bwd(const ZeggsDecDims* dp, const ZeggsDecParams* P, const ZeggsDecStats* st,
                                 const float* gaze, const float* pose, const float* rpos, const float* rrot,
                                 const float* dpose, const float* drpos, const float* drrot, const ZeggsDecGrads* G,
                                 float* dspeech, float* dstyle, void* ws, size_t ws_bytes, void* stream) {
  return zeggs_decoder_bwd_ex(dp, P, st, gaze, pose, rpos, rrot, dpose, drpos, drrot, G, dspeech, dstyle, ws, ws_bytes, stream,
                              nullptr);
}
extern "C" int zeggs_decoder_bwd_ex(const ZeggsDecDims* dp, const ZeggsDecParams* P, const ZeggsDecStats* st,
                                    const float* gaze, const float* pose, const float* rpos, const float* rrot,
                                    const float* dpose, const float* drpos, const float* drrot, const ZeggsDecGrads* G,
                                    float* dspeech, float* dstyle, void* ws, size_t ws_bytes, void* stream,
                                    const ZeggsDecCall* call) {
  const ZeggsDecDims& d = *dp;
  hipStream_t s = (hipStream_t)stream;
  const bool bwd_prepared = call && (call->prepared & 2);
  const int defer_wgrads = call ? call->defer_wgrads : 0;
  unsigned* status = call ? call->status : nullptr;
  const float gb = (call && call->grads_zeroed) ? 1.f : 0.f;      // gradient outputs are zero on entry: accumulate, no fills
  ZCHECK(defer_wgrads == 0 || call->wgrad_stream != nullptr, "decoder bwd: defer_wgrads needs ZeggsDecCall.wgrad_stream");
  Arena a(ws, ws_bytes);
  DecWs w = carve_dec(d, 1, a);
  ZCHECK(a.ok(), "decoder bwd: workspace too small (was the forward run with training=1?)");
  const int B = d.B, T = d.T, H = d.H, GL = w.GL, XD = w.XD, CI = d.PI + d.ST, POL = w.POL;
  const long sG = (long)B * GL, sH = (long)B * H, s3 = 3 * sH;
  if (!bwd_prepared) {      // (zeggs_decoder_prepare has done it)
    ZTRY(k_fill(w.dH0c, sH, 0.f, s));
    ZTRY(k_fill(w.dH1c, sH, 0.f, s));
    ZTRY(k_fill(w.carry, (long)2 * B * 8, 0.f, s));
    ZTRY(k_fill(w.DX, (long)B * XD, 0.f, s));          // slot t = 0 unused but read by the scatter
  }
  ZCHECK(T > 1, "decoder bwd: T must be > 1");
  bool wgrads_done = false;
  SideStream* ss = nullptr;
  const bool fast_path = g_decoder_fast && dec_fast_supported(d);
  // ---- batch <= 32: the whole sweep as one persistent launch (train_bwd_persistent.hip)
  bool swept = false;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess) cap = hipStreamCaptureStatusActive;   // a failed query must not read as "not capturing"
  if (fast_path && g_bwd_persistent && dec_bp_state() != 0 && dec_bp_supported(d, w) &&
      (cap == hipStreamCaptureStatusNone || dec_bp_state() == 1)) {
    ZTRY(dec_bp_run(d, P, st, w, gaze, pose, rpos, rrot, dpose, drpos, drrot, s, bwd_prepared,
                    dec_bp_state() == 1 ? status : nullptr));
    if (dec_bp_state() == 1) {
      swept = true;
    } else {      // first use on this process: validate (a bounded wait that gave up means not every workgroup was resident)
      unsigned perr = 1;
      ZCHECK(hipStreamSynchronize(s) == hipSuccess, "persistent BPTT sweep: stream sync failed");
      ZTRY(dec_bp_errors(w, &perr));
      dec_bp_set_state(perr == 0 ? 1 : 0);
      swept = perr == 0;
      if (!swept) {   // the stage kernels redo the sweep from clean carries
        ZTRY(k_fill(w.dH0c, sH, 0.f, s));
        ZTRY(k_fill(w.dH1c, sH, 0.f, s));
        ZTRY(k_fill(w.carry, (long)2 * B * 8, 0.f, s));
      }
    }
  }
  if (swept) {
    // nothing left to do here: DY, DI*, DH*, D0, DX and the final carries are in place
  } else if (fast_path) {
    ZTRY(dec_fast_pack_bwd(d, P, st, w, s));
    // The sweep is a chain of small dependent launches that leaves most of the chip idle; the weight-gradient
    // GEMMs of the steps already swept run beside it on a low-priority stream, chunk by chunk.
    const int nch = g_bwd_chunks < T - 1 ? g_bwd_chunks : T - 1;
    if (nch > 1) ZTRY(side_stream(&ss));
    for (int c = 0; c < nch; ++c) {
      const int t_hi = T - 1 - (int)((long)(T - 1) * c / nch), t_lo = T - (int)((long)(T - 1) * (c + 1) / nch);
      ZTRY(dec_fast_bwd_steps(d, P, st, w, gaze, pose, rpos, rrot, dpose, drpos, drrot, t_hi, t_lo, s));
      if (nch > 1) {
        ZCHECK(hipEventRecord(ss->chunk, s) == hipSuccess, "hipEventRecord failed");
        ZCHECK(hipStreamWaitEvent(ss->s, ss->chunk, 0) == hipSuccess, "hipStreamWaitEvent failed");
        ZTRY(dec_recurrent_wgrads(d, w, G, t_lo, t_hi, c == 0 ? gb : 1.f, 1, ss->s));
      }
    }
    if (nch > 1) {
      ZCHECK(hipEventRecord(ss->done, ss->s) == hipSuccess, "hipEventRecord failed");
      wgrads_done = true;
    }
  } else {
  for (int t = T - 1; t >= 1; --t) {
    const float* gin = w.Gin + t * sG;
    const long o = (long)t * sH;
    float* dy = w.DY + (long)t * B * POL;
    const float* dxn = (t + 1 < T) ? w.DX + (long)(t + 1) * B * XD : nullptr;
    hipLaunchKernelGGL(dec_devec_bwd_k, dim3(B), dim3(256), 0, s, d, *st, dpose, drpos, drrot, dxn, XD, gaze, pose, rpos,
                       rrot, w.carry, dy, POL, t);
    ZLAUNCH_CHECK("dec_devec_bwd");
    // dH1 total = dy W2 + carried   [film: through layer3, the modulation and layer2]
    if (d.film) {
      const long og = (long)t * B * 2 * H;
      ZTRY(gemm_nn(dy, POL, P->l3_w, H, w.dF2, H, B, d.PO, H, 0.f, s));
      hipLaunchKernelGGL(film_bwd_k, g1(sH), dim3(256), 0, s, w.dF2, (long)H, w.A2 + o, w.GAM + og + H, (long)2 * H,
                         w.D2 + o, w.DGAM + og + H, w.DBET + og + H, B, H);
      ZTRY(gemm_nn(w.D2 + o, H, P->l2_w, H, w.dH1c, H, B, H, H, 1.f, s));
    } else {
      ZTRY(gemm_nn(dy, POL, P->l2_w, H, w.dH1c, H, B, d.PO, H, 1.f, s));
    }
    hipLaunchKernelGGL(gru_gate_bwd_k, g1(sH), dim3(256), 0, s, w.dH1c, (const f4*)w.GT1 + o,
                       w.H1 + o - sH, w.DI1 + t * s3, w.DH1 + t * s3, w.t0, B, H);
    // t0 = dH1 * z (direct path); dH1c <- t0 + DH1 W_hh1 ; dH0 total = dH0c + DI1 W_ih1
    ZTRY(k_copy(w.dH1c, w.t0, sH, s));
    ZTRY(gemm_nn(w.DH1 + t * s3, 3 * H, P->w_hh1, H, w.dH1c, H, B, 3 * H, H, 1.f, s));
    ZTRY(gemm_nn(w.DI1 + t * s3, 3 * H, P->w_ih1, H, w.dH0c, H, B, 3 * H, H, 1.f, s));
    hipLaunchKernelGGL(gru_gate_bwd_k, g1(sH), dim3(256), 0, s, w.dH0c, (const f4*)w.GT0 + o,
                       w.H0 + o - sH, w.DI0 + t * s3, w.DH0 + t * s3, w.t0, B, H);
    ZTRY(k_copy(w.dH0c, w.t0, sH, s));
    ZTRY(gemm_nn(w.DH0 + t * s3, 3 * H, P->w_hh0, H, w.dH0c, H, B, 3 * H, H, 1.f, s));
    // dGin = DI0 W_ih0 -> [dhid | dx]
    ZTRY(gemm_nn(w.DI0 + t * s3, 3 * H, P->w_ih0, H + XD, w.dGin, GL, B, 3 * H, H + XD, 0.f, s));
    if (d.film) {
      const long og = (long)t * B * 2 * H;
      hipLaunchKernelGGL(film_bwd_k, g1(sH), dim3(256), 0, s, w.dGin, (long)GL, w.A0 + o, w.GAM + og, (long)2 * H,
                         w.D0 + o, w.DGAM + og, w.DBET + og, B, H);
    } else {
      hipLaunchKernelGGL(elu_bwd_rows_k, g1(sH), dim3(256), 0, s, w.D0 + o, w.dGin, gin, B, H, GL);
    }
    // dx_t = dGin[:, H:] + D0 W0
    float* dx = w.DX + (long)t * B * XD;
    hipLaunchKernelGGL(copy_cols_k, g1((long)B * XD), dim3(256), 0, s, dx, (long)XD, w.dGin, (long)GL, H, XD, B);
    ZLAUNCH_CHECK("dec_bwd_step");
    ZTRY(gemm_nn(w.D0 + o, H, P->l0_w, XD, dx, XD, B, H, XD, 1.f, s));
  }
  }
  // ---- weight gradients of the recurrent layers.  Option "defer_wgrads": the seven large GEMMs (K = B (T-1), they read only
  // what the sweep saved) start NOW on the library's second stream, beside the CellStateEncoder backward below and whatever the
  // caller enqueues on `s` after this call (the encoders' backward).  No join here: the caller makes every consumer of the
  // decoder gradients wait for zeggs_side_stream.
  if (!wgrads_done && defer_wgrads && cap == hipStreamCaptureStatusNone) {
    hipStream_t gs = (hipStream_t)call->wgrad_stream;
    hipEvent_t fork = nullptr;
    ZTRY(fork_event(&fork));
    ZCHECK(hipEventRecord(fork, s) == hipSuccess, "hipEventRecord failed");
    ZCHECK(hipStreamWaitEvent(gs, fork, 0) == hipSuccess, "hipStreamWaitEvent failed");
    // (value 2: only the first half of the parameter order here -- the caller all-reduces it while zeggs_decoder_wgrads
    //  computes the second half)
    // The bias sums (a dozen column sums over the same saves) go with them: since the split-K retune the GEMMs are the shorter
    // of the two queues.
    ZTRY(dec_recurrent_wgrads(d, w, G, 1, T - 1, gb, fast_path ? 1 : 0, gs, (defer_wgrads == 2 ? 1 : 5) | 2));
  } else if (!wgrads_done) {
    ZTRY(dec_recurrent_wgrads(d, w, G, 1, T - 1, gb, fast_path ? 1 : 0, s));
  }
  // ---- CellStateEncoder backward: dH0c / dH1c are the grads wrt its two output halves.  The input-gradient chain (four
  // batch-sized products, the ELU' factors in their epilogues) is what the caller's next kernels wait for (dstyle -> the style
  // encoder's backward): it goes first; the weight / bias gradients need only its intermediates and, with deferred GEMMs, join
  // the recurrent layers' on the weight-gradient stream.
  {
    // out = [H0_init | H1_init] = cse_b W2^T + b2
    float* db = w.t0;                       // [B,H] grad wrt cse_b
    float* da = w.t0 + sH;                  // [B,H] grad wrt cse_a
    ZTRY(gemm_nn(w.dH0c, H, P->c2_w, H, db, H, B, H, H, 0.f, s));
    ZTRY(gemm_nn_actbwd(w.dH1c, H, P->c2_w + (long)H * H, H, db, H, B, H, H, 1.f, w.cse_b, H, ACT_ELU, s));
    ZTRY(gemm_nn_actbwd(db, H, P->c1_w, H, da, H, B, H, H, 0.f, w.cse_a, H, ACT_ELU, s));
    ZTRY(gemm_nn(da, H, P->c0_w, CI, w.t1, CI, B, H, CI, 0.f, s));   // t1 = d cse_in [B, PI+ST]
    hipStream_t ws_ = s;
    if (!wgrads_done && defer_wgrads && cap == hipStreamCaptureStatusNone) {
      ws_ = (hipStream_t)call->wgrad_stream;
      hipEvent_t fork = nullptr;
      ZTRY(fork_event(&fork, 1));
      ZCHECK(hipEventRecord(fork, s) == hipSuccess, "hipEventRecord failed");
      ZCHECK(hipStreamWaitEvent(ws_, fork, 0) == hipSuccess, "hipStreamWaitEvent failed");
    }
    ZTRY(gemm_tn(w.dH0c, H, w.cse_b, H, G->c2_w, H, B, H, H, gb, ws_));
    ZTRY(gemm_tn(w.dH1c, H, w.cse_b, H, G->c2_w + (long)H * H, H, B, H, H, gb, ws_));
    ZTRY(k_colsum(G->c2_b, w.dH0c, B, H, H, gb, ws_));
    ZTRY(k_colsum(G->c2_b + H, w.dH1c, B, H, H, gb, ws_));
    ZTRY(gemm_tn(db, H, w.cse_a, H, G->c1_w, H, B, H, H, gb, ws_));
    ZTRY(k_colsum(G->c1_b, db, B, H, H, gb, ws_));
    ZTRY(gemm_tn(da, H, w.cse_in, CI, G->c0_w, CI, B, H, CI, gb, ws_));
    ZTRY(k_colsum(G->c0_b, da, B, H, H, gb, ws_));
  }
  hipLaunchKernelGGL(dec_scatter_cond_grad_k, g1((long)T * B * (d.SP + d.ST)), dim3(256), 0, s, d, w.DX, XD, dspeech,
                     dstyle);
  if (d.film) {   // the style reaches the steps through the two predictors only
    const long M1 = (long)(T - 1) * B, sg = (long)B * 2 * H, sS = (long)B * d.ST;
    ZTRY(gemm_nn(w.DGAM + sg, 2 * H, P->g_w, d.ST, w.dSTm + sS, d.ST, (int)M1, 2 * H, d.ST, 0.f, s));
    ZTRY(gemm_nn(w.DBET + sg, 2 * H, P->be_w, d.ST, w.dSTm + sS, d.ST, (int)M1, 2 * H, d.ST, 1.f, s));
    hipLaunchKernelGGL(style_grad_from_time_major_k, g1((long)T * B * d.ST), dim3(256), 0, s, d, w.dSTm, dstyle);
  }
  hipLaunchKernelGGL(add_style0_grad_k, g1((long)B * d.ST), dim3(256), 0, s, d, w.t1, dstyle);
  ZLAUNCH_CHECK("dec_bwd_tail");
  if (wgrads_done) ZCHECK(hipStreamWaitEvent(s, ss->done, 0) == hipSuccess, "hipStreamWaitEvent failed");   // join
  return 0;
}
