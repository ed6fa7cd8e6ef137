// Workspace layout of the decoder (shared by decoder.hip and decoder_fast.hip).
#pragma once
#include <string.h>

#include "../../include/zeggs_hip.h"
#include "common.h"

struct DecWs {
  // canonical (row-major) activations, time-major [T][B][.] in training, 2-slot ring in inference
  float *Gin, *H0, *H1, *GT0, *GT1, *Y;   // GT*: saved gates of both GRU layers, [T][B][H] x float4 (r, z, n, W_hn h + b_hn)
  float *cse_in, *cse_a, *cse_b;          // cell-state encoder activations
  float *gi, *gh;                          // per-step gate pre-activations [B,3H]   (generic path)
  // backward
  float *DY, *DI0, *DH0, *DI1, *DH1, *D0, *DX;
  float *dH0c, *dH1c, *dGin, *dXn, *carry, *t0, *t1;
  int GL, XD, POL;
  // FiLM variant (d.film): pre-modulation activations, modulation vectors and their gradients, time-major
  float *A0, *A2, *F2, *GAM, *BET, *DGAM, *DBET, *D2, *dF2, *STm, *dSTm;
  // ---- fast path: fragment-packed weights (see decoder_fast.hip) and activations
  int NB, nT5, nTH, nTX, nTPO, nTGI, KBH, KBX, KBPO, KB3H;
  // forward packs; the packs that feed one stage group are contiguous PER TILE (pw_g0 = [ih0h | ih0x | hh0],
  // pw_g1 = [ih1 | hh1], pw_mc = [M | Wc | copy of layer2's tile 0]) so that a wave's share of the concatenated
  // contraction is one address range (chained GEMV launches); the named pointers are sub-ranges (Seg.tkb = TG*)
  float *pw_l0, *pw_ih0h, *pw_ih0x, *pw_hh0, *pw_ih1, *pw_hh1, *pw_l2;
  // FiLM decoder on the stage kernels: layer2 is [H,H] (pw_l2: nTH tiles), layer3 [PO,H] (pw_l3); transposed packs pb_l2
  // (V[U][k] = W2[k][U]) and pb_l3 (V[U][c] = W3[c][U]); fragments of F2 = FiLM(ELU(layer2)) and of D2 = grad wrt layer2's output
  float *pw_l3, *pb_l3, *F2xf, *D2xf;
  float *pw_g0, *pw_g1, *pw_mc, *pw_l2c;
  int TG0, TG1, TMC;
  float *pb_l2, *pb_ih1, *pb_hh1, *pb_ih0, *pb_hh0, *pb_l0, *pb_mt;      // backward (transposed) packs
  float *Xxf, *HIDxf, *H0xf, *H1xf;                                      // forward activation fragments (rings of 2)
  // merged layer2 -> layer0 stage: M = W0[:, :PO] diag(sigma_o / sigma_i) W2, Wc = W0[:, PI:], cvec = b0 + W0[:, :PO] v
  int KBC;
  float *W0s, *Mc, *vvec, *cvec, *pw_m, *pw_c, *CONDxf;
  float *DYxf, *DI1xf, *DH1xf, *DI0xf, *DH0xf, *D0xf, *Rxf, *dXa;         // backward fragments
  unsigned* chain;     // arrival counters + error word of the chained (run-ahead) launches; zeroed with the forward fragments
  void* pgran;         // exchange granules + error word of the persistent decode kernel (decode_persistent.hip)
  size_t pgran_bytes;
  // persistent training rollout (train_persistent.hip): per-workgroup fragment packs, write-once time-major operand
  // fragments [T][kb][NB][64][4] of the three phases, arrival counters + error word
  float *tp_w0, *tp_w1, *tp_w3, *G0xf, *G1xf, *G3xf;
  float *tp_n0s, *tp_n0, *tp_cv0, *tp_p1x;   // fold of GRU layer 0's pose columns onto h1_{t-1} (train_persistent.hip)
  unsigned* tp_cnt;
  // persistent BPTT sweep (train_bwd_persistent.hip): per-workgroup weight tiles (register part, LDS part), write-once
  // time-major operands dy / [DI1 | dn_h1] / [DI0 | dn_h0] / D0 in the 4x4x1 B layout, dXa of the root / gaze columns,
  // arrival slots + error word
  float *bp_wr, *bp_wl, *bp_opy, *bp_op1, *bp_op0, *bp_opd, *bp_sp;
  unsigned* bp_cnt;
  size_t xf_bytes_fwd, xf_bytes_bwd;
  float *xf_base_fwd, *xf_base_bwd;
};

inline int round4(int x) { return (x + 3) / 4 * 4; }
constexpr int FILM_GB = 32;     // frames per block of inference-time FiLM modulation vectors

inline DecWs carve_dec(const ZeggsDecDims& d, int training, Arena& a) {
  DecWs w;
  memset(&w, 0, sizeof(w));
  const long B = d.B, T = d.T, H = d.H;
  w.XD = d.PI + d.SP + (d.film ? 0 : d.ST);   // film: the style is not part of the step input
  w.GL = round4(d.H + w.XD);
  w.POL = round4(d.PO);
  const long TS = training ? T : 2;   // inference keeps a 2-slot ring for the step buffers
  w.Gin = a.f(TS * B * w.GL);
  w.H0 = a.f(TS * B * H); w.H1 = a.f(TS * B * H);
  w.Y = a.f(B * (long)w.POL);
  w.cse_in = a.f(B * (long)(d.PI + d.ST));
  w.cse_a = a.f(B * H); w.cse_b = a.f(B * H);
  w.gi = a.f(B * 3 * H); w.gh = a.f(B * 3 * H);
  if (training) {
    w.GT0 = a.f(T * B * H * 4); w.GT1 = a.f(T * B * H * 4);
    w.DY = a.f(T * B * (long)w.POL);
    w.DI0 = a.f(T * B * 3 * H); w.DH0 = a.f(T * B * 3 * H);
    w.DI1 = a.f(T * B * 3 * H); w.DH1 = a.f(T * B * 3 * H);
    w.D0 = a.f(T * B * H);
    w.DX = a.f(T * B * (long)w.XD);
    w.dH0c = a.f(B * H); w.dH1c = a.f(B * H);
    w.dGin = a.f(B * (long)w.GL);
    w.dXn = a.f(B * (long)w.XD);
    w.carry = a.f(2 * B * 8);   // double-buffered (merged backward stage: one group writes, the other reads)
    w.t0 = a.f(B * 2 * H); w.t1 = a.f(B * (long)(d.PI + d.ST + 2 * H));
  }
  if (d.film) {
    w.A0 = a.f(TS * B * H); w.A2 = a.f(TS * B * H); w.F2 = a.f(TS * B * H);
    // modulation vectors: every frame's in training; in inference a ring of 2 x FILM_GB frames, filled a block of FILM_GB frames at a
    // time by one batched GEMM (decoder_fast.hip; the generic path uses slot 0 only, per frame)
    const long TSG = training ? T : 2 * FILM_GB;
    w.GAM = a.f(TSG * B * 2 * H); w.BET = a.f(TSG * B * 2 * H);
    if (training) {
      w.DGAM = a.f(T * B * 2 * H); w.DBET = a.f(T * B * 2 * H); w.D2 = a.f(T * B * H); w.dF2 = a.f(B * H);
      w.STm = a.f(T * B * (long)d.ST); w.dSTm = a.f(T * B * (long)d.ST);
    }
  }
  // ---- fast path
  w.NB = (d.B + 15) / 16;
  w.nT5 = (d.H + 4) / 5; w.nTH = d.H / 16; w.nTX = (w.XD + 15) / 16; w.nTPO = (d.PO + 15) / 16;
  w.nTGI = w.nTH + w.nTX;
  w.KBH = d.H / 16; w.KBX = w.nTX; w.KBPO = w.nTPO; w.KB3H = 3 * d.H / 16;
  const long BLK = 256;   // floats per (tile, k-block) weight fragment = 64 lanes x 4
  w.KBC = (d.SP + (d.film ? 0 : d.ST) + 15) / 16;   // conditioning columns of x (film: speech only)
  w.TG0 = w.KBH + w.KBX + w.KBH; w.TG1 = 2 * w.KBH; w.TMC = w.KBH + w.KBC + w.KBH;
  w.pw_l0 = a.f((long)w.nTH * w.KBX * BLK);
  w.pw_g0 = a.f((long)w.nT5 * w.TG0 * BLK);
  w.pw_ih0h = w.pw_g0; w.pw_ih0x = w.pw_g0 + w.KBH * BLK; w.pw_hh0 = w.pw_g0 + (w.KBH + w.KBX) * BLK;
  w.pw_g1 = a.f((long)w.nT5 * w.TG1 * BLK);
  w.pw_ih1 = w.pw_g1; w.pw_hh1 = w.pw_g1 + w.KBH * BLK;
  w.pw_l2 = a.f((long)(d.film ? w.nTH : w.nTPO) * w.KBH * BLK);
  if (d.film) w.pw_l3 = a.f((long)w.nTPO * w.KBH * BLK);
  w.pw_mc = a.f((long)w.nTH * w.TMC * BLK);
  w.pw_m = w.pw_mc; w.pw_c = w.pw_mc + w.KBH * BLK; w.pw_l2c = w.pw_mc + (w.KBH + w.KBC) * BLK;
  w.W0s = a.f(H * (long)w.POL); w.Mc = a.f(H * H); w.vvec = a.f(w.POL); w.cvec = a.f(H);
  const long XB = 256L * w.NB;  // floats per k-block of an activation fragment
  {
    size_t o0 = a.off;
    w.Xxf = a.f(2 * w.KBX * XB); w.HIDxf = a.f(w.KBH * XB); w.H0xf = a.f(2 * w.KBH * XB); w.H1xf = a.f(2 * w.KBH * XB);
    w.CONDxf = a.f(w.KBC * XB);
    if (d.film) w.F2xf = a.f(w.KBH * XB);
    w.chain = (unsigned*)a.f(4096);
    w.pgran_bytes = 36864;
    w.pgran = a.raw(w.pgran_bytes);
    w.xf_base_fwd = w.Xxf;
    w.xf_bytes_fwd = a.off - align_up(o0, 256);
  }
  if (training) {
    w.pb_l2 = a.f((long)w.nTH * (d.film ? w.KBH : w.KBPO) * BLK);
    if (d.film) w.pb_l3 = a.f((long)w.nTH * w.KBPO * BLK);
    w.pb_ih1 = a.f((long)w.nTH * w.KB3H * BLK);
    w.pb_hh1 = a.f((long)w.nTH * w.KB3H * BLK);
    w.pb_ih0 = a.f((long)w.nTGI * w.KB3H * BLK);
    w.pb_hh0 = a.f((long)w.nTH * w.KB3H * BLK);
    w.pb_l0 = a.f((long)w.nTX * w.KBH * BLK);
    w.pb_mt = a.f((long)w.nTH * w.KBH * BLK);
    size_t o0 = a.off;
    w.DYxf = a.f(w.KBPO * XB); w.DI1xf = a.f(w.KB3H * XB); w.DH1xf = a.f(w.KB3H * XB);
    w.DI0xf = a.f(w.KB3H * XB); w.DH0xf = a.f(w.KB3H * XB); w.D0xf = a.f(w.KBH * XB); w.Rxf = a.f(w.KBPO * XB);
    if (d.film) w.D2xf = a.f(w.KBH * XB);
    w.xf_base_bwd = w.DYxf;
    w.xf_bytes_bwd = a.off - align_up(o0, 256);
    w.dXa = a.f(B * (long)w.XD);
    if (d.H == 1024 && d.B <= 64 && !d.film) {
      const long KB0 = 64 + w.KBX + 64, KB3 = 64 + w.KBC;
      w.tp_w0 = a.f(256L * 8 * 26 * BLK); w.tp_w1 = a.f(256L * 8 * 16 * BLK); w.tp_w3 = a.f(256L * 8 * 9 * BLK);   // [wg][wave][block]
      w.G0xf = a.f(T * KB0 * XB); w.G1xf = a.f(T * 128 * XB); w.G3xf = a.f(T * KB3 * XB);
      w.tp_n0s = a.f(3 * H * (long)w.POL); w.tp_n0 = a.f(3 * H * H); w.tp_cv0 = a.f(3 * H); w.tp_p1x = a.f(B * 3 * H);
      w.tp_cnt = (unsigned*)a.f(8192);      // arrival slots | error word (+1024) | stamps | wait statistics (+1536)
    }
    if (d.H == 1024 && d.B <= 64 && !d.film) {
      w.bp_wr = a.f(256L * 8 * 113 * 64); w.bp_wl = a.f(256L * 8 * 64 * 64);
      w.bp_opy = a.f(T * (long)((d.PO + 15) / 16) * 512);
      w.bp_op1 = a.f(T * 4 * H * 32); w.bp_op0 = a.f(T * 4 * H * 32); w.bp_opd = a.f(T * H * 32);
      w.bp_sp = a.f(T * 9 * 32);
      w.bp_cnt = (unsigned*)a.f(16384);      // arrival slots | error word (+1024) | stamps | per-workgroup wait statistics (+1536)
    }
  }
  return w;
}

extern int g_poll_sleep;           // s_sleep units (64 clocks) between two polls of the arrival slots (option "poll_sleep")
extern int g_poll_stagger;         // != 0: two polls of the arrival slots in flight, this many s_sleep units apart (option "poll_stagger")
extern int g_persistent_spin;      // bound of the device-side waits of the persistent kernels (option "persistent_spin")
// persistent weight-stationary decode (decode_persistent.hip)
int dec_persistent_supported(const ZeggsDecDims& d, const DecWs& w);
int dec_persistent_run(const ZeggsDecDims& d, const ZeggsDecParams* P, const ZeggsDecStats* st, DecWs& w, const float* gaze,
                       const float* speech, const float* style, float* pose, float* rpos, float* rrot, const float* gin1,
                       const float* h0_init, const float* h1_init, float* h0_fin, float* h1_fin, hipStream_t s,
                       unsigned* status = nullptr);
int dec_persistent_state();
void dec_persistent_set_state(int v);
int dec_persistent_errors(const DecWs& w, unsigned* out);
int dec_persistent_errptr(const DecWs& w, unsigned** out);
// persistent training rollout (train_persistent.hip)
int dec_tp_supported(const ZeggsDecDims& d, const DecWs& w);
int dec_tp_state();
void dec_tp_set_state(int v);
int dec_tp_pack(const ZeggsDecDims& d, const ZeggsDecParams* P, const ZeggsDecStats* st, DecWs& w, hipStream_t s);
int dec_tp_zero(const ZeggsDecDims& d, DecWs& w, hipStream_t s);
int dec_tp_run(const ZeggsDecDims& d, const ZeggsDecParams* P, const ZeggsDecStats* st, DecWs& w, const float* gaze,
               const float* speech, const float* style, float* pose, float* rpos, float* rrot, hipStream_t s,
               bool zeroed = false, unsigned* status = nullptr, bool prologue_done = false);
int dec_tp_prologue(const ZeggsDecDims& d, const ZeggsDecStats* st, DecWs& w, const float* pose0, const float* rpos0,
                    const float* rrot0, const float* gaze, const float* speech, const float* style, float* pose, float* rpos,
                    float* rrot, hipStream_t s, bool zeroed);
struct GemmNtItem;
GemmNtItem dec_tp_p1x_item(const ZeggsDecDims& d, const ZeggsDecParams* P, const DecWs& w);
int dec_tp_errors(const DecWs& w, unsigned* out);
int dec_tp_errptr(const DecWs& w, unsigned** out);
// persistent BPTT sweep (train_bwd_persistent.hip)
int dec_bp_supported(const ZeggsDecDims& d, const DecWs& w);
int dec_bp_state();
void dec_bp_set_state(int v);
int dec_bp_pack(const ZeggsDecDims& d, const ZeggsDecParams* P, DecWs& w, hipStream_t s);
int dec_bp_zero_slots(DecWs& w, hipStream_t s);      // weight tiles + operand pads (weights only)
int dec_bp_run(const ZeggsDecDims& d, const ZeggsDecParams* P, const ZeggsDecStats* st, DecWs& w, const float* gaze,
               const float* pose, const float* rpos, const float* rrot, const float* dpose, const float* drpos,
               const float* drrot, hipStream_t s, bool packed = false, unsigned* status = nullptr);
int dec_bp_errors(const DecWs& w, unsigned* out);
int dec_bp_errptr(const DecWs& w, unsigned** out);
// fast path entry points (decoder_fast.hip)
int dec_fast_merge_prep(const ZeggsDecDims& d, const ZeggsDecParams* P, const ZeggsDecStats* st, DecWs& w, hipStream_t s);
void dec_timing_mark(int i, hipStream_t s);
int dec_fast_supported(const ZeggsDecDims& d);
int dec_fast_pack_fwd(const ZeggsDecDims& d, const ZeggsDecParams* P, DecWs& w, hipStream_t s);
int dec_fast_pack_bwd(const ZeggsDecDims& d, const ZeggsDecParams* P, const ZeggsDecStats* st, DecWs& w, hipStream_t s);
int dec_fast_fwd_steps(const ZeggsDecDims& d, const ZeggsDecParams* P, const ZeggsDecStats* st, DecWs& w,
                       const float* gaze, const float* speech, const float* style, float* pose, float* rpos,
                       float* rrot, int training, hipStream_t s);
int dec_fast_bwd_steps(const ZeggsDecDims& d, const ZeggsDecParams* P, const ZeggsDecStats* st, DecWs& w,
                       const float* gaze, const float* pose, const float* rpos, const float* rrot,
                       const float* dpose, const float* drpos, const float* drrot, int t_hi, int t_lo, hipStream_t s);
