// Weight-stationary persistent decode kernel: the B = 1 autoregressive rollout (generate.py's regime,
// ZEGGS/generate.py:367 -> Decoder.forward, ZEGGS/modules.py:47-162) as ONE launch for all T frames.
//
// Why: at batch 1 a decoder step is three dependent matrix-vector stages over 75.7 MB of fp32 weights.  As a chain of
// launches every stage re-streams its weights (HBM / Infinity Cache -> CU) and pays a launch boundary: 26.7 us per
// frame.  Overlapping the next stage's weight stream with the current stage (chained launches, decoder_fast.hip
// "chain") loses: the run-ahead loads fill each CU's memory queue exactly when the running stage issues its short
// dependent loads (profiles/r02_chained_launch_*).  But the whole parameter set FITS ON CHIP: 256 CUs x 512 KB of
// vector registers = 128 MB.  So each CU keeps a fixed slice of every stage's rows in registers for the whole rollout
// (162 weights per lane), and a frame costs only what is inherently serial: three all-to-all exchanges of a 4-9 KB
// vector plus a few hundred FMAs per lane.
//
// Layout (H = 1024, 256 workgroups of 512 threads, one per CU):
//   workgroup c owns hidden units 4c..4c+3 of BOTH GRU layers, rows 4c..4c+3 of the folded layer0 (hid = ELU(M h1 +
//   Wc cond + cvec + W0[:, gaze] g), M = W0[:, :PO] diag(sigma_o/sigma_i) W2 -- the same fold as the stage kernels,
//   decoder_fast.hip), output rows c, c+256, ... of layer2, and (like every workgroup) the six root rows of layer2:
//   the root integration is evaluated redundantly everywhere, so the gaze direction of x_{t+1} never crosses CUs.
//   GRU phases: the wave pair (2u, 2u+1) of unit u splits the concatenated contraction [input side | hidden side] by
//   k mod 128; a lane holds 26 (layer 0) / 16 (layer 1) k-values of the r, z, n rows.  Output phase: wave w holds two
//   complete rows (k = lane + 64 j).
// Exchange (cdna_hip_programming.md, Guideline 16, form R2): every value crosses CUs as ONE 8-byte granule
//   {tag = frame index, value} written by a relaxed agent-scope atomic store (write-through) and swept by relaxed
//   agent-scope loads until all tags match -- the data is the flag, no fence, no barrier counter.  One buffer per
//   vector suffices: nobody can publish frame t+1 of a vector before every workgroup has consumed frame t of it (the
//   three exchanges of a frame form a cycle through all workgroups).  Every sweep is bounded; on give-up the error
//   word is set, all workgroups leave, and the host falls back to the stage kernels.
#include "decoder_ws.h"
#include "dec_math.h"
#include "gemm.h"
#include "kernels.h"

int g_persistent = 1;           // zeggs_set_option("persistent", 0/1)
int g_poll_sleep = 0;
int g_poll_stagger = 0;
int g_persistent_spin = 1 << 21;   // bound of every device-side wait of the three persistent kernels ("persistent_spin")
static int g_persistent_ok = -1;   // -1 not validated yet, 0 failed once (disabled), 1 validated on this process

namespace {

typedef __attribute__((address_space(1))) unsigned long long gu64;
constexpr int PH = 1024, PTHR = 512, PNCU = 256, J0 = 26, J1 = 16, J3 = 18;

struct PArgs {
  ZeggsDecDims d;
  ZeggsDecStats st;
  const float *w_ih0, *w_hh0, *b_ih0, *b_hh0, *w_ih1, *w_hh1, *b_ih1, *b_hh1, *l2_w, *l2_b, *l0_w, *Mc, *cvec;
  const float *gaze, *speech, *style;
  float *pose, *rpos, *rrot;
  const float *gin1;           // [hid_1 | x_1] (canonical row of the first generated frame), h state before it
  const float *h0_init, *h1_init;
  float *h0_fin, *h1_fin;      // state after the last frame (streaming), may be null
  unsigned long long *g_h0, *g_h1, *g_hid, *g_xp;
  unsigned* err;
  unsigned* status;         // caller-owned sticky give-up flags (ZeggsDecCall.status), may be null
  unsigned spin;            // bound of every sweep (option "persistent_spin")
  int XD;
};

__device__ __forceinline__ void publish(unsigned long long* g, unsigned epoch, float v) {
  __hip_atomic_store((gu64*)g, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// This wave sweeps granules [lo, hi) (lane-strided, at most PER per lane) until every tag equals `epoch`, and drops the
// values into dst[i] (LDS).  Returns false when the bounded sweep gave up.
template <int PER>
__device__ __forceinline__ bool gather(const unsigned long long* g, int lo, int hi, unsigned epoch, float* dst, unsigned limit) {
  const int lane = threadIdx.x & 63;
  for (unsigned spins = 0;; ++spins) {
    bool ok = true;
    unsigned long long v[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = lo + lane + 64 * q;
      v[q] = i < hi ? __hip_atomic_load((gu64*)(g + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                    : ((unsigned long long)epoch << 32);
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = lo + lane + 64 * q;
      if ((unsigned)(v[q] >> 32) == epoch) { if (i < hi) dst[i] = __uint_as_float((unsigned)v[q]); }
      else ok = false;
    }
    if (__all(ok)) return true;
    if (spins >= limit) return false;
  }
}

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(PTHR, 2) void decode_persistent_k(PArgs a) {
  __shared__ float xcat[J0 * 128];        // operand vector of GRU layer 0: [hid | x | h0_prev], zero padded
  __shared__ float h0s[PH], h1s[PH];      // the recurrent state (full vectors)
  __shared__ float cond[128];             // speech / style columns of x_{t+1}
  __shared__ float part[8][4];
  __shared__ float rs[16];
  __shared__ float rootst[12];            // rrot(4) rpos(3) genc(3)
  __shared__ float cst[16][16];
  __shared__ float cg[6];                 // gaze columns of x: in_mean[PO..PO+2], 1 / in_std[PO..PO+2]
  __shared__ int fail;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c = blockIdx.x;
  const ZeggsDecDims& d = a.d;
  const int H = PH, T = d.T, PO = d.PO, PI = d.PI, XD = a.XD, KIN = H + XD, K0 = KIN + H, NC = d.SP + d.ST;
  const int u = wave >> 1, half = wave & 1, U = 4 * c + u;

  // ---------------------------------------------------------------- weights -> registers (once)
  float wa[3][J0], wb[3][J1], wc[2][J3];
#pragma unroll
  for (int j = 0; j < J0; ++j) {
    const int k = 128 * j + 64 * half + lane;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const long row = (long)g * H + U;
      wa[g][j] = k < KIN ? a.w_ih0[row * KIN + k] : (k < K0 ? a.w_hh0[row * H + (k - KIN)] : 0.f);
    }
  }
#pragma unroll
  for (int j = 0; j < J1; ++j) {
    const int k = 128 * j + 64 * half + lane;      // j < 8: input side (h0), j >= 8: hidden side (h1_prev)
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const long row = (long)g * H + U;
      wb[g][j] = k < H ? a.w_ih1[row * H + k] : a.w_hh1[row * H + (k - H)];
    }
  }
  // output phase, row slots 2*wave, 2*wave+1: 0..3 folded layer0 rows, 4..8 layer2 rows c + 256 i, 9..14 layer2 root rows
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int s = 2 * wave + r;
#pragma unroll
    for (int j = 0; j < J3; ++j) {
      const int k = lane + 64 * j;                  // j < 16: h1, j >= 16: cond
      float v = 0.f;
      if (s < 4) {
        const long R = 4 * c + s;
        v = k < H ? a.Mc[R * H + k] : (k - H < NC ? a.l0_w[R * XD + PI + (k - H)] : 0.f);
      } else if (s < 9) {
        const int col = c + PNCU * (s - 4);
        v = (col < PO && k < H) ? a.l2_w[(long)col * H + k] : 0.f;
      } else if (s < 15) {
        v = k < H ? a.l2_w[(long)(s - 9) * H + k] : 0.f;
      }
      wc[r][j] = v;
    }
  }
  // per-row constants of the finishing threads (threads 0..8) live in LDS: they would cost every lane a register
  // cst[tid][0..11] = b_ih0, b_hh0, b_ih1, b_hh1 (r, z, n each), [12] = cvec, [13..15] = W0[:, gaze]   (tid < 4)
  // cst[tid][0..4]  = b2, sigma_o, mu_o, mu_i, sigma_i of output column c + 256 (tid - 4), [5] = valid  (4 <= tid < 9)
  if (tid < 4) {
    const int Uu = 4 * c + tid;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      cst[tid][g] = a.b_ih0[g * H + Uu]; cst[tid][3 + g] = a.b_hh0[g * H + Uu];
      cst[tid][6 + g] = a.b_ih1[g * H + Uu]; cst[tid][9 + g] = a.b_hh1[g * H + Uu];
    }
    cst[tid][12] = a.cvec[Uu];
#pragma unroll
    for (int k = 0; k < 3; ++k) cst[tid][13 + k] = a.l0_w[(long)Uu * XD + PO + k];
  } else if (tid < 9) {
    const int col = c + PNCU * (tid - 4);
    const bool v = col < PO;
    cst[tid][0] = v ? a.l2_b[col] : 0.f; cst[tid][1] = v ? a.st.out_std[col] : 0.f; cst[tid][2] = v ? a.st.out_mean[col] : 0.f;
    cst[tid][3] = v ? a.st.in_mean[col] : 0.f; cst[tid][4] = v ? a.st.in_std[col] : 1.f; cst[tid][5] = v ? 1.f : 0.f;
  } else if (tid < 15) {      // root rows: b2, sigma_o, mu_o of output columns 0..5
    cst[tid][0] = a.l2_b[tid - 9]; cst[tid][1] = a.st.out_std[tid - 9]; cst[tid][2] = a.st.out_mean[tid - 9];
  }
  // ---------------------------------------------------------------- state before the first generated frame
  for (int i = tid; i < J0 * 128; i += PTHR) xcat[i] = 0.f;
  for (int i = tid; i < H; i += PTHR) { h0s[i] = a.h0_init[i]; h1s[i] = a.h1_init[i]; }
  if (tid < 4) rootst[tid] = a.rrot[tid];
  if (tid >= 4 && tid < 7) rootst[tid] = a.rpos[tid - 4];
  if (tid == 0) fail = 0;
  if (tid >= 32 && tid < 38) cg[tid - 32] = tid < 35 ? a.st.in_mean[PO + tid - 32] : 1.f / a.st.in_std[PO + tid - 35];
  __syncthreads();
  for (int i = tid; i < KIN; i += PTHR) xcat[i] = a.gin1[i];     // [hid_1 | x_1]
  bool bad = false;

  for (int t = 1; t < T; ++t) {
    // opaque per-frame copy of the thread index: left alone, the optimiser hoists the per-thread granule / output addresses out
    // of the frame loop and spills them; their reloads (scratch + s_waitcnt vmcnt(0)) then sit right in front of the publishing
    // stores, on the critical path of every exchange
    int tq = tid;
    asm volatile("" : "+v"(tq));
    const bool next = t + 1 < T;
    // ================================================================ GRU layer 0: operands [hid_t | x_t | h0_{t-1}]
    if (t > 1) {
      if (!gather<2>(a.g_hid, wave * 128, wave * 128 + 128, (unsigned)t, xcat, a.spin)) bad = true;
      const int per = (PO + 7) / 8;
      if (!gather<3>(a.g_xp, wave * per, min(PO, wave * per + per), (unsigned)t, xcat + H, a.spin)) bad = true;
      if (tq < 3) xcat[H + PO + tq] = rootst[7 + tq];                       // gaze direction of x_t (local)
      if (tq >= 64 && tq < 64 + NC) xcat[H + PI + (tq - 64)] = cond[tq - 64];   // speech / style of frame t
    }
    for (int i = tq; i < H; i += PTHR) xcat[KIN + i] = h0s[i];
    if (bad) fail = 1;
    __syncthreads();
    if (fail) break;
    {
      float sr = 0.f, sz = 0.f, sni = 0.f, snh = 0.f;
#pragma unroll
      for (int j = 0; j < J0; ++j) {
        const int k = 128 * j + 64 * half + lane;
        const float x = xcat[k];
        sr = fmaf(wa[0][j], x, sr);
        sz = fmaf(wa[1][j], x, sz);
        const float pn = wa[2][j] * x;
        sni += k < KIN ? pn : 0.f;
        snh += k < KIN ? 0.f : pn;
      }
      sr = wsum(sr); sz = wsum(sz); sni = wsum(sni); snh = wsum(snh);
      if (lane == 0) { part[wave][0] = sr; part[wave][1] = sz; part[wave][2] = sni; part[wave][3] = snh; }
    }
    __syncthreads();
    if (tq < 4) {
      const int Uu = 4 * c + tq;
      const float* k_ = cst[tq];
      const float r = d_sigmoid(part[2 * tq][0] + part[2 * tq + 1][0] + k_[0] + k_[3]);
      const float z = d_sigmoid(part[2 * tq][1] + part[2 * tq + 1][1] + k_[1] + k_[4]);
      const float nh = part[2 * tq][3] + part[2 * tq + 1][3] + k_[5];
      const float nn = d_tanh(part[2 * tq][2] + part[2 * tq + 1][2] + k_[2] + r * nh);
      publish(a.g_h0 + Uu, (unsigned)t, (1.f - z) * nn + z * h0s[Uu]);
    }
    __syncthreads();      // the old h0 has been read everywhere before the sweep overwrites it
    // ================================================================ GRU layer 1: operands [h0_t | h1_{t-1}]
    if (!gather<2>(a.g_h0, wave * 128, wave * 128 + 128, (unsigned)t, h0s, a.spin)) fail = 1;
    if (next && tq < NC)      // speech / style columns of frame t+1 (inputs), staged while the sweep is in flight
      cond[tq] = tq < d.SP ? a.speech[(long)(t + 1) * d.SP + tq] : a.style[(long)(t + 1) * d.ST + (tq - d.SP)];
    __syncthreads();
    if (fail) break;
    {
      float sr = 0.f, sz = 0.f, sni = 0.f, snh = 0.f;
#pragma unroll
      for (int j = 0; j < J1; ++j) {
        const int k = 128 * j + 64 * half + lane;
        const float x = j < 8 ? h0s[k] : h1s[k - H];
        sr = fmaf(wb[0][j], x, sr);
        sz = fmaf(wb[1][j], x, sz);
        if (j < 8) sni = fmaf(wb[2][j], x, sni); else snh = fmaf(wb[2][j], x, snh);
      }
      sr = wsum(sr); sz = wsum(sz); sni = wsum(sni); snh = wsum(snh);
      if (lane == 0) { part[wave][0] = sr; part[wave][1] = sz; part[wave][2] = sni; part[wave][3] = snh; }
    }
    __syncthreads();
    if (tq < 4) {
      const int Uu = 4 * c + tq;
      const float* k_ = cst[tq];
      const float r = d_sigmoid(part[2 * tq][0] + part[2 * tq + 1][0] + k_[6] + k_[9]);
      const float z = d_sigmoid(part[2 * tq][1] + part[2 * tq + 1][1] + k_[7] + k_[10]);
      const float nh = part[2 * tq][3] + part[2 * tq + 1][3] + k_[11];
      const float nn = d_tanh(part[2 * tq][2] + part[2 * tq + 1][2] + k_[8] + r * nh);
      publish(a.g_h1 + Uu, (unsigned)t, (1.f - z) * nn + z * h1s[Uu]);
    }
    __syncthreads();
    // ================================================================ output: y_t = W2 h1_t + b2, root, x_{t+1}, hid_{t+1}
    if (!gather<2>(a.g_h1, wave * 128, wave * 128 + 128, (unsigned)t, h1s, a.spin)) fail = 1;
    __syncthreads();
    if (fail) break;
    {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int j = 0; j < J3; ++j) {
        const int k = lane + 64 * j;
        const float x = j < 16 ? h1s[k] : (k - H < NC ? cond[k - H] : 0.f);
        s0 = fmaf(wc[0][j], x, s0);
        s1 = fmaf(wc[1][j], x, s1);
      }
      s0 = wsum(s0); s1 = wsum(s1);
      if (lane == 0) { rs[2 * wave] = s0; rs[2 * wave + 1] = s1; }
    }
    __syncthreads();
    if (tq == 0) {       // root integration (ZEGGS/modules.py:139-176), evaluated in every workgroup
      float p[6], rt[10], genc[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 6; ++q) p[q] = (rs[9 + q] + cst[9 + q][0]) * cst[9 + q][1] + cst[9 + q][2];
#pragma unroll
      for (int q = 0; q < 7; ++q) rt[q] = rootst[q];
      if (next) { const float* gz = a.gaze + (long)(t + 1) * 3; rt[7] = gz[0]; rt[8] = gz[1]; rt[9] = gz[2]; }
      V3 npos; Q4 nq;
      {   // root_step of decoder_fast.hip, restated here (file-local there)
        const Q4 q = Q4{rt[0], rt[1], rt[2], rt[3]};
        const V3 pos = v3(rt[4], rt[5], rt[6]);
        npos = quat_mul_vec(q, d.dt * v3(p[0], p[1], p[2])) + pos;
        const V3 uu = quat_mul_vec(q, d.dt * v3(p[3], p[4], p[5]));
        nq = quat_exp_mul(0.5f * uu, q);
        if (next) {
          const V3 gd = quat_mul_vec(quat_inv(nq), v3(rt[7], rt[8], rt[9]) - npos);
          genc[0] = (gd.x - cg[0]) * cg[3];
          genc[1] = (gd.y - cg[1]) * cg[4];
          genc[2] = (gd.z - cg[2]) * cg[5];
        }
      }
      rootst[0] = nq.w; rootst[1] = nq.x; rootst[2] = nq.y; rootst[3] = nq.z;
      rootst[4] = npos.x; rootst[5] = npos.y; rootst[6] = npos.z;
      rootst[7] = genc[0]; rootst[8] = genc[1]; rootst[9] = genc[2];
      if (c == 0) {
        float* op = a.rpos + (long)t * 3; op[0] = npos.x; op[1] = npos.y; op[2] = npos.z;
        float* oq = a.rrot + (long)t * 4; oq[0] = nq.w; oq[1] = nq.x; oq[2] = nq.y; oq[3] = nq.z;
      }
    } else if (tq >= 64 && tq < 69 && cst[tq - 60][5] != 0.f) {      // (second wave: not behind the root thread's branch)
      const int r = tq - 60;
      const float* k_ = cst[r];
      const int ocol = c + PNCU * (r - 4);
      const float pv = (rs[r] + k_[0]) * k_[1] + k_[2];
      a.pose[(long)t * PO + ocol] = pv;
      if (next) publish(a.g_xp + ocol, (unsigned)(t + 1), (pv - k_[3]) / k_[4]);
    }
    __syncthreads();      // gaze direction of x_{t+1}
    if (next && tq < 4)
      publish(a.g_hid + 4 * c + tq, (unsigned)(t + 1),
              d_elu(rs[tq] + cst[tq][12] + cst[tq][13] * rootst[7] + cst[tq][14] * rootst[8] + cst[tq][15] * rootst[9]));
  }
  if (fail) {     // a bounded sweep gave up: error word, the caller's sticky status, and NaN in what a consumer reads first
    if (tid == 0) {
      atomicOr(a.err, 1u);
      if (a.status) atomicOr(a.status, ZEGGS_GAVE_UP_DECODE);
    }
    if (c == 0) {
      const float qnan = __uint_as_float(0x7fc00000u);
      for (int i = tid; i < PO; i += PTHR) a.pose[(long)(T - 1) * PO + i] = qnan;
      if (tid < 3) a.rpos[(long)(T - 1) * 3 + tid] = qnan;
      if (tid < 4) a.rrot[(long)(T - 1) * 4 + tid] = qnan;
    }
    return;
  }
  if (c == 0 && a.h0_fin)
    for (int i = tid; i < H; i += PTHR) { a.h0_fin[i] = h0s[i]; a.h1_fin[i] = h1s[i]; }
}

}  // namespace

// dims this kernel is built for (configs_v1 / v2 of the reference: H = 1024)
int dec_persistent_supported(const ZeggsDecDims& d, const DecWs& w) {
  return !d.film && d.B == 1 && d.H == PH && d.T >= 4 && d.PI == d.PO + 3 && d.H + w.XD + d.H <= J0 * 128 &&
         d.SP + d.ST <= 128 && d.PO <= 5 * PNCU && d.PO >= 6;
}

// hid_1 / x_1 / the initial state are in the workspace (decoder.hip: dec_init_k, CellStateEncoder or h_in, layer0 of
// the first frame); Mc / cvec are the folded layer0 operands (dec_fast_merge_prep).
int dec_persistent_run(const ZeggsDecDims& d, const ZeggsDecParams* P, const ZeggsDecStats* st, DecWs& w, const float* gaze,
                       const float* speech, const float* style, float* pose, float* rpos, float* rrot, const float* gin1,
                       const float* h0_init, const float* h1_init, float* h0_fin, float* h1_fin, hipStream_t s,
                       unsigned* status) {
  int dev = 0, ncu = 0;
  ZCHECK(hipGetDevice(&dev) == hipSuccess, "hipGetDevice failed");
  ZCHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess, "device query failed");
  ZCHECK(ncu >= PNCU, "persistent decode needs %d CUs (device has %d)", PNCU, ncu);
  ZTRY(k_fill((float*)w.pgran, (long)(w.pgran_bytes / 4), 0.f, s));      // tags 0 = nothing published
  PArgs a;
  memset(&a, 0, sizeof(a));
  a.d = d; a.st = *st;
  a.w_ih0 = P->w_ih0; a.w_hh0 = P->w_hh0; a.b_ih0 = P->b_ih0; a.b_hh0 = P->b_hh0;
  a.w_ih1 = P->w_ih1; a.w_hh1 = P->w_hh1; a.b_ih1 = P->b_ih1; a.b_hh1 = P->b_hh1;
  a.l2_w = P->l2_w; a.l2_b = P->l2_b; a.l0_w = P->l0_w; a.Mc = w.Mc; a.cvec = w.cvec;
  a.gaze = gaze; a.speech = speech; a.style = style; a.pose = pose; a.rpos = rpos; a.rrot = rrot;
  a.gin1 = gin1; a.h0_init = h0_init; a.h1_init = h1_init; a.h0_fin = h0_fin; a.h1_fin = h1_fin;
  unsigned long long* g = (unsigned long long*)w.pgran;
  a.g_h0 = g; a.g_h1 = g + PH; a.g_hid = g + 2 * PH; a.g_xp = g + 3 * PH;
  a.err = (unsigned*)(g + 3 * PH + 5 * PNCU);
  a.XD = w.XD;
  a.status = status; a.spin = (unsigned)g_persistent_spin;
  hipLaunchKernelGGL(decode_persistent_k, dim3(PNCU), dim3(PTHR), 0, s, a);
  ZLAUNCH_CHECK("decode_persistent");
  return 0;
}

// 1: validated on this process, 0: failed (disabled), -1: unknown
int dec_persistent_state() { return g_persistent_ok; }
void dec_persistent_set_state(int v) { g_persistent_ok = v; }
int dec_persistent_errptr(const DecWs& w, unsigned** out) {
  *out = (unsigned*)((unsigned long long*)w.pgran + 3 * PH + 5 * PNCU);
  return 0;
}
int dec_persistent_errors(const DecWs& w, unsigned* out) {
  const unsigned long long* g = (const unsigned long long*)w.pgran;
  ZCHECK(hipMemcpy(out, (const void*)(g + 3 * PH + 5 * PNCU), sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess,
         "persistent decode: error word copy failed");
  return 0;
}
