// Weight-stationary persistent BPTT sweep of the training step: the 255 backward decoder steps of a window as ONE launch for
// up to 32 batch rows (33..64 rows: two launches, the rows are independent) -- the backward mirror of train_persistent.hip.
//
// Why a different tile shape than the forward kernel.  The transposed products of a step (W_ih1^T, W_hh1^T, W_ih0^T, W_hh0^T,
// W0^T, W2^T: 80 MB) give every one of the 256 workgroups 4 + 4 + 4 + 8 + 4 output rows; with the 16-row tiles of
// v_mfma_f32_16x16x4_f32 the padded fragments (530-650 KB per CU) do not fit the 512 KB register file + 160 KB LDS of a CU.
// v_mfma_f32_4x4x1_16b_f32 has 4-row tiles: with cbsz = 3 the 16 blocks of the instruction are 8 batch groups x 2 k-halves
// that share the A values of block `abid`, i.e. ONE VGPR holds a [4 rows x 16 k] weight tile with no padding and eight
// instructions (abid = 0..7) multiply it with a [16 k x 32 batch] operand block (measured: 131 TFLOP/s = 85 % of the 16x16x4
// rate, tools/mfma4_probe.hip).  The per-CU weights are then 354 KB: 113 VGPRs per lane + 128 KB of LDS.
//
// Per step t (T-1 .. 1) four phases, each ending in a grid hand-off (arrival slots, as in train_persistent.hip).  A hand-off
// costs ~2.5 us (store drain, flag, poll); the carry products W_hh^T (.) are not needed before the NEXT step, so they are
// cut into pieces that run in the windows before the waits ("old" operands, in brackets):
//   P1  [carry0 += first half of W_hh0^T (DI0 r,z | dn_h0) of step t+1]
//       dH1 = W2^T dy_t + carry1           -> layer-1 gate gradients DI1_t, dn_h1 (GRU backward);  carry1 = dH1 * z
//   P2  [carry0 += second half of W_hh0^T (.) of step t+1]
//       dH0 = W_ih1^T DI1_t + carry0       -> layer-0 gate gradients DI0_t, dn_h0;                 carry0 = dH0 * z
//   P3  [carry1 += first half of W_hh1^T (DI1_t r,z | dn_h1)]
//       dGin = W_ih0^T DI0_t               -> D0_t = dhid * ELU'(hid_t), dXa (kept in LDS by the row's owner)
//   P4  [carry1 += second half of W_hh1^T (.)]
//       dx_t = dXa + W0^T D0_t             -> speech / style columns to DX[t]; pose columns -> dy_{t-1}
//                                             (devectorize / vectorize backward; the 9 root / gaze columns belong to
//                                             workgroup 0, whose first 32 threads carry the root-integration adjoint)
// Workgroup c owns hidden units 4c..4c+3 of both layers (the carries never leave its registers), rows 4c..4c+3 of dhid and
// rows 8c..8c+7 of dx (c < 158: two aligned groups of four = one 16-byte store each).  Operands travel through WRITE-ONCE time-major buffers in the B layout of the instruction
// ([k-block][2][64 lanes][4]: lane = 32 * k-half + batch row), published with write-through stores; the canonical copies
// (DI1, DH1, DI0, DH0, D0, DY, DX) are written where the stage kernels write them, so the weight-gradient GEMMs and the
// CellStateEncoder backward are unchanged.  Every wait is bounded; on give-up the error word is set.
#include "decoder_ws.h"
#include "dec_math.h"
#include "kernels.h"

int g_bwd_persistent = 1;        // zeggs_set_option("bwd_persistent", 0/1)
static int g_bp_ok = -1;

namespace {

typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64t;
constexpr int BH = 1024, BTHR = 512, BNCU = 256;
// blocks (16 k each) per wave and part; block j of a wave is enumeration index e = wave + 8 j of the part.  Every workgroup
// streams the whole operand of a part through its CU (2 KB per block, ~90 GB/s per CU when all CUs read the same lines),
// which is what bounds the single-row-group parts.
constexpr int NJ1 = 9;      // P1: dy_t                                    71 blocks (PO = 1131), 1 row group
constexpr int NJ2A = 16;    // P2: DI1_t blocks 0..127 (tiles in LDS)      1 row group
constexpr int NJ2B = 8;     // P2: DI1_t blocks 128..191
constexpr int NJC = 12;     // a carry half: 96 of the 192 blocks [DI r,z (0..127) | dn_h (192..255)], 1 row group
constexpr int NJA = 16;     // P3, blocks 0..127 of DI0 (tiles in LDS):    dhid | dXa | dXa
constexpr int NJB = 8;      // P3, blocks 128..191
constexpr int NJ4 = 8;      // P4: D0_t                                    64 blocks, 3 row groups
// register-resident weight tiles of a wave (one VGPR each); the tiles of P3's first part and of P2's first part live in LDS
constexpr int O1 = 0, O2B = O1 + NJ1, OC1A = O2B + NJ2B, OC1B = OC1A + NJC, O3B = OC1B + NJC, OC0A = O3B + 3 * NJB,
              OC0B = OC0A + NJC, O4 = OC0B + NJC;
constexpr int NWR = O4 + 3 * NJ4;             // 113
constexpr int L3A = 3 * NJA, L3 = L3A + NJ2A; // 48 + 16 = 64 LDS tiles per wave
constexpr int NSP = 9;      // root / gaze columns of x: 0..5, PO..PO+2

struct BArgs {
  ZeggsDecDims d;
  ZeggsDecStats st;
  int XD, GL, POL, KBY;
  int Bact;                                  // batch rows of THIS sweep (<= 32); d.B is the full batch = the row stride of every array,
                                             // whose base pointers the host offsets to the first row of the sweep
  const float *PWR, *PWL;                    // [256][8][NWR][64], [256][8][L3][64]
  float *OPY, *OP1, *OP0, *OPD, *SP;         // operands (time-major, write-once), dXa of the root / gaze columns [T][9][32]
  const float *Gin, *H0, *H1, *GT0, *GT1;    // forward saves
  float *DY, *DI1, *DH1, *DI0, *DH0, *D0, *DX, *dH0c, *dH1c;
  const float *dpose, *drpos, *drrot, *gaze, *pose, *rpos, *rrot;
  const float* carry;                        // root adjoint after frame T-1 [B][8]
  unsigned *cnt, *err;
  unsigned* status;                          // caller-owned sticky give-up flags (ZeggsDecCall.status), may be null
  unsigned spin;                             // bound of every wait (option "persistent_spin")
  unsigned nap;                              // s_sleep units between two polls (option "poll_sleep")
  unsigned stag;                             // != 0: two staggered polls in flight (option "poll_stagger")
};

__device__ __forceinline__ void stp(float* p, float v) {       // published: write-through
  __hip_atomic_store((gu32*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 16 bytes, write-through: one lane publishes 4 consecutive contraction indices (k % 4 == 0) of its batch row, a half-wave
// of batch rows 512 contiguous bytes -- whole lines instead of byte-masked partial writes
__device__ __forceinline__ void stp4(float* p, f4 v) {
  // NOTE the "memory" clobber is required (without it the results are corrupted), and with it the compiler drains every store
  // it knows to be in flight (s_waitcnt vmcnt(0): about a microsecond) before this one: call stp4 BEFORE the plain stores of an
  // epilogue, never after them
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
// position of (batch row b, contraction index k) in an operand buffer: lane 32 * ((k >> 3) & 1) + b reads
// float4 q = (k >> 2) & 1 of block k >> 4, element k & 3 (= abid & 3 of the instruction that consumes it)
__host__ __device__ inline long op_idx(int b, int k) {
  return ((((long)(k >> 4) * 2 + ((k >> 2) & 1)) * 64 + ((((k >> 3) & 1) << 5) | b)) << 2) | (k & 3);
}

__device__ __forceinline__ bool bp_wait(const unsigned* slots, unsigned expect, unsigned limit, unsigned nap = 0) {
  const int lane = threadIdx.x & 63;
  const gu64t* q = (const gu64t*)(slots + 4 * lane);
  for (unsigned spins = 0;; ++spins) {
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool ok = (unsigned)a >= expect && (unsigned)(a >> 32) >= expect && (unsigned)b >= expect && (unsigned)(b >> 32) >= expect;
    if (__all(ok)) return true;
    if (spins >= limit) return false;
    for (unsigned i = 0; i < nap; ++i) __builtin_amdgcn_s_sleep(1);
  }
}
// two staggered samples in flight (train_persistent.hip: tp_wait2)
__device__ __forceinline__ bool bp_wait2(const unsigned* slots, unsigned expect, unsigned limit, unsigned stagger) {
  const int lane = threadIdx.x & 63;
  const gu64t* q = (const gu64t*)(slots + 4 * lane);
  auto all_in = [&](unsigned long long a, unsigned long long b) {
    return __all((unsigned)a >= expect && (unsigned)(a >> 32) >= expect && (unsigned)b >= expect && (unsigned)(b >> 32) >= expect);
  };
  unsigned long long a0 = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long b0 = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (unsigned i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(1);
  for (unsigned spins = 0;; spins += 2) {
    const unsigned long long a1 = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b1 = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (all_in(a0, b0)) return true;
    a0 = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b0 = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (all_in(a1, b1)) return true;
    if (spins >= limit) return false;
  }
}

// products of one part: slot j = 0..NJ-1 of this wave is k-block kb0 + wave + 8 j (< kb0 + nblk) of the operand, NRG row groups;
// (measured: starting every workgroup's walk at a different block, so that an XCD pulls a fresh operand in faster, is 3 % slower) weight tile (j, rg) = wr[OFF + j*NRG + rg] or, for LDS, wl[(j*NRG + rg) * 64].
// Two blocks per group, the next group's operand loads are kept ahead of this group's products by scheduling fences.
// SKIPN: the part walks the list [0..127] + [192..255] of the operand's blocks (carry products: the n rows of DI are not theirs)
template <int NRG, int NJ, int OFF, bool LDS, bool SKIPN = false>
__device__ __forceinline__ void bp_mma(const float (&wr)[NWR], const float* wl, const f4* __restrict__ xb, int wave, int kb0,
                                       int nblk, f4* acc, long twin = 0) {
#ifdef ZEGGS_BP_TWICE   // (timing experiment, results wrong: every part walks a SECOND operand -- the same blocks of a neighbouring time step,
                        // other lines -- into the same accumulators: the product stream of a 64-row step without its registers / epilogues)
  for (int rep = 0; rep < 2; ++rep, xb += twin) {
#endif
  constexpr int GU = NRG >= 3 ? 1 : 2, NG = (NJ + GU - 1) / GU;     // operand blocks per group (matrix-core bound parts: 1)
  // the block offsets are cheap scalar arithmetic; hidden from the optimiser's loop-invariant code motion, which otherwise
  // keeps ~100 of them (one per block of every part) live across the whole time loop and spills registers for it
  asm volatile("" : "+s"(wave));
  f4 xa[GU][2], xq[GU][2];
  auto load = [&](f4 (&x)[GU][2], int g) {
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      int e = wave + 8 * (GU * g + u);
      e = e < nblk ? e : nblk - 1;              // past the end: the weights of that slot are zero (bp_pack_k)
      int kb = kb0 + e;
      if (SKIPN) kb = kb < 128 ? kb : kb + 64;
      x[u][0] = xb[(long)(kb * 2) * 64];
      x[u][1] = xb[(long)(kb * 2 + 1) * 64];
    }
  };
  auto comp = [&](const f4 (&x)[GU][2], int g) {
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int j = GU * g + u;
      if (j < NJ) {
#pragma unroll
        for (int rg = 0; rg < NRG; ++rg) {
          const int p = j * NRG + rg;
          const float wv = LDS ? wl[p * 64] : wr[OFF + p];
          acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, x[u][0][0], acc[rg], 3, 0, 0);
          acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, x[u][0][1], acc[rg], 3, 1, 0);
          acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, x[u][0][2], acc[rg], 3, 2, 0);
          acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, x[u][0][3], acc[rg], 3, 3, 0);
          acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, x[u][1][0], acc[rg], 3, 4, 0);
          acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, x[u][1][1], acc[rg], 3, 5, 0);
          acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, x[u][1][2], acc[rg], 3, 6, 0);
          acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, x[u][1][3], acc[rg], 3, 7, 0);
        }
      }
    }
  };
  load(xa, 0);
#pragma unroll
  for (int g = 0; g < NG; g += 2) {
    if (g + 1 < NG) load(xq, g + 1);
    __builtin_amdgcn_sched_barrier(0);
    comp(xa, g);
    __builtin_amdgcn_sched_barrier(0);
    if (g + 2 < NG) load(xa, g + 2);
    __builtin_amdgcn_sched_barrier(0);
    if (g + 1 < NG) comp(xq, g + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
#ifdef ZEGGS_BP_TWICE
  }
#endif
}

// inputs of the root-integration backward of frame f (forward outputs and upstream gradients: old data).  NRI items per
// batch row; workgroup 0 fetches them with all its threads into LDS ahead of the products (root_item), the root thread of a
// row reads them back (RootIn::from_lds)
constexpr int NRI = 33;
__device__ __forceinline__ float root_item(const float* drpos, const float* drrot, const float* rrot, const float* rpos,
                                           const float* gaze, const float* pose, const float* dpose, int T, int PO, int b, int f,
                                           bool has_next, int item) {
  const long bf = (long)b * T + f;
  if (item < 3) return drpos[bf * 3 + item];
  if (item < 7) return drrot[bf * 4 + item - 3];
  if (item < 11) return rrot[bf * 4 + item - 7];
  if (item < 14) return rpos[bf * 3 + item - 11];
  if (item < 17) return has_next ? gaze[(bf + 1) * 3 + item - 14] : 0.f;
  if (item < 21) return rrot[(bf - 1) * 4 + item - 17];
  if (item < 27) return pose[bf * PO + item - 21];
  return dpose[bf * PO + item - 27];
}
struct RootIn {
  float a[3], e[4], rq[4], rp[3], gz[3], pq[4], pt[6], dp[6];
  template <class F>
  __device__ __forceinline__ void gather(F get) {      // get(item) -> value
#pragma unroll
    for (int i = 0; i < 3; ++i) { a[i] = get(i); rp[i] = get(11 + i); gz[i] = get(14 + i); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { e[i] = get(3 + i); rq[i] = get(7 + i); pq[i] = get(17 + i); }
#pragma unroll
    for (int i = 0; i < 6; ++i) { pt[i] = get(21 + i); dp[i] = get(27 + i); }
  }
};
// backward of the root integration of frame f with the adjoint carry in registers (decoder_fast.hip root_bwd, decoder.hip
// dec_devec_bwd_k): cr = gradient wrt (root_pos_f, root_rot_f) arriving from the later frames.
// g6: in = dpose[f][0:6] + dx/sigma_i ; out = total gradient wrt pose[f][0:6] (de-normalised output space)
__device__ __forceinline__ void root_bwd_reg(const ZeggsDecDims& d, const float* gaze_in_std /* [3] */, const RootIn& ri,
                                             const float (&dgd_in)[3], float (&cr)[7], float (&g6)[6], bool has_next = true) {
  V3 g_rp = v3(cr[0] + ri.a[0], cr[1] + ri.a[1], cr[2] + ri.a[2]);
  Q4 g_rr = Q4{cr[3] + ri.e[0], cr[4] + ri.e[1], cr[5] + ri.e[2], cr[6] + ri.e[3]};
  const Q4 q_t = Q4{ri.rq[0], ri.rq[1], ri.rq[2], ri.rq[3]};
  const V3 p_t = v3(ri.rp[0], ri.rp[1], ri.rp[2]);
  if (has_next) {
    const V3 dgd = v3(dgd_in[0] / gaze_in_std[0], dgd_in[1] / gaze_in_std[1], dgd_in[2] / gaze_in_std[2]);
    Q4 dqi; V3 dv;
    qmv_bwd(quat_inv(q_t), v3(ri.gz[0], ri.gz[1], ri.gz[2]) - p_t, dgd, dqi, dv);
    g_rr.w += dqi.w; g_rr.x -= dqi.x; g_rr.y -= dqi.y; g_rr.z -= dqi.z;
    g_rp = g_rp - dv;
  }
  const Q4 q_p = Q4{ri.pq[0], ri.pq[1], ri.pq[2], ri.pq[3]};
  const V3 vel = v3(ri.pt[0], ri.pt[1], ri.pt[2]), vrt = v3(ri.pt[3], ri.pt[4], ri.pt[5]);
  Q4 dq1; V3 dv1;
  qmv_bwd(q_p, d.dt * vel, g_rp, dq1, dv1);
  const V3 u = quat_mul_vec(q_p, d.dt * vrt);
  QExpCtx ec;
  const Q4 E = quat_exp_ctx(0.5f * u, ec);
  Q4 dE, dqy;
  qmul_bwd(E, q_p, g_rr, dE, dqy);
  const V3 du = 0.5f * qexp_bwd_ctx(0.5f * u, dE, ec);
  Q4 dq2; V3 dv2;
  qmv_bwd(q_p, d.dt * vrt, du, dq2, dv2);
  g6[0] += d.dt * dv1.x; g6[1] += d.dt * dv1.y; g6[2] += d.dt * dv1.z;
  g6[3] += d.dt * dv2.x; g6[4] += d.dt * dv2.y; g6[5] += d.dt * dv2.z;
  cr[0] = g_rp.x; cr[1] = g_rp.y; cr[2] = g_rp.z;
  cr[3] = dq1.w + dqy.w + dq2.w; cr[4] = dq1.x + dqy.x + dq2.x;
  cr[5] = dq1.y + dqy.y + dq2.y; cr[6] = dq1.z + dqy.z + dq2.z;
}

// The same backward split in two for the persistent kernel: everything that depends on forward data only (the rotations,
// quat_exp and its sine / cosine) is evaluated one phase ahead by the root thread and parked in its LDS column (33 values,
// in place of the raw inputs); what is left behind the hand-off is linear in the incoming gradients.
struct RootPre { Q4 qti; V3 w; Q4 qp; V3 dvel, dvrt, hu; Q4 E; float ea, eb_, ec; float dp[6]; };
// backward of quat_exp at a prepared point x: du = ea * g_v + (eb_ * g_w + ec * <g_v, x>) * x   (dec_math.h qexp_bwd_ctx with the
// divisions by the norm done when the frame was prepared)
__device__ __forceinline__ void qexp_bwd_coef(V3 x, const QExpCtx& c, float& ea, float& eb_, float& ec) {
  if (c.h < 1e-5f) {
    const float n = sqrtf(1.f + c.h * c.h), ne = n + 1e-5f, k = 1.f / (n * ne * ne);
    ea = 1.f / ne; eb_ = -k; ec = -k;
  } else {
    const float s = c.sh / c.h, ds = (c.h * c.ch - c.sh) / (c.h * c.h);
    ea = s; eb_ = -s; ec = ds / c.h;
  }
}
__device__ __forceinline__ void root_prepare(const ZeggsDecDims& d, const RootIn& ri, RootPre& o) {
  o.qti = quat_inv(Q4{ri.rq[0], ri.rq[1], ri.rq[2], ri.rq[3]});
  o.w = v3(ri.gz[0], ri.gz[1], ri.gz[2]) - v3(ri.rp[0], ri.rp[1], ri.rp[2]);
  o.qp = Q4{ri.pq[0], ri.pq[1], ri.pq[2], ri.pq[3]};
  o.dvel = d.dt * v3(ri.pt[0], ri.pt[1], ri.pt[2]);
  o.dvrt = d.dt * v3(ri.pt[3], ri.pt[4], ri.pt[5]);
  o.hu = 0.5f * quat_mul_vec(o.qp, o.dvrt);
  QExpCtx ctx;
  o.E = quat_exp_ctx(o.hu, ctx);
  qexp_bwd_coef(o.hu, ctx, o.ea, o.eb_, o.ec);
#pragma unroll
  for (int i = 0; i < 6; ++i) o.dp[i] = ri.dp[i];
}
template <class F>
__device__ __forceinline__ void root_pre_store(const RootPre& o, F put) {      // put(slot, value)
  put(0, o.qti.w); put(1, o.qti.x); put(2, o.qti.y); put(3, o.qti.z); put(4, o.w.x); put(5, o.w.y); put(6, o.w.z);
  put(7, o.qp.w); put(8, o.qp.x); put(9, o.qp.y); put(10, o.qp.z); put(11, o.dvel.x); put(12, o.dvel.y); put(13, o.dvel.z);
  put(14, o.dvrt.x); put(15, o.dvrt.y); put(16, o.dvrt.z); put(17, o.hu.x); put(18, o.hu.y); put(19, o.hu.z);
  put(20, o.E.w); put(21, o.E.x); put(22, o.E.y); put(23, o.E.z); put(24, o.ea); put(25, o.eb_); put(26, o.ec);
#pragma unroll
  for (int i = 0; i < 6; ++i) put(27 + i, o.dp[i]);
}
// cr: adjoint of (root_pos_f, root_rot_f) INCLUDING this frame's direct loss gradients (drpos / drrot were added when the
// frame was prepared).  The prepared values are fetched (get(slot), see root_pre_store) right where they are used: the
// workgroup's registers are full of weights, and 33 values loaded up front get spilled to scratch around this code.
template <class F>
__device__ __forceinline__ void root_apply(const ZeggsDecDims& d, const float* gaze_rstd /* 1 / in_std[PO..PO+2] */, F get,
                                           const float (&dgd_in)[3], float (&cr)[7], float (&g6)[6]) {
  V3 g_rp = v3(cr[0], cr[1], cr[2]);
  Q4 g_rr = Q4{cr[3], cr[4], cr[5], cr[6]};
  {
    const V3 dgd = v3(dgd_in[0] * gaze_rstd[0], dgd_in[1] * gaze_rstd[1], dgd_in[2] * gaze_rstd[2]);
    Q4 dqi; V3 dv;
    qmv_bwd(Q4{get(0), get(1), get(2), get(3)}, v3(get(4), get(5), get(6)), dgd, dqi, dv);
    g_rr.w += dqi.w; g_rr.x -= dqi.x; g_rr.y -= dqi.y; g_rr.z -= dqi.z;
    g_rp = g_rp - dv;
  }
  __builtin_amdgcn_sched_barrier(0);
  const Q4 qp = Q4{get(7), get(8), get(9), get(10)};
  Q4 dq1; V3 dv1;
  qmv_bwd(qp, v3(get(11), get(12), get(13)), g_rp, dq1, dv1);
  g6[0] += d.dt * dv1.x; g6[1] += d.dt * dv1.y; g6[2] += d.dt * dv1.z;
  __builtin_amdgcn_sched_barrier(0);
  Q4 dE, dqy;
  qmul_bwd(Q4{get(20), get(21), get(22), get(23)}, qp, g_rr, dE, dqy);
  __builtin_amdgcn_sched_barrier(0);
  V3 du;
  {
    const V3 hu = v3(get(17), get(18), get(19)), gv = v3(dE.x, dE.y, dE.z);
    du = 0.5f * (get(24) * gv + (get(25) * dE.w + get(26) * dot(gv, hu)) * hu);
  }
  Q4 dq2; V3 dv2;
  qmv_bwd(qp, v3(get(14), get(15), get(16)), du, dq2, dv2);
  g6[3] += d.dt * dv2.x; g6[4] += d.dt * dv2.y; g6[5] += d.dt * dv2.z;
  cr[0] = g_rp.x; cr[1] = g_rp.y; cr[2] = g_rp.z;
  cr[3] = dq1.w + dqy.w + dq2.w; cr[4] = dq1.x + dqy.x + dq2.x;
  cr[5] = dq1.y + dqy.y + dq2.y; cr[6] = dq1.z + dqy.z + dq2.z;
}

// dx rows: workgroup c owns rows 8c .. 8c+7 (two aligned groups of four: one 16-byte store each); its dXa (P3) stays in
// LDS for P4.  The 9 root / gaze columns (0..5, PO..PO+2) are the exception: whoever owns them in P3 sends their dXa to
// workgroup 0 through SP, and workgroup 0 owns them in P4 (slots 0..7 = rows 0..7, slots 8..10 = PO..PO+2).
__host__ __device__ inline int special_index(int row, int PO) { return row < 6 ? row : (row >= PO && row < PO + 3 ? 6 + row - PO : -1); }
__host__ __device__ inline int p3_row(int c, int s, int XD) {      // dXa slot s (0..7) of workgroup c
  const int r = 8 * c + s;
  return r < XD ? r : -1;
}
__host__ __device__ inline int p4_row(int c, int s, int PO, int XD) {      // dx slot s (0..11) of workgroup c, -1: empty
  if (c == 0) return s < 8 ? s : (s < 11 ? PO + s - 8 : -1);
  if (s >= 8) return -1;
  const int r = 8 * c + s;
  return (r < XD && special_index(r, PO) < 0) ? r : -1;
}

#ifdef ZEGGS_BPTIME
#define BPT(i)                                                                                              \
  do {                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (t <= 3 && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))                      \
      ((unsigned long long*)(a.err + 32))[((3 - t) * 2 + (blockIdx.x != 0)) * 32 + (i)] = wall_clock64();     \
  } while (0)
#else
#define BPT(i) do {} while (0)
#endif

__global__ __launch_bounds__(BTHR, 2) void train_bwd_persistent_k(BArgs a) {
  __shared__ float wl3[8 * L3 * 64];          // P3 weight tiles of this workgroup (128 KB)
  __shared__ float red[8][16][32];            // per-wave partial sums [row][batch]
  __shared__ float dxa[8][32];                // dXa of this workgroup's dx rows (P3 -> P4)
  __shared__ f4 ex[4][32];                    // epilogue exchange: 4 consecutive rows of a batch row -> one 16-byte store
  __shared__ float sp9[NSP][32];              // workgroup 0: dx of the root / gaze columns
  __shared__ float rin[NRI][32];              // workgroup 0: inputs of the root-integration backward of this step
  __shared__ float rst[16];
  __shared__ float crs[7][32];                // workgroup 0: adjoint of (root_pos, root_rot) per batch row                   // in_std[0..5], in_std[PO..PO+2], out_std[0..5]
  __shared__ int fail;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c = blockIdx.x;
  const ZeggsDecDims& d = a.d;
  const int B = d.B, T = d.T, H = BH, PO = d.PO, PI = d.PI, XD = a.XD, GL = a.GL, POL = a.POL;
  const long sH = (long)B * H, s3 = 3 * sH, sG = (long)B * GL;
  // ---------------------------------------------------------------- weights -> registers / LDS (once per sweep)
  float wr[NWR];
  {
    const float* p = a.PWR + ((long)(c * 8 + wave) * NWR) * 64 + lane;
#pragma unroll
    for (int i = 0; i < NWR; ++i) wr[i] = p[(long)i * 64];
    const float* q = a.PWL + ((long)(c * 8 + wave) * L3) * 64 + lane;
#pragma unroll 8
    for (int i = 0; i < L3; ++i) wl3[(wave * L3 + i) * 64 + lane] = q[(long)i * 64];
  }
  if (tid == 0) fail = 0;
  if (tid < 15) rst[tid] = tid < 6 ? 1.f / a.st.in_std[tid] : tid < 9 ? 1.f / a.st.in_std[d.PO + tid - 6] : a.st.out_std[tid - 9];
  const float* wl = wl3 + wave * L3 * 64 + lane;
  // epilogue item of this thread: output row er (0..15) of the phase, batch row eb.  The derived indices are re-derived from an
  // opaque copy of the thread index at the start of every phase (refresh): left alone, the optimiser hoists every per-thread
  // 64-bit address they feed (a few dozen) out of the time loop and spills registers to keep them
  int er = tid >> 5, eb = tid & 31;
  bool bact = eb < a.Bact;
  int U = 4 * c + (er & 3);                     // hidden unit / dhid row of the GRU items (er < 4, er 4..7)
  const int U0 = 4 * c;
  int s4 = er - 4;                              // P4 item: dx slot 0..11
  int row4 = er >= 4 ? p4_row(c, s4, PO, XD) : -1;
  int row3 = er >= 8 ? p3_row(c, er - 8, XD) : -1;     // P3 dXa item: slot er - 8
  int sp3 = row3 >= 0 ? special_index(row3, PO) : -1;
  float c1 = 0.f, c0 = 0.f;                     // carries dH1c / dH0c of (U, eb): threads er < 4
  float si4 = 1.f, so4 = 0.f;
  if (row4 >= 0 && row4 < PI) { si4 = a.st.in_std[row4]; so4 = row4 < PO ? a.st.out_std[row4] : 0.f; }
  auto refresh = [&]() {
    int tx = tid;
    asm volatile("" : "+v"(tx));
    er = tx >> 5; eb = tx & 31;
    bact = eb < a.Bact;
    U = 4 * c + (er & 3);
    s4 = er - 4;
    row4 = er >= 4 ? p4_row(c, s4, PO, XD) : -1;
    row3 = er >= 8 ? p3_row(c, er - 8, XD) : -1;
    sp3 = row3 >= 0 ? special_index(row3, PO) : -1;
  };
  // root thread of batch row eb (workgroup 0, first 32 threads); the adjoint of (root_pos, root_rot) lives in LDS
  const bool ract = c == 0 && tid < 32 && bact;
  if (ract) {
#pragma unroll
    for (int i = 0; i < 7; ++i) crs[i][eb] = a.carry[eb * 8 + i];
  }
  __syncthreads();

#ifdef ZEGGS_BPSTAT
  unsigned long long wsum[4] = {0, 0, 0, 0};      // 100 MHz ticks this workgroup spent polling, per phase kind
  unsigned long long esum[4] = {0, 0, 0, 0};      // ... and between the end of the products and its arrival, per phase
  unsigned long long e0_ = 0, q4[6] = {0, 0, 0, 0, 0, 0}, q0_ = 0;     // thread 0: P4 epilogue split
#define BPQ0() q0_ = wall_clock64()
#define BPQ(k) do { const unsigned long long n_ = wall_clock64(); q4[k] += n_ - q0_; q0_ = n_; } while (0)
#define BPS0() e0_ = wall_clock64()
#define BPS1(k) esum[k] += wall_clock64() - e0_
#else
#define BPS0() do {} while (0)
#define BPS1(k) do {} while (0)
#define BPQ0() do {} while (0)
#define BPQ(k) do {} while (0)
#endif
  auto wait_phase = [&](long p) {     // all workgroups have finished phase instance p (p < 0: nothing to wait for)
    if (p >= 0) {
#ifdef ZEGGS_BPSTAT
      const unsigned long long w0 = wall_clock64();
#endif
#ifndef ZEGGS_BP_NOPOLL      // (timing experiment, results wrong: nobody waits for anybody -- the step as pure per-CU work)
      if (wave == 1 && !(a.stag ? bp_wait2(a.cnt, (unsigned)(p + 1), a.spin, a.stag) : bp_wait(a.cnt, (unsigned)(p + 1), a.spin, a.nap))) fail = 1;     // (wave 0 of workgroup 0 prepares the root frame meanwhile)
#endif
#ifdef ZEGGS_BPSTAT
      wsum[(p + 1) & 3] += wall_clock64() - w0;
#endif
    }
    __syncthreads();
  };
  auto arrive = [&](long p) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store((gu32*)(a.cnt + c), (unsigned)(p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // wave partial sums -> red[wave][row0 + 4 rg + i][batch]  (the two k-halves of the accumulator lanes are added first)
  auto put = [&](const f4& v, int row0) {
    // swap(a, b) = ([a_lo | b_lo], [a_hi | b_hi]): the sum of the pair is the folded a in lanes < 32 and the folded b in lanes >= 32,
    // so one swap serves two rows (i, i + 2) and every lane has a value to park
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 2]), false, false);
      red[wave][row0 + i + (lane < 32 ? 0 : 2)][lane & 31] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
  };
  auto total = [&](int row) -> float {       // sum over the 8 waves of (row, eb)
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += red[w][row][eb];
    return s;
  };
  // a carry piece computed in the window before a wait: reduced right there (rows 0..3 of red), off the critical path
  auto window_total = [&](const f4& v) -> float {
    put(v, 0);
    __syncthreads();
    return (er < 4 && bact) ? total(er) : 0.f;      // (the next writer of red comes after the barrier of wait_phase)
  };
  // GRU backward of one (unit, batch row): g = total gradient wrt h_t; leaves the gate gradients in the exchange buffer
  // (kind 0 r, 1 z, 2 n, 3 hidden-side n), returns g * z
  auto gru_bwd = [&](float g, const f4& gt, float hp) -> float {
    const float r = gt[0], z = gt[1], nn = gt[2], nh = gt[3];
    const float dn = g * (1.f - z);
    const float dz = g * (hp - nn);
    const float dan = dn * (1.f - nn * nn);
    const float dar = dan * nh * r * (1.f - r);
    const float daz = dz * z * (1.f - z);
    ((float*)&ex[0][eb])[er] = dar; ((float*)&ex[1][eb])[er] = daz; ((float*)&ex[2][eb])[er] = dan;
    ((float*)&ex[3][eb])[er] = dan * r;          // hidden side: only its n rows differ from the input side
    return g * z;
  };
  // ... and 128 threads (gate kind, batch row) store them: canonical DI [B][3H] / DH [B][H] and the operand of the next phase
  auto gru_store = [&](float* DI, float* DHc, float* OP, long t) {
    if (tid < 128 && bact) {
      const int kind = tid >> 5;
      const f4 v = ex[kind][eb];
      stp4(OP + t * (4L * H * 32) + op_idx(eb, kind * H + U0), v);
      if (kind < 3) *(f4*)(DI + t * s3 + (long)eb * 3 * H + kind * H + U0) = v;
      else *(f4*)(DHc + t * sH + (long)eb * H + U0) = v;
    }
  };

  const long OPS = 4L * H * 32;       // floats per step of OP1 / OP0
#ifdef ZEGGS_BP_TWICE
#define BPTW(x) (t + 2 < T ? (long)(x) : 0L)      // the second operand of the timing build: the same part of step t + 1 (written, L2-resident)
#else
#define BPTW(x) 0L
#endif
  f4 w0[1] = {f4{0.f, 0.f, 0.f, 0.f}}, w1[1] = {f4{0.f, 0.f, 0.f, 0.f}};      // window partial sums of carry0 / carry1 (see P1)
  for (int t = T - 1; t >= 1; --t) {
    const long sidx = T - 1 - t, pA = 4 * sidx, pB = pA + 1, pC = pA + 2, pD = pA + 3;
    const f4* op1 = (const f4*)(a.OP1 + (long)t * OPS) + lane;
    const f4* op0 = (const f4*)(a.OP0 + (long)t * OPS) + lane;
    // ================================================================ P1 : dH1 = W2^T dy_t + carry1 -> layer-1 gates
    BPT(0);
    {
      refresh();
      f4 gt = f4{0.f, 0.f, 0.f, 0.f};
      float hp = 0.f;
      if (er < 4 && bact) {
        gt = ((const f4*)a.GT1)[(long)t * sH + (long)eb * H + U];
        hp = a.H1[(long)(t - 1) * sH + (long)eb * H + U];
      }
      // window: carry0 += first half of W_hh0^T (DI0 r,z | dn_h0) of step t+1.  The partial sums of a carry's two windows stay
      // in this wave's accumulator registers and become the START of the main product of the phase that consumes the carry
      // (w0: P2 of this step, w1: P1 of the next one): its reduction adds them up -- no reduction (LDS round trip + barrier)
      // of their own in any window
      w0[0] = f4{0.f, 0.f, 0.f, 0.f};
#ifndef ZEGGS_BP_NOWIN      // (timing experiment: the hand-off latency with empty windows; results are wrong)
      if (t < T - 1) bp_mma<1, NJC, OC0A, false, true>(wr, nullptr, (const f4*)(a.OP0 + (long)(t + 1) * OPS) + lane, wave, 0, 96, w0, BPTW(OPS / 4));
#endif
      f4 acc[1] = {w1[0]};
      wait_phase(pA - 1);
      if (fail) break;
      BPT(1);
      bp_mma<1, NJ1, O1, false>(wr, nullptr, (const f4*)(a.OPY + (long)t * a.KBY * 512) + lane, wave, 0, a.KBY, acc, BPTW(a.KBY * 128L));
      BPT(2);
      BPS0();
      put(acc[0], 0);
      __syncthreads();
      if (er < 4 && bact) c1 = gru_bwd(total(er) + c1, gt, hp);
      __syncthreads();
      gru_store(a.DI1, a.DH1, a.OP1, t);
      BPT(3);
      arrive(pA);
      BPS1(0);
      BPT(4);
    }
    // ================================================================ P2 : dH0 = W_ih1^T DI1_t + carry0 -> layer-0 gates
    {
      refresh();
      f4 gt = f4{0.f, 0.f, 0.f, 0.f};
      float hp = 0.f;
      if (er < 4 && bact) {
        gt = ((const f4*)a.GT0)[(long)t * sH + (long)eb * H + U];
        hp = a.H0[(long)(t - 1) * sH + (long)eb * H + U];
      }
      // window: carry0 += second half of W_hh0^T (.) of step t+1
#ifndef ZEGGS_BP_NOWIN      // (timing experiment: the hand-off latency with empty windows; results are wrong)
      if (t < T - 1) bp_mma<1, NJC, OC0B, false, true>(wr, nullptr, (const f4*)(a.OP0 + (long)(t + 1) * OPS) + lane, wave, 96, 96, w0, BPTW(OPS / 4));
#endif
      f4 acc[1] = {w0[0]};
      wait_phase(pB - 1);
      if (fail) break;
      BPT(5);
      // workgroup 0: everything the root-integration backward of frame t-1 reads (forward outputs and loss gradients:
      // old data), fetched by all threads underneath this phase's products and parked in LDS
      float rv[3] = {0.f, 0.f, 0.f};
      const bool rfetch = c == 0 && t > 1;
      if (rfetch) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {       // item er + 16 q of batch row eb: from the per-step (opaque) copy of the thread index, so that
          const int item = er + 16 * q;     // the compiler does not keep 3 x 7 source addresses per thread live across the whole sweep
          if (item < 33 && bact)
            rv[q] = root_item(a.drpos, a.drrot, a.rrot, a.rpos, a.gaze, a.pose, a.dpose, T, PO, eb, t - 1, true, item);
        }
      }
      bp_mma<1, NJ2A, 0, true>(wr, wl + L3A * 64, op1, wave, 0, 128, acc, BPTW(OPS / 4));
      bp_mma<1, NJ2B, O2B, false>(wr, nullptr, op1, wave, 128, 64, acc, BPTW(OPS / 4));
      if (rfetch) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int item = er + 16 * q;
          if (item < 33) rin[item][eb] = rv[q];
        }
      }
      BPT(6);
      BPS0();
      put(acc[0], 0);
      __syncthreads();
      if (er < 4 && bact) c0 = gru_bwd(total(er) + c0, gt, hp);
      __syncthreads();
      gru_store(a.DI0, a.DH0, a.OP0, t);
      BPT(7);
      arrive(pB);
      BPS1(1);
      BPT(8);
    }
    // ================================================================ P3 : dGin = W_ih0^T DI0_t ; carry0 += W_hh0[r,z]^T DI0_t[r,z]
    {
      refresh();
      float hid = 0.f;
      if (er >= 4 && er < 8 && bact) hid = a.Gin[(long)t * sG + (long)eb * GL + U];
      // window: carry1 += first half of W_hh1^T (DI1_t r,z | dn_h1) -- the operand is one phase old
      w1[0] = f4{0.f, 0.f, 0.f, 0.f};
#ifndef ZEGGS_BP_NOWIN      // (timing experiment: the hand-off latency with empty windows; results are wrong)
      bp_mma<1, NJC, OC1A, false, true>(wr, nullptr, op1, wave, 0, 96, w1, BPTW(OPS / 4));
#endif
      if (ract && t > 1) {      // root thread: the gradient-independent half of the root backward of frame t-1 (in place)
        RootIn ri;
        ri.gather([&](int item) { return rin[item][eb]; });
        RootPre pre;
        root_prepare(d, ri, pre);
#pragma unroll
        for (int q = 0; q < 3; ++q) crs[q][eb] += ri.a[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) crs[3 + q][eb] += ri.e[q];
        root_pre_store(pre, [&](int slot, float v) { rin[slot][eb] = v; });
      }
      f4 acc[3] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
      wait_phase(pC - 1);
      if (fail) break;
      BPT(9);
      bp_mma<3, NJA, 0, true>(wr, wl, op0, wave, 0, 128, acc, BPTW(OPS / 4));          // dhid | dXa slots 0..3 | dXa slots 4..7
      bp_mma<3, NJB, O3B, false>(wr, nullptr, op0, wave, 128, 64, acc, BPTW(OPS / 4));
      BPT(10);
      BPS0();
      put(acc[0], 0); put(acc[1], 4); put(acc[2], 8);
      __syncthreads();
      if (bact && er >= 4) {           // (threads er < 4 carry the GRU items: nothing of theirs in this phase's products)
        if (er < 8) ((float*)&ex[0][eb])[er - 4] = total(er - 4) * d_elu_grad_from_out(hid);
        else if (row3 >= 0) {
          const float v = total(er - 4);
          dxa[er - 8][eb] = v;
          if (sp3 >= 0) stp(a.SP + ((long)t * NSP + sp3) * 32 + eb, v);
        }
      }
      __syncthreads();
      if (tid < 32 && bact) {
        const f4 v = ex[0][eb];
        stp4(a.OPD + (long)t * (H * 32L) + op_idx(eb, U0), v);
        *(f4*)(a.D0 + (long)t * sH + (long)eb * H + U0) = v;
      }
      BPT(11);
      arrive(pC);
      BPS1(2);
      BPT(12);
    }
    // ================================================================ P4 : dx_t = dXa + W0^T D0_t -> DX[t], dy_{t-1}
    {
      refresh();
      float dpo = 0.f;
      if (t > 1 && bact && row4 >= 6 && row4 < PO) dpo = a.dpose[((long)eb * T + t - 1) * PO + row4];
      // window: carry1 += second half of W_hh1^T (DI1_t r,z | dn_h1)
#ifndef ZEGGS_BP_NOWIN      // (timing experiment: the hand-off latency with empty windows; results are wrong)
      bp_mma<1, NJC, OC1B, false, true>(wr, nullptr, op1, wave, 96, 96, w1, BPTW(OPS / 4));
#endif
      f4 acc[3] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
      wait_phase(pD - 1);
      if (fail) break;
      BPT(13);
      float spv = 0.f;                                     // workgroup 0: dXa of the root / gaze columns (other owners)
      if (c == 0 && er >= 4 && bact && (s4 < 6 || (s4 >= 8 && s4 < 11))) spv = a.SP[((long)t * NSP + (s4 < 6 ? s4 : s4 - 2)) * 32 + eb];
      bp_mma<3, NJ4, O4, false>(wr, nullptr, (const f4*)(a.OPD + (long)t * (H * 32L)) + lane, wave, 0, 64, acc, BPTW(H * 8L));
      BPT(14);
      BPS0();
      BPQ0();
      put(acc[0], 0); put(acc[1], 4); put(acc[2], 8);
      __syncthreads();
      BPQ(0);
      if (bact && er >= 4) {
        const float v = total(er - 4);
        {
          float gy = 0.f;
          if (c == 0 && (s4 < 6 || s4 >= 8)) {               // root / gaze columns: to the root thread of the batch row
            if (s4 < 11) sp9[s4 < 6 ? s4 : s4 - 2][eb] = v + spv;
          } else if (row4 >= 0) {
            const float dx = v + dxa[s4][eb];
            if (row4 >= PI) a.DX[((long)t * B + eb) * XD + row4] = dx;       // speech / style columns
            else if (row4 < PO) gy = (dpo + dx / si4) * so4;                   // pose columns -> dy_{t-1}
          }
          ((float*)&ex[s4 >> 2][eb])[s4 & 3] = gy;
        }
      }
      __syncthreads();
      BPQ(1);
      if (t > 1) {
        float* dyc = a.DY + ((long)(t - 1) * B + eb) * POL;
        float* opy = a.OPY + (long)(t - 1) * a.KBY * 512;
        if (c != 0) {
          const int g = er, rb = 8 * c + 4 * g;                // groups 0, 1 of this workgroup's rows (er: the per-step thread index)
          if (er < 2 && bact && rb < PO) {
            const f4 v = ex[g][eb];
            stp4(opy + op_idx(eb, rb), v);
            *(f4*)(dyc + rb) = v;
          }
        } else {
          BPT(17);
          if (ract) {
            float g6[6], dgd[3], cr[7];
#pragma unroll
            for (int q = 0; q < 7; ++q) cr[q] = crs[q][eb];
            auto pre = [&](int slot) { return rin[slot][eb]; };
#pragma unroll
            for (int q = 0; q < 6; ++q) g6[q] = pre(27 + q) + sp9[q][eb] * rst[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) dgd[q] = sp9[6 + q][eb];
            BPT(18);
            root_apply(d, rst + 6, pre, dgd, cr, g6);
            BPT(19);
#pragma unroll
            for (int q = 0; q < 7; ++q) crs[q][eb] = cr[q];
            const f4 v0 = f4{g6[0] * rst[9], g6[1] * rst[10], g6[2] * rst[11], g6[3] * rst[12]};
            const f4 e1 = ex[1][eb];                           // rows 6, 7 are ordinary pose columns
            const f4 v1 = f4{g6[4] * rst[13], g6[5] * rst[14], e1[2], e1[3]};
            stp4(opy + op_idx(eb, 0), v0); stp4(opy + op_idx(eb, 4), v1);
            *(f4*)dyc = v0; *(f4*)(dyc + 4) = v1;
          }
        }
      }
      BPT(15);
      BPQ(2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      BPQ(3);
      arrive(pD);
      BPQ(4);
      BPS1(3);
      BPT(16);
    }
  }
  if (!fail) {          // the carry0 pieces of step 1 (their windows would have been in a step 0), the carry1 windows of step 1
    f4 acct[1] = {f4{0.f, 0.f, 0.f, 0.f}};
    bp_mma<1, NJC, OC0A, false, true>(wr, nullptr, (const f4*)(a.OP0 + OPS) + lane, wave, 0, 96, acct);
    bp_mma<1, NJC, OC0B, false, true>(wr, nullptr, (const f4*)(a.OP0 + OPS) + lane, wave, 96, 96, acct);
    c0 += window_total(acct[0]);
    __syncthreads();
    c1 += window_total(w1[0]);
  }
  if (!fail && er < 4 && bact) {     // gradients wrt the initial hidden states (CellStateEncoder backward)
    a.dH1c[(long)eb * H + U] = c1;
    a.dH0c[(long)eb * H + U] = c0;
  }
#ifdef ZEGGS_BPSTAT
  if (tid == 64) {      // wave 1 polls
    unsigned long long* o = (unsigned long long*)(a.err + 512) + 4 * c;
    for (int i = 0; i < 4; ++i) o[i] = wsum[i];
    unsigned long long* o2 = (unsigned long long*)(a.err + 512) + 4 * 256 + 4 * c;
    for (int i = 0; i < 4; ++i) o2[i] = esum[i];
  }
  if (tid == 0) {
    unsigned long long* o3 = (unsigned long long*)(a.err + 512) + 8 * 256 + 8 * c;
    for (int i = 0; i < 6; ++i) o3[i] = q4[i];
  }
#endif
  if (fail) {     // a bounded wait gave up: error word, the caller's sticky status, NaN in the carries the CellStateEncoder backward reads
    if (tid == 0) {
      atomicOr(a.err, 1u);
      if (a.status) atomicOr(a.status, ZEGGS_GAVE_UP_BPTT);
    }
    if (er < 4 && bact) {
      const float qnan = __uint_as_float(0x7fc00000u);
      a.dH1c[(long)eb * H + U] = qnan;
      a.dH0c[(long)eb * H + U] = qnan;
    }
  }
}

// ---------------------------------------------------------------- weight tiles (once per optimizer step)
struct BPackArgs {
  float *PWR, *PWL;
  const float *w_ih0, *w_hh0, *w_ih1, *w_hh1, *l2_w, *l0_w;
  int XD, PO, KBY;
};
// weight of output row i (0..3) of a tile for contraction index k = 16 kb + kk of its operand (workgroup c).  Tile kinds:
//   0 dH1 (W2^T)   1 dH0 (W_ih1^T)   2 carry1 (W_hh1^T)   3 carry0 (W_hh0^T)   4 dx slots 4 rg + i (W0^T)
//   5 dhid (W_ih0^T, hid columns)   6 / 7 dXa slots 0..3 / 4..7 (W_ih0^T, x columns)
// the operands of kinds 2 / 3 are [DI (3H) | dn_h (H)]: k < 2H are the r, z rows of W_hh, k >= 3H its n rows
__device__ __forceinline__ float bp_value(const BPackArgs& p, int kind, int c, int rg, int i, int kb, int kk) {
  const int H = BH, U = 4 * c + i, k = 16 * kb + kk;
  const long ld0 = H + p.XD;
  switch (kind) {
    case 0: return k < p.PO ? p.l2_w[(long)k * H + U] : 0.f;
    case 1: return p.w_ih1[(long)k * H + U];
    case 2: return p.w_hh1[(long)(k < 2 * H ? k : k - H) * H + U];
    case 3: return p.w_hh0[(long)(k < 2 * H ? k : k - H) * H + U];
    case 4: { const int row = p4_row(c, 4 * rg + i, p.PO, p.XD); return row >= 0 ? p.l0_w[(long)k * p.XD + row] : 0.f; }
    case 5: return p.w_ih0[(long)k * ld0 + U];
    default: { const int row = p3_row(c, 4 * (kind - 6) + i, p.XD); return row >= 0 ? p.w_ih0[(long)k * ld0 + H + row] : 0.f; }
  }
}
__global__ void bp_pack_k(BPackArgs p) {
  const long nr = (long)BNCU * 8 * NWR * 64, nl = (long)BNCU * 8 * L3 * 64;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < nr + nl; idx += (long)gridDim.x * blockDim.x) {
    const bool lds = idx >= nr;
    const long r = lds ? idx - nr : idx;
    const int lane = (int)(r & 63);
    const long cws = r >> 6;
    const int per = lds ? L3 : NWR;
    const int slot = (int)(cws % per), wave = (int)((cws / per) & 7), c = (int)(cws / (8L * per));
    int kind, j, rg = 0, kb0, nblk;
    bool skipn = false;
    if (lds && slot < L3A) {   // P3, blocks 0..127 of DI0: dhid | dXa 0..3 | dXa 4..7
      j = slot / 3; rg = slot % 3; kb0 = 0; nblk = 128; kind = 5 + rg;
    } else if (lds) { kind = 1; j = slot - L3A; kb0 = 0; nblk = 128; }          // P2, blocks 0..127 of DI1
    else if (slot < O2B) { kind = 0; j = slot - O1; kb0 = 0; nblk = p.KBY; }
    else if (slot < OC1A) { kind = 1; j = slot - O2B; kb0 = 128; nblk = 64; }
    else if (slot < OC1B) { kind = 2; j = slot - OC1A; kb0 = 0; nblk = 96; skipn = true; }
    else if (slot < O3B) { kind = 2; j = slot - OC1B; kb0 = 96; nblk = 96; skipn = true; }
    else if (slot < OC0A) { j = (slot - O3B) / 3; rg = (slot - O3B) % 3; kind = 5 + rg; kb0 = 128; nblk = 64; }
    else if (slot < OC0B) { kind = 3; j = slot - OC0A; kb0 = 0; nblk = 96; skipn = true; }
    else if (slot < O4) { kind = 3; j = slot - OC0B; kb0 = 96; nblk = 96; skipn = true; }
    else { kind = 4; j = (slot - O4) / 3; rg = (slot - O4) % 3; kb0 = 0; nblk = 64; }
    const int e = wave + 8 * j;
    float v = 0.f;
    if (e < nblk) {
      int kb = kb0 + e;
      if (skipn) kb = kb < 128 ? kb : kb + 64;
      v = bp_value(p, kind, c, rg, lane & 3, kb, lane >> 2);
    }
    (lds ? p.PWL : p.PWR)[r] = v;
  }
}

// canonical [B, ld] -> operand layout
__global__ void bp_to_op_k(float* op, const float* src, long ld, int K, int B) {
  const long n = (long)B * K;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % K), b = (int)(i / K);
    op[op_idx(b, k)] = src[(long)b * ld + k];
  }
}
// gradient wrt the raw output of the LAST frame (no next step feeds on it) + the root adjoint it leaves behind
__global__ void bp_dy_last_k(ZeggsDecDims d, ZeggsDecStats st, const float* dpose, const float* drpos, const float* drrot,
                             const float* gaze, const float* pose, const float* rpos, const float* rrot, float* carry,
                             float* dy, int POL) {
  const int b = blockIdx.x, t = d.T - 1;
  const float* dpb = dpose + ((long)b * d.T + t) * d.PO;
  for (int cc = 6 + threadIdx.x; cc < d.PO; cc += blockDim.x) dy[(long)b * POL + cc] = dpb[cc] * st.out_std[cc];
  if (threadIdx.x == 0) {
    float g6[6], cr[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dgd[3] = {0.f, 0.f, 0.f};
    RootIn ri;
    ri.gather([&](int item) { return root_item(drpos, drrot, rrot, rpos, gaze, pose, dpose, d.T, d.PO, b, t, false, item); });
    for (int q = 0; q < 6; ++q) g6[q] = dpb[q];
    root_bwd_reg(d, st.in_std + d.PO, ri, dgd, cr, g6, false);
    for (int q = 0; q < 6; ++q) dy[(long)b * POL + q] = g6[q] * st.out_std[q];
    for (int q = 0; q < 7; ++q) carry[b * 8 + q] = cr[q];
  }
}

}  // namespace

int dec_bp_supported(const ZeggsDecDims& d, const DecWs& w) {
  return !d.film && d.H == BH && d.B <= 64 && d.T >= 3 && d.PI == d.PO + 3 && d.PO >= 16 && (d.PO + 15) / 16 <= 8 * NJ1 &&
         w.XD <= 8 * BNCU && w.bp_wr != nullptr;
}
int dec_bp_state() { return g_bp_ok; }
void dec_bp_set_state(int v) { g_bp_ok = v; }

// the whole sweep t = T-1 .. 1; leaves DY, DI*, DH*, D0, DX (speech / style columns), dH0c, dH1c as the stage sweep does
// what depends on the weights and the dimensions only (zeggs_decoder_prepare runs it ahead of the forward, on a second stream)
int dec_bp_pack(const ZeggsDecDims& d, const ZeggsDecParams* P, DecWs& w, hipStream_t s) {
  const int T = d.T, KBY = (d.PO + 15) / 16;
  BPackArgs p{w.bp_wr, w.bp_wl, P->w_ih0, P->w_hh0, P->w_ih1, P->w_hh1, P->l2_w, P->l0_w, w.XD, d.PO, KBY};
  hipLaunchKernelGGL(bp_pack_k, dim3(8192), dim3(256), 0, s, p);
  ZLAUNCH_CHECK("bp_pack");
  // the pad k rows of dy (PO .. 16 KBY) must be finite: zero the operand once; every other operand element that is read
  // with a non-zero weight is written by the sweep (pad batch lanes only ever feed pad batch columns)
  ZTRY(k_fill(w.bp_opy, (long)T * KBY * 512, 0.f, s));
  return 0;
}
int dec_bp_zero_slots(DecWs& w, hipStream_t s) {        // arrival slots + error word
  return k_fill((float*)w.bp_cnt, 2048, 0.f, s);
}
int dec_bp_run(const ZeggsDecDims& d, const ZeggsDecParams* P, const ZeggsDecStats* st, DecWs& w, const float* gaze,
               const float* pose, const float* rpos, const float* rrot, const float* dpose, const float* drpos,
               const float* drrot, hipStream_t s, bool packed, unsigned* status) {
  const int B = d.B, T = d.T, H = d.H, KBY = (d.PO + 15) / 16;
  int dev = 0, ncu = 0;
  ZCHECK(hipGetDevice(&dev) == hipSuccess, "hipGetDevice failed");
  ZCHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess, "device query failed");
  ZCHECK(ncu >= BNCU, "persistent BPTT sweep needs %d CUs (device has %d)", BNCU, ncu);
  if (!packed) ZTRY(dec_bp_pack(d, P, w, s));
  float* dyl = w.DY + (long)(T - 1) * B * w.POL;
  hipLaunchKernelGGL(bp_dy_last_k, dim3(B), dim3(256), 0, s, d, *st, dpose, drpos, drrot, gaze, pose, rpos, rrot, w.carry,
                     dyl, w.POL);
  ZLAUNCH_CHECK("bp_dy_last");
  if (!packed) ZTRY(dec_bp_zero_slots(w, s));         // (prepared: zeggs_decoder_prepare has zeroed them)
  dec_timing_mark(2, s);
  // batch rows are independent in the sweep: 33..64 rows run as two sweeps of <= 32 rows through the same operand buffers
  for (int b0 = 0; b0 < B; b0 += 32) {
    const int Bact = B - b0 < 32 ? B - b0 : 32;
    const long o = b0;
    if (b0) ZTRY(k_fill((float*)w.bp_cnt, 256, 0.f, s));   // the arrival slots start over; the error word accumulates
    hipLaunchKernelGGL(bp_to_op_k, dim3((Bact * d.PO + 255) / 256), dim3(256), 0, s, w.bp_opy + (long)(T - 1) * KBY * 512,
                       dyl + o * w.POL, (long)w.POL, d.PO, Bact);
    ZLAUNCH_CHECK("bp_prologue");
    BArgs a;
    memset(&a, 0, sizeof(a));
    a.d = d; a.st = *st; a.XD = w.XD; a.GL = w.GL; a.POL = w.POL; a.KBY = KBY; a.Bact = Bact;
    a.PWR = w.bp_wr; a.PWL = w.bp_wl;
    a.OPY = w.bp_opy; a.OP1 = w.bp_op1; a.OP0 = w.bp_op0; a.OPD = w.bp_opd; a.SP = w.bp_sp;
    a.Gin = w.Gin + o * w.GL; a.H0 = w.H0 + o * H; a.H1 = w.H1 + o * H; a.GT0 = w.GT0 + o * H * 4; a.GT1 = w.GT1 + o * H * 4;
    a.DY = w.DY + o * w.POL; a.DI1 = w.DI1 + o * 3 * H; a.DH1 = w.DH1 + o * H; a.DI0 = w.DI0 + o * 3 * H; a.DH0 = w.DH0 + o * H;
    a.D0 = w.D0 + o * H; a.DX = w.DX + o * w.XD; a.dH0c = w.dH0c + o * H; a.dH1c = w.dH1c + o * H;
    a.dpose = dpose + o * T * d.PO; a.drpos = drpos + o * T * 3; a.drrot = drrot + o * T * 4; a.gaze = gaze + o * T * 3;
    a.pose = pose + o * T * d.PO; a.rpos = rpos + o * T * 3; a.rrot = rrot + o * T * 4;
    a.carry = w.carry + o * 8;
    a.cnt = w.bp_cnt; a.err = w.bp_cnt + 1024;
    a.status = status; a.spin = (unsigned)g_persistent_spin; a.nap = (unsigned)g_poll_sleep; a.stag = (unsigned)g_poll_stagger;
    hipLaunchKernelGGL(train_bwd_persistent_k, dim3(BNCU), dim3(BTHR), 0, s, a);
    ZLAUNCH_CHECK("train_bwd_persistent");
  }
  dec_timing_mark(3, s);
  (void)H;
  return 0;
}
extern "C" int zeggs_bp_stamps(const ZeggsDecDims* dp, void* ws, size_t ws_bytes, unsigned long long* out /* [3][2][32] */) {
  Arena a(ws, ws_bytes);
  DecWs w = carve_dec(*dp, 1, a);
  ZCHECK(a.ok() && w.bp_cnt, "bp_stamps: workspace");
  ZCHECK(hipMemcpy(out, w.bp_cnt + 1024 + 32, 3 * 2 * 32 * 8, hipMemcpyDeviceToHost) == hipSuccess, "copy");
  return 0;
}
// -DZEGGS_BPSTAT builds: 100 MHz ticks every workgroup spent polling for the hand-off INTO phase P1..P4, summed over the sweep
extern "C" int zeggs_bp_waits(const ZeggsDecDims* dp, void* ws, size_t ws_bytes, unsigned long long* out /* host [4][256][4] */) {
  Arena a(ws, ws_bytes);
  DecWs w = carve_dec(*dp, 1, a);
  ZCHECK(a.ok() && w.bp_cnt, "bp_waits: workspace");
  ZCHECK(hipMemcpy(out, w.bp_cnt + 1024 + 512, 4 * 256 * 4 * 8, hipMemcpyDeviceToHost) == hipSuccess, "copy");
  return 0;
}
int dec_bp_errptr(const DecWs& w, unsigned** out) {
  *out = w.bp_cnt + 1024;
  return 0;
}
int dec_bp_errors(const DecWs& w, unsigned* out) {
  ZCHECK(hipMemcpy(out, w.bp_cnt + 1024, sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess,
         "persistent BPTT sweep: error word copy failed");
  return 0;
}
