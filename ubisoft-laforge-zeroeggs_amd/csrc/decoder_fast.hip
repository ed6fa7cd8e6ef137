// Fast path of the decoder step: weight-streaming MFMA stage kernels over fragment-packed operands.
//
// One decoder step is a chain of dependent matrix stages (layer0 -> GRU l0 -> GRU l1 -> layer2)
// with M = batch <= 64 rows: at B = 32 every stage is bound by streaming its weights once
// (75.7 MB per step).  layer2 of step t and layer0 of step t+1 are folded into one
// launch (dec_fast_pack_merged), and so are their backward counterparts: 3 launches per step in each
// direction.  Design for gfx950:
//   * weights are re-packed once per optimizer step into MFMA *fragment order*: for every tile of
//     16 "virtual output columns" and every block of 16 k-values, 64 lanes x float4 = one contiguous
//     1 KiB block, so each wave-level load is a perfectly coalesced global_load_dwordx4 stream
//     straight into VGPRs (no LDS round trip for a once-read operand);
//   * activations are exchanged between stages in the matching B-operand fragment order (2 KiB per
//     k-block at B = 32), written by the producing stage's epilogue next to the canonical row-major
//     copy that the weight-gradient GEMMs need;
//   * v_mfma_f32_16x16x4_f32: A = 16 weight rows, B = 16 batch columns; a float4 per lane feeds 4 MFMAs;
//   * one workgroup (8 waves) owns one 16-column tile for the FULL contraction: the waves split K,
//     reduce through LDS, and the epilogue (bias, ELU, GRU gate math, pose integration, next-step
//     vectorisation, and their backward counterparts) runs in the same launch: no cross-workgroup
//     partial sums, no grid barrier (tools/barrier_probe.hip: 6.4 us vs 2.6 us for a kernel boundary);
//   * GRU tiles interleave the r, z, n rows of 5 hidden units (15 of 16 columns) so the gate math
//     needs no second pass; the input-side and hidden-side products use two accumulator sets.
// Backward uses the transposed packs (16 output units x contraction over gate rows).
#include "decoder_ws.h"
#include "dec_math.h"
#include "gemm.h"
#include "kernels.h"
#include <mutex>

int g_stage_variant = 0;
int g_chain = 0;   // zeggs_set_option("chain", 1): chained (run-ahead) stage launches, see struct Chain
int g_launch_window = 8;   // zeggs_set_option("launch_window", steps): see struct LaunchWindow (measured: 8 -> sweeps 6-9 % faster than unbounded)

// zeggs_set_option("timing", 1): HIP events on the caller's stream around the steady-state stage sweeps (the 3-launch
// steps only, not the per-call packs), read back by zeggs_timing_ms -- bench.py's roofline figures
int g_timing = 0;
static hipEvent_t g_tev[4];
static bool g_tev_ok = false;
static void timing_mark(int i, hipStream_t s) {
  if (!g_timing) return;
  if (!g_tev_ok) {
    for (int k = 0; k < 4; ++k) (void)hipEventCreate(&g_tev[k]);
    g_tev_ok = true;
  }
  (void)hipEventRecord(g_tev[i], s);
}
extern "C" int zeggs_timing_ms(int which, float* ms) {
  ZCHECK(which == 0 || which == 1, "timing: which = 0 (forward sweep) or 1 (backward sweep)");
  ZCHECK(g_tev_ok, "timing: no sweep was recorded (zeggs_set_option(\"timing\", 1) first)");
  ZCHECK(hipEventSynchronize(g_tev[2 * which + 1]) == hipSuccess, "timing: event not recorded");
  ZCHECK(hipEventElapsedTime(ms, g_tev[2 * which], g_tev[2 * which + 1]) == hipSuccess, "timing: events incomplete");
  return 0;
}

namespace {

// zeggs_set_option("stage_variant", bits) -- measurement switches, all off in production:
//   2 no main loop / 4 no epilogue (ablation: results are garbage), 32 never / 64 always split stages over the batch,
//   128 16-wave workgroups, 1024 MFMA path for B <= 2 (no GEMV kernels), 4096 / 8192 unmerged forward / backward stages,
//   16384 every forward stage launched twice (cold vs L2-warm weights, tools/warm_probe.sh)
enum { V_NOW = 2, V_NOEPI = 4 };

// (forward epilogues first: launch_stage tells the two kernel families apart by `epi >= EPI_GRU_BWD`)
enum { EPI_ELU_HID = 0, EPI_GRU_FWD, EPI_OUT_FWD, EPI_HID_MERGED, EPI_ELU_FILM, EPI_GRU_BWD, EPI_ADD, EPI_DGIN, EPI_DX,
       EPI_GRU_BWD_M, EPI_FILM_BWD, EPI_FILM_BWD_M };

struct Seg {
  const float* w;   // packed weights  [tile][kb][64][4]
  const float* x;   // packed activations [kb][NB][64][4]
  int kb, acc;
  const float* xc;  // GEMV mode (batch <= 4, inference): the same activations in canonical row-major form
  int ldx;
  int fixed;        // 1: every workgroup streams weight tile 0 of this pack (the root columns of layer2)
  int tkb;          // k-blocks per tile in the pack when this segment is a k-sub-range of it (0: = kb; w is pre-offset)
};
struct Grp {
  Seg seg[3];
  int nseg, tiles, epi;
  // chained GEMV: the per-tile contiguous pack that holds all segments' blocks in order (tkbcat blocks per tile); blocks
  // from index kacc on accumulate into set 1
  const float* wcat;
  int tkbcat, kacc;
  const float *p0, *p1, *p2, *p3, *p4;
  float *o0, *o1, *o2, *o3, *o4, *o5;
  int ld0;   // row stride of o0 in the ELU epilogues (0: a.GL, the [hid | x] rows of Gin)
};
struct StageArgs {
  Grp g[2];
  ZeggsDecDims d;
  ZeggsDecStats st;
  int NB, t, GL, XD, POL;
  const float *gaze, *speech, *style;
  float *pose, *rpos, *rrot;               // forward outputs
  const float *cpose, *crpos, *crrot;      // backward: forward results
  const float *dpose, *drpos, *drrot;      // backward: loss gradients
  float* carry;                            // [B,8] root-state gradient carry (read)
  float* carry_out;                        // [B,8] carry written by this launch (double-buffered: other workgroups read `carry`)
  const float *aux0, *aux1;                // merged backward stage: dXa [B,XD], layer2 weight [PO,H]
  float* rxf;                              // DGIN epilogue: fragment of r = sigma_o dpose[t-1] + (sigma_o/sigma_i) dXa (pose columns >= 6)
  int variant;                             // ablation switches (tools/stage_bench.py); 0 in production
  int gemv;                                // 1: tiny-batch decode, VALU dot products over canonical activations
  // speech/style columns of x_{t+1}, staged by the GRU layer-1 launch (all null: nothing to stage)
  float *cf_gin, *cf_x, *cf_cond;
  // chained (run-ahead) launches: this launch was enqueued BEFORE its predecessor finished (alternating streams).  It
  // pre-loads its weights, then waits until the predecessor's arrival counters (8 shards, one 128-byte line each)
  // sum to wait_count, acquires, and only then touches activations; at its end every workgroup arrives on its own counters.
  unsigned *ch_wait, *ch_arrive, *ch_err;
  unsigned wait_count;
  int ch_k;     // index of this launch in the chain (timestamps of the -DZEGGS_CHTIME build)
};

// -DZEGGS_CHTIME: wall-clock (100 MHz) stamps of the phases of a chained launch, first and last workgroup, for the last 16
// launches of a rollout (tools/chain_time.py reads them from the workspace) -- measurement builds only
#ifdef ZEGGS_CHTIME
#define CHT(i)                                                                                                   \
  do {                                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    if (a.ch_err && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))                        \
      ((unsigned long long*)(a.ch_err + 32))[(((a.ch_k & 15) * 2 + (blockIdx.x != 0)) * 16) + (i)] = wall_clock64(); \
  } while (0)
#else
#define CHT(i)
#endif

// ---- hand-off between chained launches (cdna_hip_programming.md, Guideline 16, counter form): payload stores are
// write-through (agent-scope relaxed atomic stores lower to `global_store ... sc1`), every storing wave drains, ONE lane
// per workgroup arrives; the consumer polls relaxed from 8 lanes (one shard each), ONE acquire, then plain loads.
enum { CH_SHARDS = 8, CH_STRIDE = 32, CH_RING = 4, CH_SPIN_MAX = 1 << 22, CH_GPRE = 26, CH_GREG = 20 };
typedef __attribute__((address_space(1))) unsigned gu32;
__device__ __forceinline__ void st_pub(float* p, float v) {       // published (consumed by the NEXT chained launch)
  __hip_atomic_store((gu32*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int CH> __device__ __forceinline__ void st_out(float* p, float v) {
  if constexpr (CH) st_pub(p, v); else *p = v;
}
__device__ __forceinline__ void chain_wait(const unsigned* flags, unsigned expect, unsigned* err) {
  // called by wave 0 only (all 64 lanes); lanes 0..7 poll one shard each
  const int lane = threadIdx.x & 63;
  unsigned spins = 0;
  for (;;) {
    unsigned v = lane < CH_SHARDS ? __hip_atomic_load((gu32*)(flags + lane * CH_STRIDE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    v = __builtin_amdgcn_readfirstlane(v);
    if (v >= expect) break;
    if (++spins > CH_SPIN_MAX) {      // bounded: never hang the GPU; the host reads the error word
      if (lane == 0) atomicOr(err, 1u);
      break;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__device__ __forceinline__ void chain_arrive(unsigned* flags) {
  // every storing wave drains its write-through stores, then ONE lane of the workgroup arrives
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_fetch_add((gu32*)(flags + (blockIdx.x & (CH_SHARDS - 1)) * CH_STRIDE), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// column permutation of the dX stage: tile 0 holds root_vel/vrt (0..5) AND the gaze columns (PO..PO+2)
__host__ __device__ inline int perm_dx(int q, int PO) {
  if (q < 6) return q;
  if (q < 9) return PO + (q - 6);
  if (q < PO + 3) return q - 3;
  return q;
}

__device__ __forceinline__ long xf_index(int b, int k, int NB) {
  return ((((long)(k >> 4) * NB + (b >> 4)) * 64 + ((((k >> 2) & 3) << 4) | (b & 15))) << 2) | (k & 3);
}

// limiter experiments (tools/stage_bench.py with an alternative build): -DZEGGS_NOX re-reads ONE activation block
// (no activation traffic), -DZEGGS_NOMFMA replaces the matrix op by one VALU add (results are garbage, timing only)
#ifdef ZEGGS_NOX
#define ZXSEL(kb) ((kb) & 0)
#else
#define ZXSEL(kb) (kb)
#endif
#ifdef ZEGGS_NOMFMA
#define ZMAC(A, w, x) A[0] += (w) + (x)
#else
#define ZMAC(A, w, x) A = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x, A, 0, 0, 0)
#endif

// UG = k-blocks per register buffer (0: the build's default ZEGGS_U)
template <int NB, int UG = 0>   // NB here = batch blocks handled by this workgroup; LNB = batch blocks in the fragment layout
__device__ __forceinline__ void run_blocks(const f4* __restrict__ wp, const f4* __restrict__ xp, int lo, int hi,
                                           f4 (&acc)[NB], int LNB) {
  // Software pipeline with two register buffers: the loads of group g+1 are in flight while group g feeds the
  // matrix cores.  The steady-state loop is branch-free (the look-ahead index is clamped, a redundant reload of
  // the last group is cheaper than a branch that would force s_waitcnt vmcnt(0)).
#ifndef ZEGGS_U
#define ZEGGS_U 2
#endif
  constexpr int U = UG ? UG : ZEGGS_U;
  const int n = hi - lo, ng = n / U;
  f4 w0[U], x0[U][NB], w1[U], x1[U][NB];
#define ZLOAD(W, X, G)                                                                         \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                              \
    const long kb_ = lo + (long)(G) * U + u;                                                   \
    W[u] = wp[kb_ * 64];                                                                       \
    _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) X[u][nb] = xp[(ZXSEL(kb_) * LNB + nb) * 64];      \
  }
#define ZCOMP(W, X)                                                                            \
  _Pragma("unroll") for (int u = 0; u < U; ++u)                                                \
    _Pragma("unroll") for (int c = 0; c < 4; ++c)                                              \
      _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                        \
        ZMAC(acc[nb], W[u][c], X[u][nb][c]);
  if (ng > 0) {
    ZLOAD(w0, x0, 0)
    int g = 0;
    for (; g + 1 < ng; g += 2) {
      ZLOAD(w1, x1, g + 1)
      ZCOMP(w0, x0)
      const int gn = (g + 2 < ng) ? g + 2 : ng - 1;
      ZLOAD(w0, x0, gn)
      ZCOMP(w1, x1)
    }
    if (ng & 1) { ZCOMP(w0, x0) }
  }
  for (int kb = lo + ng * U; kb < hi; ++kb) {
    f4 wv = wp[(long)kb * 64], xv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) xv[nb] = xp[((long)kb * LNB + nb) * 64];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c], xv[nb][c], acc[nb], 0, 0, 0);
  }
#undef ZLOAD
#undef ZCOMP
}

// backward of the root integration of frame `f` (see dec_devec_bwd_k in decoder.hip for the derivation).
// g6: in = dpose[f][0:6] + dx/sigma_i (when a next step exists); out = total grad wrt pose[f][0:6] (de-normalised).
__device__ void root_bwd(const ZeggsDecDims& d, const ZeggsDecStats& st, int b, int f, bool has_next, const float* dgd_in,
                         const float* gaze, const float* pose, const float* rpos, const float* rrot, const float* drpos,
                         const float* drrot, const float* carry, float* carry_out, float (&g6)[6]) {
  const float* cr = carry + b * 8;
  const float* a = drpos + ((long)b * d.T + f) * 3;
  const float* e = drrot + ((long)b * d.T + f) * 4;
  V3 g_rp = v3(cr[0] + a[0], cr[1] + a[1], cr[2] + a[2]);
  Q4 g_rr = Q4{cr[3] + e[0], cr[4] + e[1], cr[5] + e[2], cr[6] + e[3]};
  const float* rq = rrot + ((long)b * d.T + f) * 4;
  const float* rp = rpos + ((long)b * d.T + f) * 3;
  Q4 q_t = Q4{rq[0], rq[1], rq[2], rq[3]};
  V3 p_t = v3(rp[0], rp[1], rp[2]);
  if (has_next) {
    const float* gz = gaze + ((long)b * d.T + f + 1) * 3;
    V3 dgd = v3(dgd_in[0] / st.in_std[d.PO], dgd_in[1] / st.in_std[d.PO + 1], dgd_in[2] / st.in_std[d.PO + 2]);
    Q4 dqi; V3 dv;
    qmv_bwd(quat_inv(q_t), v3(gz[0], gz[1], gz[2]) - p_t, dgd, dqi, dv);
    g_rr.w += dqi.w; g_rr.x -= dqi.x; g_rr.y -= dqi.y; g_rr.z -= dqi.z;
    g_rp = g_rp - dv;
  }
  const float* pq = rrot + ((long)b * d.T + f - 1) * 4;
  Q4 q_p = Q4{pq[0], pq[1], pq[2], pq[3]};
  const float* pt = pose + ((long)b * d.T + f) * d.PO;
  V3 vel = v3(pt[0], pt[1], pt[2]), vrt = v3(pt[3], pt[4], pt[5]);
  Q4 dq1; V3 dv1;
  qmv_bwd(q_p, d.dt * vel, g_rp, dq1, dv1);
  V3 u = quat_mul_vec(q_p, d.dt * vrt);
  QExpCtx ec;
  Q4 E = quat_exp_ctx(0.5f * u, ec);
  Q4 dE, dqy;
  qmul_bwd(E, q_p, g_rr, dE, dqy);
  V3 du = 0.5f * qexp_bwd_ctx(0.5f * u, dE, ec);
  Q4 dq2; V3 dv2;
  qmv_bwd(q_p, d.dt * vrt, du, dq2, dv2);
  g6[0] += d.dt * dv1.x; g6[1] += d.dt * dv1.y; g6[2] += d.dt * dv1.z;
  g6[3] += d.dt * dv2.x; g6[4] += d.dt * dv2.y; g6[5] += d.dt * dv2.z;
  if (carry_out) {
    float* co = carry_out + b * 8;
    co[0] = g_rp.x; co[1] = g_rp.y; co[2] = g_rp.z;
    co[3] = dq1.w + dqy.w + dq2.w; co[4] = dq1.x + dqy.x + dq2.x;
    co[5] = dq1.y + dqy.y + dq2.y; co[6] = dq1.z + dqy.z + dq2.z;
  }
}

// root integration of frame t from the de-normalised root velocities p6 (reference: devectorize_output,
// ZEGGS/modules.py:139-176) and, when a next step exists, the normalised gaze direction of x_{t+1}.
// rt = [rrot_{t-1}(4) | rpos_{t-1}(3) | gaze_{t+1}(3)]
__device__ __forceinline__ void root_step(const ZeggsDecDims& d, const ZeggsDecStats& st, const float (&rt)[10],
                                          const float (&p)[6], bool next, V3& npos, Q4& nq, float (&genc)[3]) {
  const Q4 q = Q4{rt[0], rt[1], rt[2], rt[3]};
  const V3 pos = v3(rt[4], rt[5], rt[6]);
  npos = quat_mul_vec(q, d.dt * v3(p[0], p[1], p[2])) + pos;
  const V3 uu = quat_mul_vec(q, d.dt * v3(p[3], p[4], p[5]));
  nq = quat_exp_mul(0.5f * uu, q);
  if (next) {
    const V3 gd = quat_mul_vec(quat_inv(nq), v3(rt[7], rt[8], rt[9]) - npos);
    genc[0] = (gd.x - st.in_mean[d.PO]) / st.in_std[d.PO];
    genc[1] = (gd.y - st.in_mean[d.PO + 1]) / st.in_std[d.PO + 1];
    genc[2] = (gd.z - st.in_mean[d.PO + 2]) / st.in_std[d.PO + 2];
  }
}

// FAM 0: forward epilogues, 1: backward; BV > 0: GEMV mode for BV rows; CH: chained (run-ahead) launch
template <int NB, int FAM, int WAVES, int BV = 0, int CH = 0>
__global__ __launch_bounds__(WAVES * 64, CH ? 4 : WAVES / 4) void stage_k(StageArgs a) {
  constexpr int NTHR = WAVES * 64;
  static_assert(16 * 16 * NB <= NTHR, "one epilogue item per thread");
  static_assert(!CH || (FAM == 0 && WAVES == 8), "chained launches: forward stages with 8 waves");
  // chained GEMV: a wave parks ALL its weight blocks on chip before the hand-off -- GREG in registers, the rest in LDS --
  // and stages its activation slice through LDS; the MFMA reduction scratch shrinks to what the GEMV reduce needs
  constexpr bool CG = CH && BV > 0;
  constexpr int GPRE = CH_GPRE, GREG = CH_GREG;
  __shared__ f4 red[CG ? 1 : WAVES][2][NB][64];
  __shared__ f4 fin[2][NB][64];
  __shared__ f4 xs[CG ? WAVES * BV * GPRE * 4 : 1];            // wave-private activation slices
  __shared__ f4 wl[CG ? WAVES * (GPRE - GREG) * 64 : 1];       // wave-private weight blocks GREG .. GPRE-1
  // NB (template) = batch blocks of 16 rows handled by THIS workgroup; a.NB = batch blocks of the fragment layout.
  // With nsplit = a.NB / NB > 1 a tile is shared by nsplit workgroups (one per batch part): more workgroups for
  // the stages with few tiles; the parts of a tile are 8 ids apart so they land on the same XCD / L2.
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int LNB = a.NB, nsplit = LNB / NB;
  int gi = 0, wg = blockIdx.x;
  if (wg >= a.g[0].tiles * nsplit) { gi = 1; wg -= a.g[0].tiles * nsplit; }
  const Grp& G = a.g[gi];
  int tile = wg, part = 0;
  if (nsplit == 2) {
    const int full = (G.tiles / 8) * 16;            // ids in complete 16-blocks: [8 tiles part 0 | same 8 tiles part 1]
    if (wg < full) { tile = (wg / 16) * 8 + (wg & 7); part = (wg >> 3) & 1; }
    else { const int r = wg - full; tile = (G.tiles / 8) * 8 + (r >> 1); part = r & 1; }
  }
  const int nb0 = part * NB;
  const ZeggsDecDims& d = a.d;
  const int B = d.B, H = d.H, BP = 16 * NB, t = a.t, PO = d.PO;

  // ---- epilogue operands are fetched FIRST so that their latency hides under the weight stream.
  // Every epilogue item (virtual column ev, batch row eb) belongs to exactly one thread (16*BP <= NTHR).
  const int ev = tid / BP, ebl = tid % BP, eb = 16 * nb0 + ebl;   // ebl: batch row within this workgroup's part
  bool eact = false;
  float pre[7];
  float rt[10];
  const bool root = (tile == 0 && gi == 0 && tid < BP && eb < B);   // then ev == 0 and eb is this thread's batch row
  auto fetch_operands = [&]() {
  switch (G.epi) {
    case EPI_ELU_HID: if constexpr (FAM == 0) {
      const int col = tile * 16 + ev;
      eact = tid < 16 * BP && eb < B && col < H;
      if (eact) pre[0] = G.p0[col];
    } break;
    case EPI_ELU_FILM: if constexpr (FAM == 0) {   // p1 / p2: gamma / beta rows of this step, [B][2H] (pre-offset to the layer's half)
      const int col = tile * 16 + ev;
      eact = tid < 16 * BP && eb < B && col < H;
      if (eact) { pre[0] = G.p0[col]; pre[1] = G.p1[(long)eb * 2 * H + col]; pre[2] = G.p2[(long)eb * 2 * H + col]; }
    } break;
    case EPI_GRU_FWD: if constexpr (FAM == 0) {
      const int U = tile * 5 + ev;
      eact = tid < 5 * BP && eb < B && U < H;
      if (eact) {
        pre[0] = G.p0[U]; pre[1] = G.p1[U]; pre[2] = G.p0[H + U]; pre[3] = G.p1[H + U];
        pre[4] = G.p0[2 * H + U]; pre[5] = G.p1[2 * H + U]; pre[6] = G.p2[(long)eb * H + U];
        if (G.p3) {   // input-side pre-activations of this step from memory ([B][3H], bias included: time-batched GEMM)
          const float* gi = G.p3 + (long)eb * 3 * H;
          pre[0] = gi[U]; pre[2] = gi[H + U]; pre[4] = gi[2 * H + U];
        }
      }
    } break;
    case EPI_OUT_FWD: if constexpr (FAM == 0) {
      const int col = tile * 16 + ev;
      eact = tid < 16 * BP && eb < B && col < PO;
      if (eact) {
        pre[0] = G.p0[col]; pre[1] = a.st.out_std[col]; pre[2] = a.st.out_mean[col];
        pre[3] = a.st.in_mean[col]; pre[4] = a.st.in_std[col];
      }
      if (root) {
        const float* rq = a.rrot + ((long)eb * d.T + t - 1) * 4;
        const float* rp = a.rpos + ((long)eb * d.T + t - 1) * 3;
        rt[0] = rq[0]; rt[1] = rq[1]; rt[2] = rq[2]; rt[3] = rq[3]; rt[4] = rp[0]; rt[5] = rp[1]; rt[6] = rp[2];
        if (t + 1 < d.T) {
          const float* gz = a.gaze + ((long)eb * d.T + t + 1) * 3;
          rt[7] = gz[0]; rt[8] = gz[1]; rt[9] = gz[2];
        }
      }
    } break;
    case EPI_HID_MERGED: if constexpr (FAM == 0) {
      const int col = tile * 16 + ev;
      eact = tid < 16 * BP && eb < B && col < H;
      if (eact) {
        const float* wgz = G.p1 + (long)col * a.XD + PO;     // gaze columns of layer0
        pre[0] = G.p0[col]; pre[1] = wgz[0]; pre[2] = wgz[1]; pre[3] = wgz[2];
        if (G.p3) { pre[4] = G.p3[(long)eb * 2 * H + col]; pre[5] = G.p4[(long)eb * 2 * H + col]; }   // film: gamma / beta of step t+1
      }
      if (tid < BP && eb < B) {
        const float* rq = a.rrot + ((long)eb * d.T + t - 1) * 4;
        const float* rp = a.rpos + ((long)eb * d.T + t - 1) * 3;
        const float* gz = a.gaze + ((long)eb * d.T + t + 1) * 3;
        rt[0] = rq[0]; rt[1] = rq[1]; rt[2] = rq[2]; rt[3] = rq[3]; rt[4] = rp[0]; rt[5] = rp[1]; rt[6] = rp[2];
        rt[7] = gz[0]; rt[8] = gz[1]; rt[9] = gz[2];
      }
    } break;
    case EPI_GRU_BWD: if constexpr (FAM == 1) {
      const int U = tile * 16 + ev;
      eact = tid < 16 * BP && eb < B && U < H;
      if (eact) {
        const long i = (long)eb * H + U;
        const f4 gt = ((const f4*)G.p0)[i];            // saved gates (r, z, n, nh)
        pre[0] = G.o0[i]; pre[1] = gt.x; pre[2] = gt.y; pre[3] = gt.z; pre[4] = gt.w; pre[5] = G.p4[i];
      }
    } break;
    case EPI_GRU_BWD_M: if constexpr (FAM == 1) {
      const int U = tile * 16 + ev;
      eact = tid < 16 * BP && eb < B && U < H;
      if (eact) {
        const long i = (long)eb * H + U;
        const f4 gt = ((const f4*)G.p0)[i];            // saved gates (r, z, n, nh)
        pre[0] = G.o0[i]; pre[1] = gt.x; pre[2] = gt.y; pre[3] = gt.z; pre[4] = gt.w; pre[5] = G.p4[i];
#pragma unroll
        for (int c = 0; c < 6; ++c) rt[c] = a.aux1[(long)c * H + U];     // layer2 rows of the 6 root columns
      }
    } break;
    case EPI_ADD: if constexpr (FAM == 1) {
      const int col = tile * 16 + ev;
      eact = tid < 16 * BP && eb < B && col < H;
      if (eact) pre[0] = G.o0[(long)eb * H + col];
    } break;
    case EPI_DGIN: if constexpr (FAM == 1) {
      const int j = tile * 16 + ev;
      eact = tid < 16 * BP && eb < B && j < H + a.XD;
      if (eact && j < H) {
        if (G.p1) { pre[0] = G.p1[(long)eb * H + j]; pre[1] = G.p2[(long)eb * 2 * H + j]; }   // film: ELU output A0, gamma
        else pre[0] = G.p0[(long)eb * a.GL + j];
      }
    } break;
    case EPI_FILM_BWD: if constexpr (FAM == 1) {   // p0: ELU output A2 [B][H]; p1: gamma rows [B][2H] (pre-offset)
      const int U = tile * 16 + ev;
      eact = tid < 16 * BP && eb < B && U < H;
      if (eact) { pre[0] = G.p0[(long)eb * H + U]; pre[1] = G.p1[(long)eb * 2 * H + U]; }
    } break;
    case EPI_FILM_BWD_M: if constexpr (FAM == 1) {
      const int U = tile * 16 + ev;
      eact = tid < 16 * BP && eb < B && U < H;
      if (eact) {
        pre[0] = G.p0[(long)eb * H + U]; pre[1] = G.p1[(long)eb * 2 * H + U];
#pragma unroll
        for (int c = 0; c < 6; ++c) rt[c] = a.aux1[(long)c * H + U];     // layer3 rows of the 6 root columns
      }
    } break;
    case EPI_DX: if constexpr (FAM == 1) {
      const int q = tile * 16 + ev;
      eact = tid < 16 * BP && eb < B && q < a.XD;
      if (eact) {
        const int j = perm_dx(q, PO);
        pre[0] = G.p0[(long)eb * a.XD + j];
        if (G.o1 && j < PO) {
          pre[1] = a.dpose[((long)eb * d.T + t - 1) * PO + j]; pre[2] = a.st.in_std[j]; pre[3] = a.st.out_std[j];
        }
      }
    } break;
  }
  };
  if constexpr (!CG) fetch_operands();   // chained GEMV: after the hand-off (the registers hold weights until then)

  if constexpr (BV > 0) {
    // ---- GEMV mode: batch <= 2 (autoregressive decode).  No MFMA padding to 16 batch rows: each lane owns
    // (column i, k-quarter) of the weight fragment and multiplies it with the matching float4 of every batch row
    // read straight from the canonical activations (16 lanes share an address: one L1 broadcast).
    float av[2][BV];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int b = 0; b < BV; ++b) av[i][b] = 0.f;
    int TB = 0;
    for (int s = 0; s < G.nseg; ++s) TB += G.seg[s].kb;
    const int b0 = wave * TB / WAVES, b1 = (wave + 1) * TB / WAVES;
    if constexpr (CH) {
      // ---- chained launch: ALL of this wave's weight blocks (<= GPRE, checked by the host) are fetched into registers
      // while the predecessor is still running; after the hand-off only the activation vector (a few KB, staged through a
      // wave-private LDS slice) and the dot products remain on the critical path.
      const int k0 = G.seg[0].kb, k1 = k0 + (G.nseg > 1 ? G.seg[1].kb : 0);
      auto locate = [&](int gb, int& sidx, int& l) {
        sidx = gb < k0 ? 0 : (gb < k1 ? 1 : 2);
        l = gb - (sidx == 0 ? 0 : (sidx == 1 ? k0 : k1));
      };
      f4 wreg[GREG];
      f4* wlw = wl + wave * ((GPRE - GREG) * 64) + lane;
      const f4* wcat = (const f4*)G.wcat + ((long)tile * G.tkbcat + b0) * 64 + lane;
      const int nblk = b1 - b0;
      auto wload = [&](int i) -> f4 {
        f4 v = wcat[(long)(i < nblk ? i : nblk - 1) * 64];     // clamped: branch-free, the surplus is zeroed
        if (i >= nblk) v = f4{0.f, 0.f, 0.f, 0.f};
        return v;
      };
      CHT(0);
      // the LDS-parked blocks first (their registers are free again before the register-resident ones are fetched)
#pragma unroll
      for (int i = GREG; i < GPRE; ++i) wlw[(i - GREG) * 64] = wload(i);
      CHT(1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < GREG; ++i) wreg[i] = wload(i);
      __builtin_amdgcn_sched_barrier(0);
#ifdef ZEGGS_CHTIME
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      CHT(2);
#endif
      if (a.ch_wait) {
        if (wave == 0) chain_wait(a.ch_wait, a.wait_count, a.ch_err);
        __syncthreads();
      }
      CHT(3);
      fetch_operands();
      f4* xw = xs + wave * (BV * GPRE * 4);
      for (int q = lane; q < BV * GPRE * 4; q += 64) {
        const int b = q / (GPRE * 4), r = q % (GPRE * 4), i = r >> 2, e = r & 3;
        const int gb = (b0 + i < b1) ? b0 + i : b1 - 1;
        int sidx, l;
        locate(gb, sidx, l);
        const float* xc = sidx == 0 ? G.seg[0].xc : (sidx == 1 ? G.seg[1].xc : G.seg[2].xc);
        const int ldx = sidx == 0 ? G.seg[0].ldx : (sidx == 1 ? G.seg[1].ldx : G.seg[2].ldx);
        xw[q] = *(const f4*)(xc + (long)b * ldx + 16 * l + 4 * e);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the slice is written and read by this wave only
      CHT(4);
#pragma unroll
      for (int i = 0; i < GPRE; ++i) {
        if ((i & 3) == 0) __builtin_amdgcn_sched_barrier(0);      // keep the LDS reads from piling up in registers
        const bool acc1 = b0 + i >= G.kacc;
#pragma unroll
        for (int b = 0; b < BV; ++b) {
          const f4 xv = xw[(b * GPRE + i) * 4 + (lane >> 4)];
          const f4 wv = i < GREG ? wreg[i < GREG ? i : 0] : wlw[(i - GREG) * 64];
          const float dd = fmaf(wv.x, xv.x, fmaf(wv.y, xv.y, fmaf(wv.z, xv.z, wv.w * xv.w)));
          av[0][b] += acc1 ? 0.f : dd;
          av[1][b] += acc1 ? dd : 0.f;
        }
      }
    } else {
    int base = 0;
    for (int s = 0; s < G.nseg; ++s) {
      const int kbs = G.seg[s].kb;
      const int lo = (b0 > base ? b0 : base) - base;
      const int hi = (b1 < base + kbs ? b1 : base + kbs) - base;
      if (lo < hi) {
        const f4* wp = (const f4*)G.seg[s].w + ((long)(G.seg[s].fixed ? 0 : tile) * (G.seg[s].tkb ? G.seg[s].tkb : kbs)) * 64 + lane;
        const float* xc = G.seg[s].xc + 4 * (lane >> 4);
        const int ldx = G.seg[s].ldx;
        float part[BV];
#pragma unroll
        for (int b = 0; b < BV; ++b) part[b] = 0.f;
        // 8 weight blocks (8 KiB per wave) in flight per round trip
        int kb = lo;
        for (; kb + 8 <= hi; kb += 8) {
          f4 wv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) wv[u] = wp[(long)(kb + u) * 64];
#pragma unroll
          for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int b = 0; b < BV; ++b) {
              const f4 xv = *(const f4*)(xc + (long)b * ldx + 16 * (kb + u));
              part[b] = fmaf(wv[u].x, xv.x, fmaf(wv[u].y, xv.y, fmaf(wv[u].z, xv.z, fmaf(wv[u].w, xv.w, part[b]))));
            }
        }
        for (; kb < hi; ++kb) {
          const f4 wv = wp[(long)kb * 64];
#pragma unroll
          for (int b = 0; b < BV; ++b) {
            const f4 xv = *(const f4*)(xc + (long)b * ldx + 16 * kb);
            part[b] = fmaf(wv.x, xv.x, fmaf(wv.y, xv.y, fmaf(wv.z, xv.z, fmaf(wv.w, xv.w, part[b]))));
          }
        }
        if (G.seg[s].acc == 0) {
#pragma unroll
          for (int b = 0; b < BV; ++b) av[0][b] += part[b];
        } else {
#pragma unroll
          for (int b = 0; b < BV; ++b) av[1][b] += part[b];
        }
      }
      base += kbs;
    }
    }
    CHT(5);
    float* redf = (float*)red;   // [wave][2][BV][16]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int b = 0; b < BV; ++b) {
        float v = av[i][b];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < 16) redf[((wave * 2 + i) * BV + b) * 16 + lane] = v;
      }
    __syncthreads();
    if (tid < 2 * BV * 16) {
      const int i = tid / (BV * 16), r = tid % (BV * 16), b = r / 16, col = r % 16;
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) sum += redf[((w * 2 + i) * BV + b) * 16 + col];
      ((float*)fin)[(((i * NB) * 64 + (((col >> 2) << 4) | b)) << 2) | (col & 3)] = sum;
    }
    __syncthreads();
  } else {
  // ---- weight stream: the waves split the concatenated k-block list of the segments
    f4 acc[2][NB];
  #pragma unroll
    for (int i = 0; i < 2; ++i)
  #pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[i][nb] = f4{0.f, 0.f, 0.f, 0.f};
    int TB = 0;
    for (int s = 0; s < G.nseg; ++s) TB += G.seg[s].kb;
    const int b0 = wave * TB / WAVES, b1 = (wave + 1) * TB / WAVES;
    int base = 0;
    for (int s = 0; s < ((a.variant & V_NOW) ? 0 : G.nseg); ++s) {
      const int kbs = G.seg[s].kb;
      const int lo = (b0 > base ? b0 : base) - base;
      const int hi = (b1 < base + kbs ? b1 : base + kbs) - base;
      if (lo < hi) {
        const f4* wp = (const f4*)G.seg[s].w + ((long)(G.seg[s].fixed ? 0 : tile) * (G.seg[s].tkb ? G.seg[s].tkb : kbs)) * 64 + lane;
        const f4* xp = (const f4*)G.seg[s].x + nb0 * 64 + lane;
        // the forward stage of 49-64 rows: one k-block per buffer (two would be 80 operand registers beside 32 accumulators
        // and the epilogue operands: 67 spilled at the 128 registers a 16-wave workgroup leaves a wave)
        constexpr int UG = (NB == 4 && FAM == 0) ? 1 : 0;
        if (G.seg[s].acc == 0) run_blocks<NB, UG>(wp, xp, lo, hi, acc[0], LNB);
        else run_blocks<NB, UG>(wp, xp, lo, hi, acc[1], LNB);
      }
      base += kbs;
    }
  #pragma unroll
    for (int i = 0; i < 2; ++i)
  #pragma unroll
      for (int nb = 0; nb < NB; ++nb) red[wave][i][nb][lane] = acc[i][nb];
    __syncthreads();
    if (tid < 2 * NB * 64) {
      const int i = tid / (NB * 64), r = tid % (NB * 64), nb = r / 64, l = r % 64;
      f4 s = red[0][i][nb][l];
  #pragma unroll
      for (int w = 1; w < WAVES; ++w) s += red[w][i][nb][l];
      fin[i][nb][l] = s;
    }
    __syncthreads();
}
  CHT(6);
  const float* finf = (const float*)fin;
  auto FV = [&](int i, int vcol, int bg) -> float {   // bg = global batch row (must belong to this part)
    const int b = bg - 16 * nb0;
    return finf[(((i * NB + (b >> 4)) * 64 + (((vcol >> 2) << 4) | (b & 15))) << 2) | (vcol & 3)];
  };
  if (a.variant & V_NOEPI) {
    if (tid == 0 && finf[0] == 123.456f) a.carry[0] = 1.f;
    return;
  }

  switch (G.epi) {
    case EPI_ELU_HID: if constexpr (FAM == 0) {   // hid = ELU(W0 x + b0) -> Gin[t][:, 0:H] and its fragment copy
      if (eact) {
        const int col = tile * 16 + ev;
        const float val = d_elu(FV(0, ev, eb) + pre[0]);
        st_out<CH>(&G.o0[(long)eb * a.GL + col], val);
        if (G.o1) st_out<CH>(&G.o1[xf_index(eb, col, LNB)], val);
      }
    } break;
    case EPI_ELU_FILM: if constexpr (FAM == 0) {  // a = ELU(W x + b); out = a (1 + gamma) + beta (reference modules.py:213-225)
      if (eact) {
        const int col = tile * 16 + ev;
        const float av = d_elu(FV(0, ev, eb) + pre[0]);
        const float val = av * (1.f + pre[1]) + pre[2];
        st_out<CH>(&G.o0[(long)eb * (G.ld0 ? G.ld0 : a.GL) + col], val);
        if (G.o1) st_out<CH>(&G.o1[xf_index(eb, col, LNB)], val);
        if (G.o2) G.o2[(long)eb * H + col] = av;      // pre-modulation activation, read by the backward pass only
      }
    } break;
    case EPI_GRU_FWD: if constexpr (FAM == 0) {   // tile = 5 hidden units x (r, z, n); acc0 = input side, acc1 = hidden side
      if (eact) {
        const int u = ev, b = eb, U = tile * 5 + u;
        const float r = d_sigmoid(FV(0, u, b) + pre[0] + (FV(1, u, b) + pre[1]));   // (p3: pre[0,2,4] = stored input side)
        const float z = d_sigmoid(FV(0, 5 + u, b) + pre[2] + (FV(1, 5 + u, b) + pre[3]));
        const float nh = FV(1, 10 + u, b) + pre[5];
        const float nn = d_tanh(FV(0, 10 + u, b) + pre[4] + r * nh);
        const long i = (long)b * H + U;
        const float h = (1.f - z) * nn + z * pre[6];
        st_out<CH>(&G.o0[i], h);
        if (G.o1) st_out<CH>(&G.o1[xf_index(b, U, LNB)], h);
        if (G.o2) ((f4*)G.o2)[i] = f4{r, z, nn, nh};   // saved gates, one 16-byte store (read by the backward pass only)
      }
      if (a.cf_gin || a.cf_x || a.cf_cond) {   // speech / style columns of x_{t+1} (inputs: independent of this step)
        const int XC = d.SP + (d.film ? 0 : d.ST);
        for (int e = blockIdx.x * NTHR + tid; e < B * XC; e += gridDim.x * NTHR) {
          const int b = e / XC, c = e % XC;
          const float val = c < d.SP ? a.speech[((long)b * d.T + t + 1) * d.SP + c]
                                     : a.style[((long)b * d.T + t + 1) * d.ST + (c - d.SP)];
          if (a.cf_gin) st_out<CH>(&a.cf_gin[(long)b * a.GL + H + d.PI + c], val);
          if (a.cf_x) st_out<CH>(&a.cf_x[xf_index(b, d.PI + c, LNB)], val);
          if (a.cf_cond) st_out<CH>(&a.cf_cond[xf_index(b, c, LNB)], val);
        }
      }
    } break;
    case EPI_OUT_FWD: if constexpr (FAM == 0) {   // y = W2 h1 + b2 -> pose[t], root integration, x_{t+1}
      float* gnext = G.o0;
      float* xnext = G.o1;
      const bool next = (t + 1 < d.T);
      if (eact) {
        const int col = tile * 16 + ev, b = eb;
        const float p = (FV(0, ev, b) + pre[0]) * pre[1] + pre[2];
        a.pose[((long)b * d.T + t) * PO + col] = p;
        if (next) {
          const float e = (p - pre[3]) / pre[4];
          if (gnext) st_out<CH>(&gnext[(long)b * a.GL + H + col], e);
          if (xnext) st_out<CH>(&xnext[xf_index(b, col, LNB)], e);
        }
      }
      if (root) {
        const int b = eb;
        float p[6], genc[3];
        for (int c = 0; c < 6; ++c) p[c] = (FV(0, c, b) + G.p0[c]) * a.st.out_std[c] + a.st.out_mean[c];
        V3 npos; Q4 nq;
        root_step(d, a.st, rt, p, next, npos, nq, genc);
        // (the root state is an epilogue operand of the launch three places down the chain, fetched before its hand-off wait)
        float* op = a.rpos + ((long)b * d.T + t) * 3;
        st_out<CH>(op, npos.x); st_out<CH>(op + 1, npos.y); st_out<CH>(op + 2, npos.z);
        float* oq = a.rrot + ((long)b * d.T + t) * 4;
        st_out<CH>(oq, nq.w); st_out<CH>(oq + 1, nq.x); st_out<CH>(oq + 2, nq.y); st_out<CH>(oq + 3, nq.z);
        if (next) {
          for (int k = 0; k < 3; ++k) {
            if (gnext) st_out<CH>(&gnext[(long)b * a.GL + H + PO + k], genc[k]);
            if (xnext) st_out<CH>(&xnext[xf_index(b, PO + k, LNB)], genc[k]);
          }
        }
      }
    } break;
    case EPI_HID_MERGED: if constexpr (FAM == 0) {   // hid_{t+1} = ELU(M h1 + Wc cond + cvec + W0[:, gaze] genc)
      float* gsh = (float*)red;                       // [BP][3] normalised gaze direction of x_{t+1}
      if (tid < BP && eb < B) {
        float p[6], genc[3];
        for (int c = 0; c < 6; ++c) p[c] = (FV(1, c, eb) + G.p2[c]) * a.st.out_std[c] + a.st.out_mean[c];
        V3 npos; Q4 nq;
        root_step(d, a.st, rt, p, true, npos, nq, genc);
        gsh[ebl * 3] = genc[0]; gsh[ebl * 3 + 1] = genc[1]; gsh[ebl * 3 + 2] = genc[2];
      }
      __syncthreads();
      if (eact) {
        const int col = tile * 16 + ev;
        const float pa = FV(0, ev, eb) + pre[0] + pre[1] * gsh[ebl * 3] + pre[2] * gsh[ebl * 3 + 1] + pre[3] * gsh[ebl * 3 + 2];
        float val = d_elu(pa);
        if (G.p3) {     // film: the modulation of layer0's output (and its pre-modulation value for the backward pass)
          if (G.o2) G.o2[(long)eb * H + col] = val;
          val = val * (1.f + pre[4]) + pre[5];
        }
        st_out<CH>(&G.o0[(long)eb * a.GL + col], val);
        if (G.o1) st_out<CH>(&G.o1[xf_index(eb, col, LNB)], val);
      }
    } break;
    case EPI_GRU_BWD: if constexpr (FAM == 1) {   // dh = W^T delta + carry -> gate gradients of this layer
      if (eact) {
        const int b = eb, U = tile * 16 + ev;
        const long i = (long)b * H + U;
        const float g = FV(0, ev, b) + pre[0];
        const float r = pre[1], z = pre[2], nn = pre[3], nh = pre[4], hp = pre[5];
        const float dn = g * (1.f - z);
        const float dz = g * (hp - nn);
        const float dan = dn * (1.f - nn * nn);
        const float dar = dan * nh * r * (1.f - r);
        const float daz = dz * z * (1.f - z);
        float* di = G.o1 + (long)b * 3 * H;
        di[U] = dar; di[H + U] = daz; di[2 * H + U] = dan;
        G.o2[i] = dan * r;                 // hidden-side gradient: only its n rows differ from di (r, z rows are shared)
        G.o3[xf_index(b, U, LNB)] = dar; G.o3[xf_index(b, H + U, LNB)] = daz; G.o3[xf_index(b, 2 * H + U, LNB)] = dan;
        G.o4[xf_index(b, U, LNB)] = dan * r;
        G.o0[i] = g * z;
      }
    } break;
    case EPI_GRU_BWD_M: if constexpr (FAM == 1) {   // layer-1 gate gradients of step t-1 in the launch that produces dy_{t-1}:
      // dH1 = M'^T D0 + W2^T r (acc 0) + W2[0:6]^T dy6 + carry, with dy6 from the root-integration backward
      // evaluated here from the root columns of W0^T D0 (acc 1); the DX group of this launch owns the carry update.
      float* dsh = (float*)red;                      // [BP][6]
      if (tid < BP && eb < B) {
        const int b = eb;
        float g6[6], dgd[3];
        for (int c = 0; c < 6; ++c)
          g6[c] = a.dpose[((long)b * d.T + t - 1) * PO + c] + (FV(1, c, b) + a.aux0[(long)b * a.XD + c]) / a.st.in_std[c];
        for (int k = 0; k < 3; ++k) dgd[k] = FV(1, 6 + k, b) + a.aux0[(long)b * a.XD + PO + k];
        root_bwd(d, a.st, b, t - 1, true, dgd, a.gaze, a.cpose, a.crpos, a.crrot, a.drpos, a.drrot, a.carry, nullptr, g6);
        for (int c = 0; c < 6; ++c) dsh[ebl * 6 + c] = g6[c] * a.st.out_std[c];
      }
      __syncthreads();
      if (eact) {
        const int b = eb, U = tile * 16 + ev;
        const long i = (long)b * H + U;
        float g = FV(0, ev, b) + pre[0];
#pragma unroll
        for (int c = 0; c < 6; ++c) g = fmaf(rt[c], dsh[ebl * 6 + c], g);
        const float r = pre[1], z = pre[2], nn = pre[3], nh = pre[4], hp = pre[5];
        const float dn = g * (1.f - z);
        const float dz = g * (hp - nn);
        const float dan = dn * (1.f - nn * nn);
        const float dar = dan * nh * r * (1.f - r);
        const float daz = dz * z * (1.f - z);
        float* di = G.o1 + (long)b * 3 * H;
        di[U] = dar; di[H + U] = daz; di[2 * H + U] = dan;
        G.o2[i] = dan * r;                 // hidden-side gradient: only its n rows differ from di (r, z rows are shared)
        G.o3[xf_index(b, U, LNB)] = dar; G.o3[xf_index(b, H + U, LNB)] = daz; G.o3[xf_index(b, 2 * H + U, LNB)] = dan;
        G.o4[xf_index(b, U, LNB)] = dan * r;
        G.o0[i] = g * z;
      }
    } break;
    case EPI_ADD: if constexpr (FAM == 1) {
      if (eact) G.o0[(long)eb * H + tile * 16 + ev] = pre[0] + FV(0, ev, eb);
    } break;
    case EPI_DGIN: if constexpr (FAM == 1) {      // dGin = W_ih0^T delta_i0 : [dhid -> ELU' -> D0 | dx part]
      if (eact) {
        const int b = eb, j = tile * 16 + ev;
        const float g = FV(0, ev, b);
        if (j < H) {
          float d0 = g * d_elu_grad_from_out(pre[0]);
          if (G.p1) {     // film: g is the gradient of the modulated value
            d0 *= 1.f + pre[1];
            G.o3[(long)b * 2 * H + j] = g * pre[0];
            G.o4[(long)b * 2 * H + j] = g;
          }
          G.o0[(long)b * H + j] = d0;
          G.o1[xf_index(b, j, LNB)] = d0;
        } else {
          const int c = j - H;
          G.o2[(long)b * a.XD + c] = g;
          if (a.rxf && c >= 6 && c < PO)     // operand of the merged W2^T product of the next launch
            a.rxf[xf_index(b, c, LNB)] = a.st.out_std[c] * (a.dpose[((long)b * d.T + t - 1) * PO + c] + g / a.st.in_std[c]);
        }
      }
    } break;
    case EPI_FILM_BWD: if constexpr (FAM == 1) {  // dF2 = W3^T dy -> D2 = dF2 (1 + gamma) ELU'(A2), dgamma = dF2 A2, dbeta = dF2
      if (eact) {
        const int b = eb, U = tile * 16 + ev;
        const float g = FV(0, ev, b);
        const float d2 = g * (1.f + pre[1]) * d_elu_grad_from_out(pre[0]);
        G.o0[(long)b * H + U] = d2;
        G.o1[xf_index(b, U, LNB)] = d2;
        G.o2[(long)b * 2 * H + U] = g * pre[0];
        G.o3[(long)b * 2 * H + U] = g;
      }
    } break;
    case EPI_FILM_BWD_M: if constexpr (FAM == 1) {   // film twin of EPI_GRU_BWD_M: dF2_{t-1} in the launch that produces dy_{t-1}:
      // dF2 = M'^T D0 + W3^T r (acc 0) + W3[0:6]^T dy6, dy6 from the root-integration backward of the root columns of W0^T D0
      // (acc 1); then the modulation / ELU backward of layer2 (EPI_FILM_BWD).  The DX group of this launch owns the carry update.
      float* dsh = (float*)red;                      // [BP][6]
      if (tid < BP && eb < B) {
        const int b = eb;
        float g6[6], dgd[3];
        for (int c = 0; c < 6; ++c)
          g6[c] = a.dpose[((long)b * d.T + t - 1) * PO + c] + (FV(1, c, b) + a.aux0[(long)b * a.XD + c]) / a.st.in_std[c];
        for (int k = 0; k < 3; ++k) dgd[k] = FV(1, 6 + k, b) + a.aux0[(long)b * a.XD + PO + k];
        root_bwd(d, a.st, b, t - 1, true, dgd, a.gaze, a.cpose, a.crpos, a.crrot, a.drpos, a.drrot, a.carry, nullptr, g6);
        for (int c = 0; c < 6; ++c) dsh[ebl * 6 + c] = g6[c] * a.st.out_std[c];
      }
      __syncthreads();
      if (eact) {
        const int b = eb, U = tile * 16 + ev;
        float g = FV(0, ev, b);
#pragma unroll
        for (int c = 0; c < 6; ++c) g = fmaf(rt[c], dsh[ebl * 6 + c], g);
        const float d2 = g * (1.f + pre[1]) * d_elu_grad_from_out(pre[0]);
        G.o0[(long)b * H + U] = d2;
        G.o1[xf_index(b, U, LNB)] = d2;
        G.o2[(long)b * 2 * H + U] = g * pre[0];
        G.o3[(long)b * 2 * H + U] = g;
      }
    } break;
    case EPI_DX: if constexpr (FAM == 1) {        // dx_t = dXa + W0^T D0 ; pose part -> dy_{t-1} (devectorize/vectorize backward)
      float* dy = G.o1;    // DY[t-1] canonical (null when t == 1)
      if (eact) {
        const int b = eb, j = perm_dx(tile * 16 + ev, PO);
        const float dx = FV(0, ev, b) + pre[0];
        if (j >= d.PI) G.o0[(long)b * a.XD + j] = dx;   // only the speech / style columns of dx_t are consumed later
        if (dy && j >= 6 && j < PO) {
          const float gy = (pre[1] + dx / pre[2]) * pre[3];
          dy[(long)b * a.POL + j] = gy;
          G.o2[xf_index(b, j, LNB)] = gy;
        }
      }
      if (root && dy) {
        const int b = eb;
        float g6[6], dgd[3];
        for (int c = 0; c < 6; ++c)
          g6[c] = a.dpose[((long)b * d.T + t - 1) * PO + c] +
                  (FV(0, c, b) + G.p0[(long)b * a.XD + c]) / a.st.in_std[c];
        for (int k = 0; k < 3; ++k) dgd[k] = FV(0, 6 + k, b) + G.p0[(long)b * a.XD + PO + k];
        root_bwd(d, a.st, b, t - 1, true, dgd, a.gaze, a.cpose, a.crpos, a.crrot, a.drpos, a.drrot, a.carry, a.carry_out, g6);
        for (int c = 0; c < 6; ++c) {
          const float gy = g6[c] * a.st.out_std[c];
          dy[(long)b * a.POL + c] = gy;
          G.o2[xf_index(b, c, LNB)] = gy;
        }
      }
    } break;
  }
  CHT(7);
  if constexpr (CH) {
#ifdef ZEGGS_CHTIME
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CHT(8);
#endif
    chain_arrive(a.ch_arrive);
    CHT(9);
  }
}

// gradient wrt the raw output of the LAST step (no next step feeds on it)
__global__ void dy_last_k(ZeggsDecDims d, ZeggsDecStats st, const float* dpose, const float* drpos, const float* drrot,
                          const float* gaze, const float* pose, const float* rpos, const float* rrot, float* carry,
                          float* dy, int POL, float* dyxf, int NB) {
  const int b = blockIdx.x, t = d.T - 1;
  const float* dpb = dpose + ((long)b * d.T + t) * d.PO;
  for (int c = 6 + threadIdx.x; c < d.PO; c += blockDim.x) {
    const float gy = dpb[c] * st.out_std[c];
    dy[(long)b * POL + c] = gy;
    dyxf[xf_index(b, c, NB)] = gy;
  }
  if (threadIdx.x == 0) {
    float g6[6];
    for (int c = 0; c < 6; ++c) g6[c] = dpb[c];
    root_bwd(d, st, b, t, false, nullptr, gaze, pose, rpos, rrot, drpos, drrot, carry, carry, g6);
    for (int c = 0; c < 6; ++c) {
      const float gy = g6[c] * st.out_std[c];
      dy[(long)b * POL + c] = gy;
      dyxf[xf_index(b, c, NB)] = gy;
    }
  }
}

// canonical [B, ld] (columns off .. off+K) -> activation fragments
__global__ void to_xfrag_k(float* xf, const float* src, long ld, int off, int K, int B, int NB) {
  long n = (long)B * K;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int k = (int)(i % K);
    int b = (int)(i / K);
    xf[xf_index(b, k, NB)] = src[(long)b * ld + off + k];
  }
}

struct PackArgs {
  float* dst;
  const float* src;
  int tiles, kb, mode, K, N, H, PO;
  long ld;
  int off;
  int tkb;   // k-blocks per tile of the destination pack (>= kb: the pack holds further segments per tile)
};
// mode 0: rows (virtual col = source row)          V[vc][k] = src[vc*ld + off + k]
// mode 1: GRU rows, tile = 5 units x (r,z,n)       V[g*5+u][k] = src[(g*H + tile*5+u)*ld + off + k]
// mode 2: columns (transposed)                     V[vc][k] = src[k*ld + off + vc]
// mode 3: columns with the dX permutation          V[q][k]  = src[k*ld + perm(q)]
// mode 4: the rows of tile 0 for EVERY tile         V[vc][k] = src[(vc % 16)*ld + off + k]
__global__ void pack_k(PackArgs p) {
  const long n = (long)p.tiles * p.kb * 64;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    const long tk = idx >> 6;
    const int kbi = (int)(tk % p.kb), tile = (int)(tk / p.kb);
    const int i = lane & 15, k0 = 16 * kbi + 4 * (lane >> 4);
    f4 v = f4{0.f, 0.f, 0.f, 0.f};
    if (p.mode <= 1 || p.mode == 4) {
      long row = -1;
      if (p.mode == 0) { if (tile * 16 + i < p.N) row = tile * 16 + i; }
      else if (p.mode == 4) { if (i < p.N) row = i; }
      else { const int g = i / 5, u = i % 5, U = tile * 5 + u; if (g < 3 && U < p.H) row = (long)g * p.H + U; }
      if (row >= 0) {
        const float* s = p.src + row * p.ld + p.off;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (k0 + c < p.K) v[c] = s[k0 + c];
      }
    } else {
      const int q = tile * 16 + i;
      if (q < p.N) {
        const int col = p.mode == 3 ? perm_dx(q, p.PO) : p.off + q;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (k0 + c < p.K) v[c] = p.src[(long)(k0 + c) * p.ld + col];
      }
    }
    ((f4*)p.dst)[((long)tile * p.tkb + kbi) * 64 + lane] = v;
  }
}

int pack(float* dst, const float* src, int tiles, int kb, int mode, int K, int N, int H, int PO, long ld, int off,
         hipStream_t s, int tkb = 0) {
  PackArgs p{dst, src, tiles, kb, mode, K, N, H, PO, ld, off, tkb ? tkb : kb};
  long n = (long)tiles * kb * 64;
  long g = (n + 255) / 256;
  hipLaunchKernelGGL(pack_k, dim3((unsigned)(g > 16384 ? 16384 : g)), dim3(256), 0, s, p);
  ZLAUNCH_CHECK("pack");
  return 0;
}

// grid size of a stage launch (same rule as launch_stage_f / launch_stage_gemv)
int stage_nsplit(const StageArgs& a) {
  if (a.gemv) return 1;
  const bool few = (a.g[0].tiles < 160 && a.g[1].tiles < 160) || (g_stage_variant & 64);
  return (few && a.NB % 2 == 0 && !(g_stage_variant & 32)) ? 2 : 1;
}
int stage_wgs(const StageArgs& a) { return (a.g[0].tiles + a.g[1].tiles) * stage_nsplit(a); }

int launch_stage_gemv(const StageArgs& a, hipStream_t s) {
  const int wgs = a.g[0].tiles + a.g[1].tiles;
  if (a.ch_arrive) {
#ifdef ZEGGS_CHAIN   // measurement builds only (profiles/r02_chained_launch_*): the parked weights spill 167 / 316 registers at 4 waves per SIMD
    switch (a.d.B) {
      case 1: hipLaunchKernelGGL((stage_k<1, 0, 8, 1, 1>), dim3(wgs), dim3(512), 0, s, a); break;
      case 2: hipLaunchKernelGGL((stage_k<1, 0, 8, 2, 1>), dim3(wgs), dim3(512), 0, s, a); break;
      default: zeggs_set_error("gemv mode needs batch <= 2"); return -1;
    }
    ZLAUNCH_CHECK("decoder_stage_gemv_chained");
    return 0;
#else
    zeggs_set_error("chained stage launches are not part of this build (-DZEGGS_CHAIN)");
    return -1;
#endif
  }
  switch (a.d.B) {
    case 1: hipLaunchKernelGGL((stage_k<1, 0, 8, 1>), dim3(wgs), dim3(512), 0, s, a); break;
    case 2: hipLaunchKernelGGL((stage_k<1, 0, 8, 2>), dim3(wgs), dim3(512), 0, s, a); break;
    default: zeggs_set_error("gemv mode needs batch <= 2"); return -1;
  }
  ZLAUNCH_CHECK("decoder_stage_gemv");
  return 0;
}

template <int FAM>
int launch_stage_f(const StageArgs& a, hipStream_t s) {
  // stages with few tiles are split over the batch (two workgroups per tile) to occupy more CUs
  const int nsplit = stage_nsplit(a);
  const int nbw = a.NB / nsplit;
  const int wgs = (a.g[0].tiles + a.g[1].tiles) * nsplit;
  const bool w8 = !(g_stage_variant & 128);   // 8 waves (2 workgroups per CU) for <= 32 batch rows per workgroup
  switch (nbw) {
    case 1:
      if (w8) hipLaunchKernelGGL((stage_k<1, FAM, 8>), dim3(wgs), dim3(512), 0, s, a);
      else hipLaunchKernelGGL((stage_k<1, FAM, 16>), dim3(wgs), dim3(1024), 0, s, a);
      break;
    case 2:
      if (w8) hipLaunchKernelGGL((stage_k<2, FAM, 8>), dim3(wgs), dim3(512), 0, s, a);
      else hipLaunchKernelGGL((stage_k<2, FAM, 16>), dim3(wgs), dim3(1024), 0, s, a);
      break;
    case 3: hipLaunchKernelGGL((stage_k<3, FAM, 16>), dim3(wgs), dim3(1024), 0, s, a); break;
    case 4: hipLaunchKernelGGL((stage_k<4, FAM, 16>), dim3(wgs), dim3(1024), 0, s, a); break;
    default: zeggs_set_error("decoder fast path: batch > 64"); return -1;
  }
  ZLAUNCH_CHECK("decoder_stage");
  return 0;
}
int launch_stage(const StageArgs& a, hipStream_t s) {
  if (a.gemv) return launch_stage_gemv(a, s);
  if (a.g[0].epi >= EPI_GRU_BWD) return launch_stage_f<1>(a, s);   // enum order: fwd < bwd
  // 16384: every forward stage twice (idempotent) -- the second launch finds its weights in L2: an upper bound of what
  // a weight prefetch across the stage boundary could buy (tools/warm_probe.sh)
  if (g_stage_variant & 16384) ZTRY(launch_stage_f<0>(a, s));
  return launch_stage_f<0>(a, s);
}

inline Seg seg(const float* w, const float* x, int kb, int acc, const float* xc = nullptr, int ldx = 0, int fixed = 0,
               int tkb = 0) {
  return Seg{w, x, kb, acc, xc, ldx, fixed, tkb};
}
// k-blocks [kb0, kb0 + kb) of a pack with `tkb` blocks per tile
inline Seg subseg(const float* w, const float* x, int kb0, int kb, int tkb, int acc) {
  return Seg{w + (long)kb0 * 256, x, kb, acc, nullptr, 0, 0, tkb};
}

// W0s = W0[:, :PO] diag(sigma_o / sigma_i) (zero padded to POL columns); v = (b2 sigma_o + mu_o - mu_i) / sigma_i
__global__ void merge_prep_k(float* W0s, float* vvec, const float* W0, const float* b2, ZeggsDecStats st, int H, int PO,
                             int POL, int XD, int skip /* zero the first `skip` columns (backward: root columns) */) {
  const long n = (long)H * POL;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % POL);
    const long r = i / POL;
    W0s[i] = (c < PO && c >= skip) ? W0[r * XD + c] * (st.out_std[c] / st.in_std[c]) : 0.f;
    if (r == 0 && vvec) vvec[c] = c < PO ? (b2[c] * st.out_std[c] + st.out_mean[c] - st.in_mean[c]) / st.in_std[c] : 0.f;
  }
}

// ---- chained launches (host side): consecutive stage launches alternate between the caller's stream and a library-owned
// second stream, so launch k+1 is resident and has fetched its weights while launch k still runs; the dependency is the
// device-side arrival counter (ring of CH_RING slots, never reset inside a rollout: the host passes cumulative counts).
struct ChainStream { hipStream_t s; hipEvent_t start, done; };
int chain_stream(ChainStream** out) {
  static ChainStream pool[16];
  static bool ready[16] = {};
  int dev = 0;
  ZCHECK(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16, "decoder chain: unsupported device index");
  if (!ready[dev]) {
    ZCHECK(hipStreamCreateWithFlags(&pool[dev].s, hipStreamNonBlocking) == hipSuccess, "chain stream creation failed");
    ZCHECK(hipEventCreateWithFlags(&pool[dev].start, hipEventDisableTiming) == hipSuccess, "event creation failed");
    ZCHECK(hipEventCreateWithFlags(&pool[dev].done, hipEventDisableTiming) == hipSuccess, "event creation failed");
    ready[dev] = true;
  }
  *out = &pool[dev];
  return 0;
}
struct Chain {
  bool on = false;
  hipStream_t s[2];
  ChainStream* cs = nullptr;
  unsigned* flags = nullptr;
  unsigned* err = nullptr;
  unsigned cum[CH_RING] = {0, 0, 0, 0};
  long k = 0;
  // fills the hand-off fields of `a` for the next launch with `wgs` workgroups and returns the stream to launch it on
  hipStream_t next(StageArgs& a, int wgs) {
    if (!on) return s[0];
    const int slot = (int)(k % CH_RING), prev = (int)((k + CH_RING - 1) % CH_RING);
    a.ch_wait = k > 0 ? flags + prev * (CH_SHARDS * CH_STRIDE) : nullptr;
    a.wait_count = cum[prev];
    a.ch_arrive = flags + slot * (CH_SHARDS * CH_STRIDE);
    a.ch_err = err;
    a.ch_k = (int)k;
    cum[slot] += (unsigned)wgs;
    return s[k++ & 1];
  }
};

// ---- bound on how far the host runs ahead of the device inside a sweep (option "launch_window" = steps between two marks, 0 = off):
// every `g_launch_window` steps an event is recorded, and the host waits for the mark before the previous one -- at most about
// 2 x window x (launches per step) launches are outstanding.  The host enqueues a stage launch in ~3.4 us (tools/launch_cost.hip),
// the device takes 6-15 us for one, so an unbounded loop fills the hardware queue within a sweep; with a full queue the same
// launches run slower (H = 512, B = 32: forward / BPTT step 26.2 / 33.1 us unbounded, 24.8 / 29.8 us with a window of 8 steps;
// FiLM: 45.6 / 54.4 -> 44.5 / 49.2 us) and, on some boxes of the pool, a sweep that follows another engine's in the same process
// dropped to a third of its speed (bench extras: 19.5 -> 45 ms per iteration).
struct LaunchWindow {
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  int n = 0, steps = 0, dev = -1;
  bool on = false;
  // thread_local instance: the events go when the host thread does (programs that create many threads leaked three per thread)
  ~LaunchWindow() {
    for (hipEvent_t& e : ev)
      if (e) { (void)hipEventDestroy(e); e = nullptr; }
  }
  int begin(hipStream_t s) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) cap = hipStreamCaptureStatusActive;
    on = g_launch_window > 0 && cap == hipStreamCaptureStatusNone;
    n = steps = 0;
    if (!on) return 0;
    int cur = 0;
    ZCHECK(hipGetDevice(&cur) == hipSuccess, "hipGetDevice failed");
    if (ev[0] && cur != dev) {        // this host thread moved to another device: events belong to the device they were created on
      for (hipEvent_t& e : ev) { (void)hipEventDestroy(e); e = nullptr; }
    }
    if (!ev[0]) {
      for (hipEvent_t& e : ev) ZCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess, "event creation failed");
      dev = cur;
    }
    return 0;
  }
  int tick(hipStream_t s, int mult = 1) {     // mult: launches per step relative to a decoder step's three or four
    if (!on || ++steps < g_launch_window * mult) return 0;
    steps = 0;
    ZCHECK(hipEventRecord(ev[n % 3], s) == hipSuccess, "hipEventRecord failed");
    if (n >= 2) ZCHECK(hipEventSynchronize(ev[(n - 2) % 3]) == hipSuccess, "hipEventSynchronize failed");
    ++n;
    return 0;
  }
};
thread_local LaunchWindow t_window;

StageArgs base_args(const ZeggsDecDims& d, const ZeggsDecStats* st, const DecWs& w) {
  StageArgs a;
  memset(&a, 0, sizeof(a));
  a.d = d; a.st = *st; a.NB = w.NB; a.GL = w.GL; a.XD = w.XD; a.POL = w.POL; a.variant = g_stage_variant;
  return a;
}

}  // namespace

// rnn_cond "film" (RecurrentDecoderFiLM, ZEGGS/modules.py:188-227) runs the same stage kernels, four launches per step and
// direction: S1* layer0 + modulation (folded into the previous step's output launch: M = W0[:, :PO] D W3), S2 / S3 the GRU
// layers, S4a F2 = FiLM(ELU(layer2)), S4b layer3 + root integration; backward B1 (W2^T D2), B2, B3 (+ layer0's modulation
// backward), B4 dx + dF2 of the previous step through the same fold.  The modulation sits between two matrix products as an
// elementwise factor that depends on (batch row, frame), so each modulated layer stays a dependent stage; the persistent
// H = 1024 kernels decline it.
int dec_fast_supported(const ZeggsDecDims& d) { return d.H % 16 == 0 && d.B <= 64 && d.PI == d.PO + 3 && d.PO >= 16; }

int dec_fast_pack_fwd(const ZeggsDecDims& d, const ZeggsDecParams* P, DecWs& w, hipStream_t s) {
  const int H = d.H, XD = w.XD;
  ZTRY(pack(w.pw_l0, P->l0_w, w.nTH, w.KBX, 0, XD, H, H, d.PO, XD, 0, s));
  ZTRY(pack(w.pw_ih0h, P->w_ih0, w.nT5, w.KBH, 1, H, 3 * H, H, d.PO, H + XD, 0, s, w.TG0));
  ZTRY(pack(w.pw_ih0x, P->w_ih0, w.nT5, w.KBX, 1, XD, 3 * H, H, d.PO, H + XD, H, s, w.TG0));
  ZTRY(pack(w.pw_hh0, P->w_hh0, w.nT5, w.KBH, 1, H, 3 * H, H, d.PO, H, 0, s, w.TG0));
  ZTRY(pack(w.pw_ih1, P->w_ih1, w.nT5, w.KBH, 1, H, 3 * H, H, d.PO, H, 0, s, w.TG1));
  ZTRY(pack(w.pw_hh1, P->w_hh1, w.nT5, w.KBH, 1, H, 3 * H, H, d.PO, H, 0, s, w.TG1));
  if (d.film) {
    ZCHECK(P->l3_w && P->l3_b && P->g_w && P->g_b && P->be_w && P->be_b, "decoder: film parameters missing");
    ZTRY(pack(w.pw_l2, P->l2_w, w.nTH, w.KBH, 0, H, H, H, d.PO, H, 0, s));
    ZTRY(pack(w.pw_l3, P->l3_w, w.nTPO, w.KBH, 0, H, d.PO, H, d.PO, H, 0, s));
    return 0;
  }
  ZTRY(pack(w.pw_l2, P->l2_w, w.nTPO, w.KBH, 0, H, d.PO, H, d.PO, H, 0, s));
  return 0;
}

// operands of the merged stage (layer2 of step t folded into layer0 of step t+1): between the two layers the
// reference only de-normalises / re-normalises the pose columns (ZEGGS/modules.py:60-76), which is affine, so
//   W0 x_{t+1} + b0 = M h1_t + Wc cond_{t+1} + W0[:, gaze] g_{t+1} + cvec
// with M = W0[:, :PO] diag(sigma_o/sigma_i) W2.  Only the 3 gaze columns depend on the (non-linear) root integration.
void dec_timing_mark(int i, hipStream_t s) { timing_mark(i, s); }

// canonical operands of the folded stage: Mc = W0[:, :PO] diag(sigma_o/sigma_i) W2 [H,H], cvec [H]
int dec_fast_merge_prep(const ZeggsDecDims& d, const ZeggsDecParams* P, const ZeggsDecStats* st, DecWs& w, hipStream_t s) {
  const int H = d.H, XD = w.XD;
  const long n = (long)H * w.POL;
  const float *ow = d.film ? P->l3_w : P->l2_w, *ob = d.film ? P->l3_b : P->l2_b;     // the output layer [PO,H]
  hipLaunchKernelGGL(merge_prep_k, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, s, w.W0s,
                     w.vvec, P->l0_w, ob, *st, H, d.PO, w.POL, XD, 0);
  ZLAUNCH_CHECK("merge_prep");
  ZTRY(gemm_nn(w.W0s, w.POL, ow, H, w.Mc, H, H, d.PO, H, 0.f, s));
  ZTRY(gemm_nt(w.vvec, w.POL, P->l0_w, XD, w.cvec, H, P->l0_b, 1, H, d.PO, ACT_NONE, 0.f, s));
  return 0;
}

int dec_fast_pack_merged(const ZeggsDecDims& d, const ZeggsDecParams* P, const ZeggsDecStats* st, DecWs& w, hipStream_t s) {
  const int H = d.H, XD = w.XD;
  ZTRY(dec_fast_merge_prep(d, P, st, w, s));
  ZTRY(pack(w.pw_m, w.Mc, w.nTH, w.KBH, 0, H, H, H, d.PO, H, 0, s, w.TMC));
  ZTRY(pack(w.pw_c, P->l0_w, w.nTH, w.KBC, 0, d.SP + (d.film ? 0 : d.ST), H, H, d.PO, XD, d.PI, s, w.TMC));
  ZTRY(pack(w.pw_l2c, d.film ? P->l3_w : P->l2_w, w.nTH, w.KBH, 4, H, 16, H, d.PO, H, 0, s, w.TMC));   // the output layer's root tile, per tile
  return 0;
}

int dec_fast_pack_bwd(const ZeggsDecDims& d, const ZeggsDecParams* P, const ZeggsDecStats* st, DecWs& w, hipStream_t s) {
  const int H = d.H, XD = w.XD;
  if (d.film) {
    ZTRY(pack(w.pb_l3, P->l3_w, w.nTH, w.KBPO, 2, d.PO, H, H, d.PO, H, 0, s));         // V[U][c] = W3[c][U]
    ZTRY(pack(w.pb_l2, P->l2_w, w.nTH, w.KBH, 2, H, H, H, d.PO, H, 0, s));             // V[U][k] = W2[k][U]
  }
  if (!(g_stage_variant & 8192) && d.T > 2) {
    // merged stage (dx of step t + layer-1 gates of step t-1): M' = W0[:, 6:PO] diag(sigma_o/sigma_i) W2[6:PO, :],
    // packed transposed (V[U][k] = M'[k][U]); the six root columns take the non-linear root-integration path.
    // W0s / Mc are the forward pass's scratch, free again by now.
    const long n = (long)H * w.POL;
    hipLaunchKernelGGL(merge_prep_k, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, s, w.W0s,
                       (float*)nullptr, P->l0_w, P->l2_b, *st, H, d.PO, w.POL, XD, 6);
    ZLAUNCH_CHECK("merge_prep");
    ZTRY(gemm_nn(w.W0s, w.POL, d.film ? P->l3_w : P->l2_w, H, w.Mc, H, H, d.PO, H, 0.f, s));   // (film: the output layer is layer3)
    ZTRY(pack(w.pb_mt, w.Mc, w.nTH, w.KBH, 2, H, H, H, d.PO, H, 0, s));
  }
  if (!d.film) ZTRY(pack(w.pb_l2, P->l2_w, w.nTH, w.KBPO, 2, d.PO, H, H, d.PO, H, 0, s));          // V[U][c] = W2[c][U]
  ZTRY(pack(w.pb_ih1, P->w_ih1, w.nTH, w.KB3H, 2, 3 * H, H, H, d.PO, H, 0, s));        // V[U][k] = W_ih1[k][U]
  ZTRY(pack(w.pb_hh1, P->w_hh1, w.nTH, w.KB3H, 2, 3 * H, H, H, d.PO, H, 0, s));
  ZTRY(pack(w.pb_ih0, P->w_ih0, w.nTGI, w.KB3H, 2, 3 * H, H + XD, H, d.PO, H + XD, 0, s));
  ZTRY(pack(w.pb_hh0, P->w_hh0, w.nTH, w.KB3H, 2, 3 * H, H, H, d.PO, H, 0, s));
  ZTRY(pack(w.pb_l0, P->l0_w, w.nTX, w.KBH, 3, H, XD, H, d.PO, XD, 0, s));             // V[q][u] = W0[u][perm(q)]
  return 0;
}

int dec_fast_fwd_steps(const ZeggsDecDims& d, const ZeggsDecParams* P, const ZeggsDecStats* st, DecWs& w,
                       const float* gaze, const float* speech, const float* style, float* pose, float* rpos,
                       float* rrot, int training, hipStream_t s) {
  const int B = d.B, T = d.T, H = d.H, NB = w.NB;
  const long sG = (long)B * w.GL, sH = (long)B * H;
  const long XB = 256L * NB;
  auto cs = [&](int t) { return training ? t : (t & 1); };
  float* Xxf[2] = {w.Xxf, w.Xxf + w.KBX * XB};
  float* H0xf[2] = {w.H0xf, w.H0xf + w.KBH * XB};
  float* H1xf[2] = {w.H1xf, w.H1xf + w.KBH * XB};
  ZTRY(k_fill(w.xf_base_fwd, (long)(w.xf_bytes_fwd / 4), 0.f, s));
  if (T < 2) return 0;
  // fragments of the initial state and of x_1 (written in canonical form by the init kernel / CSE GEMMs)
  auto conv = [&](float* xf, const float* src, long ld, int off, int K) {
    long n = (long)B * K, g = (n + 255) / 256;
    hipLaunchKernelGGL(to_xfrag_k, dim3((unsigned)(g > 1024 ? 1024 : g)), dim3(256), 0, s, xf, src, ld, off, K, B, NB);
  };
  conv(H0xf[0], w.H0 + cs(0) * sH, H, 0, H);
  conv(H1xf[0], w.H1 + cs(0) * sH, H, 0, H);
  conv(Xxf[1], w.Gin + cs(1) * sG, w.GL, H, w.XD);
  ZLAUNCH_CHECK("to_xfrag");
  // tiny-batch decode (B <= 2, no_grad; measured: B >= 3 is faster on the MFMA path): GEMV stage kernels over the
  // canonical activations (decoder.hip zero-fills the Gin ring first: the pad columns of x are read against zero
  // weights and must be finite)
  const bool gemv = !training && B <= 2 && !(g_stage_variant & 1024);
  // 3 launches per step: layer2 of step t and layer0 of step t+1 run in ONE launch (variant 4096: 4 launches)
  const bool merged = !(g_stage_variant & 4096);
  if (merged && T > 2) ZTRY(dec_fast_pack_merged(d, P, st, w, s));
  Chain ch;
  ch.s[0] = ch.s[1] = s;
  {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) cap = hipStreamCaptureStatusActive;   // a failed query must not read as "not capturing"
    const int per_wave = (w.KBH + w.KBX + w.KBH + 7) / 8;            // the widest stage (GRU layer 0)
    if (g_chain && gemv && !d.film && T > 2 && cap == hipStreamCaptureStatusNone && per_wave <= CH_GPRE &&
        !(g_stage_variant & (16384 | V_NOW | V_NOEPI))) {
      ZTRY(chain_stream(&ch.cs));
      ch.on = true;
      ch.s[1] = ch.cs->s;
      ch.flags = w.chain;                                            // zeroed by the k_fill above
      ch.err = w.chain + CH_RING * CH_SHARDS * CH_STRIDE;
      ZCHECK(hipEventRecord(ch.cs->start, s) == hipSuccess, "hipEventRecord failed");
      ZCHECK(hipStreamWaitEvent(ch.s[1], ch.cs->start, 0) == hipSuccess, "hipStreamWaitEvent failed");
    }
  }
  timing_mark(0, s);
  ZTRY(t_window.begin(s));
  for (int t = 1; t < T; ++t) {
    ZTRY(t_window.tick(s));
    const int c = t & 1, p = (t - 1) & 1;
    const long o = (long)t * sH;
    const bool next = t + 1 < T;
    StageArgs a = base_args(d, st, w);
    a.t = t; a.gaze = gaze; a.speech = speech; a.style = style; a.pose = pose; a.rpos = rpos; a.rrot = rrot;
    a.gemv = gemv;
    const float* gin_c = w.Gin + cs(t) * sG;
    float* gin_n = w.Gin + cs(t + 1) * sG;
    const float *h0p = w.H0 + cs(t - 1) * sH, *h1p = w.H1 + cs(t - 1) * sH;
    float *h0c = w.H0 + cs(t) * sH, *h1c = w.H1 + cs(t) * sH;
    const float *gam = nullptr, *bet = nullptr;
    const float *gam_n = nullptr, *bet_n = nullptr;   // ... and those of step t + 1 (merged launch: layer0 of the next step)
    if (d.film) {
      const long sg = (long)B * 2 * H;
      // ring path: frame u's vectors live in slot (u - 1) % (2 FILM_GB); block i = frames [1 + i GB, 1 + (i + 1) GB) fills half
      // i & 1 of the ring with ONE batched product per predictor (rows = frames, batch = batch rows of style [B, T, ST]) when
      // the step in front of it is reached -- its other half still holds the current block (training: every frame's, decoder.hip)
      auto gslot = [&](int u) { return training ? (long)u : (long)((u - 1) % (2 * FILM_GB)); };
      if (!training && (t == 1 || (t % FILM_GB == 0 && t + 1 < T))) {
        const int f0 = t == 1 ? 1 : t + 1;                 // first frame of the block to compute
        const int nfr = (T - f0) < FILM_GB ? (T - f0) : FILM_GB;
        for (int which = 0; which < 2 && nfr > 0; ++which) {
          GemmArgs g = gemm_args(style + (long)f0 * d.ST, which ? P->be_w : P->g_w, (which ? w.BET : w.GAM) + gslot(f0) * sg, nfr, 2 * H, d.ST);
          g.sam = d.ST; g.sak = 1; g.sbk = 1; g.sbn = d.ST; g.scm = sg; g.scn = 1;
          g.bsA0 = (long)T * d.ST; g.bsC0 = 2 * H; g.nb1 = 1; g.bias = which ? P->be_b : P->g_b;
          ZTRY(launch_gemm(g, B, s));
        }
      }
      gam = w.GAM + gslot(t) * sg; bet = w.BET + gslot(t) * sg;
      gam_n = w.GAM + gslot(t + 1) * sg; bet_n = w.BET + gslot(t + 1) * sg;
    }
    if (t == 1 || !merged) {
      // S1: hid = ELU(W0 x + b0)   [film: modulated]
      a.g[0] = Grp{}; a.g[1] = Grp{};
      a.g[0].seg[0] = seg(w.pw_l0, Xxf[c], w.KBX, 0, gin_c + H, w.GL); a.g[0].nseg = 1; a.g[0].tiles = w.nTH;
      a.g[0].wcat = w.pw_l0; a.g[0].tkbcat = w.KBX; a.g[0].kacc = w.KBX;
      a.g[0].epi = d.film ? EPI_ELU_FILM : EPI_ELU_HID;
      a.g[0].p0 = P->l0_b; a.g[0].o0 = w.Gin + cs(t) * sG; a.g[0].o1 = gemv ? nullptr : w.HIDxf;
      if (d.film) { a.g[0].p1 = gam; a.g[0].p2 = bet; a.g[0].o2 = training ? w.A0 + o : nullptr; }
      ZTRY(launch_stage(a, ch.next(a, stage_wgs(a))));
    }
    // S2: GRU layer 0
    a.g[0] = Grp{}; a.g[1] = Grp{};
    a.g[0].seg[0] = seg(w.pw_ih0h, w.HIDxf, w.KBH, 0, gin_c, w.GL, 0, w.TG0);
    a.g[0].seg[1] = seg(w.pw_ih0x, Xxf[c], w.KBX, 0, gin_c + H, w.GL, 0, w.TG0);
    a.g[0].seg[2] = seg(w.pw_hh0, H0xf[p], w.KBH, 1, h0p, H, 0, w.TG0);
    a.g[0].wcat = w.pw_g0; a.g[0].tkbcat = w.TG0; a.g[0].kacc = w.KBH + w.KBX;
    a.g[0].nseg = 3; a.g[0].tiles = w.nT5; a.g[0].epi = EPI_GRU_FWD;
    a.g[0].p0 = P->b_ih0; a.g[0].p1 = P->b_hh0; a.g[0].p2 = h0p;
    a.g[0].o0 = h0c; a.g[0].o1 = gemv ? nullptr : H0xf[c];
    if (training) a.g[0].o2 = w.GT0 + 4 * o;
    ZTRY(launch_stage(a, ch.next(a, stage_wgs(a))));
    // S3: GRU layer 1 (+ stages the speech/style columns of x_{t+1}: ring slots last read one step ago)
    a.g[0] = Grp{};
    a.g[0].seg[0] = seg(w.pw_ih1, H0xf[c], w.KBH, 0, h0c, H, 0, w.TG1);
    a.g[0].seg[1] = seg(w.pw_hh1, H1xf[p], w.KBH, 1, h1p, H, 0, w.TG1);
    a.g[0].wcat = w.pw_g1; a.g[0].tkbcat = w.TG1; a.g[0].kacc = w.KBH;
    a.g[0].nseg = 2; a.g[0].tiles = w.nT5; a.g[0].epi = EPI_GRU_FWD;
    a.g[0].p0 = P->b_ih1; a.g[0].p1 = P->b_hh1; a.g[0].p2 = h1p;
    a.g[0].o0 = h1c; a.g[0].o1 = gemv ? nullptr : H1xf[c];
    if (training) a.g[0].o2 = w.GT1 + 4 * o;
    if (next) {
      a.cf_gin = training ? nullptr : gin_n;            // training: filled for every t by dec_fill_cond_k
      a.cf_x = gemv ? nullptr : Xxf[(t + 1) & 1];
      a.cf_cond = (gemv || !merged) ? nullptr : w.CONDxf;
    }
    ZTRY(launch_stage(a, ch.next(a, stage_wgs(a))));
    a.cf_gin = a.cf_x = a.cf_cond = nullptr;
    if (d.film) {
      // S4a: F2 = FiLM(ELU(W2 h1 + b2))
      float* f2 = w.F2 + cs(t) * sH;
      a.g[0] = Grp{}; a.g[1] = Grp{};
      a.g[0].seg[0] = seg(w.pw_l2, H1xf[c], w.KBH, 0, h1c, H); a.g[0].nseg = 1; a.g[0].tiles = w.nTH;
      a.g[0].epi = EPI_ELU_FILM; a.g[0].ld0 = H;
      a.g[0].p0 = P->l2_b; a.g[0].p1 = gam + H; a.g[0].p2 = bet + H;
      a.g[0].o0 = f2; a.g[0].o1 = gemv ? nullptr : w.F2xf; a.g[0].o2 = training ? w.A2 + o : nullptr;
      ZTRY(launch_stage(a, s));
      // S4b: y = W3 F2 + b3 -> pose[t], root integration, pose/gaze columns of x_{t+1}
      //      [merged: + hid_{t+1} = FiLM(ELU(M F2 + Wc speech_{t+1} + W0[:, gaze] g_{t+1} + cvec)), M = W0[:, :PO] D W3]
      a.g[0] = Grp{};
      a.g[0].seg[0] = seg(w.pw_l3, w.F2xf, w.KBH, 0, f2, H); a.g[0].nseg = 1; a.g[0].tiles = w.nTPO;
      a.g[0].epi = EPI_OUT_FWD;
      a.g[0].p0 = P->l3_b; a.g[0].o0 = next ? gin_n : nullptr;
      a.g[0].o1 = gemv ? nullptr : Xxf[(t + 1) & 1];
      if (merged && next) {
        a.g[1].seg[0] = seg(w.pw_m, w.F2xf, w.KBH, 0, f2, H, 0, w.TMC);
        a.g[1].seg[1] = seg(w.pw_c, w.CONDxf, w.KBC, 0, gin_n + H + d.PI, w.GL, 0, w.TMC);
        a.g[1].seg[2] = seg(w.pw_l3, w.F2xf, w.KBH, 1, f2, H, 1);          // every workgroup: layer3's tile 0 (L2-resident)
        a.g[1].nseg = 3; a.g[1].tiles = w.nTH; a.g[1].epi = EPI_HID_MERGED;
        a.g[1].p0 = w.cvec; a.g[1].p1 = P->l0_w; a.g[1].p2 = P->l3_b; a.g[1].p3 = gam_n; a.g[1].p4 = bet_n;
        a.g[1].o0 = gin_n; a.g[1].o1 = gemv ? nullptr : w.HIDxf; a.g[1].o2 = training ? w.A0 + o + sH : nullptr;
      }
      ZTRY(launch_stage(a, s));
      continue;
    }
    // S4: y = W2 h1 + b2 -> pose[t], root integration, pose/gaze columns of x_{t+1}
    //     [merged: + hid_{t+1} = ELU(M h1 + Wc cond_{t+1} + W0[:, gaze] g_{t+1} + cvec) in the same launch]
    a.g[0] = Grp{}; a.g[1] = Grp{};
    a.g[0].seg[0] = seg(w.pw_l2, H1xf[c], w.KBH, 0, h1c, H); a.g[0].nseg = 1; a.g[0].tiles = w.nTPO;
    a.g[0].wcat = w.pw_l2; a.g[0].tkbcat = w.KBH; a.g[0].kacc = w.KBH;
    a.g[0].epi = EPI_OUT_FWD;
    a.g[0].p0 = P->l2_b; a.g[0].o0 = next ? gin_n : nullptr;
    a.g[0].o1 = gemv ? nullptr : Xxf[(t + 1) & 1];
    if (merged && next) {
      a.g[1].seg[0] = seg(w.pw_m, H1xf[c], w.KBH, 0, h1c, H, 0, w.TMC);
      a.g[1].seg[1] = seg(w.pw_c, w.CONDxf, w.KBC, 0, gin_n + H + d.PI, w.GL, 0, w.TMC);
      a.g[1].seg[2] = seg(w.pw_l2, H1xf[c], w.KBH, 1, h1c, H, 1);      // every workgroup: layer2's tile 0 (L2-resident)
      a.g[1].wcat = w.pw_mc; a.g[1].tkbcat = w.TMC; a.g[1].kacc = w.KBH + w.KBC;   // chained GEMV: the per-tile copy
      a.g[1].nseg = 3; a.g[1].tiles = w.nTH; a.g[1].epi = EPI_HID_MERGED;
      a.g[1].p0 = w.cvec; a.g[1].p1 = P->l0_w; a.g[1].p2 = P->l2_b;
      a.g[1].o0 = gin_n; a.g[1].o1 = gemv ? nullptr : w.HIDxf;
    }
    ZTRY(launch_stage(a, ch.next(a, stage_wgs(a))));
  }
  if (ch.on) {   // the caller's stream continues after the last launch of the second stream
    ZCHECK(hipEventRecord(ch.cs->done, ch.s[1]) == hipSuccess, "hipEventRecord failed");
    ZCHECK(hipStreamWaitEvent(s, ch.cs->done, 0) == hipSuccess, "hipStreamWaitEvent failed");
  }
  timing_mark(1, s);
  return 0;
}

int dec_fast_bwd_steps(const ZeggsDecDims& d, const ZeggsDecParams* P, const ZeggsDecStats* st, DecWs& w,
                       const float* gaze, const float* pose, const float* rpos, const float* rrot,
                       const float* dpose, const float* drpos, const float* drrot, int t_hi, int t_lo, hipStream_t s) {
  const int B = d.B, T = d.T, H = d.H, NB = w.NB;
  const long sG = (long)B * w.GL, sH = (long)B * H, s3 = 3 * sH;
  if (T < 2) return 0;
  if (t_hi == T - 1) {   // first chunk of the sweep
    ZTRY(k_fill(w.xf_base_bwd, (long)(w.xf_bytes_bwd / 4), 0.f, s));
    const bool merged0 = !(g_stage_variant & 8192) && T > 2;   // carry slot read by the first dx stage (see below)
    hipLaunchKernelGGL(dy_last_k, dim3(B), dim3(256), 0, s, d, *st, dpose, drpos, drrot, gaze, pose, rpos, rrot,
                       w.carry + (merged0 ? (long)((T - 1) & 1) * B * 8 : 0), w.DY + (long)(T - 1) * B * w.POL, w.POL,
                       w.DYxf, NB);
    ZLAUNCH_CHECK("dy_last");
  }
  // 3 launches per step: the dx stage of step t also evaluates the layer-1 gate gradients of step t-1 (variant 8192:
  // separate launches).  The root-state carry is double-buffered: slot (t & 1) is read, slot ((t - 1) & 1) written.
  const bool merged = !(g_stage_variant & 8192) && T > 2;
  if (t_hi == T - 1) timing_mark(2, s);
  ZTRY(t_window.begin(s));
  for (int t = t_hi; t >= t_lo; --t) {
    ZTRY(t_window.tick(s));
    const long o = (long)t * sH;
    StageArgs a = base_args(d, st, w);
    a.t = t; a.gaze = gaze; a.cpose = pose; a.crpos = rpos; a.crrot = rrot; a.dpose = dpose; a.drpos = drpos;
    a.drrot = drrot;
    float* c_in = w.carry + (merged ? (long)(t & 1) * B * 8 : 0);
    float* c_out = w.carry + (merged ? (long)((t - 1) & 1) * B * 8 : 0);
    a.carry = c_in; a.carry_out = c_out;
    if (d.film && (t == T - 1 || !merged)) {
      // B0: dF2 = W3^T dy -> D2 (through the modulation and layer2's ELU), dgamma / dbeta of layer2's half
      const long og = (long)t * B * 2 * H;
      a.g[0] = Grp{}; a.g[1] = Grp{};
      a.g[0].seg[0] = seg(w.pb_l3, w.DYxf, w.KBPO, 0); a.g[0].nseg = 1; a.g[0].tiles = w.nTH; a.g[0].epi = EPI_FILM_BWD;
      a.g[0].p0 = w.A2 + o; a.g[0].p1 = w.GAM + og + H;
      a.g[0].o0 = w.D2 + o; a.g[0].o1 = w.D2xf; a.g[0].o2 = w.DGAM + og + H; a.g[0].o3 = w.DBET + og + H;
      ZTRY(launch_stage(a, s));
    }
    if (t == T - 1 || !merged || d.film) {
      // B1: dH1 = W2^T dy + carry -> layer-1 gate gradients   [film: W2^T D2, D2 from B0 or from the merged launch of step t+1]
      a.g[0] = Grp{}; a.g[1] = Grp{};
      a.g[0].seg[0] = d.film ? seg(w.pb_l2, w.D2xf, w.KBH, 0) : seg(w.pb_l2, w.DYxf, w.KBPO, 0);
      a.g[0].nseg = 1; a.g[0].tiles = w.nTH; a.g[0].epi = EPI_GRU_BWD;
      a.g[0].p0 = w.GT1 + 4 * o; a.g[0].p4 = w.H1 + o - sH;
      a.g[0].o0 = w.dH1c; a.g[0].o1 = w.DI1 + t * s3; a.g[0].o2 = w.DH1 + o; a.g[0].o3 = w.DI1xf; a.g[0].o4 = w.DH1xf;
      ZTRY(launch_stage(a, s));
    }
    // B2: dH0 = W_ih1^T di1 + carry -> layer-0 gate gradients ; dH1 carry += W_hh1^T dh1
    a.g[0] = Grp{}; a.g[1] = Grp{};
    a.g[0].seg[0] = seg(w.pb_ih1, w.DI1xf, w.KB3H, 0); a.g[0].nseg = 1; a.g[0].tiles = w.nTH; a.g[0].epi = EPI_GRU_BWD;
    a.g[0].p0 = w.GT0 + 4 * o; a.g[0].p4 = w.H0 + o - sH;
    a.g[0].o0 = w.dH0c; a.g[0].o1 = w.DI0 + t * s3; a.g[0].o2 = w.DH0 + o; a.g[0].o3 = w.DI0xf; a.g[0].o4 = w.DH0xf;
    // W_hh^T dh: the r, z rows of dh are those of di (first 2H k-values of the DI fragments), the n rows come compact
    a.g[1].seg[0] = subseg(w.pb_hh1, w.DI1xf, 0, 2 * w.KBH, w.KB3H, 0);
    a.g[1].seg[1] = subseg(w.pb_hh1, w.DH1xf, 2 * w.KBH, w.KBH, w.KB3H, 0);
    a.g[1].nseg = 2; a.g[1].tiles = w.nTH; a.g[1].epi = EPI_ADD;
    a.g[1].o0 = w.dH1c;
    ZTRY(launch_stage(a, s));
    // B3: dGin = W_ih0^T di0 -> [D0 | dx part (+ the r operand of the merged stage)] ; dH0 carry += W_hh0^T dh0
    a.g[0] = Grp{}; a.g[1] = Grp{};
    a.g[0].seg[0] = seg(w.pb_ih0, w.DI0xf, w.KB3H, 0); a.g[0].nseg = 1; a.g[0].tiles = w.nTGI; a.g[0].epi = EPI_DGIN;
    a.g[0].p0 = w.Gin + t * sG; a.g[0].o0 = w.D0 + o; a.g[0].o1 = w.D0xf; a.g[0].o2 = w.dXa;
    if (d.film) {   // layer0's modulation: D0 through (1 + gamma) ELU'(A0), dgamma / dbeta of layer0's half
      const long og = (long)t * B * 2 * H;
      a.g[0].p1 = w.A0 + o; a.g[0].p2 = w.GAM + og; a.g[0].o3 = w.DGAM + og; a.g[0].o4 = w.DBET + og;
    }
    a.g[1].seg[0] = subseg(w.pb_hh0, w.DI0xf, 0, 2 * w.KBH, w.KB3H, 0);
    a.g[1].seg[1] = subseg(w.pb_hh0, w.DH0xf, 2 * w.KBH, w.KBH, w.KB3H, 0);
    a.g[1].nseg = 2; a.g[1].tiles = w.nTH; a.g[1].epi = EPI_ADD;
    a.g[1].o0 = w.dH0c;
    a.rxf = (merged && t > 1) ? w.Rxf : nullptr;
    ZTRY(launch_stage(a, s));
    a.rxf = nullptr;
    // B4: dx_t = dx part + W0^T D0 -> dy_{t-1}, root-integration backward
    //     [merged, t > 1: + layer-1 gate gradients of step t-1 from dH1 = M'^T D0 + W2^T r + W2[0:6]^T dy6 + carry]
    a.g[0] = Grp{}; a.g[1] = Grp{};
    a.g[0].seg[0] = seg(w.pb_l0, w.D0xf, w.KBH, 0); a.g[0].nseg = 1; a.g[0].tiles = w.nTX; a.g[0].epi = EPI_DX;
    a.g[0].p0 = w.dXa; a.g[0].o0 = w.DX + (long)t * B * w.XD;
    a.g[0].o1 = t > 1 ? w.DY + (long)(t - 1) * B * w.POL : nullptr; a.g[0].o2 = w.DYxf;
    if (merged && t > 1 && d.film) {   // + dF2 of step t-1 and layer2's modulation / ELU backward (EPI_FILM_BWD_M)
      const long o1 = o - sH, og1 = (long)(t - 1) * B * 2 * H;
      a.g[1].seg[0] = seg(w.pb_mt, w.D0xf, w.KBH, 0);
      a.g[1].seg[1] = seg(w.pb_l3, w.Rxf, w.KBPO, 0);
      a.g[1].seg[2] = seg(w.pb_l0, w.D0xf, w.KBH, 1, nullptr, 0, 1);
      a.g[1].nseg = 3; a.g[1].tiles = w.nTH; a.g[1].epi = EPI_FILM_BWD_M;
      a.g[1].p0 = w.A2 + o1; a.g[1].p1 = w.GAM + og1 + H;
      a.g[1].o0 = w.D2 + o1; a.g[1].o1 = w.D2xf; a.g[1].o2 = w.DGAM + og1 + H; a.g[1].o3 = w.DBET + og1 + H;
      a.aux0 = w.dXa; a.aux1 = P->l3_w;
    } else
    if (merged && t > 1) {
      const long o1 = o - sH;
      a.g[1].seg[0] = seg(w.pb_mt, w.D0xf, w.KBH, 0);
      a.g[1].seg[1] = seg(w.pb_l2, w.Rxf, w.KBPO, 0);
      a.g[1].seg[2] = seg(w.pb_l0, w.D0xf, w.KBH, 1, nullptr, 0, 1);
      a.g[1].nseg = 3; a.g[1].tiles = w.nTH; a.g[1].epi = EPI_GRU_BWD_M;
      a.g[1].p0 = w.GT1 + 4 * o1; a.g[1].p4 = w.H1 + o1 - sH;
      a.g[1].o0 = w.dH1c; a.g[1].o1 = w.DI1 + (t - 1) * s3; a.g[1].o2 = w.DH1 + o1; a.g[1].o3 = w.DI1xf;
      a.g[1].o4 = w.DH1xf;
      a.aux0 = w.dXa; a.aux1 = P->l2_w;
    }
    ZTRY(launch_stage(a, s));
  }
  if (t_lo <= 1) timing_mark(3, s);
  return 0;
}

// ---- style encoder "gru" (style_gru.hip; reference StyleEncoderGRU, ZEGGS/modules.py:307-343): the forward-direction
// recurrence h_{t+1} = GRU(x_t, h_t) and its BPTT on the same stage kernels -- ONE launch per frame and direction instead of a
// skinny GEMM + gate kernel (+ copy) each.  The input-side pre-activations GI [L][B][3H] come from one time-batched GEMM
// (EPI_GRU_FWD reads them through Grp.p3); W_hh (3 MB at H = 512) stays L2-resident across the frames: the sweep is
// launch-bound.  Saved gates are float4 (r, z, n, W_hn h + b_hn) per unit; the hidden-side gate gradients are kept compact
// (their r, z rows are those of DI) as in the decoder's sweep.
extern int g_decoder_fast;
int sg_fast_supported(int B, int H) { return g_decoder_fast && H % 16 == 0 && H >= 16 && B >= 1 && B <= 64; }
SgFast sg_fast_carve(int B, int H, int L, Arena& a) {
  SgFast f;
  memset(&f, 0, sizeof(f));
  f.NB = (B + 15) / 16; f.nT5 = (H + 4) / 5; f.nTH = H / 16; f.KBH = H / 16; f.KB3H = 3 * H / 16;
  const long XB = 256L * f.NB;
  f.pw = a.f((long)f.nT5 * f.KBH * 256);
  f.pb = a.f((long)f.nTH * f.KB3H * 256);
  f.xf = a.f(2 * f.KBH * XB + 2 * f.KB3H * XB + 2 * f.KBH * XB);
  f.xf_floats = 2 * f.KBH * XB + 2 * f.KB3H * XB + 2 * f.KBH * XB;
  f.Hxf = f.xf; f.DIxf = f.xf + 2 * f.KBH * XB; f.DHxf = f.DIxf + 2 * f.KB3H * XB;
  f.GT = a.f((long)L * B * H * 4);
  f.DHn = a.f((long)L * B * H);
  return f;
}
// ---- replayed launch sequences (option "sweep_graphs" = 1; default off).  The frame sweeps of the style recurrence are static for
// given buffers, so they can be captured once into a hipGraph and replayed (key = every pointer / size the launches depend on; the
// workspace comes from a caching allocator, so the key repeats from iteration to iteration).  Measured (tools/launch_cost.hip,
// tools/order_probe.py): a replay costs the host nothing and 1.7 us per trivial node against 3.4 us per stream launch, but the
// nodes of OUR sweep take 5.5 us each either way (the dependent-dispatch latency is the device's), and the windowed stream
// launches (LaunchWindow below) finish the FiLM + GRU iteration 0.6-0.9 ms sooner than the replay (36.2 vs 36.9 ms): a replay
// puts all 384 nodes into the queue at once.  Kept for hosts that cannot issue a launch every ~5 us.
int g_sweep_graphs = 0;
namespace {
struct SweepGraph { uint64_t key[16]; hipGraphExec_t exec; unsigned long long used; };
std::mutex g_sweep_mu;
SweepGraph g_sweeps[8];
unsigned long long g_sweep_clock = 0;
long g_sweep_captures = 0, g_sweep_replays = 0;
template <class F>
int run_graphed(const uint64_t (&key)[16], hipStream_t s, F&& enqueue) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess) cap = hipStreamCaptureStatusActive;
  if (!g_sweep_graphs || cap != hipStreamCaptureStatusNone) return enqueue(s);
  std::lock_guard<std::mutex> lk(g_sweep_mu);
  SweepGraph* slot = &g_sweeps[0];
  for (SweepGraph& e : g_sweeps) {
    if (e.exec && memcmp(e.key, key, sizeof(key)) == 0) {
      e.used = ++g_sweep_clock;
      ++g_sweep_replays;
      ZCHECK(hipGraphLaunch(e.exec, s) == hipSuccess, "sweep graph: launch failed");
      return 0;
    }
    if (e.used < slot->used) slot = &e;      // least recently used (empty slots have used == 0)
  }
  // captured on a library-owned stream (the caller's may be the legacy default stream, which cannot capture), replayed on `s`
  static hipStream_t cs[16] = {};
  int dev = 0;
  ZCHECK(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16, "sweep graph: unsupported device index");
  if (!cs[dev]) ZCHECK(hipStreamCreateWithFlags(&cs[dev], hipStreamNonBlocking) == hipSuccess, "sweep graph: stream creation failed");
  ZCHECK(hipStreamBeginCapture(cs[dev], hipStreamCaptureModeThreadLocal) == hipSuccess, "sweep graph: begin capture failed");
  const int r = enqueue(cs[dev]);
  hipGraph_t graph = nullptr;
  const hipError_t e = hipStreamEndCapture(cs[dev], &graph);      // always: the stream must leave capture mode
  if (r != 0) { if (graph) (void)hipGraphDestroy(graph); return r; }
  ZCHECK(e == hipSuccess && graph, "sweep graph: capture failed");
  hipGraphExec_t exec = nullptr;
  const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  ZCHECK(ei == hipSuccess && exec, "sweep graph: instantiate failed");
  if (slot->exec) (void)hipGraphExecDestroy(slot->exec);
  memcpy(slot->key, key, sizeof(key));
  slot->exec = exec;
  slot->used = ++g_sweep_clock;
  ++g_sweep_captures;
  ZCHECK(hipGraphLaunch(exec, s) == hipSuccess, "sweep graph: launch failed");
  return 0;
}
}  // namespace
// how often a launch sequence was captured / replayed from the cache on this process (a capture per iteration would mean the
// buffers move between iterations and the cache never hits)
extern "C" int zeggs_sweep_graph_stats(long* captures, long* replays) {
  std::lock_guard<std::mutex> lk(g_sweep_mu);
  *captures = g_sweep_captures; *replays = g_sweep_replays;
  return 0;
}

static StageArgs sg_args(int B, int H, const SgFast& f) {
  StageArgs a;
  memset(&a, 0, sizeof(a));
  a.d.B = B; a.d.H = H; a.d.T = 1; a.NB = f.NB; a.variant = 0;
  return a;
}
// Hs [(L+1)][B][H] with slot 0 = the initial state (zero), GI [L][B][3H] incl. b_ih
int sg_fast_fwd(int B, int H, int L, const float* w_hh, const float* b_hh, const float* GI, float* Hs, const SgFast& f,
                hipStream_t s_) {
  const long sH = (long)B * H, XB = 256L * f.NB;
  const uint64_t key[16] = {1, (uint64_t)B, (uint64_t)H, (uint64_t)L, (uint64_t)w_hh, (uint64_t)b_hh, (uint64_t)GI, (uint64_t)Hs,
                            (uint64_t)f.pw, (uint64_t)f.xf, (uint64_t)f.GT, 0, 0, 0, 0, 0};
  return run_graphed(key, s_, [&](hipStream_t s) -> int {
  ZTRY(pack(f.pw, w_hh, f.nT5, f.KBH, 1, H, 3 * H, H, 0, H, 0, s));
  ZTRY(k_fill(f.xf, f.xf_floats, 0.f, s));
  ZTRY(t_window.begin(s));
  for (int t = 0; t < L; ++t) {
    ZTRY(t_window.tick(s, 3));
    StageArgs a = sg_args(B, H, f);
    float *hx_in = f.Hxf + (long)(t & 1) * f.KBH * XB, *hx_out = f.Hxf + (long)((t + 1) & 1) * f.KBH * XB;
    a.g[0].seg[0] = seg(f.pw, hx_in, f.KBH, 1, Hs + t * sH, H); a.g[0].nseg = 1; a.g[0].tiles = f.nT5;
    a.g[0].epi = EPI_GRU_FWD;
    a.g[0].p0 = b_hh; a.g[0].p1 = b_hh; a.g[0].p2 = Hs + t * sH; a.g[0].p3 = GI + (long)t * 3 * sH;
    a.g[0].o0 = Hs + (t + 1) * sH; a.g[0].o1 = hx_out; a.g[0].o2 = f.GT + 4 * t * sH;
    ZTRY(launch_stage(a, s));
  }
  return 0;
  });
}
// dhc [B][H]: in = gradient wrt h_L (the last state), out = gradient wrt h_0; DI [L][B][3H] (grad wrt the input-side
// pre-activations) and f.DHn [L][B][H] (n rows of the hidden side) are left for the weight-gradient GEMMs
int sg_fast_bwd(int B, int H, int L, const float* w_hh, const float* Hs, float* DI, float* dhc, const SgFast& f, hipStream_t s_) {
  const long sH = (long)B * H, XB = 256L * f.NB;
  const uint64_t key[16] = {2, (uint64_t)B, (uint64_t)H, (uint64_t)L, (uint64_t)w_hh, (uint64_t)Hs, (uint64_t)DI, (uint64_t)dhc,
                            (uint64_t)f.pb, (uint64_t)f.xf, (uint64_t)f.GT, (uint64_t)f.DHn, 0, 0, 0, 0};
  return run_graphed(key, s_, [&](hipStream_t s) -> int {
  ZTRY(pack(f.pb, w_hh, f.nTH, f.KB3H, 2, 3 * H, H, H, 0, H, 0, s));      // V[U][k] = W_hh[k][U]
  ZTRY(k_fill(f.DIxf, 2 * f.KB3H * XB + 2 * f.KBH * XB, 0.f, s));
  ZTRY(t_window.begin(s));
  for (int t = L - 1; t >= 0; --t) {
    ZTRY(t_window.tick(s, 3));
    StageArgs a = sg_args(B, H, f);
    float *di_in = f.DIxf + (long)((t + 1) & 1) * f.KB3H * XB, *di_out = f.DIxf + (long)(t & 1) * f.KB3H * XB;
    float *dh_in = f.DHxf + (long)((t + 1) & 1) * f.KBH * XB, *dh_out = f.DHxf + (long)(t & 1) * f.KBH * XB;
    if (t < L - 1) {     // dh_t = W_hh^T dhh_{t+1} + dh_{t+1} z_{t+1}
      a.g[0].seg[0] = subseg(f.pb, di_in, 0, 2 * f.KBH, f.KB3H, 0);
      a.g[0].seg[1] = subseg(f.pb, dh_in, 2 * f.KBH, f.KBH, f.KB3H, 0);
      a.g[0].nseg = 2;
    }
    a.g[0].tiles = f.nTH; a.g[0].epi = EPI_GRU_BWD;
    a.g[0].p0 = f.GT + 4 * t * sH; a.g[0].p4 = Hs + t * sH;
    a.g[0].o0 = dhc; a.g[0].o1 = DI + (long)t * 3 * sH; a.g[0].o2 = f.DHn + t * sH; a.g[0].o3 = di_out; a.g[0].o4 = dh_out;
    ZTRY(launch_stage(a, s));
  }
  return 0;
  });
}
