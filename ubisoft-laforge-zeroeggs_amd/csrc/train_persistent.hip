// Weight-stationary persistent FORWARD rollout of the training step (batch <= 64): the 255 decoder steps of a window as
// ONE launch.  Same idea as decode_persistent.hip -- the 75.7 MB of per-step weights fit in the register files of the
// 256 CUs, so nothing is re-streamed per step -- but with batch 32 the products are MFMA tiles and the exchanged vectors
// are 128 KB, so the details differ:
//   * workgroup c (512 threads, one per CU) owns hidden units 4c..4c+3 of both GRU layers as ONE 16-row MFMA tile of
//     virtual rows (r, z, n_input, n_hidden) x 4 units (the n gate needs its input and hidden sums apart:
//     n = tanh(W_in x + b_in + r (W_hn h + b_hn)), nn.GRU), and one 16-row tile of the output stage: rows 4c..4c+3 of the
//     folded layer0 (M = W0[:, :PO] diag(sigma_o/sigma_i) W2, decoder_fast.hip), output rows c, c+256, ... of layer2 and
//     layer2's six root rows (the root integration is evaluated redundantly in every workgroup);
//   * the 8 waves split the contraction; a wave keeps its k-blocks of the two GRU tiles in registers (26 + 16 float4 in
//     MFMA A-fragment order, packed once per optimizer step by tp_pack_k) and of the output tile in LDS;
//   * activations travel in B-fragment order through WRITE-ONCE, time-major operand buffers (one contiguous operand per
//     phase and step: [hid_t | gaze, speech, style of x_t | h0_{t-1} | h1_{t-1}] -- the pose columns of x_t are folded onto
//     h1_{t-1}, see TKC below --, [h0_t | h1_{t-1}], [h1_t | cond_{t+1}]), published with 16-byte write-through stores;
//     because no address is ever rewritten inside a rollout, a consumer needs no cache invalidation: it waits until every
//     workgroup's arrival slot has reached the phase and then simply loads lines nobody has cached yet;
//   * the part of every contraction that does not depend on the preceding phase runs in the window BEFORE the hand-off
//     wait (batch <= 32: spread evenly over the three windows of a step), which hides most of the hand-off latency;
//   * canonical copies (Gin, H0, H1, saved gates, pose / root outputs) are written exactly where the stage kernels write
//     them, so the BPTT sweep and the weight-gradient GEMMs are unchanged.
// Every wait is bounded; on give-up the error word is set and the host redoes the rollout with the stage kernels.
#include "tp_common.h"
#include "dec_prologue.h"
#include "gemm.h"
#include "kernels.h"

int g_train_persistent = 1;      // zeggs_set_option("train_persistent", 0/1)
int g_tp_tiles4 = 1;             // zeggs_set_option("tp_tiles4", 0/1): the GRU phases of the training rollout on 4-row tiles (even batch-tile counts)
static int g_tp_ok = -1;

using namespace zeggs_tp;
namespace {

// Arrival slots: workgroup c publishes "I have finished phase instance p" by storing p + 1 into slot[c] (a write-through
// store, no read-modify-write to serialise); a consumer's wave 0 loads all 256 slots with one 16-byte load per lane and
// waits until every slot has reached p + 1.  Epochs are monotonic and a workgroup can run at most one phase ahead of the
// slowest one, so one 1 KB array serves every phase.  Returns false on give-up.
// `mine` (per lane): this lane's four slots = the four workgroups that produce k-block `lane` of every exchanged vector (workgroup c
// owns hidden units 4c .. 4c+3 = a quarter of block c / 4) matter to the calling wave; a wave waits for the producers of ITS
// k-blocks only (wave + 8 j: the lanes with lane % 8 == wave), the eight waves of a workgroup together for everybody.
__device__ __forceinline__ bool tp_wait(const unsigned* slots, unsigned expect, unsigned limit, bool mine = true, unsigned nap = 0) {
  const int lane = threadIdx.x & 63;
  const gu64t* q = (const gu64t*)(slots + 4 * lane);
  for (unsigned spins = 0;; ++spins) {
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool ok = !mine || ((unsigned)a >= expect && (unsigned)(a >> 32) >= expect && (unsigned)b >= expect && (unsigned)(b >> 32) >= expect);
    if (__all(ok)) return true;
    if (spins >= limit) return false;
    for (unsigned i = 0; i < nap; ++i) __builtin_amdgcn_s_sleep(1);
  }
}
// Two samples of the slots in flight, half a round trip apart (option "poll_stagger" = that half in s_sleep units, 0 = off): a
// producer's flag is seen by the first sample issued after it landed, i.e. after a quarter of a round trip on average instead of
// half of one (the round trip of a load that misses every cache is ~0.9 us: the dominant term of a hand-off).
__device__ __forceinline__ bool tp_wait2(const unsigned* slots, unsigned expect, unsigned limit, unsigned stagger) {
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

// products of one part of a phase: blocks j = 0..NJ-1 of this wave are k-blocks kb0 + 8 j (< hi), their weights wr[OFF + j]
// (registers) or wl[(OFF + j) * 64] (LDS).  Two blocks per group, the next group's activation loads in flight while this
// one feeds the matrix cores.  Blocks past `hi` are clamped: their weights are zero (tp_pack_k).
template <int NB, int NJT, int OFF, int NJ, bool WLDS>
__device__ __forceinline__ void tp_mma(const f4 (&wr)[NJT], const f4* wl, const f4* __restrict__ xb, int kb0, int hi,
                                       f4 (&acc)[NB]) {
  if constexpr (NJ <= 0) return;
  constexpr int GU = NB >= 3 ? 1 : 2;                 // k-blocks per group: GU * NB float4 of activations per buffer
  constexpr int NG = (NJ + GU - 1) / GU;
  // the block offsets are cheap scalar arithmetic; hidden from the optimiser's loop-invariant code motion, which otherwise keeps
  // one per block of every part live across the whole time loop and spills registers for it (train_bwd_persistent.hip)
  asm volatile("" : "+s"(kb0));
  f4 xa[GU][NB], xq[GU][NB];
  auto load = [&](f4 (&x)[GU][NB], int g) {
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      int kb = kb0 + 8 * (GU * g + u);
      kb = kb < hi ? kb : hi - 1;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) x[u][nb] = xb[((long)kb * NB + nb) * 64];
    }
  };
  auto comp = [&](const f4 (&x)[GU][NB], int g) {
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int i = GU * g + u;
      if (i < NJ) {
        const f4 wv = WLDS ? wl[(OFF + i) * 64] : wr[OFF + i < NJT ? OFF + i : 0];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c], x[u][nb][c], acc[nb], 0, 0, 0);
      }
    }
  };
  // the scheduling fences keep the next group's loads AHEAD of this group's products (left alone, the compiler sinks
  // every load next to its first use to shorten live ranges, which exposes the whole load latency per block)
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
}

#ifndef ZEGGS_T4_TL0
#define ZEGGS_T4_TL0 8      // LDS-parked k-blocks of GRU layer 0 in the 4-row form (B <= 32; 4 / 0 parked blocks or 3 / 4 blocks per group spill 6 .. 24 registers)
#endif
#ifndef ZEGGS_T4_GU
#define ZEGGS_T4_GU 2       // k-blocks per operand-prefetch group in the 4-row form (B <= 32; 4 / 0 parked blocks or 3 / 4 blocks per group spill 6 .. 24 registers)
#endif
// ---- 4-row tiles for the two GRU phases (option "tp_tiles4", batch tiles in pairs: 17..32 and 49..64 rows).  A 16-row tile of
// v_mfma_f32_16x16x4 holds (r, z, n_input, n_hidden) x 4 units, and every k-block multiplies one all-zero n row group (input-side
// blocks have no hidden-side n weights and vice versa): a quarter of the GRU phases' matrix-core time.  v_mfma_f32_4x4x1 with
// cbsz = 3 is a [4 rows x 2 k x 32 batch] product per instruction (train_bwd_persistent.hip): three row groups (r, z, n of the 4
// units) per k-block, 24 x 8 cycles instead of 8 x 32 per 32 batch rows, one VGPR per [4 rows x 16 k] weight tile (3 per block
// instead of 4).  The n group of a block accumulates into the input- or the hidden-side n sum, by block.
template <int NT, int NW, int OFF, int NJ, bool WLDS, int PH, int IABS>
__device__ __forceinline__ void tp_mma4(const float (&wq)[NW], const float* wl, const f4* __restrict__ xb, int kb0, int hi,
                                        f4 (&acc)[4][NT]) {
  if constexpr (NJ <= 0) return;
  constexpr int GU = NT >= 2 ? 1 : ZEGGS_T4_GU;
  constexpr int NG = (NJ + GU - 1) / GU;
  asm volatile("" : "+s"(kb0));
  f4 xa[GU][2][NT], xq[GU][2][NT];
  auto load = [&](f4 (&x)[GU][2][NT], int g) {
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      int kb = kb0 + 8 * (GU * g + u);
      kb = kb < hi ? kb : hi - 1;
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) x[u][q][nt] = xb[(((long)kb * NT + nt) * 2 + q) * 64];
    }
  };
  auto comp = [&](const f4 (&x)[GU][2][NT], int g) {
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int i = GU * g + u;
      if (i < NJ) {
        const int nd = tp4_hidden_side(PH, IABS + i) ? 3 : 2;
        const int wi = 3 * (OFF + i);
        const float w0 = WLDS ? wl[(wi + 0) * 64] : wq[wi + 0 < NW ? wi + 0 : 0];
        const float w1 = WLDS ? wl[(wi + 1) * 64] : wq[wi + 1 < NW ? wi + 1 : 0];
        const float w2 = WLDS ? wl[(wi + 2) * 64] : wq[wi + 2 < NW ? wi + 2 : 0];
#define TP4_STEP(A)                                                                                    \
  _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                                  \
    const float xv = x[u][(A) >> 2][nt][(A) & 3];                                                      \
    acc[0][nt] = __builtin_amdgcn_mfma_f32_4x4x1f32(w0, xv, acc[0][nt], 3, A, 0);                      \
    acc[1][nt] = __builtin_amdgcn_mfma_f32_4x4x1f32(w1, xv, acc[1][nt], 3, A, 0);                      \
    acc[nd][nt] = __builtin_amdgcn_mfma_f32_4x4x1f32(w2, xv, acc[nd][nt], 3, A, 0);                    \
  }
        TP4_STEP(0) TP4_STEP(1) TP4_STEP(2) TP4_STEP(3) TP4_STEP(4) TP4_STEP(5) TP4_STEP(6) TP4_STEP(7)
#undef TP4_STEP
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
}

// -DZEGGS_TPTIME: wall-clock (100 MHz) stamps of the phases of the LAST step, workgroups 0 and 255 (tools/tp_time.py)
#ifdef ZEGGS_TPTIME
#define TPT(i)                                                                                              \
  do {                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (t >= T - 4 && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {                \
      unsigned long long* sl_ = (unsigned long long*)(a.err + 32) + ((t - (T - 4)) * 2 + (blockIdx.x != 0)) * 32; \
      sl_[(i)] = wall_clock64();                                                                            \
      if ((i) == 0 || (i) == 15) sl_[16 + ((i) == 15)] = clock64();   /* shader-clock cycles of the step */  \
    }                                                                                                       \
  } while (0)
#else
#define TPT(i)
#endif

#ifndef ZEGGS_TP_W0ARRIVE
#define ZEGGS_TP_W0ARRIVE 0      // (experiment: see arrive_gru)
#endif
#ifndef ZEGGS_TP_WAVEWAIT
#define ZEGGS_TP_WAVEWAIT 0      // (measured: every wave polling for its own producers = 8x the polls: 22.5 -> 24.1 us per step)
#endif
#if ZEGGS_TP_WAVEWAIT
#define TP_FAIL_AT_WAIT
#define TP_FAIL_AT_REDUCE if (fail) break
#else
#define TP_FAIL_AT_WAIT if (fail) break
#define TP_FAIL_AT_REDUCE
#endif
template <bool C, typename A, typename B> struct tp_pick { typedef A type; };
template <typename A, typename B> struct tp_pick<false, A, B> { typedef B type; };
template <int NB, bool T4 = false>
__global__ __launch_bounds__(TTHR, 2) void train_fwd_persistent_k(TArgs a) {
  constexpr int BP = 16 * NB;
  constexpr int NT = NB >= 2 ? NB / 2 : 1;     // 32-row batch tiles of the 4-row form (T4: NB is even)
  typedef typename tp_pick<T4, f4[4][NT], f4[NB]>::type GAcc;      // accumulators of a GRU phase: (r, z, n_in, n_hid) x units, or 16-row tiles
  __shared__ f4 red[8][NB][64];
  __shared__ f4 fin[NB][64];
  __shared__ f4 w3[8 * TJ3 * 64];             // output-stage weights of this workgroup (72 KB)
  constexpr int TL0 = (T4 && NB == 2) ? ZEGGS_T4_TL0 : tl0(NB), TS0 = ts0(NB), TS1 = ts1(NB);
  constexpr bool SPREAD = TS0 > 0;       // old parts spread over all three hand-off windows
  __shared__ f4 w0l[8 * (TL0 > 0 ? TL0 : 1) * 64];   // the first TL0 (old-part) k-blocks of GRU layer 0: relieves the register file
  __shared__ float gsh[BP * 3];               // normalised gaze direction of x_{t+1} per batch row
  __shared__ float cA[4][12];                 // biases of the 4 units: b_ih0, b_hh0, b_ih1, b_hh1 (r, z, n)
  __shared__ f4 ex[BP];                       // epilogue exchange: the 4 units of a batch row -> one 16-byte store
  __shared__ float cG[6];                     // gaze columns of x: in_mean[PO..PO+2], 1 / in_std[PO..PO+2]
  __shared__ float cV[4][3];                  // constant of the folded pose columns of GRU layer 0 (r, z, n), steps t > 1
  __shared__ float cW[4][3][3];               // W_ih0[gate rows of the 4 units][gaze columns]
  __shared__ float cB[16][8];                 // output-stage row constants
  __shared__ int fail;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c = blockIdx.x;
  const ZeggsDecDims& d = a.d;
  const int B = d.B, T = d.T, H = TH, PO = d.PO, GL = a.GL;
  const long sG = (long)B * GL, sH = (long)B * H, XB = 256L * NB;
  // ---------------------------------------------------------------- weights -> registers / LDS (once per rollout)
  f4 wr0[TJ0 - TL0], wr1[TJ1];
  float wq0[3 * (TJ0 - TL0)], wq1[3 * TJ1];      // T4: one register per [4 rows x 16 k] tile, three (r, z, n) per k-block
  float* const w0lf = (float*)w0l;                // T4: the LDS-parked blocks as floats, [wave][3 TL0][64]
  if constexpr (T4) {
    const float* p0 = (const float*)a.PW0 + ((long)(c * 8 + wave) * TJ0) * 3 * 64 + lane;
#pragma unroll
    for (int i = 0; i < 3 * TL0; ++i) w0lf[(wave * 3 * TL0 + i) * 64 + lane] = p0[(long)i * 64];
#pragma unroll
    for (int i = 3 * TL0; i < 3 * TJ0; ++i) wq0[i - 3 * TL0] = p0[(long)i * 64];
    const float* p1 = (const float*)a.PW1 + ((long)(c * 8 + wave) * TJ1) * 3 * 64 + lane;
#pragma unroll
    for (int i = 0; i < 3 * TJ1; ++i) wq1[i] = p1[(long)i * 64];
  } else {
    const f4* p0 = a.PW0 + ((long)(c * 8 + wave) * TJ0) * 64 + lane;
#pragma unroll
    for (int i = 0; i < TL0; ++i) w0l[(wave * TL0 + i) * 64 + lane] = p0[(long)i * 64];
#pragma unroll
    for (int i = TL0; i < TJ0; ++i) wr0[i - TL0] = p0[(long)i * 64];
    const f4* p1 = a.PW1 + ((long)(c * 8 + wave) * TJ1) * 64 + lane;
#pragma unroll
    for (int i = 0; i < TJ1; ++i) wr1[i] = p1[(long)i * 64];
  }
  {
    const f4* p3 = a.PW3 + ((long)(c * 8 + wave) * TJ3) * 64 + lane;
#pragma unroll
    for (int i = 0; i < TJ3; ++i) w3[(wave * TJ3 + i) * 64 + lane] = p3[(long)i * 64];
  }
  if (tid < 4) {
    const int U = 4 * c + tid;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      cA[tid][g] = a.b_ih0[g * H + U]; cA[tid][3 + g] = a.b_hh0[g * H + U];
      cA[tid][6 + g] = a.b_ih1[g * H + U]; cA[tid][9 + g] = a.b_hh1[g * H + U];
      cV[tid][g] = a.cv0[g * H + U];
#pragma unroll
      for (int q = 0; q < 3; ++q) cW[tid][g][q] = a.w_ih0[(long)(g * H + U) * (H + a.XD) + H + PO + q];
    }
    cB[tid][0] = a.cvec[U];
#pragma unroll
    for (int k = 0; k < 3; ++k) cB[tid][1 + k] = a.l0_w[(long)U * a.XD + PO + k];
  } else if (tid < 9) {
    const int col = c + TNCU * (tid - 4);
    const bool v = col < PO;
    cB[tid][0] = v ? a.l2_b[col] : 0.f; cB[tid][1] = v ? a.st.out_std[col] : 0.f; cB[tid][2] = v ? a.st.out_mean[col] : 0.f;
    cB[tid][3] = v ? a.st.in_mean[col] : 0.f; cB[tid][4] = v ? a.st.in_std[col] : 1.f; cB[tid][5] = v ? 1.f : 0.f;
  } else if (tid < 15) {
    cB[tid][0] = a.l2_b[tid - 9]; cB[tid][1] = a.st.out_std[tid - 9]; cB[tid][2] = a.st.out_mean[tid - 9];
  }
  if (tid >= 32 && tid < 38) cG[tid - 32] = tid < 35 ? a.st.in_mean[PO + tid - 32] : 1.f / a.st.in_std[PO + tid - 35];
  if (tid == 0) fail = 0;
  if (tid >= 64 && tid < 64 + 3 * BP) {          // normalised gaze direction of x_1 (canonical row of step 1); later steps: root integration
    const int i = tid - 64, b = i / 3;
    gsh[i] = b < B ? a.Gin[sG + (long)b * GL + H + PO + i % 3] : 0.f;
  }
  // GRU epilogue item of this thread: unit eu, batch row eb; the previous hidden values stay in registers for the rollout
  // (re-derived from an opaque copy of the thread index at the top of every step: the per-thread addresses they feed are not
  //  worth a register pair each for the whole rollout)
  // (thread <-> item mapping = the one of the reduction below: thread nb * 64 + l owns float4 l of batch tile nb, which holds
  //  the four gate sums (r, z, n_input, n_hidden) of unit l >> 4 for batch row 16 nb + (l & 15))
  int eu = (tid & 63) >> 4, eb = (tid >> 6) * 16 + (tid & 15);
  bool gact = tid < 4 * BP && eb < B;
  int EU = 4 * c + (eu & 3);
  float hp0 = 0.f, hp1 = 0.f;
  if (!T4 && gact) { hp0 = a.H0[(long)eb * H + EU]; hp1 = a.H1[(long)eb * H + EU]; }      // state before the first generated frame
  // T4: the gate thread of batch row gb carries all four units (lanes 0..31 of wave nt = batch tile nt)
  int gb = (tid >> 6) * 32 + (tid & 31);
  bool gact4 = T4 && tid < 64 * NT && gb < B;      // both half-waves: lanes < 32 units 0, 1, lanes >= 32 units 2, 3 of row gb
  f4 hq0 = f4{0.f, 0.f, 0.f, 0.f}, hq1 = f4{0.f, 0.f, 0.f, 0.f};
  if (gact4) {      // (elements 0, 1 = this lane's two units)
    const int u0 = (tid & 32) ? 2 : 0;
    hq0[0] = a.H0[(long)gb * H + 4 * c + u0]; hq0[1] = a.H0[(long)gb * H + 4 * c + u0 + 1];
    hq1[0] = a.H1[(long)gb * H + 4 * c + u0]; hq1[1] = a.H1[(long)gb * H + 4 * c + u0 + 1];
  }
  // root thread of batch row rb (the LAST B threads: the first ones carry the GRU items): the root state of its row stays
  // in registers for the rollout
  int rb = TTHR - 1 - tid;
  bool ract = rb < B;
  int tb = tid;                 // opaque per-step copy of the thread index for the store threads (one per batch row)
  Q4 rq_ = Q4{1.f, 0.f, 0.f, 0.f};
  V3 rp_ = v3(0.f, 0.f, 0.f);
  if (ract) {
    const float* rq = a.rrot + (long)rb * T * 4;
    const float* rp = a.rpos + (long)rb * T * 3;
    rq_ = Q4{rq[0], rq[1], rq[2], rq[3]};
    rp_ = v3(rp[0], rp[1], rp[2]);
  }
  __syncthreads();
  const float* finf = (const float*)fin;
  auto FV = [&](int vcol, int b) -> float {
    return finf[((((b >> 4)) * 64 + (((vcol >> 2) << 4) | (b & 15))) << 2) | (vcol & 3)];
  };
  auto reduce = [&](f4 (&acc)[NB]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) red[wave][nb][lane] = acc[nb];
    __syncthreads();
    if (tid < NB * 64) {
      const int nb = tid / 64, l = tid % 64;
      f4 s = red[0][nb][l];
#pragma unroll
      for (int w = 1; w < 8; ++w) s += red[w][nb][l];
      fin[nb][l] = s;
    }
    __syncthreads();
  };
  // GRU phases: the thread that adds up float4 (nb, l) of the eight waves IS the gate thread of that unit / batch row, so the
  // sums go from its registers straight into the gate math (no second barrier, no trip through `fin`)
  auto reduce_gate = [&](f4 (&acc)[NB]) -> f4 {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) red[wave][nb][lane] = acc[nb];
    __syncthreads();
    f4 sfv = f4{0.f, 0.f, 0.f, 0.f};
    if (tid < NB * 64) {
      const int nb = tid >> 6, l = tid & 63;
      sfv = red[0][nb][l];
#pragma unroll
      for (int w = 1; w < 8; ++w) sfv += red[w][nb][l];
    }
    return sfv;
  };
  // T4: the accumulators of a wave hold, per gate, the four units of batch row 4 (blk % 8) + j for k-half blk / 8 (lane = 4 blk + j):
  // add the two k-halves (lanes l, l + 32), lanes < 32 park [gate][batch row] float4s (the four units) in LDS (the storage of
  // `red`), 64 threads per batch tile add up two gates each over the eight waves, the upper half hands its pair down: lanes < 32
  // of wave nt end up with (r, z, n_in, n_hid) x 4 units of batch row 32 nt + lane -- the gate thread, no further exchange
  f4 (*red4)[4][NT][32] = (f4 (*)[4][NT][32])red;
  auto swap_down = [&](f4 v) -> f4 {          // lanes < 32 receive lane + 32's value
    f4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned x = __float_as_uint(v[e]);
      const auto sw = __builtin_amdgcn_permlane32_swap(x, x, false, false);
      o[e] = __uint_as_float(sw[1]);
    }
    return o;
  };
  auto swap_halves = [&](f4 v) -> f4 {        // every lane receives the value of lane ^ 32
    f4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned x = __float_as_uint(v[e]);
      const auto sw = __builtin_amdgcn_permlane32_swap(x, x, false, false);
      o[e] = __uint_as_float(lane < 32 ? sw[1] : sw[0]);
    }
    return o;
  };
  auto reduce_gate4 = [&](f4 (&acc)[4][NT], f4 (&out)[4]) {
#ifdef ZEGGS_TP_NORED      // (timing experiment, results wrong: no cross-wave reduction -- every gate thread takes its own wave's partial sums;
    // what the LDS round trip + barrier + 16 serial reads of the reducing threads cost a GRU phase)
#pragma unroll
    for (int g = 0; g < 4; ++g) out[g] = acc[g][0];
    return;
#endif
    // one v_permlane32_swap folds the k-halves of TWO values: swap(a, b) = ([a_lo | b_lo], [a_hi | b_hi]), whose sum holds the folded a
    // in lanes < 32 and the folded b in lanes >= 32 -- gates (0, 2) and (1, 3) pair up, every lane has something to store
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[g][nt][e]), __float_as_uint(acc[g + 2][nt][e]), false, false);
          v[e] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        red4[wave][lane < 32 ? g : g + 2][nt][lane & 31] = v;
      }
    __syncthreads();
    if (ZEGGS_TP_W0ARRIVE == 2 && wave == 0) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int g = 0; g < 4; ++g) out[g] = f4{0.f, 0.f, 0.f, 0.f};
    if (tid < 64 * NT) {
      const int nt = tid >> 6, h2 = (tid >> 5) & 1, b = tid & 31;
      f4 s0 = red4[0][2 * h2][nt][b], s1 = red4[0][2 * h2 + 1][nt][b];
#pragma unroll
      for (int w = 1; w < 8; ++w) { s0 += red4[w][2 * h2][nt][b]; s1 += red4[w][2 * h2 + 1][nt][b]; }
      // both halves end up with all four gates: the lower lanes do units 0, 1 of the batch row, the upper ones units 2, 3
      const f4 t0 = swap_halves(s0), t1 = swap_halves(s1);
      out[0] = h2 ? t0 : s0; out[1] = h2 ? t1 : s1; out[2] = h2 ? s0 : t0; out[3] = h2 ? s1 : t1;
    }
  };
  auto zero_g = [&](GAcc& g) {
    if constexpr (T4) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) g[q][nt] = f4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) g[nb] = f4{0.f, 0.f, 0.f, 0.f};
    }
  };
  // products of the two GRU phases in either tile form.  I0 = position of the first block in the wave's list (old part first);
  // layer 0's first TL0 blocks live in LDS (LDS0 forms)
#define TP_MMA0(I0, NJ, X, KB0, HI, ACC)                                                                            \
  do {                                                                                                              \
    if constexpr (T4) tp_mma4<NT, 3 * (TJ0 - TL0), (I0) - TL0, NJ, false, 0, I0>(wq0, nullptr, X, KB0, HI, ACC);    \
    else tp_mma<NB, TJ0 - TL0, (I0) - TL0, NJ, false>(wr0, nullptr, X, KB0, HI, ACC);                              \
  } while (0)
#define TP_MMA0_LDS(NJ, X, KB0, HI, ACC)                                                                            \
  do {                                                                                                              \
    if constexpr (T4) tp_mma4<NT, 3 * (TJ0 - TL0), 0, NJ, true, 0, 0>(wq0, w0lf + wave * 3 * TL0 * 64 + lane, X, KB0, HI, ACC); \
    else tp_mma<NB, TJ0 - TL0, 0, NJ, true>(wr0, w0l + wave * TL0 * 64 + lane, X, KB0, HI, ACC);                   \
  } while (0)
#define TP_MMA1(I0, NJ, X, KB0, HI, ACC)                                                                            \
  do {                                                                                                              \
    if constexpr (T4) tp_mma4<NT, 3 * TJ1, I0, NJ, false, 1, I0>(wq1, nullptr, X, KB0, HI, ACC);                    \
    else tp_mma<NB, TJ1, I0, NJ, false>(wr1, nullptr, X, KB0, HI, ACC);                                            \
  } while (0)
#ifdef ZEGGS_TPSTAT
  unsigned long long wsum[3] = {0, 0, 0};      // 100 MHz ticks this workgroup spent polling, per phase kind
#endif
  auto wait_phase = [&](long p) {     // all workgroups have finished phase instance p (p < 0: nothing to wait for)
    if (p >= 0) {
#ifdef ZEGGS_TPSTAT
      const unsigned long long w0 = wall_clock64();
#endif
#if ZEGGS_TP_WAVEWAIT
      // every wave for the producers of its own k-blocks: no barrier, no broadcast; a give-up is noticed by everybody behind the
      // phase's reduction barrier (`fail` is checked there: all waves of a workgroup must meet the same barriers)
      if (!tp_wait(a.cnt, (unsigned)(p + 1), a.spin, (lane & 7) == wave)) fail = 1;
#else
#ifndef ZEGGS_TP_NOPOLL      // (timing experiment, results wrong: nobody waits for anybody -- the step as pure per-CU work)
      if (wave == 0 && !(a.stag ? tp_wait2(a.cnt, (unsigned)(p + 1), a.spin, a.stag) : tp_wait(a.cnt, (unsigned)(p + 1), a.spin, true, a.nap))) fail = 1;
#endif
#endif
#ifdef ZEGGS_TPSTAT
      wsum[(p + 1) % 3] += wall_clock64() - w0;
#endif
    }
#if !ZEGGS_TP_WAVEWAIT
    __syncthreads();
#endif
  };
  auto arrive = [&](long p) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store((gu32*)(a.cnt + c), (unsigned)(p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // GRU phases in the 4-row form with one batch tile: wave 0 is the only publisher.  Experiment (-DZEGGS_TP_W0ARRIVE=1 / 2): it raises
  // the flag alone, without the workgroup barrier, so that the other seven waves run the next phase's old-operand products beside
  // its reduction / gate math / publishes (2: wave 0 at raised priority meanwhile).  Round 3 measured form 1 at +0.5 us per step.
  auto arrive_gru = [&](long p) {
    if constexpr (ZEGGS_TP_W0ARRIVE != 0 && T4 && NT == 1) {
      if (wave == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) __hip_atomic_store((gu32*)(a.cnt + c), (unsigned)(p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ZEGGS_TP_W0ARRIVE == 2) __builtin_amdgcn_s_setprio(0);
      }
    } else arrive(p);
  };

  GAcc acc1, acc2;              // accumulators of GRU layer 0 / 1: started in the windows of earlier phases
  zero_g(acc1);
  if constexpr (SPREAD) {     // step 1 has no previous output stage to hide these behind
    const f4* x01 = (const f4*)(a.G0 + (long)a.KB0 * XB) + lane;
    TP_MMA0_LDS(TL0, x01, TFR0 + wave, a.KB0, acc1);
    TP_MMA0(TL0, TS0 - TL0, x01, TFR0 + wave + 8 * TL0, a.KB0, acc1);
  }
  for (int t = 1; t < T; ++t) {
    {
      int tx = tid;
      asm volatile("" : "+v"(tx));
      eu = (tx & 63) >> 4; eb = (tx >> 6) * 16 + (tx & 15);
      gact = tx < 4 * BP && eb < B;
      EU = 4 * c + (eu & 3);
      rb = TTHR - 1 - tx;
      ract = rb < B;
      tb = tx;
      gb = (tx >> 6) * 32 + (tx & 31);
      gact4 = T4 && tx < 64 * NT && gb < B;
    }
    const bool next = t + 1 < T;
    const long p1 = 3L * (t - 1), p2 = p1 + 1, p3 = p1 + 2;
    f4 acc[NB];
    GAcc accg;                  // (non-SPREAD variants: the GRU phases' accumulator)
    // ================================================================ GRU layer 0 : [hid_t | x_t | h0_{t-1}]
    // The old parts of the three phases (18 blocks per wave) are spread over the three hand-off windows, ~6 blocks each: a
    // hand-off (store drain, flag, poll) takes about as long as 7 blocks of products.
    TPT(0);
    if constexpr (SPREAD) {
      const f4* x0 = (const f4*)(a.G0 + (long)t * a.KB0 * XB) + lane;
      const f4* x1 = (const f4*)(a.G1 + (long)t * 128 * XB) + lane;
      // window: the rest of this phase's old part, h1_{t-1} through the fold (cond and h0_{t-1} ran in the window of the previous
      // output stage)
#ifndef ZEGGS_TP_NOWIN      // (timing experiment: the hand-off latency with empty windows; results are wrong)
      TP_MMA0(TS0, TNO0 - TS0, x0, TFR0 + wave + 8 * TS0, a.KB0, acc1);
#endif
      zero_g(acc2);
      if constexpr (TS1 > 0) TP_MMA1(0, TS1, x1, 64 + wave, 128, acc2);
      wait_phase(p1 - 1);
      TP_FAIL_AT_WAIT;
      TPT(1);
#ifdef ZEGGS_TP_L2HIT      // (timing experiment, results wrong: the "fresh" blocks are read from the PREVIOUS step's operand -- lines this XCD's L2
      // already holds -- to measure what the first touch of freshly published data costs a phase: the upper bound of ANY scheme
      // that would place the hand-off data in the consumer's L2 ahead of the wait; HISTORY round 5)
      TP_MMA0(TNO0, TNF0, (t > 1 ? x0 - (long)a.KB0 * XB / 4 : x0), wave, TFRW, acc1);
#else
      TP_MMA0(TNO0, TNF0, x0, wave, TFRW, acc1);  // fresh part
#endif
    } else {
      zero_g(accg);
      const f4* x0 = (const f4*)(a.G0 + (long)t * a.KB0 * XB) + lane;
      // old part (before the hand-off): its first TL0 blocks come from LDS
      TP_MMA0_LDS(TL0, x0, TFR0 + wave, a.KB0, accg);
      TP_MMA0(TL0, TNO0 - TL0, x0, TFR0 + wave + 8 * TL0, a.KB0, accg);
      wait_phase(p1 - 1);
      TP_FAIL_AT_WAIT;
      TPT(1);
      TP_MMA0(TNO0, TNF0, x0, wave, TFRW, accg);  // fresh part
    }
    TPT(2);
    f4 fv0 = f4{0.f, 0.f, 0.f, 0.f}, gq[4];
    if constexpr (T4) reduce_gate4(*(SPREAD ? &acc1 : &accg), gq);
    else fv0 = reduce_gate(*(SPREAD ? &acc1 : &accg));
    TP_FAIL_AT_REDUCE;
    TPT(3);
    if constexpr (T4) {
      if (gact4) {      // two units (u0, u0 + 1) of batch row gb per lane; the lower lane assembles and publishes the four
        const int u0 = (tid & 32) ? 2 : 0;
        float hh[2];
        f4 gt[2];
        const float g0 = gsh[gb * 3], g1 = gsh[gb * 3 + 1], g2 = gsh[gb * 3 + 2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int u = u0 + j;
          const float* k_ = cA[u];
          const float (*wq)[3] = cW[u];
          float xr, xz, xn;
          if (t == 1) { const float* q = a.p1x + (long)gb * 3 * H + 4 * c + u; xr = q[0]; xz = q[H]; xn = q[2 * H]; }
          else { xr = cV[u][0]; xz = cV[u][1]; xn = cV[u][2]; }
          xr += wq[0][0] * g0 + wq[0][1] * g1 + wq[0][2] * g2;
          xz += wq[1][0] * g0 + wq[1][1] * g1 + wq[1][2] * g2;
          xn += wq[2][0] * g0 + wq[2][1] * g1 + wq[2][2] * g2;
          const float r = d_sigmoid(gq[0][u] + k_[0] + xr + k_[3]);
          const float z = d_sigmoid(gq[1][u] + k_[1] + xz + k_[4]);
          const float nh = gq[3][u] + k_[5];
          const float nn = d_tanh(gq[2][u] + k_[2] + xn + r * nh);
          hh[j] = (1.f - z) * nn + z * hq0[j];
          gt[j] = f4{r, z, nn, nh};
        }
        hq0[0] = hh[0]; hq0[1] = hh[1];
        const f4 mine = f4{hh[0], hh[1], 0.f, 0.f}, oth = swap_halves(mine);
        if (!(tid & 32)) {
          const f4 hv = f4{hh[0], hh[1], oth[0], oth[1]};
          const long o = xf4(gb, 4 * c, NT);
          stp4(a.G1 + (long)t * 128 * XB + o, hv);                                                // [h0_t | .] of layer 1
          if (next) stp4(a.G0 + (long)(t + 1) * a.KB0 * XB + (long)TKH0 * XB + o, hv);            // h0 slot of layer 0, step t+1
          *(f4*)(a.H0 + (long)t * sH + (long)gb * H + 4 * c) = hv;
        }
        f4* gts = (f4*)a.GT0 + (long)t * sH + (long)gb * H + 4 * c + u0;
        gts[0] = gt[0]; gts[1] = gt[1];
      }
    } else {
    if (gact) {
      const float* k_ = cA[eu];
      // pose columns of x_t: N0 h1_{t-1} (in the products) + cv0; step 1: the product with the given first pose
      float xr, xz, xn;
      if (t == 1) { const float* q = a.p1x + (long)eb * 3 * H + EU; xr = q[0]; xz = q[H]; xn = q[2 * H]; }
      else { xr = cV[eu][0]; xz = cV[eu][1]; xn = cV[eu][2]; }
      {     // gaze columns of x_t
        const float g0 = gsh[eb * 3], g1 = gsh[eb * 3 + 1], g2 = gsh[eb * 3 + 2];
        const float (*wq)[3] = cW[eu];
        xr += wq[0][0] * g0 + wq[0][1] * g1 + wq[0][2] * g2;
        xz += wq[1][0] * g0 + wq[1][1] * g1 + wq[1][2] * g2;
        xn += wq[2][0] * g0 + wq[2][1] * g1 + wq[2][2] * g2;
      }
      const float r = d_sigmoid(fv0[0] + k_[0] + xr + k_[3]);
      const float z = d_sigmoid(fv0[1] + k_[1] + xz + k_[4]);
      const float nh = fv0[3] + k_[5];
      const float nn = d_tanh(fv0[2] + k_[2] + xn + r * nh);
      const float h = (1.f - z) * nn + z * hp0;
      hp0 = h;
      const long i = (long)t * sH + (long)eb * H + EU;
      ((f4*)a.GT0)[i] = f4{r, z, nn, nh};
      ((float*)&ex[eb])[eu] = h;
    }
    __syncthreads();
    if (tb < B) {       // one thread per batch row publishes the workgroup's four units
      const f4 v = ex[tb];
      const long o = xfi(tb, 4 * c, NB);
      stp4(a.G1 + (long)t * 128 * XB + o, v);                                                // [h0_t | .] of layer 1
      if (next) stp4(a.G0 + (long)(t + 1) * a.KB0 * XB + (long)TKH0 * XB + o, v);            // h0 slot of layer 0, step t+1
      *(f4*)(a.H0 + (long)t * sH + (long)tb * H + 4 * c) = v;
    }
    }
    TPT(4);
    arrive_gru(p1);
    TPT(5);
    // ================================================================ GRU layer 1 : [h0_t | h1_{t-1}]
    if constexpr (SPREAD) {
      const f4* x1 = (const f4*)(a.G1 + (long)t * 128 * XB) + lane;
#ifndef ZEGGS_TP_NOWIN
      TP_MMA1(TS1, TNO1 - TS1, x1, 64 + wave + 8 * TS1, 128, acc2);   // window: rest of the old part
#endif
      wait_phase(p2 - 1);
      TP_FAIL_AT_WAIT;
      TPT(6);
#ifdef ZEGGS_TP_L2HIT
      TP_MMA1(TNO1, TNF1, (t > 1 ? x1 - 128L * XB / 4 : x1), wave, 64, acc2);
#else
      TP_MMA1(TNO1, TNF1, x1, wave, 64, acc2);               // h0_t
#endif
    } else {
      zero_g(accg);
      const f4* x1 = (const f4*)(a.G1 + (long)t * 128 * XB) + lane;
      TP_MMA1(0, TNO1, x1, 64 + wave, 128, accg);             // h1_{t-1}: before the hand-off
      wait_phase(p2 - 1);
      TP_FAIL_AT_WAIT;
      TPT(6);
      TP_MMA1(TNO1, TNF1, x1, wave, 64, accg);                // h0_t
    }
    TPT(7);
    f4 fv1 = f4{0.f, 0.f, 0.f, 0.f};
    if constexpr (T4) reduce_gate4(*(SPREAD ? &acc2 : &accg), gq);
    else fv1 = reduce_gate(*(SPREAD ? &acc2 : &accg));
    TP_FAIL_AT_REDUCE;
    TPT(8);
    if constexpr (T4) {
      if (gact4) {
        const int u0 = (tid & 32) ? 2 : 0;
        float hh[2];
        f4 gt[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int u = u0 + j;
          const float* k_ = cA[u];
          const float r = d_sigmoid(gq[0][u] + k_[6] + k_[9]);
          const float z = d_sigmoid(gq[1][u] + k_[7] + k_[10]);
          const float nh = gq[3][u] + k_[11];
          const float nn = d_tanh(gq[2][u] + k_[8] + r * nh);
          hh[j] = (1.f - z) * nn + z * hq1[j];
          gt[j] = f4{r, z, nn, nh};
        }
        hq1[0] = hh[0]; hq1[1] = hh[1];
        const f4 mine = f4{hh[0], hh[1], 0.f, 0.f}, oth = swap_halves(mine);
        if (!(tid & 32)) {
          const f4 hv = f4{hh[0], hh[1], oth[0], oth[1]};
          const long o = xf4(gb, 4 * c, NT);
          stp4(a.G3 + (long)t * a.KB3 * XB + xfi(gb, 4 * c, NB), hv);                              // [h1_t | .] of the output stage (16-row tiles)
          if (next) {
            stp4(a.G1 + (long)(t + 1) * 128 * XB + 64 * XB + o, hv);                              // [. | h1_t] of t+1
            stp4(a.G0 + (long)(t + 1) * a.KB0 * XB + (long)TKH1 * XB + o, hv);                    // h1 slot of layer 0, step t+1 (fold)
          }
          *(f4*)(a.H1 + (long)t * sH + (long)gb * H + 4 * c) = hv;
        }
        f4* gts = (f4*)a.GT1 + (long)t * sH + (long)gb * H + 4 * c + u0;
        gts[0] = gt[0]; gts[1] = gt[1];
      }
    } else {
    if (gact) {
      const float* k_ = cA[eu];
      const float r = d_sigmoid(fv1[0] + k_[6] + k_[9]);
      const float z = d_sigmoid(fv1[1] + k_[7] + k_[10]);
      const float nh = fv1[3] + k_[11];
      const float nn = d_tanh(fv1[2] + k_[8] + r * nh);
      const float h = (1.f - z) * nn + z * hp1;
      hp1 = h;
      const long i = (long)t * sH + (long)eb * H + EU;
      ((f4*)a.GT1)[i] = f4{r, z, nn, nh};
      ((float*)&ex[eb])[eu] = h;
    }
    __syncthreads();
    if (tb < B) {
      const f4 v = ex[tb];
      const long o = xfi(tb, 4 * c, NB);
      stp4(a.G3 + (long)t * a.KB3 * XB + o, v);                                              // [h1_t | .] of the output stage
      if (next) {
        stp4(a.G1 + (long)(t + 1) * 128 * XB + 64 * XB + o, v);                              // [. | h1_t] of t+1
        stp4(a.G0 + (long)(t + 1) * a.KB0 * XB + (long)TKH1 * XB + o, v);                    // h1 slot of layer 0, step t+1 (fold)
      }
      *(f4*)(a.H1 + (long)t * sH + (long)tb * H + 4 * c) = v;
    }
    }
    TPT(9);
    arrive_gru(p2);
    TPT(10);
    // ================================================================ output stage : [h1_t | cond_{t+1}]
    float gz_[3] = {0.f, 0.f, 0.f};          // gaze target of frame t+1 (an input): in flight under the products
    if (ract && next) { const float* gz = a.gaze + ((long)rb * T + t + 1) * 3; gz_[0] = gz[0]; gz_[1] = gz[1]; gz_[2] = gz[2]; }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = f4{0.f, 0.f, 0.f, 0.f};
    {
      const f4* x3 = (const f4*)(a.G3 + (long)t * a.KB3 * XB) + lane;
      const f4* wl3 = w3 + wave * TJ3 * 64 + lane;
#ifndef ZEGGS_TP_NOWIN
      tp_mma<NB, TJ0 - TL0, 0, TNO3, true>(wr0, wl3, x3, 64 + wave, a.KB3, acc);          // cond_{t+1}: before the hand-off
#endif
      zero_g(acc1);
#ifndef ZEGGS_TP_NOWIN
      if constexpr (SPREAD) {
        if (next) {     // window: cond and h0_t blocks of the NEXT step's GRU layer 0 (h0_t was published two hand-offs ago)
          const f4* x0n = (const f4*)(a.G0 + (long)(t + 1) * a.KB0 * XB) + lane;
          TP_MMA0_LDS(TL0, x0n, TFR0 + wave, a.KB0, acc1);
          TP_MMA0(TL0, TS0 - TL0, x0n, TFR0 + wave + 8 * TL0, a.KB0, acc1);
        }
      }
#endif
      wait_phase(p3 - 1);
      TP_FAIL_AT_WAIT;
      TPT(11);
#ifdef ZEGGS_TP_L2HIT
      tp_mma<NB, TJ0 - TL0, TNO3, TNF3, true>(wr0, wl3, (t > 1 ? x3 - (long)a.KB3 * XB / 4 : x3), wave, 64, acc);
#else
      tp_mma<NB, TJ0 - TL0, TNO3, TNF3, true>(wr0, wl3, x3, wave, 64, acc);               // h1_t
#endif
    }
    TPT(12);
    reduce(acc);
    TP_FAIL_AT_REDUCE;
    TPT(13);
    {
      float* gnext = a.Gin + (long)(t + 1) * sG;                       // canonical [hid | x] row of step t+1
      float* xnext = a.G0 + (long)(t + 1) * a.KB0 * XB;                // its fragment copy
      if (ract) {                 // root integration of batch row rb (ZEGGS/modules.py:139-176), every workgroup
        const int b = rb;
        float p[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) p[q] = (FV(9 + q, b) + cB[9 + q][0]) * cB[9 + q][1] + cB[9 + q][2];
        const Q4 q = rq_;
        const V3 pos = rp_;
        const V3 npos = quat_mul_vec(q, d.dt * v3(p[0], p[1], p[2])) + pos;
        const V3 uu = quat_mul_vec(q, d.dt * v3(p[3], p[4], p[5]));
        const Q4 nq = quat_exp_mul(0.5f * uu, q);
        rq_ = nq; rp_ = npos;
        float genc[3] = {0.f, 0.f, 0.f};
        if (next) {
          const V3 gd = quat_mul_vec(quat_inv(nq), v3(gz_[0], gz_[1], gz_[2]) - npos);
          genc[0] = (gd.x - cG[0]) * cG[3];
          genc[1] = (gd.y - cG[1]) * cG[4];
          genc[2] = (gd.z - cG[2]) * cG[5];
        }
        gsh[b * 3] = genc[0]; gsh[b * 3 + 1] = genc[1]; gsh[b * 3 + 2] = genc[2];
        if (c == 0) {
          float* op = a.rpos + ((long)b * T + t) * 3;
          op[0] = npos.x; op[1] = npos.y; op[2] = npos.z;
          float* oq = a.rrot + ((long)b * T + t) * 4;
          oq[0] = nq.w; oq[1] = nq.x; oq[2] = nq.y; oq[3] = nq.z;
          if (next)
            for (int k = 0; k < 3; ++k) {
              gnext[(long)b * GL + H + PO + k] = genc[k];
              stp(xnext + 64 * XB + xfi(b, k, NB), genc[k]);                 // the gaze block of layer 0's operand
            }
        }
      }
      for (int item = tb; item < 5 * BP; item += TTHR) {        // layer2 rows: pose_t and the pose columns of x_{t+1} (tb: per-step thread index)
        const int vc = 4 + item / BP, b = item % BP;
        if (b < B && cB[vc][5] != 0.f) {
          const float* k_ = cB[vc];
          const int col = c + TNCU * (vc - 4);
          const float pv = (FV(vc, b) + k_[0]) * k_[1] + k_[2];
          a.pose[((long)b * T + t) * PO + col] = pv;
          if (next) {
            const float e = (pv - k_[3]) / k_[4];
            gnext[(long)b * GL + H + col] = e;        // (canonical only: the products take the pose columns through the fold)
          }
        }
      }
      __syncthreads();
      if (next && tb < 4 * BP && tb % BP < B) {                   // folded layer0 rows: hid_{t+1}
        const int vc = tb / BP, b = tb % BP;
        const float* k_ = cB[vc];
        const int col = 4 * c + vc;
        const float val = d_elu(FV(vc, b) + k_[0] + k_[1] * gsh[b * 3] + k_[2] * gsh[b * 3 + 1] + k_[3] * gsh[b * 3 + 2]);
        ((float*)&ex[b])[vc] = val;
        (void)col;
      }
      __syncthreads();
      if (next && tb < B) {
        const f4 v = ex[tb];
        stp4(xnext + (T4 ? xf4(tb, 4 * c, NT) : xfi(tb, 4 * c, NB)), v);
        *(f4*)(gnext + (long)tb * GL + 4 * c) = v;
      }
    }
    TPT(14);
    arrive(p3);
    TPT(15);
  }
#ifdef ZEGGS_TPSTAT
  if (tid == 0) {
    unsigned long long* o = (unsigned long long*)(a.err + 512) + 4 * c;
    for (int i = 0; i < 3; ++i) o[i] = wsum[i];
  }
#endif
  if (fail) {     // a bounded wait gave up: error word, the caller's sticky status, NaN in the last frame of every output row
    if (tid == 0) {
      atomicOr(a.err, 1u);
      if (a.status) atomicOr(a.status, ZEGGS_GAVE_UP_TRAIN_FWD);
    }
    const float qnan = __uint_as_float(0x7fc00000u);
    for (int i = c * TTHR + tid; i < B * PO; i += TNCU * TTHR) a.pose[((long)(i / PO) * T + T - 1) * PO + i % PO] = qnan;
    if (c == 0 && tid < B) { a.rpos[((long)tid * T + T - 1) * 3] = qnan; a.rrot[((long)tid * T + T - 1) * 4] = qnan; }
  }
}

// value of virtual row i, contraction index k of workgroup c's tile for phase ph (0: GRU l0, 1: GRU l1, 3: output stage)
struct TPackArgs {
  int t4;                                    // != 0: PW0 / PW1 as 4-row tiles (three floats per lane and k-block), see tp_mma4
  f4 *PW0, *PW1, *PW3;
  const float *w_ih0, *w_hh0, *w_ih1, *w_hh1, *l2_w, *l0_w, *Mc, *n0;
  int XD, KBX, KBC, KB0, KB3, PO, PI, NC;
};
__device__ __forceinline__ float tp_value(const TPackArgs& p, int ph, int c, int i, int k) {
  const int H = TH;
  if (ph <= 1) {
    const int u = i >> 2, g = i & 3, U = 4 * c + u;          // g: 0 r, 1 z, 2 n (input side), 3 n (hidden side)
    const long row = (long)(g < 2 ? g : 2) * H + U;
    if (ph == 0) {
      const int KIN = H + p.XD;
      if (k < H) return g == 3 ? 0.f : p.w_ih0[row * KIN + k];                                // hid
      if (k < 16 * TFR0) { const int j = k - H; return (g == 3 || j >= 3) ? 0.f : p.w_ih0[row * KIN + H + p.PO + j]; }   // gaze
      if (k < 16 * TKH0) { const int j = k - 16 * TFR0; return (g == 3 || j >= p.NC) ? 0.f : p.w_ih0[row * KIN + H + p.PI + j]; }
      if (k < 16 * TKH1) return g == 2 ? 0.f : p.w_hh0[row * H + (k - 16 * TKH0)];           // h0_{t-1}
      return g == 3 ? 0.f : p.n0[row * H + (k - 16 * TKH1)];                                  // h1_{t-1} through the fold
    }
    if (k < H) return g == 3 ? 0.f : p.w_ih1[row * H + k];
    return g == 2 ? 0.f : p.w_hh1[row * H + (k - H)];
  }
  if (i < 4) {
    const long R = 4 * c + i;
    if (k < H) return p.Mc[R * H + k];
    return (k - H) < p.NC ? p.l0_w[R * p.XD + p.PI + (k - H)] : 0.f;
  }
  if (k >= H) return 0.f;
  if (i < 9) { const int col = c + TNCU * (i - 4); return col < p.PO ? p.l2_w[(long)col * H + k] : 0.f; }
  if (i < 15) return p.l2_w[(long)(i - 9) * H + k];
  return 0.f;
}
__global__ void tp_pack_k(TPackArgs p) {
  // destination order: [workgroup][wave][block i of the wave (old part, then fresh part)][64 lanes]
  const long n0 = (long)TNCU * 8 * TJ0 * 64, n1 = (long)TNCU * 8 * TJ1 * 64, n3 = (long)TNCU * 8 * TJ3 * 64;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n0 + n1 + n3; idx += (long)gridDim.x * blockDim.x) {
    int ph, J; long r = idx; f4* dst;
    if (r < n0) { ph = 0; dst = p.PW0; J = TJ0; }
    else if (r < n0 + n1) { ph = 1; r -= n0; dst = p.PW1; J = TJ1; }
    else { ph = 3; r -= n0 + n1; dst = p.PW3; J = TJ3; }
    const int lane = (int)(r & 63);
    const long cwi = r >> 6;
    const int i = (int)(cwi % J), wave = (int)((cwi / J) & 7), c = (int)(cwi / (8L * J));
    const int kb = ph == 0 ? tp_kb(i, wave, TNO0, TFR0, p.KB0, TFRW)
                 : ph == 1 ? tp_kb(i, wave, TNO1, 64, 128, 64) : tp_kb(i, wave, TNO3, 64, p.KB3, 64);
    f4 v = f4{0.f, 0.f, 0.f, 0.f};
    if (p.t4 && ph <= 1) {
      // lane = 32 * k-half + 4 * a + unit holds, for row group rg (r, z, n), the weight of (unit, gate rg) at k = 16 kb + 8 half + a;
      // the n group is the block's input- or hidden-side n row (the other one is zero for every k of the block)
      if (kb >= 0) {
        const int u = lane & 3, k = 16 * kb + 8 * (lane >> 5) + ((lane >> 2) & 7);
        v[0] = tp_value(p, ph, c, 4 * u + 0, k);
        v[1] = tp_value(p, ph, c, 4 * u + 1, k);
        v[2] = tp_value(p, ph, c, 4 * u + 2, k) + tp_value(p, ph, c, 4 * u + 3, k);
      }
      float* d3 = (float*)dst + (cwi * 3) * 64 + lane;      // [workgroup][wave][block][rg][64]
      d3[0] = v[0]; d3[64] = v[1]; d3[128] = v[2];
      continue;
    }
    const int row = lane & 15, kk = 16 * kb + 4 * (lane >> 4);
    if (kb >= 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = tp_value(p, ph, c, row, kk + q);
    }
    dst[r] = v;
  }
}

// zero k-blocks [kb0, kb0 + nkb) of every step of a time-major operand buffer (the blocks that hold pad columns: everything
// else is written by the rollout or its prologue before it is read, and pad BATCH rows only ever feed pad batch columns)
__global__ void tp_zero_blocks_k(float* G, long KB, long XB, int kb0, int nkb, int t0, int t1) {
  const long per = (long)nkb * XB, n = (long)(t1 - t0) * per;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    G[((long)t0 + i / per) * KB * XB + (long)kb0 * XB + i % per] = 0.f;
}
// canonical [B, ld] (columns 0 .. K) -> fragments at k offset kofs of an operand buffer; the three vectors the rollout starts
// from (hid_1, h0_0, h1_0: K = H each) in one launch
struct TXfrag { float* xf[3]; const float* src[3]; long ld[3]; int kofs[3]; };
__global__ void tp_xfrag_k(TXfrag x, int K, int B, int NB, int t4) {
  const long n = (long)B * K;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < 3 * n; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i / n);
    const long r = i - j * n;
    const int k = (int)(r % K), b = (int)(r / K);
    x.xf[j][t4 ? xf4(b, x.kofs[j] + k, NB / 2) : xfi(b, x.kofs[j] + k, NB)] = x.src[j][(long)b * x.ld[j] + k];
  }
}
// speech / style columns of every step: x part of G0[t] (t >= 1) and the cond part of G3[t] (cond_{t+1}); block `bid` of `nblocks`
__device__ __forceinline__ void tp_cond_body(long bid, long nblocks, const ZeggsDecDims& d, const float* speech, const float* style,
                                             float* G0, float* G3, int KB0, int KB3, int NB, int t4) {
  const int XC = d.SP + d.ST;
  const long XB = 256L * NB, n = (long)(d.T - 1) * d.B * XC;
  for (long i = bid * blockDim.x + threadIdx.x; i < n; i += nblocks * blockDim.x) {
    const int cc = (int)(i % XC);
    const long r = i / XC;
    const int b = (int)(r % d.B), t = 1 + (int)(r / d.B);
    const float v = cc < d.SP ? speech[((long)b * d.T + t) * d.SP + cc] : style[((long)b * d.T + t) * d.ST + (cc - d.SP)];
    G0[(long)t * KB0 * XB + (long)TFR0 * XB + (t4 ? xf4(b, cc, NB / 2) : xfi(b, cc, NB))] = v;      // (G3 keeps the 16-row form)
    if (t >= 2) G3[(long)(t - 1) * KB3 * XB + 64 * XB + xfi(b, cc, NB)] = v;
  }
}
__global__ void tp_cond_k(ZeggsDecDims d, const float* speech, const float* style, float* G0, float* G3, int KB0, int KB3,
                          int NB, int t4) {
  tp_cond_body(blockIdx.x, gridDim.x, d, speech, style, G0, G3, KB0, KB3, NB, t4);
}
// Round 6: everything elementwise in front of the training rollout in ONE launch -- frame 0 + CellStateEncoder input + the pose part
// of x_1 (dec_init: blocks [0, B)), the speech / style columns of every canonical input row (dec_fill_cond: the next nfill blocks)
// and of the rollout's own operand buffers (tp_cond: the rest).  The three write disjoint memory and read inputs only.
struct TProArgs {
  ZeggsDecDims d; ZeggsDecStats st;
  const float *pose0, *rp0, *rr0, *gaze, *speech, *style;
  float *pose, *rpos, *rrot, *cse_in, *gin;
  int GL; long sG;
  float *G0, *G3; int KB0, KB3, NB, t4;
  int nfill, ncond;
};
__global__ __launch_bounds__(256) void tp_prologue_k(TProArgs a) {
  int bx = blockIdx.x;
  if (bx < a.d.B) {
    dec_init_body(bx, a.d, a.st, a.pose0, a.rp0, a.rr0, a.gaze, a.style, a.pose, a.rpos, a.rrot, a.cse_in, a.gin + a.sG, a.GL);
    return;
  }
  bx -= a.d.B;
  if (bx < a.nfill) {
    dec_fill_cond_body(bx, a.nfill, a.d, a.speech, a.style, a.gin, a.GL, 1, a.d.T - 1, a.sG, 0);
    return;
  }
  tp_cond_body(bx - a.nfill, a.ncond, a.d, a.speech, a.style, a.G0, a.G3, a.KB0, a.KB3, a.NB, a.t4);
}

// S[r][c] = W[r * ld + c] * sigma_o[c] / sigma_i[c]  (c < PO; zero padded to POL columns): the pose columns of W_ih0, rescaled from
// normalised-input to raw-output units
__global__ void tp_scale_cols_k(float* S, const float* W, long ld, ZeggsDecStats st, int rows, int PO, int POL) {
  const long n = (long)rows * POL;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % POL);
    const long r = i / POL;
    S[i] = c < PO ? W[r * ld + c] * (st.out_std[c] / st.in_std[c]) : 0.f;
  }
}

}  // namespace

int dec_tp_supported(const ZeggsDecDims& d, const DecWs& w) {
  return !d.film && d.H == TH && d.B <= 64 && d.T >= 4 && d.PI == d.PO + 3 && d.SP + d.ST <= 16 * TKC && w.KBC >= 1 &&
         w.KBC <= 8 * TNO3 && d.PO <= 5 * TNCU && d.PO >= 16 && w.G0xf != nullptr;
}
namespace zeggs_tp {
int tp_dual_supported(int NB);                                   // train_dual.hip: two 16-row chains in one launch (batch 17..32)
void tp_dual_launch(const TArgs& a, hipStream_t s);
}
// weight packs of the GRU phases as 4-row tiles (the dual-chain kernel uses them too) / activation operands in the 4-row form's
// 32-row layout (the dual-chain kernel reads 16-row tiles: tile = chain)
static bool tp_pack_t4(const DecWs& w) { return tp_dual_supported(w.NB) || (g_tp_tiles4 && w.NB % 2 == 0); }
static bool tp_use_t4(const DecWs& w) { return !tp_dual_supported(w.NB) && g_tp_tiles4 && w.NB % 2 == 0; }
int dec_tp_state() { return g_tp_ok; }
void dec_tp_set_state(int v) { g_tp_ok = v; }

// once per optimizer step: the per-workgroup fragment packs (needs Mc / cvec: dec_fast_merge_prep)
int dec_tp_pack(const ZeggsDecDims& d, const ZeggsDecParams* P, const ZeggsDecStats* st, DecWs& w, hipStream_t s) {
  const int H = d.H, KIN = H + w.XD;
  // the fold of GRU layer 0's pose columns: N0 = W_ih0[:, pose] diag(sigma_o/sigma_i) W2 [3H, H], cv0 = W_ih0[:, pose] v [3H]
  // (v = (b2 sigma_o + mu_o - mu_i) / sigma_i: dec_fast_merge_prep)
  {
    const long n = 3L * H * w.POL;
    hipLaunchKernelGGL(tp_scale_cols_k, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, s, w.tp_n0s,
                       P->w_ih0 + H, (long)KIN, *st, 3 * H, d.PO, w.POL);
    ZLAUNCH_CHECK("tp_scale_cols");
    ZTRY(gemm_nn(w.tp_n0s, w.POL, P->l2_w, H, w.tp_n0, H, 3 * H, d.PO, H, 0.f, s));
    ZTRY(gemm_nt(w.vvec, w.POL, P->w_ih0 + H, KIN, w.tp_cv0, 3 * H, nullptr, 1, 3 * H, d.PO, ACT_NONE, 0.f, s));
  }
  TPackArgs p{tp_pack_t4(w) ? 1 : 0, (f4*)w.tp_w0, (f4*)w.tp_w1, (f4*)w.tp_w3, P->w_ih0, P->w_hh0, P->w_ih1, P->w_hh1, P->l2_w, P->l0_w, w.Mc, w.tp_n0,
              w.XD, w.KBX, w.KBC, TKB0, 64 + w.KBC, d.PO, d.PI, d.SP + d.ST};
  hipLaunchKernelGGL(tp_pack_k, dim3(8192), dim3(256), 0, s, p);
  ZLAUNCH_CHECK("tp_pack");
  return 0;
}

// the rollout; H0 / H1 slot 0, Gin slot 1 (hid_1 | x_1) and frame 0 of pose / rpos / rrot are prepared by the caller
// operand buffers: zero what is read but never written (pad rows / pad columns must be finite: only the blocks with pad
// columns -- the gaze + speech / style blocks of G0, the cond blocks of G3, the h1 slot of step 1), and the arrival slots + error
// word.  Depends on nothing but the dimensions: zeggs_decoder_prepare runs it ahead of the forward.
int dec_tp_zero(const ZeggsDecDims& d, DecWs& w, hipStream_t s) {
  const int KB0 = TKB0, KB3 = 64 + w.KBC;
  const long XB = 256L * w.NB;
  hipLaunchKernelGGL(tp_zero_blocks_k, dim3(1024), dim3(256), 0, s, w.G0xf, (long)KB0, XB, 64, 1 + TKC, 1, d.T);
  hipLaunchKernelGGL(tp_zero_blocks_k, dim3(256), dim3(256), 0, s, w.G0xf, (long)KB0, XB, TKH1, 64, 1, 2);
  hipLaunchKernelGGL(tp_zero_blocks_k, dim3(1024), dim3(256), 0, s, w.G3xf, (long)KB3, XB, 64, KB3 - 64, 1, d.T);
  ZTRY(k_fill((float*)w.tp_cnt, 2048, 0.f, s));
  return 0;
}
// the elementwise prologue as one launch (tp_prologue_k); the caller then runs the CellStateEncoder, hid_1 and the step-1 pose product
// (dec_tp_p1x_item) and calls dec_tp_run(..., prologue_done = true)
int dec_tp_prologue(const ZeggsDecDims& d, const ZeggsDecStats* st, DecWs& w, const float* pose0, const float* rpos0,
                    const float* rrot0, const float* gaze, const float* speech, const float* style, float* pose, float* rpos,
                    float* rrot, hipStream_t s, bool zeroed) {
  if (!zeroed) ZTRY(dec_tp_zero(d, w, s));
  TProArgs a;
  memset(&a, 0, sizeof(a));
  a.d = d; a.st = *st; a.pose0 = pose0; a.rp0 = rpos0; a.rr0 = rrot0; a.gaze = gaze; a.speech = speech; a.style = style;
  a.pose = pose; a.rpos = rpos; a.rrot = rrot; a.cse_in = w.cse_in; a.gin = w.Gin; a.GL = w.GL; a.sG = (long)d.B * w.GL;
  a.G0 = w.G0xf; a.G3 = w.G3xf; a.KB0 = TKB0; a.KB3 = 64 + w.KBC; a.NB = w.NB; a.t4 = tp_use_t4(w) ? 1 : 0;
  const long nf = ((long)(d.T - 1) * d.B * (d.SP + d.ST) + 255) / 256;
  a.nfill = (int)(nf > 2048 ? 2048 : (nf < 1 ? 1 : nf));
  a.ncond = 1024;
  hipLaunchKernelGGL(tp_prologue_k, dim3(d.B + a.nfill + a.ncond), dim3(256), 0, s, a);
  ZLAUNCH_CHECK("tp_prologue");
  return 0;
}
// the step-1 pose product p1x[b][3H] = x_1[b][pose] W_ih0[:, pose]^T as an item of a multi-product launch
GemmNtItem dec_tp_p1x_item(const ZeggsDecDims& d, const ZeggsDecParams* P, const DecWs& w) {
  const float* gin1 = w.Gin + (long)d.B * w.GL;
  return GemmNtItem{gin1 + d.H, (long)w.GL, P->w_ih0 + d.H, (long)(d.H + w.XD), w.tp_p1x, 3L * d.H, nullptr, 3 * d.H, d.PO, ACT_NONE};
}
int dec_tp_run(const ZeggsDecDims& d, const ZeggsDecParams* P, const ZeggsDecStats* st, DecWs& w, const float* gaze,
               const float* speech, const float* style, float* pose, float* rpos, float* rrot, hipStream_t s, bool zeroed,
               unsigned* status, bool prologue_done) {
  const int B = d.B, H = d.H, NB = w.NB, KB0 = TKB0, KB3 = 64 + w.KBC;
  const long XB = 256L * NB, sG = (long)B * w.GL;
  int dev = 0, ncu = 0;
  ZCHECK(hipGetDevice(&dev) == hipSuccess, "hipGetDevice failed");
  ZCHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess, "device query failed");
  ZCHECK(ncu >= TNCU, "persistent training rollout needs %d CUs (device has %d)", TNCU, ncu);
  // operand buffers: zero (pad rows / pad columns must be finite), then the inputs that do not depend on the rollout
  // (only the blocks with pad columns: the gaze + speech / style blocks of G0, the cond blocks of G3, the h1 slot of step 1)
  if (!zeroed && !prologue_done) ZTRY(dec_tp_zero(d, w, s));
  const int t4 = tp_use_t4(w) ? 1 : 0;
  if (!prologue_done) hipLaunchKernelGGL(tp_cond_k, dim3(1024), dim3(256), 0, s, d, speech, style, w.G0xf, w.G3xf, KB0, KB3, NB, t4);
  const float* gin1 = w.Gin + sG;
  {   // hid_1, h0_0 -> operand of GRU layer 0, step 1 (its h1 slot stays zero: the pose columns of x_1 are the given first pose; its
      // gaze block is not an operand any more: the gate threads read the gaze direction of x_1 from the canonical row); h1_0 -> layer 1
    TXfrag x;
    x.xf[0] = w.G0xf + (long)KB0 * XB; x.src[0] = gin1; x.ld[0] = w.GL; x.kofs[0] = 0;
    x.xf[1] = w.G0xf + (long)KB0 * XB; x.src[1] = w.H0; x.ld[1] = H;    x.kofs[1] = 16 * TKH0;
    x.xf[2] = w.G1xf + 128 * XB;       x.src[2] = w.H1; x.ld[2] = H;    x.kofs[2] = 16 * 64;
    const long g = (3L * B * H + 255) / 256;
    hipLaunchKernelGGL(tp_xfrag_k, dim3((unsigned)(g > 1024 ? 1024 : g)), dim3(256), 0, s, x, H, B, NB, t4);
  }
  // ... whose product with W_ih0 is one small GEMM: p1x[b][3H] = x_1[b][pose] W_ih0[:, pose]^T
  if (!prologue_done) ZTRY(gemm_nt(gin1 + H, w.GL, P->w_ih0 + H, H + w.XD, w.tp_p1x, 3 * H, nullptr, B, 3 * H, d.PO, ACT_NONE, 0.f, s));
  ZLAUNCH_CHECK("tp_prologue");
  TArgs a;
  memset(&a, 0, sizeof(a));
  a.d = d; a.st = *st; a.XD = w.XD; a.GL = w.GL; a.KBX = w.KBX; a.KBC = w.KBC; a.KB0 = KB0; a.KB3 = KB3; a.POL = w.POL;
  a.PW0 = (const f4*)w.tp_w0; a.PW1 = (const f4*)w.tp_w1; a.PW3 = (const f4*)w.tp_w3;
  a.G0 = w.G0xf; a.G1 = w.G1xf; a.G3 = w.G3xf;
  a.Gin = w.Gin; a.H0 = w.H0; a.H1 = w.H1; a.GT0 = w.GT0; a.GT1 = w.GT1;
  a.b_ih0 = P->b_ih0; a.b_hh0 = P->b_hh0; a.b_ih1 = P->b_ih1; a.b_hh1 = P->b_hh1; a.cvec = w.cvec; a.l0_w = P->l0_w;
  a.l2_b = P->l2_b; a.w_ih0 = P->w_ih0; a.cv0 = w.tp_cv0; a.p1x = w.tp_p1x; a.gaze = gaze; a.pose = pose; a.rpos = rpos; a.rrot = rrot;
  a.cnt = w.tp_cnt; a.err = w.tp_cnt + TRING * TSH * TSTR;
  a.status = status; a.spin = (unsigned)g_persistent_spin; a.nap = (unsigned)g_poll_sleep; a.stag = (unsigned)g_poll_stagger;
  if (tp_dual_supported(NB)) {
    tp_dual_launch(a, s);
    ZLAUNCH_CHECK("train_fwd_dual");
    return 0;
  }
  switch (NB) {
    case 1: hipLaunchKernelGGL((train_fwd_persistent_k<1>), dim3(TNCU), dim3(TTHR), 0, s, a); break;
    case 2:
      if (t4) hipLaunchKernelGGL((train_fwd_persistent_k<2, true>), dim3(TNCU), dim3(TTHR), 0, s, a);
      else hipLaunchKernelGGL((train_fwd_persistent_k<2>), dim3(TNCU), dim3(TTHR), 0, s, a);
      break;
    case 3: hipLaunchKernelGGL((train_fwd_persistent_k<3>), dim3(TNCU), dim3(TTHR), 0, s, a); break;
    default:
      if (t4) hipLaunchKernelGGL((train_fwd_persistent_k<4, true>), dim3(TNCU), dim3(TTHR), 0, s, a);
      else hipLaunchKernelGGL((train_fwd_persistent_k<4>), dim3(TNCU), dim3(TTHR), 0, s, a);
      break;
  }
  ZLAUNCH_CHECK("train_fwd_persistent");
  return 0;
}
extern "C" int zeggs_tp_stamps(const ZeggsDecDims* dp, void* ws, size_t ws_bytes, unsigned long long* out /* [4][2][32] */) {
  Arena a(ws, ws_bytes);
  DecWs w = carve_dec(*dp, 1, a);
  ZCHECK(a.ok() && w.tp_cnt, "tp_stamps: workspace");
  ZCHECK(hipMemcpy(out, w.tp_cnt + TRING * TSH * TSTR + 32, 4 * 2 * 32 * 8, hipMemcpyDeviceToHost) == hipSuccess, "copy");
  return 0;
}
// -DZEGGS_TPSTAT builds: 100 MHz ticks every workgroup spent polling for the hand-off into phase 1..3, summed over the rollout
extern "C" int zeggs_tp_waits(const ZeggsDecDims* dp, void* ws, size_t ws_bytes, unsigned long long* out /* host [256][4] */) {
  Arena a(ws, ws_bytes);
  DecWs w = carve_dec(*dp, 1, a);
  ZCHECK(a.ok() && w.tp_cnt, "tp_waits: workspace");
  ZCHECK(hipMemcpy(out, w.tp_cnt + TRING * TSH * TSTR + 512, 256 * 4 * 8, hipMemcpyDeviceToHost) == hipSuccess, "copy");
  return 0;
}
int dec_tp_errptr(const DecWs& w, unsigned** out) {
  *out = w.tp_cnt + TRING * TSH * TSTR;
  return 0;
}
int dec_tp_errors(const DecWs& w, unsigned* out) {
  ZCHECK(hipMemcpy(out, w.tp_cnt + TRING * TSH * TSTR, sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess,
         "persistent training rollout: error word copy failed");
  return 0;
}
