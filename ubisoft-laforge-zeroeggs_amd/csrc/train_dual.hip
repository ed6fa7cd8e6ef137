// Weight-stationary persistent FORWARD rollout of the training step as TWO INDEPENDENT DEPENDENCY CHAINS in one launch
// (batch 17..32: chain A = batch rows 0..15, chain B = rows 16..31; option "tp_dual").  Same weights, same packs, same
// write-once operands, same canonical saves as train_persistent.hip -- what changes is who waits for whom.
//
// Why.  The single-chain sweep is serial ACROSS THE CHIP per phase: fresh products -> cross-wave reduction -> gate math on one
// wave -> publishes -> store drain -> flag -> poll, during most of which the matrix pipe of every CU idles (10.9 us of matrix-core
// issue inside a 20.2 us step; profiles/r05_handoff_l2hit_bound.txt).  A decoder step of batch row b depends on row b only, so two
// half-batches are two recurrences that never meet: while chain A's phase sits in its epilogue and its grid hand-off, chain B's
// products own the matrix pipe, and vice versa.  Reference: ZEGGS/modules.py:100-151 (the per-frame loop of Decoder.forward).
//
// How.
//   * Every wave keeps its k-blocks of the weight tiles exactly as in the 4-row form of train_persistent.hip (tp_pack_k, t4 packs).
//     v_mfma_f32_4x4x1 with cbsz = 2 is a [4 rows x 4 k x 16 batch columns] product per instruction: ONE weight register serves both
//     chains, the operand of a chain is the 16-row tile of the v_mfma_f32_16x16x4 B layout (xfi with NB = 2, tile = chain) -- lane l
//     reads k = 4 (l / 16) + abid of batch row l % 16 -- and the result registers hold the partial sums of k-quarter l / 16
//     (tools/dual_lane_probe.hip pins these lane semantics on the hardware).
//   * A wave walks the slots A0 B0 A1 B1 A2 B2 of a step in order (0 / 1: GRU layers, 2: output stage).  A slot is: wait for the
//     chain's previous phase, its eight fresh k-blocks into the ring, ALL products that consume that vector (this step's and the
//     next step's: every published vector is loaded once -- half the operand stream of the single-chain sweep), fold of the
//     k-quarters (v_permlane32_swap + v_permlane16_swap), partial sums to LDS, one LDS counter increment.  NO workgroup barrier
//     anywhere in the time loop.
//   * The epilogue of slot (chain X, phase L) belongs to ONE wave, 2 L + X (waves 6, 7 have none): it waits on the LDS counter for the
//     eight partial sums, adds them, does the gate math / root integration for the 16 rows of its chain (64 lanes = 16 rows x 4
//     units), publishes with 16-byte write-through stores, drains and raises the chain's arrival flag -- while the other seven waves
//     are already in the next slot, i.e. in the OTHER chain's products.  The recurrent state (previous hidden values, root
//     transform) lives in the registers of the wave that owns the slot.
//   * Hand-off: per chain, 8 x 32 arrival flags ordered by producer CLASS (workgroup c produces a quarter of k-block c / 4, which
//     wave (c / 4) % 8 of every workgroup consumes): a wave polls the one 128-byte line of ITS producers, with the poll issued
//     before the slot's old-operand products and looked at after them.  What a wave on the same CU must not overtake (the partial-sum
//     buffer of the chain being read by its epilogue wave) is guarded by an LDS epoch word.
// Every wait is bounded; give-up protocol as in train_persistent.hip.
#include "tp_common.h"
#include "kernels.h"

using namespace zeggs_tp;

int g_tp_dual = 0;      // zeggs_set_option("tp_dual", 0/1)

namespace {

constexpr int DL0 = 8;      // old-part k-blocks of GRU layer 0 parked in LDS (as the 4-row form of train_persistent.hip)
constexpr int DR = 8;       // operand ring of a wave: float4 (one k-block of one chain) slots in flight
constexpr long DXB = 512;   // floats per k-block of an operand: two 16-row tiles (chain A, chain B)
// lane row (16 lanes) that holds gate g after the fold (tools/dual_lane_probe.hip: rows hold gates 0 2 1 3)
__device__ __forceinline__ constexpr int drow(int g) { return g == 1 ? 2 : g == 2 ? 1 : g; }

// block list of a wave per phase (0 / 1: GRU layers, 2: output stage): NO old blocks, then 8 fresh ones (tp_common.h)
template <int PH> struct dcp { static constexpr int NO = PH == 0 ? TNO0 : PH == 1 ? TNO1 : TNO3; };
// k-block of position I of that list (only the output stage's conditioning block can lie past the operand: clamped, zero weights)
template <int PH, int I>
__device__ __forceinline__ int dc_kb(int wave, int KB3) {
  if constexpr (PH == 0) return I < TNO0 ? TFR0 + wave + 8 * I : wave + 8 * (I - TNO0);
  else if constexpr (PH == 1) return I < TNO1 ? 64 + wave + 8 * I : wave + 8 * (I - TNO1);
  else if constexpr (I < TNO3) { const int kb = 64 + wave + 8 * I; return kb < KB3 ? kb : KB3 - 1; }
  else return wave + 8 * (I - TNO3);
}
// one k-block of one chain into a ring slot: uniform base (operand of the step + chain tile) + uniform block offset + 16 bytes per lane
template <int PH, int I>
__device__ __forceinline__ void dc_issue(f4& dst, const char* base, unsigned loff, int wave, int KB3) {
  dst = *(const f4*)(base + (size_t)dc_kb<PH, I>(wave, KB3) * (DXB * 4) + loff);
}
// 4-row products of block position I of GRU layer PH for one chain: v_mfma_f32_4x4x1 with cbsz = 2 (12 instructions)
template <int PH, int I, int NW0, int NW1>
__device__ __forceinline__ void dc_comp4(const f4& x, const float (&wq0)[NW0], const float (&wq1)[NW1], const float* w0l_lane, f4 (&acc)[4]) {
  constexpr int nd = tp4_hidden_side(PH, I) ? 3 : 2;
  float w0, w1, w2;
  if constexpr (PH == 0 && I < DL0) { w0 = w0l_lane[(3 * I + 0) * 64]; w1 = w0l_lane[(3 * I + 1) * 64]; w2 = w0l_lane[(3 * I + 2) * 64]; }
  else if constexpr (PH == 0) { w0 = wq0[3 * (I - DL0)]; w1 = wq0[3 * (I - DL0) + 1]; w2 = wq0[3 * (I - DL0) + 2]; }
  else { w0 = wq1[3 * I]; w1 = wq1[3 * I + 1]; w2 = wq1[3 * I + 2]; }
#define DC4_STEP(A)                                                                       \
  acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(w0, x[A], acc[0], 2, A, 0);                 \
  acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(w1, x[A], acc[1], 2, A, 0);                 \
  acc[nd] = __builtin_amdgcn_mfma_f32_4x4x1f32(w2, x[A], acc[nd], 2, A, 0);
  DC4_STEP(0) DC4_STEP(1) DC4_STEP(2) DC4_STEP(3)
#undef DC4_STEP
}
// output stage, block position I: 16-row tile of v_mfma_f32_16x16x4, weights in LDS, two accumulators by block parity
template <int I>
__device__ __forceinline__ void dc_comp16(const f4& x, const f4* wl3, f4 (&acc)[2]) {
  const f4 wv = wl3[I * 64];
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) acc[I & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[cc], x[cc], acc[I & 1], 0, 0, 0);
}
template <int K> struct dci { static constexpr int value = K; };
// compile-time loop: f(dci<I0>{}), ..., f(dci<I0 + N - 1>{})
template <int I0, int N, typename F>
__device__ __forceinline__ void dc_for(F&& f) {
  if constexpr (N > 0) { f(dci<I0>{}); dc_for<I0 + 1, N - 1>(f); }
}

// fold of the four k-quarters (lane rows) of the four gate sums: acc[g][e] at lane (kq, b) -> out[e] at lane (row drow(g), b)
__device__ __forceinline__ f4 dc_fold(const f4 (&acc)[4]) {
  f4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const auto s01 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[0][e]), __float_as_uint(acc[1][e]), false, false);
    const float v1 = __uint_as_float(s01[0]) + __uint_as_float(s01[1]);
    const auto s23 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[2][e]), __float_as_uint(acc[3][e]), false, false);
    const float v2 = __uint_as_float(s23[0]) + __uint_as_float(s23[1]);
    const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v1), __float_as_uint(v2), false, false);
    o[e] = __uint_as_float(s[0]) + __uint_as_float(s[1]);
  }
  return o;
}

// -DZEGGS_DCTIME: wall-clock (100 MHz) stamps of every wave of workgroups 0 and 255 in step T - 2: [wg][wave][slot][6]
// (slot start, old products done, arrival seen, fresh products done, partial sums signalled, epilogue done); tools/dc_time.py
#ifdef ZEGGS_DCTIME
#define DCT(SLOT, I)                                                                                                      \
  do {                                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    if (t == T - 2 && lane == 0 && (c == 0 || c == TNCU - 1))                                                             \
      ((unsigned long long*)(a.cnt + 4096))[(((c != 0) * 8 + wave) * 6 + (SLOT)) * 6 + (I)] = wall_clock64();            \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
  } while (0)
#else
#define DCT(SLOT, I)
#endif

__global__ __launch_bounds__(TTHR, 2) void train_fwd_dual_k(TArgs a) {
  __shared__ f4 w3[8 * TJ3 * 64];                  // output-stage weights of this workgroup (72 KB)
  __shared__ float w0l[8 * 3 * DL0 * 64];          // the first DL0 (old-part) k-blocks of GRU layer 0 (48 KB)
  __shared__ f4 red[2][8][64];                     // partial sums of the slot in flight, per chain and wave (16 KB)
  __shared__ f4 fin[2][64];                        // output-stage sums of a chain (its epilogue wave's exchange)
  __shared__ float gsh[2][48];                     // normalised gaze direction of x_{t+1} per chain and batch row
  __shared__ float cA[4][12];                      // biases of the 4 units: b_ih0, b_hh0, b_ih1, b_hh1 (r, z, n)
  __shared__ float cG[6];                          // gaze columns of x: in_mean[PO..PO+2], 1 / in_std[PO..PO+2]
  __shared__ float cV[4][3];                       // constant of the folded pose columns of GRU layer 0 (r, z, n), steps t > 1
  __shared__ float cW[4][3][3];                    // W_ih0[gate rows of the 4 units][gaze columns]
  __shared__ float cB[16][8];                      // output-stage row constants
  __shared__ unsigned sig[2];                      // partial sums delivered per chain (8 per slot instance, monotonic)
  __shared__ unsigned epi[2];                      // slot instances whose partial sums the epilogue wave has consumed (per chain)
  __shared__ int fail;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c = blockIdx.x;
  const ZeggsDecDims& d = a.d;
  const int B = d.B, T = d.T, H = TH, PO = d.PO, GL = a.GL;
  const long sG = (long)B * GL, sH = (long)B * H;
  // ---------------------------------------------------------------- weights -> registers / LDS (once per rollout)
  float wq0[3 * (TJ0 - DL0)], wq1[3 * TJ1];      // one register per [4 rows x 16 k] tile, three (r, z, n) per k-block
  {
    const float* p0 = (const float*)a.PW0 + ((long)(c * 8 + wave) * TJ0) * 3 * 64 + lane;
#pragma unroll
    for (int i = 0; i < 3 * DL0; ++i) w0l[(wave * 3 * DL0 + i) * 64 + lane] = p0[(long)i * 64];
#pragma unroll
    for (int i = 3 * DL0; i < 3 * TJ0; ++i) wq0[i - 3 * DL0] = p0[(long)i * 64];
    const float* p1 = (const float*)a.PW1 + ((long)(c * 8 + wave) * TJ1) * 3 * 64 + lane;
#pragma unroll
    for (int i = 0; i < 3 * TJ1; ++i) wq1[i] = p1[(long)i * 64];
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
  if (tid == 0) { fail = 0; sig[0] = 0; sig[1] = 0; epi[0] = 0; epi[1] = 0; }
  if (tid >= 64 && tid < 64 + 96) {              // normalised gaze direction of x_1 (canonical row of step 1); later steps: root integration
    const int i = tid - 64, X = i / 48, r = i % 48, gb = 16 * X + r / 3;
    gsh[X][r] = gb < B ? a.Gin[sG + (long)gb * GL + H + PO + r % 3] : 0.f;
  }
  // ---------------------------------------------------------------- recurrent state of the epilogue waves
  // wave 2 L + X owns slot (chain X, phase L); lane (u = lane >> 4, b = lane & 15) of a GRU epilogue wave carries hidden unit 4 c + u
  // of batch row 16 X + b, lane b < 16 of an output-stage epilogue wave the root transform of that row
  float hst = 0.f;
  Q4 rq_ = Q4{1.f, 0.f, 0.f, 0.f};
  V3 rp_ = v3(0.f, 0.f, 0.f);
  {
    const int eX = wave & 1, eL = wave >> 1, gb = 16 * eX + (lane & 15);
    if (eL < 2 && gb < B) hst = (eL == 0 ? a.H0 : a.H1)[(long)gb * H + 4 * c + (lane >> 4)];      // state before the first generated frame
    if (eL == 2 && lane < 16 && gb < B) {
      const float* rq = a.rrot + (long)gb * T * 4;
      const float* rp = a.rpos + (long)gb * T * 3;
      rq_ = Q4{rq[0], rq[1], rq[2], rq[3]};
      rp_ = v3(rp[0], rp[1], rp[2]);
    }
  }
  __syncthreads();
  volatile int* vfail = &fail;
  // producer class / index of this workgroup's arrival flag: it produces a quarter of k-block c / 4 of every exchanged vector
  const int fcls = (c >> 2) & 7, fidx = ((c >> 5) << 2) | (c & 3);
  auto ld_flags = [&](const gu64t* q, unsigned long long& fa, unsigned long long& fb) {
    fa = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    fb = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // all producers of this wave's k-blocks have finished instance pw of chain X (flags fa / fb: a sample taken earlier), and this
  // workgroup's own epilogue wave has consumed the chain's partial sums of that instance (epi).  false: give-up.
  auto wait_arrival = [&](int X, const gu64t* q, long pw, unsigned long long fa, unsigned long long fb) -> bool {
    const unsigned expect = (unsigned)(pw + 1);
    volatile unsigned* ve = &epi[X];
    for (unsigned spins = 0;; ++spins) {
      const bool ok = (unsigned)fa >= expect && (unsigned)(fa >> 32) >= expect && (unsigned)fb >= expect && (unsigned)(fb >> 32) >= expect &&
                      *ve >= expect;
      if (__all(ok)) return true;
      if (spins >= a.spin || *vfail) return false;
      for (unsigned i = 0; i < a.nap; ++i) __builtin_amdgcn_s_sleep(1);
      ld_flags(q, fa, fb);
    }
  };
  auto signal = [&](int X) {      // this wave's partial sums of the chain's slot are in LDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(&sig[X], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto wait_sig = [&](int X, unsigned want) -> bool {
    for (unsigned n = 0;; ++n) {
      if (__hip_atomic_load(&sig[X], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= want) break;
      if (n >= (1u << 24) || *vfail) return false;
      __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
    return true;
  };
  auto consumed = [&](int X, long p) {       // the partial sums of instance p are in this wave's registers
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(&epi[X], (unsigned)(p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto arrive = [&](int X, long p) {         // everything this wave published is on its way past the L2: raise the chain's flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0)
      __hip_atomic_store((gu32*)(a.cnt + X * 256 + fcls * 32 + fidx), (unsigned)(p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  // operand of (phase, step, chain) as a uniform byte address
  auto opnd = [&](int ph, int t, int X) -> const char* {
    const float* g = ph == 0 ? a.G0 + (long)t * a.KB0 * DXB : ph == 1 ? a.G1 + (long)t * 128 * DXB : a.G3 + (long)t * a.KB3 * DXB;
    return (const char*)(g + X * 256);
  };
  f4 xr[DR];                              // the eight fresh k-blocks of the slot in flight
  unsigned long long fa = 0, fb = 0;      // sample of the arrival flags the NEXT slot waits for (taken under this slot's products)
  f4 pend0[2], pend1[2];                  // folded sums of the GRU layers' NEXT step, per chain: the products with vectors published earlier
  const float* const w0l_lane = w0l + wave * 3 * DL0 * 64 + lane;
  const f4* const wl3 = w3 + wave * TJ3 * 64 + lane;
  auto zero4 = [&](f4 (&acc)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f4{0.f, 0.f, 0.f, 0.f};
  };
  // EVERY PUBLISHED VECTOR IS LOADED ONCE.  In the single-chain sweep h0_t is read twice (layer 1 now, layer 0's hidden side next
  // step) and h1_t three times (output stage, layer 1's hidden side, layer 0's pose fold), because the second and third use are
  // what fills its hand-off windows.  Here the other chain fills them, so a slot takes the eight fresh k-blocks of its vector into
  // the ring once and runs ALL products that consume them -- 821 -> 430 KB of operands per CU and step -- and what belongs to the
  // next step's GRU sums waits in folded form (pend0 / pend1: one float4 per lane) until that step's slot adds it.
  // Slot head: the chain's previous phase has arrived -> the fresh blocks (positions NO .. NO + 7 of phase PH's list) into the ring,
  // the next slot's flag sample behind them.
  auto fetch = [&](auto PHC, auto XC, auto XNC, int t, long p, int SL) -> bool {
    constexpr int PH = decltype(PHC)::value, X = decltype(XC)::value, XN = decltype(XNC)::value, NO = dcp<PH>::NO;
    unsigned loff = (unsigned)lane * 16u;
    asm volatile("" : "+v"(loff));
    int wv = wave;                       // (opaque copies: the block offsets are cheap scalar arithmetic, not live scalar pairs)
    asm volatile("" : "+s"(wv));
    const gu64t* q = (const gu64t*)(a.cnt + X * 256 + wv * 32 + 4 * (loff >> 4 & 7));
    DCT(SL, 1);
    if (p > 0 && !wait_arrival(X, q, p - 1, fa, fb)) return false;
    DCT(SL, 2);
    const char* base = opnd(PH, t, X);
    dc_for<0, 8>([&](auto JC) {
      constexpr int J = decltype(JC)::value;
      dc_issue<PH, NO + J>(xr[J], base, loff, wv, a.KB3);
    });
    const gu64t* qn = (const gu64t*)(a.cnt + XN * 256 + wv * 32 + 4 * (loff >> 4 & 7));
    ld_flags(qn, fa, fb);
    __builtin_amdgcn_sched_barrier(0);
    return true;
  };
  // the conditioning block of a slot (position 0 of phase PH's list: speech / style columns, known before the rollout)
  auto fetch_cond = [&](auto PHC, auto XC, int t, f4& xc) {
    constexpr int PH = decltype(PHC)::value, X = decltype(XC)::value;
    unsigned loff = (unsigned)lane * 16u;
    asm volatile("" : "+v"(loff));
    int wv = wave;
    asm volatile("" : "+s"(wv));
    dc_issue<PH, 0>(xc, opnd(PH, t, X), loff, wv, a.KB3);
    __builtin_amdgcn_sched_barrier(0);
  };

  // ================================================================ GRU slot of chain X, layer L
  auto gru_slot = [&](auto LC, auto XC, int t) -> bool {
    constexpr int L = decltype(LC)::value, X = decltype(XC)::value, SL = 2 * L + X;
    const long p = 3L * (t - 1) + L;
    const bool next = t + 1 < T;
    f4 acc[4];
    zero4(acc);
    DCT(SL, 0);
    if constexpr (L == 0) {       // [cond_t | hid_t]; the h0_{t-1} / h1_{t-1} parts were added up when those vectors were fresh (pend0)
      f4 xc;
      fetch_cond(dci<0>{}, dci<X>{}, t, xc);
      if (!fetch(dci<0>{}, dci<X>{}, dci<1 - X>{}, t, p, SL)) return false;
      dc_comp4<0, 0>(xc, wq0, wq1, w0l_lane, acc);
      dc_for<0, 8>([&](auto JC) {
        constexpr int J = decltype(JC)::value;
        dc_comp4<0, TNO0 + J>(xr[J], wq0, wq1, w0l_lane, acc);
        __builtin_amdgcn_sched_barrier(0);
      });
      DCT(SL, 3);
      red[X][wave][lane] = dc_fold(acc) + pend0[X];
      signal(X);
    } else {                      // h0_t: layer 1's input side now, layer 0's hidden side of step t + 1 (-> pend0)
      f4 acc0[4];
      zero4(acc0);
      if (!fetch(dci<1>{}, dci<X>{}, dci<1 - X>{}, t, p, SL)) return false;
      dc_for<0, 8>([&](auto JC) {
        constexpr int J = decltype(JC)::value;
        dc_comp4<1, TNO1 + J>(xr[J], wq0, wq1, w0l_lane, acc);
        dc_comp4<0, 1 + J>(xr[J], wq0, wq1, w0l_lane, acc0);
        __builtin_amdgcn_sched_barrier(0);
      });
      DCT(SL, 3);
      red[X][wave][lane] = dc_fold(acc) + pend1[X];
      signal(X);
      pend0[X] = dc_fold(acc0);
    }
    DCT(SL, 4);
    if (wave != SL) return true;
    // ---------------------------------------------------------------- epilogue (this wave only)
    if (!wait_sig(X, 8u * (unsigned)(p + 1))) return false;
    int le = lane;
    asm volatile("" : "+v"(le));
    const int b = le & 15, u = le >> 4, gb = 16 * X + b;
    const bool act = gb < B;
    float s[4];
    {
      const float* rf = (const float*)&red[X][0][0] + b * 4 + u;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v = rf[(16 * drow(g)) * 4];
#pragma unroll
        for (int w = 1; w < 8; ++w) v += rf[(w * 64 + 16 * drow(g)) * 4];
        s[g] = v;
      }
    }
    consumed(X, p);
    const float* k_ = cA[u];
    float r, z, nh, nn;
    if constexpr (L == 0) {
      // pose columns of x_t: N0 h1_{t-1} (in the products) + cv0; step 1: the product with the given first pose
      float xr = cV[u][0], xz = cV[u][1], xn = cV[u][2];
      if (t == 1) {
        xr = xz = xn = 0.f;
        if (act) { const float* qq = a.p1x + (long)gb * 3 * H + 4 * c + u; xr = qq[0]; xz = qq[H]; xn = qq[2 * H]; }
      }
      const float g0 = gsh[X][b * 3], g1 = gsh[X][b * 3 + 1], g2 = gsh[X][b * 3 + 2];      // gaze columns of x_t
      const float (*wg)[3] = cW[u];
      xr += wg[0][0] * g0 + wg[0][1] * g1 + wg[0][2] * g2;
      xz += wg[1][0] * g0 + wg[1][1] * g1 + wg[1][2] * g2;
      xn += wg[2][0] * g0 + wg[2][1] * g1 + wg[2][2] * g2;
      r = d_sigmoid(s[0] + k_[0] + xr + k_[3]);
      z = d_sigmoid(s[1] + k_[1] + xz + k_[4]);
      nh = s[3] + k_[5];
      nn = d_tanh(s[2] + k_[2] + xn + r * nh);
    } else {
      r = d_sigmoid(s[0] + k_[6] + k_[9]);
      z = d_sigmoid(s[1] + k_[7] + k_[10]);
      nh = s[3] + k_[11];
      nn = d_tanh(s[2] + k_[8] + r * nh);
    }
    const float h = (1.f - z) * nn + z * hst;
    hst = h;
    const float h1 = __shfl(h, b + 16), h2 = __shfl(h, b + 32), h3 = __shfl(h, b + 48);
    if (le < 16 && act) {       // the four units of batch row gb: one float4 of every operand that takes them
      const f4 hv = f4{h, h1, h2, h3};
      const long o = xfi(gb, 4 * c, 2);
      if constexpr (L == 0) {
        stp4(a.G1 + (long)t * 128 * DXB + o, hv);                                                 // h0_t: the one operand every consumer reads
        *(f4*)(a.H0 + (long)t * sH + (long)gb * H + 4 * c) = hv;
      } else {
        stp4(a.G3 + (long)t * a.KB3 * DXB + o, hv);                                               // h1_t: the one operand every consumer reads
        *(f4*)(a.H1 + (long)t * sH + (long)gb * H + 4 * c) = hv;
      }
    }
    if (act) ((f4*)(L == 0 ? a.GT0 : a.GT1))[(long)t * sH + (long)gb * H + 4 * c + u] = f4{r, z, nn, nh};
    arrive(X, p);
    DCT(SL, 5);
    return true;
  };

  // ================================================================ output-stage slot of chain X : [h1_t | cond_{t+1}]
  // h1_t also feeds layer 1's hidden side (-> pend1) and, through the pose fold, layer 0's input side (-> pend0) of step t + 1
  auto out_slot = [&](auto XC, int t) -> bool {
    constexpr int X = decltype(XC)::value, SL = 4 + X;
    const long p = 3L * (t - 1) + 2;
    const bool next = t + 1 < T;
    float gz_[3] = {0.f, 0.f, 0.f};          // gaze target of frame t+1 (an input): in flight under the products
    {
      int lx = lane;
      asm volatile("" : "+v"(lx));
      if (wave == SL && lx < 16 && 16 * X + lx < B && next) {
        const float* gz = a.gaze + ((long)(16 * X + lx) * T + t + 1) * 3;
        gz_[0] = gz[0]; gz_[1] = gz[1]; gz_[2] = gz[2];
      }
    }
    DCT(SL, 0);
    {
      f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}}, acc1[4], acc0[4], xc;
      zero4(acc1);
      zero4(acc0);
      fetch_cond(dci<2>{}, dci<X>{}, t, xc);
      if (!fetch(dci<2>{}, dci<X>{}, dci<1 - X>{}, t, p, SL)) return false;
      dc_comp16<0>(xc, wl3, acc);
      dc_for<0, 8>([&](auto JC) {
        constexpr int J = decltype(JC)::value;
        dc_comp16<TNO3 + J>(xr[J], wl3, acc);
        dc_comp4<1, J>(xr[J], wq0, wq1, w0l_lane, acc1);
        dc_comp4<0, 9 + J>(xr[J], wq0, wq1, w0l_lane, acc0);
        __builtin_amdgcn_sched_barrier(0);
      });
      DCT(SL, 3);
      red[X][wave][lane] = acc[0] + acc[1];
      signal(X);
      pend1[X] = dc_fold(acc1);
      pend0[X] += dc_fold(acc0);
    }
    DCT(SL, 4);
    if (wave != SL) return true;
    // ---------------------------------------------------------------- epilogue (this wave only)
    if (!wait_sig(X, 8u * (unsigned)(p + 1))) return false;
    int le = lane;
    asm volatile("" : "+v"(le));
    {
      f4 s = red[X][0][le];
#pragma unroll
      for (int w = 1; w < 8; ++w) s += red[X][w][le];
      consumed(X, p);
      fin[X][le] = s;         // float4 (row group l >> 4, batch row l & 15): rows 4 (l >> 4) .. + 3
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const float* ff = (const float*)&fin[X][0];
    auto FV = [&](int vcol, int bb) -> float { return ff[((((vcol >> 2) << 4) | bb) << 2) | (vcol & 3)]; };
    float* gnext = a.Gin + (long)(t + 1) * sG;                       // canonical [hid | x] row of step t+1
    float* xnext = a.G0 + (long)(t + 1) * a.KB0 * DXB;               // its fragment copy
    const int b = le & 15, gb = 16 * X + b;
    if (le < 16) {            // root integration of batch row gb (ZEGGS/modules.py:139-176)
      float genc[3] = {0.f, 0.f, 0.f};
      if (gb < B) {
        float pv[6];
#pragma unroll
        for (int qq = 0; qq < 6; ++qq) pv[qq] = (FV(9 + qq, b) + cB[9 + qq][0]) * cB[9 + qq][1] + cB[9 + qq][2];
        const Q4 qr = rq_;
        const V3 pos = rp_;
        const V3 npos = quat_mul_vec(qr, d.dt * v3(pv[0], pv[1], pv[2])) + pos;
        const V3 uu = quat_mul_vec(qr, d.dt * v3(pv[3], pv[4], pv[5]));
        const Q4 nq = quat_exp_mul(0.5f * uu, qr);
        rq_ = nq; rp_ = npos;
        if (next) {
          const V3 gd = quat_mul_vec(quat_inv(nq), v3(gz_[0], gz_[1], gz_[2]) - npos);
          genc[0] = (gd.x - cG[0]) * cG[3];
          genc[1] = (gd.y - cG[1]) * cG[4];
          genc[2] = (gd.z - cG[2]) * cG[5];
        }
        if (c == 0) {
          float* op = a.rpos + ((long)gb * T + t) * 3;
          op[0] = npos.x; op[1] = npos.y; op[2] = npos.z;
          float* oq = a.rrot + ((long)gb * T + t) * 4;
          oq[0] = nq.w; oq[1] = nq.x; oq[2] = nq.y; oq[3] = nq.z;
          if (next)
            for (int k = 0; k < 3; ++k) {
              gnext[(long)gb * GL + H + PO + k] = genc[k];
              stp(xnext + 64 * DXB + xfi(gb, k, 2), genc[k]);               // the gaze block of layer 0's operand (unread)
            }
        }
      }
      gsh[X][b * 3] = genc[0]; gsh[X][b * 3 + 1] = genc[1]; gsh[X][b * 3 + 2] = genc[2];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {     // folded layer0 rows: hid_{t+1} (lane row = row of the tile, 0..3)
      const int vc = le >> 4;
      const float* k_ = cB[vc];
      const float val = d_elu(FV(vc, b) + k_[0] + k_[1] * gsh[X][b * 3] + k_[2] * gsh[X][b * 3 + 1] + k_[3] * gsh[X][b * 3 + 2]);
      const float v1 = __shfl(val, b + 16), v2 = __shfl(val, b + 32), v3_ = __shfl(val, b + 48);
      if (next && le < 16 && gb < B) {
        const f4 v = f4{val, v1, v2, v3_};
        stp4(xnext + xfi(gb, 4 * c, 2), v);
        *(f4*)(gnext + (long)gb * GL + 4 * c) = v;
      }
    }
    for (int item = le; item < 5 * 16; item += 64) {        // layer2 rows: pose_t and the pose columns of x_{t+1}
      const int vc = 4 + (item >> 4), bb = item & 15, gbb = 16 * X + bb;
      if (gbb < B && cB[vc][5] != 0.f) {
        const float* k_ = cB[vc];
        const int col = c + TNCU * (vc - 4);
        const float pv = (FV(vc, bb) + k_[0]) * k_[1] + k_[2];
        a.pose[((long)gbb * T + t) * PO + col] = pv;
        if (next) gnext[(long)gbb * GL + H + col] = (pv - k_[3]) / k_[4];      // (canonical only: the products take the pose columns through the fold)
      }
    }
    arrive(X, p);
    DCT(SL, 5);
    return true;
  };

  // pending sums of step 1: W_hh0 h0_0 (layer 0's operand of step 1 carries h0_0 in its h0 slot and zeros in its h1 slot: the pose
  // columns of x_1 come from the given first pose, p1x) and W_hh1 h1_0 (the second half of layer 1's operand of step 1)
  dc_for<0, 2>([&](auto XC) {
    constexpr int X = decltype(XC)::value;
    f4 acc0[4], acc1[4];
    zero4(acc0);
    zero4(acc1);
    const unsigned loff = (unsigned)lane * 16u;
    dc_for<0, 8>([&](auto JC) {
      constexpr int J = decltype(JC)::value;
      dc_issue<0, 1 + J>(xr[J], opnd(0, 1, X), loff, wave, a.KB3);
    });
    dc_for<0, 8>([&](auto JC) {
      constexpr int J = decltype(JC)::value;
      dc_comp4<0, 1 + J>(xr[J], wq0, wq1, w0l_lane, acc0);
    });
    dc_for<0, 8>([&](auto JC) {
      constexpr int J = decltype(JC)::value;
      dc_issue<1, J>(xr[J], opnd(1, 1, X), loff, wave, a.KB3);
    });
    dc_for<0, 8>([&](auto JC) {
      constexpr int J = decltype(JC)::value;
      dc_comp4<1, J>(xr[J], wq0, wq1, w0l_lane, acc1);
    });
    pend0[X] = dc_fold(acc0);
    pend1[X] = dc_fold(acc1);
  });
  bool okrun = true;
  for (int t = 1; t < T; ++t) {
    okrun = gru_slot(dci<0>{}, dci<0>{}, t) && gru_slot(dci<0>{}, dci<1>{}, t) && gru_slot(dci<1>{}, dci<0>{}, t) &&
            gru_slot(dci<1>{}, dci<1>{}, t) && out_slot(dci<0>{}, t) && out_slot(dci<1>{}, t);
    if (!okrun) break;
  }
  if (!okrun) *vfail = 1;
  __syncthreads();
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

}  // namespace

namespace zeggs_tp {
int tp_dual_supported(int NB) { return g_tp_dual && NB == 2; }
void tp_dual_launch(const TArgs& a, hipStream_t s) { hipLaunchKernelGGL(train_fwd_dual_k, dim3(TNCU), dim3(TTHR), 0, s, a); }
}  // namespace zeggs_tp

// -DZEGGS_DCTIME builds: the slot stamps of workgroups 0 and 255 (tools/dc_time.py)
extern "C" int zeggs_tp_dual_stamps(const ZeggsDecDims* dp, void* ws, size_t ws_bytes, unsigned long long* out /* [2][8][6][6] */) {
  Arena a(ws, ws_bytes);
  DecWs w = carve_dec(*dp, 1, a);
  ZCHECK(a.ok() && w.tp_cnt, "tp_dual_stamps: workspace");
  ZCHECK(hipMemcpy(out, w.tp_cnt + 4096, 2 * 8 * 6 * 6 * 8, hipMemcpyDeviceToHost) == hipSuccess, "copy");
  return 0;
}
