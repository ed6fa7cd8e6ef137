// Shared pieces of the persistent training-forward rollouts (train_persistent.hip: one chain of <= 64 batch rows per launch;
// train_dual.hip: two independent 16-row chains in one launch): block schedule constants, kernel arguments, the write-through
// publishes and the operand layouts.
#pragma once
#include "decoder_ws.h"
#include "dec_math.h"

namespace zeggs_tp {

typedef __attribute__((address_space(1))) unsigned gu32;
constexpr int TH = 1024, TTHR = 512, TNCU = 256;
// k-blocks of a wave per phase: first the OLD part of the operand (known one phase earlier: previous hidden state,
// speech / style columns), then the FRESH part (produced by the preceding phase).  Block j of a part is k-block
// lo + wave + 8 j: the parts are interleaved over the 8 waves so that every wave owns old work to do before the hand-off.
constexpr int TNO0 = 17, TNF0 = 8, TNO1 = 8, TNF1 = 8, TNO3 = 1, TNF3 = 8;
constexpr int TJ0 = TNO0 + TNF0, TJ1 = TNO1 + TNF1, TJ3 = TNO3 + TNF3;
// GRU layer 0 operand of a step, in k-blocks: [hid_t (64) | gaze direction of x_t (1) | speech / style of x_t (TKC, zero
// padded) | h0_{t-1} (64) | h1_{t-1} (64)].  The POSE columns of x_t are not an operand: between the output stage of step t-1
// and this product the reference only de-normalises / re-normalises them (modules.py:60-76), so
//   W_ih0[:, pose] x_t[pose] = N0 h1_{t-1} + cv0,  N0 = W_ih0[:, pose] diag(sigma_o / sigma_i) W2  (re-derived per optimizer step),
// which turns 71 blocks that had to wait for the output stage into 64 that are old by then.  (Step 1 takes x_1 from the given
// first pose instead: its product comes from a small prologue GEMM and the h1 slot of that step stays zero.)
// Only k-blocks [0, TFR0) are fresh; the old ones are ordered [cond | h0 | h1]: h0_{t-1} is two hand-offs old when the previous
// output stage waits, h1_{t-1} one.
constexpr int TKC = 8, TFR0 = 65, TKH0 = TFR0 + TKC, TKH1 = TKH0 + 64, TKB0 = TKH1 + 64;      // 65, 73, 137, 201
// ... of which only the 64 blocks of hid_t are walked: the gaze block would give ONE wave a ninth fresh block (every workgroup
// waits for it: +1/8 on the matrix-core time of the phase's critical part) for three columns -- the gate threads add them instead
// (9 FMAs each; every workgroup has the normalised gaze direction in LDS anyway, from its own root integration).  The block keeps
// its place in the operand layout (unread).
constexpr int TFRW = 64;
// old blocks of GRU layer 0 done one window early (cond + h0_{t-1}: in front of the previous output stage) / of GRU layer 1 done
// in layer 0's window (batch <= 32; the wider variants have no registers to spare for a second live accumulator)
constexpr int ts0(int nb) { return nb <= 2 ? 9 : 0; }
constexpr int ts1(int nb) { return 0 * nb; }
// old-part k-blocks of GRU layer 0 parked in LDS instead of registers (as many as the LDS budget of the variant allows)
constexpr int tl0(int nb) { return nb <= 2 ? 8 : nb == 3 ? 6 : 5; }
__host__ __device__ inline int tp_kb(int i, int wave, int NO, int old_lo, int old_hi, int fresh_hi) {
  if (i < NO) { const int kb = old_lo + wave + 8 * i; return kb < old_hi ? kb : -1; }
  const int kb = wave + 8 * (i - NO);
  return kb < fresh_hi ? kb : -1;
}
constexpr int TSH = 8, TSTR = 32, TRING = 4;

struct TArgs {
  ZeggsDecDims d;
  ZeggsDecStats st;
  int XD, GL, KBX, KBC, KB0, KB3, POL;
  const f4 *PW0, *PW1, *PW3;                 // per-workgroup fragment packs [256][KB][64]
  float *G0, *G1, *G3;                       // operand fragments, time-major [T][KB*][NB][64][4]
  float *Gin, *H0, *H1, *GT0, *GT1;          // canonical saves (time-major)
  const float *b_ih0, *b_hh0, *b_ih1, *b_hh1, *cvec, *l0_w, *l2_b;
  const float* w_ih0;                        // [3H][H + XD]: its three gaze columns (H + PO ..) are applied by the gate threads
  const float *cv0, *p1x;                    // folded pose term of GRU layer 0: constant [3H], step-1 product [B][3H]
  const float* gaze;
  float *pose, *rpos, *rrot;
  unsigned *cnt, *err;
  unsigned* status;                          // caller-owned sticky give-up flags (ZeggsDecCall.status), may be null
  unsigned spin;                             // bound of every wait (option "persistent_spin")
  unsigned nap;                              // s_sleep units between two polls (option "poll_sleep")
  unsigned stag;                             // != 0: two staggered polls in flight (option "poll_stagger")
};

__device__ __forceinline__ void stp(float* p, float v) {       // published: write-through
  __hip_atomic_store((gu32*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 16 bytes, write-through: the four hidden units of this workgroup are four consecutive k of one batch row = one float4 of the
// operand layout.  The "memory" clobber is required (results are corrupted without it) and makes the compiler drain the stores
// it knows about first, so stp4 goes BEFORE the plain stores of an epilogue (train_bwd_persistent.hip).
__device__ __forceinline__ void stp4(float* p, f4 v) {
#ifdef ZEGGS_TP_NOSTP      // (timing experiment, results wrong: nothing is published)
  asm volatile("" ::"v"(p), "v"(v) : "memory");
#else
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#endif
}
__device__ __forceinline__ long xfi(int b, int k, int NB) {   // B-fragment position of (batch row, k)
  return ((((long)(k >> 4) * NB + (b >> 4)) * 64 + ((((k >> 2) & 3) << 4) | (b & 15))) << 2) | (k & 3);
}

typedef __attribute__((address_space(1))) unsigned long long gu64t;
// Operand position of (batch row, k) for that instruction form: lane 32 * ((k >> 3) & 1) + (b & 31) reads float4 q = (k >> 2) & 1
// of block k >> 4 (batch tile b >> 5), element k & 3 = abid & 3.  A block is 512 floats per 32 batch rows: the same size as two
// 16-row tiles of the 16x16x4 layout, so the block offsets of the operand buffers do not change.
__host__ __device__ inline long xf4(int b, int k, int NT) {
  return (((((long)(k >> 4) * NT + (b >> 5)) * 2 + ((k >> 2) & 1)) * 64 + ((((k >> 3) & 1) << 5) | (b & 31))) << 2) | (k & 3);
}
// block i (position in the wave's list: old part first) of phase ph has hidden-side n rows: layer 0 = [cond | 8 x h0_{t-1} |
// 8 x h1_{t-1} through the fold (input side: pose columns) | 8 x hid_t], layer 1 = [8 x h1_{t-1} | 8 x h0_t]
__host__ __device__ constexpr bool tp4_hidden_side(int ph, int i) { return ph == 0 ? (i >= 1 && i <= 8) : (i < TNO1); }

}  // namespace zeggs_tp
