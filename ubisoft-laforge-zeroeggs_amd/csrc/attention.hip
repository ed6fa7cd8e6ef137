// Fused multi-head self-attention of the style encoder's FFT block (reference ZEGGS/modules.py:516-557,
// nn.MultiheadAttention with 4 heads of 32 channels): softmax(Q K^T / sqrt(hd)) -> dropout -> . V in ONE kernel, and the
// backward in two (dQ; dK + dV) that recompute the probabilities from the saved row log-sum-exp -- no [B*heads, L, L] score /
// probability matrices in memory (the GEMM + softmax + GEMM path writes and re-reads three of them: 0.85 ms of the training
// iteration's tail at L = 384, 2.5 GB at the 7 200-frame exemplar of generate.py).
//
// Shape of the work: head dimension 32 = one k-extent of sixteen v_mfma_f32_32x32x2_f32; a wave owns 32 query rows (forward,
// dQ) or 32 keys (dK / dV) and walks the other axis in tiles of 32 that the four waves of a workgroup share through LDS.
// The orientation of every score tile is chosen so that the matrix-core OUTPUT layout of one product (lane = column n,
// registers = 16 rows m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) is directly the B operand of the next one (lane = column,
// k-slot 2 kp + (lane >> 5)): the contraction index of the second product is walked in the permuted order
// slot (kp, kh) <-> row (kp & 3) + 8 (kp >> 2) + 4 kh, which only changes which LDS row the A operand is read from.
//   forward, dQ:  S^T[key][q]  = K Q^T    -> per-lane softmax statistics of query q = lane & 31 (one shuffle joins the two
//                                            k-halves), then O^T[d][q] += V^T[d][key] P^T[key][q] / dQ^T[d][q] += K^T dS^T
//   dK, dV:       S[q][key]    = Q K^T    -> dV^T[d][key] += dO^T[d][q] Pd[q][key],  dK^T[d][key] += Q^T[d][q] dS[q][key]
// Dropout masks are the library's counter hash on the element index of the (virtual) [B, heads, L, L] probability tensor:
// the same masks as the unfused path, regenerated in the backward.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int HD = 32, LD = 33, WQ = 128;      // head dim, padded LDS row, rows of a workgroup (4 waves x 32)
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

struct AttnArgs {
  const float* qkv;      // [B*L, 3E]: q | k | v, head h at columns h*32
  float* O;              // [B*L, E]   attention output (before out_proj), head h at columns h*32
  float* lse;            // [B*NH, L]  row log-sum-exp of the scaled scores, in the BASE-2 domain (log2 sum_j 2^(s_j log2 e))
  const float* dO;       // backward
  float* dqkv;           // [B*L, 3E]
  float* dbias;          // != null: [3E] += column sums of dqkv (the in-projection's bias gradient), atomics
  float* dsum;           // [B*NH, L]  rowsum(dO . O)
  int L, E, NH;
  float scale, p;
  uint64_t seed;
};

__device__ __forceinline__ int mrow(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }

// Round 5.  What the round-4 kernels spent their time on was not the matrix cores (48 / 64 products of 64 cycles per 32 x 32 tile)
// but (i) the mask hash (three 64-bit multiplies per probability: common.h), (ii) sixteen exec-masked branches per tile around
// libm's expf, (iii) a global -> LDS round trip with two barriers in the open per tile, (iv) a prologue of three staged operands
// behind six barriers.  Now: probabilities as ONE v_exp_f32 each (scores carry the factor log2 e, folded into the query scale;
// the saved row statistic is the base-2 log-sum-exp), masks from the 32-bit hash, bounds as selects; the next tile's K / V (Q /
// dO) rows are fetched into registers before the current tile's products and written to the other LDS buffer behind them (one
// barrier per tile, the fetch latency under the products); a wave's own 32 rows (Q, dO, O / K, V) are loaded straight into
// the matrix-core operand slots -- a lane reads its row (128 contiguous bytes) and keeps the 16 entries of its k-half.

// the 32 entries of row `row` of a [rows, ld] matrix -> the 16 B-operand slots of lane (row, kh): slot kp = row[2 kp + kh]
__device__ __forceinline__ void load_row_slots(float (&f)[16], const float* src, long ld, int row, float mul, int kh) {
  const f4* p = (const f4*)(src + (long)row * ld);
  f4 v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = p[j];
#pragma unroll
  for (int kp = 0; kp < 16; ++kp) {
    const f4 t = v[kp >> 1];
    const float e0 = (kp & 1) ? t[2] : t[0], e1 = (kp & 1) ? t[3] : t[1];
    f[kp] = (kh ? e1 : e0) * mul;
  }
}

// one float4 per thread of a 32-row x 32-float tile of a [rows, ld] matrix (256 threads; rows clamped to nrows - 1)
__device__ __forceinline__ f4 tile_fetch(const float* src, long ld, int row0, int nrows, int tid) {
  int row = row0 + (tid >> 3);
  row = row < nrows ? row : nrows - 1;
  return *(const f4*)(src + (long)row * ld + (tid & 7) * 4);
}
__device__ __forceinline__ void tile_put(float* dst, f4 v, float mul, int tid) {
  float* d = dst + (tid >> 3) * LD + (tid & 7) * 4;
  d[0] = v[0] * mul; d[1] = v[1] * mul; d[2] = v[2] * mul; d[3] = v[3] * mul;
}

__device__ __forceinline__ float half_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float half_sum(float v) { return v + __shfl_xor(v, 32, 64); }
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }      // v_exp_f32 (flushes to 0 far below)

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// 16 accumulator entries of lane (col, kh) -> the lane's output row: entries 4 g .. 4 g + 3 are columns 8 g + 4 kh .. + 3
__device__ __forceinline__ void store_row16(float* row, const f16v& v, float mul, int kh) {
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *(f4*)(row + 8 * g + 4 * kh) = f4{v[4 * g] * mul, v[4 * g + 1] * mul, v[4 * g + 2] * mul, v[4 * g + 3] * mul};
}

// ------------------------------------------------------------------ forward
template <bool DROP>
__global__ __launch_bounds__(256) void attn_fwd_k(AttnArgs a) {
  __shared__ float Ks[2][32 * LD], Vs[2][32 * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.NH, h = bh % a.NH, L = a.L;
  const long ld = 3L * a.E;
  const float* base = a.qkv + (long)b * L * ld + h * HD;
  const int myq = blockIdx.x * WQ + wave * 32 + l31, cq = myq < L ? myq : L - 1;
  f4 kn = tile_fetch(base + a.E, ld, 0, L, tid), vn = tile_fetch(base + 2 * a.E, ld, 0, L, tid);
  float qf[16];
  load_row_slots(qf, base, ld, cq, a.scale * LOG2E, kh);
  const uint64_t prow64 = (uint64_t)((long)bh * L + myq) * (uint64_t)L;      // element index of P[bh][myq][0]
  const uint32_t prow = (uint32_t)prow64, phi = (uint32_t)(prow64 >> 32);
  const HKey hkey = hash_key(a.seed);
  const float inv_keep = 1.f / (1.f - a.p);
  f16v o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m = -INFINITY, lsum = 0.f;
  tile_put(Ks[0], kn, 1.f, tid);
  tile_put(Vs[0], vn, 1.f, tid);
  __syncthreads();
  const int nt = (L + 31) >> 5;
  for (int t = 0; t < nt; ++t) {
    const int k0 = t * 32, cur = t & 1;
    if (t + 1 < nt) { kn = tile_fetch(base + a.E, ld, k0 + 32, L, tid); vn = tile_fetch(base + 2 * a.E, ld, k0 + 32, L, tid); }
    const float* K = Ks[cur];
    const float* V = Vs[cur];
    f16v s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int kp = 0; kp < 16; ++kp) s = MFMA32(K[l31 * LD + 2 * kp + kh], qf[kp], s);      // S^T[key][q] (times log2 e)
    float mx = m;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = (k0 + mrow(r, kh) < L) ? s[r] : -INFINITY;
      mx = fmaxf(mx, s[r]);
    }
    mx = half_max(mx);
    const float corr = ex2(m - mx);          // first tile: 2^(-inf) = 0
    float ps = 0.f, pd[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + mrow(r, kh);
      const float pv = ex2(s[r] - mx);       // masked keys: 2^(-inf) = 0
      ps += pv;
      if constexpr (DROP) {
        const uint32_t lo = prow + (uint32_t)key;
        pd[r] = pv * dropout_scale_fast(hkey, lo, phi + (lo < prow), a.p, inv_keep);
      } else pd[r] = pv;
    }
    ps = half_sum(ps);
    lsum = lsum * corr + ps;
    m = mx;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= corr;
#pragma unroll
    for (int kp = 0; kp < 16; ++kp) o = MFMA32(V[mrow(kp, kh) * LD + l31], pd[kp], o);      // O^T[d][q] += V^T P^T
    if (t + 1 < nt) { tile_put(Ks[cur ^ 1], kn, 1.f, tid); tile_put(Vs[cur ^ 1], vn, 1.f, tid); }
    __syncthreads();
  }
  if (myq < L) {
    store_row16(a.O + ((long)b * L + myq) * a.E + h * HD, o, 1.f / lsum, kh);
    if (kh == 0) a.lse[(long)bh * L + myq] = m + __builtin_amdgcn_logf(lsum);      // v_log_f32 = log2
  }
}

// ------------------------------------------------------------------ backward: dQ (and dsum = rowsum(dO . O))
template <bool DROP>
__device__ __forceinline__ void attn_bwd_q_body(const AttnArgs& a) {
  __shared__ float Ks[2][32 * LD], Vs[2][32 * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.NH, h = bh % a.NH, L = a.L;
  const long ld = 3L * a.E;
  const float* base = a.qkv + (long)b * L * ld + h * HD;
  const int myq = blockIdx.x * WQ + wave * 32 + l31, cq = myq < L ? myq : L - 1;
  f4 kn = tile_fetch(base + a.E, ld, 0, L, tid), vn = tile_fetch(base + 2 * a.E, ld, 0, L, tid);
  float qf[16], dof[16], of[16];
  load_row_slots(qf, base, ld, cq, a.scale * LOG2E, kh);
  load_row_slots(dof, a.dO + (long)b * L * a.E + h * HD, a.E, cq, 1.f, kh);
  load_row_slots(of, a.O + (long)b * L * a.E + h * HD, a.E, cq, 1.f, kh);
  float dq_ = 0.f;
#pragma unroll
  for (int kp = 0; kp < 16; ++kp) dq_ += dof[kp] * of[kp];
  const float Dq = half_sum(dq_);
  const float lse = myq < L ? a.lse[(long)bh * L + cq] : INFINITY;      // rows beyond L: every probability 2^(-inf) = 0
  const uint64_t prow64 = (uint64_t)((long)bh * L + myq) * (uint64_t)L;      // element index of P[bh][myq][0]
  const uint32_t prow = (uint32_t)prow64, phi = (uint32_t)(prow64 >> 32);
  const HKey hkey = hash_key(a.seed);
  const float inv_keep = 1.f / (1.f - a.p);
  f16v dqT;
#pragma unroll
  for (int r = 0; r < 16; ++r) dqT[r] = 0.f;
  tile_put(Ks[0], kn, 1.f, tid);
  tile_put(Vs[0], vn, 1.f, tid);
  __syncthreads();
  const int nt = (L + 31) >> 5;
  for (int t = 0; t < nt; ++t) {
    const int k0 = t * 32, cur = t & 1;
    if (t + 1 < nt) { kn = tile_fetch(base + a.E, ld, k0 + 32, L, tid); vn = tile_fetch(base + 2 * a.E, ld, k0 + 32, L, tid); }
    const float* K = Ks[cur];
    const float* V = Vs[cur];
    f16v s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int kp = 0; kp < 16; ++kp) {
      s = MFMA32(K[l31 * LD + 2 * kp + kh], qf[kp], s);         // S^T[key][q] (times log2 e)
      dp = MFMA32(V[l31 * LD + 2 * kp + kh], dof[kp], dp);      // dPd^T[key][q] = V dO^T
    }
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + mrow(r, kh);
      float pv = ex2(s[r] - lse);
      pv = key < L ? pv : 0.f;
      float sc = 1.f;
      if constexpr (DROP) {
        const uint32_t lo = prow + (uint32_t)key;
        sc = dropout_scale_fast(hkey, lo, phi + (lo < prow), a.p, inv_keep);
      }
      ds[r] = pv * (dp[r] * sc - Dq);
    }
#pragma unroll
    for (int kp = 0; kp < 16; ++kp) dqT = MFMA32(K[mrow(kp, kh) * LD + l31], ds[kp], dqT);  // dQ^T[d][q] += K^T dS^T
    if (t + 1 < nt) { tile_put(Ks[cur ^ 1], kn, 1.f, tid); tile_put(Vs[cur ^ 1], vn, 1.f, tid); }
    __syncthreads();
  }
  if (myq < L) {
    store_row16(a.dqkv + ((long)b * L + myq) * ld + h * HD, dqT, a.scale, kh);
    if (kh == 0) a.dsum[(long)bh * L + myq] = Dq;
  }
  if (a.dbias) {      // column sums over this wave's 32 queries (lanes of one half-wave hold one d each)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = myq < L ? dqT[r] * a.scale : 0.f;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (l31 == 0) atomicAdd(a.dbias + h * HD + mrow(r, kh), v);
    }
  }
}

// ------------------------------------------------------------------ backward: dK, dV
// OWN: the workgroup forms rowsum(dO . O) of every query tile itself (one more float4 per thread and tile, a dot product over the eight
// threads of a row) instead of reading what the dQ pass stored -- the two passes are then independent of each other (attn_bwd_k)
template <bool DROP, bool OWN>
__device__ __forceinline__ void attn_bwd_kv_body(const AttnArgs& a) {
  __shared__ float Qs[2][32 * LD], Gs[2][32 * LD], ls[2][32], dsm[2][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.NH, h = bh % a.NH, L = a.L;
  const long ld = 3L * a.E;
  const float* base = a.qkv + (long)b * L * ld + h * HD;
  const float* dOb = a.dO + (long)b * L * a.E + h * HD;
  const float* Ob = a.O + (long)b * L * a.E + h * HD;
  const int mykey = blockIdx.x * WQ + wave * 32 + l31, ck = mykey < L ? mykey : L - 1;
  const float qmul = a.scale * LOG2E;
  auto fetch_stats = [&](int q0, float& l, float& d) {
    if (tid < 32) {
      const int q = q0 + tid;
      const int qc = q < L ? q : L - 1;
      l = q < L ? a.lse[(long)bh * L + qc] : INFINITY;           // queries beyond L: probability 2^(-inf) = 0
      if constexpr (!OWN) d = a.dsum[(long)bh * L + qc];
    }
  };
  auto own_dsum = [&](int q0, const f4& g) {      // thread (row tid >> 3, quarter tid & 7) -> the row's sum in the quarter-0 thread
    const f4 o = tile_fetch(Ob, a.E, q0, L, tid);
    float v = g[0] * o[0] + g[1] * o[1] + g[2] * o[2] + g[3] * o[3];
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    return v;
  };
  f4 qn = tile_fetch(base, ld, 0, L, tid), gn = tile_fetch(dOb, a.E, 0, L, tid);
  float ln = 0.f, dn = 0.f, dno = 0.f;
  fetch_stats(0, ln, dn);
  if constexpr (OWN) dno = own_dsum(0, gn);
  float kf[16], vf[16];
  load_row_slots(kf, base + a.E, ld, ck, 1.f, kh);
  load_row_slots(vf, base + 2 * a.E, ld, ck, 1.f, kh);
  f16v dkT, dvT;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dkT[r] = 0.f; dvT[r] = 0.f; }
  tile_put(Qs[0], qn, qmul, tid);
  tile_put(Gs[0], gn, 1.f, tid);
  if (tid < 32) { ls[0][tid] = ln; if constexpr (!OWN) dsm[0][tid] = dn; }
  if constexpr (OWN) { if ((tid & 7) == 0) dsm[0][tid >> 3] = dno; }
  __syncthreads();
  const uint32_t koff = (uint32_t)mykey + (uint32_t)(4 * kh) * (uint32_t)L;      // this lane's share of the element index
  const HKey hkey = hash_key(a.seed);
  const float inv_keep = 1.f / (1.f - a.p);
  const int nt = (L + 31) >> 5;
  for (int t = 0; t < nt; ++t) {
    const int q0 = t * 32, cur = t & 1;
    if (t + 1 < nt) {
      qn = tile_fetch(base, ld, q0 + 32, L, tid);
      gn = tile_fetch(dOb, a.E, q0 + 32, L, tid);
      fetch_stats(q0 + 32, ln, dn);
      if constexpr (OWN) dno = own_dsum(q0 + 32, gn);
    }
    const float* Q = Qs[cur];
    const float* G = Gs[cur];
    f16v s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int kp = 0; kp < 16; ++kp) {
      s = MFMA32(Q[l31 * LD + 2 * kp + kh], kf[kp], s);         // S[q][key] (times log2 e)
      dp = MFMA32(G[l31 * LD + 2 * kp + kh], vf[kp], dp);       // dPd[q][key] = dO V^T
    }
    float pd[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qr = mrow(r, kh);
      const float pv = ex2(s[r] - ls[cur][qr]);
      float sc = 1.f;
      if constexpr (DROP) {      // element (bh, q, key) of the [B heads, L, L] probabilities: wave-uniform part + the lane's
        const uint64_t rb = (uint64_t)((long)bh * L + q0 + (mrow(r, 0))) * (uint64_t)L;
        const uint32_t lo = (uint32_t)rb + koff;
        sc = dropout_scale_fast(hkey, lo, (uint32_t)(rb >> 32) + (lo < (uint32_t)rb), a.p, inv_keep);
      }
      pd[r] = pv * sc;
      ds[r] = pv * (dp[r] * sc - dsm[cur][qr]);
    }
#pragma unroll
    for (int kp = 0; kp < 16; ++kp) {
      dvT = MFMA32(G[mrow(kp, kh) * LD + l31], pd[kp], dvT);    // dV^T[d][key] += dO^T Pd
      dkT = MFMA32(Q[mrow(kp, kh) * LD + l31], ds[kp], dkT);    // dK^T[d][key] += (scale log2 e Q)^T dS
    }
    if (t + 1 < nt) {
      tile_put(Qs[cur ^ 1], qn, qmul, tid);
      tile_put(Gs[cur ^ 1], gn, 1.f, tid);
      if (tid < 32) { ls[cur ^ 1][tid] = ln; if constexpr (!OWN) dsm[cur ^ 1][tid] = dn; }
      if constexpr (OWN) { if ((tid & 7) == 0) dsm[cur ^ 1][tid >> 3] = dno; }
    }
    __syncthreads();
  }
  if (mykey < L) {
    float* dk = a.dqkv + ((long)b * L + mykey) * ld + a.E + h * HD;
    store_row16(dk, dkT, LN2, kh);          // (the staged queries carried log2 e)
    store_row16(dk + a.E, dvT, 1.f, kh);
  }
  if (a.dbias) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float vk = mykey < L ? dkT[r] * LN2 : 0.f, vv = mykey < L ? dvT[r] : 0.f;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) { vk += __shfl_xor(vk, o, 64); vv += __shfl_xor(vv, o, 64); }
      if (l31 == 0) {
        atomicAdd(a.dbias + a.E + h * HD + mrow(r, kh), vk);
        atomicAdd(a.dbias + 2 * a.E + h * HD + mrow(r, kh), vv);
      }
    }
  }
}

template <bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_q_k(AttnArgs a) { attn_bwd_q_body<DROP>(a); }
template <bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_kv_k(AttnArgs a) { attn_bwd_kv_body<DROP, false>(a); }
// Round 6: both passes as ONE launch (blockIdx.z = pass).  Each pass alone is 384 workgroups of four waves at two waves per SIMD: 1 536
// wave-tasks on 1 024 SIMDs, so half of the CUs hold two workgroups, run them at half speed each, and the pass lasts two workgroup-times
// (profiles/r06_attention_backward_pipelining.txt); two passes = four.  768 workgroups in one grid fill 512 slots and hand the freed ones
// to the rest: three workgroup-times.  The dQ workgroups come first in the dispatch order (z = 0).
template <bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_k(AttnArgs a) {
  if (blockIdx.z == 0) attn_bwd_q_body<DROP>(a);
  else attn_bwd_kv_body<DROP, true>(a);
}

}  // namespace

int g_attn_bwd_one_launch = 1;  // zeggs_set_option("attn_bwd_one_launch", 0/1): A/B switch of the merged backward launch
int g_fused_attention = 1;      // zeggs_set_option("fused_attention", 0/1); 0 = GEMM + softmax + GEMM over [B*heads, L, L] matrices

int attn_fused_supported(int E, int NH) { return g_fused_attention && NH > 0 && E % NH == 0 && E / NH == HD && E % 4 == 0; }

int k_attn_fwd(const float* qkv, float* O, float* lse, int B, int L, int E, int NH, float p, uint64_t seed, hipStream_t s) {
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.qkv = qkv; a.O = O; a.lse = lse; a.L = L; a.E = E; a.NH = NH; a.scale = 1.0f / sqrtf((float)HD); a.p = p; a.seed = seed;
  if (p > 0.f) hipLaunchKernelGGL(attn_fwd_k<true>, dim3(cdiv(L, WQ), B * NH), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(attn_fwd_k<false>, dim3(cdiv(L, WQ), B * NH), dim3(256), 0, s, a);
  ZLAUNCH_CHECK("attn_fwd");
  return 0;
}
int k_attn_bwd(const float* qkv, const float* O, const float* lse, const float* dO, float* dqkv, float* dsum, float* dbias,
               int B, int L, int E, int NH, float p, uint64_t seed, hipStream_t s) {
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.qkv = qkv; a.O = (float*)O; a.lse = (float*)lse; a.dO = dO; a.dqkv = dqkv; a.dsum = dsum; a.dbias = dbias;
  a.L = L; a.E = E; a.NH = NH; a.scale = 1.0f / sqrtf((float)HD); a.p = p; a.seed = seed;
  if (g_attn_bwd_one_launch) {
    if (p > 0.f) hipLaunchKernelGGL(attn_bwd_k<true>, dim3(cdiv(L, WQ), B * NH, 2), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(attn_bwd_k<false>, dim3(cdiv(L, WQ), B * NH, 2), dim3(256), 0, s, a);
    ZLAUNCH_CHECK("attn_bwd");
    return 0;
  }
  if (p > 0.f) hipLaunchKernelGGL(attn_bwd_q_k<true>, dim3(cdiv(L, WQ), B * NH), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(attn_bwd_q_k<false>, dim3(cdiv(L, WQ), B * NH), dim3(256), 0, s, a);
  ZLAUNCH_CHECK("attn_bwd_q");
  if (p > 0.f) hipLaunchKernelGGL(attn_bwd_kv_k<true>, dim3(cdiv(L, WQ), B * NH), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(attn_bwd_kv_k<false>, dim3(cdiv(L, WQ), B * NH), dim3(256), 0, s, a);
  ZLAUNCH_CHECK("attn_bwd_kv");
  return 0;
}
