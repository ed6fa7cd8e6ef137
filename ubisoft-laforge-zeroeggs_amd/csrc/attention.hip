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

struct AttnArgs {
  const float* qkv;      // [B*L, 3E]: q | k | v, head h at columns h*32
  float* O;              // [B*L, E]   attention output (before out_proj), head h at columns h*32
  float* lse;            // [B*NH, L]  row log-sum-exp of the scaled scores
  const float* dO;       // backward
  float* dqkv;           // [B*L, 3E]
  float* dbias;          // != null: [3E] += column sums of dqkv (the in-projection's bias gradient), atomics
  float* dsum;           // [B*NH, L]  rowsum(dO . O)
  int L, E, NH;
  float scale, p;
  uint64_t seed;
};

__device__ __forceinline__ int mrow(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }

// 32 rows x 32 floats of a [rows, ld] matrix -> LDS tile [32][LD]; 256 threads, one float4 each; rows clamped to nrows - 1
__device__ __forceinline__ void stage32(float* dst, const float* src, long ld, int row0, int nrows, float mul, int tid) {
  const int r = tid >> 3, c4 = (tid & 7) * 4;
  int row = row0 + r;
  row = row < nrows ? row : nrows - 1;
  const f4 v = *(const f4*)(src + (long)row * ld + c4);
  float* d = dst + r * LD + c4;
  d[0] = v[0] * mul; d[1] = v[1] * mul; d[2] = v[2] * mul; d[3] = v[3] * mul;
}

__device__ __forceinline__ float half_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float half_sum(float v) { return v + __shfl_xor(v, 32, 64); }

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// ------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void attn_fwd_k(AttnArgs a) {
  __shared__ float Qs[WQ * LD], Ks[32 * LD], Vs[32 * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.NH, h = bh % a.NH, L = a.L;
  const long ld = 3L * a.E;
  const float* base = a.qkv + (long)b * L * ld + h * HD;
  for (int j = 0; j < 4; ++j) stage32(Qs + j * 32 * LD, base, ld, blockIdx.x * WQ + j * 32, L, a.scale, tid);
  __syncthreads();
  float qf[16];
#pragma unroll
  for (int kp = 0; kp < 16; ++kp) qf[kp] = Qs[(wave * 32 + l31) * LD + 2 * kp + kh];
  const int myq = blockIdx.x * WQ + wave * 32 + l31;
  const long prow = ((long)bh * L + myq) * L;          // element index of P[bh][myq][0]
  f16v o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m = -INFINITY, lsum = 0.f;
  for (int k0 = 0; k0 < L; k0 += 32) {
    __syncthreads();
    stage32(Ks, base + a.E, ld, k0, L, 1.f, tid);
    stage32(Vs, base + 2 * a.E, ld, k0, L, 1.f, tid);
    __syncthreads();
    f16v s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int kp = 0; kp < 16; ++kp) s = MFMA32(Ks[l31 * LD + 2 * kp + kh], qf[kp], s);      // S^T[key][q]
    float mx = m;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (k0 + mrow(r, kh) < L) mx = fmaxf(mx, s[r]);
    mx = half_max(mx);
    const float corr = expf(m - mx);          // first tile: exp(-inf) = 0
    float ps = 0.f, pd[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + mrow(r, kh);
      const float pv = key < L ? expf(s[r] - mx) : 0.f;
      ps += pv;
      pd[r] = pv * dropout_scale(a.seed, (uint64_t)(prow + key), a.p);
    }
    ps = half_sum(ps);
    lsum = lsum * corr + ps;
    m = mx;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= corr;
#pragma unroll
    for (int kp = 0; kp < 16; ++kp) o = MFMA32(Vs[mrow(kp, kh) * LD + l31], pd[kp], o);      // O^T[d][q] += V^T P^T
  }
  if (myq < L) {
    const float inv = 1.f / lsum;
    float* op = a.O + ((long)b * L + myq) * a.E + h * HD;
#pragma unroll
    for (int r = 0; r < 16; ++r) op[mrow(r, kh)] = o[r] * inv;
    if (kh == 0) a.lse[(long)bh * L + myq] = m + logf(lsum);
  }
}

// ------------------------------------------------------------------ backward: dQ (and dsum = rowsum(dO . O))
__global__ __launch_bounds__(256) void attn_bwd_q_k(AttnArgs a) {
  __shared__ float Ts[WQ * LD], Ks[32 * LD], Vs[32 * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.NH, h = bh % a.NH, L = a.L;
  const long ld = 3L * a.E;
  const float* base = a.qkv + (long)b * L * ld + h * HD;
  const int qw = blockIdx.x * WQ;
  const int myq = qw + wave * 32 + l31, cq = myq < L ? myq : L - 1;
  float qf[16], dof[16];
  // the three per-row operands of this wave's queries, one after the other through the same LDS tile
  for (int j = 0; j < 4; ++j) stage32(Ts + j * 32 * LD, base, ld, qw + j * 32, L, a.scale, tid);
  __syncthreads();
#pragma unroll
  for (int kp = 0; kp < 16; ++kp) qf[kp] = Ts[(wave * 32 + l31) * LD + 2 * kp + kh];
  __syncthreads();
  for (int j = 0; j < 4; ++j) stage32(Ts + j * 32 * LD, a.dO + (long)b * L * a.E + h * HD, a.E, qw + j * 32, L, 1.f, tid);
  __syncthreads();
#pragma unroll
  for (int kp = 0; kp < 16; ++kp) dof[kp] = Ts[(wave * 32 + l31) * LD + 2 * kp + kh];
  __syncthreads();
  for (int j = 0; j < 4; ++j) stage32(Ts + j * 32 * LD, a.O + (long)b * L * a.E + h * HD, a.E, qw + j * 32, L, 1.f, tid);
  __syncthreads();
  float dq_ = 0.f;
#pragma unroll
  for (int kp = 0; kp < 16; ++kp) dq_ += dof[kp] * Ts[(wave * 32 + l31) * LD + 2 * kp + kh];
  const float Dq = half_sum(dq_);
  const float lse = a.lse[(long)bh * L + cq];
  const long prow = ((long)bh * L + myq) * L;
  f16v dqT;
#pragma unroll
  for (int r = 0; r < 16; ++r) dqT[r] = 0.f;
  for (int k0 = 0; k0 < L; k0 += 32) {
    __syncthreads();
    stage32(Ks, base + a.E, ld, k0, L, 1.f, tid);
    stage32(Vs, base + 2 * a.E, ld, k0, L, 1.f, tid);
    __syncthreads();
    f16v s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int kp = 0; kp < 16; ++kp) {
      s = MFMA32(Ks[l31 * LD + 2 * kp + kh], qf[kp], s);         // S^T[key][q]
      dp = MFMA32(Vs[l31 * LD + 2 * kp + kh], dof[kp], dp);      // dPd^T[key][q] = V dO^T
    }
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + mrow(r, kh);
      const float pv = (key < L && myq < L) ? expf(s[r] - lse) : 0.f;
      ds[r] = pv * (dp[r] * dropout_scale(a.seed, (uint64_t)(prow + key), a.p) - Dq);
    }
#pragma unroll
    for (int kp = 0; kp < 16; ++kp) dqT = MFMA32(Ks[mrow(kp, kh) * LD + l31], ds[kp], dqT);  // dQ^T[d][q] += K^T dS^T
  }
  if (myq < L) {
    float* dq = a.dqkv + ((long)b * L + myq) * ld + h * HD;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[mrow(r, kh)] = dqT[r] * a.scale;
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
__global__ __launch_bounds__(256) void attn_bwd_kv_k(AttnArgs a) {
  __shared__ float Ts[WQ * LD], Qs[32 * LD], Gs[32 * LD], ls[32], dsm[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.NH, h = bh % a.NH, L = a.L;
  const long ld = 3L * a.E;
  const float* base = a.qkv + (long)b * L * ld + h * HD;
  const int kw = blockIdx.x * WQ;
  const int mykey = kw + wave * 32 + l31;
  float kf[16], vf[16];
  for (int j = 0; j < 4; ++j) stage32(Ts + j * 32 * LD, base + a.E, ld, kw + j * 32, L, 1.f, tid);
  __syncthreads();
#pragma unroll
  for (int kp = 0; kp < 16; ++kp) kf[kp] = Ts[(wave * 32 + l31) * LD + 2 * kp + kh];
  __syncthreads();
  for (int j = 0; j < 4; ++j) stage32(Ts + j * 32 * LD, base + 2 * a.E, ld, kw + j * 32, L, 1.f, tid);
  __syncthreads();
#pragma unroll
  for (int kp = 0; kp < 16; ++kp) vf[kp] = Ts[(wave * 32 + l31) * LD + 2 * kp + kh];
  f16v dkT, dvT;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dkT[r] = 0.f; dvT[r] = 0.f; }
  for (int q0 = 0; q0 < L; q0 += 32) {
    __syncthreads();
    stage32(Qs, base, ld, q0, L, a.scale, tid);
    stage32(Gs, a.dO + (long)b * L * a.E + h * HD, a.E, q0, L, 1.f, tid);
    if (tid < 32) {
      const int q = q0 + tid < L ? q0 + tid : L - 1;
      ls[tid] = a.lse[(long)bh * L + q];
      dsm[tid] = a.dsum[(long)bh * L + q];
    }
    __syncthreads();
    f16v s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int kp = 0; kp < 16; ++kp) {
      s = MFMA32(Qs[l31 * LD + 2 * kp + kh], kf[kp], s);         // S[q][key]
      dp = MFMA32(Gs[l31 * LD + 2 * kp + kh], vf[kp], dp);       // dPd[q][key] = dO V^T
    }
    float pd[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qr = mrow(r, kh), q = q0 + qr;
      const float pv = (q < L && mykey < L) ? expf(s[r] - ls[qr]) : 0.f;
      const float sc = dropout_scale(a.seed, (uint64_t)(((long)bh * L + q) * L + mykey), a.p);
      pd[r] = pv * sc;
      ds[r] = pv * (dp[r] * sc - dsm[qr]);
    }
#pragma unroll
    for (int kp = 0; kp < 16; ++kp) {
      dvT = MFMA32(Gs[mrow(kp, kh) * LD + l31], pd[kp], dvT);    // dV^T[d][key] += dO^T Pd
      dkT = MFMA32(Qs[mrow(kp, kh) * LD + l31], ds[kp], dkT);    // dK^T[d][key] += (scale Q)^T dS
    }
  }
  if (mykey < L) {
    float* dk = a.dqkv + ((long)b * L + mykey) * ld + a.E + h * HD;
    float* dv = dk + a.E;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[mrow(r, kh)] = dkT[r]; dv[mrow(r, kh)] = dvT[r]; }
  }
  if (a.dbias) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float vk = mykey < L ? dkT[r] : 0.f, vv = mykey < L ? dvT[r] : 0.f;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) { vk += __shfl_xor(vk, o, 64); vv += __shfl_xor(vv, o, 64); }
      if (l31 == 0) {
        atomicAdd(a.dbias + a.E + h * HD + mrow(r, kh), vk);
        atomicAdd(a.dbias + 2 * a.E + h * HD + mrow(r, kh), vv);
      }
    }
  }
}

}  // namespace

int g_fused_attention = 1;      // zeggs_set_option("fused_attention", 0/1); 0 = GEMM + softmax + GEMM over [B*heads, L, L] matrices

int attn_fused_supported(int E, int NH) { return g_fused_attention && NH > 0 && E % NH == 0 && E / NH == HD && E % 4 == 0; }

int k_attn_fwd(const float* qkv, float* O, float* lse, int B, int L, int E, int NH, float p, uint64_t seed, hipStream_t s) {
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.qkv = qkv; a.O = O; a.lse = lse; a.L = L; a.E = E; a.NH = NH; a.scale = 1.0f / sqrtf((float)HD); a.p = p; a.seed = seed;
  hipLaunchKernelGGL(attn_fwd_k, dim3(cdiv(L, WQ), B * NH), dim3(256), 0, s, a);
  ZLAUNCH_CHECK("attn_fwd");
  return 0;
}
int k_attn_bwd(const float* qkv, const float* O, const float* lse, const float* dO, float* dqkv, float* dsum, float* dbias,
               int B, int L, int E, int NH, float p, uint64_t seed, hipStream_t s) {
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.qkv = qkv; a.O = (float*)O; a.lse = (float*)lse; a.dO = dO; a.dqkv = dqkv; a.dsum = dsum; a.dbias = dbias;
  a.L = L; a.E = E; a.NH = NH; a.scale = 1.0f / sqrtf((float)HD); a.p = p; a.seed = seed;
  hipLaunchKernelGGL(attn_bwd_q_k, dim3(cdiv(L, WQ), B * NH), dim3(256), 0, s, a);
  ZLAUNCH_CHECK("attn_bwd_q");
  hipLaunchKernelGGL(attn_bwd_kv_k, dim3(cdiv(L, WQ), B * NH), dim3(256), 0, s, a);
  ZLAUNCH_CHECK("attn_bwd_kv");
  return 0;
}
