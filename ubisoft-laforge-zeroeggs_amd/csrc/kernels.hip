// Streaming kernels shared by the encoders (HBM-bound: coalesced along the
// feature axis, one wave per row for row reductions, grid-stride elsewhere).
#include "common.h"
#include "kernels.h"

namespace {

inline int grid1d(long n, int block = 256) {
  long g = (n + block - 1) / block;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}
#define GS_LOOP(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

__device__ __forceinline__ float* rv_row(const RowView& v, long r, int C) {
  if (v.rpb == 0) return v.p + r * C;
  const long b = r / v.rpb;
  return v.p + b * v.bstride + (r - b * v.rpb) * C;
}

__global__ void fill_k(float* p, long n, float v) { GS_LOOP(i, n) p[i] = v; }
// 16-byte stores (p 16-byte aligned, n4 = n / 4 float4s): the zero fills in front of the split / stream-K products and of the flat
// gradient buffer sit on the serial chains of the iteration
__global__ void fill4_k(f4* p, long n4, float v) { const f4 vv = f4{v, v, v, v}; GS_LOOP(i, n4) p[i] = vv; }
__global__ void copy_k(float* d, const float* s, long n) { GS_LOOP(i, n) d[i] = s[i]; }
__global__ void add_k(float* d, const float* s, long n) { GS_LOOP(i, n) d[i] += s[i]; }

__global__ void act_bwd_k(float* dx, const float* dy, const float* y, long n, int act, float ys) {
  GS_LOOP(i, n) {
    float yv = y[i] * ys, g = dy[i];
    if (act == ACT_ELU) g *= d_elu_grad_from_out(yv);
    else if (act == ACT_RELU) g = yv > 0.f ? g : 0.f;
    dx[i] = g;
  }
}

__global__ void act_bwd_v_k(RowView dx, RowView dy, RowView y, long R, int C, int act, float ys) {
  long n = R * C;
  GS_LOOP(i, n) {
    long r = i / C;
    int c = (int)(i - r * C);
    float yv = rv_row(y, r, C)[c] * ys, g = rv_row(dy, r, C)[c];
    if (act == ACT_ELU) g *= d_elu_grad_from_out(yv);
    else if (act == ACT_RELU) g = yv > 0.f ? g : 0.f;
    rv_row(dx, r, C)[c] = g;
  }
}

__global__ void dropout_k(float* x, long n, float p, uint64_t seed) {
  GS_LOOP(i, n) x[i] *= dropout_scale(seed, (uint64_t)i, p);
}

__global__ void dropout_rows_k(float* x, int rows, int cols, long ld, long batch_rows, long batch_stride,
                               float p, uint64_t seed) {
  long n = (long)rows * cols;
  GS_LOOP(i, n) {
    long r = i / cols;
    int c = (int)(i - r * cols);
    long b = r / batch_rows, rr = r - b * batch_rows;
    x[b * batch_stride + rr * ld + c] *= dropout_scale(seed, (uint64_t)i, p);
  }
}

// one wave per row
__global__ __launch_bounds__(256) void layernorm_fwd_k(RowView y, RowView x, RowView res, const float* gamma,
                                                        const float* beta, float* mean, float* rstd, int R, int C,
                                                        float eps) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  const float* xr = rv_row(x, row, C);
  const float* rr = res.p ? rv_row(res, row, C) : nullptr;
  float* yr = rv_row(y, row, C);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c] + (rr ? rr[c] : 0.f);
  float mu = wave_sum(s) / C;
  float v = 0.f;
  for (int c = lane; c < C; c += 64) {
    float d = xr[c] + (rr ? rr[c] : 0.f) - mu;
    v += d * d;
  }
  float rs = 1.0f / sqrtf(wave_sum(v) / C + eps);   // biased variance, as torch
  for (int c = lane; c < C; c += 64) {
    float d = xr[c] + (rr ? rr[c] : 0.f) - mu;
    yr[c] = d * rs * gamma[c] + beta[c];
  }
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

template <int MAXJ>
__global__ __launch_bounds__(256) void layernorm_bwd_k(RowView dxv, RowView dyv, RowView xv, RowView resv,
                                                        const float* gamma, const float* mean, const float* rstd,
                                                        float* dgamma, float* dbeta, int R, int C,
                                                        int rows_per_block) {
  __shared__ float sg[4][64 * MAXJ], sb[4][64 * MAXJ];
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float pg[MAXJ], pb[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) pg[j] = pb[j] = 0.f;
  int r0 = blockIdx.x * rows_per_block;
  for (int row = r0 + wave; row < r0 + rows_per_block && row < R; row += 4) {
    const float mu = mean[row], rs = rstd[row];
    const float* x = rv_row(xv, row, C);
    const float* res = resv.p ? rv_row(resv, row, C) : nullptr;
    const float* dy = rv_row(dyv, row, C);
    float* dx = rv_row(dxv, row, C);
    float xh[MAXJ], g[MAXJ];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      int c = lane + 64 * j;
      xh[j] = 0.f; g[j] = 0.f;
      if (c < C) {
        float xval = x[c] + (res ? res[c] : 0.f);
        float d = dy[c];
        xh[j] = (xval - mu) * rs;
        g[j] = d * gamma[c];
        pg[j] += d * xh[j];
        pb[j] += d;
        s1 += g[j];
        s2 += g[j] * xh[j];
      }
    }
    s1 = wave_sum(s1) / C;
    s2 = wave_sum(s2) / C;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      int c = lane + 64 * j;
      if (c < C) dx[c] = rs * (g[j] - s1 - xh[j] * s2);
    }
  }
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) { sg[wave][lane + 64 * j] = pg[j]; sb[wave][lane + 64 * j] = pb[j]; }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    atomicAdd(dgamma + c, sg[0][c] + sg[1][c] + sg[2][c] + sg[3][c]);
    atomicAdd(dbeta + c, sb[0][c] + sb[1][c] + sb[2][c] + sb[3][c]);
  }
}

// see kernels.h (LnFwdFused): one wave per row, the row in registers
template <int MAXJ>
__global__ __launch_bounds__(256) void ln_fwd_fused_k(LnFwdFused a) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, C = a.C;
  if (row >= a.R) return;
  float* xr = rv_row(a.x, row, C);
  const float* rr = a.res.p ? rv_row(a.res, row, C) : nullptr;
  float* yr = rv_row(a.y, row, C);
  float v[MAXJ];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int c = lane + 64 * j;
    v[j] = 0.f;
    if (c < C) {
      float xv = xr[c];
      if (a.pre_bias) xv = d_act(xv + a.pre_bias[c], a.pre_act);
      if (a.p_pre > 0.f) xv *= dropout_scale(a.seed_pre, (uint64_t)((long)row * C + c), a.p_pre);
      if (a.pre_bias || a.p_pre > 0.f) xr[c] = xv;
      v[j] = xv + (rr ? rr[c] : 0.f);
      s += v[j];
    }
  }
  const float mu = wave_sum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int c = lane + 64 * j;
    if (c < C) { const float d = v[j] - mu; q += d * d; }
  }
  const float rs = 1.0f / sqrtf(wave_sum(q) / C + a.eps);   // biased variance, as torch
  const float* tb = a.table ? a.table + (long)(row % a.table_L) * C : nullptr;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int c = lane + 64 * j;
    if (c < C) {
      float o = (v[j] - mu) * rs * a.gamma[c] + a.beta[c];
      o *= dropout_scale(a.seed_post, (uint64_t)((long)row * C + c), a.p_post);
      if (tb) o += tb[c];
      yr[c] = o;
      if (a.pad_L > 0) {
        const int l = row % a.pad_L;
        if (l == 0) yr[c - C] = 0.f;
        if (l == a.pad_L - 1) yr[c + C] = 0.f;
      }
    }
  }
  if (lane == 0) { a.mean[row] = mu; a.rstd[row] = rs; }
}

// The same pass with 16-byte lanes (round 5; as ln_bwd_fused4_k below): a row is LPR lanes x VJ float4, 64 / LPR rows per wave.
// C % 4 == 0, every row view 16-byte aligned (k_ln_fwd_fused checks; gamma / beta / table / pre_bias are read 4 bytes at a time:
// parameter views are only 4-byte aligned).
template <int LPR, int VJ>
__global__ __launch_bounds__(256) void ln_fwd_fused4_k(LnFwdFused a) {
  constexpr int RPW = 64 / LPR, NE = 4 * VJ;
  const int tid = threadIdx.x, lane = tid & 63, sub = lane / LPR, l = lane % LPR, C = a.C;
  const int row = (blockIdx.x * 4 + (tid >> 6)) * RPW + sub;
  const bool rok = row < a.R;
  const unsigned rr = rok ? row : a.R - 1;
  float* xr = a.x.rpb == 0 ? a.x.p + (size_t)rr * C : a.x.p + (size_t)(rr / (unsigned)a.x.rpb) * a.x.bstride + (size_t)(rr % (unsigned)a.x.rpb) * C;
  const float* rp = !a.res.p ? nullptr : a.res.rpb == 0 ? a.res.p + (size_t)rr * C
                                       : a.res.p + (size_t)(rr / (unsigned)a.res.rpb) * a.res.bstride + (size_t)(rr % (unsigned)a.res.rpb) * C;
  float* yr = a.y.rpb == 0 ? a.y.p + (size_t)rr * C : a.y.p + (size_t)(rr / (unsigned)a.y.rpb) * a.y.bstride + (size_t)(rr % (unsigned)a.y.rpb) * C;
  float v[NE];
  float s = 0.f;
  const bool wb = a.pre_bias != nullptr || a.p_pre > 0.f;
#pragma unroll
  for (int j = 0; j < VJ; ++j) {
    const int c0 = 4 * (l + LPR * j);
    const bool ok = c0 < C;
    f4 xv = ok ? *(const f4*)(xr + c0) : f4{0.f, 0.f, 0.f, 0.f};
    const f4 rv4 = (ok && rp) ? *(const f4*)(rp + c0) : f4{0.f, 0.f, 0.f, 0.f};
    if (ok) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (a.pre_bias) xv[e] = d_act(xv[e] + a.pre_bias[c0 + e], a.pre_act);
        if (a.p_pre > 0.f) xv[e] *= dropout_scale(a.seed_pre, (uint64_t)((size_t)rr * C + c0 + e), a.p_pre);
      }
      if (wb && rok) *(f4*)(xr + c0) = xv;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[4 * j + e] = ok ? xv[e] + rv4[e] : 0.f; s += v[4 * j + e]; }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mu = s / C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < VJ; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * (l + LPR * j) < C) { const float d = v[4 * j + e] - mu; q += d * d; }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rs = 1.0f / sqrtf(q / C + a.eps);   // biased variance, as torch
  const float* tb = a.table ? a.table + (size_t)(rr % (unsigned)a.table_L) * C : nullptr;
  const int pl = a.pad_L > 0 ? (int)(rr % (unsigned)a.pad_L) : -2;
  if (rok) {
#pragma unroll
    for (int j = 0; j < VJ; ++j) {
      const int c0 = 4 * (l + LPR * j);
      if (c0 < C) {
        f4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = (v[4 * j + e] - mu) * rs * a.gamma[c0 + e] + a.beta[c0 + e];
          t *= dropout_scale(a.seed_post, (uint64_t)((size_t)rr * C + c0 + e), a.p_post);
          if (tb) t += tb[c0 + e];
          o[e] = t;
        }
        *(f4*)(yr + c0) = o;
        if (pl == 0) *(f4*)(yr + c0 - C) = f4{0.f, 0.f, 0.f, 0.f};
        if (pl == a.pad_L - 1) *(f4*)(yr + c0 + C) = f4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (l == 0) { a.mean[rr] = mu; a.rstd[rr] = rs; }
  }
}

// see kernels.h (LnBwdFused).  One wave per row, 16 rows per block (as layernorm_bwd_k)
template <int MAXJ>
__global__ __launch_bounds__(256) void ln_bwd_fused_k(LnBwdFused a, int rows_per_block) {
  __shared__ float sg[4][64 * MAXJ], sb[4][64 * MAXJ], sc[4][64 * MAXJ];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, C = a.C, R = a.R;
  const bool ln = a.gamma != nullptr;
  float pg[MAXJ], pb[MAXJ], pc[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) pg[j] = pb[j] = pc[j] = 0.f;
  const int r0 = blockIdx.x * rows_per_block;
  for (int row = r0 + wave; row < r0 + rows_per_block && row < R; row += 4) {
    const float mu = ln ? a.mean[row] : 0.f, rs = ln ? a.rstd[row] : 1.f;
    const float* x = ln ? rv_row(a.x, row, C) : nullptr;
    const float* res = (ln && a.res.p) ? rv_row(a.res, row, C) : nullptr;
    const float* dyA = a.dy_pool ? a.dy_pool + (long)(row / a.pool_L) * C : rv_row(a.dyA, row, C);
    const float* dyB = a.dyB.p ? rv_row(a.dyB, row, C) : nullptr;
    const float dscale = a.dy_pool ? 1.f / a.pool_L : 1.f;
    float xh[MAXJ], g[MAXJ], d[MAXJ];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = lane + 64 * j;
      xh[j] = 0.f; g[j] = 0.f; d[j] = 0.f;
      if (c < C) {
        float dv = dyA[c] * dscale + (dyB ? dyB[c] : 0.f);
        dv *= dropout_scale(a.seed_pre, (uint64_t)((long)row * C + c), a.p_pre);
        d[j] = dv;
        if (ln) {
          const float xval = x[c] + (res ? res[c] : 0.f);
          xh[j] = (xval - mu) * rs;
          g[j] = dv * a.gamma[c];
          pg[j] += dv * xh[j];
          pb[j] += dv;
          s1 += g[j];
          s2 += g[j] * xh[j];
        }
      }
    }
    if (ln) {
      s1 = wave_sum(s1) / C;
      s2 = wave_sum(s2) / C;
    }
    float* outr = rv_row(a.out, row, C);
    float* rawr = a.dx_raw.p ? rv_row(a.dx_raw, row, C) : nullptr;
    const float* ys = a.ysave.p ? rv_row(a.ysave, row, C) : nullptr;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = lane + 64 * j;
      if (c < C) {
        const float dx = ln ? rs * (g[j] - s1 - xh[j] * s2) : d[j];
        if (rawr) rawr[c] = dx;
        float o = dx * dropout_scale(a.seed_post, (uint64_t)((long)row * C + c), a.p_post);
        if (ys) {
          const float yv = ys[c];
          o = a.act == ACT_ELU ? o * d_elu_grad_from_out(yv) : a.act == ACT_RELU ? (yv > 0.f ? o : 0.f) : o;
        }
        outr[c] = o;
        pc[j] += o;
        if (a.pad_L > 0) {       // edge rows of the padded buffer this row sits in
          const int l = row % a.pad_L;
          if (l == 0) outr[c - C] = 0.f;
          if (l == a.pad_L - 1) outr[c + C] = 0.f;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) { sg[wave][lane + 64 * j] = pg[j]; sb[wave][lane + 64 * j] = pb[j]; sc[wave][lane + 64 * j] = pc[j]; }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    if (ln) {
      atomicAdd(a.dgamma + c, sg[0][c] + sg[1][c] + sg[2][c] + sg[3][c]);
      atomicAdd(a.dbeta + c, sb[0][c] + sb[1][c] + sb[2][c] + sb[3][c]);
    }
    if (a.dbias) atomicAdd(a.dbias + c, sc[0][c] + sc[1][c] + sc[2][c] + sc[3][c]);
  }
}

// Round 5: the same pass with 16-byte lanes.  The kernel above gives a wave one row at a time with 4-byte lanes (C = 128: two
// floats per lane, five dependent row visits per wave, a 64-bit division per operand view and row) and ends every block of 16 rows
// in 3 C atomics: 121 us alone for 12 288 x 128 -- 44 MB that the memory system moves in ~7 us.  Here a row is LPR lanes x VJ
// float4 (C <= 128: half a wave per row, two rows per wave and pass; C <= 256: a wave, one float4; C <= 512: a wave, two), a block
// walks its rows in passes of 4 * (64 / LPR) rows with everything of a pass in flight at once, the row statistics are reduced
// inside the LPR lanes, and the column sums (dgamma, dbeta, dbias) stay in registers over the passes, meet in LDS once and
// leave as one atomic per column and block (grid ~ 2 blocks per CU).  Row views by 32-bit arithmetic.
__device__ __forceinline__ float* rv_row32(const RowView& v, unsigned r, int C) {
  if (v.rpb == 0) return v.p + (size_t)r * C;
  const unsigned b = r / (unsigned)v.rpb;
  return v.p + (size_t)b * v.bstride + (size_t)(r - b * (unsigned)v.rpb) * C;
}
template <int LPR, int VJ, int NW>
__global__ __launch_bounds__(64 * NW) void ln_bwd_fused4_k(LnBwdFused a, int rows_per_block) {
  constexpr int RPW = 64 / LPR, RPP = NW * RPW, NE = 4 * VJ, CW = LPR * NE;      // rows per wave / per pass, floats per lane, padded width
  __shared__ float sred[3][RPP][CW];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, sub = lane / LPR, l = lane % LPR, slot = wave * RPW + sub;
  const int C = a.C, R = a.R;
  const bool ln = a.gamma != nullptr;
  float pg[NE], pb[NE], pc[NE], gam[NE];
  bool cok[VJ];
#pragma unroll
  for (int j = 0; j < VJ; ++j) {
    cok[j] = 4 * (l + LPR * j) < C;
#pragma unroll
    for (int e = 0; e < 4; ++e) {      // (gamma is a view into the flat parameter buffer: 4-byte aligned only)
      gam[4 * j + e] = (ln && cok[j]) ? a.gamma[4 * (l + LPR * j) + e] : 0.f;
      pg[4 * j + e] = pb[4 * j + e] = pc[4 * j + e] = 0.f;
    }
  }
  const float invC = 1.f / C, dscale = a.dy_pool ? 1.f / a.pool_L : 1.f;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, R);
  // the operands of the NEXT pass are fetched before this pass's reductions and stores (two passes of loads in flight per wave:
  // the pass is a chain load -> row reduction -> store, and 8 waves per CU alone do not cover its latency)
  struct Pass { f4 va[VJ], vb[VJ], vx[VJ], vr[VJ], vy[VJ]; float mu, rs; unsigned rr; bool rok; };
  auto fetch = [&](int rb, Pass& P) {
    const int row = rb + slot;
    P.rok = row < r1;
    P.rr = P.rok ? row : r1 - 1;          // (idle slots shadow the block's last row: loads stay in bounds, nothing is stored / summed)
    const unsigned rr = P.rr;
    P.mu = ln ? a.mean[rr] : 0.f; P.rs = ln ? a.rstd[rr] : 1.f;
    const float* x = ln ? rv_row32(a.x, rr, C) : nullptr;
    const float* res = (ln && a.res.p) ? rv_row32(a.res, rr, C) : nullptr;
    const float* dyA = a.dy_pool ? a.dy_pool + (size_t)(rr / (unsigned)a.pool_L) * C : rv_row32(a.dyA, rr, C);
    const float* dyB = a.dyB.p ? rv_row32(a.dyB, rr, C) : nullptr;
    const float* ys = a.ysave.p ? rv_row32(a.ysave, rr, C) : nullptr;
#pragma unroll
    for (int j = 0; j < VJ; ++j) {
      const int c0 = 4 * (l + LPR * j);
      const f4 z4 = f4{0.f, 0.f, 0.f, 0.f};
      P.va[j] = P.vb[j] = P.vx[j] = P.vr[j] = P.vy[j] = z4;
      if (cok[j]) {
        P.va[j] = *(const f4*)(dyA + c0);
        if (dyB) P.vb[j] = *(const f4*)(dyB + c0);
        if (ln) P.vx[j] = *(const f4*)(x + c0);
        if (res) P.vr[j] = *(const f4*)(res + c0);
        if (ys) P.vy[j] = *(const f4*)(ys + c0);
      }
    }
  };
  Pass cur, nxt;
  fetch(r0, cur);
  for (int rb = r0; rb < r1; rb += RPP) {
    const bool more = rb + RPP < r1;
    if (more) fetch(rb + RPP, nxt);
    const bool rok = cur.rok;
    const unsigned rr = cur.rr;
    const float mu = cur.mu, rs = cur.rs;
    const bool ys = a.ysave.p != nullptr;
    float d[NE], xh[NE], g[NE], yv[NE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < VJ; ++j) {
      const int c0 = 4 * (l + LPR * j);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * j + e;
        float dv = cur.va[j][e] * dscale + cur.vb[j][e];
        dv *= dropout_scale(a.seed_pre, (uint64_t)(rr * (unsigned)C + (unsigned)(c0 + e)), a.p_pre);      // (R C < 2^31: checked by the host)
        d[i] = dv; yv[i] = cur.vy[j][e];
        xh[i] = (cur.vx[j][e] + cur.vr[j][e] - mu) * rs;
        g[i] = dv * gam[i];
        s1 += g[i];
        s2 += g[i] * xh[i];
      }
    }
    if (ln) {
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
      s1 *= invC; s2 *= invC;
    }
    float* outr = rv_row32(a.out, rr, C);
    float* rawr = a.dx_raw.p ? rv_row32(a.dx_raw, rr, C) : nullptr;
    const int pl = a.pad_L > 0 ? (int)(rr % (unsigned)a.pad_L) : -2;
#pragma unroll
    for (int j = 0; j < VJ; ++j) {
      const int c0 = 4 * (l + LPR * j);
      f4 vraw, vout;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * j + e;
        const float dx = ln ? rs * (g[i] - s1 - xh[i] * s2) : d[i];
        float o = dx * dropout_scale(a.seed_post, (uint64_t)(rr * (unsigned)C + (unsigned)(c0 + e)), a.p_post);
        if (ys) o = a.act == ACT_ELU ? o * d_elu_grad_from_out(yv[i]) : a.act == ACT_RELU ? (yv[i] > 0.f ? o : 0.f) : o;
        vraw[e] = dx; vout[e] = o;
        if (rok && cok[j]) {
          if (ln) { pg[i] += d[i] * xh[i]; pb[i] += d[i]; }
          pc[i] += o;
        }
      }
      if (rok && cok[j]) {
        if (rawr) *(f4*)(rawr + c0) = vraw;
        *(f4*)(outr + c0) = vout;
        if (pl == 0) *(f4*)(outr + c0 - C) = f4{0.f, 0.f, 0.f, 0.f};      // edge rows of the padded buffer this row sits in
        if (pl == a.pad_L - 1) *(f4*)(outr + c0 + C) = f4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (more) cur = nxt;
  }
  // column sums: registers -> LDS [3][slot][column] -> one atomic per column and block
#pragma unroll
  for (int j = 0; j < VJ; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = 4 * (l + LPR * j) + e, i = 4 * j + e;
      sred[0][slot][c] = pg[i]; sred[1][slot][c] = pb[i]; sred[2][slot][c] = pc[i];
    }
  __syncthreads();
  for (int c = tid; c < C; c += 64 * NW) {
    float tg = 0.f, tb = 0.f, tc = 0.f;
#pragma unroll
    for (int q = 0; q < RPP; ++q) { tg += sred[0][q][c]; tb += sred[1][q][c]; tc += sred[2][q][c]; }
    if (ln) { atomicAdd(a.dgamma + c, tg); atomicAdd(a.dbeta + c, tb); }
    if (a.dbias) atomicAdd(a.dbias + c, tc);
  }
}

__global__ __launch_bounds__(256) void softmax_fwd_k(float* P, float* Pd, const float* S, long R, int L, float p,
                                                      uint64_t seed) {
  long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= R) return;
  const float* s = S + row * L;
  float m = -INFINITY;
  for (int c = lane; c < L; c += 64) m = fmaxf(m, s[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float z = 0.f;
  for (int c = lane; c < L; c += 64) z += expf(s[c] - m);
  z = wave_sum(z);
  for (int c = lane; c < L; c += 64) {
    float v = expf(s[c] - m) / z;
    P[row * L + c] = v;
    if (Pd) Pd[row * L + c] = v * dropout_scale(seed, (uint64_t)(row * L + c), p);
  }
}

__global__ __launch_bounds__(256) void softmax_bwd_k(float* dS, const float* dPd, const float* P, long R, int L,
                                                      float p, uint64_t seed) {
  long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= R) return;
  float acc = 0.f;
  for (int c = lane; c < L; c += 64) {
    long i = row * L + c;
    acc += dPd[i] * dropout_scale(seed, (uint64_t)i, p) * P[i];
  }
  acc = wave_sum(acc);
  for (int c = lane; c < L; c += 64) {
    long i = row * L + c;
    dS[i] = P[i] * (dPd[i] * dropout_scale(seed, (uint64_t)i, p) - acc);
  }
}

// column sums: block = 256 threads = 64 columns x 4 row-phases; grid.x over column groups, grid.y over row chunks
__global__ __launch_bounds__(256) void colsum_k(float* out, const float* x, long R, int C, long ld,
                                                 long rows_per_block) {
  __shared__ float red[4][64];
  int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
  int c = blockIdx.x * 64 + cl;
  long r0 = (long)blockIdx.y * rows_per_block, r1 = r0 + rows_per_block;
  if (r1 > R) r1 = R;
  float s = 0.f;
  if (c < C)
    for (long r = r0 + ph; r < r1; r += 4) s += x[r * ld + c];
  red[ph][cl] = s;
  __syncthreads();
  if (ph == 0 && c < C) atomicAdd(out + c, red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}

// the same with 16-byte loads: a block covers 256 columns (64 lanes x float4) x 4 row phases
__global__ __launch_bounds__(256) void colsum4_k(float* out, const float* x, long R, int C, long ld, long rows_per_block) {
  __shared__ f4 red[4][64];
  const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cl) * 4;
  const long r0 = (long)blockIdx.y * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > R) r1 = R;
  f4 s = f4{0.f, 0.f, 0.f, 0.f};
  if (c < C)
    for (long r = r0 + ph; r < r1; r += 4) s += *(const f4*)(x + r * ld + c);
  red[ph][cl] = s;
  __syncthreads();
  if (ph == 0 && c < C) {
    const f4 t = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
    atomicAdd(out + c, t[0]); atomicAdd(out + c + 1, t[1]); atomicAdd(out + c + 2, t[2]); atomicAdd(out + c + 3, t[3]);
  }
}

__global__ __launch_bounds__(256) void colsum_v_k(float* out, RowView x, long R, int C, long rows_per_block) {
  __shared__ float red[4][64];
  int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
  int c = blockIdx.x * 64 + cl;
  long r0 = (long)blockIdx.y * rows_per_block, r1 = r0 + rows_per_block;
  if (r1 > R) r1 = R;
  float s = 0.f;
  if (c < C)
    for (long r = r0 + ph; r < r1; r += 4) s += rv_row(x, r, C)[c];
  red[ph][cl] = s;
  __syncthreads();
  if (ph == 0 && c < C) atomicAdd(out + c, red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}

__global__ void scale_k(float* p, long n, float v) { GS_LOOP(i, n) p[i] *= v; }

__global__ void pad_rows_k(float* dst, const float* src, int B, int T, int C, int pl, int pr, int mode) {
  int TP = pl + T + pr;
  long n = (long)B * TP * C;
  GS_LOOP(i, n) {
    int c = (int)(i % C);
    long r = i / C;
    int tp = (int)(r % TP);
    long b = r / TP;
    int t = tp - pl;
    float v = 0.f;
    if (t >= 0 && t < T) v = src[(b * T + t) * C + c];
    else if (mode == 1) v = src[(b * T + (t < 0 ? 0 : T - 1)) * C + c];
    dst[i] = v;
  }
}

__global__ void pad_edges_k(float* buf, int B, int T, int C, int pl, int pr, int mode) {
  int TP = pl + T + pr, NE = pl + pr;
  long n = (long)B * NE * C;
  GS_LOOP(i, n) {
    int c = (int)(i % C);
    long r = i / C;
    int e = (int)(r % NE);
    long b = r / NE;
    int tp = e < pl ? e : T + e;     // e >= pl -> pl + T + (e - pl)
    int src_t = e < pl ? pl : pl + T - 1;
    float v = mode == 1 ? buf[(b * TP + src_t) * C + c] : 0.f;
    buf[(b * TP + tp) * C + c] = v;
  }
}

__global__ void unpad_fold_k(float* dx, const float* dpad, int B, int T, int C, int pl, int pr, int mode) {
  int TP = pl + T + pr;
  long n = (long)B * T * C;
  GS_LOOP(i, n) {
    int c = (int)(i % C);
    long r = i / C;
    int t = (int)(r % T);
    long b = r / T;
    const float* d = dpad + (b * TP) * C + c;
    float v = d[(long)(t + pl) * C];
    if (mode == 1) {
      if (t == 0) for (int q = 0; q < pl; ++q) v += d[(long)q * C];
      if (t == T - 1) for (int q = 0; q < pr; ++q) v += d[(long)(pl + T + q) * C];
    }
    dx[i] = v;
  }
}

__global__ void pack_conv_w_k(float* wf, float* wb, const float* w, int Co, int Ci, int Kw) {
  long n = (long)Co * Ci * Kw;
  GS_LOOP(i, n) {
    int j = (int)(i % Kw);
    long r = i / Kw;
    int ci = (int)(r % Ci), co = (int)(r / Ci);
    float v = w[i];
    wf[((long)j * Ci + ci) * Co + co] = v;
    if (wb) wb[((long)(Kw - 1 - j) * Co + co) * Ci + ci] = v;
  }
}

struct PackConvW4 { PackConvW it[4]; long end[4]; int n; };
__global__ void pack_conv_w_multi_k(PackConvW4 q) {
  GS_LOOP(g, q.end[q.n - 1]) {
    int k = 0;
    while (g >= q.end[k]) ++k;
    const PackConvW& p = q.it[k];
    const long i = g - (k ? q.end[k - 1] : 0);
    const int j = (int)(i % p.Kw);
    const long r = i / p.Kw;
    const int ci = (int)(r % p.Ci), co = (int)(r / p.Ci);
    const float v = p.w[i];
    p.wf[((long)j * p.Ci + ci) * p.Co + co] = v;
    if (p.wb) p.wb[((long)(p.Kw - 1 - j) * p.Co + co) * p.Ci + ci] = v;
  }
}

// the same through LDS tiles (Kw <= 4): a block moves [Kw][32 ci][32 co]; reads run along co (contiguous in dWf), writes along
// (ci, j) (contiguous in dW) -- the gather above touches one cache line per element on the read side (1.7 M elements of the style
// encoder's first convolution: 36 us at the very end of the iteration, behind the last GEMM)
__global__ __launch_bounds__(256) void unpack_conv_dw_t_k(float* dw, const float* dwf, int Co, int Ci, int Kw) {
  __shared__ float t[4][32][33];
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8 threads
  for (int j = 0; j < Kw; ++j)
    for (int r = ty; r < 32; r += 8) {
      const int ci = ci0 + r, co = co0 + tx;
      t[j][r][tx] = (ci < Ci && co < Co) ? dwf[((long)j * Ci + ci) * Co + co] : 0.f;
    }
  __syncthreads();
  const int run = 32 * Kw;                                     // floats of one co row of this tile in dW: (ci0 .. ci0 + 31) x Kw
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r;
    if (co >= Co) continue;
    for (int e = tx; e < run; e += 32) {
      const int cl = e / Kw, j = e - cl * Kw, ci = ci0 + cl;
      if (ci < Ci) dw[((long)co * Ci + ci) * Kw + j] = t[j][cl][r];
    }
  }
}

__global__ void unpack_conv_dw_k(float* dw, const float* dwf, int Co, int Ci, int Kw) {
  long n = (long)Co * Ci * Kw;
  GS_LOOP(i, n) {
    int j = (int)(i % Kw);
    long r = i / Kw;
    int ci = (int)(r % Ci), co = (int)(r / Ci);
    dw[i] = dwf[((long)j * Ci + ci) * Co + co];
  }
}

__global__ void add_rows_bcast_k(float* h, const float* table, int B, int L, int C) {
  long n = (long)B * L * C, lc = (long)L * C;
  GS_LOOP(i, n) h[i] += table[i % lc];
}

__global__ void meanpool_fwd_k(float* out, const float* f, int B, int L, int C) {
  // block per (b, 64-col group); 16 row phases (one wave each: 256 workgroups of 4 waves left the rows of a 384-frame exemplar
  // 96 deep per thread -- 26 us for 25 MB on the chain in front of the forward sweep)
  __shared__ float red[16][64];
  int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
  int b = blockIdx.y, c = blockIdx.x * 64 + cl;
  float s = 0.f;
  if (c < C)
    for (int l = ph; l < L; l += 16) s += f[((long)b * L + l) * C + c];
  red[ph][cl] = s;
  __syncthreads();
  if (ph == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][cl];
    out[(long)b * C + c] = t / (float)L;
  }
}

__global__ void meanpool_bwd_k(float* df, const float* dout, int B, int L, int C) {
  long n = (long)B * L * C;
  GS_LOOP(i, n) {
    int c = (int)(i % C);
    long b = i / ((long)L * C);
    df[i] = dout[b * C + c] / (float)L;
  }
}

}  // namespace

#define L1D(kernel, n, s, ...)                                                        \
  do {                                                                                \
    if ((n) > 0) {                                                                    \
      hipLaunchKernelGGL(kernel, dim3(grid1d(n)), dim3(256), 0, s, __VA_ARGS__);      \
      ZLAUNCH_CHECK(#kernel);                                                         \
    }                                                                                 \
  } while (0)

int k_fill(float* p, long n, float v, hipStream_t s) {
  // unaligned head, 16-byte body, tail (one launch when the buffer is aligned and a multiple of four floats: the usual case)
  const long head = (((size_t)p & 15) && n > 0) ? ((16 - ((size_t)p & 15)) / 4 < (size_t)n ? (long)((16 - ((size_t)p & 15)) / 4) : n) : 0;
  const long n4 = (n - head) / 4, tail = n - head - 4 * n4;
  if (n4 < 1024) { L1D(fill_k, n, s, p, n, v); return 0; }
  if (head) L1D(fill_k, head, s, p, head, v);
  L1D(fill4_k, n4, s, (f4*)(p + head), n4, v);
  if (tail) L1D(fill_k, tail, s, p + head + 4 * n4, tail, v);
  return 0;
}
int k_copy(float* d, const float* src, long n, hipStream_t s) { L1D(copy_k, n, s, d, src, n); return 0; }
int k_add_inplace(float* d, const float* src, long n, hipStream_t s) { L1D(add_k, n, s, d, src, n); return 0; }
int k_act_bwd(float* dx, const float* dy, const float* y, long n, int act, float ys, hipStream_t s) {
  L1D(act_bwd_k, n, s, dx, dy, y, n, act, ys);
  return 0;
}
int k_dropout(float* x, long n, float p, uint64_t seed, hipStream_t s) {
  if (p <= 0.f) return 0;
  L1D(dropout_k, n, s, x, n, p, seed);
  return 0;
}
int k_dropout_rows(float* x, int rows, int cols, long ld, long batch_rows, long batch_stride, float p,
                   uint64_t seed, hipStream_t s) {
  if (p <= 0.f) return 0;
  long n = (long)rows * cols;
  L1D(dropout_rows_k, n, s, x, rows, cols, ld, batch_rows, batch_stride, p, seed);
  return 0;
}
int k_layernorm_fwd_v(RowView y, RowView x, RowView res, const float* gamma, const float* beta, float* mean,
                      float* rstd, int R, int C, float eps, hipStream_t s) {
  hipLaunchKernelGGL(layernorm_fwd_k, dim3(cdiv(R, 4)), dim3(256), 0, s, y, x, res, gamma, beta, mean, rstd, R, C, eps);
  ZLAUNCH_CHECK("layernorm_fwd");
  return 0;
}
int k_layernorm_fwd(float* y, const float* x, const float* res, const float* gamma, const float* beta,
                    float* mean, float* rstd, int R, int C, float eps, hipStream_t s) {
  return k_layernorm_fwd_v(rv(y), rv(x), rv(res), gamma, beta, mean, rstd, R, C, eps, s);
}
int k_layernorm_bwd_v(RowView dx, RowView dy, RowView x, RowView res, const float* gamma, const float* mean,
                      const float* rstd, float* dgamma, float* dbeta, int R, int C, hipStream_t s) {
  // 4 rows per wave: a row is three dependent round trips (load, wave reduction, store), so the kernel lives on the number of
  // waves in flight; the price is one atomic per column and block (R / 16 blocks onto C addresses)
  const int rpb = 16;
  ZCHECK(C <= 512, "layernorm_bwd: C=%d > 512 unsupported", C);
  if (C <= 128)
    hipLaunchKernelGGL((layernorm_bwd_k<2>), dim3(cdiv(R, rpb)), dim3(256), 0, s, dx, dy, x, res, gamma, mean, rstd,
                       dgamma, dbeta, R, C, rpb);
  else
    hipLaunchKernelGGL((layernorm_bwd_k<8>), dim3(cdiv(R, rpb)), dim3(256), 0, s, dx, dy, x, res, gamma, mean, rstd,
                       dgamma, dbeta, R, C, rpb);
  ZLAUNCH_CHECK("layernorm_bwd");
  return 0;
}
LnFwdFused ln_fwd_fused_args(int R, int C, float eps) {
  LnFwdFused a;
  memset(&a, 0, sizeof(a));
  a.R = R; a.C = C; a.eps = eps; a.table_L = 1;
  return a;
}
static bool rv_al16_(const RowView& v) { return v.p == nullptr || ((((size_t)v.p) & 15) == 0 && (v.rpb == 0 || v.bstride % 4 == 0)); }
extern int g_ln_bwd4;
int k_ln_fwd_fused(const LnFwdFused& a, hipStream_t s) {
  ZCHECK(a.C <= 512, "ln_fwd_fused: C=%d > 512 unsupported", a.C);
  if (g_ln_bwd4 && a.C % 4 == 0 && a.R > 0 && rv_al16_(a.x) && rv_al16_(a.res) && rv_al16_(a.y) && (long)a.R * a.C < (1L << 31)) {
    if (a.C <= 128) hipLaunchKernelGGL((ln_fwd_fused4_k<32, 1>), dim3(cdiv(a.R, 8)), dim3(256), 0, s, a);
    else if (a.C <= 256) hipLaunchKernelGGL((ln_fwd_fused4_k<64, 1>), dim3(cdiv(a.R, 4)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((ln_fwd_fused4_k<64, 2>), dim3(cdiv(a.R, 4)), dim3(256), 0, s, a);
    ZLAUNCH_CHECK("ln_fwd_fused4");
    return 0;
  }
  if (a.C <= 128) hipLaunchKernelGGL((ln_fwd_fused_k<2>), dim3(cdiv(a.R, 4)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((ln_fwd_fused_k<8>), dim3(cdiv(a.R, 4)), dim3(256), 0, s, a);
  ZLAUNCH_CHECK("ln_fwd_fused");
  return 0;
}
LnBwdFused ln_bwd_fused_args(int R, int C) {
  LnBwdFused a;
  memset(&a, 0, sizeof(a));
  a.R = R; a.C = C; a.pool_L = 1;
  return a;
}
int g_ln_bwd4 = 1;      // zeggs_set_option("ln_bwd4", 0/1): the 16-byte-lane form of the fused LayerNorm backward pass (A/B switch)
static bool rv_al16(const RowView& v) { return v.p == nullptr || ((((size_t)v.p) & 15) == 0 && (v.rpb == 0 || v.bstride % 4 == 0)); }
int k_ln_bwd_fused(const LnBwdFused& a, hipStream_t s) {
  ZCHECK(a.C <= 512, "ln_bwd_fused: C=%d > 512 unsupported", a.C);
  ZCHECK(a.out.p != nullptr && (a.dy_pool != nullptr || a.dyA.p != nullptr), "ln_bwd_fused: missing operand");
  const bool al = a.C % 4 == 0 && rv_al16(a.dyA) && rv_al16(a.dyB) && rv_al16(a.x) && rv_al16(a.res) && rv_al16(a.dx_raw) &&
                  rv_al16(a.out) && rv_al16(a.ysave) && (((size_t)a.dy_pool) & 15) == 0 &&
                  a.R > 0 && (long)a.R * a.C < (1L << 31);
  if (g_ln_bwd4 && al) {
    // 16-wave blocks, ~1 per CU (the column sums of a block end in 3 C atomics: with 512 four-wave blocks those 196 k atomics onto
    // 24 cache lines were half of the kernel's 20 us), whole passes per block
    // (wider rows keep 4-wave blocks: two pipelined passes of 8 float4 operands do not fit 16 waves' register budget)
    auto rows_per_block = [&](int rpp, int blocks) { const int per = cdiv(a.R, blocks); return cdiv(per, rpp) * rpp; };
    if (a.C <= 128 && g_ln_bwd4 == 2) { const int rpb = rows_per_block(8, 512); hipLaunchKernelGGL((ln_bwd_fused4_k<32, 1, 4>), dim3(cdiv(a.R, rpb)), dim3(256), 0, s, a, rpb); }
    else if (a.C <= 128) { const int rpb = rows_per_block(32, 256); hipLaunchKernelGGL((ln_bwd_fused4_k<32, 1, 16>), dim3(cdiv(a.R, rpb)), dim3(1024), 0, s, a, rpb); }
    else if (a.C <= 256) { const int rpb = rows_per_block(8, 256); hipLaunchKernelGGL((ln_bwd_fused4_k<64, 1, 8>), dim3(cdiv(a.R, rpb)), dim3(512), 0, s, a, rpb); }
    else { const int rpb = rows_per_block(4, 512); hipLaunchKernelGGL((ln_bwd_fused4_k<64, 2, 4>), dim3(cdiv(a.R, rpb)), dim3(256), 0, s, a, rpb); }
    ZLAUNCH_CHECK("ln_bwd_fused4");
    return 0;
  }
  const int rpb = 16;
  if (a.C <= 128) hipLaunchKernelGGL((ln_bwd_fused_k<2>), dim3(cdiv(a.R, rpb)), dim3(256), 0, s, a, rpb);
  else hipLaunchKernelGGL((ln_bwd_fused_k<8>), dim3(cdiv(a.R, rpb)), dim3(256), 0, s, a, rpb);
  ZLAUNCH_CHECK("ln_bwd_fused");
  return 0;
}
int k_layernorm_bwd(float* dx, const float* dy, const float* x, const float* res, const float* gamma,
                    const float* mean, const float* rstd, float* dgamma, float* dbeta, int R, int C,
                    hipStream_t s) {
  return k_layernorm_bwd_v(rv(dx), rv(dy), rv(x), rv(res), gamma, mean, rstd, dgamma, dbeta, R, C, s);
}
int k_act_bwd_v(RowView dx, RowView dy, RowView y, long R, int C, int act, float ys, hipStream_t s) {
  long n = R * C;
  L1D(act_bwd_v_k, n, s, dx, dy, y, R, C, act, ys);
  return 0;
}
int k_colsum_v(float* out, RowView x, long R, int C, float beta, hipStream_t s) {
  if (beta == 0.f) { ZTRY(k_fill(out, C, 0.f, s)); }
  long rpb = 256;
  dim3 grid(cdiv(C, 64), cdiv(R, rpb));
  hipLaunchKernelGGL(colsum_v_k, grid, dim3(256), 0, s, out, x, R, C, rpb);
  ZLAUNCH_CHECK("colsum_v");
  return 0;
}
int k_softmax_fwd(float* P, float* Pd, const float* S, long R, int L, float p, uint64_t seed, hipStream_t s) {
  hipLaunchKernelGGL(softmax_fwd_k, dim3(cdiv(R, 4)), dim3(256), 0, s, P, Pd, S, R, L, p, seed);
  ZLAUNCH_CHECK("softmax_fwd");
  return 0;
}
int k_softmax_bwd(float* dS, const float* dPd, const float* P, long R, int L, float p, uint64_t seed,
                  hipStream_t s) {
  hipLaunchKernelGGL(softmax_bwd_k, dim3(cdiv(R, 4)), dim3(256), 0, s, dS, dPd, P, R, L, p, seed);
  ZLAUNCH_CHECK("softmax_bwd");
  return 0;
}
int k_colsum(float* out, const float* x, long R, int C, long ld, float beta, hipStream_t s) {
  if (beta == 0.f) { ZTRY(k_fill(out, C, 0.f, s)); }
  else if (beta != 1.f) { L1D(scale_k, (long)C, s, out, (long)C, beta); }
  long rpb = 256;
  if (C % 4 == 0 && ld % 4 == 0 && ((uintptr_t)x & 15) == 0 && C >= 256) {
    rpb = 128;        // 4x fewer column blocks: keep the grid large
    dim3 grid(cdiv(C, 256), cdiv(R, rpb));
    hipLaunchKernelGGL(colsum4_k, grid, dim3(256), 0, s, out, x, R, C, ld, rpb);
  } else {
    dim3 grid(cdiv(C, 64), cdiv(R, rpb));
    hipLaunchKernelGGL(colsum_k, grid, dim3(256), 0, s, out, x, R, C, ld, rpb);
  }
  ZLAUNCH_CHECK("colsum");
  return 0;
}
int k_pad_rows(float* dst, const float* src, int B, int T, int C, int pl, int pr, int mode, hipStream_t s) {
  long n = (long)B * (pl + T + pr) * C;
  L1D(pad_rows_k, n, s, dst, src, B, T, C, pl, pr, mode);
  return 0;
}
int k_pad_edges(float* buf, int B, int T, int C, int pl, int pr, int mode, hipStream_t s) {
  long n = (long)B * (pl + pr) * C;
  L1D(pad_edges_k, n, s, buf, B, T, C, pl, pr, mode);
  return 0;
}
int k_unpad_fold(float* dx, const float* dpad, int B, int T, int C, int pl, int pr, int mode, hipStream_t s) {
  long n = (long)B * T * C;
  L1D(unpad_fold_k, n, s, dx, dpad, B, T, C, pl, pr, mode);
  return 0;
}
int k_pack_conv_w(float* wf, float* wb, const float* w, int Co, int Ci, int Kw, hipStream_t s) {
  long n = (long)Co * Ci * Kw;
  L1D(pack_conv_w_k, n, s, wf, wb, w, Co, Ci, Kw);
  return 0;
}
int k_pack_conv_w_multi(const PackConvW* items, int n, hipStream_t s) {
  ZCHECK(n >= 1 && n <= 4, "pack_conv_w_multi: %d items", n);
  PackConvW4 q;
  memset(&q, 0, sizeof(q));
  long tot = 0;
  for (int k = 0; k < n; ++k) { q.it[k] = items[k]; tot += (long)items[k].Co * items[k].Ci * items[k].Kw; q.end[k] = tot; }
  q.n = n;
  L1D(pack_conv_w_multi_k, tot, s, q);
  return 0;
}
int k_unpack_conv_dw(float* dw, const float* dwf, int Co, int Ci, int Kw, hipStream_t s) {
  long n = (long)Co * Ci * Kw;
  if (Kw <= 4 && n >= 65536) {
    hipLaunchKernelGGL(unpack_conv_dw_t_k, dim3(cdiv(Ci, 32), cdiv(Co, 32)), dim3(256), 0, s, dw, dwf, Co, Ci, Kw);
    ZLAUNCH_CHECK("unpack_conv_dw_t");
    return 0;
  }
  L1D(unpack_conv_dw_k, n, s, dw, dwf, Co, Ci, Kw);
  return 0;
}
int k_add_rows_bcast(float* h, const float* table, int B, int L, int C, hipStream_t s) {
  long n = (long)B * L * C;
  L1D(add_rows_bcast_k, n, s, h, table, B, L, C);
  return 0;
}
int k_meanpool_fwd(float* out, const float* f, int B, int L, int C, hipStream_t s) {
  hipLaunchKernelGGL(meanpool_fwd_k, dim3(cdiv(C, 64), B), dim3(1024), 0, s, out, f, B, L, C);
  ZLAUNCH_CHECK("meanpool_fwd");
  return 0;
}
int k_meanpool_bwd(float* df, const float* dout, int B, int L, int C, hipStream_t s) {
  long n = (long)B * L * C;
  L1D(meanpool_bwd_k, n, s, df, dout, B, L, C);
  return 0;
}
