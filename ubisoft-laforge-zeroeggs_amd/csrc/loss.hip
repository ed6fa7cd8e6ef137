// Training loss of the reference (ZEGGS/train.py:276-421) and its analytic backward.
//
// Per (batch, frame) the loss needs: 2-axis -> rotation matrices, the first joint
// moved to world space by the root transform, a 75-joint forward-kinematics pass
// with linear/angular velocities (ZEGGS/anim/txform.py:10-34), then 17 weighted
// mean-|difference| terms (4 of them on finite differences along time) and the KL
// term.  All per-frame quantities are kept as a structure-of-arrays feature table
// F[e][frame] so that the 64 lanes of a wave (= 64 consecutive frames) always touch
// consecutive addresses; the FK chain is walked sequentially per frame (thread per
// frame), the reductions run one block per (feature row, 256 frames).
#include "../../include/zeggs_hip.h"
#include "common.h"
#include "kernels.h"
#include <type_traits>

// waves of a frame workgroup (64 frames = the lanes; the waves share the joints of a tree level)
#ifndef ZEGGS_LOSS_WAVES
#define ZEGGS_LOSS_WAVES 8
#endif

namespace {

struct Off {
  int rpos, rmat, rvel, rvrt, lpos, ltxy, lvel, lvrt, cpos, cmat, cvel, cvrt, gaze, n;
};
__host__ __device__ inline Off offsets(int J) {
  Off o;
  o.rpos = 0; o.rmat = 3; o.rvel = 12; o.rvrt = 15; o.lpos = 18; o.ltxy = 18 + 3 * J; o.lvel = 18 + 9 * J;
  o.lvrt = 18 + 12 * J; o.cpos = 18 + 15 * J; o.cmat = 18 + 18 * J; o.cvel = 18 + 27 * J; o.cvrt = 18 + 30 * J;
  o.gaze = 18 + 33 * J; o.n = 21 + 33 * J;
  return o;
}

constexpr int LOSS_TSPLIT = 2;      // workgroups per feature row of loss_terms4_k
struct LossWs {
  float *FO, *FW, *LM, *G, *DQ;
  float *PT0, *PT1, *DPT;   // pose rows of both sides / their gradient, transposed to [PO][frames]
  float* PS;                // per-workgroup partial sums of loss_terms4_k [rows][LOSS_TSPLIT][2]
};
LossWs carve_loss(const ZeggsLossDims& d, Arena& a) {
  LossWs w;
  const long NF = (long)d.B * d.T;
  const Off o = offsets(d.J);
  w.FO = a.f(NF * o.n); w.FW = a.f(NF * o.n); w.LM = a.f(NF * 9 * d.J); w.G = a.f(NF * o.n); w.DQ = a.f(NF * 4);
  const long PO = 6 + 15 * d.J;
  w.PT0 = a.f(NF * PO); w.PT1 = a.f(NF * PO); w.DPT = a.f(NF * PO);
  w.PS = a.f((long)o.n * LOSS_TSPLIT * 2);
  return w;
}

struct M3 { float m[9]; };   // row-major
__device__ __forceinline__ V3 mv(const M3& a, V3 v) {
  return v3(a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z,
            a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z);
}
__device__ __forceinline__ V3 mtv(const M3& a, V3 v) {   // a^T v
  return v3(a.m[0] * v.x + a.m[3] * v.y + a.m[6] * v.z, a.m[1] * v.x + a.m[4] * v.y + a.m[7] * v.z,
            a.m[2] * v.x + a.m[5] * v.y + a.m[8] * v.z);
}
__device__ __forceinline__ M3 mm(const M3& a, const M3& b) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) c.m[i * 3 + j] = a.m[i * 3] * b.m[j] + a.m[i * 3 + 1] * b.m[3 + j] + a.m[i * 3 + 2] * b.m[6 + j];
  return c;
}
__device__ __forceinline__ M3 mtm(const M3& a, const M3& b) {   // a^T b
  M3 c;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) c.m[i * 3 + j] = a.m[i] * b.m[j] + a.m[3 + i] * b.m[3 + j] + a.m[6 + i] * b.m[6 + j];
  return c;
}
__device__ __forceinline__ M3 mmt(const M3& a, const M3& b) {   // a b^T
  M3 c;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) c.m[i * 3 + j] = a.m[i * 3] * b.m[j * 3] + a.m[i * 3 + 1] * b.m[j * 3 + 1] + a.m[i * 3 + 2] * b.m[j * 3 + 2];
  return c;
}
__device__ __forceinline__ void add_outer(M3& g, V3 a, V3 b) {   // g += a b^T
  g.m[0] += a.x * b.x; g.m[1] += a.x * b.y; g.m[2] += a.x * b.z;
  g.m[3] += a.y * b.x; g.m[4] += a.y * b.y; g.m[5] += a.y * b.z;
  g.m[6] += a.z * b.x; g.m[7] += a.z * b.y; g.m[8] += a.z * b.z;
}
// reference anim/tquat.py:54-69
__device__ __forceinline__ M3 quat_to_xform(Q4 q) {
  float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
  float xx = q.x * x2, yy = q.y * y2, wx = q.w * x2;
  float xy = q.x * y2, yz = q.y * z2, wy = q.w * y2;
  float xz = q.x * z2, zz = q.z * z2, wz = q.w * z2;
  M3 r;
  r.m[0] = 1.0f - (yy + zz); r.m[1] = xy - wz; r.m[2] = xz + wy;
  r.m[3] = xy + wz; r.m[4] = 1.0f - (xx + zz); r.m[5] = yz - wx;
  r.m[6] = xz - wy; r.m[7] = yz + wx; r.m[8] = 1.0f - (xx + yy);
  return r;
}
__device__ __forceinline__ Q4 quat_to_xform_bwd(Q4 q, const M3& g) {
  const float* m = g.m;   // g00 g01 g02 g10 g11 g12 g20 g21 g22
  Q4 d;
  d.w = 2.f * (-q.z * m[1] + q.y * m[2] + q.z * m[3] - q.x * m[5] - q.y * m[6] + q.x * m[7]);
  d.x = 2.f * (q.y * m[1] + q.z * m[2] + q.y * m[3] - 2.f * q.x * m[4] - q.w * m[5] + q.z * m[6] + q.w * m[7] - 2.f * q.x * m[8]);
  d.y = 2.f * (-2.f * q.y * m[0] + q.x * m[1] + q.w * m[2] + q.x * m[3] + q.z * m[5] - q.w * m[6] + q.z * m[7] - 2.f * q.y * m[8]);
  d.z = 2.f * (-2.f * q.z * m[0] - q.w * m[1] + q.x * m[2] + q.w * m[3] - 2.f * q.z * m[4] + q.y * m[5] + q.x * m[6] + q.y * m[7]);
  return d;
}
__device__ __forceinline__ void qmv_bwd(Q4 q, V3 v, V3 g, Q4& dq, V3& dv) {
  V3 qv = v3(q.x, q.y, q.z);
  V3 t = 2.0f * cross(qv, v);
  float dw = dot(g, t);
  V3 dt = q.w * g + cross(g, qv);
  V3 dqv = cross(t, g) + 2.0f * cross(v, dt);
  dv = g + 2.0f * cross(dt, qv);
  dq = Q4{dw, dqv.x, dqv.y, dqv.z};
}
__device__ __forceinline__ float vnorm(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
// grad of a/(|a|+e) wrt a, upstream g
__device__ __forceinline__ V3 normalize_bwd(V3 a, V3 g, float e) {
  float n = vnorm(a), ne = n + e;
  float k = n > 0.f ? dot(a, g) / (n * ne * ne) : 0.f;
  return (1.f / ne) * g - k * a;
}

struct FrameIO {
  const float *pose, *rpos, *rrot;   // [B,T,*]
};
// The frame kernels run one thread per frame: the [frame][PO] pose rows are first transposed to [PO][frame] so that
// the 64 lanes of a wave read / write consecutive addresses (a lane-per-row access touches 64 cache lines per
// instruction).  Col / ColW keep the row-style indexing p[k] over the transposed table.
struct Col {
  const float* base; long NF, f;
  __device__ __forceinline__ float operator[](int k) const { return base[(long)k * NF + f]; }
  __device__ __forceinline__ Col operator+(int k) const { return Col{base + (long)k * NF, NF, f}; }
};
struct ColW {
  float* base; long NF, f;
  __device__ __forceinline__ float& operator[](int k) const { return base[(long)k * NF + f]; }
  __device__ __forceinline__ ColW operator+(int k) const { return ColW{base + (long)k * NF, NF, f}; }
};
// dst[c][r] = src[r][c] for an [R][C] row-major source (64x64 tiles through LDS, both sides coalesced)
__global__ __launch_bounds__(256) void transpose_k(float* dst, const float* src, long R, int C) {
  __shared__ float tile[64][65];
  const long r0 = (long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4)
    if (r0 + i < R && c0 + tx < C) tile[i][tx] = src[(r0 + i) * C + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 64; i += 4)
    if (c0 + i < C && r0 + tx < R) dst[(long)(c0 + i) * R + r0 + tx] = tile[tx][i];
}

#define FE(F, e) (F)[(long)(e) * NF + f]

// Joints grouped by depth in the skeleton tree: the joints of one level are independent of each other, so the waves
// of a workgroup walk the tree level by level (critical path = tree depth, ~13 for the 75-joint rig, instead of J).
// Built in LDS by every workgroup (J is small): lvl_start[l] .. lvl_start[l+1] index lvl_joint[].
constexpr int MAXJ = 256;
struct Levels { int start[MAXJ + 1]; int joint[MAXJ]; int depth[MAXJ]; int par[MAXJ]; int pos[MAXJ]; int nlevels; int maxw; };
__device__ void build_levels(Levels& L, const int* parents, int J) {
  for (int j = threadIdx.x; j <= J; j += blockDim.x) L.start[j] = 0;
  for (int j = threadIdx.x; j < J; j += blockDim.x) L.par[j] = parents[j];      // one coalesced load; the chains are walked in LDS
  __syncthreads();
  for (int j = threadIdx.x; j < J; j += blockDim.x) {
    int dpt = 0;
    for (int p = L.par[j]; p >= 0; p = L.par[p]) ++dpt;
    L.depth[j] = dpt;
    atomicAdd(&L.start[dpt + 1], 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int nl = 0, mw = 0;
    for (int l = 0; l < J; ++l) { if (L.start[l + 1] > 0) nl = l + 1; mw = max(mw, L.start[l + 1]); L.start[l + 1] += L.start[l]; }
    L.nlevels = nl; L.maxw = mw;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < J; j += blockDim.x) {   // index order within a level (deterministic)
    const int dj = L.depth[j];
    int pos = L.start[dj];
    for (int q = 0; q < j; ++q) pos += (L.depth[q] == dj);
    L.joint[pos] = j;
    L.pos[j] = pos - L.start[dj];
  }
  __syncthreads();
}
// Round 5: the transforms a level hands to the next one (forward: a joint's character-space matrix / position / velocities for
// its children; backward: a joint's message to its parent) travel through LDS, 18 floats x 64 frames per joint, two level buffers,
// when no level of the skeleton is wider than LOSS_LW joints (a wider one walks the tables in global memory as before).  The walk's
// critical path is the tree depth (13 levels for the 75-joint rig, most of them 3-5 joints wide: the waves have little to do and
// a lot to wait for): through global memory every level paid the write acknowledgement of its stores (the barrier's vmcnt(0)) and
// an L2 round trip for the loads behind it.  The barrier of the LDS walk waits for LDS only; the table stores drain on their own.
constexpr int LOSS_LW = 16, LOSS_MSG = 18 * 64;
constexpr size_t LOSS_LDS_BYTES = (size_t)2 * LOSS_LW * LOSS_MSG * sizeof(float);      // 147 456
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
extern __shared__ float loss_msg[];

// forward per frame; side 0 = prediction (also stores local matrices LM), side 1 = ground truth.
// One workgroup = 64 consecutive frames (lanes) x 8 waves sharing the joints of each tree level.
// items gid0 .. gid_end - 1 of the 2 NF (side, frame) pairs: [0, NF) = prediction, [NF, 2 NF) = ground truth (the truth half
// depends on the batch only: zeggs_loss_prepare_truth runs it ahead of the step)
__global__ __launch_bounds__(64 * ZEGGS_LOSS_WAVES) void loss_frame_fwd_k(ZeggsLossDims d, const int* parents, FrameIO io0, FrameIO io1,
                                                        const float* PT0, const float* PT1, const float* gaze, float* F0,
                                                        float* F1, float* LM_, long gid0, long gid_end, int lds_ok) {
  __shared__ Levels lv;
  const long NF = (long)d.B * d.T;
  build_levels(lv, parents, d.J);
  const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const long gid = gid0 + (long)blockIdx.x * 64 + (threadIdx.x & 63);
  const bool live = gid < gid_end;                // dead lanes keep walking (barriers), on a clamped frame, storing nothing
  const int side = live ? gid >= NF : 0;
  const long f = live ? (side ? gid - NF : gid) : 0;
  const FrameIO io = side ? io1 : io0;
  float* F = side ? F1 : F0;
  const int J = d.J, t = (int)(f % d.T);
  const Off o = offsets(J);
  const int PO = 6 + 15 * J;
  const Col p{side ? PT1 : PT0, NF, f};
  const float* rq = io.rrot + f * 4;
  const float* rqp = io.rrot + (t > 0 ? f - 1 : f) * 4;
  Q4 q = Q4{rq[0], rq[1], rq[2], rq[3]}, qp = Q4{rqp[0], rqp[1], rqp[2], rqp[3]};
  V3 rpos = v3(io.rpos[f * 3], io.rpos[f * 3 + 1], io.rpos[f * 3 + 2]);
  V3 rvel = quat_mul_vec(qp, v3(p[0], p[1], p[2]));
  V3 rvrt = quat_mul_vec(qp, v3(p[3], p[4], p[5]));
  M3 R = quat_to_xform(q);
  if (live && wave == 0) FE(F, o.rpos) = rpos.x; if (live && wave == 0) FE(F, o.rpos + 1) = rpos.y; if (live && wave == 0) FE(F, o.rpos + 2) = rpos.z;
  for (int k = 0; k < 9; ++k) if (live && wave == 0) FE(F, o.rmat + k) = R.m[k];
  if (live && wave == 0) FE(F, o.rvel) = rvel.x; if (live && wave == 0) FE(F, o.rvel + 1) = rvel.y; if (live && wave == 0) FE(F, o.rvel + 2) = rvel.z;
  if (live && wave == 0) FE(F, o.rvrt) = rvrt.x; if (live && wave == 0) FE(F, o.rvrt + 1) = rvrt.y; if (live && wave == 0) FE(F, o.rvrt + 2) = rvrt.z;
  {
    const float* gz = gaze + f * 3;
    V3 v = v3(gz[0], gz[1], gz[2]) - rpos;
    float inv = 1.f / (vnorm(v) + 1e-8f);
    V3 gd = quat_mul_vec(quat_inv(q), inv * v);
    if (live && wave == 0) FE(F, o.gaze) = gd.x; if (live && wave == 0) FE(F, o.gaze + 1) = gd.y; if (live && wave == 0) FE(F, o.gaze + 2) = gd.z;
  }
  const Col lpos = p + 6, ltxy = p + (6 + 3 * J), lvel = p + (6 + 9 * J), lvrt = p + (6 + 12 * J);
  const int lane = threadIdx.x & 63;
  struct FIn { V3 x, yi, lp, lv, lw; };      // a joint's reads from the pose table: independent of the level above it
  auto fetch = [&](int i) {
    FIn a;
    a.x = v3(ltxy[6 * i], ltxy[6 * i + 1], ltxy[6 * i + 2]); a.yi = v3(ltxy[6 * i + 3], ltxy[6 * i + 4], ltxy[6 * i + 5]);
    a.lp = v3(lpos[3 * i], lpos[3 * i + 1], lpos[3 * i + 2]);
    a.lv = v3(lvel[3 * i], lvel[3 * i + 1], lvel[3 * i + 2]);
    a.lw = v3(lvrt[3 * i], lvrt[3 * i + 1], lvrt[3 * i + 2]);
    return a;
  };
  auto walk = [&](auto lm_) {
  constexpr bool LM = decltype(lm_)::value;
  FIn cur;
  bool have = false;
  for (int l = 0; l < lv.nlevels; ++l) {
   float* mine = loss_msg + (l & 1) * (LOSS_LW * LOSS_MSG) + lane;
   const float* prev = loss_msg + ((l & 1) ^ 1) * (LOSS_LW * LOSS_MSG) + lane;
   for (int kk = lv.start[l] + wave; kk < lv.start[l + 1]; kk += nwaves) {
    const int i = lv.joint[kk];
    if (!have) cur = fetch(i);
    // the LDS walk fetches this wave's NEXT joint (next round of the level, or its first one of the level below) before it
    // reads the parent's transforms: the table reads are in flight across the barrier
    FIn nxt;
    bool nv = false;
    if constexpr (LM) {
      int nk = kk + nwaves;
      nv = nk < lv.start[l + 1];
      if (!nv && l + 1 < lv.nlevels) { nk = lv.start[l + 1] + wave; nv = nk < lv.start[l + 2]; }
      if (nv) nxt = fetch(lv.joint[nk]);
    }
    // orthogonalise (txform.py:23-34): columns x^, y^, z^
    V3 x = cur.x, yi = cur.yi;
    V3 lp = cur.lp, lv_ = cur.lv, lw = cur.lw;
    cur = nxt; have = nv;
    if (live) { FE(F, o.ltxy + 6 * i) = x.x; FE(F, o.ltxy + 6 * i + 1) = x.y; FE(F, o.ltxy + 6 * i + 2) = x.z;
                FE(F, o.ltxy + 6 * i + 3) = yi.x; FE(F, o.ltxy + 6 * i + 4) = yi.y; FE(F, o.ltxy + 6 * i + 5) = yi.z; }
    V3 z = cross(x, yi), y = cross(z, x);
    V3 xn = (1.f / (vnorm(x) + 1e-10f)) * x, yn = (1.f / (vnorm(y) + 1e-10f)) * y, zn = (1.f / (vnorm(z) + 1e-10f)) * z;
    M3 L;
    L.m[0] = xn.x; L.m[1] = yn.x; L.m[2] = zn.x;
    L.m[3] = xn.y; L.m[4] = yn.y; L.m[5] = zn.y;
    L.m[6] = xn.z; L.m[7] = yn.z; L.m[8] = zn.z;
    if (side == 0)
      for (int k = 0; k < 9; ++k) if (live) FE(LM_, 9 * i + k) = L.m[k];
    M3 cm; V3 cp, cv, cw;
    if (i == 0) {                               // train.py:296-303: first joint to world space
      V3 rl = quat_mul_vec(q, lp);
      cp = rl + rpos;
      cm = mm(R, L);
      cv = rvel + quat_mul_vec(q, lv_) + cross(rvrt, rl);
      cw = rvrt + quat_mul_vec(q, lw);
      lp = cp; lv_ = cv; lw = cw;               // the "local" loss terms use the replaced joint 0 (train.py:305-308)
    } else {
      const int pa = lv.par[i];
      M3 pm; V3 pp, pv, pw;
      if constexpr (LM) {                       // the parent's level left them in LDS
        const float* m = prev + lv.pos[pa] * LOSS_MSG;
        for (int k = 0; k < 9; ++k) pm.m[k] = m[k * 64];
        pp = v3(m[9 * 64], m[10 * 64], m[11 * 64]);
        pv = v3(m[12 * 64], m[13 * 64], m[14 * 64]);
        pw = v3(m[15 * 64], m[16 * 64], m[17 * 64]);
      } else {
      for (int k = 0; k < 9; ++k) pm.m[k] = FE(F, o.cmat + 9 * pa + k);
      pp = v3(FE(F, o.cpos + 3 * pa), FE(F, o.cpos + 3 * pa + 1), FE(F, o.cpos + 3 * pa + 2));
      pv = v3(FE(F, o.cvel + 3 * pa), FE(F, o.cvel + 3 * pa + 1), FE(F, o.cvel + 3 * pa + 2));
      pw = v3(FE(F, o.cvrt + 3 * pa), FE(F, o.cvrt + 3 * pa + 1), FE(F, o.cvrt + 3 * pa + 2));
      }
      V3 rp = mv(pm, lp);
      cp = pp + rp;
      cm = mm(pm, L);
      cw = pw + mv(pm, lw);
      cv = pv + mv(pm, lv_) + cross(pw, rp);
    }
    if constexpr (LM) {
      float* m = mine + (kk - lv.start[l]) * LOSS_MSG;
      for (int k = 0; k < 9; ++k) m[k * 64] = cm.m[k];
      m[9 * 64] = cp.x; m[10 * 64] = cp.y; m[11 * 64] = cp.z;
      m[12 * 64] = cv.x; m[13 * 64] = cv.y; m[14 * 64] = cv.z;
      m[15 * 64] = cw.x; m[16 * 64] = cw.y; m[17 * 64] = cw.z;
    }
    if (live) {
    FE(F, o.lpos + 3 * i) = lp.x; FE(F, o.lpos + 3 * i + 1) = lp.y; FE(F, o.lpos + 3 * i + 2) = lp.z;
    FE(F, o.lvel + 3 * i) = lv_.x; FE(F, o.lvel + 3 * i + 1) = lv_.y; FE(F, o.lvel + 3 * i + 2) = lv_.z;
    FE(F, o.lvrt + 3 * i) = lw.x; FE(F, o.lvrt + 3 * i + 1) = lw.y; FE(F, o.lvrt + 3 * i + 2) = lw.z;
    FE(F, o.cpos + 3 * i) = cp.x; FE(F, o.cpos + 3 * i + 1) = cp.y; FE(F, o.cpos + 3 * i + 2) = cp.z;
    FE(F, o.cvel + 3 * i) = cv.x; FE(F, o.cvel + 3 * i + 1) = cv.y; FE(F, o.cvel + 3 * i + 2) = cv.z;
    FE(F, o.cvrt + 3 * i) = cw.x; FE(F, o.cvrt + 3 * i + 1) = cw.y; FE(F, o.cvrt + 3 * i + 2) = cw.z;
    for (int k = 0; k < 9; ++k) FE(F, o.cmat + 9 * i + k) = cm.m[k];
    }
   }
   if constexpr (LM) lds_barrier();   // the next level reads these joints' transforms from LDS; the table stores drain on their own
   else __syncthreads();              // ... from the tables (same CU: L1 is coherent within the workgroup)
  }
  };
  if (lv.maxw <= LOSS_LW && lds_ok) walk(std::true_type{});      // (lds_ok == 0: launched without the message buffers)
  else walk(std::false_type{});
}

// term of a feature row e: id/weight of the plain term and of the finite-difference term (id2 < 0: none)
__device__ __forceinline__ void term_of(int e, int J, int& id, float& w, int& id2, float& w2, int& size) {
  const Off o = offsets(J);
  id2 = -1; w2 = 0.f;
  if (e < o.rmat) { id = 0; w = 0.1f; size = 3; }
  else if (e < o.rvel) { id = 1; w = 10.f; size = 9; }
  else if (e < o.rvrt) { id = 2; w = 0.1f; size = 3; }
  else if (e < o.lpos) { id = 3; w = 5.f; size = 3; }
  else if (e < o.ltxy) { id = 4; w = 15.f; size = 3 * J; id2 = 12; w2 = 7.f; }
  else if (e < o.lvel) { id = 5; w = 15.f; size = 6 * J; id2 = 13; w2 = 8.f; }
  else if (e < o.lvrt) { id = 6; w = 10.f; size = 3 * J; }
  else if (e < o.cpos) { id = 7; w = 7.f; size = 3 * J; }
  else if (e < o.cmat) { id = 8; w = 0.1f; size = 3 * J; id2 = 14; w2 = 0.06f; }
  else if (e < o.cvel) { id = 9; w = 3.f; size = 9 * J; id2 = 15; w2 = 1.25f; }
  else if (e < o.cvrt) { id = 10; w = 0.06f; size = 3 * J; }
  else if (e < o.gaze) { id = 11; w = 1.25f; size = 3 * J; }
  else { id = 16; w = 10.f; size = 3; }
}

__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// one block = one feature row e, looping over all frames (coalesced along the frame axis);
// writes G = dLoss/dF_O and accumulates the row's share of its (one or two) loss terms
__global__ __launch_bounds__(256) void loss_terms_k(ZeggsLossDims d, const float* FO, const float* FW, float* G,
                                                     float* terms, float gscale) {
  __shared__ float red[16];
  const long NF = (long)d.B * d.T;
  const int e = blockIdx.x;
  int id, id2, size; float w, w2;
  term_of(e, d.J, id, w, id2, w2, size);
  const float n1 = (float)d.B * d.T * size, n2 = (float)d.B * (d.T - 1) * size;
  const float* fo = FO + (long)e * NF;
  const float* fw = FW + (long)e * NF;
  float s1 = 0.f, s2 = 0.f;
  for (long f = threadIdx.x; f < NF; f += blockDim.x) {
    const int t = (int)(f % d.T);
    const float o0 = fo[f], w0 = fw[f];
    const float v = w * (o0 - w0);
    s1 += fabsf(v);
    float g = w * sgn(v) / n1;
    if (id2 >= 0 && d.T > 1) {
      if (t + 1 < d.T) {
        float dd = w2 * ((fo[f + 1] - o0) / d.dt - (fw[f + 1] - w0) / d.dt);
        s2 += fabsf(dd);
        g -= w2 * sgn(dd) / (d.dt * n2);
      }
      if (t > 0) {
        float dd = w2 * ((o0 - fo[f - 1]) / d.dt - (w0 - fw[f - 1]) / d.dt);
        g += w2 * sgn(dd) / (d.dt * n2);
      }
    }
    G[(long)e * NF + f] = g * gscale / 18.0f;
  }
  s1 = block_sum(s1, red);
  if (id2 >= 0) s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {
    atomicAdd(terms + id, s1 / n1);
    if (id2 >= 0 && d.T > 1) atomicAdd(terms + id2, s2 / n2);
  }
}

// The same, four consecutive frames per thread (16-byte loads / stores; T % 4 == 0, so a thread's frames lie in one window) and
// LOSS_TSPLIT workgroups per feature row: the scalar form moved 245 MB in 110 us (2.2x its HBM bound: 2 496 workgroups are 1.2
// rounds of the chip, 4-byte accesses); identical G (same expression per element).  The workgroups' term sums go to a partials
// table that loss_kl_final_k adds up in LDS: 5 000 float atomics onto the 18 words of ONE cache line were what the first version
// of this kernel spent its time on (122 us, no faster than the scalar form).
__global__ __launch_bounds__(256) void loss_terms4_k(ZeggsLossDims d, const float* FO, const float* FW, float* G,
                                                      float* partial, float gscale) {
  __shared__ float red[16];
  const long NF = (long)d.B * d.T, NQ = NF / 4;
  const int e = blockIdx.x;
  int id, id2, size; float w, w2;
  term_of(e, d.J, id, w, id2, w2, size);
  const float n1 = (float)d.B * d.T * size, n2 = (float)d.B * (d.T - 1) * size;
  const float* fo = FO + (long)e * NF;
  const float* fw = FW + (long)e * NF;
  const bool diff = id2 >= 0 && d.T > 1;
  float s1 = 0.f, s2 = 0.f;
  const long q0 = NQ * blockIdx.y / LOSS_TSPLIT, q1 = NQ * (blockIdx.y + 1) / LOSS_TSPLIT;
  for (long q = q0 + threadIdx.x; q < q1; q += blockDim.x) {
    const long f0 = 4 * q;
    const int t0 = (int)(f0 % d.T);
    const f4 o = *(const f4*)(fo + f0), wv = *(const f4*)(fw + f0);
    float op = 0.f, wp = 0.f, on = 0.f, wn = 0.f;        // neighbours across the thread's edges (inside the window)
    if (diff && t0 > 0) { op = fo[f0 - 1]; wp = fw[f0 - 1]; }
    if (diff && t0 + 4 < d.T) { on = fo[f0 + 4]; wn = fw[f0 + 4]; }
    f4 g;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = t0 + i;
      const float o0 = o[i], w0 = wv[i];
      const float v = w * (o0 - w0);
      s1 += fabsf(v);
      float gi = w * sgn(v) / n1;
      if (diff) {
        if (t + 1 < d.T) {
          const float o1 = i < 3 ? o[i < 3 ? i + 1 : 3] : on, w1 = i < 3 ? wv[i < 3 ? i + 1 : 3] : wn;
          const float dd = w2 * ((o1 - o0) / d.dt - (w1 - w0) / d.dt);
          s2 += fabsf(dd);
          gi -= w2 * sgn(dd) / (d.dt * n2);
        }
        if (t > 0) {
          const float om = i > 0 ? o[i > 0 ? i - 1 : 0] : op, wm = i > 0 ? wv[i > 0 ? i - 1 : 0] : wp;
          const float dd = w2 * ((o0 - om) / d.dt - (w0 - wm) / d.dt);
          gi += w2 * sgn(dd) / (d.dt * n2);
        }
      }
      g[i] = gi * gscale / 18.0f;
    }
    *(f4*)(G + (long)e * NF + f0) = g;
  }
  s1 = block_sum(s1, red);
  if (id2 >= 0) s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {
    float* ps = partial + ((long)e * LOSS_TSPLIT + blockIdx.y) * 2;
    ps[0] = s1 / n1;
    ps[1] = diff ? s2 / n2 : 0.f;
  }
}

// backward through FK / first-joint transform / orthogonalisation (prediction side), thread per frame.
// Consumes G in place.  Level-parallel like the forward kernel, deepest level first; a joint never writes another
// joint's rows: it GATHERS the totals of its character-space gradients (its own rows + its children's messages, in
// descending child order = the summation order of a sequential sweep) and leaves its message to its parent in its own
// c* rows (cpos / cvel rows already are the message; cmat / cvrt rows are overwritten).  Writes dpose[6:] (transposed
// table DPT), drpos, DQ (grad wrt rrot_f from everything except the root-velocity rotation) and leaves the total grads
// wrt rvel / rvrt in G's rvel / rvrt rows.
struct Children { int start[MAXJ + 1]; int idx[MAXJ]; };
__device__ void build_children(Children& Cn, const int* parents, int J) {      // parents: the LDS copy (Levels.par)
  for (int j = threadIdx.x; j <= J; j += blockDim.x) Cn.start[j] = 0;
  __syncthreads();
  for (int j = threadIdx.x; j < J; j += blockDim.x)
    if (parents[j] >= 0) atomicAdd(&Cn.start[parents[j] + 1], 1);
  __syncthreads();
  if (threadIdx.x == 0)
    for (int j = 0; j < J; ++j) Cn.start[j + 1] += Cn.start[j];
  __syncthreads();
  for (int j = threadIdx.x; j < J; j += blockDim.x) {   // children of a parent in DESCENDING index order
    const int pa = parents[j];
    if (pa < 0) continue;
    int pos = Cn.start[pa];
    for (int q = J - 1; q > j; --q) pos += (parents[q] == pa);
    Cn.idx[pos] = j;
  }
  __syncthreads();
}

// what a joint's backward reads from the tables: nothing of it depends on the level below, so the LDS walk fetches the NEXT
// joint's while it waits for this one's messages (JIn of joint i of the next level in flight across the barrier)
struct JIn { M3 pm, L, gcm; V3 pw, lp, lv, lw, gcp, gcv, gcw, glp, glv, glw, x, yi, gx, gyi; };

__global__ __launch_bounds__(64 * ZEGGS_LOSS_WAVES) void loss_frame_bwd_k(ZeggsLossDims d, const int* parents, FrameIO io, const float* gaze,
                                                         const float* PT, const float* F, const float* LM, float* G,
                                                         float* DPT, float* drpos, float* DQ, int lds_ok) {
  __shared__ Levels lv;
  __shared__ Children ch;
  const long NF = (long)d.B * d.T;
  build_levels(lv, parents, d.J);
  build_children(ch, lv.par, d.J);
  const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6, lane = threadIdx.x & 63;
  const long gid = (long)blockIdx.x * 64 + lane;
  const bool live = gid < NF;
  const long f = live ? gid : 0;
  const int J = d.J;
  const Off o = offsets(J);
  const Col p{PT, NF, f};
  const ColW dp{DPT, NF, f};
  const float* rq = io.rrot + f * 4;
  Q4 q = Q4{rq[0], rq[1], rq[2], rq[3]};
  V3 rpos = v3(io.rpos[f * 3], io.rpos[f * 3 + 1], io.rpos[f * 3 + 2]);
  const Col lpos = p + 6, ltxy = p + (6 + 3 * J), lvel = p + (6 + 9 * J), lvrt = p + (6 + 12 * J);
  const ColW dlpos = dp + 6, dltxy = dp + (6 + 3 * J), dlvel = dp + (6 + 9 * J), dlvrt = dp + (6 + 12 * J);
  auto ld3 = [&](const float* A, int e) { return v3(A[(long)e * NF + f], A[(long)(e + 1) * NF + f], A[(long)(e + 2) * NF + f]); };
  auto st3 = [&](float* A, int e, V3 v) { if (live) { A[(long)e * NF + f] = v.x; A[(long)(e + 1) * NF + f] = v.y; A[(long)(e + 2) * NF + f] = v.z; } };
  auto ld9 = [&](const float* A, int e) { M3 m; for (int k = 0; k < 9; ++k) m.m[k] = A[(long)(e + k) * NF + f]; return m; };
  auto st9 = [&](float* A, int e, const M3& m) { if (live) for (int k = 0; k < 9; ++k) A[(long)(e + k) * NF + f] = m.m[k]; };
  auto put3 = [&](const ColW& c, int e, V3 v) { if (live) { c[e] = v.x; c[e + 1] = v.y; c[e + 2] = v.z; } };

  // a joint's table reads (pa < 0, joint 0: no parent rows)
  auto fetch = [&](int i) {
    JIn a;
    const int pa = lv.par[i];
    if (pa >= 0) { a.pm = ld9(F, o.cmat + 9 * pa); a.pw = ld3(F, o.cvrt + 3 * pa); }
    a.lp = v3(lpos[3 * i], lpos[3 * i + 1], lpos[3 * i + 2]);
    a.lv = v3(lvel[3 * i], lvel[3 * i + 1], lvel[3 * i + 2]);
    a.lw = v3(lvrt[3 * i], lvrt[3 * i + 1], lvrt[3 * i + 2]);
    a.L = ld9(LM, 9 * i);
    a.gcp = ld3(G, o.cpos + 3 * i); a.gcv = ld3(G, o.cvel + 3 * i); a.gcw = ld3(G, o.cvrt + 3 * i); a.gcm = ld9(G, o.cmat + 9 * i);
    a.glp = ld3(G, o.lpos + 3 * i); a.glv = ld3(G, o.lvel + 3 * i); a.glw = ld3(G, o.lvrt + 3 * i);
    a.x = v3(ltxy[6 * i], ltxy[6 * i + 1], ltxy[6 * i + 2]); a.yi = v3(ltxy[6 * i + 3], ltxy[6 * i + 4], ltxy[6 * i + 5]);
    a.gx = ld3(G, o.ltxy + 6 * i); a.gyi = ld3(G, o.ltxy + 6 * i + 3);
    return a;
  };
  // orthogonalisation backward for joint i given the grad of its local matrix; adds the direct ltxy-term grad
  auto orth_bwd = [&](int i, const M3& gL, const JIn& a) {
    V3 x = a.x, yi = a.yi;
    V3 z = cross(x, yi), y = cross(z, x);
    V3 gxn = v3(gL.m[0], gL.m[3], gL.m[6]), gyn = v3(gL.m[1], gL.m[4], gL.m[7]), gzn = v3(gL.m[2], gL.m[5], gL.m[8]);
    V3 gx = normalize_bwd(x, gxn, 1e-10f);
    V3 gy = normalize_bwd(y, gyn, 1e-10f);
    V3 gz = normalize_bwd(z, gzn, 1e-10f);
    gz = gz + cross(x, gy);          // y = z x x
    gx = gx + cross(gy, z);
    gx = gx + cross(yi, gz);         // z = x x yi
    V3 gyi = cross(gz, x);
    put3(dltxy, 6 * i, gx + a.gx);
    put3(dltxy, 6 * i + 3, gyi + a.gyi);
  };
  // totals of joint i's character-space gradients: own rows (already in a) + children's messages (msgs != nullptr: the
  // children's level left them in LDS -- (dm, gcp, gcv, dw), 18 rows of 64 frames per joint -- instead of in their c* rows of G)
  auto gather = [&](int i, JIn& a, const float* msgs) {
    for (int c = ch.start[i]; c < ch.start[i + 1]; ++c) {
      const int j = ch.idx[c];
      if (msgs) {
        const float* m = msgs + lv.pos[j] * LOSS_MSG;
        a.gcp = a.gcp + v3(m[9 * 64], m[10 * 64], m[11 * 64]); a.gcv = a.gcv + v3(m[12 * 64], m[13 * 64], m[14 * 64]);
        a.gcw = a.gcw + v3(m[15 * 64], m[16 * 64], m[17 * 64]);
        for (int k = 0; k < 9; ++k) a.gcm.m[k] += m[k * 64];
      } else {
        a.gcp = a.gcp + ld3(G, o.cpos + 3 * j); a.gcv = a.gcv + ld3(G, o.cvel + 3 * j); a.gcw = a.gcw + ld3(G, o.cvrt + 3 * j);
        const M3 m = ld9(G, o.cmat + 9 * j);
        for (int k = 0; k < 9; ++k) a.gcm.m[k] += m.m[k];
      }
    }
  };
  // joint i (not the root joint): message to the parent (dm, gcp, gcv, dw) -> `mine` (LDS) or its own c* rows of G; local gradients
  auto joint = [&](int i, JIn& a, const float* below, float* mine) {
    const V3 rp = mv(a.pm, a.lp);
    gather(i, a, below);
    M3 dm;
    for (int k = 0; k < 9; ++k) dm.m[k] = 0.f;
    // cvel_i = cvel_p + pm lv + pw x rp
    add_outer(dm, a.gcv, a.lv);
    V3 glv = mtv(a.pm, a.gcv);
    V3 dw = cross(rp, a.gcv);
    V3 grp = cross(a.gcv, a.pw);
    // cvrt_i = cvrt_p + pm lw
    dw = dw + a.gcw;
    add_outer(dm, a.gcw, a.lw);
    V3 glw = mtv(a.pm, a.gcw);
    // cmat_i = pm L
    M3 t1 = mmt(a.gcm, a.L);
    for (int k = 0; k < 9; ++k) dm.m[k] += t1.m[k];
    M3 gL = mtm(a.pm, a.gcm);
    // cpos_i = cpos_p + rp
    grp = grp + a.gcp;
    add_outer(dm, grp, a.lp);
    V3 glp = mtv(a.pm, grp);
    if (mine) {
      for (int k = 0; k < 9; ++k) mine[k * 64] = dm.m[k];
      mine[9 * 64] = a.gcp.x; mine[10 * 64] = a.gcp.y; mine[11 * 64] = a.gcp.z;
      mine[12 * 64] = a.gcv.x; mine[13 * 64] = a.gcv.y; mine[14 * 64] = a.gcv.z;
      mine[15 * 64] = dw.x; mine[16 * 64] = dw.y; mine[17 * 64] = dw.z;
    } else {
      st9(G, o.cmat + 9 * i, dm);
      st3(G, o.cpos + 3 * i, a.gcp); st3(G, o.cvel + 3 * i, a.gcv); st3(G, o.cvrt + 3 * i, dw);
    }
    // local features (direct "local" loss terms + FK)
    put3(dlpos, 3 * i, glp + a.glp);
    put3(dlvel, 3 * i, glv + a.glv);
    put3(dlvrt, 3 * i, glw + a.glw);
    orth_bwd(i, gL, a);
  };

  const bool lds_walk = lv.maxw <= LOSS_LW && lds_ok;
  JIn root;                                   // joint 0's reads (wave 0): fetched under the walk as well
  if (lds_walk) {
    JIn cur;
    bool have = false;
    for (int l = lv.nlevels - 1; l >= 1; --l) {
      float* mine = loss_msg + (l & 1) * (LOSS_LW * LOSS_MSG) + lane;
      const float* below = loss_msg + ((l & 1) ^ 1) * (LOSS_LW * LOSS_MSG) + lane;
      for (int kk = lv.start[l] + wave; kk < lv.start[l + 1]; kk += nwaves) {
        const int i = lv.joint[kk];
        if (!have) cur = fetch(i);
        // this wave's next joint: the next round of this level, or its first one of the level above (wave 0 at level 1: the root)
        int nk = kk + nwaves;
        bool nv = nk < lv.start[l + 1];
        if (!nv) { nk = lv.start[l - 1] + wave; nv = nk < lv.start[l]; }
        JIn nxt;
        if (nv) nxt = fetch(lv.joint[nk]);
        joint(i, cur, below, mine + (kk - lv.start[l]) * LOSS_MSG);
        cur = nxt; have = nv;
      }
      lds_barrier();
    }
    if (wave != 0) return;
    root = have ? cur : fetch(0);             // (level 0 is the root joint alone: wave 0's "next" at level 1 was joint 0)
  } else {
    for (int l = lv.nlevels - 1; l >= 1; --l) {
      for (int kk = lv.start[l] + wave; kk < lv.start[l + 1]; kk += nwaves) {
        const int i = lv.joint[kk];
        JIn a = fetch(i);
        joint(i, a, nullptr, nullptr);
      }
      __syncthreads();
    }
    if (wave != 0) return;
    root = fetch(0);
  }
  // ---- joint 0: world-space replacement (train.py:296-308)
  {
    V3 rvrt = ld3(F, o.rvrt);
    M3 R = quat_to_xform(q);
    M3 L = root.L;
    V3 lp = root.lp, lv_ = root.lv, lw = root.lw;
    V3 rl = quat_mul_vec(q, lp);
    gather(0, root, lds_walk ? loss_msg + (LOSS_LW * LOSS_MSG) + lane : nullptr);      // level 1's messages: buffer 1
    V3 gcp = root.gcp, gcv = root.gcv, gcw = root.gcw;
    M3 gcm = root.gcm;
    gcp = gcp + root.glp;      // joint 0 appears in the c* and in the "local" terms
    gcv = gcv + root.glv;
    gcw = gcw + root.glw;
    V3 g_rpos = ld3(G, o.rpos) + gcp;
    V3 g_rvel = ld3(G, o.rvel) + gcv;
    V3 g_rvrt = ld3(G, o.rvrt) + gcw + cross(rl, gcv);
    V3 g_rl = gcp + cross(gcv, rvrt);
    Q4 dq = Q4{0.f, 0.f, 0.f, 0.f}, dqt; V3 dv;
    qmv_bwd(q, lp, g_rl, dqt, dv);
    dq.w += dqt.w; dq.x += dqt.x; dq.y += dqt.y; dq.z += dqt.z;
    put3(dlpos, 0, dv);
    qmv_bwd(q, lv_, gcv, dqt, dv);
    dq.w += dqt.w; dq.x += dqt.x; dq.y += dqt.y; dq.z += dqt.z;
    put3(dlvel, 0, dv);
    qmv_bwd(q, lw, gcw, dqt, dv);
    dq.w += dqt.w; dq.x += dqt.x; dq.y += dqt.y; dq.z += dqt.z;
    put3(dlvrt, 0, dv);
    // lmat0w = R L
    M3 gR = mmt(gcm, L);
    M3 gL = mtm(R, gcm);
    orth_bwd(0, gL, root);
    M3 gRm = ld9(G, o.rmat);
    for (int k = 0; k < 9; ++k) gR.m[k] += gRm.m[k];
    dqt = quat_to_xform_bwd(q, gR);
    dq.w += dqt.w; dq.x += dqt.x; dq.y += dqt.y; dq.z += dqt.z;
    // gaze direction: gd = qmv(q^-1, n), n = v/(|v|+1e-8), v = gaze - rpos
    const float* gz = gaze + f * 3;
    V3 v = v3(gz[0], gz[1], gz[2]) - rpos;
    float inv = 1.f / (vnorm(v) + 1e-8f);
    V3 dn;
    qmv_bwd(quat_inv(q), inv * v, ld3(G, o.gaze), dqt, dn);
    dq.w += dqt.w; dq.x -= dqt.x; dq.y -= dqt.y; dq.z -= dqt.z;
    g_rpos = g_rpos - normalize_bwd(v, dn, 1e-8f);
    if (live) {
      drpos[f * 3] = g_rpos.x; drpos[f * 3 + 1] = g_rpos.y; drpos[f * 3 + 2] = g_rpos.z;
      DQ[f * 4] = dq.w; DQ[f * 4 + 1] = dq.x; DQ[f * 4 + 2] = dq.y; DQ[f * 4 + 3] = dq.z;
    }
    st3(G, o.rvel, g_rvel);
    st3(G, o.rvrt, g_rvrt);
  }
}

// root velocities are rotated by the PREVIOUS frame's root rotation (train.py:281-286): finish dpose[0:6], drrot
__global__ void loss_rootvel_bwd_k(ZeggsLossDims d, FrameIO io, const float* G, const float* DQ, float* dpose,
                                   float* drrot) {
  const long NF = (long)d.B * d.T;
  const long f = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= NF) return;
  const int t = (int)(f % d.T), J = d.J, PO = 6 + 15 * J;
  const Off o = offsets(J);
  auto ldq = [&](long fr) { const float* r = io.rrot + fr * 4; return Q4{r[0], r[1], r[2], r[3]}; };
  auto ld3 = [&](int e, long fr) { return v3(G[(long)e * NF + fr], G[(long)(e + 1) * NF + fr], G[(long)(e + 2) * NF + fr]); };
  Q4 acc = Q4{DQ[f * 4], DQ[f * 4 + 1], DQ[f * 4 + 2], DQ[f * 4 + 3]};
  // own frame: value-gradient (and the quaternion gradient when the frame rotates itself, t == 0)
  {
    const float* p = io.pose + f * PO;
    Q4 qp = ldq(t > 0 ? f - 1 : f);
    Q4 dq1, dq2; V3 dv1, dv2;
    qmv_bwd(qp, v3(p[0], p[1], p[2]), ld3(o.rvel, f), dq1, dv1);
    qmv_bwd(qp, v3(p[3], p[4], p[5]), ld3(o.rvrt, f), dq2, dv2);
    float* dp = dpose + f * PO;
    dp[0] = dv1.x; dp[1] = dv1.y; dp[2] = dv1.z; dp[3] = dv2.x; dp[4] = dv2.y; dp[5] = dv2.z;
    if (t == 0) { acc.w += dq1.w + dq2.w; acc.x += dq1.x + dq2.x; acc.y += dq1.y + dq2.y; acc.z += dq1.z + dq2.z; }
  }
  if (t + 1 < d.T) {   // the next frame's velocities are rotated by this frame's rotation
    const float* p = io.pose + (f + 1) * PO;
    Q4 q = ldq(f), dq1, dq2; V3 dv;
    qmv_bwd(q, v3(p[0], p[1], p[2]), ld3(o.rvel, f + 1), dq1, dv);
    qmv_bwd(q, v3(p[3], p[4], p[5]), ld3(o.rvrt, f + 1), dq2, dv);
    acc.w += dq1.w + dq2.w; acc.x += dq1.x + dq2.x; acc.y += dq1.y + dq2.y; acc.z += dq1.z + dq2.z;
  }
  drrot[f * 4] = acc.w; drrot[f * 4 + 1] = acc.x; drrot[f * 4 + 2] = acc.y; drrot[f * 4 + 3] = acc.z;
}

// KL term (modules.py:778-779) + final sum/18 (train.py:402-421); one block
__global__ __launch_bounds__(1024) void loss_kl_final_k(const float* mu, const float* logvar, int n, int B, float klw,
                                                        float* terms, float* dmu, float* dlogvar, float gscale,
                                                        const float* partial, int nrows, int J) {
  __shared__ float red[16];
  __shared__ float tsum[18];
  if (partial) {        // term sums of loss_terms4_k's workgroups (LDS atomics: 18 words, one workgroup)
    if (threadIdx.x < 18) tsum[threadIdx.x] = 0.f;
    __syncthreads();
    // consecutive rows belong to the same term (a wave sees one to three of them): summed per term across the wave first -- 64 lanes'
    // atomics onto ONE LDS word serialise, which made this reduction 13 of the kernel's 18 us
    const int lane = threadIdx.x & 63, total = nrows * LOSS_TSPLIT;
    auto add_by_key = [&](int key, float v) {       // key < 0: lane takes no part
      unsigned long long todo = __ballot(key >= 0);
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k = __shfl(key, leader, 64);
        const bool mine = key == k;
        const float sum = wave_sum(mine ? v : 0.f);
        if (lane == leader) atomicAdd(&tsum[k], sum);
        todo &= ~__ballot(mine);
      }
    };
    for (int i0 = threadIdx.x - lane; i0 < total; i0 += blockDim.x) {
      const int i = i0 + lane;
      int id = -1, id2 = -1, size; float w, w2;
      if (i < total) term_of(i / LOSS_TSPLIT, J, id, w, id2, w2, size);
      add_by_key(id, i < total ? partial[2 * i] : 0.f);
      add_by_key(id2, id2 >= 0 ? partial[2 * i + 1] : 0.f);
    }
    __syncthreads();
    if (threadIdx.x < 17) terms[threadIdx.x] += tsum[threadIdx.x];
    __syncthreads();
  }
  float s = 0.f;
  if (mu && klw > 0.f) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      float m = mu[i], lv = logvar[i], ev = expf(lv);
      s += 1.f + lv - m * m - ev;
      dmu[i] = klw * m / (float)n * gscale / 18.0f;
      dlogvar[i] = klw * -0.5f * (1.f - ev) / (float)n * gscale / 18.0f;
    }
  } else if (dmu) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) { dmu[i] = 0.f; dlogvar[i] = 0.f; }
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    float kl = (mu && klw > 0.f) ? klw * (-0.5f * s / (float)n) : 0.f;
    terms[17] = kl;
    float tot = 0.f;
    for (int i = 0; i < 18; ++i) tot += terms[i];
    terms[18] = tot / 18.0f;
  }
}

}  // namespace

int g_loss_lds = 1;      // zeggs_set_option("loss_lds", 0): the table walk in global memory (what a part without 160 KB of LDS gets; tests)
// the frame kernels' message buffers (dynamic LDS above the 64 KB default): per DEVICE (the function attribute is), and a part
// that refuses them (less than 160 KB of LDS per workgroup) walks the tables in global memory instead (lds_ok = 0, no dynamic LDS)
static int loss_lds_ready() {
  static int ok[64];      // 0 unknown, 1 granted, 2 refused
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (ok[dev] == 0) {
    const bool g = hipFuncSetAttribute((const void*)loss_frame_fwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LOSS_LDS_BYTES) == hipSuccess &&
                   hipFuncSetAttribute((const void*)loss_frame_bwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LOSS_LDS_BYTES) == hipSuccess;
    if (!g) (void)hipGetLastError();
    ok[dev] = g ? 1 : 2;
  }
  return ok[dev] == 1;
}

extern "C" size_t zeggs_loss_workspace_bytes(const ZeggsLossDims* d) {
  Arena a(nullptr, 0);
  carve_loss(*d, a);
  return a.off + 256;
}

// The ground-truth half of the feature pass (pose rows transposed, forward kinematics of the truth side) depends on the batch
// only: a training loop may run it ahead of the step, on any stream, into the workspace the loss call of that batch will be
// given, and pass truth_prepared = 1 there.
extern "C" int zeggs_loss_prepare_truth(const ZeggsLossDims* dp, const int* parents, const float* w_pose, const float* w_rpos,
                                        const float* w_rrot, const float* gaze, void* ws, size_t ws_bytes, void* stream) {
  const ZeggsLossDims& d = *dp;
  hipStream_t s = (hipStream_t)stream;
  Arena a(ws, ws_bytes);
  LossWs w = carve_loss(d, a);
  ZCHECK(a.ok(), "loss: workspace too small (%zu < %zu)", ws_bytes, a.off);
  ZCHECK(d.J >= 1 && d.J <= MAXJ && d.T >= 1 && d.B >= 1, "loss: bad dims");
  const int lds_ok = g_loss_lds ? loss_lds_ready() : 0;
  const long NF = (long)d.B * d.T;
  const int PO = 6 + 15 * d.J;
  FrameIO ioW{w_pose, w_rpos, w_rrot};
  hipLaunchKernelGGL(transpose_k, dim3((unsigned)cdiv(NF, 64), (unsigned)cdiv(PO, 64)), dim3(256), 0, s, w.PT1, w_pose, NF, PO);
  hipLaunchKernelGGL(loss_frame_fwd_k, dim3(cdiv(NF, 64)), dim3(64 * ZEGGS_LOSS_WAVES), lds_ok ? LOSS_LDS_BYTES : 0, s, d, parents, ioW, ioW, w.PT1, w.PT1, gaze, w.FW, w.FW,
                     w.LM, NF, 2 * NF, lds_ok);
  ZLAUNCH_CHECK("loss_prepare_truth");
  return 0;
}

extern "C" int zeggs_loss_fwd_bwd(const ZeggsLossDims* dp, const int* parents, const float* o_pose, const float* o_rpos,
                                  const float* o_rrot, const float* w_pose, const float* w_rpos, const float* w_rrot,
                                  const float* gaze, const float* mu, const float* logvar, float kl_weight, float* terms,
                                  float* dpose, float* drpos, float* drrot, float* dmu, float* dlogvar, float gscale,
                                  void* ws, size_t ws_bytes, void* stream) {
  return zeggs_loss_fwd_bwd_ex(dp, parents, o_pose, o_rpos, o_rrot, w_pose, w_rpos, w_rrot, gaze, mu, logvar, kl_weight, terms,
                               dpose, drpos, drrot, dmu, dlogvar, gscale, ws, ws_bytes, stream, 0);
}
extern "C" int zeggs_loss_fwd_bwd_ex(const ZeggsLossDims* dp, const int* parents, const float* o_pose, const float* o_rpos,
                                     const float* o_rrot, const float* w_pose, const float* w_rpos, const float* w_rrot,
                                     const float* gaze, const float* mu, const float* logvar, float kl_weight, float* terms,
                                     float* dpose, float* drpos, float* drrot, float* dmu, float* dlogvar, float gscale,
                                     void* ws, size_t ws_bytes, void* stream, int truth_prepared) {
  const ZeggsLossDims& d = *dp;
  hipStream_t s = (hipStream_t)stream;
  Arena a(ws, ws_bytes);
  LossWs w = carve_loss(d, a);
  ZCHECK(a.ok(), "loss: workspace too small (%zu < %zu)", ws_bytes, a.off);
  ZCHECK(d.J >= 1 && d.T >= 1 && d.B >= 1, "loss: bad dims");
  const int lds_ok = g_loss_lds ? loss_lds_ready() : 0;
  const long NF = (long)d.B * d.T;
  const Off o = offsets(d.J);
  FrameIO ioO{o_pose, o_rpos, o_rrot}, ioW{w_pose, w_rpos, w_rrot};
  ZTRY(k_fill(terms, 19, 0.f, s));
  const int PO = 6 + 15 * d.J;
  const dim3 tg((unsigned)cdiv(NF, 64), (unsigned)cdiv(PO, 64));
  hipLaunchKernelGGL(transpose_k, tg, dim3(256), 0, s, w.PT0, o_pose, NF, PO);
  if (!truth_prepared) hipLaunchKernelGGL(transpose_k, tg, dim3(256), 0, s, w.PT1, w_pose, NF, PO);
  ZCHECK(d.J <= MAXJ, "loss: more than %d joints", MAXJ);
  // (truth_prepared: zeggs_loss_prepare_truth has filled PT1 / FW of THIS workspace; only the prediction side is left)
  const long gend = truth_prepared ? NF : 2 * NF;
  hipLaunchKernelGGL(loss_frame_fwd_k, dim3(cdiv(gend, 64)), dim3(64 * ZEGGS_LOSS_WAVES), lds_ok ? LOSS_LDS_BYTES : 0, s, d, parents, ioO, ioW, w.PT0, w.PT1, gaze, w.FO,
                     w.FW, w.LM, 0L, gend, lds_ok);
  ZLAUNCH_CHECK("loss_frame_fwd");
  if (d.T % 4 == 0 && ((uintptr_t)w.FO % 16 == 0) && ((uintptr_t)w.FW % 16 == 0) && ((uintptr_t)w.G % 16 == 0))      // (= vec4 below)
    hipLaunchKernelGGL(loss_terms4_k, dim3(o.n, LOSS_TSPLIT), dim3(256), 0, s, d, w.FO, w.FW, w.G, w.PS, gscale);
  else
    hipLaunchKernelGGL(loss_terms_k, dim3(o.n), dim3(256), 0, s, d, w.FO, w.FW, w.G, terms, gscale);
  ZLAUNCH_CHECK("loss_terms");
  const bool vec4 = d.T % 4 == 0 && ((uintptr_t)w.FO % 16 == 0) && ((uintptr_t)w.FW % 16 == 0) && ((uintptr_t)w.G % 16 == 0);
  hipLaunchKernelGGL(loss_kl_final_k, dim3(1), dim3(1024), 0, s, mu, logvar, d.B * d.S, d.B, kl_weight, terms, dmu, dlogvar,
                     gscale, vec4 ? w.PS : (const float*)nullptr, o.n, d.J);
  ZLAUNCH_CHECK("loss_kl_final");
  if (dpose) {
    hipLaunchKernelGGL(loss_frame_bwd_k, dim3(cdiv(NF, 64)), dim3(64 * ZEGGS_LOSS_WAVES), lds_ok ? LOSS_LDS_BYTES : 0, s, d, parents, ioO, gaze, w.PT0, w.FO, w.LM, w.G,
                       w.DPT, drpos, w.DQ, lds_ok);
    ZLAUNCH_CHECK("loss_frame_bwd");
    // back to [frame][PO] (columns 0..5 hold nothing yet: the root-velocity kernel below writes them)
    hipLaunchKernelGGL(transpose_k, dim3((unsigned)cdiv(PO, 64), (unsigned)cdiv(NF, 64)), dim3(256), 0, s, dpose, w.DPT,
                       (long)PO, (int)NF);
    ZLAUNCH_CHECK("loss_transpose");
    hipLaunchKernelGGL(loss_rootvel_bwd_k, dim3(cdiv(NF, 256)), dim3(256), 0, s, d, ioO, w.G, w.DQ, dpose, drrot);
    ZLAUNCH_CHECK("loss_rootvel_bwd");
  }
  return 0;
}
