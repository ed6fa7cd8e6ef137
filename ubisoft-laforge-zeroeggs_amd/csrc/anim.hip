// Animation pre-/post-processing either side of the decoder in generate_gesture():
//   * zeggs_anim_features : exemplar clip (BVH euler channels + positions) -> the 16 feature arrays of
//     preprocess_animation (reference ZEGGS/data_pipeline.py:90-228): FK, root projection on the ground plane,
//     median gaze target, finite-difference / helical velocities, character-space FK with velocities, two-axis
//     rotation encodings;
//   * zeggs_pose_to_bvh   : decoder output (root trajectory, local positions, two-axis rotations) -> BVH channels
//     (reference generate.py:389 from_xform(orthogonalize_from_xy), utils.py:47-87 write_bvh, quat.py:111-206).
// Everything is float64 like the reference's NumPy code; the work is per-frame / per-joint independent and
// HBM-bound (a few hundred bytes per joint-frame), so the kernels are plain one-thread-per-item maps.  The only
// cross-frame steps are the sign unrolling of the local quaternions (a sequential +/-1 recurrence per joint) and
// the median of the gaze target (exact radix select over the order-preserving 64-bit image of the doubles).
#include "../../include/zeggs_hip.h"
#include "common.h"

namespace {

struct D3 { double x, y, z; };
struct DQ { double w, x, y, z; };
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 operator*(double s, D3 a) { return D3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ D3 dcross(D3 a, D3 b) { return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ double ddot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ DQ dq_inv(DQ q) { return DQ{q.w, -q.x, -q.y, -q.z}; }
__device__ __forceinline__ DQ dq_mul(DQ a, DQ b) {   // quat.py mul: (aw bw - av.bv, aw bv + bw av + av x bv)
  return DQ{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + b.w * a.x + (a.y * b.z - a.z * b.y),
            a.w * b.y + b.w * a.y + (a.z * b.x - a.x * b.z), a.w * b.z + b.w * a.z + (a.x * b.y - a.y * b.x)};
}
__device__ __forceinline__ D3 dq_mul_vec(DQ q, D3 v) {
  const D3 qv = D3{q.x, q.y, q.z};
  const D3 t = 2.0 * dcross(qv, v);
  return v + q.w * t + dcross(qv, t);
}
__device__ __forceinline__ DQ dq_abs(DQ q) { return q.w > 0.0 ? q : DQ{-q.w, -q.x, -q.y, -q.z}; }
// helical (scaled angle-axis) = 2 log(q), quat.py log: atan2(|v|, w) / |v| * v, identity when |v| < eps
__device__ __forceinline__ D3 dq_to_helical(DQ q) {
  const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
  const double s = n < 1e-5 ? 1.0 : atan2(n, q.w) / n;
  return D3{2.0 * s * q.x, 2.0 * s * q.y, 2.0 * s * q.z};
}
__device__ __forceinline__ DQ dq_axis(double angle, int axis) {
  const double h = 0.5 * angle, s = sin(h), c = cos(h);
  return DQ{c, axis == 0 ? s : 0.0, axis == 1 ? s : 0.0, axis == 2 ? s : 0.0};
}
// Channel order of the BVH rotation channels, packed: axis of channel i (0 x, 1 y, 2 z) in bits 2 i .. 2 i + 1; 0 = "zyx" (every
// ZeroEGGS rig).  from_euler takes any order (quat.py:154-163: q = q(e0, axis0) * (q(e1, axis1) * q(e2, axis2))), to_euler the two
// the reference implements (quat.py:111-127: "zyx", "xzy"; it raises for the others, and so does the host side here).
constexpr int ORDER_ZYX = 2 | (1 << 2) | (0 << 4), ORDER_XZY = 0 | (2 << 2) | (1 << 4);
__host__ __device__ inline int order_code(int order) { return order == 0 ? ORDER_ZYX : order; }
__device__ __forceinline__ DQ dq_from_euler_deg(const double* e, int order) {
  const double r = 0.017453292519943295;
  return dq_mul(dq_axis(e[0] * r, order & 3), dq_mul(dq_axis(e[1] * r, (order >> 2) & 3), dq_axis(e[2] * r, (order >> 4) & 3)));
}
__device__ __forceinline__ void dq_to_euler_deg(DQ q, double* e, int order) {
  const double d = 57.29577951308232;
  if (order == ORDER_XZY) {      // quat.py:120-125
    double sz = 2.0 * (q.x * q.y + q.z * q.w);
    sz = sz > 1.0 ? 1.0 : (sz < -1.0 ? -1.0 : sz);
    e[0] = d * atan2(2.0 * (q.x * q.w - q.y * q.z), -q.x * q.x + q.y * q.y - q.z * q.z + q.w * q.w);
    e[1] = d * atan2(2.0 * (q.y * q.w - q.x * q.z), q.x * q.x - q.y * q.y - q.z * q.z + q.w * q.w);
    e[2] = d * asin(sz);
    return;
  }
  double sy = 2.0 * (q.w * q.y - q.z * q.x);      // "zyx", quat.py:114-119
  sy = sy > 1.0 ? 1.0 : (sy < -1.0 ? -1.0 : sy);
  e[0] = d * atan2(2.0 * (q.w * q.z + q.x * q.y), 1.0 - 2.0 * (q.y * q.y + q.z * q.z));
  e[1] = d * asin(sy);
  e[2] = d * atan2(2.0 * (q.w * q.x + q.y * q.z), 1.0 - 2.0 * (q.x * q.x + q.y * q.y));
}
// rotation taking direction a to direction b, normalised (quat.py between + normalize)
__device__ __forceinline__ DQ dq_between_n(D3 a, D3 b) {
  const D3 c = dcross(a, b);
  DQ q = DQ{sqrt(ddot(a, a) * ddot(b, b)) + ddot(a, b), c.x, c.y, c.z};
  const double n = sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  return DQ{q.w / n, q.x / n, q.y / n, q.z / n};
}
// rotation matrix (columns = the orthonormalised x, y, z axes of a two-axis encoding) -> quaternion;
// branch choice as quat.py from_xform (trace / largest diagonal element)
__device__ __forceinline__ DQ dq_from_xy(D3 x, D3 yin) {
  const double eps = 1e-10;
  D3 z = dcross(x, yin);
  D3 y = dcross(z, x);
  x = (1.0 / (sqrt(ddot(x, x)) + eps)) * x;
  y = (1.0 / (sqrt(ddot(y, y)) + eps)) * y;
  z = (1.0 / (sqrt(ddot(z, z)) + eps)) * z;
  // m[r][c]: column 0 = x, 1 = y, 2 = z
  const double m00 = x.x, m10 = x.y, m20 = x.z, m01 = y.x, m11 = y.y, m21 = y.z, m02 = z.x, m12 = z.y, m22 = z.z;
  const double tr = m00 + m11 + m22;
  const double a = m21 - m12, b = m02 - m20, c = m10 - m01, p = m01 + m10, q = m02 + m20, r = m12 + m21;
  if (tr > 0.0) {
    const double s = 0.5 / sqrt(fmax(tr + 1.0, eps));
    return DQ{0.25 / s, s * a, s * b, s * c};
  }
  if (m00 > m11 && m00 > m22) {
    const double s = 2.0 * sqrt(fmax(1.0 + m00 - m11 - m22, eps));
    return DQ{a / s, 0.25 * s, p / s, q / s};
  }
  if (m11 > m22) {
    const double s = 2.0 * sqrt(fmax(1.0 + m11 - m00 - m22, eps));
    return DQ{b / s, p / s, 0.25 * s, r / s};
  }
  const double s = 2.0 * sqrt(fmax(1.0 + m22 - m00 - m11, eps));
  return DQ{c / s, q / s, r / s, 0.25 * s};
}

__device__ __forceinline__ D3 ld3(const double* p) { return D3{p[0], p[1], p[2]}; }
__device__ __forceinline__ DQ ldq(const double* p) { return DQ{p[0], p[1], p[2], p[3]}; }
__device__ __forceinline__ void st3(double* p, D3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ void stq(double* p, DQ q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }

// ------------------------------------------------------------------ feature extraction
// 1. euler channels -> raw local quaternions and the dot product with the previous frame's raw quaternion
__global__ void anim_quat_k(const double* euler, double* lrot, double* dprev, int N, int J, int order) {
  const long n = (long)N * J;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const DQ q = dq_from_euler_deg(euler + i * 3, order);
    stq(lrot + i * 4, q);
    double d = 1.0;
    if (i >= J) {
      const DQ p = dq_from_euler_deg(euler + (i - J) * 3, order);
      d = q.w * p.w + q.x * p.x + q.y * p.y + q.z * p.z;
    }
    dprev[i] = d;
  }
}
// 2. sign unrolling (quat.py unroll): frame i is negated when its dot product with the ALREADY unrolled frame i-1
//    is negative: s_i = (s_{i-1} d_i < 0) ? -1 : +1.  One thread per joint walks the frames (loads are independent of
//    the recurrence, 8 in flight).
__global__ void anim_unroll_k(const double* dprev, double* sign, int N, int J) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= J) return;
  double s = 1.0;
  sign[j] = 1.0;
  int i = 1;
  for (; i + 8 <= N; i += 8) {
    double d[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) d[u] = dprev[(long)(i + u) * J + j];
#pragma unroll
    for (int u = 0; u < 8; ++u) { s = (s * d[u] < 0.0) ? -1.0 : 1.0; sign[(long)(i + u) * J + j] = s; }
  }
  for (; i < N; ++i) { s = (s * dprev[(long)i * J + j] < 0.0) ? -1.0 : 1.0; sign[(long)i * J + j] = s; }
}
// global transform of joint j by walking up the parent chain (no per-thread joint arrays)
__device__ void global_of(const double* lrot, const double* lpos, const double* sign, const int* parents, long base, int j,
                          DQ& rot, D3& pos) {
  DQ q = ldq(lrot + (base + j) * 4);
  const double s = sign[base + j];
  rot = DQ{s * q.w, s * q.x, s * q.y, s * q.z};
  pos = ld3(lpos + (base + j) * 3);
  for (int p = parents[j]; p >= 0; p = parents[p]) {
    DQ pq = ldq(lrot + (base + p) * 4);
    const double ps = sign[base + p];
    pq = DQ{ps * pq.w, ps * pq.x, ps * pq.y, ps * pq.z};
    pos = dq_mul_vec(pq, pos) + ld3(lpos + (base + p) * 3);
    rot = dq_mul(pq, rot);
  }
}
// 3. per frame: root position / facing rotation and the gaze target candidate root_pos + 100 * look
__global__ void anim_root_k(ZeggsAnimDims d, const int* parents, const double* lrot, const double* lpos_in,
                            const double* sign, double* root_pos, double* root_rot, double* gz /* [3][N] */) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= d.N) return;
  const long base = (long)n * d.J;
  DQ r; D3 p;
  global_of(lrot, lpos_in, sign, parents, base, d.spine2, r, p);
  const D3 rp = D3{p.x, 0.0, p.z};
  global_of(lrot, lpos_in, sign, parents, base, d.hips, r, p);
  const D3 fwd = D3{0.0, 0.0, 1.0};
  D3 f = dq_mul_vec(r, fwd);
  f.y = 0.0;
  f = (1.0 / sqrt(ddot(f, f))) * f;
  const DQ rr = dq_between_n(fwd, f);
  global_of(lrot, lpos_in, sign, parents, base, d.head, r, p);
  D3 look = dq_mul_vec(r, fwd);
  look.y = 0.0;
  look = (1.0 / sqrt(ddot(look, look))) * look;
  st3(root_pos + (long)n * 3, rp);
  stq(root_rot + (long)n * 4, rr);
  gz[n] = rp.x + 100.0 * look.x; gz[d.N + n] = rp.y + 100.0 * look.y; gz[2L * d.N + n] = rp.z + 100.0 * look.z;
}

// 4. exact median per coordinate by radix select.  sel state per (coord c, rank r): prefix, remaining rank.
__device__ __forceinline__ unsigned long long dkey(double v) {
  unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double dkey_inv(unsigned long long k) {
  const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)u);
}
struct SelState { unsigned long long prefix[6]; unsigned long long rank[6]; unsigned int hist[6][256]; };
__global__ void sel_init_k(SelState* st, int N) {
  const int t = threadIdx.x;
  for (int i = t; i < 6 * 256; i += blockDim.x) st->hist[i / 256][i % 256] = 0;
  if (t < 6) { st->prefix[t] = 0; st->rank[t] = (t & 1) ? (unsigned long long)(N / 2) : (unsigned long long)((N - 1) / 2); }
}
__global__ void sel_hist_k(SelState* st, const double* gz, int N, int pass) {   // pass 7 (top byte) .. 0
  __shared__ unsigned int h[6][256];
  for (int i = threadIdx.x; i < 6 * 256; i += blockDim.x) h[i / 256][i % 256] = 0;
  __syncthreads();
  const int shift = 8 * pass;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < 3L * N; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i / N);
    const unsigned long long k = dkey(gz[i]);
    const unsigned int b = (unsigned int)((k >> shift) & 255ull);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int s = 2 * c + r;
      const bool match = pass == 7 || ((k >> (shift + 8)) == (st->prefix[s] >> (shift + 8)));
      if (match) atomicAdd(&h[s][b], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 6 * 256; i += blockDim.x)
    if (h[i / 256][i % 256]) atomicAdd(&st->hist[i / 256][i % 256], h[i / 256][i % 256]);
}
__global__ void sel_pick_k(SelState* st, int pass) {   // one block, 6 threads do the work
  const int s = threadIdx.x;
  if (s < 6) {
    unsigned long long rank = st->rank[s], cum = 0;
    int b = 0;
    for (; b < 256; ++b) {
      const unsigned long long c = st->hist[s][b];
      if (rank < cum + c) break;
      cum += c;
    }
    st->rank[s] = rank - cum;
    st->prefix[s] |= ((unsigned long long)b) << (8 * pass);
    for (int i = 0; i < 256; ++i) st->hist[s][i] = 0;
  }
}
// 5. local frame of the root: joint 0 expressed relative to the root, gaze target / direction
__global__ void anim_local_k(ZeggsAnimDims d, const SelState* st, const double* lrot_raw, const double* lpos_in,
                             const double* sign, const double* root_pos, const double* root_rot, ZeggsAnimOut o) {
  const long n = (long)d.N * d.J;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % d.J);
    const long f = i / d.J;
    DQ q = ldq(lrot_raw + i * 4);
    const double s = sign[i];
    q = DQ{s * q.w, s * q.x, s * q.y, s * q.z};
    D3 p = ld3(lpos_in + i * 3);
    if (j == 0) {
      const DQ ir = dq_inv(ldq(root_rot + f * 4));
      const D3 rp = ld3(root_pos + f * 3);
      q = dq_mul(ir, q);
      p = dq_mul_vec(ir, p - rp);
      D3 g;   // np.median: mean of the two middle order statistics
      g.x = 0.5 * (dkey_inv(st->prefix[0]) + dkey_inv(st->prefix[1]));
      g.y = 0.5 * (dkey_inv(st->prefix[2]) + dkey_inv(st->prefix[3]));
      g.z = 0.5 * (dkey_inv(st->prefix[4]) + dkey_inv(st->prefix[5]));
      st3(o.gaze_pos + f * 3, g);
      st3(o.gaze_dir + f * 3, dq_mul_vec(ir, g - rp));
    }
    stq(o.lrot + i * 4, q);
    st3(o.lpos + i * 3, p);
    const D3 tx = dq_mul_vec(q, D3{1.0, 0.0, 0.0}), ty = dq_mul_vec(q, D3{0.0, 1.0, 0.0});
    float* t = o.ltxy + i * 6;
    t[0] = (float)tx.x; t[1] = (float)tx.y; t[2] = (float)tx.z; t[3] = (float)ty.x; t[4] = (float)ty.y; t[5] = (float)ty.z;
  }
}
// 6. velocities: backward differences for frames >= 1; frame 0 extrapolated, v0 = v1 - (v3 - v2)
__device__ __forceinline__ D3 pos_diff(const double* x, long stride, long f, double dt) {
  return (1.0 / dt) * (ld3(x + f * stride) - ld3(x + (f - 1) * stride));
}
__device__ __forceinline__ D3 rot_diff(const double* q, long stride, long f, double dt) {
  return (1.0 / dt) * dq_to_helical(dq_abs(dq_mul(ldq(q + f * stride), dq_inv(ldq(q + (f - 1) * stride)))));
}
__global__ void anim_vel_k(ZeggsAnimDims d, ZeggsAnimOut o) {
  const long n = (long)d.N * (d.J + 1);   // item J of a frame = the root trajectory
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % (d.J + 1));
    const long f = i / (d.J + 1);
    const bool root = j == d.J;
    const double* px = root ? o.root_pos : o.lpos + (long)j * 3;
    const double* qx = root ? o.root_rot : o.lrot + (long)j * 4;
    const long ps = root ? 3 : (long)d.J * 3, qs = root ? 4 : (long)d.J * 4;
    D3 v, w;
    if (f > 0) { v = pos_diff(px, ps, f, d.dt); w = rot_diff(qx, qs, f, d.dt); }
    else {
      v = pos_diff(px, ps, 1, d.dt) - (pos_diff(px, ps, 3, d.dt) - pos_diff(px, ps, 2, d.dt));
      w = rot_diff(qx, qs, 1, d.dt) - (rot_diff(qx, qs, 3, d.dt) - rot_diff(qx, qs, 2, d.dt));
    }
    if (root) {   // expressed in the root frame of the PREVIOUS frame (frame 0: its own)
      const DQ ir = dq_inv(ldq(o.root_rot + (f > 0 ? f - 1 : 0) * 4));
      st3(o.root_vel + f * 3, dq_mul_vec(ir, v));
      st3(o.root_vrt + f * 3, dq_mul_vec(ir, w));
    } else {
      st3(o.lvel + (f * d.J + j) * 3, v);
      st3(o.lvrt + (f * d.J + j) * 3, w);
    }
  }
}
// 7. character space: FK with velocities (quat.py fk_vel); one thread per frame walks the joints in order and
//    reads back the parent's values it wrote itself
__global__ void anim_char_k(ZeggsAnimDims d, const int* parents, ZeggsAnimOut o) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= d.N) return;
  const long b = (long)f * d.J;
  for (int j = 0; j < d.J; ++j) {
    const DQ lq = ldq(o.lrot + (b + j) * 4);
    const D3 lp = ld3(o.lpos + (b + j) * 3), lw = ld3(o.lvrt + (b + j) * 3), lv = ld3(o.lvel + (b + j) * 3);
    DQ gq; D3 gp, gw, gv;
    if (j == 0) { gq = lq; gp = lp; gw = lw; gv = lv; }
    else {
      const int p = parents[j];
      const DQ pq = ldq(o.crot + (b + p) * 4);
      const D3 pp = ld3(o.cpos + (b + p) * 3), pw = ld3(o.cvrt + (b + p) * 3), pv = ld3(o.cvel + (b + p) * 3);
      const D3 rp = dq_mul_vec(pq, lp);
      gp = rp + pp;
      gq = dq_mul(pq, lq);
      gw = pw + dq_mul_vec(pq, lw);
      gv = pv + dq_mul_vec(pq, lv) + dcross(pw, rp);
    }
    stq(o.crot + (b + j) * 4, gq); st3(o.cpos + (b + j) * 3, gp); st3(o.cvrt + (b + j) * 3, gw); st3(o.cvel + (b + j) * 3, gv);
    const D3 tx = dq_mul_vec(gq, D3{1.0, 0.0, 0.0}), ty = dq_mul_vec(gq, D3{0.0, 1.0, 0.0});
    float* t = o.ctxy + (b + j) * 6;
    t[0] = (float)tx.x; t[1] = (float)tx.y; t[2] = (float)tx.z; t[3] = (float)ty.x; t[4] = (float)ty.y; t[5] = (float)ty.z;
  }
}

// ------------------------------------------------------------------ decoder output -> BVH channels
__global__ void pose_to_bvh_k(ZeggsBvhDims d, const float* root_pos, const float* root_rot, const float* lpos,
                              const float* ltxy, double* positions, double* euler) {
  const long n = (long)d.T * d.J;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % d.J);
    const long f = i / d.J;
    const float* t = ltxy + i * 6;
    DQ q = dq_from_xy(D3{(double)t[0], (double)t[1], (double)t[2]}, D3{(double)t[3], (double)t[4], (double)t[5]});
    D3 p = D3{(double)lpos[i * 3], (double)lpos[i * 3 + 1], (double)lpos[i * 3 + 2]};
    if (j == 0) {   // fold the (optionally re-based) root trajectory into joint 0
      DQ rr = DQ{(double)root_rot[f * 4], (double)root_rot[f * 4 + 1], (double)root_rot[f * 4 + 2], (double)root_rot[f * 4 + 3]};
      D3 rp = D3{(double)root_pos[f * 3], (double)root_pos[f * 3 + 1], (double)root_pos[f * 3 + 2]};
      if (d.rebase) {
        const DQ r0 = dq_inv(DQ{(double)root_rot[0], (double)root_rot[1], (double)root_rot[2], (double)root_rot[3]});
        const D3 p0 = D3{(double)root_pos[0], (double)root_pos[1], (double)root_pos[2]};
        const DQ sr = DQ{d.start_rot[0], d.start_rot[1], d.start_rot[2], d.start_rot[3]};
        rp = dq_mul_vec(sr, dq_mul_vec(r0, rp - p0)) + D3{d.start_pos[0], d.start_pos[1], d.start_pos[2]};
        rr = dq_mul(sr, dq_mul(r0, rr));
      }
      p = dq_mul_vec(rr, p) + rp;
      q = dq_mul(rr, q);
    }
    st3(positions + i * 3, p);
    dq_to_euler_deg(q, euler + i * 3, order_code(d.order));
  }
}

// The same conversion written straight into the BVH motion block's row layout: table[f] = [root position 3 | euler angles of
// joint seq[0], seq[1], ... (hierarchy order of the file)], for a CHUNK of frames of a longer clip: the re-basing frame (frame 0
// of the whole clip) is passed explicitly.  Lets generate_gesture() convert / download / format a chunk while the next one is
// still being decoded.
__global__ void pose_to_bvh_table_k(ZeggsBvhDims d, const float* root_pos, const float* root_rot, const float* lpos,
                                    const float* ltxy, const float* ref_pos, const float* ref_rot, const int* seq,
                                    double* table) {
  const long n = (long)d.T * d.J;
  const int cols = 3 + 3 * d.J;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % d.J);            // position in the file's joint order
    const long f = i / d.J;
    const int j = seq[k];
    const long src = f * d.J + j;
    const float* t = ltxy + src * 6;
    DQ q = dq_from_xy(D3{(double)t[0], (double)t[1], (double)t[2]}, D3{(double)t[3], (double)t[4], (double)t[5]});
    if (j == 0) {
      D3 p = D3{(double)lpos[src * 3], (double)lpos[src * 3 + 1], (double)lpos[src * 3 + 2]};
      DQ rr = DQ{(double)root_rot[f * 4], (double)root_rot[f * 4 + 1], (double)root_rot[f * 4 + 2], (double)root_rot[f * 4 + 3]};
      D3 rp = D3{(double)root_pos[f * 3], (double)root_pos[f * 3 + 1], (double)root_pos[f * 3 + 2]};
      if (d.rebase) {
        const DQ r0 = dq_inv(DQ{(double)ref_rot[0], (double)ref_rot[1], (double)ref_rot[2], (double)ref_rot[3]});
        const D3 p0 = D3{(double)ref_pos[0], (double)ref_pos[1], (double)ref_pos[2]};
        const DQ sr = DQ{d.start_rot[0], d.start_rot[1], d.start_rot[2], d.start_rot[3]};
        rp = dq_mul_vec(sr, dq_mul_vec(r0, rp - p0)) + D3{d.start_pos[0], d.start_pos[1], d.start_pos[2]};
        rr = dq_mul(sr, dq_mul(r0, rr));
      }
      p = dq_mul_vec(rr, p) + rp;
      q = dq_mul(rr, q);
      st3(table + f * cols, p);
    }
    dq_to_euler_deg(q, table + f * cols + 3 + 3 * k, order_code(d.order));
  }
}

struct AnimWs { double *lrot_raw, *dprev, *sign, *gz; SelState* sel; };
AnimWs carve_anim(const ZeggsAnimDims& d, Arena& a) {
  AnimWs w;
  const size_t NJ = (size_t)d.N * d.J;
  w.lrot_raw = (double*)a.raw(NJ * 4 * sizeof(double));
  w.dprev = (double*)a.raw(NJ * sizeof(double));
  w.sign = (double*)a.raw(NJ * sizeof(double));
  w.gz = (double*)a.raw((size_t)3 * d.N * sizeof(double));
  w.sel = (SelState*)a.raw(sizeof(SelState));
  return w;
}
inline dim3 grid_for(long n, int block) { long g = (n + block - 1) / block; return dim3((unsigned)(g > 8192 ? 8192 : (g < 1 ? 1 : g))); }

}  // namespace

extern "C" size_t zeggs_anim_features_workspace_bytes(const ZeggsAnimDims* d) {
  Arena a(nullptr, 0);
  carve_anim(*d, a);
  return a.off + 256;
}

extern "C" int zeggs_anim_features(const ZeggsAnimDims* dp, const int* parents, const double* euler_deg,
                                   const double* positions, const ZeggsAnimOut* out, void* ws, size_t ws_bytes,
                                   void* stream) {
  const ZeggsAnimDims& d = *dp;
  hipStream_t s = (hipStream_t)stream;
  ZCHECK(d.N >= 4, "anim_features: need at least 4 frames (frame 0 velocities are extrapolated from frames 1..3), got %d", d.N);
  ZCHECK(d.J >= 1 && d.hips >= 0 && d.hips < d.J && d.spine2 >= 0 && d.spine2 < d.J && d.head >= 0 && d.head < d.J,
         "anim_features: joint indices out of range");
  ZCHECK(d.dt > 0.0, "anim_features: dt must be positive");
  Arena a(ws, ws_bytes);
  AnimWs w = carve_anim(d, a);
  ZCHECK(a.ok(), "anim_features: workspace too small (%zu < %zu)", ws_bytes, a.off);
  const long NJ = (long)d.N * d.J;
  hipLaunchKernelGGL(anim_quat_k, grid_for(NJ, 256), dim3(256), 0, s, euler_deg, w.lrot_raw, w.dprev, d.N, d.J, order_code(d.order));
  hipLaunchKernelGGL(anim_unroll_k, dim3((d.J + 63) / 64), dim3(64), 0, s, w.dprev, w.sign, d.N, d.J);
  hipLaunchKernelGGL(anim_root_k, dim3((d.N + 63) / 64), dim3(64), 0, s, d, parents, w.lrot_raw, positions, w.sign,
                     out->root_pos, out->root_rot, w.gz);
  hipLaunchKernelGGL(sel_init_k, dim3(1), dim3(256), 0, s, w.sel, d.N);
  for (int pass = 7; pass >= 0; --pass) {
    hipLaunchKernelGGL(sel_hist_k, grid_for(3L * d.N, 256), dim3(256), 0, s, w.sel, w.gz, d.N, pass);
    hipLaunchKernelGGL(sel_pick_k, dim3(1), dim3(64), 0, s, w.sel, pass);
  }
  hipLaunchKernelGGL(anim_local_k, grid_for(NJ, 256), dim3(256), 0, s, d, w.sel, w.lrot_raw, positions, w.sign,
                     out->root_pos, out->root_rot, *out);
  hipLaunchKernelGGL(anim_vel_k, grid_for((long)d.N * (d.J + 1), 256), dim3(256), 0, s, d, *out);
  hipLaunchKernelGGL(anim_char_k, dim3((d.N + 63) / 64), dim3(64), 0, s, d, parents, *out);
  ZLAUNCH_CHECK("anim_features");
  return 0;
}

extern "C" int zeggs_pose_to_bvh(const ZeggsBvhDims* dp, const float* root_pos, const float* root_rot, const float* lpos,
                                 const float* ltxy, double* positions, double* euler_deg, void* stream) {
  const ZeggsBvhDims& d = *dp;
  ZCHECK(d.T >= 1 && d.J >= 1, "pose_to_bvh: empty clip");
  hipLaunchKernelGGL(pose_to_bvh_k, grid_for((long)d.T * d.J, 256), dim3(256), 0, (hipStream_t)stream, d, root_pos,
                     root_rot, lpos, ltxy, positions, euler_deg);
  ZLAUNCH_CHECK("pose_to_bvh");
  return 0;
}

extern "C" int zeggs_pose_to_bvh_table(const ZeggsBvhDims* dp, const float* root_pos, const float* root_rot, const float* lpos,
                                       const float* ltxy, const float* ref_root_pos, const float* ref_root_rot, const int* seq,
                                       double* table, void* stream) {
  const ZeggsBvhDims& d = *dp;
  ZCHECK(d.T >= 1 && d.J >= 1, "pose_to_bvh_table: empty clip");
  ZCHECK(seq != nullptr && table != nullptr, "pose_to_bvh_table: seq / table missing");
  hipLaunchKernelGGL(pose_to_bvh_table_k, grid_for((long)d.T * d.J, 256), dim3(256), 0, (hipStream_t)stream, d, root_pos,
                     root_rot, lpos, ltxy, ref_root_pos ? ref_root_pos : root_pos, ref_root_rot ? ref_root_rot : root_rot, seq,
                     table);
  ZLAUNCH_CHECK("pose_to_bvh_table");
  return 0;
}
