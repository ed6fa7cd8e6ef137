// Quaternion helpers of the decoder step (forward + analytic backward), shared by decoder.hip / decoder_fast.hip.
#pragma once
#include "common.h"

// ------------------------------------------------------------------ device helpers
// sin(h) / h and cos(h) of a half rotation angle.  The root turns a fraction of a radian per frame, so h < 1 is the only case
// that occurs in practice: two even polynomials in h^2 (truncation error 2.5e-8 / 2e-9 at h = 1, below fp32 rounding) instead of
// sinf + cosf + a division (~200 instructions with their range reduction) in the epilogue of every output stage of the
// persistent rollouts.  h >= 1: the library functions.
// -DZEGGS_EXACT_SINCOS=1: always the library functions (A/B builds of tools/drift_ab.py).
#ifndef ZEGGS_EXACT_SINCOS
#define ZEGGS_EXACT_SINCOS 0
#endif
static __device__ __forceinline__ void d_sinc_cos(float h, float& sinc, float& c) {
  if (!ZEGGS_EXACT_SINCOS && h < 1.f) {
    const float u = h * h;
    sinc = 1.f + u * (-1.f / 6.f + u * (1.f / 120.f + u * (-1.f / 5040.f + u * (1.f / 362880.f + u * (-1.f / 39916800.f)))));
    c = 1.f + u * (-0.5f + u * (1.f / 24.f + u * (-1.f / 720.f + u * (1.f / 40320.f + u * (-1.f / 3628800.f + u * (1.f / 479001600.f))))));
  } else {
    sinc = sinf(h) / h;
    c = cosf(h);
  }
}
// reference anim/tquat.py:94-107: quat_from_helical(x) = quat_exp(x/2)
static __device__ __forceinline__ Q4 quat_exp(V3 x) {
  float h = sqrtf(x.x * x.x + x.y * x.y + x.z * x.z);
  if (h < 1e-5f) {
    float n = sqrtf(1.f + h * h) + 1e-5f;
    return Q4{1.f / n, x.x / n, x.y / n, x.z / n};
  }
  float s, c;
  d_sinc_cos(h, s, c);
  return Q4{c, x.x * s, x.y * s, x.z * s};
}

// The root update  q' = quat_mul(quat_from_helical(u), q) = quat_mul(quat_exp(u / 2), q)  (ZEGGS/modules.py:739) as q + DELTA.
// Written as a product, the scalar part of exp(x) -- cos|x| = 1 - |x|^2 / 2 + ..., |x| ~ 1e-3 per frame -- is rounded to the
// fp32 grid around 1 (spacing 6e-8) BEFORE it multiplies q: an error of up to 3e-8 per frame that keeps its sign while the
// root turns at a steady rate, i.e. the norm of the (never re-normalised) root quaternion drifts linearly, 1e-3 over the
// 108 000 frames of a 30-minute decode, and every later root step is scaled by it -- the whole long-run drift of round 3
// (tools/drift_ab.py: with the root update in float64 the deviation from the reference's fp64 run falls from 5e-2 to 2e-4;
// exact gates / exact sin, cos / no algebraic folds change nothing).  The reference's own fp32 run has the same defect with
// other rounding luck (3e-3).  Here cos|x| - 1 = -|x|^2 (1/2 - |x|^2 / 24 + ...) is formed at full relative precision and the
// small correction DELTA = (cos|x| - 1) q + [ -e.qv, q.w e + e x qv ] is added to q with ONE (data-dependent, zero-mean)
// rounding.  Same function as quat_mul(quat_exp(x), q) in exact arithmetic, both branches of quat_exp (tquat.py:94-99).
static __device__ __forceinline__ Q4 quat_exp_mul(V3 x, Q4 q) {
  const float u = x.x * x.x + x.y * x.y + x.z * x.z;
  const float h = sqrtf(u);
  float cm1, s;          // scalar part of exp(x) minus one; factor of the vector part
  if (h < 1e-5f) {       // quat_normalize([1, x], eps = 1e-5) = [1, x] / (sqrt(1 + |x|^2) + 1e-5)
    const float nm1 = 0.5f * u;                    // sqrt(1 + u) - 1 for u < 1e-10
    const float ne = 1.f + (nm1 + 1e-5f);
    s = 1.f / ne;
    cm1 = -(nm1 + 1e-5f) * s;
  } else if (!ZEGGS_EXACT_SINCOS && h < 1.f) {
    s = 1.f + u * (-1.f / 6.f + u * (1.f / 120.f + u * (-1.f / 5040.f + u * (1.f / 362880.f + u * (-1.f / 39916800.f)))));
    cm1 = u * (-0.5f + u * (1.f / 24.f + u * (-1.f / 720.f + u * (1.f / 40320.f + u * (-1.f / 3628800.f + u * (1.f / 479001600.f))))));
  } else {
    const float sh = sinf(0.5f * h);
    s = sinf(h) / h;
    cm1 = -2.f * sh * sh;
  }
  const V3 e = s * x, qv = v3(q.x, q.y, q.z);
  const V3 c = cross(e, qv);
  return Q4{q.w + (cm1 * q.w - dot(e, qv)), q.x + (cm1 * q.x + q.w * e.x + c.x), q.y + (cm1 * q.y + q.w * e.y + c.y),
            q.z + (cm1 * q.z + q.w * e.z + c.z)};
}

// quat_exp and its backward for the SAME argument share the norm and its sine / cosine (the root-integration backward
// evaluates both on one thread: half the transcendental work of calling quat_exp + qexp_bwd)
struct QExpCtx { float h, sh, ch; };
static __device__ __forceinline__ Q4 quat_exp_ctx(V3 x, QExpCtx& c) {
  c.h = sqrtf(x.x * x.x + x.y * x.y + x.z * x.z);
  if (c.h < 1e-5f) {
    c.sh = 0.f; c.ch = 1.f;
    float n = sqrtf(1.f + c.h * c.h) + 1e-5f;
    return Q4{1.f / n, x.x / n, x.y / n, x.z / n};
  }
  float s;
  d_sinc_cos(c.h, s, c.ch);
  c.sh = s * c.h;
  return Q4{c.ch, x.x * s, x.y * s, x.z * s};
}
static __device__ __forceinline__ V3 qexp_bwd_ctx(V3 x, Q4 g, const QExpCtx& c) {
  V3 gv = v3(g.x, g.y, g.z);
  if (c.h < 1e-5f) {
    float n = sqrtf(1.f + c.h * c.h), ne = n + 1e-5f;
    float ug = g.w + dot(gv, x);
    float k = ug / (n * ne * ne);
    return (1.f / ne) * gv - k * x;
  }
  const float s = c.sh / c.h, ds = (c.h * c.ch - c.sh) / (c.h * c.h);
  const float cc = -g.w * s + dot(gv, x) * ds / c.h;
  return s * gv + cc * x;
}

// backward of out = quat_mul_vec(q, v) given upstream g
static __device__ __forceinline__ void qmv_bwd(Q4 q, V3 v, V3 g, Q4& dq, V3& dv) {
  V3 qv = v3(q.x, q.y, q.z);
  V3 t = 2.0f * cross(qv, v);
  float dw = dot(g, t);
  V3 dt = q.w * g + cross(g, qv);
  V3 dqv = cross(t, g) + 2.0f * cross(v, dt);
  dv = g + 2.0f * cross(dt, qv);
  dq = Q4{dw, dqv.x, dqv.y, dqv.z};
}
// backward of out = quat_mul(x, y)
static __device__ __forceinline__ void qmul_bwd(Q4 x, Q4 y, Q4 g, Q4& dx, Q4& dy) {
  dx.w = g.w * y.w + g.x * y.x + g.y * y.y + g.z * y.z;
  dx.x = -g.w * y.x + g.x * y.w - g.y * y.z + g.z * y.y;
  dx.y = -g.w * y.y + g.x * y.z + g.y * y.w - g.z * y.x;
  dx.z = -g.w * y.z - g.x * y.y + g.y * y.x + g.z * y.w;
  dy.w = g.w * x.w + g.x * x.x + g.y * x.y + g.z * x.z;
  dy.x = -g.w * x.x + g.x * x.w + g.y * x.z - g.z * x.y;
  dy.y = -g.w * x.y - g.x * x.z + g.y * x.w + g.z * x.x;
  dy.z = -g.w * x.z + g.x * x.y - g.y * x.x + g.z * x.w;
}
// backward of out = quat_exp(x)
static __device__ __forceinline__ V3 qexp_bwd(V3 x, Q4 g) {
  float h = sqrtf(x.x * x.x + x.y * x.y + x.z * x.z);
  V3 gv = v3(g.x, g.y, g.z);
  if (h < 1e-5f) {
    float n = sqrtf(1.f + h * h), ne = n + 1e-5f;
    float ug = g.w + dot(gv, x);
    float k = ug / (n * ne * ne);
    return (1.f / ne) * gv - k * x;
  }
  float sh = sinf(h), ch = cosf(h);
  float s = sh / h, ds = (h * ch - sh) / (h * h);
  float c = -g.w * s + dot(gv, x) * ds / h;
  return s * gv + c * x;
}

