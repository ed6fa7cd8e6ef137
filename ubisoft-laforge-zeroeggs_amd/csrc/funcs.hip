// The free functions of the reference's Networks layer (ZEGGS/modules.py:673-813) as stand-alone entry points, forward and
// backward: normalize, vectorize_input, devectorize_output, compute_KL_div, get_mask_from_lengths.  Inside the decoder / loss
// kernels the same arithmetic is fused into the step and feature passes (decoder.hip: dec_devec_k, loss.hip); these entry points
// exist so that the reference's OWN train.py / generate.py loop runs against `zeggs.modules` (INTEGRATION.md route 2), whose
// inline loss calls `normalize` and `compute_KL_div` on device tensors and differentiates through them.
// All of them are per-row streaming kernels (HBM / latency bound, a few KB per row).
#include "../../include/zeggs_hip.h"
#include "common.h"
#include "dec_math.h"

namespace {

// ------------------------------------------------------------------ normalize (modules.py:673-675)
// y = x / (||x||_2 + eps) over the last dimension.  One wave per row (the reference's rows are 3-vectors; any width works).
__global__ __launch_bounds__(256) void normalize_vec_fwd_k(const float* x, float* y, long rows, int width, float eps) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int l = threadIdx.x & 63;
  const float* xr = x + r * width;
  float s = 0.f;
  for (int c = l; c < width; c += 64) s += xr[c] * xr[c];
  s = wave_sum(s);
  const float inv = 1.f / (sqrtf(s) + eps);
  for (int c = l; c < width; c += 64) y[r * width + c] = xr[c] * inv;
}
// dx = dy / (n + eps) - x (x . dy) / (n (n + eps)^2)      (n = ||x||; n = 0: the second term is 0 * inf in torch -> we drop it)
__global__ __launch_bounds__(256) void normalize_vec_bwd_k(const float* x, const float* dy, float* dx, long rows, int width,
                                                           float eps) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int l = threadIdx.x & 63;
  const float* xr = x + r * width;
  const float* gr = dy + r * width;
  float s = 0.f, d = 0.f;
  for (int c = l; c < width; c += 64) { s += xr[c] * xr[c]; d += xr[c] * gr[c]; }
  s = wave_sum(s);
  d = wave_sum(d);
  const float n = sqrtf(s), ne = n + eps;
  const float k = n > 0.f ? d / (n * ne * ne) : 0.f;
  for (int c = l; c < width; c += 64) dx[r * width + c] = gr[c] / ne - k * xr[c];
}

// ------------------------------------------------------------------ vectorize_input (modules.py:677-713)
// out[b] = (cat[root_vel 3, root_vrt 3, lpos 3J, ltxy 6J, lvel 3J, lvrt 3J, gaze_dir 3] - mean) / std,
// gaze_dir = quat_inv_mul_vec(root_rot, gaze_pos - root_pos)  (NOT normalised: :693)
struct VecIn {
  const float *root_pos, *root_rot, *root_vel, *root_vrt, *lpos, *ltxy, *lvel, *lvrt, *gaze_pos;
};
struct VecGrad {
  float *root_pos, *root_rot, *root_vel, *root_vrt, *lpos, *ltxy, *lvel, *lvrt, *gaze_pos;
};
__device__ __forceinline__ const float* vec_src(const VecIn& a, int b, int J, int c, int& off) {
  // column c of the flattened pose encoding -> (array, offset inside row b)
  if (c < 3) { off = b * 3 + c; return a.root_vel; }
  if (c < 6) { off = b * 3 + c - 3; return a.root_vrt; }
  c -= 6;
  if (c < 3 * J) { off = b * 3 * J + c; return a.lpos; }
  c -= 3 * J;
  if (c < 6 * J) { off = b * 6 * J + c; return a.ltxy; }
  c -= 6 * J;
  if (c < 3 * J) { off = b * 3 * J + c; return a.lvel; }
  c -= 3 * J;
  off = b * 3 * J + c;
  return a.lvrt;
}
__global__ void vectorize_fwd_k(VecIn a, int B, int J, const float* mean, const float* stdv, float* out) {
  const int b = blockIdx.x, PO = 6 + 15 * J, PI = PO + 3;
  for (int c = threadIdx.x; c < PO; c += blockDim.x) {
    int off;
    const float* src = vec_src(a, b, J, c, off);
    out[(long)b * PI + c] = (src[off] - mean[c]) / stdv[c];
  }
  if (threadIdx.x == 0) {
    const float* rq = a.root_rot + b * 4;
    Q4 q = Q4{rq[0], rq[1], rq[2], rq[3]};
    V3 v = v3(a.gaze_pos[b * 3] - a.root_pos[b * 3], a.gaze_pos[b * 3 + 1] - a.root_pos[b * 3 + 1],
              a.gaze_pos[b * 3 + 2] - a.root_pos[b * 3 + 2]);
    V3 gd = quat_mul_vec(quat_inv(q), v);
    const float g[3] = {gd.x, gd.y, gd.z};
    for (int k = 0; k < 3; ++k) out[(long)b * PI + PO + k] = (g[k] - mean[PO + k]) / stdv[PO + k];
  }
}
__global__ void vectorize_bwd_k(VecIn a, VecGrad g, int B, int J, const float* stdv, const float* dout) {
  const int b = blockIdx.x, PO = 6 + 15 * J, PI = PO + 3;
  VecIn ga{g.root_pos, g.root_rot, g.root_vel, g.root_vrt, g.lpos, g.ltxy, g.lvel, g.lvrt, g.gaze_pos};
  for (int c = threadIdx.x; c < PO; c += blockDim.x) {
    int off;
    float* dst = const_cast<float*>(vec_src(ga, b, J, c, off));
    dst[off] = dout[(long)b * PI + c] / stdv[c];
  }
  if (threadIdx.x == 0) {
    const float* rq = a.root_rot + b * 4;
    Q4 q = Q4{rq[0], rq[1], rq[2], rq[3]};
    V3 v = v3(a.gaze_pos[b * 3] - a.root_pos[b * 3], a.gaze_pos[b * 3 + 1] - a.root_pos[b * 3 + 1],
              a.gaze_pos[b * 3 + 2] - a.root_pos[b * 3 + 2]);
    V3 dgd = v3(dout[(long)b * PI + PO] / stdv[PO], dout[(long)b * PI + PO + 1] / stdv[PO + 1],
                dout[(long)b * PI + PO + 2] / stdv[PO + 2]);
    Q4 dqi; V3 dv;
    qmv_bwd(quat_inv(q), v, dgd, dqi, dv);
    float* dq = g.root_rot + b * 4;          // d/dq of quat_inv(q): the vector part changes sign
    dq[0] = dqi.w; dq[1] = -dqi.x; dq[2] = -dqi.y; dq[3] = -dqi.z;
    float* dg = g.gaze_pos + b * 3; dg[0] = dv.x; dg[1] = dv.y; dg[2] = dv.z;
    float* dp = g.root_pos + b * 3; dp[0] = -dv.x; dp[1] = -dv.y; dp[2] = -dv.z;
  }
}

// ------------------------------------------------------------------ devectorize_output (modules.py:716-742)
// pose = predicted * std + mean (the six slices are views of it); root_pos' = quat_mul_vec(root_rot, vel dt) + root_pos;
// root_rot' = quat_mul(quat_from_helical(quat_mul_vec(root_rot, vrt dt)), root_rot)
__global__ void devectorize_fwd_k(int B, int PO, float dt, const float* pred, const float* root_pos, const float* root_rot,
                                  const float* mean, const float* stdv, float* pose, float* nrpos, float* nrrot) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < PO; c += blockDim.x) pose[(long)b * PO + c] = pred[(long)b * PO + c] * stdv[c] + mean[c];
  if (threadIdx.x == 0) {
    float p[6];
    for (int c = 0; c < 6; ++c) p[c] = pred[(long)b * PO + c] * stdv[c] + mean[c];
    Q4 q = Q4{root_rot[b * 4], root_rot[b * 4 + 1], root_rot[b * 4 + 2], root_rot[b * 4 + 3]};
    V3 pos = v3(root_pos[b * 3], root_pos[b * 3 + 1], root_pos[b * 3 + 2]);
    V3 np = quat_mul_vec(q, dt * v3(p[0], p[1], p[2])) + pos;
    V3 u = quat_mul_vec(q, dt * v3(p[3], p[4], p[5]));
    Q4 nq = quat_exp_mul(0.5f * u, q);
    nrpos[b * 3] = np.x; nrpos[b * 3 + 1] = np.y; nrpos[b * 3 + 2] = np.z;
    nrrot[b * 4] = nq.w; nrrot[b * 4 + 1] = nq.x; nrrot[b * 4 + 2] = nq.y; nrrot[b * 4 + 3] = nq.z;
  }
}
// upstream: dpose [B,PO] (may be NULL), dnrpos [B,3] (may be NULL), dnrrot [B,4] (may be NULL)
__global__ void devectorize_bwd_k(int B, int PO, float dt, const float* pred, const float* root_pos, const float* root_rot,
                                  const float* mean, const float* stdv, const float* dpose, const float* dnrpos,
                                  const float* dnrrot, float* dpred, float* drpos, float* drrot) {
  const int b = blockIdx.x;
  for (int c = 6 + threadIdx.x; c < PO; c += blockDim.x)
    dpred[(long)b * PO + c] = dpose ? dpose[(long)b * PO + c] * stdv[c] : 0.f;
  if (threadIdx.x == 0) {
    float p[6], g6[6];
    for (int c = 0; c < 6; ++c) {
      p[c] = pred[(long)b * PO + c] * stdv[c] + mean[c];
      g6[c] = dpose ? dpose[(long)b * PO + c] : 0.f;
    }
    Q4 q = Q4{root_rot[b * 4], root_rot[b * 4 + 1], root_rot[b * 4 + 2], root_rot[b * 4 + 3]};
    V3 g_rp = dnrpos ? v3(dnrpos[b * 3], dnrpos[b * 3 + 1], dnrpos[b * 3 + 2]) : v3(0.f, 0.f, 0.f);
    Q4 g_rr = dnrrot ? Q4{dnrrot[b * 4], dnrrot[b * 4 + 1], dnrrot[b * 4 + 2], dnrrot[b * 4 + 3]} : Q4{0.f, 0.f, 0.f, 0.f};
    V3 vel = v3(p[0], p[1], p[2]), vrt = v3(p[3], p[4], p[5]);
    Q4 dq1; V3 dv1;
    qmv_bwd(q, dt * vel, g_rp, dq1, dv1);
    V3 u = quat_mul_vec(q, dt * vrt);
    Q4 E = quat_exp(0.5f * u);
    Q4 dE, dqy;
    qmul_bwd(E, q, g_rr, dE, dqy);
    V3 du = 0.5f * qexp_bwd(0.5f * u, dE);
    Q4 dq2; V3 dv2;
    qmv_bwd(q, dt * vrt, du, dq2, dv2);
    g6[0] += dt * dv1.x; g6[1] += dt * dv1.y; g6[2] += dt * dv1.z;
    g6[3] += dt * dv2.x; g6[4] += dt * dv2.y; g6[5] += dt * dv2.z;
    for (int c = 0; c < 6; ++c) dpred[(long)b * PO + c] = g6[c] * stdv[c];
    drpos[b * 3] = g_rp.x; drpos[b * 3 + 1] = g_rp.y; drpos[b * 3 + 2] = g_rp.z;
    drrot[b * 4] = dq1.w + dqy.w + dq2.w; drrot[b * 4 + 1] = dq1.x + dqy.x + dq2.x;
    drrot[b * 4 + 2] = dq1.y + dqy.y + dq2.y; drrot[b * 4 + 3] = dq1.z + dqy.z + dq2.z;
  }
}

// ------------------------------------------------------------------ compute_KL_div (modules.py:764-789)
// kl = mean_b( -0.5 mean_s(1 + logvar - mu^2 - exp(logvar)) ): all rows have S entries, so one mean over B S.  One block.
__global__ __launch_bounds__(256) void kl_fwd_k(const float* mu, const float* logvar, int n, float* out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float m = mu[i], lv = logvar[i];
    s += 1.f + lv - m * m - expf(lv);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = -0.5f * s / (float)n;
}
__global__ void kl_bwd_k(const float* mu, const float* logvar, int n, const float* dout, float* dmu, float* dlogvar) {
  const float g = dout[0] / (float)n;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    dmu[i] = g * mu[i];
    dlogvar[i] = -0.5f * g * (1.f - expf(logvar[i]));
  }
}

// ------------------------------------------------------------------ get_mask_from_lengths (modules.py:802-813)
__global__ void mask_from_lengths_k(const int64_t* lengths, int B, int max_len, uint8_t* mask) {
  const long n = (long)B * max_len;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    mask[i] = (int64_t)(i % max_len) < lengths[i / max_len] ? 1 : 0;
}

}  // namespace

extern "C" int zeggs_normalize_vec_fwd(const float* x, float* y, long rows, int width, float eps, void* stream) {
  ZCHECK(rows >= 0 && width >= 1, "normalize_vec: bad dims");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(normalize_vec_fwd_k, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, y, rows, width, eps);
  ZLAUNCH_CHECK("normalize_vec_fwd");
  return 0;
}
extern "C" int zeggs_normalize_vec_bwd(const float* x, const float* dy, float* dx, long rows, int width, float eps,
                                       void* stream) {
  ZCHECK(rows >= 0 && width >= 1, "normalize_vec: bad dims");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(normalize_vec_bwd_k, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, rows, width, eps);
  ZLAUNCH_CHECK("normalize_vec_bwd");
  return 0;
}

extern "C" int zeggs_vectorize_input_fwd(int B, int J, const float* root_pos, const float* root_rot, const float* root_vel,
                                         const float* root_vrt, const float* lpos, const float* ltxy, const float* lvel,
                                         const float* lvrt, const float* gaze_pos, const float* in_mean, const float* in_std,
                                         float* out, void* stream) {
  ZCHECK(B >= 1 && J >= 1, "vectorize_input: bad dims");
  VecIn a{root_pos, root_rot, root_vel, root_vrt, lpos, ltxy, lvel, lvrt, gaze_pos};
  hipLaunchKernelGGL(vectorize_fwd_k, dim3(B), dim3(256), 0, (hipStream_t)stream, a, B, J, in_mean, in_std, out);
  ZLAUNCH_CHECK("vectorize_input_fwd");
  return 0;
}
extern "C" int zeggs_vectorize_input_bwd(int B, int J, const float* root_pos, const float* root_rot, const float* gaze_pos,
                                         const float* in_std, const float* dout, float* d_root_pos, float* d_root_rot,
                                         float* d_root_vel, float* d_root_vrt, float* d_lpos, float* d_ltxy, float* d_lvel,
                                         float* d_lvrt, float* d_gaze_pos, void* stream) {
  ZCHECK(B >= 1 && J >= 1, "vectorize_input: bad dims");
  VecIn a{root_pos, root_rot, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, gaze_pos};
  VecGrad g{d_root_pos, d_root_rot, d_root_vel, d_root_vrt, d_lpos, d_ltxy, d_lvel, d_lvrt, d_gaze_pos};
  hipLaunchKernelGGL(vectorize_bwd_k, dim3(B), dim3(256), 0, (hipStream_t)stream, a, g, B, J, in_std, dout);
  ZLAUNCH_CHECK("vectorize_input_bwd");
  return 0;
}

extern "C" int zeggs_devectorize_output_fwd(int B, int J, float dt, const float* predicted, const float* root_pos,
                                            const float* root_rot, const float* out_mean, const float* out_std, float* pose,
                                            float* new_root_pos, float* new_root_rot, void* stream) {
  ZCHECK(B >= 1 && J >= 1, "devectorize_output: bad dims");
  hipLaunchKernelGGL(devectorize_fwd_k, dim3(B), dim3(256), 0, (hipStream_t)stream, B, 6 + 15 * J, dt, predicted, root_pos,
                     root_rot, out_mean, out_std, pose, new_root_pos, new_root_rot);
  ZLAUNCH_CHECK("devectorize_output_fwd");
  return 0;
}
extern "C" int zeggs_devectorize_output_bwd(int B, int J, float dt, const float* predicted, const float* root_pos,
                                            const float* root_rot, const float* out_mean, const float* out_std,
                                            const float* d_pose, const float* d_new_root_pos, const float* d_new_root_rot,
                                            float* d_predicted, float* d_root_pos, float* d_root_rot, void* stream) {
  ZCHECK(B >= 1 && J >= 1, "devectorize_output: bad dims");
  hipLaunchKernelGGL(devectorize_bwd_k, dim3(B), dim3(256), 0, (hipStream_t)stream, B, 6 + 15 * J, dt, predicted, root_pos,
                     root_rot, out_mean, out_std, d_pose, d_new_root_pos, d_new_root_rot, d_predicted, d_root_pos, d_root_rot);
  ZLAUNCH_CHECK("devectorize_output_bwd");
  return 0;
}

extern "C" int zeggs_kl_div_fwd(const float* mu, const float* logvar, int B, int S, float* out, void* stream) {
  ZCHECK(B >= 1 && S >= 1, "kl_div: bad dims");
  hipLaunchKernelGGL(kl_fwd_k, dim3(1), dim3(256), 0, (hipStream_t)stream, mu, logvar, B * S, out);
  ZLAUNCH_CHECK("kl_div_fwd");
  return 0;
}
extern "C" int zeggs_kl_div_bwd(const float* mu, const float* logvar, int B, int S, const float* dout, float* dmu,
                                float* dlogvar, void* stream) {
  ZCHECK(B >= 1 && S >= 1, "kl_div: bad dims");
  hipLaunchKernelGGL(kl_bwd_k, dim3(cdiv((long)B * S, 256)), dim3(256), 0, (hipStream_t)stream, mu, logvar, B * S, dout, dmu,
                     dlogvar);
  ZLAUNCH_CHECK("kl_div_bwd");
  return 0;
}

extern "C" int zeggs_mask_from_lengths(const int64_t* lengths, int B, int max_len, uint8_t* mask, void* stream) {
  ZCHECK(B >= 1 && max_len >= 0, "mask_from_lengths: bad dims");
  if (max_len == 0) return 0;
  hipLaunchKernelGGL(mask_from_lengths_k, dim3(cdiv((long)B * max_len, 256)), dim3(256), 0, (hipStream_t)stream, lengths, B,
                     max_len, mask);
  ZLAUNCH_CHECK("mask_from_lengths");
  return 0;
}
