// Error state, fused RAdam, VAE re-parameterisation, dataset gathers.
#include <stdarg.h>

#include "../../include/zeggs_hip.h"
#include "common.h"
#include "kernels.h"

static thread_local char g_err[512] = "";

void zeggs_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* zeggs_last_error() { return g_err; }
extern "C" int zeggs_version() { return 101; }

namespace {

// RAdam (reference optimizers.py:59-97): v = b2 v + (1-b2) g^2 ; m = b1 m + (1-b1) g ; p -= scale * m/(sqrt(v)+eps)
// decay = weight_decay * lr (optimizers.py:88-95: p += -weight_decay * lr * p BEFORE the update, only when the step is applied)
// 16-byte vector accesses: 16 B read x4 + 12 B written per parameter = the algorithmic 28 B/param.
__global__ __launch_bounds__(256) void radam_k(float* p, const float* g, float* m, float* v, long n4, long n,
                                                float b1, float b2, float eps, float scale, int rect, float decay,
                                                unsigned* status, const float* gflag, int count_skip) {
  // guarded step (zeggs_radam_step_guarded): a persistent sweep of this iteration gave up on this rank (sticky status word) or
  // on another one (gflag: the all-reduced flag) -> the gradients are invalid, the whole step is a no-op and is counted
  if (status && (status[0] != 0u || (gflag && gflag[0] != 0.f))) {
    if (count_skip && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(status + 1, 1u);
    return;
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f4 pv = ((f4*)p)[i], gv = ((const f4*)g)[i], mv = ((f4*)m)[i], vv = ((f4*)v)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      vv[k] = vv[k] * b2 + (1.f - b2) * gv[k] * gv[k];
      mv[k] = mv[k] * b1 + (1.f - b1) * gv[k];
      if (decay != 0.f) pv[k] += -decay * pv[k];
      pv[k] += rect ? -scale * (mv[k] / (sqrtf(vv[k]) + eps)) : -scale * mv[k];
    }
    ((f4*)p)[i] = pv; ((f4*)m)[i] = mv; ((f4*)v)[i] = vv;
  }
  // tail
  if (blockIdx.x == 0 && threadIdx.x < (n - n4 * 4)) {
    long i = n4 * 4 + threadIdx.x;
    float vv = v[i] * b2 + (1.f - b2) * g[i] * g[i];
    float mv = m[i] * b1 + (1.f - b1) * g[i];
    if (decay != 0.f) p[i] += -decay * p[i];
    p[i] += rect ? -scale * (mv / (sqrtf(vv) + eps)) : -scale * mv;
    m[i] = mv; v[i] = vv;
  }
}

// dst[0] = 1 if a sticky give-up bit is set, else 0: the float that travels with the gradient all-reduce (sum over ranks)
__global__ void status_flag_k(const unsigned* status, float* dst) {
  if (threadIdx.x == 0) dst[0] = status[0] != 0u ? 1.f : 0.f;
}

__global__ void vae_fwd_k(const float* enc, const float* eps, float* z, float* mu_out, float* lv_out, int B, int S,
                          float temp) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * S) return;
  int b = i / S, c = i % S;
  float mu = enc[b * 2 * S + c], lv = enc[b * 2 * S + S + c];
  z[i] = mu + eps[i] * (expf(0.5f * lv) / temp);
  if (mu_out) mu_out[i] = mu;
  if (lv_out) lv_out[i] = lv;
}
__global__ void vae_bwd_k(const float* enc, const float* eps, const float* dz, const float* dmu, const float* dlv,
                          float* denc, int B, int S, float temp) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * S) return;
  int b = i / S, c = i % S;
  float lv = enc[b * 2 * S + S + c];
  float g = dz ? dz[i] : 0.f;
  denc[b * 2 * S + c] = g + (dmu ? dmu[i] : 0.f);
  denc[b * 2 * S + S + c] = g * eps[i] * 0.5f * (expf(0.5f * lv) / temp) + (dlv ? dlv[i] : 0.f);
}

// out[b][t][:] = frames[starts[b] + t][:]
// One (wave-)row of the output per loop trip, the columns over the lanes: the element-indexed form of round 1 spent its time on
// two 64-bit divisions per 4-byte copy (75 us for the 37 MB of a batch's pose windows: 0.5 TB/s, ALU-bound; this form: no
// division in the inner loop).  Rows are dealt to waves round-robin.
__global__ __launch_bounds__(256) void gather_windows_k(const float* frames, int width, const int64_t* starts, int B, int T, float* out) {
  const long nrows = (long)B * T;
  const int lane = threadIdx.x & 63;
  for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < nrows; r += (long)gridDim.x * 4) {
    const long b = r / T;
    const int t = (int)(r - b * T);
    const float* src = frames + (starts[b] + t) * width;
    float* dst = out + r * width;
    for (int c = lane; c < width; c += 64) dst[c] = src[c];
  }
}
__global__ __launch_bounds__(256) void gather_rows_k(const float* frames, int width, const int64_t* rows, long nrows, float* out, int out_ld) {
  const int lane = threadIdx.x & 63;
  for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < nrows; r += (long)gridDim.x * 4) {
    const float* src = frames + rows[r] * width;
    float* dst = out + r * out_ld;
    for (int c = lane; c < width; c += 64) dst[c] = src[c];
  }
}

// the style example of a batch in ONE pass, written in the padded row order the style encoder's first convolution reads
// ([B][pad + L + pad][ow]): out[b][pad + l][c] = ((c < width ? frames[rows[b L + l]][c] : 0) - mean[c]) / std[c], edge rows zero --
// what fill + gather_rows + normalize_rows + the encoder's own padding copy did in four passes over the 56 MB (the same
// arithmetic per element: bit-identical).  One wave per output row.
__global__ __launch_bounds__(256) void gather_example_k(const float* frames, int width, const int64_t* rows, int B, int L,
                                                        const float* mean, const float* stdv, float* out, int ow, int pad) {
  const int lane = threadIdx.x & 63, LP = L + 2 * pad;
  const long nrows = (long)B * LP;
  for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < nrows; r += (long)gridDim.x * 4) {
    const long b = r / LP;
    const int l = (int)(r - b * LP) - pad;
    float* dst = out + r * ow;
    if (l < 0 || l >= L) {
      for (int c = lane; c < ow; c += 64) dst[c] = 0.f;
      continue;
    }
    const float* src = frames + rows[b * L + l] * width;
    for (int c = lane; c < ow; c += 64) {
      const float x = c < width ? src[c] : 0.f;
      dst[c] = (x - mean[c]) / stdv[c];
    }
  }
}

// x[r][c] = (x[r][c] - mean[c]) / std[c]   (std == null: scalar std)
__global__ void normalize_rows_k(float* x, long rows, int width, long ld, const float* mean, const float* stdv,
                                 float std_scalar) {
  const int lane = threadIdx.x & 63;
  for (long r = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * (blockDim.x >> 6)) {
    float* xr = x + r * ld;
    for (int c = lane; c < width; c += 64) {
      const float sd = stdv ? stdv[c] : std_scalar;
      xr[c] = (xr[c] - mean[c]) / sd;
    }
  }
}

// dst = alpha * (*dev_scale, when given) * src   (gradient scaling by the upstream scalar of loss.backward(), copies)
__global__ void scale_copy_k(float* dst, const float* src, long n, const float* dev_scale, float alpha) {
  const float a = dev_scale ? alpha * dev_scale[0] : alpha;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = a * src[i];
}
// standard normal samples from the counter hash (two 24-bit uniforms -> Box-Muller); element i depends on (seed, i) only
__global__ void randn_k(float* out, long n, uint64_t seed) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float u1 = ((float)(hash_u32(seed, 2 * (uint64_t)i) >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(hash_u32(seed, 2 * (uint64_t)i + 1) >> 8) + 0.5f) * (1.0f / 16777216.0f);
    out[i] = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
  }
}
// out[b][t][:] = z[b][:]  (style encoding repeated over the window, ZEGGS/train.py:256) and its adjoint
__global__ void bcast_time_k(float* out, const float* z, int B, int T, int S) {
  long n = (long)B * T * S;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % S);
    long b = i / ((long)T * S);
    out[i] = z[b * S + c];
  }
}
__global__ void sum_time_k(float* dz, const float* dout, int B, int T, int S) {
  // one workgroup per batch row; thread (c, part) sums a strided share of the T rows, LDS combine
  extern __shared__ float sm[];
  const int b = blockIdx.x, parts = blockDim.x / S;
  const int c = threadIdx.x % S, part = threadIdx.x / S;
  float acc = 0.f;
  if (part < parts)
    for (int t = part; t < T; t += parts) acc += dout[((long)b * T + t) * S + c];
  sm[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < S) {
    float s = 0.f;
    for (int q = 0; q < parts; ++q) s += sm[q * S + threadIdx.x];
    dz[(long)b * S + threadIdx.x] = s;
  }
}

// Measurement / test hook: a stand-in CO-TENANT -- what a collective's workgroups are to the other kernels on the chip: resident for
// a stretch of wall-clock time on their CUs (so a kernel that needs whole CUs cannot be placed there, and a stream-K product whose
// equal-share workgroups cannot ALL be resident waits for its stragglers), with light memory traffic.  Bounded by the 100 MHz wall
// clock, whatever happens around it.  tools/cotenant_probe.py, tests/test_gpu_giveup.py.
__global__ void cotenant_k(float* buf, long n, unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) % (n > 0 ? n : 1);
  float v = 0.f;
  while (wall_clock64() - t0 < ticks) {
    if (n > 0) { v += buf[i]; buf[i] = v * 0.5f; i = (i + 4099) % n; }
    __builtin_amdgcn_s_sleep(8);
  }
}

inline int g1(long n) { long g = (n + 255) / 256; return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }

}  // namespace

extern "C" int zeggs_radam_step(float* p, const float* g, float* m, float* v, long n, float beta1, float beta2,
                                float eps, float step_scale, int rectified, void* stream) {
  ZCHECK(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0, "radam: buffers must be 16-byte aligned");
  if (n <= 0) return 0;
  long n4 = n / 4;
  hipLaunchKernelGGL(radam_k, dim3(g1(n4 > 0 ? n4 : 1)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, n, beta1,
                     beta2, eps, step_scale, rectified, 0.f, (unsigned*)nullptr, (const float*)nullptr, 0);
  ZLAUNCH_CHECK("radam");
  return 0;
}
extern "C" int zeggs_radam_step_wd(float* p, const float* g, float* m, float* v, long n, float beta1, float beta2, float eps,
                                   float step_scale, int rectified, float decay, unsigned* status, const float* gflag,
                                   int count_skip, void* stream) {
  ZCHECK(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0, "radam: buffers must be 16-byte aligned");
  if (n <= 0) return 0;
  long n4 = n / 4;
  hipLaunchKernelGGL(radam_k, dim3(g1(n4 > 0 ? n4 : 1)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, n, beta1,
                     beta2, eps, step_scale, rectified, decay, status, status ? gflag : (const float*)nullptr, count_skip);
  ZLAUNCH_CHECK("radam");
  return 0;
}
extern "C" int zeggs_radam_step_guarded_part(float* p, const float* g, float* m, float* v, long n, float beta1, float beta2,
                                             float eps, float step_scale, int rectified, unsigned* status,
                                             const float* gflag, int count_skip, void* stream) {
  ZCHECK(status != nullptr, "radam (guarded): the status words are required");
  return zeggs_radam_step_wd(p, g, m, v, n, beta1, beta2, eps, step_scale, rectified, 0.f, status, gflag, count_skip, stream);
}
extern "C" int zeggs_radam_step_guarded(float* p, const float* g, float* m, float* v, long n, float beta1, float beta2,
                                        float eps, float step_scale, int rectified, unsigned* status, const float* gflag,
                                        void* stream) {
  return zeggs_radam_step_guarded_part(p, g, m, v, n, beta1, beta2, eps, step_scale, rectified, status, gflag, 1, stream);
}
extern "C" int zeggs_test_cotenant(int workgroups, int threads, float ms, float* scratch, long n, void* stream) {
  ZCHECK(workgroups >= 1 && workgroups <= 4096 && threads >= 64 && threads <= 1024 && ms > 0.f && ms <= 500.f,
         "test_cotenant: bad shape (%d x %d for %g ms)", workgroups, threads, ms);
  hipLaunchKernelGGL(cotenant_k, dim3(workgroups), dim3(threads), 0, (hipStream_t)stream, scratch, scratch ? n : 0,
                     (unsigned long long)(ms * 1e5f));
  ZLAUNCH_CHECK("test_cotenant");
  return 0;
}
extern "C" int zeggs_status_flag(const unsigned* status, float* dst, void* stream) {
  hipLaunchKernelGGL(status_flag_k, dim3(1), dim3(64), 0, (hipStream_t)stream, status, dst);
  ZLAUNCH_CHECK("status_flag");
  return 0;
}

extern "C" int zeggs_vae_reparam_fwd(const float* enc, const float* eps, float* z, float* mu_out, float* logvar_out,
                                     int B, int S, float temperature, void* stream) {
  hipLaunchKernelGGL(vae_fwd_k, dim3(cdiv((long)B * S, 256)), dim3(256), 0, (hipStream_t)stream, enc, eps, z, mu_out,
                     logvar_out, B, S, temperature);
  ZLAUNCH_CHECK("vae_fwd");
  return 0;
}
extern "C" int zeggs_vae_reparam_bwd(const float* enc, const float* eps, const float* dz, const float* dmu,
                                     const float* dlogvar, float* denc, int B, int S, float temperature, void* stream) {
  hipLaunchKernelGGL(vae_bwd_k, dim3(cdiv((long)B * S, 256)), dim3(256), 0, (hipStream_t)stream, enc, eps, dz, dmu,
                     dlogvar, denc, B, S, temperature);
  ZLAUNCH_CHECK("vae_bwd");
  return 0;
}

extern "C" int zeggs_gather_windows(const float* frames, int width, const int64_t* starts, int B, int T, float* out,
                                    void* stream) {
  long n = (long)B * T * width;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(gather_windows_k, dim3(g1((long)B * T * 64)), dim3(256), 0, (hipStream_t)stream, frames, width, starts, B, T, out);
  ZLAUNCH_CHECK("gather_windows");
  return 0;
}
extern "C" int zeggs_gather_rows(const float* frames, int width, const int64_t* rows, long nrows, float* out, int out_ld,
                                 void* stream) {
  long n = nrows * width;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(gather_rows_k, dim3(g1(nrows * 64)), dim3(256), 0, (hipStream_t)stream, frames, width, rows, nrows, out,
                     out_ld);
  ZLAUNCH_CHECK("gather_rows");
  return 0;
}

extern "C" int zeggs_gather_example(const float* frames, int width, const int64_t* rows, int B, int L, const float* mean,
                                    const float* stdv, float* out, int out_width, int pad, void* stream) {
  ZCHECK(B >= 0 && L >= 0 && pad >= 0 && out_width >= width && width > 0, "gather_example: B %d L %d pad %d width %d -> %d", B, L,
         pad, width, out_width);
  const long nrows = (long)B * (L + 2 * pad);
  if (nrows <= 0) return 0;
  hipLaunchKernelGGL(gather_example_k, dim3(g1(nrows * 64)), dim3(256), 0, (hipStream_t)stream, frames, width, rows, B, L, mean,
                     stdv, out, out_width, pad);
  ZLAUNCH_CHECK("gather_example");
  return 0;
}

extern "C" int zeggs_normalize_rows(float* x, long rows, int width, long ld, const float* mean, const float* stdv,
                                    float std_scalar, void* stream) {
  long n = rows * width;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(normalize_rows_k, dim3(g1(rows * 64)), dim3(256), 0, (hipStream_t)stream, x, rows, width, ld, mean, stdv,
                     std_scalar);
  ZLAUNCH_CHECK("normalize_rows");
  return 0;
}

extern "C" int zeggs_fill(float* dst, long n, float value, void* stream) {
  if (n <= 0) return 0;
  return k_fill(dst, n, value, (hipStream_t)stream);      // 16-byte stores (kernels.hip)
}
extern "C" int zeggs_scale_copy(float* dst, const float* src, long n, const float* dev_scale, float alpha, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(scale_copy_k, dim3(g1(n)), dim3(256), 0, (hipStream_t)stream, dst, src, n, dev_scale, alpha);
  ZLAUNCH_CHECK("scale_copy");
  return 0;
}
extern "C" int zeggs_randn(float* out, long n, uint64_t seed, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(randn_k, dim3(g1(n)), dim3(256), 0, (hipStream_t)stream, out, n, seed);
  ZLAUNCH_CHECK("randn");
  return 0;
}
extern "C" int zeggs_broadcast_time(float* out, const float* z, int B, int T, int S, void* stream) {
  long n = (long)B * T * S;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(bcast_time_k, dim3(g1(n)), dim3(256), 0, (hipStream_t)stream, out, z, B, T, S);
  ZLAUNCH_CHECK("broadcast_time");
  return 0;
}
extern "C" int zeggs_sum_time(float* dz, const float* dout, int B, int T, int S, void* stream) {
  ZCHECK(S >= 1 && S <= 1024, "sum_time: 1 <= S <= 1024");
  if (B <= 0) return 0;
  int parts = 1024 / S;
  if (parts > T) parts = T > 0 ? T : 1;
  const int threads = parts * S;
  hipLaunchKernelGGL(sum_time_k, dim3(B), dim3(threads), threads * sizeof(float), (hipStream_t)stream, dz, dout, B, T, S);
  ZLAUNCH_CHECK("sum_time");
  return 0;
}

extern "C" int zeggs_dropout(float* x, long n, float p, uint64_t seed, void* stream) {
  ZCHECK(p >= 0.f && p < 1.f, "dropout: 0 <= p < 1");
  if (n <= 0) return 0;
  return k_dropout(x, n, p, seed, (hipStream_t)stream);
}
