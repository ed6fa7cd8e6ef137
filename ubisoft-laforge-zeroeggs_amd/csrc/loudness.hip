// BS.1770 loudness normalisation on the device: the pre-pass of preprocess_audio (ZEGGS/data_pipeline.py:34-39 ->
// pyloudnorm 0.1.0 Meter.integrated_loudness + normalize.loudness, restated in oracle/loudness.py).
//
// The two K-weighting biquads are IIR recurrences (scipy.signal.lfilter, direct form II transposed, float64); a 30-minute
// clip has 28.8 M samples, so the recurrence is cut into chunks that run in parallel:
//   pass A  every chunk from a ZERO state -> its final state (the zero-state response's contribution)
//   pass P  one thread chains the chunks: state_{c+1} = A^L state_c + zs_c   (A^L: the homogeneous L-sample transition, a
//           2x2 matrix computed on the host from the coefficients)
//   pass B  every chunk again from its TRUE initial state, writing the filtered samples
// -- the same arithmetic per sample as the sequential filter, only the chunk-initial states carry an extra rounding of
// ~1e-16 relative.  Then one workgroup per 400 ms gating block sums the squares (block bounds are passed in: they are
// truncated floating-point products in pyloudnorm and are computed on the host with its exact expression), and one
// workgroup applies the absolute (-70 LUFS) and relative (-10 LU) gates and writes the gain 10^((target - L)/20).
// Mono only (the reference's audio is mono); `f32_stages` reproduces pyloudnorm's behaviour on float32 input, where each
// filter stage's output is stored back into a float32 array.
#include "../../include/zeggs_hip.h"
#include "common.h"

namespace {

struct Biquad { double b0, b1, b2, a1, a2; };

// one chunk of L samples from state (s1, s2); y == nullptr: state only
__device__ __forceinline__ void biquad_run(const Biquad& q, const double* x, double* y, long n, double& s1, double& s2,
                                           int f32out) {
  for (long i = 0; i < n; ++i) {
    const double xi = x[i];
    const double yi = q.b0 * xi + s1;
    s1 = q.b1 * xi - q.a1 * yi + s2;
    s2 = q.b2 * xi - q.a2 * yi;
    if (y) y[i] = f32out ? (double)(float)yi : yi;
  }
}
__global__ void bq_pass_a_k(Biquad q, const double* x, long n, long L, double* zs /* [C][2] */) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long lo = c * L;
  if (lo >= n) return;
  double s1 = 0.0, s2 = 0.0;
  biquad_run(q, x + lo, nullptr, (lo + L < n ? L : n - lo), s1, s2, 0);
  zs[2 * c] = s1; zs[2 * c + 1] = s2;
}
__global__ void bq_pass_p_k(const double* zs, double* st /* [C][2] true initial states */, long C, double m00, double m01,
                            double m10, double m11) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  double s1 = 0.0, s2 = 0.0;
  for (long c = 0; c < C; ++c) {
    st[2 * c] = s1; st[2 * c + 1] = s2;
    const double n1 = m00 * s1 + m01 * s2 + zs[2 * c], n2 = m10 * s1 + m11 * s2 + zs[2 * c + 1];
    s1 = n1; s2 = n2;
  }
}
__global__ void bq_pass_b_k(Biquad q, const double* x, double* y, long n, long L, const double* st, int f32out) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long lo = c * L;
  if (lo >= n) return;
  double s1 = st[2 * c], s2 = st[2 * c + 1];
  biquad_run(q, x + lo, y + lo, (lo + L < n ? L : n - lo), s1, s2, f32out);
}
__global__ void widen_k(const float* x, double* y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = (double)x[i];
}
// z[j] = sum(y[lo_j:hi_j]^2) / (T_g * rate)
__global__ __launch_bounds__(256) void block_energy_k(const double* y, const long* lo, const long* hi, double inv_len,
                                                       double* z) {
  __shared__ double sm[256];
  const int j = blockIdx.x;
  double acc = 0.0;
  for (long i = lo[j] + threadIdx.x; i < hi[j]; i += 256) acc += y[i] * y[i];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) z[j] = sm[0] * inv_len;
}
__device__ double block_sum2(double a, double b, double* sm, double& outb) {   // sums of (a, b) over the workgroup
  sm[threadIdx.x] = a; sm[1024 + threadIdx.x] = b;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { sm[threadIdx.x] += sm[threadIdx.x + o]; sm[1024 + threadIdx.x] += sm[1024 + threadIdx.x + o]; }
    __syncthreads();
  }
  const double ra = sm[0];
  outb = sm[1024];
  __syncthreads();
  return ra;
}
// absolute + relative gate (pyloudnorm meter.py), result[0] = LUFS, result[1] = gain
__global__ __launch_bounds__(1024) void gate_k(const double* z, int nblocks, double target, double* result, float* gain32) {
  __shared__ double sm[2048];
  double s = 0.0, c = 0.0, cnt;
  for (int j = threadIdx.x; j < nblocks; j += 1024) {
    const double l = -0.691 + 10.0 * log10(z[j]);
    if (l >= -70.0) { s += z[j]; c += 1.0; }
  }
  const double tot = block_sum2(s, c, sm, cnt);
  const double gamma_r = -0.691 + 10.0 * log10(tot / cnt) - 10.0;      // cnt == 0 -> NaN -> nothing passes, as in pyloudnorm
  s = 0.0; c = 0.0;
  for (int j = threadIdx.x; j < nblocks; j += 1024) {
    const double l = -0.691 + 10.0 * log10(z[j]);
    if (l > gamma_r && l > -70.0) { s += z[j]; c += 1.0; }
  }
  const double tot2 = block_sum2(s, c, sm, cnt);
  if (threadIdx.x == 0) {
    const double zavg = cnt > 0.0 ? tot2 / cnt : 0.0;                  // nan_to_num(mean of empty) = 0
    const double lufs = -0.691 + 10.0 * log10(zavg);
    result[0] = lufs;
    result[1] = pow(10.0, (target - lufs) / 20.0);
    *gain32 = (float)result[1];
  }
}

}  // namespace

extern "C" size_t zeggs_loudness_workspace_bytes(long n_samples, int nblocks, long chunk) {
  const long C = (n_samples + chunk - 1) / chunk;
  return (size_t)(2 * n_samples + 4 * C + nblocks + 16) * sizeof(double) + 1024;
}

// coef[10] = stage 1 (b0 b1 b2 a1 a2), stage 2; trans[8] = A^chunk of stage 1, stage 2 (row-major 2x2), both from the host
extern "C" int zeggs_loudness_gain(const float* wav, long n_samples, int rate, double target, const double* coef,
                                   const double* trans, long chunk, const long* blk_lo, const long* blk_hi, int nblocks,
                                   int f32_stages, double* result /* device [2]: LUFS, gain */, float* gain32 /* device */,
                                   void* ws, size_t ws_bytes, void* stream) {
  ZCHECK(n_samples > 0 && nblocks >= 0 && chunk >= 64, "loudness: bad sizes");
  ZCHECK(ws_bytes >= zeggs_loudness_workspace_bytes(n_samples, nblocks, chunk), "loudness: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  Arena a(ws, ws_bytes);
  const long C = (n_samples + chunk - 1) / chunk;
  double* x = (double*)a.raw(n_samples * sizeof(double));
  double* y = (double*)a.raw(n_samples * sizeof(double));
  double* zs = (double*)a.raw(2 * C * sizeof(double));
  double* st = (double*)a.raw(2 * C * sizeof(double));
  double* z = (double*)a.raw((nblocks + 1) * sizeof(double));
  long g = (n_samples + 255) / 256;
  hipLaunchKernelGGL(widen_k, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, s, wav, x, n_samples);
  const unsigned cb = (unsigned)((C + 63) / 64);
  double *in = x, *out = y;
  for (int stage = 0; stage < 2; ++stage) {
    const double* c = coef + 5 * stage;
    const double* m = trans + 4 * stage;
    Biquad q{c[0], c[1], c[2], c[3], c[4]};
    hipLaunchKernelGGL(bq_pass_a_k, dim3(cb), dim3(64), 0, s, q, in, n_samples, chunk, zs);
    hipLaunchKernelGGL(bq_pass_p_k, dim3(1), dim3(64), 0, s, zs, st, C, m[0], m[1], m[2], m[3]);
    hipLaunchKernelGGL(bq_pass_b_k, dim3(cb), dim3(64), 0, s, q, in, out, n_samples, chunk, st, f32_stages);
    double* t = in; in = out; out = t;
  }
  ZLAUNCH_CHECK("loudness filters");
  if (nblocks > 0)
    hipLaunchKernelGGL(block_energy_k, dim3(nblocks), dim3(256), 0, s, in, blk_lo, blk_hi, 1.0 / (0.4 * rate), z);
  hipLaunchKernelGGL(gate_k, dim3(1), dim3(1024), 0, s, z, nblocks, target, result, gain32);
  ZLAUNCH_CHECK("loudness gate");
  return 0;
}
