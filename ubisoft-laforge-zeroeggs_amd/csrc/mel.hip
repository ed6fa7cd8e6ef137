// Audio front-end on the GPU: wav -> [n_frames, n_mels + 1] features (log-mel + energy at the animation rate).
//
// Reference (all float64 NumPy): ZEGGS/audio/spectrograms.py:216-269 (symmetric Hann, reflect padding,
// per-frame rfft, |.|/n_fft), :161-183 + :386-443 (Slaney mel filterbank), :57-131 (clip, dB, map to
// [0,1]); ZEGGS/data_pipeline.py:62-80 (10**(x/20) -> ln, linear resampling to the animation rate,
// energy = ||exp(mel)||_2 resampled with extrapolation).
//
// The reference's spectrogram is float64, so this kernel computes in float64 as well (MI355X runs f64
// FMAs at full vector rate): one workgroup per STFT frame, the windowed frame and an n_fft-entry
// cos/sin table are staged in LDS, each thread evaluates a direct DFT for its bins (index k*n mod n_fft
// walks the table, no trigonometry in the loop), then the 80 x 401 filterbank (742 non-zeros, given
// as a dense matrix resident in L2) and the log/normalise chain run on the same workgroup.
#include <math.h>

#include "../../include/zeggs_hip.h"
#include "common.h"

namespace {

// ---- matrix-core DFT (mel_stft_mfma_k): 32 STFT frames per workgroup, the DFT as a [frames, n_fft] x [n_fft, 2 bins] fp64
// product on v_mfma_f64_16x16x4_f64 against a cos / -sin table built once per call
constexpr int MF = 32, MWAVES = 8, MTPW = 7, MCOLP = MWAVES * MTPW * 16;      // frames, waves, column tiles per wave, 896 columns
constexpr int MFB_CAP = 2048;                                                  // filterbank non-zeros staged in LDS

struct MelWs {
  double* logmel;   // [M, n_mels]
  double* energy;   // [M]
  double* table;    // [n_fft, MCOLP]: column 2k = cos(2 pi j k / n_fft), 2k+1 = -sin(.), zero beyond bin n_fft/2
  double* win;      // [n_fft] symmetric Hann
};
MelWs carve_mel(const ZeggsMelDims& d, long M, Arena& a) {
  MelWs w;
  w.logmel = (double*)a.raw(sizeof(double) * M * d.n_mels);
  w.energy = (double*)a.raw(sizeof(double) * M);
  w.table = (double*)a.raw(sizeof(double) * (size_t)d.n_fft * MCOLP);
  w.win = (double*)a.raw(sizeof(double) * (size_t)d.n_fft);
  return w;
}

// spectrograms.py:233-246 (integer rule)
inline long stft_frames(long n, int n_fft, int hop) {
  long np = (n > n_fft ? n : n_fft) + 2 * (long)(n_fft / 2);
  return (np % hop == 0) ? (np - n_fft) / hop : 1 + (np - n_fft) / hop;
}

// STFT frames m0 .. m0 + gridDim.x - 1.  n = signal length for the reflect rule, n_avail = samples present in `wav`
// (streaming: n is unknown yet and passed as a huge value; the caller only asks for frames that end before n_avail)
__global__ __launch_bounds__(256) void mel_stft_k(ZeggsMelDims d, const float* wav, long n, long n_avail, const double* fb,
                                                   double* logmel, double* energy, long m0) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int NF = d.n_fft, NBIN = NF / 2 + 1;
  double* xw = sm;              // [NF] windowed samples
  double* ct = xw + NF;         // [NF] cos(2 pi j / NF)
  double* st = ct + NF;         // [NF] sin
  double* amp = st + NF;        // [NBIN]
  double* melv = amp + NBIN;    // [n_mels]
  const long fr = m0 + blockIdx.x, slot = blockIdx.x;
  const long neff = n > NF ? n : NF;   // zero-extended to n_fft when shorter (spectrograms.py:233-234)
  for (int j = threadIdx.x; j < NF; j += blockDim.x) {
    const long p = fr * d.hop + j - NF / 2;          // index into the (zero-extended) signal before reflect padding
    long src = p < 0 ? -p : (p >= neff ? 2 * (neff - 1) - p : p);
    const double x = (src >= 0 && src < n && src < n_avail) ? (double)wav[src] : 0.0;
    const double win = 0.5 - 0.5 * cos(2.0 * M_PI * (double)j / (double)(NF - 1));   // scipy hann(sym=True)
    xw[j] = x * win;
    const double ang = 2.0 * M_PI * (double)j / (double)NF;
    ct[j] = cos(ang);
    st[j] = sin(ang);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < NBIN; k += blockDim.x) {
    double re = 0.0, im = 0.0;
    int idx = 0;
    for (int j = 0; j < NF; ++j) {
      const double x = xw[j];
      re = fma(x, ct[idx], re);
      im = fma(-x, st[idx], im);
      idx += k;
      if (idx >= NF) idx -= NF;
    }
    amp[k] = sqrt(re * re + im * im) / (double)NF;     // real_amplitude (spectrograms.py:266-267)
  }
  __syncthreads();
  const double amin = (double)d.min_clip / (double)NF;  // spectrograms.py:88-90
  const double rng = -20.0 * log10(amin);
  for (int m = threadIdx.x; m < d.n_mels; m += blockDim.x) {
    const double* f = fb + (long)m * NBIN;
    double s = 0.0;
    for (int k = 0; k < NBIN; ++k) s = fma(f[k], amp[k], s);
    s = fabs(s);
    if (s < amin) s = amin;
    const double v = (20.0 * log10(s) + rng) / rng;    // spectrograms.py:121-129
    const double y = log(pow(10.0, v / 20.0));          // data_pipeline.py:62-63
    melv[m] = y;
    logmel[slot * d.n_mels + m] = y;
  }
  __syncthreads();
  if (threadIdx.x == 0) {                               // energy = || exp(mel) ||_2 (data_pipeline.py:28-30,75)
    double e = 0.0;
    for (int m = 0; m < d.n_mels; ++m) { const double z = exp(melv[m]); e += z * z; }
    energy[slot] = sqrt(e);
  }
}

typedef double d4 __attribute__((ext_vector_type(4)));

// DFT basis + window, once per call (the angle is reduced exactly: (j k) mod n_fft -- the same table entries the direct
// kernel above walks)
__global__ void mel_table_k(double* T, double* win, int NF, int NBIN) {
  const long n = (long)NF * MCOLP;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i / MCOLP), c = (int)(i % MCOLP), k = c >> 1;
    double v = 0.0;
    if (k < NBIN) {
      const double ang = 2.0 * M_PI * (double)(((long)j * k) % NF) / (double)NF;
      v = (c & 1) ? -sin(ang) : cos(ang);
    }
    T[i] = v;
    if (i < NF) win[i] = 0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)(NF - 1));   // scipy hann(sym=True)
  }
}

// LDS of mel_stft_mfma_k (bytes): raw samples of the 32 overlapping frames | window | amplitudes [32][NBIN] | staged filterbank
// non-zeros | band tables.  The log-mel values of the workgroup's frames reuse the sample area after the products.
__host__ __device__ inline size_t mel_fast_lds(int NF, int hop, int n_mels) {
  const size_t ns = (size_t)(MF - 1) * hop + NF;
  size_t xs = ns * sizeof(float), mv = (size_t)MF * n_mels * sizeof(double);
  if (mv > xs) xs = mv;
  xs = (xs + 15) / 16 * 16;
  return xs + sizeof(double) * ((size_t)NF + (size_t)MF * (NF / 2 + 1) + MFB_CAP) + sizeof(int) * 3 * (size_t)n_mels + 64;
}

// STFT frames m0 + 32 blockIdx.x ... (nfr frames in all): same results as mel_stft_k -- identical table entries, window, clip /
// log chain and summation order of the mel and energy sums; only the 800-term DFT sums are accumulated by the matrix cores
// (fp64, k in steps of 4) instead of one fma chain per bin.
__global__ __launch_bounds__(MWAVES * 64) void mel_stft_mfma_k(ZeggsMelDims d, const float* wav, long n, long n_avail,
                                                                const double* fb, const double* table, const double* wintab,
                                                                double* logmel, double* energy, long m0, long nfr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int NF = d.n_fft, NBIN = NF / 2 + 1, hop = d.hop, NM = d.n_mels;
  const int ns = (MF - 1) * hop + NF;
  size_t xs_bytes = (size_t)ns * sizeof(float), mv_bytes = (size_t)MF * NM * sizeof(double);
  if (mv_bytes > xs_bytes) xs_bytes = mv_bytes;
  xs_bytes = (xs_bytes + 15) / 16 * 16;
  float* xs = (float*)smem;                         // [ns] samples (reflect rule applied)
  double* melv = (double*)smem;                     // [MF][NM] after the products
  double* win = (double*)(smem + xs_bytes);         // [NF]
  double* amp = win + NF;                           // [MF][NBIN]
  double* fbs = amp + (size_t)MF * NBIN;            // [MFB_CAP] non-zero bands of the filterbank, mel after mel
  int* blo = (int*)(fbs + MFB_CAP);                 // [NM] first bin of the band
  int* bhi = blo + NM;                              // [NM] one past its last bin
  int* bof = bhi + NM;                              // [NM] offset of the band in fbs (-1: not staged, read from fb)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long fr0 = m0 + (long)blockIdx.x * MF, slot0 = (long)blockIdx.x * MF;
  const long neff = n > NF ? n : NF;
  for (int i = tid; i < ns; i += blockDim.x) {
    const long p = fr0 * hop + i - NF / 2;
    const long src = p < 0 ? -p : (p >= neff ? 2 * (neff - 1) - p : p);
    xs[i] = (src >= 0 && src < n && src < n_avail) ? wav[src] : 0.f;
  }
  for (int j = tid; j < NF; j += blockDim.x) win[j] = wintab[j];
  if (tid < NM) {                                   // band of mel filter tid
    const double* f = fb + (long)tid * NBIN;
    int lo = 0, hi = 0;
    for (int k = 0; k < NBIN; ++k)
      if (f[k] != 0.0) { if (hi == 0) lo = k; hi = k + 1; }
    blo[tid] = lo; bhi[tid] = hi;
  }
  __syncthreads();
  if (tid == 0) {
    int o = 0;
    for (int m = 0; m < NM; ++m) {
      const int w = bhi[m] - blo[m];
      if (o + w <= MFB_CAP) { bof[m] = o; o += w; } else bof[m] = -1;
    }
  }
  // ---- DFT: P[frame][col] = sum_j xw[frame][j] T[j][col]; this wave: column tiles wave * MTPW .. + MTPW - 1, both row tiles
  d4 acc[2][MTPW];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int t = 0; t < MTPW; ++t) acc[rt][t] = d4{0.0, 0.0, 0.0, 0.0};
  const int r16 = lane & 15, kk = lane >> 4;
  const double* tb = table + (long)kk * MCOLP + 16 * (wave * MTPW) + r16;
  double bq[MTPW];
#pragma unroll
  for (int t = 0; t < MTPW; ++t) bq[t] = tb[16 * t];
  for (int s4 = 0; s4 < NF; s4 += 4) {
    const int j = s4 + kk;
    const double wj = win[j];
    const double a0 = (double)xs[r16 * hop + j] * wj, a1 = (double)xs[(16 + r16) * hop + j] * wj;
    double bc[MTPW];
#pragma unroll
    for (int t = 0; t < MTPW; ++t) bc[t] = bq[t];
    if (s4 + 4 < NF) {
      const double* tn = tb + (long)(s4 + 4) * MCOLP;
#pragma unroll
      for (int t = 0; t < MTPW; ++t) bq[t] = tn[16 * t];
    }
#pragma unroll
    for (int t = 0; t < MTPW; ++t) {
      acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bc[t], acc[0][t], 0, 0, 0);
      acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bc[t], acc[1][t], 0, 0, 0);
    }
  }
  // accumulator layout of the f64 form: col = lane & 15, row = (lane >> 4) + 4 i.  Even column = Re, odd = Im of bin col / 2
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int t = 0; t < MTPW; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double v = acc[rt][t][i];
        const double o = __shfl_xor(v, 1, 64);
        const int col = 16 * (wave * MTPW + t) + r16, bin = col >> 1, fr = rt * 16 + kk + 4 * i;
        if (!(col & 1) && bin < NBIN) amp[(size_t)fr * NBIN + bin] = sqrt(v * v + o * o) / (double)NF;   // real_amplitude
      }
  __syncthreads();
  for (int m = tid; m < NM; m += blockDim.x) {       // filterbank bands -> LDS (the 742 non-zeros of the shipped filterbank)
    if (bof[m] >= 0)
      for (int k = blo[m]; k < bhi[m]; ++k) fbs[bof[m] + k - blo[m]] = fb[(long)m * NBIN + k];
  }
  __syncthreads();
  const double amin = (double)d.min_clip / (double)NF;
  const double rng = -20.0 * log10(amin);
  for (int it = tid; it < MF * NM; it += blockDim.x) {
    const int f = it / NM, m = it % NM;
    const double* av = amp + (size_t)f * NBIN;
    double s = 0.0;
    if (bof[m] >= 0) {
      const double* fv = fbs + bof[m] - blo[m];
      for (int k = blo[m]; k < bhi[m]; ++k) s = fma(fv[k], av[k], s);
    } else {
      const double* fv = fb + (long)m * NBIN;
      for (int k = blo[m]; k < bhi[m]; ++k) s = fma(fv[k], av[k], s);
    }
    s = fabs(s);
    if (s < amin) s = amin;
    const double v = (20.0 * log10(s) + rng) / rng;
    const double y = log(pow(10.0, v / 20.0));
    melv[it] = y;                                   // (the sample area: the products are done)
    if (slot0 + f < nfr) logmel[(slot0 + f) * NM + m] = y;
  }
  __syncthreads();
  if (tid < MF && slot0 + tid < nfr) {
    double e = 0.0;
    for (int m = 0; m < NM; ++m) { const double z = exp(melv[tid * NM + m]); e += z * z; }
    energy[slot0 + tid] = sqrt(e);
  }
}

// linear resampling at t_k = ((fs/hop)/fps) k : mel -> NaN outside the hull (griddata), energy extrapolates
// animation frames k0 .. k0 + n_frames - 1; logmel / energy hold the STFT frames m0 .. (indices relative to m0)
__global__ void mel_resample_k(ZeggsMelDims d, const double* logmel, const double* energy, long M, long m0, long k0,
                               int n_frames, float* out) {
  const int W = d.n_mels + 1;
  const long n = (long)n_frames * W;
  logmel -= m0 * d.n_mels;
  energy -= m0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % W);
    const long k = k0 + i / W;
    const double t = (((double)d.fs / (double)d.hop) / (double)d.fps) * (double)k;
    long hi = (long)ceil(t);            // searchsorted(side=left) over the integer grid
    if (hi < 1) hi = 1;
    if (hi > M - 1) hi = M - 1;
    const long lo = hi - 1;
    if (M < 2) { out[i] = nanf(""); continue; }
    if (c < d.n_mels) {
      if (t < 0.0 || t > (double)(M - 1)) { out[i] = nanf(""); continue; }
      const double ylo = logmel[lo * d.n_mels + c], yhi = logmel[hi * d.n_mels + c];
      out[i] = (float)((yhi - ylo) * (t - (double)lo) + ylo);
    } else {
      const double ylo = energy[lo], yhi = energy[hi];
      out[i] = (float)((yhi - ylo) * (t - (double)lo) + ylo);
    }
  }
}

}  // namespace

int g_mel_mfma = 1;      // zeggs_set_option("mel_mfma", 0/1): matrix-core DFT (default) / one workgroup per frame, direct DFT

// STFT frames m0 .. m0 + nfr - 1 -> logmel / energy slots 0 .. nfr - 1
static int launch_stft(const ZeggsMelDims& d, const MelWs& w, const float* wav, long n, long n_avail, const double* fb, long m0,
                       long nfr, hipStream_t s) {
  const int NBIN = d.n_fft / 2 + 1;
  const size_t fast_lds = mel_fast_lds(d.n_fft, d.hop, d.n_mels);
  if (g_mel_mfma && d.n_fft % 4 == 0 && 2 * NBIN <= MCOLP && fast_lds <= 160 * 1024 && nfr >= 1) {
    static bool attr_set = false;
    if (!attr_set) {       // more than 64 KB of dynamic LDS needs the opt-in
      hipFuncSetAttribute((const void*)mel_stft_mfma_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_set = true;
    }
    hipLaunchKernelGGL(mel_table_k, dim3(1024), dim3(256), 0, s, w.table, w.win, d.n_fft, NBIN);
    ZLAUNCH_CHECK("mel_table");
    hipLaunchKernelGGL(mel_stft_mfma_k, dim3((unsigned)((nfr + MF - 1) / MF)), dim3(MWAVES * 64), fast_lds, s, d, wav, n, n_avail,
                       fb, w.table, w.win, w.logmel, w.energy, m0, nfr);
    ZLAUNCH_CHECK("mel_stft_mfma");
    return 0;
  }
  const size_t lds = sizeof(double) * (3 * (size_t)d.n_fft + d.n_fft / 2 + 1 + d.n_mels);
  hipLaunchKernelGGL(mel_stft_k, dim3((unsigned)nfr), dim3(256), lds, s, d, wav, n, n_avail, fb, w.logmel, w.energy, m0);
  ZLAUNCH_CHECK("mel_stft");
  return 0;
}

extern "C" long zeggs_mel_stft_frames(const ZeggsMelDims* d, long n_samples) {
  return stft_frames(n_samples, d->n_fft, d->hop);
}

extern "C" size_t zeggs_mel_workspace_bytes(const ZeggsMelDims* d, long n_samples) {
  Arena a(nullptr, 0);
  carve_mel(*d, stft_frames(n_samples, d->n_fft, d->hop), a);
  return a.off + 256;
}

extern "C" int zeggs_mel_features(const ZeggsMelDims* dp, const float* wav, long n_samples, const double* filterbank,
                                  int n_frames, float* out, void* ws, size_t ws_bytes, void* stream) {
  const ZeggsMelDims& d = *dp;
  hipStream_t s = (hipStream_t)stream;
  ZCHECK(d.n_fft >= 2 && d.n_fft % 2 == 0 && d.hop > 0 && d.n_mels > 0, "mel: bad dims");
  ZCHECK(n_samples > 0 && n_frames >= 0, "mel: empty input");
  const long M = stft_frames(n_samples, d.n_fft, d.hop);
  Arena a(ws, ws_bytes);
  MelWs w = carve_mel(d, M, a);
  ZCHECK(a.ok(), "mel: workspace too small (%zu < %zu)", ws_bytes, a.off);
  ZTRY(launch_stft(d, w, wav, n_samples, n_samples, filterbank, 0L, M, s));
  if (n_frames > 0) {
    long n = (long)n_frames * (d.n_mels + 1), g = (n + 255) / 256;
    hipLaunchKernelGGL(mel_resample_k, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, s, d, w.logmel, w.energy, M,
                       0L, 0L, n_frames, out);
    ZLAUNCH_CHECK("mel_resample");
  }
  return 0;
}

// Streaming form: features of the animation frames [k0, k1) from the samples received so far.
//   final == 0: the signal continues; every STFT frame the range needs must end inside the n_samples present
//               (zeggs_mel_frames_ready tells how far that is) -- no right-edge reflection, no end clamps;
//   final != 0: n_samples is the whole signal: identical to rows k0..k1-1 of zeggs_mel_features.
extern "C" long zeggs_mel_frames_ready(const ZeggsMelDims* d, long n_samples) {
  // STFT frame m is complete when 200 m + 400 <= n; animation frame k interpolates STFT frames ceil(t)-1, ceil(t)
  const long mmax = (n_samples - d->n_fft / 2) / d->hop;           // last complete STFT frame (may be < 1)
  if (mmax < 1) return 0;
  const double r = ((double)d->fs / (double)d->hop) / (double)d->fps;
  long k = (long)floor((double)mmax / r);                           // largest k with t_k <= mmax
  while (k >= 0 && ceil(r * (double)k) > (double)mmax) --k;
  return k + 1;                                                     // frames 0 .. k are computable
}
extern "C" size_t zeggs_mel_range_workspace_bytes(const ZeggsMelDims* d, long k0, long k1) {
  const double r = ((double)d->fs / (double)d->hop) / (double)d->fps;
  const long mspan = (long)ceil(r * (double)(k1 - 1)) - (long)floor(r * (double)k0) + 4;
  Arena a(nullptr, 0);
  carve_mel(*d, mspan, a);
  return a.off + 256;
}
extern "C" int zeggs_mel_features_range(const ZeggsMelDims* dp, const float* wav, long n_samples, int final,
                                        const double* filterbank, long k0, long k1, float* out, void* ws, size_t ws_bytes,
                                        void* stream) {
  const ZeggsMelDims& d = *dp;
  hipStream_t s = (hipStream_t)stream;
  ZCHECK(d.n_fft >= 2 && d.n_fft % 2 == 0 && d.hop > 0 && d.n_mels > 0, "mel: bad dims");
  ZCHECK(n_samples > 0 && k0 >= 0 && k1 > k0, "mel range: empty input");
  if (!final) ZCHECK(k1 <= zeggs_mel_frames_ready(dp, n_samples), "mel range: frames %ld..%ld need samples not received yet", k0, k1);
  const long M = final ? stft_frames(n_samples, d.n_fft, d.hop) : (1L << 40);
  const double r = ((double)d.fs / (double)d.hop) / (double)d.fps;
  long m0 = (long)ceil(r * (double)k0) - 1, m1 = (long)ceil(r * (double)(k1 - 1)) + 1;   // [m0, m1)
  if (m0 < 0) m0 = 0;
  if (m1 < 2) m1 = 2;
  if (m1 > M) m1 = M;
  if (m0 > m1 - 2) m0 = m1 - 2 > 0 ? m1 - 2 : 0;
  Arena a(ws, ws_bytes);
  MelWs w = carve_mel(d, m1 - m0, a);
  ZCHECK(a.ok(), "mel range: workspace too small (%zu < %zu)", ws_bytes, a.off);
  ZTRY(launch_stft(d, w, wav, final ? n_samples : (1L << 50), n_samples, filterbank, m0, m1 - m0, s));
  const long n = (k1 - k0) * (d.n_mels + 1), g = (n + 255) / 256;
  hipLaunchKernelGGL(mel_resample_k, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, s, d, w.logmel, w.energy, M, m0, k0,
                     (int)(k1 - k0), out);
  ZLAUNCH_CHECK("mel_resample");
  return 0;
}
