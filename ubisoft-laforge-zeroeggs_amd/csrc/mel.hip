// Audio front-end on the GPU: wav -> [n_frames, n_mels + 1] features (log-mel + energy at the animation rate).
//
// Reference (all float64 NumPy): ZEGGS/audio/spectrograms.py:216-269 (symmetric Hann, reflect padding,
// per-frame rfft, |.|/n_fft), :161-183 + :386-443 (Slaney mel filterbank), :57-131 (clip, dB, map to
// [0,1]); ZEGGS/data_pipeline.py:62-80 (10**(x/20) -> ln, linear resampling to the animation rate,
// energy = ||exp(mel)||_2 resampled with extrapolation).
//
// Three forms of the STFT, same results (float32 features bit-identical, tools/mel_ab.py): the FFT form mel_stft_fft_k (round 4,
// default: a half-length complex mixed-radix FFT + split, what the reference's np.fft.rfft computes), the DFT as a float64
// matrix-core product mel_stft_mfma_k (round 3, option mel_fft = 0) and the direct DFT below (other n_fft).
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
  double* ftw;      // FFT form: [n_fft / 2 + n_fft / 2 + 1 + n_fft] complex (twiddles of the half-length transform, of the split, window)
  double* fbp;      // FFT form: the filterbank's non-zeros, band after band [MFB_CAP]
  int* bands;       // FFT form: [3 n_mels] first bin | one past the last bin | offset in fbp
  double* spl;      // resample_method = "cubic": second derivatives of the interpolating splines [M, n_mels + 1]
  double* cp;       //   ... and the right-hand sides of their tridiagonal systems [M, n_mels + 1]
};
MelWs carve_mel(const ZeggsMelDims& d, long M, Arena& a) {
  MelWs w;
  w.logmel = (double*)a.raw(sizeof(double) * M * d.n_mels);
  w.energy = (double*)a.raw(sizeof(double) * M);
  w.table = (double*)a.raw(sizeof(double) * (size_t)d.n_fft * MCOLP);
  w.win = (double*)a.raw(sizeof(double) * (size_t)d.n_fft);
  w.ftw = (double*)a.raw(sizeof(double) * 2 * ((size_t)2 * d.n_fft + 1));
  w.fbp = (double*)a.raw(sizeof(double) * MFB_CAP);
  w.bands = (int*)a.raw(sizeof(int) * 3 * (size_t)d.n_mels);
  w.spl = w.cp = nullptr;
  if (d.flags & 8) {      // MEL_CUBIC
    w.spl = (double*)a.raw(sizeof(double) * M * (d.n_mels + 1));
    w.cp = (double*)a.raw(sizeof(double) * M * (d.n_mels + 1));
  }
  return w;
}

// spectrograms.py:233-246 (integer rule)
// ZeggsMelDims.flags: bit 0 = audio_conf.centered is FALSE (no reflect padding: frame m starts at sample m hop, spectrograms.py:237-239),
// bit 1 = audio_conf.normalize_range is FALSE (the stored value is 20 log10 s, not mapped to [0, 1], spectrograms.py:123-129)
// bits 2-3 = audio_conf.resample_method (data_pipeline.py:65-79: scipy.interpolate.griddata for the mel table, interp1d for the energy):
// 0 "linear" (the shipped configurations), 4 "nearest", 8 "cubic"
#define MEL_UNCENTERED 1
#define MEL_RAW_RANGE 2
#define MEL_NEAREST 4
#define MEL_CUBIC 8
inline long stft_frames(long n, int n_fft, int hop, int flags = 0) {
  long np = (n > n_fft ? n : n_fft) + ((flags & MEL_UNCENTERED) ? 0 : 2 * (long)(n_fft / 2));
  return (np % hop == 0) ? (np - n_fft) / hop : 1 + (np - n_fft) / hop;
}

// STFT frames m0 .. m0 + gridDim.x - 1.  n = signal length for the reflect rule, n_avail = samples present in `wav`
// (streaming: n is unknown yet and passed as a huge value; the caller only asks for frames that end before n_avail)
// The clipped mel amplitude s -> what the reference stores: v = (20 log10 s + range) / range (spectrograms.py:121-129), then
// data_pipeline.py:62-63 takes y = ln(10^(v / 20)), and the frame's energy is || exp(y) ||_2 (data_pipeline.py:28-30,75).  Four
// float64 transcendentals per mel value (log10, pow, log, exp) bounded the front-end in round 4 (not memory).  Round 5:
//   y = v ln(10) / 20 = ln(s) / range + ln(10) / 20   -- the pow / log pair is an affine map -- and exp(y)^2 = exp(2 y);
//   mode 0: ln(s) and exp(2 y) on the hardware log2 / exp2 (v_log_f32 / v_exp_f32, 1 ulp of float32): the error of y is
//     <= 6e-8 |log2 s| ln 2 / range ~ 1e-8 absolute (range = -20 log10(min amplitude) ~ 10^2), of the energy 1e-7 relative -- the
//     features are float32 and the reference fixtures are matched at 2e-6;
//   mode 2 (default): the affine map in float64 (log10 + exp per value): float32 features bit-identical to mode 1, the literal
//   float64 chain (zeggs_set_option("mel_exact_log", m)).
__device__ __forceinline__ double mel_logamp(double s, double rng, int exact, int flags = 0) {
  if (flags & MEL_RAW_RANGE)      // normalize_range = false: v = 20 log10 s, y = ln(10^(v / 20)) = ln s
    return exact == 0 ? (double)(__builtin_amdgcn_logf((float)s) * 0.6931471805599453f) : exact == 1 ? log(pow(10.0, (20.0 * log10(s)) / 20.0)) : log(s);
  if (exact == 0) return (double)(__builtin_amdgcn_logf((float)s) * 0.6931471805599453f) / rng + (2.302585092994046 / 20.0);
  const double v = (20.0 * log10(s) + rng) / rng;
  return exact == 1 ? log(pow(10.0, v / 20.0)) : v * (2.302585092994046 / 20.0);
}
__device__ __forceinline__ double mel_exp_sq(double y, int exact) {      // exp(y)^2
  if (exact == 0) return (double)__builtin_amdgcn_exp2f((float)(y * (2.0 * 1.4426950408889634)));
  const double z = exp(y);
  return z * z;
}

// sample `src` of the signal the STFT sees: the wav, zero beyond its end, high-pass filtered first when audio_conf.pre_emphasis is on
// (spectrograms.py:35 -> signal_manipulation.preemphasis = lfilter([1, -c], [1], x): y[n] = x[n] - c x[n-1], y[0] = x[0], in float64)
__device__ __forceinline__ double mel_sample(const float* wav, long src, long n, long n_avail, double pe) {
  if (!(src >= 0 && src < n && src < n_avail)) return 0.0;
  double x = (double)wav[src];
  if (pe != 0.0 && src > 0) x -= pe * (double)wav[src - 1];
  return x;
}

__global__ __launch_bounds__(256) void mel_stft_k(ZeggsMelDims d, const float* wav, long n, long n_avail, const double* fb,
                                                   double* logmel, double* energy, long m0, int exact) {
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
    const long p = fr * d.hop + j - ((d.flags & MEL_UNCENTERED) ? 0 : NF / 2);          // index into the (zero-extended) signal before reflect padding
    long src = p < 0 ? -p : (p >= neff ? 2 * (neff - 1) - p : p);
    const double x = mel_sample(wav, src, n, n_avail, d.pre_emph);
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
    const double y = mel_logamp(s, rng, exact, d.flags);
    melv[m] = y;
    logmel[slot * d.n_mels + m] = y;
  }
  __syncthreads();
  if (threadIdx.x == 0) {                               // energy = || exp(mel) ||_2 (data_pipeline.py:28-30,75)
    double e = 0.0;
    for (int m = 0; m < d.n_mels; ++m) e += mel_exp_sq(melv[m], exact);
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
                                                                double* logmel, double* energy, long m0, long nfr, int exact) {
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
    const long p = fr0 * hop + i - ((d.flags & MEL_UNCENTERED) ? 0 : NF / 2);
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
    const double y = mel_logamp(s, rng, exact, d.flags);
    melv[it] = y;                                   // (the sample area: the products are done)
    if (slot0 + f < nfr) logmel[(slot0 + f) * NM + m] = y;
  }
  __syncthreads();
  if (tid < MF && slot0 + tid < nfr) {
    double e = 0.0;
    for (int m = 0; m < NM; ++m) e += mel_exp_sq(melv[tid * NM + m], exact);
    energy[slot0 + tid] = sqrt(e);
  }
}


// ---- FFT form (mel_stft_fft_k, round 4): what the reference's np.fft.rfft does (spectrograms.py:251-263) instead of the
// O(N^2) DFT as a matrix product -- an n_fft-point real transform as ONE complex FFT of half the length (even samples real part,
// odd samples imaginary part), mixed-radix Stockham autosort in LDS (radices 4 / 5 / 2: 400 = 4 4 5 5 for the shipped
// n_fft = 800), then the usual split X[k] = (Z[k] + conj Z[M-k]) / 2 - i / 2 e^{-2 pi i k / N} (Z[k] - conj Z[M-k]).
// 19 kFLOP per STFT frame instead of 1.28 MFLOP; the launch is bound by the log / exp chain of the 80 mel values and by LDS
// traffic, not by the transform.  Twiddles come from two tables built once per call in float64 (exact angle reduction), so
// the spectrum agrees with the matrix form to a few ulp; mel / clip / log chain and summation orders are the same code.
constexpr int FFB = 4;            // STFT frames per workgroup
#ifndef ZEGGS_MEL_FFT_THREADS
#define ZEGGS_MEL_FFT_THREADS 320 // 5 waves: the radix-5 stages (4 x 80 butterflies) and the mel / log chain (4 x 80 values) are ONE pass each
#endif
constexpr int FTHR = ZEGGS_MEL_FFT_THREADS;
constexpr int FMAXST = 8;         // stages
struct FftPlan { int nst; int radix[FMAXST]; };
typedef double c2 __attribute__((ext_vector_type(2)));     // (re, im)
__device__ __forceinline__ c2 cmul(c2 a, c2 b) { return c2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ c2 cmul_mi(c2 a) { return c2{a.y, -a.x}; }      // a * (-i)

// TW[n] = exp(-2 pi i n / M), n < M;  TW[M + k] = exp(-2 pi i k / (2 M)), k <= M;  TW[2 M + 1 + j] = (hann window j, 0), j < 2 M.
// Block 0 also packs the filterbank ONCE per call: bands[m] = first bin, bands[NM + m] = one past the last bin, bands[2 NM + m] =
// offset of the band's non-zeros in fbp (-1: does not fit, read from the dense matrix) -- every STFT workgroup copies these few
// KB into LDS instead of scanning the dense [n_mels, n_bins] matrix itself (that scan was 2/3 of the first version's time).
__global__ void mel_fft_table_k(c2* TW, int M, const double* fb, int NM, int* bands, double* fbp) {
  const int NF = 2 * M, n = 2 * M + 1 + NF, NBIN = M + 1;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (i < M) { const double a = 2.0 * M_PI * (double)i / (double)M; TW[i] = c2{cos(a), -sin(a)}; }
    else if (i <= 2 * M) { const double a = 2.0 * M_PI * (double)(i - M) / (double)NF; TW[i] = c2{cos(a), -sin(a)}; }
    else { const int j = i - 2 * M - 1; TW[i] = c2{0.5 - 0.5 * cos(2.0 * M_PI * (double)j / (double)(NF - 1)), 0.0}; }
  }
  if (blockIdx.x == 0) {
    // first / last non-zero bin of every mel filter: all threads walk the dense matrix together (min / max through global atomics
    // on the band table itself: a serial scan per filter was 0.13 ms of a 1.7 ms front-end)
    for (int m = threadIdx.x; m < NM; m += blockDim.x) { bands[m] = NBIN; bands[NM + m] = 0; }
    __syncthreads();
    for (int i = threadIdx.x; i < NM * NBIN; i += blockDim.x) {
      const int m = i / NBIN, k = i - m * NBIN;
      if (fb[i] != 0.0) { atomicMin(bands + m, k); atomicMax(bands + NM + m, k + 1); }
    }
    __syncthreads();
    for (int m = threadIdx.x; m < NM; m += blockDim.x)
      if (bands[NM + m] == 0) bands[m] = 0;            // an all-zero filter: empty band [0, 0)
    __syncthreads();
    if (threadIdx.x == 0) {
      int o = 0;
      for (int m = 0; m < NM; ++m) {
        const int w = bands[NM + m] - bands[m];
        if (o + w <= MFB_CAP) { bands[2 * NM + m] = o; o += w; } else bands[2 * NM + m] = -1;
      }
    }
    __syncthreads();
    for (int m = threadIdx.x; m < NM; m += blockDim.x)
      if (bands[2 * NM + m] >= 0)
        for (int k = bands[m]; k < bands[NM + m]; ++k) fbp[bands[2 * NM + m] + k - bands[m]] = fb[(long)m * NBIN + k];
  }
}

__host__ __device__ inline size_t mel_fft_lds(int NF, int n_mels) {
  return sizeof(double) * 2 * ((size_t)2 * FFB * (NF / 2)) + sizeof(double) * MFB_CAP + sizeof(int) * 3 * (size_t)n_mels + 64;
}

__global__ __launch_bounds__(FTHR) void mel_stft_fft_k(ZeggsMelDims d, FftPlan plan, const float* wav, long n, long n_avail,
                                                      const double* fb, const c2* __restrict__ TW, const int* bands,
                                                      const double* fbp, double* logmel, double* energy, long m0, long nfr, int exact) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int NF = d.n_fft, M = NF / 2, NBIN = M + 1, hop = d.hop, NM = d.n_mels;
  c2* bufA = (c2*)smem;                               // [FFB][M]
  c2* bufB = bufA + (size_t)FFB * M;                  // [FFB][M]
  double* fbs = (double*)(bufB + (size_t)FFB * M);    // [MFB_CAP]
  int* blo = (int*)(fbs + MFB_CAP);
  int* bhi = blo + NM;
  int* bof = bhi + NM;
  const int tid = threadIdx.x;
  const long fr0 = m0 + (long)blockIdx.x * FFB, slot0 = (long)blockIdx.x * FFB;
  const long neff = n > NF ? n : NF;
  const c2* WIN = TW + 2 * M + 1;
  // windowed samples, packed: z[m] = xw[2 m] + i xw[2 m + 1]
  for (int i = tid; i < FFB * M; i += blockDim.x) {
    const int f = i / M, m = i - f * M;
    double v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = 2 * m + h;
      const long p = (fr0 + f) * hop + j - ((d.flags & MEL_UNCENTERED) ? 0 : NF / 2);
      const long src = p < 0 ? -p : (p >= neff ? 2 * (neff - 1) - p : p);
      const double x = mel_sample(wav, src, n, n_avail, d.pre_emph);
      v[h] = x * WIN[j].x;
    }
    bufA[i] = c2{v[0], v[1]};
  }
  for (int i = tid; i < 3 * NM; i += blockDim.x) blo[i] = bands[i];          // blo | bhi | bof are contiguous
  for (int i = tid; i < MFB_CAP; i += blockDim.x) fbs[i] = fbp[i];
  __syncthreads();
  // ---- Stockham stages: x -> y, natural order in, natural order out
  c2* x = bufA;
  c2* y = bufB;
  int Ns = 1;
  for (int st = 0; st < plan.nst; ++st) {
    const int r = plan.radix[st], t = M / r, tws = M / (Ns * r);      // tws: table stride of this stage's twiddle
    for (int it = tid; it < FFB * t; it += blockDim.x) {
      const int f = it / t, i = it - f * t;
      const int k = i % Ns, j = (i - k) * r + k;
      const c2* xi = x + (size_t)f * M + i;
      c2* yo = y + (size_t)f * M + j;
      if (r == 4) {
        c2 u0 = xi[0], u1 = xi[t], u2 = xi[2 * t], u3 = xi[3 * t];
        if (k) { const int a = k * tws; u1 = cmul(u1, TW[a]); u2 = cmul(u2, TW[2 * a]); u3 = cmul(u3, TW[3 * a]); }
        const c2 s0 = u0 + u2, s1 = u0 - u2, s2 = u1 + u3, s3 = cmul_mi(u1 - u3);
        yo[0] = s0 + s2; yo[Ns] = s1 + s3; yo[2 * Ns] = s0 - s2; yo[3 * Ns] = s1 - s3;
      } else if (r == 5) {
        c2 u0 = xi[0], u1 = xi[t], u2 = xi[2 * t], u3 = xi[3 * t], u4 = xi[4 * t];
        if (k) {
          const int a = k * tws;
          u1 = cmul(u1, TW[a]); u2 = cmul(u2, TW[2 * a]); u3 = cmul(u3, TW[3 * a]); u4 = cmul(u4, TW[4 * a]);
        }
        const double c1 = 0.30901699437494742, c2_ = -0.80901699437494742;      // cos(2 pi / 5), cos(4 pi / 5)
        const double s1 = 0.95105651629515357, s2 = 0.58778525229247313;        // sin(2 pi / 5), sin(4 pi / 5)
        const c2 a1 = u1 + u4, a2 = u2 + u3, b1 = u1 - u4, b2 = u2 - u3;
        const c2 m1 = u0 + c1 * a1 + c2_ * a2, m2 = u0 + c2_ * a1 + c1 * a2;
        const c2 n1 = cmul_mi(s1 * b1 + s2 * b2), n2 = cmul_mi(s2 * b1 - s1 * b2);       // -i (...)
        yo[0] = u0 + a1 + a2; yo[Ns] = m1 + n1; yo[2 * Ns] = m2 + n2; yo[3 * Ns] = m2 - n2; yo[4 * Ns] = m1 - n1;
      } else {      // r == 2
        c2 u0 = xi[0], u1 = xi[t];
        if (k) u1 = cmul(u1, TW[k * tws]);
        yo[0] = u0 + u1; yo[Ns] = u0 - u1;
      }
    }
    __syncthreads();
    c2* tmp = x; x = y; y = tmp;
    Ns *= r;
  }
  // ---- split: amplitude of bin k of the real transform (real_amplitude: |X[k]| / n_fft) -> the other buffer
  double* amp = (double*)y;                           // [FFB][NBIN] doubles <= [FFB][M] complex
  for (int it = tid; it < FFB * NBIN; it += blockDim.x) {
    const int f = it / NBIN, k = it - f * NBIN;
    const c2* Z = x + (size_t)f * M;
    const c2 zk = Z[k == M ? 0 : k], zm = Z[k == 0 ? 0 : M - k];
    const c2 zc = c2{zm.x, -zm.y};
    const c2 xe = 0.5 * (zk + zc), dd = 0.5 * (zk - zc);
    const c2 xo = c2{dd.y, -dd.x};                    // (zk - conj zm) / (2 i)
    const c2 X = xe + cmul(TW[M + k], xo);
    amp[(size_t)f * NBIN + k] = sqrt(X.x * X.x + X.y * X.y) / (double)NF;
  }
  __syncthreads();
  double* melv = (double*)x;                          // [FFB][NM] squares of exp(log-mel): the spectrum buffer is free now
  const double amin = (double)d.min_clip / (double)NF;
  const double rng = -20.0 * log10(amin);
  for (int it = tid; it < FFB * NM; it += blockDim.x) {
    const int f = it / NM, m = it % NM;
    const double* av = amp + (size_t)f * NBIN;
    double sacc = 0.0;
    if (bof[m] >= 0) {
      const double* fv = fbs + bof[m] - blo[m];
      for (int k = blo[m]; k < bhi[m]; ++k) sacc = fma(fv[k], av[k], sacc);
    } else {
      const double* fv = fb + (long)m * NBIN;
      for (int k = blo[m]; k < bhi[m]; ++k) sacc = fma(fv[k], av[k], sacc);
    }
    sacc = fabs(sacc);
    if (sacc < amin) sacc = amin;
    const double yv = mel_logamp(sacc, rng, exact, d.flags);
    melv[it] = mel_exp_sq(yv, exact);                                 // (every thread its own exp; the frame's thread only adds, in mel order)
    if (slot0 + f < nfr) logmel[(slot0 + f) * NM + m] = yv;
  }
  __syncthreads();
  if (tid < FFB && slot0 + tid < nfr) {
    double e = 0.0;
    for (int m = 0; m < NM; ++m) e += melv[tid * NM + m];
    energy[slot0 + tid] = sqrt(e);
  }
}

// resample_method = "cubic": interp1d(kind = "cubic") = the interpolating cubic spline with not-a-knot ends (make_interp_spline(k = 3)) over
// the integer frame grid 0 .. M-1, one per column (80 mel channels, griddata's 1-D path, + the energy).  In terms of the second derivatives
// S_i:  S_{i-1} + 4 S_i + S_{i+1} = 6 (y_{i-1} - 2 y_i + y_{i+1}) =: r_i  (i = 1 .. M-2), and the not-a-knot conditions S_0 - 2 S_1 + S_2 = 0,
// S_{M-3} - 2 S_{M-2} + S_{M-1} = 0 turn the first and last equation into S_1 = r_1 / 6, S_{M-2} = r_{M-2} / 6.  What is left (i = 2 .. M-3) is a
// diagonally dominant tridiagonal system: one thread per column eliminates forward and substitutes back (the columns of a row are contiguous:
// coalesced).  float64 throughout, as scipy.
__global__ void mel_spline_rhs_k(ZeggsMelDims d, const double* logmel, const double* energy, long M, double* S) {
  const int W = d.n_mels + 1;
  const long n = M * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % W);
    const long m = i / W;
    auto y = [&](long q) { return c < d.n_mels ? logmel[q * d.n_mels + c] : energy[q]; };
    S[i] = (m >= 1 && m <= M - 2) ? 6.0 * ((y(m - 1) - y(m)) - (y(m) - y(m + 1))) : 0.0;
  }
}
// The system is strictly diagonally dominant: what a row feels of a row k places away decays like (2 - sqrt 3)^k = 0.268^k (1e-23 at
// k = 40).  So the elimination is cut into chunks of SPL_CH rows that start SPL_HL rows early (from an arbitrary state: forgotten long
// before the chunk's own rows) and run SPL_HL rows past their end before substituting back -- exact to the last bit of float64, one
// workgroup per chunk, one thread per column: 144 000 rows (30 minutes of audio) in ~0.1 ms instead of 45 ms as one sequential sweep.
constexpr int SPL_CH = 128, SPL_HL = 40;
__global__ __launch_bounds__(128) void mel_spline_solve_k(ZeggsMelDims d, long M, const double* R, double* S) {
  __shared__ double cpl[SPL_CH + 2 * SPL_HL];          // elimination coefficients of the rows this chunk walks (column-independent)
  __shared__ double halo[SPL_HL][128];                 // d' of the rows past the chunk's end
  const int W = d.n_mels + 1, c = threadIdx.x;
  const bool on = c < W;
  const long lo = 2, hi = M - 3;                       // interior unknowns
  const double s1 = on ? R[1 * W + c] / 6.0 : 0.0, sl = on ? R[(M - 2) * W + c] / 6.0 : 0.0;
  if (hi >= lo) {
    const long a = lo + (long)blockIdx.x * SPL_CH, b = (a + SPL_CH - 1 < hi) ? a + SPL_CH - 1 : hi;
    if (a <= hi) {
      const long st = (a - SPL_HL > lo) ? a - SPL_HL : lo, en = (b + SPL_HL < hi) ? b + SPL_HL : hi;
      double cprev = 0.0, dprev = 0.0;
      for (long i = st; i <= en; ++i) {
        double rhs = on ? R[i * W + c] : 0.0;
        if (i == lo) rhs -= s1;
        if (i == hi) rhs -= sl;
        const double den = (i == st) ? 4.0 : 4.0 - cprev;      // (i == st > lo: an arbitrary start, forgotten SPL_HL rows later)
        cprev = 1.0 / den;
        dprev = (rhs - ((i == st) ? 0.0 : dprev)) / den;
        if (c == 0) cpl[i - st] = cprev;
        if (on) {
          if (i >= a && i <= b) S[i * W + c] = dprev;
          else if (i > b) halo[i - b - 1][c] = dprev;
        }
      }
      __syncthreads();
      double x = 0.0;
      for (long i = en; i >= a; --i) {
        const double dp = (i > b) ? halo[i - b - 1][c] : (on ? S[i * W + c] : 0.0);
        x = (i == en) ? dp : dp - cpl[i - st] * x;            // (en < hi: as if the row behind were zero -- forgotten before row b)
        if (on && i <= b) S[i * W + c] = x;
      }
    }
  }
  // the two rows next to the ends and the ends themselves (not-a-knot), by the chunks that hold their neighbours
  if (on && blockIdx.x == 0) {
    const double s2 = M > 4 ? S[2 * W + c] : sl;
    S[1 * W + c] = s1;
    S[c] = 2.0 * s1 - s2;
  }
  const long nchunk = hi >= lo ? (hi - lo) / SPL_CH + 1 : 1;
  if (on && blockIdx.x == nchunk - 1) {
    const double sm3 = M > 4 ? S[(M - 3) * W + c] : s1;
    S[(M - 2) * W + c] = sl;
    S[(M - 1) * W + c] = 2.0 * sl - sm3;
  }
}

// resampling at t_k = ((fs/hop)/fps) k.  "linear" / "cubic": mel -> NaN outside the hull (griddata's fill value), the energy extrapolates
// (interp1d(fill_value = "extrapolate"): the end pieces continue); "nearest": both take the nearest frame, halves round DOWN, ends clamp
// (interp1d's bounds x_i + 1/2 searched from the left; griddata sets fill_value = "extrapolate" for this method).
// animation frames k0 .. k0 + n_frames - 1; logmel / energy hold the STFT frames m0 .. (indices relative to m0)
__global__ void mel_resample_k(ZeggsMelDims d, const double* logmel, const double* energy, const double* spl, long M, long m0, long k0,
                               int n_frames, float* out) {
  const int W = d.n_mels + 1;
  const long n = (long)n_frames * W;
  logmel -= m0 * d.n_mels;
  energy -= m0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % W);
    const long k = k0 + i / W;
    const double t = (((double)d.fs / (double)d.hop) / (double)d.fps) * (double)k;
    if (d.flags & MEL_NEAREST) {
      long q = (long)ceil(t - 0.5);
      q = q < 0 ? 0 : q > M - 1 ? M - 1 : q;
      out[i] = (float)(c < d.n_mels ? logmel[q * d.n_mels + c] : energy[q]);
      continue;
    }
    long hi = (long)ceil(t);            // searchsorted(side=left) over the integer grid
    if (hi < 1) hi = 1;
    if (hi > M - 1) hi = M - 1;
    const long lo = hi - 1;
    if (M < 2) { out[i] = nanf(""); continue; }
    if (c < d.n_mels && (t < 0.0 || t > (double)(M - 1))) { out[i] = nanf(""); continue; }
    const double ylo = c < d.n_mels ? logmel[lo * d.n_mels + c] : energy[lo], yhi = c < d.n_mels ? logmel[hi * d.n_mels + c] : energy[hi];
    if (d.flags & MEL_CUBIC) {
      const double u = t - (double)lo, v = 1.0 - u;
      out[i] = (float)(ylo * v + yhi * u + ((v * v * v - v) * spl[lo * W + c] + (u * u * u - u) * spl[hi * W + c]) / 6.0);
    } else {
      out[i] = (float)((yhi - ylo) * (t - (double)lo) + ylo);
    }
  }
}

}  // namespace

int g_mel_mfma = 1;      // zeggs_set_option("mel_mfma", 0/1): matrix-core DFT / one workgroup per frame, direct DFT (when the FFT form is off)
int g_mel_exact_log = 2;      // zeggs_set_option("mel_exact_log", 0 / 1 / 2): see mel_logamp (measured on 30 min of audio: 1.61 / 1.50 / 1.46 ms for
                              // modes 1 / 2 / 0 -- the transcendentals are NOT what bounds the kernel; mode 2 equals the literal chain bit for bit in float32)
int g_mel_fft = 1;       // zeggs_set_option("mel_fft", 0/1): the FFT form (default; n_fft / 2 must factor into 4, 5, 2)

// radices of the half-length transform: 4s first, then 5s, then a 2 (400 = 4 4 5 5); nst = 0: not this path
static FftPlan fft_plan(int M) {
  FftPlan p{};
  int m = M, n4 = 0, n5 = 0, n2 = 0;
  while (m % 4 == 0) { m /= 4; ++n4; }
  while (m % 5 == 0) { m /= 5; ++n5; }
  while (m % 2 == 0) { m /= 2; ++n2; }
  if (m != 1 || n4 + n5 + n2 > FMAXST || n4 + n5 + n2 == 0) return p;
  for (int i = 0; i < n4; ++i) p.radix[p.nst++] = 4;
  for (int i = 0; i < n5; ++i) p.radix[p.nst++] = 5;
  for (int i = 0; i < n2; ++i) p.radix[p.nst++] = 2;
  return p;
}

// STFT frames m0 .. m0 + nfr - 1 -> logmel / energy slots 0 .. nfr - 1
static int launch_stft(const ZeggsMelDims& d, const MelWs& w, const float* wav, long n, long n_avail, const double* fb, long m0,
                       long nfr, hipStream_t s) {
  const int NBIN = d.n_fft / 2 + 1;
  const FftPlan plan = fft_plan(d.n_fft / 2);
  const size_t fft_lds = mel_fft_lds(d.n_fft, d.n_mels);
  if (g_mel_fft && plan.nst > 0 && d.n_fft % 2 == 0 && fft_lds <= 160 * 1024 && (size_t)FFB * d.n_mels <= (size_t)2 * FFB * (d.n_fft / 2) &&
      nfr >= 1) {
    static bool fft_attr_set = false;
    if (!fft_attr_set) {
      if (hipFuncSetAttribute((const void*)mel_stft_fft_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        zeggs_set_error("mel: cannot raise the LDS limit of the FFT kernel");
        return -1;
      }
      fft_attr_set = true;
    }
    hipLaunchKernelGGL(mel_fft_table_k, dim3(16), dim3(256), 0, s, (c2*)w.ftw, d.n_fft / 2, fb, d.n_mels, w.bands, w.fbp);
    ZLAUNCH_CHECK("mel_fft_table");
    hipLaunchKernelGGL(mel_stft_fft_k, dim3((unsigned)((nfr + FFB - 1) / FFB)), dim3(FTHR), fft_lds, s, d, plan, wav, n, n_avail, fb,
                       (const c2*)w.ftw, w.bands, w.fbp, w.logmel, w.energy, m0, nfr, g_mel_exact_log);
    ZLAUNCH_CHECK("mel_stft_fft");
    return 0;
  }
  const size_t fast_lds = mel_fast_lds(d.n_fft, d.hop, d.n_mels);
  if (g_mel_mfma && d.pre_emph == 0.0 && d.n_fft % 4 == 0 && 2 * NBIN <= MCOLP && fast_lds <= 160 * 1024 && nfr >= 1) {   // (stages float32 samples)
    static bool attr_set = false;
    if (!attr_set) {       // more than 64 KB of dynamic LDS needs the opt-in
      (void)hipFuncSetAttribute((const void*)mel_stft_mfma_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_set = true;
    }
    hipLaunchKernelGGL(mel_table_k, dim3(1024), dim3(256), 0, s, w.table, w.win, d.n_fft, NBIN);
    ZLAUNCH_CHECK("mel_table");
    hipLaunchKernelGGL(mel_stft_mfma_k, dim3((unsigned)((nfr + MF - 1) / MF)), dim3(MWAVES * 64), fast_lds, s, d, wav, n, n_avail,
                       fb, w.table, w.win, w.logmel, w.energy, m0, nfr, g_mel_exact_log);
    ZLAUNCH_CHECK("mel_stft_mfma");
    return 0;
  }
  const size_t lds = sizeof(double) * (3 * (size_t)d.n_fft + d.n_fft / 2 + 1 + d.n_mels);
  hipLaunchKernelGGL(mel_stft_k, dim3((unsigned)nfr), dim3(256), lds, s, d, wav, n, n_avail, fb, w.logmel, w.energy, m0, g_mel_exact_log);
  ZLAUNCH_CHECK("mel_stft");
  return 0;
}

extern "C" long zeggs_mel_stft_frames(const ZeggsMelDims* d, long n_samples) {
  return stft_frames(n_samples, d->n_fft, d->hop, d->flags);
}

extern "C" size_t zeggs_mel_workspace_bytes(const ZeggsMelDims* d, long n_samples) {
  Arena a(nullptr, 0);
  carve_mel(*d, stft_frames(n_samples, d->n_fft, d->hop, d->flags), a);
  return a.off + 256;
}

extern "C" int zeggs_mel_features(const ZeggsMelDims* dp, const float* wav, long n_samples, const double* filterbank,
                                  int n_frames, float* out, void* ws, size_t ws_bytes, void* stream) {
  const ZeggsMelDims& d = *dp;
  hipStream_t s = (hipStream_t)stream;
  ZCHECK(d.n_fft >= 2 && d.n_fft % 2 == 0 && d.hop > 0 && d.n_mels > 0, "mel: bad dims");
  ZCHECK(n_samples > 0 && n_frames >= 0, "mel: empty input");
  const long M = stft_frames(n_samples, d.n_fft, d.hop, d.flags);
  Arena a(ws, ws_bytes);
  MelWs w = carve_mel(d, M, a);
  ZCHECK(a.ok(), "mel: workspace too small (%zu < %zu)", ws_bytes, a.off);
  ZTRY(launch_stft(d, w, wav, n_samples, n_samples, filterbank, 0L, M, s));
  if (n_frames > 0) {
    long n = (long)n_frames * (d.n_mels + 1), g = (n + 255) / 256;
    if (d.flags & MEL_CUBIC) {
      ZCHECK(M >= 4, "mel: resample_method \"cubic\" needs at least 4 STFT frames (got %ld)", M);      // (scipy raises the same way)
      ZCHECK(d.n_mels + 1 <= 128, "mel: resample_method \"cubic\" supports up to 127 mel channels");
      const long nn = M * (d.n_mels + 1), gg = (nn + 255) / 256;
      hipLaunchKernelGGL(mel_spline_rhs_k, dim3((unsigned)(gg > 4096 ? 4096 : gg)), dim3(256), 0, s, d, w.logmel, w.energy, M, w.cp);
      const long nchunk = M - 3 >= 2 ? (M - 5) / SPL_CH + 1 : 1;
      hipLaunchKernelGGL(mel_spline_solve_k, dim3((unsigned)nchunk), dim3(128), 0, s, d, M, w.cp, w.spl);
      ZLAUNCH_CHECK("mel_spline");
    }
    hipLaunchKernelGGL(mel_resample_k, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, s, d, w.logmel, w.energy, w.spl, M,
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
  // (uncentered: frame m covers samples m hop .. m hop + n_fft - 1)
  const long mmax = (d->flags & MEL_UNCENTERED) ? (n_samples - d->n_fft) / d->hop : (n_samples - d->n_fft / 2) / d->hop;      // last complete STFT frame (may be < 1)
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
  ZCHECK(!(d.flags & MEL_CUBIC), "mel range: resample_method \"cubic\" is a spline over the WHOLE signal -- use zeggs_mel_features");
  if (!final) ZCHECK(k1 <= zeggs_mel_frames_ready(dp, n_samples), "mel range: frames %ld..%ld need samples not received yet", k0, k1);
  const long M = final ? stft_frames(n_samples, d.n_fft, d.hop, d.flags) : (1L << 40);
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
  hipLaunchKernelGGL(mel_resample_k, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, s, d, w.logmel, w.energy, nullptr, M, m0, k0,
                     (int)(k1 - k0), out);
  ZLAUNCH_CHECK("mel_resample");
  return 0;
}
