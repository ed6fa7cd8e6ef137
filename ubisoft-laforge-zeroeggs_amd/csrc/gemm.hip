// Generic batched fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   C(m,n) = act( alpha * sum_k A(m,k) B(k,n) + beta * C(m,n) + bias[n] )
//
// A, B, C are addressed through element strides, so one kernel family covers
//   NT  (nn.Linear forward  y = x W^T)            A k-contig, B k-contig
//   NN  (input gradients    dx = dy W)            A k-contig, B n-contig
//   TN  (weight gradients   dW = dy^T x)          A m-contig, B n-contig
// plus two-level batching (attention heads), a batch-reduce K loop (weight
// gradients of batched convolutions) and overlapping-row A operands (a conv1d
// over a padded [T+2p, C] buffer is a GEMM with lda = C and K = taps*C).
// Tiles are staged global -> registers -> LDS (k-major, so MFMA operand reads are
// conflict-free ds_read_b32 of 32 consecutive floats) and the next tile's global
// loads are in flight while the current one feeds the MFMAs.
#include <type_traits>
#include "common.h"
#include <vector>

#include "gemm.h"
#include "kernels.h"

extern int g_gemm_split_bf16;                          // gemm_split.hip: the fp32-exact bf16 split (experiment, option "gemm_split_bf16", default off)
bool gemm_split_ok(const GemmArgs& g);
int launch_tn_split(GemmArgs g, hipStream_t s);

int g_gemm_streamk_wgs = 0;     // zeggs_set_option("gemm_streamk_wgs", n): stream-K workgroups per CU (0: as many as are resident)
int g_gemm_wg_target = 6144;   // split-K aims at this many workgroups (24 per CU = 6 rounds of 4 resident ones; sweep 1536..12288 in
                               // tools/gemm_bench.py: 3072 -> 6144 is 10-14 % on the weight-gradient shapes, flat beyond)

namespace {

#ifndef ZEGGS_GEMM_BK
#define ZEGGS_GEMM_BK 16
#endif
constexpr int BK = ZEGGS_GEMM_BK;      // k-extent of a tile (A/B loaders move BM * BK / threads floats per thread: 16 or 32 at 128 x 128)

template <int E>
__device__ __forceinline__ void load_contig(float (&r)[E], const float* p, int nvalid) {
  // p is only 4-byte aligned in general
  if (nvalid >= E) {
    if constexpr (E == 16) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f4u a = *(const f4u*)(p + 4 * q);
        r[4 * q] = a.x; r[4 * q + 1] = a.y; r[4 * q + 2] = a.z; r[4 * q + 3] = a.w;
      }
    } else if constexpr (E == 8) {
      f4u a = *(const f4u*)p, b = *(const f4u*)(p + 4);
      r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
    } else if constexpr (E == 4) {
      f4u a = *(const f4u*)p;
      r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
    } else if constexpr (E == 2) {
      f2u a = *(const f2u*)p;
      r[0] = a.x; r[1] = a.y;
    } else {
#pragma unroll
      for (int i = 0; i < E; ++i) r[i] = p[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < E; ++i) r[i] = (i < nvalid) ? p[i] : 0.f;
  }
}

template <int E>
__device__ __forceinline__ void load_full(float (&r)[E], const float* p) {
  if constexpr (E == 16) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f4u a = *(const f4u*)(p + 4 * q);
      r[4 * q] = a.x; r[4 * q + 1] = a.y; r[4 * q + 2] = a.z; r[4 * q + 3] = a.w;
    }
  } else if constexpr (E == 8) {
    f4u a = *(const f4u*)p, b = *(const f4u*)(p + 4);
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
  } else if constexpr (E == 4) {
    f4u a = *(const f4u*)p;
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
  } else if constexpr (E == 2) {
    f2u a = *(const f2u*)p;
    r[0] = a.x; r[1] = a.y;
  } else {
#pragma unroll
    for (int i = 0; i < E; ++i) r[i] = p[i];
  }
}

// four workgroups (waves per SIMD) resident per CU: the compiler then keeps the 64 accumulators + operands in 101 VGPRs
// instead of 78 + 64 AGPRs (3 per SIMD); 1-5 % on the training shapes
#ifndef ZEGGS_GEMM_MINB
#define ZEGGS_GEMM_MINB 4
#endif
#ifndef ZEGGS_GEMM_HALF_TILES
#define ZEGGS_GEMM_HALF_TILES 1
#endif
#ifndef ZEGGS_GEMM_SWP
#define ZEGGS_GEMM_SWP 0
#endif
#ifndef ZEGGS_GEMM_MIDSTORE
#define ZEGGS_GEMM_MIDSTORE 0      // (with ZEGGS_GEMM_SWP) k-pair in front of which the next tile is stored to LDS; 0: after the products
#endif
#ifndef ZEGGS_GEMM_ABL
#define ZEGGS_GEMM_ABL 0           // timing experiments (wrong results): 1 no per-k LDS fetch, 2 no loads / stores, 3 = 2 + no barrier, 4 = 1 + 3
#endif
#ifndef ZEGGS_GEMM_BUMP
#define ZEGGS_GEMM_BUMP 0      // carried source addresses: +3 .. 4 % on the GEMM alone, -0.6 % on the training iteration (three alternating A/B pairs)
#endif
#ifndef ZEGGS_GEMM_SWIZZLE
#define ZEGGS_GEMM_SWIZZLE 1
#endif
// One output tile (m_base, n_base) of batch entry (A, B, C) over the k-tiles [t_begin, ntiles) of its contraction (a tile has
// nkt * kbatch of them).  atomic: the result is a partial sum -> fp32 atomics onto C (split-K / stream-K segments).
template <int BM, int BN, int WM, int WN, bool AKC, bool BKC>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const float* A, const float* B, float* C, int m_base, int n_base,
                                          int t_begin, int ntiles, bool atomic, float* As2base, float* Bs2base) {
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MT = TM / 32, NTL = TN / 32;
  constexpr int EA = BM * BK / NT, EB = BN * BK / NT;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  static_assert(EA >= 1 && EB >= 1 && (EA == 1 || EA == 2 || EA == 4 || EA == 8 || EA == 16), "tile/threads");
  float (*As2)[BK * LDA] = (float (*)[BK * LDA])As2base;
  float (*Bs2)[BK * LDB] = (float (*)[BK * LDB])Bs2base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // loader coordinates
  int a_r, a_c, b_r, b_c;  // (row within tile along the non-contiguous dim, start along the contiguous dim)
  if constexpr (AKC) { a_r = tid / (BK / EA); a_c = (tid % (BK / EA)) * EA; }   // a_r = m, a_c = k0
  else               { a_r = tid / (BM / EA); a_c = (tid % (BM / EA)) * EA; }   // a_r = k, a_c = m0
  if constexpr (BKC) { b_r = tid / (BK / EB); b_c = (tid % (BK / EB)) * EB; }   // b_r = n, b_c = k0
  else               { b_r = tid / (BN / EB); b_c = (tid % (BN / EB)) * EB; }   // b_r = k, b_c = n0

  const int nkt = (g.K + BK - 1) / BK;
  float ra[EA], rb[EB];

  // interior blocks take a branch-free path (unguarded vector loads); edge blocks / the K tail use guarded loads
  const bool interior = (m_base + BM <= g.M) && (n_base + BN <= g.N);
#if ZEGGS_GEMM_BUMP
  // Tiles are loaded in order (t_begin, t_begin + 1, ...): this thread's source addresses advance by a constant per tile, so they
  // are carried instead of being rebuilt from the tile index (an integer division and four 64-bit multiply-adds, ~70 scalar
  // instructions in front of every tile's loads -- all co-resident workgroups reach them together, nobody issues products
  // meanwhile).  nx_*: state of the NEXT tile to be loaded.
  int nx_kb = t_begin / nkt, nx_k = (t_begin - nx_kb * nkt) * BK;
  const float* nx_a = A + (long)nx_kb * g.kbsA + (AKC ? (long)(m_base + a_r) * g.sam + (nx_k + a_c) : (long)(nx_k + a_r) * g.sak + (m_base + a_c));
  const float* nx_b = B + (long)nx_kb * g.kbsB + (BKC ? (long)(n_base + b_r) * g.sbn + (nx_k + b_c) : (long)(nx_k + b_r) * g.sbk + (n_base + b_c));
  const long a_step = AKC ? (long)BK : (long)BK * g.sak, b_step = BKC ? (long)BK : (long)BK * g.sbk;
  const long a_wrap = g.kbsA - (long)nkt * a_step, b_wrap = g.kbsB - (long)nkt * b_step;      // from the last k-tile of a segment to the next segment
#endif
  auto gload_r = [&](float (&ra)[EA], float (&rb)[EB], int tile) {
#if ZEGGS_GEMM_BUMP
    (void)tile;
    const int k_base = nx_k;
    const float* pa = nx_a;
    const float* pb = nx_b;
    nx_k += BK; nx_a += a_step; nx_b += b_step;
    if (nx_k >= nkt * BK) { nx_k = 0; nx_a += a_wrap; nx_b += b_wrap; }
    if (interior && k_base + BK <= g.K) {
      load_full<EA>(ra, pa);
      load_full<EB>(rb, pb);
      return;
    }
    const float* Ab = pa - (AKC ? (long)(m_base + a_r) * g.sam + (k_base + a_c) : (long)(k_base + a_r) * g.sak + (m_base + a_c));
    const float* Bb = pb - (BKC ? (long)(n_base + b_r) * g.sbn + (k_base + b_c) : (long)(k_base + b_r) * g.sbk + (n_base + b_c));
#else
    const int kb = tile / nkt, k_base = (tile % nkt) * BK;
    const float* Ab = A + (long)kb * g.kbsA;
    const float* Bb = B + (long)kb * g.kbsB;
    if (interior && k_base + BK <= g.K) {
      if constexpr (AKC) load_full<EA>(ra, Ab + (long)(m_base + a_r) * g.sam + (k_base + a_c));
      else load_full<EA>(ra, Ab + (long)(k_base + a_r) * g.sak + (m_base + a_c));
      if constexpr (BKC) load_full<EB>(rb, Bb + (long)(n_base + b_r) * g.sbn + (k_base + b_c));
      else load_full<EB>(rb, Bb + (long)(k_base + b_r) * g.sbk + (n_base + b_c));
      return;
    }
#endif
    if constexpr (AKC) {
      int m = m_base + a_r, k = k_base + a_c;
      int nv = (m < g.M) ? (g.K - k) : 0;
      load_contig<EA>(ra, Ab + (long)m * g.sam + k, nv < 0 ? 0 : nv);
    } else {
      int k = k_base + a_r, m = m_base + a_c;
      int nv = (k < g.K) ? (g.M - m) : 0;
      load_contig<EA>(ra, Ab + (long)k * g.sak + m, nv < 0 ? 0 : nv);
    }
    if constexpr (BKC) {
      int n = n_base + b_r, k = k_base + b_c;
      int nv = (n < g.N) ? (g.K - k) : 0;
      load_contig<EB>(rb, Bb + (long)n * g.sbn + k, nv < 0 ? 0 : nv);
    } else {
      int k = k_base + b_r, n = n_base + b_c;
      int nv = (k < g.K) ? (g.N - n) : 0;
      load_contig<EB>(rb, Bb + (long)k * g.sbk + n, nv < 0 ? 0 : nv);
    }
  };
  auto sstore_r = [&](const float (&ra)[EA], const float (&rb)[EB], int buf) {
    float* As = As2[buf];
    float* Bs = Bs2[buf];
    if constexpr (AKC) {
#pragma unroll
      for (int i = 0; i < EA; ++i) As[(a_c + i) * LDA + a_r] = ra[i];
    } else {
#pragma unroll
      for (int i = 0; i < EA; ++i) As[a_r * LDA + a_c + i] = ra[i];
    }
    if constexpr (BKC) {
#pragma unroll
      for (int i = 0; i < EB; ++i) Bs[(b_c + i) * LDB + b_r] = rb[i];
    } else {
#pragma unroll
      for (int i = 0; i < EB; ++i) Bs[b_r * LDB + b_c + i] = rb[i];
    }
  };

  auto gload = [&](int tile) { gload_r(ra, rb, tile); };
  auto sstore = [&](int buf) { sstore_r(ra, rb, buf); };

  f16v acc[MT][NTL];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int kh = lane >> 5, l31 = lane & 31;
  if (t_begin < ntiles) {
    gload(t_begin);
    sstore(0);
  }
  __syncthreads();
  for (int t = t_begin; t < ntiles; ++t) {
    const int cur = (t - t_begin) & 1;
    const float* As = As2[cur];
    const float* Bs = Bs2[cur];
#if ZEGGS_GEMM_ABL == 2 || ZEGGS_GEMM_ABL == 3 || ZEGGS_GEMM_ABL == 4 || ZEGGS_GEMM_ABL == 6   // (timing experiment: no global loads / LDS stores after the first tile)
    if (t + 1 < ntiles && t == t_begin) gload(t + 1);
#else
    if (t + 1 < ntiles) gload(t + 1);
#endif
#if ZEGGS_GEMM_SWP
    // operand fetch of k-pair kp + 1 in flight under the products of k-pair kp
    float a[2][MT], b[2][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i) a[0][i] = As[kh * LDA + wm * TM + i * 32 + l31];
#pragma unroll
    for (int j = 0; j < NTL; ++j) b[0][j] = Bs[kh * LDB + wn * TN + j * 32 + l31];
#pragma unroll
    for (int kp = 0; kp < BK / 2; ++kp) {
#if ZEGGS_GEMM_MIDSTORE > 0
      // the next tile goes to the other LDS buffer in the MIDDLE of this tile's products (its global loads were issued at the
      // top; the buffer's last readers passed the barrier at the end of the previous tile): the end of a tile is then only
      // the barrier, not drain + store + barrier -- which all co-resident workgroups reach together (they share the matrix
      // pipe round-robin and run in lockstep), so nobody covers it
      if (kp == ZEGGS_GEMM_MIDSTORE && t + 1 < ntiles) { sstore(cur ^ 1); __builtin_amdgcn_sched_barrier(0); }
#endif
      if (kp + 1 < BK / 2) {
#pragma unroll
        for (int i = 0; i < MT; ++i) a[(kp + 1) & 1][i] = As[(2 * kp + 2 + kh) * LDA + wm * TM + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < NTL; ++j) b[(kp + 1) & 1][j] = Bs[(2 * kp + 2 + kh) * LDB + wn * TN + j * 32 + l31];
      }
      __builtin_amdgcn_sched_barrier(0);      // (left alone, the compiler sinks the fetch back next to its first use)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kp & 1][i], b[kp & 1][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#else
#pragma unroll
    for (int kp = 0; kp < BK / 2; ++kp) {
      float a[MT], b[NTL];
#if ZEGGS_GEMM_ABL == 1 || ZEGGS_GEMM_ABL == 4      // (timing experiment, results wrong: operands fetched once per tile)
      const int kq = 0;
#else
      const int kq = kp;
#endif
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = As[(2 * kq + kh) * LDA + wm * TM + i * 32 + l31];
#pragma unroll
      for (int j = 0; j < NTL; ++j) b[j] = Bs[(2 * kq + kh) * LDB + wn * TN + j * 32 + l31];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
#endif
#if ZEGGS_GEMM_ABL == 2 || ZEGGS_GEMM_ABL == 3 || ZEGGS_GEMM_ABL == 4
    if (t + 1 < ntiles && t == t_begin) { sstore(0); sstore(1); }
#elif ZEGGS_GEMM_ABL == 5      // loads kept (and waited for), no LDS stores after the first tile
    if (t + 1 < ntiles) {
      if (t == t_begin) { sstore(0); sstore(1); }
#pragma unroll
      for (int i = 0; i < EA; ++i) asm volatile("" ::"v"(ra[i]));
#pragma unroll
      for (int i = 0; i < EB; ++i) asm volatile("" ::"v"(rb[i]));
    }
#elif !(ZEGGS_GEMM_SWP && ZEGGS_GEMM_MIDSTORE > 0)
    if (t + 1 < ntiles) sstore(cur ^ 1);
#endif
#if ZEGGS_GEMM_ABL == 3 || ZEGGS_GEMM_ABL == 4
    if (t == t_begin) __syncthreads();
#else
    __syncthreads();
#endif
  }

  // epilogue: D layout of 32x32x2: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
      const int n = n_base + wn * TN + j * 32 + l31;
      if (n >= g.N) continue;
      const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m_base + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (m >= g.M) continue;
        float* cp = C + (long)m * g.scm + (long)n * g.scn;
        if (atomic) { atomicAdd(cp, g.alpha * acc[i][j][r]); continue; }
        float v = g.alpha * acc[i][j][r] + bv;
        if (g.beta != 0.f) v += g.beta * (*cp);
        *cp = d_act(v, g.act);
      }
    }
}

template <int BM, int BN, int WM, int WN, bool AKC, bool BKC>
__global__ __launch_bounds__(WM* WN * 64, ZEGGS_GEMM_MINB) void gemm_kernel(GemmArgs g) {
  constexpr int LDA = BM + 4, LDB = BN + 4;
  __shared__ __attribute__((aligned(16))) float As2[2][BK * LDA];   // double-buffered: one barrier per k-tile
  __shared__ __attribute__((aligned(16))) float Bs2[2][BK * LDB];
  // Tile order.  Workgroups go to the 8 XCDs round-robin in launch order, and every XCD has its own L2: XCD k takes the k-th
  // CONTIGUOUS eighth of the tile list instead of every eighth tile, and within a k-split the list walks groups of GEMM_GM tile
  // rows column by column, so the tiles in flight on an XCD share a few A row panels and B column panels.
  int bx = blockIdx.x, by = blockIdx.y, zz = blockIdx.z;
  if (ZEGGS_GEMM_SWIZZLE) {
    const unsigned gx = gridDim.x, gy = gridDim.y, plane = gx * gy, total = plane * gridDim.z;
    const unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned xcd = L & 7, idx = L >> 3, q = total >> 3, r = total & 7;
    const unsigned Lp = xcd * q + (xcd < r ? xcd : r) + idx;
    const unsigned pid = Lp % plane;
    zz = (int)(Lp / plane);
    constexpr unsigned GM = 4;
    const unsigned width = GM * gx, group = pid / width, first = group * GM, gsz = gy - first < GM ? gy - first : GM;
    by = (int)(first + (pid % width) % gsz);
    bx = (int)((pid % width) / gsz);
  }
  const int m_base = by * BM, n_base = bx * BN;
  const int z = zz / g.splitk, ks = zz % g.splitk;
  const long z0 = z / g.nb1, z1 = z % g.nb1;
  const float* A = g.A + z0 * g.bsA0 + z1 * g.bsA1;
  const float* B = g.B + z0 * g.bsB0 + z1 * g.bsB1;
  float* C = g.C + z0 * g.bsC0 + z1 * g.bsC1;

  const int ntiles_all = ((g.K + BK - 1) / BK) * g.kbatch;
  const int t_begin = (int)((long)ntiles_all * ks / g.splitk), ntiles = (int)((long)ntiles_all * (ks + 1) / g.splitk);
  gemm_tile<BM, BN, WM, WN, AKC, BKC>(g, A, B, C, m_base, n_base, t_begin, ntiles, g.splitk > 1, &As2[0][0], &Bs2[0][0]);
}

// Stream-K form of the split contraction (weight gradients: few output tiles, K = B (T-1) ~ 8000): the (tile, k-tile) iteration
// space is cut into gridDim.x EQUAL contiguous ranges, one per workgroup, all resident at once (4 per CU) -- every CU gets the
// same number of matrix-core passes whatever the tile count, and a workgroup runs ~100 k-tiles per epilogue instead of the ~16
// of the 6144-workgroup split (its pipeline fill, its 16 K atomics per tile and the zero-fill of C amortise over six times the
// work).  A range that crosses a tile boundary finishes the first tile's share and starts the next: at most two epilogues,
// always atomics onto the zeroed C.  g.splitk carries the number of k-tiles per output tile here.
template <int BM, int BN, int WM, int WN, bool AKC, bool BKC>
__global__ __launch_bounds__(WM* WN * 64, (WM * WN > 4 ? 2 : ZEGGS_GEMM_MINB)) void gemm_streamk_kernel(GemmArgs g, int tiles_x, int tiles_y) {
  constexpr int LDA = BM + 4, LDB = BN + 4;
  __shared__ __attribute__((aligned(16))) float As2[2][BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs2[2][BK * LDB];
  const int kt = g.splitk;                                     // k-tiles of one output tile
  const long total = (long)tiles_x * tiles_y * kt;
  // XCD k (workgroups k, k + 8, ...) takes the k-th contiguous eighth of the ranges: neighbouring tiles share its L2
  const unsigned nwg = gridDim.x, w = blockIdx.x, xcd = w & 7, idx = w >> 3, q = nwg >> 3, r = nwg & 7;
  const unsigned wl = xcd * q + (xcd < r ? xcd : r) + idx;
  long it = total * wl / nwg;
  const long it_end = total * (wl + 1) / nwg;
  while (it < it_end) {
    const int tile = (int)(it / kt), k0 = (int)(it % kt);
    const long left = it_end - it;
    const int k1 = (long)(kt - k0) < left ? kt : k0 + (int)left;
    // tiles walk groups of 4 tile rows column by column (as the plain kernel's XCD-aware order does)
    constexpr int GM = 4;
    const int width = GM * tiles_x, group = tile / width, first = group * GM, gsz = tiles_y - first < GM ? tiles_y - first : GM;
    const int by = first + (tile % width) % gsz, bx = (tile % width) / gsz;
    gemm_tile<BM, BN, WM, WN, AKC, BKC>(g, g.A, g.B, g.C, by * BM, bx * BN, k0, k1, true, &As2[0][0], &Bs2[0][0]);
    __syncthreads();                                           // the LDS buffers start over
    it += k1 - k0;
  }
}

// skinny-M helpers: zero / finish a strided [M, N] row block (split-K accumulates with atomics, bias + activation follow)
__global__ void rows_fill_k(float* C, int M, int N, long ld) {
  long n = (long)M * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    C[(i / N) * ld + (i % N)] = 0.f;
}
__global__ void rows_bias_act_k(float* C, const float* bias, int M, int N, long ld, int act) {
  long n = (long)M * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float* c = C + (i / N) * ld + (i % N);
    *c = d_act(*c + (bias ? bias[i % N] : 0.f), act);
  }
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(const GemmArgs& g, int nbatch, hipStream_t s) {
  dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM), nbatch * g.splitk), block(WM * WN * 64);
  const bool akc = (g.sak == 1), bkc = (g.sbk == 1) && (g.sbn != 1 || g.N == 1);
  if (!akc && g.sam != 1) { zeggs_set_error("gemm: A has no unit stride (sam=%ld sak=%ld)", g.sam, g.sak); return -1; }
  if (!bkc && g.sbn != 1) { zeggs_set_error("gemm: B has no unit stride (sbk=%ld sbn=%ld)", g.sbk, g.sbn); return -1; }
  if (akc && bkc) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, true, true>), grid, block, 0, s, g);
  else if (akc && !bkc) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, true, false>), grid, block, 0, s, g);
  else if (!akc && bkc) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, false, true>), grid, block, 0, s, g);
  else hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, false, false>), grid, block, 0, s, g);
  ZLAUNCH_CHECK("gemm");
  return 0;
}

#ifndef ZEGGS_GEMM_DMA_DEFAULT
#define ZEGGS_GEMM_DMA_DEFAULT 0
#endif
// ---------------------------------------------------------------------------------------------------------------------------
// Stream-K weight-gradient product (TN: A(m,k) = dy[k][m], B(k,n) = x[k][n], both rows contiguous) with the tiles brought in by
// LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane straight into LDS, no staging registers, no ds_write pass) through a ring
// of THREE LDS buffers, so that two tiles are in flight while one is multiplied.  Measured on the register-staged kernel above
// (tools/gemm_probe.py, ablation builds): without its global loads the matrix pipe is busy 0.85 instead of 0.75 of the time,
// without the LDS store pass another 0.04 -- with one tile of prefetch distance and four co-resident workgroups that share the
// pipe oldest-first, a wave reaches its store pass ~1 us after it issued the loads, sooner than the fabric answers.
//   * a wave moves rows 4w .. 4w+3 of the [16 x 128] A and B tiles: two instructions per operand (lanes 0-31 one row, 32-63 the
//     next), the LDS image is the plain [k][128] row-major tile (the operand fetch reads 32 consecutive floats: conflict-free
//     without padding, which LDS-DMA could not produce anyway -- its destination is base + lane * 16);
//   * per tile: s_waitcnt vmcnt(4) (this wave's share of tile t has landed, tile t+1's four DMAs stay in flight), ONE raw
//     s_barrier (everybody's share has; everybody is done reading tile t-1), DMA of tile t+2 into the buffer tile t-1 used,
//     products of tile t;
//   * edge tiles: the source column of a lane is clamped into the row (duplicates feed output rows / columns >= M / N, which the
//     epilogue does not store); K must be a multiple of 16 and the rows 16-byte aligned (launch_gemm checks).
constexpr int DK = 16;
__device__ __forceinline__ unsigned lds_off(const float* p) { return (unsigned)(size_t)p; }     // low half of a flat LDS address
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ __launch_bounds__(256, 3) void gemm_tn_dma_kernel(GemmArgs g, int tiles_x, int tiles_y) {
  constexpr int BM = 128, BN = 128, TS = DK * 128;
  __shared__ __attribute__((aligned(16))) float As3[3 * TS];
  __shared__ __attribute__((aligned(16))) float Bs3[3 * TS];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const int kh = lane >> 5, l31 = lane & 31;
  const int kt = g.splitk, nkt = g.K / DK;
  const long total = (long)tiles_x * tiles_y * kt;
  const unsigned nwg = gridDim.x, w = blockIdx.x, xcd = w & 7, idx = w >> 3, q = nwg >> 3, r = nwg & 7;
  const unsigned wl = xcd * q + (xcd < r ? xcd : r) + idx;
  long it = total * wl / nwg;
  const long it_end = total * (wl + 1) / nwg;
  const int mmax = (g.M - 1) & ~3, nmax = (g.N - 1) & ~3;
  const unsigned la0 = lds_off(As3 + 4 * wave * 128), lb0 = lds_off(Bs3 + 4 * wave * 128);
  while (it < it_end) {
    const int tile = (int)(it / kt), t_begin = (int)(it % kt);
    const long left = it_end - it;
    const int ntiles = (long)(kt - t_begin) < left ? kt : t_begin + (int)left;
    constexpr int GM = 4;
    const int width = GM * tiles_x, group = tile / width, first = group * GM, gsz = tiles_y - first < GM ? tiles_y - first : GM;
    const int m_base = (first + (tile % width) % gsz) * BM, n_base = ((tile % width) / gsz) * BN;
    int mcol = m_base + 4 * l31, ncol = n_base + 4 * l31;
    mcol = mcol < mmax ? mcol : mmax;
    ncol = ncol < nmax ? ncol : nmax;
    const float* Al = g.A + (long)(4 * wave + kh) * g.sak + mcol;
    const float* Bl = g.B + (long)(4 * wave + kh) * g.sbk + ncol;
    auto issue = [&](int t, int buf) {
      const int kb = t / nkt, k_base = (t - kb * nkt) * DK;
      const float* ap = Al + (long)kb * g.kbsA + (long)k_base * g.sak;
      const float* bp = Bl + (long)kb * g.kbsB + (long)k_base * g.sbk;
      const unsigned la = la0 + buf * (TS * 4), lb = lb0 + buf * (TS * 4);
      glds16(ap, la);
      glds16(ap + 2 * g.sak, la + 2 * 128 * 4);
      glds16(bp, lb);
      glds16(bp + 2 * g.sbk, lb + 2 * 128 * 4);
    };
    f16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    issue(t_begin, 0);
    if (t_begin + 1 < ntiles) issue(t_begin + 1, 1);
    int buf = 0;
    for (int t = t_begin; t < ntiles; ++t) {
      if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (t + 2 < ntiles) issue(t + 2, buf >= 1 ? buf - 1 : 2);       // (buf + 2) % 3: the buffer tile t - 1 was read from
      const float* As = As3 + buf * TS;
      const float* Bs = Bs3 + buf * TS;
#pragma unroll
      for (int kp = 0; kp < DK / 2; ++kp) {
        float a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = As[(2 * kp + kh) * 128 + wm * 64 + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = Bs[(2 * kp + kh) * 128 + wn * 64 + j * 32 + l31];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      buf = buf == 2 ? 0 : buf + 1;
    }
    // partial sums -> fp32 atomics onto C (D layout of 32x32x2: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5))
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n_base + wn * 64 + j * 32 + l31;
        if (n >= g.N) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m_base + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
          if (m < g.M) atomicAdd(g.C + (long)m * g.scm + n, g.alpha * acc[i][j][e]);
        }
      }
    __syncthreads();                                           // the LDS ring starts over
    it += ntiles - t_begin;
  }
}

int g_gemm_dma = ZEGGS_GEMM_DMA_DEFAULT;      // zeggs_set_option("gemm_dma", 0/1)
bool dma_ok(const GemmArgs& g) {
  auto a16 = [](const void* p) { return ((size_t)p & 15) == 0; };
  return g_gemm_dma && g.sam == 1 && g.sbn == 1 && g.scn == 1 && g.K % DK == 0 && g.sak % 4 == 0 && g.sbk % 4 == 0 &&
         g.kbsA % 4 == 0 && g.kbsB % 4 == 0 && a16(g.A) && a16(g.B) && g.M >= 4 && g.N >= 4 &&
         (g.M % 4 == 0 || g.sak >= ((g.M + 3) & ~3)) && (g.N % 4 == 0 || g.sbk >= ((g.N + 3) & ~3));
}
int launch_tn_dma(GemmArgs g, hipStream_t s) {
  const int tx = cdiv(g.N, 128), ty = cdiv(g.M, 128);
  const int kt = (g.K / DK) * g.kbatch;
  g.splitk = kt;
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  long nwg = (long)ncu * 3;
  const long total = (long)tx * ty * kt;
  if (nwg > total / 8) nwg = total / 8 > 0 ? total / 8 : 1;
  hipLaunchKernelGGL(gemm_tn_dma_kernel, dim3((unsigned)nwg), dim3(256), 0, s, g, tx, ty);
  ZLAUNCH_CHECK("gemm_tn_dma");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 5: stream-K TN product WITHOUT LDS and WITHOUT barriers ("direct" form).  v_mfma_f32_32x32x2_f32 takes 64 cycles and one
// float per lane and operand, and in a TN product (weight gradients: A(m,k) = dy[k][m], B(k,n) = x[k][n], both rows contiguous)
// that float is row 2 kp + (lane >> 5), column tile + (lane & 31) of the operand as it lies in memory: ONE coalesced
// global_load_dword per operand fragment (two full 128-byte segments per wave-instruction), no transposition, no staging.  At the
// fp32 matrix rate a CU consumes ~16 bytes per clock of such loads (0.75 loads per product with 128 x 64 wave tiles) -- a quarter
// of what its L1 delivers -- so the operands can be fed from L1 / L2 directly, a few k-pairs ahead in registers, and the waves
// of a workgroup never wait for each other: no s_barrier per k-tile (the LDS-tiled kernel above loses 0.22-0.37 of the matrix-pipe
// cycles with the four co-resident workgroups of a CU in lockstep around their barriers: profiles/r04_gemm_mfma_util.json), no
// LDS write pass, no alignment conditions (4-byte loads).  The 2 x 2 waves of a workgroup still share a (2 * 32 MT) x (2 * 32 NT) tile so
// that the wave pairs that read the same A / B fragments hit the CU's L1.  Stream-K as above: equal contiguous ranges of the
// (tile, k-chunk) space per workgroup, partial tiles combined by fp32 atomics onto the zeroed / accumulating C.  K even.
template <int MT, int NT, int D>
__device__ __forceinline__ void tn_direct_body(const GemmArgs& g, int tiles_x, int tiles_y, int chunks_per_batch) {
  constexpr int CH = 8;                                        // k-pairs per chunk of the stream-K iteration space
  constexpr int BM = 64 * MT, BN = 64 * NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const int kh = lane >> 5, l31 = lane & 31;
  const int kt = chunks_per_batch * g.kbatch;                  // chunks of one output tile
  const long total = (long)tiles_x * tiles_y * kt;
  const unsigned nwg = gridDim.x, w = blockIdx.x, xcd = w & 7, idx = w >> 3, q = nwg >> 3, r = nwg & 7;
  const unsigned wl = xcd * q + (xcd < r ? xcd : r) + idx;
  long it = total * wl / nwg;
  const long it_end = total * (wl + 1) / nwg;
  const int npairs = g.K >> 1;                                 // per batch segment
  while (it < it_end) {
    const int tile = (int)(it / kt), c0 = (int)(it % kt);
    const long left = it_end - it;
    const int c1 = (long)(kt - c0) < left ? kt : c0 + (int)left;
    constexpr int GM = 4;
    const int width = GM * tiles_x, group = tile / width, first = group * GM, gsz = tiles_y - first < GM ? tiles_y - first : GM;
    const int m_base = (first + (tile % width) % gsz) * BM + wm * 32 * MT, n_base = ((tile % width) / gsz) * BN + wn * 32 * NT;
    int mo[MT], no[NT];                                        // this lane's column of every fragment (clamped: duplicates are not stored)
#pragma unroll
    for (int i = 0; i < MT; ++i) { const int m = m_base + 32 * i + l31; mo[i] = m < g.M ? m : g.M - 1; }
#pragma unroll
    for (int j = 0; j < NT; ++j) { const int n = n_base + 32 * j + l31; no[j] = n < g.N ? n : g.N - 1; }
    unsigned voa[MT], vob[NT];                                // byte offsets of this lane's operand entries within a k-pair's two rows
#pragma unroll
    for (int i = 0; i < MT; ++i) voa[i] = (unsigned)(((long)mo[i] + (long)kh * g.sak) * 4);
#pragma unroll
    for (int j = 0; j < NT; ++j) vob[j] = (unsigned)(((long)no[j] + (long)kh * g.sbk) * 4);
    f16v acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // GemmArgs.asum: the A fragments pass through this wave's registers anyway -- their sums over k are the bias gradient of the
    // layer whose weight gradient this product is.  Kept by every wave (MT adds per 2 MT NT matrix instructions), used by the
    // waves of the first column of output tiles only: each (row block, k chunk) is then counted exactly once.
    float sa[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) sa[i] = 0.f;
    // the chunk range [c0, c1) batch segment by batch segment
    int c = c0;
    while (c < c1) {
      const int kb = c / chunks_per_batch, cb = c - kb * chunks_per_batch;
      int ce = c1 - kb * chunks_per_batch;
      ce = ce < chunks_per_batch ? ce : chunks_per_batch;
      const int p0 = cb * CH, p1 = ce * CH < npairs ? ce * CH : npairs, n = p1 - p0;
      // operand addresses = wave-uniform row base (SGPR pair, advanced per pair) + this lane's 32-bit byte offset (column, k-half):
      // one VGPR per fragment instead of a 64-bit pointer each
      const float* Ab = g.A + (long)kb * g.kbsA + (long)(2 * p0) * g.sak;
      const float* Bb = g.B + (long)kb * g.kbsB + (long)(2 * p0) * g.sbk;
      const long a2 = 2 * g.sak, b2 = 2 * g.sbk;
      float ra[D][MT], rb[D][NT];
      // The refills are inline-asm loads: the compiler's own wait insertion cannot count loads across the loop's back edge (it
      // emits vmcnt(0) at the top of every iteration: the whole refill latency in the open per D pairs); these it does not see,
      // and the waits are placed by hand: at the products of slot u the D - 1 younger refill groups stay in flight
      // (cdna_hip_programming.md section 5.7, form (ii): "=v" loads, then a wait statement that names every destination "+v").
      auto fetch = [&](int slot, int pair) {
        const unsigned long long ap = (unsigned long long)(Ab + (long)pair * a2), bp = (unsigned long long)(Bb + (long)pair * b2);
        const unsigned alo = __builtin_amdgcn_readfirstlane((unsigned)ap), ahi = __builtin_amdgcn_readfirstlane((unsigned)(ap >> 32));
        const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)bp), bhi = __builtin_amdgcn_readfirstlane((unsigned)(bp >> 32));
        const unsigned long long as = ((unsigned long long)ahi << 32) | alo, bs = ((unsigned long long)bhi << 32) | blo;
#pragma unroll
        for (int i = 0; i < MT; ++i) asm volatile("global_load_dword %0, %1, %2" : "=v"(ra[slot][i]) : "v"(voa[i]), "s"(as) : "memory");
#pragma unroll
        for (int j = 0; j < NT; ++j) asm volatile("global_load_dword %0, %1, %2" : "=v"(rb[slot][j]) : "v"(vob[j]), "s"(bs) : "memory");
      };
      // the loads of slot S have landed; Y (a constant) later ones may still fly
// (GemmArgs.asum: the row sums of A are taken INSIDE the wait statement -- a compiler-level add on the asm loads' destinations
//  lengthens their live ranges, and the register allocator then copies them between the load and the wait, i.e. before the data
//  has arrived: measured, wrong results at K = 8160)
#define ZG_LANDED(S, Y)                                                                                                       \
  do {                                                                                                                         \
    if constexpr (MT == 4)                                                                                                     \
      asm volatile("s_waitcnt vmcnt(%10)\n\tv_add_f32 %6, %6, %0\n\tv_add_f32 %7, %7, %1\n\tv_add_f32 %8, %8, %2\n\tv_add_f32 %9, %9, %3" \
                   : "+v"(ra[S][0]), "+v"(ra[S][1]), "+v"(ra[S][MT > 2 ? 2 : 0]), "+v"(ra[S][MT > 3 ? 3 : 1]),              \
                     "+v"(rb[S][0]), "+v"(rb[S][1]), "+v"(sa[0]), "+v"(sa[1]), "+v"(sa[MT > 2 ? 2 : 0]), "+v"(sa[MT > 3 ? 3 : 1]) \
                   : "n"(Y) : "memory");                                                                                       \
    else                                                                                                                       \
      asm volatile("s_waitcnt vmcnt(%6)\n\tv_add_f32 %4, %4, %0\n\tv_add_f32 %5, %5, %1"                                       \
                   : "+v"(ra[S][0]), "+v"(ra[S][1]), "+v"(rb[S][0]), "+v"(rb[S][1]), "+v"(sa[0]), "+v"(sa[1]) : "n"(Y) : "memory"); \
  } while (0)
      static_assert((MT == 4 || MT == 2) && NT == 2, "direct kernel: wave tiles 128 x 64 and 64 x 64");
      constexpr int LG = MT + NT;                                // loads per refill group
      auto mma = [&](int slot) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[slot][i], rb[slot][j], acc[i][j], 0, 0, 0);
      };
#pragma unroll
      for (int u = 0; u < D; ++u)
        if (u < n) fetch(u, u);
      int p = 0;
      for (; p + 2 * D <= n; p += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
          ZG_LANDED(u, (D - 1) * LG);
          mma(u);
          __builtin_amdgcn_sched_barrier(0);      // (keeps the refill behind its slot's products and ahead of the next slot's)
          fetch(u, p + u + D);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // the last < 2 D pairs: everything in flight lands, one guarded refill round, the rest
#pragma unroll
      for (int u = 0; u < D; ++u)
        if (p + u < n) ZG_LANDED(u, 0);       // (only slots that hold a pair: the statement also adds the slot's A fragments to the row sums)
#pragma unroll
      for (int u = 0; u < D; ++u)
        if (p + u < n) {
          mma(u);
          if (p + u + D < n) { fetch(u, p + u + D); ZG_LANDED(u, 0); }
        }
      p += D;
#pragma unroll
      for (int u = 0; u < D; ++u)
        if (p + u < n) mma(u);
#undef ZG_LANDED
      c = (kb + 1) * chunks_per_batch < c1 ? (kb + 1) * chunks_per_batch : c1;
    }
    // partial sums -> fp32 atomics onto C (D layout of 32x32x2: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5))
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int nn = n_base + 32 * j + l31;
        if (nn >= g.N) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m_base + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * kh;
          if (m < g.M) atomicAdd(g.C + (long)m * g.scm + nn, g.alpha * acc[i][j][e]);
        }
      }
    if (g.asum != nullptr && (tile % width) / gsz == 0 && wn == 0) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const float v = sa[i] + __shfl_xor(sa[i], 32, 64);      // the two k-halves of row m
        const int m = m_base + 32 * i + l31;
        if (kh == 0 && m < g.M) atomicAdd(g.asum + m, g.alpha * v);
      }
    }
    it += c1 - c0;
  }
}
template <int MT, int NT, int D>
__global__ __launch_bounds__(256, 2) void gemm_tn_direct_kernel(GemmArgs g, int tiles_x, int tiles_y, int chunks_per_batch) {
  tn_direct_body<MT, NT, D>(g, tiles_x, tiles_y, chunks_per_batch);
}
// The same with the WHOLE register file of its SIMDs to itself (option "gemm_direct_shield"): one workgroup per CU, one wave per
// SIMD, 512 registers allocated per wave (the clobbers below; the body needs ~200), so that no wave of another queue becomes
// resident beside it.  The LDS-tiled kernel is shielded like that by its own footprint (2 x 256 registers per SIMD, 135 KB of LDS),
// which is why it beats this kernel inside the three-queue tail of a training iteration and loses to it alone on the chip.
template <int MT, int NT, int D>
__global__ __launch_bounds__(256, 1) void gemm_tn_direct_shield_kernel(GemmArgs g, int tiles_x, int tiles_y, int chunks_per_batch) {
  asm volatile("" ::: "v255", "a255");
  tn_direct_body<MT, NT, D>(g, tiles_x, tiles_y, chunks_per_batch);
}
// NOTE on the operand roles above: the matrix instruction computes D[m][n] += A[m][k] B[k][n] with lane = (k-half, m) for A and
// (k-half, n) for B, and D's lane index is the COLUMN n -- the A fragment's lane index is its row.  A(m, k) = A[k * sak + m]
// is what `ap[mo[i]]` reads for lane (kh, l31): row 2 pair + kh, column m.
int g_gemm_direct = 1;         // zeggs_set_option("gemm_direct", v): 0 off (the LDS-tiled stream-K kernel), 1 on (wave tile by output size),
                               // 2 / 3: always the 128 x 64 / the 64 x 64 wave tile (A/B), 5: only the small / batch-reduce products (what
                               // zeggs.engine.TrainEngine sets when it runs its three-stream schedule: see direct_ok)
int g_gemm_direct_depth = 4;   // zeggs_set_option("gemm_direct_depth", 4 / 6 / 8): k-pairs of operands in flight per wave
int g_gemm_direct_shield = 0;  // zeggs_set_option("gemm_direct_shield", 0 / 1 / 2): the variant that owns its SIMDs' register files (2: big products only)
int g_gemm_asum = 1;           // zeggs_set_option("gemm_asum", 0/1): bias column sums inside the weight-gradient product (A/B)
int g_gemm_direct_reserve = 0; // zeggs_set_option("gemm_direct_reserve", n): CUs the shield variant's grid leaves out
int g_gemm_direct_wgs = 0;     // zeggs_set_option("gemm_direct_wgs", n): workgroups per CU (0: 1 for the 128 x 64 wave tile, 2 for 64 x 64)
// Routing of the TN products is per CALLING THREAD (zeggs_gemm_route: what an engine wants for ITS launches travels with its calls,
// a second engine or a plain caller in the same process keeps the process-wide options above); -1 = the process-wide option.
static thread_local int tl_route[4] = {-1, -1, -1, -1};
static bool g_direct_failed = false;       // a variant's first-use check failed: off for the process, whatever a route says
static inline int rt_direct() { return g_direct_failed ? 0 : tl_route[0] >= 0 ? tl_route[0] : g_gemm_direct; }
static inline int rt_shield() { return tl_route[1] >= 0 ? tl_route[1] : g_gemm_direct_shield; }
static inline int rt_depth() { return tl_route[2] >= 0 ? tl_route[2] : g_gemm_direct_depth; }
static inline int rt_reserve() { return tl_route[3] >= 0 ? tl_route[3] : g_gemm_direct_reserve; }
bool direct_ok(const GemmArgs& g) {
  // 5: only the products of the encoders' backward chains (batch-reduce convolution weight gradients, small outputs): they run
  // BESIDE the decoder's resident LDS-tiled stream-K workgroups (101 VGPRs x 4 per SIMD, 135 of 160 KB LDS), where a kernel that
  // needs no LDS and <= 108 VGPRs is the one that still gets a wave per SIMD
  if (rt_direct() == 5 && !(g.kbatch > 1 || (long)g.M * g.N < 400000)) return false;
  return rt_direct() && g.sam == 1 && g.sbn == 1 && g.scn == 1 && g.K % 2 == 0 && g.K >= 64 && g.M >= 64 && g.N >= 64 &&
         ((long)g.M + g.sak) * 4 < (1L << 31) && ((long)g.N + g.sbk) * 4 < (1L << 31);
}
// Measured (tools/gemm_direct_probe.py, TFLOP/s incl. the zero fill of C; LDS-tiled stream-K kernel -> direct): dW_hh 3072 x 1024 x 8160
// 95.5 -> 134.9, dW_ih0 3072 x 2286 108.4 -> 134.9, dW_l2 1131 x 1024 97.6 -> 117.7, dW_l0 1024 x 1262 94.5 -> 113.0, style conv0
// dW 3402 x 512 x 12288 107.3 -> 128.1, 4096^3 124.6 -> 140.8.  Big outputs (>= 384 tiles of 128 x 128) take the 128 x 64 wave
// tile with one workgroup per CU (fewer operand bytes per product), the others 64 x 64 wave tiles, two workgroups per CU (finer
// grain at the ends of the stream-K ranges).
static thread_local bool tl_asum_consumed = false;      // set by launch_tn_direct when it was given GemmArgs.asum (gemm_tn_bias)
int launch_tn_direct(GemmArgs g, hipStream_t s) {
  if (g.asum) tl_asum_consumed = true;
  const bool big = rt_direct() == 2 || (rt_direct() == 1 && (long)cdiv(g.M, 128) * cdiv(g.N, 128) >= 384);      // (3, 5: 64 x 64)
  const int mt = big ? 4 : 2, nt = 2;
  const int tx = cdiv(g.N, 64 * nt), ty = cdiv(g.M, 64 * mt);
  const int cpb = cdiv(g.K / 2, 8);
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  // shield = 2: only the big single-segment products (the decoder's weight gradients on the second queue); the encoders' chain
  // products stay the kind that fits in beside other queues' workgroups
  const bool shield = rt_shield() == 1 || (rt_shield() == 2 && g.kbatch == 1 && (long)g.M * g.N >= 400000);
  long nwg = (long)ncu * (shield ? 1 : g_gemm_direct_wgs > 0 ? g_gemm_direct_wgs : (big ? 1 : 2));
  // option "gemm_direct_reserve": CUs a shielded product leaves free.  For data-parallel runs: the collective's workgroups live for
  // the whole exchange, and a stream-K product whose equal-share workgroups do not ALL become resident takes twice as long (the
  // stragglers start when the first ones end); with the CUs of the collective left out of the grid nobody waits for anybody.
  // (On one GPU, where nothing else is resident: 8 / 16 / 32 reserved CUs measured 17.13 / 17.17 / 17.02 ms against 17.03.)
  if (shield && rt_reserve() > 0 && rt_reserve() < ncu / 2) nwg = ncu - rt_reserve();
  const long total = (long)tx * ty * cpb * g.kbatch;
  if (nwg > total / 4) nwg = total / 4 > 0 ? total / 4 : 1;     // at least 4 chunks (64 k) per workgroup
  const int dep = rt_depth();
#define ZG_DIRECT(MT_, D_)                                                                                                    \
  do {                                                                                                                         \
    if (shield) hipLaunchKernelGGL((gemm_tn_direct_shield_kernel<MT_, 2, D_>), dim3((unsigned)nwg), dim3(256), 0, s, g, tx, ty, cpb); \
    else hipLaunchKernelGGL((gemm_tn_direct_kernel<MT_, 2, D_>), dim3((unsigned)nwg), dim3(256), 0, s, g, tx, ty, cpb);      \
  } while (0)
  // (deeper than 8 was measured under the shield -- 10 / 12 / 16 pairs: nothing; the wait counter's 6 bits end at (D - 1) x 6 <= 63)
  if (mt == 4) { if (dep >= 8) ZG_DIRECT(4, 8); else if (dep >= 6) ZG_DIRECT(4, 6); else ZG_DIRECT(4, 4); }
  else { if (dep >= 8) ZG_DIRECT(2, 8); else if (dep >= 6) ZG_DIRECT(2, 6); else ZG_DIRECT(2, 4); }
#undef ZG_DIRECT
  ZLAUNCH_CHECK("gemm_tn_direct");
  return 0;
}

#ifndef ZEGGS_GEMM_SK256
#define ZEGGS_GEMM_SK256 2      // 2: by output size (default)
#endif
template <int BM, int BN, int WM, int WN>
int launch_streamk_cfg(GemmArgs g, hipStream_t s) {
  const int tx = cdiv(g.N, BN), ty = cdiv(g.M, BM);
  const int kt = cdiv(g.K, BK) * g.kbatch;
  g.splitk = kt;
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  constexpr int resident = WM * WN > 4 ? 2 : ZEGGS_GEMM_MINB;
  long nwg = (long)ncu * (g_gemm_streamk_wgs > 0 ? g_gemm_streamk_wgs : resident);
  const long total = (long)tx * ty * kt;
  // at least 16 k-tiles per workgroup: every workgroup ends in one or two partial-tile epilogues of 64 atomics per lane, and with
  // 9 k-tiles each (the style encoder's second convolution: 97 tiles x 96 k-tiles over 1 024 workgroups) those were the kernel:
  // 87.7 us at 4 workgroups per CU, 71.8 us at 2 (tools/style_prof.sh)
  if (nwg > total / 16) nwg = total / 16 > 0 ? total / 16 : 1;
  dim3 grid((unsigned)nwg), block(WM * WN * 64);
  const bool akc = (g.sak == 1), bkc = (g.sbk == 1) && (g.sbn != 1 || g.N == 1);
  if (!akc && g.sam != 1) { zeggs_set_error("gemm: A has no unit stride (sam=%ld sak=%ld)", g.sam, g.sak); return -1; }
  if (!bkc && g.sbn != 1) { zeggs_set_error("gemm: B has no unit stride (sbk=%ld sbn=%ld)", g.sbk, g.sbn); return -1; }
  if (akc && bkc) hipLaunchKernelGGL((gemm_streamk_kernel<BM, BN, WM, WN, true, true>), grid, block, 0, s, g, tx, ty);
  else if (akc && !bkc) hipLaunchKernelGGL((gemm_streamk_kernel<BM, BN, WM, WN, true, false>), grid, block, 0, s, g, tx, ty);
  else if (!akc && bkc) hipLaunchKernelGGL((gemm_streamk_kernel<BM, BN, WM, WN, false, true>), grid, block, 0, s, g, tx, ty);
  else hipLaunchKernelGGL((gemm_streamk_kernel<BM, BN, WM, WN, false, false>), grid, block, 0, s, g, tx, ty);
  ZLAUNCH_CHECK("gemm_streamk");
  return 0;
}
// First use of a direct-kernel variant on this process: a small ragged product with row sums against a float64 host sum.  The
// variant's operand loads are inline asm the compiler cannot see into ("=v" destinations that are only valid after a hand-placed
// wait): its correctness rests on the register allocation of the compiler that built the library -- so it is CHECKED
// where it runs instead of trusted, like the persistent kernels' first use; a mismatch disables the direct kernel for the process
// (the LDS-tiled stream-K kernel takes over) and says so on stderr.  Never inside a stream capture (the check synchronises).
static int g_direct_checked[2][3][2];      // [128 x 64 wave tile][depth 4 / 6 / 8][shield]: 0 not yet, 1 ok
static bool direct_selftest(bool big, int dep, bool shield) {
  const int M = big ? 64 * 4 * 2 + 37 : 64 * 2 * 2 + 21, N = 128 + 27, K = 2 * 173;      // ragged tiles, odd pair count
  const long ldA = M + 3, ldB = N + 5;
  std::vector<float> hA((size_t)K * ldA), hB((size_t)K * ldB), hC((size_t)M * N), hS(M);
  unsigned lcg = 12345u;
  auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (float)((int)(lcg >> 9) - (1 << 22)) * (1.f / (1 << 22)); };
  for (auto& v : hA) v = rnd();
  for (auto& v : hB) v = rnd();
  float *dA = nullptr, *dB = nullptr, *dC = nullptr, *dS = nullptr;
  bool ok = hipMalloc(&dA, hA.size() * 4) == hipSuccess && hipMalloc(&dB, hB.size() * 4) == hipSuccess &&
            hipMalloc(&dC, hC.size() * 4) == hipSuccess && hipMalloc(&dS, hS.size() * 4) == hipSuccess;
  ok = ok && hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
       hipMemset(dC, 0, hC.size() * 4) == hipSuccess && hipMemset(dS, 0, hS.size() * 4) == hipSuccess;
  if (ok) {
    GemmArgs g = gemm_args(dA, dB, dC, M, N, K);      // C(m, n) = sum_k A[k][m] B[k][n]
    g.sam = 1; g.sak = ldA; g.sbk = ldB; g.sbn = 1; g.scm = N; g.scn = 1; g.asum = dS;
    int sv[4];
    for (int i = 0; i < 4; ++i) sv[i] = tl_route[i];
    tl_route[0] = big ? 2 : 3; tl_route[1] = shield ? 1 : 0; tl_route[2] = dep; tl_route[3] = 0;
    const bool sv_failed = g_direct_failed;
    g_direct_failed = false;
    ok = launch_tn_direct(g, (hipStream_t)0) == 0;
    tl_asum_consumed = false;
    g_direct_failed = sv_failed;
    for (int i = 0; i < 4; ++i) tl_route[i] = sv[i];
    ok = ok && hipStreamSynchronize((hipStream_t)0) == hipSuccess &&
         hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(hS.data(), dS, hS.size() * 4, hipMemcpyDeviceToHost) == hipSuccess;
  }
  if (ok) {
    double worst = 0.0;
    for (int m = 0; m < M; ++m) {
      double sa = 0.0;
      for (int k = 0; k < K; ++k) sa += hA[(size_t)k * ldA + m];
      worst = fmax(worst, fabs(sa - hS[m]));
    }
    for (int m = 0; m < M; m += 3)          // (every third row: the columns cover every lane of every fragment)
      for (int n = 0; n < N; ++n) {
        double acc = 0.0;
        for (int k = 0; k < K; ++k) acc += (double)hA[(size_t)k * ldA + m] * hB[(size_t)k * ldB + n];
        worst = fmax(worst, fabs(acc - hC[(size_t)m * N + n]));
      }
    ok = worst < 2e-3;      // sums of 346 products of magnitude <= 1: fp32 rounding is ~1e-5; a stale operand is O(1)
    if (!ok) fprintf(stderr, "zeggs: direct TN GEMM self-test (wave tile %s, depth %d%s) is off by %.3g\n", big ? "128x64" : "64x64",
                     dep, shield ? ", shield" : "", worst);
  }
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dS);
  return ok;
}
// true: the variant launch_tn_direct would pick for g is checked (or cannot be checked right now: stream capture)
static bool direct_checked(const GemmArgs& g, hipStream_t s) {
  const bool big = rt_direct() == 2 || (rt_direct() == 1 && (long)cdiv(g.M, 128) * cdiv(g.N, 128) >= 384);
  const bool shield = rt_shield() == 1 || (rt_shield() == 2 && g.kbatch == 1 && (long)g.M * g.N >= 400000);
  const int di = rt_depth() >= 8 ? 2 : rt_depth() >= 6 ? 1 : 0;
  int& st = g_direct_checked[big][di][shield];
  if (st == 1) return true;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return true;
  if (direct_selftest(big, di == 2 ? 8 : di == 1 ? 6 : 4, shield)) { st = 1; return true; }
  fprintf(stderr, "zeggs: the direct TN GEMM kernel is DISABLED for this process (self-test failed: built with another compiler?); "
                  "the LDS-tiled stream-K kernel takes its products\n");
  g_direct_failed = true;
  return false;
}
int launch_streamk(GemmArgs g, hipStream_t s) {
  if (gemm_split_ok(g)) return launch_tn_split(g, s);
  if (direct_ok(g) && direct_checked(g, s)) return launch_tn_direct(g, s);
  if (dma_ok(g)) return launch_tn_dma(g, s);
  // 256 x 128 tiles (8 waves, 2 workgroups per CU: 3/4 of the operand bytes per product) pay on the big outputs only: +5 .. +8 %
  // on dW_ih0 (432 tiles of 128 x 128), -4 .. -6 % at 72 .. 200 tiles where the coarser grain costs balance
  // (profiles/r03_gemm_ablation.txt); -DZEGGS_GEMM_SK256=0 / 1 forces never / from 512 rows on
#if ZEGGS_GEMM_SK256 == 1
  if (g.M >= 512) return launch_streamk_cfg<256, 128, 4, 2>(g, s);
#elif ZEGGS_GEMM_SK256 != 0
  if ((long)cdiv(g.M, 128) * cdiv(g.N, 128) >= 384 && g.M >= 512) return launch_streamk_cfg<256, 128, 4, 2>(g, s);
#endif
  return launch_streamk_cfg<128, 128, 2, 2>(g, s);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Batch-sized products y[M <= 64, N] = act(x W + bias) in ONE launch (the CellStateEncoder forward and backward, hid_1, the
// step-1 pose product in front of the training rollout, the per-step products of the generic decoder path): the split-K recipe
// below costs three dependent launches (zero fill, atomics, bias + activation) on chains that are nothing but launch latency.
// A workgroup owns 16 output columns, its 8 waves split K in 16-wide blocks (block = wave + 8 i), every lane loads 4 consecutive
// k of one x row (16 bytes) and the matching 4 weights -- WKC: W[n][k], 16 bytes of one row (nn.Linear forward, y = x W^T);
// else W[k][n], four 4-byte loads down a column (input gradients, dx = dy W) -- and feeds 4 v_mfma_f32_16x16x4 with them (the
// contraction order inside a block is a permutation that both operands share); the waves' partial tiles meet in LDS; bias,
// beta * y, the activation or the factor act'(saved output) of a backward chain are applied by the reducing threads.
// All workgroups read the same x (L2 hits); W is read exactly once.
struct SkinnyArgs {
  const float *x, *W, *bias;
  float* y;
  const float* ysave;      // != null: y *= act'(ysave) (ysave = the forward's activation output, row stride ldys)
  long ldx, ldw, ldy, ldys;
  int M, N, K, act;
  float beta;
};
template <int NBM, bool WKC>
__device__ __forceinline__ void skinny_body(const SkinnyArgs& a, int bx) {
  __shared__ f4 red[8][NBM][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  const int M = a.M, N = a.N, K = a.K;
  const int n0 = bx * 16, nblk = (K + 15) >> 4;
  const int wcol = n0 + r < N ? n0 + r : N - 1;                 // (clamped columns compute garbage nobody stores)
  const float* wp = WKC ? a.W + (long)wcol * a.ldw + 4 * g : a.W + (long)(4 * g) * a.ldw + wcol;
  const float* xp[NBM];
#pragma unroll
  for (int nb = 0; nb < NBM; ++nb) { const int row = nb * 16 + r; xp[nb] = a.x + (long)(row < M ? row : M - 1) * a.ldx + 4 * g; }
  f4 acc[NBM];
#pragma unroll
  for (int nb = 0; nb < NBM; ++nb) acc[nb] = f4{0.f, 0.f, 0.f, 0.f};
  constexpr int GR = 4;                                        // blocks per group: the next group's loads fly under this one's products
  float wa[GR][4], xa[GR][NBM][4], wq[GR][4], xq[GR][NBM][4];
  auto load = [&](float (&w)[GR][4], float (&xv)[GR][NBM][4], int i0) {
#pragma unroll
    for (int u = 0; u < GR; ++u) {
      const int kb = wave + 8 * (i0 + u);
      const int nv = kb < nblk ? K - (kb * 16 + 4 * g) : 0;     // valid k of this lane's quad (<= 0: all zero)
      if constexpr (WKC) load_contig<4>(w[u], wp + kb * 16, nv);
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) w[u][j] = j < nv ? wp[(long)(kb * 16 + j) * a.ldw] : 0.f;
      }
#pragma unroll
      for (int nb = 0; nb < NBM; ++nb) load_contig<4>(xv[u][nb], xp[nb] + kb * 16, nv);
    }
  };
  auto comp = [&](const float (&w)[GR][4], const float (&xv)[GR][NBM][4]) {
#pragma unroll
    for (int u = 0; u < GR; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < NBM; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u][nb][j], w[u][j], acc[nb], 0, 0, 0);
  };
  const int ni = (nblk - wave + 7) / 8;                         // blocks of this wave
  load(wa, xa, 0);
  for (int i = 0; i < ni; i += 2 * GR) {
    if (i + GR < ni) load(wq, xq, i + GR);
    comp(wa, xa);
    if (i + 2 * GR < ni) load(wa, xa, i + 2 * GR);
    if (i + GR < ni) comp(wq, xq);
  }
#pragma unroll
  for (int nb = 0; nb < NBM; ++nb) red[wave][nb][lane] = acc[nb];
  __syncthreads();
  if (tid < NBM * 64) {
    const int nb = tid >> 6, l = tid & 63;
    f4 sum = red[0][nb][l];
#pragma unroll
    for (int w = 1; w < 8; ++w) sum += red[w][nb][l];
    const int col = n0 + (l & 15);
    if (col < N) {
      const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = nb * 16 + 4 * (l >> 4) + q;             // D layout: lane holds rows 4 (lane / 16) + q of column lane % 16
        if (row < M) {
          float* yp = a.y + (long)row * a.ldy + col;
          float v = sum[q] + bv;
          if (a.beta != 0.f) v += a.beta * *yp;
          if (a.ysave) {
            const float ys = a.ysave[(long)row * a.ldys + col];
            v = a.act == ACT_ELU ? v * d_elu_grad_from_out(ys) : a.act == ACT_RELU ? (ys > 0.f ? v : 0.f) : v;
          } else v = d_act(v, a.act);
          *yp = v;
        }
      }
    }
  }
}
template <int NBM, bool WKC>
__global__ __launch_bounds__(512) void skinny_k(SkinnyArgs a) {
  skinny_body<NBM, WKC>(a, blockIdx.x);
}
// up to three INDEPENDENT batch-sized products in one launch (same row count, same weight layout): workgroups [0, nb[0]) take the
// first, the next nb[1] the second ... -- the chains these products sit in are nothing but launch latency (round 6: the decoder's
// prologue in front of the training rollout: CellStateEncoder layer 0 | hid_1 | the step-1 pose product, then the two halves of
// the CellStateEncoder's last layer)
struct SkinnyMulti { SkinnyArgs p[3]; int nb[3]; };
template <int NBM, bool WKC>
__global__ __launch_bounds__(512) void skinny_multi_k(SkinnyMulti m) {
  int bx = blockIdx.x;
  if (bx < m.nb[0]) { skinny_body<NBM, WKC>(m.p[0], bx); return; }
  bx -= m.nb[0];
  if (bx < m.nb[1]) { skinny_body<NBM, WKC>(m.p[1], bx); return; }
  skinny_body<NBM, WKC>(m.p[2], bx - m.nb[1]);
}

int launch_skinny(const SkinnyArgs& a, bool wkc, hipStream_t s) {
  const int nbm = cdiv(a.M, 16);
  dim3 grid(cdiv(a.N, 16)), block(512);
#define ZEGGS_SKINNY(NB)                                                             \
  do {                                                                               \
    if (wkc) hipLaunchKernelGGL((skinny_k<NB, true>), grid, block, 0, s, a);         \
    else hipLaunchKernelGGL((skinny_k<NB, false>), grid, block, 0, s, a);            \
  } while (0)
  if (nbm == 1) ZEGGS_SKINNY(1);
  else if (nbm == 2) ZEGGS_SKINNY(2);
  else if (nbm == 3) ZEGGS_SKINNY(3);
  else ZEGGS_SKINNY(4);
#undef ZEGGS_SKINNY
  ZLAUNCH_CHECK("gemm_skinny");
  return 0;
}
// the shapes / layouts the one-launch kernel takes: C row-major, A k-contiguous, B k- or n-contiguous, no batching
bool skinny_ok(const GemmArgs& g, int nbatch) {
  return g.M <= 64 && nbatch == 1 && g.kbatch == 1 && g.sak == 1 && g.scn == 1 && g.alpha == 1.f && g.N >= 64 && g.K >= 64 &&
         ((g.sbk == 1 && g.sbn != 1) || g.sbn == 1) && (g.beta == 0.f || (g.bias == nullptr && g.act == ACT_NONE));
}
SkinnyArgs skinny_args(const GemmArgs& g) {
  SkinnyArgs a;
  memset(&a, 0, sizeof(a));
  a.x = g.A; a.ldx = g.sam; a.W = g.B; a.y = g.C; a.ldy = g.scm; a.bias = g.bias; a.M = g.M; a.N = g.N; a.K = g.K;
  a.act = g.act; a.beta = g.beta;
  a.ldw = (g.sbk == 1 && g.sbn != 1) ? g.sbn : g.sbk;
  return a;
}

}  // namespace

void zeggs_gemm_set_dma(int on) { g_gemm_dma = on; }
void zeggs_gemm_set_direct(int mode, int wgs) { if (mode >= 0) g_gemm_direct = mode; if (wgs >= 0) g_gemm_direct_wgs = wgs; }
void zeggs_gemm_set_direct_depth(int d) { g_gemm_direct_depth = d; }
void zeggs_gemm_set_direct_shield(int on) { g_gemm_direct_shield = on; }
void zeggs_gemm_set_direct_reserve(int n) { g_gemm_direct_reserve = n; }
extern "C" int zeggs_gemm_route(int direct, int shield, int depth, int reserve) {
  tl_route[0] = direct; tl_route[1] = shield; tl_route[2] = depth; tl_route[3] = reserve;
  return 0;
}
extern "C" int zeggs_gemm_route_get(int* out4) {      // what this thread's next TN product would be routed by
  out4[0] = rt_direct(); out4[1] = rt_shield(); out4[2] = rt_depth(); out4[3] = rt_reserve();
  return 0;
}
void zeggs_gemm_set_asum(int on) { g_gemm_asum = on; }
int g_gemm_skinny = 1;          // zeggs_set_option("gemm_skinny", 0/1): batch-sized NT products in one launch
int g_gemm_streamk = 1;        // zeggs_set_option("gemm_streamk", 0/1): stream-K instead of the many-workgroup split-K
int g_gemm_mid_split = 1;      // zeggs_set_option("gemm_mid_split", 0/1): split K of the latency-bound narrow-output products

// stream-K pays from K = 1280 on.  Below, its equal-share workgroups spend their time on the fix-up atomics of tiles they share: the two
// fold products of the decoder packs (3072 x 1024 and 1024 x 1024 at K = 1131, once per optimizer step) take 108 + 56 us alone as
// stream-K and 93 + 39 us as plain split-K tiles (tools/fold_probe.py; round 6: the threshold was 1024, every other stream-K product of the
// training step has K >= 1536)
constexpr long SK_MIN_KTILES = 80;
int launch_gemm(GemmArgs g, int nbatch, hipStream_t s) {
  if (g.M <= 0 || g.N <= 0 || nbatch <= 0) return 0;
  if (g.nb1 <= 0) g.nb1 = 1;
  if (g.kbatch <= 0) g.kbatch = 1;
  g.splitk = 1;
  if (g_gemm_skinny && skinny_ok(g, nbatch)) return launch_skinny(skinny_args(g), g.sbk == 1 && g.sbn != 1, s);
  if (g.M <= 32) {
    // batch-sized products (M = B rows against a whole weight matrix: CellStateEncoder, the per-step GEMMs of the
    // generic decoder path): N / 128 workgroups would leave most of the chip idle -> split K over workgroups
    // (fp32 atomics onto a zeroed / existing C), bias + activation in a second tiny pass
    const long tiles = (long)cdiv(g.N, 128) * nbatch;
    const long ktiles = (long)cdiv(g.K, 16) * g.kbatch;
    const bool plain_beta = g.beta == 0.f || (g.beta == 1.f && g.bias == nullptr && g.act == ACT_NONE);
    if (nbatch == 1 && g.scn == 1 && plain_beta && ktiles >= 16 && tiles < 128) {
      long sk = (256 + tiles - 1) / tiles;
      if (sk > ktiles / 4) sk = ktiles / 4;
      if (sk > 64) sk = 64;
      if (sk > 1) {
        const float* bias = g.bias;
        const int act = g.act;
        const long n = (long)g.M * g.N, blocks = (n + 255) / 256;
        if (g.beta == 0.f) {
          hipLaunchKernelGGL(rows_fill_k, dim3((unsigned)(blocks > 1024 ? 1024 : blocks)), dim3(256), 0, s, g.C, g.M, g.N, g.scm);
          ZLAUNCH_CHECK("gemm_rows_fill");
        }
        g.bias = nullptr; g.act = ACT_NONE; g.splitk = (int)sk;
        ZTRY((launch_cfg<32, 128, 1, 4>(g, nbatch, s)));
        if (bias || act != ACT_NONE) {
          hipLaunchKernelGGL(rows_bias_act_k, dim3((unsigned)(blocks > 1024 ? 1024 : blocks)), dim3(256), 0, s, g.C, bias, g.M,
                             g.N, g.scm, act);
          ZLAUNCH_CHECK("gemm_rows_bias_act");
        }
        return 0;
      }
    }
    return launch_cfg<32, 128, 1, 4>(g, nbatch, s);
  }
  const bool big = g.M > 64 && g.N > 64;
  const long tiles = big ? (long)cdiv(g.M, 128) * cdiv(g.N, 128) * nbatch : (long)cdiv(g.M, 64) * cdiv(g.N, 64) * nbatch;
  const long ktiles = (long)cdiv(g.K, 16) * g.kbatch;
  // Mid-size products with a narrow output (the style / speech encoders' convolutions and projections: M = B L = 12 288 rows,
  // N = 128, K = 384 .. 1536): 100 - 400 output tiles walk 24 - 96 k-tiles each with one or two workgroups per CU -- every
  // k-tile pays its global-load latency in the open (75 - 200 us for 1 - 5 GFLOP).  Split K over a few workgroups per tile
  // (fp32 atomics onto a zeroed C; bias + activation, if any, in a tiny second pass -- the skinny-M recipe above for any M).
  {
    const bool flat_c = g.nb1 == 1 && g.scn == 1 && (nbatch == 1 || g.bsC0 == (long)g.M * g.scm);
    const bool plain_beta = g.beta == 0.f || (g.beta == 1.f && g.bias == nullptr && g.act == ACT_NONE);
    const long tiles64 = (long)cdiv(g.M, 64) * cdiv(g.N, 64) * nbatch;
    const bool streamk_case = g_gemm_streamk && nbatch == 1 && g.bias == nullptr && g.act == ACT_NONE && g.scm == g.N && big &&
                              ktiles >= SK_MIN_KTILES && tiles < 1024 * 3;
    if (g_gemm_mid_split && flat_c && plain_beta && !streamk_case && tiles64 >= 32 && tiles64 < 768 && ktiles >= 16) {
      long sk = (1024 + tiles64 - 1) / tiles64;
      if (sk > ktiles / 6) sk = ktiles / 6;
      if (sk > 8) sk = 8;
      if (sk > 1) {
        const float* bias = g.bias;
        const int act = g.act;
        const int rows = nbatch * g.M;
        const long n = (long)rows * g.N, blocks = (n + 255) / 256;
        if (g.beta == 0.f) {
          hipLaunchKernelGGL(rows_fill_k, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, s, g.C, rows, g.N, g.scm);
          ZLAUNCH_CHECK("gemm_rows_fill");
        }
        g.bias = nullptr; g.act = ACT_NONE; g.splitk = (int)sk;
        ZTRY((launch_cfg<64, 64, 2, 2>(g, nbatch, s)));
        if (bias || act != ACT_NONE) {
          hipLaunchKernelGGL(rows_bias_act_k, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, s, g.C, bias, rows,
                             g.N, g.scm, act);
          ZLAUNCH_CHECK("gemm_rows_bias_act");
        }
        return 0;
      }
    }
  }
  // few output tiles but a long contraction (weight gradients): split K over workgroups, combine with fp32 atomics
  const bool can_split = nbatch == 1 && (g.beta == 0.f || g.beta == 1.f) && g.bias == nullptr && g.act == ACT_NONE &&
                         g.scn == 1 && g.scm == g.N && ktiles >= 64;
  if (g_gemm_streamk && can_split && big && g.nb1 == 1 && tiles < 1024 * 3 && ktiles >= SK_MIN_KTILES) {
    if (g.beta == 0.f) ZTRY(k_fill(g.C, (long)g.M * g.N, 0.f, s));     // beta == 1: the atomics accumulate onto C
    return launch_streamk(g, s);
  }
  if (tiles < g_gemm_wg_target * 5 / 6 && can_split) {
    long sk = (g_gemm_wg_target + tiles - 1) / tiles;
    if (sk > ktiles / 16) sk = ktiles / 16;
    if (sk > 32) sk = 32;
    if (sk > 1) {
      g.splitk = (int)sk;
      if (g.beta == 0.f) ZTRY(k_fill(g.C, (long)g.M * g.N, 0.f, s));   // beta == 1: the atomics accumulate onto C
    }
  }
  // 1 .. 2 tiles of 128 x 128 per CU and no split (bias / activation epilogue): half the CUs would carry two tiles, the others one;
  // 128 x 64 tiles give every CU the same share (the style encoder's first conv: 384 -> 768 tiles)
  if (ZEGGS_GEMM_HALF_TILES && big && g.splitk == 1 && tiles > 256 && tiles < 512 && g.N % 64 == 0)
    return launch_cfg<128, 64, 2, 2>(g, nbatch, s);
  if (big && (tiles * g.splitk >= 128 || g.splitk > 1)) return launch_cfg<128, 128, 2, 2>(g, nbatch, s);
  return launch_cfg<64, 64, 2, 2>(g, nbatch, s);
}

GemmArgs gemm_args(const float* A, const float* B, float* C, int M, int N, int K) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K;
  g.nb1 = 1; g.kbatch = 1; g.splitk = 1; g.alpha = 1.f; g.beta = 0.f; g.act = ACT_NONE;
  return g;
}

// y[M,N] = act(x[M,K] W[N,K]^T + bias)       (nn.Linear forward)
int gemm_nt(const float* x, long ldx, const float* W, long ldw, float* y, long ldy, const float* bias,
            int M, int N, int K, int act, float beta, hipStream_t s) {
  GemmArgs g = gemm_args(x, W, y, M, N, K);
  g.sam = ldx; g.sak = 1; g.sbk = 1; g.sbn = ldw; g.scm = ldy; g.scn = 1;
  g.bias = bias; g.act = act; g.beta = beta;
  return launch_gemm(g, 1, s);
}
// n <= 3 independent nn.Linear forwards y_i = act_i(x_i W_i^T + bias_i) with the same row count M: ONE launch when every one of
// them is a batch-sized product the one-launch kernel takes (skinny_ok), else one after the other
int gemm_nt_multi(const GemmNtItem* it, int n, int M, hipStream_t s) {
  bool one = g_gemm_skinny && n >= 2 && n <= 3 && M <= 64;
  GemmArgs g[3];
  for (int i = 0; i < n && i < 3; ++i) {
    g[i] = gemm_args(it[i].x, it[i].W, it[i].y, M, it[i].N, it[i].K);
    g[i].sam = it[i].ldx; g[i].sak = 1; g[i].sbk = 1; g[i].sbn = it[i].ldw; g[i].scm = it[i].ldy; g[i].scn = 1;
    g[i].bias = it[i].bias; g[i].act = it[i].act;
    one = one && skinny_ok(g[i], 1);
  }
  if (!one) {
    for (int i = 0; i < n; ++i)
      ZTRY(gemm_nt(it[i].x, it[i].ldx, it[i].W, it[i].ldw, it[i].y, it[i].ldy, it[i].bias, M, it[i].N, it[i].K, it[i].act, 0.f, s));
    return 0;
  }
  SkinnyMulti m;
  memset(&m, 0, sizeof(m));
  int total = 0;
  for (int i = 0; i < n; ++i) { m.p[i] = skinny_args(g[i]); m.nb[i] = cdiv(it[i].N, 16); total += m.nb[i]; }
  const int nbm = cdiv(M, 16);
  dim3 grid(total), block(512);
  if (nbm == 1) hipLaunchKernelGGL((skinny_multi_k<1, true>), grid, block, 0, s, m);
  else if (nbm == 2) hipLaunchKernelGGL((skinny_multi_k<2, true>), grid, block, 0, s, m);
  else if (nbm == 3) hipLaunchKernelGGL((skinny_multi_k<3, true>), grid, block, 0, s, m);
  else hipLaunchKernelGGL((skinny_multi_k<4, true>), grid, block, 0, s, m);
  ZLAUNCH_CHECK("gemm_skinny_multi");
  return 0;
}
// dx[M,K] = beta*dx + dy[M,N] W[N,K]          (input gradient of nn.Linear)
int gemm_nn(const float* dy, long lddy, const float* W, long ldw, float* dx, long lddx, int M, int N_contract,
            int K_out, float beta, hipStream_t s) {
  GemmArgs g = gemm_args(dy, W, dx, M, K_out, N_contract);
  g.sam = lddy; g.sak = 1; g.sbk = ldw; g.sbn = 1; g.scm = lddx; g.scn = 1; g.beta = beta;
  return launch_gemm(g, 1, s);
}
// dx[M,K] = (beta*dx + dy[M,N] W[N,K]) * act'(ysave)   (input gradient of nn.Linear through the preceding activation, whose
// output the forward saved): one launch for batch-sized M, else the product followed by the elementwise pass
int gemm_nn_actbwd(const float* dy, long lddy, const float* W, long ldw, float* dx, long lddx, int M, int N_contract,
                   int K_out, float beta, const float* ysave, long ldys, int act, hipStream_t s) {
  GemmArgs g = gemm_args(dy, W, dx, M, K_out, N_contract);
  g.sam = lddy; g.sak = 1; g.sbk = ldw; g.sbn = 1; g.scm = lddx; g.scn = 1; g.beta = beta;
  if (g_gemm_skinny && skinny_ok(g, 1)) {
    SkinnyArgs a = skinny_args(g);
    a.ysave = ysave; a.ldys = ldys; a.act = act;
    return launch_skinny(a, false, s);
  }
  ZTRY(launch_gemm(g, 1, s));
  if (lddx != K_out || ldys != K_out) { zeggs_set_error("gemm_nn_actbwd: strided rows need the one-launch kernel (M <= 64)"); return -1; }
  return k_act_bwd(dx, dx, ysave, (long)M * K_out, act, 1.f, s);
}
// dW[N,K] = beta*dW + dy[M,N]^T x[M,K]        (weight gradient of nn.Linear)
int gemm_tn(const float* dy, long lddy, const float* x, long ldx, float* dW, long lddw, int M_contract, int N,
            int K, float beta, hipStream_t s) {
  GemmArgs g = gemm_args(dy, x, dW, N, K, M_contract);
  g.sam = 1; g.sak = lddy; g.sbk = ldx; g.sbn = 1; g.scm = lddw; g.scn = 1; g.beta = beta;
  return launch_gemm(g, 1, s);
}

int gemm_tn_bias(const float* dy, long lddy, const float* x, long ldx, float* dW, long lddw, int M_contract, int N,
                 int K, float beta, float* db, hipStream_t s) {
  GemmArgs g = gemm_args(dy, x, dW, N, K, M_contract);
  g.sam = 1; g.sak = lddy; g.sbk = ldx; g.sbn = 1; g.scm = lddw; g.scn = 1; g.beta = beta;
  // (beta == 0 would need db zeroed first: the separate launch does that; the bf16-split experiment has no row sums either)
  g.asum = (beta == 1.f && g_gemm_asum && !g_gemm_split_bf16) ? db : nullptr;
  tl_asum_consumed = false;
  ZTRY(launch_gemm(g, 1, s));
  if (!tl_asum_consumed) ZTRY(k_colsum(db, dy, M_contract, N, lddy, beta, s));
  return 0;
}

extern "C" int zeggs_gemm_direct_selftest(int big, int depth, int shield) {
  return direct_selftest(big != 0, depth >= 8 ? 8 : depth >= 6 ? 6 : 4, shield != 0) ? 1 : 0;
}
extern "C" int zeggs_gemm_tn_bias(const float* dy, long lddy, const float* x, long ldx, float* dW, long lddw, int M_contract, int N,
                                  int K, float beta, float* db, void* stream) {
  return gemm_tn_bias(dy, lddy, x, ldx, dW, lddw, M_contract, N, K, beta, db, (hipStream_t)stream);
}

extern "C" int zeggs_gemm(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                          long sam, long sak, long sbk, long sbn, long scm, long scn, int nbatch, long bsA,
                          long bsB, long bsC, float alpha, float beta, int act, void* stream) {
  GemmArgs g = gemm_args(A, B, C, M, N, K);
  g.sam = sam; g.sak = sak; g.sbk = sbk; g.sbn = sbn; g.scm = scm; g.scn = scn;
  g.bsA0 = bsA; g.bsB0 = bsB; g.bsC0 = bsC; g.nb1 = 1;
  g.bias = bias; g.alpha = alpha; g.beta = beta; g.act = act;
  return launch_gemm(g, nbatch, (hipStream_t)stream);
}

extern "C" int zeggs_gemm_kbatch(const float* A, const float* B, float* C, int M, int N, int K, long sam, long sak, long sbk,
                                 long sbn, long scm, long scn, int kbatch, long kbsA, long kbsB, float beta, void* stream) {
  GemmArgs g = gemm_args(A, B, C, M, N, K);
  g.sam = sam; g.sak = sak; g.sbk = sbk; g.sbn = sbn; g.scm = scm; g.scn = scn;
  g.kbatch = kbatch; g.kbsA = kbsA; g.kbsB = kbsB; g.beta = beta;
  return launch_gemm(g, 1, (hipStream_t)stream);
}
