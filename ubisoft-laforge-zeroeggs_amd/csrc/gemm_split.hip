// EXPERIMENT (option "gemm_split_bf16" = 3 / 6 / 9, default 0 = off; never used by a default path): the fp32 TN products of the
// training tail (weight gradients dW = dy^T x; ZEGGS/train.py:424 through autograd) on the BF16 matrix cores with an fp32-exact
// operand split.  gfx950 has no xf32; its fp32 MFMA runs at 1/16 of the bf16 rate and the native kernel (gemm.hip:
// gemm_tn_direct_kernel) already sits at 0.80-0.87 of that peak.  Every fp32 operand is cut into three bf16 planes
//   h = top 16 bits of a,  m = top 16 bits of (a - h),  l = top 16 bits of (a - h - m)       (truncation: a = h + m + l EXACTLY,
//                                                                                             8 + 8 + 8 significant bits)
// and the product is formed from NP of the nine plane products with fp32 accumulation (v_mfma_f32_32x32x16_bf16), small terms
// first:  NP = 9: all;  NP = 6: hh, hm, mh, mm, hl, lh (the dropped ml, lm, ll are <= 2^-24 of a term: fp32 rounding level);
// NP = 3: hh, hm, mh (2^-16: NOT fp32-exact, listed as the speed reference only).  16 / NP times the fp32 matrix rate before the
// operand feed.
//
// Same skeleton as the native direct kernel: no LDS, no barriers, stream-K over (tile, k-chunk), fp32 atomics onto C, one wave per
// SIMD with its whole register file (the "shield" form), 2 x 2 waves of a workgroup share operand fragments through L1.
// What differs:
//   * a k-step is 16 deep (one matrix instruction per plane product): a lane needs 8 CONSECUTIVE k of its row -- lanes 0..31 rows
//     k0..k0+7, lanes 32..63 rows k0+8..k0+15 of the operand as it lies in memory (k-major);
//   * the 128 x 64 wave tile is interleaved: tile t of the four A tiles holds rows m_base + 4 q + t (q = lane & 31), so ONE
//     16-byte load per k-row feeds all four tiles (8-byte loads for the two B tiles): 16 load instructions per k-step instead of
//     48, 512 contiguous bytes per half-wave; two register sets of two k-steps each alternate (one multiplied, one in flight);
//   * the split is VALU work on the loaded registers (and, sub, and, sub per value; v_perm packs two k into a register).
// Needs K % 16 == 0 per segment, 16-byte aligned A rows (lda % 4), 8-byte aligned B rows (ldb % 2), rows padded to the vector width.
#include <hip/hip_runtime.h>

#include "common.h"
#include "gemm.h"
#include "kernels.h"

int g_gemm_split_bf16 = 0;      // zeggs_set_option("gemm_split_bf16", 0 / 3 / 6 / 9)

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
inline int cdivi(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ unsigned top16_pair(float x0, float x1) {      // (x1 & 0xffff0000) | (x0 >> 16): two bf16, truncated
  return __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
}
__device__ __forceinline__ float top16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

// planes of one operand tile for one k-step: [plane h / m / l][k pair 0..3] = 8 bf16 per plane
struct Planes { unsigned p[3][4]; };
template <int NPL>
__device__ __forceinline__ void split_pair(float x0, float x1, Planes& P, int jp) {
  const float h0 = top16(x0), h1 = top16(x1);
  P.p[0][jp] = top16_pair(x0, x1);
  const float r0 = x0 - h0, r1 = x1 - h1;
  P.p[1][jp] = top16_pair(r0, r1);
  if constexpr (NPL > 2) {
    const float s0 = r0 - top16(r0), s1 = r1 - top16(r1);
    P.p[2][jp] = top16_pair(s0, s1);
  }
}
__device__ __forceinline__ bf8 plane(const Planes& P, int pl) {
  return __builtin_bit_cast(bf8, uint4{P.p[pl][0], P.p[pl][1], P.p[pl][2], P.p[pl][3]});
}

template <int NP, int D>
__device__ __forceinline__ void tn_split_body(const GemmArgs& g, int tiles_x, int tiles_y, int chunks_per_batch) {
  constexpr int MT = 4, NT = 2, BM = 64 * MT, BN = 64 * NT;
  constexpr int NPL = NP >= 6 ? 3 : 2;                         // planes per operand
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const int kh = lane >> 5, q = lane & 31;
  const int kt = chunks_per_batch * g.kbatch;                  // k-steps (16 deep) of one output tile
  const long total = (long)tiles_x * tiles_y * kt;
  const unsigned nwg = gridDim.x, w = blockIdx.x, xcd = w & 7, idx = w >> 3, qq = nwg >> 3, r = nwg & 7;
  const unsigned wl = xcd * qq + (xcd < r ? xcd : r) + idx;
  long it = total * wl / nwg;
  const long it_end = total * (wl + 1) / nwg;
  while (it < it_end) {
    const int tile = (int)(it / kt), c0 = (int)(it % kt);
    const long left = it_end - it;
    const int c1 = (long)(kt - c0) < left ? kt : c0 + (int)left;
    constexpr int GM = 4;
    const int width = GM * tiles_x, group = tile / width, first = group * GM, gsz = tiles_y - first < GM ? tiles_y - first : GM;
    const int m_base = (first + (tile % width) % gsz) * BM + wm * 128, n_base = ((tile % width) / gsz) * BN + wn * 64;
    // this lane's vector of every k-row: A columns m_base + 4 q .. + 3 (tile t = column 4 q + t), B columns n_base + 2 q, + 1;
    // clamped to the last vector inside the padded row (duplicates are not stored)
    const int Mv = (g.M + 3) & ~3, Nv = (g.N + 1) & ~1;
    int ma = m_base + 4 * q, nb = n_base + 2 * q;
    ma = ma < Mv ? ma : Mv - 4;
    nb = nb < Nv ? nb : Nv - 2;
    const unsigned voa = (unsigned)(((long)ma + (long)kh * 8 * g.sak) * 4), vob = (unsigned)(((long)nb + (long)kh * 8 * g.sbk) * 4);
    f16v acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int c = c0;
    while (c < c1) {
      const int kb = c / chunks_per_batch, cb = c - kb * chunks_per_batch;
      int ce = c1 - kb * chunks_per_batch;
      ce = ce < chunks_per_batch ? ce : chunks_per_batch;
      const int n = ce - cb;                                    // k-steps of this segment
      const float* Ab = g.A + (long)kb * g.kbsA + (long)(16 * cb) * g.sak;
      const float* Bb = g.B + (long)kb * g.kbsB + (long)(16 * cb) * g.sbk;
      // Two register sets of S = D k-steps each, in turn: while one is multiplied the other is being filled for the NEXT round -- plain
      // loads and the compiler's own wait insertion (exact here: at the top of a round the only loads in flight are the ones it is
      // about to use, issued a whole round earlier).  The native direct kernel's hand-counted inline-asm loads do not carry over: with
      // this kernel's register count the allocator split the live ranges of the asm destinations, i.e. copied registers whose loads
      // had not landed (measured: memory faults) -- the hazard ADVICE r5 describes.
      f4 xa[D][8], ya[D][8];
      f2v xb[D][8], yb[D][8];
      const char* Ap = (const char*)Ab + voa;
      const char* Bp = (const char*)Bb + vob;
      const long a1 = g.sak * 4, b1 = g.sbk * 4;
      auto load_set = [&](f4 (&sa)[D][8], f2v (&sb)[D][8], int step0) {
#pragma unroll
        for (int u = 0; u < D; ++u)
          if (step0 + u < n) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              sa[u][j] = *(const f4*)(Ap + ((long)(step0 + u) * 16 + j) * a1);
              sb[u][j] = *(const f2v*)(Bp + ((long)(step0 + u) * 16 + j) * b1);
            }
          }
      };
      // one k-step: the split of A tile t + 1 (VALU) is issued in the shadow of tile t's matrix instructions (the scheduling groups
      // below: one matrix instruction, then up to four vector instructions, twelve times); only the split of the two B tiles and of
      // A tile 0 stands in the open
      auto mma = [&](const f4 (&ra)[8], const f2v (&rb)[8]) {
        Planes PB[NT], PA[2];
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
#pragma unroll
          for (int u = 0; u < NT; ++u) split_pair<NPL>(rb[2 * jp][u], rb[2 * jp + 1][u], PB[u], jp);
          split_pair<NPL>(ra[2 * jp][0], ra[2 * jp + 1][0], PA[0], jp);
        }
        // plane products, smallest first: (a plane, b plane)
        constexpr int ORD9[9][2] = {{2, 2}, {1, 2}, {2, 1}, {0, 2}, {2, 0}, {1, 1}, {0, 1}, {1, 0}, {0, 0}};
        constexpr int ORD6[6][2] = {{0, 2}, {2, 0}, {1, 1}, {0, 1}, {1, 0}, {0, 0}};
        constexpr int ORD3[3][2] = {{0, 1}, {1, 0}, {0, 0}};
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          __builtin_amdgcn_sched_barrier(0);
          if (t + 1 < MT) {
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) split_pair<NPL>(ra[2 * jp][t + 1], ra[2 * jp + 1][t + 1], PA[(t + 1) & 1], jp);
          }
#pragma unroll
          for (int o = 0; o < NP; ++o) {
            const int pa = NP == 9 ? ORD9[o][0] : NP == 6 ? ORD6[o][0] : ORD3[o][0];
            const int pb = NP == 9 ? ORD9[o][1] : NP == 6 ? ORD6[o][1] : ORD3[o][1];
#pragma unroll
            for (int u = 0; u < NT; ++u)
              acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(plane(PA[t & 1], pa), plane(PB[u], pb), acc[t][u], 0, 0, 0);
          }
#pragma unroll
          for (int k = 0; k < NP * NT; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one matrix instruction
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // up to four vector instructions behind it
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      auto mma_set = [&](const f4 (&sa)[D][8], const f2v (&sb)[D][8], int step0) {
#pragma unroll
        for (int u = 0; u < D; ++u)
          if (step0 + u < n) mma(sa[u], sb[u]);
      };
      load_set(xa, xb, 0);
      for (int p = 0; p < n; p += 2 * D) {
        __builtin_amdgcn_sched_barrier(0);
        load_set(ya, yb, p + D);
        __builtin_amdgcn_sched_barrier(0);
        mma_set(xa, xb, p);
        __builtin_amdgcn_sched_barrier(0);
        load_set(xa, xb, p + 2 * D);
        __builtin_amdgcn_sched_barrier(0);
        mma_set(ya, yb, p + D);
      }
      c = (kb + 1) * chunks_per_batch < c1 ? (kb + 1) * chunks_per_batch : c1;
    }
    // partial sums -> fp32 atomics onto C: tile (t, u), lane column index q -> column n_base + 2 q + u, row index
    // i = (e & 3) + 8 (e >> 2) + 4 kh -> row m_base + 4 i + t
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int nn = n_base + 2 * q + u;
        if (nn >= g.N) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m_base + 4 * ((e & 3) + 8 * (e >> 2) + 4 * kh) + t;
          if (m < g.M) atomicAdd(g.C + (long)m * g.scm + nn, g.alpha * acc[t][u][e]);
        }
      }
    it += c1 - c0;
  }
}
template <int NP, int D>
__global__ __launch_bounds__(256, 1) void gemm_tn_split_kernel(GemmArgs g, int tiles_x, int tiles_y, int chunks_per_batch) {
  asm volatile("" ::: "v255", "a255");      // the whole register file of its SIMDs (gemm.hip: the shield form)
  tn_split_body<NP, D>(g, tiles_x, tiles_y, chunks_per_batch);
}

}  // namespace

bool gemm_split_ok(const GemmArgs& g) {
  const int np = g_gemm_split_bf16;
  if (!(np == 3 || np == 6 || np == 9)) return false;
  const long Mv = (g.M + 3) & ~3L, Nv = (g.N + 1) & ~1L;
  return g.sam == 1 && g.sbn == 1 && g.scn == 1 && g.K % 16 == 0 && g.K >= 64 && g.M >= 64 && g.N >= 64 && g.asum == nullptr &&
         g.sak % 4 == 0 && g.sbk % 2 == 0 && Mv <= g.sak && Nv <= g.sbk && (uintptr_t)g.A % 16 == 0 && (uintptr_t)g.B % 8 == 0 &&
         (g.kbatch <= 1 || (g.kbsA % 4 == 0 && g.kbsB % 2 == 0)) &&
         (Mv + 8 * g.sak) * 4 < (1L << 31) && (Nv + 8 * g.sbk) * 4 < (1L << 31);
}
int launch_tn_split(GemmArgs g, hipStream_t s) {
  const int tx = cdivi(g.N, 128), ty = cdivi(g.M, 256);
  const int cpb = g.K / 16;
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  long nwg = ncu;
  const long total = (long)tx * ty * cpb * g.kbatch;
  if (nwg > total / 4) nwg = total / 4 > 0 ? total / 4 : 1;
  switch (g_gemm_split_bf16) {
    case 9: hipLaunchKernelGGL((gemm_tn_split_kernel<9, 2>), dim3((unsigned)nwg), dim3(256), 0, s, g, tx, ty, cpb); break;
    case 6: hipLaunchKernelGGL((gemm_tn_split_kernel<6, 2>), dim3((unsigned)nwg), dim3(256), 0, s, g, tx, ty, cpb); break;
    default: hipLaunchKernelGGL((gemm_tn_split_kernel<3, 2>), dim3((unsigned)nwg), dim3(256), 0, s, g, tx, ty, cpb); break;
  }
  ZLAUNCH_CHECK("gemm_tn_split");
  return 0;
}
