// v_mfma_f32_32x32x16_bf16 on gfx950: operand / result lane layout check for the fp32-split TN product (csrc/gemm_split.hip).
// Hypothesis: A lane l holds row l % 32, k = 8 (l / 32) + j (j = 0..7: 8 bf16 in 4 VGPRs, low half first); B lane l holds column
// l % 32, same k; D lane l holds column l % 32, rows (e & 3) + 8 (e >> 2) + 4 (l / 32), e = 0..15.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_bf16_probe tools/mfma_bf16_probe.hip && tools/mfma_bf16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(const unsigned* a, const unsigned* b, float* d) {
  const int l = threadIdx.x;
  uint4 ua = ((const uint4*)a)[l], ub = ((const uint4*)b)[l];
  f16v acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, ua), __builtin_bit_cast(b8, ub), acc, 0, 0, 0);
  for (int e = 0; e < 16; ++e) d[l * 16 + e] = acc[e];
}
static unsigned short bf(float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)(u >> 16); }
int main() {
  // A[i][k] = small integers (exact in bf16), B[k][n] likewise
  float A[32][16], B[16][32];
  for (int i = 0; i < 32; ++i) for (int kk = 0; kk < 16; ++kk) A[i][kk] = (float)((i * 3 + kk * 5) % 13 - 6);
  for (int kk = 0; kk < 16; ++kk) for (int n = 0; n < 32; ++n) B[kk][n] = (float)((kk * 7 + n * 2) % 11 - 5);
  std::vector<unsigned> ha(256), hb(256);
  for (int l = 0; l < 64; ++l)
    for (int jp = 0; jp < 4; ++jp) {
      const int k0 = 8 * (l / 32) + 2 * jp;
      ha[l * 4 + jp] = bf(A[l % 32][k0]) | ((unsigned)bf(A[l % 32][k0 + 1]) << 16);
      hb[l * 4 + jp] = bf(B[k0][l % 32]) | ((unsigned)bf(B[k0 + 1][l % 32]) << 16);
    }
  unsigned *da, *db; float* dd;
  hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 4096);
  hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
  std::vector<float> hd(1024);
  hipMemcpy(hd.data(), dd, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int e = 0; e < 16; ++e) {
      const int n = l % 32, i = (e & 3) + 8 * (e >> 2) + 4 * (l / 32);
      float want = 0.f;
      for (int kk = 0; kk < 16; ++kk) want += A[i][kk] * B[kk][n];
      if (hd[l * 16 + e] != want) { if (bad < 6) printf("lane %d e %d: got %g want %g\n", l, e, hd[l * 16 + e], want); ++bad; }
    }
  printf("32x32x16 bf16 layout hypothesis: %s\n", bad ? "WRONG" : "confirmed");
  return bad != 0;
}
