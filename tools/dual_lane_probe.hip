// Lane-semantics probe for the dual-chain forward sweep (csrc/train_dual.hip), gfx950:
//   (1) v_mfma_f32_4x4x1 with cbsz = 2, abid = a: groups of 4 blocks share the A values of block 4 g + a, i.e. one instruction is a
//       [4 rows x 4 k x 16 batch columns] product: lane l reads B(b = l % 16, k = 4 (l / 16) + a) and accumulates the partial sum of
//       k-quarter l / 16 for batch column l % 16, rows in the 4 result registers;
//   (2) v_permlane16_swap / v_permlane32_swap: which 16-lane rows trade places (the fold of the four k-quarters).
//   hipcc --offload-arch=gfx950 -O3 -o tools/dual_lane_probe tools/dual_lane_probe.hip && tools/dual_lane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int ABID>
__global__ void bcast2_k(const float* a, const float* b, float* d) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], acc, 2, ABID, 0);
  for (int v = 0; v < 4; ++v) d[threadIdx.x * 4 + v] = acc[v];
}
__global__ void swap_k(unsigned* o) {
  const unsigned l = threadIdx.x;
  const auto s16 = __builtin_amdgcn_permlane16_swap(l, 100u + l, false, false);
  const auto s32 = __builtin_amdgcn_permlane32_swap(l, 100u + l, false, false);
  o[l] = s16[0]; o[64 + l] = s16[1]; o[128 + l] = s32[0]; o[192 + l] = s32[1];
}
// the fold itself as the kernel uses it: acc[g][e] at lane (kq, b) -> row j holds gate order[j] summed over kq
__global__ void fold_k(const float* in /* [4 gates][64 lanes] */, float* out) {
  const int l = threadIdx.x;
  auto f = [](float x) { return __float_as_uint(x); };
  const auto s01 = __builtin_amdgcn_permlane32_swap(f(in[0 * 64 + l]), f(in[1 * 64 + l]), false, false);
  const float v1 = __uint_as_float(s01[0]) + __uint_as_float(s01[1]);
  const auto s23 = __builtin_amdgcn_permlane32_swap(f(in[2 * 64 + l]), f(in[3 * 64 + l]), false, false);
  const float v2 = __uint_as_float(s23[0]) + __uint_as_float(s23[1]);
  const auto s = __builtin_amdgcn_permlane16_swap(f(v1), f(v2), false, false);
  out[l] = __uint_as_float(s[0]) + __uint_as_float(s[1]);
}

int main() {
  float *a, *b, *d;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 4096);
  std::vector<float> ha(64), hb(64), hd(256);
  for (int l = 0; l < 64; ++l) { ha[l] = 1.f + l; hb[l] = 100.f + 3.f * l; }
  hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
  int badall = 0;
  for (int ab = 0; ab < 4; ++ab) {
    if (ab == 0) hipLaunchKernelGGL(bcast2_k<0>, dim3(1), dim3(64), 0, 0, a, b, d);
    if (ab == 1) hipLaunchKernelGGL(bcast2_k<1>, dim3(1), dim3(64), 0, 0, a, b, d);
    if (ab == 2) hipLaunchKernelGGL(bcast2_k<2>, dim3(1), dim3(64), 0, 0, a, b, d);
    if (ab == 3) hipLaunchKernelGGL(bcast2_k<3>, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int v = 0; v < 4; ++v) {
        const int src = 4 * (l / 16) + ab;      // block whose A values group l / 16 uses
        const float want = ha[src * 4 + v] * hb[l];
        if (hd[l * 4 + v] != want) { if (bad < 4) printf("abid %d lane %d v %d: got %g want %g\n", ab, l, v, hd[l * 4 + v], want); ++bad; }
      }
    printf("cbsz=2 abid=%d (D[lane][v] = A[(4 (lane/16) + abid, v)] * B[lane]): %s\n", ab, bad ? "WRONG" : "confirmed");
    badall += bad;
  }
  unsigned* o; hipMalloc(&o, 1024);
  std::vector<unsigned> ho(256);
  hipLaunchKernelGGL(swap_k, dim3(1), dim3(64), 0, 0, o);
  hipMemcpy(ho.data(), o, 1024, hipMemcpyDeviceToHost);
  const char* nm[4] = {"permlane16_swap [0]", "permlane16_swap [1]", "permlane32_swap [0]", "permlane32_swap [1]"};
  for (int r = 0; r < 4; ++r) {
    printf("%s rows:", nm[r]);
    for (int row = 0; row < 4; ++row) printf("  %u..%u", ho[r * 64 + row * 16], ho[r * 64 + row * 16 + 15]);
    printf("\n");
  }
  // fold: gate g at lane (kq, b) = 1000 g + 10 b + small kq part; expect row j = gate order[j], value 4000 g + 40 b + (0 + 1 + 2 + 3)
  std::vector<float> hin(256), hout(64);
  for (int g = 0; g < 4; ++g) for (int l = 0; l < 64; ++l) hin[g * 64 + l] = 1000.f * g + 10.f * (l % 16) + (l / 16);
  float *din, *dout; hipMalloc(&din, 1024); hipMalloc(&dout, 256);
  hipMemcpy(din, hin.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(fold_k, dim3(1), dim3(64), 0, 0, din, dout);
  hipMemcpy(hout.data(), dout, 256, hipMemcpyDeviceToHost);
  printf("fold: gate held by row j:");
  int bad = 0;
  for (int j = 0; j < 4; ++j) {
    const int g = (int)(hout[j * 16] / 4000.f + 0.01f);
    printf(" %d", g);
    for (int bb = 0; bb < 16; ++bb) if (hout[j * 16 + bb] != 4000.f * g + 40.f * bb + 6.f) ++bad;
  }
  printf("   (%s)\n", bad ? "values WRONG" : "values exact");
  return badall || bad;
}
