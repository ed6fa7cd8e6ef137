// v_mfma_f32_4x4x1_16b_f32 on gfx950: operand layout check + issue rate vs v_mfma_f32_16x16x4_f32.
// (go/no-go probe for 4-row tiles in the persistent kernels: 12 useful gate rows per workgroup are 3 x 4, not 16)
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma4_probe tools/mfma4_probe.hip && tools/mfma4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void layout_k(const float* a, const float* b, float* d) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
  for (int v = 0; v < 4; ++v) d[threadIdx.x * 4 + v] = acc[v];
}

// cbsz = 3, abid = a: blocks 0-7 take the A values of block a, blocks 8-15 those of block 8 + a
template <int ABID>
__global__ void bcast_k(const float* a, const float* b, float* d) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], acc, 3, ABID, 0);
  for (int v = 0; v < 4; ++v) d[threadIdx.x * 4 + v] = acc[v];
}

template <int KIND>
__global__ __launch_bounds__(512, 2) void rate_k(const float* a, const float* b, float* d, int n) {
  f4 acc[12];
  for (int i = 0; i < 12; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  float av = a[threadIdx.x & 63], bv = b[threadIdx.x & 63];
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[i], 0, 0, 0);
      else if (KIND == 2) acc[i % 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[i % 3], 3, 5, 0);   // 3 accumulators, broadcast A
      else if (KIND == 3) acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[0], 3, 5, 0);            // ONE dependent chain per wave
      else if (KIND == 4) acc[i % 2] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[i % 2], 3, 5, 0);    // two chains
      else if (KIND == 5) acc[i % 6] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[i % 6], 3, 5, 0);    // six chains
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
    }
  }
  f4 r = acc[0];
  for (int i = 1; i < 12; ++i) r += acc[i];
  ((f4*)d)[threadIdx.x + blockIdx.x * blockDim.x] = r;
}

int main() {
  float *a, *b, *d;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 256 * 512 * 16);
  std::vector<float> ha(64), hb(64), hd(256);
  for (int l = 0; l < 64; ++l) { ha[l] = 1.f + l; hb[l] = 100.f + 3.f * l; }
  hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(layout_k, dim3(1), dim3(64), 0, 0, a, b, d);
  hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
  // hypothesis: block = lane / 4 ; A lane (blk, i) = A_blk[i] ; B lane (blk, j) = B_blk[j] ; D lane (blk, j), vgpr v = A_blk[v] * B_blk[j]
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int v = 0; v < 4; ++v) {
      const int blk = l / 4;
      const float want = ha[blk * 4 + v] * hb[l];
      if (hd[l * 4 + v] != want) { if (bad < 8) printf("lane %d v %d: got %g want %g\n", l, v, hd[l * 4 + v], want); ++bad; }
    }
  printf("layout hypothesis (D[lane=(blk,j)][v=i] = A[(blk,i)] * B[(blk,j)]): %s\n", bad ? "WRONG" : "confirmed");
  for (int abid = 0; abid < 8; abid += 5) {
    if (abid == 0) hipLaunchKernelGGL((bcast_k<0>), dim3(1), dim3(64), 0, 0, a, b, d);
    else hipLaunchKernelGGL((bcast_k<5>), dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
    bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int v = 0; v < 4; ++v) {
        const int src = (l / 32) * 8 + abid;
        if (hd[l * 4 + v] != ha[src * 4 + v] * hb[l]) ++bad;
      }
    printf("cbsz=3 abid=%d (blocks 0-7 <- A block abid, blocks 8-15 <- A block 8+abid): %s\n", abid, bad ? "WRONG" : "confirmed");
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int n = 20000;
  for (int kind = 0; kind < 6; ++kind) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL((rate_k<0>), dim3(256), dim3(512), 0, 0, a, b, d, n);
      else if (kind == 1) hipLaunchKernelGGL((rate_k<1>), dim3(256), dim3(512), 0, 0, a, b, d, n);
      else if (kind == 2) hipLaunchKernelGGL((rate_k<2>), dim3(256), dim3(512), 0, 0, a, b, d, n);
      else if (kind == 3) hipLaunchKernelGGL((rate_k<3>), dim3(256), dim3(512), 0, 0, a, b, d, n);
      else if (kind == 4) hipLaunchKernelGGL((rate_k<4>), dim3(256), dim3(512), 0, 0, a, b, d, n);
      else hipLaunchKernelGGL((rate_k<5>), dim3(256), dim3(512), 0, 0, a, b, d, n);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double per = kind == 1 ? 2048.0 : 512.0;     // flops per instruction
      const double flops = per * 12.0 * n * 8 * 256;     // 8 waves x 256 workgroups
      if (rep) printf("%s: %.3f ms, %.1f TFLOP/s, %.1f ns per instruction per SIMD\n", kind == 0 ? "4x4x1_16b, 12 acc" : kind == 1 ? "16x16x4, 12 acc" : kind == 2 ? "4x4x1 cbsz3, 3 acc" : kind == 3 ? "4x4x1 cbsz3, 1 acc (one chain per wave, 2 waves per SIMD)" : kind == 4 ? "4x4x1 cbsz3, 2 acc" : "4x4x1 cbsz3, 6 acc",
                      ms, flops / ms * 1e-9, ms * 1e6 / (12.0 * n * 2));
    }
  }
  return 0;
}
