// Can the vector ALU of a SIMD work for wave B while a matrix instruction of wave A (same SIMD) is in flight?  gfx950.
// One workgroup per CU, 8 waves = 2 per SIMD (waves w and w + 4 share SIMD w % 4).  Roles per wave: M = a loop of independent
// v_mfma_f32_32x32x2_f32 (16 passes each), V = a loop of dependent-free v_fma_f32, - = exits at once.
//   hipcc --offload-arch=gfx950 -O3 -o tools/coissue_probe tools/coissue_probe.hip && tools/coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

// role of waves 0..3 = ra, of waves 4..7 = rb: 0 nothing, 1 matrix (32x32x2 f32), 2 vector, 3 matrix 16x16x4 f32, 4 matrix 4x4x1 f32
__global__ __launch_bounds__(512, 2) void k(float* out, int n, int ra, int rb, unsigned long long* clk, int prio) {
  const int wave = threadIdx.x >> 6, role = wave < 4 ? ra : rb;
  float x = out[threadIdx.x & 63], y = x + 1.f;
  f16v acc[4];
  f4 a4[8];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int i = 0; i < 8; ++i) a4[i] = f4{0.f, 0.f, 0.f, 0.f};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = x + i;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  if (role == 1) {
    if (prio == 2) __builtin_amdgcn_s_setprio(3);      // ... or the matrix wave ahead
    for (int it = 0; it < n; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);      // 4 x 64 cycles
  } else if (role == 2) {
    if (prio == 1) __builtin_amdgcn_s_setprio(3);      // the vector wave ahead of the matrix wave in the SIMD's arbitration
    for (int it = 0; it < n; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, y);                                   // 64 x 4 cycles
  } else if (role == 3) {
    for (int it = 0; it < n; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) a4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a4[i], 0, 0, 0);        // 8 x 32 cycles
  } else if (role == 4) {
    for (int it = 0; it < n; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) a4[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a4[i], 0, 0, 0);          // 8 x 8 cycles... x4 below
  }
  const unsigned long long t1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += a4[i][0] + a4[i][1] + a4[i][2] + a4[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) clk[wave] = t1 - t0;
}

int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 64);
  hipMemset(out, 0, 256 * 512 * 4);
  const int n = 20000;
  const char* names[] = {"-", "M32", "V", "M16", "M4"};
  const int cases[][3] = {{1, 0, 0}, {2, 0, 0}, {3, 0, 0}, {4, 0, 0}, {1, 1, 0}, {2, 2, 0}, {1, 2, 0}, {1, 2, 1}, {1, 2, 2}, {2, 1, 0}, {2, 1, 1}, {3, 2, 0}, {3, 2, 1}, {4, 2, 0}, {4, 2, 1}, {3, 3, 0}, {1, 3, 0}};
  for (auto& c : cases) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, 100, c[0], c[1], clk, c[2]);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, n, c[0], c[1], clk, c[2]);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8]; hipMemcpy(h, clk, 64, hipMemcpyDeviceToHost);
    printf("%s  waves 0-3: %-3s  waves 4-7: %-3s   kernel %7.3f ms   wave 0: %6.2f us  wave 4: %6.2f us  (100 MHz clock)\n", c[2] == 1 ? "V at s_setprio 3" : c[2] == 2 ? "M at s_setprio 3" : "               ", names[c[0]], names[c[1]], ms,
           h[0] / 100.0, h[4] / 100.0);
  }
  // expectation per loop at 2.4 GHz: M32 20000 x 4 x 64 cycles = 2.13 ms; V 20000 x 64 x 4 cycles = 2.13 ms; M16 20000 x 8 x 32 = 2.13 ms
  return 0;
}
