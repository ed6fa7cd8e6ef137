// Price of an in-kernel grid barrier on this GPU (go / no-go for a persistent decoder kernel):
//   one workgroup per CU, `iters` rounds of { a small published store per thread; grid barrier }.
// Barrier forms: 0 = one monotonic counter, 1 = two-level (8 logical groups -> top counter -> per-group generation).
// Both publish with an agent-scope release and consume with an agent-scope acquire (per-XCD L2s are not coherent).
// Every spin is bounded (give-up code in status[0]); build: hipcc --offload-arch=gfx950 -O3 -o barrier_probe barrier_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Bar {
  unsigned* top;        // [32]   (one word used, padded to its own cache line)
  unsigned* grp;        // [8*32] per-group arrival counters
  unsigned* gen;        // [8*32] per-group generation words
  unsigned* status;     // [32]   status[0] != 0: a spin gave up
  int nwg, ngrp;
};

__device__ __forceinline__ bool spin_ge(unsigned* p, unsigned v, unsigned* status) {
  for (unsigned spins = 0; spins < (1u << 22); ++spins) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= v) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return false;
}

// epoch = 1, 2, 3, ... (counters are monotonic within a launch; zeroed by the host before it)
__device__ __forceinline__ bool grid_barrier(const Bar& b, unsigned epoch, int form) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave: its stores have left the CU
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bool good = true;
    if (form == 0) {
      __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      good = spin_ge(b.top, (unsigned)b.nwg * epoch, b.status);
    } else {
      const int g = blockIdx.x % b.ngrp;
      const unsigned members = (unsigned)((b.nwg - g + b.ngrp - 1) / b.ngrp);
      const unsigned old = __hip_atomic_fetch_add(b.grp + 32 * g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == members * epoch) {                   // last arriver of this group
        const unsigned o2 = __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (o2 + 1 == (unsigned)b.ngrp * epoch)           // last group: open every group's gate
          for (int x = 0; x < b.ngrp; ++x) __hip_atomic_store(b.gen + 32 * x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      good = spin_ge(b.gen + 32 * g, epoch, b.status);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    ok = good;
  }
  __syncthreads();
  return ok != 0;
}

__global__ __launch_bounds__(512) void probe_k(Bar b, float* data, int iters, int form, int payload) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x, n = gridDim.x * blockDim.x;
  float acc = 0.f;
  for (int it = 1; it <= iters; ++it) {
    // publish `payload` floats per thread, then read what ANOTHER workgroup published in the previous round
    float* slot = data + (long)(it & 1) * 16 * n;      // double-buffered: the next round must not overwrite what is read now
    for (int p = 0; p < payload; ++p) slot[(long)p * n + gid] = (float)it + acc * 1e-30f;
    if (!grid_barrier(b, (unsigned)it, form)) return;
    const int other = (gid + 512 * 37) % n;
    acc += slot[other];
  }
  if (acc != (float)iters * (iters + 1) / 2 && gid == 0) b.status[1] = 1;   // stale read detector (exact in fp32 for small iters)
}

__global__ void empty_k(float* d) { if (d == nullptr) d[0] = 0; }

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 2000;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int nwg = prop.multiProcessorCount;
  printf("CUs %d\n", nwg);
  unsigned* state;
  CHECK(hipMalloc(&state, 4096 * 4));
  float* data;
  CHECK(hipMalloc(&data, (size_t)nwg * 512 * 32 * 4));
  Bar b{state, state + 32, state + 32 + 256, state + 32 + 512, nwg, 8};
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int payload = 1; payload <= 16; payload *= 4)
    for (int form = 0; form < 2; ++form) {
      float best = 1e9f;
      unsigned st[2] = {0, 0};
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(state, 0, 4096 * 4));
        CHECK(hipMemset(data, 0, (size_t)nwg * 512 * 32 * 4));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe_k, dim3(nwg), dim3(512), 0, 0, b, data, iters, form, payload);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        CHECK(hipMemcpy(st, b.status, 8, hipMemcpyDeviceToHost));
      }
      printf("payload %2d floats/thread  form %d: %.2f us per round (gave up: %u, stale: %u)\n", payload, form,
             best * 1e3f / iters, st[0], st[1]);
    }
  // reference: dependent empty kernels
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(empty_k, dim3(nwg), dim3(512), 0, 0, data);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("empty kernel chain: %.2f us per launch\n", ms * 1e3f / 2000);
  return 0;
}
