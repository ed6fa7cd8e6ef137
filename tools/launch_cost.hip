// Host cost of a kernel launch on this stack, by argument-block size and launch form (tools/README.md):
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_cost tools/launch_cost.hip && /tmp/launch_cost
// Prints, for N dependent launches of a ~2 us kernel on one stream: host time per launch (enqueue only) and wall time per launch.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

template <int BYTES> struct Args { float* p; int n; char pad[BYTES - 12]; };
template <int BYTES> __global__ void k(Args<BYTES> a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.n) a.p[i] = a.p[i] * 1.0001f + (float)a.pad[0];
}
__global__ void k_idx(const Args<768>* table, int idx) {
  const Args<768>& a = table[idx];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.n) a.p[i] = a.p[i] * 1.0001f + (float)a.pad[0];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int BYTES> void run(float* p, int n, int N, hipStream_t s, const char* what) {
  Args<BYTES> a{};
  a.p = p; a.n = n;
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k<BYTES>, dim3(128), dim3(512), 0, s, a);
  (void)hipStreamSynchronize(s);
  const double t0 = now();
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k<BYTES>, dim3(128), dim3(512), 0, s, a);
  const double t1 = now();
  (void)hipStreamSynchronize(s);
  const double t2 = now();
  printf("%-34s args %4d B: host %.2f us / launch, wall %.2f us / launch\n", what, BYTES, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
}

int main() {
  const int n = 128 * 512, N = 4000;
  float* p;
  (void)hipMalloc(&p, n * sizeof(float));
  (void)hipMemset(p, 0, n * sizeof(float));
  hipStream_t s;
  (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  run<16>(p, n, N, s, "created stream");
  run<256>(p, n, N, s, "created stream");
  run<768>(p, n, N, s, "created stream");
  run<768>(p, n, N, nullptr, "legacy default stream");
  run<16>(p, n, N, nullptr, "legacy default stream");
  {   // argument table on the device, index as the only kernel argument
    std::vector<Args<768>> h(N);
    for (auto& a : h) { a = Args<768>{}; a.p = p; a.n = n; }
    Args<768>* d;
    (void)hipMalloc(&d, N * sizeof(Args<768>));
    const double t0 = now();
    (void)hipMemcpyAsync(d, h.data(), N * sizeof(Args<768>), hipMemcpyHostToDevice, s);
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_idx, dim3(128), dim3(512), 0, s, d, i);
    const double t1 = now();
    (void)hipStreamSynchronize(s);
    const double t2 = now();
    printf("%-34s args   12 B: host %.2f us / launch, wall %.2f us / launch\n", "device-resident argument table", (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
  }
  {   // graph replay
    Args<768> a{};
    a.p = p; a.n = n;
    hipGraph_t g; hipGraphExec_t e;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k<768>, dim3(128), dim3(512), 0, s, a);
    (void)hipStreamEndCapture(s, &g);
    (void)hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    (void)hipGraphLaunch(e, s);
    (void)hipStreamSynchronize(s);
    const double t0 = now();
    (void)hipGraphLaunch(e, s);
    const double t1 = now();
    (void)hipStreamSynchronize(s);
    const double t2 = now();
    printf("%-34s args  768 B: host %.2f us / node, wall %.2f us / node\n", "hipGraph replay", (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
  }
  return 0;
}
