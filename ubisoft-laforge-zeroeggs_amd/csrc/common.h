// Common device/host helpers for the ZeroEGGS gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

// ---------------------------------------------------------------- error state
// last-error string (thread-local); every extern "C" entry returns 0 or -1.
extern "C" const char* zeggs_last_error();
void zeggs_set_error(const char* fmt, ...);

#define ZCHECK(cond, ...)                        \
  do {                                           \
    if (!(cond)) {                               \
      zeggs_set_error(__VA_ARGS__);              \
      return -1;                                 \
    }                                            \
  } while (0)

#define ZLAUNCH_CHECK(name)                                                        \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) {                                                       \
      zeggs_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
      return -1;                                                                   \
    }                                                                              \
  } while (0)

#define ZTRY(call)                 \
  do {                             \
    int r__ = (call);              \
    if (r__ != 0) return r__;      \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// bump allocator over a caller-provided workspace (no hipMalloc in the library)
struct Arena {
  char* base;
  size_t off, cap;
  bool dry;  // dry = size query only
  Arena(void* p, size_t bytes) : base((char*)p), off(0), cap(bytes), dry(p == nullptr) {}
  float* f(size_t n) { return (float*)raw(n * sizeof(float)); }
  void* raw(size_t bytes) {
    size_t o = align_up(off, 256);
    off = o + bytes;
    return dry ? nullptr : (void*)(base + o);
  }
  bool ok() const { return dry || off <= cap; }
};

// ---------------------------------------------------------------- device math
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
// 4-byte aligned vector types: rows of PyTorch-shaped weights (K = 1262, 81, ...)
// are not 16-byte aligned; the compiler picks a legal access for the alignment.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ float d_elu(float x) { return x > 0.f ? x : expm1f(x); }
// derivative of ELU expressed with the OUTPUT y: y>0 -> 1, else y+1 (= e^x)
__device__ __forceinline__ float d_elu_grad_from_out(float y) { return y > 0.f ? 1.f : y + 1.f; }
// GRU gates on the hardware transcendentals (v_exp_f32 = 2^x and v_rcp_f32, 1 ulp each): ~7 instructions per gate instead of the
// ~35 of expf + IEEE division / ~45 of tanhf -- they sit in the epilogue of every hand-off of the persistent rollouts.  Absolute
// error <= 2e-7 (sigmoid) / 4e-7 (tanh), the rounding level of the fp32 sums they are applied to.  -DZEGGS_EXACT_GATES=1: library calls.
#ifndef ZEGGS_EXACT_GATES
#define ZEGGS_EXACT_GATES 0
#endif
#if ZEGGS_EXACT_GATES
__device__ __forceinline__ float d_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float d_tanh(float x) { return tanhf(x); }
#else
__device__ __forceinline__ float d_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float d_tanh(float x) {
  return 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)) - 1.f;
}
#endif

enum { ACT_NONE = 0, ACT_ELU = 1, ACT_RELU = 2 };
__device__ __forceinline__ float d_act(float x, int act) {
  if (act == ACT_ELU) return d_elu(x);
  if (act == ACT_RELU) return x > 0.f ? x : 0.f;
  return x;
}

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
struct Q4 {
  float w, x, y, z;
};
// reference anim/tquat.py:18-20: t = 2 (q_v x v); v + w t + q_v x t
__device__ __forceinline__ V3 quat_mul_vec(Q4 q, V3 v) {
  V3 qv = v3(q.x, q.y, q.z);
  V3 t = 2.0f * cross(qv, v);
  return v + q.w * t + cross(qv, t);
}
__device__ __forceinline__ Q4 quat_inv(Q4 q) { return Q4{q.w, -q.x, -q.y, -q.z}; }
// reference anim/tquat.py:6-15
__device__ __forceinline__ Q4 quat_mul(Q4 x, Q4 y) {
  return Q4{y.w * x.w - y.x * x.x - y.y * x.y - y.z * x.z,
            y.w * x.x + y.x * x.w - y.y * x.z + y.z * x.y,
            y.w * x.y + y.x * x.z + y.y * x.w - y.z * x.x,
            y.w * x.z - y.x * x.y + y.y * x.x + y.z * x.w};
}

// counter-based RNG for dropout masks / the VAE noise: same (seed, index) -> same bits in fwd and bwd.
// Round 5: 32-bit arithmetic.  The round-1..4 form was a 64-bit splitmix finaliser -- three 64 x 64 multiplies = ~8 quarter-rate
// v_mul_lo / v_mul_hi + 64-bit shifts, ~60 VALU instructions per element -- and the fused attention kernels draw one mask value
// per probability: the hash cost more issue time than their matrix-core products.  Now: the seed is mixed down to a 32-bit key
// on the SCALAR unit (it is wave-uniform), the element index is folded to 32 bits (the high word rotated in: indices 2^32 apart
// may share a value, which no mask of this library is long enough to notice) and goes through the three multiply-xorshift rounds
// of `triple32` (C. Wellons' hash-prospector search: bias 0.0208) -- 3 multiplies + 9 cheap ops.  Statistical quality on
// sequential counters (keep rates, neighbour / neighbouring-seed correlation, byte chi-squares, Box-Muller moments): tools/hash_quality.py.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 17; x *= 0xed5ad4bbu;
  x ^= x >> 11; x *= 0xac4c1b51u;
  x ^= x >> 15; x *= 0x31848babu;
  x ^= x >> 14;
  return x;
}
// Round 6: the seed keys TWO places.  With the key only added to the counter, every seed walks the SAME 2^32-long
// sequence from another offset, and two masks are shifted copies of each other whenever their keys lie closer than the mask is
// long (2^25 attention probabilities: ~1 % of the seed pairs of a run).  A second word derived from the seed, added between the
// first and the second multiply, makes the sequences of two seeds different functions of the counter (tools/hash_quality.py:
// window-overlap check); cost: one add.  Both words are wave-uniform: scalar registers.
struct HKey { uint32_t a, b; };
__device__ __forceinline__ HKey hash_key(uint64_t seed) {
  const uint32_t k = mix32(((uint32_t)seed * 0x9E3779B1u) ^ (uint32_t)(seed >> 32));
  return HKey{k, mix32(k ^ 0x85ebca6bu)};
}
__device__ __forceinline__ uint32_t mix32k(uint32_t x, uint32_t k2) {
  x ^= x >> 17; x *= 0xed5ad4bbu;
  x ^= x >> 11; x += k2; x *= 0xac4c1b51u;
  x ^= x >> 15; x *= 0x31848babu;
  x ^= x >> 14;
  return x;
}
__device__ __forceinline__ uint32_t hash_u32(uint64_t seed, uint64_t idx) {
  const uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
  const HKey k = hash_key(seed);
  return mix32k((lo ^ ((hi << 16) | (hi >> 16))) + k.a, k.b);
}
// keep-scale for element idx: 0 (dropped) or 1/(1-p)
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint64_t idx, float p) {
  if (p <= 0.f) return 1.f;
  float u = (float)(hash_u32(seed, idx) >> 8) * (1.0f / 16777216.0f);
  return u < p ? 0.f : 1.f / (1.f - p);
}
// the same value without the branch and with the loop invariants hoisted by the caller (key = hash_key(seed), the element index
// as two words, inv_keep = 1 / (1 - p)): the form the fused attention kernels use, one mask value per probability
__device__ __forceinline__ float dropout_scale_fast(HKey key, uint32_t lo, uint32_t hi, float p, float inv_keep) {
  const uint32_t hsh = mix32k((lo ^ ((hi << 16) | (hi >> 16))) + key.a, key.b);
  const float u = (float)(hsh >> 8) * (1.0f / 16777216.0f);
  return u < p ? 0.f : inv_keep;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum; all threads get the result. blockDim.x multiple of 64, <= 1024.
__device__ __forceinline__ float block_sum(float v, float* red /* >=16 floats LDS */) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}
