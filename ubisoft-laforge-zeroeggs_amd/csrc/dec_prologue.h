// Device bodies of the decoder's prologue kernels -- frame 0 + CellStateEncoder input + the pose part of x_1 (dec_init), the
// speech / style columns of every step's canonical input row (dec_fill_cond) -- shared by decoder.hip (one launch each) and
// train_persistent.hip (tp_prologue_k: both and the rollout's own conditioning operands in ONE launch, round 6: the ten dependent
// launches in front of the training rollout were nothing but launch latency on an idle chip).
#pragma once
#include "../../include/zeggs_hip.h"
#include "dec_math.h"

// init: frame-0 outputs, CellStateEncoder input (gaze of frame 0) and the pose part of x_1 (gaze of frame 1); one block per batch row
__device__ __forceinline__ void dec_init_body(int b, const ZeggsDecDims& d, const ZeggsDecStats& st, const float* pose0,
                                              const float* rp0, const float* rr0, const float* gaze, const float* style,
                                              float* pose, float* rpos, float* rrot, float* cse_in, float* gin1, int GL) {
  const float* p0 = pose0 + (long)b * d.PO;
  for (int c = threadIdx.x; c < d.PO; c += blockDim.x) {
    float v = p0[c];
    pose[((long)b * d.T) * d.PO + c] = v;
    float e = (v - st.in_mean[c]) / st.in_std[c];
    cse_in[(long)b * (d.PI + d.ST) + c] = e;
    if (d.T > 1) gin1[(long)b * GL + d.H + c] = e;
  }
  for (int c = threadIdx.x; c < d.ST; c += blockDim.x)
    cse_in[(long)b * (d.PI + d.ST) + d.PI + c] = style[((long)b * d.T) * d.ST + c];
  if (threadIdx.x == 0) {
    Q4 q = Q4{rr0[b * 4], rr0[b * 4 + 1], rr0[b * 4 + 2], rr0[b * 4 + 3]};
    V3 rp = v3(rp0[b * 3], rp0[b * 3 + 1], rp0[b * 3 + 2]);
    float* o = rpos + ((long)b * d.T) * 3; o[0] = rp.x; o[1] = rp.y; o[2] = rp.z;
    float* r = rrot + ((long)b * d.T) * 4; r[0] = q.w; r[1] = q.x; r[2] = q.y; r[3] = q.z;
    for (int f = 0; f < 2 && f < d.T; ++f) {
      const float* gz = gaze + ((long)b * d.T + f) * 3;
      V3 gd = quat_mul_vec(quat_inv(q), v3(gz[0], gz[1], gz[2]) - rp);
      float gv[3] = {gd.x, gd.y, gd.z};
      for (int k = 0; k < 3; ++k) {
        float e = (gv[k] - st.in_mean[d.PO + k]) / st.in_std[d.PO + k];
        if (f == 0) cse_in[(long)b * (d.PI + d.ST) + d.PO + k] = e;
        else gin1[(long)b * GL + d.H + d.PO + k] = e;
      }
    }
  }
}

// speech / style columns of x_t for one step (or all steps when nt > 1): Gin[t][b][H+PI ...]; block `bid` of `nblocks`
__device__ __forceinline__ void dec_fill_cond_body(long bid, long nblocks, const ZeggsDecDims& d, const float* speech,
                                                   const float* style, float* gin, int GL, int t0, int nt, long slot_stride,
                                                   int ring) {
  const int XC = d.SP + (d.film ? 0 : d.ST);
  long n = (long)nt * d.B * XC;
  for (long i = bid * blockDim.x + threadIdx.x; i < n; i += nblocks * blockDim.x) {
    int c = (int)(i % XC);
    long r = i / XC;
    int b = (int)(r % d.B);
    int t = t0 + (int)(r / d.B);
    float v = c < d.SP ? speech[((long)b * d.T + t) * d.SP + c] : style[((long)b * d.T + t) * d.ST + (c - d.SP)];
    int slot = ring ? (t & 1) : t;
    gin[slot * slot_stride + (long)b * GL + d.H + d.PI + c] = v;
  }
}
