// Host-side interface of the generic strided fp32 MFMA GEMM (gemm.hip).
#pragma once
#include <hip/hip_runtime.h>

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;  // per output column n, may be null
  int M, N, K;
  long sam, sak, sbk, sbn, scm, scn;           // element strides; one of (sam,sak) and one of (sbk,sbn) must be 1
  long bsA0, bsA1, bsB0, bsB1, bsC0, bsC1;     // batch z -> (z / nb1, z % nb1)
  int nb1;
  int kbatch;                                  // batch-reduce: K loop runs over kbatch segments
  long kbsA, kbsB;
  int splitk;                                  // set by launch_gemm (split-K with atomic accumulation)
  float alpha, beta;
  int act;
  float* asum;                                 // TN direct kernel only, may be null: asum[m] += alpha * sum_k A(m, k) (the bias gradient that goes with
                                               // a weight gradient dW = dy^T x: the column sums of dy, from the operand fragments the product loads anyway)
};

GemmArgs gemm_args(const float* A, const float* B, float* C, int M, int N, int K);
int launch_gemm(GemmArgs g, int nbatch, hipStream_t s);
int gemm_nt(const float* x, long ldx, const float* W, long ldw, float* y, long ldy, const float* bias, int M,
            int N, int K, int act, float beta, hipStream_t s);
// up to three independent y_i = act_i(x_i W_i^T + bias_i) of the same row count in ONE launch when they are batch-sized (gemm.hip: skinny_multi_k)
struct GemmNtItem { const float* x; long ldx; const float* W; long ldw; float* y; long ldy; const float* bias; int N, K, act; };
int gemm_nt_multi(const GemmNtItem* items, int n, int M, hipStream_t s);
int gemm_nn(const float* dy, long lddy, const float* W, long ldw, float* dx, long lddx, int M, int N_contract,
            int K_out, float beta, hipStream_t s);
int gemm_nn_actbwd(const float* dy, long lddy, const float* W, long ldw, float* dx, long lddx, int M, int N_contract,
                   int K_out, float beta, const float* ysave, long ldys, int act, hipStream_t s);
int gemm_tn(const float* dy, long lddy, const float* x, long ldx, float* dW, long lddw, int M_contract, int N,
            int K, float beta, hipStream_t s);
// ... and db[N] (+)= column sums of dy in the same launch where the direct kernel takes the product (beta == 1: both outputs
// accumulate), by a separate column-sum launch otherwise
int gemm_tn_bias(const float* dy, long lddy, const float* x, long ldx, float* dW, long lddw, int M_contract, int N,
                 int K, float beta, float* db, hipStream_t s);
