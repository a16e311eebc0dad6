// Pose-regression head in fp32: the two Linear(feat_dim, 3) layers of PoseNet
// (/root/reference/models/posenet.py:48-49,71-73), their backward with the optional NaN filter
// (filter_hook, :28-34,50-51), the ReLU gate of the feature vector (:66) and bias-gradient
// column sums.  feat_dim = 2048, B <= a few hundred: tiny, latency-bound kernels.
#pragma once
#include "common.h"

namespace mn {

// poses[b][0:3] = feat[b] . Wx^T + bx ; poses[b][3:6] = feat[b] . Wq^T + bq.  One wave per output.
static __global__ void __launch_bounds__(256) head_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ Wx,
                                                        const float* __restrict__ bx, const float* __restrict__ Wq,
                                                        const float* __restrict__ bq, float* __restrict__ poses, int B,
                                                        int K) {
  int widx = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (widx >= B * 6) return;  // whole wave exits together
  int b = widx / 6, o = widx % 6;
  const float* w = (o < 3) ? Wx + (long)o * K : Wq + (long)(o - 3) * K;
  const float* f = feat + (long)b * K;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += f[k] * w[k];
  s = wave_sum(s);
  if (lane == 0) poses[b * 6 + o] = s + ((o < 3) ? bx[o] : bq[o - 3]);
}

__device__ __forceinline__ float nan_to_zero(float v) { return (v != v) ? 0.f : v; }

// dz[b][k] = (feat[b][k] > 0) * ( sum_o dp[b][o] Wx[o][k]  +  filt( sum_o dp[b][3+o] Wq[o][k] ) )
// (feat is the post-ReLU feature, so the gate is the ReLU derivative of posenet.py:66)
static __global__ void __launch_bounds__(256) head_bwd_input_kernel(const float* __restrict__ dposes,
                                                              const float* __restrict__ feat,
                                                              const float* __restrict__ Wx, const float* __restrict__ Wq,
                                                              float* __restrict__ dz, int B, int K, int filter_nans) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * K) return;
  int k = (int)(i % K), b = (int)(i / K);
  const float* dp = dposes + b * 6;
  float gx = (dp[0] * Wx[k] + dp[1] * Wx[K + k]) + dp[2] * Wx[2 * K + k];
  float gq = (dp[3] * Wq[k] + dp[4] * Wq[K + k]) + dp[5] * Wq[2 * K + k];
  if (filter_nans) gq = nan_to_zero(gq);
  float g = gx + gq;
  dz[i] = (feat[i] > 0.f) ? g : 0.f;
}

// dW[o][k] += scale * sum_b dp[b][o] feat[b][k] ; db[o] += scale * sum_b dp[b][o]   (o = 0..5)
static __global__ void __launch_bounds__(256) head_bwd_weight_kernel(const float* __restrict__ dposes,
                                                               const float* __restrict__ feat, float* __restrict__ dWx,
                                                               float* __restrict__ dbx, float* __restrict__ dWq,
                                                               float* __restrict__ dbq, int B, int K, float scale,
                                                               int filter_nans) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 6 * (K + 1)) return;
  int o = i / (K + 1), k = i % (K + 1);
  float s = 0.f;
  if (k < K)
    for (int b = 0; b < B; ++b) s += dposes[b * 6 + o] * feat[(long)b * K + k];
  else
    for (int b = 0; b < B; ++b) s += dposes[b * 6 + o];
  s *= scale;
  if (o >= 3 && filter_nans) s = nan_to_zero(s);
  float* dst = (k < K) ? ((o < 3) ? dWx + (long)o * K + k : dWq + (long)(o - 3) * K + k)
                       : ((o < 3) ? dbx + o : dbq + (o - 3));
  *dst += s;
}

// out[n] += scale * sum_m in[m][n]
static __global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ in, float* __restrict__ out, int M, int N,
                                                      float scale) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int m = 0; m < M; ++m) s += in[(long)m * N + n];
  out[n] += s * scale;
}

}  // namespace mn
