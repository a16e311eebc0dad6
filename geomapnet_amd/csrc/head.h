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

// ---- dropout on the feature vector (models/posenet.py:68-69: F.dropout(x, p=droprate) between the ReLU and the pose heads) ----
// Philox4x32-10 (Salmon et al. 2011), counter = (element / 4, call, 0, 0), key = seed: element i of call c always draws the
// same number whatever the launch shape, so a mask can be reproduced from (seed, c) alone.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// feat[i] *= mask[i], mask[i] = (u_i >= p) / (1 - p): inverted dropout as torch.nn.functional.dropout; the mask (0 or 1/(1-p))
// is kept for the backward pass and for inspection (mn_debug_tensor "dropmask")
static __global__ void __launch_bounds__(256) dropout_fwd_kernel(float* __restrict__ feat, float* __restrict__ mask, long n, float p,
                                                                 unsigned seed_lo, unsigned seed_hi, unsigned call) {
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;  // four elements per thread: one Philox block
  if (q * 4 >= n) return;
  unsigned r[4];
  philox4x32_10((unsigned)q, call, 0u, 0u, seed_lo, seed_hi, r);
  const float keep = 1.f / (1.f - p);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const long i = q * 4 + e;
    if (i >= n) break;
    const float u = (float)(r[e] >> 8) * (1.f / 16777216.f);  // 24 random bits -> [0, 1)
    const float m = u >= p ? keep : 0.f;
    mask[i] = m;
    feat[i] *= m;
  }
}

// dz[b][k] = (feat[b][k] > 0) * ( sum_o dp[b][o] Wx[o][k]  +  filt( sum_o dp[b][3+o] Wq[o][k] ) )
// (feat is the post-ReLU feature, so the gate is the ReLU derivative of posenet.py:66)
static __global__ void __launch_bounds__(256) head_bwd_input_kernel(const float* __restrict__ dposes,
                                                              const float* __restrict__ feat,
                                                              const float* __restrict__ Wx, const float* __restrict__ Wq,
                                                              float* __restrict__ dz, int B, int K, int filter_nans,
                                                              const float* __restrict__ dropmask = nullptr) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * K) return;
  int k = (int)(i % K), b = (int)(i / K);
  const float* dp = dposes + b * 6;
  float gx = (dp[0] * Wx[k] + dp[1] * Wx[K + k]) + dp[2] * Wx[2 * K + k];
  float gq = (dp[3] * Wq[k] + dp[4] * Wq[K + k]) + dp[5] * Wq[2 * K + k];
  if (filter_nans) gq = nan_to_zero(gq);
  float g = gx + gq;
  // with dropout `feat` is the dropped vector: a dropped element is 0 (gate closed), a kept one keeps its sign, and the
  // gradient w.r.t. the vector before the dropout is g * mask
  if (dropmask) g *= dropmask[i];
  dz[i] = (feat[i] > 0.f) ? g : 0.f;
}

// dW[o][k] += scale * sum_b dp[b][o] feat[b][k] ; db[o] += scale * sum_b dp[b][o]   (o = 0..5)
// One workgroup = 64 columns k (column K = the bias) x 4 row groups: wave w adds rows w, w + 4, ... for all six outputs at once
// (one load of feat per row, the six dp values of a row are wave-uniform), the four partial sums are added in wave order.
// (Round 3 form: one thread per (o, k) walking all B rows -- 192 dependent steps, 57 us for 2.4 MFLOP.)
static __global__ void __launch_bounds__(256) head_bwd_weight_kernel(const float* __restrict__ dposes,
                                                               const float* __restrict__ feat, float* __restrict__ dWx,
                                                               float* __restrict__ dbx, float* __restrict__ dWq,
                                                               float* __restrict__ dbq, int B, int K, float scale,
                                                               int filter_nans) {
  __shared__ float red[3][6][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k = blockIdx.x * 64 + lane;
  const bool is_w = k < K;
  float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (k <= K) {
#pragma unroll 4
    for (int b = wave; b < B; b += 4) {
      const float f = is_w ? feat[(long)b * K + k] : 1.f;
#pragma unroll
      for (int o = 0; o < 6; ++o) s[o] += dposes[b * 6 + o] * f;
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int o = 0; o < 6; ++o) red[wave - 1][o][lane] = s[o];
  }
  __syncthreads();
  if (wave > 0 || k > K) return;
#pragma unroll
  for (int o = 0; o < 6; ++o) {
    float v = (((s[o] + red[0][o][lane]) + red[1][o][lane]) + red[2][o][lane]) * scale;
    if (o >= 3 && filter_nans) v = nan_to_zero(v);
    float* dst = is_w ? ((o < 3) ? dWx + (long)o * K + k : dWq + (long)(o - 3) * K + k) : ((o < 3) ? dbx + o : dbq + (o - 3));
    *dst += v;
  }
}

}  // namespace mn
