// The dense layer of the pose head (fc 512 -> feat_dim, /root/reference/models/posenet.py:46,65-66) in fp32, forward, data
// gradient and weight gradient, and the weight gradient of the two pose regressors (:48-49).
//
// These are SMALL problems -- 192 rows, 2048 x 512 weights, 0.4 GFLOP each -- that igemm.h's / wgrad.h's 128 x 128 tiles turn
// into 8 to 32 workgroups each walking K = 512 ... 2048 as one chain of fp32 MFMAs: 27 us (forward), 109 us (data gradient),
// 52 us (weight gradient), plus 57 + 48 us for two column-sum kernels whose threads walked the 192 rows one dependent load at a
// time (round 3 serial profile: ~0.4 ms of a 13.2 ms fp16 step with nothing to overlap it -- the second stream is idle at the head).
// Here the same v_mfma_f32_32x32x2_f32 chains are cut into 32 x 32 tiles with K split over the four waves of a workgroup, so a
// launch is 100 - 400 workgroups of chains 4 x shorter.  Exact fp32 products and sums as before; the order of the sums differs.
#pragma once
#include "common.h"

namespace mn {

struct DenseArgs {
  const float* A;     // [M][lda]
  const float* W;     // [N][ldw]   (K-contiguous rows: nn.Linear's weight, or its transposed copy for the data gradient)
  const float* bias;  // [N] or null
  float* C;           // [M][ldc] = act(A . W^T + bias)
  int M, N, K, lda, ldw, ldc, relu;
};

inline bool dense_nt_applies(const DenseArgs& a) {
  return a.K % 128 == 0 && a.lda % 4 == 0 && a.ldw % 4 == 0 && ((reinterpret_cast<size_t>(a.A) | reinterpret_cast<size_t>(a.W)) & 15) == 0;
}

// One workgroup = one 32 x 32 tile of C; wave w contracts k in [w K/4, (w+1) K/4).  Operands come straight from global memory
// (L2: the whole problem is 5 MB): lane (r, h) loads 16 bytes of row r at k + 4h and MFMA j contracts the pair {k + j, k + 4 + j}
// -- any k order is valid as long as A and W agree (mma_piece<float>, igemm.h).  The four partial tiles are added in wave order.
static __global__ void __launch_bounds__(256) dense_nt_kernel(DenseArgs a) {
  __shared__ float red[3][16][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles_n = (a.N + 31) / 32;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int r = lane & 31, h = lane >> 5;
  const int m = tm * 32 + r, n = tn * 32 + r;
  const bool mv = m < a.M, nv = n < a.N;
  const float* pa = a.A + (long)(mv ? m : 0) * a.lda + 4 * h;
  const float* pw = a.W + (long)(nv ? n : 0) * a.ldw + 4 * h;
  const int kq = a.K / 4, k0 = wave * kq, k1 = k0 + kq;
  floatx16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
  for (int k = k0; k < k1; k += 32) {  // (K % 128 == 0: whole groups of four 8-wide steps, their eight loads in flight together)
    floatx4 va[4], vb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      va[u] = mv ? *reinterpret_cast<const floatx4*>(pa + k + 8 * u) : zero;
      vb[u] = nv ? *reinterpret_cast<const floatx4*>(pw + k + 8 * u) : zero;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va[u][j], vb[u][j], acc, 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave - 1][i][lane] = acc[i];
  }
  __syncthreads();
  if (wave > 0) return;
  const int col = tn * 32 + r;
  const float b = (a.bias && col < a.N) ? a.bias[col] : 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = tm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;  // accumulator layout of the 32 x 32 MFMAs
    float v = ((acc[i] + red[0][i][lane]) + red[1][i][lane]) + red[2][i][lane] + b;
    if (a.relu) v = fmaxf(v, 0.f);
    if (row < a.M && col < a.N) a.C[(long)row * a.ldc + col] = v;
  }
}

inline void launch_dense_nt(const DenseArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(dense_nt_kernel, dim3(((a.M + 31) / 32) * ((a.N + 31) / 32)), dim3(256), 0, s, a);
}

struct DenseWgradArgs {
  const float* dY;  // [B][ldy]  gradient of the layer's output (before the ReLU gate has been applied: the caller's dz is gated)
  const float* X;   // [B][ldx]  the layer's input
  float* dW;        // [F][ldw] += alpha * dY^T . X
  float* db;        // [F]      += alpha * column sums of dY   (or null)
  int B, F, Cin, ldy, ldx, ldw;
  float alpha;
};

// One WAVE = one 32 (output features) x 32 (input features) tile of dW, the batch as K: lane (r, h) of MFMA b/2 takes row
// b + h, so every operand load is two 128-byte row segments.  The waves of input tile 0 also add up their dY values: the bias
// gradient.  Tiles are exclusive, so `+=` needs no atomics.
static __global__ void __launch_bounds__(256) dense_wgrad_kernel(DenseWgradArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles_c = (a.Cin + 31) / 32, tiles_f = (a.F + 31) / 32;
  const int id = blockIdx.x * 4 + wave;
  const int tf = id / tiles_c, tc = id - tf * tiles_c;
  if (tf >= tiles_f) return;  // whole wave
  const int r = lane & 31, h = lane >> 5;
  const int f = tf * 32 + r, c = tc * 32 + r;
  const bool fv = f < a.F, cv = c < a.Cin;
  const float* py = a.dY + (fv ? f : 0);
  const float* px = a.X + (cv ? c : 0);
  floatx16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float colsum = 0.f;
  for (int b = 0; b < a.B; b += 16) {  // eight MFMAs per trip, their sixteen loads in flight together; rows past B read as zero
    float va[8], vb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int bb = b + 2 * u + h;
      const bool bv = bb < a.B;
      va[u] = (bv && fv) ? py[(long)bb * a.ldy] : 0.f;
      vb[u] = (bv && cv) ? px[(long)bb * a.ldx] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      colsum += va[u];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va[u], vb[u], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = tf * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
    if (row < a.F && cv) a.dW[(long)row * a.ldw + c] += a.alpha * acc[i];
  }
  if (a.db && tc == 0) {
    colsum += __shfl_xor(colsum, 32);  // even rows + odd rows
    if (h == 0 && fv) a.db[f] += a.alpha * colsum;
  }
}

inline void launch_dense_wgrad(const DenseWgradArgs& a, hipStream_t s) {
  const int tiles = ((a.F + 31) / 32) * ((a.Cin + 31) / 32);
  hipLaunchKernelGGL(dense_wgrad_kernel, dim3((tiles + 3) / 4), dim3(256), 0, s, a);
}

}  // namespace mn
