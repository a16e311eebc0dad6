// HBM-bound kernels of the MapNet training step: layout conversion, BatchNorm statistics /
// apply / backward, max-pool, average-pool.  All activations are NHWC with C a multiple of the
// 16-byte piece (8 halves / 4 floats); every global access is one 16-byte piece per lane.
// BatchNorm semantics follow torch.nn.BatchNorm2d in training mode: biased
// batch variance for normalisation, eps inside the sqrt, running stats updated with momentum
// using the unbiased variance.
#pragma once
#include "common.h"

namespace mn {

inline int ew_grid(long work_items) {
  long b = (work_items + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

// ---- input: NCHW fp32 -> zero-padded NHWC4 (pad 3 top/left, >=3 bottom/right) -------------------
template <typename T>
static __global__ void __launch_bounds__(256) nchw_to_padded_nhwc4_kernel(const float* __restrict__ in, T* __restrict__ out,
                                                                    int B, int H, int W, int Hp, int Wp) {
  long total = (long)B * Hp * Wp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int wp = (int)(i % Wp);
    long tmp = i / Wp;
    int hp = (int)(tmp % Hp);
    int b = (int)(tmp / Hp);
    int h = hp - 3, w = wp - 3;
    T v[4] = {(T)0.f, (T)0.f, (T)0.f, (T)0.f};
    if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
      const float* p = in + ((long)b * 3 * H + h) * W + w;
      v[0] = (T)p[0];
      v[1] = (T)p[(long)H * W];
      v[2] = (T)p[2L * H * W];
    }
    T* o = out + i * 4;
    o[0] = v[0];
    o[1] = v[1];
    o[2] = v[2];
    o[3] = v[3];
  }
}

// ---- BatchNorm statistics -------------------------------------------------------------------------
// stage 1: fold the conv epilogue's per-block partials [GM][2][N] into fp64 accumulators [2][N]
static __global__ void __launch_bounds__(256) bn_reduce_partials_kernel(const float* __restrict__ partial, int GM, int N,
                                                                  double* __restrict__ accum, int rows_per_block) {
  __shared__ double red[2][4][64];
  const int c = blockIdx.y * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(GM, r0 + rows_per_block);
  double s1 = 0, s2 = 0;
  if (c < N)
    for (int r = r0 + rl; r < r1; r += 4) {
      s1 += partial[((long)r * 2 + 0) * N + c];
      s2 += partial[((long)r * 2 + 1) * N + c];
    }
  red[0][rl][threadIdx.x & 63] = s1;
  red[1][rl][threadIdx.x & 63] = s2;
  __syncthreads();
  if (rl == 0 && c < N) {
    int l = threadIdx.x;
    atomicAdd(accum + c, red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l]);
    atomicAdd(accum + N + c, red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l]);
  }
}

struct BnParams {
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  long long* num_batches_tracked;  // may be null
  float* mean;     // saved batch mean
  float* invstd;   // saved 1/sqrt(var+eps)
  float* scale;    // gamma * invstd
  float* shift;    // beta - mean * scale
  float eps, momentum;
};

// stage 2: statistics -> (mean, invstd, scale, shift), running-stat update, accumulator reset
static __global__ void __launch_bounds__(256) bn_finalize_kernel(double* __restrict__ accum, int N, double count, BnParams p,
                                                           int training) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  float mean, var;
  if (training) {
    double m = accum[c] / count;
    double v = accum[N + c] / count - m * m;
    if (v < 0) v = 0;
    accum[c] = 0;
    accum[N + c] = 0;
    mean = (float)m;
    var = (float)v;
    double unbiased = count > 1 ? v * count / (count - 1) : v;
    p.running_mean[c] = (1.f - p.momentum) * p.running_mean[c] + p.momentum * mean;
    p.running_var[c] = (1.f - p.momentum) * p.running_var[c] + p.momentum * (float)unbiased;
    if (c == 0 && p.num_batches_tracked) *p.num_batches_tracked += 1;
  } else {
    mean = p.running_mean[c];
    var = p.running_var[c];
  }
  float invstd = 1.0f / sqrtf(var + p.eps);
  p.mean[c] = mean;
  p.invstd[c] = invstd;
  float sc = p.gamma[c] * invstd;
  p.scale[c] = sc;
  p.shift[c] = p.beta[c] - mean * sc;
}

// out = [relu]( y * scale[c] + shift[c] [+ res] ), one piece per thread-iteration
template <typename T>
static __global__ void __launch_bounds__(256) bn_apply_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const T* __restrict__ res,
                                                        T* __restrict__ out, long npieces, int C, int relu) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int cpr = C / VEC;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npieces; i += (long)gridDim.x * blockDim.x) {
    int c0 = (int)(i % cpr) * VEC;
    PieceView<T> v, r, o;
    v.p = reinterpret_cast<const piece_t*>(y)[i];
    if (res) r.p = reinterpret_cast<const piece_t*>(res)[i];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float f = (float)v.e[e] * scale[c0 + e] + shift[c0 + e];
      if (res) f += (float)r.e[e];
      if (relu) f = fmaxf(f, 0.f);
      o.e[e] = (T)f;
    }
    reinterpret_cast<piece_t*>(out)[i] = o.p;
  }
}

// ---- BatchNorm backward ---------------------------------------------------------------------------
// reduce: accum[0][c] += sum gm, accum[1][c] += sum gm * xhat,  gm = g * (gate > 0 if gate)
template <typename T>
static __global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const T* __restrict__ g, const T* __restrict__ gate,
                                                             const T* __restrict__ y, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, long M, int C,
                                                             double* __restrict__ accum, int rows_per_block) {
  constexpr int VEC = ElemTraits<T>::VEC;
  __shared__ float red[2][256][VEC];
  const int cpr = C / VEC;           // pieces per row (power of two, <= 256)
  const int rlanes = 256 / cpr;
  const int cp = threadIdx.x % cpr, rl = threadIdx.x / cpr;
  const int c0 = cp * VEC;
  float mu[VEC], is[VEC], s1[VEC], s2[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    mu[e] = mean[c0 + e];
    is[e] = invstd[c0 + e];
    s1[e] = s2[e] = 0.f;
  }
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  // four rows per iteration, all loads issued before the arithmetic (memory-level parallelism)
  constexpr int U = 4;
  for (long r = r0 + rl; r < r1; r += (long)U * rlanes) {
    PieceView<T> vg[U], vy[U], vm[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long rr = r + (long)u * rlanes;
      const bool in = rr < r1;
      const long idx = (in ? rr : r) * cpr + cp;
      vg[u].p = in ? reinterpret_cast<const piece_t*>(g)[idx] : zero_piece();
      vy[u].p = reinterpret_cast<const piece_t*>(y)[idx];
      if (gate) vm[u].p = reinterpret_cast<const piece_t*>(gate)[idx];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float gv = (float)vg[u].e[e];
        if (gate && !((float)vm[u].e[e] > 0.f)) gv = 0.f;
        s1[e] += gv;
        s2[e] += gv * ((float)vy[u].e[e] - mu[e]) * is[e];
      }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    red[0][threadIdx.x][e] = s1[e];
    red[1][threadIdx.x][e] = s2[e];
  }
  __syncthreads();
  // thread (cp, e-slot) sums over row lanes: reuse threads 0 .. cpr*VEC-1 = C
  for (int idx = threadIdx.x; idx < C; idx += 256) {
    int p = idx / VEC, e = idx % VEC;
    double a = 0, b = 0;
    for (int l = 0; l < rlanes; ++l) {
      a += red[0][l * cpr + p][e];
      b += red[1][l * cpr + p][e];
    }
    atomicAdd(accum + idx, a);
    atomicAdd(accum + C + idx, b);
  }
}

// finalize: dgamma, dbeta (unscaled by 1/loss_scale), per-channel coefficients; reset accumulators
static __global__ void __launch_bounds__(256) bn_bwd_finalize_kernel(double* __restrict__ accum, int C, double count,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, float* __restrict__ k1,
                                                               float* __restrict__ mg, float* __restrict__ mgx,
                                                               float grad_unscale) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double sg = accum[c], sgx = accum[C + c];
  accum[c] = 0;
  accum[C + c] = 0;
  dgamma[c] += (float)(sgx * grad_unscale);
  dbeta[c] += (float)(sg * grad_unscale);
  k1[c] = gamma[c] * invstd[c];
  mg[c] = (float)(sg / count);
  mgx[c] = (float)(sgx / count);
}

// apply: gy = k1 * (gm - mg - xhat * mgx)
template <typename T>
static __global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ gate,
                                                            const T* __restrict__ y, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ k1,
                                                            const float* __restrict__ mg, const float* __restrict__ mgx,
                                                            T* __restrict__ gy, long npieces, int C) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int cpr = C / VEC;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npieces; i += (long)gridDim.x * blockDim.x) {
    int c0 = (int)(i % cpr) * VEC;
    PieceView<T> vg, vy, vm, o;
    vg.p = reinterpret_cast<const piece_t*>(g)[i];
    vy.p = reinterpret_cast<const piece_t*>(y)[i];
    if (gate) vm.p = reinterpret_cast<const piece_t*>(gate)[i];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      int c = c0 + e;
      float gv = (float)vg.e[e];
      if (gate && !((float)vm.e[e] > 0.f)) gv = 0.f;
      float xh = ((float)vy.e[e] - mean[c]) * invstd[c];
      o.e[e] = (T)(k1[c] * (gv - mg[c] - xh * mgx[c]));
    }
    reinterpret_cast<piece_t*>(gy)[i] = o.p;
  }
}

// direct statistics pass (standalone BatchNorm operator; the network takes its statistics
// from the conv epilogue instead): accum[0][c] += sum y, accum[1][c] += sum y^2
template <typename T>
static __global__ void __launch_bounds__(256) bn_fwd_stats_kernel(const T* __restrict__ y, long M, int C,
                                                            double* __restrict__ accum, int rows_per_block) {
  constexpr int VEC = ElemTraits<T>::VEC;
  __shared__ float red[2][256][VEC];
  const int cpr = C / VEC;
  const int rlanes = 256 / cpr;
  const int cp = threadIdx.x % cpr, rl = threadIdx.x / cpr;
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) s1[e] = s2[e] = 0.f;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  for (long r = r0 + rl; r < r1; r += rlanes) {
    PieceView<T> vy;
    vy.p = reinterpret_cast<const piece_t*>(y)[r * cpr + cp];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float f = (float)vy.e[e];
      s1[e] += f;
      s2[e] += f * f;
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    red[0][threadIdx.x][e] = s1[e];
    red[1][threadIdx.x][e] = s2[e];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < C; idx += 256) {
    int p = idx / VEC, e = idx % VEC;
    double a = 0, b = 0;
    for (int l = 0; l < rlanes; ++l) {
      a += red[0][l * cpr + p][e];
      b += red[1][l * cpr + p][e];
    }
    atomicAdd(accum + idx, a);
    atomicAdd(accum + C + idx, b);
  }
}

// BatchNorm backward = reduce -> finalize -> apply.  coef: [3][C] floats (k1, mg, mgx);
// accum: [2][C] doubles, zero on entry, zero again on exit.
template <typename T>
inline void launch_bn_bwd(const T* g, const T* gate, const T* y, long M, int C, const float* gamma, const float* mean,
                          const float* invstd, float* dgamma, float* dbeta, T* gy, float* coef, double* accum,
                          float grad_unscale, hipStream_t s) {
  constexpr int VEC = ElemTraits<T>::VEC;
  // ~4096 workgroups in flight: the reduction is HBM-bound and needs the whole chip
  const int rlanes = 256 / (C / VEC);
  long rows = (M + 4095) / 4096;
  rows = ((rows + rlanes - 1) / rlanes) * rlanes;
  if (rows < 4L * rlanes) rows = 4L * rlanes;
  int rows_per_block = (int)rows;
  hipLaunchKernelGGL((bn_bwd_reduce_kernel<T>), dim3(cdiv(M, rows_per_block)), dim3(256), 0, s, g, gate, y, mean, invstd, M,
                     C, accum, rows_per_block);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, accum, C, (double)M, gamma, invstd,
                     dgamma, dbeta, coef, coef + C, coef + 2 * C, grad_unscale);
  long np = M * C / VEC;
  hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(ew_grid(np)), dim3(256), 0, s, g, gate, y, mean, invstd,
                     (const float*)coef, (const float*)(coef + C), (const float*)(coef + 2 * C), gy, np, C);
}

// ---- global average pool --------------------------------------------------------------------------
template <typename T>
static __global__ void __launch_bounds__(256) avgpool_fwd_kernel(const T* __restrict__ in, float* __restrict__ out, int B, int HW,
                                                           int C) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  int c = i % C, b = i / C;
  float s = 0.f;
  for (int p = 0; p < HW; ++p) s += (float)in[((long)b * HW + p) * C + c];
  out[i] = s / (float)HW;
}

// g[b][p][c] = gp[b][c] / HW
template <typename T>
static __global__ void __launch_bounds__(256) avgpool_bwd_kernel(const float* __restrict__ gp, T* __restrict__ g, int B, int HW,
                                                           int C) {
  long total = (long)B * HW * C;
  float inv = 1.f / (float)HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    int b = (int)(i / ((long)HW * C));
    g[i] = (T)(gp[b * C + c] * inv);
  }
}

}  // namespace mn
