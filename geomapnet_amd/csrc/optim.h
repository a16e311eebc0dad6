// Optimiser-side kernels on the flat fp32 parameter arena: global gradient norm, fused
// clip + Adam (L2 weight decay added to the gradient, as torch.optim.Adam; reference call sites
// /root/reference/common/optimizer.py:21-23, common/train.py:357-359), and the repack of the
// fp32 master weights into the compute layouts the matrix-core kernels read.
#pragma once
#include "common.h"

namespace mn {

// accum[0] += sum g[i]^2  (fp64)
static __global__ void __launch_bounds__(256) grad_sqnorm_kernel(const float* __restrict__ g, long n, double* __restrict__ accum) {
  __shared__ double red[4];
  double s = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    double v = g[i];
    s += v * v;
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(accum, red[0] + red[1] + red[2] + red[3]);
}

struct AdamArgs {
  float* p;
  const float* g;
  float* m;
  float* v;
  long n;            // elements updated by this launch
  long n_clip;       // elements [0, n_clip) are scaled by the clip coefficient
  float lr, wd, beta1, beta2, eps;
  float bc1, bc2;    // 1 - beta^step
  float grad_mul;    // e.g. 1/world_size
  float max_norm;    // <= 0: no clipping
  const double* sqnorm;  // accumulated by grad_sqnorm_kernel over the clipped range (after grad_mul)
  const unsigned char* frozen;  // optional per-element freeze mask for the tail [n_clip, n) (criterion scalars)
  int eps_mode;      // 0: torch>=1.0  denom = sqrt(v)/sqrt(bc2)+eps ; 1: torch 0.4.1  denom = sqrt(v)+eps
};

static __global__ void __launch_bounds__(256) adam_kernel(AdamArgs a) {
  float coef = 1.f;
  if (a.max_norm > 0.f) {
    float total = (float)sqrt(*a.sqnorm) * a.grad_mul;
    float c = a.max_norm / (total + 1e-6f);
    if (c < 1.f) coef = c;
  }
  const float sq2 = sqrtf(a.bc2);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long)gridDim.x * blockDim.x) {
    if (a.frozen && i >= a.n_clip && a.frozen[i - a.n_clip]) continue;
    float p = a.p[i];
    float g = a.g[i] * a.grad_mul;
    if (i < a.n_clip) g *= coef;
    if (a.wd != 0.f) g += a.wd * p;
    float m = a.m[i];
    m = m + (g - m) * (1.f - a.beta1);  // lerp form, as torch.optim.Adam
    float v = a.beta2 * a.v[i] + (1.f - a.beta2) * g * g;
    a.m[i] = m;
    a.v[i] = v;
    float upd;
    if (a.eps_mode == 0)
      upd = (a.lr / a.bc1) * (m / (sqrtf(v) / sq2 + a.eps));
    else
      upd = (a.lr * sq2 / a.bc1) * (m / (sqrtf(v) + a.eps));
    a.p[i] = p - upd;
  }
}

// dst[perm(o,r,s,i)] = (T) src[o][r][s][i]   src is the fp32 master in OHWI
//   mode 0: same layout (cast only)
//   mode 1: data-gradient layout [I][R][S][O]
//   mode 2: stem layout [O][R][4][8]: column s' = 2*s4 + (e>>2), channel e&3; zero where s' = 7 or ch = 3
template <typename T>
static __global__ void __launch_bounds__(256) repack_kernel(const float* __restrict__ src, T* __restrict__ dst, int O, int R, int S,
                                                      int I, int mode) {
  long total = (mode == 2) ? (long)O * R * 32 : (long)O * R * S * I;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    if (mode == 0) {
      dst[idx] = (T)src[idx];
    } else if (mode == 1) {
      int o = (int)(idx % O);
      long t = idx / O;
      int s = (int)(t % S);
      t /= S;
      int r = (int)(t % R);
      int i = (int)(t / R);
      dst[idx] = (T)src[(((long)o * R + r) * S + s) * I + i];
    } else {
      int e = (int)(idx % 8);
      long t = idx / 8;
      int s4 = (int)(t % 4);
      t /= 4;
      int r = (int)(t % R);
      int o = (int)(t / R);
      int sp = 2 * s4 + (e >> 2), ch = e & 3;
      float v = (sp < S && ch < I) ? src[(((long)o * R + r) * S + sp) * I + ch] : 0.f;
      dst[idx] = (T)v;
    }
  }
}

// [rows][cols] fp32 -> [cols][rows] fp32 (fc weight for its data-gradient GEMM)
static __global__ void __launch_bounds__(256) transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows,
                                                         int cols) {
  long total = (long)rows * cols;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int r = (int)(idx % rows);
    int c = (int)(idx / rows);
    dst[idx] = src[(long)r * cols + c];
  }
}

// reference-layout import/export: OIHW fp32 (torch state_dict) <-> OHWI fp32 master
static __global__ void __launch_bounds__(256) oihw_ohwi_kernel(const float* __restrict__ src, float* __restrict__ dst, int O, int I,
                                                         int H, int W, int to_ohwi) {
  long total = (long)O * I * H * W;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    // idx enumerates OHWI
    int i = (int)(idx % I);
    long t = idx / I;
    int w = (int)(t % W);
    t /= W;
    int h = (int)(t % H);
    int o = (int)(t / H);
    long oihw = (((long)o * I + i) * H + h) * W + w;
    if (to_ohwi)
      dst[idx] = src[oihw];
    else
      dst[oihw] = src[idx];
  }
}

}  // namespace mn
