// Optimiser-side kernels on the flat fp32 parameter arena: global gradient norm, fused
// clip + Adam (L2 weight decay added to the gradient, as torch.optim.Adam; reference call sites
// /root/reference/common/optimizer.py:21-23, common/train.py:357-359), and the repack of the
// fp32 master weights into the compute layouts the matrix-core kernels read.
#pragma once
#include "common.h"

namespace mn {

// accum[0] += sum g[i]^2  (fp64); with `partials` (MN_DETERMINISTIC) the workgroup's sum goes to partials[blockIdx.x] instead
// and sqnorm_fold_kernel adds those up in index order
static __global__ void __launch_bounds__(256) grad_sqnorm_kernel(const float* __restrict__ g, long n, double* __restrict__ accum,
                                                                 double* __restrict__ partials = nullptr) {
  __shared__ double red[4];
  double s = 0;
  // 16-byte loads over the aligned body (a 16-byte aligned `g`: sqnorm_grid), the last n % 4 elements one by one
  const long n4 = ((reinterpret_cast<size_t>(g) & 15) == 0) ? n / 4 : 0;
  const floatx4* g4 = reinterpret_cast<const floatx4*>(g);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const floatx4 q = g4[i];
    const double a = q.x, b = q.y, c = q.z, d = q.w;
    s += (a * a + b * b) + (c * c + d * d);
  }
  for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    double v = g[i];
    s += v * v;
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double v = red[0] + red[1] + red[2] + red[3];
    if (partials)
      partials[blockIdx.x] = v;
    else
      atomicAdd(accum, v);
  }
}
// workgroups of a grad_sqnorm_kernel launch: two per CU -- every one ends in an atomic add onto ONE fp64 word (or one partial), and
// with 4-byte loads from 4096 workgroups the launch took 63 us for 84 MB (round 4: 16-byte loads, 512 workgroups)
inline int sqnorm_grid(long n) {
  const long b = (n / 4 + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));
}
// one workgroup: thread t adds partials t, t + 256, ... in order, the 256 sums are added in thread order
static __global__ void __launch_bounds__(256) sqnorm_fold_kernel(const double* __restrict__ partials, int n, double* __restrict__ accum) {
  __shared__ double red[256];
  double s = 0;
  for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = 0;
    for (int i = 0; i < 256; ++i) v += red[i];
    *accum = v;
  }
}

// optim.learner.zero_grad(): 16-byte stores over the gradient arena (hipMemsetAsync's fill kernel took ~240 us for the
// 89 MB arena, 0.37 TB/s; this one runs at the store bandwidth)
static __global__ void __launch_bounds__(256) zero_fill_kernel(floatx4* __restrict__ p, long n4, float* __restrict__ tail, int ntail) {
  const floatx4 z = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) p[i] = z;
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0.f;
}
inline void launch_zero_fill(float* p, long n, hipStream_t s) {  // p 16-byte aligned
  const long n4 = n / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(zero_fill_kernel, dim3((int)blocks), dim3(256), 0, s, reinterpret_cast<floatx4*>(p), n4, p + n4 * 4, (int)(n - n4 * 4));
}

// Data-parallel gradient transport in bf16 (geomapnet_amd/dp.py, MN_DP_GRAD_DTYPE=bf16): a bucket of the fp32 gradient arena is
// rounded to bf16 for the all-reduce (half the bytes on xGMI, half the time RCCL's workgroups sit on the CUs) and widened back in
// place.  bf16, not fp16: weight gradients are sums over up to a million pixels and reach O(100) at random init -- times the loss
// scale they leave fp16's range (measured: 144 NaNs in the stem + layer1 bucket of the first 2-rank test), and their small entries
// would underflow without it; bf16 has fp32's exponent range, so there is no scale to carry, at 8 bits of mantissa per rank.
static __global__ void __launch_bounds__(256) grad_pack_bf16_kernel(const float* __restrict__ g, __bf16* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (__bf16)g[i];
}
static __global__ void __launch_bounds__(256) grad_unpack_bf16_kernel(const __bf16* __restrict__ in, float* __restrict__ g, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) g[i] = (float)in[i];
}
inline int grad_transport_grid(long n) {
  long b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
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
  const float* bc_dev;  // optional device copy of {bc1, bc2} (written by adam_prep_kernel); overrides bc1/bc2
  const long long* skip = nullptr;  // optional overflow word (adam_prep_kernel): non-zero = leave everything untouched
  // method (common/optimizer.py:16-26: torch.optim.SGD / Adam / RMSprop): 0 = Adam; 1 = SGD with `momentum` (beta1),
  // `dampening` (beta2), `nesterov` -- momentum buffer in m; 2 = RMSprop (not centered) with `alpha` (beta2), eps,
  // `momentum` (beta1) -- square average in v, momentum buffer in m
  int method = 0;
  int nesterov = 0;
  int first_step = 0;  // SGD: this is the first step (the momentum buffer starts as the gradient); bc_dev[2] overrides
};

// Advances the device-resident step counter and derives the bias corrections from it, so a captured
// (hipGraph) training step needs no per-step host arguments.
// Overflow guard (fp16 steps, `overflow` non-null): the squared gradient norm is non-finite exactly when some
// gradient element is (inf^2 = inf, NaN propagates) -- one fp16 activation gradient that overflowed reaches every
// weight gradient below it.  Such a step must not touch the fp32 master weights or the Adam moments (the reference
// trains in fp32 and cannot overflow; with clipping on, inf * coef(=0) would even turn into NaN): overflow[0] = 1 tells
// adam_kernel to return, the step counter is not advanced, and overflow[1] counts the skipped steps (the host halves
// the loss scale when it sees the count move, net.hip).
// `attempt`: the host's count of enqueued optimiser steps; overflow[2] remembers the last attempt that was skipped, so the
// host can tell a skip of a step enqueued BEFORE it lowered the loss scale from one that happened under the new scale.
static __global__ void adam_prep_kernel(long long* step, float beta1, float beta2, float* bc, const double* sqnorm,
                                        long long* overflow, long long attempt = 0) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (overflow) {
      const double v = *sqnorm;
      const bool bad = !(v == v) || v > 1.7e308 || v < -1.7e308;
      overflow[0] = bad ? 1 : 0;
      overflow[3] = attempt;  // the last attempt the device has COMPLETED (applied or skipped): the host's evidence of progress
      if (bad) {
        overflow[1] += 1;
        overflow[2] = attempt;
        return;
      }
    }
    const long long t = *step + 1;
    *step = t;
    bc[0] = (float)(1.0 - pow((double)beta1, (double)t));
    bc[1] = (float)(1.0 - pow((double)beta2, (double)t));
    bc[2] = t == 1 ? 1.f : 0.f;  // first step (SGD's momentum buffer)
  }
}

static __global__ void __launch_bounds__(256) adam_kernel(AdamArgs a) {
  if (a.skip && *a.skip) return;  // overflowed step: parameters, moments and step counter stay as they are
  float coef = 1.f;
  if (a.max_norm > 0.f) {
    float total = (float)sqrt(*a.sqnorm) * a.grad_mul;
    float c = a.max_norm / (total + 1e-6f);
    if (c < 1.f) coef = c;
  }
  const float bc1 = a.bc_dev ? a.bc_dev[0] : a.bc1, bc2 = a.bc_dev ? a.bc_dev[1] : a.bc2;
  const float sq2 = sqrtf(bc2);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long)gridDim.x * blockDim.x) {
    if (a.frozen && i >= a.n_clip && a.frozen[i - a.n_clip]) continue;
    float p = a.p[i];
    float g = a.g[i] * a.grad_mul;
    if (i < a.n_clip) g *= coef;
    if (a.wd != 0.f) g += a.wd * p;
    if (a.method == 1) {  // torch.optim.SGD: buf = g on the first step, else momentum * buf + (1 - dampening) * g
      float d = g;
      if (a.beta1 != 0.f) {
        const bool first = a.bc_dev ? a.bc_dev[2] != 0.f : a.first_step != 0;
        const float buf = first ? g : a.beta1 * a.m[i] + (1.f - a.beta2) * g;
        a.m[i] = buf;
        d = a.nesterov ? g + a.beta1 * buf : buf;
      }
      a.p[i] = p - a.lr * d;
      continue;
    }
    if (a.method == 2) {  // torch.optim.RMSprop (centered = False)
      const float sq = a.beta2 * a.v[i] + (1.f - a.beta2) * g * g;
      a.v[i] = sq;
      const float avg = sqrtf(sq) + a.eps;
      if (a.beta1 > 0.f) {
        const float buf = a.beta1 * a.m[i] + g / avg;
        a.m[i] = buf;
        a.p[i] = p - a.lr * buf;
      } else {
        a.p[i] = p - a.lr * (g / avg);
      }
      continue;
    }
    float m = a.m[i];
    m = m + (g - m) * (1.f - a.beta1);  // lerp form, as torch.optim.Adam
    float v = a.beta2 * a.v[i] + (1.f - a.beta2) * g * g;
    a.m[i] = m;
    a.v[i] = v;
    float upd;
    if (a.eps_mode == 0)
      upd = (a.lr / bc1) * (m / (sqrtf(v) / sq2 + a.eps));
    else
      upd = (a.lr * sq2 / bc1) * (m / (sqrtf(v) + a.eps));
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

// All compute-layout copies of one optimiser step in ONE launch: a device-resident job table maps
// workgroup ranges to (conv weight -> forward copy, data-gradient copy | stem layout | fc transpose).
struct RepackJob {
  long src_off;   // offset of the fp32 master in the parameter arena
  void* dst_a;    // mode 0: forward copy (same layout, cast) or null; mode 2: stem layout; mode 3: transpose
  void* dst_b;    // mode 0: data-gradient layout [I][R][S][O] or null
  int O, R, S, I;
  int mode;       // 0 conv, 2 stem, 3 fp32 transpose [O][I] -> [I][O]
  int blk0, nblk; // workgroup range; each workgroup covers 4096 elements
};

// one weight element -> its hi and lo halves at (row, c) of an h2 matrix with C columns (common.h)
__device__ __forceinline__ void h2_put(void* base, long row, int C, int c, float v) {
  half* p = reinterpret_cast<half*>(base) + h2_index(row, C, c);
  const half hi = (half)v;
  p[0] = hi;
  p[32] = (half)(v - (float)hi);
}
// one weight element -> (row, c) of an h2q WEIGHT matrix (common.h): fp16 hi half where h2 has it, then the fp8 planes hi8 | lo8
__device__ __forceinline__ void h2q_put_w(void* base, long row, int C, int c, float v) {
  half* p = reinterpret_cast<half*>(base) + h2_index(row, C, c);
  const half hi = (half)v;
  p[0] = hi;
  unsigned char* q = reinterpret_cast<unsigned char*>(base) + h2q_byte(row, C, c, 0);
  q[0] = fp8_e4m3_from_float((float)hi * pow2c(kQW_HI));
  q[32] = fp8_e4m3_from_float((v - (float)hi) * pow2c(kQW_LO));
}
// H2 (T = float): mode-0 destinations are h2 matrices -- the forward operand [O][R][S][I] split along I, the data-gradient
// operand [I][R][S][O] along O (MN_DTYPE_F16X2); modes 2 and 3 (stem, fc) stay fp32.  B16 (with H2, the fp16x2m mode): the
// data-gradient operand is a plain fp16 matrix -- the backward pass contracts single fp16 operands
// AQ (with H2 and B16, the fp16x2q mode): the forward operand is an h2q weight matrix
template <typename T, bool H2 = false, bool B16 = false, bool AQ = false>
static __global__ void __launch_bounds__(256) repack_all_kernel(const RepackJob* __restrict__ jobs, int njobs,
                                                                const float* __restrict__ params, int blk_base) {
  const int bid = (int)blockIdx.x + blk_base;  // a launch may cover a sub-range of the table's workgroups
  int j = 0;
  while (j + 1 < njobs && bid >= jobs[j + 1].blk0) ++j;
  const RepackJob job = jobs[j];
  const float* src = params + job.src_off;
  const int O = job.O, R = job.R, S = job.S, I = job.I;
  const long total = job.mode == 2 ? (long)O * R * 32 : (long)O * R * S * I;
  // Transposing copies (the data-gradient layout [I][R][S][O] of a conv weight, the fc weight's [I][O]) go through a
  // 64 x 64 LDS tile per workgroup when O and I are multiples of 64 (every ResNet-34 layer): 256-byte reads along I,
  // 128-byte writes along O, instead of one scattered 2-byte write per element.  Same 4096 elements per workgroup, so
  // the job table's workgroup ranges are unchanged.
  __shared__ float tile[64][65];
  if ((job.mode == 0 || job.mode == 3) && O % 64 == 0 && I % 64 == 0) {
    const int RS = R * S, tiles_i = I / 64;
    const int local = bid - job.blk0;
    const int it = local % tiles_i, tmp = local / tiles_i;
    const int rs = tmp % RS, ot = tmp / RS;
    const int col = threadIdx.x & 63, row0 = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int row = k * 4 + row0;
      const long idx = ((long)(ot * 64 + row) * RS + rs) * I + it * 64 + col;
      const float v = src[idx];
      if (job.mode == 0 && job.dst_a) {
        if constexpr (H2 && AQ)
          h2q_put_w(job.dst_a, (long)(ot * 64 + row) * RS + rs, I, it * 64 + col, v);
        else if constexpr (H2)
          h2_put(job.dst_a, (long)(ot * 64 + row) * RS + rs, I, it * 64 + col, v);
        else
          reinterpret_cast<T*>(job.dst_a)[idx] = (T)v;
      }
      tile[row][col] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int irow = k * 4 + row0;
      const float v = tile[col][irow];
      const long didx = ((long)(it * 64 + irow) * RS + rs) * O + ot * 64 + col;
      if (job.mode == 0) {
        if constexpr (H2 && B16)
          reinterpret_cast<half*>(job.dst_b)[didx] = (half)v;
        else if constexpr (H2)
          h2_put(job.dst_b, (long)(it * 64 + irow) * RS + rs, O, ot * 64 + col, v);
        else
          reinterpret_cast<T*>(job.dst_b)[didx] = (T)v;
      } else {
        reinterpret_cast<float*>(job.dst_a)[didx] = v;
      }
    }
    return;
  }
  const long base = (long)(bid - job.blk0) * 4096;
  for (int k = 0; k < 16; ++k) {
    const long idx = base + k * 256 + threadIdx.x;
    if (idx >= total) break;
    if (job.mode == 0) {
      const float v = src[idx];
      // idx = ((o*R + r)*S + s)*I + i  ->  ((i*R + r)*S + s)*O + o
      const int i = (int)(idx % I);
      long t = idx / I;
      const int s = (int)(t % S);
      t /= S;
      const int r = (int)(t % R);
      const int o = (int)(t / R);
      if constexpr (H2) {
        if (job.dst_a) {
          if constexpr (AQ)
            h2q_put_w(job.dst_a, idx / I, I, i, v);
          else
            h2_put(job.dst_a, idx / I, I, i, v);
        }
        if constexpr (B16)
          reinterpret_cast<half*>(job.dst_b)[(((long)i * R + r) * S + s) * O + o] = (half)v;
        else
          h2_put(job.dst_b, ((long)i * R + r) * S + s, O, o, v);
      } else {
        if (job.dst_a) reinterpret_cast<T*>(job.dst_a)[idx] = (T)v;
        reinterpret_cast<T*>(job.dst_b)[(((long)i * R + r) * S + s) * O + o] = (T)v;
      }
    } else if (job.mode == 2) {
      const int e = (int)(idx % 8);
      long t = idx / 8;
      const int s4 = (int)(t % 4);
      t /= 4;
      const int r = (int)(t % R);
      const int o = (int)(t / R);
      const int sp = 2 * s4 + (e >> 2), ch = e & 3;
      const float v = (sp < S && ch < I) ? src[(((long)o * R + r) * S + sp) * I + ch] : 0.f;
      reinterpret_cast<T*>(job.dst_a)[idx] = (T)v;
    } else {
      const int i = (int)(idx % I);
      const int o = (int)(idx / I);
      reinterpret_cast<float*>(job.dst_a)[(long)i * O + o] = src[idx];
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
