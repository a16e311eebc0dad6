// HBM-bound kernels of the MapNet training step: layout conversion, BatchNorm statistics /
// apply / backward, max-pool, average-pool.  All activations are NHWC with C a multiple of the
// 16-byte piece (8 halves / 4 floats); every global access is one 16-byte piece per lane.
// BatchNorm semantics follow torch.nn.BatchNorm2d in training mode: biased
// batch variance for normalisation, eps inside the sqrt, running stats updated with momentum
// using the unbiased variance.
#pragma once
#include <stdlib.h>
#include "common.h"
#include "pool.h"

namespace mn {

inline int ew_grid(long work_items) {
  // grid-stride kernels: at most 16 workgroups per CU.  (Measured and removed, same-box A/Bs of the whole step in
  // profiles/r02/c25_*: fewer workgroups per CU -- to leave wave slots to the weight gradients running beside the
  // HBM-bound passes -- and several pieces per thread for the small tensors of layers 3-4: no gain.)
  static const long cap = getenv("MN_EW_WGS") ? atol(getenv("MN_EW_WGS")) : 256L * 16;  // (A/B knob)
  long b = (work_items + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

// ---- input: NCHW fp32 -> zero-padded NHWC4 (pad 3 top/left, >=3 bottom/right) -------------------
// out16 (optional, fp16x2m: the stem's fp16 backward kernels read an fp16 image of the input): the same pixels once more as fp16, from
// the same loads (round 5 launched the conversion twice)
template <typename T>
static __global__ void __launch_bounds__(256) nchw_to_padded_nhwc4_kernel(const float* __restrict__ in, T* __restrict__ out,
                                                                    int B, int H, int W, int Hp, int Wp, half* __restrict__ out16) {
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
    if (out16) {
      half4 h = {(half)(float)v[0], (half)(float)v[1], (half)(float)v[2], (half)0.f};
      *reinterpret_cast<half4*>(out16 + i * 4) = h;
    }
  }
}

// ---- input: uint8 NHWC (decoded image bytes) -> normalised, zero-padded NHWC4 ---------------------------
// torchvision's ToTensor + Normalize on the device: x = u8 / 255, (x - mean[c]) / std[c], as one
// multiply-add per element (scale = 1/(255 std), shift = -mean/std).  Reads 3 B/pixel instead of 12.
struct InputNorm {
  float scale[3], shift[3];
};
template <typename T>
static __global__ void __launch_bounds__(256) u8nhwc_to_padded_nhwc4_kernel(const unsigned char* __restrict__ in,
                                                                      T* __restrict__ out, int B, int H, int W, int Hp,
                                                                      int Wp, InputNorm nm, half* __restrict__ out16) {
  long total = (long)B * Hp * Wp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int wp = (int)(i % Wp);
    long tmp = i / Wp;
    int hp = (int)(tmp % Hp);
    int b = (int)(tmp / Hp);
    int h = hp - 3, w = wp - 3;
    T v[4] = {(T)0.f, (T)0.f, (T)0.f, (T)0.f};
    if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
      const unsigned char* p = in + (((long)b * H + h) * W + w) * 3;
      v[0] = (T)((float)p[0] * nm.scale[0] + nm.shift[0]);
      v[1] = (T)((float)p[1] * nm.scale[1] + nm.shift[1]);
      v[2] = (T)((float)p[2] * nm.scale[2] + nm.shift[2]);
    }
    T* o = out + i * 4;
    o[0] = v[0];
    o[1] = v[1];
    o[2] = v[2];
    o[3] = v[3];
    if (out16) {
      half4 h = {(half)(float)v[0], (half)(float)v[1], (half)(float)v[2], (half)0.f};
      *reinterpret_cast<half4*>(out16 + i * 4) = h;
    }
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
  float eps, momentum;
};

// Statistics -> per-channel (scale, shift): ONE small launch per BatchNorm (bn_finalize_fwd_kernel) turns the fp64
// accumulator rows into coef[0][c] = gamma * invstd, coef[1][c] = beta - mean * scale, publishes mean / invstd for the
// backward pass and updates the running statistics; the streaming kernels below only load those two floats per channel.
// (Round 1 derived them in the prologue of every workgroup of the apply kernel: 16 fp64 loads per channel in each of
// up to 4096 workgroups -- for the late layers, whose tensors are 17-34 MB, that prologue was as long as the streaming.)
// The accumulators are read-only here; the caller zeroes them before the next accumulation.
// Threads: 32 channels x 8 row groups per workgroup (grid = cdiv(C, 32), kBnFinalizeChannels).  Group g adds rows g, g + 8,
// ... in order and the eight group sums are combined in group order: the result depends only on the row contents, so with
// one producer per row (MN_DETERMINISTIC: thousands of rows) it is reproducible, and with the default 8 rows it is the
// plain sum of rows 0..7.
constexpr int kBnFinalizeChannels = 32;
__device__ __forceinline__ void bn_rows_sum(const double* __restrict__ accum, int rows, int C, int c, bool active,
                                            double (*part)[2][kBnFinalizeChannels], double& sa, double& sb) {
  const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
  sa = sb = 0;
  if (active)
    for (int r = grp; r < rows; r += 8) {
      sa += accum[(long)r * 2 * C + c];
      sb += accum[(long)r * 2 * C + C + c];
    }
  part[grp][0][cl] = sa;
  part[grp][1][cl] = sb;
  __syncthreads();
  if (grp == 0) {
    sa = sb = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      sa += part[g][0][cl];
      sb += part[g][1][cl];
    }
  }
  __syncthreads();
}

static __global__ void __launch_bounds__(256) bn_finalize_fwd_kernel(const double* __restrict__ accum, double count, BnParams p,
                                                                     int training, float* __restrict__ coef, int C,
                                                                     int accum_rows) {
  __shared__ double part[8][2][kBnFinalizeChannels];
  for (int c0 = blockIdx.x * kBnFinalizeChannels; c0 < C; c0 += gridDim.x * kBnFinalizeChannels) {
    const int c = c0 + (threadIdx.x & 31);
    double sa, sb;  // accum: [accum_rows][2][C]
    bn_rows_sum(accum, training ? accum_rows : 0, C, c, c < C, part, sa, sb);
    if ((threadIdx.x >> 5) != 0 || c >= C) continue;
    float mean, var;
    double unbiased = 0;
    if (training) {
      double m = sa / count;
      double v = sb / count - m * m;
      if (v < 0) v = 0;
      mean = (float)m;
      var = (float)v;
      unbiased = count > 1 ? v * count / (count - 1) : v;
    } else {
      mean = p.running_mean[c];
      var = p.running_var[c];
    }
    const float invstd = 1.0f / sqrtf(var + p.eps);
    const float sc = p.gamma[c] * invstd;
    coef[c] = sc;
    coef[C + c] = p.beta[c] - mean * sc;
    p.mean[c] = mean;
    p.invstd[c] = invstd;
    if (training) {
      p.running_mean[c] = (1.f - p.momentum) * p.running_mean[c] + p.momentum * mean;
      p.running_var[c] = (1.f - p.momentum) * p.running_var[c] + p.momentum * (float)unbiased;
      if (c == 0 && p.num_batches_tracked) *p.num_batches_tracked += 1;
    }
  }
}

// out = [relu]( y * scale[c] + shift[c] [+ res] ), one 16-byte piece per lane.
// A thread's channel piece is the same in every iteration of the grid-stride loop (the stride, gridDim.x * 256, is a
// multiple of the C / VEC pieces of a row, a power of two <= 256), so its VEC coefficient sets live in REGISTERS.  They
// get there through a per-workgroup LDS table laid out [e][piece]: the workgroup builds it cooperatively (a channel's
// coefficients are loaded from memory once per workgroup) and every lane then fetches its own VEC entries with reads whose
// addresses are consecutive across lanes -- no bank conflicts.
// (Rounds 1-2 indexed LDS arrays [c0 + e] inside the loop: lanes 8 channels apart hit the same 4 banks, an up to 8-way
// conflict on every one of 16 (forward) / 48 (backward) ds_read_b32 per piece.  A first round-3 form loaded the
// coefficients straight from memory into registers: 40 loads per thread in front of 1-8 loop iterations -- the backward
// apply went from 44 to 81 us.)
template <typename T>
static __global__ void __launch_bounds__(256) bn_apply_kernel(const T* __restrict__ y, const float* __restrict__ coef,
                                                        const T* __restrict__ res, T* __restrict__ out, long npieces,
                                                        int C, int relu) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int cpr = C / VEC;
  __shared__ floatx2 tab[512];  // [e][piece] -> (scale, shift)
  for (int c = threadIdx.x; c < C; c += 256) {
    const floatx2 v = {coef[c], coef[C + c]};
    tab[(c % VEC) * cpr + c / VEC] = v;
  }
  __syncthreads();
  const int cp = (int)(threadIdx.x % cpr);  // (blockIdx.x * 256 is a multiple of cpr)
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const floatx2 v = tab[e * cpr + cp];
    sc[e] = v[0];
    sh[e] = v[1];
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npieces; i += (long)gridDim.x * blockDim.x) {
    PieceView<T> v, r, o;
    v.p = MN_LOAD_LAST(reinterpret_cast<const piece_t*>(y) + i);
    if (res) r.p = reinterpret_cast<const piece_t*>(res)[i];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float f = (float)v.e[e] * sc[e] + sh[e];
      if (res) f += (float)r.e[e];
      if (relu) f = fmaxf(f, 0.f);
      o.e[e] = (T)f;
    }
    reinterpret_cast<piece_t*>(out)[i] = o.p;
  }
}

// Stem: BatchNorm (finalize as above) + ReLU + max-pool 3x3/2/1 in one pass: the normalised activation is
// never written.  Values are rounded to T before the comparison so the routing (first maximum) is the one the
// two-pass form (bn_apply then maxpool_fwd) takes.
template <typename T>
static __global__ void __launch_bounds__(256) bn_relu_maxpool_kernel(const T* __restrict__ y, const float* __restrict__ coef,
                                                               T* __restrict__ out, unsigned char* __restrict__ idx, int B,
                                                               int H, int W, int C, int Po, int Qo) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int cpr = C / VEC;
  float s_scale[VEC], s_shift[VEC];  // this thread's channel piece is loop invariant (see bn_apply_kernel)
  {
    __shared__ floatx2 tab[512];  // [e][piece] -> (scale, shift)
    for (int c = threadIdx.x; c < C; c += 256) {
      const floatx2 v = {coef[c], coef[C + c]};
      tab[(c % VEC) * cpr + c / VEC] = v;
    }
    __syncthreads();
    const int cp = (int)(threadIdx.x % cpr);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const floatx2 v = tab[e * cpr + cp];
      s_scale[e] = v[0];
      s_shift[e] = v[1];
    }
  }
  const long total = (long)B * Po * Qo * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cp = (int)(i % cpr);
    long tmp = i / cpr;
    const int qo = (int)(tmp % Qo);
    tmp /= Qo;
    const int po = (int)(tmp % Po);
    const int b = (int)(tmp / Po);
    float best[VEC];
    unsigned char arg[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      best[e] = -INFINITY;
      arg[e] = 0;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int h = po * 2 - 1 + r;
      if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int w = qo * 2 - 1 + s;
        if ((unsigned)w >= (unsigned)W) continue;
        PieceView<T> v;
        v.p = reinterpret_cast<const piece_t*>(y)[((long)(b * H + h) * W + w) * cpr + cp];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float f = fmaxf((float)v.e[e] * s_scale[e] + s_shift[e], 0.f);
          f = (float)(T)f;
          if (f > best[e]) {  // strict: the first maximum wins
            best[e] = f;
            arg[e] = (unsigned char)(r * 3 + s);
          }
        }
      }
    }
    PieceView<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.e[e] = (T)best[e];
    reinterpret_cast<piece_t*>(out)[i] = o.p;
    if (idx) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) idx[i * VEC + e] = arg[e];
    }
  }
}

// N consecutive elements of type TY starting at element index `i * N` (N * sizeof(TY) = 16 or 32 bytes), read as floats
template <typename TY, int N>
struct YVec {
  static constexpr int PIECES = N * (int)sizeof(TY) / 16;
  static constexpr int PER = 16 / (int)sizeof(TY);
  PieceView<TY> v[PIECES];
  __device__ __forceinline__ void load(const TY* __restrict__ base, long i) {
#pragma unroll
    for (int k = 0; k < PIECES; ++k) v[k].p = reinterpret_cast<const piece_t*>(base)[i * PIECES + k];
  }
  __device__ __forceinline__ void load_last(const TY* __restrict__ base, long i) {
#pragma unroll
    for (int k = 0; k < PIECES; ++k) v[k].p = MN_LOAD_LAST(reinterpret_cast<const piece_t*>(base) + i * PIECES + k);
  }
  __device__ __forceinline__ float at(int e) const { return (float)v[e / PER].e[e % PER]; }
};

// ---- BatchNorm backward ---------------------------------------------------------------------------
// reduce: accum[0][c] += sum gm, accum[1][c] += sum gm * xhat,  gm = g * (gate > 0 if gate)
// POOL: the gradient is gathered from (argmax, pooled gradient) -- a separate instantiation so that the common form does
// not carry the gather's registers
// TY (round 5, the fp16x2m mode): element type of the conv output y when it differs from the gradient's -- fp16 gradients against
// the fp32 conv output of the split-operand forward pass (the self gate and xhat then come from the exact forward values)
template <typename T, bool POOL = false, int U = 4, typename TY = T>
static __global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const T* __restrict__ g, const T* __restrict__ gate,
                                                             const TY* __restrict__ y, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, long M, int C,
                                                             double* __restrict__ accum, int rows_per_block,
                                                             float* __restrict__ partial,
                                                             const float* __restrict__ sg_gamma,
                                                             const float* __restrict__ sg_beta, PoolGradSrc pg,
                                                             int accum_rows) {
  // sg_gamma/sg_beta non-null ("self gate"): the consumer is this BatchNorm's own ReLU, so the gate
  // relu'(bn(y)) is recomputed from y (y*scale + shift > 0, the forward's arithmetic) instead of being read
  constexpr int VEC = ElemTraits<T>::VEC;
  __shared__ float red[2][256][VEC];
  const int cpr = C / VEC;           // pieces per row (power of two, <= 256)
  const int rlanes = 256 / cpr;
  const int cp = threadIdx.x % cpr, rl = threadIdx.x / cpr;
  const int c0 = cp * VEC;
  float mu[VEC], is[VEC], s1[VEC], s2[VEC], sc[VEC], sh[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    mu[e] = mean[c0 + e];
    is[e] = invstd[c0 + e];
    s1[e] = s2[e] = 0.f;
    sc[e] = sg_gamma ? sg_gamma[c0 + e] * is[e] : 0.f;
    sh[e] = sg_gamma ? sg_beta[c0 + e] - mu[e] * sc[e] : 0.f;
  }
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  // U rows per iteration, all loads issued before the arithmetic (memory-level parallelism)
  for (long r = r0 + rl; r < r1; r += (long)U * rlanes) {
    PieceView<T> vg[U], vm[U];
    YVec<TY, VEC> vy[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long rr = r + (long)u * rlanes;
      const bool in = rr < r1;
      const long idx = (in ? rr : r) * cpr + cp;
      if constexpr (POOL) {  // the gradient is the max-pool's input gradient, gathered on the fly
        float a[VEC];
        const long rrow = in ? rr : r;
        const int w_ = (int)(rrow % pg.W);
        const long t2 = rrow / pg.W;
        pool_grad_piece<T>(pg.idx, reinterpret_cast<const T*>(pg.gout), (int)(t2 / pg.H), (int)(t2 % pg.H), w_, cp, cpr, pg.Po,
                           pg.Qo, a);
#pragma unroll
        for (int e = 0; e < VEC; ++e) vg[u].e[e] = in ? (T)a[e] : (T)0.f;
      } else {
        vg[u].p = in ? reinterpret_cast<const piece_t*>(g)[idx] : zero_piece();
      }
      vy[u].load(y, idx);
      if (gate) vm[u].p = reinterpret_cast<const piece_t*>(gate)[idx];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float gv = (float)vg[u].e[e];
        if (gate && !((float)vm[u].e[e] > 0.f)) gv = 0.f;
        if (sg_gamma && !(vy[u].at(e) * sc[e] + sh[e] > 0.f)) gv = 0.f;
        s1[e] += gv;
        s2[e] += gv * (vy[u].at(e) - mu[e]) * is[e];
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
    if (partial) {  // [gridDim.x][2][C] fp32 partials, folded by bn_reduce_partials_kernel (few atomics)
      partial[((long)blockIdx.x * 2 + 0) * C + idx] = (float)a;
      partial[((long)blockIdx.x * 2 + 1) * C + idx] = (float)b;
    } else {  // accum: [accum_rows][2][C]; the row spreads the same-address contention
      double* row = accum + (long)((int)blockIdx.x % accum_rows) * 2 * C;
      atomicAdd(row + idx, a);
      atomicAdd(row + C + idx, b);
    }
  }
}

// The same sums when the gradient is the max-pool's input gradient (stem, fp32 tensors), walked in the order of the POOLED WINDOWS
// (round 4; stem_bwd.h does this for fp16 tensors through LDS tiles): a window's gradient goes to exactly one pixel, its recorded
// argmax, so  sum(gm) = sum over windows of gp * gate(y at the argmax)  and  sum(gm * xhat)  likewise -- one 4-byte load of y per
// (window, channel) over a quarter of the elements, instead of four windows' argmax bytes compared per conv-output element
// (bn_bwd_reduce_kernel<float, true>: 616 us at 192 images, VALU-bound on pool_grad_piece).
static __global__ void __launch_bounds__(256) bn_bwd_reduce_pool_windows_kernel(const float* __restrict__ gp,
                                                                                const unsigned char* __restrict__ idx,
                                                                                const float* __restrict__ y,
                                                                                const float* __restrict__ mean,
                                                                                const float* __restrict__ invstd,
                                                                                const float* __restrict__ sg_gamma,
                                                                                const float* __restrict__ sg_beta, int B, int H,
                                                                                int W, int Po, int Qo, int C, double* __restrict__ accum,
                                                                                int rows_per_block, int accum_rows) {
  constexpr int VEC = 4;
  __shared__ float red[2][256][VEC];
  const int cpr = C / VEC, rlanes = 256 / cpr;
  const int cp = threadIdx.x % cpr, rl = threadIdx.x / cpr, c0 = cp * VEC;
  float mu[VEC], is[VEC], sc[VEC], sh[VEC], s1[VEC], s2[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    mu[e] = mean[c0 + e];
    is[e] = invstd[c0 + e];
    sc[e] = sg_gamma[c0 + e] * is[e];
    sh[e] = sg_beta[c0 + e] - mu[e] * sc[e];
    s1[e] = s2[e] = 0.f;
  }
  const long nwin = (long)B * Po * Qo;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = r0 + rows_per_block < nwin ? r0 + rows_per_block : nwin;
  for (long r = r0 + rl; r < r1; r += rlanes) {
    PieceView<float> g;
    g.p = reinterpret_cast<const piece_t*>(gp)[r * cpr + cp];
    const unsigned taps = reinterpret_cast<const unsigned*>(idx)[r * cpr + cp];
    const int qo = (int)(r % Qo);
    const long t2 = r / Qo;
    const int po = (int)(t2 % Po), b = (int)(t2 / Po);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int tap = (int)((taps >> (8 * e)) & 0xffu);
      const int h = po * 2 - 1 + tap / 3, w = qo * 2 - 1 + tap % 3;  // inside the image: the forward pass only records valid taps
      const float yv = y[((long)(b * H + h) * W + w) * C + c0 + e];
      const float gv = (yv * sc[e] + sh[e] > 0.f) ? g.e[e] : 0.f;    // the stem's own ReLU, recomputed
      s1[e] += gv;
      s2[e] += gv * (yv - mu[e]) * is[e];
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    red[0][threadIdx.x][e] = s1[e];
    red[1][threadIdx.x][e] = s2[e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) {
    const int p = i / VEC, e = i % VEC;
    double a = 0, bsum = 0;
    for (int l = 0; l < rlanes; ++l) {
      a += red[0][l * cpr + p][e];
      bsum += red[1][l * cpr + p][e];
    }
    double* row = accum + (long)((int)blockIdx.x % accum_rows) * 2 * C;
    atomicAdd(row + i, a);
    atomicAdd(row + C + i, bsum);
  }
}

// finalize (one small launch between reduce and apply): coef[0][c] = k1 = gamma*invstd, coef[1][c] = mg = sum(gm)/M,
// coef[2][c] = mgx = sum(gm*xhat)/M, coef[3][c] = shift of the self-gate (only with sg_beta), from the fp64 accumulator
// rows; adds dgamma / dbeta (times grad_unscale = 1/loss_scale) into the gradient arena.
static __global__ void __launch_bounds__(256) bn_finalize_bwd_kernel(const double* __restrict__ accum, double count,
                                                                     const float* __restrict__ gamma,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd,
                                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                     float grad_unscale, const float* __restrict__ sg_beta,
                                                                     float* __restrict__ coef, int C, int accum_rows) {
  __shared__ double part[8][2][kBnFinalizeChannels];
  for (int c0 = blockIdx.x * kBnFinalizeChannels; c0 < C; c0 += gridDim.x * kBnFinalizeChannels) {
    const int c = c0 + (threadIdx.x & 31);
    double sg, sgx;  // accum: [accum_rows][2][C], rows summed in the fixed order of bn_rows_sum
    bn_rows_sum(accum, accum_rows, C, c, c < C, part, sg, sgx);
    if ((threadIdx.x >> 5) != 0 || c >= C) continue;
    const float k1 = gamma[c] * invstd[c];
    coef[c] = k1;
    coef[C + c] = (float)(sg / count);
    coef[2 * C + c] = (float)(sgx / count);
    if (sg_beta) coef[3 * C + c] = sg_beta[c] - mean[c] * k1;
    dgamma[c] += (float)(sgx * grad_unscale);
    dbeta[c] += (float)(sg * grad_unscale);
  }
}

// apply: gy = k1 * (gm - mg - xhat * mgx)
template <typename T, bool POOL = false, typename TY = T>
static __global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ gate,
                                                            const TY* __restrict__ y, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ coef,
                                                            T* __restrict__ gy, long npieces, int C, int self_gate,
                                                            PoolGradSrc pg) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const bool sg_beta = self_gate != 0;
  const int cpr = C / VEC;
  // gy = k1 (gm - mg - xhat mgx), xhat = (y - mean) invstd, as  k1 gm + kb (y - mean) + kd  with per-channel coefficients
  // in registers (this thread's channel piece is loop invariant: see bn_apply_kernel)
  float k1[VEC], kb[VEC], kd[VEC], mu[VEC], sh[VEC];
  {
    __shared__ floatx4 tab[512];  // [e][piece] -> (k1, kb, kd, mean)
    __shared__ float tab_sh[512];
    for (int c = threadIdx.x; c < C; c += 256) {
      const float a = coef[c];
      const floatx4 v = {a, -a * coef[2 * C + c] * invstd[c], -a * coef[C + c], mean[c]};
      const int at = (c % VEC) * cpr + c / VEC;
      tab[at] = v;
      tab_sh[at] = sg_beta ? coef[3 * C + c] : 0.f;
    }
    __syncthreads();
    const int cp = (int)(threadIdx.x % cpr);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const floatx4 v = tab[e * cpr + cp];
      k1[e] = v[0];
      kb[e] = v[1];
      kd[e] = v[2];
      mu[e] = v[3];
      sh[e] = tab_sh[e * cpr + cp];
    }
  }
  // (Walking the tensor back to front -- the reduction pass that ran just before read g and y front to back, so their
  // tails are what the caches still hold -- was measured: no gain.)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npieces; i += (long)gridDim.x * blockDim.x) {
    PieceView<T> vg, vm, o;
    YVec<TY, VEC> vy;
    if constexpr (POOL) {
      float a[VEC];
      const long row = i / cpr;
      const int w_ = (int)(row % pg.W);
      const long t2 = row / pg.W;
      pool_grad_piece<T>(pg.idx, reinterpret_cast<const T*>(pg.gout), (int)(t2 / pg.H), (int)(t2 % pg.H), w_, (int)(i % cpr), cpr,
                         pg.Po, pg.Qo, a);
#pragma unroll
      for (int e = 0; e < VEC; ++e) vg.e[e] = (T)a[e];
    } else {
      vg.p = MN_LOAD_LAST(reinterpret_cast<const piece_t*>(g) + i);
    }
    vy.load_last(y, i);
    if (gate) vm.p = reinterpret_cast<const piece_t*>(gate)[i];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float gv = (float)vg.e[e];
      if (gate && !((float)vm.e[e] > 0.f)) gv = 0.f;
      if (sg_beta && !(vy.at(e) * k1[e] + sh[e] > 0.f)) gv = 0.f;  // self gate (see the reduce kernel)
      o.e[e] = (T)(k1[e] * gv + (kb[e] * (vy.at(e) - mu[e]) + kd[e]));
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

// BatchNorm backward = reduce -> finalize -> apply.  accum: [accum_rows][2][C] doubles, zero on entry (left holding the
// sums); coef: [4][C] floats of scratch ([3][C] without the self gate).
template <typename T, typename TY = T>
inline void launch_bn_bwd(const T* g, const T* gate, const TY* y, long M, int C, const float* gamma, const float* mean,
                          const float* invstd, float* dgamma, float* dbeta, T* gy, double* accum, float* coef,
                          float grad_unscale, hipStream_t s, const float* self_gate_beta = nullptr,
                          PoolGradSrc pg = PoolGradSrc(), int accum_rows = 1, bool apply = true) {
  // apply = false: reduce + finalize only (sums, d(gamma), d(beta), coef) -- the consumer applies the coefficients itself
  // (stem_bwd.h computes d(conv output) tile by tile inside the weight-gradient kernel); gy is not written
  // self_gate_beta: the gradient g is taken w.r.t. relu(bn(y)) of THIS BatchNorm; the ReLU gate is recomputed
  // from y and `gate` is not read
  constexpr int VEC = ElemTraits<T>::VEC;
  const float* sg_gamma = self_gate_beta ? gamma : nullptr;
  if (self_gate_beta) gate = nullptr;
  // ~512 workgroups (2 per CU): the reduction is HBM-bound and needs the whole chip, but every workgroup ends with an LDS
  // reduction and 2 x C fp64 atomics -- at 4096 workgroups (round 1) layers 3-4 did ONE loop iteration per workgroup and
  // that epilogue weighed as much as the loads (same-box A/Bs of the whole step: 4096 workgroups 17.77 ms,
  // 1024 17.60, 512 17.42, 256 17.97; round 4, MN_BN_REDUCE_WGS: 512 / 768 / 1024 = 13.48 / 13.58 / 13.63 ms in fp16, equal within
  // 0.1 % on the fp16x2 mode's fp32 tensors, profiles/r04/c36_*)
  static const long target = getenv("MN_BN_REDUCE_WGS") ? atol(getenv("MN_BN_REDUCE_WGS")) : 512;
  const int rlanes = 256 / (C / VEC);
  long rows = (M + target - 1) / target;
  rows = ((rows + rlanes - 1) / rlanes) * rlanes;
  if (rows < 4L * rlanes) rows = 4L * rlanes;
  int rows_per_block = (int)rows;
  const int nblk = cdiv(M, rows_per_block);
  // (4 rows per thread per iteration in flight; 8 measured equal: 15.07 vs 14.99 ms per step)
  // rows in flight per thread: 2 for fp16 tensors (111 instead of 122 registers: one more wave fits beside the side stream's
  // weight gradient; 14.01 -> 13.89 ms per step), 4 for fp32 tensors (2: 30.74 -> 30.88 ms in the fp16x2 mode); profiles/r04/c11_*
  static const bool pool_windows = !(getenv("MN_POOL_WINDOWS") && atoi(getenv("MN_POOL_WINDOWS")) == 0);
  if (pg.idx && sizeof(T) == 4 && sizeof(TY) == 4 && self_gate_beta && pool_windows) {  // the stem on fp32 tensors: sums in pooled-window order
    const long nwin = (long)(M / ((long)pg.H * pg.W)) * pg.Po * pg.Qo;
    long wrows = (nwin + target - 1) / target;
    wrows = ((wrows + rlanes - 1) / rlanes) * rlanes;
    hipLaunchKernelGGL(bn_bwd_reduce_pool_windows_kernel, dim3(cdiv(nwin, wrows)), dim3(256), 0, s,
                       reinterpret_cast<const float*>(pg.gout), pg.idx, reinterpret_cast<const float*>(y), mean, invstd, sg_gamma,
                       self_gate_beta, (int)(M / ((long)pg.H * pg.W)), pg.H, pg.W, pg.Po, pg.Qo, C, accum, (int)wrows, accum_rows);
  } else {
  static const int reduce_u = getenv("MN_BN_REDUCE_U") ? atoi(getenv("MN_BN_REDUCE_U")) : (sizeof(T) == 2 ? 2 : 4);
  // (s_setprio 3 in the backward kernels, so that their waves are not starved by the weight gradient's MFMA waves on the same
  //  SIMD: no effect, 13.95 vs 14.00 ms fp16, 30.36 vs 30.38 ms fp16x2, profiles/r04/c12_*; removed)
  if (pg.idx)
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, true, 4, TY>), dim3(nblk), dim3(256), 0, s, g, gate, y, mean, invstd, M, C, accum,
                       rows_per_block, (float*)nullptr, sg_gamma, self_gate_beta, pg, accum_rows);
  else if (reduce_u == 2)
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, false, 2, TY>), dim3(nblk), dim3(256), 0, s, g, gate, y, mean, invstd, M, C, accum,
                       rows_per_block, (float*)nullptr, sg_gamma, self_gate_beta, pg, accum_rows);
  else
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, false, 4, TY>), dim3(nblk), dim3(256), 0, s, g, gate, y, mean, invstd, M, C, accum,
                       rows_per_block, (float*)nullptr, sg_gamma, self_gate_beta, pg, accum_rows);
  }
#ifdef MN_ABLATION_BUILD
  // timing experiment (results wrong): what the finalize launches of the units with at most MN_ABL_SKIP_FINALIZE channels cost a
  // step -- they run for the first 400 calls (coefficients of the warm-up steps stay in place), then are skipped
  static const int skip_c = getenv("MN_ABL_SKIP_FINALIZE") ? atoi(getenv("MN_ABL_SKIP_FINALIZE")) : 0;
  static long calls = 0;
  if (!(C <= skip_c && ++calls > 400))
#endif
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(cdiv(C, kBnFinalizeChannels)), dim3(256), 0, s, (const double*)accum, (double)M, gamma, mean,
                     invstd, dgamma, dbeta, grad_unscale, self_gate_beta, coef, C, accum_rows);
  if (!apply) return;
  long np = M * C / VEC;
  if (pg.idx)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, true, TY>), dim3(ew_grid(np)), dim3(256), 0, s, g, gate, y, mean, invstd,
                       (const float*)coef, gy, np, C, self_gate_beta ? 1 : 0, pg);
  else
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, false, TY>), dim3(ew_grid(np)), dim3(256), 0, s, g, gate, y, mean, invstd,
                       (const float*)coef, gy, np, C, self_gate_beta ? 1 : 0, pg);
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

// g[b][p][c] = gp[b][c] / HW, zeroed where gate[b][p][c] <= 0 (gate = the pooled activation: the ReLU that produced it)
// gate_h2 (fp16 tensors): the gate is an h2 tensor (common.h) of which the hi halves are read
template <typename T>
static __global__ void __launch_bounds__(256) avgpool_bwd_kernel(const float* __restrict__ gp, T* __restrict__ g, int B, int HW,
                                                           int C, const T* __restrict__ gate, int gate_h2) {
  long total = (long)B * HW * C;
  float inv = 1.f / (float)HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    int b = (int)(i / ((long)HW * C));
    float v = gp[b * C + c] * inv;
    if (gate && !((float)gate[gate_h2 ? h2_index(i / C, C, c) : i] > 0.f)) v = 0.f;
    g[i] = (T)v;
  }
}

}  // namespace mn
