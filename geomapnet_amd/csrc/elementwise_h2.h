// The HBM-bound kernels of the training step that PRODUCE or READ h2 tensors (common.h: fp32-class values as fp16 hi + lo
// halves, per row and 32-channel group 64 bytes of hi halves then 64 bytes of lo halves).  In the fp16x2 mode
// (MN_DTYPE_F16X2) every tensor a convolution consumes is h2 and is split exactly once, here, by the kernel that produces
// it: the normalised activations (bn_apply_h2, bn_relu_maxpool_h2) and d(conv output) (bn_bwd_apply_h2).  Their inputs
// -- raw conv outputs, data gradients -- are fp32, as are the statistics / reduction kernels of elementwise.h, which this
// mode uses unchanged.  A thread owns 8 consecutive channels of a row: two 16-byte fp32 pieces in, one 16-byte piece of hi
// halves + one of lo halves out (same bytes per element as the fp32 kernels).
#pragma once
#include "elementwise.h"

namespace mn {

// 8 channels c .. c + 7 (c % 8 == 0) of row `row`: the hi piece sits at half index h2_index(row, C, c), the lo piece 32 halves on
__device__ __forceinline__ void h2_store8(half* __restrict__ base, long row, int C, int c, const float (&x)[8]) {
  PieceView<half> hi, lo;
  split8_f16(x, hi.v, lo.v);
  half* p = base + h2_index(row, C, c);
  *reinterpret_cast<piece_t*>(p) = hi.p;
  *reinterpret_cast<piece_t*>(p + 32) = lo.p;
}
__device__ __forceinline__ void h2_load8(const half* __restrict__ base, long row, int C, int c, float (&x)[8]) {
  const half* p = base + h2_index(row, C, c);
  PieceView<half> hi, lo;
  hi.p = *reinterpret_cast<const piece_t*>(p);
  lo.p = *reinterpret_cast<const piece_t*>(p + 32);
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = (float)hi.e[e] + (float)lo.e[e];
}

// ---- h2q (common.h): fp16 hi halves where h2 has them, the lo halves and a copy of the hi halves as fp8 with fixed exponents ----
// four floats -> four fp8 e4m3 bytes (saturating at +-448)
__device__ __forceinline__ unsigned q8_pack4(float a, float b, float c, float d) {
#if defined(__HIP_DEVICE_COMPILE__)
  a = fminf(fmaxf(a, -448.f), 448.f);
  b = fminf(fmaxf(b, -448.f), 448.f);
  c = fminf(fmaxf(c, -448.f), 448.f);
  d = fminf(fmaxf(d, -448.f), 448.f);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);  // bytes 0, 1
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);       // bytes 2, 3
  return (unsigned)w;
#else
  return (unsigned)fp8_e4m3_from_float(a) | ((unsigned)fp8_e4m3_from_float(b) << 8) | ((unsigned)fp8_e4m3_from_float(c) << 16) |
         ((unsigned)fp8_e4m3_from_float(d) << 24);
#endif
}
// 8 channels c .. c + 7 of an ACTIVATION row: the hi piece where h2 has it, 8 bytes of lo8 and 8 bytes of hi8 behind the group's hi halves
__device__ __forceinline__ void h2q_store8(half* __restrict__ base, long row, int C, int c, const float (&x)[8]) {
  PieceView<half> hi, lo;
  split8_f16(x, hi.v, lo.v);
  half* p = base + h2_index(row, C, c);
  *reinterpret_cast<piece_t*>(p) = hi.p;
  float l[8], h[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float hf = (float)hi.e[e];
    l[e] = (x[e] - hf) * pow2c(kQA_LO);
    h[e] = hf * pow2c(kQA_HI);
  }
  unsigned char* q = reinterpret_cast<unsigned char*>(base) + h2q_byte(row, C, c, 0);
  u32x2 lo8 = {q8_pack4(l[0], l[1], l[2], l[3]), q8_pack4(l[4], l[5], l[6], l[7])};
  u32x2 hi8 = {q8_pack4(h[0], h[1], h[2], h[3]), q8_pack4(h[4], h[5], h[6], h[7])};
  *reinterpret_cast<u32x2*>(q) = lo8;
  *reinterpret_cast<u32x2*>(q + 32) = hi8;
}
// what a reader of the stored activation gets: hi + lo8 * 2^-kQA_LO
__device__ __forceinline__ void h2q_load8(const half* __restrict__ base, long row, int C, int c, float (&x)[8]) {
  PieceView<half> hi;
  hi.p = *reinterpret_cast<const piece_t*>(base + h2_index(row, C, c));
  const u32x2 lo8 = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned char*>(base) + h2q_byte(row, C, c, 0));
#pragma unroll
  for (int e = 0; e < 8; ++e)
    x[e] = (float)hi.e[e] + fp8_e4m3_to_float((unsigned char)(lo8[e >> 2] >> (8 * (e & 3)))) * pow2c(-kQA_LO);
}

// 8 channels of row `row` as ONE plain fp16 piece (round-to-nearest) of a [rows][C] fp16 tensor: the copy of an activation the
// single-fp16 backward pass of the fp16x2m mode reads (weight-gradient operand, ReLU gates), written by the kernel that
// produces the h2 tensor
__device__ __forceinline__ void f16_store8(half* __restrict__ base, long row, int C, int c, const float (&x)[8]) {
  PieceView<half> o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o.e[e] = (half)x[e];
  *reinterpret_cast<piece_t*>(base + row * C + c) = o.p;
}

// ---- the backward record of a BatchNorm (round 6, fp16x2m) ------------------------------------------------------------------------
// What BatchNorm's backward pass needs of the forward pass is xhat = (y - mean) * invstd, to the precision of a REDUCTION operand and
// of an O(mgx) correction term, plus -- for the unit in front of a ReLU -- the exact OUTCOME of that ReLU.  Round 5 re-read the fp32
// conv output for both (4 bytes per element in each of the two backward passes, the gate recomputed with the forward's arithmetic).
// The record is 2 bytes: fp16(xhat), round to nearest, with its lowest mantissa bit replaced by the gate -- xhat to 2^-10 relative
// (it has zero mean and unit variance per channel, so no offset eats the bits, unlike fp16(y)), the gate EXACT.  Dense [M][C] fp16.
// Cost: +2 bytes per element written by the forward apply; saving: 2 bytes per element in each backward pass.
__device__ __forceinline__ half rec_pack(float xhat, bool gate) {
  const half h = (half)xhat;
  const unsigned short b = (unsigned short)((__builtin_bit_cast(unsigned short, h) & 0xfffeu) | (gate ? 1u : 0u));
  return __builtin_bit_cast(half, b);
}
__device__ __forceinline__ float rec_xhat(half r) {
  return (float)__builtin_bit_cast(half, (unsigned short)(__builtin_bit_cast(unsigned short, r) & 0xfffeu));
}
__device__ __forceinline__ bool rec_gate(half r) { return (__builtin_bit_cast(unsigned short, r) & 1u) != 0; }

// out(h2) = [relu]( y * scale[c] + shift[c] [+ res(h2)] );  bn_apply_kernel with 8 channels per thread.  (Round 5 wrote a plain fp16
// copy of the same values beside it for the fp16x2m mode's backward pass; its kernels now read the hi halves in place.)
// rec (round 6, training passes of the fp16x2m mode; may be null): the BACKWARD RECORD of this BatchNorm (rec_pack below), written
// here because this pass holds y, the statistics and the ReLU's outcome at once.
static __global__ void __launch_bounds__(256) bn_apply_h2_kernel(const float* __restrict__ y, const float* __restrict__ coef,
                                                                 const half* __restrict__ res, half* __restrict__ out,
                                                                 long nitems, int C, int relu, int q, half* __restrict__ rec,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd) {
  constexpr int VEC = 8;
  const int cpr = C / VEC;
  __shared__ floatx4 tab[512];  // [e][piece] -> (scale, shift, mean, invstd)
  for (int c = threadIdx.x; c < C; c += 256) {
    const floatx4 v = {coef[c], coef[C + c], rec ? mean[c] : 0.f, rec ? invstd[c] : 0.f};
    tab[(c % VEC) * cpr + c / VEC] = v;
  }
  __syncthreads();
  const int cp = (int)(threadIdx.x % cpr);  // loop invariant: the grid stride is a multiple of cpr (a power of two <= 64)
  float sc[VEC], sh[VEC], mu[VEC], is[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const floatx4 v = tab[e * cpr + cp];
    sc[e] = v[0];
    sh[e] = v[1];
    mu[e] = v[2];
    is[e] = v[3];
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nitems; i += (long)gridDim.x * blockDim.x) {
    const long row = i / cpr;
    PieceView<float> v0, v1;
    v0.p = MN_LOAD_LAST(reinterpret_cast<const piece_t*>(y) + 2 * i);
    v1.p = MN_LOAD_LAST(reinterpret_cast<const piece_t*>(y) + 2 * i + 1);
    float f[VEC] = {v0.e[0], v0.e[1], v0.e[2], v0.e[3], v1.e[0], v1.e[1], v1.e[2], v1.e[3]};
    float r[VEC];
    if (res) {
      if (q)
        h2q_load8(res, row, C, cp * VEC, r);
      else
        h2_load8(res, row, C, cp * VEC, r);
    }
    PieceView<half> rc;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float xh = (f[e] - mu[e]) * is[e];
      f[e] = f[e] * sc[e] + sh[e];
      // the gate the backward pass of THIS BatchNorm may need is its own ReLU's (a unit whose ReLU follows the residual add never
      // needs one: the gradient of a block output arrives already gated)
      rc.e[e] = rec_pack(xh, !res && relu ? f[e] > 0.f : true);
      if (res) f[e] += r[e];
      if (relu) f[e] = fmaxf(f[e], 0.f);
    }
    if (q)
      h2q_store8(out, row, C, cp * VEC, f);
    else
      h2_store8(out, row, C, cp * VEC, f);
    if (rec) *reinterpret_cast<piece_t*>(rec + row * C + cp * VEC) = rc.p;
  }
}

// Stem: BatchNorm + ReLU + max-pool 3x3/2/1 in one pass, fp32 conv output in, h2 pooled activation + argmax bytes out.
// The comparison is on the fp32 values (the oracle's), so the routing is the reference's.
static __global__ void __launch_bounds__(256) bn_relu_maxpool_h2_kernel(const float* __restrict__ y, const float* __restrict__ coef,
                                                                        half* __restrict__ out, unsigned char* __restrict__ idx,
                                                                        int B, int H, int W, int C, int Po, int Qo,
                                                                        half* __restrict__ y16, int q) {
  constexpr int VEC = 8;
  const int cpr = C / VEC;
  float s_scale[VEC], s_shift[VEC];
  {
    __shared__ floatx2 tab[512];
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
    const long orow = tmp;
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
        const piece_t* src = reinterpret_cast<const piece_t*>(y) + (((long)(b * H + h) * W + w) * cpr + cp) * 2;
        PieceView<float> v0, v1;
        v0.p = src[0];
        v1.p = src[1];
        const float raw[VEC] = {v0.e[0], v0.e[1], v0.e[2], v0.e[3], v1.e[0], v1.e[1], v1.e[2], v1.e[3]};
        // y16 (fp16x2m): plain fp16 copy of the conv output for the stem's fp16 backward kernels (stem_bwd.h).  Every pixel is
        // written exactly once: by the window whose taps r, s in {1, 2} cover it (pixel (h, w) -> window (h >> 1, w >> 1))
        if (y16 && r >= 1 && s >= 1) f16_store8(y16, (long)(b * H + h) * W + w, C, cp * VEC, raw);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float f = fmaxf(raw[e] * s_scale[e] + s_shift[e], 0.f);
          if (f > best[e]) {  // strict: the first maximum wins
            best[e] = f;
            arg[e] = (unsigned char)(r * 3 + s);
          }
        }
      }
    }
    if (q)
      h2q_store8(out, orow, C, cp * VEC, best);
    else
      h2_store8(out, orow, C, cp * VEC, best);
    if (idx) {
      unsigned long long packed = 0;
#pragma unroll
      for (int e = 0; e < VEC; ++e) packed |= (unsigned long long)arg[e] << (8 * e);
      reinterpret_cast<unsigned long long*>(idx)[i] = packed;
    }
  }
}

// gy(h2) = k1 * (gm - mg - xhat * mgx): bn_bwd_apply_kernel<float> with 8 channels per thread and an h2 store.  The ReLU gate,
// where there is one, is this BatchNorm's own (recomputed from y): block-output gradients arrive already gated.
static __global__ void __launch_bounds__(256) bn_bwd_apply_h2_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd,
                                                                     const float* __restrict__ coef, half* __restrict__ gy,
                                                                     long nitems, int C, int self_gate) {
  constexpr int VEC = 8;
  const bool sg_beta = self_gate != 0;
  const int cpr = C / VEC;
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
  const int cp = (int)(threadIdx.x % cpr);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nitems; i += (long)gridDim.x * blockDim.x) {
    PieceView<float> g0, g1, y0, y1;
    g0.p = MN_LOAD_LAST(reinterpret_cast<const piece_t*>(g) + 2 * i);
    g1.p = MN_LOAD_LAST(reinterpret_cast<const piece_t*>(g) + 2 * i + 1);
    y0.p = MN_LOAD_LAST(reinterpret_cast<const piece_t*>(y) + 2 * i);
    y1.p = MN_LOAD_LAST(reinterpret_cast<const piece_t*>(y) + 2 * i + 1);
    const float gv[VEC] = {g0.e[0], g0.e[1], g0.e[2], g0.e[3], g1.e[0], g1.e[1], g1.e[2], g1.e[3]};
    const float yv[VEC] = {y0.e[0], y0.e[1], y0.e[2], y0.e[3], y1.e[0], y1.e[1], y1.e[2], y1.e[3]};
    float o[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float gg = gv[e];
      if (sg_beta && !(yv[e] * k1[e] + sh[e] > 0.f)) gg = 0.f;
      o[e] = k1[e] * gg + (kb[e] * (yv[e] - mu[e]) + kd[e]);
    }
    h2_store8(gy, i / cpr, C, cp * VEC, o);
  }
}

// BatchNorm backward with an h2 d(conv output): elementwise.h's reduce + finalize on the fp32 tensors, then the apply above
inline void launch_bn_bwd_h2(const float* g, const float* y, long M, int C, const float* gamma, const float* mean,
                             const float* invstd, float* dgamma, float* dbeta, half* gy, double* accum, float* coef,
                             float grad_unscale, hipStream_t s, const float* self_gate_beta, int accum_rows) {
  launch_bn_bwd<float>(g, nullptr, y, M, C, gamma, mean, invstd, dgamma, dbeta, nullptr, accum, coef, grad_unscale, s,
                       self_gate_beta, PoolGradSrc(), accum_rows, /*apply=*/false);
  const long ni = M * C / 8;
  hipLaunchKernelGGL(bn_bwd_apply_h2_kernel, dim3(ew_grid(ni)), dim3(256), 0, s, g, y, mean, invstd, (const float*)coef, gy, ni, C,
                     self_gate_beta ? 1 : 0);
}

// ---- BatchNorm backward from the record (fp16x2m): fp16 gradient + fp16 record in, fp16 d(conv output) out ------------------------
// reduce: accum[0][c] += sum gm, accum[1][c] += sum gm * xhat, gm = g where the record's gate bit is set (use_gate) or g as it is
template <int U>
static __global__ void __launch_bounds__(256) bn_bwd_reduce_rec_kernel(const half* __restrict__ g, const half* __restrict__ rec, long M,
                                                                       int C, double* __restrict__ accum, int rows_per_block,
                                                                       int use_gate, int accum_rows) {
  constexpr int VEC = 8;
  __shared__ float red[2][256][VEC];
  const int cpr = C / VEC, rlanes = 256 / cpr;
  const int cp = threadIdx.x % cpr, rl = threadIdx.x / cpr;
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) s1[e] = s2[e] = 0.f;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  for (long r = r0 + rl; r < r1; r += (long)U * rlanes) {
    PieceView<half> vg[U], vr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long rr = r + (long)u * rlanes;
      const bool in = rr < r1;
      const long idx = (in ? rr : r) * cpr + cp;
      vg[u].p = in ? reinterpret_cast<const piece_t*>(g)[idx] : zero_piece();
      vr[u].p = reinterpret_cast<const piece_t*>(rec)[idx];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float gv = (float)vg[u].e[e];
        if (use_gate && !rec_gate(vr[u].e[e])) gv = 0.f;
        s1[e] += gv;
        s2[e] += gv * rec_xhat(vr[u].e[e]);
      }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    red[0][threadIdx.x][e] = s1[e];
    red[1][threadIdx.x][e] = s2[e];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < C; idx += 256) {
    const int p = idx / VEC, e = idx % VEC;
    double a = 0, b = 0;
    for (int l = 0; l < rlanes; ++l) {
      a += red[0][l * cpr + p][e];
      b += red[1][l * cpr + p][e];
    }
    double* row = accum + (long)((int)blockIdx.x % accum_rows) * 2 * C;  // [accum_rows][2][C]
    atomicAdd(row + idx, a);
    atomicAdd(row + C + idx, b);
  }
}

// apply: gy = k1 * (gm - mg - xhat * mgx) = k1 gm + kx xhat + kd with (k1, kx, kd) per channel in registers (coef: bn_finalize_bwd_kernel's)
static __global__ void __launch_bounds__(256) bn_bwd_apply_rec_kernel(const half* __restrict__ g, const half* __restrict__ rec,
                                                                      const float* __restrict__ coef, half* __restrict__ gy,
                                                                      long npieces, int C, int use_gate) {
  constexpr int VEC = 8;
  const int cpr = C / VEC;
  float k1[VEC], kx[VEC], kd[VEC];
  {
    __shared__ floatx4 tab[512];  // [e][piece] -> (k1, kx, kd, -)
    for (int c = threadIdx.x; c < C; c += 256) {
      const float a = coef[c];
      const floatx4 v = {a, -a * coef[2 * C + c], -a * coef[C + c], 0.f};
      tab[(c % VEC) * cpr + c / VEC] = v;
    }
    __syncthreads();
    const int cp = (int)(threadIdx.x % cpr);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const floatx4 v = tab[e * cpr + cp];
      k1[e] = v[0];
      kx[e] = v[1];
      kd[e] = v[2];
    }
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npieces; i += (long)gridDim.x * blockDim.x) {
    PieceView<half> vg, vr, o;
    vg.p = MN_LOAD_LAST(reinterpret_cast<const piece_t*>(g) + i);
    vr.p = MN_LOAD_LAST(reinterpret_cast<const piece_t*>(rec) + i);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float gv = (float)vg.e[e];
      if (use_gate && !rec_gate(vr.e[e])) gv = 0.f;
      o.e[e] = (half)(k1[e] * gv + (kx[e] * rec_xhat(vr.e[e]) + kd[e]));
    }
    reinterpret_cast<piece_t*>(gy)[i] = o.p;
  }
}

// reduce -> finalize (elementwise.h: sums -> coefficients, d gamma, d beta) -> apply, all on the record
inline void launch_bn_bwd_rec(const half* g, const half* rec, long M, int C, const float* gamma, const float* mean, const float* invstd,
                              float* dgamma, float* dbeta, half* gy, double* accum, float* coef, float grad_unscale, hipStream_t s,
                              bool use_gate, int accum_rows) {
  constexpr int VEC = 8;
  static const long target = getenv("MN_BN_REDUCE_WGS") ? atol(getenv("MN_BN_REDUCE_WGS")) : 512;  // (launch_bn_bwd explains)
  const int rlanes = 256 / (C / VEC);
  long rows = (M + target - 1) / target;
  rows = ((rows + rlanes - 1) / rlanes) * rlanes;
  if (rows < 4L * rlanes) rows = 4L * rlanes;
  const int rows_per_block = (int)rows;
  const int nblk = cdiv(M, rows_per_block);
  static const int reduce_u = getenv("MN_BN_REDUCE_U") ? atoi(getenv("MN_BN_REDUCE_U")) : 4;
  if (reduce_u == 2)
    hipLaunchKernelGGL((bn_bwd_reduce_rec_kernel<2>), dim3(nblk), dim3(256), 0, s, g, rec, M, C, accum, rows_per_block, use_gate ? 1 : 0,
                       accum_rows);
  else
    hipLaunchKernelGGL((bn_bwd_reduce_rec_kernel<4>), dim3(nblk), dim3(256), 0, s, g, rec, M, C, accum, rows_per_block, use_gate ? 1 : 0,
                       accum_rows);
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(cdiv(C, kBnFinalizeChannels)), dim3(256), 0, s, (const double*)accum, (double)M, gamma, mean,
                     invstd, dgamma, dbeta, grad_unscale, (const float*)nullptr, coef, C, accum_rows);
  const long np = M * C / VEC;
  hipLaunchKernelGGL(bn_bwd_apply_rec_kernel, dim3(ew_grid(np)), dim3(256), 0, s, g, rec, (const float*)coef, gy, np, C, use_gate ? 1 : 0);
}

// out[i] = (float) in[i], 8 elements per thread (fp16x2m: the pooled stem activation's fp16 gradient for the stem's fp32 chain)
static __global__ void __launch_bounds__(256) widen_f16_kernel(const half* __restrict__ in, float* __restrict__ out, long npieces) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npieces; i += (long)gridDim.x * blockDim.x) {
    PieceView<half> v;
    v.p = MN_LOAD_LAST(reinterpret_cast<const piece_t*>(in) + i);
    PieceView<float> a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a.e[e] = (float)v.e[e];
      b.e[e] = (float)v.e[4 + e];
    }
    reinterpret_cast<piece_t*>(out)[2 * i] = a.p;
    reinterpret_cast<piece_t*>(out)[2 * i + 1] = b.p;
  }
}

// global average pool of an h2 activation
static __global__ void __launch_bounds__(256) avgpool_fwd_h2_kernel(const half* __restrict__ in, float* __restrict__ out, int B,
                                                                    int HW, int C, int q) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int c = i % C, b = i / C;
  float s = 0.f;
  for (int p = 0; p < HW; ++p) {
    const half* ph = in + h2_index((long)b * HW + p, C, c);
    if (q)
      s += (float)ph[0] + fp8_e4m3_to_float(reinterpret_cast<const unsigned char*>(in)[h2q_byte((long)b * HW + p, C, c, 0)]) *
                              pow2c(-kQA_LO);
    else
      s += (float)ph[0] + (float)ph[32];
  }
  out[i] = s / (float)HW;
}

// g[b][p][c] = gp[b][c] / HW (fp32), zeroed where the pooled h2 activation is <= 0 (its hi half carries the sign)
static __global__ void __launch_bounds__(256) avgpool_bwd_h2_kernel(const float* __restrict__ gp, float* __restrict__ g, int B,
                                                                    int HW, int C, const half* __restrict__ gate) {
  const long total = (long)B * HW * C;
  const float inv = 1.f / (float)HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long row = i / C;
    const int b = (int)(row / HW);
    float v = gp[b * C + c] * inv;
    if (gate && !((float)gate[h2_index(row, C, c)] > 0.f)) v = 0.f;
    g[i] = v;
  }
}

}  // namespace mn
