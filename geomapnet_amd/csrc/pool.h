// Max-pool 3x3 / stride 2 / pad 1 on NHWC activations, with the torch routing rule (the gradient
// of a window goes to its FIRST maximum in (r, s) scan order).  The forward pass records the winning
// tap (0..8) per output element as one byte, so the backward pass is a pure gather over at most four
// windows per input position and never re-reads the activation: traffic = indices + output gradient
// (each read ~2.25x through L2) + one write of the input gradient.
#pragma once
#include "common.h"

namespace mn {

template <typename T>
static __global__ void __launch_bounds__(256) maxpool_fwd_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                                  unsigned char* __restrict__ idx, int B, int H, int W,
                                                                  int C, int Po, int Qo) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int cpr = C / VEC;
  long total = (long)B * Po * Qo * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int cp = (int)(i % cpr);
    long tmp = i / cpr;
    int qo = (int)(tmp % Qo);
    tmp /= Qo;
    int po = (int)(tmp % Po);
    int b = (int)(tmp / Po);
    float best[VEC];
    unsigned char arg[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      best[e] = -INFINITY;
      arg[e] = 0;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      int h = po * 2 - 1 + r;
      if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        int w = qo * 2 - 1 + s;
        if ((unsigned)w >= (unsigned)W) continue;
        PieceView<T> v;
        v.p = reinterpret_cast<const piece_t*>(in)[((long)(b * H + h) * W + w) * cpr + cp];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float f = (float)v.e[e];
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

template <typename T>
static __global__ void __launch_bounds__(256) maxpool_bwd_kernel(const unsigned char* __restrict__ idx,
                                                                  const T* __restrict__ gout, T* __restrict__ gin, int B,
                                                                  int H, int W, int C, int Po, int Qo) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int cpr = C / VEC;
  long total = (long)B * H * W * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int cp = (int)(i % cpr);
    long tmp = i / cpr;
    int w = (int)(tmp % W);
    tmp /= W;
    int h = (int)(tmp % H);
    int b = (int)(tmp / H);
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    // windows (po, qo) with po*2-1 <= h <= po*2+1: po in [h/2, (h+1)/2]
    for (int po = h / 2; po <= (h + 1) / 2 && po < Po; ++po)
      for (int qo = w / 2; qo <= (w + 1) / 2 && qo < Qo; ++qo) {
        const unsigned char mytap = (unsigned char)((h - (po * 2 - 1)) * 3 + (w - (qo * 2 - 1)));
        const long o = ((long)(b * Po + po) * Qo + qo) * cpr + cp;
        unsigned char a[VEC];
        if (VEC == 8) {
          unsigned long long packed = reinterpret_cast<const unsigned long long*>(idx)[o];
#pragma unroll
          for (int e = 0; e < VEC; ++e) a[e] = (unsigned char)(packed >> (8 * e));
        } else {
          unsigned packed = reinterpret_cast<const unsigned*>(idx)[o];
#pragma unroll
          for (int e = 0; e < VEC; ++e) a[e] = (unsigned char)(packed >> (8 * e));
        }
        PieceView<T> gv;
        gv.p = reinterpret_cast<const piece_t*>(gout)[o];
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (a[e] == mytap) acc[e] += (float)gv.e[e];
      }
    PieceView<T> ov;
#pragma unroll
    for (int e = 0; e < VEC; ++e) ov.e[e] = (T)acc[e];
    reinterpret_cast<piece_t*>(gin)[i] = ov.p;
  }
}

// Gradient of one input position of the max-pool, gathered on the fly: sum of the output gradients of the (at
// most four) windows whose recorded argmax is this position; rounded to T like the stored gradient would be.
// Position (h, w) lies in windows po in {h/2, (h+1)/2}, qo in {w/2, (w+1)/2} (one per dimension for even coordinates,
// two for odd ones).  All four candidates are requested BEFORE any of them is used (clamped addresses, a validity flag
// each): as a loop over the valid windows every load waited for the previous one -- four L2 round trips per 16 bytes of
// gradient, which made the gathering BatchNorm reduction 2.3x slower than reading a materialised gradient (round 2).
template <typename T>
__device__ __forceinline__ void pool_grad_piece(const unsigned char* __restrict__ idx, const T* __restrict__ gout, int b,
                                                int h, int w, int cp, int cpr, int Po, int Qo,
                                                float (&acc)[ElemTraits<T>::VEC]) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int po0 = h >> 1, po1 = (h + 1) >> 1, qo0 = w >> 1, qo1 = (w + 1) >> 1;
  const bool vp1 = po1 != po0 && po1 < Po, vq1 = qo1 != qo0 && qo1 < Qo;
  const int pos[2] = {po0, vp1 ? po1 : po0}, qos[2] = {qo0, vq1 ? qo1 : qo0};
  const bool valid[4] = {true, vq1, vp1, vp1 && vq1};
  PieceView<T> gv[4];
  unsigned long long packed[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long o = ((long)(b * Po + pos[k >> 1]) * Qo + qos[k & 1]) * cpr + cp;
    if (VEC == 8)
      packed[k] = reinterpret_cast<const unsigned long long*>(idx)[o];
    else
      packed[k] = reinterpret_cast<const unsigned*>(idx)[o];
    gv[k].p = reinterpret_cast<const piece_t*>(gout)[o];
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int po = pos[k >> 1], qo = qos[k & 1];
    const unsigned mytap = (unsigned)((h - (po * 2 - 1)) * 3 + (w - (qo * 2 - 1)));
#pragma unroll
    for (int e = 0; e < VEC; ++e)
      if (valid[k] && ((unsigned)(packed[k] >> (8 * e)) & 0xffu) == mytap) acc[e] += (float)gv[k].e[e];
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = (float)(T)acc[e];
}

// Source of the incoming gradient of a BatchNorm backward pass when it is the max-pool's input gradient:
// (argmax bytes, pooled gradient) instead of a materialised tensor
struct PoolGradSrc {
  const unsigned char* idx = nullptr;  // null: read the gradient tensor
  const void* gout = nullptr;
  int H = 0, W = 0, Po = 0, Qo = 0;
};

}  // namespace mn
