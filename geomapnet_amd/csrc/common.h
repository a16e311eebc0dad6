// Shared device-side definitions for the MapNet gfx950 kernels.
//
// Data conventions:
//   * activations / activation-gradients: NHWC, element type T in {float, _Float16}
//   * conv weights in compute layout: [Cout][R][S][Cin] ("OHWI", K contiguous per output channel)
//   * every global<->LDS move is a 16-byte "piece" (8 halves or 4 floats)
//   * wavefront = 64 lanes, workgroup = 256 threads = 4 waves unless stated otherwise
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mn {

typedef _Float16 half;
typedef half half8 __attribute__((ext_vector_type(8)));
typedef half half4 __attribute__((ext_vector_type(4)));
typedef half half2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef uint32_t piece_t __attribute__((ext_vector_type(4)));  // one 16-byte piece

enum { MN_F32 = 0, MN_F16 = 1 };

// Matrix-core arithmetic of the fp32-STORAGE kernels (igemm.h, wgrad.h; GatherGeom::mma):
//   MMA_NATIVE  v_mfma_f32_32x32x2_f32, an exact fp32 FMA chain (157 TF peak)
//   MMA_F16X3   every fp32 operand is split in registers into hi + lo fp16 halves (x*s = hi + lo to ~2^-22) and a product
//               costs three v_mfma_f32_32x32x16_f16 (hi*hi + hi*lo + lo*hi, fp32 accumulate): fp32-class results on the
//               2.5 PF pipe.  Used for the forward pass (operands are O(1): activations and weights, the latter pre-scaled
//               by kX3WeightScale so that their lo halves stay normal fp16 numbers).
//   MMA_BF16X3  the same with bf16 halves (x = hi + lo to ~2^-16, fp32's exponent range: no scaling, nothing can over- or
//               underflow).  Used for the backward pass, whose operands (activation gradients) span many binades.
enum { MMA_NATIVE = 0, MMA_F16X3 = 1, MMA_BF16X3 = 2 };
constexpr float kX3WeightScale = 256.f;

template <typename T>
struct ElemTraits;
template <>
struct ElemTraits<float> {
  static constexpr int VEC = 4;  // elements per 16-byte piece
  static constexpr int DTYPE = MN_F32;
};
template <>
struct ElemTraits<half> {
  static constexpr int VEC = 8;
  static constexpr int DTYPE = MN_F16;
};

__device__ __forceinline__ piece_t zero_piece() {
  piece_t z = {0u, 0u, 0u, 0u};
  return z;
}

// piece <-> element views (bit casts through unions keep hipcc on plain register moves)
template <typename T>
union PieceView;
template <>
union PieceView<float> {
  piece_t p;
  float e[4];
  floatx4 v;
};
template <>
union PieceView<half> {
  piece_t p;
  half e[8];
  half8 v;
};

// 8 bytes = four halves (register epilogues: halo_pp.h, igemm_rt.h)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
union Half4View {
  u32x2 p;
  half e[4];
};

// ---- fp32 -> (hi, lo) operand splits of the x3 matrix-core modes -----------------------------------------
// x[0..7] * s = hi + lo.  fp16: hi = rn16(x s), lo = rn16(x s - hi) (the difference is exact in fp32): |x s - hi - lo| <=
// 2^-22 |x s| while lo is a normal fp16 number, <= 2^-25 absolute below that.
__device__ __forceinline__ void split8_f16(const float (&x)[8], float s, half8& hi, half8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = x[e] * s;
    const half h = (half)v;
    hi[e] = h;
    lo[e] = (half)(v - (float)h);
  }
}
// bf16: 8 + 8 significand bits, fp32's exponent range
__device__ __forceinline__ void split8_bf16(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h = (__bf16)x[e];
    hi[e] = h;
    lo[e] = (__bf16)(x[e] - (float)h);
  }
}

// ---- wave reductions (64 lanes) ---------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ int ceil_div(int a, int b) { return (a + b - 1) / b; }

template <typename T>
__device__ __forceinline__ float to_f(T v) {
  return (float)v;
}

}  // namespace mn

// Host-side helpers -------------------------------------------------------------------------
namespace mn {
inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
}
