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
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef uint32_t piece_t __attribute__((ext_vector_type(4)));  // one 16-byte piece

enum { MN_F32 = 0, MN_F16 = 1 };

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
