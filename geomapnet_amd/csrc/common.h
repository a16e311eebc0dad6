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

// Last-use loads of the streaming kernels are NON-TEMPORAL: the BatchNorm forward apply's conv output (read again only in the
// backward pass, a whole network later) and the backward apply's incoming gradient and conv output (never read again) do not
// have to displace what the next convolution is about to read from L2 / Infinity Cache.  Whole step 13.59 -> 13.49 ms and
// 14.33 -> 14.28 ms on two boxes (profiles/r03/c31_*, c32_*); making the backward REDUCTION's loads and the forward residual
// non-temporal as well (they are read again by the apply kernel that follows / by the backward pass) measured no better.
// The stem max-pool's input and the fused weight gradient's partial slabs as well: slightly slower (14.36 -> 14.40 ms, c33_*).
#define MN_LOAD_LAST(p) __builtin_nontemporal_load(p)

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
//               2.5 PF pipe.  Used for the forward pass (operands are O(1): activations and weights; the lo halves of small
//               values are subnormal fp16 numbers, which v_mfma_f32_32x32x16_f16 honours -- measured,
//               tools/probes/mfma_denorm_probe.hip -- so their absolute error stays below 2^-25).
//   MMA_BF16X3  the same with bf16 halves (x = hi + lo to ~2^-16, fp32's exponent range: no scaling, nothing can over- or
//               underflow).  Used for the backward pass, whose operands (activation gradients) span many binades.
//   MMA_H2      (fp16-PAIR tensors, "h2", round 4; kernels instantiated for `half`): the operands are ALREADY split -- every
//               conv-consumed tensor is stored as hi + lo fp16 halves (layout below), so tiles travel HBM -> LDS by DMA like
//               fp16 tiles and a product is the same three v_mfma_f32_32x32x16_f16, with no conversion in the K loop.
//   MMA_H2Q     (round 5, the fp16x2q mode's FORWARD convolutions; h2q tensors below): hi*hi on the fp16 pipe as MMA_H2, and BOTH cross
//               terms of a 32-channel K-step in ONE v_mfma_scale_f32_32x32x64_f8f6f4 -- the block-scaled fp8 MFMA, twice the fp16
//               rate -- from fp8 (e4m3) copies: [lo8 | hi8] of A against [hi8 | lo8] of B.  A cross term is a 2^-11 correction, so
//               the 2^-4 of an fp8 operand leaves ~2^-15 per product: poses 3.1e-4 from the fp32 oracle at the benchmark shape
//               (tools/fp8_cross_budget.py; bar 1e-3; MMA_H2: 1.6e-5), at 2 instead of 3 MFMA-equivalents per product.
enum { MMA_NATIVE = 0, MMA_F16X3 = 1, MMA_BF16X3 = 2, MMA_H2 = 3, MMA_H2Q = 4 };

// ---- the h2 ("fp16 pair") tensor layout ---------------------------------------------------------------------------------
// A tensor [rows][C] of fp32-class values, C a multiple of 32, 4 bytes per element like fp32: per row and per GROUP of 32
// channels 64 bytes of hi halves (rn/rtz fp16 of x) followed by 64 bytes of lo halves (fp16 of x - hi):
//     half index of (row, c) = row * 2C + (c / 32) * 64 + (c % 32)   [hi]      + 32   [lo]
// x = hi + lo to 2^-22 relative while lo is a normal fp16 number, 2^-25 absolute below that; |x| must stay below 65504
// (activations are O(1); gradients are kept in range by the loss scale and the overflow guard of the fp16 mode).
// Why groups of 32: a 128-byte K-step of an MFMA kernel is then ONE group -- 16-k fragments hi[0:16], hi[16:32], lo[0:16],
// lo[16:32] -- and in the transposed (weight-gradient) use a 32-row MFMA operand block is 32 hi or 32 lo channels, so
// hi*hi + hi*lo + lo*hi accumulate into ONE 32x32 tile.  To every DMA / gather path the tensor is simply an fp16 tensor
// with 2C channels.
__host__ __device__ inline long h2_index(long row, int C, int c) { return row * 2L * C + (long)(c >> 5) * 64 + (c & 31); }

// ---- the h2q layout (round 5): an h2 tensor whose lo plane is fp8 -----------------------------------------------------------------
// Same 128 bytes per row and 32-channel group, same position of the fp16 hi halves (h2_index): every reader of hi halves -- gates,
// the backward pass's copies -- is unaffected.  The second 64 bytes hold two fp8 (OCP e4m3) planes of 32 bytes with FIXED exponents:
//   activations:  bytes 64..95  lo8[c] = fp8((x - hi) * 2^kQA_LO),   bytes 96..127 hi8[c] = fp8(hi * 2^kQA_HI)
//   weights:      bytes 64..95  hi8[c] = fp8(hi * 2^kQW_HI),         bytes 96..127 lo8[c] = fp8((w - hi) * 2^kQW_LO)
// (planes swapped between the operands: a lane's scaled-MFMA operand is piece 4 + half followed by piece 6 + half of the group, and
// block 0 of A -- lo8 -- must meet block 0 of B -- hi8; tools/probes/mfma_scale_probe2.hip.)  Ranges: |activation| < 896, |weight|
// < 7 (values beyond saturate in the fp8 copies only; the activation lo plane from |x| = 448 on); below 2^-6 of the scaled value the fp8 copy is subnormal, an absolute error
// that is negligible against the product it corrects.  E8M0 scale bytes of the MFMA: 127 - exponent.
constexpr int kQA_LO = 9, kQA_HI = -1, kQW_HI = 6, kQW_LO = 16;  // (kQA_LO: the device splits with round-toward-zero hi halves, lo < 2^-10 |x|)
constexpr float pow2c(int e) { return e >= 0 ? (float)(1ll << e) : 1.f / (float)(1ll << -e); }
__host__ __device__ inline long h2q_byte(long row, int C, int c, int plane) {  // byte offset of fp8 plane 0 / 1 of element (row, c)
  return row * 4L * C + (long)(c >> 5) * 128 + 64 + plane * 32 + (c & 31);
}
// fp32 -> fp8 e4m3 (OCP, round to nearest even, saturating at +-448), portable form (host, emulator)
__host__ __device__ inline unsigned char fp8_e4m3_from_float(float f) {
  unsigned u;
  __builtin_memcpy(&u, &f, 4);
  const unsigned sign = (u >> 24) & 0x80u;
  float a = f < 0.f ? -f : f;
  if (!(a == a)) return (unsigned char)(sign | 0x7f);
  if (a >= 448.f) return (unsigned char)(sign | 0x7e);
  if (a < 0.0009765625f) return (unsigned char)sign;  // below half the smallest subnormal (2^-10): zero
  int e;
  (void)__builtin_frexpf(a, &e);  // a = m * 2^e, m in [0.5, 1)
  int ex = e - 1;                 // a = 1.xxx * 2^ex
  if (ex < -6) ex = -6;           // subnormals share the exponent of the smallest normal
  const float q = __builtin_ldexpf(a, 3 - ex);  // units of the last place
  float r = __builtin_rintf(q);                 // round to nearest even
  int mant = (int)r;
  if (ex == -6 && mant < 8) return (unsigned char)(sign | mant);  // subnormal: exponent field 0
  if (mant == 16) {
    mant = 8;
    ex += 1;
  }
  if (ex > 8 || (ex == 8 && mant > 14)) return (unsigned char)(sign | 0x7e);
  return (unsigned char)(sign | ((ex + 7) << 3) | (mant - 8));
}
__host__ __device__ inline float fp8_e4m3_to_float(unsigned char b) {
  const int ef = (b >> 3) & 15, m = b & 7;
  float v = ef == 0 ? __builtin_ldexpf((float)m, -9) : __builtin_ldexpf((float)(8 + m), ef - 10);
  if (ef == 15 && m == 7) v = __builtin_nanf("");
  return (b & 0x80) ? -v : v;
}

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
// x[0..7] = hi + lo.  fp16: hi = rn16(x), lo = rn16(x - hi) (the difference is exact in fp32): |x - hi - lo| <= 2^-22 |x|
// while lo is a normal fp16 number, <= 2^-25 absolute below that.
#if defined(__HIP_DEVICE_COMPILE__)  // (the emulator build and hipcc's host pass take the portable form below)
// Hand-picked instructions (measured on MI355X, round 3: whole fp32x3 step 43.3 -> 42.1 ms, layer3 forward 340 -> 313 us):
// hipcc lowers the portable form below to ~5 VALU operations per element (v_cvt_f32_f16 +
// v_sub_f32 + conversions; its SLP pass pairs the subtractions into v_pk_fma_f32 and with that loses the fused
// conversions).  Here a pair of elements costs 3: v_cvt_pkrtz_f16_f32 for the hi halves -- round-toward-zero is as good as
// any rounding, the residual below is exact either way -- and one v_fma_mix{lo,hi}_f16 per lo half, which computes
// x - hi in fp32 and rounds it to fp16 in one operation.
// The statements are volatile so that they stay in program order ahead of x3_fence() (below).
// (v_fma_mixlo_f16 writes bits 15:0 of its destination and keeps 31:16, v_fma_mixhi_f16 the reverse: lo first, then hi.)
__device__ __forceinline__ void split8_f16(const float (&x)[8], half8& hi, half8& lo) {
  union {
    unsigned u[4];
    half8 v;
  } H, L;
  // ONE statement per fragment (hipcc pads every asm statement boundary with an s_nop; VALU -> VALU dependences inside
  // the string are interlocked by the hardware).  Outputs are early-clobber: they are written while inputs are still live.
#pragma unroll
  for (int p = 0; p < 4; ++p) H.u[p] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x[2 * p], x[2 * p + 1]));
  asm volatile(
      "v_fma_mixlo_f16 %0, %4, 1.0, -%12 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %1, %6, 1.0, -%13 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %2, %8, 1.0, -%14 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %3, %10, 1.0, -%15 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %5, 1.0, -%12 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %1, %7, 1.0, -%13 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %2, %9, 1.0, -%14 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %3, %11, 1.0, -%15 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(L.u[0]), "=&v"(L.u[1]), "=&v"(L.u[2]), "=&v"(L.u[3])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(H.u[0]), "v"(H.u[1]),
        "v"(H.u[2]), "v"(H.u[3]));
  hi = H.v;
  lo = L.v;
}
// The MFMAs that consume split operands must not issue in the two wait states behind the VALU instruction that wrote
// them; hipcc pads that hazard for instructions it knows, not for the inside of an asm statement.  Call once after the
// last split of a group, before the first MFMA (the statements above are volatile: they stay ahead of this one).
__device__ __forceinline__ void x3_fence() {
  asm volatile("s_nop 1");
  __builtin_amdgcn_sched_barrier(0);
}
#else
__device__ __forceinline__ void split8_f16(const float (&x)[8], half8& hi, half8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const half h = (half)x[e];
    hi[e] = h;
    lo[e] = (half)(x[e] - (float)h);
  }
}
__device__ __forceinline__ void x3_fence() {}
#endif
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

// compile-time loop: f(StaticIndex<0>{}), ..., f(StaticIndex<N - 1>{}) -- for bodies whose register or immediate operands must be
// constants of the iteration (hand-counted waits, register-ring slots)
template <int I>
struct StaticIndex {
  static constexpr int value = I;
};
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(StaticIndex<I>{});
    static_for<N, I + 1>(f);
  }
}

template <typename T>
__device__ __forceinline__ float to_f(T v) {
  return (float)v;
}

}  // namespace mn

// Host-side helpers -------------------------------------------------------------------------
namespace mn {
inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
}
