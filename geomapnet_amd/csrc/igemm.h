// Implicit-GEMM convolution on the gfx950 matrix cores (forward, data-gradient, linear layers).
//
//   out[m][n] = sum_k A_gather[m][k] * Bw[n][k]        m = (b, p, q)  k = (r, s, c)
//
// A is an NHWC activation (or activation-gradient) tensor read through a parametrised gather
// (GatherGeom) that covers: forward conv of any stride/pad, the transposed gather of the data
// gradient (rsign = -1, div = stride), 1x1 projections, and plain GEMMs (R=S=Hi=Wi=1).
// Bw is the weight matrix with K contiguous per output channel ("OHWI" for forward,
// [Cin][R][S][Cout] for the data gradient).
//
// Structure (one kernel template, several tile configurations):
//   * WM x WN waves (64 lanes each); every wave owns a 64 x (TN*32) sub-tile made of 32x32 MFMA tiles
//     (v_mfma_f32_32x32x16_f16 for half, v_mfma_f32_32x32x2_f32 for float; fp32 accumulate);
//     block tile BM x BN = (WM*64) x (WN*TN*32).
//   * Tiles go global -> LDS directly with global_load_lds_dwordx4 (LDS-DMA: no VGPR staging, no ds_write
//     pass) into an NBUF-deep ring with NBUF-1 tiles in flight, one raw s_barrier per K-step and counted
//     s_waitcnt vmcnt (cdna_hip_programming.md section 5, "glds").  K advances NP 16-byte pieces per row
//     per step.  The DMA writes LDS linearly (wave-uniform base + lane*16), so the XOR swizzle that keeps
//     ds_read_b128 conflict-free is applied to the SOURCE piece a lane fetches and to the slot a fragment
//     read addresses (same involution on both sides).  Taps outside the image fetch from `zero_page`.
//   * Gather cost: each thread precomputes, once, the element offset of its rows and a bit mask of the taps
//     that fall inside the image; a K-step then costs one shift/and/add per row.
//   * Workgroup ids are remapped so each XCD (private L2) walks a contiguous range of output tiles with the N
//     tiles of one M tile adjacent (they share the gathered A rows).
//   * Why large tiles: a 128x128 tile moves 64 B of L1->LDS traffic per 4096 MACs and saturates the CU's
//     64 B/clk vector-memory path at ~27 % of the MFMA peak (measured); 256-row tiles
//     cut the bytes per MAC by 25-50 %.
//   * Epilogue: alpha, bias, ReLU and the BatchNorm column partial sums in registers; the tile is then staged
//     through LDS (fp32, 64x128 sub-blocks) so that global stores -- and the residual / ReLU-gate loads of the
//     data-gradient path -- are full 16-byte pieces along channels.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace mn {

struct GatherGeom {
  int B, Hi, Wi, C;  // A tensor [B][Hi][Wi][C]
  int P, Q;          // GEMM rows m = (b*P + p)*Q + q
  int R, S;          // taps; k = (r*S + s)*C + c     (R*S <= 32)
  int mul_p, mul_q;  // hn = p*mul_p + rsign*r + off_h ; wn = q*mul_q + ssign*s + off_w
  int rsign, ssign;
  int off_h, off_w;
  int div;  // 1, or 2 (requires rsign = ssign = -1): tap valid only if hn, wn even; input index = hn/2, wn/2
  int M, N, K;
  // K-walk order of the uniform tap walk (not part of the C ABI; set by launch_igemm): 0 = channels fastest
  // (k = (r, s, c) as stored), 1 = taps fastest: all R*S taps of one 128-byte channel chunk before the next chunk, so
  // that the R*S shifted re-reads of an activation row are close in time (L2 hits instead of refetches)
  int tap_inner = 0;
  // Tap subset (stride-2 data gradients decomposed by output parity, launch_dgrad_parity in net.hip): the gather walks
  // R x S taps, the WEIGHT walk maps them to taps (bt_r0 + 2 r, bt_s0 + 2 s) of a bt_S-wide full kernel, and a weight
  // row is ldb elements long (0 = K).  Needs the taps-fastest order.
  int ldb = 0, bt_on = 0, bt_r0 = 0, bt_s0 = 0, bt_S = 0;
  // matrix-core arithmetic for fp32 tensors (common.h MMA_*; not part of the C ABI of the operator entry points, which
  // set it from their dtype argument): ignored by the fp16 kernels
  int mma = MMA_NATIVE;
};

// n / d for 0 <= n < 2^31 without the ~35-instruction software division (Granlund-Montgomery round-up
// multiplier): q = (mulhi(n, mul) + n) >> shift.  The row -> (image, y, x) decomposition of the gather
// prologue runs once per row of every tile; for the short-K layers it was a third of the block's time.
struct FastDiv {
  unsigned mul, shift;
};
inline FastDiv make_fastdiv(int d) {
  int l = 0;
  while ((1u << l) < (unsigned)d) ++l;
  FastDiv f;
  f.mul = (unsigned)((((unsigned long long)1 << 32) * (((unsigned long long)1 << l) - (unsigned long long)d)) / (unsigned long long)d + 1);
  f.shift = (unsigned)l;
  return f;
}
__device__ __forceinline__ int fastdiv(int n, FastDiv f) {
  return (int)((__umulhi((unsigned)n, f.mul) + (unsigned)n) >> f.shift);
}
struct Epilogue {
  void* out;             // [M][ldc], element type T
  int ldc;
  float* stats;          // [grid_m][2][N] column partial sums (sum, sum of squares) or null
  double* stats_accum = nullptr;  // alternative: [stats_rows][2][N] fp64 accumulators, added to atomically (row =
  int stats_rows = 0;             // tile_m % stats_rows spreads the same-address contention); consumer sums the rows
  const float* bias;     // [N] or null
  int relu;              // max(0, .) after bias
  const void* res;       // residual [M][ldc] of type T or null
  const void* res_gate;  // if non-null the residual passes only where res_gate[m][n] > 0
  // if non-null the stored value (after alpha, bias, residual) is zeroed where out_gate[m][n] <= 0: the data gradient
  // of a block's first conv leaves the kernel already multiplied by the ReLU gate of the block BELOW it, so that none
  // of that gradient's three consumers has to read the gating activation again
  const void* out_gate = nullptr;
  // (round 6, the fp16x2m mode's fp16 backward launches) res_gate / out_gate are h2 tensors (common.h) of which the HI halves are
  // read: fp16(x) exactly, so the sign test is the one the plain fp16 copy gave; rows of 2 ldc halves, channel c at h2_index.
  // fp16 kernels (the h2 kernels always read their gates this way).
  bool gate_h2 = false;
  // output row map (parity classes of a stride-2 data gradient): GEMM row m = (b, p, q) over an om_P x om_Q grid is
  // stored at pixel (2p + om_a, 2q + om_b) of an om_H x om_W image (also applies to res / res_gate / out_gate)
  int om_on = 0, om_P = 0, om_Q = 0, om_H = 0, om_W = 0, om_a = 0, om_b = 0;
  FastDiv om_dq{0, 0}, om_dp{0, 0};
  float alpha;           // scale applied to the accumulator
  // (Measured and removed in round 3: BatchNorm-BACKWARD sums (sum g, sum g * y of the BatchNorm that consumes the stored
  // gradient next) taken in the row-wise store passes of this epilogue and igemm_halo.h's, to drop the reduction launches
  // that read the gradient back.  The extra 16-byte load of y per stored piece sits, latency exposed, in each of the 6-12
  // store passes of a tile: layer3 data gradients +29 us per launch against 19.9 us of reduction saved, layer2 +36 against
  // 34.9, whole step 14.94 vs 14.63 ms.  profiles/r03/c16_bn_backward_sums_in_dgrad_epilogue.txt.)
};

struct RowDiv {
  FastDiv q, p;  // divisors GatherGeom.Q and GatherGeom.P
};

template <typename T>
__device__ __forceinline__ void mma_piece(const PieceView<T>& a, const PieceView<T>& b, floatx16& c);
template <>
__device__ __forceinline__ void mma_piece<half>(const PieceView<half>& a, const PieceView<half>& b, floatx16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, b.v, c, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_piece<float>(const PieceView<float>& a, const PieceView<float>& b, floatx16& c) {
  // lane (i, kg) holds k = kg*4 + j; MFMA j contracts the pair {j, 4 + j}: any k order is valid as
  // long as A and B agree, and they do (same lane -> k map on both operands).
#pragma unroll
  for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.e[j], b.e[j], c, 0, 0, 0);
}

// x3 modes (common.h): one fragment = 8 consecutive k of one row = TWO fp32 pieces, split into hi / lo halves in registers
template <int MM>
struct X3Frag;
template <>
struct X3Frag<MMA_F16X3> {
  half8 hi, lo;
};
template <>
struct X3Frag<MMA_BF16X3> {
  bf16x8 hi, lo;
};
template <int MM>
__device__ __forceinline__ void x3_split(const PieceView<float>& p0, const PieceView<float>& p1, X3Frag<MM>& f) {
  const float x[8] = {p0.e[0], p0.e[1], p0.e[2], p0.e[3], p1.e[0], p1.e[1], p1.e[2], p1.e[3]};
  if constexpr (MM == MMA_F16X3)
    split8_f16(x, f.hi, f.lo);
  else
    split8_bf16(x, f.hi, f.lo);
}
// c += a b without the lo*lo term (2^-22 / 2^-16 of the product); small terms first
template <int MM>
__device__ __forceinline__ void x3_mma(const X3Frag<MM>& a, const X3Frag<MM>& b, floatx16& c) {
  if constexpr (MM == MMA_F16X3) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.lo, b.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.lo, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.hi, c, 0, 0, 0);
  } else {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, c, 0, 0, 0);
  }
}

// slot swizzle for rows of NP 16-byte pieces: rows r, r+1, .. of a 16-lane ds_read_b128 group land in
// distinct slots of the 256-byte bank row
template <int NP>
__device__ __forceinline__ int lds_swz(int row) {
  return NP == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3);
}

// XCD-aware tile order: workgroup id b runs on XCD b % 8 (observed, speed only); give each XCD a
// contiguous range of logical tiles.  Bijective for any total.
__device__ __forceinline__ int xcd_remap(int bid, int total) {
  const int q = total >> 3, r = total & 7;
  const int xcd = bid & 7, slot = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

typedef const void __attribute__((address_space(1)))* gas_ptr_t;
typedef void __attribute__((address_space(3)))* las_ptr_t;
__device__ __forceinline__ void dma16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gas_ptr_t)src, (las_ptr_t)lds_wave_base, 16, 0, 0);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// by-value helpers: __builtin_bit_cast applied directly to a vector ELEMENT expression reads element 0 (clang)
__device__ __forceinline__ unsigned f32_bits(float f) { return __builtin_bit_cast(unsigned, f); }
__device__ __forceinline__ float bits_f32(unsigned u) { return __builtin_bit_cast(float, u); }
// Buffer resource over a dense tensor (raw addressing, stride 0): a load whose byte offset reaches `bytes`
// returns zero.  Tensors on this path are far below the 4 GiB a 32-bit offset spans.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, long bytes) {
  const unsigned n = bytes > 0xfffffff0l ? 0xfffffff0u : (unsigned)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)n, 0x00020000);
}
// buffer_load_dwordx4 ... lds: lane l copies 16 bytes from (resource base + voff + soff) to
// (wave-uniform LDS base + 16*l)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (las_ptr_t)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}
// s_waitcnt simm16 (gfx9): vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]; only vmcnt waits
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

//
// SPL: the NBUF-1-ahead tile's DMA instructions are not issued in one burst after the barrier but spread over the
// K-step's MFMA sub-steps (behind each sub-step's MFMAs), so that the first fragment reads and MFMAs of a K-step do not
// queue behind the burst.
// ABL (timing experiments only, results are wrong; tools/conv_bench.py with MN_ABLATE): bit 0 = no DMA in the main
// loop, bit 1 = fragments from registers instead of ds_reads, bit 2 = no barrier in the main loop.
// (Measured and removed: a "rotated" loop -- K-step boundary before the last MFMA sub-step, first fragments of the next
// tile read there -- neutral on the 288x256, 128x128 and 128x64 configurations; a stream-K schedule -- equal (tile,
// K-step) ranges per resident workgroup with a two-party slab reduction -- 123 vs 120 us, 105 vs 103 us, 113 vs 118 us,
// whole step 19.8 vs 19.65 ms: every extra segment pays a gather prologue and a pipeline fill; profiles/r01.)
// MM (fp32 tensors only): MMA_F16X3 / MMA_BF16X3 run the contraction on the f16 / bf16 matrix pipe with every operand split
// into hi + lo halves after its LDS read, three MFMAs per product (common.h): the K-step's pieces are consumed in groups
// of four (16 k: lanes 0-31 take pieces 4 kp, 4 kp + 1, lanes 32-63 pieces 4 kp + 2, 4 kp + 3 -- any k order is valid as
// long as A and B agree).
// Main-loop ablation of the x3 form on MI355X (profiles/r03/c5_ablation_fp32x3_igemm_layer3.txt, layer3 forward, 346 us):
// no DMA 279 us, no operand splits 252, no MFMAs 225, neither splits nor MFMAs 155, nothing but prologue + epilogue 35 --
// MFMAs (~120 us), splits (~94) and DMA + fragment reads (~120) ADD almost linearly, as they did for the fp16 kernel in
// round 1: a wave alternates VALU-only and MFMA-only segments and the two workgroups of a CU do not settle half a K-step
// apart.  Forcing that alternation was tried and removed (round 3, c6_fp32x3_pingpong_experiment.txt): ONE 8-wave workgroup
// of two wave groups with a tile and an LDS ring each, a K-step = a LOAD phase (reads + splits) and an MFMA phase (MFMAs +
// DMA issue) separated by workgroup-wide barriers, group 1 one phase late -- parity-green and 2x SLOWER (layer3 306 -> 660
// us, whole step 40.9 -> 64.9 ms): every phase then lasts as long as the slowest of eight waves' waits.  What is left to
// try is software pipelining INSIDE a wave (the splits of group kp + 1 in the shadow of the MFMAs of group kp).
template <typename T, int WM, int WN, int TM, int TN, int NP, int NBUF, int MINW, bool UNI, bool SPL = false, int ABL = 0,
          int MM = MMA_NATIVE>
static __global__ void __launch_bounds__(WM* WN * 64, MINW) igemm_kernel(GatherGeom g, const T* __restrict__ A,
                                                                         const T* __restrict__ Bw, Epilogue ep, int grid_n,
                                                                         const T* __restrict__ zero_page, RowDiv rd) {
  constexpr int VEC = ElemTraits<T>::VEC;
  // h2 (MM = MMA_H2, T = half): g.C / g.K / g.ldb count the 2C fp16 "channels" of the pair layout (common.h) -- every gather
  // and DMA below is the fp16 kernel's; what differs is the MFMA order of a K-step and the epilogue (fp32 output)
  constexpr bool H2 = MM == MMA_H2 || MM == MMA_H2Q;
  constexpr bool HQ = MM == MMA_H2Q;  // h2q operands: both cross terms of a K-step in one scaled fp8 MFMA (igemm_halo.h Q)
  static_assert(!H2 || (sizeof(T) == 2 && NP == 8 && UNI), "h2: fp16 pieces, one 32-channel group per K-step");
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int WTM = TM * 32, WTN = TN * 32;
  constexpr int RPP = NT / NP;  // rows covered by one DMA pass of all threads
  // The A rows [0, BM) and B rows [BM, BM + BN) of a tile form ONE row space that the threads cover in IPT passes of RPP
  // rows; a pass may straddle the A/B boundary as long as the boundary falls between waves (the buffer resource of
  // a DMA instruction is wave-uniform).  Rows past BM + BN of the last pass are zero-filled by the bounds check (no
  // fetch) into LDS rows nobody reads, so thread counts that divide neither BM nor BN work (12 waves: 288 + 256 rows).
  constexpr int IPT = (BM + BN + RPP - 1) / RPP;  // DMA instructions per wave per tile
  constexpr int TILE_PIECES = IPT * RPP * NP;
  constexpr int D = NBUF - 1;  // tiles in flight
  static_assert(BM % (64 / NP) == 0, "the A/B boundary must fall between waves");
  static_assert(D >= 1 && D <= 3, "ring depth");
  static_assert(RPP % 16 == 0, "swizzle must not see the per-thread row stride");
  static_assert(NBUF * TILE_PIECES * 16 >= 64 * (BN < 128 ? BN : 128) * 4, "epilogue staging does not fit");
  // ONE LDS object (a second one makes hipcc drain vmcnt before every fragment read):
  // [ring of tiles][BatchNorm partial reduction: WM*BN*2 floats]
  __shared__ piece_t smem[NBUF * TILE_PIECES + WM * BN / 2];
  float* red = reinterpret_cast<float*>(&smem[NBUF * TILE_PIECES]);  // [WM][BN][2]

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);  // wave-uniform values stay in scalar registers
  const int wm = wave / WN, wn = wave % WN;
  const int KT = g.K / (NP * VEC);
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  constexpr int k0 = 0;
  const int k1 = KT;
  const int tile_m = tile / grid_n, tile_n = tile - tile_m * grid_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int pc = t % NP, lrow = t / NP;
  const int dsh = g.div == 2 ? 1 : 0;

  // Gather state.  Loads go through buffer resources (raw buffer addressing: 32-bit byte offsets, and any
  // offset >= the tensor size returns zero), so an out-of-image tap or an out-of-range row is just an
  // all-ones offset: no pointer select, no 64-bit address arithmetic in the K loop.
  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(A, (long)g.B * g.Hi * g.Wi * g.C * (long)sizeof(T));
  const int ldb = g.ldb ? g.ldb : g.K;
  const __amdgpu_buffer_rsrc_t rsrc_b = make_rsrc(Bw, (long)g.N * ldb * (long)sizeof(T));
  const int CP = g.C / VEC;  // pieces per tap
  // Source piece of this lane inside a K-step: the XOR swizzle is applied on the source side.  Rows of
  // one lane differ by multiples of RPP (a multiple of 16), which the swizzle ignores.
  const int src_piece = pc ^ lds_swz<NP>(lrow);
  // UNI: a K-step never straddles a tap (C is a multiple of the step), so the tap walk is wave-uniform and
  // lives in scalar registers; otherwise (stem: 16-byte taps) every lane walks its own piece.
  constexpr bool uni = UNI;
  // PACK: when at most three passes hold A rows (the 12-wave 288x256 configuration) their tap masks share ONE register,
  // 10 bits each (launch_igemm only picks such a configuration for R*S <= 10).  That configuration is register-capped
  // at 168 and the two registers decide whether the allocator keeps a row offset in scratch -- whose reload, inside the
  // K loop, waits on vmcnt(0) behind the LDS-DMA queue.
  constexpr int IPT_A = (BM + RPP - 1) / RPP;  // passes that hold A rows
  constexpr bool PACK = UNI && WM == 3 && IPT_A <= 3;  // only the configuration whose launch condition bounds R*S
  constexpr int PACK_BITS = 10;
  unsigned d_off[IPT], a_inv[PACK ? 1 : IPT];  // byte offset of the row (A: image pixel, B: weight row) and the A tap mask
  if constexpr (PACK) a_inv[0] = ~0u;
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int jr = lrow + i * RPP;  // joint row of this thread in pass i
    d_off[i] = 0u;
    if constexpr (!PACK) a_inv[i] = ~0u;
    const bool pure_a = (i + 1) * RPP <= BM, pure_b = i * RPP >= BM;
    if (pure_a || (!pure_b && jr < BM)) {
      const int m = m0 + jr;
      if (m < g.M) {
        const int tmp = fastdiv(m, rd.q), q = m - tmp * g.Q;
        const int b = fastdiv(tmp, rd.p), p = tmp - b * g.P;
        const int h0 = p * g.mul_p + g.off_h, w0 = q * g.mul_q + g.off_w;
        d_off[i] = (unsigned)((((b * g.Hi + (h0 >> dsh)) * g.Wi + (w0 >> dsh)) * g.C) * (int)sizeof(T)) +
                   (uni ? (unsigned)src_piece * 16u : 0u);
        // validity is separable: tap (r, s) is inside the image iff row r and column s both are
        unsigned cinv = 0;  // bit s set: column tap s outside
        for (int s = 0; s < g.S; ++s) {
          int wn_ = w0 + g.ssign * s;
          bool ok = true;
          if (dsh) {
            ok = (wn_ & 1) == 0;
            wn_ >>= 1;
          }
          ok = ok && (unsigned)wn_ < (unsigned)g.Wi;
          cinv |= (ok ? 0u : 1u) << s;
        }
        const unsigned call = (1u << g.S) - 1u;
        unsigned inv = 0;
        for (int r = 0; r < g.R; ++r) {
          int hn = h0 + g.rsign * r;
          bool ok = true;
          if (dsh) {
            ok = (hn & 1) == 0;
            hn >>= 1;
          }
          ok = ok && (unsigned)hn < (unsigned)g.Hi;
          inv |= (ok ? cinv : call) << (r * g.S);
        }
        const unsigned full = inv | (g.R * g.S < 32 ? ~0u << (g.R * g.S) : 0u);
        if constexpr (PACK) {
          const int sh = (i < IPT_A ? i : 0) * PACK_BITS;
          const unsigned field = ((1u << PACK_BITS) - 1u) << sh;
          a_inv[0] = (a_inv[0] & ~field) | ((full << sh) & field);
        } else {
          a_inv[i] = full;
        }
      }
    } else {
      const int br = jr - BM, n = n0 + br;
      d_off[i] = (n < g.N && br < BN) ? (unsigned)(n * ldb) * (unsigned)sizeof(T) + (unsigned)src_piece * 16u : ~0u;
    }
  }
  // running decomposition of the step's first piece (uniform walk) or of this lane's piece (per-lane walk)
  // into (tap = (tr, ts), channel piece cpi)
  int cpi = uni ? 0 : src_piece, tr = 0, ts = 0, tap = 0;
  while (cpi >= CP) {
    cpi -= CP;
    ++tap;
    if (++ts == g.S) {
      ts = 0;
      ++tr;
    }
  }
  unsigned b_step = (unsigned)k0 * NP * 16;  // byte offset of the K-step inside a weight row (scalar)
  const bool tapin = UNI && g.tap_inner != 0;
  const unsigned lds_wave = wave * 64;  // this wave's 64 pieces of DMA pass 0 inside a tile

  // instructions [j0, j1) of the tile's IPT DMA instructions (A passes first, then B passes); the tap walk advances
  // when the last one has been issued
  auto issue_part = [&](int buf, int j0, int j1) {
    piece_t* base = &smem[buf * TILE_PIECES + lds_wave];
    const unsigned toff =
        (unsigned)(((g.rsign * (tr >> dsh) * g.Wi + g.ssign * (ts >> dsh)) * g.C + cpi * VEC) * (int)sizeof(T));
    const int btap = g.bt_on ? (g.bt_r0 + 2 * tr) * g.bt_S + g.bt_s0 + 2 * ts : tap;
    const unsigned bs = tapin ? (unsigned)(btap * CP + cpi) * 16u : b_step;
#pragma unroll
    for (int i = 0; i < IPT; ++i)
      if (i >= j0 && i < j1) {
        const bool pure_a = (i + 1) * RPP <= BM, pure_b = i * RPP >= BM;
        piece_t* dst = base + i * (RPP * NP);
        if (pure_a || (!pure_b && wave * (64 / NP) + i * RPP < BM)) {  // wave-uniform
          // all ones where the tap is outside the image: the buffer bounds check then returns zero
          const unsigned inv = PACK ? (unsigned)__builtin_amdgcn_sbfe(a_inv[0], tap + (i < IPT_A ? i : 0) * PACK_BITS, 1)
                                    : (unsigned)__builtin_amdgcn_sbfe(a_inv[PACK ? 0 : i], tap, 1);
          dma16(rsrc_a, (d_off[i] + toff) | inv, 0u, dst);
        } else {
          dma16(rsrc_b, d_off[i], bs, dst);
        }
      }
    if (j1 == IPT) {
      b_step += NP * 16;
      if (tapin) {
        ++tap;
        if (++ts == g.S) {
          ts = 0;
          ++tr;
        }
        if (tap == g.R * g.S) {
          tap = tr = ts = 0;
          cpi += NP;
        }
      } else {
        cpi += NP;
        while (cpi >= CP) {
          cpi -= CP;
          ++tap;
          if (++ts == g.S) {
            ts = 0;
            ++tr;
          }
        }
      }
    }
  };
  auto issue_tile = [&](int buf) { issue_part(buf, 0, IPT); };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int NK = k1 - k0;  // K-steps of this segment
#pragma unroll
  for (int j = 0; j < D; ++j)
    if (j < NK) issue_tile(j);
  int cur = 0, nxt = D % NBUF;
  // x3, software-pipelined form (128-byte K-steps = two groups of 16 k, two buffers): the operand splits of a group run in
  // the shadow of the MFMAs of the group before it -- after every MFMA triple (96 matrix-pipe cycles) one fragment of the
  // NEXT group is split (12 VALU operations) -- and the K-step's barrier sits between its two groups, so that the first
  // group of tile kt + 1 can be split under the MFMAs of the second group of tile kt:
  //     [M(kt, 0) | C(kt, 1)]  wait + barrier  DMA(kt + 2)  [M(kt, 1) | C(kt + 1, 0)]
  // (the form without this overlap: layer3 forward 313 us of which ~94 us are the splits, c5_ablation_fp32x3_*.txt)
  constexpr bool X3_PIPE = MM != MMA_NATIVE && !H2 && NP == 8 && NBUF == 2 && !SPL && ABL == 0;
  if constexpr (X3_PIPE) {
    constexpr int P = TM * TN, F = TM + TN;
    X3Frag<MM> xa[2][TM], xb[2][TN];  // [group parity]
    PieceView<float> ra[TM][2], rb[TN][2];
    auto load_raw = [&](int buf, int kp) {
      const piece_t* ta = &smem[buf * TILE_PIECES];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pxs = (kp * 4 + (lane >> 5) * 2 + h) ^ lds_swz<NP>(lane & 31);
        const piece_t* pa = ta + (wm * WTM + (lane & 31)) * NP + pxs;
        const piece_t* pb = ta + (BM + wn * WTN + (lane & 31)) * NP + pxs;
#pragma unroll
        for (int i = 0; i < TM; ++i) ra[i][h].p = pa[i * 32 * NP];
#pragma unroll
        for (int j = 0; j < TN; ++j) rb[j][h].p = pb[j * 32 * NP];
      }
    };
    auto split_frag = [&](int f, int dst) {  // fragment f of the raw registers -> group slot dst (f, dst: unrolled constants)
      if (f < TM)
        x3_split<MM>(ra[f][0], ra[f][1], xa[dst][f]);
      else
        x3_split<MM>(rb[f - TM][0], rb[f - TM][1], xb[dst][f - TM]);
    };
    // MFMAs of group slot `src`; with `split`, the raw registers are split into slot src ^ 1 along the way
    auto mma_group = [&](int src, bool split) {
#pragma unroll
      for (int pth = 0; pth < P; ++pth) {
        x3_mma<MM>(xa[src][pth / TN], xb[src][pth % TN], acc[pth / TN][pth % TN]);
        __builtin_amdgcn_sched_barrier(0);
        if (split) {
#pragma unroll
          for (int f = pth * F / P; f < (pth + 1) * F / P; ++f) split_frag(f, src ^ 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (split) x3_fence();
    };
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // tile 0 visible
    if (1 < NK) issue_tile(1);
    load_raw(0, 0);
#pragma unroll
    for (int f = 0; f < F; ++f) split_frag(f, 0);
    x3_fence();
    for (int kt = 0; kt < NK; ++kt) {
      const int cb = kt & 1;
      load_raw(cb, 1);
      mma_group(0, true);  // M(kt, 0) | C(kt, 1)
      const bool next = kt + 1 < NK;
      if (next) wait_vmcnt<0>();  // this wave's pieces of tile kt + 1, requested one K-step ago
      __builtin_amdgcn_s_barrier();  // tile kt + 1 visible to all waves; every read of tile kt has been consumed
      if (kt + 2 < NK) issue_tile(cb);
      if (next) load_raw(cb ^ 1, 0);
      mma_group(1, next);  // M(kt, 1) | C(kt + 1, 0)
    }
  } else
  for (int kt = 0; kt < NK; ++kt) {
    // tiles issued so far: min(NK, kt + D); tile kt must have landed
    const int ahead = min(NK, kt + D) - (kt + 1);
    if (D >= 3 && ahead >= 2)
      wait_vmcnt<2 * IPT>();
    else if (D >= 2 && ahead >= 1)
      wait_vmcnt<IPT>();
    else
      wait_vmcnt<0>();
    if (!(ABL & 4)) __builtin_amdgcn_s_barrier();  // tile kt visible to all waves; everyone is done reading buffer `nxt`
    const bool more = (ABL & 1) ? false : kt + D < NK;
    // (Issuing behind the K-step's first MFMA group instead, which gains 2-5 % in igemm_halo.h and wgrad_fused.h, measured
    // neutral here and in the x3 pipelined loop: layer2 fp16 98.9 vs 98.8 us, whole step 14.28-14.37 vs 14.36-14.39 ms,
    // profiles/r03/c15_*.)
    if (!SPL && more) issue_tile(nxt);  // tiles are issued strictly in K order
    const piece_t* ta = &smem[cur * TILE_PIECES];
    // fragments are register double-buffered: the ds_reads of sub-step ks+1 are issued before the MFMAs
    // of sub-step ks so LDS latency hides under the matrix pipe
    // Fragment row = (wave tile origin, a multiple of 32) + 32 * tile + (lane & 31): the swizzle only looks at row
    // bits 1..3, so it is lds_swz(lane & 31) for every fragment, and the fragments of one operand differ by the constant
    // 32 * NP pieces -- two base addresses and immediate offsets instead of one address register per fragment (which
    // the 168-register 12-wave configuration spilled and reloaded -- behind s_waitcnt vmcnt(0), i.e. draining the
    // LDS-DMA queue -- inside this loop).
    // The bases are recomputed from the lane id every K-step (a handful of VALU operations): as loop invariants the
    // register-capped configurations spill them, and a spill reload sits behind s_waitcnt vmcnt(0).
#ifdef __HIP_DEVICE_COMPILE__
    unsigned opaque_zero = 0u;
    asm volatile("" : "+s"(opaque_zero));  // keeps the two instructions below inside the loop
    const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, opaque_zero));  // lane id
#else
    const int ln = lane;
#endif
    if constexpr (MM != MMA_NATIVE && !H2) {
      static_assert(sizeof(T) == 4 && NP % 4 == 0, "x3 modes: fp32 tensors, whole groups of four pieces");
      // (ABL, timing experiments: bits 0-2 as above; bit 3 = no operand splits, bit 4 = no MFMAs)
      constexpr int NKP = NP / 4;
      PieceView<float> ra[TM][2], rb[TN][2];
      auto load_raw = [&](int kp) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if constexpr ((ABL & 2) != 0) {
            const piece_t fake = {(unsigned)ln, (unsigned)kt, (unsigned)kp, (unsigned)h};
#pragma unroll
            for (int i = 0; i < TM; ++i) ra[i][h].p = fake;
#pragma unroll
            for (int j = 0; j < TN; ++j) rb[j][h].p = fake;
            continue;
          }
          const int pxs = (kp * 4 + (ln >> 5) * 2 + h) ^ lds_swz<NP>(ln & 31);
          const piece_t* pa = ta + (wm * WTM + (ln & 31)) * NP + pxs;
          const piece_t* pb = ta + (BM + wn * WTN + (ln & 31)) * NP + pxs;
#pragma unroll
          for (int i = 0; i < TM; ++i) ra[i][h].p = pa[i * 32 * NP];
#pragma unroll
          for (int j = 0; j < TN; ++j) rb[j][h].p = pb[j * 32 * NP];
        }
      };
      load_raw(0);
#pragma unroll
      for (int kp = 0; kp < NKP; ++kp) {
        X3Frag<MM> xa[TM], xb[TN];
        if constexpr ((ABL & 8) != 0) {  // raw bits stand in for the split operands
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            __builtin_memcpy(&xa[i].hi, &ra[i][0].p, 16);
            __builtin_memcpy(&xa[i].lo, &ra[i][1].p, 16);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            __builtin_memcpy(&xb[j].hi, &rb[j][0].p, 16);
            __builtin_memcpy(&xb[j].lo, &rb[j][1].p, 16);
          }
        } else {
#pragma unroll
          for (int i = 0; i < TM; ++i) x3_split<MM>(ra[i][0], ra[i][1], xa[i]);
#pragma unroll
          for (int j = 0; j < TN; ++j) x3_split<MM>(rb[j][0], rb[j][1], xb[j]);
        }
        if (kp + 1 < NKP) load_raw(kp + 1);  // the raw registers are free again: next group's reads fly under these MFMAs
        x3_fence();
        if constexpr ((ABL & 16) != 0) {
#pragma unroll
          for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(xa[i].hi), "v"(xa[i].lo));
#pragma unroll
          for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(xb[j].hi), "v"(xb[j].lo));
        } else {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) x3_mma<MM>(xa[i], xb[j], acc[i][j]);
        }
        if constexpr (SPL) {
          if (more) issue_part(nxt, kp * IPT / NKP, (kp + 1) * IPT / NKP);
        }
      }
    } else if constexpr (HQ) {
      typedef int intx8 __attribute__((ext_vector_type(8)));
      union QFrag {
        piece_t p[2];
        intx8 v;
      };
      const int lh = ln >> 5;
      const int sa = lh ? 127 - kQA_HI : 127 - kQA_LO, sb = lh ? 127 - kQW_LO : 127 - kQW_HI;
      const int sw = lds_swz<NP>(ln & 31);
      const piece_t* pa = ta + (wm * WTM + (ln & 31)) * NP;
      const piece_t* pb = ta + (BM + wn * WTN + (ln & 31)) * NP;
      QFrag qa[TM], qb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        qa[i].p[0] = pa[i * 32 * NP + ((4 + lh) ^ sw)];
        qa[i].p[1] = pa[i * 32 * NP + ((6 + lh) ^ sw)];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        qb[j].p[0] = pb[j * 32 * NP + ((4 + lh) ^ sw)];
        qb[j].p[1] = pb[j * 32 * NP + ((6 + lh) ^ sw)];
      }
      PieceView<half> ha[2][TM], hb[2][TN];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < TM; ++i) ha[h][i].p = pa[i * 32 * NP + ((2 * h + lh) ^ sw)];
#pragma unroll
        for (int j = 0; j < TN; ++j) hb[h][j].p = pb[j * 32 * NP + ((2 * h + lh) ^ sw)];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa[i].v, qb[j].v, acc[i][j], 0, 0, 0, sa, 0, sb);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (SPL) {
        if (more) issue_part(nxt, 0, IPT / 2);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) mma_piece<half>(ha[h][i], hb[h][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (SPL) {
        if (more) issue_part(nxt, IPT / 2, IPT);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (H2) {
      // h2 operands (common.h): the K-step's eight pieces are one 32-channel group, pieces 0-3 its hi halves, 4-7 its lo
      // halves; 16-k fragment (plane pl, half h of the group) = pieces 4 pl + 2 h + {0, 1}.  Per half: hi and lo fragments of
      // both operands, three MFMAs per tile pair (small terms first); the second half's reads fly under the first half's MFMAs.
      PieceView<half> fa[2][2][TM], fb[2][2][TN];  // [half of the group][plane]
      auto load_h2 = [&](int h) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          const int pxs = (4 * pl + 2 * h + (ln >> 5)) ^ lds_swz<NP>(ln & 31);
          const piece_t* pa = ta + (wm * WTM + (ln & 31)) * NP + pxs;
          const piece_t* pb = ta + (BM + wn * WTN + (ln & 31)) * NP + pxs;
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[h][pl][i].p = pa[i * 32 * NP];
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[h][pl][j].p = pb[j * 32 * NP];
        }
      };
      load_h2(0);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h == 0) load_h2(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            mma_piece<half>(fa[h][1][i], fb[h][0][j], acc[i][j]);
            mma_piece<half>(fa[h][0][i], fb[h][1][j], acc[i][j]);
            mma_piece<half>(fa[h][0][i], fb[h][0][j], acc[i][j]);
          }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SPL) {
          if (more) issue_part(nxt, h * IPT / 2, (h + 1) * IPT / 2);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
    PieceView<T> fa[2][TM], fb[2][TN];
    auto load_frags = [&](int ks, int slot) {
      const int pxs = (ks * 2 + (ln >> 5)) ^ lds_swz<NP>(ln & 31);
      const piece_t* pa = ta + (wm * WTM + (ln & 31)) * NP + pxs;
      const piece_t* pb = ta + (BM + wn * WTN + (ln & 31)) * NP + pxs;
      if constexpr ((ABL & 2) != 0) {
        const piece_t fake = {(unsigned)ln, (unsigned)kt, (unsigned)ks, (unsigned)pxs};
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[slot][i].p = fake;
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[slot][j].p = fake;
        return;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[slot][i].p = pa[i * 32 * NP];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[slot][j].p = pb[j * 32 * NP];
    };
    load_frags(0, 0);
#pragma unroll
    for (int ks = 0; ks < NP / 2; ++ks) {
      if (ks + 1 < NP / 2) load_frags(ks + 1, (ks + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);  // keep the next sub-step's ds_reads ahead of these MFMAs
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma_piece<T>(fa[ks & 1][i], fb[ks & 1][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (SPL) {
        constexpr int NSUB = NP / 2;
        if (more) issue_part(nxt, ks * IPT / NSUB, (ks + 1) * IPT / NSUB);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    }  // native / x3
    cur = cur + 1 == NBUF ? 0 : cur + 1;
    nxt = nxt + 1 == NBUF ? 0 : nxt + 1;
  }
  __syncthreads();  // all fragment reads done before the ring is reused as epilogue staging

  // ---- epilogue ---------------------------------------------------------------------------------
  // h2: the output (raw conv output / data gradient, consumed by the BatchNorm kernels) and the residual are fp32 [M][ldc];
  // the gates are h2 activations, of which only the hi halves are read (the sign of x is the sign of its hi half)
  using OT = typename std::conditional<H2, float, T>::type;
  constexpr int OVEC = 16 / (int)sizeof(OT);
  OT* out = reinterpret_cast<OT*>(ep.out);
  const OT* res = reinterpret_cast<const OT*>(ep.res);
  const T* gate = reinterpret_cast<const T*>(ep.res_gate);
  const T* ogate = reinterpret_cast<const T*>(ep.out_gate);
  const float alpha = ep.alpha;
  float* stage = reinterpret_cast<float*>(&smem[0]);  // [64][SC] fp32 sub-block
  constexpr int SC = BN < 128 ? BN : 128;             // columns staged per round
  constexpr int CPR = SC / OVEC;                      // output pieces per staged row
  // wave rows staged per round: two (64 rows, fits every ring), or all three of the 12-wave configuration (96 rows =
  // 48 KB of its 150 KB ring: 6 rounds of two full store passes instead of 12 rounds of 1.33)
  constexpr int WR = (WM == 3 && NBUF * TILE_PIECES * 16 >= 96 * SC * 4) ? 3 : 2;
  constexpr int PASSES = (WR * 32 * CPR + NT - 1) / NT;
  float s1[TN], s2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) s1[j] = s2[j] = 0.f;
  // rounds: (tile row i of each wave) x (groups of WR wave rows) x (128-column blocks)
#pragma unroll
  for (int i = 0; i < TM; ++i)
    for (int mh = 0; mh < (WM + WR - 1) / WR; ++mh)
      for (int nh = 0; nh < BN / SC; ++nh) {
        if (wm / WR == mh) {
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int lc = wn * WTN + j * 32 + (lane & 31);  // column inside the block tile
            if (lc / SC == nh) {
              const float bias = (ep.bias && n0 + lc < g.N) ? ep.bias[n0 + lc] : 0.f;
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r] * alpha + bias;
                if (ep.relu & 1) v = fmaxf(v, 0.f);
                s1[j] += v;
                s2[j] += v * v;
                const int lr = (wm % WR) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                stage[lr * SC + (lc - nh * SC)] = v;
              }
            }
          }
        }
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
          const int id = t + ps * NT;
          const int lr = id / CPR, cpc = id % CPR;
          const int row = m0 + (WR * mh + (lr >> 5)) * WTM + i * 32 + (lr & 31);
          const int col = n0 + nh * SC + cpc * OVEC;
          if (lr < WR * 32 && WR * mh + (lr >> 5) < WM && row < g.M && col < g.N) {
            float v[OVEC];
#pragma unroll
            for (int e = 0; e < OVEC; e += 4) {
              floatx4 f = *reinterpret_cast<const floatx4*>(&stage[lr * SC + cpc * OVEC + e]);
              v[e] = f[0];
              v[e + 1] = f[1];
              v[e + 2] = f[2];
              v[e + 3] = f[3];
            }
            long orow = row;
            if (ep.om_on) {
              const int tmp = fastdiv(row, ep.om_dq), oq = row - tmp * ep.om_Q;
              const int ob = fastdiv(tmp, ep.om_dp), op = tmp - ob * ep.om_P;
              orow = ((long)ob * ep.om_H + 2 * op + ep.om_a) * ep.om_W + 2 * oq + ep.om_b;
            }
            const long idx = orow * ep.ldc + col;
            if constexpr (H2) {
              const long gidx = h2_index(orow, ep.ldc, col);  // hi halves of the gate's channels col .. col + 3
              if (res) {
                PieceView<float> rv;
                Half4View gv;
                rv.p = *reinterpret_cast<const piece_t*>(res + idx);
                if (gate) gv.p = *reinterpret_cast<const u32x2*>(gate + gidx);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  float x = rv.e[e];
                  if (gate && !((float)gv.e[e] > 0.f)) x = 0.f;
                  v[e] += x;
                }
              }
              if (ogate) {
                Half4View ov;
                ov.p = *reinterpret_cast<const u32x2*>(ogate + gidx);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (!((float)ov.e[e] > 0.f)) v[e] = 0.f;
              }
              PieceView<float> o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o.e[e] = v[e];
              *reinterpret_cast<piece_t*>(out + idx) = o.p;
            } else {
            // (gate_h2: the gates are h2 tensors, hi halves of channels col .. col + 7 = one 16-byte piece of the row's group)
            const long gix = (sizeof(T) == 2 && ep.gate_h2) ? h2_index(orow, ep.ldc, col) : idx;
            if (res) {
              PieceView<T> rv, gv;
              rv.p = *reinterpret_cast<const piece_t*>(res + idx);
              if (gate) gv.p = *reinterpret_cast<const piece_t*>(gate + gix);
#pragma unroll
              for (int e = 0; e < VEC; ++e) {
                float x = (float)rv.e[e];
                if (gate && !((float)gv.e[e] > 0.f)) x = 0.f;
                v[e] += x;
              }
            }
            PieceView<T> o;
            if (ogate) {
              PieceView<T> ov;
              ov.p = *reinterpret_cast<const piece_t*>(ogate + gix);
#pragma unroll
              for (int e = 0; e < VEC; ++e)
                if (!((float)ov.e[e] > 0.f)) v[e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) o.e[e] = (T)v[e];
            *reinterpret_cast<piece_t*>(out + idx) = o.p;
            }
          }
        }
        __syncthreads();
      }
  if (ep.stats || ep.stats_accum) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      s1[j] += __shfl_xor(s1[j], 32);
      s2[j] += __shfl_xor(s2[j], 32);
      if (lane < 32) {
        const int lc = wn * WTN + j * 32 + lane;
        red[(wm * BN + lc) * 2 + 0] = s1[j];
        red[(wm * BN + lc) * 2 + 1] = s2[j];
      }
    }
    __syncthreads();
    for (int c = t; c < BN; c += NT)
      if (n0 + c < g.N) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          a += red[(w * BN + c) * 2 + 0];
          b += red[(w * BN + c) * 2 + 1];
        }
        if (ep.stats_accum) {
          double* row = ep.stats_accum + (long)(tile_m % ep.stats_rows) * 2 * g.N;
          atomicAdd(row + n0 + c, (double)a);
          atomicAdd(row + g.N + n0 + c, (double)b);
        } else {
          ep.stats[((long)tile_m * 2 + 0) * g.N + n0 + c] = a;
          ep.stats[((long)tile_m * 2 + 1) * g.N + n0 + c] = b;
        }
      }
  }
}

// upper bound of the number of M-blocks a launch uses (sizes the BatchNorm partial buffer)
inline int igemm_grid_m(int M) { return cdiv(M, 128); }

// Tile configurations (per-shape choice in launch_igemm; MN_IGEMM_CONFIG=1|8|12 forces one -- the parity tests use it to
// run the 12-wave tile on small ragged problems):
//   1: 128x128 / 128x64 (N <= 64), 4 waves of 64x64 / 64x32, 128-byte K-steps, 2 buffers   (2 workgroups/CU)
//   8: 256x128, 4 waves of 128x64, 64-byte K-steps, 3 buffers                              (2 workgroups/CU)
//  12: 288x256, 12 waves of 96x64, 128-byte K-steps, 2 buffers                             (1 workgroup/CU)
// Measured alternatives that lost everywhere and were removed: 3-4 workgroups/CU (64-byte steps), deeper rings,
// 256x64 tiles for N = 64, 8-wave 256x128 and 256x256 tiles, 256x256 tiles of 128x128 wave tiles, 576x128 tiles.
inline int igemm_config() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MN_IGEMM_CONFIG");
    v = e ? atoi(e) : 0;
  }
  return v;
}

inline int device_cus() {  // compute units of the current device (256 on MI355X)
  static int v = 0;
  if (v == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    v = cus;
  }
  return v;
}

template <typename T, int WM, int WN, int TM, int TN, int NP, int NBUF, int MINW, bool UNI = true, bool SPL = false, int ABL = 0>
inline int launch_igemm_cfg(const GatherGeom& g, const T* A, const T* Bw, const Epilogue& ep, hipStream_t stream,
                            const T* zero_page) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const int gm = cdiv(g.M, BM), gn = cdiv(g.N, BN);
  RowDiv rd;
  rd.q = make_fastdiv(g.Q);
  rd.p = make_fastdiv(g.P);
  if constexpr (sizeof(T) == 4 && NP % 4 == 0 && !SPL && WM == 2 && WN == 2 && TN <= 2) {  // (the x3 configurations)
    if (g.mma == MMA_F16X3) {
      hipLaunchKernelGGL((igemm_kernel<T, WM, WN, TM, TN, NP, NBUF, MINW, UNI, SPL, ABL, MMA_F16X3>), dim3(gm * gn),
                         dim3(WM * WN * 64), 0, stream, g, A, Bw, ep, gn, zero_page, rd);
      return gm;
    }
    if (g.mma == MMA_BF16X3) {
      hipLaunchKernelGGL((igemm_kernel<T, WM, WN, TM, TN, NP, NBUF, MINW, UNI, SPL, ABL, MMA_BF16X3>), dim3(gm * gn),
                         dim3(WM * WN * 64), 0, stream, g, A, Bw, ep, gn, zero_page, rd);
      return gm;
    }
  }
  hipLaunchKernelGGL((igemm_kernel<T, WM, WN, TM, TN, NP, NBUF, MINW, UNI, SPL, ABL>), dim3(gm * gn), dim3(WM * WN * 64), 0,
                     stream, g, A, Bw, ep, gn, zero_page, rd);
  return gm;
}

// igemm_halo.h: 288-row tiles with the A operand staged once per 64-channel chunk (fp16 3x3 stride-1 convolutions)
inline int launch_igemm_halo(const GatherGeom& g, const half* A, const half* Bw, const Epilogue& ep, hipStream_t stream,
                             int level, bool tile288_wanted);
template <typename T>
inline int maybe_launch_igemm_halo(const GatherGeom&, const T*, const T*, const Epilogue&, hipStream_t, bool) {
  return -1;
}
template <>
inline int maybe_launch_igemm_halo<half>(const GatherGeom& g, const half* A, const half* Bw, const Epilogue& ep,
                                         hipStream_t stream, bool tile288_wanted) {
  // measured on MI355X (round 2, same-box A/B of the whole step): level 0 17.34 ms, 1 (256-column shape: layer3) 17.03,
  // 2 (+ 128-column shape: layers 2 and 4) 16.79; per launch layer2 114 -> 104, layer3 93 -> 81, layer4 112 -> 87 us.
  // MN_IGEMM_HALO=0|1 (parity tests: the generic kernel on the same shapes)
  static const int level = getenv("MN_IGEMM_HALO") ? atoi(getenv("MN_IGEMM_HALO")) : 2;
  return level > 0 ? launch_igemm_halo(g, A, Bw, ep, stream, level, tile288_wanted) : -1;
}

// returns the number of M-blocks used (= rows of the stats partial buffer that were written)
template <typename T>
inline int launch_igemm(const GatherGeom& g_in, const T* A, const T* Bw, const Epilogue& ep, hipStream_t stream,
                        const T* zero_page) {
  constexpr int VEC = ElemTraits<T>::VEC;
  GatherGeom g = g_in;
  // taps-fastest K order (all R*S taps of one 128-byte channel chunk before the next chunk): layer3 data gradient
  // 98 -> 92 us, with residual 117 -> 107, layer2 120 -> 117
  g.tap_inner = (g.R * g.S > 1 || g.bt_on) ? 1 : 0;
  const bool wide_k = (g.C / VEC) % 8 == 0;  // 128-byte K-steps need taps that are a multiple of them
  int cfg = igemm_config();
  if constexpr (sizeof(T) == 4) {
    // x3 modes: 128-row tiles of 4 waves, two workgroups per CU (the split operands and the raw pieces need the 256
    // registers that occupancy leaves a wave)
    if (g.mma != MMA_NATIVE) {
      // ... except where 128x128 tiles leave a few tiles over one round of the 512 resident workgroups (layer4 at 192
      // images: 528 tiles): 256x128 tiles of 128x64 wave tiles put every tile in one round (367 -> 314 us forward,
      // 364 -> 308 us data gradient).  Measured and not kept (round 3, tools/conv_bench.py fp32x3): 64-byte K-steps with a
      // 4-deep ring (layer3 313 -> 354 us), the 256x128 tile everywhere (layer3 313 -> 369 us).
      const long x3_tiles128 = (long)cdiv(g.M, 128) * cdiv(g.N, 128);
      if ((g.C / VEC) % 4 == 0 && g.N % 128 == 0 && x3_tiles128 > 512 && x3_tiles128 <= 640)
        return launch_igemm_cfg<T, 2, 2, 4, 2, 4, 3, 2>(g, A, Bw, ep, stream, zero_page);
      if ((g.C / VEC) % 4 != 0) {
        if (g.N <= 64) return launch_igemm_cfg<T, 2, 2, 2, 1, 4, 4, 2, false>(g, A, Bw, ep, stream, zero_page);
        return launch_igemm_cfg<T, 2, 2, 2, 2, 4, 4, 2, false>(g, A, Bw, ep, stream, zero_page);
      }
      if (g.N <= 64) {
        if (wide_k) return launch_igemm_cfg<T, 2, 2, 2, 1, 8, 2, 2>(g, A, Bw, ep, stream, zero_page);
        return launch_igemm_cfg<T, 2, 2, 2, 1, 4, 4, 2>(g, A, Bw, ep, stream, zero_page);
      }
#ifdef MN_ABLATION_BUILD
      if (wide_k) {
        static const int abl = getenv("MN_ABLATE") ? atoi(getenv("MN_ABLATE")) : 0;
        switch (abl) {
          case 1: return launch_igemm_cfg<T, 2, 2, 2, 2, 8, 2, 2, true, false, 1>(g, A, Bw, ep, stream, zero_page);
          case 2: return launch_igemm_cfg<T, 2, 2, 2, 2, 8, 2, 2, true, false, 2>(g, A, Bw, ep, stream, zero_page);
          case 3: return launch_igemm_cfg<T, 2, 2, 2, 2, 8, 2, 2, true, false, 3>(g, A, Bw, ep, stream, zero_page);
          case 4: return launch_igemm_cfg<T, 2, 2, 2, 2, 8, 2, 2, true, false, 4>(g, A, Bw, ep, stream, zero_page);
          case 7: return launch_igemm_cfg<T, 2, 2, 2, 2, 8, 2, 2, true, false, 7>(g, A, Bw, ep, stream, zero_page);
          case 8: return launch_igemm_cfg<T, 2, 2, 2, 2, 8, 2, 2, true, false, 8>(g, A, Bw, ep, stream, zero_page);
          case 16: return launch_igemm_cfg<T, 2, 2, 2, 2, 8, 2, 2, true, false, 16>(g, A, Bw, ep, stream, zero_page);
          case 24: return launch_igemm_cfg<T, 2, 2, 2, 2, 8, 2, 2, true, false, 24>(g, A, Bw, ep, stream, zero_page);
          case 31: return launch_igemm_cfg<T, 2, 2, 2, 2, 8, 2, 2, true, false, 31>(g, A, Bw, ep, stream, zero_page);
          default: break;
        }
      }
#endif
      if (wide_k) return launch_igemm_cfg<T, 2, 2, 2, 2, 8, 2, 2>(g, A, Bw, ep, stream, zero_page);
      return launch_igemm_cfg<T, 2, 2, 2, 2, 4, 4, 2>(g, A, Bw, ep, stream, zero_page);
    }
  }
  // channel counts that are not a multiple of the K-step (stem pixel pairs, odd test shapes): per-lane tap walk
  if ((g.C / VEC) % 4 != 0) {
    if (g.N <= 64) return launch_igemm_cfg<T, 2, 2, 2, 1, 4, 4, 2, false>(g, A, Bw, ep, stream, zero_page);
    return launch_igemm_cfg<T, 2, 2, 2, 2, 4, 4, 2, false>(g, A, Bw, ep, stream, zero_page);
  }
  // per-shape default (tools/conv_bench.py on MI355X, B = 192): 128x128 tiles, two workgroups per CU (512 slots);
  // when that leaves a few tiles over one full round (layer4: 528 tiles), 256x128 tiles with 128-row wave tiles
  // put every tile in a single round instead (103 vs 121 us)
  if (cfg == 0) {
    const long tiles128 = (long)cdiv(g.M, 128) * cdiv(g.N, 128);
    cfg = (g.N >= 128 && g.N % 128 == 0 && tiles128 > 512 && tiles128 <= 640) ? 8 : 1;
    // 288x256 tiles when they cover the problem in ONE round of one workgroup per CU (layer3 at B = 192: 235 tiles
    // instead of 1056 128x128 tiles = 2.06 rounds of 512; 92 vs 110-118 us)
    const long tiles288 = (long)cdiv(g.M, 288) * cdiv(g.N, 256);
    if (wide_k && g.N % 256 == 0 && tiles128 > 512 && tiles288 > 192 && tiles288 <= device_cus() && g.R * g.S <= 10) cfg = 12;
  }
  {
    const int gm_halo = maybe_launch_igemm_halo<T>(g, A, Bw, ep, stream, cfg == 12 && wide_k && g.N % 256 == 0);
    if (gm_halo >= 0) return gm_halo;
  }
  if (g.N <= 64) {
    // (measured and removed for the 64-channel layers: 3-deep ring, 64-byte steps with a 4-deep ring, 256x64 tiles of
    // 64x64 wave tiles, 2-wave workgroups of 64x64 wave tiles: 145 us -> 157..183 us on layer1)
    if (wide_k) return launch_igemm_cfg<T, 2, 2, 2, 1, 8, 2, 2>(g, A, Bw, ep, stream, zero_page);
    return launch_igemm_cfg<T, 2, 2, 2, 1, 4, 4, 2>(g, A, Bw, ep, stream, zero_page);
  }
  if (cfg == 8 && g.N % 64 == 0)  // 256x128, 4 waves of 128x64, 64-byte K-steps, 3 buffers
    return launch_igemm_cfg<T, 2, 2, 4, 2, 4, 3, 2>(g, A, Bw, ep, stream, zero_page);
  // 288x256, 12 waves of 96x64, 128-byte K-steps, 2 buffers, one workgroup per CU: M = B*P*Q of the 256x341 input at
  // B = 192 is 132 * 2^k, and 288-row tiles put layer3 (67584 rows, N = 256) on 235 of the 256 CUs in ONE round.  The
  // next tile's DMA instructions are issued behind the MFMAs of each sub-step (SPL: data gradient 95.8 -> 90.1 us, with
  // residual 111.7 -> 103.6; the same change costs the 128x128 configuration 5 %, so it is not used there).
  if (cfg == 12 && wide_k && g.N % 256 == 0 && g.R * g.S <= 10) {  // (packed tap masks: 10 bits per A pass)
#ifdef MN_ABLATION_BUILD
    static const int abl = getenv("MN_ABLATE") ? atoi(getenv("MN_ABLATE")) : 0;
    if (abl == 1) return launch_igemm_cfg<T, 3, 4, 3, 2, 8, 2, 3, true, true, 1>(g, A, Bw, ep, stream, zero_page);
    if (abl == 2) return launch_igemm_cfg<T, 3, 4, 3, 2, 8, 2, 3, true, true, 2>(g, A, Bw, ep, stream, zero_page);
    if (abl == 3) return launch_igemm_cfg<T, 3, 4, 3, 2, 8, 2, 3, true, true, 3>(g, A, Bw, ep, stream, zero_page);
    if (abl == 4) return launch_igemm_cfg<T, 3, 4, 3, 2, 8, 2, 3, true, true, 4>(g, A, Bw, ep, stream, zero_page);
    if (abl == 7) return launch_igemm_cfg<T, 3, 4, 3, 2, 8, 2, 3, true, true, 7>(g, A, Bw, ep, stream, zero_page);
#endif
    return launch_igemm_cfg<T, 3, 4, 3, 2, 8, 2, 3, true, true>(g, A, Bw, ep, stream, zero_page);
  }
  // (measured and removed: the same tile with 64-byte K-steps and FOUR buffers, i.e. three tiles in flight -- 99 vs 93 us
  // on layer3, so the ring depth is not what limits it; 288x128 tiles of 6 waves, two workgroups per CU -- 131 us)
  if (wide_k) return launch_igemm_cfg<T, 2, 2, 2, 2, 8, 2, 2>(g, A, Bw, ep, stream, zero_page);
  return launch_igemm_cfg<T, 2, 2, 2, 2, 4, 4, 2>(g, A, Bw, ep, stream, zero_page);
}

// h2 operands (common.h MMA_H2): `g_in` describes the convolution in REAL channels; A is an h2 tensor [B][Hi][Wi][C], Bw an
// h2 weight matrix [N][taps][C] (row pitch ldb real elements), out / res fp32, gates h2.  The kernels see fp16 tensors with 2C
// channels.  128x128 / 128x64 tiles of four waves, two workgroups per CU (3x3 stride-1 shapes go to igemm_halo.h).
inline int launch_igemm_halo_h2(const GatherGeom& g2, const half* A, const half* Bw, const Epilogue& ep, hipStream_t stream);
// q: A and Bw are h2q tensors (common.h), the cross terms run on the scaled fp8 MFMA (MMA_H2Q)
// (halo_h2.h, round 6: layer1's 64 -> 64 h2 convolutions with the weights in registers)
inline bool conv_halo_h2_applies(const GatherGeom& g, const Epilogue& ep);
inline int conv_halo_h2_grid(const GatherGeom& g);
inline void launch_conv_halo_h2(const GatherGeom& g, const half* A, const half* Bw, const Epilogue& ep, hipStream_t stream);
inline bool use_conv_halo_h2() {  // MN_HALO_H2=0: the chunk-resident 64-column shape of igemm_halo.h (A/B measurements)
  static const bool on = !(getenv("MN_HALO_H2") && atoi(getenv("MN_HALO_H2")) == 0);
  return on;
}
inline int launch_igemm_h2(const GatherGeom& g_in, const half* A, const half* Bw, const Epilogue& ep, hipStream_t stream,
                           const half* zero_page, bool q = false) {
  if (!q && use_conv_halo_h2() && conv_halo_h2_applies(g_in, ep)) {
    launch_conv_halo_h2(g_in, A, Bw, ep, stream);
    return conv_halo_h2_grid(g_in);
  }
  GatherGeom g = g_in;
  g.C = 2 * g_in.C;
  g.K = 2 * g_in.K;
  g.ldb = 2 * g_in.ldb;
  g.mma = q ? MMA_H2Q : MMA_H2;
  g.tap_inner = (g.R * g.S > 1 || g.bt_on) ? 1 : 0;
  RowDiv rd;
  rd.q = make_fastdiv(g.Q);
  rd.p = make_fastdiv(g.P);
  static const int halo_level = getenv("MN_IGEMM_HALO") ? atoi(getenv("MN_IGEMM_HALO")) : 2;
  if (halo_level > 0) {
    const int gm_halo = launch_igemm_halo_h2(g, A, Bw, ep, stream);
    if (gm_halo >= 0) return gm_halo;
  }
#define MN_H2_CFG(WM_, WN_, TM_, TN_, NP_, NBUF_, MINW_)                                                                   \
  {                                                                                                                        \
    constexpr int BM = WM_ * TM_ * 32, BN = WN_ * TN_ * 32;                                                                \
    const int gm = cdiv(g.M, BM), gn = cdiv(g.N, BN);                                                                      \
    if (q)                                                                                                                 \
      hipLaunchKernelGGL((igemm_kernel<half, WM_, WN_, TM_, TN_, NP_, NBUF_, MINW_, true, false, 0, MMA_H2Q>), dim3(gm * gn), \
                         dim3(WM_ * WN_ * 64), 0, stream, g, A, Bw, ep, gn, zero_page, rd);                                \
    else                                                                                                                   \
      hipLaunchKernelGGL((igemm_kernel<half, WM_, WN_, TM_, TN_, NP_, NBUF_, MINW_, true, false, 0, MMA_H2>), dim3(gm * gn), \
                         dim3(WM_ * WN_ * 64), 0, stream, g, A, Bw, ep, gn, zero_page, rd);                                \
    return gm;                                                                                                             \
  }
  if (g.N <= 64) MN_H2_CFG(2, 2, 2, 1, 8, 2, 2)
  MN_H2_CFG(2, 2, 2, 2, 8, 2, 2)
#undef MN_H2_CFG
}

}  // namespace mn

#include "igemm_halo.h"
#include "halo_h2.h"
