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
// Tiling: 256 threads = 4 waves (2x2); block tile 128 x BN; each wave owns a 64 x (BN/2) sub-tile
// made of 32x32 MFMA tiles (v_mfma_f32_32x32x16_f16 for half, v_mfma_f32_32x32x2_f32 for float;
// fp32 accumulate).  K advances NP 16-byte pieces per row per step (NP = 8: 128 B, one full cache
// line per gathered pixel; NP = 4 for the stem whose K is not a multiple of 128 B).  Tiles are
// staged global -> registers -> LDS (double buffered, one barrier per K-step) with the next tile's
// global loads in flight during the MFMAs.  The 16-byte slot index inside an LDS row is
// XOR-swizzled with the row so the 16-lane groups of ds_read_b128 hit 16 distinct slots
// (MI355X_MICROARCH.md "LDS").
//
// Gather cost: each thread precomputes, once, the pixel offset of its rows and a bit mask of the
// taps that fall inside the image; a K-step then costs one shift/and/add per row.
// Workgroup ids are remapped so each XCD (private L2) walks a contiguous range of output tiles with
// the N tiles of one M tile adjacent (they share the gathered A rows).
//
// Epilogue: alpha, bias, ReLU and the BatchNorm column partial sums in registers; the tile is then
// staged through LDS (fp32) so that global stores -- and the residual / ReLU-gate loads of the
// data-gradient path -- are full 16-byte pieces along channels.
#pragma once
#include <stdlib.h>

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
};

struct Epilogue {
  void* out;             // [M][ldc], element type T
  int ldc;
  float* stats;          // [grid_m][2][N] column partial sums (sum, sum of squares) or null
  const float* bias;     // [N] or null
  int relu;              // max(0, .) after bias
  const void* res;       // residual [M][ldc] of type T or null
  const void* res_gate;  // if non-null the residual passes only where res_gate[m][n] > 0
  float alpha;           // scale applied to the accumulator
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

// Shared epilogue: acc -> (alpha, bias, ReLU, BN partial sums) -> fp32 staging in LDS -> 16-byte stores
// with optional residual / ReLU-gate.  `stage` must hold 64*BN floats, `red` 2*BN*2 floats; all waves
// must be past their last read of the LDS tiles.
template <typename T, int BN>
__device__ __forceinline__ void igemm_epilogue(const GatherGeom& g, const Epilogue& ep, floatx16 (&acc)[2][BN / 64],
                                               float* stage, float (*red)[BN][2], int m0, int n0, int tile_m) {
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int WN = 2, WTM = 64, WTN = BN / WN, TM = 2, TN = WTN / 32;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  T* out = reinterpret_cast<T*>(ep.out);
  const T* res = reinterpret_cast<const T*>(ep.res);
  const T* gate = reinterpret_cast<const T*>(ep.res_gate);
  float s1[TN], s2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) s1[j] = s2[j] = 0.f;
  constexpr int CPR = BN / VEC;         // output pieces per tile row
  constexpr int PASSES = 64 * CPR / 256;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int lc = wn * WTN + j * 32 + (lane & 31);
      const float bias = (ep.bias && n0 + lc < g.N) ? ep.bias[n0 + lc] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[i][j][r] * ep.alpha + bias;
        if (ep.relu) v = fmaxf(v, 0.f);
        s1[j] += v;
        s2[j] += v * v;
        const int lr = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        stage[lr * BN + lc] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int id = t + ps * 256;
      const int lr = id / CPR, cpi = id % CPR;
      const int row = m0 + (lr >> 5) * WTM + i * 32 + (lr & 31);
      const int col = n0 + cpi * VEC;
      if (row < g.M && col < g.N) {
        float v[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {
          floatx4 f = *reinterpret_cast<const floatx4*>(&stage[lr * BN + cpi * VEC + e]);
          v[e] = f[0];
          v[e + 1] = f[1];
          v[e + 2] = f[2];
          v[e + 3] = f[3];
        }
        const long idx = (long)row * ep.ldc + col;
        if (res) {
          PieceView<T> rv, gv;
          rv.p = *reinterpret_cast<const piece_t*>(res + idx);
          if (gate) gv.p = *reinterpret_cast<const piece_t*>(gate + idx);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            float x = (float)rv.e[e];
            if (gate && !((float)gv.e[e] > 0.f)) x = 0.f;
            v[e] += x;
          }
        }
        PieceView<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.e[e] = (T)v[e];
        *reinterpret_cast<piece_t*>(out + idx) = o.p;
      }
    }
    __syncthreads();
  }
  if (ep.stats) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      s1[j] += __shfl_xor(s1[j], 32);
      s2[j] += __shfl_xor(s2[j], 32);
      if (lane < 32) {
        red[wm][wn * WTN + j * 32 + lane][0] = s1[j];
        red[wm][wn * WTN + j * 32 + lane][1] = s2[j];
      }
    }
    __syncthreads();
    if (t < BN && n0 + t < g.N) {
      ep.stats[((long)tile_m * 2 + 0) * g.N + n0 + t] = red[0][t][0] + red[1][t][0];
      ep.stats[((long)tile_m * 2 + 1) * g.N + n0 + t] = red[0][t][1] + red[1][t][1];
    }
  }
}

template <typename T, int BN, int NP>
static __global__ void __launch_bounds__(256) igemm_kernel(GatherGeom g, const T* __restrict__ A, const T* __restrict__ Bw,
                                                            Epilogue ep, int grid_n) {
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int BM = 128;
  constexpr int WN = 2;
  constexpr int WTM = 64, WTN = BN / WN;
  constexpr int TM = 2, TN = WTN / 32;
  constexpr int RPP = 256 / NP;            // rows covered by one pass of the 256 loader threads
  constexpr int APT = BM / RPP, BPT = BN / RPP;
  static_assert(APT >= 1 && BPT >= 1 && TN >= 1, "tile too small");
  constexpr int LDS_PIECES = (BM + BN) * NP;
  static_assert(2 * LDS_PIECES * 16 >= 64 * BN * 4, "epilogue staging does not fit");

  __shared__ piece_t lds[2][LDS_PIECES];
  __shared__ float red[2][BN][2];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = tile / grid_n, tile_n = tile - tile_m * grid_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int pc = t % NP, lrow = t / NP;

  // ---- per-row gather state: pixel offset of tap (0,0) and validity mask over taps ---------
  const int dsh = g.div == 2 ? 1 : 0;
  const int ntaps = g.R * g.S;
  int a_pix[APT];
  unsigned a_mask[APT];
#pragma unroll
  for (int i = 0; i < APT; ++i) {
    const int m = m0 + lrow + i * RPP;
    a_pix[i] = 0;
    a_mask[i] = 0u;
    if (m < g.M) {
      const int q = m % g.Q, tmp = m / g.Q;
      const int p = tmp % g.P, b = tmp / g.P;
      const int h0 = p * g.mul_p + g.off_h, w0 = q * g.mul_q + g.off_w;
      a_pix[i] = (b * g.Hi + (h0 >> dsh)) * g.Wi + (w0 >> dsh);
      int tap = 0;
      for (int r = 0; r < g.R; ++r)
        for (int s = 0; s < g.S; ++s, ++tap) {
          int hn = h0 + g.rsign * r, wn_ = w0 + g.ssign * s;
          bool ok = true;
          if (dsh) {
            ok = ((hn | wn_) & 1) == 0;
            hn >>= 1;
            wn_ >>= 1;
          }
          ok = ok && (unsigned)hn < (unsigned)g.Hi && (unsigned)wn_ < (unsigned)g.Wi;
          a_mask[i] |= (ok ? 1u : 0u) << tap;
        }
    }
  }
  (void)ntaps;
  long b_off[BPT];
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    const int n = n0 + lrow + i * RPP;
    b_off[i] = n < g.N ? (long)n * g.K + pc * VEC : -1;
  }
  const int CP = g.C / VEC;  // pieces per tap
  int cp = pc, tr = 0, ts = 0;
  while (cp >= CP) {
    cp -= CP;
    if (++ts == g.S) {
      ts = 0;
      ++tr;
    }
  }

  piece_t ra[APT], rb[BPT];
  auto load_tile = [&](int kt) {
    const int tap = tr * g.S + ts;
    // tap offset in pixels; with div = 2 (rsign = ssign = -1) valid taps sit at -(r/2), -(s/2)
    const int toff = (g.rsign * (tr >> dsh) * g.Wi + g.ssign * (ts >> dsh)) * g.C + cp * VEC;
#pragma unroll
    for (int i = 0; i < APT; ++i) {
      const bool ok = (a_mask[i] >> tap) & 1u;
      const int off = a_pix[i] * g.C + toff;
      ra[i] = ok ? *reinterpret_cast<const piece_t*>(A + off) : zero_piece();
    }
#pragma unroll
    for (int i = 0; i < BPT; ++i)
      rb[i] = b_off[i] >= 0 ? *reinterpret_cast<const piece_t*>(Bw + b_off[i] + (long)kt * (NP * VEC)) : zero_piece();
    cp += NP;
    while (cp >= CP) {
      cp -= CP;
      if (++ts == g.S) {
        ts = 0;
        ++tr;
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < APT; ++i) {
      const int row = lrow + i * RPP;
      lds[buf][row * NP + (pc ^ lds_swz<NP>(row))] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const int row = lrow + i * RPP;
      lds[buf][(BM + row) * NP + (pc ^ lds_swz<NP>(row))] = rb[i];
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int KT = g.K / (NP * VEC);
  load_tile(0);
  store_tile(0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < KT; ++kt) {
    const bool more = kt + 1 < KT;
    if (more) load_tile(kt + 1);
#pragma unroll
    for (int ks = 0; ks < NP / 2; ++ks) {
      PieceView<T> fa[TM], fb[TN];
      const int piece = ks * 2 + (lane >> 5);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * WTM + i * 32 + (lane & 31);
        fa[i].p = lds[cur][row * NP + (piece ^ lds_swz<NP>(row))];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * WTN + j * 32 + (lane & 31);
        fb[j].p = lds[cur][(BM + row) * NP + (piece ^ lds_swz<NP>(row))];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma_piece<T>(fa[i], fb[j], acc[i][j]);
    }
    if (more) store_tile(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  igemm_epilogue<T, BN>(g, ep, acc, reinterpret_cast<float*>(&lds[0][0]), red, m0, n0, tile_m);
}

// ---- LDS-DMA pipelined variant ----------------------------------------------------------------
// Same tiling and epilogue, but tiles go global -> LDS directly (global_load_lds_dwordx4: no VGPR
// staging, no ds_write pass) into an NBUF-deep ring with NBUF-1 tiles in flight, one raw s_barrier
// per K-step and counted s_waitcnt vmcnt (cdna_hip_programming.md section 5, "glds").  The DMA writes
// LDS linearly (wave-uniform base + lane*16), so the XOR swizzle is applied to the SOURCE piece a lane
// fetches and, as before, to the slot a fragment read addresses (same involution on both sides).
// Taps that fall outside the image fetch from `zero_page` (>= 16 zero bytes in HBM).
typedef const void __attribute__((address_space(1)))* gas_ptr_t;
typedef void __attribute__((address_space(3)))* las_ptr_t;
__device__ __forceinline__ void dma16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gas_ptr_t)src, (las_ptr_t)lds_wave_base, 16, 0, 0);
}
// s_waitcnt simm16 (gfx9): vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]; only vmcnt waits
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

template <typename T, int BN, int NP, int NBUF>
static __global__ void __launch_bounds__(256) igemm_dma_kernel(GatherGeom g, const T* __restrict__ A,
                                                                const T* __restrict__ Bw, Epilogue ep, int grid_n,
                                                                const T* __restrict__ zero_page) {
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int BM = 128;
  constexpr int WN = 2;
  constexpr int WTM = 64, WTN = BN / WN;
  constexpr int TM = 2, TN = WTN / 32;
  constexpr int RPP = 256 / NP;
  constexpr int APT = BM / RPP, BPT = BN / RPP;
  constexpr int IPT = APT + BPT;  // DMA instructions per wave per tile
  constexpr int TILE_PIECES = (BM + BN) * NP;
  constexpr int D = NBUF - 1;     // tiles in flight
  static_assert(D >= 1 && D <= 3, "ring depth");
  static_assert(NBUF * TILE_PIECES * 16 >= 64 * BN * 4, "epilogue staging does not fit");
  // ONE LDS object (a second one makes hipcc drain vmcnt before every fragment read):
  // [ring of tiles][BatchNorm partial reduction: 2*BN*2 floats]
  __shared__ piece_t smem[NBUF * TILE_PIECES + BN];
  float(*red)[BN][2] = reinterpret_cast<float(*)[BN][2]>(&smem[NBUF * TILE_PIECES]);

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = tile / grid_n, tile_n = tile - tile_m * grid_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int pc = t % NP, lrow = t / NP;
  const int dsh = g.div == 2 ? 1 : 0;

  int a_pix[APT];
  unsigned a_mask[APT];
#pragma unroll
  for (int i = 0; i < APT; ++i) {
    const int m = m0 + lrow + i * RPP;
    a_pix[i] = 0;
    a_mask[i] = 0u;
    if (m < g.M) {
      const int q = m % g.Q, tmp = m / g.Q;
      const int p = tmp % g.P, b = tmp / g.P;
      const int h0 = p * g.mul_p + g.off_h, w0 = q * g.mul_q + g.off_w;
      a_pix[i] = ((b * g.Hi + (h0 >> dsh)) * g.Wi + (w0 >> dsh)) * g.C;
      int tap = 0;
      for (int r = 0; r < g.R; ++r)
        for (int s = 0; s < g.S; ++s, ++tap) {
          int hn = h0 + g.rsign * r, wn_ = w0 + g.ssign * s;
          bool ok = true;
          if (dsh) {
            ok = ((hn | wn_) & 1) == 0;
            hn >>= 1;
            wn_ >>= 1;
          }
          ok = ok && (unsigned)hn < (unsigned)g.Hi && (unsigned)wn_ < (unsigned)g.Wi;
          a_mask[i] |= (ok ? 1u : 0u) << tap;
        }
    }
  }
  long b_row[BPT];
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    const int n = n0 + lrow + i * RPP;
    b_row[i] = n < g.N ? (long)n * g.K : -1;
  }
  const int CP = g.C / VEC;  // pieces per tap
  // Source piece of this lane inside a K-step: the XOR swizzle is applied on the source side.  Rows of
  // one lane differ by multiples of RPP (>= 32), which the swizzle ignores, so one value serves all rows.
  const int src_piece = pc ^ lds_swz<NP>(lrow);
  // running decomposition of piece index q = kt*NP + src_piece into (tap = (tr, ts), channel piece cpi)
  int cpi = src_piece, tr = 0, ts = 0, tap = 0;
  while (cpi >= CP) {
    cpi -= CP;
    ++tap;
    if (++ts == g.S) {
      ts = 0;
      ++tr;
    }
  }
  long b_src = src_piece * VEC;

  auto issue_tile = [&](int buf) {
    piece_t* base = &smem[buf * TILE_PIECES + wave * 64];  // this wave's 64 pieces of pass 0
    const int toff = (g.rsign * (tr >> dsh) * g.Wi + g.ssign * (ts >> dsh)) * g.C + cpi * VEC;
#pragma unroll
    for (int i = 0; i < APT; ++i) {
      const bool ok = (a_mask[i] >> tap) & 1u;
      const T* src = ok ? A + (a_pix[i] + toff) : zero_page;
      dma16(src, base + i * (RPP * NP));
    }
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const T* src = b_row[i] >= 0 ? Bw + b_row[i] + b_src : zero_page;
      dma16(src, base + (BM + i * RPP) * NP);
    }
    // advance to the next K-step
    b_src += NP * VEC;
    cpi += NP;
    while (cpi >= CP) {
      cpi -= CP;
      ++tap;
      if (++ts == g.S) {
        ts = 0;
        ++tr;
      }
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int KT = g.K / (NP * VEC);
#pragma unroll
  for (int j = 0; j < D; ++j)
    if (j < KT) issue_tile(j);
  int cur = 0, nxt = D % NBUF;
  for (int kt = 0; kt < KT; ++kt) {
    // tiles issued so far: min(KT, kt + D); tile kt must have landed
    const int ahead = min(KT, kt + D) - (kt + 1);
    if (D >= 3 && ahead >= 2)
      wait_vmcnt<2 * IPT>();
    else if (D >= 2 && ahead >= 1)
      wait_vmcnt<IPT>();
    else
      wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // tile kt visible to all waves; everyone is done reading buffer `nxt`
    if (kt + D < KT) issue_tile(nxt);  // tiles are issued strictly in K order
    const piece_t* ta = &smem[cur * TILE_PIECES];
#pragma unroll
    for (int ks = 0; ks < NP / 2; ++ks) {
      PieceView<T> fa[TM], fb[TN];
      const int piece = ks * 2 + (lane >> 5);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * WTM + i * 32 + (lane & 31);
        fa[i].p = ta[row * NP + (piece ^ lds_swz<NP>(row))];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * WTN + j * 32 + (lane & 31);
        fb[j].p = ta[(BM + row) * NP + (piece ^ lds_swz<NP>(row))];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma_piece<T>(fa[i], fb[j], acc[i][j]);
    }
    cur = cur + 1 == NBUF ? 0 : cur + 1;
    nxt = nxt + 1 == NBUF ? 0 : nxt + 1;
  }
  __syncthreads();  // all fragment reads done before the ring is reused as epilogue staging
  igemm_epilogue<T, BN>(g, ep, acc, reinterpret_cast<float*>(&smem[0]), red, m0, n0, tile_m);
}

// number of M-blocks the launcher will use (needed to size the stats partial buffer)
inline int igemm_grid_m(int M) { return cdiv(M, 128); }

// Variant selection.  MN_IGEMM_VARIANT (tuning knob, read once): 0 register-staged double buffer;
// 1 DMA 128-B steps x2 buffers; 2 DMA 128-B x3; 3 DMA 64-B x4; 4 DMA 64-B x3.  DMA variants need a zero page.
inline int igemm_variant() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MN_IGEMM_VARIANT");
    v = e ? atoi(e) : 1;  // measured best on MI355X (tools/conv_bench.py): DMA, 128-byte steps, 2 buffers
  }
  return v;
}

template <typename T, int BN>
inline void launch_igemm_bn(const GatherGeom& g, const T* A, const T* Bw, const Epilogue& ep, hipStream_t stream,
                            const T* zero_page, int gm, int gn) {
  constexpr int VEC = ElemTraits<T>::VEC;
  dim3 block(256), grid(gm * gn);
  const bool wide_k = (g.K % (8 * VEC)) == 0;
  const int cp = g.C / VEC;
  const bool dma_ok = zero_page != nullptr && (cp & (cp - 1)) == 0;
  int v = dma_ok ? igemm_variant() : 0;
  if (!wide_k && (v == 1 || v == 2)) v = 3;
  switch (v) {
    case 1:
      hipLaunchKernelGGL((igemm_dma_kernel<T, BN, 8, 2>), grid, block, 0, stream, g, A, Bw, ep, gn, zero_page);
      break;
    case 2:
      hipLaunchKernelGGL((igemm_dma_kernel<T, BN, 8, 3>), grid, block, 0, stream, g, A, Bw, ep, gn, zero_page);
      break;
    case 3:
      hipLaunchKernelGGL((igemm_dma_kernel<T, BN, 4, 4>), grid, block, 0, stream, g, A, Bw, ep, gn, zero_page);
      break;
    case 4:
      hipLaunchKernelGGL((igemm_dma_kernel<T, BN, 4, 3>), grid, block, 0, stream, g, A, Bw, ep, gn, zero_page);
      break;
    default:
      if (wide_k)
        hipLaunchKernelGGL((igemm_kernel<T, BN, 8>), grid, block, 0, stream, g, A, Bw, ep, gn);
      else
        hipLaunchKernelGGL((igemm_kernel<T, BN, 4>), grid, block, 0, stream, g, A, Bw, ep, gn);
  }
}

template <typename T>
inline void launch_igemm(const GatherGeom& g, const T* A, const T* Bw, const Epilogue& ep, hipStream_t stream,
                         const T* zero_page = nullptr) {
  const int gm = cdiv(g.M, 128);
  if (g.N <= 64)
    launch_igemm_bn<T, 64>(g, A, Bw, ep, stream, zero_page, gm, cdiv(g.N, 64));
  else
    launch_igemm_bn<T, 128>(g, A, Bw, ep, stream, zero_page, gm, cdiv(g.N, 128));
}

}  // namespace mn
