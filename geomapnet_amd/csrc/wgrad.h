// Weight-gradient GEMM on the gfx950 matrix cores.
//
//   dW[n][k] += alpha * sum_m dY[m][n] * X_gather[m][k]     n = cout, k = (r, s, c), m = (b, p, q)
//
// Both operands are NHWC, i.e. the reduction index m is the SLOW memory index and channels are
// contiguous: a "TN" GEMM from the matrix core's point of view.  Tiles are staged as loaded
// ([m][channel] rows, 16-byte pieces along channels, coalesced) and each lane gathers its
// MFMA fragment (VEC consecutive m for one channel) from LDS with element reads.  (Round-1
// structure; the LDS gather is the known limiter -- see DESIGN.md section 5.)
//
// The reduction over m = B*P*Q rows is split across gridDim.z; partial tiles are combined
// with hardware fp32 atomics into a zero-initialised gradient buffer.
// `colmap` (optional) remaps a GEMM column k to a dense destination column (or -1 = drop):
// used by the stem conv whose compute layout pads 7x7x3 taps to 7x4x8.
#pragma once
#include "common.h"
#include "igemm.h"

namespace mn {

struct WgradArgs {
  GatherGeom g;       // gather of X (same geometry as the forward conv); g.N = Cout, g.K = R*S*C
  const void* dY;     // [M][ldy]
  int ldy;
  const void* X;      // NHWC input of the forward conv
  float* dW;          // [Cout][ldw] fp32, accumulated atomically
  int ldw;
  const int* colmap;  // [K] or null
  float alpha;
  int rows_per_split; // multiple of 32
  // optional scratch for the partial tiles of wgrad_fused.h (one slab per pixel range instead of fp32 atomics into dW);
  // launches that share it must be ordered on one stream
  float* ws = nullptr;
  long ws_floats = 0;
  // det (MN_DETERMINISTIC): no two workgroups may add into the same float.  The split-K kernels below then write split z
  // into its own slice dW + z * split_stride of the (zeroed) workspace and wgrad_split_reduce_kernel adds the slices into
  // the gradient in split order; the fused kernel reduces its partial tiles with one group.
  bool det = false;
  long split_stride = 0;
  // (round 6, the fp16x2m mode) X is an h2 tensor (common.h: per row and 32-channel group 64 bytes of fp16 hi halves, then 64
  // bytes of lo halves) of which only the HI halves are read -- fp16(x) exactly, what the plain fp16 copy beside the h2 tensor
  // used to hold: rows are 4 C bytes apart and channel c lies at half (c >> 5) * 64 + (c & 31) of its row.  fp16 kernels only.
  bool x_h2 = false;
};
// half index of channel c0 (a multiple of 8: one 16-byte piece never straddles a 32-channel group) inside a row of X, and the row
// pitch in halves, for plain fp16 rows / the hi halves of h2 rows
__host__ __device__ inline int x_col_halves(bool x_h2, int c0) { return x_h2 ? ((c0 >> 5) * 64 + (c0 & 31)) : c0; }
__host__ __device__ inline int x_row_halves(bool x_h2, int C) { return x_h2 ? 2 * C : C; }

template <typename T, int BMO, int BNO>
__global__ void __launch_bounds__(256) wgrad_kernel(WgradArgs a) {
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int BKM = 32;              // m rows per step
  constexpr int YCP = BMO / VEC;       // column pieces of the dY tile
  constexpr int XCP = BNO / VEC;       // column pieces of the X tile
  constexpr int YPT = BKM * YCP / 256; // pieces per thread
  constexpr int XPT = BKM * XCP / 256;
  constexpr int YRS = 256 / YCP;       // row stride between a thread's pieces
  constexpr int XRS = 256 / XCP;
  constexpr int TM = BMO / 64, TN = BNO / 64;  // 2x2 waves, 32x32 MFMA tiles
  static_assert(YPT >= 1 && XPT >= 1, "tile too small");

  __shared__ T ldsY[2][BKM][BMO];
  __shared__ T ldsX[2][BKM][BNO];

  const GatherGeom& g = a.g;
  const T* dY = reinterpret_cast<const T*>(a.dY);
  const T* X = reinterpret_cast<const T*>(a.X);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * BMO;  // cout tile
  const int k0 = blockIdx.y * BNO;  // (r,s,c) column tile
  const int m_begin = blockIdx.z * a.rows_per_split;
  const int m_end = min(g.M, m_begin + a.rows_per_split);

  // dY loader
  const int ycp = t % YCP, yrow = t / YCP;
  const bool y_col_ok = n0 + ycp * VEC < g.N;
  // X loader: fixed column piece -> fixed tap
  const int xcp = t % XCP, xrow = t / XCP;
  const int kcol = k0 + xcp * VEC;
  const bool x_col_ok = kcol < g.K;
  const int tap = x_col_ok ? kcol / g.C : 0;
  const int c0 = x_col_ok ? kcol % g.C : 0;
  const int dh = g.rsign * (tap / g.S), dw = g.ssign * (tap % g.S);
  // (x_h2: hi halves of an h2 tensor, fp16 instantiations only)
  const bool xh2 = sizeof(T) == 2 && a.x_h2;
  const int xc0 = x_col_halves(xh2, c0), xrow_h = x_row_halves(xh2, g.C);
  int xb[XPT], xp[XPT], xq[XPT];
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    int m = m_begin + xrow + i * XRS;
    xq[i] = m % g.Q;
    int tmp = m / g.Q;
    xp[i] = tmp % g.P;
    xb[i] = tmp / g.P;
  }

  piece_t ry[YPT], rx[XPT];
  auto load_tile = [&](int mt) {
#pragma unroll
    for (int i = 0; i < YPT; ++i) {
      int m = mt + yrow + i * YRS;
      ry[i] = (y_col_ok && m < m_end) ? *reinterpret_cast<const piece_t*>(dY + (long)m * a.ldy + n0 + ycp * VEC)
                                      : zero_piece();
    }
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      int m = mt + xrow + i * XRS;
      int hn = xp[i] * g.mul_p + g.off_h + dh, wn = xq[i] * g.mul_q + g.off_w + dw;
      bool ok = x_col_ok && m < m_end;
      if (g.div == 2) {
        ok = ok && ((hn | wn) & 1) == 0;
        hn >>= 1;
        wn >>= 1;
      }
      ok = ok && (unsigned)hn < (unsigned)g.Hi && (unsigned)wn < (unsigned)g.Wi;
      rx[i] = ok ? *reinterpret_cast<const piece_t*>(X + ((long)((xb[i] * g.Hi + hn) * g.Wi + wn) * xrow_h + xc0))
                 : zero_piece();
      // advance (b, p, q) by BKM rows
      xq[i] += BKM;
      while (xq[i] >= g.Q) {
        xq[i] -= g.Q;
        if (++xp[i] == g.P) {
          xp[i] = 0;
          ++xb[i];
        }
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < YPT; ++i)
      *reinterpret_cast<piece_t*>(&ldsY[buf][yrow + i * YRS][ycp * VEC]) = ry[i];
#pragma unroll
    for (int i = 0; i < XPT; ++i)
      *reinterpret_cast<piece_t*>(&ldsX[buf][xrow + i * XRS][xcp * VEC]) = rx[i];
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (m_begin < m_end) {
    load_tile(m_begin);
    store_tile(0);
  }
  __syncthreads();
  int cur = 0;
  for (int mt = m_begin; mt < m_end; mt += BKM) {
    const bool more = mt + BKM < m_end;
    if (more) load_tile(mt + BKM);
#pragma unroll
    for (int ks = 0; ks < BKM / (2 * VEC); ++ks) {
      PieceView<T> fa[TM], fb[TN];
      const int kb = ks * 2 * VEC + (lane >> 5) * VEC;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        int col = wm * (BMO / 2) + i * 32 + (lane & 31);
#pragma unroll
        for (int e = 0; e < VEC; ++e) fa[i].e[e] = ldsY[cur][kb + e][col];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        int col = wn * (BNO / 2) + j * 32 + (lane & 31);
#pragma unroll
        for (int e = 0; e < VEC; ++e) fb[j].e[e] = ldsX[cur][kb + e][col];
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

  // ---- atomic accumulation into dW ----------------------------------------------------------
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      int kc = k0 + wn * (BNO / 2) + j * 32 + (lane & 31);
      int dst = kc < g.K ? (a.colmap ? a.colmap[kc] : kc) : -1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int n = n0 + wm * (BMO / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (n < g.N && dst >= 0) unsafeAtomicAdd(a.dW + (long)blockIdx.z * a.split_stride + (long)n * a.ldw + dst, acc[i][j][r] * a.alpha);
      }
    }
}

// ---- fp16 weight gradient: LDS-DMA staging + hardware transpose reads ---------------------------------
// Tiles are DMA'd as stored ([m][channel], 16-byte pieces, double-buffered, 64 rows of m per step) and the
// MFMA fragments (8 consecutive m for one channel) come from two ds_read_b64_tr_b16 each instead of eight
// 16-bit LDS reads.  Transpose-read semantics (measured, tools/probes/tr_probe.hip): in every group of 16
// lanes, result lane i / element e is element i%4 of the 8-byte chunk addressed by source lane 4e + i/4;
// so source lane 4e+j addresses row (k0 + e), channels [c0 + 4j, c0 + 4j + 4) and lane i receives
// k0..k0+3 of channel c0 + i.  The 16-byte piece index of a row is XOR-swizzled with the row (on the DMA
// source side and in the read address) so the four rows of a transpose read fall in distinct bank groups.
typedef short v4s16 __attribute__((__vector_size__(8)));
__device__ __forceinline__ v4s16 ds_read_tr16(const void* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s16 __attribute__((address_space(3)))*)p);
}
union TrFrag {
  v4s16 h[2];
  half8 v;
};
template <int PIECES_PER_ROW>
__device__ __forceinline__ int wg_swz(int row) {
  return (row & 3) << (PIECES_PER_ROW == 16 ? 2 : 1);
}

// ASMRD (wgrad_dma_kernel): the transpose reads are issued from inline assembly.  hipcc treats the builtin
// ds_read_tr16_b64 as a read that may alias an LDS-DMA in flight and puts `s_waitcnt vmcnt(0)` in front of it -- in this
// loop that is a wait for the NEXT tile's DMA, issued a few instructions earlier, before the current tile is read: the
// double buffer degenerates into load-wait-compute (plain ds_read_b128 loads, as in igemm.h, are not treated this way).
// Assembly reads are invisible to the compiler's wait insertion, so the kernel places its own counted
// `s_waitcnt lgkmcnt` between the reads and the MFMAs that consume them; the waits take the fragments as in/out
// operands, which keeps the MFMAs behind them.
// (`addr` = byte address inside the workgroup's LDS allocation; the emulator build evaluates the same address
// arithmetic through the builtin, relative to `lds_base`.)
#ifdef __HIP_DEVICE_COMPILE__
__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return (unsigned)(size_t)(las_ptr_t) const_cast<void*>(p); }
template <int OFF>
__device__ __forceinline__ v4s16 ds_read_tr16_at(const void*, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  v4s16 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
template <int N>
__device__ __forceinline__ void wait_lgkmcnt_for(TrFrag& a) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt field");
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a.h[0]), "+v"(a.h[1]) : "n"(N));
}
#else  // emulator build, and hipcc's host pass over the kernel bodies
__device__ __forceinline__ unsigned lds_addr_of(const void*) { return 0u; }
template <int OFF>
__device__ __forceinline__ v4s16 ds_read_tr16_at(const void* lds_base, unsigned addr) {
  return ds_read_tr16(reinterpret_cast<const char*>(lds_base) + addr + OFF);
}
template <int N>
__device__ __forceinline__ void wait_lgkmcnt_for(TrFrag&) {}
#endif
// (StaticIndex / static_for: common.h)

//
// FAST (stride-1 "same" convolutions, the bulk of the network): the source address of tap (dh, dw) of pixel
// m is linear in m, so a lane's byte offset is a constant plus a per-step scalar; only the validity of the
// tap depends on (p, q), and that comes from a per-block LDS table (16-bit mask per pixel of one image,
// built once) instead of being recomputed for every 16-byte piece.  Loads go through buffer resources: an
// invalid tap / column / row past M is an all-ones offset and the bounds check returns zero.
constexpr int WG_TBL = 6144;  // pixels of one image the validity table can hold
template <int BMO, int BNO, int BKM, int MINW, bool FAST, bool ASMRD = false>
static __global__ void __launch_bounds__(256, MINW) wgrad_dma_kernel(WgradArgs a, const half* __restrict__ zero_page) {
  constexpr int VEC = 8;  // BKM = m rows per step (32 or 64)
  constexpr int YCP = BMO / VEC, XCP = BNO / VEC;
  constexpr int YPT = BKM * YCP / 256, XPT = BKM * XCP / 256;  // DMA instructions per thread per tile
  constexpr int YRS = 256 / YCP, XRS = 256 / XCP;              // rows covered per DMA pass
  constexpr int TM = BMO / 64, TN = BNO / 64;
  constexpr int TILE_Y = BKM * BMO, TILE_X = BKM * BNO;        // halves
  // ONE LDS object: [2 tiles of dY and X][validity table (FAST)]
  __shared__ half smem[2 * (TILE_Y + TILE_X) + (FAST ? WG_TBL : 0)] __attribute__((aligned(16)));
  unsigned short* tbl = reinterpret_cast<unsigned short*>(&smem[2 * (TILE_Y + TILE_X)]);

  const GatherGeom& g = a.g;
  const GatherGeom& g_ = a.g;
  const half* dY = reinterpret_cast<const half*>(a.dY);
  const half* X = reinterpret_cast<const half*>(a.X);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware order over a 1-D grid: tiles of one m-range (same split) are adjacent on one XCD, so the
  // dY / X rows they share are fetched into that XCD's L2 once
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int gx = ceil_div(g_.N, BMO), gy = ceil_div(g_.K, BNO);
  const int bx = logical % gx, by = (logical / gx) % gy, bz = logical / (gx * gy);
  const int n0 = bx * BMO, k0 = by * BNO;
  const int m_begin = bz * a.rows_per_split;
  const int m_end = min(g.M, m_begin + a.rows_per_split);

  // dY loader: LDS position (row, slot) holds source piece slot ^ swz(row); rows of one thread differ by
  // multiples of YRS (>= 16), invisible to the swizzle, so the source piece is fixed per thread
  const int yslot = t % YCP, yrow = t / YCP;
  const int ypiece = yslot ^ wg_swz<YCP>(yrow);
  const bool y_ok = n0 + ypiece * VEC < g.N;
  const int xslot = t % XCP, xrow = t / XCP;
  const int xpiece = xslot ^ wg_swz<XCP>(xrow);
  const int kcol = k0 + xpiece * VEC;
  const bool x_ok = kcol < g.K;
  const int tap = x_ok ? kcol / g.C : 0;
  const int c0 = x_ok ? kcol % g.C : 0;
  const int dh = g.rsign * (tap / g.S), dw = g.ssign * (tap % g.S);
  const int xc0 = x_col_halves(a.x_h2, c0), xrow_h = x_row_halves(a.x_h2, g.C);  // (x_h2: hi halves of an h2 tensor)
  int xb[XPT], xp[XPT], xq[XPT];
  // FAST state
  const int PQ = g.P * g.Q;
  unsigned y_fix[YPT], x_fix[XPT];
  int x_pix[XPT];
  const unsigned y_bad = y_ok ? 0u : ~0u, x_bad = x_ok ? 0u : ~0u;
  __amdgpu_buffer_rsrc_t rsrc_y, rsrc_x;
  if constexpr (FAST) {
    rsrc_y = make_rsrc(dY, (long)g.M * a.ldy * 2);
    rsrc_x = make_rsrc(X, (long)g.B * g.Hi * g.Wi * xrow_h * 2);
    // validity table: bit `tap` of tbl[pix] is set when that tap of pixel pix = p*Q + q is outside the image
    for (int pix = t; pix < PQ; pix += 256) {
      const int p = pix / g.Q, q = pix - p * g.Q;
      unsigned inv = 0;
      int tp = 0;
      for (int r = 0; r < g.R; ++r)
        for (int s = 0; s < g.S; ++s, ++tp) {
          const int hn = p + g.off_h + g.rsign * r, wn_ = q + g.off_w + g.ssign * s;
          if (!((unsigned)hn < (unsigned)g.Hi && (unsigned)wn_ < (unsigned)g.Wi)) inv |= 1u << tp;
        }
      tbl[pix] = (unsigned short)inv;
    }
#pragma unroll
    for (int i = 0; i < YPT; ++i)
      y_fix[i] = (unsigned)(((m_begin + yrow + i * YRS) * a.ldy + n0 + ypiece * VEC) * 2);
    const int shift = (g.off_h + dh) * g.Wi + g.off_w + dw;  // pixel offset of this lane's tap
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int m = m_begin + xrow + i * XRS;
      x_fix[i] = (unsigned)(((m + shift) * xrow_h + xc0) * 2);
      x_pix[i] = m % PQ;
    }
  } else {
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      int m = m_begin + xrow + i * XRS;
      xq[i] = m % g.Q;
      int tmp = m / g.Q;
      xp[i] = tmp % g.P;
      xb[i] = tmp / g.P;
    }
  }
  unsigned step_y = 0, step_x = 0;  // byte offsets of the current step from m_begin (scalar)

  auto issue_tile = [&](int mt, int buf) {
    half* ty = &smem[buf * (TILE_Y + TILE_X)];
    half* tx = ty + TILE_Y;
    if constexpr (FAST) {
#pragma unroll
      for (int i = 0; i < YPT; ++i)
        dma16(rsrc_y, (y_fix[i] + step_y) | y_bad, 0u, ty + (i * YRS * YCP + wave * 64) * VEC);
#pragma unroll
      for (int i = 0; i < XPT; ++i) {
        const unsigned inv = (unsigned)__builtin_amdgcn_sbfe((int)tbl[x_pix[i]], tap, 1);
        dma16(rsrc_x, (x_fix[i] + step_x) | inv | x_bad, 0u, tx + (i * XRS * XCP + wave * 64) * VEC);
        x_pix[i] += BKM;  // BKM <= PQ (launch condition): one conditional wrap
        x_pix[i] -= x_pix[i] >= PQ ? PQ : 0;
      }
      step_y += (unsigned)(BKM * a.ldy * 2);
      step_x += (unsigned)(BKM * xrow_h * 2);
      return;
    }
#pragma unroll
    for (int i = 0; i < YPT; ++i) {
      const int m = mt + yrow + i * YRS;
      const half* src = (y_ok && m < m_end) ? dY + ((long)m * a.ldy + n0 + ypiece * VEC) : zero_page;
      dma16(src, ty + (i * YRS * YCP + wave * 64) * VEC);
    }
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int m = mt + xrow + i * XRS;
      int hn = xp[i] * g.mul_p + g.off_h + dh, wn_ = xq[i] * g.mul_q + g.off_w + dw;
      bool ok = x_ok && m < m_end;
      if (g.div == 2) {
        ok = ok && ((hn | wn_) & 1) == 0;
        hn >>= 1;
        wn_ >>= 1;
      }
      ok = ok && (unsigned)hn < (unsigned)g.Hi && (unsigned)wn_ < (unsigned)g.Wi;
      const half* src = ok ? X + ((long)((xb[i] * g.Hi + hn) * g.Wi + wn_) * xrow_h + xc0) : zero_page;
      dma16(src, tx + (i * XRS * XCP + wave * 64) * VEC);
      xq[i] += BKM;
      while (xq[i] >= g.Q) {
        xq[i] -= g.Q;
        if (++xp[i] == g.P) {
          xp[i] = 0;
          ++xb[i];
        }
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

  // transpose-read lane geometry: 16-lane group gq = lane/16 -> column block (gq&1)*16, k half (gq>>1)*8;
  // as SOURCE lane 4e+j this lane addresses row e = (lane&15)/4, chunk j = lane&3
  const int gq = lane >> 4, i16 = lane & 15;
  const int src_row = i16 >> 2, src_chunk = (i16 & 3) * 4 + (gq & 1) * 16;
  const int kgrp = (gq >> 1) * 8;

  if constexpr (FAST) __syncthreads();  // validity table complete
  if (m_begin < m_end) issue_tile(m_begin, 0);
  int cur = 0;
  for (int mt = m_begin; mt < m_end; mt += BKM) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // tile landed for everyone; everyone is done with the other buffer
    if (mt + BKM < m_end) issue_tile(mt + BKM, cur ^ 1);
    const half* ty = &smem[cur * (TILE_Y + TILE_X)];
    const half* tx = ty + TILE_Y;
    if constexpr (ASMRD) {
      // Lane part of the addresses: row (kgrp + src_row) and the swizzled piece -- the swizzle only sees row & 3 =
      // src_row, so (ks, hlf) are immediate offsets.  K sub-steps go in pairs: all reads of the pair are issued, the
      // first sub-step's MFMAs start once ITS reads have returned (the second's stay in flight under them).
      static_assert((BKM / 16) % 2 == 0, "sub-steps are processed in pairs");
      constexpr int RPK = (TM + TN) * 2;  // reads per sub-step
      static_assert(RPK <= 15, "lgkmcnt field");
      const unsigned bo = lds_addr_of(smem) + (unsigned)(cur * (TILE_Y + TILE_X) * 2);
      unsigned aA[TM], aB[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int col = wm * (BMO / 2) + i * 32 + src_chunk;
        aA[i] = bo + (unsigned)(((kgrp + src_row) * BMO + (((col >> 3) ^ wg_swz<YCP>(src_row)) * 8) + (col & 7)) * 2);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = wn * (BNO / 2) + j * 32 + src_chunk;
        aB[j] = bo + (unsigned)((TILE_Y + (kgrp + src_row) * BNO + (((col >> 3) ^ wg_swz<XCP>(src_row)) * 8) + (col & 7)) * 2);
      }
      static_for<BKM / 32>([&](auto KP) {
        constexpr int kp = decltype(KP)::value;
        TrFrag fa[2][TM], fb[2][TN];
        __builtin_amdgcn_sched_barrier(0);
        static_for<2>([&](auto KS) {
          constexpr int ks = kp * 2 + decltype(KS)::value, s = decltype(KS)::value;
          static_for<2>([&](auto HL) {
            constexpr int hlf = decltype(HL)::value;
            static_for<TM>([&](auto I) {
              fa[s][decltype(I)::value].h[hlf] = ds_read_tr16_at<(ks * 16 + hlf * 4) * BMO * 2>(smem, aA[decltype(I)::value]);
            });
            static_for<TN>([&](auto J) {
              fb[s][decltype(J)::value].h[hlf] = ds_read_tr16_at<(ks * 16 + hlf * 4) * BNO * 2>(smem, aB[decltype(J)::value]);
            });
          });
        });
        static_for<2>([&](auto KS) {
          constexpr int s = decltype(KS)::value;
          // LDS reads return in order: at most RPK outstanding = the first sub-step's fragments are in their registers
#pragma unroll
          for (int i = 0; i < TM; ++i) wait_lgkmcnt_for<s == 0 ? RPK : 0>(fa[s][i]);
#pragma unroll
          for (int j = 0; j < TN; ++j) wait_lgkmcnt_for<s == 0 ? RPK : 0>(fb[s][j]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s][i].v, fb[s][j].v, acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);  // the second sub-step's waits stay behind these MFMAs
        });
      });
      cur ^= 1;
      continue;
    }
#pragma unroll
    for (int ks = 0; ks < BKM / 16; ++ks) {
      TrFrag fa[TM], fb[TN];
#pragma unroll
      for (int hlf = 0; hlf < 2; ++hlf) {
        const int row = ks * 16 + kgrp + hlf * 4 + src_row;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int col = wm * (BMO / 2) + i * 32 + src_chunk;
          const int piece = (col >> 3) ^ wg_swz<YCP>(row);
          fa[i].h[hlf] = ds_read_tr16(ty + row * BMO + piece * 8 + (col & 7));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = wn * (BNO / 2) + j * 32 + src_chunk;
          const int piece = (col >> 3) ^ wg_swz<XCP>(row);
          fb[j].h[hlf] = ds_read_tr16(tx + row * BNO + piece * 8 + (col & 7));
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i].v, fb[j].v, acc[i][j], 0, 0, 0);
    }
    cur ^= 1;
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      int kc = k0 + wn * (BNO / 2) + j * 32 + (lane & 31);
      int dst = kc < g.K ? (a.colmap ? a.colmap[kc] : kc) : -1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int n = n0 + wm * (BMO / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (n < g.N && dst >= 0)
          unsafeAtomicAdd(a.dW + (long)bz * a.split_stride + (long)n * a.ldw + dst, acc[i][j][r] * a.alpha);
      }
    }
}

// ---- fp32 tensors on the bf16 matrix pipe (MMA_BF16X3, common.h): the weight gradient of the fp32x3 mode -----------------
// (The 3x3 stride-1 layers -- 29 of 37 launches -- go to wgrad_fused.h's wgrad_fused_x3_kernel, which reads the pixels once for
// all nine taps: 232-260 us per launch against this kernel's 336-530; this one keeps the stride-2 / 1x1 / stem shapes.)
// wgrad_kernel's loader (any gather: stride, taps, column map; 16-byte pieces of 4 floats through registers) + the fp16
// kernel's reader: every element is split into hi + lo bf16 halves ONCE, on its way from the staging registers into LDS
// (two bf16 planes per operand, rows = m, 16-byte pieces XOR-swizzled with the row as above), and a fragment -- 8
// consecutive m of one channel -- is two ds_read_b64_tr_b16 per plane.  dW += alpha (Yh Xh + Yh Xl + Yl Xh): three MFMAs
// per product, 2^-16 relative per operand.  (The first x3 form converted after the LDS read, per fragment: every element
// was converted by both waves that read it and fetched with eight 4-byte LDS reads -- 134-195 TF on MI355X, round 3.)
union TrFragB {
  v4s16 h[2];
  bf16x8 v;
  half8 f;
};
// the hi halves and the lo halves of channels c .. c + 3 (c % 4 == 0) of row `row` of an h2 tensor with C channels: {hi, hi, lo, lo}
__device__ __forceinline__ piece_t h2_load4(const half* base, long row, int C, int c) {
  const half* p = base + h2_index(row, C, c);
  const u32x2 hi = *reinterpret_cast<const u32x2*>(p), lo = *reinterpret_cast<const u32x2*>(p + 32);
  piece_t r = {hi[0], hi[1], lo[0], lo[1]};
  return r;
}
// H2 (round 4, MMA_H2): dY and X are h2 tensors (common.h) -- already split: a thread fetches the 4 hi and the 4 lo halves of its
// four channels (two 8-byte loads) and stores them to the planes as they are; fp16 MFMAs.  (Stride-2 / 1x1 shapes only: the
// 3x3 stride-1 layers go to wgrad_fused.h's DMA-fed h2 kernel.)
template <int BMO, int BNO, bool H2 = false>
static __global__ void __launch_bounds__(256, 2) wgrad_x3_kernel(WgradArgs a) {
  constexpr int BKM = 32;                       // m rows per step
  constexpr int YCP = BMO / 4, XCP = BNO / 4;   // float pieces per row
  constexpr int YPT = BKM * YCP / 256, XPT = BKM * XCP / 256;
  constexpr int YRS = 256 / YCP, XRS = 256 / XCP;
  constexpr int TM = BMO / 64, TN = BNO / 64;   // 2x2 waves, 32x32 MFMA tiles
  constexpr int PY = BKM * BMO, PX = BKM * BNO; // elements of one plane
  constexpr int BUF = 2 * PY + 2 * PX;          // [Y hi][Y lo][X hi][X lo]
  __shared__ unsigned short smem[2 * BUF] __attribute__((aligned(16)));

  const GatherGeom& g = a.g;
  const float* dY = reinterpret_cast<const float*>(a.dY);
  const float* X = reinterpret_cast<const float*>(a.X);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * BMO, k0 = blockIdx.y * BNO;
  const int m_begin = blockIdx.z * a.rows_per_split;
  const int m_end = min(g.M, m_begin + a.rows_per_split);

  const int ycp = t % YCP, yrow = t / YCP;
  const bool y_col_ok = n0 + ycp * 4 < g.N;
  const int xcp = t % XCP, xrow = t / XCP;
  const int kcol = k0 + xcp * 4;
  const bool x_col_ok = kcol < g.K;
  const int tap = x_col_ok ? kcol / g.C : 0;
  const int c0 = x_col_ok ? kcol % g.C : 0;
  const int dh = g.rsign * (tap / g.S), dw = g.ssign * (tap % g.S);
  int xb[XPT], xp[XPT], xq[XPT];
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    const int m = m_begin + xrow + i * XRS;
    xq[i] = m % g.Q;
    const int tmp = m / g.Q;
    xp[i] = tmp % g.P;
    xb[i] = tmp / g.P;
  }
  piece_t ry[YPT], rx[XPT];
  auto load_tile = [&](int mt) {
#pragma unroll
    for (int i = 0; i < YPT; ++i) {
      const int m = mt + yrow + i * YRS;
      if constexpr (H2)
        ry[i] = (y_col_ok && m < m_end) ? h2_load4(reinterpret_cast<const half*>(a.dY), m, a.ldy, n0 + ycp * 4) : zero_piece();
      else
      ry[i] = (y_col_ok && m < m_end) ? *reinterpret_cast<const piece_t*>(dY + (long)m * a.ldy + n0 + ycp * 4) : zero_piece();
    }
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int m = mt + xrow + i * XRS;
      int hn = xp[i] * g.mul_p + g.off_h + dh, wn_ = xq[i] * g.mul_q + g.off_w + dw;
      bool ok = x_col_ok && m < m_end;
      if (g.div == 2) {
        ok = ok && ((hn | wn_) & 1) == 0;
        hn >>= 1;
        wn_ >>= 1;
      }
      ok = ok && (unsigned)hn < (unsigned)g.Hi && (unsigned)wn_ < (unsigned)g.Wi;
      if constexpr (H2)
        rx[i] = ok ? h2_load4(reinterpret_cast<const half*>(a.X), (long)(xb[i] * g.Hi + hn) * g.Wi + wn_, g.C, c0) : zero_piece();
      else
      rx[i] = ok ? *reinterpret_cast<const piece_t*>(X + ((long)((xb[i] * g.Hi + hn) * g.Wi + wn_) * g.C + c0)) : zero_piece();
      xq[i] += BKM;
      while (xq[i] >= g.Q) {
        xq[i] -= g.Q;
        if (++xp[i] == g.P) {
          xp[i] = 0;
          ++xb[i];
        }
      }
    }
  };
  // 4 floats -> 4 hi + 4 lo bf16 at (row, columns col .. col + 3) of a plane pair whose rows are W elements long
  auto split_store = [&](piece_t raw, unsigned short* hi_plane, unsigned short* lo_plane, int row, int col, int W, int swz) {
    PieceView<float> v;
    v.p = raw;
    union {
      __bf16 b[4];
      u32x2 u;
    } h, l;
    if constexpr (H2) {  // {hi halves, lo halves} as loaded
      h.u[0] = raw[0];
      h.u[1] = raw[1];
      l.u[0] = raw[2];
      l.u[1] = raw[3];
    } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h.b[e] = (__bf16)v.e[e];
      l.b[e] = (__bf16)(v.e[e] - (float)h.b[e]);
    }
    }
    const int at = row * W + (((col >> 3) ^ swz) * 8) + (col & 7);
    *reinterpret_cast<u32x2*>(hi_plane + at) = h.u;
    *reinterpret_cast<u32x2*>(lo_plane + at) = l.u;
  };
  auto store_tile = [&](int buf) {
    unsigned short* base = &smem[buf * BUF];
#pragma unroll
    for (int i = 0; i < YPT; ++i) {
      const int row = yrow + i * YRS;
      split_store(ry[i], base, base + PY, row, ycp * 4, BMO, wg_swz<BMO / 8>(row));
    }
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int row = xrow + i * XRS;
      split_store(rx[i], base + 2 * PY, base + 2 * PY + PX, row, xcp * 4, BNO, wg_swz<BNO / 8>(row));
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transpose-read lane geometry (wgrad_dma_kernel)
  const int gq = lane >> 4, i16 = lane & 15;
  const int src_row = i16 >> 2, src_chunk = (i16 & 3) * 4 + (gq & 1) * 16;
  const int kgrp = (gq >> 1) * 8;

  if (m_begin < m_end) {
    load_tile(m_begin);
    store_tile(0);
  }
  __syncthreads();
  int cur = 0;
  for (int mt = m_begin; mt < m_end; mt += BKM) {
    const bool more = mt + BKM < m_end;
    if (more) load_tile(mt + BKM);  // in flight under this step's reads and MFMAs
    const unsigned short* yh = &smem[cur * BUF];
    const unsigned short* yl = yh + PY;
    const unsigned short* xh = yh + 2 * PY;
    const unsigned short* xl = xh + PX;
#pragma unroll
    for (int ks = 0; ks < BKM / 16; ++ks) {
      TrFragB ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int hlf = 0; hlf < 2; ++hlf) {
        const int row = ks * 16 + kgrp + hlf * 4 + src_row;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int col = wm * (BMO / 2) + i * 32 + src_chunk;
          const int at = row * BMO + (((col >> 3) ^ wg_swz<BMO / 8>(row)) * 8) + (col & 7);
          ah[i].h[hlf] = ds_read_tr16(yh + at);
          al[i].h[hlf] = ds_read_tr16(yl + at);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = wn * (BNO / 2) + j * 32 + src_chunk;
          const int at = row * BNO + (((col >> 3) ^ wg_swz<BNO / 8>(row)) * 8) + (col & 7);
          bh[j].h[hlf] = ds_read_tr16(xh + at);
          bl[j].h[hlf] = ds_read_tr16(xl + at);
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (H2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i].f, bh[j].f, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i].f, bl[j].f, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i].f, bh[j].f, acc[i][j], 0, 0, 0);
          } else {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i].v, bh[j].v, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i].v, bl[j].v, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i].v, bh[j].v, acc[i][j], 0, 0, 0);
          }
        }
    }
    if (more) store_tile(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int kc = k0 + wn * (BNO / 2) + j * 32 + (lane & 31);
      const int dst = kc < g.K ? (a.colmap ? a.colmap[kc] : kc) : -1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wm * (BMO / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (n < g.N && dst >= 0)
          unsafeAtomicAdd(a.dW + (long)blockIdx.z * a.split_stride + (long)n * a.ldw + dst, acc[i][j][r] * a.alpha);
      }
    }
}

template <typename T>
struct WgradDma {
  static bool launch(const WgradArgs&, dim3, int, int, hipStream_t, const void*) { return false; }
};
template <>
struct WgradDma<half> {
  template <int BKM, int MINW>
  static void go(const WgradArgs& a, dim3 grid, int bmo, int bno, hipStream_t stream, const half* zp) {
    dim3 block(256);
    const GatherGeom& g = a.g;
    const bool fast = g.mul_p == 1 && g.mul_q == 1 && g.div == 1 && g.P == g.Hi && g.Q == g.Wi && g.R * g.S <= 16 &&
                      g.P * g.Q <= WG_TBL && g.P * g.Q >= BKM && g.C % 8 == 0 && (long)g.M * a.ldy * 2 < 0xfffffff0l &&
                      (long)g.M * x_row_halves(a.x_h2, g.C) * 2 < 0xfffffff0l;
    // fast form: transpose reads from inline assembly with hand-placed waits (ASMRD above); measured on MI355X (round 2):
    // layer1 189 -> 155 us, layer2 140 -> 122, layer3 144 -> 124, layer4 129 -> 117 against the builtin reads
    if (fast) {
      if (bmo == 64 && bno == 64)
        hipLaunchKernelGGL((wgrad_dma_kernel<64, 64, BKM, MINW, true, true>), grid, block, 0, stream, a, zp);
      else if (bmo == 64)
        hipLaunchKernelGGL((wgrad_dma_kernel<64, 128, BKM, MINW, true, true>), grid, block, 0, stream, a, zp);
      else if (bno == 64)
        hipLaunchKernelGGL((wgrad_dma_kernel<128, 64, BKM, MINW, true, true>), grid, block, 0, stream, a, zp);
      else  // (the 128x128 tile needs more than the 128 registers of 4 waves per SIMD: 3 requested = what it runs at)
        hipLaunchKernelGGL((wgrad_dma_kernel<128, 128, BKM, (MINW > 3 ? 3 : MINW), true, true>), grid, block, 0, stream, a, zp);
      return;
    }
    if (bmo == 64 && bno == 64)
      hipLaunchKernelGGL((wgrad_dma_kernel<64, 64, BKM, MINW, false>), grid, block, 0, stream, a, zp);
    else if (bmo == 64)
      hipLaunchKernelGGL((wgrad_dma_kernel<64, 128, BKM, MINW, false>), grid, block, 0, stream, a, zp);
    else if (bno == 64)
      hipLaunchKernelGGL((wgrad_dma_kernel<128, 64, BKM, MINW, false>), grid, block, 0, stream, a, zp);
    else
      hipLaunchKernelGGL((wgrad_dma_kernel<128, 128, BKM, MINW, false>), grid, block, 0, stream, a, zp);
  }
  static bool launch(const WgradArgs& a, dim3 grid3, int bmo, int bno, hipStream_t stream, const void* zero_page) {
    const half* zp = reinterpret_cast<const half*>(zero_page);
    if (!zp || (a.rows_per_split % 64) != 0) return false;
    dim3 grid(grid3.x * grid3.y * grid3.z);  // 1-D: the kernel derives (cout tile, k tile, split) itself
    // measured on MI355X (tools/conv_bench.py): 32-row steps with registers capped for 4 blocks/CU beat 64-row steps at
    // 2 blocks/CU on every 3x3 layer
    go<32, 4>(a, grid, bmo, bno, stream, zp);
    return true;
  }
};

// wgrad_fused.h: 3x3 stride-1 fp16 layers with the nine taps of a channel tile accumulated from one LDS-resident pass
// over the pixels (MN_WGRAD_FUSED=0 restores the plain GEMM form below for them)
inline bool wgrad_fused_applies(const WgradArgs& a);
inline bool wgrad_fused_x3_applies(const WgradArgs& a);
inline bool wgrad_fused_h2_applies(const WgradArgs& a);
inline void launch_wgrad_fused(const WgradArgs& a, int target_blocks, hipStream_t stream, int form);

// dW[i] += sum over the split slices, in split order (the slices already carry alpha)
static __global__ void __launch_bounds__(256) wgrad_split_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dW,
                                                                        long n, int splits, long stride) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += ws[(long)z * stride + i];
    dW[i] += v;
  }
}
inline void launch_zero_fill(float* p, long n, hipStream_t s);  // optim.h

template <typename T>
inline void launch_wgrad(WgradArgs a, int target_blocks, hipStream_t stream, const void* zero_page = nullptr) {
  const GatherGeom& g = a.g;
  static const bool fused = !(getenv("MN_WGRAD_FUSED") && atoi(getenv("MN_WGRAD_FUSED")) == 0);
  if constexpr (ElemTraits<T>::DTYPE == MN_F16) {
    if (g.mma == MMA_H2) {  // h2 tensors: the DMA-fed tap-fused kernel, or (stride-2 / 1x1 shapes) the generic loader
      if (fused && wgrad_fused_h2_applies(a)) {
        launch_wgrad_fused(a, target_blocks, stream, 2);
        return;
      }
    } else if (fused && wgrad_fused_applies(a)) {
      launch_wgrad_fused(a, target_blocks, stream, 0);
      return;
    }
  } else {
    if (fused && wgrad_fused_x3_applies(a)) {  // fp32 tensors, bf16x3 contraction
      launch_wgrad_fused(a, target_blocks, stream, 1);
      return;
    }
  }
  static const bool stem_wide = !(getenv("MN_STEM_WGRAD_WIDE") && atoi(getenv("MN_STEM_WGRAD_WIDE")) == 0);
  bool narrow_k = g.K <= 64 || (g.K % 128 != 0 && g.K < 256);
  // the stem's weight gradient on fp32 tensors (K = 224 columns): two 128-column tiles read the 1 GB d(conv output) twice,
  // four 64-column tiles four times
  // (ONE 256-column tile, d(conv output) read once: 29.55 vs 29.57 ms per step -- the launch runs on the side stream, which is not
  //  the critical path; measured and not kept, round 4 call 33)
  if (stem_wide && sizeof(T) == 4 && g.mma == MMA_BF16X3 && g.K > 128) narrow_k = false;
  const bool narrow_n = g.N <= 64;
  int bmo = narrow_n ? 64 : 128, bno = narrow_k ? 64 : 128;
  int tiles = cdiv(g.N, bmo) * cdiv(g.K, bno);
  int splits = cdiv(target_blocks, tiles);
  int max_splits = cdiv(g.M, 512);  // at least 8 steps of 64 rows per block
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  const long slice = (long)g.N * a.ldw;
  float* const dW_final = a.dW;
  if (a.det) {  // one zeroed slice of the workspace per split (fewer splits if the workspace is small; none without one)
    const int fit = a.ws ? (int)(a.ws_floats / slice) : 1;
    if (splits > fit) splits = fit < 1 ? 1 : fit;
  }
  int rows = cdiv(cdiv(g.M, splits), 64) * 64;
  splits = cdiv(g.M, rows);
  a.rows_per_split = rows;
  const bool sliced = a.det && splits > 1;
  if (sliced) {
    launch_zero_fill(a.ws, slice * splits, stream);
    a.dW = a.ws;
    a.split_stride = slice;
  }
  dim3 grid(cdiv(g.N, bmo), cdiv(g.K, bno), splits), block(256);
  if (sizeof(T) == 2 && g.mma == MMA_H2) {
    if (bmo == 64 && bno == 64)
      hipLaunchKernelGGL((wgrad_x3_kernel<64, 64, true>), grid, block, 0, stream, a);
    else if (bmo == 64)
      hipLaunchKernelGGL((wgrad_x3_kernel<64, 128, true>), grid, block, 0, stream, a);
    else if (bno == 64)
      hipLaunchKernelGGL((wgrad_x3_kernel<128, 64, true>), grid, block, 0, stream, a);
    else
      hipLaunchKernelGGL((wgrad_x3_kernel<128, 128, true>), grid, block, 0, stream, a);
  } else if (sizeof(T) == 4 && g.mma == MMA_BF16X3) {
    if constexpr (sizeof(T) == 4) {
      if (bmo == 64 && bno == 64)
        hipLaunchKernelGGL((wgrad_x3_kernel<64, 64>), grid, block, 0, stream, a);
      else if (bmo == 64)
        hipLaunchKernelGGL((wgrad_x3_kernel<64, 128>), grid, block, 0, stream, a);
      else if (bno == 64)
        hipLaunchKernelGGL((wgrad_x3_kernel<128, 64>), grid, block, 0, stream, a);
      else
        hipLaunchKernelGGL((wgrad_x3_kernel<128, 128>), grid, block, 0, stream, a);
    }
  } else if (WgradDma<T>::launch(a, grid, bmo, bno, stream, zero_page)) {
  } else if (bmo == 64 && bno == 64)
    hipLaunchKernelGGL((wgrad_kernel<T, 64, 64>), grid, block, 0, stream, a);
  else if (bmo == 64)
    hipLaunchKernelGGL((wgrad_kernel<T, 64, 128>), grid, block, 0, stream, a);
  else if (bno == 64)
    hipLaunchKernelGGL((wgrad_kernel<T, 128, 64>), grid, block, 0, stream, a);
  else
    hipLaunchKernelGGL((wgrad_kernel<T, 128, 128>), grid, block, 0, stream, a);
  if (sliced)
    hipLaunchKernelGGL(wgrad_split_reduce_kernel, dim3((unsigned)cdiv((int)((slice + 255) / 256), 1)), dim3(256), 0, stream,
                       (const float*)a.ws, dW_final, slice, splits, slice);
}

}  // namespace mn

#include "wgrad_fused.h"
