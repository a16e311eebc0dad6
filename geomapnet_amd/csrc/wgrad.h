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
};

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
      rx[i] = ok ? *reinterpret_cast<const piece_t*>(X + ((long)((xb[i] * g.Hi + hn) * g.Wi + wn) * g.C + c0))
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
        if (n < g.N && dst >= 0) unsafeAtomicAdd(a.dW + (long)n * a.ldw + dst, acc[i][j][r] * a.alpha);
      }
    }
}

template <typename T>
inline void launch_wgrad(WgradArgs a, int target_blocks, hipStream_t stream) {
  const GatherGeom& g = a.g;
  const bool narrow_n = g.N <= 64, narrow_k = g.K <= 64 || (g.K % 128 != 0 && g.K < 256);
  int bmo = narrow_n ? 64 : 128, bno = narrow_k ? 64 : 128;
  int tiles = cdiv(g.N, bmo) * cdiv(g.K, bno);
  int splits = cdiv(target_blocks, tiles);
  int max_splits = cdiv(g.M, 256);  // at least 8 steps per block
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int rows = cdiv(cdiv(g.M, splits), 32) * 32;
  splits = cdiv(g.M, rows);
  a.rows_per_split = rows;
  dim3 grid(cdiv(g.N, bmo), cdiv(g.K, bno), splits), block(256);
  if (bmo == 64 && bno == 64)
    hipLaunchKernelGGL((wgrad_kernel<T, 64, 64>), grid, block, 0, stream, a);
  else if (bmo == 64)
    hipLaunchKernelGGL((wgrad_kernel<T, 64, 128>), grid, block, 0, stream, a);
  else if (bno == 64)
    hipLaunchKernelGGL((wgrad_kernel<T, 128, 64>), grid, block, 0, stream, a);
  else
    hipLaunchKernelGGL((wgrad_kernel<T, 128, 128>), grid, block, 0, stream, a);
}

}  // namespace mn
