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
// Tiling: 256 threads = 4 waves (2x2); block tile BM x BN; each wave owns a (BM/2)x(BN/2)
// sub-tile made of 32x32 MFMA tiles (v_mfma_f32_32x32x16_f16 for half, v_mfma_f32_32x32x2_f32
// for float; fp32 accumulate).  K advances 64 bytes per step per row (4 pieces of 16 B);
// tiles are staged global -> registers -> LDS (double buffered, one barrier per K-step) with
// the next tile's global loads in flight during the MFMAs.  LDS rows are 64 B; the 16-byte
// slot index is XOR-swizzled with (row>>2)&3 so the 16-lane groups of ds_read_b128 hit 16
// distinct slots (MI355X_MICROARCH.md "LDS").
//
// Epilogue options: raw store, per-block column sum / sum-of-squares partials (BatchNorm
// statistics, deterministic two-stage), bias + ReLU, residual add with optional ReLU gate.
#pragma once
#include "common.h"

namespace mn {

struct GatherGeom {
  int B, Hi, Wi, C;  // A tensor [B][Hi][Wi][C]
  int P, Q;          // GEMM rows m = (b*P + p)*Q + q
  int R, S;          // taps; k = (r*S + s)*C + c
  int mul_p, mul_q;  // hn = p*mul_p + rsign*r + off_h ; wn = q*mul_q + ssign*s + off_w
  int rsign, ssign;
  int off_h, off_w;
  int div;  // 1, or 2: tap valid only if hn, wn even; input index = hn/2, wn/2
  int M, N, K;
};

struct Epilogue {
  void* out;             // [M][ldc], element type T
  int ldc;
  float* stats;          // [gridDim.x][2][N] column partial sums (sum, sum of squares) or null
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

__device__ __forceinline__ int lds_swz4(int row) { return (row >> 2) & 3; }

template <typename T, int BM, int BN>
__global__ void __launch_bounds__(256) igemm_kernel(GatherGeom g, const T* __restrict__ A, const T* __restrict__ Bw,
                                                     Epilogue ep) {
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int NP = 4;  // pieces per row per K-step
  constexpr int WM = 2, WN = 2;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int APT = BM * NP / 256, BPT = BN * NP / 256;
  static_assert(APT >= 1 && BPT >= 1 && TM >= 1 && TN >= 1, "tile too small");

  __shared__ piece_t lds[2][(BM + BN) * NP];
  __shared__ float red[WM][BN][2];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int pc = t % NP, lrow = t / NP;  // lrow in [0, 64)

  // ---- per-row gather state -------------------------------------------------------------
  int a_bh[APT], a_bw[APT], a_pix[APT];
#pragma unroll
  for (int i = 0; i < APT; ++i) {
    int m = m0 + lrow + i * 64;
    if (m < g.M) {
      int q = m % g.Q, tmp = m / g.Q;
      int p = tmp % g.P, b = tmp / g.P;
      a_bh[i] = p * g.mul_p + g.off_h;
      a_bw[i] = q * g.mul_q + g.off_w;
      a_pix[i] = b * g.Hi * g.Wi;
    } else {
      a_bh[i] = -(1 << 28);
      a_bw[i] = 0;
      a_pix[i] = 0;
    }
  }
  long b_off[BPT];
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    int n = n0 + lrow + i * 64;
    b_off[i] = n < g.N ? (long)n * g.K : -1;
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
    const int dh = g.rsign * tr, dw = g.ssign * ts;
#pragma unroll
    for (int i = 0; i < APT; ++i) {
      int hn = a_bh[i] + dh, wn = a_bw[i] + dw;
      bool ok = true;
      if (g.div == 2) {
        ok = ((hn | wn) & 1) == 0;
        hn >>= 1;
        wn >>= 1;
      }
      ok = ok && (unsigned)hn < (unsigned)g.Hi && (unsigned)wn < (unsigned)g.Wi;
      ra[i] = ok ? *reinterpret_cast<const piece_t*>(A + ((long)(a_pix[i] + hn * g.Wi + wn) * g.C + cp * VEC))
                 : zero_piece();
    }
#pragma unroll
    for (int i = 0; i < BPT; ++i)
      rb[i] = b_off[i] >= 0 ? *reinterpret_cast<const piece_t*>(Bw + b_off[i] + (long)(kt * NP + pc) * VEC)
                            : zero_piece();
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
      int row = lrow + i * 64;
      lds[buf][row * NP + (pc ^ lds_swz4(row))] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      int row = lrow + i * 64;
      lds[buf][(BM + row) * NP + (pc ^ lds_swz4(row))] = rb[i];
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
        int row = wm * WTM + i * 32 + (lane & 31);
        fa[i].p = lds[cur][row * NP + (piece ^ lds_swz4(row))];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        int row = wn * WTN + j * 32 + (lane & 31);
        fb[j].p = lds[cur][(BM + row) * NP + (piece ^ lds_swz4(row))];
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

  // ---- epilogue ---------------------------------------------------------------------------
  T* out = reinterpret_cast<T*>(ep.out);
  const T* res = reinterpret_cast<const T*>(ep.res);
  const T* gate = reinterpret_cast<const T*>(ep.res_gate);
  float s1[TN], s2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * WTN + j * 32 + (lane & 31);
      const float bias = (ep.bias && col < g.N) ? ep.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float v = acc[i][j][r] * ep.alpha + bias;
        if (ep.relu) v = fmaxf(v, 0.f);
        s1[j] += v;
        s2[j] += v * v;
        if (row < g.M && col < g.N) {
          const long idx = (long)row * ep.ldc + col;
          if (res) {
            float rv = (float)res[idx];
            if (gate && !((float)gate[idx] > 0.f)) rv = 0.f;
            v += rv;
          }
          out[idx] = (T)v;
        }
      }
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
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        a += red[w][t][0];
        b += red[w][t][1];
      }
      ep.stats[((long)blockIdx.x * 2 + 0) * g.N + n0 + t] = a;
      ep.stats[((long)blockIdx.x * 2 + 1) * g.N + n0 + t] = b;
    }
  }
}

// number of M-blocks the launcher will use (needed to size the stats partial buffer)
inline int igemm_grid_m(int M) { return cdiv(M, 128); }

template <typename T>
inline void launch_igemm(const GatherGeom& g, const T* A, const T* Bw, const Epilogue& ep, hipStream_t stream) {
  dim3 block(256);
  if (g.N <= 64) {
    dim3 grid(cdiv(g.M, 128), cdiv(g.N, 64));
    hipLaunchKernelGGL((igemm_kernel<T, 128, 64>), grid, block, 0, stream, g, A, Bw, ep);
  } else {
    dim3 grid(cdiv(g.M, 128), cdiv(g.N, 128));
    hipLaunchKernelGGL((igemm_kernel<T, 128, 128>), grid, block, 0, stream, g, A, Bw, ep);
  }
}

}  // namespace mn
