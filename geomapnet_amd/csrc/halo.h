// 3x3 stride-1 convolution of 64-channel fp16 tensors (ResNet layer1, forward and data gradient) with the input tile
// staged ONCE in LDS.
//
// Why: the implicit-GEMM kernel (igemm.h) fetches every tap of every K-step through the LDS-DMA path -- for C = N = 64
// that is 221 KB per 128-pixel tile (nine shifted copies of the same 128 input rows + the 74 KB weight matrix), and the
// launch runs at the ~20 B/clk/CU that path sustains (159 us for 270 MB of tensors, MFMA pipe 21 % busy).  Here a
// workgroup owns a 16x16-pixel output tile, loads its 18x18-pixel input halo (41 KB) once, and walks the nine taps as
// ADDRESS OFFSETS into that LDS image; only the 8 KB weight slice of a tap is streamed per K-step.
// 115 KB of DMA per 256 pixels instead of 442 KB.
//
//   out[b, y, x, n] = sum_{r, s, c} in[b, y + rsign*r + off_h, x + ssign*s + off_w, c] * Bw[n][(r*3 + s)*64 + c]
//
// (GatherGeom conventions of igemm.h: forward rsign = +1, off = -pad; data gradient rsign = -1, off = +pad.)
// 4 waves, wave tile 64 pixels (4 tile rows) x 64 channels = 2x2 MFMA tiles of 32x32x16; weight slices in a two-buffer
// ring; LDS 62 KB -> two workgroups per CU.  LDS image: pixel-major, 8 pieces of 16 bytes per pixel, piece slot XOR-swizzled by (pixel >> 1) & 7 on the
// source side of the DMA and in the fragment address (rows of a ds_read_b128 lane group are consecutive pixels of a
// halo row: conflict-free within a row, 2-way on two banks where a lane group spans two rows).
// (Measured and not kept: a 4-deep weight ring, 141 vs 138 us; 32-channel sub-steps -- fragment reads 256 instead of 128
// matrix-pipe cycles ahead of their use -- 130 vs 126 us.)
// Epilogue as igemm's (alpha, BatchNorm column sums, residual / gates, 16-byte stores), with tile rows mapped back to
// pixels and out-of-image pixels of ragged tiles (341 / 4 = 86 = 5*16 + 6 columns) masked out of the statistics.
#pragma once
#include "igemm.h"

namespace mn {

constexpr int kHaloTH = 16, kHaloTW = 16;

// ABL (timing experiments only, MN_HALO_ABLATE, results are wrong): bit 0 = no halo DMA, bit 1 = no weight DMA (and no
// per-tap wait / barrier), bit 2 = no epilogue stores and residual / gate loads, bit 3 = no MFMA.
template <int ABL = 0>
static __global__ void __launch_bounds__(256, 2) conv_halo_kernel(GatherGeom g, const half* __restrict__ A,
                                                                  const half* __restrict__ Bw, Epilogue ep, int tiles_x,
                                                                  int tiles_y, int grid_n) {
  constexpr int VEC = 8, NP = 8, NT = 256;
  constexpr int TH = kHaloTH, TW = kHaloTW, BM = TH * TW, BN = 64, WM = 4;
  constexpr int HW = TW + 2, HH = TH + 2, HPIX = HW * HH;          // 18 x 18 halo
  constexpr int HPASS = (HPIX * NP + NT - 1) / NT;                  // 11 DMA passes
  constexpr int HAL = HPASS * NT;                                   // pieces allocated for the halo image
  constexpr int BT = BN * NP;                                       // pieces of one weight slice (64 rows x 128 B)
  constexpr int NB = 2;  // weight-slice ring (a 4-deep ring, three taps in flight, measured the same: 141 vs 138 us)
  __shared__ piece_t smem[HAL + NB * BT + WM * BN / 2];
  float* red = reinterpret_cast<float*>(&smem[HAL + NB * BT]);  // [WM][BN][2]

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = tile % grid_n;
  int tmp = tile / grid_n;
  const int txi = tmp % tiles_x;
  tmp /= tiles_x;
  const int tyi = tmp % tiles_y;
  const int b = tmp / tiles_y;
  const int y0 = tyi * TH, x0 = txi * TW, n0 = tile_n * BN;
  const int R = g.R, S = g.S;
  // halo origin: the smallest input coordinate any tap of the tile touches
  const int oy = y0 + (g.rsign > 0 ? g.off_h : g.off_h - (R - 1));
  const int ox = x0 + (g.ssign > 0 ? g.off_w : g.off_w - (S - 1));

  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(A, (long)g.B * g.Hi * g.Wi * g.C * 2L);
  const __amdgpu_buffer_rsrc_t rsrc_b = make_rsrc(Bw, (long)g.N * g.K * 2L);

  // ---- input halo: one DMA sweep ------------------------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < HPASS; ++i) {
    if (i * NT + wave * 64 >= HPIX * NP) continue;  // wave-uniform: the last pass only has work for the first waves
    if constexpr ((ABL & 1) != 0) continue;
    const int q = t + i * NT;
    const int hp = q >> 3, pc = q & 7;
    const int hy = hp / HW, hx = hp - hy * HW;
    const int iy = oy + hy, ix = ox + hx;
    const bool ok = hp < HPIX && (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
    const unsigned off = ok ? (unsigned)((((b * g.Hi + iy) * g.Wi + ix) * g.C) * 2) + (unsigned)((pc ^ ((hp >> 1) & 7)) * 16) : ~0u;
    dma16(rsrc_a, off, 0u, &smem[wave * 64 + i * NT]);
  }
  // ---- weight slices: 64 rows x 128 bytes per tap, two-buffer ring ---------------------------------------------
  unsigned b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = t + i * NT, br = q >> 3, pc = q & 7;
    const int n = n0 + br;
    b_off[i] = n < g.N ? (unsigned)(n * g.K) * 2u + (unsigned)((pc ^ ((br >> 1) & 7)) * 16) : ~0u;
  }
  auto issue_b = [&](int tap, int buf) {
    if constexpr ((ABL & 2) != 0) return;
#pragma unroll
    for (int i = 0; i < 2; ++i) dma16(rsrc_b, b_off[i], (unsigned)(tap * g.C * 2), &smem[HAL + buf * BT + wave * 64 + i * NT]);
  };
#pragma unroll
  for (int j = 0; j < NB - 1; ++j)
    if (j < R * S) issue_b(j, j);

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // pixel of this lane inside the wave's 4 x 16 pixel block, per MFMA row tile i: rows 2i + (l >> 4), column l & 15
  const int l31 = lane & 31;
  const int prow = 4 * wm + (l31 >> 4), pcol = l31 & 15;
  const int ntaps = R * S;
  for (int tap = 0; tap < ntaps; ++tap) {
    // slices issued so far: min(ntaps, tap + NB - 1), two DMA instructions each, in tap order after the halo's
    const int ahead = min(ntaps, tap + NB - 1) - (tap + 1);
    if (ahead >= 2)
      wait_vmcnt<4>();
    else if (ahead == 1)
      wait_vmcnt<2>();
    else
      wait_vmcnt<0>();
    if ((ABL & 2) == 0 || tap == 0)
      __builtin_amdgcn_s_barrier();  // halo (first step) and this tap's weights visible; slot (tap - 1) % NB is free
    if (tap + NB - 1 < ntaps) issue_b(tap + NB - 1, (tap + NB - 1) % NB);
    const int r = tap / S, s = tap - r * S;
    const int dy = g.rsign > 0 ? r : R - 1 - r, dx = g.ssign > 0 ? s : S - 1 - s;
    const piece_t* tb = &smem[HAL + (tap % NB) * BT];
    const int hp0 = (prow + dy) * HW + pcol + dx;  // halo pixel of row tile 0; row tile 1 is two halo rows below
    PieceView<half> fa[2][2], fb[2][2];
    auto load_frags = [&](int ks, int slot) {
      const int piece = ks * 2 + (lane >> 5);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int hp = hp0 + i * 2 * HW;
        fa[slot][i].p = smem[hp * NP + (piece ^ ((hp >> 1) & 7))];
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = j * 32 + l31;
        fb[slot][j].p = tb[row * NP + (piece ^ ((row >> 1) & 7))];
      }
    };
    load_frags(0, 0);
#pragma unroll
    for (int ks = 0; ks < NP / 2; ++ks) {
      if (ks + 1 < NP / 2) load_frags(ks + 1, (ks + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr ((ABL & 8) == 0) {
            mma_piece<half>(fa[ks & 1][i], fb[ks & 1][j], acc[i][j]);
          } else {
            asm volatile("" ::"v"(fa[ks & 1][i].p), "v"(fb[ks & 1][j].p));
          }
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();  // all fragment reads done before the halo image is reused as epilogue staging

  // ---- epilogue -----------------------------------------------------------------------------------------------
  half* out = reinterpret_cast<half*>(ep.out);
  const half* res = reinterpret_cast<const half*>(ep.res);
  const half* gate = reinterpret_cast<const half*>(ep.res_gate);
  const half* ogate = reinterpret_cast<const half*>(ep.out_gate);
  float* stage = reinterpret_cast<float*>(&smem[0]);  // [128][64] fp32: MFMA row tile i of all four waves (32 KB)
  constexpr int SC = BN, CPR = SC / VEC, PASSES = 128 * CPR / NT;  // 4 output pieces per thread and round
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
  const bool want_stats = ep.stats != nullptr || ep.stats_accum != nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    // the residual / gate pieces of this round are requested first: their latency runs under the staging below
    PieceView<half> rv[PASSES], gv[PASSES], ov[PASSES];
    long idxs[PASSES];
    bool oks[PASSES];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int id = t + ps * NT;
      const int lr = id / CPR, cpc = id % CPR;
      const int rl = (lr >> 5) * 64 + i * 32 + (lr & 31);
      const int y = y0 + (rl >> 4), x = x0 + (rl & 15);
      const int col = n0 + cpc * VEC;
      oks[ps] = y < g.P && x < g.Q && col < g.N && (ABL & 4) == 0;
      idxs[ps] = (((long)b * g.P + y) * g.Q + x) * ep.ldc + col;
      if (oks[ps]) {
        if (res) rv[ps].p = *reinterpret_cast<const piece_t*>(res + idxs[ps]);
        if (res && gate) gv[ps].p = *reinterpret_cast<const piece_t*>(gate + idxs[ps]);
        if (ogate) ov[ps].p = *reinterpret_cast<const piece_t*>(ogate + idxs[ps]);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int lc = j * 32 + l31;
      const float bias = (ep.bias && n0 + lc < g.N) ? ep.bias[n0 + lc] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[i][j][r] * ep.alpha + bias;
        if (ep.relu & 1) v = fmaxf(v, 0.f);
        const int rt = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);  // row inside the 32-row MFMA tile
        if (want_stats) {
          const int rl = wm * 64 + i * 32 + rt;  // tile-local row = pixel (rl >> 4, rl & 15)
          const bool ok = y0 + (rl >> 4) < g.P && x0 + (rl & 15) < g.Q;
          const float vs = ok ? v : 0.f;
          s1[j] += vs;
          s2[j] += vs * vs;
        }
        stage[(wm * 32 + rt) * SC + lc] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int id = t + ps * NT;
      const int lr = id / CPR, cpc = id % CPR;
      if (oks[ps]) {
        float v[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {
          floatx4 f = *reinterpret_cast<const floatx4*>(&stage[lr * SC + cpc * VEC + e]);
          v[e] = f[0];
          v[e + 1] = f[1];
          v[e + 2] = f[2];
          v[e + 3] = f[3];
        }
        if (res) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            float xr = (float)rv[ps].e[e];
            if (gate && !((float)gv[ps].e[e] > 0.f)) xr = 0.f;
            v[e] += xr;
          }
        }
        if (ogate) {
#pragma unroll
          for (int e = 0; e < VEC; ++e)
            if (!((float)ov[ps].e[e] > 0.f)) v[e] = 0.f;
        }
        PieceView<half> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.e[e] = (half)v[e];
        *reinterpret_cast<piece_t*>(out + idxs[ps]) = o.p;
      }
    }
    __syncthreads();
  }
  if (want_stats) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      s1[j] += __shfl_xor(s1[j], 32);
      s2[j] += __shfl_xor(s2[j], 32);
      if (lane < 32) {
        const int lc = j * 32 + lane;
        red[(wm * BN + lc) * 2 + 0] = s1[j];
        red[(wm * BN + lc) * 2 + 1] = s2[j];
      }
    }
    __syncthreads();
    const int tile_m = tile / grid_n;
    for (int c = t; c < BN; c += NT)
      if (n0 + c < g.N) {
        float a = 0.f, bsum = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          a += red[(w * BN + c) * 2 + 0];
          bsum += red[(w * BN + c) * 2 + 1];
        }
        if (ep.stats_accum) {
          double* row = ep.stats_accum + (long)(tile_m % ep.stats_rows) * 2 * g.N;
          atomicAdd(row + n0 + c, (double)a);
          atomicAdd(row + g.N + n0 + c, (double)bsum);
        } else {
          ep.stats[((long)tile_m * 2 + 0) * g.N + n0 + c] = a;
          ep.stats[((long)tile_m * 2 + 1) * g.N + n0 + c] = bsum;
        }
      }
  }
}

// number of M-blocks the halo kernel uses (rows of a [grid_m][2][N] statistics buffer)
inline int conv_halo_grid_m(const GatherGeom& g) { return g.B * cdiv(g.P, kHaloTH) * cdiv(g.Q, kHaloTW); }

// the shapes the kernel covers: fp16, 3x3, stride 1, 64 input channels, same-size output
inline bool conv_halo_applies(const GatherGeom& g) {
  return g.R == 3 && g.S == 3 && g.C == 64 && g.mul_p == 1 && g.mul_q == 1 && g.div == 1 && g.P == g.Hi && g.Q == g.Wi &&
         g.K == 9 * 64 && g.N % 8 == 0 && !g.bt_on && (g.rsign == 1 || g.rsign == -1) && g.rsign == g.ssign &&
         (long)g.B * g.Hi * g.Wi * g.C * 2 < 0xfffffff0l;
}

// returns the number of M-blocks used
inline int launch_conv_halo(const GatherGeom& g, const half* A, const half* Bw, const Epilogue& ep, hipStream_t stream) {
  const int tx = cdiv(g.Q, kHaloTW), ty = cdiv(g.P, kHaloTH), gn = cdiv(g.N, 64);
#ifdef MN_ABLATION_BUILD
  static const int abl = getenv("MN_HALO_ABLATE") ? atoi(getenv("MN_HALO_ABLATE")) : 0;
  const dim3 grid(g.B * tx * ty * gn);
  switch (abl) {
    case 1: hipLaunchKernelGGL(conv_halo_kernel<1>, grid, dim3(256), 0, stream, g, A, Bw, ep, tx, ty, gn); return g.B * tx * ty;
    case 2: hipLaunchKernelGGL(conv_halo_kernel<2>, grid, dim3(256), 0, stream, g, A, Bw, ep, tx, ty, gn); return g.B * tx * ty;
    case 3: hipLaunchKernelGGL(conv_halo_kernel<3>, grid, dim3(256), 0, stream, g, A, Bw, ep, tx, ty, gn); return g.B * tx * ty;
    case 4: hipLaunchKernelGGL(conv_halo_kernel<4>, grid, dim3(256), 0, stream, g, A, Bw, ep, tx, ty, gn); return g.B * tx * ty;
    case 7: hipLaunchKernelGGL(conv_halo_kernel<7>, grid, dim3(256), 0, stream, g, A, Bw, ep, tx, ty, gn); return g.B * tx * ty;
    case 8: hipLaunchKernelGGL(conv_halo_kernel<8>, grid, dim3(256), 0, stream, g, A, Bw, ep, tx, ty, gn); return g.B * tx * ty;
    case 12: hipLaunchKernelGGL(conv_halo_kernel<12>, grid, dim3(256), 0, stream, g, A, Bw, ep, tx, ty, gn); return g.B * tx * ty;
    case 15: hipLaunchKernelGGL(conv_halo_kernel<15>, grid, dim3(256), 0, stream, g, A, Bw, ep, tx, ty, gn); return g.B * tx * ty;
    default: break;
  }
#endif
  hipLaunchKernelGGL(conv_halo_kernel<0>, dim3(g.B * tx * ty * gn), dim3(256), 0, stream, g, A, Bw, ep, tx, ty, gn);
  return g.B * tx * ty;
}

}  // namespace mn
