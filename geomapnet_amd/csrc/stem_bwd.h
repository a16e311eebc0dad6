// Stem backward in two launches instead of four (fp16): what autograd computes for conv1.weight below
// maxpool -> relu -> bn1 -> conv1 (torchvision ResNet stem; loss.backward() at /root/reference/common/train.py:355).
//
// Round 1 ran: maxpool_bwd (gathers the pooled gradient into a full-size tensor: 0.54 GB written), bn_bwd_reduce and
// bn_bwd_apply over that tensor and the conv output (2.7 GB of traffic, 0.54 GB written: d(conv output)), and the plain
// weight-gradient GEMM over d(conv output) and an im2col of the input -- 1.1 ms per step for a gradient of 9 408 numbers.
// Nothing below the stem consumes d(conv output) (the input needs no gradient), so it never has to exist in memory:
//   1. bn_bwd_reduce with the max-pool gradient gathered on the fly (elementwise.h, PoolGradSrc) + finalize: the two
//      BatchNorm sums, d(gamma), d(beta), and the per-channel coefficients of the apply step;
//   2. stem_wgrad_kernel (here): per 8 x 32-pixel tile, d(conv output) is COMPUTED into LDS -- gather of the pooled
//      gradient (argmax bytes + pooled gradient, pool.h), ReLU gate recomputed from the conv output, BatchNorm backward
//      with the coefficients of step 1, rounded to fp16 exactly as the stored tensor was -- and contracted with the input
//      straight from an LDS-resident image of the zero-padded NHWC4 input (as stem.h's forward: a B fragment of the
//      weight-gradient GEMM is a transpose read over consecutive pixel PAIRS, no im2col).
// Traffic: conv output 0.54 GB + argmax 0.07 GB + pooled gradient 0.13 GB (each twice, steps 1 and 2) + input 0.14 GB.
//
// Tile = 8 x 32 output pixels; the GEMM is dW[n][k] += sum_m gy[m][n] * X[m][k] with m = pixel, k = (r, s4, e) the 7 x 4
// pixel pairs x 8 halves of stem.h (224 columns, mapped to the dense 147 by `colmap`).  Wave w owns kernel rows
// r = 2w, 2w + 1 (the last wave: r = 6 only), both 32-column tiles of the 64 output channels: 4 accumulator tiles.
// Workgroups are persistent; accumulators leave through fp32 atomics once per workgroup.
#pragma once
#include "pool.h"
#include "stem.h"
#include "wgrad.h"

namespace mn {

struct StemWgradArgs {
  const half* y;             // [B][H0][W0][64] raw conv output
  const unsigned char* idx;  // [B][Po][Qo][64] winning tap of every max-pool window
  const half* gp;            // [B][Po][Qo][64] gradient w.r.t. the pooled activation
  const float* coef;         // [4][64]: k1 = gamma*invstd, mean of gm, mean of gm*xhat, shift of the self gate (finalize kernel)
  const float* mean;         // [64]
  const float* invstd;       // [64]
  const half* xpad;          // [B][Hp][Wp][4] zero-padded input
  float* dW;                 // [64][ldw] fp32, accumulated atomically
  const int* colmap;         // [224] compute column -> dense column or -1
  int ldw;
  float alpha;               // 1 / loss scale
  int B, Hp, Wp2, H0, W0, Po, Qo, tiles_x, tiles_y;
};

static __global__ void __launch_bounds__(256, 2) stem_wgrad_kernel(StemWgradArgs a) {
  constexpr int TH = kStemTH, TW = kStemTW;
  constexpr int IH = 2 * TH + 5, IW = TW + 3, IPIECES = IH * IW;  // 21 rows x 35 pairs
  constexpr int IPASS = (IPIECES + 255) / 256;
  constexpr int GYP = TH * TW * 8;  // pieces of the d(conv output) tile: 256 pixels x 64 halves
  // ONE LDS object: [2 input images][d(conv output) tile]
  __shared__ piece_t smem[2 * IPASS * 256 + GYP];
  half* gyt = reinterpret_cast<half*>(&smem[2 * IPASS * 256]);

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ntiles = a.B * a.tiles_x * a.tiles_y;
  const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(a.xpad, (long)a.B * a.Hp * a.Wp2 * 16L);
  auto issue_image = [&](int tile, int buf) {
    const int txi = tile % a.tiles_x;
    const int tmp = tile / a.tiles_x;
    const int tyi = tmp % a.tiles_y, b = tmp / a.tiles_y;
    const int y0 = tyi * TH, x0 = txi * TW;
#pragma unroll
    for (int i = 0; i < IPASS; ++i) {
      const int q = t + i * 256;
      const int row = q / IW, col = q - row * IW;
      const int iy = 2 * y0 + row, ip = x0 + col;
      const bool ok = q < IPIECES && iy < a.Hp && ip < a.Wp2;
      const unsigned off = ok ? (unsigned)(((b * a.Hp + iy) * a.Wp2 + ip) * 16) : ~0u;
      dma16(rsrc, off, 0u, &smem[buf * IPASS * 256 + wave * 64 + i * 256]);
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) issue_image(tile, 0);

  // elementwise role: this thread always handles channel piece t & 7 -> its 8 channels' coefficients live in registers
  const int cp = t & 7;
  float k1[8], mg[8], mgx[8], sh[8], mu[8], is[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = cp * 8 + e;
    k1[e] = a.coef[c];
    mg[e] = a.coef[64 + c];
    mgx[e] = a.coef[128 + c];
    sh[e] = a.coef[192 + c];
    mu[e] = a.mean[c];
    is[e] = a.invstd[c];
  }

  // GEMM role: transpose-read lane geometry (wgrad.h)
  const int gq = lane >> 4, i16 = lane & 15;
  const int src_row = i16 >> 2, chunk = i16 & 3;
  const int kgrp = (gq >> 1) * 8;
  // A operand (d(conv output) tile, rows = pixels of 128 bytes, 16-byte slots XOR-swizzled with (row & 3) << 1):
  // column tile j, columns j*32 + (gq & 1)*16 + chunk*4 ..
  unsigned aA[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = j * 32 + (gq & 1) * 16 + chunk * 4;
    aA[j] = (unsigned)(2 * IPASS * 256 * 16) +
            (unsigned)(((kgrp + src_row) * 64 + (((col >> 3) ^ wg_swz<8>(src_row)) * 8) + (col & 7)) * 2);
  }
  // B operand (input image, rows = consecutive pixel pairs of 16 bytes): source lane (row e = src_row, chunk) of column
  // block (gq & 1) addresses pair (pixel + e) + 2 (gq & 1) + (chunk >> 1), half (chunk & 1)
  const unsigned aB = (unsigned)(((kgrp + src_row) + 2 * (gq & 1) + (chunk >> 1)) * 16 + (chunk & 1) * 8);
  const int r0 = 2 * wave, nr = wave < 3 ? 2 : 1;

  floatx16 acc[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][ri][r] = 0.f;

  for (int it = 0; tile < ntiles; tile += gridDim.x, ++it) {
    const int txi = tile % a.tiles_x;
    const int tmp = tile / a.tiles_x;
    const int tyi = tmp % a.tiles_y, b = tmp / a.tiles_y;
    const int y0 = tyi * TH, x0 = txi * TW;
    __syncthreads();  // everyone is done with the previous tile's LDS reads (gradient tile, other image)
    if (tile + (int)gridDim.x < ntiles) issue_image(tile + gridDim.x, (it + 1) & 1);
    // ---- d(conv output) of the tile -> LDS: 8 pieces (pixel, 8 channels) per thread, in four groups of two whose loads
    // (conv output + the four candidate windows of the pool gradient, unconditional at clamped coordinates) are all
    // requested before the first use
#pragma unroll
    for (int grp4 = 0; grp4 < 4; ++grp4) {
      PieceView<half> vy[2];
      float g[2][8];
      bool ok[2];
      int pxs[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int px = (t >> 3) + 32 * (grp4 * 2 + u);  // tile pixel: row px >> 5, column px & 31
        const int oy = y0 + (px >> 5), ox = x0 + (px & 31);
        ok[u] = oy < a.H0 && ox < a.W0;
        pxs[u] = px;
        const int oyc = oy < a.H0 ? oy : a.H0 - 1, oxc = ox < a.W0 ? ox : a.W0 - 1;
        vy[u].p = *reinterpret_cast<const piece_t*>(a.y + (((long)b * a.H0 + oyc) * a.W0 + oxc) * 64 + cp * 8);
        pool_grad_piece<half>(a.idx, a.gp, b, oyc, oxc, cp, 8, a.Po, a.Qo, g[u]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        PieceView<half> o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float yv = (float)vy[u].e[e];
          float gv = g[u][e];
          if (!(yv * k1[e] + sh[e] > 0.f)) gv = 0.f;  // ReLU gate recomputed from the conv output (the forward's arithmetic)
          const float xh = (yv - mu[e]) * is[e];
          o.e[e] = ok[u] ? (half)(k1[e] * (gv - mg[e] - xh * mgx[e])) : (half)0.f;
        }
        *reinterpret_cast<piece_t*>(gyt + pxs[u] * 64 + ((cp ^ wg_swz<8>(pxs[u])) * 8)) = o.p;
      }
    }
    wait_vmcnt<0>();   // this tile's image (requested one tile ago, or in the prologue)
    __syncthreads();   // gradient tile and image complete for every wave
    // ---- GEMM: 16 K-steps of 16 pixels (tile row ty, half hx)
    const unsigned img = (unsigned)((it & 1) * IPASS * 256 * 16);
#pragma unroll 2
    for (int ks = 0; ks < 2 * TH; ++ks) {
      const int ty = ks >> 1, hx = ks & 1;
      const unsigned pa = (unsigned)((ty * 32 + hx * 16) * 128);
      TrFrag fa[2], fb[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        fa[j].h[0] = ds_read_tr16(reinterpret_cast<const char*>(smem) + aA[j] + pa);
        fa[j].h[1] = ds_read_tr16(reinterpret_cast<const char*>(smem) + aA[j] + pa + 4 * 128);
      }
#pragma unroll
      for (int ri = 0; ri < 2; ++ri) {
        if (ri < nr) {
          const unsigned pb = img + (unsigned)(((2 * ty + r0 + ri) * IW + hx * 16) * 16);
          fb[ri].h[0] = ds_read_tr16(reinterpret_cast<const char*>(smem) + aB + pb);
          fb[ri].h[1] = ds_read_tr16(reinterpret_cast<const char*>(smem) + aB + pb + 4 * 16);
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ri = 0; ri < 2; ++ri)
          if (ri < nr) acc[j][ri] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[j].v, fb[ri].v, acc[j][ri], 0, 0, 0);
    }
  }

  // ---- accumulators -> dW[n][colmap(k)]
#pragma unroll
  for (int ri = 0; ri < 2; ++ri) {
    if (ri >= nr) continue;
    const int k = (r0 + ri) * 32 + (lane & 31);
    const int dst = a.colmap ? a.colmap[k] : k;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (dst >= 0) unsafeAtomicAdd(a.dW + (long)n * a.ldw + dst, acc[j][ri][r] * a.alpha);
      }
  }
}

inline void launch_stem_wgrad(StemWgradArgs a, int B, int H, int W, int Wp, hipStream_t stream) {
  a.B = B; a.Hp = H + 6; a.Wp2 = Wp / 2;
  a.H0 = (H - 1) / 2 + 1; a.W0 = (W - 1) / 2 + 1;
  a.Po = (a.H0 + 2 - 3) / 2 + 1; a.Qo = (a.W0 + 2 - 3) / 2 + 1;
  a.tiles_x = cdiv(a.W0, kStemTW); a.tiles_y = cdiv(a.H0, kStemTH);
  static const int wgs = getenv("MN_STEM_WGS") ? atoi(getenv("MN_STEM_WGS")) : 512;
  const int ntiles = B * a.tiles_x * a.tiles_y;
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3(ntiles < wgs ? ntiles : wgs), dim3(256), 0, stream, a);
}

}  // namespace mn
