// Stem backward in two tile-walking launches (fp16): what autograd computes for conv1.weight, bn1.weight and bn1.bias below
// maxpool -> relu -> bn1 -> conv1 (torchvision ResNet stem; loss.backward() at /root/reference/common/train.py:355).
//
// Round 1 ran: maxpool_bwd (gathers the pooled gradient into a full-size tensor: 0.54 GB written), bn_bwd_reduce and
// bn_bwd_apply over that tensor and the conv output (2.7 GB of traffic, 0.54 GB written: d(conv output)), and the plain
// weight-gradient GEMM over d(conv output) and an im2col of the input -- 1.1 ms per step for 9 536 numbers.
// Nothing below the stem consumes d(conv output) (the input needs no gradient), so neither it nor the max-pool's input
// gradient has to exist in memory.  Both kernels here walk 4 x 32-pixel tiles with persistent workgroups and stage,
// per tile, the 3 x 17 pooled windows the tile's pixels belong to (pooled gradient 128 B + argmax bytes 64 B per window)
// in LDS by DMA, double-buffered: the gradient of a conv-output pixel is then gathered from LDS (a first version that
// gathered from global memory issued 4 x 24 B of requests per 16 bytes of conv output and ran at the L2 request rate:
// 450 + 570 us, no better than the four launches).
//   1. stem_bn_reduce_kernel: sum(gm), sum(gm * xhat) per channel, gm = gathered gradient gated by the ReLU recomputed
//      from the conv output (since round 3 in the order of the pooled windows, see the kernel); then bn_finalize_bwd_kernel
//      (elementwise.h): d(gamma), d(beta), coefficients of the apply step;
//   2. stem_wgrad_kernel: per tile, d(conv output) = k1 (gm - mean(gm) - xhat mean(gm xhat)) is computed into LDS, rounded
//      to fp16 exactly as the stored tensor was, and contracted with the input read from an LDS-resident image of the
//      zero-padded NHWC4 input (as stem.h's forward: a B fragment of the weight-gradient GEMM is a transpose read over
//      consecutive pixel PAIRS, no im2col).  The GEMM is dW[n][k] += sum_m gy[m][n] X[m][k], m = pixel, k = (r, s4, e) the
//      7 x 4 pixel pairs x 8 halves of stem.h (224 columns, mapped to the dense 147 by `colmap`); wave w owns kernel
//      rows r = 2w, 2w + 1 (the last wave: r = 6) x both 32-column tiles of the 64 output channels.  Accumulators leave
//      through fp32 atomics once per workgroup.
// Traffic: conv output 0.54 GB + argmax 0.07 GB + pooled gradient 0.13 GB, each twice, + input 0.14 GB.
#pragma once
#include "pool.h"
#include "stem.h"
#include "wgrad.h"

namespace mn {

struct StemBwdArgs {
  const half* y;             // [B][H0][W0][64] raw conv output
  const unsigned char* idx;  // [B][Po][Qo][64] winning tap of every max-pool window
  const half* gp;            // [B][Po][Qo][64] gradient w.r.t. the pooled activation
  const float* gamma;        // reduce: [64]
  const float* beta;         // reduce: [64]
  const float* coef;         // wgrad: [4][64] k1 = gamma*invstd, mean of gm, mean of gm*xhat, shift of the self gate
  const float* mean;         // [64]
  const float* invstd;       // [64]
  double* accum;             // reduce: [accum_rows][2][64] fp64 sums, added to atomically
  int accum_rows;
  const half* xpad;          // wgrad: [B][Hp][Wp][4] zero-padded input
  float* dW;                 // wgrad: [64][ldw] fp32, accumulated atomically
  const int* colmap;         // wgrad: [224] compute column -> dense column or -1
  int ldw;
  float alpha;               // wgrad: 1 / loss scale
  // gp arrives ALREADY multiplied by the stem's ReLU gate (fp16x2m: the producer of gp -- layer1.0's data gradient -- gates it with
  // the pooled activation, gate(window) = pooled value > 0 = the activation at the window's argmax > 0, which is exact; y is then an
  // fp16 COPY of the fp32 conv output, whose recomputed sign would flip for values within 2^-12 of the threshold): do not recompute
  int pre_gated = 0;
  int B, Hp, Wp2, H0, W0, Po, Qo, tiles_x, tiles_y;
};

constexpr int kSbTH = 4, kSbTW = 32;                         // tile of conv-output pixels
constexpr int kSbPR = kSbTH / 2 + 1, kSbPC = kSbTW / 2 + 1;  // 3 x 17 pooled windows touch it
constexpr int kSbGP = 448;    // pieces reserved for the pooled gradient (3*17*8 = 408, rounded up to whole waves)
constexpr int kSbPOOL = 768;  // pieces of one pool buffer: gradient, then argmax bytes (3*17*4 = 204)

// DMA of the pooled windows of a tile into `dst` (kSbPOOL pieces): window (pr, pc) = pooled position (y0/2 + pr, x0/2 + pc)
__device__ __forceinline__ void sb_issue_pool(const StemBwdArgs& a, __amdgpu_buffer_rsrc_t rsrc_g, __amdgpu_buffer_rsrc_t rsrc_i,
                                              int b, int y0, int x0, piece_t* dst, int t, int wave) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int q = t + i * 256;
    const bool is_g = (wave * 64 + i * 256) < kSbGP;  // wave-uniform: 448 = 7 waves
    const int qq = is_g ? q : q - kSbGP;
    const int win = is_g ? qq >> 3 : qq >> 2, part = is_g ? (qq & 7) : (qq & 3);
    const int pr = win / kSbPC, pc = win - pr * kSbPC;
    const int po = (y0 >> 1) + pr, qo = (x0 >> 1) + pc;
    const bool ok = win < kSbPR * kSbPC && po < a.Po && qo < a.Qo;
    const unsigned wo = (unsigned)(((b * a.Po + po) * a.Qo + qo) * 64);  // element offset of the window
    if (is_g)
      dma16(rsrc_g, ok ? wo * 2u + (unsigned)part * 16u : ~0u, 0u, dst + wave * 64 + i * 256);
    else
      dma16(rsrc_i, ok ? wo + (unsigned)part * 16u : ~0u, 0u, dst + wave * 64 + i * 256);
  }
}

// gradient of conv-output pixel (h, w) for channel piece cp, gathered from the staged windows: the sum over the (at most
// four) windows the pixel belongs to of the window's gradient where its recorded argmax is this pixel; rounded to fp16
// like the stored tensor (pool.h, pool_grad_piece)
__device__ __forceinline__ void sb_pool_grad(const piece_t* pool, int h, int w, int y0, int x0, int cp, int Po, int Qo,
                                             float (&acc)[8]) {
  const int po0 = h >> 1, po1 = (h + 1) >> 1, qo0 = w >> 1, qo1 = (w + 1) >> 1;
  const bool vp1 = po1 != po0 && po1 < Po, vq1 = qo1 != qo0 && qo1 < Qo;
  const int pos[2] = {po0, vp1 ? po1 : po0}, qos[2] = {qo0, vq1 ? qo1 : qo0};
  const bool valid[4] = {true, vq1, vp1, vp1 && vq1};
  PieceView<half> gv[4];
  unsigned long long packed[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int win = (pos[k >> 1] - (y0 >> 1)) * kSbPC + (qos[k & 1] - (x0 >> 1));
    gv[k].p = pool[win * 8 + cp];
    packed[k] = reinterpret_cast<const unsigned long long*>(pool + kSbGP)[win * 8 + cp];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int po = pos[k >> 1], qo = qos[k & 1];
    const unsigned mytap = (unsigned)((h - (po * 2 - 1)) * 3 + (w - (qo * 2 - 1)));
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (valid[k] && ((unsigned)(packed[k] >> (8 * e)) & 0xffu) == mytap) acc[e] += (float)gv[k].e[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = (float)(half)acc[e];
}

__device__ __forceinline__ void sb_tile_coords(const StemBwdArgs& a, int tile, int& b, int& y0, int& x0) {
  const int txi = tile % a.tiles_x;
  const int tmp = tile / a.tiles_x;
  b = tmp / a.tiles_y;
  y0 = (tmp % a.tiles_y) * kSbTH;
  x0 = txi * kSbTW;
}

// ---- 1. BatchNorm backward sums -----------------------------------------------------------------------------------------
// Walks the POOLED windows, not the conv-output pixels (round 3): a window's gradient goes to exactly one pixel -- its recorded
// argmax -- so  sum(gm) = sum over windows of gp * gate(y at the argmax),  sum(gm * xhat) likewise with xhat of that pixel:
// one 2-byte LDS read per (window, channel) instead of, per conv-output element, four windows' argmax bytes compared against
// the pixel's position (rounds 2: 348 us, bound by those ~110 VALU operations per 16-byte piece; the pixel order has 4x the
// elements).  A workgroup of 512 threads owns tiles of 2 x 32 windows: the 5 x 65 conv-output pixels they cover (41.6 KB) are
// staged in LDS by DMA, double-buffered; gp and the argmax bytes are read straight from memory, one 16-byte + one 8-byte
// piece per thread and tile, requested one tile ahead.  (The sums differ from the pixel-order form's in one respect: that
// form rounds a pixel's summed gradient to fp16 first, as the stored tensor of the four-launch chain was; here every window's
// fp16 gradient enters the fp32 sums directly.)
constexpr int kSrPH = 2, kSrPW = 32;                           // tile of pooled windows
constexpr int kSrCH = 2 * kSrPH + 1, kSrCW = 2 * kSrPW + 1;    // 5 x 65 conv-output pixels
constexpr int kSrPieces = kSrCH * kSrCW * 8;                   // 2600
constexpr int kSrPasses = (kSrPieces + 511) / 512;             // 6
constexpr int kSrBuf = kSrPasses * 512;                        // pieces per buffer (3072)
static __global__ void __launch_bounds__(512, 2) stem_bn_reduce_kernel(StemBwdArgs a) {
  __shared__ piece_t smem[2 * kSrBuf];  // two conv-output tiles (reused for the block reduction at the end)
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int tiles_px = (a.Qo + kSrPW - 1) / kSrPW, tiles_py = (a.Po + kSrPH - 1) / kSrPH;
  const int ntiles = a.B * tiles_px * tiles_py;
  const __amdgpu_buffer_rsrc_t rsrc_y = make_rsrc(a.y, (long)a.B * a.H0 * a.W0 * 64 * 2L);
  const int cp = t & 7, pc = (t >> 3) & 31, pr = t >> 8;
  float sc[8], sh[8], mu[8], is[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = cp * 8 + e;
    mu[e] = a.mean[c];
    is[e] = a.invstd[c];
    sc[e] = a.gamma[c] * is[e];
    sh[e] = a.beta[c] - mu[e] * sc[e];
    s1[e] = s2[e] = 0.f;
  }
  auto tile_coords = [&](int tile, int& b, int& p0, int& q0) {
    const int tx = tile % tiles_px, tmp = tile / tiles_px;
    b = tmp / tiles_py;
    p0 = (tmp % tiles_py) * kSrPH;
    q0 = tx * kSrPW;
  };
  auto issue_y = [&](int tile, int buf) {  // conv-output pixels rows 2 p0 - 1 .., columns 2 q0 - 1 ..
    int b, p0, q0;
    tile_coords(tile, b, p0, q0);
#pragma unroll
    for (int i = 0; i < kSrPasses; ++i) {
      const int q = t + i * 512;
      const int row = q / (kSrCW * 8), rem = q - row * (kSrCW * 8);
      const int iy = 2 * p0 - 1 + row, ix = 2 * q0 - 1 + (rem >> 3);
      const bool ok = q < kSrPieces && (unsigned)iy < (unsigned)a.H0 && (unsigned)ix < (unsigned)a.W0;
      const unsigned off = ok ? (unsigned)((b * a.H0 + iy) * a.W0 + ix) * 128u + (unsigned)((rem & 7) * 16) : ~0u;  // (< 4 GiB: net.hip validate())
      if (wave * 64 + i * 512 < kSrPieces) dma16(rsrc_y, off, 0u, &smem[buf * kSrBuf + wave * 64 + i * 512]);
    }
  };
  // this thread's window of a tile: gradient piece + argmax bytes, plain loads requested one tile ahead
  struct Win {
    PieceView<half> g;
    unsigned long long taps;
    bool ok;
  };
  auto load_win = [&](int tile, Win& w) {
    int b, p0, q0;
    tile_coords(tile, b, p0, q0);
    const int po = p0 + pr, qo = q0 + pc;
    w.ok = po < a.Po && qo < a.Qo;
    const long at = (((long)b * a.Po + (w.ok ? po : 0)) * a.Qo + (w.ok ? qo : 0)) * 64 + cp * 8;
    w.g.p = *reinterpret_cast<const piece_t*>(a.gp + at);
    w.taps = *reinterpret_cast<const unsigned long long*>(a.idx + at);
  };
  int tile = blockIdx.x;
  Win cur, nxt;
  if (tile < ntiles) {
    issue_y(tile, 0);
    load_win(tile, cur);
  }
  for (int it = 0; tile < ntiles; tile += gridDim.x, ++it) {
    wait_vmcnt<0>();
    __syncthreads();  // this tile's pixels are staged for everyone; everyone is done with the other buffer
    const bool more = tile + (int)gridDim.x < ntiles;
    if (more) {
      issue_y(tile + gridDim.x, (it + 1) & 1);
      load_win(tile + gridDim.x, nxt);
    }
    const half* yt = reinterpret_cast<const half*>(&smem[(it & 1) * kSrBuf]);
    const int base = ((2 * pr) * kSrCW + 2 * pc) * 64 + cp * 8;  // element offset of the window's top-left pixel
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int tap = (int)((cur.taps >> (8 * e)) & 0xffull);
      const int r = (tap * 11) >> 5, sx = tap - 3 * r;  // tap = 3 r + s, tap <= 8
      const float yv = (float)yt[base + (r * kSrCW + sx) * 64 + e];
      float gv = cur.ok ? (float)cur.g.e[e] : 0.f;
      if (!a.pre_gated && !(yv * sc[e] + sh[e] > 0.f)) gv = 0.f;  // ReLU gate recomputed from the conv output (the forward's arithmetic)
      s1[e] += gv;
      s2[e] += gv * (yv - mu[e]) * is[e];
    }
    if (more) cur = nxt;
  }
  // block reduction over the 64 threads that share a channel piece (t & 7), then fp64 atomics
  wait_vmcnt<0>();
  __syncthreads();
  float* r1 = reinterpret_cast<float*>(&smem[0]);  // [512 threads][16]: 32 KB
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    r1[t * 16 + e] = s1[e];
    r1[t * 16 + 8 + e] = s2[e];
  }
  __syncthreads();
  if (t < 128) {  // t = which * 64 + channel
    const int which = t >> 6, c = t & 63, ccp = c >> 3, ce = c & 7;
    double s = 0;
    for (int l = 0; l < 64; ++l) s += r1[(l * 8 + ccp) * 16 + which * 8 + ce];
    double* row = a.accum + (long)((int)blockIdx.x % a.accum_rows) * 2 * 64;
    atomicAdd(row + which * 64 + c, s);
  }
}

// ---- 2. weight gradient ---------------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256, 2) stem_wgrad_kernel(StemBwdArgs a) {
  constexpr int TH = kSbTH, TW = kSbTW;
  constexpr int IH = 2 * TH + 5, IW = TW + 3, IPIECES = IH * IW;  // 13 rows x 35 pairs
  constexpr int IPASS = (IPIECES + 255) / 256, IMG = IPASS * 256;
  constexpr int GYP = TH * TW * 8;  // pieces of the d(conv output) tile: 128 pixels x 64 halves
  // ONE LDS object: [2 input images][2 pool buffers][d(conv output) tile]
  __shared__ piece_t smem[2 * IMG + 2 * kSbPOOL + GYP];
  half* gyt = reinterpret_cast<half*>(&smem[2 * IMG + 2 * kSbPOOL]);

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ntiles = a.B * a.tiles_x * a.tiles_y;
  const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(a.xpad, (long)a.B * a.Hp * a.Wp2 * 16L);
  const __amdgpu_buffer_rsrc_t rsrc_g = make_rsrc(a.gp, (long)a.B * a.Po * a.Qo * 64 * 2L);
  const __amdgpu_buffer_rsrc_t rsrc_i = make_rsrc(a.idx, (long)a.B * a.Po * a.Qo * 64L);
  auto issue_tile = [&](int tile, int buf) {  // input image + pooled windows of a tile
    int b, y0, x0;
    sb_tile_coords(a, tile, b, y0, x0);
#pragma unroll
    for (int i = 0; i < IPASS; ++i) {
      const int q = t + i * 256;
      const int row = q / IW, col = q - row * IW;
      const int iy = 2 * y0 + row, ip = x0 + col;
      const bool ok = q < IPIECES && iy < a.Hp && ip < a.Wp2;
      const unsigned off = ok ? (unsigned)(((b * a.Hp + iy) * a.Wp2 + ip) * 16) : ~0u;
      dma16(rsrc, off, 0u, &smem[buf * IMG + wave * 64 + i * 256]);
    }
    sb_issue_pool(a, rsrc_g, rsrc_i, b, y0, x0, &smem[2 * IMG + buf * kSbPOOL], t, wave);
  };
  int tile = blockIdx.x;
  if (tile < ntiles) issue_tile(tile, 0);

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
    aA[j] = (unsigned)((2 * IMG + 2 * kSbPOOL) * 16) +
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

  // conv-output pieces of this thread in a tile: pixels (t >> 3) + 32 i, i < 4 -- plain loads, requested one tile ahead
  auto load_y = [&](int tl, PieceView<half> (&v)[4]) {
    int lb, ly0, lx0;
    sb_tile_coords(a, tl, lb, ly0, lx0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int px = (t >> 3) + 32 * i, oy = ly0 + (px >> 5), ox = lx0 + (px & 31);
      const int oyc = oy < a.H0 ? oy : a.H0 - 1, oxc = ox < a.W0 ? ox : a.W0 - 1;
      v[i].p = *reinterpret_cast<const piece_t*>(a.y + (((long)lb * a.H0 + oyc) * a.W0 + oxc) * 64 + cp * 8);
    }
  };
  PieceView<half> vy[4], vyn[4];
  if (tile < ntiles) load_y(tile, vy);
  for (int it = 0; tile < ntiles; tile += gridDim.x, ++it) {
    int b, y0, x0;
    sb_tile_coords(a, tile, b, y0, x0);
    wait_vmcnt<0>();
    __syncthreads();  // this tile's image and windows are staged; everyone is done with the previous tile's LDS reads
    const bool more = tile + (int)gridDim.x < ntiles;
    if (more) {
      issue_tile(tile + gridDim.x, (it + 1) & 1);
      load_y(tile + gridDim.x, vyn);
    }
    // ---- d(conv output) of the tile -> LDS.  (Measured and removed, round 3: the gather in WINDOW order as in
    //      stem_bn_reduce_kernel -- every staged window adds its gradient with ds_add_f32 to an fp32 sum of its argmax pixel,
    //      two tile rows at a time, then the sums are read back, rounded and turned into d(conv output) -- is slower, 869 vs
    //      700 us for the two launches: the LDS atomics and three more barriers per tile cost more than the byte compares
    //      they replace.  profiles/r03/c28_stem_wgrad_window_order_scatter.txt)
    const piece_t* pool = &smem[2 * IMG + (it & 1) * kSbPOOL];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int px = (t >> 3) + 32 * i, oy = y0 + (px >> 5), ox = x0 + (px & 31);
      const bool ok = oy < a.H0 && ox < a.W0;
      float g[8];
      sb_pool_grad(pool, ok ? oy : y0, ok ? ox : x0, y0, x0, cp, a.Po, a.Qo, g);
      PieceView<half> o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float yv = (float)vy[i].e[e];
        float gv = g[e];
        if (!a.pre_gated && !(yv * k1[e] + sh[e] > 0.f)) gv = 0.f;  // ReLU gate recomputed from the conv output (the forward's arithmetic)
        const float xh = (yv - mu[e]) * is[e];
        o.e[e] = ok ? (half)(k1[e] * (gv - mg[e] - xh * mgx[e])) : (half)0.f;
      }
      *reinterpret_cast<piece_t*>(gyt + px * 64 + ((cp ^ wg_swz<8>(px)) * 8)) = o.p;
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < 4; ++i) vy[i].p = vyn[i].p;
    }
    __syncthreads();  // gradient tile complete
    // ---- GEMM: 8 K-steps of 16 pixels (tile row ty, half hx)
    const unsigned img = (unsigned)((it & 1) * IMG * 16);
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

inline void stem_bwd_geometry(StemBwdArgs& a, int B, int H, int W, int Wp) {
  a.B = B; a.Hp = H + 6; a.Wp2 = Wp / 2;
  a.H0 = (H - 1) / 2 + 1; a.W0 = (W - 1) / 2 + 1;
  a.Po = (a.H0 + 2 - 3) / 2 + 1; a.Qo = (a.W0 + 2 - 3) / 2 + 1;
  a.tiles_x = cdiv(a.W0, kSbTW); a.tiles_y = cdiv(a.H0, kSbTH);
}
inline int stem_bwd_grid(const StemBwdArgs& a, int per_cu) {
  static const int wgs = getenv("MN_STEM_WGS") ? atoi(getenv("MN_STEM_WGS")) : 0;
  const int want = wgs > 0 ? wgs : 256 * per_cu;
  const int ntiles = a.B * a.tiles_x * a.tiles_y;
  return ntiles < want ? ntiles : want;
}
// sums into a.accum (zero on entry); the caller runs bn_finalize_bwd_kernel on them
inline void launch_stem_bn_reduce(StemBwdArgs a, int B, int H, int W, int Wp, hipStream_t stream) {
  stem_bwd_geometry(a, B, H, W, Wp);
  const int ntiles = a.B * cdiv(a.Po, kSrPH) * cdiv(a.Qo, kSrPW);  // (tiles of pooled windows; one workgroup per CU)
  static const int wgs = getenv("MN_STEM_WGS") ? atoi(getenv("MN_STEM_WGS")) : 0;
  const int want = wgs > 0 ? wgs : 256;
  hipLaunchKernelGGL(stem_bn_reduce_kernel, dim3(ntiles < want ? ntiles : want), dim3(512), 0, stream, a);
}
inline void launch_stem_wgrad(StemBwdArgs a, int B, int H, int W, int Wp, hipStream_t stream) {
  stem_bwd_geometry(a, B, H, W, Wp);
  static const int per_cu = getenv("MN_STEM_WGRAD_PER_CU") ? atoi(getenv("MN_STEM_WGRAD_PER_CU")) : 2;  // (A/B knob)
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3(stem_bwd_grid(a, per_cu)), dim3(256), 0, stream, a);
}

}  // namespace mn
