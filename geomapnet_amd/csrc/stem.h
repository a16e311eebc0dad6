// Stem convolution (7x7 / stride 2 / pad 3, 3 -> 64 channels: torchvision ResNet conv1, call site
// /root/reference/models/posenet.py:65 through feature_extractor) for fp16 tensors, forward.
//
// The implicit-GEMM kernel (igemm.h) treats the stem as a 7x4 convolution over pixel PAIRS of the zero-padded NHWC4 image
// (K = 7 * 4 * 8 = 224) and stages an im2col tile per workgroup: 128 rows x 448 B of A plus the 28 KB weight matrix per
// 128 output pixels -- 85 KB of LDS-DMA for 900 matrix-core cycles, i.e. the launch runs at the DMA rate (392 us for 79
// GFLOP; its HBM floor -- 538 MB of output, 140 MB of input -- is ~130 us).  Here nothing is im2col'd:
//   * the WEIGHTS (28 KB) live in registers: 14 K-steps x 2 column tiles x 4 VGPRs = 112 per lane, loaded once;
//   * a workgroup owns an 8 x 32-pixel output tile and DMAs the 21 x 35 pixel pairs it reads (11.8 KB) into LDS once;
//   * the A fragment of MFMA step (r, kk) for output pixel x is ONE 16-byte pair -- row 2y + r, pair x + 2 kk + (lane >> 5)
//     of that image -- read straight from LDS (consecutive lanes = consecutive pairs: conflict-free ds_read_b128).
// One wave computes two output rows of 32 pixels x 64 channels (28 MFMAs each), rounds to fp16 through a private 4 KB
// LDS staging block (16-byte stores along channels) and keeps the BatchNorm column sums of its pixels.
// Workgroups are persistent (512 of them walk the 18 432 tiles of a 192-image batch): the weights are loaded once per
// workgroup instead of once per tile (516 MB of L2 traffic otherwise), the next tile's image is DMA'd while the current
// one is computed, and the BatchNorm sums leave each workgroup once (128 fp64 atomics instead of 128 per tile).
#pragma once
#include "igemm.h"

namespace mn {

struct StemArgs {
  const half* xpad;  // [B][Hp][Wp][4] zero-padded input (pad 3 top/left): pixel pairs of 16 bytes, Wp even
  const half* wf;    // [64][224] weights in the pair layout (optim.h, repack mode 2)
  half* y;           // [B][H0][W0][64] raw conv output
  double* stats_accum;  // [stats_rows][2][64] fp64 column sums (sum, sum of squares), added to atomically; or null
  int stats_rows;
  int B, Hp, Wp2;    // Wp2 = Wp / 2 pairs per padded row
  int H0, W0;
  int tiles_x, tiles_y;
};

constexpr int kStemTH = 8, kStemTW = 32;

static __global__ void __launch_bounds__(256, 2) stem_conv_kernel(StemArgs a) {
  constexpr int TH = kStemTH, TW = kStemTW;
  constexpr int IH = 2 * TH + 5, IW = TW + 3, IPIECES = IH * IW;  // 21 rows x 35 pairs
  constexpr int IPASS = (IPIECES + 255) / 256;
  // ONE LDS object: [2 input images][4 staging blocks of 32 pixels x 64 halves][column sums 4 x 64 x 2 floats]
  __shared__ piece_t smem[2 * IPASS * 256 + 4 * 256 + 128];
  piece_t* stage_all = &smem[2 * IPASS * 256];
  float* red = reinterpret_cast<float*>(&smem[2 * IPASS * 256 + 4 * 256]);

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ntiles = a.B * a.tiles_x * a.tiles_y;
  const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(a.xpad, (long)a.B * a.Hp * a.Wp2 * 16L);

  // input image of a tile: one DMA sweep (rows 2 y0 .. 2 y0 + 20 of the padded image, pairs x0 .. x0 + 34)
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
  // persistent workgroups: tile = blockIdx.x, + gridDim.x, ...; the weights are loaded ONCE, the next tile's image is in
  // flight while the current one is computed, the BatchNorm sums leave the workgroup once
  int tile = blockIdx.x;
  if (tile < ntiles) issue_image(tile, 0);

  // ---- weights -> registers while the first image is in flight: B fragment of step s, column tile j = 8 halves of row
  // n = 32 j + (lane & 31) at k = 16 s + 8 (lane >> 5)
  PieceView<half> wb[14][2];
  {
    const half* wrow = a.wf + (lane & 31) * 224 + (lane >> 5) * 8;
#pragma unroll
    for (int s = 0; s < 14; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) wb[s][j].p = *reinterpret_cast<const piece_t*>(wrow + j * 32 * 224 + s * 16);
  }

  const int l31 = lane & 31, kh = lane >> 5;
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
  half* stage = reinterpret_cast<half*>(stage_all + wave * 256);  // [32 pixels][64 channels]
  for (int it = 0; tile < ntiles; tile += gridDim.x, ++it) {
    wait_vmcnt<0>();               // this tile's image (and the previous tile's stores)
    __builtin_amdgcn_s_barrier();  // ... for every wave; everyone is done reading the other image
    if (tile + (int)gridDim.x < ntiles) issue_image(tile + gridDim.x, (it + 1) & 1);
    const int txi = tile % a.tiles_x;
    const int tmp = tile / a.tiles_x;
    const int tyi = tmp % a.tiles_y, b = tmp / a.tiles_y;
    const int y0 = tyi * TH, x0 = txi * TW;
    const piece_t* image = &smem[(it & 1) * IPASS * 256];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int ty = 2 * wave + rr;  // tile row of this pass
      floatx16 acc[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      const piece_t* img = image + (2 * ty) * IW + l31 + kh;
#pragma unroll
      for (int s = 0; s < 14; ++s) {
        PieceView<half> fa;
        fa.p = img[(s >> 1) * IW + 2 * (s & 1)];
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa.v, wb[s][j].v, acc[j], 0, 0, 0);
      }
      // epilogue of the row: column sums over the pixels inside the image, fp16 through the wave's staging block
      const int oy = y0 + ty;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = (r & 3) + 8 * (r >> 2) + 4 * kh;
          const float v = acc[j][r];
          const bool ok = oy < a.H0 && x0 + px < a.W0;
          const float vs = ok ? v : 0.f;
          s1[j] += vs;
          s2[j] += vs * vs;
          stage[px * 64 + j * 32 + l31] = (half)v;
        }
      // the same wave wrote and now reads the block: lockstep + in-order LDS on the GPU (the builtin emits no
      // instruction; the emulator's fibers rendezvous there)
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int id = lane + 64 * i, px = id >> 3, cp = id & 7;
        const piece_t v = *reinterpret_cast<const piece_t*>(stage + px * 64 + cp * 8);
        if (oy < a.H0 && x0 + px < a.W0)
          *reinterpret_cast<piece_t*>(a.y + (((long)b * a.H0 + oy) * a.W0 + x0 + px) * 64 + cp * 8) = v;
      }
      __builtin_amdgcn_wave_barrier();  // reads done before the next row's values overwrite the block
    }
  }
  if (a.stats_accum) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      s1[j] += __shfl_xor(s1[j], 32);
      s2[j] += __shfl_xor(s2[j], 32);
      if (lane < 32) {
        red[(wave * 64 + j * 32 + lane) * 2 + 0] = s1[j];
        red[(wave * 64 + j * 32 + lane) * 2 + 1] = s2[j];
      }
    }
    __syncthreads();
    if (t < 64) {
      float sa = 0.f, sb = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        sa += red[(w * 64 + t) * 2 + 0];
        sb += red[(w * 64 + t) * 2 + 1];
      }
      double* row = a.stats_accum + (long)((int)blockIdx.x % a.stats_rows) * 2 * 64;
      atomicAdd(row + t, (double)sa);
      atomicAdd(row + 64 + t, (double)sb);
    }
  }
}

// xpad: [B][H + 6][Wp][4] with Wp even >= W + 7; y: [B][H0][W0][64], H0 = (H - 1) / 2 + 1, W0 = (W - 1) / 2 + 1
inline void launch_stem_conv(const half* xpad, const half* wf, half* y, double* stats_accum, int stats_rows, int B, int H, int W,
                             int Wp, hipStream_t stream) {
  StemArgs a;
  a.xpad = xpad; a.wf = wf; a.y = y; a.stats_accum = stats_accum; a.stats_rows = stats_rows > 0 ? stats_rows : 1;
  a.B = B; a.Hp = H + 6; a.Wp2 = Wp / 2;
  a.H0 = (H - 1) / 2 + 1; a.W0 = (W - 1) / 2 + 1;
  a.tiles_x = cdiv(a.W0, kStemTW); a.tiles_y = cdiv(a.H0, kStemTH);
  // persistent: two workgroups per CU (the 220-register kernel's occupancy), each walking tiles blockIdx.x, + grid, ...
  static const int wgs = getenv("MN_STEM_WGS") ? atoi(getenv("MN_STEM_WGS")) : 512;
  const int ntiles = B * a.tiles_x * a.tiles_y;
  hipLaunchKernelGGL(stem_conv_kernel, dim3(ntiles < wgs ? ntiles : wgs), dim3(256), 0, stream, a);
}

// ---- fp32 tensors (fp32x3 / fp16x2 modes): the same kernel shape with split operands --------------------------------------------
// The parity modes ran the stem through igemm.h's per-lane tap walk (the only configuration that handles 16-byte taps) on
// fp32 tensors with the operands split after every LDS read: 760 us per launch at 192 images against this launch's HBM floor of
// ~300 us (0.28 GB of input, 1.08 GB of fp32 output).  Here, as above, nothing is im2col'd:
//   * WEIGHTS: read once per persistent workgroup as fp32, split: the hi halves live in registers (112 per lane), the lo halves
//     in LDS as ready-made B fragments (28 KB, conflict-free ds_read_b128);
//   * INPUT: the 21 x 35 pixel pairs of a tile are loaded as fp32 into registers a whole tile ahead, split, and stored as TWO
//     fp16 images (hi, lo) that the A fragments are read from exactly as in the fp16 kernel;
//   * a product = three MFMAs (lo*hi + hi*lo + hi*hi); the output leaves the accumulators as fp32, 128 contiguous bytes per
//     half-wave and register (no staging).
struct StemX3Args {
  const float* xpad;  // [B][Hp][Wp][4] zero-padded fp32 input
  const float* wf;    // [64][224] fp32 weights in the pair layout (optim.h, repack mode 2)
  float* y;           // [B][H0][W0][64] raw conv output
  double* stats_accum;
  int stats_rows;
  int B, Hp, Wp2, H0, W0, tiles_x, tiles_y;
};

static __global__ void __launch_bounds__(256, 2) stem_conv_x3_kernel(StemX3Args a) {
  constexpr int TH = kStemTH, TW = kStemTW;
  constexpr int IH = 2 * TH + 5, IW = TW + 3, IPIECES = IH * IW;  // 21 rows x 35 pairs
  constexpr int IPASS = (IPIECES + 255) / 256, IMG = IPASS * 256;  // pieces of one fp16 image (a piece = one pixel pair)
  constexpr int WL = 14 * 2 * 64;                                  // lo-half B fragments: [step][column tile][lane]
  // ONE LDS object: [2 buffers][hi image | lo image][lo weights][column sums 4 x 64 x 2 floats]
  __shared__ piece_t smem[4 * IMG + WL + 128];
  piece_t* wl = &smem[4 * IMG];
  float* red = reinterpret_cast<float*>(&smem[4 * IMG + WL]);

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ntiles = a.B * a.tiles_x * a.tiles_y;

  // fp32 pixel pairs of a tile -> registers (two 16-byte pixels per pair; zero outside the padded image / past the tile's pairs)
  struct Staged {
    piece_t px[IPASS][2];
  };
  auto load_image = [&](int tile, Staged& r) {
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
      const piece_t* src = reinterpret_cast<const piece_t*>(a.xpad) + (((long)b * a.Hp + iy) * a.Wp2 + ip) * 2;
      r.px[i][0] = ok ? src[0] : zero_piece();
      r.px[i][1] = ok ? src[1] : zero_piece();
    }
  };
  auto store_image = [&](const Staged& r, int buf) {
#pragma unroll
    for (int i = 0; i < IPASS; ++i) {
      PieceView<float> p0, p1;
      p0.p = r.px[i][0];
      p1.p = r.px[i][1];
      const float x[8] = {p0.e[0], p0.e[1], p0.e[2], p0.e[3], p1.e[0], p1.e[1], p1.e[2], p1.e[3]};
      PieceView<half> hi, lo;
      split8_f16(x, hi.v, lo.v);
      smem[(2 * buf + 0) * IMG + t + i * 256] = hi.p;
      smem[(2 * buf + 1) * IMG + t + i * 256] = lo.p;
    }
  };

  // persistent workgroups: tile = blockIdx.x, + gridDim.x, ...
  int tile = blockIdx.x;
  Staged st;
  if (tile < ntiles) {
    load_image(tile, st);
    store_image(st, 0);
  }
  if (tile + (int)gridDim.x < ntiles) load_image(tile + gridDim.x, st);  // in flight under the weight setup and the first tile

  // weights: B fragment of step s, column tile j = 8 values of row n = 32 j + (lane & 31) at k = 16 s + 8 (lane >> 5): hi halves
  // -> registers, lo halves -> LDS (every wave writes the same values to the same slots)
  PieceView<half> wb[14][2];
  {
    const float* wrow = a.wf + (lane & 31) * 224 + (lane >> 5) * 8;
#pragma unroll
    for (int s = 0; s < 14; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const piece_t* src = reinterpret_cast<const piece_t*>(wrow + j * 32 * 224 + s * 16);
        PieceView<float> p0, p1;
        p0.p = src[0];
        p1.p = src[1];
        const float x[8] = {p0.e[0], p0.e[1], p0.e[2], p0.e[3], p1.e[0], p1.e[1], p1.e[2], p1.e[3]};
        PieceView<half> lo;
        split8_f16(x, wb[s][j].v, lo.v);
        if (wave == 0) wl[(s * 2 + j) * 64 + lane] = lo.p;
      }
  }

  const int l31 = lane & 31, kh = lane >> 5;
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
  for (int it = 0; tile < ntiles; tile += gridDim.x, ++it) {
    __syncthreads();  // this tile's images (and, the first time, the lo weights) are in LDS; everyone is done with the other buffer
    const int txi = tile % a.tiles_x;
    const int tmp = tile / a.tiles_x;
    const int tyi = tmp % a.tiles_y, b = tmp / a.tiles_y;
    const int y0 = tyi * TH, x0 = txi * TW;
    const piece_t* image_hi = &smem[(2 * (it & 1) + 0) * IMG];
    const piece_t* image_lo = &smem[(2 * (it & 1) + 1) * IMG];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int ty = 2 * wave + rr;  // tile row of this pass
      floatx16 acc[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      const int base = (2 * ty) * IW + l31 + kh;
#pragma unroll
      for (int s = 0; s < 14; ++s) {
        PieceView<half> fh, fl;
        fh.p = image_hi[base + (s >> 1) * IW + 2 * (s & 1)];
        fl.p = image_lo[base + (s >> 1) * IW + 2 * (s & 1)];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          PieceView<half> wlo;
          wlo.p = wl[(s * 2 + j) * 64 + lane];
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl.v, wb[s][j].v, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh.v, wlo.v, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh.v, wb[s][j].v, acc[j], 0, 0, 0);
        }
      }
      // epilogue of the row: column sums over the pixels inside the image; fp32 stores straight from the accumulators (a
      // register of a half-wave = 32 consecutive channels of one pixel = 128 bytes)
      const int oy = y0 + ty;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = (r & 3) + 8 * (r >> 2) + 4 * kh;
          const float v = acc[j][r];
          const bool ok = oy < a.H0 && x0 + px < a.W0;
          if (ok) {
            s1[j] += v;
            s2[j] += v * v;
            a.y[(((long)b * a.H0 + oy) * a.W0 + x0 + px) * 64 + j * 32 + l31] = v;
          }
        }
    }
    // the tile after next: its loads were issued a tile ago; split + store them into the buffer the previous tile used (every
    // wave passed this iteration's barrier after its last read of it), then request the one after that
    if (tile + (int)gridDim.x < ntiles) {
      store_image(st, (it + 1) & 1);
      if (tile + 2 * (int)gridDim.x < ntiles) load_image(tile + 2 * gridDim.x, st);
    }
  }
  if (a.stats_accum) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      s1[j] += __shfl_xor(s1[j], 32);
      s2[j] += __shfl_xor(s2[j], 32);
      if (lane < 32) {
        red[(wave * 64 + j * 32 + lane) * 2 + 0] = s1[j];
        red[(wave * 64 + j * 32 + lane) * 2 + 1] = s2[j];
      }
    }
    __syncthreads();
    if (t < 64) {
      float sa = 0.f, sb = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        sa += red[(w * 64 + t) * 2 + 0];
        sb += red[(w * 64 + t) * 2 + 1];
      }
      double* row = a.stats_accum + (long)((int)blockIdx.x % a.stats_rows) * 2 * 64;
      atomicAdd(row + t, (double)sa);
      atomicAdd(row + 64 + t, (double)sb);
    }
  }
}

inline void launch_stem_conv_x3(const float* xpad, const float* wf, float* y, double* stats_accum, int stats_rows, int B, int H,
                                int W, int Wp, hipStream_t stream) {
  StemX3Args a;
  a.xpad = xpad; a.wf = wf; a.y = y; a.stats_accum = stats_accum; a.stats_rows = stats_rows > 0 ? stats_rows : 1;
  a.B = B; a.Hp = H + 6; a.Wp2 = Wp / 2;
  a.H0 = (H - 1) / 2 + 1; a.W0 = (W - 1) / 2 + 1;
  a.tiles_x = cdiv(a.W0, kStemTW); a.tiles_y = cdiv(a.H0, kStemTH);
  static const int wgs = getenv("MN_STEM_WGS") ? atoi(getenv("MN_STEM_WGS")) : 512;
  const int ntiles = B * a.tiles_x * a.tiles_y;
  hipLaunchKernelGGL(stem_conv_x3_kernel, dim3(ntiles < wgs ? ntiles : wgs), dim3(256), 0, stream, a);
}

}  // namespace mn
