// 3x3 stride-1 "same" convolutions of fp16 tensors with 128-column tiles and 128 x 64 REGISTER tiles per wave
// (layer2: 128 -> 128 channels on 32 x 43 maps, forward and data gradient).
//
// Why: every MFMA of the 12-wave kernels (igemm.h, igemm_halo.h) is fed from LDS with 96 x 32 / 96 x 64 wave tiles --
// for the 128-column shape 4 fragment reads per 3 MFMAs = 1.33 KB of ds_read_b128 per 32-cycle MFMA, i.e. at full
// matrix rate the four SIMDs would need 170 B/clk of an LDS that delivers 128.  The launches sit at ~30 % of the MFMA
// peak (layer2: 105 us for 78 GFLOP) because the LDS read port is the bound (profiles/r02: halo_pp ablations -- the
// fragment reads alone take longer than the MFMAs).  Here a wave owns a 128 x 64 tile (4 x 2 MFMA tiles, 128
// accumulator registers): 6 fragment reads per 8 MFMAs = 0.75 KB per MFMA.
//
// Shape: 256 x 128 tile, 4 waves (2 x 2), 256 threads, <= 256 registers -> two workgroups per CU (80 KB of LDS each):
//   * A: ONE chunk-resident image (igemm_halo.h's idea): rows [m0 - (W+1), m0 + 256 + (W+1)) of the flattened pixel
//     order for the current 64-channel chunk; a tap is a row shift, image-border lanes read a 16-byte zero slot.  The
//     next chunk's image is requested when the last tap of the current one has been read by every wave (one extra
//     barrier + DMA round trip per chunk -- the CU's other workgroup computes meanwhile);
//   * B: the tap's 128 x 64 weight slice through a two-slot ring, requested one K-step ahead;
//   * one s_waitcnt vmcnt(0) + s_barrier per K-step (a K-step = one tap of one chunk = 32 MFMAs per wave).
// Epilogue in registers (halo_pp.h's): the MFMA operands are swapped -- weights as A, pixels as B -- so a lane holds four
// consecutive channels of one pixel per accumulator quad: 8-byte stores / residual / gate loads, no LDS staging, no
// barrier.  BatchNorm column sums: per-lane sums over the lane's pixels, folded across the 32 pixel lanes.
#pragma once
#include "halo_pp.h"
#include "igemm.h"

namespace mn {

// STATS: the form with BatchNorm column sums (forward) takes no residual / gates -- the sums' 64 registers replace theirs.
template <int kAH, bool STATS>
static __global__ void __launch_bounds__(256, 2) igemm_rt_kernel(GatherGeom g, const half* __restrict__ A,
                                                                 const half* __restrict__ Bw, Epilogue ep, int grid_n,
                                                                 RowDiv rd) {
  constexpr int NP = 8, WM = 2, WN = 2, TM = 4, TN = 2, NT = 256;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, WTM = TM * 32, WTN = TN * 32, RPP = NT / NP;  // 32 rows per DMA pass
  constexpr int A_IMG = kAH * NP, B_SLOT = BN * NP, RING = A_IMG + 2 * B_SLOT;                     // pieces
  constexpr int A_PASSES = kAH / RPP, B_PASSES = BN / RPP;
  constexpr unsigned kOob = 0x80000000u;
  static_assert(kAH % RPP == 0, "image rows per DMA pass");
  static_assert((RING + 1 + WM * BN / 2) * 16 <= 80 * 1024, "two workgroups per CU");
  __shared__ piece_t smem[RING + 1 + WM * BN / 2];
  float* red = reinterpret_cast<float*>(&smem[RING + 1]);  // [WM][BN][2]

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = tile / grid_n, tile_n = tile - tile_m * grid_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int W = g.Wi, halo = W + 1;
  const int gM = g.M, gN = g.N, gC = g.C, gK = g.K, ldc = ep.ldc;
  const int NCH = gC / 64, KT = 9 * NCH;
  if (t == 0) smem[RING] = zero_piece();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the zero slot is in LDS before this wave reaches the first barrier

  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(A, (long)gM * gC * 2L);
  const __amdgpu_buffer_rsrc_t rsrc_b = make_rsrc(Bw, (long)gN * gK * 2L);
  const int pc = t % NP, lrow = t / NP;
  const int src_piece = pc ^ lds_swz<NP>(lrow);  // rows of one thread differ by multiples of 32: invisible to the swizzle

  // A image of `chunk`: pass i = rows lrow + 32 i of the image (pixels m0 - halo + row)
  auto issue_a = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int pix = m0 - halo + lrow + i * RPP;
      const unsigned off = (pix >= 0 && pix < gM) ? (unsigned)pix * (unsigned)(gC * 2) + (unsigned)src_piece * 16u : ~0u;
      dma16(rsrc_a, off, (unsigned)(chunk * 128), &smem[i * RPP * NP + wave * 64]);
    }
  };
  unsigned b_off[B_PASSES];
#pragma unroll
  for (int j = 0; j < B_PASSES; ++j) {
    const int n = n0 + lrow + j * RPP;
    b_off[j] = n < gN ? (unsigned)(n * gK) * 2u + (unsigned)src_piece * 16u : ~0u;
  }
  auto issue_b = [&](int kt) __attribute__((always_inline)) {
    const int chunk = kt / 9, tap = kt - chunk * 9;
    const unsigned soff = (unsigned)((tap * gC + chunk * 64) * 2);
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) dma16(rsrc_b, b_off[j], soff, &smem[A_IMG + (kt & 1) * B_SLOT + j * RPP * NP + wave * 64]);
  };

  // validity of (row tile, tap) for this lane's pixel: bit 9 i + tap set = outside the image (reads the zero slot)
  const int l31 = lane & 31, hi = lane >> 5;
  unsigned long long inv_mask = 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * WTM + i * 32 + l31;
    unsigned inv = 0x1ffu;
    if (m < gM) {
      const int tmp = fastdiv(m, rd.q), x = m - tmp * g.Q;
      const int bb = fastdiv(tmp, rd.p), y = tmp - bb * g.P;
      inv = 0;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int yy = y + g.off_h + g.rsign * r, xx = x + g.off_w + g.ssign * s;
          const bool ok = (unsigned)yy < (unsigned)g.Hi && (unsigned)xx < (unsigned)g.Wi;
          inv |= (ok ? 0u : 1u) << (r * 3 + s);
        }
    }
    inv_mask |= (unsigned long long)inv << (9 * i);
  }
  const int tap_base = halo + g.off_h * W + g.off_w, tap_dr = g.rsign * W, tap_ds = g.ssign;

  // acc[i][j]: pixels wm * 128 + i * 32 + (lane & 31) x channels wn * 64 + j * 32 + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  issue_a(0);
  issue_b(0);
  int chunk = 0, tap = 0, tr = 0, ts = 0;
  for (int kt = 0; kt < KT; ++kt) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // this step's B slice (and a first image) landed; the previous step's reads are done
    if (kt + 1 < KT) issue_b(kt + 1);
    if (tap == 0 && chunk > 0) {
      issue_a(chunk);  // every wave is past the previous chunk's last tap
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
    }
    const piece_t* tb = &smem[A_IMG + (kt & 1) * B_SLOT];
    const int shift = tap_base + tr * tap_dr + ts * tap_ds;  // scalar
    int aidx[TM];  // piece index of this lane's row (before the K-sub-step's piece), or < 0 = the zero slot
    int aswz[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int arow = wm * WTM + i * 32 + l31 + shift;
      aswz[i] = lds_swz<NP>(arow);
      aidx[i] = ((inv_mask >> (9 * i + tap)) & 1ull) ? -1 : arow * NP;
    }
    const int brow = wn * WTN + l31, bswz = lds_swz<NP>(l31);
    PieceView<half> fa[2][TM], fb[2][TN];
    auto load_frags = [&](int ks, int slot) __attribute__((always_inline)) {
      const int piece = ks * 2 + hi;
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[slot][i].p = smem[aidx[i] < 0 ? RING : aidx[i] + (piece ^ aswz[i])];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[slot][j].p = tb[(brow + j * 32) * NP + (piece ^ bswz)];
    };
    load_frags(0, 0);
#pragma unroll
    for (int ks = 0; ks < NP / 2; ++ks) {
      if (ks + 1 < NP / 2) load_frags(ks + 1, (ks + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma_piece<half>(fb[ks & 1][j], fa[ks & 1][i], acc[i][j]);  // weights = A operand
      __builtin_amdgcn_sched_barrier(0);
    }
    if (++ts == 3) {
      ts = 0;
      ++tr;
    }
    if (++tap == 9) {
      tap = tr = ts = 0;
      ++chunk;
    }
  }

  // ---- epilogue (registers only) -------------------------------------------------------------------------------
  const long out_bytes = (long)gM * ldc * 2L;
  const __amdgpu_buffer_rsrc_t rsrc_out = make_rsrc(ep.out, out_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_res = make_rsrc(ep.res, ep.res ? out_bytes : 0L);  // null tensor: loads return zeros
  const __amdgpu_buffer_rsrc_t rsrc_g1 = make_rsrc(ep.res_gate ? ep.res_gate : ep.out_gate, (ep.res_gate || ep.out_gate) ? out_bytes : 0L);
  const bool has_res = !STATS && ep.res != nullptr, has_gate = !STATS && ep.res_gate != nullptr;
  const bool has_ogate = !STATS && ep.out_gate != nullptr;
  const bool side = has_res || has_gate || has_ogate;
  const float alpha = ep.alpha;
  const bool relu = (ep.relu & 1) != 0;
  unsigned voff[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * WTM + i * 32 + l31;
    voff[i] = m < gM ? (unsigned)((m * ldc + n0 + wn * WTN) * 2 + hi * 8) : kOob;
  }
  // residual and THE gate (res_gate or out_gate) of row tile i, requested two row tiles ahead of their use
  Half4View rv[2][TN][4], gv[2][TN][4];
  auto load_side = [&](int i, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        rv[slot][j][q].p = __builtin_amdgcn_raw_buffer_load_b64(rsrc_res, (int)voff[i], j * 64 + q * 16, 0);
        gv[slot][j][q].p = __builtin_amdgcn_raw_buffer_load_b64(rsrc_g1, (int)voff[i], j * 64 + q * 16, 0);
      }
  };
  float st1[STATS ? TN : 1][16], st2[STATS ? TN : 1][16];
  if constexpr (STATS) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) st1[j][r] = st2[j][r] = 0.f;
  }
  if (side) {
    load_side(0, 0);
    load_side(1, 1);
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        Half4View o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = q * 4 + e;
          float v = acc[i][j][r] * alpha;  // rows past M hold exact zeros (every tap read the zero slot)
          if (relu) v = fmaxf(v, 0.f);
          if constexpr (STATS) {
            st1[j][r] += v;
            st2[j][r] += v * v;
          }
          if (has_res) {
            float xr = (float)rv[i & 1][j][q].e[e];
            if (has_gate && !((float)gv[i & 1][j][q].e[e] > 0.f)) xr = 0.f;
            v += xr;
          }
          if (has_ogate && !((float)gv[i & 1][j][q].e[e] > 0.f)) v = 0.f;
          o.e[e] = (half)v;
        }
        __builtin_amdgcn_raw_buffer_store_b64(o.p, rsrc_out, (int)voff[i], j * 64 + q * 16, 0);
      }
    if (side && i + 2 < TM) load_side(i + 2, i & 1);
  }
  if constexpr (STATS) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float a = st1[j][r], b = st2[j][r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
          a += __shfl_xor(a, o);
          b += __shfl_xor(b, o);
        }
        if (l31 == 0) {
          const int lc = wn * WTN + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          red[(wm * BN + lc) * 2 + 0] = a;
          red[(wm * BN + lc) * 2 + 1] = b;
        }
      }
    __syncthreads();
    for (int c = t; c < BN; c += NT)
      if (n0 + c < gN) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          a += red[(w * BN + c) * 2 + 0];
          b += red[(w * BN + c) * 2 + 1];
        }
        if (ep.stats_accum) {
          double* row = ep.stats_accum + (long)(tile_m % ep.stats_rows) * 2 * gN;
          atomicAdd(row + n0 + c, (double)a);
          atomicAdd(row + gN + n0 + c, (double)b);
        } else {
          ep.stats[((long)tile_m * 2 + 0) * gN + n0 + c] = a;
          ep.stats[((long)tile_m * 2 + 1) * gN + n0 + c] = b;
        }
      }
  }
}

constexpr int kRtImageRows = 352;  // 256 + 2 (W + 1) <= 352: W <= 47

// the launches the kernel covers: igemm_halo's (fp16 3x3 stride 1 same size, 64-channel chunks) with N a multiple of
// 128, dense [M][N] output, no bias, at most one gate
inline bool igemm_rt_applies(const GatherGeom& g, const Epilogue& ep) {
  return g.R == 3 && g.S == 3 && g.mul_p == 1 && g.mul_q == 1 && g.div == 1 && g.P == g.Hi && g.Q == g.Wi &&
         g.C % 64 == 0 && g.N % 128 == 0 && g.K == 9 * g.C && !g.bt_on && (g.ldb == 0 || g.ldb == g.K) &&
         (g.rsign == 1 || g.rsign == -1) && g.rsign == g.ssign && 256 + 2 * (g.Wi + 1) <= kRtImageRows &&
         g.M == g.B * g.P * g.Q && (long)g.M * g.C * 2 < 0x7ffffff0l && (long)g.N * g.K * 2 < 0xfffffff0l && !ep.om_on &&
         ep.bias == nullptr && ep.ldc % 4 == 0 && (long)g.M * ep.ldc * 2 < 0x7ffffff0l && !(ep.res_gate && ep.out_gate) &&
         (ep.stats_accum == nullptr || ep.stats_rows > 0) &&
         (!(ep.stats || ep.stats_accum) || !(ep.res || ep.res_gate || ep.out_gate));
}

// returns the number of M-blocks used (rows of a [grid_m][2][N] statistics buffer)
inline int launch_igemm_rt(const GatherGeom& g, const half* A, const half* Bw, const Epilogue& ep, hipStream_t stream) {
  const int gm = cdiv(g.M, 256), gn = g.N / 128;
  RowDiv rd;
  rd.q = make_fastdiv(g.Q);
  rd.p = make_fastdiv(g.P);
  if (ep.stats || ep.stats_accum)
    hipLaunchKernelGGL((igemm_rt_kernel<kRtImageRows, true>), dim3(gm * gn), dim3(256), 0, stream, g, A, Bw, ep, gn, rd);
  else
    hipLaunchKernelGGL((igemm_rt_kernel<kRtImageRows, false>), dim3(gm * gn), dim3(256), 0, stream, g, A, Bw, ep, gn, rd);
  return gm;
}

}  // namespace mn
