// 3x3 stride-1 convolution of 64 -> 64 channel fp16 tensors (ResNet layer1, forward and data gradient): persistent
// "ping-pong" form of halo.h.
//
// halo.h runs one 16x16-pixel tile per 4-wave workgroup: halo DMA -> wait -> nine taps with a barrier each (the tap's
// weight slice streams through a two-slot ring) -> epilogue.  Its ablations (profiles/r02) show the three phases ADD:
// a CU holds two such workgroups and they drift into the same phase, so the matrix pipe idles through both DMA waits,
// nine barrier bubbles per tile and the epilogue (121 us for a launch whose MFMA time is 35 us and whose HBM floor is
// 20 us).  Here a workgroup is 8 waves in TWO groups of four, persistent over the tiles of the launch (one workgroup
// per CU), and the groups alternate by construction:
//
//   phase p:   group p & 1        computes tile p          (144 MFMAs per wave, NO barrier inside: all nine weight
//                                                           slices, 72 KB, are LDS-resident for the whole launch)
//              the other group    stores tile p - 1 (epilogue), then DMAs the halo of tile p + 1 into its own buffer
//   one s_barrier per phase.
//
// So the matrix pipe of each SIMD always has one wave in its MFMA loop while the other wave of that SIMD does the
// memory-side work of the neighbouring tile.  LDS: weights 72 KB + two halo images 2 x 41 KB + column sums = 158 KB.
// The epilogue is WAVE-LOCAL (no barrier can be used: the other group is mid-loop): a wave rounds its own 64 pixel x
// 64 channel block through 8.5 KB of ITS part of the group's halo buffer -- the part only that wave DMAs the next halo
// into, so staging and the next DMA need no cross-wave ordering.  Residual / gate pieces of a tile are requested
// before its MFMA loop and arrive under it.  BatchNorm column sums stay in registers across tiles and leave the
// workgroup once.
#pragma once
#include "halo.h"

namespace mn {

constexpr int kPpWeightPieces = 9 * 512;  // nine 64 x 64 fp16 slices
constexpr int kPpWaveRegion = 640;        // pieces of a halo buffer one wave DMAs (and stages its epilogue in)
constexpr int kPpHaloBuf = 2624;          // 4 x 640 + the 32-piece tail (padded to one wave-wide DMA)

// SIDE: epilogue rounds (0, 1, 2) whose residual / gate pieces are requested BEFORE the tile's MFMA loop (32 registers
// each, held through the loop); the remaining rounds' pieces are requested at the start of the epilogue.
template <int SIDE = 1>
static __global__ void __launch_bounds__(512, 1) conv_halo_pp_kernel(GatherGeom g, const half* __restrict__ A,
                                                                     const half* __restrict__ Bw, Epilogue ep, int tiles_x,
                                                                     int tiles_y, int ntiles) {
  constexpr int VEC = 8, NP = 8;
  constexpr int TH = kHaloTH, TW = kHaloTW, HW = TW + 2, HPIX = HW * (TH + 2);
  constexpr int SROW = 68;  // staging row stride in floats: rows 4 banks apart, 16-byte reads of 2 rows x 8 pieces conflict-free
  static_assert(HPIX * NP == 4 * kPpWaveRegion + 32, "halo split");
  static_assert(32 * SROW * 4 <= kPpWaveRegion * 16, "staging fits the wave's region");
  __shared__ piece_t smem[kPpWeightPieces + 2 * kPpHaloBuf + 256];
  float* red = reinterpret_cast<float*>(&smem[kPpWeightPieces + 2 * kPpHaloBuf]);  // [8 waves][64][2]

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = wave >> 2, wq = wave & 3;
  const int l31 = lane & 31, kh = lane >> 5;
  const int G = gridDim.x;
  const int wl = xcd_remap(blockIdx.x, G);
  const int nitems = wl < ntiles ? (ntiles - wl + G - 1) / G : 0;  // tiles wl, wl + G, ...
  piece_t* hbuf = &smem[kPpWeightPieces + grp * kPpHaloBuf];
  const bool flip = g.rsign < 0;

  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(A, (long)g.B * g.Hi * g.Wi * g.C * 2L);
  const __amdgpu_buffer_rsrc_t rsrc_b = make_rsrc(Bw, (long)g.N * g.K * 2L);

  auto tile_coords = [&](int tile, int& b, int& y0, int& x0) {
    const int txi = tile % tiles_x;
    const int tmp = tile / tiles_x;
    const int tyi = tmp % tiles_y;
    b = tmp / tiles_y;
    y0 = tyi * TH;
    x0 = txi * TW;
  };
  // halo of a tile -> this group's buffer: wave wq DMAs pieces [640 wq, 640 wq + 640), wave 0 also the 32-piece tail
  auto issue_halo = [&](int tile) {
    int b, y0, x0;
    tile_coords(tile, b, y0, x0);
    const int oy = y0 + (flip ? g.off_h - 2 : g.off_h), ox = x0 + (flip ? g.off_w - 2 : g.off_w);
    auto one = [&](int q, piece_t* dst) {
      const int hp = q >> 3, pc = q & 7;
      const int hy = hp / HW, hx = hp - hy * HW;
      const int iy = oy + hy, ix = ox + hx;
      const bool ok = hp < HPIX && (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
      const unsigned off = ok ? (unsigned)((((b * g.Hi + iy) * g.Wi + ix) * g.C) * 2) + (unsigned)((pc ^ ((hp >> 1) & 7)) * 16) : ~0u;
      dma16(rsrc_a, off, 0u, dst);
    };
#pragma unroll
    for (int i = 0; i < kPpWaveRegion / 64; ++i) one(wq * kPpWaveRegion + i * 64 + lane, hbuf + wq * kPpWaveRegion + i * 64);
    if (wq == 0) one(4 * kPpWaveRegion + lane, hbuf + 4 * kPpWaveRegion);
  };

  // ---- prologue: all nine weight slices (pass i = tap i: 64 rows x 8 pieces, swizzled on the source side), the first
  // tile of each group
  {
    const int br = t >> 3, pc = t & 7;
    const unsigned boff = br < g.N ? (unsigned)(br * g.K) * 2u + (unsigned)((pc ^ ((br >> 1) & 7)) * 16) : ~0u;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) dma16(rsrc_b, boff, (unsigned)(tap * g.C * 2), &smem[tap * 512 + wave * 64]);
  }
  if (grp < nitems) issue_halo(wl + grp * G);

  floatx16 acc[2][2];
  // residual / gates of the tile in flight: buffer loads (a null tensor has zero records: every load returns zeros,
  // no branch), assigned unconditionally so that nothing is carried around the phase loop
  // Two tensors are requested ahead of the tile's MFMA loop (64 registers): the residual and ONE gate -- res_gate if
  // there is one, else out_gate; with both gates present out_gate is read in the epilogue.
  PieceView<half> rv[2][4], gv[2][4];
  const __amdgpu_buffer_rsrc_t rsrc_res = make_rsrc(ep.res, ep.res ? (long)g.B * g.P * g.Q * ep.ldc * 2L : 0L);
  const __amdgpu_buffer_rsrc_t rsrc_gate = make_rsrc(ep.res_gate, ep.res_gate ? (long)g.B * g.P * g.Q * ep.ldc * 2L : 0L);
  const __amdgpu_buffer_rsrc_t rsrc_ogate = make_rsrc(ep.out_gate, ep.out_gate ? (long)g.B * g.P * g.Q * ep.ldc * 2L : 0L);
  const __amdgpu_buffer_rsrc_t rsrc_g1 = ep.res_gate ? rsrc_gate : rsrc_ogate;
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
  half* out = reinterpret_cast<half*>(ep.out);
  const half* res = reinterpret_cast<const half*>(ep.res);
  const half* gate = reinterpret_cast<const half*>(ep.res_gate);
  const half* ogate = reinterpret_cast<const half*>(ep.out_gate);
  const bool want_stats = ep.stats_accum != nullptr;

  // output piece `ps` of epilogue round i of this wave: pixel and channel piece
  auto side_index = [&](int b, int y0, int x0, int i, int ps, long& idx) -> bool {
    const int id = lane + 64 * ps;
    const int lr = id >> 3, cpc = id & 7;
    const int rl = wq * 64 + i * 32 + lr;
    const int y = y0 + (rl >> 4), x = x0 + (rl & 15);
    idx = (((long)b * g.P + y) * g.Q + x) * ep.ldc + cpc * VEC;
    return y < g.P && x < g.Q;
  };
  auto load_side = [&](int tile, int i) {
    int b, y0, x0;
    tile_coords(tile, b, y0, x0);
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      long idx;
      const bool ok = side_index(b, y0, x0, i, ps, idx);
      const int voff = ok ? (int)(idx * 2) : -1;
      rv[i][ps].p = __builtin_amdgcn_raw_buffer_load_b128(rsrc_res, voff, 0, 0);
      gv[i][ps].p = __builtin_amdgcn_raw_buffer_load_b128(rsrc_g1, voff, 0, 0);
    }
  };

  const int prow = 4 * wq + (l31 >> 4), pcol = l31 & 15;
  auto compute = [&](int tile) {
    if constexpr (SIDE >= 1) load_side(tile, 0);
    if constexpr (SIDE >= 2) load_side(tile, 1);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // tap (r, s) reads halo pixel (prow + dy, pcol + dx), dy = r (forward) or 2 - r (data gradient)
    const int hpf = prow * HW + pcol + (flip ? 2 * HW + 2 : 0);
    const int sg = flip ? -1 : 1;
    // fragments of K-sub-step ks live in register slot ks; a sub-step's reads are issued two sub-steps (8 MFMAs, 256
    // matrix-pipe cycles) ahead of its MFMAs.  The tap loop is a real loop: fully unrolled, the 144 fragment addresses
    // get hoisted and spill.
    PieceView<half> fa[4][2], fb[4][2];
    auto load_frags = [&](int tap, int ks) {
      const int piece = ks * 2 + kh;
      const int hp0 = hpf + sg * ((tap / 3) * HW + tap % 3);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int hp = hp0 + i * 2 * HW;
        fa[ks][i].p = hbuf[hp * NP + (piece ^ ((hp >> 1) & 7))];
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = j * 32 + l31;
        fb[ks][j].p = smem[tap * 512 + row * NP + (piece ^ ((row >> 1) & 7))];
      }
    };
    load_frags(0, 0);
    load_frags(0, 1);
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks < 2)
          load_frags(tap, ks + 2);
        else if (tap + 1 < 9)
          load_frags(tap + 1, ks - 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) mma_piece<half>(fa[ks][i], fb[ks][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // wave-local epilogue of the tile whose accumulators this wave holds
  auto epilogue = [&](int tile) {
    int b, y0, x0;
    tile_coords(tile, b, y0, x0);
    float* stage = reinterpret_cast<float*>(hbuf + wq * kPpWaveRegion);  // [32 rows][SROW]
    if constexpr (SIDE < 1) load_side(tile, 0);
    if constexpr (SIDE < 2) load_side(tile, 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int lc = j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][j][r] * ep.alpha;
          if (ep.relu & 1) v = fmaxf(v, 0.f);
          const int rt = (r & 3) + 8 * (r >> 2) + 4 * kh;  // row inside the 32-row MFMA tile
          if (want_stats) {
            const int rl = wq * 64 + i * 32 + rt;
            const bool ok = y0 + (rl >> 4) < g.P && x0 + (rl & 15) < g.Q;
            const float vs = ok ? v : 0.f;
            s1[j] += vs;
            s2[j] += vs * vs;
          }
          stage[rt * SROW + lc] = v;
        }
      }
      __builtin_amdgcn_wave_barrier();  // same wave writes and reads the block (in-order LDS; the emulator's fibers meet here)
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        long idx;
        const bool ok = side_index(b, y0, x0, i, ps, idx);
        const int id = lane + 64 * ps;
        const int lr = id >> 3, cpc = id & 7;
        float v[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {
          const floatx4 f = *reinterpret_cast<const floatx4*>(&stage[lr * SROW + cpc * VEC + e]);
          v[e] = f[0];
          v[e + 1] = f[1];
          v[e + 2] = f[2];
          v[e + 3] = f[3];
        }
        if (ok) {
          if (res) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              float xr = (float)rv[i][ps].e[e];
              if (gate && !((float)gv[i][ps].e[e] > 0.f)) xr = 0.f;
              v[e] += xr;
            }
          }
          if (ogate) {
            PieceView<half> og = gv[i][ps];
            if (gate) og.p = __builtin_amdgcn_raw_buffer_load_b128(rsrc_ogate, (int)(idx * 2), 0, 0);
#pragma unroll
            for (int e = 0; e < VEC; ++e)
              if (!((float)og.e[e] > 0.f)) v[e] = 0.f;
          }
          PieceView<half> o;
#pragma unroll
          for (int e = 0; e < VEC; ++e) o.e[e] = (half)v[e];
          *reinterpret_cast<piece_t*>(out + idx) = o.p;
        }
      }
      __builtin_amdgcn_wave_barrier();  // reads done before round 1 (or the next halo's DMA) overwrites the block
    }
  };

  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();  // weights and both first halos visible
  for (int p = 0; p <= nitems; ++p) {
    if ((p & 1) == grp) {
      if (p < nitems) compute(wl + p * G);
    } else if (p >= 1) {
      epilogue(wl + (p - 1) * G);
      if (p + 1 < nitems) issue_halo(wl + (p + 1) * G);
    }
    wait_vmcnt<0>();               // this wave's halo DMAs (and stores)
    __builtin_amdgcn_s_barrier();  // the computing group is done reading its buffer; the other group's halo has landed
  }

  if (want_stats) {
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
      for (int w = 0; w < 8; ++w) {
        sa += red[(w * 64 + t) * 2 + 0];
        sb += red[(w * 64 + t) * 2 + 1];
      }
      double* row = ep.stats_accum + (long)((int)blockIdx.x % ep.stats_rows) * 2 * g.N;
      atomicAdd(row + t, (double)sa);
      atomicAdd(row + g.N + t, (double)sb);
    }
  }
}

// the shapes the persistent kernel covers: halo.h's with exactly 64 output channels, statistics (if any) through the
// fp64 accumulator rows, no bias
inline bool conv_halo_pp_applies(const GatherGeom& g, const Epilogue& ep) {
  return conv_halo_applies(g) && g.N == 64 && ep.stats == nullptr && ep.bias == nullptr && !ep.om_on &&
         (ep.stats_accum == nullptr || ep.stats_rows > 0);
}

// wgs: persistent workgroups (0 = one per CU, or MN_HALO_PP_WGS)
inline void launch_conv_halo_pp(const GatherGeom& g, const half* A, const half* Bw, const Epilogue& ep, hipStream_t stream,
                                int wgs_arg = 0) {
  const int tx = cdiv(g.Q, kHaloTW), ty = cdiv(g.P, kHaloTH);
  const int ntiles = g.B * tx * ty;
  static const int wgs_env = getenv("MN_HALO_PP_WGS") ? atoi(getenv("MN_HALO_PP_WGS")) : 256;  // one per CU
  const int wgs = wgs_arg > 0 ? wgs_arg : wgs_env;
  static const int side = getenv("MN_HALO_PP_SIDE") ? atoi(getenv("MN_HALO_PP_SIDE")) : 1;  // 2 spills 23 registers
  const dim3 grid(ntiles < wgs ? ntiles : wgs);
  if (side >= 2)
    hipLaunchKernelGGL(conv_halo_pp_kernel<2>, grid, dim3(512), 0, stream, g, A, Bw, ep, tx, ty, ntiles);
  else if (side == 1)
    hipLaunchKernelGGL(conv_halo_pp_kernel<1>, grid, dim3(512), 0, stream, g, A, Bw, ep, tx, ty, ntiles);
  else
    hipLaunchKernelGGL(conv_halo_pp_kernel<0>, grid, dim3(512), 0, stream, g, A, Bw, ep, tx, ty, ntiles);
}

}  // namespace mn
