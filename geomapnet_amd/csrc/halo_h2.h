// 3x3 stride-1 convolution of 64 -> 64 channel h2 tensors (ResNet layer1, FORWARD pass of the fp16x2 / fp16x2m modes) with the
// WEIGHTS IN REGISTERS (round 6).
//
//   y[b, y, x, n] = sum_{r, s, c} a[b, y + r - 1, x + s - 1, c] * w[n][(r*3 + s)*64 + c]       a, w: h2 (fp16 hi + lo), y: fp32
//
// Why: the chunk-resident kernel (igemm_halo.h, 64-column shape) re-streams the layer's WHOLE weight matrix -- 147 KB in h2 -- through
// LDS-DMA for every 192-row tile and sits at the rate that path sustains: 278 us per launch at 192 images for 93 us of MFMAs, the
// worst matrix-pipe occupancy of the step (0.38).  halo_pp.h's answer for fp16 -- all nine weight slices LDS-resident beside two
// input halos -- does not fit: h2 weights (147 KB) + one 18 x 18-pixel h2 halo (83 KB) exceed a CU's 160 KB.  But a CU's REGISTER
// files hold 512 KB, and a weight fragment is an MFMA operand that never changes:
//
//   * 4 waves = (32-channel half of the outputs) x (pixel half of a tile), ONE wave per SIMD, up to 512 registers each.  A wave keeps
//     the hi and lo halves of ITS 32 output channels' weights -- 9 taps x 64 input channels x 32 outputs x 2 halves = 72 KB = 72
//     16-byte pieces per lane = 288 registers -- as ready-made A operands for the whole launch (loaded once per persistent workgroup:
//     256 x 147 KB = 38 MB per launch instead of 5504 x 147 KB = 810 MB of LDS-DMA);
//   * LDS holds only the input: two halos of an 8 x 16-pixel tile (10 x 18 pixels x 256 B = 45 KB each), the next tile's requested
//     by DMA while this one is contracted.  Pixels are the B operand: lane l of a 32-pixel block (2 tile rows x 16 columns) reads
//     halo pixel (row + dy, col + dx) as an address OFFSET -- the nine taps of a K sub-step are nine immediates;
//   * a product is the h2 kernels' three MFMAs (lo*hi + hi*lo + hi*hi) on v_mfma_f32_32x32x16_f16; weights = A, pixels = B, so a
//     lane's accumulator holds 4 consecutive output channels of ONE pixel and the fp32 result leaves in 16-byte stores with no
//     staging (halo_pp.h's epilogue); BatchNorm column sums are per-lane partial sums over all tiles, folded once per workgroup.
//
// LDS image: pixel-major, 16 pieces of 16 bytes per halo pixel (32-channel group g: pieces 8g .. 8g+3 = hi halves, 8g+4 .. 8g+7 = lo
// halves), the piece slot XOR-swizzled by the halo COLUMN (& 15) on the source side of the DMA and in the fragment address: the 16
// lanes of a ds_read_b128 group sit in 16 different columns (two rows of a 32-pixel block share the column set), so every group
// covers all 64 banks.
#pragma once
#include "igemm.h"

namespace mn {

constexpr int kH2TH = 8, kH2TW = 16;                              // output tile: 128 pixels = 4 MFMA pixel blocks of 2 rows x 16 columns
constexpr int kH2HW = kH2TW + 2, kH2HPIX = (kH2TH + 2) * kH2HW;   // halo: 10 x 18 = 180 pixels
constexpr int kH2Pieces = kH2HPIX * 16;                           // 2880 pieces = 46 080 bytes per halo image
constexpr int kH2Passes = (kH2Pieces + 255) / 256;                // 12 DMA passes of 256 lanes (the last one: wave 0 only)

// g: the convolution's own geometry (C = 64 channels, NOT the doubled h2 view launch_igemm_h2 hands to the fp16-typed kernels)
inline bool conv_halo_h2_applies(const GatherGeom& g, const Epilogue& ep) {
  return g.R == 3 && g.S == 3 && g.C == 64 && g.N == 64 && g.K == 9 * 64 && g.mul_p == 1 && g.mul_q == 1 && g.div == 1 &&
         g.P == g.Hi && g.Q == g.Wi && g.rsign == 1 && g.ssign == 1 && g.off_h == -1 && g.off_w == -1 && !g.bt_on && g.ldb == 0 &&
         ep.ldc == 64 && ep.stats == nullptr && ep.bias == nullptr && ep.relu == 0 && ep.res == nullptr &&
         ep.res_gate == nullptr && ep.out_gate == nullptr && !ep.om_on && (ep.stats_accum == nullptr || ep.stats_rows > 0) &&
         (long)g.B * g.Hi * g.Wi * 256L < 0x7ffffff0l;  // (masked lanes use offset 2^31 / all ones)
}

// ABL (timing experiments only, ablation build, results are wrong): bit 0 = halo DMA only for the first tile, bit 1 = no fragment
// reads, bit 2 = no stores, bit 3 = no MFMAs, bit 4 = no per-tile barrier.
// PD: K sub-steps a B fragment is requested ahead of its MFMAs.  SG: 0 = a step's reads and its MFMAs as two fenced blocks (first
// version), 1 = the scheduler is asked to interleave them (sched_group_barrier: it moves every read right in front of its use),
// 2 = interleaved by hand: one read (or one DMA of the next halo) behind each MFMA.
template <bool STATS, int ABL = 0, int PD = 1, int SG = 2>
static __global__ void __launch_bounds__(256, 1) conv_halo_h2_kernel(GatherGeom g, const half* __restrict__ A,
                                                                     const half* __restrict__ Bw, Epilogue ep, int tiles_x,
                                                                     int tiles_y, int ntiles) {
  constexpr int TH = kH2TH, TW = kH2TW, HW = kH2HW;
  constexpr unsigned kOob = 0x80000000u;
  __shared__ piece_t smem[2 * kH2Pieces];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ch = wave & 1, ph = wave >> 1;  // output-channel half, pixel half of the tile
  const int l31 = lane & 31, kh = lane >> 5;
  const int G = gridDim.x;
  const int wl = xcd_remap(blockIdx.x, G);
  const int nitems = wl < ntiles ? (ntiles - wl + G - 1) / G : 0;  // tiles wl, wl + G, ...
  const int gP = g.P, gQ = g.Q, gHi = g.Hi, gWi = g.Wi;
  const float alpha = ep.alpha;

  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(A, (long)g.B * g.Hi * g.Wi * 256L);
  const __amdgpu_buffer_rsrc_t rsrc_out = make_rsrc(ep.out, (long)g.B * g.P * g.Q * 64L * 4L);

  // ---- this wave's weights: A operands of v_mfma_f32_32x32x16_f16 (lane: output channel ch * 32 + (lane & 31), k = 8 * (lane >> 5)
  //      .. + 7 of a 16-k sub-step).  Piece i = tap * 4 + g * 2 + s: tap, 32-channel group g, 16-k half s of the group.
  PieceView<half> wh[36], wlo[36];
  {
    const half* row = Bw + (long)(ch * 32 + l31) * (2 * 9 * 64);  // h2 row: per 32 k 32 hi halves then 32 lo halves
#pragma unroll
    for (int i = 0; i < 36; ++i) {
      const int tap = i >> 2, gg = (i >> 1) & 1, s = i & 1;
      const half* p = row + (tap * 2 + gg) * 64 + (2 * s + kh) * 8;
      wh[i].p = *reinterpret_cast<const piece_t*>(p);
      wlo[i].p = *reinterpret_cast<const piece_t*>(p + 32);
    }
  }

  auto tile_coords = [&](int tile, int& b, int& y0, int& x0) __attribute__((always_inline)) {
    const int txi = tile % tiles_x;
    const int tmp = tile / tiles_x;
    const int tyi = tmp % tiles_y;
    b = tmp / tiles_y;
    y0 = tyi * TH;
    x0 = txi * TW;
  };
  // halo of a tile -> buffer `buf`: LDS slot sidx = pass * 256 + t (a DMA writes lane l of a wave to 16 * l behind the wave's base)
  // holds piece (slot ^ (halo column & 15)) of halo pixel sidx >> 4.  (The slot geometry is recomputed per tile -- a dozen VALU
  // operations per DMA, ~150 per tile beside 216 MFMAs -- rather than held: the weights leave no registers to spare.)
  // (one pass = one wave-wide DMA instruction per wave; SG = 2 issues the passes of the next tile's halo between the MFMAs of this one)
  int nb = 0, ny0 = 0, nx0 = 0;  // coordinates of the tile whose halo is being requested
  auto halo_pass = [&](int i, int buf) __attribute__((always_inline)) {
    if (i * 256 + wave * 64 >= kH2Pieces) return;  // wave-uniform: the last pass has 64 slots
    const int oy = ny0 - 1, ox = nx0 - 1;
    const int base = ((nb * gHi + oy) * gWi + ox) * 256;  // (may be negative for the first row / column: only added where valid)
    const int sidx = i * 256 + t;
    const int hp = sidx >> 4, slot = sidx & 15;
    const int hy = hp / HW, hx = hp - hy * HW;
    const int iy = oy + hy, ix = ox + hx;
    const bool ok = (unsigned)iy < (unsigned)gHi && (unsigned)ix < (unsigned)gWi;
    const int rel = (hy * gWi + hx) * 256 + ((slot ^ (hx & 15)) << 4);
    dma16(rsrc_a, ok ? (unsigned)(base + rel) : ~0u, 0u, &smem[buf * kH2Pieces + i * 256 + wave * 64]);
  };
  auto issue_halo = [&](int tile, int buf) __attribute__((always_inline)) {
    tile_coords(tile, nb, ny0, nx0);
#pragma unroll
    for (int i = 0; i < kH2Passes; ++i) halo_pass(i, buf);
  };

  floatx16 acc[2];
  float st1[STATS ? 16 : 1], st2[STATS ? 16 : 1];  // per-lane column sums (4 q x 4 channels) over this lane's pixels
  if constexpr (STATS) {
#pragma unroll
    for (int r = 0; r < 16; ++r) st1[r] = st2[r] = 0.f;
  }

  // fragment addressing: pixel block bb of this wave = tile rows (ph * 2 + bb) * 2, + 1; lane = pixel (l31 >> 4, l31 & 15)
  const int col = l31 & 15;
  unsigned pbase[2];  // byte offset of the block's tap-(0,0) halo pixel
#pragma unroll
  for (int bb = 0; bb < 2; ++bb) pbase[bb] = (unsigned)((((ph * 2 + bb) * 2 + (l31 >> 4)) * HW + col) * 256);
  unsigned xk[3];  // ((lane's k half) ^ (halo column & 15)) << 4 for dx = 0, 1, 2: the lane-dependent part of the swizzled slot
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) xk[dx] = (unsigned)((kh ^ ((col + dx) & 15)) << 4);

  auto compute = [&](int buf, bool next) __attribute__((always_inline)) {
    const char* hb = reinterpret_cast<const char*>(&smem[buf * kH2Pieces]);
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[bb][r] = 0.f;
    // Step j = ((dx * 2 + g) * 2 + s) * 3 + dy: the three taps of a halo COLUMN offset are consecutive steps, so the four fragment
    // addresses of a (dx, g, s) group -- two pixel blocks x (hi, lo) -- are computed once and dy is the immediate offset of the read
    // (any K order is valid as long as weights and pixels agree).  B fragments are requested PD steps ahead of their MFMAs.
    constexpr int NS = 36;
    PieceView<half> fh[PD + 1][2], fl[PD + 1][2];
    if constexpr ((ABL & 2) != 0) {
#pragma unroll
      for (int k = 0; k <= PD; ++k)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) fh[k][bb].p = fl[k][bb].p = zero_piece();
    }
    // fragment k of step j: 0 = hi of block 0, 1 = lo of block 0, 2 = hi of block 1, 3 = lo of block 1
    auto load_frag = [&](int j, int k) __attribute__((always_inline)) {
      if constexpr ((ABL & 2) != 0) return;
      const int dy = j % 3, grp = j / 3;
      const int s = grp & 1, gg = (grp >> 1) & 1, dx = grp >> 2;
      const unsigned qc = (unsigned)((gg * 8 + 2 * s) << 4);  // even slot of the hi piece; the lo piece is 4 slots further
      const int bb = k >> 1;
      const char* pp = hb + (pbase[bb] + (unsigned)(dx * 256) + ((qc + ((k & 1) ? 64u : 0u)) ^ xk[dx]));
      const piece_t v = *reinterpret_cast<const piece_t*>(pp + dy * HW * 256);
      if (k & 1)
        fl[j % (PD + 1)][bb].p = v;
      else
        fh[j % (PD + 1)][bb].p = v;
    };
    auto load_frags = [&](int j) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < 4; ++k) load_frag(j, k);
    };
#pragma unroll
    for (int j = 0; j < PD; ++j) load_frags(j);
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int dy = j % 3, grp = j / 3;
      const int i = (dy * 3 + (grp >> 2)) * 4 + (grp & 3);  // weight piece: tap * 4 + g * 2 + s
      const int sl = j % (PD + 1);
      if constexpr (SG == 2 && (ABL & 8) == 0) {
        // ONE wave per SIMD: whatever is not an MFMA must issue in the shadow of one (a 32-cycle MFMA hides about five single-issue
        // instructions, MI355X_MICROARCH.md).  The first version issued a step's four fragment reads as a block between two blocks of
        // six MFMAs: the matrix pipe drained behind every block (SQ counters: MFMA busy 0.49, the wave ACTIVE a third of its cycles).
        // Here every MFMA is followed by at most one fragment read of step j + PD and, every third step, one DMA of the next halo.
        const bool more = j + PD < NS;
        mma_piece<half>(wlo[i], fh[sl][0], acc[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (more) load_frag(j + PD, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma_piece<half>(wlo[i], fh[sl][1], acc[1]);
        __builtin_amdgcn_sched_barrier(0);
        if (more) load_frag(j + PD, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_piece<half>(wh[i], fl[sl][0], acc[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (more) load_frag(j + PD, 2);
        __builtin_amdgcn_sched_barrier(0);
        mma_piece<half>(wh[i], fl[sl][1], acc[1]);
        __builtin_amdgcn_sched_barrier(0);
        if (more) load_frag(j + PD, 3);
        __builtin_amdgcn_sched_barrier(0);
        mma_piece<half>(wh[i], fh[sl][0], acc[0]);
        __builtin_amdgcn_sched_barrier(0);
        if ((ABL & 1) == 0 && next && j % 3 == 0) halo_pass(j / 3, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_piece<half>(wh[i], fh[sl][1], acc[1]);
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      if (j + PD < NS) load_frags(j + PD);
      if constexpr (SG == 0) __builtin_amdgcn_sched_barrier(0);
      // weights = A operand (rows), pixels = B (columns): lo*hi + hi*lo + hi*hi; the two pixel blocks alternate so that no MFMA
      // waits for the accumulator of the one before it
      if constexpr ((ABL & 8) == 0) {
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) mma_piece<half>(wlo[i], fh[sl][bb], acc[bb]);
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) mma_piece<half>(wh[i], fl[sl][bb], acc[bb]);
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) mma_piece<half>(wh[i], fh[sl][bb], acc[bb]);
      } else {
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) asm volatile("" ::"v"(fh[sl][bb].p), "v"(fl[sl][bb].p), "v"(wh[i].p), "v"(wlo[i].p));
      }
      if constexpr (SG != 1) {
        __builtin_amdgcn_sched_barrier(0);
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            // one MFMA
          if (k < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one LDS read
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);            // up to two VALU (address arithmetic, AGPR -> VGPR moves)
        }
      }
    }
  };

  // fp32 output straight from the accumulators: lane = pixel l31 of block bb, register r = channel ch * 32 + 8 (r >> 2) + 4 kh + (r & 3)
  auto store = [&](int tile) __attribute__((always_inline)) {
    int b, y0, x0;
    tile_coords(tile, b, y0, x0);
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      const int y = y0 + (ph * 2 + bb) * 2 + (l31 >> 4), x = x0 + col;
      const bool ok = y < gP && x < gQ;
      const unsigned voff = ok ? (unsigned)((((b * gP + y) * gQ + x) * 64 + ch * 32 + 4 * kh) * 4) : kOob;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        PieceView<float> o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc[bb][q * 4 + e] * alpha;
          o.e[e] = v;
          if constexpr (STATS) {
            const float vs = ok ? v : 0.f;
            st1[q * 4 + e] += vs;
            st2[q * 4 + e] += vs * vs;
          }
        }
        if constexpr ((ABL & 4) == 0) {
          __builtin_amdgcn_raw_buffer_store_b128(o.p, rsrc_out, (int)voff, q * 32, 0);
        } else {
          asm volatile("" ::"v"(o.p));
        }
      }
    }
  };

  if (nitems > 0) issue_halo(wl, 0);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();  // the first halo is visible to every wave
  for (int p = 0; p < nitems; ++p) {
    const int buf = p & 1;
    const bool next = p + 1 < nitems;
    if (next) tile_coords(wl + (p + 1) * G, nb, ny0, nx0);
    if constexpr (SG != 2 || (ABL & 8) != 0) {
      if ((ABL & 1) == 0 && next) issue_halo(wl + (p + 1) * G, buf ^ 1);  // in flight under this tile's MFMAs
    }
    compute(buf, next);
    // this wave's share of the next halo has landed -- waited for BEFORE this tile's stores are issued, so that the stores stay in
    // flight across the barrier instead of being waited for (the first version waited for both: 47 us of a 300 us launch)
    wait_vmcnt<0>();
    store(wl + p * G);
    if constexpr ((ABL & 16) == 0) __builtin_amdgcn_s_barrier();  // every wave is done reading `buf`; every wave's share of the next halo is visible
  }
  wait_vmcnt<0>();

  if constexpr (STATS) {
    if (ep.stats_accum) {
      // fold the per-lane sums over the 32 pixel lanes of each half-wave (halo_pp.h: the same precision class), then the two pixel
      // halves of the workgroup, then one fp64 atomic per channel and sum into this workgroup's accumulator row
      float* red = reinterpret_cast<float*>(&smem[0]);  // [4 waves][32 channels][2]; every wave is past its last fragment read
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float a = st1[r], bsum = st2[r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
          a += __shfl_xor(a, o);
          bsum += __shfl_xor(bsum, o);
        }
        if (l31 == 0) {
          const int n = 8 * (r >> 2) + 4 * kh + (r & 3);
          red[(wave * 32 + n) * 2 + 0] = a;
          red[(wave * 32 + n) * 2 + 1] = bsum;
        }
      }
      __syncthreads();
      if (t < 64) {
        const int c2 = t >> 5, n = t & 31;  // channel t: waves c2 and c2 + 2 hold its two pixel halves
        const float sa = red[(c2 * 32 + n) * 2 + 0] + red[((c2 + 2) * 32 + n) * 2 + 0];
        const float sb = red[(c2 * 32 + n) * 2 + 1] + red[((c2 + 2) * 32 + n) * 2 + 1];
        double* row = ep.stats_accum + (long)((int)blockIdx.x % ep.stats_rows) * 2 * 64;
        atomicAdd(row + t, (double)sa);
        atomicAdd(row + 64 + t, (double)sb);
      }
    }
  }
}

// (Measured and removed, round 6: an 8-wave form -- the reduction split over two waves per SIMD, (32 outputs) x (pixel half) x (input
// group), 144 weight registers per wave, partial sums exchanged through 2 x 32 KB of LDS, one barrier per tile -- on the theory that
// with two waves per SIMD one wave's reads sit beside the other's MFMAs: 293-299 us per launch against this kernel's 261, whole step
// 18.94 against 18.55 ms (profiles/r06/c12_*).  The ablations of both forms say the same thing: MFMAs alone ~100 us; fragment reads
// +100, halo DMA +30-60, stores +40, and they ADD in either form although the LDS array is busy a sixth of the time and conflict-free
// (SQ_LDS_BANK_CONFLICT = 0, SQ_LDS_IDX_ACTIVE 4.15 cycles per read): what a memory instruction costs here is its ISSUE, ~40 cycles
// of a CU's wave time per ds_read_b128, and a second wave per SIMD does not hide it.)
// persistent workgroups: one per CU (MN_HALO_H2_WGS overrides), at most one per tile
inline int conv_halo_h2_grid(const GatherGeom& g) {
  const int ntiles = g.B * cdiv(g.P, kH2TH) * cdiv(g.Q, kH2TW);
  static const int wgs_env = getenv("MN_HALO_H2_WGS") ? atoi(getenv("MN_HALO_H2_WGS")) : 256;
  return ntiles < wgs_env ? ntiles : wgs_env;
}
inline void launch_conv_halo_h2(const GatherGeom& g, const half* A, const half* Bw, const Epilogue& ep, hipStream_t stream) {
  const int tx = cdiv(g.Q, kH2TW), ty = cdiv(g.P, kH2TH);
  const int ntiles = g.B * tx * ty;
  const dim3 grid(conv_halo_h2_grid(g));
#ifdef MN_ABLATION_BUILD
  static const int abl = getenv("MN_HALO_H2_ABLATE") ? atoi(getenv("MN_HALO_H2_ABLATE")) : 0;
  {
#define H2_CASE(V_) case V_: hipLaunchKernelGGL((conv_halo_h2_kernel<true, V_, 1, 2>), grid, dim3(256), 0, stream, g, A, Bw, ep, tx, ty, ntiles); return
  switch (abl) {
    H2_CASE(1); H2_CASE(2); H2_CASE(4); H2_CASE(8); H2_CASE(16); H2_CASE(5); H2_CASE(7); H2_CASE(10); H2_CASE(23); H2_CASE(31);
    case 100: hipLaunchKernelGGL((conv_halo_h2_kernel<true, 0, 2, 2>), grid, dim3(256), 0, stream, g, A, Bw, ep, tx, ty, ntiles); return;  // not ablations:
    case 110: hipLaunchKernelGGL((conv_halo_h2_kernel<true, 0, 1, 0>), grid, dim3(256), 0, stream, g, A, Bw, ep, tx, ty, ntiles); return;  // read-ahead, SG
    case 111: hipLaunchKernelGGL((conv_halo_h2_kernel<true, 0, 1, 1>), grid, dim3(256), 0, stream, g, A, Bw, ep, tx, ty, ntiles); return;
    default: break;
  }
  }
#undef H2_CASE
#endif
  if (ep.stats_accum)
    hipLaunchKernelGGL(conv_halo_h2_kernel<true>, grid, dim3(256), 0, stream, g, A, Bw, ep, tx, ty, ntiles);
  else
    hipLaunchKernelGGL(conv_halo_h2_kernel<false>, grid, dim3(256), 0, stream, g, A, Bw, ep, tx, ty, ntiles);
}

}  // namespace mn
