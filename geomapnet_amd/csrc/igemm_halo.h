// The 12-wave 288-row implicit-GEMM tile of igemm.h for 3x3 stride-1 "same" convolutions of fp16 tensors, with the A operand
// staged per 64-channel CHUNK instead of per K-step (default for layers 2-4 since round 2: MN_IGEMM_HALO=2; 0 restores
// igemm.h; measured layer2 114 -> 104 us, layer3 93 -> 81, layer4 112 -> 87).
//
// igemm.h moves 288 A rows + 256 B rows = 69.6 KB through LDS-DMA per K-step (one tap of one 64-channel chunk); the nine
// taps of a chunk fetch nine shifted copies of the same pixels.  Here a chunk's A rows are fetched ONCE, with a halo:
// in the flattened (b, y, x) pixel order a tap (dy, dx) is the row shift dy*W + dx, so the image holds rows
// [m0 - (W+1), m0 + 288 + (W+1)) and a K-step reads its A fragments at row offset (W+1) + dy*W + dx.  What differs per
// row is validity (image borders, rows past M): a lane whose (row, tap) is outside the image reads a 16-byte zero slot
// instead.  DMA per K-step: 32 KB (B slice) + 44 KB / 9 (A image) = 36.9 KB instead of 69.6 KB.
//
// LDS (158 of 160 KiB for the 256-column shape): two A images of kAH rows (the next chunk's image is fetched, one DMA pass per
// K-step, while the current one is read), a two-slot B ring of 256 rows, the zero slot, BatchNorm scratch.
// kAH >= 288 + 2 (W + 1) bounds the image width W.
// Synchronisation is igemm.h's: one `s_waitcnt vmcnt(0)` + raw barrier per K-step; every DMA is issued after a
// barrier (behind the K-step's first MFMA group, see DP) and waited for at the next one, so image pass p of chunk c+1 (issued in K-step p of chunk c) has landed eight
// barriers before it is read, and the image it overwrites was last read in chunk c-1.
#pragma once
#include "igemm.h"

namespace mn {

// Two shapes: <256, 352> (N in 256-column tiles, W <= 31: layer3) and <128, 384> (128-column tiles, W <= 47, 131 KB of
// LDS: layer2 with its 43-pixel rows, and layer4, whose 16 896 rows give 59 x 4 = 236 tiles = one round of the chip).
// The B ring has TWO slots: every K-step waits for the slice it requested one K-step earlier.  (Measured and removed: three
// slots with counted waits -- two slices of DMA instructions in flight -- are SLOWER for the 128-column shape, layer2 109.8
// vs 105.1 us, layer4 92.0 vs 85.8 us, whole step 15.36 vs 15.27 ms: request latency is not what bounds these launches.)
// The epilogue stages the accumulators through LDS for 16-byte stores.  (Measured and removed in round 3: a register epilogue
// -- MFMA operands swapped so that a lane holds 4 consecutive channels of one pixel, 8-byte buffer stores, residual / gate
// loads in the same pattern, no staging and no barriers, as halo_pp.h does -- is SLOWER: layer2 forward 131 vs 103 us,
// +residual 150 vs 120, layer3 89 vs 80, whole step 15.1 vs 14.6 ms; a store instruction then touches 32 pixel rows x 16
// bytes instead of 4 rows x 256 bytes.  profiles/r03/c11_igemm_halo_register_epilogue.txt.  What the staged epilogue
// costs: profiles/r03/c10_igemm_halo_prologue_kloop_epilogue.txt -- 18 of 103 us at layer2, 12 of 80 at layer3.)
// DP: where in a K-step the LDS-DMA instructions of the next step are issued: 0 = right after the barrier (rounds 1-2), 1 / 2 =
// after the first / second of the four MFMA groups.  Measured (profiles/r03/c13_igemm_halo_dma_position.txt, two repeats):
// layer2 105.7 / 100.6 / 104.0 us, layer3 83.1 / 81.8 / 83.5, layer4 88.3 / 85.5 / 90.0 -- right after the barrier the requests
// compete with every wave's first fragment reads; DP = 1 is what ships.
// ABL (timing experiments only, ablation build, MN_HALO_ABLATE; results are wrong): bit 0 = no epilogue (no staging, stores,
// residual / gate loads, statistics), bit 1 = K loop cut to its first K-step, bit 2 = no DMA in the K loop, bit 3 = fragment
// reads only in the first K-step, bit 4 = no MFMAs.
// H2 (round 4): h2 operands (common.h MMA_H2) -- g.C / g.K count the 2C fp16 channels of the pair layout, a 64-"channel" chunk
// is one 32-channel group (hi halves | lo halves), a K-step = three MFMAs per tile pair and 16-k half of the group, the output
// (and the residual) fp32, the gates h2.  Same LDS-DMA traffic per K-step as the fp16 kernel for 1.5x its MFMAs.
// WM x WN (round 4): the wave grid, wave tiles of 96 x (BN / WN).  3 x 4 = the 12-wave 288-row tile of rounds 1-3; 4 x 2 with BN =
// 128 = an 8-wave 384-row tile of 96 x 64 wave tiles for the 128-column layers in the fp16x2 mode: 0.55 instead of 0.89 fragment
// reads per MFMA and, at two waves per SIMD, registers for both halves' fragments.
// A1 (round 4, layer1 in the fp16x2 mode: 64 columns, rows of 86 pixels): ONE A image instead of two -- the next chunk's image is
// requested when the last K-step of the current chunk has been read (one barrier, its latency exposed once per chunk) -- so that
// a 192-row tile of 4 waves needs 63 KB of LDS and TWO workgroups share a CU: each one's prologue, image reload and epilogue run
// under the other's MFMAs.
// (Round 5 built an FBN variant -- the consumer-side BatchNorm fusion the reviews of rounds 2-4 asked for: the producing layer's
// normalise + ReLU applied to every A fragment after its LDS read -- and measured it: layer3 76 -> 109 us, layer4 81 -> 146 us per
// convolution against the 7-11 us an elementwise pass over the same activation takes.  Removed from the library in round 6; the
// record is profiles/r05/c3_consumer_side_batchnorm_fusion_per_launch.txt and DESIGN.md's history.)
// OCC: workgroups per CU the shape is built for (LDS budget, register cap).
// (Measured and removed, round 5: BIG single-image tiles -- A1 with OCC = 1, 576 rows x 128 / 64 columns of 12 waves, 6 x 2 of
// 96 x 64 / 96 x 32, and 384 rows of 8 waves for layer1.  The idea: the B slices of a tile are the layer's WHOLE weight matrix,
// re-streamed through LDS-DMA by every tile -- for layer2 590 KB (h2) / 295 KB (fp16) per 192-row tile against 147 / 74 KB of A
// image -- so three times the rows per tile cut the DMA bytes per MFMA by 40-58 %.  Per launch at 192 images
// (profiles/r05/c4_big_single_image_tiles_per_launch.txt): fp16 layer2 90.2 -> 104.3 us forward, 87.1 -> 93.2 data gradient; h2
// layer2 213 -> 224, h2 layer1 305 -> 331 (576 rows) / 341 (384 rows); layer4's 120 tiles are half a round, 83 -> 127 us.  DMA
// bytes are not what bounds these launches; a single image exposes its reload once per chunk with nothing beside it on the CU.)
// (Measured and removed, round 5: "BG" -- the B operand never passes through LDS: a lane's MFMA B fragment is 16 contiguous bytes of
// the [N][K] weight matrix, so each wave loaded its fragments straight from L2 into a four-slot register ring three 16-k sub-steps
// ahead; no B slice DMA, no B fragment reads, and only ONE barrier per 64-channel chunk instead of one per K-step.  Parity-green in
// the emulator; per launch at 192 images layer2 90 -> 129 us, layer4 82 -> 115, layer3 77 -> 144 (that shape also spilled):
// a 16-byte-per-lane load whose 64 lanes touch 32 different cache lines runs at the texture addresser's rate, far below what the
// LDS-DMA + ds_read_b128 path delivers.  The per-K-step barrier is cheaper than any way around it that was tried in five rounds.
// profiles/r05/c8_weights_from_l2_into_registers_per_launch.txt)
// Q (round 5, with H2: the fp16x2q mode's forward convolutions): h2q operands (common.h MMA_H2Q) -- per K-step and tile pair ONE
// v_mfma_scale_f32_32x32x64_f8f6f4 contracts both cross terms from the group's fp8 planes ([lo8 | hi8] of A against [hi8 | lo8] of B:
// a lane's operand is piece 4 + half followed by piece 6 + half; scale bytes are constants of the lane half), then the two fp16
// MFMAs of hi*hi: 128 instead of 192 matrix-pipe cycles per K-step and tile pair, the same LDS-DMA bytes and fragment reads.
template <int BN, int kAH, int ABL = 0, int DP = 1, bool H2 = false, int WM = 3, int WN = 4, bool A1 = false,
          int OCC = (A1 ? 2 : 1), bool Q = false>
static __global__ void __launch_bounds__(WM* WN * 64, OCC * WM* WN / 4) igemm_halo_kernel(GatherGeom g, const half* __restrict__ A,
                                                                   const half* __restrict__ Bw, Epilogue ep, int grid_n,
                                                                   RowDiv rd) {
  constexpr int VEC = 8, NP = 8, TM = 3, TN = BN / (WN * 32), NT = WM * WN * 64;
  constexpr int BM = WM * 96, WTM = 96, WTN = BN / WN, RPP = NT / NP;  // rows per DMA pass
  static_assert(TN == 1 || TN == 2, "wave tiles of 96 x 32 or 96 x 64");
  constexpr int NBS = 2, NIMG = A1 ? 1 : 2, A_IMG = kAH * NP, B_SLOT = BN * NP, RING = NIMG * A_IMG + NBS * B_SLOT;  // pieces
  constexpr int SC = BN < 128 ? BN : 128;  // columns staged per epilogue round
  static_assert(!Q || H2, "Q: an h2 variant");
  static_assert((RING + 1 + WM * BN / 2) * 16 <= (OCC == 2 ? 80 : 160) * 1024, "LDS");
  static_assert(RING * 16 >= WM * 32 * SC * 4, "epilogue staging (the ring is free by then)");
  constexpr int A_PASSES = (kAH + RPP - 1) / RPP, B_PASSES = (BN + RPP - 1) / RPP;   // 4, 3
  static_assert(A1 || A_PASSES <= 9, "one image pass per K-step of the chunk before");
  __shared__ piece_t smem[RING + 1 + WM * BN / 2];
  float* red = reinterpret_cast<float*>(&smem[RING + 1]);  // [WM][BN][2]

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = tile / grid_n, tile_n = tile - tile_m * grid_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int W = g.Wi, halo = W + 1;
  const int NCH = g.C / 64, KT = (ABL & 2) ? 1 : 9 * NCH;
  if (t == 0) smem[RING] = zero_piece();
  // lgkmcnt(0): the store has reached LDS before this wave arrives at the K loop's first barrier (a raw s_barrier does
  // not wait for outstanding LDS writes), after which every wave may read the slot
  __builtin_amdgcn_s_waitcnt(0xc07f);

  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(A, (long)g.M * g.C * 2L);
  const __amdgpu_buffer_rsrc_t rsrc_b = make_rsrc(Bw, (long)g.N * g.K * 2L);
  const int pc = t % NP, lrow = t / NP;
  const int src_piece = pc ^ lds_swz<NP>(lrow);  // rows of one thread differ by multiples of RPP (a multiple of 16): invisible to the swizzle
  static_assert(RPP % 16 == 0, "swizzle");

  // DMA state: byte offset of this thread's row in every pass (all ones = outside the tensor: the bounds check returns 0)
  // (a shape whose image takes more than twelve passes would recompute a pass's offset when it is issued instead of holding it)
  constexpr bool A_OFF_REGS = A_PASSES <= 12;
  unsigned a_off[A_OFF_REGS ? A_PASSES : 1], b_off[B_PASSES];
  auto a_offset = [&](int i) -> unsigned {
    const int pix = m0 - halo + lrow + i * RPP;
    return (pix >= 0 && pix < g.M) ? (unsigned)pix * (unsigned)(g.C * 2) + (unsigned)src_piece * 16u : ~0u;
  };
  if constexpr (A_OFF_REGS) {
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) a_off[i] = a_offset(i);
  }
#pragma unroll
  for (int j = 0; j < B_PASSES; ++j) {
    const int n = n0 + lrow + j * RPP;
    b_off[j] = n < g.N ? (unsigned)(n * g.K) * 2u + (unsigned)src_piece * 16u : ~0u;
  }
  // pass p of the A image of `chunk`; a wave whose rows lie past the image skips the instruction (wave-uniform)
  auto issue_a = [&](int chunk, int p) {
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i)
      if (i == p && wave * (64 / NP) + i * RPP < kAH)
        dma16(rsrc_a, A_OFF_REGS ? a_off[A_OFF_REGS ? i : 0] : a_offset(i), (unsigned)(chunk * 128),
              &smem[(A1 ? 0 : (chunk & 1)) * A_IMG + i * RPP * NP + wave * 64]);
  };
  auto issue_b = [&](int kt) {
    const int chunk = kt / 9, tap = kt - chunk * 9;
    const unsigned soff = (unsigned)((tap * g.C + chunk * 64) * 2);
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j)
      if (wave * (64 / NP) + j * RPP < BN)
        dma16(rsrc_b, b_off[j], soff, &smem[NIMG * A_IMG + (kt % NBS) * B_SLOT + j * RPP * NP + wave * 64]);
  };
  // Fragment rows of this lane: tile-local row, its validity mask (bit tap set = outside the image, 9 bits per row tile)
  const int l31 = lane & 31, hi = lane >> 5;
  unsigned inv_mask = 0;  // [i] at bits 9i .. 9i+8
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * WTM + i * 32 + l31;
    unsigned inv = 0x1ffu;
    if (m < g.M) {
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
    inv_mask |= inv << (9 * i);
  }

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // prologue: the whole image of chunk 0 and the first NBS - 1 B slices
#pragma unroll
  for (int p = 0; p < A_PASSES; ++p) issue_a(0, p);
#pragma unroll
  for (int j = 0; j < NBS - 1; ++j)
    if (j < KT) issue_b(j);

  int chunk = 0, tap = 0, tr = 0, ts = 0;
  for (int kt = 0; kt < KT; ++kt) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // this step's B slice (and any image pass in flight) landed; last step's reads are done
    auto issue_step = [&]() {
      if constexpr ((ABL & 4) == 0) {
        if (kt + NBS - 1 < KT) issue_b(kt + NBS - 1);
        if constexpr (!A1)
          if (tap < A_PASSES && chunk + 1 < NCH) issue_a(chunk + 1, tap);
      }
    };
    if constexpr (DP == 0) issue_step();
    const piece_t* img = &smem[(A1 ? 0 : (chunk & 1)) * A_IMG];
    const piece_t* tb = &smem[NIMG * A_IMG + (kt % NBS) * B_SLOT];
    const int shift = halo + (g.off_h + g.rsign * tr) * W + g.off_w + g.ssign * ts;  // scalar
    // per row tile: LDS row of this lane for this tap, its swizzle, and whether the lane reads the zero slot instead
    int arow[TM], aswz[TM];
    bool ainv[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      arow[i] = wm * WTM + i * 32 + l31 + shift;
      aswz[i] = lds_swz<NP>(arow[i]);
      ainv[i] = ((inv_mask >> (9 * i + tap)) & 1u) != 0;
    }
    const int brow = wn * WTN + l31, bswz = lds_swz<NP>(l31);  // B rows: multiples of 32 plus l31
    if constexpr (H2 && Q) {
      typedef int intx8 __attribute__((ext_vector_type(8)));
      union QFrag {
        piece_t p[2];
        intx8 v;
      };
      const int sa = hi ? 127 - kQA_HI : 127 - kQA_LO, sb = hi ? 127 - kQW_LO : 127 - kQW_HI;  // E8M0 scales of block `hi`
      QFrag qa[TM], qb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const piece_t* p0 = ainv[i] ? &smem[RING] : img + arow[i] * NP + ((4 + hi) ^ aswz[i]);
        const piece_t* p1 = ainv[i] ? &smem[RING] : img + arow[i] * NP + ((6 + hi) ^ aswz[i]);
        qa[i].p[0] = *p0;
        qa[i].p[1] = *p1;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        qb[j].p[0] = tb[(brow + j * 32) * NP + ((4 + hi) ^ bswz)];
        qb[j].p[1] = tb[(brow + j * 32) * NP + ((6 + hi) ^ bswz)];
      }
      PieceView<half> ha[TM], hb[TN];
      auto load_hi = [&](int h) {  // the fp16 hi halves of 16-k half h: piece 2 h + {0, 1}
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const piece_t* p = ainv[i] ? &smem[RING] : img + arow[i] * NP + ((2 * h + hi) ^ aswz[i]);
          ha[i].p = *p;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) hb[j].p = tb[(brow + j * 32) * NP + ((2 * h + hi) ^ bswz)];
      };
      constexpr bool EARLY = !(NT == 768 && TN == 2);  // (the 256-column shape has no registers for both groups at once)
      if constexpr (EARLY) load_hi(0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa[i].v, qb[j].v, acc[i][j], 0, 0, 0, sa, 0, sb);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DP > 0) {
        issue_step();
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h == 1 || !EARLY) load_hi(h);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) mma_piece<half>(ha[i], hb[j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (H2) {
      // fragment (plane pl, 16-k half h of the group) = pieces 4 pl + 2 h + {0, 1}.  The 128-column shape (48 accumulator
      // registers) keeps both halves' fragments in registers -- the second half's reads fly under the first half's MFMAs --
      // the 256-column shape (96 accumulator registers of the 168 a wave may have) one half at a time.
      constexpr int NS = (NT == 768 && TN == 2) ? 1 : 2;
      PieceView<half> xa[NS][2][TM], xb[NS][2][TN];  // [slot][plane]
      auto load_h2 = [&](int h, int slot) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          const int piece = 4 * pl + 2 * h + hi;
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const piece_t* p = ainv[i] ? &smem[RING] : img + arow[i] * NP + (piece ^ aswz[i]);
            xa[slot][pl][i].p = *p;
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) xb[slot][pl][j].p = tb[(brow + j * 32) * NP + (piece ^ bswz)];
        }
      };
      load_h2(0, 0);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int sl = NS == 2 ? h : 0;
        if (NS == 2 && h == 0) load_h2(1, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            mma_piece<half>(xa[sl][1][i], xb[sl][0][j], acc[i][j]);
            mma_piece<half>(xa[sl][0][i], xb[sl][1][j], acc[i][j]);
            mma_piece<half>(xa[sl][0][i], xb[sl][0][j], acc[i][j]);
          }
        __builtin_amdgcn_sched_barrier(0);
        if (NS == 1 && h == 0) load_h2(1, 0);
        if constexpr (DP > 0)
          if (h == 0) {
            issue_step();
            __builtin_amdgcn_sched_barrier(0);
          }
      }
    } else {
    PieceView<half> fa[2][TM], fb[2][TN];
    auto load_frags = [&](int ks, int slot) {
      const int piece = ks * 2 + hi;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const piece_t* p = ainv[i] ? &smem[RING] : img + arow[i] * NP + (piece ^ aswz[i]);
        fa[slot][i].p = *p;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[slot][j].p = tb[(brow + j * 32) * NP + (piece ^ bswz)];
    };
    if ((ABL & 8) == 0 || kt == 0) load_frags(0, 0);
#pragma unroll
    for (int ks = 0; ks < NP / 2; ++ks) {
      if (ks + 1 < NP / 2 && ((ABL & 8) == 0 || kt == 0)) load_frags(ks + 1, (ks + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((ABL & 16) == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) mma_piece<half>(fa[ks & 1][i], fb[ks & 1][j], acc[i][j]);
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa[ks & 1][i].p));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fb[ks & 1][j].p));
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DP > 0)
        if (ks + 1 == DP) {
          issue_step();
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    }  // fp16 / h2
    if constexpr (A1) {  // the chunk's last K-step has been read by this wave: once every wave is here the image may be replaced
      if (tap == 8 && chunk + 1 < NCH && (ABL & 4) == 0) {
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) issue_a(chunk + 1, p);
      }
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
  __syncthreads();  // all fragment reads done before the ring is reused as epilogue staging
  if constexpr ((ABL & 1) != 0) {
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) keep += acc[i][j][0] + acc[i][j][15];
    if (keep == 12345.678f) reinterpret_cast<half*>(ep.out)[0] = (half)keep;  // keeps the accumulators live
    return;
  }

  // ---- epilogue: igemm.h's (alpha, bias, ReLU, BatchNorm column sums, residual / gates, 16-byte stores), three wave
  //      rows staged per round; no output row map, no stream-K -------------------------------------------------------
  using OT = typename std::conditional<H2, float, half>::type;  // h2: fp32 output and residual, h2 gates (hi halves read)
  constexpr int OVEC = 16 / (int)sizeof(OT);
  OT* out = reinterpret_cast<OT*>(ep.out);
  const OT* res = reinterpret_cast<const OT*>(ep.res);
  const half* gate = reinterpret_cast<const half*>(ep.res_gate);
  const half* ogate = reinterpret_cast<const half*>(ep.out_gate);
  float* stage = reinterpret_cast<float*>(&smem[0]);  // [WM * 32][SC] fp32
  constexpr int CPR = SC / OVEC, PASSES = (WM * 32 * CPR + NT - 1) / NT;  // 2 (h2: 4)
  float s1[TN], s2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
  for (int i = 0; i < TM; ++i)
    for (int nh = 0; nh < BN / SC; ++nh) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int lc = wn * WTN + j * 32 + l31;
        if (lc / SC == nh) {
          const float bias = (ep.bias && n0 + lc < g.N) ? ep.bias[n0 + lc] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[i][j][r] * ep.alpha + bias;
            if (ep.relu & 1) v = fmaxf(v, 0.f);
            s1[j] += v;
            s2[j] += v * v;
            const int lr = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            stage[lr * SC + (lc - nh * SC)] = v;
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        const int id = t + ps * NT;
        const int lr = id / CPR, cpc = id % CPR;
        const int row = m0 + (lr >> 5) * WTM + i * 32 + (lr & 31);
        const int col = n0 + nh * SC + cpc * OVEC;
        if (lr < WM * 32 && row < g.M && col < g.N) {
          float v[OVEC];
#pragma unroll
          for (int e = 0; e < OVEC; e += 4) {
            floatx4 f = *reinterpret_cast<const floatx4*>(&stage[lr * SC + cpc * OVEC + e]);
            v[e] = f[0];
            v[e + 1] = f[1];
            v[e + 2] = f[2];
            v[e + 3] = f[3];
          }
          const long idx = (long)row * ep.ldc + col;
          if constexpr (H2) {
            const long gidx = h2_index(row, ep.ldc, col);
            if (res) {
              PieceView<float> rv;
              Half4View gv;
              rv.p = *reinterpret_cast<const piece_t*>(res + idx);
              if (gate) gv.p = *reinterpret_cast<const u32x2*>(gate + gidx);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float x = rv.e[e];
                if (gate && !((float)gv.e[e] > 0.f)) x = 0.f;
                v[e] += x;
              }
            }
            if (ogate) {
              Half4View ov;
              ov.p = *reinterpret_cast<const u32x2*>(ogate + gidx);
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (!((float)ov.e[e] > 0.f)) v[e] = 0.f;
            }
            PieceView<float> o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o.e[e] = v[e];
            *reinterpret_cast<piece_t*>(out + idx) = o.p;
          } else {
          const long gix = ep.gate_h2 ? h2_index(row, ep.ldc, col) : idx;  // (gate_h2: hi halves of an h2 gate tensor, igemm.h)
          if (res) {
            PieceView<half> rv, gv;
            rv.p = *reinterpret_cast<const piece_t*>(res + idx);
            if (gate) gv.p = *reinterpret_cast<const piece_t*>(gate + gix);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              float x = (float)rv.e[e];
              if (gate && !((float)gv.e[e] > 0.f)) x = 0.f;
              v[e] += x;
            }
          }
          if (ogate) {
            PieceView<half> ov;
            ov.p = *reinterpret_cast<const piece_t*>(ogate + gix);
#pragma unroll
            for (int e = 0; e < VEC; ++e)
              if (!((float)ov.e[e] > 0.f)) v[e] = 0.f;
          }
          PieceView<half> o;
#pragma unroll
          for (int e = 0; e < VEC; ++e) o.e[e] = (half)v[e];
          *reinterpret_cast<piece_t*>(out + idx) = o.p;
          }
        }
      }
      __syncthreads();
    }
  if (ep.stats || ep.stats_accum) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      s1[j] += __shfl_xor(s1[j], 32);
      s2[j] += __shfl_xor(s2[j], 32);
      if (lane < 32) {
        const int lc = wn * WTN + j * 32 + lane;
        red[(wm * BN + lc) * 2 + 0] = s1[j];
        red[(wm * BN + lc) * 2 + 1] = s2[j];
      }
    }
    __syncthreads();
    for (int c = t; c < BN; c += NT)
      if (n0 + c < g.N) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          a += red[(w * BN + c) * 2 + 0];
          b += red[(w * BN + c) * 2 + 1];
        }
        if (ep.stats_accum) {
          double* row = ep.stats_accum + (long)(tile_m % ep.stats_rows) * 2 * g.N;
          atomicAdd(row + n0 + c, (double)a);
          atomicAdd(row + g.N + n0 + c, (double)b);
        } else {
          ep.stats[((long)tile_m * 2 + 0) * g.N + n0 + c] = a;
          ep.stats[((long)tile_m * 2 + 1) * g.N + n0 + c] = b;
        }
      }
  }
}

// (Measured and removed, round 4: static wave priorities -- s_setprio 0 / 1 / 2 for the three waves that share a SIMD (w, w + 4,
// w + 8), or 0 / 1 / 1 -- in the 8- and 12-wave shapes: layers 2-4 forward / data gradient within +-1 % in fp16 and fp16x2, whole
// step 13.43 / 13.46 / 13.46 and 13.42 / 13.42 / 13.41 ms; the per-K-step barrier re-aligns the waves.  profiles/r04/c41_*.)
// the launches the kernels cover: fp16 3x3, stride 1, same size, 64-channel chunks
inline bool igemm_halo_applies(const GatherGeom& g, const Epilogue& ep, int bn, int ah, int bm = 288) {
  return g.R == 3 && g.S == 3 && g.mul_p == 1 && g.mul_q == 1 && g.div == 1 && g.P == g.Hi && g.Q == g.Wi &&
         g.C % 64 == 0 && g.N % bn == 0 && g.K == 9 * g.C && !g.bt_on && (g.ldb == 0 || g.ldb == g.K) &&
         (g.rsign == 1 || g.rsign == -1) && g.rsign == g.ssign && bm + 2 * (g.Wi + 1) <= ah && g.M == g.B * g.P * g.Q &&
         (long)g.M * g.C * 2 < 0xfffffff0l && (long)g.N * g.K * 2 < 0xfffffff0l && !ep.om_on;
}

// MN_IGEMM_HALO: 1 = the 256-column shape where igemm.h would pick its own 288x256 tile (layer3 at 192 images, or a
// forced MN_IGEMM_CONFIG=12); 2 = additionally the 128-column shape for every other launch it covers (layers 2 and 4).
// Returns the number of M-blocks used (rows of a [grid_m][2][N] statistics buffer), or -1 if the launch is not taken.
inline int launch_igemm_halo(const GatherGeom& g, const half* A, const half* Bw, const Epilogue& ep, hipStream_t stream,
                             int level, bool tile288_wanted) {
  const int gm = cdiv(g.M, 288);
  RowDiv rd;
  rd.q = make_fastdiv(g.Q);
  rd.p = make_fastdiv(g.P);
#ifdef MN_ABLATION_BUILD
  static const int abl = getenv("MN_HALO_ABLATE") ? atoi(getenv("MN_HALO_ABLATE")) : 0;
#endif
  // (MN_HALO_A1_F16: the same switch for the plain-fp16 launches alone -- in the fp16x2m mode those are the data gradients, whose
  //  70 KB form leaves room for a weight-gradient workgroup of the side stream on the same CU: A/B knob)
  static const int a1 = getenv("MN_HALO_A1_F16") ? atoi(getenv("MN_HALO_A1_F16")) : (getenv("MN_HALO_A1") ? atoi(getenv("MN_HALO_A1")) : 1);
  if (level >= 1 && tile288_wanted && a1 != 2 && igemm_halo_applies(g, ep, 256, 352)) {
#ifdef MN_ABLATION_BUILD
#define MN_HALO_ABL(BN_, AH_, V_)                                                                                          \
  if (abl == V_) {                                                                                                         \
    hipLaunchKernelGGL((igemm_halo_kernel<BN_, AH_, V_>), dim3(gm * (g.N / BN_)), dim3(768), 0, stream, g, A, Bw, ep,      \
                       g.N / BN_, rd);                                                                                     \
    return gm;                                                                                                             \
  }
    MN_HALO_ABL(256, 352, 1) MN_HALO_ABL(256, 352, 2) MN_HALO_ABL(256, 352, 3) MN_HALO_ABL(256, 352, 4)
    MN_HALO_ABL(256, 352, 8) MN_HALO_ABL(256, 352, 16) MN_HALO_ABL(256, 352, 12) MN_HALO_ABL(256, 352, 20)
    MN_HALO_ABL(256, 352, 24) MN_HALO_ABL(256, 352, 28)
#endif
    hipLaunchKernelGGL((igemm_halo_kernel<256, 352>), dim3(gm * (g.N / 256)), dim3(768), 0, stream, g, A, Bw, ep, g.N / 256, rd);
    return gm;
  }
  // 192-row tiles of 4 waves (2 x 2 of 96 x 64) with ONE A image (the next chunk's image requested after the chunk's last K-step):
  // 70 KB of LDS, so two workgroups share a CU and each one's prologue / image reload / epilogue runs under the other's MFMAs --
  // for the 128-column layers with at least two rounds of such tiles (layer2 at 192 images: 1376 tiles).  Round 4, same-box A/B
  // against the 8-wave 384-row tile below: fp16 forward 90.8 -> 90.1 us, data gradient 87.5 -> 84.2, with residual 110.9 -> 105.8,
  // step 13.22 -> 13.16 ms; fp16x2 224 -> 202, 218 -> 199, 282 -> 250 us, step 28.82 -> 28.55 ms (profiles/r04/c30_*).
  // For layers 3 and 4 (704 / 352 such tiles: 1.4 / 0.7 rounds) it loses to their one-round shapes: fp16 76 -> 85 and 81 -> 88 us,
  // fp16x2 191 -> 207 and 223 -> 234 us (call 31).  MN_HALO_A1: 0 off, 2 always (parity tests)
  if (level >= 2 && a1 > 0 && (a1 == 2 || (long)cdiv(g.M, 192) * (g.N / 128) >= 2L * device_cus()) &&
      igemm_halo_applies(g, ep, 128, 288, 192)) {
    hipLaunchKernelGGL((igemm_halo_kernel<128, 288, 0, 1, false, 2, 2, true>), dim3(cdiv(g.M, 192) * (g.N / 128)), dim3(256), 0, stream, g,
                       A, Bw, ep, g.N / 128, rd);
    return cdiv(g.M, 192);
  }
  // (layer1 in fp16 on this shape with 64 columns, measured and removed: 132.5 us forward / 134.8 data gradient against 112.8 / 111.1
  //  for halo_pp.h's persistent kernel with resident weights; whole step 13.32 -> 13.63 ms, profiles/r04/c34_*)
  // the 8-wave 384-row tile of 96 x 64 wave tiles where it fills the chip's rounds as well as the 288-row tile does (layer2 at 192
  // images: 688 tiles = 2.69 rounds against 918 = 3.59): 0.55 instead of 0.89 fragment reads per MFMA; layer2 forward 99.3 -> 92.7
  // us, data gradient 97.2 -> 90.5, whole step 13.95 -> 13.79 ms (round 4, profiles/r04/c28_*).  MN_HALO384: 0 never, 2 always.
  static const int bm384 = getenv("MN_HALO384") ? atoi(getenv("MN_HALO384")) : 1;
  if (level >= 2 && bm384 > 0 && igemm_halo_applies(g, ep, 128, 480, 384)) {
    const long t384 = (long)cdiv(g.M, 384) * (g.N / 128), t288 = (long)gm * (g.N / 128);
    const int cus = device_cus();
    const double e384 = (double)t384 / ((double)cdiv(t384, cus) * cus), e288 = (double)t288 / ((double)cdiv(t288, cus) * cus);
    if (bm384 == 2 || e384 >= e288 - 0.03) {
      hipLaunchKernelGGL((igemm_halo_kernel<128, 480, 0, 1, false, 4, 2>), dim3(cdiv(g.M, 384) * (g.N / 128)), dim3(512), 0, stream, g, A,
                         Bw, ep, g.N / 128, rd);
      return cdiv(g.M, 384);
    }
  }
  if (level >= 2 && igemm_halo_applies(g, ep, 128, 384)) {
#ifdef MN_ABLATION_BUILD
    MN_HALO_ABL(128, 384, 1) MN_HALO_ABL(128, 384, 2) MN_HALO_ABL(128, 384, 3) MN_HALO_ABL(128, 384, 4)
    MN_HALO_ABL(128, 384, 8) MN_HALO_ABL(128, 384, 16) MN_HALO_ABL(128, 384, 12) MN_HALO_ABL(128, 384, 20)
    MN_HALO_ABL(128, 384, 24) MN_HALO_ABL(128, 384, 28)
#undef MN_HALO_ABL
#endif
    hipLaunchKernelGGL((igemm_halo_kernel<128, 384>), dim3(gm * (g.N / 128)), dim3(768), 0, stream, g, A, Bw, ep, g.N / 128, rd);
    return gm;
  }
  return -1;
}

// (Measured and removed, round 4: a layer1 kernel for the fp16x2 mode with the WHOLE K extent of A resident -- both 32-channel
// groups of rows [m0 - (W+1), m0 + 288 + (W+1)), 2 x 59 KB, nine K-steps that only stream 16 KB of weights, 12 waves = split-K
// over the two groups x 3 x 2 wave tiles of 96 x 32, accumulators exchanged through LDS -- parity-green and exactly as fast as
// the generic 128x64 kernel, 338 us per launch at 192 images.  Ablation (profiles/r04/c8_*): epilogue 95 us (270 MB of fp32
// output with nothing to overlap it: 153 KB of LDS = one workgroup per CU), prologue + first K-step 73 us, the MFMAs 126 us
// (= their floor) and 62 us of fragment reads / barriers, all ADDITIVE; 96 x 32 wave tiles need 0.9 ds_read_b128 per MFMA.
// The generic kernel is DMA-bound at the same 330 us (24 KB of LDS-DMA per K-step of 48 MFMAs at ~20 B/clk/CU).)
// h2 form: `g2` is the doubled geometry launch_igemm_h2 builds (g2.C = 2 x real channels); every 3x3 stride-1 shape the fp16
// kernels cover, the 256-column shape when it fills the chip in one round (layer3), the 128-column shape otherwise
inline int launch_igemm_halo_h2(const GatherGeom& g2, const half* A, const half* Bw, const Epilogue& ep, hipStream_t stream) {
  const int gm = cdiv(g2.M, 288);
  const bool q = g2.mma == MMA_H2Q;  // h2q operands: the scaled-fp8 cross terms (igemm_halo_kernel Q)
  RowDiv rd;
  rd.q = make_fastdiv(g2.Q);
  rd.p = make_fastdiv(g2.P);
  const long tiles288 = (long)gm * (g2.N / 256);
  static const bool force256 = getenv("MN_H2_HALO256") && atoi(getenv("MN_H2_HALO256")) != 0;  // (parity tests on small problems)
  static const int a1 = getenv("MN_HALO_A1") ? atoi(getenv("MN_HALO_A1")) : 1;  // (see launch_igemm_halo)
  if (g2.N % 256 == 0 && ((tiles288 > 192 && tiles288 <= device_cus()) || force256) && igemm_halo_applies(g2, ep, 256, 352)) {
    if (q)
      hipLaunchKernelGGL((igemm_halo_kernel<256, 352, 0, 1, true, 3, 4, false, 1, true>), dim3(gm * (g2.N / 256)), dim3(768), 0, stream, g2, A, Bw, ep,
                       g2.N / 256, rd);
    else
      hipLaunchKernelGGL((igemm_halo_kernel<256, 352, 0, 1, true>), dim3(gm * (g2.N / 256)), dim3(768), 0, stream, g2, A, Bw, ep,
                       g2.N / 256, rd);
    return gm;
  }
  // layer1 (64 -> 64 channels, rows of up to 87 pixels): 192-row tiles of 4 waves, one A image, two workgroups per CU
  static const bool halo64 = !(getenv("MN_H2_HALO64") && atoi(getenv("MN_H2_HALO64")) == 0);
  // (measured: 338 -> 304 us forward, 356 -> 326 data gradient, step 29.69 -> 29.52 ms; 288-row tiles of 6 waves at two workgroups
  //  per CU -- 27 % less DMA per row -- 409 us: six waves do not spread over four SIMDs; profiles/r04/c15_*, c16_*)
  if (halo64 && g2.N == 64 && igemm_halo_applies(g2, ep, 64, 368, 192)) {
    if (q)
      hipLaunchKernelGGL((igemm_halo_kernel<64, 368, 0, 1, true, 2, 2, true, 2, true>), dim3(cdiv(g2.M, 192)), dim3(256), 0, stream, g2, A, Bw, ep, 1,
                       rd);
    else
      hipLaunchKernelGGL((igemm_halo_kernel<64, 368, 0, 1, true, 2, 2, true>), dim3(cdiv(g2.M, 192)), dim3(256), 0, stream, g2, A, Bw, ep, 1,
                       rd);
    return cdiv(g2.M, 192);
  }
  // layer1's two-workgroup shape for the 128-column layers with at least two rounds of 192-row tiles (layer2): see launch_igemm_halo
  if (a1 > 0 && g2.N % 128 == 0 && (a1 == 2 || (long)cdiv(g2.M, 192) * (g2.N / 128) >= 2L * device_cus()) &&
      igemm_halo_applies(g2, ep, 128, 288, 192)) {
    if (q)
      hipLaunchKernelGGL((igemm_halo_kernel<128, 288, 0, 1, true, 2, 2, true, 2, true>), dim3(cdiv(g2.M, 192) * (g2.N / 128)), dim3(256), 0, stream,
                       g2, A, Bw, ep, g2.N / 128, rd);
    else
      hipLaunchKernelGGL((igemm_halo_kernel<128, 288, 0, 1, true, 2, 2, true>), dim3(cdiv(g2.M, 192) * (g2.N / 128)), dim3(256), 0, stream,
                       g2, A, Bw, ep, g2.N / 128, rd);
    return cdiv(g2.M, 192);
  }
  // 128-column layers: the 8-wave 384-row tile where it fills the chip's rounds about as well as the 288-row tile does (layer2 at
  // 192 images: 688 tiles = 2.69 rounds against 918 = 3.59; layer4: 176 tiles = 0.69 of a round against 236 = 0.92 -> 288 rows)
  static const int bm384 = getenv("MN_H2_HALO384") ? atoi(getenv("MN_H2_HALO384")) : 1;  // 0: never, 1: by tile count, 2: always
  if (bm384 > 0 && igemm_halo_applies(g2, ep, 128, 480, 384)) {
    const long t384 = (long)cdiv(g2.M, 384) * (g2.N / 128), t288 = (long)gm * (g2.N / 128);
    const int cus = device_cus();
    const double e384 = (double)t384 / ((double)cdiv(t384, cus) * cus), e288 = (double)t288 / ((double)cdiv(t288, cus) * cus);
    if (bm384 == 2 || e384 >= e288 - 0.03) {
    if (q)
      hipLaunchKernelGGL((igemm_halo_kernel<128, 480, 0, 1, true, 4, 2, false, 1, true>), dim3(cdiv(g2.M, 384) * (g2.N / 128)), dim3(512), 0, stream, g2,
                         A, Bw, ep, g2.N / 128, rd);
    else
      hipLaunchKernelGGL((igemm_halo_kernel<128, 480, 0, 1, true, 4, 2>), dim3(cdiv(g2.M, 384) * (g2.N / 128)), dim3(512), 0, stream, g2,
                         A, Bw, ep, g2.N / 128, rd);
      return cdiv(g2.M, 384);
    }
  }
  if (igemm_halo_applies(g2, ep, 128, 384)) {
    if (q)
      hipLaunchKernelGGL((igemm_halo_kernel<128, 384, 0, 1, true, 3, 4, false, 1, true>), dim3(gm * (g2.N / 128)), dim3(768), 0, stream, g2, A, Bw, ep,
                       g2.N / 128, rd);
    else
      hipLaunchKernelGGL((igemm_halo_kernel<128, 384, 0, 1, true>), dim3(gm * (g2.N / 128)), dim3(768), 0, stream, g2, A, Bw, ep,
                       g2.N / 128, rd);
    return gm;
  }
  return -1;
}

}  // namespace mn
