// Tap-fused weight gradient of 3x3 stride-1 "same" convolutions (fp16 operands, fp32 accumulate): what autograd
// computes for conv2d's weight under loss.backward() (/root/reference/common/train.py:351-355).
//
//   dW[n][tap][c] += alpha * sum_m dY[m][n] * X[m + shift(tap)][c] * inside(m, tap)
//
// wgrad.h treats this as a plain TN GEMM with K = 9*C columns: a 128 x 128 output tile moves 16 KB through LDS-DMA per
// 32 pixels, 62 B/clk/CU at full matrix-core rate against the ~15-20 B/clk/CU that path sustains -- the launches sit at
// 18-23 % of the MFMA peak, DMA-bound.  But the nine "columns blocks" of one channel tile are the SAME pixels, shifted.
// Here a workgroup owns dW[64 n][9 taps][64 c] -- 36 accumulator tiles of 32 x 32 over 4 waves (144 registers each) --
// and walks a range of pixels once: per 32 pixels it fetches 32 x 128 B of dY and 32 x 128 B of X (8 KB for 36 864 x 32
// MACs: 4.5x fewer bytes per MAC), and the nine taps read their B fragments from the same LDS-resident X rows at nine
// row offsets.
//
// Border handling without masks: pixels are enumerated in a PADDED order -- one zero column after every image row, one
// zero row after every image: j = (b (P+1) + p) (Q+1) + q, real iff p < P and q < Q.  In that order tap (dh, dw) is the
// pure shift dh (Q+1) + dw, a neighbour outside the image IS a padding position, and padding positions are zero-filled
// by the buffer bounds check of the DMA (all-ones offset), for X and for dY alike.  Cost: (1 + 1/P)(1 + 1/Q) - 1 extra
// MACs (2.7 % on layer1, 22 % on layer4's 8 x 11 maps).
//
// X lives in a ring of `ring` LDS rows (row = one position, 64 channels = 128 B, 16-byte pieces XOR-swizzled with
// row & 3 like wgrad.h so the four rows of a transpose read fall in distinct bank groups): step s reads ring rows
// [32 s + Gpad + shift, + 32) for each tap while the DMAs of steps s+1 .. s+D land 32 new rows each right behind the live
// window (ring = 32 (D+1) + 2 Gpad rows, Gpad = Q + 2 rounded up to 16; D+1 dY tiles; counted vmcnt waits -- at two
// workgroups per CU one step of 18 MFMAs per wave is 0.3 us, a third of the DMA latency: with D = 1 the kernel ran at
// 390 TF, latency-bound, round-2 measurement).  Fragments come from ds_read_b64_tr_b16 issued from inline
// assembly with counted lgkmcnt waits (hipcc drains vmcnt(0) in front of the builtin while an LDS-DMA is in flight,
// wgrad.h); the reads of tap t+1 are in flight under the MFMA of tap t.  A tap window that wraps around the end of the
// ring (one step in ring/32 per tap) takes a slower path that wraps every row address.
#pragma once
#include "wgrad.h"  // (included from the end of wgrad.h)

namespace mn {

struct WgradFusedArgs {
  const half* dY;  // [M][ldy], M = B*P*Q
  const half* X;   // [B][P][Q][C]
  float* dW;       // [N][ldw] fp32, column (tap*C + c), accumulated atomically when there is no workspace
  float* ws;       // optional [nchunks][N][9*C] fp32: every workgroup stores its partial tile there (plain stores) and
                   // wgrad_fused_reduce_kernel adds the chunks up in order; null = fp32 atomics straight into dW
  int ldy, ldw;
  int B, P, Q, C, N;
  int Qp;       // Q + 1
  int J;        // B * (P+1) * (Q+1) padded positions
  int chunk;    // positions per workgroup (multiple of the step)
  int nchunks;  // workgroups per (n tile, c tile) pair
  int tiles_n, tiles_c;
  int Gpad;     // halo rows kept on each side of the live window (>= Q + 2, multiple of 16)
  int ring;     // LDS rows of the X ring = (D+1)*BKM + 2*Gpad
  FastDiv dq, dp;  // divisors Q+1 and P+1
  float alpha;
  bool det = false;  // MN_DETERMINISTIC: one reduction group (plain ordered sums, no atomics anywhere)
  bool x_h2 = false; // (round 6) X is an h2 tensor of which the hi halves are read (WgradArgs::x_h2); wgrad_fused_h2_kernel<.., LO = false>
};

// LDS rows of the X ring: (D+1) steps + 2 Gpad; bounds the image width (Gpad <= 96 -> Q <= 94, i.e. inputs up to 376
// pixels wide at layer1).  4-wave form: 32-row steps; 8-wave form: 64-row steps.
constexpr int WGF_RING_MAX8 = 448;

// 8 waves = 512 threads, 64-pixel steps, one workgroup per CU.  Wave roles: 32-column block of the 64 input channels (wc),
// pixel half of every step (grp: rows 0-31 / 32-63 -- split-K inside the workgroup: at the end the two halves are added
// through LDS, so a CU flushes ONE partial tile; the partial-tile volume of a launch, what the reduce launch has to read
// back, is the number of resident accumulator tiles x 147 KB = 38 MB) and tap half (tset: taps 0-4 or 5-8).  A wave owns
// BOTH 32-row blocks of the 64 output channels for its taps, so a B fragment (X rows at the tap's shift) feeds TWO MFMAs:
// 1.4 fragment-read instructions per MFMA and one address add + one counted wait per two MFMAs.  Waves w and w + 4 (one
// SIMD) hold the 5-tap and the 4-tap half: 18 MFMAs per SIMD per K sub-step.
// (Round 2's form gave a wave one output-channel block for all nine taps -- a B fragment per MFMA, 2.2 read instructions
// per MFMA.  Same time per launch in isolation, 101-109 us at every layer, so the loop is NOT issue-bound as round 2
// concluded; the whole step runs 0.8 % faster with this form beside the other stream's kernels, 14.45 vs 14.60 ms,
// profiles/r03/c8_*.  A 4-wave form -- two 256-thread workgroups per CU, twice the partial tiles -- was slower and is gone.)
// ABL (timing experiments only, MN_WGF_ABLATE in the ablation build; results are wrong): bit 0 = no DMA after the prologue,
// bit 1 = no B-fragment reads, bit 2 = no MFMA, bit 3 = no stores / atomics.
// PD: B fragments requested ahead of the MFMAs that consume them.
// DPI: the LDS-DMA of step s + D is issued behind the MFMAs of item DPI of step s (-1: right after the barrier, rounds 1-2).
// Measured per launch at layers 1-4 (profiles/r03/c14_*, c15_*): -1: 112 / 105 / 101 / 109 us, 1: 111 / 101 / 98 / 106,
// 4: 111 / 103 / 99 / 108, 7: 108 / 101 / 97 / 105, 8: 110 / 109 / 102 / 109, 9: 116 / 112 / 106 / 114; whole step 13.94 (-1) vs
// 13.85 ms (7) on one box -- right after the barrier the requests (and the address arithmetic of their pixels) compete with
// every wave's first fragment reads.
// (Measured and removed, round 3, profiles/r03/c9_*: a ROTATED loop -- the A fragments and the first PD B fragments of step
// s + 1 requested at the end of step s, in flight across the barrier, for which step s + 1 has to be visible one barrier
// early, i.e. one DMA step fewer in flight -- 101-109 -> 106-115 us per launch, step 14.49 -> 14.59 ms.)
// No item ever wraps around the ring: the 64 rows behind the ring mirror its first block (the DMA that fills block 0 is
// issued twice), so a tap window that starts near the end simply runs on into the mirror.
template <int D, int ABL = 0, int PD = 3, int DPI = 7>
static __global__ void __launch_bounds__(512, 2) wgrad_fused_kernel(WgradFusedArgs a) {
  constexpr int NW = 8;
  static_assert(D >= 1 && D <= 3, "steps in flight");
  constexpr int BKM = 8 * NW;              // pixels per step = one DMA pass of all threads per operand
  constexpr int ROWH = 64;                 // halves per LDS row (64 channels / 64 output channels)
  constexpr int TILE_Y = BKM * ROWH;       // halves
  constexpr int NY = D + 1;                // dY tiles
  constexpr int RING_MAX = WGF_RING_MAX8;
  // ONE LDS object: [NY dY tiles][X ring (RING rows used)][mirror of ring rows 0 .. BKM-1, right behind row RING-1]
  __shared__ half smem[NY * TILE_Y + (RING_MAX + BKM) * ROWH] __attribute__((aligned(16)));
  half* ring = &smem[NY * TILE_Y];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wc = wave & 1, grp = (wave >> 1) & 1, tset = wave >> 2;  // 32-column block of the 64 inputs; pixel half; tap half
  const int tap0 = tset * 5, ntap = tset == 0 ? 5 : 4;
  // logical id = (chunk, pair): the pairs of one pixel range are adjacent, i.e. on one XCD, and share its L2 lines
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int pairs = a.tiles_n * a.tiles_c;
  const int ci = logical / pairs, pr = logical - ci * pairs;
  const int n0 = (pr / a.tiles_c) * 64, c0 = (pr % a.tiles_c) * 64;
  const int j0 = ci * a.chunk;
  const int j1 = min(a.J, j0 + a.chunk);
  const int nsteps = (j1 - j0 + BKM - 1) / BKM;
  const int Gpad = a.Gpad, RING = a.ring;

  const __amdgpu_buffer_rsrc_t rsrc_y = make_rsrc(a.dY, (long)a.B * a.P * a.Q * a.ldy * 2L);
  const __amdgpu_buffer_rsrc_t rsrc_x = make_rsrc(a.X, (long)a.B * a.P * a.Q * a.C * 2L);

  // DMA role of this thread: row t/8 of a BKM-row block, 16-byte slot t%8; LDS slot s of row r holds source piece
  // s ^ swz(r), and every block starts at a multiple of 32 rows, so the source piece is fixed per thread
  const int drow = t >> 3, dslot = t & 7;
  const int dpiece = dslot ^ wg_swz<8>(drow);
  const unsigned ycol = (unsigned)((n0 + dpiece * 8) * 2), xcol = (unsigned)((c0 + dpiece * 8) * 2);
  const bool y_ok = n0 + dpiece * 8 < a.N, x_ok = c0 + dpiece * 8 < a.C;
  // padded position -> pixel index (or -1: padding / outside the tensor)
  auto pixel_of = [&](int j) -> int {
    if ((unsigned)j >= (unsigned)a.J) return -1;
    const int r = fastdiv(j, a.dq), q = j - r * a.Qp;
    const int b = fastdiv(r, a.dp), p = r - b * (a.P + 1);
    return (q < a.Q && p < a.P) ? (b * a.P + p) * a.Q + q : -1;
  };
  auto issue_y = [&](int step) {  // positions [j0 + BKM step, + BKM) of dY -> tile step % NY (zero past the chunk's end)
    const int j = j0 + step * BKM + drow;
    const int m = j < j1 ? pixel_of(j) : -1;
    const unsigned off = (m >= 0 && y_ok) ? (unsigned)m * (unsigned)(a.ldy * 2) + ycol : ~0u;
    dma16(rsrc_y, off, 0u, &smem[(step % NY) * TILE_Y + wave * 64 * 8]);
  };
  auto issue_x = [&](int u0) {  // ring-relative rows [u0, u0 + BKM): positions j0 - Gpad + u0 + ..
    const int m = pixel_of(j0 - Gpad + u0 + drow);
    const unsigned off = (m >= 0 && x_ok) ? (unsigned)m * (unsigned)(a.C * 2) + xcol : ~0u;
    const int rr = u0 % RING;  // wave-uniform; blocks never straddle the end (RING and u0 are multiples of BKM)
    dma16(rsrc_x, off, 0u, ring + rr * ROWH + wave * 64 * 8);
    // block 0 also goes to the mirror.  (One more DMA instruction in the queue on these steps: the counted waits below
    // then wait for one instruction more than they need to, never for one fewer -- retirement is in order.)
    if (rr == 0) dma16(rsrc_x, off, 0u, ring + RING * ROWH + wave * 64 * 8);
  };

  floatx16 acc[2][5];  // [32-row block of the output channels][tap slot: tap = tap0 + slot]
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][i][r] = 0.f;

  // transpose-read lane geometry (wgrad.h): 16-lane group gq -> column block (gq & 1) * 16, k half (gq >> 1) * 8; as a
  // SOURCE lane this lane addresses row lrow = kgrp + (lane & 15) / 4 and the 8-byte chunk (lane & 3) of its column block
  const int gq = lane >> 4, i16 = lane & 15;
  const int src_row = i16 >> 2, src_chunk = (i16 & 3) * 4 + (gq & 1) * 16;
  const int lrow = grp * 32 + (gq >> 1) * 8 + src_row;  // + this wave group's half of the step
  const unsigned lds0 = lds_addr_of(smem);
  // A operand (dY tile): row lrow, column nb*32 + src_chunk; the swizzle sees row & 3 = src_row
  unsigned aA[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int colA = nb * 32 + src_chunk;
    aA[nb] = lds0 + (unsigned)((lrow * ROWH + (((colA >> 3) ^ wg_swz<8>(src_row)) * 8) + (colA & 7)) * 2);
  }
  // B operand (X ring), tap tp: byte address = ring + rbB[tp] (scalar: first row of the tap's window, advanced per step)
  // + xoffB[tp] (lane: row lrow, column with the swizzle of (window start + src_row) & 3 -- window starts move in
  // multiples of 4, so the key is loop invariant per tap) + an immediate for the K sub-step
  const int colB = wc * 32 + src_chunk;
  unsigned xoffB[5];
  int rbB[5];
#pragma unroll
  for (int sl = 0; sl < 5; ++sl) {
    const int tp = tap0 + (sl < ntap ? sl : ntap - 1);  // (the 4-tap half re-reads its last tap in slot 4: no MFMA follows)
    const int sh = Gpad + (tp / 3 - 1) * a.Qp + (tp % 3 - 1);  // in [0, 2 Gpad] < RING
    const int key = (sh + src_row) & 3;
    xoffB[sl] = lds0 + (unsigned)((NY * TILE_Y + lrow * ROWH + ((colB >> 3) ^ (key << 1)) * 8 + (colB & 7)) * 2);
    rbB[sl] = sh * (ROWH * 2);
  }
  const int ring_bytes = RING * ROWH * 2;

  // prologue: the halo rows [0, 2 Gpad), then steps 0 .. D-1 (dY tile + BKM ring rows each), in the order the waits count
  for (int u0 = 0; u0 < 2 * Gpad; u0 += BKM) issue_x(u0);
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < nsteps) {
      issue_y(d);
      issue_x(2 * Gpad + BKM * d);
    }

  for (int s = 0; s < nsteps; ++s) {
    // steps issued so far: min(nsteps, s + D), at least two DMA instructions each, retired in order: step s has landed
    // when at most the later ones are outstanding
    const int ahead = min(nsteps, s + D) - (s + 1);
    if (D >= 3 && ahead >= 2)
      wait_vmcnt<4>();
    else if (D >= 2 && ahead >= 1)
      wait_vmcnt<2>();
    else
      wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // step s visible to everyone; everyone is done with step s-1's reads
    auto issue_step = [&]() {
      if (s + D < nsteps && (ABL & 1) == 0) {  // into dY tile (s - 1) % NY and the ring rows behind the live + in-flight window
        issue_y(s + D);
        issue_x(2 * Gpad + BKM * (s + D));
      }
    };
    if constexpr (DPI < 0) issue_step();
    const unsigned tyoff = (unsigned)((s % NY) * TILE_Y * 2);
    // One step = 10 (K sub-step, tap slot) items per wave, each TWO MFMAs (both output-channel blocks) fed by the A
    // fragments of that sub-step and ONE B fragment read at the tap's row shift.  B fragments travel through a ring of
    // PD + 1 register buffers: item i + PD is requested right before the MFMAs of item i.
    constexpr int NKS = 2, ITEMS = NKS * 5, NB = PD + 1;
    static_assert(2 * PD + 4 <= 15, "lgkmcnt field");
    TrFrag fa[NKS][2], fb[NB];
    __builtin_amdgcn_sched_barrier(0);
    static_for<NKS>([&](auto KS) {
      constexpr int ks = decltype(KS)::value;
      static_for<2>([&](auto NBk) {
        constexpr int nb = decltype(NBk)::value;
        fa[ks][nb].h[0] = ds_read_tr16_at<(ks * 16) * ROWH * 2>(smem, aA[nb] + tyoff);
        fa[ks][nb].h[1] = ds_read_tr16_at<(ks * 16 + 4) * ROWH * 2>(smem, aA[nb] + tyoff);
      });
    });
    auto read_b = [&](auto I, TrFrag& f) {
      constexpr int it = decltype(I)::value, ks = it / 5, sl = it % 5;
      if constexpr ((ABL & 2) != 0) {
        f.h[0] = fa[ks][0].h[1];
        f.h[1] = fa[ks][0].h[0];
        return;
      }
      const unsigned ad = xoffB[sl] + (unsigned)rbB[sl];
      f.h[0] = ds_read_tr16_at<(ks * 16) * ROWH * 2>(smem, ad);
      f.h[1] = ds_read_tr16_at<(ks * 16 + 4) * ROWH * 2>(smem, ad);
    };
    static_for<(PD < ITEMS ? PD : ITEMS)>([&](auto I) { read_b(I, fb[decltype(I)::value % NB]); });
    static_for<ITEMS>([&](auto I) {
      constexpr int it = decltype(I)::value, ks = it / 5, sl = it % 5;
      if constexpr (it + PD < ITEMS) read_b(StaticIndex<it + PD>{}, fb[(it + PD) % NB]);
      // LDS reads return in order: once at most the requests issued AFTER item `it` are outstanding (two per item), its
      // fragment -- and the A fragments, requested before every B fragment -- are in their registers
      constexpr int later = (ITEMS - 1 - it) < PD ? (ITEMS - 1 - it) : PD;
      wait_lgkmcnt_for<2 * later>(fb[it % NB]);
      if constexpr (sl == 0) {
        wait_lgkmcnt_for<2 * later>(fa[ks][0]);
        wait_lgkmcnt_for<2 * later>(fa[ks][1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (sl < 4 || tset == 0) {  // wave-uniform: slot 4 of the 4-tap half carries no MFMA
        if constexpr ((ABL & 4) == 0) {
          acc[0][sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks][0].v, fb[it % NB].v, acc[0][sl], 0, 0, 0);
          acc[1][sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks][1].v, fb[it % NB].v, acc[1][sl], 0, 0, 0);
        } else {
          asm volatile("" ::"v"(fa[ks][0].v), "v"(fa[ks][1].v), "v"(fb[it % NB].v));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (it == DPI) {
        issue_step();
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    // the tap windows move on by one step (scalar)
#pragma unroll
    for (int sl = 0; sl < 5; ++sl) {
      rbB[sl] += BKM * ROWH * 2;
      rbB[sl] -= rbB[sl] >= ring_bytes ? ring_bytes : 0;
    }
  }

  // the two pixel groups hold partial sums of the same tiles: group 1 hands its block-0 tiles to group 0, group 0 hands its
  // block-1 tiles to group 1 (through LDS, lane-contiguous), then each flushes the block it collected
  {
    float* xch = reinterpret_cast<float*>(smem);
    static_assert((NY * TILE_Y + (RING_MAX + BKM) * ROWH) * 2 >= 4 * 5 * 16 * 64 * 4, "exchange buffer");
    const int pr = wc + 2 * tset;  // the (column block, tap half) pair this wave shares with its partner of the other group
    __syncthreads();  // every fragment read of the K loop is done
    if (grp == 1) {
#pragma unroll
      for (int sl = 0; sl < 5; ++sl)
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[((pr * 5 + sl) * 16 + r) * 64 + lane] = acc[0][sl][r];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int sl = 0; sl < 5; ++sl)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][sl][r] += xch[((pr * 5 + sl) * 16 + r) * 64 + lane];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int sl = 0; sl < 5; ++sl)
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[((pr * 5 + sl) * 16 + r) * 64 + lane] = acc[1][sl][r];
    }
    __syncthreads();
    if (grp == 1) {
#pragma unroll
      for (int sl = 0; sl < 5; ++sl)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[1][sl][r] += xch[((pr * 5 + sl) * 16 + r) * 64 + lane];
    }
  }

  // partial tile -> workspace slab of this pixel range (plain stores, summed in chunk order by the reduce kernel), or
  // fp32 atomics straight into dW[n][tap*C + c]
  const int K9 = 9 * a.C;
#pragma unroll
  for (int sl = 0; sl < 5; ++sl) {
    if (sl >= ntap) continue;  // wave-uniform
    const int tp = tap0 + sl;
    const int c = c0 + wc * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + grp * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const float v = grp == 0 ? acc[0][sl][r] : acc[1][sl][r];
      if (n < a.N && c < a.C && (ABL & 8) == 0) {
        if (a.ws)
          a.ws[((long)ci * a.N + n) * K9 + tp * a.C + c] = v;
        else
          unsafeAtomicAdd(a.dW + (long)n * a.ldw + tp * a.C + c, v * a.alpha);
      }
    }
  }
}

// ---- fp32x3 form (round 3): the same tap-fused pass for fp32 tensors contracted on the bf16 matrix pipe -------------------
// wgrad.h's wgrad_x3_kernel is the plain-GEMM form: every one of the nine tap tiles of a channel pair re-reads the pixels
// (336-530 us per launch at 192 images, a third of it MFMA time).  Here the pixels of a range are read ONCE, split into bf16
// hi + lo halves on their way from the staging registers into LDS (wgrad_x3_kernel's convert-at-LDS-write: two planes per
// operand, same rows / swizzle / padded pixel order / mirrored ring as the fp16 kernel above), and an item = the four
// transpose reads of a tap's B fragment (hi, lo) + three MFMAs (lo*hi + hi*lo + hi*hi).
// fp32 rows cannot come by LDS-DMA (they have to pass the split), so loads are register-staged TWO steps ahead: the loads of
// step s + 2 are requested before the MFMAs of step s; the rows of step s + 1 (requested a step earlier) are split and stored
// behind those MFMAs -- the rows they overwrite were last read in step s - 1 -- one barrier per step.  (One step ahead, with
// the fp16 kernel's wave roles and 160 accumulator registers: 230-250 us per launch, bound by the latency of the 32 KB a
// workgroup can have in flight; profiles/r03/c20_*.)  To afford the second set of staging registers a wave owns ONE 32-row
// block of the output channels (80 accumulator registers) and walks all 64 pixels of a step: wave = (column block of the
// input channels) x (block of the output channels) x (taps 0-4 / 5-8); no exchange at the end.
// LDS: 2 x (2 dY tiles + ring of 2 x 64 + 2 Gpad rows + mirror) = 128 KB, one workgroup per CU.
constexpr int WGF_X3_RING_MAX = 2 * 64 + 2 * 96;
// Measured on this kernel (profiles/r03/c23_*, c25_*, c26_*), layer2, 231 us per launch including the 11 us reduce launch:
// everything but the step loop 32 us; the loop with MFMAs only 108 us, with fragment reads only 52, with loads + splits + LDS
// stores only 93 (68 steps, latency exposed without the MFMAs; removing them from the full kernel saves 54); the barrier 8.
// Three restructurings left the 231 us unchanged: software-pipelined fragment reads (B fragments two items ahead, counted
// waits), the split + store of the staged rows interleaved piece by piece under the MFMAs of the following step instead of
// behind the last MFMA, and one instead of two steps of prefetch -- as in the fp16 kernel (103 us: 74 without its MFMAs, 88
// without its DMA) the instruction classes of two lockstepped waves per SIMD overlap only partly.  Dependent MFMA chains are
// not it: three back-to-back MFMAs into one accumulator run at 15.7 ns each against 14.4 ns round-robin
// (tools/probes/mfma_dep_probe.hip, c24_*).
// ABL (timing experiments only, MN_WGF_ABLATE in the ablation build; results are wrong): bit 0 = no loads / splits / LDS stores
// after the prologue, bit 1 = fragment reads only in the first step, bit 2 = no MFMAs, bit 3 = no barrier per step.
template <int ABL = 0>
static __global__ void __launch_bounds__(512, 2) wgrad_fused_x3_kernel(WgradFusedArgs a) {
  constexpr int NW = 8, BKM = 8 * NW, ROWH = 64, NY = 2;
  constexpr int TILE_Y = BKM * ROWH;                       // elements of one dY tile plane
  constexpr int XPLANE = (WGF_X3_RING_MAX + BKM) * ROWH;   // ring + mirror of its first block
  // [Y hi tiles][Y lo tiles][X hi ring + mirror][X lo ring + mirror], bf16
  __shared__ unsigned short smem[2 * NY * TILE_Y + 2 * XPLANE] __attribute__((aligned(16)));
  unsigned short* const yh = smem;
  unsigned short* const yl = smem + NY * TILE_Y;
  unsigned short* const xh = smem + 2 * NY * TILE_Y;
  unsigned short* const xl = xh + XPLANE;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wc = wave & 1, nb = (wave >> 1) & 1, tset = wave >> 2;
  const int tap0 = tset * 5, ntap = tset == 0 ? 5 : 4;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int pairs = a.tiles_n * a.tiles_c;
  const int ci = logical / pairs, pr = logical - ci * pairs;
  const int n0 = (pr / a.tiles_c) * 64, c0 = (pr % a.tiles_c) * 64;
  const int j0 = ci * a.chunk;
  const int j1 = min(a.J, j0 + a.chunk);
  const int nsteps = (j1 - j0 + BKM - 1) / BKM;
  const int Gpad = a.Gpad, RING = a.ring;
  const float* __restrict__ dY = reinterpret_cast<const float*>(a.dY);
  const float* __restrict__ X = reinterpret_cast<const float*>(a.X);

  // loader role: rows t/16 and t/16 + 32 of a 64-row block, the 16-byte piece (4 floats) t%16 of the row
  const int ldrow = t >> 4, ldp = t & 15;
  const bool y_ok = n0 + ldp * 4 < a.N, x_ok = c0 + ldp * 4 < a.C;
  auto pixel_of = [&](int j) -> int {
    if ((unsigned)j >= (unsigned)a.J) return -1;
    const int r = fastdiv(j, a.dq), q = j - r * a.Qp;
    const int b = fastdiv(r, a.dp), p = r - b * (a.P + 1);
    return (q < a.Q && p < a.P) ? (b * a.P + p) * a.Q + q : -1;
  };
  struct Staged {
    piece_t y[2], x[2];
  };
  auto load_y = [&](int step, Staged& r) {  // positions [j0 + BKM step, + BKM) of dY (zero past the chunk's end)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = j0 + step * BKM + ldrow + 32 * i;
      const int m = j < j1 ? pixel_of(j) : -1;
      r.y[i] = (m >= 0 && y_ok) ? *reinterpret_cast<const piece_t*>(dY + (long)m * a.ldy + n0 + ldp * 4) : zero_piece();
    }
  };
  auto load_x = [&](int u0, Staged& r) {  // ring-relative rows [u0, u0 + BKM): positions j0 - Gpad + u0 + ..
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = pixel_of(j0 - Gpad + u0 + ldrow + 32 * i);
      r.x[i] = (m >= 0 && x_ok) ? *reinterpret_cast<const piece_t*>(X + (long)m * a.C + c0 + ldp * 4) : zero_piece();
    }
  };
  // 4 floats -> 4 hi + 4 lo bf16 at (row, channels 4 ldp ..) of a plane pair; 16-byte pieces (8 channels) swizzled with row & 3
  auto split_store = [&](piece_t raw, unsigned short* hp, unsigned short* lp, int row) {
    PieceView<float> v;
    v.p = raw;
    union {
      __bf16 b[4];
      u32x2 u;
    } h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h.b[e] = (__bf16)v.e[e];
      l.b[e] = (__bf16)(v.e[e] - (float)h.b[e]);
    }
    const int at = row * ROWH + (((ldp >> 1) ^ wg_swz<8>(row)) * 8) + (ldp & 1) * 4;
    *reinterpret_cast<u32x2*>(hp + at) = h.u;
    *reinterpret_cast<u32x2*>(lp + at) = l.u;
  };
  auto store_y = [&](int step, const Staged& r) {
#pragma unroll
    for (int i = 0; i < 2; ++i) split_store(r.y[i], yh + (step % NY) * TILE_Y, yl + (step % NY) * TILE_Y, ldrow + 32 * i);
  };
  auto store_x = [&](int u0, const Staged& r) {
    const int rr = u0 % RING;  // blocks never straddle the end (RING and u0 are multiples of BKM)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      split_store(r.x[i], xh, xl, rr + ldrow + 32 * i);
      if (rr == 0) split_store(r.x[i], xh, xl, RING + ldrow + 32 * i);  // block 0 also goes to the mirror
    }
  };

  floatx16 acc[5];  // [tap slot: tap = tap0 + slot] of the output-channel block nb
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // transpose-read lane geometry (the fp16 kernel's), as ELEMENT offsets inside a plane; a wave walks all 64 rows of a step
  const int gq = lane >> 4, i16 = lane & 15;
  const int src_row = i16 >> 2, src_chunk = (i16 & 3) * 4 + (gq & 1) * 16;
  const int lrow = (gq >> 1) * 8 + src_row;
  const int colA = nb * 32 + src_chunk;
  const int aA = lrow * ROWH + (((colA >> 3) ^ wg_swz<8>(src_row)) * 8) + (colA & 7);
  const int colB = wc * 32 + src_chunk;
  int xoffB[5], rbB[5];
#pragma unroll
  for (int sl = 0; sl < 5; ++sl) {
    const int tp = tap0 + (sl < ntap ? sl : ntap - 1);
    const int sh = Gpad + (tp / 3 - 1) * a.Qp + (tp % 3 - 1);  // in [0, 2 Gpad] < RING
    const int key = (sh + src_row) & 3;
    xoffB[sl] = lrow * ROWH + (((colB >> 3) ^ (key << 1)) * 8) + (colB & 7);
    rbB[sl] = sh * ROWH;
  }
  const int ring_elems = RING * ROWH;

  // prologue: the halo rows [0, 2 Gpad) and step 0 into LDS; step 1 into the first register set
  Staged st0, st1;
  for (int u0 = 0; u0 < 2 * Gpad; u0 += BKM) {
    load_x(u0, st0);
    store_x(u0, st0);
  }
  if (nsteps > 0) {
    load_y(0, st0);
    load_x(2 * Gpad, st0);
    store_y(0, st0);
    store_x(2 * Gpad, st0);
  }
  if (nsteps > 1) {
    load_y(1, st1);
    load_x(2 * Gpad + BKM, st1);
  }
  __syncthreads();

  // one step: request step s + 2 into `nxt2`, MFMAs of step s, split + store step s + 1 from `nxt1` (requested a step ago)
  auto step = [&](int s, Staged& nxt2, Staged& nxt1) {
    if (s + 2 < nsteps && (ABL & 1) == 0) {
      load_y(s + 2, nxt2);
      load_x(2 * Gpad + BKM * (s + 2), nxt2);
    }
    const unsigned short* tyh = yh + (s % NY) * TILE_Y;
    const unsigned short* tyl = yl + (s % NY) * TILE_Y;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      TrFragB ah, al;
      const bool rd = (ABL & 2) == 0 || s == 0;
#pragma unroll
      for (int hlf = 0; hlf < 2; ++hlf) {
        const int off = aA + (ks * 16 + hlf * 4) * ROWH;
        ah.h[hlf] = ds_read_tr16(tyh + (rd ? off : 0));
        al.h[hlf] = ds_read_tr16(tyl + (rd ? off : 0));
      }
#pragma unroll
      for (int sl = 0; sl < 5; ++sl) {
        TrFragB bh, bl;
        if ((ABL & 2) != 0 && s > 0) {
          bh = ah;
          bl = al;
        } else {
#pragma unroll
          for (int hlf = 0; hlf < 2; ++hlf) {
            const int off = xoffB[sl] + rbB[sl] + (ks * 16 + hlf * 4) * ROWH;
            bh.h[hlf] = ds_read_tr16(xh + off);
            bl.h[hlf] = ds_read_tr16(xl + off);
          }
        }
        if constexpr ((ABL & 4) != 0) {
          asm volatile("" ::"v"(al.v), "v"(ah.v), "v"(bh.v), "v"(bl.v));
        } else if (sl < 4 || tset == 0) {  // wave-uniform: slot 4 of the 4-tap half carries no MFMA
          acc[sl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al.v, bh.v, acc[sl], 0, 0, 0);
          acc[sl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bl.v, acc[sl], 0, 0, 0);
          acc[sl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bh.v, acc[sl], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int sl = 0; sl < 5; ++sl) {
      rbB[sl] += BKM * ROWH;
      rbB[sl] -= rbB[sl] >= ring_elems ? ring_elems : 0;
    }
    if (s + 1 < nsteps && (ABL & 1) == 0) {  // dY tile (s + 1) % 2 and the ring rows [64 (s - 1), 64 s) were last read in step s - 1
      store_y(s + 1, nxt1);
      store_x(2 * Gpad + BKM * (s + 1), nxt1);
    }
    if constexpr ((ABL & 8) == 0) __syncthreads();
  };
  for (int s = 0; s < nsteps; s += 2) {
    step(s, st0, st1);
    if (s + 1 < nsteps) step(s + 1, st1, st0);
  }

  // partial tile -> workspace slab of this pixel range, or fp32 atomics straight into dW[n][tap*C + c]
  const int K9 = 9 * a.C;
#pragma unroll
  for (int sl = 0; sl < 5; ++sl) {
    if (sl >= ntap) continue;  // wave-uniform
    const int tp = tap0 + sl;
    const int c = c0 + wc * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (n < a.N && c < a.C) {
        if (a.ws)
          a.ws[((long)ci * a.N + n) * K9 + tp * a.C + c] = acc[sl][r];
        else
          unsafeAtomicAdd(a.dW + (long)n * a.ldw + tp * a.C + c, acc[sl][r] * a.alpha);
      }
    }
  }
}

// transpose reads requested after item `it` by the time its MFMAs wait (wgrad_fused_h2_kernel): the request groups of the next
// PD items, four reads each, plus the A reads for the first item of a K sub-step (its group carries the A fragments)
constexpr int wgf_h2_reads_after(int it, int pd, int items, int a_reads, int b_reads) {
  int n = 0;
  for (int k = it + 1; k <= it + pd && k < items; ++k) n += (k % 5 == 0) ? b_reads + a_reads : b_reads;
  return n;
}
// ---- h2 form (round 4): the same tap-fused pass for h2 tensors (common.h MMA_H2) ---------------------------------------------
// dY and X arrive ALREADY split into fp16 hi / lo halves, so the four planes of a step (dY hi, dY lo, X hi, X lo: 64 pixels x
// 128 B each) go HBM -> LDS by DMA like the fp16 kernel's two -- no staging registers, no conversion, no LDS stores -- and the
// compute side is the fp32x3 kernel's: wave = (32-column block of the input channels) x (32-row block of the output channels) x
// (taps 0-4 / 5-8), all 64 pixels of a step, an item = the four transpose reads of a tap's B fragment (hi, lo) + three
// v_mfma_f32_32x32x16_f16 (lo*hi + hi*lo + hi*hi).  Transpose reads from inline assembly with counted lgkmcnt waits (the builtin
// would drain the DMA queue in front of every read, wgrad.h): request groups (A fragments of a K sub-step + B fragments of an
// item) run PD items ahead of the MFMAs.  One DMA step in flight (two dY tiles, ring of 2 x 64 + 2 Gpad rows + mirror, four
// planes: 128 KB of LDS), issued behind item DPI of the step before.
constexpr int WGF_H2_RING_MAX = 2 * 64 + 2 * 96;
// BOTH = false (first form): wave = (column block) x (32-row block nb of the output channels) x (tap half), all 64 pixels of
// a step: per K sub-step 4 A reads, per item 4 B reads + 3 MFMAs -- 1.6 transpose reads per MFMA, as much LDS time as MFMA time.
// BOTH = true: the fp16 kernel's roles -- wave = (column block) x (pixel half of every step) x (tap half) and BOTH output-channel
// blocks: a B fragment pair (hi, lo) feeds six MFMAs, 0.93 reads per MFMA; the two pixel halves meet through LDS at the end.
// LO = false (the fp16 mode's default since round 4, MN_WGF_LIGHT): plain fp16 tensors through the same kernel -- no lo
// planes, one MFMA per item -- i.e. the fp16 tap-fused weight gradient with ~110 instead of 219 registers per lane, so that the
// BatchNorm kernels of the main stream can share its CUs' register files.
template <int ABL = 0, int PD = 2, int DPI = 3, bool BOTH = true, bool LO = true>
static __global__ void __launch_bounds__(512, 2) wgrad_fused_h2_kernel(WgradFusedArgs a) {
  constexpr int NW = 8, BKM = 8 * NW, ROWH = 64, NY = 2, NPL = LO ? 2 : 1, EB = LO ? 4 : 2;  // planes; bytes per tensor element
  constexpr int TILE_Y = BKM * ROWH;                      // halves of one dY tile plane
  constexpr int XPLANE = (WGF_H2_RING_MAX + BKM) * ROWH;  // ring + mirror of its first block
  constexpr int YLO = NY * TILE_Y * 2, XLO = XPLANE * 2;  // byte distance hi plane -> lo plane
  static_assert(XLO + (3 * 16 + 4) * ROWH * 2 < 65536, "ds offset field");
  // ONE LDS object: [Y hi tiles][Y lo tiles][X hi ring + mirror][X lo ring + mirror]
  __shared__ half smem[NPL * NY * TILE_Y + NPL * XPLANE] __attribute__((aligned(16)));
  half* const xh = smem + NPL * NY * TILE_Y;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wc = wave & 1, sub = (wave >> 1) & 1, tset = wave >> 2;  // sub: output-channel block (!BOTH) or pixel half (BOTH)
  const int tap0 = tset * 5, ntap = tset == 0 ? 5 : 4;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int pairs = a.tiles_n * a.tiles_c;
  const int ci = logical / pairs, pr = logical - ci * pairs;
  const int n0 = (pr / a.tiles_c) * 64, c0 = (pr % a.tiles_c) * 64;
  const int j0 = ci * a.chunk;
  const int j1 = min(a.J, j0 + a.chunk);
  const int nsteps = (j1 - j0 + BKM - 1) / BKM;
  const int Gpad = a.Gpad, RING = a.ring;

  // the h2 tensors as the DMA sees them: rows of 4 ldy / 4 C bytes
  const __amdgpu_buffer_rsrc_t rsrc_y = make_rsrc(a.dY, (long)a.B * a.P * a.Q * a.ldy * (long)EB);
  // (LO = false with WgradFusedArgs::x_h2: plain fp16 dY against the HI halves of an h2 X -- rows of 4 C bytes, the h2 column map)
  const bool xh2 = LO || a.x_h2;
  const int EBX = xh2 ? 4 : 2;
  const __amdgpu_buffer_rsrc_t rsrc_x = make_rsrc(a.X, (long)a.B * a.P * a.Q * a.C * (long)EBX);
  // DMA role (the fp16 kernel's): row t/8 of a 64-row block, LDS slot t%8 = source piece slot ^ swz(row) = channels
  // 8 piece .. 8 piece + 7 of the 64-channel tile; the lo halves of the same channels lie 64 bytes behind the hi halves
  const int drow = t >> 3, dslot = t & 7;
  const int dpiece = dslot ^ wg_swz<8>(drow);
  const int ych = n0 + dpiece * 8, xch = c0 + dpiece * 8;
  const unsigned ycol = LO ? (unsigned)(((ych >> 5) * 64 + (ych & 31)) * 2) : (unsigned)(ych * 2);
  const unsigned xcol = xh2 ? (unsigned)(((xch >> 5) * 64 + (xch & 31)) * 2) : (unsigned)(xch * 2);
  const bool y_ok = ych < a.N, x_ok = xch < a.C;
  auto pixel_of = [&](int j) -> int {
    if ((unsigned)j >= (unsigned)a.J) return -1;
    const int r = fastdiv(j, a.dq), q = j - r * a.Qp;
    const int b = fastdiv(r, a.dp), p = r - b * (a.P + 1);
    return (q < a.Q && p < a.P) ? (b * a.P + p) * a.Q + q : -1;
  };
  // (an all-ones offset is outside the tensor whatever the scalar offset adds: the bounds check ignores soffset)
  auto issue_y = [&](int step) {
    const int j = j0 + step * BKM + drow;
    const int m = j < j1 ? pixel_of(j) : -1;
    const unsigned off = (m >= 0 && y_ok) ? (unsigned)m * (unsigned)(a.ldy * EB) + ycol : ~0u;
    half* dst = &smem[(step % NY) * TILE_Y + wave * 64 * 8];
    dma16(rsrc_y, off, 0u, dst);
    if constexpr (LO) dma16(rsrc_y, off, 64u, dst + NY * TILE_Y);
  };
  auto issue_x = [&](int u0) {
    const int m = pixel_of(j0 - Gpad + u0 + drow);
    const unsigned off = (m >= 0 && x_ok) ? (unsigned)m * (unsigned)(a.C * EBX) + xcol : ~0u;
    const int rr = u0 % RING;
    half* dst = xh + rr * ROWH + wave * 64 * 8;
    dma16(rsrc_x, off, 0u, dst);
    if constexpr (LO) dma16(rsrc_x, off, 64u, dst + XPLANE);
    if (rr == 0) {  // block 0 also goes to the mirror behind the ring
      dma16(rsrc_x, off, 0u, xh + RING * ROWH + wave * 64 * 8);
      if constexpr (LO) dma16(rsrc_x, off, 64u, xh + RING * ROWH + wave * 64 * 8 + XPLANE);
    }
  };

  constexpr int NBK = BOTH ? 2 : 1;  // output-channel blocks a wave accumulates
  floatx16 acc[NBK][5];              // [block][tap slot: tap = tap0 + slot]
#pragma unroll
  for (int nbk = 0; nbk < NBK; ++nbk)
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nbk][i][r] = 0.f;

  // transpose-read lane geometry (the fp16 kernel's)
  const int gq = lane >> 4, i16 = lane & 15;
  const int src_row = i16 >> 2, src_chunk = (i16 & 3) * 4 + (gq & 1) * 16;
  const int lrow = (BOTH ? sub * 32 : 0) + (gq >> 1) * 8 + src_row;  // BOTH: this wave group's half of the step
  const unsigned lds0 = lds_addr_of(smem);
  unsigned aA[NBK];
#pragma unroll
  for (int nbk = 0; nbk < NBK; ++nbk) {
    const int colA = (BOTH ? nbk : sub) * 32 + src_chunk;
    aA[nbk] = lds0 + (unsigned)((lrow * ROWH + (((colA >> 3) ^ wg_swz<8>(src_row)) * 8) + (colA & 7)) * 2);
  }
  const int colB = wc * 32 + src_chunk;
  unsigned xoffB[5];
  int rbB[5];
#pragma unroll
  for (int sl = 0; sl < 5; ++sl) {
    const int tp = tap0 + (sl < ntap ? sl : ntap - 1);
    const int sh = Gpad + (tp / 3 - 1) * a.Qp + (tp % 3 - 1);  // in [0, 2 Gpad] < RING
    const int key = (sh + src_row) & 3;
    xoffB[sl] = lds0 + (unsigned)((NPL * NY * TILE_Y + lrow * ROWH + ((colB >> 3) ^ (key << 1)) * 8 + (colB & 7)) * 2);
    rbB[sl] = sh * (ROWH * 2);
  }
  const int ring_bytes = RING * ROWH * 2;

  // prologue: the halo rows [0, 2 Gpad), then step 0
  for (int u0 = 0; u0 < 2 * Gpad; u0 += BKM) issue_x(u0);
  if (nsteps > 0) {
    issue_y(0);
    issue_x(2 * Gpad);
  }

  for (int s = 0; s < nsteps; ++s) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // step s visible to everyone; everyone is done with step s-1's reads
    auto issue_step = [&]() {
      if (s + 1 < nsteps && (ABL & 1) == 0) {  // into dY tile (s + 1) % 2 (read in step s - 1) and the ring rows behind the live window
        issue_y(s + 1);
        issue_x(2 * Gpad + BKM * (s + 1));
      }
    };
    if constexpr (DPI < 0) issue_step();
    const unsigned tyoff = (unsigned)((s % NY) * TILE_Y * 2);
    constexpr int NKS = BOTH ? 2 : 4, ITEMS = NKS * 5, NB = PD + 1, AR = 2 * NPL * NBK, BR = 2 * NPL;  // A / B reads of a group
    TrFrag fah[2][NBK], fal[2][NBK], fbh[NB], fbl[NB];
    __builtin_amdgcn_sched_barrier(0);
    // request group of item `it`: the A fragments of its K sub-step (with the sub-step's first item) + its B fragments
    auto request = [&](auto I) {
      constexpr int it = decltype(I)::value, ks = it / 5, sl = it % 5;
      if constexpr (sl == 0) {
        static_for<NBK>([&](auto NBk) {
          constexpr int nbk = decltype(NBk)::value;
          fah[ks & 1][nbk].h[0] = ds_read_tr16_at<(ks * 16) * ROWH * 2>(smem, aA[nbk] + tyoff);
          fah[ks & 1][nbk].h[1] = ds_read_tr16_at<(ks * 16 + 4) * ROWH * 2>(smem, aA[nbk] + tyoff);
          if constexpr (LO) {
            fal[ks & 1][nbk].h[0] = ds_read_tr16_at<YLO + (ks * 16) * ROWH * 2>(smem, aA[nbk] + tyoff);
            fal[ks & 1][nbk].h[1] = ds_read_tr16_at<YLO + (ks * 16 + 4) * ROWH * 2>(smem, aA[nbk] + tyoff);
          }
        });
      }
      const unsigned ad = xoffB[sl] + (unsigned)rbB[sl];
      fbh[it % NB].h[0] = ds_read_tr16_at<(ks * 16) * ROWH * 2>(smem, ad);
      fbh[it % NB].h[1] = ds_read_tr16_at<(ks * 16 + 4) * ROWH * 2>(smem, ad);
      if constexpr (LO) {
        fbl[it % NB].h[0] = ds_read_tr16_at<XLO + (ks * 16) * ROWH * 2>(smem, ad);
        fbl[it % NB].h[1] = ds_read_tr16_at<XLO + (ks * 16 + 4) * ROWH * 2>(smem, ad);
      }
    };
    static_for<(PD < ITEMS ? PD : ITEMS)>([&](auto I) { request(I); });
    static_for<ITEMS>([&](auto I) {
      constexpr int it = decltype(I)::value, ks = it / 5, sl = it % 5;
      if constexpr (it + PD < ITEMS) request(StaticIndex<it + PD>{});
      // LDS reads return in order: item `it` is in its registers once at most the reads requested AFTER it are outstanding
      // (capped at the counter field's 15: a smaller count only waits for more than it has to)
      constexpr int later_all = wgf_h2_reads_after(it, PD, ITEMS, AR, BR), later = later_all > 15 ? 15 : later_all;
      wait_lgkmcnt_for<later>(fbh[it % NB]);
      if constexpr (LO) wait_lgkmcnt_for<later>(fbl[it % NB]);
      if constexpr (sl == 0) {
        static_for<NBK>([&](auto NBk) {
          constexpr int nbk = decltype(NBk)::value;
          wait_lgkmcnt_for<later>(fah[ks & 1][nbk]);
          if constexpr (LO) wait_lgkmcnt_for<later>(fal[ks & 1][nbk]);
        });
      }
      __builtin_amdgcn_sched_barrier(0);
      if (sl < 4 || tset == 0) {  // wave-uniform: slot 4 of the 4-tap half carries no MFMA
        if constexpr ((ABL & 4) == 0) {
          static_for<NBK>([&](auto NBk) {
            constexpr int nbk = decltype(NBk)::value;
            if constexpr (LO) {
              acc[nbk][sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[ks & 1][nbk].v, fbh[it % NB].v, acc[nbk][sl], 0, 0, 0);
              acc[nbk][sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[ks & 1][nbk].v, fbl[it % NB].v, acc[nbk][sl], 0, 0, 0);
            }
            acc[nbk][sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[ks & 1][nbk].v, fbh[it % NB].v, acc[nbk][sl], 0, 0, 0);
          });
        } else {
          asm volatile("" ::"v"(fah[ks & 1][0].v), "v"(fbh[it % NB].v));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (it == DPI) {
        issue_step();
        __builtin_amdgcn_sched_barrier(0);
      }
    });
#pragma unroll
    for (int sl = 0; sl < 5; ++sl) {
      rbB[sl] += BKM * ROWH * 2;
      rbB[sl] -= rbB[sl] >= ring_bytes ? ring_bytes : 0;
    }
  }

  if constexpr (BOTH) {
    // the two pixel halves hold partial sums of the same tiles: half 1 hands its block-0 tiles to half 0, half 0 its block-1
    // tiles to half 1 (through LDS, lane-contiguous), then each flushes the block it collected (the fp16 kernel's exchange)
    float* xch = reinterpret_cast<float*>(smem);
    static_assert((NPL * NY * TILE_Y + NPL * XPLANE) * 2 >= 4 * 5 * 16 * 64 * 4, "exchange buffer");
    const int prw = wc + 2 * tset;
    __syncthreads();  // every fragment read of the K loop is done
    if (sub == 1) {
#pragma unroll
      for (int sl = 0; sl < 5; ++sl)
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[((prw * 5 + sl) * 16 + r) * 64 + lane] = acc[0][sl][r];
    }
    __syncthreads();
    if (sub == 0) {
#pragma unroll
      for (int sl = 0; sl < 5; ++sl)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][sl][r] += xch[((prw * 5 + sl) * 16 + r) * 64 + lane];
    }
    __syncthreads();
    if (sub == 0) {
#pragma unroll
      for (int sl = 0; sl < 5; ++sl)
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[((prw * 5 + sl) * 16 + r) * 64 + lane] = acc[NBK - 1][sl][r];
    }
    __syncthreads();
    if (sub == 1) {
#pragma unroll
      for (int sl = 0; sl < 5; ++sl)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[NBK - 1][sl][r] += xch[((prw * 5 + sl) * 16 + r) * 64 + lane];
    }
  }

  // partial tile -> workspace slab of this pixel range, or fp32 atomics straight into dW[n][tap*C + c]; this wave flushes the
  // output-channel block `sub` (BOTH: the block it collected)
  const int K9 = 9 * a.C;
#pragma unroll
  for (int sl = 0; sl < 5; ++sl) {
    if (sl >= ntap) continue;  // wave-uniform
    const int tp = tap0 + sl;
    const int c = c0 + wc * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const float v = (BOTH && sub == 1) ? acc[NBK - 1][sl][r] : acc[0][sl][r];
      if (n < a.N && c < a.C) {
        if (a.ws)
          a.ws[((long)ci * a.N + n) * K9 + tp * a.C + c] = v;
        else
          unsafeAtomicAdd(a.dW + (long)n * a.ldw + tp * a.C + c, v * a.alpha);
      }
    }
  }
}

// dW[n][k] += alpha * sum over chunks of ws[chunk][n][k], chunks added in index order.  blockIdx.y splits the chunk
// range when there are few columns and many chunks (layer1: 36 864 columns x 512 chunks); with more than one group the
// groups meet in dW through atomics, with one group (layers 3-4) the sum is a plain read-modify-write.
static __global__ void __launch_bounds__(256) wgrad_fused_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dW,
                                                                         int N, int K9, int ldw, int chunks,
                                                                         int chunks_per_group, float alpha) {
  const long quads = (long)N * K9 / 4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= quads) return;
  const int c_begin = blockIdx.y * chunks_per_group;
  const int c_end = min(chunks, c_begin + chunks_per_group);
  floatx4 s = {0.f, 0.f, 0.f, 0.f};
  const floatx4* p = reinterpret_cast<const floatx4*>(ws) + (long)c_begin * quads + i;
  for (int c = c_begin; c < c_end; ++c, p += quads) s += *p;
  const long e = i * 4;
  const int n = (int)(e / K9), k = (int)(e - (long)n * K9);
  float* d = dW + (long)n * ldw + k;
  if (gridDim.y == 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] += s[j] * alpha;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) unsafeAtomicAdd(d + j, s[j] * alpha);
  }
}

// Applies to the weight gradient of a 3x3 stride-1 pad-1 convolution of an NHWC fp16 tensor whose channel counts are
// multiples of 8 (ResNet-34: 29 of the 37 conv layers, 93 % of the weight-gradient FLOPs).
inline bool wgrad_fused_applies(const WgradArgs& a) {
  const GatherGeom& g = a.g;
  return g.R == 3 && g.S == 3 && g.mul_p == 1 && g.mul_q == 1 && g.div == 1 && g.rsign == 1 && g.ssign == 1 && g.off_h == -1 &&
         g.off_w == -1 && g.P == g.Hi && g.Q == g.Wi && g.C % 8 == 0 && g.N % 8 == 0 && a.colmap == nullptr && a.ldw >= 9 * g.C &&
         ((g.Q + 2 + 31) / 32) * 32 <= 96 && (long)g.M * a.ldy * 2 < 0xfffffff0l &&
         (long)g.M * x_row_halves(a.x_h2, g.C) * 2 < 0xfffffff0l && (!a.x_h2 || g.C % 32 == 0) &&
         (long)g.B * (g.P + 1) * (g.Q + 1) < (1L << 30);
}

// the fp32x3 form: fp32 tensors (16-byte pieces of 4 floats), otherwise the same shapes
inline bool wgrad_fused_x3_applies(const WgradArgs& a) {
  const GatherGeom& g = a.g;
  return g.mma == MMA_BF16X3 && g.R == 3 && g.S == 3 && g.mul_p == 1 && g.mul_q == 1 && g.div == 1 && g.rsign == 1 &&
         g.ssign == 1 && g.off_h == -1 && g.off_w == -1 && g.P == g.Hi && g.Q == g.Wi && g.C % 4 == 0 && g.N % 4 == 0 &&
         a.ldy % 4 == 0 && a.colmap == nullptr && a.ldw >= 9 * g.C && ((g.Q + 2 + 31) / 32) * 32 <= 96 &&
         (long)g.B * (g.P + 1) * (g.Q + 1) < (1L << 30);
}

// the h2 form: h2 tensors (channel counts in whole 32-channel groups), otherwise the same shapes
inline bool wgrad_fused_h2_applies(const WgradArgs& a) {
  const GatherGeom& g = a.g;
  return g.mma == MMA_H2 && g.R == 3 && g.S == 3 && g.mul_p == 1 && g.mul_q == 1 && g.div == 1 && g.rsign == 1 && g.ssign == 1 &&
         g.off_h == -1 && g.off_w == -1 && g.P == g.Hi && g.Q == g.Wi && g.C % 32 == 0 && g.N % 32 == 0 && a.ldy % 32 == 0 &&
         a.colmap == nullptr && a.ldw >= 9 * g.C && ((g.Q + 2 + 31) / 32) * 32 <= 96 && (long)g.M * a.ldy * 4 < 0xfffffff0l &&
         (long)g.M * g.C * 4 < 0xfffffff0l && (long)g.B * (g.P + 1) * (g.Q + 1) < (1L << 30);
}

inline void wgrad_fused_reduce(const WgradFusedArgs& a, hipStream_t stream) {
  if (!a.ws) return;
#ifdef MN_ABLATION_BUILD
  static const bool skip = getenv("MN_WGF_SKIP_REDUCE") != nullptr;  // timing experiment: what the reduce launches cost a step
  if (skip) return;
#endif
  const int K9 = 9 * a.C;
  const long quads = (long)a.N * K9 / 4;
  // enough threads to pull the slabs at full bandwidth: split the chunk range over blockIdx.y while columns are few
  int groups = (int)(131072 / quads);
  if (groups > a.nchunks) groups = a.nchunks;
  if (groups < 1 || a.det) groups = 1;  // one group: every element is a plain ordered sum
  const int per = cdiv(a.nchunks, groups);
  groups = cdiv(a.nchunks, per);
  hipLaunchKernelGGL(wgrad_fused_reduce_kernel, dim3(cdiv(quads, 256), groups), dim3(256), 0, stream, (const float*)a.ws, a.dW,
                     a.N, K9, a.ldw, a.nchunks, per, a.alpha);
}

// floats of workspace a launch with `blocks` workgroups needs (a workgroup's partial tile is 64 x 9 x 64 at most)
inline long wgrad_fused_ws_floats(int blocks) { return (long)blocks * 64 * 9 * 64; }
constexpr int WGF_BLOCKS = 512;  // workgroups per launch: two per CU (the register budget), i.e. one round of the chip

// form: 0 = fp16 tensors, 1 = fp32 tensors on the bf16 pipe (x3), 2 = h2 tensors
inline void launch_wgrad_fused(const WgradArgs& w, int target_blocks, hipStream_t stream, int form = 0) {
#ifdef MN_ABLATION_BUILD
  static const bool skip_all = getenv("MN_ABL_SKIP_WGF") != nullptr;  // timing experiment: the step without the fused weight gradients
  if (skip_all) return;
#endif
  const GatherGeom& g = w.g;
  constexpr int NW = 8, BKM = 8 * NW;  // (the 4-wave form -- two 256-thread workgroups per CU, twice the partial tiles -- is not launched)
  WgradFusedArgs a;
  a.dY = reinterpret_cast<const half*>(w.dY);
  a.X = reinterpret_cast<const half*>(w.X);
  a.dW = w.dW;
  a.ws = nullptr;
  a.ldy = w.ldy;
  a.ldw = w.ldw;
  a.B = g.B; a.P = g.P; a.Q = g.Q; a.C = g.C; a.N = g.N;
  a.Qp = g.Q + 1;
  a.J = g.B * (g.P + 1) * (g.Q + 1);
  a.tiles_n = cdiv(g.N, 64);
  a.tiles_c = cdiv(g.C, 64);
  a.Gpad = ((g.Q + 2 + 15) / 16) * 16;
  a.Gpad = ((a.Gpad + 31) / 32) * 32;  // 2 Gpad must be a multiple of the 64-row DMA block
  constexpr int D = 3;  // DMA steps in flight (the fp32x3 form stages one step ahead through registers)
  // fp16 tensors: the low-register form of the kernel (134 instead of 219 registers per lane, 64 instead of 96 KB of LDS) is the
  // default since round 4: equal per launch (111 / 102 / 98 / 106 us against 106 / 105 / 101 / 108 at layers 1-4) and 0.17 ms
  // faster per step, 13.57 vs 13.75 ms (same-box A/B, two interleaved repeats, profiles/r04/c10_*): two 224-register waves per
  // SIMD left no room for a BatchNorm-backward wave (122 registers) of the main stream, so the two streams time-sliced the CUs.
  // MN_WGF_LIGHT=0 restores wgrad_fused_kernel.
  static const bool light = !(getenv("MN_WGF_LIGHT") && atoi(getenv("MN_WGF_LIGHT")) == 0);
  if (form == 0 && (light || w.x_h2)) form = 3;  // (the hi halves of an h2 X: the low-register kernel is the one that reads them)
  a.x_h2 = w.x_h2;
  a.ring = ((form != 0 ? 1 : D) + 1) * BKM + 2 * a.Gpad;
  a.dq = make_fastdiv(a.Qp);
  a.dp = make_fastdiv(g.P + 1);
  a.alpha = w.alpha;
  a.det = w.det;
  // pixel ranges: one round of resident workgroups (two 4-wave or one 8-wave workgroup per CU), each range at least 8
  // halos long (the ring prologue fetches 2 Gpad + D steps of rows that belong to the neighbouring ranges)
  const int pairs = a.tiles_n * a.tiles_c;
  (void)target_blocks;  // the plain-GEMM form's split count; this kernel wants exactly one round of resident workgroups
  int chunks = cdiv(WGF_BLOCKS / 2, pairs);  // one 512-thread workgroup per CU
  const int min_chunk = 16 * a.Gpad;
  if ((long)chunks * min_chunk > a.J) chunks = (int)(a.J / min_chunk);
  if (chunks < 1) chunks = 1;
  a.chunk = cdiv(cdiv(a.J, chunks), BKM) * BKM;
  a.nchunks = cdiv(a.J, a.chunk);
  if (w.ws && (long)a.nchunks * a.N * 9 * a.C <= w.ws_floats) a.ws = w.ws;
  static const bool trace = getenv("MN_TRACE_DISPATCH") != nullptr;
  if (trace)
    fprintf(stderr, "wgrad_fused<%d waves>: B %d P %d Q %d C %d N %d  chunk %d x %d chunks x %d pairs, ring %d rows, ws %d\n", NW,
            a.B, a.P, a.Q, a.C, a.N, a.chunk, a.nchunks, pairs, a.ring, a.ws != nullptr);
  const dim3 grid(a.nchunks * pairs), block(NW * 64);
  const bool x3 = form == 1;
  if (form == 3) {  // fp16 tensors through the low-register kernel
    hipLaunchKernelGGL((wgrad_fused_h2_kernel<0, 3, 3, false, false>), grid, block, 0, stream, a);
    wgrad_fused_reduce(a, stream);
    return;
  }
  if (form == 2) {
#ifdef MN_ABLATION_BUILD
    static const int ablh = getenv("MN_WGF_ABLATE") ? atoi(getenv("MN_WGF_ABLATE")) : 0;
    if (ablh == 1) { hipLaunchKernelGGL((wgrad_fused_h2_kernel<1>), grid, block, 0, stream, a); wgrad_fused_reduce(a, stream); return; }
    if (ablh == 4) { hipLaunchKernelGGL((wgrad_fused_h2_kernel<4>), grid, block, 0, stream, a); wgrad_fused_reduce(a, stream); return; }
    if (ablh == 5) { hipLaunchKernelGGL((wgrad_fused_h2_kernel<5>), grid, block, 0, stream, a); wgrad_fused_reduce(a, stream); return; }
#endif
    // one output-channel block per wave (150 registers) by default: the both-blocks form is 5-10 % faster per launch (224 / 195 /
    // 199 / 210 us against 236 / 220 / 226 / 223 at layers 1-4) and 0.7 ms SLOWER per step, 31.22 vs 30.52 ms: at 240 registers x 2
    // waves per SIMD no BatchNorm wave of the main stream fits beside it (profiles/r04/c9_*)
    static const int roles = getenv("MN_WGF_H2_ROLES") ? atoi(getenv("MN_WGF_H2_ROLES")) : 0;
    if (roles == 0)
      hipLaunchKernelGGL((wgrad_fused_h2_kernel<0, 2, 3, false>), grid, block, 0, stream, a);
    else if (roles == 2)
      hipLaunchKernelGGL((wgrad_fused_h2_kernel<0, 1, 3, true>), grid, block, 0, stream, a);
    else
      hipLaunchKernelGGL((wgrad_fused_h2_kernel<0, 2, 3, true>), grid, block, 0, stream, a);
    wgrad_fused_reduce(a, stream);
    return;
  }
  if (x3) {
#ifdef MN_ABLATION_BUILD
    static const int ablx = getenv("MN_WGF_ABLATE") ? atoi(getenv("MN_WGF_ABLATE")) : 0;
#define MN_WGFX_ABL(V_)                                                          \
  if (ablx == V_) {                                                              \
    hipLaunchKernelGGL((wgrad_fused_x3_kernel<V_>), grid, block, 0, stream, a); \
    wgrad_fused_reduce(a, stream);                                               \
    return;                                                                      \
  }
    MN_WGFX_ABL(1) MN_WGFX_ABL(2) MN_WGFX_ABL(4) MN_WGFX_ABL(8) MN_WGFX_ABL(3) MN_WGFX_ABL(5) MN_WGFX_ABL(6) MN_WGFX_ABL(7) MN_WGFX_ABL(15)
#undef MN_WGFX_ABL
#endif
    hipLaunchKernelGGL((wgrad_fused_x3_kernel<0>), grid, block, 0, stream, a);
    wgrad_fused_reduce(a, stream);
    return;
  }
#ifdef MN_ABLATION_BUILD
  static const int abl = getenv("MN_WGF_ABLATE") ? atoi(getenv("MN_WGF_ABLATE")) : 0;
  {
    switch (abl) {
      case 1: hipLaunchKernelGGL((wgrad_fused_kernel<D, 1>), grid, block, 0, stream, a); wgrad_fused_reduce(a, stream); return;
      case 2: hipLaunchKernelGGL((wgrad_fused_kernel<D, 2>), grid, block, 0, stream, a); wgrad_fused_reduce(a, stream); return;
      case 4: hipLaunchKernelGGL((wgrad_fused_kernel<D, 4>), grid, block, 0, stream, a); wgrad_fused_reduce(a, stream); return;
      case 8: hipLaunchKernelGGL((wgrad_fused_kernel<D, 8>), grid, block, 0, stream, a); return;
      case 9: hipLaunchKernelGGL((wgrad_fused_kernel<D, 9>), grid, block, 0, stream, a); return;
      default: break;
    }
  }
#endif
  // (B-fragment read-ahead PD = 2 / 4 / 6 and DMA depth 1-3 measured equal in round 2; depth 4 measured slower in round 3:
  //  104-112 vs 101-110 us per launch, profiles/r03/c10_wgf_depth.txt; PD = 3, D = 3)
  hipLaunchKernelGGL((wgrad_fused_kernel<D>), grid, block, 0, stream, a);
  wgrad_fused_reduce(a, stream);
}

}  // namespace mn
