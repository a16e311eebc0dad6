// Hardware probe: do a wave's LDS fragment reads overlap its own MFMAs?
// The layer1 ablations (profiles/r02/c14_halo_pp_ablation.txt) show the MFMA loop of halo_pp.h taking the SUM of its
// fragment-read time and its MFMA time (4.5 us per tile = 2.65 + 1.9), not the maximum.  This probe runs that loop shape
// alone -- per K-sub-step NR ds_read_b128 (issued PD sub-steps ahead into rotating register slots) and NM dependent-free
// MFMAs, no barriers, no global memory -- with the accumulators in VGPRs (what hipcc chooses) or in AccVGPRs (inline asm).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/lds_mfma_probe tools/probes/lds_mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned piece_t __attribute__((ext_vector_type(4)));
union PV {
  piece_t p;
  half8 v;
};

// MODE 0: reads + MFMA (VGPR accumulators)   1: reads + MFMA (AccVGPR accumulators)   2: MFMA only   3: reads only
// TM x TN register tile: TM + TN reads and TM * TN MFMAs per sub-step
template <int MODE, int TM, int TN, int PD>
__global__ void __launch_bounds__(256, 2) k(float* out, int iters) {
  __shared__ piece_t lds[4096];  // 64 KB
  const int t = threadIdx.x, lane = t & 63;
  for (int i = t; i < 4096; i += 256) lds[i] = piece_t{(unsigned)i * 2654435761u, 0x3c003c00u, (unsigned)i, 0x38003800u};
  __syncthreads();
  floatx16 acc[TM][TN];
  for (int i = 0; i < TM; ++i)
    for (int j = 0; j < TN; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  constexpr int SLOTS = 4;
  PV fa[SLOTS][TM], fb[SLOTS][TN];
  for (int s = 0; s < SLOTS; ++s) {
    for (int i = 0; i < TM; ++i) fa[s][i].p = lds[(lane + 64 * i) & 4095];
    for (int j = 0; j < TN; ++j) fb[s][j].p = lds[(lane + 64 * j + 512) & 4095];
  }
  auto load = [&](int step, int slot) __attribute__((always_inline)) {
    if (MODE == 2) return;
    // consecutive lanes read consecutive 16-byte pieces: conflict-free ds_read_b128
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[slot][i].p = lds[(step * 64 + lane + i * 1024) & 4095];
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[slot][j].p = lds[(step * 64 + lane + j * 1024 + 2048) & 4095];
  };
  int step = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks, ++step) {
      load(step + PD, (ks + PD) & 3);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (MODE == 0 || MODE == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[ks][j].v, fa[ks][i].v, acc[i][j], 0, 0, 0);
          } else if (MODE == 1) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fb[ks][j].v), "v"(fa[ks][i].v));
          } else {
            asm volatile("" ::"v"(fa[ks][i].p), "v"(fb[ks][j].p));
          }
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (MODE == 1) {
    asm volatile("s_nop 15");
    asm volatile("s_nop 15");
  }
  float s = 0;
  for (int i = 0; i < TM; ++i)
    for (int j = 0; j < TN; ++j) s += acc[i][j][0] + acc[i][j][9];
  out[blockIdx.x * 256 + t] = s;
}

template <int MODE, int TM, int TN, int PD>
void run(const char* name, int blocks) {
  float* d;
  hipMalloc(&d, blocks * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, TM, TN, PD>), dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, TM, TN, PD>), dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double substeps = (double)iters * 4;
  const double flops = (double)blocks * 4 * substeps * TM * TN * 2.0 * 32 * 32 * 16;
  printf("%-58s blocks=%4d  %7.1f ns / sub-step  %7.1f TFLOP/s  LDS %6.1f B/clk/CU @2.4GHz\n", name, blocks, ms * 1e6 / substeps,
         MODE == 3 ? 0.0 : flops / ms / 1e9,
         MODE == 2 ? 0.0 : (double)(blocks / 256.0) * 4 * (TM + TN) * 1024 / (ms * 1e6 / substeps * 2.4));
  hipFree(d);
}

int main() {
  printf("2x2 register tile (4 reads, 4 MFMAs per sub-step), read-ahead 2; one workgroup per CU = one wave per SIMD\n");
  run<2, 2, 2, 2>("MFMA only", 256);
  run<3, 2, 2, 2>("LDS reads only", 256);
  run<0, 2, 2, 2>("reads + MFMA, accumulators in VGPRs", 256);
  run<1, 2, 2, 2>("reads + MFMA, accumulators in AccVGPRs", 256);
  run<0, 2, 2, 1>("reads + MFMA, VGPR accumulators, read-ahead 1", 256);
  run<0, 2, 2, 3>("reads + MFMA, VGPR accumulators, read-ahead 3", 256);
  printf("... two workgroups per CU = two waves per SIMD\n");
  run<2, 2, 2, 2>("MFMA only", 512);
  run<3, 2, 2, 2>("LDS reads only", 512);
  run<0, 2, 2, 2>("reads + MFMA, accumulators in VGPRs", 512);
  run<1, 2, 2, 2>("reads + MFMA, accumulators in AccVGPRs", 512);
  printf("4x2 register tile (6 reads, 8 MFMAs per sub-step)\n");
  run<2, 4, 2, 2>("MFMA only", 256);
  run<3, 4, 2, 2>("LDS reads only", 256);
  run<0, 4, 2, 2>("reads + MFMA, accumulators in VGPRs", 256);
  run<1, 4, 2, 2>("reads + MFMA, accumulators in AccVGPRs", 256);
  run<0, 4, 2, 2>("reads + MFMA, accumulators in VGPRs", 512);
  run<1, 4, 2, 2>("reads + MFMA, accumulators in AccVGPRs", 512);
  return 0;
}
