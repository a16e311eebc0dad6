// Hardware probe: achievable v_mfma_f32_32x32x16_f16 rate (no memory traffic), for calibrating rooflines.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int NACC, bool BARRIER>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  half8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f - threadIdx.x * 0.002f); }
  floatx16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (BARRIER) __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int rep = 0; rep < 16 / NACC; ++rep)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][7];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, bool BARRIER>
void run(const char* name, int blocks) {
  float* d; hipMalloc(&d, blocks * 256 * 4);
  int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, BARRIER>), dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, BARRIER>), dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
  printf("%-40s blocks=%5d  %8.3f ms  %8.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
  hipFree(d);
}
int main() {
  run<4, false>("4 acc, no barrier, 1 block/CU", 256);
  run<4, false>("4 acc, no barrier, 2 blocks/CU", 512);
  run<4, false>("4 acc, no barrier, 4 blocks/CU", 1024);
  run<4, true>("4 acc, barrier per 16 MFMA, 2 blocks/CU", 512);
  run<2, false>("2 acc, no barrier, 2 blocks/CU", 512);
  run<1, false>("1 acc, no barrier, 2 blocks/CU", 512);
  return 0;
}
