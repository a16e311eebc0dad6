// Hardware probe: what do back-to-back DEPENDENT MFMAs (same accumulator) cost on gfx950?  The x3 modes issue three
// v_mfma_f32_32x32x16_{f16,bf16} in a row into one accumulator (lo*hi + hi*lo + hi*hi).  One wave per SIMD (256 threads per
// workgroup, one workgroup per CU) runs N MFMAs with the accumulator index cycling over G accumulators: G = 1 is a fully
// dependent chain, G = 3 / 4 / 6 leave 2 / 3 / 5 independent MFMAs between two uses of an accumulator.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int G, int RUN>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(1.0f + threadIdx.x * 1e-3f); b[e] = (__bf16)(0.5f); }
  floatx16 c[G];
  for (int g = 0; g < G; ++g) for (int r = 0; r < 16; ++r) c[g][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int r = 0; r < RUN; ++r)  // RUN consecutive MFMAs into the same accumulator, then the next accumulator
        c[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[g], 0, 0, 0);
  }
  float s = 0.f;
  for (int g = 0; g < G; ++g) s += c[g][0] + c[g][15];
  if (s == 12345.f) out[0] = s;
}
template <int G, int RUN>
static void run(const char* what) {
  float* d; hipMalloc(&d, 64);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<G, RUN>), dim3(256), dim3(256), 0, 0, d, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<G, RUN>), dim3(256), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * G * RUN;  // MFMAs per wave
  printf("%-58s %7.2f ns per MFMA per SIMD  (%.2f PF whole chip)\n", what, ms * 1e6 / n, 32768.0 * n * 1024 / (ms * 1e-3) / 1e15);
  hipFree(d);
}
int main() {
  run<1, 6>("one accumulator (fully dependent chain)");
  run<2, 3>("2 accumulators, runs of 3 (the x3 item order)");
  run<5, 3>("5 accumulators, runs of 3 (x3 weight gradient, 5 taps)");
  run<2, 1>("2 accumulators alternating");
  run<3, 1>("3 accumulators round-robin");
  run<4, 1>("4 accumulators round-robin");
  run<6, 1>("6 accumulators round-robin");
  return 0;
}
