// Hardware probe: what the L2 -> LDS path (buffer_load_dwordx4 ... lds) sustains per CU, alone and next to MFMA work.
// DESIGN.md 5.1 item 3 prices the convolution kernels against ~20 B/clk/CU inferred from whole-kernel timings; this
// measures it directly.  Every workgroup streams tiles from its own L2-resident window (so HBM is not the limit) into a
// two-buffer LDS ring, one `s_waitcnt vmcnt(0)` + barrier per tile as in igemm.h, with MFMAS matrix instructions per
// wave between the barrier and the next wait.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/dma_probe tools/probes/dma_probe.hip && tools/probes/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef void __attribute__((address_space(3)))* las_ptr_t;

constexpr unsigned WINDOW = 1u << 16;  // 64 KiB per workgroup, re-read every tile or two: 16-64 MB in all, L2 / MALL hits
template <int NT, int PIECES, int MFMAS>
__global__ void __launch_bounds__(NT) k(const char* __restrict__ src, float* out, int iters) {
  constexpr int TILE = NT * PIECES * 16;  // bytes per tile
  __shared__ char lds[2 * TILE];
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src + (long)blockIdx.x * WINDOW), (short)0, (int)WINDOW, 0x00020000);
  half8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(t * 0.001f + e); b[e] = (_Float16)(e * 0.5f - t * 0.002f); }
  floatx16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  unsigned off = (unsigned)t * 16u;
  auto issue = [&](int buf) {
#pragma unroll
    for (int p = 0; p < PIECES; ++p)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (las_ptr_t)(lds + buf * TILE + (p * NT + wave * 64) * 16), 16,
                                               (int)((off + (unsigned)(p * NT * 16)) & (WINDOW - 1)), 0, 0, 0);
    off = (off + TILE) & (WINDOW - 1);
  };
  issue(0);
  for (int it = 0; it < iters; ++it) {
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));  // vmcnt(0)
    __builtin_amdgcn_s_barrier();
    issue((it + 1) & 1);
#pragma unroll
    for (int m = 0; m < MFMAS; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
  }
  float s = lds[(t * 16) % TILE];
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  out[blockIdx.x * NT + t] = s;
}

template <int NT, int PIECES, int MFMAS>
void run(const char* name, int blocks_per_cu, int cus, double mhz) {
  const long window = WINDOW;
  const int blocks = blocks_per_cu * cus, iters = 400;
  char* src; float* out;
  hipMalloc(&src, (size_t)blocks * window); hipMemset(src, 1, (size_t)blocks * window);
  hipMalloc(&out, (size_t)blocks * NT * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NT, PIECES, MFMAS>), dim3(blocks), dim3(NT), 0, 0, src, out, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NT, PIECES, MFMAS>), dim3(blocks), dim3(NT), 0, 0, src, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)blocks * iters * NT * PIECES * 16;
  const double flops = (double)blocks * (NT / 64) * iters * MFMAS * 2.0 * 32 * 32 * 16;
  printf("%-58s %7.3f ms  %7.2f TB/s  %6.1f B/clk/CU @%4.0f MHz nominal  %7.1f TFLOP/s\n", name, ms, bytes / ms / 1e9,
         bytes / (ms * 1e-3) / cus / (mhz * 1e6), mhz, flops / ms / 1e9);
  hipFree(src); hipFree(out);
}

int main() {
  int dev = 0, cus = 256, khz = 2400000;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev);
  const double mhz = khz / 1e3;
  printf("CUs %d, nominal clock %.0f MHz (sustained clocks under MFMA load are lower: compare rows, not absolutes)\n", cus, mhz);
  run<256, 8, 0>("256 thr, 32 KB tiles, DMA only, 2 WG/CU", 2, cus, mhz);
  run<256, 8, 0>("256 thr, 32 KB tiles, DMA only, 4 WG/CU", 4, cus, mhz);
  run<256, 8, 16>("256 thr, 32 KB tiles + 16 MFMA/wave/tile (128x128 igemm ratio), 2 WG/CU", 2, cus, mhz);
  run<768, 6, 0>("768 thr, 72 KB tiles, DMA only, 1 WG/CU", 1, cus, mhz);
  run<768, 6, 24>("768 thr, 72 KB tiles + 24 MFMA/wave/tile (288x256 igemm ratio), 1 WG/CU", 1, cus, mhz);
  run<768, 3, 24>("768 thr, 36 KB tiles + 24 MFMA/wave/tile (half the bytes per MAC), 1 WG/CU", 1, cus, mhz);
  run<256, 4, 8>("256 thr, 16 KB tiles + 8 MFMA/wave/tile (weight-gradient ratio), 3 WG/CU", 3, cus, mhz);
  return 0;
}
