// The BatchNorm kernels of the fp16x2m step (elementwise_h2.h, the library's own source) timed ALONE on rotating buffer sets, next to
// stream_probe's ceilings for the same access patterns: is a pass slow by itself, or only inside the step?
//   hipcc --offload-arch=gfx950 -O3 -I geomapnet_amd/csrc -o tools/probes/bn_probe tools/probes/bn_probe.hip && tools/probes/bn_probe
#include "elementwise_h2.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

// ---- experiment (round 6, NOT in the library): the forward apply with 4 channels per thread and one piece per thread -----------------
// Alone it is 10-14 % faster than the library's persistent 8-channel form on layer1-2's tensors (profiles/r06/c15_to_c20_*); inside the
// step the same-box A/B showed 18.39 / 18.42 ms against 18.42 / 18.44: the launch tail it shortens is hidden behind the next launch there.
namespace mn {
// four elements (the 4-channel elementwise kernels): the same instructions, the same bits as split8_f16's
__device__ __forceinline__ void split4_f16(const float (&x)[4], Half4View& hi, Half4View& lo) {
  unsigned H0 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x[0], x[1]));
  unsigned H1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x[2], x[3]));
  unsigned L0, L1;
  asm volatile(
      "v_fma_mixlo_f16 %0, %2, 1.0, -%6 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %1, %4, 1.0, -%7 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %3, 1.0, -%6 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %1, %5, 1.0, -%7 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(L0), "=&v"(L1)
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(H0), "v"(H1));
  hi.p[0] = H0;
  hi.p[1] = H1;
  lo.p[0] = L0;
  lo.p[1] = L1;
}
// The same pass with FOUR channels per thread (h2 tensors, q = 0): a wave's load of y is one contiguous kilobyte -- the 8-channel form
// above reads two ADJACENT pieces per lane, so each of its load instructions touches every line of a 2 KB span and uses half of it -- and
// the per-channel table costs half the registers.  8-byte accesses to the h2 halves and to the record; same arithmetic, same bits.
static __global__ void __launch_bounds__(256) bn_apply_h2v4_kernel(const float* __restrict__ y, const float* __restrict__ coef,
                                                                   const half* __restrict__ res, half* __restrict__ out, long npieces,
                                                                   int C, int relu, half* __restrict__ rec,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd) {
  constexpr int VEC = 4;
  const int cpr = C / VEC;
  __shared__ floatx4 tab[512];  // [e][piece] -> (scale, shift, mean, invstd)
  for (int c = threadIdx.x; c < C; c += 256) {
    const floatx4 v = {coef[c], coef[C + c], rec ? mean[c] : 0.f, rec ? invstd[c] : 0.f};
    tab[(c % VEC) * cpr + c / VEC] = v;
  }
  __syncthreads();
  const int cp = (int)(threadIdx.x % cpr);  // loop invariant: the grid stride is a multiple of cpr (a power of two <= 128)
  float sc[VEC], sh[VEC], mu[VEC], is[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const floatx4 v = tab[e * cpr + cp];
    sc[e] = v[0];
    sh[e] = v[1];
    mu[e] = v[2];
    is[e] = v[3];
  }
  const int hoff = (cp >> 3) * 64 + (cp & 7) * 4;  // h2_index(row, C, cp * 4) - row * 2C
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npieces; i += (long)gridDim.x * blockDim.x) {
    const long row = i / cpr;
    PieceView<float> v;
    v.p = MN_LOAD_LAST(reinterpret_cast<const piece_t*>(y) + i);
    float f[VEC] = {v.e[0], v.e[1], v.e[2], v.e[3]};
    Half4View rh, rl;
    if (res) {
      const half* p = res + row * 2 * C + hoff;
      rh.p = *reinterpret_cast<const u32x2*>(p);
      rl.p = *reinterpret_cast<const u32x2*>(p + 32);
    }
    Half4View rc;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float xh = (f[e] - mu[e]) * is[e];
      f[e] = f[e] * sc[e] + sh[e];
      rc.e[e] = rec_pack(xh, !res && relu ? f[e] > 0.f : true);
      if (res) f[e] += (float)rh.e[e] + (float)rl.e[e];
      if (relu) f[e] = fmaxf(f[e], 0.f);
    }
    Half4View hi, lo;
    split4_f16(f, hi, lo);
    half* po = out + row * 2 * C + hoff;
    *reinterpret_cast<u32x2*>(po) = hi.p;
    *reinterpret_cast<u32x2*>(po + 32) = lo.p;
    if (rec) reinterpret_cast<u32x2*>(rec)[i] = rc.p;
  }
}
}  // namespace mn
using namespace mn;

#define CHECK(x)                                     \
  do {                                               \
    hipError_t e_ = (x);                             \
    if (e_ != hipSuccess) {                          \
      printf("%s: %s\n", #x, hipGetErrorString(e_)); \
      exit(1);                                       \
    }                                                \
  } while (0)

__global__ void fill_f32(float* p, long n, unsigned seed) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    p[i] = ((int)(h & 0xffff) - 32768) * (1.f / 16384.f);
  }
}
__global__ void fill_f16(half* p, long n, unsigned seed) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    p[i] = (half)(((int)(h & 0xffff) - 32768) * (1.f / 16384.f));
  }
}

// every SIMD busy with matrix instructions for ~iters x 4 x 32 cycles: the power state the convolutions leave the chip in
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef float floatx16v __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) heater(float* out, int iters) {
  half8v a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (_Float16)(threadIdx.x * 0.001f + e);
    b[e] = (_Float16)(e * 0.5f - threadIdx.x * 0.002f);
  }
  floatx16v acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  if (s == 1.2345f) out[0] = s;
}

static hipEvent_t e0, e1;
static float* g_sink;
static int g_heat_iters = 0;
// per-launch events around the timed kernel only; with g_heat_iters > 0 a heater launch precedes every timed launch
template <typename F>
static double time_each_us(F&& launch, int reps) {
  std::vector<hipEvent_t> ev(2 * reps);
  for (auto& e : ev) CHECK(hipEventCreate(&e));
  launch(0);
  CHECK(hipDeviceSynchronize());
  for (int r = 0; r < reps; ++r) {
    if (g_heat_iters) hipLaunchKernelGGL(heater, dim3(1024), dim3(256), 0, 0, g_sink, g_heat_iters);
    CHECK(hipEventRecord(ev[2 * r], 0));
    launch(r);
    CHECK(hipEventRecord(ev[2 * r + 1], 0));
  }
  CHECK(hipDeviceSynchronize());
  double tot = 0;
  for (int r = 0; r < reps; ++r) {
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, ev[2 * r], ev[2 * r + 1]));
    tot += ms;
  }
  for (auto& e : ev) CHECK(hipEventDestroy(e));
  return tot * 1e3 / reps;
}
template <typename F>
static double time_us(F&& launch, int reps) {
  launch(0);
  launch(1);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) launch(r);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  struct Shape { const char* name; long M; int C; };
  const Shape shapes[] = {{"layer1", 192L * 64 * 86, 64}, {"layer2", 192L * 32 * 43, 128}, {"layer3", 192L * 16 * 22, 256},
                          {"layer4", 192L * 8 * 11, 512}};  // BASELINE configs[2]: 192 images of 256 x 341
  constexpr int NBUF = 6;
  const long emax = shapes[0].M * shapes[0].C;
  float* Y[NBUF];
  half *R[NBUF], *O[NBUF], *K[NBUF], *G[NBUF], *GY[NBUF];
  for (int i = 0; i < NBUF; ++i) {
    CHECK(hipMalloc(&Y[i], emax * 4));
    CHECK(hipMalloc(&R[i], emax * 4));
    CHECK(hipMalloc(&O[i], emax * 4));
    CHECK(hipMalloc(&K[i], emax * 2));
    CHECK(hipMalloc(&G[i], emax * 2));
    CHECK(hipMalloc(&GY[i], emax * 2));
    hipLaunchKernelGGL(fill_f32, dim3(4096), dim3(256), 0, 0, Y[i], emax, 17u + i);
    hipLaunchKernelGGL(fill_f16, dim3(4096), dim3(256), 0, 0, R[i], emax * 2, 31u + i);
    hipLaunchKernelGGL(fill_f16, dim3(4096), dim3(256), 0, 0, G[i], emax, 43u + i);
  }
  float *coef, *mean, *invstd, *gamma, *dgamma, *dbeta;
  double* accum;
  CHECK(hipMalloc(&coef, 4 * 512 * 4));
  CHECK(hipMalloc(&mean, 512 * 4));
  CHECK(hipMalloc(&invstd, 512 * 4));
  CHECK(hipMalloc(&gamma, 512 * 4));
  CHECK(hipMalloc(&dgamma, 512 * 4));
  CHECK(hipMalloc(&dbeta, 512 * 4));
  CHECK(hipMalloc(&accum, 8 * 2 * 512 * 8));
  hipLaunchKernelGGL(fill_f32, dim3(8), dim3(256), 0, 0, coef, 2048, 5u);
  hipLaunchKernelGGL(fill_f32, dim3(2), dim3(256), 0, 0, mean, 512, 6u);
  hipLaunchKernelGGL(fill_f32, dim3(2), dim3(256), 0, 0, invstd, 512, 7u);
  hipLaunchKernelGGL(fill_f32, dim3(2), dim3(256), 0, 0, gamma, 512, 8u);
  CHECK(hipMemset(accum, 0, 8 * 2 * 512 * 8));
  CHECK(hipDeviceSynchronize());
  CHECK(hipMalloc(&g_sink, 64));
  if (getenv("BN_PROBE_HEAT")) {
    // heater experiment: the layer1 / layer2 passes alone (per-launch events), then each preceded by ~200 us of MFMA work
    const int iters = atoi(getenv("BN_PROBE_HEAT"));
    const int reps = 48;
    {
      g_heat_iters = 0;
      double us = time_each_us([&](int) { hipLaunchKernelGGL(heater, dim3(1024), dim3(256), 0, 0, g_sink, iters); }, 12);
      printf("heater(%d iterations) alone: %.1f us\n", iters, us);
    }
    for (int s = 0; s < 3; ++s) {
      const Shape& sh = shapes[s];
      const long elems = sh.M * sh.C, ni = elems / 8;
      const int grid = ew_grid(ni);
      const int rlanes = 256 / (sh.C / 8);
      long rows = (sh.M + 511) / 512;
      rows = ((rows + rlanes - 1) / rlanes) * rlanes;
      const int nblk = cdiv(sh.M, (int)rows);
      for (int heat = 0; heat < 2; ++heat) {
        g_heat_iters = heat ? iters : 0;
        double a = time_each_us([&](int r) {
          const int b = r % NBUF;
          hipLaunchKernelGGL(bn_apply_h2_kernel, dim3(grid), dim3(256), 0, 0, (const float*)Y[b], (const float*)coef, (const half*)R[b],
                             O[b], ni, sh.C, 1, 0, K[b], (const float*)mean, (const float*)invstd);
        }, reps);
        double b_ = time_each_us([&](int r) {
          const int b = r % NBUF;
          hipLaunchKernelGGL((bn_bwd_reduce_rec_kernel<4>), dim3(nblk), dim3(256), 0, 0, (const half*)G[b], (const half*)K[b], sh.M,
                             sh.C, accum, (int)rows, 1, 8);
        }, reps);
        double c = time_each_us([&](int r) {
          const int b = r % NBUF;
          hipLaunchKernelGGL(bn_bwd_apply_rec_kernel, dim3(grid), dim3(256), 0, 0, (const half*)G[b], (const half*)K[b],
                             (const float*)coef, GY[b], ni, sh.C, 1);
        }, reps);
        printf("%s %s: bn_apply_h2(res, rec) %.1f us (%.2f TB/s)  bwd reduce %.1f us (%.2f)  bwd apply %.1f us (%.2f)\n", sh.name,
               heat ? "after a heater launch" : "alone (per-launch events)", a, elems * 14.0 / a * 1e-6, b_, elems * 4.0 / b_ * 1e-6, c,
               elems * 6.0 / c * 1e-6);
      }
    }
    return 0;
  }
  const int reps = 24;
  const int grid_env = getenv("BN_PROBE_GRID") ? atoi(getenv("BN_PROBE_GRID")) : 0;
  for (const Shape& sh : shapes) {
    const long elems = sh.M * sh.C, ni = elems / 8;
    const int grid = grid_env ? grid_env : ew_grid(ni);
    printf("%s: M %ld C %d (%.1f M elements), grid %d\n", sh.name, sh.M, sh.C, elems * 1e-6, grid);
    // forward apply: the first unit of a block (own ReLU, record with gate), the second (residual + ReLU, record without gate)
    for (int res = 0; res < 2; ++res)
      for (int rec = 0; rec < 2; ++rec) {
        double us = time_us([&](int r) {
          const int b = r % NBUF;
          hipLaunchKernelGGL(bn_apply_h2_kernel, dim3(grid), dim3(256), 0, 0, (const float*)Y[b], (const float*)coef,
                             res ? (const half*)R[b] : (const half*)nullptr, O[b], ni, sh.C, 1, 0, rec ? K[b] : (half*)nullptr,
                             (const float*)mean, (const float*)invstd);
        }, reps);
        const double bytes = elems * (4.0 + (res ? 4.0 : 0.0) + 4.0 + (rec ? 2.0 : 0.0));
        printf("  bn_apply_h2 res=%d rec=%d            %7.1f us  %5.2f TB/s\n", res, rec, us, bytes / us * 1e-6);
      }
    // the 4-channel forward apply against the 8-channel one, over the workgroup cap
    for (int res = 0; res < 2; ++res)
      for (long cap : {4096L, 8192L, 16384L, 32768L, 1L << 30}) {
        const long ni4 = elems / 4;
        const int g8 = (int)(((ni + 255) / 256) < cap ? ((ni + 255) / 256) : cap);
        const int g4 = (int)(((ni4 + 255) / 256) < cap ? ((ni4 + 255) / 256) : cap);
        double a = time_us([&](int r) {
          const int b = r % NBUF;
          hipLaunchKernelGGL(bn_apply_h2_kernel, dim3(g8), dim3(256), 0, 0, (const float*)Y[b], (const float*)coef,
                             res ? (const half*)R[b] : (const half*)nullptr, O[b], ni, sh.C, 1, 0, K[b], (const float*)mean,
                             (const float*)invstd);
        }, reps);
        double c = time_us([&](int r) {
          const int b = r % NBUF;
          hipLaunchKernelGGL(bn_apply_h2v4_kernel, dim3(g4), dim3(256), 0, 0, (const float*)Y[b], (const float*)coef,
                             res ? (const half*)R[b] : (const half*)nullptr, O[b], ni4, sh.C, 1, K[b], (const float*)mean,
                             (const float*)invstd);
        }, reps);
        const double bytes = elems * (res ? 14.0 : 10.0);
        printf("  apply res=%d rec=1: 8 channels/thread wgs=%6d %7.1f us %5.2f TB/s | 4 channels/thread wgs=%6d %7.1f us %5.2f TB/s\n", res,
               g8, a, bytes / a * 1e-6, g4, c, bytes / c * 1e-6);
      }
    for (long cap : {4096L, 8192L, 16384L, 1L << 30}) {
      const int g8 = (int)(((ni + 255) / 256) < cap ? ((ni + 255) / 256) : cap);
      double us = time_us([&](int r) {
        const int b = r % NBUF;
        hipLaunchKernelGGL(bn_bwd_apply_rec_kernel, dim3(g8), dim3(256), 0, 0, (const half*)G[b], (const half*)K[b],
                           (const float*)coef, GY[b], ni, sh.C, 1);
      }, reps);
      printf("  bn_bwd_apply_rec wgs=%6d            %7.1f us  %5.2f TB/s\n", g8, us, elems * 6.0 / us * 1e-6);
    }
    // backward: reduce alone, apply alone, the launcher's three launches
    const int rlanes = 256 / (sh.C / 8);
    for (long target : {512L, 1024L, 2048L}) {
      long rows = (sh.M + target - 1) / target;
      rows = ((rows + rlanes - 1) / rlanes) * rlanes;
      if (rows < 4L * rlanes) rows = 4L * rlanes;
      const int nblk = cdiv(sh.M, (int)rows);
      for (int gate = 0; gate < 2; ++gate) {
        double us = time_us([&](int r) {
          const int b = r % NBUF;
          hipLaunchKernelGGL((bn_bwd_reduce_rec_kernel<4>), dim3(nblk), dim3(256), 0, 0, (const half*)G[b], (const half*)K[b], sh.M,
                             sh.C, accum, (int)rows, gate, 8);
        }, reps);
        printf("  bn_bwd_reduce_rec<4> wgs=%4d gate=%d  %7.1f us  %5.2f TB/s\n", nblk, gate, us, elems * 4.0 / us * 1e-6);
      }
    }
    for (int gate = 0; gate < 2; ++gate) {
      double us = time_us([&](int r) {
        const int b = r % NBUF;
        hipLaunchKernelGGL(bn_bwd_apply_rec_kernel, dim3(grid), dim3(256), 0, 0, (const half*)G[b], (const half*)K[b],
                           (const float*)coef, GY[b], ni, sh.C, gate);
      }, reps);
      printf("  bn_bwd_apply_rec gate=%d               %7.1f us  %5.2f TB/s\n", gate, us, elems * 6.0 / us * 1e-6);
    }
    {
      double us = time_us([&](int r) {
        const int b = r % NBUF;
        launch_bn_bwd_rec(G[b], K[b], sh.M, sh.C, gamma, mean, invstd, dgamma, dbeta, GY[b], accum, coef, 1.f, 0, true, 8);
      }, reps);
      printf("  launch_bn_bwd_rec (3 launches)        %7.1f us  %5.2f TB/s\n", us, elems * 10.0 / us * 1e-6);
    }
  }
  return 0;
}
