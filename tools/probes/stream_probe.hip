// Hardware probe: what the HBM path sustains for the access patterns of the BatchNorm passes, as a function of the
// launch shape.  (profiles/r06/c15_to_c20_*: the library's BatchNorm passes run at 0.9-1.0 of these figures.)
// The ceiling of each pattern with nothing else in the kernel:
//   read   : sum of two fp16 streams (the backward reduce), blocked per workgroup or grid-strided, U pieces in flight per stream
//   apply  : 2 B + 2 B in, 2 B out (the backward apply)
//   fwd    : 4 B (+ 4 B residual, h2 layout) in, 4 B h2 + 2 B record out (the forward apply); y read as two adjacent pieces per
//            lane (ADJ) or as pieces L and L+64 of a 128-piece block (fully coalesced instructions)
// Buffers rotate over NBUF distinct allocations so that nothing is served from the 256 MB Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/stream_probe tools/probes/stream_probe.hip && tools/probes/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                              \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

template <bool NT>
__device__ inline uint4v ld(const uint4v* p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

__device__ inline float fold(uint4v a, uint4v b) {
  half8 x = __builtin_bit_cast(half8, a), y = __builtin_bit_cast(half8, b);
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += (float)x[e] * (float)y[e];
  return s;
}

// two streams, grid-strided
template <int U, bool NT>
__global__ void __launch_bounds__(256) read_strided(const uint4v* __restrict__ a, const uint4v* __restrict__ b, long n, float* out) {
  float acc = 0.f;
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride * U) {
    uint4v va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long j = i + u * stride < n ? i + u * stride : i;
      va[u] = ld<NT>(a + j);
      vb[u] = ld<NT>(b + j);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += fold(va[u], vb[u]);
  }
  if (acc == 1.2345f) out[0] = acc;
}

// two streams, one contiguous chunk per workgroup (the reduce kernel's shape)
template <int U, bool NT>
__global__ void __launch_bounds__(256) read_blocked(const uint4v* __restrict__ a, const uint4v* __restrict__ b, long n, float* out) {
  float acc = 0.f;
  const long per = (n + gridDim.x - 1) / gridDim.x;
  const long i0 = (long)blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
  for (long i = i0 + threadIdx.x; i < i1; i += 256 * U) {
    uint4v va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long j = i + u * 256 < i1 ? i + u * 256 : i;
      va[u] = ld<NT>(a + j);
      vb[u] = ld<NT>(b + j);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += fold(va[u], vb[u]);
  }
  if (acc == 1.2345f) out[0] = acc;
}

// backward apply: out = f(a, b), grid-strided, U pieces in flight
template <int U, bool NT>
__global__ void __launch_bounds__(256) apply_strided(const uint4v* __restrict__ a, const uint4v* __restrict__ b, uint4v* __restrict__ o,
                                                     long n) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride * U) {
    uint4v va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + u * stride < n) {
        va[u] = ld<NT>(a + i + u * stride);
        vb[u] = ld<NT>(b + i + u * stride);
      }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + u * stride < n) {
        half8 x = __builtin_bit_cast(half8, va[u]), y = __builtin_bit_cast(half8, vb[u]);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = x[e] * (_Float16)1.5f + y[e];
        o[i + u * stride] = __builtin_bit_cast(uint4v, x);
      }
  }
}

// forward apply.  item = (row, 32-channel group): y 128 B, res hi 64 B + lo 64 B, out hi 64 B + lo 64 B, rec 64 B.
// MODE 0: 4 lanes per item, lane reads y pieces 2i, 2i+1 (what bn_apply_h2_kernel does).
// MODE 1: 8 lanes per item, lane reads ONE y piece (4 channels): every instruction fully contiguous; 8-byte h2 / rec accesses.
template <int MODE, bool RES, bool REC, bool NT>
__global__ void __launch_bounds__(256) fwd_apply(const float* __restrict__ y, const _Float16* __restrict__ res, _Float16* __restrict__ out,
                                                 _Float16* __restrict__ rec, long nitems) {
  if (MODE == 0) {
    const long n = nitems * 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
      const long item = i >> 2;
      const int sub = (int)(i & 3);
      floatx4 v0 = __builtin_bit_cast(floatx4, ld<NT>(reinterpret_cast<const uint4v*>(y) + 2 * i));
      floatx4 v1 = __builtin_bit_cast(floatx4, ld<NT>(reinterpret_cast<const uint4v*>(y) + 2 * i + 1));
      float f[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      if (RES) {
        half8 h = __builtin_bit_cast(half8, *(reinterpret_cast<const uint4v*>(res) + item * 8 + sub));
        half8 l = __builtin_bit_cast(half8, *(reinterpret_cast<const uint4v*>(res) + item * 8 + 4 + sub));
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += (float)h[e] + (float)l[e];
      }
      half8 h, l, r;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        f[e] = fmaxf(f[e] * 1.25f + 0.5f, 0.f);
        h[e] = (_Float16)f[e];
        l[e] = (_Float16)(f[e] - (float)h[e]);
        r[e] = (_Float16)(f[e] * 0.75f);
      }
      *(reinterpret_cast<uint4v*>(out) + item * 8 + sub) = __builtin_bit_cast(uint4v, h);
      *(reinterpret_cast<uint4v*>(out) + item * 8 + 4 + sub) = __builtin_bit_cast(uint4v, l);
      if (REC) *(reinterpret_cast<uint4v*>(rec) + i) = __builtin_bit_cast(uint4v, r);
    }
  } else {
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    typedef unsigned int uint2v __attribute__((ext_vector_type(2)));
    const long n = nitems * 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
      const long item = i >> 3;
      const int sub = (int)(i & 7);
      floatx4 v0 = __builtin_bit_cast(floatx4, ld<NT>(reinterpret_cast<const uint4v*>(y) + i));
      float f[4] = {v0[0], v0[1], v0[2], v0[3]};
      if (RES) {
        half4 h = __builtin_bit_cast(half4, *(reinterpret_cast<const uint2v*>(res) + item * 16 + sub));
        half4 l = __builtin_bit_cast(half4, *(reinterpret_cast<const uint2v*>(res) + item * 16 + 8 + sub));
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] += (float)h[e] + (float)l[e];
      }
      half4 h, l, r;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f[e] = fmaxf(f[e] * 1.25f + 0.5f, 0.f);
        h[e] = (_Float16)f[e];
        l[e] = (_Float16)(f[e] - (float)h[e]);
        r[e] = (_Float16)(f[e] * 0.75f);
      }
      *(reinterpret_cast<uint2v*>(out) + item * 16 + sub) = __builtin_bit_cast(uint2v, h);
      *(reinterpret_cast<uint2v*>(out) + item * 16 + 8 + sub) = __builtin_bit_cast(uint2v, l);
      if (REC) *(reinterpret_cast<uint2v*>(rec) + i) = __builtin_bit_cast(uint2v, r);
    }
  }
}

static hipEvent_t e0, e1;
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
  const long elems = argc > 1 ? atol(argv[1]) : 192L * 64 * 86 * 64;  // layer1 of BASELINE configs[2] (256 x 341 images): 67.6 M elements
  constexpr int NBUF = 6;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  // per buffer set: a, b (2 B), o (2 B), y (4 B), res (4 B), out (4 B), rec (2 B)
  char *A[NBUF], *B[NBUF], *O[NBUF], *Y[NBUF], *R[NBUF], *H[NBUF], *K[NBUF];
  for (int i = 0; i < NBUF; ++i) {
    CHECK(hipMalloc(&A[i], elems * 2));
    CHECK(hipMalloc(&B[i], elems * 2));
    CHECK(hipMalloc(&O[i], elems * 2));
    CHECK(hipMalloc(&Y[i], elems * 4));
    CHECK(hipMalloc(&R[i], elems * 4));
    CHECK(hipMalloc(&H[i], elems * 4));
    CHECK(hipMalloc(&K[i], elems * 2));
    CHECK(hipMemset(A[i], 0x11, elems * 2));
    CHECK(hipMemset(B[i], 0x12, elems * 2));
    CHECK(hipMemset(Y[i], 0x13, elems * 4));
    CHECK(hipMemset(R[i], 0x14, elems * 4));
  }
  float* sink;
  CHECK(hipMalloc(&sink, 64));
  const long np = elems / 8;  // 16-byte fp16 pieces per stream
  const int reps = 24;
  printf("elements %ld (%.1f MB per fp16 stream), %d buffer sets, %d launches per figure\n", elems, elems * 2e-6, NBUF, reps);

#define RD(K_, U_, NT_, WGS)                                                                                                      \
  {                                                                                                                               \
    double us = time_us([&](int r) { hipLaunchKernelGGL((K_<U_, NT_>), dim3(WGS), dim3(256), 0, 0, (const uint4v*)A[r % NBUF],   \
                                                        (const uint4v*)B[r % NBUF], np, sink); }, reps);                        \
    printf("%-13s U=%d nt=%d wgs=%5d  %7.1f us  %5.2f TB/s\n", #K_, U_, NT_, WGS, us, elems * 4.0 / us * 1e-6);                    \
  }
  for (int wgs : {512, 1024, 2048, 4096, 8192}) {
    RD(read_blocked, 1, false, wgs) RD(read_blocked, 2, false, wgs) RD(read_blocked, 4, false, wgs) RD(read_blocked, 8, false, wgs)
    RD(read_blocked, 4, true, wgs)
    RD(read_strided, 1, false, wgs) RD(read_strided, 2, false, wgs) RD(read_strided, 4, false, wgs) RD(read_strided, 8, false, wgs)
    RD(read_strided, 4, true, wgs)
  }
  {
    const int full = (int)((np + 255) / 256);
    RD(read_strided, 1, false, full) RD(read_strided, 1, true, full)
  }
#define AP(U_, NT_, WGS)                                                                                                          \
  {                                                                                                                               \
    double us = time_us([&](int r) { hipLaunchKernelGGL((apply_strided<U_, NT_>), dim3(WGS), dim3(256), 0, 0,                    \
                                                        (const uint4v*)A[r % NBUF], (const uint4v*)B[r % NBUF],                  \
                                                        (uint4v*)O[r % NBUF], np); }, reps);                                     \
    printf("apply_strided U=%d nt=%d wgs=%5d  %7.1f us  %5.2f TB/s\n", U_, NT_, WGS, us, elems * 6.0 / us * 1e-6);                 \
  }
  for (int wgs : {1024, 2048, 4096, 8192, 16384}) { AP(1, false, wgs) AP(2, false, wgs) AP(4, false, wgs) AP(1, true, wgs) AP(2, true, wgs) }
  {
    const int full = (int)((np + 255) / 256);
    AP(1, false, full) AP(1, true, full)
  }
  const long nitems = elems / 32;
#define FW(MODE_, RES_, REC_, NT_, WGS)                                                                                           \
  {                                                                                                                               \
    double us = time_us([&](int r) { hipLaunchKernelGGL((fwd_apply<MODE_, RES_, REC_, NT_>), dim3(WGS), dim3(256), 0, 0,         \
                                                        (const float*)Y[r % NBUF], (const _Float16*)R[r % NBUF],                 \
                                                        (_Float16*)H[r % NBUF], (_Float16*)K[r % NBUF], nitems); }, reps);       \
    const double bytes = elems * (4.0 + (RES_ ? 4.0 : 0.0) + 4.0 + (REC_ ? 2.0 : 0.0));                                          \
    printf("fwd_apply mode=%d res=%d rec=%d nt=%d wgs=%5d  %7.1f us  %5.2f TB/s\n", MODE_, RES_, REC_, NT_, WGS, us,              \
           bytes / us * 1e-6);                                                                                                    \
  }
  for (int wgs : {2048, 4096, 8192, 16384}) {
    FW(0, true, true, true, wgs) FW(0, true, true, false, wgs) FW(1, true, true, true, wgs) FW(1, true, true, false, wgs)
    FW(0, false, true, true, wgs) FW(1, false, true, true, wgs) FW(0, false, false, true, wgs) FW(1, false, false, true, wgs)
  }
  {
    const int full0 = (int)((nitems * 4 + 255) / 256), full1 = (int)((nitems * 8 + 255) / 256);
    FW(0, true, true, true, full0) FW(1, true, true, true, full1) FW(0, true, true, false, full0) FW(1, true, true, false, full1)
  }
  return 0;
}
