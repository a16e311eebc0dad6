// Stand-in for an RCCL ring all-reduce on a ONE-GPU box: a kernel that occupies `workgroups`
// workgroups of `threads` threads for `microseconds`, streaming `bytes` of reduce traffic (dst += src) at an even pace over that
// time -- what the reduction workgroups of a ring step do to the CUs and to HBM while the convolutions of the next backward stage
// want the whole chip.  It models residency and local traffic, not xGMI: the duration is an input (geomapnet_amd/dp.py derives it
// from the bucket size at an assumed bus bandwidth).  Used by dp.py under MN_DP_STANDIN and by tools/rccl_rehearsal.py only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace mn {

// LDS_KB: LDS the workgroup holds while resident (an RCCL workgroup keeps its ncclShmem staging there; a convolution workgroup of
// this library needs 150 of the CU's 160 KB, so a CU that hosts a collective workgroup with tens of KB cannot host one)
template <int LDS_KB>
__global__ void __launch_bounds__(1024) occupy_kernel(const floatx4* __restrict__ src, floatx4* __restrict__ dst, long n16,
                                                      long long ticks, int chunks) {
  __shared__ float pad[LDS_KB > 0 ? LDS_KB * 256 : 1];
  pad[threadIdx.x % (LDS_KB > 0 ? LDS_KB * 256 : 1)] = (float)threadIdx.x;  // (the allocation must be live)
  const long long t0 = wall_clock64();
  const long per_wg = (n16 + gridDim.x - 1) / gridDim.x;
  const long lo = (long)blockIdx.x * per_wg;
  const long hi = lo + per_wg < n16 ? lo + per_wg : n16;
  const long per_chunk = (per_wg + chunks - 1) / chunks;
  for (int c = 0; c < chunks; ++c) {
    const long a = lo + (long)c * per_chunk;
    const long b = a + per_chunk < hi ? a + per_chunk : hi;
    for (long i = a + threadIdx.x; i < b; i += blockDim.x) {
      dst[i] = dst[i] + src[i];
    }
    // pace: chunk c may not end before its share of the duration has passed (an RCCL workgroup waits for its peer's next slice)
    const long long until = t0 + ticks * (c + 1) / chunks;
    while (wall_clock64() < until) __builtin_amdgcn_s_sleep(32);
  }
  if (n16 < 0) dst[0][0] = pad[0];  // (never true: keeps `pad` observable)
}

// ticks of wall_clock64 per microsecond (constant-rate counter: hipDeviceAttributeWallClockRate is in kHz)
inline double wall_ticks_per_us() {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 100.0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) {
    (void)hipGetLastError();
    return 100.0;  // 100 MHz on every CDNA part so far
  }
  return khz / 1e3;
}

inline void launch_occupy(int workgroups, int threads, float microseconds, const void* src, void* dst, long bytes, int lds_kb,
                          hipStream_t s) {
  const long n16 = src && dst ? bytes / 16 : 0;
  const long long ticks = (long long)(microseconds * wall_ticks_per_us());
  if (lds_kb >= 64)
    hipLaunchKernelGGL(occupy_kernel<64>, dim3(workgroups), dim3(threads), 0, s, (const floatx4*)src, (floatx4*)dst, n16, ticks, 16);
  else if (lds_kb >= 32)
    hipLaunchKernelGGL(occupy_kernel<32>, dim3(workgroups), dim3(threads), 0, s, (const floatx4*)src, (floatx4*)dst, n16, ticks, 16);
  else
    hipLaunchKernelGGL(occupy_kernel<0>, dim3(workgroups), dim3(threads), 0, s, (const floatx4*)src, (floatx4*)dst, n16, ticks, 16);
}

}  // namespace mn
