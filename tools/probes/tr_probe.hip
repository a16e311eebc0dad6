// Hardware probe: semantics of ds_read_b64_tr_b16 on gfx950 (which LDS halves land in which lane/element).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short u16;
__global__ void probe(int mode, u16* out) {
  __shared__ u16 lds[8192];
  int l = threadIdx.x;
  for (int i = l; i < 8192; i += 64) lds[i] = (u16)i;
  __syncthreads();
  unsigned addr;
  if (mode == 0) addr = l * 8;                              // consecutive 8-byte chunks
  else if (mode == 1) addr = (l & 15) * 256 + (l >> 4) * 8; // [16 rows][128 halves]: lane = row, group = column chunk
  else addr = (l & 15) * 64 + (l >> 4) * 8;                 // rows of 32 halves
  addr += (unsigned)(size_t)(&lds[0]) & 0xffff;
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (u16)(v >> (16 * e));
}
int main() {
  u16* d; hipMalloc(&d, 64 * 4 * 2);
  u16 h[256];
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, mode, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
