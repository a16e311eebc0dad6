// Hardware probe: which A elements (lane, byte) does the scale byte of ONE lane multiply in v_mfma_scale_f32_32x32x64_f8f6f4 (fp8)?
// (see mfma_scale_probe.hip)  Measured on MI355X (profiles/r05/c12_mfma_scale_probe.txt): the scale byte of lane i (i < 32) multiplies
// bytes 0-15 of lane i AND of lane i + 32; the scale byte of lane i + 32 multiplies bytes 16-31 of both.  So the MX block b of row i
// (32 consecutive k) is bytes 16 b .. 16 b + 15 of the two lanes that hold the row, and its scale lives in lane i + 32 b:
// k = 16 (l >> 5) + (e & 15) + 32 (e >> 4)  (hypothesis H2 of mfma_scale_probe.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void k(const unsigned char* A, float* D, int scaled_lane) {
  const int l = threadIdx.x;
  intx8 a, b;
  for (int r = 0; r < 8; ++r) {
    a[r] = ((const int*)A)[l * 8 + r];
    b[r] = 0x38383838;  // fp8 1.0 everywhere
  }
  floatx16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  const int sa = l == scaled_lane ? 128 : 127;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, 127);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
int main() {
  unsigned char* dA; float* dD;
  (void)hipMalloc(&dA, 2048); (void)hipMalloc(&dD, 4096);
  static unsigned char hA[2048]; static float hD[1024];
  for (int sl : {5, 37}) {
    for (int al : {5, 37}) {
      printf("scale 2^1 on lane %2d; A = 1.0 at (lane %2d, byte e), e = 0..31 -> D[row 5][0]: ", sl, al);
      for (int e = 0; e < 32; ++e) {
        memset(hA, 0, sizeof hA);
        hA[al * 32 + e] = 0x38;
        (void)hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dD, sl);
        (void)hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
        printf("%g", hD[5 * 32 + 0]);
      }
      printf("\n");
    }
  }
  return 0;
}
