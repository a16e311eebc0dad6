// Hardware probe: does v_mfma_f32_32x32x16_f16 keep subnormal fp16 INPUTS?  (The x3 modes of igemm.h rely on the lo halves
// of small operands, which are subnormal fp16 numbers, reaching the product.)  A[i][k] = 2^-20 (subnormal: fp16's smallest
// normal is 2^-14), B[k][j] = 2^10: every product is 2^-10 and an output element is 16 * 2^-10 = 2^-6 if subnormal inputs
// are honoured, 0 if they are flushed.  Also checks v_cvt / v_fma_mix style conversions producing subnormals.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float tiny, float big) {
  half8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)tiny; b[e] = (_Float16)big; }
  floatx16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) {
    out[0] = c[0];
    out[1] = (float)a[0];               // the conversion itself: 2^-20 survives as an fp16 subnormal?
    const float x = 1.0f + 3.0f * tiny;  // hi + lo split of a value whose lo part is subnormal in fp16
    const _Float16 h = (_Float16)x;
    out[2] = (float)(_Float16)(x - (float)h);
  }
}
int main() {
  float* d; hipMalloc(&d, 64);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, ldexpf(1.f, -20), 1024.f);
  float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
  printf("mfma(2^-20 subnormal fp16 A, 2^10 B): out = %g (2^-6 = %g if subnormal inputs are kept, 0 if flushed)\n", h[0], ldexpf(1.f, -6));
  printf("(half)2^-20 back to float: %g (expected %g)\n", h[1], ldexpf(1.f, -20));
  printf("lo half of 1 + 3 * 2^-20: %g (expected %g)\n", h[2], 3 * ldexpf(1.f, -20));
  return 0;
}
