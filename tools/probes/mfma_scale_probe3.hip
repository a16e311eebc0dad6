// Hardware probe: per-lane-half scale bytes on BOTH operands of v_mfma_scale_f32_32x32x64_f8f6f4 (see mfma_scale_probe2.hip).
// Integer data, scale of lane half 0 / 1: A 2^-1 / 2^1, B 2^2 / 2^-2.  Models for "which lane's byte scales MX block b":
//   M1  A: lane (row + 32 b), B: lane (col + 32 b)      M2  A as M1, B: lane (col + 32 (1 - b))
//   M3  A: lane (row + 32 (1 - b)), B as M1             M4  both swapped
// Measured on MI355X: M1 (0 of 1024 outputs differ; M2-M4: > 1000) -- also with the fp16x2q kernels' own scale bytes (118 / 128 on A,
// 121 / 111 on B: products of 2^-15 in both blocks), profiles/r05/c12_mfma_scale_probe.txt.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void k(const unsigned char* A, const unsigned char* B, float* D) {
  const int l = threadIdx.x;
  intx8 a, b;
  for (int r = 0; r < 8; ++r) {
    a[r] = ((const int*)A)[l * 8 + r];
    b[r] = ((const int*)B)[l * 8 + r];
  }
  floatx16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  const int sa = l < 32 ? 126 : 128, sb = l < 32 ? 129 : 125;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
static unsigned char enc(int v) {
  static const unsigned char t[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4a, 0x4c, 0x4e, 0x50};
  const int m = v < 0 ? -v : v;
  return (unsigned char)(t[m] | (v < 0 ? 0x80 : 0));
}
int main() {
  static int Am[64][32], Bm[64][32];  // [lane][byte]
  srand(5);
  for (int l = 0; l < 64; ++l)
    for (int e = 0; e < 32; ++e) { Am[l][e] = rand() % 9 - 4; Bm[l][e] = rand() % 7 - 3; }
  static unsigned char hA[2048], hB[2048];
  for (int l = 0; l < 64; ++l)
    for (int e = 0; e < 32; ++e) { hA[l * 32 + e] = enc(Am[l][e]); hB[l * 32 + e] = enc(Bm[l][e]); }
  unsigned char *dA, *dB; float* dD;
  (void)hipMalloc(&dA, 2048); (void)hipMalloc(&dB, 2048); (void)hipMalloc(&dD, 4096);
  (void)hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  static float hD[1024];
  (void)hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
  const float sA[2] = {0.5f, 2.f}, sB[2] = {4.f, 0.25f};  // scale of lane half 0 / 1
  for (int model = 1; model <= 4; ++model) {
    int bad = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        float s = 0;
        for (int blk = 0; blk < 2; ++blk) {
          float p = 0;
          for (int half = 0; half < 2; ++half)
            for (int e = 0; e < 16; ++e) p += (float)(Am[i + 32 * half][16 * blk + e] * Bm[j + 32 * half][16 * blk + e]);
          const int ha = (model == 3 || model == 4) ? 1 - blk : blk, hb = (model == 2 || model == 4) ? 1 - blk : blk;
          s += p * sA[ha] * sB[hb];
        }
        bad += hD[i * 32 + j] != s;
      }
    printf("model M%d: %d of 1024 outputs differ\n", model, bad);
  }
  return 0;
}
