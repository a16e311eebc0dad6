// Hardware probe (round 5, for the next step of the parity mode's forward pass: DESIGN.md section 5, "open after round 5"):
// operand layout and scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) operands.
//   D[i][j] = sum_k A[i][k] B[k][j] * 2^(scale_a - 127) * 2^(scale_b - 127),  K = 64, one E8M0 scale byte per lane and operand.
// A lane holds 32 bytes of A (8 VGPRs) and 32 bytes of B.  Hypotheses for the k index of byte e of lane l (row / column l & 31):
//   H1  k = 32 (l >> 5) + e                      (a lane holds one contiguous half of K)
//   H2  k = 16 (l >> 5) + (e & 15) + 32 (e >> 4) (two 16-byte groups, as two consecutive 32x32x32 steps would read them)
// The probe fills A and B with asymmetric small integers under each hypothesis, runs the instruction and compares D with the host's
// product; then it checks that scale bytes multiply the result by powers of two, which byte of the scale VGPR `opsel` selects, and
// that the scale is taken PER LANE (a lane's 32 k values share its byte).  (A dot product does not care how k is numbered as long as
// A and B agree, so H1 and H2 both pass: what the probe pins is row / column = l & 31, the K halves by l >> 5, the C/D map.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ void k(const unsigned char* A, const unsigned char* B, float* D, int scale_a, int scale_b, int opsel) {
  const int l = threadIdx.x;
  if (opsel == 2) {  // per-lane scales: the A scale of lanes 0-31 (the first K half) is 2^1, everything else 2^0
    scale_a = l < 32 ? 128 : 127;
    scale_b = 127;
    opsel = 0;
  }
  intx8 a, b;
  for (int r = 0; r < 8; ++r) {
    a[r] = ((const int*)A)[l * 8 + r];
    b[r] = ((const int*)B)[l * 8 + r];
  }
  floatx16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  if (opsel == 0)
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
  else
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 1, scale_a, 1, scale_b);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;  // the 32x32 C/D map of the f16 forms
    D[row * 32 + col] = c[r];
  }
}

static unsigned char enc(int v) {  // small integers in fp8 e4m3 (OCP): bias 7, 3 mantissa bits
  static const unsigned char t[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4a, 0x4c, 0x4e, 0x50};  // 0..8
  const int m = v < 0 ? -v : v;
  return (unsigned char)(t[m] | (v < 0 ? 0x80 : 0));
}
static int kmap(int hyp, int l, int e) { return hyp == 1 ? 32 * (l >> 5) + e : 16 * (l >> 5) + (e & 15) + 32 * (e >> 4); }

int main() {
  static int Am[32][64], Bm[64][32];
  srand(3);
  for (int i = 0; i < 32; ++i)
    for (int kk = 0; kk < 64; ++kk) Am[i][kk] = rand() % 9 - 4;
  for (int kk = 0; kk < 64; ++kk)
    for (int j = 0; j < 32; ++j) Bm[kk][j] = rand() % 7 - 3;
  static float ref[32][32];
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      float s = 0;
      for (int kk = 0; kk < 64; ++kk) s += (float)(Am[i][kk] * Bm[kk][j]);
      ref[i][j] = s;
    }
  unsigned char *dA, *dB;
  float* dD;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 4096);
  static unsigned char hA[2048], hB[2048];
  static float hD[1024];
  for (int hyp = 1; hyp <= 2; ++hyp) {
    for (int l = 0; l < 64; ++l)
      for (int e = 0; e < 32; ++e) {
        hA[l * 32 + e] = enc(Am[l & 31][kmap(hyp, l, e)]);
        hB[l * 32 + e] = enc(Bm[kmap(hyp, l, e)][l & 31]);
      }
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, 127, 127, 0);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) bad += hD[i * 32 + j] != ref[i][j];
    printf("hypothesis H%d (scale bytes 127 = 2^0): %d of 1024 outputs differ from the host product%s\n", hyp, bad, bad ? "" : "  <-- the layout");
    if (!bad) {
      // scales: 2^3 on A, 2^-1 on B -> x4; then the same bytes in byte 1 of the VGPR with opsel = 1
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, 130, 126, 0);
      hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
      int b2 = 0;
      for (int i = 0; i < 1024; ++i) b2 += hD[i] != 4.f * ref[i / 32][i % 32];
      printf("  scale_a = 130, scale_b = 126 (opsel 0, byte 0): %d outputs differ from 4 x the product\n", b2);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, (130 << 8) | 127, (126 << 8) | 127, 1);
      hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
      b2 = 0;
      for (int i = 0; i < 1024; ++i) b2 += hD[i] != 4.f * ref[i / 32][i % 32];
      printf("  the same scales in byte 1 of the scale registers, opsel = 1: %d outputs differ from 4 x the product\n", b2);
      // one scale byte PER LANE: lanes 0-31 hold one half of K for every row, lanes 32-63 the other
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, 0, 0, 2);
      hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
      b2 = 0;
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          float e2 = 0;
          for (int l = 0; l < 64; l += 32)
            for (int e = 0; e < 32; ++e) {
              const int kk = kmap(hyp, l, e);
              e2 += (l < 32 ? 2.f : 1.f) * (float)(Am[i][kk] * Bm[kk][j]);
            }
          b2 += hD[i * 32 + j] != e2;
        }
      printf("  A scale 2^1 in lanes 0-31 only: %d outputs differ from 2 x (the bytes of lanes 0-31) + (the other lanes' bytes)%s\n", b2,
             b2 ? "  (expected to differ: a scale byte belongs to an MX block = bytes 16 b .. 16 b + 15 of BOTH lanes of a row, mfma_scale_probe2.hip)" : "");
    }
  }
  return 0;
}
