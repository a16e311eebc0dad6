// Emulator self-test: barrier + LDS + shuffle reduction + all four MFMA builtins.
#include <hip/hip_runtime.h>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ void k_reduce(const float* x, float* out, int n) {
  __shared__ float part[4];
  float s = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) s += x[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// D[32][32] = A[32][16] * B[16][32] (A row-major, Bt[n][k] row-major)
__global__ void k_mfma32h(const _Float16* A, const _Float16* Bt, float* D) {
  int l = threadIdx.x;
  half8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = A[(l & 31) * 16 + 8 * (l >> 5) + e];
    b[e] = Bt[(l & 31) * 16 + 8 * (l >> 5) + e];
  }
  floatx16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
__global__ void k_mfma16h(const _Float16* A, const _Float16* Bt, float* D) {  // 16x16x32
  int l = threadIdx.x;
  half8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = A[(l & 15) * 32 + 8 * (l >> 4) + e];
    b[e] = Bt[(l & 15) * 32 + 8 * (l >> 4) + e];
  }
  floatx4 c = {0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
__global__ void k_mfma32f(const float* A, const float* Bt, float* D) {  // 32x32x2
  int l = threadIdx.x;
  floatx16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * 2 + (l >> 5)], Bt[(l & 31) * 2 + (l >> 5)], c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
__global__ void k_mfma16f(const float* A, const float* Bt, float* D) {  // 16x16x4
  int l = threadIdx.x;
  floatx4 c = {0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15) * 4 + (l >> 4)], Bt[(l & 15) * 4 + (l >> 4)], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

template <class T>
static double check(const std::vector<T>& A, const std::vector<T>& Bt, const std::vector<float>& D, int M, int N,
                    int K) {
  double md = 0;
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < N; ++j) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)Bt[j * K + k];
      md = fmax(md, fabs(s - D[i * N + j]));
    }
  return md;
}

int main() {
  int fails = 0;
  {
    int n = 100000;
    std::vector<float> x(n);
    double ref = 0;
    for (int i = 0; i < n; ++i) { x[i] = (i % 17) * 0.25f; ref += x[i]; }
    float out = 0;
    hipLaunchKernelGGL(k_reduce, dim3(13), dim3(256), 0, nullptr, (const float*)x.data(), &out, n);
    printf("reduce: %f vs %f\n", out, ref);
    if (fabs(out - ref) > 1e-2 * ref) fails++;
  }
  auto rnd = [](int i) { return (float)(((i * 2654435761u) >> 16) % 200) / 100.0f - 1.0f; };
  {
    std::vector<_Float16> A(32 * 16), Bt(32 * 16); std::vector<float> D(32 * 32);
    for (int i = 0; i < 512; ++i) { A[i] = (_Float16)rnd(i); Bt[i] = (_Float16)rnd(i + 777); }
    hipLaunchKernelGGL(k_mfma32h, dim3(1), dim3(64), 0, nullptr, (const _Float16*)A.data(), (const _Float16*)Bt.data(), D.data());
    double e = check(A, Bt, D, 32, 32, 16); printf("mfma 32x32x16 f16 err %g\n", e); if (e > 1e-4) fails++;
  }
  {
    std::vector<_Float16> A(16 * 32), Bt(16 * 32); std::vector<float> D(16 * 16);
    for (int i = 0; i < 512; ++i) { A[i] = (_Float16)rnd(i); Bt[i] = (_Float16)rnd(i + 777); }
    hipLaunchKernelGGL(k_mfma16h, dim3(1), dim3(64), 0, nullptr, (const _Float16*)A.data(), (const _Float16*)Bt.data(), D.data());
    double e = check(A, Bt, D, 16, 16, 32); printf("mfma 16x16x32 f16 err %g\n", e); if (e > 1e-4) fails++;
  }
  {
    std::vector<float> A(64), Bt(64), D(32 * 32);
    for (int i = 0; i < 64; ++i) { A[i] = rnd(i); Bt[i] = rnd(i + 99); }
    hipLaunchKernelGGL(k_mfma32f, dim3(1), dim3(64), 0, nullptr, (const float*)A.data(), (const float*)Bt.data(), D.data());
    double e = check(A, Bt, D, 32, 32, 2); printf("mfma 32x32x2 f32 err %g\n", e); if (e > 1e-6) fails++;
  }
  {
    std::vector<float> A(64), Bt(64), D(16 * 16);
    for (int i = 0; i < 64; ++i) { A[i] = rnd(i); Bt[i] = rnd(i + 99); }
    hipLaunchKernelGGL(k_mfma16f, dim3(1), dim3(64), 0, nullptr, (const float*)A.data(), (const float*)Bt.data(), D.data());
    double e = check(A, Bt, D, 16, 16, 4); printf("mfma 16x16x4 f32 err %g\n", e); if (e > 1e-6) fails++;
  }
  printf(fails ? "EMU SELFTEST FAILED\n" : "EMU SELFTEST OK\n");
  return fails;
}
