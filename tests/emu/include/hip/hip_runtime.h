// TEST INFRASTRUCTURE ONLY -- SIMT emulator shim.
//
// This header shadows <hip/hip_runtime.h> when the kernel sources under
// geomapnet_amd/csrc are compiled for the x86 host by tests/emu/Makefile.  It lets the
// CPU-only test-suite execute the *same kernel source* (LDS tiling, swizzles, MFMA fragment
// maps, wave reductions, barriers) that hipcc compiles for gfx950, so kernel logic is
// debugged before any GPU minute is spent.  Every GPU thread is a fiber; __syncthreads and
// the wave-collective intrinsics are cooperative barriers; MFMA builtins are evaluated
// from the documented gfx950 lane->element maps.
//
// The product (geomapnet_amd) never loads the emulator build: geomapnet_amd/_binding.py
// only ever dlopens libmapnet_hip.so and raises if it is missing.
#pragma once
#include <time.h>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <utility>

#define MAPNET_EMU_BUILD 1

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

struct uint3 {
  unsigned x, y, z;
};
struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef void* hipStream_t;
typedef struct emu_event* hipEvent_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorUnknown = 999 };
enum hipMemcpyKind {
  hipMemcpyHostToHost = 0,
  hipMemcpyHostToDevice = 1,
  hipMemcpyDeviceToHost = 2,
  hipMemcpyDeviceToDevice = 3,
  hipMemcpyDefault = 4
};

namespace emu {
struct Ctx {
  uint3 tid, bid;
  dim3 bdim, gdim;
  int lane, wave, linear;
};
extern thread_local Ctx* cur;
void block_sync();
void wave_sync();
// per-wave exchange buffer: 64 slots of 256 bytes
unsigned char* wave_slot(int lane);
int wave_live_lanes();
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
}  // namespace emu

#define threadIdx (emu::cur->tid)
#define blockIdx (emu::cur->bid)
#define blockDim (emu::cur->bdim)
#define gridDim (emu::cur->gdim)
#define warpSize 64

inline void __syncthreads() { emu::block_sync(); }
inline void __threadfence() {}

// ---- host API subset ------------------------------------------------------------------
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t = nullptr) {
  memset(p, v, n);
  return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) {
  memcpy(d, s, n);
  return hipSuccess;
}
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
  memcpy(d, s, n);
  return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) {
  *p = calloc(1, n);
  return *p ? hipSuccess : hipErrorUnknown;
}
inline hipError_t hipHostFree(void* p) {
  free(p);
  return hipSuccess;
}
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
// launches are synchronous here, so extra streams and cross-stream waits are trivially satisfied
enum { hipEventDisableTiming = 2, hipStreamNonBlocking = 1 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
  *s = nullptr;
  return hipErrorUnknown;
}
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, unsigned, const unsigned*) {
  *s = nullptr;
  return hipErrorUnknown;
}
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) {
  *d = 0;
  return hipSuccess;
}
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }

template <typename... KArgs, typename... Args>
inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t /*shmem*/,
                               hipStream_t /*stream*/, Args... args) {
  emu::launch(grid, block, [=]() { kernel(args...); });
}

// ---- atomics (blocks run on several OS threads) ----------------------------------------
template <typename T>
inline T emu_atomic_add(T* p, T v) {
  T old, neu;
  do {
    __atomic_load(p, &old, __ATOMIC_RELAXED);
    neu = old + v;
  } while (!__atomic_compare_exchange(p, &old, &neu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  return old;
}
inline float atomicAdd(float* p, float v) { return emu_atomic_add(p, v); }
inline double atomicAdd(double* p, double v) { return emu_atomic_add(p, v); }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline float unsafeAtomicAdd(float* p, float v) { return emu_atomic_add(p, v); }
inline float atomicExch(float* p, float v) {
  unsigned old, nv;
  memcpy(&nv, &v, 4);
  old = __atomic_exchange_n(reinterpret_cast<unsigned*>(p), nv, __ATOMIC_SEQ_CST);
  float r;
  memcpy(&r, &old, 4);
  return r;
}
inline int atomicExch(int* p, int v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
enum { hipDeviceAttributeMultiprocessorCount = 63, hipDeviceAttributeWallClockRate = 10017 };
inline hipError_t hipDeviceGetAttribute(int* v, int attr, int) {
  *v = attr == hipDeviceAttributeWallClockRate ? 100000  // kHz: wall_clock64() below ticks at 100 MHz
                                               : 4;     // a small "chip": stream-K launches use 8 workgroups under the emulator
  return hipSuccess;
}
// constant-rate device clock (100 MHz on the hardware): the host's steady clock in 10 ns ticks
inline long long wall_clock64() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (long long)ts.tv_sec * 100000000ll + ts.tv_nsec / 10;
}
inline void __builtin_amdgcn_s_sleep(int) {}

inline double unsafeAtomicAdd(double* p, double v) { return emu_atomic_add(p, v); }

// ---- wave collectives ---------------------------------------------------------------------
template <typename T>
inline T emu_wave_exchange(T v, int src_lane) {
  static_assert(sizeof(T) <= 256, "slot too small");
  memcpy(emu::wave_slot(emu::cur->lane), &v, sizeof(T));
  emu::wave_sync();
  T r;
  memcpy(&r, emu::wave_slot(src_lane & 63), sizeof(T));
  emu::wave_sync();
  return r;
}
template <typename T>
inline T __shfl(T v, int src, int width = 64) {
  int l = emu::cur->lane;
  int base = l & ~(width - 1);
  return emu_wave_exchange(v, base + (src & (width - 1)));
}
template <typename T>
inline T __shfl_xor(T v, int mask, int width = 64) {
  int l = emu::cur->lane;
  int t = l ^ mask;
  if ((t & ~(width - 1)) != (l & ~(width - 1))) t = l;
  return emu_wave_exchange(v, t);
}
template <typename T>
inline T __shfl_down(T v, unsigned d, int width = 64) {
  int l = emu::cur->lane;
  int t = l + (int)d;
  if ((t & ~(width - 1)) != (l & ~(width - 1))) t = l;
  return emu_wave_exchange(v, t);
}
inline unsigned long long __ballot(int pred) {
  int p = pred ? 1 : 0;
  memcpy(emu::wave_slot(emu::cur->lane), &p, sizeof(int));
  emu::wave_sync();
  unsigned long long m = 0;
  int n = emu::wave_live_lanes();
  for (int i = 0; i < n; ++i) {
    int q;
    memcpy(&q, emu::wave_slot(i), sizeof(int));
    if (q) m |= 1ull << i;
  }
  emu::wave_sync();
  return m;
}

// ---- math helpers that exist in HIP device code -------------------------------------------
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __expf(float x) { return expf(x); }
inline float __fdividef(float a, float b) { return a / b; }

// ---- MFMA builtins, from the gfx950 lane->element maps (cdna_hip_programming.md section 3) ----
typedef _Float16 emu_half8 __attribute__((ext_vector_type(8)));
typedef float emu_floatx4 __attribute__((ext_vector_type(4)));
typedef float emu_floatx16 __attribute__((ext_vector_type(16)));

struct emu_mfma_slot_h {
  emu_half8 a, b;
};
struct emu_mfma_slot_f {
  float a, b;
};

// v_mfma_f32_32x32x16_{f16,bf16}: A[i=l&31][k=8*(l>>5)+e], B[k=8*(l>>5)+e][j=l&31],
// D: col=l&31, row=(reg&3)+8*(reg>>2)+4*(l>>5).  Every lane deposits its operands ALREADY CONVERTED to float (the
// conversion is exact) plus a flag "my A fragment is all zero"; an output row whose 16 A values are zero is skipped when the
// lane's B column is finite (adding sixteen 0 * b terms; the test problems fill a small corner of the 288- / 128-row
// tiles, so most rows are -- the full-tile loop made the fp16 network tests spend 80 % of their time here).  The products
// are accumulated in the same order and with the same expression as before: results are unchanged (except that an
// accumulator holding -0.0f keeps its sign where the additions of +0.0f would have flipped it).
struct emu_mfma_slot_x {
  float a[8], b[8];
  int azero;
};
template <typename V8>
inline emu_floatx16 emu_mfma_32x32x16(V8 a, V8 b, emu_floatx16 c) {
  int l = emu::cur->lane;
  emu_mfma_slot_x s;
  s.azero = 1;
  for (int e = 0; e < 8; ++e) {
    s.a[e] = (float)a[e];
    s.b[e] = (float)b[e];
    if (s.a[e] != 0.0f) s.azero = 0;  // (NaN != 0: a NaN fragment is never skipped)
  }
  memcpy(emu::wave_slot(l), &s, sizeof(s));
  emu::wave_sync();
  int j = l & 31;
  const emu_mfma_slot_x* sb0 = reinterpret_cast<const emu_mfma_slot_x*>(emu::wave_slot(j));
  const emu_mfma_slot_x* sb1 = reinterpret_cast<const emu_mfma_slot_x*>(emu::wave_slot(j + 32));
  float bc[16];
  bool bfinite = true;
  for (int k = 0; k < 8; ++k) {
    bc[k] = sb0->b[k];
    bc[8 + k] = sb1->b[k];
    bfinite = bfinite && std::isfinite(bc[k]) && std::isfinite(bc[8 + k]);
  }
  for (int reg = 0; reg < 16; ++reg) {
    int i = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
    const emu_mfma_slot_x* sa0 = reinterpret_cast<const emu_mfma_slot_x*>(emu::wave_slot(i));
    const emu_mfma_slot_x* sa1 = reinterpret_cast<const emu_mfma_slot_x*>(emu::wave_slot(i + 32));
    if (bfinite && sa0->azero && sa1->azero) continue;
    float acc = c[reg];
    for (int k = 0; k < 8; ++k) acc += sa0->a[k] * bc[k];
    for (int k = 0; k < 8; ++k) acc += sa1->a[k] * bc[8 + k];
    c[reg] = acc;
  }
  emu::wave_sync();
  return c;
}
inline emu_floatx16 __builtin_amdgcn_mfma_f32_32x32x16_f16(emu_half8 a, emu_half8 b, emu_floatx16 c, int, int,
                                                           int) {
  return emu_mfma_32x32x16(a, b, c);
}
typedef __bf16 emu_bf16x8 __attribute__((ext_vector_type(8)));
inline emu_floatx16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_floatx16 c, int, int,
                                                            int) {
  return emu_mfma_32x32x16(a, b, c);
}

// v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) operands (cbsz = blgp = 0), as probed on MI355X (tools/probes/mfma_scale_probe*.hip,
// profiles/r05/c12_mfma_scale_probe.txt): lane l holds row / column l & 31; its 32 bytes are bytes 0-15 = 16 k values of MX block 0,
// bytes 16-31 = 16 k values of MX block 1 (the two lanes of a row complete a block's 32 values); block b of a row is scaled by
// 2^(byte - 127) of the scale register of lane (row + 32 b), `opsel` choosing the byte; C/D in the f16 forms' map.
struct emu_mfma_slot_q {
  unsigned char a[32], b[32];
  unsigned char sa, sb;
};
inline float emu_fp8_e4m3(unsigned char v) {
  const int ef = (v >> 3) & 15, m = v & 7;
  float f = ef == 0 ? ldexpf((float)m, -9) : ldexpf((float)(8 + m), ef - 10);
  if (ef == 15 && m == 7) f = NAN;
  return (v & 0x80) ? -f : f;
}
template <typename V8>
inline emu_floatx16 __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(V8 a, V8 b, emu_floatx16 c, int, int, int opsel_a, int scale_a,
                                                                    int opsel_b, int scale_b) {
  static_assert(sizeof(V8) == 32, "8 dwords per operand");
  int l = emu::cur->lane;
  emu_mfma_slot_q s;
  memcpy(s.a, &a, 32);
  memcpy(s.b, &b, 32);
  s.sa = (unsigned char)((unsigned)scale_a >> (8 * opsel_a));
  s.sb = (unsigned char)((unsigned)scale_b >> (8 * opsel_b));
  memcpy(emu::wave_slot(l), &s, sizeof(s));
  emu::wave_sync();
  const int j = l & 31;
  const emu_mfma_slot_q* b0 = reinterpret_cast<const emu_mfma_slot_q*>(emu::wave_slot(j));
  const emu_mfma_slot_q* b1 = reinterpret_cast<const emu_mfma_slot_q*>(emu::wave_slot(j + 32));
  float bv[2][32];  // [block][k in block]: k < 16 from the lane of half 0, k >= 16 from the lane of half 1
  for (int blk = 0; blk < 2; ++blk)
    for (int e = 0; e < 16; ++e) {
      bv[blk][e] = emu_fp8_e4m3(b0->b[16 * blk + e]);
      bv[blk][16 + e] = emu_fp8_e4m3(b1->b[16 * blk + e]);
    }
  const float sbf[2] = {ldexpf(1.f, (int)b0->sb - 127), ldexpf(1.f, (int)b1->sb - 127)};
  for (int reg = 0; reg < 16; ++reg) {
    const int i = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
    const emu_mfma_slot_q* a0 = reinterpret_cast<const emu_mfma_slot_q*>(emu::wave_slot(i));
    const emu_mfma_slot_q* a1 = reinterpret_cast<const emu_mfma_slot_q*>(emu::wave_slot(i + 32));
    const float saf[2] = {ldexpf(1.f, (int)a0->sa - 127), ldexpf(1.f, (int)a1->sa - 127)};
    float acc = c[reg];
    for (int blk = 0; blk < 2; ++blk) {
      float sum = 0.f;
      for (int e = 0; e < 16; ++e) {
        sum += emu_fp8_e4m3(a0->a[16 * blk + e]) * bv[blk][e];
        sum += emu_fp8_e4m3(a1->a[16 * blk + e]) * bv[blk][16 + e];
      }
      acc += sum * saf[blk] * sbf[blk];
    }
    c[reg] = acc;
  }
  emu::wave_sync();
  return c;
}

// v_mfma_f32_16x16x32_f16: A[i=l&15][k=8*(l>>4)+e], B[k=8*(l>>4)+e][j=l&15],
// D: col=l&15, row=4*(l>>4)+reg
inline emu_floatx4 __builtin_amdgcn_mfma_f32_16x16x32_f16(emu_half8 a, emu_half8 b, emu_floatx4 c, int, int, int) {
  int l = emu::cur->lane;
  emu_mfma_slot_h s{a, b};
  memcpy(emu::wave_slot(l), &s, sizeof(s));
  emu::wave_sync();
  int j = l & 15;
  for (int reg = 0; reg < 4; ++reg) {
    int i = 4 * (l >> 4) + reg;
    float acc = c[reg];
    for (int k = 0; k < 32; ++k) {
      emu_mfma_slot_h sa, sb;
      memcpy(&sa, emu::wave_slot(i + 16 * (k >> 3)), sizeof(sa));
      memcpy(&sb, emu::wave_slot(j + 16 * (k >> 3)), sizeof(sb));
      acc += (float)sa.a[k & 7] * (float)sb.b[k & 7];
    }
    c[reg] = acc;
  }
  emu::wave_sync();
  return c;
}

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; exact k-ordered fmaf chain.
inline emu_floatx16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, emu_floatx16 c, int, int, int) {
  int l = emu::cur->lane;
  emu_mfma_slot_f s{a, b};
  memcpy(emu::wave_slot(l), &s, sizeof(s));
  emu::wave_sync();
  int j = l & 31;
  for (int reg = 0; reg < 16; ++reg) {
    int i = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
    float acc = c[reg];
    for (int k = 0; k < 2; ++k) {
      emu_mfma_slot_f sa, sb;
      memcpy(&sa, emu::wave_slot(i + 32 * k), sizeof(sa));
      memcpy(&sb, emu::wave_slot(j + 32 * k), sizeof(sb));
      acc = fmaf(sa.a, sb.b, acc);
    }
    c[reg] = acc;
  }
  emu::wave_sync();
  return c;
}

// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]
inline emu_floatx4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, emu_floatx4 c, int, int, int) {
  int l = emu::cur->lane;
  emu_mfma_slot_f s{a, b};
  memcpy(emu::wave_slot(l), &s, sizeof(s));
  emu::wave_sync();
  int j = l & 15;
  for (int reg = 0; reg < 4; ++reg) {
    int i = 4 * (l >> 4) + reg;
    float acc = c[reg];
    for (int k = 0; k < 4; ++k) {
      emu_mfma_slot_f sa, sb;
      memcpy(&sa, emu::wave_slot(i + 16 * k), sizeof(sa));
      memcpy(&sb, emu::wave_slot(j + 16 * k), sizeof(sb));
      acc = fmaf(sa.a, sb.b, acc);
    }
    c[reg] = acc;
  }
  emu::wave_sync();
  return c;
}

inline void __builtin_amdgcn_s_setprio(int) {}
inline void __builtin_amdgcn_sched_barrier(int) {}
inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}

// integer min/max exist as device overloads in HIP
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline long min(long a, long b) { return a < b ? a : b; }
inline long max(long a, long b) { return a > b ? a : b; }

// ---- LDS-DMA and explicit synchronisation builtins used by the pipelined conv kernel ------------
// global_load_lds: each lane copies `size` bytes from its own global address to
// (wave-uniform LDS base) + lane*size + offset.  The emulator performs the copy immediately, so it
// validates addressing only; the vmcnt/barrier discipline is validated on hardware.
inline void __builtin_amdgcn_global_load_lds(const __attribute__((address_space(1))) void* g,
                                             __attribute__((address_space(3))) void* l, unsigned size, int offset, int) {
  memcpy((char*)(uintptr_t)l + (size_t)emu::cur->lane * size + offset, (const void*)(uintptr_t)g, size);
}
// buffer resources (raw addressing): {base, num_records}; a load whose byte offset (voffset + immediate)
// does not fit below num_records returns zero, as the hardware bounds check does
struct emu_buffer_rsrc {
  const char* base;
  unsigned num_records;
};
typedef emu_buffer_rsrc __amdgpu_buffer_rsrc_t;
inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, short, int num_records, int) {
  return emu_buffer_rsrc{(const char*)p, (unsigned)num_records};
}
inline void __builtin_amdgcn_raw_ptr_buffer_load_lds(__amdgpu_buffer_rsrc_t r, __attribute__((address_space(3))) void* l,
                                                     unsigned size, int voffset, int soffset, int offset, int) {
  char* dst = (char*)(uintptr_t)l + (size_t)emu::cur->lane * size + offset;
  const unsigned long off = (unsigned long)(unsigned)voffset + (unsigned)offset;
  if (off + size > r.num_records)
    memset(dst, 0, size);
  else
    memcpy(dst, r.base + off + (unsigned)soffset, size);
}
typedef unsigned emu_u32x4 __attribute__((ext_vector_type(4)));
inline void __builtin_amdgcn_raw_buffer_store_b128(emu_u32x4 v, __amdgpu_buffer_rsrc_t r, int voffset, int soffset, int) {
  if ((unsigned long)(unsigned)voffset + 16 <= r.num_records) memcpy(const_cast<char*>(r.base) + (unsigned)voffset + (unsigned)soffset, &v, 16);
}
inline emu_u32x4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, int voffset, int soffset, int) {
  emu_u32x4 v = {0, 0, 0, 0};
  if ((unsigned long)(unsigned)voffset + 16 <= r.num_records) memcpy(&v, r.base + (unsigned)voffset + (unsigned)soffset, 16);
  return v;
}
typedef unsigned emu_u32x2 __attribute__((ext_vector_type(2)));
inline void __builtin_amdgcn_raw_buffer_store_b64(emu_u32x2 v, __amdgpu_buffer_rsrc_t r, int voffset, int soffset, int) {
  if ((unsigned long)(unsigned)voffset + 8 <= r.num_records) memcpy(const_cast<char*>(r.base) + (unsigned)voffset + (unsigned)soffset, &v, 8);
}
inline emu_u32x2 __builtin_amdgcn_raw_buffer_load_b64(__amdgpu_buffer_rsrc_t r, int voffset, int soffset, int) {
  emu_u32x2 v = {0, 0};
  if ((unsigned long)(unsigned)voffset + 8 <= r.num_records) memcpy(&v, r.base + (unsigned)voffset + (unsigned)soffset, 8);
  return v;
}
inline int __builtin_amdgcn_sbfe(int v, unsigned off, unsigned width) {
  return (int)((unsigned)v << (32 - off - width)) >> (32 - width);
}
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }  // callers pass wave-uniform values
inline void __builtin_amdgcn_s_waitcnt(int) {}
inline void __builtin_amdgcn_s_barrier() { emu::block_sync(); }
// on the GPU a wave's lanes run in lockstep and its LDS operations execute in order; the fibers of the emulator need an
// explicit rendezvous where one lane reads what another lane of the same wave has just written
inline void __builtin_amdgcn_wave_barrier() { emu::wave_sync(); }

// ds_read_b64_tr_b16 (gfx950 LDS transpose read), semantics measured on MI355X with
// tools/probes/tr_probe.hip: within each group of 16 lanes, result lane i element e is element (i % 4)
// of the 8-byte chunk addressed by source lane (4*e + i/4) of the same group.
typedef short emu_v4s __attribute__((__vector_size__(8)));
inline emu_v4s __builtin_amdgcn_ds_read_tr16_b64_v4i16(__attribute__((address_space(3))) emu_v4s* p) {
  int l = emu::cur->lane;
  uintptr_t addr = (uintptr_t)p;
  memcpy(emu::wave_slot(l), &addr, sizeof(addr));
  emu::wave_sync();
  emu_v4s r;
  int g = l & ~15, i = l & 15;
  for (int e = 0; e < 4; ++e) {
    uintptr_t src;
    memcpy(&src, emu::wave_slot(g + 4 * e + (i >> 2)), sizeof(src));
    short v;
    memcpy(&v, (const char*)src + 2 * (i & 3), 2);
    r[e] = v;
  }
  emu::wave_sync();
  return r;
}

// ---- hipGraph subset: the emulator does not capture (BeginCapture fails, callers fall back to eager) ----
typedef struct emu_graph* hipGraph_t;
typedef struct emu_graph_exec* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorUnknown; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorUnknown; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, void*, void*, size_t) { *e = nullptr; return hipErrorUnknown; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorUnknown; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
