// Shared host/device helpers for libcstp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "cstp_hip.h"

namespace cstp {

extern thread_local char g_err[512];

inline int fail(const char* fmt, const char* a = "", long b = 0, long c = 0) {
  snprintf(g_err, sizeof(g_err), fmt, a, b, c);
  return 1;
}

#define CSTP_REQUIRE(cond, msg)                                              \
  do {                                                                       \
    if (!(cond)) return ::cstp::fail("%s (line %ld)", msg ": " #cond, __LINE__); \
  } while (0)

#define CSTP_LAUNCH_CHECK()                                                          \
  do {                                                                               \
    hipError_t e_ = hipGetLastError();                                               \
    if (e_ != hipSuccess) return ::cstp::fail("HIP launch failed: %s (line %ld)", hipGetErrorString(e_), __LINE__); \
  } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- device-side reductions (wave = 64 lanes) -------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}

template <typename T>
__device__ __forceinline__ T wave_sum_all(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64); result valid in thread 0.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* smem /* >= 16 entries */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum(v);
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  const int nw = (blockDim.x + 63) >> 6;
  T r = (threadIdx.x < nw) ? smem[threadIdx.x] : T(0);
  if (wave == 0) r = wave_sum(r);
  __syncthreads();
  return r;
}

}  // namespace cstp
