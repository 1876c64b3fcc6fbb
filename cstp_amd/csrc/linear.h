// Fully connected layers on a handful of rows (the projection / prediction / overlap heads, r21d_byol.py:159-176, 318-334: F.linear
// on [2B][F] with 2B <= 32): y[n][k] = sum_c x[n][c] w[k][c] + b[k] and dx[n][c] = sum_k dy[n][k] w[k][c].
//
// As 1x1x1 convolutions over 32 "positions" these calls filled four to eight CUs with one column tile each and walked the whole
// reduction axis in sequence: 109 us for 4096 -> 512 (8 MB of weights, 134 MFLOP), 1.1 ms of a 53 ms step over the six heads.
// They are weight-STREAMING problems: every weight is used N <= 32 times.  Here the reduction axis is cut into slices of 64 and
// the output features into groups of 256: a block owns (feature group, slice), every LANE owns one output feature, the slice of
// the small operand comes through scalar loads and the products are plain fp32 FMAs in a fixed order -- no cross-lane
// reduction, no LDS, no atomics; a second launch sums the slices in order (and adds the bias / the gradient already in dx).  Exact fp32
// arithmetic: these layers do not go through the f16-pair split.
//   forward : lane = feature k, its weights are 256 contiguous bytes of ITS row (two whole lines, used up inside the block)
//   gradient: lane = input feature c, the 64 rows of the slice are read as 64 coalesced row segments
#pragma once
#include "common.h"

namespace cstp {

constexpr int LIN_NMAX = 32;        // rows of the small operand
constexpr int LIN_SLICE = 64;       // reduction indices per block

// acc[j] += sum_r wv[r] * a[n0 + j][r0 + r], eight rows at a time.  The small operand's addresses are wave-uniform: its values
// arrive through SCALAR loads and enter the FMAs as SGPR operands -- no LDS, no barrier, no vector register spent on them.
// Rows past N re-read row N - 1 (finite values; their sums are not stored).
__device__ __forceinline__ void lin_products8(const float4 (&wv)[LIN_SLICE / 4], const float* __restrict__ a, int N, int R, int r0, int n0,
                                              float (&acc)[8]) {
  const float* row[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int n = n0 + j < N ? n0 + j : N - 1;
    row[j] = a + (size_t)n * R + r0;
  }
#pragma unroll
  for (int q = 0; q < LIN_SLICE / 4; ++q) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = acc[j];
      t = __builtin_fmaf(wv[q].x, row[j][4 * q + 0], t);
      t = __builtin_fmaf(wv[q].y, row[j][4 * q + 1], t);
      t = __builtin_fmaf(wv[q].z, row[j][4 * q + 2], t);
      t = __builtin_fmaf(wv[q].w, row[j][4 * q + 3], t);
      acc[j] = t;
    }
  }
}

// part[slice][n][k] = sum_{c in slice} x[n][c] * w[k][c]        grid (ceil(K / 256), C / 64); C a multiple of 64
template <int NB>
__global__ void __launch_bounds__(256)
linear_fwd_part_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ part, int N, int C, int K) {
  const int slice = blockIdx.y, c0 = slice * LIN_SLICE;
  const int k = blockIdx.x * 256 + threadIdx.x;
  float4 wv[LIN_SLICE / 4];
#pragma unroll
  for (int q = 0; q < LIN_SLICE / 4; ++q)
    wv[q] = k < K ? *reinterpret_cast<const float4*>(w + (size_t)k * C + c0 + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
  for (int n0 = 0; n0 < NB; n0 += 8) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    lin_products8(wv, x, N, C, c0, n0, acc);
    if (k < K) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (n0 + j < N) part[((size_t)slice * N + n0 + j) * K + k] = acc[j];
    }
  }
}

// part[slice][n][c] = sum_{k in slice} dy[n][k] * w[k][c]       grid (ceil(C / 256), K / 64); K a multiple of 64
template <int NB>
__global__ void __launch_bounds__(256)
linear_dgrad_part_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ part, int N, int C, int K) {
  const int slice = blockIdx.y, k0 = slice * LIN_SLICE;
  const int c = blockIdx.x * 256 + threadIdx.x;
  float wr[LIN_SLICE];
#pragma unroll
  for (int j = 0; j < LIN_SLICE; ++j) wr[j] = c < C ? w[(size_t)(k0 + j) * C + c] : 0.f;
  float4 wv[LIN_SLICE / 4];
#pragma unroll
  for (int q = 0; q < LIN_SLICE / 4; ++q) wv[q] = make_float4(wr[4 * q], wr[4 * q + 1], wr[4 * q + 2], wr[4 * q + 3]);
#pragma unroll 1
  for (int n0 = 0; n0 < NB; n0 += 8) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    lin_products8(wv, dy, N, K, k0, n0, acc);
    if (c < C) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (n0 + j < N) part[((size_t)slice * N + n0 + j) * C + c] = acc[j];
    }
  }
}

// dw[k][c] = (accumulate ? dw[k][c] : 0) + sum_n dy[n][k] * x[n][c], rows in order.     grid (ceil(C / 256), ceil(K / 16))
// lane = input feature c holding x[0 .. N)[c]; the 16 rows k of the block take dy through scalar loads.  8 MB of dw written once.
template <int NB>
__global__ void __launch_bounds__(256)
linear_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw, int N, int C, int K, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x, k0 = blockIdx.y * 16;
  const int cc = c < C ? c : C - 1;
  float xv[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) xv[n] = n < N ? x[(size_t)n * C + cc] : 0.f;
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  if (k0 + 16 <= K) {                                // (block-uniform) sixteen consecutive dy values per row: one scalar load
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const float* __restrict__ row = dy + (size_t)(n < N ? n : N - 1) * K + k0;
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = __builtin_fmaf(row[j], xv[n], acc[j]);
    }
  } else {
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const float* __restrict__ row = dy + (size_t)(n < N ? n : N - 1) * K;
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = __builtin_fmaf(row[k0 + j < K ? k0 + j : K - 1], xv[n], acc[j]);
    }
  }
  if (c < C) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (k0 + j < K) {
        float* o = dw + (size_t)(k0 + j) * C + c;
        *o = accumulate ? *o + acc[j] : acc[j];
      }
    }
  }
}

// out[n][j] = (accumulate ? out[n][j] : 0) + bias[j] + sum_s part[s][n][j], slices in order
__global__ void __launch_bounds__(256)
linear_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ out, int S, int N, int J,
                     int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * J) return;
  float a = 0.f;
  for (int s = 0; s < S; ++s) a += part[(size_t)s * N * J + i];
  if (bias != nullptr) a += bias[i % J];
  if (accumulate) a += out[i];
  out[i] = a;
}

}  // namespace cstp
