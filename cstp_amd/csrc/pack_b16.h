// The bf16-storage path's weight pack (b16.hip) as a device function, shared with the replay kernel of igemm.hip (cstp_pack_replay,
// record kind 4): fp32 [kout][cin][taps] -> bf16 GEMM operand rows
//   forward:        wp[m = kout (Mp rows)][k = tap * cin + c  (Kw, zero beyond taps * cin)]
//   data gradient:  wp[m = cin  (Mp rows)][k = tap * kout + ko]
#pragma once
#include "common.h"

namespace cstp {

typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned short pack_f2bf(float a) {        // round to nearest even (v_cvt_pk_bf16_f32)
  pk_f32x2 v = {a, 0.f};
  return (unsigned short)(__builtin_bit_cast(unsigned, __builtin_convertvector(v, pk_bf16x2)) & 0xffffu);
}

__device__ __forceinline__ void pack_w_b16_body(const float* __restrict__ w, unsigned short* __restrict__ wp, int kout, int cin, int ntaps,
                                                int Mp, int Kw, int dgrad, int blk, int nblk) {
  const size_t total = (size_t)Mp * Kw;
  const int inner = dgrad ? kout : cin, mreal = dgrad ? cin : kout;
  for (size_t i = (size_t)blk * 256 + threadIdx.x; i < total; i += (size_t)nblk * 256) {
    const int k = (int)(i % Kw), m = (int)(i / Kw);
    const int tap = k / inner, c = k - tap * inner;
    float v = 0.f;
    if (m < mreal && tap < ntaps) v = dgrad ? w[((size_t)c * cin + m) * ntaps + tap] : w[((size_t)m * cin + c) * ntaps + tap];
    wp[i] = pack_f2bf(v);
  }
}

// the pack-plan hooks of igemm.hip (cstp_pack_mode / cstp_pack_register): true = the caller has replayed this workspace's pack
bool pack_skip(const void* dst);
// mode 1: append the launch about to be made to the calling thread's record list
void pack_record_b16(const float* w, void* dst, int nblocks, int kout, int cin, int ntaps, int Mp, int Kw, int dgrad);

}  // namespace cstp
