// Implicit-GEMM convolution on the bf16 matrix cores with fp32-equivalent arithmetic ("3xbf16 split").
//
// Every fp32 operand value x is written EXACTLY-to-2^-25 as the sum of three bf16 numbers
//     x = hi + mid + lo,  hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)      (each difference is exact in fp32)
// and a product a*b is accumulated in fp32 from the six partial products whose weight is >= 2^-16 relative:
//     a*b ~= ah*bh + (ah*bm + am*bh) + (ah*bl + al*bh + am*bm)            (dropped: am*bl, al*bm, al*bl <= 3 * 2^-24 |ab|)
// i.e. the error per product is of the size of ONE fp32 rounding.  Six v_mfma_f32_16x16x32_bf16 (16 cycles each for
// 16*16*32 MACs) replace the 8 v_mfma_f32_16x16x4_f32 (32 cycles each) of the native fp32 path: 96 vs 256 matrix-pipe
// cycles per 16x16x32 block product, a 2.67x higher ceiling (2516 / 6 = 419 TFLOP/s fp32-equivalent on MI355X).
//
// K1S  igemm_k1s<MT, DGRAD>:  out[m][n] = sum_k W[k][m] * Xcol[k][n]   (forward / data gradient, see igemm.hip)
//   block = 512 threads = 8 waves, tile (16*MT) x 128 positions, K-tile 32 = two independent 16-channel groups
//   (each group lies inside one filter tap, so the per-tap channel count only needs padding to 16);
//   the waves are SPECIALISED: waves 4-7 (one per SIMD) are producers -- they gather the fp32 operand, split it and
//   stage both operands into the LDS double buffer, with two K-tiles of global loads in flight per thread -- and
//   waves 0-3 (one per SIMD) are consumers that only read fragments
//   and issue MFMAs, so on every SIMD the vector ALU work of the split runs in the issue slots the matrix pipe leaves
//   free instead of alternating with it;
//   weights arrive pre-split (pack_weights_split_kernel: [group][m][plane][16] bf16), activations are gathered as fp32
//   (coalesced along positions), split in registers and stored to LDS as bf16 planes;
//   LDS image per operand row (one m or one position): [plane0: 32 k][plane1][plane2] = 192 B with the four 16-byte k-chunks of a plane XOR-swizzled by the row, which makes
//   both the 16-byte fragment reads of the MFMA (16 rows x one 8-k slice per quarter wave) and the staging stores
//   bank-conflict free;  consumer wave w owns columns 32w .. 32w+31 (two 16-column tiles) of all MT row tiles.
#pragma once

#ifndef CSTP_DIAG
#define CSTP_DIAG 0      // diagnostic builds: 1 = consumers only synchronise (producer-bound time); 2 = producers only synchronise
#endif

namespace cstp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int SPL_ROW = 12;   // uint4 (16 B) per LDS operand row: 3 planes x 64 B, no padding
// 16-byte k-chunk c (0..3) of a plane sits at c ^ spl_swz(row): with the non-contiguous 16-lane groups of ds_read_b128 and the
// 8-lane groups of ds_write_b128 this makes BOTH the fragment reads and the staging stores bank-conflict free (searched
// exhaustively over row strides 12..19 x XOR swizzles; stride 13 unswizzled reads 2-way, stride 14 writes 2-way).
__device__ __forceinline__ int spl_swz(int row) { return (row >> 1) & 3; }

// two fp32 -> three packed bf16 pairs (element 0 in the low half)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  f32x2 v = {x0, x1};
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  f32x2 hf = {__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};
  f32x2 r = v - hf;
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  f32x2 mf = {__builtin_bit_cast(float, m << 16), __builtin_bit_cast(float, m & 0xffff0000u)};
  f32x2 s = r - mf;
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(s, bf16x2));
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- the two-term variant ("2xf16 split", NP == 2 below) ------------------------------------------------------------------
// x * 2^s = hi + lo with hi = f16(x * 2^s), lo = f16(x * 2^s - hi) (the difference is exact in fp32): 22 significand bits, and
//     a*b ~= ah*bh + (ah*bl + al*bh)                                        (dropped: al*bl <= 2^-22 |ab|)
// from THREE v_mfma_f32_16x16x32_f16 -- half the matrix-pipe work, two thirds of the LDS bytes and of the split VALU work of
// the three-term bf16 variant.  Per product the error is <= 3 * 2^-22 (operand representation 2 * 2^-22 worst case, dropped
// term 2^-22), typically 6e-8 rms -- the size of the fp32 roundings any fp32 summation order commits (measured against fp64:
// profiles/r01/split_accuracy.txt).  f16 has a 5-bit exponent, so every operand is scaled by a power of two (exact) that puts
// its largest magnitude in [2^14, 2^15): per TENSOR for a gathered activation operand (absmax_kernel, or a cell its producer
// filled), per ROW for the packed weight operand (pack_weights_split2_kernel); elements more than 2^14 below the largest
// lose relative (never absolute: the f16 subnormal spacing is 2^-39 of the largest magnitude) precision.  The epilogue
// multiplies the accumulators by the inverse powers of two.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void f16_scale(unsigned absmax_bits, float& scale, float& inv) {
  const int e = (int)(absmax_bits >> 23) & 0xff;
  if (e == 0 || e == 255) { scale = 1.f; inv = 1.f; return; }      // all-zero / denormal operand, or inf / nan (propagates)
  int sh = 141 - e;                                                // |max| * 2^sh in [2^14, 2^15)
  sh = sh > 126 ? 126 : sh;
  scale = __builtin_bit_cast(float, (unsigned)(127 + sh) << 23);
  inv = __builtin_bit_cast(float, (unsigned)(127 - sh) << 23);
}

// two fp32 (times the operand scale) -> two packed f16 pairs:  h = f16(x * sc),  l = f16(x * sc - h)  (both roundings to nearest
// even; x * sc and the difference are exact in fp32).  FOUR instructions per pair of elements: the mixed-precision FMAs take the
// f16 halves of h as their addend directly and write an f16 half of the result register (v_fma_mixlo/hi_f16: fp32 FMA, then
// the conversion), where the generic form spends six (packed multiply, packed convert, two f16 -> f32 conversions, packed
// subtract, packed convert) -- the split is most of the vector-ALU work of every gather / staging wave of the split kernels.
#ifndef CSTP_SPLIT_MIX
#define CSTP_SPLIT_MIX 1
#endif
__device__ __forceinline__ void split2h(float x0, float x1, float sc, unsigned& h, unsigned& l) {
#if CSTP_SPLIT_MIX
  unsigned hh, ll;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(hh) : "v"(x0), "v"(sc));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(hh) : "v"(x1), "v"(sc));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(ll) : "v"(x0), "v"(sc), "v"(hh));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(ll) : "v"(x1), "v"(sc), "v"(hh));
  h = hh;
  l = ll;
#else
  f32x2 v = {x0, x1};
  v = v * sc;
  const f16x2 hh = __builtin_convertvector(v, f16x2);
  const f32x2 r = v - __builtin_convertvector(hh, f32x2);
  const f16x2 ll = __builtin_convertvector(r, f16x2);
  h = __builtin_bit_cast(unsigned, hh);
  l = __builtin_bit_cast(unsigned, ll);
#endif
}

// largest magnitude of a tensor, as fp32 bits with the sign cleared (a NaN compares above everything and propagates)
__global__ void absmax_kernel(const float* __restrict__ x, size_t n, unsigned* __restrict__ cell) {
  unsigned mx = 0;
  size_t head = ((16 - (reinterpret_cast<uintptr_t>(x) & 15)) & 15) >> 2;      // elements in front of the first 16-byte boundary
  head = head < n ? head : n;
  const size_t n4 = (n - head) >> 2, tail = (n - head) & 3;
  const u32x4* x4 = reinterpret_cast<const u32x4*>(x + head);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const u32x4 v = __builtin_nontemporal_load(x4 + i);
    unsigned a = v[0] & 0x7fffffffu, b = v[1] & 0x7fffffffu, c = v[2] & 0x7fffffffu, d = v[3] & 0x7fffffffu;
    a = a > b ? a : b; c = c > d ? c : d; a = a > c ? a : c;
    mx = mx > a ? mx : a;
  }
  if (blockIdx.x == 0) {
    if (threadIdx.x < head) mx = __builtin_bit_cast(unsigned, x[threadIdx.x]) & 0x7fffffffu;
    if (threadIdx.x >= 64 && threadIdx.x - 64 < tail) {
      const unsigned a = __builtin_bit_cast(unsigned, x[head + n4 * 4 + (threadIdx.x - 64)]) & 0x7fffffffu;
      mx = mx > a ? mx : a;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)mx, off, 64); mx = mx > o ? mx : o; }
  if ((threadIdx.x & 63) == 0 && mx != 0) atomicMax(cell, mx);
}

// raw buffer resource (stride 0, bounds-checked against `bytes`): out-of-range loads return 0
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void buf_load_x4(u32x4& d, unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned soff) {
  d = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
__device__ __forceinline__ void buf_load_x1(float& d, unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned soff) {
  d = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}

// weights: native [k_out][c_in][taps] -> split GEMM operand  wps[group = k/16][m (Mp)][plane (3)][k%16]  bf16,
// k = tap*Cp + c (forward, m = k_out) or tap*Cp + k_out (dgrad, m = c_in); zero padded.
__global__ void pack_weights_split_kernel(const float* __restrict__ w, unsigned short* __restrict__ wps, int kout, int cin,
                                          int ntaps, int Cp, int Mp, int ngroups, int dgrad) {
  const size_t total = (size_t)ngroups * Mp * 8;      // one thread per (group, m, pair of k)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int kp = (int)(i & 7);
    const size_t gm = i >> 3;
    const int m = (int)(gm % Mp), grp = (int)(gm / Mp);
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k = grp * 16 + kp * 2 + e;
      const int tap = k / Cp, c = k - tap * Cp;
      float x = 0.f;
      if (tap < ntaps) {
        if (!dgrad) {
          if (m < kout && c < cin) x = w[((size_t)m * cin + c) * ntaps + tap];
        } else {
          if (m < cin && c < kout) x = w[((size_t)c * cin + m) * ntaps + tap];
        }
      }
      v[e] = x;
    }
    unsigned h, mm, l;
    split2(v[0], v[1], h, mm, l);
    unsigned* dst = reinterpret_cast<unsigned*>(wps + gm * 48) + kp;   // 48 bf16 per (group, m): 3 planes x 16
    dst[0] = h;
    dst[8] = mm;
    dst[16] = l;
  }
}

// 2xf16 split weights: wps[group = k/16][m (Mp)][plane (2)][k%16] f16 with row m scaled by a power of two (inv_a[m] = its
// inverse); one block per row m: largest magnitude of the row, then split and store.  Block 0 also zeroes the `ncells`
// absmax cells of the activation operand(s), which the kernels launched next fill.
// (the body is a device function of the block's row m: pack_replay_kernel -- igemm.hip -- runs the packs of a whole network pass
//  from ONE launch with it)
template <int PAIRS>      // pairs of the row a thread keeps in registers between the maximum and the split (0: fetch twice -- the replay kernel)
__device__ __forceinline__ void
pack_split2_body(const float* __restrict__ w, unsigned* __restrict__ wps, float* __restrict__ inv_a,
                 unsigned* __restrict__ cells, int ncells, int kout, int cin, int ntaps, int Cp, int Mp, int ngroups,
                 int dgrad, const int m_in) {
  __shared__ unsigned red[4];
  const int m = __builtin_amdgcn_readfirstlane(m_in);      // (the row is block-uniform: keeps the row's address arithmetic scalar)
  const int t = threadIdx.x;
  if (m == 0 && t < ncells) cells[t] = 0;
  const int mreal = dgrad ? cin : kout, creal = dgrad ? kout : cin;
  auto fetch = [&](int k) __attribute__((always_inline)) -> float {
    const int tap = k / Cp, c = k - tap * Cp;
    if (tap >= ntaps || m >= mreal || c >= creal) return 0.f;
    return dgrad ? w[((size_t)c * cin + m) * ntaps + tap] : w[((size_t)m * cin + c) * ntaps + tap];
  };
  // one pass over the row: thread t owns the k-pairs t, t + 256, ... and keeps up to PAIRS of them in registers between the
  // maximum and the split (longer rows -- K > 512 * PAIRS -- fetch the tail a second time)
  const int npairs = ngroups * 8;
  float v0[PAIRS > 0 ? PAIRS : 1], v1[PAIRS > 0 ? PAIRS : 1];
  unsigned mx = 0;
#pragma unroll
  for (int i = 0; i < PAIRS; ++i) {
    const int pi = t + 256 * i;
    v0[i] = v1[i] = 0.f;
    if (pi < npairs) { v0[i] = fetch(2 * pi); v1[i] = fetch(2 * pi + 1); }
    const unsigned a = __builtin_bit_cast(unsigned, v0[i]) & 0x7fffffffu, b = __builtin_bit_cast(unsigned, v1[i]) & 0x7fffffffu;
    mx = mx > a ? mx : a;
    mx = mx > b ? mx : b;
  }
  for (int pi = t + 256 * PAIRS; pi < npairs; pi += 256) {
    const unsigned a = __builtin_bit_cast(unsigned, fetch(2 * pi)) & 0x7fffffffu, b = __builtin_bit_cast(unsigned, fetch(2 * pi + 1)) & 0x7fffffffu;
    mx = mx > a ? mx : a;
    mx = mx > b ? mx : b;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)mx, off, 64); mx = mx > o ? mx : o; }
  if ((t & 63) == 0) red[t >> 6] = mx;
  __syncthreads();
  mx = red[0] > red[1] ? red[0] : red[1];
  { const unsigned o = red[2] > red[3] ? red[2] : red[3]; mx = mx > o ? mx : o; }
  float sc, inv;
  f16_scale(mx, sc, inv);
  if (t == 0) inv_a[m] = inv;
  auto put = [&](int pi, float x0, float x1) __attribute__((always_inline)) {
    unsigned h, l;
    split2h(x0, x1, sc, h, l);
    unsigned* dst = wps + ((size_t)(pi >> 3) * Mp + m) * 16 + (pi & 7);       // 32 f16 per (group, m): plane 0 = dwords 0..7
    dst[0] = h;
    dst[8] = l;
  };
#pragma unroll
  for (int i = 0; i < PAIRS; ++i) {
    const int pi = t + 256 * i;
    if (pi < npairs) put(pi, v0[i], v1[i]);
  }
  for (int pi = t + 256 * PAIRS; pi < npairs; pi += 256) put(pi, fetch(2 * pi), fetch(2 * pi + 1));
}
// (the stand-alone kernel keeps its own text: as a wrapper around pack_split2_body<24> hipcc gave it 256 VGPRs instead of 83 and
//  it ran 40 us instead of 9 us per launch, rocprofv3 round 4)
__global__ void __launch_bounds__(256)
pack_weights_split2_kernel(const float* __restrict__ w, unsigned* __restrict__ wps, float* __restrict__ inv_a,
                           unsigned* __restrict__ cells, int ncells, int kout, int cin, int ntaps, int Cp, int Mp, int ngroups,
                           int dgrad) {
  __shared__ unsigned red[4];
  const int m = blockIdx.x, t = threadIdx.x;
  if (m == 0 && t < ncells) cells[t] = 0;
  const int mreal = dgrad ? cin : kout, creal = dgrad ? kout : cin;
  auto fetch = [&](int k) __attribute__((always_inline)) -> float {
    const int tap = k / Cp, c = k - tap * Cp;
    if (tap >= ntaps || m >= mreal || c >= creal) return 0.f;
    return dgrad ? w[((size_t)c * cin + m) * ntaps + tap] : w[((size_t)m * cin + c) * ntaps + tap];
  };
  // one pass over the row: thread t owns the k-pairs t, t + 256, ... and keeps up to PAIRS of them in registers between the
  // maximum and the split (longer rows -- K > 512 * PAIRS -- fetch the tail a second time)
  constexpr int PAIRS = 24;
  const int npairs = ngroups * 8;
  float v0[PAIRS], v1[PAIRS];
  unsigned mx = 0;
#pragma unroll
  for (int i = 0; i < PAIRS; ++i) {
    const int pi = t + 256 * i;
    v0[i] = v1[i] = 0.f;
    if (pi < npairs) { v0[i] = fetch(2 * pi); v1[i] = fetch(2 * pi + 1); }
    const unsigned a = __builtin_bit_cast(unsigned, v0[i]) & 0x7fffffffu, b = __builtin_bit_cast(unsigned, v1[i]) & 0x7fffffffu;
    mx = mx > a ? mx : a;
    mx = mx > b ? mx : b;
  }
  for (int pi = t + 256 * PAIRS; pi < npairs; pi += 256) {
    const unsigned a = __builtin_bit_cast(unsigned, fetch(2 * pi)) & 0x7fffffffu, b = __builtin_bit_cast(unsigned, fetch(2 * pi + 1)) & 0x7fffffffu;
    mx = mx > a ? mx : a;
    mx = mx > b ? mx : b;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)mx, off, 64); mx = mx > o ? mx : o; }
  if ((t & 63) == 0) red[t >> 6] = mx;
  __syncthreads();
  mx = red[0] > red[1] ? red[0] : red[1];
  { const unsigned o = red[2] > red[3] ? red[2] : red[3]; mx = mx > o ? mx : o; }
  float sc, inv;
  f16_scale(mx, sc, inv);
  if (t == 0) inv_a[m] = inv;
  auto put = [&](int pi, float x0, float x1) __attribute__((always_inline)) {
    unsigned h, l;
    split2h(x0, x1, sc, h, l);
    unsigned* dst = wps + ((size_t)(pi >> 3) * Mp + m) * 16 + (pi & 7);       // 32 f16 per (group, m): plane 0 = dwords 0..7
    dst[0] = h;
    dst[8] = l;
  };
#pragma unroll
  for (int i = 0; i < PAIRS; ++i) {
    const int pi = t + 256 * i;
    if (pi < npairs) put(pi, v0[i], v1[i]);
  }
  for (int pi = t + 256 * PAIRS; pi < npairs; pi += 256) put(pi, fetch(2 * pi), fetch(2 * pi + 1));
}

// NH = 128-column halves per block tile (1 or 2).  NH = 2 (256 positions per block) halves the weight-operand traffic per
// FLOP -- every block re-reads the whole packed weight matrix from L2, 6.2 of the 9.8 GB the 64->144 3x3 layer moves into
// the CUs -- and the LDS fragment reads per MFMA; each producer thread then gathers two positions.
// NP = planes per operand: 3 = bf16 triple (six products), 2 = f16 pair (three products; inv_a = per-row inverse scales of the
// packed weights, bcell = largest magnitude of the gathered tensor; 128-byte LDS rows, so the 128-column tiles fit twice
// into a CU's LDS).
// STR ("straddle", forward only): layers with fewer than 8 input channels (the 3-channel stems).  The reduction index runs
// k = tap * Cs + c with NO channel padding, so a 16-k group straddles filter taps; the source is a ZERO-PADDED copy of the
// input (pad_input_kernel: no halo checks at all), every k has its own constant offset from the position's base -- a table in
// LDS, built once per block -- and the gather adds it to the per-thread offset instead of using the scalar channel stride.
__global__ void __launch_bounds__(256)
pad_input_kernel(const float* __restrict__ x, float* __restrict__ xp, unsigned* __restrict__ cell, int planes, int D, int H,
                 int W, int pt, int ph, int pw) {
  // xp[plane][D + 2pt][H + 2ph][W + 2pw] = x[plane][D][H][W] inside, 0 around; *cell = max |x| (fp32 bits) as a by-product.
  // One WAVE per padded row at a time (coalesced along w, the row decoded once per wave): the first version decoded every
  // element with three 64-bit divisions and ran at 0.8 TB/s -- as long as the stem's GEMM itself.
  const int Dq = D + 2 * pt, Hq = H + 2 * ph, Wq = W + 2 * pw;
  const int nrows = planes * Dq * Hq;
  const int lane = threadIdx.x & 63;
  unsigned mx = 0;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < nrows; row += gridDim.x * 4) {
    const int h = row % Hq, r2 = row / Hq;
    const int d = r2 % Dq, pl = r2 / Dq;
    const int id = d - pt, ih = h - ph;
    const bool in = (unsigned)id < (unsigned)D && (unsigned)ih < (unsigned)H;
    const float* src = x + ((size_t)(pl * D + (in ? id : 0)) * H + (in ? ih : 0)) * W;
    float* dst = xp + (size_t)row * Wq;
    for (int w = lane; w < Wq; w += 64) {
      const int iw = w - pw;
      float v = 0.f;
      if (in && (unsigned)iw < (unsigned)W) v = src[iw];
      dst[w] = v;
      const unsigned a = __builtin_bit_cast(unsigned, v) & 0x7fffffffu;
      mx = mx > a ? mx : a;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)mx, off, 64); mx = mx > o ? mx : o; }
  if (lane == 0 && mx != 0) atomicMax(cell, mx);
}

constexpr int STR_KMAX = 1056;                       // table entries: 7x7x7 taps x 3 channels = 1029, padded to 16

// AFF (forward, f16 pair): the gathered tensor is z = act(src * scale + shift) with one (scale, shift) pair per (BatchNorm group,
// channel) -- the train-mode BatchNorm + ReLU in front of the temporal convolution (r21d_byol.py:94-97) applied to the raw
// operands between their load and their split, so z never exists in HBM.  The 16 channels of a producer's group are
// wave-uniform: their pairs arrive through scalar loads.  A halo / padding position is an out-of-range load (0) and must
// STAY 0 (zero padding applies to z): the clamp that implements the ReLU, med3(v, 0, cap), has cap = 0 at such positions.
// Host conditions: Cs a multiple of 16, no column tile straddles two BatchNorm groups (aff_gpos positions per group, a
// multiple of the tile's columns); *bcell is the largest magnitude of z (cstp_bn_finalize_pre).
// (AFF, row tiles <= 4: THREE blocks per CU like the plain instantiation reaches by itself at 79 VGPRs)
template <int MT, bool DGRAD, int NH, int NP, bool STR = false, bool AFF = false>
__global__ void __launch_bounds__(512, (NP == 2 && NH == 1) ? ((AFF && MT <= 4) ? 6 : 4) : 1)      // the f16-pair 128-column tiles: two blocks per CU (<= 128 VGPRs)
igemm_k1s(const Geom g, const uint4* __restrict__ wps, const float* __restrict__ src, const float* __restrict__ bias,
          float* __restrict__ out, int n_tiles_x, int n_tiles_m, const float* __restrict__ inv_a,
          const unsigned* __restrict__ bcell, const float2* __restrict__ aff_ss, int aff_gpos, int aff_relu) {
  static_assert(NP == 2 || NP == 3, "planes per operand");
  static_assert(!AFF || (NP == 2 && !DGRAD && !STR), "the fused input transform serves the f16-pair forward gather");
  constexpr int BM = 16 * MT, BN = 128 * NH;
  constexpr int NC = 2 * NH;                         // 16-column tiles per consumer wave
  constexpr int ACPR = 2 * NP;                       // 16-byte chunks per A row and 16-k group
  constexpr int A_CH = BM * ACPR;                    // 16-byte chunks per A half-tile (one 16-k group)
  // LDS image: NP == 3: 192-byte rows, chunk c of a plane at c ^ spl_swz(row) (see SPL_ROW); NP == 2: 128-byte rows, chunk
  // L = 4 * plane + c at L ^ (row & 7) -- the 16 rows x 4 chunks of a fragment read and the 8 consecutive rows of a staging
  // store each cover 16 / 8 distinct 16-byte slots of the 256-byte bank line (two rows per line).
  constexpr int ROWC = NP == 2 ? 8 : SPL_ROW;
  __shared__ uint4 As[2][BM * ROWC];
  __shared__ uint4 Bs[2][BN * ROWC];
  auto chunk_at = [](int row, int plane, int c) __attribute__((always_inline)) -> int {
    return NP == 2 ? ((plane * 4 + c) ^ (row & 7)) : (plane * 4 + (c ^ spl_swz(row)));
  };
  static_assert(!STR || (!DGRAD && NH == 1), "the straddle mode serves the forward 128-column tiles");
  __shared__ int vtap[28];                          // DGRAD: the taps that hit this stride-parity class, in order
  __shared__ __attribute__((aligned(16))) unsigned ktab[STR ? STR_KMAX : 4];   // STR: byte offset of reduction index k
  __shared__ __attribute__((aligned(16))) float inva_s[NP == 2 ? BM : 4];   // NP == 2: inverse row scales of this row tile

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // in an SGPR: everything derived from it stays scalar

  // XCD-aware tile order (see igemm_k1)
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int mtile = slot % n_tiles_m;
  const int chunk = (n_tiles_x + 7) >> 3;
  const int nt_in = slot / n_tiles_m;
  const int ntile = xcd * chunk + nt_in;
  if (nt_in >= chunk || ntile >= n_tiles_x) return;

  int zt = 0, zh = 0, zw = 0;
  int Dp = g.Dp, Hp = g.Hp, Wp = g.Wp;
  if (DGRAD) {
    int z = blockIdx.y;
    zw = z % g.sw; z /= g.sw;
    zh = z % g.sh; zt = z / g.sh;
    Dp = (g.Dp - zt + g.st - 1) / g.st;
    Hp = (g.Hp - zh + g.sh - 1) / g.sh;
    Wp = (g.Wp - zw + g.sw - 1) / g.sw;
  }
  const int npos = g.Nb * Dp * Hp * Wp;
  const int n0 = ntile * BN;
  if (n0 >= npos) return;
  const int m0 = mtile * BM;

  const int khw = g.kh * g.kw, ntaps = g.kt * khw;
  const int gpt = g.Cp >> 4;                         // 16-channel groups per tap

  // ---- the tap sequence (block-uniform)
  int nvt = ntaps;
  if (DGRAD) {
    if (t == 0) {
      int c = 0;
      for (int tp = 0; tp < ntaps; ++tp) {
        const int dt = tp / khw, rr = tp - dt * khw, dh = rr / g.kw, dw = rr - dh * g.kw;
        const int et = zt + g.pt - dt, eh = zh + g.ph - dh, ew = zw + g.pw - dw;
        if ((et % g.st) == 0 && (eh % g.sh) == 0 && (ew % g.sw) == 0) vtap[c++] = tp;
      }
      vtap[27] = c;
    }
    __syncthreads();
    nvt = __builtin_amdgcn_readfirstlane(vtap[27]);
  }
  if (STR) {
    const int HWq = g.Hs * g.Ws, kreal = ntaps * g.Cs;
    for (int k = t; k < g.Ktot; k += 512) {
      const int kk = k < kreal ? k : kreal - 1;       // the padding k's carry zero weights: any in-range address will do
      const int tp = kk / g.Cs, c = kk - tp * g.Cs;
      const int dt = tp / khw, rr = tp - dt * khw, dh = rr / g.kw, dw = rr - dh * g.kw;
      ktab[k] = (unsigned)((c * g.Ds + dt) * HWq + dh * g.Ws + dw) * 4u;
    }
    __syncthreads();
  }
  const int ngroups = STR ? (g.Ktot >> 4) : nvt * gpt;
  const int ntiles = (ngroups + 1) >> 1;

  if (wave >= 4) {
    // =================================== producer waves: gather, split, stage ===================================
    // 256 threads: thread (col, g2) gathers the 16 channels of group g2 (waves 4,5: first 16-k group of the K-tile,
    // waves 6,7: second) for ONE position and 1/128 of that group's A half-tile.
    // All global reads are raw BUFFER loads: the wave-uniform part of every address (channel block, A half-tile) rides
    // in the scalar offset, the per-thread part is a VGPR that only changes with the filter tap, and a masked element
    // (halo / padding position, missing half-tile) is an out-of-range offset that the hardware answers with 0 -- no
    // address arithmetic and no selects in the loop; the vector ALU only does the bf16 split.
    // TWO K-tiles of raw operands are in flight per thread (see the steady-state loop below for how the compiler is
    // brought to emit counted s_waitcnt for them).
    const int tp_ = t - 256;
    const int col = tp_ & 127, g2 = (wave - 4) >> 1;
    const int HWs = g.Hs * g.Ws, DHWs = g.Ds * HWs;
    bool nvalid[NH];
    int npd[NH], nph[NH], npw[NH];
    unsigned src_b4[NH];
    constexpr unsigned OOB = 0x80000000u;            // host guarantees both buffers are < 2 GiB
#pragma unroll
    for (int h = 0; h < NH; ++h) {                    // my position in each 128-column half
      nvalid[h] = (n0 + col + 128 * h) < npos;
      int n = nvalid[h] ? (n0 + col + 128 * h) : 0;
      npw[h] = n % Wp; n /= Wp;
      nph[h] = n % Hp; n /= Hp;
      npd[h] = n % Dp;
      const int nb = n / Dp;
      src_b4[h] = (unsigned)((size_t)nb * g.Cs * DHWs) * 4u;
    }
    const __amdgpu_buffer_rsrc_t rs_src = make_rsrc(src, (unsigned)((size_t)g.Nb * g.Cs * DHWs * 4));
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(wps, (unsigned)((size_t)(g.Ktot >> 4) * g.Mp * (32 * NP)));
    float sb = 1.f;                                   // operand scale of the gathered tensor (NP == 2)
    if (NP == 2) { float inv_unused; f16_scale(__builtin_amdgcn_readfirstlane(*bcell), sb, inv_unused); }
    // A half-tile of my group: 16-byte chunk idc = col + 128 j of BM*6, j = 0..A_IT-1 (the last may be partial)
    constexpr int A_IT = (A_CH + 127) / 128;
    static_assert(A_IT <= 7, "A staging holds at most 7 chunks per producer thread");
    // NP == 2: rows in the order r, r+2, r+1, r+3 inside every group of 16 chunks (bits 2 and 3 of the chunk index swapped):
    // the two rows an 8-lane store group touches then differ in bit 1 and land on disjoint 16-byte slots (rows r, r+1 share them)
    const int cola = NP == 2 ? ((col & ~12) | (((col >> 2) & 1) << 3) | (((col >> 3) & 1) << 2)) : col;
    const bool a_last_ok = cola + 128 * (A_IT - 1) < A_CH;
    const unsigned va_full = (unsigned)cola * 16u;
    const unsigned va_last = a_last_ok ? va_full : OOB;
    int a_lds[A_IT];
#pragma unroll
    for (int j = 0; j < A_IT; ++j) {
      int idc = cola + 128 * j;
      if (idc >= A_CH) idc = 0;
      const int row = idc / ACPR, w6 = idc - row * ACPR;
      a_lds[j] = row * ROWC + chunk_at(row, w6 >> 1, g2 * 2 + (w6 & 1));
    }
    const int b_lds = col * ROWC;
    int bq[3][2];                                     // my two chunks of every plane (the swizzle uses row bits 0..2 only:
#pragma unroll                                        //   the same in both 128-column halves)
    for (int pl = 0; pl < 3; ++pl) { bq[pl][0] = chunk_at(col, pl, g2 * 2); bq[pl][1] = chunk_at(col, pl, g2 * 2 + 1); }
    const unsigned ch4 = (unsigned)DHWs * 4u;         // byte stride between channels

    // my group sequence: e = g2, g2 + 2, ... ; (ord, cg) = (tap ordinal, 16-channel block inside the tap), kept
    // incrementally (no division in the loop)
    const int gpt_ = STR ? 1 : gpt;                   // (STR: gpt is 0 -- no per-tap channel groups)
    int ord = g2 / gpt_, cg = g2 - ord * gpt_;
    int e = g2;
    unsigned vb0[NH];                                 // per-thread byte offset of channel 0 at the current tap (or OOB); the
                                                      // channel stride rides in the scalar offset of each load
    auto set_tap = [&]() __attribute__((always_inline)) {
      const int tp = DGRAD ? __builtin_amdgcn_readfirstlane(vtap[ord < nvt ? ord : 0]) : ord;
      const int dt = tp / khw, rr = tp - dt * khw, dh = rr / g.kw, dw = rr - dh * g.kw;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        int id, ih, iw;
        if (DGRAD) {
          const int et = zt + g.pt - dt, eh = zh + g.ph - dh, ew = zw + g.pw - dw;
          id = npd[h] + et / g.st; ih = nph[h] + eh / g.sh; iw = npw[h] + ew / g.sw;
        } else {
          id = npd[h] * g.st - g.pt + dt; ih = nph[h] * g.sh - g.ph + dh; iw = npw[h] * g.sw - g.pw + dw;
        }
        const bool ok = nvalid[h] && (unsigned)id < (unsigned)g.Ds && (unsigned)ih < (unsigned)g.Hs && (unsigned)iw < (unsigned)g.Ws;
        vb0[h] = ok ? src_b4[h] + (unsigned)(id * HWs + ih * g.Ws + iw) * 4u : OOB;
      }
      return tp;
    };
    int tap = 0;
    if (STR) {
      // zero-padded source: the window of output position (d, h, w) starts at (d st, h sh, w sw), always inside
#pragma unroll
      for (int h = 0; h < NH; ++h)
        vb0[h] = nvalid[h] ? src_b4[h] + (unsigned)((npd[h] * g.st) * HWs + (nph[h] * g.sh) * g.Ws + npw[h] * g.sw) * 4u : OOB;
    } else {
      tap = set_tap();
    }

    u32x4 ra0[A_IT], ra1[A_IT];
    float rb0[NH][16], rb1[NH][16];
    // AFF: what a register set needs at split time -- its channel block and, per position, the clamp's upper end
    struct AffSet { int cg; float cap[NH]; };
    AffSet as0 = {}, as1 = {};
    const float2* const ssg = AFF ? aff_ss + (size_t)(n0 / aff_gpos) * g.Cs : nullptr;       // my tile's BatchNorm group (uniform)

    // issue the loads of my current group into the given register set, then advance to my next group
    auto issue_loads = [&](u32x4 (&ra)[A_IT], float (&rb)[NH][16], AffSet& as) __attribute__((always_inline)) {
      const bool have = e < ngroups;                  // uniform; a missing group loads zeros (OOB offsets)
      if constexpr (AFF) {
        as.cg = cg;
#pragma unroll
        for (int h = 0; h < NH; ++h) as.cap[h] = vb0[h] != OOB ? __builtin_inff() : 0.f;
      }
      const unsigned sa = (unsigned)(((size_t)(STR ? e : tap * gpt + cg) * g.Mp + m0) * (32 * NP));
      const unsigned vfull = have ? va_full : OOB, vlast = have ? va_last : OOB;
#pragma unroll
      for (int j = 0; j < A_IT; ++j) buf_load_x4(ra[j], j == A_IT - 1 ? vlast : vfull, rs_w, sa + 2048u * j);
      if constexpr (STR) {
        // every k of the group has its own offset: four 16-byte reads of the table (same address in all lanes: broadcast)
        const uint4* kt4 = reinterpret_cast<const uint4*>(ktab + 16 * (have ? e : 0));
        const unsigned vo = have ? vb0[0] : OOB;
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
          const uint4 o = kt4[j4];
          buf_load_x1(rb[0][4 * j4 + 0], vo + o.x, rs_src, 0);
          buf_load_x1(rb[0][4 * j4 + 1], vo + o.y, rs_src, 0);
          buf_load_x1(rb[0][4 * j4 + 2], vo + o.z, rs_src, 0);
          buf_load_x1(rb[0][4 * j4 + 3], vo + o.w, rs_src, 0);
        }
        e += 2;
        return;
      }
      const unsigned sb = have ? (unsigned)(cg << 4) * ch4 : 0u;    // scalar part: first channel of the block
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const unsigned vo = have ? vb0[h] : OOB;      // the range check looks at the vector offset only
#pragma unroll
        for (int j = 0; j < 16; ++j) buf_load_x1(rb[h][j], vo, rs_src, sb + ch4 * (unsigned)j);
      }
      e += 2;
      cg += 2;
      if (cg >= gpt) {                                // uniform: next tap(s)
        do { cg -= gpt; ++ord; } while (cg >= gpt);
        tap = set_tap();
      }
    };

    auto split_store = [&](int buf, u32x4 (&ra)[A_IT], float (&rb)[NH][16], const AffSet& as) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < A_IT; ++j)
        if (j < A_IT - 1 || a_last_ok) As[buf][a_lds[j]] = make_uint4(ra[j].x, ra[j].y, ra[j].z, ra[j].w);
      if constexpr (AFF) {
        // the group's 16 (scale, shift) pairs: wave-uniform, read with scalar loads (no vector registers: the kernel keeps its
        // three blocks per CU); one packed FMA per two elements, one clamp per element
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const float2* tb = ssg + (as.cg << 4);
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
          const float2 p0 = tb[j], p1 = tb[j + 1];
          const f32x2 a2 = {p0.x, p1.x}, b2 = {p0.y, p1.y};
#pragma unroll
          for (int h = 0; h < NH; ++h) {
            const f32x2 v2 = {rb[h][j], rb[h][j + 1]};
            const f32x2 z2 = __builtin_elementwise_fma(v2, a2, b2);
            const float lo = aff_relu ? 0.f : -as.cap[h];
            rb[h][j] = __builtin_amdgcn_fmed3f(z2[0], lo, as.cap[h]);
            rb[h][j + 1] = __builtin_amdgcn_fmed3f(z2[1], lo, as.cap[h]);
          }
        }
      }
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        uint4* brow = &Bs[buf][b_lds + 128 * h * ROWC];
        if constexpr (NP == 2) {
          uint4 ph[2], pl[2];
          unsigned hh, ll;
#define CSTP_SPLITH(J, DST, F) split2h(rb[h][J], rb[h][(J) + 1], sb, hh, ll); ph[DST].F = hh; pl[DST].F = ll;
          CSTP_SPLITH(0, 0, x) CSTP_SPLITH(2, 0, y) CSTP_SPLITH(4, 0, z) CSTP_SPLITH(6, 0, w)
          CSTP_SPLITH(8, 1, x) CSTP_SPLITH(10, 1, y) CSTP_SPLITH(12, 1, z) CSTP_SPLITH(14, 1, w)
#undef CSTP_SPLITH
          brow[bq[0][0]] = ph[0]; brow[bq[0][1]] = ph[1];
          brow[bq[1][0]] = pl[0]; brow[bq[1][1]] = pl[1];
          continue;
        }
        uint4 ph[2], pm[2], pl[2];
        unsigned hh, mm, ll;
#define CSTP_SPLIT(J, DST, F) split2(rb[h][J], rb[h][(J) + 1], hh, mm, ll); ph[DST].F = hh; pm[DST].F = mm; pl[DST].F = ll;
        CSTP_SPLIT(0, 0, x) CSTP_SPLIT(2, 0, y) CSTP_SPLIT(4, 0, z) CSTP_SPLIT(6, 0, w)
        CSTP_SPLIT(8, 1, x) CSTP_SPLIT(10, 1, y) CSTP_SPLIT(12, 1, z) CSTP_SPLIT(14, 1, w)
#undef CSTP_SPLIT
        brow[bq[0][0]] = ph[0]; brow[bq[0][1]] = ph[1];
        brow[bq[1][0]] = pm[0]; brow[bq[1][1]] = pm[1];
        brow[bq[2][0]] = pl[0]; brow[bq[2][1]] = pl[1];
      }
    };

    // tile i lives in set (i & 1) and is staged into LDS buffer (i & 1) during the consumption of tile i - 1.
    // The steady-state loop contains NO conditional load: the compiler's s_waitcnt accounting then knows that exactly
    // one younger tile (NLOADS loads) is outstanding when a set is consumed and emits counted waits; with a load under
    // `if` it must assume the younger loads may not exist and waits for everything, exposing the memory latency.
    int i = 0;
#if CSTP_DIAG == 2
    __syncthreads();
    for (; i < ntiles; ++i) __syncthreads();
    return;
#endif
    if (ntiles >= 4) {
      issue_loads(ra0, rb0, as0);                     // tile 0
      issue_loads(ra1, rb1, as1);                     // tile 1
      split_store(0, ra0, rb0, as0);
      issue_loads(ra0, rb0, as0);                     // tile 2
      __syncthreads();
      while (i + 4 < ntiles) {
        split_store(1, ra1, rb1, as1);                // tile i+1
        issue_loads(ra1, rb1, as1);                   // tile i+3
        __syncthreads();
        split_store(0, ra0, rb0, as0);                // tile i+2
        issue_loads(ra0, rb0, as0);                   // tile i+4
        __syncthreads();
        i += 2;
      }
    } else {
      if (ntiles > 0) issue_loads(ra0, rb0, as0);
      if (ntiles > 1) issue_loads(ra1, rb1, as1);
      if (ntiles > 0) {
        split_store(0, ra0, rb0, as0);
        if (ntiles > 2) issue_loads(ra0, rb0, as0);
      }
      __syncthreads();
    }
    for (; i < ntiles; i += 2) {                      // tail (and the whole loop of short reductions)
      if (i + 1 < ntiles) {                           // stage tile i+1 (set 1) while tile i is consumed
        split_store(1, ra1, rb1, as1);
        if (i + 3 < ntiles) issue_loads(ra1, rb1, as1);
      }
      __syncthreads();
      if (i + 1 >= ntiles) break;
      if (i + 2 < ntiles) {                           // stage tile i+2 (set 0) while tile i+1 is consumed
        split_store(0, ra0, rb0, as0);
        if (i + 4 < ntiles) issue_loads(ra0, rb0, as0);
      }
      __syncthreads();
    }
    return;
  }

  // ===================================== consumer waves: LDS -> MFMA -> output =====================================
  // lane l feeds A[m = l&15][k = 8*(l>>4) + j] and B[k = 8*(l>>4) + j][n = l&15]; wave w owns columns 32*NH*w .. +32*NH-1
  const int wn = wave;
  const int fr = lane & 15, fk = lane >> 4;
  f32x4 acc[MT][NC];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][c][r] = 0.f;

  if (NP == 2 && t < BM) inva_s[t] = inv_a[m0 + t];  // rows < Mp: always readable; read back after the K loop's barriers
#if CSTP_DIAG == 1
  __syncthreads();
  for (int i = 0; i < ntiles; ++i) __syncthreads();
  if (ntiles >= 0) return;
#endif
  __syncthreads();
  {
    int buf = 0;
    for (int i = 0; i < ntiles; ++i) {
      // fragment rows are fr + 16 * tile: (row & 7) == (fr & 7), so the chunk positions are per-lane constants
      const uint4* Bb = &Bs[buf][(wn * 32 * NH + fr) * ROWC];
      const uint4* Ab = &As[buf][fr * ROWC];
      const int q0 = chunk_at(fr, 0, fk), q1 = chunk_at(fr, 1, fk), q2 = chunk_at(fr, 2, fk);
      if constexpr (NP == 2) {
        f16x8 bh[NC], bl[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          bh[c] = __builtin_bit_cast(f16x8, Bb[c * 16 * ROWC + q0]);
          bl[c] = __builtin_bit_cast(f16x8, Bb[c * 16 * ROWC + q1]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f16x8 ah = __builtin_bit_cast(f16x8, Ab[mt * 16 * ROWC + q0]);
          const f16x8 al = __builtin_bit_cast(f16x8, Ab[mt * 16 * ROWC + q1]);
#pragma unroll
          for (int c = 0; c < NC; c += 2) {   // small terms first; two column tiles' accumulation chains interleaved
            f32x4 a0 = acc[mt][c], a1 = acc[mt][c + 1];
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[c], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[c + 1], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[c], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[c + 1], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[c], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[c + 1], a1, 0, 0, 0);
            acc[mt][c] = a0;
            acc[mt][c + 1] = a1;
          }
        }
      } else {
        bf16x8 bh[NC], bm[NC], bl[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          bh[c] = __builtin_bit_cast(bf16x8, Bb[c * 16 * ROWC + q0]);
          bm[c] = __builtin_bit_cast(bf16x8, Bb[c * 16 * ROWC + q1]);
          bl[c] = __builtin_bit_cast(bf16x8, Bb[c * 16 * ROWC + q2]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const bf16x8 ah = __builtin_bit_cast(bf16x8, Ab[mt * 16 * ROWC + q0]);
          const bf16x8 am = __builtin_bit_cast(bf16x8, Ab[mt * 16 * ROWC + q1]);
          const bf16x8 al = __builtin_bit_cast(bf16x8, Ab[mt * 16 * ROWC + q2]);
#pragma unroll
          for (int c = 0; c < NC; c += 2) {   // smallest terms first; two column tiles' accumulation chains interleaved
            f32x4 a0 = acc[mt][c], a1 = acc[mt][c + 1];
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[c], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[c + 1], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[c], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[c + 1], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm[c], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm[c + 1], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh[c], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh[c + 1], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm[c], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm[c + 1], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[c], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[c + 1], a1, 0, 0, 0);
            acc[mt][c] = a0;
            acc[mt][c + 1] = a1;
          }
        }
      }
      __syncthreads();
      buf ^= 1;
    }
  }

  // ---- epilogue: C layout col = lane&15, row = (lane>>4)*4 + reg; per pair of column tiles the lane groups q and q^1
  // swap one register so that 32 consecutive lanes hold 32 consecutive columns of ONE row (whole 128-byte lines)
  const int lcol = lane & 31;
  const int q = lane >> 4;
  const bool odd = (q & 1) != 0;
  // NP == 2: undo the operand scales (powers of two: exact).  Lane group q stores rows (q & ~1) * 4 + 0..7 of every row tile:
  // their inverse scales come out of LDS in two 16-byte reads per row tile, outside the conditional stores.
  float invb = 1.f;
  if (NP == 2) { float sc_unused; f16_scale(__builtin_amdgcn_readfirstlane(*bcell), sc_unused, invb); }
#pragma unroll
  for (int pr = 0; pr < NH; ++pr) {
    const int n = n0 + wn * 32 * NH + pr * 32 + lcol;
    size_t obase = 0, cstride = 0;
    const bool nok = n < npos;
    if (nok) {
      if (DGRAD) {
        int qq = n;
        const int pw = qq % Wp; qq /= Wp;
        const int ph = qq % Hp; qq /= Hp;
        const int pd = qq % Dp; const int b = qq / Dp;
        const int HWf = g.Hp * g.Wp;
        cstride = (size_t)g.Dp * HWf;
        obase = (size_t)b * g.M * cstride + (size_t)(zt + g.st * pd) * HWf + (zh + g.sh * ph) * g.Wp + (zw + g.sw * pw);
      } else {
        const int S = Dp * Hp * Wp;
        const int b = n / S, sp = n - b * S;
        cstride = (size_t)S;
        obase = (size_t)b * g.M * cstride + sp;
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f32x4 ia0 = {1.f, 1.f, 1.f, 1.f}, ia1 = ia0;
      if (NP == 2) {
        ia0 = *reinterpret_cast<const f32x4*>(&inva_s[mt * 16 + (q & ~1) * 4]) * invb;
        ia1 = *reinterpret_cast<const f32x4*>(&inva_s[mt * 16 + (q & ~1) * 4 + 4]) * invb;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v0 = acc[mt][2 * pr][r], v1 = acc[mt][2 * pr + 1][r];
        const float recv = __shfl_xor(odd ? v0 : v1, 16, 64);
        const int m_even = m0 + mt * 16 + (q & ~1) * 4 + r, m_odd = m_even + 4;
        float ve = odd ? recv : v0;
        float vo = odd ? v1 : recv;
        if (NP == 2) { ve *= ia0[r]; vo *= ia1[r]; }
        if (nok && m_even < g.M) {
          if (bias != nullptr) ve += bias[m_even];
          if (g.acc) ve += out[obase + (size_t)m_even * cstride];
          CSTP_STORE(out + obase + (size_t)m_even * cstride, ve);
        }
        if (nok && m_odd < g.M) {
          if (bias != nullptr) vo += bias[m_odd];
          if (g.acc) vo += out[obase + (size_t)m_odd * cstride];
          CSTP_STORE(out + obase + (size_t)m_odd * cstride, vo);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// K2S  igemm_k2s<MT>:  weight gradient  dwp[m][j] += sum_{n in split} dY[m][n] * Xcol[j][n]   on the split path.
// BOTH operands are fp32 activations whose memory-contiguous axis (positions n) is the REDUCTION axis, while the MFMA
// wants 8 consecutive k per lane.  Producer threads therefore stay position-major -- thread (n, slot) gathers 16 rows m
// of dY and 16 channels of one filter tap of x at its position (coalesced along n, one position decode per tile) --
// split them and stage LDS images [k = n][column] of 8-byte pieces (4 consecutive columns of one k-row), and the consumers
// read the k-contiguous MFMA fragments with the hardware-transposing ds_read_b64_tr_b16 (lane 4q+p of a 16-lane group
// supplies row q / columns 4p..4p+3, lane i receives column i of the 4 rows).
// Images per plane: main = 32 k-rows x 32 pieces (128 columns), piece c4 of row r at c4 ^ 4*((r&3)|((r>>3&1)<<2)) ^ 2*(r>>2&1):
// the 32 pieces one half-wave transposed read touches (rows r0..r0+3 and r0+8..r0+11) cover all 64 banks once, and the
// 16-byte halves the 8 lanes of a ds_write_b128 group write are distinct; MT == 9 adds a 32 x 4-piece image for rows 128..143
// with its k-rows permuted so that the same 8 rows are contiguous.  Rows m >= M, channels >= Cs and taps beyond the filter
// are CLAMPED (finite garbage whose products land in outputs nobody reads); positions outside the split / the tensor are
// out-of-range buffer offsets = zeros.  Tile (16*MT) x 128 outputs, K-tile 32 positions, split-K with f32 atomics as K2.
// NP as in igemm_k1s; NP == 2: *xcell / *dycell = largest magnitude of x / dY, the slab receives the SCALED sums and
// unpack_wgrad_kernel multiplies by the inverse powers of two.
// STR (NP == 2 only): layers with fewer than 8 input channels (the 3-channel stems).  The columns run j = tap * Cs + c with NO
// channel padding (147 columns for the 1x7x7 stem instead of 49 taps x 32), the source is the ZERO-PADDED copy of x
// (pad_input_kernel), and each of a producer's 16 columns has its own constant offset from the position's base.
// AFF (NP == 2): x stands for z = act(x * scale + shift), the BatchNorm + ReLU in front of the convolution recomputed in this
// gather exactly as igemm_k1s<.., AFF> applied it in the forward pass (*xcell = largest magnitude of z).  A K-tile's 32
// positions lie in one BatchNorm group (aff_gpos positions per group, a multiple of 32: host condition); the (scale, shift)
// pairs of the block's 128 columns sit in LDS for every group and a producer reads its 16 pairs per tile (same address across
// its 32 lanes: broadcast).
template <int MT, int NP, bool STR = false, bool AFF = false>
__global__ void __launch_bounds__(512, NP == 2 ? 4 : 1)      // f16 pair: the LDS images fit twice into a CU (4 waves per SIMD: <= 128 VGPRs)
igemm_k2s(const Geom g, const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dwp, int Jtot, int Jp,
          int ktiles_total, int ktiles_per_split, int ntm, int ntj, int nsplit, const unsigned* __restrict__ xcell,
          const unsigned* __restrict__ dycell, size_t det_stride, const float2* __restrict__ aff_ss, int aff_gpos, int aff_groups,
          int aff_relu) {
  static_assert(MT == 4 || MT == 8 || MT == 9, "row tiles: 64 or 128 main rows (+16)");
  static_assert(NP == 2 || NP == 3, "planes per operand");
  static_assert(!AFF || (NP == 2 && !STR), "the fused input transform serves the f16-pair gather");
  __shared__ __attribute__((aligned(16))) float aff_a[AFF ? 2 * 128 : 4], aff_b[AFF ? 2 * 128 : 4];   // scale / shift [group (<= 2)][column of this block]
  constexpr int BM = 16 * MT, BJ = 128;
  constexpr bool XTRA = MT == 9;
  constexpr int AP = MT == 4 ? 16 : 32;              // 8-byte pieces per k-row of the dY image (64 / 128 columns)
  constexpr int AR = AP / 2;                         // dY rows per producer slot (8 / 16)
  __shared__ uint2 Am[2][NP][32 * AP];
  __shared__ uint2 Bm[2][NP][32 * 32];
  __shared__ uint2 Ax[XTRA ? 2 : 1][NP][32 * 4];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  // XCD-aware order as igemm_k2: the ntj column tiles reading one dY panel sit on consecutive slots of one XCD
  const int xcd = blockIdx.x & 7, slot_b = blockIdx.x >> 3;
  const int jt = slot_b % ntj;
  const int panel = (slot_b / ntj) * 8 + xcd;
  if (panel >= ntm * nsplit) return;
  const int mtile = panel % ntm, split = panel / ntm;
  const int m0 = mtile * BM, j0 = jt * BJ;
  const int S = g.Dp * g.Hp * g.Wp;
  const int npos = g.Nb * S;
  const int kt_begin = split * ktiles_per_split;
  int kt_end = kt_begin + ktiles_per_split;
  if (kt_end > ktiles_total) kt_end = ktiles_total;
  if (kt_begin >= kt_end) return;
  const int ntiles = kt_end - kt_begin;
  const int n_end = kt_end * 32 < npos ? kt_end * 32 : npos;

  auto pc = [](int r, int c4) __attribute__((always_inline)) -> int {
    return c4 ^ (4 * ((r & 3) | (((r >> 3) & 1) << 2))) ^ (2 * ((r >> 2) & 1));
  };
  // 16-piece rows (MT == 4): reads conflict-free, the 16-byte stores 2-way
  auto pca = [&](int r, int c4) __attribute__((always_inline)) -> int {
    return AP == 32 ? pc(r, c4) : (c4 ^ (4 * (((r >> 1) & 1) | (((r >> 3) & 1) << 1))) ^ (2 * ((r >> 2) & 1)));
  };
  auto prow = [](int r) __attribute__((always_inline)) -> int {
    return (r & 3) | (((r >> 3) & 1) << 2) | (((r >> 2) & 1) << 3) | (r & 16);
  };

  if constexpr (AFF) {
    // column j0 + col = (tap, channel), clamped as the producers' offsets below are
    if (t < aff_groups * 128) {
      const int grp = t >> 7, col = t & 127;
      const int jg = j0 + col;
      int c = jg - (jg / g.Cp) * g.Cp;
      c = c < g.Cs ? c : g.Cs - 1;
      const float2 p = aff_ss[grp * g.Cs + c];
      aff_a[t] = p.x; aff_b[t] = p.y;
    }
    __syncthreads();
  }

  if (wave >= 4) {
    // ================================================= producers =================================================
    const int tp_ = t - 256;
    const int r = tp_ & 31, q = tp_ >> 5;            // k-row (position inside the tile), slot 0..7
    const int HWs = g.Hs * g.Ws, DHWs = g.Ds * HWs;
    const int HWo = g.Hp * g.Wp;
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_dy =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, (int)((size_t)g.Nb * g.M * S * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)((size_t)g.Nb * g.Cs * DHWs * 4), 0x00020000);
    // loop-invariant per-thread offsets: my 16 (+2) dY rows and my 16 channels of my tap
    unsigned moff[AR], mxoff[2], coff[16];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
      int m = m0 + AR * q + j;
      m = m < g.M ? m : g.M - 1;
      moff[j] = (unsigned)m * (unsigned)S * 4u;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int m = m0 + 128 + 2 * q + j;
      m = m < g.M ? m : g.M - 1;
      mxoff[j] = (unsigned)m * (unsigned)S * 4u;
    }
    const int khw = g.kh * g.kw, ntaps = g.kt * khw;
    int tap, c0;
    {
      const int jg = j0 + 16 * q;
      tap = jg / g.Cp;
      c0 = jg - tap * g.Cp;
      if (tap >= ntaps) tap = ntaps - 1;
    }
    int dt = tap / khw, rr_ = tap - dt * khw, dh = rr_ / g.kw, dw = rr_ - dh * g.kw;
    if constexpr (STR) {
      // my 16 columns (tap, channel) of the padded source, each a constant offset from the position's base; columns past
      // the last one repeat it (finite values in slab cells nobody reads)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        int col = j0 + 16 * q + j;
        col = col < Jtot ? col : Jtot - 1;
        const int tp2 = col / g.Cs, c = col - tp2 * g.Cs;
        const int t2 = tp2 / khw, r2 = tp2 - t2 * khw, h2 = r2 / g.kw, w2 = r2 - h2 * g.kw;
        coff[j] = (unsigned)(((c * g.Ds + t2) * g.Hs + h2) * g.Ws + w2) * 4u;
      }
      dt = dh = dw = 0;                              // (the tap offsets live in coff; the source has no halo to check)
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        int c = c0 + j;
        c = c < g.Cs ? c : g.Cs - 1;
        coff[j] = (unsigned)c * (unsigned)DHWs * 4u;
      }
    }
    // LDS slots (uint2 index inside one plane) of my stores
    const int a_slot = r * 32 + pc(r, 4 * q);        // 4 pieces = 32 contiguous bytes (the swizzle permutes whole 32-B
    const int a_half = (r >> 2) & 1;                 //   segments and swaps their 16-byte halves)
    const int a8_slot = r * AP + pca(r, 2 * q);      // MT == 4: my 8 dY rows = 2 pieces = one aligned 16-byte chunk
    const int x_slot = prow(r) * 4 + (q >> 1);       // extra image: piece q>>1, dword q&1

    float ra0[AR], rb0[16], rx0[2], ra1[AR], rb1[16], rx1[2];
    struct AffSet { int tab; float cap; };            // AFF: first table entry of the tile's group; the clamp's upper end
    AffSet as0 = {}, as1 = {};
    float sc_x = 1.f, sc_dy = 1.f;                    // NP == 2: operand scales
    if (NP == 2) {
      float inv_unused;
      f16_scale(__builtin_amdgcn_readfirstlane(*xcell), sc_x, inv_unused);
      f16_scale(__builtin_amdgcn_readfirstlane(*dycell), sc_dy, inv_unused);
    }

    auto issue_loads = [&](int i, float (&ra)[AR], float (&rb)[16], float (&rx)[2], AffSet& as) __attribute__((always_inline)) {
      const int n = (kt_begin + i) * 32 + r;
      const bool valid = n < n_end;
      const int nn = valid ? n : 0;
      const int b = nn / S, sp = nn - b * S;
      const int d = sp / HWo, rem = sp - d * HWo;
      const int h = rem / g.Wp, w = rem - h * g.Wp;
      const unsigned base_dy = valid ? ((unsigned)b * (unsigned)g.M * (unsigned)S + (unsigned)sp) * 4u : OOB;
      const int id = d * g.st - g.pt + dt, ih = h * g.sh - g.ph + dh, iw = w * g.sw - g.pw + dw;
      const bool okx = valid && (unsigned)id < (unsigned)g.Ds && (unsigned)ih < (unsigned)g.Hs && (unsigned)iw < (unsigned)g.Ws;
      const unsigned base_x = okx ? ((unsigned)b * (unsigned)g.Cs * (unsigned)DHWs + (unsigned)(id * HWs + ih * g.Ws + iw)) * 4u : OOB;
      if constexpr (AFF) {
        int grp = ((kt_begin + i) * 32) / aff_gpos;   // uniform; tiles past the end of the tensor read zeros whatever the group
        grp = grp < aff_groups ? grp : aff_groups - 1;
        as.tab = grp * 128 + 16 * q;
        as.cap = okx ? __builtin_inff() : 0.f;
      }
#pragma unroll
      for (int j = 0; j < AR; ++j)
        ra[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_dy, base_dy + moff[j], 0, 0));
      if (XTRA) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          rx[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_dy, base_dy + mxoff[j], 0, 0));
      }
      if constexpr (AFF) {
        // whole 16-channel groups (host condition): no clamped channel -- one per-lane offset, the channel stride rides in the
        // SCALAR offset (uniform), 15 registers and 15 additions per tile less
        const unsigned vx = base_x + coff[0];
#pragma unroll
        for (int j = 0; j < 16; ++j)
          rb[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, vx, (unsigned)j * ((unsigned)DHWs * 4u), 0));
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j)
          rb[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, base_x + coff[j], 0, 0));
      }
    };

    // img[p] = plane p of one operand image; sc = operand scale (NP == 2)
    auto store16 = [&](uint2 (&img)[NP][32 * 32], int slot, const float (&v)[16], float sc) __attribute__((always_inline)) {
      // 16 columns = 4 pieces per plane; the two 16-byte halves swap with the row's bit 2
      if constexpr (NP == 2) {
        uint4 ph[2], pl[2];
        unsigned hh, ll;
#define CSTP_SPLITH(J, DST, F) split2h(v[J], v[(J) + 1], sc, hh, ll); ph[DST].F = hh; pl[DST].F = ll;
        CSTP_SPLITH(0, 0, x) CSTP_SPLITH(2, 0, y) CSTP_SPLITH(4, 0, z) CSTP_SPLITH(6, 0, w)
        CSTP_SPLITH(8, 1, x) CSTP_SPLITH(10, 1, y) CSTP_SPLITH(12, 1, z) CSTP_SPLITH(14, 1, w)
#undef CSTP_SPLITH
        uint4* d0 = reinterpret_cast<uint4*>(img[0] + (slot & ~3));
        uint4* d1 = reinterpret_cast<uint4*>(img[1] + (slot & ~3));
        d0[a_half] = ph[0]; d0[a_half ^ 1] = ph[1];
        d1[a_half] = pl[0]; d1[a_half ^ 1] = pl[1];
      } else {
        uint4 ph[2], pm[2], pl[2];
        unsigned hh, mm, ll;
#define CSTP_SPLIT(J, DST, F) split2(v[J], v[(J) + 1], hh, mm, ll); ph[DST].F = hh; pm[DST].F = mm; pl[DST].F = ll;
        CSTP_SPLIT(0, 0, x) CSTP_SPLIT(2, 0, y) CSTP_SPLIT(4, 0, z) CSTP_SPLIT(6, 0, w)
        CSTP_SPLIT(8, 1, x) CSTP_SPLIT(10, 1, y) CSTP_SPLIT(12, 1, z) CSTP_SPLIT(14, 1, w)
#undef CSTP_SPLIT
        uint4* d0 = reinterpret_cast<uint4*>(img[0] + (slot & ~3));
        uint4* d1 = reinterpret_cast<uint4*>(img[1] + (slot & ~3));
        uint4* d2 = reinterpret_cast<uint4*>(img[NP - 1] + (slot & ~3));
        d0[a_half] = ph[0]; d0[a_half ^ 1] = ph[1];
        d1[a_half] = pm[0]; d1[a_half ^ 1] = pm[1];
        d2[a_half] = pl[0]; d2[a_half ^ 1] = pl[1];
      }
    };
    auto store8 = [&](uint2 (&img)[NP][32 * AP], int slot, const float* v, float sc) __attribute__((always_inline)) {
      if constexpr (NP == 2) {
        uint4 ph, pl;
        unsigned hh, ll;
        split2h(v[0], v[1], sc, hh, ll); ph.x = hh; pl.x = ll;
        split2h(v[2], v[3], sc, hh, ll); ph.y = hh; pl.y = ll;
        split2h(v[4], v[5], sc, hh, ll); ph.z = hh; pl.z = ll;
        split2h(v[6], v[7], sc, hh, ll); ph.w = hh; pl.w = ll;
        *reinterpret_cast<uint4*>(img[0] + (slot & ~1)) = ph;
        *reinterpret_cast<uint4*>(img[1] + (slot & ~1)) = pl;
      } else {
        uint4 ph, pm, pl;
        unsigned hh, mm, ll;
        split2(v[0], v[1], hh, mm, ll); ph.x = hh; pm.x = mm; pl.x = ll;
        split2(v[2], v[3], hh, mm, ll); ph.y = hh; pm.y = mm; pl.y = ll;
        split2(v[4], v[5], hh, mm, ll); ph.z = hh; pm.z = mm; pl.z = ll;
        split2(v[6], v[7], hh, mm, ll); ph.w = hh; pm.w = mm; pl.w = ll;
        *reinterpret_cast<uint4*>(img[0] + (slot & ~1)) = ph;
        *reinterpret_cast<uint4*>(img[1] + (slot & ~1)) = pm;
        *reinterpret_cast<uint4*>(img[NP - 1] + (slot & ~1)) = pl;
      }
    };
    auto split_store = [&](int buf, const float (&ra)[AR], float (&rb)[16], const float (&rx)[2], const AffSet& as) __attribute__((always_inline)) {
      if constexpr (AR == 16) store16(Am[buf], a_slot, reinterpret_cast<const float(&)[16]>(ra), sc_dy);
      else store8(Am[buf], a8_slot, ra, sc_dy);
      if constexpr (AFF) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const float lo = aff_relu ? 0.f : -as.cap;
        const f32x4* ta = reinterpret_cast<const f32x4*>(aff_a + as.tab);
        const f32x4* tb = reinterpret_cast<const f32x4*>(aff_b + as.tab);
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
          const f32x4 a4 = ta[j4], b4 = tb[j4];         // scales / shifts of channels 4 j4 .. 4 j4 + 3 (two lane groups, two addresses)
#pragma unroll
          for (int jj = 0; jj < 4; jj += 2) {
            const int j = 4 * j4 + jj;
            const f32x2 v2 = {rb[j], rb[j + 1]}, a2 = {a4[jj], a4[jj + 1]}, b2 = {b4[jj], b4[jj + 1]};
            const f32x2 z2 = __builtin_elementwise_fma(v2, a2, b2);
            rb[j] = __builtin_amdgcn_fmed3f(z2[0], lo, as.cap);
            rb[j + 1] = __builtin_amdgcn_fmed3f(z2[1], lo, as.cap);
          }
        }
      }
      store16(Bm[buf], a_slot, rb, sc_x);
      if (XTRA) {
        if constexpr (NP == 2) {
          unsigned hh, ll;
          split2h(rx[0], rx[1], sc_dy, hh, ll);
          reinterpret_cast<unsigned*>(&Ax[buf][0][x_slot])[q & 1] = hh;
          reinterpret_cast<unsigned*>(&Ax[buf][1][x_slot])[q & 1] = ll;
        } else {
          unsigned hh, mm, ll;
          split2(rx[0], rx[1], hh, mm, ll);
          reinterpret_cast<unsigned*>(&Ax[buf][0][x_slot])[q & 1] = hh;
          reinterpret_cast<unsigned*>(&Ax[buf][1][x_slot])[q & 1] = mm;
          reinterpret_cast<unsigned*>(&Ax[buf][NP - 1][x_slot])[q & 1] = ll;
        }
      }
    };

    // No load is conditional (tiles past the end of the split read zeros through OOB offsets): the compiler's counted
    // s_waitcnt keeps two K-tiles in flight (see igemm_k1s).
    issue_loads(0, ra0, rb0, rx0, as0);
    issue_loads(1, ra1, rb1, rx1, as1);
    split_store(0, ra0, rb0, rx0, as0);
    issue_loads(2, ra0, rb0, rx0, as0);
    __syncthreads();
    for (int i = 0; i < ntiles; i += 2) {
      split_store(1, ra1, rb1, rx1, as1);             // tile i+1
      issue_loads(i + 3, ra1, rb1, rx1, as1);
      __syncthreads();
      if (i + 1 >= ntiles) break;
      split_store(0, ra0, rb0, rx0, as0);             // tile i+2
      issue_loads(i + 4, ra0, rb0, rx0, as0);
      __syncthreads();
    }
    return;
  }

  // ================================================== consumers ==================================================
  const int wn = wave;                                // owns column tiles 2*wn, 2*wn+1 (j) of all MT row tiles (m)
  const int grp = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
  f32x4 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) { acc[i][0][rr] = 0.f; acc[i][1][rr] = 0.f; }
  // transposed-read slots: rows 8*grp + lq (+4), piece ct*4 + lp of column tile ct
  const int r_lo = 8 * grp + lq, r_hi = r_lo + 4;
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  auto tr_frag = [&](const uint2* img, int ct) __attribute__((always_inline)) -> s16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + r_lo * 32 + pc(r_lo, ct * 4 + lp)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + r_hi * 32 + pc(r_hi, ct * 4 + lp)));
    return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
  auto tr_frag_a = [&](const uint2* img, int ct) __attribute__((always_inline)) -> s16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + r_lo * AP + pca(r_lo, ct * 4 + lp)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + r_hi * AP + pca(r_hi, ct * 4 + lp)));
    return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
  auto tr_frag_x = [&](const uint2* img) __attribute__((always_inline)) -> s16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + prow(r_lo) * 4 + lp));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + prow(r_hi) * 4 + lp));
    return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };

  __syncthreads();
  int buf = 0;
  for (int i = 0; i < ntiles; ++i) {
    s16x8 bp[NP][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) bp[pl][c] = tr_frag(Bm[buf][pl], 2 * wn + c);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      s16x8 ap[NP];
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
        ap[pl] = (XTRA && mt == 8) ? tr_frag_x(Ax[XTRA ? buf : 0][pl]) : tr_frag_a(Am[buf][pl], mt);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        f32x4 a = acc[mt][c];
        if constexpr (NP == 2) {
#define CSTP_MM(P, Q) a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ap[P]), __builtin_bit_cast(f16x8, bp[Q][c]), a, 0, 0, 0)
          CSTP_MM(1, 0); CSTP_MM(0, 1); CSTP_MM(0, 0);
#undef CSTP_MM
        } else {
#define CSTP_MM(P, Q) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ap[P]), __builtin_bit_cast(bf16x8, bp[Q][c]), a, 0, 0, 0)
          CSTP_MM(NP - 1, 0); CSTP_MM(0, NP - 1); CSTP_MM(1, 1); CSTP_MM(1, 0); CSTP_MM(0, 1); CSTP_MM(0, 0);
#undef CSTP_MM
        }
        acc[mt][c] = a;
      }
    }
    __syncthreads();
    buf ^= 1;
  }

  // C layout: col (j) = lane & 15, row (m) = (lane >> 4) * 4 + reg
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int j = j0 + (2 * wn + c) * 16 + li;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int m = m0 + mt * 16 + grp * 4 + rr;
        if (m < g.M && j < Jtot) wgrad_out(&dwp[(size_t)split * det_stride + (size_t)m * Jp + j], acc[mt][c][rr], det_stride != 0);
      }
    }
}

}  // namespace cstp
