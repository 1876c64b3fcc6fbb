// The bf16-STORAGE path (BASELINE configs[4]: "3D-ResNet-50 backbone swap ... bf16"; models/BE/r3d_byol.py:100-206).
//
// Spec (the reference has no reduced-precision mode; this is what torch.autocast(bfloat16) does to its modules, stated so that
// the oracle can restate it -- oracle/r3d_byol_oracle.py `storage="bf16"`):
//   * every 5-D activation tensor in HBM is bf16: the clip (cast once), convolution outputs, BatchNorm(+residual)(+ReLU) outputs,
//     MaxPool3d output, and in the backward pass every gradient of such a tensor;
//   * all arithmetic between a load and a store is fp32 (fp64 inside the BatchNorm reductions); each stored value is the fp32
//     result rounded to nearest-even ONCE;
//   * convolution operands are the bf16 activations and the fp32 master weights rounded to bf16 at use
//     (v_mfma_f32_16x16x32_bf16: exact products, fp32 accumulation -- no operand split, ceiling 2 516 TFLOP/s);
//   * parameters, their gradients (accumulated in fp32 from the bf16 operands), BatchNorm statistics and running statistics,
//     the pooled features, predictor / heads / losses, optimizer: fp32 as on the fp32-storage path.
//
// Kernels (layout NCDHW, positions contiguous; one wave = 64 lanes):
//   conv_b16_kernel<MT, TAB>   implicit GEMM, forward AND data gradient: tile = 128 positions x 16*MT channels, K-tile 32,
//                              X gathered with 2-byte buffer loads (halo = out-of-range offset = 0), both operands through a
//                              swizzled LDS double buffer, one barrier per K-tile.  TAB: reduction index k = tap * C + c over a
//                              zero-padded copy with a per-k offset table (the 3-channel 7x7x7 stem).
//   wgrad_b16_kernel<TAB>      dW[m][k] += sum_p dY[m][p] * Xcol[k][p]: 64 x 64 tile, reduction over 64-position chunks,
//                              split over the positions; every split leaves an fp32 slab, b16_wgrad_reduce_kernel sums them in a
//                              fixed order into dw[m][c][tap] (deterministic; no atomics).
//   split-K                    layers with few positions (14x14 / 7x7 frames) split the K loop of conv_b16_kernel over blockIdx.y
//                              into fp32 slabs summed by b16_sum_slabs_kernel: 16 blocks x 432 K-tiles become 512 x 14.
//   BatchNorm / pooling / cast: HBM-streaming, 16-byte (8 x bf16) accesses where rows allow.
#include <stdlib.h>

#include "common.h"
#include "pack_b16.h"

namespace cstp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

__device__ __forceinline__ float bf2f(u16 v) { return __builtin_bit_cast(float, (unsigned)v << 16); }
__device__ __forceinline__ float bflo(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bfhi(unsigned v) { return __builtin_bit_cast(float, v & 0xffff0000u); }
// two fp32 -> packed bf16 pair (element 0 in the low half), round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pack_bf2(float a, float b) {
  f32x2v v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ u16 f2bf(float a) { return (u16)(pack_bf2(a, 0.f) & 0xffffu); }

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld16_last(const u16* p) {          // streaming (last-use) 16-byte load
  const u32x4v v = __builtin_nontemporal_load(reinterpret_cast<const u32x4v*>(p));
  return make_uint4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t b16_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

constexpr unsigned B16_OOB = 0x80000000u;      // host guarantees every gathered tensor is < 2 GiB
constexpr int B16_MAXTAPS = 27;
constexpr int B16_KTAB = 1056;                 // 7x7x7 taps x 3 channels = 1029, padded to 32

// ---- fp32 -> bf16 ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) cast_b16_kernel(const float* __restrict__ x, u16* __restrict__ y, size_t n) {
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    reinterpret_cast<uint2*>(y)[i] = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[n4 * 4 + threadIdx.x] = f2bf(x[n4 * 4 + threadIdx.x]);
}

// zero-padded copy xp[plane][D + 2pt][H + 2ph][W + 2pw] of x[plane][D][H][W] (both bf16): one wave per padded row
__global__ void __launch_bounds__(256)
pad_b16_kernel(const u16* __restrict__ x, u16* __restrict__ xp, int planes, int D, int H, int W, int pt, int ph, int pw) {
  const int Dq = D + 2 * pt, Hq = H + 2 * ph, Wq = W + 2 * pw;
  const int nrows = planes * Dq * Hq;
  const int lane = threadIdx.x & 63;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < nrows; row += gridDim.x * 4) {
    const int h = row % Hq, r2 = row / Hq;
    const int d = r2 % Dq, pl = r2 / Dq;
    const int id = d - pt, ih = h - ph;
    const bool in = (unsigned)id < (unsigned)D && (unsigned)ih < (unsigned)H;
    const u16* src = x + ((size_t)(pl * D + (in ? id : 0)) * H + (in ? ih : 0)) * W;
    u16* dst = xp + (size_t)row * Wq;
    for (int w = lane; w < Wq; w += 64) {
      const int iw = w - pw;
      dst[w] = (in && (unsigned)iw < (unsigned)W) ? src[iw] : (u16)0;
    }
  }
}

// ---- weights: fp32 [kout][cin][taps] -> bf16 GEMM operand rows ------------------------------------------------------------------
// forward:        wp[m = kout (Mp rows)][k = tap * cin + c  (Kw, zero beyond taps * cin)]
// data gradient:  wp[m = cin  (Mp rows)][k = tap * kout + ko]
__global__ void __launch_bounds__(256)
pack_w_b16_kernel(const float* __restrict__ w, u16* __restrict__ wp, int kout, int cin, int ntaps, int Mp, int Kw, int dgrad) {
  pack_w_b16_body(w, wp, kout, cin, ntaps, Mp, Kw, dgrad, (int)blockIdx.x, (int)gridDim.x);
}

// ---- implicit-GEMM convolution, forward and data gradient ---------------------------------------------------------------------
// out[nb][m][o(q)] = sum over the tap list and the source channels of  W[m][wtap][c] * src[nb][c][q * ss + off(tap)]
//   forward:        q = output position, ss = stride, off = tap - padding, o(q) = q
//   data gradient:  one launch per stride-parity class z of the INPUT positions: q enumerates the positions of the class
//                   (input coordinate = q * stride + z), src = dY, ss = 1, the tap list holds the taps with
//                   (z + pad - tap) % stride == 0, off = (z + pad - tap) / stride, o(q) = q * stride + z
struct B16Conv {
  int Nb, Cs, Ds, Hs, Ws;          // source tensor [Nb][Cs][Ds][Hs][Ws]
  int Dq, Hq, Wq;                  // enumerated positions per sample
  int sst, ssh, ssw;               // source coordinate = q * ss + off
  int M, Do, Ho, Wo;               // output tensor [Nb][M][Do][Ho][Wo]
  int ost, osh, osw, ozt, ozh, ozw;  // output coordinate = q * os + oz
  int ntaps;                       // entries of the tap list (grouped mode), or taps of the table mode
  int Kw;                          // elements per packed weight row
  int contig;                      // four consecutive q are four consecutive, 8-byte aligned outputs of one sample
  int n_tiles_x, n_tiles_m;
  int acc;                         // data gradient: out = bf16(bf16(result) + out) -- the sum of a residual connection's two gradients
                                   // formed here instead of by a separate add pass (ops.GradJoin); the rounding points are autograd's
  int ksplit;                      // > 1: blockIdx.y owns a slice of the K-tiles and leaves an fp32 partial tile in its slab
  int wf32;                        // PW forward: `wp` is the fp32 weight tensor [M][Kw] itself (k contiguous): rounded on the fly, no pack launch
  int kh, kw;                      // TAB: kernel extent (for the offset table)
  short off[B16_MAXTAPS][4];       // per list entry: source offset (t, h, w), weight tap index
};

// 16-byte chunk c (0..3) of the 64-byte LDS row `row` (32 k of one position / one channel) sits at c ^ ((row >> 2) & 3):
// the 16 rows x one chunk of a fragment read then cover all 64 banks once.
__device__ __forceinline__ int b16_slot(int row, int c) { return row * 4 + (c ^ ((row >> 2) & 3)); }

// PW ("pointwise" first, then every stride-1 layer whose rows are multiples of 8 positions): a thread gathers EIGHT consecutive
// positions of one channel with one 16-byte load -- for a tap that shifts the row by one element plus the neighbouring dword,
// the eight values then come out of two v_alignbit per dword pair, and an element beyond the row's end is an out-of-range
// load = 0; a tap row outside the frame is an out-of-range offset for the whole octet --
// (two per thread and K-tile instead of sixteen 2-byte loads: the generic loop is bound by the number of gather instructions),
// the LDS image is k-major (32 channel rows of 128 positions, 288-byte row stride) and the position fragments come out of it
// k-contiguous through the transposing ds_read_b64_tr_b16 (lane 4q + p of a 16-lane group supplies row q / positions
// 4p .. 4p + 3, lane i receives position i of the four rows).
constexpr int PW_ROW = 18;                           // uint4 per k-row of the PW image: 16 octets of positions + 32 bytes of padding

template <int MT, bool TAB, bool PW = false>
__global__ void __launch_bounds__(256)
conv_b16_kernel(const B16Conv g, const u16* __restrict__ src, const uint4* __restrict__ wp, u16* __restrict__ out,
                float* __restrict__ slab, size_t slab_stride) {
  static_assert(!(TAB && PW), "the pointwise mode has no offset table");
  constexpr int BM = 16 * MT;
  constexpr int NW = BM / 64;                       // 16-byte weight chunks per thread and K-tile
  __shared__ uint4 Xs[2][PW ? 32 * PW_ROW : 128 * 4];
  __shared__ uint4 Ws[2][BM * 4];
  __shared__ int taps[B16_MAXTAPS * 4];
  __shared__ unsigned ktab[TAB ? B16_KTAB : 4];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // XCD-aware order: the blocks of one XCD (bid & 7) walk the row tiles of one position tile back to back
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int mtile = slot % g.n_tiles_m;
  const int chunk = (g.n_tiles_x + 7) >> 3;
  const int nt_in = slot / g.n_tiles_m;
  const int ntile = xcd * chunk + nt_in;
  if (nt_in >= chunk || ntile >= g.n_tiles_x) return;
  const int npq = g.Dq * g.Hq * g.Wq;
  const int npos = g.Nb * npq;
  const int n0 = ntile * 128, m0 = mtile * BM;
  const int HWs = g.Hs * g.Ws, DHWs = g.Ds * HWs;

  if (!TAB) {
    if (t < g.ntaps * 4) taps[t] = g.off[t >> 2][t & 3];
  } else {
    const int khw = g.kh * g.kw, kreal = g.ntaps * g.Cs;
    for (int k = t; k < g.Kw; k += 256) {
      const int kk = k < kreal ? k : 0;               // the padding k's carry zero weights: any in-range address will do
      const int tp = kk / g.Cs, c = kk - tp * g.Cs;
      const int dt = tp / khw, rr = tp - dt * khw, dh = rr / g.kw, dw = rr - dh * g.kw;
      ktab[k] = (unsigned)((c * g.Ds + dt) * HWs + dh * g.Ws + dw) * 2u;
    }
  }
  __syncthreads();

  // ---- my part of the gather: position p, 16-k group g2 of every K-tile
  const int p = t & 127, g2 = t >> 7;
  const int P = n0 + p;
  const bool pvalid = P < npos;
  int qd, qh, qw, nb;
  {
    int n = pvalid ? P : 0;
    qw = n % g.Wq; n /= g.Wq;
    qh = n % g.Hq; n /= g.Hq;
    qd = n % g.Dq; nb = n / g.Dq;
  }
  const int cd0 = qd * g.sst, ch0 = qh * g.ssh, cw0 = qw * g.ssw;
  const unsigned nb_off = (unsigned)((size_t)nb * g.Cs * DHWs);          // elements
  const unsigned cstride = (unsigned)DHWs * 2u;                         // bytes between channels
  const __amdgpu_buffer_rsrc_t rs_src = b16_rsrc(src, (unsigned)((size_t)g.Nb * g.Cs * DHWs * 2));
  const int gpt = (TAB || PW) ? 1 : (g.Cs >> 4);                        // 16-channel groups per tap
  const int ngroups = (TAB || PW) ? (g.Kw >> 4) : g.ntaps * gpt;
  const int cblocks = (g.Cs + 31) >> 5;                                 // PW: 32-channel blocks per tap
  const int ntiles_all = PW ? g.ntaps * cblocks : (ngroups + 1) >> 1;
  // split-K: layers with few positions (a 7x7 frame: 4 position tiles) would leave most of the chip idle and every block
  // latency-bound on a long K loop -- blockIdx.y takes K-tiles [kt_lo, kt_hi) and the partial tiles are summed by
  // b16_sum_slabs_kernel (deterministic: no atomics)
  const int tps = (ntiles_all + g.ksplit - 1) / g.ksplit;
  const int kt_lo = blockIdx.y * tps;
  const int kt_hi = (kt_lo + tps) < ntiles_all ? (kt_lo + tps) : ntiles_all;
  int ti = (2 * kt_lo + g2) / gpt, cg = (2 * kt_lo + g2) - ti * gpt;    // (tap entry, channel group) of my group in my first K-tile
  const unsigned tab_base = (unsigned)(((size_t)nb * g.Cs * g.Ds + cd0) * HWs + ch0 * g.Ws + cw0) * 2u;   // TAB: bytes

  // ---- my part of the weight tile: row wrow (+64), chunk wc of the K-tile (group wc >> 1, half wc & 1)
  const int wrow = t >> 2, wc = t & 3;
  const char* wbase = reinterpret_cast<const char*>(wp) + (size_t)(m0 + wrow) * g.Kw * 2;
  const size_t wrow64 = (size_t)64 * g.Kw * 2;

  u16 xr[16];
  u32x4v xq[PW ? 2 : 1];
  unsigned xe[2] = {0, 0};                             // PW: the neighbouring dword of an octet under a +-1 tap
  uint4 wr[NW];
  // PW: my channel row of the K-tile and my two octets of positions (lanes 0..7 of a row: 128 contiguous bytes)
  const int krow = t >> 3;
  unsigned pw_off[2] = {B16_OOB, B16_OOB};
  int pw_d[2] = {0, 0}, pw_h[2] = {0, 0}, pw_w[2] = {0, 0};
  int last_dw = 0;
  if (PW) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int Po = n0 + 8 * ((t & 7) + 8 * e);
      if (Po < npos) {
        int n = Po;
        pw_w[e] = n % g.Wq; n /= g.Wq;
        pw_h[e] = n % g.Hq; n /= g.Hq;
        pw_d[e] = n % g.Dq; const int nbo = n / g.Dq;
        // (source coordinate = position + tap offset: the source may be larger than the output by the padding it lacks)
        pw_off[e] = (unsigned)((((size_t)nbo * g.Cs * g.Ds + pw_d[e]) * HWs + pw_h[e] * g.Ws + pw_w[e]) * 2);
      }
    }
  }

  auto issue = [&](int kt) __attribute__((always_inline)) {
    // X
    unsigned vo = B16_OOB;
    if (PW) {
      const int pti = kt / cblocks, cb = kt - pti * cblocks;
      const int dt = taps[pti * 4 + 0], dh = taps[pti * 4 + 1], dw = taps[pti * 4 + 2];
      const int c = cb * 32 + krow;
      const unsigned coff = (unsigned)c * cstride + (unsigned)((dt * HWs + dh * g.Ws) * 2);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const bool ok = kt < kt_hi && c < g.Cs && pw_off[e] != B16_OOB && (unsigned)(pw_d[e] + dt) < (unsigned)g.Ds &&
                        (unsigned)(pw_h[e] + dh) < (unsigned)g.Hs;
        const unsigned v = ok ? pw_off[e] + coff : B16_OOB;
        xq[e] = __builtin_bit_cast(u32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs_src, v, 0, 0));
        if (dw != 0) {                                 // (block-uniform)
          unsigned ve = B16_OOB;
          if (ok) {
            if (dw < 0) { if (pw_w[e] > 0) ve = v - 4u; }
            else if (pw_w[e] + 8 < g.Ws) ve = v + 16u;
          }
          xe[e] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs_src, ve, 0, 0);
        }
      }
      last_dw = dw;
    } else if (TAB) {
      const int k0 = (kt * 2 + g2) * 16;
      if (pvalid && k0 < g.Kw) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
          xr[j] = (u16)__builtin_amdgcn_raw_buffer_load_b16(rs_src, tab_base + ktab[k0 + j], 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) xr[j] = 0;
      }
    } else {
      const int gi = kt * 2 + g2;
      if (pvalid && gi < ngroups) {
        const int cd = cd0 + taps[ti * 4 + 0], chh = ch0 + taps[ti * 4 + 1], cw = cw0 + taps[ti * 4 + 2];
        if ((unsigned)cd < (unsigned)g.Ds && (unsigned)chh < (unsigned)g.Hs && (unsigned)cw < (unsigned)g.Ws)
          vo = (nb_off + (unsigned)(cg * 16) * (unsigned)DHWs + (unsigned)(cd * HWs + chh * g.Ws + cw)) * 2u;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j)
        xr[j] = (u16)__builtin_amdgcn_raw_buffer_load_b16(rs_src, vo, (unsigned)j * cstride, 0);
    }
    // W
    int kel;                                                            // element offset of my chunk inside the packed row
    bool wok;
    if (PW) {
      const int pti = kt / cblocks, cb = kt - pti * cblocks;
      wok = kt < kt_hi && cb * 32 + wc * 8 < g.Cs;
      kel = taps[(wok ? pti : 0) * 4 + 3] * g.Cs + cb * 32 + wc * 8;
    } else if (TAB) {
      kel = kt * 32 + wc * 8;
      wok = kel < g.Kw;
    } else {
      const int gi = kt * 2 + (wc >> 1);
      wok = gi < ngroups;
      const int wti = wok ? gi / gpt : 0, wcg = gi - wti * gpt;
      kel = taps[wti * 4 + 3] * g.Cs + wcg * 16 + (wc & 1) * 8;
    }
    if (PW && g.wf32) {
      // the 1x1x1 forward weight [kout][cin] already IS the row-major GEMM operand: eight floats -> eight bf16 here
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        wr[j] = make_uint4(0, 0, 0, 0);
        if (wok && m0 + wrow + 64 * j < g.M) {
          const float4* pf = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(wp) + (size_t)(m0 + wrow + 64 * j) * g.Kw + kel);
          const float4 f0 = pf[0], f1 = pf[1];
          wr[j] = make_uint4(pack_bf2(f0.x, f0.y), pack_bf2(f0.z, f0.w), pack_bf2(f1.x, f1.y), pack_bf2(f1.z, f1.w));
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        wr[j] = make_uint4(0, 0, 0, 0);
        if (wok) wr[j] = *reinterpret_cast<const uint4*>(wbase + j * wrow64 + (size_t)kel * 2);
      }
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {
    if (!TAB && !PW) {
      cg += 2;
      while (cg >= gpt) { cg -= gpt; ++ti; }
    }
  };
  auto stage = [&](int buf) __attribute__((always_inline)) {
    if constexpr (PW) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        u32x4v q = xq[e];
        if (last_dw < 0)           // element i of the octet = source element i - 1: the previous dword's high half comes in front
          q = u32x4v{__builtin_amdgcn_alignbit(q[0], xe[e], 16), __builtin_amdgcn_alignbit(q[1], q[0], 16),
                     __builtin_amdgcn_alignbit(q[2], q[1], 16), __builtin_amdgcn_alignbit(q[3], q[2], 16)};
        else if (last_dw > 0)      // ... = source element i + 1: the next dword's low half closes the octet
          q = u32x4v{__builtin_amdgcn_alignbit(q[1], q[0], 16), __builtin_amdgcn_alignbit(q[2], q[1], 16),
                     __builtin_amdgcn_alignbit(q[3], q[2], 16), __builtin_amdgcn_alignbit(xe[e], q[3], 16)};
        Xs[buf][krow * PW_ROW + (t & 7) + 8 * e] = make_uint4(q[0], q[1], q[2], q[3]);
      }
    } else {
      uint4 a, b;
      a.x = xr[0] | ((unsigned)xr[1] << 16); a.y = xr[2] | ((unsigned)xr[3] << 16);
      a.z = xr[4] | ((unsigned)xr[5] << 16); a.w = xr[6] | ((unsigned)xr[7] << 16);
      b.x = xr[8] | ((unsigned)xr[9] << 16); b.y = xr[10] | ((unsigned)xr[11] << 16);
      b.z = xr[12] | ((unsigned)xr[13] << 16); b.w = xr[14] | ((unsigned)xr[15] << 16);
      Xs[buf][b16_slot(p, g2 * 2)] = a;
      Xs[buf][b16_slot(p, g2 * 2 + 1)] = b;
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) Ws[buf][b16_slot(wrow + 64 * j, wc)] = wr[j];
  };

  f32x4 acc[2][MT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(kt_lo);
  advance();
  stage(0);
  __syncthreads();
  const int frow = lane & 15, fchunk = lane >> 4;
  for (int kt = kt_lo; kt < kt_hi; ++kt) {
    const int buf = (kt - kt_lo) & 1;
    const bool more = kt + 1 < kt_hi;
    if (more) { issue(kt + 1); advance(); }
    bf16x8 xa[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      if constexpr (PW) {
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const int lq = frow >> 2, lp = frow & 3, ct = wave * 2 + a;
        const uint2* img = reinterpret_cast<const uint2*>(&Xs[buf][0]);
        const int r_lo = 8 * fchunk + lq;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + r_lo * (2 * PW_ROW) + ct * 4 + lp));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + (r_lo + 4) * (2 * PW_ROW) + ct * 4 + lp));
        xa[a] = __builtin_bit_cast(bf16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
      } else {
        const int row = wave * 32 + a * 16 + frow;
        xa[a] = __builtin_bit_cast(bf16x8, Xs[buf][b16_slot(row, fchunk)]);
      }
    }
#pragma unroll
    for (int b = 0; b < MT; ++b) {
      const int row = b * 16 + frow;
      const bf16x8 wb = __builtin_bit_cast(bf16x8, Ws[buf][b16_slot(row, fchunk)]);
#pragma unroll
      for (int a = 0; a < 2; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[a], wb, acc[a][b], 0, 0, 0);
    }
    if (more) stage(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds positions 4 * (lane >> 4) + r (r = 0..3) of position tile a, channel (lane & 15) of row tile b
  const int DHWo = g.Do * g.Ho * g.Wo, HWo = g.Ho * g.Wo;
  float* sl = g.ksplit > 1 ? slab + (size_t)blockIdx.y * slab_stride : nullptr;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int P0 = n0 + wave * 32 + a * 16 + 4 * (lane >> 4);
    if (g.contig && sl == nullptr) {
      // bf16 results, 8 bytes per lane.  As the accumulator tile stands, lane = 16 * (position quad) + channel: the 16 lanes of a
      // store pass would write 16 pieces of 8 bytes, one per channel row.  The packed values travel to lane 4 * channel + quad
      // first (two ds_bpermute per tile), so that a pass writes four runs of 32 contiguous bytes (igemm_k1p's KP_EPI_PERM).
      const int sq = lane & 3, sfr = lane >> 2;
      const int perm_src = (16 * sq + sfr) * 4;
      const int Ps = n0 + wave * 32 + a * 16 + 4 * sq;
      const bool pok = Ps < npos;
      const int nbo = (pok ? Ps : 0) / npq, pos = (pok ? Ps : 0) - nbo * npq;
#pragma unroll
      for (int b = 0; b < MT; ++b) {
        struct F4 { float a, b, c, d; };
        const F4 t4 = __builtin_bit_cast(F4, acc[a][b]);      // (plain struct: ext_vector component reads have miscompiled, DESIGN)
        const int lo = __builtin_amdgcn_ds_bpermute(perm_src, (int)pack_bf2(t4.a, t4.b));
        const int hi = __builtin_amdgcn_ds_bpermute(perm_src, (int)pack_bf2(t4.c, t4.d));
        const int m = m0 + b * 16 + sfr;
        if (pok && m < g.M) {
          uint2* o2 = reinterpret_cast<uint2*>(out + ((size_t)nbo * g.M + m) * DHWo + pos);
          unsigned l2 = (unsigned)lo, h2 = (unsigned)hi;
          if (g.acc) {
            const uint2 old = *o2;
            l2 = pack_bf2(bflo(l2) + bflo(old.x), bfhi(l2) + bfhi(old.x));
            h2 = pack_bf2(bflo(h2) + bflo(old.y), bfhi(h2) + bfhi(old.y));
          }
          *o2 = make_uint2(l2, h2);
        }
      }
    } else if (g.contig) {
      if (P0 >= npos) continue;
      const int nbo = P0 / npq, pos = P0 - nbo * npq;
#pragma unroll
      for (int b = 0; b < MT; ++b) {
        const int m = m0 + b * 16 + (lane & 15);
        if (m >= g.M) continue;
        const f32x4 v = acc[a][b];
        *reinterpret_cast<f32x4*>(sl + ((size_t)nbo * g.M + m) * DHWo + pos) = v;
      }
    } else {
      size_t ooff[4];
      bool ok[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int n = P0 + r;
        ok[r] = n < npos;
        n = ok[r] ? n : 0;
        const int w_ = n % g.Wq; n /= g.Wq;
        const int h_ = n % g.Hq; n /= g.Hq;
        const int d_ = n % g.Dq; const int nbo = n / g.Dq;
        ooff[r] = (size_t)nbo * g.M * DHWo + (size_t)(d_ * g.ost + g.ozt) * HWo + (h_ * g.osh + g.ozh) * g.Wo + (w_ * g.osw + g.ozw);
      }
#pragma unroll
      for (int b = 0; b < MT; ++b) {
        const int m = m0 + b * 16 + (lane & 15);
        if (m >= g.M) continue;
        const f32x4 v = acc[a][b];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (!ok[r]) continue;
          if (sl != nullptr) sl[ooff[r] + (size_t)m * DHWo] = v[r];
          else {
            u16* o1 = out + ooff[r] + (size_t)m * DHWo;
            *o1 = g.acc ? f2bf(bf2f(f2bf(v[r])) + bf2f(*o1)) : f2bf(v[r]);
          }
        }
      }
    }
  }
}

// out[i] = bf16(sum over the S slabs of slab[s][i]) (fixed order: bit-reproducible)
__global__ void __launch_bounds__(256)
b16_sum_slabs_kernel(const float* __restrict__ slab, int S, size_t stride, u16* __restrict__ out, size_t n, int acc = 0) {
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 a = *reinterpret_cast<const f32x4*>(slab + 4 * i);
    for (int s = 1; s < S; ++s) a += *reinterpret_cast<const f32x4*>(slab + (size_t)s * stride + 4 * i);
    unsigned lo = pack_bf2(a[0], a[1]), hi = pack_bf2(a[2], a[3]);
    if (acc) {                                        // (as the convolution kernel's epilogue: bf16(bf16(result) + out))
      const uint2 old = reinterpret_cast<const uint2*>(out)[i];
      lo = pack_bf2(bflo(lo) + bflo(old.x), bfhi(lo) + bfhi(old.x));
      hi = pack_bf2(bflo(hi) + bflo(old.y), bfhi(hi) + bfhi(old.y));
    }
    reinterpret_cast<uint2*>(out)[i] = make_uint2(lo, hi);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    float a = slab[i];
    for (int s = 1; s < S; ++s) a += slab[(size_t)s * stride + i];
    out[i] = acc ? f2bf(bf2f(f2bf(a)) + bf2f(out[i])) : f2bf(a);
  }
}

// ---- weight gradient ------------------------------------------------------------------------------------------------------------
struct B16Wgrad {
  int Nb, Cs, Ds, Hs, Ws;          // x (or its zero-padded copy, TAB) [Nb][Cs][Ds][Hs][Ws]
  int M, Do, Ho, Wo;               // dY [Nb][M][Do][Ho][Wo]
  int st, sh, sw;                  // source coordinate = o * s + off(tap)
  int ntaps, K;                    // columns: k = tap * Cs + c, K = ntaps * Cs
  int vec8;                        // Do*Ho*Wo % 8 == 0: eight consecutive positions of a dY row are one aligned 16-byte load
  int nchunks, chunks_per_split;   // 64-position chunks
  int kh, kw;
  int Kp;                          // slab row length (K rounded up to 64)
  short off[B16_MAXTAPS][4];
};

constexpr int WG_ROW = 9;          // uint4 per LDS row: 64 positions x 2 B + 16 B padding (fragment reads conflict-free)

// PW (the 1x1x1 stride-1 layers, frames a multiple of 8 positions): column k is channel k and its positions are contiguous -- the X
// rows are fetched like the dY rows, two 16-byte loads per thread and chunk instead of sixteen 2-byte gathers.
template <bool TAB, bool PW = false>
__global__ void __launch_bounds__(256)
wgrad_b16_kernel(const B16Wgrad g, const u16* __restrict__ x, const u16* __restrict__ dy, float* __restrict__ slab) {
  static_assert(!(TAB && PW), "the pointwise mode has no offset table");
  __shared__ uint4 Ys[2][64 * WG_ROW];
  __shared__ uint4 Xs[2][64 * WG_ROW];
  __shared__ int taps[B16_MAXTAPS * 4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int k0 = blockIdx.x * 64, m0 = blockIdx.y * 64;
  const int c_lo = blockIdx.z * g.chunks_per_split;
  int c_hi = c_lo + g.chunks_per_split;
  c_hi = c_hi < g.nchunks ? c_hi : g.nchunks;      // (the host sizes the grid so that every split owns at least one chunk)
  if (!TAB && t < g.ntaps * 4) taps[t] = g.off[t >> 2][t & 3];
  __syncthreads();
  const int npo = g.Do * g.Ho * g.Wo, npos = g.Nb * npo;
  const int HWs = g.Hs * g.Ws, DHWs = g.Ds * HWs;
  const __amdgpu_buffer_rsrc_t rs_x = b16_rsrc(x, (unsigned)((size_t)g.Nb * g.Cs * DHWs * 2));
  const unsigned cstride = (unsigned)DHWs * 2u;

  // X: thread (p, kq) gathers the 16 columns k0 + 16 kq + j of position p
  const int p = t & 63, kq = t >> 6;
  const int gpt = TAB ? 1 : (g.Cs >> 4);
  const int gi = (k0 >> 4) + kq;
  const bool gvalid = TAB ? true : gi < g.ntaps * gpt;
  const int ti = gvalid ? gi / gpt : 0, c0 = (gi - ti * gpt) * 16;
  int toff[3] = {0, 0, 0};
  if (!TAB) { toff[0] = taps[ti * 4]; toff[1] = taps[ti * 4 + 1]; toff[2] = taps[ti * 4 + 2]; }
  unsigned ktb[16];
  if (TAB) {
    const int khw = g.kh * g.kw;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      int k = k0 + kq * 16 + j;
      k = k < g.K ? k : 0;
      const int tp = k / g.Cs, c = k - tp * g.Cs;
      const int dt = tp / khw, rr = tp - dt * khw, dh = rr / g.kw, dwv = rr - dh * g.kw;
      ktb[j] = (unsigned)((c * g.Ds + dt) * HWs + dh * g.Ws + dwv) * 2u;
    }
  }
  // dY: thread (row, two octets)
  const int yrow = t >> 2, yo = (t & 3) * 2;
  const bool mvalid = m0 + yrow < g.M;

  u16 xr[16];
  uint4 yr[2], xq[2];
  const bool cvalid = k0 + yrow < g.K;                 // PW: my column (= channel) exists
  auto issue = [&](int ci) __attribute__((always_inline)) {
    const int P0 = ci * 64;
    if constexpr (PW) {
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const int Pq = P0 + (yo + o) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (cvalid && Pq < npos) {
          const int nb = Pq / npo, pos = Pq - nb * npo;
          v = *reinterpret_cast<const uint4*>(x + ((size_t)nb * g.Cs + k0 + yrow) * npo + pos);
        }
        xq[o] = v;
      }
    } else {
      int n = P0 + p;
      const bool pv = n < npos && gvalid;
      n = pv ? n : 0;
      const int ow = n % g.Wo; n /= g.Wo;
      const int oh = n % g.Ho; n /= g.Ho;
      const int od = n % g.Do; const int nb = n / g.Do;
      if (TAB) {
        const unsigned base = (unsigned)(((size_t)nb * g.Cs * g.Ds + od * g.st) * HWs + oh * g.sh * g.Ws + ow * g.sw) * 2u;
#pragma unroll
        for (int j = 0; j < 16; ++j)
          xr[j] = pv ? (u16)__builtin_amdgcn_raw_buffer_load_b16(rs_x, base + ktb[j], 0, 0) : (u16)0;
      } else {
        unsigned vo = B16_OOB;
        const int cd = od * g.st + toff[0], chh = oh * g.sh + toff[1], cw = ow * g.sw + toff[2];
        if (pv && (unsigned)cd < (unsigned)g.Ds && (unsigned)chh < (unsigned)g.Hs && (unsigned)cw < (unsigned)g.Ws)
          vo = (unsigned)(((size_t)nb * g.Cs + c0) * DHWs + cd * HWs + chh * g.Ws + cw) * 2u;
#pragma unroll
        for (int j = 0; j < 16; ++j) xr[j] = (u16)__builtin_amdgcn_raw_buffer_load_b16(rs_x, vo, (unsigned)j * cstride, 0);
      }
    }
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int Pq = P0 + (yo + o) * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (mvalid && Pq < npos) {
        const int nb = Pq / npo, pos = Pq - nb * npo;
        const u16* rowp = dy + ((size_t)nb * g.M + m0 + yrow) * npo + pos;
        if (g.vec8) {
          v = *reinterpret_cast<const uint4*>(rowp);
        } else {
          unsigned e[8];
          int nbb = nb, pp = pos;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            e[j] = (Pq + j < npos) ? dy[((size_t)nbb * g.M + m0 + yrow) * npo + pp] : 0;
            if (++pp == npo) { pp = 0; ++nbb; }
          }
          v = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
        }
      }
      yr[o] = v;
    }
  };
  auto stage = [&](int buf) __attribute__((always_inline)) {
    if constexpr (PW) {
      Xs[buf][yrow * WG_ROW + yo] = xq[0];
      Xs[buf][yrow * WG_ROW + yo + 1] = xq[1];
    } else {
      u16* xs = reinterpret_cast<u16*>(&Xs[buf][0]);
#pragma unroll
      for (int j = 0; j < 16; ++j) xs[(kq * 16 + j) * (WG_ROW * 8) + p] = xr[j];
    }
    Ys[buf][yrow * WG_ROW + yo] = yr[0];
    Ys[buf][yrow * WG_ROW + yo + 1] = yr[1];
  };

  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int mt0 = (wave >> 1) * 2, nt0 = (wave & 1) * 2;
  const int frow = lane & 15, fchunk = lane >> 4;

  issue(c_lo);
  stage(0);
  __syncthreads();
  for (int ci = c_lo; ci < c_hi; ++ci) {
    const int buf = (ci - c_lo) & 1;
    const bool more = ci + 1 < c_hi;
    if (more) issue(ci + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ya[2], xb[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) ya[a] = __builtin_bit_cast(bf16x8, Ys[buf][((mt0 + a) * 16 + frow) * WG_ROW + ks * 4 + fchunk]);
#pragma unroll
      for (int b = 0; b < 2; ++b) xb[b] = __builtin_bit_cast(bf16x8, Xs[buf][((nt0 + b) * 16 + frow) * WG_ROW + ks * 4 + fchunk]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ya[a], xb[b], acc[a][b], 0, 0, 0);
    }
    if (more) stage(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds rows m = 4 * (lane >> 4) + r of row tile a, column (lane & 15) of column tile b: the partial tile
  //      goes to this split's slab [M rounded to 64][Kp] in GEMM layout (64-byte runs; the first version added straight into
  //      dw[m][c][tap] with fp32 atomics: 64 cache lines per instruction at the memory side -- most of the kernel's time)
  float* sl = slab + (size_t)blockIdx.z * ((size_t)gridDim.y * 64 * g.Kp);
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int col = k0 + (nt0 + b) * 16 + (lane & 15);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + (mt0 + a) * 16 + 4 * (lane >> 4) + r;
        sl[(size_t)m * g.Kp + col] = acc[a][b][r];
      }
    }
  }
}

// dw[m][c][tap] (+)= sum over the S slabs of slab[s][m][k = tap * C + c]   (fixed order: bit-reproducible).
// Threads walk dw in ITS order (coalesced read-modify-write); the slab reads of a wave fall on `ntaps` rows of 64-byte lines that
// the neighbouring waves share (the k-ordered version scattered 4-byte writes 4 * ntaps bytes apart: 2.8 ms per R3D-50 step).
__global__ void __launch_bounds__(256)
b16_wgrad_reduce_kernel(const float* __restrict__ slab, int S, size_t stride, float* __restrict__ dw, int M, int Cs, int ntaps, int K,
                        int Kp, int accumulate) {
  const size_t total = (size_t)M * K;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int tap = (int)(i % ntaps);
    const size_t r = i / ntaps;
    const int c = (int)(r % Cs), m = (int)(r / Cs);
    const float* p = slab + (size_t)m * Kp + tap * Cs + c;
    float a = p[0];
    for (int s = 1; s < S; ++s) a += p[(size_t)s * stride];
    dw[i] = accumulate ? dw[i] + a : a;
  }
}

// ... the same for MANY slabs over a small gradient (a 64 x 64 1x1x1 layer of the first stage: 349 position splits): 16 outputs x
// 16 split lanes per block, lane s sums slabs s, s + 16, ... and the 16 partial sums are added in lane order (fixed: reproducible)
__global__ void __launch_bounds__(256)
b16_wgrad_reduce_wide_kernel(const float* __restrict__ slab, int S, size_t stride, float* __restrict__ dw, int M, int Cs, int ntaps,
                             int K, int Kp, int accumulate) {
  __shared__ float red[16][17];
  const int il = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const size_t total = (size_t)M * K;
  const size_t i = (size_t)blockIdx.x * 16 + il;
  float a = 0.f;
  if (i < total) {
    const int tap = (int)(i % ntaps);
    const size_t r = i / ntaps;
    const int c = (int)(r % Cs), m = (int)(r / Cs);
    const float* p = slab + (size_t)m * Kp + tap * Cs + c;
    for (int s = sl; s < S; s += 16) a += p[(size_t)s * stride];
  }
  red[sl][il] = a;
  __syncthreads();
  if (sl == 0 && i < total) {
    float t = red[0][il];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += red[q][il];
    dw[i] = accumulate ? dw[i] + t : t;
  }
}

// ---- train-mode BatchNorm (+ residual) (+ ReLU) on bf16 tensors: fp64 statistics, fp32 apply ------------------------------------
static inline int b16_bn_nsplit(int n, int c) {
  int ns = cdiv(2048, c);
  if (ns > n) ns = n;
  return ns < 1 ? 1 : ns;
}

// MODE 0: (sum x, sum x^2)      MODE 1: (sum g, sum g * xhat), g = dy * mask
template <int MODE, bool VEC8>
__global__ void __launch_bounds__(256)
b16_bn_reduce_kernel(const u16* __restrict__ x, const u16* __restrict__ y, const u16* __restrict__ dy, const float* __restrict__ mean,
                     const float* __restrict__ invstd, double* __restrict__ part, int npg, int c, int s, int nsplit, int relu,
                     const float2* __restrict__ ss) {
  __shared__ double sm[16];
  const int ch = blockIdx.x, grp = blockIdx.y / nsplit, j = blockIdx.y - grp * nsplit;
  double a0 = 0.0, a1 = 0.0;
  float mu = 0.f, is = 0.f, sc = 0.f, sh = 0.f;
  if (MODE == 1) { mu = mean[grp * c + ch]; is = invstd[grp * c + ch]; }
  const bool remask = (MODE == 1) && relu && (y == nullptr);
  if (remask) { const float2 t2 = ss[grp * c + ch]; sc = t2.x; sh = t2.y; }
  auto one = [&](float v, float gq, float o) __attribute__((always_inline)) {
    if (MODE == 0) {
      a0 += (double)v; a1 += (double)v * v;
    } else {
      if (remask) { if (!(__builtin_fmaf(v, sc, sh) > 0.f)) gq = 0.f; }
      else if (relu && !(o > 0.f)) gq = 0.f;
      a0 += (double)gq; a1 += (double)(gq * ((v - mu) * is));
    }
  };
  for (int rr = j; rr < npg; rr += nsplit) {
    const size_t base = ((size_t)(grp * npg + rr) * c + ch) * s;
    if (VEC8) {
      const uint4* xp = reinterpret_cast<const uint4*>(x + base);
      const uint4* gp = reinterpret_cast<const uint4*>(dy + base);
      const uint4* yp = reinterpret_cast<const uint4*>(y + base);
      for (int i = threadIdx.x; i < (s >> 3); i += 256) {
        const uint4 v = xp[i];
        uint4 gv = make_uint4(0, 0, 0, 0), ov = make_uint4(0, 0, 0, 0);
        if (MODE == 1) { gv = gp[i]; if (relu && !remask) ov = yp[i]; }
        const unsigned vv[4] = {v.x, v.y, v.z, v.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w}, oo[4] = {ov.x, ov.y, ov.z, ov.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { one(bflo(vv[e]), bflo(gg[e]), bflo(oo[e])); one(bfhi(vv[e]), bfhi(gg[e]), bfhi(oo[e])); }
      }
    } else {
      for (int i = threadIdx.x; i < s; i += 256)
        one(bf2f(x[base + i]), MODE == 1 ? bf2f(dy[base + i]) : 0.f, (MODE == 1 && relu && !remask) ? bf2f(y[base + i]) : 0.f);
    }
  }
  a0 = block_sum(a0, sm);
  a1 = block_sum(a1, sm);
  if (threadIdx.x == 0) {
    part[((size_t)ch * gridDim.y + blockIdx.y) * 2 + 0] = a0;
    part[((size_t)ch * gridDim.y + blockIdx.y) * 2 + 1] = a1;
  }
}

// Sums of one (channel, group) from its `nsplit` partials: every lane of wave 0 calls it and gets the same two doubles (lanes
// stride over the partials, xor-shuffle tree: one fixed order, so every block of the channel derives the same bits).
__device__ __forceinline__ void b16_fold_partials(const double* __restrict__ part, int ch, int groups, int g, int nsplit, double& s0,
                                                 double& s1) {
  const int lane = threadIdx.x & 63;
  const double* p = part + ((size_t)ch * groups + g) * nsplit * 2;
  s0 = 0.0; s1 = 0.0;
  for (int j = lane; j < nsplit; j += 64) { s0 += p[2 * j]; s1 += p[2 * j + 1]; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); }
}

// (mean, invstd, scale, shift) of one (channel, group) from the forward sums
__device__ __forceinline__ void b16_bn_stats_of(double s0, double s1, double count, float eps, float gamma, float beta, double& mu,
                                               double& var, float& isf, float& sc, float& sh) {
  mu = s0 / count;
  var = s1 / count - mu * mu;
  if (var < 0.0) var = 0.0;
  isf = (float)(1.0 / sqrt(var + (double)eps));
  sc = isf * gamma;
  sh = beta - (float)mu * sc;
}

constexpr int B16_UNROLL = 4;     // 16-byte vectors per thread per block

// y = bf16(act(fma(x, scale, shift) + residual)); one block = one chunk of one (sample, channel) row.  The statistics' finalize
// is folded in (it was a launch of its own between the reduction and this pass: 106 launches per 3D-ResNet-50 step): wave 0 of
// every block folds the partial sums of ITS (channel, group); the block of the channel's first row and chunk also writes
// save_mean / save_invstd / scale_shift for every group and moves the running statistics, group after group.
template <bool VEC8>
__global__ void __launch_bounds__(256)
b16_bn_apply_fwd_kernel(const u16* __restrict__ x, const u16* __restrict__ res, u16* __restrict__ y, const double* __restrict__ part,
                        int nsplit, int groups, double count, float eps, float momentum, const float* __restrict__ gamma,
                        const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
                        float* __restrict__ save_mean, float* __restrict__ save_invstd, float2* __restrict__ ss, int c, int s, int npg,
                        int relu, int chunks) {
  constexpr int W = VEC8 ? 8 : 1;
  __shared__ float s_ss[2];
  const int row = blockIdx.x / chunks, chunk = blockIdx.x - row * chunks;
  const int ch = row % c, grp = (row / c) / npg;
  // the block's x values first: in flight while wave 0 folds the statistics
  uint4 xv[B16_UNROLL];
  if (VEC8) {
#pragma unroll
    for (int u = 0; u < B16_UNROLL; ++u) {
      const int e = (chunk * B16_UNROLL * 256 + u * 256 + threadIdx.x) * W;
      xv[u] = make_uint4(0, 0, 0, 0);
      if (e < s) xv[u] = ld16_last(x + (size_t)row * s + e);
    }
  }
  if (threadIdx.x < 64) {
    const float ga = gamma[ch], be = beta[ch];
    double s0, s1, mu, var;
    float isf, scv, shv;
    b16_fold_partials(part, ch, groups, grp, nsplit, s0, s1);
    b16_bn_stats_of(s0, s1, count, eps, ga, be, mu, var, isf, scv, shv);
    if (threadIdx.x == 0) { s_ss[0] = scv; s_ss[1] = shv; }
    if (row == ch && chunk == 0) {                    // first sample of group 0: the channel's bookkeeping, once
      float rm = 0.f, rv = 0.f;
      if (running_mean != nullptr) { rm = running_mean[ch]; rv = running_var[ch]; }
      for (int g = 0; g < groups; ++g) {
        b16_fold_partials(part, ch, groups, g, nsplit, s0, s1);
        b16_bn_stats_of(s0, s1, count, eps, ga, be, mu, var, isf, scv, shv);
        if (threadIdx.x == 0) {
          save_mean[g * c + ch] = (float)mu;
          save_invstd[g * c + ch] = isf;
          ss[g * c + ch] = make_float2(scv, shv);
        }
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        rm = (float)((1.0 - momentum) * rm + momentum * mu);      // group after group, like successive calls
        rv = (float)((1.0 - momentum) * rv + momentum * unb);
      }
      if (threadIdx.x == 0 && running_mean != nullptr) { running_mean[ch] = rm; running_var[ch] = rv; }
    }
  }
  __syncthreads();
  const float sc = s_ss[0], sh = s_ss[1];
  const size_t base = (size_t)row * s;
#pragma unroll
  for (int u = 0; u < B16_UNROLL; ++u) {
    const int e = (chunk * B16_UNROLL * 256 + u * 256 + threadIdx.x) * W;
    if (e >= s) continue;
    if (VEC8) {
      const uint4 v = xv[u];
      uint4 r = make_uint4(0, 0, 0, 0);
      if (res != nullptr) r = *reinterpret_cast<const uint4*>(res + base + e);
      const unsigned vv[4] = {v.x, v.y, v.z, v.w}, rr[4] = {r.x, r.y, r.z, r.w};
      unsigned o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float a = __builtin_fmaf(bflo(vv[q]), sc, sh) + bflo(rr[q]), b = __builtin_fmaf(bfhi(vv[q]), sc, sh) + bfhi(rr[q]);
        if (relu) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
        o[q] = pack_bf2(a, b);
      }
      *reinterpret_cast<uint4*>(y + base + e) = make_uint4(o[0], o[1], o[2], o[3]);
    } else {
      float a = __builtin_fmaf(bf2f(x[base + e]), sc, sh);
      if (res != nullptr) a += bf2f(res[base + e]);
      if (relu) a = fmaxf(a, 0.f);
      y[base + e] = f2bf(a);
    }
  }
}

// eval mode (model.eval(): validation / test of the fine-tuned network): the running statistics are the statistics
template <bool VEC8>
__global__ void __launch_bounds__(256)
b16_bn_eval_kernel(const u16* __restrict__ x, const u16* __restrict__ res, u16* __restrict__ y, const float* __restrict__ gamma,
                   const float* __restrict__ beta, const float* __restrict__ running_mean, const float* __restrict__ running_var,
                   int c, int s, float eps, int relu, int chunks) {
  constexpr int W = VEC8 ? 8 : 1;
  const int row = blockIdx.x / chunks, chunk = blockIdx.x - row * chunks;
  const int ch = row % c;
  const float sc = gamma[ch] / sqrtf(running_var[ch] + eps);
  const float sh = beta[ch] - running_mean[ch] * sc;
  const size_t base = (size_t)row * s;
#pragma unroll
  for (int u = 0; u < B16_UNROLL; ++u) {
    const int e = (chunk * B16_UNROLL * 256 + u * 256 + threadIdx.x) * W;
    if (e >= s) continue;
    if (VEC8) {
      const uint4 v = ld16_last(x + base + e);
      uint4 r = make_uint4(0, 0, 0, 0);
      if (res != nullptr) r = *reinterpret_cast<const uint4*>(res + base + e);
      const unsigned vv[4] = {v.x, v.y, v.z, v.w}, rr[4] = {r.x, r.y, r.z, r.w};
      unsigned o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float a = __builtin_fmaf(bflo(vv[q]), sc, sh) + bflo(rr[q]), b = __builtin_fmaf(bfhi(vv[q]), sc, sh) + bfhi(rr[q]);
        if (relu) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
        o[q] = pack_bf2(a, b);
      }
      *reinterpret_cast<uint4*>(y + base + e) = make_uint4(o[0], o[1], o[2], o[3]);
    } else {
      float a = __builtin_fmaf(bf2f(x[base + e]), sc, sh);
      if (res != nullptr) a += bf2f(res[base + e]);
      if (relu) a = fmaxf(a, 0.f);
      y[base + e] = f2bf(a);
    }
  }
}

// dx = bf16(gamma * invstd * (g - mean(g) - xhat * mean(g * xhat))), dres = bf16(g), g = dy * mask.  The backward finalize is
// folded in as in the forward pass: wave 0 folds the (sum g, sum g * xhat) partials of the block's (channel, group); the block
// of the channel's first row and chunk writes dgamma / dbeta (summed over the groups; accumulate: +=).
template <bool VEC8>
__global__ void __launch_bounds__(256)
b16_bn_apply_bwd_kernel(const u16* __restrict__ x, const u16* __restrict__ y, const u16* __restrict__ dy, const float* __restrict__ gamma,
                        const float* __restrict__ mean, const float* __restrict__ invstd, const double* __restrict__ part, int nsplit,
                        int groups, float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate, u16* __restrict__ dx,
                        u16* __restrict__ dres, int c, int s, int npg, float inv_count, int relu, const float2* __restrict__ ss,
                        int chunks) {
  constexpr int W = VEC8 ? 8 : 1;
  __shared__ float s_g[2];
  const bool remask = relu && (y == nullptr);
  const int row = blockIdx.x / chunks, chunk = blockIdx.x - row * chunks;
  const int ch = row % c, grp = (row / c) / npg, gc = grp * c + ch;
  // the block's x and dy values first: in flight while wave 0 folds the partial sums
  uint4 xv[B16_UNROLL], gvv[B16_UNROLL];
  if (VEC8) {
#pragma unroll
    for (int u = 0; u < B16_UNROLL; ++u) {
      const int e = (chunk * B16_UNROLL * 256 + u * 256 + threadIdx.x) * W;
      xv[u] = gvv[u] = make_uint4(0, 0, 0, 0);
      if (e < s) { xv[u] = ld16_last(x + (size_t)row * s + e); gvv[u] = ld16_last(dy + (size_t)row * s + e); }
    }
  }
  if (threadIdx.x < 64) {
    double s0, s1;
    b16_fold_partials(part, ch, groups, grp, nsplit, s0, s1);
    if (threadIdx.x == 0) { s_g[0] = (float)s0; s_g[1] = (float)s1; }
    if (row == ch && chunk == 0) {
      double t0 = 0.0, t1 = 0.0;
      for (int g = 0; g < groups; ++g) {
        b16_fold_partials(part, ch, groups, g, nsplit, s0, s1);
        t0 += s0; t1 += s1;
      }
      if (threadIdx.x == 0) {       // the affine parameters are shared by the groups
        dbeta[ch] = (accumulate ? dbeta[ch] : 0.f) + (float)t0;
        dgamma[ch] = (accumulate ? dgamma[ch] : 0.f) + (float)t1;
      }
    }
  }
  __syncthreads();
  const float mu = mean[gc], is = invstd[gc];
  const float k = gamma[ch] * is;
  const float mb = s_g[0] * inv_count, mg = s_g[1] * inv_count;
  float sc = 0.f, sh = 0.f;
  if (remask) { const float2 t2 = ss[gc]; sc = t2.x; sh = t2.y; }
  const size_t base = (size_t)row * s;
  auto one = [&](float v, float gq, float o, float& gout) __attribute__((always_inline)) -> float {
    if (remask) { if (!(__builtin_fmaf(v, sc, sh) > 0.f)) gq = 0.f; }
    else if (relu && !(o > 0.f)) gq = 0.f;
    gout = gq;
    return k * (gq - mb - (v - mu) * is * mg);
  };
#pragma unroll
  for (int u = 0; u < B16_UNROLL; ++u) {
    const int e = (chunk * B16_UNROLL * 256 + u * 256 + threadIdx.x) * W;
    if (e >= s) continue;
    if (VEC8) {
      const uint4 v = xv[u];
      const uint4 gv = gvv[u];
      uint4 ov = make_uint4(0, 0, 0, 0);
      if (relu && !remask) ov = *reinterpret_cast<const uint4*>(y + base + e);
      const unsigned vv[4] = {v.x, v.y, v.z, v.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w}, oo[4] = {ov.x, ov.y, ov.z, ov.w};
      unsigned o[4], gr[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float g0, g1;
        const float a = one(bflo(vv[q]), bflo(gg[q]), bflo(oo[q]), g0), b = one(bfhi(vv[q]), bfhi(gg[q]), bfhi(oo[q]), g1);
        o[q] = pack_bf2(a, b);
        gr[q] = pack_bf2(g0, g1);
      }
      *reinterpret_cast<uint4*>(dx + base + e) = make_uint4(o[0], o[1], o[2], o[3]);
      if (dres != nullptr) *reinterpret_cast<uint4*>(dres + base + e) = make_uint4(gr[0], gr[1], gr[2], gr[3]);
    } else {
      float g0;
      const float a = one(bf2f(x[base + e]), bf2f(dy[base + e]), (relu && !remask) ? bf2f(y[base + e]) : 0.f, g0);
      dx[base + e] = f2bf(a);
      if (dres != nullptr) dres[base + e] = f2bf(g0);
    }
  }
}

// ---- SMALL tensors (as bn.hip's bn_small_*_kernel: <= 16 values per thread and group on blocks of 256 or 1024 threads): ONE
// launch per pass -- a block owns a channel, group after group holds the group's values in registers, sums them in fp64, does the
// channel's bookkeeping (b16_bn_stats_of, the running statistics group after group), applies and writes.  Half of the ~160
// BatchNorm calls of a 3D-ResNet-50 step at configs[4]'s share are such tensors (28 x 28 and smaller), each two launches before.
constexpr int B16_SMALL_PT = 16;

__device__ __forceinline__ unsigned b16_small_off(int e, int s, int g, int npg, int c, int ch) {
  const int r = e / s, i = e - r * s;
  return ((unsigned)(g * npg + r) * c + ch) * s + i;       // (host: the tensor has < 2^31 elements)
}

template <int TPB>
__global__ void __launch_bounds__(TPB)
b16_bn_small_fwd_kernel(const u16* __restrict__ x, const u16* __restrict__ res, u16* __restrict__ y, const float* __restrict__ gamma,
                        const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
                        float* __restrict__ save_mean, float* __restrict__ save_invstd, float2* __restrict__ ss, int c, int s, int npg,
                        int groups, float eps, float momentum, int relu) {
  __shared__ double sm[16];
  __shared__ float s_ss[2];
  const int ch = blockIdx.x, E = npg * s;
  const double count = (double)E;
  const float ga = gamma[ch], be = beta[ch];
  float rm = 0.f, rv = 0.f;
  if (threadIdx.x == 0 && running_mean != nullptr) { rm = running_mean[ch]; rv = running_var[ch]; }
  for (int g = 0; g < groups; ++g) {
    float v[B16_SMALL_PT];
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int u = 0; u < B16_SMALL_PT; ++u) {
      const int e = u * TPB + threadIdx.x;
      v[u] = e < E ? bf2f(x[b16_small_off(e, s, g, npg, c, ch)]) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < B16_SMALL_PT; ++u) { a0 += (double)v[u]; a1 += (double)v[u] * v[u]; }
    a0 = block_sum(a0, sm);
    a1 = block_sum(a1, sm);
    if (threadIdx.x == 0) {
      double mu, var;
      float isf, scv, shv;
      b16_bn_stats_of(a0, a1, count, eps, ga, be, mu, var, isf, scv, shv);
      save_mean[g * c + ch] = (float)mu;
      save_invstd[g * c + ch] = isf;
      ss[g * c + ch] = make_float2(scv, shv);
      s_ss[0] = scv; s_ss[1] = shv;
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      rm = (float)((1.0 - momentum) * rm + momentum * mu);      // group after group, like successive calls
      rv = (float)((1.0 - momentum) * rv + momentum * unb);
    }
    __syncthreads();
    const float sc = s_ss[0], sh = s_ss[1];
#pragma unroll
    for (int u = 0; u < B16_SMALL_PT; ++u) {
      const int e = u * TPB + threadIdx.x;
      if (e >= E) continue;
      const unsigned off = b16_small_off(e, s, g, npg, c, ch);
      float a = __builtin_fmaf(v[u], sc, sh);
      if (res != nullptr) a += bf2f(res[off]);
      if (relu) a = fmaxf(a, 0.f);
      y[off] = f2bf(a);
    }
    __syncthreads();                                  // s_ss is rewritten by the next group
  }
  if (threadIdx.x == 0 && running_mean != nullptr) { running_mean[ch] = rm; running_var[ch] = rv; }
}

__global__ void __launch_bounds__(256)
b16_bn_small_bwd_kernel(const u16* __restrict__ x, const u16* __restrict__ y, const u16* __restrict__ dy, const float* __restrict__ gamma,
                        const float* __restrict__ mean, const float* __restrict__ invstd, const float2* __restrict__ ss,
                        u16* __restrict__ dx, u16* __restrict__ dres, float* __restrict__ dgamma, float* __restrict__ dbeta, int c, int s,
                        int npg, int groups, int relu, int accumulate) {
  __shared__ double sm[16];
  __shared__ float s_g[2];
  const int ch = blockIdx.x, E = npg * s;
  const float inv_count = (float)(1.0 / (double)E);
  const bool remask = relu && (y == nullptr);
  const float ga = gamma[ch];
  double t0 = 0.0, t1 = 0.0;
  for (int g = 0; g < groups; ++g) {
    const int gc = g * c + ch;
    const float mu = mean[gc], is = invstd[gc];
    float sc = 0.f, sh = 0.f;
    if (remask) { const float2 t2 = ss[gc]; sc = t2.x; sh = t2.y; }
    float v[B16_SMALL_PT], gr[B16_SMALL_PT];
#pragma unroll
    for (int u = 0; u < B16_SMALL_PT; ++u) {
      const int e = u * 256 + threadIdx.x;
      v[u] = 0.f; gr[u] = 0.f;
      if (e < E) {
        const unsigned off = b16_small_off(e, s, g, npg, c, ch);
        v[u] = bf2f(x[off]);
        float gq = bf2f(dy[off]);
        if (remask) { if (!(__builtin_fmaf(v[u], sc, sh) > 0.f)) gq = 0.f; }
        else if (relu && !(bf2f(y[off]) > 0.f)) gq = 0.f;
        gr[u] = gq;
      }
    }
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int u = 0; u < B16_SMALL_PT; ++u) { a0 += (double)gr[u]; a1 += (double)(gr[u] * ((v[u] - mu) * is)); }
    a0 = block_sum(a0, sm);
    a1 = block_sum(a1, sm);
    if (threadIdx.x == 0) { s_g[0] = (float)a0; s_g[1] = (float)a1; t0 += a0; t1 += a1; }
    __syncthreads();
    const float mb = s_g[0] * inv_count, mg = s_g[1] * inv_count;
    const float k = ga * is;
#pragma unroll
    for (int u = 0; u < B16_SMALL_PT; ++u) {
      const int e = u * 256 + threadIdx.x;
      if (e >= E) continue;
      const unsigned off = b16_small_off(e, s, g, npg, c, ch);
      dx[off] = f2bf(k * (gr[u] - mb - (v[u] - mu) * is * mg));
      if (dres != nullptr) dres[off] = f2bf(gr[u]);
    }
    __syncthreads();                                  // s_g is rewritten by the next group
  }
  if (threadIdx.x == 0) {
    dbeta[ch] = (accumulate ? dbeta[ch] : 0.f) + (float)t0;
    dgamma[ch] = (accumulate ? dgamma[ch] : 0.f) + (float)t1;
  }
}

// ---- pooling ------------------------------------------------------------------------------------------------------------------
// MaxPool3d (models/BE/r3d_byol.py:158): as maxpool3d_fwd/bwd_kernel of misc.hip on bf16 values (comparisons are exact)
__global__ void b16_maxpool3d_fwd_kernel(const u16* __restrict__ x, u16* __restrict__ y, int32_t* __restrict__ idx, int rows, int D, int H,
                                         int W, int Do, int Ho, int Wo, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph,
                                         int pw) {
  const size_t total = (size_t)rows * Do * Ho * Wo;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho); r /= Ho;
    const int dd = (int)(r % Do); const size_t row = r / Do;
    const u16* xp = x + row * (size_t)D * H * W;
    float best = -INFINITY;
    u16 bb = 0xff80;     // -inf
    int bi = -1;
    for (int a = 0; a < kd; ++a) {
      const int id = dd * sd - pd + a;
      if ((unsigned)id >= (unsigned)D) continue;
      for (int b = 0; b < kh; ++b) {
        const int ih = ho * sh - ph + b;
        if ((unsigned)ih >= (unsigned)H) continue;
        for (int c = 0; c < kw; ++c) {
          const int iw = wo * sw - pw + c;
          if ((unsigned)iw >= (unsigned)W) continue;
          const int fi = (id * H + ih) * W + iw;
          const u16 raw = xp[fi];
          const float v = bf2f(raw);
          if (v > best || v != v || bi < 0) { best = v; bb = raw; bi = fi; }
        }
      }
    }
    y[i] = bb;
    idx[i] = bi;
  }
}

__global__ void b16_maxpool3d_bwd_kernel(const u16* __restrict__ dy, const int32_t* __restrict__ idx, u16* __restrict__ dx, int rows, int D,
                                         int H, int W, int Do, int Ho, int Wo, int kd, int kh, int kw, int sd, int sh, int sw, int pd,
                                         int ph, int pw) {
  const size_t total = (size_t)rows * D * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H); r /= H;
    const int d = (int)(r % D); const size_t row = r / D;
    const int fi = (d * H + h) * W + w;
    const int d_lo = (d + pd - kd + sd) > 0 ? (d + pd - kd + sd) / sd : 0, d_hi = (d + pd) / sd < Do - 1 ? (d + pd) / sd : Do - 1;
    const int h_lo = (h + ph - kh + sh) > 0 ? (h + ph - kh + sh) / sh : 0, h_hi = (h + ph) / sh < Ho - 1 ? (h + ph) / sh : Ho - 1;
    const int w_lo = (w + pw - kw + sw) > 0 ? (w + pw - kw + sw) / sw : 0, w_hi = (w + pw) / sw < Wo - 1 ? (w + pw) / sw : Wo - 1;
    const size_t obase = row * (size_t)Do * Ho * Wo;
    float acc = 0.f;
    for (int a = d_lo; a <= d_hi; ++a)
      for (int b = h_lo; b <= h_hi; ++b)
        for (int c = w_lo; c <= w_hi; ++c) {
          const size_t o = obase + ((size_t)a * Ho + b) * Wo + c;
          if (idx[o] == fi) acc += bf2f(dy[o]);
        }
    dx[i] = f2bf(acc);
  }
}

// AdaptiveAvgPool3d(1): bf16 rows -> fp32 means (one wave per row); backward fp32 dy -> bf16 dx
__global__ void b16_avgpool_fwd_kernel(const u16* __restrict__ x, float* __restrict__ y, int rows, int s) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const u16* p = x + (size_t)row * s;
  float a = 0.f;
  for (int i = lane; i < s; i += 64) a += bf2f(p[i]);
  a = wave_sum(a);
  if (lane == 0) y[row] = a / (float)s;
}

__global__ void b16_avgpool_bwd_kernel(const float* __restrict__ dy, u16* __restrict__ dx, int rows, int s) {
  const size_t total = (size_t)rows * s;
  const float inv = 1.f / (float)s;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) dx[i] = f2bf(dy[i / s] * inv);
}

}  // namespace cstp

using namespace cstp;

// CSTP_BN_SMALL=0: every BatchNorm through the two-launch sequence (A/B switch shared with bn.hip; read once)
static bool b16_bn_small_on() {
  static const bool on = [] { const char* e = getenv("CSTP_BN_SMALL"); return !(e && e[0] == '0'); }();
  return on;
}

static inline unsigned b16_grid(size_t n, int per_block) {
  size_t b = (n + per_block - 1) / per_block;
  if (b > 65536) b = 65536;
  return (unsigned)(b < 1 ? 1 : b);
}

struct B16Geom {
  int Do, Ho, Wo, ntaps;
  bool tab;       // reduction over a zero-padded copy with an offset table: channel counts that are not multiples of 16
  int Kw, Dp, Hp, Wp;
};

static bool b16_geom(const cstp_conv_desc* d, B16Geom& q) {
  if (d == nullptr || d->n <= 0 || d->c <= 0 || d->k <= 0 || d->kt <= 0 || d->kh <= 0 || d->kw <= 0 || d->st <= 0 || d->sh <= 0 ||
      d->sw <= 0 || d->pt < 0 || d->ph < 0 || d->pw < 0)
    return false;
  q.Do = (d->d + 2 * d->pt - d->kt) / d->st + 1;
  q.Ho = (d->h + 2 * d->ph - d->kh) / d->sh + 1;
  q.Wo = (d->w + 2 * d->pw - d->kw) / d->sw + 1;
  if (q.Do <= 0 || q.Ho <= 0 || q.Wo <= 0) return false;
  q.ntaps = d->kt * d->kh * d->kw;
  q.tab = (d->c % 16) != 0;
  q.Dp = d->d + 2 * d->pt; q.Hp = d->h + 2 * d->ph; q.Wp = d->w + 2 * d->pw;
  q.Kw = q.tab ? (int)align_up((size_t)q.ntaps * d->c, 32) : q.ntaps * d->c;
  return true;
}

static size_t b16_pad_bytes(const cstp_conv_desc* d, const B16Geom& q) {
  return q.tab ? align_up((size_t)d->n * d->c * q.Dp * q.Hp * q.Wp * 2, 256) : 0;
}

// split-K factor of a forward / data-gradient launch sequence: a function of the geometry alone (so that the workspace
// query and the launches agree).  blocks0 = blocks without splitting, ntiles = K-tiles of the longest tap list.
constexpr size_t B16_SLAB_CAP = (size_t)128 << 20;
static int b16_conv_split(int blocks0, int ntiles, size_t out_elems) {
  if (blocks0 >= 768 || ntiles < 8) return 1;
  int S = 768 / blocks0;
  if (S > ntiles / 4) S = ntiles / 4;
  const size_t cap = B16_SLAB_CAP / (out_elems * sizeof(float));
  if ((size_t)S > cap) S = (int)cap;
  return S < 2 ? 1 : S;
}
static int b16_fwd_split(const cstp_conv_desc* d, const B16Geom& q) {
  const int BM = d->k > 64 ? 128 : 64;
  const int blocks0 = cdiv(d->n * q.Do * q.Ho * q.Wo, 128) * cdiv(d->k, BM);
  const int ngroups = q.tab ? q.Kw / 16 : q.ntaps * (d->c / 16);
  return b16_conv_split(blocks0, (ngroups + 1) / 2, (size_t)d->n * d->k * q.Do * q.Ho * q.Wo);
}
static int b16_dgrad_split(const cstp_conv_desc* d, const B16Geom& q) {
  if ((d->k % 16) != 0) return 1;
  const int BM = d->c > 64 ? 128 : 64;
  // the launches of the stride-parity classes run back to back: together they have about the blocks of one un-strided launch
  const int blocks0 = cdiv(d->n * d->d * d->h * d->w, 128) * cdiv(d->c, BM);
  const int taps_class = cdiv(d->kt, d->st) * cdiv(d->kh, d->sh) * cdiv(d->kw, d->sw);
  return b16_conv_split(blocks0, (taps_class * (d->k / 16) + 1) / 2, (size_t)d->n * d->c * d->d * d->h * d->w);
}
// weight gradient: positions split over S blocks per tile, each leaving an fp32 slab [M rounded to 64][K rounded to 64]
static int b16_wgrad_split(const cstp_conv_desc* d, const B16Geom& q) {
  const int K = q.ntaps * d->c;
  const int tiles = cdiv(K, 64) * cdiv(d->k, 64);
  const int nchunks = cdiv(d->n * q.Do * q.Ho * q.Wo, 64);
  int S = cdiv(1536, tiles);
  if (S > nchunks) S = nchunks;
  const size_t slab = (size_t)cdiv(d->k, 64) * 64 * cdiv(K, 64) * 64 * sizeof(float);
  const size_t cap = B16_SLAB_CAP / slab;
  if ((size_t)S > cap) S = (int)cap;
  if (S < 1) S = 1;
  const int cps = cdiv(nchunks, S);
  return cdiv(nchunks, cps);                 // every split owns at least one chunk
}

static size_t b16_wpack_bytes(const cstp_conv_desc* d, const B16Geom& q) {
  const size_t mp_f = align_up((size_t)d->k, 128), mp_d = align_up((size_t)d->c, 128);
  const size_t wf = align_up(mp_f * q.Kw * 2, 256), wd = align_up(mp_d * (size_t)q.ntaps * d->k * 2, 256);
  return wf > wd ? wf : wd;
}

extern "C" size_t cstp_b16_conv3d_workspace_bytes(const cstp_conv_desc* desc) {
  B16Geom q;
  if (!b16_geom(desc, q)) return 0;
  const int sf = b16_fwd_split(desc, q), sd = b16_dgrad_split(desc, q), sw = b16_wgrad_split(desc, q);
  size_t slabs = 0;
  if (sf > 1) slabs = (size_t)sf * desc->n * desc->k * q.Do * q.Ho * q.Wo * sizeof(float);
  if (sd > 1) { const size_t b = (size_t)sd * desc->n * desc->c * desc->d * desc->h * desc->w * sizeof(float); slabs = b > slabs ? b : slabs; }
  { const size_t b = (size_t)sw * cdiv(desc->k, 64) * 64 * cdiv(q.ntaps * desc->c, 64) * 64 * sizeof(float); slabs = b > slabs ? b : slabs; }
  return b16_wpack_bytes(desc, q) + b16_pad_bytes(desc, q) + align_up(slabs, 256) + 256;
}

extern "C" int cstp_b16_cast(void* stream, const float* x, uint16_t* y, size_t n) {
  CSTP_REQUIRE(x && y && n > 0, "bad argument");
  CSTP_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0, "cast: unaligned tensor");
  hipLaunchKernelGGL(cast_b16_kernel, dim3(b16_grid(n / 4 + 1, 256 * 4)), dim3(256), 0, as_stream(stream), x, y, n);
  CSTP_LAUNCH_CHECK();
  return 0;
}

template <bool TAB, bool PW = false>
static void b16_launch_conv(hipStream_t st, const B16Conv& g, int M, const uint16_t* src, const void* wp, uint16_t* out, float* slab,
                            size_t slab_stride) {
  const int chunk = (g.n_tiles_x + 7) / 8;
  const dim3 grid((unsigned)(8 * chunk * g.n_tiles_m), (unsigned)g.ksplit);
  if (M > 64)
    hipLaunchKernelGGL((conv_b16_kernel<8, TAB, PW>), grid, dim3(256), 0, st, g, src, reinterpret_cast<const uint4*>(wp), out, slab, slab_stride);
  else
    hipLaunchKernelGGL((conv_b16_kernel<4, TAB, PW>), grid, dim3(256), 0, st, g, src, reinterpret_cast<const uint4*>(wp), out, slab, slab_stride);
}

// the octet-gather path (conv_b16_kernel<.., PW>): stride 1, channel counts multiples of 16, 16-byte aligned source, and either
// 1x1x1 without padding on frames of a multiple of 8 positions (positions are then one linear run per channel), or taps whose
// column offsets are -1 / 0 / +1 on rows that are multiples of 8 positions in BOTH the source (ws) and the enumerated tensor (wq)
static bool b16_pointwise(const cstp_conv_desc* d, const void* src) {
  return d->kt == 1 && d->kh == 1 && d->kw == 1 && d->st == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 && d->ph == 0 && d->pw == 0 &&
         ((d->d * d->h * d->w) % 8) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (d->c % 16) == 0 && (d->k % 16) == 0;
}
static bool b16_octets(const cstp_conv_desc* d, const void* src, int ws, int wq) {
  if (b16_pointwise(d, src)) return true;
  return d->st == 1 && d->sh == 1 && d->sw == 1 && d->kw <= 3 && d->pw <= 1 && d->kw - 1 - d->pw <= 1 &&
         d->kt * d->kh * d->kw <= B16_MAXTAPS && (ws % 8) == 0 && (wq % 8) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 &&
         (d->c % 16) == 0 && (d->k % 16) == 0;
}

static float* b16_slabs(void* ws, const cstp_conv_desc* d, const B16Geom& q) {
  return reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + b16_wpack_bytes(d, q) + b16_pad_bytes(d, q));
}

extern "C" int cstp_b16_conv3d_forward(void* stream, const cstp_conv_desc* d, const uint16_t* x, const float* w, uint16_t* y, void* ws,
                                       size_t ws_bytes) {
  B16Geom q;
  CSTP_REQUIRE(b16_geom(d, q), "bad convolution geometry");
  CSTP_REQUIRE(x && w && y && ws && ws_bytes >= cstp_b16_conv3d_workspace_bytes(d), "null argument or workspace too small");
  CSTP_REQUIRE(q.tab || q.ntaps <= B16_MAXTAPS, "bf16 path: at most 27 filter taps for channel counts that are multiples of 16");
  CSTP_REQUIRE(!q.tab || q.Kw <= B16_KTAB, "bf16 path: reduction too long for the offset table");
  CSTP_REQUIRE((size_t)d->n * d->c * q.Dp * q.Hp * q.Wp * 2 < (1ull << 31), "bf16 path: gathered tensor must be < 2 GiB");
  CSTP_REQUIRE((reinterpret_cast<uintptr_t>(y) & 7) == 0, "unaligned output");
  hipStream_t st = as_stream(stream);
  const int BM = d->k > 64 ? 128 : 64;
  const int Mp = (int)align_up((size_t)d->k, BM);
  u16* wp = reinterpret_cast<u16*>(ws);
  const bool pw = !q.tab && b16_pointwise(d, x) && (reinterpret_cast<uintptr_t>(w) & 15) == 0;
  if (!pw && !pack_skip(wp)) {      // (pointwise forward: the kernel rounds the fp32 rows itself; pack plan: replayed by the caller)
    const unsigned nb = b16_grid((size_t)Mp * q.Kw, 256);
    pack_record_b16(w, wp, (int)nb, d->k, d->c, q.ntaps, Mp, q.Kw, 0);
    hipLaunchKernelGGL(pack_w_b16_kernel, dim3(nb), dim3(256), 0, st, w, wp, d->k, d->c, q.ntaps, Mp, q.Kw, 0);
    CSTP_LAUNCH_CHECK();
  }
  B16Conv g;
  memset(&g, 0, sizeof(g));
  g.Nb = d->n; g.Cs = d->c;
  g.Dq = q.Do; g.Hq = q.Ho; g.Wq = q.Wo;
  g.sst = d->st; g.ssh = d->sh; g.ssw = d->sw;
  g.M = d->k; g.Do = q.Do; g.Ho = q.Ho; g.Wo = q.Wo;
  g.ost = g.osh = g.osw = 1;
  g.Kw = q.Kw;
  g.contig = ((q.Do * q.Ho * q.Wo) % 4) == 0;
  g.n_tiles_x = cdiv(d->n * q.Do * q.Ho * q.Wo, 128);
  g.n_tiles_m = Mp / BM;
  g.kh = d->kh; g.kw = d->kw;
  g.ntaps = q.ntaps;
  g.ksplit = b16_fwd_split(d, q);
  float* slab = b16_slabs(ws, d, q);
  const size_t out_elems = (size_t)d->n * d->k * q.Do * q.Ho * q.Wo;
  if (q.tab) {
    u16* xp = reinterpret_cast<u16*>(reinterpret_cast<char*>(ws) + b16_wpack_bytes(d, q));
    hipLaunchKernelGGL(pad_b16_kernel, dim3(b16_grid((size_t)d->n * d->c * q.Dp * q.Hp, 4)), dim3(256), 0, st, x, xp, d->n * d->c, d->d,
                       d->h, d->w, d->pt, d->ph, d->pw);
    CSTP_LAUNCH_CHECK();
    g.Ds = q.Dp; g.Hs = q.Hp; g.Ws = q.Wp;
    b16_launch_conv<true>(st, g, d->k, xp, wp, y, slab, out_elems);
  } else {
    g.Ds = d->d; g.Hs = d->h; g.Ws = d->w;
    int i = 0;
    for (int a = 0; a < d->kt; ++a)
      for (int b = 0; b < d->kh; ++b)
        for (int c = 0; c < d->kw; ++c, ++i) {
          g.off[i][0] = (short)(a - d->pt); g.off[i][1] = (short)(b - d->ph); g.off[i][2] = (short)(c - d->pw); g.off[i][3] = (short)i;
        }
    if (pw) { g.wf32 = 1; b16_launch_conv<false, true>(st, g, d->k, x, w, y, slab, out_elems); }
    else if (b16_octets(d, x, d->w, q.Wo)) b16_launch_conv<false, true>(st, g, d->k, x, wp, y, slab, out_elems);
    else b16_launch_conv<false>(st, g, d->k, x, wp, y, slab, out_elems);
  }
  CSTP_LAUNCH_CHECK();
  if (g.ksplit > 1) {
    hipLaunchKernelGGL(b16_sum_slabs_kernel, dim3(b16_grid(out_elems / 4 + 1, 256)), dim3(256), 0, st, slab, g.ksplit, out_elems, y, out_elems);
    CSTP_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int cstp_b16_conv3d_backward_data(void* stream, const cstp_conv_desc* d, const uint16_t* dy, const float* w, uint16_t* dx,
                                             void* ws, size_t ws_bytes) {
  return cstp_b16_conv3d_backward_data_acc(stream, d, dy, w, dx, ws, ws_bytes, 0);
}

extern "C" int cstp_b16_conv3d_backward_data_acc(void* stream, const cstp_conv_desc* d, const uint16_t* dy, const float* w, uint16_t* dx,
                                                 void* ws, size_t ws_bytes, int32_t accumulate) {
  B16Geom q;
  CSTP_REQUIRE(b16_geom(d, q), "bad convolution geometry");
  CSTP_REQUIRE(dy && w && dx && ws && ws_bytes >= cstp_b16_conv3d_workspace_bytes(d), "null argument or workspace too small");
  CSTP_REQUIRE((d->k % 16) == 0 && q.ntaps <= B16_MAXTAPS, "bf16 data gradient: output channels a multiple of 16, at most 27 taps");
  CSTP_REQUIRE((size_t)d->n * d->k * q.Do * q.Ho * q.Wo * 2 < (1ull << 31), "bf16 path: gathered tensor must be < 2 GiB");
  CSTP_REQUIRE((reinterpret_cast<uintptr_t>(dx) & 7) == 0, "unaligned output");
  hipStream_t st = as_stream(stream);
  const int BM = d->c > 64 ? 128 : 64;
  const int Mp = (int)align_up((size_t)d->c, BM);
  const int Kw = q.ntaps * d->k;
  u16* wp = reinterpret_cast<u16*>(ws);
  if (!pack_skip(wp)) {
    const unsigned nb = b16_grid((size_t)Mp * Kw, 256);
    pack_record_b16(w, wp, (int)nb, d->k, d->c, q.ntaps, Mp, Kw, 1);
    hipLaunchKernelGGL(pack_w_b16_kernel, dim3(nb), dim3(256), 0, st, w, wp, d->k, d->c, q.ntaps, Mp, Kw, 1);
    CSTP_LAUNCH_CHECK();
  }
  const int ksplit = b16_dgrad_split(d, q);
  float* slab = b16_slabs(ws, d, q);
  const size_t out_elems = (size_t)d->n * d->c * d->d * d->h * d->w;
  for (int zt = 0; zt < d->st; ++zt)
    for (int zh = 0; zh < d->sh; ++zh)
      for (int zw = 0; zw < d->sw; ++zw) {
        B16Conv g;
        memset(&g, 0, sizeof(g));
        g.Nb = d->n; g.Cs = d->k; g.Ds = q.Do; g.Hs = q.Ho; g.Ws = q.Wo;
        g.Dq = (d->d - zt + d->st - 1) / d->st; g.Hq = (d->h - zh + d->sh - 1) / d->sh; g.Wq = (d->w - zw + d->sw - 1) / d->sw;
        if (g.Dq <= 0 || g.Hq <= 0 || g.Wq <= 0) continue;
        g.sst = g.ssh = g.ssw = 1;
        g.M = d->c; g.Do = d->d; g.Ho = d->h; g.Wo = d->w;
        g.ost = d->st; g.osh = d->sh; g.osw = d->sw; g.ozt = zt; g.ozh = zh; g.ozw = zw;
        g.Kw = Kw;
        g.contig = (d->st == 1 && d->sh == 1 && d->sw == 1 && ((d->d * d->h * d->w) % 4) == 0) ? 1 : 0;
        g.n_tiles_x = cdiv(d->n * g.Dq * g.Hq * g.Wq, 128);
        g.n_tiles_m = Mp / BM;
        g.kh = d->kh; g.kw = d->kw;
        int i = 0, tp = 0;
        for (int a = 0; a < d->kt; ++a)
          for (int b = 0; b < d->kh; ++b)
            for (int c = 0; c < d->kw; ++c, ++tp) {
              const int et = zt + d->pt - a, eh = zh + d->ph - b, ew = zw + d->pw - c;
              if ((et % d->st) != 0 || (eh % d->sh) != 0 || (ew % d->sw) != 0) continue;
              g.off[i][0] = (short)(et / d->st); g.off[i][1] = (short)(eh / d->sh); g.off[i][2] = (short)(ew / d->sw); g.off[i][3] = (short)tp;
              ++i;
            }
        g.ntaps = i;
        g.ksplit = ksplit;
        g.acc = (accumulate && ksplit <= 1) ? 1 : 0;
        if (b16_octets(d, dy, q.Wo, d->w)) b16_launch_conv<false, true>(st, g, d->c, dy, wp, dx, slab, out_elems);
        else b16_launch_conv<false>(st, g, d->c, dy, wp, dx, slab, out_elems);
        CSTP_LAUNCH_CHECK();
      }
  if (ksplit > 1) {
    hipLaunchKernelGGL(b16_sum_slabs_kernel, dim3(b16_grid(out_elems / 4 + 1, 256)), dim3(256), 0, st, slab, ksplit, out_elems, dx, out_elems,
                       accumulate ? 1 : 0);
    CSTP_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int cstp_b16_conv3d_backward_weight(void* stream, const cstp_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dw,
                                               void* ws, size_t ws_bytes, int32_t accumulate) {
  B16Geom q;
  CSTP_REQUIRE(b16_geom(d, q), "bad convolution geometry");
  CSTP_REQUIRE(x && dy && dw && ws && ws_bytes >= cstp_b16_conv3d_workspace_bytes(d), "null argument or workspace too small");
  CSTP_REQUIRE(q.tab || q.ntaps <= B16_MAXTAPS, "bf16 path: at most 27 filter taps for channel counts that are multiples of 16");
  CSTP_REQUIRE((size_t)d->n * d->c * q.Dp * q.Hp * q.Wp * 2 < (1ull << 31), "bf16 path: gathered tensor must be < 2 GiB");
  hipStream_t st = as_stream(stream);
  B16Wgrad g;
  memset(&g, 0, sizeof(g));
  g.Nb = d->n; g.Cs = d->c;
  g.M = d->k; g.Do = q.Do; g.Ho = q.Ho; g.Wo = q.Wo;
  g.st = d->st; g.sh = d->sh; g.sw = d->sw;
  g.ntaps = q.ntaps; g.K = q.ntaps * d->c;
  g.Kp = cdiv(g.K, 64) * 64;
  const int npo = q.Do * q.Ho * q.Wo;
  g.vec8 = ((npo % 8) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0) ? 1 : 0;
  g.nchunks = cdiv(d->n * npo, 64);
  const int S = b16_wgrad_split(d, q);
  g.chunks_per_split = cdiv(g.nchunks, S);
  const dim3 grid(cdiv(g.K, 64), cdiv(d->k, 64), S);
  float* slab = b16_slabs(ws, d, q);
  const size_t slab_stride = (size_t)grid.y * 64 * g.Kp;
  if (q.tab) {
    u16* xp = reinterpret_cast<u16*>(reinterpret_cast<char*>(ws) + b16_wpack_bytes(d, q));
    hipLaunchKernelGGL(pad_b16_kernel, dim3(b16_grid((size_t)d->n * d->c * q.Dp * q.Hp, 4)), dim3(256), 0, st, x, xp, d->n * d->c, d->d,
                       d->h, d->w, d->pt, d->ph, d->pw);
    CSTP_LAUNCH_CHECK();
    g.Ds = q.Dp; g.Hs = q.Hp; g.Ws = q.Wp;
    g.kh = d->kh; g.kw = d->kw;
    hipLaunchKernelGGL((wgrad_b16_kernel<true>), grid, dim3(256), 0, st, g, xp, dy, slab);
  } else {
    g.Ds = d->d; g.Hs = d->h; g.Ws = d->w;
    int i = 0;
    for (int a = 0; a < d->kt; ++a)
      for (int b = 0; b < d->kh; ++b)
        for (int c = 0; c < d->kw; ++c, ++i) {
          g.off[i][0] = (short)(a - d->pt); g.off[i][1] = (short)(b - d->ph); g.off[i][2] = (short)(c - d->pw); g.off[i][3] = (short)i;
        }
    // (pointwise layers: 16-byte row loads of x -- needs the 16-byte dY rows' condition on x too)
    if (b16_pointwise(d, x) && g.vec8) hipLaunchKernelGGL((wgrad_b16_kernel<false, true>), grid, dim3(256), 0, st, g, x, dy, slab);
    else hipLaunchKernelGGL((wgrad_b16_kernel<false>), grid, dim3(256), 0, st, g, x, dy, slab);
  }
  CSTP_LAUNCH_CHECK();
  if (S >= 16 && (size_t)d->k * g.K <= ((size_t)1 << 20))
    hipLaunchKernelGGL(b16_wgrad_reduce_wide_kernel, dim3((unsigned)cdiv((int)((size_t)d->k * g.K), 16)), dim3(256), 0, st, slab, S,
                       slab_stride, dw, d->k, d->c, q.ntaps, g.K, g.Kp, accumulate ? 1 : 0);
  else
    hipLaunchKernelGGL(b16_wgrad_reduce_kernel, dim3(b16_grid((size_t)d->k * g.K, 256)), dim3(256), 0, st, slab, S, slab_stride, dw, d->k,
                       d->c, q.ntaps, g.K, g.Kp, accumulate ? 1 : 0);
  CSTP_LAUNCH_CHECK();
  return 0;
}

// ---- BatchNorm --------------------------------------------------------------------------------------------------------------------
extern "C" size_t cstp_b16_bn_workspace_bytes(int32_t n, int32_t c, int32_t s, int32_t groups) {
  (void)s;
  if (n <= 0 || c <= 0 || groups <= 0 || (n % groups) != 0) return 0;
  // [c][groups][nsplit][2] fp64 partials, then [groups][c][2] fp32 per-group backward sums
  return align_up((size_t)c * groups * b16_bn_nsplit(n / groups, c) * 2 * sizeof(double), 256) + align_up((size_t)groups * c * 2 * sizeof(float), 256);
}

extern "C" int cstp_b16_bn_forward_train(void* stream, const uint16_t* x, const uint16_t* residual, uint16_t* y, const float* gamma,
                                         const float* beta, float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                                         float* scale_shift, int32_t n, int32_t c, int32_t s, int32_t groups, float eps, float momentum,
                                         int32_t relu, void* ws, size_t ws_bytes) {
  CSTP_REQUIRE(x && y && gamma && beta && save_mean && save_invstd && scale_shift, "null argument");
  CSTP_REQUIRE(n > 0 && c > 0 && s > 0 && groups > 0 && (n % groups) == 0, "bad shape");
  CSTP_REQUIRE((size_t)(n / groups) * s > 1, "train-mode BatchNorm needs more than 1 value per channel");
  CSTP_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "running stats must come as a pair");
  CSTP_REQUIRE(ws && ws_bytes >= cstp_b16_bn_workspace_bytes(n, c, s, groups), "workspace too small");
  hipStream_t st = as_stream(stream);
  const int npg = n / groups, ns = b16_bn_nsplit(npg, c);
  if (b16_bn_small_on() && (size_t)npg * s <= (size_t)B16_SMALL_PT * 1024 && (size_t)n * c * s < (1ull << 31)) {
    float2* ss2 = reinterpret_cast<float2*>(scale_shift);
    if ((size_t)npg * s <= (size_t)B16_SMALL_PT * 256)
      hipLaunchKernelGGL(b16_bn_small_fwd_kernel<256>, dim3(c), dim3(256), 0, st, x, residual, y, gamma, beta, running_mean, running_var,
                         save_mean, save_invstd, ss2, c, s, npg, groups, eps, momentum, relu);
    else
      hipLaunchKernelGGL(b16_bn_small_fwd_kernel<1024>, dim3(c), dim3(1024), 0, st, x, residual, y, gamma, beta, running_mean, running_var,
                         save_mean, save_invstd, ss2, c, s, npg, groups, eps, momentum, relu);
    CSTP_LAUNCH_CHECK();
    return 0;
  }
  double* part = reinterpret_cast<double*>(ws);
  const bool v8 = (s % 8) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15) == 0;
  const dim3 rgrid(c, groups * ns);
  if (v8) hipLaunchKernelGGL((b16_bn_reduce_kernel<0, true>), rgrid, dim3(256), 0, st, x, x, x, nullptr, nullptr, part, npg, c, s, ns, 0, nullptr);
  else hipLaunchKernelGGL((b16_bn_reduce_kernel<0, false>), rgrid, dim3(256), 0, st, x, x, x, nullptr, nullptr, part, npg, c, s, ns, 0, nullptr);
  CSTP_LAUNCH_CHECK();
  const int chunks = cdiv(s, B16_UNROLL * 256 * (v8 ? 8 : 1));
  const dim3 agrid((unsigned)((size_t)n * c * chunks));
  float2* ss = reinterpret_cast<float2*>(scale_shift);
  if (v8) hipLaunchKernelGGL((b16_bn_apply_fwd_kernel<true>), agrid, dim3(256), 0, st, x, residual, y, part, ns, groups, (double)npg * s, eps, momentum, gamma, beta, running_mean, running_var, save_mean, save_invstd, ss, c, s, npg, relu, chunks);
  else hipLaunchKernelGGL((b16_bn_apply_fwd_kernel<false>), agrid, dim3(256), 0, st, x, residual, y, part, ns, groups, (double)npg * s, eps, momentum, gamma, beta, running_mean, running_var, save_mean, save_invstd, ss, c, s, npg, relu, chunks);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_b16_bn_backward(void* stream, const uint16_t* x, const uint16_t* y, const uint16_t* dy, const float* gamma,
                                    const float* save_mean, const float* save_invstd, const float* scale_shift, uint16_t* dx,
                                    uint16_t* dresidual, float* dgamma, float* dbeta, int32_t n, int32_t c, int32_t s, int32_t groups,
                                    int32_t relu, void* ws, size_t ws_bytes, int32_t accumulate) {
  CSTP_REQUIRE(x && dy && gamma && save_mean && save_invstd && dx && dgamma && dbeta, "null argument");
  CSTP_REQUIRE(y != nullptr || !relu || scale_shift != nullptr, "ReLU mask needs y or scale_shift");
  CSTP_REQUIRE(n > 0 && c > 0 && s > 0 && groups > 0 && (n % groups) == 0, "bad shape");
  CSTP_REQUIRE(ws && ws_bytes >= cstp_b16_bn_workspace_bytes(n, c, s, groups), "workspace too small");
  hipStream_t st = as_stream(stream);
  const int npg = n / groups, ns = b16_bn_nsplit(npg, c);
  if (b16_bn_small_on() && (size_t)npg * s <= (size_t)B16_SMALL_PT * 256 && (size_t)n * c * s < (1ull << 31)) {
    hipLaunchKernelGGL(b16_bn_small_bwd_kernel, dim3(c), dim3(256), 0, st, x, y, dy, gamma, save_mean, save_invstd,
                       reinterpret_cast<const float2*>(scale_shift), dx, dresidual, dgamma, dbeta, c, s, npg, groups, relu, accumulate ? 1 : 0);
    CSTP_LAUNCH_CHECK();
    return 0;
  }
  double* part = reinterpret_cast<double*>(ws);
  const float2* ss = reinterpret_cast<const float2*>(scale_shift);
  const bool v8 = (s % 8) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy) |
                                    reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(dresidual)) & 15) == 0;
  const dim3 rgrid(c, groups * ns);
  if (v8) hipLaunchKernelGGL((b16_bn_reduce_kernel<1, true>), rgrid, dim3(256), 0, st, x, y, dy, save_mean, save_invstd, part, npg, c, s, ns, relu, ss);
  else hipLaunchKernelGGL((b16_bn_reduce_kernel<1, false>), rgrid, dim3(256), 0, st, x, y, dy, save_mean, save_invstd, part, npg, c, s, ns, relu, ss);
  CSTP_LAUNCH_CHECK();
  const float inv_count = (float)(1.0 / ((double)npg * s));
  const int chunks = cdiv(s, B16_UNROLL * 256 * (v8 ? 8 : 1));
  const dim3 agrid((unsigned)((size_t)n * c * chunks));
  if (v8) hipLaunchKernelGGL((b16_bn_apply_bwd_kernel<true>), agrid, dim3(256), 0, st, x, y, dy, gamma, save_mean, save_invstd, part, ns, groups, dgamma, dbeta, accumulate ? 1 : 0, dx, dresidual, c, s, npg, inv_count, relu, ss, chunks);
  else hipLaunchKernelGGL((b16_bn_apply_bwd_kernel<false>), agrid, dim3(256), 0, st, x, y, dy, gamma, save_mean, save_invstd, part, ns, groups, dgamma, dbeta, accumulate ? 1 : 0, dx, dresidual, c, s, npg, inv_count, relu, ss, chunks);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_b16_bn_forward_eval(void* stream, const uint16_t* x, const uint16_t* residual, uint16_t* y, const float* gamma,
                                        const float* beta, const float* running_mean, const float* running_var, int32_t n, int32_t c,
                                        int32_t s, float eps, int32_t relu) {
  CSTP_REQUIRE(x && y && gamma && beta && running_mean && running_var, "null argument");
  CSTP_REQUIRE(n > 0 && c > 0 && s > 0, "bad shape");
  const bool v8 = (s % 8) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15) == 0;
  const int chunks = cdiv(s, B16_UNROLL * 256 * (v8 ? 8 : 1));
  const dim3 agrid((unsigned)((size_t)n * c * chunks));
  if (v8) hipLaunchKernelGGL((b16_bn_eval_kernel<true>), agrid, dim3(256), 0, as_stream(stream), x, residual, y, gamma, beta, running_mean, running_var, c, s, eps, relu, chunks);
  else hipLaunchKernelGGL((b16_bn_eval_kernel<false>), agrid, dim3(256), 0, as_stream(stream), x, residual, y, gamma, beta, running_mean, running_var, c, s, eps, relu, chunks);
  CSTP_LAUNCH_CHECK();
  return 0;
}

// ---- pooling ------------------------------------------------------------------------------------------------------------------
static bool b16_pool_dims(int D, int H, int W, const int32_t* k, const int32_t* st, const int32_t* pd, int& Do, int& Ho, int& Wo) {
  for (int i = 0; i < 3; ++i)
    if (k[i] <= 0 || st[i] <= 0 || pd[i] < 0 || 2 * pd[i] > k[i]) return false;
  Do = (D + 2 * pd[0] - k[0]) / st[0] + 1;
  Ho = (H + 2 * pd[1] - k[1]) / st[1] + 1;
  Wo = (W + 2 * pd[2] - k[2]) / st[2] + 1;
  return Do > 0 && Ho > 0 && Wo > 0;
}

extern "C" int cstp_b16_maxpool3d_forward(void* stream, const uint16_t* x, uint16_t* y, int32_t* argmax, int32_t rows, int32_t d, int32_t h,
                                          int32_t w, const int32_t* kernel3, const int32_t* stride3, const int32_t* pad3) {
  CSTP_REQUIRE(x && y && argmax && kernel3 && stride3 && pad3 && rows > 0 && d > 0 && h > 0 && w > 0, "bad argument");
  int Do, Ho, Wo;
  CSTP_REQUIRE(b16_pool_dims(d, h, w, kernel3, stride3, pad3, Do, Ho, Wo), "bad pooling geometry");
  CSTP_REQUIRE((size_t)d * h * w < (1ull << 31), "plane too large for int32 argmax");
  const size_t total = (size_t)rows * Do * Ho * Wo;
  hipLaunchKernelGGL(b16_maxpool3d_fwd_kernel, dim3(b16_grid(total, 256)), dim3(256), 0, as_stream(stream), x, y, argmax, rows, d, h, w, Do,
                     Ho, Wo, kernel3[0], kernel3[1], kernel3[2], stride3[0], stride3[1], stride3[2], pad3[0], pad3[1], pad3[2]);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_b16_maxpool3d_backward(void* stream, const uint16_t* dy, const int32_t* argmax, uint16_t* dx, int32_t rows, int32_t d,
                                           int32_t h, int32_t w, const int32_t* kernel3, const int32_t* stride3, const int32_t* pad3) {
  CSTP_REQUIRE(dy && argmax && dx && kernel3 && stride3 && pad3 && rows > 0 && d > 0 && h > 0 && w > 0, "bad argument");
  int Do, Ho, Wo;
  CSTP_REQUIRE(b16_pool_dims(d, h, w, kernel3, stride3, pad3, Do, Ho, Wo), "bad pooling geometry");
  const size_t total = (size_t)rows * d * h * w;
  hipLaunchKernelGGL(b16_maxpool3d_bwd_kernel, dim3(b16_grid(total, 256)), dim3(256), 0, as_stream(stream), dy, argmax, dx, rows, d, h, w, Do,
                     Ho, Wo, kernel3[0], kernel3[1], kernel3[2], stride3[0], stride3[1], stride3[2], pad3[0], pad3[1], pad3[2]);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_b16_avgpool_forward(void* stream, const uint16_t* x, float* y, int32_t rows, int32_t s) {
  CSTP_REQUIRE(x && y && rows > 0 && s > 0, "bad argument");
  hipLaunchKernelGGL(b16_avgpool_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), x, y, rows, s);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_b16_avgpool_backward(void* stream, const float* dy, uint16_t* dx, int32_t rows, int32_t s) {
  CSTP_REQUIRE(dy && dx && rows > 0 && s > 0, "bad argument");
  hipLaunchKernelGGL(b16_avgpool_bwd_kernel, dim3(b16_grid((size_t)rows * s, 256)), dim3(256), 0, as_stream(stream), dy, dx, rows, s);
  CSTP_LAUNCH_CHECK();
  return 0;
}
