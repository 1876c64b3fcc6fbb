// Implicit-GEMM convolution kernels for gfx950 (MI355X), exact fp32 on the f32 MFMA
// (v_mfma_f32_32x32x2_f32: bit-for-bit an fmaf chain, 157 TF/s dense = the fp32 vector peak).
//
// One kernel family serves every GEMM-shaped op of the CSTP step:
//   K1  igemm_k1<MT, DGRAD, STRADDLE>   out[m][n] = sum_k Wp[k][m] * Xcol[k][n]
//         forward conv  (m = out channel, n = output position, k = (tap, in channel))
//         data gradient (m = in channel,  n = input position of one stride-parity class,
//                        k = (tap, out channel); only the taps that hit the class are visited)
//   K2  igemm_k2<MT, STRADDLE>          dWp[m][j] += sum_n dY[m][n] * Xcol[j][n]   (weight gradient,
//         n = output position, split over blocks, fp32 atomics into a packed slab)
// nn.Linear is the D=H=W=1, 1x1x1 case of the same kernels.
//
// Block = 256 threads = 4 waves; block tile (32*MT) x 128; each wave owns a (32*MT) x 32 strip,
// i.e. MT accumulators of 32x32 (16 VGPRs each).  Operands are staged global -> registers -> LDS
// (coalesced along n, the NCDHW-contiguous axis) and double buffered in LDS.
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "common.h"

#ifndef CSTP_PIN_PREFETCH
#define CSTP_PIN_PREFETCH 0
#endif
#ifndef CSTP_SETPRIO
#define CSTP_SETPRIO 0
#endif
#ifndef CSTP_NT_STORE
#define CSTP_NT_STORE 1
#endif
#if CSTP_NT_STORE
#define CSTP_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define CSTP_STORE(ptr, val) (*(ptr) = (val))
#endif
#ifndef CSTP_M16
#define CSTP_M16 1          // 144-row tiles on the 16x16x4 MFMA for 129..144-channel layers
#endif
#ifndef CSTP_K2_BKN
#define CSTP_K2_BKN 32      // positions per weight-gradient reduction tile (32 or 64)
#endif

namespace cstp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Geom {
  int Cs, Ds, Hs, Ws;      // gather source tensor [Nb][Cs][Ds][Hs][Ws]
  int Nb, Dp, Hp, Wp;      // position space (fwd: y dims; dgrad: FULL x dims; wgrad: dy dims)
  int kt, kh, kw, st, sh, sw, pt, ph, pw;
  int Cp;                  // channels per tap in the packed K / J order
  int M;                   // valid GEMM rows
  int Mp;                  // leading dimension of the packed A operand
  int Ktot;                // ntaps * Cp
  int acc = 0;             // forward / data-gradient kernels: out += instead of out = (the caller's gradient accumulation)
};

// ------------------------------------------------------------------------------------------
// weight packing: native [k_out][c_in][taps]  ->  K-major GEMM operand, zero padded
//   forward: Wp[(tap*Cp + c)][m = k_out]        dgrad: Wp[(tap*Cp + k_out)][m = c_in]
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void pack_native_body(const float* __restrict__ w, float* __restrict__ wp, int kout, int cin,
                                                 int ntaps, int Cp, int Mp, int Kp, int dgrad, const int blk, const int nblk) {
  const size_t total = (size_t)Kp * Mp;
  for (size_t i = (size_t)blk * blockDim.x + threadIdx.x; i < total; i += (size_t)nblk * blockDim.x) {
    const int m = (int)(i % Mp);
    const int k = (int)(i / Mp);
    const int tap = k / Cp, c = k - tap * Cp;
    float v = 0.f;
    if (tap < ntaps) {
      if (!dgrad) {
        if (m < kout && c < cin) v = w[((size_t)m * cin + c) * ntaps + tap];
      } else {
        if (m < cin && c < kout) v = w[((size_t)c * cin + m) * ntaps + tap];
      }
    }
    wp[i] = v;
  }
}
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int kout, int cin,
                                    int ntaps, int Cp, int Mp, int Kp, int dgrad) {
  pack_native_body(w, wp, kout, cin, ntaps, Cp, Mp, Kp, dgrad, (int)blockIdx.x, (int)gridDim.x);
}

// split-K output of the weight-gradient kernels: f32 atomics into ONE slab (default), or a plain store into this split's own
// slab (deterministic mode: cstp_set_deterministic; unpack_wgrad_kernel then sums the slabs in a fixed order)
__device__ __forceinline__ void wgrad_out(float* p, float v, bool det) {
  if (det) *p = v; else atomicAdd(p, v);
}

// dw[m][c][tap] = dwp[m][tap*Cp + c]  (x inv_x x inv_dy: the absmax cells of the 2xf16-split kernel, else null)
__device__ __forceinline__ void f16_scale(unsigned absmax_bits, float& scale, float& inv);
// nslabs > 1 (deterministic mode): the split-K partial slabs are summed here in a fixed order instead of by atomics.
// accumulate: dw += (the caller's gradient accumulation, e.g. straight into the flat gradient arena) instead of dw =.
__global__ void unpack_wgrad_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int M, int cin, int ntaps,
                                    int Cp, int Jp, const unsigned* __restrict__ xcell, const unsigned* __restrict__ dycell,
                                    int nslabs, size_t slab_stride, int accumulate) {
  float i0 = 1.f, i1 = 1.f;
  if (xcell != nullptr) { float sc; f16_scale(*xcell, sc, i0); f16_scale(*dycell, sc, i1); }
  const size_t total = (size_t)M * cin * ntaps;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int tap = (int)(i % ntaps);
    const size_t r = i / ntaps;
    const int c = (int)(r % cin);
    const int m = (int)(r / cin);
    const float* p = dwp + (size_t)m * Jp + tap * Cp + c;
    float v = p[0];
    for (int sidx = 1; sidx < nslabs; ++sidx) v += p[(size_t)sidx * slab_stride];
    v = v * i0 * i1;
    dw[i] = accumulate ? dw[i] + v : v;
  }
}

// ------------------------------------------------------------------------------------------
// K1: forward / data-gradient implicit GEMM
// ------------------------------------------------------------------------------------------
// WM = waves along M (1, 2 or 4), 4/WM waves along N: block tile (32*MT*WM) x (32*4/WM).  The wide
// 128-column tile (WM = 1) serves the big early layers; the 64- and 32-column tiles keep >= 256
// blocks in flight on the deep layers, whose position count is small (2B*2*7*7) but K is long.
// XFORM: the gathered operand is transformed on the fly, z = act(x * scale[g][c] + shift[g][c]) with a
// per-(BN group, channel) table -- i.e. the train-mode BatchNorm(+ReLU) that precedes this convolution
// is applied inside its gather and the normalised tensor never exists in HBM.  Zero padding applies
// to z (masked elements stay exactly 0).
// M16: the rows are covered by MT tiles of 16 (v_mfma_f32_16x16x4_f32, same FLOP rate) instead of 32, so a
// 144-channel layer (the S1/T1 class, 40 % of all FLOPs) runs an exact 144-row tile instead of padding to 160.
// TPB: K-tiles staged and multiplied per barrier (1 or 2).  Two tiles per barrier halve the block-wide
// synchronisations and let twice as many gathers be in flight; the autotuner decides per geometry.
template <int MT, int WM, bool DGRAD, bool STRADDLE, bool XFORM, bool M16, int TPB>
__global__ void __launch_bounds__(256)
igemm_k1(const Geom g, const float* __restrict__ wp, const float* __restrict__ src, const float* __restrict__ bias,
         float* __restrict__ out, int n_tiles_x, int n_tiles_m, const float2* __restrict__ in_ss, int in_npg,
         int in_relu) {
  constexpr int WN = 4 / WM;
  constexpr int BM = M16 ? 16 * MT : 32 * MT * WM, BN = 32 * WN, BK = 16;
  constexpr int BNP = M16 ? BN + 16 : BN;   // LDS row stride of the B tile (16x16x4 reads 2 k-rows per 32 lanes)
  constexpr int BR = BK * BN / 256;      // B-tile rows gathered per thread (8 / 4 / 2)
  constexpr int BRS = 256 / BN;          // row stride between them (2 / 4 / 8)
  static_assert(BM <= 160, "A staging holds at most 3 float4 per thread");
  static_assert(!M16 || WM == 1, "the 16-row variant uses the 128-column tile");
  __shared__ __attribute__((aligned(16))) float As[2 * TPB][BK * BM];    // [buffer * TPB + slot]
  __shared__ __attribute__((aligned(16))) float Bs[2 * TPB][BK * BNP];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

  // XCD-aware tile order: blocks b and b+8 share an XCD (L2).  Each XCD works through ONE contiguous
  // chunk of position tiles (neighbouring tiles share their 3x3 / temporal halo rows, so the halo is
  // an L2 hit instead of a second HBM fetch), and the n_tiles_m row tiles that read the same input
  // panel occupy consecutive slots of that XCD.
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

  const int HWs = g.Hs * g.Ws, DHWs = g.Ds * HWs;
  const int khw = g.kh * g.kw, ntaps = g.kt * khw;

  // ---- loader coordinates: this thread gathers column `col` of the B tile, rows krow0 + BRS*r
  const int col = t % BN, krow0 = t / BN;
  const bool nvalid = (n0 + col) < npos;
  int nb, npd, nph, npw;
  {
    int n = nvalid ? (n0 + col) : 0;
    npw = n % Wp; n /= Wp;
    nph = n % Hp; n /= Hp;
    npd = n % Dp; nb = n / Dp;
  }
  const size_t src_b = (size_t)nb * g.Cs * DHWs;
  const int ss_b = XFORM ? (nb / in_npg) * g.Cs : 0;   // row of the (scale, shift) table for this sample's BN group

  f32x16 acc[M16 ? 1 : MT];
  f32x4 acc16[M16 ? MT : 1][2];
#pragma unroll
  for (int i = 0; i < (M16 ? 1 : MT); ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
  for (int i = 0; i < (M16 ? MT : 1); ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc16[i][0][r] = 0.f; acc16[i][1][r] = 0.f; }

  constexpr int A_F4 = 4 * BM;                   // float4 per A tile (16 rows x BM/4)
  constexpr int A_ITERS = (A_F4 + 255) / 256;
  struct Stage {              // one K-tile in flight: named float4 members (an indexed float4 array ends up in scratch)
    float4 a0, a1, a2;
    float b[BR];
  };
  Stage sg0, sg1;
  sg0.a0 = sg0.a1 = sg0.a2 = sg1.a0 = sg1.a1 = sg1.a2 = make_float4(0.f, 0.f, 0.f, 0.f);
  static_assert(A_ITERS <= 3, "A tile staging assumes at most 3 float4 per thread");

  // iteration state
  int tap = -1, c0 = 0, it = 0;
  int toff = 0;
  bool tvalid = false;

  auto setup_tap = [&](int tp) __attribute__((always_inline)) -> bool {   // returns false if the tap never hits this parity class
    const int dt = tp / khw, rr = tp - dt * khw, dh = rr / g.kw, dw = rr - dh * g.kw;
    int id, ih, iw;
    if (DGRAD) {
      const int et = zt + g.pt - dt, eh = zh + g.ph - dh, ew = zw + g.pw - dw;
      if ((et % g.st) != 0 || (eh % g.sh) != 0 || (ew % g.sw) != 0) return false;
      id = npd + et / g.st; ih = nph + eh / g.sh; iw = npw + ew / g.sw;
    } else {
      id = npd * g.st - g.pt + dt; ih = nph * g.sh - g.ph + dh; iw = npw * g.sw - g.pw + dw;
    }
    tvalid = nvalid && (unsigned)id < (unsigned)g.Ds && (unsigned)ih < (unsigned)g.Hs && (unsigned)iw < (unsigned)g.Ws;
    toff = id * HWs + ih * g.Ws + iw;
    return true;
  };

  auto first_tile = [&]() __attribute__((always_inline)) -> bool {
    if (STRADDLE) { it = 0; return g.Ktot > 0; }
    tap = 0; c0 = 0;
    while (tap < ntaps && !setup_tap(tap)) ++tap;
    return tap < ntaps;
  };
  auto advance = [&]() __attribute__((always_inline)) -> bool {
    if (STRADDLE) { ++it; return it * BK < g.Ktot; }
    c0 += BK;
    if (c0 < g.Cp) return true;
    c0 = 0; ++tap;
    while (tap < ntaps && !setup_tap(tap)) ++tap;
    return tap < ntaps;
  };

  auto load_tile = [&](Stage& sg) __attribute__((always_inline)) {
    const int kbase = STRADDLE ? it * BK : tap * g.Cp + c0;
    {
      const float* abase = wp + (size_t)kbase * g.Mp + m0;
      auto a_at = [&](int i) __attribute__((always_inline)) -> float4 {
        int idx = t + 256 * i;
        if (idx >= A_F4) idx = 0;   // tail threads re-read element 0; their LDS store is skipped
        const int row = idx / (BM / 4), c4 = idx - row * (BM / 4);
        return *reinterpret_cast<const float4*>(abase + (size_t)row * g.Mp + c4 * 4);
      };
      sg.a0 = a_at(0);
      if (A_ITERS > 1) sg.a1 = a_at(1);
      if (A_ITERS > 2) sg.a2 = a_at(2);
    }
#pragma unroll
    for (int r = 0; r < BR; ++r) {
      const int kr = krow0 + BRS * r;
      // Loads are UNCONDITIONAL (address clamped to element 0 when masked) and the zero is selected
      // afterwards: a branch around each load makes hipcc wait vmcnt(0) per element (serialised).
      bool ok;
      size_t off;
      if (STRADDLE) {
        const int k = kbase + kr;
        const int kc = k < g.Ktot ? k : 0;
        const int tp = kc / g.Cp, c = kc - tp * g.Cp;
        const int dt = tp / khw, rr = tp - dt * khw, dh = rr / g.kw, dw = rr - dh * g.kw;
        const int id = npd * g.st - g.pt + dt, ih = nph * g.sh - g.ph + dh, iw = npw * g.sw - g.pw + dw;
        ok = nvalid && k < g.Ktot && c < g.Cs && (unsigned)id < (unsigned)g.Ds && (unsigned)ih < (unsigned)g.Hs &&
             (unsigned)iw < (unsigned)g.Ws;
        off = src_b + (size_t)c * DHWs + id * HWs + ih * g.Ws + iw;
      } else {
        const int c = c0 + kr;
        ok = tvalid && c < g.Cs;
        off = src_b + (size_t)c * DHWs + toff;
      }
      const float v = src[ok ? off : 0];
      if (XFORM) {
        const float2 ss = in_ss[ss_b + (ok ? c0 + kr : 0)];
        float z = v * ss.x + ss.y;
        if (in_relu) z = fmaxf(z, 0.f);
        sg.b[r] = ok ? z : 0.f;
      } else {
        sg.b[r] = ok ? v : 0.f;
      }
    }
  };
  auto store_tile = [&](int buf, const Stage& sg) __attribute__((always_inline)) {   // buf = buffer * TPB + slot
    if (t < A_F4) *reinterpret_cast<float4*>(&As[buf][t * 4]) = sg.a0;
    if (A_ITERS > 1 && t + 256 < A_F4) *reinterpret_cast<float4*>(&As[buf][(t + 256) * 4]) = sg.a1;
    if (A_ITERS > 2 && t + 512 < A_F4) *reinterpret_cast<float4*>(&As[buf][(t + 512) * 4]) = sg.a2;
#pragma unroll
    for (int r = 0; r < BR; ++r) Bs[buf][(krow0 + BRS * r) * BNP + col] = sg.b[r];
  };

  const int lrow = lane >> 5, lcol = lane & 31;
  const int wm = wave % WM, wn = wave / WM;
  auto compute = [&](int buf) __attribute__((always_inline)) {     // buf = buffer * TPB + slot
    if (M16) {
      // 16x16x4: lane l feeds A[m = l&15][k = l>>4] and B[k = l>>4][n = l&15]; each wave owns 32 columns = 2 tiles
      const float* Ab = &As[buf][(lane >> 4) * BM + (lane & 15)];
      const float* Bb = &Bs[buf][(lane >> 4) * BNP + wn * 32 + (lane & 15)];
#pragma unroll
      for (int kk = 0; kk < BK; kk += 4) {
        const float b0 = Bb[kk * BNP], b1 = Bb[kk * BNP + 16];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const float a = Ab[kk * BM + mt * 16];
          acc16[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc16[mt][0], 0, 0, 0);
          acc16[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc16[mt][1], 0, 0, 0);
        }
      }
    } else {
      // operand fragments of k-pair kk+2 are read from LDS while the MFMAs of k-pair kk run
      const float* Ab = &As[buf][lrow * BM + wm * MT * 32 + lcol];
      const float* Bb = &Bs[buf][lrow * BN + wn * 32 + lcol];
      float a_cur[MT], a_nxt[MT], b_cur, b_nxt = 0.f;
      b_cur = Bb[0];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) { a_cur[mt] = Ab[mt * 32]; a_nxt[mt] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < BK; kk += 2) {
        if (kk + 2 < BK) {
          b_nxt = Bb[(kk + 2) * BN];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) a_nxt[mt] = Ab[(kk + 2) * BM + mt * 32];
        }
#pragma unroll
        for (int mt = 0; mt < (M16 ? 1 : MT); ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[mt], b_cur, acc[mt], 0, 0, 0);
        b_cur = b_nxt;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a_cur[mt] = a_nxt[mt];
      }
    }
  };

  if (TPB == 1) {
    bool have = first_tile();
    if (have) { load_tile(sg0); store_tile(0, sg0); }
    __syncthreads();
    int buf = 0;
    while (have) {
      const bool have_next = advance();
      if (have_next) load_tile(sg0);
      compute(buf);
      if (have_next) store_tile(buf ^ 1, sg0);
      __syncthreads();
      buf ^= 1;
      have = have_next;
    }
  } else {
    bool h0 = first_tile(), h1 = false;
    if (h0) {
      load_tile(sg0);
      h1 = advance();
      if (h1) load_tile(sg1);
      store_tile(0, sg0);
      if (h1) store_tile(1, sg1);
    }
    __syncthreads();
    int buf = 0;
    while (h0) {
      const bool n0 = h1 && advance();     // a missing second tile means the sequence has ended
      if (n0) load_tile(sg0);
      const bool n1 = n0 && advance();
      if (n1) load_tile(sg1);
      compute(buf * 2);
      if (h1) compute(buf * 2 + 1);
      if (n0) store_tile((buf ^ 1) * 2, sg0);
      if (n1) store_tile((buf ^ 1) * 2 + 1, sg1);
      __syncthreads();
      buf ^= 1;
      h0 = n0;
      h1 = n1;
    }
  }

  if (M16) {
    // C layout of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg, two column tiles per wave.  Lane groups
    // q and q^1 swap one register (lane ^ 16) so that 32 consecutive lanes hold 32 consecutive columns of ONE row:
    // every store instruction then writes whole 128-byte lines (the un-swapped layout writes 64-byte halves and
    // the PMC WRITE_SIZE showed 1.37x the output bytes).
    const int n = n0 + wn * 32 + lcol;                 // this lane's column after the swap
    const int q = lane >> 4;
    const bool odd = (q & 1) != 0;
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
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v0 = acc16[mt][0][r], v1 = acc16[mt][1][r];
        const float recv = __shfl_xor(odd ? v0 : v1, 16, 64);
        const int m_even = m0 + mt * 16 + (q & ~1) * 4 + r, m_odd = m_even + 4;
        float ve = odd ? recv : v0;                    // row m_even: own cols 0-15 | partner's cols 16-31
        float vo = odd ? v1 : recv;                    // row m_odd
        // streaming (nt) stores: the output is not re-read by this kernel, keep the L2 for the input halo rows
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
    return;
  }
  // ---- epilogue: C layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int n = n0 + wn * 32 + lcol;
  if (n < npos) {
    size_t obase, cstride;
    if (DGRAD) {
      int q = n;
      const int pw = q % Wp; q /= Wp;
      const int ph = q % Hp; q /= Hp;
      const int pd = q % Dp; const int b = q / Dp;
      const int HWf = g.Hp * g.Wp;
      cstride = (size_t)g.Dp * HWf;
      obase = (size_t)b * g.M * cstride + (size_t)(zt + g.st * pd) * HWf + (zh + g.sh * ph) * g.Wp + (zw + g.sw * pw);
    } else {
      const int S = Dp * Hp * Wp;
      const int b = n / S, sp = n - b * S;
      cstride = (size_t)S;
      obase = (size_t)b * g.M * cstride + sp;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
        if (m < g.M) {
          float v = acc[mt][r];
          if (bias != nullptr) v += bias[m];
          if (g.acc) v += out[obase + (size_t)m * cstride];
          CSTP_STORE(out + obase + (size_t)m * cstride, v);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// K2: weight gradient.  dwp[m][j] += sum_{n in split} dy[m][n] * xcol[j][n],  j = tap*Cp + c
// ------------------------------------------------------------------------------------------
template <int MT, bool STRADDLE, bool VEC4, int BKN, bool XFORM, bool M16>
__global__ void __launch_bounds__(256)
igemm_k2(const Geom g, const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dwp, int Jtot,
         int Jp, int ktiles_total, int ktiles_per_split, int ntm, int ntj, int nsplit,
         const float2* __restrict__ in_ss, int in_npg, int in_groups, int in_relu, size_t det_stride) {
  // Block tile (32*MT) x 128 outputs; the reduction runs over positions in tiles of 32, staged
  // global -> registers -> LDS ([row][pos], row stride 33: conflict-free both for the coalesced
  // stores along pos and for the MFMA operand reads along rows) with the NEXT tile's loads in flight
  // while the current one is multiplied (LDS double buffered, one barrier per tile).
  // M16: rows in MT tiles of 16 on the 16x16x4 MFMA (exact 144-row tile); its operand reads want an even
  // row stride (2 k-columns per 32 lanes), the 32x32x2 reads an odd one
  constexpr int BM = M16 ? 16 * MT : 32 * MT, BJ = 128, LD = M16 ? BKN + 2 : BKN + 1;
  constexpr int ARS = 256 / BKN;             // scalar A loader: dy-row step between a thread's loads
  constexpr int A4S = 1024 / BKN;            // float4 A loader: dy-row step
  constexpr int BSUB = 64 / BKN;             // B loader: xcol rows covered by one wave instruction
  constexpr int AR = VEC4 ? (BM + A4S - 1) / A4S : (BM + ARS - 1) / ARS;   // dy loads per thread (float4 / single floats)
  constexpr int BR = 32 / BSUB;              // xcol rows per thread (the wave gathers exactly the 32 columns it consumes)
  __shared__ float As[2][BM * LD];
  __shared__ float Bs[2][BJ * LD];
  // XFORM: (scale, shift) of this block's 128 gathered columns for up to 4 BN groups; the x operand is
  // turned into z = act(x*scale + shift) as it is staged (see igemm_k1)
  __shared__ float2 Ss[XFORM ? 4 * BJ : 1];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // XCD-aware order (blocks b, b+8 share an L2): the ntj column tiles that read the same dY panel of
  // one (row tile, split) sit on consecutive slots of ONE xcd.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int jt = slot % ntj;
  const int panel = (slot / ntj) * 8 + xcd;          // panel = (row tile, split): one dY stream
  if (panel >= ntm * nsplit) return;
  const int mtile = panel % ntm, split = panel / ntm;
  const int m0 = mtile * BM, j0 = jt * BJ;
  const int S = g.Dp * g.Hp * g.Wp;
  const int npos = g.Nb * S;
  const int kt_begin = split * ktiles_per_split;
  int kt_end = kt_begin + ktiles_per_split;
  if (kt_end > ktiles_total) kt_end = ktiles_total;
  if (kt_begin >= kt_end) return;

  const int HWs = g.Hs * g.Ws, DHWs = g.Ds * HWs;
  const int khw = g.kh * g.kw;
  const int jw0 = j0 + wave * 32;
  const bool wave_active = jw0 < Jtot;
  int cw0 = 0, wdt = 0, wdh = 0, wdw = 0;
  if (!STRADDLE) {
    const int tapw = jw0 / g.Cp;
    cw0 = jw0 - tapw * g.Cp;
    wdt = tapw / khw;
    const int rr = tapw - wdt * khw;
    wdh = rr / g.kw; wdw = rr - wdh * g.kw;
  }

  f32x16 acc[M16 ? 1 : MT];
  f32x4 acc16[M16 ? MT : 1][2];
#pragma unroll
  for (int i = 0; i < (M16 ? 1 : MT); ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
  for (int i = 0; i < (M16 ? MT : 1); ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc16[i][0][r] = 0.f; acc16[i][1][r] = 0.f; }

  const int lrow = lane >> 5, lcol = lane & 31;
  if (XFORM) {
    for (int i = t; i < in_groups * BJ; i += 256) {
      const int gi = i / BJ, jj = i - gi * BJ;
      const int j = j0 + jj;
      const int c = j < Jtot ? (j % g.Cp) : g.Cs;
      Ss[i] = c < g.Cs ? in_ss[gi * g.Cs + c] : make_float2(0.f, 0.f);
    }
    // visible to every wave after the first __syncthreads() below (before any tile is staged? no:
    // the prologue stages tile 0 first) -> explicit barrier
    __syncthreads();
  }
  const int pos = t % BKN, arow = t / BKN;   // scalar A loader: position within the tile, first dy row
  const int q4 = t % (BKN / 4), arow4 = t / (BKN / 4);   // float4 A loader: which 4 positions, first dy row
  const int bpos = lane % BKN, bsub = lane / BKN;        // B loader: position, first xcol row of the wave's 32
  float va[VEC4 ? 4 * AR : AR], vb[BR];
  bool a_valid4 = false;
  bool a_valid = false;                       // this thread's position exists (tile-level)
  unsigned b_mask = 0;                        // per-row validity of the gathered x elements
  int b_grp = 0;                              // BN group of this thread's position (XFORM)

  auto load_tile = [&](int kti) __attribute__((always_inline)) {
    const int n = kti * BKN + (VEC4 ? bpos : pos);   // VEC4: coordinates serve the B gather only
    const bool nvalid = n < npos;
    int b = 0, sp = 0, od = 0, oh = 0, ow = 0;
    if (nvalid) {
      b = n / S; sp = n - b * S;
      int q = sp;
      ow = q % g.Wp; q /= g.Wp;
      oh = q % g.Hp; od = q / g.Hp;
    }
    a_valid = nvalid;
    if (VEC4) {
      // 4 consecutive positions never straddle a clip (S % 4 == 0) and are 16-byte aligned
      const int n4 = kti * BKN + 4 * q4;
      const bool v4 = n4 < npos;
      const int b4 = v4 ? n4 / S : 0, sp4 = v4 ? n4 - b4 * S : 0;
      a_valid4 = v4;
      const size_t ab4 = (size_t)b4 * g.M * S + sp4;
#pragma unroll
      for (int r = 0; r < AR; ++r) {
        const int m = m0 + arow4 + A4S * r;
        const float4 v = *reinterpret_cast<const float4*>(dy + ((v4 && m < g.M) ? ab4 + (size_t)m * S : 0));
        va[4 * r + 0] = v.x; va[4 * r + 1] = v.y; va[4 * r + 2] = v.z; va[4 * r + 3] = v.w;
      }
    } else {
      const size_t ab = (size_t)b * g.M * S + sp;
#pragma unroll
      for (int r = 0; r < AR; ++r) {            // unconditional clamped loads, zero selected at store time
        const int m = m0 + arow + ARS * r;
        va[r] = dy[(nvalid && m < g.M) ? ab + (size_t)m * S : 0];
      }
    }
    b_mask = 0;
    if (XFORM) b_grp = b / in_npg;
    if (wave_active) {
      const size_t xb = (size_t)b * g.Cs * DHWs;
      if (!STRADDLE) {
        const int id = od * g.st - g.pt + wdt, ih = oh * g.sh - g.ph + wdh, iw = ow * g.sw - g.pw + wdw;
        const bool v0 = nvalid && (unsigned)id < (unsigned)g.Ds && (unsigned)ih < (unsigned)g.Hs &&
                        (unsigned)iw < (unsigned)g.Ws;
        const int toff = id * HWs + ih * g.Ws + iw;
#pragma unroll
        for (int r = 0; r < BR; ++r) {
          const int c = cw0 + bsub + BSUB * r;
          const bool ok = v0 && c < g.Cs;
          vb[r] = x[ok ? xb + (size_t)c * DHWs + toff : 0];
          b_mask |= (ok ? 1u : 0u) << r;
        }
      } else {
#pragma unroll
        for (int r = 0; r < BR; ++r) {
          const int j = jw0 + bsub + BSUB * r;
          const int jc = j < Jtot ? j : 0;
          const int tp = jc / g.Cp, c = jc - tp * g.Cp;
          const int dt = tp / khw, rr = tp - dt * khw, dh = rr / g.kw, dw = rr - dh * g.kw;
          const int id = od * g.st - g.pt + dt, ih = oh * g.sh - g.ph + dh, iw = ow * g.sw - g.pw + dw;
          const bool ok = nvalid && j < Jtot && c < g.Cs && (unsigned)id < (unsigned)g.Ds &&
                          (unsigned)ih < (unsigned)g.Hs && (unsigned)iw < (unsigned)g.Ws;
          vb[r] = x[ok ? xb + (size_t)c * DHWs + id * HWs + ih * g.Ws + iw : 0];
          b_mask |= (ok ? 1u : 0u) << r;
        }
      }
    }
  };
  auto store_tile = [&](int buf) __attribute__((always_inline)) {
    if (VEC4) {
#pragma unroll
      for (int r = 0; r < AR; ++r) {
        const int ml = arow4 + A4S * r;
        const bool ok = a_valid4 && (m0 + ml) < g.M;
        if (ml < BM) {
#pragma unroll
          for (int i = 0; i < 4; ++i) As[buf][ml * LD + 4 * q4 + i] = ok ? va[4 * r + i] : 0.f;
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < AR; ++r) {
        const int ml = arow + ARS * r;
        if (ml < BM) As[buf][ml * LD + pos] = (a_valid && (m0 + ml) < g.M) ? va[r] : 0.f;
      }
    }
    if (wave_active) {
#pragma unroll
      for (int r = 0; r < BR; ++r) {
        float v = vb[r];
        if (XFORM) {
          const float2 ss = Ss[b_grp * BJ + wave * 32 + bsub + BSUB * r];
          v = v * ss.x + ss.y;
          if (in_relu) v = fmaxf(v, 0.f);
        }
        Bs[buf][(wave * 32 + bsub + BSUB * r) * LD + bpos] = ((b_mask >> r) & 1u) ? v : 0.f;
      }
    }
  };

  load_tile(kt_begin);
  store_tile(0);
  __syncthreads();
  int buf = 0;
  for (int kti = kt_begin; kti < kt_end; ++kti) {
    const bool have_next = (kti + 1) < kt_end;
    if (have_next) load_tile(kti + 1);
    if (wave_active && M16) {
      const float* Ab = &As[buf][(lane & 15) * LD + (lane >> 4)];
      const float* Bb = &Bs[buf][(wave * 32 + (lane & 15)) * LD + (lane >> 4)];
#pragma unroll
      for (int kk = 0; kk < BKN; kk += 4) {
        const float b0 = Bb[kk], b1 = Bb[16 * LD + kk];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const float a = Ab[mt * 16 * LD + kk];
          acc16[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc16[mt][0], 0, 0, 0);
          acc16[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc16[mt][1], 0, 0, 0);
        }
      }
    } else if (wave_active) {
      const float* Ab = &As[buf][lcol * LD + lrow];
      const float* Bb = &Bs[buf][(wave * 32 + lcol) * LD + lrow];
      float a_cur[MT], a_nxt[MT], b_cur, b_nxt = 0.f;
      b_cur = Bb[0];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) { a_cur[mt] = Ab[mt * 32 * LD]; a_nxt[mt] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < BKN; kk += 2) {
        if (kk + 2 < BKN) {
          b_nxt = Bb[kk + 2];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) a_nxt[mt] = Ab[mt * 32 * LD + kk + 2];
        }
#pragma unroll
        for (int mt = 0; mt < (M16 ? 1 : MT); ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[mt], b_cur, acc[mt], 0, 0, 0);
        b_cur = b_nxt;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a_cur[mt] = a_nxt[mt];
      }
    }
    if (have_next) store_tile(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  if (M16) {
    if (wave_active) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int j = jw0 + nt * 16 + (lane & 15);
        if (j >= Jtot) continue;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = m0 + mt * 16 + (lane >> 4) * 4 + r;
            if (m < g.M) wgrad_out(&dwp[(size_t)split * det_stride + (size_t)m * Jp + j], acc16[mt][nt][r], det_stride != 0);
          }
        }
      }
    }
    return;
  }
  const int j = jw0 + lcol;
  if (wave_active && j < Jtot) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
        if (m < g.M) wgrad_out(&dwp[(size_t)split * det_stride + (size_t)m * Jp + j], acc[mt][r], det_stride != 0);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
}  // namespace cstp
#include "igemm_split.h"
#include "igemm_patch.h"
#include "igemm_tpatch.h"
#include "igemm_twres.h"
#include "igemm_wpatch.h"
#include "igemm_wtpatch.h"
#include "linear.h"
#include "pack_b16.h"
namespace cstp {

static int pick_mt(int M) {   // K2 (weight gradient): rows per block = 32*mt, minimise padded rows
  int best = 1;
  double bestc = 1e30;
  for (int mt = 1; mt <= 5; ++mt) {
    const int bm = 32 * mt;
    const double c = (double)cdiv(M, bm) * bm * (1.0 + 0.25 / mt);
    if (c < bestc - 1e-9) { bestc = c; best = mt; }
  }
  return best;
}

// sp = 1: the split kernels igemm_k1s (igemm_split.h), tile (16*mt) x 128, 512 threads
// sp = 2: the LDS-resident-patch kernel igemm_k1p (igemm_patch.h), tile (16*mt) x 224, stride-1 1x3x3 layers, f16 pair only
struct Tile { int mt, wm, m16, tpb, sp; };   // tpb: 0/1 = one K-tile per barrier, 2 = two
static inline int tile_bm(const Tile& t) { return (t.m16 || t.sp) ? 16 * t.mt : 32 * t.mt * t.wm; }
static inline int tile_bn(const Tile& t) {   // split: wm = 128-column halves
  return t.sp == 2 ? KP_NPOS : t.sp ? (t.wm == 2 ? 256 : 128) : 32 * (4 / t.wm);
}
static inline bool patch_mt_ok(int mt) { return mt == 4 || mt == 8 || mt == 9; }
static bool native_only();
static int split_planes();
// the patch kernel serves 1x3x3, stride 1, padding (0,1,1) in the f16-pair arithmetic, frames whose patch fits its LDS image
static bool patch_geom_ok(const cstp_conv_desc& d) {
  if (!(d.kt == 1 && d.kh == 3 && d.kw == 3 && d.st == 1 && d.sh == 1 && d.sw == 1 && d.pt == 0 && d.ph == 1 && d.pw == 1))
    return false;
  if (native_only() || split_planes() != 2 || d.c < 16 || d.k < 16) return false;
  return patch_rows_needed(d.n * d.d, d.h, d.w) <= KP_ROWS;
}
// the temporal patch kernel igemm_k1t (igemm_tpatch.h): 3x1x1, stride 1, padding (1,0,0), f16 pair; tiles of 8 frames x 28
// columns without remainder, whole 16-channel groups on both sides (the data gradient gathers the output channels)
static bool tpatch_geom_ok(const cstp_conv_desc& d) {
  if (!(d.kt == 3 && d.kh == 1 && d.kw == 1 && d.st == 1 && d.sh == 1 && d.sw == 1 && d.pt == 1 && d.ph == 0 && d.pw == 0))
    return false;
  if (native_only() || split_planes() != 2 || d.c < 16 || d.k < 16 || (d.c & 15) != 0 || (d.k & 15) != 0) return false;
  return d.d % KT_DT == 0 && (d.h * d.w) % KT_WT == 0;
}
// the weight-gradient patch kernel igemm_k2p (igemm_wpatch.h): the same layers; its x staging leads by <= 5 intervals of 64 rows
static bool wpatch_geom_ok(const cstp_conv_desc& d) {
  if (!(d.kt == 1 && d.kh == 3 && d.kw == 3 && d.st == 1 && d.sh == 1 && d.sw == 1 && d.pt == 0 && d.ph == 1 && d.pw == 1))
    return false;
  if (native_only() || split_planes() != 2 || d.c < 16 || d.k < 16) return false;
  // (stream rows and frame counts stay below 2^25 and the divisors below 129: the kernel divides by multiplication)
  if ((long)d.n * d.d * (d.h + 1) * (d.w + 2) >= (1l << 25) || d.h + 1 > 128 || d.d > 128) return false;
  return 2 * (d.w + 2) + 66 <= 5 * 64;
}
// the temporal weight-gradient kernel igemm_k2t (igemm_wtpatch.h): 3x1x1, stride 1, padding (1,0,0), f16 pair, frames of whole
// 32-position chunks (one aligned 128-byte line per channel, frame and chunk); item / frame-line counts below 2^26 (the kernel
// divides by multiplication)
static bool twpatch_geom_ok(const cstp_conv_desc& d) {
  if (!(d.kt == 3 && d.kh == 1 && d.kw == 1 && d.st == 1 && d.sh == 1 && d.sw == 1 && d.pt == 1 && d.ph == 0 && d.pw == 0))
    return false;
  if (native_only() || split_planes() != 2 || d.c < 16 || d.k < 16) return false;
  if ((d.h * d.w) % 16 != 0 || d.d > 255) return false;      // chunks of 32 positions, or of 16 (28 x 28 frames)
  return (long)d.n * (d.h * d.w / 16) * (d.d + 1) < (1l << 26);
}
static inline bool split_mt_ok(int mt) { return mt == 2 || mt == 3 || mt == 4 || mt == 5 || mt == 6 || mt == 8 || mt == 9; }
static inline bool split_tile_ok(const Tile& t) { return split_mt_ok(t.mt) && (t.wm != 2 || t.mt >= 8); }

// K1 tile choice.  Model: blocks are dealt to the 256 CUs in rounds (a CU's resident blocks share its
// matrix pipes, so time ~ max blocks per CU x work per block); per-block work ~ BM x BN (K is fixed);
// tile efficiency falls with operand traffic per FLOP (1/BM + 1/BN); a grid of <= 1 block per CU
// cannot overlap its own loads with another block's MFMAs.
static Tile pick_tile(int M, long npos, int nclass) {
  // developer override for tile experiments: CSTP_TILE="mt,wm[,tpb]" (e.g. "4,1" or "2,2,2"); unset in production
  static const char* ov = getenv("CSTP_TILE");
  if (ov != nullptr) {
    int mt = 0, wm = 0, tpb = 1;
    if (ov[0] == 's' && sscanf(ov + 1, "%d", &mt) == 1 && split_mt_ok(mt))    // "s9": split kernel; "s9x": 256-column tile
      return Tile{mt, (strchr(ov, 'x') != nullptr && mt >= 8) ? 2 : 1, 0, 1, 1};
    if (sscanf(ov, "%d,%d,%d", &mt, &wm, &tpb) >= 2 && mt >= 1 &&
        ((wm == 1 && mt <= 5) || (wm == 2 && mt <= 2) || (wm == 4 && mt == 1)))
      return Tile{mt, wm, 0, tpb == 2 ? 2 : 1};
  }
  if (CSTP_M16 && M > 128 && M <= 144 && npos * nclass >= 1024) return Tile{9, 1, 1};   // exact 144-row tile
  static const Tile cand[] = {{1, 1, 0}, {2, 1, 0}, {3, 1, 0}, {4, 1, 0}, {5, 1, 0}, {1, 2, 0}, {2, 2, 0}, {1, 4, 0}};
  Tile best = cand[0];
  double bestc = 1e300;
  for (const Tile& t : cand) {
    const int bm = 32 * t.mt * t.wm, bn = 32 * (4 / t.wm);
    const double ntm = cdiv(M, bm), ntx = (double)((npos + bn - 1) / bn);
    const double blocks = ntm * ntx * nclass;
    double eff = 1.0 / (1.0 + 8.0 * (1.0 / bm + 1.0 / bn));
    if (blocks <= 256.0) eff *= 0.8;
    const double rounds = blocks >= 2048.0 ? blocks / 256.0 : (double)(((long)blocks + 255) / 256);
    const double c = rounds * bm * bn / eff;
    if (c < bestc * (1.0 - 1e-9)) { bestc = c; best = t; }
  }
  return best;
}

// ---- measured tile choices (cstp_conv3d_autotune) ------------------------------------------------
// The analytic model above ranks tiles poorly on the deep layers (long K, few positions: L2 panel reuse and
// grid fill interact), so the host may time the candidates once per (layer geometry, direction) and the
// winner is remembered here.  All tiles run the same k-ordered fmaf chains: the choice never changes results.
struct TuneKey {
  int v[16];
  bool operator==(const TuneKey& o) const { return memcmp(v, o.v, sizeof(v)) == 0; }
};
struct TuneKeyHash {
  size_t operator()(const TuneKey& k) const {
    size_t h = 1469598103934665603ull;
    for (int i = 0; i < 16; ++i) h = (h ^ (size_t)(unsigned)k.v[i]) * 1099511628211ull;
    return h;
  }
};
static std::mutex g_tune_mu;
static std::unordered_map<TuneKey, Tile, TuneKeyHash> g_tuned;
static thread_local const Tile* g_force_tile = nullptr;   // set only inside cstp_conv3d_autotune
static thread_local int g_force_mode = -1;

// tuned entries live in one class per GEMM arithmetic (cstp_gemm_set_split_terms / CSTP_GEMM): 1 = tiles chosen among the
// native f32 MFMA kernels only, 2 / 3 = chosen with the f16-pair / bf16-triple split kernels among the candidates
static TuneKey tune_key(const cstp_conv_desc& d, int mode) {
  TuneKey k;
  const int f[16] = {d.n, d.c, d.d, d.h, d.w, d.k, d.kt, d.kh, d.kw, d.st, d.sh, d.sw, d.pt, d.ph, d.pw,
                     mode + 16 * (native_only() ? 1 : split_planes())};
  memcpy(k.v, f, sizeof(f));
  return k;
}
static bool lookup_tuned(const cstp_conv_desc& d, int mode, Tile& t) {
  if (g_force_tile != nullptr && g_force_mode == mode) { t = *g_force_tile; return true; }
  std::lock_guard<std::mutex> lk(g_tune_mu);
  auto it = g_tuned.find(tune_key(d, mode));
  if (it == g_tuned.end()) return false;
  t = it->second;
  return true;
}

struct ConvPlan {
  int Do, Ho, Wo, ntaps;
  // forward
  Tile f_t; int f_Cp, f_Mp, f_Kp; bool f_straddle;
  // dgrad
  Tile d_t; int d_Cp, d_Mp, d_Kp;
  // wgrad
  int w_mt, w_blocks, w_Cp, w_Jtot, w_Jp; bool w_straddle; bool w_split; bool w_patch; bool w_tpatch;
};

// with_affine: the plan of a FORWARD call that carries an input transform (cstp_in_affine: the BatchNorm + ReLU in front applied in
// the kernel's staging).  The tuner times plain forwards, where the gather kernel igemm_k1s wins the 64-row temporal layers of the
// first stage (0.40 ms against 0.48 for igemm_k1w); WITH the transform the gather kernel pays it once per filter tap and cannot
// leave the next BatchNorm's sums (0.524 ms + 0.075 ms of bn_reduce against 0.523 ms, sums included; same-box step A/B 56.57 ->
// 55.97 ms, profiles/r04).  So such a call runs the weight-resident kernel wherever it applies (CSTP_K1W=0: the tuner's tile).
static bool k1w_preferred(const cstp_conv_desc& d, const Tile& t);
static bool make_plan(const cstp_conv_desc& d, ConvPlan& p, bool with_affine = false) {
  if (d.n <= 0 || d.c <= 0 || d.k <= 0 || d.d <= 0 || d.h <= 0 || d.w <= 0) return false;
  if (d.kt <= 0 || d.kh <= 0 || d.kw <= 0 || d.st <= 0 || d.sh <= 0 || d.sw <= 0) return false;
  if (d.pt < 0 || d.ph < 0 || d.pw < 0) return false;
  p.Do = (d.d + 2 * d.pt - d.kt) / d.st + 1;
  p.Ho = (d.h + 2 * d.ph - d.kh) / d.sh + 1;
  p.Wo = (d.w + 2 * d.pw - d.kw) / d.sw + 1;
  if (p.Do <= 0 || p.Ho <= 0 || p.Wo <= 0) return false;
  p.ntaps = d.kt * d.kh * d.kw;
  // forward: M = k, gather channels = c
  if (!lookup_tuned(d, 0, p.f_t)) p.f_t = pick_tile(d.k, (long)d.n * p.Do * p.Ho * p.Wo, 1);
  p.f_straddle = (d.c < 8);
  // the split kernels address their operands with 31-bit buffer offsets (bit 31 = "masked")
  const bool x_small = (size_t)d.n * d.c * d.d * d.h * d.w < (1ull << 29);
  const bool y_small = (size_t)d.n * d.k * p.Do * p.Ho * p.Wo < (1ull << 29);
  if (p.f_t.sp == 2 && !(x_small && y_small && patch_mt_ok(p.f_t.mt) && (patch_geom_ok(d) || tpatch_geom_ok(d)))) p.f_t = Tile{2, 1, 0, 1, 0};
  // (the 3-channel stems run a split tile in its straddle mode: zero-padded input copy, per-k offset table, f16 pair only)
  const bool stem_split_ok = p.f_straddle && p.f_t.sp == 1 && p.f_t.wm == 1 && (p.f_t.mt >= 4 && p.f_t.mt <= 6) &&
                             split_planes() == 2 && p.ntaps * d.c <= STR_KMAX - 16 &&
                             (size_t)d.n * d.c * (d.d + 2 * d.pt) * (d.h + 2 * d.ph) * (d.w + 2 * d.pw) < (1ull << 29);
  if (p.f_t.sp == 1 && !stem_split_ok &&
      (p.f_straddle || !x_small || p.ntaps > 27 || !split_tile_ok(p.f_t) || native_only()))
    p.f_t = Tile{2, 1, 0, 1, 0};
  if (p.f_t.sp == 1 && stem_split_ok && (!y_small || native_only())) p.f_t = Tile{2, 1, 0, 1, 0};
  if (with_affine && x_small && y_small && k1w_preferred(d, p.f_t)) p.f_t = Tile{4, 2, 0, 1, 2};
  p.f_Cp = p.f_straddle ? d.c : (int)align_up(d.c, 16);
  p.f_Kp = (int)align_up((size_t)p.ntaps * p.f_Cp, 16);
  p.f_Mp = cdiv(d.k, tile_bm(p.f_t)) * tile_bm(p.f_t);
  // dgrad: M = c, gather channels = k
  if (!lookup_tuned(d, 1, p.d_t))
    p.d_t = pick_tile(d.c, (long)d.n * cdiv(d.d, d.st) * cdiv(d.h, d.sh) * cdiv(d.w, d.sw), d.st * d.sh * d.sw);
  if (p.d_t.sp == 2 && !(x_small && y_small && patch_mt_ok(p.d_t.mt) && (patch_geom_ok(d) || tpatch_geom_ok(d)))) p.d_t = Tile{2, 1, 0, 1, 0};
  if (p.d_t.sp == 1 && (!y_small || p.ntaps > 27 || !split_tile_ok(p.d_t) || native_only())) p.d_t = Tile{2, 1, 0, 1, 0};
  p.d_Cp = (int)align_up(d.k, 16);
  p.d_Kp = p.ntaps * p.d_Cp;
  p.d_Mp = cdiv(d.c, tile_bm(p.d_t)) * tile_bm(p.d_t);
  // wgrad: M = k, J = (tap, c)
  p.w_straddle = (d.c < 8);
  p.w_mt = (CSTP_M16 && !p.w_straddle && d.k > 128 && d.k <= 144) ? 9 : pick_mt(d.k);   // 9 = nine 16-row tiles
  p.w_blocks = 2048;   // ~8 blocks per CU: measured 12 % faster than 4 per CU over the R18 layer set
  p.w_split = false;
  p.w_patch = false;
  p.w_tpatch = false;
  {
    Tile wt;
    bool have_wt = lookup_tuned(d, 2, wt);
    // developer override: CSTP_WTILE="s<mt>,<blocks/256>" forces the split weight-gradient kernel; unset in production
    static const char* wov = getenv("CSTP_WTILE");
    if (!have_wt && wov != nullptr && wov[0] == 'p') { wt = Tile{9, 1, 0, 0, 2}; have_wt = true; }
    if (!have_wt && wov != nullptr && wov[0] == 's') {
      int mt = 0, bl = 8;
      if (sscanf(wov + 1, "%d,%d", &mt, &bl) >= 1 && (mt == 4 || mt == 8 || mt == 9)) { wt = Tile{mt, bl, 0, 0, 1}; have_wt = true; }
    }
    if (have_wt && wt.sp == 2) {       // igemm_k2p: x resident in LDS across the nine taps (144-row blocks, f16 pair)
      p.w_patch = !p.w_straddle && x_small && y_small && wpatch_geom_ok(d);
      p.w_tpatch = !p.w_straddle && x_small && y_small && twpatch_geom_ok(d);       // igemm_k2t: the temporal layers' stream kernel
      have_wt = false;
    }
    if (have_wt) {
      p.w_mt = wt.m16 ? 9 : wt.mt;
      p.w_blocks = 256 * wt.wm;
      // igemm_k2s (3xbf16 split): 128- or 144-row tiles, 31-bit buffer offsets
      // (the 3-channel stems: igemm_k2s<.., STR> over the zero-padded input copy, f16 pair only)
      const bool stem_ok = split_planes() == 2 && p.ntaps * d.c <= STR_KMAX - 16 &&
                           (size_t)d.n * d.c * (d.d + 2 * d.pt) * (d.h + 2 * d.ph) * (d.w + 2 * d.pw) < (1ull << 29);
      p.w_split = wt.sp && !native_only() && (!p.w_straddle || stem_ok) && x_small && y_small && (wt.mt == 4 || wt.mt == 8 || wt.mt == 9);
      if (wt.sp && !p.w_split) p.w_mt = pick_mt(d.k);
    }
  }
  p.w_Cp = p.w_straddle ? d.c : (int)align_up(d.c, 32);
  p.w_Jtot = p.ntaps * p.w_Cp;
  p.w_Jp = (int)align_up(p.w_Jtot, 32);
  return true;
}

static bool k1w_preferred(const cstp_conv_desc& d, const Tile& t) {
  static const bool on = [] { const char* e = getenv("CSTP_K1W"); return e == nullptr || atoi(e) != 0; }();
  if (!on || g_force_tile != nullptr) return false;                 // (a pinned / timed tile is run as given)
  if (!((t.sp == 1 || t.sp == 2) && t.mt == 4)) return false;      // the 64-row tiles of the gather / ring kernels
  return tpatch_geom_ok(d) && k1w_fits(d.k, d.c) && d.c <= KW_AFFC;
}

// planes per operand of the split kernels: 2 = f16 pair / three products (default), 3 = bf16 triple / six products
// (CSTP_GEMM=bf16x3); CSTP_GEMM=f32 keeps every GEMM on the native f32 MFMA kernels
static std::atomic<int> g_split_terms{0};          // 0 = not overridden (cstp_gemm_set_split_terms); 1 = native f32 only
static int split_planes() {
  static const int env_np = [] {
    const char* e = getenv("CSTP_GEMM");
    return (e != nullptr && strcmp(e, "bf16x3") == 0) ? 3 : 2;
  }();
  const int o = g_split_terms.load(std::memory_order_relaxed);
  return o >= 2 ? o : env_np;
}
static bool native_only() {
  static const bool env_f32 = [] {
    const char* e = getenv("CSTP_GEMM");
    return e != nullptr && strcmp(e, "f32") == 0;
  }();
  return env_f32 || g_split_terms.load(std::memory_order_relaxed) == 1;
}
// Fully connected layers on <= 32 rows (linear.h): exact fp32 FMAs on weight-streaming kernels instead of a one-tile GEMM.
// CSTP_LINEAR=0 keeps them on the convolution kernels; so do the native-f32 mode (every GEMM on v_mfma_f32 there) and a pinned /
// timed tile.
static bool linear_shape(const cstp_conv_desc& d) {
  static const bool on = [] { const char* e = getenv("CSTP_LINEAR"); return e == nullptr || atoi(e) != 0; }();
  if (!on || g_force_tile != nullptr || native_only()) return false;
  return d.d == 1 && d.h == 1 && d.w == 1 && d.kt == 1 && d.kh == 1 && d.kw == 1 && d.st == 1 && d.sh == 1 && d.sw == 1 &&
         d.pt == 0 && d.ph == 0 && d.pw == 0 && d.n <= LIN_NMAX;
}
// returns false when the call does not qualify (alignment, workspace): the caller falls through to the convolution kernels
static bool run_linear(hipStream_t s, const cstp_conv_desc& d, bool dgrad, const float* a, const float* w, const float* bias, float* out,
                       void* ws, size_t ws_bytes, bool accumulate) {
  const int R = dgrad ? d.k : d.c, J = dgrad ? d.c : d.k;            // reduction length, output features
  if ((R % LIN_SLICE) != 0 || ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(ws)) & 15) != 0) return false;
  const int S = R / LIN_SLICE;
  if ((size_t)S * d.n * J * sizeof(float) > ws_bytes) return false;
  float* part = reinterpret_cast<float*>(ws);
  const dim3 grid((unsigned)cdiv(J, 256), (unsigned)S);
  if (dgrad) {
    if (d.n <= 16) hipLaunchKernelGGL((linear_dgrad_part_kernel<16>), grid, dim3(256), 0, s, a, w, part, d.n, d.c, d.k);
    else hipLaunchKernelGGL((linear_dgrad_part_kernel<32>), grid, dim3(256), 0, s, a, w, part, d.n, d.c, d.k);
  } else {
    if (d.n <= 16) hipLaunchKernelGGL((linear_fwd_part_kernel<16>), grid, dim3(256), 0, s, a, w, part, d.n, d.c, d.k);
    else hipLaunchKernelGGL((linear_fwd_part_kernel<32>), grid, dim3(256), 0, s, a, w, part, d.n, d.c, d.k);
  }
  hipLaunchKernelGGL(linear_reduce_kernel, dim3((unsigned)cdiv(d.n * J, 256)), dim3(256), 0, s, part, bias, out, S, d.n, J, accumulate ? 1 : 0);
  return true;
}

// deterministic mode (cstp_set_deterministic / CSTP_DETERMINISTIC=1): weight gradients through a two-stage split-K reduction
// (every split writes its own slab, unpack sums them in order) instead of f32 atomics: bit-reproducible, for debugging
constexpr int DET_MAX_SPLITS = 32;
static std::atomic<int> g_deterministic{-1};
static bool deterministic() {
  int v = g_deterministic.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("CSTP_DETERMINISTIC");
    v = (e != nullptr && e[0] == '1') ? 1 : 0;
    g_deterministic.store(v, std::memory_order_relaxed);
  }
  return v != 0;
}
// tail of the workspace: [0, 256) absmax cells of the activation operand(s), then the per-row inverse scales of the packed
// weights (<= max(k, c) + 160 rows)
static size_t plan_tail_bytes(const cstp_conv_desc& d) { return 256 + align_up((size_t)((d.k > d.c ? d.k : d.c) + 160) * 4, 256); }
static size_t plan_main_bytes(const cstp_conv_desc& d, const ConvPlan& p);
static size_t plan_ws_bytes(const cstp_conv_desc& d, const ConvPlan& p) { return plan_main_bytes(d, p) + plan_tail_bytes(d); }
static size_t plan_main_bytes(const cstp_conv_desc& d, const ConvPlan& p) {
  // packed-operand rows are padded to the tile height (<= 160): size for the tallest padding so that any tile
  // (heuristic or tuned later) fits the workspace the caller sized once
  // (the split kernels' packed operand is three bf16 planes = 6 bytes per element)
  size_t f = (size_t)p.f_Kp * ((size_t)d.k + 160) * 6, g = (size_t)p.d_Kp * ((size_t)d.c + 160) * 6;
  // the patch kernel's packed operand: 32-channel blocks x 9 taps x (rows padded to <= 144) x 128 bytes
  if (d.kt == 1 && d.kh == 3 && d.kw == 3) {
    const size_t pf = (size_t)cdiv(d.c, 32) * 9 * ((size_t)d.k + 144) * 128, pg = (size_t)cdiv(d.k, 32) * 9 * ((size_t)d.c + 144) * 128;
    if (pf > f) f = pf;
    if (pg > g) g = pg;
  }
  // the stems' split path keeps a zero-padded copy of the input behind the packed weights
  if (d.c < 8) f = align_up(f, 256) + align_up((size_t)d.n * d.c * (d.d + 2 * d.pt) * (d.h + 2 * d.ph) * (d.w + 2 * d.pw) * 4, 256);
  size_t w = align_up((size_t)d.k * p.w_Jp * sizeof(float), 256) * (deterministic() ? DET_MAX_SPLITS : 1);
  // (the stems' split weight gradient keeps its zero-padded input copy behind the slab(s) and the absmax cells)
  if (d.c < 8) w += 256 + align_up((size_t)d.n * d.c * (d.d + 2 * d.pt) * (d.h + 2 * d.ph) * (d.w + 2 * d.pw) * 4, 256);
  size_t m = f > g ? f : g;
  if (w > m) m = w;
  return align_up(m, 256);
}

template <bool DGRAD, bool STRADDLE, bool XFORM, int TPB>
static void launch_k1_t(Tile tl, dim3 grid, hipStream_t s, const Geom& g, const float* wp, const float* src,
                        const float* bias, float* out, int ntx, int ntm, const float2* in_ss, int in_npg, int in_relu) {
#define CSTP_K1(MT_, WM_)                                                                                      \
  hipLaunchKernelGGL((igemm_k1<MT_, WM_, DGRAD, STRADDLE, XFORM, false, TPB>), grid, dim3(256), 0, s, g, wp, src, bias, out, \
                     ntx, ntm, in_ss, in_npg, in_relu)
  if (tl.m16) {
    if (!STRADDLE)
      hipLaunchKernelGGL((igemm_k1<9, 1, DGRAD, false, XFORM, true, TPB>), grid, dim3(256), 0, s, g, wp, src, bias, out, ntx,
                         ntm, in_ss, in_npg, in_relu);
  } else if (tl.wm == 1) {
    switch (tl.mt) {
      case 1: CSTP_K1(1, 1); break;
      case 2: CSTP_K1(2, 1); break;
      case 3: CSTP_K1(3, 1); break;
      case 4: CSTP_K1(4, 1); break;
      default: CSTP_K1(5, 1); break;
    }
  } else if (tl.wm == 2) {
    if (tl.mt == 1) CSTP_K1(1, 2); else CSTP_K1(2, 2);
  } else {
    CSTP_K1(1, 4);
  }
#undef CSTP_K1
}

template <bool DGRAD, bool STRADDLE, bool XFORM>
static void launch_k1(Tile tl, dim3 grid, hipStream_t s, const Geom& g, const float* wp, const float* src,
                      const float* bias, float* out, int ntx, int ntm, const float2* in_ss, int in_npg, int in_relu) {
  // the two-tiles-per-barrier variant exists for the plain (non-stem, non-fused-BN) kernels only
  if (tl.tpb == 2 && !STRADDLE && !XFORM)
    launch_k1_t<DGRAD, false, false, 2>(tl, grid, s, g, wp, src, bias, out, ntx, ntm, in_ss, in_npg, in_relu);
  else
    launch_k1_t<DGRAD, STRADDLE, XFORM, 1>(tl, grid, s, g, wp, src, bias, out, ntx, ntm, in_ss, in_npg, in_relu);
}

template <bool STRADDLE, bool VEC4, int BKN, bool XFORM>
static void launch_k2(int mt, dim3 grid, hipStream_t s, const Geom& g, const float* dy, const float* x, float* dwp,
                      int Jtot, int Jp, int kt_total, int kt_per, int ntm, int ntj, int nsplit, const float2* in_ss,
                      int in_npg, int in_groups, int in_relu, size_t det_stride) {
#define CSTP_K2(MT_)                                                                                              \
  hipLaunchKernelGGL((igemm_k2<MT_, STRADDLE, VEC4, BKN, XFORM, false>), grid, dim3(256), 0, s, g, dy, x, dwp, Jtot, Jp, \
                     kt_total, kt_per, ntm, ntj, nsplit, in_ss, in_npg, in_groups, in_relu, det_stride)
  if (mt == 9) {   // 144-row tile on the 16x16x4 MFMA
    if (!STRADDLE)
      hipLaunchKernelGGL((igemm_k2<9, false, VEC4, BKN, XFORM, true>), grid, dim3(256), 0, s, g, dy, x, dwp, Jtot, Jp,
                         kt_total, kt_per, ntm, ntj, nsplit, in_ss, in_npg, in_groups, in_relu, det_stride);
    return;
  }
  switch (mt) {
    case 1: CSTP_K2(1); break;
    case 2: CSTP_K2(2); break;
    case 3: CSTP_K2(3); break;
    case 4: CSTP_K2(4); break;
    default: CSTP_K2(5); break;
  }
#undef CSTP_K2
}

static int pack_grid(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// ---- weight packs of a whole network pass from ONE launch (cstp_pack_mode / cstp_pack_recorded / cstp_pack_replay, cstp_hip.h).
// Every weight-pack launch of the convolution entry points goes through one of the three helpers below.  Per calling thread:
// mode 0 = launch it (default); mode 1 = launch it AND append what was launched to the thread's record list; mode 2 = skip it --
// the caller has replayed the recorded packs of this very call (same descriptor, same tile table, same weight and workspace
// pointers) earlier on the stream.  A record IS the launch (kernel kind, pointers, integer arguments, block count), so whatever
// variant the dispatch picked is what gets replayed.
static thread_local int tl_pack_mode = 0;
static thread_local std::vector<cstp_pack_rec> tl_pack_recs;
// ... and, without any per-call mode switch: workspaces the caller REGISTERED as holding replayed packs (cstp_pack_register) are
// never packed into by the calls that receive them (process-wide, mutex-guarded; a relaxed counter keeps the common empty case
// lock-free)
static std::mutex g_prepacked_mu;
static std::unordered_set<const void*> g_prepacked;
static std::atomic<int> g_prepacked_n{0};
bool pack_skip(const void* dst) {
  if (tl_pack_mode == 2) return true;
  if (tl_pack_mode == 1 || g_prepacked_n.load(std::memory_order_relaxed) == 0) return false;
  std::lock_guard<std::mutex> lk(g_prepacked_mu);
  return g_prepacked.count(dst) != 0;
}

void pack_record_b16(const float* w, void* dst, int nblocks, int kout, int cin, int ntaps, int Mp, int Kw, int dgrad) {
  if (tl_pack_mode != 1) return;
  cstp_pack_rec r{};
  r.kind = 4; r.nblocks = nblocks; r.w = w; r.dst = dst; r.inv_a = nullptr; r.cells = nullptr;
  const int a[6] = {kout, cin, ntaps, Mp, Kw, dgrad};
  memcpy(r.a, a, sizeof(a));
  tl_pack_recs.push_back(r);
}

static void pack_site_split2(hipStream_t s, const float* w, unsigned* wps, float* inv_a, unsigned* cells, int ncells, int kout,
                             int cin, int ntaps, int Cp, int Mp, int ngroups, int dgrad) {
  if (pack_skip(wps)) return;
  if (tl_pack_mode == 1) {
    cstp_pack_rec r{};
    r.kind = 1; r.nblocks = Mp; r.w = w; r.dst = wps; r.inv_a = inv_a; r.cells = cells;
    const int a[8] = {ncells, kout, cin, ntaps, Cp, Mp, ngroups, dgrad};
    memcpy(r.a, a, sizeof(a));
    tl_pack_recs.push_back(r);
  }
  hipLaunchKernelGGL(pack_weights_split2_kernel, dim3(Mp), dim3(256), 0, s, w, wps, inv_a, cells, ncells, kout, cin, ntaps, Cp, Mp,
                     ngroups, dgrad);
}
static void pack_site_patch(hipStream_t s, const float* w, uint4* wpk, float* inv_a, unsigned* cells, int ncells, int kout, int cin,
                            int ncb, int rows_per_blk, int nblk_rows, int dgrad, int nt) {
  if (pack_skip(wpk)) return;
  if (tl_pack_mode == 1) {
    cstp_pack_rec r{};
    r.kind = 2; r.nblocks = nblk_rows; r.w = w; r.dst = wpk; r.inv_a = inv_a; r.cells = cells;
    const int a[7] = {ncells, kout, cin, ncb, rows_per_blk, dgrad, nt};
    memcpy(r.a, a, sizeof(a));
    tl_pack_recs.push_back(r);
  }
  hipLaunchKernelGGL(pack_weights_patch_kernel, dim3(nblk_rows), dim3(256), 0, s, w, wpk, inv_a, cells, ncells, kout, cin, ncb,
                     rows_per_blk, dgrad, nt);
}
static void pack_site_native(hipStream_t s, const float* w, float* wp, int kout, int cin, int ntaps, int Cp, int Mp, int Kp, int dgrad) {
  if (pack_skip(wp)) return;
  const int nb = pack_grid((size_t)Kp * Mp);
  if (tl_pack_mode == 1) {
    cstp_pack_rec r{};
    r.kind = 3; r.nblocks = nb; r.w = w; r.dst = wp; r.inv_a = nullptr; r.cells = nullptr;
    const int a[7] = {kout, cin, ntaps, Cp, Mp, Kp, dgrad};
    memcpy(r.a, a, sizeof(a));
    tl_pack_recs.push_back(r);
  }
  hipLaunchKernelGGL(pack_weights_kernel, dim3(nb), dim3(256), 0, s, w, wp, kout, cin, ntaps, Cp, Mp, Kp, dgrad);
}

// block b runs record r with first[r] <= b < first[r + 1] as that record's block b - first[r]
__global__ void __launch_bounds__(256)
pack_replay_kernel(const cstp_pack_rec* __restrict__ recs, const int* __restrict__ first, int n) {
  int lo = 0, hi = n - 1;
  const int b = (int)blockIdx.x;
  while (lo < hi) {                                    // (uniform: scalar loads)
    const int mid = (lo + hi + 1) >> 1;
    if (first[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const cstp_pack_rec R = recs[lo];
  const int lb = b - first[lo];
  const int* a = R.a;
  if (R.kind == 1)
    pack_split2_body<8>(R.w, reinterpret_cast<unsigned*>(R.dst), R.inv_a, R.cells, a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], lb);
  else if (R.kind == 2)
    pack_patch_body(R.w, reinterpret_cast<uint4*>(R.dst), R.inv_a, R.cells, a[0], a[1], a[2], a[3], a[4], a[5], a[6], lb);
  else if (R.kind == 3)
    pack_native_body(R.w, reinterpret_cast<float*>(R.dst), a[0], a[1], a[2], a[3], a[4], a[5], a[6], lb, R.nblocks);
  else
    pack_w_b16_body(R.w, reinterpret_cast<unsigned short*>(R.dst), a[0], a[1], a[2], a[3], a[4], a[5], lb, R.nblocks);
}

// optional fused input transform of a convolution (see cstp_in_affine in cstp_hip.h); npg = clips per BatchNorm group
struct InAffine { const float2* ss; int npg, groups, relu; };

template <bool DGRAD, int NP, bool AFF = false>
static void launch_k1s_np(const Tile& tl, dim3 grid, hipStream_t s, const Geom& g, const uint4* wps, const float* src,
                          const float* bias, float* out, int ntx, int ntm, const float* inv_a, const unsigned* bcell,
                          const InAffine* ia = nullptr) {
  const float2* aff_ss = AFF ? ia->ss : nullptr;
  const int aff_gpos = AFF ? ia->npg * g.Dp * g.Hp * g.Wp : 1, aff_relu = AFF ? ia->relu : 0;
#define CSTP_K1S(MT_, NH_) \
  hipLaunchKernelGGL((igemm_k1s<MT_, DGRAD, NH_, NP, false, AFF>), grid, dim3(512), 0, s, g, wps, src, bias, out, ntx, ntm, inv_a, \
                     bcell, aff_ss, aff_gpos, aff_relu)
  if (tl.wm == 2) {             // 256-column tiles: only the tall row tiles (registers: 16*MT*4 accumulators per lane)
    if (tl.mt == 8) CSTP_K1S(8, 2); else CSTP_K1S(9, 2);
    return;
  }
  switch (tl.mt) {
    case 2: CSTP_K1S(2, 1); break;
    case 3: CSTP_K1S(3, 1); break;
    case 4: CSTP_K1S(4, 1); break;
    case 5: CSTP_K1S(5, 1); break;
    case 6: CSTP_K1S(6, 1); break;
    case 8: CSTP_K1S(8, 1); break;
    default: CSTP_K1S(9, 1); break;
  }
#undef CSTP_K1S
}

static int absmax_grid(size_t n) {
  size_t b = (n / 4 + 1023) / 1024;                   // >= 4 uint4 per thread
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

// pack the weights for the split kernel, make sure the gathered tensor's absmax cell is filled (NP == 2), launch it
template <bool DGRAD>
static void run_k1s(const Tile& tl, dim3 grid, hipStream_t s, const Geom& g, const cstp_conv_desc& d, int ntaps, int Kp,
                    const float* w, const float* src, size_t src_elems, const float* bias, float* out, int ntx, int ntm,
                    void* ws, size_t main_bytes, const uint32_t* src_absmax, const InAffine* ia = nullptr) {
  if (split_planes() == 3) {
    const size_t tot = (size_t)Kp * g.Mp;
    hipLaunchKernelGGL(pack_weights_split_kernel, dim3(pack_grid(tot / 2)), dim3(256), 0, s, w,
                       reinterpret_cast<unsigned short*>(ws), d.k, d.c, ntaps, g.Cp, g.Mp, Kp / 16, DGRAD ? 1 : 0);
    launch_k1s_np<DGRAD, 3>(tl, grid, s, g, reinterpret_cast<const uint4*>(ws), src, bias, out, ntx, ntm, nullptr, nullptr);
    return;
  }
  unsigned* cells = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + main_bytes);
  float* inv_a = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + main_bytes + 256);
  pack_site_split2(s, w, reinterpret_cast<unsigned*>(ws), inv_a, cells, 1, d.k, d.c, ntaps, g.Cp, g.Mp, Kp / 16, DGRAD ? 1 : 0);
  if constexpr (!DGRAD) {
    if (ia != nullptr && ia->ss != nullptr) {         // the gathered tensor is act(src * scale + shift); src_absmax is ITS maximum (checked by the caller)
      launch_k1s_np<false, 2, true>(tl, grid, s, g, reinterpret_cast<const uint4*>(ws), src, bias, out, ntx, ntm, inv_a, src_absmax, ia);
      return;
    }
  }
  if (src_absmax == nullptr) {
    // (the pack kernel zeroes the cells; when the caller replayed the packs -- possibly once for several calls on this
    //  workspace -- the cell is zeroed here, so that it holds THIS call's operand maximum exactly)
    if (pack_skip(ws)) (void)hipMemsetAsync(cells, 0, 8, s);
    hipLaunchKernelGGL(absmax_kernel, dim3(absmax_grid(src_elems)), dim3(256), 0, s, src, src_elems, cells);
  }
  launch_k1s_np<DGRAD, 2>(tl, grid, s, g, reinterpret_cast<const uint4*>(ws), src, bias, out, ntx, ntm, inv_a,
                          src_absmax != nullptr ? src_absmax : cells);
}

// the 3-channel stems on the f16-pair split kernel (igemm_k1s<.., STR = true>): zero-padded input copy (+ its absmax as a
// by-product), weights packed in k = tap * c_in + c order, per-k offset table inside the kernel
static void run_k1s_stem(const Tile& tl, hipStream_t s, const cstp_conv_desc& d, const ConvPlan& p, const float* w, const float* x,
                         const float* bias, float* y, void* ws, size_t main_bytes) {
  const int Dq = d.d + 2 * d.pt, Hq = d.h + 2 * d.ph, Wq = d.w + 2 * d.pw;
  const int kreal = p.ntaps * d.c, Kp = (int)align_up(kreal, 16);
  const int bm = 16 * tl.mt, Mp = cdiv(d.k, bm) * bm;
  const size_t packed = align_up((size_t)Kp * ((size_t)d.k + 160) * 6, 256);
  float* xp = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + packed);
  unsigned* cells = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + main_bytes);
  float* inv_a = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + main_bytes + 256);
  pack_site_split2(s, w, reinterpret_cast<unsigned*>(ws), inv_a, cells, 1, d.k, d.c, p.ntaps, d.c, Mp, Kp / 16, 0);
  {
    const int nrows = d.n * d.c * Dq * Hq;
    const int pgrid = nrows / 4 < 2048 ? (nrows + 3) / 4 : 2048;
    if (pack_skip(ws)) (void)hipMemsetAsync(cells, 0, 8, s);       // (as above: the pad kernel takes the maximum into the cell)
    hipLaunchKernelGGL(pad_input_kernel, dim3(pgrid), dim3(256), 0, s, x, xp, cells, d.n * d.c, d.d, d.h, d.w, d.pt, d.ph, d.pw);
  }
  Geom g;
  g.Cs = d.c; g.Ds = Dq; g.Hs = Hq; g.Ws = Wq;
  g.Nb = d.n; g.Dp = p.Do; g.Hp = p.Ho; g.Wp = p.Wo;
  g.kt = d.kt; g.kh = d.kh; g.kw = d.kw; g.st = d.st; g.sh = d.sh; g.sw = d.sw; g.pt = 0; g.ph = 0; g.pw = 0;
  g.Cp = d.c; g.M = d.k; g.Mp = Mp; g.Ktot = Kp;
  const int npos = d.n * p.Do * p.Ho * p.Wo;
  const int ntx = cdiv(npos, 128), ntm = cdiv(d.k, bm);
  dim3 grid((unsigned)(align_up(ntx, 8) * ntm), 1, 1);
#define CSTP_K1S_STR(MT_) \
  hipLaunchKernelGGL((igemm_k1s<MT_, false, 1, 2, true>), grid, dim3(512), 0, s, g, reinterpret_cast<const uint4*>(ws), xp, bias, \
                     y, ntx, ntm, inv_a, cells, (const float2*)nullptr, 1, 0)
  if (tl.mt == 4) CSTP_K1S_STR(4); else if (tl.mt == 5) CSTP_K1S_STR(5); else CSTP_K1S_STR(6);
#undef CSTP_K1S_STR
}

// pack the weights for the patch kernel, make sure the gathered tensor's absmax cell is filled, launch it
// (forward: src = x, Cs = c, M = k;  data gradient: src = dy, Cs = k, M = c -- a 3x3 stride-1 convolution with mirrored taps)
// BatchNorm partial sums as a by-product of a forward patch launch (igemm_k1p<MT, true>): possible when every 224-position tile
// is full and lies inside one BN group, the output allows 16-byte stores, and a block meets one row block only
struct K1pStats { double* part; int groups; const float* pivot; unsigned* zcell; };
static int k1p_grid_slots(const cstp_conv_desc& d, int M, int mt, int* ntiles_out, int* nmblk_out, int groups = 1);
static int k1p_stats_nsplit(const Tile& tl, const cstp_conv_desc& d, int groups) {
  if (tl.sp != 2 || groups < 1 || groups > 2 || d.n % groups != 0) return 0;
  const long gpos = (long)(d.n / groups) * d.d * d.h * d.w;
  if (gpos % KP_NPOS != 0 || ((d.h * d.w) & 3) != 0) return 0;
  int ntiles, nmblk;
  const int slots = k1p_grid_slots(d, d.k, tl.mt, &ntiles, &nmblk, groups);
  // the slots of an XCD are dealt to the groups alternately; every block meets one (group, row block) only
  if (slots == 0 || slots % (groups * nmblk) != 0) return 0;
  return 8 * (slots / groups) / nmblk;
}

// the temporal layers: igemm_k1t (same packed-weight format with three taps, same grid, same partial-sum table)
static void run_k1t(const Tile& tl, hipStream_t s, const cstp_conv_desc& d, bool dgrad, const float* w, const float* src,
                    float* out, void* ws, size_t main_bytes, const uint32_t* src_absmax, const K1pStats* st, bool accumulate,
                    const InAffine* ia) {
  TGeom g{};
  g.Cs = dgrad ? d.k : d.c;
  g.ncb = cdiv(g.Cs, 32);
  g.D = d.d; g.HW = d.h * d.w; g.Nb = d.n;
  g.M = dgrad ? d.c : d.k;
  g.nwt = g.HW / KT_WT; g.ndt = d.d / KT_DT;
  g.groups = st ? st->groups : 1;
  g.gclips = st ? d.n / st->groups : 1;
  g.acc = accumulate ? 1 : 0;
  g.aff_npg = ia ? ia->npg : 1; g.aff_groups = ia ? ia->groups : 1; g.aff_relu = ia ? ia->relu : 0;
  const int bm = 16 * tl.mt, nmblk = cdiv(g.M, bm);
  const int ntiles = d.n * g.ndt * g.nwt;
  unsigned* cells = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + main_bytes);
  float* inv_a = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + main_bytes + 256);
  pack_site_patch(s, w, reinterpret_cast<uint4*>(ws), inv_a, cells, 1, d.k, d.c, g.ncb, bm, nmblk * bm, dgrad ? 1 : 0, 3);
  const size_t src_elems = (size_t)d.n * g.Cs * d.d * d.h * d.w;
  if (src_absmax == nullptr) {
    // (the pack kernel zeroes the cells; when the caller replayed the packs -- possibly once for several calls on this
    //  workspace -- the cell is zeroed here, so that it holds THIS call's operand maximum exactly)
    if (pack_skip(ws)) (void)hipMemsetAsync(cells, 0, 8, s);
    hipLaunchKernelGGL(absmax_kernel, dim3(absmax_grid(src_elems)), dim3(256), 0, s, src, src_elems, cells);
  }
  const unsigned* bcell = src_absmax != nullptr ? src_absmax : cells;
  const int slots = k1p_grid_slots(d, g.M, tl.mt, nullptr, nullptr, st ? st->groups : 1);
  dim3 grid((unsigned)(8 * slots), 1, 1);
  double* part = st ? st->part : nullptr;
  const float* pivot = st ? st->pivot : nullptr;
  unsigned* zcell = st ? st->zcell : nullptr;
  const float2* ss = ia ? ia->ss : nullptr;
  // the 64-row forward layers whose packed weights fit LDS whole: igemm_k1w (igemm_twres.h), tile {4, wm = 2, sp = 2}
  if (tl.wm == 2 && tl.mt == 4 && !dgrad && !accumulate && k1w_fits(g.M, g.Cs) && (ss == nullptr || g.Cs <= KW_AFFC)) {
#define CSTP_K1W_(ST_, AF_) \
  hipLaunchKernelGGL((igemm_k1w<ST_, AF_>), grid, dim3(512), 0, s, g, reinterpret_cast<const uint4*>(ws), src, out, inv_a, bcell, ntiles, part, pivot, zcell, ss)
    if (part != nullptr && ss != nullptr) CSTP_K1W_(true, true);
    else if (part != nullptr) CSTP_K1W_(true, false);
    else if (ss != nullptr) CSTP_K1W_(false, true);
    else CSTP_K1W_(false, false);
#undef CSTP_K1W_
    return;
  }
#define CSTP_K1T_(MT_, ST_, AF_) \
  hipLaunchKernelGGL((igemm_k1t<MT_, ST_, AF_>), grid, dim3(512), 0, s, g, reinterpret_cast<const uint4*>(ws), src, out, inv_a, bcell, ntiles, nmblk, part, pivot, zcell, ss)
#define CSTP_K1T(MT_) \
  do { \
    if (part != nullptr && ss != nullptr) CSTP_K1T_(MT_, true, true); \
    else if (part != nullptr) CSTP_K1T_(MT_, true, false); \
    else if (ss != nullptr) CSTP_K1T_(MT_, false, true); \
    else CSTP_K1T_(MT_, false, false); \
  } while (0)
  if (tl.mt == 4) CSTP_K1T(4); else if (tl.mt == 8) CSTP_K1T(8); else CSTP_K1T(9);
#undef CSTP_K1T
#undef CSTP_K1T_
}

static void run_k1p(const Tile& tl, hipStream_t s, const cstp_conv_desc& d, bool dgrad, const float* w, const float* src,
                    float* out, void* ws, size_t main_bytes, const uint32_t* src_absmax, const K1pStats* st = nullptr,
                    bool accumulate = false, const InAffine* ia = nullptr) {
  if (tpatch_geom_ok(d)) { run_k1t(tl, s, d, dgrad, w, src, out, ws, main_bytes, src_absmax, st, accumulate, ia); return; }
  PGeom g;
  g.acc = accumulate ? 1 : 0;
  g.gpos = st ? (int)((long)(d.n / st->groups) * d.d * d.h * d.w) : 1;
  g.groups = st ? st->groups : 1;
  g.Cs = dgrad ? d.k : d.c;
  g.ncb = cdiv(g.Cs, 32);
  g.H = d.h; g.W = d.w; g.D = d.d; g.NF = d.n * d.d;
  g.M = dgrad ? d.c : d.k;
  g.rows_lds = patch_rows_needed(g.NF, g.H, g.W);
  {
    // staging by 16-byte loads (igemm_patch.h, QUAD): whole column quads, whole 8-channel groups, aligned lines, three rounds
    static const bool quad_on = [] { const char* e = getenv("CSTP_K1P_QUAD"); return e == nullptr || atoi(e) != 0; }();
    const int lines = g.rows_lds / (g.W + 2);
    g.quad = quad_on && (g.W & 3) == 0 && (g.Cs & 7) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 &&
             lines * (g.W / 4) * 2 <= 192 && g.rows_lds + 32 <= KP_ROWS ? 1 : 0;
  }
  const int bm = 16 * tl.mt, nmblk = cdiv(g.M, bm);
  const long P = (long)g.NF * g.H * g.W;
  const int ntiles = (int)((P + KP_NPOS - 1) / KP_NPOS);
  unsigned* cells = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + main_bytes);
  float* inv_a = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + main_bytes + 256);
  pack_site_patch(s, w, reinterpret_cast<uint4*>(ws), inv_a, cells, 1, d.k, d.c, g.ncb, bm, nmblk * bm, dgrad ? 1 : 0, 9);
  const size_t src_elems = (size_t)d.n * g.Cs * d.d * d.h * d.w;
  if (src_absmax == nullptr) {
    // (the pack kernel zeroes the cells; when the caller replayed the packs -- possibly once for several calls on this
    //  workspace -- the cell is zeroed here, so that it holds THIS call's operand maximum exactly)
    if (pack_skip(ws)) (void)hipMemsetAsync(cells, 0, 8, s);
    hipLaunchKernelGGL(absmax_kernel, dim3(absmax_grid(src_elems)), dim3(256), 0, s, src, src_elems, cells);
  }
  const unsigned* bcell = src_absmax != nullptr ? src_absmax : cells;
  const int slots = k1p_grid_slots(d, g.M, tl.mt, nullptr, nullptr, st ? st->groups : 1);
  dim3 grid((unsigned)(8 * slots), 1, 1);
  double* part = st ? st->part : nullptr;
  const float* pivot = st ? st->pivot : nullptr;
  unsigned* zcell = st ? st->zcell : nullptr;
#define CSTP_K1P(MT_) \
  do { \
    if (part != nullptr) hipLaunchKernelGGL((igemm_k1p<MT_, true>), grid, dim3(512), 0, s, g, reinterpret_cast<const uint4*>(ws), src, out, inv_a, bcell, ntiles, nmblk, part, pivot, zcell); \
    else hipLaunchKernelGGL((igemm_k1p<MT_, false>), grid, dim3(512), 0, s, g, reinterpret_cast<const uint4*>(ws), src, out, inv_a, bcell, ntiles, nmblk, part, pivot, zcell); \
  } while (0)
  if (tl.mt == 4) CSTP_K1P(4); else if (tl.mt == 8) CSTP_K1P(8); else CSTP_K1P(9);
#undef CSTP_K1P
}

// persistent blocks, one per CU (the LDS image fills it): 256 of them, or fewer when there is less work; returns slots per XCD
// (groups > 1: a BatchNorm-statistics launch over that many view groups -- the slots of an XCD are dealt to the groups in turn,
//  so their number is a multiple of groups * row blocks; 0 = no such grid)
static int k1p_grid_slots(const cstp_conv_desc& d, int M, int mt, int* ntiles_out, int* nmblk_out, int groups) {
  static const int n_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    // developer knob: CSTP_PERSIST_CUS=<n> caps the persistent kernels' grids (leaves CUs to kernels of the other stream)
    const char* e = getenv("CSTP_PERSIST_CUS");
    if (e != nullptr && atoi(e) >= 8 && atoi(e) < n) n = atoi(e);
    return n > 8 ? n / 8 * 8 : 8;
  }();
  const long P = (long)d.n * d.d * d.h * d.w;
  const int ntiles = (int)((P + KP_NPOS - 1) / KP_NPOS), nmblk = cdiv(M, 16 * mt);
  if (ntiles_out) *ntiles_out = ntiles;
  if (nmblk_out) *nmblk_out = nmblk;
  if (groups > 1) {
    const int per_xcd_g = cdiv(cdiv(ntiles, groups), 8) * nmblk * groups;
    const int slots_g = per_xcd_g < n_cu / 8 ? per_xcd_g : n_cu / 8;
    return slots_g / (groups * nmblk) * (groups * nmblk);
  }
  const int per_xcd = cdiv(ntiles, 8) * nmblk;        // items of the busiest XCD
  const int slots = per_xcd < n_cu / 8 ? per_xcd : n_cu / 8;
  return slots > 0 ? slots : 1;
}

static int cu_count() {
  static const int n_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    const char* e = getenv("CSTP_PERSIST_CUS");
    if (e != nullptr && atoi(e) >= 8 && atoi(e) < n) n = atoi(e);
    return n > 8 ? n / 8 * 8 : 8;
  }();
  return n_cu;
}

// weight gradient on the LDS-resident x ring (igemm_k2p): one block per (144-row block, 32-channel block, frame range);
// fills the packed slab(s) like igemm_k2s (scaled sums; unpack_wgrad_kernel applies the inverse operand scales)
static int run_k2p(hipStream_t s, const cstp_conv_desc& d, const ConvPlan& p, const float* x, const float* dy, float* dwp,
                   const unsigned* xcell, const unsigned* dycell, bool det, size_t det_stride, int* nslabs_out) {
  WPGeom g;
  g.C = d.c; g.M = d.k;
  g.ncb = cdiv(d.c, 32); g.nmblk = cdiv(d.k, WP_BM);
  g.H = d.h; g.W = d.w; g.D = d.d; g.NF = d.n * d.d;
  g.lead = cdiv(2 * (d.w + 2) + 66, 64);
  g.Jp = p.w_Jp; g.Cp = p.w_Cp;
  g.mg_pitch = (unsigned)((1ull << 32) / (unsigned)(d.w + 2) + 1);
  g.mg_hp1 = (unsigned)((1ull << 32) / (unsigned)(d.h + 1) + 1);
  g.mg_d = (unsigned)((1ull << 32) / (unsigned)d.d + 1);
  const int ncombo = g.ncb * g.nmblk;
  int ns = cu_count() / ncombo;
  if (ns < 1) ns = 1;
  if (ns > g.NF) ns = g.NF;
  if (det && ns > DET_MAX_SPLITS) ns = DET_MAX_SPLITS;
  g.fper = cdiv(g.NF, ns);
  g.nsplit = cdiv(g.NF, g.fper);
  dim3 grid((unsigned)align_up((size_t)ncombo * g.nsplit, 8), 1, 1);
  hipLaunchKernelGGL(igemm_k2p, grid, dim3(512), 0, s, g, dy, x, dwp, xcell, dycell, det ? det_stride : (size_t)0);
  *nslabs_out = det ? g.nsplit : 1;
  return 0;
}

// weight gradient of the temporal layers on streams of 32-position chunks (igemm_k2t): one block per (144 x-channels, 64 dY
// channels, item range); slab(s) as igemm_k2p
static int run_k2t(hipStream_t s, const cstp_conv_desc& d, const ConvPlan& p, const float* x, const float* dy, float* dwp,
                   const unsigned* xcell, const unsigned* dycell, bool det, size_t det_stride, int* nslabs_out,
                   const float2* ss, int aff_npg, int aff_groups, int aff_relu) {
  WTGeom g;
  g.C = d.c; g.M = d.k;
  g.nrb = cdiv(d.c, WP_BM); g.ncb = cdiv(d.k, 64);
  g.D = d.d; g.HW = d.h * d.w; g.Nb = d.n;
  const int chunk = (g.HW % 32 == 0) ? 32 : 16;
  g.nchunk = g.HW / chunk;
  g.nitems = d.n * g.nchunk;
  g.Jp = p.w_Jp; g.Cp = p.w_Cp;
  g.mg_fp1 = (unsigned)((1ull << 32) / (unsigned)(d.d + 1) + 1);
  g.mg_nchunk = (unsigned)((1ull << 32) / (unsigned)g.nchunk + 1);
  g.aff_npg = aff_npg; g.aff_groups = aff_groups; g.aff_relu = aff_relu;
  const int ncombo = g.nrb * g.ncb;
  int ns = cu_count() / ncombo;
  if (ns < 1) ns = 1;
  if (ns > g.nitems) ns = g.nitems;
  if (det && ns > DET_MAX_SPLITS) ns = DET_MAX_SPLITS;
  g.iper = cdiv(g.nitems, ns);
  g.nsplit = cdiv(g.nitems, g.iper);
  dim3 grid((unsigned)align_up((size_t)ncombo * g.nsplit, 8), 1, 1);
#define CSTP_K2T(AF_, CH_) hipLaunchKernelGGL((igemm_k2t<AF_, CH_>), grid, dim3(512), 0, s, g, dy, x, dwp, xcell, dycell, det ? det_stride : (size_t)0, ss)
  if (chunk == 32) { if (ss != nullptr) CSTP_K2T(true, 32); else CSTP_K2T(false, 32); }
  else { if (ss != nullptr) CSTP_K2T(true, 16); else CSTP_K2T(false, 16); }
#undef CSTP_K2T
  *nslabs_out = det ? g.nsplit : 1;
  return 0;
}

static int parse_in_affine(const cstp_in_affine* a, const cstp_conv_desc& d, InAffine& o) {
  o.ss = nullptr; o.npg = 1; o.groups = 1; o.relu = 0;
  if (a == nullptr || a->scale_shift == nullptr) return 0;
  if (a->groups < 1 || a->groups > 4 || (d.n % a->groups) != 0) return fail("in_affine: bad group count%s", "");
  if (d.c < 8) return fail("in_affine: not supported on the <8-channel (stem) path%s", "");
  o.ss = reinterpret_cast<const float2*>(a->scale_shift);
  o.groups = a->groups; o.npg = d.n / a->groups; o.relu = a->relu ? 1 : 0;
  return 0;
}

// Can a split (f16-pair gather) kernel apply the transform itself?  It needs the largest magnitude of the TRANSFORMED tensor
// (absmax: the caller's cell, cstp_bn_finalize_pre), whole 16-channel groups, and column / K tiles (cols positions) that never
// straddle two BatchNorm groups.
static bool aff_split_ok(const cstp_conv_desc& d, const InAffine& ia, const uint32_t* absmax, long out_positions_per_clip, int cols) {
  if (ia.ss == nullptr || absmax == nullptr || split_planes() != 2) return false;
  if ((d.c & 15) != 0 || ia.groups > 2) return false;
  const long gpos = (long)ia.npg * out_positions_per_clip;
  return gpos % cols == 0 && gpos < (1l << 30);
}

// ... or the temporal patch kernel (igemm_k1t<.., AFF>): tables of both groups in LDS, the transform once per staged element
static bool aff_tpatch_ok(const cstp_conv_desc& d, const Tile& t, const InAffine& ia, const uint32_t* absmax, const void* x, const void* y) {
  if (ia.ss == nullptr || absmax == nullptr || t.sp != 2 || !tpatch_geom_ok(d)) return false;
  if (ia.groups > 2 || d.c > KT_AFFC) return false;
  return ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
}

}  // namespace cstp

using namespace cstp;

extern "C" size_t cstp_conv3d_workspace_bytes(const cstp_conv_desc* desc) {
  ConvPlan p, pa;
  if (desc == nullptr || !make_plan(*desc, p) || !make_plan(*desc, pa, true)) return 0;
  const size_t a = plan_ws_bytes(*desc, p), b = plan_ws_bytes(*desc, pa);      // (a forward with an input transform may run another tile)
  return a > b ? a : b;
}

extern "C" int cstp_conv3d_forward(void* stream, const cstp_conv_desc* desc, const float* x, const float* w,
                                   const float* bias, const cstp_in_affine* in_affine, float* y, void* ws,
                                   size_t ws_bytes) {
  return cstp_conv3d_forward_am(stream, desc, x, w, bias, in_affine, y, ws, ws_bytes, nullptr);
}

extern "C" int32_t cstp_conv3d_in_affine_fused(const cstp_conv_desc* desc, int32_t groups) {
  ConvPlan p;
  if (desc == nullptr || !make_plan(*desc, p, true) || groups < 1 || desc->n % groups != 0) return 0;
  const cstp_conv_desc& d = *desc;
  InAffine ia{reinterpret_cast<const float2*>(desc), d.n / groups, groups, 1};       // (ss: any non-null pointer, never read here)
  const uint32_t* some = reinterpret_cast<const uint32_t*>(desc);
  const long S = (long)p.Do * p.Ho * p.Wo;
  const bool fwd = (p.f_t.sp == 1 && !p.f_straddle && aff_split_ok(d, ia, some, S, tile_bn(p.f_t))) ||
                   aff_tpatch_ok(d, p.f_t, ia, some, nullptr, nullptr);
  const bool wgr = (p.w_split && !p.w_straddle && !p.w_patch && !p.w_tpatch && aff_split_ok(d, ia, some, S, 32)) ||
                   (p.w_tpatch && groups <= 2);
  return fwd && wgr ? 1 : 0;
}

extern "C" int32_t cstp_conv3d_bnstats_nsplit(const cstp_conv_desc* desc, int32_t groups) {
  ConvPlan p;
  if (desc == nullptr || !make_plan(*desc, p)) return 0;
  return k1p_stats_nsplit(p.f_t, *desc, groups);
}

extern "C" int32_t cstp_conv3d_bnstats_nsplit_aff(const cstp_conv_desc* desc, int32_t groups) {
  ConvPlan p;
  if (desc == nullptr || !make_plan(*desc, p, true)) return 0;
  return k1p_stats_nsplit(p.f_t, *desc, groups);
}

extern "C" int cstp_conv3d_forward_bnstats(void* stream, const cstp_conv_desc* desc, const float* x, const float* w, float* y,
                                           void* ws, size_t ws_bytes, const uint32_t* x_absmax, int32_t groups, const float* pivot,
                                           double* part, size_t part_bytes, int32_t* nsplit_out, uint32_t* z_cell,
                                           const cstp_in_affine* in_affine) {
  CSTP_REQUIRE(desc && x && w && y && ws && part && nsplit_out, "null argument");
  ConvPlan p;
  CSTP_REQUIRE(make_plan(*desc, p, in_affine != nullptr && in_affine->scale_shift != nullptr), "invalid conv descriptor");
  InAffine ia;
  if (parse_in_affine(in_affine, *desc, ia)) return 1;
  const int ns = k1p_stats_nsplit(p.f_t, *desc, groups);
  *nsplit_out = 0;
  // this layer's kernel cannot deliver the sums (or cannot apply the input transform next to them): plain forward
  if (ns == 0 || ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(x)) & 15) != 0 ||
      (ia.ss != nullptr && !aff_tpatch_ok(*desc, p.f_t, ia, x_absmax, x, y)))
    return cstp_conv3d_forward_am(stream, desc, x, w, nullptr, in_affine, y, ws, ws_bytes, x_absmax);
  CSTP_REQUIRE(ws_bytes >= plan_ws_bytes(*desc, p), "workspace too small");
  CSTP_REQUIRE(part_bytes >= ((size_t)desc->k * groups * ns * 3 + desc->k) * sizeof(double), "partial-sum buffer too small");
  const cstp_conv_desc& d = *desc;
  CSTP_REQUIRE((size_t)d.n * d.c * d.d * d.h * d.w < (1ull << 30) && (size_t)d.n * d.k * p.Do * p.Ho * p.Wo < (1ull << 30),
               "tensor too large for 32-bit byte offsets (>= 4 GiB)");
  const K1pStats st{part, groups, pivot, z_cell};
  run_k1p(p.f_t, as_stream(stream), d, false, w, x, y, ws, plan_main_bytes(d, p), x_absmax, &st, false, ia.ss != nullptr ? &ia : nullptr);
  CSTP_LAUNCH_CHECK();
  *nsplit_out = ns;
  return 0;
}

extern "C" int cstp_conv3d_forward_am(void* stream, const cstp_conv_desc* desc, const float* x, const float* w,
                                      const float* bias, const cstp_in_affine* in_affine, float* y, void* ws,
                                      size_t ws_bytes, const uint32_t* x_absmax) {
  CSTP_REQUIRE(desc && x && w && y && ws, "null argument");
  ConvPlan p;
  CSTP_REQUIRE(make_plan(*desc, p, bias == nullptr && in_affine != nullptr && in_affine->scale_shift != nullptr), "invalid conv descriptor");
  CSTP_REQUIRE(ws_bytes >= plan_ws_bytes(*desc, p), "workspace too small");
  const cstp_conv_desc& d = *desc;
  CSTP_REQUIRE((size_t)d.n * d.c * d.d * d.h * d.w < (1ull << 30) && (size_t)d.n * d.k * p.Do * p.Ho * p.Wo < (1ull << 30),
               "tensor too large for 32-bit byte offsets (>= 4 GiB)");
  hipStream_t s = as_stream(stream);
  float* wp = reinterpret_cast<float*>(ws);
  InAffine ia;
  if (parse_in_affine(in_affine, d, ia)) return 1;
  if (ia.ss == nullptr && linear_shape(d) && run_linear(s, d, false, x, w, bias, y, ws, ws_bytes, false)) {
    CSTP_LAUNCH_CHECK();
    return 0;
  }
  // the fused input transform on the f16-pair gather kernel (igemm_k1s<.., AFF>): see aff_split_ok
  const bool aff_split = ia.ss != nullptr && p.f_t.sp == 1 && !p.f_straddle &&
                         aff_split_ok(d, ia, x_absmax, (long)p.Do * p.Ho * p.Wo, tile_bn(p.f_t));
  const bool aff_tpatch = bias == nullptr && aff_tpatch_ok(d, p.f_t, ia, x_absmax, x, y);
  // (the patch kernels store / load 16 bytes per lane: misaligned tensors take the gather kernel)
  if (p.f_t.sp == 2 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) != 0 && tpatch_geom_ok(d)) {
    p.f_t = Tile{9, 1, 0, 1, 1};
    p.f_Mp = cdiv(d.k, tile_bm(p.f_t)) * tile_bm(p.f_t);
  }
  if (p.f_t.sp != 0 && ia.ss != nullptr && !aff_split && !aff_tpatch) {
    // no fused input transform on this layer's split kernel: such a call runs a native tile (and its operand padding)
    p.f_t = pick_tile(d.k, (long)d.n * p.Do * p.Ho * p.Wo, 1);
    if (p.f_t.sp) p.f_t = Tile{2, 1, 0, 1, 0};
    p.f_Mp = cdiv(d.k, tile_bm(p.f_t)) * tile_bm(p.f_t);
  }
  if (p.f_t.sp == 2 && bias == nullptr) {
    run_k1p(p.f_t, s, d, false, w, x, y, ws, plan_main_bytes(d, p), x_absmax, nullptr, false, aff_tpatch ? &ia : nullptr);
    CSTP_LAUNCH_CHECK();
    return 0;
  }
  if (p.f_t.sp == 1 && p.f_straddle && !(in_affine != nullptr && in_affine->scale_shift != nullptr)) {
    run_k1s_stem(p.f_t, s, d, p, w, x, bias, y, ws, plan_main_bytes(d, p));
    CSTP_LAUNCH_CHECK();
    return 0;
  }
  if (p.f_t.sp == 2) {                               // (a bias rides only on the Linear layers: never a 3x3 geometry)
    p.f_t = Tile{9, 1, 0, 1, 1};
    if (!split_tile_ok(p.f_t) || p.f_straddle) p.f_t = Tile{2, 1, 0, 1, 0};
    p.f_Mp = cdiv(d.k, tile_bm(p.f_t)) * tile_bm(p.f_t);
  }
  const size_t tot = (size_t)p.f_Kp * p.f_Mp;
  const bool f_split = p.f_t.sp && !p.f_straddle;
  if (!f_split)
    pack_site_native(s, w, wp, d.k, d.c, p.ntaps, p.f_Cp, p.f_Mp, p.f_Kp, 0);
  Geom g;
  g.Cs = d.c; g.Ds = d.d; g.Hs = d.h; g.Ws = d.w;
  g.Nb = d.n; g.Dp = p.Do; g.Hp = p.Ho; g.Wp = p.Wo;
  g.kt = d.kt; g.kh = d.kh; g.kw = d.kw; g.st = d.st; g.sh = d.sh; g.sw = d.sw; g.pt = d.pt; g.ph = d.ph; g.pw = d.pw;
  g.Cp = p.f_Cp; g.M = d.k; g.Mp = p.f_Mp; g.Ktot = p.ntaps * p.f_Cp;
  const int npos = d.n * p.Do * p.Ho * p.Wo;
  const int f_bm = tile_bm(p.f_t), f_bn = tile_bn(p.f_t);
  const int ntx = cdiv(npos, f_bn), ntm = cdiv(d.k, f_bm);
  dim3 grid((unsigned)(align_up(ntx, 8) * ntm), 1, 1);
  if (f_split) run_k1s<false>(p.f_t, grid, s, g, d, p.ntaps, p.f_Kp, w, x, (size_t)d.n * d.c * d.d * d.h * d.w, bias, y, ntx, ntm,
                              ws, plan_main_bytes(d, p), x_absmax, aff_split ? &ia : nullptr);
  else if (p.f_straddle) launch_k1<false, true, false>(p.f_t, grid, s, g, wp, x, bias, y, ntx, ntm, nullptr, 1, 0);
  else if (ia.ss) launch_k1<false, false, true>(p.f_t, grid, s, g, wp, x, bias, y, ntx, ntm, ia.ss, ia.npg, ia.relu);
  else launch_k1<false, false, false>(p.f_t, grid, s, g, wp, x, bias, y, ntx, ntm, nullptr, 1, 0);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_conv3d_backward_data(void* stream, const cstp_conv_desc* desc, const float* dy, const float* w,
                                         float* dx, void* ws, size_t ws_bytes) {
  return cstp_conv3d_backward_data_am(stream, desc, dy, w, dx, ws, ws_bytes, nullptr);
}

extern "C" int cstp_conv3d_backward_data_am(void* stream, const cstp_conv_desc* desc, const float* dy, const float* w,
                                            float* dx, void* ws, size_t ws_bytes, const uint32_t* dy_absmax) {
  return cstp_conv3d_backward_data_acc(stream, desc, dy, w, dx, ws, ws_bytes, dy_absmax, 0);
}

extern "C" int cstp_conv3d_backward_data_acc(void* stream, const cstp_conv_desc* desc, const float* dy, const float* w,
                                             float* dx, void* ws, size_t ws_bytes, const uint32_t* dy_absmax,
                                             int32_t accumulate) {
  CSTP_REQUIRE(desc && dy && w && dx && ws, "null argument");
  ConvPlan p;
  CSTP_REQUIRE(make_plan(*desc, p), "invalid conv descriptor");
  CSTP_REQUIRE(ws_bytes >= plan_ws_bytes(*desc, p), "workspace too small");
  const cstp_conv_desc& d = *desc;
  CSTP_REQUIRE((size_t)d.n * d.c * d.d * d.h * d.w < (1ull << 30) && (size_t)d.n * d.k * p.Do * p.Ho * p.Wo < (1ull << 30),
               "tensor too large for 32-bit byte offsets (>= 4 GiB)");
  hipStream_t s = as_stream(stream);
  float* wp = reinterpret_cast<float*>(ws);
  if (linear_shape(d) && run_linear(s, d, true, dy, w, nullptr, dx, ws, ws_bytes, accumulate != 0)) {
    CSTP_LAUNCH_CHECK();
    return 0;
  }
  if (p.d_t.sp == 2 && tpatch_geom_ok(d) && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) != 0) {
    p.d_t = Tile{9, 1, 0, 1, 1};
    p.d_Mp = cdiv(d.c, tile_bm(p.d_t)) * tile_bm(p.d_t);
  }
  if (p.d_t.sp == 2) {
    run_k1p(p.d_t, s, d, true, w, dy, dx, ws, plan_main_bytes(d, p), dy_absmax, nullptr, accumulate != 0);
    CSTP_LAUNCH_CHECK();
    return 0;
  }
  const size_t tot = (size_t)p.d_Kp * p.d_Mp;
  const bool d_split = p.d_t.sp != 0;
  if (!d_split) pack_site_native(s, w, wp, d.k, d.c, p.ntaps, p.d_Cp, p.d_Mp, p.d_Kp, 1);
  Geom g;
  g.Cs = d.k; g.Ds = p.Do; g.Hs = p.Ho; g.Ws = p.Wo;     // gather from dy
  g.Nb = d.n; g.Dp = d.d; g.Hp = d.h; g.Wp = d.w;         // FULL x dims; classes subsample inside
  g.kt = d.kt; g.kh = d.kh; g.kw = d.kw; g.st = d.st; g.sh = d.sh; g.sw = d.sw; g.pt = d.pt; g.ph = d.ph; g.pw = d.pw;
  g.Cp = p.d_Cp; g.M = d.c; g.Mp = p.d_Mp; g.Ktot = p.ntaps * p.d_Cp;
  g.acc = accumulate ? 1 : 0;
  const int nclass = d.st * d.sh * d.sw;
  const int npos_max = d.n * cdiv(d.d, d.st) * cdiv(d.h, d.sh) * cdiv(d.w, d.sw);
  const int d_bm = tile_bm(p.d_t), d_bn = tile_bn(p.d_t);
  const int ntx = cdiv(npos_max, d_bn), ntm = cdiv(d.c, d_bm);
  dim3 grid((unsigned)(align_up(ntx, 8) * ntm), (unsigned)nclass, 1);
  if (p.ntaps == 1 && nclass > 1 && d.pt == 0 && d.ph == 0 && d.pw == 0) {
    // a strided pointwise layer (the shortcut's spatial half, r21d_byol.py:122-125): only stride class 0 holds a tap, the other
    // positions of dx are exact zeros -- one fill instead of three classes of strided zero stores (or nothing when accumulating)
    if (!accumulate) CSTP_REQUIRE(hipMemsetAsync(dx, 0, (size_t)d.n * d.c * d.d * d.h * d.w * sizeof(float), s) == hipSuccess, "hipMemsetAsync");
    grid.y = 1;
  }
  if (d_split) run_k1s<true>(p.d_t, grid, s, g, d, p.ntaps, p.d_Kp, w, dy, (size_t)d.n * d.k * p.Do * p.Ho * p.Wo, nullptr, dx, ntx,
                             ntm, ws, plan_main_bytes(d, p), dy_absmax);
  else launch_k1<true, false, false>(p.d_t, grid, s, g, wp, dy, nullptr, dx, ntx, ntm, nullptr, 1, 0);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_conv3d_backward_weight(void* stream, const cstp_conv_desc* desc, const float* x,
                                           const cstp_in_affine* in_affine, const float* dy, float* dw, void* ws,
                                           size_t ws_bytes) {
  return cstp_conv3d_backward_weight_am(stream, desc, x, in_affine, dy, dw, ws, ws_bytes, nullptr, nullptr);
}

extern "C" int cstp_conv3d_backward_weight_am(void* stream, const cstp_conv_desc* desc, const float* x,
                                              const cstp_in_affine* in_affine, const float* dy, float* dw, void* ws,
                                              size_t ws_bytes, const uint32_t* x_absmax, const uint32_t* dy_absmax) {
  return cstp_conv3d_backward_weight_acc(stream, desc, x, in_affine, dy, dw, ws, ws_bytes, x_absmax, dy_absmax, 0);
}

extern "C" int cstp_pack_mode(int32_t mode) {
  CSTP_REQUIRE(mode >= 0 && mode <= 2, "pack mode: 0 (pack inside the calls), 1 (... and record), 2 (skip: the caller replayed the packs)");
  tl_pack_mode = mode;
  return 0;
}

extern "C" int32_t cstp_pack_recorded(cstp_pack_rec* out, int32_t cap) {
  const int32_t n = (int32_t)tl_pack_recs.size();
  if (out != nullptr) {
    for (int32_t i = 0; i < n && i < cap; ++i) out[i] = tl_pack_recs[i];
    tl_pack_recs.clear();
  }
  return n;
}

extern "C" int cstp_pack_register(const void* const* workspaces, int32_t n, int32_t on) {
  CSTP_REQUIRE(n >= 0 && (n == 0 || workspaces != nullptr), "bad argument");
  std::lock_guard<std::mutex> lk(g_prepacked_mu);
  if (n == 0 && !on) g_prepacked.clear();
  for (int32_t i = 0; i < n; ++i) {
    if (on) g_prepacked.insert(workspaces[i]); else g_prepacked.erase(workspaces[i]);
  }
  g_prepacked_n.store((int)g_prepacked.size(), std::memory_order_relaxed);
  return 0;
}

extern "C" int cstp_pack_replay(void* stream, const cstp_pack_rec* recs_dev, const int32_t* first_block_dev, int32_t n,
                                int32_t total_blocks) {
  CSTP_REQUIRE(recs_dev && first_block_dev && n > 0 && total_blocks > 0, "bad argument");
  hipLaunchKernelGGL(pack_replay_kernel, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream), recs_dev, first_block_dev, n);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_set_deterministic(int32_t on) {
  g_deterministic.store(on ? 1 : 0, std::memory_order_relaxed);
  return 0;
}

extern "C" int32_t cstp_get_deterministic(void) { return deterministic() ? 1 : 0; }

extern "C" int cstp_conv3d_backward_weight_acc(void* stream, const cstp_conv_desc* desc, const float* x,
                                               const cstp_in_affine* in_affine, const float* dy, float* dw, void* ws,
                                               size_t ws_bytes, const uint32_t* x_absmax, const uint32_t* dy_absmax,
                                               int32_t accumulate) {
  CSTP_REQUIRE(desc && x && dy && dw && ws, "null argument");
  ConvPlan p;
  CSTP_REQUIRE(make_plan(*desc, p), "invalid conv descriptor");
  CSTP_REQUIRE(ws_bytes >= plan_ws_bytes(*desc, p), "workspace too small");
  const cstp_conv_desc& d = *desc;
  CSTP_REQUIRE((size_t)d.n * d.c * d.d * d.h * d.w < (1ull << 30) && (size_t)d.n * d.k * p.Do * p.Ho * p.Wo < (1ull << 30),
               "tensor too large for 32-bit byte offsets (>= 4 GiB)");
  hipStream_t s = as_stream(stream);
  float* dwp = reinterpret_cast<float*>(ws);
  const size_t slab = (size_t)d.k * p.w_Jp * sizeof(float);
  // the absmax cells of the 2xf16-split kernel sit right behind the slab (256-byte aligned) and are zeroed with it
  const size_t slab_al = align_up(slab, 256);
  const bool det = deterministic();
  Geom g;
  g.Cs = d.c; g.Ds = d.d; g.Hs = d.h; g.Ws = d.w;         // gather from x
  g.Nb = d.n; g.Dp = p.Do; g.Hp = p.Ho; g.Wp = p.Wo;      // reduction over dy positions
  g.kt = d.kt; g.kh = d.kh; g.kw = d.kw; g.st = d.st; g.sh = d.sh; g.sw = d.sw; g.pt = d.pt; g.ph = d.ph; g.pw = d.pw;
  g.Cp = p.w_Cp; g.M = d.k; g.Mp = 0; g.Ktot = p.w_Jtot;
  InAffine ia;
  if (parse_in_affine(in_affine, d, ia)) return 1;
  if (ia.ss == nullptr && linear_shape(d)) {
    const dim3 lgrid((unsigned)cdiv(d.c, 256), (unsigned)cdiv(d.k, 16));
    if (d.n <= 16) hipLaunchKernelGGL((linear_wgrad_kernel<16>), lgrid, dim3(256), 0, s, dy, x, dw, d.n, d.c, d.k, accumulate ? 1 : 0);
    else hipLaunchKernelGGL((linear_wgrad_kernel<32>), lgrid, dim3(256), 0, s, dy, x, dw, d.n, d.c, d.k, accumulate ? 1 : 0);
    CSTP_LAUNCH_CHECK();
    return 0;
  }
  if (p.w_patch && ia.ss == nullptr) {
    // igemm_k2p.  Slab(s) + absmax cells zeroed together; in deterministic mode one slab per frame-range split.
    const size_t det_stride_p = det ? slab_al / sizeof(float) : 0;
    int ns_max = cu_count() / (cdiv(d.c, 32) * cdiv(d.k, WP_BM));
    ns_max = ns_max < 1 ? 1 : (ns_max > DET_MAX_SPLITS ? DET_MAX_SPLITS : ns_max);
    const size_t slabs_bytes_p = slab_al * (det ? ns_max : 1);
    unsigned* cells_p = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + slabs_bytes_p);
    if (hipMemsetAsync(dwp, 0, slabs_bytes_p + 256, s) != hipSuccess) return fail("hipMemsetAsync failed%s", "");
    const size_t nx = (size_t)d.n * d.c * d.d * d.h * d.w, ny = (size_t)d.n * d.k * p.Do * p.Ho * p.Wo;
    if (x_absmax == nullptr) hipLaunchKernelGGL(absmax_kernel, dim3(absmax_grid(nx)), dim3(256), 0, s, x, nx, cells_p);
    if (dy_absmax == nullptr) hipLaunchKernelGGL(absmax_kernel, dim3(absmax_grid(ny)), dim3(256), 0, s, dy, ny, cells_p + 1);
    const unsigned* xc = x_absmax != nullptr ? x_absmax : cells_p;
    const unsigned* dyc = dy_absmax != nullptr ? dy_absmax : cells_p + 1;
    int nslabs = 1;
    run_k2p(s, d, p, x, dy, dwp, xc, dyc, det, det_stride_p, &nslabs);
    CSTP_LAUNCH_CHECK();
    const size_t tot_p = (size_t)d.k * d.c * p.ntaps;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(pack_grid(tot_p)), dim3(256), 0, s, dwp, dw, d.k, d.c, p.ntaps, p.w_Cp, p.w_Jp,
                       xc, dyc, nslabs, det_stride_p, accumulate ? 1 : 0);
    CSTP_LAUNCH_CHECK();
    return 0;
  }
  if (p.w_tpatch && (ia.ss == nullptr || (ia.groups <= 2 && x_absmax != nullptr))) {
    // igemm_k2t: the temporal layers, both operands staged once (AFF: the transform of x inside; x_absmax is then T(x)'s)
    const size_t det_stride_p = det ? slab_al / sizeof(float) : 0;
    int ns_max = cu_count() / (cdiv(d.c, WP_BM) * cdiv(d.k, 64));
    ns_max = ns_max < 1 ? 1 : (ns_max > DET_MAX_SPLITS ? DET_MAX_SPLITS : ns_max);
    const size_t slabs_bytes_p = slab_al * (det ? ns_max : 1);
    unsigned* cells_p = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + slabs_bytes_p);
    if (hipMemsetAsync(dwp, 0, slabs_bytes_p + 256, s) != hipSuccess) return fail("hipMemsetAsync failed%s", "");
    const size_t nx = (size_t)d.n * d.c * d.d * d.h * d.w, ny = (size_t)d.n * d.k * p.Do * p.Ho * p.Wo;
    if (x_absmax == nullptr) hipLaunchKernelGGL(absmax_kernel, dim3(absmax_grid(nx)), dim3(256), 0, s, x, nx, cells_p);
    if (dy_absmax == nullptr) hipLaunchKernelGGL(absmax_kernel, dim3(absmax_grid(ny)), dim3(256), 0, s, dy, ny, cells_p + 1);
    const unsigned* xc = x_absmax != nullptr ? x_absmax : cells_p;
    const unsigned* dyc = dy_absmax != nullptr ? dy_absmax : cells_p + 1;
    int nslabs = 1;
    run_k2t(s, d, p, x, dy, dwp, xc, dyc, det, det_stride_p, &nslabs, ia.ss, ia.npg, ia.groups, ia.relu);
    CSTP_LAUNCH_CHECK();
    const size_t tot_p = (size_t)d.k * d.c * p.ntaps;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(pack_grid(tot_p)), dim3(256), 0, s, dwp, dw, d.k, d.c, p.ntaps, p.w_Cp, p.w_Jp,
                       xc, dyc, nslabs, det_stride_p, accumulate ? 1 : 0);
    CSTP_LAUNCH_CHECK();
    return 0;
  }
  // the fused input transform: on the f16-pair gather kernel (igemm_k2s<.., AFF>) where aff_split_ok, else the native kernel
  // with its analytic tile
  const bool aff_split = ia.ss != nullptr && p.w_split && !p.w_straddle && aff_split_ok(d, ia, x_absmax, (long)p.Do * p.Ho * p.Wo, 32);
  const bool w_split = p.w_split && (ia.ss == nullptr || aff_split);
  if (p.w_split && !w_split) p.w_mt = (CSTP_M16 && d.k > 128 && d.k <= 144) ? 9 : pick_mt(d.k);
  const int npos = d.n * p.Do * p.Ho * p.Wo;
  const int bkn = CSTP_K2_BKN;
  const int kt_total = cdiv(npos, bkn);
  const int ntm = w_split ? cdiv(d.k, 16 * p.w_mt) : cdiv(d.k, p.w_mt == 9 ? 144 : 32 * p.w_mt), ntj = cdiv(p.w_Jtot, 128);
  int splits = cdiv(p.w_blocks, ntm * ntj);
  if (splits > cdiv(kt_total, 256 / bkn)) splits = cdiv(kt_total, 256 / bkn);
  if (det && splits > DET_MAX_SPLITS) splits = DET_MAX_SPLITS;
  if (splits < 1) splits = 1;
  const int kt_per = cdiv(kt_total, splits);
  splits = cdiv(kt_total, kt_per);
  // one slab filled by atomics, or (deterministic) one slab per split; the absmax cells sit in the workspace tail
  // (the absmax cells of the 2xf16-split kernel sit right behind the slab(s), 256-byte aligned, and are zeroed with them)
  const size_t det_stride = det ? slab_al / sizeof(float) : 0;
  const size_t slabs_bytes = slab_al * (det ? splits : 1);
  unsigned* cells = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + slabs_bytes);
  if (hipMemsetAsync(dwp, 0, slabs_bytes + 256, s) != hipSuccess) return fail("hipMemsetAsync failed%s", "");
  dim3 grid((unsigned)(align_up((size_t)splits * ntm, 8) * ntj), 1, 1);
  const bool v4 = ((p.Do * p.Ho * p.Wo) % 4) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0;
#define CSTP_K2_ARGS p.w_mt, grid, s, g, dy, x, dwp, p.w_Jtot, p.w_Jp, kt_total, kt_per, ntm, ntj, splits, ia.ss, ia.npg, ia.groups, ia.relu, det_stride
  const bool w_f16 = w_split && split_planes() == 2;
  const unsigned* xcell = x_absmax != nullptr ? x_absmax : cells;
  const unsigned* dycell = dy_absmax != nullptr ? dy_absmax : cells + 1;
  if (w_split && p.w_straddle) {
    // the stem: zero-padded copy of x behind the slab(s) + cells (its absmax is the pad kernel's by-product), columns (tap, c)
    const int Dq = d.d + 2 * d.pt, Hq = d.h + 2 * d.ph, Wq = d.w + 2 * d.pw;
    float* xp = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + slabs_bytes + 256);
    const int nrows = d.n * d.c * Dq * Hq;
    const int pgrid = nrows / 4 < 2048 ? (nrows + 3) / 4 : 2048;
    hipLaunchKernelGGL(pad_input_kernel, dim3(pgrid), dim3(256), 0, s, x, xp, cells, d.n * d.c, d.d, d.h, d.w, d.pt, d.ph, d.pw);
    const size_t ny = (size_t)d.n * d.k * p.Do * p.Ho * p.Wo;
    if (dy_absmax == nullptr) hipLaunchKernelGGL(absmax_kernel, dim3(absmax_grid(ny)), dim3(256), 0, s, dy, ny, cells + 1);
    Geom gs = g;
    gs.Ds = Dq; gs.Hs = Hq; gs.Ws = Wq; gs.pt = 0; gs.ph = 0; gs.pw = 0; gs.Cp = d.c;
#define CSTP_K2S_STR(MT_) \
  hipLaunchKernelGGL((igemm_k2s<MT_, 2, true>), grid, dim3(512), 0, s, gs, dy, xp, dwp, p.w_Jtot, p.w_Jp, kt_total, kt_per, ntm, ntj, splits, cells, dycell, det_stride, (const float2*)nullptr, 1, 1, 0)
    if (p.w_mt == 9) CSTP_K2S_STR(9); else if (p.w_mt == 4) CSTP_K2S_STR(4); else CSTP_K2S_STR(8);
#undef CSTP_K2S_STR
    CSTP_LAUNCH_CHECK();
    const size_t tot_s = (size_t)d.k * d.c * p.ntaps;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(pack_grid(tot_s)), dim3(256), 0, s, dwp, dw, d.k, d.c, p.ntaps, p.w_Cp, p.w_Jp,
                       cells, dycell, det ? splits : 1, det_stride, accumulate ? 1 : 0);
    CSTP_LAUNCH_CHECK();
    return 0;
  }
  if (w_split) {
#define CSTP_K2S(MT_, NP_) \
  hipLaunchKernelGGL((igemm_k2s<MT_, NP_>), grid, dim3(512), 0, s, g, dy, x, dwp, p.w_Jtot, p.w_Jp, kt_total, kt_per, ntm, ntj, splits, xcell, dycell, det_stride, (const float2*)nullptr, 1, 1, 0)
#define CSTP_K2S_AFF(MT_) \
  hipLaunchKernelGGL((igemm_k2s<MT_, 2, false, true>), grid, dim3(512), 0, s, g, dy, x, dwp, p.w_Jtot, p.w_Jp, kt_total, kt_per, ntm, ntj, splits, xcell, dycell, det_stride, ia.ss, ia.npg * p.Do * p.Ho * p.Wo, ia.groups, ia.relu)
    if (w_f16) {
      const size_t nx = (size_t)d.n * d.c * d.d * d.h * d.w, ny = (size_t)d.n * d.k * p.Do * p.Ho * p.Wo;
      if (x_absmax == nullptr) hipLaunchKernelGGL(absmax_kernel, dim3(absmax_grid(nx)), dim3(256), 0, s, x, nx, cells);
      if (dy_absmax == nullptr) hipLaunchKernelGGL(absmax_kernel, dim3(absmax_grid(ny)), dim3(256), 0, s, dy, ny, cells + 1);
      if (aff_split) { if (p.w_mt == 9) CSTP_K2S_AFF(9); else if (p.w_mt == 4) CSTP_K2S_AFF(4); else CSTP_K2S_AFF(8); }
      else if (p.w_mt == 9) CSTP_K2S(9, 2); else if (p.w_mt == 4) CSTP_K2S(4, 2); else CSTP_K2S(8, 2);
    } else {
      if (p.w_mt == 9) CSTP_K2S(9, 3); else if (p.w_mt == 4) CSTP_K2S(4, 3); else CSTP_K2S(8, 3);
    }
#undef CSTP_K2S
#undef CSTP_K2S_AFF
  } else if (p.w_straddle) {
    if (v4) launch_k2<true, true, CSTP_K2_BKN, false>(CSTP_K2_ARGS);
    else launch_k2<true, false, CSTP_K2_BKN, false>(CSTP_K2_ARGS);
  } else if (ia.ss) {
    if (v4) launch_k2<false, true, CSTP_K2_BKN, true>(CSTP_K2_ARGS);
    else launch_k2<false, false, CSTP_K2_BKN, true>(CSTP_K2_ARGS);
  } else {
    if (v4) launch_k2<false, true, CSTP_K2_BKN, false>(CSTP_K2_ARGS);
    else launch_k2<false, false, CSTP_K2_BKN, false>(CSTP_K2_ARGS);
  }
#undef CSTP_K2_ARGS
  CSTP_LAUNCH_CHECK();
  const size_t tot = (size_t)d.k * d.c * p.ntaps;
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(pack_grid(tot)), dim3(256), 0, s, dwp, dw, d.k, d.c, p.ntaps, p.w_Cp, p.w_Jp,
                     w_f16 ? xcell : nullptr, w_f16 ? dycell : nullptr, det ? splits : 1, det_stride, accumulate ? 1 : 0);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_conv3d_query_tile(const cstp_conv_desc* desc, int32_t mode, int32_t* out4) {
  CSTP_REQUIRE(desc && out4, "null argument");
  CSTP_REQUIRE(mode >= 0 && mode <= 3, "mode must be 0 (forward), 1 (backward_data), 2 (backward_weight) or 3 (forward with an in_affine)");
  ConvPlan p;
  CSTP_REQUIRE(make_plan(*desc, p, mode == 3), "invalid conv descriptor");
  if (mode == 3) mode = 0;
  if (mode == 2 && p.w_patch) {       // igemm_k2p: 144 rows x (32 channels x 9 taps), f16 pair
    out4[0] = WP_BM; out4[1] = 288; out4[2] = 2; out4[3] = 0;
    return 0;
  }
  if (mode == 2 && p.w_tpatch) {      // igemm_k2t: 144 x-channels x (64 dY channels x 3 taps), f16 pair
    out4[0] = WP_BM; out4[1] = 192; out4[2] = 2; out4[3] = 0;
    return 0;
  }
  if (mode == 2) {
    out4[0] = p.w_split ? 16 * p.w_mt : (p.w_mt == 9 ? 144 : 32 * p.w_mt);
    out4[1] = 128;
    out4[2] = p.w_split ? split_planes() : 0;
    out4[3] = p.w_blocks / 256;
    return 0;
  }
  const Tile& t = mode == 0 ? p.f_t : p.d_t;
  out4[0] = tile_bm(t);
  out4[1] = tile_bn(t);
  out4[2] = t.sp ? split_planes() : 0;       // (the patch kernel and the stems' straddle mode: f16 pair only -- make_plan)
  out4[3] = t.tpb == 2 ? 2 : 1;
  return 0;
}

extern "C" int cstp_conv3d_get_tile(const cstp_conv_desc* desc, int32_t mode, int32_t* tile4) {
  CSTP_REQUIRE(desc && tile4, "null argument");
  CSTP_REQUIRE(mode == 0 || mode == 1 || mode == 2, "mode must be 0 (forward), 1 (backward_data) or 2 (backward_weight)");
  Tile t;
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    auto it = g_tuned.find(tune_key(*desc, mode));
    if (it == g_tuned.end()) { tile4[0] = tile4[1] = tile4[2] = tile4[3] = -1; return 0; }
    t = it->second;
  }
  tile4[0] = t.sp;
  tile4[1] = t.m16 ? 9 : t.mt;
  tile4[2] = t.wm;                                   // native: waves along rows; split: 128-column halves; mode 2: blocks / 256
  tile4[3] = mode == 2 ? 0 : (t.tpb == 2 ? 2 : 1);
  return 0;
}

#if KP_DIAG & 16
extern "C" int cstp_debug_stamps(unsigned long long* out8) {      // diagnostic builds only: read and reset the k1p stamps
  unsigned long long zero[8] = {};
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(cstp::kp_stamp), sizeof(zero)) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(cstp::kp_stamp), zero, sizeof(zero)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int cstp_gemm_set_split_terms(int32_t terms) {
  CSTP_REQUIRE(terms >= 0 && terms <= 3, "split terms: 2 (f16 pair), 3 (bf16 triple), 1 (native f32 MFMA only) or 0 (environment default)");
  g_split_terms.store(terms, std::memory_order_relaxed);
  return 0;
}

extern "C" int32_t cstp_gemm_get_split_terms(void) { return native_only() ? 1 : split_planes(); }

extern "C" int cstp_conv3d_set_tile(const cstp_conv_desc* desc, int32_t mode, const int32_t* tile4) {
  CSTP_REQUIRE(desc && tile4, "null argument");
  CSTP_REQUIRE(mode == 0 || mode == 1 || mode == 2, "mode must be 0 (forward), 1 (backward_data) or 2 (backward_weight)");
  Tile t{};
  const int split = tile4[0], mt = tile4[1];
  if (mode == 2) {
    const int blocks = tile4[2];
    CSTP_REQUIRE(blocks >= 1 && blocks <= 64, "split-K block target / 256 out of range");
    if (split == 2) {
      CSTP_REQUIRE(mt == 9, "patch weight-gradient tile: 9 row tiles of 16");
      t = Tile{9, 1, 0, 0, 2};
    } else if (split) {
      CSTP_REQUIRE(mt == 4 || mt == 8 || mt == 9, "split weight-gradient tiles: 4, 8 or 9 row tiles of 16");
      t = Tile{mt, blocks, 0, 0, 1};
    } else {
      CSTP_REQUIRE((mt >= 1 && mt <= 5) || mt == 9, "native weight-gradient tiles: 1..5 row tiles of 32, or 9 (144 rows)");
      t = Tile{mt, blocks, mt == 9 ? 1 : 0, 0, 0};
    }
  } else if (split == 2) {
    CSTP_REQUIRE(patch_mt_ok(mt), "patch tiles: 4, 8 or 9 row tiles of 16");
    // tile4[2] == 2: the weight-resident temporal forward kernel igemm_k1w (64 rows; run_k1t falls back to igemm_k1t where it does not apply)
    CSTP_REQUIRE(tile4[2] == 0 || tile4[2] == 1 || (tile4[2] == 2 && mt == 4 && mode == 0), "patch tiles: variant 2 = weight-resident, 64 rows, forward");
    t = Tile{mt, tile4[2] == 2 ? 2 : 1, 0, 1, 2};
  } else if (split) {
    CSTP_REQUIRE(split_mt_ok(mt), "split tiles: 2, 3, 4, 5, 6, 8 or 9 row tiles of 16");
    CSTP_REQUIRE(tile4[2] == 0 || tile4[2] == 1 || (tile4[2] == 2 && mt >= 8), "split tiles: 128 columns, or 256 with 8 / 9 row tiles");
    t = Tile{mt, tile4[2] == 2 ? 2 : 1, 0, 1, 1};
  } else {
    const int wm = tile4[2], tpb = tile4[3];
    CSTP_REQUIRE((wm == 1 && ((mt >= 1 && mt <= 5) || mt == 9)) || (wm == 2 && mt >= 1 && mt <= 2) || (wm == 4 && mt == 1),
                 "native tile shape");
    CSTP_REQUIRE(tpb == 1 || tpb == 2, "K-tiles per barrier: 1 or 2");
    t = Tile{mt, wm, mt == 9 ? 1 : 0, tpb, 0};       // mt == 9: the 144-row tile on the 16x16x4 MFMA
  }
  std::lock_guard<std::mutex> lk(g_tune_mu);
  g_tuned[tune_key(*desc, mode)] = t;
  return 0;
}

extern "C" int cstp_conv3d_autotune(void* stream, const cstp_conv_desc* desc, int32_t mode, const float* src,
                                    const float* w, float* out, void* ws, size_t ws_bytes, int32_t iters) {
  CSTP_REQUIRE(desc && src && w && out && ws, "null argument");
  CSTP_REQUIRE(mode == 0 || mode == 1 || mode == 2, "mode must be 0 (forward), 1 (backward_data) or 2 (backward_weight)");
  CSTP_REQUIRE(iters >= 1 && iters <= 100, "bad iteration count");
  const cstp_conv_desc& d = *desc;
  // The activation operands' largest magnitudes are measured ONCE here and handed to every candidate: in a training step the
  // cells come with the tensors (BatchNorm by-product), so the candidates are timed the way they will run.
  static unsigned* tune_cells = nullptr;
  if (tune_cells == nullptr && hipMalloc(&tune_cells, 256) != hipSuccess) return fail("hipMalloc failed%s", "");
  {
    ConvPlan tp;
    CSTP_REQUIRE(make_plan(d, tp), "invalid conv descriptor");
    hipStream_t s0 = as_stream(stream);
    if (hipMemsetAsync(tune_cells, 0, 8, s0) != hipSuccess) return fail("hipMemsetAsync failed%s", "");
    const size_t nx = (size_t)d.n * d.c * d.d * d.h * d.w, ny = (size_t)d.n * d.k * tp.Do * tp.Ho * tp.Wo;
    const size_t n0 = mode == 1 ? ny : nx;            // src = dy for the data gradient, x otherwise
    hipLaunchKernelGGL(absmax_kernel, dim3(absmax_grid(n0)), dim3(256), 0, s0, src, n0, tune_cells);
    if (mode == 2) hipLaunchKernelGGL(absmax_kernel, dim3(absmax_grid(ny)), dim3(256), 0, s0, w, ny, tune_cells + 1);
    CSTP_LAUNCH_CHECK();
  }
  if (mode == 2) {
    // weight gradient (src = x, w = dy, out = dw): row-tile height x split-K block target
    const bool stem = d.c < 8;
    Tile wc[24];
    int nw = 0;
    const int base = pick_mt(d.k);
    const bool allow_split2 = !native_only();
    for (int blocks = 4; blocks <= 16; blocks *= 2) {
      wc[nw++] = Tile{base, blocks, 0};
      if (!stem && d.k > 128 && d.k <= 144) wc[nw++] = Tile{9, blocks, 1};
      for (int mt = 2; mt <= 5; ++mt)
        if (mt != base && cdiv(d.k, 32 * mt) * 32 * mt <= cdiv(d.k, 32 * base) * 32 * base + 16 && nw < 15) wc[nw++] = Tile{mt, blocks, 0};
    }
    if (allow_split2 && stem && split_planes() == 2 && d.k >= 48) {      // the stems on igemm_k2s<.., STR> (f16 pair)
      int smt = 4, pad = cdiv(d.k, 64) * 64 - d.k;
      if (cdiv(d.k, 128) * 128 - d.k <= pad) { smt = 8; pad = cdiv(d.k, 128) * 128 - d.k; }
      if (cdiv(d.k, 144) * 144 - d.k < pad) smt = 9;
      for (int blocks = 4; blocks <= 16; blocks *= 2) wc[nw++] = Tile{smt, blocks, 0, 0, 1};
    }
    if (allow_split2 && !stem && d.k >= 48) {      // igemm_k2s: 64- / 128- / 144-row tiles, whichever pads the rows least
      int smt = 4, pad = cdiv(d.k, 64) * 64 - d.k;
      if (cdiv(d.k, 128) * 128 - d.k <= pad) { smt = 8; pad = cdiv(d.k, 128) * 128 - d.k; }
      if (cdiv(d.k, 144) * 144 - d.k < pad) smt = 9;
      for (int blocks = 4; blocks <= 16; blocks *= 2) wc[nw++] = Tile{smt, blocks, 0, 0, 1};
    }
    if (allow_split2 && !stem && (wpatch_geom_ok(d) || twpatch_geom_ok(d))) wc[nw++] = Tile{9, 1, 0, 0, 2};      // igemm_k2p / igemm_k2t
    hipStream_t s2 = as_stream(stream);
    hipEvent_t a0, a1;
    if (hipEventCreate(&a0) != hipSuccess || hipEventCreate(&a1) != hipSuccess) return fail("hipEventCreate failed%s", "");
    float bms = 1e30f;
    int bi = -1, rc2 = 0;
    for (int i = 0; i < nw && rc2 == 0; ++i) {
      g_force_tile = &wc[i];
      g_force_mode = 2;
      for (int it = -1; it < iters && rc2 == 0; ++it) {
        if (it == 0) (void)hipEventRecord(a0, s2);
        rc2 = cstp_conv3d_backward_weight_am(stream, desc, src, nullptr, w, out, ws, ws_bytes, tune_cells, tune_cells + 1);
      }
      g_force_tile = nullptr;
      if (rc2 != 0) break;
      (void)hipEventRecord(a1, s2);
      if (hipEventSynchronize(a1) != hipSuccess) { rc2 = fail("hipEventSynchronize failed%s", ""); break; }
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, a0, a1);
      if (ms < bms) { bms = ms; bi = i; }
    }
    g_force_tile = nullptr;
    (void)hipEventDestroy(a0);
    (void)hipEventDestroy(a1);
    if (rc2 != 0) return rc2;
    CSTP_REQUIRE(bi >= 0, "no weight-gradient candidate ran");
    std::lock_guard<std::mutex> lk(g_tune_mu);
    g_tuned[tune_key(d, 2)] = wc[bi];
    return 0;
  }
  const int M = mode == 0 ? d.k : d.c;
  const bool straddle = (mode == 0 && d.c < 8);
  Tile cand[44] = {{1, 1, 0, 1}, {2, 1, 0, 1}, {3, 1, 0, 1}, {4, 1, 0, 1}, {5, 1, 0, 1}, {1, 2, 0, 1}, {2, 2, 0, 1}, {1, 4, 0, 1},
                   {9, 1, 1, 1}};
  int ncand = (CSTP_M16 && !straddle && M > 128 && M <= 144) ? 9 : 8;
  if (!straddle) {       // the same tiles with two K-tiles per barrier
    const int n1 = ncand;
    for (int i = 0; i < n1; ++i) { cand[ncand] = cand[i]; cand[ncand].tpb = 2; ++ncand; }
  }
  // the 3xbf16-split kernels (fp32-equivalent products on the bf16 matrix cores), unless CSTP_GEMM=f32
  const bool allow_split = !native_only();
  if (allow_split && !straddle && d.kt * d.kh * d.kw <= 27) {
    // row-tile heights 32..144; keep those that pad M by at most ~1/8 (and the two smallest paddings regardless)
    static const int smt[] = {2, 3, 4, 5, 6, 8, 9};
    int best_pad = 1 << 30;
    for (int mt : smt) { const int pad = cdiv(M, 16 * mt) * 16 * mt - M; if (pad < best_pad) best_pad = pad; }
    for (int mt : smt) {
      const int pad = cdiv(M, 16 * mt) * 16 * mt - M;
      if ((pad <= best_pad + M / 8) && ncand < 38) {
        cand[ncand++] = Tile{mt, 1, 0, 1, 1};
        if (mt >= 8) cand[ncand++] = Tile{mt, 2, 0, 1, 1};      // 256-column tile
      }
    }
  }
  if (allow_split && straddle && split_planes() == 2) {      // the stems: split tiles in straddle mode, 64 / 80 / 96 rows
    for (int mt = 4; mt <= 6; ++mt)
      if (cdiv(M, 16 * mt) * 16 * mt - M <= 16 + M / 8 && ncand < 40) cand[ncand++] = Tile{mt, 1, 0, 1, 1};
  }
  if (allow_split && (patch_geom_ok(d) || tpatch_geom_ok(d))) {       // the LDS-resident-patch kernels: row blocks of 64 / 128 / 144
    static const int pmt[] = {4, 8, 9};
    int best_pad = 1 << 30;
    for (int mt : pmt) { const int pad = cdiv(M, 16 * mt) * 16 * mt - M; if (pad < best_pad) best_pad = pad; }
    for (int mt : pmt)
      if (cdiv(M, 16 * mt) * 16 * mt - M <= best_pad + M / 8 && ncand < 40) cand[ncand++] = Tile{mt, 1, 0, 1, 2};
    // the weight-resident temporal forward kernel igemm_k1w (64 rows, <= 15 K-tiles)
    if (mode == 0 && tpatch_geom_ok(d) && k1w_fits(d.k, d.c) && ncand < 41) cand[ncand++] = Tile{4, 2, 0, 1, 2};
  }
  hipStream_t s = as_stream(stream);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail("hipEventCreate failed%s", "");
  float best_ms = 1e30f;
  int best = -1, rc = 0;
  for (int i = 0; i < ncand && rc == 0; ++i) {
    if (cdiv(M, tile_bm(cand[i])) * tile_bm(cand[i]) > M + 160) continue;
    g_force_tile = &cand[i];
    g_force_mode = mode;
    for (int it = -1; it < iters && rc == 0; ++it) {      // it == -1: untimed warm-up launch
      if (it == 0) (void)hipEventRecord(e0, s);
      rc = mode == 0 ? cstp_conv3d_forward_am(stream, desc, src, w, nullptr, nullptr, out, ws, ws_bytes, tune_cells)
                     : cstp_conv3d_backward_data_am(stream, desc, src, w, out, ws, ws_bytes, tune_cells);
    }
    g_force_tile = nullptr;
    if (rc != 0) break;
    (void)hipEventRecord(e1, s);
    if (hipEventSynchronize(e1) != hipSuccess) { rc = fail("hipEventSynchronize failed%s", ""); break; }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // Candidates are ranked the way the training step runs them: every convolution of these networks feeds a train-mode
    // BatchNorm, and a forward candidate that cannot leave that BatchNorm's sums in its epilogue (anything but the patch
    // kernels, on a layer where those can) is followed by a statistics pass over y -- one read of the output at the ~5.5 TB/s
    // the BatchNorm passes reach.  (Measured on T1, 144 -> 64 at 16 x 56 x 56: gather 0.524 ms + 0.075 ms of bn_reduce against
    // 0.523 ms for igemm_k1w with the sums inside.)
    if (mode == 0 && cand[i].sp != 2 && k1p_stats_nsplit(Tile{4, 1, 0, 1, 2}, d, 2) > 0) {
      const double ybytes = 4.0 * d.n * d.k * (double)d.d * d.h * d.w;
      ms += (float)(iters * ybytes / 5.5e9);
    }
    {
      static const bool verbose = getenv("CSTP_TUNE_VERBOSE") != nullptr;     // developer knob: the ranking as the tuner saw it
      if (verbose)
        fprintf(stderr, "cstp tune mode %d [%d %d %d %d %d -> %d, %dx%dx%d]: tile {sp %d, mt %d, wm %d, tpb %d}  %.4f ms per launch (as ranked)\n",
                mode, d.n, d.c, d.d, d.h, d.w, d.k, d.kt, d.kh, d.kw, cand[i].sp, cand[i].m16 ? 9 : cand[i].mt, cand[i].wm,
                cand[i].tpb, ms / iters);
    }
    if (ms < best_ms) { best_ms = ms; best = i; }
  }
  g_force_tile = nullptr;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc != 0) return rc;
  CSTP_REQUIRE(best >= 0, "no tile candidate ran");
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    g_tuned[tune_key(d, mode)] = cand[best];
  }
  return 0;
}
