// K2T  igemm_k2t<AFF>: weight gradient of the stride-1 3x1x1 pad-(1,0,0) TEMPORAL convolutions with both operands staged ONCE --
// igemm_k2p's stream scheme (igemm_wpatch.h) turned to the frame axis, f16-pair arithmetic (igemm_split.h).
//
//   dW[m][c][dt] = sum over (clip, d, hw) of  dY[m][d][hw] * z[c][d + dt - 1][hw],      z = x  or  act(x * scale + shift)  (AFF)
//
// The gather kernel igemm_k2s tiles the OUTPUT (64 rows x 128 (tap, channel) columns): on the first stage's layer (144 -> 64
// channels at 16 x 56 x 56) it fetches z once per tap and dY once per column tile -- 3.09 GB for 1.34 GB of operands (PMC,
// profiles/r03/pmc_T1_wgrad.txt: 2.3 x) at 0.70 ms, 0.24 of the HBM roofline, on a layer with 66 FLOP per byte.  Here a work
// item is (clip, chunk of 32 consecutive frame positions): its 16 frames at that chunk form a STREAM of 32-row pieces, one
// 128-byte line per channel and frame, preceded by one all-zero halo frame (shared with the item before), so that a temporal
// tap is a CONSTANT shift of 32 stream rows:   dW[dt] += z_stream[q] (x) dY_stream[q + (1 - dt) * 32].
//   * z  (144 channels = nine MFMA row tiles): a [32 rows][144 channels] image per K-step, igemm_k2p's dY image layout and
//     fragment reads; AFF: the BatchNorm + ReLU in front applied between load and split, once per element (the gather kernel
//     pays it once per tap), zero on the halo frames;
//   * dY (64 channels = four column tiles, one per consumer wave): a RING of 256 stream rows per 32-channel block and plane,
//     igemm_k2p's x ring layout; a row enters two intervals ahead and serves the three taps from LDS.
// A block owns one (144 z channels, 64 dY channels) pair -- the first stage's layer has exactly one -- and a contiguous range of
// items, keeps the 144 x 64 x 3 accumulator tile in registers (27 tiles per consumer wave) and adds it into the packed slab once.
// Both operands leave HBM once, in whole aligned lines.
#pragma once

namespace cstp {

constexpr int WT_RING = 256;          // dY ring rows (4 segments of 64)
constexpr int WT_AFFC = 160;          // AFF: channels per group the LDS tables hold (one 144-row block)

struct WTGeom {
  int C, M;          // channels of x (z) / of dY
  int nrb, ncb;      // 144-channel blocks of x, 64-channel blocks of dY
  int D, HW, Nb;     // frames per clip, frame size (a multiple of 32), clips
  int nchunk;        // HW / chunk
  int nitems;        // Nb * nchunk work items
  int nsplit, iper;  // blocks per (row block, column block) pair, items per block
  int Jp, Cp;        // slab row pitch; channels per tap in the slab's column order j = tap * Cp + c
  unsigned mg_fp1, mg_nchunk;           // 2^32 / (D + 1), / nchunk, rounded up: division by multiplication
  int aff_npg, aff_groups, aff_relu;    // AFF: clips per BatchNorm group of x, groups (<= 2), ReLU
};

// CH = positions per chunk: 32 (one aligned 128-byte line per channel and frame; a K-step = one frame) or 16 (frame sizes that are
// multiples of 16 only -- 28 x 28: a K-step = two frames of half lines, a tap = a shift of 16 stream rows)
template <bool AFF, int CH>
__global__ void __launch_bounds__(512, 2)
igemm_k2t(const WTGeom g, const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dwp,
          const unsigned* __restrict__ xcell, const unsigned* __restrict__ dycell, size_t det_stride,
          const float2* __restrict__ aff_ss) {
  __shared__ uint2 Ad[4 * WP_AIMG];            // z images of four K-steps = two intervals (double buffer)
  __shared__ uint2 Yr[2][2][WT_RING * 8];      // dY ring: [32-channel block][plane][row][piece]
  __shared__ float2 aff_t[AFF ? 2 * WT_AFFC : 1];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  // XCD-major enumeration: the pairs of one item range sit on one XCD
  const int u = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int ncombo = g.nrb * g.ncb;
  const int split = u / ncombo, combo = u - split * ncombo;
  if (split >= g.nsplit) return;
  const int rb = combo / g.ncb, cbk = combo - rb * g.ncb;
  const int i_begin = split * g.iper;
  int I = g.nitems - i_begin;
  I = I < g.iper ? I : g.iper;
  if (I <= 0) return;
  const int D = g.D, HW = g.HW, FP1 = D + 1;
  static_assert(CH == 32 || CH == 16, "chunk");
  const int NI = (I * FP1 * CH + 63) >> 6;             // intervals (64 rows = two K-steps) of this block's stream
  const int c0 = rb * WP_BM, m0 = cbk * 64;

  // the ring starts all zero: the first K-steps read rows the stream has not written (against z rows that are zero -- the
  // products vanish, but only against FINITE ring contents)
  for (int e = t; e < 2 * 2 * WT_RING * 8; e += 512) (&Yr[0][0][0])[e] = make_uint2(0u, 0u);
  if constexpr (AFF) {
    for (int e = t; e < g.aff_groups * WP_BM; e += 512) {
      const int grp = e / WP_BM, c = e - grp * WP_BM;
      const int cc = c0 + c < g.C ? c0 + c : g.C - 1;
      aff_t[grp * WT_AFFC + c] = aff_ss[grp * g.C + cc];
    }
  }
  __syncthreads();

  if (wave >= 4) {
    // ================================================= producers =================================================
    const int pw = wave - 4;                           // my channel set: z channels 36 pw .. 36 pw + 35, dY channels 16 pw .. 16 pw + 15
    const int r = lane & 31, half = lane >> 5;         // my stream row inside the interval: K-step `half`, k-row r
    constexpr unsigned OOB = 0x80000000u;
    const unsigned S4 = (unsigned)D * (unsigned)HW * 4u;      // channel stride in bytes
    const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(dy, (unsigned)((size_t)g.Nb * g.M * D * HW * 4));
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x, (unsigned)((size_t)g.Nb * g.C * D * HW * 4));
    float sc_x, sc_dy, inv_unused;
    f16_scale(__builtin_amdgcn_readfirstlane(*xcell), sc_x, inv_unused);
    f16_scale(__builtin_amdgcn_readfirstlane(*dycell), sc_dy, inv_unused);

    // my row of interval k -> (item, frame) by multiplication with 2^32 / divisor -> byte offset of channel 0 there (z: nch = C,
    // dY: nch = M), or OOB (halo frame, past the block's last item); *grp: the BatchNorm group of the row's clip
    auto row_base = [&](int k, int nch, int* grp) __attribute__((always_inline)) -> unsigned {
      const unsigned sr = (unsigned)(k * 64 + lane);
      const unsigned L = sr / (unsigned)CH, p = sr & (unsigned)(CH - 1);
      const unsigned it = __umulhi(L, g.mg_fp1), ll = L - it * (unsigned)FP1;
      const unsigned ig = (unsigned)i_begin + it;
      const unsigned nb = __umulhi(ig, g.mg_nchunk), ch = ig - nb * (unsigned)g.nchunk;
      const bool ok = it < (unsigned)I && ll >= 1;
      if (grp != nullptr) *grp = ok ? (int)(nb / (unsigned)g.aff_npg) : 0;
      return ok ? ((nb * (unsigned)nch * D + (ll - 1)) * HW + ch * (unsigned)CH + p) * 4u : OOB;
    };
    // channels past the tensors' last ones are CLAMPED (finite values whose products land in slab cells nobody reads);
    // the channel rides in the load's scalar offset (wave-uniform)
    auto x_soff = [&](int j) __attribute__((always_inline)) -> unsigned {
      int c = c0 + 36 * pw + j;
      c = c < g.C ? c : g.C - 1;
      return (unsigned)c * S4;
    };
    auto dy_soff = [&](int j) __attribute__((always_inline)) -> unsigned {
      int m = m0 + 16 * pw + j;
      m = m < g.M ? m : g.M - 1;
      return (unsigned)m * S4;
    };
    int kx = 0, ky = 0;                                 // next z interval / dY segment to request
    struct Regs { float a[36], b[16]; int grp; float cap; };
    auto load_y = [&](float (&v)[16]) __attribute__((always_inline)) {
      const unsigned b = row_base(ky++, g.M, nullptr);
#pragma unroll
      for (int j = 0; j < 16; ++j) buf_load_x1(v[j], b, rs_dy, dy_soff(j));
    };
    // dY ring: my 16 channels (pieces 4 (pw & 1) .. + 3 of 32-channel block pw >> 1) of stream row 64 * seg + lane
    auto store_y = [&](int seg, const float (&v)[16]) __attribute__((always_inline)) {
      const int row = ((seg << 6) + lane) & (WT_RING - 1);
      uint2* P0 = Yr[pw >> 1][0];
      uint2* P1 = Yr[pw >> 1][1];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned h0, l0, h1, l1;
        split2h(v[4 * q], v[4 * q + 1], sc_dy, h0, l0);
        split2h(v[4 * q + 2], v[4 * q + 3], sc_dy, h1, l1);
        const int s = row * 8 + wp_slot(row, 4 * (pw & 1) + q);
        P0[s] = make_uint2(h0, h1);
        P1[s] = make_uint2(l0, l1);
      }
    };
    auto issue = [&](Regs& R) __attribute__((always_inline)) {       // the next item: z of an interval + one dY segment
      int grp = 0;
      const unsigned bx = row_base(kx++, g.C, &grp);
#pragma unroll
      for (int j = 0; j < 36; ++j) buf_load_x1(R.a[j], bx, rs_x, x_soff(j));
      R.grp = grp;
      R.cap = bx != OOB ? __builtin_inff() : 0.f;      // AFF: halo / padding rows stay zero through the clamp
      load_y(R.b);
    };
    // z image: my row's nine pieces (channels 36 pw + 4 i ..) -> sub-image (row tile, plane), slot of (k-row r, piece)
    const int a_rb = wp_aslot(r, 0) & ~12, a_xr = ((r >> 2) & 1) | (((r >> 4) & 1) << 1);
    auto store = [&](int buf, int item, const Regs& R) __attribute__((always_inline)) {
      uint2* img = Ad + (2 * buf + half) * WP_AIMG;
      const float lo = g.aff_relu ? 0.f : -R.cap;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        float z[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          z[e] = R.a[4 * i + e];
          if constexpr (AFF) {
            const float2 ss = aff_t[R.grp * WT_AFFC + 36 * pw + 4 * i + e];
            z[e] = __builtin_amdgcn_fmed3f(__builtin_fmaf(z[e], ss.x, ss.y), lo, R.cap);
          }
        }
        unsigned h0, l0, h1, l1;
        split2h(z[0], z[1], sc_x, h0, l0);
        split2h(z[2], z[3], sc_x, h1, l1);
        const int pj = 9 * pw + i;                    // piece of the 144-channel row: row tile pj / 4, piece pj % 4
        uint2* sub = img + (pj >> 2) * 256 + (a_rb | (((pj & 3) ^ a_xr) << 2));
        sub[0] = make_uint2(h0, h1);
        sub[128] = make_uint2(l0, l1);
      }
      store_y(item + 1, R.b);                          // lead = 2 intervals: segment item + 1 goes in with interval `item`
    };

    // ---- prologue: dY segment 0, then item 0 (z of interval 0 + dY segment 1)
    {
      float py[16];
      load_y(py);
      store_y(0, py);
    }
    Regs R0, R1;
    issue(R0);
    issue(R1);
    store(0, 0, R0);
    issue(R0);
    __syncthreads();
    for (int i = 0; i < NI; i += 2) {
      store(1, i + 1, R1);
      issue(R1);
      __syncthreads();
      if (i + 1 >= NI) break;
      store(0, i + 2, R0);
      issue(R0);
      __syncthreads();
    }
    return;
  }

  // ================================================== consumers ==================================================
  const int ct = wave;                                // my 16-channel column tile of the 64-channel dY block
  const int grp = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
  const int r_lo = 8 * grp + lq, r_hi = r_lo + 4;
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  auto tr2 = [&](const uint2* lo_p, const uint2* hi_p) __attribute__((always_inline)) -> s16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)lo_p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)hi_p);
    return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
  const int a_lo = wp_aslot(r_lo, lp), a_hi = wp_aslot(r_hi, lp);
  const uint2* Y0 = Yr[ct >> 1][0];
  const uint2* Y1 = Yr[ct >> 1][1];
  // my three taps: ring slot of my k-rows r_lo / r_hi shifted by (1 - dt) * CH rows, swizzle included (it reads row bits 2..4,
  // which a K-step of 32 rows does not change); a K-step moves the slots on by 32 rows = 256 slots (mod the ring)
  int I_lo[3], I_hi[3];
#pragma unroll
  for (int tp = 0; tp < 3; ++tp) {
    const int sh = (1 - tp) * CH;
    I_lo[tp] = ((sh + r_lo) & (WT_RING - 1)) * 8 + wp_slot(sh + r_lo, (ct & 1) * 4 + lp);
    I_hi[tp] = ((sh + r_hi) & (WT_RING - 1)) * 8 + wp_slot(sh + r_hi, (ct & 1) * 4 + lp);
  }
  f32x4 acc[3][9];
#pragma unroll
  for (int tp = 0; tp < 3; ++tp)
#pragma unroll
    for (int m = 0; m < 9; ++m)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) acc[tp][m][rr] = 0.f;

  __builtin_amdgcn_s_setprio(2);
  __syncthreads();                                    // the prologue's data is staged
  for (int it = 0; it < NI; ++it) {
#pragma unroll
    for (int hk = 0; hk < 2; ++hk) {                 // the interval's two K-steps
      const uint2* A = Ad + (2 * (it & 1) + hk) * WP_AIMG;
      auto a_frag = [&](int pl, int m) __attribute__((always_inline)) -> s16x8 {
        return tr2(A + (2 * m + pl) * 128 + a_lo, A + (2 * m + pl) * 128 + a_hi);
      };
      s16x8 b[3][2];
#pragma unroll
      for (int tp = 0; tp < 3; ++tp) {
        b[tp][0] = tr2(Y0 + I_lo[tp], Y0 + I_hi[tp]);
        b[tp][1] = tr2(Y1 + I_lo[tp], Y1 + I_hi[tp]);
        I_lo[tp] = (I_lo[tp] + 256) & (WT_RING * 8 - 1);
        I_hi[tp] = (I_hi[tp] + 256) & (WT_RING * 8 - 1);
      }
      s16x8 f_lo = a_frag(1, 0), f_hi = a_frag(0, 0);
#pragma unroll
      for (int m = 0; m < 9; ++m) {
        // lo * hi, hi * lo, hi * hi (smallest first); the taps inner, so that dependent products are three apart
#define CSTP_MM(A_, Q)                                                                                              \
  _Pragma("unroll") for (int tp = 0; tp < 3; ++tp)                                                                  \
    acc[tp][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A_), __builtin_bit_cast(f16x8, b[tp][Q]), acc[tp][m], 0, 0, 0);
        CSTP_MM(f_lo, 0)
        if (m + 1 < 9) f_lo = a_frag(1, m + 1);
        CSTP_MM(f_hi, 1)
        CSTP_MM(f_hi, 0)
        if (m + 1 < 9) f_hi = a_frag(0, m + 1);
#undef CSTP_MM
      }
    }
    __syncthreads();
  }

  // ---- epilogue: C layout col (dY channel) = lane & 15, row (z channel) = (lane >> 4) * 4 + reg; one add per value, once per block
  const bool det = det_stride != 0;
  float* slab = dwp + (size_t)split * det_stride;
  const int mch = m0 + ct * 16 + li;
#pragma unroll
  for (int tp = 0; tp < 3; ++tp) {
#pragma unroll
    for (int m = 0; m < 9; ++m) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int c = c0 + m * 16 + grp * 4 + rr;
        if (mch < g.M && c < g.C) wgrad_out(&slab[(size_t)mch * g.Jp + tp * g.Cp + c], acc[tp][m][rr], det);
      }
    }
  }
}

}  // namespace cstp
