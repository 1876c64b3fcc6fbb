// K2P  igemm_k2p: weight gradient of the stride-1 1x3x3 pad-(0,1,1) spatial convolutions with the x operand RESIDENT IN LDS
// across the nine filter taps (north_star: "LDS-staged input tiles"), f16-pair arithmetic (igemm_split.h).
//
//   dW[m][c][dh][dw] = sum over (frame, h, w) of  dY[m][h][w] * x[c][h + dh - 1][w + dw - 1]
//
// The gather kernel igemm_k2s stages, per 128-column tile, every x element once per tap it meets and every dY element once
// per column tile -- nine and 4.5 times on the S1 layer -- and its producers' split work is as long as the consumers' MFMA
// work.  Here both operands are STREAMS over padded positions: a block walks its frames line by line with a pitch of W + 2
// (two halo columns per line, one halo line between frames), so that a filter tap is a CONSTANT shift of the x stream
// against the dY stream,  x_stream[q + dh * PITCH + dw]  <->  dY_stream[q]  (dY is zero on its halo positions, x on its
// own, which makes the products at the image borders vanish without any mask).  A K-step is 32 stream positions:
//   * dY: a [32 positions][144 channels] image per K-step, one 1 KiB sub-image per (16-channel row tile, plane), whose slot
//     order makes the transposing fragment reads (ds_read_b64_tr_b16: the reduction axis is the memory-contiguous one) and the
//     staging stores bank-conflict free and puts every fragment at lane base + compile-time offset;
//   * x : a RING of 512 stream rows x 32 channels, one array per plane, 64-byte rows of 8-byte pieces (4 channels), the
//     piece slot XOR-swizzled by the row so that transposing reads of ANY row alignment and the staging stores are
//     bank-conflict free; rows enter `lead` K-steps ahead of their first use and each row serves all nine taps (and both
//     16-channel column tiles) from LDS: x and dY are each split ONCE per (row block, channel block).
// A block owns one (144-row block of dY channels, 32-channel block of x) pair and a contiguous range of frames, keeps the
// 144 x 32 x 9 accumulator tile in registers over the whole range (162 MFMA tiles over the four consumer waves: column tile x
// tap half, the middle tap shared 5 : 4 row tiles) and adds it into the packed slab once at the end -- 41 MB of atomics
// for the S1 layer where igemm_k2s wrote 132 MB.
// Waves 4..7 are producers, waves 0..3 consumers.  The producers work in INTERVALS of 64 stream rows (two K-steps, one
// barrier): lane = stream row, wave = a fixed set of channels (36 of dY, 8 of x), so a load instruction covers 64 consecutive
// positions of one channel with the channel in the SCALAR offset -- no per-load address arithmetic -- and the position decode
// (three multiplications by 2^32 / divisor) is paid once per 64 rows.  That matters because on this one-block-per-CU kernel the
// SIMDs execute vector-ALU and matrix instructions almost exclusively one AFTER the other (PMC: SQ_VALU_MFMA_COEXEC_CYCLES is 3 %
// of the MFMA-busy cycles, profiles/r03/): every producer instruction is time the matrix pipe idles.  History on the S1 layer:
// first version 1.24 ms (igemm_k2s: 1.10), without the readfirstlane loops the compiler wraps around loads whose scalar
// offset differs across a wave 0.83, this version see DESIGN.
#pragma once

#ifndef WP_DIAG
#define WP_DIAG 0     // timing-only diagnostic builds (wrong results): bit 0 = producers only synchronise after the prologue,
#endif                // bit 1 = consumers only synchronise, bit 2 = no memory traffic (all loads out of range), bit 3 = consumers
                      // read their fragments but issue no products, bit 4 = producers store the raw bits (no split work)

namespace cstp {

constexpr int WP_RING = 512;          // x ring rows (8 segments of 64)
constexpr int WP_BM = 144;            // dY channels per block (9 MFMA row tiles)
constexpr int WP_AIMG = 9 * 2 * 128;  // uint2 slots of one K-step's dY image: [row tile][plane][128 slots]

struct WPGeom {
  int C, M;          // channels of x / of dY
  int ncb, nmblk;    // 32-channel blocks of x, 144-row blocks of dY
  int H, W, D, NF;   // frame size, frames per clip, frames in total
  int lead;          // intervals (64 rows) the x staging runs ahead: 64 * lead >= 2 * (W + 2) + 66
  int nsplit, fper;  // blocks per (row block, channel block) pair, frames per block
  int Jp, Cp;        // slab row pitch; channels per tap in the slab's column order j = tap * Cp + c
  unsigned mg_pitch, mg_hp1, mg_d;      // 2^32 / (W + 2), / (H + 1), / D, rounded up: division by multiplication
};

// x ring: slot of piece `pi` (0..7) in ring row `row`: the 4-piece halves swap with bit 3 of the row (rows r and r + 8 of a
// transposing read share a 16-bank window) and the pieces inside a half permute with bits 2 and 4 (the 8 rows r + 4 k of a
// staging store share a window): checked exhaustively, the 32 pieces a half-wave's ds_read_b64_tr_b16 touches -- rows
// r..r+3 and r+8..r+11 of one half, ANY r -- cover all 64 banks once, and so do 32 consecutive rows x one piece of a staging
// store.  The swizzle reads bits 2..4 of the row only, which a K-step's 32 rows do not change.
__device__ __forceinline__ int wp_slot(int row, int pi) {
  return pi ^ (((row >> 3) & 1) << 2) ^ (((row >> 2) & 1) | (((row >> 4) & 1) << 1));
}
// dY sub-image (one row tile, one plane: 32 k-rows x 4 pieces of 4 channels): slot of (k-row, piece).  Bank-pair bits =
// {row bits 0-1, piece ^ (row bits 2, 4), row bit 3}: a half-wave's transposing read (rows r..r+3, r+8..r+11 with r a
// multiple of 4, four pieces) and a 32-row store of one piece both touch 32 distinct bank pairs.
__device__ __forceinline__ int wp_aslot(int row, int piece) {
  return (row & 3) | ((piece ^ (((row >> 2) & 1) | (((row >> 4) & 1) << 1))) << 2) | (((row >> 3) & 1) << 4) |
         (((row >> 2) & 1) << 5) | (((row >> 4) & 1) << 6);
}

template <int TH>      // tap half of a consumer wave (compile-time so that every accumulator index is a constant)
__device__ __forceinline__ bool wp_mine(int tl, int m) { return TH == 0 ? (tl < 4 || m < 5) : (tl > 0 || m >= 5); }

__global__ void __launch_bounds__(512, 2)
igemm_k2p(const WPGeom g, const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dwp,
          const unsigned* __restrict__ xcell, const unsigned* __restrict__ dycell, size_t det_stride) {
  __shared__ uint2 Ad[4 * WP_AIMG];        // dY images of four K-steps = two intervals (double buffer)
  __shared__ uint2 Xr[2][WP_RING * 8];     // x ring: [plane][row][piece]

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  // XCD-major enumeration: the (row block, channel block) pairs of one frame range sit on one XCD and share its L2
  const int u = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int ncombo = g.ncb * g.nmblk;
  const int split = u / ncombo, combo = u - split * ncombo;
  if (split >= g.nsplit) return;
  const int mblk = combo / g.ncb, cb = combo - mblk * g.ncb;
  const int f_begin = split * g.fper;
  int F = g.NF - f_begin;
  F = F < g.fper ? F : g.fper;
  if (F <= 0) return;
  const int H = g.H, W = g.W, D = g.D, HW = H * W, PITCH = W + 2, HP1 = H + 1;
  const int NI = (F * HP1 * PITCH + 63) >> 6;            // intervals (64 rows = two K-steps) of this block's dY stream
  const int m0 = mblk * WP_BM;

  if (wave >= 4) {
    // ================================================= producers =================================================
    const int pw = wave - 4;                           // my channel set: dY rows 36 pw .. 36 pw + 35, x channels 8 pw .. 8 pw + 7
    const int r = lane & 31, half = lane >> 5;         // my stream row inside the interval: K-step `half`, k-row r
    constexpr unsigned OOB = 0x80000000u;
    const unsigned S4 = (unsigned)D * (unsigned)HW * 4u;      // channel stride in bytes
    const int Nb = g.NF / D;
    const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(dy, (unsigned)((size_t)Nb * g.M * D * HW * 4));
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x, (unsigned)((size_t)Nb * g.C * D * HW * 4));
    float sc_x, sc_dy, inv_unused;
    f16_scale(__builtin_amdgcn_readfirstlane(*xcell), sc_x, inv_unused);
    f16_scale(__builtin_amdgcn_readfirstlane(*dycell), sc_dy, inv_unused);

    // stream row -> (column, line of the frame's span, frame) by multiplication with 2^32 / divisor (exact below 2^25 rows
    // for divisors <= 128; the host checks): no cursor state, no branches
    // my row of interval k -> byte offset of channel 0 of dY there, or OOB (halo, past the end)
    auto dy_base = [&](int k) __attribute__((always_inline)) -> unsigned {
      const unsigned sr = (unsigned)(k * 64 + lane);
      const unsigned L = __umulhi(sr, g.mg_pitch), c = sr - L * PITCH;
      const unsigned fl = __umulhi(L, g.mg_hp1), ll = L - fl * HP1;
      const unsigned f = f_begin + fl, nb = D == 1 ? f : __umulhi(f, g.mg_d), d = f - nb * D;
      const bool ok = fl < (unsigned)F && ll >= 1 && c >= 1 && c <= (unsigned)W;
      if (WP_DIAG & 4) return OOB;       // (diagnostic: no memory traffic)
      return ok ? ((nb * g.M * D + d) * HW + (ll - 1) * W + (c - 1)) * 4u : OOB;
    };
    // x stream: columns 0, 1 are halo; line 0 of a frame's span is the LAST image line of the previous frame, line 1 the
    // halo line between the two, lines 2.. this frame's lines 0..
    auto x_base = [&](int k) __attribute__((always_inline)) -> unsigned {
      const unsigned sr = (unsigned)(k * 64 + lane);
      const unsigned L = __umulhi(sr, g.mg_pitch), c = sr - L * PITCH;
      const unsigned fl = __umulhi(L, g.mg_hp1), ll = L - fl * HP1;
      const bool prev = ll == 0;                        // (fl >= 1 is checked below: f_begin + fl - 1 does not wrap then)
      const unsigned f = f_begin + fl - (prev ? 1u : 0u), nb = D == 1 ? f : __umulhi(f, g.mg_d), d = f - nb * D;
      const unsigned h = prev ? (unsigned)(H - 1) : ll - 2;
      const bool ok = c >= 2 && (prev ? (fl >= 1 && fl <= (unsigned)F) : (ll >= 2 && fl < (unsigned)F));
      if (WP_DIAG & 4) return OOB;
      return ok ? ((nb * g.C * D + d) * HW + h * W + (c - 2)) * 4u : OOB;
    };
    // channels past the tensors' last ones are CLAMPED (finite values whose products land in slab cells nobody reads);
    // the channel rides in the load's scalar offset (wave-uniform)
    auto dy_soff = [&](int j) __attribute__((always_inline)) -> unsigned {
      int m = m0 + 36 * pw + j;
      m = m < g.M ? m : g.M - 1;
      return (unsigned)m * S4;
    };
    auto x_soff = [&](int j) __attribute__((always_inline)) -> unsigned {
      int c = cb * 32 + 8 * pw + j;
      c = c < g.C ? c : g.C - 1;
      return (unsigned)c * S4;
    };
    int kd = 0, kx = 0;                                 // next dY interval / x segment to request
    auto load_x = [&](float (&v)[8]) __attribute__((always_inline)) {
      const unsigned b = x_base(kx++);
#pragma unroll
      for (int j = 0; j < 8; ++j) buf_load_x1(v[j], b, rs_x, x_soff(j));
    };
    // x ring: my 8 channels (pieces 2 pw, 2 pw + 1) of stream row 64 * seg + lane
    auto store_x = [&](int seg, const float (&v)[8]) __attribute__((always_inline)) {
      const int row = ((seg << 6) + lane) & (WP_RING - 1);
      unsigned h[4], l[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split2h(v[2 * j], v[2 * j + 1], sc_x, h[j], l[j]);
      const int s0 = row * 8 + wp_slot(row, 2 * pw), s1 = row * 8 + wp_slot(row, 2 * pw + 1);
      Xr[0][s0] = make_uint2(h[0], h[1]); Xr[0][s1] = make_uint2(h[2], h[3]);
      Xr[1][s0] = make_uint2(l[0], l[1]); Xr[1][s1] = make_uint2(l[2], l[3]);
    };
    struct Regs { float a[36], b[8]; };
    auto issue = [&](Regs& R) __attribute__((always_inline)) {       // the next item: dY of an interval + one x segment
      const unsigned bd = dy_base(kd++);
#pragma unroll
      for (int j = 0; j < 36; ++j) buf_load_x1(R.a[j], bd, rs_dy, dy_soff(j));
      load_x(R.b);
    };
    // dY image: my row's nine pieces (channels 36 pw + 4 i ..) -> sub-image (row tile, plane), slot of (k-row r, piece);
    // the piece number is wave-uniform but not a constant, so the slot is put together arithmetically (wp_aslot)
    const int a_rb = wp_aslot(r, 0) & ~12, a_xr = ((r >> 2) & 1) | (((r >> 4) & 1) << 1);
    auto store = [&](int buf, int item, const Regs& R) __attribute__((always_inline)) {
      uint2* img = Ad + (2 * buf + half) * WP_AIMG;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        unsigned h0, l0, h1, l1;
        if (WP_DIAG & 16) {      // (diagnostic: no split work)
          h0 = __builtin_bit_cast(unsigned, R.a[4 * i]); l0 = __builtin_bit_cast(unsigned, R.a[4 * i + 1]);
          h1 = __builtin_bit_cast(unsigned, R.a[4 * i + 2]); l1 = __builtin_bit_cast(unsigned, R.a[4 * i + 3]);
        } else {
          split2h(R.a[4 * i], R.a[4 * i + 1], sc_dy, h0, l0);
          split2h(R.a[4 * i + 2], R.a[4 * i + 3], sc_dy, h1, l1);
        }
        const int pj = 9 * pw + i;                    // piece of the 144-channel row: row tile pj / 4, piece pj % 4
        uint2* sub = img + (pj >> 2) * 256 + (a_rb | (((pj & 3) ^ a_xr) << 2));
        sub[0] = make_uint2(h0, h1);
        sub[128] = make_uint2(l0, l1);
      }
      store_x(item + g.lead - 1, R.b);
    };

    // ---- prologue: x segments 0 .. lead - 2, then item 0 (dY of interval 0 + x segment lead - 1)
    {
      float px[4][8];
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (s < g.lead - 1) load_x(px[s]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (s < g.lead - 1) store_x(s, px[s]);
    }
    // Two register sets = two intervals (four K-steps, ~5 us) of global loads in flight: while the consumers multiply interval
    // i the producers store item i + 1 and request item i + 3.  (Depth beyond that changed nothing: 2 vs 4 sets of the
    // 32-row version.)
    Regs R0, R1;
    issue(R0);
    issue(R1);
    store(0, 0, R0);
    issue(R0);
    __syncthreads();
    for (int i = 0; i < NI; i += 2) {
#if !(WP_DIAG & 1)
      store(1, i + 1, R1);
      issue(R1);
#endif
      __syncthreads();
      if (i + 1 >= NI) break;
#if !(WP_DIAG & 1)
      store(0, i + 2, R0);
      issue(R0);
#endif
      __syncthreads();
    }
    return;
  }

  // ================================================== consumers ==================================================
  const int ct = wave & 1;                            // my 16-channel column tile of the 32-channel block
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
  // dY fragments: my two slots inside a sub-image; row tile m / plane pl add the compile-time offset (2 m + pl) * 128
  const int a_lo = wp_aslot(r_lo, lp), a_hi = wp_aslot(r_hi, lp);

#ifndef WP_NOPRIO
  __builtin_amdgcn_s_setprio(2);
#endif
  auto body = [&](auto th_tag) __attribute__((always_inline)) {
    constexpr int TH = decltype(th_tag)::value;
    // my five taps 4 * TH .. 4 * TH + 4: ring slot (hi plane) of my k-rows r_lo / r_hi, swizzle included; a K-step moves
    // it on by 32 rows = 256 slots (mod the ring)
    int I_lo[5], I_hi[5];
#pragma unroll
    for (int tl = 0; tl < 5; ++tl) {
      const int tap = 4 * TH + tl, dh = tap / 3, dw = tap - 3 * dh;
      const int sh = dh * PITCH + dw;
      I_lo[tl] = ((sh + r_lo) & (WP_RING - 1)) * 8 + wp_slot(sh + r_lo, ct * 4 + lp);
      I_hi[tl] = ((sh + r_hi) & (WP_RING - 1)) * 8 + wp_slot(sh + r_hi, ct * 4 + lp);
    }
    f32x4 acc[5][9];
#pragma unroll
    for (int tl = 0; tl < 5; ++tl)
#pragma unroll
      for (int m = 0; m < 9; ++m)
        if (wp_mine<TH>(tl, m)) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) acc[tl][m][rr] = 0.f;
        }

    __syncthreads();                                    // the prologue's data is staged
    for (int it = 0; it < NI; ++it) {
#pragma unroll
      for (int hk = 0; hk < 2; ++hk) {                 // the interval's two K-steps
        const uint2* A = Ad + (2 * (it & 1) + hk) * WP_AIMG;
        auto a_frag = [&](int pl, int m) __attribute__((always_inline)) -> s16x8 {
          return tr2(A + (2 * m + pl) * 128 + a_lo, A + (2 * m + pl) * 128 + a_hi);
        };
        s16x8 b[5][2];
#if WP_DIAG & 2
        {       // (diagnostic: one fragment read per K-step keeps the staging stores alive, no products)
          const s16x8 v0 = tr2(Xr[0] + I_lo[0], Xr[1] + I_hi[0]), v1 = a_frag(0, 0), v2 = a_frag(1, 8);
          asm volatile("" :: "v"(v0), "v"(v1), "v"(v2));
        }
#else
#pragma unroll
        for (int tl = 0; tl < 5; ++tl) {
          b[tl][0] = tr2(Xr[0] + I_lo[tl], Xr[0] + I_hi[tl]);
          b[tl][1] = tr2(Xr[1] + I_lo[tl], Xr[1] + I_hi[tl]);
          I_lo[tl] = (I_lo[tl] + 256) & (WP_RING * 8 - 1);
          I_hi[tl] = (I_hi[tl] + 256) & (WP_RING * 8 - 1);
        }
        // One register set for the dY fragment: the lo plane of row tile m + 1 is requested as soon as tile m's lo * hi
        // products are issued, its hi plane behind tile m's last product (the five lo * hi products of m + 1 cover it).
        s16x8 f_lo = a_frag(1, 0), f_hi = a_frag(0, 0);
#pragma unroll
        for (int m = 0; m < 9; ++m) {
          // lo * hi, hi * lo, hi * hi (smallest first); the taps inner, so that dependent products are five apart
#define CSTP_MM(A_, Q)                                                                                              \
  _Pragma("unroll") for (int tl = 0; tl < 5; ++tl)                                                                  \
    if (wp_mine<TH>(tl, m)) {                                                                                       \
      if (WP_DIAG & 8) asm volatile("" :: "v"(A_), "v"(b[tl][Q]));                                                  \
      else acc[tl][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A_), __builtin_bit_cast(f16x8, b[tl][Q]), acc[tl][m], 0, 0, 0); \
    }
          CSTP_MM(f_lo, 0)
          if (m + 1 < 9) f_lo = a_frag(1, m + 1);
          CSTP_MM(f_hi, 1)
          CSTP_MM(f_hi, 0)
          if (m + 1 < 9) f_hi = a_frag(0, m + 1);
#undef CSTP_MM
        }
#endif
      }
      __syncthreads();
    }

    // ---- epilogue: C layout col (c) = lane & 15, row (m) = (lane >> 4) * 4 + reg; one add per value, once per block
    const bool det = det_stride != 0;
    float* slab = dwp + (size_t)split * det_stride;
#pragma unroll
    for (int tl = 0; tl < 5; ++tl) {
      const int j = (4 * TH + tl) * g.Cp + cb * 32 + ct * 16 + li;
#pragma unroll
      for (int m = 0; m < 9; ++m) {
        if (!wp_mine<TH>(tl, m)) continue;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int mg = m0 + m * 16 + grp * 4 + rr;
          if (mg < g.M) wgrad_out(&slab[(size_t)mg * g.Jp + j], acc[tl][m][rr], det);
        }
      }
    }
  };
  if ((wave >> 1) == 0) body(std::integral_constant<int, 0>{});
  else body(std::integral_constant<int, 1>{});
}

}  // namespace cstp
