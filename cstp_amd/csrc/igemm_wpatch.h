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
//   * dY: a [32 positions][144 channels] image per K-step (double buffered; the layout and the transposing fragment reads
//     of igemm_k2s -- the reduction axis is the memory-contiguous one);
//   * x : a RING of 256 stream rows x 32 channels ([row][hi plane 32 ch | lo plane], 8-byte pieces of 4 channels, piece
//     slot XOR-swizzled by the row so that the transposing reads of ANY row alignment and the staging stores are
//     bank-conflict free); 32 new rows enter per K-step, `lead` K-steps ahead of their first use, and each row serves all
//     nine taps (and both 16-channel column tiles) from LDS: x and dY are each split ONCE per (row block, channel block).
// A block owns one (144-row block of dY channels, 32-channel block of x) pair and a contiguous range of frames, keeps the
// 144 x 32 x 9 accumulator tile in registers over the whole range (162 MFMA tiles over the four consumer waves: column tile x
// tap half, the middle tap shared 5 : 4 row tiles) and adds it into the packed slab once at the end -- 41 MB of atomics
// for the S1 layer where igemm_k2s wrote 132 MB.  Waves 4..7 are producers (global -> split -> LDS, two K-steps of loads in flight, no
// conditional load), waves 0..3 consumers; one barrier per K-step.
#pragma once

#ifndef WP_DIAG
#define WP_DIAG 0     // timing-only diagnostic builds (wrong results): bit 0 = producers only synchronise after the prologue,
#endif                // bit 1 = consumers only synchronise

namespace cstp {

constexpr int WP_RING = 256;          // x ring rows (8 segments of 32)
constexpr int WP_BM = 144;            // dY channels per block (9 MFMA row tiles)
#ifndef WP_NS
#define WP_NS 4                       // producer register sets = K-steps of global loads in flight (even)
#endif

struct WPGeom {
  int C, M;          // channels of x / of dY
  int ncb, nmblk;    // 32-channel blocks of x, 144-row blocks of dY
  int H, W, D, NF;   // frame size, frames per clip, frames in total
  int lead;          // K-steps the x staging runs ahead: 32 * lead >= 2 * (W + 2) + 34
  int nsplit, fper;  // blocks per (row block, channel block) pair, frames per block
  int Jp, Cp;        // slab row pitch; channels per tap in the slab's column order j = tap * Cp + c
  unsigned mg_pitch, mg_hp1, mg_d;      // 2^32 / (W + 2), / (H + 1), / D, rounded up: division by multiplication
};

// x ring: one array per plane, 64-byte rows of eight 8-byte pieces (4 channels each).  Slot of piece `pi` in ring row `row`:
// the 4-piece halves swap with bit 3 of the row (rows r and r + 8 of a transposing read share a 16-bank window) and the
// pieces inside a half permute with bits 2 and 4 (the 8 rows r + 4 k of a staging store share a window): checked
// exhaustively, the 32 pieces a half-wave's ds_read_b64_tr_b16 touches -- rows r..r+3 and r+8..r+11 of one half, ANY r --
// cover all 64 banks once, and so do the 32 rows x one piece of a staging store.  The swizzle reads bits 2..4 of the row
// only, which a K-step's 32 rows do not change, and the lo plane sits at a constant offset from the hi plane.
__device__ __forceinline__ int wp_slot(int row, int pi) {
  return pi ^ (((row >> 3) & 1) << 2) ^ (((row >> 2) & 1) | (((row >> 4) & 1) << 1));
}

template <int TH>      // tap half of a consumer wave (compile-time so that every accumulator index is a constant)
__device__ __forceinline__ bool wp_mine(int tl, int m) { return TH == 0 ? (tl < 4 || m < 5) : (tl > 0 || m >= 5); }

__global__ void __launch_bounds__(512, 2)
igemm_k2p(const WPGeom g, const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dwp,
          const unsigned* __restrict__ xcell, const unsigned* __restrict__ dycell, size_t det_stride) {
  __shared__ uint2 Am[2][2][32 * 32];      // dY rows 0..127: [buffer][plane][k-row][piece]
  __shared__ uint2 Ax[2][2][32 * 4];       // dY rows 128..143
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
  const int KS = (F * HP1 * PITCH + 31) >> 5;           // K-steps of this block's dY stream
  const int m0 = mblk * WP_BM;

  auto pc = [](int r, int c4) __attribute__((always_inline)) -> int {      // igemm_k2s's dY image swizzle
    return c4 ^ (4 * ((r & 3) | (((r >> 3) & 1) << 2))) ^ (2 * ((r >> 2) & 1));
  };
  auto prow = [](int r) __attribute__((always_inline)) -> int {
    return (r & 3) | (((r >> 3) & 1) << 2) | (((r >> 2) & 1) << 3) | (r & 16);
  };

  if (wave >= 4) {
    // ================================================= producers =================================================
    const int tp_ = t - 256;
    const int r = tp_ & 31, q = tp_ >> 5;             // stream row inside a K-step, slot 0..7
    constexpr unsigned OOB = 0x80000000u;
    const unsigned S4 = (unsigned)D * (unsigned)HW * 4u;      // channel stride in bytes
    const int Nb = g.NF / D;
    const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(dy, (unsigned)((size_t)Nb * g.M * D * HW * 4));
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x, (unsigned)((size_t)Nb * g.C * D * HW * 4));
    // loop-invariant per-thread offsets: my 16 (+2) dY rows, my 4 x channels (rows / channels past the end are CLAMPED:
    // finite values whose products land in slab cells nobody reads).  They ride in the loads' VECTOR offset: a wave holds two
    // slots q, and a non-uniform scalar offset makes the compiler wrap every load in a readfirstlane loop.
    unsigned moff[16], mxoff[2], coff[4];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      int m = m0 + 16 * q + j;
      m = m < g.M ? m : g.M - 1;
      moff[j] = (unsigned)m * S4;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int m = m0 + 128 + 2 * q + j;
      m = m < g.M ? m : g.M - 1;
      mxoff[j] = (unsigned)m * S4;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int c = cb * 32 + 4 * q + j;
      c = c < g.C ? c : g.C - 1;
      coff[j] = (unsigned)c * S4;
    }
    float sc_x, sc_dy, inv_unused;
    f16_scale(__builtin_amdgcn_readfirstlane(*xcell), sc_x, inv_unused);
    f16_scale(__builtin_amdgcn_readfirstlane(*dycell), sc_dy, inv_unused);

    // stream row -> (column, line of the frame's span, frame) by multiplication with 2^32 / divisor (exact below 2^25 rows for
    // divisors <= 128; the host checks): no cursor state, no branches -- a version that stepped wave-uniform cursors with
    // `while` loops spent ~50 scalar branches per K-step and made the producers, alone, slower than the consumers (0.92 vs
    // 0.49 ms on the S1 layer)
    int kd = 0, kx = 0;                                 // next dY K-step / x segment to request
    // my row of K-step k -> byte offset of channel 0 of dY there, or OOB (halo, past the end)
    auto dy_base = [&](int k) __attribute__((always_inline)) -> unsigned {
      const unsigned sr = (unsigned)(k * 32 + r);
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
      const unsigned sr = (unsigned)(k * 32 + r);
      const unsigned L = __umulhi(sr, g.mg_pitch), c = sr - L * PITCH;
      const unsigned fl = __umulhi(L, g.mg_hp1), ll = L - fl * HP1;
      const bool prev = ll == 0;                        // (fl >= 1 is checked below: f_begin + fl - 1 does not wrap then)
      const unsigned f = f_begin + fl - (prev ? 1u : 0u), nb = D == 1 ? f : __umulhi(f, g.mg_d), d = f - nb * D;
      const unsigned h = prev ? (unsigned)(H - 1) : ll - 2;
      const bool ok = c >= 2 && (prev ? (fl >= 1 && fl <= (unsigned)F) : (ll >= 2 && fl < (unsigned)F));
      if (WP_DIAG & 4) return OOB;
      return ok ? ((nb * g.C * D + d) * HW + h * W + (c - 2)) * 4u : OOB;
    };

    auto load_x = [&](float (&v)[4]) __attribute__((always_inline)) {
      const unsigned b = x_base(kx++);
#pragma unroll
      for (int j = 0; j < 4; ++j) buf_load_x1(v[j], b + coff[j], rs_x, 0);
    };
    // x ring: my 4 channels of stream row 32 * seg + r
    auto store_x = [&](int seg, const float (&v)[4]) __attribute__((always_inline)) {
      const int row = ((seg << 5) + r) & (WP_RING - 1);
      const int s = row * 8 + wp_slot(row, q);
      unsigned h0, l0, h1, l1;
      split2h(v[0], v[1], sc_x, h0, l0);
      split2h(v[2], v[3], sc_x, h1, l1);
      Xr[0][s] = make_uint2(h0, h1);
      Xr[1][s] = make_uint2(l0, l1);
    };
    struct Regs { float a[16], ax[2], b[4]; };
    auto issue = [&](Regs& R) __attribute__((always_inline)) {       // the next item: dY of a K-step + one x segment
      const unsigned bd = dy_base(kd++);
#pragma unroll
      for (int j = 0; j < 16; ++j) buf_load_x1(R.a[j], bd + moff[j], rs_dy, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) buf_load_x1(R.ax[j], bd + mxoff[j], rs_dy, 0);
      load_x(R.b);
    };
    const int a_slot = r * 32 + pc(r, 4 * q);
    const int a_half = (r >> 2) & 1;
    const int x_slot = prow(r) * 4 + (q >> 1);
    auto store = [&](int buf, int item, const Regs& R) __attribute__((always_inline)) {
      uint4 ph[2], pl[2];
      unsigned hh, ll;
#define CSTP_SPLITH(J, DST, F_) split2h(R.a[J], R.a[(J) + 1], sc_dy, hh, ll); ph[DST].F_ = hh; pl[DST].F_ = ll;
      CSTP_SPLITH(0, 0, x) CSTP_SPLITH(2, 0, y) CSTP_SPLITH(4, 0, z) CSTP_SPLITH(6, 0, w)
      CSTP_SPLITH(8, 1, x) CSTP_SPLITH(10, 1, y) CSTP_SPLITH(12, 1, z) CSTP_SPLITH(14, 1, w)
#undef CSTP_SPLITH
      uint4* d0 = reinterpret_cast<uint4*>(Am[buf][0] + (a_slot & ~3));
      uint4* d1 = reinterpret_cast<uint4*>(Am[buf][1] + (a_slot & ~3));
      d0[a_half] = ph[0]; d0[a_half ^ 1] = ph[1];
      d1[a_half] = pl[0]; d1[a_half ^ 1] = pl[1];
      split2h(R.ax[0], R.ax[1], sc_dy, hh, ll);
      reinterpret_cast<unsigned*>(&Ax[buf][0][x_slot])[q & 1] = hh;
      reinterpret_cast<unsigned*>(&Ax[buf][1][x_slot])[q & 1] = ll;
      store_x(item + g.lead - 1, R.b);
    };

    // ---- prologue: x segments 0 .. lead - 2, then item 0 (dY of K-step 0 + x segment lead - 1)
    {
      float px[6][4];
#pragma unroll
      for (int s = 0; s < 6; ++s)
        if (s < g.lead - 1) load_x(px[s]);
#pragma unroll
      for (int s = 0; s < 6; ++s)
        if (s < g.lead - 1) store_x(s, px[s]);
    }
    // WP_NS register sets = items in flight: while the consumers multiply K-step i the producers store item i + 1 and request
    // item i + 1 + WP_NS.  (Two sets, as in igemm_k2s, left the kernel at 1.24 ms on the S1 layer with the consumers alone
    // needing 0.49: a K-step here is ~1.2 us and HBM answers in ~2 under load.)
    Regs R[WP_NS];
#pragma unroll
    for (int s = 0; s < WP_NS; ++s) issue(R[s]);
    store(0, 0, R[0]);
    issue(R[0]);
    __syncthreads();
    for (int i = 0; i < KS; i += WP_NS) {
#pragma unroll
      for (int s = 0; s < WP_NS; ++s) {
        if (i + s >= KS) return;
#if !(WP_DIAG & 1)
        store((s + 1) & 1, i + s + 1, R[(s + 1) % WP_NS]);
        issue(R[(s + 1) % WP_NS]);
#endif
        __syncthreads();
      }
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
  // dY fragments: slots inside one plane of one buffer (uint2 units), row tile mt adds pc(.., mt * 4 + lp)
  const int a_lo_row = r_lo * 32, a_hi_row = r_hi * 32;
  const int ax_lo = prow(r_lo) * 4 + lp, ax_hi = prow(r_hi) * 4 + lp;

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
    // dY fragment slots of row tile m inside one plane (a compile-time function of m up to the lane's rows)
    auto a_frag = [&](int buf, int pl, int m) __attribute__((always_inline)) -> s16x8 {
      if (m < 8) {
        const uint2* A = Am[buf][pl];
        return tr2(A + a_lo_row + pc(r_lo, m * 4 + lp), A + a_hi_row + pc(r_hi, m * 4 + lp));
      }
      return tr2(Ax[buf][pl] + ax_lo, Ax[buf][pl] + ax_hi);
    };

    __syncthreads();                                    // the prologue's data is staged
    int buf = 0;
    for (int ks = 0; ks < KS; ++ks) {
      s16x8 b[5][2];
#if WP_DIAG & 2
      {       // (diagnostic: one fragment read per K-step keeps the staging stores alive, no products)
        const s16x8 v0 = tr2(Xr[0] + I_lo[0], Xr[1] + I_hi[0]), v1 = a_frag(buf, 0, 0), v2 = a_frag(buf, 1, 8);
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
      s16x8 a_lo = a_frag(buf, 1, 0), a_hi = a_frag(buf, 0, 0);
#pragma unroll
      for (int m = 0; m < 9; ++m) {
        // lo * hi, hi * lo, hi * hi (smallest first); the taps inner, so that dependent products are five apart
#define CSTP_MM(A_, Q)                                                                                              \
  _Pragma("unroll") for (int tl = 0; tl < 5; ++tl)                                                                  \
    if (wp_mine<TH>(tl, m))                                                                                         \
      acc[tl][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A_), __builtin_bit_cast(f16x8, b[tl][Q]), acc[tl][m], 0, 0, 0);
        CSTP_MM(a_lo, 0)
        if (m + 1 < 9) a_lo = a_frag(buf, 1, m + 1);
        CSTP_MM(a_hi, 1)
        CSTP_MM(a_hi, 0)
        if (m + 1 < 9) a_hi = a_frag(buf, 0, m + 1);
#undef CSTP_MM
      }
#endif
      __syncthreads();
      buf ^= 1;
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
