// K1T  igemm_k1t<MT>: forward / data gradient of the stride-1 3x1x1 TEMPORAL convolutions with the input patch resident in LDS --
// igemm_k1p's scheme (igemm_patch.h: persistent blocks, packed weights by LDS-DMA into a ring of three K-tiles, consumers 2 x 2
// with the transposed accumulator tile, BatchNorm sums / range from the epilogue) turned by ninety degrees: a block owns a tile of
// 8 FRAMES x 28 COLUMNS = 224 output positions of one clip and keeps, per 32-channel block of the gathered tensor, those 28
// columns of the 8 frames plus one halo frame before and after -- 10 x 28 = 280 LDS rows of split f16 pairs.  A temporal tap is
// then a shift of 28 LDS rows: one gather + one split of 1.25 x the tile (1.0 x when the clip has 8 frames: the halo frames are
// the zero padding) serve the three taps, where the gather kernel igemm_k1s fetches and splits every element once per tap.
//
// What the temporal layers change against igemm_k1p:
//  * three K-tiles per channel block instead of nine, so the staging waves have a third of the time per staged element: they
//    load FOUR consecutive columns of a channel per instruction (16 bytes per lane -- a frame row of the tile is 112 contiguous
//    bytes) and a lane assembles the 8-channel chunks of its four LDS rows from eight such loads.  140 (row quad, 8-channel
//    group) tasks per wave and channel block = three rounds of 64 lanes; round r of the channel block after next is loaded at
//    K-tile r right after round r of the next one left its registers for LDS: 24 loads of 16 bytes in flight per lane.
//  * AFF (forward): the gathered tensor is z = relu(src * scale + shift), the train-mode BatchNorm + ReLU between the spatial and
//    the temporal convolution (r21d_byol.py:94-97), applied ONCE per element between load and split (igemm_k1s<.., AFF>
//    pays it once per tap); the (scale, shift) tables of the (at most two) BatchNorm groups sit in LDS.  Halo frames are
//    out-of-range loads (0) and stay 0 through the clamp med3(v, 0, cap) with cap = 0 there.
//  * no ragged tiles (host: frames a multiple of 8, frame size a multiple of 28): every position of every tile is valid.
#pragma once

#include <type_traits>

namespace cstp {

constexpr int KT_DT = 8, KT_WT = 28;                 // tile: frames x columns (KT_DT * KT_WT == KP_NPOS)
constexpr int KT_ROWS = (KT_DT + 2) * KT_WT;         // 280 LDS rows per patch buffer
constexpr int KT_QUADS = KT_ROWS / 4;                // 70 quads of rows (four consecutive columns of one frame)
constexpr int KT_TASKS = 2 * KT_QUADS;               // per staging wave and channel block: (quad, 8-channel group)
constexpr int KT_NR = 3;                             // rounds of 64 tasks
constexpr int KT_AFFC = 576;                         // AFF: channels the LDS tables hold (two groups)
static_assert(KT_DT * KT_WT == KP_NPOS && KT_WT % 4 == 0 && KT_TASKS <= 64 * KT_NR, "tile shape");

struct TGeom {
  int Cs;           // channels of the gathered tensor (a multiple of 16)
  int ncb;          // its 32-channel blocks
  int D, HW, Nb;    // frames per clip, frame size, clips
  int M;            // valid output rows (channels of `out`)
  int nwt, ndt;     // tiles per frame row (HW / 28) and per clip depth (D / 8)
  int gclips;       // STATS: clips per BatchNorm group of the OUTPUT
  int groups;       // STATS: its groups (<= 2)
  int acc;          // out += instead of out =
  int aff_npg;      // AFF: clips per BatchNorm group of the INPUT transform
  int aff_groups;   // AFF: its groups (<= 2)
  int aff_relu;
};

template <int MT, bool STATS = false, bool AFF = false>
__global__ void __launch_bounds__(512, 2)
igemm_k1t(const TGeom g, const uint4* __restrict__ wpk, const float* __restrict__ src, float* __restrict__ out,
          const float* __restrict__ inv_a, const unsigned* __restrict__ bcell, int ntiles, int nmblk, double* __restrict__ part,
          const float* __restrict__ pivot, unsigned* __restrict__ zcell, const float2* __restrict__ aff_ss) {
  constexpr int BM = 16 * MT;
  constexpr int A_U4 = BM * 8;                       // uint4 per packed K-tile
  constexpr int A_DMA = BM / 8;                      // 1 KiB LDS-DMA pieces per K-tile
  constexpr int P_U4 = KT_ROWS * 8;
  constexpr int RING = KP_RING6 && MT == 4 ? 6 : 3;   // weight ring: six K-tiles at 64 rows (igemm_k1p explains)
  __shared__ uint4 smem[RING * A_U4 + 2 * P_U4 + 2 * (BM / 4) + (STATS ? BM + BM / 2 : 0)];
  __shared__ __attribute__((aligned(16))) float aff_a[AFF ? 2 * KT_AFFC : 4], aff_b[AFF ? 2 * KT_AFFC : 4];
  uint4* const ring = smem;
  uint4* const patch = smem + RING * A_U4;
  float* const inva_s = reinterpret_cast<float*>(smem + RING * A_U4 + 2 * P_U4);      // [2][BM], by item parity
  double* const stat_s = reinterpret_cast<double*>(smem + RING * A_U4 + 2 * P_U4 + 2 * (BM / 4));   // STATS: [BM][2] sums of the block's group
  unsigned* const mm_s = reinterpret_cast<unsigned*>(smem + RING * A_U4 + 2 * P_U4 + 2 * (BM / 4) + BM);   // STATS: [BM][2] range keys

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

  // persistent blocks, work item = (position tile, row block), XCD-aware order: exactly igemm_k1p's
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  // STATS with two BatchNorm groups: slots dealt to the groups alternately, a block walks tiles of its group only (igemm_k1p)
  const int ngrp = STATS ? g.groups : 1;
  const int bgrp = slot % ngrp, gslot = slot / ngrp, gnslots = nslots / ngrp;
  const int gtiles = ntiles / ngrp;
  const int chunk = (gtiles + 7) >> 3;
  int tiles_x = gtiles - xcd * chunk;
  tiles_x = tiles_x < 0 ? 0 : (tiles_x < chunk ? tiles_x : chunk);
  const int cnt_x = tiles_x * nmblk;
  const int nitems = gslot < cnt_x ? (cnt_x - gslot + gnslots - 1) / gnslots : 0;
  const int st_mblk = gslot % nmblk, st_nsplit = gnslots * 8 / nmblk, st_j = xcd * (gnslots / nmblk) + gslot / nmblk;
  unsigned* const mmk = STATS ? reinterpret_cast<unsigned*>(part + (size_t)g.M * g.groups * st_nsplit * 2 + g.M) : nullptr;
  auto write_part = [&](bool zeros) __attribute__((always_inline)) {
    for (int e = threadIdx.x; e < BM * 2; e += 256) {
      const int k = e & 1, row = e >> 1;
      const int ch = st_mblk * BM + row;
      if (ch < g.M) {
        const size_t at = (((size_t)ch * g.groups + bgrp) * st_nsplit + st_j) * 2 + k;
        part[at] = zeros ? 0.0 : stat_s[row * 2 + k];
        mmk[at] = zeros ? (k == 0 ? 0xffffffffu : 0u) : mm_s[row * 2 + k];
      }
    }
    if (st_j == 0 && bgrp == 0) {
      for (int row = threadIdx.x; row < BM; row += 256) {
        const int ch = st_mblk * BM + row;
        if (ch < g.M) part[(size_t)g.M * g.groups * st_nsplit * 2 + ch] = pivot != nullptr ? (double)pivot[ch] : 0.0;
      }
    }
  };
  if constexpr (STATS) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && zcell != nullptr) *zcell = 0;
  }
  if (nitems == 0) {
    if constexpr (STATS) { if (threadIdx.x < 256) write_part(true); }
    return;
  }
  auto item_of = [&](int it, int& tile, int& mblk) __attribute__((always_inline)) {
    const int idx = gslot + it * gnslots;
    const int t_in = idx / nmblk;
    mblk = idx - t_in * nmblk;
    tile = bgrp * gtiles + xcd * chunk + t_in;
  };
  // tile -> (clip, first frame, first column); the column tiles of a frame row are neighbours in the order (shared cache lines)
  auto tile_at = [&](int tile, int& nb, int& d0, int& hw0) __attribute__((always_inline)) {
    const int wt = tile % g.nwt, rest = tile / g.nwt;
    const int dt = rest % g.ndt;
    nb = rest / g.ndt; d0 = dt * KT_DT; hw0 = wt * KT_WT;
  };

  const int HW = g.HW;
  const int nkt = g.ncb * 3;
  const size_t chs = (size_t)g.D * HW;               // channel stride of src / row stride of out (elements)

  if constexpr (AFF) {
    // the input transform's tables, both groups, once per block (read by the staging waves from their first store on: the
    // barrier behind the prologue orders it)
    for (int e = t; e < g.aff_groups * g.Cs; e += 512) {
      const int grp = e / g.Cs, c = e - grp * g.Cs;
      const float2 p = aff_ss[e];
      aff_a[grp * KT_AFFC + c] = p.x; aff_b[grp * KT_AFFC + c] = p.y;
    }
    __syncthreads();
  }

  if (wave == 4) {
    // ============================================ weight DMA wave (4) ============================================
    // All 1 KiB pieces of K-tile k + 2 are requested at K-tile k; this wave issues no other vector-memory instruction, so its
    // wait is COUNTED (the batch just issued stays in flight across the barrier).  (The patch loads below stay in flight for
    // six K-tiles: in a wave that also waited for weight pieces, the in-order counter would retire them K-tile by K-tile.)
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(wpk, (unsigned)((size_t)nmblk * nkt * A_U4 * 16));
    int d_it = 0, d_kt = 0, d_mblk, tl_unused, d_ring = 0;
    item_of(0, tl_unused, d_mblk);
    auto dma_next = [&]() __attribute__((always_inline)) {
      const unsigned so = (unsigned)((((size_t)d_mblk * nkt + d_kt) * A_U4) * 16);
      uint4* dst = ring + d_ring * A_U4;
#pragma unroll
      for (int piece = 0; piece < A_DMA; ++piece)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(dst + piece * 64), 16,
                                                 (unsigned)(lane * 16 + piece * 1024), so, 0, 0);
      d_ring = d_ring == RING - 1 ? 0 : d_ring + 1;   // (past the last item: the last K-tile again, into slots nobody reads)
      if (d_it + 1 < nitems || d_kt + 1 < nkt) {
        if (++d_kt == nkt) { d_kt = 0; ++d_it; item_of(d_it, tl_unused, d_mblk); }
      }
    };
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) dma_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int total = nitems * nkt;
    static_assert(A_DMA == 8 || A_DMA == 16 || A_DMA == 18, "the counted wait below lists the piece counts");
#pragma unroll 1
    for (int k = 0; k < total; ++k) {
      dma_next();                                     // K-tile k + RING - 1 -> the slot K-tile k - 1 was read from
      // (counted: everything but the youngest RING - 2 batches has landed = K-tile k + 1 is in LDS)
      if constexpr (A_DMA == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
      else if constexpr (A_DMA == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if constexpr (RING == 6) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  if (wave >= 5) {
    // ============================================ patch staging waves (5, 6, 7) ============================================
    // Staging task tk = (row quad tk % 70, 8-channel group tk / 70 of the 32-channel block): 280 tasks per channel block, 94
    // (93 for the last) per wave = a full round of 64 lanes and a round of 30.  The registers of a round hold channel block
    // X + 2 or X + 3 while block X is multiplied: at the first K-tile of X the first round of block X + 1 goes to LDS and the
    // loads of block X + 3 are issued into the registers it left, at the second K-tile the same for the second round -- 16
    // loads of 16 bytes in flight per lane, two channel blocks (six K-tiles) of latency.
    const int sw = wave - 5;
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_src = make_rsrc(src, (unsigned)((size_t)g.Nb * g.Cs * chs * 4));
    float sb, inv_unused;
    f16_scale(__builtin_amdgcn_readfirstlane(*bcell), sb, inv_unused);
    const unsigned ch4 = (unsigned)(chs * 4);
    constexpr int TPW = (4 * KT_QUADS + 2) / 3;        // 94 tasks per wave

    struct Task { int row, dl, j0, c8; bool ok; };
    auto task_of = [&](int tk, bool ok) __attribute__((always_inline)) -> Task {
      Task k;
      k.ok = ok && tk < 4 * KT_QUADS;
      const int tq = k.ok ? tk : 0;
      k.c8 = tq / KT_QUADS;
      const int quad = tq - k.c8 * KT_QUADS;
      k.row = 4 * quad;
      k.dl = k.row / KT_WT;
      k.j0 = k.row - k.dl * KT_WT;
      return k;
    };
    const Task tm = task_of(TPW * sw + lane, true);
    const Task tt = task_of(TPW * sw + 64 + lane, lane < TPW - 64);
    // the load stream runs ahead of the consumers by up to three channel blocks: its own (item, channel block) cursor
    int l_it = 0, l_cb = 0, l_grp = 0;
    unsigned l_vm = OOB, l_vt = OOB;
    auto set_item = [&](int it) __attribute__((always_inline)) {
      l_vm = OOB; l_vt = OOB;
      if (it < nitems) {
        int tile, mb_unused, nb, d0, hw0;
        item_of(it, tile, mb_unused);
        tile_at(tile, nb, d0, hw0);
        if (AFF) l_grp = nb / g.aff_npg;
        const int dm = d0 - 1 + tm.dl, dt = d0 - 1 + tt.dl;
        if (tm.ok && dm >= 0 && dm < g.D)
          l_vm = (unsigned)((((size_t)nb * g.Cs * g.D + dm) * HW + hw0 + tm.j0) * 4) + (unsigned)(tm.c8 * 8) * ch4;
        if (tt.ok && dt >= 0 && dt < g.D)
          l_vt = (unsigned)((((size_t)nb * g.Cs * g.D + dt) * HW + hw0 + tt.j0) * 4) + (unsigned)(tt.c8 * 8) * ch4;
      }
    };
    struct Round { u32x4 v[8]; int cb; int grp; float cap; };
    auto load_round = [&](const Task& k, unsigned voff, Round& rd) __attribute__((always_inline)) {
      // (channels past the tensor's last one: whole 16-channel groups -- zeros against zero weights)
      const bool have = l_cb * 32 + k.c8 * 8 < g.Cs;
      const unsigned vo = have ? voff : OOB;
      const unsigned so = (unsigned)(l_cb * 32) * ch4;
#pragma unroll
      for (int e = 0; e < 8; ++e) buf_load_x4(rd.v[e], vo, rs_src, so + (unsigned)e * ch4);
      rd.cb = l_cb; rd.grp = l_grp; rd.cap = vo != OOB ? __builtin_inff() : 0.f;
    };
    auto advance = [&]() __attribute__((always_inline)) {
      if (++l_cb == g.ncb) { l_cb = 0; ++l_it; set_item(l_it); }
    };
    auto store_round = [&](int buf, const Task& k, Round& rd) __attribute__((always_inline)) {
      float a[8], b[8];
      if constexpr (AFF) {
        int c0 = rd.cb * 32 + k.c8 * 8;
        c0 = c0 < g.Cs ? c0 : 0;
        const f32x4* ta = reinterpret_cast<const f32x4*>(aff_a + rd.grp * KT_AFFC + c0);
        const f32x4* tb = reinterpret_cast<const f32x4*>(aff_b + rd.grp * KT_AFFC + c0);
        const f32x4 a0 = ta[0], a1 = ta[1], b0 = tb[0], b1 = tb[1];
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = a0[e]; a[4 + e] = a1[e]; b[e] = b0[e]; b[4 + e] = b1[e]; }
      }
      const float lo = g.aff_relu ? 0.f : -rd.cap;
      // (the components through a float vector: indexing rd.v[e][i] directly made hipcc read component 0 for every i)
      f32x4 vf[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) vf[e] = __builtin_bit_cast(f32x4, rd.v[e]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {                   // my four LDS rows
        float z[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          z[e] = vf[e][i];
          if constexpr (AFF) z[e] = __builtin_amdgcn_fmed3f(__builtin_fmaf(z[e], a[e], b[e]), lo, rd.cap);
        }
        uint4 ph, pl;
        unsigned hh, ll;
        split2h(z[0], z[1], sb, hh, ll); ph.x = hh; pl.x = ll;
        split2h(z[2], z[3], sb, hh, ll); ph.y = hh; pl.y = ll;
        split2h(z[4], z[5], sb, hh, ll); ph.z = hh; pl.z = ll;
        split2h(z[6], z[7], sb, hh, ll); ph.w = hh; pl.w = ll;
        if (k.ok) {
          const int row = k.row + i;
          uint4* prow = patch + buf * P_U4 + row * 8;
          const int x7 = row & 7;
          prow[k.c8 ^ x7] = ph;
          prow[(4 + k.c8) ^ x7] = pl;
        }
      }
    };

    Round rm[2], rt[2];                                // by parity of the channel block they carry
    set_item(0);
    // ---- prologue: channel block 0 of the first item into buffer 0; blocks 1 and 2 requested
    load_round(tm, l_vm, rm[0]);
    load_round(tt, l_vt, rt[0]);
    store_round(0, tm, rm[0]);
    store_round(0, tt, rt[0]);
    advance();
    load_round(tm, l_vm, rm[1]);
    load_round(tt, l_vt, rt[1]);
    advance();
    load_round(tm, l_vm, rm[0]);
    load_round(tt, l_vt, rt[0]);
    advance();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- steady state.  No load is conditional (past the end: out-of-range offsets).
    const int total = nitems * g.ncb;
    auto cblock = [&](auto par_tag, int pbuf) __attribute__((always_inline)) {
      constexpr int PAR = decltype(par_tag)::value;    // parity of block X + 1 = the register set that goes to LDS
      store_round(pbuf ^ 1, tm, rm[PAR]);               // K-tile 0: first round
      load_round(tm, l_vm, rm[PAR]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      store_round(pbuf ^ 1, tt, rt[PAR]);               // K-tile 1: second round
      load_round(tt, l_vt, rt[PAR]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_barrier();                     // K-tile 2
      advance();
    };
    int pb = 0;
#pragma unroll 1
    for (int x = 0; x < total; x += 2) {
      cblock(std::integral_constant<int, 1>{}, pb);      // block x: block x + 1 (odd) goes to LDS
      pb ^= 1;
      if (x + 1 >= total) break;
      cblock(std::integral_constant<int, 0>{}, pb);
      pb ^= 1;
    }
    return;
  }

  // =================================================== consumers (igemm_k1p's) ===================================================
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;
  constexpr int NIF = MT / 2;
  const int mt0 = wm * NIF;
  const int qa0 = fk ^ (fr & 7), qa1 = (4 + fk) ^ (fr & 7);
  float invb, sc_unused;
  f16_scale(__builtin_amdgcn_readfirstlane(*bcell), sc_unused, invb);
  const int q = lane >> 4;

  if constexpr (STATS) {
    for (int e = t; e < BM * 2; e += 256) { stat_s[e] = 0.0; mm_s[e] = (e & 1) ? 0u : 0xffffffffu; }
  }
  __builtin_amdgcn_s_barrier();                        // the first item's prologue data is staged
  __builtin_amdgcn_s_setprio(2);

  auto body = [&](auto xj0_tag, auto xjn_tag) __attribute__((always_inline)) {
  constexpr int NI = NIF;
  constexpr int XJ0 = decltype(xj0_tag)::value, XJN = decltype(xjn_tag)::value;      // the shared row tile: my column tiles of it
  constexpr int XA = XJN > 0 ? 1 : 0;
  int pb = 0, slot3 = 0;
  for (int it = 0; it < nitems; ++it) {
    int tile, mblk, nb, d0, hw0;
    item_of(it, tile, mblk);
    tile_at(tile, nb, d0, hw0);
    float* const inva = inva_s + (it & 1) * BM;
    if (t < BM) inva[t] = inv_a[mblk * BM + t];       // read back in this item's epilogue, many barriers later

    f32x4 acc[NI][KP_NTW];
    f32x4 accx[XJN > 0 ? XJN : 1];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < KP_NTW; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int j = 0; j < (XJN > 0 ? XJN : 1); ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) accx[j][r] = 0.f;

    // LDS row of my position in each of my 7 column tiles at tap 0: the tile-local position itself (frame-major, 28 per
    // frame, the patch starts one frame earlier); a tap adds 28
    const int base0 = wn * (KP_NTW * 16) + fr;

    f16x8 bh[3], bl[3];
    const unsigned patch_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) uint4*)patch);
    auto b_addr = [&](int j, unsigned pbuf_bytes, int ts) __attribute__((always_inline)) -> unsigned {
      const int row = base0 + 16 * j + ts;
      const int qq = fk ^ (row & 7);
      return pbuf_bytes + (unsigned)(row * 8 + qq) * 16u;                      // the lo plane sits 4 chunks (64 bytes) away: ^ 64
    };
    auto issue_b = [&](f16x8& dh, f16x8& dl, unsigned addr) __attribute__((always_inline)) {
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "=&v"(dh), "=&v"(dl) : "v"(addr), "v"(addr ^ 64u));
      __builtin_amdgcn_sched_barrier(0);
    };
    issue_b(bh[0], bl[0], b_addr(0, patch_lds + pb * (P_U4 * 16), 0));
    issue_b(bh[1], bl[1], b_addr(1, patch_lds + pb * (P_U4 * 16), 0));
    const int arow0 = (mt0 * 16 + fr) * 8;
    unsigned addr_n = b_addr(2, patch_lds + pb * (P_U4 * 16), 0);

    int tap = 0;                                       // (slot3, the ring slot of the next K-tile, runs on across the items)
    auto ktile = [&](auto ph_tag) __attribute__((always_inline)) {
      constexpr int PH = decltype(ph_tag)::value;        // buffer of this K-tile's column tile 0
      const uint4* Ab = ring + slot3 * A_U4;
      const unsigned Bp = patch_lds + pb * (P_U4 * 16);
      const int ts = tap * KT_WT;
      int ntap = tap + 1, npb = pb;
      if (ntap == 3) { ntap = 0; npb ^= 1; }
      const unsigned Bn = patch_lds + npb * (P_U4 * 16);
      const int nts = ntap * KT_WT;

      f16x8 ah[NI + XA], al[NI + XA];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        ah[i] = __builtin_bit_cast(f16x8, Ab[arow0 + i * 128 + qa0]);
        al[i] = __builtin_bit_cast(f16x8, Ab[arow0 + i * 128 + qa1]);
      }
      if constexpr (XA != 0) {                            // the shared last row tile
        ah[NI] = __builtin_bit_cast(f16x8, Ab[((MT - 1) * 16 + fr) * 8 + qa0]);
        al[NI] = __builtin_bit_cast(f16x8, Ab[((MT - 1) * 16 + fr) * 8 + qa1]);
      }
#pragma unroll
      for (int j = 0; j < KP_NTW; ++j) {
        issue_b(bh[(PH + j + 2) % 3], bl[(PH + j + 2) % 3], addr_n);
        if (j + 3 < KP_NTW) addr_n = b_addr(j + 3, Bp, ts);
        else addr_n = b_addr(j + 3 - KP_NTW, Bn, nts);
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const f16x8 bhj = bh[(PH + j) % 3], blj = bl[(PH + j) % 3];
        const bool xj = XJN > 0 && j >= XJ0 && j < XJ0 + XJN;
        const int jx = xj ? j - XJ0 : 0;
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bhj, al[i], acc[i][j], 0, 0, 0);
        if (xj) accx[jx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bhj, al[NI + XA - 1], accx[jx], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(blj, ah[i], acc[i][j], 0, 0, 0);
        if (xj) accx[jx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(blj, ah[NI + XA - 1], accx[jx], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bhj, ah[i], acc[i][j], 0, 0, 0);
        if (xj) accx[jx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bhj, ah[NI + XA - 1], accx[jx], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      slot3 = slot3 == RING - 1 ? 0 : slot3 + 1;
      tap = ntap; pb = npb;
      __builtin_amdgcn_s_barrier();
    };
#pragma unroll 1
    for (int kt = 0; kt < nkt; kt += 3) {
      ktile(std::integral_constant<int, 0>{});
      ktile(std::integral_constant<int, 1>{});
      ktile(std::integral_constant<int, 2>{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue (igemm_k1p's: transposed tile, one 16-byte store per lane and column tile).  A lane's four positions
    // p0 .. p0 + 3 are columns j0 .. j0 + 3 of frame d0 + p0 / 28 (28 is a multiple of four: never across frames).
    // (KP_EPI_PERM, as in igemm_k1p: the values travel to lane 4 * channel + quad before they are stored)
    const int sq = KP_EPI_PERM ? (lane & 3) : q, sfr = KP_EPI_PERM ? (lane >> 2) : fr;
    const int perm_src = (16 * (lane & 3) + (lane >> 2)) * 4;
    size_t obase[KP_NTW];
#pragma unroll
    for (int j = 0; j < KP_NTW; ++j) {
      const int p0 = (wn * KP_NTW + j) * 16 + 4 * sq;
      const int dl = p0 / KT_WT, jj = p0 - dl * KT_WT;
      obase[j] = ((size_t)nb * g.M * g.D + d0 + dl) * HW + hw0 + jj;
    }
#pragma unroll
    for (int i = 0; i < NI + XA; ++i) {
      const int mrow = (i < NI ? (mt0 + i) : (MT - 1)) * 16 + fr;
      const float sc = inva[mrow] * invb;
      const int m = mblk * BM + mrow;
      const int ms = mblk * BM + (i < NI ? (mt0 + i) : (MT - 1)) * 16 + sfr;
      float* orow = out + (size_t)ms * chs;
      f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
      float vmin = __builtin_inff(), vmax = -__builtin_inff();
      float pv = 0.f;
      if constexpr (STATS) { if (pivot != nullptr && m < g.M) pv = pivot[m] / sc; }
#pragma unroll
      for (int j = (i < NI ? 0 : XJ0); j < (i < NI ? KP_NTW : XJ0 + XJN); ++j) {
        const f32x4 v = i < NI ? acc[i < NI ? i : 0][j] : accx[i < NI ? 0 : j - XJ0];
        f32x4 vs = v * sc;
#if KP_EPI_PERM
        {   // (the components through a plain struct: ext_vector component reads have miscompiled to component 0 here, see DESIGN)
            struct F4 { float a, b, c, d; };
            const F4 t4 = __builtin_bit_cast(F4, vs);
          const float p0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm_src, __builtin_bit_cast(int, t4.a)));
          const float p1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm_src, __builtin_bit_cast(int, t4.b)));
          const float p2 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm_src, __builtin_bit_cast(int, t4.c)));
          const float p3 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm_src, __builtin_bit_cast(int, t4.d)));
          vs = f32x4{p0, p1, p2, p3};
        }
#endif
        if (ms < g.M) {
          f32x4* dst = reinterpret_cast<f32x4*>(orow + obase[j]);
          *dst = g.acc ? *dst + vs : vs;
        }
        if constexpr (STATS) {
          const f32x4 dv = v - pv; s1 += dv; s2 += dv * dv;
          vmin = __builtin_fminf(__builtin_fminf(vmin, v[0]), v[1]); vmin = __builtin_fminf(__builtin_fminf(vmin, v[2]), v[3]);
          vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, v[0]), v[1]); vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, v[2]), v[3]);
        }
      }
      if constexpr (STATS) {
        float a = (s1[0] + s1[1]) + (s1[2] + s1[3]), b = (s2[0] + s2[1]) + (s2[2] + s2[3]);
        a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
        a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
        if (q == 0 && m < g.M) {
          double* dst = stat_s + mrow * 2;
          __hip_atomic_fetch_add(dst, (double)(a * sc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(dst + 1, (double)(b * sc) * (double)sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        vmin = __builtin_fminf(vmin, __shfl_xor(vmin, 16, 64)); vmax = __builtin_fmaxf(vmax, __shfl_xor(vmax, 16, 64));
        vmin = __builtin_fminf(vmin, __shfl_xor(vmin, 32, 64)); vmax = __builtin_fmaxf(vmax, __shfl_xor(vmax, 32, 64));
        if (q == 0 && m < g.M) {
          __hip_atomic_fetch_min(mm_s + mrow * 2, key_of_float(vmin * sc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_max(mm_s + mrow * 2 + 1, key_of_float(vmax * sc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
  }
  };
  using std::integral_constant;
  if constexpr (MT % 2 == 0) {
    body(integral_constant<int, 0>{}, integral_constant<int, 0>{});
  } else {
    if (wm == 0) body(integral_constant<int, 0>{}, integral_constant<int, 4>{});
    else body(integral_constant<int, 4>{}, integral_constant<int, KP_NTW - 4>{});
  }
  if constexpr (STATS) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    write_part(false);
  }
}

}  // namespace cstp
