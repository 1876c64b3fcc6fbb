// K1W  igemm_k1w<MT, STATS, AFF>: forward of the stride-1 3x1x1 TEMPORAL convolutions whose WHOLE packed weight matrix fits in LDS
// next to one input patch -- the first stage's layers (144 -> 64 channels at 16 x 56 x 56: 15 K-tiles of 64 rows = 120 KB), which are
// HBM-bound (66 FLOP per byte: 1.34 GB per launch against 0.11 ms of matrix work) and were the layer furthest below that bound.
//
// igemm_k1t (igemm_tpatch.h) streams a weight K-tile per barrier through a ring by LDS-DMA.  With 64 output rows a K-tile is 42
// products per consumer wave = 672 matrix cycles, but took ~2 400: every K-tile waited for a weight piece requested two (short)
// K-tiles earlier and for a barrier, so the kernel lost to the gather kernel igemm_k1s (0.53 against 0.45 ms, round 3).  Here
//  * the packed weights (igemm_k1t's format, one row block) are brought into LDS ONCE per persistent block and stay: no weight
//    traffic, no DMA wave and no weight wait in the steady state;
//  * the patch (8 + 2 frames x 28 columns x 32 channels, split f16 pairs: 35 KB) has ONE buffer: the consumers multiply the
//    three K-tiles of a channel block without any barrier, then barrier A ("patch read"), the staging waves store the next
//    channel block -- transformed and split in registers BEFORE the barrier, beside the consumers' products (its loads were issued
//    two channel blocks earlier) -- barrier B ("patch written").  The consumers idle between A and B for eight LDS stores per
//    staging lane; at the end of an item they run the EPILOGUE there;
//  * 16-byte staging loads, BatchNorm + ReLU of the input once per staged element (AFF), BatchNorm sums / range of the output
//    from the epilogue (STATS), persistent XCD-ordered grid, transposed accumulator tile: igemm_k1t's.
#pragma once

#include <type_traits>

namespace cstp {

constexpr int KW_AFFC = 160;                          // AFF: channels per group the LDS tables hold
constexpr int KW_NKT = 15;                            // K-tiles (32 channels x one tap) the resident weight image holds at 64 rows

static inline bool k1w_fits(int M, int Cs) { return M <= 64 && ((Cs + 31) / 32) * 3 <= KW_NKT; }

template <bool STATS, bool AFF>
__global__ void __launch_bounds__(512, 2)
igemm_k1w(const TGeom g, const uint4* __restrict__ wpk, const float* __restrict__ src, float* __restrict__ out,
          const float* __restrict__ inv_a, const unsigned* __restrict__ bcell, int ntiles, double* __restrict__ part,
          const float* __restrict__ pivot, unsigned* __restrict__ zcell, const float2* __restrict__ aff_ss) {
  constexpr int MT = 4, BM = 16 * MT;
  constexpr int A_U4 = BM * 8;                       // uint4 per packed K-tile
  constexpr int P_U4 = KT_ROWS * 8;
  __shared__ uint4 smem[KW_NKT * A_U4 + P_U4 + BM / 4 + (STATS ? BM + BM / 2 : 0)];
  __shared__ __attribute__((aligned(16))) float aff_a[AFF ? 2 * KW_AFFC : 4], aff_b[AFF ? 2 * KW_AFFC : 4];
  uint4* const wres = smem;
  uint4* const patch = smem + KW_NKT * A_U4;
  float* const inva_s = reinterpret_cast<float*>(smem + KW_NKT * A_U4 + P_U4);
  double* const stat_s = reinterpret_cast<double*>(smem + KW_NKT * A_U4 + P_U4 + BM / 4);                 // STATS: [BM][2]
  unsigned* const mm_s = reinterpret_cast<unsigned*>(smem + KW_NKT * A_U4 + P_U4 + BM / 4 + BM);          // STATS: [BM][2] keys

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

  // persistent blocks, work item = position tile (one row block), XCD-aware order, view groups dealt to the slots: igemm_k1t's
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int ngrp = STATS ? g.groups : 1;
  const int bgrp = slot % ngrp, gslot = slot / ngrp, gnslots = nslots / ngrp;
  const int gtiles = ntiles / ngrp;
  const int chunk = (gtiles + 7) >> 3;
  int tiles_x = gtiles - xcd * chunk;
  tiles_x = tiles_x < 0 ? 0 : (tiles_x < chunk ? tiles_x : chunk);
  const int nitems = gslot < tiles_x ? (tiles_x - gslot + gnslots - 1) / gnslots : 0;
  const int st_nsplit = gnslots * 8, st_j = xcd * gnslots + gslot;
  unsigned* const mmk = STATS ? reinterpret_cast<unsigned*>(part + (size_t)g.M * g.groups * st_nsplit * 2 + g.M) : nullptr;
  auto write_part = [&](bool zeros) __attribute__((always_inline)) {
    for (int e = threadIdx.x; e < BM * 2; e += 256) {
      const int k = e & 1, ch = e >> 1;
      if (ch < g.M) {
        const size_t at = (((size_t)ch * g.groups + bgrp) * st_nsplit + st_j) * 2 + k;
        part[at] = zeros ? 0.0 : stat_s[ch * 2 + k];
        mmk[at] = zeros ? (k == 0 ? 0xffffffffu : 0u) : mm_s[ch * 2 + k];
      }
    }
    if (st_j == 0 && bgrp == 0) {
      for (int ch = threadIdx.x; ch < BM; ch += 256)
        if (ch < g.M) part[(size_t)g.M * g.groups * st_nsplit * 2 + ch] = pivot != nullptr ? (double)pivot[ch] : 0.0;
    }
  };
  if constexpr (STATS) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && zcell != nullptr) *zcell = 0;
  }
  if (nitems == 0) {
    if constexpr (STATS) { if (threadIdx.x < 256) write_part(true); }
    return;
  }
  auto item_tile = [&](int it) __attribute__((always_inline)) -> int { return bgrp * gtiles + xcd * chunk + gslot + it * gnslots; };
  auto tile_at = [&](int tile, int& nb, int& d0, int& hw0) __attribute__((always_inline)) {
    const int wt = tile % g.nwt, rest = tile / g.nwt;
    const int dt = rest % g.ndt;
    nb = rest / g.ndt; d0 = dt * KT_DT; hw0 = wt * KT_WT;
  };

  const int HW = g.HW;
  const int nkt = g.ncb * 3;
  const size_t chs = (size_t)g.D * HW;               // channel stride of src / row stride of out (elements)

  // ---- once per block: the packed weights (all K-tiles of the one row block) into LDS by LDS-DMA, 1 KiB per wave instruction
  {
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(wpk, (unsigned)((size_t)nkt * A_U4 * 16));
    const int npieces = nkt * (A_U4 / 64);
    for (int p = wave; p < npieces; p += 8)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(wres + p * 64), 16,
                                               (unsigned)(lane * 16), (unsigned)(p * 1024), 0, 0);
  }
  if constexpr (AFF) {
    for (int e = t; e < g.aff_groups * g.Cs; e += 512) {
      const int grp = e / g.Cs, c = e - grp * g.Cs;
      const float2 p = aff_ss[e];
      aff_a[grp * KW_AFFC + c] = p.x; aff_b[grp * KW_AFFC + c] = p.y;
    }
  }
  if (t < BM) inva_s[t] = inv_a[t];
  if constexpr (STATS) {
    for (int e = t; e < BM * 2; e += 512) { stat_s[e] = 0.0; mm_s[e] = (e & 1) ? 0u : 0xffffffffu; }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                        // tables and weights are in place (the staging waves read the tables)

  if (wave == 4) return;                               // (eight waves bring the weights in; seven go on)

  if (wave >= 5) {
    // ============================================ patch staging waves (5, 6, 7): igemm_k1t's tasks ============================================
    const int sw = wave - 5;
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_src = make_rsrc(src, (unsigned)((size_t)g.Nb * g.Cs * chs * 4));
    float sb, inv_unused;
    f16_scale(__builtin_amdgcn_readfirstlane(*bcell), sb, inv_unused);
    const unsigned ch4 = (unsigned)(chs * 4);
    constexpr int TPW = (4 * KT_QUADS + 2) / 3;        // 94 tasks per wave: a round of 64 lanes and a round of 30

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
        int nb, d0, hw0;
        tile_at(item_tile(it), nb, d0, hw0);
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
      // (channels past the tensor's last one: whole 8-channel groups -- zeros against zero weights)
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
    // a round on its way into the patch, in two steps: prep_round turns the raw values into the packed (hi, lo) f16 pieces of
    // its four LDS rows IN REGISTERS -- all of the staging arithmetic (BatchNorm + ReLU, split) -- and runs while the consumers
    // still multiply the previous channel block; write_round is the eight 16-byte LDS stores, the only work left between the
    // barriers A and B (the consumers wait there).  (First version: everything between A and B, ~1 600 cycles per channel block
    // against ~2 000 of matrix work.)
    struct Packed { uint4 ph[4], pl[4]; };
    auto prep_round = [&](const Task& k, const Round& rd, Packed& pk) __attribute__((always_inline)) {
      float a[8], b[8];
      if constexpr (AFF) {
        int c0 = rd.cb * 32 + k.c8 * 8;
        c0 = c0 < g.Cs ? c0 : 0;
        const f32x4* ta = reinterpret_cast<const f32x4*>(aff_a + rd.grp * KW_AFFC + c0);
        const f32x4* tb = reinterpret_cast<const f32x4*>(aff_b + rd.grp * KW_AFFC + c0);
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
        pk.ph[i] = ph; pk.pl[i] = pl;
      }
    };
    auto write_round = [&](const Task& k, const Packed& pk) __attribute__((always_inline)) {
      if (k.ok) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = k.row + i;
          uint4* prow = patch + row * 8;
          const int x7 = row & 7;
          prow[k.c8 ^ x7] = pk.ph[i];
          prow[(4 + k.c8) ^ x7] = pk.pl[i];
        }
      }
    };

    Round rm[2], rt[2];                                // by parity of the channel block they carry
    set_item(0);
    // ---- prologue: channel block 0 of the first item into the patch; blocks 1 and 2 requested
    Packed pm, pt;
    load_round(tm, l_vm, rm[0]);
    load_round(tt, l_vt, rt[0]);
    prep_round(tm, rm[0], pm);
    prep_round(tt, rt[0], pt);
    write_round(tm, pm);
    write_round(tt, pt);
    advance();
    load_round(tm, l_vm, rm[1]);
    load_round(tt, l_vt, rt[1]);
    advance();
    load_round(tm, l_vm, rm[0]);
    load_round(tt, l_vt, rt[0]);
    advance();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // B(0): the patch holds channel block 0

    // ---- steady state, per channel block X:  barrier A (patch read) -> block X + 1 from its registers to the patch, block X + 3
    // requested into them -> barrier B.  No load is conditional (past the end: out-of-range offsets).
    const int total = nitems * g.ncb;
    auto boundary = [&](auto par_tag) __attribute__((always_inline)) {
      constexpr int PAR = decltype(par_tag)::value;    // parity of block X + 1
      prep_round(tm, rm[PAR], pm);                      // (beside the consumers' products of block X)
      prep_round(tt, rt[PAR], pt);
      __builtin_amdgcn_s_barrier();                     // A
      write_round(tm, pm);
      write_round(tt, pt);
      load_round(tm, l_vm, rm[PAR]);
      load_round(tt, l_vt, rt[PAR]);
      advance();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                     // B
    };
#pragma unroll 1
    for (int x = 0; x < total; x += 2) {
      boundary(std::integral_constant<int, 1>{});
      if (x + 1 >= total) break;
      boundary(std::integral_constant<int, 0>{});
    }
    return;
  }

  // =================================================== consumers ===================================================
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;
  constexpr int NI = MT / 2;
  const int mt0 = wm * NI;
  const int qa0 = fk ^ (fr & 7), qa1 = (4 + fk) ^ (fr & 7);
  float invb, sc_unused;
  f16_scale(__builtin_amdgcn_readfirstlane(*bcell), sc_unused, invb);
  const int q = lane >> 4;
  const int arow0 = (mt0 * 16 + fr) * 8;
  const int base0 = wn * (KP_NTW * 16) + fr;           // LDS row of my position in column tile 0 at tap 0
  __builtin_amdgcn_s_barrier();                        // B(0)
  __builtin_amdgcn_s_setprio(2);

  for (int it = 0; it < nitems; ++it) {
    int nb, d0, hw0;
    tile_at(item_tile(it), nb, d0, hw0);
    f32x4 acc[NI][KP_NTW];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < KP_NTW; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // B fragment of tile s = tap * 7 + j of the channel block: LDS row base0 + 16 j + 28 tap, chunk fk (hi plane) / 4 + fk (lo).
    // The reads are INLINE ASM two tiles ahead with hand-counted waits, as in igemm_k1p / igemm_k1t: left to the compiler each
    // pair was issued right in front of its first use (one exposed LDS latency per six products: the first build of this kernel
    // ran the layer in 0.51 ms, slower than the ring kernel).  LDS operations complete in order, so `lgkmcnt(4)` behind the issue
    // of tile s + 2 means tile s has landed; the compiler's own weight-fragment reads in between can only lengthen such a wait.
    const unsigned patch_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) uint4*)patch);
    auto b_addr = [&](int s) __attribute__((always_inline)) -> unsigned {
      const int row = base0 + 16 * (s % KP_NTW) + KT_WT * (s / KP_NTW);
      return patch_lds + (unsigned)(row * 8 + (fk ^ (row & 7))) * 16u;              // the lo plane sits 4 chunks (64 bytes) away: ^ 64
    };
    auto issue_b = [&](f16x8& dh, f16x8& dl, unsigned addr) __attribute__((always_inline)) {
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "=&v"(dh), "=&v"(dl) : "v"(addr), "v"(addr ^ 64u));
      __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    for (int cb = 0; cb < g.ncb; ++cb) {
      const uint4* Wk = wres + (size_t)cb * 3 * A_U4;
      f16x8 bh[3], bl[3], ah[2][NI], al[2][NI];
      issue_b(bh[0], bl[0], b_addr(0));
      issue_b(bh[1], bl[1], b_addr(1));
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        ah[0][i] = __builtin_bit_cast(f16x8, Wk[arow0 + i * 128 + qa0]);
        al[0][i] = __builtin_bit_cast(f16x8, Wk[arow0 + i * 128 + qa1]);
      }
      unsigned addr_n = b_addr(2);
#pragma unroll
      for (int s = 0; s < 3 * KP_NTW; ++s) {
        const int tap = s / KP_NTW, j = s % KP_NTW;
        if (s + 2 < 3 * KP_NTW) {
          issue_b(bh[(s + 2) % 3], bl[(s + 2) % 3], addr_n);
          if (s + 3 < 3 * KP_NTW) addr_n = b_addr(s + 3);
          asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        } else if (s + 1 < 3 * KP_NTW) {
          asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        const f16x8 bhj = bh[s % 3], blj = bl[s % 3];
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bhj, al[tap & 1][i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(blj, ah[tap & 1][i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bhj, ah[tap & 1][i], acc[i][j], 0, 0, 0);
        if (j == 3 && tap + 1 < 3) {                  // the next tap's weight fragments (resident: nothing to wait for but LDS)
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            ah[(tap + 1) & 1][i] = __builtin_bit_cast(f16x8, Wk[(tap + 1) * A_U4 + arow0 + i * 128 + qa0]);
            al[(tap + 1) & 1][i] = __builtin_bit_cast(f16x8, Wk[(tap + 1) * A_U4 + arow0 + i * 128 + qa1]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (every read of the patch has landed)
      __builtin_amdgcn_s_barrier();                     // A: the patch may be overwritten
      if (cb + 1 < g.ncb) __builtin_amdgcn_s_barrier(); // B: it holds the next channel block (after the last: behind the epilogue)
    }

    // ---- epilogue (igemm_k1t's), while the staging waves store the next item's first channel block
    const int sq = lane & 3, sfr = lane >> 2;
    const int perm_src = (16 * (lane & 3) + (lane >> 2)) * 4;
    size_t obase[KP_NTW];
#pragma unroll
    for (int j = 0; j < KP_NTW; ++j) {
      const int p0 = (wn * KP_NTW + j) * 16 + 4 * sq;
      const int dl = p0 / KT_WT, jj = p0 - dl * KT_WT;
      obase[j] = ((size_t)nb * g.M * g.D + d0 + dl) * HW + hw0 + jj;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int mrow = (mt0 + i) * 16 + fr;
      const float sc = inva_s[mrow] * invb;
      const int ms = (mt0 + i) * 16 + sfr;
      float* orow = out + (size_t)ms * chs;
      f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
      float vmin = __builtin_inff(), vmax = -__builtin_inff();
      float pv = 0.f;
      if constexpr (STATS) { if (pivot != nullptr && mrow < g.M) pv = pivot[mrow] / sc; }
#pragma unroll
      for (int j = 0; j < KP_NTW; ++j) {
        const f32x4 v = acc[i][j];
        f32x4 vs = v * sc;
        {   // (the components through a plain struct: ext_vector component reads have miscompiled to component 0 here, see DESIGN)
          struct F4 { float a, b, c, d; };
          const F4 t4 = __builtin_bit_cast(F4, vs);
          const float p0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm_src, __builtin_bit_cast(int, t4.a)));
          const float p1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm_src, __builtin_bit_cast(int, t4.b)));
          const float p2 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm_src, __builtin_bit_cast(int, t4.c)));
          const float p3 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm_src, __builtin_bit_cast(int, t4.d)));
          vs = f32x4{p0, p1, p2, p3};
        }
        if (ms < g.M) *reinterpret_cast<f32x4*>(orow + obase[j]) = vs;
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
        vmin = __builtin_fminf(vmin, __shfl_xor(vmin, 16, 64)); vmax = __builtin_fmaxf(vmax, __shfl_xor(vmax, 16, 64));
        vmin = __builtin_fminf(vmin, __shfl_xor(vmin, 32, 64)); vmax = __builtin_fmaxf(vmax, __shfl_xor(vmax, 32, 64));
        if (q == 0 && mrow < g.M) {
          double* dst = stat_s + mrow * 2;
          __hip_atomic_fetch_add(dst, (double)(a * sc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(dst + 1, (double)(b * sc) * (double)sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          // (sc > 0 and rounding is monotone: min(v) * sc IS the smallest stored output)
          __hip_atomic_fetch_min(mm_s + mrow * 2, key_of_float(vmin * sc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_max(mm_s + mrow * 2 + 1, key_of_float(vmax * sc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // B behind the item's last channel block
  }
  if constexpr (STATS) {
    // the four consumer waves are the block's only live waves here (the staging waves returned behind their last barrier)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    write_part(false);
  }
}

}  // namespace cstp
