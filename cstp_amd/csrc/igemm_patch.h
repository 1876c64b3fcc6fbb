// K1P  igemm_k1p<MT>: forward / data gradient of the stride-1 1x3x3 spatial convolutions with the INPUT PATCH RESIDENT IN LDS
// (north_star: "LDS-staged input tiles"), f16-pair arithmetic (igemm_split.h: three v_mfma_f32_16x16x32_f16 per fp32 product).
//
// The gather kernels igemm_k1s fetch every activation element once per filter tap (x9) and re-split it each time: the S1 layer
// (64 -> 144 channels at 16x56x56) moves 10.5 GB from the L2s into the CUs for 1.34 GB of HBM traffic (profiles/r01).  Here a
// block owns 224 consecutive output positions (2 wave columns x 7 MFMA column tiles of 16; 7 because every frame width of
// the network -- 56, 28, 14, 7 -- is a multiple of 7) and ALL 16*MT rows of one row block, and keeps, per 32-channel block
// `cb` of the gathered tensor, the whole input patch those positions touch -- their image rows plus one halo row above and below
// per frame, W + 2 columns -- in LDS as split f16 pairs.  A filter tap is then a constant shift of the LDS row index
// (dh * (W + 2) + dw): nine K-tiles are served by one gather + one split of ~1.5x the patch instead of nine.
//
// LDS image of the patch: one 128-byte row per (image row, column) = [hi plane: 32 channels f16][lo plane], 16-byte chunk L
// (L = 4 * plane + k / 8) at L ^ (row & 7) -- the layout of igemm_k1s<.., NP = 2>, conflict-free for the fragment reads of 16
// consecutive rows and for the staging stores.  Two patch buffers (cb, cb + 1).
//
// The packed weights of one (cb, tap) K-tile arrive by LDS-DMA (buffer_load ... lds, 1 KiB per wave instruction): the pack
// kernel writes them to global memory in exactly the swizzled LDS image order, so staging them costs the producers no VGPR, no
// VALU and no ds_write.  Ring of three K-tiles.
//
// Waves: 6, 7 stage the NEXT channel block's patch (gather + split + store, spread over the current block's nine taps); 4, 5 issue
// the weight DMA two K-tiles ahead (no other memory instruction, so their wait is counted); 0..3 consumers as 2 (rows) x 2 (columns): MT/2 x 7 accumulator tiles each, an odd last row tile shared by all four.  One raw
// s_barrier per K-tile.  Blocks are PERSISTENT (one per CU) and the K-tile sequence runs on across a block's work items, so
// only the first item of a block has an exposed prologue; the consumers fetch activation fragments two column tiles ahead (also
// the next K-tile's, which are in LDS long before), so only the weight fragments wait for the barrier.
#pragma once

#include <type_traits>

#ifndef KP_RING6
#define KP_RING6 0    // 1: six-slot weight ring at 64 rows (measured neutral: 58.19 vs 58.11 ms/step -- the weight DMA is not what the short K-tiles wait for)
#endif
#ifndef KP_GROUP3
#define KP_GROUP3 0   // 1: 64-row tiles run THREE K-tiles (one filter row) per barrier through a six-slot weight ring -- a K-tile of 42 products
#endif                // per wave lasts 672 matrix cycles.  MEASURED NEUTRAL (S1 data gradient 0.856 vs 0.834 ms, step 54.97 vs 54.92 ms same-box,
                      // parity green): neither the barrier nor the weight-fragment wait is what the short K-tiles lose their time to
#ifndef KP_EPI_PERM
#define KP_EPI_PERM 1  // epilogue: lanes re-ordered (ds_bpermute) so that the four lanes of a channel are neighbours: a 16-lane
#endif                 // group of a store then writes 4 runs of 64 bytes instead of 16 pieces of 16 bytes (one per channel row)
#ifndef KP_XSPLIT
#define KP_XSPLIT 4   // odd MT: column tiles of the shared last row tile that the FIRST row-wave pair takes (the second takes the rest)
#endif
#ifndef KP_DIAG
#define KP_DIAG 0     // timing-only diagnostic builds (wrong results): bit 0 = no patch staging after the prologue, bit 1 = no weight
#endif                // DMA after the prologue, bit 2 = no products, bit 3 = no output stores

namespace cstp {

#if KP_DIAG & 16
// in-kernel stamps of consumer wave 0 of block 0 (diagnostic builds only): [0] cycles in K loops, [1] of them waiting at the
// barrier, [2] from the top of a K-tile until its weight fragments have landed,
// [3] epilogue cycles, [4] K-tiles, [5] items, [6] s_memrealtime ticks (100 MHz) over the K loops
__device__ unsigned long long kp_stamp[8];
#define KP_T() __builtin_amdgcn_s_memtime()
#endif

constexpr int KP_NPOS = 224;        // output positions per block
constexpr int KP_NTW = 7;           // 16-column MFMA tiles per consumer wave (two wave columns)
constexpr int KP_ROWS = 400;        // LDS rows (image positions incl. halo) per patch buffer: the host checks the geometry fits

struct PGeom {
  int Cs;           // channels of the gathered tensor
  int ncb;          // its 32-channel blocks
  int H, W, D, NF;  // frame size, frames per clip, frames in total (Nb * D)
  int M;            // valid output rows (channels of `out`)
  int rows_lds;     // LDS rows a tile can touch (<= KP_ROWS)
  int gpos;         // STATS: output positions per BatchNorm group (a multiple of KP_NPOS: a group is a whole number of tiles)
  int groups;       // STATS: BatchNorm groups (<= 2)
  int acc = 0;      // out += instead of out = (the caller's gradient accumulation; data gradient)
  int quad = 0;     // staging by 16-byte loads (four columns of one channel per lane): W % 4 == 0, Cs % 8 == 0, 16-byte-aligned src
};

// packed weights for igemm_k1p: wpk[mblk][kt = cb * 9 + tap][row (16*MT)][physical chunk (8)][8 f16], row m scaled by a power of
// two (inv_a[m] = its inverse); logical chunk L = 4 * plane + (k % 32) / 8 sits at L ^ (row & 7).  One block per row m.
// forward: value(m, c, tap) = w[m][c][tap];  data gradient: rows are INPUT channels, k runs over OUTPUT channels and the
// taps are mirrored: value(m = c_in, c = k_out, tap) = w[k_out][c_in][8 - tap].
__device__ __forceinline__ void
pack_patch_body(const float* __restrict__ w, uint4* __restrict__ wpk, float* __restrict__ inv_a,
                unsigned* __restrict__ cells, int ncells, int kout, int cin, int ncb, int rows_per_blk, int dgrad,
                int nt, const int m_in) {    // nt: filter taps (9: igemm_k1p; 3: the temporal layers' igemm_k1t); m: the block's row
  __shared__ unsigned red[4];
  const int m = __builtin_amdgcn_readfirstlane(m_in);
  const int t = threadIdx.x;
  if (m == 0 && t < ncells) cells[t] = 0;
  const int mreal = dgrad ? cin : kout, creal = dgrad ? kout : cin;
  auto fetch = [&](int c, int tap) __attribute__((always_inline)) -> float {
    if (m >= mreal || c >= creal) return 0.f;
    return dgrad ? w[((size_t)c * cin + m) * nt + (nt - 1 - tap)] : w[((size_t)m * cin + c) * nt + tap];
  };
  // one item = 8 consecutive channels of one tap = one 16-byte chunk per plane
  const int nitems = ncb * nt * 4;
  unsigned mx = 0;
  for (int it = t; it < nitems; it += 256) {
    const int c8 = it & 3, kt = it >> 2, cb = kt / nt, tap = kt - cb * nt;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned a = __builtin_bit_cast(unsigned, fetch(cb * 32 + c8 * 8 + e, tap)) & 0x7fffffffu;
      mx = mx > a ? mx : a;
    }
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
  const int mblk = m / rows_per_blk, rloc = m - mblk * rows_per_blk;
  for (int it = t; it < nitems; it += 256) {
    const int c8 = it & 3, kt = it >> 2, cb = kt / nt, tap = kt - cb * nt;
    uint4 ph, pl;
    unsigned hh, ll;
    const int c0 = cb * 32 + c8 * 8;
    split2h(fetch(c0, tap), fetch(c0 + 1, tap), sc, hh, ll); ph.x = hh; pl.x = ll;
    split2h(fetch(c0 + 2, tap), fetch(c0 + 3, tap), sc, hh, ll); ph.y = hh; pl.y = ll;
    split2h(fetch(c0 + 4, tap), fetch(c0 + 5, tap), sc, hh, ll); ph.z = hh; pl.z = ll;
    split2h(fetch(c0 + 6, tap), fetch(c0 + 7, tap), sc, hh, ll); ph.w = hh; pl.w = ll;
    uint4* row = wpk + (((size_t)mblk * (ncb * nt) + kt) * rows_per_blk + rloc) * 8;
    row[c8 ^ (rloc & 7)] = ph;
    row[(4 + c8) ^ (rloc & 7)] = pl;
  }
}
__global__ void __launch_bounds__(256)
pack_weights_patch_kernel(const float* __restrict__ w, uint4* __restrict__ wpk, float* __restrict__ inv_a,
                          unsigned* __restrict__ cells, int ncells, int kout, int cin, int ncb, int rows_per_blk, int dgrad,
                          int nt) {
  pack_patch_body(w, wpk, inv_a, cells, ncells, kout, cin, ncb, rows_per_blk, dgrad, nt, (int)blockIdx.x);
}

// STATS: the forward launch also leaves, per output channel and BatchNorm group, the sums of the outputs and of their squares
// for the train-mode BatchNorm that consumes them (part[((ch * groups + grp) * nsplit + j) * 2 + {0, 1}], fp64, j = this block's
// number among the nsplit blocks that own the same row block) -- bn_reduce's pass over the tensor is then not needed.  The sums
// are taken around a per-channel PIVOT c (the caller's guess of the mean: the BatchNorm's running mean; null = 0):
// sum(y - c) and sum((y - c)^2), c stored behind the partials.  The fp32 roundings of the per-lane partial sums (28 values
// and their squares before the sums go to fp64) are then relative to the spread of the channel around c instead of to its
// magnitude, and var = E[(y-c)^2] - (E[y-c])^2 no longer cancels for channels with |mean| >> std (round-2 ADVICE).  A lane of
// the transposed accumulator tile holds 28 values of ONE channel, so the per-item work is 2 FMAs per value, two cross-row adds
// and one fp64 LDS atomic per lane-channel; the block keeps its sums in LDS across its items and writes them once at the end.
// Next to the sums the launch leaves each channel's SMALLEST and LARGEST output per group (ordered-integer keys of the fp32
// values, key_of_float, behind the pivots: mm[((ch * groups + grp) * nsplit + j) * 2 + {min, max}]): with them the BatchNorm
// finalize knows the exact range of relu(scale * y + shift) -- the operand scale of the convolution that consumes the
// normalised tensor WITHOUT that tensor ever being written (cstp_bn_finalize_pre, in_affine of the temporal convolution).
__device__ __forceinline__ unsigned key_of_float(float f) {
  const unsigned b = __builtin_bit_cast(unsigned, f);
  return b ^ ((unsigned)((int)b >> 31) | 0x80000000u);           // monotone: a < b  <=>  key(a) < key(b)
}
__device__ __forceinline__ float float_of_key(unsigned k) {
  return __builtin_bit_cast(float, (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

template <int MT, bool STATS = false>
__global__ void __launch_bounds__(512, 2)
igemm_k1p(const PGeom g, const uint4* __restrict__ wpk, const float* __restrict__ src, float* __restrict__ out,
          const float* __restrict__ inv_a, const unsigned* __restrict__ bcell, int ntiles, int nmblk, double* __restrict__ part,
          const float* __restrict__ pivot, unsigned* __restrict__ zcell) {
  constexpr int BM = 16 * MT;
  constexpr int A_U4 = BM * 8;                       // uint4 per packed K-tile
  constexpr int A_DMA = BM / 8;                      // 1 KiB LDS-DMA pieces per K-tile
  constexpr int P_U4 = KP_ROWS * 8;
  // Weight ring: three K-tiles where LDS is full (128 / 144 rows); SIX at 64 rows -- a K-tile of 42 products per consumer wave
  // lasts ~0.3 us, and a weight piece requested two K-tiles ahead (0.6 us) is not back from L2 when its K-tile starts: the
  // DMA waves' wait, and behind it the barrier, set the K-tile time.  Five K-tiles of lead cost 24 KB.
  constexpr int GK = (KP_GROUP3 && MT == 4) ? 3 : 1;           // K-tiles per barrier
  constexpr int RING = (GK == 3 || (KP_RING6 && MT == 4)) ? 6 : 3;
  __shared__ uint4 smem[RING * A_U4 + 2 * P_U4 + 2 * (BM / 4) + (STATS ? BM + BM / 2 : 0)];
  uint4* const ring = smem;
  uint4* const patch = smem + RING * A_U4;
  float* const inva_s = reinterpret_cast<float*>(smem + RING * A_U4 + 2 * P_U4);      // [2][BM], by item parity
  double* const stat_s = reinterpret_cast<double*>(smem + RING * A_U4 + 2 * P_U4 + 2 * (BM / 4));   // STATS: [BM][2] sums of the block's group
  unsigned* const mm_s = reinterpret_cast<unsigned*>(smem + RING * A_U4 + 2 * P_U4 + 2 * (BM / 4) + BM);   // STATS: [BM][2] range keys

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

  // PERSISTENT blocks (one per CU: the LDS image fills it).  Work item = (position tile, row block).  XCD-aware order: XCD x
  // owns a contiguous chunk of position tiles; its blocks (slots) walk that chunk's items round-robin, so at any moment the
  // blocks of one XCD work on neighbouring tiles and on the row blocks of the same tile (shared patch and halo rows in L2).
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  // STATS with two BatchNorm groups: the slots of an XCD are dealt to the groups alternately and a block walks the tiles of ITS
  // group only (host: nslots is a multiple of groups * nmblk, a group is a whole number of tiles), so the block's sums and its
  // range live in one group's worth of LDS and no tile sequence ever crosses a group boundary.
  const int ngrp = STATS ? g.groups : 1;
  const int bgrp = slot % ngrp, gslot = slot / ngrp, gnslots = nslots / ngrp;
  const int gtiles = ntiles / ngrp;
  const int chunk = (gtiles + 7) >> 3;
  int tiles_x = gtiles - xcd * chunk;
  tiles_x = tiles_x < 0 ? 0 : (tiles_x < chunk ? tiles_x : chunk);
  const int cnt_x = tiles_x * nmblk;                 // items of this XCD (in my group)
  const int nitems = gslot < cnt_x ? (cnt_x - gslot + gnslots - 1) / gnslots : 0;
  // STATS: this block's slot in the partial-sum table (gnslots is a multiple of nmblk, so a block meets one row block only)
  const int st_mblk = gslot % nmblk, st_nsplit = gnslots * 8 / nmblk, st_j = xcd * (gnslots / nmblk) + gslot / nmblk;
  // the range table sits behind the partial sums and the pivots: mm[((ch * groups + grp) * nsplit + j) * 2 + {min, max}]
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
    // the pivot the sums are taken around rides behind the partials (one writer per row block), so that the fold works with
    // exactly the value this launch used
    if (st_j == 0 && bgrp == 0) {
      for (int row = threadIdx.x; row < BM; row += 256) {
        const int ch = st_mblk * BM + row;
        if (ch < g.M) part[(size_t)g.M * g.groups * st_nsplit * 2 + ch] = pivot != nullptr ? (double)pivot[ch] : 0.0;
      }
    }
  };
  if constexpr (STATS) {
    // the cell the BatchNorm finalize takes the consumer's operand maximum into (atomicMax): zeroed here, a launch earlier
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

  const int H = g.H, W = g.W, HW = H * W, PITCH = W + 2;
  const int P = g.NF * HW;                           // positions in total
  const int nkt = g.ncb * 9;
  const size_t chs = (size_t)g.D * HW;               // channel stride of src / row stride of out (elements)

  if (wave == 4 || wave == 5) {
    // ============================================ weight DMA waves (4, 5) ============================================
    // The K-tile sequence is FLAT across the block's items (nkt is a multiple of 3, so the ring slot sequence simply runs on):
    // while the consumers finish item i these waves already request item i + 1's first weight K-tiles.  Each wave moves half
    // of a K-tile's 1 KiB pieces; these waves issue no other vector-memory instruction, so the wait below can be COUNTED: it
    // leaves the batch just issued (K-tile k + 2) in flight across the barrier and retires K-tile k + 1.
    constexpr int HALF_DMA = A_DMA / 2;
    static_assert(A_DMA % 2 == 0, "two DMA waves share a K-tile's pieces evenly");
    const int dw_ = wave - 4;
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(wpk, (unsigned)((size_t)nmblk * nkt * A_U4 * 16));
    int d_it = 0, d_kt = 0, d_mblk, tl_unused, d_ring = 0;
    item_of(0, tl_unused, d_mblk);
    auto dma_next = [&]() __attribute__((always_inline)) {
      // past the last item the requests re-read its last K-tile into a ring slot nobody reads any more: the instruction
      // count per iteration stays constant, which is what the counted wait relies on
      const unsigned so = (unsigned)((((size_t)d_mblk * nkt + d_kt) * A_U4) * 16);
      uint4* dst = ring + d_ring * A_U4;
      d_ring = d_ring == RING - 1 ? 0 : d_ring + 1;
#pragma unroll
      for (int pc = 0; pc < HALF_DMA; ++pc) {
        const int piece = dw_ * HALF_DMA + pc;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(dst + piece * 64), 16,
                                                 (unsigned)(lane * 16 + piece * 1024), so, 0, 0);
      }
      if (d_it + 1 < nitems || d_kt + 1 < nkt) {
        if (++d_kt == nkt) { d_kt = 0; ++d_it; item_of(d_it, tl_unused, d_mblk); }
      }
    };
    if constexpr (GK == 3) {
      // groups of three K-tiles: group g + 1 is requested into the three slots group g - 1 was read from and has the whole of
      // group g's products (~1 us) to land; this wave has nothing else to do, so it simply waits for it
      dma_next(); dma_next(); dma_next();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int ngroups = nitems * nkt / 3;
#pragma unroll 1
      for (int gq = 0; gq < ngroups; ++gq) {
        if (!(KP_DIAG & 2)) { dma_next(); dma_next(); dma_next(); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) dma_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int total = nitems * nkt;
#pragma unroll 1
    for (int k = 0; k < total; ++k) {
      if (!(KP_DIAG & 2)) dma_next();                 // K-tile k + RING - 1 -> the slot K-tile k - 1 was read from
      // (counted: everything but the youngest RING - 2 batches has landed = K-tile k + 1 is in LDS)
      if constexpr (HALF_DMA == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      else if constexpr (HALF_DMA == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr (RING == 6) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  if (wave >= 6) {
    // ============================================ patch staging waves (6, 7) ============================================
    // Wave 6 gathers channels 0..15 of each 32-channel block, wave 7 channels 16..31; lane l owns LDS rows l + 64 r.  They share
    // their SIMDs with the second row-wave's consumers (waves 2, 3).  Seven rounds
    // of (16 dword gathers -> split -> four 16-byte LDS stores) stage the NEXT channel block's patch while the current one is
    // multiplied (schedule below).  No load is conditional -- a round with nothing to stage
    // gathers through out-of-range offsets (zeros, no memory traffic) -- so the compiler's s_waitcnt accounting keeps two
    // rounds in flight with counted waits (igemm_split.h explains what a conditional load costs).
    constexpr int NR = (KP_ROWS + 63) / 64;            // 7
    static_assert(NR <= 7, "rounds must fit the nine taps of a channel block");
#ifdef KP_SPRIO
    __builtin_amdgcn_s_setprio(KP_SPRIO);
#endif
    const int half = wave - 6;
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_src = make_rsrc(src, (unsigned)((size_t)(g.NF / g.D) * g.Cs * chs * 4));
    float sb, inv_unused;
    f16_scale(__builtin_amdgcn_readfirstlane(*bcell), sb, inv_unused);

    // byte offset of channel 0 at the (frame, image row, column) of my LDS rows in a tile's patch, or OOB (halo / padding)
    auto patch_offsets = [&](int tile, unsigned (&voff)[NR]) __attribute__((always_inline)) {
      const int pos0 = tile * KP_NPOS;
      const int pos_last = (pos0 + KP_NPOS - 1 < P ? pos0 + KP_NPOS - 1 : P - 1);
      const int v_lo = pos0 / W, v_last = pos_last / W;
      const int f_lo = v_lo / H;
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int l = lane + 64 * r;
        const int line = l / PITCH, col = l - line * PITCH - 1;
        // line -> (frame, h): frame k of the tile owns (its image rows inside the tile) + 2 lines, halo above and below
        int rem = line, f = f_lo, v0 = v_lo, h = -1;
        bool ok = false;
#pragma unroll 1
        for (int i = 0; i < 8; ++i) {
          int fend = (f + 1) * H - 1;
          fend = fend < v_last ? fend : v_last;
          const int cnt = fend - v0 + 3;
          if (rem < cnt) { h = v0 - 1 + rem - f * H; ok = true; break; }
          rem -= cnt;
          ++f;
          v0 = f * H;
          if (v0 > v_last) break;
        }
        ok = ok && h >= 0 && h < H && col >= 0 && col < W && f < g.NF && l < g.rows_lds;
        const int nb = f / g.D, d = f - nb * g.D;
        voff[r] = ok ? (unsigned)(((size_t)nb * g.Cs * chs + (size_t)d * HW + h * W + col) * 4) : OOB;
      }
    };
    const unsigned ch4 = (unsigned)(chs * 4);
    // channels past the tensor's last one (ragged last block) re-read the last channel: their packed weights are zero
    auto b_load = [&](unsigned vo, int cb, float (&v)[16]) __attribute__((always_inline)) {
#if KP_DIAG & 64
      { _Pragma("unroll") for (int j = 0; j < 16; ++j) v[j] = __builtin_bit_cast(float, vo + j); return; }   // no loads: split + store only
#endif
      const int c0 = cb * 32 + half * 16;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int c = c0 + j < g.Cs ? c0 + j : g.Cs - 1;
        buf_load_x1(v[j], vo, rs_src, (unsigned)c * ch4);
      }
    };
    auto b_store = [&](int buf, int r, const float (&v)[16]) __attribute__((always_inline)) {
      const int l = lane + 64 * r;
#if KP_DIAG & 32
      { _Pragma("unroll") for (int j = 0; j < 16; ++j) asm volatile("" :: "v"(v[j])); return; }      // loads only: no split, no store
#endif
      uint4 ph[2], pl[2];
      unsigned hh, ll;
#define CSTP_SPLITH(J, DST, F) split2h(v[J], v[(J) + 1], sb, hh, ll); ph[DST].F = hh; pl[DST].F = ll;
      CSTP_SPLITH(0, 0, x) CSTP_SPLITH(2, 0, y) CSTP_SPLITH(4, 0, z) CSTP_SPLITH(6, 0, w)
      CSTP_SPLITH(8, 1, x) CSTP_SPLITH(10, 1, y) CSTP_SPLITH(12, 1, z) CSTP_SPLITH(14, 1, w)
#undef CSTP_SPLITH
      if (l < KP_ROWS) {
        uint4* row = patch + buf * P_U4 + l * 8;
        const int x7 = l & 7;
        row[(2 * half) ^ x7] = ph[0];
        row[(2 * half + 1) ^ x7] = ph[1];
        row[(4 + 2 * half) ^ x7] = pl[0];
        row[(5 + 2 * half) ^ x7] = pl[1];
      }
    };

    if (g.quad) {
      // ---- QUAD staging (round 4).  The dword rounds above keep 16 x 4-byte loads per lane and round in flight and a wave can have
      // 63 vector-memory operations outstanding in all (vmcnt is six bits): the staging stream was latency-bound by its own
      // instruction count, and its 112 loads + address work per channel block came straight out of the issue slots of the SIMDs it
      // shares with two consumer waves (diagnostic builds, S1 forward: 0.74 ms with the staging, 0.56 without; the staging alone
      // 0.18: fully additive).  Here a task is (image line, four consecutive columns, 8-channel group): eight 16-byte loads (one
      // per channel: 64 consecutive lanes cover up to 1 KiB of one channel's contiguous lines) give a lane the two 16-byte
      // chunks (hi, lo) of FOUR LDS rows.  lines * (W / 4) * 2 tasks per wave (wave 6: channel groups 0, 1; wave 7: 2, 3) = 168 at
      // 56 x 56 and 28 x 28 = three rounds: 24 loads per lane and channel block instead of 112.  The halo COLUMNS (-1 and W) of
      // every line are the zero padding itself: their LDS rows are zeroed once.
      constexpr int QR = 3;
      const int QW = W >> 2, nlines = g.rows_lds / PITCH;
      const int ntask = nlines * QW * 2;
      for (int e = lane + 64 * half; e < 2 * nlines * 2 * 8; e += 128) {     // (buffer, line, left / right halo row, chunk)
        const int c8 = e & 7, side = (e >> 3) & 1, rest = e >> 4, line = rest % nlines, buf = rest / nlines;
        patch[buf * P_U4 + (line * PITCH + (side ? W + 1 : 0)) * 8 + c8] = make_uint4(0u, 0u, 0u, 0u);
      }
      int q_row0[QR], q_line[QR], q_col[QR], q_gch[QR];
      bool q_ok[QR];
#pragma unroll
      for (int r = 0; r < QR; ++r) {
        const int tk = lane + 64 * r;
        q_ok[r] = tk < ntask;
        const int tq = q_ok[r] ? tk : 0;
        const int qi = tq >> 1;
        q_gch[r] = 2 * half + (tq & 1);                  // my 8-channel group of the 32-channel block
        q_line[r] = qi / QW;
        q_col[r] = 4 * (qi - q_line[r] * QW);
        // (lanes past the last task stage zeros into four spare rows behind the tile's image -- host: rows_lds + 32 <= KP_ROWS --
        //  so that no store is conditional)
        q_row0[r] = q_ok[r] ? q_line[r] * PITCH + 1 + q_col[r] : g.rows_lds + 4 * (lane & 7);
      }
      const unsigned ch4q = (unsigned)(chs * 4);
      auto quad_offsets = [&](int tile, unsigned (&voff)[QR]) __attribute__((always_inline)) {
        const int pos0 = tile * KP_NPOS;
        const int pos_last = (pos0 + KP_NPOS - 1 < P ? pos0 + KP_NPOS - 1 : P - 1);
        const int v_lo = pos0 / W, v_last = pos_last / W;
        const int f_lo = v_lo / H;
#pragma unroll
        for (int r = 0; r < QR; ++r) {
          // line -> (frame, h): frame k of the tile owns (its image rows inside the tile) + 2 lines, halo above and below
          int rem = q_line[r], f = f_lo, v0 = v_lo, h = -1;
          bool ok = false;
#pragma unroll 1
          for (int i = 0; i < 8; ++i) {
            int fend = (f + 1) * H - 1;
            fend = fend < v_last ? fend : v_last;
            const int cnt = fend - v0 + 3;
            if (rem < cnt) { h = v0 - 1 + rem - f * H; ok = true; break; }
            rem -= cnt;
            ++f;
            v0 = f * H;
            if (v0 > v_last) break;
          }
          ok = ok && q_ok[r] && h >= 0 && h < H && f < g.NF;
          const int nb = f / g.D, d = f - nb * g.D;
          voff[r] = ok ? (unsigned)(((size_t)nb * g.Cs * chs + (size_t)d * HW + h * W + q_col[r]) * 4) + (unsigned)(q_gch[r] * 8) * ch4q
                       : OOB;
        }
      };
      struct QRound { u32x4 v[8]; };
      auto q_load = [&](int r, unsigned vo, int cb, QRound& rd) __attribute__((always_inline)) {
        // (channel groups past the tensor's last channel: zeros against zero weights)
        const unsigned v = (cb * 32 + q_gch[r] * 8 < g.Cs) ? vo : OOB;
        const unsigned so = (unsigned)(cb * 32) * ch4q;
#pragma unroll
        for (int e = 0; e < 8; ++e) buf_load_x4(rd.v[e], v, rs_src, so + (unsigned)e * ch4q);
      };
      auto q_store = [&](int buf, int r, const QRound& rd) __attribute__((always_inline)) {
        // (the components through a float vector: indexing rd.v[e][i] directly has made hipcc read component 0 for every i)
        f32x4 vf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) vf[e] = __builtin_bit_cast(f32x4, rd.v[e]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {                   // my four LDS rows
          uint4 ph, pl;
          unsigned hh, ll;
          split2h(vf[0][i], vf[1][i], sb, hh, ll); ph.x = hh; pl.x = ll;
          split2h(vf[2][i], vf[3][i], sb, hh, ll); ph.y = hh; pl.y = ll;
          split2h(vf[4][i], vf[5][i], sb, hh, ll); ph.z = hh; pl.z = ll;
          split2h(vf[6][i], vf[7][i], sb, hh, ll); ph.w = hh; pl.w = ll;
          const int row = q_row0[r] + i;
          uint4* prow = patch + buf * P_U4 + row * 8;
          const int x7 = row & 7;
          prow[q_gch[r] ^ x7] = ph;
          prow[(4 + q_gch[r]) ^ x7] = pl;
        }
      };
#ifndef KP_DEEP
#define KP_DEEP 1
#endif
      if constexpr (KP_DEEP && MT != 4 && GK == 1 && !(KP_DIAG & 1)) {
        // ---- a whole channel block of lead (128 / 144 rows): channel block x + 2 (flat over the block's items) is REQUESTED
        // while block x is multiplied and STORED into the free patch buffer one block later, nine K-tiles after its loads
        // instead of four, at the first three taps of a block instead of taps 4..6.  Two register sets of three rounds (192
        // VGPRs: the staging waves have the whole 256 of a two-waves-per-SIMD block); the set stored at block x is the one
        // block x + 2's successor is loaded into.  No load is conditional (past the end: OOB).  Same box: S1 forward 0.742 ->
        // 0.718 ms.  NOT at 64 rows (the data gradients' 144-channel patches, five channel blocks per item): 0.865 -> 0.898 ms
        // there, so that instantiation keeps the four-tap schedule below.
        unsigned qv[QR];
        int l_it = 0, l_cb = 0, tile, mb_unused;
        item_of(0, tile, mb_unused);
        quad_offsets(tile, qv);
        auto advance = [&]() __attribute__((always_inline)) {
          if (++l_cb == g.ncb) {
            l_cb = 0;
            ++l_it;
            if (l_it < nitems) { item_of(l_it, tile, mb_unused); quad_offsets(tile, qv); }
            else {
#pragma unroll
              for (int r = 0; r < QR; ++r) qv[r] = OOB;
            }
          }
        };
        QRound s0[QR], s1[QR];                           // by parity of the flat channel-block index they carry
#pragma unroll
        for (int r = 0; r < QR; ++r) q_load(r, qv[r], l_cb, s0[r]);
#pragma unroll
        for (int r = 0; r < QR; ++r) q_store(0, r, s0[r]);
        advance();
#pragma unroll
        for (int r = 0; r < QR; ++r) q_load(r, qv[r], l_cb, s1[r]);
        advance();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int pb = 0;
        const int total = nitems * g.ncb;
        auto block = [&](auto ptag) __attribute__((always_inline)) {
          constexpr int PX = decltype(ptag)::value;      // parity of the block being multiplied
          QRound (&sst)[QR] = PX ? s0 : s1;              // block x + 1: into the free buffer now
          QRound (&sld)[QR] = PX ? s1 : s0;              // block x + 2: requested now
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
            if (tap < QR) {
              q_store(pb ^ 1, tap, sst[tap]);
              q_load(tap, qv[tap], l_cb, sld[tap]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
          }
          advance();
          pb ^= 1;
        };
#pragma unroll 1
        for (int x = 0; x < total; x += 2) {
          block(std::integral_constant<int, 0>{});
          if (x + 1 >= total) break;
          block(std::integral_constant<int, 1>{});
        }
        return;
      }
      unsigned qv_cur[QR], qv_nxt[QR];
      int tile, mb_unused;
      item_of(0, tile, mb_unused);
      quad_offsets(tile, qv_cur);
#pragma unroll
      for (int r = 0; r < QR; ++r) qv_nxt[r] = OOB;
      QRound rq[QR];
#pragma unroll
      for (int r = 0; r < QR; ++r) q_load(r, qv_cur[r], 0, rq[r]);
#pragma unroll
      for (int r = 0; r < QR; ++r) q_store(0, r, rq[r]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      int pb = 0;
      for (int it = 0; it < nitems; ++it) {
        const bool next_item = it + 1 < nitems;
        for (int cb = 0; cb < g.ncb; ++cb) {
          const bool last_cb = cb + 1 == g.ncb;
          const bool stage = (!last_cb || next_item) && !(KP_DIAG & 1);
          if (last_cb && next_item) {
            item_of(it + 1, tile, mb_unused);
            quad_offsets(tile, qv_nxt);
          }
          const int ncb_ = last_cb ? 0 : cb + 1;
          // round t is LOADED at tap t (t = 0, 1, 2) and STORED four taps later
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
            if (tap >= 4 && tap - 4 < QR) q_store(pb ^ 1, tap - 4, rq[tap - 4]);
            if (tap < QR) q_load(tap, stage ? (last_cb ? qv_nxt[tap] : qv_cur[tap]) : OOB, ncb_, rq[tap]);
            if (GK == 1 || tap % 3 == 2) {
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              __builtin_amdgcn_s_barrier();
            }
          }
          pb ^= 1;
        }
#pragma unroll
        for (int r = 0; r < QR; ++r) qv_cur[r] = qv_nxt[r];
      }
      return;
    }

    unsigned voff_cur[NR], voff_nxt[NR];
    int tile, mb_unused;
    item_of(0, tile, mb_unused);
    patch_offsets(tile, voff_cur);
    // ---- prologue: the first item's channel block 0, all rounds in flight at once
    {
      float pro[NR][16];
#pragma unroll
      for (int r = 0; r < NR; ++r) b_load(voff_cur[r], 0, pro[r]);
#pragma unroll
      for (int r = 0; r < NR; ++r) b_store(0, r, pro[r]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    float rb[NR][16];                                 // all rounds of a channel block in flight
    int pb = 0;                                       // patch buffer of the channel block being consumed
    for (int it = 0; it < nitems; ++it) {
      const bool next_item = it + 1 < nitems;
      for (int cb = 0; cb < g.ncb; ++cb) {
        const bool last_cb = cb + 1 == g.ncb;
        const bool stage = (!last_cb || next_item) && !(KP_DIAG & 1);   // a next channel block (of this or the next item)
        if (last_cb && next_item) {
          item_of(it + 1, tile, mb_unused);
          patch_offsets(tile, voff_nxt);
        }
        const int ncb_ = last_cb ? 0 : cb + 1;
        // rounds 2 t, 2 t + 1 are LOADED at tap t (t = 0..3) and STORED four taps later: a gather has ~4 K-tiles (4-5 us) to
        // come back from HBM.  (Stored two taps after the load, the staging waves regularly reached the barrier late -- in-kernel
        // stamps: 277 instead of 58 cycles of barrier wait per K-tile, 0.81 instead of 0.66 ms for the S1 layer.)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int rs = 2 * (tap - 4) + e;            // the round stored at this tap
            if (tap >= 4 && rs < NR) b_store(pb ^ 1, rs, rb[rs]);
          }
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int rl = 2 * tap + e;                   // the round loaded at this tap
            if (rl < NR) {
              const unsigned vo = stage ? (last_cb ? voff_nxt[rl] : voff_cur[rl]) : OOB;
              b_load(vo, ncb_, rb[rl]);
            }
          }
          if (GK == 1 || tap % 3 == 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
          }
        }
        pb ^= 1;
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) voff_cur[r] = voff_nxt[r];
    }
    return;
  }

  // =================================================== consumers ===================================================
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;
  // Row tiles: each row-wave owns MT / 2 whole row tiles (all 7 of its column tiles).  When MT is odd the LAST row tile is
  // shared by all four consumers, each taking it on a part of its own column tiles -- the first row-wave on column tiles
  // 0..3, the second on 4..6: 32 / 31 products per tap and wave instead of 35 / 28 when one row-wave owned five row tiles.
  constexpr int NIF = MT / 2;
  const int mt0 = wm * NIF;
  // my A fragment chunks (rows fr + 16 i: (row & 7) == (fr & 7))
  const int qa0 = fk ^ (fr & 7), qa1 = (4 + fk) ^ (fr & 7);
  float invb, sc_unused;
  f16_scale(__builtin_amdgcn_readfirstlane(*bcell), sc_unused, invb);
  const int q = lane >> 4;

  if constexpr (STATS) {
    for (int e = t; e < BM * 2; e += 256) { stat_s[e] = 0.0; mm_s[e] = (e & 1) ? 0u : 0xffffffffu; }   // (consumer threads are t < 256; read many barriers later)
  }
  __builtin_amdgcn_s_barrier();                        // the first item's prologue data is staged
#ifndef KP_PRIO
#define KP_PRIO 2
#endif
  if (KP_PRIO > 0) __builtin_amdgcn_s_setprio(KP_PRIO);    // the matrix stream outranks the staging waves it shares SIMDs with

  // two instantiations of the body (the shared row tile's column range is a compile-time constant) instead of branches
  auto body = [&](auto xj0_tag, auto xjn_tag) __attribute__((always_inline)) {
  constexpr int NI = NIF;
  constexpr int XJ0 = decltype(xj0_tag)::value, XJN = decltype(xjn_tag)::value;      // the shared row tile: my column tiles of it
  constexpr int XA = XJN > 0 ? 1 : 0;
  int pb = 0, slot3 = 0;
  for (int it = 0; it < nitems; ++it) {
    int tile, mblk;
#if KP_DIAG & 16
    const unsigned long long t_item0 = KP_T();
#endif
    item_of(it, tile, mblk);
    const int pos0 = tile * KP_NPOS;
    // (pos0 is uniform: these three divisions are the item's only ones -- every per-lane position below is reached from
    // them by stepping; one division pair per column tile and lane, here and in the epilogue, was 28 x ~35 instructions per item)
    const int v_lo = pos0 / W, w_lo = pos0 - v_lo * W;
    const int f_lo = v_lo / H, h_lo = v_lo - f_lo * H;
    float* const inva = inva_s + (it & 1) * BM;
    if (t < BM) inva[t] = inv_a[mblk * BM + t];       // read back in this item's epilogue, >= 9 barriers later

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
#if KP_DIAG & 128
    // timing-only diagnostic (wrong results): the whole row tiles' products as v_mfma_f32_32x32x16_f16 -- the same multiply-add
    // count in half the instructions, each holding the SIMD's issue port 8 of 32 cycles instead of 8 of 16
    f32x16 accq[KP_NTW];
#pragma unroll
    for (int j = 0; j < KP_NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) accq[j][r] = 0.f;
#endif

    // LDS row of my column in each of my 7 column tiles, one line above / one column left of it; a tap adds dh * PITCH + dw
    int base[KP_NTW];
    {
      int w = w_lo + wn * (KP_NTW * 16) + fr, h = h_lo, dl = 0;      // column, line of the frame, patch lines below the tile's first
#pragma unroll
      for (int j = 0; j < KP_NTW; ++j) {
        while (w >= W) { w -= W; ++dl; if (++h == H) { h = 0; dl += 2; } }
        // (ragged last tile: past the end read row 0 -- something valid, never stored)
        base[j] = pos0 + (wn * KP_NTW + j) * 16 + fr < P ? dl * PITCH + w : 0;
        w += 16;
      }
    }

    // B fragments (my column tile j at one tap) sit in LDS long before they are needed, also the next K-tile's; they are
    // fetched TWO column tiles ahead into three register buffers.  (One tile ahead hides the LDS latency behind a group of
    // 15 products -- MT = 9 -- but not behind the 6 of MT = 4: in-kernel stamps showed 1970 cycles per K-tile for 672 cycles
    // of products there.)  Only the weight fragments wait for the barrier.  Loop order: column tile outer, row tile inner
    // (independent accumulators back to back).
    // The B fragment reads are INLINE ASM with hand-placed COUNTED waits: left to the compiler (VGPR budget exhausted by the
    // accumulators) it hoisted half of each pair to the top of the K-tile and issued the other half right in front of its first
    // use -- seven exposed LDS latencies per K-tile, 22 instead of 16 cycles per MFMA (in-kernel stamps).  LDS operations
    // complete in order, so `lgkmcnt(4)` behind the issue of tile j + 2 means tile j has landed whatever the compiler's own
    // (weight fragment) reads in between.  (Scalar memory loads share the counter and return out of order, but they can only
    // make this wait longer: of the operations that must have completed to reach the count, at most the scalar ones are not
    // LDS reads, and the LDS reads among them are the OLDEST ones.  tools/check_k1p_isa.sh lists any that sit among the
    // products -- there are none in the K loops -- and fails on scratch use.)
    // The tile sequence runs on across K-tiles (7 per K-tile, buffer = sequence number mod 3), so the K-tile loop is
    // unrolled by three -- nkt is a multiple of 9 -- with the buffer phase a compile-time constant.
    f16x8 bh[3], bl[3];
    const unsigned patch_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) uint4*)patch);
    // The address of a read is computed one group before its issue (among the previous group's products), so that between two
    // groups of products there is only the wait and the two reads: with the six dependent address instructions there the
    // matrix pipe idled ~45 cycles per group (21 instead of 16 cycles per MFMA).
    auto b_addr = [&](int j, unsigned pbuf_bytes, int ts) __attribute__((always_inline)) -> unsigned {
      const int row = base[j] + ts;
      const int qq = fk ^ (row & 7);
      return pbuf_bytes + (unsigned)(row * 8 + qq) * 16u;                      // the lo plane sits 4 chunks (64 bytes) away: ^ 64
    };
    auto issue_b = [&](f16x8& dh, f16x8& dl, unsigned addr) __attribute__((always_inline)) {
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "=&v"(dh), "=&v"(dl) : "v"(addr), "v"(addr ^ 64u));
      __builtin_amdgcn_sched_barrier(0);             // the products that follow stay BEHIND the issue (rule: asm orders nothing)
    };
    // K-tile 0 of this item is already staged: its column tiles 0 and 1
    issue_b(bh[0], bl[0], b_addr(0, patch_lds + pb * (P_U4 * 16), 0));
    issue_b(bh[1], bl[1], b_addr(1, patch_lds + pb * (P_U4 * 16), 0));
    const int arow0 = (mt0 * 16 + fr) * 8;
    unsigned addr_n = b_addr(2, patch_lds + pb * (P_U4 * 16), 0);               // column tile 2 of K-tile 0

    // one K-tile per call, NOT unrolled over the taps: with the tap a compile-time constant the compiler hoists all
    // 9 x 7 x 2 fragment addresses out of the loop (126 VGPRs) and spills the accumulators
    int tap = 0, dh_pitch = 0, dw = 0;                 // (slot3, the ring slot of the next K-tile, runs on across the items)
#if KP_DIAG & 16
    const bool stamp = blockIdx.x == 0 && wave == 0;
    unsigned long long s_loop = 0, s_bar = 0, s_a = 0;
    const unsigned long long t_loop0 = KP_T(), r_loop0 = __builtin_amdgcn_s_memrealtime();
#endif
    auto ktile = [&](auto ph_tag) __attribute__((always_inline)) {
      constexpr int PH = decltype(ph_tag)::value;        // buffer of this K-tile's column tile 0
#if KP_DIAG & 16
      const unsigned long long t_top = KP_T();
#endif
      const uint4* Ab = ring + slot3 * A_U4;
      const unsigned Bp = patch_lds + pb * (P_U4 * 16);
      const int ts = dh_pitch + dw;
      // the NEXT K-tile's tap shift / patch buffer
      int ntap = tap + 1, ndw = dw + 1, ndh = dh_pitch, npb = pb;
      if (ndw == 3) { ndw = 0; ndh += PITCH; }
      if (ntap == 9) { ntap = 0; ndh = 0; npb ^= 1; }
      const unsigned Bn = patch_lds + npb * (P_U4 * 16);
      const int nts = ndh + ndw;

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
#if KP_DIAG & 16
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      s_a += KP_T() - t_top;
#endif
#pragma unroll
      for (int j = 0; j < KP_NTW; ++j) {
        // column tile j + 2 (past the last: tiles 0, 1 of the next K-tile; past the item's last K-tile harmless reads,
        // issued again at the top of the next item) is requested now; tile j + 3's address is computed among tile j's products
        // GK == 3: the staging waves write the NEXT channel block's patch until the barrier behind this block's last K-tile
        // (tap 8), so its first two column tiles are NOT requested ahead of that barrier (the K loop requests them behind it)
        const bool hold = GK == 3 && PH == 2 && j + 2 >= KP_NTW && tap == 8;
        if (!hold) issue_b(bh[(PH + j + 2) % 3], bl[(PH + j + 2) % 3], addr_n);
        if (j + 3 < KP_NTW) addr_n = b_addr(j + 3, Bp, ts);
        else addr_n = b_addr(j + 3 - KP_NTW, Bn, nts);
        // tile j (requested two groups ago) has landed once at most the four youngest reads are outstanding
        if (!hold) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else if (j == KP_NTW - 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const f16x8 bhj = bh[(PH + j) % 3], blj = bl[(PH + j) % 3];
#if KP_DIAG & 4
        asm volatile("" :: "v"(bhj), "v"(blj));
        if (j == 0) { _Pragma("unroll") for (int i = 0; i < NI + XA; ++i) asm volatile("" :: "v"(ah[i]), "v"(al[i])); }
        continue;
#endif
        // (j is a constant after unrolling: this test folds)
        const bool xj = XJN > 0 && j >= XJ0 && j < XJ0 + XJN;
        const int jx = xj ? j - XJ0 : 0;
#if KP_DIAG & 128
        if constexpr (NI == 4) {
#pragma unroll
          for (int i = 0; i < NI; i += 2) {
            accq[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bhj, al[i], accq[j], 0, 0, 0);
            accq[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(blj, ah[i + 1], accq[j], 0, 0, 0);
            accq[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bhj, ah[i], accq[j], 0, 0, 0);
          }
          if (xj) {
            accx[jx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bhj, al[NI + XA - 1], accx[jx], 0, 0, 0);
            accx[jx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(blj, ah[NI + XA - 1], accx[jx], 0, 0, 0);
            accx[jx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bhj, ah[NI + XA - 1], accx[jx], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          continue;
        }
#endif
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bhj, al[i], acc[i][j], 0, 0, 0);
        if (xj) accx[jx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bhj, al[NI + XA - 1], accx[jx], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(blj, ah[i], acc[i][j], 0, 0, 0);
        if (xj) accx[jx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(blj, ah[NI + XA - 1], accx[jx], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bhj, ah[i], acc[i][j], 0, 0, 0);
        if (xj) accx[jx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bhj, ah[NI + XA - 1], accx[jx], 0, 0, 0);
        // nothing moves across this point: the next group's issue stays behind these products
        __builtin_amdgcn_sched_barrier(0);
      }
      slot3 = slot3 == RING - 1 ? 0 : slot3 + 1;
      tap = ntap; dw = ndw; dh_pitch = ndh; pb = npb;
#if KP_DIAG & 16
      const unsigned long long t_b0 = KP_T();
#endif
      // (no drain in front of the barrier: every read of this K-tile's ring slot and of this channel block's patch buffer has
      // been waited for above; the two tiles in flight belong to the next K-tile and stay in flight across the barrier)
      if (GK == 1 || PH == 2) __builtin_amdgcn_s_barrier();
#if KP_DIAG & 16
      s_bar += KP_T() - t_b0;
#endif
    };
#pragma unroll 1
    for (int kt = 0; kt < nkt; kt += 3) {
      ktile(std::integral_constant<int, 0>{});
      ktile(std::integral_constant<int, 1>{});
      ktile(std::integral_constant<int, 2>{});
      if constexpr (GK == 3) {
        if (tap == 0 && kt + 3 < nkt) {               // a new channel block of this item: its patch is complete behind the barrier
          const unsigned Bq = patch_lds + pb * (P_U4 * 16);
          issue_b(bh[0], bl[0], b_addr(0, Bq, 0));
          issue_b(bh[1], bl[1], b_addr(1, Bq, 0));
          addr_n = b_addr(2, Bq, 0);
        }
      }
    }
    // the next item's prologue re-issues into buffers 0 and 1: the two reads still in flight must have landed before that
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#if KP_DIAG & 16
    const unsigned long long t_loop1 = KP_T(), r_loop1 = __builtin_amdgcn_s_memrealtime();
    s_loop = t_loop1 - t_loop0;
#endif
    // (pb has moved on to the buffer that holds the next item's first channel block)
#if KP_DIAG & 128
    if constexpr (NI == 4) {
#pragma unroll
      for (int j = 0; j < KP_NTW; ++j)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = accq[j][4 * i + r];
    }
#endif

    // ---- epilogue.  The products are issued with the POSITIONS as the MFMA's row operand and the weight rows as its column
    // operand, so the accumulator tile is the transpose of the usual one: col = lane & 15 is an OUTPUT ROW (channel) and
    // (lane >> 4) * 4 + reg are FOUR CONSECUTIVE POSITIONS of it -- a lane's four registers are 16 contiguous bytes of the
    // NCDHW output and go out as one store, no shuffles.  (History, measured with in-kernel stamps: one 4-byte store per value
    // from the untransposed tile cost 18 k cycles per item, store-issue bound; 16-byte stores behind in-quad DPP transposes
    // 10.7 k.)  The stores are PLAIN, not streaming: the 64-byte runs of one instruction are half lines, the other half comes
    // from the next column tile's store, and L2 merges the two before writing back -- with non-temporal stores the halves went
    // to memory separately (WRITE_SIZE 1.26 x the output bytes; plain stores in this order: 1.002 x, kernel -3.7 %).
    // Row-tile-outer order so that the two halves of a line are adjacent instructions.  (Frames whose size is not a multiple
    // of 4 positions -- 7 x 7 -- may straddle a frame inside the four: those layers store value by value.)
    // Output offsets without divisions: the tile's first position is (frame f_lo, line h_lo, column w_lo); a lane's positions
    // are reached from there by stepping (frame-internal offset sp, frame of the clip d, clip nb).
    const int nb_lo = f_lo / g.D, d_lo = f_lo - nb_lo * g.D;      // uniform
    struct Cur { int sp, d, nb; };
    auto norm = [&](Cur& c) __attribute__((always_inline)) {
      while (c.sp >= HW) { c.sp -= HW; if (++c.d == g.D) { c.d = 0; ++c.nb; } }
    };
    auto offs = [&](const Cur& c) __attribute__((always_inline)) -> size_t {
      return ((size_t)c.nb * g.M * g.D + c.d) * HW + c.sp;
    };
    const int sp_lo = h_lo * W + w_lo;
    const bool vec_ok = (HW & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;     // uniform
    if (vec_ok) {
      // The accumulator tile has lane = 16 * (position quad q) + (channel fr): stored as it is, the 16 lanes of one pass of a
      // 16-byte store hit 16 different channel rows.  KP_EPI_PERM: the values travel to lane 4 * fr + q first (ds_bpermute,
      // no memory), so that a pass writes four runs of 64 contiguous bytes; addresses follow the NEW lane's (channel, quad).
      const int sq = KP_EPI_PERM ? (lane & 3) : q;                    // position quad / channel whose values I STORE
      const int sfr = KP_EPI_PERM ? (lane >> 2) : fr;
      const int perm_src = (16 * (lane & 3) + (lane >> 2)) * 4;        // byte index of the lane whose values I receive
      size_t obase[KP_NTW];
      bool nok[KP_NTW];
      {
        Cur c = {sp_lo + wn * (KP_NTW * 16) + 4 * sq, d_lo, nb_lo};
#pragma unroll
        for (int j = 0; j < KP_NTW; ++j) {
          norm(c);
          nok[j] = pos0 + (wn * KP_NTW + j) * 16 + 4 * sq < P;
          obase[j] = offs(c);
          c.sp += 16;
        }
      }
#pragma unroll
      for (int i = 0; i < NI + XA; ++i) {
        const int mrow = (i < NI ? (mt0 + i) : (MT - 1)) * 16 + fr;
        const float sc = inva[mrow] * invb;
        const int m = mblk * BM + mrow;
        const int ms = mblk * BM + (i < NI ? (mt0 + i) : (MT - 1)) * 16 + sfr;      // the channel I store
        float* orow = out + (size_t)ms * chs;
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        float vmin = __builtin_inff(), vmax = -__builtin_inff();      // STATS: smallest / largest accumulator of my channel
        float pv = 0.f;                                  // STATS: the pivot in accumulator units (y = v * sc)
        if constexpr (STATS) { if (pivot != nullptr && m < g.M) pv = pivot[m] / sc; }
#pragma unroll
        for (int j = (i < NI ? 0 : XJ0); j < (i < NI ? KP_NTW : XJ0 + XJN); ++j) {
          const f32x4 v = i < NI ? acc[i < NI ? i : 0][j] : accx[i < NI ? 0 : j - XJ0];
          if ((KP_DIAG & 8) && v[0] != 12345.f) continue;
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
          if (nok[j] && ms < g.M) {
            f32x4* dst = reinterpret_cast<f32x4*>(orow + obase[j]);
            *dst = g.acc ? *dst + vs : vs;
          }
          if constexpr (STATS) {                           // (every position of every tile is valid: host condition)
            const f32x4 dv = v - pv; s1 += dv; s2 += dv * dv;
            vmin = __builtin_fminf(__builtin_fminf(vmin, v[0]), v[1]); vmin = __builtin_fminf(__builtin_fminf(vmin, v[2]), v[3]);
            vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, v[0]), v[1]); vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, v[2]), v[3]);
          }
        }
        if constexpr (STATS) {
          // my channel's 28 (or 16 / 12) values -> the four lanes that share it (lane, lane ^ 16, ^ 32, ^ 48) -> LDS
          float a = (s1[0] + s1[1]) + (s1[2] + s1[3]), b = (s2[0] + s2[1]) + (s2[2] + s2[3]);
          a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
          a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
          if (q == 0 && m < g.M) {
            double* dst = stat_s + mrow * 2;
            __hip_atomic_fetch_add(dst, (double)(a * sc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(dst + 1, (double)(b * sc) * (double)sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          // the range: sc > 0 and rounding is monotone, so min(v) * sc IS the smallest stored output
          vmin = __builtin_fminf(vmin, __shfl_xor(vmin, 16, 64)); vmax = __builtin_fmaxf(vmax, __shfl_xor(vmax, 16, 64));
          vmin = __builtin_fminf(vmin, __shfl_xor(vmin, 32, 64)); vmax = __builtin_fmaxf(vmax, __shfl_xor(vmax, 32, 64));
          if (q == 0 && m < g.M) {       // (LDS atomics next to the sums; round 3 sent two device-scope atomics per lane-channel and ITEM)
            __hip_atomic_fetch_min(mm_s + mrow * 2, key_of_float(vmin * sc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_max(mm_s + mrow * 2 + 1, key_of_float(vmax * sc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    } else {
      Cur c = {sp_lo + wn * (KP_NTW * 16) + 4 * q, d_lo, nb_lo};
#pragma unroll
      for (int j = 0; j < KP_NTW; ++j) {
        size_t ob[4];
        bool nk[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          norm(c);
          nk[r] = pos0 + (wn * KP_NTW + j) * 16 + 4 * q + r < P;
          ob[r] = offs(c);
          c.sp += 1;
        }
        c.sp += 12;
#pragma unroll
        for (int i = 0; i < NI + XA; ++i) {
          if (i == NI && !(j >= XJ0 && j < XJ0 + XJN)) continue;
          const int mrow = (i < NI ? (mt0 + i) : (MT - 1)) * 16 + fr;
          const float sc = inva[mrow] * invb;
          const int m = mblk * BM + mrow;
          const f32x4 v = i < NI ? acc[i < NI ? i : 0][j] : accx[(i == NI && j >= XJ0 && j < XJ0 + XJN) ? j - XJ0 : 0];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if ((KP_DIAG & 8) && v[r] != 12345.f) continue;
            if (nk[r] && m < g.M) {
              float* dst = out + ob[r] + (size_t)m * chs;
              CSTP_STORE(dst, g.acc ? *dst + v[r] * sc : v[r] * sc);
            }
          }
        }
      }
    }
#if KP_DIAG & 16
    if (stamp && lane == 0) {
      kp_stamp[0] += s_loop; kp_stamp[1] += s_bar; kp_stamp[2] += s_a; kp_stamp[3] += KP_T() - t_loop1;
      kp_stamp[4] += (unsigned long long)nkt; kp_stamp[5] += 1; kp_stamp[6] += r_loop1 - r_loop0;
      kp_stamp[7] += KP_T() - t_item0;              // the whole item: set-up + K loops + epilogue
    }
#endif
  }
  };
  using std::integral_constant;
  if constexpr (MT % 2 == 0) {
    body(integral_constant<int, 0>{}, integral_constant<int, 0>{});
  } else {
    if (wm == 0) body(integral_constant<int, 0>{}, integral_constant<int, KP_XSPLIT>{});
    else body(integral_constant<int, KP_XSPLIT>{}, integral_constant<int, KP_NTW - KP_XSPLIT>{});
  }
  if constexpr (STATS) {
    // the four consumer waves are the block's only live waves here (the others returned behind their last barrier)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    write_part(false);
  }
}

// LDS rows the 224-position tiles of NF frames of H x W touch at most: image rows spanned + 2 halo lines per frame spanned.
// The tile phase repeats once a tile starts on a frame boundary again, so the scan is short (7 or 14 tiles for this network).
static inline int patch_rows_needed(int NF, int H, int W) {
  const long HW = (long)H * W, P = (long)NF * HW;
  int worst = 0, it = 0;
  for (long pos0 = 0; pos0 < P; pos0 += KP_NPOS) {
    if (pos0 > 0 && pos0 % HW == 0) break;
    if (++it > 4096) return 1 << 30;                  // no short period: not a geometry this kernel is meant for
    const long last = pos0 + KP_NPOS - 1 < P ? pos0 + KP_NPOS - 1 : P - 1;
    const long v_lo = pos0 / W, v_last = last / W;
    const int lines = (int)(v_last - v_lo + 1 + 2 * (v_last / H - v_lo / H + 1));
    if (lines > worst) worst = lines;
  }
  return worst * (W + 2);
}

}  // namespace cstp
