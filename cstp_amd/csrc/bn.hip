// Train-mode BatchNorm (3d and 1d) fused with the residual add and ReLU that follow it, forward and
// backward.  HBM-streaming kernels: x is read once for the statistics (fp64 accumulators, wave
// shuffles + LDS for the cross-lane part, deterministic two-stage reduction -- no atomics) and once
// more for the normalise/activate pass.
//
// Layout: x[n][c][s], s = D*H*W contiguous.  For s == 1 (BatchNorm1d over [B][F]) a dedicated
// single-launch kernel keeps lanes along the contiguous feature axis.
//
// GROUPS: the batch may hold `groups` independent BN calls back to back (n = groups * n_per_group);
// statistics are per (group, channel) and the running stats are updated group after group, exactly
// as `groups` successive F.batch_norm calls would.  The model uses groups = 2 to push both views
// of a clip pair through one launch sequence (same weights, per-view statistics -- identical maths
// to the reference's two calls, r21d_byol.py:359-360, with half the launches and twice the grid).
#include <stdlib.h>

#include "common.h"

namespace cstp {

#ifndef CSTP_BN_NT
#define CSTP_BN_NT 1       // streaming (non-temporal) vector stores / last-use loads in the apply kernels
#endif
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4(float* p, const float4& v) {
#if CSTP_BN_NT
  __builtin_nontemporal_store(f32x4v{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4v*>(p));
#else
  *reinterpret_cast<float4*>(p) = v;
#endif
}
__device__ __forceinline__ float4 ld4_last(const float* p) {
#if CSTP_BN_NT
  const f32x4v v = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *reinterpret_cast<const float4*>(p);
#endif
}


static inline int bn_nsplit(int n, int c) {
  int ns = cdiv(2048, c);
  if (ns > n) ns = n;
  if (ns < 1) ns = 1;
  return ns;
}

// ---- stage 1: per (channel, split) partial sums ------------------------------------------------
// MODE 0: (sum x, sum x^2)          MODE 1: (sum g, sum g*xhat), g = dy * (relu ? y>0 : 1)
template <int MODE, bool VEC4>
__global__ void __launch_bounds__(256)
bn_reduce_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                 const float* __restrict__ mean, const float* __restrict__ invstd, double* __restrict__ part, int npg,
                 int c, int s, int nsplit, int relu, const float2* __restrict__ ss, unsigned* __restrict__ cell = nullptr,
                 float* __restrict__ gout = nullptr) {
  // gout (MODE 1): the masked gradient g is also WRITTEN there -- it is the residual branch's gradient, and the apply pass then
  // reads (x, g) instead of (x, y, dy): seven passes over the tensor instead of eight
  __shared__ double sm[16];
  // (the absmax cell of the apply pass two launches on: absmax_fold_kernel takes the maximum into it)
  if (cell != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *cell = 0;
  const int ch = blockIdx.x, grp = blockIdx.y / nsplit, j = blockIdx.y - grp * nsplit;
  double a0 = 0.0, a1 = 0.0;
  float mu = 0.f, is = 0.f;
  if (MODE == 1) { mu = mean[grp * c + ch]; is = invstd[grp * c + ch]; }
  // ReLU mask: sign of the forward OUTPUT y when it was materialised, else recomputed from x with the
  // (scale, shift) the consumer convolution applied in its gather (fused BN->ReLU->conv)
  const bool remask = (MODE == 1) && relu && (y == nullptr);
  float sc = 0.f, sh = 0.f;
  if (remask) { const float2 t2 = ss[grp * c + ch]; sc = t2.x; sh = t2.y; }
  for (int rr = j; rr < npg; rr += nsplit) {
    const int row = grp * npg + rr;
    const size_t base = ((size_t)row * c + ch) * s;
    if (VEC4) {
      const float4* xp = reinterpret_cast<const float4*>(x + base);
      const float4* yp = reinterpret_cast<const float4*>(y + base);
      const float4* gp = reinterpret_cast<const float4*>(dy + base);
      const int s4 = s >> 2;
      for (int i = threadIdx.x; i < s4; i += 256) {
        const float4 v = xp[i];
        if (MODE == 0) {
          a0 += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
          a1 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        } else {
          float4 g = gp[i];
          if (remask) {
            g.x = (v.x * sc + sh) > 0.f ? g.x : 0.f; g.y = (v.y * sc + sh) > 0.f ? g.y : 0.f;
            g.z = (v.z * sc + sh) > 0.f ? g.z : 0.f; g.w = (v.w * sc + sh) > 0.f ? g.w : 0.f;
          } else if (relu) {
            const float4 o = yp[i];
            g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f;
            g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
          }
          if (gout != nullptr) reinterpret_cast<float4*>(gout + base)[i] = g;
          a0 += (double)g.x + (double)g.y + (double)g.z + (double)g.w;
          a1 += (double)(g.x * ((v.x - mu) * is)) + (double)(g.y * ((v.y - mu) * is)) +
                (double)(g.z * ((v.z - mu) * is)) + (double)(g.w * ((v.w - mu) * is));
        }
      }
    } else {
      for (int i = threadIdx.x; i < s; i += 256) {
        const float v = x[base + i];
        if (MODE == 0) {
          a0 += (double)v; a1 += (double)v * v;
        } else {
          float g = dy[base + i];
          if (remask) { if (!((v * sc + sh) > 0.f)) g = 0.f; }
          else if (relu && !(y[base + i] > 0.f)) g = 0.f;
          if (gout != nullptr) gout[base + i] = g;
          a0 += (double)g; a1 += (double)(g * ((v - mu) * is));
        }
      }
    }
  }
  a0 = block_sum(a0, sm);
  a1 = block_sum(a1, sm);
  if (threadIdx.x == 0) {
    part[((size_t)ch * gridDim.y + blockIdx.y) * 2 + 0] = a0;
    part[((size_t)ch * gridDim.y + blockIdx.y) * 2 + 1] = a1;
  }
}

// ---- stage 2 (forward): mean / invstd / running stats ------------------------------------------
__global__ void bn_finalize_fwd_kernel(const double* __restrict__ part, float* __restrict__ save_mean,
                                       float* __restrict__ save_invstd, float* __restrict__ running_mean,
                                       float* __restrict__ running_var, int c, int groups, int nsplit, double count,
                                       float eps, float momentum, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float2* __restrict__ ss, unsigned* __restrict__ cell) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch == 0 && cell != nullptr) *cell = 0;          // absmax_fold_kernel takes the maximum into it
  if (ch >= c) return;
  float rm = 0.f, rv = 0.f;
  if (running_mean != nullptr) { rm = running_mean[ch]; rv = running_var[ch]; }
  for (int g = 0; g < groups; ++g) {
    double s0 = 0.0, s1 = 0.0;
    const double* p = part + ((size_t)ch * groups + g) * nsplit * 2;
    for (int j = 0; j < nsplit; ++j) { s0 += p[2 * j]; s1 += p[2 * j + 1]; }
    const double mu = s0 / count;
    double var = s1 / count - mu * mu;
    if (var < 0.0) var = 0.0;
    save_mean[g * c + ch] = (float)mu;
    const float isf = (float)(1.0 / sqrt(var + (double)eps));
    save_invstd[g * c + ch] = isf;
    if (ss != nullptr) {                 // the affine form bn_apply uses: y = x*scale + shift
      const float scl = isf * gamma[ch];
      ss[g * c + ch] = make_float2(scl, beta[ch] - (float)mu * scl);
    }
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    rm = (float)((1.0 - momentum) * rm + momentum * mu);      // group after group, like successive calls
    rv = (float)((1.0 - momentum) * rv + momentum * unb);
  }
  if (running_mean != nullptr) { running_mean[ch] = rm; running_var[ch] = rv; }
}

__device__ __forceinline__ float key_to_float(unsigned k) {        // inverse of igemm_patch.h's key_of_float
  return __builtin_bit_cast(float, (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// ... the same for MANY partials per channel (the sums a convolution's persistent blocks left: up to 256 per channel and group):
// one wave per channel, lanes stride over the partials
__global__ void __launch_bounds__(64)
bn_finalize_fwd_wide_kernel(const double* __restrict__ part, float* __restrict__ save_mean, float* __restrict__ save_invstd,
                            float* __restrict__ running_mean, float* __restrict__ running_var, int c, int groups, int nsplit,
                            double count, float eps, float momentum, const float* __restrict__ gamma,
                            const float* __restrict__ beta, float2* __restrict__ ss, unsigned* __restrict__ cell,
                            const unsigned* __restrict__ mm, int relu) {
  const int ch = blockIdx.x, lane = threadIdx.x;
  // mm == null: the apply pass that follows measures its output, absmax_fold_kernel takes the maximum into the zeroed cell.
  // mm != null (cstp_bn_finalize_pre): no apply pass follows -- the largest magnitude of act(x * scale + shift) comes from the
  // channel's smallest and largest x (the map is monotone in x, and so is its fp32 rounding: the extreme outputs ARE the
  // outputs of the extreme inputs), one atomicMax per channel into the cell the producing launch zeroed.
  if (ch == 0 && lane == 0 && cell != nullptr && mm == nullptr) *cell = 0;
  unsigned zmax = 0;
  float rm = 0.f, rv = 0.f;
  if (running_mean != nullptr) { rm = running_mean[ch]; rv = running_var[ch]; }
  // the partials are sums of (x - pivot) and (x - pivot)^2; the pivot the producing launch used sits behind them
  const double pivot = part[(size_t)c * groups * nsplit * 2 + ch];
  for (int g = 0; g < groups; ++g) {
    double s0 = 0.0, s1 = 0.0;
    const double* p = part + ((size_t)ch * groups + g) * nsplit * 2;
    for (int j = lane; j < nsplit; j += 64) { s0 += p[2 * j]; s1 += p[2 * j + 1]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); }
    const double dm = s0 / count;
    const double mu = pivot + dm;
    double var = s1 / count - dm * dm;
    if (var < 0.0) var = 0.0;
    const float isf = (float)(1.0 / sqrt(var + (double)eps));
    if (lane == 0) {
      save_mean[g * c + ch] = (float)mu;
      save_invstd[g * c + ch] = isf;
      if (ss != nullptr) {
        const float scl = isf * gamma[ch];
        ss[g * c + ch] = make_float2(scl, beta[ch] - (float)mu * scl);
      }
    }
    if (mm != nullptr) {
      unsigned kmin = 0xffffffffu, kmax = 0u;
      const unsigned* q = mm + ((size_t)ch * groups + g) * nsplit * 2;
      for (int j = lane; j < nsplit; j += 64) {
        const unsigned a = q[2 * j], b = q[2 * j + 1];
        kmin = kmin < a ? kmin : a; kmax = kmax > b ? kmax : b;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const unsigned a = (unsigned)__shfl_xor((int)kmin, off, 64), b = (unsigned)__shfl_xor((int)kmax, off, 64);
        kmin = kmin < a ? kmin : a; kmax = kmax > b ? kmax : b;
      }
      // the consumer's gather computes fma(x, scale, shift) with exactly these two table entries
      const float scl = isf * gamma[ch];
      const float sh = beta[ch] - (float)mu * scl;
      float z0 = __builtin_fmaf(key_to_float(kmin), scl, sh), z1 = __builtin_fmaf(key_to_float(kmax), scl, sh);
      if (relu) { z0 = fmaxf(z0, 0.f); z1 = fmaxf(z1, 0.f); }
      const unsigned a0 = __builtin_bit_cast(unsigned, z0) & 0x7fffffffu, a1 = __builtin_bit_cast(unsigned, z1) & 0x7fffffffu;
      const unsigned a = a0 > a1 ? a0 : a1;
      zmax = zmax > a ? zmax : a;
    }
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    rm = (float)((1.0 - momentum) * rm + momentum * mu);
    rv = (float)((1.0 - momentum) * rv + momentum * unb);
  }
  if (lane == 0 && running_mean != nullptr) { running_mean[ch] = rm; running_var[ch] = rv; }
  if (lane == 0 && mm != nullptr && cell != nullptr && zmax != 0) atomicMax(cell, zmax);
}

// ---- stage 3: elementwise passes.  One block = one chunk of ONE (sample, channel) row, so the per-channel
//      constants are block-uniform scalars and no per-element index division is needed (the flat-index form
//      spent ~100 VALU instructions per float4 on 64-bit divisions and was VALU- rather than HBM-bound).
constexpr int BN_UNROLL = 4;    // vectors per thread per block

// optional by-product of an apply pass: the largest magnitude of the tensor it writes (fp32 bits, sign cleared), for the
// consumer convolution's 2xf16-split operand scale (cstp_conv3d_*_am).  Every wave stores its own maximum into a slot of the
// workspace (no atomics: same-address atomics execute at the memory side and serialise) and a one-block kernel folds the
// slots into the caller's cell.
__device__ __forceinline__ unsigned abs_bits(float v) { return __builtin_bit_cast(unsigned, v) & 0x7fffffffu; }
__device__ __forceinline__ unsigned umax4(unsigned m, const float4& v) {
  unsigned a = abs_bits(v.x), b = abs_bits(v.y), c = abs_bits(v.z), d = abs_bits(v.w);
  a = a > b ? a : b; c = c > d ? c : d; a = a > c ? a : c;
  return m > a ? m : a;
}
__device__ __forceinline__ unsigned wave_umax(unsigned mx) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)mx, off, 64); mx = mx > o ? mx : o; }
  return mx;
}
__device__ __forceinline__ void absmax_slot(unsigned mx, unsigned* slots) {
  mx = wave_umax(mx);
  if ((threadIdx.x & 63) == 0) slots[blockIdx.x * 4 + (threadIdx.x >> 6)] = mx;
}
// FOLD_BLOCKS blocks, one atomic each into the cell the finalize kernel zeroed (a handful of atomics: no contention to speak of)
constexpr int FOLD_BLOCKS = 32;
__global__ void __launch_bounds__(256) absmax_fold_kernel(const unsigned* __restrict__ slots, int n, unsigned* __restrict__ cell) {
  __shared__ unsigned red[4];
  unsigned mx = 0;
  const uint4* s4 = reinterpret_cast<const uint4*>(slots);            // n is a multiple of 4 (4 slots per block), 256-byte aligned
  for (int i = blockIdx.x * 256 + threadIdx.x; i < (n >> 2); i += FOLD_BLOCKS * 256) {
    const uint4 v = s4[i];
    unsigned a = v.x > v.y ? v.x : v.y, b = v.z > v.w ? v.z : v.w;
    a = a > b ? a : b;
    mx = mx > a ? mx : a;
  }
  mx = wave_umax(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned a = red[0] > red[1] ? red[0] : red[1], b = red[2] > red[3] ? red[2] : red[3];
    a = a > b ? a : b;
    if (a != 0) atomicMax(cell, a);
  }
}

// The finalize step folded into the apply passes (it used to be a launch of its own between the reduction and the apply pass:
// ~70 launches of a few microseconds per R(2+1)D-18 step on the critical chain).  `fin.part` != null: wave 0 of every block sums
// the partials of ITS (channel, group) -- sequentially, in the order bn_finalize_*_kernel uses: the same bits -- while the
// block's x loads are already in flight; the block of the channel's first row and chunk also does the channel's bookkeeping.
struct BnFin {
  const double* part;     // [c][groups][nsplit][2]; null = read the finished per-channel arrays instead
  int nsplit, groups;
  double count;
  float eps, momentum;
  float* running_mean;    // forward bookkeeping (may be null)
  float* running_var;
  float* save_mean;
  float* save_invstd;
  float2* ss;             // may be null
  float* dgamma;          // backward bookkeeping
  float* dbeta;
  int accumulate;
};

__device__ __forceinline__ void bn_sum_partials(const double* __restrict__ part, int ch, int groups, int g, int nsplit, double& s0,
                                               double& s1) {
  const double* p = part + ((size_t)ch * groups + g) * nsplit * 2;
  s0 = 0.0; s1 = 0.0;
  for (int j = 0; j < nsplit; ++j) { s0 += p[2 * j]; s1 += p[2 * j + 1]; }
}

// forward: y = act((x-mean)*invstd*gamma + beta + residual)
template <bool VEC4>
__global__ void __launch_bounds__(256)
bn_apply_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res, float* __restrict__ y,
                    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                    const float* __restrict__ invstd, int c, int s, int npg, int relu, int chunks,
                    unsigned* __restrict__ cell, const BnFin fin) {
  constexpr int W = VEC4 ? 4 : 1;
  __shared__ float s_ss[2];
  const int row = blockIdx.x / chunks, chunk = blockIdx.x - row * chunks;
  const int ch = row % c, grp = (row / c) / npg, gc = grp * c + ch;
  const size_t base = (size_t)row * s;
  // the block's x values first: in flight while wave 0 folds the statistics
  float4 xv[BN_UNROLL];
  float xs1[BN_UNROLL];
#pragma unroll
  for (int u = 0; u < BN_UNROLL; ++u) {
    const int e = (chunk * BN_UNROLL * 256 + u * 256 + threadIdx.x) * W;
    xv[u] = make_float4(0.f, 0.f, 0.f, 0.f); xs1[u] = 0.f;
    if (e < s) { if (VEC4) xv[u] = ld4_last(x + base + e); else xs1[u] = x[base + e]; }
  }
  float sc, sh;
  if (fin.part != nullptr) {
    if (threadIdx.x == 0) {
      const float ga = gamma[ch], be = beta[ch];
      double s0, s1;
      bn_sum_partials(fin.part, ch, fin.groups, grp, fin.nsplit, s0, s1);
      double mu = s0 / fin.count;
      double var = s1 / fin.count - mu * mu;
      if (var < 0.0) var = 0.0;
      float isf = (float)(1.0 / sqrt(var + (double)fin.eps));
      { const float scl = isf * ga; s_ss[0] = scl; s_ss[1] = be - (float)mu * scl; }
      if (row == ch && chunk == 0) {                  // the channel's bookkeeping, once: exactly bn_finalize_fwd_kernel
        float rm = 0.f, rv = 0.f;
        if (fin.running_mean != nullptr) { rm = fin.running_mean[ch]; rv = fin.running_var[ch]; }
        for (int g = 0; g < fin.groups; ++g) {
          bn_sum_partials(fin.part, ch, fin.groups, g, fin.nsplit, s0, s1);
          mu = s0 / fin.count;
          var = s1 / fin.count - mu * mu;
          if (var < 0.0) var = 0.0;
          fin.save_mean[g * c + ch] = (float)mu;
          isf = (float)(1.0 / sqrt(var + (double)fin.eps));
          fin.save_invstd[g * c + ch] = isf;
          if (fin.ss != nullptr) {
            const float scl = isf * ga;
            fin.ss[g * c + ch] = make_float2(scl, be - (float)mu * scl);
          }
          const double unb = fin.count > 1.0 ? var * fin.count / (fin.count - 1.0) : var;
          rm = (float)((1.0 - fin.momentum) * rm + fin.momentum * mu);      // group after group, like successive calls
          rv = (float)((1.0 - fin.momentum) * rv + fin.momentum * unb);
        }
        if (fin.running_mean != nullptr) { fin.running_mean[ch] = rm; fin.running_var[ch] = rv; }
      }
    }
    __syncthreads();
    sc = s_ss[0]; sh = s_ss[1];
  } else {
    sc = invstd[gc] * gamma[ch];
    sh = beta[ch] - mean[gc] * sc;
  }
  unsigned mx = 0;
#pragma unroll
  for (int u = 0; u < BN_UNROLL; ++u) {
    const int e = (chunk * BN_UNROLL * 256 + u * 256 + threadIdx.x) * W;
    if (e >= s) continue;
    if (VEC4) {
      float4 v = xv[u];
      v.x = v.x * sc + sh; v.y = v.y * sc + sh; v.z = v.z * sc + sh; v.w = v.w * sc + sh;
      if (res != nullptr) {
        const float4 r = *reinterpret_cast<const float4*>(res + base + e);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      st4(y + base + e, v);
      mx = umax4(mx, v);
    } else {
      float v = xs1[u] * sc + sh;
      if (res != nullptr) v += res[base + e];
      if (relu) v = fmaxf(v, 0.f);
      y[base + e] = v;
      const unsigned a = abs_bits(v);
      mx = mx > a ? mx : a;
    }
  }
  if (cell != nullptr) absmax_slot(mx, cell);
}

// backward: dx = gamma*invstd*(g - dbeta/cnt - xhat*dgamma/cnt); dres = g
template <bool VEC4>
__global__ void __launch_bounds__(256)
bn_apply_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
                    const float* __restrict__ gsum, float* __restrict__ dx,
                    float* __restrict__ dres, int c, int s, int npg, float inv_count, int relu,
                    const float2* __restrict__ ss, int chunks, unsigned* __restrict__ cell, const BnFin fin) {
  constexpr int W = VEC4 ? 4 : 1;
  __shared__ float s_g[2];
  unsigned mx = 0;
  const bool remask = relu && (y == nullptr);
  const int row = blockIdx.x / chunks, chunk = blockIdx.x - row * chunks;
  const int ch = row % c, grp = (row / c) / npg, gc = grp * c + ch;
  const size_t base = (size_t)row * s;
  // the block's x and dy values first: in flight while thread 0 folds the (sum g, sum g * xhat) partials
  float4 xv[BN_UNROLL], gv[BN_UNROLL];
  float xs1[BN_UNROLL], gs1[BN_UNROLL];
#pragma unroll
  for (int u = 0; u < BN_UNROLL; ++u) {
    const int e = (chunk * BN_UNROLL * 256 + u * 256 + threadIdx.x) * W;
    xv[u] = gv[u] = make_float4(0.f, 0.f, 0.f, 0.f); xs1[u] = gs1[u] = 0.f;
    if (e < s) {
      if (VEC4) { xv[u] = ld4_last(x + base + e); gv[u] = ld4_last(dy + base + e); }
      else { xs1[u] = x[base + e]; gs1[u] = dy[base + e]; }
    }
  }
  float sum_g, sum_gx;
  if (fin.part != nullptr) {
    if (threadIdx.x == 0) {
      double s0, s1;
      bn_sum_partials(fin.part, ch, fin.groups, grp, fin.nsplit, s0, s1);
      s_g[0] = (float)s0; s_g[1] = (float)s1;
      if (row == ch && chunk == 0) {                  // dgamma / dbeta of the channel, once: exactly bn_finalize_bwd_kernel
        double t0 = 0.0, t1 = 0.0;
        for (int g = 0; g < fin.groups; ++g) {
          bn_sum_partials(fin.part, ch, fin.groups, g, fin.nsplit, s0, s1);
          t0 += s0; t1 += s1;
        }
        fin.dbeta[ch] = (fin.accumulate ? fin.dbeta[ch] : 0.f) + (float)t0;
        fin.dgamma[ch] = (fin.accumulate ? fin.dgamma[ch] : 0.f) + (float)t1;
      }
    }
    __syncthreads();
    sum_g = s_g[0]; sum_gx = s_g[1];
  } else {
    sum_g = gsum[gc * 2]; sum_gx = gsum[gc * 2 + 1];
  }
  const float mu = mean[gc], is = invstd[gc];
  const float k = gamma[ch] * is;
  const float mb = sum_g * inv_count, mg = sum_gx * inv_count;
  float sc = 0.f, sh = 0.f;
  if (remask) { const float2 t2 = ss[gc]; sc = t2.x; sh = t2.y; }
#pragma unroll
  for (int u = 0; u < BN_UNROLL; ++u) {
    const int e = (chunk * BN_UNROLL * 256 + u * 256 + threadIdx.x) * W;
    if (e >= s) continue;
    if (VEC4) {
      const float4 v = xv[u];
      float4 g = gv[u];
      if (remask) {
        g.x = (v.x * sc + sh) > 0.f ? g.x : 0.f; g.y = (v.y * sc + sh) > 0.f ? g.y : 0.f;
        g.z = (v.z * sc + sh) > 0.f ? g.z : 0.f; g.w = (v.w * sc + sh) > 0.f ? g.w : 0.f;
      } else if (relu) {
        const float4 o = *reinterpret_cast<const float4*>(y + base + e);
        g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f;
        g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
      }
      if (dres != nullptr) st4(dres + base + e, g);
      float4 o;
      o.x = k * (g.x - mb - (v.x - mu) * is * mg);
      o.y = k * (g.y - mb - (v.y - mu) * is * mg);
      o.z = k * (g.z - mb - (v.z - mu) * is * mg);
      o.w = k * (g.w - mb - (v.w - mu) * is * mg);
      st4(dx + base + e, o);
      mx = umax4(mx, o);
    } else {
      float g = gs1[u];
      if (remask) { if (!((xs1[u] * sc + sh) > 0.f)) g = 0.f; }
      else if (relu && !(y[base + e] > 0.f)) g = 0.f;
      if (dres != nullptr) dres[base + e] = g;
      const float o = k * (g - mb - (xs1[u] - mu) * is * mg);
      dx[base + e] = o;
      const unsigned a = abs_bits(o);
      mx = mx > a ? mx : a;
    }
  }
  if (cell != nullptr) absmax_slot(mx, cell);
}

// ---- SMALL tensors (npg * s <= BN_SMALL_E = 16 384 values per channel and group: the 7 x 7 and 14 x 14 stages) ----------------
// The three-launch sequence above costs such a tensor 35 - 60 us of launch latencies and near-empty blocks (one block per 98-value
// row).  Here ONE block owns a channel: group after group it holds the group's values in registers, sums them (fp64, fixed
// order: thread t takes values t, t + 256, ...; wave shuffles, then the four wave sums in wave order), does the channel's
// bookkeeping exactly as bn_finalize_*_kernel, applies and writes.  The absmax by-product leaves through the slots
// (absmax_store_kernel: one block, a plain store -- nothing has zeroed the cell here).
constexpr int BN_SMALL_PT = 16;                     // values per thread and group
constexpr int BN_SMALL_E = BN_SMALL_PT * 1024;      // ... with blocks of 256 (<= 4096 values) or 1024 threads

__device__ __forceinline__ unsigned bn_small_off(int e, int s, int g, int npg, int c, int ch) {
  const int r = e / s, i = e - r * s;
  return ((unsigned)(g * npg + r) * c + ch) * s + i;       // (host: the tensor has < 2^31 elements)
}

template <int TPB>
__global__ void __launch_bounds__(TPB)
bn_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res, float* __restrict__ y,
                    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ running_mean,
                    float* __restrict__ running_var, float* __restrict__ save_mean, float* __restrict__ save_invstd,
                    float2* __restrict__ ss, int c, int s, int npg, int groups, float eps, float momentum, int relu,
                    unsigned* __restrict__ slots) {
  __shared__ double sm[16];
  __shared__ float s_ss[2];
  const int ch = blockIdx.x, E = npg * s;
  const double count = (double)E;
  const float ga = gamma[ch], be = beta[ch];
  float rm = 0.f, rv = 0.f;
  if (threadIdx.x == 0 && running_mean != nullptr) { rm = running_mean[ch]; rv = running_var[ch]; }
  unsigned mx = 0;
  for (int g = 0; g < groups; ++g) {
    float v[BN_SMALL_PT];
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int u = 0; u < BN_SMALL_PT; ++u) {
      const int e = u * TPB + threadIdx.x;
      v[u] = e < E ? x[bn_small_off(e, s, g, npg, c, ch)] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < BN_SMALL_PT; ++u) { a0 += (double)v[u]; a1 += (double)v[u] * v[u]; }
    a0 = block_sum(a0, sm);
    a1 = block_sum(a1, sm);
    if (threadIdx.x == 0) {
      const double mu = a0 / count;
      double var = a1 / count - mu * mu;
      if (var < 0.0) var = 0.0;
      save_mean[g * c + ch] = (float)mu;
      const float isf = (float)(1.0 / sqrt(var + (double)eps));
      save_invstd[g * c + ch] = isf;
      const float scl = isf * ga;
      s_ss[0] = scl; s_ss[1] = be - (float)mu * scl;
      if (ss != nullptr) ss[g * c + ch] = make_float2(scl, be - (float)mu * scl);
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      rm = (float)((1.0 - momentum) * rm + momentum * mu);      // group after group, like successive calls
      rv = (float)((1.0 - momentum) * rv + momentum * unb);
    }
    __syncthreads();
    const float sc = s_ss[0], sh = s_ss[1];
#pragma unroll
    for (int u = 0; u < BN_SMALL_PT; ++u) {
      const int e = u * TPB + threadIdx.x;
      if (e >= E) continue;
      const unsigned off = bn_small_off(e, s, g, npg, c, ch);
      float o = v[u] * sc + sh;
      if (res != nullptr) o += res[off];
      if (relu) o = fmaxf(o, 0.f);
      y[off] = o;
      const unsigned a = abs_bits(o);
      mx = mx > a ? mx : a;
    }
    __syncthreads();                                  // s_ss is rewritten by the next group
  }
  if (threadIdx.x == 0 && running_mean != nullptr) { running_mean[ch] = rm; running_var[ch] = rv; }
  if (slots != nullptr) {                             // 16 slots per block (the fold takes the maximum; unused ones hold 0)
    mx = wave_umax(mx);
    if ((threadIdx.x & 63) == 0) slots[blockIdx.x * 16 + (threadIdx.x >> 6)] = mx;
    else if (threadIdx.x < 16 && threadIdx.x >= TPB / 64) slots[blockIdx.x * 16 + threadIdx.x] = 0;
  }
}

template <int TPB>
__global__ void __launch_bounds__(TPB)
bn_small_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
                    const float2* __restrict__ ss, float* __restrict__ dx, float* __restrict__ dres,
                    float* __restrict__ dgamma, float* __restrict__ dbeta, int c, int s, int npg, int groups, int relu,
                    int accumulate, unsigned* __restrict__ slots) {
  __shared__ double sm[16];
  __shared__ float s_g[2];
  const int ch = blockIdx.x, E = npg * s;
  const float inv_count = (float)(1.0 / (double)E);
  const bool remask = relu && (y == nullptr);
  const float ga = gamma[ch];
  double t0 = 0.0, t1 = 0.0;
  unsigned mx = 0;
  for (int g = 0; g < groups; ++g) {
    const int gc = g * c + ch;
    const float mu = mean[gc], is = invstd[gc];
    float sc = 0.f, sh = 0.f;
    if (remask) { const float2 t2 = ss[gc]; sc = t2.x; sh = t2.y; }
    float v[BN_SMALL_PT], gr[BN_SMALL_PT];
#pragma unroll
    for (int u = 0; u < BN_SMALL_PT; ++u) {
      const int e = u * TPB + threadIdx.x;
      v[u] = 0.f; gr[u] = 0.f;
      if (e < E) {
        const unsigned off = bn_small_off(e, s, g, npg, c, ch);
        v[u] = x[off];
        float gg = dy[off];
        if (remask) { if (!((v[u] * sc + sh) > 0.f)) gg = 0.f; }
        else if (relu && !(y[off] > 0.f)) gg = 0.f;
        gr[u] = gg;
      }
    }
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int u = 0; u < BN_SMALL_PT; ++u) { a0 += (double)gr[u]; a1 += (double)(gr[u] * ((v[u] - mu) * is)); }
    a0 = block_sum(a0, sm);
    a1 = block_sum(a1, sm);
    if (threadIdx.x == 0) { s_g[0] = (float)a0; s_g[1] = (float)a1; t0 += a0; t1 += a1; }
    __syncthreads();
    const float mb = s_g[0] * inv_count, mg = s_g[1] * inv_count;
    const float k = ga * is;
#pragma unroll
    for (int u = 0; u < BN_SMALL_PT; ++u) {
      const int e = u * TPB + threadIdx.x;
      if (e >= E) continue;
      const unsigned off = bn_small_off(e, s, g, npg, c, ch);
      if (dres != nullptr) dres[off] = gr[u];
      const float o = k * (gr[u] - mb - (v[u] - mu) * is * mg);
      dx[off] = o;
      const unsigned a = abs_bits(o);
      mx = mx > a ? mx : a;
    }
    __syncthreads();                                  // s_g is rewritten by the next group
  }
  if (threadIdx.x == 0) {
    dbeta[ch] = (accumulate ? dbeta[ch] : 0.f) + (float)t0;
    dgamma[ch] = (accumulate ? dgamma[ch] : 0.f) + (float)t1;
  }
  if (slots != nullptr) {
    mx = wave_umax(mx);
    if ((threadIdx.x & 63) == 0) slots[blockIdx.x * 16 + (threadIdx.x >> 6)] = mx;
    else if (threadIdx.x < 16 && threadIdx.x >= TPB / 64) slots[blockIdx.x * 16 + threadIdx.x] = 0;
  }
}

// one block: the maximum of n slots (a multiple of 4) STORED into the cell
__global__ void __launch_bounds__(256) absmax_store_kernel(const unsigned* __restrict__ slots, int n, unsigned* __restrict__ cell) {
  __shared__ unsigned red[4];
  unsigned mx = 0;
  const uint4* s4 = reinterpret_cast<const uint4*>(slots);
  for (int i = threadIdx.x; i < (n >> 2); i += 256) {
    const uint4 v = s4[i];
    unsigned a = v.x > v.y ? v.x : v.y, b = v.z > v.w ? v.z : v.w;
    a = a > b ? a : b;
    mx = mx > a ? mx : a;
  }
  mx = wave_umax(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned a = red[0] > red[1] ? red[0] : red[1], b = red[2] > red[3] ? red[2] : red[3];
    *cell = a > b ? a : b;
  }
}

// ---- BatchNorm1d (s == 1): one thread per feature, lanes along the contiguous feature axis ------
__global__ void bn1d_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res, float* __restrict__ y,
                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                float* __restrict__ running_mean, float* __restrict__ running_var,
                                float* __restrict__ save_mean, float* __restrict__ save_invstd, int npg, int groups, int c,
                                float eps, float momentum, int relu) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  float rm = 0.f, rv = 0.f;
  if (running_mean != nullptr) { rm = running_mean[ch]; rv = running_var[ch]; }
  const float ga = gamma[ch], be = beta[ch];
  for (int g = 0; g < groups; ++g) {
    const size_t r0 = (size_t)g * npg;
    double s0 = 0.0, s1 = 0.0;
    for (int r = 0; r < npg; ++r) { const double v = x[(r0 + r) * c + ch]; s0 += v; s1 += v * v; }
    const double mu = s0 / npg;
    double var = s1 / npg - mu * mu;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[g * c + ch] = (float)mu;
    save_invstd[g * c + ch] = is;
    const double unb = npg > 1 ? var * npg / (npg - 1.0) : var;
    rm = (float)((1.0 - momentum) * rm + momentum * mu);
    rv = (float)((1.0 - momentum) * rv + momentum * unb);
    const float sc = is * ga, sh = be - (float)mu * sc;
    for (int r = 0; r < npg; ++r) {
      float v = x[(r0 + r) * c + ch] * sc + sh;
      if (res != nullptr) v += res[(r0 + r) * c + ch];
      if (relu) v = fmaxf(v, 0.f);
      y[(r0 + r) * c + ch] = v;
    }
  }
  if (running_mean != nullptr) { running_mean[ch] = rm; running_var[ch] = rv; }
}

__global__ void bn1d_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                const float* __restrict__ gamma, const float* __restrict__ mean,
                                const float* __restrict__ invstd, float* __restrict__ dx, float* __restrict__ dres,
                                float* __restrict__ dgamma, float* __restrict__ dbeta, int npg, int groups, int c, int relu,
                                int accumulate) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  double t0 = 0.0, t1 = 0.0;
  const float ga = gamma[ch];
  for (int g = 0; g < groups; ++g) {
    const size_t r0 = (size_t)g * npg;
    const float mu = mean[g * c + ch], is = invstd[g * c + ch];
    double s0 = 0.0, s1 = 0.0;
    for (int r = 0; r < npg; ++r) {
      float gr = dy[(r0 + r) * c + ch];
      if (relu && !(y[(r0 + r) * c + ch] > 0.f)) gr = 0.f;
      s0 += (double)gr;
      s1 += (double)(gr * ((x[(r0 + r) * c + ch] - mu) * is));
    }
    t0 += s0; t1 += s1;
    const float k = ga * is, mb = (float)s0 / npg, mg = (float)s1 / npg;
    for (int r = 0; r < npg; ++r) {
      float gr = dy[(r0 + r) * c + ch];
      if (relu && !(y[(r0 + r) * c + ch] > 0.f)) gr = 0.f;
      if (dres != nullptr) dres[(r0 + r) * c + ch] = gr;
      dx[(r0 + r) * c + ch] = k * (gr - mb - (x[(r0 + r) * c + ch] - mu) * is * mg);
    }
  }
  dbeta[ch] = (accumulate ? dbeta[ch] : 0.f) + (float)t0;
  dgamma[ch] = (accumulate ? dgamma[ch] : 0.f) + (float)t1;
}


// ---- eval mode: the running statistics are the statistics (r21d_byol.py cls/val/test under model.eval()) ----
__global__ void bn_eval_prepare_kernel(const float* __restrict__ running_var, float* __restrict__ invstd, int c, float eps,
                                       unsigned* __restrict__ cell) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && cell != nullptr) *cell = 0;      // absmax_fold_kernel takes the maximum into it
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch < c) invstd[ch] = 1.0f / sqrtf(running_var[ch] + eps);
}

__global__ void bn1d_eval_kernel(const float* __restrict__ x, const float* __restrict__ res, float* __restrict__ y,
                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                 const float* __restrict__ running_mean, const float* __restrict__ running_var, int n, int c,
                                 float eps, int relu) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  const float sc = gamma[ch] / sqrtf(running_var[ch] + eps);
  const float sh = beta[ch] - running_mean[ch] * sc;
  for (int r = 0; r < n; ++r) {
    float v = x[(size_t)r * c + ch] * sc + sh;
    if (res != nullptr) v += res[(size_t)r * c + ch];
    if (relu) v = fmaxf(v, 0.f);
    y[(size_t)r * c + ch] = v;
  }
}


}  // namespace cstp

using namespace cstp;

static size_t bn_slot_bytes(int n, int c, int s) {      // 4 per block of an apply pass; 16 per channel for the small-tensor kernels
  size_t slots = (size_t)n * c * cdiv(s, 1024) * 4;
  if (slots < (size_t)c * 16) slots = (size_t)c * 16;
  return align_up(slots * sizeof(unsigned), 256);
}
static unsigned* bn_slots(void* ws, int n, int c, int groups) {
  const int npg = n / groups;
  return reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + align_up((size_t)c * groups * bn_nsplit(npg, c) * 2 * sizeof(double), 256) +
                                     align_up((size_t)groups * c * 2 * sizeof(float), 256));
}

// CSTP_BN_SMALL=0: every BatchNorm3d through the three-launch sequence (A/B switch; read once)
static bool bn_small_enabled() {
  static const bool on = [] { const char* e = getenv("CSTP_BN_SMALL"); return !(e && e[0] == '0'); }();
  return on;
}
// the single-launch kernels: <= 16 384 values per (channel, group), 32-bit element offsets, 16 slots per channel in the workspace
static bool bn_small_ok(int n, int c, int s, int groups) {
  const size_t e = (size_t)(n / groups) * s;
  return bn_small_enabled() && e <= (size_t)BN_SMALL_E && (size_t)n * c * s < (1ull << 31);
}

extern "C" size_t cstp_bn_workspace_bytes(int32_t n, int32_t c, int32_t s, int32_t groups) {
  (void)s;
  if (n <= 0 || c <= 0 || groups <= 0 || (n % groups) != 0) return 0;
  const int npg = n / groups;
  // [c][groups][nsplit][2] fp64 partials, then [groups][c][2] fp32 per-group backward sums
  // ... then one absmax slot per wave of the apply pass (4 per block; <= n * c * ceil(s / 1024) blocks)
  return align_up((size_t)c * groups * bn_nsplit(npg, c) * 2 * sizeof(double), 256) +
         align_up((size_t)groups * c * 2 * sizeof(float), 256) + bn_slot_bytes(n, c, s);
}

extern "C" int cstp_bn_forward_train(void* stream, const float* x, const float* residual, float* y, const float* gamma,
                                     const float* beta, float* running_mean, float* running_var, float* save_mean,
                                     float* save_invstd, float* scale_shift, int32_t n, int32_t c, int32_t s,
                                     int32_t groups, float eps, float momentum, int32_t relu, void* ws, size_t ws_bytes) {
  return cstp_bn_forward_train_am(stream, x, residual, y, gamma, beta, running_mean, running_var, save_mean, save_invstd,
                                  scale_shift, n, c, s, groups, eps, momentum, relu, ws, ws_bytes, nullptr);
}

static int bn_forward_train_impl(void* stream, const float* x, const float* residual, float* y, const float* gamma,
                                 const float* beta, float* running_mean, float* running_var, float* save_mean,
                                 float* save_invstd, float* scale_shift, int32_t n, int32_t c, int32_t s, int32_t groups,
                                 float eps, float momentum, int32_t relu, void* ws, size_t ws_bytes, uint32_t* y_absmax,
                                 const double* pre_part, int32_t pre_nsplit);

extern "C" int cstp_bn_forward_train_am(void* stream, const float* x, const float* residual, float* y, const float* gamma,
                                        const float* beta, float* running_mean, float* running_var, float* save_mean,
                                        float* save_invstd, float* scale_shift, int32_t n, int32_t c, int32_t s,
                                        int32_t groups, float eps, float momentum, int32_t relu, void* ws, size_t ws_bytes,
                                        uint32_t* y_absmax) {
  return bn_forward_train_impl(stream, x, residual, y, gamma, beta, running_mean, running_var, save_mean, save_invstd,
                               scale_shift, n, c, s, groups, eps, momentum, relu, ws, ws_bytes, y_absmax, nullptr, 0);
}

extern "C" int cstp_bn_forward_train_pre(void* stream, const float* x, const float* residual, float* y, const float* gamma,
                                         const float* beta, float* running_mean, float* running_var, float* save_mean,
                                         float* save_invstd, float* scale_shift, int32_t n, int32_t c, int32_t s,
                                         int32_t groups, float eps, float momentum, int32_t relu, void* ws, size_t ws_bytes,
                                         uint32_t* y_absmax, const double* part, int32_t nsplit) {
  CSTP_REQUIRE(part != nullptr && nsplit > 0 && s > 1, "precomputed statistics: a partial-sum table from cstp_conv3d_forward_bnstats (BatchNorm3d only)");
  return bn_forward_train_impl(stream, x, residual, y, gamma, beta, running_mean, running_var, save_mean, save_invstd,
                               scale_shift, n, c, s, groups, eps, momentum, relu, ws, ws_bytes, y_absmax, part, nsplit);
}

static int bn_forward_train_impl(void* stream, const float* x, const float* residual, float* y, const float* gamma,
                                 const float* beta, float* running_mean, float* running_var, float* save_mean,
                                 float* save_invstd, float* scale_shift, int32_t n, int32_t c, int32_t s, int32_t groups,
                                 float eps, float momentum, int32_t relu, void* ws, size_t ws_bytes, uint32_t* y_absmax,
                                 const double* pre_part, int32_t pre_nsplit) {
  CSTP_REQUIRE(x && y && gamma && beta && save_mean && save_invstd, "null argument");
  CSTP_REQUIRE(n > 0 && c > 0 && s > 0 && groups > 0 && (n % groups) == 0, "bad shape");
  CSTP_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "running stats must come as a pair");
  const int npg = n / groups;
  CSTP_REQUIRE((size_t)npg * s > 1, "train-mode BatchNorm needs more than 1 value per channel");
  hipStream_t st = as_stream(stream);
  if (s == 1) {
    CSTP_REQUIRE(y_absmax == nullptr, "absmax by-product: BatchNorm3d (s > 1) only");
    hipLaunchKernelGGL(bn1d_fwd_kernel, dim3(cdiv(c, 64)), dim3(64), 0, st, x, residual, y, gamma, beta, running_mean,
                       running_var, save_mean, save_invstd, npg, groups, c, eps, momentum, relu);
    CSTP_LAUNCH_CHECK();
    return 0;
  }
  CSTP_REQUIRE(ws && ws_bytes >= cstp_bn_workspace_bytes(n, c, s, groups), "workspace too small");
  if (pre_part == nullptr && bn_small_ok(n, c, s, groups)) {
    unsigned* slots = y_absmax != nullptr ? bn_slots(ws, n, c, groups) : nullptr;
    if ((size_t)npg * s <= (size_t)BN_SMALL_PT * 256)
      hipLaunchKernelGGL(bn_small_fwd_kernel<256>, dim3(c), dim3(256), 0, st, x, residual, y, gamma, beta, running_mean, running_var,
                         save_mean, save_invstd, reinterpret_cast<float2*>(scale_shift), c, s, npg, groups, eps, momentum, relu, slots);
    else
      hipLaunchKernelGGL(bn_small_fwd_kernel<1024>, dim3(c), dim3(1024), 0, st, x, residual, y, gamma, beta, running_mean, running_var,
                         save_mean, save_invstd, reinterpret_cast<float2*>(scale_shift), c, s, npg, groups, eps, momentum, relu, slots);
    CSTP_LAUNCH_CHECK();
    if (slots != nullptr) {
      hipLaunchKernelGGL(absmax_store_kernel, dim3(1), dim3(256), 0, st, slots, c * 16, y_absmax);
      CSTP_LAUNCH_CHECK();
    }
    return 0;
  }
  double* part = reinterpret_cast<double*>(ws);
  const int ns = bn_nsplit(npg, c);
  const bool v4 = (s % 4) == 0;
  const dim3 rgrid(c, groups * ns);
  BnFin fin;
  memset(&fin, 0, sizeof(fin));
  if (pre_part != nullptr) {
    // the producing convolution left the sums (cstp_conv3d_forward_bnstats): no pass over x, a wave per channel folds them
    hipLaunchKernelGGL(bn_finalize_fwd_wide_kernel, dim3(c), dim3(64), 0, st, pre_part, save_mean, save_invstd, running_mean,
                       running_var, c, groups, pre_nsplit, (double)npg * s, eps, momentum, gamma, beta,
                       reinterpret_cast<float2*>(scale_shift), y_absmax, nullptr, 0);
  } else {
    // statistics pass; its finalize is folded into the apply pass below (BnFin), the absmax cell is zeroed here
    if (v4) hipLaunchKernelGGL((bn_reduce_kernel<0, true>), rgrid, dim3(256), 0, st, x, x, x, nullptr, nullptr, part, npg, c, s, ns, 0, nullptr, y_absmax);
    else hipLaunchKernelGGL((bn_reduce_kernel<0, false>), rgrid, dim3(256), 0, st, x, x, x, nullptr, nullptr, part, npg, c, s, ns, 0, nullptr, y_absmax);
    fin.part = part; fin.nsplit = ns; fin.groups = groups; fin.count = (double)npg * s; fin.eps = eps; fin.momentum = momentum;
    fin.running_mean = running_mean; fin.running_var = running_var; fin.save_mean = save_mean; fin.save_invstd = save_invstd;
    fin.ss = reinterpret_cast<float2*>(scale_shift);
  }
  CSTP_LAUNCH_CHECK();
  const int chunks = cdiv(s, BN_UNROLL * 256 * (v4 ? 4 : 1));
  const dim3 agrid((unsigned)((size_t)n * c * chunks));
  unsigned* slots = y_absmax != nullptr ? bn_slots(ws, n, c, groups) : nullptr;
  if (v4) hipLaunchKernelGGL((bn_apply_fwd_kernel<true>), agrid, dim3(256), 0, st, x, residual, y, gamma, beta, save_mean, save_invstd, c, s, npg, relu, chunks, slots, fin);
  else hipLaunchKernelGGL((bn_apply_fwd_kernel<false>), agrid, dim3(256), 0, st, x, residual, y, gamma, beta, save_mean, save_invstd, c, s, npg, relu, chunks, slots, fin);
  CSTP_LAUNCH_CHECK();
  if (slots != nullptr) {
    hipLaunchKernelGGL(absmax_fold_kernel, dim3(FOLD_BLOCKS), dim3(256), 0, st, slots, (int)agrid.x * 4, y_absmax);
    CSTP_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int cstp_bn_finalize_pre(void* stream, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                    float* save_mean, float* save_invstd, float* scale_shift, int32_t n, int32_t c, int32_t s,
                                    int32_t groups, float eps, float momentum, int32_t relu, const double* part, int32_t nsplit,
                                    uint32_t* z_cell) {
  CSTP_REQUIRE(gamma && beta && save_mean && save_invstd && scale_shift && part && z_cell, "null argument");
  CSTP_REQUIRE(n > 0 && c > 0 && s > 1 && groups > 0 && (n % groups) == 0 && nsplit > 0, "bad shape");
  CSTP_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "running stats must come as a pair");
  const int npg = n / groups;
  const unsigned* mm = reinterpret_cast<const unsigned*>(part + (size_t)c * groups * nsplit * 2 + c);
  hipLaunchKernelGGL(bn_finalize_fwd_wide_kernel, dim3(c), dim3(64), 0, as_stream(stream), part, save_mean, save_invstd,
                     running_mean, running_var, c, groups, nsplit, (double)npg * s, eps, momentum, gamma, beta,
                     reinterpret_cast<float2*>(scale_shift), z_cell, mm, relu);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t cstp_bn_eval_workspace_bytes(int32_t c) { return c > 0 ? align_up((size_t)c * sizeof(float), 256) : 0; }

extern "C" int cstp_bn_forward_eval(void* stream, const float* x, const float* residual, float* y, const float* gamma,
                                    const float* beta, const float* running_mean, const float* running_var, int32_t n,
                                    int32_t c, int32_t s, float eps, int32_t relu, void* ws, size_t ws_bytes) {
  return cstp_bn_forward_eval_am(stream, x, residual, y, gamma, beta, running_mean, running_var, n, c, s, eps, relu, ws, ws_bytes,
                                 nullptr);
}

extern "C" int cstp_bn_forward_eval_am(void* stream, const float* x, const float* residual, float* y, const float* gamma,
                                       const float* beta, const float* running_mean, const float* running_var, int32_t n,
                                       int32_t c, int32_t s, float eps, int32_t relu, void* ws, size_t ws_bytes,
                                       uint32_t* y_absmax) {
  CSTP_REQUIRE(x && y && gamma && beta && running_mean && running_var, "null argument");
  CSTP_REQUIRE(n > 0 && c > 0 && s > 0, "bad shape");
  hipStream_t st = as_stream(stream);
  if (s == 1) {
    hipLaunchKernelGGL(bn1d_eval_kernel, dim3(cdiv(c, 64)), dim3(64), 0, st, x, residual, y, gamma, beta, running_mean,
                       running_var, n, c, eps, relu);
    CSTP_LAUNCH_CHECK();
    return 0;
  }
  CSTP_REQUIRE(ws && ws_bytes >= cstp_bn_eval_workspace_bytes(c), "workspace too small");
  // with the absmax by-product the workspace is the train-mode one (cstp_bn_workspace_bytes(n, c, s, 1)): [invstd][wave slots]
  CSTP_REQUIRE(y_absmax == nullptr || ws_bytes >= cstp_bn_eval_workspace_bytes(c) + bn_slot_bytes(n, c, s), "workspace too small");
  float* invstd = reinterpret_cast<float*>(ws);
  unsigned* slots = y_absmax != nullptr ? reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + cstp_bn_eval_workspace_bytes(c)) : nullptr;
  hipLaunchKernelGGL(bn_eval_prepare_kernel, dim3(cdiv(c, 64)), dim3(64), 0, st, running_var, invstd, c, eps, y_absmax);
  CSTP_LAUNCH_CHECK();
  const bool v4 = (s % 4) == 0;
  const int chunks = cdiv(s, BN_UNROLL * 256 * (v4 ? 4 : 1));
  const dim3 agrid((unsigned)((size_t)n * c * chunks));
  // one "group" spanning the whole batch: the apply kernel reads mean/invstd at [channel]
  BnFin nofin;
  memset(&nofin, 0, sizeof(nofin));
  if (v4) hipLaunchKernelGGL((bn_apply_fwd_kernel<true>), agrid, dim3(256), 0, st, x, residual, y, gamma, beta, running_mean, invstd, c, s, n, relu, chunks, slots, nofin);
  else hipLaunchKernelGGL((bn_apply_fwd_kernel<false>), agrid, dim3(256), 0, st, x, residual, y, gamma, beta, running_mean, invstd, c, s, n, relu, chunks, slots, nofin);
  CSTP_LAUNCH_CHECK();
  if (slots != nullptr) {
    hipLaunchKernelGGL(absmax_fold_kernel, dim3(FOLD_BLOCKS), dim3(256), 0, st, slots, (int)agrid.x * 4, y_absmax);
    CSTP_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int cstp_bn_stats_train(void* stream, const float* x, const float* gamma, const float* beta, float* running_mean,
                                   float* running_var, float* save_mean, float* save_invstd, float* scale_shift, int32_t n,
                                   int32_t c, int32_t s, int32_t groups, float eps, float momentum, void* ws,
                                   size_t ws_bytes) {
  CSTP_REQUIRE(x && gamma && beta && save_mean && save_invstd && scale_shift, "null argument");
  CSTP_REQUIRE(n > 0 && c > 0 && s > 1 && groups > 0 && (n % groups) == 0, "bad shape");
  CSTP_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "running stats must come as a pair");
  CSTP_REQUIRE(ws && ws_bytes >= cstp_bn_workspace_bytes(n, c, s, groups), "workspace too small");
  hipStream_t st = as_stream(stream);
  const int npg = n / groups;
  double* part = reinterpret_cast<double*>(ws);
  const int ns = bn_nsplit(npg, c);
  const dim3 rgrid(c, groups * ns);
  if ((s % 4) == 0) hipLaunchKernelGGL((bn_reduce_kernel<0, true>), rgrid, dim3(256), 0, st, x, x, x, nullptr, nullptr, part, npg, c, s, ns, 0, nullptr);
  else hipLaunchKernelGGL((bn_reduce_kernel<0, false>), rgrid, dim3(256), 0, st, x, x, x, nullptr, nullptr, part, npg, c, s, ns, 0, nullptr);
  CSTP_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(cdiv(c, 64)), dim3(64), 0, st, part, save_mean, save_invstd, running_mean,
                     running_var, c, groups, ns, (double)npg * s, eps, momentum, gamma, beta,
                     reinterpret_cast<float2*>(scale_shift), nullptr);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_bn_backward(void* stream, const float* x, const float* y, const float* dy, const float* gamma,
                                const float* save_mean, const float* save_invstd, const float* scale_shift, float* dx,
                                float* dresidual, float* dgamma, float* dbeta, int32_t n, int32_t c, int32_t s,
                                int32_t groups, int32_t relu, void* ws, size_t ws_bytes) {
  return cstp_bn_backward_am(stream, x, y, dy, gamma, save_mean, save_invstd, scale_shift, dx, dresidual, dgamma, dbeta, n, c, s,
                             groups, relu, ws, ws_bytes, nullptr, 0);
}

extern "C" int cstp_bn_backward_am(void* stream, const float* x, const float* y, const float* dy, const float* gamma,
                                   const float* save_mean, const float* save_invstd, const float* scale_shift, float* dx,
                                   float* dresidual, float* dgamma, float* dbeta, int32_t n, int32_t c, int32_t s,
                                   int32_t groups, int32_t relu, void* ws, size_t ws_bytes, uint32_t* dx_absmax,
                                   int32_t accumulate) {
  CSTP_REQUIRE(x && dy && gamma && save_mean && save_invstd && dx && dgamma && dbeta, "null argument");
  CSTP_REQUIRE(y != nullptr || !relu || scale_shift != nullptr, "ReLU mask needs y or scale_shift");
  CSTP_REQUIRE(y != nullptr || s > 1, "BatchNorm1d backward needs y");
  const float2* ss2 = reinterpret_cast<const float2*>(scale_shift);
  CSTP_REQUIRE(n > 0 && c > 0 && s > 0 && groups > 0 && (n % groups) == 0, "bad shape");
  const int npg = n / groups;
  hipStream_t st = as_stream(stream);
  if (s == 1) {
    CSTP_REQUIRE(dx_absmax == nullptr, "absmax by-product: BatchNorm3d (s > 1) only");
    hipLaunchKernelGGL(bn1d_bwd_kernel, dim3(cdiv(c, 64)), dim3(64), 0, st, x, y, dy, gamma, save_mean, save_invstd, dx,
                       dresidual, dgamma, dbeta, npg, groups, c, relu, accumulate ? 1 : 0);
    CSTP_LAUNCH_CHECK();
    return 0;
  }
  CSTP_REQUIRE(ws && ws_bytes >= cstp_bn_workspace_bytes(n, c, s, groups), "workspace too small");
  // (the backward pass on 1024-thread blocks measured SLOWER than the three-launch sequence on the 14 x 14 stage -- 77 vs 56 us at
  //  576 channels: two tensors in registers per thread -- so only the 256-thread form is used here)
  if (bn_small_ok(n, c, s, groups) && (size_t)npg * s <= (size_t)BN_SMALL_PT * 256) {
    unsigned* slots = dx_absmax != nullptr ? bn_slots(ws, n, c, groups) : nullptr;
    hipLaunchKernelGGL(bn_small_bwd_kernel<256>, dim3(c), dim3(256), 0, st, x, y, dy, gamma, save_mean, save_invstd, ss2, dx, dresidual,
                       dgamma, dbeta, c, s, npg, groups, relu, accumulate ? 1 : 0, slots);
    CSTP_LAUNCH_CHECK();
    if (slots != nullptr) {
      hipLaunchKernelGGL(absmax_store_kernel, dim3(1), dim3(256), 0, st, slots, c * 16, dx_absmax);
      CSTP_LAUNCH_CHECK();
    }
    return 0;
  }
  double* part = reinterpret_cast<double*>(ws);
  const int ns = bn_nsplit(npg, c);
  float* gsum = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) +
                                         align_up((size_t)c * groups * ns * 2 * sizeof(double), 256));
  const bool v4 = (s % 4) == 0;
  const dim3 rgrid(c, groups * ns);
  // residual + ReLU (a block's last BatchNorm): the reduction pass leaves the masked gradient in dresidual, the apply pass reads it
  const bool g_first = relu && y != nullptr && dresidual != nullptr && dresidual != dy;
  float* gout = g_first ? dresidual : nullptr;
  if (v4) hipLaunchKernelGGL((bn_reduce_kernel<1, true>), rgrid, dim3(256), 0, st, x, y, dy, save_mean, save_invstd, part, npg, c, s, ns, relu, ss2, dx_absmax, gout);
  else hipLaunchKernelGGL((bn_reduce_kernel<1, false>), rgrid, dim3(256), 0, st, x, y, dy, save_mean, save_invstd, part, npg, c, s, ns, relu, ss2, dx_absmax, gout);
  CSTP_LAUNCH_CHECK();
  if (g_first) { y = nullptr; dy = dresidual; dresidual = nullptr; relu = 0; }
  // (the finalize -- per-group sums for the dx pass, dgamma / dbeta -- is folded into the apply pass: BnFin)
  BnFin fin;
  memset(&fin, 0, sizeof(fin));
  fin.part = part; fin.nsplit = ns; fin.groups = groups; fin.dgamma = dgamma; fin.dbeta = dbeta; fin.accumulate = accumulate ? 1 : 0;
  const float inv_count = (float)(1.0 / ((double)npg * s));
  const int chunks = cdiv(s, BN_UNROLL * 256 * (v4 ? 4 : 1));
  const dim3 agrid((unsigned)((size_t)n * c * chunks));
  unsigned* slots = dx_absmax != nullptr ? bn_slots(ws, n, c, groups) : nullptr;
  if (v4) hipLaunchKernelGGL((bn_apply_bwd_kernel<true>), agrid, dim3(256), 0, st, x, y, dy, gamma, save_mean, save_invstd, gsum, dx, dresidual, c, s, npg, inv_count, relu, ss2, chunks, slots, fin);
  else hipLaunchKernelGGL((bn_apply_bwd_kernel<false>), agrid, dim3(256), 0, st, x, y, dy, gamma, save_mean, save_invstd, gsum, dx, dresidual, c, s, npg, inv_count, relu, ss2, chunks, slots, fin);
  CSTP_LAUNCH_CHECK();
  if (slots != nullptr) {
    hipLaunchKernelGGL(absmax_fold_kernel, dim3(FOLD_BLOCKS), dim3(256), 0, st, slots, (int)agrid.x * 4, dx_absmax);
    CSTP_LAUNCH_CHECK();
  }
  return 0;
}
