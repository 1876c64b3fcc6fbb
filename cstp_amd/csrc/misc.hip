// Pooling, loss heads and the per-step flat-arena utilities (EMA, grad-norm clip, SGD).
// All HBM-streaming or tiny; fp32 data, fp64 only inside reductions.
#include "common.h"

namespace cstp {

thread_local char g_err[512] = {0};

// ---- AdaptiveAvgPool3d(1): one wave per (b, c) row ----------------------------------------------
__global__ void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int s) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* p = x + (size_t)row * s;
  float a = 0.f;
  for (int i = lane; i < s; i += 64) a += p[i];
  a = wave_sum(a);
  if (lane == 0) y[row] = a / (float)s;
}

__global__ void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int rows, int s) {
  const size_t total = (size_t)rows * s;
  const float inv = 1.f / (float)s;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) dx[i] = dy[i / s] * inv;
}

__global__ void channel_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int n, int c, int s) {
  if (s == 1) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    double a = 0.0;
    for (int r = 0; r < n; ++r) a += (double)x[(size_t)r * c + ch];
    out[ch] = (float)a;
  } else {
    __shared__ double sm[16];
    const int ch = blockIdx.x;
    double a = 0.0;
    for (int r = 0; r < n; ++r) {
      const float* p = x + ((size_t)r * c + ch) * s;
      for (int i = threadIdx.x; i < s; i += blockDim.x) a += (double)p[i];
    }
    a = block_sum(a, sm);
    if (threadIdx.x == 0) out[ch] = (float)a;
  }
}

// ---- MaxPool3d (models/BE/r3d_byol.py:158: kernel 3, stride 2, padding 1; any k/stride/pad here) -------------------
// forward: y = max over the window (padding = -inf), idx = flat index (d*H + h)*W + w of the FIRST maximum in (d, h, w)
// scan order (what aten's CPU kernel keeps: it only replaces on `>` or NaN); one thread per output element, lanes along w.
__global__ void maxpool3d_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int32_t* __restrict__ idx, int rows,
                                     int D, int H, int W, int Do, int Ho, int Wo, int kd, int kh, int kw, int sd, int sh,
                                     int sw, int pd, int ph, int pw) {
  const size_t total = (size_t)rows * Do * Ho * Wo;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho); r /= Ho;
    const int dd = (int)(r % Do); const size_t row = r / Do;
    const float* xp = x + row * (size_t)D * H * W;
    float best = -INFINITY;
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
          const float v = xp[fi];
          if (v > best || v != v || bi < 0) { best = v; bi = fi; }
        }
      }
    }
    y[i] = best;
    idx[i] = bi;
  }
}

// backward as a GATHER (deterministic, no atomics): dx[p] = sum of dy over the (at most ceil(k/s)^3) windows that cover p
// and whose recorded argmax is p.
__global__ void maxpool3d_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ idx, float* __restrict__ dx,
                                     int rows, int D, int H, int W, int Do, int Ho, int Wo, int kd, int kh, int kw, int sd,
                                     int sh, int sw, int pd, int ph, int pw) {
  const size_t total = (size_t)rows * D * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H); r /= H;
    const int d = (int)(r % D); const size_t row = r / D;
    const int fi = (d * H + h) * W + w;
    // output coordinates o with o*s - p <= c <= o*s - p + k - 1
    const int d_lo = (d + pd - kd + sd) > 0 ? (d + pd - kd + sd) / sd : 0, d_hi = (d + pd) / sd < Do - 1 ? (d + pd) / sd : Do - 1;
    const int h_lo = (h + ph - kh + sh) > 0 ? (h + ph - kh + sh) / sh : 0, h_hi = (h + ph) / sh < Ho - 1 ? (h + ph) / sh : Ho - 1;
    const int w_lo = (w + pw - kw + sw) > 0 ? (w + pw - kw + sw) / sw : 0, w_hi = (w + pw) / sw < Wo - 1 ? (w + pw) / sw : Wo - 1;
    const size_t obase = row * (size_t)Do * Ho * Wo;
    float acc = 0.f;
    for (int a = d_lo; a <= d_hi; ++a)
      for (int b = h_lo; b <= h_hi; ++b)
        for (int c = w_lo; c <= w_hi; ++c) {
          const size_t o = obase + ((size_t)a * Ho + b) * Wo + c;
          if (idx[o] == fi) acc += dy[o];
        }
    dx[i] = acc;
  }
}

// ---- BYOL loss: 2 - 2 <x/|x|, y/|y|>; one wave per row -------------------------------------------
__global__ void byol_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ loss, int b,
                                int f) {
  const int row = blockIdx.x, lane = threadIdx.x;
  const float* xp = x + (size_t)row * f;
  const float* yp = y + (size_t)row * f;
  float xx = 0.f, yy = 0.f, xy = 0.f;
  for (int i = lane; i < f; i += 64) { const float a = xp[i], c = yp[i]; xx += a * a; yy += c * c; xy += a * c; }
  xx = wave_sum(xx); yy = wave_sum(yy); xy = wave_sum(xy);
  if (lane == 0) {
    const float nx = fmaxf(sqrtf(xx), 1e-12f), ny = fmaxf(sqrtf(yy), 1e-12f);
    loss[row] = 2.f - 2.f * (xy / (nx * ny));
  }
}

__global__ void byol_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dloss,
                                float* __restrict__ dx, int b, int f) {
  const int row = blockIdx.x, lane = threadIdx.x;
  const float* xp = x + (size_t)row * f;
  const float* yp = y + (size_t)row * f;
  float xx = 0.f, yy = 0.f, xy = 0.f;
  for (int i = lane; i < f; i += 64) { const float a = xp[i], c = yp[i]; xx += a * a; yy += c * c; xy += a * c; }
  xx = wave_sum_all(xx); yy = wave_sum_all(yy); xy = wave_sum_all(xy);
  const float nx = fmaxf(sqrtf(xx), 1e-12f), ny = fmaxf(sqrtf(yy), 1e-12f);
  const float cosv = xy / (nx * ny);
  const float g = -2.f * dloss[row] / nx;
  for (int i = lane; i < f; i += 64) dx[(size_t)row * f + i] = g * (yp[i] / ny - cosv * xp[i] / nx);
}

// ---- F.normalize(x, p=2, dim=1) (r21d_byol.py:396): y = x / max(|x|, eps); one wave per row -------
__global__ void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ norm, int f,
                                  float eps) {
  const int row = blockIdx.x, lane = threadIdx.x;
  const float* xp = x + (size_t)row * f;
  float a = 0.f;
  for (int i = lane; i < f; i += 64) a += xp[i] * xp[i];
  a = wave_sum_all(a);
  const float nrm = fmaxf(sqrtf(a), eps);
  if (lane == 0) norm[row] = nrm;
  const float inv = 1.0f / nrm;
  for (int i = lane; i < f; i += 64) y[(size_t)row * f + i] = xp[i] * inv;
}

// dx = (dy - y <y, dy>) / max(|x|, eps)   (rows with |x| < eps: dx = dy / eps, as the clamp has zero slope)
__global__ void l2norm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ norm, const float* __restrict__ dy,
                                  float* __restrict__ dx, int f, float eps) {
  const int row = blockIdx.x, lane = threadIdx.x;
  const float* yp = y + (size_t)row * f;
  const float* gp = dy + (size_t)row * f;
  float d = 0.f;
  for (int i = lane; i < f; i += 64) d += yp[i] * gp[i];
  d = wave_sum_all(d);
  const float nrm = norm[row];
  if (!(nrm > eps)) d = 0.f;
  const float inv = 1.0f / nrm;
  for (int i = lane; i < f; i += 64) dx[(size_t)row * f + i] = (gp[i] - yp[i] * d) * inv;
}

// ---- CrossEntropyLoss(mean) ----------------------------------------------------------------------
__global__ void ce_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, float* __restrict__ loss,
                              int b, int k) {
  __shared__ double sm[16];
  double a = 0.0;
  for (int r = threadIdx.x; r < b; r += blockDim.x) {
    const float* p = logits + (size_t)r * k;
    float mx = p[0];
    for (int j = 1; j < k; ++j) mx = fmaxf(mx, p[j]);
    float se = 0.f;
    for (int j = 0; j < k; ++j) se += expf(p[j] - mx);
    a += (double)(logf(se) + mx - p[labels[r]]);
  }
  a = block_sum(a, sm);
  if (threadIdx.x == 0) loss[0] = (float)(a / b);
}

__global__ void ce_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                              const float* __restrict__ dloss, float* __restrict__ dlogits, int b, int k) {
  const float gscale = dloss[0] / (float)b;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < b; r += gridDim.x * blockDim.x) {
    const float* p = logits + (size_t)r * k;
    float mx = p[0];
    for (int j = 1; j < k; ++j) mx = fmaxf(mx, p[j]);
    float se = 0.f;
    for (int j = 0; j < k; ++j) se += expf(p[j] - mx);
    const int lab = (int)labels[r];
    for (int j = 0; j < k; ++j) dlogits[(size_t)r * k + j] = gscale * (expf(p[j] - mx) / se - (j == lab ? 1.f : 0.f));
  }
}

// ---- NT-Xent ---------------------------------------------------------------------------------------
// ws layout (floats): norms[2n] | lse[2n] | sim[2n*2n] | G[2n*2n]
__global__ void ntx_norm_kernel(const float* __restrict__ reps, float* __restrict__ norms, int two_n, int f) {
  const int row = blockIdx.x, lane = threadIdx.x;
  float a = 0.f;
  for (int i = lane; i < f; i += 64) { const float v = reps[(size_t)row * f + i]; a += v * v; }
  a = wave_sum(a);
  if (lane == 0) norms[row] = sqrtf(a);
}

// one block (256 threads = 4 waves) per row i; wave w handles columns j = w, w+4, ...
__global__ void ntx_sim_kernel(const float* __restrict__ reps, const float* __restrict__ norms, float* __restrict__ sim,
                               int two_n, int f) {
  const int i = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* ri = reps + (size_t)i * f;
  for (int j = wave; j < two_n; j += 4) {
    const float* rj = reps + (size_t)j * f;
    float a = 0.f;
    for (int q = lane; q < f; q += 64) a += ri[q] * rj[q];
    a = wave_sum(a);
    if (lane == 0) sim[(size_t)i * two_n + j] = a / fmaxf(norms[i] * norms[j], 1e-8f);
  }
}

__global__ void ntx_loss_kernel(const float* __restrict__ sim, float* __restrict__ lse, float* __restrict__ loss, int two_n,
                                float inv_t) {
  __shared__ double sm[16];
  const int n = two_n >> 1;
  double a = 0.0;
  for (int i = threadIdx.x; i < two_n; i += blockDim.x) {
    const float* p = sim + (size_t)i * two_n;
    float mx = -3.0e38f;
    for (int j = 0; j < two_n; ++j) if (j != i) mx = fmaxf(mx, p[j] * inv_t);
    float se = 0.f;
    for (int j = 0; j < two_n; ++j) if (j != i) se += expf(p[j] * inv_t - mx);
    const float l = logf(se) + mx;
    lse[i] = l;
    const int pos = (i + n) % two_n;
    a += (double)(l - p[pos] * inv_t);
  }
  a = block_sum(a, sm);
  if (threadIdx.x == 0) loss[0] = (float)(a / two_n);
}

__global__ void ntx_g_kernel(const float* __restrict__ sim, const float* __restrict__ lse, const float* __restrict__ dloss,
                             float* __restrict__ G, int two_n, float inv_t) {
  const int n = two_n >> 1;
  const float sc = dloss[0] * inv_t / (float)two_n;
  const int total = two_n * two_n;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int i = e / two_n, j = e - i * two_n;
    float g = 0.f;
    if (j != i) {
      g = expf(sim[e] * inv_t - lse[i]);
      if (j == (i + n) % two_n) g -= 1.f;
      g *= sc;
    }
    G[e] = g;
  }
}

// one block per row i: d_hat = sum_j (G_ij + G_ji) rhat_j ; dreps_i = (d_hat - rhat_i <rhat_i, d_hat>) / |r_i|
__global__ void ntx_bwd_kernel(const float* __restrict__ reps, const float* __restrict__ norms, const float* __restrict__ G,
                               float* __restrict__ dreps, int two_n, int f) {
  __shared__ float sm[16];
  const int i = blockIdx.x;
  const float ni = fmaxf(norms[i], 1e-8f);
  float dot = 0.f;
  // each thread owns features q = tid, tid+256, ... (f <= 256*8 supported)
  float dh[8];
  int cnt = 0;
  for (int q = threadIdx.x; q < f && cnt < 8; q += blockDim.x, ++cnt) {
    float a = 0.f;
    for (int j = 0; j < two_n; ++j) {
      const float w = G[(size_t)i * two_n + j] + G[(size_t)j * two_n + i];
      a += w * reps[(size_t)j * f + q] / fmaxf(norms[j], 1e-8f);
    }
    dh[cnt] = a;
    dot += a * reps[(size_t)i * f + q] / ni;
  }
  dot = block_sum(dot, sm);
  __shared__ float bdot;
  if (threadIdx.x == 0) bdot = dot;
  __syncthreads();
  dot = bdot;
  cnt = 0;
  for (int q = threadIdx.x; q < f && cnt < 8; q += blockDim.x, ++cnt)
    dreps[(size_t)i * f + q] = (dh[cnt] - reps[(size_t)i * f + q] / ni * dot) / ni;
}

// ---- flat-arena utilities ---------------------------------------------------------------------------
__global__ void ema_kernel(float* __restrict__ t, const float* __restrict__ o, size_t n, float m, float om) {
  const size_t n4 = n >> 2;
  float4* t4 = reinterpret_cast<float4*>(t);
  const float4* o4 = reinterpret_cast<const float4*>(o);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 a = t4[i];
    const float4 b = o4[i];
    a.x = a.x * m + b.x * om; a.y = a.y * m + b.y * om; a.z = a.z * m + b.z * om; a.w = a.w * m + b.w * om;
    t4[i] = a;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = (n4 << 2) + threadIdx.x;
    t[i] = t[i] * m + o[i] * om;
  }
}

__global__ void sumsq_partial_kernel(const float* __restrict__ g, size_t n, double* __restrict__ part) {
  __shared__ double sm[16];
  double a = 0.0;
  const size_t n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = g4[i];
    a += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; a += (double)v * v; }
  a = block_sum(a, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = a;
}

__global__ void sumsq_final_kernel(const double* __restrict__ part, int nb, float* __restrict__ out) {
  __shared__ double sm[16];
  double a = 0.0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) a += part[i];
  a = block_sum(a, sm);
  if (threadIdx.x == 0) out[0] = (float)a;
}

__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float max_norm, float* __restrict__ coef,
                                 float* __restrict__ norm_out) {
  const float nrm = sqrtf(sumsq[0]);
  const float c = max_norm / (nrm + 1e-6f);
  coef[0] = c < 1.f ? c : 1.f;
  if (norm_out != nullptr) norm_out[0] = nrm;
}

__global__ void sgd_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ buf, size_t n,
                           const float* __restrict__ lr_p, float momentum, float wd, const float* __restrict__ coef_p,
                           int first_step, int write_back) {
  const float lr = lr_p[0];
  const float coef = coef_p != nullptr ? coef_p[0] : 1.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float gr = g[i] * coef;
    if (write_back) g[i] = gr;
    const float pv = p[i];
    gr = gr + wd * pv;
    const float b = first_step ? gr : momentum * buf[i] + gr;
    buf[i] = b;
    p[i] = pv - lr * b;
  }
}

// torch.optim.Adam / AdamW (no amsgrad), bias corrections passed in as 1 - beta^t
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            size_t n, const float* __restrict__ lr_p, float b1, float b2, float eps, float wd, int decoupled,
                            float bc1, float bc2_sqrt) {
  const float lr = lr_p[0];
  const float step_size = lr / bc1;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float pv = p[i], gr = g[i];
    if (decoupled) pv *= (1.f - lr * wd);
    else gr += wd * pv;
    const float mi = b1 * m[i] + (1.f - b1) * gr;
    const float vi = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mi;
    v[i] = vi;
    p[i] = pv - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
  }
}

static inline int stream_grid(size_t n) {
  size_t b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace cstp

using namespace cstp;

extern "C" int cstp_abi_version(void) { return CSTP_ABI_VERSION; }
extern "C" const char* cstp_last_error(void) { return g_err; }

extern "C" int cstp_avgpool_forward(void* stream, const float* x, float* y, int32_t rows, int32_t s) {
  CSTP_REQUIRE(x && y && rows > 0 && s > 0, "bad argument");
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), x, y, rows, s);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_avgpool_backward(void* stream, const float* dy, float* dx, int32_t rows, int32_t s) {
  CSTP_REQUIRE(dy && dx && rows > 0 && s > 0, "bad argument");
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(stream_grid((size_t)rows * s)), dim3(256), 0, as_stream(stream), dy, dx, rows, s);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_channel_sum(void* stream, const float* x, float* out, int32_t n, int32_t c, int32_t s, void* ws,
                                size_t ws_bytes) {
  (void)ws; (void)ws_bytes;
  CSTP_REQUIRE(x && out && n > 0 && c > 0 && s > 0, "bad argument");
  if (s == 1) hipLaunchKernelGGL(channel_sum_kernel, dim3(cdiv(c, 64)), dim3(64), 0, as_stream(stream), x, out, n, c, s);
  else hipLaunchKernelGGL(channel_sum_kernel, dim3(c), dim3(256), 0, as_stream(stream), x, out, n, c, s);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_byol_loss_forward(void* stream, const float* x, const float* y, float* loss, int32_t b, int32_t f) {
  CSTP_REQUIRE(x && y && loss && b > 0 && f > 0, "bad argument");
  hipLaunchKernelGGL(byol_fwd_kernel, dim3(b), dim3(64), 0, as_stream(stream), x, y, loss, b, f);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_byol_loss_backward(void* stream, const float* x, const float* y, const float* dloss, float* dx,
                                       int32_t b, int32_t f) {
  CSTP_REQUIRE(x && y && dloss && dx && b > 0 && f > 0, "bad argument");
  hipLaunchKernelGGL(byol_bwd_kernel, dim3(b), dim3(64), 0, as_stream(stream), x, y, dloss, dx, b, f);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_l2_normalize_forward(void* stream, const float* x, float* y, float* norm, int32_t rows, int32_t f,
                                         float eps) {
  CSTP_REQUIRE(x && y && norm && rows > 0 && f > 0, "bad argument");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(rows), dim3(64), 0, as_stream(stream), x, y, norm, f, eps);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_l2_normalize_backward(void* stream, const float* y, const float* norm, const float* dy, float* dx,
                                          int32_t rows, int32_t f, float eps) {
  CSTP_REQUIRE(y && norm && dy && dx && rows > 0 && f > 0, "bad argument");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(rows), dim3(64), 0, as_stream(stream), y, norm, dy, dx, f, eps);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_cross_entropy_forward(void* stream, const float* logits, const int64_t* labels, float* loss,
                                          int32_t b, int32_t k) {
  CSTP_REQUIRE(logits && labels && loss && b > 0 && k > 0, "bad argument");
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(1), dim3(256), 0, as_stream(stream), logits, labels, loss, b, k);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_cross_entropy_backward(void* stream, const float* logits, const int64_t* labels, const float* dloss,
                                           float* dlogits, int32_t b, int32_t k) {
  CSTP_REQUIRE(logits && labels && dloss && dlogits && b > 0 && k > 0, "bad argument");
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(cdiv(b, 256)), dim3(256), 0, as_stream(stream), logits, labels, dloss, dlogits, b, k);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t cstp_ntxent_workspace_bytes(int32_t two_n, int32_t f) {
  (void)f;
  if (two_n <= 0) return 0;
  return align_up(((size_t)2 * two_n + (size_t)2 * two_n * two_n) * sizeof(float), 256);
}

extern "C" int cstp_ntxent_forward(void* stream, const float* reps, float* loss, int32_t two_n, int32_t f,
                                   float temperature, void* ws, size_t ws_bytes) {
  CSTP_REQUIRE(reps && loss && ws, "null argument");
  CSTP_REQUIRE(two_n >= 4 && (two_n % 2) == 0 && f > 0 && temperature > 0.f, "bad shape");
  CSTP_REQUIRE(ws_bytes >= cstp_ntxent_workspace_bytes(two_n, f), "workspace too small");
  hipStream_t s = as_stream(stream);
  float* norms = reinterpret_cast<float*>(ws);
  float* lse = norms + two_n;
  float* sim = lse + two_n;
  hipLaunchKernelGGL(ntx_norm_kernel, dim3(two_n), dim3(64), 0, s, reps, norms, two_n, f);
  hipLaunchKernelGGL(ntx_sim_kernel, dim3(two_n), dim3(256), 0, s, reps, norms, sim, two_n, f);
  hipLaunchKernelGGL(ntx_loss_kernel, dim3(1), dim3(256), 0, s, sim, lse, loss, two_n, 1.f / temperature);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_ntxent_backward(void* stream, const float* reps, const float* dloss, float* dreps, int32_t two_n,
                                    int32_t f, float temperature, void* ws, size_t ws_bytes) {
  CSTP_REQUIRE(reps && dloss && dreps && ws, "null argument");
  CSTP_REQUIRE(two_n >= 4 && (two_n % 2) == 0 && f > 0 && f <= 2048 && temperature > 0.f, "bad shape");
  CSTP_REQUIRE(ws_bytes >= cstp_ntxent_workspace_bytes(two_n, f), "workspace too small");
  hipStream_t s = as_stream(stream);
  float* norms = reinterpret_cast<float*>(ws);
  float* lse = norms + two_n;
  float* sim = lse + two_n;
  float* G = sim + (size_t)two_n * two_n;
  hipLaunchKernelGGL(ntx_g_kernel, dim3(cdiv(two_n * two_n, 256)), dim3(256), 0, s, sim, lse, dloss, G, two_n, 1.f / temperature);
  hipLaunchKernelGGL(ntx_bwd_kernel, dim3(two_n), dim3(256), 0, s, reps, norms, G, dreps, two_n, f);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_ema_update(void* stream, float* target, const float* online, size_t n, double m) {
  CSTP_REQUIRE(target && online && n > 0, "bad argument");
  CSTP_REQUIRE((reinterpret_cast<uintptr_t>(target) & 15) == 0 && (reinterpret_cast<uintptr_t>(online) & 15) == 0,
               "arenas must be 16-byte aligned");
  hipLaunchKernelGGL(ema_kernel, dim3(stream_grid(n / 4 + 1)), dim3(256), 0, as_stream(stream), target, online, n, (float)m,
                     (float)(1.0 - m));
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_sumsq(void* stream, const float* g, size_t n, float* out, void* ws, size_t ws_bytes) {
  CSTP_REQUIRE(g && out && ws && n > 0, "bad argument");
  CSTP_REQUIRE(ws_bytes >= 8192, "workspace too small (need 8 KiB)");
  CSTP_REQUIRE((reinterpret_cast<uintptr_t>(g) & 15) == 0, "arena must be 16-byte aligned");
  int nb = stream_grid(n / 4 + 1);
  if (nb > 1024) nb = 1024;
  double* part = reinterpret_cast<double*>(ws);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, as_stream(stream), g, n, part);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), part, nb, out);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_clip_coef(void* stream, const float* sumsq, float max_norm, float* coef, float* norm_out) {
  CSTP_REQUIRE(sumsq && coef, "null argument");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, as_stream(stream), sumsq, max_norm, coef, norm_out);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_sgd_step(void* stream, float* p, float* g, float* buf, size_t n, const float* lr, float momentum,
                             float weight_decay, const float* coef, int32_t first_step, int32_t write_back_grad) {
  CSTP_REQUIRE(p && g && buf && lr && n > 0, "bad argument");
  hipLaunchKernelGGL(sgd_kernel, dim3(stream_grid(n)), dim3(256), 0, as_stream(stream), p, g, buf, n, lr, momentum,
                     weight_decay, coef, first_step, write_back_grad);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_adam_step(void* stream, float* p, const float* g, float* exp_avg, float* exp_avg_sq, size_t n,
                              const float* lr, float beta1, float beta2, float eps, float weight_decay, int32_t decoupled,
                              int32_t step) {
  CSTP_REQUIRE(p && g && exp_avg && exp_avg_sq && lr && n > 0 && step > 0, "bad argument");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(stream_grid(n)), dim3(256), 0, as_stream(stream), p, g, exp_avg, exp_avg_sq, n, lr,
                     beta1, beta2, eps, weight_decay, decoupled, (float)bc1, (float)sqrt(bc2));
  CSTP_LAUNCH_CHECK();
  return 0;
}

static inline bool pool_dims(int D, int H, int W, const int32_t* k, const int32_t* st, const int32_t* pd, int& Do, int& Ho, int& Wo) {
  if (D <= 0 || H <= 0 || W <= 0) return false;
  for (int i = 0; i < 3; ++i) if (k[i] <= 0 || st[i] <= 0 || pd[i] < 0 || 2 * pd[i] > k[i]) return false;
  Do = (D + 2 * pd[0] - k[0]) / st[0] + 1;
  Ho = (H + 2 * pd[1] - k[1]) / st[1] + 1;
  Wo = (W + 2 * pd[2] - k[2]) / st[2] + 1;
  return Do > 0 && Ho > 0 && Wo > 0;
}

extern "C" int cstp_maxpool3d_forward(void* stream, const float* x, float* y, int32_t* argmax, int32_t rows, int32_t d, int32_t h,
                                      int32_t w, const int32_t* kernel3, const int32_t* stride3, const int32_t* pad3) {
  CSTP_REQUIRE(x && y && argmax && kernel3 && stride3 && pad3 && rows > 0, "bad argument");
  int Do, Ho, Wo;
  CSTP_REQUIRE(pool_dims(d, h, w, kernel3, stride3, pad3, Do, Ho, Wo), "bad pooling geometry");
  CSTP_REQUIRE((size_t)d * h * w < (1ull << 31), "plane too large for int32 argmax");
  const size_t total = (size_t)rows * Do * Ho * Wo;
  hipLaunchKernelGGL(maxpool3d_fwd_kernel, dim3(stream_grid(total) * 8), dim3(256), 0, as_stream(stream), x, y, argmax, rows, d, h, w,
                     Do, Ho, Wo, kernel3[0], kernel3[1], kernel3[2], stride3[0], stride3[1], stride3[2], pad3[0], pad3[1], pad3[2]);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_maxpool3d_backward(void* stream, const float* dy, const int32_t* argmax, float* dx, int32_t rows, int32_t d,
                                       int32_t h, int32_t w, const int32_t* kernel3, const int32_t* stride3, const int32_t* pad3) {
  CSTP_REQUIRE(dy && argmax && dx && kernel3 && stride3 && pad3 && rows > 0, "bad argument");
  int Do, Ho, Wo;
  CSTP_REQUIRE(pool_dims(d, h, w, kernel3, stride3, pad3, Do, Ho, Wo), "bad pooling geometry");
  const size_t total = (size_t)rows * d * h * w;
  hipLaunchKernelGGL(maxpool3d_bwd_kernel, dim3(stream_grid(total) * 8), dim3(256), 0, as_stream(stream), dy, argmax, dx, rows, d, h, w,
                     Do, Ho, Wo, kernel3[0], kernel3[1], kernel3[2], stride3[0], stride3[1], stride3[2], pad3[0], pad3[1], pad3[2]);
  CSTP_LAUNCH_CHECK();
  return 0;
}
