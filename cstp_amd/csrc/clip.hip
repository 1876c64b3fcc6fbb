// GPU clip assembly for the CSTP data path: decoded uint8 video frames resident in HBM -> normalised fp32 clip tensors.
// Per frame of a clip (reference: data_process/datasets.py:876-948 + preprocess_data.py:479-581,1103-1110, the `null_transform`
// path):  Image.transpose(ROTATE_*) -> Image.crop(box) -> Image.resize((S, S), Image.BICUBIC) -> [FLIP_LEFT_RIGHT] ->
// ToTensor -> x * 2 - 1 clamped to [-1, 1] -> stacked as [3][T][S][S].
// The resize is Pillow's algorithm (libImaging/Resample.c) bit for bit: separable, horizontal pass first with a uint8
// intermediate, 22-bit fixed-point coefficients (computed on the host exactly as precompute_coeffs / normalize_coeffs_8bpc do
// and passed in), accumulator seeded with 1 << 21, arithmetic shift, clamp to 0..255.  Rotation and crop are folded into the
// source indexing of the horizontal pass (a box may reach past the rotated frame -- the second crop of a pair whose clips are
// rotated differently does, preprocess_data.py:535-541 checks only two of its four sides -- and reads zeros there, as
// Image.crop does), the flip and the normalisation into the store of the vertical pass, so a frame is read
// once and the clip is written once; both kernels are HBM/latency-trivial next to a training step (a 16-frame 112x112 clip from
// 240x320 frames: 3.7 MB in, 2.4 MB out).
#include "common.h"

namespace cstp {

constexpr int CLIP_PRECISION_BITS = 22;

__device__ __forceinline__ int clip8_fixed(int acc) {
  const int v = acc >> CLIP_PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// tmp[t][y][xx][c] = clip8(1 << 21 + sum_i src(t, y + row_first, x0(xx) + i, c) * kh[xx][i])   over the cropped image
__global__ void clip_resize_h_kernel(const uint8_t* __restrict__ frames, const int32_t* __restrict__ frame_idx,
                                     uint8_t* __restrict__ tmp, int F, int H, int W, int T, int rot, int bx0, int by0,
                                     int size, const int32_t* __restrict__ kh, const int32_t* __restrict__ bh, int ksh,
                                     int row_first, int rows) {
  const size_t total = (size_t)T * rows * size;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % size);
    const size_t r = i / size;
    const int y = (int)(r % rows), t = (int)(r / rows);
    int f = frame_idx[t];
    f = f < 0 ? 0 : (f >= F ? F - 1 : f);
    const uint8_t* fp = frames + (size_t)f * H * W * 3;
    const int x0 = bh[2 * xx], n = bh[2 * xx + 1];
    const int ry = by0 + y + row_first;                  // row in the ROTATED frame
    int a0 = 1 << (CLIP_PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int j = 0; j < n; ++j) {
      const int rx = bx0 + x0 + j;                        // column in the rotated frame
      int ox, oy;                                         // the same pixel in the stored frame (inverse of Image.transpose)
      if (rot == 90) { ox = W - 1 - ry; oy = rx; }
      else if (rot == 180) { ox = W - 1 - rx; oy = H - 1 - ry; }
      else if (rot == 270) { ox = ry; oy = H - 1 - rx; }
      else { ox = rx; oy = ry; }
      if (ox < 0 || ox >= W || oy < 0 || oy >= H) continue;   // Image.crop beyond the frame reads black (0, 0, 0)
      const uint8_t* p = fp + ((size_t)oy * W + ox) * 3;
      const int k = kh[xx * ksh + j];
      a0 += (int)p[0] * k; a1 += (int)p[1] * k; a2 += (int)p[2] * k;
    }
    uint8_t* o = tmp + i * 3;
    o[0] = (uint8_t)clip8_fixed(a0); o[1] = (uint8_t)clip8_fixed(a1); o[2] = (uint8_t)clip8_fixed(a2);
  }
}

// out[c][t][yy][flip ? S-1-xx : xx] = clamp(clip8(1 << 21 + sum_i tmp[t][y0(yy) - row_first + i][xx][c] * kv[yy][i]) / 255 * 2 - 1)
// U8: the resized frames stay 8-bit RGB [t][yy][xx][3] (no flip, no normalisation) for the base_transform operations below
template <bool U8>
__global__ void clip_resize_v_kernel(const uint8_t* __restrict__ tmp, float* __restrict__ out, uint8_t* __restrict__ out8, int T,
                                     int size, int flip, const int32_t* __restrict__ kv, const int32_t* __restrict__ bv, int ksv,
                                     int row_first, int rows) {
  const size_t total = (size_t)T * size * size;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % size);
    const size_t r = i / size;
    const int yy = (int)(r % size), t = (int)(r / size);
    const int y0 = bv[2 * yy] - row_first, n = bv[2 * yy + 1];
    int a0 = 1 << (CLIP_PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int j = 0; j < n; ++j) {
      const uint8_t* p = tmp + (((size_t)t * rows + (y0 + j)) * size + xx) * 3;
      const int k = kv[yy * ksv + j];
      a0 += (int)p[0] * k; a1 += (int)p[1] * k; a2 += (int)p[2] * k;
    }
    if constexpr (U8) {
      uint8_t* o8 = out8 + i * 3;
      o8[0] = (uint8_t)clip8_fixed(a0); o8[1] = (uint8_t)clip8_fixed(a1); o8[2] = (uint8_t)clip8_fixed(a2);
      continue;
    }
    const int xo = flip ? size - 1 - xx : xx;
    const size_t plane = (size_t)T * size * size;
    const size_t o = ((size_t)t * size + yy) * size + xo;
    const int c8[3] = {clip8_fixed(a0), clip8_fixed(a1), clip8_fixed(a2)};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = (float)c8[c] / 255.0f;                   // transforms.ToTensor()
      v = v * 2.0f - 1.0f;                               // ClipNormalize('tf')
      out[c * plane + o] = fminf(fmaxf(v, -1.0f), 1.0f);
    }
  }
}

// ---- the base_transform branch (preprocess_data.py:1110-1121) on 8-bit RGB clips [t][h][w][3]; every kernel reproduces the
//      Pillow call the reference makes bit for bit (oracle/pil_ops.py restates them in numpy and is pinned against Pillow) ----

// Image.rotate(angle), NEAREST, same size, black fill: Geometry.c affine_fixed, 16.16 fixed point (coefficients from the host)
__global__ void clip_rotate_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int T, int H, int W, int a0, int a1,
                                   int a2, int a3, int a4, int a5) {
  const size_t total = (size_t)T * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const size_t r = i / W;
    const int y = (int)(r % H), t = (int)(r / H);
    const int xin = (a2 + a1 * y + a0 * x) >> 16, yin = (a5 + a4 * y + a3 * x) >> 16;
    uint8_t v0 = 0, v1 = 0, v2 = 0;
    if (xin >= 0 && xin < W && yin >= 0 && yin < H) {
      const uint8_t* p = src + (((size_t)t * H + yin) * W + xin) * 3;
      v0 = p[0]; v1 = p[1]; v2 = p[2];
    }
    uint8_t* o = dst + i * 3;
    o[0] = v0; o[1] = v1; o[2] = v2;
  }
}

// Image.convert('L') of RGB (Convert.c rgb2l): ITU-R 601-2 luma in 16-bit fixed point
__device__ __forceinline__ int luma8(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// per-frame mean of the luma for ImageEnhance.Contrast: int(sum / count + 0.5) (ImageStat.Stat(...).mean[0] is a double)
__global__ void __launch_bounds__(256) clip_luma_mean_kernel(const uint8_t* __restrict__ src, int32_t* __restrict__ mean, int HW) {
  __shared__ unsigned long long red[4];
  const uint8_t* p = src + (size_t)blockIdx.x * HW * 3;
  unsigned long long acc = 0;
  for (int i = threadIdx.x; i < HW; i += 256) acc += (unsigned long long)luma8(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) mean[blockIdx.x] = (int32_t)((double)(red[0] + red[1] + red[2] + red[3]) / (double)HW + 0.5);
}

// Image.blend(degenerate, image, alpha) (Blend.c; single precision, NO fused multiply-add): interpolation truncates,
// extrapolation clips.  mode 0: degenerate = black (Brightness); 1: the frame's mean luma (Contrast); 2: the pixel's luma (Color)
// (hipcc contracts a * b + c into a fused multiply-add by default and its __fmul_rn / __fadd_rn are plain operators: the
//  functions that must round like the C code they reproduce switch the contraction off AND pass every product that feeds an
//  addition through an opaque register move, which no later pass can fuse across)
__device__ __forceinline__ float rounded(float x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ uint8_t blend8(int deg, int v, float alpha, bool inside) {
#pragma clang fp contract(off)
  const float t = __fadd_rn((float)deg, rounded(__fmul_rn(alpha, (float)(v - deg))));
  if (inside) return (uint8_t)(int)t;
  return t <= 0.0f ? 0 : (t >= 255.0f ? 255 : (uint8_t)(int)t);
}
__global__ void clip_blend_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int T, int HW, int mode, float alpha,
                                  const int32_t* __restrict__ mean) {
  const bool inside = alpha >= 0.0f && alpha <= 1.0f;
  const size_t total = (size_t)T * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const uint8_t* p = src + i * 3;
    const int r = p[0], g = p[1], b = p[2];
    const int deg = mode == 0 ? 0 : (mode == 1 ? mean[i / HW] : luma8(r, g, b));
    uint8_t* o = dst + i * 3;
    o[0] = blend8(deg, r, alpha, inside); o[1] = blend8(deg, g, alpha, inside); o[2] = blend8(deg, b, alpha, inside);
  }
}

// torchvision adjust_hue on PIL images: RGB -> HSV (Convert.c rgb2hsv_row), h += shift modulo 256, HSV -> RGB (hsv2rgb).  The
// float / double mix of the C statements is reproduced operation by operation (oracle/pil_ops.py: exact on all 2^24 colours).
__device__ __forceinline__ void rgb2hsv8(int r, int g, int b, int& uh, int& us, int& uv) {
#pragma clang fp contract(off)
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  uv = maxc;
  if (minc == maxc) { uh = 0; us = 0; return; }
  const float cr = (float)(maxc - minc);
  const float s = __fdiv_rn(cr, (float)maxc);
  const float rc = __fdiv_rn((float)(maxc - r), cr), gc = __fdiv_rn((float)(maxc - g), cr), bc = __fdiv_rn((float)(maxc - b), cr);
  float h;
  if (r == maxc) h = __fsub_rn(bc, gc);
  else if (g == maxc) h = (float)__dsub_rn(__dadd_rn(2.0, (double)rc), (double)bc);
  else h = (float)__dsub_rn(__dadd_rn(4.0, (double)gc), (double)rc);
  h = (float)fmod(__dadd_rn(__ddiv_rn((double)h, 6.0), 1.0), 1.0);
  const int ih = (int)__dmul_rn((double)h, 255.0), is = (int)__dmul_rn((double)s, 255.0);
  uh = ih < 0 ? 0 : (ih > 255 ? 255 : ih);
  us = is < 0 ? 0 : (is > 255 ? 255 : is);
}
__device__ __forceinline__ int round_half_away(float x) {      // C round() of a non-negative value, clipped to 0..255
  const int v = (int)floor((double)x + 0.5);
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}
__device__ __forceinline__ void hsv2rgb8(int h, int s, int v, int& r, int& g, int& b) {
#pragma clang fp contract(off)
  if (s == 0) { r = g = b = v; return; }
  const float hf = __fdiv_rn(__fmul_rn((float)h, 6.0f), 255.0f);
  const int i = (int)floorf(hf);
  const float f = __fsub_rn(hf, (float)i);
  const float fs = __fdiv_rn((float)s, 255.0f), vf = (float)v;
  const int p = round_half_away(__fmul_rn(vf, __fsub_rn(1.0f, fs)));
  const int q = round_half_away(__fmul_rn(vf, __fsub_rn(1.0f, rounded(__fmul_rn(fs, f)))));
  const int t = round_half_away(__fmul_rn(vf, __fsub_rn(1.0f, rounded(__fmul_rn(fs, __fsub_rn(1.0f, f))))));
  switch (i % 6) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}
// mode 0: hue shift (RGB -> RGB); 1: RGB -> HSV; 2: HSV -> RGB (the two conversions alone, for the exhaustive tests)
__global__ void clip_hue_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t npix, int shift, int mode) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
    const uint8_t* p = src + i * 3;
    int a = p[0], b = p[1], c = p[2], h, s, v, r, g, bb;
    if (mode == 2) { h = a; s = b; v = c; }
    else rgb2hsv8(a, b, c, h, s, v);
    if (mode == 1) { r = h; g = s; bb = v; }
    else { h = (h + shift) & 255; hsv2rgb8(h, s, v, r, g, bb); }
    uint8_t* o = dst + i * 3;
    o[0] = (uint8_t)r; o[1] = (uint8_t)g; o[2] = (uint8_t)bb;
  }
}

// ClipRandomGray.grayscale: frame t keeps channel ch[t] in all three channels (ch[t] < 0: the frame is copied)
__global__ void clip_gray_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int T, int HW,
                                 const int32_t* __restrict__ ch) {
  const size_t total = (size_t)T * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = ch[i / HW];
    const uint8_t* p = src + i * 3;
    uint8_t* o = dst + i * 3;
    if (c < 0) { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; }
    else { const uint8_t v = p[c]; o[0] = v; o[1] = v; o[2] = v; }
  }
}

// one pass of BoxBlur.c ImagingLineBoxBlur8 along x (vertical == 0) or y: box of 2 r + 1 pixels of weight ww plus the two next
// pixels of weight fw, 24-bit fixed point, edge pixels repeated beyond the line's ends
__global__ void clip_box_blur_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int T, int H, int W, int vertical,
                                     int radius, unsigned ww, unsigned fw) {
  const size_t total = (size_t)T * H * W;
  const int n = vertical ? H : W;
  const size_t step = vertical ? (size_t)W * 3 : 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const size_t rr = i / W;
    const int y = (int)(rr % H);
    const int pos = vertical ? y : x;
    const uint8_t* line = src + i * 3 - (size_t)pos * step;        // element 0 of my line
    unsigned s0 = 0, s1 = 0, s2 = 0;
    for (int k = pos - radius; k <= pos + radius; ++k) {
      const uint8_t* p = line + (size_t)(k < 0 ? 0 : (k > n - 1 ? n - 1 : k)) * step;
      s0 += p[0]; s1 += p[1]; s2 += p[2];
    }
    const int kl = pos - radius - 1, kr = pos + radius + 1;
    const uint8_t* pl = line + (size_t)(kl < 0 ? 0 : kl) * step;
    const uint8_t* pr = line + (size_t)(kr > n - 1 ? n - 1 : kr) * step;
    uint8_t* o = dst + i * 3;
    o[0] = (uint8_t)((s0 * ww + ((unsigned)pl[0] + pr[0]) * fw + (1u << 23)) >> 24);
    o[1] = (uint8_t)((s1 * ww + ((unsigned)pl[1] + pr[1]) * fw + (1u << 23)) >> 24);
    o[2] = (uint8_t)((s2 * ww + ((unsigned)pl[2] + pr[2]) * fw + (1u << 23)) >> 24);
  }
}

// [FLIP_LEFT_RIGHT] -> ToTensor -> x * 2 - 1 clamped: uint8 [t][s][s][3] -> fp32 [3][t][s][s]
__global__ void clip_finish_kernel(const uint8_t* __restrict__ src, float* __restrict__ out, int T, int H, int W, int flip) {
  const size_t total = (size_t)T * H * W, plane = total;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const size_t o = i - x + (flip ? W - 1 - x : x);
    const uint8_t* p = src + i * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = (float)p[c] / 255.0f;
      v = v * 2.0f - 1.0f;
      out[c * plane + o] = fminf(fmaxf(v, -1.0f), 1.0f);
    }
  }
}

static inline int clip_grid(size_t n) {
  size_t b = (n + 255) / 256;
  return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

}  // namespace cstp

using namespace cstp;

extern "C" int cstp_clip_assemble(void* stream, const uint8_t* frames, int32_t f, int32_t h, int32_t w,
                                  const int32_t* frame_idx, int32_t t, int32_t rot, int32_t box_x0, int32_t box_y0,
                                  int32_t size, int32_t flip, const int32_t* kh, const int32_t* bh, int32_t ksh,
                                  const int32_t* kv, const int32_t* bv, int32_t ksv, int32_t row_first, int32_t rows,
                                  uint8_t* tmp, float* out) {
  CSTP_REQUIRE(frames && frame_idx && kh && bh && kv && bv && tmp && out, "null argument");
  CSTP_REQUIRE(f > 0 && h > 0 && w > 0 && t > 0 && size > 0 && ksh > 0 && ksv > 0 && rows > 0 && row_first >= 0, "bad shape");
  CSTP_REQUIRE(rot == 0 || rot == 90 || rot == 180 || rot == 270, "rotation must be 0 / 90 / 180 / 270");
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(clip_resize_h_kernel, dim3(clip_grid((size_t)t * rows * size)), dim3(256), 0, s, frames, frame_idx, tmp, f, h, w,
                     t, rot, box_x0, box_y0, size, kh, bh, ksh, row_first, rows);
  CSTP_LAUNCH_CHECK();
  hipLaunchKernelGGL((clip_resize_v_kernel<false>), dim3(clip_grid((size_t)t * size * size)), dim3(256), 0, s, tmp, out, nullptr, t,
                     size, flip ? 1 : 0, kv, bv, ksv, row_first, rows);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_clip_assemble_u8(void* stream, const uint8_t* frames, int32_t f, int32_t h, int32_t w,
                                     const int32_t* frame_idx, int32_t t, int32_t rot, int32_t box_x0, int32_t box_y0,
                                     int32_t size, const int32_t* kh, const int32_t* bh, int32_t ksh, const int32_t* kv,
                                     const int32_t* bv, int32_t ksv, int32_t row_first, int32_t rows, uint8_t* tmp,
                                     uint8_t* out) {
  CSTP_REQUIRE(frames && frame_idx && kh && bh && kv && bv && tmp && out, "null argument");
  CSTP_REQUIRE(f > 0 && h > 0 && w > 0 && t > 0 && size > 0 && ksh > 0 && ksv > 0 && rows > 0 && row_first >= 0, "bad shape");
  CSTP_REQUIRE(rot == 0 || rot == 90 || rot == 180 || rot == 270, "rotation must be 0 / 90 / 180 / 270");
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(clip_resize_h_kernel, dim3(clip_grid((size_t)t * rows * size)), dim3(256), 0, s, frames, frame_idx, tmp, f, h, w,
                     t, rot, box_x0, box_y0, size, kh, bh, ksh, row_first, rows);
  CSTP_LAUNCH_CHECK();
  hipLaunchKernelGGL((clip_resize_v_kernel<true>), dim3(clip_grid((size_t)t * size * size)), dim3(256), 0, s, tmp, nullptr, out, t,
                     size, 0, kv, bv, ksv, row_first, rows);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_clip_rotate(void* stream, const uint8_t* src, uint8_t* dst, int32_t t, int32_t h, int32_t w,
                                const int32_t* coef6) {
  CSTP_REQUIRE(src && dst && coef6 && src != dst, "null or aliased argument");
  CSTP_REQUIRE(t > 0 && h > 0 && w > 0 && h < 16384 && w < 16384, "bad shape");
  hipLaunchKernelGGL(clip_rotate_kernel, dim3(clip_grid((size_t)t * h * w)), dim3(256), 0, as_stream(stream), src, dst, t, h, w,
                     coef6[0], coef6[1], coef6[2], coef6[3], coef6[4], coef6[5]);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_clip_blend(void* stream, const uint8_t* src, uint8_t* dst, int32_t t, int32_t h, int32_t w, int32_t mode,
                               float alpha, int32_t* ws_means) {
  CSTP_REQUIRE(src && dst, "null argument");
  CSTP_REQUIRE(t > 0 && h > 0 && w > 0 && mode >= 0 && mode <= 2, "bad shape or mode");
  CSTP_REQUIRE(mode != 1 || ws_means != nullptr, "contrast needs t int32 of workspace");
  hipStream_t s = as_stream(stream);
  if (mode == 1) {
    hipLaunchKernelGGL(clip_luma_mean_kernel, dim3(t), dim3(256), 0, s, src, ws_means, h * w);
    CSTP_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(clip_blend_kernel, dim3(clip_grid((size_t)t * h * w)), dim3(256), 0, s, src, dst, t, h * w, mode, alpha, ws_means);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_clip_hue(void* stream, const uint8_t* src, uint8_t* dst, size_t npix, int32_t shift, int32_t mode) {
  CSTP_REQUIRE(src && dst, "null argument");
  CSTP_REQUIRE(npix > 0 && shift >= 0 && shift <= 255 && mode >= 0 && mode <= 2, "bad size, shift or mode");
  hipLaunchKernelGGL(clip_hue_kernel, dim3(clip_grid(npix)), dim3(256), 0, as_stream(stream), src, dst, npix, shift, mode);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_clip_gray(void* stream, const uint8_t* src, uint8_t* dst, int32_t t, int32_t h, int32_t w,
                              const int32_t* channel) {
  CSTP_REQUIRE(src && dst && channel, "null argument");
  CSTP_REQUIRE(t > 0 && h > 0 && w > 0, "bad shape");
  hipLaunchKernelGGL(clip_gray_kernel, dim3(clip_grid((size_t)t * h * w)), dim3(256), 0, as_stream(stream), src, dst, t, h * w, channel);
  CSTP_LAUNCH_CHECK();
  return 0;
}

extern "C" int cstp_clip_box_blur(void* stream, uint8_t* img, uint8_t* tmp, int32_t t, int32_t h, int32_t w, int32_t radius,
                                  uint32_t ww, uint32_t fw, int32_t passes) {
  CSTP_REQUIRE(img && tmp && img != tmp, "null or aliased argument");
  CSTP_REQUIRE(t > 0 && h > 0 && w > 0 && radius >= 0 && radius < 4096 && passes >= 1 && passes <= 8, "bad shape, radius or passes");
  // (the weights of a pass sum to at most 1 << 24, so the 32-bit accumulators cannot overflow)
  CSTP_REQUIRE((unsigned long long)ww * (2 * radius + 1) + 2ull * fw <= (1ull << 24), "weights exceed 1 << 24");
  hipStream_t s = as_stream(stream);
  uint8_t* a = img;
  uint8_t* b = tmp;
  for (int v = 0; v < 2; ++v)
    for (int p = 0; p < passes; ++p) {
      hipLaunchKernelGGL(clip_box_blur_kernel, dim3(clip_grid((size_t)t * h * w)), dim3(256), 0, s, a, b, t, h, w, v, radius, ww, fw);
      CSTP_LAUNCH_CHECK();
      uint8_t* sw = a; a = b; b = sw;
    }
  if (a != img) {        // an odd number of launches ends in tmp
    if (hipMemcpyAsync(img, a, (size_t)t * h * w * 3, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail("hipMemcpyAsync failed%s", "");
  }
  return 0;
}

extern "C" int cstp_clip_finish(void* stream, const uint8_t* src, float* out, int32_t t, int32_t h, int32_t w, int32_t flip) {
  CSTP_REQUIRE(src && out, "null argument");
  CSTP_REQUIRE(t > 0 && h > 0 && w > 0, "bad shape");
  hipLaunchKernelGGL(clip_finish_kernel, dim3(clip_grid((size_t)t * h * w)), dim3(256), 0, as_stream(stream), src, out, t, h, w,
                     flip ? 1 : 0);
  CSTP_LAUNCH_CHECK();
  return 0;
}
