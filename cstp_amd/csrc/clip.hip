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
__global__ void clip_resize_v_kernel(const uint8_t* __restrict__ tmp, float* __restrict__ out, int T, int size, int flip,
                                     const int32_t* __restrict__ kv, const int32_t* __restrict__ bv, int ksv, int row_first,
                                     int rows) {
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
  hipLaunchKernelGGL(clip_resize_v_kernel, dim3(clip_grid((size_t)t * size * size)), dim3(256), 0, s, tmp, out, t, size, flip ? 1 : 0,
                     kv, bv, ksv, row_first, rows);
  CSTP_LAUNCH_CHECK();
  return 0;
}
