"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the PIL operations on the reference's `null_transform` clip path
(data_process/preprocess_data.py:479-581,1103-1110; data_process/datasets.py:19,876-881): Image.transpose(ROTATE_90/180/270,
FLIP_LEFT_RIGHT), Image.crop, Image.resize(size, Image.BICUBIC) on 8-bit RGB, ToTensor and the 'tf' normalisation.

The resize follows Pillow's published algorithm (src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
ImagingResampleHorizontal_8bpc / Vertical_8bpc): separable, horizontal pass first with a uint8 intermediate, coefficients in
22-bit fixed point, accumulator seeded with 1 << 21, result clamped to 0..255 after an arithmetic shift.  Pillow is a
dependency of the reference that IS present in this image (PIL 12.2.0): tests/test_clip_oracle.py pins every function here
bit-for-bit against PIL itself.  Nothing in cstp_amd imports this file.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
ROTATE_CODES = (0, 90, 180, 270)      # datasets.py:19  ROTATE = [0, Image.ROTATE_90, Image.ROTATE_180, Image.ROTATE_270]


def bicubic_filter(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, in0: float, in1: float, out_size: int):
    """-> (ksize, bounds int32 [out][2] = (first tap, tap count), integer coefficients int32 [out][ksize])."""
    scale = filterscale = (in1 - in0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        for x in range(xmax):
            v = k[x] / ww if ww != 0.0 else k[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _clip8(acc: np.ndarray) -> np.ndarray:
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bicubic(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """img uint8 [H][W][C] -> uint8 [out_h][out_w][C], as Image.resize((out_w, out_h), Image.BICUBIC)."""
    h, w, _ = img.shape
    need_h, need_v = out_w != w, out_h != h
    cur = img
    _, bv, kv = precompute_coeffs(h, 0.0, float(h), out_h) if need_v else (0, None, None)
    if need_h:
        _, bh, kh = precompute_coeffs(w, 0.0, float(w), out_w)
        first, last = (int(bv[0, 0]), int(bv[-1, 0] + bv[-1, 1])) if need_v else (0, h)
        tmp = np.zeros((last - first, out_w, img.shape[2]), dtype=np.uint8)
        src = cur[first:last].astype(np.int64)
        for xx in range(out_w):
            x0, n = int(bh[xx, 0]), int(bh[xx, 1])
            acc = np.full((last - first, img.shape[2]), 1 << (PRECISION_BITS - 1), dtype=np.int64)
            acc += (src[:, x0:x0 + n, :] * kh[xx, :n].astype(np.int64)[None, :, None]).sum(axis=1)
            tmp[:, xx, :] = _clip8(acc)
        cur = tmp
        if need_v:
            bv = bv.copy()
            bv[:, 0] -= first
    if need_v:
        out = np.zeros((out_h, cur.shape[1], img.shape[2]), dtype=np.uint8)
        src = cur.astype(np.int64)
        for yy in range(out_h):
            y0, n = int(bv[yy, 0]), int(bv[yy, 1])
            acc = np.full((cur.shape[1], img.shape[2]), 1 << (PRECISION_BITS - 1), dtype=np.int64)
            acc += (src[y0:y0 + n] * kv[yy, :n].astype(np.int64)[:, None, None]).sum(axis=0)
            out[yy] = _clip8(acc)
        cur = out
    return cur


def transpose(img: np.ndarray, code: int) -> np.ndarray:
    """Image.transpose: code 90 / 180 / 270 = ROTATE_* (counter-clockwise), 'flip' = FLIP_LEFT_RIGHT, 0 = identity."""
    if code == 0:
        return img
    if code == 90:
        return np.ascontiguousarray(np.rot90(img, 1))
    if code == 180:
        return np.ascontiguousarray(np.rot90(img, 2))
    if code == 270:
        return np.ascontiguousarray(np.rot90(img, 3))
    if code == "flip":
        return np.ascontiguousarray(img[:, ::-1])
    raise ValueError(code)


def crop(img: np.ndarray, box) -> np.ndarray:
    """Image.crop((x0, y0, x1, y1)); pixels outside the image are 0 (libImaging/Crop.c pastes into a zero-filled image).  The
    reference's second crop can reach past the frame when the two clips are rotated differently (preprocess_data.py:535-541)."""
    x0, y0, x1, y1 = box
    h, w = img.shape[:2]
    out = np.zeros((max(y1 - y0, 0), max(x1 - x0, 0)) + img.shape[2:], dtype=img.dtype)
    sx0, sy0, sx1, sy1 = max(x0, 0), max(y0, 0), min(x1, w), min(y1, h)
    if sx1 > sx0 and sy1 > sy0:
        out[sy0 - y0:sy1 - y0, sx0 - x0:sx1 - x0] = img[sy0:sy1, sx0:sx1]
    return out


def to_tensor_tf(img: np.ndarray) -> np.ndarray:
    """transforms.ToTensor() (uint8 HWC -> float32 CHW / 255) then ClipNormalize('tf'): x * 2 - 1, clamped to [-1, 1]
    (preprocess_data.py:358-364)."""
    t = img.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    return np.clip(t * np.float32(2.0) - np.float32(1.0), -1.0, 1.0).astype(np.float32)


def assemble_clip(frames: np.ndarray, idx, rot_code: int, box, size: int, flip: bool) -> np.ndarray:
    """frames uint8 [F][H][W][3]; per frame: transpose(rot) -> crop(box) -> resize(size, size, BICUBIC) -> [flip] -> tensor.
    Returns float32 [3][T][size][size] (torch.stack(clip).transpose(0, 1), datasets.py:856)."""
    out = []
    for f in idx:
        im = transpose(frames[f], rot_code)
        im = resize_bicubic(crop(im, box), size, size)
        if flip:
            im = transpose(im, "flip")
        out.append(to_tensor_tf(im))
    return np.stack(out, axis=1)


# ------------------------------------------------------------------------------------------------------------------------
# The `base_transform` branch (preprocess_data.py:1110-1121), applied to the 112 x 112 crops of a clip with probability 0.3
# (TwoClipTransform :713-741): RandomRotation(10) :1060-1100 (Image.rotate, NEAREST), ClipColorJitter :584-672 (torchvision's
# adjust_brightness / _contrast / _saturation = PIL ImageEnhance blends, adjust_hue = an HSV round trip with a uint8 hue shift),
# ClipRandomGray :690-711 (one channel copied into all three), ClipGaussianBlur :675-687 (ImageFilter.GaussianBlur = three
# extended box blurs per axis).  Restated from Pillow's published algorithms (libImaging/Geometry.c affine_fixed, Blend.c,
# Convert.c rgb2l / rgb2hsv / hsv2rgb, BoxBlur.c) and pinned bit-for-bit against Pillow itself in tests/test_clip_oracle.py.
# ------------------------------------------------------------------------------------------------------------------------
def rotate_matrix(w: int, h: int, angle: float):
    """Image.rotate's reverse affine matrix (Image.py: rotation about (w / 2, h / 2), no expand, no translation)."""
    a = -math.radians(angle % 360.0)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    cx, cy = w / 2, h / 2
    m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2]
    m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5]
    m[2] += cx
    m[5] += cy
    return m


def affine_fixed_coeffs(m):
    """Geometry.c affine_fixed: the six coefficients in 16.16 fixed point, the half-pixel centre folded into the offsets."""
    def fix(v):
        return int(math.floor(v * 65536.0 + 0.5))
    return (fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5))


def rotate_nearest(img: np.ndarray, angle: float) -> np.ndarray:
    """Image.rotate(angle) with the defaults the reference uses (:1094): NEAREST, same size, black fill."""
    a = angle % 360.0
    h, w = img.shape[:2]
    if a == 0:
        return img.copy()
    if a == 180:
        return transpose(img, 180)
    if a in (90, 270) and w == h:
        return transpose(img, int(a))
    a0, a1, a2, a3, a4, a5 = affine_fixed_coeffs(rotate_matrix(w, h, angle))
    ys, xs = np.mgrid[0:h, 0:w].astype(np.int64)
    xin = (a2 + a1 * ys + a0 * xs) >> 16
    yin = (a5 + a4 * ys + a3 * xs) >> 16
    ok = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
    out = np.zeros_like(img)
    out[ok] = img[yin[ok], xin[ok]]
    return out


def rgb_to_l(img: np.ndarray) -> np.ndarray:
    """Image.convert('L') of an RGB image: ITU-R 601-2 luma in 16-bit fixed point (Convert.c L24 / rgb2l)."""
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(deg: np.ndarray, img: np.ndarray, alpha: float) -> np.ndarray:
    """Image.blend(deg, img, alpha) (Blend.c): interpolation inside [0, 1] truncates, extrapolation clips then truncates;
    the arithmetic is single precision."""
    if alpha == 0.0:
        return deg.copy()
    if alpha == 1.0:
        return img.copy()
    a = np.float32(alpha)
    d = deg.astype(np.int32)
    t = d.astype(np.float32) + a * (img.astype(np.int32) - d).astype(np.float32)
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int32))).astype(np.uint8)


def adjust_brightness(img: np.ndarray, factor: float) -> np.ndarray:
    return blend(np.zeros_like(img), img, factor)


def adjust_contrast(img: np.ndarray, factor: float) -> np.ndarray:
    lum = rgb_to_l(img)
    mean = int(float(lum.astype(np.float64).sum()) / lum.size + 0.5)
    return blend(np.full_like(img, mean), img, factor)


def adjust_saturation(img: np.ndarray, factor: float) -> np.ndarray:
    lum = rgb_to_l(img)
    return blend(np.repeat(lum[..., None], 3, axis=2), img, factor)


def rgb_to_hsv(img: np.ndarray) -> np.ndarray:
    """Image.convert('HSV') (Convert.c rgb2hsv_row): float variables, but the constants 2.0 / 4.0 / 6.0 / 255.0 are doubles,
    so each of those statements is evaluated in double and rounded to float on assignment.  Exact on all 2^24 colours."""
    f32, f64 = np.float32, np.float64
    r, g, b = (img[..., i].astype(np.int32) for i in range(3))
    maxc = np.maximum(r, np.maximum(g, b))
    minc = np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        s = cr / maxc.astype(f32)
        rc = (maxc - r).astype(f32) / cr
        gc = (maxc - g).astype(f32) / cr
        bc = (maxc - b).astype(f32) / cr
        h = np.where(r == maxc, (bc - gc).astype(f64),
                     np.where(g == maxc, 2.0 + rc.astype(f64) - bc.astype(f64), 4.0 + gc.astype(f64) - rc.astype(f64))).astype(f32)
        h = np.fmod(h.astype(f64) / 6.0 + 1.0, 1.0).astype(f32)
        uh = np.clip((h.astype(f64) * 255.0).astype(np.int64), 0, 255)
        us = np.clip((s.astype(f64) * 255.0).astype(np.int64), 0, 255)
    gray = minc == maxc
    uh = np.where(gray, 0, uh)
    us = np.where(gray, 0, us)
    return np.stack([uh, us, maxc], axis=-1).astype(np.uint8)


def hsv_to_rgb(img: np.ndarray) -> np.ndarray:
    """Image.convert('RGB') of an HSV image (Convert.c hsv2rgb, single precision, round half away from zero)."""
    f32 = np.float32
    h, s, v = (img[..., i].astype(np.int32) for i in range(3))
    hf = h.astype(f32) * f32(6.0) / f32(255.0)
    i = np.floor(hf).astype(np.int32)
    f = (hf - i.astype(f32)).astype(f32)
    fs = s.astype(f32) / f32(255.0)
    vf = v.astype(f32)

    def rnd(x):
        return np.clip(np.floor(x.astype(np.float64) + 0.5).astype(np.int32), 0, 255)
    p = rnd(vf * (f32(1.0) - fs))
    q = rnd(vf * (f32(1.0) - fs * f))
    t = rnd(vf * (f32(1.0) - fs * (f32(1.0) - f)))
    i = i % 6
    r = np.select([i == 0, i == 1, i == 2, i == 3, i == 4, i == 5], [v, q, p, p, t, v])
    g = np.select([i == 0, i == 1, i == 2, i == 3, i == 4, i == 5], [t, v, v, q, p, p])
    b = np.select([i == 0, i == 1, i == 2, i == 3, i == 4, i == 5], [p, p, t, v, v, q])
    gray = s == 0
    r, g, b = np.where(gray, v, r), np.where(gray, v, g), np.where(gray, v, b)
    return np.stack([r, g, b], axis=-1).astype(np.uint8)


def adjust_hue(img: np.ndarray, factor: float) -> np.ndarray:
    """torchvision.transforms.functional.adjust_hue on a PIL image: HSV, h += uint8(factor * 255) modulo 256, back to RGB."""
    if not -0.5 <= factor <= 0.5:
        raise ValueError("hue_factor is not in [-0.5, 0.5]")
    hsv = rgb_to_hsv(img)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + int(np.array(factor * 255).astype(np.uint8))) & 255
    return hsv_to_rgb(hsv)


def channel_gray(img: np.ndarray, channel: int) -> np.ndarray:
    """ClipRandomGray.grayscale (:704-709): one channel copied into all three."""
    return np.repeat(img[:, :, channel:channel + 1], 3, axis=2)


def gaussian_box_radius(radius: float, passes: int = 3) -> np.float32:
    """BoxBlur.c _gaussian_blur_radius: the extended-box radius whose `passes`-fold convolution has the Gaussian's variance.
    float variables; the statements that contain a double constant (12.0, 1.0, 2.0, 3.0) are evaluated in double and
    rounded to float on assignment."""
    f32, f64 = np.float32, np.float64
    sigma2 = f32(f32(radius) * f32(radius) / f32(passes))
    big_l = f32(math.sqrt(12.0 * f64(sigma2) + 1.0))
    small_l = f32(math.floor((f64(big_l) - 1.0) / 2.0))
    a = f32(f64(f32(2) * small_l + f32(1)) * (f64(small_l * (small_l + f32(1))) - 3.0 * f64(sigma2)))
    a = f32(a / f32(f32(6) * f32(sigma2 - (small_l + f32(1)) * (small_l + f32(1)))))
    return f32(small_l + a)


def _box_blur_line(line: np.ndarray, radius: int, ww: int, fw: int) -> np.ndarray:
    """BoxBlur.c ImagingLineBoxBlur8 on one line [n][channels] (uint8): a box of 2 * radius + 1 pixels of weight ww plus two
    far pixels of weight fw, 24-bit fixed point, the line's edge pixels repeated beyond its ends."""
    n = line.shape[0]
    last = n - 1
    src = line.astype(np.int64)
    idx = np.arange(n)

    def px(i):
        return src[np.clip(i, 0, last)]
    csum = np.concatenate([np.zeros((1,) + src.shape[1:], dtype=np.int64), np.cumsum(src, axis=0)])

    lo, hi = idx - radius, idx + radius
    lo_c, hi_c = np.clip(lo, 0, last), np.clip(hi, 0, last)
    acc = csum[hi_c + 1] - csum[lo_c]
    acc = acc + (lo_c - lo)[:, None] * src[0][None, :] + (hi - hi_c)[:, None] * src[last][None, :]
    bulk = acc * ww + (px(idx - radius - 1) + px(idx + radius + 1)) * fw
    return ((bulk + (1 << 23)) >> 24).astype(np.uint8)


def box_blur(img: np.ndarray, radius: float, passes: int) -> np.ndarray:
    """ImagingBoxBlur with equal x / y radius: `passes` horizontal line blurs, transpose, `passes` again, transpose back."""
    f32 = np.float32
    r_int = int(radius)
    ww = int(f32(1 << 24) / f32(f32(radius) * f32(2) + f32(1)))      # (UINT32)(1 << 24) / (floatRadius * 2 + 1): float division
    fw = ((1 << 24) - (r_int * 2 + 1) * ww) // 2

    def hpass(a):
        out = a
        for _ in range(passes):
            out = np.stack([_box_blur_line(out[y], r_int, ww, fw) for y in range(out.shape[0])])
        return out
    out = hpass(img)
    out = hpass(out.transpose(1, 0, 2)).transpose(1, 0, 2)
    return np.ascontiguousarray(out)


def gaussian_blur(img: np.ndarray, radius: float) -> np.ndarray:
    """Image.filter(ImageFilter.GaussianBlur(radius)) on 8-bit RGB (BoxBlur.c ImagingGaussianBlur, 3 passes)."""
    if radius == 0:
        return img.copy()
    return box_blur(img, gaussian_box_radius(radius, 3), 3)


def base_transform_clip(frames: np.ndarray, base, flip: bool) -> np.ndarray:
    """base_transform (preprocess_data.py:1110-1121) on resized 8-bit frames [T][S][S][3] with the draws of a
    cstp_amd.sampler.BasePlan-like object (angle, jitter [(op, factor)] | None, gray [channel per frame] | None, blur_sigma | None):
    rotate -> colour operations in their drawn order -> channel gray -> Gaussian blur -> [flip] -> tensor.  fp32 [3][T][S][S]."""
    colour = {"brightness": adjust_brightness, "contrast": adjust_contrast, "saturation": adjust_saturation, "hue": adjust_hue}
    out = []
    for i, im in enumerate(frames):
        im = rotate_nearest(im, base.angle)
        for op, factor in (base.jitter or ()):
            im = colour[op](im, factor)
        if base.gray is not None:
            im = channel_gray(im, base.gray[i])
        if base.blur_sigma is not None:
            im = gaussian_blur(im, base.blur_sigma)
        if flip:
            im = transpose(im, "flip")
        out.append(to_tensor_tf(im))
    return np.stack(out, axis=1)
