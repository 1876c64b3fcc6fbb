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
