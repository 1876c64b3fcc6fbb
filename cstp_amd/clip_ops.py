"""Host side of the GPU clip assembly (libcstp_hip.so: cstp_clip_assemble): coefficient tables of Pillow's bicubic resize,
the launch, and a clip-pair builder / dataset that turns decoded uint8 videos resident in HBM into the reference's training
sample ``([clip_1, clip_2], [spa_label, tem_label, pb_label, [rot_label_1, rot_label_2]])`` (datasets.py:855-857) with the
decisions of cstp_amd.sampler.

Replaces the reference's per-worker PIL pipeline, both its `null_transform` path and its `base_transform` path (small-angle
rotation, colour jitter, channel gray, Gaussian blur on the resized 8-bit frames: preprocess_data.py:1110-1121) -- (Image.open -> transpose -> crop -> resize ->
flip -> ToTensor -> normalise on the CPU, 6 DataLoader workers per GPU, preprocess_data.py:1103-1130): the frames are uploaded
once as uint8 and every clip is produced where it is consumed.  There is no CPU implementation here.
"""
from __future__ import annotations

import ctypes
import math
import random
from functools import lru_cache
from typing import List

import numpy as np
import torch

from . import _lib, sampler
from ._lib import check

PRECISION_BITS = 32 - 8 - 2        # Pillow: libImaging/Resample.c


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


@lru_cache(maxsize=4096)
def resize_tables(in_size: int, out_size: int):
    """(ksize, bounds int32 [out][2], coefficients int32 [out][ksize]) of Image.resize(..., BICUBIC) along one axis of length
    in_size -> out_size: Resample.c precompute_coeffs (double arithmetic, same operation order) + normalize_coeffs_8bpc."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        for x in range(xmax):
            v = k[x] / ww if ww != 0.0 else k[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


_dev_tables = {}


def _device_tables(in_size: int, out_size: int, device: torch.device):
    key = (in_size, out_size, device.index)
    t = _dev_tables.get(key)
    if t is None:
        if len(_dev_tables) > 8192:
            _dev_tables.clear()
        ks, b, k = resize_tables(in_size, out_size)
        t = (ks, torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), int(b[0, 0]), int(b[-1, 0] + b[-1, 1]))
        _dev_tables[key] = t
    return t


# ---- the base_transform branch: host-side constants of the Pillow algorithms the kernels reproduce --------------------------
def rotate_coeffs(w: int, h: int, angle: float):
    """Image.rotate(angle)'s reverse affine matrix about (w / 2, h / 2) (PIL/Image.py) as Geometry.c affine_fixed's six 16.16
    fixed-point coefficients; None for the angles Image.rotate serves by a transpose or a copy."""
    a = angle % 360.0
    if a == 0 or a == 180 or (a in (90, 270) and w == h):
        return None
    r = -math.radians(a)
    m = [round(math.cos(r), 15), round(math.sin(r), 15), 0.0, round(-math.sin(r), 15), round(math.cos(r), 15), 0.0]
    cx, cy = w / 2, h / 2
    m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2] + cx
    m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5] + cy

    def fix(v):
        return int(math.floor(v * 65536.0 + 0.5))
    return [fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5)]


def gaussian_box_weights(sigma: float, passes: int = 3):
    """ImageFilter.GaussianBlur(sigma) -> (integer box radius, ww, fw) of BoxBlur.c: _gaussian_blur_radius in its float / double
    mix, then the two 24-bit fixed-point weights of ImagingLineBoxBlur8 (a float division)."""
    f32, f64 = np.float32, np.float64
    sigma2 = f32(f32(sigma) * f32(sigma) / f32(passes))
    big_l = f32(math.sqrt(12.0 * f64(sigma2) + 1.0))
    small_l = f32(math.floor((f64(big_l) - 1.0) / 2.0))
    a = f32(f64(f32(2) * small_l + f32(1)) * (f64(small_l * (small_l + f32(1))) - 3.0 * f64(sigma2)))
    a = f32(a / f32(f32(6) * f32(sigma2 - (small_l + f32(1)) * (small_l + f32(1)))))
    radius = f32(small_l + a)
    r_int = int(radius)
    ww = int(f32(1 << 24) / f32(radius * f32(2) + f32(1)))
    fw = ((1 << 24) - (r_int * 2 + 1) * ww) // 2
    return r_int, ww, fw


def _u8_clip(t):
    if not t.is_cuda or t.dtype != torch.uint8 or t.dim() != 4 or t.shape[3] != 3:
        raise _lib.CstpError("expected a uint8 [T, H, W, 3] clip on a HIP device (cstp_amd has no CPU path)")
    return t.contiguous()


def clip_rotate(clip: torch.Tensor, angle: float) -> torch.Tensor:
    """Every frame of the clip through Image.rotate(angle) (RandomRotation, preprocess_data.py:1091-1094)."""
    clip = _u8_clip(clip)
    t, h, w, _ = clip.shape
    a = angle % 360.0
    coef = rotate_coeffs(w, h, angle)
    if coef is None:      # Image.rotate's fast paths: copy / transpose
        return clip.clone() if a == 0 else torch.rot90(clip, {90: 1, 180: 2, 270: 3}[int(a)], dims=(1, 2)).contiguous()
    out = torch.empty_like(clip)
    check(_lib.load().cstp_clip_rotate(torch.cuda.current_stream().cuda_stream, clip.data_ptr(), out.data_ptr(), t, h, w,
                                       (ctypes.c_int32 * 6)(*coef)), "cstp_clip_rotate")
    return out


_BLEND_MODE = {"brightness": 0, "contrast": 1, "saturation": 2}


def clip_colour(clip: torch.Tensor, op: str, factor: float) -> torch.Tensor:
    """torchvision adjust_brightness / adjust_contrast / adjust_saturation / adjust_hue on every frame (ClipColorJitter)."""
    clip = _u8_clip(clip)
    t, h, w, _ = clip.shape
    lib, st = _lib.load(), torch.cuda.current_stream().cuda_stream
    out = torch.empty_like(clip)
    if op == "hue":
        if not -0.5 <= factor <= 0.5:
            raise ValueError("hue_factor is not in [-0.5, 0.5]")
        shift = int(np.array(factor * 255).astype(np.uint8))
        check(lib.cstp_clip_hue(st, clip.data_ptr(), out.data_ptr(), t * h * w, shift, 0), "cstp_clip_hue")
        return out
    if op not in _BLEND_MODE:
        raise ValueError("colour operation %r" % (op,))
    if factor == 1.0:      # Image.blend returns a copy of the image
        return clip.clone()
    means = torch.empty(t, dtype=torch.int32, device=clip.device) if op == "contrast" else None
    check(lib.cstp_clip_blend(st, clip.data_ptr(), out.data_ptr(), t, h, w, _BLEND_MODE[op], float(factor),
                              None if means is None else means.data_ptr()), "cstp_clip_blend")
    return out


def clip_gray(clip: torch.Tensor, channels) -> torch.Tensor:
    """ClipRandomGray.grayscale: frame i keeps channels[i] in all three channels."""
    clip = _u8_clip(clip)
    t, h, w, _ = clip.shape
    if len(channels) != t:
        raise ValueError("%d channel choices for %d frames" % (len(channels), t))
    ch = torch.tensor([int(c) for c in channels], dtype=torch.int32, device=clip.device)
    out = torch.empty_like(clip)
    check(_lib.load().cstp_clip_gray(torch.cuda.current_stream().cuda_stream, clip.data_ptr(), out.data_ptr(), t, h, w,
                                     ch.data_ptr()), "cstp_clip_gray")
    return out


def clip_gaussian_blur(clip: torch.Tensor, sigma: float) -> torch.Tensor:
    """Every frame through ImageFilter.GaussianBlur(radius=sigma) (ClipGaussianBlur)."""
    clip = _u8_clip(clip)
    if sigma == 0:
        return clip.clone()
    t, h, w, _ = clip.shape
    r_int, ww, fw = gaussian_box_weights(sigma)
    out, tmp = clip.clone(), torch.empty_like(clip)
    check(_lib.load().cstp_clip_box_blur(torch.cuda.current_stream().cuda_stream, out.data_ptr(), tmp.data_ptr(), t, h, w, r_int,
                                         ww, fw, 3), "cstp_clip_box_blur")
    return out


def clip_finish(clip: torch.Tensor, flip: bool) -> torch.Tensor:
    """[flip] -> ToTensor -> 'tf' normalise: uint8 [T][S][S][3] -> fp32 [3][T][S][S]."""
    clip = _u8_clip(clip)
    t, h, w, _ = clip.shape
    out = torch.empty((3, t, h, w), dtype=torch.float32, device=clip.device)
    check(_lib.load().cstp_clip_finish(torch.cuda.current_stream().cuda_stream, clip.data_ptr(), out.data_ptr(), t, h, w,
                                       1 if flip else 0), "cstp_clip_finish")
    return out


def apply_base_transform(clip: torch.Tensor, base: "sampler.BasePlan", flip: bool) -> torch.Tensor:
    """base_transform (preprocess_data.py:1110-1121) on the resized 8-bit clip, in Compose order."""
    clip = clip_rotate(clip, base.angle)
    for op, factor in (base.jitter or ()):
        clip = clip_colour(clip, op, factor)
    if base.gray is not None:
        clip = clip_gray(clip, base.gray)
    if base.blur_sigma is not None:
        clip = clip_gaussian_blur(clip, base.blur_sigma)
    return clip_finish(clip, flip)


def assemble_clip(frames: torch.Tensor, plan: "sampler.ClipPlan", size: int) -> torch.Tensor:
    """frames: uint8 [F][H][W][3] on a HIP device -> fp32 [3][T][size][size] (torch.stack(clip).transpose(0, 1)).
    A plan that carries base_transform draws (plan.base) is resized to 8-bit frames first and taken through that branch."""
    lib = _lib.load()
    if not frames.is_cuda or frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3:
        raise _lib.CstpError("frames must be a uint8 [F, H, W, 3] tensor on a HIP device (cstp_amd has no CPU path)")
    frames = frames.contiguous()
    f, h, w, _ = frames.shape
    x0, y0, x1, y1 = plan.box
    if not (x0 < x1 and y0 < y1):
        raise _lib.CstpError("empty crop box %s" % (plan.box,))
    dev = frames.device
    ksh, bh, kh, _, _ = _device_tables(x1 - x0, size, dev)
    ksv, bv, kv, first, last = _device_tables(y1 - y0, size, dev)
    t = len(plan.frames)
    idx = torch.tensor(plan.frames, dtype=torch.int32, device=dev)
    tmp = torch.empty((t, last - first, size, 3), dtype=torch.uint8, device=dev)
    if getattr(plan, "base", None) is not None:
        u8 = torch.empty((t, size, size, 3), dtype=torch.uint8, device=dev)
        check(lib.cstp_clip_assemble_u8(torch.cuda.current_stream().cuda_stream, frames.data_ptr(), f, h, w, idx.data_ptr(), t,
                                        int(plan.rotate), int(x0), int(y0), int(size), kh.data_ptr(), bh.data_ptr(), ksh,
                                        kv.data_ptr(), bv.data_ptr(), ksv, first, last - first, tmp.data_ptr(), u8.data_ptr()),
              "cstp_clip_assemble_u8")
        return apply_base_transform(u8, plan.base, plan.flip)
    out = torch.empty((3, t, size, size), dtype=torch.float32, device=dev)
    check(lib.cstp_clip_assemble(torch.cuda.current_stream().cuda_stream, frames.data_ptr(), f, h, w, idx.data_ptr(), t,
                                 int(plan.rotate), int(x0), int(y0), int(size), 1 if plan.flip else 0, kh.data_ptr(), bh.data_ptr(),
                                 ksh, kv.data_ptr(), bv.data_ptr(), ksv, first, last - first, tmp.data_ptr(), out.data_ptr()),
          "cstp_clip_assemble")
    return out


def assemble_pair(frames: torch.Tensor, plan: "sampler.PairPlan", size: int):
    """-> ([clip_1, clip_2], [spa_label, tem_label, pb_label, [rot_label_1, rot_label_2]])  (datasets.py:855-857)."""
    return ([assemble_clip(frames, plan.clip_1, size), assemble_clip(frames, plan.clip_2, size)],
            [plan.spa_label, plan.tem_label, plan.pb_label, list(plan.rot_labels)])


class GpuVideoClips:
    """Stands in for UcfRepre / Kin400RepreLMDB on synthetic data: ``n_videos`` decoded videos (uint8 frames, smooth moving
    patterns) live in HBM; batch(i) draws clip pairs with cstp_amd.sampler and assembles them on the device.  Selected by the
    pre-training driver with ``--dataset synthetic_video``."""

    def __init__(self, device, n_videos=4, frames=96, height=128, width=171, sample_duration=16, sample_size=112, length=256,
                 seed=1):
        self.device, self.t, self.size, self.length, self.seed = torch.device(device), sample_duration, sample_size, length, seed
        g = torch.Generator(device=self.device).manual_seed(seed)
        ys = torch.linspace(0, 1, height, device=self.device).view(1, height, 1, 1)
        xs = torch.linspace(0, 1, width, device=self.device).view(1, 1, width, 1)
        ts = torch.linspace(0, 1, frames, device=self.device).view(frames, 1, 1, 1)
        self.videos = []
        for v in range(n_videos):
            ph = torch.rand(3, generator=g, device=self.device).view(1, 1, 1, 3) * 6.28
            fr = 2 + 3 * torch.rand(3, generator=g, device=self.device).view(1, 1, 1, 3)
            img = 0.5 + 0.35 * torch.sin(6.28 * (fr * xs + (v + 1) * ys) + ph + 6.28 * ts) \
                + 0.15 * torch.rand((frames, height, width, 3), generator=g, device=self.device)
            self.videos.append((img.clamp(0, 1) * 255).to(torch.uint8).contiguous())

    def __len__(self):
        return self.length

    def sample(self, index: int):
        rng = random.Random(self.seed * 1000003 + index)
        np_rng = np.random.RandomState((self.seed * 1000003 + index) & 0x7fffffff)      # ClipRandomGray's np.random.choice
        video = self.videos[index % len(self.videos)]
        f, h, w, _ = video.shape
        return assemble_pair(video, sampler.sample_pair(f, w, h, self.t, rng, np_rng=np_rng), self.size)

    def batch(self, indices: List[int]):
        """-> (clip_1 [B,3,T,S,S], clip_2, spa, tem, pb, rot_1, rot_2) on the device, labels int64."""
        samples = [self.sample(i) for i in indices]
        c1 = torch.stack([s[0][0] for s in samples])
        c2 = torch.stack([s[0][1] for s in samples])
        lab = lambda f: torch.tensor([f(s[1]) for s in samples], dtype=torch.int64, device=self.device)  # noqa: E731
        return (c1, c2, lab(lambda l: l[0]), lab(lambda l: l[1]), lab(lambda l: l[2]), lab(lambda l: l[3][0]),
                lab(lambda l: l[3][1]))


class GpuClipLoader:
    """DataLoader + DistributedSampler for GpuVideoClips in one object: per epoch an epoch-seeded permutation of the sample
    indices, this rank's stride of it (utils.py:109-113 semantics: shuffle, drop_last, per-rank batch), batches assembled on the
    device in the reference's collated layout ``([clip_1, clip_2], [spa, tem, pb, [rot_1, rot_2]])``.  No worker processes and
    no host->device copy: the training loop's ``.to(device)`` calls are no-ops on these tensors."""

    def __init__(self, dataset: GpuVideoClips, batch_size: int, rank: int = 0, world_size: int = 1, seed: int = 0):
        self.dataset, self.batch_size, self.rank, self.world_size, self.seed, self.epoch = dataset, batch_size, rank, world_size, \
            seed, 0
        if batch_size < 1 or not 0 <= rank < world_size:
            raise ValueError("batch_size %r, rank %r of %r" % (batch_size, rank, world_size))

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def indices(self) -> List[int]:
        order = list(range(len(self.dataset)))
        random.Random(self.seed * 7919 + self.epoch).shuffle(order)
        per_rank = len(order) // self.world_size                  # every rank the same count (no padding duplicates)
        return order[self.rank:per_rank * self.world_size:self.world_size]

    def __len__(self):
        return (len(self.dataset) // self.world_size) // self.batch_size

    def __iter__(self):
        idx = self.indices()
        for b in range(len(self)):
            c1, c2, spa, tem, pb, r1, r2 = self.dataset.batch(idx[b * self.batch_size:(b + 1) * self.batch_size])
            yield [c1, c2], [spa, tem, pb, [r1, r2]]
