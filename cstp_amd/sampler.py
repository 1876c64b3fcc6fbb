"""Clip-pair sampler and pretext-label generator of the CSTP pre-training data path -- the host-side decisions of
/root/reference/data_process/datasets.py:859-948 (``repre_train_clip``: playback rate, rotations, temporal overlap) and
/root/reference/data_process/preprocess_data.py:479-565,568-581,713-741 (``ClipRandomSizedCropOverlap`` spatial-overlap crop
geometry, ``ClipRandomHorizontalFlip``, ``TwoClipTransform``'s base-vs-null choice), with every draw taken from one
``random.Random`` in the ORDER the reference takes it from the global ``random`` module, so that a plan is a pure function of
(video length, frame size, options, seed).

What a plan says and what executes it: frame indices, 90-degree rotation codes, crop boxes, flips and the five labels are
decided here on the host (a few dozen integers per pair); the pixels never touch the host -- ``cstp_amd.clip_ops.assemble_pair``
runs rotate -> crop -> bicubic resize -> flip -> tensor -> 'tf' normalise on the decoded uint8 frames resident in HBM.

The reference applies its ``base_transform`` (preprocess_data.py:1110-1121: RandomRotation(10), ClipColorJitter(0.4, 0.4, 0.4,
0.1) with p = 0.8, ClipRandomGray(p = 0.2), ClipGaussianBlur([0.1, 2]) with p = 0.5, then flip / tensor / normalise) to a clip
with probability 0.3 and the ``null_transform`` (flip, tensor, normalise) otherwise (TwoClipTransform :713-741).  The plan
records that choice and, for a base clip, every parameter the branch draws (``BasePlan``), in the reference's draw order --
``transforms.RandomApply`` as the torchvision of the reference's era implements it (``if self.p < random.random(): skip``;
current torchvision draws from torch's generator instead), ``random.shuffle`` of the four colour operations, the channel of
ClipRandomGray per frame from a NumPy ``RandomState`` (the reference calls ``np.random.choice(3)``).  The executor
(cstp_amd.clip_ops) runs both branches on the GPU.  The reference bug that reads clip 2 of the LMDB dataset from
``start_frame`` instead of ``start_frame_2`` (datasets.py:1397) is not replicated.
"""
from __future__ import annotations

import math
import random
from dataclasses import dataclass
from typing import List, Optional, Tuple

PACE = [1, 2, 4, 8]                               # datasets.py:17
OVERLAP_TEM_RATE = [1.0, 0.8, 0.6, 0.4, 0.2]      # datasets.py:18
ROTATE = [0, 90, 180, 270]                        # datasets.py:19 (Image.ROTATE_90 / 180 / 270, counter-clockwise)
OVERLAP_SPA_RATE = [1.0, 0.8, 0.6, 0.4, 0.2]      # preprocess_data.py:18


@dataclass
class BasePlan:
    """The draws of one pass through base_transform (preprocess_data.py:1110-1121), applied to the resized 8-bit frames."""
    angle: float                               # RandomRotation(10): Image.rotate(angle)
    jitter: Optional[List[Tuple[str, float]]]  # ClipColorJitter: ("brightness" | "contrast" | "saturation" | "hue", factor) in
                                               #   the shuffled order, or None (RandomApply p = 0.8 skipped it)
    gray: Optional[List[int]]                  # ClipRandomGray: the channel kept per frame, or None
    blur_sigma: Optional[float]                # ClipGaussianBlur: one sigma per clip, or None (RandomApply p = 0.5)


@dataclass
class ClipPlan:
    frames: List[int]            # 0-based frame indices, in clip order
    rotate: int                  # 0 / 90 / 180 / 270, applied to the whole frame before cropping
    box: Tuple[int, int, int, int]   # (x0, y0, x1, y1) in the ROTATED frame
    flip: bool
    use_base: bool               # TwoClipTransform chose base_transform for this clip (p = 0.3)
    base: Optional[BasePlan] = None   # its draws (None: null_transform)


@dataclass
class PairPlan:
    clip_1: ClipPlan
    clip_2: ClipPlan
    spa_label: int               # spatial-overlap class 0..4  (preprocess_data.py:520)
    tem_label: int               # temporal-overlap class 0..4 (datasets.py:915)
    pb_label: int                # playback-rate class 0..3    (datasets.py:872-873)
    rot_labels: Tuple[int, int]  # rotation classes 0..3       (datasets.py:878-881)


def sample_frames(total_frames: int, sample_duration: int, rng: random.Random):
    """datasets.py:859-948 without the image I/O: -> (idx_1, idx_2, tem_label, pb_label, (rot_label_1, rot_label_2)),
    0-based frame indices (the reference's file names are 1-based: '%05d.jpg' % (start_frame + i))."""
    max_pb = int(math.log2(total_frames / (sample_duration - 1)))
    pb_label = rng.randint(0, min(3, max_pb))
    sample_rate = PACE[pb_label]
    clip_range = (sample_duration - 1) * sample_rate
    rot_label_1 = rng.randint(0, 3)
    rot_label_2 = rng.randint(0, 3)
    if total_frames - clip_range <= 0:
        # short video: wrap around, both clips read the same frames, temporal label 0 (:884-909)
        index_clip, idx_frame = [], 0
        while len(index_clip) < sample_duration:
            index_clip.append(idx_frame)
            idx_frame += sample_rate
            if idx_frame >= total_frames:
                idx_frame = 0
        return list(index_clip), list(index_clip), 0, pb_label, (rot_label_1, rot_label_2)
    start_frame = rng.randint(1, total_frames - clip_range)
    while True:
        tem_label = rng.randint(0, 4)
        tem_rate = OVERLAP_TEM_RATE[tem_label]
        front_behind = rng.randint(0, 1)
        if front_behind == 0:
            start_frame_2 = start_frame - int((1 - tem_rate) * clip_range)
            if start_frame_2 < 1:
                continue
        else:
            start_frame_2 = start_frame + int((1 - tem_rate) * clip_range)
            if start_frame_2 > total_frames - clip_range:
                continue
        offs = list(range(0, clip_range + 1, sample_rate))
        return ([start_frame - 1 + i for i in offs], [start_frame_2 - 1 + i for i in offs], tem_label, pb_label,
                (rot_label_1, rot_label_2))


class OverlapCrop:
    """ClipRandomSizedCropOverlap (preprocess_data.py:479-565): the first call (flag 0) picks a random-sized crop, the second
    (flag 1) a crop of the same size whose overlap with the first is one of OVERLAP_SPA_RATE, anchored at a random corner."""

    def __init__(self, rng: random.Random, p: float = 1.0, bottom_area: float = 0.2):
        self.rng, self.threshold, self.bottom_area = rng, p, bottom_area
        self.pick_size = None
        self.pick_loc = None

    def first(self, img_w: int, img_h: int) -> Tuple[int, int, int, int]:
        rng = self.rng
        if not rng.random() < self.threshold:
            raise NotImplementedError("p < 1 (center crop) is not on the pre-training path (get_transforms builds p = 1.0)")
        while True:
            area = img_w * img_h
            target_area = rng.uniform(self.bottom_area, 1) * area
            aspect_ratio = rng.uniform(3. / 4, 4. / 3)
            w = int(round(math.sqrt(target_area * aspect_ratio)))
            h = int(round(math.sqrt(target_area / aspect_ratio)))
            if rng.random() < 0.5:
                w, h = h, w
            if w <= img_w and h <= img_h:
                x1 = rng.randint(0, img_w - w)
                y1 = rng.randint(0, img_h - h)
                self.pick_size = [w, h]
                self.pick_loc = [x1, y1]
                return (x1, y1, x1 + w, y1 + h)

    def second(self, img_w: int, img_h: int):
        rng = self.rng
        if not rng.random() < self.threshold:
            raise NotImplementedError("p < 1 (center crop) is not on the pre-training path")
        p_w, p_h = self.pick_size
        p_x, p_y = self.pick_loc
        while True:
            rng.uniform(self.bottom_area, 1)          # the reference draws (and ignores) these two on every attempt (:496-497)
            rng.uniform(3. / 4, 4. / 3)
            spa_label = rng.randint(0, 4)
            spa_rate = OVERLAP_SPA_RATE[spa_label]
            corner = rng.randint(0, 3)
            s_w = rng.randint(int(spa_rate * p_w), p_w)
            s_h = int(spa_rate * p_w * p_h / s_w)
            if corner == 0:
                e_w, e_h = p_x + s_w, p_y + s_h
                ok = e_w - p_w >= 0 and e_h - p_h >= 0
            elif corner == 1:
                e_w, e_h = p_x + p_w - s_w + p_w, p_y + s_h
                ok = e_w <= img_w and e_h - p_h >= 0
            elif corner == 2:
                e_w, e_h = p_x + s_w, p_y + p_h - s_h + p_h
                ok = e_w - p_w >= 0 and e_h <= img_h
            else:
                e_w, e_h = p_x + p_w - s_w + p_w, p_y + p_h - s_h + p_h
                ok = e_w <= img_w and e_h <= img_h
            if ok:
                return (e_w - p_w, e_h - p_h, e_w, e_h), spa_label


def _rotated_size(w: int, h: int, rot: int) -> Tuple[int, int]:
    return (h, w) if rot in (90, 270) else (w, h)


def base_draws(rng: random.Random, np_rng, n_frames: int, sample_duration: int) -> BasePlan:
    """base_transform's random draws up to (not including) the flip, in Compose order (preprocess_data.py:1110-1119)."""
    angle = rng.uniform(-10, 10)                               # RandomRotation(10) :1091
    jitter = None
    if not 0.8 < rng.random():                                 # transforms.RandomApply(p = 0.8)
        rng.random()                                           # ClipColorJitter.__call__: random.random() < p (= 1.0) :658
        ops = [("brightness", rng.uniform(0.6, 1.4)), ("contrast", rng.uniform(0.6, 1.4)),
               ("saturation", rng.uniform(0.6, 1.4)), ("hue", rng.uniform(-0.1, 0.1))]       # get_params :632-648
        rng.shuffle(ops)                                       # random.shuffle(transforms) :650
        jitter = ops
    gray = None
    if rng.random() < 0.2:                                     # ClipRandomGray(p = 0.2) :699
        if np_rng is None:
            raise ValueError("the ClipRandomGray channel draws need a numpy RandomState (np.random.choice(3), :705)")
        gray = [int(np_rng.choice(3)) for _ in range(n_frames)]
    sigma = None
    if not 0.5 < rng.random():                                 # transforms.RandomApply(p = 0.5)
        for idx in range(n_frames):                            # ClipGaussianBlur :682-686: a new sigma every sample_duration frames
            if idx % sample_duration == 0:
                sigma = rng.uniform(0.1, 2.0)
    return BasePlan(angle, jitter, gray, sigma)


def sample_pair(total_frames: int, frame_w: int, frame_h: int, sample_duration: int, rng: random.Random,
                p_base: float = 0.3, np_rng=None) -> PairPlan:
    """One training sample of the 'pre_train' pipeline: repre_train_clip (frames, rotations, temporal / playback labels), then
    TwoClipTransform (preprocess_data.py:713-741): base-or-null draw for each clip, overlap crop of clip 1, its flip, overlap
    crop of clip 2, its flip -- with the base_transform draws of a clip (``base_draws``) between its crop and its flip.  Both clips of a pair are cropped in the coordinates of THEIR OWN rotated frames; the reference
    takes the image size from the first frame of each clip list, which is what this does."""
    idx_1, idx_2, tem_label, pb_label, (r1, r2) = sample_frames(total_frames, sample_duration, rng)
    rng.choices(range(2), weights=[1, 0])                     # TransformController picks TwoClipTransform (:779, weights [1, 0])
    use_base_1 = rng.random() < p_base
    use_base_2 = rng.random() < p_base
    crop = OverlapCrop(rng)
    w1, h1 = _rotated_size(frame_w, frame_h, ROTATE[r1])
    box_1 = crop.first(w1, h1)
    base_1 = base_draws(rng, np_rng, len(idx_1), sample_duration) if use_base_1 else None     # q = tr1(q) :736
    flip_1 = rng.random() < 0.5
    w2, h2 = _rotated_size(frame_w, frame_h, ROTATE[r2])
    box_2, spa_label = crop.second(w2, h2)
    base_2 = base_draws(rng, np_rng, len(idx_2), sample_duration) if use_base_2 else None     # k = tr2(k) :738
    flip_2 = rng.random() < 0.5
    return PairPlan(ClipPlan(idx_1, ROTATE[r1], box_1, flip_1, use_base_1, base_1),
                    ClipPlan(idx_2, ROTATE[r2], box_2, flip_2, use_base_2, base_2), spa_label, tem_label, pb_label, (r1, r2))
