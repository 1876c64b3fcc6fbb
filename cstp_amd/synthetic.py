"""Synthetic clip pairs with the reference's sample layout (datasets.py:848-857):
``([clip_1, clip_2], [spa, tem, pb, [rot_1, rot_2]])``, clips fp32 3xTxHxW in [-1, 1]
(tf normalisation, preprocess_data.py:361-364), labels int64 with spa,tem in 0..4 and
pb,rot in 0..3 (datasets.py:873-881,915; preprocess_data.py:520).  Stands in for the
out-of-scope PIL/LMDB data pipeline; selected with ``--dataset synthetic``."""
from __future__ import annotations

import torch
from torch.utils.data import Dataset


class SyntheticClips(Dataset):
    def __init__(self, length=256, sample_duration=16, sample_size=112, seed=1):
        self.length, self.t, self.hw, self.seed = length, sample_duration, sample_size, seed

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed * 1000003 + idx)
        shape = (3, self.t, self.hw, self.hw)
        clip_1 = torch.rand(shape, generator=g) * 2 - 1
        clip_2 = torch.rand(shape, generator=g) * 2 - 1
        lab = torch.randint(0, 20, (5,), generator=g)
        return [clip_1, clip_2], [int(lab[0]) % 5, int(lab[1]) % 5, int(lab[2]) % 4, [int(lab[3]) % 4, int(lab[4]) % 4]]


def device_batch(batch, t, hw, device, seed=1):
    """One resident synthetic batch generated directly in HBM (bench / smoke): U[-1,1] clips,
    labels in the reference ranges; seed offset per rank keeps shards disjoint."""
    g = torch.Generator(device=device).manual_seed(seed)
    clip_1 = torch.rand((batch, 3, t, hw, hw), generator=g, device=device) * 2 - 1
    clip_2 = torch.rand((batch, 3, t, hw, hw), generator=g, device=device) * 2 - 1
    labels = {
        "spa": torch.randint(0, 5, (batch,), generator=g, device=device),
        "tem": torch.randint(0, 5, (batch,), generator=g, device=device),
        "pb": torch.randint(0, 4, (batch,), generator=g, device=device),
        "rot1": torch.randint(0, 4, (batch,), generator=g, device=device),
        "rot2": torch.randint(0, 4, (batch,), generator=g, device=device),
    }
    return clip_1, clip_2, labels


def _class_pattern(label: int, n_classes: int, t: int, hw: int) -> torch.Tensor:
    """A smooth class-specific 3xTxHxW pattern in [-1, 1]: spatial orientation/frequency and temporal drift are
    functions of the label, so a classifier on encoder features can separate the classes."""
    import math
    ang = math.pi * label / max(n_classes, 1)
    freq = 1.0 + (label % 4)
    drift = 0.5 * (1 + (label // 4) % 3)
    ys = torch.linspace(-1, 1, hw).view(1, hw, 1)
    xs = torch.linspace(-1, 1, hw).view(1, 1, hw)
    ts = torch.linspace(0, 1, t).view(t, 1, 1)
    phase = freq * math.pi * (math.cos(ang) * xs + math.sin(ang) * ys) + 2 * math.pi * drift * ts
    base = torch.sin(phase)
    return torch.stack((base, torch.cos(phase), -base), dim=0)


class SyntheticLabelledClips(Dataset):
    """Stands in for UcfFineTune / Kin400FTOfflineLMDB (datasets.py:952-1098): ``(clip [3,T,H,W] fp32, label)`` for
    data_type 'train'/'val' and ``(clips [n_clips,3,T,H,W], label)`` for 'test' (:993-1001).  Clips are a
    class-specific pattern plus uniform noise, so fine-tuning has something to learn; train/val/test draw from
    disjoint seeds."""

    def __init__(self, data_type="train", length=64, sample_duration=16, sample_size=112, n_classes=101, seed=1,
                 test_clips=3, noise=0.5):
        if data_type not in ("train", "val", "test"):
            raise ValueError("data_type %r" % (data_type,))
        self.data_type, self.length, self.t, self.hw = data_type, length, sample_duration, sample_size
        self.n_classes, self.seed, self.test_clips, self.noise = n_classes, seed, test_clips, noise
        self._salt = {"train": 11, "val": 23, "test": 37}[data_type]

    def __len__(self):
        return self.length

    def _clip(self, g, label):
        pat = _class_pattern(label, self.n_classes, self.t, self.hw)
        noise = torch.rand((3, self.t, self.hw, self.hw), generator=g) * 2 - 1
        return ((1 - self.noise) * pat + self.noise * noise).clamp_(-1, 1)

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed((self.seed * 1000003 + idx) * 101 + self._salt)
        label = int(torch.randint(0, self.n_classes, (1,), generator=g))
        if self.data_type == "test":
            return torch.stack([self._clip(g, label) for _ in range(self.test_clips)]), label
        return self._clip(g, label), label
