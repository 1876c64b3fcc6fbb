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
