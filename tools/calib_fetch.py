#!/usr/bin/env python3
"""FETCH_SIZE calibration on a known byte count with 4-byte-per-lane coalesced loads (the access
width of the conv gathers): avgpool over a 1 GiB tensor reads each byte exactly once."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cstp_amd import ops  # noqa: E402

x = torch.rand(4096, 64, 1024, device="cuda")   # 1 GiB, rows of 4 KiB, one wave per row, 4 B per lane
for _ in range(3):
    y = ops.global_avg_pool(x.view(4096, 64, 1, 32, 32))
torch.cuda.synchronize()
print("bytes", x.numel() * 4)
