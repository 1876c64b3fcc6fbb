#!/usr/bin/env python3
"""Run ONE convolution shape a few times (target for rocprofv3 --pmc passes).
usage: one_conv.py [fwd|dgrad|wgrad] [batch]   -- the S1 shape 64->144 1x3x3 @16x56x56"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cstp_amd import ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
b = int(sys.argv[2]) if len(sys.argv) > 2 else 32
x = torch.rand(b, 64, 16, 56, 56, device="cuda") * 2 - 1
w = (torch.rand(144, 64, 1, 3, 3, device="cuda") * 2 - 1) * 0.05
if mode == "dgrad":
    x.requires_grad_(True)
if mode == "wgrad":
    w.requires_grad_(True)
for _ in range(4):
    # forward as the training step issues it: the BatchNorm behind the layer takes its sums from this launch (two view groups)
    y = ops.conv3d(x, w, None, 1, (0, 1, 1), bn_groups=2 if mode == "fwd" else 0)
    if mode != "fwd":
        y.backward(torch.ones_like(y))
torch.cuda.synchronize()
print("done", mode, b)
