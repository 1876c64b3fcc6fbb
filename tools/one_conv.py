#!/usr/bin/env python3
"""Run ONE convolution shape a few times (target for rocprofv3 --pmc passes).
usage: one_conv.py [fwd|dgrad|wgrad] [batch] [layer] [tile=sp,mt,x,y]
   layer: S1 (default, 64->144 1x3x3 @16x56x56), T1 (144->64 3x1x1 @16x56x56), S3, T3, S5, T5
   tile : pin the kernel variant of the timed direction (cstp_conv3d_set_tile), e.g. tile=2,9,1,0 = igemm_k2p for wgrad"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cstp_amd import ops  # noqa: E402

LAYERS = {"S1": ((64, 16, 56, 56), 144, (1, 3, 3), (0, 1, 1)), "T1": ((144, 16, 56, 56), 64, (3, 1, 1), (1, 0, 0)),
          "S3": ((128, 8, 28, 28), 288, (1, 3, 3), (0, 1, 1)), "T3": ((288, 8, 28, 28), 128, (3, 1, 1), (1, 0, 0)),
          "S5": ((256, 4, 14, 14), 576, (1, 3, 3), (0, 1, 1)), "T5": ((576, 4, 14, 14), 256, (3, 1, 1), (1, 0, 0))}
args = [a for a in sys.argv[1:] if not a.startswith("tile=")]
pin = [tuple(int(v) for v in a[5:].split(",")) for a in sys.argv[1:] if a.startswith("tile=")]
mode = args[0] if len(args) > 0 else "fwd"
b = int(args[1]) if len(args) > 1 else 32
layer = args[2] if len(args) > 2 else "S1"
(c, d, h, wd), k, ks, pad = LAYERS[layer]
x = torch.rand(b, c, d, h, wd, device="cuda") * 2 - 1
w = (torch.rand((k, c) + ks, device="cuda") * 2 - 1) * 0.05
if pin:
    ops.set_conv_tile(tuple(x.shape), tuple(w.shape), (1, 1, 1), pad, {"fwd": 0, "dgrad": 1, "wgrad": 2}[mode], pin[0])
if mode == "dgrad":
    x.requires_grad_(True)
if mode == "wgrad":
    w.requires_grad_(True)
for _ in range(4):
    # forward as the training step issues it: the BatchNorm behind the layer takes its sums from this launch (two view groups)
    y = ops.conv3d(x, w, None, 1, pad, bn_groups=2 if (mode == "fwd" and ks[0] == 1) else 0)
    if mode != "fwd":
        y.backward(torch.ones_like(y))
torch.cuda.synchronize()
print("done", mode, b, layer)
