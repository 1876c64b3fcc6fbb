#!/usr/bin/env python3
"""Run ONE bf16-storage convolution of the 3D-ResNet-50 step a few times (target for rocprofv3 --pmc passes), forward + backward.
usage: one_conv_b16.py [layer]   layers (cfg5 share: 8 clips of 3x16x224x224 after the stem / max-pool):
   L1_3x3x3  64 -> 64 3x3x3 @ 8x56x56      L1_pw_in 256 -> 64 1x1x1 @ 8x56x56      L1_pw_out 64 -> 256 1x1x1 @ 8x56x56
   L2_3x3x3 128 -> 128 3x3x3 @ 4x28x28     stem 3 -> 64 7x7x7 stride (1,2,2) @ 16x224x224"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cstp_amd import ops  # noqa: E402

LAYERS = {"L1_3x3x3": ((8, 64, 8, 56, 56), 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
          "L1_pw_in": ((8, 256, 8, 56, 56), 64, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
          "L1_pw_out": ((8, 64, 8, 56, 56), 256, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
          "L2_3x3x3": ((8, 128, 4, 28, 28), 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
          "stem": ((8, 3, 16, 224, 224), 64, (7, 7, 7), (1, 2, 2), (3, 3, 3))}
name = sys.argv[1] if len(sys.argv) > 1 else "L1_3x3x3"
xs, k, ks, st, pad = LAYERS[name]
x = (torch.rand(xs, device="cuda") * 2 - 1).to(torch.bfloat16).requires_grad_(xs[1] > 3)
w = ((torch.rand((k, xs[1]) + ks, device="cuda") * 2 - 1) * 0.05).requires_grad_(True)
for _ in range(4):
    y = ops._Conv3dB16.apply(x, w, st, pad)
    y.backward(torch.ones_like(y))
    ops._join_side_streams()
torch.cuda.synchronize()
gf = 2.0 * y.numel() * xs[1] * ks[0] * ks[1] * ks[2] / 1e9
nbytes = 2 * (x.numel() + y.numel()) + 4 * w.numel()
print("done", name, "algorithmic per launch: %.2f GFLOP, %.1f MB" % (gf, nbytes / 1e6))
