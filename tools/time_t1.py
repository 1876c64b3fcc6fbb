#!/usr/bin/env python3
"""T1 temporal forward (144 -> 64 channels, 3x1x1 at 16x56x56, 32 clips) AS THE TRAINING STEP ISSUES IT: input = the spatial
convolution's raw output, the BatchNorm + ReLU in front applied inside the launch (in_affine), sums / range for the BatchNorm
behind from its epilogue where the kernel can -- per pinned forward tile.  usage: time_t1.py [tile ...]   default: gather, ring, resident"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cstp_amd import ops  # noqa: E402

tiles = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(1, 4, 0, 0), (2, 4, 0, 0), (2, 4, 2, 0)]
xs, mid, k = (32, 64, 16, 56, 56), 144, 64
g = torch.Generator().manual_seed(1)
x = (torch.rand(xs, generator=g) * 2 - 1).cuda()
w_s = ((torch.rand((mid, xs[1], 1, 3, 3), generator=g) * 2 - 1) * 0.05).cuda()
w_t = ((torch.rand((k, mid, 3, 1, 1), generator=g) * 2 - 1) * 0.05).cuda()
gamma, beta = (torch.rand(mid, generator=g) + 0.5).cuda(), (torch.randn(mid, generator=g) * 0.1).cuda()
ys = (xs[0], mid) + xs[2:]
ref = None
for tile in tiles:
    ops.set_conv_tile(ys, tuple(w_t.shape), (1, 1, 1), (1, 0, 0), 0, tile)
    ts = []
    for rep in range(6):
        rm, rv = torch.zeros(mid, device="cuda"), torch.ones(mid, device="cuda")
        rm2 = torch.zeros(k, device="cuda")
        with torch.no_grad():
            y = ops.conv3d(x, w_s, None, 1, (0, 1, 1), bn_groups=2, bn_pivot=rm)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = ops.bn_relu_conv3d(y, gamma, beta, rm, rv, w_t, 1, (1, 0, 0), 2, True, bn_groups=2, bn_pivot=rm2)
            b.record()
            b.synchronize()
        ts.append(a.elapsed_time(b))
    st = getattr(out, "_cstp_bnstats", None)
    if ref is None:
        ref = out.clone()
    print("T1 fwd with in_affine, tile %s: min %.3f ms med %.3f ms (finalize + pack + conv); BatchNorm sums from the epilogue: %s; "
          "max |diff| to the first tile %.3g" % (tile, min(ts[1:]), sorted(ts[1:])[2], st is not None,
                                               float((out - ref).abs().max())), flush=True)
