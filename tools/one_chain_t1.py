#!/usr/bin/env python3
"""The first stage's (2+1)D chain AS THE TRAINING STEP RUNS IT, a few times (target for rocprofv3 --pmc / --kernel-trace):
spatial conv S1 (sums + range for the BatchNorm behind it) -> temporal conv T1 with that BatchNorm + ReLU inside (in_affine) and
the next BatchNorm's sums from its epilogue -> backward of both.  Kernels it launches at the cfg2 size (32 clips of 16x56x56):
igemm_k1p<9, true> (S1 forward), igemm_k1w<true, true> (T1 forward), igemm_k1t<9, false, false> (T1 data gradient),
igemm_k2t<true> (T1 weight gradient), igemm_k1p<4, false> (S1 data gradient), igemm_k2p (S1 weight gradient), bn kernels.
usage: one_chain_t1.py [batch=32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cstp_amd import ops  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
xs, mid, k = (b, 64, 16, 56, 56), 144, 64
g = torch.Generator().manual_seed(1)
x = (torch.rand(xs, generator=g) * 2 - 1).cuda().requires_grad_(True)
w_s = ((torch.rand((mid, xs[1], 1, 3, 3), generator=g) * 2 - 1) * 0.05).cuda().requires_grad_(True)
w_t = ((torch.rand((k, mid, 3, 1, 1), generator=g) * 2 - 1) * 0.05).cuda().requires_grad_(True)
gamma, beta = (torch.rand(mid, generator=g) + 0.5).cuda().requires_grad_(True), (torch.randn(mid, generator=g) * 0.1).cuda().requires_grad_(True)
dy = None
for _ in range(4):
    rm, rv, rm2 = torch.zeros(mid, device="cuda"), torch.ones(mid, device="cuda"), torch.zeros(k, device="cuda")
    y = ops.conv3d(x, w_s, None, 1, (0, 1, 1), bn_groups=2, bn_pivot=rm)
    out = ops.bn_relu_conv3d(y, gamma, beta, rm, rv, w_t, 1, (1, 0, 0), 2, True, bn_groups=2, bn_pivot=rm2)
    if dy is None:
        dy = torch.rand(out.shape, device="cuda") * 2 - 1
        ops._tag_absmax(dy, dy.abs().max().view(torch.int32).clone())
    out.backward(dy)
    ops._join_side_streams()
torch.cuda.synchronize()
print("done chain", b)
