#!/usr/bin/env python3
"""Weight-gradient timing per layer shape via the C ABI directly (no autograd accumulation in the way)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_convs import shapes, timeit  # noqa: E402
from cstp_amd import _lib, ops  # noqa: E402

lib = _lib.load()
tot = 0.0
for name, xs, k, ks, st, pd, cnt in shapes(32):
    x = torch.randn(xs, device="cuda")
    w = torch.randn((k, xs[1]) + ks, device="cuda") * 0.05
    desc = ops._desc(x.shape, w.shape, st, pd)
    y = torch.empty(ops.conv_out_shape(x.shape, w.shape, st, pd), device="cuda")
    dy = torch.randn_like(y)
    dw = torch.empty_like(w)
    ws = torch.empty(lib.cstp_conv3d_workspace_bytes(ctypes.byref(desc)), dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream

    def f():
        _lib.check(lib.cstp_conv3d_backward_weight(s, ctypes.byref(desc), x.data_ptr(), None, dy.data_ptr(), dw.data_ptr(),
                                                   ws.data_ptr(), ws.numel()), "wgrad")
    t = timeit(f, 5)
    gf = 2.0 * y.numel() * xs[1] * ks[0] * ks[1] * ks[2] / 1e9
    tot += t * cnt
    if len(sys.argv) > 1:
        print("%-12s %8.3f ms %6.1f TF  x%d" % (name, t, gf / t, cnt))
print("wgrad per encoder pass: %.2f ms" % tot)
