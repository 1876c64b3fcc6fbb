#!/usr/bin/env python3
"""A/B the LDS-resident-patch kernel (igemm_k1p) against the gather kernels on the stride-1 3x3 layer shapes of R(2+1)D-18 at
the cfg2 batch (32 clips per launch): forward and data gradient, interleaved rounds in one process.
usage: ab_patch.py [layer ...]   layers: S1 S3 S5 S7"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cstp_amd import ops  # noqa: E402

LAYERS = {"S1": ((32, 64, 16, 56, 56), 144), "S3": ((32, 128, 8, 28, 28), 288), "S5": ((32, 256, 4, 14, 14), 576),
          "S7": ((32, 512, 2, 7, 7), 1152)}
VARIANTS = {
    0: {"gather 144x128": (1, 9, 0, 0), "gather 144x256": (1, 9, 2, 0), "patch 144": (2, 9, 0, 0), "patch 128": (2, 8, 0, 0)},
    1: {"gather 64x128": (1, 4, 0, 0), "gather 128x128": (1, 8, 0, 0), "patch 64": (2, 4, 0, 0), "patch 128": (2, 8, 0, 0)},
}


def timeit(fn, n):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n


for name in (sys.argv[1:] or list(LAYERS)):
    xs, k = LAYERS[name]
    x = torch.randn(xs, device="cuda")
    w = torch.randn(k, xs[1], 1, 3, 3, device="cuda") * 0.05
    ws = (k, xs[1], 1, 3, 3)
    gf = 2.0 * xs[0] * xs[2] * xs[3] * xs[4] * k * xs[1] * 9 / 1e9
    lib = ops._lib.load()
    import ctypes
    desc = ops._desc(xs, ws, (1, 1, 1), (0, 1, 1))
    y = torch.empty((xs[0], k) + xs[2:], device="cuda")
    dy = torch.randn_like(y)
    dx = torch.empty_like(x)
    wsb = torch.empty(lib.cstp_conv3d_workspace_bytes(ctypes.byref(desc)), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for mode in (0, 1):
        res = {}
        for rnd in range(3):
            for vn, tile in VARIANTS[mode].items():
                if mode == 1 and 16 * tile[1] > xs[1] + 16 * tile[1] // 2:
                    continue
                ops.set_conv_tile(xs, ws, (1, 1, 1), (0, 1, 1), mode, tile)
                if mode == 0:
                    fn = lambda: ops.check(lib.cstp_conv3d_forward(st, ctypes.byref(desc), x.data_ptr(), w.data_ptr(), None, None,
                                                                   y.data_ptr(), wsb.data_ptr(), wsb.numel()), "fwd")
                else:
                    fn = lambda: ops.check(lib.cstp_conv3d_backward_data(st, ctypes.byref(desc), dy.data_ptr(), w.data_ptr(),
                                                                         dx.data_ptr(), wsb.data_ptr(), wsb.numel()), "dgrad")
                res.setdefault(vn, []).append(timeit(fn, 10))
        for vn, ts in res.items():
            t = min(ts)
            print("%-3s %-5s %-16s min %.3f ms  med %.3f ms  %.1f TF/s (incl. pack + absmax)"
                  % (name, "fwd" if mode == 0 else "dgrad", vn, t, sorted(ts)[len(ts) // 2], gf / t), flush=True)
