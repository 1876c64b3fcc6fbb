#!/usr/bin/env python3
"""Time the forward / data-gradient kernel of the R(2+1)D-18 layer shapes at the cfg2 batch under a pinned tile (or the
table's).  usage: ab_conv.py fwd|dgrad [tile=sp,mt,wm,tpb] [layer ...]   e.g.  ab_conv.py fwd tile=1,4,2,0 T1 T3"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cstp_amd import ops  # noqa: E402

LAYERS = {"S1": ((32, 64, 16, 56, 56), 144, (1, 3, 3), (1, 1, 1), (0, 1, 1)), "T1": ((32, 144, 16, 56, 56), 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
          "S3": ((32, 128, 8, 28, 28), 288, (1, 3, 3), (1, 1, 1), (0, 1, 1)), "T3": ((32, 288, 8, 28, 28), 128, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
          "S5": ((32, 256, 4, 14, 14), 576, (1, 3, 3), (1, 1, 1), (0, 1, 1)), "T5": ((32, 576, 4, 14, 14), 256, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
          "T0": ((32, 83, 16, 56, 56), 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)), "S2": ((32, 64, 16, 56, 56), 230, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
          "T2": ((32, 230, 16, 28, 28), 128, (3, 1, 1), (2, 1, 1), (1, 0, 0))}
mode = {"fwd": 0, "dgrad": 1}[sys.argv[1]]
PIN = [tuple(int(v) for v in a[5:].split(",")) for a in sys.argv[2:] if a.startswith("tile=")]
NAMES = [a for a in sys.argv[2:] if not a.startswith("tile=")]
lib = ops._lib.load()
st = torch.cuda.current_stream().cuda_stream
for name in (NAMES or list(LAYERS)):
    xs, k, ks, stride, pad = LAYERS[name]
    wsz = (k, xs[1]) + ks
    x = torch.randn(xs, device="cuda")
    w = torch.randn(wsz, device="cuda") * 0.05
    ys = ops.conv_out_shape(xs, wsz, stride, pad)
    desc = ops._desc(xs, wsz, stride, pad)
    wsb = torch.empty(lib.cstp_conv3d_workspace_bytes(ctypes.byref(desc)), dtype=torch.uint8, device="cuda")
    src = x if mode == 0 else torch.randn(ys, device="cuda")
    out = torch.empty(ys if mode == 0 else xs, device="cuda")
    if PIN:
        ops.set_conv_tile(xs, wsz, stride, pad, mode, PIN[0])
    else:
        ops._autotune(lib, desc, mode, src, w, out, wsb)
    cell = src.abs().max().view(torch.int32).clone()
    if mode == 0:
        fn = lambda: ops.check(lib.cstp_conv3d_forward_am(st, ctypes.byref(desc), src.data_ptr(), w.data_ptr(), None, None, out.data_ptr(),
                                                          wsb.data_ptr(), wsb.numel(), cell.data_ptr()), "fwd")
    else:
        fn = lambda: ops.check(lib.cstp_conv3d_backward_data_am(st, ctypes.byref(desc), src.data_ptr(), w.data_ptr(), out.data_ptr(),
                                                                wsb.data_ptr(), wsb.numel(), cell.data_ptr()), "dgrad")
    for _ in range(3):
        fn()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) / 10)
    t = (ctypes.c_int32 * 4)()
    lib.cstp_conv3d_query_tile(ctypes.byref(desc), mode, t)
    gf = 2.0 * ys[0] * ys[2] * ys[3] * ys[4] * k * xs[1] * ks[0] * ks[1] * ks[2] / 1e9
    print("%s %s tile %s: min %.3f ms  med %.3f ms  %.1f TF/s (incl. the weight pack)" % (name, sys.argv[1], list(t), min(ts), sorted(ts)[2], gf / min(ts)))
