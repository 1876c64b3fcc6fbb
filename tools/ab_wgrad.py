#!/usr/bin/env python3
"""Time the weight-gradient kernel the library picks (persisted table / pinned tile) on the layer shapes of R(2+1)D-18 at the
cfg2 batch -- for A/B-ing library variants (CSTP_LIB_PATH).  usage: ab_wgrad.py [tile=sp,mt,blocks,0] [layer ...]
(tile=2,9,1,0 pins the LDS-resident-x kernel igemm_k2p, tile=1,9,8,0 the gather kernel igemm_k2s)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cstp_amd import ops  # noqa: E402

LAYERS = {"S1": ((32, 64, 16, 56, 56), 144, (1, 3, 3), (1, 1, 1), (0, 1, 1)), "T1": ((32, 144, 16, 56, 56), 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
          "S3": ((32, 128, 8, 28, 28), 288, (1, 3, 3), (1, 1, 1), (0, 1, 1)), "T3": ((32, 288, 8, 28, 28), 128, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
          "S5": ((32, 256, 4, 14, 14), 576, (1, 3, 3), (1, 1, 1), (0, 1, 1)), "T5": ((32, 576, 4, 14, 14), 256, (3, 1, 1), (1, 1, 1), (1, 0, 0))}
lib = ops._lib.load()
st = torch.cuda.current_stream().cuda_stream
PIN = [tuple(int(v) for v in a[5:].split(",")) for a in sys.argv[1:] if a.startswith("tile=")]
NAMES = [a for a in sys.argv[1:] if not a.startswith("tile=")]
for name in (NAMES or list(LAYERS)):
    xs, k, ks, stride, pad = LAYERS[name]
    ws = (k, xs[1]) + ks
    x = torch.randn(xs, device="cuda")
    dy = torch.randn(ops.conv_out_shape(xs, ws, stride, pad), device="cuda")
    dw = torch.empty(ws, device="cuda")
    desc = ops._desc(xs, ws, stride, pad)
    wsb = torch.empty(lib.cstp_conv3d_workspace_bytes(ctypes.byref(desc)), dtype=torch.uint8, device="cuda")
    if PIN:
        ops.set_conv_tile(xs, ws, stride, pad, 2, PIN[0])
    else:
        ops._autotune(lib, desc, 2, x, dy, dw, wsb)
    xc, dc = x.abs().max().view(torch.int32).clone(), dy.abs().max().view(torch.int32).clone()
    fn = lambda: ops.check(lib.cstp_conv3d_backward_weight_am(st, ctypes.byref(desc), x.data_ptr(), None, dy.data_ptr(), dw.data_ptr(),
                                                              wsb.data_ptr(), wsb.numel(), xc.data_ptr(), dc.data_ptr()), "wgrad")
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
    lib.cstp_conv3d_query_tile(ctypes.byref(desc), 2, t)
    gf = 2.0 * xs[0] * dy.shape[2] * dy.shape[3] * dy.shape[4] * k * xs[1] * ks[0] * ks[1] * ks[2] / 1e9
    print("%s wgrad tile %s: min %.3f ms  med %.3f ms  %.1f TF/s" % (name, list(t), min(ts), sorted(ts)[2], gf / min(ts)))
