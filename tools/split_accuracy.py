#!/usr/bin/env python3
"""Accuracy of the convolution kernels against fp64 (PyTorch CPU) for the native-f32 tiles and the 3xbf16-split tiles:
max-abs-diff / max-abs-ref and rms-diff / rms-ref, forward, data gradient and weight gradient.
usage: [CSTP_TILE=s9] [CSTP_WTILE=s9,8] split_accuracy.py      (unset = native f32 MFMA kernels)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cstp_amd import _lib, ops  # noqa: E402

SHAPES = [((2, 64, 8, 28, 28), 144, (1, 3, 3), (1, 1, 1), (0, 1, 1)), ((2, 144, 8, 28, 28), 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
          ((2, 256, 4, 14, 14), 576, (1, 3, 3), (1, 1, 1), (0, 1, 1)), ((4, 1152, 2, 7, 7), 512, (3, 1, 1), (1, 1, 1), (1, 0, 0))]
lib = _lib.load()
torch.manual_seed(0)
for xs, k, ks, st, pd in SHAPES:
    x = torch.randn(xs, dtype=torch.float64)
    w = torch.randn((k, xs[1]) + ks, dtype=torch.float64) * (2.0 / (xs[1] * ks[0] * ks[1] * ks[2])) ** 0.5
    xr = x.clone().requires_grad_(True)
    y = torch.nn.functional.conv3d(xr, w, None, st, pd)
    dy = torch.randn_like(y)
    y.backward(dy)
    xd, wd, dyd = x.float().cuda(), w.float().cuda(), dy.float().cuda()
    desc = ops._desc(xs, wd.shape, st, pd)
    ws = torch.empty(lib.cstp_conv3d_workspace_bytes(ctypes.byref(desc)), dtype=torch.uint8, device="cuda")
    yg = torch.empty(y.shape, device="cuda")
    dxg = torch.empty(xs, device="cuda")
    dwg = torch.empty(wd.shape, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.cstp_conv3d_forward(s, ctypes.byref(desc), xd.data_ptr(), wd.data_ptr(), None, None, yg.data_ptr(), ws.data_ptr(),
                                       ws.numel()), "fwd")
    _lib.check(lib.cstp_conv3d_backward_data(s, ctypes.byref(desc), dyd.data_ptr(), wd.data_ptr(), dxg.data_ptr(), ws.data_ptr(),
                                             ws.numel()), "dgrad")
    _lib.check(lib.cstp_conv3d_backward_weight(s, ctypes.byref(desc), xd.data_ptr(), None, dyd.data_ptr(), dwg.data_ptr(),
                                               ws.data_ptr(), ws.numel()), "wgrad")
    # the fp32 rounding of the INPUTS is common to both paths: compare against fp64 on the rounded inputs
    xr2 = xd.double().cpu().requires_grad_(True)
    wr2 = wd.double().cpu().requires_grad_(True)
    y2 = torch.nn.functional.conv3d(xr2, wr2, None, st, pd)
    y2.backward(dyd.double().cpu())

    def errs(a, b):
        d = a.double().cpu() - b
        return float(d.abs().max() / b.abs().max()), float(d.pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
    print("K=%5d M=%4d  fwd max %.2e rms %.2e | dgrad max %.2e rms %.2e | wgrad max %.2e rms %.2e" % (
        (xs[1] * ks[0] * ks[1] * ks[2], k) + errs(yg, y2.detach()) + errs(dxg, xr2.grad) + errs(dwg, wr2.grad)))
