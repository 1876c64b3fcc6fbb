#!/usr/bin/env python3
"""Time cstp_conv3d_forward / cstp_conv3d_backward_data per R(2+1)D layer shape through the C ABI (no autograd),
under whatever tile the environment forces (CSTP_TILE=..., CSTP_AUTOTUNE is irrelevant here: no tuning call is made),
and check the result against the default tile's.   usage: time_k1.py [--batch 32] [--only S1]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cstp_amd import _lib, ops  # noqa: E402
from tools.bench_convs import shapes  # noqa: E402


def run(lib, mode, desc, src, w, out, ws, iters):
    s = torch.cuda.current_stream().cuda_stream

    def call():
        if mode == 0:
            rc = lib.cstp_conv3d_forward(s, ctypes.byref(desc), src.data_ptr(), w.data_ptr(), None, None, out.data_ptr(),
                                         ws.data_ptr(), ws.numel())
        else:
            rc = lib.cstp_conv3d_backward_data(s, ctypes.byref(desc), src.data_ptr(), w.data_ptr(), out.data_ptr(),
                                               ws.data_ptr(), ws.numel())
        _lib.check(rc, "conv")
    call()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        call()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    lib = _lib.load()
    tot = [0.0, 0.0]
    for name, xs, k, ks, st, pd, cnt in shapes(a.batch):
        if a.only and a.only not in name:
            continue
        if xs[1] < 8:
            continue
        x = torch.rand(xs, device="cuda") * 2 - 1
        if os.environ.get("CSTP_TIME_ZERO"):         # operand of zeros: what the kernel does at the clock an idle data path allows
            x.zero_()
        w = (torch.rand((k, xs[1]) + ks, device="cuda") * 2 - 1) * 0.05
        desc = ops._desc(xs, w.shape, st, pd)
        y = torch.empty(ops.conv_out_shape(xs, w.shape, st, pd), device="cuda")
        dy = torch.rand_like(y) * 2 - 1
        dx = torch.empty_like(x)
        ws = torch.empty(lib.cstp_conv3d_workspace_bytes(ctypes.byref(desc)), dtype=torch.uint8, device="cuda")
        gf = 2.0 * y.numel() * xs[1] * ks[0] * ks[1] * ks[2] / 1e9
        tf = run(lib, 0, desc, x, w, y, ws, a.iters)
        td = run(lib, 1, desc, dy, w, dx, ws, a.iters)
        ref_y = torch.nn.functional.conv3d(x.double().cpu(), w.double().cpu(), None, st, pd) if y.numel() < 3e6 else None
        err = float((y.cpu().double() - ref_y).abs().max() / ref_y.abs().max()) if ref_y is not None else float("nan")
        tot[0] += tf * cnt
        tot[1] += td * cnt
        print("%-10s M=%4d K=%5d  %7.1f GF x%d | fwd %7.3f ms %6.1f TF/s | dgrad %7.3f ms %6.1f TF/s | err %.1e  cs %.6e %.6e" % (
            name, k, xs[1] * ks[0] * ks[1] * ks[2], gf, cnt, tf, gf / tf, td, gf / td, err, float(y.double().abs().sum()),
            float(dx.double().abs().sum())), flush=True)
    print("total per encoder pass: fwd %.2f ms, dgrad %.2f ms" % (tot[0], tot[1]))


if __name__ == "__main__":
    main()
