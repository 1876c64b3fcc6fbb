#!/usr/bin/env python3
"""Per-shape timing of every convolution of the R(2+1)D encoder (forward, data-grad, weight-grad)
through the C ABI, HIP events on the launch stream.  Prints achieved TFLOP/s against the f32 peak
and the algorithmic HBM bytes (4*(N_in + N_out) + 4*N_w per call)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cstp_amd import ops  # noqa: E402

PEAK = 157.3


def shapes(b, t=16, hw=112, depth=18):
    ls = {1: (1, 1, 1, 1), 18: (2, 2, 2, 2), 34: (3, 4, 6, 3)}[depth]
    out = []

    def stconv(name, cin, cout, k, stride, pad, d, h, cnt):
        m = (k[0] * k[1] * k[2] * cin * cout) // (k[1] * k[2] * cin + k[0] * cout)
        out.append((name + ".S", (b, cin, d, h, h), m, (1, k[1], k[2]), (1, stride[1], stride[2]), (0, pad[1], pad[2]), cnt))
        h2 = (h + 2 * pad[1] - k[1]) // stride[1] + 1
        out.append((name + ".T", (b, m, d, h2, h2), cout, (k[0], 1, 1), (stride[0], 1, 1), (pad[0], 0, 0), cnt))
        return (d + 2 * pad[0] - k[0]) // stride[0] + 1, h2

    d, h = stconv("stem", 3, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3), t, hw, 1)
    chans = [(64, 64, False), (64, 128, True), (128, 256, True), (256, 512, True)]
    for li, ((cin, cout, ds), n) in enumerate(zip(chans, ls)):
        s = (2, 2, 2) if ds else (1, 1, 1)
        if ds:
            stconv("c%d.short" % (li + 2), cin, cout, (1, 1, 1), (2, 2, 2), (0, 0, 0), d, h, 1)
        d2, h2 = stconv("c%d.b1c1" % (li + 2), cin, cout, (3, 3, 3), s, (1, 1, 1), d, h, 1)
        stconv("c%d.same" % (li + 2), cout, cout, (3, 3, 3), (1, 1, 1), (1, 1, 1), d2, h2, 2 * n - 1)
        d, h = d2, h2
    return out


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--depth", type=int, default=18)
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    ideal = 0.0
    print("%-12s %-26s %5s %-9s %8s | %8s %6s | %8s %6s | %8s %6s" % ("layer", "input", "k", "kernel", "GFLOP", "fwd_ms", "TF/s",
                                                                       "dgrad_ms", "TF/s", "wgrad_ms", "TF/s"))
    for name, xs, k, ks, st, pd, cnt in shapes(args.batch, depth=args.depth):
        x = torch.randn(xs, device="cuda")
        w = torch.randn((k, xs[1]) + ks, device="cuda") * 0.05
        y = ops.conv3d(x, w, None, st, pd)
        dy = torch.randn_like(y)
        gf = 2.0 * y.numel() * xs[1] * ks[0] * ks[1] * ks[2] / 1e9
        xr = x.clone().requires_grad_(True)
        wr = w.clone().requires_grad_(True)

        t_f = timeit(lambda: ops.conv3d(x, w, None, st, pd), args.iters)

        def dgrad():
            yy = ops.conv3d(xr, w, None, st, pd)
            yy.backward(dy)

        def wgrad():
            yy = ops.conv3d(x, wr, None, st, pd)
            yy.backward(dy)

        t_d = timeit(dgrad, args.iters) - t_f if name != "stem.S" else 0.0
        t_w = timeit(wgrad, args.iters) - t_f
        print("%-12s %-26s %5d %-9s %8.1f | %8.3f %6.1f | %8.3f %6.1f | %8.3f %6.1f   x%d" % (
            name, "x".join(map(str, xs)), k, "x".join(map(str, ks)), gf, t_f, gf / t_f, t_d, gf / max(t_d, 1e-9), t_w,
            gf / max(t_w, 1e-9), cnt))
        tot["fwd"] += t_f * cnt
        tot["dgrad"] += t_d * cnt
        tot["wgrad"] += t_w * cnt
        ideal += gf * cnt / PEAK
    print("per encoder pass: fwd %.2f ms, dgrad %.2f ms, wgrad %.2f ms; ideal at %.1f TF/s: %.2f ms each"
          % (tot["fwd"], tot["dgrad"], tot["wgrad"], PEAK, ideal))
    print("per training step (4 fwd + 2 dgrad + 2 wgrad): %.1f ms conv time; ideal %.1f ms"
          % (4 * tot["fwd"] + 2 * tot["dgrad"] + 2 * tot["wgrad"], 8 * ideal))


if __name__ == "__main__":
    main()
