#!/usr/bin/env python3
"""Pre-training throughput of the 3D-ResNet-BYOL wrapper on one MI355X (synthetic clips resident in HBM): the per-GPU share of
BASELINE.json configs[4] (B = 32 over 8 GPUs -> 4 clip pairs per GPU, 3x16x224x224) on the BasicBlock depths the reference can
can run (10 / 18 / 34) and on the Bottleneck depth 50 that configs[4] names (corrected wrapper: cstp_amd/r3d_byol.py), with fp32 or
bf16 activation storage (--act_dtype; configs[4] says bf16).

    python tools/bench_r3d.py --depth 50 --batch 4 --size 224 --steps 10 --act_dtype bf16

Not the headline metric (bench.py is) -- a sizing aid for the backbone swap.  Like bench.py the line carries a ``roofline``
block: every C-ABI call of two extra steps (stream overlaps off, so an event pair brackets one kernel chain running alone) is
timed with HIP events on its launch stream and summed per layer CLASS (3x3x3 / 1x1x1 / stem convolutions by direction,
BatchNorm forward / backward) next to the class's ALGORITHMIC work -- 2 * k * c * taps FLOP per output position; bytes = every
operand touched once at its storage width (bf16 activations 2 B, fp32 weights / statistics 4 B) -- its fraction of the dense
bf16 MFMA peak (2 516.6 TFLOP/s: bf16 storage multiplies bf16 operands directly, one product per product) or of the f16-pair
ceiling (fp32 storage), and of the 8 TB/s HBM peak; ``roofline`` itself is the class that takes the most time."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cstp_amd.optim import FlatSGD  # noqa: E402
from cstp_amd.r3d_byol import R3DBYOL  # noqa: E402
from cstp_amd.synthetic import device_batch  # noqa: E402
from cstp_amd.train import PretrainStep  # noqa: E402


HBM_PEAK_TBS, BF16_PEAK, F32_PEAK = 8.0, 2516.6, 157.3        # MI355X_MICROARCH.md


class AllTimers:
    """cstp_amd.ops.kernel_timer hook: a HIP-event pair on the launch stream around EVERY spanned C-ABI call."""

    def __init__(self):
        self.enabled = False
        self.pairs = {}

    class _Span:
        def __init__(self, sink):
            self.sink = sink

        def __enter__(self):
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()
            return self

        def __exit__(self, *exc):
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            self.sink.append((self.a, b))
            return False

    def span(self, what, key):
        if not self.enabled:
            return None
        return AllTimers._Span(self.pairs.setdefault((what, key), []))


def classify(what, key):
    """(class name, algorithmic FLOP, algorithmic bytes) of one spanned call."""
    b16 = "bf16" in key
    k = tuple(v for v in key if v != "bf16")
    ab = 2 if b16 else 4                              # activation storage width
    if what.startswith("conv3d"):
        n, c, d, h, w, ko, kt, kh, kw, st, sh, sw, pt, ph, pw = k
        do, ho, wo = (d + 2 * pt - kt) // st + 1, (h + 2 * ph - kh) // sh + 1, (w + 2 * pw - kw) // sw + 1
        pos = n * do * ho * wo
        taps = kt * kh * kw
        flop = 2.0 * pos * ko * c * taps
        nbytes = ab * (n * c * d * h * w + pos * ko) + 4.0 * ko * c * taps
        shape = "stem %dx%dx%d" % (kt, kh, kw) if c < 8 else ("1x1x1" if taps == 1 else "%dx%dx%d" % (kt, kh, kw))
        if d * h * w == 1:
            shape = "linear"
        direction = {"conv3d_forward": "fwd", "conv3d_backward_data": "dgrad", "conv3d_backward_weight": "wgrad"}[what]
        return "conv %s %s" % (shape, direction), flop, nbytes
    n, c, s = k[0], k[1], k[2]
    res = bool(k[4])
    if what == "bn_forward":
        return "BatchNorm fwd", 0.0, (3 * ab + (ab if res else 0)) * float(n) * c * s      # x twice (statistics, apply), y once
    return "BatchNorm bwd", 0.0, (5 * ab + (ab if res else 0)) * float(n) * c * s          # (x, dy) twice, dx once


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=18)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--act_dtype", default="fp32", choices=("fp32", "bf16"))
    a = ap.parse_args()
    torch.manual_seed(1)
    dev = torch.device("cuda", 0)
    opts = argparse.Namespace(model_depth=a.depth, sample_size=a.size, sample_duration=a.frames, sc_type="B", n_classes=400, act_dtype=a.act_dtype)
    model = R3DBYOL(pretrain=True, opts=opts).cuda()
    arenas = model.flatten_parameters()
    model.train()
    opt = FlatSGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4, arenas=arenas)
    step = PretrainStep(model, opt, (0.1, 1.0, 1.0, 1.0, 1.0), clip_grad_norm=True)
    x1, x2, lab = device_batch(a.batch, a.frames, a.size, dev, seed=1)

    def run(n):
        for _ in range(n):
            step(x1, x2, lab["spa"], lab["tem"], lab["pb"], lab["rot1"], lab["rot2"]).to_host()
    run(1 + a.warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.steps)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3

    # ---- per-class roofline table: two more steps, stream overlaps off, every spanned call under a HIP-event pair
    from cstp_amd import ops, r3d_byol as _r3d
    tm = AllTimers()
    ops.kernel_timer = tm
    saved = (ops.OVERLAP_WGRAD, getattr(_r3d, "OVERLAP_TARGET_FORWARD", None))
    ops.OVERLAP_WGRAD = False
    if saved[1] is not None:
        _r3d.OVERLAP_TARGET_FORWARD = False
    try:
        from cstp_amd import r21d_byol as _rb
        rb_saved = _rb.OVERLAP_TARGET_FORWARD
        _rb.OVERLAP_TARGET_FORWARD = False
    except Exception:
        _rb, rb_saved = None, None
    tm.enabled = True
    nrep = 2
    run(nrep)
    torch.cuda.synchronize()
    tm.enabled = False
    ops.kernel_timer = None
    ops.OVERLAP_WGRAD = saved[0]
    if saved[1] is not None:
        _r3d.OVERLAP_TARGET_FORWARD = saved[1]
    if _rb is not None:
        _rb.OVERLAP_TARGET_FORWARD = rb_saved
    classes = {}
    for (what, key), pairs in tm.pairs.items():
        name, flop, nbytes = classify(what, key)
        c = classes.setdefault(name, {"calls": 0, "ms": 0.0, "gflop": 0.0, "bytes": 0.0})
        t = sum(x.elapsed_time(y) for x, y in pairs)
        c["calls"] += len(pairs); c["ms"] += t; c["gflop"] += flop * len(pairs) / 1e9; c["bytes"] += nbytes * len(pairs)
    peak = BF16_PEAK if a.act_dtype == "bf16" else BF16_PEAK / 3
    rows = []
    for name, c in sorted(classes.items(), key=lambda kv: -kv[1]["ms"]):
        per = 1.0 / nrep
        row = {"class": name, "calls_per_step": c["calls"] * per, "ms_per_step": round(c["ms"] * per, 3),
               "algorithmic_gflop_per_step": round(c["gflop"] * per, 1), "algorithmic_GB_per_step": round(c["bytes"] * per / 1e9, 3)}
        sec = c["ms"] * 1e-3
        if sec > 0:
            row["tflops"] = round(c["gflop"] / 1e3 / sec, 1)
            row["frac_compute"] = round(c["gflop"] / 1e3 / sec / peak, 4)
            row["hbm_tbs"] = round(c["bytes"] / 1e12 / sec, 3)
            row["frac_hbm"] = round(c["bytes"] / 1e12 / sec / HBM_PEAK_TBS, 4)
            row["bound"] = "hbm" if (row["frac_hbm"] > row["frac_compute"] or c["gflop"] == 0) else "mfma"
        rows.append(row)
    dom = rows[0] if rows else None
    roofline = None
    if dom is not None and "bound" in dom:
        if dom["bound"] == "mfma":
            roofline = {"bound": "mfma", "achieved": dom["tflops"], "peak": round(peak, 1), "unit": "TFLOP/s", "frac": dom["frac_compute"]}
        else:
            roofline = {"bound": "hbm", "achieved": dom["hbm_tbs"] * 1e3, "peak": HBM_PEAK_TBS * 1e3, "unit": "GB/s", "frac": dom["frac_hbm"]}
        roofline.update({"kernel_class": dom["class"], "traffic": None,
                         "peak_note": ("dense bf16 MFMA 2516.6 TFLOP/s (bf16 storage: one product per product)" if a.act_dtype == "bf16"
                                       else "f16-pair ceiling 2516.6 / 3 (fp32 storage, three products per fp32 product)"),
                         "classes": rows,
                         "step_gflop_algorithmic": round(sum(r["algorithmic_gflop_per_step"] for r in rows), 1),
                         "step_tflops_end_to_end": round(sum(r["algorithmic_gflop_per_step"] for r in rows) / ms, 1),
                         "note": "HIP events around each C-ABI call (its packs included) on its launch stream, %d steps after the "
                                 "timed region with the stream overlaps off; per-class sums" % nrep})
    print(json.dumps({"roofline": roofline,"config": {"workload": "r3d_byol 3D-ResNet-%d, B=%d clip pairs 3x%dx%dx%d, full loss_com, clip 18, SGD; %s activation storage"
                                 % (a.depth, a.batch, a.frames, a.size, a.size, a.act_dtype)},
                      "ms_per_step": round(ms, 2), "clips_per_s": round(a.batch / ms * 1e3, 2),
                      "max_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))


if __name__ == "__main__":
    main()
