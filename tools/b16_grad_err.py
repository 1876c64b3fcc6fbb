#!/usr/bin/env python3
"""Per-tensor gradient-norm distance of the bf16-storage HIP step from the bf16-storage oracle (fp64 between the rounding
points), next to the oracle's own distance when it runs fp32 between the rounding points -- i.e. how much of the HIP path's
distance is the spec's sensitivity to fp32-level perturbations (rounding flips amplified by depth) and how much is not.

    python tools/b16_grad_err.py --depth 50 --batch 4 --frames 8 --size 64
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cstp_amd.optim import FlatSGD  # noqa: E402
from cstp_amd.r3d_byol import R3DBYOL  # noqa: E402
from cstp_amd.train import PretrainStep  # noqa: E402
from oracle import r21d_byol_oracle as orc  # noqa: E402
from oracle import r3d_byol_oracle as r3d  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--size", type=int, default=64)
    a = ap.parse_args()
    layers = r3d.for_depth(a.depth)
    sd = r3d.closed_form_state(r3d.model_spec(layers), torch.float32)
    y1, y2, _ = orc.closed_form_clips(a.batch, a.frames, a.size, torch.float32)
    labels = r3d.closed_form_labels(a.batch)
    w = (0.1, 1.0, 1.0, 1.0, 1.0)

    def oracle(dt):
        osd = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        return r3d.train_step(osd, {}, y1.to(dt), y2.to(dt), labels, layers, 0.05, 0.9, 5e-4, w, False)

    r3d.set_storage("bf16")
    i64, i32 = oracle(torch.float64), oracle(torch.float32)
    r3d.set_storage(None)
    opts = argparse.Namespace(model_depth=a.depth, sample_size=a.size, sample_duration=a.frames, sc_type="B", n_classes=101,
                              act_dtype="bf16")
    model = R3DBYOL(pretrain=True, opts=opts)
    model.load_state_dict(sd, strict=True)
    model.cuda()
    arenas = model.flatten_parameters()
    model.train()
    opt = FlatSGD(model.parameters(), lr=0.0, momentum=0.0, weight_decay=0.0, arenas=arenas)
    step = PretrainStep(model, opt, w, clip_grad_norm=False)
    lab = {k: v.cuda() for k, v in labels.items()}
    out = step(y1.cuda(), y2.cuda(), lab["spa"], lab["tem"], lab["pb"], lab["rot1"], lab["rot2"])
    torch.cuda.synchronize()
    print("global grad norm: hip %.4f  oracle64 %.4f  oracle32 %.4f" % (float(arenas["grad"].double().norm()), float(i64["grad_norm"]),
                                                                       float(i32["grad_norm"])))
    rows = []
    for k, p in model.named_parameters():
        if k not in i64["grads"]:
            continue
        g64 = i64["grads"][k].double()
        gh = p.grad.detach().cpu().double()
        g32 = i32["grads"][k].double()
        n = float(g64.norm())
        rows.append((n, k, float((gh - g64).norm()) / max(n, 1e-30), float((g32 - g64).norm()) / max(n, 1e-30)))
    rows.sort(reverse=True)
    print("%-48s %12s %12s %12s" % ("tensor (by gradient norm)", "|g|", "hip-o64", "o32-o64"))
    for n, k, eh, e3 in rows[:25]:
        print("%-48s %12.4e %12.3e %12.3e" % (k, n, eh, e3))
    eh = np.array([r[2] for r in rows]); e3 = np.array([r[3] for r in rows])
    print("median over %d tensors: hip-o64 %.3e  o32-o64 %.3e ; max hip-o64 %.3e (%s)  max o32-o64 %.3e" % (
        len(rows), np.median(eh), np.median(e3), eh.max(), rows[int(eh.argmax())][1], e3.max()))
    r3d.for_depth(18)


if __name__ == "__main__":
    main()
