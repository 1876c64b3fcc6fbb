#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE implementation (CPU, fp64 truth).

Runs only in the build container: it imports /root/reference (read-only) and refuses
to run without it.  Nothing of the reference is copied -- the outputs are data
(.npz fixtures committed under tests/golden/).

The Python caller main_byol.py:60-91 cannot be imported here (needs CUDA +
torchvision), so this script drives the reference *module* with exactly that
sequence: model(clip_1, clip_2, o_type='loss_com') -> 6x nn.CrossEntropyLoss ->
loss_weight sum -> zero_grad -> backward -> clip_grad_norm_(18) -> torch.optim.SGD.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [config ...]
"""
import math
import os
import sys
import time

import numpy as np
import torch

REF = "/root/reference"
if not os.path.isdir(REF):
    raise SystemExit("make_golden.py needs the reference at /root/reference (build container only)")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from models.pace import r21d_byol as ref_model  # noqa: E402  (reference)
from loss.NTXent import NTXentLoss  # noqa: E402  (reference)
from scheduler.cosine_anneal import CosineAnnealingWarmupRestarts  # noqa: E402 (reference)

from oracle import r21d_byol_oracle as orc  # noqa: E402  (closed-form fills only)

CONFIGS = {
    # name: (depth, B, T, HW, steps, lr, wd)
    "d1_small": (1, 4, 8, 56, 3, 0.005, 5e-4),
    "d1_cfg1": (1, 4, 16, 112, 1, 0.05, 5e-4),
    "r18_small": (18, 4, 8, 56, 2, 0.005, 5e-4),
    "r18_cfg2": (18, 4, 16, 112, 1, 0.05, 5e-4),
    "r34_small": (34, 8, 8, 64, 1, 0.05, 5e-4),
    # BASELINE configs[3] at its TRUE per-clip shape: R(2+1)D-34, 32-frame clips of 112x112 (B = 2: BatchNorm1d needs >= 2)
    "r34_cfg4": (34, 2, 32, 112, 1, 0.05, 5e-4),
    # heavy-tailed fills (oracle.closed_form_*: heavy=True): magnitudes inside every weight tensor and inside the clips
    # span six decades, half of the pixels are zero -- stresses the per-tensor operand scale of the 2xf16-split kernels
    "d1_heavy": (1, 4, 8, 56, 1, 0.05, 5e-4),
    "r18_heavy": (18, 4, 8, 56, 1, 0.05, 5e-4),
    "r34_heavy": (34, 8, 8, 64, 1, 0.05, 5e-4),
    # BASELINE configs[1]'s REAL objective at its clip shape (B = 4): "NT-Xent + overlap-rate head only" -- the reference
    # module's loss_com outputs with loss_weight (0.1, 1, 1, 0, 0) PLUS 1 x the reference's own NTXentLoss on the two online
    # projections, built as main_byol.py:191-197 builds it (batch_size = the batch, --temperature default 0.5, cosine).  The
    # projections are taken from the module's own forward by a hook on online_net.project, so NT-Xent's gradient flows into
    # the projector and the whole online encoder exactly as if the driver had added the term.  (d1_ntx: the same on depth 1.)
    "r18_cfg2nt": (18, 4, 16, 112, 1, 0.05, 5e-4),
    "d1_ntx": (1, 4, 8, 56, 2, 0.005, 5e-4),
}
LOSS_WEIGHT = (0.1, 1.0, 1.0, 1.0, 1.0)
# name: (loss_weight, NT-Xent weight, temperature)
NTX = {"r18_cfg2nt": ((0.1, 1.0, 1.0, 0.0, 0.0), 1.0, 0.5), "d1_ntx": ((0.1, 1.0, 1.0, 0.0, 0.0), 1.0, 0.5)}


def build_reference(layer_sizes, dtype, heavy=False):
    m = ref_model.R21DBYOL(pretrain=True)
    if tuple(layer_sizes) != (1, 1, 1, 1):
        m.online_net = ref_model.R2Plus1DNet(layer_sizes=tuple(layer_sizes), proj_flag=True)
        m.target_net = ref_model.R2Plus1DNet(layer_sizes=tuple(layer_sizes), proj_flag=True)
        m._set_grad(m.target_net, False)
    sd = orc.closed_form_state(layer_sizes, dtype=torch.float64, heavy=heavy)
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert list(m.state_dict().keys()) == list(sd.keys()), "state-dict order differs from oracle spec"
    m = m.to(dtype)
    m.train()
    return m


def checksums(named):
    out = {}
    for k, v in named:
        v = v.detach().double()
        out[k] = np.array([float(v.sum()), float(v.abs().sum())])
    return out


def run_config(name):
    depth, b, t, hw, steps, lr, wd = CONFIGS[name]
    ls = orc.layer_sizes_for_depth(depth)
    dtype = torch.float64
    heavy = name.endswith("_heavy")
    model = build_reference(ls, dtype, heavy)
    x1, x2, labels = orc.closed_form_clips(b, t, hw, dtype=dtype, heavy=heavy)
    crit = torch.nn.CrossEntropyLoss()
    opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=0.9, weight_decay=wd)
    names = [k for k, _ in model.named_parameters()]
    w, ntw, tau = NTX.get(name, (LOSS_WEIGHT, 0.0, 0.5))
    out = {"meta": np.array([depth, b, t, hw, steps], dtype=np.int64), "heavy": np.array(int(heavy)), "lr": np.array(lr), "wd": np.array(wd),
           "loss_weight": np.array(w), "ntxent_weight": np.array(ntw), "temperature": np.array(tau)}
    projs = []
    if ntw:
        model.online_net.project.register_forward_hook(lambda mod, inp, outp: projs.append(outp))
        crit_ctr = NTXentLoss(device="cpu", batch_size=b, temperature=tau, use_cosine_similarity=True)
    for step in range(1, steps + 1):
        t0 = time.time()
        # capture forward internals through hooks-free re-computation: run pieces as forward does
        loss_byol, logits = model(x1, x2, o_type="loss_com")
        loss_byol = loss_byol.mean()
        ce = [crit(logits[0], labels["spa"]), crit(logits[1], labels["tem"]), crit(logits[2], labels["pb"]),
              crit(logits[3], labels["pb"]), crit(logits[4], labels["rot1"]), crit(logits[5], labels["rot2"])]
        total = (w[0] * loss_byol + w[1] * ce[0] + w[2] * ce[1] + w[3] * ce[2] + w[3] * ce[3]
                 + w[4] * ce[4] + w[4] * ce[5])
        if ntw:
            z1, z2 = projs          # online_net(x1), online_net(x2) in forward order (r21d_byol.py:359-360)
            del projs[:]
            nt = crit_ctr(z1, z2)
            out["s%d.ntxent" % step] = np.array(float(nt))
            total = total + ntw * nt
        opt.zero_grad()
        total.backward()
        gn = {k: float(p.grad.detach().norm()) for k, p in model.named_parameters() if p.grad is not None}
        gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), 18)
        opt.step()
        pre = "s%d." % step
        out[pre + "loss_byol"] = np.array(float(loss_byol))
        out[pre + "loss_total"] = np.array(float(total))
        out[pre + "ce"] = np.array([float(c) for c in ce])
        out[pre + "grad_norm"] = np.array(float(gnorm))
        out[pre + "logits"] = np.stack([l.detach().numpy() for l in logits]).astype(np.float32)
        out[pre + "grad_norms"] = np.array([gn.get(k, -1.0) for k in names])
        cs = checksums(model.state_dict().items())
        out[pre + "state_cs"] = np.stack([cs[k] for k in model.state_dict().keys()])
        mcs = []
        for p in model.parameters():
            st = opt.state.get(p, {})
            buf = st.get("momentum_buffer")
            mcs.append([float(buf.double().sum()), float(buf.double().abs().sum())] if buf is not None else [0.0, 0.0])
        out[pre + "mom_cs"] = np.array(mcs)
        print("  [%s] step %d: byol %.6f total %.6f gnorm %.4f (%.1fs)" % (name, step, float(loss_byol),
                                                                            float(total), float(gnorm), time.time() - t0), flush=True)
    out["state_keys"] = np.array(list(model.state_dict().keys()))
    out["param_keys"] = np.array(names)

    # forward internals from a fresh model (step-1 state) -- features/projections/predictions
    del model, opt, total, loss_byol, logits, ce
    model = build_reference(ls, dtype, heavy)
    with torch.no_grad():
        f1, z1 = model.online_net(x1)
        f2, z2 = model.online_net(x2)
        p1 = model.predictor(z1)
        p2 = model.predictor(z2)
        model._update_target_net()
        _, t1 = model.target_net(x1)
        _, t2 = model.target_net(x2)
    for k, v in (("feat_1", f1), ("feat_2", f2), ("proj_1", z1), ("proj_2", z2), ("pred_1", p1), ("pred_2", p2),
                 ("tproj_1", t1), ("tproj_2", t2)):
        out["fwd." + k] = v.numpy().astype(np.float32)
    nt = NTXentLoss(device="cpu", batch_size=b, temperature=0.5, use_cosine_similarity=True)
    out["fwd.ntxent"] = np.array(float(nt(z1, z2)))
    # fp32 run of the reference for the noise-floor record
    m32 = build_reference(ls, torch.float32, heavy)
    a1, a2, _ = orc.closed_form_clips(b, t, hw, dtype=torch.float32, heavy=heavy)
    with torch.no_grad():
        l32, lg32 = m32(a1, a2, o_type="loss_com")
    out["fp32.loss_byol"] = np.array(float(l32))
    out["fp32.logits"] = np.stack([l.numpy() for l in lg32])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def run_misc():
    out = {}
    # NT-Xent known answers (loss/NTXent.py)
    for n, tau in ((4, 0.5), (8, 0.1), (16, 0.5)):
        i = torch.arange(n * 64, dtype=torch.float64)
        zi = (torch.sin(0.11 * i + 0.3) + 0.2 * torch.cos(0.7 * i)).view(n, 64)
        zj = (torch.sin(0.13 * i + 1.3) - 0.3 * torch.cos(0.5 * i)).view(n, 64)
        nt = NTXentLoss(device="cpu", batch_size=n, temperature=tau, use_cosine_similarity=True)
        zi.requires_grad_(True)
        zj.requires_grad_(True)
        l = nt(zi, zj)
        l.backward()
        out["ntxent.%d.%g" % (n, tau)] = np.array(float(l))
        out["ntxent.%d.%g.gi" % (n, tau)] = zi.grad.numpy()
        out["ntxent.%d.%g.gj" % (n, tau)] = zj.grad.numpy()
    # LR schedule as main_byol.py drives it
    for n_epochs, lr in ((300, 0.09), (10, 0.03)):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=lr, momentum=0.9)
        sch = CosineAnnealingWarmupRestarts(opt, first_cycle_steps=n_epochs, cycle_mult=1.0, max_lr=lr,
                                            min_lr=0.00001, warmup_steps=0.5 * n_epochs, gamma=0.5)
        lrs = []
        for _ in range(n_epochs):
            lrs.append(opt.param_groups[-1]["lr"])
            opt.step()
            sch.step()
        out["lrs.%d.%g" % (n_epochs, lr)] = np.array(lrs)
    # init pin: per-tensor checksums of R21DBYOL(pretrain=True) under manual_seed(1) (opts.py:160)
    torch.manual_seed(1)
    m = ref_model.R21DBYOL(pretrain=True)
    cs = checksums(m.state_dict().items())
    out["init.keys"] = np.array(list(m.state_dict().keys()))
    out["init.cs"] = np.stack([cs[k] for k in m.state_dict().keys()])
    np.savez_compressed(os.path.join(HERE, "misc.npz"), **out)
    print("wrote misc")


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or (["misc"] + list(CONFIGS))
    for c in which:
        if c == "misc":
            run_misc()
        else:
            run_config(c)
