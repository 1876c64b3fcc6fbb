#!/usr/bin/env python3
"""Golden vectors for the fine-tune / validation / test path, from the REFERENCE implementation (CPU, fp64 truth).

Runs only in the build container: it imports /root/reference (read-only) and refuses to run without it.  Nothing of
the reference is copied -- the outputs are data (tests/golden/ft_*.npz).

main_ft_mp.py / test.py cannot be imported here (they need CUDA + torchvision), so this script drives the reference
*module* with exactly their sequence (main_ft_mp.py:199-212, 261-262; test.py:74-82):
    model(inputs, o_type=task) -> nn.CrossEntropyLoss -> zero_grad -> backward -> torch.optim.SGD.step   (train mode)
    model.eval(); no_grad; model(inputs, o_type=task)                                                    (validation)
    mean over the clips of one video of model(clips, None, o_type='test'), top-5                          (test)
with the parameter list of models/model.py:123-145 (get_fine_tuning_parameters, index 0 / 5).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ft.py [config ...]
"""
import os
import sys
import time

import numpy as np
import torch

REF = "/root/reference"
if not os.path.isdir(REF):
    raise SystemExit("make_golden_ft.py needs the reference at /root/reference (build container only)")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from models.pace import r21d_byol as ref_model  # noqa: E402  (reference)

from oracle import r21d_byol_oracle as orc  # noqa: E402  (closed-form fills only)
from oracle import r21d_ft_oracle as ftorc  # noqa: E402  (closed-form fills only)

CONFIGS = {
    # name: (depth, task, B, T, HW, classes, steps, lr, wd)
    "ft_all_d1": (1, "ft_all", 4, 8, 56, 11, 2, 0.01, 5e-4),
    "ft_fc_d1": (1, "ft_fc", 4, 8, 56, 11, 2, 0.05, 5e-4),
    "ft_all_r18": (18, "ft_all", 4, 8, 56, 101, 1, 0.01, 5e-4),
}


def build_reference(layer_sizes, num_classes, dtype):
    m = ref_model.R21DBYOL(pretrain=False, num_classes=num_classes, cls_bn=True)
    if tuple(layer_sizes) != (1, 1, 1, 1):
        m.online_net = ref_model.R2Plus1DNet(layer_sizes=tuple(layer_sizes), proj_flag=False)
    sd = ftorc.closed_form_state(layer_sizes, num_classes, dtype=torch.float64)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert list(m.state_dict().keys()) == list(sd.keys()), "state-dict order differs from oracle spec"
    return m.to(dtype).train()


def checksums(items):
    return np.stack([np.array([float(v.detach().double().sum()), float(v.detach().double().abs().sum())]) for _, v in items])


def run_config(name):
    depth, task, b, t, hw, k, steps, lr, wd = CONFIGS[name]
    ls = orc.layer_sizes_for_depth(depth)
    dtype = torch.float64
    model = build_reference(ls, k, dtype)
    x_train, x_val, labels = ftorc.closed_form_batch(b, t, hw, k, dtype=dtype)
    crit = torch.nn.CrossEntropyLoss()
    params = ref_model.get_fine_tuning_parameters(model, 0 if task == "ft_all" else 5)
    opt = torch.optim.SGD(params, lr=lr, momentum=0.9, weight_decay=wd)
    names = [n for n, _ in model.named_parameters()]
    out = {"meta": np.array([depth, b, t, hw, k, steps], dtype=np.int64), "task": np.array(task), "lr": np.array(lr),
           "wd": np.array(wd), "labels": labels.numpy()}
    for step in range(1, steps + 1):
        t0 = time.time()
        model.train()
        outputs = model(x_train, o_type=task)
        loss = crit(outputs, labels)
        opt.zero_grad()
        loss.backward()
        gn = {n: (float(p.grad.detach().norm()) if p.grad is not None else -1.0) for n, p in model.named_parameters()}
        opt.step()
        pre = "s%d." % step
        out[pre + "loss"] = np.array(float(loss))
        out[pre + "logits"] = outputs.detach().numpy().astype(np.float32)
        out[pre + "grad_norms"] = np.array([gn[n] for n in names])
        out[pre + "state_cs"] = checksums(model.state_dict().items())
        mcs = []
        for p in model.parameters():
            buf = opt.state.get(p, {}).get("momentum_buffer")
            mcs.append([float(buf.double().sum()), float(buf.double().abs().sum())] if buf is not None else [0.0, 0.0])
        out[pre + "mom_cs"] = np.array(mcs)
        # validation forward with the post-step weights and running statistics (main_ft_mp.py:261-262)
        model.eval()
        with torch.no_grad():
            val = model(x_val, o_type=task)
            vloss = crit(val, labels)
            vid = model(x_val, None, o_type="test")           # test.py:81: the clips of ONE video
        out[pre + "val_logits"] = val.numpy().astype(np.float32)
        out[pre + "val_loss"] = np.array(float(vloss))
        out[pre + "video_mean"] = vid.mean(dim=0, keepdim=True).numpy().astype(np.float32)
        out[pre + "video_top5"] = vid.mean(dim=0, keepdim=True).topk(5, 1, True)[1][0].numpy()
        print("  [%s] step %d: loss %.6f val_loss %.6f (%.1fs)" % (name, step, float(loss), float(vloss), time.time() - t0),
              flush=True)
    out["state_keys"] = np.array(list(model.state_dict().keys()))
    out["param_keys"] = np.array(names)
    out["requires_grad"] = np.array([p.requires_grad for p in model.parameters()])
    out["group_lrs"] = np.array([g["lr"] for g in opt.param_groups])
    # fp32 run of the reference for the noise-floor record
    m32 = build_reference(ls, k, torch.float32)
    a_train, _, _ = ftorc.closed_form_batch(b, t, hw, k, dtype=torch.float32)
    with torch.no_grad():
        out["fp32.logits"] = m32(a_train, o_type=task).numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def run_plateau():
    """ReduceLROnPlateau('min', patience) as main_ft_mp.py:153,279 drives it: lr trace for a fixed loss sequence."""
    out = {}
    for patience in (2, 10):
        p = torch.nn.Parameter(torch.zeros(1))
        q = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([{"params": p}, {"params": q, "lr": 0.0}], lr=0.05, momentum=0.9)
        sch = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, "min", patience=patience)
        i = np.arange(60)
        losses = 2.0 * np.exp(-i / 6.0) + 0.3 + 0.02 * np.sin(1.7 * i) + (i > 30) * 0.001 * (i - 30)
        lrs = []
        for l in losses:
            opt.step()
            sch.step(float(l))
            lrs.append([g["lr"] for g in opt.param_groups])
        out["plateau.%d.losses" % patience] = losses
        out["plateau.%d.lrs" % patience] = np.array(lrs)
    np.savez_compressed(os.path.join(HERE, "ft_misc.npz"), **out)
    print("wrote ft_misc")


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or (["misc"] + list(CONFIGS))
    for c in which:
        if c == "misc":
            run_plateau()
        else:
            run_config(c)
