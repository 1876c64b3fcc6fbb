#!/usr/bin/env python3
"""Golden vectors for the 3D-ResNet-BYOL wrapper from the REFERENCE implementation (CPU, fp64 truth).

Runs only in the build container: it imports /root/reference/models/BE/r3d_byol.py (read-only) and refuses to run without
it.  Nothing of the reference is copied -- the outputs are data (tests/golden/r3d_*.npz).  The driver sequence is the one of
main_byol.py:60-91 (6x CrossEntropy, loss_weight sum, zero_grad, backward, clip_grad_norm_(18), SGD), restated because
main_byol.py itself needs CUDA + torchvision.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r3d.py [config ...]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

REF = "/root/reference"
if not os.path.isdir(REF):
    raise SystemExit("make_golden_r3d.py needs the reference at /root/reference (build container only)")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from models.BE import r3d_byol as ref_model  # noqa: E402  (reference)

from oracle import r21d_byol_oracle as orc  # noqa: E402  (closed-form fills only)
from oracle import r3d_byol_oracle as r3d  # noqa: E402  (closed-form fills only)

CONFIGS = {
    # name: (depth, B, T, HW, steps, lr, wd)
    "r3d_10_small": (10, 4, 8, 56, 2, 0.005, 5e-4),
    "r3d_18_small": (18, 4, 8, 56, 1, 0.05, 5e-4),
    "r3d_34_small": (34, 4, 8, 64, 1, 0.05, 5e-4),
}
LOSS_WEIGHT = (0.1, 1.0, 1.0, 1.0, 1.0)


def ref_opts(depth, t, hw, k=101):
    return argparse.Namespace(model_depth=depth, sample_size=hw, sample_duration=t, sc_type="B", n_classes=k)


def build(depth, t, hw, dtype):
    m = ref_model.R3DBYOL(pretrain=True, opts=ref_opts(depth, t, hw))
    sd = r3d.closed_form_state(r3d.model_spec(r3d.for_depth(depth)), torch.float64)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert list(m.state_dict().keys()) == list(sd.keys()), "state-dict order differs from oracle spec"
    return m.to(dtype).train()


def checksums(items):
    return np.stack([np.array([float(v.detach().double().sum()), float(v.detach().double().abs().sum())]) for _, v in items])


def run_config(name):
    depth, b, t, hw, steps, lr, wd = CONFIGS[name]
    dtype = torch.float64
    model = build(depth, t, hw, dtype)
    x1, x2, _ = orc.closed_form_clips(b, t, hw, dtype=dtype)
    labels = r3d.closed_form_labels(b)
    crit = torch.nn.CrossEntropyLoss()
    opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=0.9, weight_decay=wd)
    names = [k for k, _ in model.named_parameters()]
    out = {"meta": np.array([depth, b, t, hw, steps], dtype=np.int64), "lr": np.array(lr), "wd": np.array(wd),
           "loss_weight": np.array(LOSS_WEIGHT)}
    for step in range(1, steps + 1):
        t0 = time.time()
        loss_byol, logits = model(x1, x2, o_type="loss_com")
        loss_byol = loss_byol.mean()
        ce = [crit(logits[0], labels["spa"]), crit(logits[1], labels["tem"]), crit(logits[2], labels["pb"]),
              crit(logits[3], labels["pb"]), crit(logits[4], labels["rot1"]), crit(logits[5], labels["rot2"])]
        w = LOSS_WEIGHT
        total = w[0] * loss_byol + w[1] * ce[0] + w[2] * ce[1] + w[3] * ce[2] + w[3] * ce[3] + w[4] * ce[4] + w[4] * ce[5]
        opt.zero_grad()
        total.backward()
        gn = {k: float(p.grad.detach().norm()) for k, p in model.named_parameters() if p.grad is not None}
        gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), 18)
        opt.step()
        pre = "s%d." % step
        out[pre + "loss_byol"] = np.array(float(loss_byol.detach()))
        out[pre + "loss_total"] = np.array(float(total.detach()))
        out[pre + "ce"] = np.array([float(c.detach()) for c in ce])
        out[pre + "grad_norm"] = np.array(float(gnorm))
        out[pre + "logits_5"] = np.stack([l.detach().numpy() for l in logits[:2]]).astype(np.float32)
        out[pre + "logits_4"] = np.stack([l.detach().numpy() for l in logits[2:]]).astype(np.float32)
        out[pre + "grad_norms"] = np.array([gn.get(k, -1.0) for k in names])
        out[pre + "state_cs"] = checksums(model.state_dict().items())
        mcs = []
        for p in model.parameters():
            buf = opt.state.get(p, {}).get("momentum_buffer")
            mcs.append([float(buf.double().sum()), float(buf.double().abs().sum())] if buf is not None else [0.0, 0.0])
        out[pre + "mom_cs"] = np.array(mcs)
        print("  [%s] step %d: byol %.6f total %.6f gnorm %.4f (%.1fs)" % (name, step, float(loss_byol), float(total), float(gnorm),
                                                                            time.time() - t0), flush=True)
    out["state_keys"] = np.array(list(model.state_dict().keys()))
    out["param_keys"] = np.array(names)
    # forward internals from the step-1 state
    model = build(depth, t, hw, dtype)
    with torch.no_grad():
        f1 = model.online_net(x1)
        f2 = model.online_net(x2)
        p1, p2 = model.predictor(f1), model.predictor(f2)
        model._update_target_net()
        t1, t2 = model.target_net(x1), model.target_net(x2)
    for k, v in (("feat_1", f1), ("feat_2", f2), ("pred_1", p1), ("pred_2", p2), ("tfeat_1", t1), ("tfeat_2", t2)):
        out["fwd." + k] = v.numpy().astype(np.float32)
    # fine-tune / test wrapper on the same encoder weights: train-mode and eval-mode logits (r3d_byol.py:420-428)
    ft = ref_model.R3DBYOL(pretrain=False, cls_bn=True, opts=ref_opts(depth, t, hw, k=11))
    sdf = r3d.closed_form_state(r3d.ft_spec(r3d.for_depth(depth), 11), torch.float64)
    res = ft.load_state_dict(sdf, strict=True)
    assert not res.missing_keys and not res.unexpected_keys and list(ft.state_dict().keys()) == list(sdf.keys())
    ft = ft.to(dtype).train()
    with torch.no_grad():
        out["ft.train_logits"] = ft(x1, o_type="ft_all").numpy().astype(np.float32)
        ft.eval()
        out["ft.eval_logits"] = ft(x2, o_type="test").numpy().astype(np.float32)
    out["ft.state_keys"] = np.array(list(ft.state_dict().keys()))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def run_backbone50(name="r3d_50_backbone", b=4, t=8, hw=64):
    """``r3d_50_backbone_224`` (round 3): the same at BASELINE configs[4]'s TRUE clip shape, 3x16x224x224 (B = 2: fp64 memory).
    Depth 50: the reference WRAPPER is shape-broken (view(-1, 512) of 2048 features, r3d_byol.py:204), its BACKBONE
    modules are not.  Drive ResNet(Bottleneck, [3, 4, 6, 3]) layer by layer up to the average pool (the statements of
    ResNet.forward :193-203 without the broken view) in train mode, fp64: pooled features of two clip batches, and the
    per-tensor gradient norms of  sum(features * c)  for a closed-form c -- pins Bottleneck forward and backward."""
    net = ref_model.resnet50(sample_size=hw, sample_duration=t, shortcut_type="B", num_classes=101)
    layers = r3d.for_depth(50)
    sd = r3d.closed_form_state(r3d.encoder_spec("online_net", layers), torch.float64)
    sd = {k[len("online_net."):]: v for k, v in sd.items()}
    res = net.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys and list(net.state_dict().keys()) == list(sd.keys())
    net = net.double().train()

    def backbone(x):
        x = net.maxpool(net.relu(net.bn1(net.conv1(x))))
        x = net.layer4(net.layer3(net.layer2(net.layer1(x))))
        return net.avgpool(x).flatten(1)

    x1, x2, _ = orc.closed_form_clips(b, t, hw, dtype=torch.float64)
    out = {"meta": np.array([50, b, t, hw, 0], dtype=np.int64)}
    f1 = backbone(x1)
    c = orc.hash_uniform(f1.numel(), 4242).reshape(f1.shape)
    (f1 * c).sum().backward()
    names = [k for k, _ in net.named_parameters()]
    out["feat_1"] = f1.detach().numpy().astype(np.float32)
    out["grad_norms"] = np.array([float(p.grad.norm()) for _, p in net.named_parameters()])
    out["param_keys"] = np.array(names)
    out["state_keys"] = np.array(list(net.state_dict().keys()))
    out["state_cs_after_fwd"] = checksums(net.state_dict().items())           # BN running statistics moved once
    with torch.no_grad():
        out["feat_2"] = backbone(x2).numpy().astype(np.float32)
        net.eval()
        out["feat_eval"] = backbone(x1).numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, out["feat_1"].shape)


if __name__ == "__main__":
    torch.set_num_threads(8)
    for c in (sys.argv[1:] or list(CONFIGS) + ["r3d_50_backbone"]):
        if c == "r3d_50_backbone":
            run_backbone50()
        elif c == "r3d_50_backbone_224":
            run_backbone50(c, 2, 16, 224)
        else:
            run_config(c)
