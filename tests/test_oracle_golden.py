"""Pin the CPU oracle (oracle/r21d_byol_oracle.py, fp32) against golden vectors captured from the
reference implementation run in fp64 (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import r21d_byol_oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")
# Tolerances per optimisation step: (outputs/losses/state, per-tensor gradient & momentum).
# Step 1 is the parity bar (fp32 vs the fp64 truth, max-abs-diff / max-abs-ref, BASELINE.json: 1e-4).
# Gradients go back through 24+ train-mode BN layers over a batch of 4, where stock PyTorch fp32
# -- this oracle AND the reference module itself run in fp32 -- sits 3e-4..7e-4 from the fp64 truth
# on BN gamma/beta gradients at depth 1, 6e-3 at R18 and 1.3e-2 at R34 (more BN layers).  From step 2 on the fp32 and fp64 trajectories separate (BN1d heads
# over 4 samples are ill-conditioned); measured here for the REFERENCE fp32 vs its own fp64 run on
# d1_small: logits 1.5e-6 / 1.9e-4 / 1.2e-2 and momentum 6.6e-4 / 1.9e-2 / 1.3e-1 at steps 1 / 2 / 3.
# Later-step checks therefore only guard semantics (EMA order, momentum, weight decay, running
# stats), whose errors are O(1).
TOLS = {1: (1e-4, 2e-2), 2: (3e-2, 2e-1), 3: (5e-2, 5e-1)}
TOL = TOLS[1][0]
# R(2+1)D-34 (33 conv+BN layers): stock PyTorch fp32 itself sits 0.9e-4 from the fp64 truth on the
# projector outputs of r34_small, so that fixture's output bar is 3e-4 for every implementation.
# (The same holds for the other R(2+1)D-34 fixtures: r34_heavy, and r34_cfg4 = BASELINE configs[3]'s true clip shape.)
OUT_SCALE = {"r34_small": 3.0, "r34_heavy": 3.0, "r34_cfg4": 3.0}
# ... and its per-tensor gradient / momentum checksums (66 train-mode BN layers over a batch of 8) sit 1.3e-2 (stock PyTorch
# fp32), 1.5e-2..2.03e-2 (the HIP kernels, depending on the tile the autotuner picks -- each kernel variant is 4e-7 rms from
# fp64 per convolution, tools/split_accuracy.py) from the fp64 truth: noise amplification, not arithmetic.  3e-2 for that fixture.
# r34_heavy (six-decade magnitudes inside every weight tensor): stock PyTorch fp32 -- this oracle -- sits 3.2e-2 from the fp64
# truth on the momentum checksums (outputs: 3e-5), so 5e-2 there.
GRAD_SCALE = {"r34_small": 1.5, "r34_heavy": 2.5, "r34_cfg4": 1.5}
# post-step parameter checksums carry lr x (gradient noise): 2e-3 at step 1 (lr up to 0.05)
STATE_TOLS = {1: 2e-3, 2: 3e-2, 3: 5e-2}
# r34_cfg4 is a batch of TWO clips (fp64 memory): its BatchNorm1d heads normalise over two samples, the pre-clip gradient
# norm is 2315 and the noise in the clipped update is larger -- stock PyTorch fp32 (this oracle) sits 1.0e-3 from the
# fp64 truth on the post-step checksums, the HIP kernels 2.7e-3: 4e-3 for that fixture.
STATE_SCALE = {"r34_cfg4": 2.0}


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def cs_err(ours, ref):
    """checksum rows (sum, abs-sum): error relative to the abs-sum scale."""
    ours, ref = np.asarray(ours), np.asarray(ref)
    scale = np.maximum(np.abs(ref[:, 1]), 1e-12)
    return float((np.abs(ours - ref).max(axis=1) / scale).max())


def state_checksums(sd, keys):
    out = []
    for k in keys:
        v = sd[k].detach().double()
        out.append([float(v.sum()), float(v.abs().sum())])
    return np.array(out)


def is_heavy(g):
    """fixtures written since round 2 say whether their closed-form fills are the heavy-tailed ones"""
    return bool(int(g["heavy"])) if "heavy" in g.files else False


def ntx_weight(g):
    """fixtures written since round 3 may carry the NT-Xent term of BASELINE configs[1] (make_golden.py: NTX)"""
    return float(g["ntxent_weight"]) if "ntxent_weight" in g.files else 0.0


def run_oracle(name, steps=None):
    g = load(name)
    depth, b, t, hw, nsteps = [int(v) for v in g["meta"]]
    nsteps = steps or nsteps
    ls = orc.layer_sizes_for_depth(depth)
    sd = orc.closed_form_state(ls, torch.float32, heavy=is_heavy(g))
    x1, x2, labels = orc.closed_form_clips(b, t, hw, torch.float32, heavy=is_heavy(g))
    mom, infos, states, moms = {}, [], [], []
    keys = [str(k) for k in g["state_keys"]]
    pkeys = [str(k) for k in g["param_keys"]]
    for _ in range(nsteps):
        info = orc.train_step(sd, mom, x1, x2, labels, ls, float(g["lr"]), 0.9, float(g["wd"]), tuple(g["loss_weight"]), True,
                              ntxent_weight=ntx_weight(g), temperature=float(g["temperature"]) if "temperature" in g.files else 0.5)
        infos.append(info)
        states.append(state_checksums(sd, keys))
        moms.append(np.array([[float(mom[k].double().sum()), float(mom[k].double().abs().sum())] if k in mom else [0.0, 0.0]
                              for k in pkeys]))
    return g, infos, states, moms, pkeys


@pytest.mark.parametrize("name", ["d1_small", "r18_small", "r34_small", "d1_cfg1", "d1_heavy", "r18_heavy", "r34_heavy",
                                  "r34_cfg4", "d1_ntx"])
def test_oracle_matches_reference_golden(name):
    g, infos, states, moms, pkeys = run_oracle(name)
    for s, info in enumerate(infos, start=1):
        tol, gtol = TOLS[s]
        tol *= OUT_SCALE.get(name, 1.0)
        gtol *= GRAD_SCALE.get(name, 1.0)
        pre = "s%d." % s
        assert rel(float(info["loss_byol"]), g[pre + "loss_byol"]) < tol
        assert rel(float(info["loss_total"]), g[pre + "loss_total"]) < tol
        assert rel([float(c) for c in info["ce"]], g[pre + "ce"]) < tol
        assert rel(float(info["grad_norm"]), g[pre + "grad_norm"]) < gtol
        assert rel(torch.stack(info["logits"]).numpy(), g[pre + "logits"]) < tol
        if ntx_weight(g):
            assert rel(float(info["ntxent"]), g[pre + "ntxent"]) < tol
        gn = np.array([float(info["grads"][k].norm()) if k in info["grads"] else -1.0 for k in pkeys])
        assert rel(gn, g[pre + "grad_norms"]) < gtol
        assert cs_err(states[s - 1], g[pre + "state_cs"]) < STATE_TOLS[s] * STATE_SCALE.get(name, 1.0)
        assert cs_err(moms[s - 1], g[pre + "mom_cs"]) < gtol
        if s == 1:
            for k in ("feat_1", "feat_2", "proj_1", "proj_2", "pred_1", "pred_2", "tproj_1", "tproj_2"):
                assert rel(info[k].numpy(), g["fwd." + k]) < tol, k
            assert rel(float(orc.ntxent(info["proj_1"], info["proj_2"], 0.5)), g["fwd.ntxent"]) < tol


def test_state_spec_matches_reference_state_dict():
    for name in ("d1_small", "r18_small", "r34_small"):
        g = load(name)
        depth = int(g["meta"][0])
        spec = orc.model_spec(orc.layer_sizes_for_depth(depth))
        assert [k for k, _, _ in spec] == [str(k) for k in g["state_keys"]]
        params = [k for k, _, kind in spec if orc.is_param(kind)]
        assert params == [str(k) for k in g["param_keys"]]
    # parameter counts quoted in SURVEY 2.3 (measured on the reference)
    spec = orc.model_spec((1, 1, 1, 1))
    n_all = sum(int(np.prod(s)) for k, s, kind in spec if orc.is_param(kind))
    n_train = sum(int(np.prod(s)) for k, s, kind in spec if orc.is_param(kind) and not k.startswith("target_net."))
    assert n_all == 43997954 and n_train == 25425547


def test_ntxent_known_answers():
    g = load("misc")
    for n, tau in ((4, 0.5), (8, 0.1), (16, 0.5)):
        i = torch.arange(n * 64, dtype=torch.float64)
        zi = (torch.sin(0.11 * i + 0.3) + 0.2 * torch.cos(0.7 * i)).view(n, 64).requires_grad_(True)
        zj = (torch.sin(0.13 * i + 1.3) - 0.3 * torch.cos(0.5 * i)).view(n, 64).requires_grad_(True)
        l = orc.ntxent(zi, zj, tau)
        l.backward()
        key = "ntxent.%d.%g" % (n, tau)
        assert rel(float(l), g[key]) < 1e-9
        assert rel(zi.grad.numpy(), g[key + ".gi"]) < 1e-9
        assert rel(zj.grad.numpy(), g[key + ".gj"]) < 1e-9


def test_lr_schedule_known_answers():
    g = load("misc")
    for n_epochs, lr in ((300, 0.09), (10, 0.03)):
        ref = g["lrs.%d.%g" % (n_epochs, lr)]
        assert rel(orc.cosine_warmup_lrs(n_epochs, lr), ref) < 1e-12
    ref = g["lrs.300.0.09"]
    assert abs(ref[0] - 1e-5) < 1e-12 and abs(ref[150] - 0.09) < 1e-9
