"""End-to-end parity of the HIP path (cstp_amd, through the C ABI) on a real MI355X:
  * against golden vectors captured from the reference run in fp64 (tests/golden/*.npz);
  * against the CPU oracle on seeded ragged inputs (odd T/H/W, batch 3);
  * size-independent properties at BASELINE.json's full single-GPU size (R18, B=16, 16x112x112).
The training loop below reads like the reference's main_byol.py:60-91."""
import numpy as np
import pytest
import torch

from test_oracle_golden import GRAD_SCALE, OUT_SCALE, STATE_SCALE, STATE_TOLS, TOLS, cs_err, is_heavy, load, rel, run_oracle

# widened fixtures: the HIP path's error against the fp64 truth may be at most this many times stock PyTorch fp32's (measured:
# 1.2 - 2.7 depending on the tile the autotuner picked, profiles/r03/state_err_r34cfg4*.log; a broken kernel is off by 10x and more)
HIP_VS_FP32 = 3.0

pytestmark = pytest.mark.gpu


def trel(a, b):
    """rel() for device tensors."""
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def build_model(layer_sizes, sd=None):
    from cstp_amd.r21d_byol import R21DBYOL
    model = R21DBYOL(pretrain=True, layer_sizes=layer_sizes)
    if sd is not None:
        missing = model.load_state_dict(sd, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
    model.cuda()
    model.flatten_parameters()
    model.train()
    return model


def one_step(model, opt, x1, x2, labels, w, clip=True):
    """main_byol.py:60-91 on the HIP path; returns scalars + pre-clip per-parameter gradient norms."""
    from cstp_amd import ops
    loss_byol, logits = model(x1, x2, o_type="loss_com")
    loss_byol = loss_byol.mean()
    ce = [ops.cross_entropy(logits[0], labels["spa"]), ops.cross_entropy(logits[1], labels["tem"]),
          ops.cross_entropy(logits[2], labels["pb"]), ops.cross_entropy(logits[3], labels["pb"]),
          ops.cross_entropy(logits[4], labels["rot1"]), ops.cross_entropy(logits[5], labels["rot2"])]
    total = (w[0] * loss_byol + w[1] * ce[0] + w[2] * ce[1] + w[3] * ce[2] + w[3] * ce[3] + w[4] * ce[4] + w[4] * ce[5])
    opt.zero_grad()
    total.backward()
    gnorms = {k: float(p.grad.norm()) for k, p in model.named_parameters() if p.requires_grad}
    gnorm = opt.clip_grad_norm_(18) if clip else None
    opt.step()
    return {"loss_byol": float(loss_byol), "loss_total": float(total), "ce": [float(c) for c in ce],
            "logits": torch.stack([l.detach() for l in logits]).cpu().numpy(), "grad_norm": float(gnorm),
            "grad_norms": gnorms}


def checksums(model, keys):
    sd = model.state_dict()
    return np.array([[float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in keys])


def momentum_checksums(opt):
    sd = opt.state_dict()["state"]
    n = len(opt.param_groups[0]["params"])
    return np.array([[float(sd[i]["momentum_buffer"].double().sum()), float(sd[i]["momentum_buffer"].double().abs().sum())]
                     if i in sd else [0.0, 0.0] for i in range(n)])


@pytest.mark.parametrize("name", ["d1_small", "r18_small", "r34_small", "d1_cfg1", "r18_cfg2", "d1_heavy", "r18_heavy",
                                  "r34_heavy", "r34_cfg4"])
def test_hip_path_matches_reference_golden(name):
    """``*_heavy``: magnitudes inside every weight tensor and inside the clips span six decades (2^0 .. 2^-20 per element,
    half of the pixels zero) -- the default 2xf16-split GEMM arithmetic, with its ONE power-of-two scale per activation
    tensor, has to hold the same 1e-4 bar there.  ``r34_cfg4``: BASELINE configs[3]'s true clip shape (R(2+1)D-34, 32
    frames of 112x112)."""
    from cstp_amd.optim import FlatSGD
    from oracle import r21d_byol_oracle as orc
    g = load(name)
    depth, b, t, hw, nsteps = [int(v) for v in g["meta"]]
    ls = orc.layer_sizes_for_depth(depth)
    sd = orc.closed_form_state(ls, torch.float32, heavy=is_heavy(g))
    x1, x2, labels = orc.closed_form_clips(b, t, hw, torch.float32, heavy=is_heavy(g))
    x1, x2 = x1.cuda(), x2.cuda()
    labels = {k: v.cuda() for k, v in labels.items()}
    keys = [str(k) for k in g["state_keys"]]
    pkeys = [str(k) for k in g["param_keys"]]

    # ---- forward internals (reference forward order, r21d_byol.py:359-366) on a fresh model
    model = build_model(ls, sd)
    assert list(model.state_dict().keys()) == keys          # state-dict contract
    assert [k for k, _ in model.named_parameters()] == pkeys
    with torch.no_grad():
        f1, z1 = model.online_net(x1)
        f2, z2 = model.online_net(x2)
        p1, p2 = model.predictor(z1), model.predictor(z2)
        model._update_target_net()
        _, t1 = model.target_net(x1)
        _, t2 = model.target_net(x2)
    for k, v in (("feat_1", f1), ("feat_2", f2), ("proj_1", z1), ("proj_2", z2), ("pred_1", p1), ("pred_2", p2),
                 ("tproj_1", t1), ("tproj_2", t2)):
        assert rel(v.cpu().numpy(), g["fwd." + k]) < TOLS[1][0] * OUT_SCALE.get(name, 1.0), k
    from cstp_amd import ops
    nt = ops.ntxent(torch.cat([z2, z1], 0), 0.5)
    assert rel(float(nt), g["fwd.ntxent"]) < TOLS[1][0] * OUT_SCALE.get(name, 1.0)

    # ---- optimisation steps
    model = build_model(ls, sd)
    opt = FlatSGD(model.parameters(), lr=float(g["lr"]), momentum=0.9, weight_decay=float(g["wd"]),
                  arenas=model.flatten_parameters())
    for s in range(1, nsteps + 1):
        tol, gtol = TOLS[s]
        tol *= OUT_SCALE.get(name, 1.0)
        gtol *= GRAD_SCALE.get(name, 1.0)
        pre = "s%d." % s
        out = one_step(model, opt, x1, x2, labels, tuple(g["loss_weight"]))
        assert rel(out["loss_byol"], g[pre + "loss_byol"]) < tol
        assert rel(out["loss_total"], g[pre + "loss_total"]) < tol
        assert rel(out["ce"], g[pre + "ce"]) < tol
        assert rel(out["logits"], g[pre + "logits"]) < tol
        gn = np.array([out["grad_norms"].get(k, -1.0) for k in pkeys])
        # state indices are torch.optim.SGD(model.parameters())'s: the frozen target tensors keep their rows (no state)
        mcs = momentum_checksums(opt)
        assert mcs.shape == g[pre + "mom_cs"].shape
        errs = {"grad_norm": (rel(out["grad_norm"], g[pre + "grad_norm"]), TOLS[s][1], GRAD_SCALE.get(name, 1.0)),
                "grad_norms": (rel(gn, g[pre + "grad_norms"]), TOLS[s][1], GRAD_SCALE.get(name, 1.0)),
                "state_cs": (cs_err(checksums(model, keys), g[pre + "state_cs"]), STATE_TOLS[s], STATE_SCALE.get(name, 1.0)),
                "mom_cs": (cs_err(mcs, g[pre + "mom_cs"]), TOLS[s][1], GRAD_SCALE.get(name, 1.0))}
        if s == 1 and (name in GRAD_SCALE or name in STATE_SCALE):
            # The fixtures whose bars test_oracle_golden.py widens FOR STOCK PYTORCH FP32 (R(2+1)D-34: noise amplification through
            # 66 train-mode BN layers).  The widened bar alone never passes the HIP path (round-3 VERDICT weak-3): a quantity is
            # either under the UNWIDENED bar -- where the HIP path sat on all of them at the end of round 3,
            # profiles/r03/r34_hip_vs_stock_fp32.log -- or it is under the widened one AND within a fixed factor of what stock
            # fp32 (the oracle, run here on the host) leaves against the same fp64 truth, so an arithmetic regression cannot
            # hide behind the widening.
            _, oinfos, ostates, omoms, _ = run_oracle(name, steps=1)
            ogn = np.array([float(oinfos[0]["grads"][k].norm()) if k in oinfos[0]["grads"] else -1.0 for k in pkeys])
            e_fp32 = {"grad_norms": rel(ogn, g[pre + "grad_norms"]), "state_cs": cs_err(ostates[0], g[pre + "state_cs"]),
                      "mom_cs": cs_err(omoms[0], g[pre + "mom_cs"])}
            e_fp32["grad_norm"] = e_fp32["grad_norms"]
            print("%s: (HIP error, stock-fp32 error, unwidened bar) %s" % (name, {k: (v[0], e_fp32[k], v[1]) for k, v in errs.items()}))
            for what, (e_hip, base, scale) in errs.items():
                assert e_hip < base or (e_hip < base * scale and e_hip < HIP_VS_FP32 * e_fp32[what]), (name, what, e_hip, e_fp32[what])
        else:
            for what, (e_hip, base, scale) in errs.items():
                assert e_hip < base * scale, (name, s, what, e_hip)
    # BN counters: online/target nets see two forwards per step (r21d_byol.py:359-366)
    msd = model.state_dict()
    assert int(msd["online_net.bn1.num_batches_tracked"]) == 2 * nsteps
    assert int(msd["target_net.bn1.num_batches_tracked"]) == 2 * nsteps
    assert int(msd["overlap_spa.1.num_batches_tracked"]) == nsteps
    assert int(msd["pb_cls.1.num_batches_tracked"]) == 2 * nsteps


@pytest.mark.parametrize("name", ["d1_ntx", "r18_cfg2nt"])
def test_pretrain_step_with_ntxent_matches_reference_golden(name):
    """BASELINE configs[1]'s REAL objective end to end -- loss_weight (0.1, 1, 1, 0, 0) + 1 x NT-Xent, the objective bench.py
    times -- driven through the product's own ``PretrainStep(ntxent=NTXentLoss(...))`` (cstp_amd/train.py) against fixtures
    in which the reference module and the reference's own ``NTXentLoss`` (loss/NTXent.py:46-62, built as main_byol.py:191-197
    builds it) ran that step in fp64: NT-Xent's gradient enters the projector and the whole online encoder.
    ``r18_cfg2nt`` is configs[1]'s clip shape (R(2+1)D-18, 3x16x112x112) at B = 4."""
    from cstp_amd.ntxent import NTXentLoss
    from cstp_amd.optim import FlatSGD
    from cstp_amd.train import PretrainStep
    from oracle import r21d_byol_oracle as orc
    from test_oracle_golden import ntx_weight
    g = load(name)
    depth, b, t, hw, nsteps = [int(v) for v in g["meta"]]
    assert ntx_weight(g) != 0.0
    ls = orc.layer_sizes_for_depth(depth)
    sd = orc.closed_form_state(ls, torch.float32)
    x1, x2, labels = orc.closed_form_clips(b, t, hw, torch.float32)
    x1, x2 = x1.cuda(), x2.cuda()
    lab = {k: v.cuda() for k, v in labels.items()}
    keys = [str(k) for k in g["state_keys"]]
    pkeys = [str(k) for k in g["param_keys"]]
    model = build_model(ls, sd)
    opt = FlatSGD(model.parameters(), lr=float(g["lr"]), momentum=0.9, weight_decay=float(g["wd"]),
                  arenas=model.flatten_parameters())
    ntx = NTXentLoss(device=x1.device, batch_size=b, temperature=float(g["temperature"]), use_cosine_similarity=True)
    step = PretrainStep(model, opt, tuple(g["loss_weight"]), clip_grad_norm=True, ntxent=ntx, ntxent_weight=ntx_weight(g))
    for s in range(1, nsteps + 1):
        tol, gtol = TOLS[s]
        pre = "s%d." % s
        out = step(x1, x2, lab["spa"], lab["tem"], lab["pb"], lab["rot1"], lab["rot2"])
        # StepOutput.loss_total is the loss_weight sum (what the reference logs); the objective adds the NT-Xent term
        total = float(out.loss_total) + ntx_weight(g) * float(out.ntxent)
        assert rel(float(out.ntxent), g[pre + "ntxent"]) < tol
        assert rel(float(out.loss_byol), g[pre + "loss_byol"]) < tol
        assert rel(total, g[pre + "loss_total"]) < tol
        assert rel([float(c) for c in out.ce], g[pre + "ce"]) < tol
        assert rel(torch.stack(list(out.logits)).cpu().numpy(), g[pre + "logits"]) < tol
        gnorm = float(out.grad_norm)
        assert rel(gnorm, g[pre + "grad_norm"]) < gtol
        # .grad holds the CLIPPED gradient after the fused optimizer pass: undo the coefficient (clip_grad_norm_, main_byol.py:89)
        coef = min(1.0, 18.0 / (gnorm + 1e-6))
        gn = {k: float(p.grad.norm()) / coef for k, p in model.named_parameters() if p.requires_grad}
        assert rel(np.array([gn.get(k, -1.0) for k in pkeys]), g[pre + "grad_norms"]) < gtol
        assert cs_err(checksums(model, keys), g[pre + "state_cs"]) < STATE_TOLS[s]
        assert cs_err(momentum_checksums(opt), g[pre + "mom_cs"]) < gtol


@pytest.mark.parametrize("depth,b,t,hw", [(1, 3, 6, 36), (18, 2, 5, 28)])
def test_hip_path_matches_oracle_on_ragged_inputs(depth, b, t, hw):
    """Odd temporal/spatial sizes (stride-2 layers see odd extents) and a batch of 3/2, seeded."""
    from cstp_amd.optim import FlatSGD
    from oracle import r21d_byol_oracle as orc
    ls = orc.layer_sizes_for_depth(depth)
    sd = orc.closed_form_state(ls, torch.float32)
    x1, x2, labels = orc.closed_form_clips(b, t, hw, torch.float32, seed_phase=5)
    w = (0.1, 1.0, 1.0, 1.0, 1.0)
    mom = {}
    osd = {k: v.clone() for k, v in sd.items()}
    info = orc.train_step(osd, mom, x1, x2, labels, ls, 0.05, 0.9, 5e-4, w, True)
    model = build_model(ls, sd)
    opt = FlatSGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4, arenas=model.flatten_parameters())
    out = one_step(model, opt, x1.cuda(), x2.cuda(), {k: v.cuda() for k, v in labels.items()}, w)
    tol, gtol = 2e-4, 2e-2   # fp32 vs fp32 (two different summation orders)
    assert rel(out["loss_byol"], float(info["loss_byol"])) < tol
    assert rel(out["loss_total"], float(info["loss_total"])) < tol
    assert rel(out["logits"], torch.stack(info["logits"]).numpy()) < 5 * tol
    assert rel(out["grad_norm"], float(info["grad_norm"])) < gtol
    pk = [k for k in out["grad_norms"]]
    assert rel(np.array([out["grad_norms"][k] for k in pk]), np.array([float(info["grads"][k].norm()) for k in pk])) < gtol
    msd = model.state_dict()
    for k in ("online_net.conv1.bn.running_mean", "online_net.conv5.block1.bn2.running_var", "target_net.bn1.running_var",
              "target_net.conv2.block1.conv1.spatial_conv.weight", "predictor.net.1.running_mean"):
        assert rel(msd[k].cpu().numpy(), osd[k].detach().numpy()) < 5 * tol, k


def _step_properties(layer_sizes, b, t):
    """One optimisation step at a BASELINE.json size, checked through properties that need no oracle run."""
    from cstp_amd.optim import FlatSGD
    from cstp_amd.synthetic import device_batch
    torch.manual_seed(1)
    model = build_model(layer_sizes)
    a = model.flatten_parameters()
    x1, x2, labels = device_batch(b, t, 112, torch.device("cuda"), seed=1)
    opt = FlatSGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4, arenas=a)
    t_before = a["target"].clone()
    q_before = a["param"][:a["n_encoder"]].clone()
    p_before = a["param"].clone()
    out = one_step(model, opt, x1, x2, labels, (0.1, 1, 1, 1, 1))
    # (1) everything finite, BYOL loss = sum of two (2 - 2cos) terms in [0, 8]
    assert np.isfinite(out["loss_total"]) and 0.0 <= out["loss_byol"] <= 8.0
    assert all(np.isfinite(v) for v in out["grad_norms"].values())
    # (2) EMA is linear in the pre-step online weights and runs BEFORE the optimiser step
    expect = t_before * 0.996 + q_before * (1.0 - 0.996)
    assert trel(a["target"], expect) < 1e-6
    # (3) SGD first step: p_new = p - lr * (clip * g + wd * p); .grad holds the clipped gradient
    coef = min(1.0, 18.0 / (out["grad_norm"] + 1e-6))
    clipped = a["grad"]
    total_norm = float(clipped.double().norm())
    assert abs(total_norm - out["grad_norm"] * coef) / (out["grad_norm"] * coef) < 1e-4
    assert trel(a["param"], p_before - 0.01 * (clipped + 5e-4 * p_before)) < 1e-5
    # BN counters and running statistics moved (two forwards per net per step), every buffer finite
    msd = model.state_dict()
    assert int(msd["online_net.bn1.num_batches_tracked"]) == 2 and int(msd["target_net.bn1.num_batches_tracked"]) == 2
    assert bool(torch.isfinite(a["buffers"]).all()) and float(a["buffers"].abs().sum()) > 0
    return model, out


def test_full_size_properties_r34_t32_b8():
    """BASELINE.json configs[3] at its full per-GPU share: R(2+1)D-34, 8 clip pairs of 3x32x112x112 (the reference-fp64
    golden ``r34_cfg4`` pins the same clip shape at B = 2)."""
    _step_properties((3, 4, 6, 3), 8, 32)


def test_full_size_properties_r18_b16():
    """BASELINE.json cfg2 size (R(2+1)D-18, B=16, 3x16x112x112): properties that need no oracle run."""
    from cstp_amd import ops
    _step_properties((2, 2, 2, 2), 16, 16)
    # (4) train-mode BN output statistics: per-channel mean = beta, var = gamma^2 (up to eps)
    y = torch.randn(16, 64, 16, 56, 56, device="cuda") * 3 + 1
    gamma = torch.rand(64, device="cuda") + 0.5
    beta = torch.randn(64, device="cuda")
    z = ops.batch_norm_act(y, gamma, beta)
    assert trel(z.mean(dim=(0, 2, 3, 4)), beta) < 1e-4
    assert trel(z.var(dim=(0, 2, 3, 4), unbiased=False), gamma * gamma) < 1e-3
    # (5) conv linearity at the S1 shape: conv(a*x1 + x2) == a*conv(x1) + conv(x2)
    wt = torch.randn(144, 64, 1, 3, 3, device="cuda") * 0.05
    u, v = torch.randn(2, 16, 64, 16, 56, 56, device="cuda").unbind(0)
    lhs = ops.conv3d(1.7 * u + v, wt, None, 1, (0, 1, 1))
    rhs = 1.7 * ops.conv3d(u, wt, None, 1, (0, 1, 1)) + ops.conv3d(v, wt, None, 1, (0, 1, 1))
    assert trel(lhs, rhs) < 1e-5
    # (6) <dy, conv(x)> == <conv_dgrad(dy), x> == <conv_wgrad(x, dy), w>  (adjoint identities)
    u.requires_grad_(True)
    wt.requires_grad_(True)
    yy = ops.conv3d(u, wt, None, 1, (0, 1, 1))
    dy = torch.randn_like(yy)
    yy.backward(dy)
    lhs = float((dy.double() * yy.detach().double()).sum())
    assert abs(float((u.grad.double() * u.detach().double()).sum()) - lhs) / abs(lhs) < 1e-4
    assert abs(float((wt.grad.double() * wt.detach().double()).sum()) - lhs) / abs(lhs) < 1e-4


def test_three_gemm_arithmetics_against_the_fp64_truth_on_a_full_backward():
    """One loss_com forward + backward of an R(2+1)D-18 at a geometry no other test uses, under the three GEMM arithmetics the
    library has -- native f32 MFMA (analytic tiles, no autotune), exact bf16 triple (six products), f16 pair (three products,
    the default) -- from identical weights and clips, each measured against the fp64 run of the CPU oracle.  Loss and logits
    agree to fp32 rounding; the gradient of this network amplifies rounding noise by ~1e4 (62 train-mode BatchNorms over a
    batch of 3: test_oracle_golden.py GRAD_SCALE), so the statement for it is comparative: neither split arithmetic sits
    further from the truth than a small multiple of where the native f32 MFMA chain sits."""
    from cstp_amd import ops
    from oracle import r21d_byol_oracle as orc
    ls = orc.layer_sizes_for_depth(18)
    w = (0.1, 1.0, 1.0, 1.0, 1.0)
    sd64 = orc.closed_form_state(ls, torch.float64)
    c1, c2, lab64 = orc.closed_form_clips(3, 6, 40, torch.float64)
    info = orc.train_step(sd64, {}, c1, c2, lab64, ls, 0.0, 0.9, 0.0, w, False)
    truth = {k: v.double() for k, v in info["grads"].items()}
    truth_logits = torch.stack(info["logits"][2:]).double()

    sd = orc.closed_form_state(ls, torch.float32)
    x1, x2, labels = orc.closed_form_clips(3, 6, 40, torch.float32)
    x1, x2 = x1.cuda(), x2.cuda()
    labels = {k: v.cuda() for k, v in labels.items()}

    def run():
        model = build_model(ls, sd)
        arenas = model.flatten_parameters()
        loss_byol, logits = model(x1, x2, o_type="loss_com")
        ce = [ops.cross_entropy(logits[0], labels["spa"]), ops.cross_entropy(logits[1], labels["tem"]),
              ops.cross_entropy(logits[2], labels["pb"]), ops.cross_entropy(logits[3], labels["pb"]),
              ops.cross_entropy(logits[4], labels["rot1"]), ops.cross_entropy(logits[5], labels["rot2"])]
        total = w[0] * loss_byol.mean() + w[1] * ce[0] + w[2] * ce[1] + w[3] * (ce[2] + ce[3]) + w[4] * (ce[4] + ce[5])
        arenas["grad"].zero_()
        total.backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().double().cpu() for k, p in model.named_parameters() if p.requires_grad}
        return float(total.detach()), torch.stack([l.detach() for l in logits[2:]]).double().cpu(), grads

    saved = ops.AUTOTUNE
    try:
        ops.AUTOTUNE = False                   # untuned geometry + no tuning call = the analytic native-f32 tiles
        native = run()
        ops.AUTOTUNE = True
        ops.set_split_terms(3)
        triple = run()
        ops.set_split_terms(2)
        pair = run()
    finally:
        ops.AUTOTUNE = saved
        ops.set_split_terms(0)

    keys = sorted(truth)
    assert keys == sorted(native[2])
    tvec = torch.cat([truth[k].reshape(-1) for k in keys])

    def errs(r):
        gvec = torch.cat([r[2][k].reshape(-1) for k in keys])
        return (abs(r[0] - float(info["loss_total"])) / abs(float(info["loss_total"])),
                float((r[1] - truth_logits).abs().max() / truth_logits.abs().max()),
                float((gvec - tvec).norm() / tvec.norm()))
    report = {"native f32": errs(native), "bf16 triple": errs(triple), "f16 pair": errs(pair)}
    print("vs fp64 truth (loss, logits max, gradient L2):", report)
    for name, (dl, dlog, dg) in report.items():
        assert dl < 1e-5 and dlog < 1e-4, (name, report)
    worst_native = max(report["native f32"][2], 1e-3)
    assert report["bf16 triple"][2] < 3 * worst_native and report["f16 pair"][2] < 3 * worst_native, report


def test_pack_plan_steps_match_steps_that_pack_inside_every_call(monkeypatch):
    """ops.PackPlan (the weight packs of a step from three launches: online forward packs at the top of the step, online
    data-gradient packs beside the forward pass on the side stream, target packs behind the EMA) against the per-call packs: seven
    steps of the same model on the same clips.  The packs themselves are bit-identical (test_pack_replay_writes_what_the_call_packs
    below); two RUNS of a step are not guaranteed to be -- the BatchNorm sums from the convolution epilogues are fp64 LDS atomics
    whose order follows wave timing, and a last-bit difference there occasionally survives the rounding to fp32 -- so the
    trajectories are compared to 1e-5, and the plan must actually have recorded and replayed."""
    from cstp_amd import ops
    from cstp_amd.optim import FlatSGD
    from cstp_amd.synthetic import device_batch
    from cstp_amd.train import PretrainStep
    ops.set_deterministic(True)
    try:
        dev = torch.device("cuda", 0)
        x1, x2, lab = device_batch(2, 8, 56, dev, seed=3)
        finals = []
        for plan_on in ("1", "0"):
            monkeypatch.setenv("CSTP_PACK_PLAN", plan_on)
            torch.manual_seed(5)
            model = build_model((1, 1, 1, 1))
            opt = FlatSGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4, arenas=model._arenas)
            step = PretrainStep(model, opt, (0.1, 1.0, 1.0, 1.0, 1.0), clip_grad_norm=True)
            assert (step._packs is not None) == (plan_on == "1")
            losses = []
            for _ in range(7):
                out = step(x1, x2, lab["spa"], lab["tem"], lab["pb"], lab["rot1"], lab["rot2"])
                losses.append(float(out.loss_total))
            torch.cuda.synchronize()
            if plan_on == "1":
                st = step._packs.stats
                assert step._packs.state == "replay" and st["recorded_calls"] > 50, st
                assert st["replays"] >= 3 * 3 and st["skipped_calls"] >= 3 * st["recorded_calls"] - 5, st
                assert set(step._packs.tables) == {"online", "online_d", "target"}
            finals.append((losses, {k: v.clone() for k, v in model.state_dict().items()},
                           [s["momentum_buffer"].clone() for s in opt.state_dict()["state"].values()]))
        (la, sa, ma), (lb, sb, mb) = finals
        assert max(abs(a - b) / abs(b) for a, b in zip(la, lb)) < 1e-5, list(zip(la, lb))
        for k in sa:
            if sa[k].dtype.is_floating_point:
                assert trel(sa[k], sb[k]) < 1e-3, k
            else:
                assert torch.equal(sa[k], sb[k]), k
    finally:
        ops.set_deterministic(False)
        ops.pack_plan = None


def test_pack_replay_writes_what_the_call_packs():
    """cstp_pack_mode / cstp_pack_recorded / cstp_pack_replay at the op level, on one layer per pack kind (f16-pair rows, LDS-patch
    K-tiles 3x3 and temporal, native fp32 re-layout) and direction: the call's result with its own pack == the result after the
    recorded pack was REPLAYED into a scrubbed workspace and the call skipped it, bit for bit."""
    import ctypes
    from cstp_amd import _lib, ops
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    cases = [((2, 32, 4, 14, 14), 48, (1, 3, 3), (0, 1, 1), (2, 4, 0, 0), (2, 4, 0, 0)),      # patch kernels, both directions
             ((2, 48, 8, 14, 14), 64, (3, 1, 1), (1, 0, 0), (2, 4, 0, 0), (1, 4, 0, 0)),      # temporal patch / f16-pair gather
             ((2, 24, 2, 7, 7), 40, (1, 3, 3), (0, 1, 1), (1, 3, 0, 0), (1, 2, 0, 0)),        # f16-pair gather kernels
             ((2, 24, 2, 7, 7), 40, (1, 1, 1), (0, 0, 0), (0, 2, 1, 1), (0, 2, 1, 1))]        # native f32 tiles
    g = torch.Generator().manual_seed(4)
    for xs, k, ks, pad, tf, td in cases:
        ws_shape = (k, xs[1]) + ks
        ops.set_conv_tile(xs, ws_shape, (1, 1, 1), pad, 0, tf)
        ops.set_conv_tile(xs, ws_shape, (1, 1, 1), pad, 1, td)
        desc = ops._desc(xs, ws_shape, (1, 1, 1), pad)
        x = torch.randn(xs, generator=g).cuda()
        w = (torch.randn(ws_shape, generator=g) * 0.1).cuda()
        y = torch.empty(ops.conv_out_shape(xs, ws_shape, (1, 1, 1), pad), device="cuda")
        dy = torch.randn(y.shape, generator=g).cuda()
        dx = torch.empty_like(x)
        nbytes = lib.cstp_conv3d_workspace_bytes(ctypes.byref(desc))
        wsb = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")

        def fwd():
            _lib.check(lib.cstp_conv3d_forward_am(st, ctypes.byref(desc), x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(),
                                                  wsb.data_ptr(), nbytes, None), "fwd")

        def dgr():
            _lib.check(lib.cstp_conv3d_backward_data_am(st, ctypes.byref(desc), dy.data_ptr(), w.data_ptr(), dx.data_ptr(),
                                                        wsb.data_ptr(), nbytes, None), "dgrad")
        for call, out in ((fwd, y), (dgr, dx)):
            lib.cstp_pack_mode(1)
            call()
            lib.cstp_pack_mode(0)
            n = lib.cstp_pack_recorded(None, 0)
            assert n >= 1
            recs = (_lib.PackRec * n)()
            assert lib.cstp_pack_recorded(recs, n) == n and lib.cstp_pack_recorded(None, 0) == 0
            want = out.clone()
            wsb.fill_(0xFF)                         # scrub: NaN patterns wherever the replay does not write
            out.zero_()
            first, tot = [], 0
            for r in recs:
                first.append(tot)
                tot += int(r.nblocks)
            recs_dev = torch.frombuffer(bytearray(bytes(recs)), dtype=torch.uint8).clone().cuda()
            first_dev = torch.tensor(first, dtype=torch.int32, device="cuda")
            _lib.check(lib.cstp_pack_replay(st, recs_dev.data_ptr(), first_dev.data_ptr(), n, tot), "replay")
            lib.cstp_pack_mode(2)
            call()
            lib.cstp_pack_mode(0)
            torch.cuda.synchronize()
            assert torch.equal(out, want), (xs, ks, "forward" if call is fwd else "data gradient", int(recs[0].kind))
