"""Parity of the fine-tune / validation / test path on a real MI355X (HIP kernels through the C ABI):
  * the kernels this path adds (eval-mode BatchNorm, L2 normalise, Adam/AdamW) against PyTorch CPU fp64;
  * R21DBYOL(pretrain=False) train steps, model.eval() validation and video-level test against golden vectors captured
    from the reference in fp64 (tests/golden/ft_*.npz) and against the CPU oracle on ragged inputs;
  * the drivers end to end: pre-training checkpoint -> main_ft_mp.py (ft_all, ft_fc) -> test.py.
The loops read like the reference's main_ft_mp.py:199-212,261-262 and test.py:74-82."""
import importlib.util
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from test_ft_oracle_golden import TOLS, VAL_TOLS, cs_err, load, rel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rand(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1


@pytest.mark.parametrize("shape,relu,res", [((3, 10, 4, 6, 6), True, False), ((2, 7, 3, 5, 5), False, True),
                                            ((4, 64, 2, 8, 8), True, True), ((5, 12), False, False), ((1, 33), True, False)])
def test_bn_eval(shape, relu, res):
    from cstp_amd import ops
    c = shape[1]
    x, r = _rand(shape, 1), (_rand(shape, 2) if res else None)
    gamma, beta, rm, rv = _rand((c,), 3), _rand((c,), 4), _rand((c,), 5), _rand((c,), 6).abs() + 0.1
    ref = F.batch_norm(x, rm, rv, gamma, beta, False, 0.1, 1e-5)
    if r is not None:
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    d = lambda t: None if t is None else t.float().cuda()
    rm_d, rv_d = d(rm), d(rv)
    with torch.no_grad():
        y = ops.batch_norm_eval(d(x), d(gamma), d(beta), rm_d, rv_d, d(r), relu)
    assert rel_err(y, ref) < 1e-5
    assert torch.equal(rm_d.cpu(), rm.float()) and torch.equal(rv_d.cpu(), rv.float())   # nothing updated
    xg = d(x).requires_grad_(True)
    with pytest.raises(RuntimeError):
        ops.batch_norm_eval(xg, d(gamma), d(beta), rm_d, rv_d, d(r), relu)               # forward-only, loudly


def test_l2_normalize_fwd_bwd():
    from cstp_amd import ops
    for rows, f, seed in ((4, 512, 1), (3, 70, 2), (1, 64, 3)):
        x = (_rand((rows, f), seed) * 3).requires_grad_(True)
        dy = _rand((rows, f), seed + 10)
        ref = F.normalize(x, p=2, dim=1)
        ref.backward(dy)
        xd = x.detach().float().cuda().requires_grad_(True)
        y = ops.l2_normalize(xd)
        y.backward(dy.float().cuda())
        assert rel_err(y, ref) < 1e-5 and rel_err(xd.grad, x.grad) < 1e-5
    z = torch.zeros(2, 8, device="cuda")
    assert torch.equal(ops.l2_normalize(z), z)                     # |x| < eps rows: x / eps


@pytest.mark.parametrize("decoupled", [False, True])
def test_flat_adam_matches_torch(decoupled):
    from cstp_amd.optim import FlatAdam
    sizes = [(5, 3), (7,), (2, 4, 3)]
    offs, n = [], 0
    for s in sizes:
        offs.append(n)
        n += (int(np.prod(s)) + 3) // 4 * 4
    arena = {"param": torch.zeros(n, device="cuda"), "grad": torch.zeros(n, device="cuda")}
    ref_params, params = [], []
    for i, (s, o) in enumerate(zip(sizes, offs)):
        v = _rand(s, 20 + i)
        ref_params.append(torch.nn.Parameter(v.clone()))
        p = torch.nn.Parameter(torch.empty(s, device="cuda"))
        p.data = arena["param"][o:o + v.numel()].view(s)
        p.data.copy_(v.float())
        p.grad = arena["grad"][o:o + v.numel()].view(s)
        params.append(p)
    kw = dict(lr=0.01, betas=(0.9, 0.99), weight_decay=5e-2)
    ref = (torch.optim.AdamW if decoupled else torch.optim.Adam)(ref_params, **kw)
    opt = FlatAdam(params, decoupled=decoupled, arenas=arena, **kw)
    for step in range(4):
        for i, (rp, p) in enumerate(zip(ref_params, params)):
            g = _rand(rp.shape, 100 + 10 * step + i)
            rp.grad = g.clone()
            p.grad.copy_(g.float())
        ref.step()
        opt.step()
    for rp, p in zip(ref_params, params):
        assert rel_err(p.data, rp.data) < 1e-5
    sd = opt.state_dict()
    assert float(sd["state"][0]["step"]) == 4 and sd["state"][2]["exp_avg_sq"].shape == (2, 4, 3)


# ---------------------------------------------------------------------------------------------------------
def build_ft(layer_sizes, k, sd, task):
    from cstp_amd.optim import FlatSGD
    from cstp_amd.r21d_byol import R21DBYOL, get_fine_tuning_parameters
    model = R21DBYOL(pretrain=False, num_classes=k, cls_bn=True, layer_sizes=layer_sizes)
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model.cuda()
    arenas = model.flatten_parameters()
    params = get_fine_tuning_parameters(model, 0 if task == "ft_all" else 5)
    return model, arenas, params


def run_ft(name_or_cfg, sd=None, batch=None, lr=None, wd=None):
    """main_ft_mp.py:199-212 + :261-262 + test.py:81-82 on the HIP path."""
    from cstp_amd import ops
    from cstp_amd.optim import FlatSGD
    from cstp_amd.train import FineTuneStep
    from oracle import r21d_byol_oracle as orc
    from oracle import r21d_ft_oracle as ftorc
    depth, task, b, t, hw, k, steps = name_or_cfg
    ls = orc.layer_sizes_for_depth(depth)
    sd = sd if sd is not None else ftorc.closed_form_state(ls, k, torch.float32)
    x_train, x_val, labels = batch if batch is not None else ftorc.closed_form_batch(b, t, hw, k, torch.float32)
    model, arenas, params = build_ft(ls, k, sd, task)
    opt = FlatSGD(params, lr=lr, momentum=0.9, weight_decay=wd, arenas=arenas)
    step_fn = FineTuneStep(model, opt, task)
    xt, xv, lab = x_train.cuda(), x_val.cuda(), labels.cuda()
    names = [n for n, _ in model.named_parameters()]
    out = []
    for _ in range(steps):
        model.train()
        loss, logits = step_fn(xt, lab)
        rec = {"loss": float(loss), "logits": logits.cpu().numpy(),
               "grad_norms": np.array([float(p.grad.norm()) if p.requires_grad else -1.0 for p in model.parameters()])}
        st = model.state_dict()
        rec["state_cs"] = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in st.values()])
        osd = opt.state_dict()["state"]
        rec["mom_cs"] = np.array([[float(osd[i]["momentum_buffer"].double().sum()),
                                   float(osd[i]["momentum_buffer"].double().abs().sum())] if i in osd else [0.0, 0.0]
                                  for i in range(len(names))])
        model.eval()
        with torch.no_grad():
            val = model(xv, o_type=task)
            rec["val_logits"] = val.cpu().numpy()
            rec["val_loss"] = float(ops.cross_entropy(val, lab))
            vid = model(xv, None, o_type="test")
            rec["video_mean"] = vid.mean(dim=0, keepdim=True).cpu().numpy()
            rec["video_top5"] = vid.mean(dim=0, keepdim=True).topk(5, 1, True)[1][0].cpu().numpy()
        out.append(rec)
    return model, out


@pytest.mark.parametrize("name", ["ft_all_d1", "ft_fc_d1", "ft_all_r18"])
def test_ft_hip_matches_reference_golden(name):
    g = load(name)
    depth, b, t, hw, k, steps = [int(v) for v in g["meta"]]
    task = str(g["task"])
    model, recs = run_ft((depth, task, b, t, hw, k, steps), lr=float(g["lr"]), wd=float(g["wd"]))
    assert list(model.state_dict().keys()) == [str(s) for s in g["state_keys"]]
    for s, rec in enumerate(recs, start=1):
        tol, gtol, stol = TOLS[s]
        pre = "s%d." % s
        assert rel(rec["loss"], g[pre + "loss"]) < tol
        assert rel(rec["logits"], g[pre + "logits"]) < tol
        assert rel(rec["grad_norms"], g[pre + "grad_norms"]) < gtol
        assert cs_err(rec["state_cs"], g[pre + "state_cs"]) < stol
        assert cs_err(rec["mom_cs"], g[pre + "mom_cs"]) < gtol
        assert rel(rec["val_logits"], g[pre + "val_logits"]) < VAL_TOLS[s]
        assert rel(rec["val_loss"], g[pre + "val_loss"]) < VAL_TOLS[s]
        assert rel(rec["video_mean"], g[pre + "video_mean"]) < VAL_TOLS[s]
        if s == 1:
            assert np.array_equal(rec["video_top5"], g[pre + "video_top5"])
    nbt = model.state_dict()["cls_bn.num_batches_tracked"]
    assert int(nbt) == steps                      # eval forwards do not count


def test_ft_hip_matches_oracle_ragged():
    """Odd T/H/W, batch 3, 7 classes: HIP vs the CPU oracle, train step + eval forward."""
    from oracle import r21d_byol_oracle as orc
    from oracle import r21d_ft_oracle as ftorc
    ls, k = (1, 1, 1, 1), 7
    sd = ftorc.closed_form_state(ls, k, torch.float32)
    x1, x2, _ = orc.closed_form_clips(3, 5, 38, torch.float32, seed_phase=5)
    x1, x2 = x1[..., :37], x2[..., :37]           # H=38, W=37
    labels = torch.tensor([6, 0, 3])
    osd, mom = {kk: v.clone() for kk, v in sd.items()}, {}
    info = ftorc.ft_train_step(osd, mom, x1, labels, ls, k, 0.02, 0.9, 1e-3, "ft_all")
    with torch.no_grad():
        oval = ftorc.ft_forward(osd, x2, ls, training=False)
    model, recs = run_ft((1, "ft_all", 3, 5, 38, k, 1), sd=sd, batch=(x1, x2, labels), lr=0.02, wd=1e-3)
    assert rel(recs[0]["loss"], float(info["loss"])) < 1e-4
    assert rel(recs[0]["logits"], info["logits"].numpy()) < 1e-4
    assert rel(recs[0]["val_logits"], oval.numpy()) < 2e-3
    st = model.state_dict()
    for key in ("online_net.bn1.running_var", "cls_bn.running_mean", "cls_bn.running_var"):
        assert rel(st[key].cpu().numpy(), osd[key].numpy()) < 1e-4, key


def test_eval_is_deterministic_and_stateless():
    """model.eval() forwards leave every buffer and parameter untouched and are batch-composition independent
    (running statistics): logits of a clip do not depend on its neighbours."""
    from oracle import r21d_ft_oracle as ftorc
    ls, k = (1, 1, 1, 1), 5
    sd = ftorc.closed_form_state(ls, k, torch.float32)
    model, _, _ = build_ft(ls, k, sd, "ft_all")
    x, _, _ = ftorc.closed_form_batch(4, 4, 32, k, torch.float32)
    x = x.cuda()
    model.eval()
    before = {kk: v.clone() for kk, v in model.state_dict().items()}
    with torch.no_grad():
        full = model(x, o_type="test")
        solo = torch.cat([model(x[i:i + 1], o_type="test") for i in range(4)])    # batch of ONE works in eval mode
    assert float((full - solo).abs().max()) < 1e-5 * float(full.abs().max())
    for kk, v in model.state_dict().items():
        assert torch.equal(v, before[kk]), kk
    model.train()
    with pytest.raises(ValueError):
        model(x[:1], o_type="ft_all")            # train-mode BatchNorm1d over one sample, as in the reference


def _load_script(name):
    spec = importlib.util.spec_from_file_location("cstp_script_" + name, os.path.join(ROOT, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_drivers_pretrain_checkpoint_to_finetune_to_test(tmp_path, capsys):
    """The loop the reference exists for: pre-training checkpoint (wire format of main_byol.py:132-140) ->
    main_ft_mp.py --task ft_all (neq load, training, validation, plateau scheduler, best checkpoint) ->
    main_ft_mp.py --task ft_fc -> test.py (video-level accuracy)."""
    from cstp_amd.opts import parse_opts
    ft = _load_script("main_ft_mp")
    common = ["--dataset", "synthetic", "--n_classes", "4", "--batch_size", "8", "--sample_duration", "4", "--sample_size",
              "32", "--model_name", "r21d_byol", "--model_depth", "1", "--n_workers", "0", "--synthetic_len", "64",
              "--result_path", str(tmp_path), "--weight_decay", "1e-4", "--lr_patience", "1"]
    # 100 one-iteration epochs of main_byol.py write save_100.pth (checkpoints go out every 100 epochs, :132-140)
    pre_opts = parse_opts(common + ["--task", "loss_com", "--loss_weight", "0.1", "1", "1", "1", "1", "--n_epochs", "100",
                                    "--max_steps", "1", "--learning_rate", "0.01"])
    _load_script("main_byol").main(pre_opts)
    ckpt = str(tmp_path / "synthetic" / "loss_com" / "save_100.pth")
    pre_sd = torch.load(ckpt, map_location="cpu")
    assert pre_sd["arch"] == "r21d_byol-1" and pre_sd["epoch"] == 101      # epoch + 1, main_byol.py:134
    pre_sd = pre_sd["state_dict"]
    assert "module.target_net.bn1.running_mean" in pre_sd and "module.predictor.net.0.weight" in pre_sd
    # --resume_md_path continues the run: epochs 101-102 are appended, at the schedule's lr for those epochs
    res_opts = parse_opts(common + ["--task", "loss_com", "--loss_weight", "0.1", "1", "1", "1", "1", "--n_epochs", "102",
                                    "--max_steps", "1", "--learning_rate", "0.01", "--resume_md_path", ckpt])
    _load_script("main_byol").main(res_opts)
    rows = open(str(tmp_path / "synthetic" / "loss_com" / "synthetic_train_clip4modelr21d_byol1.log")).read().strip().split("\n")
    assert [r.split("\t")[0] for r in rows[-3:]] == ["100", "101", "102"]
    from cstp_amd.scheduler import CosineAnnealingWarmupRestarts

    class _Opt:
        param_groups = [{"lr": 0.0}]
    sch = CosineAnnealingWarmupRestarts(_Opt(), first_cycle_steps=102, cycle_mult=1.0, max_lr=0.01, min_lr=0.00001,
                                        warmup_steps=51.0, gamma=0.5)
    for _ in range(101):
        sch.step()
    assert abs(float(rows[-1].split("\t")[-1]) - float("{:.5f}".format(_Opt.param_groups[0]["lr"]))) < 1e-9
    accs = {}
    for task, lr, epochs in (("ft_all", "0.02", "6"), ("ft_fc", "0.05", "2")):
        opts = parse_opts(common + ["--task", task, "--pretrained_path", ckpt, "--learning_rate", lr, "--n_epochs", epochs])
        opts.highest_val = {"name": 0}
        ft.main(opts)
        d = tmp_path / "synthetic" / task
        best = [f for f in os.listdir(d) if f.endswith("_max.pth")]
        assert len(best) == 1, best                                  # the previous best is replaced, not kept
        md = torch.load(str(d / best[0]), map_location="cpu")
        assert md["arch"] == "r21d_byol-1" and "module.classify.weight" in md["state_dict"]
        assert all(k.startswith("module.") for k in md["state_dict"])
        rows = open(str(d / "synthetic_val_clip4modelr21d_byol1.log")).read().strip().split("\n")
        assert rows[0].split("\t") == ["epoch", "loss", "acc"] and len(rows) == 1 + int(epochs)
        accs[task] = max(float(r.split("\t")[2]) for r in rows[1:])
        if task == "ft_fc":      # frozen encoder: identical to the pre-training checkpoint's
            key = "module.online_net.conv3.block1.conv1.spatial_conv.weight"
            assert torch.equal(md["state_dict"][key], pre_sd[key])
            assert not torch.equal(md["state_dict"]["module.online_net.bn1.running_mean"],
                                   pre_sd["module.online_net.bn1.running_mean"])
            assert [s for s in md["optimizer"]["state"]] == [len(md["optimizer"]["param_groups"]) - 4,
                                                             len(md["optimizer"]["param_groups"]) - 3]
    assert accs["ft_all"] > 0.7          # 4 separable classes: far above the 0.25 of chance after 6 epochs
    tst = _load_script("test")
    opts = parse_opts(common + ["--task", "test", "--t_ft_task", "ft_all"])
    acc = tst.run(opts)
    out = capsys.readouterr().out
    assert "Video accuracy" in out and 0.5 < acc <= 1.0
    res = tmp_path / "synthetic" / "test_r21d_byol1_synthetic_1_RGB_4_plusone.txt"
    assert res.exists() and "Video accuracy" in res.read_text()
