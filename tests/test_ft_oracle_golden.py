"""Pin the fine-tune / validation / test oracle (oracle/r21d_ft_oracle.py, fp32) and the host logic of that path
against golden vectors captured from the reference run in fp64 (tests/golden/make_golden_ft.py).  CPU only."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import r21d_byol_oracle as orc
from oracle import r21d_ft_oracle as ftorc

GOLD = os.path.join(os.path.dirname(__file__), "golden")
# (outputs/losses, per-tensor gradient & momentum, post-step state checksums) per optimisation step.  Step 1 outputs
# carry the parity bar (1e-4 rel of the fp64 truth); gradients go back through 8-16 train-mode BatchNorms over a batch
# of 4 plus the BatchNorm1d on L2-normalised features (values ~0.04), where stock fp32 sits up to 1e-2 from fp64;
# step 2 starts from step 1's perturbed weights.  Eval-mode outputs additionally carry the running statistics.
TOLS = {1: (1e-4, 2e-2, 2e-3), 2: (2e-2, 2e-1, 3e-2)}
VAL_TOLS = {1: 2e-3, 2: 3e-2}


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def cs_err(ours, ref):
    ours, ref = np.asarray(ours), np.asarray(ref)
    return float((np.abs(ours - ref).max(axis=1) / np.maximum(np.abs(ref[:, 1]), 1e-12)).max())


def checksums(sd, keys):
    return np.array([[float(sd[k].detach().double().sum()), float(sd[k].detach().double().abs().sum())] for k in keys])


@pytest.mark.parametrize("name", ["ft_all_d1", "ft_fc_d1", "ft_all_r18"])
def test_ft_oracle_matches_reference_golden(name):
    g = load(name)
    depth, b, t, hw, k, steps = [int(v) for v in g["meta"]]
    task = str(g["task"])
    ls = orc.layer_sizes_for_depth(depth)
    sd = ftorc.closed_form_state(ls, k, torch.float32)
    x_train, x_val, labels = ftorc.closed_form_batch(b, t, hw, k, torch.float32)
    assert np.array_equal(labels.numpy(), g["labels"])
    keys = [str(s) for s in g["state_keys"]]
    pkeys = [str(s) for s in g["param_keys"]]
    assert [s for s, _, _ in ftorc.ft_spec(ls, k)] == keys
    assert [s for s, _, kind in ftorc.ft_spec(ls, k) if orc.is_param(kind)] == pkeys
    train_keys = ftorc.trainable_keys(ls, k, task)
    assert [p in train_keys for p in pkeys] == [bool(v) for v in g["requires_grad"]]
    mom = {}
    for s in range(1, steps + 1):
        tol, gtol, stol = TOLS[s]
        pre = "s%d." % s
        info = ftorc.ft_train_step(sd, mom, x_train, labels, ls, k, float(g["lr"]), 0.9, float(g["wd"]), task)
        assert rel(float(info["loss"]), g[pre + "loss"]) < tol
        assert rel(info["logits"].numpy(), g[pre + "logits"]) < tol
        gn = np.array([float(info["grads"][p].norm()) if p in info["grads"] else -1.0 for p in pkeys])
        assert rel(gn, g[pre + "grad_norms"]) < gtol
        assert cs_err(checksums(sd, keys), g[pre + "state_cs"]) < stol
        mcs = np.array([[float(mom[p].double().sum()), float(mom[p].double().abs().sum())] if p in mom else [0.0, 0.0]
                        for p in pkeys])
        assert cs_err(mcs, g[pre + "mom_cs"]) < gtol
        # model.eval(): running statistics after s training forwards
        with torch.no_grad():
            val = ftorc.ft_forward(sd, x_val, ls, training=False)
        assert rel(val.numpy(), g[pre + "val_logits"]) < VAL_TOLS[s]
        assert rel(float(torch.nn.functional.cross_entropy(val, labels)), g[pre + "val_loss"]) < VAL_TOLS[s]
        mean, top5 = ftorc.video_prediction(sd, x_val, ls)
        assert rel(mean.numpy(), g[pre + "video_mean"]) < VAL_TOLS[s]
        if s == 1:
            assert np.array_equal(top5.numpy(), g[pre + "video_top5"])
    if task == "ft_fc":   # frozen tensors: untouched parameters, but BN buffers still move (train-mode forward)
        sd0 = ftorc.closed_form_state(ls, k, torch.float32)
        for p in pkeys:
            assert torch.equal(sd[p].detach(), sd0[p]) == (p not in train_keys), p
        assert not torch.equal(sd["online_net.bn1.running_mean"], sd0["online_net.bn1.running_mean"])
        assert int(sd["cls_bn.num_batches_tracked"]) == steps


def test_reduce_lr_on_plateau_matches_torch():
    from cstp_amd.scheduler import ReduceLROnPlateau

    class Opt:
        def __init__(self):
            self.param_groups = [{"lr": 0.05}, {"lr": 0.0}]

    g = load("ft_misc")
    for patience in (2, 10):
        opt = Opt()
        sch = ReduceLROnPlateau(opt, "min", patience=patience)
        lrs = []
        for l in g["plateau.%d.losses" % patience]:
            sch.step(float(l))
            lrs.append([grp["lr"] for grp in opt.param_groups])
        assert np.allclose(np.array(lrs), g["plateau.%d.lrs" % patience], rtol=1e-12, atol=0)
    assert len(set(g["plateau.2.lrs"][:, 0].tolist())) >= 3      # the fixture does exercise reductions
    with pytest.raises(ValueError):
        ReduceLROnPlateau(Opt(), factor=1.0)


def test_fine_tuning_parameters_and_model_keys():
    from cstp_amd.model import SingleDeviceParallel, neq_load_customized
    from cstp_amd.r21d_byol import R21DBYOL, get_fine_tuning_parameters
    g = load("ft_fc_d1")
    m = R21DBYOL(pretrain=False, num_classes=11, cls_bn=True)
    assert list(m.state_dict().keys()) == [str(s) for s in g["state_keys"]]
    assert get_fine_tuning_parameters(m, 0) is not None and all(p.requires_grad for p in m.parameters())
    plan = get_fine_tuning_parameters(SingleDeviceParallel(m), 5)
    assert [p.requires_grad for p in m.parameters()] == [bool(v) for v in g["requires_grad"]]
    assert [grp.get("lr", None) for grp in plan] == [None if v else 0.0 for v in g["requires_grad"]]
    # ft_begin_index 1..4 names 'layer<i>', which no module of this model carries: classifier only, as the reference
    m2 = R21DBYOL(pretrain=False, num_classes=11, cls_bn=True)
    get_fine_tuning_parameters(m2, 2)
    assert [n for n, p in m2.named_parameters() if p.requires_grad] == ["classify.weight", "classify.bias"]
    # pre-training checkpoint -> fine-tune model: encoder keys carry over, projector/heads/target are dropped
    torch.manual_seed(3)
    pre = SingleDeviceParallel(R21DBYOL(pretrain=True))
    ckpt = pre.state_dict()
    assert all(k.startswith("module.") for k in ckpt)
    ft = SingleDeviceParallel(R21DBYOL(pretrain=False, num_classes=11, cls_bn=True))
    before = {k: v.clone() for k, v in ft.state_dict().items()}
    neq_load_customized(ft, ckpt, verbose=False)
    after = ft.state_dict()
    for k in after:
        if k in ckpt:
            assert torch.equal(after[k], ckpt[k]), k
        else:
            assert k.startswith("module.classify") or k.startswith("module.cls_bn")
            assert torch.equal(after[k], before[k]), k
    assert "module.online_net.conv5.block1.conv2.temporal_conv.weight" in ckpt
    with pytest.raises(AttributeError):
        ft(torch.zeros(1), torch.zeros(1), o_type="loss_com")
    with pytest.raises(ValueError):
        ft(torch.zeros(1), o_type="scratch")     # r21d_byol.py:400-401


def test_flat_optimizer_runs_plan():
    """The launch plan of the flat optimizers: adjacent trainable tensors with equal hyper-parameters share one
    launch; frozen tensors are skipped (torch skips them because .grad is None)."""
    from cstp_amd.optim import FlatSGD
    sizes = [10, 6, 33, 8, 4]
    offs, n = [], 0
    for s in sizes:
        offs.append(n)
        n += (s + 3) // 4 * 4
    arena = {"param": torch.zeros(n), "grad": torch.zeros(n)}
    params = [torch.nn.Parameter(arena["param"][o:o + s]) for o, s in zip(offs, sizes)]
    for p, o, s in zip(params, offs, sizes):
        p.data = arena["param"][o:o + s]
    opt = FlatSGD(params, lr=0.1, momentum=0.9, weight_decay=1e-4, arenas=arena)
    assert [(r[0], r[1]) for r in opt._plan()] == [(0, n)]
    params[1].requires_grad = False
    groups = [{"params": p} if p.requires_grad else {"params": p, "lr": 0.0} for p in params]
    opt = FlatSGD(groups, lr=0.1, momentum=0.9, weight_decay=1e-4, arenas=arena)
    assert [(r[0], r[1]) for r in opt._plan()] == [(0, 12), (20, n - 20)]
    opt.param_groups[3]["lr"] = 0.01       # a different lr splits the run
    assert [(r[0], r[1]) for r in opt._plan()] == [(0, 12), (20, 36), (56, 8), (64, 4)]
    with pytest.raises(ValueError):
        FlatSGD([torch.nn.Parameter(torch.zeros(4))], lr=0.1, arenas=arena)
    with pytest.raises(ValueError):
        FlatSGD(params, lr=0.1, arenas=None)


def test_labelled_synthetic_dataset_and_loader():
    from cstp_amd.opts import parse_opts
    from cstp_amd.synthetic import SyntheticLabelledClips
    from cstp_amd.utils import calculate_accuracy, get_dataloader
    ds = SyntheticLabelledClips("train", length=10, sample_duration=4, sample_size=16, n_classes=7, seed=1)
    clip, label = ds[2]
    assert clip.shape == (3, 4, 16, 16) and clip.dtype == torch.float32 and float(clip.abs().max()) <= 1.0 and 0 <= label < 7
    assert torch.equal(ds[2][0], clip)
    clips, _ = SyntheticLabelledClips("test", 4, 4, 16, 7, 1, test_clips=3)[0]
    assert clips.shape == (3, 3, 4, 16, 16)
    o = parse_opts(["--batch_size", "4", "--n_workers", "0"])
    o.distributed = False
    loader, sampler = get_dataloader(ds, o, "train")
    assert sampler is None and len(loader) == 2            # drop_last on train
    loader, _ = get_dataloader(ds, o, "val")
    assert len(loader) == 3                                # ... not on val
    out = torch.tensor([[0.1, 0.9], [0.8, 0.2], [0.3, 0.7], [0.6, 0.4]])
    assert math.isclose(calculate_accuracy(out, torch.tensor([1, 0, 0, 0])), 0.75)


def _cpu_arena_params(sizes):
    offs, n = [], 0
    for s in sizes:
        offs.append(n)
        n += (int(np.prod(s)) + 3) // 4 * 4
    arena = {"param": torch.arange(n, dtype=torch.float32) * 0.01, "grad": torch.zeros(n)}
    params = []
    for s, o in zip(sizes, offs):
        p = torch.nn.Parameter(torch.empty(s))
        p.data = arena["param"][o:o + int(np.prod(s))].view(s)
        params.append(p)
    return arena, params, offs


def test_optimizer_state_dict_is_torch_wire_format():
    """Checkpoints carry ``optimizer.state_dict()`` (main_byol.py:139, main_ft_mp.py:291): what the flat optimizers
    write must load into torch.optim.SGD / Adam / AdamW (the reference's resume path, main_ft_mp.py:149-150) and step
    there, and what torch writes must load back."""
    from cstp_amd.optim import FlatAdam, FlatSGD
    sizes = [(3, 5), (6,), (2, 2, 3)]
    arena, params, offs = _cpu_arena_params(sizes)
    opt = FlatSGD(params, lr=0.05, momentum=0.9, weight_decay=5e-4, arenas=arena)
    opt._buf.copy_(torch.arange(opt._buf.numel(), dtype=torch.float32) * 0.5)
    opt._steps = 1
    sd = opt.state_dict()
    clones = [torch.nn.Parameter(p.detach().clone()) for p in params]
    ref = torch.optim.SGD(clones, lr=1.0, momentum=0.0)
    ref.load_state_dict(sd)
    assert ref.param_groups[0]["lr"] == 0.05 and ref.param_groups[0]["momentum"] == 0.9
    for (s, o), c in zip(zip(sizes, offs), clones):
        n = int(np.prod(s))
        assert torch.equal(ref.state[c]["momentum_buffer"], opt._buf[o:o + n].view(s))
        c.grad = torch.ones_like(c)
    ref.step()                                            # every key torch's step needs is present
    back = FlatSGD(params, lr=0.1, momentum=0.0, arenas=arena)
    back.load_state_dict(ref.state_dict())
    assert back._steps == 1 and back.param_groups[0]["lr"] == 0.05 and back.param_groups[0]["weight_decay"] == 5e-4
    for (s, o), c in zip(zip(sizes, offs), clones):
        n = int(np.prod(s))
        assert torch.equal(back._buf[o:o + n].view(s), ref.state[c]["momentum_buffer"])
    for decoupled, cls in ((False, torch.optim.Adam), (True, torch.optim.AdamW)):
        opt = FlatAdam(params, lr=0.01, betas=(0.9, 0.99), weight_decay=1e-2, decoupled=decoupled, arenas=arena)
        opt._m.fill_(0.25)
        opt._v.fill_(0.5)
        opt._steps = 3
        clones = [torch.nn.Parameter(p.detach().clone()) for p in params]
        ref = cls(clones, lr=1.0)
        ref.load_state_dict(opt.state_dict())
        assert ref.param_groups[0]["betas"] == (0.9, 0.99) and float(ref.state[clones[0]]["step"]) == 3
        for c in clones:
            c.grad = torch.ones_like(c)
        ref.step()
        back = FlatAdam(params, decoupled=decoupled, arenas=arena)
        back.load_state_dict(ref.state_dict())
        assert back._steps == 4 and back.param_groups[0]["lr"] == 0.01


def test_get_dataloader_shards_across_ranks():
    """utils.py:91-163 under DDP: GLOBAL batch split over ranks, disjoint shards, train drops the ragged tail."""
    from cstp_amd.opts import parse_opts
    from cstp_amd.synthetic import SyntheticLabelledClips
    from cstp_amd.utils import get_dataloader
    ds = SyntheticLabelledClips("train", length=22, sample_duration=2, sample_size=8, n_classes=3, seed=1)
    seen = []
    for rank in (0, 1):
        o = parse_opts(["--batch_size", "8", "--n_workers", "0"])
        o.distributed, o.world_size, o.rank = True, 2, rank
        loader, sampler = get_dataloader(ds, o, "train")
        sampler.set_epoch(3)
        assert loader.batch_size == 4 and o.batch_size == 8 and len(loader) == 2     # 11 per rank -> 2 full batches
        seen.append(set(iter(sampler)))
        vloader, vsampler = get_dataloader(ds, o, "val")
        assert len(vloader) == 3 and list(iter(vsampler)) == list(range(rank, 22, 2))
        o2 = parse_opts(["--batch_size", "8", "--n_workers", "0"])
        o2.distributed, o2.world_size, o2.rank = True, 2, rank
        get_dataloader(ds, o2, "byol")
        assert o2.batch_size == 4                                                    # utils.py:98 overwrites it
    assert seen[0].isdisjoint(seen[1]) and len(seen[0] | seen[1]) == 22


def test_optimizer_keeps_frozen_tensors_in_the_torch_index_space():
    """main_byol.py:228 builds optim.SGD(model.parameters()) over [online | TARGET (requires_grad False) | predictor |
    heads]: the frozen target tensors occupy param_groups / state indices although they never get state.  A checkpoint
    written here must load into torch.optim.SGD built over the same list (the reference's resume) with predictor / head
    momentum at the reference's indices, and a reference checkpoint must load back."""
    from cstp_amd.optim import FlatSGD
    sizes = [(3, 5), (6,), (2, 2, 3), (4,)]
    arena, trainable, offs = _cpu_arena_params(sizes)
    frozen = [torch.nn.Parameter(torch.randn(3, 5), requires_grad=False), torch.nn.Parameter(torch.randn(6), requires_grad=False)]
    # model.parameters() order: online (2 tensors), target (2, frozen, outside the arena), predictor + head (2)
    plist = trainable[:2] + frozen + trainable[2:]
    opt = FlatSGD(plist, lr=0.05, momentum=0.9, weight_decay=5e-4, arenas=arena)
    assert len(opt.param_groups[0]["params"]) == 6
    opt._buf.copy_(torch.arange(opt._buf.numel(), dtype=torch.float32) + 1.0)
    opt._steps = 1
    sd = opt.state_dict()
    assert sd["param_groups"][0]["params"] == [0, 1, 2, 3, 4, 5] and sorted(sd["state"]) == [0, 1, 4, 5]
    clones = [torch.nn.Parameter(p.detach().clone(), requires_grad=p.requires_grad) for p in plist]
    ref = torch.optim.SGD(clones, lr=1.0, momentum=0.0)
    ref.load_state_dict(sd)                                 # raises on a param-count mismatch
    for idx, ti in ((0, 0), (1, 1), (4, 2), (5, 3)):
        n = int(np.prod(sizes[ti]))
        assert torch.equal(ref.state[clones[idx]]["momentum_buffer"], opt._buf[offs[ti]:offs[ti] + n].view(sizes[ti]))
    assert clones[2] not in ref.state and clones[3] not in ref.state
    for c in clones:
        if c.requires_grad:
            c.grad = torch.ones_like(c)
    ref.step()
    back = FlatSGD(plist, lr=0.1, momentum=0.0, arenas=arena)
    back.load_state_dict(ref.state_dict())
    for idx, ti in ((0, 0), (1, 1), (4, 2), (5, 3)):
        n = int(np.prod(sizes[ti]))
        assert torch.equal(back._buf[offs[ti]:offs[ti] + n].view(sizes[ti]), ref.state[clones[idx]]["momentum_buffer"])
    # the launch plan covers the trainable arena as ONE run and never touches the frozen tensors
    assert [(o, n) for o, n, _, _ in back._plan()] == [(0, arena["param"].numel())]
    # a trainable tensor outside the arena is still an error
    with pytest.raises(ValueError):
        FlatSGD(trainable + [torch.nn.Parameter(torch.zeros(3))], lr=0.1, arenas=arena)
