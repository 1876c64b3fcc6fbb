"""Model factory -- the ``r21d_byol`` branch of /root/reference/models/model.py:11-134
(``generate_model(opts) -> (model, parameters)``, ``neq_load_customized``).

Kept: class selection by opts.model_name / opts.task (pre-training: ``R21DBYOL(pretrain=True)``; ft_fc / ft_all /
scratch / test: ``R21DBYOL(pretrain=False, num_classes=opts.n_classes, cls_bn=True)``), device placement on
opts.local_rank, the (degenerate) sync_bn flag, the DDP wrap, the checkpoint handling per task:
  * test      -> ``load_state_dict(test_md['state_dict'])`` after ``assert opts.arch == test_md['arch']``; returns the
                 model alone (models/model.py:110-115);
  * resume    -> strict load of resume_md_path, returns (model, parameters) (:116-121);
  * ft_fc/all -> ``neq_load_customized`` of the pre-training checkpoint (the ``module.online_net.*`` keys carry over, the
                 projector / predictor / heads / target do not), then ``get_fine_tuning_parameters`` with
                 ft_begin_index 5 / 0 (:122-145).
The reference converts BN to SyncBatchNorm over a ONE-rank group (:95-96), i.e. statistics stay per-GPU: our BN kernel
is exactly that local BN, so --sync_bn 0/1 select the same arithmetic.  Without DDP the reference wraps the model in
nn.DataParallel, which on one device only adds the ``module.`` key prefix; ``SingleDeviceParallel`` keeps that prefix so
checkpoints are interchangeable between launch modes and with the reference.  New: --model_depth picks the layer sizes;
``--task resume`` (undefined model in the reference: neither task list of :44-49 contains it) builds the fine-tune model.
"""
from __future__ import annotations

import torch
from torch import nn

from .r21d_byol import R21DBYOL, get_fine_tuning_parameters, layer_sizes_for_depth
from .r3d_byol import R3DBYOL

PRETRAIN_TASKS = ("r_byol", "loss_com")
FINETUNE_TASKS = ("ft_fc", "ft_all", "scratch", "test", "resume")


class SingleDeviceParallel(nn.Module):
    """What nn.DataParallel is on one device: a pass-through whose state-dict keys start with ``module.``."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def neq_load_customized(model, pretrained_dict, verbose=True):
    """Load a checkpoint into a partially different model: keys present in both are taken from the checkpoint, the
    rest keep the model's values (models/model.py:11-36)."""
    model_dict = model.state_dict()
    tmp = {k: v for k, v in pretrained_dict.items() if k in model_dict}
    if verbose:
        print("\n=======Check Weights Loading======")
        print("Weights not used from pretrained file:")
        print("---------------------------")
        print("Weights not loaded into new model:")
        for k in model_dict:
            if k not in pretrained_dict:
                print(k)
        print("===================================\n")
    model_dict.update(tmp)
    model.load_state_dict(model_dict)
    return model


def _load_checkpoint(path, device):
    if not path:
        raise ValueError("this task needs a checkpoint path (--pretrained_path / --test_md_path / --resume_md_path)")
    return torch.load(path, map_location=device)


def generate_model(opts):
    if opts.model_name not in ("r21d_byol", "r3d_byol"):
        raise ValueError("Please check the input backbone! (cstp_amd provides model_name=r21d_byol | r3d_byol, got %r)"
                         % (opts.model_name,))
    if opts.task not in PRETRAIN_TASKS + FINETUNE_TASKS:
        raise ValueError("task %r: r21d_byol serves %s" % (opts.task, PRETRAIN_TASKS + FINETUNE_TASKS))
    if not torch.cuda.is_available():
        raise RuntimeError("generate_model needs a HIP device: cstp_amd has no CPU execution path")
    if opts.model_name == "r3d_byol":        # models/model.py:65-70: R3DBYOL(pretrain=..., [cls_bn=True,] opts=opts)
        if opts.task in PRETRAIN_TASKS:
            model = R3DBYOL(pretrain=True, opts=opts)
        else:
            model = R3DBYOL(pretrain=False, cls_bn=True, opts=opts)
    else:
        if getattr(opts, "act_dtype", "fp32") not in ("fp32", None):
            raise NotImplementedError("--act_dtype %s: the bf16-storage kernels serve --model_name r3d_byol (BASELINE configs[4]); "
                                      "r21d_byol runs fp32 storage" % (opts.act_dtype,))
        layer_sizes = layer_sizes_for_depth(opts.model_depth)
        if opts.task in PRETRAIN_TASKS:
            model = R21DBYOL(pretrain=True, layer_sizes=layer_sizes)
        else:
            model = R21DBYOL(pretrain=False, num_classes=opts.n_classes, cls_bn=True, layer_sizes=layer_sizes)
    local_rank = opts.local_rank if getattr(opts, "local_rank", -1) not in (-1, None) else 0
    torch.cuda.set_device(local_rank)
    model.cuda(local_rank)
    model.flatten_parameters()
    model.train()
    inner = model
    if opts.task == "ft_fc":
        # freeze BEFORE the DDP wrap so its reducer only registers the classifier (the reference freezes after the
        # wrap and needs find_unused_parameters=True for it, models/model.py:87-94)
        print("Fine-tune FC layer!")
        opts.ft_begin_index = 5
    elif opts.task == "ft_all":
        print("Fine-tune all layers")
        opts.ft_begin_index = 0
    frozen_plan = None
    if opts.task in ("ft_fc", "ft_all"):
        if opts.ft_begin_index != 0:    # substring match on names: the later ``module.`` prefix cannot change it
            frozen_plan = get_fine_tuning_parameters(inner, opts.ft_begin_index)
    if getattr(opts, "distributed", False):
        # gradient all-reduce (mean) on RCCL over xGMI.  DDP's default broadcast_buffers=True is kept, but the training
        # steps run forward/backward under no_sync(), which stops DDP's own buffer broadcast after the first forward: they
        # broadcast rank 0's BN buffers themselves at the top of every step (cstp_amd.train.sync_buffers)
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], output_device=local_rank,
                                                          find_unused_parameters=False,
                                                          bucket_cap_mb=getattr(opts, "bucket_cap_mb", 25))
    else:
        model = SingleDeviceParallel(model)

    if opts.task in ("scratch",) + PRETRAIN_TASKS:
        return model, model.parameters()
    device = torch.device("cuda", local_rank)
    if "test" in opts.task:
        print("Test model {}!".format(opts.test_md_path))
        test_md = _load_checkpoint(opts.test_md_path, device)
        assert opts.arch == test_md["arch"]
        model.load_state_dict(test_md["state_dict"])
        return model
    if opts.task == "resume":
        print("Resume model {}!".format(opts.resume_md_path))
        resume_md = _load_checkpoint(opts.resume_md_path, device)
        assert opts.arch == resume_md["arch"]
        model.load_state_dict(resume_md["state_dict"])
        return model, model.parameters()
    # ft_fc / ft_all
    checkpoint = _load_checkpoint(opts.pretrained_path, torch.device("cpu"))
    assert (opts.arch in checkpoint["arch"] or checkpoint["arch"] in opts.arch)
    print("adjust input weights according to new network")
    model = neq_load_customized(model, checkpoint["state_dict"], verbose=True)
    print("loaded pretrained checkpoint '{}' (epoch {})".format(opts.pretrained_path, checkpoint["epoch"]))
    parameters = frozen_plan if frozen_plan is not None else model.parameters()
    return model, parameters
