"""Model factory -- the ``r21d_byol`` + DistributedDataParallel branch of
/root/reference/models/model.py:39-109 (``generate_model(opts) -> (model, parameters)``).

Kept: class selection by opts.model_name / opts.task, device placement on opts.local_rank,
the (degenerate) sync_bn flag, DDP wrap with find_unused_parameters=False, return of
``model.parameters()``.  The reference converts BN to SyncBatchNorm over a ONE-rank group
(models/model.py:95-96), i.e. statistics stay per-GPU: our BN kernel is exactly that local
BN, so --sync_bn 0/1 select the same arithmetic.  New: --model_depth picks the layer sizes.
"""
from __future__ import annotations

import torch

from .r21d_byol import R21DBYOL, layer_sizes_for_depth

PRETRAIN_TASKS = ("r_byol", "loss_com")


def generate_model(opts):
    if opts.model_name != "r21d_byol":
        raise ValueError("Please check the input backbone! (cstp_amd provides model_name=r21d_byol, got %r)"
                         % (opts.model_name,))
    if opts.task not in PRETRAIN_TASKS:
        raise NotImplementedError("cstp_amd covers the pre-training tasks %s; task %r (fine-tune/test/resume) is a "
                                  "later scope row" % (PRETRAIN_TASKS, opts.task))
    if not torch.cuda.is_available():
        raise RuntimeError("generate_model needs a HIP device: cstp_amd has no CPU execution path")
    model = R21DBYOL(pretrain=True, layer_sizes=layer_sizes_for_depth(opts.model_depth))
    local_rank = opts.local_rank if getattr(opts, "local_rank", -1) not in (-1, None) else 0
    torch.cuda.set_device(local_rank)
    model.cuda(local_rank)
    model.flatten_parameters()
    model.train()
    if getattr(opts, "distributed", False):
        # gradient all-reduce (mean) on RCCL over xGMI, bucketed + overlapped with backward by DDP;
        # BN-buffer broadcast from rank 0 at each forward kept (DDP default broadcast_buffers=True)
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], output_device=local_rank,
                                                          find_unused_parameters=False,
                                                          bucket_cap_mb=getattr(opts, "bucket_cap_mb", 25))
    return model, model.parameters()
