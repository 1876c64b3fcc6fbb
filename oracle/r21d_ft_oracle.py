"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the CSTP fine-tune / validation / test path.

Functional restatement (flat ``dict`` of tensors keyed like the reference ``state_dict``) executed with stock PyTorch
CPU ops; pinned against golden vectors captured from the reference itself (``tests/golden/make_golden_ft.py`` imports
``/root/reference`` in the build container; fixtures ``tests/golden/ft_*.npz``).  Nothing in ``cstp_amd`` imports it.

What each function follows (paths relative to /root/reference):

* ``ft_spec``            models/pace/r21d_byol.py:293-299 (R21DBYOL(pretrain=False): online_net without projector,
                          classify = Linear(512, num_classes), cls_bn = BatchNorm1d(512)), registration order
* ``ft_forward``         models/pace/r21d_byol.py:394-399 (o_type 'ft_fc' / 'ft_all' / 'test'); ``training=False`` is
                          model.eval(): every BatchNorm uses its running statistics
* ``trainable_keys``     models/pace/r21d_byol.py:10-35 (get_fine_tuning_parameters) as models/model.py:123-145 calls it:
                          ft_all -> index 0 -> everything; ft_fc -> index 5 -> names containing 'classify'
* ``ft_train_step``      main_ft_mp.py:199-212 (CrossEntropy, zero_grad, backward, SGD step; no clipping)
* ``accuracy``           utils.py:58-66
* ``video_prediction``   test.py:79-82 (mean of clip logits, top-5)
"""
from __future__ import annotations

from collections import OrderedDict
from typing import List

import torch
import torch.nn.functional as F

from . import r21d_byol_oracle as base


def ft_spec(layer_sizes, num_classes: int):
    spec = base.encoder_spec("online_net", layer_sizes, proj=False)
    spec += [("classify.weight", (num_classes, 512), "lin_w"), ("classify.bias", (num_classes,), "lin_b")]
    spec += base._bn_spec("cls_bn", 512)
    return spec


def closed_form_state(layer_sizes, num_classes: int, dtype=torch.float32) -> "OrderedDict[str, torch.Tensor]":
    sd = OrderedDict()
    for key, shape, kind in ft_spec(layer_sizes, num_classes):
        t = base.closed_form_tensor(key, shape, kind)
        sd[key] = t if kind == "buf_nbt" else t.to(dtype)
    return sd


def closed_form_batch(b: int, t: int, hw: int, num_classes: int, dtype=torch.float32, seed_phase: int = 0):
    """(train clips, held-out clips, labels): the two clips of ``closed_form_clips`` and labels (7j+3) mod K."""
    x1, x2, _ = base.closed_form_clips(b, t, hw, dtype, seed_phase)
    j = torch.arange(b, dtype=torch.int64)
    return x1, x2, (j * 7 + 3) % num_classes


def trainable_keys(layer_sizes, num_classes: int, task: str) -> List[str]:
    keys = [k for k, _, kind in ft_spec(layer_sizes, num_classes) if base.is_param(kind)]
    if task == "ft_all":
        return keys
    if task == "ft_fc":
        return [k for k in keys if "classify" in k]
    raise ValueError(task)


def ft_forward(sd, x, layer_sizes, training=True):
    feat = base.encoder_forward(sd, "online_net", x, layer_sizes, training, proj=False)
    feat = F.normalize(feat, p=2, dim=1)
    feat = base._bn(sd, "cls_bn", feat, training)
    return F.linear(feat, sd["classify.weight"], sd["classify.bias"])


def accuracy(outputs, targets) -> float:
    return float((outputs.argmax(dim=1) == targets).float().sum()) / targets.shape[0]


def ft_train_step(sd, mom, x, targets, layer_sizes, num_classes, lr, momentum=0.9, weight_decay=0.0, task="ft_all"):
    """One fine-tune step; ``sd`` tensors are mutated/replaced, ``mom`` is the SGD momentum dict."""
    keys = trainable_keys(layer_sizes, num_classes, task)
    for k in keys:
        sd[k] = sd[k].detach().requires_grad_(True)
    out = ft_forward(sd, x, layer_sizes, True)
    loss = F.cross_entropy(out, targets)
    grads = torch.autograd.grad(loss, [sd[k] for k in keys])
    out_grads = {}
    with torch.no_grad():
        for k, g in zip(keys, grads):
            out_grads[k] = g.detach().clone()
            p = sd[k].detach()
            if weight_decay != 0:
                g = g + weight_decay * p
            mom[k] = mom[k] * momentum + g if k in mom else g.clone()
            sd[k] = p - lr * mom[k]
    return {"loss": loss.detach(), "logits": out.detach(), "acc": accuracy(out.detach(), targets), "grads": out_grads}


def video_prediction(sd, clips, layer_sizes):
    with torch.no_grad():
        out = ft_forward(sd, clips, layer_sizes, training=False)
    mean = out.mean(dim=0, keepdim=True)
    return mean, mean.topk(min(5, mean.shape[1]), 1, True)[1][0]
