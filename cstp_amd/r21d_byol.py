"""R(2+1)D-BYOL for MI355X -- host-side mirror of the reference model interface.

Mirrors (same class names, attribute names, state-dict keys, argument meaning and error
behaviour) /root/reference/models/pace/r21d_byol.py:
  get_fine_tuning_parameters :10-35, SpatioTemporalConv :38-97, SpatioTemporalResBlock :100-148,
  SpatioTemporalResLayer :151-181, R2Plus1DNet :184-229, Projector :232-243, Predictor :246-257, R21DBYOL :260-401
  (pretrain=True: o_type "loss_com"; pretrain=False: o_type "ft_fc" / "ft_all" / "test" -- online_net -> F.normalize ->
  BatchNorm1d(512) -> Linear(512, num_classes), :293-299,394-399; BN uses running statistics under model.eval()).
The module tree only holds parameters/buffers and sequences kernel launches; all arithmetic
runs in the HIP kernels of libcstp_hip.so (cstp_amd.ops).  Differences from the reference,
all additive:
  * ``layer_sizes`` is a constructor argument of R21DBYOL (the reference hard-codes (1,1,1,1),
    :268-269) so --model_depth 18/34 means R(2+1)D-18/34;
  * BN+ReLU and BN+residual+ReLU are single kernels; optionally (CSTP_FUSE_BN=1) a BN+ReLU that feeds exactly
    one convolution is folded into that convolution's gather and never materialised;
  * parameters live in flat HBM arenas (``flatten_parameters``) so EMA / clip / SGD are one
    streaming kernel each instead of ~80 tiny ones (:331-337 rebinding .data per tensor).
"""
from __future__ import annotations

import math
import os
from typing import List, Sequence, Tuple

import torch
import torch.nn as nn

from . import ops

LAYER_SIZES = {1: (1, 1, 1, 1), 18: (2, 2, 2, 2), 34: (3, 4, 6, 3)}
# Fold BN+ReLU into the consumer convolution's gather (ops.bn_relu_conv3d) instead of materialising it.
FUSE_BN_INTO_CONV = os.environ.get("CSTP_FUSE_BN", "0") == "1"
# Default path: the BatchNorm + ReLU between a stride-1 3x3 spatial convolution and its temporal convolution is applied inside
# the temporal convolution's gather (forward and weight gradient, igemm_k1s / igemm_k2s <.., AFF>) wherever the spatial
# convolution's kernel left the statistics AND the range of its output (igemm_k1p<MT, true>): the normalised mid-channel
# tensor -- the largest activations of the network -- is never written or re-read.  CSTP_FUSE_BN_T=0 materialises it.
FUSE_BN_TEMPORAL = os.environ.get("CSTP_FUSE_BN_T", "1") == "1"


def _temporal_fused(x, conv, groups) -> bool:
    # asked on every call (a plan lookup in the library): the answer follows the layer's CURRENT tiles, which the autotuner may
    # still change during the first steps
    return ops.in_affine_fused(x.shape, conv.weight.shape, conv.stride, conv.padding, groups)


# Run the (no-grad) target-network forward on a second HIP stream, staggered behind the online network's stem + conv2 stage
# (R21DBYOL.forward): -2.5 ms/step at cfg2 on MI355X.  CSTP_OVERLAP_TARGET=0 runs the two forwards back to back.
OVERLAP_TARGET_FORWARD = os.environ.get("CSTP_OVERLAP_TARGET", "1") == "1"


def layer_sizes_for_depth(depth: int) -> Tuple[int, int, int, int]:
    if int(depth) not in LAYER_SIZES:
        raise ValueError("r21d_byol supports --model_depth in %s, got %r" % (sorted(LAYER_SIZES), depth))
    return LAYER_SIZES[int(depth)]


def _triple(v):
    return (v, v, v) if isinstance(v, int) else tuple(v)


def _pad4(n):
    return (n + 3) // 4 * 4


def get_fine_tuning_parameters(model, ft_begin_index):
    """r21d_byol.py:10-35.  ft_begin_index 0: every parameter.  Otherwise only parameters whose NAME contains one of
    'layer<i>' (i = ft_begin_index..4) or 'classify' stay trainable -- and since this model's stages are called
    conv1..conv5, that is the classifier alone for every non-zero index (kept as the reference behaves).  Every other
    parameter is frozen (requires_grad False) and listed with lr 0.0, one param group per tensor."""
    if ft_begin_index == 0:
        return model.parameters()
    ft_module_names = []
    if ft_begin_index <= 4:
        for i in range(ft_begin_index, 5):
            ft_module_names.append("layer{}".format(i))
    ft_module_names.append("classify")
    print("Modules to finetune : ", ft_module_names)
    parameters = []
    for k, v in model.named_parameters():
        if any(name in k for name in ft_module_names):
            print("Layers to finetune : ", k)
            parameters.append({"params": v})
        else:
            v.requires_grad = False
            parameters.append({"params": v, "lr": 0.0})
    return parameters


# ---------------------------------------------------------------------------------------------
# leaf modules: parameter holders with nn.Conv3d / nn.BatchNormNd / nn.Linear compatible
# attribute names and default initialisers (so the CPU RNG stream is consumed identically)
# ---------------------------------------------------------------------------------------------
class Conv3d(nn.Module):
    """nn.Conv3d(bias=False) stand-in; weight [out, in, kt, kh, kw]."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False):
        super().__init__()
        if bias:
            raise NotImplementedError("the CSTP path uses bias-free convolutions (r21d_byol.py:52)")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = _triple(kernel_size), _triple(stride), _triple(padding)
        self.weight = nn.Parameter(torch.empty((out_channels, in_channels) + self.kernel_size))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))  # nn.Conv3d.reset_parameters

    def forward(self, x, bn_groups=0, bn_pivot=None, grad_join=None):
        """``bn_groups`` > 0: a train-mode BatchNorm with that many groups consumes the result; ``bn_pivot``: its running
        mean, the value its statistics are summed around; ``grad_join``: x feeds a second op too (ops.conv3d)."""
        return ops.conv3d(x, self.weight, None, self.stride, self.padding, bn_groups, bn_pivot, grad_join)


class _BatchNorm(nn.Module):
    """Batch norm with optional fused residual add and ReLU (batch statistics in train mode, running statistics
    in eval mode)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, x, residual=None, relu=False, groups=1, grad_join=None):
        if not self.training:   # model.eval(): running statistics, nothing updated (validation / test)
            return ops.batch_norm_eval(x, self.weight, self.bias, self.running_mean, self.running_var, residual, relu, self.eps)
        if x.numel() // x.shape[1] // groups <= 1:
            # same failure the reference hits in nn.BatchNorm*d (train mode)
            raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (tuple(x.shape),))
        y = ops.batch_norm_act(x, self.weight, self.bias, self.running_mean, self.running_var, residual, relu, self.eps,
                               self.momentum, groups, grad_join)
        if not getattr(self, "_nbt_in_arena", False):
            self.num_batches_tracked += groups   # else: one add per net per forward, see R21DBYOL.forward
        return y

    def relu_then(self, conv, x, groups=1, bn_groups=0, bn_pivot=None):
        """conv(relu(self(x))) with this BN's apply+ReLU folded into ``conv``'s gather: only the statistics
        pass touches x before the convolution, and relu(bn(x)) is never written to HBM.  ``bn_groups`` / ``bn_pivot``: the
        BatchNorm behind ``conv`` (ops.conv3d)."""
        if not self.training:
            return conv(self(x, relu=True))
        y = ops.bn_relu_conv3d(x, self.weight, self.bias, self.running_mean, self.running_var, conv.weight, conv.stride,
                               conv.padding, groups, True, self.eps, self.momentum, bn_groups, bn_pivot)
        if not getattr(self, "_nbt_in_arena", False):
            self.num_batches_tracked += groups
        return y


class BatchNorm3d(_BatchNorm):
    pass


class BatchNorm1d(_BatchNorm):
    pass


class Linear(nn.Module):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))  # nn.Linear.reset_parameters
        bound = 1 / math.sqrt(in_features)
        nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class ReLU(nn.Module):
    """Placeholder keeping the reference's Sequential indices (net.2); ReLU itself is fused into BN."""

    def forward(self, x):  # pragma: no cover - never called on the fused path
        raise RuntimeError("ReLU is fused into the preceding BatchNorm kernel")


class _MLP(nn.Sequential):
    """Linear -> BatchNorm1d -> ReLU -> Linear with the reference's Sequential indices 0,1,2,3."""

    def __init__(self, dim, hidden, out):
        super().__init__(Linear(dim, hidden), BatchNorm1d(hidden), ReLU(), Linear(hidden, out))

    def forward(self, x, groups=1):
        if self.training and x.shape[0] // groups < 2:
            # same failure the reference hits in nn.BatchNorm1d (train mode)
            raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (tuple(x.shape),))
        h = self[0](x)
        h = self[1](h, relu=True, groups=groups)
        return self[3](h)


# ---------------------------------------------------------------------------------------------
# the reference module hierarchy
# ---------------------------------------------------------------------------------------------
class SpatioTemporalConv(nn.Module):
    """(2+1)D factored conv: spatial 1xkxk -> BN -> ReLU -> temporal tx1x1 (r21d_byol.py:38-97)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False, first_conv=False):
        super().__init__()
        kernel_size, stride, padding = _triple(kernel_size), _triple(stride), _triple(padding)
        intermed_channels = int(math.floor(
            (kernel_size[0] * kernel_size[1] * kernel_size[2] * in_channels * out_channels) /
            (kernel_size[1] * kernel_size[2] * in_channels + kernel_size[0] * out_channels)))
        self.spatial_conv = Conv3d(in_channels, intermed_channels, (1, kernel_size[1], kernel_size[2]),
                                   stride=(1, stride[1], stride[2]), padding=(0, padding[1], padding[2]), bias=bias)
        self.bn = BatchNorm3d(intermed_channels)
        self.relu = ReLU()
        self.temporal_conv = Conv3d(intermed_channels, out_channels, (kernel_size[0], 1, 1), stride=(stride[0], 1, 1),
                                    padding=(padding[0], 0, 0), bias=bias)

    def forward(self, x, groups=1, pre_bn=None, grad_join=None, out_bn=None):
        """temporal_conv(relu(bn(spatial_conv(x)))) (r21d_byol.py:94-97).  ``grad_join``: x feeds a second op of the block
        as well (the residual / the shortcut): their gradients are summed inside the ops (ops.GradJoin).  ``pre_bn``: the BatchNorm whose
        apply+ReLU precedes this module in the block (bn1 -> relu1 -> conv2, :142-143).  ``out_bn``: the BatchNorm that consumes this
        module's output (bn1 / bn2 / downsamplebn, :142-147): in train mode the temporal convolution leaves its statistics where
        its kernel can.
        With FUSE_BN_INTO_CONV each BN+ReLU is folded into the following convolution's gather (the normalised
        tensor is never written: -36 % BN traffic, -7 ms/step of BN kernels, -7 GB of activations at cfg2) --
        but the per-element affine costs the MFMA kernels' gather more than the two HBM passes it removes
        (+8 ms/step on MI355X, profiles/r01), so the default materialises BN outputs."""
        if FUSE_BN_INTO_CONV:
            x = self.spatial_conv(x) if pre_bn is None else pre_bn.relu_then(self.spatial_conv, x, groups)
            return self.bn.relu_then(self.temporal_conv, x, groups)
        if pre_bn is not None:
            x = pre_bn(x, relu=True, groups=groups)
        x = self.spatial_conv(x, groups if self.bn.training else 0, self.bn.running_mean, grad_join if pre_bn is None else None)
        og = groups if (out_bn is not None and out_bn.training) else 0
        opv = out_bn.running_mean if og else None
        if (FUSE_BN_TEMPORAL and self.bn.training and ops._bnstats_of(x, groups) is not None
                and _temporal_fused(x, self.temporal_conv, groups)):
            return self.bn.relu_then(self.temporal_conv, x, groups, og, opv)
        return self.temporal_conv(self.bn(x, relu=True, groups=groups), og, opv)


class SpatioTemporalResBlock(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, downsample=False):
        super().__init__()
        self.downsample = downsample
        padding = kernel_size // 2
        if self.downsample:
            self.downsampleconv = SpatioTemporalConv(in_channels, out_channels, 1, stride=2)
            self.downsamplebn = BatchNorm3d(out_channels)
            self.conv1 = SpatioTemporalConv(in_channels, out_channels, kernel_size, padding=padding, stride=2)
        else:
            self.conv1 = SpatioTemporalConv(in_channels, out_channels, kernel_size, padding=padding)
        self.bn1 = BatchNorm3d(out_channels)
        self.relu1 = ReLU()
        self.conv2 = SpatioTemporalConv(out_channels, out_channels, kernel_size, padding=padding)
        self.bn2 = BatchNorm3d(out_channels)
        self.outrelu = ReLU()

    def forward(self, x, groups=1):
        # the block's input feeds two ops -- conv1 and the residual addition (or, in a downsample block, conv1 and the shortcut
        # convolution): their two gradients are summed inside the second op's kernel instead of by a separate add pass
        join = ops.GradJoin(2) if (self.training and torch.is_grad_enabled() and x.requires_grad and not FUSE_BN_INTO_CONV) else None
        # conv2(relu1(bn1(conv1(x))))
        res = self.conv2(self.conv1(x, groups, grad_join=join, out_bn=self.bn1), groups, pre_bn=self.bn1, out_bn=self.bn2)
        if self.downsample:
            x = self.downsamplebn(self.downsampleconv(x, groups, grad_join=join, out_bn=self.downsamplebn), groups=groups)
            join = None
        # relu(x + bn2(res)) as one kernel (r21d_byol.py:143,148)
        return self.bn2(res, residual=x, relu=True, groups=groups, grad_join=join)


class SpatioTemporalResLayer(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, layer_size, block_type=SpatioTemporalResBlock,
                 downsample=False):
        super().__init__()
        self.block1 = block_type(in_channels, out_channels, kernel_size, downsample)
        self.blocks = nn.ModuleList([])
        for _ in range(layer_size - 1):
            self.blocks += [block_type(out_channels, out_channels, kernel_size)]

    def forward(self, x, groups=1):
        x = self.block1(x, groups)
        for block in self.blocks:
            x = block(x, groups)
        return x


class Projector(nn.Module):
    def __init__(self, dim, projection_size, projection_hidden_size=4096):
        super().__init__()
        self.net = _MLP(dim, projection_hidden_size, projection_size)

    def forward(self, x, groups=1):
        return self.net(x, groups)


class Predictor(nn.Module):
    def __init__(self, dim, prediction_size, prediction_hidden_size=4096):
        super().__init__()
        self.net = _MLP(dim, prediction_hidden_size, prediction_size)

    def forward(self, x, groups=1):
        return self.net(x, groups)


class R2Plus1DNet(nn.Module):
    def __init__(self, layer_sizes=(1, 1, 1, 1), block_type=SpatioTemporalResBlock, proj_flag=False):
        super().__init__()
        self.conv1 = SpatioTemporalConv(3, 64, (3, 7, 7), stride=(1, 2, 2), padding=(1, 3, 3))
        self.bn1 = BatchNorm3d(64)
        self.relu1 = ReLU()
        self.conv2 = SpatioTemporalResLayer(64, 64, 3, layer_sizes[0], block_type=block_type)
        self.conv3 = SpatioTemporalResLayer(64, 128, 3, layer_sizes[1], block_type=block_type, downsample=True)
        self.conv4 = SpatioTemporalResLayer(128, 256, 3, layer_sizes[2], block_type=block_type, downsample=True)
        self.conv5 = SpatioTemporalResLayer(256, 512, 3, layer_sizes[3], block_type=block_type, downsample=True)
        self.proj_flag = proj_flag
        if self.proj_flag:
            self.project = Projector(dim=512, projection_size=512, projection_hidden_size=4096)

    # gradient stages in the order the backward pass completes them (ByolBase.grad_stage_slices): everything behind the
    # encoder's last stage, then conv5 .. conv2, then the stem
    GRAD_STAGES = ("head", "conv5", "conv4", "conv3", "conv2", "stem")

    def forward(self, x, groups=1, after_conv2=None, stage_done=None):
        """``groups`` > 1: x holds that many independent forward calls back to back along the batch axis
        (BN statistics stay per call); convolutions are per-sample, so the result equals separate calls.
        ``after_conv2``: called once the conv2 stage is enqueued (R21DBYOL starts the target network's stream there).
        ``stage_done(i)``: called DURING BACKWARD when the gradient of stage i's input exists, i.e. when every parameter
        gradient of GRAD_STAGES[i] (and of the stages before it) has been enqueued -- the data-parallel step starts that
        slice's all-reduce there instead of after the whole backward pass (cstp_amd.train.StagedAllReduce)."""
        def mark(t, i):
            if stage_done is not None and t.requires_grad:
                t.register_hook(lambda g, i=i: stage_done(i))
            return t
        x = mark(self.bn1(self.conv1(x, groups), relu=True, groups=groups), 4)      # conv2's input: conv2 is complete
        x = mark(self.conv2(x, groups), 3)
        if after_conv2 is not None:
            after_conv2()
        x = mark(self.conv3(x, groups), 2)
        x = mark(self.conv4(x, groups), 1)
        x = mark(self.conv5(x, groups), 0)                                         # projector, predictor, heads are complete
        x = ops.global_avg_pool(x)  # AdaptiveAvgPool3d(1) + view(-1, 512)
        if self.proj_flag:
            return x, self.project(x, groups)
        return x


class ByolBase(nn.Module):
    """What the BYOL wrappers of both backbones share (R21DBYOL here, R3DBYOL in r3d_byol.py): the Glorot re-initialisation,
    the flat parameter / gradient / target arenas, the EMA and the regression loss.  Subclasses provide ``online_net``,
    ``target_net`` (pretrain), ``pretrain`` and ``_head_bn_calls()``."""

    _arenas = None
    _grad_stage_cb = None      # set by the data-parallel training step: called with a stage index during backward

    def grad_stage_slices(self):
        """[(offset, numel)] of the flat gradient arena per gradient stage, in the order the backward pass completes them
        (R2Plus1DNet.GRAD_STAGES): [project | predictor | heads], conv5, conv4, conv3, conv2, stem.  parameters() order is
        module registration order, so every stage is one contiguous run of the arena; together they tile it exactly.
        None when the model has no arenas or its encoder does not mark stages (the step then reduces the arena in one piece)."""
        net = self.online_net
        if self._arenas is None or not hasattr(net, "GRAD_STAGES") or not getattr(net, "proj_flag", False):
            return None
        sizes = [(p.numel() + 3) // 4 * 4 for p in self.trainable_parameters()]
        owner = {}
        for name in ("conv1", "bn1"):
            for p in getattr(net, name).parameters():
                owner[id(p)] = 5
        for i, name in ((4, "conv2"), (3, "conv3"), (2, "conv4"), (1, "conv5")):
            for p in getattr(net, name).parameters():
                owner[id(p)] = i
        runs, off = [], 0
        for p, n in zip(self.trainable_parameters(), sizes):
            st = owner.get(id(p), 0)           # projector, predictor, heads
            if runs and runs[-1][0] == st:
                runs[-1][2] += n
            else:
                runs.append([st, off, n])
            off += n
        if sorted(r[0] for r in runs) != list(range(6)) or off != self._arenas["grad"].numel():
            return None                        # a stage is not contiguous: fall back to the flat reduce
        by_stage = {r[0]: (r[1], r[2]) for r in runs}
        return [by_stage[i] for i in range(6)]

    def _head_bn_calls(self):
        """[(module holding BatchNorms outside the two encoders, forward calls per training step)] (pretrain only)."""
        raise NotImplementedError

    def _glorot_all(self, leaf_types):
        # Glorot-uniform overwrite of every Linear/Conv3d/BatchNorm weight, in modules() order
        # (r21d_byol.py:301-329, r3d_byol.py:265-273) -- BN gamma becomes U(+-sqrt(6/C)), not 1.
        for m in self.modules():
            if isinstance(m, leaf_types):
                self._glorot_uniform(m.weight)

    # -- init helpers (r21d_byol.py:311-329) ---------------------------------------------------
    @staticmethod
    def _calculate_fan_in_and_fan_out(tensor):
        if tensor.dim() < 2:
            return int(tensor.size(0) / 2), int(tensor.size(0) / 2)
        receptive = tensor[0][0].numel() if tensor.dim() > 2 else 1
        return tensor.size(1) * receptive, tensor.size(0) * receptive

    @torch.no_grad()
    def _glorot_uniform(self, tensor):
        fan_in, fan_out = self._calculate_fan_in_and_fan_out(tensor)
        std = math.sqrt(6.0 / float(fan_in + fan_out))
        return tensor.uniform_(-std, std)

    def _set_grad(self, model, val):
        for p in model.parameters():
            p.requires_grad = val

    # -- flat HBM arenas --------------------------------------------------------------------------
    def trainable_parameters(self) -> List[nn.Parameter]:
        """The parameters an optimizer may update, in parameters() order: everything but the EMA target network
        (pretrain: online_net, predictor, heads; fine-tune: online_net, classify, cls_bn -- frozen or not, so the
        arena layout does not depend on the fine-tune task)."""
        if not self.pretrain:
            return list(self.parameters())
        tgt = {id(p) for p in self.target_net.parameters()}
        return [p for p in self.parameters() if id(p) not in tgt]

    @torch.no_grad()
    def flatten_parameters(self):
        """Re-home parameters, gradients and BN counters into flat arenas on the module's device:
          train arena  = [online_net params | predictor | heads]   (+ same-layout grad arena)
          target arena = [target_net params]  with the layout of the online_net prefix
        Each tensor starts on a 16-byte boundary (float4 streaming).  Idempotent."""
        if self._arenas is not None:
            return self._arenas
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("flatten_parameters() needs the model on a HIP device (call .cuda() first)")

        def layout(params):
            offs, n = [], 0
            for p in params:
                offs.append(n)
                n += (p.numel() + 3) // 4 * 4
            return offs, n

        train = self.trainable_parameters()
        online = list(self.online_net.parameters())
        target = list(self.target_net.parameters()) if self.pretrain else []
        assert (not self.pretrain) or [p.shape for p in online] == [p.shape for p in target]
        assert all(a is b for a, b in zip(train[:len(online)], online)), "online_net must lead the trainable order"
        offs, n_train = layout(train)
        _, n_enc = layout(online)
        p_arena = torch.zeros(n_train, dtype=torch.float32, device=dev)
        g_arena = torch.zeros(n_train, dtype=torch.float32, device=dev)
        t_arena = torch.zeros(n_enc if self.pretrain else 0, dtype=torch.float32, device=dev)
        for p, o in zip(train, offs):
            v = p_arena[o:o + p.numel()].view_as(p)
            v.copy_(p.data)
            p.data = v
            p.grad = g_arena[o:o + p.numel()].view_as(p)
        for p, o in zip(target, offs[:len(target)]):
            v = t_arena[o:o + p.numel()].view_as(p)
            v.copy_(p.data)
            p.data = v
        # BN running statistics: ONE float arena (running_mean | running_var of every BatchNorm, modules() order) and ONE
        # int64 arena for the num_batches_tracked counters (a single add_ per net per forward instead of 24+).  Flat, so
        # that DDP's buffer broadcast from rank 0 (models/model.py:97-103, broadcast_buffers=True) is two collectives
        # (broadcast_buffers_) instead of one per tensor.
        bns = [m for m in self.modules() if isinstance(m, _BatchNorm)]
        b_arena = torch.zeros(sum(2 * _pad4(m.num_features) for m in bns), dtype=torch.float32, device=dev)
        o = 0
        for m in bns:
            for name in ("running_mean", "running_var"):
                v = b_arena[o:o + m.num_features]
                v.copy_(getattr(m, name))
                setattr(m, name, v)
                o += _pad4(m.num_features)
        if self.pretrain:
            nets = (("online", self.online_net), ("target", self.target_net), ("heads", None))
        else:
            nets = (("all", self),)    # fine-tune: every BN (encoder + cls_bn) runs once per training forward
        groups, head_inc = [], []
        for name, net in nets:
            if net is not None:
                mods = [m for m in net.modules() if isinstance(m, _BatchNorm)]
            else:
                mods = []
                for h, calls in self._head_bn_calls():
                    hm = [m for m in h.modules() if isinstance(m, _BatchNorm)]
                    mods += hm
                    head_inc += [calls] * len(hm)
            groups.append((name, mods))
        nbt_all = torch.zeros(sum(len(mods) for _, mods in groups), dtype=torch.long, device=dev)
        nbt, o = {}, 0
        for name, mods in groups:
            arena = nbt_all[o:o + len(mods)]
            for i, m in enumerate(mods):
                arena[i] = m.num_batches_tracked
                m.num_batches_tracked = arena[i]
                m._nbt_in_arena = True
            nbt[name] = arena
            o += len(mods)
        # forward() calls per step: predictor x2, overlap_spa x1, overlap_tem x1, pb_cls x2, rotate_cls x2
        if self.pretrain:
            nbt["heads_inc"] = torch.tensor(head_inc, dtype=torch.long, device=dev)
        self._arenas = {"param": p_arena, "grad": g_arena, "target": t_arena, "n_encoder": n_enc, "nbt": nbt,
                        "buffers": b_arena, "nbt_all": nbt_all}
        return self._arenas

    def broadcast_buffers_(self, src=0, group=None):
        """DDP's per-forward buffer broadcast (models/model.py:97-103: DistributedDataParallel default
        broadcast_buffers=True -> rank 0's BN running statistics and counters overwrite every other rank's at the start
        of each forward) as TWO collectives over the flat buffer arenas.  No-op without an initialised process group of
        more than one rank.  The training steps (cstp_amd.train) call it at the top of every step because they run
        forward/backward under ``ddp.no_sync()``, which switches DDP's own broadcast off after the first step."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
            return False
        if self._arenas is not None:
            dist.broadcast(self._arenas["buffers"], src, group=group)
            dist.broadcast(self._arenas["nbt_all"], src, group=group)
        else:
            for b in self.buffers():
                dist.broadcast(b, src, group=group)
        return True

    def _side_stream(self, device):
        st = getattr(self, "_side", None)
        if st is None or st.device != device:
            st = torch.cuda.Stream(device=device)
            self._side = st
        return st

    def _update_target_net(self):
        """EMA of the online encoder+projector into the target (r21d_byol.py:331-337)."""
        if self._arenas is not None:
            a = self._arenas
            ops.ema_update_(a["target"], a["param"][:a["n_encoder"]], self.momentum)
            if ops.pack_plan is not None:
                ops.pack_plan.replay("target")          # the target network's weight packs of this step: one launch behind the EMA
        else:
            for pq, pk in zip(self.online_net.parameters(), self.target_net.parameters()):
                ops.ema_update_(pk.data, pq.data, self.momentum)

    def _loss_fn(self, x, y):
        return ops.byol_regression_loss(x, y)

    def _cal_loss(self, online_feat_1, online_feat_2, target_feat_1, target_feat_2):
        return self._loss_fn(online_feat_1, target_feat_2) + self._loss_fn(online_feat_2, target_feat_1)


class R21DBYOL(ByolBase):
    """forward(x1, x2, o_type='loss_com') -> (loss_byol, (pred_spa, pred_tem, pred_pb_1, pred_pb_2,
    pred_rot_1, pred_rot_2)) exactly as r21d_byol.py:357-382."""

    def __init__(self, pretrain=True, momentum=0.996, layer_sizes=(1, 1, 1, 1), **kwargs):
        super().__init__()
        self.pretrain = bool(pretrain)
        self.layer_sizes = tuple(layer_sizes)
        if pretrain:
            self.momentum = momentum
            self.online_net = R2Plus1DNet(layer_sizes=self.layer_sizes, proj_flag=True)
            self.target_net = R2Plus1DNet(layer_sizes=self.layer_sizes, proj_flag=True)
            self.predictor = Predictor(dim=512, prediction_size=512, prediction_hidden_size=4096)
            self._set_grad(self.target_net, False)
            self.overlap_spa = _MLP(1024, 1024, 5)
            self.overlap_tem = _MLP(1024, 1024, 5)
            self.pb_cls = _MLP(512, 512, 5)
            self.rotate_cls = _MLP(512, 512, 5)
        else:
            # fine-tune / test model (r21d_byol.py:293-299): encoder without projector + classifier
            self.online_net = R2Plus1DNet(layer_sizes=self.layer_sizes, proj_flag=False)
            self.classify = Linear(512, kwargs["num_classes"])
            self.cls_bn = kwargs["cls_bn"]
            if self.cls_bn:
                print("classify_bn is true, Feature norm and Batch norm on final features")
                self.cls_bn = BatchNorm1d(512)
        self._glorot_all((Linear, Conv3d, BatchNorm1d, BatchNorm3d))
        self._arenas = None

    def _head_bn_calls(self):
        # forward() calls per step: predictor x2, overlap_spa x1, overlap_tem x1, pb_cls x2, rotate_cls x2
        return [(self.predictor, 2), (self.overlap_spa, 1), (self.overlap_tem, 1), (self.pb_cls, 2), (self.rotate_cls, 2)]

    def forward(self, x1, x2=None, o_type=None):
        if o_type == "loss_com":
            if not self.pretrain:
                raise AttributeError("R21DBYOL(pretrain=False) has no target_net/predictor: o_type='loss_com' needs "
                                     "pretrain=True")
            if x2 is None or x2.shape != x1.shape:
                raise ValueError("o_type='loss_com' needs two clips of identical shape")
            b = x1.shape[0]
            # Both views go through ONE launch sequence per network as a batch of 2B with two BN groups:
            # convolutions are per-sample and BN statistics stay per view, so this is the reference's
            # online_net(x1); online_net(x2) (r21d_byol.py:359-360) with half the launches, one weight
            # pack per layer and twice the grid on the small deep layers.
            x = torch.cat((x1, x2), dim=0)
            if OVERLAP_TARGET_FORWARD and x.is_cuda:
                # The target network's forward depends on nothing the online forward produces (the EMA reads the online
                # PARAMETERS, which no forward modifies), so it runs on a second HIP stream: its HBM-bound BatchNorm kernels
                # execute beside the other network's matrix-core-bound convolutions instead of alternating with them.
                main = torch.cuda.current_stream(x.device)
                side = self._side_stream(x.device)
                tgt = {}

                def start_target():
                    # STAGGERED: the side stream starts once the online network's stem and conv2 stage (the big 56x56 layers,
                    # where the dominant kernel lives) are through, so those launches run alone -- their HIP-event timing stays
                    # the kernel's own duration -- and the target forward overlaps the online conv3..conv5 stages and heads.
                    side.wait_stream(main)
                    with torch.cuda.stream(side), torch.no_grad():
                        self._update_target_net()              # EMA BEFORE the target forward (:364)
                        _, target_proj = self.target_net(x, groups=2)
                        tgt["swapped"] = torch.cat((target_proj[b:], target_proj[:b]), dim=0).detach()

                online_feat, online_proj = self.online_net(x, groups=2, after_conv2=start_target,
                                                           stage_done=self._grad_stage_cb)
                online_pred = self.predictor(online_proj, groups=2)
                main.wait_stream(side)
                target_swapped = tgt["swapped"]
                target_swapped.record_stream(main)
                x.record_stream(side)
            else:
                online_feat, online_proj = self.online_net(x, groups=2, stage_done=self._grad_stage_cb)
                online_pred = self.predictor(online_proj, groups=2)
                with torch.no_grad():
                    self._update_target_net()                      # EMA BEFORE the target forward (:364)
                    _, target_proj = self.target_net(x, groups=2)  # train-mode BN, own running stats (:365-366)
                    target_swapped = torch.cat((target_proj[b:], target_proj[:b]), dim=0).detach()
            # loss_fn(pred_1, tproj_2) + loss_fn(pred_2, tproj_1)  (:351-355)
            rows = self._loss_fn(online_pred, target_swapped)
            loss = rows[:b] + rows[b:]
            online_feat_1, online_feat_2 = online_feat[:b], online_feat[b:]
            feat_cat = torch.cat((online_feat_1, online_feat_2), dim=1)
            pred_spa = self.overlap_spa(feat_cat)
            pred_tem = self.overlap_tem(feat_cat)
            pred_pb = self.pb_cls(online_feat, groups=2)
            pred_rot = self.rotate_cls(online_feat, groups=2)
            pred_pb_1, pred_pb_2 = pred_pb[:b], pred_pb[b:]
            pred_rot_1, pred_rot_2 = pred_rot[:b], pred_rot[b:]
            if self._arenas is not None:   # BN num_batches_tracked: three adds instead of 108
                nbt = self._arenas["nbt"]
                nbt["online"] += 2
                nbt["target"] += 2
                nbt["heads"] += nbt["heads_inc"]
            # kept for the NT-Xent head and for parity tests (no extra work)
            self.last_projections = (online_proj[:b], online_proj[b:])
            return loss.mean(), (pred_spa, pred_tem, pred_pb_1, pred_pb_2, pred_rot_1, pred_rot_2)
        elif o_type == "r_byol":
            raise NotImplementedError("o_type='r_byol' is shape-broken in the reference (predictor fed a tuple, "
                                      "r21d_byol.py:384-385); use o_type='loss_com'")
        elif o_type in ["ft_fc", "ft_all", "test"]:
            if self.pretrain:
                raise AttributeError("R21DBYOL(pretrain=True) has no classify/cls_bn: o_type=%r needs pretrain=False" % o_type)
            online_feat = self.online_net(x1)                       # r21d_byol.py:395 (proj_flag False: features only)
            online_feat = ops.l2_normalize(online_feat)             # F.normalize(p=2, dim=1) :396
            if self.cls_bn is False or self.cls_bn is None:
                raise TypeError("'bool' object is not callable")    # the reference calls self.cls_bn unconditionally (:397)
            online_feat = self.cls_bn(online_feat)                  # :397
            out = self.classify(online_feat)                        # :398
            if self.training and self._arenas is not None:
                self._arenas["nbt"]["all"] += 1
            return out
        else:
            raise ValueError("Output cls is not exist!")
