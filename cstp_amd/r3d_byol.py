"""3D-ResNet-BYOL for MI355X -- host-side mirror of /root/reference/models/BE/r3d_byol.py (the backbone swap of
BASELINE.json configs[4]): BasicBlock depths 10 / 18 / 34 and the Bottleneck depth 50 (/ 101 / 152).

Mirrors (same class names, attribute names, state-dict keys, argument meaning and error behaviour):
  conv3x3x3 :45-53, BasicBlock :69-97, ResNet :139-206 (7x7x7 stem stride (1,2,2) -> BN -> ReLU -> MaxPool3d(3, 2, 1) ->
  layer1..4 -> AdaptiveAvgPool3d(1) -> view(-1, 512)), Predictor :223-234, R3DBYOL :237-433 (o_type 'loss_com' :381-405,
  'ft_fc' / 'ft_all' / 'test' :420-428, 'scratch' :429-432), resnet10/18/34 :436-455, get_fine_tuning_parameters :18-42.
Differences from the R(2+1)D wrapper that the reference makes and this file keeps: the encoder has no projector (the
predictor and the target comparison act on the 512-d features), target_net is a deepcopy of online_net (identical initial
weights), the pretext heads are plain Linear layers with 4-way playback-rate / rotation outputs, the fine-tune BatchNorm is
called ``classify_bn``.  Shortcut type 'A' (:56-66) builds a CPU tensor inside forward and is refused -- 'B' (1x1x1 conv + BN)
is the default (opts.py: --sc_type B).

Bottleneck depths (configs[4] names 3D-ResNet-50).  The reference's Bottleneck BACKBONE (:100-137, ResNet.__init__ :139-191) is
sound and is mirrored module for module (pinned: tests/golden/r3d_50_backbone.npz drives the reference's layers up to the
average pool).  Its WRAPPER cannot run at these depths: ``x.view(-1, 512)`` (:204) turns the [B, 2048] pooled features into
[4B, 512], and Projector / Predictor / the four heads / classify are hard-wired to 512 inputs (:212,226,249-252,262), so the
logits come out with 4B rows against B labels.  SPEC of the corrected wrapper implemented here (PARITY-UNPINNED -- the
reference has nothing to compare with): feature width F = 512 x block.expansion replaces every literal 512 that means "the
encoder's output": ``view(-1, F)``; Predictor Linear(F, 4096) -> BN1d -> ReLU -> Linear(4096, F) (BYOL compares the prediction
with the F-wide target feature); overlap_spa / overlap_tem Linear(2F, 5); pb_cls / rot_cls Linear(F, 4); classify_bn
BatchNorm1d(F); classify Linear(F, n_classes).  For F = 512 this is the reference, key for key.

bf16 STORAGE (configs[4] says bf16; ``opts.act_dtype == "bf16"``; pre-training, and the fine-tune / validation / test forwards): the clip is rounded to bf16 once and every
5-D activation / activation gradient of the encoders is bf16 in HBM, arithmetic fp32, parameters / gradients / statistics /
pooled features / heads / losses fp32 -- the spec in include/cstp_hip.h ("bf16-STORAGE path") and csrc/b16.hip, restated by
oracle/r3d_byol_oracle.py (storage="bf16").  cstp_amd.ops dispatches on the dtype of the activation tensor.

All arithmetic runs in the HIP kernels of libcstp_hip.so through cstp_amd.ops: the 3x3x3 convolutions are the 27-tap case of the
implicit-GEMM kernels, the stem the 343-tap / 3-channel case, MaxPool3d has its own kernel pair.
"""
from __future__ import annotations

import copy
import os

import torch
import torch.nn as nn

from . import ops
from .r21d_byol import (OVERLAP_TARGET_FORWARD, BatchNorm1d, BatchNorm3d, ByolBase, Conv3d, Linear, Predictor, ReLU,  # noqa: F401
                        get_fine_tuning_parameters)

LAYERS = {10: (1, 1, 1, 1), 18: (2, 2, 2, 2), 34: (3, 4, 6, 3)}
BOTTLENECK_LAYERS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}     # resnet50/101/152 :458-479


def conv3x3x3(in_planes, out_planes, stride=1):
    return Conv3d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


RESIDUAL_JOIN = os.environ.get("CSTP_R3D_JOIN", "1") != "0"      # A/B switch: 0 = autograd's add pass sums the two gradients


def _residual_join(block, x):
    """A block's input feeds conv1 AND the residual addition (or the shortcut convolution): in a training forward its two gradients
    are summed inside the ops (ops.GradJoin: the second contributor adds in its kernel's epilogue) instead of by autograd's add pass."""
    # (bf16 storage only: 16 ATen adds per 3D-ResNet-50 step gone, 19.58 -> 19.50 ms; with fp32 storage the accumulating epilogue
    #  costs what the add pass did -- 28.26 vs 28.18 ms -- and the blocks keep autograd's add: profiles/r04/ab_r3d_gradjoin.log)
    return ops.GradJoin(2) if (RESIDUAL_JOIN and x.dtype == torch.bfloat16 and block.training and torch.is_grad_enabled()
                               and x.requires_grad) else None


class _Downsample(nn.Sequential):
    """nn.Sequential(Conv3d 1x1x1 stride s, BatchNorm3d): keys ``downsample.0`` / ``downsample.1`` (r3d_byol.py:177-183)."""

    def __init__(self, inplanes, planes, stride):
        super().__init__(Conv3d(inplanes, planes, kernel_size=1, stride=stride, bias=False), BatchNorm3d(planes))

    def forward(self, x, groups=1, grad_join=None):
        return self[1](self[0](x, grad_join=grad_join), groups=groups)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3x3(inplanes, planes, stride)
        self.bn1 = BatchNorm3d(planes)
        self.relu = ReLU()
        self.conv2 = conv3x3x3(planes, planes)
        self.bn2 = BatchNorm3d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x, groups=1):
        join = _residual_join(self, x)
        out = self.bn1(self.conv1(x, grad_join=join), relu=True, groups=groups)            # relu(bn1(conv1 x))   :84-86
        out = self.conv2(out)
        if self.downsample is None:
            return self.bn2(out, residual=x, relu=True, groups=groups, grad_join=join)   # relu(bn2(.) + residual)  :88-95, one kernel
        return self.bn2(out, residual=self.downsample(x, groups, grad_join=join), relu=True, groups=groups)


class Bottleneck(nn.Module):
    """1x1x1 -> BN -> ReLU -> 3x3x3 (stride) -> BN -> ReLU -> 1x1x1 (x4 channels) -> BN -> (+ residual) -> ReLU
    (r3d_byol.py:100-137).  The 1x1x1 convolutions are the one-tap case of the implicit-GEMM kernels (pure channel GEMMs: the
    MFMA 1x1 path north_star names); BN + residual + ReLU is one kernel."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv3d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = BatchNorm3d(planes)
        self.conv2 = Conv3d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = BatchNorm3d(planes)
        self.conv3 = Conv3d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = BatchNorm3d(planes * 4)
        self.relu = ReLU()
        self.downsample = downsample
        self.stride = stride

    def forward(self, x, groups=1):
        join = _residual_join(self, x)
        out = self.bn1(self.conv1(x, grad_join=join), relu=True, groups=groups)
        out = self.bn2(self.conv2(out), relu=True, groups=groups)
        out = self.conv3(out)
        if self.downsample is None:
            return self.bn3(out, residual=x, relu=True, groups=groups, grad_join=join)
        return self.bn3(out, residual=self.downsample(x, groups, grad_join=join), relu=True, groups=groups)


class _Stage(nn.Sequential):
    def forward(self, x, groups=1):
        for block in self:
            x = block(x, groups)
        return x


class ResNet(nn.Module):
    def __init__(self, block, layers, sample_size=224, sample_duration=16, shortcut_type="B", num_classes=400):
        super().__init__()
        if shortcut_type != "B":
            raise NotImplementedError("shortcut type %r: cstp_amd implements the reference default 'B' (1x1x1 conv + BN); 'A' "
                                      "allocates a CPU tensor inside forward (r3d_byol.py:56-66)" % (shortcut_type,))
        self.inplanes = 64
        self.feat_dim = 512 * block.expansion
        self.conv1 = Conv3d(3, 64, kernel_size=7, stride=(1, 2, 2), padding=(3, 3, 3), bias=False)
        self.bn1 = BatchNorm3d(64)
        self.relu = ReLU()
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = _Downsample(self.inplanes, planes * block.expansion, stride)
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return _Stage(*layers)

    def forward(self, x, groups=1, after_layer1=None):
        """``groups`` > 1: x holds that many independent forward calls back to back along the batch axis (per-call BN
        statistics); convolutions and pooling are per-sample, so the result equals separate calls.
        ``after_layer1``: called once layer1 is enqueued (R3DBYOL starts the target network's stream there)."""
        x = self.bn1(self.conv1(x), relu=True, groups=groups)
        x = ops.max_pool3d(x, 3, 2, 1)
        x = self.layer1(x, groups)
        if after_layer1 is not None:
            after_layer1()
        x = self.layer2(x, groups)
        x = self.layer3(x, groups)
        x = self.layer4(x, groups)
        return ops.global_avg_pool(x)      # AdaptiveAvgPool3d(1) + view(-1, 512) -- view(-1, feat_dim) at the Bottleneck depths


def _resnet(depth, **kwargs):
    if int(depth) in LAYERS:
        return ResNet(BasicBlock, LAYERS[int(depth)], **kwargs)
    if int(depth) in BOTTLENECK_LAYERS:
        return ResNet(Bottleneck, BOTTLENECK_LAYERS[int(depth)], **kwargs)
    raise ValueError("r3d_byol supports --model_depth %s, got %r" % (sorted(LAYERS) + sorted(BOTTLENECK_LAYERS), depth))


def resnet10(**kwargs):
    return _resnet(10, **kwargs)


def resnet18(**kwargs):
    return _resnet(18, **kwargs)


def resnet34(**kwargs):
    return _resnet(34, **kwargs)


def resnet50(**kwargs):
    return _resnet(50, **kwargs)


class R3DBYOL(ByolBase):
    """forward(x1, x2, o_type='loss_com') -> (loss_byol, (pred_spa, pred_tem, pred_pb_1, pred_pb_2, pred_rot_1, pred_rot_2))
    with [B,5], [B,5], [B,4] x4 logits (r3d_byol.py:381-405)."""

    def __init__(self, momentum=0.996, pretrain=True, cls_bn=False, opts=None):
        super().__init__()
        self.pretrain = bool(pretrain)
        act = getattr(opts, "act_dtype", "fp32") or "fp32"
        if act not in ("fp32", "bf16"):
            raise ValueError("--act_dtype %r: fp32 | bf16" % (act,))
        self.act_bf16 = act == "bf16"
        kw = dict(sample_size=opts.sample_size, sample_duration=opts.sample_duration, shortcut_type=opts.sc_type,
                  num_classes=opts.n_classes)
        if pretrain:
            self.momentum = momentum
            self.online_net = _resnet(opts.model_depth, **kw)
            f = self.online_net.feat_dim         # 512 for the BasicBlock depths (= the reference); 2048 at depth 50 (spec above)
            self.target_net = copy.deepcopy(self.online_net)
            self.predictor = Predictor(dim=f, prediction_size=f, prediction_hidden_size=4096)
            self._set_grad(self.target_net, False)
            self.overlap_spa = Linear(2 * f, 5)
            self.overlap_tem = Linear(2 * f, 5)
            self.pb_cls = Linear(f, 4)
            self.rot_cls = Linear(f, 4)
        else:
            self.online_net = _resnet(opts.model_depth, **kw)
            f = self.online_net.feat_dim
            self.cls_bn = cls_bn
            if self.cls_bn:
                self.classify_bn = BatchNorm1d(f)
            self.classify = Linear(f, opts.n_classes)
        self._glorot_all((Linear, Conv3d, BatchNorm1d, BatchNorm3d))   # :265-273 (the deep-copied target is re-drawn too)
        self._arenas = None

    def _head_bn_calls(self):
        return [(self.predictor, 2)]

    def forward(self, x1, x2=None, o_type="r_byol"):
        if o_type == "loss_com":
            if not self.pretrain:
                raise AttributeError("R3DBYOL(pretrain=False) has no target_net/predictor: o_type='loss_com' needs pretrain=True")
            if x2 is None or x2.shape != x1.shape:
                raise ValueError("o_type='loss_com' needs two clips of identical shape")
            b = x1.shape[0]
            x = torch.cat((x1, x2), dim=0)     # both views through one launch sequence, per-view BN statistics (groups=2)
            if self.act_bf16:
                x = ops.to_bf16(x)             # bf16 storage: ops dispatch on the activation dtype from here on
            if OVERLAP_TARGET_FORWARD and x.is_cuda:
                # target forward on a second HIP stream, staggered behind the online stem + layer1 (see R21DBYOL.forward)
                main = torch.cuda.current_stream(x.device)
                side = self._side_stream(x.device)
                tgt = {}

                def start_target():
                    side.wait_stream(main)
                    with torch.cuda.stream(side), torch.no_grad():
                        self._update_target_net()             # EMA BEFORE the target forward (:388)
                        target_feat = self.target_net(x, groups=2)
                        tgt["swapped"] = torch.cat((target_feat[b:], target_feat[:b]), dim=0).detach()

                online_feat = self.online_net(x, groups=2, after_layer1=start_target)
                online_pred = self.predictor(online_feat, groups=2)
                main.wait_stream(side)
                target_swapped = tgt["swapped"]
                target_swapped.record_stream(main)
                x.record_stream(side)
            else:
                online_feat = self.online_net(x, groups=2)
                online_pred = self.predictor(online_feat, groups=2)
                with torch.no_grad():
                    self._update_target_net()                     # EMA BEFORE the target forward (:388)
                    target_feat = self.target_net(x, groups=2)
                    target_swapped = torch.cat((target_feat[b:], target_feat[:b]), dim=0).detach()
            rows = self._loss_fn(online_pred, target_swapped)   # loss_fn(pred_1, t_2) + loss_fn(t_1, pred_2)  (:317-321)
            loss = rows[:b] + rows[b:]
            f1, f2 = online_feat[:b], online_feat[b:]
            feat_cat = torch.cat((f1, f2), dim=1)
            pred_spa = self.overlap_spa(feat_cat)
            pred_tem = self.overlap_tem(feat_cat)
            pred_pb = self.pb_cls(online_feat)
            pred_rot = self.rot_cls(online_feat)
            if self._arenas is not None:
                nbt = self._arenas["nbt"]
                nbt["online"] += 2
                nbt["target"] += 2
                nbt["heads"] += nbt["heads_inc"]
            self.last_projections = (online_feat[:b], online_feat[b:])   # NT-Xent head input (no projector in this wrapper)
            return loss.mean(), (pred_spa, pred_tem, pred_pb[:b], pred_pb[b:], pred_rot[:b], pred_rot[b:])
        if o_type == "r_byol":
            raise NotImplementedError("o_type='r_byol' reads an attribute the reference never sets (self.shuffle_bn, "
                                      "r3d_byol.py:410); use o_type='loss_com'")
        if o_type in ["ft_fc", "ft_all", "test", "scratch"]:
            if self.pretrain:
                raise AttributeError("R3DBYOL(pretrain=True) has no classify: o_type=%r needs pretrain=False" % o_type)
            online_feat = self.online_net(ops.to_bf16(x1) if self.act_bf16 else x1)     # pooled features are fp32 either way
            if o_type != "scratch" and self.cls_bn:             # :420-428 vs :429-432
                online_feat = ops.l2_normalize(online_feat)
                online_feat = self.classify_bn(online_feat)
            out = self.classify(online_feat)
            if self.training and self._arenas is not None:
                self._arenas["nbt"]["all"] += 1
                if self.cls_bn and o_type == "scratch":
                    self.classify_bn.num_batches_tracked -= 1    # not called on the scratch branch (:429-432)
            return out
        return None     # the reference falls off the end of forward for any other o_type
