"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the CSTP R(2+1)D-BYOL pre-training step.

A functional restatement (no nn.Module; a flat ``dict`` of tensors keyed like the
reference ``state_dict``) of the reference hot path, executed with stock PyTorch
CPU ops.  It is pinned against golden vectors captured from the reference itself
(``tests/golden/make_golden.py`` imports ``/root/reference`` in the build
container; fixtures in ``tests/golden/*.npz``).  Nothing in ``cstp_amd`` imports it.

What each function follows (paths relative to /root/reference):

* ``intermed_channels``       models/pace/r21d_byol.py:74-76
* ``st_conv``                 models/pace/r21d_byol.py:94-97   (SpatioTemporalConv.forward)
* ``res_block``               models/pace/r21d_byol.py:141-148 (SpatioTemporalResBlock.forward)
* ``encoder_forward``         models/pace/r21d_byol.py:215-229 (R2Plus1DNet.forward)
* ``mlp``                     models/pace/r21d_byol.py:232-257, 276-291 (Projector/Predictor/heads)
* ``ema_update``              models/pace/r21d_byol.py:331-337
* ``byol_loss``               models/pace/r21d_byol.py:346-355
* ``model_forward``           models/pace/r21d_byol.py:357-382 (o_type == "loss_com")
* ``loss_total``              main_byol.py:62-73
* ``train_step``              main_byol.py:60-91 (zero_grad, backward, clip_grad_norm_ 18, SGD)
* ``ntxent``                  loss/NTXent.py:23-62
* ``cosine_warmup_lrs``       scheduler/cosine_anneal.py:46-88 as driven by main_byol.py:252-269
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
EMA_MOMENTUM = 0.996
CLIP_VALUE = 18.0

LAYER_SIZES = {1: (1, 1, 1, 1), 18: (2, 2, 2, 2), 34: (3, 4, 6, 3)}


def layer_sizes_for_depth(depth: int) -> Tuple[int, int, int, int]:
    return LAYER_SIZES[int(depth)]


def intermed_channels(cin: int, cout: int, k: Tuple[int, int, int]) -> int:
    kt, kh, kw = k
    return int(math.floor((kt * kh * kw * cin * cout) / (kh * kw * cin + kt * cout)))


# --------------------------------------------------------------------------- #
# state-dict specification (registration order == reference parameters() order)
# --------------------------------------------------------------------------- #
def _bn_spec(prefix: str, c: int):
    return [
        (prefix + ".weight", (c,), "bn_w"),
        (prefix + ".bias", (c,), "bn_b"),
        (prefix + ".running_mean", (c,), "buf_mean"),
        (prefix + ".running_var", (c,), "buf_var"),
        (prefix + ".num_batches_tracked", (), "buf_nbt"),
    ]


def _stconv_spec(prefix: str, cin: int, cout: int, k: Tuple[int, int, int]):
    m = intermed_channels(cin, cout, k)
    spec = [(prefix + ".spatial_conv.weight", (m, cin, 1, k[1], k[2]), "conv_w")]
    spec += _bn_spec(prefix + ".bn", m)
    spec += [(prefix + ".temporal_conv.weight", (cout, m, k[0], 1, 1), "conv_w")]
    return spec


def _block_spec(prefix: str, cin: int, cout: int, downsample: bool):
    spec = []
    if downsample:
        spec += _stconv_spec(prefix + ".downsampleconv", cin, cout, (1, 1, 1))
        spec += _bn_spec(prefix + ".downsamplebn", cout)
    spec += _stconv_spec(prefix + ".conv1", cin, cout, (3, 3, 3))
    spec += _bn_spec(prefix + ".bn1", cout)
    spec += _stconv_spec(prefix + ".conv2", cout, cout, (3, 3, 3))
    spec += _bn_spec(prefix + ".bn2", cout)
    return spec


def _mlp_spec(prefix: str, din: int, dhid: int, dout: int):
    spec = [(prefix + ".0.weight", (dhid, din), "lin_w"), (prefix + ".0.bias", (dhid,), "lin_b")]
    spec += _bn_spec(prefix + ".1", dhid)
    spec += [(prefix + ".3.weight", (dout, dhid), "lin_w"), (prefix + ".3.bias", (dout,), "lin_b")]
    return spec


def encoder_spec(prefix: str, layer_sizes: Sequence[int], proj: bool = True):
    spec = _stconv_spec(prefix + ".conv1", 3, 64, (3, 7, 7))
    spec += _bn_spec(prefix + ".bn1", 64)
    chans = [(64, 64, False), (64, 128, True), (128, 256, True), (256, 512, True)]
    for li, ((cin, cout, ds), n) in enumerate(zip(chans, layer_sizes)):
        lp = "%s.conv%d" % (prefix, li + 2)
        spec += _block_spec(lp + ".block1", cin, cout, ds)
        for bi in range(n - 1):
            spec += _block_spec("%s.blocks.%d" % (lp, bi), cout, cout, False)
    if proj:     # proj_flag (r21d_byol.py:211-213)
        spec += _mlp_spec(prefix + ".project.net", 512, 4096, 512)
    return spec


def model_spec(layer_sizes: Sequence[int]):
    """Ordered (key, shape, kind) of R21DBYOL(pretrain=True) -- r21d_byol.py:265-291."""
    spec = encoder_spec("online_net", layer_sizes)
    spec += encoder_spec("target_net", layer_sizes)
    spec += _mlp_spec("predictor.net", 512, 4096, 512)
    spec += _mlp_spec("overlap_spa", 1024, 1024, 5)
    spec += _mlp_spec("overlap_tem", 1024, 1024, 5)
    spec += _mlp_spec("pb_cls", 512, 512, 5)
    spec += _mlp_spec("rotate_cls", 512, 512, 5)
    return spec


def is_param(kind: str) -> bool:
    return not kind.startswith("buf")


def trainable_keys(layer_sizes) -> List[str]:
    return [k for k, _, kind in model_spec(layer_sizes) if is_param(kind) and not k.startswith("target_net.")]


def encoder_param_pairs(layer_sizes) -> List[Tuple[str, str]]:
    on = [k for k, _, kind in encoder_spec("online_net", layer_sizes) if is_param(kind)]
    return [(k, "target_net" + k[len("online_net"):]) for k in on]


# --------------------------------------------------------------------------- #
# closed-form deterministic fill (shared by golden generator, tests and smoke)
# --------------------------------------------------------------------------- #
def _name_seed(name: str) -> int:
    h = 0
    for ch in name:
        h = (h * 131 + ord(ch)) % 1000003
    return h


def hash_uniform(n: int, seed: int) -> torch.Tensor:
    """n values in [-1, 1): an exact 32-bit integer hash of the element index (bit-identical on
    every platform -- no libm involved), returned as float64."""
    m32 = 0xFFFFFFFF
    h = (torch.arange(n, dtype=torch.int64) * 2654435761 + (seed * 40503 + 12345)) & m32
    h = ((h ^ (h >> 16)) * 73244475) & m32
    h = ((h ^ (h >> 16)) * 73244475) & m32
    h = h ^ (h >> 16)
    return h.to(torch.float64) / 2147483648.0 - 1.0


def _glorot_bound(shape) -> float:
    if len(shape) < 2:
        fi = fo = int(shape[0] / 2)
    else:
        rf = 1
        for s in shape[2:]:
            rf *= s
        fi, fo = shape[1] * rf, shape[0] * rf
    return math.sqrt(6.0 / float(fi + fo))


def hash_pow2(n: int, seed: int, span: int = 21) -> torch.Tensor:
    """n exact powers of two 2^0 .. 2^-(span-1), the exponent drawn by the integer hash: a log-uniform magnitude over
    six decades (2^-20 ~ 1e-6) with no libm call, so the fill is bit-identical everywhere."""
    m32 = 0xFFFFFFFF
    h = (torch.arange(n, dtype=torch.int64) * 2246822519 + (seed * 9176 + 777)) & m32
    h = ((h ^ (h >> 15)) * 2654435761) & m32
    h = h ^ (h >> 13)
    return torch.ldexp(torch.ones(n, dtype=torch.float64), -(h % span).to(torch.int32))


def hash_mask(n: int, seed: int) -> torch.Tensor:
    """n values in {0, 1}, half of them zero (post-ReLU-like sparsity)."""
    m32 = 0xFFFFFFFF
    h = (torch.arange(n, dtype=torch.int64) * 3266489917 + (seed * 374761 + 393)) & m32
    h = ((h ^ (h >> 15)) * 2246822519) & m32
    return ((h >> 11) & 1).to(torch.float64)


def closed_form_tensor(name: str, shape, kind: str, heavy: bool = False) -> torch.Tensor:
    """``heavy``: every weight / BN gamma is additionally multiplied by its own power of two from 2^0 .. 2^-20 (x 4, which
    restores the RMS): magnitudes inside ONE tensor then span six decades -- the case a per-tensor operand scale
    (the 2xf16-split GEMM kernels of the HIP path) has to survive."""
    n = 1
    for s in shape:
        n *= s
    if kind == "buf_nbt":
        return torch.zeros((), dtype=torch.int64)
    if kind == "buf_mean":
        return torch.zeros(shape, dtype=torch.float64)
    if kind == "buf_var":
        return torch.ones(shape, dtype=torch.float64)
    u = hash_uniform(n, _name_seed(name))
    if heavy and kind in ("conv_w", "lin_w", "bn_w"):
        u = u * hash_pow2(n, _name_seed(name)) * 4.0
    if kind in ("conv_w", "lin_w", "bn_w"):
        # the reference re-initialises conv/linear weights AND BN gamma ~ U(+-sqrt(6/(fi+fo)))
        # (r21d_byol.py:301-329); mimic the scale so the dynamics are the reference's
        v = _glorot_bound(shape) * u
    elif kind == "bn_b":
        v = 0.05 * u
    elif kind == "lin_b":
        v = u / math.sqrt(float(shape[0]))
    else:
        raise ValueError(kind)
    return v.reshape(shape)


def closed_form_state(layer_sizes, dtype=torch.float32, heavy: bool = False) -> "OrderedDict[str, torch.Tensor]":
    sd = OrderedDict()
    for key, shape, kind in model_spec(layer_sizes):
        t = closed_form_tensor(key, shape, kind, heavy)
        sd[key] = t if kind == "buf_nbt" else t.to(dtype)
    return sd


def closed_form_clips(b: int, t: int, hw: int, dtype=torch.float32, seed_phase: int = 0, heavy: bool = False):
    """Two deterministic clips in [-1, 1] plus labels in the reference ranges
    (datasets.py:873-881,915; preprocess_data.py:520).  ``heavy``: half of the pixels are exactly zero and the rest
    carry a log-uniform magnitude over 2^0 .. 2^-20 (see closed_form_tensor)."""
    n = b * 3 * t * hw * hw
    x1 = hash_uniform(n, 7001 + int(seed_phase))
    x2 = hash_uniform(n, 9001 + int(seed_phase))
    if heavy:
        x1 = x1 * hash_pow2(n, 7101 + int(seed_phase)) * hash_mask(n, 7201 + int(seed_phase))
        x2 = x2 * hash_pow2(n, 9101 + int(seed_phase)) * hash_mask(n, 9201 + int(seed_phase))
    shp = (b, 3, t, hw, hw)
    j = torch.arange(b, dtype=torch.int64)
    labels = {
        "spa": (j * 7 + 3) % 5,
        "tem": (j * 3 + 1) % 5,
        "pb": (j * 5 + 2) % 4,
        "rot1": (j + 1) % 4,
        "rot2": (j * 3 + 2) % 4,
    }
    return x1.reshape(shp).to(dtype), x2.reshape(shp).to(dtype), labels


# --------------------------------------------------------------------------- #
# functional forward
# --------------------------------------------------------------------------- #
def _bn(sd, prefix: str, x: torch.Tensor, training: bool = True) -> torch.Tensor:
    out = F.batch_norm(
        x,
        sd[prefix + ".running_mean"],
        sd[prefix + ".running_var"],
        sd[prefix + ".weight"],
        sd[prefix + ".bias"],
        training,
        BN_MOMENTUM,
        BN_EPS,
    )
    if training:
        sd[prefix + ".num_batches_tracked"] += 1
    return out


def st_conv(sd, prefix: str, x, k, stride, pad, training=True):
    """spatial 1xkxk conv -> BN -> ReLU -> temporal tx1x1 conv."""
    x = F.conv3d(x, sd[prefix + ".spatial_conv.weight"], None, (1, stride[1], stride[2]), (0, pad[1], pad[2]))
    x = F.relu(_bn(sd, prefix + ".bn", x, training))
    x = F.conv3d(x, sd[prefix + ".temporal_conv.weight"], None, (stride[0], 1, 1), (pad[0], 0, 0))
    return x


def res_block(sd, prefix: str, x, downsample: bool, training=True):
    s = (2, 2, 2) if downsample else (1, 1, 1)
    res = st_conv(sd, prefix + ".conv1", x, (3, 3, 3), s, (1, 1, 1), training)
    res = F.relu(_bn(sd, prefix + ".bn1", res, training))
    res = st_conv(sd, prefix + ".conv2", res, (3, 3, 3), (1, 1, 1), (1, 1, 1), training)
    res = _bn(sd, prefix + ".bn2", res, training)
    if downsample:
        x = st_conv(sd, prefix + ".downsampleconv", x, (1, 1, 1), (2, 2, 2), (0, 0, 0), training)
        x = _bn(sd, prefix + ".downsamplebn", x, training)
    return F.relu(x + res)


def mlp(sd, prefix: str, x, training=True):
    x = F.linear(x, sd[prefix + ".0.weight"], sd[prefix + ".0.bias"])
    x = F.relu(_bn(sd, prefix + ".1", x, training))
    return F.linear(x, sd[prefix + ".3.weight"], sd[prefix + ".3.bias"])


def encoder_forward(sd, prefix: str, x, layer_sizes, training=True, proj=True):
    x = st_conv(sd, prefix + ".conv1", x, (3, 7, 7), (1, 2, 2), (1, 3, 3), training)
    x = F.relu(_bn(sd, prefix + ".bn1", x, training))
    for li, n in enumerate(layer_sizes):
        lp = "%s.conv%d" % (prefix, li + 2)
        x = res_block(sd, lp + ".block1", x, li > 0, training)
        for bi in range(n - 1):
            x = res_block(sd, "%s.blocks.%d" % (lp, bi), x, False, training)
    feat = x.mean(dim=(2, 3, 4)).view(-1, 512)
    if not proj:     # proj_flag False: features only (r21d_byol.py:225-229)
        return feat
    return feat, mlp(sd, prefix + ".project.net", feat, training)


def ema_update(sd, layer_sizes, m: float = EMA_MOMENTUM):
    with torch.no_grad():
        for kq, kk in encoder_param_pairs(layer_sizes):
            sd[kk] = sd[kk] * m + sd[kq].detach() * (1.0 - m)


def byol_loss(p1, p2, t1, t2):
    def one(x, y):
        x = F.normalize(x, dim=-1, p=2)
        y = F.normalize(y, dim=-1, p=2)
        return 2 - 2 * (x * y).sum(dim=-1)

    return one(p1, t2) + one(p2, t1)


def model_forward(sd, x1, x2, layer_sizes, training=True):
    """o_type == 'loss_com'.  Mutates sd: BN buffers (online+target) and EMA'd target params."""
    f1, z1 = encoder_forward(sd, "online_net", x1, layer_sizes, training)
    f2, z2 = encoder_forward(sd, "online_net", x2, layer_sizes, training)
    p1 = mlp(sd, "predictor.net", z1, training)
    p2 = mlp(sd, "predictor.net", z2, training)
    with torch.no_grad():
        ema_update(sd, layer_sizes)
        _, t1 = encoder_forward(sd, "target_net", x1, layer_sizes, training)
        _, t2 = encoder_forward(sd, "target_net", x2, layer_sizes, training)
    loss = byol_loss(p1, p2, t1.detach(), t2.detach()).mean()
    fc = torch.cat((f1, f2), dim=1)
    logits = (
        mlp(sd, "overlap_spa", fc, training),
        mlp(sd, "overlap_tem", fc, training),
        mlp(sd, "pb_cls", f1, training),
        mlp(sd, "pb_cls", f2, training),
        mlp(sd, "rotate_cls", f1, training),
        mlp(sd, "rotate_cls", f2, training),
    )
    extras = {"feat_1": f1, "feat_2": f2, "proj_1": z1, "proj_2": z2, "pred_1": p1, "pred_2": p2,
              "tproj_1": t1, "tproj_2": t2}
    return loss, logits, extras


def loss_total(loss_byol, logits, labels, loss_weight):
    ce = [
        F.cross_entropy(logits[0], labels["spa"]),
        F.cross_entropy(logits[1], labels["tem"]),
        F.cross_entropy(logits[2], labels["pb"]),
        F.cross_entropy(logits[3], labels["pb"]),
        F.cross_entropy(logits[4], labels["rot1"]),
        F.cross_entropy(logits[5], labels["rot2"]),
    ]
    w = loss_weight
    total = (w[0] * loss_byol + w[1] * ce[0] + w[2] * ce[1] + w[3] * ce[2] + w[3] * ce[3]
             + w[4] * ce[4] + w[4] * ce[5])
    return total, ce


def train_step(sd, mom, x1, x2, labels, layer_sizes, lr, momentum=0.9, weight_decay=0.0,
               loss_weight=(0.1, 1, 1, 1, 1), clip=True, ntxent_weight=0.0, temperature=0.5):
    """One optimisation step.  ``sd`` tensors are mutated/replaced; ``mom`` is the SGD momentum dict.
    ``ntxent_weight`` != 0: BASELINE configs[1]'s objective -- the criterion main_byol.py:191-197 builds,
    NTXentLoss(zis = online projection of clip 1, zjs = of clip 2), is added to the loss_weight sum and its gradient
    flows into the projector and the online encoder."""
    keys = trainable_keys(layer_sizes)
    for k in keys:
        sd[k] = sd[k].detach().requires_grad_(True)
    loss_byol, logits, extras = model_forward(sd, x1, x2, layer_sizes, True)
    total, ce = loss_total(loss_byol, logits, labels, loss_weight)
    nt = None
    if ntxent_weight != 0.0:
        nt = ntxent(extras["proj_1"], extras["proj_2"], temperature)
        total = total + ntxent_weight * nt
    grads = torch.autograd.grad(total, [sd[k] for k in keys])
    gnorm = torch.sqrt(sum((g.detach() ** 2).sum() for g in grads))
    coef = 1.0
    if clip:
        coef = float(min(1.0, CLIP_VALUE / (float(gnorm) + 1e-6)))
    out_grads = {}
    with torch.no_grad():
        for k, g in zip(keys, grads):
            out_grads[k] = g.detach().clone()
            g = g * coef
            p = sd[k].detach()
            if weight_decay != 0:
                g = g + weight_decay * p
            if k in mom:
                mom[k] = mom[k] * momentum + g
            else:
                mom[k] = g.clone()
            sd[k] = p - lr * mom[k]
    info = {
        "loss_byol": loss_byol.detach(), "loss_total": total.detach(), "ce": [c.detach() for c in ce],
        "logits": [l.detach() for l in logits], "grad_norm": gnorm.detach(), "grads": out_grads,
    }
    info.update({k: v.detach() for k, v in extras.items()})
    if nt is not None:
        info["ntxent"] = nt.detach()
    return info


# --------------------------------------------------------------------------- #
# NT-Xent
# --------------------------------------------------------------------------- #
def ntxent(zis, zjs, temperature: float = 0.5):
    n = zis.shape[0]
    reps = torch.cat([zjs, zis], dim=0)
    nrm = reps.norm(dim=1, keepdim=True)
    sim = (reps @ reps.t()) / torch.clamp(nrm * nrm.t(), min=1e-8)
    idx = torch.arange(2 * n)
    pos = sim[idx, (idx + n) % (2 * n)]
    mask = torch.ones(2 * n, 2 * n, dtype=torch.bool)
    mask[idx, idx] = False
    mask[idx, (idx + n) % (2 * n)] = False
    neg = sim[mask].view(2 * n, -1)
    logits = torch.cat([pos.view(-1, 1), neg], dim=1) / temperature
    return F.cross_entropy(logits, torch.zeros(2 * n, dtype=torch.long), reduction="sum") / (2 * n)


# --------------------------------------------------------------------------- #
# LR schedule as the driver uses it (per-epoch, starts at min_lr)
# --------------------------------------------------------------------------- #
def cosine_warmup_lrs(n_epochs: int, max_lr: float, min_lr: float = 1e-5) -> List[float]:
    """lr in effect during epoch e = 1..n_epochs (main_byol.py:252-269)."""
    warm = 0.5 * n_epochs
    lrs = []
    step = 0  # step_in_cycle after construction (_LRScheduler.__init__ performs one step())
    for _ in range(n_epochs):
        if step < warm:
            lrs.append((max_lr - min_lr) * step / warm + min_lr)
        else:
            lrs.append(min_lr + (max_lr - min_lr) * (1 + math.cos(math.pi * (step - warm) / (n_epochs - warm))) / 2)
        step += 1
        if step >= n_epochs:
            step -= n_epochs
            max_lr = max_lr * 0.5
    return lrs
