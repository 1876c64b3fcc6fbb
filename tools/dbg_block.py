import sys, torch, torch.nn as nn
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import rel_err
from cstp_amd import ops, r21d_byol as rb
groups = 1
ops.set_split_terms(2)
torch.manual_seed(7)
blk = rb.SpatioTemporalResBlock(64, 64, 3).cuda().train()
x = torch.randn(2 * groups, 64, 8, 28, 28, device="cuda")
ys, ts = (2 * groups, 144, 8, 28, 28), (64, 144, 3, 1, 1)
import os
tf = eval(os.environ.get("TF", "(2,4,0,0)")); td = eval(os.environ.get("TD", "(2,9,0,0)"))
ops.set_conv_tile(ys, ts, (1, 1, 1), (1, 0, 0), 0, tf)
ops.set_conv_tile(ys, ts, (1, 1, 1), (1, 0, 0), 1, td)
ops.set_conv_tile(x.shape, (144, 64, 1, 3, 3), (1, 1, 1), (0, 1, 1), 0, (2, 9, 0, 0))
state = {k: v.clone() for k, v in blk.state_dict().items()}
res = []
for fuse_t, fuse_stats in ((True, True), (False, False), (True, False), (False, True)):
    rb.FUSE_BN_TEMPORAL, ops.FUSE_BN_STATS = fuse_t, fuse_stats
    blk.load_state_dict(state); blk.zero_grad(set_to_none=True)
    xg = x.clone().requires_grad_(True)
    y = blk(xg, groups); y.square().mean().backward(); ops._join_side_streams()
    res.append((y.detach().clone(), xg.grad.clone(), {n: p.grad.clone() for n, p in blk.named_parameters()}))
def ref_stc(pre):
    m = nn.Sequential(nn.Conv3d(64, 144, (1, 3, 3), 1, (0, 1, 1), bias=False), nn.BatchNorm3d(144), nn.ReLU(), nn.Conv3d(144, 64, (3, 1, 1), 1, (1, 0, 0), bias=False)).double()
    m[0].weight.data = state[pre + ".spatial_conv.weight"].double().cpu()
    m[1].weight.data = state[pre + ".bn.weight"].double().cpu(); m[1].bias.data = state[pre + ".bn.bias"].double().cpu()
    m[3].weight.data = state[pre + ".temporal_conv.weight"].double().cpu()
    return m
c1, c2 = ref_stc("conv1"), ref_stc("conv2")
bn1, bn2 = nn.BatchNorm3d(64).double(), nn.BatchNorm3d(64).double()
xd = x.double().cpu().requires_grad_(True)
h = c1(xd); h = torch.relu(bn1(h)); h = c2(h); out = torch.relu(xd + bn2(h)); out.square().mean().backward()
refg = {"conv1.spatial_conv.weight": c1[0].weight.grad, "conv1.temporal_conv.weight": c1[3].weight.grad, "conv2.spatial_conv.weight": c2[0].weight.grad,
        "conv2.temporal_conv.weight": c2[3].weight.grad, "conv1.bn.weight": c1[1].weight.grad, "bn1.weight": bn1.weight.grad, "bn2.weight": bn2.weight.grad, "bn2.bias": bn2.bias.grad}
for name, r in zip(("fused+stats", "plain", "fused only", "stats only"), res):
    print(name, "y %.2e dx %.2e" % (rel_err(r[0], out.detach()), rel_err(r[1], xd.grad)), {k: "%.1e" % rel_err(r[2][k], v) for k, v in refg.items()})
