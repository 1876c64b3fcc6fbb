import sys, torch, torch.nn as nn
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import rel_err
from cstp_amd import ops, r21d_byol as rb
groups = 1
ops.set_split_terms(2)
torch.manual_seed(7)
blk = rb.SpatioTemporalResBlock(64, 64, 3).cuda().train()
x = torch.randn(2, 64, 8, 28, 28, device="cuda")
ys, ts = (2, 144, 8, 28, 28), (64, 144, 3, 1, 1)
ops.set_conv_tile(ys, ts, (1, 1, 1), (1, 0, 0), 0, (2, 4, 0, 0))
ops.set_conv_tile(ys, ts, (1, 1, 1), (1, 0, 0), 1, (2, 9, 0, 0))
ops.set_conv_tile(x.shape, (144, 64, 1, 3, 3), (1, 1, 1), (0, 1, 1), 0, (2, 9, 0, 0))
state = {k: v.clone() for k, v in blk.state_dict().items()}
rb.FUSE_BN_TEMPORAL = False
def fwd(self, x, groups, s1, s2, join_on):
    join = ops.GradJoin(2) if join_on else None
    h = self.conv1(x, groups, grad_join=join, out_bn=self.bn1 if s1 else None)
    res = self.conv2(h, groups, pre_bn=self.bn1, out_bn=self.bn2 if s2 else None)
    return self.bn2(res, residual=x, relu=True, groups=groups, grad_join=join)
ref = None
for s1, s2, sS, join_on in ((0, 0, 0, 1), (1, 0, 0, 1), (0, 0, 1, 1)):
    ops.FUSE_BN_STATS = True
    # sS: spatial convolutions leave stats (bn_groups passed in SpatioTemporalConv.forward) -- toggled through the conv wrapper
    orig = rb.Conv3d.forward
    if not sS:
        def nf(self, x, bn_groups=0, bn_pivot=None, grad_join=None, _o=orig):
            if self.weight.shape[2] == 1: bn_groups = 0
            return _o(self, x, bn_groups, bn_pivot, grad_join)
        rb.Conv3d.forward = nf
    blk.load_state_dict(state); blk.zero_grad(set_to_none=True)
    xg = x.clone().requires_grad_(True)
    y = fwd(blk, xg, groups, s1, s2, join_on); y.square().mean().backward(); ops._join_side_streams()
    rb.Conv3d.forward = orig
    cur = (xg.grad.clone(), {n: p.grad.clone() for n, p in blk.named_parameters()})
    if ref is None: ref = cur
    d = (cur[0] - ref[0]).abs(); mx = ref[0].abs().max()
    print("   elements of dx off by > 1e-5 max:", int((d > 1e-5 * mx).sum()), "of", d.numel(), " > 1e-4:", int((d > 1e-4 * mx).sum()), "worst at", (d == d.max()).nonzero()[0].tolist() if d.max() > 0 else None)
    print((s1, s2, sS, join_on), "dx %.2e" % rel_err(cur[0], ref[0]), {k.replace("_conv.weight", "").replace(".weight", ".w"): "%.0e" % rel_err(v, ref[1][k]) for k, v in cur[1].items() if "bias" not in k})
