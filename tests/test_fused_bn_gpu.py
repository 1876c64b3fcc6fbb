"""temporal_conv(relu(bn(spatial_conv(x)))) (reference r21d_byol.py:94-97) with the BatchNorm + ReLU applied INSIDE the temporal
convolution's gather (ops.bn_relu_conv3d on the f16-pair kernels igemm_k1s / igemm_k2s <.., AFF>; statistics and range from the
spatial convolution's epilogue, cstp_bn_finalize_pre): the normalised tensor is never written.  The fused path must reproduce
the materialising path BIT FOR BIT where the arithmetic is the same sequence of operations (forward output, operand maximum,
data gradient, BatchNorm backward), and to summation-order noise in the weight gradient (bit for bit in deterministic mode);
both must match PyTorch fp64 within the 1e-4 bar of BASELINE.json."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu

FUSED_GEOMS = {
    # name: (x shape, mid channels, out channels, spatial row tile, temporal forward tile (sp, mt, wm, tpb), weight-gradient tile)
    "R18 conv2_x shape class": ((4, 16, 4, 28, 28), 144, 64, 9, (1, 4, 0, 0), (1, 4, 8, 0)),
    "two row blocks, 256-column tile": ((4, 32, 8, 28, 28), 288, 128, 9, (1, 8, 2, 0), (1, 8, 8, 0)),
    "144-row temporal tile": ((4, 16, 4, 28, 28), 48, 144, 9, (1, 9, 0, 0), (1, 9, 8, 0)),
    "64-row spatial tiles": ((2, 32, 2, 56, 56), 64, 48, 4, (1, 3, 0, 0), (1, 4, 4, 0)),
    # the temporal convolution on the LDS-resident-patch kernel igemm_k1t: the transform once per staged element
    "temporal patch kernel, 64 rows": ((4, 16, 8, 28, 28), 144, 64, 9, (2, 4, 0, 0), (1, 4, 8, 0)),
    "temporal patch kernel, 144 rows": ((2, 16, 16, 28, 28), 48, 144, 9, (2, 9, 0, 0), (1, 9, 8, 0)),
}


def _inputs(name, seed=11):
    xs, mid, k, mt_s, tile_t, tile_w = FUSED_GEOMS[name]
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(xs, generator=g) + 0.2
    w_s = torch.randn((mid, xs[1], 1, 3, 3), generator=g) * 0.1
    w_t = torch.randn((k, mid, 3, 1, 1), generator=g) * 0.1
    gamma = torch.rand(mid, generator=g) * 1.5 - 0.25          # some negative scales: the range maps through the MINIMUM there
    beta = torch.randn(mid, generator=g) * 0.5
    return x, w_s, w_t, gamma, beta


def _pin(name):
    from cstp_amd import ops
    xs, mid, k, mt_s, tile_t, tile_w = FUSED_GEOMS[name]
    ys = (xs[0], mid) + tuple(xs[2:])
    ops.set_conv_tile(xs, (mid, xs[1], 1, 3, 3), (1, 1, 1), (0, 1, 1), 0, (2, mt_s, 0, 0))
    ops.set_conv_tile(ys, (k, mid, 3, 1, 1), (1, 1, 1), (1, 0, 0), 0, tile_t)
    ops.set_conv_tile(ys, (k, mid, 3, 1, 1), (1, 1, 1), (1, 0, 0), 2, tile_w)
    return ys


def _run(name, groups, fused, dy_seed=3):
    """One forward + backward of the chain on the GPU; returns everything the two paths must agree on."""
    from cstp_amd import ops
    x, w_s, w_t, gamma, beta = _inputs(name)
    xg = x.cuda().requires_grad_(True)
    wsg, wtg = w_s.cuda().requires_grad_(True), w_t.cuda().requires_grad_(True)
    gg, bg = gamma.cuda().requires_grad_(True), beta.cuda().requires_grad_(True)
    rm, rv = torch.zeros(gamma.numel(), device="cuda"), torch.ones(gamma.numel(), device="cuda")
    y = ops.conv3d(xg, wsg, None, 1, (0, 1, 1), bn_groups=groups, bn_pivot=rm)
    assert ops._bnstats_of(y, groups) is not None
    zcell = ops._bnstats_of(y, groups)[2]
    if fused:
        out = ops.bn_relu_conv3d(y, gg, bg, rm, rv, wtg, 1, (1, 0, 0), groups, True, 1e-5, 0.1)
        zmax = int(zcell.item())
    else:
        z = ops.batch_norm_act(y, gg, bg, rm, rv, None, True, 1e-5, 0.1, groups)
        zmax = int(ops._absmax_of(z).item())
        out = ops.conv3d(z, wtg, None, 1, (1, 0, 0))
    dy = (torch.rand(out.shape, generator=torch.Generator().manual_seed(dy_seed)) * 2 - 1).cuda()
    out.backward(dy)
    ops._join_side_streams()
    torch.cuda.synchronize()
    return dict(out=out.detach(), zmax=zmax, dx=xg.grad, dws=wsg.grad, dwt=wtg.grad, dgamma=gg.grad, dbeta=bg.grad, rm=rm, rv=rv)


@pytest.mark.parametrize("groups", [1, 2])
@pytest.mark.parametrize("name", list(FUSED_GEOMS))
def test_fused_path_equals_the_materialising_path(name, groups):
    from cstp_amd import ops
    ops.set_split_terms(2)
    ops.set_deterministic(True)
    try:
        ys = _pin(name)
        xs, mid, k = FUSED_GEOMS[name][:3]
        assert ops.in_affine_fused(ys, (k, mid, 3, 1, 1), 1, (1, 0, 0), groups)
        a, b = _run(name, groups, True), _run(name, groups, False)
        assert a["zmax"] == b["zmax"] and a["zmax"] != 0, "operand maximum from (min, max) != measured maximum of the written tensor"
        for key in ("out", "dwt", "dx", "dws", "dgamma", "dbeta", "rm", "rv"):
            assert torch.equal(a[key], b[key]), (key, float((a[key] - b[key]).abs().max()))
    finally:
        ops.set_deterministic(False)
        ops.set_split_terms(0)


@pytest.mark.parametrize("name", ["R18 conv2_x shape class", "144-row temporal tile"])
def test_fused_path_against_fp64(name):
    from cstp_amd import ops
    ops.set_split_terms(2)
    try:
        _pin(name)
        groups = 2
        got = _run(name, groups, True)
        x, w_s, w_t, gamma, beta = [t.double().requires_grad_(True) for t in _inputs(name)]
        y = F.conv3d(x, w_s, None, 1, (0, 1, 1))
        rm, rv = torch.zeros(gamma.numel(), dtype=torch.float64), torch.ones(gamma.numel(), dtype=torch.float64)
        z = torch.cat([F.relu(F.batch_norm(p, rm, rv, gamma, beta, True, 0.1, 1e-5)) for p in y.chunk(groups, 0)], 0)
        out = F.conv3d(z, w_t, None, 1, (1, 0, 0))
        dy = (torch.rand(out.shape, generator=torch.Generator().manual_seed(3)) * 2 - 1).double()
        out.backward(dy)
        assert rel_err(got["out"], out.detach()) < 1e-4
        for key, ref in (("dx", x.grad), ("dws", w_s.grad), ("dwt", w_t.grad), ("dgamma", gamma.grad), ("dbeta", beta.grad),
                         ("rm", rm), ("rv", rv)):
            assert rel_err(got[key], ref) < 1e-4, key
    finally:
        ops.set_split_terms(0)


def test_geometries_the_fused_gather_declines():
    """Not whole 16-channel groups, three BatchNorm groups, a BatchNorm group that ends inside a K-tile / column tile: the query
    says no (callers materialise), and a call that insists still computes the right thing on the native kernels."""
    from cstp_amd import _lib, ops
    ops.set_split_terms(2)
    try:
        assert not ops.in_affine_fused((4, 40, 4, 28, 28), (64, 40, 3, 1, 1), 1, (1, 0, 0), 2)       # 40 channels
        assert not ops.in_affine_fused((6, 48, 4, 28, 28), (64, 48, 3, 1, 1), 1, (1, 0, 0), 3)       # three groups
        assert not ops.in_affine_fused((2, 48, 4, 7, 7), (64, 48, 3, 1, 1), 1, (1, 0, 0), 2)         # 196 positions per group
        # the insisting call: statistics pass + native kernels with the transform in their gather
        g = torch.Generator().manual_seed(2)
        x = torch.randn((2, 40, 4, 7, 7), generator=g)
        w = torch.randn((24, 40, 3, 1, 1), generator=g) * 0.1
        gamma, beta = torch.rand(40, generator=g) + 0.5, torch.randn(40, generator=g)
        xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
        out = ops.bn_relu_conv3d(xg, gamma.cuda(), beta.cuda(), None, None, wg, 1, (1, 0, 0), 2, True, 1e-5, 0.1)
        xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
        z = torch.cat([F.relu(F.batch_norm(p, None, None, gamma.double(), beta.double(), True, 0.1, 1e-5)) for p in xd.chunk(2, 0)], 0)
        ref = F.conv3d(z, wd, None, 1, (1, 0, 0))
        dy = torch.rand(ref.shape, generator=g).double()
        ref.backward(dy)
        out.backward(dy.float().cuda())
        ops._join_side_streams()
        assert rel_err(out.detach(), ref.detach()) < 1e-4 and rel_err(wg.grad, wd.grad) < 1e-4 and rel_err(xg.grad, xd.grad) < 1e-4
    finally:
        ops.set_split_terms(0)


def test_spatiotemporal_module_takes_the_fused_path_and_matches_the_unfused_module():
    """SpatioTemporalConv (train mode, 2 view groups): default = fused; CSTP_FUSE_BN_T=0 semantics through the module flag."""
    from cstp_amd import ops, r21d_byol as rb
    ops.set_deterministic(True)
    try:
        torch.manual_seed(4)
        m = rb.SpatioTemporalConv(64, 64, 3, padding=1).cuda().train()          # 144 mid channels
        x = torch.randn(4, 64, 4, 28, 28, device="cuda")
        assert ops.in_affine_fused((4, 144, 4, 28, 28), tuple(m.temporal_conv.weight.shape), 1, (1, 0, 0), 2)
        outs = []
        for fused in (True, False):
            rb.FUSE_BN_TEMPORAL = fused
            m.bn.running_mean.zero_(); m.bn.running_var.fill_(1.0)
            m.zero_grad(set_to_none=True)
            xg = x.clone().requires_grad_(True)
            y = m(xg, groups=2)
            y.square().sum().backward()
            ops._join_side_streams()
            outs.append((y.detach().clone(), xg.grad.clone(), [p.grad.clone() for p in m.parameters()]))
        rb.FUSE_BN_TEMPORAL = True
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        for a, b in zip(outs[0][2], outs[1][2]):
            assert torch.equal(a, b)
    finally:
        rb.FUSE_BN_TEMPORAL = True
        ops.set_deterministic(False)
