"""Per-op parity of the HIP kernels (through the C ABI) against plain PyTorch CPU ops in fp64.
Tolerance: fp32 1e-4 relative (max-abs-diff / max-abs-ref), the bar BASELINE.json states."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _rand(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1


CONVS = [
    # (n, c, d, h, w, k, kernel, stride, pad)            what
    (2, 3, 4, 28, 28, 83, (1, 7, 7), (1, 2, 2), (0, 3, 3)),      # stem S0 (straddle path)
    (2, 64, 4, 14, 14, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1)),    # S1
    (2, 64, 4, 14, 14, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1)),    # S2 strided
    (2, 83, 4, 14, 14, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)),     # T0
    (2, 230, 8, 7, 7, 128, (3, 1, 1), (2, 1, 1), (1, 0, 0)),     # T2 strided
    (2, 64, 4, 14, 14, 42, (1, 1, 1), (1, 2, 2), (0, 0, 0)),     # shortcut spatial half
    (2, 42, 4, 7, 7, 128, (1, 1, 1), (2, 1, 1), (0, 0, 0)),      # shortcut temporal half
    (3, 5, 3, 9, 11, 7, (3, 3, 3), (2, 2, 2), (1, 1, 1)),        # generic ragged 3-D conv
    (2, 16, 3, 7, 7, 20, (1, 3, 3), (1, 2, 2), (0, 1, 1)),       # odd H/W with stride 2
    (1, 512, 2, 7, 7, 1152, (1, 3, 3), (1, 1, 1), (0, 1, 1)),    # S7
    (1, 1152, 2, 7, 7, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0)),    # T7
    (2, 256, 2, 7, 7, 921, (1, 3, 3), (1, 2, 2), (0, 1, 1)),     # S6
    (1, 6, 1, 1, 1, 9, (1, 1, 1), (1, 1, 1), (0, 0, 0)),         # degenerate single position
    (2, 144, 4, 14, 14, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)),    # T1: data gradient on the 144-row 16x16x4 tile
    (3, 40, 2, 9, 9, 136, (1, 3, 3), (1, 1, 1), (0, 1, 1)),      # 129..144 rows, ragged positions (16x16x4 tile tails)
]


@pytest.mark.parametrize("cfg", CONVS, ids=lambda c: "n%dc%dd%dh%dw%dk%d_%s_s%s" % (c[0], c[1], c[2], c[3], c[4], c[5], "x".join(map(str, c[6])), "x".join(map(str, c[7]))))
def test_conv3d_fwd_bwd(cfg):
    from cstp_amd import ops
    n, c, d, h, w, k, ks, st, pd = cfg
    x = _rand((n, c, d, h, w), 1).requires_grad_(True)
    wt = (_rand((k, c) + ks, 2) * 0.2).requires_grad_(True)
    y = F.conv3d(x, wt, None, st, pd)
    dy = _rand(tuple(y.shape), 3)
    y.backward(dy)

    xg = x.detach().float().cuda().requires_grad_(True)
    wg = wt.detach().float().cuda().requires_grad_(True)
    yg = ops.conv3d(xg, wg, None, st, pd)
    assert tuple(yg.shape) == tuple(y.shape)
    yg.backward(dy.float().cuda())
    torch.cuda.synchronize()
    assert rel_err(yg, y) < TOL
    assert rel_err(xg.grad, x.grad) < TOL
    assert rel_err(wg.grad, wt.grad) < TOL


# (<= 32 rows, reduction length a multiple of 4: the weight-streaming kernels of csrc/linear.h; else the 1x1x1 convolution)
@pytest.mark.parametrize("b,fin,fout", [(16, 512, 4096), (16, 4096, 512), (4, 1024, 5), (2, 512, 512), (130, 70, 33), (32, 4096, 512),
                                        (32, 512, 4096), (32, 1024, 1000), (17, 100, 36), (32, 70, 33), (33, 512, 64)])
def test_linear_fwd_bwd(b, fin, fout):
    from cstp_amd import ops
    x = _rand((b, fin), 4).requires_grad_(True)
    wt = (_rand((fout, fin), 5) * 0.1).requires_grad_(True)
    bs = _rand((fout,), 6).requires_grad_(True)
    y = F.linear(x, wt, bs)
    dy = _rand(tuple(y.shape), 7)
    y.backward(dy)
    xg = x.detach().float().cuda().requires_grad_(True)
    wg = wt.detach().float().cuda().requires_grad_(True)
    bg = bs.detach().float().cuda().requires_grad_(True)
    yg = ops.linear(xg, wg, bg)
    yg.backward(dy.float().cuda())
    assert rel_err(yg, y) < TOL
    assert rel_err(xg.grad, x.grad) < TOL
    assert rel_err(wg.grad, wt.grad) < TOL
    assert rel_err(bg.grad, bs.grad) < TOL


BNS = [
    # shape, residual, relu
    ((4, 83, 4, 14, 14), False, True),
    ((4, 64, 4, 14, 14), True, True),
    ((3, 64, 2, 7, 7), False, False),    # s = 98 (not a multiple of 4: scalar path)
    ((3, 17, 2, 7, 7), True, True),
    ((16, 4096), False, True),           # BatchNorm1d
    ((4, 512), False, True),
    ((2, 1024), False, False),
    ((2, 6, 3, 5, 5), True, False),
    # more than 64 partial sums per channel (nsplit = min(2048 / c, n) = 80) AND a frame size that is not a multiple of 4: the
    # statistics folded inside the apply passes (BnFin) on the scalar path (round-3 ADVICE)
    ((80, 8, 5, 3, 3), False, True),
    ((72, 6, 3, 5, 5), True, True),
    # the single-launch kernels for small tensors (bn_small_*_kernel: <= 4096 values per channel and group): the 7 x 7 stage's
    # shape class, the boundary (8 * 512 = 4096 values) and the first size past it (three-launch sequence)
    ((16, 40, 2, 7, 7), True, True),
    ((8, 5, 8, 8, 8), False, True),
    ((8, 5, 8, 8, 9), True, True),
    # ... on blocks of 1024 threads (4096 < values <= 16 384: the 14 x 14 stage), its boundary, and the first size past it
    ((16, 12, 4, 14, 14), True, True),
    ((16, 4, 4, 16, 16), False, True),
    ((17, 4, 4, 16, 16), True, False),
]


@pytest.mark.parametrize("shape,use_res,relu", BNS, ids=lambda v: str(v))
def test_bn_act_fwd_bwd(shape, use_res, relu):
    from cstp_amd import ops
    c = shape[1]
    x = (_rand(shape, 8) * 2 + 0.5).requires_grad_(True)
    gamma = _rand((c,), 9).requires_grad_(True)
    beta = (_rand((c,), 10) * 0.1).requires_grad_(True)
    res = _rand(shape, 11).requires_grad_(True) if use_res else None
    rm, rv = _rand((c,), 12) * 0.1, _rand((c,), 13).abs() + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
    if use_res:
        y = y + res
    if relu:
        y = F.relu(y)
    dy = _rand(shape, 14)
    y.backward(dy)

    xg = x.detach().float().cuda().requires_grad_(True)
    gg = gamma.detach().float().cuda().requires_grad_(True)
    bg = beta.detach().float().cuda().requires_grad_(True)
    rg = res.detach().float().cuda().requires_grad_(True) if use_res else None
    rmg, rvg = rm.float().cuda(), rv.float().cuda()
    yg = ops.batch_norm_act(xg, gg, bg, rmg, rvg, rg, relu)
    yg.backward(dy.float().cuda())
    assert rel_err(yg, y) < TOL
    assert rel_err(rmg, rm_ref) < TOL and rel_err(rvg, rv_ref) < TOL
    assert rel_err(xg.grad, x.grad) < TOL
    assert rel_err(gg.grad, gamma.grad) < TOL
    assert rel_err(bg.grad, beta.grad) < TOL
    if use_res:
        assert rel_err(rg.grad, res.grad) < TOL


@pytest.mark.parametrize("shape,relu", [((6, 24, 2, 6, 6), True), ((8, 40), True), ((4, 9, 1, 7, 7), False), ((32, 20, 2, 7, 7), True),
                                        ((16, 6, 8, 8, 8), True), ((32, 6, 4, 14, 14), True)])
def test_bn_groups_equal_successive_calls(shape, relu):
    """groups=2 over a 2B batch == two successive F.batch_norm calls (per-view stats, sequential running stats)."""
    from cstp_amd import ops
    c, half = shape[1], shape[0] // 2
    x = (_rand(shape, 31) * 1.5 + 0.3).requires_grad_(True)
    gamma = _rand((c,), 32).requires_grad_(True)
    beta = (_rand((c,), 33) * 0.1).requires_grad_(True)
    rm, rv = torch.zeros(c, dtype=torch.float64), torch.ones(c, dtype=torch.float64)
    ys = [F.batch_norm(x[i * half:(i + 1) * half], rm, rv, gamma, beta, True, 0.1, 1e-5) for i in range(2)]
    y = torch.cat(ys, 0)
    if relu:
        y = F.relu(y)
    dy = _rand(shape, 34)
    y.backward(dy)
    xg = x.detach().float().cuda().requires_grad_(True)
    gg = gamma.detach().float().cuda().requires_grad_(True)
    bg = beta.detach().float().cuda().requires_grad_(True)
    rmg, rvg = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    yg = ops.batch_norm_act(xg, gg, bg, rmg, rvg, None, relu, groups=2)
    yg.backward(dy.float().cuda())
    assert rel_err(yg, y) < TOL and rel_err(rmg, rm) < TOL and rel_err(rvg, rv) < TOL
    assert rel_err(xg.grad, x.grad) < TOL and rel_err(gg.grad, gamma.grad) < TOL and rel_err(bg.grad, beta.grad) < TOL


FUSED = [
    # x shape, out channels, kernel, stride, pad, groups   (BN -> ReLU -> conv fused)
    ((4, 144, 4, 14, 14), 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), 2),     # SpatioTemporalConv.bn -> temporal conv
    ((4, 64, 4, 14, 14), 144, (1, 3, 3), (1, 1, 1), (0, 1, 1), 2),      # block bn1 -> conv2.spatial_conv
    ((2, 230, 8, 7, 7), 128, (3, 1, 1), (2, 1, 1), (1, 0, 0), 1),       # strided temporal
    ((6, 42, 2, 7, 7), 128, (1, 1, 1), (2, 1, 1), (0, 0, 0), 2),        # shortcut temporal half, s = 98 (scalar paths)
    ((4, 83, 3, 9, 9), 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), 2),         # ragged channels / extents
]


@pytest.mark.parametrize("xs,k,ks,st,pd,groups", FUSED, ids=lambda v: str(v))
def test_bn_relu_conv3d_fused(xs, k, ks, st, pd, groups):
    """Fused BN(train)->ReLU->conv == F.batch_norm per group -> relu -> F.conv3d, forward and all gradients."""
    from cstp_amd import ops
    c, half = xs[1], xs[0] // groups
    x = (_rand(xs, 41) * 1.3 + 0.2).requires_grad_(True)
    gamma = _rand((c,), 42).requires_grad_(True)
    beta = (_rand((c,), 43) * 0.2).requires_grad_(True)
    w = (_rand((k, c) + ks, 44) * 0.1).requires_grad_(True)
    rm, rv = torch.zeros(c, dtype=torch.float64), torch.ones(c, dtype=torch.float64)
    zs = [F.relu(F.batch_norm(x[i * half:(i + 1) * half], rm, rv, gamma, beta, True, 0.1, 1e-5)) for i in range(groups)]
    y = F.conv3d(torch.cat(zs, 0), w, None, st, pd)
    dy = _rand(tuple(y.shape), 45)
    y.backward(dy)
    xg = x.detach().float().cuda().requires_grad_(True)
    gg = gamma.detach().float().cuda().requires_grad_(True)
    bg = beta.detach().float().cuda().requires_grad_(True)
    wg = w.detach().float().cuda().requires_grad_(True)
    rmg, rvg = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    yg = ops.bn_relu_conv3d(xg, gg, bg, rmg, rvg, wg, st, pd, groups=groups, relu=True)
    yg.backward(dy.float().cuda())
    assert rel_err(yg, y) < TOL and rel_err(rmg, rm) < TOL and rel_err(rvg, rv) < TOL
    assert rel_err(wg.grad, w.grad) < TOL
    assert rel_err(xg.grad, x.grad) < TOL
    assert rel_err(gg.grad, gamma.grad) < TOL and rel_err(bg.grad, beta.grad) < TOL


def test_bn_rejects_single_value():
    from cstp_amd import ops, _lib
    x = torch.ones(1, 8, device="cuda")
    with pytest.raises(_lib.CstpError):
        ops.batch_norm_act(x, torch.ones(8, device="cuda"), torch.zeros(8, device="cuda"))


@pytest.mark.parametrize("shape", [(4, 512, 2, 7, 7), (2, 512, 1, 4, 4), (3, 7, 1, 1, 1)])
def test_avgpool(shape):
    from cstp_amd import ops
    x = _rand(shape, 15).requires_grad_(True)
    y = x.mean(dim=(2, 3, 4))
    dy = _rand(tuple(y.shape), 16)
    y.backward(dy)
    xg = x.detach().float().cuda().requires_grad_(True)
    yg = ops.global_avg_pool(xg)
    yg.backward(dy.float().cuda())
    assert rel_err(yg, y) < TOL and rel_err(xg.grad, x.grad) < TOL


def test_byol_loss():
    from cstp_amd import ops
    x = _rand((16, 512), 17).requires_grad_(True)
    t = _rand((16, 512), 18)
    l = 2 - 2 * (F.normalize(x, dim=-1) * F.normalize(t, dim=-1)).sum(-1)
    dl = _rand((16,), 19)
    l.backward(dl)
    xg = x.detach().float().cuda().requires_grad_(True)
    lg = ops.byol_regression_loss(xg, t.float().cuda())
    lg.backward(dl.float().cuda())
    assert rel_err(lg, l) < TOL and rel_err(xg.grad, x.grad) < TOL


@pytest.mark.parametrize("b,k", [(16, 5), (4, 5), (300, 7)])
def test_cross_entropy(b, k):
    from cstp_amd import ops
    z = (_rand((b, k), 20) * 3).requires_grad_(True)
    lab = (torch.arange(b) * 7 + 3) % k
    l = F.cross_entropy(z, lab)
    (l * 1.7).backward()
    zg = z.detach().float().cuda().requires_grad_(True)
    lg = ops.cross_entropy(zg, lab.cuda())
    (lg * 1.7).backward()
    assert rel_err(lg, l) < TOL and rel_err(zg.grad, z.grad) < TOL


@pytest.mark.parametrize("n,f,tau", [(4, 64, 0.5), (16, 512, 0.5), (8, 64, 0.1), (128, 512, 0.5)])
def test_ntxent(n, f, tau):
    from cstp_amd import ops
    from oracle import r21d_byol_oracle as orc
    zi = _rand((n, f), 21).requires_grad_(True)
    zj = _rand((n, f), 22).requires_grad_(True)
    l = orc.ntxent(zi, zj, tau)
    l.backward()
    reps = torch.cat([zj.detach(), zi.detach()], 0).float().cuda().requires_grad_(True)
    lg = ops.ntxent(reps, tau)
    lg.backward()
    assert rel_err(lg, l) < TOL
    assert rel_err(reps.grad, torch.cat([zj.grad, zi.grad], 0)) < TOL


def test_flat_utils():
    from cstp_amd import ops
    n = 1000003
    t = _rand((n,), 23).float()
    o = _rand((n,), 24).float()
    tg, og = t.cuda(), o.cuda()
    ops.ema_update_(tg, og, 0.996)
    assert rel_err(tg, t * 0.996 + o * (1.0 - 0.996)) < 1e-6
    g = (_rand((n,), 25) * 0.05).float()
    gg = g.cuda()
    ss = torch.zeros(1, device="cuda")
    coef = torch.zeros(1, device="cuda")
    nrm = torch.zeros(1, device="cuda")
    ops.grad_sumsq(gg, ss)
    ops.clip_coef(ss, 18.0, coef, nrm)
    ref_norm = g.double().norm()
    assert abs(float(nrm) - float(ref_norm)) / float(ref_norm) < 1e-6
    ref_coef = min(1.0, 18.0 / (float(ref_norm) + 1e-6))
    assert abs(float(coef) - ref_coef) < 1e-6
    # SGD, two steps, against torch.optim.SGD on CPU
    p = torch.nn.Parameter(t.clone())
    opt = torch.optim.SGD([p], lr=0.05, momentum=0.9, weight_decay=5e-4)
    pg, buf = t.cuda(), torch.zeros(n, device="cuda")
    lr = torch.full((1,), 0.05, device="cuda")
    for step in range(2):
        p.grad = g.clone() * ref_coef
        opt.step()
        g2 = g.cuda()
        ops.sgd_step_(pg, g2, buf, lr, 0.9, 5e-4, coef, step == 0, True)
        assert rel_err(g2, g * ref_coef) < 1e-6
    assert rel_err(pg, p.detach()) < 1e-6


@pytest.mark.parametrize("shape,groups,use_res", [((32, 24, 2, 7, 7), 2, True), ((16, 10, 8, 8, 8), 2, False), ((6, 7, 1, 5, 5), 1, False),
                                                  ((32, 9, 4, 14, 14), 2, True), ((2, 3, 1, 3, 3), 1, False)])
def test_bn_small_tensor_kernels_leave_the_absmax_cells(shape, groups, use_res):
    """bn_small_fwd / bwd_kernel (one launch per pass + a one-block fold that STORES the cell): max |y| and max |dx| as bits."""
    from cstp_amd import ops
    ops.set_split_terms(2)
    try:
        c = shape[1]
        x = (_rand(shape, 51) * 3).float().cuda().requires_grad_(True)
        res = _rand(shape, 52).float().cuda().requires_grad_(True) if use_res else None
        g = (_rand((c,), 53).abs() + 0.5).float().cuda().requires_grad_(True)
        b = _rand((c,), 54).float().cuda().requires_grad_(True)
        for _ in range(2):                                # the second call finds the cell of the first one used, not zeroed
            y = ops.batch_norm_act(x, g, b, None, None, res, True, groups=groups)
            cell, ver = y._cstp_absmax
            assert ver == y._version
            assert int(cell.item()) == int(y.detach().abs().max().view(torch.int32).item())
        dy = _rand(shape, 55).float().cuda()
        y.backward(dy)
        dcell = ops._absmax_of(x.grad)
        if dcell is not None:
            assert int(dcell.item()) == int(x.grad.abs().max().view(torch.int32).item())
    finally:
        ops.set_split_terms(0)
