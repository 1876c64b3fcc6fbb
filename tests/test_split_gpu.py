"""Parity of EVERY GEMM kernel variant against PyTorch CPU fp64, independent of which one the autotuner would pick on the
machine at hand: each test pins one variant for one geometry through cstp_conv3d_set_tile (C ABI) and runs forward, data
gradient and weight gradient through the normal autograd path.  Variants: the native f32 MFMA tiles and the split kernels
(igemm_k1s row tiles 2..9 x 16, igemm_k2s 64/128/144-row tiles) in BOTH arithmetics -- f16 pair / three products (default) and
bf16 triple / six products (cstp_gemm_set_split_terms).  Tolerance: the 1e-4 bar of BASELINE.json (measured errors are ~1e-6,
tools/split_accuracy.py)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4

GEOMS = {
    # name: (x shape, k, kernel, stride, pad)
    "S1": ((2, 64, 4, 14, 14), 144, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    "T1": ((2, 144, 4, 14, 14), 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    "S2s": ((2, 64, 4, 14, 14), 230, (1, 3, 3), (1, 2, 2), (0, 1, 1)),          # strided spatial, 230 rows
    "T2s": ((2, 230, 8, 7, 7), 128, (3, 1, 1), (2, 1, 1), (1, 0, 0)),           # strided temporal (parity classes in dgrad)
    "S7": ((1, 512, 2, 7, 7), 1152, (1, 3, 3), (1, 1, 1), (0, 1, 1)),           # long reduction, few positions
    "odd": ((3, 40, 3, 9, 11), 136, (3, 3, 3), (1, 2, 1), (1, 1, 1)),           # ragged everything, 27 taps, channel padding
    "lin": ((6, 96, 1, 1, 1), 40, (1, 1, 1), (1, 1, 1), (0, 0, 0)),             # nn.Linear as a 1x1x1 convolution
    "short": ((2, 64, 4, 14, 14), 42, (1, 1, 1), (1, 2, 2), (0, 0, 0)),         # shortcut's spatial half: strided pointwise
    "lin64": ((6, 192, 1, 1, 1), 128, (1, 1, 1), (1, 1, 1), (0, 0, 0)),         # nn.Linear on the weight-streaming kernels (csrc/linear.h)
}


def _rand(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1


@pytest.fixture(params=[2, 3], ids=["f16x2", "bf16x3"])
def terms(request):
    from cstp_amd import ops
    ops.set_split_terms(request.param)
    yield request.param
    ops.set_split_terms(0)


def _run(name, tiles, scale_x=1.0, scale_w=1.0, scale_dy=1.0, tol=TOL):
    """tiles: {mode: tile4}.  Returns nothing; asserts parity of y, dx, dw."""
    from cstp_amd import ops
    xs, k, ks, st, pd = GEOMS[name]
    x = (_rand(xs, 1) * scale_x).requires_grad_(True)
    w = (_rand((k, xs[1]) + ks, 2) * 0.2 * scale_w).requires_grad_(True)
    y = F.conv3d(x, w, None, st, pd)
    dy = _rand(y.shape, 3) * scale_dy
    y.backward(dy)
    for mode, tile in tiles.items():
        ops.set_conv_tile(xs, w.shape, st, pd, mode, tile)
    xg = x.detach().float().cuda().requires_grad_(True)
    wg = w.detach().float().cuda().requires_grad_(True)
    yg = ops.conv3d(xg, wg, None, st, pd)
    yg.backward(dy.float().cuda())
    assert rel_err(yg, y) < tol, ("forward", name, tiles)
    assert rel_err(xg.grad, x.grad) < tol, ("backward_data", name, tiles)
    assert rel_err(wg.grad, w.grad) < tol, ("backward_weight", name, tiles)


@pytest.mark.parametrize("mt", [2, 3, 4, 5, 6, 8, 9])
@pytest.mark.parametrize("name", list(GEOMS))
def test_split_forward_and_data_gradient(name, mt, terms):
    _run(name, {0: (1, mt, 0, 0), 1: (1, mt, 0, 0), 2: (0, 2, 8, 0)})


@pytest.mark.parametrize("mt", [8, 9])
@pytest.mark.parametrize("name", list(GEOMS))
def test_split_256_column_tile(name, mt, terms):
    _run(name, {0: (1, mt, 2, 0), 1: (1, mt, 2, 0), 2: (0, 2, 8, 0)})


@pytest.mark.parametrize("mt,blocks", [(4, 4), (8, 8), (9, 8), (9, 16)])
@pytest.mark.parametrize("name", list(GEOMS))
def test_split_weight_gradient(name, mt, blocks, terms):
    _run(name, {0: (0, 2, 1, 1), 1: (0, 2, 1, 1), 2: (1, mt, blocks, 0)})


@pytest.mark.parametrize("tile", [(0, 1, 1, 1), (0, 3, 1, 2), (0, 5, 1, 1), (0, 2, 2, 1), (0, 1, 4, 2)])
@pytest.mark.parametrize("name", ["S1", "T2s", "odd"])
def test_native_tiles(name, tile):
    _run(name, {0: tile, 1: tile, 2: (0, 3, 4, 0)})


def test_query_and_set_tile_roundtrip():
    import ctypes
    from cstp_amd import _lib, ops
    lib = _lib.load()
    xs, k, ks, st, pd = GEOMS["S1"]
    ops.set_conv_tile(xs, (k, xs[1]) + ks, st, pd, 0, (1, 9, 0, 0))
    desc = ops._desc(xs, (k, xs[1]) + ks, st, pd)
    out = (ctypes.c_int32 * 4)()
    _lib.check(lib.cstp_conv3d_query_tile(ctypes.byref(desc), 0, out), "query")
    assert list(out) == [144, 128, lib.cstp_gemm_get_split_terms(), 1] and out[2] in (2, 3)
    ops.set_split_terms(3)              # every arithmetic has its own class of pinned / tuned tiles
    _lib.check(lib.cstp_conv3d_query_tile(ctypes.byref(desc), 0, out), "query")
    assert out[2] == 0 and lib.cstp_gemm_get_split_terms() == 3          # nothing pinned in the bf16-triple class yet
    ops.set_conv_tile(xs, (k, xs[1]) + ks, st, pd, 0, (1, 9, 0, 0))
    _lib.check(lib.cstp_conv3d_query_tile(ctypes.byref(desc), 0, out), "query")
    assert list(out) == [144, 128, 3, 1]
    got = (ctypes.c_int32 * 4)()
    _lib.check(lib.cstp_conv3d_get_tile(ctypes.byref(desc), 0, got), "get")
    assert list(got) == [1, 9, 1, 1]                                      # set_tile's encoding, replayable
    ops.set_split_terms(1)              # native f32 MFMA only
    _lib.check(lib.cstp_conv3d_query_tile(ctypes.byref(desc), 0, out), "query")
    assert out[2] == 0 and lib.cstp_gemm_get_split_terms() == 1
    _lib.check(lib.cstp_conv3d_get_tile(ctypes.byref(desc), 0, got), "get")
    assert got[0] == -1                                                   # untuned in this class
    ops.set_split_terms(0)
    with pytest.raises(_lib.CstpError):
        ops.set_split_terms(4)
    ops.set_conv_tile(xs, (k, xs[1]) + ks, st, pd, 0, (0, 2, 2, 2))
    _lib.check(lib.cstp_conv3d_query_tile(ctypes.byref(desc), 0, out), "query")
    assert list(out) == [128, 64, 0, 2]
    with pytest.raises(_lib.CstpError):
        ops.set_conv_tile(xs, (k, xs[1]) + ks, st, pd, 0, (1, 7, 0, 0))        # no 112-row split tile
    with pytest.raises(_lib.CstpError):
        ops.set_conv_tile(xs, (k, xs[1]) + ks, st, pd, 2, (1, 5, 8, 0))


# ---- the f16-pair arithmetic: operand scales, the largest-magnitude cells, their producers ----------------------------------
@pytest.mark.parametrize("scale_x,scale_w,scale_dy", [(1e-20, 1.0, 1e10), (3e18, 1e-12, 1e-10), (1e-3, 7e9, 1e-5),
                                                      (1.0, 1e-25, 1e5)])
@pytest.mark.parametrize("name", ["S1", "T2s", "lin"])
def test_f16_pair_is_scale_invariant(name, scale_x, scale_w, scale_dy):
    """Operands far outside f16 range: the power-of-two operand scales (per tensor for activations / gradients, per row for the
    packed weights) bring them in, so parity holds at any magnitude fp32 itself can represent."""
    from cstp_amd import ops
    ops.set_split_terms(2)
    try:
        _run(name, {0: (1, 9, 0, 0), 1: (1, 4, 0, 0), 2: (1, 8, 8, 0)}, scale_x=scale_x, scale_w=scale_w, scale_dy=scale_dy)
    finally:
        ops.set_split_terms(0)


def test_f16_pair_rows_of_very_different_magnitude():
    """Weight rows (output channels) 12 orders of magnitude apart: each row carries its own scale, every output channel keeps
    its relative accuracy; activations whose channels differ by 1e4 share one scale and stay within the bar."""
    from cstp_amd import ops
    ops.set_split_terms(2)
    try:
        xs, k, ks, st, pd = GEOMS["S1"]
        x = _rand(xs, 5) * torch.logspace(-2, 2, xs[1], dtype=torch.float64).view(1, -1, 1, 1, 1)
        w = _rand((k, xs[1]) + ks, 6) * torch.logspace(-6, 6, k, dtype=torch.float64).view(-1, 1, 1, 1, 1)
        ops.set_conv_tile(xs, w.shape, st, pd, 0, (1, 9, 2, 0))
        y = F.conv3d(x, w, None, st, pd)
        yg = ops.conv3d(x.float().cuda(), w.float().cuda(), None, st, pd).double().cpu()
        per_row = (yg - y).abs().amax(dim=(0, 2, 3, 4)) / y.abs().amax(dim=(0, 2, 3, 4))
        assert float(per_row.max()) < TOL
    finally:
        ops.set_split_terms(0)


def test_absmax_cells_from_batchnorm_feed_the_convolution():
    """cstp_bn_forward_train_am / cstp_bn_backward_am leave max |y| / max |dx| behind; handing that cell to the convolution gives
    the bits the self-measured path gives; a stale tag (in-place update) is dropped."""
    from cstp_amd import ops
    ops.set_split_terms(2)
    try:
        torch.manual_seed(0)
        x = (torch.randn(4, 64, 4, 14, 14, device="cuda") * 3).requires_grad_(True)
        g = torch.rand(64, device="cuda") + 0.5
        b = torch.randn(64, device="cuda")
        w = (torch.randn(144, 64, 1, 3, 3, device="cuda") * 0.05).requires_grad_(True)
        ops.set_conv_tile(x.shape, w.shape, (1, 1, 1), (0, 1, 1), 0, (1, 9, 0, 0))
        ops.set_conv_tile(x.shape, w.shape, (1, 1, 1), (0, 1, 1), 1, (1, 4, 0, 0))
        ops.set_conv_tile(x.shape, w.shape, (1, 1, 1), (0, 1, 1), 2, (1, 9, 8, 0))
        y = ops.batch_norm_act(x, g, b, relu=True, groups=2)
        cell, ver = y._cstp_absmax
        assert ver == y._version
        assert int(cell.item()) == int(y.detach().abs().max().view(torch.int32).item())       # fp32 bits of max |y|
        before = dict(ops.absmax_stats)
        z = ops.conv3d(y, w, None, 1, (0, 1, 1))
        assert ops.absmax_stats["hit"] == before["hit"] + 1
        z2 = ops.conv3d(y.detach().clone(), w.detach(), None, 1, (0, 1, 1))                     # untagged: measured inside
        assert ops.absmax_stats["miss"] == before["miss"] + 1
        assert torch.equal(z.detach(), z2)
        # backward: the BN backward of a following BN tags dz; conv backward consumes it; results equal the untagged run
        g2 = torch.rand(144, device="cuda") + 0.5
        b2 = torch.randn(144, device="cuda")
        out = ops.batch_norm_act(z, g2, b2, relu=True, groups=2)
        hits = ops.absmax_stats["hit"]
        out.square().sum().backward()
        assert ops.absmax_stats["hit"] >= hits + 1
        gx, gw = x.grad.clone(), w.grad.clone()
        x.grad = None; w.grad = None
        ops.FUSE_ABSMAX = False
        try:
            y = ops.batch_norm_act(x, g, b, relu=True, groups=2)
            assert not hasattr(y, "_cstp_absmax")
            out = ops.batch_norm_act(ops.conv3d(y, w, None, 1, (0, 1, 1)), g2, b2, relu=True, groups=2)
            out.square().sum().backward()
        finally:
            ops.FUSE_ABSMAX = True
        assert torch.equal(gx, x.grad)
        assert rel_err(gw, w.grad) < 1e-5      # split-K f32 atomics: the summation order differs from run to run
        # a tag does not survive an in-place update of its tensor
        y = ops.batch_norm_act(x.detach(), g, b, relu=True, groups=2)
        y.mul_(2.0)
        assert ops._absmax_of(y) is None
    finally:
        ops.set_split_terms(0)


def test_absmax_of_unaligned_ragged_and_zero_tensors():
    """The self-measuring path on views that start off a 16-byte boundary, sizes that are not multiples of 4, and all-zero
    operands (scale 1, result exactly 0)."""
    from cstp_amd import ops
    ops.set_split_terms(2)
    try:
        base = torch.randn(1 + 2 * 33 * 2 * 5 * 7, device="cuda")
        base[17] = 900.0                                  # the maximum sits in the unaligned head / body
        x = base[1:].view(2, 33, 2, 5, 7)
        w = torch.randn(20, 33, 1, 3, 3, device="cuda") * 0.1
        ops.set_conv_tile(x.shape, w.shape, (1, 1, 1), (0, 1, 1), 0, (1, 2, 0, 0))
        y = ops.conv3d(x, w, None, 1, (0, 1, 1))
        ref = F.conv3d(x.double().cpu(), w.double().cpu(), None, 1, (0, 1, 1))
        assert rel_err(y, ref) < TOL
        z = ops.conv3d(torch.zeros_like(x), w, None, 1, (0, 1, 1))
        assert float(z.abs().max()) == 0.0
        z = ops.conv3d(x, torch.zeros_like(w), None, 1, (0, 1, 1))
        assert float(z.abs().max()) == 0.0
    finally:
        ops.set_split_terms(0)


# ---- the LDS-resident-patch kernel igemm_k1p (csrc/igemm_patch.h): stride-1 1x3x3 layers, forward and data gradient ----------
PATCH_GEOMS = {
    "S1": ((2, 64, 4, 14, 14), 144),        # 14x14 frames: tiles of 224 positions straddle frames
    "S1big": ((1, 64, 2, 56, 56), 144),     # the real S1 frame: tiles = 4 image rows
    "S3": ((2, 128, 2, 28, 28), 288),       # two row blocks of 144
    "S7": ((1, 512, 2, 7, 7), 1152),        # 16 channel blocks, one ragged tile of 98 positions spanning two frames
    "ragged": ((3, 40, 3, 9, 11), 136),     # odd frame size, last channel block 8 of 32, rows padded 136 -> 144
    "thin": ((2, 16, 1, 5, 6), 24),         # half a channel block, 60 positions
}


@pytest.mark.parametrize("mt_f,mt_d", [(9, 4), (4, 8), (8, 9)])
@pytest.mark.parametrize("name", list(PATCH_GEOMS))
def test_patch_kernel_forward_and_data_gradient(name, mt_f, mt_d):
    from cstp_amd import ops
    xs, k = PATCH_GEOMS[name]
    GEOMS["_patch"] = (xs, k, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    ops.set_split_terms(2)
    try:
        _run("_patch", {0: (2, mt_f, 0, 0), 1: (2, mt_d, 0, 0), 2: (0, 2, 8, 0)})
        import ctypes
        from cstp_amd import _lib
        out = (ctypes.c_int32 * 4)()
        desc = ops._desc(xs, (k, xs[1], 1, 3, 3), (1, 1, 1), (0, 1, 1))
        _lib.check(_lib.load().cstp_conv3d_query_tile(ctypes.byref(desc), 0, out), "query")
        assert list(out)[:3] == [16 * mt_f, 224, 2]          # the pinned patch tile is what runs
    finally:
        ops.set_split_terms(0)
        del GEOMS["_patch"]


def test_patch_kernel_scale_invariance_and_zero_operands():
    from cstp_amd import ops
    xs, k = PATCH_GEOMS["S1"]
    GEOMS["_patch"] = (xs, k, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    ops.set_split_terms(2)
    try:
        for sx, sw, sdy in ((1e-20, 1.0, 1e10), (3e18, 1e-12, 1e-10)):
            _run("_patch", {0: (2, 9, 0, 0), 1: (2, 4, 0, 0), 2: (0, 2, 8, 0)}, scale_x=sx, scale_w=sw, scale_dy=sdy)
        x = torch.randn(xs, device="cuda")
        w = torch.randn(k, xs[1], 1, 3, 3, device="cuda")
        assert float(ops.conv3d(torch.zeros_like(x), w, None, 1, (0, 1, 1)).abs().max()) == 0.0
        assert float(ops.conv3d(x, torch.zeros_like(w), None, 1, (0, 1, 1)).abs().max()) == 0.0
    finally:
        ops.set_split_terms(0)
        del GEOMS["_patch"]


def test_patch_tile_is_refused_where_the_kernel_does_not_apply():
    """A patch tile pinned on a strided / temporal / bf16-triple call falls back to another kernel at call time."""
    from cstp_amd import ops
    for name in ("S2s", "T1"):
        _run(name, {0: (2, 9, 0, 0), 1: (2, 4, 0, 0), 2: (0, 2, 8, 0)})
    xs, k = PATCH_GEOMS["S1"]
    GEOMS["_patch"] = (xs, k, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    ops.set_split_terms(3)
    try:
        _run("_patch", {0: (2, 9, 0, 0), 1: (2, 4, 0, 0), 2: (0, 2, 8, 0)})
    finally:
        ops.set_split_terms(0)
        del GEOMS["_patch"]


# ---- the weight gradient on the LDS-resident x ring, igemm_k2p (csrc/igemm_wpatch.h): the same layers ------------------------
WPATCH_GEOMS = dict(PATCH_GEOMS)
WPATCH_GEOMS.update({
    "S5": ((2, 256, 2, 14, 14), 576),       # 8 channel blocks x 4 row blocks: more (row, channel) pairs than frames per block
    "wide": ((1, 32, 3, 6, 120), 48),       # W = 120: the x staging leads by five 64-row intervals (the most the kernel takes)
    "one frame": ((1, 32, 1, 28, 28), 144), # a single frame: more blocks than frames (the surplus blocks return at once)
})


@pytest.mark.parametrize("name", list(WPATCH_GEOMS))
def test_patch_kernel_weight_gradient(name):
    from cstp_amd import ops
    xs, k = WPATCH_GEOMS[name]
    GEOMS["_wpatch"] = (xs, k, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    ops.set_split_terms(2)
    try:
        _run("_wpatch", {0: (0, 2, 1, 1), 1: (0, 2, 1, 1), 2: (2, 9, 1, 0)})
        import ctypes
        from cstp_amd import _lib
        out = (ctypes.c_int32 * 4)()
        desc = ops._desc(xs, (k, xs[1], 1, 3, 3), (1, 1, 1), (0, 1, 1))
        _lib.check(_lib.load().cstp_conv3d_query_tile(ctypes.byref(desc), 2, out), "query")
        assert list(out)[:3] == [144, 288, 2]                # the pinned patch kernel is what runs
        tile = (ctypes.c_int32 * 4)()
        _lib.check(_lib.load().cstp_conv3d_get_tile(ctypes.byref(desc), 2, tile), "get")
        assert list(tile)[:2] == [2, 9]
        for sx, sw, sdy in ((1e-20, 1.0, 1e10), (3e18, 1e-12, 1e-10)):
            _run("_wpatch", {2: (2, 9, 1, 0)}, scale_x=sx, scale_w=sw, scale_dy=sdy)
    finally:
        ops.set_split_terms(0)
        del GEOMS["_wpatch"]


def test_patch_weight_gradient_is_refused_where_the_kernel_does_not_apply():
    """Pinned on a strided / temporal / too-wide / bf16-triple call it falls back to another kernel at call time."""
    from cstp_amd import ops
    for name in ("S2s", "T1", "odd"):
        _run(name, {2: (2, 9, 1, 0)})
    GEOMS["_wpatch"] = ((1, 16, 1, 4, 130), 16, (1, 3, 3), (1, 1, 1), (0, 1, 1))       # W = 130 > 125
    try:
        _run("_wpatch", {2: (2, 9, 1, 0)})
        GEOMS["_wpatch"] = PATCH_GEOMS["S1"] + ((1, 3, 3), (1, 1, 1), (0, 1, 1))
        ops.set_split_terms(3)
        _run("_wpatch", {2: (2, 9, 1, 0)})
    finally:
        ops.set_split_terms(0)
        del GEOMS["_wpatch"]


# ---- BatchNorm statistics as a by-product of the patch kernel's forward launch -------------------------------------------------
BNSTAT_GEOMS = {
    # (x shape, k, row-tile height): positions per BN group are a multiple of 224 in all of them
    "one row block": ((4, 16, 2, 28, 28), 144, 9),        # 28 tiles over 8 XCDs: XCD 7 owns none (its blocks write zeros)
    "two row blocks": ((4, 40, 2, 28, 28), 288, 9),       # a block meets one row block only (slots % 2 == 0)
    "128-row tiles": ((2, 32, 4, 28, 28), 256, 8),
    "64-row tiles": ((2, 32, 1, 56, 56), 64, 4),
    "ragged rows": ((4, 16, 2, 28, 28), 136, 9),          # 136 channels in a 144-row block: the padded rows leave no sums
}


@pytest.mark.gpu
@pytest.mark.parametrize("groups", [1, 2])
@pytest.mark.parametrize("name", list(BNSTAT_GEOMS))
def test_patch_kernel_leaves_batchnorm_statistics(name, groups):
    """conv3d(..., bn_groups) -> batch_norm_act: the BatchNorm takes its sums from the convolution's epilogue
    (cstp_conv3d_forward_bnstats -> cstp_bn_forward_train_pre) and must equal both the separate-pass result and fp64."""
    import ctypes
    from cstp_amd import _lib, ops
    xs, k, mt = BNSTAT_GEOMS[name]
    ws = (k, xs[1], 1, 3, 3)
    lib = _lib.load()
    ops.set_split_terms(2)
    ops.set_conv_tile(xs, ws, (1, 1, 1), (0, 1, 1), 0, (2, mt, 0, 0))
    try:
        desc = ops._desc(xs, ws, (1, 1, 1), (0, 1, 1))
        assert lib.cstp_conv3d_bnstats_nsplit(ctypes.byref(desc), groups) > 0
        assert lib.cstp_conv3d_bnstats_nsplit(ctypes.byref(desc), 3) == 0            # at most two groups
        g = torch.Generator().manual_seed(5)
        x = (torch.randn(xs, generator=g) + 0.3).cuda()
        w = (torch.randn(ws, generator=g) * 0.1).cuda()
        gamma, beta = (torch.rand(k, generator=g) + 0.5).cuda(), torch.randn(k, generator=g).cuda()

        def run(fused):
            rm, rv = torch.zeros(k, device="cuda"), torch.ones(k, device="cuda")
            y = ops.conv3d(x, w, None, 1, (0, 1, 1), bn_groups=groups if fused else 0)
            assert (getattr(y, "_cstp_bnstats", None) is not None) == fused
            return ops.batch_norm_act(y, gamma, beta, rm, rv, None, True, 1e-5, 0.1, groups), rm, rv

        a, rma, rva = run(True)
        b, rmb, rvb = run(False)
        assert rel_err(a, b) < 2e-6 and rel_err(rma, rmb) < 2e-6 and rel_err(rva, rvb) < 2e-6
        # fp64 truth
        yc = F.conv3d(x.double().cpu(), w.double().cpu(), None, 1, (0, 1, 1))
        outs, rm, rv = [], torch.zeros(k, dtype=torch.float64), torch.ones(k, dtype=torch.float64)
        for part in yc.chunk(groups, 0):
            outs.append(F.relu(F.batch_norm(part, rm, rv, gamma.double().cpu(), beta.double().cpu(), True, 0.1, 1e-5)))
        ref = torch.cat(outs, 0)
        assert rel_err(a.cpu().double(), ref) < 1e-4
        assert rel_err(rma.cpu().double(), rm) < 1e-4 and rel_err(rva.cpu().double(), rv) < 1e-4
    finally:
        ops.set_split_terms(0)


@pytest.mark.gpu
@pytest.mark.parametrize("groups", [1, 2])
def test_patch_kernel_batchnorm_statistics_of_channels_whose_mean_is_far_above_their_spread(groups):
    """Round-2 ADVICE: the fused sums are fp32 per lane before they reach fp64, and var = E[y^2] - mean^2 cancels when
    |mean| >> std.  Channels here sit at |mean| / std ~ 1e2 (mean 14, std 0.12).  With the BatchNorm's running mean as the
    PIVOT the sums are taken around (cstp_conv3d_forward_bnstats: sum(y - c), sum((y - c)^2)) the fused path holds the
    same 1e-4 bar against fp64 as the separate fp64-accumulating pass -- also with a pivot that is 3 std off, and the
    running statistics come out right."""
    from cstp_amd import ops
    xs, k, mt = BNSTAT_GEOMS["one row block"]
    ws = (k, xs[1], 1, 3, 3)
    ops.set_split_terms(2)
    ops.set_conv_tile(xs, ws, (1, 1, 1), (0, 1, 1), 0, (2, mt, 0, 0))
    try:
        g = torch.Generator().manual_seed(9)
        # (the mean rides on the centre tap, which zero padding does not touch at the image borders)
        x = (1.0 + 0.03 * torch.randn(xs, generator=g)).cuda()
        w = 0.002 * torch.randn(ws, generator=g)
        w[:, :, 0, 1, 1] = 0.85 + 0.1 * torch.rand((k, xs[1]), generator=g)
        w = w.cuda()
        gamma, beta = (torch.rand(k, generator=g) + 0.5).cuda(), torch.randn(k, generator=g).cuda()
        yc = F.conv3d(x.double().cpu(), w.double().cpu(), None, 1, (0, 1, 1))
        mean, std = yc.mean(dim=(0, 2, 3, 4)), yc.std(dim=(0, 2, 3, 4))
        assert float((mean.abs() / std).min()) > 50
        pivot0 = (mean + 3 * std).float()               # a running mean that is off by three standard deviations

        def truth():
            outs, rm, rv = [], pivot0.double().clone(), torch.ones(k, dtype=torch.float64)
            for part in yc.chunk(groups, 0):
                outs.append(F.relu(F.batch_norm(part, rm, rv, gamma.double().cpu(), beta.double().cpu(), True, 0.1, 1e-5)))
            return torch.cat(outs, 0), rm, rv

        def run(fused, pivot):
            rm, rv = pivot0.cuda().clone(), torch.ones(k, device="cuda")
            y = ops.conv3d(x, w, None, 1, (0, 1, 1), bn_groups=groups if fused else 0, bn_pivot=rm if pivot else None)
            assert (getattr(y, "_cstp_bnstats", None) is not None) == fused
            return ops.batch_norm_act(y, gamma, beta, rm, rv, None, True, 1e-5, 0.1, groups), rm, rv

        ref, rm64, rv64 = truth()
        sep, _, _ = run(False, False)
        piv, rmp, rvp = run(True, True)
        nop, _, _ = run(True, False)
        e_sep, e_piv, e_nop = rel_err(sep.cpu().double(), ref), rel_err(piv.cpu().double(), ref), rel_err(nop.cpu().double(), ref)
        print("BN output vs fp64 at |mean|/std ~ %.0f: separate pass %.2e, fused around the pivot %.2e, fused around zero %.2e"
              % (float((mean.abs() / std).mean()), e_sep, e_piv, e_nop))
        assert e_sep < 1e-4 and e_piv < 1e-4
        assert e_piv < 2 * e_sep + 2e-6                  # the fusion costs no accuracy
        assert rel_err(rmp.cpu().double(), rm64) < 1e-6 and rel_err(rvp.cpu().double(), rv64) < 1e-4
    finally:
        ops.set_split_terms(0)


@pytest.mark.gpu
def test_batchnorm_statistics_by_product_is_declined_where_it_cannot_be_exact():
    """Tiles that straddle two BN groups, a gather-kernel tile, or frames that rule out 16-byte stores: nsplit = 0 and the
    call is the plain forward (BatchNorm then makes its own pass)."""
    import ctypes
    from cstp_amd import _lib, ops
    lib = _lib.load()
    ops.set_split_terms(2)
    try:
        for xs, k, tile, groups in (((2, 16, 1, 14, 14), 144, (2, 9, 0, 0), 2),      # 196 positions per group
                                    ((4, 16, 2, 28, 28), 144, (1, 9, 0, 0), 2),      # gather kernel pinned
                                    ((2, 16, 32, 7, 7), 144, (2, 9, 0, 0), 1)):      # 7 x 7 frames (1568 = 7 * 224, but 49 % 4 != 0)
            ws = (k, xs[1], 1, 3, 3)
            ops.set_conv_tile(xs, ws, (1, 1, 1), (0, 1, 1), 0, tile)
            desc = ops._desc(xs, ws, (1, 1, 1), (0, 1, 1))
            assert lib.cstp_conv3d_bnstats_nsplit(ctypes.byref(desc), groups) == 0
            x, w = torch.randn(xs, device="cuda"), torch.randn(ws, device="cuda") * 0.1
            y = ops.conv3d(x, w, None, 1, (0, 1, 1), bn_groups=groups)
            assert getattr(y, "_cstp_bnstats", None) is None
            assert rel_err(y, ops.conv3d(x, w, None, 1, (0, 1, 1))) == 0.0
    finally:
        ops.set_split_terms(0)


# ---- weight gradient: accumulation into the caller's buffer, and the deterministic (two-stage split-K) mode -------------------
def _wgrad_call(lib, ops, desc, x, dy, dw, accumulate):
    import ctypes
    ws = torch.empty(lib.cstp_conv3d_workspace_bytes(ctypes.byref(desc)), dtype=torch.uint8, device="cuda")
    ops.check(lib.cstp_conv3d_backward_weight_acc(torch.cuda.current_stream().cuda_stream, ctypes.byref(desc), x.data_ptr(), None,
                                                  dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(), None, None, accumulate),
              "cstp_conv3d_backward_weight_acc")


@pytest.mark.parametrize("tile", [(1, 9, 8, 0), (0, 3, 4, 0), (2, 9, 1, 0)], ids=["split", "native", "patch"])
def test_weight_gradient_accumulate_flag(tile):
    from cstp_amd import _lib, ops
    lib = _lib.load()
    xs, k, ks, st, pd = GEOMS["S1"]
    x = _rand(xs, 11).float().cuda()
    w = (_rand((k, xs[1]) + ks, 12) * 0.2).double().requires_grad_(True)
    y = F.conv3d(x.double().cpu(), w, None, st, pd)
    dy = _rand(y.shape, 13)
    y.backward(dy)
    ops.set_conv_tile(xs, w.shape, st, pd, 2, tile)
    desc = ops._desc(xs, tuple(w.shape), st, pd)
    base = torch.full(w.shape, 0.75, device="cuda")
    dw = base.clone()
    _wgrad_call(lib, ops, desc, x, dy.float().cuda(), dw, 1)
    assert rel_err(dw - base, w.grad) < TOL                      # dw += gradient
    _wgrad_call(lib, ops, desc, x, dy.float().cuda(), dw, 0)
    assert rel_err(dw, w.grad) < TOL                             # dw = gradient


@pytest.mark.parametrize("tile", [(1, 9, 8, 0), (1, 4, 16, 0), (0, 3, 4, 0), (0, 9, 8, 0), (2, 9, 1, 0)],
                         ids=["split9", "split4", "native3", "native144", "patch"])
def test_deterministic_weight_gradient_is_bit_reproducible(tile):
    """CSTP_DETERMINISTIC / cstp_set_deterministic: per-split slabs summed in a fixed order instead of f32 atomics -- the same
    inputs give the same bits, and the result keeps the parity bar."""
    from cstp_amd import _lib, ops
    lib = _lib.load()
    xs, k, ks, st, pd = (4, 64, 8, 28, 28), 144, (1, 3, 3), (1, 1, 1), (0, 1, 1)
    x = _rand(xs, 21).float().cuda()
    w = (_rand((k, xs[1]) + ks, 22) * 0.2).double().requires_grad_(True)
    y = F.conv3d(x.double().cpu(), w, None, st, pd)
    dy = _rand(y.shape, 23)
    y.backward(dy)
    dyg = dy.float().cuda()
    ops.set_conv_tile(xs, w.shape, st, pd, 2, tile)
    desc = ops._desc(xs, tuple(w.shape), st, pd)
    ops.set_deterministic(True)
    try:
        assert lib.cstp_get_deterministic() == 1
        runs = []
        for _ in range(3):
            dw = torch.empty(w.shape, device="cuda")
            _wgrad_call(lib, ops, desc, x, dyg, dw, 0)
            runs.append(dw)
        assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
        assert rel_err(runs[0], w.grad) < TOL
    finally:
        ops.set_deterministic(False)
    assert lib.cstp_get_deterministic() == 0


# ---- the 3-channel stems on the split kernel's straddle mode (zero-padded input copy, per-k offset table) ---------------------
STEMS = {
    "r21d": ((2, 3, 4, 28, 28), 83, (1, 7, 7), (1, 2, 2), (0, 3, 3)),        # R(2+1)D stem: 147-long reduction, 83 rows -> 96
    "r21d_odd": ((3, 3, 3, 23, 19), 83, (1, 7, 7), (1, 2, 2), (0, 3, 3)),    # odd extents, ragged last tile
    "r3d": ((1, 3, 6, 20, 20), 64, (7, 7, 7), (1, 2, 2), (3, 3, 3)),         # 3D-ResNet stem: 1029-long reduction, padding in D too
}


@pytest.mark.parametrize("mt", [4, 5, 6])
@pytest.mark.parametrize("name", list(STEMS))
def test_stem_on_the_split_kernel(name, mt):
    from cstp_amd import ops
    import ctypes
    from cstp_amd import _lib
    GEOMS["_stem"] = STEMS[name]
    xs, k, ks, st, pd = STEMS[name]
    ops.set_split_terms(2)
    try:
        _run("_stem", {0: (1, mt, 0, 0), 1: (0, 2, 1, 1), 2: (0, 3, 4, 0)})
        out = (ctypes.c_int32 * 4)()
        desc = ops._desc(xs, (k, xs[1]) + ks, st, pd)
        _lib.check(_lib.load().cstp_conv3d_query_tile(ctypes.byref(desc), 0, out), "query")
        assert list(out)[:3] == [16 * mt, 128, 2]
        _run("_stem", {0: (1, mt, 0, 0), 1: (0, 2, 1, 1), 2: (0, 3, 4, 0)}, scale_x=3e15, scale_w=1e-9)
    finally:
        ops.set_split_terms(0)
        del GEOMS["_stem"]


@pytest.mark.parametrize("mt,blocks", [(4, 4), (8, 8), (9, 16)])
@pytest.mark.parametrize("name", list(STEMS))
def test_stem_weight_gradient_on_the_split_kernel(name, mt, blocks):
    """igemm_k2s<.., STR>: the stems' weight gradient over the zero-padded input copy, columns (tap, channel) without channel
    padding (round 3); pinned, queried back, scale-invariant, and bit-reproducible in the deterministic mode."""
    import ctypes
    from cstp_amd import _lib, ops
    GEOMS["_stem"] = STEMS[name]
    xs, k, ks, st, pd = STEMS[name]
    ops.set_split_terms(2)
    try:
        _run("_stem", {0: (0, 2, 1, 1), 2: (1, mt, blocks, 0)})
        out = (ctypes.c_int32 * 4)()
        desc = ops._desc(xs, (k, xs[1]) + ks, st, pd)
        _lib.check(_lib.load().cstp_conv3d_query_tile(ctypes.byref(desc), 2, out), "query")
        assert list(out) == [16 * mt, 128, 2, blocks]
        _run("_stem", {0: (0, 2, 1, 1), 2: (1, mt, blocks, 0)}, scale_x=3e15, scale_dy=1e-9)
        ops.set_deterministic(True)
        try:
            lib = _lib.load()
            x, dy = _rand(xs, 31).float().cuda(), None
            w = torch.zeros((k, xs[1]) + ks)
            dy = _rand(F.conv3d(x.cpu().double(), w.double(), None, st, pd).shape, 33).float().cuda()
            runs = []
            for _ in range(2):
                dw = torch.empty(w.shape, device="cuda")
                _wgrad_call(lib, ops, desc, x, dy, dw, 0)
                runs.append(dw)
            assert torch.equal(runs[0], runs[1])
        finally:
            ops.set_deterministic(False)
    finally:
        ops.set_split_terms(0)
        del GEOMS["_stem"]


@pytest.mark.parametrize("name,tile", [("S1", (2, 4, 0, 0)), ("S1", (1, 4, 0, 0)), ("S1", (0, 2, 1, 1)), ("S2s", (1, 9, 0, 0)),
                                       ("S2s", (0, 2, 2, 1)), ("T2s", (1, 4, 0, 0)), ("odd", (1, 3, 0, 0)), ("lin", (1, 2, 0, 0)),
                                       ("short", (1, 4, 0, 0)), ("short", (0, 2, 2, 1)), ("lin64", (1, 2, 0, 0))],
                         ids=["patch", "split", "native", "split strided", "native strided", "split temporal strided", "split odd",
                              "split linear", "split strided pointwise", "native strided pointwise", "streaming linear"])
def test_data_gradient_accumulate_flag(name, tile):
    """cstp_conv3d_backward_data_acc: dx += the data gradient (the sum of a residual connection's two gradients formed in the
    convolution's epilogue, ops.GradJoin) on every kernel family, incl. the stride-parity classes of a strided layer."""
    import ctypes
    from cstp_amd import _lib, ops
    lib = _lib.load()
    xs, k, ks, st, pd = GEOMS[name] if name != "S1" else (PATCH_GEOMS["S1big"][0], PATCH_GEOMS["S1big"][1], (1, 3, 3), (1, 1, 1), (0, 1, 1))
    ops.set_split_terms(2)
    try:
        x = _rand(xs, 41).requires_grad_(True)
        w = (_rand((k, xs[1]) + ks, 42) * 0.2)
        y = F.conv3d(x, w, None, st, pd)
        dy = _rand(y.shape, 43)
        y.backward(dy)
        ops.set_conv_tile(xs, tuple(w.shape), st, pd, 1, tile)
        desc = ops._desc(xs, tuple(w.shape), st, pd)
        ws = torch.empty(lib.cstp_conv3d_workspace_bytes(ctypes.byref(desc)), dtype=torch.uint8, device="cuda")
        base = _rand(xs, 44).float().cuda()
        dx = base.clone()
        call = lambda acc: _lib.check(lib.cstp_conv3d_backward_data_acc(   # noqa: E731
            torch.cuda.current_stream().cuda_stream, ctypes.byref(desc), dy.float().cuda().data_ptr(), w.float().cuda().data_ptr(),
            dx.data_ptr(), ws.data_ptr(), ws.numel(), None, acc), "cstp_conv3d_backward_data_acc")
        dyg, wg = dy.float().cuda(), w.float().cuda()
        _lib.check(lib.cstp_conv3d_backward_data_acc(torch.cuda.current_stream().cuda_stream, ctypes.byref(desc), dyg.data_ptr(),
                                                     wg.data_ptr(), dx.data_ptr(), ws.data_ptr(), ws.numel(), None, 1), "acc")
        assert rel_err(dx - base, x.grad) < TOL                      # dx += gradient
        _lib.check(lib.cstp_conv3d_backward_data_acc(torch.cuda.current_stream().cuda_stream, ctypes.byref(desc), dyg.data_ptr(),
                                                     wg.data_ptr(), dx.data_ptr(), ws.data_ptr(), ws.numel(), None, 0), "set")
        assert rel_err(dx, x.grad) < TOL                             # dx = gradient
    finally:
        ops.set_split_terms(0)


def test_streaming_linear_weight_gradient_accumulate_flag():
    """csrc/linear.h's weight gradient (n <= 32 rows): dw += and dw = through cstp_conv3d_backward_weight_acc, ragged row count
    (k = 40: the last block of 16 rows is partial) and whole blocks (k = 128)."""
    from cstp_amd import _lib, ops
    lib = _lib.load()
    for k in (40, 128):
        xs = (6, 192, 1, 1, 1)
        x = _rand(xs, 71).float().cuda()
        w = (_rand((k, xs[1], 1, 1, 1), 72) * 0.2).double().requires_grad_(True)
        y = F.conv3d(x.double().cpu(), w, None, 1, 0)
        dy = _rand(y.shape, 73)
        y.backward(dy)
        desc = ops._desc(xs, tuple(w.shape), (1, 1, 1), (0, 0, 0))
        base = torch.full(w.shape, 0.75, device="cuda")
        dw = base.clone()
        _wgrad_call(lib, ops, desc, x, dy.float().cuda(), dw, 1)
        assert rel_err(dw - base, w.grad) < TOL
        _wgrad_call(lib, ops, desc, x, dy.float().cuda(), dw, 0)
        assert rel_err(dw, w.grad) < TOL
