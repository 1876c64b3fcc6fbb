"""Parity of EVERY GEMM kernel variant against PyTorch CPU fp64, independent of which one the autotuner would pick on the
machine at hand: each test pins one variant for one geometry through cstp_conv3d_set_tile (C ABI) and runs forward, data
gradient and weight gradient through the normal autograd path.  Variants: the native f32 MFMA tiles and the 3xbf16-split
kernels (igemm_k1s row tiles 2..9 x 16, igemm_k2s 64/128/144-row tiles).  Tolerance: the 1e-4 bar of BASELINE.json (measured
errors are ~1e-6, tools/split_accuracy.py)."""
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
}


def _rand(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1


def _run(name, tiles):
    """tiles: {mode: tile4}.  Returns nothing; asserts parity of y, dx, dw."""
    from cstp_amd import ops
    xs, k, ks, st, pd = GEOMS[name]
    x = _rand(xs, 1).requires_grad_(True)
    w = (_rand((k, xs[1]) + ks, 2) * 0.2).requires_grad_(True)
    y = F.conv3d(x, w, None, st, pd)
    dy = _rand(y.shape, 3)
    y.backward(dy)
    for mode, tile in tiles.items():
        ops.set_conv_tile(xs, w.shape, st, pd, mode, tile)
    xg = x.detach().float().cuda().requires_grad_(True)
    wg = w.detach().float().cuda().requires_grad_(True)
    yg = ops.conv3d(xg, wg, None, st, pd)
    yg.backward(dy.float().cuda())
    assert rel_err(yg, y) < TOL, ("forward", name, tiles)
    assert rel_err(xg.grad, x.grad) < TOL, ("backward_data", name, tiles)
    assert rel_err(wg.grad, w.grad) < TOL, ("backward_weight", name, tiles)


@pytest.mark.parametrize("mt", [2, 3, 4, 5, 6, 8, 9])
@pytest.mark.parametrize("name", list(GEOMS))
def test_split_forward_and_data_gradient(name, mt):
    _run(name, {0: (1, mt, 0, 0), 1: (1, mt, 0, 0), 2: (0, 2, 8, 0)})


@pytest.mark.parametrize("mt", [8, 9])
@pytest.mark.parametrize("name", list(GEOMS))
def test_split_256_column_tile(name, mt):
    _run(name, {0: (1, mt, 2, 0), 1: (1, mt, 2, 0), 2: (0, 2, 8, 0)})


@pytest.mark.parametrize("mt,blocks", [(4, 4), (8, 8), (9, 8), (9, 16)])
@pytest.mark.parametrize("name", list(GEOMS))
def test_split_weight_gradient(name, mt, blocks):
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
    assert list(out) == [144, 128, 1, 1]
    ops.set_conv_tile(xs, (k, xs[1]) + ks, st, pd, 0, (0, 2, 2, 2))
    _lib.check(lib.cstp_conv3d_query_tile(ctypes.byref(desc), 0, out), "query")
    assert list(out) == [128, 64, 0, 2]
    with pytest.raises(_lib.CstpError):
        ops.set_conv_tile(xs, (k, xs[1]) + ks, st, pd, 0, (1, 7, 0, 0))        # no 112-row split tile
    with pytest.raises(_lib.CstpError):
        ops.set_conv_tile(xs, (k, xs[1]) + ks, st, pd, 2, (1, 5, 8, 0))
