"""igemm_k1t (csrc/igemm_tpatch.h): the stride-1 3x1x1 temporal convolutions with the input patch resident in LDS (tiles of
8 frames x 28 columns; forward and data gradient; BatchNorm statistics of the output from the epilogue; the BatchNorm + ReLU in
front applied once per staged element).  Parity against PyTorch CPU fp64 (1e-4 bar of BASELINE.json), against the gather kernel
igemm_k1s on the same inputs, and -- for the fused input transform -- bit for bit against the materialising path."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from test_split_gpu import GEOMS, _run

pytestmark = pytest.mark.gpu

TPATCH_GEOMS = {
    "T8": ((2, 144, 8, 14, 14), 64),          # one tile of frames per clip: both halo frames are the zero padding
    "T16": ((1, 64, 16, 14, 28), 144),        # two frame tiles: the halo between them is real data; 144 rows
    "ragged blocks": ((2, 48, 8, 14, 14), 80),  # half-empty second channel block; 80 rows in 64- / 128- / 144-row blocks
    "wide": ((1, 32, 8, 28, 28), 32),         # 28 column tiles per frame row
}


@pytest.mark.parametrize("mt_f,mt_d", [(4, 9), (8, 4), (9, 8)])
@pytest.mark.parametrize("name", list(TPATCH_GEOMS))
def test_temporal_patch_kernel_forward_and_data_gradient(name, mt_f, mt_d):
    from cstp_amd import _lib, ops
    xs, k = TPATCH_GEOMS[name]
    GEOMS["_tpatch"] = (xs, k, (3, 1, 1), (1, 1, 1), (1, 0, 0))
    ops.set_split_terms(2)
    try:
        _run("_tpatch", {0: (2, mt_f, 0, 0), 1: (2, mt_d, 0, 0), 2: (1, 4, 8, 0)})
        out = (ctypes.c_int32 * 4)()
        desc = ops._desc(xs, (k, xs[1], 3, 1, 1), (1, 1, 1), (1, 0, 0))
        _lib.check(_lib.load().cstp_conv3d_query_tile(ctypes.byref(desc), 0, out), "query")
        assert list(out)[:3] == [16 * mt_f, 224, 2]          # the pinned patch tile is what runs
        _lib.check(_lib.load().cstp_conv3d_query_tile(ctypes.byref(desc), 1, out), "query")
        assert list(out)[:3] == [16 * mt_d, 224, 2]
    finally:
        ops.set_split_terms(0)
        del GEOMS["_tpatch"]


def test_temporal_patch_tile_is_refused_where_the_kernel_does_not_apply():
    from cstp_amd import _lib, ops
    lib = _lib.load()
    out = (ctypes.c_int32 * 4)()
    ops.set_split_terms(2)
    try:
        for xs, k, st, why in (((2, 144, 4, 14, 14), 64, (1, 1, 1), "4 frames"), ((2, 144, 8, 7, 7), 64, (1, 1, 1), "49 columns"),
                               ((2, 40, 8, 14, 14), 64, (1, 1, 1), "40 channels"), ((2, 144, 8, 14, 14), 64, (2, 1, 1), "strided")):
            ws = (k, xs[1], 3, 1, 1)
            ops.set_conv_tile(xs, ws, st, (1, 0, 0), 0, (2, 4, 0, 0))
            desc = ops._desc(xs, ws, st, (1, 0, 0))
            _lib.check(lib.cstp_conv3d_query_tile(ctypes.byref(desc), 0, out), "query")
            assert out[1] != 224, why
    finally:
        ops.set_split_terms(0)


@pytest.mark.parametrize("groups", [1, 2])
@pytest.mark.parametrize("mt", [4, 9])
def test_temporal_patch_kernel_leaves_batchnorm_statistics(mt, groups):
    """conv3d(.., bn_groups) on a temporal layer -> batch_norm_act takes sums and range from the convolution's epilogue."""
    from cstp_amd import ops
    xs, k = (4, 48, 8, 14, 14), (64 if mt == 4 else 144)
    ws = (k, xs[1], 3, 1, 1)
    ops.set_split_terms(2)
    ops.set_conv_tile(xs, ws, (1, 1, 1), (1, 0, 0), 0, (2, mt, 0, 0))
    try:
        g = torch.Generator().manual_seed(5)
        x = (torch.randn(xs, generator=g) + 0.3).cuda()
        w = (torch.randn(ws, generator=g) * 0.1).cuda()
        gamma, beta = (torch.rand(k, generator=g) + 0.5).cuda(), torch.randn(k, generator=g).cuda()

        def run(fused):
            rm, rv = torch.zeros(k, device="cuda"), torch.ones(k, device="cuda")
            y = ops.conv3d(x, w, None, 1, (1, 0, 0), bn_groups=groups if fused else 0, bn_pivot=rm)
            assert (getattr(y, "_cstp_bnstats", None) is not None) == fused
            return ops.batch_norm_act(y, gamma, beta, rm, rv, None, True, 1e-5, 0.1, groups), rm, rv

        a, rma, rva = run(True)
        b, rmb, rvb = run(False)
        assert rel_err(a, b) < 2e-6 and rel_err(rma, rmb) < 2e-6 and rel_err(rva, rvb) < 2e-6
        yc = F.conv3d(x.double().cpu(), w.double().cpu(), None, 1, (1, 0, 0))
        outs, rm, rv = [], torch.zeros(k, dtype=torch.float64), torch.ones(k, dtype=torch.float64)
        for part in yc.chunk(groups, 0):
            outs.append(F.relu(F.batch_norm(part, rm, rv, gamma.double().cpu(), beta.double().cpu(), True, 0.1, 1e-5)))
        assert rel_err(a.cpu().double(), torch.cat(outs, 0)) < 1e-4
        assert rel_err(rma.cpu().double(), rm) < 1e-4 and rel_err(rva.cpu().double(), rv) < 1e-4
    finally:
        ops.set_split_terms(0)


def _bulk_err(a, b):
    """Error of all but the worst 2 % of the elements, relative to the largest reference magnitude.  Gradients that pass through
    ReLUs are compared this way: an element whose pre-activation is within rounding of zero takes either side of the ReLU
    depending on the last bit of the BatchNorm statistics (any two correct implementations disagree on a handful of such
    elements per million), and ONE flipped mask element moves a few thousand input-gradient elements of a block by ~1e-3."""
    d = (a.double().cpu() - b.double().cpu()).abs().flatten()
    k = max(1, int(d.numel() * 0.98))
    return float(d.kthvalue(k).values) / max(float(b.abs().max()), 1e-30)


@pytest.mark.parametrize("groups", [1, 2])
def test_residual_block_fused_on_the_patch_kernels_matches_the_materialising_block(groups):
    """A whole SpatioTemporalResBlock (train mode): spatial convolutions on igemm_k1p, temporal ones on igemm_k1t with the
    BatchNorm in front applied in their staging and the BatchNorm behind fed from their epilogue -- against the same block with
    every BatchNorm output materialised and every statistic from its own pass, and against PyTorch fp64."""
    from cstp_amd import ops, r21d_byol as rb
    ops.set_split_terms(2)
    torch.manual_seed(7)
    blk = rb.SpatioTemporalResBlock(64, 64, 3).cuda().train()
    x = torch.randn(2 * groups, 64, 8, 28, 28, device="cuda")
    ys, ts = (2 * groups, 144, 8, 28, 28), (64, 144, 3, 1, 1)
    ops.set_conv_tile(ys, ts, (1, 1, 1), (1, 0, 0), 0, (2, 4, 0, 0))
    ops.set_conv_tile(ys, ts, (1, 1, 1), (1, 0, 0), 1, (2, 9, 0, 0))
    ops.set_conv_tile(x.shape, (144, 64, 1, 3, 3), (1, 1, 1), (0, 1, 1), 0, (2, 9, 0, 0))
    state = {k: v.clone() for k, v in blk.state_dict().items()}
    res = []
    try:
        for fuse_t, fuse_stats in ((True, True), (False, False)):
            rb.FUSE_BN_TEMPORAL, ops.FUSE_BN_STATS = fuse_t, fuse_stats
            blk.load_state_dict(state)
            blk.zero_grad(set_to_none=True)
            xg = x.clone().requires_grad_(True)
            y = blk(xg, groups)
            y.square().mean().backward()
            ops._join_side_streams()
            res.append((y.detach().clone(), xg.grad.clone(), {n: p.grad.clone() for n, p in blk.named_parameters()},
                        {n: b.clone() for n, b in blk.named_buffers()}))
        assert ops.in_affine_fused(ys, ts, 1, (1, 0, 0), groups)
        a, b = res
        assert rel_err(a[0], b[0]) < 1e-5 and _bulk_err(a[1], b[1]) < 1e-5
        for n in a[2]:      # (a weight-gradient entry sums over all positions, the ones a flipped element reaches included)
            assert _bulk_err(a[2][n], b[2][n]) < 5e-4, n
        for n in a[3]:
            if a[3][n].dtype.is_floating_point:
                assert rel_err(a[3][n], b[3][n]) < 1e-5, n
        # fp64 reference of the same block
        import torch.nn as nn

        def ref_stc(cin, cout, pre):
            mid = blk.conv1.spatial_conv.weight.shape[0]
            m = nn.Sequential(nn.Conv3d(cin, mid, (1, 3, 3), 1, (0, 1, 1), bias=False), nn.BatchNorm3d(mid), nn.ReLU(),
                              nn.Conv3d(mid, cout, (3, 1, 1), 1, (1, 0, 0), bias=False)).double()
            m[0].weight.data = state[pre + ".spatial_conv.weight"].double().cpu()
            m[1].weight.data = state[pre + ".bn.weight"].double().cpu(); m[1].bias.data = state[pre + ".bn.bias"].double().cpu()
            m[3].weight.data = state[pre + ".temporal_conv.weight"].double().cpu()
            return m
        c1, c2 = ref_stc(64, 64, "conv1"), ref_stc(64, 64, "conv2")
        bn1, bn2 = nn.BatchNorm3d(64).double(), nn.BatchNorm3d(64).double()
        for bn, pre in ((bn1, "bn1"), (bn2, "bn2")):
            bn.weight.data = state[pre + ".weight"].double().cpu(); bn.bias.data = state[pre + ".bias"].double().cpu()
        xd = x.double().cpu().requires_grad_(True)

        def per_group(f, t):
            return torch.cat([f(p) for p in t.chunk(groups, 0)], 0)
        h = per_group(c1[0], xd); h = torch.relu(per_group(c1[1], h)); h = per_group(c1[3], h)
        h = torch.relu(per_group(bn1, h))
        h = per_group(c2[0], h); h = torch.relu(per_group(c2[1], h)); h = per_group(c2[3], h)
        out = torch.relu(xd + per_group(bn2, h))
        out.square().mean().backward()
        assert rel_err(a[0], out.detach()) < 1e-4 and _bulk_err(a[1], xd.grad) < 1e-4
        assert rel_err(a[2]["conv2.temporal_conv.weight"], c2[3].weight.grad) < 1e-4      # (behind the last ReLU: no mask in its path)
        assert _bulk_err(a[2]["conv1.spatial_conv.weight"], c1[0].weight.grad) < 5e-4
    finally:
        rb.FUSE_BN_TEMPORAL, ops.FUSE_BN_STATS = True, True
        ops.set_split_terms(0)


# ---- igemm_k1w (csrc/igemm_twres.h): the 64-row temporal forward layers with the packed weights resident in LDS (tile (2, 4, 2, .))
K1W_GEOMS = {
    "conv2_x class": ((2, 144, 16, 28, 28), 64),      # 15 K-tiles, half-empty fifth channel block, two frame tiles per clip
    "one frame tile": ((2, 64, 8, 14, 14), 64),       # 6 K-tiles; both halo frames are the zero padding
    "48 rows": ((1, 96, 8, 14, 28), 48),              # fewer valid rows than the 64-row block
    "many tiles": ((4, 32, 16, 14, 14), 64),          # more items than blocks on small grids: the item loop runs
}


@pytest.mark.parametrize("name", list(K1W_GEOMS))
def test_weight_resident_temporal_forward_against_fp64_and_the_ring_kernel(name):
    from cstp_amd import ops
    xs, k = K1W_GEOMS[name]
    ws = (k, xs[1], 3, 1, 1)
    ops.set_split_terms(2)
    try:
        g = torch.Generator().manual_seed(21)
        x = (torch.randn(xs, generator=g) + 0.1).cuda()
        w = (torch.randn(ws, generator=g) * 0.1).cuda()
        ops.set_conv_tile(xs, ws, (1, 1, 1), (1, 0, 0), 0, (2, 4, 2, 0))
        y = ops.conv3d(x, w, None, 1, (1, 0, 0))
        ops.set_conv_tile(xs, ws, (1, 1, 1), (1, 0, 0), 0, (2, 4, 0, 0))
        y_ring = ops.conv3d(x, w, None, 1, (1, 0, 0))
        ref = F.conv3d(x.double().cpu(), w.double().cpu(), None, 1, (1, 0, 0))
        assert rel_err(y.cpu().double(), ref) < 1e-4
        assert torch.equal(y, y_ring)         # the same products in the same order: bit for bit igemm_k1t's result
    finally:
        ops.set_split_terms(0)


@pytest.mark.parametrize("groups", [1, 2])
def test_weight_resident_temporal_forward_leaves_batchnorm_statistics(groups):
    from cstp_amd import ops
    xs, k = (4, 48, 8, 14, 14), 64
    ws = (k, xs[1], 3, 1, 1)
    ops.set_split_terms(2)
    ops.set_conv_tile(xs, ws, (1, 1, 1), (1, 0, 0), 0, (2, 4, 2, 0))
    try:
        g = torch.Generator().manual_seed(5)
        x = (torch.randn(xs, generator=g) + 0.3).cuda()
        w = (torch.randn(ws, generator=g) * 0.1).cuda()
        gamma, beta = (torch.rand(k, generator=g) + 0.5).cuda(), torch.randn(k, generator=g).cuda()

        def run(fused):
            rm, rv = torch.zeros(k, device="cuda"), torch.ones(k, device="cuda")
            y = ops.conv3d(x, w, None, 1, (1, 0, 0), bn_groups=groups if fused else 0, bn_pivot=rm)
            assert (getattr(y, "_cstp_bnstats", None) is not None) == fused
            return ops.batch_norm_act(y, gamma, beta, rm, rv, None, True, 1e-5, 0.1, groups), rm, rv

        a, rma, rva = run(True)
        b, rmb, rvb = run(False)
        assert rel_err(a, b) < 2e-6 and rel_err(rma, rmb) < 2e-6 and rel_err(rva, rvb) < 2e-6
        yc = F.conv3d(x.double().cpu(), w.double().cpu(), None, 1, (1, 0, 0))
        outs, rm, rv = [], torch.zeros(k, dtype=torch.float64), torch.ones(k, dtype=torch.float64)
        for part in yc.chunk(groups, 0):
            outs.append(F.relu(F.batch_norm(part, rm, rv, gamma.double().cpu(), beta.double().cpu(), True, 0.1, 1e-5)))
        assert rel_err(a.cpu().double(), torch.cat(outs, 0)) < 1e-4
        assert rel_err(rma.cpu().double(), rm) < 1e-4 and rel_err(rva.cpu().double(), rv) < 1e-4
    finally:
        ops.set_split_terms(0)


@pytest.mark.parametrize("groups", [1, 2])
def test_weight_resident_temporal_forward_applies_the_batchnorm_in_front_bit_for_bit(groups):
    """temporal_conv(relu(bn(spatial_conv(x)))) with the BatchNorm + ReLU applied in igemm_k1w's staging: the forward output equals
    the materialising path's bit for bit (tests/test_fused_bn_gpu.py's chain, the temporal forward pinned on the new kernel)."""
    import test_fused_bn_gpu as fb
    from cstp_amd import ops
    name = "temporal patch kernel, 64 rows"
    old = fb.FUSED_GEOMS[name]
    fb.FUSED_GEOMS[name] = old[:4] + ((2, 4, 2, 0),) + old[5:]
    ops.set_split_terms(2)
    try:
        ys = fb._pin(name)
        xs, mid, k = fb.FUSED_GEOMS[name][:3]
        assert ops.in_affine_fused(ys, (k, mid, 3, 1, 1), 1, (1, 0, 0), groups)
        ops.set_deterministic(True)
        a, b = fb._run(name, groups, True), fb._run(name, groups, False)
        assert a["zmax"] == b["zmax"] and a["zmax"] != 0
        for key in ("out", "dwt", "dx", "dws", "dgamma", "dbeta", "rm", "rv"):
            assert torch.equal(a[key], b[key]), (key, float((a[key] - b[key]).abs().max()))
    finally:
        ops.set_deterministic(False)
        fb.FUSED_GEOMS[name] = old
        ops.set_split_terms(0)


def test_weight_resident_tile_falls_back_to_the_ring_kernel_where_the_weights_do_not_fit():
    """(2, 4, 2, .) on a layer with more than 15 K-tiles (or in the data gradient) runs igemm_k1t: same results, no error."""
    from cstp_amd import ops
    xs, k = (1, 192, 8, 14, 14), 64              # 6 channel blocks = 18 K-tiles
    ws = (k, xs[1], 3, 1, 1)
    ops.set_split_terms(2)
    try:
        g = torch.Generator().manual_seed(3)
        x = torch.randn(xs, generator=g).cuda()
        w = (torch.randn(ws, generator=g) * 0.1).cuda()
        ops.set_conv_tile(xs, ws, (1, 1, 1), (1, 0, 0), 0, (2, 4, 2, 0))
        y = ops.conv3d(x, w, None, 1, (1, 0, 0))
        assert rel_err(y.cpu().double(), F.conv3d(x.double().cpu(), w.double().cpu(), None, 1, (1, 0, 0))) < 1e-4
    finally:
        ops.set_split_terms(0)


# ---- igemm_k2t (csrc/igemm_wtpatch.h): the temporal layers' weight gradient on streams of 32-position chunks (mode-2 tile (2, 9, ., .))
K2T_GEOMS = {
    "T1 class": ((2, 144, 8, 8, 16), 64),          # exactly one (144, 64) pair; 4 chunks per frame
    "two column blocks": ((2, 48, 4, 8, 8), 80),   # 80 dY channels = 64 + 16; 48 x channels of a 144-row block
    "two row blocks": ((1, 160, 4, 8, 4), 32),     # 160 x channels = 144 + 16; one chunk per frame; 32 dY channels
    "long clip": ((1, 32, 16, 4, 8), 64),          # 16 frames, one chunk
    "many items": ((6, 64, 2, 16, 16), 48),        # 48 items of three stream frames each: blocks walk item ranges
    "stem class": ((2, 83, 4, 8, 8), 64),          # 83 x channels (the stem's temporal layer): a ragged last 4-channel piece
    # frame sizes that are multiples of 16 only: chunks of 16 positions, a K-step = two frames, K-steps straddle items
    "16-position chunks": ((2, 48, 8, 4, 12), 64),
    "conv3_x class": ((2, 288, 8, 28, 28), 128),   # 784 positions per frame: two row blocks x two column blocks
}


@pytest.mark.parametrize("name", list(K2T_GEOMS))
def test_temporal_stream_weight_gradient_against_fp64(name):
    from cstp_amd import _lib, ops
    xs, k = K2T_GEOMS[name]
    GEOMS["_k2t"] = (xs, k, (3, 1, 1), (1, 1, 1), (1, 0, 0))
    ops.set_split_terms(2)
    try:
        _run("_k2t", {0: (1, 4, 0, 0), 1: (1, 4, 0, 0), 2: (2, 9, 1, 0)})
        out = (ctypes.c_int32 * 4)()
        desc = ops._desc(xs, (k, xs[1], 3, 1, 1), (1, 1, 1), (1, 0, 0))
        _lib.check(_lib.load().cstp_conv3d_query_tile(ctypes.byref(desc), 2, out), "query")
        assert list(out)[:3] == [144, 192, 2]              # the stream kernel is what ran
    finally:
        ops.set_split_terms(0)
        del GEOMS["_k2t"]


def test_temporal_stream_weight_gradient_deterministic_mode_is_reproducible():
    from cstp_amd import ops
    xs, k = K2T_GEOMS["many items"]
    ws = (k, xs[1], 3, 1, 1)
    ops.set_split_terms(2)
    ops.set_deterministic(True)
    ops.set_conv_tile(xs, ws, (1, 1, 1), (1, 0, 0), 2, (2, 9, 1, 0))
    try:
        g = torch.Generator().manual_seed(9)
        x = torch.randn(xs, generator=g).cuda()
        dy = torch.randn((xs[0], k) + xs[2:], generator=g).cuda()
        res = []
        for _ in range(2):
            w = torch.zeros(ws, device="cuda", requires_grad=True)
            ops.conv3d(x, w, None, 1, (1, 0, 0)).backward(dy)
            ops._join_side_streams()
            res.append(w.grad.clone())
        assert torch.equal(res[0], res[1])
        wd = torch.zeros(ws, dtype=torch.float64, requires_grad=True)
        F.conv3d(x.double().cpu(), wd, None, 1, (1, 0, 0)).backward(dy.double().cpu())
        assert rel_err(res[0].cpu().double(), wd.grad) < 1e-4
    finally:
        ops.set_deterministic(False)
        ops.set_split_terms(0)


@pytest.mark.parametrize("groups", [1, 2])
def test_temporal_stream_weight_gradient_with_the_batchnorm_in_front_inside(groups):
    """The fused chain of tests/test_fused_bn_gpu.py with the temporal weight gradient on igemm_k2t<AFF>: bit for bit the
    materialising path's (which runs igemm_k2t on the written tensor) in deterministic mode, fp64 within the 1e-4 bar."""
    import test_fused_bn_gpu as fb
    from cstp_amd import ops
    fb.FUSED_GEOMS["_k2t"] = ((4, 16, 7, 16, 16), 144, 64, 9, (1, 4, 0, 0), (2, 9, 1, 0))      # 7 frames x 256 positions: whole 224-position tiles per group AND whole 32-position chunks
    ops.set_split_terms(2)
    ops.set_deterministic(True)
    try:
        ys = fb._pin("_k2t")
        xs, mid, k = fb.FUSED_GEOMS["_k2t"][:3]
        assert ops.in_affine_fused(ys, (k, mid, 3, 1, 1), 1, (1, 0, 0), groups)
        a, b = fb._run("_k2t", groups, True), fb._run("_k2t", groups, False)
        assert a["zmax"] == b["zmax"] and a["zmax"] != 0
        for key in ("out", "dwt", "dx", "dws", "dgamma", "dbeta", "rm", "rv"):
            assert torch.equal(a[key], b[key]), (key, float((a[key] - b[key]).abs().max()))
        x, w_s, w_t, gamma, beta = [t.double().requires_grad_(True) for t in fb._inputs("_k2t")]
        y = F.conv3d(x, w_s, None, 1, (0, 1, 1))
        rm, rv = torch.zeros(gamma.numel(), dtype=torch.float64), torch.ones(gamma.numel(), dtype=torch.float64)
        z = torch.cat([F.relu(F.batch_norm(p, rm, rv, gamma, beta, True, 0.1, 1e-5)) for p in y.chunk(groups, 0)], 0)
        out = F.conv3d(z, w_t, None, 1, (1, 0, 0))
        dy = (torch.rand(out.shape, generator=torch.Generator().manual_seed(3)) * 2 - 1).double()
        out.backward(dy)
        assert rel_err(a["dwt"], w_t.grad) < 1e-4 and rel_err(a["out"], out.detach()) < 1e-4
    finally:
        ops.set_deterministic(False)
        fb.FUSED_GEOMS.pop("_k2t", None)
        ops.set_split_terms(0)
