"""The C-ABI library loads (no GPU needed) and exports every symbol include/cstp_hip.h declares;
the ctypes binding table covers exactly that set.  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cstp_hip.h")


def header_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cstp_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = header_symbols()
    assert "cstp_conv3d_forward" in syms and "cstp_bn_forward_train" in syms and "cstp_sgd_step" in syms
    assert len(syms) >= 20


def test_binding_table_matches_header():
    from cstp_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_symbols()


def test_library_exports_every_declared_symbol():
    from cstp_amd import _lib
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__ as entry
        entry.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), "libcstp_hip.so lacks " + name
    loaded = _lib.load()
    assert loaded.cstp_abi_version() == _lib.ABI_VERSION


def test_argument_validation_without_gpu():
    """Pure host-side checks of the ABI run before any HIP call."""
    from cstp_amd import _lib
    lib = _lib.load()
    bad = _lib.ConvDesc(0, 3, 4, 8, 8, 8, 1, 3, 3, 1, 1, 1, 0, 1, 1)
    assert lib.cstp_conv3d_workspace_bytes(ctypes.byref(bad)) == 0
    ok = _lib.ConvDesc(2, 64, 4, 14, 14, 144, 1, 3, 3, 1, 1, 1, 0, 1, 1)
    assert lib.cstp_conv3d_workspace_bytes(ctypes.byref(ok)) >= 9 * 64 * 144 * 4
    rc = lib.cstp_conv3d_forward(None, ctypes.byref(ok), None, None, None, None, None, None, 0)
    assert rc != 0 and b"null argument" in lib.cstp_last_error()
    assert lib.cstp_bn_workspace_bytes(16, 144, 50176, 1) > 0 and lib.cstp_bn_workspace_bytes(15, 144, 50176, 2) == 0
    assert lib.cstp_ntxent_workspace_bytes(32, 512) >= (2 * 32 + 2 * 32 * 32) * 4


def test_every_bn_entry_point_fails_cleanly_on_its_early_return_path():
    """The precondition checks of the BatchNorm entry points (train, train with by-products, precomputed statistics, statistics only,
    eval, backward) return an error code and a readable message before any HIP call -- the path on which round 1's first GPU run
    once died inside the error formatting.  Null pointers, bad shapes, a missing workspace: no GPU needed."""
    from cstp_amd import _lib
    lib = _lib.load()
    one = ctypes.c_void_p(8)          # any non-null address: never dereferenced before the checks fail
    f = ctypes.c_float
    calls = [
        ("cstp_bn_forward_train", (None,) * 11 + (4, 8, 16, 1, f(1e-5), f(0.1), 0, None, 0), b"null argument"),
        ("cstp_bn_forward_train", (None, one, None, one, one, one, None, None, one, one, None, 4, 8, 16, 3, f(1e-5), f(0.1), 0, None, 0),
         b"bad shape"),
        ("cstp_bn_forward_train", (None, one, None, one, one, one, None, None, one, one, None, 1, 8, 1, 1, f(1e-5), f(0.1), 0, None, 0),
         b"more than 1 value"),
        ("cstp_bn_forward_train", (None, one, None, one, one, one, None, None, one, one, None, 4, 8, 16, 1, f(1e-5), f(0.1), 0, None, 0),
         b"workspace too small"),
        ("cstp_bn_forward_train_pre", (None, one, None, one, one, one, None, None, one, one, None, 4, 8, 16, 1, f(1e-5), f(0.1), 0, one,
                                       1 << 20, None, None, 0), b"precomputed statistics"),
        ("cstp_bn_stats_train", (None,) * 9 + (4, 8, 16, 1, f(1e-5), f(0.1), None, 0), b"null argument"),
        ("cstp_bn_stats_train", (None, one, one, one, None, None, one, one, one, 4, 8, 1, 1, f(1e-5), f(0.1), one, 1 << 20), b"bad shape"),
        ("cstp_bn_stats_train", (None, one, one, one, one, None, one, one, one, 4, 8, 16, 1, f(1e-5), f(0.1), one, 1 << 20),
         b"running stats must come as a pair"),
    ]
    for name, args, needle in calls:
        rc = getattr(lib, name)(*args)
        msg = lib.cstp_last_error()
        assert rc != 0 and needle in msg, (name, rc, msg)
        assert b"line" in msg                                   # the fixed "<text> (line N)" format


def test_bf16_storage_entry_points_fail_cleanly_before_any_hip_call():
    """The cstp_b16_* entry points (csrc/b16.hip): geometry limits, null pointers and workspace sizes are refused with a message
    before anything is enqueued; workspace queries are pure host functions."""
    from cstp_amd import _lib
    lib = _lib.load()
    one = ctypes.c_void_p(16)
    f = ctypes.c_float
    ok = _lib.ConvDesc(8, 64, 8, 56, 56, 64, 3, 3, 3, 1, 1, 1, 1, 1, 1)            # layer1 conv2 of the 3D-ResNet-50 at 224^2
    stem = _lib.ConvDesc(8, 3, 16, 224, 224, 64, 7, 7, 7, 1, 2, 2, 3, 3, 3)
    small = _lib.ConvDesc(8, 512, 1, 7, 7, 512, 3, 3, 3, 1, 1, 1, 1, 1, 1)        # layer4 conv2: split-K, slabs in the workspace
    odd = _lib.ConvDesc(1, 24, 2, 4, 4, 16, 5, 5, 5, 1, 1, 1, 2, 2, 2)            # 24 channels x 125 taps: not served
    bad = _lib.ConvDesc(0, 64, 8, 56, 56, 64, 3, 3, 3, 1, 1, 1, 1, 1, 1)
    assert lib.cstp_b16_conv3d_workspace_bytes(ctypes.byref(bad)) == 0
    assert lib.cstp_b16_conv3d_workspace_bytes(ctypes.byref(ok)) >= 128 * 27 * 64 * 2
    pad_bytes = 8 * 3 * 22 * 230 * 230 * 2
    assert lib.cstp_b16_conv3d_workspace_bytes(ctypes.byref(stem)) >= pad_bytes + 128 * 1056 * 2
    assert lib.cstp_b16_conv3d_workspace_bytes(ctypes.byref(small)) >= 2 * 8 * 512 * 49 * 4        # at least two fp32 slabs
    calls = [
        ("cstp_b16_conv3d_forward", (None, ctypes.byref(ok), None, None, None, None, 0), b"null argument"),
        ("cstp_b16_conv3d_forward", (None, ctypes.byref(ok), one, one, one, one, 1024), b"workspace too small"),
        ("cstp_b16_conv3d_forward", (None, ctypes.byref(odd), one, one, one, one, 1 << 30), b"offset table"),
        ("cstp_b16_conv3d_forward", (None, ctypes.byref(bad), one, one, one, one, 1 << 30), b"bad convolution geometry"),
        ("cstp_b16_conv3d_backward_data", (None, ctypes.byref(ok), one, one, None, one, 1 << 30), b"null argument"),
        ("cstp_b16_conv3d_backward_data", (None, ctypes.byref(stem), one, one, one, one, 1 << 30), b"at most 27 taps"),
        ("cstp_b16_conv3d_backward_weight", (None, ctypes.byref(ok), one, None, one, one, 1 << 30, 0), b"null argument"),
        ("cstp_b16_bn_forward_train", (None,) * 11 + (4, 8, 16, 1, f(1e-5), f(0.1), 0, None, 0), b"null argument"),
        ("cstp_b16_bn_forward_train", (None, one, None, one, one, one, None, None, one, one, one, 4, 8, 16, 3, f(1e-5), f(0.1), 0, None, 0),
         b"bad shape"),
        ("cstp_b16_bn_forward_train", (None, one, None, one, one, one, None, None, one, one, one, 1, 8, 1, 1, f(1e-5), f(0.1), 0, None, 0),
         b"more than 1 value"),
        ("cstp_b16_bn_forward_train", (None, one, None, one, one, one, None, None, one, one, one, 4, 8, 16, 1, f(1e-5), f(0.1), 0, None, 0),
         b"workspace too small"),
        ("cstp_b16_bn_backward", (None, one, None, one, one, one, one, None, one, None, one, one, 4, 8, 16, 1, 1, one, 1 << 20, 0),
         b"ReLU mask needs y or scale_shift"),
        ("cstp_b16_bn_forward_eval", (None, one, None, one, one, one, one, None, 4, 8, 16, f(1e-5), 0), b"null argument"),
        ("cstp_b16_maxpool3d_forward", (None, one, one, one, 4, 8, 8, 8, (ctypes.c_int32 * 3)(3, 3, 3), (ctypes.c_int32 * 3)(2, 2, 2),
                                        (ctypes.c_int32 * 3)(2, 2, 2)), b"bad pooling geometry"),
        ("cstp_b16_avgpool_forward", (None, None, one, 4, 8), b"bad argument"),
        ("cstp_b16_cast", (None, one, ctypes.c_void_p(6), 16), b"unaligned"),
    ]
    for name, args, needle in calls:
        rc = getattr(lib, name)(*args)
        msg = lib.cstp_last_error()
        assert rc != 0 and needle in msg, (name, rc, msg)
    assert lib.cstp_b16_bn_workspace_bytes(8, 64, 200704, 2) > 0 and lib.cstp_b16_bn_workspace_bytes(7, 64, 200704, 2) == 0


def test_product_path_has_no_cpu_fallback():
    import torch
    from cstp_amd import _lib, ops
    with pytest.raises(_lib.CstpError):
        ops.conv3d(torch.zeros(1, 3, 2, 8, 8), torch.zeros(4, 3, 1, 3, 3), None, 1, (0, 1, 1))
    with pytest.raises(_lib.CstpError):
        ops.batch_norm_act(torch.zeros(4, 8), torch.ones(8), torch.zeros(8))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "cstp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
    for f in ("main_byol.py", "opts.py"):
        src = open(os.path.join(ROOT, f)).read()
        assert "oracle" not in src
