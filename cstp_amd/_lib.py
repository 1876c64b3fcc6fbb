"""ctypes binding of libcstp_hip.so (the C ABI declared in include/cstp_hip.h).

The product path has no CPU fallback: if the HIP library is missing or fails to load this
module raises, loudly, at first use.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_uint32, POINTER, c_char_p, c_double, c_float, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# CSTP_LIB_PATH: developer override for A/B-ing kernel builds (tools/ab_*.sh); unset in production
LIB_PATH = os.environ.get("CSTP_LIB_PATH") or os.path.join(_HERE, "lib", "libcstp_hip.so")
ABI_VERSION = 18


class ConvDesc(ctypes.Structure):
    """struct cstp_conv_desc (include/cstp_hip.h)."""
    _fields_ = [(n, c_int32) for n in
                ("n", "c", "d", "h", "w", "k", "kt", "kh", "kw", "st", "sh", "sw", "pt", "ph", "pw")]


class InAffine(ctypes.Structure):
    """struct cstp_in_affine: BN(+ReLU) folded into a convolution's gather."""
    _fields_ = [("scale_shift", c_void_p), ("groups", c_int32), ("relu", c_int32)]


class PackRec(ctypes.Structure):
    """struct cstp_pack_rec: one recorded weight-pack launch (cstp_pack_mode / cstp_pack_recorded / cstp_pack_replay)."""
    _fields_ = [("kind", c_int32), ("nblocks", c_int32), ("w", c_void_p), ("dst", c_void_p), ("inv_a", c_void_p),
                ("cells", c_void_p), ("a", c_int32 * 10)]


class CstpError(RuntimeError):
    pass


_P = c_void_p  # device pointers travel as integers (tensor.data_ptr())

# name -> (restype, argtypes).  Must list EVERY symbol include/cstp_hip.h declares
# (tests/test_abi.py parses the header and checks this table and the .so against it).
SIGNATURES = {
    "cstp_abi_version": (c_int32, []),
    "cstp_last_error": (c_char_p, []),
    "cstp_conv3d_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "cstp_conv3d_forward": (c_int32, [_P, POINTER(ConvDesc), _P, _P, _P, POINTER(InAffine), _P, _P, c_size_t]),
    "cstp_conv3d_backward_data": (c_int32, [_P, POINTER(ConvDesc), _P, _P, _P, _P, c_size_t]),
    "cstp_conv3d_backward_weight": (c_int32, [_P, POINTER(ConvDesc), _P, POINTER(InAffine), _P, _P, _P, c_size_t]),
    "cstp_conv3d_forward_am": (c_int32, [_P, POINTER(ConvDesc), _P, _P, _P, POINTER(InAffine), _P, _P, c_size_t, _P]),
    "cstp_conv3d_backward_data_am": (c_int32, [_P, POINTER(ConvDesc), _P, _P, _P, _P, c_size_t, _P]),
    "cstp_conv3d_backward_data_acc": (c_int32, [_P, POINTER(ConvDesc), _P, _P, _P, _P, c_size_t, _P, c_int32]),
    "cstp_conv3d_backward_weight_am": (c_int32, [_P, POINTER(ConvDesc), _P, POINTER(InAffine), _P, _P, _P, c_size_t, _P, _P]),
    "cstp_conv3d_backward_weight_acc": (c_int32, [_P, POINTER(ConvDesc), _P, POINTER(InAffine), _P, _P, _P, c_size_t, _P, _P,
                                                 c_int32]),
    "cstp_pack_mode": (c_int32, [c_int32]),
    "cstp_pack_recorded": (c_int32, [POINTER(PackRec), c_int32]),
    "cstp_pack_replay": (c_int32, [_P, _P, _P, c_int32, c_int32]),
    "cstp_pack_register": (c_int32, [POINTER(c_void_p), c_int32, c_int32]),
    "cstp_set_deterministic": (c_int32, [c_int32]),
    "cstp_get_deterministic": (c_int32, []),
    "cstp_gemm_set_split_terms": (c_int32, [c_int32]),
    "cstp_gemm_get_split_terms": (c_int32, []),
    "cstp_conv3d_query_tile": (c_int32, [POINTER(ConvDesc), c_int32, POINTER(c_int32)]),
    "cstp_conv3d_set_tile": (c_int32, [POINTER(ConvDesc), c_int32, POINTER(c_int32)]),
    "cstp_conv3d_get_tile": (c_int32, [POINTER(ConvDesc), c_int32, POINTER(c_int32)]),
    "cstp_conv3d_autotune": (c_int32, [_P, POINTER(ConvDesc), c_int32, _P, _P, _P, _P, c_size_t, c_int32]),
    "cstp_bn_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32, c_int32]),
    "cstp_bn_forward_train": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32,
                                        c_float, c_float, c_int32, _P, c_size_t]),
    "cstp_bn_forward_train_am": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32,
                                           c_float, c_float, c_int32, _P, c_size_t, _P]),
    "cstp_bn_forward_train_pre": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32,
                                            c_float, c_float, c_int32, _P, c_size_t, _P, _P, c_int32]),
    "cstp_conv3d_bnstats_nsplit": (c_int32, [POINTER(ConvDesc), c_int32]),
    "cstp_conv3d_bnstats_nsplit_aff": (c_int32, [POINTER(ConvDesc), c_int32]),
    "cstp_conv3d_forward_bnstats": (c_int32, [_P, POINTER(ConvDesc), _P, _P, _P, _P, c_size_t, _P, c_int32, _P, _P, c_size_t,
                                              POINTER(c_int32), _P, _P]),
    "cstp_conv3d_in_affine_fused": (c_int32, [POINTER(ConvDesc), c_int32]),
    "cstp_bn_finalize_pre": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_float, c_float,
                                       c_int32, _P, c_int32, _P]),
    "cstp_bn_backward_am": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32,
                                      c_int32, _P, c_size_t, _P, c_int32]),
    "cstp_bn_stats_train": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_float,
                                      c_float, _P, c_size_t]),
    "cstp_bn_backward": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32,
                                   c_int32, _P, c_size_t]),
    "cstp_bn_eval_workspace_bytes": (c_size_t, [c_int32]),
    "cstp_bn_forward_eval": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_float, c_int32, _P,
                                       c_size_t]),
    "cstp_bn_forward_eval_am": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_float, c_int32, _P,
                                          c_size_t, _P]),
    "cstp_avgpool_forward": (c_int32, [_P, _P, _P, c_int32, c_int32]),
    "cstp_avgpool_backward": (c_int32, [_P, _P, _P, c_int32, c_int32]),
    "cstp_maxpool3d_forward": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, POINTER(c_int32), POINTER(c_int32),
                                         POINTER(c_int32)]),
    "cstp_maxpool3d_backward": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, POINTER(c_int32), POINTER(c_int32),
                                          POINTER(c_int32)]),
    "cstp_channel_sum": (c_int32, [_P, _P, _P, c_int32, c_int32, c_int32, _P, c_size_t]),
    "cstp_byol_loss_forward": (c_int32, [_P, _P, _P, _P, c_int32, c_int32]),
    "cstp_byol_loss_backward": (c_int32, [_P, _P, _P, _P, _P, c_int32, c_int32]),
    "cstp_l2_normalize_forward": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_float]),
    "cstp_l2_normalize_backward": (c_int32, [_P, _P, _P, _P, _P, c_int32, c_int32, c_float]),
    "cstp_cross_entropy_forward": (c_int32, [_P, _P, _P, _P, c_int32, c_int32]),
    "cstp_cross_entropy_backward": (c_int32, [_P, _P, _P, _P, _P, c_int32, c_int32]),
    "cstp_ntxent_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "cstp_ntxent_forward": (c_int32, [_P, _P, _P, c_int32, c_int32, c_float, _P, c_size_t]),
    "cstp_ntxent_backward": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_float, _P, c_size_t]),
    "cstp_clip_assemble": (c_int32, [_P, _P, c_int32, c_int32, c_int32, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P,
                                     c_int32, _P, _P, c_int32, c_int32, c_int32, _P, _P]),
    "cstp_clip_assemble_u8": (c_int32, [_P, _P, c_int32, c_int32, c_int32, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P,
                                        c_int32, _P, _P, c_int32, c_int32, c_int32, _P, _P]),
    "cstp_clip_rotate": (c_int32, [_P, _P, _P, c_int32, c_int32, c_int32, _P]),
    "cstp_clip_blend": (c_int32, [_P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_float, _P]),
    "cstp_clip_hue": (c_int32, [_P, _P, _P, c_size_t, c_int32, c_int32]),
    "cstp_clip_gray": (c_int32, [_P, _P, _P, c_int32, c_int32, c_int32, _P]),
    "cstp_clip_box_blur": (c_int32, [_P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_uint32, c_uint32, c_int32]),
    "cstp_clip_finish": (c_int32, [_P, _P, _P, c_int32, c_int32, c_int32, c_int32]),
    "cstp_ema_update": (c_int32, [_P, _P, _P, c_size_t, c_double]),
    "cstp_sumsq": (c_int32, [_P, _P, c_size_t, _P, _P, c_size_t]),
    "cstp_clip_coef": (c_int32, [_P, _P, c_float, _P, _P]),
    "cstp_sgd_step": (c_int32, [_P, _P, _P, _P, c_size_t, _P, c_float, c_float, _P, c_int32, c_int32]),
    "cstp_adam_step": (c_int32, [_P, _P, _P, _P, _P, c_size_t, _P, c_float, c_float, c_float, c_float, c_int32, c_int32]),
    # the bf16-storage path (csrc/b16.hip)
    "cstp_b16_cast": (c_int32, [_P, _P, _P, c_size_t]),
    "cstp_b16_conv3d_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "cstp_b16_conv3d_forward": (c_int32, [_P, POINTER(ConvDesc), _P, _P, _P, _P, c_size_t]),
    "cstp_b16_conv3d_backward_data": (c_int32, [_P, POINTER(ConvDesc), _P, _P, _P, _P, c_size_t]),
    "cstp_b16_conv3d_backward_data_acc": (c_int32, [_P, POINTER(ConvDesc), _P, _P, _P, _P, c_size_t, c_int32]),
    "cstp_b16_conv3d_backward_weight": (c_int32, [_P, POINTER(ConvDesc), _P, _P, _P, _P, c_size_t, c_int32]),
    "cstp_b16_bn_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32, c_int32]),
    "cstp_b16_bn_forward_train": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32,
                                            c_float, c_float, c_int32, _P, c_size_t]),
    "cstp_b16_bn_backward": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32,
                                       c_int32, _P, c_size_t, c_int32]),
    "cstp_b16_bn_forward_eval": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_float, c_int32]),
    "cstp_b16_maxpool3d_forward": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, POINTER(c_int32),
                                             POINTER(c_int32), POINTER(c_int32)]),
    "cstp_b16_maxpool3d_backward": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, POINTER(c_int32),
                                              POINTER(c_int32), POINTER(c_int32)]),
    "cstp_b16_avgpool_forward": (c_int32, [_P, _P, _P, c_int32, c_int32]),
    "cstp_b16_avgpool_backward": (c_int32, [_P, _P, _P, c_int32, c_int32]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load libcstp_hip.so (once).  Raises CstpError when it is absent -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise CstpError(
            "libcstp_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  cstp_amd has no CPU fallback." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # missing ROCm runtime etc.
        raise CstpError("failed to load %s: %s" % (LIB_PATH, e)) from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise CstpError("libcstp_hip.so lacks symbol %s (stale build?)" % name) from e
        fn.restype = res
        fn.argtypes = args
    ver = lib.cstp_abi_version()
    if ver != ABI_VERSION:
        raise CstpError("libcstp_hip.so ABI version %d != binding %d (rebuild)" % (ver, ABI_VERSION))
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().cstp_last_error()
        raise CstpError("%s failed: %s" % (what, msg.decode() if msg else "unknown error"))
