#!/usr/bin/env python3
"""Time the patch kernel alone (cells and packed weights prepared once per call by the library; absmax cell handed in so no
measuring pass runs) -- for diagnostic library variants (CSTP_LIB_PATH).  usage: diag_patch.py [S1|S3|S5] [fwd|fwdbn|dgrad]
(fwdbn: the forward as the training step issues it -- igemm_k1p<MT, true>, BatchNorm sums and range of two view groups from the epilogue)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cstp_amd import ops  # noqa: E402

LAYERS = {"S1": ((32, 64, 16, 56, 56), 144), "S3": ((32, 128, 8, 28, 28), 288), "S5": ((32, 256, 4, 14, 14), 576),
          "T1": ((32, 144, 16, 56, 56), 64), "T3": ((32, 288, 8, 28, 28), 128), "T5": ((32, 576, 4, 14, 14), 256)}
name = sys.argv[1] if len(sys.argv) > 1 else "S1"
mode = 1 if (len(sys.argv) > 2 and sys.argv[2] == "dgrad") else 0
bn = len(sys.argv) > 2 and sys.argv[2] == "fwdbn"
tile = tuple(int(v) for v in sys.argv[3].split(",")) if len(sys.argv) > 3 else ((2, 9, 0, 0) if mode == 0 else (2, 4, 0, 0))
xs, k = LAYERS[name]
lib = ops._lib.load()
x = torch.randn(xs, device="cuda")
temporal = name.startswith("T")
ks, pad = ((3, 1, 1), (1, 0, 0)) if temporal else ((1, 3, 3), (0, 1, 1))
w = torch.randn((k, xs[1]) + ks, device="cuda") * 0.05
desc = ops._desc(xs, tuple(w.shape), (1, 1, 1), pad)
y = torch.empty((xs[0], k) + xs[2:], device="cuda")
dy = torch.randn_like(y)
dx = torch.empty_like(x)
wsb = torch.empty(lib.cstp_conv3d_workspace_bytes(ctypes.byref(desc)), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
ops.set_conv_tile(xs, tuple(w.shape), (1, 1, 1), pad, mode, tile)
cell = (x if mode == 0 else dy).abs().max().view(torch.int32).clone()
if bn:
    ops._tag_absmax(x, cell)
    fn = lambda: ops.conv3d(x, w, None, 1, pad, bn_groups=2)
elif mode == 0:
    fn = lambda: ops.check(lib.cstp_conv3d_forward_am(st, ctypes.byref(desc), x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(),
                                                      wsb.data_ptr(), wsb.numel(), cell.data_ptr()), "fwd")
else:
    fn = lambda: ops.check(lib.cstp_conv3d_backward_data_am(st, ctypes.byref(desc), dy.data_ptr(), w.data_ptr(), dx.data_ptr(),
                                                            wsb.data_ptr(), wsb.numel(), cell.data_ptr()), "dgrad")
for _ in range(3):
    fn()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn()
    b.record()
    b.synchronize()
    ts.append(a.elapsed_time(b) / 10)
gf = 2.0 * xs[0] * xs[2] * xs[3] * xs[4] * k * xs[1] * (3 if temporal else 9) / 1e9
print("%s %s tile %s lib %s: min %.3f ms med %.3f ms  %.1f TF/s (incl. %s pack)" % (
    name, "dgrad" if mode else ("fwdbn" if bn else "fwd"), tile, os.path.basename(os.environ.get("CSTP_LIB_PATH", "default")), min(ts),
    sorted(ts)[2], gf / min(ts), "weight"))

if hasattr(lib, "cstp_debug_stamps"):          # a -DKP_DIAG=16 build: in-kernel stamps of consumer wave 0 of block 0
    import ctypes as C
    buf = (C.c_ulonglong * 8)()
    lib.cstp_debug_stamps(buf)                 # reset
    fn()
    torch.cuda.synchronize()
    lib.cstp_debug_stamps(buf)
    loop, bar, a, epi, kts, items, real = [int(v) for v in buf][:7]
    whole = int(buf[7])
    print("stamps (one launch, wave 0 of block 0): %d items, %d K-tiles; per K-tile %.0f cycles, of which barrier wait %.0f, "
          "A-fragment wait %.0f; epilogue %.0f cycles per item; whole item %.0f cycles (K loops %.0f); clock %.2f GHz"
          % (items, kts, loop / max(kts, 1), bar / max(kts, 1), a / max(kts, 1), epi / max(items, 1),
             whole / max(items, 1), loop / max(items, 1), loop / max(real, 1) * 0.1))
