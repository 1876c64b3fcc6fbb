#!/usr/bin/env python3
"""Run the forward autotuner on one layer and print its ranking (CSTP_TUNE_VERBOSE=1).  usage: tune_one.py [T1|S1|T3]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["CSTP_TUNE_TABLE"] = "0"
from cstp_amd import ops
L = {"T1": ((32, 144, 16, 56, 56), 64, (3, 1, 1), (1, 0, 0)), "S1": ((32, 64, 16, 56, 56), 144, (1, 3, 3), (0, 1, 1)),
     "T3": ((32, 288, 8, 28, 28), 128, (3, 1, 1), (1, 0, 0))}
xs, k, ks, pad = L[sys.argv[1] if len(sys.argv) > 1 else "T1"]
x = torch.randn(xs, device="cuda"); w = torch.randn((k, xs[1]) + ks, device="cuda") * 0.05
y = ops.conv3d(x, w, None, 1, pad)
torch.cuda.synchronize()
lib = ops._lib.load(); arr = (ctypes.c_int32 * 4)()
lib.cstp_conv3d_get_tile(ctypes.byref(ops._desc(xs, tuple(w.shape), (1, 1, 1), pad)), 0, arr)
print("picked", list(arr))
