#!/usr/bin/env python3
"""Turn a tools/pmc_s1.sh summary into the stamped record bench.py reports `roofline.traffic` from.
usage: write_pmc_record.py <pmc_fwd.txt> <out.json> [commit]     (on the GPU box: asks the library which tile it runs)"""
import ctypes
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cstp_amd import _lib, ops  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
commit = sys.argv[3] if len(sys.argv) > 3 else os.environ.get("CSTP_COMMIT", "unknown")
vals, kernel = {}, None
for line in open(src):
    m = re.match(r"(void .+?)\s{2,}(\w+)\s+n=\s*\d+\s+last4 avg ([\d.e+]+)", line)
    if m and "igemm" in m.group(1):
        kernel = kernel or m.group(1).replace("void ", "")
        if m.group(1).replace("void ", "") == kernel:
            vals[m.group(2)] = float(m.group(3))
# the S1 forward convolution of tools/one_conv.py: 64 -> 144, 1x3x3, stride 1, pad (0,1,1), 32 clips of 16x56x56
xs, ws = (32, 64, 16, 56, 56), (144, 64, 1, 3, 3)
lib = _lib.load()
desc = ops._desc(xs, ws, (1, 1, 1), (0, 1, 1))
import torch  # noqa: E402
ops.conv3d(torch.rand(xs, device="cuda"), torch.rand(ws, device="cuda") * 0.05, None, 1, (0, 1, 1))   # the layer gets its tile (table or timing)
torch.cuda.synchronize()
tile = (ctypes.c_int32 * 4)()
_lib.check(lib.cstp_conv3d_query_tile(ctypes.byref(desc), 0, tile), "query")
out_bytes = 32 * 144 * 16 * 56 * 56 * 4
in_bytes = 32 * 64 * 16 * 56 * 56 * 4
rec = {
    "kernel": "%s  (S1 spatial convolution forward, 64->144 1x3x3 @16x56x56, 32 clips per launch; the <.., true> instantiation also "
              "leaves the per-channel sums for the BatchNorm behind it, as in the training step)" % kernel,
    "tile": list(tile),
    "tile_note": "cstp_conv3d_query_tile(S1 descriptor, mode 0) when measured: rows, positions, split terms (2 = f16 pair), "
                 "K-tiles per barrier -- bench.py reports `traffic` only while the library still answers this",
    "commit": commit,
    "hbm_bytes_per_launch": int(2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024),
    "algorithmic_bytes_per_launch": in_bytes + out_bytes,
    "FETCH_SIZE_KB_reported": vals["FETCH_SIZE"],
    "FETCH_SIZE_note": "gfx950 reports half the bytes of coalesced streaming reads (MI355X_MICROARCH.md, HBM/rocprofv3 section; "
                       "the x2 holds for 16 B/lane loads -- this kernel's QUAD staging since round 4 -- and was calibrated with tools/calib_fetch.py on 4 B/lane loads too)",
    "WRITE_SIZE_KB_reported": vals["WRITE_SIZE"],
    "counters": vals,
    "command": "rocprofv3 --pmc <counter set> --kernel-trace --output-format csv -- python3 tools/one_conv.py fwd 32   "
               "(tools/pmc_s1.sh via tools/profile_round.sh: one run per counter set; values = average of the last 4 launches)",
}
with open(dst, "w") as f:
    json.dump(rec, f, indent=1)
print("wrote", dst, rec["tile"], rec["hbm_bytes_per_launch"] / 1e6, "MB per launch vs algorithmic", rec["algorithmic_bytes_per_launch"] / 1e6)
