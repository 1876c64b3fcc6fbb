#!/bin/bash
# igemm_k1p's consumer loop waits for its LDS reads with COUNTED s_waitcnt lgkmcnt(N): that is only sound while no scalar
# memory load (they return out of order and share the counter) sits between the products.  Disassemble and check.
set -e
cd "$(dirname "$0")/.."
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S -Iinclude -Icstp_amd/csrc -o $tmp/igemm.s cstp_amd/csrc/igemm.hip 2>/dev/null
python3 - $tmp/igemm.s <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
bad = 0
for mt in (4, 8, 9):
    i = txt.index("_ZN4cstp9igemm_k1pILi%dEEEvNS_5PGeom" % mt)
    body = txt[i:txt.index(".Lfunc_end", i)].split("\n")
    idx = [k for k, l in enumerate(body) if "v_mfma" in l]
    sl = [l.strip() for l in body[idx[0]:idx[-1]] if re.search(r"\bs_(buffer_)?load|\bs_memtime|\bs_memrealtime", l)]
    sp = sum("scratch_" in l for l in body)
    print("igemm_k1p<%d>: %d MFMA, scalar memory instructions among them: %d, scratch instructions: %d" % (mt, len(idx), len(sl), sp))
    bad += len(sl) + sp
sys.exit(1 if bad else 0)
PY
rm -rf $tmp
