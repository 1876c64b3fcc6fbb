#!/bin/bash
# igemm_k1p's consumer loop waits for its LDS reads with COUNTED s_waitcnt lgkmcnt(N).  Scalar memory loads share that counter
# (they can only lengthen such a wait, see the kernel's comment); list the ones the compiler placed between the first and the
# last product of each instantiation (item-level code between the two consumer bodies counts too), and FAIL on scratch use
# (spilled accumulators).
set -e
cd "$(dirname "$0")/.."
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S -Iinclude -Icstp_amd/csrc -o $tmp/igemm.s cstp_amd/csrc/igemm.hip 2>/dev/null
python3 - $tmp/igemm.s <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
bad = 0
for mt, st in ((4, 0), (8, 0), (9, 0), (4, 1), (8, 1), (9, 1)):
    i = txt.index("_ZN4cstp9igemm_k1pILi%dELb%dEEEvNS_5PGeom" % (mt, st))
    body = txt[i:txt.index(".Lfunc_end", i)].split("\n")
    idx = [k for k, l in enumerate(body) if "v_mfma" in l]
    sl = [l.strip() for l in body[idx[0]:idx[-1]] if re.search(r"\bs_(buffer_)?load|\bs_memtime|\bs_memrealtime", l)]
    sp = sum("scratch_" in l for l in body)
    print("igemm_k1p<%d, %s>: %d MFMA, scalar memory instructions among them: %d, scratch instructions: %d" % (mt, "true" if st else "false", len(idx), len(sl), sp))
    bad += sp
# igemm_k2p (weight gradient, igemm_wpatch.h): 256 VGPRs are its whole budget -- a spilled accumulator shows up as scratch
i = txt.index("_ZN4cstp9igemm_k2pE")
body = txt[i:txt.index(".Lfunc_end", i)].split("\n")
sp = sum("scratch_" in l for l in body)
wf = sum("v_readfirstlane" in l for l in body)
print("igemm_k2p: %d MFMA, %d transposing LDS reads, scratch instructions: %d, readfirstlane (waterfall loops around loads): %d"
      % (sum("v_mfma" in l for l in body), sum("ds_read_b64_tr" in l for l in body), sp, wf))
bad += sp + (1 if wf > 12 else 0)
sys.exit(1 if bad else 0)
PY
# every ds_read_b128 of the patch kernels is consumed behind an lgkmcnt wait that covers it (hand-counted waits: round-2 ADVICE)
python3 tools/check_lds_waits.py $tmp/igemm.s
rm -rf $tmp
