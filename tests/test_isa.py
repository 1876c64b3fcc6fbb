"""The hand-scheduled kernels rely on properties of the GENERATED code that a compiler upgrade could silently break (round-2
ADVICE): igemm_k1p's counted LDS waits assume no scalar memory load among its products' fragment reads and no spilled
accumulator; igemm_k2p lives on exactly the 256 registers a two-waves-per-SIMD block gets and must not wrap its loads in
readfirstlane loops.  tools/check_k1p_isa.sh compiles igemm.hip to gfx950 assembly (no GPU needed) and checks; it fails on
scratch use, and on any LDS fragment read whose first use is not behind an lgkmcnt wait that covers it
(tools/check_lds_waits.py over every igemm_k1p / igemm_k1t instantiation).  ~80 s of hipcc."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="needs hipcc")
def test_generated_code_of_the_hand_scheduled_kernels():
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "check_k1p_isa.sh")], capture_output=True, text=True, timeout=900)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    waits = [l for l in r.stdout.splitlines() if "every first use behind a covering s_waitcnt lgkmcnt" in l]
    # tools/check_lds_waits.py: all 18 patch-kernel instantiations, every ds_read_b128 consumed behind a wait that covers it
    assert len(waits) == 18 and all(l.rstrip().endswith("yes") for l in waits), waits
    lines = [l for l in r.stdout.splitlines() if l.startswith("igemm_") and l not in waits]
    assert len(lines) == 7 and all("scratch instructions: 0" in l for l in lines)
    k2p = [l for l in lines if l.startswith("igemm_k2p")][0]
    assert "486 MFMA" in k2p and "224 transposing LDS reads" in k2p      # 162 tiles x 3 products; 56 reads x 2 K-steps x 2 bodies
