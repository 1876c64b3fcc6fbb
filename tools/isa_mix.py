#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -S listing.  usage: isa_mix.py file.s <mangled-name-substring>"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sys.argv[2] in l and l.rstrip().split(":")[0].endswith(sys.argv[2].split("@")[-1]) or (l.startswith("_Z") and sys.argv[2] in l.split(":")[0]))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
label, blocks = "entry", {}
order = []
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(";"):
        continue
    if re.match(r"^\.LBB\d+_\d+:", t):
        label = t.split(":")[0]
        continue
    if t.startswith("."):
        continue
    if label not in blocks:
        blocks[label] = Counter()
        order.append(label)
    x = t.split()[0]
    c = blocks[label]
    if x.startswith("v_mfma"): c["mfma"] += 1
    elif x.startswith("ds_read") or x.startswith("ds_load"): c["ds_read"] += 1
    elif x.startswith("ds_write") or x.startswith("ds_store"): c["ds_write"] += 1
    elif x.startswith(("global_load", "buffer_load")): c["gload"] += 1
    elif x.startswith(("global_store", "buffer_store")): c["gstore"] += 1
    elif x.startswith("v_"): c["valu"] += 1
    elif x.startswith("s_waitcnt"): c["waitcnt"] += 1
    elif x.startswith("s_barrier"): c["barrier"] += 1
    elif x.startswith("s_"): c["salu"] += 1
    else: c["other"] += 1
for lb in order:
    n = sum(blocks[lb].values())
    if n >= 30:
        print(lb, n, dict(blocks[lb]))
