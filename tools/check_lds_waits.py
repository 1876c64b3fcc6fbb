#!/usr/bin/env python3
"""Static check of the hand-counted LDS waits in the patch kernels (round-2 ADVICE: the B-fragment ds_read_b128 pairs of
igemm_k1p / igemm_k1t are inline asm consumed behind hand-counted `s_waitcnt lgkmcnt(N)` that the compiler's own wait
insertion cannot see).  On the gfx950 assembly of igemm.hip, for EVERY ds_read_b128 of every igemm_k1p / igemm_k1t
instantiation: walk forward (through the loop's back edge once) to the first instruction that READS one of its destination
registers and require an `s_waitcnt` with an lgkmcnt term in between whose count N is at most the number of LDS / scalar-memory
instructions issued after the read up to that wait (the counter retires in order, so the read has landed iff it is not among
the youngest N).  A compiler that moved a use above its wait, or copied a fragment register early, fails here.

    python3 tools/check_lds_waits.py igemm.s        (tools/check_k1p_isa.sh runs it)
"""
import re
import sys

VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LGKM = re.compile(r"lgkmcnt\((\d+)\)")


def regs(tok):
    out = set()
    for m in VREG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse(line):
    """(mnemonic, dest regs, source regs) of one instruction line; stores / MFMA accumulators handled conservatively."""
    code = line.split(";")[0].strip()
    if not code or code.endswith(":") or code.startswith("."):
        return None
    parts = code.split(None, 1)
    mn = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    if mn.startswith(("ds_write", "ds_store", "global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "ds_add",
                      "ds_min", "ds_max")):
        return mn, set(), set().union(*[regs(o) for o in ops]) if ops else set()
    if not ops:
        return mn, set(), set()
    return mn, regs(ops[0]), set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()


def check_function(name, body):
    ins = [(i, parse(l), l) for i, l in enumerate(body)]
    ins = [(i, p, l) for i, p, l in ins if p is not None]
    # label -> position in `ins` order
    label_pos = {}
    for i, l in enumerate(body):
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            nxt = next((k for k, (j, _, _) in enumerate(ins) if j > i), None)
            if nxt is not None:
                label_pos[m.group(1)] = nxt
    n_reads = bad = 0
    for k, (i, (mn, dst, src), l) in enumerate(ins):
        if not mn.startswith("ds_read_b128"):
            continue
        n_reads += 1
        later = 0          # LGKM-counter instructions issued after the read
        ok_wait = False
        j, hops = k + 1, 0
        found = None
        steps = 0
        live = set(dst)    # destination registers that still hold the fragment (any later write to one ends its life there)
        while j < len(ins) and steps < 6000:
            steps += 1
            _, (m2, d2, s2), l2 = ins[j]
            if m2 == "s_waitcnt":
                mm = LGKM.search(l2)
                if mm and int(mm.group(1)) <= later:
                    ok_wait = True
            if s2 & live:
                found = l2
                break
            if d2 >= dst and m2.startswith("ds_read"):       # the ring slot is re-loaded: this read's value was never used past here
                break
            live -= d2       # (an MFMA's accumulator operand appears among its sources too, so an overwritten fragment register is
            if not live:     #  never mistaken for a use; a fragment the compiler recycles as a temporary before reloading it is dead)
                break
            if m2.startswith(("ds_", "s_load", "s_buffer_load")):
                later += 1
            if m2.startswith(("s_cbranch", "s_branch")) and hops < 2:
                tgt = l2.split()[-1]
                if tgt in label_pos and label_pos[tgt] <= j:      # a back edge: follow it once (fragments prefetched across K-tiles)
                    j = label_pos[tgt]
                    hops += 1
                    continue
            j += 1
        if found is not None and not ok_wait:
            bad += 1
            if bad <= 3:
                print("  %s: %s  is read by  %s  without a covering lgkmcnt wait" % (name, l.strip(), found.strip()))
    return n_reads, bad


def main():
    txt = open(sys.argv[1]).read()
    total_bad = 0
    for m in re.finditer(r"^(_ZN4cstp9igemm_k1[pt]\S+):", txt, re.M):
        name = m.group(1)
        body = txt[m.end():txt.index(".Lfunc_end", m.end())].split("\n")
        short = re.sub(r"^_ZN4cstp9(igemm_k1[pt])ILi(\d+)ELb(\d)(?:ELb(\d))?.*", lambda q: "%s<%s, %s%s>" % (
            q.group(1), q.group(2), q.group(3), (", " + q.group(4)) if q.group(4) is not None else ""), name)
        n, bad = check_function(short, body)
        print("%-28s %4d ds_read_b128, every first use behind a covering s_waitcnt lgkmcnt: %s" % (short, n, "yes" if bad == 0 else "NO (%d)" % bad))
        total_bad += bad
    sys.exit(1 if total_bad else 0)


if __name__ == "__main__":
    main()
