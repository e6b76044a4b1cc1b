#!/usr/bin/env python3
"""Static check of a kernel's ISA for reads of LDS results that have not landed yet.

The MFMA kernels issue their operand reads by hand (inline `ds_read*`) several steps ahead and wait
with a counted `s_waitcnt lgkmcnt(N)`.  To the compiler the destination of such a read holds its value
right after the asm statement, so it is free to COPY that register (tied twice in a constraint list,
live-range split, spill) before the wait -- a copy of a value that is not there yet.  This walks each
function linearly, models the in-order LDS queue (every `s_waitcnt lgkmcnt(n)` leaves the youngest n
operations outstanding) and reports any instruction that reads or overwrites a register still in flight.

usage: hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only file.hip -o file.s ; check_lds_inflight.py file.s [name-filter]
"""
import re
import sys


def regs(tok):
    out = []
    for m in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


def check(text, name_filter=""):
    total = 0
    for f in re.split(r"\n(?=_Z\w+:)", text)[1:]:
        name = f.split(":")[0]
        if name_filter not in name or ".amdhsa_" not in f:
            continue
        inflight, bad = [], 0
        for ln, l in enumerate(f.split("\n")):
            if not l.startswith("\t") or l.startswith("\t.") or l.startswith("\t;"):
                continue
            t = l.strip()
            op = t.split()[0]
            if op in ("s_endpgm", "s_branch", "s_setpc_b64"):
                inflight = []             # straight-line flow ends here; what follows is reached from elsewhere
                continue
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", t)
                if m:
                    n = int(m.group(1))
                    inflight = inflight[len(inflight) - n:] if n > 0 else []
                continue
            parts = [a.strip() for a in t[len(op):].split(";")[0].split(",")]
            if op.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle")):
                dst, src = regs(parts[0]), sum((regs(x) for x in parts[1:]), [])
            elif op.startswith(("ds_write", "buffer_store", "global_store", "scratch_store", "flat_store")):
                dst, src = [], sum((regs(x) for x in parts), [])
            else:
                dst, src = (regs(parts[0]) if parts else []), sum((regs(x) for x in parts[1:]), [])
            fl = set(r for d in inflight for r in d)
            hit = [r for r in src if r in fl]
            hitw = [] if op.startswith("ds_") else [r for r in dst if r in fl]
            if hit or hitw:
                bad += 1
                if bad <= 8:
                    print(f"{name[:70]}: line {ln}: {t[:80]}  reads in flight {hit} overwrites in flight {hitw}")
            if op.startswith("ds_"):
                inflight.append(dst)
                inflight = inflight[-15:]
        print(f"{name[:90]}: {bad} violation(s)")
        total += bad
    return total


if __name__ == "__main__":
    sys.exit(1 if check(open(sys.argv[1]).read(), sys.argv[2] if len(sys.argv) > 2 else "") else 0)
