#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV and reports, per queue, every kernel that STARTED before the kernel dispatched
just before it on the same queue had ENDED (an in-order queue with the barrier bit set must never show one)."""
import csv
import glob
import sys
from collections import defaultdict

paths = [p for a in sys.argv[1:] for p in glob.glob(a, recursive=True)]
rows = []
for p in paths:
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append(r)
print("rows", len(rows), "columns", list(rows[0].keys()) if rows else None)
byq = defaultdict(list)
for r in rows:
    byq[(r.get("Agent_Id"), r.get("Queue_Id"), r.get("Stream_Id", ""))].append(r)
for q, rs in sorted(byq.items()):
    rs.sort(key=lambda r: int(r.get("Dispatch_Id") or r.get("Correlation_Id")))
    overlaps, worst = 0, 0
    gaps = []
    for a, b in zip(rs, rs[1:]):
        ea, sb = int(a["End_Timestamp"]), int(b["Start_Timestamp"])
        gaps.append(sb - ea)
        if sb < ea:
            overlaps += 1
            worst = max(worst, ea - sb)
            if overlaps <= 12:
                print(f"  OVERLAP on {q}: {b['Kernel_Name'][:60]} started {ea - sb} ns before {a['Kernel_Name'][:60]} ended "
                      f"(dispatch ids {a.get('Dispatch_Id')} -> {b.get('Dispatch_Id')})")
    gaps.sort()
    print(f"queue {q}: {len(rs)} dispatches, {overlaps} overlapping successors (worst {worst} ns); "
          f"gap min/median {gaps[0] if gaps else None}/{gaps[len(gaps) // 2] if gaps else None} ns")
# concurrency across queues, as a sanity check that the two streams really ran side by side
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
cur = mx = 0
for _, d in ev:
    cur += d; mx = max(mx, cur)
print("most kernels in flight at once:", mx)
