#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats kernel_stats.csv into a markdown table.

usage: summarize_rocprof.py <kernel_stats.csv> <passes> [title] > profiles/<name>.md
`passes` = number of workload passes the profiled command ran (warm-up + timed +
the profiling pass), used to express totals per pass."""
import csv
import re
import sys


def main():
    path, passes = sys.argv[1], int(sys.argv[2])
    title = sys.argv[3] if len(sys.argv) > 3 else path
    rows = list(csv.DictReader(open(path)))
    mine = lambda r: "rst::" in r["Name"] or r["Name"].startswith("_ZN3rst")
    ours = [r for r in rows if mine(r)]
    other = [r for r in rows if not mine(r)]
    tot_ours = sum(float(r["TotalDurationNs"]) for r in ours) / 1e6
    tot_other = sum(float(r["TotalDurationNs"]) for r in other) / 1e6
    print(f"# {title}\n")
    print(f"passes in the profiled command: {passes}; library kernels {tot_ours / passes:.2f} ms per pass; "
          f"other kernels (MIOpen/ATen, including MIOpen's one-off solver search) {tot_other:.1f} ms in total\n")
    print("| kernel | calls | calls/pass | avg us | total ms | ms/pass |")
    print("|---|---|---|---|---|---|")
    for r in sorted(ours, key=lambda r: -float(r["TotalDurationNs"])):
        name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        mm = re.match(r"_ZN3rst(?:12_GLOBAL__N_1)?(\d+)", name)
        if mm:
            name = "rst::" + name[mm.end():mm.end() + int(mm.group(1))]
        c = int(r["Calls"])
        t = float(r["TotalDurationNs"]) / 1e6
        print(f"| `{name}` | {c} | {c / passes:.1f} | {float(r['AverageNs']) / 1e3:.1f} | {t:.2f} | {t / passes:.3f} |")
    print("\n## other kernels (top 12 by time)\n")
    print("| kernel | calls | avg us | total ms |")
    print("|---|---|---|---|")
    for r in sorted(other, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
        print(f"| `{r['Name'][:80]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['TotalDurationNs']) / 1e6:.2f} |")


if __name__ == "__main__":
    main()
