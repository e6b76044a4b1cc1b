#!/usr/bin/env python3
"""Turn one tools/collect_evidence.sh output directory into the tracked files under profiles/.

usage: publish_evidence.py gpurun_out/<tag> <prefix>        (prefix e.g. r2_b)
Kernel traces were taken with --steps 5 --warmup 3 (+ bench.py's own profiling pass = 9 passes), counters with
--steps 1 --warmup 1 (3 passes)."""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, prefix = sys.argv[1], sys.argv[2]
prof = os.path.join(ROOT, "profiles")


def one(pattern):
    files = glob.glob(os.path.join(src, pattern))
    return files[0] if files else None


for name, dst in (("bench.json", "bench"), ("bench_bf16.json", "bench_bf16"), ("bench_train.json", "bench_train"),
                  ("bench_train_bf16.json", "bench_train_bf16"), ("bench_bf16_2streams.json", "bench_bf16_2streams")):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(prof, f"{prefix}_{dst}.json"))

TITLES = {"kt_fp32": ("kernel_stats", "python3 bench.py --steps 5 --warmup 3 (fp32 inference, 16x3x256x256)"),
          "kt_bf16": ("bf16_kernel_stats", "python3 bench.py --dtype bf16 --steps 5 --warmup 3 (bf16-storage inference, 16x3x256x256)"),
          "kt_train": ("train_kernel_stats", "python3 bench.py --mode train --steps 5 --warmup 3 (fp32 training step, 8x3x128x128)"),
          "kt_train_bf16": ("train_bf16_kernel_stats", "python3 bench.py --mode train --dtype bf16 --steps 5 --warmup 3 (bf16 training step, 8x3x128x128)")}
for d, (dst, what) in TITLES.items():
    st = one(f"{d}/*/*_kernel_stats.csv")
    if not st:
        continue
    shutil.copy(st, os.path.join(prof, f"{prefix}_{dst}.csv"))
    md = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_rocprof.py"), st, "9",
                         f"{prefix}: rocprofv3 --kernel-trace --stats -- {what}"], capture_output=True, text=True, check=True).stdout
    open(os.path.join(prof, f"{prefix}_{dst}.md"), "w").write(md)

for dt, dst, dtype, what in (("fp32", "hbm_traffic", "f32", "bench.py default (Restormer dim=48 fp32 inference, 16x3x256x256)"),
                             ("bf16", "bf16_hbm_traffic", "bf16", "bench.py --dtype bf16 (Restormer dim=48 bf16-storage inference, 16x3x256x256)")):
    f, w = one(f"pmc_fetch_{dt}/*/*_counter_collection.csv"), one(f"pmc_write_{dt}/*/*_counter_collection.csv")
    if f and w:
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_pmc.py"), "traffic", f, w,
                        os.path.join(prof, f"{prefix}_{dst}.json"), "3", what, dtype], check=True)
for dt, dst, dtype, what in (("fp32", "train_hbm_traffic", "f32", "bench.py --mode train (Restormer dim=48 fp32 training step, 8x3x128x128)"),
                             ("bf16", "train_bf16_hbm_traffic", "bf16", "bench.py --mode train --dtype bf16 (Restormer dim=48 bf16 training step, 8x3x128x128)")):
    f, w = one(f"pmc_fetch_train_{dt}/*/*_counter_collection.csv"), one(f"pmc_write_train_{dt}/*/*_counter_collection.csv")
    if f and w:
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_pmc.py"), "traffic", f, w,
                        os.path.join(prof, f"{prefix}_{dst}.json"), "3", what, dtype, "8", "128", "train"], check=True)
m = one("pmc_mfma/*/*_counter_collection.csv")
if m:
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_pmc.py"), "mfma", m,
                    os.path.join(prof, f"{prefix}_mfma_utilisation.json")], check=True)
print("published", prefix)
