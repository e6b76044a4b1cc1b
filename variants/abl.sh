#!/bin/bash
for abl in 0 1 2 4 8 14 15; do
  RST_RING_ABL=$abl timeout -k 10 120 python bench.py --dtype bf16 --steps 10 --warmup 3 --no-side-runs > gpurun_out/abl_${abl}.json 2>/dev/null || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/abl_${abl}.json"))
k=d["kernels"]
print("abl=${abl}: step %.3f ms  pwdw %.3f  gate %.3f" % (d["ms_per_step"], k["pk_pwdw"]["ms"], k["pk_pwdw_gate"]["ms"]))
PY
done
