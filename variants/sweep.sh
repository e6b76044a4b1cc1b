#!/bin/bash
# usage: sweep.sh  (prints pk_pwdw / pk_pwdw_gate ms per setting)
for nwb in 2 3 4 8; do for rows in 32 64; do
  RST_BF16_RING_NWB=$nwb RST_BF16_RING_ROWS=$rows timeout -k 10 120 python bench.py --dtype bf16 --steps 10 --warmup 3 --no-side-runs > gpurun_out/sw_${nwb}_${rows}.json 2>/dev/null || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/sw_${nwb}_${rows}.json"))
k=d["kernels"]
print("nwb=${nwb} rows=${rows}: step %.3f ms  pwdw %.3f  gate %.3f" % (d["ms_per_step"], k["pk_pwdw"]["ms"], k["pk_pwdw_gate"]["ms"]))
PY
done; done
