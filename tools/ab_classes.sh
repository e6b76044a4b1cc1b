#!/bin/bash
# Same-box A/B of library builds: per-class milliseconds of the headline step (fp32 inference, 16x3x256x256) for the
# in-tree library and for every other build named on the command line (files under tools/bin/, selected through
# RST_LIB_PATH).  Box-to-box spread is +-3 %, so variants are only ever compared inside one gpurun call.
#   bash tools/ab_classes.sh [classes=gemm_f16s,pwdw_f16s,...] [lib_a.so lib_b.so ...]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
classes="gemm_f16s,pwdw_f16s,pwdw_gate_f16s,gram_f16s,gemm_f16s_ln,conv3x3_f16s,gemm_nn"
case "$1" in classes=*) classes="${1#classes=}"; shift;; esac
export RST_RANGE_CHECK=0        # (timing-only builds return garbage: keep the range guard from switching routes)
for lib in "" "$@"; do
  if [ -n "$lib" ]; then export RST_LIB_PATH=$root/tools/bin/$lib; fi
  timeout -k 10 200 python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-runs > /tmp/ab_classes.json 2> /tmp/ab_classes.err \
    || { echo "FAILED ${lib:-in-tree}"; tail -3 /tmp/ab_classes.err; exit 1; }
  python3 - "${lib:-in-tree}" "$classes" <<'PY'
import json, sys
d = json.loads(open('/tmp/ab_classes.json').read().strip().splitlines()[-1])
k = d['kernels']
print(sys.argv[1], 'step %.2f ms ' % d['ms_per_step'], ' '.join('%s %.3f' % (n, k[n]['ms']) for n in sys.argv[2].split(',') if n in k))
PY
done
