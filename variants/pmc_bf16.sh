#!/bin/bash
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmcq
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pm="--dtype bf16 --steps 1 --warmup 1 --no-cpu-baseline --no-side-runs"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_bf16 -- python3 $root/bench.py $pm > $out/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_bf16 -- python3 $root/bench.py $pm > $out/pmc_write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -- python3 $root/bench.py $pm > $out/pmc_sq.log 2>&1 || true
echo done
