#!/bin/bash
# Runs on the GPU box: bench lines, kernel traces and HBM counters of the headline workload (fp32 and bf16) and of a
# training step.  Output under gpurun_out/$1/.  Every GPU step has its own time limit; the script stops at the first failure.
set -e
tag=${1:-ev}
part=${2:-all}          # "bench" = bench lines + kernel traces, "pmc" = the counter passes, "all" = both
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
if [ $part != pmc ]; then
timeout -k 10 400 python3 $root/bench.py > $out/bench.json 2> $out/bench.err
timeout -k 10 200 python3 $root/bench.py --dtype bf16 --no-cpu-baseline --no-side-runs > $out/bench_bf16.json 2> $out/bench_bf16.err
timeout -k 10 200 python3 $root/bench.py --mode train --no-cpu-baseline --no-side-runs > $out/bench_train.json 2> $out/bench_train.err
timeout -k 10 200 python3 $root/bench.py --mode train --dtype bf16 --no-cpu-baseline --no-side-runs > $out/bench_train_bf16.json 2> $out/bench_train_bf16.err
# the opt-in two-stream mode of the bf16 path, then per-kernel evidence with the batch as one launch sequence (the default)
RST_TWO_STREAMS=1 timeout -k 10 200 python3 $root/bench.py --dtype bf16 --no-cpu-baseline --no-side-runs > $out/bench_bf16_2streams.json 2> $out/bench_bf16_2streams.err
export RST_TWO_STREAMS=0
common="--steps 5 --warmup 3 --no-cpu-baseline --no-side-runs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_fp32 -- python3 $root/bench.py $common > $out/kt_fp32.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_bf16 -- python3 $root/bench.py --dtype bf16 $common > $out/kt_bf16.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_train -- python3 $root/bench.py --mode train $common > $out/kt_train.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_train_bf16 -- python3 $root/bench.py --mode train --dtype bf16 $common > $out/kt_train_bf16.log 2>&1
fi
if [ $part = bench ]; then echo done > $out/DONE_bench; exit 0; fi
export RST_TWO_STREAMS=0
pm="--steps 1 --warmup 1 --no-cpu-baseline --no-side-runs"
for dt in fp32 bf16; do
  extra=""; [ $dt = bf16 ] && extra="--dtype bf16"
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$dt -- python3 $root/bench.py $extra $pm > $out/pmc_fetch_$dt.log 2>&1
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$dt -- python3 $root/bench.py $extra $pm > $out/pmc_write_$dt.log 2>&1
done
for dt in fp32 bf16; do
  extra=""; [ $dt = bf16 ] && extra="--dtype bf16"
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_train_$dt -- python3 $root/bench.py --mode train $extra $pm > $out/pmc_fetch_train_$dt.log 2>&1
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_train_$dt -- python3 $root/bench.py --mode train $extra $pm > $out/pmc_write_train_$dt.log 2>&1
done
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma -- python3 $root/bench.py $pm > $out/pmc_mfma.log 2>&1
echo done > $out/DONE
