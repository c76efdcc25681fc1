#!/bin/bash
# Runs on the GPU box (gpurun): the judged bench line, the rocprofv3 kernel stats of the same command, three
# PMC passes (FETCH_SIZE / WRITE_SIZE / TCC_ATOMIC_sum; separate runs, counters only, no trace domains), the
# same bench without the two-stream overlap, and the kernel stats of the full-size training step.
# -> gpurun_out/r02/ ; tools/summarise_pmc_r02.py turns them into profiles/r02_* and profiles/traffic.json.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 python $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-step"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $ROOT/bench.py --no-cpu-baseline --no-train-step > $OUT/stats.json 2> $OUT/stats.log
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
timeout 900 rocprofv3 --pmc TCC_ATOMIC_sum --output-format csv -d $OUT/pmc_atomic -o p -- $CMD > $OUT/pmc_atomic.log 2>&1
CNC_BWD_OVERLAP=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_no_overlap -o bench -- python $ROOT/bench.py --no-cpu-baseline --no-train-step > $OUT/bench_no_overlap.json 2> $OUT/stats_no_overlap.log
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_train -o train -- python $ROOT/tools/bench_train.py --no-profile > $OUT/train.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete      # large; the stats csv is what is kept
find $OUT -name "*.db" -delete
ls -R $OUT | head -60
