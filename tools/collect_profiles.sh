#!/bin/bash
# Runs on the GPU box (gpurun): the judged bench line, the rocprofv3 kernel stats of the same command,
# and the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, no trace domains) -> gpurun_out/r01/.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r01
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 python $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
# kernel stats of the SAME command as the bench line above (default flags)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/bench.py > $OUT/stats.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
# the same bench with the coarse backward call NOT forked onto a side stream: the three backward kernels
# then run back to back and the sum of their averages is the call duration bench.py reports
CNC_BWD_OVERLAP=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_no_overlap -- python $ROOT/bench.py --no-cpu-baseline > $OUT/bench_no_overlap.json 2> $OUT/stats_no_overlap.log
find $OUT -name "*kernel_trace.csv" -delete      # large; the stats csv is what is kept
ls -R $OUT | head -40
