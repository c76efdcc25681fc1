#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel stats of the paths next to the bench — the full-size
# training step, the 800x800 evaluation render and the per-kernel table — into gpurun_out/r01_secondary/.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r01_secondary
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -- python $ROOT/tools/bench_train.py --no-profile > $OUT/train.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eval -- python $ROOT/tools/bench_eval.py > $OUT/eval.log 2>&1
timeout 900 python $ROOT/tools/bench_kernels.py > $OUT/kernel_table.md 2> $OUT/kernel_table.err
find $OUT -name "*kernel_trace.csv" -delete
tail -2 $OUT/train.log $OUT/eval.log
