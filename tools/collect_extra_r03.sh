#!/bin/bash
# Runs on the GPU box: the judged bench line with the fresh traffic.json, MFMA counters of the MLP work, the CU-mask probe.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
[ -x $ROOT/tools/cumask_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $ROOT/tools/cumask_probe $ROOT/tools/cumask_probe.hip
$ROOT/tools/cumask_probe > $OUT/cumask.jsonl 2> $OUT/cumask.err
rm -rf $OUT/mfma_r03
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma_r03 -o m -- python $ROOT/tools/mlp_pmc_r03.py > $OUT/mfma_r03.log 2>&1
find $OUT/mfma_r03 -name "*.db" -delete
timeout 900 python $ROOT/bench.py > $OUT/bench_r03b.json 2> $OUT/bench_r03b.err
cat $OUT/cumask.jsonl
