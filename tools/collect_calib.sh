#!/bin/bash
# Runs on the GPU box (gpurun): the counter-calibration probe (tools/fetch_calib.hip, built here by hipcc) alone
# for its timings, then under four separate rocprofv3 --pmc passes (counters only, no trace domains).
# -> gpurun_out/calib/ ; tools/summarise_calib.py turns it into profiles/r03_counter_calibration.{md,json}.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/calib
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BIN=$ROOT/tools/fetch_calib
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $BIN $ROOT/tools/fetch_calib.hip
timeout 300 $BIN > $OUT/timing.jsonl 2> $OUT/timing.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $BIN > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $BIN > $OUT/write.log 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --output-format csv -d $OUT/rdreq -o p -- $BIN > $OUT/rdreq.log 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/wrreq -o p -- $BIN > $OUT/wrreq.log 2>&1
find $OUT -name "*.db" -delete
cat $OUT/timing.jsonl
ls -R $OUT | head -40
