#!/bin/bash
# Runs on the GPU box: the headline backward call through each scatter route (tools/headline_bwd_routes.py) — time per call
# untraced, then one counter pass per route (TCC_ATOMIC_sum alone, no trace domains), and the distinct-cell count.
#   bash tools/collect_routes.sh <tag>   -> gpurun_out/<tag>_routes/ ; python tools/summarise_routes.py <tag> -> profiles/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
OUT=$ROOT/gpurun_out/${TAG}_routes
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for r in product runs merge cells carry; do
  timeout 300 python $ROOT/tools/headline_bwd_routes.py --route $r > $OUT/time_$r.json 2> $OUT/time_$r.err
  timeout 300 rocprofv3 --pmc TCC_ATOMIC_sum --output-format csv -d $OUT/pmc_$r -o p -- python $ROOT/tools/headline_bwd_routes.py --route $r > $OUT/pmc_$r.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$r -o s -- python $ROOT/tools/headline_bwd_routes.py --route $r > $OUT/stats_$r.log 2>&1
done
timeout 600 python $ROOT/tools/headline_bwd_routes.py --count > $OUT/count.txt 2>&1
for r in runs merge cells carry; do timeout 200 python $ROOT/tools/headline_bwd_routes.py --route $r --levels 10 2>/dev/null | tail -1; done > $OUT/coarse10.txt
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
cat $OUT/time_*.json; tail -3 $OUT/count.txt
