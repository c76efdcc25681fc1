#!/bin/bash
# Runs on the GPU box: untraced training-step time, then rocprofv3 kernel stats of the same script, top kernels per step.
# usage: tools/quick_train_stats.sh <tag> [pytest -k expression]
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-q}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ -n "${2:-}" ]; then
  (cd $ROOT && timeout 900 python -m pytest tests -m gpu -x -q -k "$2" 2>&1 | tail -5) > $OUT/pytest.txt
fi
timeout 600 python $ROOT/tools/bench_train.py --no-profile > $OUT/train_untraced.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_train -o train -- python $ROOT/tools/bench_train.py --no-profile > $OUT/train.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
python - <<PY > $OUT/summary.txt
import csv, glob
f = glob.glob("$OUT/stats_train/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print('GPU ms/step', tot/300/1e6, 'launches/step', sum(int(r['Calls']) for r in rows)/300)
for r in rows[:40]:
    n = r['Name'].replace('void ','').replace('at::native::','')[:64]
    print(f"{n:64s} {int(r['Calls'])/300:7.1f} {float(r['AverageNs'])/1e3:9.1f}us {int(r['TotalDurationNs'])/300/1e6:7.3f}ms")
PY
cat $OUT/pytest.txt 2>/dev/null; grep "train step" $OUT/train_untraced.log; head -30 $OUT/summary.txt
