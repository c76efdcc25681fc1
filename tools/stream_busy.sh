#!/bin/bash
# Per-stream GPU busy time of the training step (gpurun): rocprofv3 --kernel-trace of tools/bench_train.py, then per
# queue: kernels, busy ms per step; and the union (any stream busy) per step.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/stream_busy
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/sb -o t -- python $ROOT/tools/bench_train.py --no-profile > $OUT/log.txt 2>&1
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections
f = glob.glob("/tmp/sb/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steady state: the last 200 of 300 steps ~ last 2/3 of the kernels by time
t_end = int(rows[-1]["End_Timestamp"]); t_beg = int(rows[0]["Start_Timestamp"])
cut = t_end - (t_end - t_beg) * 0.5
sel = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e6
per_q = collections.defaultdict(lambda: [0, 0.0])
iv = []
for r in sel:
    q = r.get("Queue_Id") or r.get("Stream_Id") or "?"
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    per_q[q][0] += 1; per_q[q][1] += (e - s) / 1e6
    iv.append((s, e))
iv.sort()
union = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s <= ce: ce = max(ce, e)
    else: union += ce - cs; cs, ce = s, e
union += ce - cs
print(f"window {span:.1f} ms, kernels {len(sel)}; union busy {union/1e6:.1f} ms = {union/1e6/span*100:.1f} % of the window")
for q, (n, ms) in sorted(per_q.items(), key=lambda t: -t[1][1]):
    print(f"queue {q}: {n} kernels, busy {ms:.1f} ms = {ms/span*100:.1f} % of the window")
PY
grep "train step" $OUT/log.txt
