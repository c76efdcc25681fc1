#!/bin/bash
# One steady-state training step as a kernel timeline per queue (gpurun): rocprofv3 --kernel-trace of tools/bench_train.py,
# then the kernels of ONE non-refresh step near the end in start order: offset from the step's first kernel, duration,
# queue, name.  The step boundary = the fetch's first kernel (the largest gap-free marker: k_field_prepare is gone, so the
# boundary is found as the kernel that follows the optimizer's last multi_tensor_apply of the step before).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/step_kernels
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/sk -o t -- python $ROOT/tools/bench_train.py --no-profile > $OUT/log.txt 2>&1
python3 - <<PY > $OUT/timeline.txt
import csv, glob, re
f = glob.glob("/tmp/sk/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# step boundaries: the first kernel after a run of optimizer kernels (multi_tensor_apply ... FusedAdam) ends
is_opt = ["FusedAdam" in n or "FusedOptimizer" in n for n in names]
bounds = [i + 1 for i in range(len(rows) - 1) if is_opt[i] and not is_opt[i + 1]]
# the context optimizer follows the field's: keep the LAST boundary of each pair (gaps between boundaries < 50 kernels: same step)
steps = [b for k, b in enumerate(bounds) if k + 1 == len(bounds) or bounds[k + 1] - b > 50]
pick = None
for k in range(len(steps) - 3, 0, -1):          # a step without the occupancy refresh: no k_vote_plan / partition-heavy step
    seg = rows[steps[k]:steps[k + 1]]
    if not any("k_occ" in r["Kernel_Name"] or "vote_plan" in r["Kernel_Name"] for r in seg):
        pick = seg
        break
t0 = int(pick[0]["Start_Timestamp"])
print(f"# {len(pick)} kernels, span {(int(pick[-1]['End_Timestamp']) - t0) / 1e6:.3f} ms")
qs = sorted({r["Queue_Id"] for r in pick})
for r in pick:
    n = re.sub(r"\(.*", "", r["Kernel_Name"])
    n = re.sub(r"at::native::|rocprim::ROCPRIM_\d+_NS::detail::|void |cnc::", "", n)[:90]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{qs.index(r['Queue_Id'])}  {n}")
PY
head -3 $OUT/timeline.txt
grep "train step" $OUT/log.txt
