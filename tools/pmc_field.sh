#!/bin/bash
# SQ counters of the fused field kernels (gpurun): two --pmc passes with --kernel-trace only.
#   bash tools/pmc_field.sh [extra args of tools/bench_field.py]   -> gpurun_out/pmc_field/{a,b}.csv + summary
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_field
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/bench_field.py --only fused --reps 3 $*"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o p -- $CMD > $OUT/a.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d $OUT/b -o p -- $CMD > $OUT/b.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH --output-format csv -d $OUT/c -o p -- $CMD > $OUT/c.log 2>&1
python3 - <<PY
import csv, glob, collections
for tag in "abc":
    files = glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_field_fused" not in k: continue
            acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(tag, k)
        for c, v in sorted(d.items()):
            print("    %-34s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
tail -3 $OUT/a.log
