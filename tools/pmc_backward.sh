#!/bin/bash
# SQ counters of the backward kernels of the bench frame (gpurun): --pmc passes with --kernel-trace only.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_bwd
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-train-step --no-field"
export CNC_BWD_OVERLAP=0
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o p -- $CMD > $OUT/a.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d $OUT/b -o p -- $CMD > $OUT/b.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ATOMIC_RETURN --output-format csv -d $OUT/c -o p -- $CMD > $OUT/c.log 2>&1
python3 - <<PY
import csv, glob, collections
for tag in "abc":
    files = glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not any(s in k for s in ("bwd_merge", "k_bwd_bin", "k_bwd_owner", "fwd_bits")): continue
            acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(tag, k)
        for c, v in sorted(d.items()):
            print("    %-34s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
