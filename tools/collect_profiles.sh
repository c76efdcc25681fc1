#!/bin/bash
# Runs on the GPU box (gpurun): the judged bench line, rocprofv3 kernel stats of the same command, PMC passes
# (FETCH_SIZE / WRITE_SIZE / raw L2->fabric request counters / TCC_ATOMIC_sum: separate runs, counters only, no trace
# domains), the bench without the stream overlap, the counter calibration probe, and the kernel stats of the
# full-size training step.  -> gpurun_out/<tag>/ ; `python tools/summarise_pmc.py <tag>` turns them into
# profiles/<tag>_* and profiles/traffic.json (with the git blob hashes of the kernels it was measured on).
#   bash tools/collect_profiles.sh <tag> [all|bench]        (tag: r04, ...)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WHAT=${2:-all}
timeout 1200 python $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-step"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $ROOT/bench.py --no-cpu-baseline --no-train-step > $OUT/stats.json 2> $OUT/stats.log
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
timeout 900 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/pmc_req -o p -- $CMD > $OUT/pmc_req.log 2>&1
timeout 900 rocprofv3 --pmc TCC_ATOMIC_sum --output-format csv -d $OUT/pmc_atomic -o p -- $CMD > $OUT/pmc_atomic.log 2>&1
# SURVEY 8(d) Input B: the reference composition's four encoders on 2^18 marched samples (the same counters, own passes)
CMDB="python $ROOT/tools/bench_input_b.py"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_b_fetch -o p -- $CMDB > $OUT/pmc_b_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_b_write -o p -- $CMDB > $OUT/pmc_b_write.log 2>&1
timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/pmc_b_req -o p -- $CMDB > $OUT/pmc_b_req.log 2>&1
timeout 600 rocprofv3 --pmc TCC_ATOMIC_sum --output-format csv -d $OUT/pmc_b_atomic -o p -- $CMDB > $OUT/pmc_b_atomic.log 2>&1
timeout 600 python $ROOT/tools/bench_input_b.py > $OUT/input_b.log 2>&1
CNC_BWD_OVERLAP=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_no_overlap -o bench -- python $ROOT/bench.py --no-cpu-baseline --no-train-step > $OUT/bench_no_overlap.json 2> $OUT/stats_no_overlap.log
if [ "$WHAT" = "all" ]; then
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_train -o train -- python $ROOT/tools/bench_train.py --no-profile > $OUT/train.log 2>&1
  bash $ROOT/tools/collect_calib.sh > $OUT/calib.log 2>&1
  # the fused field kernels: MFMA busy cycles next to the wall clock (counters only, with --kernel-trace)
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_field -o p -- python $ROOT/tools/bench_field.py --reps 3 > $OUT/pmc_field.log 2>&1
  CNC_FUSED_FIELD_MFMA=f32 timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_field_f32 -o p -- python $ROOT/tools/bench_field.py --only fused --reps 3 > $OUT/pmc_field_f32.log 2>&1
  timeout 600 python $ROOT/tools/bench_field.py > $OUT/field_untraced.log 2>&1
  # the 800x800 evaluation render: kernel stats; the training step's phase timeline (no profiler) and library launches
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_eval -o eval -- python $ROOT/tools/bench_eval.py > $OUT/eval.log 2>&1
  timeout 600 python $ROOT/tools/bench_eval.py > $OUT/eval_untraced.log 2>&1
  timeout 600 python $ROOT/tools/step_phases.py > $OUT/step_phases.txt 2>&1
  timeout 600 python $ROOT/tools/aten_by_range.py > $OUT/aten_by_range.txt 2>&1
  timeout 600 python $ROOT/tools/bench_train.py --no-profile > $OUT/train_untraced.log 2>&1
  # one steady-state step as a kernel timeline per queue; the field's gradient pass alone (forward + backward at 2^18 samples)
  bash $ROOT/tools/step_kernels.sh > $OUT/step_kernels.log 2>&1
  cp $ROOT/gpurun_out/step_kernels/timeline.txt $OUT/step_timeline.txt 2>/dev/null
  timeout 600 python $ROOT/tools/bench_chain.py --profile > $OUT/bench_chain.log 2>&1
  timeout 600 python $ROOT/tools/step_wall_by_kind.py > $OUT/step_wall_by_kind.txt 2>&1
fi
find $OUT -name "*kernel_trace.csv" -delete      # large; the stats csv is what is kept
find $OUT -name "*.db" -delete
ls -R $OUT | head -60
