#!/bin/bash
# The 5000-step procedural protocol under 3 seeds x 2 schedules (gpurun): how far apart are two runs of ONE configuration?
# -> gpurun_out/seeds/{results.txt, run_*.log}; tools/seed_spread_summary.py turns them into profiles/<tag>_procedural_end_to_end.md
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=$ROOT/gpurun_out/seeds
rm -rf $OUT; mkdir -p $OUT
for thread in 1 0; do
  for seed in 42 43 44; do
    CNC_CTX_THREAD=$thread timeout 900 python -m cnc_amd.train --dataset procedural --image_size 400 --n_features 8 \
      --sample_num 150000 --max_steps 5000 --test_views 8 --seed $seed --results $OUT/results_t${thread}.txt \
      --out_dir /tmp/bits_${thread}_${seed} > $OUT/run_t${thread}_s${seed}.log 2>&1
    tail -2 $OUT/run_t${thread}_s${seed}.log
  done
done
cat $OUT/results_t1.txt $OUT/results_t0.txt
