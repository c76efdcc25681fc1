set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w2
timeout 1500 python -m pytest tests/test_gpu_field_fused.py tests/test_gpu_field_golden.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python tools/bench_field.py --only fused
timeout 300 python tools/bench_field.py --only fused
bash tools/pmc_field.sh --mode density > gpurun_out/w2/pmc_lean2.log 2>&1
grep -A9 "^a void cnc::k_field_fused16w2<8u, 5, false" gpurun_out/w2/pmc_lean2.log | head -12
grep -A8 "^b void cnc::k_field_fused16w2<8u, 5, false" gpurun_out/w2/pmc_lean2.log | head -10
