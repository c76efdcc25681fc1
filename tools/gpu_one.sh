set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w2
timeout 1500 python -m pytest tests/test_gpu_field_fused.py tests/test_gpu_field_golden.py -x -q -m gpu 2>&1 | tail -6
timeout 300 python tools/bench_field.py --only fused
timeout 300 python tools/bench_field.py --only fused
CNC_FUSED_FIELD_WAVES=4 timeout 300 python tools/bench_field.py --only fused --mode rgb
