set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mix
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mix_probe tools/mix_probe.hip 2>/dev/null
timeout 300 /tmp/mix_probe > gpurun_out/mix/mix_probe.jsonl 2>&1
cat gpurun_out/mix/mix_probe.jsonl
timeout 1500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_field_fused.py tests/test_gpu_field_golden.py -x -q -m gpu 2>&1 | tail -4
