set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/train
timeout 900 python tools/step_phases.py > gpurun_out/train/phases.txt 2>&1
timeout 900 python tools/bench_train.py --no-profile > gpurun_out/train/bench_train.txt 2>&1
timeout 900 python tools/aten_by_range.py > gpurun_out/train/aten.txt 2>&1
tail -45 gpurun_out/train/phases.txt; cat gpurun_out/train/bench_train.txt | tail -3; grep -n "library kernels" -A50 gpurun_out/train/aten.txt | head -70
