set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05b
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r05b/gpu_tests.log
cat gpurun_out/r05b/gpu_tests.log
bash tools/collect_profiles.sh r05 all > gpurun_out/r05_collect.log 2>&1
tail -3 gpurun_out/r05_collect.log
