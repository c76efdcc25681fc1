set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_field_chain.py tests/test_gpu_field_golden.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/bench_chain.py --profile 2>&1 | grep -E "chain=|bwd_chain|Self CUDA time"
for flags in "-DCNC_BWD_WAVES=2 -DCNC_BWD_DB=true" "-DCNC_BWD_WAVES=3 -DCNC_BWD_DB=true"; do
  CNC_HIP_EXTRA_FLAGS="$flags" python -m cnc_amd.build --force > /dev/null 2>&1
  echo "== $flags"
  timeout 600 python tools/bench_chain.py --profile 2>&1 | grep -E "chain=True|bwd_chain"
done
