set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fwd
: > gpurun_out/fwd/log.txt
for flags in "-DCNC_NO_XPAIR" "-DCNC_NO_XPAIR -DCNC_FWD_BITS_BOUNDS=__launch_bounds__(256,8)" ""; do
  CNC_HIP_EXTRA_FLAGS="$flags" python -m cnc_amd.build --force > /dev/null 2>&1
  echo "== flags: $flags" >> gpurun_out/fwd/log.txt
  for r in 1 2; do
  timeout 600 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-train-step --no-field 2>/dev/null | python -c "
import json,sys
o=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print(o['value'], o['ms_per_step'], {k:round(v['avg_ms'],4) for k,v in o['kernels'].items() if 'forward' in k or 'backward' in k})" >> gpurun_out/fwd/log.txt
  done
  timeout 300 python tools/bench_field.py --only fused 2>&1 | grep fused >> gpurun_out/fwd/log.txt
done
cat gpurun_out/fwd/log.txt
