# Timing experiments on the two-wave fused field kernel (results of the experiment builds are NOT valid outputs).
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w2
: > gpurun_out/w2/exp2.log
for flags in "" "-DCNC_EXP_SAMEROW" "-DCNC_EXP_HOTWEIGHTS" "-DCNC_EXP_SAMEROW -DCNC_EXP_HOTWEIGHTS" "-DCNC_EXP_NOMFMA" "-DCNC_EXP_NOFILL" "-DCNC_EXP_NOFILL -DCNC_EXP_HOTWEIGHTS" "-DCNC_EXP_NOFILL -DCNC_EXP_NOMFMA" "-DCNC_EXP_NOFILL -DCNC_EXP_NOBARRIER"; do
  CNC_HIP_EXTRA_FLAGS="$flags" python -m cnc_amd.build --force > /dev/null 2>&1
  echo "== flags: $flags" >> gpurun_out/w2/exp2.log
  timeout 300 python tools/bench_field.py --only fused --mode density 2>&1 | grep fused >> gpurun_out/w2/exp2.log
done
cat gpurun_out/w2/exp2.log
