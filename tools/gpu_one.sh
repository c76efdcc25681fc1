set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w2
CNC_HIP_EXTRA_FLAGS="-DCNC_EXP_NOFILL" python -m cnc_amd.build --force > /dev/null 2>&1
bash tools/pmc_field.sh --mode density > gpurun_out/w2/pmc_nofill.log 2>&1
grep -A9 "^a void cnc::k_field_fused16w2<8u, 5, false" gpurun_out/w2/pmc_nofill.log | head -12
grep -A8 "^b void cnc::k_field_fused16w2<8u, 5, false" gpurun_out/w2/pmc_nofill.log | head -10
