set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_field_fused.py -x -q -m gpu -k "oracle_at_full_size or repeatable or guard" 2>&1 | tail -15
