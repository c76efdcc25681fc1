set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_field_chain.py -x -q -m gpu 2>&1 | tail -25
