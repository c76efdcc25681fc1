set -u
cd $GRAFT_REPO_ROOT
timeout 900 python tools/ab_train.py chain 2>&1 | grep "chain o"
