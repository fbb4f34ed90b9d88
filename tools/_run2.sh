set -o pipefail
cd $GRAFT_REPO_ROOT
python tools/cloud_churn.py || exit 1
