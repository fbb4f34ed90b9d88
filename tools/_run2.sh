set -o pipefail
cd $GRAFT_REPO_ROOT
python tools/scheme_breakdown.py N300 16 incremental || exit 1
python tools/scheme_breakdown.py N300 16 icp_edge || exit 1
python tools/scheme_breakdown.py N300 16 ndt_edge || exit 1
