set -e
export MGX_LIB=$GRAFT_REPO_ROOT/mettagrid_amd/libmgx_timing.so
echo "== rung 3"; timeout -k 10 300 python scripts/obs_timing.py 3 30
echo "== rung 4"; timeout -k 10 300 python scripts/obs_timing.py 4 20
