set -e
export MGX_LIB=$GRAFT_REPO_ROOT/mettagrid_amd/libmgx_timing.so
timeout -k 10 300 python scripts/world_timing.py 30
