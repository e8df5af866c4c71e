set -e
export MGX_LIB=$GRAFT_REPO_ROOT/mettagrid_amd/libmgx_timing.so
timeout -k 10 400 python scripts/aoe_timing.py 20
