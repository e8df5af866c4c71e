# Shader-clock phase breakdowns of the world, observation and area-effect kernels (needs scripts/build_timing.sh first).
# Usage (GPU box): bash scripts/gpu_timing.sh
set -e
export MGX_LIB=$GRAFT_REPO_ROOT/mettagrid_amd/libmgx_timing.so
timeout -k 10 300 python scripts/world_timing.py 30
echo "== obs rung 3"; timeout -k 10 300 python scripts/obs_timing.py 3 30
echo "== obs rung 4"; timeout -k 10 300 python scripts/obs_timing.py 4 20
timeout -k 10 400 python scripts/aoe_timing.py 20
