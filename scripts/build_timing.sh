#!/bin/bash
# Instrumented developer build (-DMGX_WORLD_TIMING: shader-clock totals per phase of the world kernels) ->
# mettagrid_amd/libmgx_timing.so; load it with MGX_LIB=mettagrid_amd/libmgx_timing.so (scripts/world_timing*.py).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
C="$ROOT/mettagrid_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMGX_WORLD_TIMING -I $ROOT/include -I $C"
O=$(mktemp -d)
trap 'rm -rf "$O"' EXIT
hipcc $F -c $C/mgx_engine.hip -o $O/e.o &
hipcc $F -DMGX_SLOT=0 -c $C/mgx_world_fast.hip -o $O/f0.o &
hipcc $F -DMGX_SLOT=1 -c $C/mgx_world_fast.hip -o $O/f1.o &
hipcc $F -c $C/mgx_world_x.hip -o $O/x.o &
hipcc $F -c $C/mgx_aoe.hip -o $O/a.o &
hipcc $F -c $C/mgx_decode.hip -o $O/d.o &
hipcc $F -c $C/mgx_obs_box.hip -o $O/b.o &
hipcc $F -DMGX_SLOT=0 -c $C/mgx_act_fast.hip -o $O/af0.o &
hipcc $F -c $C/mgx_act_x.hip -o $O/ax.o &
hipcc $F -c $C/mgx_pack.hip -o $O/pk.o &
for job in $(jobs -p); do wait $job; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/mettagrid_amd/libmgx_timing.so $O/*.o
echo built libmgx_timing.so
