#!/bin/bash
# Developer builds for A/B measurements: build/libmgx_<name>.so with extra compiler flags for chosen units.
# Usage: scripts/build_variant.sh <name> <unit: engine|fast|x|aoe|decode|box|actf|actx|all> "<extra flags>"; run with MGX_LIB=build/libmgx_<name>.so
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
C="$ROOT/mettagrid_amd/csrc"
NAME="$1"; UNIT="$2"; EXTRA="$3"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $ROOT/include -I $C"
O=$(mktemp -d)
trap 'rm -rf "$O"' EXIT
x() { [ "$UNIT" = "$1" ] || [ "$UNIT" = all ] && echo "$EXTRA"; }
hipcc $F $(x engine) -c $C/mgx_engine.hip -o $O/e.o &
hipcc $F $(x fast) -DMGX_SLOT=0 -c $C/mgx_world_fast.hip -o $O/f0.o &
hipcc $F $(x fast) -DMGX_SLOT=1 -c $C/mgx_world_fast.hip -o $O/f1.o &
hipcc $F $(x x) -c $C/mgx_world_x.hip -o $O/x.o &
hipcc $F $(x aoe) -c $C/mgx_aoe.hip -o $O/a.o &
hipcc $F $(x decode) -c $C/mgx_decode.hip -o $O/d.o &
hipcc $F $(x box) -c $C/mgx_obs_box.hip -o $O/b.o &
hipcc $F $(x actf) -DMGX_SLOT=0 -c $C/mgx_act_fast.hip -o $O/af0.o &
hipcc $F $(x actx) -c $C/mgx_act_x.hip -o $O/ax.o &
for job in $(jobs -p); do wait $job; done
mkdir -p $ROOT/build
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/libmgx_$NAME.so $O/*.o
echo built build/libmgx_$NAME.so
