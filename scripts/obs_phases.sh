#!/bin/bash
# Builds libmgx_stopK.so (K = 1, 3, 4): the observation kernel returns after phase K (1 staging; 3 classification,
# token lists, visible-cell lists, global tokens; 4 encode).  Run bench.py with MGX_LIB=... on the GPU box; differences
# of the obs-kernel time attribute it.  Not parity builds.
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Imettagrid_amd/csrc"
mkdir -p /tmp/obs_phases
/opt/rocm/bin/hipcc $F -c mettagrid_amd/csrc/mgx_world_fast.hip -o /tmp/obs_phases/fast.o &
for k in 1 3 4; do
  /opt/rocm/bin/hipcc $F -DMGX_OBS_STOP=$k -c mettagrid_amd/csrc/mgx_engine.hip -o /tmp/obs_phases/eng$k.o &
done
wait
for k in 1 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mettagrid_amd/libmgx_stop$k.so /tmp/obs_phases/fast.o /tmp/obs_phases/eng$k.o
done
