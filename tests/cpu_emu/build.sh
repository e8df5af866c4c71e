#!/bin/bash
# TEST INFRASTRUCTURE ONLY: sanitizer build of the lane-per-env kernels for x86 (see README.md in this directory).
#   tests/cpu_emu/build.sh            AddressSanitizer + UndefinedBehaviorSanitizer
#   SAN="" tests/cpu_emu/build.sh     plain -O1 build (faster, for logic debugging)
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(dirname "$(dirname "$HERE")")"
CXX=/opt/rocm/lib/llvm/bin/clang++
SAN=${SAN--fsanitize=address,undefined -fno-sanitize-recover=undefined}
FLAGS="-std=c++17 -O1 -g -fno-omit-frame-pointer -ffp-contract=off $SAN -DMGX_CPU_EMU -fPIC -I$HERE -I$ROOT/include -I$ROOT/mettagrid_amd/csrc -x c++"
OBJ=$(mktemp -d)
trap 'rm -rf "$OBJ"' EXIT
C="$ROOT/mettagrid_amd/csrc"
$CXX $FLAGS -c $C/mgx_engine.hip -o $OBJ/engine.o &
$CXX $FLAGS -DMGX_SLOT=0 -c $C/mgx_world_fast.hip -o $OBJ/fast0.o &
$CXX $FLAGS -DMGX_SLOT=1 -c $C/mgx_world_fast.hip -o $OBJ/fast1.o &
$CXX $FLAGS -c $C/mgx_world_x.hip -o $OBJ/x.o &
$CXX $FLAGS -c $C/mgx_aoe.hip -o $OBJ/aoe.o &
$CXX $FLAGS -c $HERE/emu_globals.cpp -o $OBJ/globals.o &
for job in $(jobs -p); do wait $job || { echo "compile failed"; exit 1; }; done
$CXX $SAN -shared -o $HERE/libmgx_emu.so $OBJ/*.o
echo "built $HERE/libmgx_emu.so"
