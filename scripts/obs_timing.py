"""Shader-clock breakdown of the observation kernel (wavefront 0 of every workgroup) at rung 3 or 4 (needs
scripts/build_timing.sh).  Usage: MGX_LIB=mettagrid_amd/libmgx_timing.so python scripts/obs_timing.py [rung] [steps]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mettagrid_amd import engine, presets  # noqa: E402
from mettagrid_amd.compiler import compile_spec  # noqa: E402
from mettagrid_amd.mapgen import random_class_maps  # noqa: E402

rung = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
E = 65536
if rung == 4:
    prog = compile_spec(presets.rung4_spec(), 64, 64, max_objects=presets.RUNG4_MAX_OBJECTS)
    cms = random_class_maps(prog, 64, 64, dict(presets.RUNG4_OBJECTS), dict(presets.RUNG4_AGENTS), range(2048))[np.arange(E) % 2048]
else:
    prog = compile_spec(presets.rung3_spec(), 32, 32, max_objects=192)
    cms = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, range(8192))[np.arange(E) % 8192]
A = prog.num_agents
eng = engine.BatchedMettaGrid(prog, cms, np.arange(E, dtype=np.uint32), device=0, buffers="device")
lib = engine.load_lib()
n = len(prog.action_names)
gen = torch.Generator(device="cuda").manual_seed(42)
pa = torch.randint(0, n, (8, E * A), dtype=torch.int32, device="cuda", generator=gen)
pv = torch.randint(0, n, (8, E * A), dtype=torch.int32, device="cuda", generator=gen)
ext = torch.cuda.ExternalStream(eng.stream)
out = (C.c_ulonglong * 16)()
warm = 60
for t in range(warm + steps):
    if t == warm:
        eng.sync()
        lib.mgx_debug_obs_cycles(out, 1)
    with torch.cuda.stream(ext):
        eng.actions.copy_(pa[t % 8]); eng.vibe_actions.copy_(pv[t % 8])
        eng.step()
eng.sync()
lib.mgx_debug_obs_cycles(out, 0)
names = {8: "stage env (to the first barrier)", 10: "classify + token lists + window lists + global tokens (to the encode barrier)",
         11: "encode (this wavefront's agents)", 12: "wait for the other encode wavefronts", 13: "visited stamps + token stats", 14: "late rewards"}
tot = sum(out[k] for k in names)
for k, nm in names.items():
    print(f"{nm:90s} {out[k] / steps / E:10.0f} cycles / workgroup / step  {100.0 * out[k] / max(1, tot):5.1f} %")
print(f"{'total':90s} {tot / steps / E:10.0f}   (obs variant {eng.obs_variant})")
