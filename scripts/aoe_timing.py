"""Shader-clock breakdown of the lane-per-agent area-effect kernel at rung 4 (needs scripts/build_timing.sh): cycles of
wavefront 0 of every workgroup between the pieces of the phase.
Usage: MGX_LIB=mettagrid_amd/libmgx_timing.so python scripts/aoe_timing.py [steps]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mettagrid_amd import engine, presets  # noqa: E402
from mettagrid_amd.compiler import compile_spec  # noqa: E402
from mettagrid_amd.mapgen import random_class_maps  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
prog = compile_spec(presets.rung4_spec(), 64, 64, max_objects=presets.RUNG4_MAX_OBJECTS)
E, A = 65536, prog.num_agents
cms = random_class_maps(prog, 64, 64, dict(presets.RUNG4_OBJECTS), dict(presets.RUNG4_AGENTS), range(2048))[np.arange(E) % 2048]
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
        lib.mgx_debug_aoe_cycles(out, 1)
    with torch.cuda.stream(ext):
        eng.actions.copy_(pa[t % 8]); eng.vibe_actions.copy_(pv[t % 8])
        eng.step()
eng.sync()
lib.mgx_debug_aoe_cycles(out, 0)
names = ["load issue", "on_tick (+ first use of the loads)", "fixed AoE (2 passes + net deltas)", "territory", "mobile AoE", "coverage", "store"]
groups = E // 4
tot = sum(out[k] for k in range(len(names)))
for k, nm in enumerate(names):
    print(f"{nm:40s} {out[k] / steps / groups:10.0f} cycles / wavefront / step  {100.0 * out[k] / max(1, tot):5.1f} %")
print(f"{'total':40s} {tot / steps / groups:10.0f}")
