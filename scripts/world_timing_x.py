"""Per-phase shader-clock breakdown of the extended world kernel at rung 4 (needs scripts/build_timing.sh).
Usage: MGX_LIB=mettagrid_amd/libmgx_timing.so python scripts/world_timing_x.py [steps]"""
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
cms = random_class_maps(prog, 64, 64, dict(presets.RUNG4_OBJECTS), dict(presets.RUNG4_AGENTS), range(E))
eng = engine.BatchedMettaGrid(prog, cms, np.arange(E, dtype=np.uint32), device=0, buffers="device")
lib = engine.load_lib()
n = len(prog.action_names)
gen = torch.Generator(device="cuda").manual_seed(42)
pa = torch.randint(0, n, (8, E * A), dtype=torch.int32, device="cuda", generator=gen)
pv = torch.randint(0, n, (8, E * A), dtype=torch.int32, device="cuda", generator=gen)
ext = torch.cuda.ExternalStream(eng.stream)
out = (C.c_ulonglong * 16)()
for t in range(5 + steps):
    if t == 5:
        eng.sync()
        lib.mgx_debug_world_x_cycles(out, 1)
    with torch.cuda.stream(ext):
        eng.actions.copy_(pa[t % 8]); eng.vibe_actions.copy_(pv[t % 8])
        eng.step()
eng.sync()
lib.mgx_debug_world_x_cycles(out, 0)
names = ["stage agents", "shuffle", "stream0 (move/noop)", "stream1 (vibe)", "events/on_tick/aoe/game tick", "flush + coverage",
         "  do_move + vibe (in streams)", "  action bookkeeping (in streams)", "    move handler: cell read", "    move handler: apply",
         "    use handler: cell read", "    use handler: apply"]
waves = (E + 31) // 32
for k, nm in enumerate(names):
    print(f"{nm:36s} {out[k] / steps / waves:12.0f} cycles / wave / step (all launches of the step)")
