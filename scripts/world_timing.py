"""Per-phase shader-clock breakdown of the world kernel (needs a -DMGX_WORLD_TIMING build, see MGX_LIB).
Usage: MGX_LIB=mettagrid_amd/libmgx_timing.so python scripts/world_timing.py [steps]"""
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
prog = compile_spec(presets.rung3_spec(), 32, 32, max_objects=192)
E, A = 65536, prog.num_agents
cms = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, range(E))
eng = engine.BatchedMettaGrid(prog, cms, np.arange(E, dtype=np.uint32), device=0, buffers="device")
lib = engine.load_lib()
n = len(prog.action_names)
gen = torch.Generator(device="cuda").manual_seed(42)
pa = torch.randint(0, n, (8, E * A), dtype=torch.int32, device="cuda", generator=gen)
pv = torch.randint(0, n, (8, E * A), dtype=torch.int32, device="cuda", generator=gen)
ext = torch.cuda.ExternalStream(eng.stream)
out = (C.c_ulonglong * 16)()
out2 = (C.c_ulonglong * 16)()
for t in range(5 + steps):
    if t == 5:
        eng.sync()
        lib.mgx_debug_world_cycles(out, 1)
        lib.mgx_debug_obs_cycles(out2, 1)
    with torch.cuda.stream(ext):
        eng.actions.copy_(pa[t % 8]); eng.vibe_actions.copy_(pv[t % 8])
        eng.step()
eng.sync()
lib.mgx_debug_world_cycles(out, 0)
names = ["stage agents", "shuffle", "stream0 (move/noop)", "stream1 (vibe)", "on_tick", "coverage", "  do_move (in streams)", "  action bookkeeping"]
waves = (E + 63) // 64
for k, nm in enumerate(names):
    print(f"{nm:28s} {out[k] / steps / waves:12.0f} cycles / wave / step")
lib.mgx_debug_obs_cycles(out2, 0)
print("observation kernel, thread 0 of every workgroup:")
for k, nm in [(8, "stage grid/agents"), (9, "object token cache"), (10, "first observer"), (11, "encode 4 agents"),
              (12, "barrier wait"), (13, "visited + token stats"), (14, "rewards")]:
    print(f"{nm:28s} {out2[k] / steps / E:12.0f} cycles / workgroup / step")
