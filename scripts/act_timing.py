"""Per-phase shader-clock breakdown of the lane-per-agent action kernel (needs scripts/build_timing.sh).
Usage: MGX_LIB=mettagrid_amd/libmgx_timing.so python scripts/act_timing.py [rung] [steps]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mettagrid_amd import engine, presets  # noqa: E402
from mettagrid_amd.compiler import compile_spec  # noqa: E402
from mettagrid_amd.mapgen import random_class_maps  # noqa: E402

rung = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
E = 65536
if rung == 3:
    prog = compile_spec(presets.rung3_spec(), 32, 32, max_objects=192)
    maps = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, range(4096))
    epg, fn = 16, "mgx_debug_act_fast_cycles"
else:
    prog = compile_spec(presets.rung4_spec(), 64, 64, max_objects=presets.RUNG4_MAX_OBJECTS)
    maps = random_class_maps(prog, 64, 64, presets.RUNG4_OBJECTS, presets.RUNG4_AGENTS, range(1024))
    epg, fn = 4, "mgx_debug_act_x_cycles"
cms = maps[np.arange(E) % len(maps)]
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
        getattr(lib, fn)(out, 1)
    with torch.cuda.stream(ext):
        eng.actions.copy_(pa[t % 8]); eng.vibe_actions.copy_(pv[t % 8])
        eng.step()
eng.sync()
getattr(lib, fn)(out, 0)
wgs = (E + epg - 1) // epg
for k, nm in enumerate(["stage", "shuffle", "primary stream (rounds)", "vibe stream", "on_tick", "bookkeeping + coverage"]):
    print(f"{nm:28s} {out[k] / steps / wgs:10.0f} cycles / workgroup / step")
print(f"{'  do_move (in streams)':28s} {out[6] / steps / wgs:10.0f}")
print(f"{'  conflict ordering':28s} {out[9] / steps / wgs:10.0f}")
print(f"{'  dispatch (both streams)':28s} {out[10] / steps / wgs:10.0f}")
print(f"rounds per wavefront: {out[8] / steps / wgs:.2f}")
