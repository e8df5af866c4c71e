"""Where the rung-4 world-update time goes: the action-phase / AoE / tail segments (engine HIP events) of spec variants
with one ingredient removed and of restricted action mixes.  Usage (GPU box): python scripts/rung4_attrib.py [envs]"""
import dataclasses
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mettagrid_amd import presets  # noqa: E402
from mettagrid_amd.compiler import compile_spec  # noqa: E402
from mettagrid_amd.engine import BatchedMettaGrid  # noqa: E402
from mettagrid_amd.mapgen import random_class_maps  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
base = presets.rung4_spec()


def variant(name):
    s = presets.rung4_spec()
    if name == "no_on_tick":
        s = dataclasses.replace(s, agents=[dataclasses.replace(a, on_tick=None) for a in s.agents])
    elif name == "no_events":
        s = dataclasses.replace(s, events={})
    elif name == "no_events_no_matq":
        s = dataclasses.replace(s, events={}, materialize_queries=[])
    elif name == "no_mobile_aoe":
        s = dataclasses.replace(s, agents=[dataclasses.replace(a, aoes=[]) for a in s.agents])
    return s


def run(name, actions="random"):
    spec = variant(name)
    prog = compile_spec(spec, 64, 64, max_objects=presets.RUNG4_MAX_OBJECTS)
    cms = random_class_maps(prog, 64, 64, dict(presets.RUNG4_OBJECTS), dict(presets.RUNG4_AGENTS), range(E))
    eng = BatchedMettaGrid(prog, cms, np.arange(E, dtype=np.uint32), buffers="device")
    n = len(prog.action_names)
    A = prog.num_agents
    gen = torch.Generator(device="cuda").manual_seed(42)
    if actions == "random":
        pa = torch.randint(0, n, (8, E * A), dtype=torch.int32, device="cuda", generator=gen)
    elif actions == "noop":
        pa = torch.full((8, E * A), prog.action_names.index("noop"), dtype=torch.int32, device="cuda")
    else:  # moves only
        ids = torch.tensor([i for i, a in enumerate(prog.action_names) if a.startswith("move")], dtype=torch.int32, device="cuda")
        pa = ids[torch.randint(0, len(ids), (8, E * A), device="cuda", generator=gen)]
    ext = torch.cuda.ExternalStream(eng.stream)
    eng.set_profiling(True)
    seg = {}
    steps, warm = 20, 5
    for t in range(warm + steps):
        with torch.cuda.stream(ext):
            eng.actions.copy_(pa[t % 8]); eng.vibe_actions.copy_(pa[(t + 3) % 8])
            eng.step()
        s = eng.step_timing_segments_ms()
        if t >= warm:
            for k, v in s.items():
                seg[k] = seg.get(k, 0.0) + v / steps
    bits, first = eng.poll_errors()
    print(f"{name:20s} actions={actions:7s} " + " ".join(f"{k}={v:7.3f}" for k, v in seg.items()) + (f"  ERR {bits}" if bits else ""), flush=True)
    eng.close()


run("full")
run("full", "noop")
run("full", "move")
for v in ("no_on_tick", "no_events", "no_events_no_matq", "no_mobile_aoe"):
    run(v)
