"""Run-time specialisation of the extended dispatch kernel at BASELINE.json configs[3]'s shape: a rung-4 game whose chest
handler differs from the preset's (so the build-time code does not apply), generic against specialised.
Usage (GPU box): python scripts/jit_rung4.py [envs] [steps]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mettagrid_amd import presets  # noqa: E402
from mettagrid_amd import spec as S  # noqa: E402
from mettagrid_amd.compiler import compile_spec  # noqa: E402
from mettagrid_amd.engine import BatchedMettaGrid  # noqa: E402
from mettagrid_amd.mapgen import random_class_maps  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
spec = presets.rung4_spec()
spec.objects["chest"].on_use = S.Handler([S.ResourceFilter(S.ACTOR, "ore", 2)],
                                         [S.ResourceTransfer(S.ACTOR, S.TARGET, "ore", -2),
                                          S.SetStat("chest.ore", S.InventoryValue("ore"), scope="game", entity=S.TARGET)], "deposit2")
prog = compile_spec(spec, 64, 64, max_objects=presets.RUNG4_MAX_OBJECTS)
cms = random_class_maps(prog, 64, 64, dict(presets.RUNG4_OBJECTS), dict(presets.RUNG4_AGENTS), range(2048))[np.arange(E) % 2048]
A, n = prog.num_agents, len(prog.action_names)
gen = torch.Generator(device="cuda").manual_seed(1)
acts = torch.randint(0, n, (8, E * A), dtype=torch.int32, device="cuda", generator=gen)
out = {}
t0 = time.perf_counter()
for mode in (False, "sync"):
    eng = BatchedMettaGrid(prog, cms, np.arange(E, dtype=np.uint32), specialize=mode)
    if mode:
        out["code_object_ready_after_s"] = time.perf_counter() - t0
    ext = torch.cuda.ExternalStream(eng.stream)
    eng.set_profiling(True)
    seg = {}
    for t in range(20 + steps):
        with torch.cuda.stream(ext):
            eng.actions.copy_(acts[t % 8]); eng.vibe_actions.copy_(acts[(t + 3) % 8])
            eng.step()
        if t >= 20:
            for k, v in eng.step_timing_segments_ms().items():
                seg[k] = seg.get(k, 0.0) + v / steps
    out["specialised" if mode else "generic"] = {"variants": [eng.act_variant, eng.handler_variant], "segments_ms": seg,
                                                 "ms_per_step": sum(seg.values()), "errors": eng.jit_errors}
    eng.close()
print(json.dumps(out))
