"""How many observation tokens does the rung-4 workload need?  Steps it with a generous budget and records the largest
number of tokens any agent's observation held.  Usage (GPU box): python scripts/token_budget.py [envs] [steps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mettagrid_amd import presets  # noqa: E402
from mettagrid_amd.compiler import compile_spec  # noqa: E402
from mettagrid_amd.engine import BatchedMettaGrid  # noqa: E402
from mettagrid_amd.mapgen import random_class_maps  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
prog = compile_spec(presets.rung4_spec(obs_tokens=448), 64, 64, max_objects=presets.RUNG4_MAX_OBJECTS)
cms = random_class_maps(prog, 64, 64, dict(presets.RUNG4_OBJECTS), dict(presets.RUNG4_AGENTS), range(E))
eng = BatchedMettaGrid(prog, cms, np.arange(E, dtype=np.uint32), buffers="device")
n = len(prog.action_names)
gen = torch.Generator(device="cuda").manual_seed(42)
worst = 0
hist = []
for t in range(steps):
    eng.actions.copy_(torch.randint(0, n, eng.actions.shape, dtype=torch.int32, device="cuda", generator=gen))
    eng.vibe_actions.copy_(torch.randint(0, n, eng.actions.shape, dtype=torch.int32, device="cuda", generator=gen))
    eng.wait_for_caller()
    eng.step()
    eng.caller_waits()
    if t % 10 == 9:
        m = int((eng.obs[:, :, 0] != 255).sum(1).max())
        worst = max(worst, m)
        hist.append(m)
print({"envs": E, "steps": steps, "max_tokens_in_any_observation": worst, "every_10_steps": hist[-10:], "errors": eng.poll_errors()})
