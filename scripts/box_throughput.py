"""Dense policy input three ways at the bench batch (rung 3, 65 536 envs x 16 agents): token path alone, token path +
standalone decode (mgx_decode_obs, float32), fused box output of the observation kernel (float32 / bfloat16).
Usage (GPU box): python scripts/box_throughput.py [steps]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mettagrid_amd import presets  # noqa: E402
from mettagrid_amd.compiler import compile_spec  # noqa: E402
from mettagrid_amd.engine import BatchedMettaGrid  # noqa: E402
from mettagrid_amd.mapgen import random_class_maps  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
E = 65536
prog = compile_spec(presets.rung3_spec(), 32, 32, max_objects=192)
cms = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, range(4096))[np.arange(E) % 4096]
A, C = prog.num_agents, len(prog.feature_norms)
n = len(prog.action_names)
gen = torch.Generator(device="cuda").manual_seed(1)
acts = torch.randint(0, n, (8, E * A), dtype=torch.int32, device="cuda", generator=gen)
out = {"workload": f"rung 3, {E} envs x {A} agents, box [{C}][11][11] per agent, {steps} timed steps"}
for mode in ("tokens", "tokens+decode_f32", "fused_f32", "fused_bf16"):
    eng = BatchedMettaGrid(prog, cms, np.arange(E, dtype=np.uint32), buffers="device")
    box = None
    if mode != "tokens":
        box = torch.empty((E * A, C, 11, 11), dtype=torch.bfloat16 if mode == "fused_bf16" else torch.float32, device="cuda")
    if mode.startswith("fused"):
        eng.set_box_output(box)
    ext = torch.cuda.ExternalStream(eng.stream)

    def one(t):
        with torch.cuda.stream(ext):
            eng.actions.copy_(acts[t % 8]); eng.vibe_actions.copy_(acts[(t + 3) % 8])
            eng.step()
            if mode == "tokens+decode_f32":
                eng.decode_obs(out=box)
    for t in range(10):
        one(t)
    eng.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(steps):
        one(t)
    eng.sync(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out[mode] = {"ms_per_step": dt * 1e3, "agent_steps_per_s": E * A / dt,
                 "output_bytes_per_step": int(box.numel() * box.element_size()) if box is not None else E * A * prog.num_tokens * 3}
    eng.close()
    del eng, box
    torch.cuda.empty_cache()
print(json.dumps(out))
