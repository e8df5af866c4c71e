"""Throughput of the policy-side token decode (mgx_decode_obs): rung-3 observations of 8 192 envs
(131 072 agent rows x 200 tokens) -> dense f32 box [rows, C, 11, 11].  The kernel is an HBM write stream: 4*C*H*W bytes
per row out, 3*T in.  Usage (GPU box): python scripts/decode_throughput.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mettagrid_amd import presets  # noqa: E402
from mettagrid_amd.compiler import compile_spec  # noqa: E402
from mettagrid_amd.engine import BatchedMettaGrid  # noqa: E402
from mettagrid_amd.mapgen import random_class_maps  # noqa: E402

E = 8192
prog = compile_spec(presets.rung3_spec(), 32, 32, max_objects=192)
cms = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, range(E))
eng = BatchedMettaGrid(prog, cms, np.arange(E, dtype=np.uint32), buffers="device")
n = len(prog.action_names)
for t in range(20):   # some history in the observations
    eng.actions.copy_(torch.randint(0, n, eng.actions.shape, dtype=torch.int32, device="cuda"))
    eng.step()
eng.sync()
box = eng.decode_obs()
ext = torch.cuda.ExternalStream(eng.stream)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
for _ in range(5):
    eng.decode_obs(out=box)
ev0.record(ext)
for _ in range(reps):
    eng.decode_obs(out=box)
ev1.record(ext)
eng.sync()
ms = ev0.elapsed_time(ev1) / reps
rows = box.shape[0]
out_bytes, in_bytes = box.numel() * 4, rows * prog.num_tokens * 3
print(json.dumps({"kernel": "mgx_decode_kernel", "rows": rows, "box_shape": list(box.shape), "ms_per_launch": ms,
                  "rows_per_s": rows / (ms * 1e-3), "GBps": (out_bytes + in_bytes) / (ms * 1e-3) / 1e9,
                  "frac_of_8TBps": (out_bytes + in_bytes) / (ms * 1e-3) / 8e12}))
