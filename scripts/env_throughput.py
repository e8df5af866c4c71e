"""Agent-steps/s through the PufferEnv-shaped wrapper (MettaGridBatchedEnv): rung-3 rules, 65 536 envs, max_steps=1000,
a device map pool of 256 maps with on-device auto-reset and desynchronised first episodes, joint action ids (core +
vibe) decoded on the device — everything the reference's MettaGridPufferEnv.step does around MettaGrid::step
(python/src/mettagrid/envs/mettagrid_puffer_env.py:287-330).  Usage (GPU box): python scripts/env_throughput.py [envs] [steps]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mettagrid_amd import presets  # noqa: E402
from mettagrid_amd.compiler import compile_spec  # noqa: E402
from mettagrid_amd.envs import MettaGridBatchedEnv  # noqa: E402
from mettagrid_amd.mapgen import random_class_maps  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
RUNG = int(os.environ.get("MGX_ENV_RUNG", "3"))   # 4: the rung-4 preset (64 x 64, 64 agents, area effects / territory / events / queries)
if RUNG == 4:
    spec = presets.rung4_spec()
    spec.max_steps = 1000
    prog = compile_spec(spec, 64, 64, max_objects=presets.RUNG4_MAX_OBJECTS)
    pool = random_class_maps(prog, 64, 64, dict(presets.RUNG4_OBJECTS), dict(presets.RUNG4_AGENTS), range(256))
else:
    spec = presets.rung3_spec()
    spec.max_steps = 1000
    prog = compile_spec(spec, 32, 32, max_objects=192)
    pool = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, range(256))
out = {}
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None   # e.g. "unchecked_actions" (profiling runs)
for validate, stats in ((False, True), (False, False), (True, True)):
    key = ("validate_actions" if validate else "unchecked_actions") + ("" if stats else "_no_episode_stats")
    if only and key not in only:
        continue
    env = MettaGridBatchedEnv(prog, E, map_pool=pool, pool_stride=7, desync=True, validate_actions=validate, seed=1, episode_stats=stats,
                              specialize=os.environ.get("MGX_ENV_SPECIALIZE", "auto"))
    obs, _ = env.reset()
    n = env.transport_action_n   # joint ids: core actions, then (core, vibe) pairs
    gen = torch.Generator(device="cuda").manual_seed(3)
    acts = torch.randint(0, int(n), (8, env.num_agents), dtype=torch.int32, device="cuda", generator=gen)
    for t in range(50):
        env.step(acts[t % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_infos = n_eps = 0
    for t in range(steps):
        obs, rew, term, trunc, infos = env.step(acts[t % 8])
        if infos:
            n_infos += 1
            n_eps += infos["episodes"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ep, _ = env.engine.episodes()
    bits, first = env.engine.poll_errors()
    out[key + "_variants"] = [env.engine.obs_variant, env.engine.handler_variant]
    out[key] = {"agent_steps_per_s": env.num_agents * steps / dt, "ms_per_step": dt * 1e3 / steps,
                "episodes_finished": int(ep.sum()), "env_error_bits": int(bits), "infos_returned": n_infos, "episodes_in_infos": n_eps}
    env.close()
print(json.dumps({"workload": f"rung{RUNG} rules through MettaGridBatchedEnv, {E} envs x {prog.num_agents} agents, max_steps=1000, pool of 256 maps, "
                              f"desync, {steps} timed steps", **out}))
