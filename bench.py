#!/usr/bin/env python3
"""Headline benchmark: agent-steps/s of the MI355X MettaGrid step engine (BASELINE.json metric).

Workload (N=1): BASELINE.json configs[2] — 65 536 envs x 32x32 random maps x 16 agents (2 teams), 8-direction move +
4 vibes, full attack/on-use handler chains, tokenised 11x11 observations with T=200 (SURVEY.md §8d rung 3).
One "step" = one tick of all envs (world-update kernel + observation/reward kernel).  Inputs (actions) are resident
in HBM before the timed region.  N>1: one process per GPU (torch.distributed / RCCL), envs sharded by index, every
rank steps its own 65 536 envs (weak scaling); the path has no exchange step so no data-path collective is issued
unless --gather is given.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def cpu_baseline(spec, prog, steps: int) -> dict:
    """Reference C++ engine (oracle/_ref, kind "reference") or the CPU restatement (kind "port") on ONE env of the same
    workload on one host core — a reported baseline, never the measured product."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from mettagrid_amd import presets
    cells = presets.rung3_map(0)
    rng = np.random.RandomState(42)
    n, A = len(prog.action_names), prog.num_agents
    acts = rng.randint(0, n, (steps, A)).astype(np.int32)
    vibes = rng.randint(0, n, (steps, A)).astype(np.int32)
    try:
        import ref_driver
        if not ref_driver.ref_available():
            raise RuntimeError("no _ref")
        sim = ref_driver.RefSim(spec, cells, 42, prog)
        kind = "reference"

        def step(t):
            sim.A[:] = acts[t]
            sim.VA[:] = vibes[t]
            sim.c.step()
    except Exception:
        import oracle_py
        sim = oracle_py.OracleSim(prog, prog.class_map(cells), 42)
        kind = "port"

        def step(t):
            sim.step(acts[t], vibes[t])
    for t in range(min(500, steps)):
        step(t)
    t0 = time.perf_counter()
    for t in range(steps):
        step(t)
    dt = time.perf_counter() - t0
    return {"value": steps * A / dt, "unit": "agent-steps/s", "cores": 1, "kind": kind,
            "sample": f"1 env (map seed 0) of the same rung-3 workload, {steps} steps, random actions, 1 host core"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--gather", choices=["none", "scalars", "obs"], default="none",
                    help="optional per-step RCCL all_gather of rewards/terminals/truncations (+obs)")
    ap.add_argument("--cpu-steps", type=int, default=300000, help="CPU baseline sample: about 10 s of one host core")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the step engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from mettagrid_amd import presets
    from mettagrid_amd.compiler import compile_spec
    from mettagrid_amd.engine import BatchedMettaGrid
    from mettagrid_amd.mapgen import random_class_maps

    spec = presets.rung3_spec()
    H = W = 32
    prog = compile_spec(spec, H, W, max_objects=192)
    E, A, T = args.envs, prog.num_agents, prog.num_tokens
    env0 = rank * E
    cms = random_class_maps(prog, H, W, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8},
                            range(env0, env0 + E))
    seeds = np.arange(env0, env0 + E, dtype=np.uint32)
    eng = BatchedMettaGrid(prog, cms, seeds, device=local_rank, buffers="device")
    del cms

    # pre-generated actions resident in HBM (protocol of python/src/mettagrid/perf/harness.py:32-34)
    n_actions = len(prog.action_names)
    cycle = 8
    gen = torch.Generator(device="cuda").manual_seed(42 + rank)
    pre_a = torch.randint(0, n_actions, (cycle, E * A), dtype=torch.int32, device="cuda", generator=gen)
    pre_v = torch.randint(0, n_actions, (cycle, E * A), dtype=torch.int32, device="cuda", generator=gen)
    ext = torch.cuda.ExternalStream(eng.stream, device=torch.device("cuda", local_rank))
    torch.cuda.synchronize()

    gathered = None
    if world > 1 and args.gather != "none":
        gathered = [torch.empty(world * E * A, dtype=torch.float32, device="cuda"),
                    torch.empty(world * E * A, dtype=torch.bool, device="cuda"),
                    torch.empty(world * E * A, dtype=torch.bool, device="cuda")]
        if args.gather == "obs":
            gathered.append(torch.empty((world * E * A, T, 3), dtype=torch.uint8, device="cuda"))

    def one_step(t: int) -> None:
        with torch.cuda.stream(ext):  # everything is ordered on the engine's stream
            eng.actions.copy_(pre_a[t % cycle], non_blocking=True)
            eng.vibe_actions.copy_(pre_v[t % cycle], non_blocking=True)
            eng.step()
            if gathered is not None:
                dist.all_gather_into_tensor(gathered[0], eng.rewards)
                dist.all_gather_into_tensor(gathered[1], eng.terminals)
                dist.all_gather_into_tensor(gathered[2], eng.truncations)
                if args.gather == "obs":
                    dist.all_gather_into_tensor(gathered[3], eng.obs)

    for t in range(args.warmup):
        one_step(t)
    eng.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    with torch.cuda.stream(ext):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(ext)
    t0 = time.perf_counter()
    for t in range(args.steps):
        one_step(args.warmup + t)
    with torch.cuda.stream(ext):
        ev1.record(ext)
    eng.sync()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if dist is not None:
        dist.barrier()
        tmax = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall = float(tmax.item())
    bits, first = eng.poll_errors()
    if bits:
        raise SystemExit(f"engine reported env error bits {bits} (first env {first})")

    # per-kernel durations for the roofline line: HIP events between the two kernels on the engine stream
    eng.set_profiling(True)
    k_world, k_obs, nprof = 0.0, 0.0, min(50, max(5, args.steps))
    for t in range(nprof):
        one_step(t)
        w_ms, o_ms = eng.step_timing_ms()
        k_world += w_ms
        k_obs += o_ms
    eng.set_profiling(False)
    k_world /= nprof
    k_obs /= nprof

    if rank == 0:
        agent_steps = world * E * A * args.steps
        value = agent_steps / wall
        # algorithmic bytes of the dominant (observation) kernel per agent-step, DESIGN.md "Roofline accounting":
        # obs out 3T + reward 4 + terminal 1 + truncation 1 + one read of the env state S_env/A.
        s_env = 2 * H * W + A * 87 + 12 * 32 + 8
        bytes_per_agent_step = 3 * T + 6 + s_env / A
        achieved = E * A * bytes_per_agent_step / (k_obs * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("mgx_obs_kernel_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "agent-steps/sec (whole node) at 65 536 envs, 32x32x16-agent; HBM GB/s vs peak",
            "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "rung3: 65536 envs/GPU x 32x32 random map x 16 agents, 8-dir move + 4 vibes, "
                                   "attack/on-use handler chains, 11x11 token obs T=200",
                       "envs_per_gpu": E, "agents_per_env": A, "obs_tokens": T, "gather": args.gather,
                       "parallelism": f"env-shard x{world}"},
            "device_ms_per_step": dev_ms / args.steps,
            "kernels_ms": {"mgx_world_kernel": k_world, "mgx_obs_kernel": k_obs},
            "roofline": {"bound": "hbm", "kernel": "mgx_obs_kernel", "achieved": achieved, "peak": 8000.0,
                         "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
                         "bytes_per_agent_step": bytes_per_agent_step},
        }
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(spec, prog, args.cpu_steps)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
