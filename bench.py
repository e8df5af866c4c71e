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


def workload(rung: int):
    """(spec, H, W, max_objects, object counts, agent counts, map factory, description, algorithmic bytes per agent-step
    of the whole tick / of the observation kernel / of the world update) for BASELINE.json configs[rung - 1]."""
    from mettagrid_amd import presets
    if rung == 3:
        spec, H, W, S = presets.rung3_spec(), 32, 32, 192
        objs, agents, mapf = {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, presets.rung3_map
        desc = ("rung3: 65536 envs/GPU x 32x32 random map x 16 agents, 8-dir move + 4 vibes, attack/on-use handler "
                "chains, 11x11 token obs T=200")
        A, T = 16, spec.obs.num_tokens
        s_env = 2 * H * W + A * 87 + 12 * 32 + 8
    elif rung == 4:
        spec, H, W, S = presets.rung4_spec(), 64, 64, presets.RUNG4_MAX_OBJECTS
        objs, agents, mapf = dict(presets.RUNG4_OBJECTS), dict(presets.RUNG4_AGENTS), presets.rung4_map
        desc = ("rung4: 65536 envs/GPU x 64x64 random map x 64 agents (4 teams), rung-3 rules + 16 static AoE + mobile "
                "AoE per agent + territory (8 sources) + aoe_mask obs + 3 events + materialized closure query, T=256")
        A, T = 64, spec.obs.num_tokens
        R, n_rw, n_dyn = len(spec.resource_names), 3, sum(v for k, v in objs.items() if k != "wall")
        s_agent = 4 + 1 + 32 + 2 * R + R + 16 + 4 * n_rw             # SURVEY.md §8d S_agent
        s_env = 2 * H * W + A * s_agent + n_dyn * 32 + 8
    else:
        raise SystemExit("--rung must be 3 or 4")
    per = {"tick": 3 * T + 14 + 2 * s_env / A, "obs": 3 * T + 6 + s_env / A, "world": 8 + s_env / A}
    return spec, H, W, S, objs, agents, mapf, desc, per


def cpu_baseline(spec, prog, steps: int, cells) -> dict:
    """Reference C++ engine (oracle/_ref, kind "reference") or the CPU restatement (kind "port") on ONE env of the same
    workload on one host core — a reported baseline, never the measured product."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    rng = np.random.RandomState(42)
    n, A = len(prog.action_names), prog.num_agents
    acts = rng.randint(0, n, (steps, A)).astype(np.int32)
    vibes = rng.randint(0, n, (steps, A)).astype(np.int32)
    kind, step = _cpu_sim(spec, prog, cells, 42, acts, vibes)
    for t in range(min(500, steps)):
        step(t)
    t0 = time.perf_counter()
    for t in range(steps):
        step(t)
    dt = time.perf_counter() - t0
    return {"value": steps * A / dt, "unit": "agent-steps/s", "cores": 1, "kind": kind,
            "sample": f"1 env (map seed 0) of the same workload, {steps} steps, random actions, 1 host core"}


def cpu_worker(rung: int, steps: int, seed: int, start_at: float) -> None:
    """One process of the whole-host baseline: its own env (map seed = worker index), starts stepping at ``start_at``."""
    from mettagrid_amd.compiler import compile_spec
    spec, H, W, S, objs, agents, mapf, desc, per = workload(rung)
    prog = compile_spec(spec, H, W, max_objects=S)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    rng = np.random.RandomState(seed)
    n, A = len(prog.action_names), prog.num_agents
    acts = rng.randint(0, n, (steps, A)).astype(np.int32)
    vibes = rng.randint(0, n, (steps, A)).astype(np.int32)
    kind, step = _cpu_sim(spec, prog, mapf(seed), seed, acts, vibes)
    for t in range(min(200, steps)):
        step(t)
    while time.time() < start_at:
        time.sleep(0.005)
    t0 = time.time()
    for t in range(steps):
        step(t)
    print(json.dumps({"kind": kind, "agent_steps": steps * A, "t0": t0, "t1": time.time()}), flush=True)


def host_cores() -> int:
    """Cores this process may really use: the affinity mask cut down to the cgroup CPU quota, and to 16 — the CPU share
    of one GPU on the bench boxes — unless MGX_BENCH_CORES says otherwise."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return int(os.environ.get("MGX_BENCH_CORES", min(n, 16)))


def cpu_baseline_host(rung: int, steps: int) -> dict:
    """The same CPU engine on every host core at once: one process per core, one env each, started together.  Child
    processes are started before this process touches the GPU."""
    import subprocess
    cores = host_cores()
    start_at = time.time() + (20.0 if rung == 4 else 12.0)   # imports + map + program compile of every worker
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(rung), str(steps), str(i),
                               repr(start_at)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for i in range(cores)]
    outs = []
    deadline = time.time() + 150
    for p in procs:
        try:
            o, _ = p.communicate(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.communicate()
            continue
        if p.returncode == 0 and o.strip():
            outs.append(json.loads(o.strip().splitlines()[-1]))
    if not outs:
        return {"value": None, "cores": cores, "error": "no worker finished"}
    late = sum(1 for o in outs if o["t0"] > start_at + 0.5)
    span = max(o["t1"] for o in outs) - min(o["t0"] for o in outs)
    return {"value": sum(o["agent_steps"] for o in outs) / span, "cores": len(outs), "kind": outs[0]["kind"],
            "late_starters": late}


def _cpu_sim(spec, prog, cells, seed, acts, vibes):
    """(kind, step(t)) of the reference C++ engine (oracle/_ref) or, where that build is absent, the CPU restatement."""
    try:
        import ref_driver
        if not ref_driver.ref_available():
            raise RuntimeError("no _ref")
        sim = ref_driver.RefSim(spec, cells, seed, prog)

        def step(t):
            sim.A[:] = acts[t]
            sim.VA[:] = vibes[t]
            sim.c.step()
        return "reference", step
    except Exception:
        import oracle_py
        sim = oracle_py.OracleSim(prog, prog.class_map(cells), seed)

        def step(t):
            sim.step(acts[t], vibes[t])
        return "port", step


def main() -> None:
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        cpu_worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--rung", type=int, default=3, help="3 = BASELINE.json configs[2] (the headline metric), 4 = configs[3]")
    ap.add_argument("--gather", choices=["none", "scalars", "obs"], default="none",
                    help="optional per-step gather of rewards/terminals/truncations (+obs) to rank 0: grouped RCCL "
                         "send/recv over xGMI on a side stream, overlapped with the next step (mettagrid_amd/dist.py)")
    ap.add_argument("--groups", type=int, default=1,
                    help="env groups per GPU (mettagrid_amd/groups.py): 2 = the world update of one half of the envs runs "
                         "beside the observation kernel of the other half (measured slower on MI355X: DESIGN.md); "
                         "1 = one engine, kernels back to back")
    ap.add_argument("--cpu-steps", type=int, default=0, help="CPU baseline sample (default: about 10 s of one host core)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    cpu = None
    if not args.no_cpu and world == 1:
        # CPU baseline first: its worker processes are started before this process initialises the GPU.  One core:
        # about 10 s; then every host core at once, one env per process, about 5 s of stepping each.
        from mettagrid_amd.compiler import compile_spec as _cs
        spec_c, H_c, W_c, S_c, _, _, mapf_c, _, _ = workload(args.rung)
        prog_c = _cs(spec_c, H_c, W_c, max_objects=S_c)
        cpu_steps = args.cpu_steps or (300000 if args.rung == 3 else 25000)
        one = cpu_baseline(spec_c, prog_c, cpu_steps, mapf_c(0))
        host = cpu_baseline_host(args.rung, max(1000, cpu_steps // 2))
        if host.get("value"):
            cpu = {"value": host["value"], "unit": "agent-steps/s", "cores": host["cores"], "kind": host["kind"],
                   "sample": f"{host['cores']} processes (one per host core), 1 env each (map seed = process index) of the "
                             f"same workload, {max(1000, cpu_steps // 2)} steps each, random actions, started together",
                   "one_core": one}
        else:
            cpu = one
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the step engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from mettagrid_amd.compiler import compile_spec
    from mettagrid_amd.dist import GatherToRoot, env_shard, shard_seeds
    from mettagrid_amd.groups import EnvGroups
    from mettagrid_amd.mapgen import random_class_maps

    spec, H, W, S, objs, agents, mapf, desc, per = workload(args.rung)
    prog = compile_spec(spec, H, W, max_objects=S)
    E, A, T = args.envs, prog.num_agents, prog.num_tokens
    cms = random_class_maps(prog, H, W, objs, agents, env_shard(rank, world, E))   # map seed = global env index
    seeds = shard_seeds(rank, world, E)
    grp = EnvGroups(prog, cms, seeds, device=local_rank, groups=args.groups)
    del cms
    G = args.groups

    # pre-generated actions resident in HBM (protocol of python/src/mettagrid/perf/harness.py:32-34)
    n_actions = len(prog.action_names)
    cycle = 8
    gen = torch.Generator(device="cuda").manual_seed(42 + rank)
    pre_a = torch.randint(0, n_actions, (cycle, E * A), dtype=torch.int32, device="cuda", generator=gen)
    pre_v = torch.randint(0, n_actions, (cycle, E * A), dtype=torch.int32, device="cuda", generator=gen)
    exts = [torch.cuda.ExternalStream(eng.stream, device=torch.device("cuda", local_rank)) for eng in grp.engines]
    ext = exts[-1]   # the last group finishes a step last
    torch.cuda.synchronize()

    gather = None
    if world > 1 and args.gather != "none":
        gather = GatherToRoot(dist, root=0, device=torch.device("cuda", local_rank), producer_stream=ext)

    def one_step(t: int, only=None) -> None:
        for g, eng in enumerate(grp.engines):
            if only is not None and g != only:
                continue
            rows = grp.rows(g)
            with torch.cuda.stream(exts[g]):  # everything of a group is ordered on its engine's stream
                eng.actions.copy_(pre_a[t % cycle][rows], non_blocking=True)
                eng.vibe_actions.copy_(pre_v[t % cycle][rows], non_blocking=True)
                eng.step()
        if gather is not None:   # staging copy + grouped send/recv on the side stream; step t+1 is not held back
            if G > 1:            # the gather reads all groups' rows: order the last stream behind the others
                for g in range(G - 1):
                    ev = torch.cuda.Event()
                    ev.record(exts[g])
                    ext.wait_event(ev)
            out = {"rewards": grp.rewards, "terminals": grp.terminals, "truncations": grp.truncations}
            if args.gather == "obs":
                out["observations"] = grp.obs
            gather.submit(out)

    for t in range(args.warmup):
        one_step(t)
    grp.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(exts[0])      # (all streams are idle here; the first group starts a step first ...)
    t0 = time.perf_counter()
    for t in range(args.steps):
        one_step(args.warmup + t)
    ev1.record(ext)          # (... and the last group ends it)
    grp.sync()
    if gather is not None:
        gather.finish()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if dist is not None:
        dist.barrier()
        tmax = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall = float(tmax.item())
    bits, first = grp.poll_errors()
    if bits:
        raise SystemExit(f"engine reported env error bits {bits} (first env {first})")

    # per-kernel durations for the roofline line: HIP events between the two kernels on the engine stream
    # (one group at a time, so that a duration is the kernel's own and not its share of an overlapped pair; a launch
    # covers E / G envs)
    grp.set_profiling(True)
    nprof = min(50, max(5, args.steps))
    seg = {}
    for t in range(nprof):
        for g, eng in enumerate(grp.engines):
            one_step(t, only=g)
            for k, v in eng.step_timing_segments_ms().items():   # waits for the step
                seg[k] = seg.get(k, 0.0) + v / (nprof * G)
    grp.set_profiling(False)

    if rank == 0:
        agent_steps = world * E * A * args.steps
        value = agent_steps / wall
        # Roofline of the dominant kernel (DESIGN.md "Roofline accounting").  Algorithmic bytes per agent-step: the
        # observation kernel owns obs out 3T + reward 4 + terminal 1 + truncation 1 + one read of the env state
        # S_env/A; the world-update kernel owns the two action streams (8) + one pass over S_env/A.
        names = {"obs": "mgx_obs_kernel", "actions": "mgx_world_kernel_x" if args.rung == 4 else "mgx_world_kernel_fast"}
        dom = "obs" if seg["obs"] >= seg["actions"] else "actions"
        bytes_per_agent_step = per["obs"] if dom == "obs" else per["world"]
        achieved = (E // G) * A * bytes_per_agent_step / (seg[dom] * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(f"rung{args.rung}", {}).get(names[dom] + "_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "agent-steps/sec (whole node) at 65 536 envs, 32x32x16-agent; HBM GB/s vs peak",
            "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": desc, "baseline_config": f"configs[{args.rung - 1}]",
                       "envs_per_gpu": E, "agents_per_env": A, "obs_tokens": T, "gather": args.gather,
                       "parallelism": f"env-shard x{world}",
                       "env_groups_per_gpu": G, "envs_per_launch": E // G},
            "device_ms_per_step": dev_ms / args.steps,
            "kernels_ms": {"world_actions": seg["actions"], "aoe": seg["aoe"], "world_tail": seg["tail"],
                           "mgx_obs_kernel": seg["obs"], "rewards_ext": seg["rewards"]},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": 8000.0,
                         "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
                         "bytes_per_agent_step": bytes_per_agent_step,
                         "tick_bytes_per_agent_step": per["tick"]},
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
