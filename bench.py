#!/usr/bin/env python3
"""Headline benchmark: agent-steps/s of the MI355X MettaGrid step engine (BASELINE.json metric).

Workload (N=1): BASELINE.json configs[2] — 65 536 envs x 32x32 random maps x 16 agents (2 teams), 8-direction move +
4 vibes, full attack/on-use handler chains, tokenised 11x11 observations with T=200 (SURVEY.md §8d rung 3).
One "step" = one tick of all envs (world-update kernel + observation/reward kernel).  Inputs (actions) are resident
in HBM before the timed region.  N>1: one process per GPU (torch.distributed / RCCL), envs sharded by index, every
rank steps its own 65 536 envs (weak scaling); the path has no exchange step so no data-path collective is issued
unless --gather is given.

Protocol (python/src/mettagrid/perf/harness.py:100-209 of the reference): pre-generated actions, warm-up, then the K timed
steps as R rounds of K/R steps; the whole region is ONE bracket (barrier + synchronize on both sides, `value` comes
from it) and the round boundaries are HIP events on the engine's stream, so the per-round mean / sigma / CV cost no host
synchronisation; CV > 20 % is flagged like benchmarks/perf/perf_benchmark.py:216-218.

The default N=1 run also measures, after the headline and outside its bracket: BASELINE.json configs[3] (rung 4) as the
object "rung4" (same fields, short run) and the throughput a trainer sees through the PufferEnv-shaped wrapper
("env_wrapper").  Prints ONE JSON line on rank 0.
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


# ---- host-side map construction (before the GPU is touched: the workers are forked) ---------------------------------
def _maps_chunk(job):
    rung, lo, hi = job
    from mettagrid_amd.compiler import compile_spec
    from mettagrid_amd.mapgen import random_class_maps
    spec, H, W, S, objs, agents, _, _, _ = workload(rung)
    prog = compile_spec(spec, H, W, max_objects=S)
    return random_class_maps(prog, H, W, objs, agents, range(lo, hi))


def build_maps(rung: int, envs: range, workers: int) -> np.ndarray:
    """uint16 [E][H][W] class maps of the global envs ``envs`` (map seed = global env index), built by ``workers`` forked
    processes: the reference's RandomMapBuilder shuffle is one numpy generator per map (64 s on one core for rung 4)."""
    n = len(envs)
    workers = max(1, min(workers, n // 256 or 1))
    if workers == 1:
        return _maps_chunk((rung, envs.start, envs.stop))
    import multiprocessing as mp
    step = (n + workers * 4 - 1) // (workers * 4)
    jobs = [(rung, lo, min(lo + step, envs.stop)) for lo in range(envs.start, envs.stop, step)]
    with mp.get_context("fork").Pool(workers) as pool:
        return np.concatenate(pool.map(_maps_chunk, jobs))


# ---- CPU baseline (the reference engine on the host cores; never the measured product) -------------------------------
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


def cpu_leg(rung: int, cpu_steps: int) -> dict:
    """One core (about cpu_steps steps), then every host core at once, for one rung."""
    from mettagrid_amd.compiler import compile_spec
    spec, H, W, S, _, _, mapf, _, _ = workload(rung)
    prog = compile_spec(spec, H, W, max_objects=S)
    one = cpu_baseline(spec, prog, cpu_steps, mapf(0))
    host = cpu_baseline_host(rung, max(1000, cpu_steps // 2))
    if host.get("value"):
        return {"value": host["value"], "unit": "agent-steps/s", "cores": host["cores"], "kind": host["kind"],
                "sample": f"{host['cores']} processes (one per host core), 1 env each (map seed = process index) of the "
                          f"same workload, {max(1000, cpu_steps // 2)} steps each, random actions, started together",
                "one_core": one}
    return one


# ---- the GPU measurement of one rung ---------------------------------------------------------------------------------
def measure(rung: int, cms: np.ndarray, *, envs: int, steps: int, warmup: int, rounds: int, groups: int, gather_kind: str,
            rank: int, local_rank: int, world: int, dist, spec=None, generic: bool = False, specialize="auto") -> dict:
    """spec: another game on the rung's maps (the run-time specialisation leg); generic: the kernels any program gets — no
    build-time preset instance, no generated handlers (MGX_NO_GEN / MGX_OBS_GENERIC at mgx_create), no run-time code objects."""
    import torch
    from mettagrid_amd.compiler import compile_spec
    from mettagrid_amd.dist import GatherToRoot, shard_seeds
    from mettagrid_amd.groups import EnvGroups

    spec0, H, W, S, objs, agents, mapf, desc, per = workload(rung)
    prog = compile_spec(spec if spec is not None else spec0, H, W, max_objects=S)
    E, A, T = envs, prog.num_agents, prog.num_tokens
    seeds = shard_seeds(rank, world, E)
    saved = {k: os.environ.get(k) for k in ("MGX_NO_GEN", "MGX_OBS_GENERIC")}
    if generic:
        os.environ.update(MGX_NO_GEN="1", MGX_OBS_GENERIC="1")
    try:
        grp = EnvGroups(prog, cms, seeds, device=local_rank, groups=groups, specialize=False if generic else specialize)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    variants = {"obs": grp.engines[0].obs_variant, "act": grp.engines[0].act_variant, "handler": grp.engines[0].handler_variant,
                "pairs": grp.engines[0].dispatch_pairs, "int_book": grp.engines[0].integer_bookkeeping}
    G = groups

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
    if world > 1 and gather_kind != "none":
        # the step that next writes the gathered buffers is ordered behind the staging copy: only its observation kernel
        # waits (mgx_wait_before_outputs), the world update runs beside the copy
        gather = GatherToRoot(dist, root=0, device=torch.device("cuda", local_rank), producer_stream=ext,
                              packed=("observations",) if gather_kind == "obs-packed" else (),
                              output_fence=grp.wait_before_outputs)

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
            if gather_kind in ("obs", "obs-packed"):
                out["observations"] = grp.obs
            gather.submit(out)

    for t in range(warmup):
        one_step(t)
    grp.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    # round boundaries: events on the stream that finishes a step last (no host synchronisation inside the bracket)
    rounds = max(1, min(rounds, steps))
    bounds = [round(i * steps / rounds) for i in range(rounds + 1)]
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(rounds + 1)]
    ev0 = torch.cuda.Event(enable_timing=True)
    ev0.record(exts[0])      # (all streams are idle here; the first group starts a step first ...)
    marks[0].record(ext)
    t0 = time.perf_counter()
    r = 1
    for t in range(steps):
        one_step(warmup + t)
        if t + 1 == bounds[r]:
            marks[r].record(ext)   # (... and the last group ends it)
            r += 1
    grp.sync()
    if gather is not None:
        gather.finish()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(marks[rounds])
    if dist is not None:
        dist.barrier()
        tmax = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall = float(tmax.item())
    bits, first = grp.poll_errors()
    if bits:
        raise SystemExit(f"engine reported env error bits {bits} (first env {first})")
    per_round = []
    for i in range(rounds):
        n_i = bounds[i + 1] - bounds[i]
        if n_i > 0:
            per_round.append(E * A * n_i / (marks[i].elapsed_time(marks[i + 1]) * 1e-3))
    mean = float(np.mean(per_round))
    std = float(np.std(per_round))
    cv = std / mean if mean else 0.0

    # per-kernel durations for the roofline line: HIP events between the two kernels on the engine stream
    # (one group at a time, so that a duration is the kernel's own and not its share of an overlapped pair; a launch
    # covers E / G envs)
    grp.set_profiling(True)
    nprof = min(50, max(5, steps))
    seg = {}
    for t in range(nprof):
        for g, eng in enumerate(grp.engines):
            one_step(t, only=g)
            for k, v in eng.step_timing_segments_ms().items():   # waits for the step
                seg[k] = seg.get(k, 0.0) + v / (nprof * G)
    grp.set_profiling(False)
    grp.close()
    del grp, pre_a, pre_v
    torch.cuda.empty_cache()

    # Roofline of the dominant kernel (DESIGN.md "Roofline accounting").  Algorithmic bytes per agent-step: the
    # observation kernel owns obs out 3T + reward 4 + terminal 1 + truncation 1 + one read of the env state
    # S_env/A; the world-update kernel owns the two action streams (8) + one pass over S_env/A.
    names = {"obs": "mgx_obs_kernel", "actions": "mgx_act_kernel_x" if rung == 4 else "mgx_world_kernel_fast"}
    dom = "obs" if seg["obs"] >= seg["actions"] else "actions"
    bytes_per_agent_step = per["obs"] if dom == "obs" else per["world"]
    achieved = (E // G) * A * bytes_per_agent_step / (seg[dom] * 1e-3) / 1e9
    traffic, source, valu_frac = None, None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            pj = json.load(open(pmc))
            traffic = pj.get(f"rung{rung}", {}).get(names[dom] + "_bytes_per_launch")
            valu_frac = pj.get(f"rung{rung}", {}).get(names[dom] + "_valu_frac")
            source = f"profiles/pmc_traffic.json@{pj.get('build', '?')} (PMC FETCH_SIZE x2 + WRITE_SIZE of a profiled run, not this run)"
        except Exception:
            traffic = None
    return {
        "value": world * E * A * steps / wall, "ms_per_step": wall * 1e3 / steps, "device_ms_per_step": dev_ms / steps,
        "steps": steps, "warmup": warmup, "variants": variants,
        "rounds": {"n": len(per_round), "steps_per_round": steps // rounds, "mean": mean * world, "std": std * world, "cv": cv,
                   "unstable": bool(cv > 0.20), "per_gpu": world > 1},
        "kernels_ms": {"world_actions": seg["actions"], "aoe": seg["aoe"], "world_tail": seg["tail"],
                       "mgx_obs_kernel": seg["obs"], "rewards_ext": seg["rewards"]},
        "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                     "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": source,
                     "valu_frac": valu_frac,   # share of SIMD cycles issuing VALU instructions (same profiled run as `traffic`)
                     "bytes_per_agent_step": bytes_per_agent_step, "tick_bytes_per_agent_step": per["tick"],
                     "kernel_ms": seg[dom]},
        "config": {"workload": desc, "baseline_config": f"configs[{rung - 1}]", "envs_per_gpu": E, "agents_per_env": A,
                   "obs_tokens": T, "gather": gather_kind, "parallelism": f"env-shard x{world}",
                   "env_groups_per_gpu": G, "envs_per_launch": E // G},
    }


def env_wrapper_throughput(envs: int, steps: int, local_rank: int) -> dict:
    """Agent-steps/s a trainer sees through the PufferEnv-shaped wrapper (mettagrid_amd/envs.py): rung-3 rules, max_steps =
    1000, device map pool of 256 maps with on-device auto-reset and desynchronised first episodes, joint action ids decoded on
    the device — what MettaGridPufferEnv.step does around MettaGrid::step (mettagrid_puffer_env.py:296-408)."""
    import torch
    from mettagrid_amd import presets
    from mettagrid_amd.compiler import compile_spec
    from mettagrid_amd.envs import MettaGridBatchedEnv
    from mettagrid_amd.mapgen import random_class_maps
    spec = presets.rung3_spec()
    spec.max_steps = 1000
    prog = compile_spec(spec, 32, 32, max_objects=192)
    pool = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, range(256))
    out = {}
    for validate in (False, True):
        env = MettaGridBatchedEnv(prog, envs, map_pool=pool, pool_stride=7, desync=True, validate_actions=validate, seed=1,
                                  device=local_rank)
        env.reset()
        gen = torch.Generator(device="cuda").manual_seed(3)
        acts = torch.randint(0, int(env.transport_action_n), (8, env.num_agents), dtype=torch.int32, device="cuda", generator=gen)
        for t in range(20):
            env.step(acts[t % 8])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(steps):
            env.step(acts[t % 8])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ep, _ = env.engine.episodes()
        bits, _ = env.engine.poll_errors()
        out["validate_actions" if validate else "unchecked_actions"] = {
            "value": env.num_agents * steps / dt, "ms_per_step": dt * 1e3 / steps, "episodes_finished": int(ep.sum()),
            "env_error_bits": int(bits)}
        env.close()
    out["unit"] = "agent-steps/s"
    out["sample"] = (f"rung-3 rules through MettaGridBatchedEnv, {envs} envs x 16 agents, max_steps=1000, pool of 256 maps, "
                     f"desync, {steps} timed steps")
    return out


def launch_ranks(n: int) -> int:
    """One child process per GPU with the environment torch.distributed.run would give it (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT), same argv.  Children are fresh interpreters started BEFORE anything here touches the GPU.
    Returns the exit code: 0 only if every rank exited 0."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    code = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                rc = p.poll()
                if rc is None:
                    continue
                pending.remove(p)
                if rc != 0 and code == 0:   # one rank failed: the others would wait for it in the next barrier forever
                    code = rc if rc > 0 else 1
                    for q in pending:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return code


def stub_rank(args, rank: int, world: int) -> int:
    """--stub: the multi-rank plumbing without a GPU — rendezvous over gloo, the env shard of every rank, one all-reduce in
    place of the timing reduction; rank 0 prints the line."""
    import torch
    from mettagrid_amd.dist import env_shard
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
        if dist.get_world_size() != args.gpus:
            return 3
        n = torch.tensor([len(env_shard(rank, world, args.envs))], dtype=torch.int64)
        dist.all_reduce(n)
        dist.barrier()
        total, size = int(n.item()), dist.get_world_size()
        dist.destroy_process_group()
    else:
        total, size = len(env_shard(0, 1, args.envs)), 1
    if rank == 0:
        print(json.dumps({"metric": "agent-steps/sec (whole node) at 65 536 envs, 32x32x16-agent; HBM GB/s vs peak", "value": None,
                          "stub": True, "n_gpus": size, "steps": args.steps, "warmup": args.warmup, "envs_total": total,
                          "scaling": "weak"}))
    return 0


def main() -> None:
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        cpu_worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=10, help="the timed steps are reported as this many rounds (mean / std / CV)")
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--rung", type=int, default=3, help="3 = BASELINE.json configs[2] (the headline metric), 4 = configs[3]")
    ap.add_argument("--gather", choices=["none", "scalars", "obs", "obs-packed"], default="none",
                    help="optional per-step gather of rewards/terminals/truncations (+obs) to rank 0: grouped RCCL "
                         "send/recv over xGMI on a side stream, overlapped with the next step (mettagrid_amd/dist.py)")
    ap.add_argument("--groups", type=int, default=1,
                    help="env groups per GPU (mettagrid_amd/groups.py): 2 = the world update of one half of the envs runs "
                         "beside the observation kernel of the other half (measured slower on MI355X: DESIGN.md); "
                         "1 = one engine, kernels back to back")
    ap.add_argument("--cpu-steps", type=int, default=0, help="CPU baseline sample (default: about 10 s of one host core)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only: skip the rung-4 object and the env-wrapper throughput")
    ap.add_argument("--rung4-steps", type=int, default=100)
    ap.add_argument("--stub", action="store_true",
                    help="launcher self-test (tests/test_bench_launcher.py): the ranks rendezvous over gloo and do no GPU work; "
                         "rank 0 prints a line whose n_gpus is the communicator's size and whose value is null")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by hand: start the N ranks ourselves (what torch.distributed.run does for the driver).
        # This parent never touches the GPU; rank 0's line is the output.
        raise SystemExit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.stub:
        raise SystemExit(stub_rank(args, rank, world))
    extras = world == 1 and args.rung == 3 and not args.no_extras
    from mettagrid_amd.dist import env_shard

    # Everything that forks or spawns comes first: this process has not touched the GPU yet.
    workers = host_cores() if world == 1 else max(1, min(4, host_cores() // max(1, world)))
    cms = build_maps(args.rung, env_shard(rank, world, args.envs), workers)   # map seed = global env index
    cms4 = build_maps(4, env_shard(0, 1, args.envs), workers) if extras else None
    cpu = cpu4 = None
    if not args.no_cpu and world == 1:
        # about 10 s on one core, then every host core at once (one env per process, about 5 s of stepping each)
        cpu = cpu_leg(args.rung, args.cpu_steps or (300000 if args.rung == 3 else 25000))
        if extras:
            cpu4 = cpu_leg(4, 12000)

    jit_spec = jit_jobs = None
    jit_t0 = time.perf_counter()
    if extras:   # the run-time specialisation leg: start its compiles now (child processes), they run beside everything below
        from mettagrid_amd import jit, presets
        from mettagrid_amd.compiler import compile_spec
        jit_spec = presets.rung3_spec(obs_tokens=192, use_attack_mutation=False)   # not a build-time preset: other handlers, other shape
        if jit.enabled():
            jit_jobs = jit.start(compile_spec(jit_spec, 32, 32, max_objects=192), True)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the step engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the RCCL communicator has {dist.get_world_size()} ranks")

    common = dict(envs=args.envs, groups=args.groups, gather_kind=args.gather, rank=rank, local_rank=local_rank, world=world, dist=dist)
    head = measure(args.rung, cms, steps=args.steps, warmup=args.warmup, rounds=args.rounds, **common)
    r4 = wrap = gen = jit_leg = long_run = None
    if extras:
        # the headline workload over the reference harness' protocol length whatever --steps says
        # (python/src/mettagrid/perf/harness.py:100-209: >= 1 000 iterations, >= 10 rounds)
        lr = measure(3, cms, steps=max(1000, args.steps), warmup=max(20, args.warmup), rounds=10, **common)
        long_run = {k: lr[k] for k in ("value", "ms_per_step", "device_ms_per_step", "steps", "warmup", "rounds", "kernels_ms")}
        short = dict(steps=min(args.steps, 100), warmup=10, rounds=5)
        g3 = measure(3, cms, generic=True, **short, **common)
        gen = {k: g3[k] for k in ("value", "ms_per_step", "steps", "variants", "kernels_ms")}
        gen["note"] = "same workload on the kernels every program gets: shape read at run time, handler interpreter"
        # another program (flat damage instead of the Attack mutation, 192 tokens): generic kernels, then the code objects
        # compiled for it while this bench was running
        j0 = measure(3, cms, spec=jit_spec, generic=True, **short, **common)
        if jit_jobs is not None:
            for j in jit_jobs:
                j.wait(600)
            compile_s = time.perf_counter() - jit_t0
            j1 = measure(3, cms, spec=jit_spec, specialize="sync", **short, **common)
            jit_leg = {"workload": "rung-3 maps, flat damage instead of the Attack mutation, T = 192 (not a build-time preset)",
                       "generic": {k: j0[k] for k in ("value", "ms_per_step", "variants", "kernels_ms")},
                       "specialised": {k: j1[k] for k in ("value", "ms_per_step", "variants", "kernels_ms")},
                       "code_objects_ready_after_s": compile_s, "errors": [j.error for j in jit_jobs if j.error]}
    del cms
    if extras:
        r4 = measure(4, cms4, steps=args.rung4_steps, warmup=20, rounds=min(5, args.rung4_steps), **common)
        del cms4
        wrap = env_wrapper_throughput(args.envs, 100, local_rank)

    if rank == 0:
        out = {
            "metric": "agent-steps/sec (whole node) at 65 536 envs, 32x32x16-agent; HBM GB/s vs peak",
            "value": head["value"], "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": head["config"],
            "device_ms_per_step": head["device_ms_per_step"],
            "rounds": head["rounds"],
            "kernels_ms": head["kernels_ms"],
            "variants": head["variants"],
            "roofline": head["roofline"],
        }
        if long_run is not None:
            out["long_run"] = long_run
        if gen is not None:
            out["generic"] = gen
        if jit_leg is not None:
            out["run_time_specialisation"] = jit_leg
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if r4 is not None:
            r4 = dict(r4, metric="agent-steps/sec at 65 536 envs, 64x64x64-agent (BASELINE.json configs[3])", unit="agent-steps/s",
                      dtype="u8", data="synthetic")
            if cpu4 is not None:
                r4["cpu_baseline"] = cpu4
            out["rung4"] = r4
        if wrap is not None:
            out["env_wrapper"] = wrap
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
