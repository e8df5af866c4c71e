"""N>1 path on CPU: world_size-2 gloo, driving the code bench.py itself uses for N > 1 (mettagrid_amd/dist.py: env_shard,
shard_seeds, GatherToRoot).  Each rank steps the oracle on ITS env shard; what the root gathers — over several pipelined
submits — must equal a single-process run over all envs (shard mapping, seed assignment, rank-major row order)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

E_PER_RANK, STEPS = 3, 12


def _run_envs(env_ids):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as op
    from mettagrid_amd import presets
    from mettagrid_amd.compiler import compile_spec
    from mettagrid_amd.mapgen import random_class_maps

    spec = presets.rung3_spec()
    prog = compile_spec(spec, 32, 32, max_objects=192)
    cms = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, env_ids)
    sims = [op.OracleSim(prog, cms[i], int(e)) for i, e in enumerate(env_ids)]
    A, n = prog.num_agents, len(prog.action_names)
    per_step = []
    for t in range(STEPS):
        for i, e in enumerate(env_ids):
            rng = np.random.RandomState(1000 * int(e) + t)
            sims[i].step(rng.randint(0, n, A), rng.randint(0, n, A))
        snaps = [s.snapshot() for s in sims]
        per_step.append({k: np.concatenate([s[k] for s in snaps]) for k in ("rewards", "terminals", "truncations", "obs")})
    return per_step


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from mettagrid_amd.dist import GatherToRoot, env_shard, max_over_ranks, shard_seeds
    ids = list(env_shard(rank, world, E_PER_RANK))
    assert list(shard_seeds(rank, world, E_PER_RANK)) == ids
    steps = _run_envs(ids)
    gather = GatherToRoot(dist, root=0)
    packed = GatherToRoot(dist, root=0, packed=("observations",))   # the same rows as used prefixes (bench.py --gather obs-packed)
    got = []
    for out in steps:                      # one submit per step, like bench.py --gather obs
        tensors = {"rewards": torch.from_numpy(out["rewards"]), "terminals": torch.from_numpy(out["terminals"]),
                   "truncations": torch.from_numpy(out["truncations"]), "observations": torch.from_numpy(out["obs"])}
        gather.submit(tensors)
        res = gather.result()
        assert (res is None) == (rank != 0)
        packed.submit(tensors)
        pres = packed.result()
        if res is not None:
            got.append({k: v.numpy().copy() for k, v in res.items()})
            # packed mode: same scalars, and the expanded rows are the whole-row gather byte for byte — from a third of the bytes
            assert torch.equal(pres["rewards"], res["rewards"]) and torch.equal(pres["terminals"], res["terminals"])
            assert torch.equal(packed.expand("observations"), res["observations"])
            n_used = int((res["observations"][:, :, 0] != 0xFF).sum())
            assert pres["observations_packed"].shape == (n_used, 3) and int(pres["observations_counts"].sum()) == n_used
            assert pres["observations_packed"].numel() < 0.6 * res["observations"].numel()
    gather.finish()
    packed.finish()
    t = max_over_ranks(dist, 1.0 + rank)
    if rank == 0:
        q.put((got, t))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_shard_and_gather():
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered, tmax = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    single = _run_envs(list(range(world * E_PER_RANK)))
    assert len(gathered) == STEPS
    for t in range(STEPS):
        assert np.array_equal(gathered[t]["rewards"], single[t]["rewards"])
        assert np.array_equal(gathered[t]["terminals"], single[t]["terminals"])
        assert np.array_equal(gathered[t]["truncations"], single[t]["truncations"])
        assert np.array_equal(gathered[t]["observations"], single[t]["obs"])
    assert tmax == 2.0
