"""GPU: the per-action bookkeeping counters kept as integers beside the stat rows (MgxDev::shadow, csrc/mgx_world.h tail_shadow,
csrc/mgx_episode.h mgx_shadow_flush_kernel) against the reference's arithmetic (actions/action_handler.hpp:78-105: +1.f on
action.<kind>.success / .failed and action.failed, status.max_steps_without_motion as a running maximum; objects/agent.cpp:49-57
for the two coverage stats).  Three checks: the integer form against the stat-row form (MGX_NO_SHADOW=1) by state digest and
stat for stat, the integer form against the oracle mid-episode (every mgx_get_stats goes through the flush), and a program whose
reward READS one of the counters — the engine must then keep the stat rows (nothing on the device may see a stale cell)."""
import dataclasses

import numpy as np
import pytest

import helpers as hp
import oracle_py as op
from mettagrid_amd import presets
from mettagrid_amd import spec as S
from mettagrid_amd.compiler import compile_spec
from mettagrid_amd.engine import BatchedMettaGrid
from mettagrid_amd.mapgen import random_class_maps

pytestmark = pytest.mark.gpu


def _rung3(E):
    prog = compile_spec(presets.rung3_spec(), 32, 32, max_objects=192)
    cms = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, range(E))
    return prog, cms


def _run(prog, cms, seeds, steps, rng_seed, probe=()):
    """Steps an engine with seeded random actions; returns (digests, {env: raw_stats}, integer_bookkeeping)."""
    import torch
    E, A = len(cms), prog.num_agents
    eng = BatchedMettaGrid(prog, cms, seeds, buffers="device")
    rng = np.random.default_rng(rng_seed)
    n_act = len(prog.action_names)
    for t in range(steps):
        a = rng.integers(-1, n_act + 1, E * A).astype(np.int32)
        v = rng.integers(0, n_act, E * A).astype(np.int32)
        eng.actions.copy_(torch.from_numpy(a)); eng.vibe_actions.copy_(torch.from_numpy(v)); torch.cuda.synchronize()
        eng.step()
        eng.sync()   # (the next copy into the action buffers must not overtake this step's kernels)
    out = (eng.state_digests(), {i: eng.raw_stats(i) for i in probe}, eng.integer_bookkeeping)
    assert eng.poll_errors()[0] == 0
    eng.close()
    return out


def test_integer_counters_equal_the_stat_rows(monkeypatch, E=2048, steps=70):
    prog, cms = _rung3(E)
    seeds = np.arange(E, dtype=np.uint32) * 3 + 1
    probe = (0, 1, E // 2, E - 1)
    dig_i, stats_i, mode_i = _run(prog, cms, seeds, steps, 11, probe)
    assert mode_i == 3, mode_i   # counters + coverage stats as integers in the lean lane-per-env kernel
    monkeypatch.setenv("MGX_NO_SHADOW", "1")
    dig_r, stats_r, mode_r = _run(prog, cms, seeds, steps, 11, probe)
    assert mode_r == 0
    assert np.array_equal(dig_i, dig_r), f"{int((dig_i != dig_r).sum())} of {E} envs differ"
    for i in probe:
        for x, y in zip(stats_i[i], stats_r[i]):
            assert np.array_equal(np.asarray(x), np.asarray(y)), f"env {i}"


def test_lane_per_agent_dispatch_keeps_the_counters_as_integers(monkeypatch, E=512, steps=50):
    prog, cms = _rung3(E)
    seeds = np.arange(E, dtype=np.uint32) + 9
    monkeypatch.setenv("MGX_ACT_LEAN", "1")
    dig_i, stats_i, mode_i = _run(prog, cms, seeds, steps, 5, (0, E - 1))
    assert mode_i == 1, mode_i   # counters only (the coverage pass of that kernel writes the stat rows)
    monkeypatch.setenv("MGX_NO_SHADOW", "1")
    dig_r, stats_r, mode_r = _run(prog, cms, seeds, steps, 5, (0, E - 1))
    assert mode_r == 0
    assert np.array_equal(dig_i, dig_r)
    for i in (0, E - 1):
        for x, y in zip(stats_i[i], stats_r[i]):
            assert np.array_equal(np.asarray(x), np.asarray(y)), f"env {i}"


def test_stats_read_in_the_middle_of_an_episode_match_the_oracle(E=6, steps=40):
    import torch
    prog, cms = _rung3(E)
    seeds = np.arange(E, dtype=np.uint32) + 21
    A = prog.num_agents
    eng = BatchedMettaGrid(prog, cms, seeds, buffers="device")
    assert eng.integer_bookkeeping == 3
    oracles = [op.OracleSim(prog, cms[i], int(seeds[i])) for i in range(E)]
    for o in oracles:
        o.reinit_buffers()
    rng = np.random.default_rng(2)
    n_act = len(prog.action_names)
    for t in range(steps):
        a = rng.integers(-1, n_act + 1, E * A).astype(np.int32)
        v = rng.integers(0, n_act, E * A).astype(np.int32)
        eng.actions.copy_(torch.from_numpy(a)); eng.vibe_actions.copy_(torch.from_numpy(v)); torch.cuda.synchronize()
        eng.step()
        eng.sync()
        for i, o in enumerate(oracles):
            o.step(a[i * A:(i + 1) * A], v[i * A:(i + 1) * A])
        if t % 7 == 3 or t == steps - 1:   # a read flushes; the steps after it keep counting from the integers
            for i, o in enumerate(oracles):
                for x, y in zip(o.raw_stats(), eng.raw_stats(i)):
                    assert np.array_equal(np.asarray(x), np.asarray(y)), f"env {i} step {t + 1}"
    eng.close()


def test_a_reward_that_reads_a_counter_keeps_the_stat_rows(E=4, steps=30):
    base = presets.rung3_spec()
    agents = [dataclasses.replace(a, rewards=list(a.rewards) + [S.RewardSpec(S.StatValue("action.move.success", "agent"))]) for a in base.agents]
    spec = dataclasses.replace(base, agents=agents)
    prog = compile_spec(spec, 32, 32, max_objects=192)
    cms = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, range(E))
    seeds = np.arange(E, dtype=np.uint32) + 2
    import torch
    A = prog.num_agents
    eng = BatchedMettaGrid(prog, cms, seeds, buffers="device")
    assert eng.integer_bookkeeping == 0   # the observation kernel evaluates that reward from the stat row every step
    oracles = [op.OracleSim(prog, cms[i], int(seeds[i])) for i in range(E)]
    for o in oracles:
        o.reinit_buffers()
    rng = np.random.default_rng(4)
    n_act = len(prog.action_names)
    for t in range(steps):
        a = rng.integers(0, n_act, E * A).astype(np.int32)
        v = rng.integers(0, n_act, E * A).astype(np.int32)
        eng.actions.copy_(torch.from_numpy(a)); eng.vibe_actions.copy_(torch.from_numpy(v)); torch.cuda.synchronize()
        eng.step()
        snap = eng.snapshot()
        for i, o in enumerate(oracles):
            o.step(a[i * A:(i + 1) * A], v[i * A:(i + 1) * A])
            hp.compare_snapshots(o.snapshot(), {k: x[i * A:(i + 1) * A] for k, x in snap.items()}, f"env {i} step {t + 1}")
    eng.close()
