"""GPU: the action dispatch with one lane per AGENT (csrc/mgx_act.h) — agents of an env run side by side in rounds
ordered by their cell footprints — must give what the reference's one-after-another walk in shuffled order gives
(mettagrid_c.cpp:966-999).  Checked three ways: every scenario of the parity suite again with the lean games forced onto
that kernel; crowded arenas where most agents touch another agent every step (attack / loot / swap chains, the case a
swapped agent attacks from its new cell before a later agent's turn); and the two kernels against each other on thousands
of envs by state digest."""
import os

import numpy as np
import pytest

import helpers as hp
import oracle_py as op
from mettagrid_amd import presets
from mettagrid_amd.compiler import compile_spec
from mettagrid_amd.engine import BatchedMettaGrid
from mettagrid_amd.mapgen import random_class_maps, random_map

pytestmark = pytest.mark.gpu


def _against_oracle(prog, cms, seeds, steps, rng_seed, where, expect_variant):
    import torch
    E, A = len(cms), prog.num_agents
    eng = BatchedMettaGrid(prog, cms, seeds, buffers="device")
    assert eng.act_variant == expect_variant, (where, eng.act_variant)
    oracles = [op.OracleSim(prog, cms[i], int(seeds[i])) for i in range(E)]
    for o in oracles:
        o.reinit_buffers()
    rng = np.random.default_rng(rng_seed)
    n_act = len(prog.action_names)
    for t in range(steps):
        a = rng.integers(-1, n_act + 1, E * A).astype(np.int32)
        v = rng.integers(0, n_act, E * A).astype(np.int32)
        eng.actions.copy_(torch.from_numpy(a)); eng.vibe_actions.copy_(torch.from_numpy(v)); torch.cuda.synchronize()
        eng.step()
        snap = eng.snapshot()
        for i, o in enumerate(oracles):
            o.step(a[i * A:(i + 1) * A], v[i * A:(i + 1) * A])
            hp.compare_snapshots(o.snapshot(), {k: x[i * A:(i + 1) * A] for k, x in snap.items()}, f"{where} env {i} step {t + 1}")
    bits = 0
    for o in oracles:
        bits |= int(o.error)
    assert eng.poll_errors()[0] == bits     # (crowded arenas may overflow the token budget: both sides flag it)
    snap = eng.snapshot()
    for i, o in enumerate(oracles):
        mine = {k: x[i * A:(i + 1) * A] for k, x in snap.items()}
        pa = hp.payload_from_raw(prog, o.raw_objects(), o.current_stat_reward(), o.raw_stats(), o.snapshot(), steps, int(seeds[i]))
        pb = hp.payload_from_raw(prog, eng.raw_objects(i), eng.current_stat_reward(i), eng.raw_stats(i), mine, steps, int(seeds[i]))
        assert pa == pb, f"{where} env {i}: signature payload differs: {hp.diff_payload(pa, pb)}"
    eng.close()


@pytest.mark.parametrize("name", list(hp.SCENARIOS))
def test_every_scenario_with_lean_games_on_the_agent_kernel(name, monkeypatch):
    monkeypatch.setenv("MGX_ACT_LEAN", "1")
    from test_gpu_parity import test_step_parity
    test_step_parity(name, "device")


def test_crowded_lean_arena(monkeypatch):
    """16 agents of two teams on a 9x9 floor (rung-3 rules: attack + loot, same-team swap, extractors, chests): nearly every
    move meets another agent, chains of three and more are the rule."""
    monkeypatch.setenv("MGX_ACT_LEAN", "1")
    prog = compile_spec(presets.rung3_spec(), 11, 11, max_objects=192)
    E = 24
    maps = [random_map(11, 11, {"wall": 3, "extractor": 4, "chest": 2}, {"red": 8, "blue": 8}, 900 + s) for s in range(E)]
    cms = np.stack([prog.class_map(m) for m in maps])
    _against_oracle(prog, cms, np.arange(E, dtype=np.uint32) + 77, 60, 5, "crowded lean arena", 1)


def test_two_game_stat_sets_in_one_chain(monkeypatch):
    """on_use sets a game stat to a LARGE value, the actor's on_after_use sets it to a SMALL one: game_stats->set is
    last-write-wins, so the small value must survive on the lane-per-agent kernel too (its per-env cell orders the sets by
    (order position, set count), not by bit pattern)."""
    from mettagrid_amd import spec as S
    monkeypatch.setenv("MGX_ACT_LEAN", "1")
    spec = presets.rung3_spec()
    A_, T_ = S.ACTOR, S.TARGET
    spec.objects["chest"].on_use = S.Handler([], [S.SetStat("chest.mark", S.ConstValue(1000.0), scope="game", entity=T_)], "mark_big")
    for ag in spec.agents:
        ag.on_after_use = S.Handler([], [S.SetStat("chest.mark", S.SumValue([S.InventoryValue("laser")], [1.0]), scope="game", entity=A_)],
                                    "mark_small")
    prog = compile_spec(spec, 11, 11, max_objects=192)
    E = 12
    maps = [random_map(11, 11, {"wall": 3, "extractor": 2, "chest": 6}, {"red": 8, "blue": 8}, 40 + s) for s in range(E)]
    cms = np.stack([prog.class_map(m) for m in maps])
    _against_oracle(prog, cms, np.arange(E, dtype=np.uint32) + 9, 40, 8, "double game-stat set", 1)


def test_crowded_extended_arena():
    """BASELINE.json configs[3] rules (64 agents, four teams) squeezed into 20x20: AoE, territory and events run behind a
    dispatch in which a third of the agents conflict with somebody every step."""
    prog = compile_spec(presets.rung4_spec(obs_tokens=400), 20, 20, max_objects=presets.RUNG4_MAX_OBJECTS)
    objs = {"wall": 8, "extractor": 6, "chest": 4, "healer_red": 1, "healer_blue": 1, "healer_green": 1, "healer_yellow": 1,
            "flag_red": 1, "flag_blue": 1, "flag_green": 1, "flag_yellow": 1, "hub": 1, "wire": 3}
    E = 8
    maps = [random_map(20, 20, dict(objs), dict(presets.RUNG4_AGENTS), 300 + s) for s in range(E)]
    cms = np.stack([prog.class_map(m) for m in maps])
    _against_oracle(prog, cms, np.arange(E, dtype=np.uint32) + 5, 56, 6, "crowded extended arena", 1)


@pytest.mark.parametrize("rung", [3, 4])
def test_shuffle_replay_path(rung, monkeypatch):
    """The lanes of an env draw the shuffle's swaps in parallel; a Lemire rejection (about one draw in a million) makes the
    env replay them serially.  MGX_ACT_SHUFFLE_REPLAY sends every step down that path: same digests as the env kernel."""
    monkeypatch.setenv("MGX_ACT_SHUFFLE_REPLAY", "1")
    test_agent_kernel_equals_env_kernel_by_digest(rung, monkeypatch, E=2048, steps=30)


@pytest.mark.parametrize("rung", [3, 4])
def test_agent_kernel_equals_env_kernel_by_digest(rung, monkeypatch, E=8192, steps=None):
    """Same maps, seeds and action traces through both kernels: equal state digests after every tenth step, 8 192 envs."""
    import torch
    if rung == 3:
        prog = compile_spec(presets.rung3_spec(), 32, 32, max_objects=192)
        maps = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, range(min(E, 1024)))
        steps = steps or 120
    else:
        prog = compile_spec(presets.rung4_spec(), 64, 64, max_objects=presets.RUNG4_MAX_OBJECTS)
        maps = random_class_maps(prog, 64, 64, presets.RUNG4_OBJECTS, presets.RUNG4_AGENTS, range(min(E, 512)))
        steps = steps or 60
    cms = maps[np.arange(E) % len(maps)]
    seeds = np.arange(E, dtype=np.uint32)
    monkeypatch.setenv("MGX_ACT_LEAN", "1")
    par = BatchedMettaGrid(prog, cms, seeds, buffers="device")
    monkeypatch.setenv("MGX_ACT_SERIAL", "1")
    ser = BatchedMettaGrid(prog, cms, seeds, buffers="device")
    assert (par.act_variant, ser.act_variant) == (1, 0)
    A, n_act = prog.num_agents, len(prog.action_names)
    gen = torch.Generator(device="cuda").manual_seed(7 + rung)
    for t in range(steps):
        a = torch.randint(-1, n_act + 1, (E * A,), dtype=torch.int32, device="cuda", generator=gen)
        v = torch.randint(0, n_act, (E * A,), dtype=torch.int32, device="cuda", generator=gen)
        for eng in (par, ser):
            eng.actions.copy_(a); eng.vibe_actions.copy_(v)
            eng.wait_for_caller(); eng.step(); eng.caller_waits()
        if (t + 1) % 10 == 0:
            dp, ds = par.state_digests(), ser.state_digests()
            bad = np.nonzero(dp != ds)[0]
            assert len(bad) == 0, f"rung {rung} step {t + 1}: envs {bad[:8].tolist()} differ ({len(bad)})"
    assert par.poll_errors()[0] == 0 and ser.poll_errors()[0] == 0
    par.close(); ser.close()


def test_large_map_takes_the_all_pairs_conflict_walk(monkeypatch):
    """Maps above 64 x 64 cells have no cell map in LDS: conflicts are found by walking the pending lanes (ballot + readlane).
    Same rules, 16 agents crowded into one corner region of an 80 x 96 map would rarely meet — so the arena is forced onto
    that path with MGX_ACT_NO_MAP as well."""
    monkeypatch.setenv("MGX_ACT_LEAN", "1")
    prog = compile_spec(presets.rung3_spec(), 80, 96, max_objects=600)
    E = 4
    cms = random_class_maps(prog, 80, 96, {"wall": 150, "extractor": 30, "chest": 10}, {"red": 8, "blue": 8}, range(E))
    _against_oracle(prog, cms, np.arange(E, dtype=np.uint32) + 3, 12, 8, "80x96 map", 1)
    monkeypatch.setenv("MGX_ACT_NO_MAP", "1")
    prog = compile_spec(presets.rung3_spec(), 11, 11, max_objects=192)
    E = 12
    maps = [random_map(11, 11, {"wall": 3, "extractor": 4, "chest": 2}, {"red": 8, "blue": 8}, 400 + s) for s in range(E)]
    cms = np.stack([prog.class_map(m) for m in maps])
    _against_oracle(prog, cms, np.arange(E, dtype=np.uint32) + 9, 50, 9, "crowded arena, all-pairs walk", 1)


def test_auto_reset_behind_the_agent_kernel():
    """Episodes that end and restart on the device (map pool, max_steps) with the dispatch on the agent kernel: the state
    digests of both kernels stay equal across restarts."""
    import torch
    spec = presets.rung4_spec(max_steps=12)
    prog = compile_spec(spec, 64, 64, max_objects=presets.RUNG4_MAX_OBJECTS)
    E = 96
    maps = random_class_maps(prog, 64, 64, presets.RUNG4_OBJECTS, presets.RUNG4_AGENTS, range(8))
    cms = maps[np.arange(E) % len(maps)]
    seeds = np.arange(E, dtype=np.uint32) + 40
    engines = []
    for serial in (False, True):
        if serial:
            os.environ["MGX_ACT_SERIAL"] = "1"
        try:
            eng = BatchedMettaGrid(prog, cms, seeds, buffers="device")
        finally:
            os.environ.pop("MGX_ACT_SERIAL", None)
        eng.set_map_pool(maps)
        eng.set_auto_reset(True, pool_stride=3)
        engines.append(eng)
    par, ser = engines
    assert (par.act_variant, ser.act_variant) == (1, 0)
    A, n_act = prog.num_agents, len(prog.action_names)
    g = torch.Generator(device="cuda").manual_seed(5)
    for t in range(40):
        a = torch.randint(0, n_act, (E * A,), dtype=torch.int32, device="cuda", generator=g)
        v = torch.randint(0, n_act, (E * A,), dtype=torch.int32, device="cuda", generator=g)
        for eng in engines:
            eng.actions.copy_(a); eng.vibe_actions.copy_(v)
            eng.wait_for_caller(); eng.step(); eng.caller_waits()
        assert np.array_equal(par.state_digests(), ser.state_digests()), f"step {t + 1}"
        assert torch.equal(par.obs, ser.obs) and torch.equal(par.rewards, ser.rewards) and torch.equal(par.terminals, ser.terminals)
    par.close(); ser.close()


def test_variant_choice(monkeypatch):
    """Extended games whose action handlers stay with actor and target take the agent kernel; a handler that mutates tags
    (rung-4 `dynamic` scenario), more than 64 agents, or MGX_ACT_SERIAL keep the env kernel."""
    def variant(name):
        spec_f, map_f, _, _ = hp.SCENARIOS[name]
        m = map_f(0)
        prog = hp.compile_scenario(name, spec_f(), *m.shape)
        eng = BatchedMettaGrid(prog, prog.class_map(m)[None], [1], buffers="host")
        v = eng.act_variant
        eng.close()
        return v
    assert variant("rung4_full") == 1
    assert variant("rung3") == 0            # lean games: lane per env unless MGX_ACT_LEAN is set (measured slower at 16 agents)
    assert variant("crowd") == 0            # 70 agents
    monkeypatch.setenv("MGX_ACT_LEAN", "1")
    assert variant("rung3") == 1
    assert variant("crowd") == 0
    monkeypatch.setenv("MGX_ACT_SERIAL", "1")
    assert variant("rung4_full") == 0


def test_paired_dispatch_equals_one_agent_at_a_time(monkeypatch):
    """The lean kernel's paired dispatch (MgxDev::duo: the env's own lane and its helper lane run two agents of the order side
    by side when their footprints are disjoint) against the same kernel with one agent at a time (MGX_NO_DUO): equal state
    digests over 8 192 envs, and against the oracle on crowded 11x11 arenas where a third of the pairs conflict."""
    import torch
    prog = compile_spec(presets.rung3_spec(), 32, 32, max_objects=192)
    E, steps = 8192, 120
    maps = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, range(1024))
    cms = maps[np.arange(E) % len(maps)]
    seeds = np.arange(E, dtype=np.uint32)
    duo = BatchedMettaGrid(prog, cms, seeds, buffers="device", specialize=False)
    monkeypatch.setenv("MGX_NO_DUO", "1")
    one = BatchedMettaGrid(prog, cms, seeds, buffers="device", specialize=False)
    monkeypatch.delenv("MGX_NO_DUO")
    A, n_act = prog.num_agents, len(prog.action_names)
    gen = torch.Generator(device="cuda").manual_seed(9)
    for t in range(steps):
        a = torch.randint(-1, n_act + 1, (E * A,), dtype=torch.int32, device="cuda", generator=gen)
        v = torch.randint(0, n_act, (E * A,), dtype=torch.int32, device="cuda", generator=gen)
        for g in (duo, one):   # (the engines run on their own streams: order them behind the action writes and back)
            g.actions.copy_(a); g.vibe_actions.copy_(v)
            g.wait_for_caller(); g.step(); g.caller_waits()
        if t % 10 == 9:
            assert np.array_equal(duo.state_digests(), one.state_digests()), t
    torch.cuda.synchronize()
    assert (duo.L.mgx_obs_variant(duo.h), one.L.mgx_obs_variant(one.h)) == (3, 3)
    assert torch.equal(duo.obs, one.obs) and torch.equal(duo.rewards, one.rewards)
    duo.close(); one.close()
    # crowded: 16 agents on a 9x9 floor — most pairs share a cell
    prog = compile_spec(presets.rung3_spec(), 11, 11, max_objects=192)
    E = 24
    maps = [random_map(11, 11, {"wall": 3, "extractor": 4, "chest": 2}, {"red": 8, "blue": 8}, 400 + s) for s in range(E)]
    cms = np.stack([prog.class_map(m) for m in maps])
    _against_oracle(prog, cms, np.arange(E, dtype=np.uint32) + 31, 60, 7, "crowded arena, paired dispatch", 0)
