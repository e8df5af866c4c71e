"""The oracle restatement against the REAL reference engine (oracle/_ref/mettagrid_c*.so, built from
/root/reference/cpp by `make -C oracle ref`).  Skipped where that build is absent."""
import numpy as np
import pytest

import helpers as hp
import oracle_py as op
import ref_driver as rd
from mettagrid_amd import signature as sg
from mettagrid_amd.compiler import compile_spec

pytestmark = pytest.mark.skipif(not rd.ref_available(), reason="oracle/_ref not built (no /root/reference here)")


@pytest.mark.parametrize("name", list(hp.SCENARIOS))
@pytest.mark.parametrize("seed", [5, 6])
def test_oracle_equals_reference(name, seed):
    spec_f, map_f, steps, invalid = hp.SCENARIOS[name]
    spec, cells = spec_f(), map_f(seed)
    prog = hp.compile_scenario(name, spec, *cells.shape)
    ref = rd.RefSim(spec, cells, seed, prog)
    ora = op.OracleSim(prog, prog.class_map(cells), seed)
    acts, vibes = hp.make_actions(prog, seed, steps, invalid)
    hp.compare_snapshots(ref.snapshot(), ora.snapshot(), f"{name} step 0")
    for t in range(steps):
        ref.step(acts[t], vibes[t])
        ora.step(acts[t], vibes[t])
        hp.compare_snapshots(ref.snapshot(), ora.snapshot(), f"{name} seed {seed} step {t + 1}")
    c = ref.c
    pa = sg.payload(c.grid_objects(), c.get_episode_stats(), c.action_success(), c.get_episode_rewards(), c.current_step, seed)
    pb = hp.payload_from_raw(prog, ora.raw_objects(), ora.current_stat_reward(), ora.raw_stats(), ora.snapshot(),
                             ora.current_step, seed)
    assert pa == pb, hp.diff_payload(pa, pb)


def test_restated_rng_equals_libstdcxx():
    L = op.lib()
    for seed in (0, 1, 42, 123456789):
        for n in (1, 2, 3, 4, 5, 16, 24, 63, 64, 255):
            assert L.mgxo_selftest_shuffle(seed, n, 300) == 0


def test_reference_inventory_token_order_follows_emulated_unordered_map():
    """Observation token order of inventory items is libstdc++ unordered_map order; the spec's dict
    order of initial inventories must survive compile -> oracle exactly as it does config -> reference."""
    from mettagrid_amd import spec as S
    for initial in ({"a": 1, "d": 2, "b": 3}, {"d": 5, "c": 1, "a": 9, "b": 2}, {"b": 1}, {"c": 3, "a": 0, "d": 4}):
        spec = S.GameSpec(resource_names=["a", "b", "c", "d"],
                          agents=[S.AgentSpec(inventory=S.Inventory(initial=dict(initial)))],
                          objects={"wall": S.ObjectSpec("wall", kind="wall"),
                                   "box": S.ObjectSpec("box", inventory=S.Inventory(initial=dict(initial)))},
                          obs=S.ObsSpec(5, 5, 60))
        cells = np.array([["wall"] * 5, ["wall", "agent.agent", "box", "empty", "wall"], ["wall"] * 5])
        prog = compile_spec(spec, *cells.shape)
        ref = rd.RefSim(spec, cells, 1, prog)
        ora = op.OracleSim(prog, prog.class_map(cells), 1)
        assert np.array_equal(ref.snapshot()["obs"], ora.snapshot()["obs"]), initial


def test_far_invalid_action_indices_match_reference_stat_keys():
    """The reference keys "action.invalid_index.<k>" by the raw index (mettagrid_c.cpp:916-918); the oracle's (k, count)
    pairs must produce the same stats dict."""
    spec_f, map_f, _, _ = hp.SCENARIOS["rung1"]
    spec, cells = spec_f(), map_f(0)
    prog = hp.compile_scenario("rung1", spec, *cells.shape)
    ref = rd.RefSim(spec, cells, 3, prog)
    ora = op.OracleSim(prog, prog.class_map(cells), 3)
    n, A = len(prog.action_names), prog.num_agents
    for a, v in ((1000, 0), (-500, 0), (1000, 2 ** 31 - 1), (n + 40, -5)):
        acts, vibes = np.full(A, a, np.int32), np.full(A, v, np.int32)
        ref.step(acts, vibes)
        ora.step(acts, vibes)
    assert ora.error == 0
    mine = sg.stats_dicts(prog, *ora.raw_stats(), extra=ora.invalid_index_extra())
    theirs = ref.c.get_episode_stats()
    for i in range(A):
        want = {k: v for k, v in theirs["agent"][i].items() if k.startswith("action.invalid_index")}
        got = {k: v for k, v in mine["agent"][i].items() if k.startswith("action.invalid_index")}
        assert got == want
