"""`infos` against the REFERENCE's own StatsTracker.  tests/golden/infos_<scenario>.json holds what
StatsTracker.on_episode_end (python/src/mettagrid/envs/stats_tracker.py:26-76) put into infos when the reference — its
Python Simulation on its C++ engine, buffers bound as MettaGridPufferEnv binds them — played the committed action traces
of ref_navigation / ref_chains to the end of the episode (tests/golden/make_infos_fixture.py).  Here: the oracle replays
the trace and the test suite's restatement of on_episode_end (helpers.reference_infos) must give those dicts float for
float — which pins what tests/test_gpu_episode_stats.py compares the device reduction with; on the GPU the engine's own
episode record must give them too."""
import json
import os

import numpy as np
import pytest

import helpers as hp
import oracle_py as op
import ref_tree
from mettagrid_amd import from_reference

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ["navigation", "chains"]


def _load(name):
    doc = json.load(open(os.path.join(HERE, "golden", f"ref_{name}.json")))
    want = json.load(open(os.path.join(HERE, "golden", f"infos_{name}.json")))
    z = np.load(os.path.join(HERE, "golden", f"ref_{name}.npz"))
    cfg = ref_tree.load(doc["config"])
    prog = from_reference.compile_reference_config(cfg, len(doc["map"]), len(doc["map"][0]))
    return doc, want, z, prog


def _same_dict(got: dict, want: dict, where: str) -> None:
    assert sorted(got) == sorted(want), (where, sorted(set(got) ^ set(want)))
    for k, v in want.items():
        assert np.float64(got[k]).tobytes() == np.float64(v).tobytes(), (where, k, got[k], v)


def _check(got: dict, want: dict, name: str) -> None:
    _same_dict(got["game"], want["game"], f"{name} game")
    _same_dict(got["agent"], want["agent"], f"{name} agent")
    assert sorted(got["per_agent"]) == sorted(want["per_agent"])
    for i, d in want["per_agent"].items():
        _same_dict(got["per_agent"][i], d, f"{name} per_agent[{i}]")
    assert [float(x) for x in got["episode_rewards"]] == want["episode_rewards"]
    assert got["steps"] == want["attributes"]["steps"] == want["steps_played"]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_infos_equal_the_reference_stats_tracker(name):
    doc, want, z, prog = _load(name)
    o = op.OracleSim(prog, prog.class_map(doc["map"]), doc["seed"])
    o.reinit_buffers()    # the wrapper's set_buffers: a second initial-observation pass (mettagrid_c.cpp:1165-1184)
    for t in range(want["steps_played"]):
        for when, agent_id, inv in doc["set_inventory"]:
            if when == t:
                o.set_inventory(agent_id, dict(map(tuple, inv)))
        o.step(z["actions"][t], z["vibe_actions"][t])
    s = o.snapshot()
    assert s["truncations"].all() or s["terminals"].all()
    _check(hp.reference_infos(prog, o), want, name)
    assert want["attributes"]["max_steps"] == int(prog.words[11]) and want["attributes"]["seed"] == doc["seed"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_engine_episode_record_equals_the_reference_stats_tracker(name):
    from mettagrid_amd.engine import BatchedMettaGrid
    doc, want, z, prog = _load(name)
    eng = BatchedMettaGrid(prog, prog.class_map(doc["map"])[None], [doc["seed"]], buffers="host", specialize=False)
    eng.set_episode_stats(True, log_capacity=2, log_per_agent=True)
    for t in range(want["steps_played"]):
        for when, agent_id, inv in doc["set_inventory"]:
            if when == t:
                eng.set_inventory(0, agent_id, dict(map(tuple, inv)))
        eng.actions[:] = z["actions"][t]
        eng.vibe_actions[:] = z["vibe_actions"][t]
        eng.step()
    assert eng.truncations.all() or eng.terminals.all()
    eng.record_episodes([1])               # host-driven path: record the finished env before a restart would wipe it
    (rec,), dropped = eng.drain_episode_log()
    assert dropped == 0 and rec["seed"] == doc["seed"]
    got = dict(rec, per_agent={str(i): d for i, d in enumerate(rec["per_agent"])})
    _check(got, want, name)
    tot = eng.drain_episode_stats()
    assert tot["episodes"] == 1 and tot["length_sum"] == float(want["steps_played"])
    _same_dict(tot["game_sum"], want["game"], f"{name} totals game")      # one episode: the sums are its values
    _same_dict(tot["agent_sum"], want["agent"], f"{name} totals agent")
    eng.close()
