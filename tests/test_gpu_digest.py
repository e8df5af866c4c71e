"""GPU: batched state digests (mgx_state_digests) — one kernel and one copy give a 64-bit digest of every
env's signature state; it must equal the digest computed on the host from the oracle's dumps (and from the engine's own
per-env dumps), for the rule sets of every scenario and for envs on both sides of workgroup boundaries."""
import numpy as np
import pytest

import helpers as hp
import oracle_py as op
from mettagrid_amd import signature as sg
from mettagrid_amd.engine import BatchedMettaGrid

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["rung3", "torture", "rung4", "dynamic", "rung4_full", "crowd"])
def test_digest_equals_host_digest_of_oracle_state(name):
    spec_f, map_f, steps, invalid = hp.SCENARIOS[name]
    steps = min(steps, 40)
    spec = spec_f()
    E = 5
    maps = [map_f(s) for s in range(E)]
    prog = hp.compile_scenario(name, spec, *maps[0].shape)
    cms = np.stack([prog.class_map(m) for m in maps])
    seeds = np.arange(E, dtype=np.uint32) + 21
    eng = BatchedMettaGrid(prog, cms, seeds, buffers="host")
    oracles = [op.OracleSim(prog, cms[i], int(seeds[i])) for i in range(E)]
    for o in oracles:
        o.reinit_buffers()
    acts = [hp.make_actions(prog, i, steps, invalid) for i in range(E)]
    A = prog.num_agents
    for t in range(steps):
        eng.actions[:] = np.concatenate([acts[i][0][t] for i in range(E)])
        eng.vibe_actions[:] = np.concatenate([acts[i][1][t] for i in range(E)])
        eng.step()
        for i, o in enumerate(oracles):
            o.step(acts[i][0][t], acts[i][1][t])
        if t in (0, steps // 2, steps - 1):
            dig = eng.state_digests()
            for i, o in enumerate(oracles):
                s = o.snapshot()
                want = sg.state_digest(o.raw_objects(), o.raw_stats(), s["episode_rewards"], s["action_success"],
                                       o.current_stat_reward(), o.current_step)
                assert int(dig[i]) == want, f"{name} env {i} step {t + 1}"
    # the engine's own per-env getters describe the same state
    snap = eng.snapshot()
    dig = eng.state_digests()
    for i in range(E):
        mine = sg.state_digest(eng.raw_objects(i), eng.raw_stats(i), snap["episode_rewards"][i * A:(i + 1) * A],
                               snap["action_success"][i * A:(i + 1) * A], eng.current_stat_reward(i), int(eng.current_steps()[i]))
        assert int(dig[i]) == mine
    assert len(set(int(x) for x in dig)) == E   # different maps / seeds -> different states


def test_digest_covers_far_invalid_action_indices():
    """The (k, count) pairs behind "action.invalid_index.<k>" for indices far outside the action table are part of the state."""
    spec_f, map_f, _, _ = hp.SCENARIOS["rung1"]
    spec, cells = spec_f(), map_f(0)
    prog = hp.compile_scenario("rung1", spec, *cells.shape)
    cm = prog.class_map(cells)[None]
    eng = BatchedMettaGrid(prog, cm, [3], buffers="host")
    ora = op.OracleSim(prog, cm[0], 3)
    ora.reinit_buffers()
    A = prog.num_agents
    before = int(eng.state_digests()[0])
    for a, v in ((1000, 0), (-500, 0), (1000, 2 ** 31 - 1)):
        eng.actions[:] = a
        eng.vibe_actions[:] = v
        eng.step()
        ora.step(np.full(A, a, np.int32), np.full(A, v, np.int32))
    s = ora.snapshot()
    want = sg.state_digest(ora.raw_objects(), ora.raw_stats(), s["episode_rewards"], s["action_success"], ora.current_stat_reward(),
                           ora.current_step, invalid_extra=ora.invalid_index_extra())
    plain = sg.state_digest(ora.raw_objects(), ora.raw_stats(), s["episode_rewards"], s["action_success"], ora.current_stat_reward(),
                            ora.current_step)
    got = int(eng.state_digests()[0])
    assert got == want and got != plain and got != before


@pytest.mark.parametrize("name", ["rung3", "rung4", "dynamic"])
def test_batched_object_export_equals_per_env_path_and_oracle(name):
    """mgx_get_objects_batch (one kernel + one copy for a list of envs, any order, repeats allowed) == mgx_get_objects per
    env == the oracle's object dump, after objects have moved, traded, been removed and spawned."""
    spec_f, map_f, steps, invalid = hp.SCENARIOS[name]
    steps = min(steps, 40)
    E = 9
    maps = [map_f(s) for s in range(E)]
    prog = hp.compile_scenario(name, spec_f(), *maps[0].shape)
    cms = np.stack([prog.class_map(m) for m in maps])
    seeds = np.arange(E, dtype=np.uint32) + 5
    eng = BatchedMettaGrid(prog, cms, seeds, buffers="host")
    oracles = [op.OracleSim(prog, cms[i], int(seeds[i])) for i in range(E)]
    for o in oracles:
        o.reinit_buffers()
    acts = [hp.make_actions(prog, i, steps, invalid) for i in range(E)]
    for t in range(steps):
        eng.actions[:] = np.concatenate([acts[i][0][t] for i in range(E)])
        eng.vibe_actions[:] = np.concatenate([acts[i][1][t] for i in range(E)])
        eng.step()
        for i, o in enumerate(oracles):
            o.step(acts[i][0][t], acts[i][1][t])
    order = [8, 0, 3, 3, 7, 1]
    batch = eng.raw_objects_batch(order)
    assert len(batch) == len(order)
    for e, raw in zip(order, batch):
        assert np.array_equal(raw, eng.raw_objects(e)), f"{name} env {e}: batch != per-env export"
        assert np.array_equal(raw, oracles[e].raw_objects()), f"{name} env {e}: export != oracle"
    dicts = eng.grid_objects_batch([2, 4])
    assert dicts[0] == eng.grid_objects(2) and dicts[1] == eng.grid_objects(4)
    assert eng.raw_objects_batch([]) == []
    with pytest.raises(ValueError):
        eng.raw_objects_batch([E])
    eng.close()
