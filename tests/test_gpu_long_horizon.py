"""GPU: long-horizon parity at BASELINE.json's full batch (the determinism bar of the reference's
tests/simulator/test_deterministic_signature.py:14-17, at 65 536 envs).

The per-step full-size tests stop after three steps; by step 200 agents have met, fought, traded and (rung 4) crossed
territories, and the timestep events at 50 / 75 / 100 have fired.  Here the whole batch is stepped that far with random
actions drawn on the device (invalid ids on both sides of the action table included) and no host round trip except the
sampled rows; then ``mgx_state_digests`` — one kernel over all envs — must equal the digest of ≥ 256 sampled envs' oracle
state (objects, every stat value and key, episode rewards, action success, reward state), replayed on the host cores
from the same action traces, and a handful of envs is compared buffer by buffer after every step.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import helpers as hp
import oracle_py as op
from mettagrid_amd import presets
from mettagrid_amd import signature as sg
from mettagrid_amd.compiler import compile_spec
from mettagrid_amd.engine import BatchedMettaGrid
from mettagrid_amd.mapgen import random_class_maps

pytestmark = pytest.mark.gpu

E = 65536


def _workload(rung: int):
    """Program, class maps and seeds of the bench workload.  Map construction is one numpy generator per map on the host
    (a minute for 65 536 rung-4 maps), so the batch plays ``distinct`` maps tiled over the envs; every env still has its
    own engine seed and its own action trace."""
    if rung == 3:
        prog = compile_spec(presets.rung3_spec(), 32, 32, max_objects=192)
        distinct = 8192
        maps = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, range(distinct))
    else:
        prog = compile_spec(presets.rung4_spec(), 64, 64, max_objects=presets.RUNG4_MAX_OBJECTS)
        distinct = 2048
        maps = random_class_maps(prog, 64, 64, presets.RUNG4_OBJECTS, presets.RUNG4_AGENTS, range(distinct))
    cms = maps[np.arange(E) % distinct]
    return prog, cms, np.arange(E, dtype=np.uint32)


def _long_run(rung: int, steps: int, n_digest: int, n_traced: int) -> None:
    import torch
    prog, cms, seeds = _workload(rung)
    A, n_act = prog.num_agents, len(prog.action_names)
    eng = BatchedMettaGrid(prog, cms, seeds, buffers="device")
    rng = np.random.default_rng(20260 + rung)
    sample = np.unique(np.concatenate([[0, 31, 32, 63, 64, E - 65, E - 1], rng.choice(E, n_digest, replace=False)]))
    traced = sample[:: max(1, len(sample) // n_traced)][:n_traced]          # envs compared buffer by buffer
    s_dev = torch.as_tensor(sample, device="cuda")
    t_dev = torch.as_tensor(traced, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(99 + rung)
    acts = np.empty((steps, len(sample), A), np.int32)
    vibes = np.empty((steps, len(sample), A), np.int32)
    trace = []
    for t in range(steps):
        a = torch.randint(-1, n_act + 1, (E * A,), dtype=torch.int32, device="cuda", generator=gen)   # invalid ids on both sides
        v = torch.randint(0, n_act, (E * A,), dtype=torch.int32, device="cuda", generator=gen)
        eng.actions.copy_(a)
        eng.vibe_actions.copy_(v)
        eng.wait_for_caller()
        eng.step()
        eng.caller_waits()
        acts[t] = a.view(E, A)[s_dev].cpu().numpy()
        vibes[t] = v.view(E, A)[s_dev].cpu().numpy()
        trace.append(dict(obs=eng.obs.view(E, A, -1, 3)[t_dev].cpu().numpy(), rewards=eng.rewards.view(E, A)[t_dev].cpu().numpy(),
                          terminals=eng.terminals.view(E, A)[t_dev].cpu().numpy().astype(bool),
                          truncations=eng.truncations.view(E, A)[t_dev].cpu().numpy().astype(bool)))
    eng.sync()
    bits, first = eng.poll_errors()
    assert bits == 0, (bits, first)
    digests = eng.state_digests()
    assert (eng.current_steps() == steps).all()
    traced_pos = {int(e): k for k, e in enumerate(traced)}

    def replay(j: int):
        e = int(sample[j])
        o = op.OracleSim(prog, cms[e], int(seeds[e]))
        o.reinit_buffers()
        k = traced_pos.get(e)
        for t in range(steps):
            o.step(acts[t, j], vibes[t, j])        # (the C call releases the GIL: the replays run on all host cores)
            if k is not None:
                s = o.snapshot()
                for name in ("obs", "rewards", "terminals", "truncations"):
                    assert np.array_equal(s[name], trace[t][name][k]), f"rung {rung} env {e} step {t + 1}: '{name}' differs"
        s = o.snapshot()
        assert o.error == 0
        return (o.raw_objects(), o.raw_stats(), s["episode_rewards"], s["action_success"], o.current_stat_reward(), o.current_step)

    with ThreadPoolExecutor(max_workers=16) as pool:
        dumps = list(pool.map(replay, range(len(sample))))
    want = sg.state_digest_many(dumps)
    got = digests[sample]
    bad = np.nonzero(want != got)[0]
    assert len(bad) == 0, f"rung {rung}: state digest differs after {steps} steps for envs {sample[bad][:8].tolist()} ({len(bad)} of {len(sample)})"
    assert len(np.unique(digests)) == E            # every env went its own way
    eng.close()


@pytest.mark.timeout(600)
def test_rung3_full_batch_200_steps():
    _long_run(3, steps=200, n_digest=320, n_traced=6)


@pytest.mark.timeout(800)
def test_rung4_full_batch_110_steps_past_all_events():
    _long_run(4, steps=110, n_digest=256, n_traced=4)


@pytest.mark.parametrize("name", ["torture", "dynamic", "rung4", "thirteen", "minmax", "crowd", "lit"])
def test_scenario_soak(name):
    """scripts/soak.py inside the suite: seeds and action traces the goldens do not hold, 32 envs per scenario, every
    caller-visible buffer after every step and the whole signature payload at the end."""
    spec_f, map_f, steps, invalid = hp.SCENARIOS[name]
    steps = min(2 * steps, 120)
    Es = 32
    maps = [map_f(1000 + s) for s in range(Es)]
    prog = hp.compile_scenario(name, spec_f(), *maps[0].shape)
    cms = np.stack([prog.class_map(m) for m in maps])
    seeds = np.arange(Es, dtype=np.uint32) + 7000
    eng = BatchedMettaGrid(prog, cms, seeds, buffers="host")
    oracles = [op.OracleSim(prog, cms[i], int(seeds[i])) for i in range(Es)]
    for o in oracles:
        o.reinit_buffers()
    acts = [hp.make_actions(prog, 500 + i, steps, invalid) for i in range(Es)]
    A = prog.num_agents
    for t in range(steps):
        eng.actions[:] = np.concatenate([acts[i][0][t] for i in range(Es)])
        eng.vibe_actions[:] = np.concatenate([acts[i][1][t] for i in range(Es)])
        eng.step()
        snap = eng.snapshot()
        for i, o in enumerate(oracles):
            o.step(acts[i][0][t], acts[i][1][t])
            hp.compare_snapshots(o.snapshot(), {k: v[i * A:(i + 1) * A] for k, v in snap.items()}, f"{name} env {i} step {t + 1}")
    both_overflow = bool(eng.poll_errors()[0] & 1)
    for i, o in enumerate(oracles):
        if o.error & 1 and both_overflow:
            continue   # token budget exceeded on both sides: the reference raises there (mettagrid_c.cpp:364-375)
        pa = hp.payload_from_raw(prog, eng.raw_objects(i), eng.current_stat_reward(i), eng.raw_stats(i),
                                 {k: v[i * A:(i + 1) * A] for k, v in snap.items()}, steps, int(seeds[i]))
        pb = hp.payload_from_raw(prog, o.raw_objects(), o.current_stat_reward(), o.raw_stats(), o.snapshot(), steps, int(seeds[i]))
        assert pa == pb, f"{name} env {i}: {hp.diff_payload(pa, pb)[:400]}"
    eng.close()
