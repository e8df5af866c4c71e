"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle on identical seeded inputs.

Bit-exact on every caller-visible buffer after every step (observations u8, rewards f32, terminals, truncations,
action_success, episode rewards) and on the generalised episode signature (objects + every stat key/value).
"""
import numpy as np
import pytest

import helpers as hp
import oracle_py as op
from mettagrid_amd.compiler import compile_spec
from mettagrid_amd.engine import BatchedMettaGrid

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(hp.SCENARIOS))
@pytest.mark.parametrize("buffers", ["device", "host"])
def test_step_parity(name, buffers):
    spec_f, map_f, steps, invalid = hp.SCENARIOS[name]
    if buffers == "host" and name not in ("rung2", "torture"):
        pytest.skip("host-buffer path covered on two scenarios")
    spec = spec_f()
    E = 6
    maps = [map_f(s) for s in range(E)]
    prog = hp.compile_scenario(name, spec, *maps[0].shape)
    cms = np.stack([prog.class_map(m) for m in maps])
    seeds = np.arange(E, dtype=np.uint32) * 7 + 3
    eng = BatchedMettaGrid(prog, cms, seeds, buffers=buffers)
    oracles = [op.OracleSim(prog, cms[i], int(seeds[i])) for i in range(E)]
    for o in oracles:  # the engine binds caller buffers after construction = a second set_buffers in the reference
        o.reinit_buffers()
    acts = [hp.make_actions(prog, i, steps, invalid) for i in range(E)]
    A = prog.num_agents

    def check(t):
        snap = eng.snapshot()
        for i, o in enumerate(oracles):
            mine = {k: v[i * A:(i + 1) * A] for k, v in snap.items()}
            hp.compare_snapshots(o.snapshot(), mine, f"{name} env {i} step {t}")

    check(0)
    for t in range(steps):
        a = np.concatenate([acts[i][0][t] for i in range(E)])
        v = np.concatenate([acts[i][1][t] for i in range(E)])
        if buffers == "device":
            import torch
            eng.actions.copy_(torch.from_numpy(a))
            eng.vibe_actions.copy_(torch.from_numpy(v))
            torch.cuda.synchronize()
        else:
            eng.actions[:] = a
            eng.vibe_actions[:] = v
        eng.step()
        for i, o in enumerate(oracles):
            o.step(acts[i][0][t], acts[i][1][t])
        check(t + 1)
    bits, first = eng.poll_errors()
    assert bits == 0, (bits, first)
    snap = eng.snapshot()
    for i, o in enumerate(oracles):
        mine = {k: v[i * A:(i + 1) * A] for k, v in snap.items()}
        pa = hp.payload_from_raw(prog, o.raw_objects(), o.current_stat_reward(), o.raw_stats(), o.snapshot(), steps, int(seeds[i]))
        pb = hp.payload_from_raw(prog, eng.raw_objects(i), eng.current_stat_reward(i), eng.raw_stats(i), mine, steps, int(seeds[i]))
        assert pa == pb, f"{name} env {i}: signature payload differs: {hp.diff_payload(pa, pb)}"
