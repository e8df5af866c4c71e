"""GPU parity at batch sizes that cross workgroup / wavefront boundaries, and at BASELINE.json's full size.

The small-batch parity tests run 6 envs (one partial wavefront).  Here: (1) 200 envs — envs on both sides of every
32-lane wavefront and 64-env workgroup boundary plus the ragged last workgroup are compared with the oracle after every
step; (2) the full 65 536-env rung-3 workload of the bench: a spread of sampled envs against the oracle, and
size-independent well-formedness of EVERY observation row (tokens are a prefix, 0xFF padding after it, token counts
equal to the tokens_written game stat delta is covered by the sampled parity).
"""
import numpy as np
import pytest

import helpers as hp
import oracle_py as op
from mettagrid_amd import presets
from mettagrid_amd.compiler import compile_spec
from mettagrid_amd.engine import BatchedMettaGrid
from mettagrid_amd.mapgen import random_class_maps

pytestmark = pytest.mark.gpu


def _run_and_check(E, sample, steps, seed0=0, rung=3, actions=None, obs_tokens=None, generic=False):
    import torch
    if rung == 2:
        spec = presets.rung2_spec()
        prog = compile_spec(spec, 32, 32, max_objects=192)
        cms = np.stack([prog.class_map(presets.rung2_map(seed0 + e)) for e in range(E)])
    elif rung == 3:
        spec = presets.rung3_spec()
        prog = compile_spec(spec, 32, 32, max_objects=192)
        cms = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8},
                                range(seed0, seed0 + E))
    else:
        spec = presets.rung4_spec() if obs_tokens is None else presets.rung4_spec(obs_tokens=obs_tokens)
        prog = compile_spec(spec, 64, 64, max_objects=presets.RUNG4_MAX_OBJECTS)
        distinct = min(E, 4096)   # one numpy generator per map on the host: a minute for 65 536 maps; seeds stay per env
        cms = random_class_maps(prog, 64, 64, presets.RUNG4_OBJECTS, presets.RUNG4_AGENTS, range(seed0, seed0 + distinct))
        if distinct < E:
            cms = cms[np.arange(E) % distinct]
    A, T = prog.num_agents, prog.num_tokens
    seeds = np.arange(seed0, seed0 + E, dtype=np.uint32)
    eng = BatchedMettaGrid(prog, cms, seeds, buffers="device", specialize=False)
    oracles = {i: op.OracleSim(prog, cms[i], int(seeds[i])) for i in sample}
    for o in oracles.values():
        o.reinit_buffers()
    n_act = len(prog.action_names)
    rng = np.random.default_rng(1234)

    def rows(t, i):
        return t[i * A:(i + 1) * A].cpu().numpy()

    def check(step):
        eng.sync()
        succ, ep = eng.action_success(), eng.episode_rewards()
        for i, o in oracles.items():
            mine = dict(obs=rows(eng.obs, i), rewards=rows(eng.rewards, i), terminals=rows(eng.terminals, i).astype(bool),
                        truncations=rows(eng.truncations, i).astype(bool), action_success=succ[i * A:(i + 1) * A],
                        episode_rewards=ep[i * A:(i + 1) * A])
            hp.compare_snapshots(o.snapshot(), mine, f"E={E} env {i} step {step}")

    check(0)
    for t in range(steps):
        a = rng.integers(-1, n_act + 1, size=E * A).astype(np.int32)   # includes invalid ids on both sides
        v = rng.integers(0, n_act, size=E * A).astype(np.int32)
        if actions is not None:
            a, v = actions(t, a, v)
        eng.actions.copy_(torch.from_numpy(a))
        eng.vibe_actions.copy_(torch.from_numpy(v))
        torch.cuda.synchronize()
        eng.step()
        for i, o in oracles.items():
            o.step(a[i * A:(i + 1) * A], v[i * A:(i + 1) * A])
        check(t + 1)
    bits, first = eng.poll_errors()
    assert bits == 0, (bits, first)
    if generic:
        assert (eng.obs_variant, eng.handler_variant) == (0, 0), (eng.obs_variant, eng.handler_variant)
    elif obs_tokens is None and E >= 150 and rung != 2:   # configs[2] at its own shape runs the specialised observation kernel
        assert eng.obs_variant == (3 if rung == 3 else 0), eng.obs_variant
    return eng, T


def test_wavefront_and_workgroup_boundaries():
    _run_and_check(200, [0, 1, 31, 32, 33, 63, 64, 65, 95, 96, 127, 128, 191, 192, 199], steps=6)


def test_full_size_sampled_parity_and_row_wellformedness():
    import torch
    E = 65536
    sample = [0, 63, 64, 4095, 4096, 12345, 32767, 32768, 50001, 65535 - 64, 65535]
    eng, T = _run_and_check(E, sample, steps=3, seed0=0)
    obs = eng.obs                                  # [E*A, T, 3] u8 on the GPU
    empty = (obs == 0xFF).all(dim=2)               # token slot unused
    # a row is a run of tokens followed only by padding: once empty, always empty
    assert not (empty[:, :-1] & ~empty[:, 1:]).any().item()
    # padding is all-0xFF triples and every used token has a location byte other than 0xFF
    assert ((obs[..., 0] == 0xFF) == empty).all().item()
    # every agent sees at least its global tokens and itself
    assert (~empty[:, 0]).all().item()
    assert torch.isfinite(eng.rewards).all().item()


@pytest.mark.parametrize("rung", [3, 4])
def test_full_size_generic_kernels_against_the_oracle(rung, monkeypatch):
    """The kernels ANY program gets — observation shape read at run time, handler interpreter — at BASELINE's full size against
    the oracle (the specialised instances are what the two tests around this one run)."""
    monkeypatch.setenv("MGX_NO_GEN", "1")
    monkeypatch.setenv("MGX_OBS_GENERIC", "1")
    sample = [0, 63, 64, 4097, 32767, 32768, 50001, 65535]
    _run_and_check(65536, sample, steps=3, rung=rung, generic=True)


def test_rung2_4096_envs_first_32_envs_128_steps():
    """SURVEY.md 8d rung 2 (BASELINE.json configs[1]): 4 096 envs x 32x32, 16 agents, move + change_vibe; the first 32 envs
    against the oracle after every one of 128 steps, then the state digest of ALL 4 096 envs against envs replayed on the
    oracle's side of the engine boundary (same engine code, 32-env batches cannot hide a cross-env leak at 4 096)."""
    import torch
    eng, _ = _run_and_check(4096, list(range(32)), steps=128, rung=2)
    big = eng.state_digests()
    # the same 4 096 envs, stepped in batches of 512 with the same action stream: equal digests env by env
    prog = eng.prog
    A, n_act = prog.num_agents, len(prog.action_names)
    cms = np.stack([prog.class_map(presets.rung2_map(e)) for e in range(4096)])
    for b0 in range(0, 4096, 512):
        part = BatchedMettaGrid(prog, cms[b0:b0 + 512], np.arange(b0, b0 + 512, dtype=np.uint32), buffers="device", specialize=False)
        rng = np.random.default_rng(1234)
        for t in range(128):
            a = rng.integers(-1, n_act + 1, size=4096 * A).astype(np.int32)
            v = rng.integers(0, n_act, size=4096 * A).astype(np.int32)
            part.actions.copy_(torch.from_numpy(a[b0 * A:(b0 + 512) * A]))
            part.vibe_actions.copy_(torch.from_numpy(v[b0 * A:(b0 + 512) * A]))
            torch.cuda.synchronize()
            part.step()
        assert np.array_equal(part.state_digests(), big[b0:b0 + 512]), b0
        part.close()


def test_rung4_boundaries():
    """BASELINE.json configs[3] rules and shape (64x64, 64 agents) across wavefront / workgroup boundaries; 55 steps reach
    the first timestep event (t = 50, max_targets with the env's RNG)."""
    _run_and_check(150, [0, 31, 32, 63, 64, 65, 127, 128, 149], steps=55, rung=4)


def test_rung4_full_size_sampled_parity_and_row_wellformedness():
    """configs[3] at its full size: 65 536 envs x 64x64 x 64 agents, AoE + territory + events (4.2 M agents per step)."""
    E = 65536
    sample = [0, 63, 64, 4097, 32767, 32768, 50001, 65535]
    eng, T = _run_and_check(E, sample, steps=3, rung=4)
    obs = eng.obs
    empty = (obs == 0xFF).all(dim=2)
    assert not (empty[:, :-1] & ~empty[:, 1:]).any().item()
    assert ((obs[..., 0] == 0xFF) == empty).all().item()
    assert (~empty[:, 0]).all().item()


def test_large_rectangular_map():
    """80 x 96 map (grid larger than one staging pass of the observation kernel, H != W), 600 object slots."""
    import torch
    spec = presets.rung3_spec()
    H, W = 80, 96
    prog = compile_spec(spec, H, W, max_objects=600)
    A = prog.num_agents
    E = 3
    cms = random_class_maps(prog, H, W, {"wall": 150, "extractor": 30, "chest": 10}, {"red": 8, "blue": 8}, range(E))
    seeds = np.arange(E, dtype=np.uint32) + 11
    eng = BatchedMettaGrid(prog, cms, seeds, buffers="device")
    oracles = [op.OracleSim(prog, cms[i], int(seeds[i])) for i in range(E)]
    for o in oracles:
        o.reinit_buffers()
    rng = np.random.default_rng(5)
    n_act = len(prog.action_names)
    for t in range(8):
        snap = eng.snapshot()
        for i, o in enumerate(oracles):
            hp.compare_snapshots(o.snapshot(), {k: v[i * A:(i + 1) * A] for k, v in snap.items()}, f"large map env {i} step {t}")
        a = rng.integers(0, n_act, E * A).astype(np.int32)
        v = rng.integers(0, n_act, E * A).astype(np.int32)
        eng.actions.copy_(torch.from_numpy(a)); eng.vibe_actions.copy_(torch.from_numpy(v)); torch.cuda.synchronize()
        eng.step()
        for i, o in enumerate(oracles):
            o.step(a[i * A:(i + 1) * A], v[i * A:(i + 1) * A])
    assert eng.poll_errors()[0] == 0


@pytest.mark.parametrize("obs_tokens", [200, 400])
def test_rung4_other_token_budgets(obs_tokens):
    """The 512-thread observation kernel picks its number of encode wavefronts by what fits in LDS (4 at T = 200, 3 at the
    preset's 256, 2 at 400): every instantiation against the oracle."""
    _run_and_check(40, [0, 1, 31, 32, 39], 12, seed0=300, rung=4, obs_tokens=obs_tokens)
