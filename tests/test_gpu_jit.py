"""GPU: run-time specialisation (mettagrid_amd/jit.py + mgx_attach_code): for programs the build did not specialise, the world
kernel with generated handler code and the observation kernel with the shape folded in are compiled on the spot, attached,
and must reproduce the oracle bit for bit."""
import glob
import json
import os

import numpy as np
import pytest

import helpers as hp
import oracle_py as op
import ref_tree
from mettagrid_amd import from_reference, jit
from mettagrid_amd.compiler import compile_spec
from mettagrid_amd.engine import BatchedMettaGrid

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _programs():
    """(name, program, class maps [E]) of three lean programs that are not build-time presets: the reference's arena and
    navigation configs (fixtures of its own converter) and the handler-VM torture scenario."""
    out = []
    for name in ("arena", "navigation"):
        doc = json.load(open(os.path.join(HERE, "golden", f"ref_{name}.json")))
        cfg = ref_tree.load(doc["config"])
        prog = from_reference.compile_reference_config(cfg, len(doc["map"]), len(doc["map"][0]))
        out.append((name, prog, np.stack([prog.class_map(doc["map"])] * 6)))
    spec_f, map_f, _, _ = hp.SCENARIOS["torture"]
    maps = [map_f(s) for s in range(6)]
    prog = compile_spec(spec_f(), *maps[0].shape)
    out.append(("torture", prog, np.stack([prog.class_map(m) for m in maps])))
    return out


def test_specialised_kernels_match_the_oracle(tmp_path, monkeypatch):
    monkeypatch.setenv("MGX_JIT_CACHE", str(tmp_path))
    progs = _programs()
    jobs = [jit.start(p, True) for _, p, _ in progs]      # all compiles side by side (the world unit takes a minute or two)
    for js in jobs:
        assert sorted(j.kind for j in js) == ["obs", "world"]
        for j in js:
            assert j.wait(900) and j.error is None, j.error
    for name, prog, cms in progs:
        E, A = cms.shape[0], prog.num_agents
        seeds = np.arange(E, dtype=np.uint32) + 3
        eng = BatchedMettaGrid(prog, cms, seeds, buffers="host", specialize="sync")     # finds the code objects in the cache
        assert eng.jit_errors == [] and not eng.jit_pending()
        assert eng.handler_variant == 9 and eng.obs_variant == 9, (name, eng.handler_variant, eng.obs_variant)
        plain = BatchedMettaGrid(prog, cms, seeds, buffers="host", specialize=False)
        assert plain.handler_variant == 0 and plain.obs_variant == 0
        oracles = [op.OracleSim(prog, cms[e], int(seeds[e])) for e in range(E)]
        for o in oracles:
            o.reinit_buffers()
        rng = np.random.default_rng(11)
        n = len(prog.action_names)
        for t in range(60):
            a = rng.integers(0, n, E * A).astype(np.int32)
            v = rng.integers(0, n, E * A).astype(np.int32)
            for g in (eng, plain):
                g.actions[:] = a
                g.vibe_actions[:] = v
                g.step()
            snap, snap0 = eng.snapshot(), plain.snapshot()
            for e, o in enumerate(oracles):
                o.step(a[e * A:(e + 1) * A], v[e * A:(e + 1) * A])
                hp.compare_snapshots(o.snapshot(), {k: x[e * A:(e + 1) * A] for k, x in snap.items()}, f"{name} jit env {e} step {t}")
            for k in snap:
                assert np.array_equal(snap[k], snap0[k]), (name, k, t)
        assert np.array_equal(eng.state_digests(), plain.state_digests())
        assert eng.poll_errors()[0] == 0
        eng.close(); plain.close()


def test_code_object_of_another_program_is_refused(tmp_path, monkeypatch):
    monkeypatch.setenv("MGX_JIT_CACHE", str(tmp_path))
    (_, p_arena, cms_arena), (_, p_nav, cms_nav), _ = _programs()
    jobs = jit.start(p_arena, True, kinds=("obs",))
    assert jobs[0].wait(600) and jobs[0].error is None
    eng = BatchedMettaGrid(p_nav, cms_nav, np.arange(6, dtype=np.uint32), buffers="host", specialize=False)
    rc = eng.L.mgx_attach_code(eng.h, jit.KINDS["obs"], jobs[0].path.encode())
    assert rc != 0 and b"another observation shape" in eng.L.mgx_last_error()
    assert eng.obs_variant == 0
    eng.step()
    eng.close()


def test_extended_dispatch_kernel_specialised_at_run_time(tmp_path, monkeypatch):
    """An extended program that is not the rung-4 preset (its chests mark a game stat): the lane-per-agent dispatch kernel is
    compiled with this program's handlers as straight-line code (csrc/mgx_jit_act_x.hip), attached, and must reproduce the
    oracle and the interpreter-driven kernel."""
    from mettagrid_amd import presets
    from mettagrid_amd import spec as S
    from mettagrid_amd.mapgen import random_map
    monkeypatch.setenv("MGX_JIT_CACHE", str(tmp_path))
    spec = presets.rung4_spec(obs_tokens=400)
    spec.objects["chest"].on_use = S.Handler([S.ResourceFilter(S.ACTOR, "ore", 2)],
                                             [S.ResourceTransfer(S.ACTOR, S.TARGET, "ore", -2),
                                              S.SetStat("chest.ore", S.InventoryValue("ore"), scope="game", entity=S.TARGET)], "deposit2")
    prog = compile_spec(spec, 20, 20, max_objects=presets.RUNG4_MAX_OBJECTS)
    objs = {"wall": 8, "extractor": 6, "chest": 4, "healer_red": 1, "healer_blue": 1, "healer_green": 1, "healer_yellow": 1,
            "flag_red": 1, "flag_blue": 1, "flag_green": 1, "flag_yellow": 1, "hub": 1, "wire": 3}
    E = 6
    cms = np.stack([prog.class_map(random_map(20, 20, dict(objs), dict(presets.RUNG4_AGENTS), 700 + s)) for s in range(E)])
    seeds = np.arange(E, dtype=np.uint32) + 21
    assert sorted(jit.plan(prog)) == ["actx"]
    # the reference's handler-chain config (fixture of its own converter: AoE, an event, a materialized query) compiles beside it
    doc = json.load(open(os.path.join(HERE, "golden", "ref_chains.json")))
    zc = np.load(os.path.join(HERE, "golden", "ref_chains.npz"))
    chains = from_reference.compile_reference_config(ref_tree.load(doc["config"]), len(doc["map"]), len(doc["map"][0]))
    pending = jit.start(prog, True) + jit.start(chains, True)
    for j in pending:
        assert j.wait(900) and j.error is None, j.error
    ce = BatchedMettaGrid(chains, chains.class_map(doc["map"])[None], [doc["seed"]], buffers="host", specialize="sync")
    assert ce.jit_errors == []
    assert ce.handler_variant == (9 if ce.act_variant == 1 else 0), (ce.act_variant, ce.handler_variant)
    keys = ("obs", "rewards", "terminals", "truncations", "action_success", "episode_rewards")
    for t in range(doc["steps"]):          # against the REFERENCE's recorded trace
        for when, agent_id, inv in doc["set_inventory"]:
            if when == t:
                ce.set_inventory(0, agent_id, dict(map(tuple, inv)))
        ce.actions[:] = zc["actions"][t]
        ce.vibe_actions[:] = zc["vibe_actions"][t]
        ce.step()
        hp.compare_snapshots({k: zc[k][t + 1] for k in keys}, ce.snapshot(), f"ref_chains (variant {ce.handler_variant}) step {t + 1}")
    ce.close()
    eng = BatchedMettaGrid(prog, cms, seeds, buffers="host", specialize="sync")
    assert eng.jit_errors == [] and eng.act_variant == 1 and eng.handler_variant == 9, (eng.jit_errors, eng.act_variant, eng.handler_variant)
    plain = BatchedMettaGrid(prog, cms, seeds, buffers="host", specialize=False)
    assert plain.handler_variant == 0
    oracles = [op.OracleSim(prog, cms[e], int(seeds[e])) for e in range(E)]
    for o in oracles:
        o.reinit_buffers()
    A, n = prog.num_agents, len(prog.action_names)
    rng = np.random.default_rng(4)
    for t in range(56):
        a = rng.integers(0, n, E * A).astype(np.int32)
        v = rng.integers(0, n, E * A).astype(np.int32)
        for g in (eng, plain):
            g.actions[:] = a
            g.vibe_actions[:] = v
            g.step()
        snap = eng.snapshot()
        for e, o in enumerate(oracles):
            o.step(a[e * A:(e + 1) * A], v[e * A:(e + 1) * A])
            hp.compare_snapshots(o.snapshot(), {k: x[e * A:(e + 1) * A] for k, x in snap.items()}, f"extended jit env {e} step {t}")
    assert np.array_equal(eng.state_digests(), plain.state_digests())
    eng.close(); plain.close()
