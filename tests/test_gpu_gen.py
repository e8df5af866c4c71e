"""GPU: the handler code generated at build() for the benchmark presets (mettagrid_amd/gen_handlers.py ->
csrc/mgx_handlers_gen.h) against the handler interpreter it replaces: same program, maps, seeds and action traces through
both, equal state digests; and the selection rule — only a program whose handler tables hash to the preset's fingerprint
runs generated code."""
import numpy as np
import pytest

import helpers as hp
from mettagrid_amd import gen_handlers, presets
from mettagrid_amd.compiler import compile_spec
from mettagrid_amd.engine import BatchedMettaGrid
from mettagrid_amd.mapgen import random_class_maps, random_map

pytestmark = pytest.mark.gpu


def _preset(rung):
    if rung == 3:
        prog = compile_spec(presets.rung3_spec(), 32, 32, max_objects=192)
        maps = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, range(1024))
    else:
        prog = compile_spec(presets.rung4_spec(), 64, 64, max_objects=presets.RUNG4_MAX_OBJECTS)
        maps = random_class_maps(prog, 64, 64, presets.RUNG4_OBJECTS, presets.RUNG4_AGENTS, range(512))
    return prog, maps


@pytest.mark.parametrize("rung,lean_act", [(3, False), (3, True), (4, False)])
def test_generated_handlers_equal_interpreter_by_digest(rung, lean_act, monkeypatch):
    import torch
    E, steps = 8192, 80
    prog, maps = _preset(rung)
    cms = maps[np.arange(E) % len(maps)]
    seeds = np.arange(E, dtype=np.uint32) + 11
    if lean_act:
        monkeypatch.setenv("MGX_ACT_LEAN", "1")
    gen = BatchedMettaGrid(prog, cms, seeds, buffers="device")
    monkeypatch.setenv("MGX_NO_GEN", "1")
    itp = BatchedMettaGrid(prog, cms, seeds, buffers="device")
    assert (gen.handler_variant, itp.handler_variant) == (rung, 0)
    A, n_act = prog.num_agents, len(prog.action_names)
    g = torch.Generator(device="cuda").manual_seed(31 + rung)
    for t in range(steps):
        a = torch.randint(-1, n_act + 1, (E * A,), dtype=torch.int32, device="cuda", generator=g)
        v = torch.randint(0, n_act, (E * A,), dtype=torch.int32, device="cuda", generator=g)
        for eng in (gen, itp):
            eng.actions.copy_(a); eng.vibe_actions.copy_(v)
            eng.wait_for_caller(); eng.step(); eng.caller_waits()
        if (t + 1) % 10 == 0:
            bad = np.nonzero(gen.state_digests() != itp.state_digests())[0]
            assert len(bad) == 0, f"rung {rung} step {t + 1}: envs {bad[:8].tolist()} differ ({len(bad)})"
    assert gen.poll_errors()[0] == 0 and itp.poll_errors()[0] == 0
    gen.close(); itp.close()


def test_selection_by_fingerprint():
    """The presets at any map size share their handler tables (generated code applies); a scenario with other handlers, or
    the rung-3 rules with one mutation changed, runs the interpreter."""
    def variant(spec, h, w, cells):
        prog = compile_spec(spec, h, w, max_objects=presets.RUNG4_MAX_OBJECTS)
        eng = BatchedMettaGrid(prog, prog.class_map(cells)[None], [1], buffers="host")
        v = eng.handler_variant
        eng.close()
        return v
    small = random_map(11, 11, {"wall": 3, "extractor": 2, "chest": 1}, {"red": 8, "blue": 8}, 1)
    assert variant(presets.rung3_spec(), 11, 11, small) == 3
    assert variant(presets.rung3_spec(use_attack_mutation=False), 11, 11, small) == 0
    spec_f, map_f, _, _ = hp.SCENARIOS["torture"]
    m = map_f(0)
    assert variant(spec_f(), *m.shape, m) == 0


def test_preset_shape_with_an_episode_length_runs_its_own_instance():
    """The env wrappers run the presets with max_steps set: same shape, so the observation kernel instance compiled for it
    (max_steps read at run time, obs_variant 5) — against the oracle across a truncation."""
    from test_gpu_act import _against_oracle
    spec = presets.rung3_spec()
    spec.max_steps = 9
    prog = compile_spec(spec, 32, 32, max_objects=192)
    E = 6
    cms = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, range(E))
    eng = BatchedMettaGrid(prog, cms, np.arange(E, dtype=np.uint32), buffers="host")
    assert eng.obs_variant == 5 and eng.handler_variant == 3
    eng.close()
    _against_oracle(prog, cms, np.arange(E, dtype=np.uint32) + 2, 12, 4, "rung 3 with max_steps", 0)
