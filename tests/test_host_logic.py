"""Host logic: libstdc++ order emulation, observation pattern, compiler tables, map generation."""
import os

import numpy as np
import pytest

from mettagrid_amd import presets
from mettagrid_amd import spec as S
from mettagrid_amd.compiler import UnsupportedFeature, compile_spec, observation_offsets
from mettagrid_amd.fmt import K
from mettagrid_amd.mapgen import random_class_maps
from mettagrid_amd.umap import UMap, from_pydict
import helpers as hp

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_umap_matches_libstdcxx_golden_orders():
    """tests/golden/libstdcxx_umap_orders.txt was produced by gen_umap_orders.cpp with g++ 11.4 (the oracle toolchain)."""
    n = 0
    for line in open(os.path.join(GOLD, "libstdcxx_umap_orders.txt")):
        ops, res = line.split(" =")
        exp, nb = res.split("| nb")
        toks = ops.split()
        m = UMap()
        if int(toks[1]) >= 0:
            m.reserve(int(toks[1]))
        for t in toks[2:]:
            (m.erase if t[0] == "E" else m.insert)(int(t[1:]))
        assert m.keys() == [int(x) for x in exp.split()] and m.nb == int(nb), line
        n += 1
    assert n == 400


def test_umap_survey_vector():
    """SURVEY.md §7.3.2 verified vector: insert 3,0,7,1 -> 1 7 0 3; erase 0 + reinsert -> 0 1 7 3; 16 collides with 3."""
    m = UMap()
    for k in (3, 0, 7, 1):
        m.insert(k)
    assert m.keys() == [1, 7, 0, 3]
    m.erase(0)
    m.insert(0)
    assert m.keys() == [0, 1, 7, 3]
    m.insert(16)
    assert m.keys() == [0, 1, 7, 16, 3]
    m.insert(13)
    assert m.keys() == [13, 0, 1, 7, 16, 3]
    assert from_pydict([5, 2, 9]).nb == 3


def _sorted_reference(h, w):
    """compute_sorted_offsets of /root/reference/tests/test_observations.cpp:16-38 (the reference's own oracle)."""
    res = [(dr, dc) for dr in range(-(h // 2), h // 2 + 1) for dc in range(-(w // 2), w // 2 + 1)]
    return sorted(res, key=lambda a: (abs(a[0]) + abs(a[1]), a))


@pytest.mark.parametrize("hw", [(3, 9), (7, 3), (5, 5), (1, 1), (1, 5), (5, 1)])
def test_observation_pattern_matches_reference_offsets(hw):
    """tests/test_observations.cpp:84-104 MatchesReferenceOffsets — the unmasked pattern (mask disabled by testing
    shapes where every in-rectangle cell is compared after removing the mask)."""
    h, w = hw
    full = _sorted_reference(h, w)
    got = observation_offsets(h, w)
    assert got == [o for o in full if o in set(got)]          # order = reference order
    assert len(set(got)) == len(got)                            # unique (OffsetsAreUnique)
    d = [abs(a) + abs(b) for a, b in got]
    assert d == sorted(d)                                       # OffsetsInManhattanOrder


def test_observation_mask_sizes():
    """11x11 -> 89 cells (disc r=5 plus 8 tip cells), 13x13 -> 121... wait-free known counts."""
    assert len(observation_offsets(11, 11)) == 89
    assert len(observation_offsets(7, 7)) == 37
    assert len(observation_offsets(5, 5)) == 21
    assert len(observation_offsets(3, 3)) == 5
    assert observation_offsets(1, 1) == [(0, 0)]
    assert observation_offsets(11, 11)[:5] == [(0, 0), (-1, 0), (0, -1), (0, 1), (1, 0)]


def test_compiler_action_space_and_ids():
    prog = compile_spec(presets.rung3_spec(), 32, 32)
    assert prog.action_names[:3] == ["noop", "move_north", "move_south"]
    assert prog.action_names[-4:] == ["change_vibe_default", "change_vibe_a", "change_vibe_b", "change_vibe_c"]
    assert prog.type_names == sorted(prog.type_names) and prog.tag_names == sorted(prog.tag_names)
    assert "type:agent" in prog.tag_names and "team:red" in prog.tag_names
    f = prog.feature_ids
    assert (f["agent:group"], f["episode_completion_pct"], f["last_action"], f["last_reward"], f["goal"], f["vibe"],
            f["tag"], f["lp:east"], f["agent_id"]) == (0, 1, 2, 3, 4, 5, 6, 7, 11)
    assert f["inv:ore"] == 12 and f["inv:ore:p1"] == 13 and f["inv:hp"] == 14
    w = prog.words
    assert w[K.H_MAGIC] == K.MAGIC and w[K.H_TOTAL_WORDS] == w.size and w[K.H_NUM_OBS_OFFSETS] == 89
    assert w[K.H_HP_RESOURCE] == prog.resource_names.index("hp") and w[K.H_MAX_PRIORITY] == 1


def test_compiler_rejects_unsupported_features():
    spec = presets.rung2_spec()
    spec.resource_names = [f"r{i}" for i in range(14)]
    with pytest.raises(UnsupportedFeature):
        compile_spec(spec, 32, 32)
    spec = presets.rung2_spec()
    spec.obs = S.ObsSpec(width=17, height=17)
    with pytest.raises(RuntimeError, match="exceeds maximum packable size"):
        compile_spec(spec, 32, 32)


def test_unknown_map_cell_raises_like_reference():
    prog = compile_spec(presets.rung1_spec(), 16, 16)
    cells = presets.rung1_map()
    cells[3, 3] = "unicorn"
    with pytest.raises(RuntimeError, match="Unknown object type: unicorn"):
        prog.class_map(cells)


def test_random_class_maps_equal_reference_builder_semantics():
    prog = compile_spec(presets.rung3_spec(), 32, 32)
    cms = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, range(5))
    for s in range(5):
        assert np.array_equal(cms[s], prog.class_map(presets.rung3_map(s)))
    assert int((cms[0] > 0).sum()) == 124 + 40 + 8 + 4 + 16


def test_split_supervisor_actions_inplace_known_answers():
    """python/src/mettagrid/policy/supervisor_actions.py:8-40 restated: primary labels stay, vibe labels move to the vibe
    stream as engine action ids, anything outside [0, P + V) raises."""
    from mettagrid_amd.envs import split_supervisor_actions_inplace
    teacher = np.array([0, 4, 5, 7, 2, 6], np.int32)           # P = 5 primary actions, V = 3 vibes
    vibe = np.full(6, 99, np.int32)
    ids = np.array([9, 10, 11])                                # engine ids of change_vibe_<v>
    split_supervisor_actions_inplace(teacher, vibe, num_primary_actions=5, vibe_action_ids_by_index=ids)
    assert vibe.tolist() == [0, 0, 9, 11, 0, 10] and teacher.tolist() == [0, 4, 5, 7, 2, 6]
    with pytest.raises(ValueError, match="invalid action id 8 for agent 1"):
        split_supervisor_actions_inplace(np.array([1, 8], np.int32), np.zeros(2, np.int32), num_primary_actions=5,
                                         vibe_action_ids_by_index=ids)
    with pytest.raises(ValueError):
        split_supervisor_actions_inplace(np.array([-1], np.int32), np.zeros(1, np.int32), num_primary_actions=5,
                                         vibe_action_ids_by_index=ids)


def test_early_reset_draws_equal_numpy_generators():
    """mettagrid_amd/early_reset.py restates default_rng(seed).integers(1, max_steps + 1) (EarlyResetHandler,
    python/src/mettagrid/envs/early_reset_handler.py:6-22) for arrays of seeds: SeedSequence, PCG64, 32-bit Lemire."""
    from mettagrid_amd.early_reset import first_integers
    for high in (1, 2, 7, 1000, 12345, 2 ** 20):
        for seeds in (np.arange(3000), np.random.RandomState(high).randint(0, 2 ** 32, 3000, dtype=np.uint64)):
            want = [int(np.random.default_rng(int(s)).integers(1, high + 1)) for s in seeds]
            assert first_integers(seeds, high).tolist() == want


def test_batched_env_seeds_and_early_steps_are_vectorised_equivalents():
    from mettagrid_amd import presets
    from mettagrid_amd.compiler import compile_spec
    from mettagrid_amd.envs import MettaGridBatchedEnv
    spec = presets.rung2_spec()
    spec.max_steps = 77
    prog = compile_spec(spec, 32, 32)
    env = MettaGridBatchedEnv(prog, 500, map_fn=lambda e, ep: None, seed=2 ** 32 - 3)
    seeds = env._seeds()
    assert seeds.dtype == np.uint32 and seeds.tolist() == [(2 ** 32 - 3 + e) & 0xFFFFFFFF for e in range(500)]
    assert env.early_end_steps().tolist() == [int(np.random.default_rng(int(s)).integers(1, 78)) for s in seeds]
    custom = MettaGridBatchedEnv(prog, 5, map_fn=lambda e, ep: None, seed=10, seed_fn=lambda b, e, ep: b * 100 + e)
    assert custom._seeds().tolist() == [1000, 1001, 1002, 1003, 1004]


def test_state_digest_many_equals_scalar_digest():
    """signature.state_digest_many (what the long-horizon GPU test checks mgx_state_digests against) == state_digest per env,
    including envs with far-invalid action indices."""
    import helpers as hp
    import oracle_py as op
    from mettagrid_amd import signature as sg
    for name in ("rung3", "rung4", "rung1_invalid"):
        spec_f, map_f, steps, invalid = hp.SCENARIOS[name]
        dumps = []
        for seed in range(4):
            cells = map_f(seed)
            prog = hp.compile_scenario(name, spec_f(), *cells.shape)
            o = op.OracleSim(prog, prog.class_map(cells), seed)
            o.reinit_buffers()
            acts, vibes = hp.make_actions(prog, seed, 12 + seed, invalid)
            for t in range(12 + seed):
                o.step(acts[t] + (1000 if name == "rung1_invalid" and seed == 2 else 0), vibes[t])
            s = o.snapshot()
            dumps.append((o.raw_objects(), o.raw_stats(), s["episode_rewards"], s["action_success"], o.current_stat_reward(),
                          o.current_step, o.invalid_index_extra()))
        got = sg.state_digest_many(dumps)
        assert [int(g) for g in got] == [sg.state_digest(*d) for d in dumps]


def test_more_than_thirteen_resources_is_refused_cleanly_at_every_layer():
    """R <= 13 is the engine's inventory limit (one 4-bit id per item in the order word; 13 = the bucket count of libstdc++'s
    smallest unordered_map table, so insertion order is 'new node at the head').  The reference has no such limit
    (objects/inventory.hpp:45): a config with 14+ resources must be refused with a message that names the limit — by the spec
    compiler, by the lowering of a reference config tree, and by mgx_create itself when handed a forged program."""
    import ctypes as C
    import json
    import os

    import ref_tree
    from mettagrid_amd import engine, from_reference as fr, presets
    from mettagrid_amd import spec as S
    from mettagrid_amd.compiler import UnsupportedFeature, compile_spec
    from mettagrid_amd.fmt import K
    for n in (14, 29):
        sp = presets.rung2_spec()
        sp.resource_names = [f"r{i}" for i in range(n)]
        with pytest.raises(UnsupportedFeature, match="at most 13 resources"):
            compile_spec(sp, 32, 32)
    # through the reference's config tree (fixture made by the reference's own converter): resource list grown to 14
    here = os.path.dirname(os.path.abspath(__file__))
    doc = json.load(open(os.path.join(here, "golden", "ref_chains.json")))
    cfg = ref_tree.load(doc["config"])
    cfg.resource_names = list(cfg.resource_names) + [f"extra{i}" for i in range(14 - len(cfg.resource_names))]
    with pytest.raises(UnsupportedFeature, match="at most 13 resources"):
        fr.compile_reference_config(cfg, len(doc["map"]), len(doc["map"][0]))
    # a forged blob: header says 14 resources.  mgx_create validates before it touches the device.
    prog = compile_spec(presets.rung2_spec(), 32, 32)
    words = np.ascontiguousarray(prog.words, dtype=np.int32).copy()
    words[K.H_NUM_RESOURCES] = 14
    L = engine.load_lib()
    maps = np.zeros((1, 32, 32), np.uint16)
    seeds = np.zeros(1, np.uint32)
    h = C.c_void_p()
    rc = L.mgx_create(words.ctypes.data, words.size, maps.ctypes.data, seeds.ctypes.data, 1, 0, C.byref(h))
    assert rc == -3 and b"resources<=13" in L.mgx_last_error()


def test_handler_generator_is_deterministic_and_header_is_current():
    """build() regenerates csrc/mgx_handlers_gen.h from the presets; the text depends on nothing else, and the fingerprints
    the engine compares against are the ones of the programs the presets compile to today."""
    from mettagrid_amd import gen_handlers
    a, fa = gen_handlers.render()
    b, fb = gen_handlers.render()
    assert a == b and fa == fb
    assert open(gen_handlers.OUT).read() == a
    assert "MGX_GEN_R3_FP" in fa and "MGX_GEN_R4_FP" in fa and "0x0000000000000000" not in fa


def test_jit_plan_and_observation_unit_compiles(tmp_path, monkeypatch):
    """mettagrid_amd/jit.py: lean programs get a world and an observation code object, extended ones none; the observation
    unit (seconds to compile) goes through hipcc --genco here, the world unit (minutes) on the GPU box (tests/test_gpu_jit.py)."""
    from mettagrid_amd import jit, presets
    monkeypatch.setenv("MGX_JIT_CACHE", str(tmp_path))
    spec_f, map_f, _, _ = hp.SCENARIOS["torture"]
    prog = compile_spec(spec_f(), *map_f(0).shape)
    plan = jit.plan(prog)
    assert sorted(plan) == ["obs", "world"] and all(str(tmp_path) in plan[k][0] for k in plan)
    assert "MgxGenJ" in plan["world"][2][1] and any(a.startswith("-DMGX_JIT_FP=0x") for a in plan["world"][1])
    r4 = compile_spec(presets.rung4_spec(), 64, 64, max_objects=presets.RUNG4_MAX_OBJECTS)
    assert sorted(jit.plan(r4)) == ["actx"]          # extended programs: the dispatch kernel only
    (job,) = jit.start(prog, True, kinds=("obs",))
    assert job.wait(600) and job.error is None, job.error
    blob = open(job.path, "rb").read()
    assert b"mgx_jit_obs_r" in blob and b"mgx_jit_obs_n" in blob and b"mgx_jit_obs_info" in blob
    (again,) = jit.start(prog, True, kinds=("obs",))     # second request: straight from the cache
    assert again.ready() and again.path == job.path
