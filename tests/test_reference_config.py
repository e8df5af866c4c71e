"""The drop-in boundary on the CONFIG side: config trees produced by the reference's own converter
(``mettagrid/config/mettagrid_c_config.py:576-1007``, run on ``mettagrid_amd.mettagrid_c`` records by
tests/golden/make_reference_fixtures.py) are lowered by ``mettagrid_amd.from_reference``, the ids the compiler assigns
are checked against the reference's, and the CPU oracle reproduces the reference engine's trace and episode signature —
including the fixed scenario of /root/reference/scripts/deterministic_episode_signature.py."""
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers as hp
import oracle_py as op
import ref_tree
from mettagrid_amd import from_reference as fr
from mettagrid_amd import signature as sg

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = sorted(glob.glob(os.path.join(HERE, "golden", "ref_*.json")))
KEYS = ("obs", "rewards", "terminals", "truncations", "action_success", "episode_rewards")
# SHA-256 the reference script prints here (container toolchain g++ 11.4)
SCRIPT_SIGNATURE = "0a8fe5cd26e34f712ba3035e386418fe428a5cdf268a92f7efd8658d5fef4d24"


def load_fixture(path):
    doc = json.load(open(path))
    cfg = ref_tree.load(doc["config"])
    trace = np.load(path[:-5] + ".npz")
    return doc, cfg, trace


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[4:-5] for p in FIX])
def test_oracle_reproduces_reference_from_converted_config(path):
    doc, cfg, z = load_fixture(path)
    cells = np.asarray(doc["map"], dtype=object)
    prog = fr.compile_reference_config(cfg, *cells.shape)       # includes the id checks
    assert prog.action_names == doc["action_names"]
    assert prog.resource_names == doc["resource_names"]
    assert prog.type_names == [n for n in doc["object_type_names"] if n]  # (the reference pads the list, mettagrid_c.cpp:204)
    o = op.OracleSim(prog, prog.class_map(cells), doc["seed"])
    hp.compare_snapshots({k: z[k][0] for k in KEYS}, o.snapshot(), f"{doc['scenario']} step 0")
    for t in range(doc["steps"]):
        for when, agent_id, inv in doc["set_inventory"]:
            if when == t:
                o.set_inventory(agent_id, dict(map(tuple, inv)))
        o.step(z["actions"][t], z["vibe_actions"][t])
        hp.compare_snapshots({k: z[k][t + 1] for k in KEYS}, o.snapshot(), f"{doc['scenario']} step {t + 1}")
    mine = hp.payload_from_raw(prog, o.raw_objects(), o.current_stat_reward(), o.raw_stats(), o.snapshot(), o.current_step,
                               doc["seed"])
    mine = json.loads(json.dumps(mine))
    assert mine == doc["payload"], hp.diff_payload(doc["payload"], mine)
    assert sg.signature(mine) == doc["signature"]
    assert o.error == 0


def test_signature_script_scenario_hash():
    doc = json.load(open(os.path.join(HERE, "golden", "ref_signature.json")))
    assert doc["signature"] == SCRIPT_SIGNATURE      # what the reference engine produced when the fixture was made


def test_id_mismatch_is_an_error():
    doc, cfg, _ = load_fixture(os.path.join(HERE, "golden", "ref_chains.json"))
    cells = np.asarray(doc["map"], dtype=object)
    cfg.feature_ids["vibe"] += 1
    with pytest.raises(fr.ReferenceConfigError):
        fr.compile_reference_config(cfg, *cells.shape)


def test_record_classes_cover_what_the_converter_imports():
    """Every name the reference's converter modules import from ``mettagrid.mettagrid_c`` exists in the drop-in."""
    from mettagrid_amd import mettagrid_c as C
    needed = """ActionConfig AgentConfig AOEConfig AttackActionConfig AttackOutcome ChangeVibeActionConfig ClosureQueryConfig
    EventConfig FilteredQueryConfig GameConfig GameValueFilterConfig GlobalObsConfig GridObjectConfig Handler HandlerConfig
    HandlerMode InventoryConfig LimitDef MaterializedQueryTag MaxDistanceFilterConfig MoveActionConfig MultiHandler
    NegFilterConfig ObsValueConfig OrFilterConfig PeriodicFilterConfig QueryOrderBy RaycastQueryConfig ResourceDelta
    ResourceFilterConfig RewardConfig RewardEntry SharedTagPrefixFilterConfig TagPrefixFilterConfig TagQueryConfig
    TargetIsUsableFilterConfig TargetLocEmptyFilterConfig TerritoryConfig TerritoryControlConfig VibeFilterConfig WallConfig
    make_query_config AddTagMutationConfig ChangeVibeMutationConfig ClearInventoryMutationConfig EntityRef
    GameValueMutationConfig PushObjectMutationConfig QueryInventoryMutationConfig RaycastSpawnMutationConfig
    RecomputeMaterializedQueryMutationConfig RelocateMutationConfig RemoveTagMutationConfig RemoveTagsWithPrefixMutationConfig
    ResourceDeltaMutationConfig ResourceTransferMutationConfig SpawnObjectMutationConfig StatsEntity StatsMutationConfig
    StatsTarget SwapMutationConfig UseTargetMutationConfig ConstValueConfig GameValueScope InventoryValueConfig MaxValueConfig
    MinValueConfig QueryCountValueConfig QueryInventoryValueConfig RatioValueConfig StatValueConfig SumValueConfig
    PackedCoordinate dtype_observations dtype_terminals dtype_truncations dtype_rewards dtype_actions dtype_masks dtype_success
    EpisodeStats AttackMutationConfig QueryResourceFilterConfig QueryConfigHolder""".split()
    missing = [n for n in needed if not hasattr(C, n)]
    assert not missing, missing
    assert C.PackedCoordinate.pack(5, 10) == 90 and C.PackedCoordinate.unpack(90) == (5, 10)
    with pytest.raises(ValueError):
        C.PackedCoordinate.pack(15, 0)


@pytest.mark.skipif(not os.path.isdir("/root/reference/python/src"), reason="needs the reference checkout (build container)")
@pytest.mark.parametrize("name", ["signature", "chains"])
def test_committed_tree_equals_a_fresh_conversion(name, tmp_path):
    """The committed config tree is what the reference's converter produces TODAY from the drop-in's classes."""
    gen = os.path.join(HERE, "golden", "make_reference_fixtures.py")
    prefix = str(tmp_path / ("ref_" + name))
    subprocess.check_call([sys.executable, gen, "shim", name, prefix])
    fresh = json.load(open(prefix + ".shim.json"))
    committed = json.load(open(os.path.join(HERE, "golden", f"ref_{name}.json")))
    assert fresh["config"] == committed["config"]
    assert fresh["map"] == committed["map"] and fresh["action_names"] == committed["action_names"]
