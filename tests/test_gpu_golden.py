"""GPU: the HIP engine (C ABI) against the committed golden vectors of the real reference engine."""
import glob
import json
import os

import numpy as np
import pytest

import helpers as hp
from mettagrid_amd import signature as sg
from mettagrid_amd.compiler import compile_spec
from mettagrid_amd.engine import BatchedMettaGrid

pytestmark = pytest.mark.gpu
GOLD = sorted(p for p in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz"))
              if not os.path.basename(p).startswith("ref_"))  # ref_*: fixtures of test_reference_config.py


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_hip_reproduces_reference_trace(path):
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    payload = json.loads(bytes(z["payload"]).decode())
    spec = hp.SCENARIOS[meta["scenario"]][0]()
    cm = z["class_map"]
    prog = hp.compile_scenario(meta["scenario"], spec, *cm.shape)
    # engine-internal buffers: exactly one initial-observation pass, like a bare reference MettaGrid(cfg, map, seed)
    eng = BatchedMettaGrid.__new__(BatchedMettaGrid)
    BatchedMettaGrid.__init__(eng, prog, cm[None], [meta["seed"]], buffers="host")
    # binding caller buffers re-initialises (a second set_buffers); account for its token statistics below
    keys = ("obs", "rewards", "terminals", "truncations", "action_success", "episode_rewards")
    hp.compare_snapshots({k: z[k][0] for k in keys}, eng.snapshot(), f"{meta['scenario']} step 0")
    for t in range(meta["steps"]):
        eng.actions[:] = z["actions"][t]
        eng.vibe_actions[:] = z["vibe_actions"][t]
        eng.step()
        hp.compare_snapshots({k: z[k][t + 1] for k in keys}, eng.snapshot(), f"{meta['scenario']} step {t + 1}")
    assert eng.poll_errors()[0] == 0
    mine = json.loads(json.dumps(hp.payload_from_raw(prog, eng.raw_objects(0), eng.current_stat_reward(0),
                                                    eng.raw_stats(0), eng.snapshot(), meta["steps"], meta["seed"])))
    # the fixture was recorded with ONE initial observation pass; this engine ran two (create + set_buffers):
    # remove the second pass' contribution to the token statistics before comparing (mettagrid_c.cpp:1165-1184).
    init_tokens = int((z["obs"][0][:, :, 0] != 0xFF).sum())
    A, T = prog.num_agents, prog.num_tokens
    game = dict(map(tuple, mine["stats"]["game"]))
    game["tokens_written"] = round(game["tokens_written"] - init_tokens, 8)
    game["tokens_free_space"] = round(game["tokens_free_space"] - (A * T - init_tokens), 8)
    mine["stats"]["game"] = [[k, v] for k, v in sorted(game.items())]
    assert mine == payload, hp.diff_payload(payload, mine)
    assert sg.signature(mine) == meta["signature"]
