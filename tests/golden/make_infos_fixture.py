"""BUILD-CONTAINER ONLY — golden `infos` of the reference's own StatsTracker.

    python tests/golden/make_infos_fixture.py

Replays the action traces of the committed fixtures ref_navigation / ref_chains (tests/golden/make_reference_fixtures.py:
same configs, seeds, actions and set_inventory calls) on the REFERENCE — its Python ``Simulation`` with a ``StatsTracker``
attached (python/src/mettagrid/envs/stats_tracker.py:26-76) and caller buffers bound the way ``MettaGridPufferEnv`` binds them
(mettagrid_puffer_env.py:200-216: a second ``set_buffers``), on the reference C++ engine oracle/_ref — until the episode ends,
and records what ``on_episode_end`` put into ``infos``: ``game``, ``agent``, ``per_agent``, the episode rewards and the
attributes that do not depend on the wall clock.  Output: tests/golden/infos_<scenario>.json (data only).
"""
from __future__ import annotations

import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_reference_fixtures as mrf  # noqa: E402


def run(name: str) -> dict:
    import numpy as np
    mrf.import_reference(shim=False)
    from mettagrid.envs.stats_tracker import StatsTracker
    from mettagrid.simulator import Simulation
    from mettagrid.simulator.simulator import Buffers
    from mettagrid.util.stats_writer import NoopStatsWriter
    cfg, seed, steps, _ = mrf.SCENARIOS[name]()
    z = np.load(os.path.join(HERE, f"ref_{name}.npz"))
    doc = json.load(open(os.path.join(HERE, f"ref_{name}.json")))
    A, T = z["obs"].shape[1], z["obs"].shape[2]
    bufs = Buffers(observations=np.zeros((A, T, 3), np.uint8), terminals=np.zeros(A, bool), truncations=np.zeros(A, bool),
                   rewards=np.zeros(A, np.float32), masks=np.ones(A, bool), actions=np.zeros(A, np.int32),
                   teacher_actions=np.zeros(A, np.int32), vibe_actions=np.zeros(A, np.int32))
    sim = Simulation(cfg, seed=seed, event_handlers=[StatsTracker(NoopStatsWriter())], buffers=bufs)
    c = sim._c_sim
    done_at = None
    for t in range(steps):
        for when, agent_id, inv in doc["set_inventory"]:
            if when == t:
                c.set_inventory(agent_id, {int(k): int(v) for k, v in inv})
        bufs.actions[:] = z["actions"][t]
        bufs.vibe_actions[:] = z["vibe_actions"][t]
        sim.step()
        if sim.is_done():
            done_at = t + 1
            break
    assert done_at is not None, f"{name}: the episode did not end within the fixture's trace"
    infos = sim._context["infos"]
    att = infos["attributes"]
    return {"scenario": name, "steps_played": done_at,
            "game": {k: float(v) for k, v in infos["game"].items()},
            "agent": {k: float(v) for k, v in infos["agent"].items()},
            "per_agent": {i: {k: float(v) for k, v in d.items()} for i, d in infos["per_agent"].items()},
            "episode_rewards": [float(x) for x in sim.episode_rewards],
            "attributes": {k: att[k] for k in ("seed", "map_w", "map_h", "steps", "max_steps")}}


def main() -> None:
    if len(sys.argv) == 3 and sys.argv[1] == "child":
        out = run(sys.argv[2])
        json.dump(out, open(os.path.join(HERE, f"infos_{sys.argv[2]}.json"), "w"), separators=(",", ":"), sort_keys=True)
        return
    import subprocess
    for name in ("navigation", "chains"):
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "child", name])
        p = os.path.join(HERE, f"infos_{name}.json")
        d = json.load(open(p))
        print(p, os.path.getsize(p), "bytes; ended at step", d["steps_played"], "; game keys", len(d["game"]), "; agent keys", len(d["agent"]))


if __name__ == "__main__":
    main()
