"""TEST INFRASTRUCTURE ONLY — world-state parity of the sanitizer build (tests/cpu_emu/libmgx_emu.so) against the oracle.

The sanitizer build runs the construction and world-update kernels of mettagrid_amd/csrc on the host, one work-item
after another (tests/cpu_emu/hip/hip_runtime.h).  It has NO observation kernel (no observations, no rewards), so what is
compared after every step is the world state: every object (class, position, vibe, inventory in iteration order, tags),
every stat key/value except the ones the observation phase writes, and action_success.

  python tests/cpu_emu/run_emu.py [scenario ...]        (default: all scenarios of tests/helpers.py)
"""
from __future__ import annotations

import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import helpers as hp  # noqa: E402
import oracle_py as op  # noqa: E402
from mettagrid_amd import engine  # noqa: E402

OBS_PHASE_AGENT_STATS = {"cell.visited"}
OBS_PHASE_GAME_STATS = {"tokens_written", "tokens_dropped", "tokens_free_space"}


def use_emu_library() -> None:
    path = os.environ.get("MGX_EMU_LIB", os.path.join(HERE, "libmgx_emu.so"))
    real = engine.LIB_PATH
    engine.LIB_PATH = path
    engine._lib = None
    engine.load_lib()
    engine.LIB_PATH = real


def world_state(prog, raw_objects, raw_stats, success):
    gv, gt, av, at = (np.array(x) for x in raw_stats)
    for j, name in enumerate(prog.game_stat_names):
        if name in OBS_PHASE_GAME_STATS:
            gv[j] = 0; gt[j] = 0
    for j, name in enumerate(prog.agent_stat_names):
        if name in OBS_PHASE_AGENT_STATS:
            av[:, j] = 0; at[:, j] = 0
    return dict(objects=np.array(raw_objects), game_values=gv, game_touched=gt, agent_values=av, agent_touched=at,
                success=np.array(success))


def compare(prog, a: dict, b: dict, where: str) -> None:
    for k in a:
        if a[k].shape != b[k].shape or not np.array_equal(a[k], b[k]):
            msg = f"{where}: '{k}' differs"
            if a[k].shape == b[k].shape:
                idx = np.argwhere(a[k] != b[k])[0]
                msg += f" first at {idx.tolist()}: oracle {a[k][tuple(idx)]} vs engine {b[k][tuple(idx)]}"
                if k.startswith("agent_"):
                    msg += f" (stat '{prog.agent_stat_names[idx[1]]}')"
                if k.startswith("game_"):
                    msg += f" (stat '{prog.game_stat_names[idx[0]]}')"
                if k == "objects":
                    msg += f"\n oracle {a[k][idx[0]].tolist()}\n engine {b[k][idx[0]].tolist()}"
            else:
                msg += f" shapes {a[k].shape} vs {b[k].shape}"
            raise AssertionError(msg)


def run(name: str, E: int = 3, steps=None) -> None:
    spec_f, map_f, nsteps, invalid = hp.SCENARIOS[name]
    steps = nsteps if steps is None else steps
    spec = spec_f()
    maps = [map_f(s) for s in range(E)]
    prog = hp.compile_scenario(name, spec, *maps[0].shape)
    cms = np.stack([prog.class_map(m) for m in maps])
    seeds = np.arange(E, dtype=np.uint32) * 7 + 3
    eng = engine.BatchedMettaGrid(prog, cms, seeds, buffers="host")
    oracles = [op.OracleSim(prog, cms[i], int(seeds[i])) for i in range(E)]
    acts = [hp.make_actions(prog, i, steps, invalid) for i in range(E)]
    A = prog.num_agents

    def check(t):
        succ = eng.action_success()
        for i, o in enumerate(oracles):
            wa = world_state(prog, o.raw_objects(), o.raw_stats(), o.snapshot()["action_success"])
            wb = world_state(prog, eng.raw_objects(i), eng.raw_stats(i), succ[i * A:(i + 1) * A])
            compare(prog, wa, wb, f"{name} env {i} step {t}")

    check(0)
    for t in range(steps):
        eng.actions[:] = np.concatenate([acts[i][0][t] for i in range(E)])
        eng.vibe_actions[:] = np.concatenate([acts[i][1][t] for i in range(E)])
        eng.step()
        for i, o in enumerate(oracles):
            o.step(acts[i][0][t], acts[i][1][t])
        check(t + 1)
    bits, first = eng.poll_errors()
    assert bits == 0, f"{name}: env error bits {bits} (first env {first})"
    eng.close()


if __name__ == "__main__":
    use_emu_library()
    # minmax, delta: their rewards / observation values read an agent stat, and reading creates the stat key — in the
    # observation kernel, which this build does not run (README.md)
    names = sys.argv[1:] or [n for n in hp.SCENARIOS if n not in ("minmax", "delta")]
    for n in names:
        run(n)
        print("ok", n, flush=True)
