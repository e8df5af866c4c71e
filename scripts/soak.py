"""Randomised parity soak (GPU box): many more seeds and steps than the committed goldens.  Every env of every scenario
is stepped on the HIP engine and on the oracle with its own random action trace; all caller-visible buffers are compared
after every step and the full signature payload at the end.  Usage: python scripts/soak.py [envs] [steps] [scenario ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import helpers as hp  # noqa: E402
import oracle_py as op  # noqa: E402
from mettagrid_amd.engine import BatchedMettaGrid  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 48
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
names = sys.argv[3:] or ["rung3", "rung4_full", "torture", "dynamic", "rung4", "thirteen", "minmax", "crowd"]
bad = 0
for name in names:
    spec_f, map_f, _, invalid = hp.SCENARIOS[name]
    spec = spec_f()
    maps = [map_f(1000 + s) for s in range(E)]
    prog = hp.compile_scenario(name, spec, *maps[0].shape)
    cms = np.stack([prog.class_map(m) for m in maps])
    seeds = np.arange(E, dtype=np.uint32) + 7000
    eng = BatchedMettaGrid(prog, cms, seeds, buffers="host")
    oracles = [op.OracleSim(prog, cms[i], int(seeds[i])) for i in range(E)]
    for o in oracles:
        o.reinit_buffers()
    acts = [hp.make_actions(prog, 500 + i, steps, invalid) for i in range(E)]
    A = prog.num_agents
    ok = True
    for t in range(steps):
        eng.actions[:] = np.concatenate([acts[i][0][t] for i in range(E)])
        eng.vibe_actions[:] = np.concatenate([acts[i][1][t] for i in range(E)])
        eng.step()
        snap = eng.snapshot()
        for i, o in enumerate(oracles):
            o.step(acts[i][0][t], acts[i][1][t])
            so = o.snapshot()
            mine = {k: v[i * A:(i + 1) * A] for k, v in snap.items()}
            try:
                hp.compare_snapshots(so, mine, f"{name} env {i} step {t + 1}")
            except AssertionError as ex:
                print("MISMATCH", str(ex)[:400])
                ok = False
                break
        if not ok:
            break
    if ok:
        for i, o in enumerate(oracles):
            pa = hp.payload_from_raw(prog, eng.raw_objects(i), eng.current_stat_reward(i), eng.raw_stats(i),
                                     {k: v[i * A:(i + 1) * A] for k, v in snap.items()}, steps, int(seeds[i]))
            pb = hp.payload_from_raw(prog, o.raw_objects(), o.current_stat_reward(), o.raw_stats(), o.snapshot(), steps, int(seeds[i]))
            if o.error & 1 and eng.poll_errors()[0] & 1:
                continue   # token budget exceeded on both sides: the reference raises there (mettagrid_c.cpp:364-375)
            if pa != pb:
                print("PAYLOAD MISMATCH", name, "env", i, hp.diff_payload(pa, pb)[:400])
                ok = False
                break
    bits, first = eng.poll_errors()
    print(("ok      " if ok else "FAILED  ") + f"{name}: {E} envs x {steps} steps, env error bits {bits}", flush=True)
    bad += 0 if ok else 1
    eng.close()
sys.exit(1 if bad else 0)
