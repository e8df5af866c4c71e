"""Debug helper: step a scenario on the HIP engine and the oracle side by side and report the first difference."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import helpers as hp  # noqa: E402
import oracle_py as op  # noqa: E402
from mettagrid_amd.engine import BatchedMettaGrid  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "rung4_full"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 5
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
spec_f, map_f, _, invalid = hp.SCENARIOS[name]
spec = spec_f()
maps = [map_f(s) for s in range(E)]
prog = hp.compile_scenario(name, spec, *maps[0].shape)
cms = np.stack([prog.class_map(m) for m in maps])
seeds = np.arange(E, dtype=np.uint32) + 21
eng = BatchedMettaGrid(prog, cms, seeds, buffers="host")
oracles = [op.OracleSim(prog, cms[i], int(seeds[i])) for i in range(E)]
for o in oracles:
    o.reinit_buffers()
acts = [hp.make_actions(prog, i, steps, invalid) for i in range(E)]
A = prog.num_agents
prev_objs = [None] * E
for t in range(steps):
    eng.actions[:] = np.concatenate([acts[i][0][t] for i in range(E)])
    eng.vibe_actions[:] = np.concatenate([acts[i][1][t] for i in range(E)])
    eng.step()
    snap = eng.snapshot()
    for i, o in enumerate(oracles):
        o.step(acts[i][0][t], acts[i][1][t])
        so = o.snapshot()
        pa = hp.payload_from_raw(prog, eng.raw_objects(i), eng.current_stat_reward(i), eng.raw_stats(i),
                                 {k: v[i * A:(i + 1) * A] for k, v in snap.items()}, t + 1, int(seeds[i]))
        pb = hp.payload_from_raw(prog, o.raw_objects(), o.current_stat_reward(), o.raw_stats(), so, t + 1, int(seeds[i]))
        if pa != pb:
            print(f"env {i} step {t + 1}: payload differs:\n{hp.diff_payload(pa, pb)}")
            from mettagrid_amd.signature import objects_from_raw
            now_e = objects_from_raw(prog, eng.raw_objects(i))
            now_o = objects_from_raw(prog, o.raw_objects())
            bad = [k for k in now_o if now_o[k]["location"] != now_e.get(k, {}).get("location")]
            print("objects at different places:", bad)
            for k in bad:
                c0, r0 = prev_objs[i][k]["location"]
                print("object", k, "agent", now_o[k].get("agent_id"), "was at (r,c)", (r0, c0), "oracle now", now_o[k]["location"][::-1], "engine now", now_e[k]["location"][::-1])
                for k2, ob in prev_objs[i].items():
                    c2, r2 = ob["location"]
                    if abs(r2 - r0) <= 2 and abs(c2 - c0) <= 2 and k2 != k:
                        aid = ob.get("agent_id")
                        print("   near:", k2, ob["type_name"], "agent", aid, "was (r,c)", (r2, c2), "oracle now", now_o[k2]["location"][::-1],
                              "engine now", now_e[k2]["location"][::-1], "action", prog.action_names[acts[i][0][t][aid]] if aid is not None else None,
                              "vibe-stream", prog.action_names[acts[i][1][t][aid]] if aid is not None else None)
            sys.exit(1)
        from mettagrid_amd.signature import objects_from_raw as _ofr
        prev_objs[i] = _ofr(prog, o.raw_objects())
print("no difference")
