"""Developer check: the lane-per-agent action dispatch (mgx_act.h) against the lane-per-env kernels, same inputs, digest per
step; at the first difference the offending env is dumped from both engines.
Usage (GPU box): python scripts/act_diff.py [rung] [envs] [steps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mettagrid_amd import presets  # noqa: E402
from mettagrid_amd.compiler import compile_spec  # noqa: E402
from mettagrid_amd.engine import BatchedMettaGrid  # noqa: E402
from mettagrid_amd.mapgen import random_class_maps  # noqa: E402

rung = int(sys.argv[1]) if len(sys.argv) > 1 else 4
E = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 120
if rung == 3:
    prog = compile_spec(presets.rung3_spec(), 32, 32, max_objects=192)
    maps = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, range(min(E, 2048)))
else:
    prog = compile_spec(presets.rung4_spec(), 64, 64, max_objects=presets.RUNG4_MAX_OBJECTS)
    maps = random_class_maps(prog, 64, 64, presets.RUNG4_OBJECTS, presets.RUNG4_AGENTS, range(min(E, 1024)))
cms = maps[np.arange(E) % len(maps)]
seeds = np.arange(E, dtype=np.uint32)
A, n_act = prog.num_agents, len(prog.action_names)
par = BatchedMettaGrid(prog, cms, seeds, buffers="device")
os.environ["MGX_ACT_SERIAL"] = "1"
os.environ["MGX_NO_GEN"] = "1"          # the baseline: lane per env, handler interpreter
ser = BatchedMettaGrid(prog, cms, seeds, buffers="device")
del os.environ["MGX_ACT_SERIAL"], os.environ["MGX_NO_GEN"]
print("dispatch variants:", par.act_variant, ser.act_variant, "handler variants:", par.handler_variant, ser.handler_variant)
gen = torch.Generator(device="cuda").manual_seed(99 + rung)
DBG_ENV = int(os.environ.get("ACT_DBG_ENV", "-1"))
DBG_STEP = int(os.environ.get("ACT_DBG_STEP", "-1"))
for t in range(steps):
    if t + 1 == DBG_STEP:
        par.sync()
        par.L.mgx_debug_act_env(DBG_ENV)
        print("objects before", par.raw_objects(DBG_ENV)[:, :8].tolist())
    a = torch.randint(-1, n_act + 1, (E * A,), dtype=torch.int32, device="cuda", generator=gen)
    v = torch.randint(0, n_act, (E * A,), dtype=torch.int32, device="cuda", generator=gen)
    for eng in (par, ser):
        eng.actions.copy_(a); eng.vibe_actions.copy_(v)
        eng.wait_for_caller(); eng.step(); eng.caller_waits()
    if t + 1 == DBG_STEP:
        import ctypes
        buf = (ctypes.c_uint32 * 192)()
        par.L.mgx_debug_act_read(buf)
        for p_ in range(64):
            print(f"pos {p_:2d} agent {buf[p_] & 0xFF:2d} round {buf[p_] >> 8} own ({buf[64 + p_] >> 24},{(buf[64 + p_] >> 16) & 0xFF}) tgt ({(buf[64 + p_] >> 8) & 0xFF},{buf[64 + p_] & 0xFF}) act {buf[128 + p_]}")
        par.L.mgx_debug_act_env(-1)
    if (t + 1) % int(os.environ.get("ACT_DIFF_EVERY", "1")) and t + 1 < steps:
        continue
    dp, ds = par.state_digests(), ser.state_digests()
    bad = np.nonzero(dp != ds)[0]
    if len(bad):
        e = int(bad[0])
        print(f"step {t + 1}: {len(bad)} envs differ, first {e}")
        op_, os_ = par.raw_objects(e), ser.raw_objects(e)
        sp, ss = par.raw_stats(e), ser.raw_stats(e)
        acts = a.view(E, A)[e].cpu().numpy()
        print("actions", acts.tolist())
        print("action names", prog.action_names)
        if op_.shape != os_.shape or not np.array_equal(op_, os_):
            rows = np.nonzero((op_ != os_).any(axis=tuple(range(1, op_.ndim))))[0] if op_.shape == os_.shape else []
            for r in rows[:8]:
                print("obj row", r, "\n par", op_[r].tolist(), "\n ser", os_[r].tolist())
        for k in range(len(sp)):
            if not np.array_equal(np.asarray(sp[k]), np.asarray(ss[k])):
                x, y = np.asarray(sp[k]), np.asarray(ss[k])
                idx = np.argwhere(x != y)[:10] if x.shape == y.shape else None
                print("stats part", k, x.shape, y.shape, None if idx is None else [(tuple(i), x[tuple(i)], y[tuple(i)]) for i in idx])
        succ_p, succ_s = par.action_success()[e * A:(e + 1) * A], ser.action_success()[e * A:(e + 1) * A]
        print("success par", succ_p.astype(int).tolist()); print("success ser", succ_s.astype(int).tolist())
        break
else:
    print(f"no difference in {steps} steps, {E} envs")
