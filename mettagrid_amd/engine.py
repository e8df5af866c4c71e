"""Python host side of the MI355X MettaGrid step engine: ctypes binding of libmgx (include/mgx.h) plus mirrors of
the reference's Python-visible interface.

* ``BatchedMettaGrid`` — E envs per engine, buffers ``[E*A, T, 3]`` etc. resident in HBM (torch tensors are used
  only as device-memory handles).
* ``MettaGrid`` — same constructor, methods and error behaviour as the reference's pybind class
  ``mettagrid.mettagrid_c.MettaGrid`` (/root/reference/cpp/bindings/mettagrid_py.cpp:242-397, type stub
  python/src/mettagrid/mettagrid_c.pyi:752-806) for ONE env, so callers written against the reference
  (``c.actions()[:] = a; c.step(); c.observations()``) run unchanged.

There is no CPU fallback: if libmgx.so (the HIP build) is missing this module raises at first use.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .compiler import Program, compile_spec
from .fmt import K
from .signature import objects_from_raw, stats_dicts

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MGX_LIB") or os.path.join(_HERE, "libmgx.so")  # MGX_LIB: instrumented dev builds
_lib = None
OBJ_RECORD_WORDS = 42

ENV_TOKEN_OVERFLOW, ENV_INVALID_KEY_RANGE, ENV_DEPTH, ENV_TOO_MANY_OBJECTS, ENV_TOKEN_POOL = 1, 2, 4, 8, 16
ENV_PROXY_INVENTORY, ENV_AGENT_LIFECYCLE = 32, 64   # territory proxy given an inventory / an agent spawned or removed


class MgxError(RuntimeError):
    pass


def load_lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MgxError(
            f"{LIB_PATH} not found: the HIP engine is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    try:  # one HIP runtime per process: torch bundles libamdhip64.so.7; load it first so libmgx binds to the same one
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.mgx_last_error.restype = C.c_char_p
    L.mgx_create.argtypes = [vp, C.c_size_t, vp, vp, i32, i32, C.POINTER(vp)]
    L.mgx_destroy.argtypes = [vp]
    L.mgx_destroy.restype = None
    L.mgx_set_buffers.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i64, i32]
    L.mgx_step.argtypes = [vp]
    L.mgx_reset_envs.argtypes = [vp, vp, vp, vp]
    L.mgx_sync.argtypes = [vp]
    L.mgx_stream.argtypes = [vp]
    L.mgx_stream.restype = vp
    L.mgx_chain_world.argtypes = [vp, vp]
    L.mgx_wait_before_outputs.argtypes = [vp, vp]
    L.mgx_set_joint_actions.argtypes = [vp, vp, i32, vp, i32]
    L.mgx_poll_action_errors.argtypes = [vp, i32, C.POINTER(C.c_uint32), C.POINTER(i64), C.POINTER(i32)]
    L.mgx_get_buffers.argtypes = [vp] + [C.POINTER(vp)] * 6 + [C.POINTER(i32)]
    L.mgx_get_episode_rewards.argtypes = [vp, vp]
    L.mgx_get_action_success.argtypes = [vp, vp]
    L.mgx_get_current_steps.argtypes = [vp, vp]
    L.mgx_get_stats.argtypes = [vp, i32, vp, vp, vp, vp]
    L.mgx_get_objects.argtypes = [vp, i32, vp, C.POINTER(i32)]
    L.mgx_get_objects_batch.argtypes = [vp, vp, i32, vp, vp]
    L.mgx_get_invalid_index_extra.argtypes = [vp, i32, vp, vp]
    L.mgx_get_reward_state.argtypes = [vp, i32, vp]
    L.mgx_poll_errors.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(i32)]
    L.mgx_set_inventory.argtypes = [vp, i32, i32, vp, vp, i32]
    L.mgx_decode_obs.argtypes = [vp, vp, i64, vp, i32, vp]
    L.mgx_set_box_output.argtypes = [vp, vp, i32, i32, vp]
    L.mgx_state_digests.argtypes = [vp, vp]
    L.mgx_set_map_pool.argtypes = [vp, vp, i32]
    L.mgx_reset_envs_from_pool.argtypes = [vp, vp, vp, vp]
    L.mgx_set_auto_reset.argtypes = [vp, i32, i32, vp]
    L.mgx_get_episodes.argtypes = [vp, vp, vp]
    L.mgx_set_episode_stats.argtypes = [vp, i32, i32, i32]
    L.mgx_episode_stats_layout.argtypes = [vp, vp]
    L.mgx_record_episodes.argtypes = [vp, vp]
    L.mgx_request_episode_stats.argtypes = [vp]
    L.mgx_fetch_episode_stats.argtypes = [vp, i32, vp, C.POINTER(i32)]
    L.mgx_drain_episode_log.argtypes = [vp, vp, i32, C.POINTER(i32), C.POINTER(i32)]
    L.mgx_count_objects_with_tag.argtypes = [vp, i32, i32, C.POINTER(i32)]
    L.mgx_set_profiling.argtypes = [vp, i32]
    L.mgx_get_step_timing.argtypes = [vp, vp]
    L.mgx_attach_code.argtypes = [vp, i32, C.c_char_p]
    if hasattr(L, "mgx_pack_rows"):   # (absent from the CPU sanitizer build, tests/cpu_emu: wavefront-cooperative kernels)
        L.mgx_pack_scratch_bytes.argtypes = [i64]
        L.mgx_pack_scratch_bytes.restype = i64
        L.mgx_pack_rows.argtypes = [vp, i64, i32, vp, vp, i64, vp, vp]
        L.mgx_pack_result.argtypes = [vp, i64, C.POINTER(i64), C.POINTER(i32), vp]
        L.mgx_unpack_rows.argtypes = [vp, vp, i64, i32, vp, vp, vp]
    for name in ("mgx_num_envs", "mgx_num_agents", "mgx_num_tokens", "mgx_obs_variant", "mgx_act_variant", "mgx_handler_variant",
                 "mgx_world_prog_in_lds", "mgx_is_extended", "mgx_dispatch_pairs", "mgx_integer_bookkeeping"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = i32
    L.mgx_state_bytes.argtypes = [vp]
    L.mgx_state_bytes.restype = i64
    _lib = L
    return L


def exported_symbols() -> list:
    """Entry points declared in include/mgx.h (used by the CPU-side ABI test)."""
    import re
    hdr = open(os.path.join(os.path.dirname(_HERE), "include", "mgx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mgx_[a-z_]+)\s*\(", hdr)))


def _check(rc: int) -> None:
    if rc != 0:
        msg = load_lib().mgx_last_error().decode()
        if rc == -1:
            raise ValueError(msg) if "shape" not in msg and "token" not in msg else RuntimeError(msg)
        raise MgxError(f"libmgx error {rc}: {msg}")


class BatchedMettaGrid:
    """E independent envs stepped together on one MI355X.

    ``buffers="device"``: observation/reward/terminal/truncation/action buffers are torch CUDA tensors owned by this
    object (or passed in through ``set_buffers``) and the kernels write into them directly.
    ``buffers="host"``: numpy arrays; actions are uploaded and results downloaded on every step (PCIe-inclusive).
    """

    def __init__(self, prog: Program, class_maps: np.ndarray, seeds, device: int = 0, buffers: str = "device",
                 specialize="auto") -> None:
        """``specialize``: compile the world / observation kernels for THIS program in the background and switch to them when
        they are ready (mettagrid_amd/jit.py; results are identical).  ``"auto"``: for batches of at least 1 024 envs whose
        program is not one the build already specialised; ``"async"`` / ``"sync"`` (wait in the constructor): always;
        ``False``: never."""
        self.L = load_lib()
        self.prog = prog
        cm = np.ascontiguousarray(class_maps, dtype=np.uint16)
        H, W = int(prog.words[K.H_HEIGHT]), int(prog.words[K.H_WIDTH])
        if cm.ndim == 2:
            cm = cm[None]
        if cm.shape[1:] != (H, W):
            raise ValueError(f"class_maps has shape {cm.shape} but the program was compiled for {H}x{W} maps")
        self.E = cm.shape[0]
        self.A, self.T = prog.num_agents, prog.num_tokens
        seeds = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.uint32), (self.E,)))
        words = np.ascontiguousarray(prog.words, dtype=np.int32)
        self.device = device
        self.h = C.c_void_p()
        _check(self.L.mgx_create(words.ctypes.data, words.size, cm.ctypes.data, seeds.ctypes.data, self.E, device,
                                 C.byref(self.h)))
        self.kind = buffers
        rows = self.E * self.A
        if isinstance(buffers, dict):   # the caller's own device tensors (e.g. slices of one batch: EnvGroups)
            self.kind = "device"
            want = {"obs": ((rows, self.T, 3), "uint8"), "terminals": ((rows,), "bool"), "truncations": ((rows,), "bool"),
                    "rewards": ((rows,), "float32"), "actions": ((rows,), "int32"), "vibe_actions": ((rows,), "int32")}
            for name, (shape, dt) in want.items():
                t = buffers[name]
                if tuple(t.shape) != shape or str(t.dtype) != "torch." + dt or not t.is_contiguous() or t.device.index != device:
                    raise ValueError(f"buffer {name!r} must be a contiguous {dt}{list(shape)} tensor on cuda:{device}")
                setattr(self, name, t)
            self._bind(*(t.data_ptr() for t in (self.obs, self.terminals, self.truncations, self.rewards,
                                                self.actions, self.vibe_actions)), mem_kind=1)
        elif buffers == "device":
            import torch
            dev = torch.device("cuda", device)
            self.obs = torch.empty((rows, self.T, 3), dtype=torch.uint8, device=dev)
            self.terminals = torch.empty(rows, dtype=torch.bool, device=dev)
            self.truncations = torch.empty(rows, dtype=torch.bool, device=dev)
            self.rewards = torch.empty(rows, dtype=torch.float32, device=dev)
            self.actions = torch.zeros(rows, dtype=torch.int32, device=dev)
            self.vibe_actions = torch.zeros(rows, dtype=torch.int32, device=dev)
            torch.cuda.synchronize(dev)
            self._bind(*(t.data_ptr() for t in (self.obs, self.terminals, self.truncations, self.rewards,
                                                self.actions, self.vibe_actions)), mem_kind=1)
        elif buffers == "host":
            self.obs = np.empty((rows, self.T, 3), np.uint8)
            self.terminals = np.zeros(rows, np.bool_)
            self.truncations = np.zeros(rows, np.bool_)
            self.rewards = np.zeros(rows, np.float32)
            self.actions = np.zeros(rows, np.int32)
            self.vibe_actions = np.zeros(rows, np.int32)
            self._bind(*(a.ctypes.data for a in (self.obs, self.terminals, self.truncations, self.rewards,
                                                 self.actions, self.vibe_actions)), mem_kind=0)
        else:
            raise ValueError("buffers must be 'device' or 'host'")
        self._jit_jobs, self.jit_errors = [], []
        if specialize and (specialize != "auto" or self.E >= 1024):
            self._start_jit(wait=specialize == "sync")

    # ---- run-time specialisation (include/mgx.h mgx_attach_code) ----
    def _start_jit(self, wait: bool = False) -> None:
        from . import jit
        if not jit.enabled():
            return
        if self.L.mgx_is_extended(self.h):   # extended programs: the lane-per-agent dispatch kernel, when the engine runs it
            kinds = ["actx"] if self.act_variant == 1 and self.handler_variant == 0 and self.L.mgx_world_prog_in_lds(self.h) else []
        else:
            kinds = [k for k, have in (("world", self.handler_variant != 0 or self.act_variant != 0), ("obs", self.obs_variant != 0)) if not have]
        if kinds:
            self._jit_jobs = jit.start(self.prog, bool(self.L.mgx_world_prog_in_lds(self.h)), kinds)
        if wait:
            for j in self._jit_jobs:
                j.wait()
        self._poll_jit()

    def _poll_jit(self) -> None:
        from . import jit
        for j in list(self._jit_jobs):
            if not j.ready():
                continue
            self._jit_jobs.remove(j)
            if j.error is None:
                rc = self.L.mgx_attach_code(self.h, jit.KINDS[j.kind], j.path.encode())
                if rc != 0:
                    j.error = self.L.mgx_last_error().decode()
            if j.error is not None:   # the generic kernels stay: slower, same results
                self.jit_errors.append((j.kind, j.error))
                if os.environ.get("MGX_VERBOSE"):
                    print(f"[mgx] run-time specialisation of the {j.kind} kernel failed: {j.error}", flush=True)

    def jit_pending(self) -> bool:
        """True while a code object for this engine is still compiling (``step`` attaches it when it is done)."""
        return bool(self._jit_jobs)

    def _bind(self, obs, term, trunc, rew, act, vact, mem_kind: int, rows=None, tokens=None) -> None:
        _check(self.L.mgx_set_buffers(self.h, obs, term, trunc, rew, act, vact,
                                      self.E * self.A if rows is None else rows,
                                      self.T if tokens is None else tokens, mem_kind))

    def close(self) -> None:
        if getattr(self, "h", None) and self.h:
            self.L.mgx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- stepping ----
    def step(self, check_errors: bool = False) -> None:
        if self._jit_jobs:
            self._poll_jit()
        _check(self.L.mgx_step(self.h))
        if check_errors:
            self.raise_env_errors()

    def sync(self) -> None:
        _check(self.L.mgx_sync(self.h))

    def set_joint_actions(self, joint, num_primary: int, vibe_ids) -> None:
        """One joint discrete id per agent -> the engine's primary / vibe action buffers, decoded on the device on the
        engine's stream (mgx.h mgx_set_joint_actions; MettaGridPufferEnv.step's decoding without its range checks).
        ``joint``: contiguous int32 CUDA tensor [E*A]; call ``wait_for_caller()`` first if it was written on another stream."""
        ids = np.ascontiguousarray(np.asarray(vibe_ids, dtype=np.int32))
        _check(self.L.mgx_set_joint_actions(self.h, joint.data_ptr(), int(num_primary), ids.ctypes.data if ids.size else None, int(ids.size)))

    def poll_action_errors(self, wait: bool = False):
        """(flags, row, value) of the range check ``set_joint_actions`` makes on the device (include/mgx.h
        mgx_poll_action_errors): flags 0 = all ids seen so far were in range.  ``wait=False`` does not synchronise."""
        flags, row, value = C.c_uint32(0), C.c_int64(-1), C.c_int32(0)
        _check(self.L.mgx_poll_action_errors(self.h, 1 if wait else 0, C.byref(flags), C.byref(row), C.byref(value)))
        return flags.value, row.value, value.value

    def chain_world_after(self, other: "BatchedMettaGrid | None") -> None:
        """From now on this engine's world-update kernels start only when ``other``'s most recent ones have finished
        (include/mgx.h mgx_chain_world): the building block of ``EnvGroups``."""
        _check(self.L.mgx_chain_world(self.h, other.h if other is not None else None))

    def wait_before_outputs(self, event) -> None:
        """The next engine work that writes observations / rewards / terminals / truncations waits for ``event`` (a recorded
        ``torch.cuda.Event``, kept alive here until then); the next step's world update does not (include/mgx.h
        mgx_wait_before_outputs).  Used by ``dist.GatherToRoot`` to order a step behind the staging copy of the previous one."""
        self._out_fences = (getattr(self, "_out_fences", []) + [event])[-4:]   # (several consumers: none is freed while pending)
        _check(self.L.mgx_wait_before_outputs(self.h, C.c_void_p(event.cuda_event)))

    def reset_envs(self, env_mask, class_maps=None, seeds=None) -> None:
        """Restart the selected envs in place on the device (new episode): optional new maps / seeds for them.
        Their buffer rows are cleared and receive the initial observations, exactly like a fresh reference
        ``MettaGrid`` + ``set_buffers``."""
        mask = np.ascontiguousarray(np.asarray(env_mask, dtype=np.uint8).reshape(self.E))
        cm_ptr = sd_ptr = None
        if class_maps is not None:
            cm = np.ascontiguousarray(class_maps, dtype=np.uint16)
            H, W = int(self.prog.words[K.H_HEIGHT]), int(self.prog.words[K.H_WIDTH])
            if cm.shape != (self.E, H, W):
                raise ValueError(f"class_maps must have shape {(self.E, H, W)} (only masked envs are read)")
            cm_ptr = cm.ctypes.data
        if seeds is not None:
            sd = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.uint32), (self.E,)))
            sd_ptr = sd.ctypes.data
        _check(self.L.mgx_reset_envs(self.h, mask.ctypes.data, cm_ptr, sd_ptr))

    # ---- device-resident map pool + auto-reset ----
    def set_map_pool(self, class_maps) -> None:
        """Upload finished maps once (uint16 [M, H, W], class index + 1); episodes then restart from them on the device."""
        cm = np.ascontiguousarray(class_maps, dtype=np.uint16)
        H, W = int(self.prog.words[K.H_HEIGHT]), int(self.prog.words[K.H_WIDTH])
        if cm.ndim != 3 or cm.shape[1:] != (H, W):
            raise ValueError(f"map pool must have shape [M, {H}, {W}]")
        _check(self.L.mgx_set_map_pool(self.h, cm.ctypes.data, cm.shape[0]))
        self.pool_size = cm.shape[0]

    def reset_envs_from_pool(self, env_mask, pool_index, seeds=None) -> None:
        mask = np.ascontiguousarray(np.asarray(env_mask, dtype=np.uint8).reshape(self.E))
        idx = np.ascontiguousarray(np.broadcast_to(np.asarray(pool_index, dtype=np.int32), (self.E,)))
        sd_ptr = None
        if seeds is not None:
            sd = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.uint32), (self.E,)))
            sd_ptr = sd.ctypes.data
        _check(self.L.mgx_reset_envs_from_pool(self.h, mask.ctypes.data, idx.ctypes.data, sd_ptr))

    def set_auto_reset(self, enabled: bool = True, pool_stride: int = 1, early_end_steps=None) -> None:
        """Lazy auto-reset on the device (MettaGridPufferEnv.step, mettagrid_puffer_env.py:299-302); ``early_end_steps``
        [E]: step at which each env's first episode is truncated early (EarlyResetHandler), 0 = never."""
        ptr = None
        if early_end_steps is not None:
            ee = np.ascontiguousarray(np.asarray(early_end_steps, dtype=np.uint32).reshape(self.E))
            ptr = ee.ctypes.data
        _check(self.L.mgx_set_auto_reset(self.h, 1 if enabled else 0, int(pool_stride), ptr))

    def episodes(self):
        ep, mi = np.empty(self.E, np.uint32), np.empty(self.E, np.int32)
        _check(self.L.mgx_get_episodes(self.h, ep.ctypes.data, mi.ctypes.data))
        return ep, mi

    # ---- episode-end statistics (include/mgx.h "Episode-end statistics"; the reference's StatsTracker.on_episode_end) ----
    EPL = ("NG", "NS", "A", "TOTALS_WORDS", "REC_WORDS", "LOG_WORDS", "OFF_GAME", "OFF_GAME_BITS", "OFF_AGENT", "OFF_AGENT_BITS",
           "OFF_REWARDS", "OFF_PA", "OFF_PA_BITS", "OFF_INVK", "OFF_INVN", "PER_AGENT", "LOG_CAPACITY")
    EPT = ("episodes", "return_sum", "return_min", "return_max", "length_sum", "length_min", "length_max", "terminated")

    def set_episode_stats(self, enabled: bool = True, log_capacity: int = 0, log_per_agent: bool = False) -> None:
        """Reduce the stats of every finished env into batch totals on the device (and keep up to ``log_capacity`` per-episode
        records, with the per-agent rows if asked) before the auto-reset wipes them."""
        _check(self.L.mgx_set_episode_stats(self.h, 1 if enabled else 0, int(log_capacity), 1 if log_per_agent else 0))
        self._epl = None
        if enabled:
            out = np.zeros(len(self.EPL), np.int32)
            _check(self.L.mgx_episode_stats_layout(self.h, out.ctypes.data))
            self._epl = {k: int(v) for k, v in zip(self.EPL, out)}

    def record_episodes(self, env_mask) -> None:
        """Host-driven restarts: record the masked envs' episodes as finished now (call before ``reset_envs``)."""
        mask = np.ascontiguousarray(np.asarray(env_mask, dtype=np.uint8).reshape(self.E))
        _check(self.L.mgx_record_episodes(self.h, mask.ctypes.data))

    def request_episode_stats(self) -> None:
        _check(self.L.mgx_request_episode_stats(self.h))

    def _fetch_raw(self, wait: bool):
        out = np.zeros(self._epl["TOTALS_WORDS"], np.float64)
        ready = C.c_int32(0)
        _check(self.L.mgx_fetch_episode_stats(self.h, 1 if wait else 0, out.ctypes.data, C.byref(ready)))
        return out if ready.value else None

    def fetch_episode_stats(self, wait: bool = False):
        """The snapshot requested last as a dict (see ``episode_totals_dict``), or None while its copy is still in flight."""
        raw = self._fetch_raw(wait)
        return None if raw is None else self.episode_totals_dict(raw)

    def drain_episode_stats(self):
        """Everything recorded up to now, synchronously: a snapshot still in flight (requested earlier, not fetched) is merged
        with a fresh one (sums and counts add, minima / maxima combine)."""
        first = self._fetch_raw(True)          # None unless a request was pending
        self.request_episode_stats()
        tot = self._fetch_raw(True)
        if first is not None:
            lo, hi = [2, 5], [3, 6]            # MGX_EPT_RETURN_MIN / LENGTH_MIN, ..._MAX
            merged = first + tot
            merged[lo] = np.minimum(first[lo], tot[lo])
            merged[hi] = np.maximum(first[hi], tot[hi])
            tot = merged
        return self.episode_totals_dict(tot)

    def episode_totals_dict(self, totals: np.ndarray) -> dict:
        """Raw totals -> {"episodes", "return_sum", ..., "game_sum": {name: f64}, "game_count": {name: int}, "agent_sum", "agent_count"}
        (only keys some finished episode held)."""
        NG, NS, H = self._epl["NG"], self._epl["NS"], len(self.EPT)
        out = {k: float(totals[i]) for i, k in enumerate(self.EPT)}
        out["episodes"] = int(out["episodes"])
        gs, gc = totals[H:H + NG], totals[H + NG:H + 2 * NG]
        as_, ac = totals[H + 2 * NG:H + 2 * NG + NS], totals[H + 2 * NG + NS:H + 2 * NG + 2 * NS]
        out["game_sum"] = {self.prog.game_stat_names[i]: float(gs[i]) for i in range(NG) if gc[i]}
        out["game_count"] = {self.prog.game_stat_names[i]: int(gc[i]) for i in range(NG) if gc[i]}
        out["agent_sum"] = {self.prog.agent_stat_names[i]: float(as_[i]) for i in range(NS) if ac[i]}
        out["agent_count"] = {self.prog.agent_stat_names[i]: int(ac[i]) for i in range(NS) if ac[i]}
        return out

    def drain_episode_log(self):
        """(list of per-episode dicts, dropped count).  Each dict: env, episode, map_index, seed, steps, flags, episode_rewards
        (f32 [A]), game {name: value}, agent {name: f64 mean over agents} — the reference's infos["game"] / infos["agent"] —
        and, with log_per_agent, per_agent [A] dicts (incl. the "action.invalid_index.<k>" keys without a stat column)."""
        Lw = self._epl
        cap, W = Lw["LOG_CAPACITY"], Lw["LOG_WORDS"]
        if cap <= 0:
            raise ValueError("no episode log: set_episode_stats(log_capacity=...)")
        raw = np.zeros((cap, W), np.uint32)
        n, dropped = C.c_int32(0), C.c_int32(0)
        _check(self.L.mgx_drain_episode_log(self.h, raw.ctypes.data, cap, C.byref(n), C.byref(dropped)))
        return [self.episode_record_dict(raw[i]) for i in range(n.value)], dropped.value

    def episode_record_dict(self, rec: np.ndarray) -> dict:
        Lw = self._epl
        NG, NS, A = Lw["NG"], Lw["NS"], Lw["A"]
        gnames, anames = self.prog.game_stat_names, self.prog.agent_stat_names

        def bits(off, n):
            w = rec[off:off + (n + 31) // 32]
            return [(int(w[i >> 5]) >> (i & 31)) & 1 for i in range(n)]
        gv = rec[Lw["OFF_GAME"]:Lw["OFF_GAME"] + NG].view(np.float32)
        gb = bits(Lw["OFF_GAME_BITS"], NG)
        av = np.ascontiguousarray(rec[Lw["OFF_AGENT"]:Lw["OFF_AGENT"] + 2 * NS]).view(np.float64)
        ab = bits(Lw["OFF_AGENT_BITS"], NS)
        out = {"env": int(rec[0]), "episode": int(rec[1]), "map_index": int(np.int32(rec[2])), "seed": int(rec[3]), "steps": int(rec[4]),
               "flags": int(rec[5]), "return_sum": float(np.ascontiguousarray(rec[6:8]).view(np.float64)[0]),
               "episode_rewards": rec[Lw["OFF_REWARDS"]:Lw["OFF_REWARDS"] + A].view(np.float32).copy(),
               "game": {gnames[i]: float(gv[i]) for i in range(NG) if gb[i]},
               "agent": {anames[i]: float(av[i]) for i in range(NS) if ab[i]}}
        if Lw["PER_AGENT"]:
            pa = rec[Lw["OFF_PA"]:Lw["OFF_PA"] + A * NS].view(np.float32).reshape(A, NS)
            nsw = (NS + 31) // 32
            ik = rec[Lw["OFF_INVK"]:Lw["OFF_INVK"] + A * K.INVALID_EXTRA].view(np.int32).reshape(A, K.INVALID_EXTRA)
            inn = rec[Lw["OFF_INVN"]:Lw["OFF_INVN"] + A * K.INVALID_EXTRA].view(np.float32).reshape(A, K.INVALID_EXTRA)
            per = []
            for a in range(A):
                pb = bits(Lw["OFF_PA_BITS"] + a * nsw, NS)
                dct = {anames[i]: float(pa[a, i]) for i in range(NS) if pb[i]}
                for q in range(K.INVALID_EXTRA):
                    if inn[a, q] != 0:
                        dct[f"action.invalid_index.{int(ik[a, q])}"] = float(inn[a, q])
                per.append(dct)
            out["per_agent"] = per
            extra = {}   # keys without a stat column take part in infos["agent"] like any other (stats_tracker.py:41-46)
            for dct in per:
                for k, v in dct.items():
                    if k.startswith("action.invalid_index.") and k not in anames:
                        extra[k] = extra.get(k, 0) + v
            for k, v in extra.items():
                out["agent"][k] = v / A
        return out

    @property
    def stream(self) -> int:
        return int(self.L.mgx_stream(self.h) or 0)

    # ---- ordering against the caller's torch stream (device buffers) ----
    def _ext_stream(self):
        import torch
        if getattr(self, "_ext", None) is None:
            self._ext = torch.cuda.ExternalStream(self.stream, device=torch.device("cuda", self.device))
        return self._ext

    def wait_for_caller(self) -> None:
        """Order the engine's stream behind everything the caller has enqueued on its current torch stream (the action
        writes): the engine's kernels then read the actions the caller meant."""
        import torch
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._ext_stream().wait_event(ev)

    def caller_waits(self) -> None:
        """Order the caller's current torch stream behind the engine's stream: tensors returned after ``step`` hold the
        step's results for any kernel the caller enqueues next (no host synchronisation)."""
        import torch
        ev = torch.cuda.Event()
        ev.record(self._ext_stream())
        torch.cuda.current_stream(self.device).wait_event(ev)

    def poll_errors(self):
        bits, first = C.c_uint32(0), C.c_int32(-1)
        _check(self.L.mgx_poll_errors(self.h, C.byref(bits), C.byref(first)))
        return bits.value, first.value

    def raise_env_errors(self) -> None:
        bits, first = self.poll_errors()
        if bits & ENV_TOKEN_OVERFLOW:  # reference: std::runtime_error -> RuntimeError (mettagrid_c.cpp:364-375)
            raise RuntimeError(f"Observation token budget exceeded (first env {first}, budget={self.T})")
        if bits & ENV_TOO_MANY_OBJECTS:
            raise MgxError(f"env {first}: map holds more objects than max_objects={self.prog.max_objects}")
        if bits & ENV_DEPTH:
            raise MgxError(f"env {first}: handler / inventory-limit recursion exceeds the engine's depth")
        if bits & ENV_INVALID_KEY_RANGE:
            raise MgxError(f"env {first}: an agent sent more than {K.INVALID_EXTRA} distinct invalid action indices far outside "
                           "the action table in one episode (action.invalid_index.<k> keys exhausted)")
        if bits & ENV_TOKEN_POOL:
            raise MgxError(f"env {first}: the observation kernel's per-env token cache is exhausted "
                           "(objects' token lists would be dropped from observations)")
        if bits & (ENV_PROXY_INVENTORY | ENV_AGENT_LIFECYCLE):
            raise MgxError(f"env {first}: a handler touched an agent's existence or a territory proxy's inventory (bits {bits})")

    # ---- readbacks ----
    def episode_rewards(self) -> np.ndarray:
        out = np.empty(self.E * self.A, np.float32)
        _check(self.L.mgx_get_episode_rewards(self.h, out.ctypes.data))
        return out

    def action_success(self) -> np.ndarray:
        out = np.empty(self.E * self.A, np.uint8)
        _check(self.L.mgx_get_action_success(self.h, out.ctypes.data))
        return out.astype(bool)

    def current_steps(self) -> np.ndarray:
        out = np.empty(self.E, np.uint32)
        _check(self.L.mgx_get_current_steps(self.h, out.ctypes.data))
        return out

    def raw_stats(self, env: int = 0):
        ng, na = len(self.prog.game_stat_names), len(self.prog.agent_stat_names)
        gv, gt = np.zeros(ng, np.float32), np.zeros(ng, np.uint8)
        av, at = np.zeros((self.A, na), np.float32), np.zeros((self.A, na), np.uint8)
        _check(self.L.mgx_get_stats(self.h, env, gv.ctypes.data, gt.ctypes.data, av.ctypes.data, at.ctypes.data))
        return gv, gt, av, at

    def invalid_index_extra(self, env: int = 0) -> list:
        """Per agent {k: count} of "action.invalid_index.<k>" for the k that have no fixed stat column (mgx.h)."""
        k = np.zeros((self.A, K.INVALID_EXTRA), np.int32)
        n = np.zeros((self.A, K.INVALID_EXTRA), np.float32)
        _check(self.L.mgx_get_invalid_index_extra(self.h, env, k.ctypes.data, n.ctypes.data))
        return [{int(k[a, q]): float(n[a, q]) for q in range(K.INVALID_EXTRA) if n[a, q] != 0} for a in range(self.A)]

    def get_episode_stats(self, env: int = 0) -> dict:
        return stats_dicts(self.prog, *self.raw_stats(env), extra=self.invalid_index_extra(env))

    def raw_objects(self, env: int = 0) -> np.ndarray:
        out = np.zeros((self.prog.max_objects, OBJ_RECORD_WORDS), np.int32)
        n = C.c_int32(0)
        _check(self.L.mgx_get_objects(self.h, env, out.ctypes.data, C.byref(n)))
        return out[: n.value]

    def raw_objects_batch(self, envs) -> list:
        """``raw_objects`` of many envs with one kernel and one device->host copy (include/mgx.h mgx_get_objects_batch)."""
        idx = np.ascontiguousarray(np.asarray(envs, dtype=np.int32).reshape(-1))
        if idx.size == 0:
            return []
        out = np.zeros((idx.size, self.prog.max_objects, OBJ_RECORD_WORDS), np.int32)
        n = np.zeros(idx.size, np.int32)
        _check(self.L.mgx_get_objects_batch(self.h, idx.ctypes.data, int(idx.size), out.ctypes.data, n.ctypes.data))
        return [out[i, : n[i]] for i in range(idx.size)]

    def grid_objects_batch(self, envs) -> list:
        """``grid_objects()`` dicts (cpp/bindings/mettagrid_py.cpp:28-139) of many envs; the per-agent reward state is still
        one small copy per env."""
        return [objects_from_raw(self.prog, raw, self.current_stat_reward(int(e)))
                for e, raw in zip(np.asarray(envs).reshape(-1), self.raw_objects_batch(envs))]

    def current_stat_reward(self, env: int = 0) -> np.ndarray:
        out = np.zeros(self.A, np.float32)
        _check(self.L.mgx_get_reward_state(self.h, env, out.ctypes.data))
        return out

    def grid_objects(self, env: int = 0) -> dict:
        return objects_from_raw(self.prog, self.raw_objects(env), self.current_stat_reward(env))

    def snapshot(self, env: int | None = None) -> dict:
        """Host copies of the caller-visible buffers (all envs, or the rows of one env)."""
        self.sync()
        if self.kind == "device":
            import torch
            torch.cuda.synchronize(self.device)
            obs, rew = self.obs.cpu().numpy(), self.rewards.cpu().numpy()
            term, trunc = self.terminals.cpu().numpy(), self.truncations.cpu().numpy()
        else:
            obs, rew, term, trunc = self.obs.copy(), self.rewards.copy(), self.terminals.copy(), self.truncations.copy()
        out = dict(obs=obs, rewards=rew, terminals=term.astype(bool), truncations=trunc.astype(bool),
                   action_success=self.action_success(), episode_rewards=self.episode_rewards())
        if env is not None:
            sl = slice(env * self.A, (env + 1) * self.A)
            out = {k: v[sl] for k, v in out.items()}
        return out

    def set_inventory(self, env: int, agent_id: int, inventory: dict) -> None:
        """``MettaGrid.set_inventory`` (cpp/bindings/mettagrid_py.cpp:203-207) for agent ``agent_id`` of env ``env``;
        ``inventory`` maps resource ids to amounts.  The reference walks the std::unordered_map pybind11 builds from
        the dict, so the items are applied in that map's iteration order (mettagrid_amd/umap.py)."""
        from .umap import from_pydict
        ids = [int(k) for k in inventory]
        order = from_pydict(ids).keys()
        items = np.asarray(order, dtype=np.int32)
        amounts = np.asarray([int(inventory[k]) for k in order], dtype=np.int32)
        _check(self.L.mgx_set_inventory(self.h, int(env), int(agent_id), items.ctypes.data, amounts.ctypes.data, len(order)))

    def count_objects_with_tag(self, env: int, tag_id: int) -> int:
        out = C.c_int32(0)
        _check(self.L.mgx_count_objects_with_tag(self.h, int(env), int(tag_id), C.byref(out)))
        return out.value

    def state_digests(self) -> np.ndarray:
        """uint64 [E]: digest of every env's signature state (objects, stats, rewards, success, step) in one kernel."""
        out = np.empty(self.E, np.uint64)
        _check(self.L.mgx_state_digests(self.h, out.ctypes.data))
        bits, first = self.poll_errors()
        if bits & 0x8000:  # MGX_ENV_INTERNAL: the digest kernel found a per-agent mirror out of step with its object row
            raise MgxError(f"env {first}: internal consistency check failed (per-agent mirror differs from the object row)")
        return out

    # ---- policy-side token decode ----
    def feature_scale(self) -> np.ndarray:
        """scale[f] = max(normalization of feature f, 1) (grid_obs_wrapper.py:39-44), from the compiled program."""
        scale = np.ones(256, np.float32)
        for i, z in enumerate(self.prog.feature_norms):
            scale[i] = max(float(z), 1.0)
        return scale

    def set_box_output(self, out=None):
        """Fused form of ``decode_obs`` (include/mgx.h mgx_set_box_output): from the next observation pass on the observation
        kernel writes the dense box of every agent into ``out`` — a contiguous torch CUDA tensor [E*A, C, H, W], float32
        (bit-identical to ``GridObsWrapper._convert``) or bfloat16 (that box rounded to nearest even) — instead of the token
        rows.  ``None`` switches back to token rows.  The tensor is kept alive here."""
        import torch
        if out is None:
            self._box = None
            _check(self.L.mgx_set_box_output(self.h, None, 0, 0, None))
            return
        Cn = len(self.prog.feature_norms)
        Hh, Ww = int(self.prog.words[K.H_OBS_HEIGHT]), int(self.prog.words[K.H_OBS_WIDTH])
        if tuple(out.shape) != (self.E * self.A, Cn, Hh, Ww) or out.dtype not in (torch.float32, torch.bfloat16) or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous float32 / bfloat16 tensor of shape {(self.E * self.A, Cn, Hh, Ww)}")
        scale = self.feature_scale()
        self._box = out
        _check(self.L.mgx_set_box_output(self.h, out.data_ptr(), 1 if out.dtype == torch.float32 else 2, Cn, scale.ctypes.data))

    def decode_obs(self, out=None, tokens=None):
        """Dense float32 box [rows, C, H, W] of the current observations (``GridObsWrapper._convert``), computed on the
        GPU and ordered on the engine's stream.  ``out``: torch CUDA tensor to fill (allocated if None); ``tokens``: another
        token tensor u8 [rows, T, 3] on the device instead of the engine's own observation buffer."""
        import torch
        Cn = len(self.prog.feature_norms)
        Hh, Ww = int(self.prog.words[K.H_OBS_HEIGHT]), int(self.prog.words[K.H_OBS_WIDTH])
        dev = torch.device("cuda", self.device)
        rows = self.E * self.A if tokens is None else int(tokens.shape[0])
        if out is None:
            out = torch.empty((rows, Cn, Hh, Ww), dtype=torch.float32, device=dev)
        if tuple(out.shape) != (rows, Cn, Hh, Ww) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous float32 tensor of shape {(rows, Cn, Hh, Ww)}")
        scale = self.feature_scale()
        tok_ptr = None
        if tokens is not None:
            if tokens.dtype != torch.uint8 or not tokens.is_contiguous() or tokens.shape[1:] != (self.T, 3):
                raise ValueError("tokens must be a contiguous uint8 tensor [rows, T, 3]")
            tok_ptr = tokens.data_ptr()
        _check(self.L.mgx_decode_obs(self.h, tok_ptr, rows, out.data_ptr(), Cn, scale.ctypes.data))
        return out

    def set_profiling(self, on: bool) -> None:
        _check(self.L.mgx_set_profiling(self.h, 1 if on else 0))

    TIMING_SEGMENTS = ("actions", "aoe", "tail", "obs", "rewards")   # include/mgx.h MGX_T_*

    def step_timing_segments_ms(self) -> dict:
        out = np.zeros(len(self.TIMING_SEGMENTS), np.float32)
        _check(self.L.mgx_get_step_timing(self.h, out.ctypes.data))
        return {k: float(v) for k, v in zip(self.TIMING_SEGMENTS, out)}

    def step_timing_ms(self):
        """(world update, observation + rewards) of the most recent step in milliseconds."""
        t = self.step_timing_segments_ms()
        return t["actions"] + t["aoe"] + t["tail"], t["obs"] + t["rewards"]

    @property
    def obs_variant(self) -> int:
        """0: generic observation kernel; 3: the instance compiled for the shape of BASELINE.json configs[2] (gen_presets.py);
        5: that shape with an episode length read at run time; 9: an instance compiled for this program at run time (jit.py)."""
        return int(self.L.mgx_obs_variant(self.h))

    @property
    def handler_variant(self) -> int:
        """0: handler interpreter; 3 / 4: code generated at build() for the rung-3 / rung-4 preset (gen_handlers.py); 9: code
        generated for this program at run time (jit.py)."""
        return int(self.L.mgx_handler_variant(self.h))

    @property
    def act_variant(self) -> int:
        """0: action dispatch with one lane per env; 1: one lane per agent in conflict-ordered rounds (csrc/mgx_act.h)."""
        return int(self.L.mgx_act_variant(self.h))

    @property
    def dispatch_pairs(self) -> int:
        """1: the lean lane-per-env dispatch runs two agents of an env per trip where their footprints are disjoint."""
        return int(self.L.mgx_dispatch_pairs(self.h))

    @property
    def integer_bookkeeping(self) -> int:
        """1: the per-action bookkeeping counters live as integers beside the stat rows (written into them before any read)."""
        return int(self.L.mgx_integer_bookkeeping(self.h))

    @property
    def state_bytes(self) -> int:
        return int(self.L.mgx_state_bytes(self.h))


class MettaGrid:
    """One env with the reference's pybind surface (``MettaGrid(env_cfg, map, seed)``; here ``env_cfg`` is a compiled
    ``Program`` or a ``GameSpec``).  Buffers are caller-visible numpy arrays, stored by reference like the
    reference does (cpp/bindings/mettagrid_c.cpp:1165-1184): ``step()`` writes into them."""

    def __init__(self, env_cfg, map, seed: int, device: int = 0) -> None:  # noqa: A002 - reference argument name
        H, W = len(map), len(map[0])
        prog = env_cfg if isinstance(env_cfg, Program) else compile_spec(env_cfg, H, W)
        self.prog = prog
        self._b = BatchedMettaGrid(prog, prog.class_map(map), [int(seed) & 0xFFFFFFFF], device=device, buffers="host")
        w = prog.words
        self.obs_width, self.obs_height = int(w[K.H_OBS_WIDTH]), int(w[K.H_OBS_HEIGHT])
        self.max_steps = int(w[K.H_MAX_STEPS])
        self.map_width, self.map_height = W, H
        self.object_type_names = list(prog.type_names)
        self.resource_names = list(prog.resource_names)
        self._num_agents = prog.num_agents
        self._profiling = False

    # -- buffers --
    def set_buffers(self, observations, terminals, truncations, rewards, actions, vibe_actions=None) -> None:
        n = self._num_agents
        if vibe_actions is None:
            vibe_actions = np.zeros(n, np.int32)
        want = ((observations, np.uint8), (terminals, np.bool_), (truncations, np.bool_), (rewards, np.float32),
                (actions, np.int32), (vibe_actions, np.int32))
        for arr, dt in want:  # pybind .noconvert(): dtype and C-contiguity are part of the signature
            if not isinstance(arr, np.ndarray) or arr.dtype != dt or not arr.flags.c_contiguous:
                raise TypeError("set_buffers(): incompatible function arguments (dtype / contiguity)")
        if observations.ndim != 3:
            raise RuntimeError(f"observations has {observations.ndim} dimensions but expected 3")
        if observations.shape[0] != n or observations.shape[2] != 3:
            raise RuntimeError(f"observations has shape {list(observations.shape)} but expected [{n}, [something], 3]")
        for name, arr in (("terminals", terminals), ("truncations", truncations), ("vibe_actions", vibe_actions),
                          ("rewards", rewards)):
            if arr.ndim != 1 or arr.shape[0] != n:
                raise RuntimeError(f"{name} has the wrong shape")
        b = self._b
        b.obs, b.terminals, b.truncations, b.rewards, b.actions, b.vibe_actions = (
            observations, terminals, truncations, rewards, actions, vibe_actions)
        b._bind(*(a.ctypes.data for a in (observations, terminals, truncations, rewards, actions, vibe_actions)),
                mem_kind=0, rows=observations.shape[0], tokens=observations.shape[1])

    def step(self) -> None:
        b = self._b
        if b.actions.ndim != 1:
            raise RuntimeError("actions must be 1D array")
        if b.actions.shape[0] != self._num_agents:
            raise RuntimeError("actions has the wrong shape")
        b.step(check_errors=True)

    def observations(self): return self._b.obs
    def terminals(self): return self._b.terminals
    def truncations(self): return self._b.truncations
    def rewards(self): return self._b.rewards
    def actions(self): return self._b.actions
    def vibe_actions(self): return self._b.vibe_actions

    def masks(self):
        return np.ones((self._num_agents, len(self.prog.action_names)), np.bool_)

    def get_episode_rewards(self): return self._b.episode_rewards()
    def get_episode_stats(self): return self._b.get_episode_stats(0)
    def action_success(self): return [bool(x) for x in self._b.action_success()]

    def get_game_stat(self, key: str):
        return self.get_episode_stats()["game"].get(key)

    def get_agent_stat(self, agent_id: int, key: str):
        st = self.get_episode_stats()["agent"]
        return st[agent_id].get(key) if 0 <= agent_id < len(st) else None

    @property
    def current_step(self) -> int:
        return int(self._b.current_steps()[0])

    def set_inventory(self, agent_id: int, inventory: dict) -> None:
        self._b.set_inventory(0, agent_id, inventory)

    def tag_index(self):
        from .mettagrid_c import TagIndex
        return TagIndex(self)

    # -- profiling properties (cpp/bindings/profiling_py.cpp:8-30); device time of the most recent step --
    def _timing(self) -> dict:
        if not self._profiling:
            self._b.set_profiling(True)
            self._profiling = True
            return {k: 0.0 for k in self._b.TIMING_SEGMENTS}
        try:
            return self._b.step_timing_segments_ms()
        except (MgxError, ValueError, RuntimeError):
            return {k: 0.0 for k in self._b.TIMING_SEGMENTS}

    @property
    def step_timing(self):
        """StepTimingStats: the fields of the reference's struct, filled from the engine's timing segments (a segment
        that covers several reference phases is reported under the first of them; see include/mgx.h MGX_T_*)."""
        t = self._timing()
        ns = {k: int(v * 1e6) for k, v in t.items()}
        from types import SimpleNamespace
        return SimpleNamespace(reset_ns=0, events_ns=0, actions_ns=ns["actions"], on_tick_ns=ns["tail"], aoe_ns=ns["aoe"],
                               observations_ns=ns["obs"], rewards_ns=ns["rewards"], truncation_ns=0, total_ns=sum(ns.values()))

    @property
    def last_obs_time_ns(self) -> int:
        return int(self._timing()["obs"] * 1e6)

    @property
    def obs_validation_stats(self):
        """The reference compares its two observation code paths when validation is compiled in; there is one path
        here, so the counters stay zero."""
        from types import SimpleNamespace
        return SimpleNamespace(comparison_count=0, mismatch_count=0, original_time_ns=0, optimized_time_ns=0)

    def grid_objects(self, min_row=-1, max_row=-1, min_col=-1, max_col=-1, ignore_types=()):
        objs = self._b.grid_objects(0)
        use_bounds = min_row >= 0 and max_row >= 0 and min_col >= 0 and max_col >= 0
        out = {}
        for oid, o in objs.items():
            if o["type_name"] in ignore_types:
                continue
            if use_bounds and not (min_row <= o["r"] < max_row and min_col <= o["c"] < max_col):
                continue
            out[oid] = o
        return out
