"""Batched PufferEnv-shaped environment over the MI355X step engine.

Mirrors the surface PufferLib drives on the reference's ``MettaGridPufferEnv``
(/root/reference/python/src/mettagrid/envs/mettagrid_puffer_env.py): ``reset(seed) -> (obs, infos)``,
``step(actions) -> (obs, rewards, terminals, truncations, infos)``, the same action encodings (1-D combined index or
``[N, 2]`` primary/vibe columns, :305-394), the same dtypes (:59-63) and lazy auto-reset (an env whose agents are all
terminal or all truncated is restarted at the START of the next ``step`` call, then stepped, :299-302) — for
``num_agents = E * A`` agents at once, with every buffer resident in HBM.  pufferlib itself is not imported.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

from .compiler import Program
from .engine import BatchedMettaGrid


def split_action_names(action_names: list) -> tuple:
    """PolicyEnvInterface._split_action_names (python/src/mettagrid/policy/policy_env_interface.py:110-119)."""
    primary = [a for a in action_names if not a.startswith("change_vibe_")]
    vibe = [a for a in action_names if a.startswith("change_vibe_")]
    return primary, vibe


def decode_actions(actions, num_primary: int, vibe_action_ids, xp=np):
    """Action decoding rules of MettaGridPufferEnv.step (:305-381).  Returns (core_actions, vibe_actions or None).
    ``xp`` is numpy or torch (the arrays stay on their device)."""
    num_vibe = len(vibe_action_ids)
    if num_primary <= 0:
        raise ValueError("Environment must expose at least one non-vibe action")
    a = actions
    vibe = None
    if a.ndim == 2:
        if a.shape[1] == 1:
            core = a[:, 0]
        elif a.shape[1] == 2:
            core = a[:, 0]
            if num_vibe <= 0:
                raise ValueError("Received 2D actions with vibe column, but environment has no configured vibe action space")
            raw = a[:, 1].to(xp.int64) if xp is not np else a[:, 1].astype(np.int64)
            if bool((raw < 0).any()):
                raise ValueError(f"Vibe actions must be non-negative, got min={int(raw.min())}")
            if bool((raw >= num_vibe).any()):
                raise ValueError(f"Vibe action indices out of range [0,{num_vibe}), min={int(raw.min())} max={int(raw.max())}")
            vibe = vibe_action_ids[raw]
        else:
            raise ValueError(f"Expected step actions shape [num_agents] or [num_agents,2], got {tuple(a.shape)}")
    elif a.ndim == 1:
        a64 = a.to(xp.int64) if xp is not np else a.astype(np.int64)
        if bool((a64 < 0).any()):
            raise ValueError(f"Actions must be non-negative, got min={int(a64.min())}")
        core = a64
        enc = a64 >= num_primary
        if bool(enc.any()):
            if num_vibe <= 0:
                raise ValueError("Received encoded vibe actions, but environment has no configured vibe action space")
            max_valid = num_primary + num_primary * num_vibe
            if bool((a64 >= max_valid).any()):
                raise ValueError(f"Action indices out of range [0, {max_valid}), min={int(a64.min())} max={int(a64.max())}")
            off = a64 - num_primary
            core = xp.where(enc, off // num_vibe, a64)
            vibe = xp.where(enc, vibe_action_ids[(off % num_vibe) * enc], 0 * a64)
    else:
        raise ValueError(f"Expected step actions shape [num_agents] or [num_agents,2], got {tuple(a.shape)}")
    c64 = core.to(xp.int64) if xp is not np else core.astype(np.int64)
    if bool((c64 < 0).any()) or bool((c64 >= num_primary).any()):
        raise ValueError(f"Core actions out of range [0,{num_primary}), min={int(c64.min())} max={int(c64.max())}")
    return core, vibe


def split_supervisor_actions_inplace(teacher_actions, vibe_actions, *, num_primary_actions: int, vibe_action_ids_by_index) -> None:
    """Supervisor (teacher) labels in split-action id space -> primary labels + the simulator's vibe stream, in place:
    ``split_supervisor_actions_inplace`` of the reference (python/src/mettagrid/policy/supervisor_actions.py:8-40).  Labels
    in [0, P) are primary actions; labels in [P, P + V) are vibe v = label - P and fill ``vibe_actions`` with that vibe's
    engine action id, 0 elsewhere.  numpy arrays or torch tensors (torch: the range check is one device->host read)."""
    is_np = isinstance(teacher_actions, np.ndarray)
    t64 = teacher_actions.astype(np.int64) if is_np else teacher_actions.long()
    n_vibe = int(len(vibe_action_ids_by_index))
    max_id = num_primary_actions + n_vibe - 1
    invalid = (t64 < 0) | (t64 > max_id)
    if bool(invalid.any()):
        bad = int(np.flatnonzero(invalid)[0]) if is_np else int(invalid.nonzero()[0, 0])
        raise ValueError(f"Supervisor produced invalid action id {int(teacher_actions[bad])} for agent {bad}")
    primary = t64 < num_primary_actions
    if is_np:
        vibe_actions.fill(0)
        idx = t64[~primary] - num_primary_actions
        vibe_actions[~primary] = np.asarray(vibe_action_ids_by_index)[idx].astype(vibe_actions.dtype)
    else:
        import torch
        ids = torch.as_tensor(vibe_action_ids_by_index, device=teacher_actions.device).long()
        idx = (t64 - num_primary_actions).clamp(min=0)
        vibe_actions.copy_(torch.where(primary, torch.zeros_like(t64), ids[idx * (~primary)] if n_vibe else torch.zeros_like(t64))
                           .to(vibe_actions.dtype))


def decode_actions_unchecked(a, num_primary: int, vibe_action_ids):
    """decode_actions for torch tensors without the range checks (each of them is a device->host read)."""
    import torch
    num_vibe = len(vibe_action_ids)
    if a.ndim == 2:
        if a.shape[1] == 1:
            return a[:, 0], None
        return a[:, 0], vibe_action_ids[a[:, 1].to(torch.int64)]
    a64 = a.to(torch.int64)
    if num_vibe <= 0:
        return a64, None
    enc = a64 >= num_primary
    off = (a64 - num_primary).clamp(min=0)
    core = torch.where(enc, off // num_vibe, a64)
    vibe = torch.where(enc, vibe_action_ids[(off % num_vibe) * enc], torch.zeros_like(a64))
    return core, vibe


class MettaGridBatchedEnv:
    """E envs x A agents behind the PufferEnv call pattern.

    Two sources of maps for new episodes:

    * ``map_pool`` (uint16 [M, H, W], class index + 1): the maps are uploaded to the GPU once and finished envs restart
      ON THE DEVICE (``mgx_set_map_pool`` / ``mgx_set_auto_reset``): no host round trip per step, episode ``k`` of env
      ``e`` plays pool map ``(e + k * pool_stride) mod M``.  ``desync=True`` ends every env's FIRST episode early at a
      step drawn like the reference's ``EarlyResetHandler`` (envs/early_reset_handler.py:6-22:
      ``default_rng(seed).integers(1, max_steps + 1)``), so that envs sharing ``max_steps`` do not all restart on the
      same step.
    * ``map_fn(env_index, episode_index) -> uint16 class map [H, W]``: maps built on the host once per episode, as in the
      reference (simulator.py:83); the done test then costs a device->host read per step.

    Engine seeds: ``seed_fn(base, env, episode)`` (default: the seed given to ``reset`` plus the env index, constant
    across auto-resets like the reference's ``_current_seed``).

    With device buffers the returned tensors are ordered on the caller's current torch stream (no host synchronisation).
    ``validate_actions``: ``True`` — the reference's range checks on the action tensor are made by the decode kernel on the
    device and a bad id raises the reference's ValueError from a LATER ``step`` / ``check_actions()`` call (no host read on the
    good path; 1-D int32 tensors — other layouts take the host-synchronous path); ``"strict"`` — checked on the host before
    the step runs, as the reference does (a device->host read per check); ``False`` — no check.

    ``infos`` (the last element of what ``step`` returns): the reference's wrapper returns, from the step that ends an episode
    on, the dict StatsTracker.on_episode_end built (stats_tracker.py:26-76: ``game``, ``agent``, ``per_agent``, ``attributes``
    ...).  Here dozens of the E envs finish in any one step; their stats are reduced on the device before the restart wipes
    them (``episode_stats=True``: include/mgx.h "Episode-end statistics") and ``step`` returns the batch aggregate of the
    episodes that finished since the last non-empty ``infos``: ``{"episodes": n, "game": {key: mean over the episodes that
    hold the key}, "agent": {key: mean of the per-episode infos["agent"] values}, "game_count" / "agent_count",
    "episode_return": {mean, min, max}, "episode_length": {...}, "terminated": n, "attributes": {...}}`` — every
    ``stats_interval`` steps a snapshot is requested and the one requested before is returned if its copy has landed (no
    host synchronisation: the aggregate trails the step that produced it by ``stats_interval`` steps), ``{}`` otherwise.
    ``episode_infos()`` returns the finished episodes one by one in the reference's shape from a bounded device log
    (``episode_log`` records; ``log_per_agent`` adds ``per_agent``).
    """

    def __init__(self, prog: Program, num_envs: int, map_fn: Optional[Callable[[int, int], np.ndarray]] = None,
                 seed_fn: Optional[Callable[[int, int, int], int]] = None, device: int = 0, seed: int = 0,
                 buffers: str = "device", map_pool: Optional[np.ndarray] = None, pool_stride: int = 1,
                 desync: bool = False, validate_actions: bool = True, episode_stats: bool = True, stats_interval: int = 1,
                 episode_log: int = 0, log_per_agent: bool = False, specialize="auto") -> None:
        if (map_fn is None) == (map_pool is None):
            raise ValueError("give exactly one of map_fn and map_pool")
        self.prog = prog
        self.E = num_envs
        self.map_fn = map_fn
        self.map_pool = None if map_pool is None else np.ascontiguousarray(map_pool, dtype=np.uint16)
        self.pool_stride = pool_stride
        self.desync = desync
        self.validate_actions = validate_actions
        self._default_seed_fn = seed_fn is None
        self.seed_fn = seed_fn or (lambda base, env, episode: (base + env) & 0xFFFFFFFF)
        self._seed = seed
        self._device = device
        self._kind = buffers
        self.action_names, self.vibe_action_names = split_action_names(prog.action_names)
        self.num_agents = num_envs * prog.num_agents
        self.episode = np.zeros(num_envs, np.int64)
        self._eng: Optional[BatchedMettaGrid] = None
        self._vibe_ids = None
        self.supervisor = None          # object with step_batch(raw_observations, teacher_actions), see set_supervisor
        self.teacher_actions = None
        self.episode_stats = episode_stats
        self.stats_interval = max(1, int(stats_interval))
        self.episode_log = int(episode_log)
        self.log_per_agent = log_per_agent
        self.specialize = specialize      # BatchedMettaGrid(specialize=...): run-time code objects for this program (jit.py)
        self._steps = 0

    def set_supervisor(self, supervisor) -> None:
        """Supervisor-policy path of MettaGridPufferEnv (mettagrid_puffer_env.py:399-426): after every step
        ``supervisor.step_batch(observations, teacher_actions)`` fills ``teacher_actions`` (split-action id space) and the
        vibe labels are routed into the engine's vibe stream for the next step."""
        self.supervisor = supervisor

    def disable_supervisor(self) -> None:
        self.supervisor = None

    def _compute_supervisor_actions(self) -> None:
        eng = self.engine
        if self.teacher_actions is None:
            if self._kind == "device":
                import torch
                self.teacher_actions = torch.zeros(self.num_agents, dtype=torch.int32, device=eng.obs.device)
            else:
                self.teacher_actions = np.zeros(self.num_agents, np.int32)
        self.supervisor.step_batch(eng.obs, self.teacher_actions)
        split_supervisor_actions_inplace(self.teacher_actions, eng.vibe_actions, num_primary_actions=len(self.action_names),
                                         vibe_action_ids_by_index=self._vibe_ids if self._kind != "device" else self._vibe_ids)

    # spaces (shapes only; gymnasium is not a dependency)
    @property
    def single_observation_shape(self):
        return (self.prog.num_tokens, 3)

    @property
    def single_action_n(self) -> int:
        return len(self.action_names)

    @property
    def transport_action_n(self) -> int:
        n, v = len(self.action_names), len(self.vibe_action_names)
        return n * (v + 1) if v else n

    def _maps(self, envs) -> np.ndarray:
        H = int(self.prog.words[3]), int(self.prog.words[4])
        out = np.zeros((self.E,) + H, np.uint16)
        for e in envs:
            out[e] = self.map_fn(int(e), int(self.episode[e]))
        return out

    def _seeds(self) -> np.ndarray:
        if self._default_seed_fn:   # (base + env) mod 2^32 for every env at once
            return ((np.int64(self._seed) + np.arange(self.E, dtype=np.int64)) & 0xFFFFFFFF).astype(np.uint32)
        return np.array([self.seed_fn(self._seed, e, int(self.episode[e])) for e in range(self.E)], dtype=np.uint32)

    def early_end_steps(self) -> np.ndarray:
        """EarlyResetHandler.on_episode_start for every env: one draw from a generator seeded with the env's seed."""
        ms = int(self.prog.words[11])   # MGX_H_MAX_STEPS
        if ms <= 0:
            return np.zeros(self.E, np.uint32)
        from .early_reset import first_integers
        return first_integers(self._seeds(), ms)

    def reset(self, seed: Optional[int] = None):
        if seed is not None:
            self._seed = seed
        if self._eng is not None:
            self._eng.close()
        self.episode[:] = 0
        if self.map_pool is not None:
            M = self.map_pool.shape[0]
            first = self.map_pool[np.arange(self.E) % M]
            self._eng = BatchedMettaGrid(self.prog, first, self._seeds(), device=self._device, buffers=self._kind,
                                         specialize=self.specialize)
            self._eng.set_map_pool(self.map_pool)
            self._eng.set_auto_reset(True, self.pool_stride, self.early_end_steps() if self.desync else None)
        else:
            self._eng = BatchedMettaGrid(self.prog, self._maps(range(self.E)), self._seeds(), device=self._device,
                                         buffers=self._kind, specialize=self.specialize)
        if self.episode_stats:
            self._eng.set_episode_stats(True, self.episode_log, self.log_per_agent)
        self._steps = 0
        ids = [self.prog.action_names.index(n) for n in self.vibe_action_names]
        self._vibe_ids_host = np.asarray(ids, dtype=np.int32)
        if self._kind == "device":
            import torch
            self._vibe_ids = torch.tensor(ids, dtype=torch.int64, device=self._eng.obs.device)
            self._eng.caller_waits()
        else:
            self._vibe_ids = np.asarray(ids, dtype=np.int64)
        return self._eng.obs, {}

    @property
    def engine(self) -> BatchedMettaGrid:
        if self._eng is None:
            raise RuntimeError("Simulation is closed")
        return self._eng

    def _done_envs(self) -> np.ndarray:
        eng = self.engine
        A = self.prog.num_agents
        if self._kind == "device":
            eng.sync()
            term = eng.terminals.view(self.E, A).all(dim=1) | eng.truncations.view(self.E, A).all(dim=1)
            return term.cpu().numpy()
        return eng.terminals.reshape(self.E, A).all(1) | eng.truncations.reshape(self.E, A).all(1)

    # ---- infos ----
    def _infos_from_totals(self, tot: Optional[dict]) -> dict:
        """Batch aggregate of the finished episodes in the vocabulary of StatsTracker.on_episode_end (stats_tracker.py:26-76)."""
        if not tot or tot["episodes"] == 0:
            return {}
        n = tot["episodes"]
        words = self.prog.words
        return {"episodes": n,
                "game": {k: v / tot["game_count"][k] for k, v in tot["game_sum"].items()},
                "agent": {k: v / tot["agent_count"][k] for k, v in tot["agent_sum"].items()},
                "game_count": tot["game_count"], "agent_count": tot["agent_count"],
                "episode_return": {"mean": tot["return_sum"] / n, "min": tot["return_min"], "max": tot["return_max"]},
                "episode_length": {"mean": tot["length_sum"] / n, "min": tot["length_min"], "max": tot["length_max"]},
                "terminated": int(tot["terminated"]),
                "attributes": {"map_w": int(words[4]), "map_h": int(words[3]), "max_steps": int(words[11]), "seed": self._seed}}

    def _step_infos(self) -> dict:
        if not self.episode_stats:
            return {}
        eng = self.engine
        self._steps += 1
        if self.map_pool is None or self._kind != "device":   # these paths synchronise with the device every step anyway
            return self._infos_from_totals(eng.drain_episode_stats()) if self._steps % self.stats_interval == 0 else {}
        out = {}
        if self._steps % self.stats_interval == 0:
            out = self._infos_from_totals(eng.fetch_episode_stats(wait=False))   # the snapshot requested one interval ago
            eng.request_episode_stats()
        return out

    def check_actions(self, wait: bool = True) -> None:
        """Raise the reference's ValueError for a joint action id the device-side decode found out of range
        (mettagrid_puffer_env.py:336-360).  ``step`` calls this with ``wait=False`` (no synchronisation: an error surfaces at
        a later step, the offending ids having been played as invalid actions); call it with ``wait=True`` for certainty."""
        flags, row, value = self.engine.poll_action_errors(wait)
        if flags & 1:
            raise ValueError(f"Actions must be non-negative, got {value} for agent {row}")
        if flags & 2:
            n, v = len(self.action_names), len(self.vibe_action_names)
            raise ValueError(f"Action indices out of range [0, {n + n * v if v else n}), got {value} for agent {row}")

    def episode_infos(self) -> list:
        """The episodes that finished since the last call, one dict each in the reference's shape (stats_tracker.py:26-76):
        ``game``, ``agent``, ``per_agent`` (with ``log_per_agent``), ``episode_rewards``, ``attributes`` (seed, map_w, map_h,
        steps, max_steps) + ``env`` / ``episode`` / ``map_index``.  Needs ``episode_log`` > 0; synchronises with the device."""
        recs, dropped = self.engine.drain_episode_log()
        words = self.prog.words
        out = []
        for r in recs:
            info = {"game": r["game"], "agent": r["agent"], "episode_rewards": r["episode_rewards"],
                    "attributes": {"seed": r["seed"], "map_w": int(words[4]), "map_h": int(words[3]), "steps": r["steps"],
                                   "max_steps": int(words[11])},
                    "env": r["env"], "episode": r["episode"], "map_index": r["map_index"]}
            if "per_agent" in r:
                info["per_agent"] = {str(i): dct for i, dct in enumerate(r["per_agent"])}
            out.append(info)
        self.episodes_dropped = dropped
        return out

    def step(self, actions):
        eng = self.engine
        if self.map_pool is None:  # host map source: the done test needs the flags on the host
            done = self._done_envs()
            if done.any():  # lazy auto-reset at the start of the next step (mettagrid_puffer_env.py:299-302)
                idx = np.nonzero(done)[0]
                if self.episode_stats:
                    eng.record_episodes(done)
                self.episode[idx] += 1
                eng.reset_envs(done, self._maps(idx), self._seeds())
        if self._kind == "device":
            import torch
            a = actions if isinstance(actions, torch.Tensor) else torch.as_tensor(np.asarray(actions), device=eng.obs.device)
            if self.supervisor is None and self.validate_actions != "strict" and a.ndim == 1 and a.dtype == torch.int32 and a.is_contiguous():
                # one kernel on the engine's stream instead of a dozen elementwise torch kernels; it also makes the reference's
                # range checks (mettagrid_puffer_env.py:336-360) and raises a flag the host reads without synchronising
                if tuple(a.shape) != tuple(eng.actions.shape):
                    raise ValueError(f"Expected {tuple(eng.actions.shape)} actions, got {tuple(a.shape)}")
                if self.validate_actions:
                    self.check_actions(wait=False)      # a bad id of an EARLIER step whose kernel has run by now
                eng.wait_for_caller()
                eng.set_joint_actions(a, len(self.action_names), self._vibe_ids_host)
            else:
                if self.validate_actions:
                    core, vibe = decode_actions(a, len(self.action_names), self._vibe_ids, xp=torch)
                else:
                    core, vibe = decode_actions_unchecked(a, len(self.action_names), self._vibe_ids)
                    if self.supervisor is not None and vibe is not None and a.ndim == 1:
                        # the reference only overwrites the vibe stream when some action of the batch carries a vibe
                        # (:361-381); decided on the device, no host read
                        vibe = torch.where((a >= len(self.action_names)).any(), vibe, eng.vibe_actions.to(vibe.dtype))
                if tuple(core.shape) != tuple(eng.actions.shape):
                    raise ValueError(f"Expected {tuple(eng.actions.shape)} core actions, got {tuple(core.shape)}")
                eng.actions.copy_(core.to(torch.int32))
                if vibe is not None:
                    eng.vibe_actions.copy_(vibe.to(torch.int32))
                elif self.supervisor is None:   # with a supervisor the teacher's vibe labels of the last step stay
                    eng.vibe_actions.zero_()    # (mettagrid_puffer_env.py:389-391)
                eng.wait_for_caller()   # the engine's kernels read the actions written on the caller's stream ...
            eng.step()
            eng.caller_waits()      # ... and whatever the caller enqueues next sees this step's results
            if self.supervisor is not None:
                self._compute_supervisor_actions()
        else:
            a = np.asarray(actions)
            core, vibe = decode_actions(a, len(self.action_names), self._vibe_ids, xp=np)
            if core.shape != eng.actions.shape:
                raise ValueError(f"Expected {eng.actions.shape} core actions, got {core.shape}")
            np.copyto(eng.actions, core.astype(np.int32))
            if vibe is not None:
                np.copyto(eng.vibe_actions, vibe.astype(np.int32))
            elif self.supervisor is None:       # mettagrid_puffer_env.py:389-391
                eng.vibe_actions.fill(0)
            eng.step()
            if self.supervisor is not None:
                self._compute_supervisor_actions()
        return eng.obs, eng.rewards, eng.terminals, eng.truncations, self._step_infos()

    def close(self) -> None:
        if self._eng is not None:
            self._eng.close()
            self._eng = None


class MettaGridParallelEnv:
    """PettingZoo ``ParallelEnv`` call pattern over ONE env of the engine — the surface of the reference's
    ``MettaGridPettingZooEnv`` (python/src/mettagrid/envs/pettingzoo_env.py:22-230): integer agent ids, ``reset(seed) ->
    (obs dict, info dict)``, ``step({agent: action}) -> (obs, rewards, terminations, truncations, infos)`` as dicts keyed by
    agent id, finished agents dropped from ``agents``, one shared observation / action space for all agents.  pettingzoo and
    gymnasium themselves are not imported: spaces are described by shape / size."""

    def __init__(self, env_cfg, map, seed: int = 0, device: int = 0) -> None:  # noqa: A002 - reference argument name
        from .engine import MettaGrid
        self._make = lambda s: MettaGrid(env_cfg, map, s, device=device)
        self._seed = seed
        self._sim = self._make(seed)
        names = self._sim.prog.action_names
        self._action_indices = list(range(len(names)))      # PolicyEnvInterface.action_names -> engine action ids
        self.possible_agents = list(range(self._sim._num_agents))
        self.agents = list(self.possible_agents)
        self.observation_shape = (self._sim.prog.num_tokens, 3)

    def reset(self, seed: Optional[int] = None, options=None):
        if seed is not None:
            self._seed = seed
        self._sim._b.close()
        self._sim = self._make(self._seed)
        self.agents = list(self.possible_agents)
        obs = self._sim.observations()
        return {a: obs[a] for a in self.agents}, {a: {} for a in self.agents}

    def step(self, actions: dict):
        sim = self._sim
        arr = sim.actions()
        for a in self.agents:
            if a in actions:
                idx = int(np.asarray(actions[a], dtype=np.int32).reshape(()).item())
                if idx < 0 or idx >= len(self._action_indices):
                    raise ValueError(f"Action index {idx} out of range for PettingZoo action space.")
                arr[a] = self._action_indices[idx]
        sim.step()
        obs, rew, term, trunc = sim.observations(), sim.rewards(), sim.terminals(), sim.truncations()
        out = ({a: obs[a] for a in self.agents}, {a: float(rew[a]) for a in self.agents}, {a: bool(term[a]) for a in self.agents},
               {a: bool(trunc[a]) for a in self.agents}, {a: {} for a in self.agents})
        self.agents = [a for a in self.agents if not (out[2][a] or out[3][a])]
        return out

    def observation_space_shape(self, agent: int):
        return self.observation_shape

    def action_space_n(self, agent: int) -> int:
        return len(self._action_indices)

    def state(self) -> np.ndarray:   # pettingzoo_env.py:178-190: a zero vector of the flattened observation size
        return np.zeros(len(self.possible_agents) * int(np.prod(self.observation_shape)), np.uint8)

    @property
    def max_num_agents(self) -> int:
        return len(self.possible_agents)

    @property
    def max_steps(self) -> int:
        return self._sim.max_steps

    def render(self) -> None:
        pass

    def close(self) -> None:
        if self._sim is not None:
            self._sim._b.close()
            self._sim = None
