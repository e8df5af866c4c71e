"""Env groups: one batch of envs stepped as G engines on G streams of the same GPU, pipelined against each other.

Why.  The two kernels of a tick bind on different things: the world update is a chain of dependent memory accesses at
one wavefront per SIMD (latency; it leaves the issue slots idle), the observation kernel is instruction-issue and LDS
bound.  Run back to back over the whole batch they add up.  With the batch cut into two groups, group B's world update
runs beside group A's observation kernel and the other way round; ``mgx_chain_world`` keeps the stagger (the world
update of one group starts when the other group's has finished), so the two never fall into lock-step.

Semantics are unchanged: envs are independent, every env advances exactly one tick per ``step()``, rows
of the caller-visible buffers keep the global order (row = env * A + agent) because each engine writes into its slice of
one set of tensors.  The overlap spans ``step()`` calls: it needs a caller that does not wait for the device between
steps — actions for step k+1 already enqueued on the caller's stream, or, in a training loop, the usual double
buffering (policy on group A's observations while group B steps; ``step_group``).  Reference: the same idea as
pufferlib's multi-buffer vecenv, which the reference's trainer uses to overlap env workers with the policy
(python/src/mettagrid/envs/mettagrid_puffer_env.py is one such worker).
"""
from __future__ import annotations

import numpy as np

from .engine import BatchedMettaGrid


class EnvGroups:
    def __init__(self, prog, class_maps, seeds, device: int = 0, groups: int = 2, specialize="auto") -> None:
        import torch
        cm = np.ascontiguousarray(class_maps, dtype=np.uint16)
        E = cm.shape[0]
        if groups < 1 or E % groups:
            raise ValueError(f"{E} envs do not split into {groups} equal groups")
        self.prog, self.E, self.A, self.T, self.G, self.device = prog, E, prog.num_agents, prog.num_tokens, groups, device
        seeds = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.uint32), (E,)))
        dev = torch.device("cuda", device)
        rows = E * self.A
        self.obs = torch.empty((rows, self.T, 3), dtype=torch.uint8, device=dev)
        self.terminals = torch.empty(rows, dtype=torch.bool, device=dev)
        self.truncations = torch.empty(rows, dtype=torch.bool, device=dev)
        self.rewards = torch.empty(rows, dtype=torch.float32, device=dev)
        self.actions = torch.zeros(rows, dtype=torch.int32, device=dev)
        self.vibe_actions = torch.zeros(rows, dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        eg = E // groups
        self.envs_per_group = eg
        self.engines = []
        for g in range(groups):
            r0, r1 = g * eg * self.A, (g + 1) * eg * self.A
            bufs = {n: getattr(self, n)[r0:r1] for n in ("obs", "terminals", "truncations", "rewards", "actions", "vibe_actions")}
            self.engines.append(BatchedMettaGrid(prog, cm[g * eg:(g + 1) * eg], seeds[g * eg:(g + 1) * eg], device=device, buffers=bufs,
                                                 specialize=specialize))
        if groups > 1:   # ring: world(g) after world(g-1); world(0) of the next step after world(G-1) of this one
            for g in range(groups):
                self.engines[g].chain_world_after(self.engines[g - 1])

    # ---- stepping ----
    def step(self) -> None:
        for eng in self.engines:
            eng.step()

    def step_group(self, g: int) -> None:
        """One tick of group ``g`` only (double-buffered training loops)."""
        self.engines[g].step()

    def rows(self, g: int) -> slice:
        return slice(g * self.envs_per_group * self.A, (g + 1) * self.envs_per_group * self.A)

    def sync(self) -> None:
        for eng in self.engines:
            eng.sync()

    def wait_for_caller(self) -> None:
        for eng in self.engines:
            eng.wait_for_caller()

    def caller_waits(self) -> None:
        for eng in self.engines:
            eng.caller_waits()

    def wait_before_outputs(self, event) -> None:
        for eng in self.engines:
            eng.wait_before_outputs(event)

    def poll_errors(self):
        bits, first = 0, -1
        for g, eng in enumerate(self.engines):
            b, f = eng.poll_errors()
            if b and first < 0:
                first = g * self.envs_per_group + f
            bits |= b
        return bits, first

    def set_profiling(self, on: bool) -> None:
        for eng in self.engines:
            eng.set_profiling(on)

    def engine_of(self, env: int):
        """(engine, local env index) of a global env."""
        return self.engines[env // self.envs_per_group], env % self.envs_per_group

    def snapshot(self) -> dict:
        self.sync()
        parts = [eng.snapshot() for eng in self.engines]
        return {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}

    def close(self) -> None:
        for eng in self.engines:
            eng.close()
        self.engines = []
