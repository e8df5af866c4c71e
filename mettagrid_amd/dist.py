"""Multi-GPU sharding of the step path: one process per GPU, envs partitioned by index, no exchange inside a tick.

Envs are fully independent (no cross-env state, per-env RNG), so rank r of W owns the global envs
[r*E, (r+1)*E) and the tick needs no collective.  What a consumer may want afterwards is the rank-ordered
concatenation of the per-agent outputs on ONE rank (the trainer): ``GatherToRoot`` ships them with grouped point-to-point
sends — every peer writes its rows straight to the root over its own xGMI link (7 links x ~153 GB/s into the root), which
is what SURVEY.md §8e prescribes for the 629 MB/GPU observation payload; a ring all-gather would be bound by one link.
The sends run on a side stream from double-buffered staging copies, so step k+1 overlaps the gather of step k; the
step that next overwrites the gathered buffers is ordered behind the staging copy (event ``copied``).
Rows keep the global order: row = global_env * A + agent.  (gloo on CPU in the tests, RCCL = backend "nccl" on GPUs.)
"""
from __future__ import annotations

import numpy as np


def env_shard(rank: int, world: int, envs_per_rank: int) -> range:
    """Global env indices owned by ``rank`` (weak scaling: every rank owns ``envs_per_rank`` envs)."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    return range(rank * envs_per_rank, (rank + 1) * envs_per_rank)


def shard_seeds(rank: int, world: int, envs_per_rank: int) -> np.ndarray:
    """Engine seed of a global env = its global index (SURVEY.md §8d rung 2)."""
    r = env_shard(rank, world, envs_per_rank)
    return np.arange(r.start, r.stop, dtype=np.uint32)


class GatherToRoot:
    """Per-step gather of named per-rank tensors to ``root`` (rank-major rows), pipelined behind the producer.

    ``submit(tensors)`` — called right after the producing work has been enqueued on ``producer_stream`` — copies the
    tensors into staging slot ``k % slots`` on a side stream and issues the grouped send/recv there.  Two orderings keep the
    pipeline sound with no host synchronisation:

    * the staging copy starts only when the producer has finished the step (event ``ready``), and
    * whatever overwrites the live tensors next starts only when the staging copy has finished (event ``copied``).
      By default the whole producer stream waits for it.  With ``output_fence`` (a callable taking the event: the engine's
      ``wait_before_outputs``) only the kernels that WRITE those tensors wait — the next step's world update, which does
      not touch observations / rewards / terminals / truncations, runs beside the copy.

    The sends of slot ``s`` are on the (serial) side stream in front of the next copy into ``s``, so a slot is never
    rewritten under a send.  ``result()`` returns the root's view of the most recent completed gather (dict name -> tensor
    of world * rows) or None on the other ranks; ``result_of(k)`` that of submit number ``k`` while its slot is still alive.
    On CPU tensors (gloo) everything is synchronous.
    """

    def __init__(self, dist, root: int = 0, device=None, producer_stream=None, output_fence=None, slots: int = 2) -> None:
        import torch
        self.dist, self.root, self.torch = dist, root, torch
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device
        self.cuda = device is not None and torch.device(device).type == "cuda"
        self.producer = producer_stream
        self.output_fence = output_fence
        self.side = torch.cuda.Stream(device=device) if self.cuda else None
        self.slots = max(2, int(slots))
        self.stage = [None] * self.slots
        self.out = [None] * self.slots
        self.done = [None] * self.slots
        self.copied = [None] * self.slots
        self.k = 0

    def _alloc(self, tensors: dict) -> None:
        torch = self.torch
        for s in range(self.slots):
            self.stage[s] = {n: torch.empty_like(t, dtype=torch.uint8 if t.dtype == torch.bool else t.dtype) for n, t in tensors.items()}
            if self.rank == self.root:
                self.out[s] = {n: torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=self.stage[s][n].dtype,
                                              device=t.device) for n, t in tensors.items()}

    def submit(self, tensors: dict) -> None:
        torch, dist = self.torch, self.dist
        if self.stage[0] is None:
            self._alloc(tensors)
        s = self.k % self.slots
        self.k += 1
        if self.cuda:
            producer = self.producer if self.producer is not None else torch.cuda.current_stream(self.device)
            ready = torch.cuda.Event()
            ready.record(producer)
            self.side.wait_event(ready)
            ctx = torch.cuda.stream(self.side)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx:
            for n, t in tensors.items():
                self.stage[s][n].copy_(t)          # bool -> u8 here (gloo has no bool transport)
            if self.cuda:
                # the step that overwrites the live tensors must not start (writing them) before this copy has read them
                self.copied[s] = torch.cuda.Event()
                self.copied[s].record(self.side)
                if self.output_fence is not None:
                    self.output_fence(self.copied[s])
                else:
                    producer.wait_event(self.copied[s])
            ops = []
            for n, st in self.stage[s].items():
                if self.rank == self.root:
                    rows = st.shape[0]
                    self.out[s][n][self.root * rows:(self.root + 1) * rows].copy_(st)
                    for peer in range(self.world):
                        if peer != self.root:
                            ops.append(dist.P2POp(dist.irecv, self.out[s][n][peer * rows:(peer + 1) * rows], peer))
                else:
                    ops.append(dist.P2POp(dist.isend, st, self.root))
            if ops:
                for req in dist.batch_isend_irecv(ops):   # one ncclGroupStart/End: all peers write concurrently
                    req.wait()
            if self.cuda:
                self.done[s] = torch.cuda.Event()
                self.done[s].record(self.side)
        self.last = s

    def _view(self, s: int, bool_names):
        if self.cuda:
            self.done[s].synchronize()
        return {n: (t.to(self.torch.bool) if n in bool_names else t) for n, t in self.out[s].items()}

    def result(self, bool_names=("terminals", "truncations")):
        if self.k == 0 or self.rank != self.root:
            return None
        return self._view(self.last, bool_names)

    def result_of(self, k: int, bool_names=("terminals", "truncations")):
        """Root's rows of submit number ``k`` (0-based); only the last ``slots`` submits are still held."""
        if self.rank != self.root:
            return None
        if not self.k - self.slots <= k < self.k or k < 0:
            raise IndexError(f"submit {k} is no longer held (have {max(0, self.k - self.slots)}..{self.k - 1})")
        return self._view(k % self.slots, bool_names)

    def finish(self) -> None:
        if self.cuda and self.k:
            self.side.synchronize()


def max_over_ranks(dist, seconds: float, device="cpu") -> float:
    """bench.py contract: the timed region's duration is the MAX over ranks."""
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
