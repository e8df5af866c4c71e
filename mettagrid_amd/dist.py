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


# ---- packed observation rows (include/mgx.h mgx_pack_rows) --------------------------------------------------------------
def pack_rows(rows, scratch=None, stream=None):
    """Token rows u8 [n, T, 3] (a prefix of used tokens, then 0xFF padding) -> (counts int16 [n], packed u8 [total, 3]).
    CUDA tensors: libmgx's kernels on ``stream`` (a torch stream; default: the current one) + one host read of the total;
    CPU tensors (the gloo tests): the same result from torch ops."""
    import torch
    n, T = int(rows.shape[0]), int(rows.shape[1])
    if not rows.is_cuda:
        used = rows[:, :, 0] != 0xFF
        return used.sum(1).to(torch.int16), rows[used]
    from .engine import load_lib
    L = load_lib()
    st = stream if stream is not None else torch.cuda.current_stream(rows.device)
    with torch.cuda.stream(st):
        counts = torch.empty(n, dtype=torch.int16, device=rows.device)
        packed = torch.empty((n * T, 3), dtype=torch.uint8, device=rows.device)   # worst case; the used part is returned
        if scratch is None or scratch.numel() * 4 < L.mgx_pack_scratch_bytes(n):
            scratch = torch.empty((L.mgx_pack_scratch_bytes(n) + 3) // 4, dtype=torch.int32, device=rows.device)
        rc = L.mgx_pack_rows(rows.data_ptr(), n, T, counts.data_ptr(), packed.data_ptr(), n * T, scratch.data_ptr(), st.cuda_stream)
        if rc:
            raise RuntimeError(f"mgx_pack_rows failed ({rc})")
        import ctypes as C
        total, ovf = C.c_int64(0), C.c_int32(0)
        rc = L.mgx_pack_result(scratch.data_ptr(), n, C.byref(total), C.byref(ovf), st.cuda_stream)
        if rc or ovf.value:
            raise RuntimeError(f"mgx_pack_result failed ({rc}, {ovf.value} rows dropped)")
    return counts, packed[: total.value]


def unpack_rows(counts, packed, T: int, out=None, stream=None):
    """The inverse of ``pack_rows``: u8 [n, T, 3] with the 0xFF padding restored."""
    import torch
    n = int(counts.shape[0])
    if out is None:
        out = torch.empty((n, T, 3), dtype=torch.uint8, device=counts.device)
    if not counts.is_cuda:
        out.fill_(0xFF)
        used = torch.arange(T)[None, :] < counts.to(torch.int64)[:, None]
        out[used] = packed
        return out
    from .engine import load_lib
    L = load_lib()
    st = stream if stream is not None else torch.cuda.current_stream(counts.device)
    with torch.cuda.stream(st):
        scratch = torch.empty((L.mgx_pack_scratch_bytes(n) + 3) // 4, dtype=torch.int32, device=counts.device)
        if packed.numel() == 0:
            packed = torch.zeros((1, 3), dtype=torch.uint8, device=counts.device)
        rc = L.mgx_unpack_rows(packed.data_ptr(), counts.data_ptr(), n, T, out.data_ptr(), scratch.data_ptr(), st.cuda_stream)
        if rc:
            raise RuntimeError(f"mgx_unpack_rows failed ({rc})")
        scratch.record_stream(st)
    return out


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

    ``packed`` names token-row tensors (u8 [rows, T, 3]) that travel as their used prefixes (``pack_rows``: per-row counts +
    the used tokens back to back — about a third of the bytes at BASELINE's shapes; the root's seven xGMI links are what
    bounds the gather of BASELINE.json configs[4]).  The packing kernels read the live tensor on the side stream — they ARE
    the staging copy — and the byte count of every rank reaches the others through a small all-gather, which costs one host
    wait for the side stream per submit (the whole-row mode has none).  The root receives ``<name>_counts`` and
    ``<name>_packed`` (rank-major) and ``expand(name)`` rebuilds whole rows.

    The sends of slot ``s`` are on the (serial) side stream in front of the next copy into ``s``, so a slot is never
    rewritten under a send.  ``result()`` returns the root's view of the most recent completed gather (dict name -> tensor
    of world * rows) or None on the other ranks; ``result_of(k)`` that of submit number ``k`` while its slot is still alive.
    On CPU tensors (gloo) everything is synchronous.
    """

    def __init__(self, dist, root: int = 0, device=None, producer_stream=None, output_fence=None, slots: int = 2, packed=()) -> None:
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
        self.packed = tuple(packed)
        self.pk = [dict() for _ in range(self.slots)]   # per slot: name -> (counts, packed, T) on this rank / root's gathered pieces

    def _alloc(self, tensors: dict) -> None:
        torch = self.torch
        plain = {n: t for n, t in tensors.items() if n not in self.packed}
        for s in range(self.slots):
            self.stage[s] = {n: torch.empty_like(t, dtype=torch.uint8 if t.dtype == torch.bool else t.dtype) for n, t in plain.items()}
            if self.rank == self.root:
                self.out[s] = {n: torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=self.stage[s][n].dtype,
                                              device=t.device) for n, t in plain.items()}

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
                if n in self.packed:               # the packing kernels read the live rows: they are this tensor's staging copy
                    counts, pk = pack_rows(t, stream=self.side if self.cuda else None)
                    self.pk[s][n] = (counts, pk, int(t.shape[1]), int(t.shape[0]))
                else:
                    self.stage[s][n].copy_(t)      # bool -> u8 here (gloo has no bool transport)
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
            for n in self.packed:
                if n not in tensors:
                    continue
                counts, pk, T, rows = self.pk[s][n]
                # every rank learns every rank's token total (the receive sizes): one small all-gather
                mine = torch.tensor([pk.shape[0]], dtype=torch.int64, device=pk.device)
                totals = [torch.zeros_like(mine) for _ in range(self.world)]
                dist.all_gather(totals, mine)
                totals = [int(x.item()) for x in totals]
                if self.rank == self.root:
                    all_counts = torch.empty(self.world * rows, dtype=torch.int16, device=pk.device)
                    all_packed = torch.empty((sum(totals), 3), dtype=torch.uint8, device=pk.device)
                    starts = [sum(totals[:p]) for p in range(self.world)]
                    all_counts[self.root * rows:(self.root + 1) * rows].copy_(counts)
                    all_packed[starts[self.root]:starts[self.root] + totals[self.root]].copy_(pk)
                    for peer in range(self.world):
                        if peer != self.root:
                            ops.append(dist.P2POp(dist.irecv, all_counts[peer * rows:(peer + 1) * rows], peer))
                            if totals[peer]:
                                ops.append(dist.P2POp(dist.irecv, all_packed[starts[peer]:starts[peer] + totals[peer]], peer))
                    self.pk[s][n] = (all_counts, all_packed, T, self.world * rows)
                else:
                    ops.append(dist.P2POp(dist.isend, counts, self.root))
                    if totals[self.rank]:
                        ops.append(dist.P2POp(dist.isend, pk, self.root))
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
        out = {n: (t.to(self.torch.bool) if n in bool_names else t) for n, t in self.out[s].items()}
        for n, (counts, pk, T, rows) in self.pk[s].items():
            out[n + "_counts"], out[n + "_packed"] = counts, pk
        return out

    def expand(self, name: str, k: int = None):
        """Whole rows u8 [world * rows, T, 3] of packed tensor ``name`` of the most recent (or k-th) submit, on the root."""
        if self.rank != self.root:
            return None
        s = self.last if k is None else k % self.slots
        if self.cuda:
            self.done[s].synchronize()
        counts, pk, T, rows = self.pk[s][name]
        return unpack_rows(counts, pk, T)

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
