"""Multi-GPU sharding of the step path: one process per GPU, envs partitioned by index, no exchange inside a tick.

Envs are fully independent (no cross-env state, per-env RNG), so rank r of W owns the global envs
[r*E, (r+1)*E).  The tick itself needs no collective.  What a trainer may want afterwards is the rank-ordered
concatenation of the small per-agent outputs (rewards f32, terminals/truncations bool) and — optionally — of the
observation buffer; ``gather_outputs`` does that with ``all_gather_into_tensor`` (RCCL over xGMI on GPUs, gloo on
CPU in the tests).  Rows keep the global order: row = global_env * A + agent.
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


def gather_outputs(dist, rewards, terminals, truncations, observations=None):
    """all_gather the per-rank output rows into rank-major global tensors.  ``dist`` is ``torch.distributed``."""
    import torch
    world = dist.get_world_size()

    def gather(t):
        out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        if t.dtype == torch.bool:  # gloo has no bool all_gather; ship as u8
            tmp = torch.empty(out.shape, dtype=torch.uint8, device=t.device)
            dist.all_gather_into_tensor(tmp, t.to(torch.uint8).contiguous())
            return tmp.to(torch.bool)
        dist.all_gather_into_tensor(out, t.contiguous())
        return out

    res = {"rewards": gather(rewards), "terminals": gather(terminals), "truncations": gather(truncations)}
    if observations is not None:
        res["observations"] = gather(observations)
    return res


def max_over_ranks(dist, seconds: float, device="cpu") -> float:
    """bench.py contract: the timed region's duration is the MAX over ranks."""
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
