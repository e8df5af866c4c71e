"""GPU: the per-step gather pipeline of mettagrid_amd/dist.py under bench.py's own stepping pattern — every step enqueued
behind the previous one with NO host synchronisation in between, staging copies and sends on a side stream.  On CPU
tensors (tests/test_dist_gloo.py) everything is synchronous, so the stream / event ordering only ever runs here: the rows the
root holds for step k must be step k's, not rows the next step has already started to overwrite."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

E, STEPS = 2048, 10


def test_pack_rows_kernels_round_trip():
    """mgx_pack_rows / mgx_unpack_rows (csrc/mgx_pack.hip): counts, offsets and bytes against the torch statement of the same
    thing, on real observation rows and on edge cases (empty rows, full rows, a row count that is not a multiple of anything)."""
    import torch
    from mettagrid_amd import presets
    from mettagrid_amd.compiler import compile_spec
    from mettagrid_amd.dist import pack_rows, unpack_rows
    from mettagrid_amd.engine import BatchedMettaGrid
    from mettagrid_amd.mapgen import random_class_maps

    prog = compile_spec(presets.rung3_spec(), 32, 32, max_objects=192)
    cms = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, range(700))
    eng = BatchedMettaGrid(prog, cms, np.arange(700, dtype=np.uint32))
    eng.step()
    eng.sync()
    rows = eng.obs.clone()
    rows[5] = 0xFF                                       # an empty row
    rows[7, :, :] = 3                                    # a full row (no padding at all)
    T = rows.shape[1]
    counts, packed = pack_rows(rows)
    torch.cuda.synchronize()
    c_ref, p_ref = pack_rows(rows.cpu())
    assert torch.equal(counts.cpu(), c_ref) and torch.equal(packed.cpu(), p_ref)
    assert int(counts[5]) == 0 and int(counts[7]) == T and packed.shape[0] == int(c_ref.to(torch.int64).sum())
    assert packed.numel() < 0.6 * rows.numel()           # what the gather saves
    back = unpack_rows(counts, packed, T)
    torch.cuda.synchronize()
    assert torch.equal(back, rows)
    for n in (1, 63, 4097, 11201):                       # across the scan's workgroup boundary (4 096 rows)
        sub = rows[:n] if n <= rows.shape[0] else rows.repeat(2, 1, 1)[:n]
        c, p = pack_rows(sub)
        assert torch.equal(unpack_rows(c, p, T), sub), n
    eng.close()


@pytest.mark.parametrize("fence", ["producer_stream", "output_kernels", "packed"])
def test_gather_rows_are_the_steps_own(fence):
    import torch
    import torch.distributed as dist
    from mettagrid_amd import presets
    from mettagrid_amd.compiler import compile_spec
    from mettagrid_amd.dist import GatherToRoot, shard_seeds
    from mettagrid_amd.groups import EnvGroups
    from mettagrid_amd.mapgen import random_class_maps

    prog = compile_spec(presets.rung3_spec(), 32, 32, max_objects=192)
    cms = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, range(E))
    seeds = shard_seeds(0, 1, E)
    A, n = prog.num_agents, len(prog.action_names)
    gen = torch.Generator(device="cuda").manual_seed(7)
    acts = torch.randint(0, n, (STEPS, E * A), dtype=torch.int32, device="cuda", generator=gen)
    vibes = torch.randint(0, n, (STEPS, E * A), dtype=torch.int32, device="cuda", generator=gen)
    names = ("observations", "rewards", "terminals", "truncations")

    def outputs(grp):
        return {"observations": grp.obs, "rewards": grp.rewards, "terminals": grp.terminals, "truncations": grp.truncations}

    # what every step must look like: the same batch stepped with a host synchronisation and a clone after every step
    ref = EnvGroups(prog, cms, seeds, device=0, groups=1)
    want = []
    for t in range(STEPS):
        ref.actions.copy_(acts[t])
        ref.vibe_actions.copy_(vibes[t])
        torch.cuda.synchronize()
        ref.step()
        ref.sync()
        want.append({k: v.clone() for k, v in outputs(ref).items()})
    ref.close()
    assert not torch.equal(want[0]["observations"], want[1]["observations"])   # steps differ: a stale row would show

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        grp = EnvGroups(prog, cms, seeds, device=0, groups=1)
        eng = grp.engines[0]
        ext = torch.cuda.ExternalStream(eng.stream, device=torch.device("cuda", 0))
        gather = GatherToRoot(dist, root=0, device=torch.device("cuda", 0), producer_stream=ext, slots=STEPS,
                              output_fence=grp.wait_before_outputs if fence in ("output_kernels", "packed") else None,
                              packed=("observations",) if fence == "packed" else ())
        torch.cuda.synchronize()
        for t in range(STEPS):             # bench.py one_step: nothing here waits for the device
            with torch.cuda.stream(ext):
                eng.actions.copy_(acts[t], non_blocking=True)
                eng.vibe_actions.copy_(vibes[t], non_blocking=True)
                eng.step()
            gather.submit(outputs(grp))
        gather.finish()
        grp.sync()
        torch.cuda.synchronize()
        for t in range(STEPS):
            got = gather.result_of(t)
            if fence == "packed":          # the used prefixes travelled; expanded they are the step's rows byte for byte
                got = dict(got, observations=gather.expand("observations", t))
            for k in names:
                assert torch.equal(got[k], want[t][k]), f"{fence}: gathered {k} of step {t} are not step {t}'s rows"
        assert grp.poll_errors()[0] == 0
        with pytest.raises(IndexError):
            gather.result_of(STEPS)
        grp.close()
    finally:
        dist.destroy_process_group()


def test_output_fence_is_one_shot_and_clearable():
    """mgx_wait_before_outputs: consumed by the next writer of the output buffers; NULL clears it."""
    import torch
    from mettagrid_amd import presets
    from mettagrid_amd.compiler import compile_spec
    from mettagrid_amd.engine import BatchedMettaGrid
    from mettagrid_amd.mapgen import random_class_maps

    prog = compile_spec(presets.rung2_spec(), 32, 32)
    cms = random_class_maps(prog, 32, 32, {"wall": 40}, 16, range(4))
    eng = BatchedMettaGrid(prog, cms, np.arange(4, dtype=np.uint32))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        torch.cuda._sleep(50_000_000)      # ~20+ ms of device time the fenced observation kernel has to sit out
        ev = torch.cuda.Event()
        ev.record(side)
    eng.wait_before_outputs(ev)
    eng.step()
    eng.sync()
    assert ev.query()                      # the step finished, so the event it waited for had completed
    eng.step()                             # one-shot: nothing pending any more
    eng.sync()
    import ctypes as C
    assert eng.L.mgx_wait_before_outputs(eng.h, C.c_void_p(None)) == 0
    eng.close()
