"""Env groups (mettagrid_amd/groups.py): a batch stepped as two chained engines on two streams must produce exactly what
one engine produces for the same envs — and both must equal the oracle (tests/helpers.py scenarios do that part)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(E):
    from mettagrid_amd import presets
    from mettagrid_amd.compiler import compile_spec
    from mettagrid_amd.mapgen import random_class_maps
    prog = compile_spec(presets.rung3_spec(), 32, 32, max_objects=192)
    cms = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, range(E))
    return prog, cms, np.arange(E, dtype=np.uint32)


@pytest.mark.parametrize("groups", [2, 4])
def test_groups_match_single_engine(groups):
    import torch
    from mettagrid_amd.engine import BatchedMettaGrid
    from mettagrid_amd.groups import EnvGroups
    E, steps = 8, 25
    prog, cms, seeds = _setup(E)
    one = BatchedMettaGrid(prog, cms, seeds, buffers="device")
    grp = EnvGroups(prog, cms, seeds, groups=groups)
    n = len(prog.action_names)
    a0, b0 = one.snapshot(), grp.snapshot()
    for k in a0:
        assert np.array_equal(a0[k], b0[k]), k
    gen = torch.Generator(device="cuda").manual_seed(7)
    for t in range(steps):   # no host synchronisation between steps: the groups overlap across steps
        a = torch.randint(0, n, (E * prog.num_agents,), dtype=torch.int32, device="cuda", generator=gen)
        v = torch.randint(0, n, (E * prog.num_agents,), dtype=torch.int32, device="cuda", generator=gen)
        for tgt in (one, grp):
            tgt.actions.copy_(a)
            tgt.vibe_actions.copy_(v)
            tgt.wait_for_caller()
            tgt.step()
            tgt.caller_waits()    # the next copy_ into the action buffers may not overtake this step
    a1, b1 = one.snapshot(), grp.snapshot()
    for k in a1:
        assert np.array_equal(a1[k], b1[k]), k
    assert grp.poll_errors() == (0, -1)
    grp.close()
    one.close()


def test_groups_reject_uneven_split():
    from mettagrid_amd.groups import EnvGroups
    prog, cms, seeds = _setup(3)
    with pytest.raises(ValueError):
        EnvGroups(prog, cms, seeds, groups=2)
