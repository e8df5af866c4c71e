"""GPU: the `MettaGrid`-shaped mirror (one env) keeps the reference pybind surface — buffer sharing by reference,
shape/dtype validation errors, accessors (tests/test_buffers.py:90-461, tests/test_buffer_reuse.py:19-122 of the
reference)."""
import numpy as np
import pytest

from mettagrid_amd import presets
from mettagrid_amd.engine import MettaGrid

pytestmark = pytest.mark.gpu


def make():
    return MettaGrid(presets.rung1_spec(), presets.rung1_map().tolist(), 42)


def test_surface_and_zero_copy_buffers():
    c = make()
    assert (c.obs_width, c.obs_height, c.map_width, c.map_height, c.max_steps) == (13, 13, 16, 16, 0)
    assert c.object_type_names == ["agent", "wall"] and c.current_step == 0
    A, T = 4, 100
    obs = np.zeros((A, T, 3), np.uint8)
    term, trunc = np.ones(A, np.bool_), np.ones(A, np.bool_)
    rew = np.ones(A, np.float32)
    act, vact = np.zeros(A, np.int32), np.zeros(A, np.int32)
    c.set_buffers(obs, term, trunc, rew, act, vact)
    assert c.observations() is obs and c.actions() is act        # stored by reference
    assert not term.any() and not trunc.any() and (rew == 0).all()  # cleared by _init_buffers
    assert (obs[:, 0, 0] == 0xFE).all()                           # initial observations were written into OUR array
    before = obs.copy()
    act[:] = [c.prog.action_names.index("move_east")] * A
    c.step()
    assert c.current_step == 1 and not np.array_equal(before, obs)
    assert c.action_success() == [True] * 4 or isinstance(c.action_success()[0], bool)
    assert c.masks().shape == (A, 5) and c.masks().all()
    assert c.get_episode_rewards().shape == (A,)
    st = c.get_episode_stats()
    assert set(st) == {"game", "agent"} and len(st["agent"]) == A and st["game"]["objects.wall"] == 71.0
    assert c.get_game_stat("objects.wall") == 71.0 and c.get_game_stat("nope") is None
    assert c.get_agent_stat(0, "action.move.success") in (1.0, None) and c.get_agent_stat(99, "x") is None
    objs = c.grid_objects()
    assert len(objs) == 75 and len(c.grid_objects(ignore_types=["wall"])) == 4
    assert all(0 <= o["r"] < 4 for o in c.grid_objects(0, 4, 0, 16).values())


def test_set_buffers_validation_errors():
    c = make()
    A, T = 4, 100
    good = lambda: [np.zeros((A, T, 3), np.uint8), np.zeros(A, np.bool_), np.zeros(A, np.bool_), np.zeros(A, np.float32),  # noqa: E731
                    np.zeros(A, np.int32), np.zeros(A, np.int32)]
    b = good(); b[0] = np.zeros((A, T), np.uint8)
    with pytest.raises(RuntimeError, match="dimensions but expected 3"):
        c.set_buffers(*b)
    b = good(); b[0] = np.zeros((A + 1, T, 3), np.uint8)
    with pytest.raises(RuntimeError, match=r"expected \[4, \[something\], 3\]"):
        c.set_buffers(*b)
    b = good(); b[1] = np.zeros(A + 1, np.bool_)
    with pytest.raises(RuntimeError, match="terminals has the wrong shape"):
        c.set_buffers(*b)
    b = good(); b[3] = np.zeros(A, np.float64)
    with pytest.raises(TypeError):
        c.set_buffers(*b)
    b = good(); b[0] = np.zeros((A, T, 6), np.uint8)[:, :, ::2]
    with pytest.raises(TypeError):
        c.set_buffers(*b)
    b = good(); b[0] = np.zeros((A, T + 5, 3), np.uint8)
    with pytest.raises(RuntimeError):
        c.set_buffers(*b)


def test_token_overflow_raises_runtime_error():
    from mettagrid_amd import spec as S
    spec = presets.rung1_spec()
    spec.obs = S.ObsSpec(width=13, height=13, num_tokens=8)
    with pytest.raises(RuntimeError, match="token budget exceeded"):
        c = MettaGrid(spec, presets.rung1_map().tolist(), 1)
        c.step()
