"""Action decoding rules of the PufferEnv surface (reference: python/src/mettagrid/envs/mettagrid_puffer_env.py:305-394,
pinned there by tests/test_mettagrid_puffer_env.py) — host logic, no GPU."""
import numpy as np
import pytest

from mettagrid_amd.envs import decode_actions, split_action_names

NAMES = ["noop", "move_north", "move_south", "change_vibe_default", "change_vibe_a"]


def test_split_names():
    assert split_action_names(NAMES) == (["noop", "move_north", "move_south"], ["change_vibe_default", "change_vibe_a"])


def test_combined_index_encoding():
    vibe_ids = np.array([3, 4], np.int64)
    #           plain  plain  enc: primary 0 + vibe 0   enc: primary 2 + vibe 1
    a = np.array([0, 2, 3 + 0 * 2 + 0, 3 + 2 * 2 + 1], np.int32)
    core, vibe = decode_actions(a, 3, vibe_ids)
    assert core.tolist() == [0, 2, 0, 2] and vibe.tolist() == [0, 0, 3, 4]
    core, vibe = decode_actions(np.array([1, 0], np.int32), 3, vibe_ids)
    assert core.tolist() == [1, 0] and vibe is None
    with pytest.raises(ValueError, match="out of range"):
        decode_actions(np.array([3 + 3 * 2], np.int32), 3, vibe_ids)
    with pytest.raises(ValueError, match="non-negative"):
        decode_actions(np.array([-1], np.int32), 3, vibe_ids)
    with pytest.raises(ValueError, match="no configured vibe action space"):
        decode_actions(np.array([4], np.int32), 3, np.zeros(0, np.int64))


def test_two_column_encoding():
    vibe_ids = np.array([3, 4], np.int64)
    core, vibe = decode_actions(np.array([[1, 0], [2, 1]], np.int64), 3, vibe_ids)
    assert core.tolist() == [1, 2] and vibe.tolist() == [3, 4]
    core, vibe = decode_actions(np.array([[1], [2]], np.int32), 3, vibe_ids)
    assert core.tolist() == [1, 2] and vibe is None
    with pytest.raises(ValueError, match="Vibe action indices out of range"):
        decode_actions(np.array([[1, 2]], np.int32), 3, vibe_ids)
    with pytest.raises(ValueError, match="Core actions out of range"):
        decode_actions(np.array([[5, 0]], np.int32), 3, vibe_ids)
    with pytest.raises(ValueError, match="Expected step actions shape"):
        decode_actions(np.zeros((2, 3), np.int32), 3, vibe_ids)
