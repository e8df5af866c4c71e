"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's policy-side token decode: sparse observation tokens ->
dense float32 box [agents, C, H, W] (/root/reference/python/src/mettagrid/envs/grid_obs_wrapper.py:57-95
``GridObsWrapper._convert``; per-feature scale from ``__init__`` :39-44).  Pinned by the boxes the reference produced for
the fixtures tests/golden/ref_*.npz.  Checker for mgx_decode_obs; never imported by the product."""
from __future__ import annotations

import numpy as np

PADDING, GLOBAL = 0xFF, 0xFE


def feature_scale(norms, size: int = 256) -> np.ndarray:
    """scale[id] = max(normalization, 1.0) for the features the config defines, 1.0 elsewhere (:39-44)."""
    scale = np.ones(max(size, len(norms)), dtype=np.float32)
    for i, z in enumerate(norms):
        scale[i] = max(float(z), 1.0)
    return scale


def decode(raw_obs: np.ndarray, num_features: int, height: int, width: int, scale: np.ndarray) -> np.ndarray:
    """Token rows u8 [N, T, 3] -> box f32 [N, C, H, W].  Tokens that fall on the same (feature, y, x) ADD, in token order
    (np.add.at), global tokens (location 0xFE) land on the centre cell, padding (0xFF) is skipped (:57-95)."""
    n, t = raw_obs.shape[0], raw_obs.shape[1]
    grid = np.zeros((n, num_features, height, width), dtype=np.float32)
    cy, cx = height // 2, width // 2
    for i in range(n):
        for k in range(t):
            coord, fid, val = int(raw_obs[i, k, 0]), int(raw_obs[i, k, 1]), np.float32(raw_obs[i, k, 2])
            if coord == PADDING:
                continue
            y, x = (cy, cx) if coord == GLOBAL else ((coord >> 4) & 0xF, coord & 0xF)
            if y >= height or x >= width or fid >= num_features:
                continue
            grid[i, fid, y, x] = np.float32(grid[i, fid, y, x] + np.float32(val / scale[fid]))
    return grid
