"""GPU: the token -> dense box decode kernel (mgx_decode_obs) against the reference's own boxes
(tests/golden/ref_*.npz, produced by GridObsWrapper._convert) and against the numpy restatement oracle/obs_decode.py on
adversarial token rows; bit-exact float32."""
import glob
import json
import os

import numpy as np
import pytest

import obs_decode
import ref_tree
from mettagrid_amd import from_reference as fr
from mettagrid_amd import presets
from mettagrid_amd.compiler import compile_spec
from mettagrid_amd.engine import BatchedMettaGrid
from mettagrid_amd.mapgen import random_class_maps

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FIX = sorted(glob.glob(os.path.join(HERE, "golden", "ref_*.json")))


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[4:-5] for p in FIX])
def test_decode_equals_reference_boxes(path):
    import torch
    doc = json.load(open(path))
    z = np.load(path[:-5] + ".npz")
    cells = np.asarray(doc["map"], dtype=object)
    prog = fr.compile_reference_config(ref_tree.load(doc["config"]), *cells.shape)
    eng = BatchedMettaGrid(prog, prog.class_map(cells)[None], [doc["seed"]], buffers="device")
    assert np.array_equal(eng.feature_scale()[:len(prog.feature_norms)], z["feature_scale"])
    for key, obs in (("box_first", z["obs"][0]), ("box_last", z["obs"][-1])):
        tok = torch.from_numpy(np.ascontiguousarray(obs)).cuda()
        box = eng.decode_obs(tokens=tok)
        eng.sync()
        assert tuple(box.shape[1:]) == tuple(int(x) for x in z["box_shape"])
        assert np.array_equal(box.cpu().numpy(), z[key]), (doc["scenario"], key)
    # the engine's own observation buffer right after construction = the reference's first observation
    box = eng.decode_obs()
    eng.sync()
    assert np.array_equal(box.cpu().numpy(), z["box_first"])


@pytest.mark.parametrize("tokens_per_row", [200, 37, 513])
def test_decode_adversarial_rows(tokens_per_row):
    """Many tokens on one cell (order-dependent float sums), out-of-window coordinates, feature ids past the table, global
    and padding tokens in the middle of a row, row lengths that are not a multiple of four / past 512 tokens."""
    import torch
    spec = presets.rung3_spec(obs_tokens=tokens_per_row)
    prog = compile_spec(spec, 32, 32, max_objects=192)
    cms = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, range(1))
    eng = BatchedMettaGrid(prog, cms, [0], buffers="device")
    C, H, W = len(prog.feature_norms), 11, 11
    rng = np.random.default_rng(tokens_per_row)
    rows = 48
    tok = np.full((rows, tokens_per_row, 3), 0xFF, np.uint8)
    for r in range(rows):
        n = rng.integers(0, tokens_per_row + 1)
        coords = rng.choice([0x55, 0x56, 0x00, 0xAA, 0xFE, 0xFF, 0xEE, 0x5B, 0xB5], size=n, p=[.3, .1, .1, .1, .1, .1, .05, .1, .05])
        tok[r, :n, 0] = coords
        tok[r, :n, 1] = rng.choice([0, 1, 5, 6, 12, C - 1, C, 200], size=n)
        tok[r, :n, 2] = rng.integers(0, 256, size=n)
    box = eng.decode_obs(tokens=torch.from_numpy(tok).cuda())
    eng.sync()
    want = obs_decode.decode(tok, C, H, W, obs_decode.feature_scale(prog.feature_norms))
    got = box.cpu().numpy()
    assert np.array_equal(got, want), np.argwhere(got != want)[:5]


def test_decode_of_live_observations_at_scale():
    import torch
    spec = presets.rung3_spec()
    prog = compile_spec(spec, 32, 32, max_objects=192)
    E = 4096
    cms = random_class_maps(prog, 32, 32, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, range(E))
    eng = BatchedMettaGrid(prog, cms, np.arange(E, dtype=np.uint32), buffers="device")
    A, n_act = prog.num_agents, len(prog.action_names)
    gen = torch.Generator(device="cuda").manual_seed(1)
    for _ in range(3):
        eng.actions.copy_(torch.randint(0, n_act, (E * A,), dtype=torch.int32, device="cuda", generator=gen))
        eng.vibe_actions.copy_(torch.randint(0, n_act, (E * A,), dtype=torch.int32, device="cuda", generator=gen))
        torch.cuda.synchronize()
        eng.step()
    box = eng.decode_obs()
    eng.sync()
    obs = eng.obs.cpu().numpy()
    C, H, W = box.shape[1:]
    sample = [0, 1, 17, A * 100 + 3, E * A // 2, E * A - 1]
    want = obs_decode.decode(obs[sample], C, H, W, obs_decode.feature_scale(prog.feature_norms))
    assert np.array_equal(box[sample].cpu().numpy(), want)
    # size-independent property over ALL rows: every value lands somewhere — the box total equals the sum of the
    # normalised token values (float64 reference, tolerance for the float32 accumulation order only across cells)
    scale = torch.from_numpy(eng.feature_scale()).cuda().double()
    o = eng.obs
    valid = o[..., 0] != 0xFF
    total = ((o[..., 2].double() / scale[o[..., 1].long()]) * valid).sum().item()
    assert abs(box.double().sum().item() - total) <= 1e-6 * max(1.0, total)
    assert torch.isfinite(box).all().item() and (box >= 0).all().item()


def _bf16_bits(x: np.ndarray) -> np.ndarray:
    """float32 -> bfloat16 bit patterns, round to nearest even (finite inputs)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


@pytest.mark.parametrize("name", ["rung3", "rung4", "lit", "crowd", "torture_base10", "rung4_full"])
def test_fused_box_output_equals_decode_of_the_token_path(name):
    """mgx_set_box_output (fused form): the observation kernel writes the dense box straight from its LDS
    staging rows.  Two engines on identical seeds and actions — one on the token path (whose rows go through the numpy
    restatement of GridObsWrapper._convert, oracle/obs_decode.py, and through the standalone decode kernel), one on the fused
    path — must give the same box after every step: float32 bit for bit, bfloat16 = the float32 box rounded to nearest even.
    Rewards, flags and the final state digests must not notice the output mode."""
    import torch

    import helpers as hp
    spec_f, map_f, steps, invalid = hp.SCENARIOS[name]
    steps = min(steps, 14)
    E = 5
    maps = [map_f(s) for s in range(E)]
    prog = hp.compile_scenario(name, spec_f(), *maps[0].shape)
    cms = np.stack([prog.class_map(m) for m in maps])
    seeds = np.arange(E, dtype=np.uint32) + 3
    A, C = prog.num_agents, len(prog.feature_norms)
    H, W = int(prog.words[13]), int(prog.words[14])      # MGX_H_OBS_HEIGHT / WIDTH
    tok_eng = BatchedMettaGrid(prog, cms, seeds, buffers="device")
    engs = {torch.float32: BatchedMettaGrid(prog, cms, seeds, buffers="device"), torch.bfloat16: BatchedMettaGrid(prog, cms, seeds, buffers="device")}
    boxes = {dt: torch.full((E * A, C, H, W), 7.0, dtype=dt, device="cuda") for dt in engs}
    for dt, eng in engs.items():
        eng.set_box_output(boxes[dt])
    scale = obs_decode.feature_scale(prog.feature_norms)
    acts = [hp.make_actions(prog, i, steps, invalid) for i in range(E)]
    for t in range(steps):
        a = torch.from_numpy(np.concatenate([acts[i][0][t] for i in range(E)])).cuda()
        v = torch.from_numpy(np.concatenate([acts[i][1][t] for i in range(E)])).cuda()
        for eng in [tok_eng, *engs.values()]:
            eng.actions.copy_(a)
            eng.vibe_actions.copy_(v)
        torch.cuda.synchronize()
        for eng in [tok_eng, *engs.values()]:
            eng.step()
            eng.sync()
        torch.cuda.synchronize()
        want = obs_decode.decode(tok_eng.obs.cpu().numpy(), C, H, W, scale)
        standalone = tok_eng.decode_obs()
        tok_eng.sync()
        assert np.array_equal(standalone.cpu().numpy(), want)
        assert np.array_equal(boxes[torch.float32].cpu().numpy(), want), f"{name} step {t + 1}: fused float32 box differs"
        got16 = boxes[torch.bfloat16].view(torch.int16).cpu().numpy().view(np.uint16)
        assert np.array_equal(got16, _bf16_bits(want)), f"{name} step {t + 1}: fused bfloat16 box differs"
        for eng in engs.values():
            assert torch.equal(eng.rewards, tok_eng.rewards) and torch.equal(eng.truncations, tok_eng.truncations)
    dig = tok_eng.state_digests()
    for eng in engs.values():
        assert np.array_equal(eng.state_digests(), dig)      # token statistics included
        assert eng.poll_errors()[0] == 0
    # back to token rows
    eng = engs[torch.float32]
    eng.set_box_output(None)
    eng.actions.zero_(); eng.vibe_actions.zero_(); tok_eng.actions.zero_(); tok_eng.vibe_actions.zero_()
    torch.cuda.synchronize()
    eng.step(); tok_eng.step(); eng.sync(); tok_eng.sync()
    assert torch.equal(eng.obs, tok_eng.obs)
    with pytest.raises(ValueError):
        eng.set_box_output(torch.zeros((1, C, H, W), device="cuda"))
