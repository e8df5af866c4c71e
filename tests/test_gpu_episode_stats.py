"""GPU: episode-end statistics of auto-resetting envs — the `infos` of MettaGridPufferEnv.step
(python/src/mettagrid/envs/stats_tracker.py:26-76, mettagrid_puffer_env.py:230-283) reduced on the device before the
restart wipes them (include/mgx.h "Episode-end statistics").  Every finished episode is compared with an oracle env that is
rebuilt on the host exactly when the reference wrapper would build a new Simulation; the batch totals are compared bit for
bit with the same sums taken on the host in the engine's documented order."""
import math

import numpy as np
import pytest

import oracle_py as op
from mettagrid_amd import presets
from mettagrid_amd.compiler import compile_spec
from mettagrid_amd.engine import BatchedMettaGrid
from mettagrid_amd.envs import MettaGridBatchedEnv
from mettagrid_amd.signature import stats_dicts

pytestmark = pytest.mark.gpu


from helpers import reference_infos  # noqa: E402  (pinned to the reference's StatsTracker by tests/test_infos_fixture.py)


class HostTotals:
    """The engine's accumulation order restated: per step the finished envs in ascending env index, in chunks of 256 whose
    sums (started from 0.0) are added to the totals in chunk order; f64 throughout (csrc/mgx_episode.h)."""

    def __init__(self, prog):
        self.prog = prog
        self.gn, self.an = list(prog.game_stat_names), list(prog.agent_stat_names)
        self.clear()

    def clear(self):
        self.n = 0
        self.rs, self.rmin, self.rmax = 0.0, math.inf, -math.inf
        self.ls, self.lmin, self.lmax = 0.0, math.inf, -math.inf
        self.term = 0
        self.gs = {k: 0.0 for k in self.gn}; self.gc = {k: 0 for k in self.gn}
        self.as_ = {k: 0.0 for k in self.an}; self.ac = {k: 0 for k in self.an}

    def add_step(self, episodes: list):
        """episodes: [(reference infos, all_terminal)] of this step's finished envs in ascending env index."""
        A = self.prog.num_agents
        for c0 in range(0, len(episodes), 256):
            chunk = episodes[c0:c0 + 256]
            gs = {k: 0.0 for k in self.gn}; as_ = {k: 0.0 for k in self.an}
            rs = ls = 0.0
            for info, all_term in chunk:
                for k in self.gn:
                    gs[k] += info["game"].get(k, 0.0)
                    self.gc[k] += k in info["game"]
                for k in self.an:
                    as_[k] += info["agent"].get(k, 0.0)
                    self.ac[k] += k in info["agent"]
                ret = 0.0
                for r in info["episode_rewards"]:
                    ret += float(r)
                ret /= A
                rs += ret; ls += float(info["steps"])
                self.rmin, self.rmax = min(self.rmin, ret), max(self.rmax, ret)
                self.lmin, self.lmax = min(self.lmin, float(info["steps"])), max(self.lmax, float(info["steps"]))
                self.term += bool(all_term)
            for k in self.gn:
                self.gs[k] += gs[k]
            for k in self.an:
                self.as_[k] += as_[k]
            self.rs += rs; self.ls += ls
            self.n += len(chunk)

    def as_dict(self) -> dict:
        return {"episodes": self.n, "return_sum": self.rs, "return_min": self.rmin, "return_max": self.rmax,
                "length_sum": self.ls, "length_min": self.lmin, "length_max": self.lmax, "terminated": float(self.term),
                "game_sum": {k: v for k, v in self.gs.items() if self.gc[k]}, "game_count": {k: v for k, v in self.gc.items() if v},
                "agent_sum": {k: v for k, v in self.as_.items() if self.ac[k]}, "agent_count": {k: v for k, v in self.ac.items() if v}}


def _same(a, b, where):
    """Dict equality with bit-exact floats (== would let -0.0 pass for 0.0 and fail on nan)."""
    assert a.keys() == b.keys(), (where, sorted(set(a) ^ set(b)))
    for k in a:
        x, y = a[k], b[k]
        if isinstance(x, dict):
            _same(x, y, f"{where}/{k}")
        elif isinstance(x, float) or isinstance(y, float):
            assert np.float64(x).tobytes() == np.float64(y).tobytes(), (where, k, x, y)
        else:
            assert x == y, (where, k, x, y)


def _engine_run(workload: str, E: int, steps: int, max_steps: int, invalid_share: float = 0.0, drain_every: int = 37):
    M, stride = 7, 3
    if workload == "rung4":
        spec = presets.rung4_spec(max_steps=max_steps)
        prog = compile_spec(spec, 64, 64, max_objects=presets.RUNG4_MAX_OBJECTS)
        pool = np.stack([prog.class_map(presets.rung4_map(50 + m)) for m in range(M)])
    else:
        spec = presets.rung3_spec()
        spec.max_steps = max_steps
        spec.episode_truncates = True
        prog = compile_spec(spec, 32, 32, max_objects=192)
        pool = np.stack([prog.class_map(presets.rung3_map(50 + m)) for m in range(M)])
    A, n_act = prog.num_agents, len(prog.action_names)
    seeds = (11 + np.arange(E)).astype(np.uint32)
    from mettagrid_amd.early_reset import first_integers
    early = first_integers(seeds, max_steps)
    eng = BatchedMettaGrid(prog, pool[np.arange(E) % M], seeds, buffers="host")
    eng.set_map_pool(pool)
    eng.set_auto_reset(True, stride, early)
    eng.set_episode_stats(True, log_capacity=4 * E, log_per_agent=True)
    oracles = [op.OracleSim(prog, pool[e % M], int(seeds[e])) for e in range(E)]
    for o in oracles:
        o.reinit_buffers()
    episode, ended = [0] * E, [False] * E
    rng = np.random.default_rng(5)
    want = HostTotals(prog)
    expected_log = []
    n_eps = 0
    for t in range(steps):
        for e in range(E):
            if ended[e]:
                episode[e] += 1
                oracles[e] = op.OracleSim(prog, pool[(e + episode[e] * stride) % M], int(seeds[e]))
                oracles[e].reinit_buffers()
                ended[e] = False
        a = rng.integers(0, n_act, E * A).astype(np.int32)
        v = rng.integers(0, n_act, E * A).astype(np.int32)
        if invalid_share:   # a few indices far outside the action table: "action.invalid_index.<k>" keys without a stat column
            bad = rng.random(E * A) < invalid_share
            a[bad] = rng.choice([n_act + 40, -25, n_act + 77], size=int(bad.sum()))
        eng.actions[:] = a
        eng.vibe_actions[:] = v
        eng.step()
        finished = []
        for e, o in enumerate(oracles):
            o.step(a[e * A:(e + 1) * A], v[e * A:(e + 1) * A])
            s = o.snapshot()
            early_end = episode[e] == 0 and o.current_step >= early[e]
            all_term, all_trunc = bool(s["terminals"].all()), bool(s["truncations"].all()) or early_end
            ended[e] = all_term or all_trunc
            if ended[e]:
                info = reference_infos(prog, o)
                info.update(env=e, episode=episode[e], map_index=(e + episode[e] * stride) % M if episode[e] else -1, seed=int(seeds[e]),
                            flags=(1 if all_term else 0) | (2 if all_trunc else 0) | (4 if early_end else 0))
                finished.append((info, all_term))
        want.add_step(finished)
        expected_log += [f[0] for f in finished]
        n_eps += len(finished)
        if (t + 1) % drain_every == 0 or t == steps - 1:
            got = eng.drain_episode_stats()
            _same(got, want.as_dict(), f"{workload} totals at step {t}")
            want.clear()
            recs, dropped = eng.drain_episode_log()
            assert dropped == 0 and len(recs) == len(expected_log), (t, dropped, len(recs), len(expected_log))
            for r, x in zip(recs, expected_log):
                where = f"{workload} env {x['env']} episode {x['episode']}"
                for k in ("env", "episode", "map_index", "seed", "steps", "flags"):
                    assert r[k] == x[k], (where, k, r[k], x[k])
                assert np.array_equal(r["episode_rewards"], x["episode_rewards"]), where
                _same(r["game"], x["game"], where + " game")
                _same(r["agent"], x["agent"], where + " agent")
                _same({str(i): dct for i, dct in enumerate(r["per_agent"])}, x["per_agent"], where + " per_agent")
            expected_log = []
    assert eng.poll_errors()[0] & ~2 == 0
    return n_eps


def test_episode_stats_against_oracle_rung3():
    """96 envs of the rung-3 rules, 11-step episodes, map pool + desync, 320 steps: ~2 700 finished episodes, every one and
    every total compared."""
    n = _engine_run("rung3", E=96, steps=320, max_steps=11)
    assert n >= 2000


def test_episode_stats_invalid_index_keys():
    n = _engine_run("rung3", E=12, steps=60, max_steps=9, invalid_share=0.02, drain_every=20)
    assert n >= 50


def test_episode_stats_extended_game():
    n = _engine_run("rung4", E=10, steps=30, max_steps=7, drain_every=11)
    assert n >= 30


def test_chunked_totals_when_every_env_finishes_at_once():
    """600 envs without desync all finish on the same step: three chunks (256 + 256 + 88) in the accumulation order."""
    spec = presets.rung2_spec()
    spec.max_steps = 6
    spec.episode_truncates = True
    prog = compile_spec(spec, 32, 32, max_objects=192)
    E, A = 600, prog.num_agents
    pool = np.stack([prog.class_map(presets.rung2_map(50 + m)) for m in range(5)])
    seeds = (3 + np.arange(E)).astype(np.uint32)
    eng = BatchedMettaGrid(prog, pool[np.arange(E) % 5], seeds, buffers="host")
    eng.set_map_pool(pool)
    eng.set_auto_reset(True, 1, None)
    eng.set_episode_stats(True, log_capacity=100)   # smaller than one step's episodes: the rest is counted as dropped
    oracles = [op.OracleSim(prog, pool[e % 5], int(seeds[e])) for e in range(E)]
    for o in oracles:
        o.reinit_buffers()
    rng = np.random.default_rng(1)
    n_act = len(prog.action_names)
    want = HostTotals(prog)
    for t in range(6):
        a = rng.integers(0, n_act, E * A).astype(np.int32)
        eng.actions[:] = a
        eng.vibe_actions[:] = 0
        eng.step()
        for e, o in enumerate(oracles):
            o.step(a[e * A:(e + 1) * A], np.zeros(A, np.int32))
    assert all(o.snapshot()["truncations"].all() for o in oracles)
    want.add_step([(reference_infos(prog, o), False) for o in oracles])
    _same(eng.drain_episode_stats(), want.as_dict(), "three chunks")
    recs, dropped = eng.drain_episode_log()
    assert len(recs) == 100 and dropped == 500 and [r["env"] for r in recs] == list(range(100))
    assert eng.drain_episode_stats()["episodes"] == 0


def test_batched_env_returns_infos():
    """MettaGridBatchedEnv.step returns the aggregate of the finished episodes (no host synchronisation: the snapshot
    requested at one step is returned by a later one); summed over the run it accounts for every finished episode, and the
    per-episode infos of episode_infos() have the reference's shape."""
    import torch
    spec = presets.rung3_spec()
    spec.max_steps = 8
    spec.episode_truncates = True
    prog = compile_spec(spec, 32, 32, max_objects=192)
    E, A = 64, prog.num_agents
    pool = np.stack([prog.class_map(presets.rung3_map(50 + m)) for m in range(4)])
    env = MettaGridBatchedEnv(prog, E, map_pool=pool, desync=True, seed=3, validate_actions=False, stats_interval=2, episode_log=4096,
                              log_per_agent=True)
    obs, infos = env.reset()
    assert infos == {}
    n = env.transport_action_n
    g = torch.Generator(device="cuda").manual_seed(0)
    total, seen = 0, []
    for t in range(80):
        a = torch.randint(0, n, (E * A,), device="cuda", dtype=torch.int32, generator=g)
        *_, rew, _, _, infos = env.step(a)
        float(rew.sum())   # a consumer reads the step's results (here: a host read, as a policy's action sampling ends in one)
        if infos:
            total += infos["episodes"]
            seen.append(infos)
            assert set(infos) >= {"episodes", "game", "agent", "episode_return", "episode_length", "attributes"}
            assert infos["episode_length"]["max"] <= 8 and infos["episode_length"]["min"] >= 1
            assert infos["attributes"]["max_steps"] == 8 and infos["attributes"]["map_w"] == 32
    env.engine.sync()
    last = env.engine.drain_episode_stats()
    total += last["episodes"]
    ep, _ = env.engine.episodes()
    done_now = int((env.engine.terminals.view(E, A).all(1) | env.engine.truncations.view(E, A).all(1)).sum())
    assert total == int(ep.sum()) + done_now and total > 400 and len(seen) >= 10
    per_episode = env.episode_infos()
    assert len(per_episode) == total and env.episodes_dropped == 0
    one = per_episode[0]
    assert set(one) >= {"game", "agent", "per_agent", "attributes", "episode_rewards"} and len(one["per_agent"]) == A
    assert 1 <= one["attributes"]["steps"] <= 8 and one["agent"] and one["game"]
    # the aggregate of a window is the mean of its episodes' dicts
    k = "tokens_written"
    if k in seen[0]["game"]:
        first = per_episode[:seen[0]["episodes"]]
        assert seen[0]["game"][k] == sum(x["game"][k] for x in first) / len(first)
    env.close()


def test_host_driven_restart_records_episodes():
    """map_fn mode (maps built on the host per episode): the wrapper records finished envs before it restarts them."""
    spec = presets.rung2_spec()
    spec.max_steps = 5
    spec.episode_truncates = True
    prog = compile_spec(spec, 32, 32)
    E, A = 3, prog.num_agents
    env = MettaGridBatchedEnv(prog, E, map_fn=lambda e, ep: prog.class_map(presets.rung2_map(100 * ep + e)), seed=7, buffers="host")
    env.reset()
    got = []
    for t in range(12):
        *_, infos = env.step(np.zeros(E * A, np.int32))
        if infos:
            got.append((t, infos["episodes"], infos["episode_length"]["mean"]))
    assert got == [(5, 3, 5.0), (10, 3, 5.0)]
    env.close()
