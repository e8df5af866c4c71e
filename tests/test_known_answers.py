"""Known-answer tests restated from the reference's own behavioural tests (tiny maps, a handful of steps), run
against the CPU oracle always and against the HIP engine when a GPU is present.

Sources (under /root/reference/tests): test_move.py:127-671, test_actions.py, test_observations.py:93-330,
test_observation_token_budget.py:8, test_chest.py:10-64, test_buffers.py:90-461, test_agent_id_obs.py:62-102,
test_local_position_obs.py:85-277, test_last_action_move_observation.py:8, test_rewards.py:26-120.
"""
import numpy as np
import pytest

import oracle_py as op
from mettagrid_amd import spec as S
from mettagrid_amd.compiler import compile_spec
from mettagrid_amd.mapgen import ascii_map

BACKENDS = ["oracle", pytest.param("hip", marks=pytest.mark.gpu)]
CHARS = {"#": "wall", ".": "empty", "@": "agent.agent", "1": "agent.red", "2": "agent.blue", "C": "chest", "E": "extractor"}
A, T = S.ACTOR, S.TARGET


class Sim:
    """Uniform view of one env on either backend."""

    def __init__(self, backend, spec, lines, seed=0):
        cells = ascii_map(lines, CHARS)
        self.prog = compile_spec(spec, *cells.shape)
        self.backend = backend
        cm = self.prog.class_map(cells)
        if backend == "oracle":
            self.o = op.OracleSim(self.prog, cm, seed)
        else:
            from mettagrid_amd.engine import BatchedMettaGrid
            self.e = BatchedMettaGrid(self.prog, cm[None], [seed], buffers="host")
        self.n = self.prog.num_agents

    def act(self, *names, vibe=None):
        idx = [self.prog.action_names.index(n) for n in names]
        a = np.array(idx + [0] * (self.n - len(idx)), np.int32)
        v = np.zeros(self.n, np.int32) if vibe is None else np.array([self.prog.action_names.index(x) for x in vibe], np.int32)
        if self.backend == "oracle":
            self.o.step(a, v)
        else:
            self.e.actions[:] = a
            self.e.vibe_actions[:] = v
            self.e.step()

    def snap(self):
        return self.o.snapshot() if self.backend == "oracle" else self.e.snapshot()

    def error(self):
        return self.o.error if self.backend == "oracle" else self.e.poll_errors()[0]

    def objects(self):
        from mettagrid_amd.signature import objects_from_raw
        raw = self.o.raw_objects() if self.backend == "oracle" else self.e.raw_objects(0)
        return objects_from_raw(self.prog, raw)

    def stats(self):
        from mettagrid_amd.signature import stats_dicts
        raw = self.o.raw_stats() if self.backend == "oracle" else self.e.raw_stats(0)
        extra = self.o.invalid_index_extra() if self.backend == "oracle" else self.e.invalid_index_extra(0)
        return stats_dicts(self.prog, *raw, extra=extra)

    def agent_pos(self, i=0):
        for o in self.objects().values():
            if o.get("agent_id") == i:
                return (o["r"], o["c"])

    def tokens(self, agent=0):
        obs = self.snap()["obs"][agent]
        return [tuple(int(x) for x in t) for t in obs if t[0] != 0xFF]

    def feature(self, name):
        return self.prog.feature_ids[name]


def basic_spec(**kw):
    base = dict(resource_names=["ore", "hp"], agents=[S.AgentSpec()],
                objects={"wall": S.ObjectSpec("wall", kind="wall")},
                move_directions=["north", "south", "west", "east", "northwest", "northeast", "southwest", "southeast"],
                obs=S.ObsSpec(width=3, height=3, num_tokens=50), max_steps=10)
    base.update(kw)
    return S.GameSpec(**base)


ROOM = ["#####", "#...#", "#.@.#", "#...#", "#####"]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("direction,delta", [("north", (-1, 0)), ("east", (0, 1)), ("south", (1, 0)), ("west", (0, -1)),
                                             ("northeast", (-1, 1)), ("northwest", (-1, -1)), ("southeast", (1, 1)),
                                             ("southwest", (1, -1))])
def test_8way_movement_all_directions(backend, direction, delta):
    sim = Sim(backend, basic_spec(), ROOM)
    before = sim.agent_pos()
    sim.act(f"move_{direction}")
    after = sim.agent_pos()
    assert bool(sim.snap()["action_success"][0])
    assert (after[0] - before[0], after[1] - before[1]) == delta


@pytest.mark.parametrize("backend", BACKENDS)
def test_move_blocked_by_wall_and_boundary(backend):
    sim = Sim(backend, basic_spec(), ["###", "#@#", "###"])
    sim.act("move_north")
    assert not sim.snap()["action_success"][0] and sim.agent_pos() == (1, 1)
    st = sim.stats()["agent"][0]
    assert st["action.move.failed"] == 1.0 and st["action.failed"] == 1.0 and "action.move.success" not in st
    edge = Sim(backend, basic_spec(), ["@..", "...", "..."])  # no border: moving off-grid fails (move.hpp:89-93)
    edge.act("move_north")
    assert not edge.snap()["action_success"][0] and edge.agent_pos() == (0, 0)
    edge.act("move_west")
    assert not edge.snap()["action_success"][0]
    edge.act("move_southeast")
    assert edge.snap()["action_success"][0] and edge.agent_pos() == (1, 1)


@pytest.mark.parametrize("backend", BACKENDS)
def test_agents_block_each_other_and_steps_without_motion(backend):
    spec = basic_spec(agents=[S.AgentSpec(), S.AgentSpec()])
    sim = Sim(backend, spec, ["#####", "#@@.#", "#####"])
    sim.act("move_east", "move_east")  # one of them moves first (shuffled order); both end up adjacent, no overlap
    pos = {sim.agent_pos(0), sim.agent_pos(1)}
    assert len(pos) == 2 and all(p[0] == 1 for p in pos)
    for _ in range(3):
        sim.act("noop", "noop")
    assert sim.stats()["agent"][0]["status.max_steps_without_motion"] >= 3.0


@pytest.mark.parametrize("backend", BACKENDS)
def test_observation_token_layout_and_order(backend):
    """tests/test_observations.py:93-231 / test_agent_id_obs.py: global tokens first at 0xFE, then cells in Manhattan
    order; per object tags ascending, vibe, inventory, agent:group, agent_id."""
    spec = basic_spec(agents=[S.AgentSpec(tags=["team:x"], inventory=S.Inventory(initial={"hp": 300}))],
                      vibe_names=["default", "happy"])
    sim = Sim(backend, spec, ROOM)
    toks = sim.tokens()
    f = sim.feature
    assert toks[0] == (0xFE, f("episode_completion_pct"), 0)
    assert toks[1] == (0xFE, f("last_action"), 0)
    assert toks[2] == (0xFE, f("last_reward"), 0)
    centre = (1 << 4) | 1
    tag_ids = sorted(sim.prog.tag_names.index(t) for t in ("team:x", "type:agent"))
    assert toks[3:5] == [(centre, f("tag"), tag_ids[0]), (centre, f("tag"), tag_ids[1])]
    assert toks[5] == (centre, f("inv:hp"), 300 % 256) and toks[6] == (centre, f("inv:hp:p1"), 1)
    assert toks[7] == (centre, f("agent:group"), 0) and toks[8] == (centre, f("agent_id"), 0)
    assert len(toks) == 9  # 3x3 window -> 5 cells (centre + 4 neighbours), all neighbours empty
    sim.act("noop", vibe=["change_vibe_happy"])
    toks = sim.tokens()
    assert (centre, f("vibe"), 1) in toks and toks.index((centre, f("vibe"), 1)) == 5
    assert toks[1] == (0xFE, f("last_action"), sim.prog.action_names.index("change_vibe_happy"))


@pytest.mark.parametrize("backend", BACKENDS)
def test_episode_completion_pct_and_truncation(backend):
    """tests/test_observations.py:295,314: 25 at 1/10 and 51 at 2/10; tests/test_buffers.py truncation vs terminal."""
    sim = Sim(backend, basic_spec(max_steps=10, episode_truncates=True), ROOM)
    pct = sim.feature("episode_completion_pct")
    sim.act("noop")
    assert (0xFE, pct, 25) in sim.tokens()
    sim.act("move_south")
    assert (0xFE, pct, 51) in sim.tokens()
    for _ in range(7):
        sim.act("noop")
    s = sim.snap()
    assert not s["truncations"][0] and not s["terminals"][0]
    sim.act("noop")
    s = sim.snap()
    assert s["truncations"][0] and not s["terminals"][0] and (0xFE, pct, 255) in sim.tokens()
    term = Sim(backend, basic_spec(max_steps=2, episode_truncates=False), ROOM)
    term.act("noop")
    term.act("noop")
    s = term.snap()
    assert s["terminals"][0] and not s["truncations"][0]


@pytest.mark.parametrize("backend", BACKENDS)
def test_observation_token_budget_overflow_is_an_error(backend):
    """tests/test_observation_token_budget.py:8 — exceeding num_observation_tokens is an error, not truncation."""
    spec = basic_spec(obs=S.ObsSpec(width=3, height=3, num_tokens=6))
    sim = Sim(backend, spec, ["###", "#@#", "###"])  # 3 globals + agent (3 tokens) + 4 walls > 6
    assert sim.error() & 1
    ok = Sim(backend, basic_spec(obs=S.ObsSpec(width=3, height=3, num_tokens=10)), ["###", "#@#", "###"])
    assert ok.error() == 0 and len(ok.tokens()) == 10


@pytest.mark.parametrize("backend", BACKENDS)
def test_local_position_and_last_action_move_tokens(backend):
    """tests/test_local_position_obs.py:85-277, test_last_action_move_observation.py:8."""
    spec = basic_spec(obs=S.ObsSpec(width=3, height=3, num_tokens=50, local_position=True, last_action_move=True))
    sim = Sim(backend, spec, ["#######", "#.....#", "#..@..#", "#.....#", "#######"])
    f = sim.feature
    names = {f("lp:east"), f("lp:west"), f("lp:north"), f("lp:south")}
    assert not [t for t in sim.tokens() if t[1] in names]           # at spawn: no lp tokens
    sim.act("move_east")
    assert (0xFE, f("lp:east"), 1) in sim.tokens() and (0xFE, f("last_action_move"), 1) in sim.tokens()
    sim.act("move_north")
    toks = sim.tokens()
    assert (0xFE, f("lp:east"), 1) in toks and (0xFE, f("lp:north"), 1) in toks
    sim.act("move_north")                                            # wall: failed move
    assert (0xFE, f("last_action_move"), 0) in sim.tokens() and (0xFE, f("last_action"), 0) in sim.tokens()
    sim.act("move_west")
    sim.act("move_west")
    assert (0xFE, f("lp:west"), 1) in sim.tokens()


@pytest.mark.parametrize("backend", BACKENDS)
def test_chest_deposit_and_extractor_withdraw(backend):
    """tests/test_chest.py:17-64 patterns: bump = use; deposit all / withdraw with limits."""
    spec = basic_spec(
        agents=[S.AgentSpec(inventory=S.Inventory(initial={"ore": 7}, limits=[S.Limit(["ore"], base=10)]))],
        objects={"wall": S.ObjectSpec("wall", kind="wall"),
                 "chest": S.ObjectSpec("chest", inventory=S.Inventory(), on_use=S.Handler(
                     [], [S.ResourceTransfer(A, T, "ore", -1)], "deposit")),
                 "extractor": S.ObjectSpec("extractor", inventory=S.Inventory(initial={"ore": 50}), on_use=S.Handler(
                     [], [S.ResourceTransfer(T, A, "ore", 8)], "withdraw"))})
    sim = Sim(backend, spec, ["#####", "#C@E#", "#####"])
    inv = lambda name: next(o["inventory"] for o in sim.objects().values() if o["type_name"] == name)  # noqa: E731
    sim.act("move_west")
    assert sim.snap()["action_success"][0] and inv("chest") == {0: 7} and inv("agent") == {}
    assert sim.stats()["agent"][0]["ore.deposited"] == 7.0 and sim.agent_pos() == (1, 2)
    sim.act("move_east")
    assert inv("agent") == {0: 8} and inv("extractor") == {0: 42}
    sim.act("move_east")                     # limit 10: only 2 more fit
    assert inv("agent") == {0: 10} and inv("extractor") == {0: 40}
    st = sim.stats()["agent"][0]
    assert st["ore.gained"] == 10.0 and st["ore.amount"] == 10.0 and st["ore.lost"] == 7.0


@pytest.mark.parametrize("backend", BACKENDS)
def test_attack_mutation_damage_formula_and_death(backend):
    """tests/test_handler.cpp:500-532: damage = max(0, weapon * pct / 100 - armor); hp hitting 0 emits `death`."""
    res = ["hp", "laser", "armor"]
    att = S.Handler([], [S.Attack("laser", "armor", "hp", 150)], "hit")
    mk = lambda team, inv: S.AgentSpec(team_id=team, inventory=S.Inventory(initial=inv), on_use=att)  # noqa: E731
    spec = basic_spec(resource_names=res, agents=[mk(0, {"hp": 9, "laser": 4, "armor": 1}), mk(1, {"hp": 9, "laser": 1, "armor": 5})])
    sim = Sim(backend, spec, ["####", "#12#", "####"])
    hp_of = lambda i: next(o["inventory"].get(0, 0) for o in sim.objects().values() if o.get("agent_id") == i)  # noqa: E731
    sim.act("move_east", "noop")             # 4*150/100 - 5 = 1
    assert hp_of(1) == 8
    sim.act("noop", "move_west")             # 1*150/100 - 1 = 0 -> no damage
    assert hp_of(0) == 9
    for _ in range(8):
        sim.act("move_east", "noop")
    assert hp_of(1) == 0 and sim.stats()["agent"][1]["death"] == 1.0


@pytest.mark.parametrize("backend", BACKENDS)
def test_rewards_inventory_delta_and_per_tick(backend):
    """tests/test_rewards.py: non-accumulating entries pay the change of the value, per_tick entries pay it every step."""
    spec = basic_spec(
        agents=[S.AgentSpec(inventory=S.Inventory(initial={"hp": 4}),
                            rewards=[S.RewardSpec(S.InventoryValue("ore")), S.RewardSpec(S.SumValue([S.InventoryValue("hp")], [0.5]), per_tick=True)])],
        objects={"wall": S.ObjectSpec("wall", kind="wall"),
                 "extractor": S.ObjectSpec("extractor", inventory=S.Inventory(initial={"ore": 50}), on_use=S.Handler(
                     [], [S.ResourceTransfer(T, A, "ore", 3)], "w"))})
    sim = Sim(backend, spec, ["####", "#@E#", "####"])
    sim.act("noop")
    assert sim.snap()["rewards"][0] == np.float32(2.0)
    sim.act("move_east")
    assert sim.snap()["rewards"][0] == np.float32(5.0)
    sim.act("noop")
    s = sim.snap()
    assert s["rewards"][0] == np.float32(2.0) and s["episode_rewards"][0] == np.float32(9.0)


@pytest.mark.parametrize("backend", BACKENDS)
def test_invalid_action_index_counts_once_per_priority_level(backend):
    """mettagrid_c.cpp:914-919,966-973: the check runs in both priority passes of the stream."""
    sim = Sim(backend, basic_spec(), ROOM)
    n = len(sim.prog.action_names)
    if backend == "oracle":
        sim.o.step(np.array([n + 3], np.int32), np.array([0], np.int32))
    else:
        sim.e.actions[:] = n + 3
        sim.e.vibe_actions[:] = 0
        sim.e.step()
    st = sim.stats()["agent"][0]
    assert st["action.invalid_index"] == 2.0 and st[f"action.invalid_index.{n + 3}"] == 2.0
    assert not sim.snap()["action_success"][0]


@pytest.mark.parametrize("backend", BACKENDS)
def test_invalid_action_index_far_outside_the_action_table(backend):
    """mettagrid_c.cpp:916-918 creates one "action.invalid_index.<k>" key per distinct k, whatever k is.  Indices next
    to the action table have fixed stat columns; any other k takes one of the agent's MGX_INVALID_EXTRA (k, count)
    pairs, and only a fifth distinct far k in one episode is refused (env error bit 2)."""
    sim = Sim(backend, basic_spec(), ROOM)
    n = len(sim.prog.action_names)

    def step(a, v=0):
        if backend == "oracle":
            sim.o.step(np.array([a], np.int32), np.array([v], np.int32))
        else:
            sim.e.actions[:] = a
            sim.e.vibe_actions[:] = v
            sim.e.step()

    for a in (1000, -500, 1000, n + 40):
        step(a)
    step(0, 2 ** 31 - 1)      # the vibe stream is checked the same way
    st = sim.stats()["agent"][0]
    assert st["action.invalid_index"] == 10.0
    assert st["action.invalid_index.1000"] == 4.0 and st["action.invalid_index.-500"] == 2.0
    assert st[f"action.invalid_index.{n + 40}"] == 2.0 and st[f"action.invalid_index.{2 ** 31 - 1}"] == 2.0
    assert sim.error() == 0
    step(-(2 ** 31))          # fifth distinct far index
    assert sim.error() & 2
