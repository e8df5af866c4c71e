"""Benchmark/parity configurations = the rungs of BASELINE.json ``configs``."""
from __future__ import annotations

import numpy as np

from . import spec as S
from .mapgen import ascii_map, random_map

RESOURCES10 = ["ore", "hp", "laser", "armor", "gear", "energy", "carbon", "oxygen", "germanium", "silicon"]


def rung1_spec() -> S.GameSpec:
    """16x16 ASCII map, 4 agents, noop + 4 moves, obs 13x13 / T=100 (benchmarks/test_mettagrid_env_benchmark.py:21-29)."""
    return S.GameSpec(
        resource_names=list(RESOURCES10),
        agents=[S.AgentSpec(team_id=0) for _ in range(4)],
        objects={"wall": S.ObjectSpec(name="wall", kind="wall")},
        vibe_names=["default"], change_vibe_enabled=False,
        move_directions=["north", "south", "west", "east"],
        obs=S.ObsSpec(width=13, height=13, num_tokens=100), max_steps=0)


RUNG1_LINES = [
    "################",
    "#..............#",
    "#..@....#......#",
    "#.......#......#",
    "#.......#...@..#",
    "#..............#",
    "#....###.......#",
    "#..............#",
    "#..............#",
    "#.......###....#",
    "#..@...........#",
    "#..........#...#",
    "#..........#.@.#",
    "#..............#",
    "#..............#",
    "################",
]


def rung1_map() -> np.ndarray:
    return ascii_map(RUNG1_LINES, {"#": "wall", ".": "empty", "@": "agent.agent"})


def rung2_spec(obs_tokens: int = 200) -> S.GameSpec:
    """32x32 random map, 16 agents, noop + 4 moves + 4 vibes, obs 11x11 / T=200 (SURVEY.md §8d rung 2)."""
    return S.GameSpec(
        resource_names=list(RESOURCES10),
        agents=[S.AgentSpec(team_id=0) for _ in range(16)],
        objects={"wall": S.ObjectSpec(name="wall", kind="wall")},
        vibe_names=["default", "a", "b", "c"], change_vibe_enabled=True,
        move_directions=["north", "south", "west", "east"],
        obs=S.ObsSpec(width=11, height=11, num_tokens=obs_tokens), max_steps=0)


def rung2_map(seed: int, size: int = 32, agents: int = 16, walls: int = 40) -> np.ndarray:
    return random_map(size, size, {"wall": walls}, agents, seed)


def rung3_spec(obs_tokens: int = 200, use_attack_mutation: bool = True) -> S.GameSpec:
    """Rung 3: handler chains + tokenised obs (SURVEY.md §8d rung 3 and Appendix C).

    2 teams x 8 agents, extractors (withdraw by vibe), chests (deposit + stat), agent-vs-agent on_use
    (laser cost, attack or flat damage, loot), same-team swap, periodic regen, inventory/limit-modifier rewards.
    """
    A, T = S.ACTOR, S.TARGET
    extractor_use = S.FirstMatch([
        S.Handler([S.VibeFilter(A, "a")], [S.ResourceTransfer(T, A, "ore", 5)], "withdraw5"),
        S.Handler([], [S.ResourceTransfer(T, A, "ore", 1)], "withdraw1"),
    ])
    chest_use = S.Handler(
        [S.ResourceFilter(A, "ore", 1)],
        [S.ResourceTransfer(A, T, "ore", -1),
         S.SetStat("chest.ore", S.InventoryValue("ore"), scope="game", entity=T)], "deposit")
    damage = S.Attack("laser", "armor", "hp", 100) if use_attack_mutation else S.ResourceDelta(T, "hp", -10)
    agent_use = S.FirstMatch([
        S.Handler([S.NegFilter([S.SharedTagPrefixFilter("team:")]), S.ResourceFilter(A, "laser", 1)],
                  [S.ResourceDelta(A, "laser", -1), damage, S.ResourceTransfer(T, A, "ore", -1)], "attack"),
        S.Handler([S.SharedTagPrefixFilter("team:")], [S.Swap()], "swap"),
    ])
    regen = S.Handler([S.PeriodicFilter(10)], [S.ResourceDelta(T, "hp", 1)], "regen")

    def agent(team: int) -> S.AgentSpec:
        return S.AgentSpec(
            team_id=team, tags=["team:red" if team == 0 else "team:blue"],
            inventory=S.Inventory(initial={"hp": 100, "laser": 5, "armor": 2}, default_limit=100,
                                  limits=[S.Limit(["ore"], base=20, max=60, modifiers={"gear": 10})]),
            rewards=[S.RewardSpec(S.InventoryValue("ore")),
                     S.RewardSpec(S.SumValue([S.InventoryValue("hp")], [0.01]), per_tick=True)],
            on_use=agent_use, on_tick=regen)

    return S.GameSpec(
        resource_names=list(RESOURCES10),
        agents=[agent(0) for _ in range(8)] + [agent(1) for _ in range(8)],
        objects={
            "wall": S.ObjectSpec(name="wall", kind="wall"),
            "extractor": S.ObjectSpec(name="extractor", inventory=S.Inventory(initial={"ore": 50}, default_limit=100),
                                      on_use=extractor_use),
            "chest": S.ObjectSpec(name="chest", inventory=S.Inventory(initial={}, default_limit=65535,
                                                                     limits=[S.Limit(["ore"], base=65535)]),
                                  on_use=chest_use),
        },
        tags=["team:red", "team:blue"],
        vibe_names=["default", "a", "b", "c"], change_vibe_enabled=True,
        move_directions=["north", "south", "west", "east", "northwest", "northeast", "southwest", "southeast"],
        obs=S.ObsSpec(width=11, height=11, num_tokens=obs_tokens), max_steps=0)


def rung3_map(seed: int, size: int = 32) -> np.ndarray:
    return random_map(size, size, {"wall": 40, "extractor": 8, "chest": 4}, {"red": 8, "blue": 8}, seed)


def periodic(start: int, period: int, end: int = 100000) -> list:
    """Timesteps start, start+period, ... <= end (constant-period case of the reference's event helper
    python/src/mettagrid/config/event_config.py:27-48, whose default end is 100 000)."""
    if period <= 0:
        raise ValueError(f"period must be positive, got {period}")
    return list(range(start, end + 1, period))


RUNG4_TEAMS = ["red", "blue", "green", "yellow"]
RUNG4_OBJECTS = {"wall": 160, "extractor": 32, "chest": 16,
                 "healer_red": 4, "healer_blue": 4, "healer_green": 4, "healer_yellow": 4,
                 "flag_red": 2, "flag_blue": 2, "flag_green": 2, "flag_yellow": 2, "hub": 2, "wire": 12}
RUNG4_AGENTS = {t: 16 for t in RUNG4_TEAMS}
RUNG4_MAX_OBJECTS = 576   # 252 border walls + 246 placed objects + 64 agents = 562, rounded up


def rung4_spec(obs_tokens: int = 256, max_steps: int = 0) -> S.GameSpec:
    """Rung 4 = BASELINE.json configs[3]: the rung-3 rules on 64x64 maps with 64 agents in 4 teams,
    plus 16 static AoE sources (r=3, same-team filter, hp +2, presence shield +1), a mobile AoE on every agent
    (r=1, enemies lose 1 hp), one territory type over the team tags with 8 sources (strength 5, decay 1; presence
    energy +1, enter/exit stats), the aoe_mask observation token, three timestep events (periodic(50, 50) over a tag
    query with max_targets=4, a once(100) event that recomputes a materialized closure query, a periodic one whose
    target query comes up empty so its fallback fires) and that materialized closure query (hub -> wires within 2)."""
    A, T = S.ACTOR, S.TARGET
    extractor_use = S.FirstMatch([
        S.Handler([S.VibeFilter(A, "a")], [S.ResourceTransfer(T, A, "ore", 5)], "withdraw5"),
        S.Handler([], [S.ResourceTransfer(T, A, "ore", 1)], "withdraw1"),
    ])
    chest_use = S.Handler(
        [S.ResourceFilter(A, "ore", 1)],
        [S.ResourceTransfer(A, T, "ore", -1),
         S.SetStat("chest.ore", S.InventoryValue("ore"), scope="game", entity=T)], "deposit")
    agent_use = S.FirstMatch([
        S.Handler([S.NegFilter([S.SharedTagPrefixFilter("team:")]), S.ResourceFilter(A, "laser", 1)],
                  [S.ResourceDelta(A, "laser", -1), S.Attack("laser", "armor", "hp", 100),
                   S.ResourceTransfer(T, A, "ore", -1)], "attack"),
        S.Handler([S.SharedTagPrefixFilter("team:")], [S.Swap()], "swap"),
    ])
    regen = S.Handler([S.PeriodicFilter(10)], [S.ResourceDelta(T, "hp", 1)], "regen")
    sting = S.AOESpec(radius=1, is_static=False, filters=[S.NegFilter([S.SharedTagPrefixFilter("team:")])],
                      mutations=[S.ResourceDelta(T, "hp", -1)])

    def agent(team: int) -> S.AgentSpec:
        return S.AgentSpec(
            team_id=team, tags=[f"team:{RUNG4_TEAMS[team]}"],
            inventory=S.Inventory(initial={"hp": 100, "laser": 5, "armor": 2, "energy": 10}, default_limit=100,
                                  limits=[S.Limit(["ore"], base=20, max=60, modifiers={"gear": 10}),
                                          S.Limit(["shield"], base=5)]),
            rewards=[S.RewardSpec(S.InventoryValue("ore")),
                     S.RewardSpec(S.SumValue([S.InventoryValue("hp")], [0.01]), per_tick=True),
                     S.RewardSpec(S.StatValue("zone.entered", "agent"))],
            on_use=agent_use, on_tick=regen, aoes=[sting])

    heal = S.AOESpec(radius=3, filters=[S.SharedTagPrefixFilter("team:")], mutations=[S.ResourceDelta(T, "hp", 2)],
                     presence_deltas={"shield": 1})
    count_up = lambda name: S.SetStat(name, S.SumValue([S.StatValue(name, "agent"), S.ConstValue(1.0)]),  # noqa: E731
                                      scope="agent", entity=T)
    zone = S.TerritorySpec(
        tag_prefix="team:",
        on_enter=[S.Handler([], [count_up("zone.entered")])],
        on_exit=[S.Handler([], [count_up("zone.left")])],
        presence=[S.Handler([S.SharedTagPrefixFilter("team:")], [S.ResourceDelta(T, "energy", 1)])])
    objects = {
        "wall": S.ObjectSpec(name="wall", kind="wall"),
        "extractor": S.ObjectSpec(name="extractor", inventory=S.Inventory(initial={"ore": 50}, default_limit=100),
                                  on_use=extractor_use),
        "chest": S.ObjectSpec(name="chest", inventory=S.Inventory(initial={}, default_limit=65535,
                                                                 limits=[S.Limit(["ore"], base=65535)]),
                              on_use=chest_use),
        "hub": S.ObjectSpec(name="hub"),
        "wire": S.ObjectSpec(name="wire"),
    }
    for t in RUNG4_TEAMS:
        objects[f"healer_{t}"] = S.ObjectSpec(name=f"healer_{t}", tags=[f"team:{t}"], aoes=[heal])
        objects[f"flag_{t}"] = S.ObjectSpec(name=f"flag_{t}", tags=[f"team:{t}"],
                                            territory_controls=[S.TerritoryControl("zone", strength=5, decay=1)])
    events = {
        "supply": S.EventSpec(S.TagQuery("type:agent"), periodic(50, 50), [S.ResourceFilter(T, "hp", 1)],
                              [S.ResourceDelta(T, "energy", 3)], max_targets=4),
        "rewire": S.EventSpec(S.TagQuery("type:hub"), [100], [], [S.RecomputeMaterializedQuery("net")]),
        "audit": S.EventSpec(S.TagQuery("net", [S.ResourceFilter(T, "ore", 1)]), periodic(75, 75), [],
                             [S.ResourceDelta(T, "ore", -1)], fallback="relief"),
        "relief": S.EventSpec(S.TagQuery("type:agent", max_items=2, order_by="random"), [], [],
                              [S.ResourceDelta(T, "hp", 4)]),
    }
    return S.GameSpec(
        resource_names=list(RESOURCES10) + ["shield"],
        agents=[agent(t) for t in range(4) for _ in range(16)],
        objects=objects,
        tags=[f"team:{t}" for t in RUNG4_TEAMS] + ["net"],
        vibe_names=["default", "a", "b", "c"], change_vibe_enabled=True,
        move_directions=["north", "south", "west", "east", "northwest", "northeast", "southwest", "southeast"],
        obs=S.ObsSpec(width=11, height=11, num_tokens=obs_tokens, aoe_mask=True),
        events=events,
        materialize_queries=[S.MaterializedQuery("net", S.ClosureQuery(S.TagQuery("type:hub"), S.TagQuery("type:wire"),
                                                                       edge_filters=[S.MaxDistanceFilter(T, 2)]))],
        territories={"zone": zone},
        max_steps=max_steps, episode_truncates=True)


def rung4_map(seed: int, size: int = 64) -> np.ndarray:
    return random_map(size, size, dict(RUNG4_OBJECTS), dict(RUNG4_AGENTS), seed)
