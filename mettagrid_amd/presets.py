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
