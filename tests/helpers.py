"""Shared helpers for the parity tests: scenario table, stepping an engine pair with identical actions, and the
comparison of every caller-visible buffer plus the generalised signature payload."""
from __future__ import annotations

import numpy as np

from mettagrid_amd import presets, signature as sg
from mettagrid_amd import spec as S
from mettagrid_amd.compiler import compile_spec
from mettagrid_amd.mapgen import random_map


def torture_spec(max_steps: int = 40, truncates: bool = True) -> S.GameSpec:
    """Exercises the parts of the handler VM the benchmark rungs do not: limit groups with modifiers and drop
    enforcement, ClearInventory, Or / TagPrefix / GameValue filters, AllOf, on_after_use, obs values, local
    position, last_action_move, multi-digit inventory tokens (base 16), truncation."""
    A, T = S.ACTOR, S.TARGET
    shrine_use = S.AllOf([
        S.Handler([S.OrFilter([S.VibeFilter(A, "b"), S.ResourceFilter(A, "gear", 2)])],
                  [S.ResourceDelta(A, "gear", -1), S.ResourceDelta(A, "energy", 300)], "burn_gear"),
        S.Handler([S.TagPrefixFilter(A, "team:blue")], [S.ClearInventory(A, ["carbon", "oxygen"])], "clear_some"),
        S.Handler([S.GameValueFilter(A, S.InventoryValue("energy"), S.ConstValue(600.0))],
                  [S.ClearInventory(A), S.SetStat("shrine.wipes", S.StatValue("shrine.wipes", "game"), scope="game")],
                  "wipe"),
    ])
    forge_use = S.FirstMatch([
        S.Handler([S.ResourceFilter(A, "ore", 3)],
                  [S.ResourceDelta(A, "ore", -3), S.ResourceDelta(A, "gear", 1),
                   S.SetStat("forged", S.SumValue([S.InventoryValue("gear"), S.StatValue("forged", "agent")]),
                             scope="agent", entity=A)], "forge"),
        S.Handler([], [S.ResourceTransfer(T, A, "carbon", 2), S.ResourceTransfer(T, A, "oxygen", 7)], "scraps"),
    ])
    mine_use = S.Handler([], [S.ResourceTransfer(T, A, "ore", 4), S.ChangeVibe(A, "c")], "mine")
    agent_use = S.FirstMatch([
        S.Handler([S.NegFilter([S.SharedTagPrefixFilter("team:")])],
                  [S.ResourceTransfer(T, A, "gear", 1), S.Attack("laser", "armor", "hp", 150)], "rob"),
        S.Handler([S.SharedTagPrefixFilter("team:")], [S.Swap()], "swap"),
    ])
    after_use = S.Handler([], [S.ResourceDelta(A, "energy", -1)], "tired")

    def agent(team: int, i: int) -> S.AgentSpec:
        return S.AgentSpec(
            team_id=team, tags=["team:red" if team == 0 else "team:blue"], vibe=i % 3,
            inventory=S.Inventory(
                initial={"laser": 3, "hp": 40 + i, "gear": i % 3, "energy": 20, "armor": 1, "oxygen": 0},
                default_limit=1000,
                limits=[S.Limit(["ore", "carbon", "oxygen"], base=6, max=40, modifiers={"gear": 8}),
                        S.Limit(["energy"], base=1000)]),
            rewards=[S.RewardSpec(S.InventoryValue("gear")),
                     S.RewardSpec(S.SumValue([S.InventoryValue("hp"), S.InventoryValue("energy")], [0.5, 0.25], log=True),
                                  per_tick=True),
                     S.RewardSpec(S.StatValue("forged", "agent")),
                     S.RewardSpec(S.RatioValue(S.InventoryValue("ore"), S.MaxValue([S.InventoryValue("gear"), S.ConstValue(1.0)])))],
            on_use=agent_use, on_after_use=after_use,
            on_tick=S.Handler([S.PeriodicFilter(3, 2)], [S.ResourceDelta(T, "energy", -2), S.ResourceDelta(T, "hp", -1)], "decay"))

    return S.GameSpec(
        resource_names=["ore", "hp", "laser", "armor", "gear", "energy", "carbon", "oxygen"],
        agents=[agent(0, i) for i in range(3)] + [agent(1, i) for i in range(3)],
        objects={
            "wall": S.ObjectSpec(name="wall", kind="wall"),
            "block": S.ObjectSpec(name="block", kind="wall", tags=["solid"]),
            "mine": S.ObjectSpec(name="mine", tags=["res:ore"], inventory=S.Inventory(initial={"ore": 500}), on_use=mine_use),
            "forge": S.ObjectSpec(name="forge", vibe=2, inventory=S.Inventory(initial={"oxygen": 300, "carbon": 9}, default_limit=400),
                                  on_use=forge_use),
            "shrine": S.ObjectSpec(name="shrine", on_use=shrine_use),
        },
        tags=["team:red", "team:blue"],
        vibe_names=["default", "a", "b", "c"], change_vibe_enabled=True,
        move_directions=["north", "south", "west", "east", "northeast", "southwest"],
        obs=S.ObsSpec(width=7, height=9, num_tokens=120, token_value_base=16, last_action_move=True, local_position=True,
                      values={"my_energy": S.InventoryValue("energy"), "forged": S.StatValue("forged", "agent")}),
        max_steps=max_steps, episode_truncates=truncates)


def torture_map(seed: int) -> np.ndarray:
    return random_map(12, 14, {"wall": 8, "block": 5, "mine": 4, "forge": 3, "shrine": 3}, {"red": 3, "blue": 3}, seed)


SCENARIOS = {
    # name: (spec factory, map factory(seed), steps, allow invalid action indices)
    "rung1": (presets.rung1_spec, lambda s: presets.rung1_map(), 120, False),
    "rung1_invalid": (presets.rung1_spec, lambda s: presets.rung1_map(), 60, True),
    "rung2": (presets.rung2_spec, presets.rung2_map, 128, False),
    "rung3": (presets.rung3_spec, presets.rung3_map, 160, False),
    "rung3_flat_damage": (lambda: presets.rung3_spec(use_attack_mutation=False), presets.rung3_map, 100, True),
    "torture": (torture_spec, torture_map, 45, False),
    "torture_terminal": (lambda: torture_spec(25, False), torture_map, 30, True),
}


# object slots per env where the default (H * W) is not what the preset ships with
MAX_OBJECTS = {"rung4_full": presets.RUNG4_MAX_OBJECTS}


def compile_scenario(name: str, spec, height: int, width: int):
    return compile_spec(spec, height, width, max_objects=MAX_OBJECTS.get(name))


def make_actions(prog, seed: int, steps: int, invalid: bool) -> tuple:
    rng = np.random.RandomState(1000 + seed)
    n = len(prog.action_names)
    lo, hi = (-2, n + 2) if invalid else (0, n)
    return (rng.randint(lo, hi, (steps, prog.num_agents)).astype(np.int32),
            rng.randint(lo, hi, (steps, prog.num_agents)).astype(np.int32))


def compare_snapshots(a: dict, b: dict, where: str) -> None:
    for k in ("obs", "rewards", "terminals", "truncations", "action_success", "episode_rewards"):
        if not np.array_equal(np.asarray(a[k]), np.asarray(b[k])):
            detail = ""
            if k == "obs":
                idx = np.argwhere((a[k] != b[k]).any(axis=2))
                ag, tok = idx[0]
                detail = f" agent {ag} token {tok}: {a[k][ag, max(0, tok - 2):tok + 3].tolist()} vs {b[k][ag, max(0, tok - 2):tok + 3].tolist()}"
            else:
                detail = f" {np.asarray(a[k]).tolist()} vs {np.asarray(b[k]).tolist()}"
            raise AssertionError(f"{where}: '{k}' differs.{detail}")


def payload_from_raw(prog, raw_objects, current_stat_reward, raw_stats, snap, steps, seed) -> dict:
    return sg.payload(sg.objects_from_raw(prog, raw_objects, current_stat_reward), sg.stats_dicts(prog, *raw_stats),
                      snap["action_success"], snap["episode_rewards"], steps, seed)


def diff_payload(pa: dict, pb: dict) -> str:
    out = []
    for k in pa:
        if pa[k] != pb[k]:
            if k == "stats":
                ga, gb = pa[k]["game"], pb[k]["game"]
                out.append(f"game stats: only-a {[x for x in ga if x not in gb]} only-b {[x for x in gb if x not in ga]}")
                for i, (x, y) in enumerate(zip(pa[k]["agent"], pb[k]["agent"])):
                    if x != y:
                        out.append(f"agent {i} stats: only-a {[e for e in x if e not in y]} only-b {[e for e in y if e not in x]}")
                        break
            elif k == "objects":
                for x, y in zip(pa[k], pb[k]):
                    if x != y:
                        out.append(f"object: {x} vs {y}")
                        break
            else:
                out.append(f"{k}: {pa[k]} vs {pb[k]}")
    return "; ".join(out)


def rung4_spec(max_steps: int = 0) -> S.GameSpec:
    """Rung 4: static + mobile AoE, a territory type with enter/exit/presence handlers and the aoe_mask
    token, timestep events (max_targets with the shared RNG, a fallback, a once event), tag mutations with a
    lifecycle handler, a materialized closure query that an event recomputes, query values/filters/mutations."""
    A, T = S.ACTOR, S.TARGET
    teams = ["red", "blue", "green"]

    def agent(team: int, i: int) -> S.AgentSpec:
        return S.AgentSpec(
            team_id=team, tags=[f"team:{teams[team]}"],
            inventory=S.Inventory(initial={"hp": 50, "energy": 10}, default_limit=200,
                                  limits=[S.Limit(["shield"], base=3)]),
            rewards=[S.RewardSpec(S.InventoryValue("hp")), S.RewardSpec(S.StatValue("zone.entered", "agent")),
                     S.RewardSpec(S.QueryCountValue(S.TagQuery("marked")), per_tick=True)],
            aoes=[S.AOESpec(radius=1, is_static=False, filters=[S.NegFilter([S.SharedTagPrefixFilter("team:")])],
                            mutations=[S.ResourceDelta(T, "hp", -1)])],
            on_tag_add={"marked": S.Handler([], [S.ResourceDelta(T, "energy", 5)], "marked_bonus")},
            on_tag_remove={"marked": S.Handler([], [S.ResourceDelta(T, "energy", -1)], "unmarked")},
            on_use=S.FirstMatch([
                S.Handler([S.NegFilter([S.SharedTagPrefixFilter("team:")])],
                          [S.AddTag(T, "marked"), S.GameValueMutation(S.InventoryValue("energy"), S.ConstValue(-2.0), A),
                           S.GameValueMutation(S.StatValue("tags.given", "agent"), S.ConstValue(1.0), A)], "mark"),
                S.Handler([], [S.Swap()], "swap")]))

    heal = S.AOESpec(radius=3, filters=[S.TagPrefixFilter(T, "team:")],
                     mutations=[S.ResourceDelta(T, "hp", 2), S.ResourceDelta(T, "energy", -1)],
                     presence_deltas={"shield": 1})
    burn = S.AOESpec(radius=2, filters=[], mutations=[S.ResourceDelta(T, "hp", -3)])
    zone = S.TerritorySpec(
        tag_prefix="team:",
        on_enter=[S.Handler([], [S.SetStat("zone.entered", S.SumValue([S.StatValue("zone.entered", "agent"), S.ConstValue(1.0)]),
                                           scope="agent", entity=T)])],
        on_exit=[S.Handler([S.SharedTagPrefixFilter("team:")], [S.ResourceDelta(T, "energy", -1)])],
        presence=[S.Handler([S.SharedTagPrefixFilter("team:")], [S.ResourceDelta(T, "energy", 1)]),
                  S.Handler([S.NegFilter([S.SharedTagPrefixFilter("team:")])], [S.ResourceDelta(T, "hp", -1)])])

    def flag(team: str) -> S.ObjectSpec:
        return S.ObjectSpec(name=f"flag_{team}", tags=[f"team:{team}"],
                            territory_controls=[S.TerritoryControl("zone", strength=4, decay=1)])

    events = {
        "supply": S.EventSpec(S.TagQuery("type:agent"), list(range(5, 200, 10)), [S.ResourceFilter(T, "hp", 1)],
                              [S.ResourceDelta(T, "energy", 3)], max_targets=4),
        "purge": S.EventSpec(S.TagQuery("marked"), [12, 30, 31, 60], [], [S.RemoveTag(T, "marked")], fallback="mercy"),
        "mercy": S.EventSpec(S.TagQuery("type:agent", max_items=2, order_by="random"), [],
                             [], [S.ResourceDelta(T, "hp", 4)]),
        "rewire": S.EventSpec(S.TagQuery("type:hub"), [8, 25, 50], [], [S.RecomputeMaterializedQuery("net")]),
        "tax": S.EventSpec(S.TagQuery("type:hub"), [20, 40],
                           # EventConfig has no add_query_resource_filter binding; NOT(NOT(x)) carries it
                           [S.NegFilter([S.NegFilter([S.QueryResourceFilter(S.TagQuery("type:agent"), {"energy": 30})])])],
                           [S.QueryInventoryMutation(S.FilteredQuery(S.TagQuery("type:agent"), [S.MaxDistanceFilter(T, 6, S.TagQuery("type:hub"))]),
                                                     {"energy": -2}, source=T, transfer_stat_names={"energy": "tax.energy"})]),
        "beam": S.EventSpec(S.RaycastQuery(S.TagQuery("type:hub"), max_range=5, blocker=[S.TagPrefixFilter(T, "type:wall")],
                                           include_blocker=False), [15, 35], [S.TagPrefixFilter(T, "team:")],
                            [S.ResourceDelta(T, "hp", -2), S.RemoveTagsWithPrefix(T, "mark")]),
    }
    return S.GameSpec(
        resource_names=["hp", "energy", "shield", "ore"],
        agents=[agent(t, i) for t in range(3) for i in range(4)],
        objects={
            "wall": S.ObjectSpec("wall", kind="wall"),
            "healer": S.ObjectSpec("healer", aoes=[heal]),
            "fire": S.ObjectSpec("fire", aoes=[burn]),
            "hub": S.ObjectSpec("hub", inventory=S.Inventory(initial={}, default_limit=1000)),
            "wire": S.ObjectSpec("wire"),
            "flag_red": flag("red"), "flag_blue": flag("blue"), "flag_green": flag("green"),
        },
        tags=["team:red", "team:blue", "team:green", "marked", "net"],
        vibe_names=["default", "a"], change_vibe_enabled=True,
        move_directions=["north", "south", "west", "east", "northwest", "southeast"],
        obs=S.ObsSpec(width=9, height=9, num_tokens=160, aoe_mask=True,
                      values={"marked_n": S.QueryCountValue(S.TagQuery("marked")),
                              "team_hp": S.QueryInventoryValue("hp", S.FilteredQuery(S.TagQuery("type:agent"), [S.SharedTagPrefixFilter("team:")]))}),
        events=events,
        materialize_queries=[S.MaterializedQuery("net", S.ClosureQuery(S.TagQuery("type:hub"), S.TagQuery("type:wire"),
                                                                       edge_filters=[S.MaxDistanceFilter(T, 2)]))],
        territories={"zone": zone},
        on_tick=S.Handler([S.PeriodicFilter(7, 3)], [S.SetStat("ticks7", S.SumValue([S.StatValue("ticks7", "game"), S.ConstValue(1.0)]))]),
        max_steps=max_steps, episode_truncates=True)


def rung4_map(seed: int) -> np.ndarray:
    return random_map(20, 22, {"wall": 18, "healer": 3, "fire": 3, "hub": 2, "wire": 10, "flag_red": 2, "flag_blue": 2,
                               "flag_green": 2}, {"red": 4, "blue": 4, "green": 4}, seed)


SCENARIOS["rung4"] = (rung4_spec, rung4_map, 70, False)
# BASELINE.json configs[3] at its own shape (64x64, 64 agents / 4 teams): events fire at 50, 75 and 100
SCENARIOS["rung4_full"] = (presets.rung4_spec, presets.rung4_map, 104, False)
SCENARIOS["rung4_truncating"] = (lambda: rung4_spec(33), rung4_map, 36, True)


def dynamic_spec() -> S.GameSpec:
    """Objects created and removed at run time + move handlers with range: ranged beam (MaxDistance scan), push,
    depletable extractors (remove_source_when_empty with a live AoE), event-driven smash-and-replace
    (tests/test_spawn_in_event.py pattern), RaycastSpawn from a totem, spawned objects carrying AoEs and tags."""
    A, T = S.ACTOR, S.TARGET
    marker_aoe = S.AOESpec(radius=1, filters=[S.TagPrefixFilter(T, "type:agent")], mutations=[S.ResourceDelta(T, "hp", -1)])
    well_aoe = S.AOESpec(radius=2, filters=[], mutations=[], presence_deltas={"mob": 2})
    beam = S.Handler([S.VibeFilter(A, "a"), S.MaxDistanceFilter(T, 3), S.TagPrefixFilter(T, "type:agent")],
                     [S.ResourceDelta(T, "mob", -1), S.ResourceDelta(A, "ore", -1)], "beam")
    push = S.Handler([S.VibeFilter(A, "b"), S.TagPrefixFilter(T, "type:boulder")], [S.PushObject(), S.Relocate()], "push")
    well_use = S.Handler([], [S.ResourceTransfer(T, A, "ore", 1, remove_source_when_empty=True)], "draw")
    totem_use = S.Handler([S.ResourceFilter(A, "ore", 1)],
                          [S.ResourceDelta(A, "ore", -1),
                           S.RaycastSpawn("marker", [(-1, 0), (1, 0), (0, 1), (0, -1)], 2, [S.TagPrefixFilter(T, "type:wall")])], "totem")
    events = {
        "smash": S.EventSpec(S.TagQuery("type:crate", [S.ResourceFilter(T, "hp", 1)]), [3, 9, 20, 21], [],
                             [S.ResourceDelta(T, "hp", -1), S.ResourceTransfer(T, A, "hp", 0, remove_source_when_empty=True),
                              S.SpawnObject("marker")], max_targets=2),
        "decay": S.EventSpec(S.TagQuery("spawned"), [15, 30], [], [S.AddTag(T, "old")]),
    }
    return S.GameSpec(
        resource_names=["hp", "ore", "mob"],
        agents=[S.AgentSpec(team_id=0, inventory=S.Inventory(initial={"hp": 30, "mob": 5, "ore": 2}, default_limit=50))
                for _ in range(5)],
        objects={
            "wall": S.ObjectSpec("wall", kind="wall"),
            "boulder": S.ObjectSpec("boulder"),
            "crate": S.ObjectSpec("crate", inventory=S.Inventory(initial={"hp": 1}, limits=[S.Limit(["hp"], base=1, max=1)])),
            "marker": S.ObjectSpec("marker", tags=["spawned"], vibe=1, aoes=[marker_aoe],
                                   inventory=S.Inventory(initial={"ore": 4})),
            "well": S.ObjectSpec("well", inventory=S.Inventory(initial={"ore": 3}), on_use=well_use, aoes=[well_aoe]),
            "totem": S.ObjectSpec("totem", on_use=totem_use),
        },
        tags=["spawned", "old"],
        vibe_names=["default", "a", "b"], change_vibe_enabled=True,
        move_directions=["north", "south", "west", "east"],
        move_handlers=[beam, push],
        obs=S.ObsSpec(width=7, height=7, num_tokens=120, last_action_move=True),
        events=events, max_steps=0)


def dynamic_map(seed: int) -> np.ndarray:
    return random_map(12, 12, {"wall": 6, "boulder": 6, "crate": 5, "well": 4, "totem": 3}, 5, seed)


SCENARIOS["dynamic"] = (dynamic_spec, dynamic_map, 60, False)


def wide_spec() -> S.GameSpec:
    """Rung-3 rules with shapes the other scenarios do not reach: 24 agents (two encode passes of 16, the second one
    half empty), a 13x13 window, and a token budget that is not a multiple of four (ragged row tail, unaligned row
    pitch in the observation buffer)."""
    import copy
    from mettagrid_amd import presets
    sp = presets.rung3_spec()
    red, blue = sp.agents[0], sp.agents[-1]
    sp.agents = [copy.deepcopy(red) for _ in range(12)] + [copy.deepcopy(blue) for _ in range(12)]
    sp.obs = S.ObsSpec(width=13, height=13, num_tokens=243)
    return sp


def wide_map(seed: int) -> np.ndarray:
    return random_map(20, 20, {"wall": 20, "extractor": 10, "chest": 6}, {"red": 12, "blue": 12}, seed)


SCENARIOS["wide"] = (wide_spec, wide_map, 25, True)


def crowd_spec() -> S.GameSpec:
    """Rung-3 rules with more agents than a wavefront has lanes (70): agent loops that stride by 64, five encode
    passes, the serial fallback of the token statistics."""
    import copy
    from mettagrid_amd import presets
    sp = presets.rung3_spec()
    red, blue = sp.agents[0], sp.agents[-1]
    sp.agents = [copy.deepcopy(red) for _ in range(35)] + [copy.deepcopy(blue) for _ in range(35)]
    sp.obs = S.ObsSpec(width=9, height=9, num_tokens=200)
    return sp


def crowd_map(seed: int) -> np.ndarray:
    return random_map(30, 30, {"wall": 30, "extractor": 12, "chest": 6}, {"red": 35, "blue": 35}, seed)


SCENARIOS["crowd"] = (crowd_spec, crowd_map, 12, False)


def torture_base10_spec() -> S.GameSpec:
    """The torture rules with a token value base that is not a power of two (division path of the digit encoder)."""
    sp = torture_spec()
    sp.obs.token_value_base = 10
    sp.obs.num_tokens = 200
    return sp


SCENARIOS["torture_base10"] = (torture_base10_spec, torture_map, 30, False)


def keyhole_spec() -> S.GameSpec:
    """Rung-3 rules seen through the smallest window (3x3: five in-mask cells) with an odd agent count."""
    import copy
    from mettagrid_amd import presets
    sp = presets.rung3_spec()
    red, blue = sp.agents[0], sp.agents[-1]
    sp.agents = [copy.deepcopy(red) for _ in range(3)] + [copy.deepcopy(blue) for _ in range(2)]
    sp.obs = S.ObsSpec(width=3, height=3, num_tokens=60)
    return sp


def keyhole_map(seed: int) -> np.ndarray:
    return random_map(7, 10, {"wall": 6, "extractor": 4, "chest": 3}, {"red": 3, "blue": 2}, seed)


SCENARIOS["keyhole"] = (keyhole_spec, keyhole_map, 40, True)


def letterbox_spec() -> S.GameSpec:
    """Rung-3 rules with the widest window the reference allows in one direction and a narrow one in the other
    (15x7), on a map smaller than the window: most window cells are out of bounds for every agent."""
    import copy
    from mettagrid_amd import presets
    sp = presets.rung3_spec()
    red, blue = sp.agents[0], sp.agents[-1]
    sp.agents = [copy.deepcopy(red) for _ in range(3)] + [copy.deepcopy(blue) for _ in range(3)]
    sp.obs = S.ObsSpec(width=15, height=7, num_tokens=250)
    return sp


def letterbox_map(seed: int) -> np.ndarray:
    return random_map(9, 12, {"wall": 10, "extractor": 5, "chest": 3}, {"red": 3, "blue": 3}, seed)


SCENARIOS["letterbox"] = (letterbox_spec, letterbox_map, 40, False)


def minmax_spec() -> S.GameSpec:
    """The torture rules with MinValue everywhere a game value can stand (config/game_value_config.hpp MinValueConfig):
    a per-tick reward, nested under MaxValue and RatioValue, as the threshold of a GameValueFilter and as an obs value."""
    import dataclasses
    base = torture_spec(40, True)
    lo = S.MinValue([S.InventoryValue("hp"), S.InventoryValue("energy"), S.ConstValue(30.0)])
    nested = S.MaxValue([S.MinValue([S.InventoryValue("gear"), S.InventoryValue("laser")]), S.ConstValue(0.5)])
    agents = []
    for a in base.agents:
        rewards = [S.RewardSpec(lo, per_tick=True),
                   S.RewardSpec(S.RatioValue(S.InventoryValue("ore"), nested)),
                   S.RewardSpec(S.MinValue([S.StatValue("forged", "agent"), S.ConstValue(2.0)]))]
        agents.append(dataclasses.replace(a, rewards=rewards))
    A = S.ACTOR
    gate = S.Handler([S.GameValueFilter(A, S.InventoryValue("hp"), S.MinValue([S.InventoryValue("energy"), S.ConstValue(25.0)]))],
                     [S.ResourceDelta(A, "energy", 2)], "gate")
    objects = dict(base.objects)
    objects["shrine"] = dataclasses.replace(objects["shrine"], on_use=gate)
    obs = dataclasses.replace(base.obs, values={"floor": lo, "pair": S.MinValue([S.InventoryValue("gear"), S.InventoryValue("laser")])})
    return dataclasses.replace(base, agents=agents, objects=objects, obs=obs)


SCENARIOS["minmax"] = (minmax_spec, torture_map, 45, False)


def thirteen_spec() -> S.GameSpec:
    """The torture rules at the engine's resource limit (13 = the bucket count of libstdc++'s smallest unordered_map
    table): every agent starts with all five extra resources in a scrambled insertion order, the mine
    drains some to zero (erase) and hands others back (re-insert at the list head), so inventory token order walks
    through erase / insert at the highest ids.  R = 14 is refused (test_host_logic.py)."""
    import dataclasses
    base = torture_spec(40, True)
    extra = ["x8", "x9", "x10", "x11", "x12"]
    A, T = S.ACTOR, S.TARGET
    agents = []
    for i, a in enumerate(base.agents):
        init = dict(a.inventory.initial)
        for j in range(5):   # scrambled, agent-dependent insertion order
            name = extra[(j * 2 + i) % 5]
            init[name] = 1 + (i + j) % 3
        agents.append(dataclasses.replace(a, inventory=dataclasses.replace(a.inventory, initial=init)))
    mine_use = S.Handler([], [S.ResourceTransfer(A, T, "x12", -1), S.ResourceTransfer(A, T, "x9", 1),
                              S.ResourceTransfer(T, A, "x8", 2), S.ResourceTransfer(T, A, "ore", 4),
                              S.ResourceDelta(A, "x10", -1), S.ResourceDelta(A, "x11", 1)], "mine13")
    objects = dict(base.objects)
    objects["mine"] = dataclasses.replace(objects["mine"], on_use=mine_use,
                                          inventory=S.Inventory(initial={"ore": 500, "x8": 40, "x12": 0}, default_limit=1000))
    return dataclasses.replace(base, resource_names=list(base.resource_names) + extra, agents=agents, objects=objects)


SCENARIOS["thirteen"] = (thirteen_spec, torture_map, 45, False)


def lit_spec() -> S.GameSpec:
    """A materialized RAYCAST tag with no tag mutation anywhere in the program: the rays of every lamp put the tag "lit" on
    whatever they cross — walls, boxes, agents — so classes that never carry a leaf tag of the query (plain walls) must
    still show it in observations (query_config.hpp: a raycast collects the objects on its rays, blockers included).
    Agents block rays, so the event that recomputes the query moves the lit set as they walk."""
    T = S.TARGET
    rays = S.RaycastQuery(S.TagQuery("type:lamp"), max_range=4, blocker=[S.TagPrefixFilter(T, "type:agent")],
                          include_blocker=True)
    events = {"relight": S.EventSpec(S.TagQuery("type:lamp"), list(range(4, 200, 5)), [], [S.RecomputeMaterializedQuery("lit")])}
    return S.GameSpec(
        resource_names=["hp", "ore"],
        agents=[S.AgentSpec(team_id=0, inventory=S.Inventory(initial={"hp": 9}, default_limit=50),
                            rewards=[S.RewardSpec(S.QueryCountValue(S.TagQuery("lit")), per_tick=True)]) for _ in range(5)],
        objects={"wall": S.ObjectSpec("wall", kind="wall"), "lamp": S.ObjectSpec("lamp"),
                 "box": S.ObjectSpec("box", inventory=S.Inventory(initial={"ore": 3}))},
        tags=["lit"],
        vibe_names=["default", "a"], change_vibe_enabled=True,
        move_directions=["north", "south", "west", "east"],
        obs=S.ObsSpec(width=9, height=9, num_tokens=150),
        events=events,
        materialize_queries=[S.MaterializedQuery("lit", rays)],
        max_steps=0)


def lit_map(seed: int) -> np.ndarray:
    return random_map(12, 13, {"wall": 14, "lamp": 3, "box": 5}, 5, seed)


SCENARIOS["lit"] = (lit_spec, lit_map, 40, False)


def delta_spec() -> S.GameSpec:
    """StatValue(delta=True) everywhere a game value can stand — a top-level reward entry (tests/test_reward_config.py:85 of the
    reference), a per-tick entry nested in a sum, an observation value, a filter threshold, the source of a SetStat.  The
    reference engine resolves every value afresh, so the delta baseline is never consumed and the flag changes nothing
    (mettagrid_amd/spec.py StatValue); the goldens of this scenario come from the real engine."""
    import dataclasses
    base = torture_spec(40, True)
    forged = S.StatValue("forged", "agent", delta=True)
    agents = []
    for a in base.agents:
        rewards = [S.RewardSpec(forged),
                   S.RewardSpec(S.SumValue([S.StatValue("ore.gained", "agent", delta=True), S.InventoryValue("hp")], [0.2, 0.01]), per_tick=True),
                   S.RewardSpec(S.StatValue("ticks", "game", delta=True))]
        agents.append(dataclasses.replace(a, rewards=rewards))
    A = S.ACTOR
    gate = S.Handler([S.GameValueFilter(A, S.InventoryValue("hp"), S.StatValue("forged", "agent", delta=True))],
                     [S.ResourceDelta(A, "energy", 2), S.SetStat("forged.seen", S.StatValue("forged", "agent", delta=True), scope="agent", entity=A)], "gate")
    objects = dict(base.objects)
    objects["shrine"] = dataclasses.replace(objects["shrine"], on_use=gate)
    obs = dataclasses.replace(base.obs, values={"d_forged": forged, "d_gain": S.StatValue("ore.gained", "agent", delta=True)})
    return dataclasses.replace(base, agents=agents, objects=objects, obs=obs,
                               on_tick=S.Handler([], [S.SetStat("ticks", S.SumValue([S.StatValue("ticks", "game"), S.ConstValue(1.0)]))]))


SCENARIOS["delta"] = (delta_spec, torture_map, 45, False)


def reference_infos(prog, o) -> dict:
    """What StatsTracker.on_episode_end puts into infos for ONE env (python/src/mettagrid/envs/stats_tracker.py:30-47), from
    an oracle env's stats: the game dict, per key the sum over the agents that hold it in agent order / num_agents (Python
    floats), the per-agent dicts.  Pinned to the reference's own StatsTracker by tests/test_infos_fixture.py."""
    A = prog.num_agents
    sd = sg.stats_dicts(prog, *o.raw_stats(), extra=o.invalid_index_extra())
    agent = {}
    for agent_stats in sd["agent"]:
        for n, v in agent_stats.items():
            agent[n] = agent.get(n, 0) + v
    for n, v in agent.items():
        agent[n] = v / A
    return {"game": sd["game"], "agent": agent, "per_agent": {str(i): dict(s) for i, s in enumerate(sd["agent"])},
            "episode_rewards": o.snapshot()["episode_rewards"], "steps": o.current_step}
