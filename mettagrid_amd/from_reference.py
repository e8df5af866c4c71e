"""Lowering of the reference's C++ game config tree into the engine's own description, and the ``MettaGrid`` class the
reference's Python layer constructs.

Input: the tree of config records ``mettagrid/config/mettagrid_c_config.py:576-1007`` (``convert_to_cpp_game_config``)
builds from the classes of ``mettagrid_amd.mettagrid_c`` — the same tree it hands to the reference's pybind ``MettaGrid``.
Everything in it is already id-based (resource / tag / vibe / type / feature ids); ``game_spec`` turns it back into names
with the tables the tree carries (``resource_names``, ``vibe_names``, ``tag_id_map``, ``feature_ids``), produces a
``GameSpec`` and ``compile_reference_config`` compiles it and CHECKS that the compiler assigned exactly the ids the
reference assigned (tags, object types, observation features, action names) — the two id schemes are restated
independently (compiler.py / id_map.py:161-235), so a mismatch is an error, not something to paper over.
"""
from __future__ import annotations

import numpy as np

from . import mettagrid_c as C
from . import spec as S
from .compiler import Program, UnsupportedFeature, compile_spec, type_tag

_TEAM_IDS = {name: tid for tid, name in S.TEAM_NAMES.items()}


class ReferenceConfigError(ValueError):
    """The reference config tree uses something the engine cannot represent, or the two id schemes disagree."""


class _Lowering:
    def __init__(self, cfg) -> None:
        self.cfg = cfg
        self.res = list(cfg.resource_names)
        self.vibes = list(cfg.vibe_names)
        self.tags = dict(cfg.tag_id_map)                       # id -> name
        self.feature_name = {fid: name for name, fid in cfg.feature_ids.items()}

    # ---- small helpers -----------------------------------------------------------------------------------------
    def ent(self, e) -> str:
        return S.ACTOR if e in (C.EntityRef.actor, C.StatsEntity.actor) else S.TARGET

    def tag_names(self, ids) -> list:
        return [self.tags[int(t)] for t in ids]

    # ---- game values (mettagrid_c_value_config.py:36-99) ---------------------------------------------------------
    def value(self, v):
        n = type(v).__name__
        if n == "InventoryValueConfig":
            return S.InventoryValue(self.res[v.id])
        if n == "StatValueConfig":
            return S.StatValue(v.stat_name, "agent" if v.scope == C.GameValueScope.AGENT else "game",
                               delta=bool(getattr(v, "delta", False)))   # (no effect in the reference engine: spec.StatValue)
        if n == "ConstValueConfig":
            return S.ConstValue(float(v.value))
        if n == "SumValueConfig":
            w = list(v.weights) if getattr(v, "weights", None) else None
            return S.SumValue([self.value(x) for x in v.values], w, bool(v.log))
        if n == "RatioValueConfig":
            return S.RatioValue(self.value(v.numerator), self.value(v.denominator))
        if n == "MaxValueConfig":
            return S.MaxValue([self.value(x) for x in v.values])
        if n == "MinValueConfig":
            return S.MinValue([self.value(x) for x in v.values])
        if n == "QueryInventoryValueConfig":
            return S.QueryInventoryValue(self.res[v.id], self.query(v.query))
        if n == "QueryCountValueConfig":
            return S.QueryCountValue(self.query(v.query))
        raise ReferenceConfigError(f"unknown game value config {n}")

    # ---- queries (mettagrid_c_config.py:83-180) --------------------------------------------------------------------
    def query(self, holder):
        q = holder.config if type(holder).__name__ == "QueryConfigHolder" else holder
        n = type(q).__name__
        mx = None if getattr(q, "max_items", None) is None else self.value(q.max_items)
        order = "random" if getattr(q, "order_by", C.QueryOrderBy.none) == C.QueryOrderBy.random else "none"
        if n == "TagQueryConfig":
            return S.TagQuery(self.tags[q.tag_id], self.filters(q.filters), mx, order)
        if n == "FilteredQueryConfig":
            return S.FilteredQuery(self.query(q.source), self.filters(q.filters), mx, order)
        if n == "ClosureQueryConfig":
            return S.ClosureQuery(self.query(q.source), None if q.candidates is None else self.query(q.candidates),
                                  self.filters(q.edge_filters), self.filters(q.result_filters), mx, order)
        if n == "RaycastQueryConfig":
            return S.RaycastQuery(self.query(q.source), self.value(q.max_range), [tuple(d) for d in q.directions],
                                  self.filters(q.blocker), bool(q.include_blocker), mx, order)
        raise ReferenceConfigError(f"unknown query config {n}")

    # ---- filters ---------------------------------------------------------------------------------------------------
    def filters(self, flist) -> list:
        return [self.filter(f) for f in flist]

    def filter(self, f):
        n = type(f).__name__
        if n == "VibeFilterConfig":
            return S.VibeFilter(self.ent(f.entity), self.vibes[f.vibe_id])
        if n == "ResourceFilterConfig":
            return S.ResourceFilter(self.ent(f.entity), self.res[f.resource_id], int(f.min_amount))
        if n == "SharedTagPrefixFilterConfig":
            return S.SharedTagPrefixFilter("", self.tag_names(f.tag_ids))
        if n == "TagPrefixFilterConfig":
            return S.TagPrefixFilter(self.ent(f.entity), "", self.tag_names(f.tag_ids))
        if n == "GameValueFilterConfig":
            return S.GameValueFilter(self.ent(f.entity), self.value(f.value), self.value(f.threshold))
        if n == "PeriodicFilterConfig":
            return S.PeriodicFilter(int(f.period), int(f.start_on))
        if n == "NegFilterConfig":
            return S.NegFilter(self.filters(f.inner))
        if n == "OrFilterConfig":
            return S.OrFilter(self.filters(f.inner))
        if n == "MaxDistanceFilterConfig":
            return S.MaxDistanceFilter(self.ent(f.entity), int(f.radius), None if f.source is None else self.query(f.source))
        if n == "QueryResourceFilterConfig":
            return S.QueryResourceFilter(self.query(f.query), {self.res[r]: int(m) for r, m in f.requirements})
        if n == "TargetLocEmptyFilterConfig":
            return S.TargetLocEmptyFilter()
        if n == "TargetIsUsableFilterConfig":
            return S.TargetIsUsableFilter()
        raise ReferenceConfigError(f"unknown filter config {n}")

    # ---- mutations (mettagrid_c_mutations.py:105-269) --------------------------------------------------------------
    def mutations(self, mlist) -> list:
        return [self.mutation(m) for m in mlist]

    def mutation(self, m):
        n = type(m).__name__
        if n == "ResourceDeltaMutationConfig":
            return S.ResourceDelta(self.ent(m.entity), self.res[m.resource_id], int(m.delta))
        if n == "ResourceTransferMutationConfig":
            return S.ResourceTransfer(self.ent(m.source), self.ent(m.destination), self.res[m.resource_id], int(m.amount),
                                      bool(m.remove_source_when_empty))
        if n == "ClearInventoryMutationConfig":
            return S.ClearInventory(self.ent(m.entity), [self.res[r] for r in m.resource_ids])
        if n == "AttackMutationConfig":
            return S.Attack(self.res[m.weapon_resource], self.res[m.armor_resource], self.res[m.health_resource],
                            int(m.damage_multiplier_pct))
        if n == "StatsMutationConfig":
            return S.SetStat(m.stat_name, self.value(m.source), "game" if m.target == C.StatsTarget.game else "agent",
                             self.ent(m.entity))
        if n == "AddTagMutationConfig":
            return S.AddTag(self.ent(m.entity), self.tags[m.tag_id])
        if n == "RemoveTagMutationConfig":
            return S.RemoveTag(self.ent(m.entity), self.tags[m.tag_id])
        if n == "RemoveTagsWithPrefixMutationConfig":
            return S.RemoveTagsWithPrefix(self.ent(m.entity), "", self.tag_names(m.tag_ids))
        if n == "ChangeVibeMutationConfig":
            return S.ChangeVibe(self.ent(m.entity), self.vibes[m.vibe_id])
        if n == "GameValueMutationConfig":
            return S.GameValueMutation(self.value(m.value), self.value(m.source), self.ent(m.target))
        if n == "RecomputeMaterializedQueryMutationConfig":
            return S.RecomputeMaterializedQuery(self.tags[m.tag_id])
        if n == "QueryInventoryMutationConfig":
            return S.QueryInventoryMutation(self.query(m.query), {self.res[r]: int(dl) for r, dl in m.deltas},
                                            self.ent(m.source) if m.has_source else None,
                                            {self.res[r]: name for r, name in m.transfer_stat_names})
        if n == "SpawnObjectMutationConfig":
            return S.SpawnObject(m.object_type)
        if n == "RaycastSpawnMutationConfig":
            return S.RaycastSpawn(m.object_type, [tuple(d) for d in m.directions], self.value(m.max_range),
                                  self.filters(m.blocker))
        simple = {"RelocateMutationConfig": S.Relocate, "SwapMutationConfig": S.Swap, "UseTargetMutationConfig": S.UseTarget,
                  "PushObjectMutationConfig": S.PushObject}
        if n in simple:
            return simple[n]()
        raise ReferenceConfigError(f"unknown mutation config {n}")

    # ---- handlers --------------------------------------------------------------------------------------------------
    def leaf(self, hc) -> S.Handler:
        return S.Handler(self.filters(hc.filters), self.mutations(hc.mutations), hc.name or "h")

    def handler(self, h):
        if h is None:
            return None
        n = type(h).__name__
        if n == "MultiHandler":
            kids = [self.handler(k) for k in h.handlers]
            return S.FirstMatch(kids) if h.mode == C.HandlerMode.FirstMatch else S.AllOf(kids)
        if n == "Handler":
            return self.leaf(h.config)
        if n == "HandlerConfig":
            return self.leaf(h)
        raise ReferenceConfigError(f"unknown handler object {n}")

    def tag_handlers(self, pairs) -> dict:
        """[(tag_id, HandlerConfig)] -> {tuple of tag names: Handler}; the converter registers ONE handler config under
        every tag its prefix matches (mettagrid_c_config.py:448-469), so pairs sharing a config are one entry."""
        out, groups = {}, {}
        for tag_id, hc in pairs:
            groups.setdefault(id(hc), (hc, []))[1].append(self.tags[tag_id])
        for hc, names in groups.values():
            out[tuple(names)] = self.leaf(hc)
        return out

    def aoes(self, configs) -> list:
        return [S.AOESpec(int(a.radius), bool(a.is_static), bool(a.effect_self), self.filters(a.filters),
                          self.mutations(a.mutations), {self.res[p.resource_id]: int(p.delta) for p in a.presence_deltas})
                for a in configs]

    def controls(self, configs, terr_names) -> list:
        return [S.TerritoryControl(terr_names[c.territory_index], int(c.strength), int(c.decay)) for c in configs]

    def inventory(self, oc) -> S.Inventory:
        """Every LimitDef explicitly, in the converter's order: nothing is left to the compiler's default-limit rule."""
        limits = [S.Limit([self.res[r] for r in ld.resources], int(ld.min_limit), int(ld.max_limit),
                          {self.res[r]: int(b) for r, b in ld.modifiers.items()}) for ld in oc.inventory_config.limit_defs]
        return S.Inventory({self.res[r]: int(a) for r, a in oc.initial_inventory.items()}, limits, 65535)

    # ---- top level -------------------------------------------------------------------------------------------------
    def game_spec(self) -> S.GameSpec:
        cfg = self.cfg
        terr_names = [f"territory_{i}" for i in range(len(cfg.territories))]
        type_tags = set()
        # agents: the per-agent cells "agent.<group>.<k>" in the converter's order; a one-agent team only has its aliases
        groups: dict = {}
        for cell, oc in cfg.objects.items():
            if isinstance(oc, C.AgentConfig):
                groups.setdefault(oc.group_id, {})[id(oc)] = oc
        agents, team_of_group = [], {}
        for gid in sorted(groups):
            members = list(groups[gid].values())
            gname = members[0].group_name
            team_of_group[gid] = _TEAM_IDS.get(gname, 100 + gid)
            for oc in members:
                names = self.tag_names(oc.tag_ids)
                type_tags.add(type_tag(oc.type_name))
                agents.append(S.AgentSpec(
                    team_id=team_of_group[gid], name=oc.type_name, tags=[t for t in names if t != type_tag(oc.type_name)],
                    vibe=int(oc.initial_vibe), inventory=self.inventory(oc),
                    rewards=[S.RewardSpec(self.value(e.reward), bool(e.accumulate)) for e in oc.reward_config.entries],
                    on_use=self.handler(oc.on_use_handler), on_tick=self.handler(oc.on_tick),
                    on_after_use=self.handler(oc.on_after_use_handler), aoes=self.aoes(oc.aoe_configs),
                    territory_controls=self.controls(oc.territory_controls, terr_names),
                    on_tag_add=self.tag_handlers(oc.on_tag_add), on_tag_remove=self.tag_handlers(oc.on_tag_remove)))
        if len(agents) != int(cfg.num_agents):
            raise ReferenceConfigError(f"config has {len(agents)} agent configs but num_agents={cfg.num_agents}")
        objects = {}
        for cell, oc in cfg.objects.items():
            if isinstance(oc, C.AgentConfig):
                continue
            names = self.tag_names(oc.tag_ids)
            is_wall = isinstance(oc, C.WallConfig)
            # Every non-wall GridObject of the reference owns an (unlimited by default) inventory that handlers may fill
            # whether or not the config mentions it, so none of them is treated as immutable here.
            has_inv = True
            objects[cell] = S.ObjectSpec(
                name=oc.type_name, map_name=cell, kind="wall" if is_wall else "object",
                tags=[t for t in names if t != type_tag(oc.type_name)], vibe=int(oc.initial_vibe),
                inventory=self.inventory(oc) if (has_inv and not is_wall) else None,
                on_use=self.handler(oc.on_use_handler), aoes=self.aoes(oc.aoe_configs),
                territory_controls=self.controls(oc.territory_controls, terr_names),
                on_tag_add=self.tag_handlers(oc.on_tag_add), on_tag_remove=self.tag_handlers(oc.on_tag_remove))
        g = cfg.global_obs
        values = {}
        for ov in g.obs:
            if ov.feature_id not in self.feature_name:
                raise ReferenceConfigError(f"global obs value uses feature id {ov.feature_id} that feature_ids does not name")
            values[self.feature_name[ov.feature_id]] = self.value(ov.value)
        move = cfg.actions["move"]
        n_vibes = int(cfg.actions["change_vibe"].number_of_vibes)
        events = {name: S.EventSpec(self.query(ev.target_query), [int(t) for t in ev.timesteps], self.filters(ev.filters),
                                    self.mutations(ev.mutations), None if int(ev.max_targets) < 0 else int(ev.max_targets),
                                    ev.fallback or None)
                  for name, ev in cfg.events.items()}
        territories = {terr_names[i]: S.TerritorySpec("", tags=self.tag_names(t.tag_prefix_ids),
                                                      on_enter=[self.leaf(h) for h in t.on_enter],
                                                      on_exit=[self.leaf(h) for h in t.on_exit],
                                                      presence=[self.leaf(h) for h in t.presence])
                       for i, t in enumerate(cfg.territories)}
        all_tag_names = [self.tags[i] for i in sorted(self.tags)]
        return S.GameSpec(
            resource_names=self.res, agents=agents, objects=objects,
            vibe_names=(self.vibes[:n_vibes] if n_vibes > 0 else (self.vibes or ["default"])), change_vibe_enabled=n_vibes > 0,
            move_directions=list(move.allowed_directions), move_handlers=[self.leaf(h) for h in move.handlers],
            tags=all_tag_names,
            obs=S.ObsSpec(width=int(cfg.obs_width), height=int(cfg.obs_height), num_tokens=int(cfg.num_observation_tokens),
                          token_value_base=int(cfg.token_value_base), episode_completion_pct=bool(g.episode_completion_pct),
                          last_action=bool(g.last_action), last_action_move=bool(g.last_action_move),
                          last_reward=bool(g.last_reward), local_position=bool(g.local_position), values=values,
                          aoe_mask="aoe_mask" in cfg.feature_ids),
            max_steps=int(cfg.max_steps), episode_truncates=bool(cfg.episode_truncates),
            protocol_details_obs=bool(cfg.protocol_details_obs), events=events,
            materialize_queries=[S.MaterializedQuery(self.tags[mq.tag_id], self.query(mq.query))
                                 for mq in cfg.materialized_queries],
            territories=territories, on_tick=self.handler(cfg.on_tick))


def game_spec(cfg) -> S.GameSpec:
    return _Lowering(cfg).game_spec()


def check_ids(cfg, prog: Program) -> None:
    """The ids the compiler assigned against the ids the reference assigned."""
    want_tags = {int(i): n for i, n in cfg.tag_id_map.items()}
    have_tags = dict(enumerate(prog.tag_names))
    if want_tags != have_tags:
        raise ReferenceConfigError(f"tag ids differ: reference {want_tags} vs engine {have_tags}")
    for cell, oc in cfg.objects.items():
        if oc.type_name not in prog.type_names or prog.type_names.index(oc.type_name) != int(oc.type_id):
            raise ReferenceConfigError(f"type id of '{oc.type_name}' differs: reference {oc.type_id} vs engine "
                                       f"{prog.type_names.index(oc.type_name) if oc.type_name in prog.type_names else None}")
    for name, fid in prog.feature_ids.items():
        if name in cfg.feature_ids and int(cfg.feature_ids[name]) != fid:
            raise ReferenceConfigError(f"feature id of '{name}' differs: reference {cfg.feature_ids[name]} vs engine {fid}")
    missing = [n for n in prog.feature_ids if n not in cfg.feature_ids and not n.startswith("lp:")]
    if missing:
        raise ReferenceConfigError(f"engine features the reference does not list: {missing}")


def compile_reference_config(cfg, height: int, width: int, max_objects=None) -> Program:
    prog = compile_spec(game_spec(cfg), height, width, max_objects=max_objects)
    check_ids(cfg, prog)
    return prog


class ReferenceMettaGrid:
    """``mettagrid.mettagrid_c.MettaGrid(env_cfg, map, seed)`` (cpp/bindings/mettagrid_py.cpp:242-316) on the MI355X
    engine: ``env_cfg`` is the C++ config tree the reference's converter produced from ``mettagrid_amd.mettagrid_c``
    classes, ``map`` the list-of-lists of cell names after ``rename_map_agents``."""

    def __new__(cls, env_cfg, map, seed: int, device: int = 0):  # noqa: A002 - reference argument name
        from .engine import MettaGrid
        height, width = len(map), len(map[0])
        prog = compile_reference_config(env_cfg, height, width)
        mg = MettaGrid(prog, np.asarray(map, dtype=object), seed, device=device)
        # object_type_names is indexed by type id and sized by the number of config entries (mettagrid_c.cpp:204-218)
        mg.object_type_names = list(prog.type_names) + [""] * max(0, len(env_cfg.objects) - len(prog.type_names))
        return mg
