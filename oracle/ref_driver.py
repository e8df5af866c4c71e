"""TEST INFRASTRUCTURE ONLY — drives the REAL reference C++ engine (oracle/_ref/mettagrid_c*.so).

Nothing in the product path (mettagrid_amd/, bench.py's timed region) may import this module.  It lowers a
``mettagrid_amd.spec.GameSpec`` into the reference's pybind11 config classes exactly as the reference's own
converter does (/root/reference/python/src/mettagrid/config/mettagrid_c_config.py:576-1007), constructs
``mettagrid_c.MettaGrid(config, map, seed)`` (cpp/bindings/mettagrid_py.cpp:242-272) and exposes step/trace helpers.
The .so is built from the reference's sources in place by ``make -C oracle ref``; it is never committed.
"""
from __future__ import annotations

import glob
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))

from mettagrid_amd import spec as S  # noqa: E402
from mettagrid_amd.compiler import compile_spec, type_tag  # noqa: E402

_mod = None


def ref_available() -> bool:
    return bool(glob.glob(os.path.join(_HERE, "_ref", "mettagrid_c*.so")))


def ref_module():
    global _mod
    if _mod is None:
        paths = glob.glob(os.path.join(_HERE, "_ref", "mettagrid_c*.so"))
        if not paths:
            raise RuntimeError("oracle/_ref/mettagrid_c*.so missing: run `make -C oracle ref` where /root/reference exists")
        loader_spec = importlib.util.spec_from_file_location("mettagrid_c", paths[0])
        _mod = importlib.util.module_from_spec(loader_spec)
        loader_spec.loader.exec_module(_mod)
    return _mod


class _Lower:
    def __init__(self, spec: S.GameSpec, prog) -> None:
        self.m = ref_module()
        self.spec = spec
        self.prog = prog
        self.res = {n: i for i, n in enumerate(spec.resource_names)}
        self.vibe = {n: i for i, n in enumerate(spec.vibe_names)}
        self.tag = {n: i for i, n in enumerate(prog.tag_names)}

    def ent(self, e):
        return self.m.EntityRef.actor if e == S.ACTOR else self.m.EntityRef.target

    def prefix(self, p):
        return [i for n, i in self.tag.items() if n.startswith(p)]

    # ---- game values ----
    def value(self, v):
        m = self.m
        if isinstance(v, S.InventoryValue):
            c = m.InventoryValueConfig()
            c.scope = m.GameValueScope.AGENT
            c.id = self.res[v.item]
            return c
        if isinstance(v, S.StatValue):
            c = m.StatValueConfig()
            c.scope = m.GameValueScope.AGENT if v.scope == "agent" else m.GameValueScope.GAME
            c.stat_name = v.name
            c.delta = bool(getattr(v, "delta", False))
            return c
        if isinstance(v, S.ConstValue):
            c = m.ConstValueConfig()
            c.value = float(v.value)
            return c
        if isinstance(v, (int, float)):
            c = m.ConstValueConfig()
            c.value = float(v)
            return c
        if isinstance(v, S.SumValue):
            c = m.SumValueConfig()
            c.values = [self.value(x) for x in v.values]
            c.weights = [float(w) for w in (v.weights or [])]
            c.log = bool(v.log)
            return c
        if isinstance(v, S.RatioValue):
            c = m.RatioValueConfig()
            c.numerator = self.value(v.numerator)
            c.denominator = self.value(v.denominator)
            return c
        if isinstance(v, S.MaxValue):
            c = m.MaxValueConfig()
            c.values = [self.value(x) for x in v.values]
            return c
        if isinstance(v, S.MinValue):
            c = m.MinValueConfig()
            c.values = [self.value(x) for x in v.values]
            return c
        if isinstance(v, S.QueryInventoryValue):
            c = m.QueryInventoryValueConfig()
            c.id = self.res[v.item]
            c.set_query(self.query(v.query))
            return c
        if isinstance(v, S.QueryCountValue):
            c = m.QueryCountValueConfig()
            c.set_query(self.query(v.query))
            return c
        raise TypeError(v)

    # ---- queries ----
    def query(self, q):
        m = self.m
        if isinstance(q, str):
            q = S.TagQuery(q)
        if isinstance(q, S.MaterializedQuery):
            q = S.TagQuery(q.tag)
        if isinstance(q, S.TagQuery):
            c = m.TagQueryConfig()
            c.tag_id = self.tag[q.tag]
            for f in q.filters:
                self.add_filter(c, f)
        elif isinstance(q, S.FilteredQuery):
            c = m.FilteredQueryConfig()
            c.set_source(self.query(q.source))
            for f in q.filters:
                self.add_filter(c, f)
        elif isinstance(q, S.ClosureQuery):
            c = m.ClosureQueryConfig()
            c.set_source(self.query(q.source))
            if q.candidates is not None:
                c.set_candidates(self.query(q.candidates))
            for f in q.edge_filters:
                self.add_filter(c, f, "edge_")
            for f in q.result_filters:
                self.add_filter(c, f, "result_")
        elif isinstance(q, S.RaycastQuery):
            c = m.RaycastQueryConfig()
            c.set_source(self.query(q.source))
            c.max_range = self.value(q.max_range)
            c.directions = [tuple(d) for d in q.directions]
            c.include_blocker = bool(q.include_blocker)
            for f in q.blocker:
                self.add_filter(c, f, "blocker_")
        else:
            raise TypeError(q)
        if q.max_items is not None:
            c.max_items = self.value(q.max_items)
        if q.order_by == "random":
            c.order_by = m.QueryOrderBy.random
        return m.make_query_config(c)

    # ---- filters ----
    def add_filter(self, holder, f, prefix=""):
        m = self.m

        def add(kind, cfg):
            getattr(holder, f"add_{prefix}{kind}_filter")(cfg)

        if isinstance(f, S.VibeFilter):
            if f.vibe in self.vibe:
                add("vibe", m.VibeFilterConfig(self.ent(f.entity), self.vibe[f.vibe]))
        elif isinstance(f, S.ResourceFilter):
            add("resource", m.ResourceFilterConfig(self.ent(f.entity), self.res[f.resource], f.min_amount))
        elif isinstance(f, S.SharedTagPrefixFilter):
            add("shared_tag_prefix", m.SharedTagPrefixFilterConfig(self.prefix(f.prefix)))
        elif isinstance(f, S.TagPrefixFilter):
            add("tag_prefix", m.TagPrefixFilterConfig(self.ent(f.entity), self.prefix(f.prefix)))
        elif isinstance(f, S.NegFilter):
            n = m.NegFilterConfig()
            for g in f.inner:
                self.add_filter(n, g)
            add("neg", n)
        elif isinstance(f, S.OrFilter):
            o = m.OrFilterConfig()
            for g in f.inner:
                self.add_filter(o, g)
            add("or", o)
        elif isinstance(f, S.TargetLocEmptyFilter):
            add("target_loc_empty", m.TargetLocEmptyFilterConfig())
        elif isinstance(f, S.TargetIsUsableFilter):
            add("target_is_usable", m.TargetIsUsableFilterConfig())
        elif isinstance(f, S.PeriodicFilter):
            add("periodic", m.PeriodicFilterConfig(f.period, f.period if f.start_on is None else f.start_on))
        elif isinstance(f, S.GameValueFilter):
            add("game_value", m.GameValueFilterConfig(self.value(f.value), self.value(f.threshold), self.ent(f.entity)))
        elif isinstance(f, S.MaxDistanceFilter):
            c = m.MaxDistanceFilterConfig()
            c.entity = self.ent(f.entity)
            c.radius = f.radius
            if f.source is not None:
                c.set_source(self.query(f.source))
            add("max_distance", c)
        elif isinstance(f, S.QueryResourceFilter):
            c = m.QueryResourceFilterConfig()
            c.requirements = [(self.res[r], v) for r, v in f.requirements.items()]
            c.set_query(self.query(f.query))
            add("query_resource", c)
        else:
            raise TypeError(f)

    # ---- mutations ----
    def add_mutation(self, hc, mu):
        m = self.m
        if isinstance(mu, S.ResourceDelta):
            hc.add_resource_delta_mutation(m.ResourceDeltaMutationConfig(self.ent(mu.entity), self.res[mu.resource], mu.delta))
        elif isinstance(mu, S.ResourceTransfer):
            hc.add_resource_transfer_mutation(m.ResourceTransferMutationConfig(
                self.ent(mu.source), self.ent(mu.destination), self.res[mu.resource], mu.amount,
                mu.remove_source_when_empty))
        elif isinstance(mu, S.ClearInventory):
            hc.add_clear_inventory_mutation(m.ClearInventoryMutationConfig(
                self.ent(mu.entity), [self.res[r] for r in mu.resources]))
        elif isinstance(mu, S.Attack):
            hc.add_attack_mutation(m.AttackMutationConfig(
                self.res[mu.weapon], self.res[mu.armor], self.res[mu.health], mu.damage_multiplier_pct))
        elif isinstance(mu, S.SetStat):
            c = m.StatsMutationConfig(mu.name, m.StatsTarget.game if mu.scope == "game" else m.StatsTarget.agent,
                                      m.StatsEntity.actor if mu.entity == S.ACTOR else m.StatsEntity.target)
            c.source = self.value(mu.value)
            hc.add_stats_mutation(c)
        elif isinstance(mu, S.ChangeVibe):
            hc.add_change_vibe_mutation(m.ChangeVibeMutationConfig(self.ent(mu.entity), self.vibe[mu.vibe]))
        elif isinstance(mu, S.AddTag):
            hc.add_add_tag_mutation(m.AddTagMutationConfig(self.ent(mu.entity), self.tag[mu.tag]))
        elif isinstance(mu, S.RemoveTag):
            hc.add_remove_tag_mutation(m.RemoveTagMutationConfig(self.ent(mu.entity), self.tag[mu.tag]))
        elif isinstance(mu, S.RemoveTagsWithPrefix):
            hc.add_remove_tags_with_prefix_mutation(
                m.RemoveTagsWithPrefixMutationConfig(self.ent(mu.entity), self.prefix(mu.prefix)))
        elif isinstance(mu, S.GameValueMutation):
            hc.add_game_value_mutation(m.GameValueMutationConfig(self.value(mu.value), self.ent(mu.target), self.value(mu.source)))
        elif isinstance(mu, S.RecomputeMaterializedQuery):
            c = m.RecomputeMaterializedQueryMutationConfig()
            c.tag_id = self.tag[mu.tag]
            hc.add_recompute_materialized_query_mutation(c)
        elif isinstance(mu, S.QueryInventoryMutation):
            c = m.QueryInventoryMutationConfig()
            c.deltas = [(self.res[r], d) for r, d in mu.deltas.items()]
            if mu.source is not None:
                c.source = self.ent(mu.source)
                c.has_source = True
            c.transfer_stat_names = [(self.res[r], n) for r, n in mu.transfer_stat_names.items()]
            c.set_query(self.query(mu.query))
            hc.add_query_inventory_mutation(c)
        elif isinstance(mu, S.PushObject):
            hc.add_push_object_mutation(m.PushObjectMutationConfig())
        elif isinstance(mu, S.SpawnObject):
            c = m.SpawnObjectMutationConfig()
            c.object_type = mu.object_type
            hc.add_spawn_object_mutation(c)
        elif isinstance(mu, S.RaycastSpawn):
            c = m.RaycastSpawnMutationConfig()
            c.object_type = mu.object_type
            c.directions = [tuple(d) for d in mu.directions]
            c.max_range = self.value(mu.max_range)
            for f in mu.blocker:
                self.add_filter(c, f, "blocker_")
            hc.add_raycast_spawn_mutation(c)
        elif isinstance(mu, S.Relocate):
            hc.add_relocate_mutation(m.RelocateMutationConfig())
        elif isinstance(mu, S.Swap):
            hc.add_swap_mutation(m.SwapMutationConfig())
        elif isinstance(mu, S.UseTarget):
            hc.add_use_target_mutation(m.UseTargetMutationConfig())
        else:
            raise TypeError(mu)

    def handler_config(self, h: S.Handler):
        hc = self.m.HandlerConfig(h.name or "h")
        for f in h.filters:
            self.add_filter(hc, f)
        for mu in h.mutations:
            self.add_mutation(hc, mu)
        return hc

    def handler(self, h):
        m = self.m
        if h is None:
            return None
        if isinstance(h, S.Handler):
            return m.Handler(self.handler_config(h))
        kids = [k for k in (self.handler(c) for c in h.handlers) if k is not None]
        if not kids:
            return None
        mode = m.HandlerMode.FirstMatch if isinstance(h, S.FirstMatch) else m.HandlerMode.All
        return m.MultiHandler(kids, mode)

    def object_extras(self, cfg, src):
        m = self.m
        if src.aoes:
            lst = []
            for a in src.aoes:
                c = m.AOEConfig()
                c.name = "aoe"
                c.radius, c.is_static, c.effect_self = a.radius, a.is_static, a.effect_self
                for f in a.filters:
                    self.add_filter(c, f)
                for mu in a.mutations:
                    self.add_mutation(c, mu)
                c.presence_deltas = [m.ResourceDelta(self.res[r], d) for r, d in a.presence_deltas.items()]
                lst.append(c)
            cfg.aoe_configs = lst
        if src.territory_controls:
            names = list(self.spec.territories.keys())
            lst = []
            for tc in src.territory_controls:
                c = m.TerritoryControlConfig()
                c.strength, c.decay, c.territory_index = tc.strength, tc.decay, names.index(tc.territory)
                lst.append(c)
            cfg.territory_controls = lst
        for mapping, adder in ((src.on_tag_add, cfg.add_on_tag_add_handler), (src.on_tag_remove, cfg.add_on_tag_remove_handler)):
            for prefix, h in mapping.items():
                hc = self.handler_config(S.Handler(h.filters, h.mutations, prefix))
                for t in self.prefix(prefix):
                    adder(t, hc)

    # ---- inventory ----
    def limit_defs_agent(self, a: S.AgentSpec, default_limit: int):
        m = self.m
        defs, configured = [], set()
        for lim in a.inventory.limits:
            mods = {self.res[n]: b for n, b in lim.modifiers.items() if n in self.res}
            defs.append(m.LimitDef([self.res[n] for n in lim.resources], lim.base, lim.max, mods))
            configured.update(lim.resources)
        for rn in self.spec.resource_names:
            if rn not in configured:
                defs.append(m.LimitDef([self.res[rn]], default_limit))
        return defs

    def game_config(self):
        m, sp, prog = self.m, self.spec, self.prog
        objects = {}
        teams: dict = {}
        for a in sp.agents:
            teams.setdefault(a.team_id, []).append(a)
        default_limit = sp.agents[0].inventory.default_limit
        for gid, (team_id, members) in enumerate(teams.items()):
            gname = S.TEAM_NAMES.get(team_id, f"group_{gid}")
            first = None
            for idx, a in enumerate(members):
                inv = m.InventoryConfig()
                inv.limit_defs = self.limit_defs_agent(a, default_limit)
                rc = m.RewardConfig()
                entries = []
                for rw in a.rewards:
                    e = m.RewardEntry()
                    e.reward = self.value(rw.value)
                    e.accumulate = bool(rw.per_tick)
                    entries.append(e)
                rc.entries = entries
                cfg = m.AgentConfig(
                    type_id=prog.type_names.index(a.name), type_name=a.name, group_id=gid, group_name=gname,
                    initial_vibe=a.vibe, inventory_config=inv, reward_config=rc,
                    initial_inventory={self.res[k]: v for k, v in a.inventory.initial.items()})
                cfg.tag_ids = [self.tag[t] for t in list(a.tags) + [type_tag(a.name)]]
                cfg.on_tick = self.handler(a.on_tick)
                cfg.on_use_handler = self.handler(a.on_use)
                cfg.on_after_use_handler = self.handler(a.on_after_use)
                self.object_extras(cfg, a)
                objects[f"agent.{gname}.{idx}"] = cfg
                first = first or cfg
            aliases = [f"agent.{gname}", f"agent.team_{gid}"]
            if gid == 0:
                aliases += ["agent.default", "agent.agent"]
            for al in aliases:
                objects[al] = first
        for _key, o in sp.objects.items():
            tid = prog.type_names.index(o.name)
            if o.kind == "wall":
                cfg = m.WallConfig(type_id=tid, type_name=o.name, initial_vibe=o.vibe)
            else:
                cfg = m.GridObjectConfig(type_id=tid, type_name=o.name, initial_vibe=o.vibe)
                if o.inventory is not None:
                    if o.inventory.initial:
                        cfg.initial_inventory = {self.res[k]: v for k, v in o.inventory.initial.items() if k in self.res}
                    defs, configured = [], set()
                    for lim in o.inventory.limits:
                        rids = [self.res[n] for n in lim.resources if n in self.res]
                        configured.update(lim.resources)
                        if rids:
                            mods = {self.res[n]: b for n, b in lim.modifiers.items() if n in self.res}
                            defs.append(m.LimitDef(rids, lim.base, lim.max, mods))
                    for rn in o.inventory.initial:
                        if rn not in configured and rn in self.res:
                            defs.append(m.LimitDef([self.res[rn]], o.inventory.default_limit))
                    if defs:
                        inv = m.InventoryConfig()
                        inv.limit_defs = defs
                        cfg.inventory_config = inv
                cfg.on_use_handler = self.handler(o.on_use)
            cfg.tag_ids = [self.tag[t] for t in list(o.tags) + [type_tag(o.name)]]
            self.object_extras(cfg, o)
            objects[o.cell] = cfg

        obs = sp.obs
        obs_cpp = []
        for name, v in obs.values.items():
            oc = m.ObsValueConfig()
            oc.value = self.value(v)
            oc.feature_id = prog.feature_ids[name]
            obs_cpp.append(oc)
        gobs = m.GlobalObsConfig(episode_completion_pct=obs.episode_completion_pct, last_action=obs.last_action,
                                 last_action_move=obs.last_action_move, last_reward=obs.last_reward, goal_obs=False,
                                 local_position=obs.local_position, obs=obs_cpp)
        move_kwargs = dict(allowed_directions=list(sp.move_directions), required_resources={}, consumed_resources={})
        if sp.move_handlers:
            move_kwargs["handlers"] = [self.handler_config(h) for h in sp.move_handlers]
        actions = {
            "noop": m.ActionConfig(required_resources={}, consumed_resources={}),
            "move": m.MoveActionConfig(**move_kwargs),
            "attack": m.AttackActionConfig(required_resources={}, consumed_resources={}, defense_resources={},
                                           armor_resources={}, weapon_resources={},
                                           success=m.AttackOutcome({}, {}, []), enabled=False, vibes=[], vibe_bonus={}),
            "change_vibe": m.ChangeVibeActionConfig(
                required_resources={}, consumed_resources={},
                number_of_vibes=len(sp.vibe_names) if sp.change_vibe_enabled else 0),
        }
        extra = {}
        if sp.territories:
            terr = []
            for _name, tc in sp.territories.items():
                c = m.TerritoryConfig()
                c.tag_prefix_ids = self.prefix(tc.tag_prefix)
                c.on_enter = [self.handler_config(S.Handler(h.filters, h.mutations, "on_enter")) for h in tc.on_enter]
                c.on_exit = [self.handler_config(S.Handler(h.filters, h.mutations, "on_exit")) for h in tc.on_exit]
                c.presence = [self.handler_config(S.Handler(h.filters, h.mutations, "presence")) for h in tc.presence]
                terr.append(c)
            extra["territories"] = terr
        if sp.events:
            evs = {}
            for name, ev in sp.events.items():
                c = m.EventConfig(name)
                c.timesteps = [int(t) for t in ev.timesteps]
                c.max_targets = -1 if ev.max_targets is None else int(ev.max_targets)
                c.fallback = ev.fallback or ""
                c.set_target_query(self.query(ev.target_query))
                for f in ev.filters:
                    self.add_filter(c, f)
                for mu in ev.mutations:
                    self.add_mutation(c, mu)
                evs[name] = c
            extra["events"] = evs
        if sp.materialize_queries:
            mqs = []
            for mq in sp.materialize_queries:
                c = m.MaterializedQueryTag()
                c.tag_id = self.tag[mq.tag]
                c.set_query(self.query(mq.query))
                mqs.append(c)
            extra["materialized_queries"] = mqs
        if sp.on_tick is not None:
            extra["on_tick"] = self.handler(sp.on_tick)
        return m.GameConfig(
            **extra,
            num_agents=len(sp.agents), max_steps=sp.max_steps, episode_truncates=sp.episode_truncates,
            obs_width=obs.width, obs_height=obs.height, resource_names=list(sp.resource_names),
            vibe_names=list(sp.vibe_names), num_observation_tokens=obs.num_tokens, global_obs=gobs,
            feature_ids=dict(prog.feature_ids), actions=actions, objects=objects,
            tag_id_map={i: n for i, n in enumerate(prog.tag_names)}, protocol_details_obs=sp.protocol_details_obs,
            token_value_base=obs.token_value_base)


class RefSim:
    """One reference env.  ``cells`` is the [H][W] list of cell-name strings (per-agent renaming applied here)."""

    def __init__(self, spec: S.GameSpec, cells, seed: int, prog=None) -> None:
        H, W = len(cells), len(cells[0])
        self.prog = prog or compile_spec(spec, H, W)
        self.cfg = _Lower(spec, self.prog).game_config()
        cmap = self.prog.class_map(cells)
        names = [["empty" if cmap[r, c] == 0 else self.prog.class_cells[cmap[r, c] - 1] for c in range(W)]
                 for r in range(H)]
        # keep the caller's alias for single-agent teams (the stat key "objects.<cell>" uses the map's spelling)
        for r in range(H):
            for c in range(W):
                cell = str(cells[r][c])
                if cmap[r, c] and cell in self.prog.cell_to_class:
                    names[r][c] = cell
        self.c = ref_module().MettaGrid(self.cfg, names, int(seed))
        self.A = self.c.actions()
        self.VA = self.c.vibe_actions()

    def step(self, actions, vibe_actions=None) -> None:
        self.A[:] = actions
        self.VA[:] = 0 if vibe_actions is None else vibe_actions
        self.c.step()

    def snapshot(self) -> dict:
        return dict(obs=np.array(self.c.observations()), rewards=np.array(self.c.rewards()),
                    terminals=np.array(self.c.terminals()), truncations=np.array(self.c.truncations()),
                    action_success=np.array(self.c.action_success(), dtype=bool),
                    episode_rewards=np.array(self.c.get_episode_rewards()))
