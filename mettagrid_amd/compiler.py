"""GameSpec -> flat int32 program (format: include/mgx_program.h).

Host-side counterpart of the reference's config converter
(/root/reference/python/src/mettagrid/config/mettagrid_c_config.py:576-1007) and of the C++ constructors that turn
configs into handler/filter/mutation objects (cpp/src/mettagrid/handler/handler.cpp:56-74,
cpp/src/mettagrid/actions/action_handler_factory.cpp:15-79).  Everything name-based (resources, tags, vibes, stat
keys, feature ids) is resolved here; the device only sees integers.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field

import numpy as np

from . import spec as S
from .fmt import K
from .umap import UMap, from_pydict

ORIENTATIONS = {  # cpp/include/mettagrid/actions/orientation.hpp:6-15
    "north": 0, "south": 1, "west": 2, "east": 3, "northwest": 4, "northeast": 5, "southwest": 6, "southeast": 7,
}


class UnsupportedFeature(ValueError):
    """The spec uses a reference feature the engine does not implement yet (see DESIGN.md scope table)."""


def f32_bits(x: float) -> int:
    return struct.unpack("<i", struct.pack("<f", float(x)))[0]


def type_tag(name: str) -> str:  # python/src/mettagrid/config/tag.py:8-13
    return f"type:{name}"


def num_tokens_needed(max_value: int, base: int) -> int:  # systems/observation_encoder.hpp:89-104
    if max_value == 0:
        return 1
    n = 0
    while max_value > 0:
        max_value //= base
        n += 1
    return n


def observation_offsets(obs_h: int, obs_w: int) -> list[tuple[int, int]]:
    """Manhattan-shell order masked to the reference's disc/ellipse.

    Restates PackedCoordinate::ObservationPattern (cpp/include/mettagrid/systems/packed_coordinate.hpp:87-156) and
    within_observation_shape (cpp/src/mettagrid/core/observation_shape.cpp:19-52).
    """
    row_r, col_r = obs_h >> 1, obs_w >> 1

    def inside(dr: int, dc: int) -> bool:
        if row_r == 0 and col_r == 0:
            return dr == 0 and dc == 0
        if row_r == 0:
            return dr == 0 and abs(dc) <= col_r
        if col_r == 0:
            return dc == 0 and abs(dr) <= row_r
        if row_r == col_r:
            d2 = dr * dr + dc * dc
            if d2 <= row_r * row_r:
                return True
            return row_r >= 2 and d2 == row_r * row_r + 1 and (abs(dr) == row_r or abs(dc) == col_r)
        return dr * dr * col_r * col_r + dc * dc * row_r * row_r <= row_r * row_r * col_r * col_r

    out = []
    row_min, row_max, col_min, col_max = -(obs_h // 2), obs_h // 2, -(obs_w // 2), obs_w // 2
    # C++ integer division truncates toward zero; -h/2 for positive h equals -(h//2).
    emitted, d = 0, 0
    while emitted < obs_h * obs_w:
        for dr in range(-d, d + 1):
            dc = d - abs(dr)
            for c in ([0] if dc == 0 else [-dc, dc]):
                if row_min <= dr <= row_max and col_min <= c <= col_max:
                    emitted += 1
                    if inside(dr, c):
                        out.append((dr, c))
        d += 1
        if d > obs_h + obs_w:
            break
    return out


class _Label:
    __slots__ = ("pc",)

    def __init__(self) -> None:
        self.pc = None


@dataclass
class Program:
    words: np.ndarray                  # int32 blob
    spec: S.GameSpec
    resource_names: list
    vibe_names: list
    tag_names: list                    # index = tag id
    type_names: list                   # index = type id (sorted)
    class_cells: list                  # index = class id -> canonical cell name
    cell_to_class: dict                # every accepted map cell name -> class id (aliases included)
    agent_rename: dict                 # group cell name -> [per-agent class ids] (rename_map_agents)
    agent_stat_names: list
    game_stat_names: list
    action_names: list
    feature_ids: dict                  # feature name -> id
    max_objects: int = 0
    feature_norms: list = field(default_factory=list)   # id -> the reference's normalisation constant (id_map.py)

    @property
    def num_agents(self) -> int:
        return int(self.words[K.H_NUM_AGENTS])

    @property
    def num_tokens(self) -> int:
        return int(self.words[K.H_NUM_TOKENS])

    def class_map(self, cells) -> np.ndarray:
        """Map of cell-name strings [H][W] -> uint16 class index + 1 (0 = empty), applying the reference's
        per-agent renaming (mettagrid_c_config.py:549-568): the k-th cell of a team names that team's k-th agent."""
        H, W = len(cells), len(cells[0])
        out = np.zeros((H, W), dtype=np.uint16)
        counters: dict = {}
        for r in range(H):
            for c in range(W):
                cell = str(cells[r][c])
                if cell in ("empty", ".", " "):  # cpp/bindings/mettagrid_c.cpp:228
                    continue
                if cell in self.agent_rename:
                    group = self.agent_rename[cell]
                    k = counters.get(id(group), 0)
                    if k >= len(group):
                        raise ValueError(f"Map has more '{cell}' cells than agents in the group ({len(group)})")
                    counters[id(group)] = k + 1
                    out[r, c] = group[k] + 1
                elif cell in self.cell_to_class:
                    out[r, c] = self.cell_to_class[cell] + 1
                else:
                    raise RuntimeError(f"Unknown object type: {cell}")  # mettagrid_c.cpp:232-234
        return out


class _Compiler:
    def __init__(self, spec: S.GameSpec, max_objects: int | None) -> None:
        self.spec = spec
        self.max_objects = max_objects
        self.sections: dict[int, list[int]] = {s: [] for s in range(K.SEC_COUNT)}
        self.counts: dict[int, int] = {s: 0 for s in range(K.SEC_COUNT)}
        self.header = [0] * K.H_WORDS
        self.atom_labels: list = []  # (word index in atoms section, label)
        self.indexed_tags: set = set()
        self.spawns = False
        self.removals = False
        self.pending_class_refs: list = []
        self.query_depth = 0
        self._qdepth = 0          # nesting level of the query currently being emitted (filters/values inherit it)
        self.dynamic_tags = False
        self.tag_mutations = False       # AddTag / RemoveTag / RemoveTagsWithPrefix somewhere in the program
        # Records emitted once per distinct content: the reference's converter builds one config per agent
        # (mettagrid_c_config.py:657-790), so 64 identical agents would otherwise bring 64 copies of every handler,
        # limit group and reward list — a program that no longer fits the world kernel's LDS copy or its caches.
        self._memo: dict = {}

    # ---- id tables -------------------------------------------------------------------------------------------
    def build_ids(self) -> None:
        sp = self.spec
        if len(sp.resource_names) > K.MAX_RESOURCES:
            raise UnsupportedFeature(f"at most {K.MAX_RESOURCES} resources are supported (got {len(sp.resource_names)})")
        if len(sp.agents) == 0 or len(sp.agents) > K.MAX_AGENTS:
            raise ValueError("need 1..256 agents")
        self.res_id = {n: i for i, n in enumerate(sp.resource_names)}
        vibes = list(sp.vibe_names)
        self.vibe_id = {n: i for i, n in enumerate(vibes)}
        type_names = {o.name for o in sp.objects.values()} | {a.name for a in sp.agents}
        self.type_names = sorted(type_names)                       # mettagrid_c_config.py:608-612
        self.type_id = {n: i for i, n in enumerate(self.type_names)}
        tags = set(sp.tags)
        for o in sp.objects.values():
            tags.update(o.tags)
            tags.add(type_tag(o.name))
        for a in sp.agents:
            tags.update(a.tags)
            tags.add(type_tag(a.name))
        tags.update(mq.tag for mq in sp.materialize_queries)
        self.tag_names = sorted(tags)                              # :629-646
        if len(self.tag_names) > 256:
            raise ValueError("Too many unique tags (max 256)")
        self.tag_id = {n: i for i, n in enumerate(self.tag_names)}

        # ---- stat tables (cpp/include/mettagrid/systems/stats_tracker.hpp; keys composed at runtime there) ----
        R = len(sp.resource_names)
        a: list[str] = ["action.failed", "action.noop.success", "action.noop.failed", "action.move.success",
                        "action.move.failed", "action.change_vibe.success", "action.change_vibe.failed",
                        "action.invalid_index"]
        self.stat_well_known: dict[int, int] = {
            K.S_ACTION_FAILED: 0, K.S_NOOP_SUCCESS: 1, K.S_NOOP_FAILED: 2, K.S_MOVE_SUCCESS: 3, K.S_MOVE_FAILED: 4,
            K.S_VIBE_SUCCESS: 5, K.S_VIBE_FAILED: 6, K.S_INVALID_INDEX: 7,
        }
        # the per-tick bookkeeping stats directly behind the action counters: all of them share the first cache line
        # of an agent's stat row on the device (rows are 128-byte aligned there)
        for key, name in ((K.S_MAX_STEPS_WITHOUT_MOTION, "status.max_steps_without_motion"),
                          (K.S_SWAP, "actions.swap"), (K.S_DEATH, "death"), (K.S_CELL_VISITED, "cell.visited"),
                          (K.S_CELL_UNIQUE, "cell.unique_visited"), (K.S_CELL_MAXDIST, "cell.max_distance_from_spawn")):
            self.stat_well_known[key] = len(a)
            a.append(name)
        self.stat_well_known[K.S_INVALID_NEG_BASE] = len(a)
        a += [f"action.invalid_index.{k}" for k in range(-K.INVALID_WINDOW, 0)]
        self.stat_well_known[K.S_INVALID_POS_BASE] = len(a)
        self._invalid_pos_slot = len(a)
        a += [None] * K.INVALID_WINDOW  # names filled once n_actions is known
        for key, suffix in ((K.S_RES_AMOUNT_BASE, "amount"), (K.S_RES_GAINED_BASE, "gained"),
                            (K.S_RES_LOST_BASE, "lost"), (K.S_RES_DEPOSITED_BASE, "deposited")):
            self.stat_well_known[key] = len(a)
            a += [f"{r}.{suffix}" for r in sp.resource_names]
        self.agent_stats = a
        g = ["tokens_written", "tokens_dropped", "tokens_free_space"]
        self.game_stats = g
        self.stat_well_known[K.S_GAME_TOKENS_WRITTEN] = 0
        self.stat_well_known[K.S_GAME_TOKENS_DROPPED] = 1
        self.stat_well_known[K.S_GAME_TOKENS_FREE] = 2
        _ = R

    def stat(self, scope: str, name: str) -> int:
        table = self.agent_stats if scope == "agent" else self.game_stats
        if name in table:
            return table.index(name)
        table.append(name)
        return len(table) - 1

    # ---- emit helpers ----------------------------------------------------------------------------------------
    def emit(self, sec: int, rec: list[int]) -> int:
        idx = self.counts[sec]
        self.sections[sec].extend(x if isinstance(x, tuple) else int(x) for x in rec)
        self.counts[sec] += 1
        return idx

    def emit_words(self, words: list[int]) -> int:
        """Append raw words to the WORDLIST section; returns the start offset (in words)."""
        start = len(self.sections[K.SEC_WORDLIST])
        self.sections[K.SEC_WORDLIST].extend(int(x) for x in words)
        self.counts[K.SEC_WORDLIST] = len(self.sections[K.SEC_WORDLIST])
        return start

    def ent(self, e: str) -> int:
        if e == S.ACTOR:
            return K.ENT_ACTOR
        if e == S.TARGET:
            return K.ENT_TARGET
        raise ValueError(f"unknown entity ref {e!r}")

    def tag_mask_words(self, tag_ids) -> list[int]:
        w = [0] * K.TAG_WORDS
        for t in tag_ids:
            w[t >> 5] |= 1 << (t & 31)
        return [x - (1 << 32) if x >= (1 << 31) else x for x in w]

    def prefix_tags(self, prefix, tags=None) -> list[int]:  # mettagrid_c_config.py:74-75
        if tags is not None:          # explicit names (a config that already went through the reference's converter)
            return [self.tag_id[n] for n in tags]
        if isinstance(prefix, tuple):
            return [self.tag_id[n] for n in prefix]
        return [i for n, i in self.tag_id.items() if n.startswith(prefix)]

    # ---- game values -----------------------------------------------------------------------------------------
    def gv_code(self, v, code: list) -> None:
        if isinstance(v, S.InventoryValue):
            code.append([K.GOP_INVENTORY, self.res_id[v.item], 0, 0])
        elif isinstance(v, S.StatValue):
            scope = 0 if v.scope == "agent" else 1
            code.append([K.GOP_STAT, scope, self.stat(v.scope, v.name), 0])
        elif isinstance(v, S.ConstValue):
            code.append([K.GOP_CONST, f32_bits(v.value), 0, 0])
        elif isinstance(v, S.SumValue):
            if v.weights is not None and len(v.weights) != len(v.values):
                raise RuntimeError("SumValueConfig.weights size must match values size")  # game_value.cpp:70-72
            code.append([K.GOP_CONST, f32_bits(0.0), 0, 0])
            for i, child in enumerate(v.values):
                self.gv_code(child, code)
                has_w = 1 if v.weights else 0
                code.append([K.GOP_ADD_TERM, 1 if v.log else 0, has_w, f32_bits(v.weights[i]) if has_w else 0])
        elif isinstance(v, S.RatioValue):
            self.gv_code(v.numerator, code)
            self.gv_code(v.denominator, code)
            code.append([K.GOP_RATIO, 0, 0, 0])
        elif isinstance(v, (S.MaxValue, S.MinValue)):
            is_max = isinstance(v, S.MaxValue)
            if not v.values:
                code.append([K.GOP_CONST, f32_bits(0.0), 0, 0])
                return
            lowest = -3.4028234663852886e38 if is_max else 3.4028234663852886e38
            code.append([K.GOP_CONST, f32_bits(lowest), 0, 0])
            for child in v.values:
                self.gv_code(child, code)
                code.append([K.GOP_MAX2 if is_max else K.GOP_MIN2, 0, 0, 0])
        elif isinstance(v, S.QueryInventoryValue):
            code.append([K.GOP_QUERY_INVENTORY, self.res_id[v.item], self.query(v.query), 0])
        elif isinstance(v, S.QueryCountValue):
            code.append([K.GOP_QUERY_COUNT, self.query(v.query), 0, 0])
        elif isinstance(v, (int, float)):
            code.append([K.GOP_CONST, f32_bits(float(v)), 0, 0])
        else:
            raise UnsupportedFeature(f"game value {type(v).__name__} is not supported yet")

    def gv_emit(self, v) -> tuple[int, int]:
        code: list = []
        self.gv_code(v, code)
        depth = peak = 0  # the device evaluates on an 8-entry register stack (MgxValueStack)
        for ins in code:
            depth += -1 if ins[0] in (K.GOP_ADD_TERM, K.GOP_RATIO, K.GOP_MAX2, K.GOP_MIN2) else 1
            peak = max(peak, depth)
        if peak > 8:
            raise UnsupportedFeature("game value expression nests deeper than 8 stack entries")
        start = self.counts[K.SEC_GV_CODE]
        for ins in code:
            self.emit(K.SEC_GV_CODE, ins)
        return start, len(code)

    def value_ref(self, v, feature: int = 0) -> int:
        """Index of a (start,count,feature) record in SEC_OBS_VALUES (generic value table)."""
        start, count = self.gv_emit(v)
        return self.emit(K.SEC_OBS_VALUES, [start, count, feature])

    # ---- queries (core/query_config.hpp; converter mettagrid_c_config.py:83-180) -------------------------------
    def query(self, q, depth: int | None = None) -> int:
        depth = self._qdepth + 1 if depth is None else depth
        prev = self._qdepth
        self._qdepth = depth
        try:
            return self._query(q, depth)
        finally:
            self._qdepth = prev

    def _query(self, q, depth: int) -> int:
        self.query_depth = max(self.query_depth, depth)
        if isinstance(q, str):
            q = S.TagQuery(q)
        if isinstance(q, S.MaterializedQuery):
            q = S.TagQuery(q.tag)
        mx = -1 if q.max_items is None else self.value_ref(q.max_items)
        order = 1 if q.order_by == "random" else 0
        if isinstance(q, S.TagQuery):
            if q.tag not in self.tag_id:
                raise ValueError(f"Tag query references unknown tag '{q.tag}'. Add it to GameSpec.tags or object tags.")
            self.indexed_tags.add(self.tag_id[q.tag])
            return self.emit(K.SEC_QUERIES, [K.QK_TAG, self.tag_id[q.tag], self.filters_emit(q.filters), 0, 0, mx, order, 0])
        if isinstance(q, S.FilteredQuery):
            src = self.query(q.source, depth + 1)
            return self.emit(K.SEC_QUERIES, [K.QK_FILTERED, src, self.filters_emit(q.filters), 0, 0, mx, order, 0])
        if isinstance(q, S.ClosureQuery):
            src = self.query(q.source, depth + 1)
            cand = -1 if q.candidates is None else self.query(q.candidates, depth + 1)
            return self.emit(K.SEC_QUERIES, [K.QK_CLOSURE, src, cand, self.filters_emit(q.edge_filters),
                                             self.filters_emit(q.result_filters), mx, order, 0])
        if isinstance(q, S.RaycastQuery):
            src = self.query(q.source, depth + 1)
            dirs = list(q.directions) or [(-1, 0), (1, 0), (0, 1), (0, -1)]  # query_system.cpp:265-266
            doff = self.emit_words([x for d in dirs for x in d])
            blocker = self.filters_emit([S.OrFilter(list(q.blocker))]) if q.blocker else K.PC_FAIL
            if q.include_blocker:
                order |= 0x100
            return self.emit(K.SEC_QUERIES, [K.QK_RAYCAST, src, self.value_ref(q.max_range), doff, len(dirs), mx, order, blocker])
        raise UnsupportedFeature(f"query {type(q).__name__} is not supported")

    # ---- filters -> short-circuit atoms ----------------------------------------------------------------------
    def _atom(self, op: int, a0: int, a1: int, a2: int, on_true, on_false) -> None:
        idx = self.emit(K.SEC_ATOMS, [op, a0, a1, a2, 0, 0, 0, 0])
        base = idx * K.AT_WORDS
        self.atom_labels.append((base + K.AT_ON_TRUE, on_true))
        self.atom_labels.append((base + K.AT_ON_FALSE, on_false))

    def _filter(self, f, on_true, on_false) -> None:
        if isinstance(f, S.NegFilter):          # filters/neg_filter.hpp:29-37: NOT(AND(inner))
            self._filter_and(f.inner, on_false, on_true)
        elif isinstance(f, S.OrFilter):         # filters/or_filter.hpp:21-28
            if not f.inner:
                self._atom(K.FOP_FALSE, 0, 0, 0, on_true, on_false)
                return
            for i, g in enumerate(f.inner):
                nxt = _Label() if i + 1 < len(f.inner) else None
                self._filter(g, on_true, nxt if nxt is not None else on_false)
                if nxt is not None:
                    nxt.pc = self.counts[K.SEC_ATOMS]
        elif isinstance(f, S.VibeFilter):
            if f.vibe not in self.vibe_id:      # unknown vibe names are dropped (mettagrid_c_config.py:222-236)
                self._atom(K.FOP_TRUE, 0, 0, 0, on_true, on_false)
            else:
                self._atom(K.FOP_VIBE, self.ent(f.entity), self.vibe_id[f.vibe], 0, on_true, on_false)
        elif isinstance(f, S.ResourceFilter):
            self._atom(K.FOP_RESOURCE, self.ent(f.entity), self.res_id[f.resource], f.min_amount, on_true, on_false)
        elif isinstance(f, S.SharedTagPrefixFilter):
            off = self.emit_words(self.tag_mask_words(self.prefix_tags(f.prefix, f.tags)))
            self._atom(K.FOP_SHARED_TAG, off, 0, 0, on_true, on_false)
        elif isinstance(f, S.TagPrefixFilter):
            off = self.emit_words(self.tag_mask_words(self.prefix_tags(f.prefix, f.tags)))
            self._atom(K.FOP_TAG, self.ent(f.entity), off, 0, on_true, on_false)
        elif isinstance(f, S.TargetLocEmptyFilter):
            self._atom(K.FOP_TARGET_LOC_EMPTY, 0, 0, 0, on_true, on_false)
        elif isinstance(f, S.TargetIsUsableFilter):
            self._atom(K.FOP_TARGET_IS_USABLE, 0, 0, 0, on_true, on_false)
        elif isinstance(f, S.PeriodicFilter):
            start_on = f.period if f.start_on is None else f.start_on  # mettagrid_c_config.py:302-310
            self._atom(K.FOP_PERIODIC, f.period, start_on, 0, on_true, on_false)
        elif isinstance(f, S.MaxDistanceFilter):
            qi = -1 if f.source is None else self.query(f.source)
            self._atom(K.FOP_MAX_DISTANCE, self.ent(f.entity), f.radius, qi, on_true, on_false)
        elif isinstance(f, S.QueryResourceFilter):
            pairs = [x for r, m in f.requirements.items() for x in (self.res_id[r], m)]
            off = self.emit_words(pairs) if pairs else 0
            self._atom(K.FOP_QUERY_RESOURCE, self.query(f.query), off, len(pairs) // 2, on_true, on_false)
        elif isinstance(f, S.GameValueFilter):
            self._atom(K.FOP_GAME_VALUE, self.ent(f.entity), self.value_ref(f.value), self.value_ref(f.threshold),
                       on_true, on_false)
        else:
            raise UnsupportedFeature(f"filter {type(f).__name__} is not supported yet")

    def _filter_and(self, filters, on_true, on_false) -> None:
        if not filters:
            self._atom(K.FOP_TRUE, 0, 0, 0, on_true, on_true)
            return
        for i, g in enumerate(filters):
            nxt = _Label() if i + 1 < len(filters) else None
            self._filter(g, nxt if nxt is not None else on_true, on_false)
            if nxt is not None:
                nxt.pc = self.counts[K.SEC_ATOMS]

    def filters_emit(self, filters) -> int:
        if not filters:
            return K.PC_PASS
        start = self.counts[K.SEC_ATOMS]
        self._filter_and(filters, K.PC_PASS, K.PC_FAIL)
        return start

    # ---- mutations -------------------------------------------------------------------------------------------
    def mutation(self, m) -> list[int]:
        if isinstance(m, S.ResourceDelta):
            return [K.MOP_RESOURCE_DELTA, self.ent(m.entity), self.res_id[m.resource], m.delta, 0, 0]
        if isinstance(m, S.ResourceTransfer):
            if m.remove_source_when_empty:
                self.removals = True
            return [K.MOP_RESOURCE_TRANSFER, self.ent(m.source), self.ent(m.destination), self.res_id[m.resource],
                    m.amount, 1 if m.remove_source_when_empty else 0]
        if isinstance(m, S.ClearInventory):
            ids = [self.res_id[r] for r in m.resources]
            off = self.emit_words(ids) if ids else 0
            return [K.MOP_CLEAR_INVENTORY, self.ent(m.entity), off, len(ids), 0, 0]
        if isinstance(m, S.Attack):
            return [K.MOP_ATTACK, self.res_id[m.weapon], self.res_id[m.armor], self.res_id[m.health],
                    m.damage_multiplier_pct, 0]
        if isinstance(m, S.SetStat):
            scope = 0 if m.scope == "game" else 1
            sid = self.stat("game" if scope == 0 else "agent", m.name)
            return [K.MOP_STATS, scope, self.ent(m.entity), sid, self.value_ref(m.value), 0]
        if isinstance(m, S.ChangeVibe):
            return [K.MOP_CHANGE_VIBE, self.ent(m.entity), self.vibe_id[m.vibe], 0, 0, 0]
        if isinstance(m, (S.AddTag, S.RemoveTag)):
            self.dynamic_tags = True
            self.tag_mutations = True
            op = K.MOP_ADD_TAG if isinstance(m, S.AddTag) else K.MOP_REMOVE_TAG
            return [op, self.ent(m.entity), self.tag_id[m.tag], 0, 0, 0]
        if isinstance(m, S.RemoveTagsWithPrefix):
            self.dynamic_tags = True
            self.tag_mutations = True
            ids = self.prefix_tags(m.prefix, m.tags)
            return [K.MOP_REMOVE_TAGS_PREFIX, self.ent(m.entity), self.emit_words(ids) if ids else 0, len(ids), 0, 0]
        if isinstance(m, S.GameValueMutation):
            if not isinstance(m.value, (S.InventoryValue, S.StatValue)):
                raise RuntimeError("Cannot update a read-only ResolvedGameValue")  # resolved_game_value.hpp:52-56
            return [K.MOP_GAME_VALUE, self.ent(m.target), self.value_ref(m.value), self.value_ref(m.source), 0, 0]
        if isinstance(m, S.RecomputeMaterializedQuery):
            return [K.MOP_RECOMPUTE_QUERY, self.tag_id[m.tag], 0, 0, 0, 0]
        if isinstance(m, S.QueryInventoryMutation):
            pairs = [x for r, dl in m.deltas.items() for x in (self.res_id[r], dl)]
            stats = [x for r, n in m.transfer_stat_names.items() for x in (self.res_id[r], self.stat("game", n))]
            return [K.MOP_QUERY_INVENTORY, self.query(m.query), self.emit_words(pairs) if pairs else 0, len(pairs) // 2,
                    -1 if m.source is None else self.ent(m.source), self.emit_words(stats) if stats else 0,
                    len(stats) // 2]
        if isinstance(m, S.PushObject):
            return [K.MOP_PUSH_OBJECT, 0, 0, 0, 0, 0]
        if isinstance(m, (S.SpawnObject, S.RaycastSpawn)):
            self.spawns = True
            self.dynamic_tags = True      # spawned objects register with the tag index at run time
            self.pending_class_refs.append(m.object_type)
            if isinstance(m, S.SpawnObject):
                return [K.MOP_SPAWN_OBJECT, ("class", m.object_type), 0, 0, 0, 0]
            dirs = [x for d in m.directions for x in d]
            blocker = self.filters_emit([S.OrFilter(list(m.blocker))]) if m.blocker else K.PC_FAIL
            return [K.MOP_RAYCAST_SPAWN, ("class", m.object_type), self.emit_words(dirs) if dirs else 0,
                    len(m.directions), self.value_ref(m.max_range), blocker]
        if isinstance(m, S.Relocate):
            return [K.MOP_RELOCATE, 0, 0, 0, 0, 0]
        if isinstance(m, S.Swap):
            return [K.MOP_SWAP, 0, 0, 0, 0, 0]
        if isinstance(m, S.UseTarget):
            return [K.MOP_USE_TARGET, 0, 0, 0, 0, 0]
        raise UnsupportedFeature(f"mutation {type(m).__name__} is not supported yet")

    def mutations_emit(self, mutations) -> tuple[int, int]:
        recs = [self.mutation(m) for m in mutations]
        start = self.counts[K.SEC_MUTS]
        for rec in recs:
            self.emit(K.SEC_MUTS, (rec + [0, 0])[:K.MU_WORDS])
        return start, len(recs)

    def tag_handlers(self, mapping: dict) -> tuple[int, int]:
        """on_tag_add / on_tag_remove: one (tag, leaf handler) record per tag matching the prefix
        (mettagrid_c_config.py:448-469)."""
        recs = []
        for prefix, h in mapping.items():
            ids = self.prefix_tags(prefix)
            if not ids:
                raise ValueError(f"on_tag handler prefix '{prefix}' matched no tags")
            hid = self.handler(S.Handler(h.filters, h.mutations, prefix))
            recs += [[t, hid] for t in ids]
        start = self.counts[K.SEC_TAG_HANDLERS]
        for r in recs:
            self.emit(K.SEC_TAG_HANDLERS, r)
        if recs:
            self.dynamic_tags = True
        return start, len(recs)

    def aoes(self, aoes: list) -> tuple[int, int]:
        key = ("aoes", repr(aoes))
        if key in self._memo:
            return self._memo[key]
        recs = []
        for a in aoes:
            fpc = self.filters_emit(a.filters)
            ms, mc = self.mutations_emit(a.mutations)
            ps = self.counts[K.SEC_PRESENCE]
            for r, dl in a.presence_deltas.items():
                self.emit(K.SEC_PRESENCE, [self.res_id[r], dl])
            recs.append([a.radius, 1 if a.is_static else 0, 1 if a.effect_self else 0, fpc, ms, mc, ps,
                         len(a.presence_deltas)])
        start = self.counts[K.SEC_AOES]
        for r in recs:
            self.emit(K.SEC_AOES, r)
        self._memo[key] = (start, len(recs))
        return start, len(recs)

    def terr_controls(self, controls: list) -> tuple[int, int]:
        names = list(self.spec.territories.keys())
        start = self.counts[K.SEC_TERR_CONTROLS]
        for tc in controls:
            if tc.territory not in names:
                raise ValueError(f"TerritoryControl references unknown territory '{tc.territory}'.")
            self.emit(K.SEC_TERR_CONTROLS, [tc.strength, tc.decay, names.index(tc.territory), 0])
        return start, len(controls)

    # ---- handlers --------------------------------------------------------------------------------------------
    def handler(self, h, fresh: bool = False) -> int:
        """Returns handler index or -1 (None / empty multi; mettagrid_c_config.py:405-427).  ``fresh``: always emit a
        new record (lists of consecutive leaf handlers)."""
        if h is None:
            return -1
        key = ("handler", repr(h))
        if key in self._memo and not fresh:
            return self._memo[key]
        hid = self._handler(h)
        self._memo[key] = hid
        return hid

    def _handler(self, h) -> int:
        if isinstance(h, S.Handler):
            fpc = self.filters_emit(h.filters)
            muts = [self.mutation(m) for m in h.mutations]
            mstart = self.counts[K.SEC_MUTS]
            for rec in muts:
                self.emit(K.SEC_MUTS, (rec + [0, 0])[:K.MU_WORDS])
            return self.emit(K.SEC_HANDLERS, [K.HK_LEAF, fpc, mstart, len(muts), 0, 0, 0, 0])
        if isinstance(h, (S.FirstMatch, S.AllOf)):
            kids = [k for k in (self.handler(c) for c in h.handlers) if k >= 0]
            if not kids:
                return -1
            cstart = self.counts[K.SEC_CHILDREN]
            for k in kids:
                self.emit(K.SEC_CHILDREN, [k])
            kind = K.HK_FIRST_MATCH if isinstance(h, S.FirstMatch) else K.HK_ALL
            return self.emit(K.SEC_HANDLERS, [kind, K.PC_PASS, 0, 0, cstart, len(kids), 0, 0])
        raise TypeError(f"Expected Handler, FirstMatch, or AllOf, got {type(h)}")

    # ---- inventory configs -----------------------------------------------------------------------------------
    def limits(self, limit_defs: list) -> tuple[int, int, list[int], int]:
        """limit_defs: list of (resource_ids, min, max, {item: bonus}).  Returns (start, count, res_limit, mod_mask).

        Restates Inventory::Inventory (cpp/src/mettagrid/objects/inventory.cpp:13-24): later defs overwrite the
        per-resource pointer; drop order follows the iteration order of the `_limits` unordered_map
        (inventory.cpp:141-173) which this computes with the libstdc++ emulator.
        """
        key = ("limits", repr(limit_defs))
        if key in self._memo:
            return self._memo[key]
        start = self.counts[K.SEC_LIMITS]
        owner: dict[int, int] = {}
        lim_map = UMap()
        for li, (rids, _mn, _mx, _mods) in enumerate(limit_defs):
            for r in rids:
                lim_map.insert(r)
                owner[r] = li
        res_limit = [-1] * K.MAX_RESOURCES
        mod_mask = 0
        for li, (rids, mn, mx, mods) in enumerate(limit_defs):
            reachable = [r for r in lim_map.keys() if owner[r] == li]
            mask = 0
            for r in reachable:
                mask |= 1 << r
                res_limit[r] = start + li
            mstart = self.counts[K.SEC_MODS]
            for item, bonus in mods.items():
                self.emit(K.SEC_MODS, [item, bonus])
                if reachable:
                    mod_mask |= 1 << item
            dstart = self.counts[K.SEC_DROP_ORDER]
            for r in reachable:
                self.emit(K.SEC_DROP_ORDER, [r])
            self.emit(K.SEC_LIMITS, [mask, mn, mx, mstart, len(mods), dstart, len(reachable)])
        self._memo[key] = (start, len(limit_defs), res_limit, mod_mask)
        return self._memo[key]

    def init_inventory(self, initial: dict, keep_zero: bool) -> tuple[int, int]:
        """(item, amount) list in the order the reference's constructors insert them: iteration order of the
        config-side unordered_map that pybind11 builds from the Python dict."""
        key = ("init_inv", repr(list(initial.items())), keep_zero)
        if key in self._memo:
            return self._memo[key]
        ids = [self.res_id[k] for k in initial if k in self.res_id]
        amounts = {self.res_id[k]: int(v) for k, v in initial.items() if k in self.res_id}
        order = from_pydict(ids).keys()
        start = self.counts[K.SEC_INIT_INV]
        n = 0
        for item in order:
            if amounts[item] > 0 or keep_zero:
                self.emit(K.SEC_INIT_INV, [item, amounts[item]])
                n += 1
        self._memo[key] = (start, n)
        return start, n

    def emit_class(self, *, kind, type_id, vibe, group, on_use, on_tick, on_after_use, lim, init_inv, rewards,
                   static, cell, tags, src=None) -> int:
        lstart, lcount, res_limit, mod_mask = lim
        rec = [0] * K.C_WORDS
        rec[K.C_KIND] = kind
        rec[K.C_TYPE_ID] = type_id
        rec[K.C_INITIAL_VIBE] = vibe
        rec[K.C_GROUP] = group
        rec[K.C_ON_USE] = on_use
        rec[K.C_ON_TICK] = on_tick
        rec[K.C_ON_AFTER_USE] = on_after_use
        rec[K.C_LIMIT_START], rec[K.C_LIMIT_COUNT] = lstart, lcount
        rec[K.C_INIT_INV_START], rec[K.C_INIT_INV_COUNT] = init_inv
        rec[K.C_REWARD_START], rec[K.C_REWARD_COUNT] = rewards
        rec[K.C_STATIC] = 1 if static else 0
        rec[K.C_OBJECTS_STAT] = self.stat("game", f"objects.{cell}")
        rec[K.C_MODIFIER_MASK] = mod_mask
        rec[K.C_TAGS:K.C_TAGS + K.TAG_WORDS] = self.tag_mask_words(tags)
        rec[K.C_RES_LIMIT:K.C_RES_LIMIT + K.MAX_RESOURCES] = res_limit
        if src is not None:
            rec[K.C_TAG_ADD_START], rec[K.C_TAG_ADD_COUNT] = self.tag_handlers(src.on_tag_add)
            rec[K.C_TAG_REMOVE_START], rec[K.C_TAG_REMOVE_COUNT] = self.tag_handlers(src.on_tag_remove)
            rec[K.C_AOE_START], rec[K.C_AOE_COUNT] = self.aoes(src.aoes)
            rec[K.C_TERR_START], rec[K.C_TERR_COUNT] = self.terr_controls(src.territory_controls)
            if src.aoes or src.territory_controls:
                rec[K.C_STATIC] = 0
        return self.emit(K.SEC_CLASSES, rec)

    def _matq_mutable_classes(self) -> set:
        """Classes whose objects a materialized query can tag: fixpoint over 'carries a leaf tag of the query'."""
        def leaves(q, out):
            if isinstance(q, str):
                out.add(q)
            elif isinstance(q, S.MaterializedQuery):
                out.add(q.tag)
            elif isinstance(q, S.TagQuery):
                out.add(q.tag)
            else:
                for attr in ("source", "candidates"):
                    sub = getattr(q, attr, None)
                    if sub is not None:
                        leaves(sub, out)
        cw = self.sections[K.SEC_CLASSES]
        n = self.counts[K.SEC_CLASSES]

        def has_raycast(q) -> bool:
            if isinstance(q, S.RaycastQuery):
                return True
            if isinstance(q, S.MaterializedQuery):
                return has_raycast(q.query)
            return any(has_raycast(getattr(q, attr)) for attr in ("source", "candidates") if getattr(q, attr, None) is not None
                       and not isinstance(getattr(q, attr), str))
        # A raycast returns whatever lies on its rays (blockers included), not a subset of its source objects
        # (query_config.hpp "collect objects on unblocked rays"): any class can then receive the tag.
        if any(has_raycast(mq.query) for mq in self.spec.materialize_queries):
            return set(range(n))

        def has(c, t):
            return (cw[c * K.C_WORDS + K.C_TAGS + (t >> 5)] >> (t & 31)) & 1

        mutable: set = set()
        given: dict = {}   # class -> tags a query may add
        changed = True
        while changed:
            changed = False
            for mq in self.spec.materialize_queries:
                lv: set = set()
                leaves(mq.query, lv)
                ids = [self.tag_id[t] for t in lv if t in self.tag_id]
                tid = self.tag_id[mq.tag]
                for c in range(n):
                    if any(has(c, t) or t in given.get(c, ()) for t in ids) and tid not in given.setdefault(c, set()):
                        given[c].add(tid)
                        mutable.add(c)
                        changed = True
        for c in range(n):  # a class that carries a materialized tag from the start can lose it
            if any(has(c, self.tag_id[mq.tag]) for mq in self.spec.materialize_queries):
                mutable.add(c)
        return mutable

    # ---- top level -------------------------------------------------------------------------------------------
    def compile(self) -> Program:
        sp = self.spec
        self.build_ids()
        R = len(sp.resource_names)
        obs = sp.obs
        if obs.width > 15 or obs.height > 15:  # mettagrid_c.cpp:63-68 (PackedCoordinate 4-bit coords)
            raise RuntimeError(f"Observation window size ({obs.width}x{obs.height}) exceeds maximum packable size")
        base = obs.token_value_base
        if base <= 1 or base > 256:
            raise RuntimeError("Base must be greater than 1")  # systems/observation_encoder.hpp:90-92
        digits = num_tokens_needed(65535, base)
        if digits > K.IF_WORDS:
            raise UnsupportedFeature("token_value_base too small")

        # ---- feature ids (python/src/mettagrid/config/id_map.py:161-235) ----
        # (name -> id, and the normalisation constant the reference attaches to each feature, id_map.py:161-235: what the
        # policy-side decoders divide token values by)
        feats: dict[str, int] = {}
        norms: list[float] = []

        def add_feature(name: str, normalization: float) -> None:
            feats[name] = len(feats)
            norms.append(float(normalization))

        for n, z in (("agent:group", 10.0), ("episode_completion_pct", 255.0), ("last_action", 10.0), ("last_reward", 100.0),
                     ("goal", 100.0), ("vibe", 255.0), ("tag", 10.0), ("lp:east", 255.0), ("lp:west", 255.0), ("lp:north", 255.0),
                     ("lp:south", 255.0), ("agent_id", 255.0)):
            add_feature(n, z)
        for r in sp.resource_names:
            add_feature(f"inv:{r}", float(base))
            for p in range(1, digits):
                add_feature(f"inv:{r}:p{p}", float(base))
        if sp.protocol_details_obs:
            for r in sp.resource_names:
                add_feature(f"protocol_input:{r}", 100.0)
            for r in sp.resource_names:
                add_feature(f"protocol_output:{r}", 100.0)
        for name in obs.values:
            add_feature(name, float(base))
            for p in range(1, digits):
                add_feature(f"{name}:p{p}", float(base))
        if obs.aoe_mask:
            add_feature("aoe_mask", 3.0)
        if obs.last_action_move:
            add_feature("last_action_move", 1.0)
        self.feature_norms = norms
        if len(feats) > 255:
            raise ValueError("too many observation features")
        self.feats = feats

        # ---- actions (action_handler_factory.cpp:15-79) ----
        action_names = ["noop"]
        self.emit(K.SEC_ACTIONS, [K.AK_NOOP, 0])
        for d in sp.move_directions:
            if d in ORIENTATIONS:  # unknown directions are skipped (actions/move.hpp:71-76)
                action_names.append(f"move_{d}")
                self.emit(K.SEC_ACTIONS, [K.AK_MOVE, ORIENTATIONS[d]])
        if sp.change_vibe_enabled:
            for i, v in enumerate(sp.vibe_names):
                action_names.append(f"change_vibe_{v}")
                self.emit(K.SEC_ACTIONS, [K.AK_VIBE, i])
        n_actions = len(action_names)
        for j in range(K.INVALID_WINDOW):
            self.agent_stats[self._invalid_pos_slot + j] = f"action.invalid_index.{n_actions + j}"

        # ---- global obs values first so that record i < NUM_OBS_VALUES is the i-th global obs value ----
        for name, v in obs.values.items():
            self.value_ref(v, feats[name])
        n_obs_values = len(obs.values)

        # ---- move handlers: custom first, then the two defaults (action_handler_factory.cpp:34-45) ----
        move_handlers = []
        for h in sp.move_handlers:
            if not isinstance(h, S.Handler):
                raise TypeError("move handlers must be plain Handler objects")
            accepts_empty = any(isinstance(f, S.TargetLocEmptyFilter) for f in h.filters)
            max_range = 1
            for f in h.filters:  # actions/move.hpp:31-40: the last MaxDistance filter sets the scan range
                if isinstance(f, S.MaxDistanceFilter):
                    max_range = f.radius if f.radius > 0 else 1
            move_handlers.append((self.handler(h), max_range, accepts_empty))
        move_handlers.append((self.handler(S.Handler([S.TargetLocEmptyFilter()], [S.Relocate()], "move")), 1, True))
        move_handlers.append((self.handler(S.Handler([S.TargetIsUsableFilter()], [S.UseTarget()], "use_target")),
                              1, False))
        for hid, rng, acc in move_handlers:
            self.emit(K.SEC_MOVE_HANDLERS, [hid, rng, 1 if acc else 0])

        # ---- agent classes (mettagrid_c_config.py:657-790) ----
        class_cells: list[str] = []
        cell_to_class: dict[str, int] = {}
        agent_rename: dict[str, list] = {}
        teams: dict[int, list] = {}
        for a in sp.agents:
            teams.setdefault(a.team_id, []).append(a)
        default_limit = sp.agents[0].inventory.default_limit
        for gid, (team_id, members) in enumerate(teams.items()):
            tags0 = set(members[0].tags)
            for m in members[1:]:
                if set(m.tags) != tags0:
                    raise ValueError(f"All agents in team {team_id} must have identical tags.")
            gname = S.TEAM_NAMES.get(team_id, f"group_{gid}")
            per_agent: list[int] = []
            for idx, a in enumerate(members):
                limit_defs, configured = [], set()
                for lim in a.inventory.limits:
                    rids = [self.res_id[n] for n in lim.resources]
                    mods = {self.res_id[n]: b for n, b in lim.modifiers.items() if n in self.res_id}
                    limit_defs.append((rids, lim.base, lim.max, mods))
                    configured.update(lim.resources)
                for rn in sp.resource_names:
                    if rn not in configured:
                        limit_defs.append(([self.res_id[rn]], default_limit, 65535, {}))
                rkey = ("rewards", repr(a.rewards))
                if rkey in self._memo:
                    rstart = self._memo[rkey]
                else:
                    rstart = self._memo[rkey] = self.counts[K.SEC_REWARDS]
                    for rw in a.rewards:
                        gstart, gcount = self.gv_emit(rw.value)
                        ts, tid = -1, -1
                        if isinstance(rw.value, S.StatValue):  # resolved (and the key created) at init: reward.hpp:45-53
                            ts = 0 if rw.value.scope == "agent" else 1
                            tid = self.stat(rw.value.scope, rw.value.name)
                        self.emit(K.SEC_REWARDS, [gstart, gcount, 1 if rw.per_tick else 0, ts, tid])
                cell = f"agent.{gname}.{idx}"
                cid = self.emit_class(
                    kind=K.KIND_AGENT, type_id=self.type_id[a.name], vibe=a.vibe, group=gid,
                    on_use=self.handler(a.on_use), on_tick=self.handler(a.on_tick),
                    on_after_use=self.handler(a.on_after_use), lim=self.limits(limit_defs),
                    init_inv=self.init_inventory(a.inventory.initial, keep_zero=True),
                    rewards=(rstart, len(a.rewards)), static=False, cell=cell, src=a,
                    tags=[self.tag_id[t] for t in list(a.tags) + [type_tag(a.name)]])
                class_cells.append(cell)
                cell_to_class[cell] = cid
                per_agent.append(cid)
            aliases = [f"agent.{gname}", f"agent.team_{gid}"]
            if team_id != gid:
                aliases.append(f"agent.team_{team_id}")
            if gid in S.TEAM_NAMES:
                aliases.append(f"agent.{S.TEAM_NAMES[gid]}")
            if team_id in S.TEAM_NAMES and team_id != gid:
                aliases.append(f"agent.{S.TEAM_NAMES[team_id]}")
            if gid == 0:
                aliases += ["agent.default", "agent.agent"]
            for al in aliases:
                if len(members) > 1:
                    agent_rename[al] = per_agent
                elif al not in cell_to_class:
                    # single-agent team: the map keeps the alias spelling, and the game stat key is
                    # "objects.<cell as spelled in the map>" (mettagrid_c.cpp:244) -> one class copy per alias.
                    woff = per_agent[0] * K.C_WORDS
                    rec = list(self.sections[K.SEC_CLASSES][woff:woff + K.C_WORDS])
                    rec[K.C_OBJECTS_STAT] = self.stat("game", f"objects.{al}")
                    cell_to_class[al] = self.emit(K.SEC_CLASSES, rec)
                    class_cells.append(al)
            # aliases of a multi-agent team share ONE counter per alias name in the reference; the reference maps
            # produced by its builders use a single alias per team, which is what class_map() implements.

        # ---- object classes (mettagrid_c_config.py:794-861) ----
        any_custom_move = bool(sp.move_handlers)
        for key, o in sp.objects.items():
            tags = [self.tag_id[t] for t in list(o.tags) + [type_tag(o.name)]]
            if o.kind == "wall":
                cid = self.emit_class(kind=K.KIND_WALL, type_id=self.type_id[o.name], vibe=o.vibe, group=0,
                                      on_use=-1, on_tick=-1, on_after_use=-1, lim=(0, 0, [-1] * K.MAX_RESOURCES, 0),
                                      init_inv=(0, 0), rewards=(0, 0), static=not any_custom_move, cell=o.cell,
                                      tags=tags, src=o)
            elif o.kind == "object":
                limit_defs, init = [], (0, 0)
                if o.inventory is not None:
                    configured = set()
                    for lim in o.inventory.limits:
                        rids = [self.res_id[n] for n in lim.resources if n in self.res_id]
                        configured.update(lim.resources)
                        if rids:
                            mods = {self.res_id[n]: b for n, b in lim.modifiers.items() if n in self.res_id}
                            limit_defs.append((rids, lim.base, lim.max, mods))
                    for rn in o.inventory.initial:
                        if rn not in configured and rn in self.res_id:
                            limit_defs.append(([self.res_id[rn]], o.inventory.default_limit, 65535, {}))
                    if o.inventory.initial:
                        init = self.init_inventory(o.inventory.initial, keep_zero=False)
                on_use = self.handler(o.on_use)
                static = (on_use < 0 and o.inventory is None and not any_custom_move)
                cid = self.emit_class(kind=K.KIND_OBJECT, type_id=self.type_id[o.name], vibe=o.vibe, group=0,
                                      on_use=on_use, on_tick=-1, on_after_use=-1, lim=self.limits(limit_defs),
                                      init_inv=init, rewards=(0, 0), static=static, cell=o.cell, tags=tags, src=o)
            else:
                raise ValueError(f"Unknown object kind: {o.kind} (key={key})")
            class_cells.append(o.cell)
            cell_to_class[o.cell] = cid

        # ---- events (sorted by name = std::map order; schedule stable-sorted by timestep: event_scheduler.cpp:8-34)
        ev_names = sorted(sp.events.keys())
        ev_recs = []
        for name in ev_names:
            ev = sp.events[name]
            qi = self.query(ev.target_query)
            fpc = self.filters_emit(ev.filters)
            ms, mc = self.mutations_emit(ev.mutations)
            mt = -1 if ev.max_targets is None else int(ev.max_targets)   # mettagrid_c_config.py:434
            fb = -1
            if ev.fallback:
                if ev.fallback not in ev_names:
                    raise ValueError(f"event '{name}': unknown fallback '{ev.fallback}'")
                fb = ev_names.index(ev.fallback)
            ev_recs.append([qi, fpc, ms, mc, mt, fb, 0, 0])
        for r in ev_recs:
            self.emit(K.SEC_EVENTS, r)
        sched = [(int(t), i) for i, name in enumerate(ev_names) for t in sp.events[name].timesteps]
        sched.sort(key=lambda x: x[0])  # Python's sort is stable, like std::stable_sort
        for t, i in sched:
            self.emit(K.SEC_SCHEDULE, [t, i])
        # ---- materialized queries (query_system.cpp:91-117) ----
        for mq in sp.materialize_queries:
            self.dynamic_tags = True
            self.indexed_tags.add(self.tag_id[mq.tag])
            self.emit(K.SEC_MATQ, [self.tag_id[mq.tag], self.query(mq.query)])
        # ---- territories (territory_tracker.cpp:72-100) ----
        def leaf_list(handlers) -> tuple[int, int]:
            ids = [self.handler(S.Handler(h.filters, h.mutations, "t"), fresh=True) for h in handlers]
            for a, b in zip(ids, ids[1:]):
                assert b == a + 1
            return (ids[0] if ids else 0), len(ids)
        for name, tc in sp.territories.items():
            self.dynamic_tags = True
            tg = self.prefix_tags(tc.tag_prefix, tc.tags)
            toff = self.emit_words(tg) if tg else 0
            es, ec = leaf_list(tc.on_enter)
            xs, xc = leaf_list(tc.on_exit)
            ps, pc = leaf_list(tc.presence)
            self.emit(K.SEC_TERRITORIES, [toff, len(tg), es, ec, xs, xc, ps, pc])
        game_on_tick = self.handler(sp.on_tick)
        if sp.events or sp.materialize_queries:
            self.dynamic_tags = self.dynamic_tags or bool(sp.materialize_queries)

        # ---- observation tables ----
        offs = observation_offsets(obs.height, obs.width)
        for dr, dc in offs:
            self.emit(K.SEC_OBS_OFFSETS, [dr, dc])
        for r in sp.resource_names:
            rec = [0] * K.IF_WORDS
            rec[0] = feats[f"inv:{r}"]
            for p in range(1, digits):
                rec[p] = feats[f"inv:{r}:p{p}"]
            self.emit(K.SEC_INV_FEATURES, rec)

        # ---- header ----
        h = self.header
        h[K.H_MAGIC], h[K.H_VERSION] = K.MAGIC, K.VERSION
        h[K.H_NUM_AGENTS] = len(sp.agents)
        h[K.H_NUM_RESOURCES] = R
        h[K.H_NUM_TAGS] = len(self.tag_names)
        h[K.H_NUM_VIBES] = len(sp.vibe_names)
        h[K.H_NUM_ACTIONS] = n_actions
        h[K.H_MAX_STEPS] = sp.max_steps
        h[K.H_EPISODE_TRUNCATES] = 1 if sp.episode_truncates else 0
        h[K.H_OBS_HEIGHT], h[K.H_OBS_WIDTH] = obs.height, obs.width
        h[K.H_NUM_TOKENS] = obs.num_tokens
        h[K.H_TOKEN_BASE] = base
        h[K.H_MAX_PRIORITY] = 1
        flags = 0
        flags |= K.G_COMPLETION if obs.episode_completion_pct else 0
        flags |= K.G_LAST_ACTION if obs.last_action else 0
        flags |= K.G_LAST_ACTION_MOVE if obs.last_action_move else 0
        flags |= K.G_LAST_REWARD if obs.last_reward else 0
        flags |= K.G_LOCAL_POSITION if obs.local_position else 0
        h[K.H_GLOBAL_FLAGS] = flags
        h[K.H_HP_RESOURCE] = self.res_id.get("hp", -1)
        h[K.H_NUM_CLASSES] = self.counts[K.SEC_CLASSES]
        h[K.H_NUM_OBS_OFFSETS] = len(offs)
        h[K.H_NUM_MOVE_HANDLERS] = len(move_handlers)
        h[K.H_NUM_OBS_VALUES] = n_obs_values
        h[K.H_NUM_EVENTS] = self.counts[K.SEC_EVENTS]
        h[K.H_NUM_SCHEDULE] = self.counts[K.SEC_SCHEDULE]
        h[K.H_NUM_MATQ] = self.counts[K.SEC_MATQ]
        h[K.H_NUM_TERRITORIES] = self.counts[K.SEC_TERRITORIES]
        h[K.H_GAME_ON_TICK] = game_on_tick
        h[K.H_DYNAMIC_TAGS] = 1 if self.dynamic_tags else 0
        h[K.H_QUERY_DEPTH] = self.query_depth
        tag_lists = [-1] * 256
        for li, t in enumerate(sorted(self.indexed_tags)):
            tag_lists[t] = li
        h[K.H_NUM_INDEXED_TAGS] = len(self.indexed_tags)
        self.sections[K.SEC_TAG_LISTS] = tag_lists
        self.counts[K.SEC_TAG_LISTS] = 256
        h[K.H_TAG_MUTATIONS] = 1 if self.tag_mutations else 0
        h[K.H_NUM_MATQ_TAGS] = len({mq.tag for mq in sp.materialize_queries})
        if self.dynamic_tags:
            cw = self.sections[K.SEC_CLASSES]
            if self.tag_mutations:   # nothing is immutable once handlers can change tags under any object
                mutable = set(range(self.counts[K.SEC_CLASSES]))
            else:
                # Only materialized queries change tags: a query result is a subset of the objects its leaf tag
                # queries list, so only classes that carry (or can be given) one of those tags can change.
                mutable = self._matq_mutable_classes()
            for c in mutable:
                cw[c * K.C_WORDS + K.C_STATIC] = 0
        fb = K.H_FEAT_BASE
        h[fb + K.F_GROUP] = feats["agent:group"]
        h[fb + K.F_COMPLETION] = feats["episode_completion_pct"]
        h[fb + K.F_LAST_ACTION] = feats["last_action"]
        h[fb + K.F_LAST_REWARD] = feats["last_reward"]
        h[fb + K.F_LAST_ACTION_MOVE] = feats.get("last_action_move", 0)
        h[fb + K.F_VIBE] = feats["vibe"]
        h[fb + K.F_TAG] = feats["tag"]
        h[fb + K.F_LP_EAST], h[fb + K.F_LP_WEST] = feats["lp:east"], feats["lp:west"]
        h[fb + K.F_LP_NORTH], h[fb + K.F_LP_SOUTH] = feats["lp:north"], feats["lp:south"]
        h[fb + K.F_AGENT_ID] = feats["agent_id"]
        h[fb + K.F_AOE_MASK] = feats.get("aoe_mask", 0)
        for key, sid in self.stat_well_known.items():
            h[K.H_STAT_BASE + key] = sid
        h[K.H_NUM_AGENT_STATS] = len(self.agent_stats)
        h[K.H_NUM_GAME_STATS] = len(self.game_stats)
        if len(self.agent_stats) > 1024 or len(self.game_stats) > 1024:
            raise RuntimeError("Exceeded maximum number of stats (MAX_STATS)")  # stats_tracker.hpp:57-67

        # ---- resolve symbolic class references (spawn mutations may name classes emitted later) ----
        for sec_words in self.sections.values():
            for i, x in enumerate(sec_words):
                if isinstance(x, tuple):
                    if x[1] not in cell_to_class:
                        raise ValueError(f"spawn references unknown object '{x[1]}'")
                    sec_words[i] = cell_to_class[x[1]]
        h[K.H_SPAWNS] = 1 if self.spawns else 0
        if self.spawns or self.removals:
            cw = self.sections[K.SEC_CLASSES]
            for c in range(self.counts[K.SEC_CLASSES]):
                cw[c * K.C_WORDS + K.C_STATIC] = 0

        # ---- resolve atom labels, lay out sections ----
        atoms = self.sections[K.SEC_ATOMS]
        for word_idx, target in self.atom_labels:
            atoms[word_idx] = target.pc if isinstance(target, _Label) else target
            assert atoms[word_idx] is not None
        off = K.H_WORDS
        for s in range(K.SEC_COUNT):
            while len(self.sections[s]) % 4:      # every section starts 16-byte aligned (vector loads of records)
                self.sections[s].append(0)
            h[K.H_SECTION_BASE + 2 * s] = off
            h[K.H_SECTION_BASE + 2 * s + 1] = self.counts[s]
            off += len(self.sections[s])
        h[K.H_TOTAL_WORDS] = off
        return h, class_cells, cell_to_class, agent_rename, action_names


def compile_spec(spec: S.GameSpec, height: int, width: int, max_objects: int | None = None) -> Program:
    """Compile ``spec`` for maps of ``height`` x ``width``.  ``max_objects`` = object slots per env (default H*W)."""
    c = _Compiler(spec, max_objects)
    h, class_cells, cell_to_class, agent_rename, action_names = c.compile()
    h[K.H_HEIGHT], h[K.H_WIDTH] = height, width
    if height > 255 or width > 255:
        raise UnsupportedFeature("maps larger than 255x255 are not supported yet")
    slots = max_objects if max_objects is not None else height * width
    if slots > 65535:
        raise UnsupportedFeature("more than 65535 object slots")
    h[K.H_MAX_OBJECTS] = slots
    words: list[int] = list(h)
    for s in range(K.SEC_COUNT):
        words.extend(c.sections[s])
    arr = np.asarray(words, dtype=np.int64)
    arr = np.where(arr >= (1 << 31), arr - (1 << 32), arr).astype(np.int32)
    assert arr.size == h[K.H_TOTAL_WORDS]
    return Program(words=arr, spec=spec, resource_names=list(spec.resource_names), vibe_names=list(spec.vibe_names),
                   tag_names=c.tag_names, type_names=c.type_names, class_cells=class_cells,
                   cell_to_class=cell_to_class, agent_rename=agent_rename, agent_stat_names=list(c.agent_stats),
                   game_stat_names=list(c.game_stats), action_names=action_names, feature_ids=dict(c.feats),
                   max_objects=slots, feature_norms=list(c.feature_norms))
