"""Drop-in for the reference's native module ``mettagrid.mettagrid_c`` on an MI355X.

The reference's Python layer talks to its C++ core through one pybind11 module
(/root/reference/cpp/bindings/mettagrid_py.cpp:242-397 and the ``bind_*`` functions it calls): ~70 config classes that
``mettagrid/config/mettagrid_c_config.py:576-1007`` fills in, the ``MettaGrid`` class, ``PackedCoordinate`` and the
buffer dtypes.  This module exports the same names with the same constructor signatures, attributes and methods, so
that ``sys.modules["mettagrid.mettagrid_c"] = mettagrid_amd.mettagrid_c`` makes the unchanged
reference converter build a config tree that ``MettaGrid(cfg, map, seed)`` here lowers (``from_reference.py``) into
the engine's program and runs on the GPU through libmgx.

The config classes are plain records: they store what the converter passes.  Nothing is computed in them.
"""
from __future__ import annotations

import enum

import numpy as np

# ---- buffer dtypes (cpp/include/mettagrid/core/types.hpp:11-44) ---------------------------------------------------
dtype_observations = np.dtype(np.uint8)
dtype_terminals = np.dtype(np.bool_)
dtype_truncations = np.dtype(np.bool_)
dtype_rewards = np.dtype(np.float32)
dtype_actions = np.dtype(np.int32)
dtype_masks = np.dtype(np.bool_)
dtype_success = np.dtype(np.bool_)
EpisodeStats = dict


class PackedCoordinate:
    """cpp/include/mettagrid/systems/packed_coordinate.hpp:28-84,160-184: (row, col) in one byte, 0xFF = empty."""
    MAX_PACKABLE_COORD = 14
    EMPTY = 0xFF
    GLOBAL_LOCATION = 0xFE

    @staticmethod
    def pack(row: int, col: int) -> int:
        if row > PackedCoordinate.MAX_PACKABLE_COORD or col > PackedCoordinate.MAX_PACKABLE_COORD or row < 0 or col < 0:
            raise ValueError(f"Coordinates ({row}, {col}) exceed maximum packable coordinate "
                             f"{PackedCoordinate.MAX_PACKABLE_COORD}")
        return (row << 4) | col

    @staticmethod
    def unpack(packed: int):
        if packed == PackedCoordinate.EMPTY:
            return None
        return (packed >> 4) & 0xF, packed & 0xF

    @staticmethod
    def is_empty(packed: int) -> bool:
        return packed == PackedCoordinate.EMPTY


# ---- enums ----------------------------------------------------------------------------------------------------------
class GameValueScope(enum.Enum):
    AGENT = 0
    GAME = 1


class EntityRef(enum.Enum):
    actor = 0
    target = 1


class HandlerMode(enum.Enum):
    FirstMatch = 0
    All = 1


class StatsTarget(enum.Enum):
    game = 0
    agent = 1


class StatsEntity(enum.Enum):
    target = 0
    actor = 1


class QueryOrderBy(enum.Enum):
    none = 0
    random = 1


# ---- config records -------------------------------------------------------------------------------------------------
_FILTER_LISTS = {"": "filters", "edge_": "edge_filters", "result_": "result_filters", "blocker_": "blocker"}


class _Record:
    """Base of every config class: constructor arguments by position (``_fields``) or keyword, defaults from
    ``_defaults``, free attribute assignment, and the uniform ``add_<kind>_filter`` / ``add_<kind>_mutation`` /
    ``set_<what>`` methods of the pybind classes (each pushes to / sets one ordered member, like the C++ side does)."""
    _fields: tuple = ()
    _defaults: dict = {}
    _filter_member = "filters"

    def __init__(self, *args, **kwargs) -> None:
        for k, v in self._defaults.items():
            setattr(self, k, v() if callable(v) else v)
        if len(args) > len(self._fields):
            raise TypeError(f"{type(self).__name__}(): takes at most {len(self._fields)} positional arguments")
        for name, v in zip(self._fields, args):
            setattr(self, name, v)
        for k, v in kwargs.items():
            if k not in self._fields and k not in self._defaults:
                raise TypeError(f"{type(self).__name__}(): unexpected keyword argument '{k}'")
            setattr(self, k, v)

    def __getattr__(self, name: str):
        if name.startswith("add_") and name.endswith("_filter"):
            body = name[4:-7]
            prefix = next((p for p in ("edge_", "result_", "blocker_") if body.startswith(p)), "")
            member = self._filter_member if not prefix else _FILTER_LISTS[prefix]
            return lambda cfg: self.__dict__.setdefault(member, []).append(cfg)
        if name.startswith("add_") and name.endswith("_mutation"):
            return lambda cfg: self.__dict__.setdefault("mutations", []).append(cfg)
        if name.startswith("set_"):
            member = name[4:]
            return lambda value: setattr(self, member, value)
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    def __repr__(self) -> str:
        return f"{type(self).__name__}({', '.join(f'{k}={v!r}' for k, v in self.__dict__.items())})"


def _record(name: str, fields=(), defaults=None, filter_member: str = "filters", base=_Record):
    return type(name, (base,), {"_fields": tuple(fields), "_defaults": dict(defaults or {}),
                                "_filter_member": filter_member, "__module__": __name__})


# game values (handler/handler_bindings.hpp:24-86)
InventoryValueConfig = _record("InventoryValueConfig", defaults={"scope": GameValueScope.AGENT, "id": 0})
StatValueConfig = _record("StatValueConfig", defaults={"scope": GameValueScope.AGENT, "id": 0, "delta": False, "stat_name": ""})
ConstValueConfig = _record("ConstValueConfig", defaults={"value": 0.0})
QueryInventoryValueConfig = _record("QueryInventoryValueConfig", defaults={"id": 0, "query": None})
QueryCountValueConfig = _record("QueryCountValueConfig", defaults={"query": None})


class _ValueList(_Record):
    def add_value(self, value) -> None:
        self.values.append(value)


SumValueConfig = _record("SumValueConfig", defaults={"values": list, "weights": list, "log": False}, base=_ValueList)
MaxValueConfig = _record("MaxValueConfig", defaults={"values": list}, base=_ValueList)
MinValueConfig = _record("MinValueConfig", defaults={"values": list}, base=_ValueList)
RatioValueConfig = _record("RatioValueConfig", defaults={"numerator": None, "denominator": None})

# filters (handler/handler_bindings.hpp:101-280, config/mettagrid_config.hpp:284-292)
VibeFilterConfig = _record("VibeFilterConfig", ("entity", "vibe_id"), {"entity": EntityRef.target, "vibe_id": 0})
ResourceFilterConfig = _record("ResourceFilterConfig", ("entity", "resource_id", "min_amount"),
                               {"entity": EntityRef.target, "resource_id": 0, "min_amount": 1})
SharedTagPrefixFilterConfig = _record("SharedTagPrefixFilterConfig", ("tag_ids",), {"tag_ids": list})
TagPrefixFilterConfig = _record("TagPrefixFilterConfig", ("entity", "tag_ids"), {"entity": EntityRef.target, "tag_ids": list})
QueryResourceFilterConfig = _record("QueryResourceFilterConfig", defaults={"requirements": list, "query": None})
GameValueFilterConfig = _record("GameValueFilterConfig", ("value", "threshold", "entity"),
                                {"value": InventoryValueConfig, "threshold": ConstValueConfig, "entity": EntityRef.target})
PeriodicFilterConfig = _record("PeriodicFilterConfig", ("period", "start_on"), {"period": 1, "start_on": 0})
NegFilterConfig = _record("NegFilterConfig", defaults={"inner": list}, filter_member="inner")
OrFilterConfig = _record("OrFilterConfig", defaults={"inner": list}, filter_member="inner")
MaxDistanceFilterConfig = _record("MaxDistanceFilterConfig", defaults={"entity": EntityRef.target, "radius": 0, "source": None})
TargetLocEmptyFilterConfig = _record("TargetLocEmptyFilterConfig")
TargetIsUsableFilterConfig = _record("TargetIsUsableFilterConfig")

# mutations (handler/handler_bindings.hpp:282-470)
ResourceDeltaMutationConfig = _record("ResourceDeltaMutationConfig", ("entity", "resource_id", "delta"),
                                      {"entity": EntityRef.target, "resource_id": 0, "delta": 0})
ResourceTransferMutationConfig = _record(
    "ResourceTransferMutationConfig", ("source", "destination", "resource_id", "amount", "remove_source_when_empty"),
    {"source": EntityRef.actor, "destination": EntityRef.target, "resource_id": 0, "amount": -1,
     "remove_source_when_empty": False})
ClearInventoryMutationConfig = _record("ClearInventoryMutationConfig", ("entity", "resource_ids"),
                                       {"entity": EntityRef.target, "resource_ids": list})
AttackMutationConfig = _record("AttackMutationConfig", ("weapon_resource", "armor_resource", "health_resource",
                                                       "damage_multiplier_pct"),
                               {"weapon_resource": 0, "armor_resource": 0, "health_resource": 0, "damage_multiplier_pct": 100})
StatsMutationConfig = _record("StatsMutationConfig", ("stat_name", "target", "entity"),
                              {"stat_name": "", "target": StatsTarget.game, "entity": StatsEntity.target, "source": None})
AddTagMutationConfig = _record("AddTagMutationConfig", ("entity", "tag_id"), {"entity": EntityRef.target, "tag_id": -1})
RemoveTagMutationConfig = _record("RemoveTagMutationConfig", ("entity", "tag_id"), {"entity": EntityRef.target, "tag_id": -1})
ChangeVibeMutationConfig = _record("ChangeVibeMutationConfig", ("entity", "vibe_id"), {"entity": EntityRef.target, "vibe_id": 0})
RemoveTagsWithPrefixMutationConfig = _record("RemoveTagsWithPrefixMutationConfig", ("entity", "tag_ids"),
                                             {"entity": EntityRef.target, "tag_ids": list})
GameValueMutationConfig = _record("GameValueMutationConfig", ("value", "target", "source"),
                                  {"value": InventoryValueConfig, "target": EntityRef.target, "source": InventoryValueConfig})
QueryInventoryMutationConfig = _record("QueryInventoryMutationConfig",
                                       defaults={"deltas": list, "source": EntityRef.target, "has_source": False,
                                                 "transfer_stat_names": list, "query": None})
RecomputeMaterializedQueryMutationConfig = _record("RecomputeMaterializedQueryMutationConfig", defaults={"tag_id": -1})
RelocateMutationConfig = _record("RelocateMutationConfig")
SwapMutationConfig = _record("SwapMutationConfig")
UseTargetMutationConfig = _record("UseTargetMutationConfig")
PushObjectMutationConfig = _record("PushObjectMutationConfig")
SpawnObjectMutationConfig = _record("SpawnObjectMutationConfig", defaults={"object_type": ""})
RaycastSpawnMutationConfig = _record("RaycastSpawnMutationConfig",
                                     defaults={"object_type": "", "max_range": None, "directions": list, "blocker": list})

# handlers (handler/handler_bindings.hpp:472-646)
HandlerConfig = _record("HandlerConfig", ("name",), {"name": "", "filters": list, "mutations": list})
ResourceDelta = _record("ResourceDelta", ("resource_id", "delta"), {"resource_id": 0, "delta": 0})
AOEConfig = _record("AOEConfig", ("name",), {"name": "", "filters": list, "mutations": list, "radius": 1, "is_static": True,
                                             "effect_self": False, "presence_deltas": list})
TerritoryConfig = _record("TerritoryConfig", defaults={"tag_prefix_ids": list, "on_enter": list, "on_exit": list, "presence": list})
TerritoryControlConfig = _record("TerritoryControlConfig", defaults={"strength": 1, "decay": 1, "territory_index": 0})
Handler = _record("Handler", ("config",), {"config": None})


class MultiHandler(_Record):
    _fields = ("handlers", "mode")
    _defaults = {"handlers": list, "mode": HandlerMode.FirstMatch}

    def __len__(self) -> int:
        return len(self.handlers)

    def __bool__(self) -> bool:
        return len(self.handlers) > 0


# queries (config/mettagrid_config.hpp:107-345)
TagQueryConfig = _record("TagQueryConfig", defaults={"tag_id": -1, "max_items": None, "order_by": QueryOrderBy.none, "filters": list})
FilteredQueryConfig = _record("FilteredQueryConfig", defaults={"source": None, "max_items": None, "order_by": QueryOrderBy.none,
                                                              "filters": list})
ClosureQueryConfig = _record("ClosureQueryConfig", defaults={"source": None, "candidates": None, "max_items": None,
                                                            "order_by": QueryOrderBy.none, "edge_filters": list,
                                                            "result_filters": list})
RaycastQueryConfig = _record("RaycastQueryConfig", defaults={"source": None, "max_range": None, "max_items": None,
                                                            "directions": list, "include_blocker": True, "blocker": list})
QueryConfigHolder = _record("QueryConfigHolder", ("config",), {"config": None})
MaterializedQueryTag = _record("MaterializedQueryTag", defaults={"tag_id": -1, "query": None})


def make_query_config(query) -> "QueryConfigHolder":
    return QueryConfigHolder(query)


# events (handler/event_bindings.hpp:16-122)
EventConfig = _record("EventConfig", ("name",), {"name": "", "timesteps": list, "max_targets": -1, "fallback": "",
                                                 "target_query": None, "filters": list, "mutations": list})

# objects, agents, inventory, rewards
LimitDef = _record("LimitDef", ("resources", "min_limit", "max_limit", "modifiers"),
                   {"resources": list, "min_limit": 0, "max_limit": 65535, "modifiers": dict})
InventoryConfig = _record("InventoryConfig", defaults={"limit_defs": list})
RewardEntry = _record("RewardEntry", defaults={"reward": None, "accumulate": False})
RewardConfig = _record("RewardConfig", defaults={"entries": list})


class GridObjectConfig(_Record):
    """cpp/bindings/mettagrid_py.cpp:318-346."""
    _fields = ("type_id", "type_name", "initial_vibe")
    _defaults = {"type_id": 0, "type_name": "", "initial_vibe": 0, "name": "", "tag_ids": list, "on_use_handler": None,
                 "aoe_configs": list, "territory_controls": list, "initial_inventory": dict, "inventory_config": InventoryConfig,
                 "on_tag_add": list, "on_tag_remove": list}

    def add_on_tag_add_handler(self, tag_id: int, handler) -> None:
        self.on_tag_add.append((tag_id, handler))

    def add_on_tag_remove_handler(self, tag_id: int, handler) -> None:
        self.on_tag_remove.append((tag_id, handler))


class WallConfig(GridObjectConfig):
    """cpp/include/mettagrid/objects/wall.hpp:29-37."""


class AgentConfig(GridObjectConfig):
    """cpp/include/mettagrid/objects/agent_config.hpp:49-78."""
    _fields = ("type_id", "type_name", "group_id", "group_name", "initial_vibe", "inventory_config", "reward_config",
               "initial_inventory", "on_tick")
    _defaults = dict(GridObjectConfig._defaults, type_name="agent", group_id=0, group_name="", reward_config=RewardConfig,
                     on_tick=None, on_after_use_handler=None)


ObsValueConfig = _record("ObsValueConfig", defaults={"value": None, "feature_id": 0})
GlobalObsConfig = _record("GlobalObsConfig", ("episode_completion_pct", "last_action", "last_action_move", "last_reward",
                                              "goal_obs", "local_position", "obs"),
                          {"episode_completion_pct": True, "last_action": True, "last_action_move": False, "last_reward": True,
                           "goal_obs": False, "local_position": False, "obs": list})

# actions
ActionConfig = _record("ActionConfig", ("required_resources", "consumed_resources"),
                       {"required_resources": dict, "consumed_resources": dict})
MoveActionConfig = _record("MoveActionConfig", ("allowed_directions", "required_resources", "consumed_resources", "handlers"),
                           {"allowed_directions": lambda: ["north", "south", "west", "east"], "required_resources": dict,
                            "consumed_resources": dict, "handlers": list})
ChangeVibeActionConfig = _record("ChangeVibeActionConfig", ("required_resources", "consumed_resources", "number_of_vibes"),
                                 {"required_resources": dict, "consumed_resources": dict, "number_of_vibes": 0})
AttackOutcome = _record("AttackOutcome", ("actor", "target", "loot"), {"actor": dict, "target": dict, "loot": list})
AttackActionConfig = _record("AttackActionConfig",
                             ("required_resources", "consumed_resources", "defense_resources", "armor_resources",
                              "weapon_resources", "success", "enabled", "vibes", "vibe_bonus"),
                             {"required_resources": dict, "consumed_resources": dict, "defense_resources": dict,
                              "armor_resources": dict, "weapon_resources": dict, "success": AttackOutcome, "enabled": True,
                              "vibes": list, "vibe_bonus": dict})

GameConfig = _record(
    "GameConfig",
    ("num_agents", "max_steps", "episode_truncates", "obs_width", "obs_height", "resource_names", "vibe_names",
     "num_observation_tokens", "global_obs", "feature_ids", "actions", "objects", "tag_id_map", "territories",
     "protocol_details_obs", "reward_estimates", "token_value_base", "events", "materialized_queries", "on_tick"),
    {"tag_id_map": dict, "territories": list, "protocol_details_obs": True, "reward_estimates": dict, "token_value_base": 256,
     "events": dict, "materialized_queries": list, "on_tick": None})


# ---- the engine -----------------------------------------------------------------------------------------------------
class TagIndex:
    """mettagrid_py.cpp:377-380: what ``MettaGrid.tag_index()`` returns."""

    def __init__(self, engine) -> None:
        self._engine = engine

    def count_objects_with_tag(self, tag_id: int) -> int:
        return self._engine._b.count_objects_with_tag(0, int(tag_id))


def __getattr__(name: str):
    if name == "MettaGrid":  # imported lazily: the engine needs libmgx, the config records do not
        from .from_reference import ReferenceMettaGrid
        return ReferenceMettaGrid
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
