"""Game description ("spec") for the MI355X MettaGrid step engine.

This is the engine's own input language.  It carries the same information the reference converter feeds its C++
core (/root/reference/python/src/mettagrid/config/mettagrid_c_config.py:576-1007): resources, vibes, tags, per-agent
configs, map objects, handler chains (filters -> mutations), the action set, observation layout and reward
expressions.  ``mettagrid_amd.compiler.compile_spec`` lowers a ``GameSpec`` into the flat int32 program of
``include/mgx_program.h``; ``oracle/ref_driver.py`` (test-only) lowers the same spec into the reference's pybind
config objects so both engines can be driven from one description.

Names follow the reference's domain vocabulary (handler, filter, mutation, vibe, inventory limit ...).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Union

ACTOR = "actor"
TARGET = "target"


# ---------------------------------------------------------------------------------------------------------------
# Game values (reference: cpp/include/mettagrid/core/game_value_config.hpp)
# ---------------------------------------------------------------------------------------------------------------
@dataclass
class InventoryValue:
    item: str


@dataclass
class StatValue:
    name: str
    scope: str = "agent"  # "agent" | "game"
    # StatValueConfig.delta (core/game_value_config.hpp:27).  Carried for the config boundary; the engine ignores it because
    # the reference does: every consumer resolves the value afresh (handler_context.cpp:8-14, reward.hpp:49 with a const
    # read(), mettagrid_c.cpp:1219) into a ResolvedGameValue whose baseline prev_value starts at 0 and is only moved by
    # read_delta() / reset_delta() (core/resolved_game_value.hpp:28-42), which nothing calls — read() of a delta value is
    # current - 0.  tests/golden/delta_s*.npz pins that against the reference engine.
    delta: bool = False


@dataclass
class ConstValue:
    value: float


@dataclass
class SumValue:
    values: list
    weights: Optional[list] = None
    log: bool = False


@dataclass
class RatioValue:
    numerator: object
    denominator: object


@dataclass
class MaxValue:
    values: list


@dataclass
class MinValue:
    values: list


@dataclass
class QueryInventoryValue:
    item: str
    query: object


@dataclass
class QueryCountValue:
    query: object


GameValue = Union[InventoryValue, StatValue, ConstValue, SumValue, RatioValue, MaxValue, MinValue,
                  QueryInventoryValue, QueryCountValue]


# ---------------------------------------------------------------------------------------------------------------
# Queries (reference: cpp/include/mettagrid/core/query_config.hpp)
# ---------------------------------------------------------------------------------------------------------------
@dataclass
class TagQuery:
    tag: str
    filters: list = field(default_factory=list)
    max_items: object = None      # GameValue / number; None = unlimited (-1)
    order_by: str = "none"        # "none" | "random"


@dataclass
class FilteredQuery:
    source: object
    filters: list = field(default_factory=list)
    max_items: object = None
    order_by: str = "none"


@dataclass
class ClosureQuery:
    source: object
    candidates: object = None
    edge_filters: list = field(default_factory=list)
    result_filters: list = field(default_factory=list)
    max_items: object = None
    order_by: str = "none"


@dataclass
class RaycastQuery:
    source: object
    max_range: object = 2
    directions: list = field(default_factory=list)   # [(dr, dc)]; empty = 4 cardinals
    blocker: list = field(default_factory=list)      # filters, any match blocks the ray
    include_blocker: bool = True
    max_items: object = None
    order_by: str = "none"


@dataclass
class MaterializedQuery:
    tag: str
    query: object


# ---------------------------------------------------------------------------------------------------------------
# Filters (reference: cpp/include/mettagrid/core/filter_config.hpp)
# ---------------------------------------------------------------------------------------------------------------
@dataclass
class VibeFilter:
    entity: str
    vibe: str


@dataclass
class ResourceFilter:
    entity: str
    resource: str
    min_amount: int = 1


@dataclass
class SharedTagPrefixFilter:
    prefix: str
    tags: Optional[list] = None   # explicit tag names (what the reference's C++ config carries); overrides the prefix


@dataclass
class TagPrefixFilter:
    entity: str
    prefix: str
    tags: Optional[list] = None


@dataclass
class NegFilter:
    inner: list  # NOT(AND(inner))


@dataclass
class OrFilter:
    inner: list


@dataclass
class TargetLocEmptyFilter:
    pass


@dataclass
class TargetIsUsableFilter:
    pass


@dataclass
class PeriodicFilter:
    period: int
    start_on: Optional[int] = None  # reference converter default: start_on = period


@dataclass
class MaxDistanceFilter:
    entity: str
    radius: int = 0
    source: object = None   # query; None = binary form (distance to ctx.source / actor)


@dataclass
class QueryResourceFilter:
    query: object
    requirements: dict = field(default_factory=dict)   # resource -> min total


@dataclass
class GameValueFilter:
    entity: str
    value: object
    threshold: object = field(default_factory=lambda: ConstValue(0.0))


# ---------------------------------------------------------------------------------------------------------------
# Mutations (reference: cpp/include/mettagrid/core/mutation_config.hpp)
# ---------------------------------------------------------------------------------------------------------------
@dataclass
class ResourceDelta:
    entity: str
    resource: str
    delta: int


@dataclass
class ResourceTransfer:
    source: str
    destination: str
    resource: str
    amount: int = -1  # -1 = all
    remove_source_when_empty: bool = False


@dataclass
class ClearInventory:
    entity: str
    resources: list = field(default_factory=list)  # empty = all


@dataclass
class Attack:
    weapon: str
    armor: str
    health: str
    damage_multiplier_pct: int = 100


@dataclass
class SetStat:
    name: str
    value: object
    scope: str = "game"    # "game" | "agent"
    entity: str = TARGET   # which entity's stats / value context


@dataclass
class ChangeVibe:
    entity: str
    vibe: str


@dataclass
class AddTag:
    entity: str
    tag: str


@dataclass
class RemoveTag:
    entity: str
    tag: str


@dataclass
class RemoveTagsWithPrefix:
    entity: str
    prefix: str
    tags: Optional[list] = None   # explicit tag names, in removal order


@dataclass
class GameValueMutation:
    value: object            # InventoryValue or StatValue (the mutable kinds)
    source: object           # delta expression
    target: str = TARGET


@dataclass
class RecomputeMaterializedQuery:
    tag: str


@dataclass
class QueryInventoryMutation:
    query: object
    deltas: dict = field(default_factory=dict)          # resource -> delta
    source: Optional[str] = None                         # entity; set = transfer mode
    transfer_stat_names: dict = field(default_factory=dict)  # resource -> game stat name


@dataclass
class PushObject:
    pass


@dataclass
class SpawnObject:
    object_type: str    # map cell name of the class to create at ctx.target_location


@dataclass
class RaycastSpawn:
    object_type: str
    directions: list = field(default_factory=list)    # [(dr, dc)]
    max_range: object = 2
    blocker: list = field(default_factory=list)


@dataclass
class Relocate:
    pass


@dataclass
class Swap:
    pass


@dataclass
class UseTarget:
    pass


# ---------------------------------------------------------------------------------------------------------------
# Handlers
# ---------------------------------------------------------------------------------------------------------------
@dataclass
class Handler:
    filters: list = field(default_factory=list)
    mutations: list = field(default_factory=list)
    name: str = "h"


@dataclass
class FirstMatch:
    handlers: list


@dataclass
class AllOf:
    handlers: list


# ---------------------------------------------------------------------------------------------------------------
# Inventory
# ---------------------------------------------------------------------------------------------------------------
@dataclass
class Limit:
    resources: list
    base: int
    max: int = 65535
    modifiers: dict = field(default_factory=dict)  # resource name -> bonus per item held


@dataclass
class Inventory:
    initial: dict = field(default_factory=dict)    # resource name -> amount (dict order is meaningful)
    limits: list = field(default_factory=list)     # list[Limit]
    default_limit: int = 65535


# ---------------------------------------------------------------------------------------------------------------
# Objects and agents
# ---------------------------------------------------------------------------------------------------------------
@dataclass
class AOESpec:
    radius: int = 1
    is_static: bool = True
    effect_self: bool = False
    filters: list = field(default_factory=list)
    mutations: list = field(default_factory=list)
    presence_deltas: dict = field(default_factory=dict)   # resource -> delta (+ on enter, - on exit)


@dataclass
class TerritoryControl:
    territory: str
    strength: int = 1
    decay: int = 1


@dataclass
class TerritorySpec:
    tag_prefix: str
    tags: Optional[list] = None   # explicit tag names; overrides the prefix
    on_enter: list = field(default_factory=list)   # list[Handler]
    on_exit: list = field(default_factory=list)
    presence: list = field(default_factory=list)


@dataclass
class EventSpec:
    target_query: object
    timesteps: list
    filters: list = field(default_factory=list)
    mutations: list = field(default_factory=list)
    max_targets: Optional[int] = None
    fallback: Optional[str] = None


@dataclass
class ObjectSpec:
    name: str                      # type name
    map_name: Optional[str] = None  # cell name in the map (defaults to name)
    kind: str = "object"           # "object" | "wall"
    tags: list = field(default_factory=list)
    vibe: int = 0
    inventory: Optional[Inventory] = None
    on_use: object = None
    aoes: list = field(default_factory=list)                 # list[AOESpec]
    territory_controls: list = field(default_factory=list)   # list[TerritoryControl]
    on_tag_add: dict = field(default_factory=dict)           # tag prefix (or tuple of tag names) -> Handler
    on_tag_remove: dict = field(default_factory=dict)

    @property
    def cell(self) -> str:
        return self.map_name or self.name


@dataclass
class RewardSpec:
    value: object
    per_tick: bool = False


@dataclass
class AgentSpec:
    team_id: int = 0
    name: str = "agent"
    tags: list = field(default_factory=list)
    vibe: int = 0
    inventory: Inventory = field(default_factory=Inventory)
    rewards: list = field(default_factory=list)  # list[RewardSpec]
    on_use: object = None
    on_tick: object = None
    on_after_use: object = None
    aoes: list = field(default_factory=list)
    territory_controls: list = field(default_factory=list)
    on_tag_add: dict = field(default_factory=dict)
    on_tag_remove: dict = field(default_factory=dict)


TEAM_NAMES = {0: "red", 1: "blue", 2: "green", 3: "yellow", 4: "purple", 5: "orange"}


@dataclass
class ObsSpec:
    width: int = 13
    height: int = 13
    num_tokens: int = 500
    token_value_base: int = 256
    episode_completion_pct: bool = True
    last_action: bool = True
    last_action_move: bool = False
    last_reward: bool = True
    local_position: bool = False
    values: dict = field(default_factory=dict)  # feature name -> GameValue (global obs tokens)
    aoe_mask: bool = False                       # per-tile territory observability token


@dataclass
class GameSpec:
    resource_names: list
    agents: list                                   # list[AgentSpec], one per agent
    objects: dict = field(default_factory=dict)    # key -> ObjectSpec (a "wall" entry is not implicit)
    vibe_names: list = field(default_factory=lambda: ["default"])
    change_vibe_enabled: bool = True
    move_directions: list = field(default_factory=lambda: ["north", "south", "west", "east"])
    move_handlers: list = field(default_factory=list)   # custom Handler list tried before the default chain
    tags: list = field(default_factory=list)
    obs: ObsSpec = field(default_factory=ObsSpec)
    max_steps: int = 0
    episode_truncates: bool = False
    protocol_details_obs: bool = False
    events: dict = field(default_factory=dict)              # name -> EventSpec
    materialize_queries: list = field(default_factory=list) # list[MaterializedQuery]
    territories: dict = field(default_factory=dict)         # name -> TerritorySpec
    on_tick: object = None                                  # game-level handler

    @property
    def num_agents(self) -> int:
        return len(self.agents)
