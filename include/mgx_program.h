/*
 * mgx_program.h — flat "game program" consumed by the MI355X step engine (libmgx) and by the CPU oracle.
 *
 * The reference hands its engine a tree of ~80 pybind config classes
 * (/root/reference/cpp/include/mettagrid/config/mettagrid_config.hpp:45-77, built by
 * /root/reference/python/src/mettagrid/config/mettagrid_c_config.py:576-1007).  A GPU cannot chase that tree, so
 * the host compiles the same information ONCE into a single int32 blob: a fixed header, a section table and flat
 * record arrays, every section starting on a 16-byte boundary (classes, limit groups, handler tree, short-circuit filter code, mutation lists, game-value
 * postfix code, observation tables, stat ids).  Floats are stored bit-cast into int32 words.  Strings never reach
 * the device; the host keeps the name<->id tables (mettagrid_amd/compiler.py).
 *
 * This header is a FORMAT description (no algorithm) and is shared by the product (mettagrid_amd/csrc) and by the
 * test-only CPU restatement (oracle/mgx_oracle.cpp).
 */
#ifndef MGX_PROGRAM_H_
#define MGX_PROGRAM_H_

#include <stdint.h>

#define MGX_MAGIC 0x3158474d /* "MGX1" little-endian */
#define MGX_VERSION 8

#define MGX_MAX_RESOURCES 13 /* inventory order list = 4-bit ids in one u64, 0xF terminator; see DESIGN.md */
#define MGX_TAG_WORDS 8      /* 256 tags = 8 x u32 (reference kMaxTags, core/types.hpp:62) */
#define MGX_MAX_AGENTS 256
#define MGX_INVALID_WINDOW 16 /* action.invalid_index.<k> is a fixed stat column for k in [-16,-1] and [n_actions, n_actions+15] */
#define MGX_INVALID_EXTRA 4   /* ... and one of this many (k, count) pairs per agent and episode for any other k */

/* ---- header word indices ------------------------------------------------------------------------------------- */
enum {
  MGX_H_MAGIC = 0,
  MGX_H_VERSION,
  MGX_H_TOTAL_WORDS,
  MGX_H_HEIGHT,
  MGX_H_WIDTH,
  MGX_H_NUM_AGENTS,
  MGX_H_MAX_OBJECTS,   /* object slots per env (>= objects on any map; slot = reference object id - 1) */
  MGX_H_NUM_RESOURCES,
  MGX_H_NUM_TAGS,
  MGX_H_NUM_VIBES,
  MGX_H_NUM_ACTIONS,
  MGX_H_MAX_STEPS,
  MGX_H_EPISODE_TRUNCATES,
  MGX_H_OBS_HEIGHT,
  MGX_H_OBS_WIDTH,
  MGX_H_NUM_TOKENS,
  MGX_H_TOKEN_BASE,
  MGX_H_MAX_PRIORITY,   /* reference: always 1 (inert Attack handler, actions/attack.hpp:77) */
  MGX_H_GLOBAL_FLAGS,   /* MGX_G_* bits */
  MGX_H_HP_RESOURCE,    /* id of the resource literally named "hp" (objects/agent.cpp:117), or -1 */
  MGX_H_NUM_AGENT_STATS,
  MGX_H_NUM_GAME_STATS,
  MGX_H_NUM_CLASSES,
  MGX_H_NUM_OBS_OFFSETS,
  MGX_H_NUM_MOVE_HANDLERS,
  MGX_H_NUM_OBS_VALUES,
  MGX_H_NUM_EVENTS,
  MGX_H_NUM_SCHEDULE,
  MGX_H_NUM_MATQ,
  MGX_H_NUM_TERRITORIES,
  MGX_H_GAME_ON_TICK,    /* handler index or -1 (GameConfig.on_tick, mettagrid_c.cpp:1050-1052) */
  MGX_H_DYNAMIC_TAGS,    /* 1: objects carry their own tag bitset (tag mutations / materialized queries / territories) */
  MGX_H_NUM_INDEXED_TAGS,/* tags that own a TagIndex list (referenced by a TagQuery) */
  MGX_H_QUERY_DEPTH,     /* max nesting of queries */
  MGX_H_SPAWNS,          /* 1: the program can create objects at run time (Spawn / RaycastSpawn) */
  MGX_H_TAG_MUTATIONS,   /* 1: some handler adds / removes tags (any object's tag set may change); 0 with DYNAMIC_TAGS set:
                            only materialized queries change tags, and only on classes whose MGX_C_STATIC is 0 */
  MGX_H_NUM_MATQ_TAGS,   /* distinct tags materialized queries can add to an object (bound of its extra tag tokens) */
  MGX_H_FEAT_BASE = 40, /* MGX_F_* feature ids follow */
  MGX_H_STAT_BASE = 56, /* MGX_S_* well-known stat ids follow */
  MGX_H_SECTION_BASE = 104, /* section s: offset at BASE+2s, record count at BASE+2s+1 */
  MGX_H_WORDS = 176
};

enum { /* MGX_H_GLOBAL_FLAGS bits (config/mettagrid_config.hpp:37-45 GlobalObsConfig) */
  MGX_G_COMPLETION = 1, MGX_G_LAST_ACTION = 2, MGX_G_LAST_ACTION_MOVE = 4, MGX_G_LAST_REWARD = 8,
  MGX_G_LOCAL_POSITION = 16
};

enum { /* feature ids, offsets from MGX_H_FEAT_BASE (python/src/mettagrid/config/id_map.py:161-235) */
  MGX_F_GROUP = 0, MGX_F_COMPLETION, MGX_F_LAST_ACTION, MGX_F_LAST_REWARD, MGX_F_LAST_ACTION_MOVE, MGX_F_VIBE,
  MGX_F_TAG, MGX_F_LP_EAST, MGX_F_LP_WEST, MGX_F_LP_NORTH, MGX_F_LP_SOUTH, MGX_F_AGENT_ID, MGX_F_AOE_MASK
};

enum { /* well-known stat ids, offsets from MGX_H_STAT_BASE; agent scope unless noted; -1 = absent */
  MGX_S_ACTION_FAILED = 0,      /* "action.failed" */
  MGX_S_NOOP_SUCCESS, MGX_S_NOOP_FAILED, MGX_S_MOVE_SUCCESS, MGX_S_MOVE_FAILED, MGX_S_VIBE_SUCCESS,
  MGX_S_VIBE_FAILED,            /* "action.<noop|move|change_vibe>.<success|failed>" */
  MGX_S_INVALID_INDEX,          /* "action.invalid_index" */
  MGX_S_INVALID_NEG_BASE,       /* "action.invalid_index.<k>" k=-16..-1  -> id = base + (k+16) */
  MGX_S_INVALID_POS_BASE,       /* k = n_actions..n_actions+15           -> id = base + (k-n_actions) */
  MGX_S_MAX_STEPS_WITHOUT_MOTION,
  MGX_S_SWAP,                   /* "actions.swap" */
  MGX_S_DEATH,
  MGX_S_CELL_VISITED, MGX_S_CELL_UNIQUE, MGX_S_CELL_MAXDIST,
  MGX_S_RES_AMOUNT_BASE,        /* "<res>.amount"   id = base + res */
  MGX_S_RES_GAINED_BASE, MGX_S_RES_LOST_BASE, MGX_S_RES_DEPOSITED_BASE,
  MGX_S_GAME_TOKENS_WRITTEN,    /* game scope */
  MGX_S_GAME_TOKENS_DROPPED, MGX_S_GAME_TOKENS_FREE
};

/* ---- sections ------------------------------------------------------------------------------------------------- */
enum {
  MGX_SEC_CLASSES = 0, MGX_SEC_LIMITS, MGX_SEC_MODS, MGX_SEC_DROP_ORDER, MGX_SEC_INIT_INV, MGX_SEC_HANDLERS,
  MGX_SEC_CHILDREN, MGX_SEC_ATOMS, MGX_SEC_MUTS, MGX_SEC_ACTIONS, MGX_SEC_MOVE_HANDLERS, MGX_SEC_OBS_OFFSETS,
  MGX_SEC_INV_FEATURES, MGX_SEC_GV_CODE, MGX_SEC_REWARDS, MGX_SEC_OBS_VALUES, MGX_SEC_WORDLIST,
  MGX_SEC_QUERIES, MGX_SEC_EVENTS, MGX_SEC_MATQ, MGX_SEC_TAG_HANDLERS, MGX_SEC_AOES,
  MGX_SEC_PRESENCE, MGX_SEC_TERRITORIES, MGX_SEC_TERR_CONTROLS, MGX_SEC_TAG_LISTS /* 256 words: tag -> list idx or -1 */,
  MGX_SEC_SCHEDULE /* last: the only section that grows with the episode length (periodic events); the world kernels
                      keep everything in front of it in LDS and read the schedule, one entry per firing, from HBM */,
  MGX_SEC_COUNT
};

/* Object class = one entry of GameConfig.objects (keyed by map cell name; per-agent cells "agent.red.3" are
 * separate classes that share a type_id; mettagrid_c_config.py:757-776). */
enum {
  MGX_C_KIND = 0,       /* MGX_KIND_* */
  MGX_C_TYPE_ID,
  MGX_C_INITIAL_VIBE,
  MGX_C_GROUP,          /* agents */
  MGX_C_ON_USE,         /* handler index or -1 */
  MGX_C_ON_TICK,
  MGX_C_ON_AFTER_USE,
  MGX_C_LIMIT_START, MGX_C_LIMIT_COUNT,
  MGX_C_INIT_INV_START, MGX_C_INIT_INV_COUNT, /* (item, amount) pairs, in the order the reference inserts them */
  MGX_C_REWARD_START, MGX_C_REWARD_COUNT,
  MGX_C_STATIC,         /* 1: no mutable state (wall-like): tokens come from the class tag mask only */
  MGX_C_OBJECTS_STAT,   /* game stat id of "objects.<cell name>" */
  MGX_C_MODIFIER_MASK,  /* bit r set: resource r is a limit modifier for this class (Inventory::is_modifier) */
  MGX_C_TAGS,           /* MGX_TAG_WORDS words */
  MGX_C_RES_LIMIT = MGX_C_TAGS + MGX_TAG_WORDS, /* MGX_MAX_RESOURCES words: limit index (global) or -1 */
  MGX_C_TAG_ADD_START = MGX_C_RES_LIMIT + MGX_MAX_RESOURCES, MGX_C_TAG_ADD_COUNT, /* MGX_SEC_TAG_HANDLERS records */
  MGX_C_TAG_REMOVE_START, MGX_C_TAG_REMOVE_COUNT,
  MGX_C_AOE_START, MGX_C_AOE_COUNT,        /* MGX_SEC_AOES records */
  MGX_C_TERR_START, MGX_C_TERR_COUNT,      /* MGX_SEC_TERR_CONTROLS records */
  MGX_C_WORDS = ((MGX_C_TERR_COUNT + 1 + 3) / 4) * 4
};
enum { MGX_KIND_WALL = 0, MGX_KIND_OBJECT = 1, MGX_KIND_AGENT = 2 };

/* Shared inventory limit (objects/inventory.hpp:17-40). */
enum { MGX_L_RES_MASK = 0, MGX_L_MIN, MGX_L_MAX, MGX_L_MOD_START, MGX_L_MOD_COUNT, MGX_L_DROP_START,
       MGX_L_DROP_COUNT, MGX_L_WORDS };
enum { MGX_MOD_ITEM = 0, MGX_MOD_BONUS, MGX_MOD_WORDS };
enum { MGX_II_ITEM = 0, MGX_II_AMOUNT, MGX_II_WORDS };

/* Handler tree (handler/handler.cpp:76-93, multi_handler.cpp:8-21). */
enum { MGX_HD_KIND = 0, MGX_HD_FILTER_PC, /* first atom or MGX_PC_PASS */ MGX_HD_MUT_START, MGX_HD_MUT_COUNT,
       MGX_HD_CHILD_START, MGX_HD_CHILD_COUNT, MGX_HD_PAD0, MGX_HD_PAD1, MGX_HD_WORDS /* 8: two 16-byte loads */ };
enum { MGX_HK_LEAF = 0, MGX_HK_FIRST_MATCH = 1, MGX_HK_ALL = 2 };

/* Filter atoms: linear short-circuit code; every atom carries the pc to continue at when it is true / false. */
enum { MGX_AT_OP = 0, MGX_AT_A0, MGX_AT_A1, MGX_AT_A2, MGX_AT_ON_TRUE, MGX_AT_ON_FALSE, MGX_AT_PAD0, MGX_AT_PAD1,
       MGX_AT_WORDS /* 8 */ };
#define MGX_PC_PASS (-1)
#define MGX_PC_FAIL (-2)
enum {
  MGX_FOP_VIBE = 0,          /* a0 entity, a1 vibe                      filters/vibe_filter.hpp:13-29 */
  MGX_FOP_RESOURCE,          /* a0 entity, a1 resource, a2 min          filters/resource_filter.hpp:12-27 */
  MGX_FOP_SHARED_TAG,        /* a0 WORDLIST offset of 8-word mask         filters/shared_tag_filter.hpp:16-36 */
  MGX_FOP_TAG,               /* a0 entity, a1 mask offset               filters/shared_tag_filter.hpp:38-61 */
  MGX_FOP_TARGET_LOC_EMPTY,  /*                                         filters/target_loc_empty_filter.hpp */
  MGX_FOP_TARGET_IS_USABLE,  /*                                         filters/target_is_usable_filter.hpp */
  MGX_FOP_PERIODIC,          /* a0 period, a1 start_on                  filters/periodic_filter.hpp:15-29 */
  MGX_FOP_GAME_VALUE,        /* a0 entity, a1 value code, a2 threshold  filters/game_value_filter.hpp:17-29 */
  MGX_FOP_MAX_DISTANCE,      /* a0 entity, a1 radius, a2 source query or -1 (binary form)  filters/max_distance_filter.hpp:27-67 */
  MGX_FOP_QUERY_RESOURCE,    /* a0 query, a1 WORDLIST start of (resource, min) pairs, a2 pair count  query_resource_filter.hpp:20-47 */
  MGX_FOP_TRUE, MGX_FOP_FALSE
};
enum { MGX_ENT_ACTOR = 0, MGX_ENT_TARGET = 1 };
#define MGX_SLOT_NONE (-1)
#define MGX_SLOT_PROXY (-2) /* territory proxy cell: carries only the winning tag (territory_tracker.cpp:289-342) */

/* Mutations (handler/mutations/). */
enum { MGX_MU_OP = 0, MGX_MU_A0, MGX_MU_A1, MGX_MU_A2, MGX_MU_A3, MGX_MU_A4, MGX_MU_PAD0, MGX_MU_PAD1, MGX_MU_WORDS /* 8 */ };
enum {
  MGX_MOP_RESOURCE_DELTA = 0, /* a0 entity, a1 resource, a2 delta              resource_mutation.hpp:21-50 */
  MGX_MOP_RESOURCE_TRANSFER,  /* a0 src, a1 dst, a2 resource, a3 amount(-1 all), a4 remove_when_empty  :52-103 */
  MGX_MOP_CLEAR_INVENTORY,    /* a0 entity, a1 WORDLIST start, a2 count (0 = all)  :105-132 */
  MGX_MOP_ATTACK,             /* a0 weapon, a1 armor, a2 health, a3 pct        attack_mutation.hpp:16-42 */
  MGX_MOP_STATS,              /* a0 scope(0 game,1 agent), a1 entity, a2 stat id, a3 value code  stats_mutation.hpp */
  MGX_MOP_CHANGE_VIBE,        /* a0 entity, a1 vibe                            change_vibe_mutation.hpp */
  MGX_MOP_RELOCATE,           /*                                               relocate_mutation.hpp */
  MGX_MOP_SWAP,               /*                                               swap_mutation.hpp */
  MGX_MOP_USE_TARGET,         /*                                               use_target_mutation.hpp */
  MGX_MOP_GAME_VALUE,         /* a0 target entity, a1 value code, a2 source code  game_value_mutation.hpp */
  MGX_MOP_ADD_TAG, MGX_MOP_REMOVE_TAG, /* a0 entity, a1 tag                    tag_mutation.hpp:16-45 */
  MGX_MOP_REMOVE_TAGS_PREFIX, /* a0 entity, a1 WORDLIST start, a2 count         tag_mutation.hpp:47-67 */
  MGX_MOP_RECOMPUTE_QUERY,    /* a0 tag                                         recompute_materialized_query_mutation.hpp */
  MGX_MOP_PUSH_OBJECT,        /*                                                push_object_mutation.hpp:29-71 */
  MGX_MOP_SPAWN_OBJECT,       /* a0 class                                       spawn_object_mutation.cpp:10-64 */
  MGX_MOP_RAYCAST_SPAWN,      /* a0 class, a1 WORDLIST (dr, dc) pairs, a2 count, a3 max_range value, a4 blocker pc  raycast_spawn_mutation.cpp:15-92 */
  MGX_MOP_QUERY_INVENTORY     /* a0 query, a1 WORDLIST (res, delta) pairs, a2 count, a3 source entity or -1,
                                 a4 WORDLIST (res, game stat id) pairs, PAD0 = their count   query_inventory_mutation.hpp:22-75 */
};

/* Actions (actions/action_handler_factory.cpp:15-79): index space [noop, move_<dir>..., change_vibe_<v>...]. */
enum { MGX_AC_KIND = 0, MGX_AC_ARG, MGX_AC_WORDS };
enum { MGX_AK_NOOP = 0, MGX_AK_MOVE = 1, MGX_AK_VIBE = 2 };
enum { MGX_MH_HANDLER = 0, MGX_MH_MAX_RANGE, MGX_MH_ACCEPTS_EMPTY, MGX_MH_WORDS }; /* actions/move.hpp:26-46 */

enum { MGX_OO_DR = 0, MGX_OO_DC, MGX_OO_WORDS }; /* core/observation_shape.cpp:54-66 */
/* MGX_SEC_INV_FEATURES: record r = [digits] feature ids for resource r (observation_encoder.hpp:60-87);
 * record width = MGX_IF_WORDS. */
enum { MGX_IF_WORDS = 16 };

/* Game-value postfix code (core/game_value.cpp:14-148). A value is (start, count) into MGX_SEC_GV_CODE; each
 * instruction is MGX_GV_WORDS words. Evaluation uses an f32 stack. */
enum { MGX_GV_OP = 0, MGX_GV_A0, MGX_GV_A1, MGX_GV_A2, MGX_GV_WORDS };
enum {
  MGX_GOP_INVENTORY = 0, /* a0 resource: push f32(entity.inventory[resource]) (0 when the entity is null)        */
  MGX_GOP_STAT,          /* a0 scope (0 agent, 1 game), a1 stat id: touch the key, push its value                 */
  MGX_GOP_CONST,         /* a0 f32 bits                                                                            */
  MGX_GOP_ADD_TERM,      /* SumValue step: t=pop; if a0: t=logf(t+1); if a1: t*=f32(a2); acc=pop; push(acc+t)     */
  MGX_GOP_RATIO,         /* den=pop, num=pop; push(den>0 ? num/den : num)                                          */
  MGX_GOP_MAX2,          /* v=pop, best=pop; push(std::max(best,v))                                                */
  MGX_GOP_MIN2,
  MGX_GOP_QUERY_INVENTORY, /* a0 resource, a1 query: sum of f32(amount) over the query result (game_value.cpp:45-57) */
  MGX_GOP_QUERY_COUNT      /* a0 query */
};
/* A "value code" argument elsewhere is the index of a MGX_SEC_OBS_VALUES-style (start,count) pair:
 * MGX_SEC_REWARDS record = {gv_start, gv_count, accumulate, init_touch_stat_scope, init_touch_stat_id} */
enum { MGX_RW_GV_START = 0, MGX_RW_GV_COUNT, MGX_RW_ACCUMULATE, MGX_RW_TOUCH_SCOPE, MGX_RW_TOUCH_STAT, MGX_RW_WORDS };
/* MGX_SEC_OBS_VALUES record = {gv_start, gv_count, feature_id}; also used as the generic "value code" table:
 * filters/mutations reference values by record index in this section. */
enum { MGX_OV_GV_START = 0, MGX_OV_GV_COUNT, MGX_OV_FEATURE, MGX_OV_WORDS };

/* Queries (core/query_config.hpp, core/query_system.cpp:178-330). */
enum { MGX_Q_KIND = 0, MGX_Q_A0, MGX_Q_A1, MGX_Q_A2, MGX_Q_A3, MGX_Q_MAX_ITEMS /* value record or -1 */,
       MGX_Q_ORDER /* bit0 random, bit8 include_blocker */, MGX_Q_A4, MGX_Q_WORDS };
enum {
  MGX_QK_TAG = 0,   /* a0 tag, a1 filter pc */
  MGX_QK_FILTERED,  /* a0 source query, a1 filter pc */
  MGX_QK_CLOSURE,   /* a0 source, a1 candidates or -1, a2 edge filter pc, a3 result filter pc */
  MGX_QK_RAYCAST    /* a0 source, a1 max_range value record, a2 WORDLIST (dr, dc) pairs, a3 pair count, a4 blocker pc */
};
/* Events (handler/event.cpp:34-101, event_scheduler.cpp:8-53). Schedule = (timestep, event) sorted as the reference. */
enum { MGX_EV_QUERY = 0, MGX_EV_FILTER_PC, MGX_EV_MUT_START, MGX_EV_MUT_COUNT, MGX_EV_MAX_TARGETS, MGX_EV_FALLBACK,
       MGX_EV_PAD0, MGX_EV_PAD1, MGX_EV_WORDS };
enum { MGX_SC_TIMESTEP = 0, MGX_SC_EVENT, MGX_SC_WORDS };
enum { MGX_MQ_TAG = 0, MGX_MQ_QUERY, MGX_MQ_WORDS };
enum { MGX_TH_TAG = 0, MGX_TH_HANDLER, MGX_TH_WORDS };  /* on_tag_add / on_tag_remove (core/grid_object.cpp:93-123) */
/* AoE (handler/handler_config.hpp:55-66, core/aoe_tracker.cpp). */
enum { MGX_AO_RADIUS = 0, MGX_AO_STATIC, MGX_AO_EFFECT_SELF, MGX_AO_FILTER_PC, MGX_AO_MUT_START, MGX_AO_MUT_COUNT,
       MGX_AO_PRES_START, MGX_AO_PRES_COUNT, MGX_AO_WORDS };
enum { MGX_PR_RESOURCE = 0, MGX_PR_DELTA, MGX_PR_WORDS };
/* Territories (handler/territory_config.hpp, core/territory_tracker.cpp). Handler lists = consecutive leaf handlers. */
enum { MGX_TE_TAGS_START = 0 /* WORDLIST */, MGX_TE_TAGS_COUNT, MGX_TE_ENTER_START, MGX_TE_ENTER_COUNT,
       MGX_TE_EXIT_START, MGX_TE_EXIT_COUNT, MGX_TE_PRES_START, MGX_TE_PRES_COUNT, MGX_TE_WORDS };
enum { MGX_TC_STRENGTH = 0, MGX_TC_DECAY, MGX_TC_TERRITORY, MGX_TC_PAD, MGX_TC_WORDS };

static inline int mgx_sec_off(const int32_t* p, int s) { return p[MGX_H_SECTION_BASE + 2 * s]; }
static inline int mgx_sec_cnt(const int32_t* p, int s) { return p[MGX_H_SECTION_BASE + 2 * s + 1]; }

#endif /* MGX_PROGRAM_H_ */
