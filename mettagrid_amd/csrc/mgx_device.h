// mgx_device.h — device-side state layout and small helpers of the MI355X MettaGrid step engine.
//
// State lives in HBM for the whole episode.  Two layouts are used on purpose (see DESIGN.md "Data layout"):
//   * ENV-MAJOR blocks  [env][...]  for spatial/object/agent state: the observation kernel maps one workgroup to
//     one env and streams that env's grid block with coalesced 16-byte loads into LDS; the world-update kernel
//     (one env per wavefront lane) touches only a few dozen scattered cells/records per env and step.
//   * ENV-MINOR arrays  [word][env]  for the per-env Mersenne-Twister state: all 64 lanes of a wavefront read the
//     same word index of 64 consecutive envs -> one coalesced 256-byte access per draw.
#ifndef MGX_DEVICE_H_
#define MGX_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgx_program.h"

#define MGX_WAVE 64
#define MGX_NO_AGENT 0xFF
#define MGX_DEAD_CLASS 0xFFFF
#define MGX_AOE_MAX_STATS 32
#define MGX_INV_PITCH 16  // u16 entries per inventory row: a row is two aligned 16-byte loads (R <= 13 resources)

struct MgxDev {
  const int32_t* P;       // program blob (device copy)
  int sec[MGX_SEC_COUNT]; // section offsets (words) — hoisted from the header
  int E, H, W, A, S, R, T;
  int NS, NG, NSW, NGW;   // agent/game stat counts and their touched-bit word counts
  int NSP;                // pitch (floats) of one agent's stat row: NS rounded up to 32 = 128-byte aligned rows
  int NRW;                // reward "prev value" slots per agent (max over classes)
  int SEENW;              // words of the per-agent visited-cell bitmap
  int NOFF;               // observation offsets
  int base;               // token_value_base
  int max_steps, truncates, max_priority, nact, flags, hp_res, n_obs_values, n_move_handlers, any_on_tick;
  int tick_in_aoe;        // 1: mgx_aoe_kernel runs the per-agent on_tick handlers (one lane per agent) in front of the area effects
  int cov_in_aoe;         // 1: ... and the coverage tracking behind them (no tail launch)
  int flat_top;           // 1: top-level handlers of the action phase run on the register VM (MgxEnvT::apply_top)
  int hot_lo;             // first program word of the LDS copy (extended world kernel: sections LIMITS .. TERR_CONTROLS; their
                          // sec[] entries are relative to it in that kernel's copy of this table)
  int x_aoe_lds;          // 1: the extended world kernel runs the AoE phase itself and keeps its scratch in LDS
  int defer_book;         // 1: per-action bookkeeping stats are applied in one batched pass at the end of the tick
  int shadow;             // bit 0: ... and into the integer record ag_cnt instead of the agents' stat rows; the float cells are
                          // written from it (mgx_shadow_flush_kernel) before anything reads them.  bit 1: the two coverage stats
                          // likewise (from ag_unique / ag_maxdist; lean lane-per-env dispatch).  0, 1 (lane-per-agent dispatch) or 3
  int gen_prog;           // 3 / 4: the program's handler tables equal the preset the build generated straight-line code for (0: none)
  int act_par;            // 1: the action dispatch runs in mgx_act_kernel (one lane per AGENT, conflict-ordered rounds; mgx_act.h)
  int act_tick;           // 1: ... and the per-agent on_tick handlers too (lean games), one lane per agent
  int duo;                // 1: the lean lane-per-env kernel dispatches TWO agents of an env at a time — the env's own lane the next
                          //    agent in order, its helper lane (MGX_WORLD_HELPERS) the one after it when their cell footprints are
                          //    disjoint (host analysis like act_par: handlers stay with actor and target, move range 1)
  int tick_split;         // 1: every per-agent on_tick handler touches its own agent only: in the lean lane-per-env kernel the
                          //    env's helper lane (MGX_WORLD_HELPERS) runs them for the second half of the agents
  int act_ngset;          // game-scope stats the action-phase handlers SET (StatsMutation): applied in agent order at the end
  int act_gset_ids[4];
  int act_lds_extra;      // bytes per workgroup behind the game-stat cells: footprints u32[A'][EPG] | cell map u8[EPG][H*W] (0: no cell map)
  int act_replay;         // 1 (env MGX_ACT_SHUFFLE_REPLAY): the shuffle always takes its serial replay path (tests)
  int act_map;            // 1: conflicts are looked up through the cell map (H*W <= 4096); 0: all-pairs walk over the pending lanes
  int feat[16];
  int wk[32];             // well-known stat ids (MGX_S_*)

  // ---- static per-class observation tokens (built by the host at mgx_create) ----
  const uint32_t* cls_tokinfo;  // [classes] start(16) | group(8) | ntags(6) | is_agent(1) | is_static(1)
  const uint16_t* cls_tok;      // tag tokens: feature | tag id << 8
  // ---- env-major object state ----
  uint16_t* grid;         // [E][H*W]   0 = empty, else slot + 1
  uint16_t* obj_cls;      // [E][S]     class id, MGX_DEAD_CLASS = unused slot
  uint16_t* obj_rc;       // [E][S]     (r << 8) | c
  uint8_t* obj_vibe;      // [E][S]
  uint8_t* obj_agent;     // [E][S]     agent index or MGX_NO_AGENT
  uint32_t* obj_visited;  // [E][S]     GridObject::visited (core/grid_object.hpp:113)
  uint16_t* obj_inv;      // [E][S][MGX_INV_PITCH]  amount by resource id (entries >= R unused)
  unsigned long long* obj_order;  // [E][S] inventory iteration order: 4-bit ids, begin() in the low nibble, 0xF ends
  uint32_t* num_objs;     // [E]
  // ---- env-major agent state ----
  uint16_t* ag_obj;       // [E][A] slot of agent i
  uint16_t* ag_prev;      // [E][A] Agent::prev_location
  uint16_t* ag_spawn;     // [E][A]
  uint16_t* ag_rc;        // [E][A] the agent's cell = obj_rc of its object, kept beside it by every writer (move_object, the swap
                          // mutation, construction): staging an env's agents is then ONE round of loads, not slot -> object row
  uint16_t* ag_cls;       // [E][A] class of the agent's object (an object never changes class)
  uint32_t* ag_rwinfo;    // [E][A] reward records of the agent's class: start | count << 16 (an object never changes class)
  uint16_t* ag_stepprev;  // [E][A] MettaGrid::_prev_agent_locations
  uint16_t* ag_covrc;     // [E][A] position at the last coverage update (0xFFFF = never)
  int32_t* ag_invk;       // [E][A][MGX_INVALID_EXTRA] out-of-window invalid action indices seen this episode ...
  float* ag_invn;         // ... and how often (0 = free pair)
  uint32_t* ag_swm;       // [E][A] steps_without_motion
  uint32_t* ag_cnt;       // [E][A][8] (32-byte records) noop ok/failed, move ok/failed, change_vibe ok/failed, action.failed,
                          // max steps without motion — what the per-action bookkeeping counts, as integers (d.shadow)
  uint32_t* ag_maxdist;   // [E][A]
  uint32_t* ag_unique;    // [E][A]
  uint32_t* ag_seen;      // [E][A][SEENW]
  float* ag_rprev;        // [E][A][NRW]
  float* ag_stats;        // [E][A][NSP]
  uint32_t* ag_touched;   // [E][A][NSW]
  float* game_stats;      // [E][NG]
  uint32_t* game_touched; // [E][NGW]
  uint32_t* step;         // [E]
  uint32_t* err;          // [E]
  int32_t* executed;      // [E][A] action executed this step (0 = noop / failed)
  uint8_t* success;       // [E][A]
  float* episode_rewards; // [E*A]
  // ---- extended state (rung 4: dynamic tags, tag index, AoE, territory, events, query workspace) ----
  int X;                  // 1: extended kernels are in use
  int NL;                 // indexed tags (TagIndex lists)
  int NF, NM, NTS, NT;    // capacities: fixed AoE sources, mobile AoE sources, territory sources; territory types
  int AW;                 // words of a bitmap over the env's agents
  int FW, MW;             // words of one agent's bitmap over the fixed / mobile AoE sources (fx_inside / mb_inside)
  // lane-per-agent area-effect kernel (mgx_aoe_local.h): the agent stats its records can touch, staged per lane
  int aoe_nstat;                 // number of such stat ids (<= MGX_AOE_MAX_STATS)
  const int16_t* aoe_stat_ids;   // [aoe_nstat] stat id of local index k
  const uint8_t* aoe_stat_map;   // [256] stat id -> local index, 0xFF = not staged (read / written in HBM directly)
  int QB, QD;             // query workspace buffers per env, max query nesting
  int SW;                 // words of a per-env object bitmap
  int game_on_tick, n_events, n_schedule, n_matq, aoe_mask_feat;
  uint32_t* obj_tags;     // [E][S][8]   GridObject::tag_bits
  uint16_t* tl_items;     // [E][NL][S]  TagIndex lists in registration order
  uint16_t* tl_count;     // [E][NL]
  uint16_t* fx_obj;       // [E][NF] fixed AoE sources in registration order
  uint16_t* fx_aoe;       // [E][NF]
  uint16_t* fx_rc;        // [E][NF] location at registration
  uint32_t* fx_inside;    // [E][A][FW]  bit f of agent ai: ai is inside fixed source f
  uint32_t* fx_pack;      // [E][NF] rc | radius << 16 | live << 24, rebuilt every step for the lane-per-agent AoE kernel
  uint16_t* fx_count;     // [E]
  uint16_t* mb_obj;       // [E][NM] mobile AoE sources
  uint16_t* mb_aoe;       // [E][NM]
  uint32_t* mb_inside;    // [E][A][MW]
  uint32_t* mb_pack;      // [E][NM] like fx_pack, with the source's CURRENT location
  uint16_t* mb_count;     // [E]
  uint16_t* ts_obj;       // [E][NTS] territory sources in registration order
  uint16_t* ts_ctrl;      // [E][NTS]
  uint16_t* ts_rc;        // [E][NTS] registered location
  uint16_t* ts_count;     // [E]
  int16_t* terr_prev;     // [E][A][NT] previous owner tag or -1
  uint32_t* obsval;       // [E][A][n_obs_values] query-backed global obs values (mgx_values_kernel), else nullptr
  uint16_t* terr_owner;   // [E][NT][H*W] cached cell ownership (winning tag, 0xFFFF = none), rebuilt by mgx_terr_kernel
  uint8_t* terr_dirty;    // [E] 1: a territory source moved / changed tags since the map was built
  uint32_t* next_event;   // [E]
  uint8_t* obj_flags;     // [E][S] bit0 removed from the grid, bit1 created at run time (no inventory tokens)
  uint16_t* def_aoe;      // [E][S] objects spawned this tick whose AoEs register at flush_deferred
  uint16_t* def_count;    // [E]
  uint16_t* qws;          // [E][QB][S] query workspace
  uint32_t* qvis;         // [E][QD+1][SW] visited bitmaps
  // ---- env-minor RNG ----
  uint32_t* mt;           // [624][E]
  uint32_t* mt_idx;       // [E]
  // ---- caller-visible buffers (device pointers) ----
  uint8_t* obs;           // [E*A][T][3]
  float* rewards;         // [E*A]
  uint8_t* terminals;     // [E*A]
  uint8_t* truncations;   // [E*A]
  const int32_t* actions;       // [E*A]
  const int32_t* vibe_actions;  // [E*A]
};

// How device code reaches the engine's MgxDev: every kernel takes it BY VALUE (it lives in the kernarg segment, so inside
// the kernel and everything inlined into it each field is a scalar, invariant s_load) and MgxEnvT keeps a reference to
// that argument.  Per launch, per engine: there is no process-wide __constant__ copy to keep in step (round 1 had one
// and had to drain the device whenever two engines alternated).
// NOTE (measured, gfx950 / ROCm 7.2): __builtin_amdgcn_kernarg_segment_ptr() must not be used to reach it from
// functions that may stay out of line — outside the kernel function itself the compiler folds that intrinsic to a null
// pointer (the callee is never passed the kernarg pointer), and the first field access faults at address 0.
#define MGX_KERNARG_ENTRY(dv) ((void)0)
// MGX_CONST_DEV (the lean world kernel's translation unit only): that kernel keeps the engine's MgxDev in constant
// memory instead — measured 0.44 ms against 0.67 ms per step with the by-value argument (rung 3, 65 536 envs: the
// argument form keeps ~70 fields live in SGPRs across a 30 000-instruction kernel and spills them).  The unit is
// compiled once per MGX_SLOT (two slots), engines take slots round-robin, and the host keeps a slot's symbol equal to
// the launching engine's table (content compare; a change first waits for the kernels still reading the old content),
// so two engines alive in one process (train + eval) do not disturb each other.
#ifdef MGX_CONST_DEV
#ifdef MGX_JIT_UNIT   // a code object loaded at run time (mgx_jit_world.hip): the host reaches the symbol through hipModuleGetGlobal
__constant__ MgxDev g_mgx_dev;
#else
static __constant__ MgxDev g_mgx_dev;
#endif
#endif

template <class D> __device__ __forceinline__ const int32_t* mgx_cls(const D& d, int cls) {
  return d.P + d.sec[MGX_SEC_CLASSES] + cls * MGX_C_WORDS;
}

// glibc 2.35 logf restated (sysdeps/ieee754/flt-32/e_logf.c; table e_logf_data.c).  The reference calls
// std::log(float) in SumValue(log=True) (cpp/src/mettagrid/core/game_value.cpp:77-79) and the parity signature
// exposes every bit of it.  tests/golden/logf_check.c proves this restatement equal to the host libm for all
// 2 139 095 039 positive finite floats.  Double arithmetic with contraction disabled.
static __device__ __noinline__ float mgx_logf(float x) {
#pragma clang fp contract(off)
  static const double INVC[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0,  0x1.3c995b0b80385p+0,
                           0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,  0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
                           0x1.0953f419900a7p+0, 0x1p+0,               0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
                           0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
  static const double LOGC[16] = {-0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3,
                           -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3,   -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4,
                           -0x1.252f438e10c1ep-5, 0x0p+0,                0x1.aa5aa5df25984p-5,  0x1.c5e53aa362eb4p-4,
                           0x1.526e57720db08p-3,  0x1.bc2860d22477p-3,   0x1.1058bc8a07ee1p-2,  0x1.4043057b6ee09p-2};
  uint32_t ix = __float_as_uint(x);
  if (ix == 0x3f800000u) return 0.0f;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
    if (ix * 2u == 0u) return -__builtin_inff();
    if (ix == 0x7f800000u) return x;
    if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return __builtin_nanf("");
    ix = __float_as_uint(x * 0x1p23f);
    ix -= 23u << 23;
  }
  uint32_t tmp = ix - 0x3f330000u;
  int i = (int)((tmp >> 19) & 15u);
  int k = (int32_t)tmp >> 23;
  uint32_t iz = ix - (tmp & 0xff800000u);
  double z = (double)__uint_as_float(iz);
  double r = z * INVC[i] - 1.0;
  double y0 = LOGC[i] + (double)k * 0x1.62e42fefa39efp-1;
  double r2 = r * r;
  double y = 0x1.5575b0be00b6ap-2 * r + -0x1.ffffef20a4123p-2;
  y = -0x1.00ea348b88334p-2 * r2 + y;
  y = y * r2 + (y0 + r);
  return (float)y;
}

#endif  // MGX_DEVICE_H_
