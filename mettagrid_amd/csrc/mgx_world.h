// mgx_world.h — world-update kernel: one env per wavefront lane.
//
// Covers phases 1-11 of MettaGrid::_step (/root/reference/cpp/bindings/mettagrid_c.cpp:921-1056): snapshot of
// previous locations, step counter, the per-step agent shuffle, action dispatch by priority and stream
// (noop / move with the handler-chain line scan / change_vibe), per-agent on_tick handlers and coverage tracking.
// The order of agents inside one env is a true sequential dependence (collisions, first-come resources), so an env
// is executed serially by ONE lane; 64 envs advance in lock-step per wavefront and every lane runs the same
// compiled program, which bounds divergence.  Observations, rewards and termination are mgx_obs.h.
#ifndef MGX_WORLD_H_
#define MGX_WORLD_H_

#include "mgx_device.h"

struct MgxCtx {  // handler/handler_context.hpp:38-112 (the fields the supported filters/mutations read)
  int actor, target;  // object slots, -1 = null
  int target_r, target_c;
  int move_direction;
  bool mutation_failed;
};

// Program pointer types: the world kernel keeps the whole program in LDS when it fits (address space 3 pointer ->
// ds_read, no vector-memory traffic for table lookups); everything else reads it from global memory.
typedef const int32_t* MgxGlobalProg;
typedef const __attribute__((address_space(3))) int32_t* MgxLdsProg;

template <class PP>
struct MgxEnvT {  // per-lane view of one env
  const MgxDev& d;
  PP P;
  int env;
  uint32_t step;
  __device__ MgxEnvT(const MgxDev& dd, PP prog, int e) : d(dd), P(prog), env(e), step(0) {}
  __device__ __forceinline__ PP cls(int c) const { return P + d.sec[MGX_SEC_CLASSES] + c * MGX_C_WORDS; }

  __device__ __forceinline__ size_t so(int slot) const { return (size_t)env * d.S + slot; }
  __device__ __forceinline__ size_t ao(int agent) const { return (size_t)env * d.A + agent; }
  __device__ __forceinline__ uint16_t& inv(int slot, int item) const { return d.obj_inv[so(slot) * d.R + item]; }
  __device__ __forceinline__ PP cls_of(int slot) const { return cls(d.obj_cls[so(slot)]); }
  __device__ __forceinline__ int agent_of(int slot) const {
    if (slot < 0) return -1;
    int a = d.obj_agent[so(slot)];
    return a == MGX_NO_AGENT ? -1 : a;
  }
  __device__ __forceinline__ void flag(uint32_t bit) const { d.err[env] |= bit; }

  // ---- stats (systems/stats_tracker.hpp:69-90) ----
  __device__ void astat_touch(int agent, int id) const { d.ag_touched[ao(agent) * d.NSW + (id >> 5)] |= 1u << (id & 31); }
  // add(): every caller adds a strictly positive amount to a key that only ever grows, so "value != 0" already
  // says the key exists; the host ORs that into the touched flags (mgx_get_stats) and the bit RMW is skipped here.
  __device__ void astat_add(int agent, int id, float v) const {
    if (id < 0) return;
    d.ag_stats[ao(agent) * d.NS + id] += v;
  }
  __device__ void astat_set(int agent, int id, float v) const {
    if (id < 0) return;
    d.ag_stats[ao(agent) * d.NS + id] = v;
    astat_touch(agent, id);
  }
  __device__ float astat_get(int agent, int id) const { return id < 0 ? 0.f : d.ag_stats[ao(agent) * d.NS + id]; }
  __device__ void gstat_touch(int id) const { d.game_touched[(size_t)env * d.NGW + (id >> 5)] |= 1u << (id & 31); }
  __device__ void gstat_set(int id, float v) const {
    if (id < 0) return;
    d.game_stats[(size_t)env * d.NG + id] = v;
    gstat_touch(id);
  }
  __device__ void gstat_add(int id, float v) const {
    if (id < 0) return;
    d.game_stats[(size_t)env * d.NG + id] += v;
    gstat_touch(id);
  }

  // ---- inventory (cpp/src/mettagrid/objects/inventory.cpp) ----
  __device__ int effective_limit(int slot, PP L) const {  // objects/inventory.hpp:26-40
    int sum = 0;
    PP mods = P + d.sec[MGX_SEC_MODS] + L[MGX_L_MOD_START] * MGX_MOD_WORDS;
    for (int i = 0; i < L[MGX_L_MOD_COUNT]; i++)
      sum += (int)inv(slot, mods[i * MGX_MOD_WORDS + MGX_MOD_ITEM]) * mods[i * MGX_MOD_WORDS + MGX_MOD_BONUS];
    int eff = min(L[MGX_L_MAX], max(L[MGX_L_MIN], sum));
    return min(max(eff, 0), 65535);
  }
  __device__ int group_amount(int slot, PP L) const {
    int s = 0;
    uint32_t mask = (uint32_t)L[MGX_L_RES_MASK];
    while (mask) {
      int r = __ffs(mask) - 1;
      mask &= mask - 1;
      s += inv(slot, r);
    }
    return s;
  }
  __device__ PP limit_of(PP C, int item) const {
    int li = C[MGX_C_RES_LIMIT + item];
    return li < 0 ? (PP)nullptr : P + d.sec[MGX_SEC_LIMITS] + li * MGX_L_WORDS;
  }
  __device__ void on_inventory_change(int slot, int item, int delta, int amount) const {  // objects/agent.cpp:106-121
    int a = agent_of(slot);
    if (a < 0 || delta == 0) return;
    if (delta > 0) astat_add(a, mgx_wk(d, MGX_S_RES_GAINED_BASE) + item, (float)delta);
    else astat_add(a, mgx_wk(d, MGX_S_RES_LOST_BASE) + item, (float)(-delta));
    astat_set(a, mgx_wk(d, MGX_S_RES_AMOUNT_BASE) + item, (float)amount);
    if (amount == 0 && delta < 0 && item == d.hp_res) astat_add(a, mgx_wk(d, MGX_S_DEATH), 1.f);
  }
  // Inventory::update (inventory.cpp:38-86).  DEPTH bounds the update -> enforce_all_limits -> update recursion.
  template <int DEPTH>
  __device__ int inv_update(int slot, int item, int delta, bool ignore_limits = false, bool notify = true) const {
    PP C = cls_of(slot);
    int initial = inv(slot, item);
    int new_amount = initial + delta;
    int mx = 65535;
    if (!ignore_limits) {
      PP L = limit_of(C, item);
      if (L) {
        int used = group_amount(slot, L) - initial;
        if (used < 0) used = 0;
        int m = effective_limit(slot, L) - used;
        mx = m < 0 ? 0 : m;
      }
    }
    int clamped = min(max(new_amount, 0), mx);
    if (clamped != initial) {
      unsigned long long ord = d.obj_order[so(slot)];
      if (initial == 0) {  // new node goes to the list head (libstdc++ _M_insert_bucket_begin; mettagrid_amd/umap.py)
        ord = (ord << 4) | (unsigned long long)item;
        d.obj_order[so(slot)] = ord;
      } else if (clamped == 0) {  // erase keeps the order of the rest
        int p = 0;
        while (p < 16 && ((ord >> (4 * p)) & 0xF) != (unsigned long long)item) p++;
        unsigned long long low = p ? (ord & ((1ull << (4 * p)) - 1ull)) : 0ull;
        unsigned long long high = p >= 15 ? 0ull : (ord >> (4 * (p + 1)));
        d.obj_order[so(slot)] = low | (high << (4 * p)) | (0xFull << 60);
      }
      inv(slot, item) = (uint16_t)clamped;
    }
    int dl = clamped - initial;
    if (notify && dl != 0) on_inventory_change(slot, item, dl, clamped);
    if (dl < 0 && (C[MGX_C_MODIFIER_MASK] & (1 << item))) {
      if constexpr (DEPTH > 0) enforce_all_limits<DEPTH - 1>(slot, C);
      else flag(4u);
    }
    return dl;
  }
  template <int DEPTH>
  __device__ void enforce_all_limits(int slot, PP C) const {  // inventory.cpp:141-173
    for (int li = 0; li < C[MGX_C_LIMIT_COUNT]; li++) {
      PP L = P + d.sec[MGX_SEC_LIMITS] + (C[MGX_C_LIMIT_START] + li) * MGX_L_WORDS;
      if (L[MGX_L_DROP_COUNT] == 0) continue;
      int excess = group_amount(slot, L) - effective_limit(slot, L);
      if (excess <= 0) continue;
      PP drop = P + d.sec[MGX_SEC_DROP_ORDER] + L[MGX_L_DROP_START];
      for (int k = 0; k < L[MGX_L_DROP_COUNT]; k++) {
        int item = drop[k];
        int to_drop = min((int)inv(slot, item), excess);
        if (to_drop > 0) {
          inv_update<DEPTH>(slot, item, -to_drop);
          excess = group_amount(slot, L) - effective_limit(slot, L);
        }
        if (excess <= 0) break;
      }
    }
  }
  __device__ int free_space(int slot, int item) const {  // inventory.cpp:97-110
    PP L = limit_of(cls_of(slot), item);
    if (!L) return 65535 - inv(slot, item);
    int used = group_amount(slot, L), eff = effective_limit(slot, L);
    return eff > used ? eff - used : 0;
  }
  __device__ int transfer(int src, int dst, int item, int delta) const {  // objects/has_inventory.cpp:76-108
    if (delta <= 0) return 0;
    int give = min((int)inv(src, item), delta);
    int amount = min(give, free_space(dst, item));
    inv_update<1>(src, item, -amount);
    inv_update<1>(dst, item, amount);
    return amount;
  }

  // ---- game values (cpp/src/mettagrid/core/game_value.cpp:14-148): postfix code on a small f32 stack ----
  __device__ float eval_code(int start, int count, int entity) const {
    float st[8];
    int sp = 0;
    PP code = P + d.sec[MGX_SEC_GV_CODE] + start * MGX_GV_WORDS;
    for (int i = 0; i < count; i++, code += MGX_GV_WORDS) {
      int a0 = code[MGX_GV_A0], a1 = code[MGX_GV_A1], a2 = code[MGX_GV_A2];
      switch (code[MGX_GV_OP]) {
        case MGX_GOP_INVENTORY: st[sp++ & 7] = entity >= 0 ? (float)inv(entity, a0) : 0.f; break;
        case MGX_GOP_STAT: {
          float v = 0.f;
          if (a0 == 1) { gstat_touch(a1); v = d.game_stats[(size_t)env * d.NG + a1]; }
          else { int a = agent_of(entity); if (a >= 0) { astat_touch(a, a1); v = astat_get(a, a1); } }
          st[sp++ & 7] = v;
          break;
        }
        case MGX_GOP_CONST: st[sp++ & 7] = __int_as_float(a0); break;
        case MGX_GOP_ADD_TERM: {
          float t = st[--sp & 7];
          if (a0) t = mgx_logf(__fadd_rn(t, 1.0f));
          if (a1) t = __fmul_rn(t, __int_as_float(a2));
          float acc = st[--sp & 7];
          st[sp++ & 7] = __fadd_rn(acc, t);
          break;
        }
        case MGX_GOP_RATIO: { float den = st[--sp & 7], num = st[--sp & 7]; st[sp++ & 7] = den > 0.f ? __fdiv_rn(num, den) : num; break; }
        case MGX_GOP_MAX2: { float v = st[--sp & 7], b = st[--sp & 7]; st[sp++ & 7] = (b < v) ? v : b; break; }
        case MGX_GOP_MIN2: { float v = st[--sp & 7], b = st[--sp & 7]; st[sp++ & 7] = (v < b) ? v : b; break; }
      }
    }
    return sp > 0 ? st[(sp - 1) & 7] : 0.f;
  }
  __device__ float eval_value(int rec, int entity) const {
    PP V = P + d.sec[MGX_SEC_OBS_VALUES] + rec * MGX_OV_WORDS;
    return eval_code(V[MGX_OV_GV_START], V[MGX_OV_GV_COUNT], entity);
  }

  // ---- filters (handler/filters/*.hpp) as short-circuit code ----
  __device__ __forceinline__ int resolve(const MgxCtx& c, int ent) const { return ent == MGX_ENT_ACTOR ? c.actor : c.target; }
  __device__ bool atom(PP a, const MgxCtx& c) const {
    int a0 = a[MGX_AT_A0], a1 = a[MGX_AT_A1], a2 = a[MGX_AT_A2];
    switch (a[MGX_AT_OP]) {
      case MGX_FOP_VIBE: { int e = resolve(c, a0); return e >= 0 && d.obj_vibe[so(e)] == a1; }
      case MGX_FOP_RESOURCE: { int e = resolve(c, a0); return e >= 0 && (int)inv(e, a1) >= a2; }
      case MGX_FOP_SHARED_TAG: {
        if (c.actor < 0 || c.target < 0) return false;
        PP mask = P + d.sec[MGX_SEC_WORDLIST] + a0;
        PP x = cls_of(c.actor) + MGX_C_TAGS; PP y = cls_of(c.target) + MGX_C_TAGS;
        uint32_t any = 0;
        for (int w = 0; w < MGX_TAG_WORDS; w++) any |= (uint32_t)x[w] & (uint32_t)y[w] & (uint32_t)mask[w];
        return any != 0;
      }
      case MGX_FOP_TAG: {
        int e = resolve(c, a0);
        if (e < 0) return false;
        PP mask = P + d.sec[MGX_SEC_WORDLIST] + a1;
        PP x = cls_of(e) + MGX_C_TAGS;
        uint32_t any = 0;
        for (int w = 0; w < MGX_TAG_WORDS; w++) any |= (uint32_t)x[w] & (uint32_t)mask[w];
        return any != 0;
      }
      case MGX_FOP_TARGET_LOC_EMPTY: return c.target < 0;
      case MGX_FOP_TARGET_IS_USABLE: return c.target >= 0;
      case MGX_FOP_PERIODIC: return step >= (uint32_t)a1 && ((step - (uint32_t)a1) % (uint32_t)a0) == 0;
      case MGX_FOP_GAME_VALUE: {
        int e = resolve(c, a0);
        float v = eval_value(a1, e);
        float t = eval_value(a2, e);
        return v >= t;
      }
      case MGX_FOP_TRUE: return true;
      default: return false;
    }
  }
  __device__ bool check_filters(int pc, const MgxCtx& c) const {  // handler/handler.cpp:95-103
    PP atoms = P + d.sec[MGX_SEC_ATOMS];
    while (pc >= 0) {
      PP a = atoms + pc * MGX_AT_WORDS;
      pc = atom(a, c) ? a[MGX_AT_ON_TRUE] : a[MGX_AT_ON_FALSE];
    }
    return pc == MGX_PC_PASS;
  }

  // ---- grid (core/grid.hpp:75-113) ----
  __device__ __forceinline__ uint16_t& cell(int r, int c) const { return d.grid[(size_t)env * d.H * d.W + r * d.W + c]; }
  __device__ bool move_object(int slot, int r, int c) const {
    if (r < 0 || c < 0 || r >= d.H || c >= d.W) return false;
    if (cell(r, c) != 0) return false;
    uint16_t rc = d.obj_rc[so(slot)];
    cell(r, c) = (uint16_t)(slot + 1);
    cell(rc >> 8, rc & 0xFF) = 0;
    d.obj_rc[so(slot)] = (uint16_t)((r << 8) | c);
    return true;
  }

  // ---- mutations (handler/mutations/*.hpp) ----
  template <int DEPTH>
  __device__ void mutate(PP m, MgxCtx& c) const {
    int a0 = m[MGX_MU_A0], a1 = m[MGX_MU_A1], a2 = m[MGX_MU_A2], a3 = m[MGX_MU_A3];
    switch (m[MGX_MU_OP]) {
      case MGX_MOP_RESOURCE_DELTA: { int e = resolve(c, a0); if (e >= 0) inv_update<1>(e, a1, a2); break; }
      case MGX_MOP_RESOURCE_TRANSFER: {  // resource_mutation.hpp:60-98
        int s = resolve(c, a0), t = resolve(c, a1);
        if (s < 0 || t < 0) break;
        int amount = a3 < 0 ? (int)inv(s, a2) : a3;
        int moved = transfer(s, t, a2, amount);
        int sa = agent_of(s);
        if (moved > 0 && sa >= 0) astat_add(sa, mgx_wk(d, MGX_S_RES_DEPOSITED_BASE) + a2, (float)moved);
        break;
      }
      case MGX_MOP_CLEAR_INVENTORY: {  // resource_mutation.hpp:111-128
        int e = resolve(c, a0);
        if (e < 0) break;
        if (a2 == 0) {
          unsigned long long ord = d.obj_order[so(e)];  // iterate a copy, as the reference does
          for (int k = 0; k < 16; k++) {
            int item = (int)((ord >> (4 * k)) & 0xF);
            if (item == 0xF) break;
            inv_update<1>(e, item, -(int)inv(e, item));
          }
        } else {
          PP ids = P + d.sec[MGX_SEC_WORDLIST] + a1;
          for (int i = 0; i < a2; i++) inv_update<1>(e, ids[i], -(int)inv(e, ids[i]));
        }
        break;
      }
      case MGX_MOP_ATTACK: {  // attack_mutation.hpp:20-38
        if (c.actor < 0 || c.target < 0) break;
        int weapon = inv(c.actor, a0), armor = inv(c.target, a1);
        int dmg = max(0, (weapon * a3) / 100 - armor);
        if (dmg > 0) inv_update<1>(c.target, a2, -dmg);
        break;
      }
      case MGX_MOP_STATS: {  // stats_mutation.hpp:21-41
        int e = resolve(c, a1);
        float v = eval_value(a3, e);
        if (a0 == 0) gstat_set(a2, v);
        else { int a = agent_of(e); if (a >= 0) astat_set(a, a2, v); }
        break;
      }
      case MGX_MOP_CHANGE_VIBE: { int e = resolve(c, a0); if (e >= 0) d.obj_vibe[so(e)] = (uint8_t)a1; break; }
      case MGX_MOP_RELOCATE: if (agent_of(c.actor) >= 0) move_object(c.actor, c.target_r, c.target_c); break;
      case MGX_MOP_SWAP: {  // swap_mutation.hpp:15-21, core/grid.hpp:92-105
        int xa = agent_of(c.actor), ya = agent_of(c.target);
        if (xa < 0 || ya < 0) break;
        uint16_t rx = d.obj_rc[so(c.actor)], ry = d.obj_rc[so(c.target)];
        cell(rx >> 8, rx & 0xFF) = (uint16_t)(c.target + 1);
        cell(ry >> 8, ry & 0xFF) = (uint16_t)(c.actor + 1);
        d.obj_rc[so(c.actor)] = ry;
        d.obj_rc[so(c.target)] = rx;
        astat_add(xa, mgx_wk(d, MGX_S_SWAP), 1.f);
        break;
      }
      case MGX_MOP_USE_TARGET: {  // use_target_mutation.hpp:17-29, core/grid_object.cpp:72-80
        if (c.target < 0 || agent_of(c.actor) < 0) { c.mutation_failed = true; break; }
        int h = cls_of(c.target)[MGX_C_ON_USE];
        bool ok = false;
        if constexpr (DEPTH > 0) {
          if (h >= 0) { MgxCtx use = c; ok = apply_handler<DEPTH - 1>(h, use); }
        } else {
          flag(4u);
        }
        if (!ok) { c.mutation_failed = true; break; }
        int after = cls_of(c.actor)[MGX_C_ON_AFTER_USE];
        if constexpr (DEPTH > 0) { if (after >= 0) apply_handler<DEPTH - 1>(after, c); }
        break;
      }
    }
  }

  // Handler::try_apply (handler/handler.cpp:76-93) and MultiHandler::try_apply (multi_handler.cpp:8-21).
  // DEPTH bounds nesting (multi -> leaf -> use_target -> on_use multi -> leaf); exceeded depth raises MGX_ENV_DEPTH.
  template <int DEPTH>
  __device__ bool apply_handler(int h, MgxCtx& c) const {
    PP hd = P + d.sec[MGX_SEC_HANDLERS] + h * MGX_HD_WORDS;
    if (hd[MGX_HD_KIND] == MGX_HK_LEAF) {
      if (!check_filters(hd[MGX_HD_FILTER_PC], c)) return false;
      c.mutation_failed = false;
      PP m = P + d.sec[MGX_SEC_MUTS] + hd[MGX_HD_MUT_START] * MGX_MU_WORDS;
      for (int i = 0; i < hd[MGX_HD_MUT_COUNT]; i++, m += MGX_MU_WORDS) {
        mutate<DEPTH>(m, c);
        if (c.mutation_failed) return false;
      }
      return true;
    }
    bool any = false;
    if constexpr (DEPTH > 0) {
      PP kids = P + d.sec[MGX_SEC_CHILDREN] + hd[MGX_HD_CHILD_START];
      bool first = hd[MGX_HD_KIND] == MGX_HK_FIRST_MATCH;
      for (int i = 0; i < hd[MGX_HD_CHILD_COUNT]; i++) {
        if (apply_handler<DEPTH - 1>(kids[i], c)) {
          any = true;
          if (first) return true;
        }
      }
    } else {
      flag(4u);
    }
    return any;
  }

  // ---- actions ----
  __device__ bool do_move(int slot, int orient) const {  // actions/move.hpp:81-115, orientation.hpp:28-48
    const int dx = (orient == 2 || orient == 4 || orient == 6) ? -1 : (orient == 3 || orient == 5 || orient == 7) ? 1 : 0;
    const int dy = (orient == 0 || orient == 4 || orient == 5) ? -1 : (orient == 1 || orient == 6 || orient == 7) ? 1 : 0;
    PP mh = P + d.sec[MGX_SEC_MOVE_HANDLERS];
    for (int k = 0; k < d.n_move_handlers; k++, mh += MGX_MH_WORDS) {
      uint16_t rc = d.obj_rc[so(slot)];
      for (int i = 1; i <= mh[MGX_MH_MAX_RANGE]; i++) {
        int r = (rc >> 8) + dy * i, c = (rc & 0xFF) + dx * i;
        if (r < 0 || c < 0 || r >= d.H || c >= d.W) break;
        int t = (int)cell(r, c) - 1;
        if (t < 0 && !mh[MGX_MH_ACCEPTS_EMPTY]) continue;
        MgxCtx ctx;
        ctx.actor = slot; ctx.target = t; ctx.target_r = r; ctx.target_c = c; ctx.move_direction = orient;
        ctx.mutation_failed = false;
        if (apply_handler<3>(mh[MGX_MH_HANDLER], ctx)) return true;
        break;
      }
    }
    return false;
  }
  __device__ bool handle_action(int ai, int action) const {  // actions/action_handler.hpp:78-105
    PP ac = P + d.sec[MGX_SEC_ACTIONS] + action * MGX_AC_WORDS;
    int kind = ac[MGX_AC_KIND];
    int slot = d.ag_obj[ao(ai)];
    bool ok = true;
    if (kind == MGX_AK_MOVE) ok = do_move(slot, ac[MGX_AC_ARG]);
    else if (kind == MGX_AK_VIBE) d.obj_vibe[so(slot)] = (uint8_t)ac[MGX_AC_ARG];  // actions/change_vibe.hpp:48-57
    uint16_t rc = d.obj_rc[so(slot)];
    if (rc == d.ag_prev[ao(ai)]) {
      uint32_t swm = ++d.ag_swm[ao(ai)];
      int sid = mgx_wk(d, MGX_S_MAX_STEPS_WITHOUT_MOTION);
      if ((float)swm > astat_get(ai, sid)) astat_set(ai, sid, (float)swm);
    } else {
      d.ag_swm[ao(ai)] = 0;
    }
    d.ag_prev[ao(ai)] = rc;
    int s_ok = kind == MGX_AK_NOOP ? MGX_S_NOOP_SUCCESS : kind == MGX_AK_MOVE ? MGX_S_MOVE_SUCCESS : MGX_S_VIBE_SUCCESS;
    if (ok) astat_add(ai, mgx_wk(d, s_ok), 1.f);
    else { astat_add(ai, mgx_wk(d, s_ok + 1), 1.f); astat_add(ai, mgx_wk(d, MGX_S_ACTION_FAILED), 1.f); }
    return ok;
  }

  __device__ void track_coverage(int ai) const {  // objects/agent.cpp:49-57
    uint16_t rc = d.obj_rc[so(d.ag_obj[ao(ai)])];
    // Unchanged position since the last call => the set is unchanged and both stats.set() calls would store the
    // values they already hold (keys exist since Agent::init): nothing to do.
    if (rc == d.ag_covrc[ao(ai)]) return;
    d.ag_covrc[ao(ai)] = rc;
    uint16_t sp = d.ag_spawn[ao(ai)];
    int r = rc >> 8, c = rc & 0xFF;
    int bit = r * d.W + c;
    uint32_t& w = d.ag_seen[ao(ai) * d.SEENW + (bit >> 5)];
    uint32_t uniq = d.ag_unique[ao(ai)];
    if (!(w & (1u << (bit & 31)))) { w |= 1u << (bit & 31); d.ag_unique[ao(ai)] = ++uniq; }
    astat_set(ai, mgx_wk(d, MGX_S_CELL_UNIQUE), (float)uniq);
    int dist = abs((int)(sp >> 8) - r) + abs(c - (int)(sp & 0xFF));
    uint32_t md = max(d.ag_maxdist[ao(ai)], (uint32_t)dist);
    d.ag_maxdist[ao(ai)] = md;
    astat_set(ai, mgx_wk(d, MGX_S_CELL_MAXDIST), (float)md);
  }

  // ---- std::mt19937 + libstdc++ uniform_int_distribution / shuffle ----
  __device__ uint32_t rng_next() const {  // incremental twist: identical stream to the batch _M_gen_rand
    uint32_t i = d.mt_idx[env];
    uint32_t i1 = i + 1 == 624 ? 0 : i + 1;
    uint32_t im = i + 397 >= 624 ? i + 397 - 624 : i + 397;
    size_t E = (size_t)d.E;
    uint32_t y = (d.mt[i * E + env] & 0x80000000u) | (d.mt[i1 * E + env] & 0x7fffffffu);
    uint32_t x = d.mt[im * E + env] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    d.mt[i * E + env] = x;
    d.mt_idx[env] = i1;
    x ^= x >> 11;
    x ^= (x << 7) & 0x9d2c5680u;
    x ^= (x << 15) & 0xefc60000u;
    x ^= x >> 18;
    return x;
  }
  __device__ uint32_t rng_below(uint32_t range) const {  // bits/uniform_int_dist.h:246-270 (Lemire, 64-bit product)
    unsigned long long p = (unsigned long long)rng_next() * range;
    uint32_t low = (uint32_t)p;
    if (low < range) {
      uint32_t thr = (0u - range) % range;
      while (low < thr) {
        p = (unsigned long long)rng_next() * range;
        low = (uint32_t)p;
      }
    }
    return (uint32_t)(p >> 32);
  }
};

typedef MgxEnvT<MgxGlobalProg> MgxEnv;

// order: LDS, [k][lane] bytes (k-major so that a wavefront access is bank-conflict free)
__device__ __forceinline__ void mgx_swap(uint8_t* order, int lane, int i, int j) {
  uint8_t a = order[i * MGX_WAVE + lane], b = order[j * MGX_WAVE + lane];
  order[i * MGX_WAVE + lane] = b;
  order[j * MGX_WAVE + lane] = a;
}

template <class PP>
__device__ __forceinline__ void mgx_world_body(const MgxDev& d, PP P, uint8_t* order, int lane, int env) {
  MgxEnvT<PP> e(d, P, env);
  const int A = d.A;
  e.step = ++d.step[env];

  const bool want_stepprev = (d.flags & MGX_G_LAST_ACTION_MOVE) != 0;
  for (int i = 0; i < A; i++) {  // mettagrid_c.cpp:929-944 (executed/success are cleared by the host-side memset)
    if (want_stepprev) d.ag_stepprev[e.ao(i)] = d.obj_rc[e.so(d.ag_obj[e.ao(i)])];
    order[i * MGX_WAVE + lane] = (uint8_t)i;
  }
  // std::shuffle (bits/stl_algo.h:3729-3795): one draw for an even n, then paired draws
  if (A >= 2) {
    uint32_t i = 1;
    if ((A & 1) == 0) {
      uint32_t j = e.rng_below(2);
      mgx_swap(order, lane, 1, (int)j);
      i = 2;
    }
    while (i < (uint32_t)A) {
      uint32_t s = i + 1;
      uint32_t x = e.rng_below(s * (s + 1));
      mgx_swap(order, lane, (int)i, (int)(x / (s + 1)));
      mgx_swap(order, lane, (int)i + 1, (int)(x % (s + 1)));
      i += 2;
    }
  }
  // Action dispatch (mettagrid_c.cpp:966-999).  The reference loops over priority levels max..0 and, inside each,
  // over the primary then the vibe stream.  Every real action handler has priority 0 and the only thing the
  // higher (empty) levels do is repeat the invalid-index bookkeeping, so one pass per stream with the invalid-index
  // stats added (max_priority + 1) times is equivalent — the stat keys involved are touched by nothing else.
  PP acts = P + d.sec[MGX_SEC_ACTIONS];
  const int repeats = d.max_priority + 1;
  for (int stream = 0; stream < 2; stream++) {
    const int32_t* src = stream == 0 ? d.actions : d.vibe_actions;
    for (int k = 0; k < A; k++) {
      int ai = order[k * MGX_WAVE + lane];
      int a = src[e.ao(ai)];
      if (a < 0 || a >= d.nact) {  // _handle_invalid_action :914-919
        for (int rep = 0; rep < repeats; rep++) {
          e.astat_add(ai, mgx_wk(d, MGX_S_INVALID_INDEX), 1.f);
          if (a < 0 && a >= -MGX_INVALID_WINDOW) e.astat_add(ai, mgx_wk(d, MGX_S_INVALID_NEG_BASE) + a + MGX_INVALID_WINDOW, 1.f);
          else if (a >= d.nact && a < d.nact + MGX_INVALID_WINDOW) e.astat_add(ai, mgx_wk(d, MGX_S_INVALID_POS_BASE) + a - d.nact, 1.f);
          else e.flag(2u);
        }
        d.success[e.ao(ai)] = 0;
        continue;
      }
      bool is_vibe = acts[a * MGX_AC_WORDS + MGX_AC_KIND] == MGX_AK_VIBE;
      if (is_vibe != (stream == 1)) continue;
      if (e.handle_action(ai, a)) {
        d.executed[e.ao(ai)] = a;
        d.success[e.ao(ai)] = 1;
      }
    }
  }
  if (d.any_on_tick) {
    for (int i = 0; i < A; i++) {  // per-agent on_tick (mettagrid_c.cpp:1019-1024)
      int slot = d.ag_obj[e.ao(i)];
      int h = e.cls_of(slot)[MGX_C_ON_TICK];
      if (h >= 0) {
        MgxCtx c;
        c.actor = c.target = slot; c.target_r = c.target_c = 0; c.move_direction = 0; c.mutation_failed = false;
        e.template apply_handler<3>(h, c);
      }
    }
  }
  for (int i = 0; i < A; i++) e.track_coverage(i);  // mettagrid_c.cpp:1054-1056
}

// PROG_LDS: the program blob is first copied into LDS (16-byte coalesced loads) and every table lookup of the
// handler VM becomes a ds_read.  Dynamic LDS: order u8[A][64] | program i32[prog_words].
template <bool PROG_LDS>
__global__ void __launch_bounds__(MGX_WAVE) mgx_world_kernel(MgxDev d, int prog_words) {
  extern __shared__ __align__(16) uint8_t wsmem[];
  uint8_t* order = wsmem;
  const int lane = threadIdx.x;
  const int env = blockIdx.x * MGX_WAVE + lane;
  if (PROG_LDS) {
    int32_t* lprog = (int32_t*)(wsmem + ((d.A * MGX_WAVE + 15) & ~15));
    const int4* src = (const int4*)d.P;
    int4* dst = (int4*)lprog;
    for (int i = lane; i < prog_words / 4; i += MGX_WAVE) dst[i] = src[i];
    __syncthreads();
    if (env >= d.E) return;
    mgx_world_body<MgxLdsProg>(d, (MgxLdsProg)lprog, order, lane, env);
  } else {
    if (env >= d.E) return;
    mgx_world_body<MgxGlobalProg>(d, d.P, order, lane, env);
  }
}

// Construction: MettaGrid ctor + _init_grid (mettagrid_c.cpp:42-191, 200-269).  One lane per env scans the class
// map in row-major order; object slot = reference object id - 1; agent index = order of appearance.
__global__ void __launch_bounds__(MGX_WAVE) mgx_init_kernel(MgxDev d, const uint16_t* class_maps, const uint32_t* seeds) {
  const int env = blockIdx.x * MGX_WAVE + threadIdx.x;
  if (env >= d.E) return;
  MgxEnv e(d, d.P, env);
  const size_t E = (size_t)d.E;
  uint32_t x = seeds[env];  // std::mt19937(seed): bits/random.tcc seed()
  d.mt[env] = x;
  for (uint32_t i = 1; i < 624; i++) {
    x = 1812433253u * (x ^ (x >> 30)) + i;
    d.mt[i * E + env] = x;
  }
  d.mt_idx[env] = 0;
  d.step[env] = 0;
  d.err[env] = 0;
  e.gstat_touch(mgx_wk(d, MGX_S_GAME_TOKENS_WRITTEN));
  e.gstat_touch(mgx_wk(d, MGX_S_GAME_TOKENS_DROPPED));
  e.gstat_touch(mgx_wk(d, MGX_S_GAME_TOKENS_FREE));
  const int HW = d.H * d.W;
  const uint16_t* cm = class_maps + (size_t)env * HW;
  int nobj = 0, nag = 0;
  for (int cellidx = 0; cellidx < HW; cellidx++) {
    int k = cm[cellidx];
    if (!k) continue;
    if (nobj >= d.S) { e.flag(8u); break; }
    int slot = nobj++;
    int cls = k - 1;
    const int32_t* C = mgx_cls(d, cls);
    int r = cellidx / d.W, c = cellidx % d.W;
    d.grid[(size_t)env * HW + cellidx] = (uint16_t)(slot + 1);
    d.obj_cls[e.so(slot)] = (uint16_t)cls;
    d.obj_rc[e.so(slot)] = (uint16_t)((r << 8) | c);
    d.obj_vibe[e.so(slot)] = (uint8_t)C[MGX_C_INITIAL_VIBE];
    int ai = -1;
    if (C[MGX_C_KIND] == MGX_KIND_AGENT && nag < d.A) {
      ai = nag++;
      d.ag_obj[e.ao(ai)] = (uint16_t)slot;
      uint16_t rc = (uint16_t)((r << 8) | c);
      d.ag_prev[e.ao(ai)] = rc;
      d.ag_spawn[e.ao(ai)] = rc;
      d.ag_stepprev[e.ao(ai)] = rc;
      d.ag_covrc[e.ao(ai)] = 0xFFFF;
    }
    d.obj_agent[e.so(slot)] = ai < 0 ? MGX_NO_AGENT : (uint8_t)ai;
    const int32_t* ii = d.P + d.sec[MGX_SEC_INIT_INV] + C[MGX_C_INIT_INV_START] * MGX_II_WORDS;
    for (int i = 0; i < C[MGX_C_INIT_INV_COUNT]; i++, ii += MGX_II_WORDS) {
      // objects/agent.cpp:79-84 (limits ignored, no callback, "<res>.amount" set), grid_object_factory.cpp:83-87
      e.inv_update<0>(slot, ii[MGX_II_ITEM], ii[MGX_II_AMOUNT], true, false);
      if (ai >= 0) e.astat_set(ai, mgx_wk(d, MGX_S_RES_AMOUNT_BASE) + ii[MGX_II_ITEM], (float)ii[MGX_II_AMOUNT]);
    }
    e.gstat_add(C[MGX_C_OBJECTS_STAT], 1.f);
  }
  d.num_objs[env] = (uint32_t)nobj;
  for (int ai = 0; ai < nag; ai++) {
    e.track_coverage(ai);  // Agent::init -> reset_coverage_tracking (agent.cpp:25-28,41-47)
    const int32_t* C = e.cls_of(d.ag_obj[e.ao(ai)]);
    const int32_t* rw = d.P + d.sec[MGX_SEC_REWARDS] + C[MGX_C_REWARD_START] * MGX_RW_WORDS;
    for (int i = 0; i < C[MGX_C_REWARD_COUNT]; i++, rw += MGX_RW_WORDS) {  // systems/reward.hpp:45-53
      if (rw[MGX_RW_TOUCH_SCOPE] == 0) e.astat_touch(ai, rw[MGX_RW_TOUCH_STAT]);
      else if (rw[MGX_RW_TOUCH_SCOPE] == 1) e.gstat_touch(rw[MGX_RW_TOUCH_STAT]);
    }
  }
}

#endif  // MGX_WORLD_H_
