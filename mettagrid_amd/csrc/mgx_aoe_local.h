// mgx_aoe_local.h — the area-effect phase with one lane per AGENT (mgx_aoe_kernel): per-agent on_tick handlers, fixed AoEs,
// territory effects, mobile AoEs and coverage tracking of MettaGrid::_step (/root/reference/cpp/bindings/mettagrid_c.cpp
// :1019-1056) for games whose records in that phase only touch their TARGET (the host proves it: mgx_aoe_is_target_local).
//
// Why its own interpreter.  When every record of the phase reads and writes nothing but the agent it is applied to, the
// agents of an env are independent, and the only mutable object a lane ever touches is ITS agent: inventory row, inventory
// order, vibe, and a handful of that agent's stat cells.  Round 2 ran the generic interpreter here: every Inventory::update
// re-loaded class, row, order and agent index, every stat update was its own read-modify-write, every source scan read the
// per-source records and bitmaps from HBM — about 150 dependent round trips per agent and 1.3 GB of partial-line writes per
// launch for a few changed cells.  Here the lane loads its agent ONCE (three levels of independent loads: agent-indexed
// state, then slot-indexed state, then what depends on the position), interprets every record on that copy, and writes
// back what changed at the end:
//   * inventory row (8 registers) + order word, vibe, tag words: registers;
//   * the agent stats the phase can touch — the host lists them from the program (mgx_aoe_collect_stats): gained / lost /
//     amount of every resource a local mutation names, death, the targets of SetStat / game-value mutations, the stats local
//     value code reads, the two coverage stats — one LDS cell per lane and stat, with dirty / touched bit sets in registers;
//     any other stat id falls back to HBM (same semantics, slower);
//   * the source records of the env (packed location / radius / live, object, AoE index) are staged per wavefront in LDS
//     by its lanes (64 sources per pass), so the all-pairs range test agent x source is LDS + VALU only — this replaces the
//     per-source HBM scan for every radius, without a grid-neighbourhood special case;
//   * "agent is inside source" bits are stored agent-major ([E][A][words]): the lane owns its words, no atomics.
// The order of effects on an agent is the reference's: on_tick, exits of fixed AoEs, fixed AoEs in registration order with
// the deferred net deltas, territories, mobile sources in registration order, coverage.  f32 stat sums see the same addends
// in the same order as the serial form, so results are bit-identical.
#ifndef MGX_AOE_LOCAL_H_
#define MGX_AOE_LOCAL_H_

#include "mgx_world.h"

#define MGX_AOE_THREADS 256
#define MGX_AOE_WAVES (MGX_AOE_THREADS / MGX_WAVE)
#define MGX_AOE_DEF_ROWS 13  // deferred target deltas of apply_fixed, one per resource
// dynamic LDS: stat map u8[256] | staged stat ids i16[32] | per-wave source staging u32[MGX_AOE_WAVES][4][64] | per-thread
// stat cells f32[nstat][threads] | per-thread deferred deltas i32[13][threads] | (optional) hot program range
#define MGX_AOE_HDR (256 + MGX_AOE_MAX_STATS * 2)
__host__ __device__ inline int mgx_aoe_lds_bytes(int nstat) {
  return MGX_AOE_HDR + MGX_AOE_WAVES * 4 * MGX_WAVE * 4 + (nstat + MGX_AOE_DEF_ROWS) * MGX_AOE_THREADS * 4;
}

namespace MGX_TU_NS {

// Packed source record written by mgx_aoe_prep_kernel every step: rc (16) | radius (8) << 16 | live << 24 — live = the
// source is registered and has an effect (territory-style AoEs without mutations and presence deltas never do).
__device__ __forceinline__ bool mgx_pack_covers(uint32_t p, int r, int c) {
  const int dr = r - (int)((p >> 8) & 0xFF), dc = c - (int)(p & 0xFF), rad = (int)((p >> 16) & 0xFF);
  return ((p >> 24) & 1u) && dr * dr + dc * dc <= rad * rad;
}
// One packed record per registered AoE source (mgx_aoe_prep_kernel, one thread per source).
template <class Env>
__device__ __forceinline__ void mgx_aoe_pack_source(const Env& e, int k) {
  const MgxDev& d = e.d;
  const bool fixed = k < d.NF;
  const int i = fixed ? k : k - d.NF;
  const size_t q = (size_t)e.envi() * (fixed ? d.NF : d.NM) + i;
  const int n = fixed ? (d.NF ? d.fx_count[e.envi()] : 0) : (d.NM ? d.mb_count[e.envi()] : 0);
  uint32_t p = 0;
  if (i < n) {
    const int obj = fixed ? d.fx_obj[q] : d.mb_obj[q];
    if (obj != 0xFFFF) {
      auto a = e.aoe(fixed ? d.fx_aoe[q] : d.mb_aoe[q]);
      const uint16_t rc = fixed ? d.fx_rc[q] : d.obj_rc[e.so(obj)];  // fixed: location at registration (:166-200)
      const bool effect = !fixed || a[MGX_AO_MUT_COUNT] > 0 || a[MGX_AO_PRES_COUNT] > 0;
      p = (uint32_t)rc | ((uint32_t)(a[MGX_AO_RADIUS] & 0xFF) << 16) | (effect ? 1u << 24 : 0u);
    }
  }
  if (fixed) d.fx_pack[q] = p; else d.mb_pack[q] = p;
}

// The lane's agent, resident in registers / LDS for the whole phase.
template <class Env>
struct MgxLocalAgent {
  typedef typename Env::InvRow InvRow;
  typedef typename Env::ProgPtr PP;   // program records: the kernel's LDS copy of the hot sections, or the blob in HBM / L2
  typedef typename Env::CP CP;
  const Env& e;
  const MgxDev& d;
  int ai, slot, r, c;
  CP C;  // class record of the agent's object
  InvRow row;
  unsigned long long ord;
  uint32_t tags[MGX_TAG_WORDS];
  uint32_t vibe;
  bool inv_dirty, vibe_dirty;
  // stat cells: LDS [k][thread]; which are dirty / touched: registers
  float* sval;
  const uint8_t* smap;
  const int16_t* sids;
  uint32_t sdirty, stouch;
  // deferred deltas of apply_fixed (core/aoe_tracker.cpp:283-289, 347-361): LDS [resource][thread]; seen mask, first-seen order
  int* dd;
  uint32_t def_seen;
  unsigned long long def_order;
  int def_cnt;

  __device__ __forceinline__ MgxLocalAgent(const Env& env) : e(env), d(env.d), ai(0), slot(0), r(0), c(0) {}

  // ---- load: everything the phase may read of this agent, in three rounds of independent loads ----
  __device__ __forceinline__ void load(int agent, float* stat_cells, const uint8_t* stat_map, const int16_t* stat_ids, int* deferred) {
    ai = agent;
    sval = stat_cells; smap = stat_map; sids = stat_ids; dd = deferred;
    sdirty = stouch = 0;
    inv_dirty = vibe_dirty = false;
    def_seen = 0; def_order = 0; def_cnt = 0;
    const size_t ao = e.ao(ai);
    slot = d.ag_obj[ao];
    for (int k = 0; k < d.aoe_nstat; k++) sval[k * MGX_AOE_THREADS] = d.ag_stats[ao * d.NSP + sids[k]];
    const size_t so = e.so(slot);
    const uint16_t rc = d.obj_rc[so];
    const uint16_t cls = d.obj_cls[so];
    row = e.inv_row(slot);
    ord = d.obj_order[so];
    vibe = d.obj_vibe[so];
    C = e.cls(cls);
#pragma unroll
    for (int w = 0; w < MGX_TAG_WORDS; w++) tags[w] = d.obj_tags ? d.obj_tags[so * MGX_TAG_WORDS + w] : (uint32_t)C[MGX_C_TAGS + w];
    r = rc >> 8; c = rc & 0xFF;
  }

  // ---- stats (systems/stats_tracker.hpp:69-90) on the staged cells ----
  __device__ __forceinline__ int sidx(int id) const { return (id >= 0 && id < 256) ? (int)smap[id] : 0xFF; }
  __device__ __forceinline__ float sget(int id, bool touch) {
    if (id < 0) return 0.f;
    const int k = sidx(id);
    if (k != 0xFF) { if (touch) stouch |= 1u << k; return sval[k * MGX_AOE_THREADS]; }
    if (touch) e.astat_touch(ai, id);
    return e.astat_get(ai, id);
  }
  __device__ __forceinline__ void sset(int id, float v, bool touch) {
    if (id < 0) return;
    const int k = sidx(id);
    if (k != 0xFF) { sval[k * MGX_AOE_THREADS] = v; sdirty |= 1u << k; if (touch) stouch |= 1u << k; return; }
    d.ag_stats[e.ao(ai) * d.NSP + id] = v;
    if (touch) e.astat_touch(ai, id);
  }
  __device__ __forceinline__ void sadd(int id, float dv, bool touch) { if (id >= 0) sset(id, __fadd_rn(sget(id, false), dv), touch); }

  // ---- inventory (cpp/src/mettagrid/objects/inventory.cpp) on the register copy ----
  // Agent::on_inventory_change (objects/agent.cpp:106-121)
  __device__ __forceinline__ void on_inventory_change(int item, int delta, int amount) {
    if (delta == 0) return;
    sadd((delta > 0 ? d.wk[MGX_S_RES_GAINED_BASE] : d.wk[MGX_S_RES_LOST_BASE]) + item, (float)(delta > 0 ? delta : -delta), false);
    sset(d.wk[MGX_S_RES_AMOUNT_BASE] + item, (float)amount, true);
    if (amount == 0 && delta < 0 && item == d.hp_res) sadd(d.wk[MGX_S_DEATH], 1.f, false);
  }
  // Inventory::update (inventory.cpp:38-86); DEPTH bounds the update -> enforce_all_limits -> update recursion
  template <int DEPTH>
  __device__ __forceinline__ int inv_update(int item, int delta) {
    const int initial = row.get(item);
    int mx = 65535;
    PP L = e.limit_of(C, item);
    if (L) {
      int used = e.group_amount(row, L) - initial;
      if (used < 0) used = 0;
      const int m = e.effective_limit(row, L) - used;
      mx = m < 0 ? 0 : m;
    }
    const int clamped = min(max(initial + delta, 0), mx);
    if (clamped != initial) {
      if (initial == 0) {  // new node goes to the list head (libstdc++ _M_insert_bucket_begin; mettagrid_amd/umap.py)
        ord = (ord << 4) | (unsigned long long)item;
      } else if (clamped == 0) {  // erase keeps the order of the rest
        int p = 0;
        while (p < 16 && ((ord >> (4 * p)) & 0xF) != (unsigned long long)item) p++;
        const unsigned long long low = p ? (ord & ((1ull << (4 * p)) - 1ull)) : 0ull;
        const unsigned long long high = p >= 15 ? 0ull : (ord >> (4 * (p + 1)));
        ord = low | (high << (4 * p)) | (0xFull << 60);
      }
      row.set(item, clamped);
      inv_dirty = true;
    }
    const int dl = clamped - initial;
    if (dl != 0) on_inventory_change(item, dl, clamped);
    if (dl < 0 && (C[MGX_C_MODIFIER_MASK] & (1 << item))) {
      if constexpr (DEPTH > 0) enforce_all_limits<DEPTH - 1>();
      else e.flag(4u);
    }
    return dl;
  }
  template <int DEPTH>
  __device__ __forceinline__ void enforce_all_limits() {  // inventory.cpp:141-173
    for (int li = 0; li < C[MGX_C_LIMIT_COUNT]; li++) {
      PP L = e.prog() + d.sec[MGX_SEC_LIMITS] + (C[MGX_C_LIMIT_START] + li) * MGX_L_WORDS;
      if (L[MGX_L_DROP_COUNT] == 0) continue;
      int excess = e.group_amount(row, L) - e.effective_limit(row, L);
      if (excess <= 0) continue;
      PP drop = e.prog() + d.sec[MGX_SEC_DROP_ORDER] + L[MGX_L_DROP_START];
      for (int k = 0; k < L[MGX_L_DROP_COUNT]; k++) {
        const int item = drop[k];
        const int to_drop = min(row.get(item), excess);
        if (to_drop > 0) {
          inv_update<DEPTH>(item, -to_drop);
          excess = e.group_amount(row, L) - e.effective_limit(row, L);
        }
        if (excess <= 0) break;
      }
    }
  }
  __device__ __forceinline__ void presence(PP a, int mult) {  // apply_presence_deltas (core/aoe_tracker.cpp:122-126)
    PP pr = e.prog() + d.sec[MGX_SEC_PRESENCE] + a[MGX_AO_PRES_START] * MGX_PR_WORDS;
    for (int i = 0; i < a[MGX_AO_PRES_COUNT]; i++, pr += MGX_PR_WORDS) inv_update<1>(pr[MGX_PR_RESOURCE], pr[MGX_PR_DELTA] * mult);
  }

  // ---- what filters and value code read ----
  __device__ __forceinline__ uint32_t tagword(int s, int w, const MgxCtx& c) const {
    if (s == slot) {  // (run-time w over the eight registers)
      uint32_t v = 0;
#pragma unroll
      for (int q = 0; q < MGX_TAG_WORDS; q++) v = (w == q) ? tags[q] : v;
      return v;
    }
    return e.tagword(s, w, c);
  }
  __device__ __forceinline__ int inv_of(int s, int item) const { return s == slot ? row.get(item) : e.inv_of(s, item); }
  __device__ __forceinline__ uint16_t rc_of(int s) const { return s == slot ? (uint16_t)((r << 8) | c) : d.obj_rc[e.so(s)]; }
  __device__ __forceinline__ int resolve(const MgxCtx& c, int ent) const { return ent == MGX_ENT_ACTOR ? c.actor : c.target; }

  // game values (core/game_value.cpp:14-148) restricted to what mgx_aoe_is_target_local admits
  __device__ __forceinline__ float eval_code(int start, int count, int entity) {
    MgxValueStack st;
    PP code = e.prog() + d.sec[MGX_SEC_GV_CODE] + start * MGX_GV_WORDS;
    for (int i = 0; i < count; i++, code += MGX_GV_WORDS) {
      const int a0 = code[MGX_GV_A0], a1 = code[MGX_GV_A1], a2 = code[MGX_GV_A2];
      switch (code[MGX_GV_OP]) {
        case MGX_GOP_INVENTORY: st.push(entity >= 0 ? (float)inv_of(entity, a0) : 0.f); break;
        case MGX_GOP_STAT: {  // agent scope (the host rejects game scope here); reading creates the key
          float v = 0.f;
          if (entity == slot) v = sget(a1, true);
          else { const int a = e.agent_of(entity); if (a >= 0) { e.astat_touch(a, a1); v = e.astat_get(a, a1); } }
          st.push(v);
          break;
        }
        case MGX_GOP_CONST: st.push(__int_as_float(a0)); break;
        case MGX_GOP_ADD_TERM: {
          float t = st.pop();
          if (a0) t = mgx_logf(__fadd_rn(t, 1.0f));
          if (a1) t = __fmul_rn(t, __int_as_float(a2));
          const float acc = st.pop();
          st.push(__fadd_rn(acc, t));
          break;
        }
        case MGX_GOP_RATIO: { const float den = st.pop(), num = st.pop(); st.push(den > 0.f ? __fdiv_rn(num, den) : num); break; }
        case MGX_GOP_MAX2: { const float v = st.pop(), b = st.pop(); st.push((b < v) ? v : b); break; }
        case MGX_GOP_MIN2: { const float v = st.pop(), b = st.pop(); st.push((v < b) ? v : b); break; }
        default: e.flag(4u); st.push(0.f); break;
      }
    }
    return st.n > 0 ? st.s0 : 0.f;
  }
  __device__ __forceinline__ float eval_value(int rec, int entity) {
    PP V = e.prog() + d.sec[MGX_SEC_OBS_VALUES] + rec * MGX_OV_WORDS;
    return eval_code(V[MGX_OV_GV_START], V[MGX_OV_GV_COUNT], entity);
  }

  // filters (handler/filters/*.hpp)
  __device__ __forceinline__ bool atom(PP a, const MgxCtx& c) {
    const int a0 = a[MGX_AT_A0], a1 = a[MGX_AT_A1], a2 = a[MGX_AT_A2];
    switch (a[MGX_AT_OP]) {
      case MGX_FOP_VIBE: {
        const int s = resolve(c, a0);
        return s != MGX_SLOT_NONE && (s >= 0 ? (s == slot ? (int)vibe : (int)d.obj_vibe[e.so(s)]) : 0) == a1;
      }
      case MGX_FOP_RESOURCE: { const int s = resolve(c, a0); return s != MGX_SLOT_NONE && inv_of(s, a1) >= a2; }
      case MGX_FOP_SHARED_TAG: {
        PP mask = e.prog() + d.sec[MGX_SEC_WORDLIST] + a0;
        uint32_t any = 0;
        for (int w = 0; w < MGX_TAG_WORDS; w++) {
          const uint32_t mine = tagword(c.target, w, c) & (uint32_t)mask[w];  // the target is this agent: registers
          if (mine) any |= mine & tagword(c.actor, w, c);
        }
        return any != 0;
      }
      case MGX_FOP_TAG: {
        const int s = resolve(c, a0);
        if (s == MGX_SLOT_NONE) return false;
        PP mask = e.prog() + d.sec[MGX_SEC_WORDLIST] + a1;
        uint32_t any = 0;
        for (int w = 0; w < MGX_TAG_WORDS; w++)
          if (mask[w]) any |= tagword(s, w, c) & (uint32_t)mask[w];
        return any != 0;
      }
      case MGX_FOP_TARGET_LOC_EMPTY: return c.target == MGX_SLOT_NONE;
      case MGX_FOP_TARGET_IS_USABLE: return c.target != MGX_SLOT_NONE;
      case MGX_FOP_PERIODIC: return e.step >= (uint32_t)a1 && ((e.step - (uint32_t)a1) % (uint32_t)a0) == 0;
      case MGX_FOP_GAME_VALUE: {
        const int s = resolve(c, a0);
        const float v = eval_value(a1, s);
        const float t = eval_value(a2, s);
        return v >= t;
      }
      case MGX_FOP_MAX_DISTANCE: {  // the binary form (filters/max_distance_filter.hpp:27-43)
        const int s = resolve(c, a0);
        if (s < 0 || a2 >= 0) return false;
        const int ref = c.source >= 0 ? c.source : c.actor;
        if (ref < 0) return false;
        if (a1 == 0) return true;
        const uint16_t erc = rc_of(s), rrc = rc_of(ref);
        const long long rr = a1, dr = (int)(erc >> 8) - (int)(rrc >> 8), dc = (int)(erc & 0xFF) - (int)(rrc & 0xFF);
        return dr * dr + dc * dc <= rr * rr;
      }
      case MGX_FOP_TRUE: return true;
      default: return false;
    }
  }
  __device__ __forceinline__ bool check_filters(int pc, const MgxCtx& c) {  // handler/handler.cpp:95-103
    PP atoms = e.prog() + d.sec[MGX_SEC_ATOMS];
    while (pc >= 0) {
      PP a = atoms + pc * MGX_AT_WORDS;
      pc = atom(a, c) ? a[MGX_AT_ON_TRUE] : a[MGX_AT_ON_FALSE];
    }
    return pc == MGX_PC_PASS;
  }

  // mutations (handler/mutations/*.hpp) whose entity is this agent
  __device__ __forceinline__ void mutate(PP m, const MgxCtx& c) {
    const int a0 = m[MGX_MU_A0], a1 = m[MGX_MU_A1], a2 = m[MGX_MU_A2], a3 = m[MGX_MU_A3];
    switch (m[MGX_MU_OP]) {
      case MGX_MOP_RESOURCE_DELTA: {  // resource_mutation.hpp:25-47
        if (resolve(c, a0) != slot) { e.flag(4u); break; }
        if (c.deferred && a0 == MGX_ENT_TARGET && !(C[MGX_C_MODIFIER_MASK] & (1 << a1))) {
          if (!((def_seen >> a1) & 1u)) {
            def_seen |= 1u << a1;
            def_order |= (unsigned long long)a1 << (4 * def_cnt);
            def_cnt++;
            dd[a1 * MGX_AOE_THREADS] = 0;
          }
          dd[a1 * MGX_AOE_THREADS] += a2;
        } else {
          inv_update<1>(a1, a2);
        }
        break;
      }
      case MGX_MOP_CLEAR_INVENTORY: {  // resource_mutation.hpp:111-128
        if (resolve(c, a0) != slot) { e.flag(4u); break; }
        if (a2 == 0) {
          const unsigned long long copy = ord;  // iterate a copy, as the reference does
          for (int k = 0; k < 16; k++) {
            const int item = (int)((copy >> (4 * k)) & 0xF);
            if (item == 0xF) break;
            inv_update<1>(item, -row.get(item));
          }
        } else {
          PP ids = e.prog() + d.sec[MGX_SEC_WORDLIST] + a1;
          for (int i = 0; i < a2; i++) inv_update<1>(ids[i], -row.get(ids[i]));
        }
        break;
      }
      case MGX_MOP_CHANGE_VIBE:
        if (resolve(c, a0) != slot) { e.flag(4u); break; }
        vibe = (uint32_t)a1 & 0xFF; vibe_dirty = true;
        break;
      case MGX_MOP_STATS: {  // stats_mutation.hpp:21-41, agent scope
        const int s = resolve(c, a1);
        const float v = eval_value(a3, s);
        if (s == slot) sset(a2, v, true);
        else e.flag(4u);
        break;
      }
      case MGX_MOP_GAME_VALUE: {  // game_value_mutation.hpp:21-27
        const int s = resolve(c, a0);
        if (s != slot) { e.flag(4u); break; }
        const float delta = eval_value(a2, s);
        PP V = e.prog() + d.sec[MGX_SEC_OBS_VALUES] + a1 * MGX_OV_WORDS;
        PP code = e.prog() + d.sec[MGX_SEC_GV_CODE] + V[MGX_OV_GV_START] * MGX_GV_WORDS;
        if (code[MGX_GV_OP] == MGX_GOP_INVENTORY) inv_update<1>(code[MGX_GV_A0], (int)delta);
        else if (code[MGX_GV_OP] == MGX_GOP_STAT) sadd(code[MGX_GV_A1], delta, true);
        break;
      }
      default: e.flag(4u); break;  // not reachable: the host only selects this kernel for the operations above
    }
  }
  // filters, then EVERY mutation (no stop on mutation_failed: AoE sources core/aoe_tracker.cpp:99-113, territory handlers
  // core/territory_tracker.cpp:62-66, leaf on_tick handlers)
  __device__ __forceinline__ bool apply_all(int filter_pc, int mut_start, int mut_count, const MgxCtx& c) {
    if (!check_filters(filter_pc, c)) return false;
    PP m = e.prog() + d.sec[MGX_SEC_MUTS] + mut_start * MGX_MU_WORDS;
    for (int i = 0; i < mut_count; i++, m += MGX_MU_WORDS) mutate(m, c);
    return true;
  }

  // ---- the phase, piece by piece (the kernel stages the source records between the pieces) ----
  __device__ __forceinline__ void on_tick() {  // mettagrid_c.cpp:1019-1024
    const int h = C[MGX_C_ON_TICK];
    if (h < 0) return;
    PP hd = e.prog() + d.sec[MGX_SEC_HANDLERS] + h * MGX_HD_WORDS;
    MgxCtx tc = mgx_ctx(slot, slot);
    apply_all(hd[MGX_HD_FILTER_PC], hd[MGX_HD_MUT_START], hd[MGX_HD_MUT_COUNT], tc);
  }
  // AOETracker::apply_fixed (core/aoe_tracker.cpp:278-362), sources [f0, f0 + n) staged in s_pack / s_info (obj | aoe << 16).
  // pass 0: every exit of this agent first; pass 1: the covering sources in registration order.
  // (`valid`: this lane holds an agent; every lane of the wavefront runs the scan loop — the records come out of lane
  // registers with v_readlane — and only valid lanes do anything with the result.  `mine`: lane q's packed record.)
  __device__ __forceinline__ void fixed_chunk(bool valid, int pass, int f0, int n, uint32_t mine, const uint32_t* s_pack, const uint32_t* s_info) {
    uint32_t* words = &d.fx_inside[e.ao(valid ? ai : 0) * d.FW + (f0 >> 5)];
    const int nw = (n + 31) >> 5;  // 1 or 2 words of this agent's bitmap
    uint32_t in0 = valid ? words[0] : 0u, in1 = (valid && nw > 1) ? words[1] : 0u;
    const uint32_t old0 = in0, old1 = in1;
    unsigned long long todo = 0;
    for (int q = 0; q < n; q++) {
#ifdef MGX_CPU_EMU
      (void)mine;
      const uint32_t p = s_pack[q];
#else
      (void)s_pack;
      const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)mine, q);
#endif
      const bool covers = mgx_pack_covers(p, r, c);
      const bool was = (((q < 32 ? in0 : in1) >> (q & 31)) & 1u) != 0;
      const bool pick = pass == 0 ? (((p >> 24) & 1u) && was && !covers) : covers;
      if (pick) todo |= 1ull << q;
    }
    if (!valid) return;
    while (todo) {
      const int q = __ffsll((unsigned long long)todo) - 1;
      todo &= todo - 1;
      const uint32_t info = s_info[q];
      const int src = (int)(info & 0xFFFF);
      PP a = e.aoe((int)(info >> 16));
      uint32_t& w = q < 32 ? in0 : in1;
      const uint32_t bit = 1u << (q & 31);
      if (pass == 0) {
        w &= ~bit;
        presence(a, -1);
        continue;
      }
      const bool skip_self = !a[MGX_AO_EFFECT_SELF] && src == slot;
      MgxCtx fc = mgx_ctx(src, slot);
      fc.deferred = true;
      const bool passes = !skip_self && check_filters(a[MGX_AO_FILTER_PC], fc);
      const bool was = (w & bit) != 0;
      if (passes && !was) { w |= bit; presence(a, +1); }
      else if (!passes && was) { w &= ~bit; presence(a, -1); }
      if (passes && a[MGX_AO_MUT_COUNT] > 0) apply_all(a[MGX_AO_FILTER_PC], a[MGX_AO_MUT_START], a[MGX_AO_MUT_COUNT], fc);
    }
    if (in0 != old0) words[0] = in0;
    if (nw > 1 && in1 != old1) words[1] = in1;
  }
  __device__ __forceinline__ void fixed_finish() {  // net delta per resource, first-seen order (:347-361)
    for (int k = 0; k < def_cnt; k++) {
      const int res = (int)((def_order >> (4 * k)) & 0xF);
      const int dl = dd[res * MGX_AOE_THREADS];
      if (dl != 0) inv_update<1>(res, dl);
    }
  }
  // TerritoryTracker::apply_effects (core/territory_tracker.cpp:275-346); ownership from the current map
  __device__ __forceinline__ void territory() {
    for (int ti = 0; ti < d.NT; ti++) {
      PP TE = e.prog() + d.sec[MGX_SEC_TERRITORIES] + ti * MGX_TE_WORDS;
      const uint16_t ov = d.terr_owner[((size_t)e.envi() * d.NT + ti) * (size_t)(d.H * d.W) + r * d.W + c];
      const int cur = ov == 0xFFFF ? -1 : (int)ov;
      int16_t& pv = d.terr_prev[e.ao(ai) * d.NT + ti];
      const int prev = pv;
      if (prev != cur) pv = (int16_t)cur;
      for (int pass = 0; pass < 3; pass++) {
        const int tag = pass == 0 ? prev : cur;
        const bool run = pass == 0 ? (prev != cur && prev >= 0) : pass == 1 ? (prev != cur && cur >= 0) : cur >= 0;
        if (!run) continue;
        const int start = TE[pass == 0 ? MGX_TE_EXIT_START : pass == 1 ? MGX_TE_ENTER_START : MGX_TE_PRES_START];
        const int count = TE[pass == 0 ? MGX_TE_EXIT_COUNT : pass == 1 ? MGX_TE_ENTER_COUNT : MGX_TE_PRES_COUNT];
        for (int i = 0; i < count; i++) {
          PP hd = e.prog() + d.sec[MGX_SEC_HANDLERS] + (start + i) * MGX_HD_WORDS;
          MgxCtx tc = mgx_ctx(MGX_SLOT_PROXY, slot);
          tc.proxy_tag = tag;
          tc.target_r = r; tc.target_c = c;
          apply_all(hd[MGX_HD_FILTER_PC], hd[MGX_HD_MUT_START], hd[MGX_HD_MUT_COUNT], tc);
        }
      }
    }
  }
  // AOETracker::apply_mobile (core/aoe_tracker.cpp:364-415): this agent against the mobile sources [m0, m0 + n), in
  // registration order; only sources in range (or left since the last tick) take the full path.
  __device__ __forceinline__ void mobile_chunk(bool valid, int m0, int n, uint32_t mine, const uint32_t* s_pack, const uint32_t* s_info) {
    uint32_t* words = &d.mb_inside[e.ao(valid ? ai : 0) * d.MW + (m0 >> 5)];
    const int nw = (n + 31) >> 5;
    uint32_t in0 = valid ? words[0] : 0u, in1 = (valid && nw > 1) ? words[1] : 0u;
    const uint32_t old0 = in0, old1 = in1;
    unsigned long long todo = 0, inr = 0;
    for (int q = 0; q < n; q++) {
#ifdef MGX_CPU_EMU
      (void)mine;
      const uint32_t p = s_pack[q];
#else
      (void)s_pack;
      const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)mine, q);
#endif
      const bool in_range = mgx_pack_covers(p, r, c);
      const bool was = (((q < 32 ? in0 : in1) >> (q & 31)) & 1u) != 0;
      if (((p >> 24) & 1u) && (in_range || was)) todo |= 1ull << q;
      if (in_range) inr |= 1ull << q;
    }
    if (!valid) return;
    while (todo) {
      const int q = __ffsll((unsigned long long)todo) - 1;
      todo &= todo - 1;
      const bool in_range = (inr >> q) & 1ull;
      const uint32_t info = s_info[q];
      const int src = (int)(info & 0xFFFF);
      PP a = e.aoe((int)(info >> 16));
      if (!a[MGX_AO_EFFECT_SELF] && src == slot) continue;
      uint32_t& w = q < 32 ? in0 : in1;
      const uint32_t bit = 1u << (q & 31);
      const bool was = (w & bit) != 0;
      if (!in_range) { if (was) { w &= ~bit; presence(a, -1); } continue; }
      MgxCtx mc = mgx_ctx(src, slot);
      if (check_filters(a[MGX_AO_FILTER_PC], mc)) {
        if (!was) { w |= bit; presence(a, +1); }
        if (a[MGX_AO_MUT_COUNT] > 0) apply_all(a[MGX_AO_FILTER_PC], a[MGX_AO_MUT_START], a[MGX_AO_MUT_COUNT], mc);
      } else if (was) {
        w &= ~bit;
        presence(a, -1);
      }
    }
    if (in0 != old0) words[0] = in0;
    if (nw > 1 && in1 != old1) words[1] = in1;
  }
  // Agent::track_coverage (objects/agent.cpp:49-57); nothing moves after this phase (no game on_tick)
  __device__ __forceinline__ void coverage() {
    const size_t ao = e.ao(ai);
    const uint16_t rc = (uint16_t)((r << 8) | c);
    if (rc == d.ag_covrc[ao]) return;  // unchanged position: both stats already hold their values
    d.ag_covrc[ao] = rc;
    const uint16_t sp = d.ag_spawn[ao];
    const int bit = r * d.W + c;
    uint32_t& w = d.ag_seen[ao * d.SEENW + (bit >> 5)];
    uint32_t uniq = d.ag_unique[ao];
    const uint32_t md0 = d.ag_maxdist[ao];
    if (!(w & (1u << (bit & 31)))) { w |= 1u << (bit & 31); d.ag_unique[ao] = ++uniq; }
    sset(d.wk[MGX_S_CELL_UNIQUE], (float)uniq, true);
    const int dist = abs((int)(sp >> 8) - r) + abs(c - (int)(sp & 0xFF));
    const uint32_t md = max(md0, (uint32_t)dist);
    d.ag_maxdist[ao] = md;
    sset(d.wk[MGX_S_CELL_MAXDIST], (float)md, true);
  }
  // ---- write back what changed ----
  __device__ __forceinline__ void store() {
    const size_t so = e.so(slot), ao = e.ao(ai);
    if (inv_dirty) {
      uint4* p = (uint4*)(d.obj_inv + so * MGX_INV_PITCH);
      p[0] = make_uint4(row.w[0], row.w[1], row.w[2], row.w[3]);
      p[1] = make_uint4(row.w[4], row.w[5], row.w[6], row.w[7]);
      d.obj_order[so] = ord;
    }
    if (vibe_dirty) d.obj_vibe[so] = (uint8_t)vibe;
    // "key exists" bits of the touched cells: collected per word, then one load + store per word that gains a bit
    uint32_t tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < d.aoe_nstat; k++) {
      const int id = sids[k];
      if ((sdirty >> k) & 1u) d.ag_stats[ao * d.NSP + id] = sval[k * MGX_AOE_THREADS];
      if ((stouch >> k) & 1u) {
#pragma unroll
        for (int q = 0; q < 8; q++) tm[q] |= ((id >> 5) == q) ? (1u << (id & 31)) : 0u;
      }
    }
    uint32_t cur[8];
#pragma unroll
    for (int q = 0; q < 8; q++) cur[q] = (q < d.NSW && tm[q]) ? d.ag_touched[ao * d.NSW + q] : 0u;
#pragma unroll
    for (int q = 0; q < 8; q++)
      if (q < d.NSW && (tm[q] & ~cur[q])) d.ag_touched[ao * d.NSW + q] = cur[q] | tm[q];
  }
};

}  // namespace MGX_TU_NS

#endif  // MGX_AOE_LOCAL_H_
