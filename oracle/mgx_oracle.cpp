// TEST INFRASTRUCTURE ONLY — scalar CPU restatement of the reference MettaGrid step (one env, array-of-structs,
// no GPU).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
// product path (mettagrid_amd/csrc/mgx_engine.hip) never links or calls it.
//
// PARITY PINNED: this restatement is checked step-by-step against the real reference engine built from
// /root/reference/cpp into oracle/_ref/ (tests/test_oracle_vs_reference.py) and against golden traces generated
// from that build (tests/golden/, generator tests/golden/make_golden.py).
//
// Input is the flat program of include/mgx_program.h (compiled by mettagrid_amd/compiler.py), a class map
// (uint16 [H][W], class index + 1, 0 = empty) and a seed.  Every function cites the reference code it follows
// (paths relative to /root/reference).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <random>
#include <vector>

#include "mgx_program.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// RNG: std::mt19937 + libstdc++ std::shuffle, restated (cpp/bindings/mettagrid_c.cpp:52,958-960;
// /usr/include/c++/11/bits/random.tcc mersenne_twister_engine; bits/uniform_int_dist.h:246-317 (Lemire, 64-bit
// product because mt19937's result_type is uint_fast32_t = 64 bit); bits/stl_algo.h:3729-3795 (paired draws)).
// ---------------------------------------------------------------------------------------------------------------
struct MT {
  uint32_t x[624];
  int idx;
  void seed(uint32_t s) {
    x[0] = s;
    for (int i = 1; i < 624; i++) x[i] = 1812433253u * (x[i - 1] ^ (x[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
  }
  uint32_t next() {
    if (idx >= 624) {
      for (int i = 0; i < 624; i++) {
        uint32_t y = (x[i] & 0x80000000u) | (x[(i + 1) % 624] & 0x7fffffffu);
        x[i] = x[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      idx = 0;
    }
    uint32_t y = x[idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  // uniform_int_distribution<size_t>{0, range-1}, range <= 2^32 - 1
  uint32_t below(uint32_t range) {
    uint64_t p = (uint64_t)next() * (uint64_t)range;
    uint32_t low = (uint32_t)p;
    if (low < range) {
      uint32_t thr = (uint32_t)(-range) % range;
      while (low < thr) {
        p = (uint64_t)next() * (uint64_t)range;
        low = (uint32_t)p;
      }
    }
    return (uint32_t)(p >> 32);
  }
  template <class T>
  void shuffle(T* a, uint32_t n) {
    if (n < 2) return;
    uint32_t i = 1;
    if ((n % 2) == 0) {
      uint32_t j = below(2);
      std::swap(a[1], a[j]);
      i = 2;
    }
    while (i < n) {
      uint32_t s = i + 1;
      uint32_t xx = below(s * (s + 1));
      std::swap(a[i], a[xx / (s + 1)]);
      std::swap(a[i + 1], a[xx % (s + 1)]);
      i += 2;
    }
  }
};

struct Stats {  // cpp/include/mettagrid/systems/stats_tracker.hpp:25-147 with a compile-time name table
  std::vector<float> v;
  std::vector<uint8_t> touched;
  void init(int n) { v.assign(n, 0.f); touched.assign(n, 0); }
  void add(int id, float a) { if (id < 0) return; touched[id] = 1; v[id] += a; }
  void set(int id, float a) { if (id < 0) return; touched[id] = 1; v[id] = a; }
  float get(int id) const { return id < 0 ? 0.f : v[id]; }
  void touch(int id) { if (id >= 0) touched[id] = 1; }
};

struct Obj {
  int cls = 0, r = 0, c = 0, vibe = 0;
  bool alive = true;
  int agent = -1;
  uint32_t visited = 0;
  uint16_t inv[MGX_MAX_RESOURCES] = {0};
  uint8_t order[MGX_MAX_RESOURCES];  // inventory iteration order (front = begin())
  int norder = 0;
  uint32_t tags[MGX_TAG_WORDS] = {0};  // GridObject::tag_bits (core/grid_object.hpp:117)
  bool no_inv_obs = false;             // spawned objects are created without an ObservationEncoder
};

struct Agent {
  int obj = 0;
  int prev_r = 0, prev_c = 0;      // Agent::prev_location (action_handler.hpp:95)
  int spawn_r = 0, spawn_c = 0;
  int step_prev_r = 0, step_prev_c = 0;  // MettaGrid::_prev_agent_locations (mettagrid_c.cpp:929-931)
  uint32_t steps_without_motion = 0;
  uint32_t max_dist = 0, unique = 0;
  std::vector<uint8_t> seen;  // unique_cells_visited
  std::vector<float> reward_prev;
  Stats stats;
  int inv_k[MGX_INVALID_EXTRA] = {};      // "action.invalid_index.<k>" for k without a stat column (include/mgx.h)
  float inv_n[MGX_INVALID_EXTRA] = {};
};

struct Deferred {  // AOETracker::apply_fixed accumulators (core/aoe_tracker.cpp:283-289)
  int delta[MGX_MAX_RESOURCES] = {0};
  bool seen[MGX_MAX_RESOURCES] = {false};
  int order[MGX_MAX_RESOURCES];
  int n = 0;
};

struct Ctx {  // handler/handler_context.hpp:38-112
  int actor = -1, target = -1, source = -1;  // object index, MGX_SLOT_NONE or MGX_SLOT_PROXY
  int proxy_tag = -1;                        // tag carried by the territory proxy cell
  int target_r = 0, target_c = 0;
  int move_direction = 0;
  bool mutation_failed = false;
  bool skip_trigger = false;                 // skip_on_update_trigger
  Deferred* deferred = nullptr;
};

struct FixedSrc { int obj, aoe, r, c; bool alive; std::vector<uint8_t> inside; };
struct MobileSrc { int obj, aoe; bool alive; std::vector<uint8_t> inside; };
struct TerrSrc { int obj, ctrl, r, c; };

struct Engine {
  const int32_t* P = nullptr;
  std::vector<int32_t> prog;
  int H, W, A, R, T, base, nact, max_steps;
  std::vector<int> grid;  // 0 empty else obj index + 1
  std::vector<Obj> objs;
  std::vector<Agent> agents;
  Stats game;
  MT rng;
  uint32_t step = 0;
  int error = 0;
  std::vector<int> tag_lists[256];   // TagIndex::_objects_by_tag (core/tag_index.cpp)
  std::vector<FixedSrc> fixed;       // AOETracker fixed sources in registration order
  std::vector<MobileSrc> mobile;
  std::vector<TerrSrc> terr;         // TerritoryTracker sources in registration order
  std::vector<int> terr_prev;        // [A][num_territories] previous owner tag or -1
  size_t next_event = 0;
  std::vector<int> deferred_aoe;     // AOETracker::_deferred_registrations (objects spawned this tick)
  std::vector<uint8_t> obs;
  std::vector<float> rewards, episode_rewards;
  std::vector<uint8_t> terminals, truncations, action_success;

  // -- program accessors --
  const int32_t* sec(int s) const { return P + mgx_sec_off(P, s); }
  const int32_t* cls(int c) const { return sec(MGX_SEC_CLASSES) + c * MGX_C_WORDS; }
  int feat(int f) const { return P[MGX_H_FEAT_BASE + f]; }
  int wk(int s) const { return P[MGX_H_STAT_BASE + s]; }
  bool is_agent(int o) const { return o >= 0 && objs[o].agent >= 0; }

  // ---- Inventory (cpp/src/mettagrid/objects/inventory.cpp) ----------------------------------------------------
  int effective_limit(const Obj& o, const int32_t* L) const {  // objects/inventory.hpp:26-40
    int sum = 0;
    const int32_t* mods = sec(MGX_SEC_MODS) + L[MGX_L_MOD_START] * MGX_MOD_WORDS;
    for (int i = 0; i < L[MGX_L_MOD_COUNT]; i++)
      sum += (int)o.inv[mods[i * MGX_MOD_WORDS + MGX_MOD_ITEM]] * mods[i * MGX_MOD_WORDS + MGX_MOD_BONUS];
    int eff = std::min(L[MGX_L_MAX], std::max(L[MGX_L_MIN], sum));
    return std::clamp(eff, 0, 65535);
  }
  int group_amount(const Obj& o, const int32_t* L) const {
    int s = 0;
    for (int r = 0; r < R; r++)
      if (L[MGX_L_RES_MASK] & (1 << r)) s += o.inv[r];
    return s;
  }
  const int32_t* limit_of(const Obj& o, int item) const {
    int li = cls(o.cls)[MGX_C_RES_LIMIT + item];
    return li < 0 ? nullptr : sec(MGX_SEC_LIMITS) + li * MGX_L_WORDS;
  }
  void order_insert_front(Obj& o, int item) {
    for (int i = o.norder; i > 0; i--) o.order[i] = o.order[i - 1];
    o.order[0] = (uint8_t)item;
    o.norder++;
  }
  void order_erase(Obj& o, int item) {
    int k = 0;
    for (int i = 0; i < o.norder; i++)
      if (o.order[i] != item) o.order[k++] = o.order[i];
    o.norder = k;
  }
  void on_inventory_change(int oi, int item, int delta) {  // objects/agent.cpp:106-121
    Obj& o = objs[oi];
    if (o.agent < 0 || delta == 0) return;
    Stats& st = agents[o.agent].stats;
    if (delta > 0) st.add(wk(MGX_S_RES_GAINED_BASE) + item, (float)delta);
    else st.add(wk(MGX_S_RES_LOST_BASE) + item, (float)(-delta));
    st.set(wk(MGX_S_RES_AMOUNT_BASE) + item, (float)o.inv[item]);
    if (o.inv[item] == 0 && delta < 0 && item == P[MGX_H_HP_RESOURCE]) st.add(wk(MGX_S_DEATH), 1.f);
  }
  int inv_update(int oi, int item, int delta, bool ignore_limits = false, bool notify = true) {  // inventory.cpp:38-86
    Obj& o = objs[oi];
    int initial = o.inv[item];
    int new_amount = initial + delta;
    int mx = 65535;
    const int32_t* L = limit_of(o, item);
    if (!ignore_limits && L) {
      int eff = effective_limit(o, L);
      int used = group_amount(o, L) - initial;
      if (used < 0) used = 0;
      int m = eff - used;
      if (m < 0) m = 0;
      mx = m;
    }
    int clamped = std::clamp(new_amount, 0, mx);
    if (initial == 0 && clamped != 0) order_insert_front(o, item);
    if (initial != 0 && clamped == 0) order_erase(o, item);
    o.inv[item] = (uint16_t)clamped;
    int d = clamped - initial;
    if (notify && d != 0) on_inventory_change(oi, item, d);
    if (d < 0 && (cls(o.cls)[MGX_C_MODIFIER_MASK] & (1 << item))) enforce_all_limits(oi);
    return d;
  }
  void enforce_all_limits(int oi) {  // inventory.cpp:141-173
    const int32_t* C = cls(objs[oi].cls);
    for (int li = 0; li < C[MGX_C_LIMIT_COUNT]; li++) {
      const int32_t* L = sec(MGX_SEC_LIMITS) + (C[MGX_C_LIMIT_START] + li) * MGX_L_WORDS;
      if (L[MGX_L_DROP_COUNT] == 0) continue;
      int excess = group_amount(objs[oi], L) - effective_limit(objs[oi], L);
      if (excess <= 0) continue;
      const int32_t* drop = sec(MGX_SEC_DROP_ORDER) + L[MGX_L_DROP_START];
      for (int k = 0; k < L[MGX_L_DROP_COUNT]; k++) {
        int item = drop[k];
        int to_drop = std::min((int)objs[oi].inv[item], excess);
        if (to_drop > 0) {
          inv_update(oi, item, -to_drop);
          excess = group_amount(objs[oi], L) - effective_limit(objs[oi], L);
        }
        if (excess <= 0) break;
      }
    }
  }
  int free_space(int oi, int item) const {  // inventory.cpp:97-110
    const Obj& o = objs[oi];
    const int32_t* L = limit_of(o, item);
    if (!L) return 65535 - o.inv[item];
    int used = group_amount(o, L), eff = effective_limit(o, L);
    return eff > used ? eff - used : 0;
  }
  int transfer(int src, int dst, int item, int delta) {  // objects/has_inventory.cpp:76-108
    if (delta <= 0) return 0;
    int give = std::min((int)objs[src].inv[item], delta);
    int amount = std::min(give, free_space(dst, item));
    inv_update(src, item, -amount);
    inv_update(dst, item, amount);
    return amount;
  }

  // ---- game values (cpp/src/mettagrid/core/game_value.cpp:14-148) ---------------------------------------------
  float eval_value(int rec, int entity, const Ctx& c) {
    const int32_t* V = sec(MGX_SEC_OBS_VALUES) + rec * MGX_OV_WORDS;
    return eval_code(V[MGX_OV_GV_START], V[MGX_OV_GV_COUNT], entity, c);
  }
  float eval_code(int start, int count, int entity, const Ctx& outer) {
    float st[32];
    int sp = 0;
    const int32_t* code = sec(MGX_SEC_GV_CODE) + start * MGX_GV_WORDS;
    for (int i = 0; i < count; i++, code += MGX_GV_WORDS) {
      int a0 = code[MGX_GV_A0], a1 = code[MGX_GV_A1], a2 = code[MGX_GV_A2];
      switch (code[MGX_GV_OP]) {
        case MGX_GOP_INVENTORY: st[sp++] = entity >= 0 ? (float)objs[entity].inv[a0] : 0.f; break;
        case MGX_GOP_STAT: {
          Stats* tr = a0 == 1 ? &game : (is_agent(entity) ? &agents[objs[entity].agent].stats : nullptr);
          if (tr) { tr->touch(a1); st[sp++] = tr->get(a1); } else st[sp++] = 0.f;
          break;
        }
        case MGX_GOP_CONST: { float f; std::memcpy(&f, &a0, 4); st[sp++] = f; break; }
        case MGX_GOP_ADD_TERM: {
          float t = st[--sp];
          if (a0) t = std::log(t + 1.0f);
          if (a1) { float w; std::memcpy(&w, &a2, 4); t *= w; }
          float acc = st[--sp];
          st[sp++] = acc + t;
          break;
        }
        case MGX_GOP_RATIO: { float den = st[--sp], num = st[--sp]; st[sp++] = den > 0.f ? num / den : num; break; }
        case MGX_GOP_MAX2: { float v = st[--sp], b = st[--sp]; st[sp++] = std::max(b, v); break; }
        case MGX_GOP_MIN2: { float v = st[--sp], b = st[--sp]; st[sp++] = std::min(b, v); break; }
        case MGX_GOP_QUERY_INVENTORY: {  // game_value.cpp:45-57 (captured ctx: actor = the value's entity)
          Ctx q = outer; q.actor = entity;
          float total = 0.f;
          for (int o : eval_query(a1, q)) total += (float)objs[o].inv[a0];
          st[sp++] = total;
          break;
        }
        case MGX_GOP_QUERY_COUNT: {
          Ctx q = outer; q.actor = entity;
          st[sp++] = (float)eval_query(a0, q).size();
          break;
        }
      }
    }
    return sp > 0 ? st[sp - 1] : 0.f;
  }

  // ---- tags -------------------------------------------------------------------------------------------------
  bool has_tag(int o, int t) const { return o >= 0 && ((objs[o].tags[t >> 5] >> (t & 31)) & 1u); }
  void tags_of(int o, const Ctx& c, uint32_t* out) const {  // proxy cell carries exactly the winning tag
    for (int w = 0; w < MGX_TAG_WORDS; w++) out[w] = o >= 0 ? objs[o].tags[w] : 0u;
    if (o == MGX_SLOT_PROXY && c.proxy_tag >= 0) out[c.proxy_tag >> 5] |= 1u << (c.proxy_tag & 31);
  }
  void fire_tag_handlers(int o, int tag, int start_field, const Ctx& c) {  // core/grid_object.cpp:83-123
    const int32_t* C = cls(objs[o].cls);
    const int32_t* th = sec(MGX_SEC_TAG_HANDLERS) + C[start_field] * MGX_TH_WORDS;
    for (int i = 0; i < C[start_field + 1]; i++, th += MGX_TH_WORDS) {
      if (th[MGX_TH_TAG] != tag) continue;
      Ctx h = c;
      h.actor = h.target = o;
      h.skip_trigger = false;
      apply_handler(th[MGX_TH_HANDLER], h);
    }
  }
  void add_tag(int o, int tag, const Ctx& c) {
    if (o < 0 || tag < 0 || tag >= 256 || has_tag(o, tag)) return;
    objs[o].tags[tag >> 5] |= 1u << (tag & 31);
    tag_lists[tag].push_back(o);
    if (!c.skip_trigger) fire_tag_handlers(o, tag, MGX_C_TAG_ADD_START, c);
  }
  void remove_tag(int o, int tag, const Ctx& c) {
    if (o < 0 || tag < 0 || tag >= 256 || !has_tag(o, tag)) return;
    objs[o].tags[tag >> 5] &= ~(1u << (tag & 31));
    auto& v = tag_lists[tag];
    v.erase(std::remove(v.begin(), v.end(), o), v.end());
    if (!c.skip_trigger) fire_tag_handlers(o, tag, MGX_C_TAG_REMOVE_START, c);
  }

  // ---- queries (cpp/src/mettagrid/core/query_system.cpp:28-89, 178-330) ---------------------------------------
  bool matches(int pc, Ctx c, int obj) { c.target = obj; return pc == MGX_PC_PASS || check_filters(pc, c); }
  std::vector<int> apply_limits(std::vector<int> res, const int32_t* Q, const Ctx& c) {
    if (Q[MGX_Q_ORDER] & 1) rng.shuffle(res.data(), (uint32_t)res.size());
    int mx = Q[MGX_Q_MAX_ITEMS] >= 0 ? (int)eval_value(Q[MGX_Q_MAX_ITEMS], c.actor, c) : -1;
    if (mx >= 0 && (int)res.size() > mx) res.resize(mx);
    return res;
  }
  std::vector<int> eval_query(int qi, const Ctx& c) {
    const int32_t* Q = sec(MGX_SEC_QUERIES) + qi * MGX_Q_WORDS;
    std::vector<int> res;
    switch (Q[MGX_Q_KIND]) {
      case MGX_QK_TAG:
        for (int o : tag_lists[Q[MGX_Q_A0]]) if (matches(Q[MGX_Q_A1], c, o)) res.push_back(o);
        break;
      case MGX_QK_FILTERED:
        for (int o : eval_query(Q[MGX_Q_A0], c)) if (matches(Q[MGX_Q_A1], c, o)) res.push_back(o);
        break;
      case MGX_QK_CLOSURE: {
        std::vector<int> roots = eval_query(Q[MGX_Q_A0], c);
        if (Q[MGX_Q_A1] < 0) return apply_limits(std::move(roots), Q, c);
        std::vector<int> pool = eval_query(Q[MGX_Q_A1], c);
        std::vector<uint8_t> visited(objs.size(), 0);
        std::vector<int> frontier;
        for (int o : roots) if (!visited[o]) { visited[o] = 1; frontier.push_back(o); res.push_back(o); }
        for (size_t head = 0; head < frontier.size(); head++) {
          int cur = frontier[head];
          for (int cand : pool) {
            if (visited[cand]) continue;
            if (Q[MGX_Q_A2] != MGX_PC_PASS) {
              Ctx e = c; e.source = cur; e.target = cand;
              if (!check_filters(Q[MGX_Q_A2], e)) continue;
            }
            visited[cand] = 1; frontier.push_back(cand); res.push_back(cand);
          }
        }
        if (Q[MGX_Q_A3] != MGX_PC_PASS) {
          std::vector<int> f;
          for (int o : res) if (matches(Q[MGX_Q_A3], c, o)) f.push_back(o);
          res = std::move(f);
        }
        break;
      }
      case MGX_QK_RAYCAST: {
        std::vector<int> sources = eval_query(Q[MGX_Q_A0], c);
        std::vector<uint8_t> seen(objs.size(), 0);
        const int32_t* dirs = sec(MGX_SEC_WORDLIST) + Q[MGX_Q_A2];
        for (int src : sources) {
          Ctx sc = c; sc.actor = sc.target = src;
          int range = (int)eval_value(Q[MGX_Q_A1], src, sc);
          if (range <= 0) continue;
          for (int d = 0; d < Q[MGX_Q_A3]; d++)
            for (int dist = 1; dist <= range; dist++) {
              int r = objs[src].r + dirs[d * 2] * dist, cc = objs[src].c + dirs[d * 2 + 1] * dist;
              if (r < 0 || cc < 0 || r >= H || cc >= W) break;
              int o = grid[r * W + cc] - 1;
              if (o < 0) continue;
              bool blocker = false;
              if (Q[MGX_Q_A4] != MGX_PC_FAIL) { Ctx b = c; b.target = o; blocker = check_filters(Q[MGX_Q_A4], b); }
              if (blocker) {
                if ((Q[MGX_Q_ORDER] & 0x100) && !seen[o]) { seen[o] = 1; res.push_back(o); }
                break;
              }
              if (!seen[o]) { seen[o] = 1; res.push_back(o); }
            }
        }
        break;
      }
    }
    return apply_limits(std::move(res), Q, c);
  }
  void compute_all_queries() {  // query_system.cpp:91-117
    Ctx t; t.skip_trigger = true;
    const int32_t* mq = sec(MGX_SEC_MATQ);
    for (int i = 0; i < P[MGX_H_NUM_MATQ]; i++, mq += MGX_MQ_WORDS) {
      std::vector<int> tagged = tag_lists[mq[MGX_MQ_TAG]];
      for (int o : tagged) remove_tag(o, mq[MGX_MQ_TAG], t);
      Ctx g;
      for (int o : eval_query(mq[MGX_MQ_QUERY], g)) add_tag(o, mq[MGX_MQ_TAG], t);
    }
  }
  void recompute_query(int tag, const Ctx& c) {  // query_system.cpp:119-175
    Ctx t = c; t.skip_trigger = true;
    std::vector<int> lost, keep;
    const int32_t* mq = sec(MGX_SEC_MATQ);
    for (int i = 0; i < P[MGX_H_NUM_MATQ]; i++, mq += MGX_MQ_WORDS) {
      if (mq[MGX_MQ_TAG] != tag) continue;
      lost = tag_lists[tag];
      for (int o : lost) { t.actor = t.target = o; remove_tag(o, tag, t); }
      for (int o : eval_query(mq[MGX_MQ_QUERY], c)) {
        if (std::find(keep.begin(), keep.end(), o) == keep.end()) keep.push_back(o);
        t.actor = t.target = o; add_tag(o, tag, t);
      }
      break;
    }
    t.skip_trigger = false;
    for (int o : lost)
      if (std::find(keep.begin(), keep.end(), o) == keep.end()) { t.actor = t.target = o; fire_tag_handlers(o, tag, MGX_C_TAG_REMOVE_START, t); }
    for (int o : keep)
      if (std::find(lost.begin(), lost.end(), o) == lost.end()) { t.actor = t.target = o; fire_tag_handlers(o, tag, MGX_C_TAG_ADD_START, t); }
  }

  // ---- filters (cpp/include/mettagrid/handler/filters/*.hpp), short-circuit code ------------------------------
  int resolve(const Ctx& c, int ent) const { return ent == MGX_ENT_ACTOR ? c.actor : c.target; }
  int inv_of(int e, int item) const { return e >= 0 ? objs[e].inv[item] : 0; }
  bool atom(const int32_t* a, const Ctx& c) {
    int a0 = a[MGX_AT_A0], a1 = a[MGX_AT_A1], a2 = a[MGX_AT_A2];
    switch (a[MGX_AT_OP]) {
      case MGX_FOP_VIBE: { int e = resolve(c, a0); return e != MGX_SLOT_NONE && (e >= 0 ? objs[e].vibe : 0) == a1; }
      case MGX_FOP_RESOURCE: { int e = resolve(c, a0); return e != MGX_SLOT_NONE && inv_of(e, a1) >= a2; }
      case MGX_FOP_SHARED_TAG: {
        uint32_t x[MGX_TAG_WORDS], y[MGX_TAG_WORDS];
        const int32_t* mask = sec(MGX_SEC_WORDLIST) + a0;
        tags_of(c.actor, c, x);
        tags_of(c.target, c, y);
        for (int w = 0; w < MGX_TAG_WORDS; w++) if (x[w] & y[w] & (uint32_t)mask[w]) return true;
        return false;
      }
      case MGX_FOP_TAG: {
        int e = resolve(c, a0);
        if (e == MGX_SLOT_NONE) return false;
        uint32_t x[MGX_TAG_WORDS];
        const int32_t* mask = sec(MGX_SEC_WORDLIST) + a1;
        tags_of(e, c, x);
        for (int w = 0; w < MGX_TAG_WORDS; w++) if (x[w] & (uint32_t)mask[w]) return true;
        return false;
      }
      case MGX_FOP_TARGET_LOC_EMPTY: return c.target == MGX_SLOT_NONE;
      case MGX_FOP_TARGET_IS_USABLE: return c.target != MGX_SLOT_NONE;  // every GridObject derives from Usable
      case MGX_FOP_PERIODIC: return step >= (uint32_t)a1 && ((step - (uint32_t)a1) % (uint32_t)a0) == 0;
      case MGX_FOP_GAME_VALUE: {
        int e = resolve(c, a0);
        float v = eval_value(a1, e, c);
        float t = eval_value(a2, e, c);
        return v >= t;
      }
      case MGX_FOP_MAX_DISTANCE: {  // filters/max_distance_filter.hpp:27-67
        int e = resolve(c, a0);
        if (e < 0) return false;
        long r = a1;
        if (a2 < 0) {
          int ref = c.source >= 0 ? c.source : c.actor;
          if (ref < 0) return false;
          if (a1 == 0) return true;
          long dr = objs[e].r - objs[ref].r, dc = objs[e].c - objs[ref].c;
          return dr * dr + dc * dc <= r * r;
        }
        std::vector<int> srcs = eval_query(a2, c);
        if (a1 == 0) return !srcs.empty();
        for (int sidx : srcs) {
          long dr = objs[e].r - objs[sidx].r, dc = objs[e].c - objs[sidx].c;
          if (dr * dr + dc * dc <= r * r) return true;
        }
        return false;
      }
      case MGX_FOP_QUERY_RESOURCE: {  // filters/query_resource_filter.hpp:26-41
        std::vector<int> res = eval_query(a0, c);
        const int32_t* req = sec(MGX_SEC_WORDLIST) + a1;
        for (int i = 0; i < a2; i++) {
          uint32_t total = 0;
          for (int o : res) { total += objs[o].inv[req[i * 2]]; if (total >= (uint32_t)req[i * 2 + 1]) break; }
          if (total < (uint32_t)req[i * 2 + 1]) return false;
        }
        return true;
      }
      case MGX_FOP_TRUE: return true;
      default: return false;
    }
  }
  bool check_filters(int pc, const Ctx& c) {  // handler/handler.cpp:95-103
    const int32_t* atoms = sec(MGX_SEC_ATOMS);
    while (pc >= 0) {
      const int32_t* a = atoms + pc * MGX_AT_WORDS;
      pc = atom(a, c) ? a[MGX_AT_ON_TRUE] : a[MGX_AT_ON_FALSE];
    }
    return pc == MGX_PC_PASS;
  }

  // ---- grid (cpp/include/mettagrid/core/grid.hpp:75-113) ------------------------------------------------------
  bool move_object(int oi, int r, int c) {
    if (r < 0 || c < 0 || r >= H || c >= W) return false;
    if (grid[r * W + c] != 0) return false;
    Obj& o = objs[oi];
    grid[r * W + c] = oi + 1;
    grid[o.r * W + o.c] = 0;
    o.r = r; o.c = c;
    return true;
  }

  // ---- dynamic objects ----------------------------------------------------------------------------------------
  int spawn_object(int cls_id, int r, int c) {  // create_object_from_config + Grid::add_object + TagIndex + deferred AoE
    const int32_t* C = cls(cls_id);
    if (C[MGX_C_KIND] == MGX_KIND_AGENT) { error |= 64; return -1; }  // spawning agents is not supported
    Obj o;
    o.cls = cls_id; o.r = r; o.c = c; o.vibe = C[MGX_C_INITIAL_VIBE];
    o.no_inv_obs = true;
    for (int w = 0; w < MGX_TAG_WORDS; w++) o.tags[w] = (uint32_t)C[MGX_C_TAGS + w];
    int oi = (int)objs.size();
    objs.push_back(o);
    grid[r * W + c] = oi + 1;
    const int32_t* ii = sec(MGX_SEC_INIT_INV) + C[MGX_C_INIT_INV_START] * MGX_II_WORDS;
    for (int i = 0; i < C[MGX_C_INIT_INV_COUNT]; i++, ii += MGX_II_WORDS) inv_update(oi, ii[MGX_II_ITEM], ii[MGX_II_AMOUNT], true, true);
    for (int t = 0; t < 256; t++) if (has_tag(oi, t)) tag_lists[t].push_back(oi);
    if (C[MGX_C_AOE_COUNT] > 0) deferred_aoe.push_back(oi);
    return oi;
  }
  void remove_object(int oi) {  // resource_mutation.hpp:88-97: aoe unregister, Grid::remove_from_grid, TagIndex unregister
    if (is_agent(oi)) { error |= 64; return; }
    for (auto& f : fixed)
      if (f.alive && f.obj == oi) {
        for (int ai = 0; ai < A; ai++) if (f.inside[ai]) { f.inside[ai] = 0; presence(f.aoe, agents[ai].obj, -1); }
        f.alive = false;
      }
    for (auto& mo : mobile)
      if (mo.alive && mo.obj == oi) {
        for (int ai = 0; ai < A; ai++) if (mo.inside[ai]) { mo.inside[ai] = 0; presence(mo.aoe, agents[ai].obj, -1); }
        mo.alive = false;
      }
    grid[objs[oi].r * W + objs[oi].c] = 0;
    objs[oi].alive = false;
    for (int t = 0; t < 256; t++)
      if (has_tag(oi, t)) { auto& v = tag_lists[t]; v.erase(std::remove(v.begin(), v.end(), oi), v.end()); }
  }

  // ---- mutations (cpp/include/mettagrid/handler/mutations/*.hpp) ----------------------------------------------
  void mutate(const int32_t* m, Ctx& c) {
    int a0 = m[MGX_MU_A0], a1 = m[MGX_MU_A1], a2 = m[MGX_MU_A2], a3 = m[MGX_MU_A3];
    switch (m[MGX_MU_OP]) {
      case MGX_MOP_RESOURCE_DELTA: {  // resource_mutation.hpp:25-47
        if (c.deferred && a0 == MGX_ENT_TARGET && c.target >= 0 &&
            !(cls(objs[c.target].cls)[MGX_C_MODIFIER_MASK] & (1 << a1))) {
          Deferred& D = *c.deferred;
          if (!D.seen[a1]) { D.seen[a1] = true; D.order[D.n++] = a1; }
          D.delta[a1] += a2;
          break;
        }
        int e = resolve(c, a0);
        if (e >= 0) inv_update(e, a1, a2); else if (e == MGX_SLOT_PROXY) error |= 32;
        break;
      }
      case MGX_MOP_RESOURCE_TRANSFER: {  // resource_mutation.hpp:60-98
        int s = resolve(c, a0), d = resolve(c, a1);
        if (s < 0 || d < 0) break;
        int amount = a3 < 0 ? (int)objs[s].inv[a2] : a3;
        int moved = transfer(s, d, a2, amount);
        if (moved > 0 && is_agent(s)) agents[objs[s].agent].stats.add(wk(MGX_S_RES_DEPOSITED_BASE) + a2, (float)moved);
        if (m[MGX_MU_A4] && objs[s].norder == 0) remove_object(s);  // resource_mutation.hpp:88-97
        break;
      }
      case MGX_MOP_CLEAR_INVENTORY: {  // resource_mutation.hpp:111-128
        int e = resolve(c, a0);
        if (e < 0) break;
        if (a2 == 0) {
          uint8_t items[MGX_MAX_RESOURCES];
          int n = objs[e].norder;
          std::memcpy(items, objs[e].order, n);
          for (int i = 0; i < n; i++) inv_update(e, items[i], -(int)objs[e].inv[items[i]]);
        } else {
          const int32_t* ids = sec(MGX_SEC_WORDLIST) + a1;
          for (int i = 0; i < a2; i++) inv_update(e, ids[i], -(int)objs[e].inv[ids[i]]);
        }
        break;
      }
      case MGX_MOP_ATTACK: {  // attack_mutation.hpp:20-38
        if (c.actor < 0 || c.target < 0) break;
        int weapon = objs[c.actor].inv[a0], armor = objs[c.target].inv[a1];
        int raw = (weapon * a3) / 100;
        int dmg = std::max(0, raw - armor);
        if (dmg > 0) inv_update(c.target, a2, -dmg);
        break;
      }
      case MGX_MOP_STATS: {  // stats_mutation.hpp:21-41: a0 scope (0 game, 1 agent), a1 entity, a2 stat, a3 value
        int e = resolve(c, a1);
        float v = eval_value(a3, e, c);
        if (a0 == 0) game.set(a2, v);
        else if (is_agent(e)) agents[objs[e].agent].stats.set(a2, v);
        break;
      }
      case MGX_MOP_CHANGE_VIBE: { int e = resolve(c, a0); if (e >= 0) objs[e].vibe = a1; break; }
      case MGX_MOP_ADD_TAG: add_tag(resolve(c, a0), a1, c); break;          // tag_mutation.hpp:16-30
      case MGX_MOP_REMOVE_TAG: remove_tag(resolve(c, a0), a1, c); break;    // :32-45
      case MGX_MOP_REMOVE_TAGS_PREFIX: {                                    // :47-67
        int e = resolve(c, a0);
        const int32_t* ids = sec(MGX_SEC_WORDLIST) + a1;
        for (int i = 0; i < a2; i++) remove_tag(e, ids[i], c);
        break;
      }
      case MGX_MOP_GAME_VALUE: {  // game_value_mutation.hpp:21-27 (a0 target entity, a1 value, a2 source)
        int e = resolve(c, a0);
        float delta = eval_value(a2, e, c);
        const int32_t* V = sec(MGX_SEC_OBS_VALUES) + a1 * MGX_OV_WORDS;
        const int32_t* code = sec(MGX_SEC_GV_CODE) + V[MGX_OV_GV_START] * MGX_GV_WORDS;
        if (code[MGX_GV_OP] == MGX_GOP_INVENTORY) {
          if (e >= 0) inv_update(e, code[MGX_GV_A0], (int)delta);
        } else if (code[MGX_GV_OP] == MGX_GOP_STAT) {
          Stats* tr = code[MGX_GV_A0] == 1 ? &game : (is_agent(e) ? &agents[objs[e].agent].stats : nullptr);
          if (tr) tr->add(code[MGX_GV_A1], delta);
        }
        break;
      }
      case MGX_MOP_RECOMPUTE_QUERY: recompute_query(a0, c); break;
      case MGX_MOP_QUERY_INVENTORY: {  // query_inventory_mutation.hpp:26-52
        std::vector<int> res = eval_query(a0, c);
        const int32_t* dl = sec(MGX_SEC_WORDLIST) + a1;
        const int32_t* sn = sec(MGX_SEC_WORDLIST) + m[MGX_MU_A4];
        int nsn = m[MGX_MU_PAD0];
        if (a3 >= 0) {
          int srcobj = resolve(c, a3);
          if (srcobj < 0) break;
          for (int o : res)
            for (int i = 0; i < a2; i++) {
              int item = dl[i * 2], delta = dl[i * 2 + 1], actual = 0;
              if (delta > 0) actual = transfer(srcobj, o, item, delta);
              else if (delta < 0) actual = transfer(o, srcobj, item, -delta);
              if (actual != 0)
                for (int k = nsn - 1; k >= 0; k--)  // later duplicates win in the reference's map build
                  if (sn[k * 2] == item) { game.add(sn[k * 2 + 1], (float)actual); break; }
            }
        } else {
          for (int o : res)
            for (int i = 0; i < a2; i++) inv_update(o, dl[i * 2], dl[i * 2 + 1]);
        }
        break;
      }
      case MGX_MOP_PUSH_OBJECT: {  // push_object_mutation.hpp:33-67
        if (c.actor < 0 || c.target < 0) { c.mutation_failed = true; break; }
        int dr = std::clamp(objs[c.target].r - objs[c.actor].r, -1, 1), dc = std::clamp(objs[c.target].c - objs[c.actor].c, -1, 1);
        int nr = objs[c.target].r + dr, nc = objs[c.target].c + dc;
        if (nr < 0 || nc < 0 || nr >= H || nc >= W || grid[nr * W + nc] != 0) { c.mutation_failed = true; break; }
        int orr = objs[c.target].r, occ = objs[c.target].c;
        if (!move_object(c.target, nr, nc)) { c.mutation_failed = true; break; }
        territory_moved(c.target, orr, occ);
        break;
      }
      case MGX_MOP_SPAWN_OBJECT: {  // spawn_object_mutation.cpp:10-64
        if (grid[c.target_r * W + c.target_c] != 0) { c.mutation_failed = true; break; }
        c.target = spawn_object(a0, c.target_r, c.target_c);
        break;
      }
      case MGX_MOP_RAYCAST_SPAWN: {  // raycast_spawn_mutation.cpp:15-92
        if (c.target < 0) { c.mutation_failed = true; break; }
        int orr = objs[c.target].r, occ = objs[c.target].c;
        Ctx tc = c; tc.actor = c.target;
        int range = (int)eval_value(a3, c.target, tc);
        if (range <= 0) break;
        const int32_t* dirs = sec(MGX_SEC_WORDLIST) + a1;
        for (int k = 0; k < a2; k++)
          for (int dist = 1; dist <= range; dist++) {
            int r = orr + dirs[k * 2] * dist, cc = occ + dirs[k * 2 + 1] * dist;
            if (r < 0 || cc < 0 || r >= H || cc >= W) break;
            int ex = grid[r * W + cc] - 1;
            if (ex >= 0) {
              bool blocker = false;
              if (m[MGX_MU_A4] != MGX_PC_FAIL) { Ctx b = c; b.target = ex; blocker = check_filters(m[MGX_MU_A4], b); }
              if (blocker) break;
              continue;
            }
            spawn_object(a0, r, cc);
          }
        break;
      }
      case MGX_MOP_RELOCATE:
        if (is_agent(c.actor)) {
          int orr = objs[c.actor].r, occ = objs[c.actor].c;
          if (move_object(c.actor, c.target_r, c.target_c)) territory_moved(c.actor, orr, occ);
        }
        break;
      case MGX_MOP_SWAP: {  // swap_mutation.hpp:15-21, core/grid.hpp:92-105
        if (!is_agent(c.actor) || !is_agent(c.target)) break;
        Obj &x = objs[c.actor], &y = objs[c.target];
        grid[x.r * W + x.c] = c.target + 1;
        grid[y.r * W + y.c] = c.actor + 1;
        int xr = x.r, xc = x.c, yr = y.r, yc = y.c;
        std::swap(x.r, y.r);
        std::swap(x.c, y.c);
        territory_moved(c.actor, xr, xc);  // Grid::swap_objects -> on_object_moved twice (core/grid.hpp:100-103)
        territory_moved(c.target, yr, yc);
        agents[x.agent].stats.add(wk(MGX_S_SWAP), 1.f);
        break;
      }
      case MGX_MOP_USE_TARGET: {  // use_target_mutation.hpp:17-29, core/grid_object.cpp:72-80
        if (c.target < 0 || !is_agent(c.actor)) { c.mutation_failed = true; break; }
        int h = cls(objs[c.target].cls)[MGX_C_ON_USE];
        Ctx use = c;
        if (h < 0 || !apply_handler(h, use)) { c.mutation_failed = true; break; }
        int after = cls(objs[c.actor].cls)[MGX_C_ON_AFTER_USE];
        if (after >= 0) apply_handler(after, c);
        break;
      }
    }
  }

  bool apply_handler(int h, Ctx& c) {  // handler.cpp:76-93, multi_handler.cpp:8-21
    const int32_t* hd = sec(MGX_SEC_HANDLERS) + h * MGX_HD_WORDS;
    if (hd[MGX_HD_KIND] == MGX_HK_LEAF) {
      if (!check_filters(hd[MGX_HD_FILTER_PC], c)) return false;
      c.mutation_failed = false;
      const int32_t* m = sec(MGX_SEC_MUTS) + hd[MGX_HD_MUT_START] * MGX_MU_WORDS;
      for (int i = 0; i < hd[MGX_HD_MUT_COUNT]; i++, m += MGX_MU_WORDS) {
        mutate(m, c);
        if (c.mutation_failed) return false;
      }
      return true;
    }
    bool any = false;
    const int32_t* kids = sec(MGX_SEC_CHILDREN) + hd[MGX_HD_CHILD_START];
    for (int i = 0; i < hd[MGX_HD_CHILD_COUNT]; i++) {
      if (apply_handler(kids[i], c)) {
        any = true;
        if (hd[MGX_HD_KIND] == MGX_HK_FIRST_MATCH) return true;
      }
    }
    return any;
  }

  // Leaf handler with "apply every mutation" semantics (events, AoE sources, territory handlers do not stop on
  // mutation_failed: handler/event.cpp:86-92, core/aoe_tracker.cpp:99-113, core/territory_tracker.cpp:62-66).
  bool apply_all(int filter_pc, int mut_start, int mut_count, Ctx& c) {
    if (!check_filters(filter_pc, c)) return false;
    const int32_t* m = sec(MGX_SEC_MUTS) + mut_start * MGX_MU_WORDS;
    for (int i = 0; i < mut_count; i++, m += MGX_MU_WORDS) mutate(m, c);
    return true;
  }

  // ---- events (handler/event.cpp:34-101, handler/event_scheduler.cpp:36-53) -----------------------------------
  int execute_event(int ev) {
    const int32_t* E = sec(MGX_SEC_EVENTS) + ev * MGX_EV_WORDS;
    Ctx g;
    std::vector<int> targets = eval_query(E[MGX_EV_QUERY], g);
    int mt = E[MGX_EV_MAX_TARGETS];
    if (mt >= 0 && (int)targets.size() > mt) rng.shuffle(targets.data(), (uint32_t)targets.size());
    int applied = 0;
    for (int t : targets) {
      if (mt >= 0 && applied >= mt) break;
      Ctx c;
      c.actor = c.target = t;
      c.target_r = objs[t].r; c.target_c = objs[t].c;
      if (apply_all(E[MGX_EV_FILTER_PC], E[MGX_EV_MUT_START], E[MGX_EV_MUT_COUNT], c)) applied++;
    }
    if (applied == 0 && E[MGX_EV_FALLBACK] >= 0) return execute_event(E[MGX_EV_FALLBACK]);
    return applied;
  }
  void process_events() {
    const int32_t* sc = sec(MGX_SEC_SCHEDULE);
    size_t n = (size_t)P[MGX_H_NUM_SCHEDULE];
    while (next_event < n && (uint32_t)sc[next_event * MGX_SC_WORDS + MGX_SC_TIMESTEP] <= step) {
      execute_event(sc[next_event * MGX_SC_WORDS + MGX_SC_EVENT]);
      next_event++;
    }
  }

  // ---- AoE (core/aoe_tracker.cpp) -----------------------------------------------------------------------------
  const int32_t* aoe(int a) const { return sec(MGX_SEC_AOES) + a * MGX_AO_WORDS; }
  void register_aoes(int oi) {  // register_source :128-134, register_fixed :166-200, register_mobile :202-205
    const int32_t* C = cls(objs[oi].cls);
    for (int i = 0; i < C[MGX_C_AOE_COUNT]; i++) {
      int a = C[MGX_C_AOE_START] + i;
      if (aoe(a)[MGX_AO_STATIC]) fixed.push_back({oi, a, objs[oi].r, objs[oi].c, true, std::vector<uint8_t>(A, 0)});
      else mobile.push_back({oi, a, true, std::vector<uint8_t>(A, 0)});
    }
  }
  bool fixed_covers(const FixedSrc& f, int r, int c) const {
    const int32_t* a = aoe(f.aoe);
    long range = a[MGX_AO_RADIUS], dr = r - f.r, dc = c - f.c;
    if (dr < -range || dr > range || dc < -range || dc > range) return false;
    long d2 = dr * dr + dc * dc;
    if (d2 > range * range) return false;
    bool territory_style = a[MGX_AO_MUT_COUNT] == 0 && a[MGX_AO_PRES_COUNT] == 0 && range > 0;
    if (territory_style && range >= 2 && d2 == range * range && (dr == 0 || dc == 0)) return false;
    return true;
  }
  void presence(int a, int target, int mult) {  // apply_presence_deltas :122-126
    const int32_t* A_ = aoe(a);
    const int32_t* pr = sec(MGX_SEC_PRESENCE) + A_[MGX_AO_PRES_START] * MGX_PR_WORDS;
    for (int i = 0; i < A_[MGX_AO_PRES_COUNT]; i++, pr += MGX_PR_WORDS) inv_update(target, pr[MGX_PR_RESOURCE], pr[MGX_PR_DELTA] * mult);
  }
  void apply_fixed(int ai) {  // :278-362
    int tgt = agents[ai].obj;
    Deferred D;
    Ctx tc;
    tc.actor = MGX_SLOT_NONE; tc.target = tgt; tc.deferred = &D;
    int r = objs[tgt].r, c = objs[tgt].c;
    // exits: sources the target was inside whose cell list no longer holds the target's cell.  The reference walks
    // an unordered_set<AOESource*> here (address order); registration order is used instead.
    for (auto& f : fixed)
      if (f.alive && f.inside[ai] && !fixed_covers(f, r, c)) { f.inside[ai] = 0; presence(f.aoe, tgt, -1); }
    for (auto& f : fixed) {
      if (!f.alive || !fixed_covers(f, r, c)) continue;
      const int32_t* a = aoe(f.aoe);
      if (a[MGX_AO_MUT_COUNT] == 0 && a[MGX_AO_PRES_COUNT] == 0) continue;
      bool skip_self = !a[MGX_AO_EFFECT_SELF] && f.obj == tgt;
      Ctx fc = tc; fc.actor = f.obj;
      bool passes = !skip_self && check_filters(a[MGX_AO_FILTER_PC], fc);
      bool was = f.inside[ai];
      if (passes && !was) { f.inside[ai] = 1; presence(f.aoe, tgt, +1); }
      else if (!passes && was) { f.inside[ai] = 0; presence(f.aoe, tgt, -1); }
      if (passes && a[MGX_AO_MUT_COUNT] > 0) { Ctx ac = tc; ac.actor = f.obj; apply_all(a[MGX_AO_FILTER_PC], a[MGX_AO_MUT_START], a[MGX_AO_MUT_COUNT], ac); }
    }
    for (int k = 0; k < D.n; k++)
      if (D.delta[D.order[k]] != 0) inv_update(tgt, D.order[k], D.delta[D.order[k]]);
  }
  void apply_mobile() {  // :364-415
    for (auto& m : mobile) {
      if (!m.alive) continue;
      const int32_t* a = aoe(m.aoe);
      long range = a[MGX_AO_RADIUS];
      for (int ai = 0; ai < A; ai++) {
        int tgt = agents[ai].obj;
        if (!a[MGX_AO_EFFECT_SELF] && m.obj == tgt) continue;
        bool was = m.inside[ai];
        long dr = objs[m.obj].r - objs[tgt].r, dc = objs[m.obj].c - objs[tgt].c;
        if (dr * dr + dc * dc > range * range) {
          if (was) { m.inside[ai] = 0; presence(m.aoe, tgt, -1); }
          continue;
        }
        Ctx c; c.actor = m.obj; c.target = tgt;
        if (check_filters(a[MGX_AO_FILTER_PC], c)) {
          if (!was) { m.inside[ai] = 1; presence(m.aoe, tgt, +1); }
          if (a[MGX_AO_MUT_COUNT] > 0) { Ctx ac; ac.actor = m.obj; ac.target = tgt; apply_all(a[MGX_AO_FILTER_PC], a[MGX_AO_MUT_START], a[MGX_AO_MUT_COUNT], ac); }
        } else if (was) {
          m.inside[ai] = 0; presence(m.aoe, tgt, -1);
        }
      }
    }
  }

  // ---- territory (core/territory_tracker.cpp) -----------------------------------------------------------------
  const int32_t* tctrl(int i) const { return sec(MGX_SEC_TERR_CONTROLS) + i * MGX_TC_WORDS; }
  void register_territory(int oi) {  // register_source :102-127
    const int32_t* C = cls(objs[oi].cls);
    for (int i = 0; i < C[MGX_C_TERR_COUNT]; i++) terr.push_back({oi, C[MGX_C_TERR_START] + i, objs[oi].r, objs[oi].c});
  }
  void territory_moved(int oi, int, int) {  // notify_source_moved :187-199: entries re-registered at the end
    std::vector<TerrSrc> mine;
    for (size_t i = 0; i < terr.size();)
      if (terr[i].obj == oi) { mine.push_back(terr[i]); terr.erase(terr.begin() + i); } else i++;
    for (auto t : mine) { t.r = objs[oi].r; t.c = objs[oi].c; terr.push_back(t); }
  }
  static uint64_t floor_sqrt_u64(uint64_t value) {  // :17-33
    uint64_t root = 0, bit = 1ULL << 62;
    while (bit > value) bit >>= 2;
    while (bit != 0) {
      if (value >= root + bit) { value -= root + bit; root = (root >> 1) + bit; } else root >>= 1;
      bit >>= 2;
    }
    return root;
  }
  int cell_owner(int r, int c, int ti) const {  // compute_cell_ownership :215-252 -> winning tag or -1
    const int32_t* TE = sec(MGX_SEC_TERRITORIES) + ti * MGX_TE_WORDS;
    const int32_t* prefix = sec(MGX_SEC_WORDLIST) + TE[MGX_TE_TAGS_START];
    int64_t score[256];
    bool used[256] = {false};
    for (const auto& t : terr) {
      const int32_t* tc = tctrl(t.ctrl);
      if (tc[MGX_TC_TERRITORY] != ti) continue;
      int strength = tc[MGX_TC_STRENGTH], decay = tc[MGX_TC_DECAY];
      long range = decay > 0 ? strength / decay : strength;
      long dr = r - t.r, dc = c - t.c;
      if (dr < -range || dr > range || dc < -range || dc > range || dr * dr + dc * dc > range * range) continue;
      int tag = -1;
      for (int k = 0; k < TE[MGX_TE_TAGS_COUNT]; k++) if (has_tag(t.obj, prefix[k])) { tag = prefix[k]; break; }
      if (tag < 0) continue;
      long sr = objs[t.obj].r - r, sc = objs[t.obj].c - c;  // score uses the source's CURRENT location
      uint64_t d2 = (uint64_t)(sr * sr + sc * sc);
      int64_t sc_ = (int64_t)strength * 1024 - (int64_t)decay * (int64_t)floor_sqrt_u64(d2 * 1024ull * 1024ull);
      if (sc_ > 0) { if (!used[tag]) { used[tag] = true; score[tag] = 0; } score[tag] += sc_; }
    }
    int win = -1; int64_t best = 0; bool tied = false;
    for (int t = 0; t < 256; t++) {
      if (!used[t]) continue;
      if (score[t] > best) { win = t; best = score[t]; tied = false; }
      else if (score[t] == best && win >= 0) tied = true;
    }
    return tied ? -1 : win;
  }
  void run_territory_handlers(int start, int count, int tag, int tgt) {
    for (int i = 0; i < count; i++) {
      const int32_t* hd = sec(MGX_SEC_HANDLERS) + (start + i) * MGX_HD_WORDS;
      Ctx c; c.actor = MGX_SLOT_PROXY; c.proxy_tag = tag; c.target = tgt; c.target_r = objs[tgt].r; c.target_c = objs[tgt].c;
      apply_all(hd[MGX_HD_FILTER_PC], hd[MGX_HD_MUT_START], hd[MGX_HD_MUT_COUNT], c);
    }
  }
  void apply_territory(int ai) {  // apply_effects :275-346
    int NT = P[MGX_H_NUM_TERRITORIES];
    int tgt = agents[ai].obj;
    for (int ti = 0; ti < NT; ti++) {
      const int32_t* TE = sec(MGX_SEC_TERRITORIES) + ti * MGX_TE_WORDS;
      int cur = cell_owner(objs[tgt].r, objs[tgt].c, ti);
      int prev = terr_prev[ai * NT + ti];
      if (prev != cur && prev >= 0) run_territory_handlers(TE[MGX_TE_EXIT_START], TE[MGX_TE_EXIT_COUNT], prev, tgt);
      if (prev != cur && cur >= 0) run_territory_handlers(TE[MGX_TE_ENTER_START], TE[MGX_TE_ENTER_COUNT], cur, tgt);
      terr_prev[ai * NT + ti] = cur;
      if (cur >= 0) run_territory_handlers(TE[MGX_TE_PRES_START], TE[MGX_TE_PRES_COUNT], cur, tgt);
    }
  }
  int tile_observability(int r, int c, int observer) const {  // compute_observability_at :254-273
    for (int ti = 0; ti < P[MGX_H_NUM_TERRITORIES]; ti++) {
      int w = cell_owner(r, c, ti);
      if (w < 0) continue;
      return has_tag(observer, w) ? 1 : 2;
    }
    return 0;
  }

  // ---- actions ------------------------------------------------------------------------------------------------
  bool do_move(int ai, int orient) {  // actions/move.hpp:81-115, orientation.hpp:28-48
    static const int DX[8] = {0, 0, -1, 1, -1, 1, -1, 1}, DY[8] = {-1, 1, 0, 0, -1, -1, 1, 1};
    int oi = agents[ai].obj;
    int nmh = P[MGX_H_NUM_MOVE_HANDLERS];
    const int32_t* mh = sec(MGX_SEC_MOVE_HANDLERS);
    for (int k = 0; k < nmh; k++, mh += MGX_MH_WORDS) {
      for (int i = 1; i <= mh[MGX_MH_MAX_RANGE]; i++) {
        int r = objs[oi].r + DY[orient] * i, c = objs[oi].c + DX[orient] * i;
        if (r < 0 || c < 0 || r >= H || c >= W) break;
        int t = grid[r * W + c] - 1;
        if (t < 0 && !mh[MGX_MH_ACCEPTS_EMPTY]) continue;
        Ctx ctx;
        ctx.actor = oi; ctx.target = t; ctx.target_r = r; ctx.target_c = c; ctx.move_direction = orient;
        if (apply_handler(mh[MGX_MH_HANDLER], ctx)) return true;
        break;
      }
    }
    return false;
  }
  bool handle_action(int ai, int action) {  // actions/action_handler.hpp:78-105
    Agent& ag = agents[ai];
    const int32_t* ac = sec(MGX_SEC_ACTIONS) + action * MGX_AC_WORDS;
    int kind = ac[MGX_AC_KIND];
    bool ok = true;
    if (kind == MGX_AK_MOVE) ok = do_move(ai, ac[MGX_AC_ARG]);
    else if (kind == MGX_AK_VIBE) objs[ag.obj].vibe = ac[MGX_AC_ARG];  // actions/change_vibe.hpp:48-57
    Obj& o = objs[ag.obj];
    if (o.r == ag.prev_r && o.c == ag.prev_c) {
      ag.steps_without_motion += 1;
      if ((float)ag.steps_without_motion > ag.stats.get(wk(MGX_S_MAX_STEPS_WITHOUT_MOTION)))
        ag.stats.set(wk(MGX_S_MAX_STEPS_WITHOUT_MOTION), (float)ag.steps_without_motion);
    } else {
      ag.steps_without_motion = 0;
    }
    ag.prev_r = o.r; ag.prev_c = o.c;
    int s_ok = kind == MGX_AK_NOOP ? MGX_S_NOOP_SUCCESS : kind == MGX_AK_MOVE ? MGX_S_MOVE_SUCCESS : MGX_S_VIBE_SUCCESS;
    if (ok) ag.stats.add(wk(s_ok), 1.f);
    else { ag.stats.add(wk(s_ok + 1), 1.f); ag.stats.add(wk(MGX_S_ACTION_FAILED), 1.f); }
    return ok;
  }

  // ---- observations (cpp/bindings/mettagrid_c.cpp:665-824) ----------------------------------------------------
  struct Tok { uint8_t f, v; };
  int object_tokens(const Obj& o, Tok* out) {  // core/grid_object.cpp:178-203, objects/agent.cpp:142-154
    int n = 0;
    const int32_t* C = cls(o.cls);
    for (int t = 0; t < 256; t++)
      if (o.tags[t >> 5] & (1u << (t & 31))) out[n++] = {(uint8_t)feat(MGX_F_TAG), (uint8_t)t};
    if (o.vibe != 0) out[n++] = {(uint8_t)feat(MGX_F_VIBE), (uint8_t)o.vibe};
    for (int k = 0; k < (o.no_inv_obs ? 0 : o.norder); k++) {  // observation_encoder.hpp:198-225; grid_object.cpp:194 (obs_encoder null for spawned objects)
      int item = o.order[k];
      const int32_t* F = sec(MGX_SEC_INV_FEATURES) + item * MGX_IF_WORDS;
      uint32_t rem = o.inv[item];
      out[n++] = {(uint8_t)F[0], (uint8_t)(rem % base)};
      rem /= base;
      int p = 1;
      while (rem > 0) { out[n++] = {(uint8_t)F[p], (uint8_t)(rem % base)}; rem /= base; p++; }
    }
    if (o.agent >= 0) {
      out[n++] = {(uint8_t)feat(MGX_F_GROUP), (uint8_t)C[MGX_C_GROUP]};
      out[n++] = {(uint8_t)feat(MGX_F_AGENT_ID), (uint8_t)o.agent};
    }
    return n;
  }
  void compute_observation(int ai, int action) {
    Agent& ag = agents[ai];
    const Obj& me = objs[ag.obj];
    uint8_t* out = obs.data() + (size_t)ai * T * 3;
    size_t attempted = 0;
    auto put = [&](uint8_t loc, uint8_t f, uint8_t v) {
      if (attempted < (size_t)T) { out[attempted * 3] = loc; out[attempted * 3 + 1] = f; out[attempted * 3 + 2] = v; }
      attempted++;
    };
    int flags = P[MGX_H_GLOBAL_FLAGS];
    if (flags & MGX_G_COMPLETION) {
      uint8_t pct = 0;
      if (max_steps > 0) pct = step >= (uint32_t)max_steps ? 255 : (uint8_t)(256u * step / (uint32_t)max_steps);
      put(0xFE, feat(MGX_F_COMPLETION), pct);
    }
    if (flags & MGX_G_LAST_ACTION) put(0xFE, feat(MGX_F_LAST_ACTION), (uint8_t)action);
    if ((flags & MGX_G_LAST_ACTION_MOVE) && feat(MGX_F_LAST_ACTION_MOVE) != 0)
      put(0xFE, feat(MGX_F_LAST_ACTION_MOVE), (me.r != ag.step_prev_r || me.c != ag.step_prev_c) ? 1 : 0);
    if (flags & MGX_G_LAST_REWARD)
      put(0xFE, feat(MGX_F_LAST_REWARD), (uint8_t)std::round(rewards[ai] * 100.0f));
    if (flags & MGX_G_LOCAL_POSITION) {
      int dc = me.c - ag.spawn_c, dr = ag.spawn_r - me.r;
      if (dc > 0) put(0xFE, feat(MGX_F_LP_EAST), std::min(dc, 255));
      else if (dc < 0) put(0xFE, feat(MGX_F_LP_WEST), std::min(-dc, 255));
      if (dr > 0) put(0xFE, feat(MGX_F_LP_NORTH), std::min(dr, 255));
      else if (dr < 0) put(0xFE, feat(MGX_F_LP_SOUTH), std::min(-dr, 255));
    }
    for (int i = 0; i < P[MGX_H_NUM_OBS_VALUES]; i++) {  // mettagrid_c.cpp:1207-1238
      const int32_t* V = sec(MGX_SEC_OBS_VALUES) + i * MGX_OV_WORDS;
      Ctx vc; vc.actor = vc.target = ag.obj;
      float raw = eval_code(V[MGX_OV_GV_START], V[MGX_OV_GV_COUNT], ag.obj, vc);
      uint32_t rem = (uint32_t)raw;
      int f = V[MGX_OV_FEATURE];
      put(0xFE, f, rem % base);
      rem /= base;
      f++;
      while (rem > 0) { put(0xFE, f, rem % base); rem /= base; f++; }
    }
    int hr = P[MGX_H_OBS_HEIGHT] >> 1, wr = P[MGX_H_OBS_WIDTH] >> 1;
    const int32_t* offs = sec(MGX_SEC_OBS_OFFSETS);
    Tok toks[512];
    for (int k = 0; k < P[MGX_H_NUM_OBS_OFFSETS]; k++) {
      int r = me.r + offs[k * 2], c = me.c + offs[k * 2 + 1];
      if (r < 0 || c < 0 || r >= H || c >= W) continue;
      int oi = grid[r * W + c] - 1;
      if (feat(MGX_F_AOE_MASK) != 0) {  // _emit_tile_observability_tokens :337-362
        int mask = tile_observability(r, c, ag.obj);
        if (mask != 0) put((uint8_t)(((r - me.r + hr) << 4) | (c - me.c + wr)), feat(MGX_F_AOE_MASK), mask);
      }
      if (oi < 0) continue;
      Obj& o = objs[oi];
      if (o.visited < step) {  // mettagrid_c.cpp:789-796
        ag.stats.add(wk(MGX_S_CELL_VISITED), (float)(step - o.visited));
        o.visited = step;
      }
      uint8_t loc = (uint8_t)(((r - me.r + hr) << 4) | (c - me.c + wr));
      int n = object_tokens(o, toks);
      for (int i = 0; i < n; i++) put(loc, toks[i].f, toks[i].v);
    }
    if (attempted > (size_t)T) { error |= 1; return; }  // reference throws (mettagrid_c.cpp:813-819)
    game.v[wk(MGX_S_GAME_TOKENS_WRITTEN)] += (float)attempted;
    game.v[wk(MGX_S_GAME_TOKENS_DROPPED)] += 0.f;
    game.v[wk(MGX_S_GAME_TOKENS_FREE)] += (float)((size_t)T - attempted);
  }
  void compute_observations(const std::vector<int>& executed) {
    for (int i = 0; i < A; i++) compute_observation(i, executed[i]);
  }

  // ---- construction (mettagrid_c.cpp:42-191, 200-269, 294-319) ------------------------------------------------
  void track_coverage(Agent& ag) {  // objects/agent.cpp:49-57
    const Obj& o = objs[ag.obj];
    if (!ag.seen[o.r * W + o.c]) { ag.seen[o.r * W + o.c] = 1; ag.unique++; }
    ag.stats.set(wk(MGX_S_CELL_UNIQUE), (float)ag.unique);
    int d = std::abs(ag.spawn_r - o.r) + std::abs(o.c - ag.spawn_c);
    ag.max_dist = std::max(ag.max_dist, (uint32_t)d);
    ag.stats.set(wk(MGX_S_CELL_MAXDIST), (float)ag.max_dist);
  }
  void init_buffers() {
    std::fill(terminals.begin(), terminals.end(), 0);
    std::fill(truncations.begin(), truncations.end(), 0);
    std::fill(episode_rewards.begin(), episode_rewards.end(), 0.f);
    std::fill(rewards.begin(), rewards.end(), 0.f);
    std::fill(obs.begin(), obs.end(), 0xFF);
    std::vector<int> ex(A, 0);
    compute_observations(ex);
  }
  void create(const int32_t* program, const uint16_t* class_map, uint32_t seed) {
    prog.assign(program, program + program[MGX_H_TOTAL_WORDS]);
    P = prog.data();
    H = P[MGX_H_HEIGHT]; W = P[MGX_H_WIDTH]; A = P[MGX_H_NUM_AGENTS]; R = P[MGX_H_NUM_RESOURCES];
    T = P[MGX_H_NUM_TOKENS]; base = P[MGX_H_TOKEN_BASE]; nact = P[MGX_H_NUM_ACTIONS]; max_steps = P[MGX_H_MAX_STEPS];
    rng.seed(seed);
    grid.assign(H * W, 0);
    game.init(P[MGX_H_NUM_GAME_STATS]);
    game.touch(wk(MGX_S_GAME_TOKENS_WRITTEN));
    game.touch(wk(MGX_S_GAME_TOKENS_DROPPED));
    game.touch(wk(MGX_S_GAME_TOKENS_FREE));
    for (int r = 0; r < H; r++)
      for (int c = 0; c < W; c++) {
        int k = class_map[r * W + c];
        if (!k) continue;
        Obj o;
        o.cls = k - 1; o.r = r; o.c = c;
        const int32_t* C = cls(o.cls);
        o.vibe = C[MGX_C_INITIAL_VIBE];
        int oi = (int)objs.size();
        if (C[MGX_C_KIND] == MGX_KIND_AGENT) {
          o.agent = (int)agents.size();
          Agent ag;
          ag.obj = oi; ag.prev_r = ag.spawn_r = ag.step_prev_r = r; ag.prev_c = ag.spawn_c = ag.step_prev_c = c;
          ag.seen.assign(H * W, 0);
          ag.stats.init(P[MGX_H_NUM_AGENT_STATS]);
          ag.reward_prev.assign(C[MGX_C_REWARD_COUNT], 0.f);
          agents.push_back(std::move(ag));
        }
        for (int w = 0; w < MGX_TAG_WORDS; w++) o.tags[w] = (uint32_t)C[MGX_C_TAGS + w];
        objs.push_back(o);
        grid[r * W + c] = oi + 1;
        const int32_t* ii = sec(MGX_SEC_INIT_INV) + C[MGX_C_INIT_INV_START] * MGX_II_WORDS;
        for (int i = 0; i < C[MGX_C_INIT_INV_COUNT]; i++, ii += MGX_II_WORDS) {
          // objects/agent.cpp:79-84 (notify=false, stat set even for 0), core/grid_object_factory.cpp:83-87
          inv_update(oi, ii[MGX_II_ITEM], ii[MGX_II_AMOUNT], true, C[MGX_C_KIND] != MGX_KIND_AGENT);
          if (objs[oi].agent >= 0)
            agents[objs[oi].agent].stats.set(wk(MGX_S_RES_AMOUNT_BASE) + ii[MGX_II_ITEM], (float)ii[MGX_II_AMOUNT]);
        }
        game.add(C[MGX_C_OBJECTS_STAT], 1.f);
        for (int t = 0; t < 256; t++) if (has_tag(oi, t)) tag_lists[t].push_back(oi);  // TagIndex::register_object
      }
    // AoE / territory registration happens per object in creation order, after every agent exists so that the
    // per-agent "inside" vectors have their final size (mettagrid_c.cpp:249-257).
    A = (int)agents.size();
    for (int oi = 0; oi < (int)objs.size(); oi++) { register_aoes(oi); register_territory(oi); }
    terr_prev.assign((size_t)A * std::max(1, P[MGX_H_NUM_TERRITORIES]), -1);
    compute_all_queries();  // mettagrid_c.cpp:162-163
    for (auto& ag : agents) {
      track_coverage(ag);  // Agent::init -> reset_coverage_tracking (agent.cpp:25-28,41-47)
      const int32_t* C = cls(objs[ag.obj].cls);
      const int32_t* rw = sec(MGX_SEC_REWARDS) + C[MGX_C_REWARD_START] * MGX_RW_WORDS;
      for (int i = 0; i < C[MGX_C_REWARD_COUNT]; i++, rw += MGX_RW_WORDS) {  // systems/reward.hpp:45-53
        if (rw[MGX_RW_TOUCH_SCOPE] == 0) ag.stats.touch(rw[MGX_RW_TOUCH_STAT]);
        else if (rw[MGX_RW_TOUCH_SCOPE] == 1) game.touch(rw[MGX_RW_TOUCH_STAT]);
      }
    }
    obs.assign((size_t)A * T * 3, 0xFF);
    rewards.assign(A, 0.f); episode_rewards.assign(A, 0.f);
    terminals.assign(A, 0); truncations.assign(A, 0); action_success.assign(A, 0);
    init_buffers();
  }

  // ---- the tick (cpp/bindings/mettagrid_c.cpp:921-1102) -------------------------------------------------------
  void do_step(const int32_t* actions, const int32_t* vibe_actions) {
    for (auto& ag : agents) { ag.step_prev_r = objs[ag.obj].r; ag.step_prev_c = objs[ag.obj].c; }
    std::fill(rewards.begin(), rewards.end(), 0.f);
    std::fill(obs.begin(), obs.end(), 0xFF);
    std::fill(action_success.begin(), action_success.end(), 0);
    step++;
    std::vector<size_t> order(A);
    for (int i = 0; i < A; i++) order[i] = i;
    rng.shuffle(order.data(), (uint32_t)A);
    std::vector<int> executed(A, 0);
    int maxp = P[MGX_H_MAX_PRIORITY];
    for (int off = 0; off <= maxp; off++) {
      int prio = maxp - off;
      for (int stream = 0; stream < 2; stream++)
        for (size_t k = 0; k < order.size(); k++) {
          int ai = (int)order[k];
          int a = stream == 0 ? actions[ai] : vibe_actions[ai];
          if (a < 0 || a >= nact) {  // mettagrid_c.cpp:914-919,970-973
            Stats& st = agents[ai].stats;
            st.add(wk(MGX_S_INVALID_INDEX), 1.f);
            if (a < 0 && a >= -MGX_INVALID_WINDOW) st.add(wk(MGX_S_INVALID_NEG_BASE) + a + MGX_INVALID_WINDOW, 1.f);
            else if (a >= nact && a < nact + MGX_INVALID_WINDOW) st.add(wk(MGX_S_INVALID_POS_BASE) + a - nact, 1.f);
            else {  // a key of its own per distinct k (mettagrid_c.cpp:916-918); MGX_INVALID_EXTRA of them per agent
              Agent& ag = agents[ai];
              int q = 0;
              for (; q < MGX_INVALID_EXTRA; q++) {
                if (ag.inv_n[q] == 0.f) { ag.inv_k[q] = a; ag.inv_n[q] = 1.f; break; }
                if (ag.inv_k[q] == a) { ag.inv_n[q] += 1.f; break; }
              }
              if (q == MGX_INVALID_EXTRA) error |= 2;
            }
            action_success[ai] = 0;
            continue;
          }
          bool is_vibe = sec(MGX_SEC_ACTIONS)[a * MGX_AC_WORDS + MGX_AC_KIND] == MGX_AK_VIBE;
          if (is_vibe != (stream == 1)) continue;
          if (prio != 0) continue;  // every real action handler has priority 0
          if (handle_action(ai, a)) { executed[ai] = a; action_success[ai] = 1; }
        }
    }
    process_events();  // mettagrid_c.cpp:1009-1011
    for (auto& ag : agents) {  // per-agent on_tick (mettagrid_c.cpp:1019-1024)
      int h = cls(objs[ag.obj].cls)[MGX_C_ON_TICK];
      if (h >= 0) { Ctx c; c.actor = c.target = ag.obj; apply_handler(h, c); }
    }
    for (int ai = 0; ai < A; ai++) {  // mettagrid_c.cpp:1032-1035
      apply_fixed(ai);
      apply_territory(ai);
    }
    apply_mobile();  // :1038
    for (int oi : deferred_aoe) register_aoes(oi);  // AOETracker::flush_deferred :1042
    deferred_aoe.clear();
    if (P[MGX_H_GAME_ON_TICK] >= 0) { Ctx c; apply_handler(P[MGX_H_GAME_ON_TICK], c); }  // :1050-1052
    for (auto& ag : agents) track_coverage(ag);
    compute_observations(executed);
    for (int i = 0; i < A; i++) {  // systems/reward.hpp:56-77
      Agent& ag = agents[i];
      const int32_t* C = cls(objs[ag.obj].cls);
      const int32_t* rw = sec(MGX_SEC_REWARDS) + C[MGX_C_REWARD_START] * MGX_RW_WORDS;
      if (C[MGX_C_REWARD_COUNT] == 0) continue;
      float total = 0.f;
      for (int k = 0; k < C[MGX_C_REWARD_COUNT]; k++, rw += MGX_RW_WORDS) {
        Ctx vc; vc.actor = vc.target = ag.obj;
        float val = eval_code(rw[MGX_RW_GV_START], rw[MGX_RW_GV_COUNT], ag.obj, vc);
        if (rw[MGX_RW_ACCUMULATE]) total += val;
        else total += val - ag.reward_prev[k];
        ag.reward_prev[k] = val;
      }
      if (total != 0.f) rewards[i] += total;
    }
    for (int i = 0; i < A; i++) episode_rewards[i] += rewards[i];
    if (max_steps > 0 && step >= (uint32_t)max_steps) {
      if (P[MGX_H_EPISODE_TRUNCATES]) std::fill(truncations.begin(), truncations.end(), 1);
      else std::fill(terminals.begin(), terminals.end(), 1);
    }
  }
};

}  // namespace

extern "C" {
void* mgxo_create(const int32_t* prog, const uint16_t* class_map, uint32_t seed) {
  if (prog[MGX_H_MAGIC] != MGX_MAGIC || prog[MGX_H_VERSION] != MGX_VERSION) return nullptr;
  Engine* e = new Engine();
  e->create(prog, class_map, seed);
  return e;
}
void mgxo_destroy(void* h) { delete (Engine*)h; }
void mgxo_reinit_buffers(void* h) { ((Engine*)h)->init_buffers(); }  // MettaGrid::set_buffers -> _init_buffers
void mgxo_step(void* h, const int32_t* actions, const int32_t* vibe_actions) { ((Engine*)h)->do_step(actions, vibe_actions); }
const uint8_t* mgxo_obs(void* h) { return ((Engine*)h)->obs.data(); }
const float* mgxo_rewards(void* h) { return ((Engine*)h)->rewards.data(); }
const float* mgxo_episode_rewards(void* h) { return ((Engine*)h)->episode_rewards.data(); }
const uint8_t* mgxo_terminals(void* h) { return ((Engine*)h)->terminals.data(); }
const uint8_t* mgxo_truncations(void* h) { return ((Engine*)h)->truncations.data(); }
const uint8_t* mgxo_action_success(void* h) { return ((Engine*)h)->action_success.data(); }
int mgxo_error(void* h) { return ((Engine*)h)->error; }
uint32_t mgxo_current_step(void* h) { return ((Engine*)h)->step; }
int mgxo_num_objects(void* h) { return (int)((Engine*)h)->objs.size(); }
// Per object record, int32 x (8 + 2*MGX_MAX_RESOURCES + MGX_TAG_WORDS): id, class, r, c, vibe, alive, agent_id,
// norder, order[13], amounts-by-item[13], tag words[8]
void mgxo_objects(void* h, int32_t* out) {
  Engine* e = (Engine*)h;
  const int RW = 8 + 2 * MGX_MAX_RESOURCES + MGX_TAG_WORDS;
  for (size_t i = 0; i < e->objs.size(); i++) {
    const Obj& o = e->objs[i];
    int32_t* w = out + i * RW;
    w[0] = (int)i + 1; w[1] = o.cls; w[2] = o.r; w[3] = o.c; w[4] = o.vibe; w[5] = o.alive; w[6] = o.agent; w[7] = o.norder;
    for (int k = 0; k < MGX_MAX_RESOURCES; k++) { w[8 + k] = k < o.norder ? o.order[k] : -1; w[8 + MGX_MAX_RESOURCES + k] = o.inv[k]; }
    for (int k = 0; k < MGX_TAG_WORDS; k++) w[8 + 2 * MGX_MAX_RESOURCES + k] = (int32_t)o.tags[k];
  }
}
void mgxo_stats(void* h, float* game_v, uint8_t* game_t, float* agent_v, uint8_t* agent_t) {
  Engine* e = (Engine*)h;
  int ng = e->P[MGX_H_NUM_GAME_STATS], na = e->P[MGX_H_NUM_AGENT_STATS];
  std::memcpy(game_v, e->game.v.data(), ng * 4);
  std::memcpy(game_t, e->game.touched.data(), ng);
  for (int i = 0; i < e->A; i++) {
    std::memcpy(agent_v + (size_t)i * na, e->agents[i].stats.v.data(), na * 4);
    std::memcpy(agent_t + (size_t)i * na, e->agents[i].stats.touched.data(), na);
  }
}
void mgxo_invalid_index_extra(void* h, int32_t* k_out, float* n_out) {
  Engine* e = (Engine*)h;
  for (int i = 0; i < e->A; i++)
    for (int q = 0; q < MGX_INVALID_EXTRA; q++) {
      k_out[i * MGX_INVALID_EXTRA + q] = e->agents[i].inv_k[q];
      n_out[i * MGX_INVALID_EXTRA + q] = e->agents[i].inv_n[q];
    }
}
void mgxo_reward_state(void* h, float* current_stat_reward) {  // systems/reward.hpp:36-42 current_reward()
  Engine* e = (Engine*)h;
  for (int i = 0; i < e->A; i++) {
    float t = 0.f;
    for (float p : e->agents[i].reward_prev) t += p;
    current_stat_reward[i] = t;
  }
}
// MettaGrid::set_inventory -> Agent::set_inventory (cpp/bindings/mettagrid_py.cpp:203-207, objects/agent.cpp:86-104):
// every held item is removed through Inventory::update and its "<res>.amount" stat set to 0, then the given items are set
// one by one; `items` / `amounts` come in the iteration order of the unordered_map the reference receives.
void mgxo_set_inventory(void* h, int agent, const int32_t* items, const int32_t* amounts, int n) {
  Engine* e = (Engine*)h;
  if (agent < 0 || agent >= e->A) return;
  const int oi = e->agents[agent].obj;
  uint8_t held[MGX_MAX_RESOURCES];
  const int nh = e->objs[oi].norder;
  std::memcpy(held, e->objs[oi].order, nh);
  for (int i = 0; i < nh; i++) {
    e->inv_update(oi, held[i], -(int)e->objs[oi].inv[held[i]]);
    e->agents[agent].stats.set(e->wk(MGX_S_RES_AMOUNT_BASE) + held[i], 0.f);
  }
  for (int i = 0; i < n; i++) e->inv_update(oi, items[i], amounts[i] - (int)e->objs[oi].inv[items[i]]);
}
// Self-check of the restated RNG against this toolchain's libstdc++ (the reference's actual dependency).
int mgxo_selftest_shuffle(uint32_t seed, int n, int rounds) {
  MT mine; mine.seed(seed);
  std::mt19937 ref(seed);
  std::vector<size_t> a(n), b(n);
  for (int r = 0; r < rounds; r++) {
    for (int i = 0; i < n; i++) a[i] = b[i] = i;
    mine.shuffle(a.data(), (uint32_t)n);
    std::shuffle(b.begin(), b.end(), ref);
    if (a != b) return r + 1;
  }
  return 0;
}
}
