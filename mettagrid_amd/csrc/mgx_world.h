// mgx_world.h — world-update kernel: one env per wavefront lane.
//
// Covers phases 1-11 of MettaGrid::_step (/root/reference/cpp/bindings/mettagrid_c.cpp:921-1056): snapshot of
// previous locations, step counter, the per-step agent shuffle, action dispatch by priority and stream
// (noop / move with the handler-chain line scan / change_vibe), timestep events, per-agent on_tick handlers, fixed and
// mobile AoE, territory effects, the game-level on_tick handler and coverage tracking.
// The order of agents inside one env is a true sequential dependence (collisions, first-come resources), so an env
// is executed serially by ONE lane; 32 (lean) or 64 (extended) envs advance in lock-step per wavefront and every lane
// runs the same compiled program, which bounds divergence.  Observations, rewards and termination are mgx_obs.h.
//
// Two instantiations exist: X = false ("lean", rungs 1-3: static tags, no queries/events/AoE/territory — the
// benchmarked configuration; iterative handler VM run_handler, one flat kernel, built in mgx_world_fast.hip) and
// X = true ("extended", everything; recursive handler templates, out-of-line functions, built in mgx_engine.hip).
// The X-only code is compiled out of the lean variant.
#ifndef MGX_WORLD_H_
#define MGX_WORLD_H_

#include <type_traits>

#include "mgx_device.h"

// Each translation unit instantiates these templates under its own set of build switches (MGX_WORLD_IDS,
// MGX_BIG, MGX_OUTLINE ...): on the GPU every unit is its own code object, but the single host binary of the
// sanitizer build (tests/cpu_emu) would merge the differing inline definitions.  One namespace per unit keeps them apart.
#ifndef MGX_TU_NS
#define MGX_TU_NS mgx_tu_engine
#endif
extern __shared__ __align__(16) uint8_t mgx_dyn_lds[];  // dynamic LDS of the world kernels
namespace MGX_TU_NS {

struct MgxCtx {  // handler/handler_context.hpp:38-112 (the fields the supported filters/mutations read)
  int actor, target, source;  // object slots, MGX_SLOT_NONE (-1) or MGX_SLOT_PROXY (-2)
  int proxy_tag;              // tag carried by the territory proxy cell
  int target_r, target_c;
  int move_direction;
  bool mutation_failed;
  bool skip_trigger;          // skip_on_update_trigger
  bool deferred;              // AOETracker::apply_fixed: ResourceDelta on the target is accumulated
};
__device__ __forceinline__ MgxCtx mgx_ctx(int actor, int target) {
  MgxCtx c;
  c.actor = actor; c.target = target; c.source = MGX_SLOT_NONE; c.proxy_tag = -1;
  c.target_r = c.target_c = 0; c.move_direction = 0;
  c.mutation_failed = false; c.skip_trigger = false; c.deferred = false;
  return c;
}

// Program pointer types: the world kernel keeps the whole program in LDS when it fits (address space 3 pointer ->
// ds_read, no vector-memory traffic for table lookups); everything else reads it from global memory.
typedef const int32_t* MgxGlobalProg;
typedef const __attribute__((address_space(3))) int32_t* MgxLdsProg;

// Per-lane LDS scratch of the extended world kernel ([k][lane] layouts, bank-conflict free per wavefront).
struct MgxXLds {
  int* def_delta;            // [28][stride] deferred target resource deltas (core/aoe_tracker.cpp:283-289)
  long long* terr_score;     // [8][stride]  per-prefix-tag influence sums (core/territory_tracker.cpp:223-236)
  int lane, stride;
};

// Own-agent state of the world kernel, staged in LDS ([agent][lane]) at kernel start: slot, position, prev_location,
// steps_without_motion and both action streams.  It is a read cache with write-through: every update also goes to HBM.
struct MgxALds {
  uint16_t* slot;
  uint16_t* rc;
  uint16_t* prev;
  uint32_t* swm;
  int16_t* act;   // [2][A][64] action ids, saturated to int16 (see mgx_sat16)
  uint16_t* cls;  // [A][64] class of the agent's object (fixed for the episode)
  int lane, A;
};
// MGX_WORLD_TIMING (debug builds only): per-phase shader-clock totals of lane 0 of every wavefront.
#ifdef MGX_WORLD_TIMING
#ifndef MGX_DBG_LINKAGE
#define MGX_DBG_LINKAGE static
#endif
MGX_DBG_LINKAGE __device__ unsigned long long mgx_dbg_cycles[16];
#define MGX_TICK(k) do { unsigned long long _n = clock64(); if (threadIdx.x == 0) atomicAdd(&mgx_dbg_cycles[k], _n - _t); _t = _n; } while (0)
#define MGX_TICK0() unsigned long long _t = clock64()
#else
#define MGX_TICK(k)
#define MGX_TICK0()
#endif

// MGX_BIG: inlining policy of the large interpreter functions.  The lean world kernel is built in its own translation
// unit with MGX_BIG=__forceinline__: its handler VM is iterative, so everything collapses into one flat kernel.  The
// extended variant keeps real calls (its four template levels have three nested call sites each; fully inlined the
// kernel is ~1 M instructions).
#ifndef MGX_BIG
#define MGX_BIG
#endif
// MGX_WORLD_LPW: envs (active lanes) per wavefront of the world kernel.  A workgroup always owns 64 envs; with LPW < 64
// they are spread over 64 / LPW wavefronts that use their first LPW lanes.  Fewer lanes per wavefront = fewer distinct
// handler paths serialised inside one wavefront, more wavefronts per SIMD to overlap their memory latency.
#ifndef MGX_WORLD_LPW
#define MGX_WORLD_LPW 64
#endif
// MGX_WORLD_EPG: envs per workgroup = stride of the [k][env] LDS arrays.  The LDS a workgroup needs grows with the
// agents per env (17 B per agent and env, + 320 B per env in the extended variant); fewer envs per workgroup let more
// workgroups share a CU's 160 KB when there are many agents.
#ifndef MGX_WORLD_EPG
#define MGX_WORLD_EPG 64
#endif
#define MGX_WORLD_THREADS (MGX_WAVE * (MGX_WORLD_EPG / MGX_WORLD_LPW))
// MGX_WORLD_HELPERS (the lean lane-per-env kernel at 32 envs per wavefront): the upper 32 lanes of a wavefront, which used
// to leave at once, stay as HELPERS of the lower 32 — helper lane h shares env, LDS cells and everything with lane h - 32 and
// takes the second half of the agents in the three batched per-agent passes (staging, deferred bookkeeping, coverage), whose
// time is the number of chunks of eight agents a lane walks; during the serial action dispatch it is masked off and costs
// nothing.  Same instruction stream, half the trips of those passes.
#if defined(MGX_WORLD_HELPERS) && !defined(MGX_CPU_EMU) && !defined(MGX_ACT_TU) && MGX_WORLD_LPW * 2 == MGX_WAVE
#define MGX_HELPERS_ON 1
#else
#define MGX_HELPERS_ON 0
#endif
#ifdef MGX_ACT_TU
// Agent-parallel action kernel (mgx_act.h): a workgroup owns MGX_WORLD_EPG envs, each env a run of A' = blockDim / EPG
// consecutive lanes (A' = agents per env rounded up to a power of two, <= 64: an env never straddles a wavefront).
__device__ __forceinline__ int mgx_act_shift() { return (31 - __clz((int)blockDim.x)) - (31 - __clz(MGX_WORLD_EPG)); }
__device__ __forceinline__ int mgx_world_lane() { return (int)(threadIdx.x >> mgx_act_shift()); }
#else
__device__ __forceinline__ int mgx_world_lane() {  // index of this lane's env inside the workgroup's 64-env group
#if MGX_HELPERS_ON
  return (int)((threadIdx.x >> 6) * MGX_WORLD_LPW + (threadIdx.x & (MGX_WORLD_LPW - 1)));   // (a helper lane: its partner's)
#else
  return (int)((threadIdx.x >> 6) * MGX_WORLD_LPW + (threadIdx.x & (MGX_WAVE - 1)));
#endif
}
#endif

#define MGX_ACT_GSET 4
// Extended handler VM: frames and handler contexts per lane (see MgxEnvT::vm_run).
#define MGX_VM_FRAMES 6
#define MGX_VM_CTXS 6
#define MGX_VM_WORDS (MGX_VM_FRAMES * 4 + MGX_VM_CTXS * 2)
// MGX_OUTLINE: the few large functions of the extended variant that stay real calls (one instance each, no recursion).
#ifndef MGX_OUTLINE
#define MGX_OUTLINE
#endif
#ifdef MGX_NO_CLS_STAGE  // extended kernel: the agents' classes are not staged (only the serial on_tick loop reads them)
__host__ __device__ inline int mgx_world_alds_bytes(int A) { return A * MGX_WORLD_EPG * (2 + 2 + 2 + 4 + 4); }
#else
__host__ __device__ inline int mgx_world_alds_bytes(int A) { return A * MGX_WORLD_EPG * (2 + 2 + 2 + 2 + 4 + 4); }
#endif
// byte offset of the extended variant's per-lane scratch (deferred deltas | territory scores | VM words) in the LDS
__host__ __device__ inline int mgx_world_xlds_off(int A) { return ((A * MGX_WORLD_EPG + 15) & ~15) + ((mgx_world_alds_bytes(A) + 15) & ~15); }
// Dynamic LDS of the world kernels: order u8[A][64] | swm u32[A][64] | act i16[2][A][64] | slot, rc, prev, cls u16[A][64] |
// [X: VM u32[MGX_VM_WORDS][64] | (in-kernel AoE phase only:) deferred i32[28][64] | territory i64[8][64]] | program
// i32[prog_words] (when it fits).
// (aoe_lds: the deferred-delta and territory-score scratch of the in-kernel AoE phase; games whose AoE runs in
// mgx_aoe_kernel do without it, which is what lets a fourth 32-env workgroup fit a CU at 64 agents per env)
// (act_extra: the lane-per-agent kernels' footprint table + cell map, MgxDev::act_lds_extra)
__host__ __device__ inline int mgx_world_lds_fixed(int A, bool X, bool aoe_lds = true, int act_extra = 0) {
  int o = ((A * MGX_WORLD_EPG + 15) & ~15) + ((mgx_world_alds_bytes(A) + 15) & ~15);
  if (X) o += MGX_VM_WORDS * MGX_WORLD_EPG * 4 + (aoe_lds ? 28 * MGX_WORLD_EPG * 4 + 8 * MGX_WORLD_EPG * 8 : 0);
#if defined(MGX_ACT_TU) || MGX_HELPERS_ON
  o += MGX_ACT_GSET * MGX_WORLD_EPG * 8;  // (order position + 1) << 32 | f32 bits of the last game-stat set, per env
  o += act_extra;
#endif
  return o;
}
__device__ __forceinline__ MgxALds mgx_world_alds(uint8_t* lds, int A, int lane) {
  const int off = (A * MGX_WORLD_EPG + 15) & ~15;
  MgxALds al;
  al.lane = lane; al.A = A;
  al.swm = (uint32_t*)(lds + off);
  al.act = (int16_t*)(lds + off + A * MGX_WORLD_EPG * 4);
  al.slot = (uint16_t*)(lds + off + A * MGX_WORLD_EPG * 8);
  al.rc = al.slot + A * MGX_WORLD_EPG;
  al.prev = al.rc + A * MGX_WORLD_EPG;
#ifdef MGX_NO_CLS_STAGE
  al.cls = nullptr;
#else
  al.cls = al.prev + A * MGX_WORLD_EPG;
#endif
  return al;
}

__device__ __forceinline__ unsigned long long mgx_floor_sqrt(unsigned long long v) {  // == floor_sqrt_u64 (:17-33), v < 2^53
  unsigned long long r = (unsigned long long)__dsqrt_rn((double)v);
  while (r * r > v) r--;
  while ((r + 1) * (r + 1) <= v) r++;
  return r;
}

// Evaluation stack of the game-value postfix code: eight f32 held in registers, top of stack in s0.  The compiler
// bounds expression depth at 8 (compile_spec), like the array this replaces.
struct MgxValueStack {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
  int n = 0;
  __device__ __forceinline__ void push(float v) { s7 = s6; s6 = s5; s5 = s4; s4 = s3; s3 = s2; s2 = s1; s1 = s0; s0 = v; n++; }
  __device__ __forceinline__ float pop() { float v = s0; s0 = s1; s1 = s2; s2 = s3; s3 = s4; s4 = s5; s5 = s6; s6 = s7; n--; return v; }
};

template <class Env> struct MgxGenR3;   // generated at build(): mgx_handlers_gen.h
template <class Env> struct MgxGenR4;
template <class Env> struct MgxGenJ;    // generated at run time for the program at hand (mettagrid_amd/jit.py): MGX_GEN_HEADER
template <class PP, bool X>
struct MgxEnvT {  // per-lane view of one env
  typedef PP ProgPtr;
#ifdef MGX_CONST_DEV
  static constexpr const MgxDev& d = g_mgx_dev;
#else
  const MgxDev& d;  // the launching kernel's own by-value argument
#endif
  PP P_;
  int env_;
  uint32_t step;
  MgxXLds xl;
  MgxALds al_;
  mutable int cur_agent, cur_slot;  // agent whose action is being executed (LDS write-through of its position)
  mutable int grid_dirty;           // a grid cell was written since do_move last looked at the move target
#if defined(MGX_ACT_TU) || MGX_HELPERS_ON
  int act_pos = 0;  // this lane's place in the order the reference walks the agents in (shuffled order / agent index)
  bool duo_on = false;   // (MGX_HELPERS_ON) two lanes of this env dispatch agents side by side: shared per-env words need care
  mutable uint32_t act_seq = 0;   // game-stat SETs this lane has made in this tick (the later of two sets by one agent wins)
  // StatsMutation on a game-scope stat from concurrently running agents: the set of the LAST agent in order must win.
  // Each set goes to a per-env LDS cell as (position + 1) << 32 | value bits under a 64-bit max; mgx_act_body applies the
  // surviving value when the phase is over.  (The host only selects this kernel when nothing reads those stats mid-phase.)
  __device__ __forceinline__ unsigned long long* act_gset_cells() const {
    return (unsigned long long*)(mgx_dyn_lds + mgx_world_lds_fixed(d.A, X, d.x_aoe_lds != 0) - MGX_ACT_GSET * MGX_WORLD_EPG * 8);  // (in front of act_lds_extra)
  }
  __device__ __forceinline__ void act_gstat_set(int id, float v) const {
    int k = 0;
    while (k < d.act_ngset - 1 && d.act_gset_ids[k] != id) k++;
    // key: order position, then this lane's set count — game_stats->set is last-write-wins (stats_tracker.hpp), so of two
    // sets by the same agent in one chain (on_use, then on_after_use) the later one must survive, not the larger bit pattern
    const uint32_t seq = act_seq < 255u ? act_seq++ : 255u;
    const unsigned long long w = ((unsigned long long)(act_pos + 1) << 40) | ((unsigned long long)seq << 32) | (unsigned long long)__float_as_uint(v);
    atomicMax(act_gset_cells() + k * MGX_WORLD_EPG + mgx_world_lane(), w);
  }
#endif
#ifdef MGX_CONST_DEV
  __device__ MgxEnvT(const MgxDev&, PP prog, int e) : P_(prog), env_(e), step(0), cur_agent(-1), cur_slot(-1), grid_dirty(1) {
#else
  __device__ MgxEnvT(const MgxDev& dd, PP prog, int e) : d(dd), P_(prog), env_(e), step(0), cur_agent(-1), cur_slot(-1), grid_dirty(1) {
#endif
    al_.slot = nullptr; al_.rc = nullptr; al_.prev = nullptr; al_.swm = nullptr; al_.act = nullptr; al_.cls = nullptr; al_.lane = 0; al_.A = 0;
    xl.def_delta = nullptr; xl.terr_score = nullptr; xl.lane = 0; xl.stride = MGX_WAVE; }
  // MGX_WORLD_IDS (the lane-per-env world kernel's own translation unit): env index, program base and the LDS agent
  // staging are recomputed from the work-item ids and constant memory wherever they are needed.  As members they
  // live in the kernel's stack frame, and an out-of-line handler function has to re-load them through `this` after
  // every store that might alias it.
#ifdef MGX_WORLD_IDS
  __device__ __forceinline__ int envi() const { return (int)(blockIdx.x * MGX_WORLD_EPG) + mgx_world_lane(); }
  __device__ __forceinline__ PP prog() const {
#ifdef MGX_ACT_TU
    if constexpr (std::is_same<PP, MgxLdsProg>::value) return (MgxLdsProg)(int32_t*)(mgx_dyn_lds + mgx_world_lds_fixed(d.A, X, d.x_aoe_lds != 0, d.act_lds_extra));
#else
    if constexpr (std::is_same<PP, MgxLdsProg>::value) return (MgxLdsProg)(int32_t*)(mgx_dyn_lds + mgx_world_lds_fixed(d.A, X, d.x_aoe_lds != 0));
#endif
    else return d.P;
  }
  __device__ __forceinline__ MgxALds AL() const { return mgx_world_alds(mgx_dyn_lds, d.A, mgx_world_lane()); }
  __device__ __forceinline__ MgxXLds XL() const {  // per-lane scratch of the extended variant
    MgxXLds x;
    x.def_delta = (int*)(mgx_dyn_lds + mgx_world_xlds_off(d.A) + MGX_VM_WORDS * MGX_WORLD_EPG * 4);
    x.terr_score = (long long*)(mgx_dyn_lds + mgx_world_xlds_off(d.A) + MGX_VM_WORDS * MGX_WORLD_EPG * 4 + 28 * MGX_WORLD_EPG * 4);
    x.lane = mgx_world_lane(); x.stride = MGX_WORLD_EPG;
    return x;
  }
#else
  __device__ __forceinline__ const MgxXLds& XL() const { return xl; }
  __device__ __forceinline__ int envi() const { return env_; }
  __device__ __forceinline__ PP prog() const { return P_; }
  __device__ __forceinline__ const MgxALds& AL() const { return al_; }
#endif
  // Class records: with MGX_HOT_PROG (extended world kernel) the LDS program copy holds only the sections the handler VM
  // walks (limits .. territory controls); the class table (one record per agent: 15 KB at 64 agents) and the tag-list
  // index stay in HBM / L2 and are read through the global pointer.
#ifdef MGX_HOT_PROG
  typedef MgxGlobalProg CP;
  __device__ __forceinline__ CP cls(int c) const { return d.P + d.sec[MGX_SEC_CLASSES] + c * MGX_C_WORDS; }
#else
  typedef PP CP;
  __device__ __forceinline__ CP cls(int c) const { return prog() + d.sec[MGX_SEC_CLASSES] + c * MGX_C_WORDS; }
#endif

  __device__ __forceinline__ size_t so(int slot) const { return (size_t)envi() * d.S + slot; }
  __device__ __forceinline__ size_t ao(int agent) const { return (size_t)envi() * d.A + agent; }
  __device__ __forceinline__ uint16_t& inv(int slot, int item) const { return d.obj_inv[so(slot) * MGX_INV_PITCH + item]; }
  __device__ __forceinline__ int inv_of(int slot, int item) const { return slot >= 0 ? (int)inv(slot, item) : 0; }
  __device__ __forceinline__ CP cls_of(int slot) const { return cls(d.obj_cls[so(slot)]); }
  __device__ __forceinline__ int agent_of(int slot) const {
    if (slot < 0) return -1;
    int a = d.obj_agent[so(slot)];
    return a == MGX_NO_AGENT ? -1 : a;
  }
#if MGX_HELPERS_ON
  __device__ __forceinline__ void flag(uint32_t bit) const { atomicOr(&d.err[envi()], bit); }   // (two lanes of an env may flag at once)
#else
  __device__ __forceinline__ void flag(uint32_t bit) const { d.err[envi()] |= bit; }
#endif

  // ---- stats (systems/stats_tracker.hpp:69-90) ----
  __device__ __forceinline__ void astat_touch(int agent, int id) const { d.ag_touched[ao(agent) * d.NSW + (id >> 5)] |= 1u << (id & 31); }
  // add(): every caller adds a strictly positive amount to a key that only ever grows, so "value != 0" already
  // says the key exists; the host ORs that into the touched flags (mgx_get_stats) and the bit RMW is skipped here.
  // Plain read-modify-write on purpose: float atomics execute at the memory side on gfx950 (nothing stays in L2) and
  // 64 lanes adding into 64 different rows is their slowest shape (MI355X_MICROARCH.md "Global float atomics").
  __device__ __forceinline__ void astat_add(int agent, int id, float v) const {
    if (id < 0) return;
    d.ag_stats[ao(agent) * d.NSP + id] += v;
  }
  __device__ __forceinline__ void astat_add_touch(int agent, int id, float v) const {  // for deltas that may be zero or negative
    if (id < 0) return;
    d.ag_stats[ao(agent) * d.NSP + id] += v;
    astat_touch(agent, id);
  }
  __device__ __forceinline__ void astat_set(int agent, int id, float v) const {
    if (id < 0) return;
    d.ag_stats[ao(agent) * d.NSP + id] = v;
    astat_touch(agent, id);
  }
  __device__ __forceinline__ float astat_get(int agent, int id) const {
    return id < 0 ? 0.f : d.ag_stats[ao(agent) * d.NSP + id];
  }
  __device__ __forceinline__ void gstat_touch(int id) const { d.game_touched[(size_t)envi() * d.NGW + (id >> 5)] |= 1u << (id & 31); }
  __device__ __forceinline__ void gstat_set(int id, float v) const {
    if (id < 0) return;
    d.game_stats[(size_t)envi() * d.NG + id] = v;
    gstat_touch(id);
  }
  __device__ __forceinline__ void gstat_add(int id, float v) const {
    if (id < 0) return;
    d.game_stats[(size_t)envi() * d.NG + id] += v;
    gstat_touch(id);
  }

  // ---- inventory (cpp/src/mettagrid/objects/inventory.cpp) ----
  // One object's whole inventory row in registers: two independent 16-byte loads instead of one dependent load per
  // resource a limit group or modifier list touches.
  struct InvRow {
    uint32_t w[MGX_INV_PITCH / 2];
    __device__ __forceinline__ uint32_t pair(int i) const {  // w[i] for a run-time i without indexing the array
      uint32_t a = (i & 1) ? w[1] : w[0], b = (i & 1) ? w[3] : w[2], c = (i & 1) ? w[5] : w[4], e = (i & 1) ? w[7] : w[6];
      uint32_t ab = (i & 2) ? b : a, ce = (i & 2) ? e : c;
      return (i & 4) ? ce : ab;
    }
    __device__ __forceinline__ int get(int item) const { uint32_t v = pair(item >> 1); return (int)((item & 1) ? v >> 16 : v & 0xFFFFu); }
    template <int K> __device__ __forceinline__ int at() const { return (int)((K & 1) ? w[K >> 1] >> 16 : w[K >> 1] & 0xFFFFu); }
    __device__ __forceinline__ void set(int item, int v) {  // run-time item, again without indexing the array
      const int i = item >> 1;
      const uint32_t old = pair(i);
      const uint32_t nw = (item & 1) ? ((old & 0xFFFFu) | ((uint32_t)v << 16)) : ((old & 0xFFFF0000u) | ((uint32_t)v & 0xFFFFu));
#pragma unroll
      for (int q = 0; q < MGX_INV_PITCH / 2; q++) w[q] = (i == q) ? nw : w[q];
    }
  };
  __device__ __forceinline__ InvRow inv_row(int slot) const {
    const uint4* p = (const uint4*)(d.obj_inv + so(slot) * MGX_INV_PITCH);
    const uint4 lo = p[0], hi = p[1];
    InvRow r;
    r.w[0] = lo.x; r.w[1] = lo.y; r.w[2] = lo.z; r.w[3] = lo.w; r.w[4] = hi.x; r.w[5] = hi.y; r.w[6] = hi.z; r.w[7] = hi.w;
    return r;
  }
  __device__ __forceinline__ int effective_limit(const InvRow& row, PP L) const {  // objects/inventory.hpp:26-40
    int sum = 0;
    PP mods = prog() + d.sec[MGX_SEC_MODS] + L[MGX_L_MOD_START] * MGX_MOD_WORDS;
    for (int i = 0; i < L[MGX_L_MOD_COUNT]; i++)
      sum += row.get(mods[i * MGX_MOD_WORDS + MGX_MOD_ITEM]) * mods[i * MGX_MOD_WORDS + MGX_MOD_BONUS];
    int eff = min(L[MGX_L_MAX], max(L[MGX_L_MIN], sum));
    return min(max(eff, 0), 65535);
  }
  __device__ __forceinline__ int group_amount(const InvRow& row, PP L) const {
    const uint32_t mask = (uint32_t)L[MGX_L_RES_MASK];
    int s = 0;
    s += (mask >> 0) & 1 ? row.template at<0>() : 0;   s += (mask >> 1) & 1 ? row.template at<1>() : 0;
    s += (mask >> 2) & 1 ? row.template at<2>() : 0;   s += (mask >> 3) & 1 ? row.template at<3>() : 0;
    s += (mask >> 4) & 1 ? row.template at<4>() : 0;   s += (mask >> 5) & 1 ? row.template at<5>() : 0;
    s += (mask >> 6) & 1 ? row.template at<6>() : 0;   s += (mask >> 7) & 1 ? row.template at<7>() : 0;
    s += (mask >> 8) & 1 ? row.template at<8>() : 0;   s += (mask >> 9) & 1 ? row.template at<9>() : 0;
    s += (mask >> 10) & 1 ? row.template at<10>() : 0; s += (mask >> 11) & 1 ? row.template at<11>() : 0;
    s += (mask >> 12) & 1 ? row.template at<12>() : 0;
    return s;
  }
  __device__ __forceinline__ int effective_limit(int slot, PP L) const { return effective_limit(inv_row(slot), L); }
  __device__ __forceinline__ int group_amount(int slot, PP L) const { return group_amount(inv_row(slot), L); }
  __device__ __forceinline__ PP limit_of(CP C, int item) const {
    int li = C[MGX_C_RES_LIMIT + item];
    return li < 0 ? (PP) nullptr : prog() + d.sec[MGX_SEC_LIMITS] + li * MGX_L_WORDS;
  }
  // Agent::on_inventory_change (objects/agent.cpp:106-121); `a` = agent index of the object or -1 (the caller has it
  // from the same round trip as the inventory row).  The three cells it updates are loaded together.
  __device__ MGX_BIG void on_inventory_change(int a, int item, int delta, int amount) const {
    if (a < 0 || delta == 0) return;
    const size_t sb = ao(a) * d.NSP;
    const int s_flow = (delta > 0 ? d.wk[MGX_S_RES_GAINED_BASE] : d.wk[MGX_S_RES_LOST_BASE]) + item;
    const int s_amt = d.wk[MGX_S_RES_AMOUNT_BASE] + item;
    const bool died = amount == 0 && delta < 0 && item == d.hp_res;
    const int s_death = d.wk[MGX_S_DEATH];
    const float flow = d.ag_stats[sb + s_flow];
    uint32_t* tw = &d.ag_touched[ao(a) * d.NSW + (s_amt >> 5)];
    const uint32_t tword = *tw;
    const float deaths = d.ag_stats[sb + max(s_death, 0)];
    d.ag_stats[sb + s_flow] = __fadd_rn(flow, (float)(delta > 0 ? delta : -delta));  // add(): value != 0 marks the key
    d.ag_stats[sb + s_amt] = (float)amount;                                          // set(): key exists from now on
    *tw = tword | (1u << (s_amt & 31));
    if (died && s_death >= 0) d.ag_stats[sb + s_death] = __fadd_rn(deaths, 1.f);
  }

  // Inventory::update (inventory.cpp:38-86).  DEPTH bounds the update -> enforce_all_limits -> update recursion.
  template <int DEPTH>
  __device__ MGX_BIG int inv_update(int slot, int item, int delta, bool ignore_limits = false, bool notify = true) const {
    // class, inventory row and iteration order of the object: independent loads, one round trip
    const uint16_t cls_id = d.obj_cls[so(slot)];
    const InvRow row = inv_row(slot);
    unsigned long long ord = d.obj_order[so(slot)];
    const uint8_t ag = d.obj_agent[so(slot)];
    CP C = cls(cls_id);
    int initial = row.get(item);
    int new_amount = initial + delta;
    int mx = 65535;
    if (!ignore_limits) {
      PP L = limit_of(C, item);
      if (L) {
        int used = group_amount(row, L) - initial;
        if (used < 0) used = 0;
        int m = effective_limit(row, L) - used;
        mx = m < 0 ? 0 : m;
      }
    }
    int clamped = min(max(new_amount, 0), mx);
    if (clamped != initial) {
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
    if (notify && dl != 0) on_inventory_change(ag == MGX_NO_AGENT ? -1 : (int)ag, item, dl, clamped);
    if (dl < 0 && (C[MGX_C_MODIFIER_MASK] & (1 << item))) {
      if constexpr (DEPTH > 0) enforce_all_limits<DEPTH - 1>(slot, C);
      else flag(4u);
    }
    return dl;
  }
  template <int DEPTH>
  __device__ MGX_BIG void enforce_all_limits(int slot, CP C) const {  // inventory.cpp:141-173
    for (int li = 0; li < C[MGX_C_LIMIT_COUNT]; li++) {
      PP L = prog() + d.sec[MGX_SEC_LIMITS] + (C[MGX_C_LIMIT_START] + li) * MGX_L_WORDS;
      if (L[MGX_L_DROP_COUNT] == 0) continue;
      int excess;
      { const InvRow row = inv_row(slot); excess = group_amount(row, L) - effective_limit(row, L); }
      if (excess <= 0) continue;
      PP drop = prog() + d.sec[MGX_SEC_DROP_ORDER] + L[MGX_L_DROP_START];
      for (int k = 0; k < L[MGX_L_DROP_COUNT]; k++) {
        int item = drop[k];
        int to_drop = min((int)inv(slot, item), excess);
        if (to_drop > 0) {
          inv_update<DEPTH>(slot, item, -to_drop);
          const InvRow row = inv_row(slot);
          excess = group_amount(row, L) - effective_limit(row, L);
        }
        if (excess <= 0) break;
      }
    }
  }
  __device__ MGX_BIG int free_space(int slot, int item) const {  // inventory.cpp:97-110
    const uint16_t cls_id = d.obj_cls[so(slot)];
    const InvRow row = inv_row(slot);
    PP L = limit_of(cls(cls_id), item);
    if (!L) return 65535 - row.get(item);
    int used = group_amount(row, L), eff = effective_limit(row, L);
    return eff > used ? eff - used : 0;
  }
  __device__ MGX_BIG int transfer(int src, int dst, int item, int delta) const {  // objects/has_inventory.cpp:76-108
    if (delta <= 0) return 0;
    int give = min((int)inv(src, item), delta);
    int amount = min(give, free_space(dst, item));
    inv_update<1>(src, item, -amount);
    inv_update<1>(dst, item, amount);
    return amount;
  }

  // ---- tags ----
  __device__ __forceinline__ uint32_t tagword(int slot, int w, const MgxCtx& c) const {
    if (slot >= 0) {
      if constexpr (X) { if (d.obj_tags) return d.obj_tags[so(slot) * MGX_TAG_WORDS + w]; }
      return (uint32_t)cls_of(slot)[MGX_C_TAGS + w];
    }
    if (slot == MGX_SLOT_PROXY && c.proxy_tag >= 0 && (c.proxy_tag >> 5) == w) return 1u << (c.proxy_tag & 31);
    return 0u;
  }
  __device__ __forceinline__ bool has_tag(int slot, int t) const {
    if (slot < 0) return false;
    if constexpr (X) { if (d.obj_tags) return (d.obj_tags[so(slot) * MGX_TAG_WORDS + (t >> 5)] >> (t & 31)) & 1u; }
    return ((uint32_t)cls_of(slot)[MGX_C_TAGS + (t >> 5)] >> (t & 31)) & 1u;
  }
#ifdef MGX_HOT_PROG
  __device__ __forceinline__ int tag_list(int tag) const { return d.P[d.sec[MGX_SEC_TAG_LISTS] + tag]; }
#else
  __device__ __forceinline__ int tag_list(int tag) const { return prog()[d.sec[MGX_SEC_TAG_LISTS] + tag]; }
#endif
  __device__ __forceinline__ uint16_t* tl_items(int li) const { return d.tl_items + ((size_t)envi() * d.NL + li) * d.S; }
  __device__ __forceinline__ uint16_t& tl_count(int li) const { return d.tl_count[(size_t)envi() * d.NL + li]; }
  // GridObject::add_tag / remove_tag without the lifecycle handlers (core/grid_object.cpp:93-123): bitset + TagIndex.
  // Returns true when the tag set changed; the handler VM (vm_run) then fires on_tag_add / on_tag_remove.
  __device__ __forceinline__ bool tag_set(int o, int tag) const {
    if (o < 0 || tag < 0 || tag >= 256 || !d.obj_tags || has_tag(o, tag)) return false;
    d.obj_tags[so(o) * MGX_TAG_WORDS + (tag >> 5)] |= 1u << (tag & 31);
    int li = tag_list(tag);
    if (li >= 0) { uint16_t n = tl_count(li); if (n < d.S) { tl_items(li)[n] = (uint16_t)o; tl_count(li) = n + 1; } }
    territory_tags_changed(o);
    return true;
  }
  __device__ __forceinline__ bool tag_clear(int o, int tag) const {
    if (o < 0 || tag < 0 || tag >= 256 || !d.obj_tags || !has_tag(o, tag)) return false;
    d.obj_tags[so(o) * MGX_TAG_WORDS + (tag >> 5)] &= ~(1u << (tag & 31));
    int li = tag_list(tag);
    if (li >= 0) {  // stable erase (core/tag_index.cpp:37-43)
      uint16_t* it = tl_items(li);
      int n = tl_count(li), k = 0;
      for (int i = 0; i < n; i++) { uint16_t v = it[i]; if (v != (uint16_t)o) it[k++] = v; }
      tl_count(li) = (uint16_t)k;
    }
    territory_tags_changed(o);
    return true;
  }
  // first on_tag_add / on_tag_remove record of object o's class for `tag` at or after record j, or -1
  __device__ __forceinline__ int next_tag_handler(int o, int tag, int start_field, int j, int* handler) const {
    CP C = cls_of(o);
    PP th = prog() + d.sec[MGX_SEC_TAG_HANDLERS] + C[start_field] * MGX_TH_WORDS;
    const int n = C[start_field + 1];
    for (; j < n; j++)
      if (th[j * MGX_TH_WORDS + MGX_TH_TAG] == tag) { *handler = th[j * MGX_TH_WORDS + MGX_TH_HANDLER]; return j; }
    return -1;
  }

  // ---- query workspace ----
  __device__ __forceinline__ uint16_t* qbuf(int idx) const { return d.qws + ((size_t)envi() * d.QB + idx) * d.S; }
  __device__ __forceinline__ uint32_t* qvis(int depth) const { return d.qvis + ((size_t)envi() * (d.QD + 1) + depth) * d.SW; }
  enum { QB_EVENT = 0, QB_LOST = 1, QB_KEEP = 2, QB_BASE = 3 };  // then two buffers per nesting level

  // ---- queries (core/query_system.cpp:28-89, 178-330).  Results land in qbuf(QB_BASE + 2*depth); QD bounds the
  // query -> filter -> query template recursion, `depth` selects the workspace level at run time. ----
  template <int QD>
  __device__ bool matches(int pc, MgxCtx c, int obj, int depth) const {
    c.target = obj;
    return pc == MGX_PC_PASS || check_filters<QD>(pc, c, depth);
  }
  template <int QD>
  __device__ int apply_limits(uint16_t* res, int n, PP Q, const MgxCtx& c, int depth) const {  // :74-89
    if (Q[MGX_Q_ORDER] & 1) {  // std::shuffle with the envi()'s mt19937 (bits/stl_algo.h:3729-3795)
      if (n >= 2) {
        uint32_t i = 1;
        if ((n & 1) == 0) { uint32_t j = rng_below(2); uint16_t t = res[1]; res[1] = res[j]; res[j] = t; i = 2; }
        while (i < (uint32_t)n) {
          uint32_t s = i + 1, x = rng_below(s * (s + 1));
          uint32_t j0 = x / (s + 1), j1 = x % (s + 1);
          uint16_t t = res[i]; res[i] = res[j0]; res[j0] = t;
          t = res[i + 1]; res[i + 1] = res[j1]; res[j1] = t;
          i += 2;
        }
      }
    }
    int mx = Q[MGX_Q_MAX_ITEMS] >= 0 ? (int)eval_value<QD - 1>(Q[MGX_Q_MAX_ITEMS], c.actor, c, depth + 1) : -1;
    return (mx >= 0 && n > mx) ? mx : n;
  }
  template <int QD>
  __device__ MGX_OUTLINE int eval_query(int qi, const MgxCtx& c, int depth) const {
    if constexpr (!X || QD == 0) {
      flag(4u);
      return 0;
    } else {
      if (depth > d.QD) { flag(4u); return 0; }
      PP Q = prog() + d.sec[MGX_SEC_QUERIES] + qi * MGX_Q_WORDS;
      uint16_t* out = qbuf(QB_BASE + 2 * depth);
      uint16_t* tmp = qbuf(QB_BASE + 2 * depth + 1);
      int n = 0;
      const int kind = Q[MGX_Q_KIND], a0 = Q[MGX_Q_A0], a1 = Q[MGX_Q_A1], a2 = Q[MGX_Q_A2], a3 = Q[MGX_Q_A3];
      if (kind == MGX_QK_TAG) {
        int li = tag_list(a0);
        if (li >= 0) {
          const uint16_t* it = tl_items(li);
          int cnt = tl_count(li);
          for (int i = 0; i < cnt; i++) { int o = it[i]; if (matches<QD - 1>(a1, c, o, depth + 1)) out[n++] = (uint16_t)o; }
        }
      } else if (kind == MGX_QK_FILTERED) {
        int m = eval_query<QD - 1>(a0, c, depth + 1);
        const uint16_t* src = qbuf(QB_BASE + 2 * (depth + 1));
        for (int i = 0; i < m; i++) tmp[i] = src[i];
        for (int i = 0; i < m; i++) { int o = tmp[i]; if (matches<QD - 1>(a1, c, o, depth + 1)) out[n++] = (uint16_t)o; }
      } else if (kind == MGX_QK_CLOSURE) {
        uint32_t* vis = qvis(depth);
        for (int w = 0; w < d.SW; w++) vis[w] = 0;
        int nr = eval_query<QD - 1>(a0, c, depth + 1);
        const uint16_t* src = qbuf(QB_BASE + 2 * (depth + 1));
        if (a1 < 0) {
          for (int i = 0; i < nr; i++) out[i] = src[i];
          return apply_limits<QD>(out, nr, Q, c, depth);
        }
        for (int i = 0; i < nr; i++) {
          int o = src[i];
          if (!((vis[o >> 5] >> (o & 31)) & 1u)) { vis[o >> 5] |= 1u << (o & 31); out[n++] = (uint16_t)o; }
        }
        int np = eval_query<QD - 1>(a1, c, depth + 1);
        for (int i = 0; i < np; i++) tmp[i] = src[i];
        for (int head = 0; head < n; head++) {  // BFS: the result list is the queue
          int cur = out[head];
          for (int i = 0; i < np; i++) {
            int cand = tmp[i];
            if ((vis[cand >> 5] >> (cand & 31)) & 1u) continue;
            if (a2 != MGX_PC_PASS) {
              MgxCtx e = c;
              e.source = cur; e.target = cand;
              if (!check_filters<QD - 1>(a2, e, depth + 1)) continue;
            }
            vis[cand >> 5] |= 1u << (cand & 31);
            out[n++] = (uint16_t)cand;
          }
        }
        if (a3 != MGX_PC_PASS) {
          int k = 0;
          for (int i = 0; i < n; i++) { int o = out[i]; if (matches<QD - 1>(a3, c, o, depth + 1)) out[k++] = (uint16_t)o; }
          n = k;
        }
      } else if (kind == MGX_QK_RAYCAST) {
        uint32_t* seen = qvis(depth);
        for (int w = 0; w < d.SW; w++) seen[w] = 0;
        int ns = eval_query<QD - 1>(a0, c, depth + 1);
        const uint16_t* src = qbuf(QB_BASE + 2 * (depth + 1));
        for (int i = 0; i < ns; i++) tmp[i] = src[i];
        PP dirs = prog() + d.sec[MGX_SEC_WORDLIST] + a2;
        const int blocker_pc = Q[MGX_Q_A4];
        for (int i = 0; i < ns; i++) {
          int s = tmp[i];
          MgxCtx sc = c;
          sc.actor = sc.target = s;
          int range = (int)eval_value<QD - 1>(a1, s, sc, depth + 1);
          if (range <= 0) continue;
          uint16_t rc = d.obj_rc[so(s)];
          for (int k = 0; k < a3; k++)
            for (int dist = 1; dist <= range; dist++) {
              int r = (rc >> 8) + dirs[k * 2] * dist, cc = (rc & 0xFF) + dirs[k * 2 + 1] * dist;
              if (r < 0 || cc < 0 || r >= d.H || cc >= d.W) break;
              int o = (int)cell(r, cc) - 1;
              if (o < 0) continue;
              bool blocker = false;
              if (blocker_pc != MGX_PC_FAIL) { MgxCtx b = c; b.target = o; blocker = check_filters<QD - 1>(blocker_pc, b, depth + 1); }
              bool was = (seen[o >> 5] >> (o & 31)) & 1u;
              if (blocker) {
                if ((Q[MGX_Q_ORDER] & 0x100) && !was) { seen[o >> 5] |= 1u << (o & 31); out[n++] = (uint16_t)o; }
                break;
              }
              if (!was) { seen[o >> 5] |= 1u << (o & 31); out[n++] = (uint16_t)o; }
            }
        }
      }
      return apply_limits<QD>(out, n, Q, c, depth);
    }
  }

  // ---- game values (cpp/src/mettagrid/core/game_value.cpp:14-148): postfix code on a small f32 stack ----
  template <int QD>
  __device__ MGX_OUTLINE float eval_code(int start, int count, int entity, const MgxCtx& outer, int depth) const {
    MgxValueStack st;  // registers, not a dynamically indexed array (which the compiler would place in scratch)
    PP code = prog() + d.sec[MGX_SEC_GV_CODE] + start * MGX_GV_WORDS;
    for (int i = 0; i < count; i++, code += MGX_GV_WORDS) {
      int a0 = code[MGX_GV_A0], a1 = code[MGX_GV_A1], a2 = code[MGX_GV_A2];
      switch (code[MGX_GV_OP]) {
        case MGX_GOP_INVENTORY: st.push(entity >= 0 ? (float)inv(entity, a0) : 0.f); break;
        case MGX_GOP_STAT: {
          float v = 0.f;
          if (a0 == 1) { gstat_touch(a1); v = d.game_stats[(size_t)envi() * d.NG + a1]; }
          else { int a = agent_of(entity); if (a >= 0) { astat_touch(a, a1); v = astat_get(a, a1); } }
          st.push(v);
          break;
        }
        case MGX_GOP_CONST: st.push(__int_as_float(a0)); break;
        case MGX_GOP_ADD_TERM: {
          float t = st.pop();
          if (a0) t = mgx_logf(__fadd_rn(t, 1.0f));
          if (a1) t = __fmul_rn(t, __int_as_float(a2));
          float acc = st.pop();
          st.push(__fadd_rn(acc, t));
          break;
        }
        case MGX_GOP_RATIO: { float den = st.pop(), num = st.pop(); st.push(den > 0.f ? __fdiv_rn(num, den) : num); break; }
        case MGX_GOP_MAX2: { float v = st.pop(), b = st.pop(); st.push((b < v) ? v : b); break; }
        case MGX_GOP_MIN2: { float v = st.pop(), b = st.pop(); st.push((v < b) ? v : b); break; }
        case MGX_GOP_QUERY_INVENTORY: {  // game_value.cpp:45-57 (captured ctx: actor = the value's entity)
          float total = 0.f;
          if constexpr (X && QD > 0) {
            MgxCtx q = outer;
            q.actor = entity;
            int n = eval_query<QD>(a1, q, depth);
            const uint16_t* res = qbuf(QB_BASE + 2 * depth);
            for (int k = 0; k < n; k++) total = __fadd_rn(total, (float)inv(res[k], a0));
          } else {
            flag(4u);
          }
          st.push(total);
          break;
        }
        case MGX_GOP_QUERY_COUNT: {
          float cnt = 0.f;
          if constexpr (X && QD > 0) {
            MgxCtx q = outer;
            q.actor = entity;
            cnt = (float)eval_query<QD>(a0, q, depth);
          } else {
            flag(4u);
          }
          st.push(cnt);
          break;
        }
      }
    }
    return st.n > 0 ? st.s0 : 0.f;
  }
  template <int QD>
  __device__ float eval_value(int rec, int entity, const MgxCtx& c, int depth) const {
    PP V = prog() + d.sec[MGX_SEC_OBS_VALUES] + rec * MGX_OV_WORDS;
    return eval_code<QD>(V[MGX_OV_GV_START], V[MGX_OV_GV_COUNT], entity, c, depth);
  }

  // ---- filters (handler/filters/*.hpp) as short-circuit code ----
  __device__ __forceinline__ int resolve(const MgxCtx& c, int ent) const { return ent == MGX_ENT_ACTOR ? c.actor : c.target; }
  template <int QD>
  __device__ MGX_BIG bool atom(PP a, const MgxCtx& c, int depth) const {
    return atom_v<QD>(a[MGX_AT_OP], a[MGX_AT_A0], a[MGX_AT_A1], a[MGX_AT_A2], c, depth);
  }
  // (operands by value: the generated handler code — mgx_handlers_gen.h — passes them as constants and the switch folds)
  template <int QD>
  __device__ MGX_BIG bool atom_v(int op, int a0, int a1, int a2, const MgxCtx& c, int depth) const {
    switch (op) {
      case MGX_FOP_VIBE: { int e = resolve(c, a0); return e != MGX_SLOT_NONE && (e >= 0 ? (int)d.obj_vibe[so(e)] : 0) == a1; }
      case MGX_FOP_RESOURCE: { int e = resolve(c, a0); return e != MGX_SLOT_NONE && inv_of(e, a1) >= a2; }
      case MGX_FOP_SHARED_TAG: {
        PP mask = prog() + d.sec[MGX_SEC_WORDLIST] + a0;
        uint32_t any = 0;
        for (int w = 0; w < MGX_TAG_WORDS; w++) any |= tagword(c.actor, w, c) & tagword(c.target, w, c) & (uint32_t)mask[w];
        return any != 0;
      }
      case MGX_FOP_TAG: {
        int e = resolve(c, a0);
        if (e == MGX_SLOT_NONE) return false;
        PP mask = prog() + d.sec[MGX_SEC_WORDLIST] + a1;
        uint32_t any = 0;
        for (int w = 0; w < MGX_TAG_WORDS; w++) any |= tagword(e, w, c) & (uint32_t)mask[w];
        return any != 0;
      }
      case MGX_FOP_TARGET_LOC_EMPTY: return c.target == MGX_SLOT_NONE;
      case MGX_FOP_TARGET_IS_USABLE: return c.target != MGX_SLOT_NONE;
      case MGX_FOP_PERIODIC: return step >= (uint32_t)a1 && ((step - (uint32_t)a1) % (uint32_t)a0) == 0;
      case MGX_FOP_GAME_VALUE: {
        int e = resolve(c, a0);
        float v = eval_value<QD>(a1, e, c, depth);
        float t = eval_value<QD>(a2, e, c, depth);
        return v >= t;
      }
      case MGX_FOP_MAX_DISTANCE: {  // filters/max_distance_filter.hpp:27-67
        int e = resolve(c, a0);
        if (e < 0) return false;
        uint16_t erc = d.obj_rc[so(e)];
        long long r = a1;
        if (a2 < 0) {
          int ref = c.source >= 0 ? c.source : c.actor;
          if (ref < 0) return false;
          if (a1 == 0) return true;
          uint16_t rrc = d.obj_rc[so(ref)];
          long long dr = (int)(erc >> 8) - (int)(rrc >> 8), dc = (int)(erc & 0xFF) - (int)(rrc & 0xFF);
          return dr * dr + dc * dc <= r * r;
        }
        if constexpr (X && QD > 0) {
          int n = eval_query<QD>(a2, c, depth);
          if (a1 == 0) return n > 0;
          const uint16_t* res = qbuf(QB_BASE + 2 * depth);
          for (int i = 0; i < n; i++) {
            uint16_t src = d.obj_rc[so(res[i])];
            long long dr = (int)(erc >> 8) - (int)(src >> 8), dc = (int)(erc & 0xFF) - (int)(src & 0xFF);
            if (dr * dr + dc * dc <= r * r) return true;
          }
        } else {
          flag(4u);
        }
        return false;
      }
      case MGX_FOP_QUERY_RESOURCE: {  // filters/query_resource_filter.hpp:26-41
        if constexpr (X && QD > 0) {
          int n = eval_query<QD>(a0, c, depth);
          const uint16_t* res = qbuf(QB_BASE + 2 * depth);
          PP req = prog() + d.sec[MGX_SEC_WORDLIST] + a1;
          for (int i = 0; i < a2; i++) {
            uint32_t total = 0, need = (uint32_t)req[i * 2 + 1];
            for (int k = 0; k < n; k++) { total += inv(res[k], req[i * 2]); if (total >= need) break; }
            if (total < need) return false;
          }
          return true;
        } else {
          flag(4u);
          return false;
        }
      }
      case MGX_FOP_TRUE: return true;
      default: return false;
    }
  }
  template <int QD>
  __device__ MGX_OUTLINE bool check_filters(int pc, const MgxCtx& c, int depth) const {  // handler/handler.cpp:95-103
    PP atoms = prog() + d.sec[MGX_SEC_ATOMS];
    while (pc >= 0) {
      PP a = atoms + pc * MGX_AT_WORDS;
      pc = atom<QD>(a, c, depth) ? a[MGX_AT_ON_TRUE] : a[MGX_AT_ON_FALSE];
    }
    return pc == MGX_PC_PASS;
  }
#ifdef MGX_ACT_TU  // the lane-per-agent kernels only run programs whose action-phase handlers evaluate no query (MgxDev::act_par)
  static constexpr int TOPQ = 0;
#else
  static constexpr int TOPQ = X ? 3 : 0;  // query nesting available to top-level filter/value evaluation
#endif

  // ---- grid (core/grid.hpp:75-113) ----
  __device__ __forceinline__ uint16_t& cell(int r, int c) const { return d.grid[(size_t)envi() * d.H * d.W + r * d.W + c]; }
  __device__ void territory_moved(int slot) const {  // TerritoryTracker::notify_source_moved (:187-199)
    if constexpr (X) {
      if (d.NTS == 0) return;
      uint16_t* to = d.ts_obj + (size_t)envi() * d.NTS;
      uint16_t* tc = d.ts_ctrl + (size_t)envi() * d.NTS;
      uint16_t* tr = d.ts_rc + (size_t)envi() * d.NTS;
      int n = d.ts_count[envi()];
      // entries of the moved object are re-registered at the END, keeping their relative order
      int mine = 0;
      for (int i = 0; i < n; i++) mine += to[i] == (uint16_t)slot;
      if (!mine) return;
      d.terr_dirty[envi()] = 1;  // the cached ownership map (mgx_terr_kernel) is stale
      uint16_t rc = d.obj_rc[so(slot)];
      for (int pass = 0; pass < mine; pass++) {
        int i = 0;
        while (to[i] != (uint16_t)slot) i++;  // originals of this object always precede the entries already moved
        uint16_t ctrl = tc[i];
        for (int k = i; k + 1 < n; k++) { to[k] = to[k + 1]; tc[k] = tc[k + 1]; tr[k] = tr[k + 1]; }
        to[n - 1] = (uint16_t)slot; tc[n - 1] = ctrl; tr[n - 1] = rc;
      }
    }
  }
  __device__ __forceinline__ void territory_tags_changed(int o) const {  // ownership depends on the sources' tags (:223-229)
    if constexpr (X) {
      if (d.NTS > 0 && cls_of(o)[MGX_C_TERR_COUNT] > 0) d.terr_dirty[envi()] = 1;
    }
  }
  // Grid::move_object.  known_empty: the caller saw (r, c) empty and no grid cell has been written since (the acting
  // agent stepping into the cell its own line scan just read) — the target is then not read again.
  __device__ MGX_BIG bool move_object(int slot, int r, int c, bool known_empty = false) const {
    if (r < 0 || c < 0 || r >= d.H || c >= d.W) return false;
    if (!known_empty && cell(r, c) != 0) return false;
    const bool own = slot == cur_slot && AL().rc != nullptr;
    const uint16_t rc = own ? AL().rc[cur_agent * MGX_WORLD_EPG + AL().lane] : d.obj_rc[so(slot)];
    cell(r, c) = (uint16_t)(slot + 1);
    cell(rc >> 8, rc & 0xFF) = 0;
    grid_dirty = 1;
    d.obj_rc[so(slot)] = (uint16_t)((r << 8) | c);
    {
      const int a = slot == cur_slot ? cur_agent : agent_of(slot);
      if (a >= 0) {
        d.ag_rc[ao(a)] = (uint16_t)((r << 8) | c);
        if (AL().rc) AL().rc[a * MGX_WORLD_EPG + AL().lane] = (uint16_t)((r << 8) | c);
      }
    }
    territory_moved(slot);
    return true;
  }

  // ---- objects created / removed at run time ----
  __device__ int spawn_object(int cls_id, int r, int c) const {  // create_object_from_config + Grid::add_object + TagIndex
    if constexpr (!X) { flag(4u); return -1; } else {
      CP C = cls(cls_id);
      if (C[MGX_C_KIND] == MGX_KIND_AGENT) { flag(64u); return -1; }
      uint32_t n = d.num_objs[envi()];
      if ((int)n >= d.S) { flag(8u); return -1; }
      d.num_objs[envi()] = n + 1;
      const int slot = (int)n;
      d.obj_cls[so(slot)] = (uint16_t)cls_id;
      d.obj_rc[so(slot)] = (uint16_t)((r << 8) | c);
      d.obj_vibe[so(slot)] = (uint8_t)C[MGX_C_INITIAL_VIBE];
      d.obj_agent[so(slot)] = MGX_NO_AGENT;
      d.obj_visited[so(slot)] = 0;
      d.obj_order[so(slot)] = ~0ull;
      for (int k = 0; k < d.R; k++) inv(slot, k) = 0;
      d.obj_flags[so(slot)] = 2;  // no ObservationEncoder: inventory is not observable (grid_object.cpp:194)
      cell(r, c) = (uint16_t)(slot + 1); grid_dirty = 1;
      PP ii = prog() + d.sec[MGX_SEC_INIT_INV] + C[MGX_C_INIT_INV_START] * MGX_II_WORDS;
      for (int i = 0; i < C[MGX_C_INIT_INV_COUNT]; i++, ii += MGX_II_WORDS) inv_update<0>(slot, ii[MGX_II_ITEM], ii[MGX_II_AMOUNT], true, true);
      for (int w = 0; w < MGX_TAG_WORDS; w++) d.obj_tags[so(slot) * MGX_TAG_WORDS + w] = (uint32_t)C[MGX_C_TAGS + w];
      for (int t = 0; t < 256 && d.NL > 0; t++) {
        if (!(((uint32_t)C[MGX_C_TAGS + (t >> 5)] >> (t & 31)) & 1u)) continue;
        int li = tag_list(t);
        if (li >= 0) { uint16_t k = tl_count(li); if (k < d.S) { tl_items(li)[k] = (uint16_t)slot; tl_count(li) = k + 1; } }
      }
      if (C[MGX_C_AOE_COUNT] > 0) { uint16_t k = d.def_count[envi()]; d.def_aoe[(size_t)envi() * d.S + k] = (uint16_t)slot; d.def_count[envi()] = k + 1; }
      return slot;
    }
  }
  __device__ void register_aoes(int slot) const {  // AOETracker::register_source :128-134 (flush_deferred :154-159)
    CP C = cls_of(slot);
    for (int i = 0; i < C[MGX_C_AOE_COUNT]; i++) {
      int a = C[MGX_C_AOE_START] + i;
      if (aoe(a)[MGX_AO_STATIC]) {
        uint16_t n = d.fx_count[envi()];
        if (n >= d.NF) { flag(8u); continue; }
        size_t q = (size_t)envi() * d.NF + n;
        d.fx_obj[q] = (uint16_t)slot; d.fx_aoe[q] = (uint16_t)a; d.fx_rc[q] = d.obj_rc[so(slot)];
        for (int ai = 0; ai < d.A; ai++) fx_in(n, ai) &= ~(1u << (n & 31));
        d.fx_count[envi()] = n + 1;
      } else {
        uint16_t n = d.mb_count[envi()];
        if (n >= d.NM) { flag(8u); continue; }
        size_t q = (size_t)envi() * d.NM + n;
        d.mb_obj[q] = (uint16_t)slot; d.mb_aoe[q] = (uint16_t)a;
        for (int ai = 0; ai < d.A; ai++) mb_in(n, ai) &= ~(1u << (n & 31));
        d.mb_count[envi()] = n + 1;
      }
    }
  }
  __device__ void remove_object(int slot) const {  // resource_mutation.hpp:88-97
    if (agent_of(slot) >= 0) { flag(64u); return; }
    if (d.NF) {
      const size_t fb = (size_t)envi() * d.NF;
      for (int f = 0; f < d.fx_count[envi()]; f++) {
        if (d.fx_obj[fb + f] != (uint16_t)slot) continue;
        PP a = aoe(d.fx_aoe[fb + f]);
        for (int ai = 0; ai < d.A; ai++) {
          uint32_t& w = fx_in(f, ai);
          if ((w >> (f & 31)) & 1u) { w &= ~(1u << (f & 31)); presence(a, d.ag_obj[ao(ai)], -1); }
        }
        d.fx_obj[fb + f] = 0xFFFF;  // unregistered
      }
    }
    if (d.NM) {
      const size_t mb = (size_t)envi() * d.NM;
      for (int f = 0; f < d.mb_count[envi()]; f++) {
        if (d.mb_obj[mb + f] != (uint16_t)slot) continue;
        PP a = aoe(d.mb_aoe[mb + f]);
        for (int ai = 0; ai < d.A; ai++) {
          uint32_t& w = mb_in(f, ai);
          if ((w >> (f & 31)) & 1u) { w &= ~(1u << (f & 31)); presence(a, d.ag_obj[ao(ai)], -1); }
        }
        d.mb_obj[mb + f] = 0xFFFF;
      }
    }
    uint16_t rc = d.obj_rc[so(slot)];
    cell(rc >> 8, rc & 0xFF) = 0; grid_dirty = 1;
    d.obj_flags[so(slot)] |= 1;
    for (int t = 0; t < 256 && d.NL > 0; t++) {  // TagIndex::unregister_object (core/tag_index.cpp:21-31)
      if (!has_tag(slot, t)) continue;
      int li = tag_list(t);
      if (li < 0) continue;
      uint16_t* it = tl_items(li);
      int n = tl_count(li), k = 0;
      for (int i = 0; i < n; i++) { uint16_t v = it[i]; if (v != (uint16_t)slot) it[k++] = v; }
      tl_count(li) = (uint16_t)k;
    }
  }

  // ---- mutations (handler/mutations/*.hpp) ----
  // Every mutation except the ones that apply a handler (UseTarget, tag changes with lifecycle handlers, materialized
  // query recomputation): those are steps of the handler VMs (run_handler / vm_run), which own all nesting.
  __device__ MGX_BIG void mutate(PP m, MgxCtx& c) const {
    const int op = m[MGX_MU_OP];
    if (op < MGX_MOP_GAME_VALUE) mutate_v(op, m[MGX_MU_A0], m[MGX_MU_A1], m[MGX_MU_A2], m[MGX_MU_A3], m[MGX_MU_A4], c);
    else if constexpr (X) mutate_ext(m, c);
    else flag(4u);
  }
  // the mutations every variant has (operands by value, see atom_v)
  __device__ MGX_BIG void mutate_v(int op, int a0, int a1, int a2, int a3, int a4, MgxCtx& c) const {
    switch (op) {
      case MGX_MOP_RESOURCE_DELTA: {  // resource_mutation.hpp:25-47
        if constexpr (X) {
          if (c.deferred && a0 == MGX_ENT_TARGET && c.target >= 0 && !(cls_of(c.target)[MGX_C_MODIFIER_MASK] & (1 << a1))) {
            int* dd = XL().def_delta + a1 * XL().stride + XL().lane;
            int* meta = XL().def_delta + 13 * XL().stride + XL().lane;       // seen mask
            int* cnt = XL().def_delta + 14 * XL().stride + XL().lane;        // number of first-seen entries
            if (!((*meta >> a1) & 1)) { *meta |= 1 << a1; XL().def_delta[(15 + *cnt) * XL().stride + XL().lane] = a1; *cnt += 1; *dd = 0; }
            *dd += a2;
            break;
          }
        }
        int e = resolve(c, a0);
        if (e >= 0) inv_update<1>(e, a1, a2);
        else if (e == MGX_SLOT_PROXY) flag(32u);
        break;
      }
      case MGX_MOP_RESOURCE_TRANSFER: {  // resource_mutation.hpp:60-98
        int s = resolve(c, a0), t = resolve(c, a1);
        if (s < 0 || t < 0) break;
        int amount = a3 < 0 ? (int)inv(s, a2) : a3;
        int moved = transfer(s, t, a2, amount);
        int sa = agent_of(s);
        if (moved > 0 && sa >= 0) astat_add(sa, d.wk[MGX_S_RES_DEPOSITED_BASE] + a2, (float)moved);
#ifndef MGX_ACT_TU
        if constexpr (X) {
          if (a4 && (d.obj_order[so(s)] & 0xF) == 0xF) remove_object(s);  // resource_mutation.hpp:88-97
        }
#endif
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
          PP ids = prog() + d.sec[MGX_SEC_WORDLIST] + a1;
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
        float v = eval_value<TOPQ>(a3, e, c, 0);
#ifdef MGX_ACT_TU
        if (a0 == 0) act_gstat_set(a2, v);
#elif MGX_HELPERS_ON
        if (a0 == 0) { if (duo_on) act_gstat_set(a2, v); else gstat_set(a2, v); }
#else
        if (a0 == 0) gstat_set(a2, v);
#endif
        else { int a = agent_of(e); if (a >= 0) astat_set(a, a2, v); }
        break;
      }
      case MGX_MOP_CHANGE_VIBE: { int e = resolve(c, a0); if (e >= 0) d.obj_vibe[so(e)] = (uint8_t)a1; break; }
      case MGX_MOP_RELOCATE:  // relocate_mutation.hpp: agents only
        if (c.actor >= 0 && (c.actor == cur_slot || agent_of(c.actor) >= 0))
          move_object(c.actor, c.target_r, c.target_c, c.target == MGX_SLOT_NONE && !grid_dirty);
        break;
      case MGX_MOP_SWAP: {  // swap_mutation.hpp:15-21, core/grid.hpp:92-105
        int xa = agent_of(c.actor), ya = agent_of(c.target);
        if (xa < 0 || ya < 0) break;
        uint16_t rx = d.obj_rc[so(c.actor)], ry = d.obj_rc[so(c.target)];
        cell(rx >> 8, rx & 0xFF) = (uint16_t)(c.target + 1);
        cell(ry >> 8, ry & 0xFF) = (uint16_t)(c.actor + 1); grid_dirty = 1;
        d.obj_rc[so(c.actor)] = ry;
        d.obj_rc[so(c.target)] = rx;
        d.ag_rc[ao(xa)] = ry;
        d.ag_rc[ao(ya)] = rx;
        if (AL().rc) { AL().rc[xa * MGX_WORLD_EPG + AL().lane] = ry; AL().rc[ya * MGX_WORLD_EPG + AL().lane] = rx; }
        territory_moved(c.actor);  // on_object_moved twice (core/grid.hpp:100-103)
        territory_moved(c.target);
        astat_add(xa, d.wk[MGX_S_SWAP], 1.f);
        break;
      }
      default: flag(4u); break;
    }
  }
  __device__ MGX_BIG void mutate_ext(PP m, MgxCtx& c) const {
    int a0 = m[MGX_MU_A0], a1 = m[MGX_MU_A1], a2 = m[MGX_MU_A2], a3 = m[MGX_MU_A3];
    switch (m[MGX_MU_OP]) {
      case MGX_MOP_GAME_VALUE: {  // game_value_mutation.hpp:21-27 (a0 target entity, a1 value, a2 source)
        int e = resolve(c, a0);
        float delta = eval_value<TOPQ>(a2, e, c, 0);
        PP V = prog() + d.sec[MGX_SEC_OBS_VALUES] + a1 * MGX_OV_WORDS;
        PP code = prog() + d.sec[MGX_SEC_GV_CODE] + V[MGX_OV_GV_START] * MGX_GV_WORDS;
        if (code[MGX_GV_OP] == MGX_GOP_INVENTORY) {
          if (e >= 0) inv_update<1>(e, code[MGX_GV_A0], (int)delta);
        } else if (code[MGX_GV_OP] == MGX_GOP_STAT) {
          if (code[MGX_GV_A0] == 1) gstat_add(code[MGX_GV_A1], delta);
          else { int a = agent_of(e); if (a >= 0) astat_add_touch(a, code[MGX_GV_A1], delta); }
        }
        break;
      }
#ifndef MGX_ACT_TU  // (tag mutations, object creation / removal, query mutations never reach the lane-per-agent kernels)
      case MGX_MOP_PUSH_OBJECT: {  // push_object_mutation.hpp:33-67
        if (c.actor < 0 || c.target < 0) { c.mutation_failed = true; break; }
        uint16_t arc = d.obj_rc[so(c.actor)], trc = d.obj_rc[so(c.target)];
        int dr = min(max((int)(trc >> 8) - (int)(arc >> 8), -1), 1), dc = min(max((int)(trc & 0xFF) - (int)(arc & 0xFF), -1), 1);
        int nr = (trc >> 8) + dr, nc = (trc & 0xFF) + dc;
        if (nr < 0 || nc < 0 || nr >= d.H || nc >= d.W || cell(nr, nc) != 0 || !move_object(c.target, nr, nc)) c.mutation_failed = true;
        break;
      }
      case MGX_MOP_SPAWN_OBJECT: {  // spawn_object_mutation.cpp:10-64
        if (cell(c.target_r, c.target_c) != 0) { c.mutation_failed = true; break; }
        int o = spawn_object(a0, c.target_r, c.target_c);
        if (o < 0) { c.mutation_failed = true; break; }
        c.target = o;
        break;
      }
      case MGX_MOP_RAYCAST_SPAWN: {  // raycast_spawn_mutation.cpp:15-92
        if (c.target < 0) { c.mutation_failed = true; break; }
        uint16_t orc = d.obj_rc[so(c.target)];
        MgxCtx tc = c;
        tc.actor = c.target;
        int range = (int)eval_value<TOPQ>(a3, c.target, tc, 0);
        if (range <= 0) break;
        PP dirs = prog() + d.sec[MGX_SEC_WORDLIST] + a1;
        const int blocker_pc = m[MGX_MU_A4];
        for (int k = 0; k < a2; k++)
          for (int dist = 1; dist <= range; dist++) {
            int r = (orc >> 8) + dirs[k * 2] * dist, cc = (orc & 0xFF) + dirs[k * 2 + 1] * dist;
            if (r < 0 || cc < 0 || r >= d.H || cc >= d.W) break;
            int ex = (int)cell(r, cc) - 1;
            if (ex >= 0) {
              bool blocker = false;
              if (blocker_pc != MGX_PC_FAIL) { MgxCtx b = c; b.target = ex; blocker = check_filters<TOPQ>(blocker_pc, b, 0); }
              if (blocker) break;
              continue;
            }
            spawn_object(a0, r, cc);
          }
        break;
      }
      case MGX_MOP_QUERY_INVENTORY: {  // query_inventory_mutation.hpp:26-52
        int n = eval_query<TOPQ>(a0, c, 0);
        const uint16_t* res = qbuf(QB_BASE);
        PP dl = prog() + d.sec[MGX_SEC_WORDLIST] + a1;
        PP sn = prog() + d.sec[MGX_SEC_WORDLIST] + m[MGX_MU_A4];
        int nsn = m[MGX_MU_PAD0];
        if (a3 >= 0) {
          int srcobj = resolve(c, a3);
          if (srcobj < 0) break;
          for (int k = 0; k < n; k++)
            for (int i = 0; i < a2; i++) {
              int o = res[k], item = dl[i * 2], delta = dl[i * 2 + 1], actual = 0;
              if (delta > 0) actual = transfer(srcobj, o, item, delta);
              else if (delta < 0) actual = transfer(o, srcobj, item, -delta);
              if (actual != 0)
                for (int q = nsn - 1; q >= 0; q--)
                  if (sn[q * 2] == item) { gstat_add(sn[q * 2 + 1], (float)actual); break; }
            }
        } else {
          for (int k = 0; k < n; k++)
            for (int i = 0; i < a2; i++) inv_update<1>(res[k], dl[i * 2], dl[i * 2 + 1]);
        }
        break;
      }
#endif
      default: flag(4u); break;
    }
  }
  // Handler::try_apply (handler/handler.cpp:76-93) and MultiHandler::try_apply (multi_handler.cpp:8-21) are executed
  // by two iterative VMs — no recursion anywhere in the handler path:
  //   run_handler (lean variant): frames in registers, the only nesting is multi-handler children and UseTarget;
  //   vm_run (extended variant): frames and handler contexts in LDS, also tag lifecycle handlers and the phases of a
  //   materialized-query recomputation.
  // Iterative form of apply_handler<3> for the lean variant (no tag handlers, no queries: the only nested handler
  // applications are a multi-handler's children and UseTarget's on_use / on_after_use).  An explicit stack of at most
  // four frames (the template depth of the recursive form) lives in registers; there is ONE copy of the filter and
  // mutation interpreters and no recursion, so the whole VM can be inlined into the kernel.
  // The only ctx field a lean handler chain changes is mutation_failed, and on_use runs on a COPY of the ctx
  // (use_target_mutation.hpp:17-29): ctx copies are therefore one `failed` bit per ctx slot, the rest is shared.
  __device__ __forceinline__ bool run_handler(int h0, MgxCtx& c) const {
    enum { ST_ENTER = 0, ST_KIDS = 1, ST_MUTS = 2, ST_USE_RET = 3, ST_AFTER_RET = 4 };
    int fh0 = h0, fh1 = 0, fh2 = 0, fh3 = 0;  // handler of frame k
    int fs0 = 0, fs1 = 0, fs2 = 0, fs3 = 0;   // i | stage << 16 | any << 20 | ctx slot << 24
    int sp = 0;
    bool rv = false;
    uint32_t failed = 0;
    PP handlers = prog() + d.sec[MGX_SEC_HANDLERS];
    while (sp >= 0) {
      const int h = sp == 0 ? fh0 : sp == 1 ? fh1 : sp == 2 ? fh2 : fh3;
      const int fs = sp == 0 ? fs0 : sp == 1 ? fs1 : sp == 2 ? fs2 : fs3;
      int i = fs & 0xFFFF, stage = (fs >> 16) & 0xF, any = (fs >> 20) & 1;
      const int cs = (fs >> 24) & 3;
      PP hd = handlers + h * MGX_HD_WORDS;
      int push_h = -1, push_cs = 0;  // handler to apply next (new frame), if any
      bool pop = false;
      // stages in sequence (not else-if): a lane chains return -> enter -> kids / mutation within one trip
      if (stage == ST_USE_RET) {
        stage = ST_MUTS;
        if (!rv) failed |= 1u << cs;
        else {
          const int after = cls_of(c.actor)[MGX_C_ON_AFTER_USE];
          if (after >= 0) { stage = ST_AFTER_RET; push_h = after; push_cs = cs; }
        }
      } else if (stage == ST_AFTER_RET) {  // the result of on_after_use is ignored
        stage = ST_MUTS;
      }
      if (stage == ST_ENTER && push_h < 0) {
        if (hd[MGX_HD_KIND] == MGX_HK_LEAF) {
          if (!check_filters<0>(hd[MGX_HD_FILTER_PC], c, 0)) { rv = false; pop = true; }
          else { failed &= ~(1u << cs); stage = ST_MUTS; i = 0; }
        } else if (sp == 3) {  // a multi-handler at the last level: recursive form flags and returns "none applied"
          flag(4u);
          rv = false;
          pop = true;
        } else {
          stage = ST_KIDS; i = 0; any = 0;
        }
      }
      if (stage == ST_KIDS && !pop && push_h < 0) {
        bool done = false;
        if (i > 0 && rv) {
          any = 1;
          if (hd[MGX_HD_KIND] == MGX_HK_FIRST_MATCH) { rv = true; pop = true; done = true; }
        }
        if (!done) {
          if (i < hd[MGX_HD_CHILD_COUNT]) {
            push_h = (prog() + d.sec[MGX_SEC_CHILDREN] + hd[MGX_HD_CHILD_START])[i];
            push_cs = cs;
            i++;
          } else {
            rv = any != 0;
            pop = true;
          }
        }
      }
      if (stage == ST_MUTS && !pop && push_h < 0) {
        if (i > 0 && ((failed >> cs) & 1u)) { rv = false; pop = true; }
        else if (i >= hd[MGX_HD_MUT_COUNT]) { rv = true; pop = true; }
        else {
          PP m = prog() + d.sec[MGX_SEC_MUTS] + (hd[MGX_HD_MUT_START] + i) * MGX_MU_WORDS;
          i++;
          if (m[MGX_MU_OP] == MGX_MOP_USE_TARGET) {  // use_target_mutation.hpp:17-29, core/grid_object.cpp:72-80
            int h2 = -1;
            const bool valid = c.target >= 0 && c.actor >= 0 && (c.actor == cur_slot || agent_of(c.actor) >= 0);
            if (valid) {
              if (sp == 3) flag(4u);
              else h2 = cls_of(c.target)[MGX_C_ON_USE];
            }
            if (h2 < 0) failed |= 1u << cs;
            else { stage = ST_USE_RET; push_h = h2; push_cs = cs + 1; }
          } else {
            c.mutation_failed = false;
            mutate(m, c);
            if (c.mutation_failed) failed |= 1u << cs;
          }
        }
      }
      if (pop) { sp--; continue; }
      const int nfs = i | (stage << 16) | (any << 20) | (cs << 24);
      if (sp == 0) fs0 = nfs; else if (sp == 1) fs1 = nfs; else if (sp == 2) fs2 = nfs; else fs3 = nfs;
      if (push_h >= 0) {
        sp++;
        const int pfs = (ST_ENTER << 16) | (push_cs << 24);
        if (sp == 1) { fh1 = push_h; fs1 = pfs; } else if (sp == 2) { fh2 = push_h; fs2 = pfs; } else { fh3 = push_h; fs3 = pfs; }
      }
    }
    c.mutation_failed = (failed & 1u) != 0;
    return rv;
  }
  // ---- extended handler VM ----
  // One loop, one copy of the filter and mutation interpreters, an explicit stack of MGX_VM_FRAMES frames and
  // MGX_VM_CTXS handler contexts per lane in LDS ([word][lane], bank-conflict free).  A frame is four words:
  //   [0] handler index, -1 = the "raw" top-level record (filters + mutation list of an event / AoE / territory handler)
  //   [1] i (16) | stage (4) << 16 | any << 20 | nostop << 21 | owns its ctx slot << 22 | ctx slot (4) << 24
  //   [2] sub-state of the mutation in progress (0 = not started)     [3] aux (list lengths of a query recomputation)
  // What the recursive form did by calling itself is a push here: multi-handler children and on_after_use run on
  // their parent's context slot (MultiHandler::try_apply passes ctx by reference, use_target_mutation.hpp:27 applies
  // on_after_use to the caller's ctx), UseTarget's on_use on a copy (:17-25), tag lifecycle handlers on a copy with
  // actor = target = the tagged object (core/grid_object.cpp:83-91).
  __device__ __forceinline__ uint32_t* vm_words() const {
#ifdef MGX_WORLD_IDS
    return (uint32_t*)(mgx_dyn_lds + mgx_world_xlds_off(d.A)) + mgx_world_lane();
#else
    return nullptr;  // the handler VM only runs in the lane-per-env world kernels
#endif
  }
  static __device__ __forceinline__ void ctx_store(uint32_t* vm, int cs, const MgxCtx& c) {
    vm[(MGX_VM_FRAMES * 4 + cs * 2) * MGX_WORLD_EPG] = (uint32_t)((c.actor + 2) & 0xFFFF) | ((uint32_t)((c.target + 2) & 0xFFFF) << 16);
    vm[(MGX_VM_FRAMES * 4 + cs * 2 + 1) * MGX_WORLD_EPG] = (uint32_t)(c.target_r & 0xFF) | ((uint32_t)(c.target_c & 0xFF) << 8) |
        ((uint32_t)((c.proxy_tag + 1) & 0x1FF) << 16) | (c.mutation_failed ? 1u << 28 : 0u) | (c.skip_trigger ? 1u << 29 : 0u) |
        (c.deferred ? 1u << 30 : 0u);
  }
  static __device__ __forceinline__ MgxCtx ctx_load(const uint32_t* vm, int cs) {
    const uint32_t w0 = vm[(MGX_VM_FRAMES * 4 + cs * 2) * MGX_WORLD_EPG], w1 = vm[(MGX_VM_FRAMES * 4 + cs * 2 + 1) * MGX_WORLD_EPG];
    MgxCtx c;
    c.actor = (int)(w0 & 0xFFFF) - 2; c.target = (int)(w0 >> 16) - 2; c.source = MGX_SLOT_NONE;
    c.target_r = (int)(w1 & 0xFF); c.target_c = (int)((w1 >> 8) & 0xFF); c.proxy_tag = (int)((w1 >> 16) & 0x1FF) - 1;
    c.move_direction = 0;
    c.mutation_failed = (w1 >> 28) & 1u; c.skip_trigger = (w1 >> 29) & 1u; c.deferred = (w1 >> 30) & 1u;
    return c;
  }
  // h0 >= 0: Handler / MultiHandler h0 (returns try_apply's result).  h0 < 0: the raw record (raw_fpc, raw_ms, raw_mc):
  // filters, then EVERY mutation with no stop on mutation_failed (events handler/event.cpp:86-92, AoE sources
  // core/aoe_tracker.cpp:99-113, territory handlers core/territory_tracker.cpp:62-66); returns whether the filters passed.
  __device__ MGX_OUTLINE bool vm_run(int h0, int raw_fpc, int raw_ms, int raw_mc, MgxCtx& c0) const {
    enum { ST_ENTER = 0, ST_KIDS = 1, ST_MUTS = 2 };
    uint32_t* vm = vm_words();
    PP handlers = prog() + d.sec[MGX_SEC_HANDLERS];
    PP muts = prog() + d.sec[MGX_SEC_MUTS];
    int sp = 0, nctx = 1;
    bool rv = false;
    ctx_store(vm, 0, c0);
    vm[0] = (uint32_t)h0; vm[MGX_WORLD_EPG] = (uint32_t)(ST_ENTER << 16) | (h0 < 0 ? 1u << 21 : 0u); vm[2 * MGX_WORLD_EPG] = 0; vm[3 * MGX_WORLD_EPG] = 0;
    while (sp >= 0) {
      uint32_t* fr = vm + sp * 4 * MGX_WORLD_EPG;
      const int h = (int)fr[0];
      const uint32_t st = fr[MGX_WORLD_EPG];
      uint32_t sub = fr[2 * MGX_WORLD_EPG];
      int i = st & 0xFFFF, stage = (st >> 16) & 0xF, any = (st >> 20) & 1;
      const bool nostop = (st >> 21) & 1;
      const int cs = (st >> 24) & 0xF;
      MgxCtx c = ctx_load(vm, cs);
      int kind = MGX_HK_LEAF, fpc = raw_fpc, ms = raw_ms, mc = raw_mc, cstart = 0, ccount = 0;
      if (h >= 0) {
        PP hd = handlers + h * MGX_HD_WORDS;
        kind = hd[MGX_HD_KIND]; fpc = hd[MGX_HD_FILTER_PC]; ms = hd[MGX_HD_MUT_START]; mc = hd[MGX_HD_MUT_COUNT];
        cstart = hd[MGX_HD_CHILD_START]; ccount = hd[MGX_HD_CHILD_COUNT];
      }
      int push_h = -1, push_cs = cs;  // handler to apply next (new frame); push_cs == nctx: on the context `pc`
      MgxCtx pc = c;
      bool pop = false;
      if (stage == ST_ENTER) {
        if (kind == MGX_HK_LEAF) {
          if (!check_filters<TOPQ>(fpc, c, 0)) { rv = false; pop = true; }
          else { c.mutation_failed = false; stage = ST_MUTS; i = 0; sub = 0; }
        } else {
          stage = ST_KIDS; i = 0; any = 0;
        }
      }
      if (stage == ST_KIDS && !pop) {
        if (i > 0 && rv) {
          any = 1;
          if (kind == MGX_HK_FIRST_MATCH) { rv = true; pop = true; }
        }
        if (!pop) {
          if (i < ccount) { push_h = (prog() + d.sec[MGX_SEC_CHILDREN] + cstart)[i]; i++; }
          else { rv = any != 0; pop = true; }
        }
      } else if (stage == ST_MUTS && !pop) {
        if (sub == 0) {
          if (!nostop && i > 0 && c.mutation_failed) { rv = false; pop = true; }
          else if (i >= mc) { rv = true; pop = true; }
        }
        if (!pop) {
          PP m = muts + (ms + i) * MGX_MU_WORDS;
          const int op = m[MGX_MU_OP], a0 = m[MGX_MU_A0], a1 = m[MGX_MU_A1], a2 = m[MGX_MU_A2];
          bool done = true;  // this mutation is finished (no handler left to apply for it)
          if (op == MGX_MOP_USE_TARGET) {  // use_target_mutation.hpp:17-29, core/grid_object.cpp:72-80
            if (sub == 0) {
              int h2 = -1;
              if (c.target >= 0 && agent_of(c.actor) >= 0) h2 = cls_of(c.target)[MGX_C_ON_USE];
              if (h2 < 0) c.mutation_failed = true;
              else { push_h = h2; push_cs = nctx; sub = 1; done = false; }
            } else if (sub == 1) {
              if (!rv) c.mutation_failed = true;
              else {
                const int after = cls_of(c.actor)[MGX_C_ON_AFTER_USE];
                if (after >= 0) { push_h = after; sub = 2; done = false; }
              }
            }
          } else if (op == MGX_MOP_ADD_TAG || op == MGX_MOP_REMOVE_TAG) {  // tag_mutation.hpp:16-45
            const int o = resolve(c, a0);
            const bool add = op == MGX_MOP_ADD_TAG;
            bool fire = sub != 0;
            if (sub == 0) { fire = (add ? tag_set(o, a1) : tag_clear(o, a1)) && !c.skip_trigger; sub = 1; }
            if (fire) {
              int hh = -1;
              const int j = next_tag_handler(o, a1, add ? MGX_C_TAG_ADD_START : MGX_C_TAG_REMOVE_START, (int)sub - 1, &hh);
              if (j >= 0) { sub = (uint32_t)j + 2; push_h = hh; push_cs = nctx; pc.actor = pc.target = o; pc.skip_trigger = false; done = false; }
            }
          } else if (op == MGX_MOP_REMOVE_TAGS_PREFIX) {  // tag_mutation.hpp:47-67: remove_tag per matching tag id
            const int o = resolve(c, a0);
            PP ids = prog() + d.sec[MGX_SEC_WORDLIST] + a1;
            int idx = (int)(sub >> 12), jj = (int)(sub & 0xFFF);  // jj: 0 = tag idx not removed yet, else 1 + next record
            while (idx < a2) {
              if (jj == 0) {
                if (!(tag_clear(o, ids[idx]) && !c.skip_trigger)) { idx++; continue; }
                jj = 1;
              }
              int hh = -1;
              const int j = next_tag_handler(o, ids[idx], MGX_C_TAG_REMOVE_START, jj - 1, &hh);
              if (j >= 0) { sub = ((uint32_t)idx << 12) | (uint32_t)(j + 2); push_h = hh; push_cs = nctx; pc.actor = pc.target = o; pc.skip_trigger = false; done = false; break; }
              idx++; jj = 0;
            }
          } else if (op == MGX_MOP_RECOMPUTE_QUERY) {  // query_system.cpp:119-175
            uint16_t* lost = qbuf(QB_LOST);
            uint16_t* keep = qbuf(QB_KEEP);
            uint32_t aux = fr[3 * MGX_WORLD_EPG];
            if (sub == 0) {  // untag the old members and tag the new ones silently, then fire handlers for the differences
              int nl = 0, nk = 0;
              PP mq = prog() + d.sec[MGX_SEC_MATQ];
              for (int q = 0; q < d.n_matq; q++, mq += MGX_MQ_WORDS) {
                if (mq[MGX_MQ_TAG] != a0) continue;
                const int li = tag_list(a0);
                if (li >= 0) { nl = tl_count(li); const uint16_t* it = tl_items(li); for (int k = 0; k < nl; k++) lost[k] = it[k]; }
                for (int k = 0; k < nl; k++) tag_clear(lost[k], a0);
                const int n = eval_query<TOPQ>(mq[MGX_MQ_QUERY], c, 0);
                const uint16_t* res = qbuf(QB_BASE);
                for (int k = 0; k < n; k++) {
                  const int o = res[k];
                  bool dup = false;
                  for (int z = 0; z < nk; z++) dup |= keep[z] == (uint16_t)o;
                  if (!dup) keep[nk++] = (uint16_t)o;
                  tag_set(o, a0);
                }
                break;
              }
              aux = (uint32_t)nl | ((uint32_t)nk << 16);
              fr[3 * MGX_WORLD_EPG] = aux;
              sub = 1u << 28;  // phase 1 (objects that lost the tag), object 0, record 0
            }
            const int nl = (int)(aux & 0xFFFF), nk = (int)(aux >> 16);
            int phase = (int)(sub >> 28), k = (int)(sub & 0xFFFF), j = (int)((sub >> 16) & 0xFFF);
            while (phase <= 2) {
              const int cnt = phase == 1 ? nl : nk;
              const uint16_t* mine = phase == 1 ? lost : keep;
              const uint16_t* other = phase == 1 ? keep : lost;
              const int no = phase == 1 ? nk : nl;
              bool pushed = false;
              for (; k < cnt; k++, j = 0) {
                const int o = mine[k];
                bool both = false;
                for (int z = 0; z < no; z++) both |= other[z] == (uint16_t)o;
                if (both) continue;
                int hh = -1;
                const int jn = next_tag_handler(o, a0, phase == 1 ? MGX_C_TAG_REMOVE_START : MGX_C_TAG_ADD_START, j, &hh);
                if (jn < 0) continue;
                sub = ((uint32_t)phase << 28) | ((uint32_t)(jn + 1) << 16) | (uint32_t)k;
                push_h = hh; push_cs = nctx; pc.actor = pc.target = o; pc.skip_trigger = false; done = false; pushed = true;
                break;
              }
              if (pushed) break;
              phase++; k = 0; j = 0;
            }
          } else {
            mutate(m, c);
          }
          if (done) { sub = 0; i++; }
        }
      }
      ctx_store(vm, cs, c);
      if (pop) {
        if (st & (1u << 22)) nctx--;  // the frame ran on its own context copy: slots are released in stack order
        sp--;
        continue;
      }
      fr[MGX_WORLD_EPG] = (uint32_t)i | ((uint32_t)stage << 16) | ((uint32_t)any << 20) | (st & (3u << 21)) | ((uint32_t)cs << 24);
      fr[2 * MGX_WORLD_EPG] = sub;
      if (push_h >= 0) {
        if (sp + 1 >= MGX_VM_FRAMES || push_cs >= MGX_VM_CTXS) {  // deeper than the engine supports: the nested handler
          flag(4u);                                                // counts as "did not apply"
          rv = false;
          continue;
        }
        const bool own = push_cs == nctx;
        if (own) { ctx_store(vm, nctx, pc); nctx++; }
        sp++;
        uint32_t* nf = vm + sp * 4 * MGX_WORLD_EPG;
        nf[0] = (uint32_t)push_h; nf[MGX_WORLD_EPG] = (uint32_t)(ST_ENTER << 16) | (own ? 1u << 22 : 0u) | ((uint32_t)push_cs << 24);
        nf[2 * MGX_WORLD_EPG] = 0; nf[3 * MGX_WORLD_EPG] = 0;
      }
    }
    c0 = ctx_load(vm, 0);
    return rv;
  }
  // apply a top-level handler: register VM in the lean variant, LDS VM in the extended one
  // (extended variant: programs whose move / on_use / on_after_use / on_tick / game on_tick handlers contain no tag
  // mutation, no query recomputation and no query filter — flat_top, decided by the host — never need the LDS VM for
  // them: the register VM of the lean variant runs them, frames and contexts in registers)
  __device__ __forceinline__ bool apply_top(int h, MgxCtx& c) const {
#ifdef MGX_GEN_HANDLERS  // the unit carries straight-line code for one preset's handlers (mgx_handlers_gen.h)
    if (d.gen_prog == MGX_GEN_ID) return MGX_GEN_HANDLERS<MgxEnvT>::top(*this, h, c);
#endif
    if constexpr (X) {
      if (d.flat_top) return run_handler(h, c);
      return vm_run(h, 0, 0, 0, c);
    } else {
      return run_handler(h, c);
    }
  }

  // Filters + every mutation, no stop on mutation_failed (events, AoE sources, territory handlers).
  __device__ __forceinline__ bool apply_all(int filter_pc, int mut_start, int mut_count, MgxCtx& c) const {
    return vm_run(-1, filter_pc, mut_start, mut_count, c);
  }

  // ---- events (handler/event.cpp:34-101, handler/event_scheduler.cpp:36-53) ----
  __device__ int execute_event(int ev) const {
    for (int hop = 0; hop < 8; hop++) {  // fallback chain
      PP E = prog() + d.sec[MGX_SEC_EVENTS] + ev * MGX_EV_WORDS;
      MgxCtx g = mgx_ctx(MGX_SLOT_NONE, MGX_SLOT_NONE);
      int n = eval_query<TOPQ>(E[MGX_EV_QUERY], g, 0);
      uint16_t* targets = qbuf(QB_EVENT);
      const uint16_t* res = qbuf(QB_BASE);
      for (int i = 0; i < n; i++) targets[i] = res[i];
      int mt = E[MGX_EV_MAX_TARGETS];
      if (mt >= 0 && n > mt && n >= 2) {  // std::shuffle
        uint32_t i = 1;
        if ((n & 1) == 0) { uint32_t j = rng_below(2); uint16_t t = targets[1]; targets[1] = targets[j]; targets[j] = t; i = 2; }
        while (i < (uint32_t)n) {
          uint32_t s = i + 1, x = rng_below(s * (s + 1));
          uint32_t j0 = x / (s + 1), j1 = x % (s + 1);
          uint16_t t = targets[i]; targets[i] = targets[j0]; targets[j0] = t;
          t = targets[i + 1]; targets[i + 1] = targets[j1]; targets[j1] = t;
          i += 2;
        }
      }
      int applied = 0;
      for (int i = 0; i < n; i++) {
        if (mt >= 0 && applied >= mt) break;
        int t = targets[i];
        MgxCtx c = mgx_ctx(t, t);
        uint16_t rc = d.obj_rc[so(t)];
        c.target_r = rc >> 8; c.target_c = rc & 0xFF;
        if (apply_all(E[MGX_EV_FILTER_PC], E[MGX_EV_MUT_START], E[MGX_EV_MUT_COUNT], c)) applied++;
      }
      if (applied != 0 || E[MGX_EV_FALLBACK] < 0) return applied;
      ev = E[MGX_EV_FALLBACK];
    }
    flag(4u);
    return 0;
  }
  __device__ void process_events() const {
    const int32_t* sc = d.P + d.sec[MGX_SEC_SCHEDULE];  // always from HBM: not part of the LDS program copy
    uint32_t k = d.next_event[envi()];
    while (k < (uint32_t)d.n_schedule && (uint32_t)sc[k * MGX_SC_WORDS + MGX_SC_TIMESTEP] <= step) {
      execute_event(sc[k * MGX_SC_WORDS + MGX_SC_EVENT]);
      k++;
    }
    d.next_event[envi()] = k;
  }

  // ---- AoE (core/aoe_tracker.cpp) ----
  // "agent ai is inside source f": one bit per (agent, source), stored AGENT-major — [E][A][FW] / [E][A][MW] words — so that
  // the lane-per-agent kernel owns its words (plain loads and stores, one word covers 32 sources of the agent) and the
  // serial form below walks the same bits.
  __device__ __forceinline__ uint32_t& fx_in(int f, int ai) const { return d.fx_inside[ao(ai) * d.FW + (f >> 5)]; }
  __device__ __forceinline__ uint32_t& mb_in(int m, int ai) const { return d.mb_inside[ao(ai) * d.MW + (m >> 5)]; }
  __device__ __forceinline__ PP aoe(int a) const { return prog() + d.sec[MGX_SEC_AOES] + a * MGX_AO_WORDS; }
  __device__ bool fixed_covers(PP a, uint16_t src_rc, int r, int c) const {  // register_fixed :166-200
    long long range = a[MGX_AO_RADIUS], dr = r - (int)(src_rc >> 8), dc = c - (int)(src_rc & 0xFF);
    if (dr < -range || dr > range || dc < -range || dc > range) return false;
    long long d2 = dr * dr + dc * dc;
    if (d2 > range * range) return false;
    bool territory_style = a[MGX_AO_MUT_COUNT] == 0 && a[MGX_AO_PRES_COUNT] == 0 && range > 0;
    if (territory_style && range >= 2 && d2 == range * range && (dr == 0 || dc == 0)) return false;
    return true;
  }
  __device__ void presence(PP a, int target, int mult) const {  // apply_presence_deltas :122-126
    PP pr = prog() + d.sec[MGX_SEC_PRESENCE] + a[MGX_AO_PRES_START] * MGX_PR_WORDS;
    for (int i = 0; i < a[MGX_AO_PRES_COUNT]; i++, pr += MGX_PR_WORDS) inv_update<1>(target, pr[MGX_PR_RESOURCE], pr[MGX_PR_DELTA] * mult);
  }
  __device__ void apply_fixed(int ai) const {  // :278-362
    const int nf = d.NF ? d.fx_count[envi()] : 0;
    if (nf == 0) return;
    const int tgt = d.ag_obj[ao(ai)];
    const uint16_t rc = d.obj_rc[so(tgt)];
    const int r = rc >> 8, c = rc & 0xFF;
    const size_t fb = (size_t)envi() * d.NF;
    XL().def_delta[13 * XL().stride + XL().lane] = 0;  // seen mask
    XL().def_delta[14 * XL().stride + XL().lane] = 0;  // count
    // exits first.  The reference walks an unordered_set<AOESource*> here (address order); registration order is used.
    for (int f = 0; f < nf; f++) {
      if (d.fx_obj[fb + f] == 0xFFFF) continue;
      uint32_t& w = fx_in(f, ai);
      if (!((w >> (f & 31)) & 1u)) continue;
      PP a = aoe(d.fx_aoe[fb + f]);
      if (!fixed_covers(a, d.fx_rc[fb + f], r, c)) { w &= ~(1u << (f & 31)); presence(a, tgt, -1); }
    }
    for (int f = 0; f < nf; f++) {
      if (d.fx_obj[fb + f] == 0xFFFF) continue;
      PP a = aoe(d.fx_aoe[fb + f]);
      if (!fixed_covers(a, d.fx_rc[fb + f], r, c)) continue;
      if (a[MGX_AO_MUT_COUNT] == 0 && a[MGX_AO_PRES_COUNT] == 0) continue;
      const int src = d.fx_obj[fb + f];
      bool skip_self = !a[MGX_AO_EFFECT_SELF] && src == tgt;
      MgxCtx fc = mgx_ctx(src, tgt);
      fc.deferred = true;
      bool passes = !skip_self && check_filters<TOPQ>(a[MGX_AO_FILTER_PC], fc, 0);
      uint32_t& w = fx_in(f, ai);
      bool was = (w >> (f & 31)) & 1u;
      if (passes && !was) { w |= 1u << (f & 31); presence(a, tgt, +1); }
      else if (!passes && was) { w &= ~(1u << (f & 31)); presence(a, tgt, -1); }
      if (passes && a[MGX_AO_MUT_COUNT] > 0) {
        MgxCtx ac = mgx_ctx(src, tgt);
        ac.deferred = true;
        apply_all(a[MGX_AO_FILTER_PC], a[MGX_AO_MUT_START], a[MGX_AO_MUT_COUNT], ac);
      }
    }
    const int cnt = XL().def_delta[14 * XL().stride + XL().lane];
    for (int k = 0; k < cnt; k++) {  // net delta per resource, first-seen order (:347-361)
      int res = XL().def_delta[(15 + k) * XL().stride + XL().lane];
      int dl = XL().def_delta[res * XL().stride + XL().lane];
      if (dl != 0) inv_update<1>(tgt, res, dl);
    }
  }
  __device__ void apply_mobile() const {  // :364-415
    const int nm = d.NM ? d.mb_count[envi()] : 0;
    const size_t mb = (size_t)envi() * d.NM;
    for (int m = 0; m < nm; m++) {
      if (d.mb_obj[mb + m] == 0xFFFF) continue;
      PP a = aoe(d.mb_aoe[mb + m]);
      const int src = d.mb_obj[mb + m];
      const long long range = a[MGX_AO_RADIUS];
      for (int ai = 0; ai < d.A; ai++) {
        const int tgt = d.ag_obj[ao(ai)];
        if (!a[MGX_AO_EFFECT_SELF] && src == tgt) continue;
        uint32_t& w = mb_in(m, ai);
        bool was = (w >> (m & 31)) & 1u;
        uint16_t s = d.obj_rc[so(src)], t = d.obj_rc[so(tgt)];
        long long dr = (int)(s >> 8) - (int)(t >> 8), dc = (int)(s & 0xFF) - (int)(t & 0xFF);
        if (dr * dr + dc * dc > range * range) {
          if (was) { w &= ~(1u << (m & 31)); presence(a, tgt, -1); }
          continue;
        }
        MgxCtx c = mgx_ctx(src, tgt);
        if (check_filters<TOPQ>(a[MGX_AO_FILTER_PC], c, 0)) {
          if (!was) { w |= 1u << (m & 31); presence(a, tgt, +1); }
          if (a[MGX_AO_MUT_COUNT] > 0) { MgxCtx ac = mgx_ctx(src, tgt); apply_all(a[MGX_AO_FILTER_PC], a[MGX_AO_MUT_START], a[MGX_AO_MUT_COUNT], ac); }
        } else if (was) {
          w &= ~(1u << (m & 31));
          presence(a, tgt, -1);
        }
      }
    }
  }

  // (area effects with one lane per AGENT — mgx_aoe_kernel — live in mgx_aoe_local.h: a register / LDS resident copy of
  // the agent the lane owns, on which the same filter / mutation records are interpreted)

  // ---- territory (core/territory_tracker.cpp) ----
  __device__ int cell_owner(int r, int c, int ti) const {  // compute_cell_ownership :215-252 -> winning tag or -1
    PP TE = prog() + d.sec[MGX_SEC_TERRITORIES] + ti * MGX_TE_WORDS;
    PP prefix = prog() + d.sec[MGX_SEC_WORDLIST] + TE[MGX_TE_TAGS_START];
    const int np = min(TE[MGX_TE_TAGS_COUNT], 8);
    if (TE[MGX_TE_TAGS_COUNT] > 8) flag(4u);
    long long* score = XL().terr_score + XL().lane;
    for (int k = 0; k < np; k++) score[k * XL().stride] = 0;
    const size_t tb = (size_t)envi() * d.NTS;
    const int n = d.ts_count[envi()];
    for (int i = 0; i < n; i++) {
      PP tc = prog() + d.sec[MGX_SEC_TERR_CONTROLS] + d.ts_ctrl[tb + i] * MGX_TC_WORDS;
      if (tc[MGX_TC_TERRITORY] != ti) continue;
      const int strength = tc[MGX_TC_STRENGTH], decay = tc[MGX_TC_DECAY];
      const long long range = decay > 0 ? strength / decay : strength;
      const uint16_t reg = d.ts_rc[tb + i];
      long long dr = r - (int)(reg >> 8), dc = c - (int)(reg & 0xFF);
      if (dr < -range || dr > range || dc < -range || dc > range || dr * dr + dc * dc > range * range) continue;
      const int src = d.ts_obj[tb + i];
      int k = 0;
      while (k < np && !has_tag(src, prefix[k])) k++;
      if (k >= np) continue;
      const uint16_t cur = d.obj_rc[so(src)];  // the score uses the source's CURRENT location (:230-231)
      long long sr = (int)(cur >> 8) - r, sc = (int)(cur & 0xFF) - c;
      unsigned long long d2 = (unsigned long long)(sr * sr + sc * sc);
      long long s = (long long)strength * 1024 - (long long)decay * (long long)mgx_floor_sqrt(d2 << 20);
      if (s > 0) score[k * XL().stride] += s;
    }
    // unique maximum wins, any tie at the maximum -> none: a rule that does not depend on the order the tags are
    // visited in (the reference walks an unordered_map).
    int win = -1;
    long long best = 0;
    bool tied = false;
    for (int k = 0; k < np; k++) {
      long long s = score[k * XL().stride];
      if (s <= 0) continue;
      if (s > best) { win = prefix[k]; best = s; tied = false; }
      else if (s == best && win >= 0) tied = true;
    }
    return tied ? -1 : win;
  }
  // Owner of a cell from the per-env ownership map when it is current (mgx_terr_kernel rebuilds it whenever a source
  // moved or changed tags), else computed on the spot.
  __device__ __forceinline__ int owner_at(int r, int c, int ti) const {
    if (d.terr_owner && !d.terr_dirty[envi()]) {
      const uint16_t v = d.terr_owner[((size_t)envi() * d.NT + ti) * (size_t)(d.H * d.W) + r * d.W + c];
      return v == 0xFFFF ? -1 : (int)v;
    }
    return cell_owner(r, c, ti);
  }
  __device__ void run_territory_handlers(int start, int count, int tag, int tgt) const {
    for (int i = 0; i < count; i++) {
      PP hd = prog() + d.sec[MGX_SEC_HANDLERS] + (start + i) * MGX_HD_WORDS;
      MgxCtx c = mgx_ctx(MGX_SLOT_PROXY, tgt);
      c.proxy_tag = tag;
      uint16_t rc = d.obj_rc[so(tgt)];
      c.target_r = rc >> 8; c.target_c = rc & 0xFF;
      apply_all(hd[MGX_HD_FILTER_PC], hd[MGX_HD_MUT_START], hd[MGX_HD_MUT_COUNT], c);
    }
  }
  __device__ void apply_territory(int ai) const {  // apply_effects :275-346
    const int tgt = d.ag_obj[ao(ai)];
    for (int ti = 0; ti < d.NT; ti++) {
      PP TE = prog() + d.sec[MGX_SEC_TERRITORIES] + ti * MGX_TE_WORDS;
      uint16_t rc = d.obj_rc[so(tgt)];
      int cur = owner_at(rc >> 8, rc & 0xFF, ti);
      int16_t& pv = d.terr_prev[ao(ai) * d.NT + ti];
      int prev = pv;
      if (prev != cur && prev >= 0) run_territory_handlers(TE[MGX_TE_EXIT_START], TE[MGX_TE_EXIT_COUNT], prev, tgt);
      if (prev != cur && cur >= 0) run_territory_handlers(TE[MGX_TE_ENTER_START], TE[MGX_TE_ENTER_COUNT], cur, tgt);
      pv = (int16_t)cur;
      if (cur >= 0) run_territory_handlers(TE[MGX_TE_PRES_START], TE[MGX_TE_PRES_COUNT], cur, tgt);
    }
  }

  // ---- actions ----
  __device__ MGX_BIG bool do_move(int slot, int orient) const {  // actions/move.hpp:81-115, orientation.hpp:28-48
    const int dx = (orient == 2 || orient == 4 || orient == 6) ? -1 : (orient == 3 || orient == 5 || orient == 7) ? 1 : 0;
    const int dy = (orient == 0 || orient == 4 || orient == 5) ? -1 : (orient == 1 || orient == 6 || orient == 7) ? 1 : 0;
    PP mh = prog() + d.sec[MGX_SEC_MOVE_HANDLERS];
    for (int k = 0; k < d.n_move_handlers; k++, mh += MGX_MH_WORDS) {
      uint16_t rc = (AL().rc && slot == cur_slot) ? AL().rc[cur_agent * MGX_WORLD_EPG + AL().lane] : d.obj_rc[so(slot)];
      for (int i = 1; i <= mh[MGX_MH_MAX_RANGE]; i++) {
        int r = (rc >> 8) + dy * i, c = (rc & 0xFF) + dx * i;
        if (r < 0 || c < 0 || r >= d.H || c >= d.W) break;
        int t = (int)cell(r, c) - 1;
        grid_dirty = 0;
        if (t < 0 && !mh[MGX_MH_ACCEPTS_EMPTY]) continue;
        MgxCtx ctx = mgx_ctx(slot, t);
        ctx.target_r = r; ctx.target_c = c; ctx.move_direction = orient;
        const bool applied = apply_top(mh[MGX_MH_HANDLER], ctx);
        grid_dirty = 1;  // the "target seen empty" shortcut is only valid inside this handler chain
        if (applied) return true;
        break;
      }
    }
    return false;
  }
  __device__ MGX_BIG bool handle_action(int ai, int action, int stream) const {  // actions/action_handler.hpp:78-105
    PP ac = prog() + d.sec[MGX_SEC_ACTIONS] + action * MGX_AC_WORDS;
    int kind = ac[MGX_AC_KIND];
    const int li = ai * MGX_WORLD_EPG + AL().lane;
    int slot = AL().slot ? (int)AL().slot[li] : (int)d.ag_obj[ao(ai)];
    cur_agent = ai;
    cur_slot = slot;
    bool ok = true;
    MGX_TICK0();
    if (kind == MGX_AK_MOVE) ok = do_move(slot, ac[MGX_AC_ARG]);
    else if (kind == MGX_AK_VIBE) d.obj_vibe[so(slot)] = (uint8_t)ac[MGX_AC_ARG];  // actions/change_vibe.hpp:48-57
    MGX_TICK(6);
    if (d.defer_book) {
      // Bookkeeping touches only this agent's own counters.  Here: the LDS copies and a result byte (in place of the
      // consumed action id); the HBM side — swm / prev_location write-back and the stat updates — is applied for all
      // agents in one batched pass (bookkeeping_flush), off the serial chain.
      const uint16_t rc = AL().rc[li], prev = AL().prev[li];
      const bool moved = rc != prev;
      if (moved) { AL().swm[li] = 0; AL().prev[li] = rc; }
      else AL().swm[li] += 1;
      AL().act[(stream * d.A + ai) * MGX_WORLD_EPG + AL().lane] = (int16_t)(1 | (kind << 1) | (ok ? 8 : 0) | (moved ? 16 : 0));
      cur_agent = cur_slot = -1;
      return ok;
    }
    uint16_t rc = AL().rc ? AL().rc[li] : d.obj_rc[so(slot)];
    uint16_t prev = AL().prev ? AL().prev[li] : d.ag_prev[ao(ai)];
    if (rc == prev) {
      uint32_t swm = (AL().swm ? AL().swm[li] : d.ag_swm[ao(ai)]) + 1;
      if (AL().swm) AL().swm[li] = swm;
      d.ag_swm[ao(ai)] = swm;
      int sid = d.wk[MGX_S_MAX_STEPS_WITHOUT_MOTION];
      if ((float)swm > astat_get(ai, sid)) astat_set(ai, sid, (float)swm);
    } else {
      if (AL().swm) AL().swm[li] = 0;
      d.ag_swm[ao(ai)] = 0;
      if (AL().prev) AL().prev[li] = rc;
      d.ag_prev[ao(ai)] = rc;
    }
    cur_agent = cur_slot = -1;
    int s_ok = kind == MGX_AK_NOOP ? MGX_S_NOOP_SUCCESS : kind == MGX_AK_MOVE ? MGX_S_MOVE_SUCCESS : MGX_S_VIBE_SUCCESS;
    if (ok) astat_add(ai, d.wk[s_ok], 1.f);
    else { astat_add(ai, d.wk[s_ok + 1], 1.f); astat_add(ai, d.wk[MGX_S_ACTION_FAILED], 1.f); }
    MGX_TICK(7);
    return ok;
  }

  // "action.invalid_index.<k>" for a k without a stat column: one of the agent's (k, count) pairs (include/mgx.h)
  __device__ void invalid_extra(int ai, int a) const {
    int32_t* k = d.ag_invk + ao(ai) * MGX_INVALID_EXTRA;
    float* n = d.ag_invn + ao(ai) * MGX_INVALID_EXTRA;
    for (int q = 0; q < MGX_INVALID_EXTRA; q++) {
      if (n[q] == 0.f) { k[q] = a; n[q] = 1.f; return; }
      if (k[q] == a) { n[q] += 1.f; return; }
    }
    flag(2u);
  }
  __device__ MGX_BIG void track_coverage(int ai) const {  // objects/agent.cpp:49-57
    uint16_t rc = AL().rc ? AL().rc[ai * MGX_WORLD_EPG + AL().lane] : d.obj_rc[so(d.ag_obj[ao(ai)])];
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
    astat_set(ai, d.wk[MGX_S_CELL_UNIQUE], (float)uniq);
    int dist = abs((int)(sp >> 8) - r) + abs(c - (int)(sp & 0xFF));
    uint32_t md = max(d.ag_maxdist[ao(ai)], (uint32_t)dist);
    d.ag_maxdist[ao(ai)] = md;
    astat_set(ai, d.wk[MGX_S_CELL_MAXDIST], (float)md);
  }

  // ---- std::mt19937 + libstdc++ uniform_int_distribution / shuffle ----
  // Deferred half of handle_action's bookkeeping (actions/action_handler.hpp:78-105) for every agent: replays the up to
  // two calls of the tick (primary stream, vibe stream) from their result bytes.  Eight agents at a time, all loads
  // of a chunk in flight together.  Per agent at most four stat cells change: the result counter of each stream
  // (success or failed), action.failed (+1 per failed call, added one by one like the reference) and
  // max_steps_without_motion (set when a call's incremented counter exceeds it).
  __device__ MGX_BIG void bookkeeping_flush(int a_lo = 0, int a_hi = 1 << 30) const {   // agents [a_lo, min(a_hi, A))
    const int A = min(d.A, a_hi), lane = AL().lane;
    const int AS = d.A;   // stride of the vibe-stream result bytes
    const int s_max = d.wk[MGX_S_MAX_STEPS_WITHOUT_MOTION], s_failed = d.wk[MGX_S_ACTION_FAILED];
    // the six result counters as scalars (a lane-varying index into d.wk would be a memory load per agent)
    const int w_noop_ok = d.wk[MGX_S_NOOP_SUCCESS], w_noop_no = d.wk[MGX_S_NOOP_SUCCESS + 1];
    const int w_move_ok = d.wk[MGX_S_MOVE_SUCCESS], w_move_no = d.wk[MGX_S_MOVE_SUCCESS + 1];
    const int w_vibe_ok = d.wk[MGX_S_VIBE_SUCCESS], w_vibe_no = d.wk[MGX_S_VIBE_SUCCESS + 1];
    for (int i0 = a_lo; i0 < A; i0 += 8) {
      int id0[8], id1[8], nfail[8];
      uint32_t res0[8], res1[8], swm0[8], tw[8];
      float v0[8], v1[8], vf[8], vm[8];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int i = min(i0 + q, A - 1);
        const int li = i * MGX_WORLD_EPG + lane;
        res0[q] = (uint32_t)(uint16_t)AL().act[li];
        res1[q] = (uint32_t)(uint16_t)AL().act[AS * MGX_WORLD_EPG + li];
        if (i0 + q >= A) res0[q] = res1[q] = 0;
        // result bytes have bit 0 set; an action id that was never handled (invalid / wrong stream) is cleared by the caller
        auto stat_of = [&](uint32_t r) {
          const int kind = (r >> 1) & 3;
          const bool ok = (r & 8) != 0;
          const int id = kind == MGX_AK_NOOP ? (ok ? w_noop_ok : w_noop_no) : kind == MGX_AK_MOVE ? (ok ? w_move_ok : w_move_no)
                                                                                                    : (ok ? w_vibe_ok : w_vibe_no);
          return (r & 1) ? id : -1;
        };
        id0[q] = stat_of(res0[q]);
        id1[q] = stat_of(res1[q]);
        nfail[q] = ((res0[q] & 9) == 1 ? 1 : 0) + ((res1[q] & 9) == 1 ? 1 : 0);
        const size_t sb = ao(i) * d.NSP;
        // unconditional loads (a dummy in-range cell when there is nothing to update): no branch between them, so
        // the whole chunk is in flight at once
        swm0[q] = d.ag_swm[ao(i)];
        v0[q] = d.ag_stats[sb + max(id0[q], 0)];
        v1[q] = d.ag_stats[sb + max(id1[q], 0)];
        vf[q] = d.ag_stats[sb + max(s_failed, 0)];
        vm[q] = d.ag_stats[sb + max(s_max, 0)];
        tw[q] = d.ag_touched[ao(i) * d.NSW + (max(s_max, 0) >> 5)];
      }
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int i = i0 + q;
        if (i >= A || !((res0[q] | res1[q]) & 1)) continue;
        const int li = i * MGX_WORLD_EPG + lane;
        const size_t sb = ao(i) * d.NSP;
        // replay the calls: a call that did not move increments the counter and may raise the max stat
        uint32_t swm = swm0[q];
        float mx = vm[q];
        bool set_max = false;
#pragma unroll
        for (int call = 0; call < 2; call++) {
          const uint32_t r = call ? res1[q] : res0[q];
          if (!(r & 1)) continue;
          if (r & 16) swm = 0;
          else { swm += 1; if ((float)swm > mx) { mx = (float)swm; set_max = true; } }
        }
        d.ag_swm[ao(i)] = swm;
        d.ag_prev[ao(i)] = AL().prev[li];
        if (set_max && s_max >= 0) {
          d.ag_stats[sb + s_max] = mx;
          d.ag_touched[ao(i) * d.NSW + (s_max >> 5)] = tw[q] | (1u << (s_max & 31));
        }
        if (id0[q] >= 0) d.ag_stats[sb + id0[q]] = __fadd_rn(v0[q], 1.f);
        if (id1[q] >= 0) d.ag_stats[sb + id1[q]] = __fadd_rn(v1[q], 1.f);
        if (nfail[q] && s_failed >= 0) {
          float f = __fadd_rn(vf[q], 1.f);
          if (nfail[q] > 1) f = __fadd_rn(f, 1.f);
          d.ag_stats[sb + s_failed] = f;
        }
      }
    }
  }

  // d.shadow: bookkeeping_flush + track_coverage_all in ONE pass over the agents, with the counters of the bookkeeping kept
  // as integers in the agent's 32-byte ag_cnt record instead of six cells of its stat row (one 128-byte line per agent
  // read and written back every step for +1.f on two of them): two 16-byte loads and stores.  The float cells — and
  // cell.unique_visited / cell.max_distance_from_spawn, which ag_unique / ag_maxdist already hold — are written by
  // mgx_shadow_flush_kernel before anything reads them.  (float)min(count, 2^24) is what count additions of 1.f give.
  __device__ MGX_BIG void tail_shadow(int a_lo = 0, int a_hi = 1 << 30) const {   // agents [a_lo, min(a_hi, A))
    const int A = min(d.A, a_hi), lane = AL().lane;
    const int AS = d.A;   // stride of the vibe-stream result bytes
    const uint4* __restrict__ cnt = (const uint4*)d.ag_cnt;
    uint4* __restrict__ cnt_w = (uint4*)d.ag_cnt;
    for (int i0 = a_lo; i0 < A; i0 += 8) {
      uint32_t res0[8], res1[8], swm0[8];
      uint4 c0[8], c1[8];
      uint16_t rc8[8], cov8[8], sp8[8];
      uint32_t w8[8], un8[8], md8[8];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int i = min(i0 + q, A - 1);
        const int li = i * MGX_WORLD_EPG + lane;
        res0[q] = (uint32_t)(uint16_t)AL().act[li];
        res1[q] = (uint32_t)(uint16_t)AL().act[AS * MGX_WORLD_EPG + li];
        if (i0 + q >= A) res0[q] = res1[q] = 0;
        swm0[q] = d.ag_swm[ao(i)];
        c0[q] = cnt[ao(i) * 2];
        c1[q] = cnt[ao(i) * 2 + 1];
        rc8[q] = AL().rc[li];
        cov8[q] = d.ag_covrc[ao(i)];
      }
#pragma unroll
      for (int q = 0; q < 8; q++) {   // second level, only for agents that moved
        const int i = i0 + q;
        if (i < A && rc8[q] != cov8[q]) {
          const int bit = (rc8[q] >> 8) * d.W + (rc8[q] & 0xFF);
          sp8[q] = d.ag_spawn[ao(i)];
          w8[q] = d.ag_seen[ao(i) * d.SEENW + (bit >> 5)];
          un8[q] = d.ag_unique[ao(i)];
          md8[q] = d.ag_maxdist[ao(i)];
        }
      }
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int i = i0 + q;
        if (i < A && rc8[q] != cov8[q]) {   // track_coverage (objects/agent.cpp:49-57): the position changed since the last call
          const int r = rc8[q] >> 8, c = rc8[q] & 0xFF;
          const int bit = r * d.W + c;
          d.ag_covrc[ao(i)] = rc8[q];
          if (!(w8[q] & (1u << (bit & 31)))) {
            d.ag_seen[ao(i) * d.SEENW + (bit >> 5)] = w8[q] | (1u << (bit & 31));
            d.ag_unique[ao(i)] = un8[q] + 1;
          }
          const int dist = abs((int)(sp8[q] >> 8) - r) + abs(c - (int)(sp8[q] & 0xFF));
          if ((uint32_t)dist > md8[q]) d.ag_maxdist[ao(i)] = (uint32_t)dist;
        }
        if (i >= A || !((res0[q] | res1[q]) & 1)) continue;
        const int li = i * MGX_WORLD_EPG + lane;
        // replay the calls (actions/action_handler.hpp:78-105): result bytes have bit 0 set, bit 3 = success, bit 4 = moved
        uint32_t swm = swm0[q];
        uint4 a = c0[q], b = c1[q];
#pragma unroll
        for (int call = 0; call < 2; call++) {
          const uint32_t r = call ? res1[q] : res0[q];
          if (!(r & 1)) continue;
          if (r & 16) swm = 0;
          else { swm += 1; b.w = max(b.w, swm); }
          const int kind = (r >> 1) & 3;
          const uint32_t ok = (r >> 3) & 1u, no = ok ^ 1u;
          if (kind == MGX_AK_NOOP) { a.x += ok; a.y += no; }
          else if (kind == MGX_AK_MOVE) { a.z += ok; a.w += no; }
          else { b.x += ok; b.y += no; }
          b.z += no;
        }
        d.ag_swm[ao(i)] = swm;
        d.ag_prev[ao(i)] = AL().prev[li];
        cnt_w[ao(i) * 2] = a;
        cnt_w[ao(i) * 2 + 1] = b;
      }
    }
  }

  // bookkeeping_flush for ONE agent (the lane-per-agent kernel, mgx_act.h: every lane flushes its own agent)
  __device__ MGX_BIG void bookkeeping_flush_one(int i) const {
    const int A = d.A, lane = AL().lane;
    const int s_max = d.wk[MGX_S_MAX_STEPS_WITHOUT_MOTION], s_failed = d.wk[MGX_S_ACTION_FAILED];
    const int li = i * MGX_WORLD_EPG + lane;
    const uint32_t res0 = (uint32_t)(uint16_t)AL().act[li], res1 = (uint32_t)(uint16_t)AL().act[A * MGX_WORLD_EPG + li];
    if (!((res0 | res1) & 1)) return;
    auto stat_of = [&](uint32_t r) {
      const int kind = (r >> 1) & 3;
      const int base = kind == MGX_AK_NOOP ? MGX_S_NOOP_SUCCESS : kind == MGX_AK_MOVE ? MGX_S_MOVE_SUCCESS : MGX_S_VIBE_SUCCESS;
      return (r & 1) ? d.wk[base + ((r & 8) ? 0 : 1)] : -1;
    };
    const int id0 = stat_of(res0), id1 = stat_of(res1);
    const int nfail = ((res0 & 9) == 1 ? 1 : 0) + ((res1 & 9) == 1 ? 1 : 0);
    if (d.shadow) {   // the counters as integers in the agent's ag_cnt record (see tail_shadow)
      uint32_t swm = d.ag_swm[ao(i)];
      uint4 a = ((const uint4*)d.ag_cnt)[ao(i) * 2], b = ((const uint4*)d.ag_cnt)[ao(i) * 2 + 1];
#pragma unroll
      for (int call = 0; call < 2; call++) {
        const uint32_t r = call ? res1 : res0;
        if (!(r & 1)) continue;
        if (r & 16) swm = 0;
        else { swm += 1; b.w = max(b.w, swm); }
        const int kind = (r >> 1) & 3;
        const uint32_t ok = (r >> 3) & 1u, no = ok ^ 1u;
        if (kind == MGX_AK_NOOP) { a.x += ok; a.y += no; }
        else if (kind == MGX_AK_MOVE) { a.z += ok; a.w += no; }
        else { b.x += ok; b.y += no; }
        b.z += no;
      }
      d.ag_swm[ao(i)] = swm;
      d.ag_prev[ao(i)] = AL().prev[li];
      ((uint4*)d.ag_cnt)[ao(i) * 2] = a;
      ((uint4*)d.ag_cnt)[ao(i) * 2 + 1] = b;
      return;
    }
    const size_t sb = ao(i) * d.NSP;
    uint32_t swm = d.ag_swm[ao(i)];
    const float v0 = d.ag_stats[sb + max(id0, 0)], v1 = d.ag_stats[sb + max(id1, 0)];
    const float vf = d.ag_stats[sb + max(s_failed, 0)];
    float mx = d.ag_stats[sb + max(s_max, 0)];
    const uint32_t tw = d.ag_touched[ao(i) * d.NSW + (max(s_max, 0) >> 5)];
    bool set_max = false;
#pragma unroll
    for (int call = 0; call < 2; call++) {
      const uint32_t r = call ? res1 : res0;
      if (!(r & 1)) continue;
      if (r & 16) swm = 0;
      else { swm += 1; if ((float)swm > mx) { mx = (float)swm; set_max = true; } }
    }
    d.ag_swm[ao(i)] = swm;
    d.ag_prev[ao(i)] = AL().prev[li];
    if (set_max && s_max >= 0) {
      d.ag_stats[sb + s_max] = mx;
      d.ag_touched[ao(i) * d.NSW + (s_max >> 5)] = tw | (1u << (s_max & 31));
    }
    // (id0 == id1 cannot happen: the primary stream never holds a vibe action, the vibe stream nothing else)
    if (id0 >= 0) d.ag_stats[sb + id0] = __fadd_rn(v0, 1.f);
    if (id1 >= 0) d.ag_stats[sb + id1] = __fadd_rn(v1, 1.f);
    if (nfail && s_failed >= 0) {
      float f = __fadd_rn(vf, 1.f);
      if (nfail > 1) f = __fadd_rn(f, 1.f);
      d.ag_stats[sb + s_failed] = f;
    }
  }

  // `cnt` (<= 8) consecutive generator outputs, same stream as cnt calls of rng_next: word i of the incremental twist
  // needs the OLD words i + 1 and i + 397, and none of the words this block rewrites is within 397 of another.
  __device__ __forceinline__ void rng_block(uint32_t (&r)[8], uint32_t cnt) const {
    const size_t E = (size_t)d.E;
    const int ev = envi();
    const uint32_t i0 = d.mt_idx[ev];
    uint32_t w[9], m[8];
#pragma unroll
    for (int q = 0; q < 9; q++) {
      uint32_t i = i0 + q; i = i >= 624 ? i - 624 : i;
      w[q] = (uint32_t)q <= cnt ? d.mt[i * E + ev] : 0u;
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
      uint32_t i = i0 + q + 397; i = i >= 624 ? i - 624 : i; i = i >= 624 ? i - 624 : i;
      m[q] = (uint32_t)q < cnt ? d.mt[i * E + ev] : 0u;
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
      if ((uint32_t)q < cnt) {
        uint32_t i = i0 + q; i = i >= 624 ? i - 624 : i;
        const uint32_t y = (w[q] & 0x80000000u) | (w[q + 1] & 0x7fffffffu);
        uint32_t x = m[q] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        d.mt[i * E + ev] = x;
        x ^= x >> 11;
        x ^= (x << 7) & 0x9d2c5680u;
        x ^= (x << 15) & 0xefc60000u;
        x ^= x >> 18;
        r[q] = x;
      } else {
        r[q] = 0;
      }
    }
    uint32_t i1 = i0 + cnt;
    d.mt_idx[ev] = i1 >= 624 ? i1 - 624 : i1;
  }

  // track_coverage of every agent (mettagrid_c.cpp:1054-1056).  Agents are independent here, so the loads of eight
  // of them are issued together.  Both stat keys exist since Agent::init (mgx_init_kernel calls track_coverage).
  __device__ MGX_BIG void track_coverage_all(int a_lo = 0, int a_hi = 1 << 30) const {   // agents [a_lo, min(a_hi, A))
    const int A = min(d.A, a_hi), lane = AL().lane;
    const int su = d.wk[MGX_S_CELL_UNIQUE], sm = d.wk[MGX_S_CELL_MAXDIST];
    for (int i0 = a_lo; i0 < A; i0 += 8) {
      uint16_t rc8[8], cov8[8];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int i = min(i0 + q, A - 1);
        rc8[q] = AL().rc[i * MGX_WORLD_EPG + lane];
        cov8[q] = d.ag_covrc[ao(i)];
      }
      uint16_t sp8[8];
      uint32_t w8[8], un8[8], md8[8];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int i = i0 + q;
        if (i < A && rc8[q] != cov8[q]) {
          const int bit = (rc8[q] >> 8) * d.W + (rc8[q] & 0xFF);
          sp8[q] = d.ag_spawn[ao(i)];
          w8[q] = d.ag_seen[ao(i) * d.SEENW + (bit >> 5)];
          un8[q] = d.ag_unique[ao(i)];
          md8[q] = d.ag_maxdist[ao(i)];
        }
      }
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int i = i0 + q;
        if (i < A && rc8[q] != cov8[q]) {
          const int r = rc8[q] >> 8, c = rc8[q] & 0xFF;
          const int bit = r * d.W + c;
          d.ag_covrc[ao(i)] = rc8[q];
          uint32_t uniq = un8[q];
          if (!(w8[q] & (1u << (bit & 31)))) {
            d.ag_seen[ao(i) * d.SEENW + (bit >> 5)] = w8[q] | (1u << (bit & 31));
            d.ag_unique[ao(i)] = ++uniq;
          }
          if (su >= 0) d.ag_stats[ao(i) * d.NSP + su] = (float)uniq;
          const int dist = abs((int)(sp8[q] >> 8) - r) + abs(c - (int)(sp8[q] & 0xFF));
          const uint32_t md = max(md8[q], (uint32_t)dist);
          d.ag_maxdist[ao(i)] = md;
          if (sm >= 0) d.ag_stats[ao(i) * d.NSP + sm] = (float)md;
        }
      }
    }
  }

  __device__ __forceinline__ uint32_t rng_next() const {  // incremental twist: identical stream to the batch _M_gen_rand
    uint32_t i = d.mt_idx[envi()];
    uint32_t i1 = i + 1 == 624 ? 0 : i + 1;
    uint32_t im = i + 397 >= 624 ? i + 397 - 624 : i + 397;
    size_t E = (size_t)d.E;
    uint32_t y = (d.mt[i * E + envi()] & 0x80000000u) | (d.mt[i1 * E + envi()] & 0x7fffffffu);
    uint32_t x = d.mt[im * E + envi()] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    d.mt[i * E + envi()] = x;
    d.mt_idx[envi()] = i1;
    x ^= x >> 11;
    x ^= (x << 7) & 0x9d2c5680u;
    x ^= (x << 15) & 0xefc60000u;
    x ^= x >> 18;
    return x;
  }
  __device__ __forceinline__ uint32_t rng_below(uint32_t range) const {  // bits/uniform_int_dist.h:246-270 (Lemire, 64-bit product)
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

typedef MgxEnvT<MgxGlobalProg, false> MgxEnv;
typedef MgxEnvT<MgxGlobalProg, true> MgxEnvX;

// Action ids are staged as int16.  Saturation keeps every distinction the dispatch makes: valid ids are < nact
// (<= 32 000, checked at mgx_create), ids inside the +-16 invalid-index window keep their value, anything further out
// stays further out.
__device__ __forceinline__ int16_t mgx_sat16(int32_t a) { return (int16_t)max(-32768, min(32767, a)); }

// order: LDS, [k][lane] bytes (k-major so that a wavefront access is bank-conflict free)
__device__ __forceinline__ void mgx_swap(uint8_t* order, int lane, int i, int j) {
  uint8_t a = order[i * MGX_WORLD_EPG + lane], b = order[j * MGX_WORLD_EPG + lane];
  order[i * MGX_WORLD_EPG + lane] = b;
  order[j * MGX_WORLD_EPG + lane] = a;
}

// std::shuffle (bits/stl_algo.h:3729-3795): one draw for an even n, then paired draws.  The generator outputs
// are produced eight at a time (rng_block): one round trip for the state words of the whole shuffle instead of
// three dependent ones per draw.  A Lemire rejection (probability < 1e-7) simply consumes one more output.
template <class ENV>
__device__ __forceinline__ void mgx_shuffle_order(const ENV& e, uint8_t* order, int lane, int A) {
  if (A >= 2) {
    const uint32_t ndraw = (uint32_t)A / 2;  // A even: 1 + (A - 2) / 2, A odd: (A - 1) / 2
    uint32_t j = 0;                          // draws done
    while (j < ndraw) {
      uint32_t r[8];
      const uint32_t cnt = min(8u, ndraw - j);
      e.rng_block(r, cnt);
#pragma unroll
      for (int q = 0; q < 8; q++) {
        if ((uint32_t)q < cnt) {
          const bool first_even = (A & 1) == 0 && j == 0;
          const uint32_t i = (A & 1) == 0 ? 2 * j : 2 * j + 1;  // index of the pair's first element (paired draws)
          const uint32_t sft = i + 1;
          const uint32_t range = first_even ? 2u : sft * (sft + 1);
          const unsigned long long pr = (unsigned long long)r[q] * range;
          const uint32_t low = (uint32_t)pr;
          const bool accept = low >= range || low >= (0u - range) % range;
          if (accept) {
            const uint32_t x = (uint32_t)(pr >> 32);
            if (first_even) {
              mgx_swap(order, lane, 1, (int)x);
            } else {
              mgx_swap(order, lane, (int)i, (int)(x / (sft + 1)));
              mgx_swap(order, lane, (int)i + 1, (int)(x % (sft + 1)));
            }
            j++;
          }
        }
      }
    }
  }
}

// One agent's action of one stream (the body of the dispatch loop, mettagrid_c.cpp:966-999): invalid-index bookkeeping,
// the stream / kind check, ActionHandler::handle_action and the executed / success rows.
template <class ENV, class PP>
__device__ __forceinline__ void mgx_dispatch_one(const ENV& e, const MgxDev& d, PP acts, const MgxALds& al, int ai, int stream, int lane, int repeats) {
  const int A = d.A;
  int a = al.act[(stream * A + ai) * MGX_WORLD_EPG + lane];
  if (a < 0 || a >= d.nact) {  // _handle_invalid_action :914-919
    a = (stream == 0 ? d.actions : d.vibe_actions)[e.ao(ai)];  // the raw index (the staged copy is saturated to int16)
    for (int rep = 0; rep < repeats; rep++) {
      e.astat_add(ai, d.wk[MGX_S_INVALID_INDEX], 1.f);
      if (a < 0 && a >= -MGX_INVALID_WINDOW) e.astat_add(ai, d.wk[MGX_S_INVALID_NEG_BASE] + a + MGX_INVALID_WINDOW, 1.f);
      else if (a >= d.nact && a < d.nact + MGX_INVALID_WINDOW) e.astat_add(ai, d.wk[MGX_S_INVALID_POS_BASE] + a - d.nact, 1.f);
      else e.invalid_extra(ai, a);
    }
    d.success[e.ao(ai)] = 0;
    al.act[(stream * A + ai) * MGX_WORLD_EPG + lane] = 0;  // no handle_action call: empty result byte
    return;
  }
  const bool is_vibe = acts[a * MGX_AC_WORDS + MGX_AC_KIND] == MGX_AK_VIBE;
  if (is_vibe != (stream == 1)) { al.act[(stream * A + ai) * MGX_WORLD_EPG + lane] = 0; return; }
  if (e.handle_action(ai, a, stream)) {
    d.executed[e.ao(ai)] = a;
    d.success[e.ao(ai)] = 1;
  }
}

// phases (extended variant; the lean one always runs everything): MGX_PH_ACTIONS = steps 1-6 of _step (snapshot, ++step,
// shuffle, action dispatch, timestep events, per-agent on_tick), MGX_PH_AOE = fixed AoE + territory per agent, mobile
// AoE, deferred AoE registration (the general, serial form; games whose area effects only touch their target run
// mgx_aoe_kernel instead: one lane per AGENT), MGX_PH_TAIL = game on_tick + coverage tracking.
// MGX_PH_EVENTS = timestep events + the serial per-agent on_tick loop (part of MGX_PH_ACTIONS' launch unless the dispatch
// itself ran in mgx_act_kernel).
enum { MGX_PH_ACTIONS = 1, MGX_PH_AOE = 2, MGX_PH_TAIL = 4, MGX_PH_EVENTS = 8, MGX_PH_ALL = 15 };
template <class PP, bool X>
__device__ __forceinline__ void mgx_world_body(const MgxDev& d, PP P, uint8_t* order, MgxXLds xl, MgxALds al, int lane, int env, int phases,
                                               const bool helper = false) {
  MgxEnvT<PP, X> e(d, P, env);
  e.xl = xl;
  const int A = d.A;
  MGX_TICK0();
  if constexpr (!X) phases = MGX_PH_ALL;
  const bool act = (phases & MGX_PH_ACTIONS) != 0;
  const bool evt = (phases & MGX_PH_EVENTS) != 0;
  // the agents a lane walks in the batched passes: all of them, or (MGX_WORLD_HELPERS) the first chunks for the env's own
  // lane and the rest for its helper
  constexpr bool HELP = !X && MGX_HELPERS_ON;
  const int a_split = HELP ? min(A, ((A + 1) / 2 + 7) & ~7) : A;
  const int a_lo = helper ? a_split : 0, a_hi = helper ? A : a_split;
  {   // ++current_step (:933): both lanes of a pair read the old value in the same instruction, the env's own lane stores
    const uint32_t s0 = d.step[env];
    e.step = act ? s0 + 1 : s0;
    if (act && !helper) d.step[env] = e.step;
  }
  if constexpr (X) {
    if (phases == MGX_PH_EVENTS && !(d.any_on_tick && !d.tick_in_aoe)) {  // launched behind mgx_act_kernel: most steps have no event due
      const uint32_t k = d.next_event[env];
      if (k >= (uint32_t)d.n_schedule || (uint32_t)d.P[d.sec[MGX_SEC_SCHEDULE] + k * MGX_SC_WORDS + MGX_SC_TIMESTEP] > e.step) return;
    }
  }
  if (act && !helper) {  // executed_actions / _action_success cleared (mettagrid_c.cpp:944,962-964): done here (a few 16-byte
              // stores per env) instead of two memset launches per step
    if ((A & 3) == 0) {
      uint4* ex = (uint4*)(d.executed + e.ao(0));
      for (int i = 0; i < A / 4; i++) ex[i] = make_uint4(0u, 0u, 0u, 0u);
      uint32_t* su = (uint32_t*)(d.success + e.ao(0));
      for (int i = 0; i < A / 4; i++) su[i] = 0u;
    } else {
      for (int i = 0; i < A; i++) { d.executed[e.ao(i)] = 0; d.success[e.ao(i)] = 0; }
    }
  }

  // ---- stage every agent's own state + both action streams in LDS.  The loads of one chunk are independent, so
  // they are all in flight together instead of one HBM round trip per agent inside the serial loop below. ----
  const bool want_stepprev = (d.flags & MGX_G_LAST_ACTION_MOVE) != 0;
  for (int i0 = a_lo; i0 < a_hi; i0 += 8) {
    uint16_t slot8[8], prev8[8], rc8[8], cls8[8];
    uint32_t swm8[8];
    int32_t a8[8], v8[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      int i = min(i0 + q, A - 1);
      slot8[q] = d.ag_obj[e.ao(i)];
      prev8[q] = d.ag_prev[e.ao(i)];
      swm8[q] = d.ag_swm[e.ao(i)];
      a8[q] = d.actions[e.ao(i)];
      v8[q] = d.vibe_actions[e.ao(i)];
      rc8[q] = d.ag_rc[e.ao(i)];     // (per-agent mirrors of the object row: no second round of loads behind the slot)
      cls8[q] = d.ag_cls[e.ao(i)];
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
      int i = i0 + q;
      if (i < a_hi) {
        int li = i * MGX_WORLD_EPG + lane;
        al.slot[li] = slot8[q]; al.rc[li] = rc8[q]; al.prev[li] = prev8[q]; al.swm[li] = swm8[q];
        if (al.cls) al.cls[li] = cls8[q];
        al.act[li] = mgx_sat16(a8[q]); al.act[A * MGX_WORLD_EPG + li] = mgx_sat16(v8[q]);
        if (want_stepprev && act) d.ag_stepprev[e.ao(i)] = rc8[q];  // mettagrid_c.cpp:929-931
        order[li] = (uint8_t)i;
      }
    }
  }
  e.al_ = al;
  MGX_TICK(0);
  if constexpr (HELP) {   // the helper's half of the staging is read by the env's own lane below (same wavefront: program order)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  const bool split_tick = HELP && d.tick_split != 0;   // the agents' on_tick handlers stay with their agent: the pair shares them
  if (!helper && act) mgx_shuffle_order(e, order, lane, A);
  // Action dispatch (mettagrid_c.cpp:966-999).  The reference loops over priority levels max..0 and, inside each,
  // over the primary then the vibe stream.  Every real action handler has priority 0 and the only thing the
  // higher (empty) levels do is repeat the invalid-index bookkeeping, so one pass per stream with the invalid-index
  // stats added (max_priority + 1) times is equivalent — the stat keys involved are touched by nothing else.
  PP acts = P + d.sec[MGX_SEC_ACTIONS];
  const int repeats = d.max_priority + 1;
  MGX_TICK(1);
  // MgxDev::duo (MGX_WORLD_HELPERS): the env's own lane runs the next agent a of the order and its helper lane the agent b
  // behind it IN THE SAME TRIP when their cell footprints {own cell, cell ahead} are disjoint — then nothing a's action
  // reads or writes is touched by b's (host analysis: handlers stay with actor and target), and b cannot have been moved by
  // a (a swap or a move onto b's cell shares a cell) — otherwise b waits one trip.  Every env advances through its own
  // order by one or two agents per trip; the wavefront loops until its slowest env is through.  The vibe stream only
  // writes the acting agent: always two at a time.  Game-scope stat SETs go through the per-env LDS cells of the
  // lane-per-agent kernels (position in the order, then set count: last-write-wins whatever lane ran first).
  bool duo = false;
#if MGX_HELPERS_ON
  if constexpr (HELP) {
    duo = d.duo != 0 && act;
    if (duo) {
      e.duo_on = true;
      if (!helper) {
        unsigned long long* cells = e.act_gset_cells();
        for (int k = 0; k < MGX_ACT_GSET; k++) cells[k * MGX_WORLD_EPG + lane] = 0ull;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
  }
#endif
  for (int stream = 0; stream < (act ? 2 : 0); stream++) {
    if constexpr (!X) {
      // The vibe stream of a lean game: change_vibe (actions/change_vibe.hpp:48-57) writes the acting agent's own vibe and
      // its own bookkeeping and nothing reads either before the stream ends, so the order of the agents cannot be observed
      // — one straight pass per lane over its half of the agents (this lane's and its helper's), without the trip loop's
      // pairing, fences and the call into handle_action (measured: 6 000 cycles per trip for one store).
      if (stream == 1 && d.defer_book) {
        for (int i = a_lo; i < a_hi; i++) {
          const int li = i * MGX_WORLD_EPG + lane, ri = (A + i) * MGX_WORLD_EPG + lane;
          const int a = al.act[ri];
          if (a < 0 || a >= d.nact) { mgx_dispatch_one(e, d, acts, al, i, 1, lane, repeats); continue; }   // invalid index: the general path
          if (acts[a * MGX_AC_WORDS + MGX_AC_KIND] != MGX_AK_VIBE) { al.act[ri] = 0; continue; }           // wrong stream: no call
          d.obj_vibe[e.so(al.slot[li])] = (uint8_t)acts[a * MGX_AC_WORDS + MGX_AC_ARG];
          const uint16_t rc = al.rc[li], prev = al.prev[li];   // handle_action's bookkeeping (action_handler.hpp:78-105), deferred form
          const bool moved = rc != prev;
          if (moved) { al.swm[li] = 0; al.prev[li] = rc; }
          else al.swm[li] += 1;
          al.act[ri] = (int16_t)(1 | (MGX_AK_VIBE << 1) | 8 | (moved ? 16 : 0));
          d.executed[e.ao(i)] = a;
          d.success[e.ao(i)] = 1;
        }
        MGX_TICK(3);
        continue;
      }
    }
    int q = 0;   // this env's place in its order (the same in both lanes of a pair)
    for (;;) {
      const bool more = q < A && (duo || !helper);
      if (__ballot(more) == 0ull) break;
      bool run = more && !helper;
      int k = q, adv = 1;
#if MGX_HELPERS_ON
      if constexpr (HELP) {
        if (duo) {
          const bool has_b = q + 1 < A;
          k = min(q + (helper ? 1 : 0), A - 1);
          uint32_t F = 0xFFFFFFFFu;   // own cell << 16 | cell ahead (= own cell when the action is no move)
          if (more && stream == 0) {
            const int ai = order[k * MGX_WORLD_EPG + lane];
            const uint32_t own = al.rc[ai * MGX_WORLD_EPG + lane];
            uint32_t tgt = own;
            const int a = al.act[ai * MGX_WORLD_EPG + lane];
            if (a >= 0 && a < d.nact && acts[a * MGX_AC_WORDS + MGX_AC_KIND] == MGX_AK_MOVE && d.n_move_handlers > 0) {
              const int orient = acts[a * MGX_AC_WORDS + MGX_AC_ARG];  // orientation.hpp:28-48
              const int dx = (orient == 2 || orient == 4 || orient == 6) ? -1 : (orient == 3 || orient == 5 || orient == 7) ? 1 : 0;
              const int dy = (orient == 0 || orient == 4 || orient == 5) ? -1 : (orient == 1 || orient == 6 || orient == 7) ? 1 : 0;
              const int r = (int)(own >> 8) + dy, c = (int)(own & 0xFF) + dx;
              if (r >= 0 && c >= 0 && r < d.H && c < d.W) tgt = (uint32_t)((r << 8) | c);
            }
            F = (own << 16) | tgt;
          }
          const uint32_t Fo = (uint32_t)__shfl_xor((int)F, 32);
          const uint32_t Fa = helper ? Fo : F, Fb = helper ? F : Fo;
          const bool disjoint = (Fa >> 16) != (Fb & 0xFFFFu) && (Fa & 0xFFFFu) != (Fb >> 16) && (Fa & 0xFFFFu) != (Fb & 0xFFFFu);
          const bool indep = has_b && (stream == 1 || disjoint);
          run = more && (!helper || indep);
          adv = indep ? 2 : 1;
          e.act_pos = k;
        }
      }
#endif
      if (run) mgx_dispatch_one(e, d, acts, al, order[k * MGX_WORLD_EPG + lane], stream, lane, repeats);
#if MGX_HELPERS_ON
      if constexpr (HELP) {
        if (duo) {   // what the pair wrote is visible to both before the next trip's footprints and actions
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
          __builtin_amdgcn_wave_barrier();
        }
      }
#endif
      q += adv;
    }
    MGX_TICK(2 + stream);
  }
#if MGX_HELPERS_ON
  if constexpr (HELP) {
    if (duo) {   // the surviving game-stat sets (mgx_act.h mgx_act_apply_gsets)
      if (!helper) {
        unsigned long long* cells = e.act_gset_cells();
        for (int k = 0; k < d.act_ngset; k++) {
          const unsigned long long w = cells[k * MGX_WORLD_EPG + lane];
          if (w) e.gstat_set(d.act_gset_ids[k], __uint_as_float((uint32_t)w));
        }
      }
      e.duo_on = false;
    }
  }
#endif
  if (!helper) {
  if constexpr (X) {
    if (evt && d.n_schedule > 0) e.process_events();  // mettagrid_c.cpp:1009-1011
  }
  if (d.any_on_tick && evt && !d.tick_in_aoe && !split_tick) {
    for (int i = 0; i < A; i++) {  // per-agent on_tick (mettagrid_c.cpp:1019-1024); slot and class come from LDS
      const int li = i * MGX_WORLD_EPG + lane;
      const int slot = al.slot[li];
      const int h = (al.cls ? e.cls(al.cls[li]) : e.cls_of(slot))[MGX_C_ON_TICK];
      if (h >= 0) {
        MgxCtx c = mgx_ctx(slot, slot);
        e.apply_top(h, c);
      }
    }
  }
  if constexpr (X) {
   if (phases & MGX_PH_AOE) {
    if (d.NF > 0 || d.NT > 0)
      for (int i = 0; i < A; i++) {  // mettagrid_c.cpp:1032-1035
        if (d.NF > 0) e.apply_fixed(i);
        if (d.NT > 0) e.apply_territory(i);
      }
    if (d.NM > 0) e.apply_mobile();  // :1038
    if (d.def_aoe) {                 // AOETracker::flush_deferred :1042
      int n = d.def_count[env];
      for (int k = 0; k < n; k++) e.register_aoes(d.def_aoe[(size_t)env * d.S + k]);
      d.def_count[env] = 0;
    }
   }
    if ((phases & MGX_PH_TAIL) && d.game_on_tick >= 0) {       // :1050-1052
      MgxCtx c = mgx_ctx(MGX_SLOT_NONE, MGX_SLOT_NONE);
      e.apply_top(d.game_on_tick, c);
    }
  }
  }   // !helper
  if constexpr (HELP) {   // ... and the dispatch's result bytes / positions by the helper
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  if constexpr (HELP) {
    if (d.any_on_tick && evt && split_tick) {
      for (int i = a_lo; i < a_hi; i++) {  // per-agent on_tick (mettagrid_c.cpp:1019-1024), each lane of the pair its half
        const int li = i * MGX_WORLD_EPG + lane;
        const int slot = al.slot[li];
        const int h = (al.cls ? e.cls(al.cls[li]) : e.cls_of(slot))[MGX_C_ON_TICK];
        if (h >= 0) {
          MgxCtx c = mgx_ctx(slot, slot);
          e.apply_top(h, c);
        }
      }
    }
  }
  MGX_TICK(4);
  if (!X && d.shadow == 3 && act) e.tail_shadow(a_lo, a_hi);   // (d.shadow implies d.defer_book)
  else {
    if (d.defer_book && act) e.bookkeeping_flush(a_lo, a_hi);
    if (phases & MGX_PH_TAIL) e.track_coverage_all(a_lo, a_hi);
  }
  MGX_TICK(5);
}

// PROG_LDS: the program blob is first copied into LDS (16-byte coalesced loads) and every table lookup of the
// handler VM becomes a ds_read.
template <bool PROG_LDS, bool X>
__device__ __forceinline__ void mgx_world_entry(const MgxDev& d, int prog_words, int phases = MGX_PH_ALL) {
  uint8_t* order = mgx_dyn_lds;
  const int lane = mgx_world_lane();
  const bool active = (threadIdx.x & (MGX_WAVE - 1)) < MGX_WORLD_LPW;
  const bool helper = !X && MGX_HELPERS_ON && !active;   // upper half of the wavefront (see MGX_WORLD_HELPERS)
  const int env = blockIdx.x * MGX_WORLD_EPG + lane;
  MgxXLds xl;
  xl.lane = lane;
  xl.stride = MGX_WORLD_EPG;
  const MgxALds al = mgx_world_alds(mgx_dyn_lds, d.A, lane);
  int off = ((d.A * MGX_WORLD_EPG + 15) & ~15) + ((mgx_world_alds_bytes(d.A) + 15) & ~15);
  if (X) {
    off += MGX_VM_WORDS * MGX_WORLD_EPG * 4;
    xl.def_delta = (int*)(mgx_dyn_lds + off);   // (both only exist — and are only touched — with the in-kernel AoE phase)
    xl.terr_score = (long long*)(mgx_dyn_lds + off + 28 * MGX_WORLD_EPG * 4);
    if (d.x_aoe_lds) off += 28 * MGX_WORLD_EPG * 4 + 8 * MGX_WORLD_EPG * 8;
  } else {
    xl.def_delta = nullptr;
    xl.terr_score = nullptr;
  }
#if MGX_HELPERS_ON
  off += MGX_ACT_GSET * MGX_WORLD_EPG * 8;   // the game-stat cells of the paired dispatch (mgx_world_lds_fixed): the program copy follows them
#endif
  if (PROG_LDS) {
    int32_t* lprog = (int32_t*)(mgx_dyn_lds + off);
    const int4* src = (const int4*)(d.P + d.hot_lo);  // (hot_lo: 0, or the first word of the hot range: MGX_HOT_PROG)
    int4* dst = (int4*)lprog;
    for (int i = threadIdx.x; i < prog_words / 4; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
    if ((!active && !helper) || env >= d.E) return;
    mgx_world_body<MgxLdsProg, X>(d, (MgxLdsProg)lprog, order, xl, al, lane, env, phases, helper);
  } else {
    if ((!active && !helper) || env >= d.E) return;
    mgx_world_body<MgxGlobalProg, X>(d, d.P, order, xl, al, lane, env, phases, helper);
  }
}


#ifdef MGX_GEN_HANDLERS
#ifndef MGX_GEN_HEADER
#define MGX_GEN_HEADER "mgx_handlers_gen.h"
#endif
#include MGX_GEN_HEADER
#endif
}  // namespace MGX_TU_NS
using namespace MGX_TU_NS;

// Host launcher of the non-extended world kernels (defined in mgx_world_fast.hip).
// (one copy per constant-memory slot: mgx_world_fast.hip is compiled with MGX_SLOT = 0 and 1)
#define MGX_FAST_SLOTS 2
size_t mgx_world_fast_lds_bytes(int A);   // dynamic LDS of the unit's kernels without the program copy
void mgx_launch_world_fast_s0(bool prog_lds, size_t lds, hipStream_t stream, const MgxDev& d, int prog_words);
void mgx_launch_world_fast_s1(bool prog_lds, size_t lds, hipStream_t stream, const MgxDev& d, int prog_words);
bool mgx_world_fast_set_lds_s0(size_t lds);  // raises the kernels' dynamic LDS limit (needed past 64 KB)
bool mgx_world_fast_set_lds_s1(size_t lds);
// the lane-per-agent action kernels (mgx_act_fast.hip, mgx_act_x.hip; mgx_act.h)
size_t mgx_act_fast_lds_bytes(int A, int extra);
int mgx_act_fast_epg();   // envs per workgroup of the unit
void mgx_launch_act_fast_s0(bool prog_lds, size_t lds, hipStream_t stream, const MgxDev& d, int prog_words);
bool mgx_act_fast_set_lds_s0(size_t lds);
void mgx_launch_act_x(bool prog_lds, size_t lds, hipStream_t stream, const MgxDev& d, const MgxDev* dev_copy, int prog_words);
bool mgx_act_x_set_lds(size_t lds);
size_t mgx_act_x_lds_bytes(int A, bool aoe_lds, int extra);
int mgx_act_x_epg();
// ... and of the extended one (mgx_world_x.hip)
void mgx_launch_world_x(bool prog_lds, size_t lds, hipStream_t stream, const MgxDev& d, const MgxDev* dev_copy, int prog_words, int phases);
bool mgx_world_x_set_lds(size_t lds);
size_t mgx_world_x_lds_bytes(int A, bool aoe_lds);
size_t mgx_world_x_private_bytes();
void mgx_launch_values(hipStream_t stream, const MgxDev& d, const MgxDev* dev_copy, int phase, const uint8_t* env_mask);
// lane-per-agent area effects (mgx_aoe.hip) and the host analysis that allows them
// (dev_copy: absolute section offsets, for the source-record kernel; hot_copy + prog_words: the copy whose hot sections are
// relative to the kernel's LDS program copy, or nullptr / 0 when the program stays in HBM / L2)
void mgx_launch_aoe(hipStream_t stream, const MgxDev& d, const MgxDev* dev_copy, const MgxDev* hot_copy, int prog_words);
bool mgx_aoe_is_target_local(const int32_t* program);
bool mgx_aoe_on_tick_local(const int32_t* program);  // + every per-agent on_tick handler is a leaf that only touches its own agent
bool mgx_aoe_set_lds(int nstat, int prog_words);
#ifdef __cplusplus
#include <vector>
void mgx_aoe_collect_stats(const int32_t* program, bool with_on_tick, bool with_coverage, std::vector<int16_t>& out);
#endif

#ifndef MGX_WORLD_FAST_TU
// Construction: MettaGrid ctor + _init_grid (mettagrid_c.cpp:42-191, 200-269).  One lane per env scans the class
// map in row-major order; object slot = reference object id - 1; agent index = order of appearance.
// class_maps + map_index: env e is built from map `map_index ? map_index[e] : e` of `class_maps` (the engine's per-env
// maps, or its device-resident map pool: mgx_set_map_pool).
__global__ void __launch_bounds__(MGX_WAVE) mgx_init_kernel(const MgxDev* __restrict__ dp, const uint16_t* class_maps,
                                                            const int32_t* map_index, const uint32_t* seeds,
                                                            const uint8_t* env_mask) {
  const MgxDev& d = *dp;  // per-engine copy in device memory (see mgx_world_x.hip)
  const int env = blockIdx.x * MGX_WAVE + threadIdx.x;
  if (env >= d.E) return;
  if (env_mask && !env_mask[env]) return;
  MgxEnvX e(d, d.P, env);
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
  e.gstat_touch(d.wk[MGX_S_GAME_TOKENS_WRITTEN]);
  e.gstat_touch(d.wk[MGX_S_GAME_TOKENS_DROPPED]);
  e.gstat_touch(d.wk[MGX_S_GAME_TOKENS_FREE]);
  const int HW = d.H * d.W;
  const uint16_t* cm = class_maps + (size_t)(map_index ? map_index[env] : env) * HW;
  int nobj = 0, nag = 0, nf = 0, nm = 0, nts = 0;
  for (int cellidx = 0; cellidx < HW; cellidx++) {
    int k = cm[cellidx];
    if (!k) continue;
    if (nobj >= d.S) { e.flag(8u); break; }
    int slot = nobj++;
    int cls = k - 1;
    const int32_t* C = mgx_cls(d, cls);
    int r = cellidx / d.W, c = cellidx % d.W;
    const uint16_t rc = (uint16_t)((r << 8) | c);
    d.grid[(size_t)env * HW + cellidx] = (uint16_t)(slot + 1);
    d.obj_cls[e.so(slot)] = (uint16_t)cls;
    d.obj_rc[e.so(slot)] = rc;
    d.obj_vibe[e.so(slot)] = (uint8_t)C[MGX_C_INITIAL_VIBE];
    int ai = -1;
    if (C[MGX_C_KIND] == MGX_KIND_AGENT && nag < d.A) {
      ai = nag++;
      d.ag_obj[e.ao(ai)] = (uint16_t)slot;
      d.ag_prev[e.ao(ai)] = rc;
      d.ag_spawn[e.ao(ai)] = rc;
      d.ag_rc[e.ao(ai)] = rc;
      d.ag_cls[e.ao(ai)] = (uint16_t)cls;
      d.ag_rwinfo[e.ao(ai)] = ((uint32_t)C[MGX_C_REWARD_START] & 0xFFFFu) | ((uint32_t)C[MGX_C_REWARD_COUNT] << 16);
      d.ag_stepprev[e.ao(ai)] = rc;
      d.ag_covrc[e.ao(ai)] = 0xFFFF;
      for (int q = 0; q < MGX_INVALID_EXTRA; q++) d.ag_invn[e.ao(ai) * MGX_INVALID_EXTRA + q] = 0.f;
    }
    d.obj_agent[e.so(slot)] = ai < 0 ? MGX_NO_AGENT : (uint8_t)ai;
    const int32_t* ii = d.P + d.sec[MGX_SEC_INIT_INV] + C[MGX_C_INIT_INV_START] * MGX_II_WORDS;
    for (int i = 0; i < C[MGX_C_INIT_INV_COUNT]; i++, ii += MGX_II_WORDS) {
      // objects/agent.cpp:79-84 (limits ignored, no callback, "<res>.amount" set), grid_object_factory.cpp:83-87
      e.inv_update<0>(slot, ii[MGX_II_ITEM], ii[MGX_II_AMOUNT], true, false);
      if (ai >= 0) e.astat_set(ai, d.wk[MGX_S_RES_AMOUNT_BASE] + ii[MGX_II_ITEM], (float)ii[MGX_II_AMOUNT]);
    }
    e.gstat_add(C[MGX_C_OBJECTS_STAT], 1.f);
    if (d.X) {
      if (d.obj_tags)
        for (int w = 0; w < MGX_TAG_WORDS; w++) d.obj_tags[e.so(slot) * MGX_TAG_WORDS + w] = (uint32_t)C[MGX_C_TAGS + w];
      for (int w = 0; w < MGX_TAG_WORDS && d.NL > 0; w++) {  // TagIndex::register_object (core/tag_index.cpp:9-19), ascending tag id
        for (uint32_t bits = (uint32_t)C[MGX_C_TAGS + w]; bits; bits &= bits - 1) {
          int li = e.tag_list(w * 32 + __ffs(bits) - 1);
          if (li >= 0) { uint16_t n = e.tl_count(li); e.tl_items(li)[n] = (uint16_t)slot; e.tl_count(li) = n + 1; }
        }
      }
      for (int i = 0; i < C[MGX_C_AOE_COUNT]; i++) {  // AOETracker::register_source (mettagrid_c.cpp:249-252)
        int a = C[MGX_C_AOE_START] + i;
        const int32_t* AO = d.P + d.sec[MGX_SEC_AOES] + a * MGX_AO_WORDS;
        if (AO[MGX_AO_STATIC]) {
          if (nf < d.NF) { size_t q = (size_t)env * d.NF + nf; d.fx_obj[q] = (uint16_t)slot; d.fx_aoe[q] = (uint16_t)a; d.fx_rc[q] = rc; nf++; }
          else e.flag(8u);  // more sources than the capacity sized at mgx_create / mgx_set_map_pool
        } else if (nm < d.NM) {
          size_t q = (size_t)env * d.NM + nm; d.mb_obj[q] = (uint16_t)slot; d.mb_aoe[q] = (uint16_t)a; nm++;
        } else {
          e.flag(8u);
        }
      }
      for (int i = 0; i < C[MGX_C_TERR_COUNT]; i++)  // TerritoryTracker::register_source (:254-257)
        if (nts < d.NTS) { size_t q = (size_t)env * d.NTS + nts; d.ts_obj[q] = (uint16_t)slot; d.ts_ctrl[q] = (uint16_t)(C[MGX_C_TERR_START] + i); d.ts_rc[q] = rc; nts++; }
        else e.flag(8u);
    }
  }
  d.num_objs[env] = (uint32_t)nobj;
  if (d.X) {
    if (d.NF) d.fx_count[env] = (uint16_t)nf;
    if (d.NM) d.mb_count[env] = (uint16_t)nm;
    if (d.NTS) d.ts_count[env] = (uint16_t)nts;
    for (int i = 0; i < d.A * d.NT; i++) d.terr_prev[(size_t)env * d.A * d.NT + i] = -1;
    d.next_event[env] = 0;
    // QuerySystem::compute_all (query_system.cpp:91-117, mettagrid_c.cpp:162-163)
    const int32_t* mq = d.P + d.sec[MGX_SEC_MATQ];
    for (int i = 0; i < d.n_matq; i++, mq += MGX_MQ_WORDS) {
      MgxCtx t = mgx_ctx(MGX_SLOT_NONE, MGX_SLOT_NONE);
      t.skip_trigger = true;
      int tag = mq[MGX_MQ_TAG];
      int li = e.tag_list(tag);
      if (li >= 0) {
        uint16_t* lost = e.qbuf(MgxEnvX::QB_LOST);
        int nl = e.tl_count(li);
        for (int k = 0; k < nl; k++) lost[k] = e.tl_items(li)[k];
        for (int k = 0; k < nl; k++) e.tag_clear(lost[k], tag);
      }
      MgxCtx g = mgx_ctx(MGX_SLOT_NONE, MGX_SLOT_NONE);
      int n = e.eval_query<3>(mq[MGX_MQ_QUERY], g, 0);
      uint16_t* keep = e.qbuf(MgxEnvX::QB_KEEP);
      const uint16_t* res = e.qbuf(MgxEnvX::QB_BASE);
      for (int k = 0; k < n; k++) keep[k] = res[k];
      for (int k = 0; k < n; k++) e.tag_set(keep[k], tag);
    }
  }
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

#ifndef MGX_CPU_EMU
// The same construction with one WORKGROUP per env (the kernel the engine launches; the lane-per-env kernel above is its
// serial statement and what the CPU sanitizer build runs).  An episode restart rebuilds a few dozen envs per step, and its
// time is the latency of ONE env's construction: with one lane per env that was a 0.6 ms chain of 1 024 dependent cell
// reads, with one wavefront per env (round 2) 16 passes of 64 cells, each a chain of ~8 dependent accesses (class ->
// initial inventory -> limits -> stats), launched as E single-wavefront workgroups of which a few dozen had work: 0.11 ms
// on the critical path of every step of an auto-resetting batch (measured round 4: 0.05 ms of it was launching and
// retiring the idle wavefronts — about 3 us per 1 000 of them —, 0.02 ms the object pass, the rest the first step's
// full-batch launch in the average).  Now the wavefronts of a 256-thread workgroup take the 64-cell passes
// p = wave, wave + NW, ...: (A) ballots
// count the occupied / agent cells of each pass into LDS, (B) a scan over the passes gives every pass its first object slot
// (= number of occupied cells before it, the reference's object id - 1) and agent index, (C) each lane initialises its own
// object — one chain per lane, four passes of a 32 x 32 map at a time.  What is order dependent — the registration lists of
// the extended variant (tag index, AoE sources, territory sources, in object order) and the materialized queries — is done
// by lane 0 of wavefront 0 afterwards over the objects, while the last lane of the workgroup seeds the Mersenne Twister
// (a serial recurrence of 623 steps).  Object counts per class are whole numbers: atomic float adds are exact in any order.
// The envs to build are all (env_mask == env_list == nullptr: grid = E), the masked ones (grid = E), or — episode restarts —
// the entries of a device LIST whose length the host need not know: a small fixed grid walks it (`*env_list_n` entries).
#define MGX_INIT_THREADS 256
#define MGX_INIT_MAX_PASSES 1024   // maps up to 255 x 255 = 1 017 passes of 64 cells
#define MGX_INIT_STAT_CELLS 256     // game stat ids below this are counted in LDS during construction (objects.<type>)
static __device__ void mgx_init_wave_env(const MgxDev& d, const int env, const uint16_t* class_maps, const int32_t* map_index,
                                         const uint32_t* seeds, uint16_t* s_occ, uint16_t* s_ag, int* s_tot, uint32_t* s_cnt);
// std::mt19937(seed) (bits/random.tcc seed()) of whole batches: one LANE per env, so that the 624 stores of a wavefront
// are 256 contiguous bytes each (the state is env-minor: [word][env]).  Construction of all envs and masked restarts use it
// (the in-kernel seeding of mgx_init_wave_kernel writes one 4-byte word per 64-byte line and env: 8 GB of partial-line
// writes for 65 536 envs); list-driven restarts of a few dozen envs keep the in-kernel form, which hides behind the rest.
__global__ void __launch_bounds__(256) mgx_seed_mt_kernel(const MgxDev* __restrict__ dp, const uint32_t* __restrict__ seeds,
                                                          const uint8_t* __restrict__ env_mask) {
  const MgxDev& d = *dp;
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= d.E || (env_mask && !env_mask[env])) return;
  const size_t E = (size_t)d.E;
  uint32_t x = seeds[env];
  d.mt[env] = x;
  for (uint32_t i = 1; i < 624; i++) {
    x = 1812433253u * (x ^ (x >> 30)) + i;
    d.mt[i * E + env] = x;
  }
}
__global__ void __launch_bounds__(MGX_INIT_THREADS) mgx_init_wave_kernel(const MgxDev* __restrict__ dp, const uint16_t* class_maps,
                                                                         const int32_t* map_index, const uint32_t* seeds,
                                                                         const uint8_t* env_mask, const int32_t* env_list,
                                                                         const uint32_t* env_list_n) {
  const MgxDev& d = *dp;
  __shared__ uint16_t s_occ[MGX_INIT_MAX_PASSES], s_ag[MGX_INIT_MAX_PASSES];
  __shared__ int s_tot[2];
  __shared__ uint32_t s_cnt[MGX_INIT_STAT_CELLS];
  if (env_list) {
    const int n = (int)*env_list_n;
    for (int k = blockIdx.x; k < n; k += gridDim.x) {
      mgx_init_wave_env(d, env_list[k], class_maps, map_index, seeds, s_occ, s_ag, s_tot, s_cnt);
      __syncthreads();
    }
    return;
  }
  const int env = blockIdx.x;
  if (env >= d.E) return;
  if (env_mask && !env_mask[env]) return;
  mgx_init_wave_env(d, env, class_maps, map_index, seeds, s_occ, s_ag, s_tot, s_cnt);
}
static __device__ void mgx_init_wave_env(const MgxDev& d, const int env, const uint16_t* class_maps, const int32_t* map_index,
                                         const uint32_t* seeds, uint16_t* s_occ, uint16_t* s_ag, int* s_tot, uint32_t* s_cnt) {
  MgxEnvX e(d, d.P, env);
  const size_t E = (size_t)d.E;
  const int tid = threadIdx.x, lane = tid & (MGX_WAVE - 1), wave = tid / MGX_WAVE, NW = (int)blockDim.x / MGX_WAVE;
  if (tid == 0) {
    d.mt_idx[env] = 0;
    d.step[env] = 0;
    d.err[env] = 0;
    e.gstat_touch(d.wk[MGX_S_GAME_TOKENS_WRITTEN]);
    e.gstat_touch(d.wk[MGX_S_GAME_TOKENS_DROPPED]);
    e.gstat_touch(d.wk[MGX_S_GAME_TOKENS_FREE]);
  }
  const int HW = d.H * d.W, npass = (HW + MGX_WAVE - 1) / MGX_WAVE;
  const uint16_t* cm = class_maps + (size_t)(map_index ? map_index[env] : env) * HW;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int t = tid; t < MGX_INIT_STAT_CELLS; t += (int)blockDim.x) s_cnt[t] = 0u;   // (visible behind the barriers of (A) and (B))
  // (A) occupied / agent cells per pass
  for (int p = wave; p < npass; p += NW) {
    const int cellidx = p * MGX_WAVE + lane;
    const int k = cellidx < HW ? (int)cm[cellidx] : 0;
    const unsigned long long occ = __ballot(k != 0);
    const unsigned long long agm = __ballot(k != 0 && mgx_cls(d, k ? k - 1 : 0)[MGX_C_KIND] == MGX_KIND_AGENT);
    if (lane == 0) { s_occ[p] = (uint16_t)__popcll(occ); s_ag[p] = (uint16_t)__popcll(agm); }
  }
  __threadfence_block();   // err = 0 / touched bits (thread 0) before any lane's atomicOr below
  __syncthreads();
  // (B) exclusive scan over the passes (wavefront 0, 64 passes at a time)
  if (wave == 0) {
    int co = 0, ca = 0;
    for (int p0 = 0; p0 < npass; p0 += MGX_WAVE) {
      const int p = p0 + lane;
      const int vo = p < npass ? (int)s_occ[p] : 0, va = p < npass ? (int)s_ag[p] : 0;
      int io = vo, ia = va;
      for (int o = 1; o < MGX_WAVE; o <<= 1) {
        const int to = __shfl_up(io, o), ta = __shfl_up(ia, o);
        if (lane >= o) { io += to; ia += ta; }
      }
      if (p < npass) { s_occ[p] = (uint16_t)min(co + io - vo, 0xFFFF); s_ag[p] = (uint16_t)min(ca + ia - va, 0xFFFF); }
      co += __shfl(io, MGX_WAVE - 1);
      ca += __shfl(ia, MGX_WAVE - 1);
    }
    if (lane == 0) { s_tot[0] = co; s_tot[1] = ca; }
  }
  __syncthreads();
  // (C) every lane builds the object of its cell
  bool overflow = false;
  for (int p = wave; p < npass; p += NW) {
    const int cellidx = p * MGX_WAVE + lane;
    const int k = cellidx < HW ? (int)cm[cellidx] : 0;
    const unsigned long long occ = __ballot(k != 0);
    const int32_t* C = mgx_cls(d, k ? k - 1 : 0);
    const bool agent_cell = k != 0 && C[MGX_C_KIND] == MGX_KIND_AGENT;
    const unsigned long long agm = __ballot(agent_cell);
    const int slot = (int)s_occ[p] + __popcll(occ & lt);
    const int ai_raw = (int)s_ag[p] + __popcll(agm & lt);
    if (k == 0) continue;
    if (slot >= d.S) { overflow = true; continue; }   // (the serial kernel stops at the first object that does not fit)
    const int cls = k - 1;
    const int r = cellidx / d.W, c = cellidx % d.W;
    const uint16_t rc = (uint16_t)((r << 8) | c);
    d.grid[(size_t)env * HW + cellidx] = (uint16_t)(slot + 1);
    d.obj_cls[e.so(slot)] = (uint16_t)cls;
    d.obj_rc[e.so(slot)] = rc;
    d.obj_vibe[e.so(slot)] = (uint8_t)C[MGX_C_INITIAL_VIBE];
    int ai = -1;
    if (agent_cell && ai_raw < d.A) {
      ai = ai_raw;
      d.ag_obj[e.ao(ai)] = (uint16_t)slot;
      d.ag_prev[e.ao(ai)] = rc;
      d.ag_spawn[e.ao(ai)] = rc;
      d.ag_rc[e.ao(ai)] = rc;
      d.ag_cls[e.ao(ai)] = (uint16_t)cls;
      d.ag_rwinfo[e.ao(ai)] = ((uint32_t)C[MGX_C_REWARD_START] & 0xFFFFu) | ((uint32_t)C[MGX_C_REWARD_COUNT] << 16);
      d.ag_stepprev[e.ao(ai)] = rc;
      d.ag_covrc[e.ao(ai)] = 0xFFFF;
      for (int q = 0; q < MGX_INVALID_EXTRA; q++) d.ag_invn[e.ao(ai) * MGX_INVALID_EXTRA + q] = 0.f;
    }
    d.obj_agent[e.so(slot)] = ai < 0 ? MGX_NO_AGENT : (uint8_t)ai;
    const int32_t* ii = d.P + d.sec[MGX_SEC_INIT_INV] + C[MGX_C_INIT_INV_START] * MGX_II_WORDS;
    for (int i = 0; i < C[MGX_C_INIT_INV_COUNT]; i++, ii += MGX_II_WORDS) {
      e.inv_update<0>(slot, ii[MGX_II_ITEM], ii[MGX_II_AMOUNT], true, false);
      if (ai >= 0) e.astat_set(ai, d.wk[MGX_S_RES_AMOUNT_BASE] + ii[MGX_II_ITEM], (float)ii[MGX_II_AMOUNT]);
    }
    const int os = C[MGX_C_OBJECTS_STAT];
    if (os >= 0) {
      // objects.<type> += 1 (mettagrid_c.cpp:243-245): counted in LDS and added once per stat behind the barrier.  As one
      // float atomic + one atomicOr per object in HBM — float atomics execute at the memory side on gfx950 — the counters
      // WERE the construction of a whole batch: 12 M atomics at rung 3 (8.6 ms), 52 M at rung 4 (102 ms).
      if (os < MGX_INIT_STAT_CELLS) atomicAdd(&s_cnt[os], 1u);
      else {
        atomicAdd(&d.game_stats[(size_t)env * d.NG + os], 1.f);
        atomicOr(&d.game_touched[(size_t)env * d.NGW + (os >> 5)], 1u << (os & 31));
      }
    }
    if (d.X && d.obj_tags)
      for (int w = 0; w < MGX_TAG_WORDS; w++) d.obj_tags[e.so(slot) * MGX_TAG_WORDS + w] = (uint32_t)C[MGX_C_TAGS + w];
  }
  if (__ballot(overflow) && lane == 0) atomicOr(&d.err[env], 8u);
  __threadfence_block();   // the objects written by other wavefronts are read below
  __syncthreads();
  // the object counts: whole numbers added to cells nothing else writes during construction (n additions of 1.f = + n)
  for (int t0 = 0; t0 < min(d.NG, MGX_INIT_STAT_CELLS); t0 += (int)blockDim.x) {
    const int t = t0 + tid;
    const uint32_t n = t < min(d.NG, MGX_INIT_STAT_CELLS) ? s_cnt[t] : 0u;
    if (n) d.game_stats[(size_t)env * d.NG + t] += (float)n;
    const unsigned long long m = __ballot(n != 0);   // this wavefront's 64 stat ids = two words of the key-exists bits
    if (m && lane == 0) {
      const int w = (t0 + wave * MGX_WAVE) >> 5;
      if ((uint32_t)m) d.game_touched[(size_t)env * d.NGW + w] |= (uint32_t)m;
      if ((uint32_t)(m >> 32)) d.game_touched[(size_t)env * d.NGW + w + 1] |= (uint32_t)(m >> 32);
    }
  }
  __syncthreads();   // wavefront 0 reuses the counters for the list lengths below
  const int nobj = min(s_tot[0], d.S), nag = min(s_tot[1], d.A);
  if (tid == (int)blockDim.x - 1 && seeds) {   // std::mt19937(seed): bits/random.tcc seed() — beside the registration pass below
    uint32_t x = seeds[env];   // (seeds == nullptr: mgx_seed_mt_kernel has done it)
    d.mt[env] = x;
    for (uint32_t i = 1; i < 624; i++) {
      x = 1812433253u * (x ^ (x >> 30)) + i;
      d.mt[i * E + env] = x;
    }
  }
  if (wave != 0) return;
  if (lane == 0) d.num_objs[env] = (uint32_t)nobj;
  // Registrations in object order (TagIndex, AOETracker, TerritoryTracker), 64 objects at a time: a list's entries of a chunk
  // are the lanes that have it, in lane order (ballot rank / wavefront scan), behind the entries of the chunks before —
  // what the one-lane loop over the objects gave (`slot` ascending), without its ~3 dependent HBM accesses per object and
  // tag (a rung-4 env: ~400 objects, 3 ms of every step that restarted an env).  List lengths live in LDS (s_cnt).
  int nf = 0, nm = 0, nts = 0;
  if (d.X) {
    const bool lists_in_lds = d.NL <= MGX_INIT_STAT_CELLS;
    volatile uint32_t* s_len = s_cnt;   // list lengths: written by lane 0, read by every lane in the next round (same wavefront: LDS in program order)
    for (int t = lane; t < min(d.NL, MGX_INIT_STAT_CELLS); t += MGX_WAVE) s_len[t] = 0u;
    auto excl_scan = [&](int v, int& total) {
      int inc = v;
      for (int o = 1; o < MGX_WAVE; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
      total = __shfl(inc, MGX_WAVE - 1);
      return inc - v;
    };
    for (int base = 0; base < nobj; base += MGX_WAVE) {
      const int slot = base + lane;
      const bool valid = slot < nobj;
      const int32_t* C = mgx_cls(d, valid ? d.obj_cls[e.so(slot)] : d.obj_cls[e.so(base)]);
      const uint16_t rc = valid ? d.obj_rc[e.so(slot)] : (uint16_t)0;
      for (int w = 0; w < MGX_TAG_WORDS && d.NL > 0; w++) {  // TagIndex::register_object (core/tag_index.cpp:9-19)
        const uint32_t bits = valid ? (uint32_t)C[MGX_C_TAGS + w] : 0u;
        uint32_t any = bits;   // union over the wavefront
        for (int o = 1; o < MGX_WAVE; o <<= 1) any |= (uint32_t)__shfl_xor((int)any, o);
        for (; any; any &= any - 1) {
          const int b = __ffs(any) - 1;
          const int li = e.tag_list(w * 32 + b);
          if (li < 0) continue;
          const unsigned long long m = __ballot((bits >> b) & 1u);
          if (lists_in_lds) {
            const uint32_t n0 = s_len[li];
            if ((bits >> b) & 1u) e.tl_items(li)[n0 + (uint32_t)__popcll(m & lt)] = (uint16_t)slot;
            if (lane == 0) s_len[li] = n0 + (uint32_t)__popcll(m);
          } else if (lane == 0) {   // more lists than cells: the serial form for this tag
            for (unsigned long long mm = m; mm; mm &= mm - 1) {
              const uint16_t n = e.tl_count(li);
              e.tl_items(li)[n] = (uint16_t)(base + __ffsll((long long)mm) - 1);
              e.tl_count(li) = n + 1;
            }
          }
        }
      }
      {  // AOETracker::register_source (mettagrid_c.cpp:249-252): an object's sources in class order, fixed and mobile lists apart
        const int na = valid ? C[MGX_C_AOE_COUNT] : 0;
        int lf = 0, lm = 0;
        for (int i = 0; i < na; i++) {
          if (d.P[d.sec[MGX_SEC_AOES] + (C[MGX_C_AOE_START] + i) * MGX_AO_WORDS + MGX_AO_STATIC]) lf++; else lm++;
        }
        int tf, tm;
        int pf = nf + excl_scan(lf, tf), pm = nm + excl_scan(lm, tm);
        bool over = false;
        for (int i = 0; i < na; i++) {
          const int a = C[MGX_C_AOE_START] + i;
          if (d.P[d.sec[MGX_SEC_AOES] + a * MGX_AO_WORDS + MGX_AO_STATIC]) {
            if (pf < d.NF) { const size_t q = (size_t)env * d.NF + pf; d.fx_obj[q] = (uint16_t)slot; d.fx_aoe[q] = (uint16_t)a; d.fx_rc[q] = rc; }
            else over = true;
            pf++;
          } else {
            if (pm < d.NM) { const size_t q = (size_t)env * d.NM + pm; d.mb_obj[q] = (uint16_t)slot; d.mb_aoe[q] = (uint16_t)a; }
            else over = true;
            pm++;
          }
        }
        nf = min(nf + tf, d.NF); nm = min(nm + tm, d.NM);
        // TerritoryTracker::register_source (:254-257)
        const int nt = valid ? C[MGX_C_TERR_COUNT] : 0;
        int tt;
        int pt = nts + excl_scan(nt, tt);
        for (int i = 0; i < nt; i++, pt++) {
          if (pt < d.NTS) { const size_t q = (size_t)env * d.NTS + pt; d.ts_obj[q] = (uint16_t)slot; d.ts_ctrl[q] = (uint16_t)(C[MGX_C_TERR_START] + i); d.ts_rc[q] = rc; }
          else over = true;
        }
        nts = min(nts + tt, d.NTS);
        if (over) atomicOr(&d.err[env], 8u);
      }
    }
    if (lists_in_lds)
      for (int t = lane; t < d.NL; t += MGX_WAVE) e.tl_count(t) = (uint16_t)s_len[t];
    for (int i = lane; i < d.A * d.NT; i += MGX_WAVE) d.terr_prev[(size_t)env * d.A * d.NT + i] = -1;
    __threadfence_block();   // the lists are read by lane 0 below (materialized queries)
  }
  if (lane == 0) {
    if (d.X) {
      if (d.NF) d.fx_count[env] = (uint16_t)nf;
      if (d.NM) d.mb_count[env] = (uint16_t)nm;
      if (d.NTS) d.ts_count[env] = (uint16_t)nts;
      d.next_event[env] = 0;
      const int32_t* mq = d.P + d.sec[MGX_SEC_MATQ];   // QuerySystem::compute_all (query_system.cpp:91-117)
      for (int i = 0; i < d.n_matq; i++, mq += MGX_MQ_WORDS) {
        int tag = mq[MGX_MQ_TAG];
        int li = e.tag_list(tag);
        if (li >= 0) {
          uint16_t* lost = e.qbuf(MgxEnvX::QB_LOST);
          int nl = e.tl_count(li);
          for (int k = 0; k < nl; k++) lost[k] = e.tl_items(li)[k];
          for (int k = 0; k < nl; k++) e.tag_clear(lost[k], tag);
        }
        MgxCtx g = mgx_ctx(MGX_SLOT_NONE, MGX_SLOT_NONE);
        int n = e.eval_query<3>(mq[MGX_MQ_QUERY], g, 0);
        uint16_t* keep = e.qbuf(MgxEnvX::QB_KEEP);
        const uint16_t* res = e.qbuf(MgxEnvX::QB_BASE);
        for (int k = 0; k < n; k++) keep[k] = res[k];
        for (int k = 0; k < n; k++) e.tag_set(keep[k], tag);
      }
    }
  }
  for (int ai = lane; ai < nag; ai += MGX_WAVE) {   // per-agent tail: own rows only (game-scope touches are atomic)
    e.track_coverage(ai);  // Agent::init -> reset_coverage_tracking (agent.cpp:25-28,41-47)
    const int32_t* C = e.cls_of(d.ag_obj[e.ao(ai)]);
    const int32_t* rw = d.P + d.sec[MGX_SEC_REWARDS] + C[MGX_C_REWARD_START] * MGX_RW_WORDS;
    for (int i = 0; i < C[MGX_C_REWARD_COUNT]; i++, rw += MGX_RW_WORDS) {  // systems/reward.hpp:45-53
      if (rw[MGX_RW_TOUCH_SCOPE] == 0) e.astat_touch(ai, rw[MGX_RW_TOUCH_STAT]);
      else if (rw[MGX_RW_TOUCH_SCOPE] == 1)
        atomicOr(&d.game_touched[(size_t)env * d.NGW + (rw[MGX_RW_TOUCH_STAT] >> 5)], 1u << (rw[MGX_RW_TOUCH_STAT] & 31));
    }
  }
}
#endif

// ---- episode restart on the device -----------------------------------------------------------------------------------
struct MgxRow { uint8_t* base; unsigned long long row_bytes; int fill; int pad; };  // one env-major state array
// One workgroup per env: for a masked env, fill its row of every state array in the table (what a fresh construction
// would find there) and — auto-reset mode — start its next episode: bump the episode counter and pick the next map of
// the pool, (env + episode * stride) mod pool size.
// Envs: the masked ones (grid = E) or the entries of a device list walked by a small fixed grid (see mgx_init_wave_kernel).
static __device__ void mgx_clear_rows_env(const MgxRow* rows, int n_rows, int env, uint32_t* episodes, int32_t* map_index,
                                          int pool_size, int pool_stride);
__global__ void __launch_bounds__(256) mgx_clear_rows_kernel(const MgxRow* rows, int n_rows, const uint8_t* env_mask, int E,
                                                             uint32_t* episodes, int32_t* map_index, int pool_size,
                                                             int pool_stride, const int32_t* env_list, const uint32_t* env_list_n) {
  if (env_list) {
    const int n = (int)*env_list_n;
    for (int k = blockIdx.x; k < n; k += gridDim.x) mgx_clear_rows_env(rows, n_rows, env_list[k], episodes, map_index, pool_size, pool_stride);
    return;
  }
  const int env = blockIdx.x;
  if (env >= E || !env_mask[env]) return;
  mgx_clear_rows_env(rows, n_rows, env, episodes, map_index, pool_size, pool_stride);
}
static __device__ void mgx_clear_rows_env(const MgxRow* rows, int n_rows, int env, uint32_t* episodes, int32_t* map_index,
                                          int pool_size, int pool_stride) {
  for (int k = 0; k < n_rows; k++) {
    uint8_t* row = rows[k].base + (size_t)env * rows[k].row_bytes;
    const size_t n = (size_t)rows[k].row_bytes;
    const uint8_t f = (uint8_t)rows[k].fill;
    if (((uintptr_t)row & 15) == 0 && (n & 15) == 0) {
      const uint32_t f4 = f * 0x01010101u;
      uint4* r16 = (uint4*)row;
      for (size_t i = threadIdx.x; i < n / 16; i += blockDim.x) r16[i] = make_uint4(f4, f4, f4, f4);
    } else {
      for (size_t i = threadIdx.x; i < n; i += blockDim.x) row[i] = f;
    }
  }
  if (episodes && threadIdx.x == blockDim.x - 1) {
    const uint32_t ep = episodes[env] + 1;
    episodes[env] = ep;
    if (map_index && pool_size > 0) map_index[env] = (int32_t)(((unsigned long long)env + (unsigned long long)ep * (unsigned)pool_stride) % (unsigned)pool_size);
  }
}
// Masked scatter of host-packed values (k-th packed entry belongs to env idx[k]): per-env maps / seeds / pool indices of
// the envs being restarted arrive in ONE contiguous upload each.
__global__ void mgx_scatter_maps_kernel(uint16_t* dmaps, const uint16_t* packed, const int32_t* idx, int n, int HW) {
  const int k = blockIdx.y;
  if (k >= n) return;
  uint16_t* dst = dmaps + (size_t)idx[k] * HW;
  const uint16_t* src = packed + (size_t)k * HW;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void mgx_scatter_words_kernel(uint32_t* dst, const uint32_t* packed, const int32_t* idx, int n) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) dst[idx[k]] = packed[k];
}
// bit 8q of the result is set iff byte q of w is non-zero
__device__ __forceinline__ uint32_t mgx_nonzero_bytes(uint32_t w) { return ((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu | w) & 0x80808080u; }
// End of a step in auto-reset mode: the one-off early end of every env's first episode (EarlyResetHandler,
// python/src/mettagrid/envs/early_reset_handler.py:6-22: truncations set once current_step >= the drawn step), then the
// env-is-done test of Simulation.is_done (simulator.py:145-146) -> mask of the envs to restart at the start of the next
// step (lazy auto-reset, mettagrid_puffer_env.py:299-302).  The number of done envs and the step's sequence number go to
// host-visible memory, so a host that has already synchronised with this step knows without a copy whether the restart
// launches are needed at all.
#define MGX_EPEND_THREADS 1024   // (few workgroups: every one of them takes a ticket on one address)
__global__ void __launch_bounds__(MGX_EPEND_THREADS) mgx_episode_end_kernel(const MgxDev* __restrict__ dp, const uint32_t* early_steps,
                                                              const uint32_t* episodes, uint8_t* next_mask,
                                                              uint32_t* done_count, volatile uint32_t* host_flags, uint32_t seq,
                                                              uint32_t* blocks_done, int32_t* done_list, uint32_t* done_n) {
  const MgxDev& d = *dp;
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  bool done = false;
  if (env < d.E) {
    const size_t r0 = (size_t)env * d.A;
    const bool early = early_steps && episodes[env] == 0 && early_steps[env] > 0 && d.step[env] >= early_steps[env];
    bool all_term = true, all_trunc = true;
    if ((d.A & 3) == 0 && ((((uintptr_t)d.truncations) | ((uintptr_t)d.terminals)) & 3) == 0) {   // rows of A flag bytes as 32-bit words
      uint32_t* tr = (uint32_t*)(d.truncations + r0);
      const uint32_t* te = (const uint32_t*)(d.terminals + r0);
      for (int a = 0; a < d.A / 4; a++) {
        if (early) tr[a] = 0x01010101u;
        const uint32_t t = te[a], u = early ? 0x01010101u : tr[a];
        all_term = all_term && (t & 0xFF) && (t & 0xFF00) && (t & 0xFF0000) && (t & 0xFF000000u);
        all_trunc = all_trunc && (u & 0xFF) && (u & 0xFF00) && (u & 0xFF0000) && (u & 0xFF000000u);
      }
    } else {
      if (early) for (int a = 0; a < d.A; a++) d.truncations[r0 + a] = 1;
      for (int a = 0; a < d.A; a++) { all_term = all_term && d.terminals[r0 + a]; all_trunc = all_trunc && d.truncations[r0 + a]; }
    }
    done = all_term || all_trunc;
    next_mask[env] = done ? 1 : 0;
  }
#ifdef MGX_CPU_EMU
  if (done) atomicAdd(done_count, 1u);
#else
  {
    const unsigned long long m = __ballot(done);   // one add per wavefront
    if (m && (threadIdx.x & (MGX_WAVE - 1)) == 0) atomicAdd(done_count, (uint32_t)__popcll(m));
  }
#endif
  __threadfence();
  __syncthreads();
  __shared__ uint32_t s_last, s_wsum[MGX_EPEND_THREADS / MGX_WAVE];
  if (threadIdx.x == 0) {
    const uint32_t ticket = atomicAdd(blocks_done, 1u);
    s_last = ticket == gridDim.x - 1 ? 1u : 0u;
    if (s_last) {  // the last workgroup publishes the totals
      __threadfence();
      const uint32_t n = atomicAdd(done_count, 0u);
      host_flags[1] = n;
      __threadfence_system();
      host_flags[0] = seq;
      *done_count = 0;
      *blocks_done = 0;
      if (done_n) *done_n = n;
    }
  }
  __syncthreads();
  // ... and lists the done envs in ascending order (what the restart kernels and the episode-statistics kernels walk:
  // a few dozen entries per step instead of a mask of E bytes tested by E workgroups).
  if (!s_last || !done_list) return;
#ifdef MGX_CPU_EMU
  {
    int n = 0;
    for (int k = 0; k < d.E; k++) if (next_mask[k]) done_list[n++] = k;
  }
#else
  // Thread t owns the contiguous envs [t * per, (t + 1) * per): it loads its mask bytes in one go (independent 16-byte
  // loads; the mask was written by the other workgroups of this launch, made visible by their fences and the acquire
  // fence behind the ticket), counts, and one scan over the workgroup gives it its place in the list.
  const int tid = threadIdx.x, lane = tid & (MGX_WAVE - 1), wave = tid / MGX_WAVE;
  const int NT = (int)blockDim.x, NWV = NT / MGX_WAVE;
  const int per = (((d.E + NT - 1) / NT) + 15) & ~15;   // envs per thread, a multiple of 16
  const int first = tid * per;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  uint32_t cnt = 0;
  for (int q0 = 0; q0 < per; q0 += 16) {
    const int e0 = first + q0;
    if (e0 >= d.E) break;
    if (e0 + 16 <= d.E) {
      const uint4 w = *(const uint4*)(next_mask + e0);
      cnt += __popc(mgx_nonzero_bytes(w.x)) + __popc(mgx_nonzero_bytes(w.y)) + __popc(mgx_nonzero_bytes(w.z)) + __popc(mgx_nonzero_bytes(w.w));
    } else {
      for (int q = 0; q < 16 && e0 + q < d.E; q++) cnt += next_mask[e0 + q] ? 1u : 0u;
    }
  }
  uint32_t incl = cnt;
  for (int o = 1; o < MGX_WAVE; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
  if (lane == MGX_WAVE - 1) s_wsum[wave] = incl;
  __syncthreads();
  uint32_t off = incl - cnt;
  for (int q = 0; q < wave && q < NWV; q++) off += s_wsum[q];
  if (cnt)   // (a few dozen threads per step: second pass over the own bytes, 64 envs per trip)
    for (int q0 = 0; q0 < per && first + q0 < d.E; q0 += 64) {
      const int e0 = first + q0;
      if (e0 + 64 <= d.E) {
        uint32_t w[16];
        for (int v = 0; v < 4; v++) {
          const uint4 x = *(const uint4*)(next_mask + e0 + 16 * v);
          w[4 * v] = x.x; w[4 * v + 1] = x.y; w[4 * v + 2] = x.z; w[4 * v + 3] = x.w;
        }
        for (int v = 0; v < 16; v++)
          for (uint32_t m = mgx_nonzero_bytes(w[v]); m; m &= m - 1) done_list[off++] = e0 + 4 * v + ((__ffs(m) - 1) >> 3);
      } else {
        for (int q = 0; q < 64 && e0 + q < d.E; q++) if (next_mask[e0 + q]) done_list[off++] = e0 + q;
      }
    }
#endif
}

#endif  // !MGX_WORLD_FAST_TU

#endif  // MGX_WORLD_H_
