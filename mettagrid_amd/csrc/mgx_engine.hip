// mgx_engine.hip — host side of libmgx: device state allocation, kernel launches and the C ABI of include/mgx.h.
// gfx950 only.  There is NO CPU fallback in this library: every entry point that computes runs HIP kernels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "mgx.h"
#include "mgx_device.h"
#include "mgx_obs.h"
#include "mgx_presets_gen.h"   // MgxObsShapeR3 / R4: the benchmark presets' shapes (mettagrid_amd/gen_presets.py, written at build())
#include "mgx_handlers_fp.h"   // fingerprints of the presets' handler tables (mettagrid_amd/gen_handlers.py, written at build())
#include "mgx_world.h"
#include "mgx_aoe_local.h"
#include "mgx_episode.h"


// Territory ownership map (TerritoryTracker::compute_cell_ownership, core/territory_tracker.cpp:215-252) of every cell,
// per territory type: one workgroup per env, rebuilt only for envs whose sources moved or changed tags since the last
// build (terr_dirty).  Static sources (flags) make this a once-per-episode cost; the observation kernel's aoe_mask
// token and the per-agent territory effects then read one u16 per cell instead of re-deriving the influence sums.
__global__ void __launch_bounds__(256) mgx_terr_kernel(MgxDev d) {
  MGX_KERNARG_ENTRY(d);
  extern __shared__ __align__(16) uint8_t smem[];
  const int env = blockIdx.x;
  if (!d.terr_dirty[env]) return;
  MgxEnvX e(d, d.P, env);
  e.xl.terr_score = (long long*)smem;
  e.xl.lane = (int)threadIdx.x;
  e.xl.stride = 256;
  const int HW = d.H * d.W;
  for (int ti = 0; ti < d.NT; ti++)
    for (int cellidx = (int)threadIdx.x; cellidx < HW; cellidx += 256) {
      const int owner = e.cell_owner(cellidx / d.W, cellidx % d.W, ti);
      d.terr_owner[((size_t)env * d.NT + ti) * (size_t)HW + cellidx] = owner < 0 ? (uint16_t)0xFFFF : (uint16_t)owner;
    }
  __syncthreads();
  if (threadIdx.x == blockDim.x - 1) d.terr_dirty[env] = 0;
}

// MGX_TRACE=1: name every launch on stderr and wait for it, so that a faulting kernel is the last one named.
static bool g_trace = getenv("MGX_TRACE") != nullptr;
#define MGX_TRACE_POINT(e, what)                                                   \
  do {                                                                             \
    if (g_trace) { (void)hipStreamSynchronize((e)->stream); fprintf(stderr, "[mgx] done: %s\n", what); fflush(stderr); } \
  } while (0)
// Agent::set_inventory (objects/agent.cpp:86-104) for one agent: a single work-item, the same Inventory::update code
// the world kernels use.
__global__ void mgx_set_inventory_kernel(const MgxDev* __restrict__ dp, int env, int agent, const int32_t* items,
                                         const int32_t* amounts, int n) {
  const MgxDev& d = *dp;
  MgxEnvX e(d, d.P, env);
  e.step = d.step[env];
  const int slot = d.ag_obj[e.ao(agent)];
  unsigned long long ord = d.obj_order[e.so(slot)];  // iterate a copy of the current items
  for (int k = 0; k < 16; k++) {
    const int item = (int)((ord >> (4 * k)) & 0xF);
    if (item == 0xF) break;
    e.inv_update<1>(slot, item, -(int)e.inv(slot, item));
    e.astat_set(agent, d.wk[MGX_S_RES_AMOUNT_BASE] + item, 0.f);
  }
  for (int i = 0; i < n; i++) e.inv_update<1>(slot, items[i], amounts[i] - (int)e.inv(slot, items[i]));
}

// ---- batched state digests --------------------------------------------------------------------------
// One 64-bit digest per env over exactly what the episode signature is made of — the mgx_get_objects records, every stat
// value + "key exists" flag, episode rewards, action_success, current_stat_reward, current step — so that the signature of
// all 65 536 envs costs one kernel and one 512 KB copy instead of 65 536 round trips.  FNV-1a over 32-bit words; the
// objects are hashed per slot by the lanes of a wavefront and the slot hashes chained in slot order.
__device__ __forceinline__ unsigned long long mgx_fnv(unsigned long long h, uint32_t w) {
  return (h ^ (unsigned long long)w) * 1099511628211ull;
}
#define MGX_FNV_BASIS 14695981039346656037ull
__global__ void __launch_bounds__(MGX_WAVE) mgx_digest_kernel(const MgxDev* __restrict__ dp, unsigned long long* out) {
  const MgxDev& d = *dp;
  __shared__ unsigned long long slot_hash[MGX_WAVE];
  const int env = blockIdx.x, lane = threadIdx.x;
  MgxEnvX e(d, d.P, env);
  const int n = (int)d.num_objs[env];
  unsigned long long h = MGX_FNV_BASIS;
  h = mgx_fnv(h, d.step[env]);
  h = mgx_fnv(h, (uint32_t)n);
  for (int s0 = 0; s0 < n; s0 += MGX_WAVE) {
    const int s = s0 + lane;
    unsigned long long hs = MGX_FNV_BASIS;
    if (s < n) {   // the 42 words of the slot's mgx_get_objects record
      const size_t o = e.so(s);
      const uint16_t cls = d.obj_cls[o], rc = d.obj_rc[o];
      const uint8_t oflags = d.obj_flags ? d.obj_flags[o] : 0;
      const uint8_t ag = d.obj_agent[o];
      const unsigned long long ord = d.obj_order[o];
      hs = mgx_fnv(hs, (uint32_t)(s + 1)); hs = mgx_fnv(hs, cls); hs = mgx_fnv(hs, rc >> 8); hs = mgx_fnv(hs, rc & 0xFF);
      hs = mgx_fnv(hs, d.obj_vibe[o]); hs = mgx_fnv(hs, (cls != MGX_DEAD_CLASS && !(oflags & 1)) ? 1u : 0u);
      hs = mgx_fnv(hs, ag == MGX_NO_AGENT ? 0xFFFFFFFFu : (uint32_t)ag);
      int cnt = 0;
      uint32_t items[MGX_MAX_RESOURCES];
      bool ended = false;
#pragma unroll
      for (int k = 0; k < MGX_MAX_RESOURCES; k++) {
        const int item = (int)((ord >> (4 * k)) & 0xF);
        ended = ended || item == 0xF;
        items[k] = ended ? 0xFFFFFFFFu : (uint32_t)item;
        cnt += ended ? 0 : 1;
      }
      hs = mgx_fnv(hs, (uint32_t)cnt);
#pragma unroll
      for (int k = 0; k < MGX_MAX_RESOURCES; k++) hs = mgx_fnv(hs, items[k]);
      for (int k = 0; k < MGX_MAX_RESOURCES; k++) hs = mgx_fnv(hs, k < d.R ? (uint32_t)d.obj_inv[o * MGX_INV_PITCH + k] : 0u);
      for (int k = 0; k < MGX_TAG_WORDS; k++)
        hs = mgx_fnv(hs, d.obj_tags ? d.obj_tags[o * MGX_TAG_WORDS + k]
                                     : (cls != MGX_DEAD_CLASS ? (uint32_t)mgx_cls(d, cls)[MGX_C_TAGS + k] : 0u));
    }
    slot_hash[lane] = hs;
    __syncthreads();
    if (lane == 0)
      for (int k = 0; k < MGX_WAVE && s0 + k < n; k++) { h = mgx_fnv(h, (uint32_t)slot_hash[k]); h = mgx_fnv(h, (uint32_t)(slot_hash[k] >> 32)); }
    __syncthreads();
  }
  if (lane != 0) return;
  for (int i = 0; i < d.NG; i++) {   // game stats: value bits + key exists (mgx_get_stats)
    const float v = d.game_stats[(size_t)env * d.NG + i];
    const uint32_t t = ((d.game_touched[(size_t)env * d.NGW + (i >> 5)] >> (i & 31)) & 1u) | (v != 0.f ? 1u : 0u);
    h = mgx_fnv(h, __float_as_uint(v)); h = mgx_fnv(h, t);
  }
  for (int a = 0; a < d.A; a++) {
    for (int i = 0; i < d.NS; i++) {
      const float v = d.ag_stats[e.ao(a) * d.NSP + i];
      const uint32_t t = ((d.ag_touched[e.ao(a) * d.NSW + (i >> 5)] >> (i & 31)) & 1u) | (v != 0.f ? 1u : 0u);
      h = mgx_fnv(h, __float_as_uint(v)); h = mgx_fnv(h, t);
    }
    h = mgx_fnv(h, __float_as_uint(d.episode_rewards[e.ao(a)]));
    h = mgx_fnv(h, (uint32_t)d.success[e.ao(a)]);
    {  // the per-agent mirrors of the object row (MgxDev::ag_rc, ag_cls) against the row itself
      const size_t so = e.so(d.ag_obj[e.ao(a)]);
      if (d.ag_rc[e.ao(a)] != d.obj_rc[so] || d.ag_cls[e.ao(a)] != d.obj_cls[so]) d.err[env] |= MGX_ENV_INTERNAL;
    }
    const uint16_t c = d.obj_cls[e.so(d.ag_obj[e.ao(a)])];   // current_stat_reward (systems/reward.hpp:36-42)
    const int nrw = c == MGX_DEAD_CLASS ? 0 : mgx_cls(d, c)[MGX_C_REWARD_COUNT];
    float tot = 0.f;
    for (int k = 0; k < nrw; k++) tot = __fadd_rn(tot, d.ag_rprev[e.ao(a) * d.NRW + k]);
    h = mgx_fnv(h, __float_as_uint(tot));
    for (int q = 0; q < MGX_INVALID_EXTRA; q++) {   // "action.invalid_index.<k>" keys beyond the stat columns (mgx.h)
      const float nq = d.ag_invn[e.ao(a) * MGX_INVALID_EXTRA + q];
      if (nq != 0.f) { h = mgx_fnv(h, (uint32_t)d.ag_invk[e.ao(a) * MGX_INVALID_EXTRA + q]); h = mgx_fnv(h, __float_as_uint(nq)); }
    }
  }
  out[env] = h;
}

// ---- batched object export (the replay-writer shape of grid_objects(), cpp/bindings/mettagrid_py.cpp:28-139) ----
// One workgroup per LISTED env, one thread per object slot: the slot's mgx_get_objects record (MGX_OBJ_RECORD_WORDS int32,
// layout in include/mgx.h) goes to a packed device buffer [n_envs][S][words] that the host fetches with one copy.
__global__ void __launch_bounds__(256) mgx_objects_kernel(const MgxDev* __restrict__ dp, const int32_t* __restrict__ envs, int32_t* __restrict__ out,
                                                          int32_t* __restrict__ counts) {
  const MgxDev& d = *dp;
  const int env = envs[blockIdx.x];
  MgxEnvX e(d, d.P, env);
  const int n = (int)d.num_objs[env];
  if (threadIdx.x == 0) counts[blockIdx.x] = n;
  for (int s = (int)threadIdx.x; s < n; s += (int)blockDim.x) {
    int32_t* w = out + ((size_t)blockIdx.x * d.S + s) * MGX_OBJ_RECORD_WORDS;
    const size_t o = e.so(s);
    const uint16_t cls = d.obj_cls[o], rc = d.obj_rc[o];
    const uint8_t oflags = d.obj_flags ? d.obj_flags[o] : 0;
    const uint8_t ag = d.obj_agent[o];
    const unsigned long long ord = d.obj_order[o];
    w[0] = s + 1; w[1] = cls; w[2] = rc >> 8; w[3] = rc & 0xFF; w[4] = d.obj_vibe[o];
    w[5] = (cls != MGX_DEAD_CLASS && !(oflags & 1)) ? 1 : 0;
    w[6] = ag == MGX_NO_AGENT ? -1 : (int)ag;
    int cnt = 0;
    bool ended = false;
    for (int k = 0; k < MGX_MAX_RESOURCES; k++) {
      const int item = (int)((ord >> (4 * k)) & 0xF);
      ended = ended || item == 0xF;
      w[8 + k] = ended ? -1 : item;
      cnt += ended ? 0 : 1;
    }
    w[7] = cnt;
    for (int k = 0; k < MGX_MAX_RESOURCES; k++) w[8 + MGX_MAX_RESOURCES + k] = k < d.R ? (int)d.obj_inv[o * MGX_INV_PITCH + k] : 0;
    for (int k = 0; k < MGX_TAG_WORDS; k++)
      w[8 + 2 * MGX_MAX_RESOURCES + k] = d.obj_tags ? (int32_t)d.obj_tags[o * MGX_TAG_WORDS + k]
                                                    : (cls != MGX_DEAD_CLASS ? mgx_cls(d, cls)[MGX_C_TAGS + k] : 0);
  }
}

// dense-output instances of the observation kernel (mgx_obs_box.hip)
bool mgx_launch_obs_box(hipStream_t stream, const MgxDev& dd, size_t lds, int pool_tokens, int pool_prefix, const uint8_t* mask, int blk_start,
                        int blk_words, int rewards_early, bool with_rewards, bool X, bool PL, int threads, int ew, void* box, const float* scale,
                        int C, int dtype, const int32_t* env_list, const uint32_t* env_list_n, int list_grid, int stat_passes);
bool mgx_obs_box_set_lds(size_t lds);
// token decode kernel (mgx_decode.hip)
int mgx_launch_decode(hipStream_t stream, const uint8_t* tokens, float* box, const float* scale_dev, long long rows, int T, int C, int H, int W);

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return fail(MGX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                 \
  } while (0)

struct mgx_engine {
  int num_tags = 0;  // MGX_H_NUM_TAGS
  MgxDev d{};
  MgxDev* d_dev = nullptr;     // copy of `d` in device memory for the kernels that take it by pointer (extended path)
  MgxDev d_dev_host{};         // what d_dev holds
  bool d_dev_valid = false;
  MgxDev* d_hot = nullptr;     // the extended world kernel's copy: sec[] of the hot program range relative to its LDS copy
  MgxDev d_hot_host{};
  bool d_hot_valid = false;
  int hot_lo = 0, hot_hi = 0;  // [hot_lo, hot_hi): program words of that range
  int device = 0;
  hipStream_t stream = nullptr;
  std::vector<void*> allocs;
  int64_t state_bytes = 0;
  std::vector<int32_t> prog;
  // internally owned caller-visible buffers (always allocated; bound by default)
  uint8_t *own_obs = nullptr, *own_term = nullptr, *own_trunc = nullptr;
  float* own_rew = nullptr;
  int32_t *own_act = nullptr, *own_vact = nullptr;
  // host-bound buffers (MGX_MEM_HOST)
  int mem_kind = MGX_MEM_DEVICE;
  uint8_t *h_obs = nullptr, *h_term = nullptr, *h_trunc = nullptr;
  float* h_rew = nullptr;
  int32_t *h_act = nullptr, *h_vact = nullptr;
  bool external = false;
  size_t lds_world = 0, lds_obs = 0, lds_act = 0;
  int obs_threads = MGX_OBS_THREADS, obs_ew = MGX_OBS_THREADS / MGX_WAVE;
  int obs_variant = 0;   // 0: generic observation kernel; 3: the instance specialised for the shape of BASELINE.json configs[2]
  int box_dtype = 0, box_C = 0;   // mgx_set_box_output: the observation kernel writes the dense box instead of token rows
  void* box_out = nullptr;
  int pool_tokens = 0;   // capacity of the LDS token pool (entries), including the class-tag prefix
  int pool_prefix = 0;   // entries of the per-class static tag table at the head of the pool
  bool prog_in_lds = false;
  int prog_lds_words = 0;   // program words in front of the schedule section (what the world kernels copy into LDS)
  int obs_blk_start = 0, obs_blk_words = 0;  // program block the observation kernel interprets
  bool obs_blk_lds = false;
  std::vector<int> class_list_tokens;  // worst-case per-step token list length per class (0: static class)
  bool pool_from_maps = false;         // pool sized from the class maps (no run-time object creation / tag changes)
  // max over the selected envs of the summed worst-case list lengths of the objects on their class maps
  long long list_tokens_bound(const uint16_t* class_maps, size_t first, size_t count, const uint8_t* mask) const {
    const size_t hw = (size_t)d.H * d.W;
    const int nc = (int)class_list_tokens.size();
    long long best = 0;
    for (size_t env = first; env < first + count; env++) {
      if (mask && !mask[env]) continue;
      const uint16_t* cm = class_maps + env * hw;
      long long t = 0;
      for (size_t i = 0; i < hw; i++) {
        const int k = cm[i];
        if (k > 0 && k <= nc) t += class_list_tokens[k - 1];
      }
      best = std::max(best, t);
    }
    return best;
  }
  bool verbose = false;
  bool rewards_early = false;  // reward expressions have no stat operands: evaluated beside the token-cache phase
  bool terr_fresh = false;     // the territory ownership maps were refreshed in this step and nothing behind that can change them
  bool rewards_mid = false;    // extended games: ... read nothing the observation kernel writes: evaluated during its encode phase
  int slot = 0;                // constant-memory slot of the lean world kernel (mgx_world_fast.hip, MGX_SLOT)
  uint32_t* h_act_err = nullptr;      // pinned + mapped [4]: mgx_set_joint_actions' range check: flags, lowest bad row, its value
  uint32_t* h_act_err_dev = nullptr;
  unsigned long long* d_first_bad = nullptr;   // (row << 32 | value) of the lowest bad row so far, ~0 = none
  int32_t* d_vibe_ids = nullptr;  // mgx_set_joint_actions: action index of each vibe action
  int32_t vibe_ids_host[256] = {};
  int n_vibe_ids = 0;
  bool aoe_local = false;      // area effects only touch their target: one lane per agent (mgx_aoe_kernel)
  bool aoe_prog_lds = false;   // ... with the hot program range copied into its LDS
  int hot_words = 0;           // words of the hot program range, rounded up to 4
  bool rewards_ext = false;    // reward expressions have query operands: evaluated by mgx_values_kernel after the obs kernel
  uint16_t* dmaps = nullptr;
  uint32_t* dseeds = nullptr;
  uint8_t* dmask = nullptr;
  struct Row { void* base; size_t row_bytes; int fill; };
  std::vector<Row> rows_state;  // per-env state arrays (env-major) that an episode restart clears
  MgxRow* d_rows = nullptr;     // device table: rows_state + the caller-visible terminals / truncations / rewards rows
  int n_rows = 0;
  std::vector<MgxRow> rows_host;
  // device-resident map pool + auto-reset
  uint16_t* d_pool = nullptr;   // [n_pool][H][W]
  int n_pool = 0;
  int32_t* d_map_index = nullptr;   // [E] pool map of each env's current episode
  uint32_t* d_episodes = nullptr;   // [E] episodes started by auto-reset
  uint8_t* d_next_mask = nullptr;   // [E] envs found done at the end of the last step
  uint32_t* d_early = nullptr;      // [E] step at which each env's FIRST episode ends early (desync), or nullptr
  uint32_t* d_counters = nullptr;   // [2] done count, block ticket
  uint32_t* h_flags = nullptr;      // pinned + mapped: [0] sequence number of the last finished step, [1] its done count
  uint32_t* h_flags_dev = nullptr;  // device address of h_flags
  bool auto_reset = false;
  int pool_stride = 1;
  uint32_t step_seq = 0;
  // code objects specialised for this engine's program at run time (mgx_attach_code; mettagrid_amd/jit.py)
  struct Jit {
    hipModule_t mod = nullptr;
    hipFunction_t f0 = nullptr, f1 = nullptr;   // world: the kernel for this engine's prog_in_lds; obs: with / without rewards
    void* dev_sym = nullptr;                     // world: address of the module's g_mgx_dev
    MgxDev dev_host{};                           // ... and what it holds
    bool dev_valid = false;
    int epg = 0, threads = 0;
  } jit_world, jit_obs, jit_actx;
  unsigned long long handler_fp = 0;  // FNV-1a of the handler tables (gen_handlers.py fingerprint())
  int32_t* d_done_list = nullptr;   // [E] the envs of d_next_mask in ascending order (mgx_episode_end_kernel) ...
  uint32_t* d_done_n = nullptr;     // [1] ... and how many: what the restart and episode-statistics kernels walk
  // episode-end statistics (mgx_set_episode_stats; csrc/mgx_episode.h)
  bool ep_stats = false;
  MgxEpLayout ep{};
  uint32_t* d_ep_rec = nullptr;     // [E][ep.rec_words] records of the envs that finished in the last step
  double* d_ep_partial = nullptr;   // [ceil(E / 256)][TW] chunk sums
  double* d_ep_totals = nullptr;    // [TW] batch totals since the last snapshot
  double* d_ep_snap = nullptr;      // [TW] snapshot taken by mgx_request_episode_stats
  double* h_ep_snap = nullptr;      // pinned host copy of the snapshot
  uint32_t* d_ep_ticket = nullptr;  // [1]
  uint32_t* d_ep_log = nullptr;     // [ep_log_cap][ep.log_words] per-episode records kept for mgx_drain_episode_log
  uint32_t* d_ep_log_state = nullptr;  // [2] records in the log, records dropped
  int ep_log_cap = 0;
  hipEvent_t ep_ev = nullptr;       // recorded behind the snapshot copy
  bool ep_pending = false;          // a snapshot has been requested and not fetched yet
  int ep_tw() const { return MGX_EP_TOT_HDR + 2 * (ep.NG + ep.NS); }
  unsigned long long* d_digest = nullptr;   // [E] mgx_state_digests
  int32_t* d_objs = nullptr;        // mgx_get_objects_batch: env list | counts | packed records
  size_t objs_cap = 0;              // envs the buffer holds
  float* d_scale = nullptr;         // per-feature scale of the token decode (mgx_decode_obs)
  float scale_host[256] = {};
  bool scale_valid = false;
  void* d_stage = nullptr;          // staging for contiguous uploads of restart arguments
  size_t stage_bytes = 0;
  bool profiling = false;
  bool timing_valid = false;   // a step has been recorded since profiling was switched on
  hipEvent_t out_fence = nullptr;    // mgx_wait_before_outputs: pending, consumed by the next writer of the output buffers
  hipEvent_t world_done = nullptr;   // recorded after the world-update kernels of every step (mgx_chain_world)
  mgx_engine* world_after = nullptr;
  bool world_chained = false;
        // some engine waits on world_done // this engine's world kernels wait for that engine's most recent world_done
  hipEvent_t ev[MGX_T_COUNT + 1] = {};  // boundaries of the timing segments of the most recent step (profiling only)

  template <class T>
  int alloc_env(T** p, size_t per_env, int fill = 0) {  // env-major array: remembered for episode restarts
    int rc = alloc(p, per_env * (size_t)d.E, fill);
    if (rc == MGX_OK && per_env) rows_state.push_back({(void*)*p, per_env * sizeof(T), fill});
    return rc;
  }
  template <class T>
  int alloc(T** p, size_t count, int fill = 0) {
    void* q = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return fail(MGX_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    e = hipMemsetAsync(q, fill, bytes, stream);
    if (e != hipSuccess) return fail(MGX_ERR_HIP, std::string("hipMemset: ") + hipGetErrorString(e));
    allocs.push_back(q);
    state_bytes += (int64_t)bytes;
    *p = (T*)q;
    return MGX_OK;
  }
};

// Dynamic LDS of the observation kernel for the current pool capacity (mgx_create; mgx_reset_envs when new maps need
// a larger pool).
static int size_obs_lds(mgx_engine* e) {
  const MgxDev& d = e->d;
  const int xmode = mgx_obs_xmode(d.X != 0, d.X && d.aoe_mask_feat != 0 && d.NT > 0, d.S, e->num_tags);
  // extended games with many agents per env: 512 threads, of which EW wavefronts encode (4 staging rows each).  Fewer
  // encode wavefronts = fewer rows in LDS: the largest EW of 4, 3, 2 that lets three workgroups share a CU's 160 KB is
  // taken (rung 4, T = 200: EW 4, 52.8 KB, 4.3 ms; all eight wavefronts encoding: 66.6 KB, two workgroups, 5.2 ms;
  // T = 256: EW 3, 52.6 KB, 4.8 ms against 6.1 ms for EW 4 at two workgroups; 1 024 threads 8.6 ms; 256 threads 7.25 ms;
  // the lean kernel with 512 threads 1.14 instead of 0.67 ms)
  e->obs_threads = (d.X && d.A >= 48 && !getenv("MGX_OBS_256")) ? 512 : MGX_OBS_THREADS;
  e->obs_ew = e->obs_threads / MGX_WAVE;
  auto lds_for = [&](int ew) {
    return (size_t)mgx_obs_lds_layout(d.H * d.W, d.NOFF, d.S, d.A, d.T, e->pool_tokens, xmode, d.n_obs_values,
                                      e->obs_blk_lds ? e->obs_blk_words : 0, mgx_obs_gt(d.n_obs_values, d.base, d.flags),
                                      e->rewards_early, ew).total;
  };
  if (e->obs_threads == 512) {
    e->obs_ew = 4;
    for (int ew : {4, 3, 2})
      if (lds_for(ew) <= 53760) { e->obs_ew = ew; break; }
  }
  e->lds_obs = lds_for(e->obs_ew);
  if (const char* pad = getenv("MGX_OBS_LDS_PAD")) e->lds_obs += (size_t)atoi(pad);   // (occupancy experiments: unused LDS behind the layout)
  const bool keep_jit = e->obs_variant == 9 && e->jit_obs.mod && e->lds_obs <= 64 * 1024;   // (the shape does not depend on the pool size)
  e->obs_variant = 0;
  if (keep_jit) e->obs_variant = 9;
  else if (!getenv("MGX_OBS_GENERIC")) {
    if (!d.X && e->obs_blk_lds && mgx_obs_shape_matches<MgxObsShapeR3>(d, e->obs_blk_words, (e->rewards_early ? 1 : e->rewards_mid ? 2 : 0))) e->obs_variant = 3;
    else if (!d.X && e->obs_blk_lds && mgx_obs_shape_matches<MgxObsShapeR3AnyLength>(d, e->obs_blk_words, (e->rewards_early ? 1 : e->rewards_mid ? 2 : 0))) e->obs_variant = 5;   // same shape, max_steps set
    // (an instance for the shape of configs[3], MgxObsShapeR4, was measured too: 5.07 ms against the generic kernel's 5.01 —
    // the extended kernel's time is barrier and LDS latency, not scalar arithmetic; it is not built)
  }
  if (e->lds_obs > 160 * 1024)
    return fail(MGX_ERR_PROGRAM, "map/object count too large for the LDS staging of the observation kernel");
  if (e->verbose || getenv("MGX_VERBOSE"))
    fprintf(stderr, "[mgx] obs: lds=%zu B pool=%d tokens (prefix %d) blk_lds=%d rewards_early=%d threads=%d encode wavefronts=%d (4 would need %zu B)\n",
            e->lds_obs, e->pool_tokens, e->pool_prefix, (int)e->obs_blk_lds, (e->rewards_early ? 1 : e->rewards_mid ? 2 : 0), e->obs_threads, e->obs_ew, lds_for(4));
  // The attribute is per kernel and process-wide: keep one maximum and only ever raise it, so that a second engine
  // with a smaller requirement cannot lower the limit under a live one.
  static std::mutex mu;
  static size_t cur_max_dev[64] = {};   // (function attributes are per device)
  if (e->device < 0 || e->device >= 64) return fail(MGX_ERR_BAD_ARG, "device ordinal out of range");
  size_t& cur_max = cur_max_dev[e->device];
  std::lock_guard<std::mutex> lock(mu);
  if (e->lds_obs > cur_max) {
    const void* fns[] = {(const void*)mgx_obs_kernel<true, false, true, MGX_OBS_THREADS, MGX_OBS_THREADS / MGX_WAVE, MgxObsShapeR3>,
                         (const void*)mgx_obs_kernel<false, false, true, MGX_OBS_THREADS, MGX_OBS_THREADS / MGX_WAVE, MgxObsShapeR3>,
                         (const void*)mgx_obs_kernel<true, false, true, MGX_OBS_THREADS, MGX_OBS_THREADS / MGX_WAVE, MgxObsShapeR3AnyLength>,
                         (const void*)mgx_obs_kernel<false, false, true, MGX_OBS_THREADS, MGX_OBS_THREADS / MGX_WAVE, MgxObsShapeR3AnyLength>,
                         (const void*)mgx_obs_kernel<true, false, false>, (const void*)mgx_obs_kernel<false, false, false>,
                         (const void*)mgx_obs_kernel<true, false, true>,  (const void*)mgx_obs_kernel<false, false, true>,
                         (const void*)mgx_obs_kernel<true, true, false>,  (const void*)mgx_obs_kernel<false, true, false>,
                         (const void*)mgx_obs_kernel<true, true, false, 512, 4>, (const void*)mgx_obs_kernel<false, true, false, 512, 4>,
                         (const void*)mgx_obs_kernel<true, true, false, 512, 3>, (const void*)mgx_obs_kernel<false, true, false, 512, 3>,
                         (const void*)mgx_obs_kernel<true, true, false, 512, 2>, (const void*)mgx_obs_kernel<false, true, false, 512, 2>};
    for (const void* f : fns) HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_obs));
    if (!mgx_obs_box_set_lds(e->lds_obs)) return fail(MGX_ERR_HIP, "cannot raise the dense-output observation kernels' dynamic LDS limit");
    cur_max = e->lds_obs;
  }
  return MGX_OK;
}

// Grid of a restart kernel that walks a device list whose length only the device knows: enough workgroups to fill the chip,
// each taking every grid-th entry.  n_host >= 0: the host knows the length (host-driven restarts) and launches exactly that.
struct MgxList { const int32_t* list = nullptr; const uint32_t* n = nullptr; int n_host = -1; int passes = 1; };   // passes: see mgx_obs.h (token statistics of a restart)
// Launching and retiring idle workgroups is not free (measured: ~3 us per 1 000 wavefronts), so the grid follows the number
// of envs that finished in the most recent step the host has heard of (h_flags, written by mgx_episode_end_kernel): twice
// that, at least `lo` workgroups — a wrong guess only changes how many entries each workgroup walks.
static unsigned list_grid(const mgx_engine* e, const MgxList& l, int lo, int hi) {
  if (l.n_host >= 0) return (unsigned)std::max(1, std::min(l.n_host, e->d.E));
  const int hint = e->h_flags ? (int)std::min<uint32_t>(e->h_flags[1], 1u << 20) : 0;
  return (unsigned)std::max(1, std::min(e->d.E, std::min(hi, std::max(lo, 2 * hint))));
}
template <bool X, bool PL, int NTH = MGX_OBS_THREADS, int EW = NTH / MGX_WAVE, class K = MgxObsShapeDyn>
static void launch_obs_t(mgx_engine* e, bool with_rewards, const uint8_t* mask, const MgxList& l) {
  const bool listed = !with_rewards && l.list;
  dim3 grid(listed ? list_grid(e, l, 128, 2048) : (unsigned)e->d.E), block(NTH);
  const int32_t* ll = listed ? l.list : nullptr;
  const uint32_t* ln = listed ? l.n : nullptr;
  if (listed) mask = nullptr;
  MgxDev dd = e->d;
  if (PL)  // the interpreted sections are addressed relative to their LDS copy
    for (int k = MGX_SEC_INV_FEATURES; k < MGX_SEC_WORDLIST; k++) dd.sec[k] -= e->obs_blk_start;
  if (with_rewards)
    hipLaunchKernelGGL((mgx_obs_kernel<true, X, PL, NTH, EW, K>), grid, block, e->lds_obs, e->stream, dd, e->pool_tokens, e->pool_prefix, mask, e->obs_blk_start, e->obs_blk_words, (e->rewards_early ? 1 : e->rewards_mid ? 2 : 0), (void*)nullptr, (const float*)nullptr, 0, 0, ll, ln, l.passes);
  else
    hipLaunchKernelGGL((mgx_obs_kernel<false, X, PL, NTH, EW, K>), grid, block, e->lds_obs, e->stream, dd, e->pool_tokens, e->pool_prefix, mask, e->obs_blk_start, e->obs_blk_words, (e->rewards_early ? 1 : e->rewards_mid ? 2 : 0), (void*)nullptr, (const float*)nullptr, 0, 0, ll, ln, l.passes);
}
// The observation kernel of a run-time code object: same arguments as mgx_obs_kernel, through hipModuleLaunchKernel.
static int launch_obs_jit(mgx_engine* e, bool with_rewards, const uint8_t* mask, const MgxList& l) {
  const bool listed = !with_rewards && l.list;
  MgxDev dd = e->d;
  for (int k = MGX_SEC_INV_FEATURES; k < MGX_SEC_WORDLIST; k++) dd.sec[k] -= e->obs_blk_start;   // (PL: relative to the LDS copy)
  int pool_tokens = e->pool_tokens, pool_prefix = e->pool_prefix, blk_start = e->obs_blk_start, blk_words = e->obs_blk_words;
  int rmode = e->rewards_early ? 1 : e->rewards_mid ? 2 : 0, zero = 0, passes = l.passes;
  void* box = nullptr;
  const float* scale = nullptr;
  const uint8_t* m = listed ? nullptr : mask;
  const int32_t* ll = listed ? l.list : nullptr;
  const uint32_t* ln = listed ? l.n : nullptr;
  void* params[] = {&dd, &pool_tokens, &pool_prefix, &m, &blk_start, &blk_words, &rmode, &box, &scale, &zero, &zero, &ll, &ln, &passes};
  const unsigned grid = listed ? list_grid(e, l, 128, 2048) : (unsigned)e->d.E;
  HIP_TRY(hipModuleLaunchKernel(with_rewards ? e->jit_obs.f0 : e->jit_obs.f1, grid, 1, 1, (unsigned)e->jit_obs.threads, 1, 1, (unsigned)e->lds_obs,
                                e->stream, params, nullptr));
  return MGX_OK;
}
// The lean world kernel of a run-time code object: its MgxDev lives in the module's constant memory.
static int launch_world_jit(mgx_engine* e, int prog_words) {
  mgx_engine::Jit& j = e->jit_world;
  if (!j.dev_valid || memcmp(&j.dev_host, &e->d, sizeof(MgxDev)) != 0) {
    memcpy(&j.dev_host, &e->d, sizeof(MgxDev));   // (stream-ordered: the kernels still reading the old content are ahead of the copy)
    HIP_TRY(hipMemcpyAsync(j.dev_sym, &j.dev_host, sizeof(MgxDev), hipMemcpyHostToDevice, e->stream));
    j.dev_valid = true;
  }
  void* params[] = {&prog_words};
  const unsigned grid = (unsigned)((e->d.E + j.epg - 1) / j.epg);
  HIP_TRY(hipModuleLaunchKernel(j.f0, grid, 1, 1, (unsigned)j.threads, 1, 1, (unsigned)e->lds_world, e->stream, params, nullptr));
  return MGX_OK;
}
// The extended games' lane-per-agent dispatch of a run-time code object (same geometry as mgx_launch_act_x).
static int launch_act_x_jit(mgx_engine* e, const MgxDev* dp, int prog_words) {
  int ap = 1;
  while (ap < e->d.A) ap <<= 1;
  const int epg = e->jit_actx.epg;
  void* params[] = {&dp, &prog_words};
  HIP_TRY(hipModuleLaunchKernel(e->jit_actx.f0, (unsigned)((e->d.E + epg - 1) / epg), 1, 1, (unsigned)(epg * ap), 1, 1, (unsigned)e->lds_act, e->stream,
                                params, nullptr));
  return MGX_OK;
}
// Device-memory copy of e->d, brought up to date (stream-ordered) whenever the host table changed.
static const MgxDev* dev_copy(mgx_engine* e) {
  if (!e->d_dev_valid || memcmp(&e->d_dev_host, &e->d, sizeof(MgxDev)) != 0) {
    (void)hipMemcpyAsync(e->d_dev, &e->d, sizeof(MgxDev), hipMemcpyHostToDevice, e->stream);
    memcpy(&e->d_dev_host, &e->d, sizeof(MgxDev));
    e->d_dev_valid = true;
  }
  return e->d_dev;
}
// ... and the extended world kernel's variant of it when that kernel keeps the hot program range in LDS.
static const MgxDev* dev_copy_hot(mgx_engine* e);
static const MgxDev* dev_copy_world_x(mgx_engine* e) { return e->prog_in_lds ? dev_copy_hot(e) : dev_copy(e); }
static const MgxDev* dev_copy_hot(mgx_engine* e) {
  MgxDev h = e->d;
  h.hot_lo = e->hot_lo;
  for (int k = 0; k < MGX_SEC_COUNT; k++)
    if (h.sec[k] >= e->hot_lo && h.sec[k] <= e->hot_hi && k != MGX_SEC_TAG_LISTS && k != MGX_SEC_SCHEDULE && k != MGX_SEC_CLASSES) h.sec[k] -= e->hot_lo;
  if (!e->d_hot_valid || memcmp(&e->d_hot_host, &h, sizeof(MgxDev)) != 0) {
    (void)hipMemcpyAsync(e->d_hot, &h, sizeof(MgxDev), hipMemcpyHostToDevice, e->stream);
    memcpy(&e->d_hot_host, &h, sizeof(MgxDev));
    e->d_hot_valid = true;
  }
  return e->d_hot;
}
// mgx_wait_before_outputs: whatever writes the bound output buffers next goes behind the caller's event.
static int consume_out_fence(mgx_engine* e) {
  if (e->out_fence) {
    hipEvent_t ev = e->out_fence;
    e->out_fence = nullptr;
    HIP_TRY(hipStreamWaitEvent(e->stream, ev, 0));
  }
  return MGX_OK;
}
static int launch_terr(mgx_engine* e) {  // refresh the ownership maps of the envs whose territory sources changed
  if (e->d.X && e->d.NT > 0 && e->d.terr_owner) {
    hipLaunchKernelGGL(mgx_terr_kernel, dim3(e->d.E), dim3(256), 8 * 256 * 8, e->stream, e->d);
    HIP_TRY(hipGetLastError());
  }
  return MGX_OK;
}
static int launch_obs(mgx_engine* e, bool with_rewards, const uint8_t* mask = nullptr, const MgxList& l = MgxList()) {
#ifdef MGX_CPU_EMU
  (void)with_rewards; (void)mask; (void)l;
  return launch_terr(e);  // sanitizer build: the wavefront-cooperative observation kernel is not emulated
#endif
  const bool terr_fresh = e->terr_fresh;   // (consumed here whatever happens below: a stale "fresh" would skip a needed refresh)
  e->terr_fresh = false;
  int trc = consume_out_fence(e);
  if (trc) return trc;
  if (!terr_fresh) trc = launch_terr(e);   // (mgx_step already refreshed the maps for the area-effect kernel and nothing since can have changed them)
  if (trc) return trc;
  MGX_TRACE_POINT(e, "terr kernel");
  if (e->d.obsval) mgx_launch_values(e->stream, e->d, dev_copy(e), 0, mask);
  MGX_TRACE_POINT(e, "values kernel");
  if (e->rewards_ext) with_rewards = false;
  if (e->box_dtype != MGX_BOX_OFF) {   // fused dense output (mgx_set_box_output): the BOX instances live in mgx_obs_box.hip
    MgxDev dd = e->d;
    const bool pl = !e->d.X && e->obs_blk_lds;
    if (pl)
      for (int k = MGX_SEC_INV_FEATURES; k < MGX_SEC_WORDLIST; k++) dd.sec[k] -= e->obs_blk_start;
    if (!mgx_launch_obs_box(e->stream, dd, e->lds_obs, e->pool_tokens, e->pool_prefix, mask, e->obs_blk_start, e->obs_blk_words,
                            (e->rewards_early ? 1 : e->rewards_mid ? 2 : 0), with_rewards, e->d.X != 0, pl, e->obs_threads, e->obs_ew, e->box_out, e->d_scale,
                            e->box_C, e->box_dtype, with_rewards ? nullptr : l.list, l.n, (int)list_grid(e, l, 128, 2048), l.passes))
      return fail(MGX_ERR_PROGRAM, "mgx_step: no dense-output instance of the observation kernel for this configuration");
    HIP_TRY(hipGetLastError());
    return MGX_OK;
  }
  if (e->obs_variant == 9) {   // the instance compiled for this program at run time (mgx_attach_code)
    int jrc = launch_obs_jit(e, with_rewards, mask, l);
    if (jrc) return jrc;
    return MGX_OK;
  }
  // a program whose shape equals a preset's runs that preset's instance of the kernel (shape = compile-time constants)
  if (e->obs_variant == 3) launch_obs_t<false, true, MGX_OBS_THREADS, MGX_OBS_THREADS / MGX_WAVE, MgxObsShapeR3>(e, with_rewards, mask, l);
  else if (e->obs_variant == 5) launch_obs_t<false, true, MGX_OBS_THREADS, MGX_OBS_THREADS / MGX_WAVE, MgxObsShapeR3AnyLength>(e, with_rewards, mask, l);
  else if (e->d.X && e->obs_threads == 512 && e->obs_ew == 4) launch_obs_t<true, false, 512, 4>(e, with_rewards, mask, l);
  else if (e->d.X && e->obs_threads == 512 && e->obs_ew == 3) launch_obs_t<true, false, 512, 3>(e, with_rewards, mask, l);
  else if (e->d.X && e->obs_threads == 512) launch_obs_t<true, false, 512, 2>(e, with_rewards, mask, l);
  else if (e->d.X) launch_obs_t<true, false>(e, with_rewards, mask, l);
  else if (e->obs_blk_lds) launch_obs_t<false, true>(e, with_rewards, mask, l);
  else launch_obs_t<false, false>(e, with_rewards, mask, l);
  HIP_TRY(hipGetLastError());
  return MGX_OK;
}

// MettaGrid::_init_buffers (mettagrid_c.cpp:294-319): clear the bound buffers, initial observations (action 0).
static int init_buffers(mgx_engine* e) {
  const MgxDev& d = e->d;
  size_t rows = (size_t)d.E * d.A;
  { int frc = consume_out_fence(e); if (frc) return frc; }
  HIP_TRY(hipMemsetAsync(d.terminals, 0, rows, e->stream));
  HIP_TRY(hipMemsetAsync(d.truncations, 0, rows, e->stream));
  HIP_TRY(hipMemsetAsync(d.rewards, 0, rows * 4, e->stream));
  HIP_TRY(hipMemsetAsync(d.episode_rewards, 0, rows * 4, e->stream));
  HIP_TRY(hipMemsetAsync(d.executed, 0, rows * 4, e->stream));
  int rc = launch_obs(e, false);
  if (rc) return rc;
  if (e->mem_kind == MGX_MEM_HOST) {
    HIP_TRY(hipMemcpyAsync(e->h_obs, d.obs, rows * d.T * 3, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(e->h_term, d.terminals, rows, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(e->h_trunc, d.truncations, rows, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(e->h_rew, d.rewards, rows * 4, hipMemcpyDeviceToHost, e->stream));
  }
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MGX_OK;
}

static void free_episode_stats(mgx_engine* e) {
  for (void* p : {(void*)e->d_ep_rec, (void*)e->d_ep_partial, (void*)e->d_ep_totals, (void*)e->d_ep_snap, (void*)e->d_ep_ticket,
                  (void*)e->d_ep_log, (void*)e->d_ep_log_state})
    if (p) (void)hipFree(p);
  if (e->h_ep_snap) (void)hipHostFree(e->h_ep_snap);
  if (e->ep_ev) (void)hipEventDestroy(e->ep_ev);
  e->d_ep_rec = nullptr; e->d_ep_partial = nullptr; e->d_ep_totals = nullptr; e->d_ep_snap = nullptr; e->d_ep_ticket = nullptr;
  e->d_ep_log = nullptr; e->d_ep_log_state = nullptr; e->h_ep_snap = nullptr; e->ep_ev = nullptr;
  e->ep_stats = false; e->ep_pending = false; e->ep_log_cap = 0;
}
// d.shadow: bring the float stat cells of the listed envs (or all) up to date with the integer counters (mgx_episode.h)
static int flush_shadow(mgx_engine* e, const int32_t* list, const uint32_t* list_n, unsigned grid) {
#ifndef MGX_CPU_EMU
  if (!e->d.shadow) return MGX_OK;
  hipLaunchKernelGGL(mgx_shadow_flush_kernel, dim3(grid), dim3(256), 0, e->stream, dev_copy(e), list, list_n);
  HIP_TRY(hipGetLastError());
#else
  (void)e; (void)list; (void)list_n; (void)grid;
#endif
  return MGX_OK;
}
static int flush_shadow_all(mgx_engine* e) {
  return flush_shadow(e, nullptr, nullptr, (unsigned)std::min<long long>(((long long)e->d.E * e->d.A + 255) / 256, 4096));
}
// Episode-end statistics of the envs in the done list (csrc/mgx_episode.h): records, then batch totals (+ log).
static int launch_episode_stats(mgx_engine* e) {
#ifndef MGX_CPU_EMU
  const MgxDev& d = e->d;
  MgxList dl;
  dl.n_host = -1;
  {
    int frc = flush_shadow(e, (const int32_t*)e->d_done_list, (const uint32_t*)e->d_done_n, (list_grid(e, dl, 128, 2048) * (unsigned)d.A + 255) / 256);
    if (frc) return frc;
  }
  hipLaunchKernelGGL(mgx_episode_record_kernel, dim3((list_grid(e, dl, 128, 2048) + 3) / 4), dim3(256), 0, e->stream, dev_copy(e), e->ep,
                     (const int32_t*)e->d_done_list, (const uint32_t*)e->d_done_n, e->d_ep_rec, e->d_ep_log, (const uint32_t*)e->d_ep_log_state,
                     e->ep_log_cap, (const uint32_t*)e->d_early, (const uint32_t*)e->d_episodes, (const int32_t*)e->d_map_index,
                     (const uint32_t*)e->dseeds);
  HIP_TRY(hipGetLastError());
  const int nchunks_max = (d.E + MGX_EP_CHUNK - 1) / MGX_EP_CHUNK;
  hipLaunchKernelGGL(mgx_episode_accum_kernel, dim3((unsigned)std::min(nchunks_max, 16)), dim3(256), 0, e->stream, e->ep,
                     (const uint32_t*)e->d_done_n, (const uint32_t*)e->d_ep_rec, e->d_ep_partial, e->d_ep_totals, e->d_ep_ticket,
                     e->d_ep_log_state, e->ep_log_cap, e->d_ep_log ? 1 : 0);
  HIP_TRY(hipGetLastError());
#else
  (void)e;
#endif
  return MGX_OK;
}

extern "C" {

const char* mgx_last_error(void) { return g_err.c_str(); }

// construction / episode restart: one wavefront per env on the GPU, the lane-per-env statement in the CPU sanitizer build
// (dc, maps, map_index, seeds, mask, list): `list` is an MgxList; the sanitizer build only knows masks
#ifdef MGX_CPU_EMU
#define MGX_LAUNCH_INIT(stream, dc, maps, mi, sd, mask, l) hipLaunchKernelGGL(mgx_init_kernel, dim3((d.E + MGX_WAVE - 1) / MGX_WAVE), dim3(MGX_WAVE), 0, stream, dc, maps, mi, sd, mask)
#else
#define MGX_LAUNCH_INIT(stream, dc, maps, mi, sd, mask, l)                                                                          \
  do {                                                                                                                              \
    const bool listed_ = (l).list != nullptr && (l).n_host < 0;   /* a list of unknown length: seeded inside the kernel */             \
    if (!listed_) hipLaunchKernelGGL(mgx_seed_mt_kernel, dim3((unsigned)((d.E + 255) / 256)), dim3(256), 0, stream, dc, sd, mask);    \
    /* a few listed envs: 256 threads each (latency of ONE construction); whole batches: one wavefront per env, so that as  \
       many serial registration passes as possible are in flight (rung 4, 65 536 envs: 58 ms against 265 ms) */              \
    hipLaunchKernelGGL(mgx_init_wave_kernel, dim3((l).list ? list_grid(e, (l), 128, 2048) : (unsigned)d.E),                          \
                       dim3((unsigned)(listed_ ? std::min(MGX_INIT_THREADS, std::max(MGX_WAVE, (d.H * d.W + MGX_WAVE - 1) / MGX_WAVE * MGX_WAVE)) : MGX_WAVE)), 0, \
                       stream, dc, maps, mi, listed_ ? (sd) : (const uint32_t*)nullptr, (l).list ? (const uint8_t*)nullptr : (mask),  \
                       (l).list, (l).n);                                                                                             \
  } while (0)
#endif
static std::mutex g_live_mu;
static std::vector<mgx_engine*> g_live;  // engines between mgx_create and mgx_destroy

int mgx_create(const int32_t* program, size_t program_words, const uint16_t* class_maps, const uint32_t* seeds,
               int32_t num_envs, int32_t device, mgx_engine** out) {
  if (!program || !class_maps || !seeds || !out || num_envs <= 0) return fail(MGX_ERR_BAD_ARG, "mgx_create: null/empty argument");
  if (program_words < MGX_H_WORDS || program[MGX_H_MAGIC] != MGX_MAGIC || program[MGX_H_VERSION] != MGX_VERSION ||
      (size_t)program[MGX_H_TOTAL_WORDS] != program_words)
    return fail(MGX_ERR_PROGRAM, "mgx_create: not a version-" + std::to_string(MGX_VERSION) + " mgx program");
  const int32_t* P = program;
  if (P[MGX_H_NUM_RESOURCES] > MGX_MAX_RESOURCES || P[MGX_H_NUM_AGENTS] >= 255 || P[MGX_H_HEIGHT] > 255 ||
      P[MGX_H_WIDTH] > 255 || P[MGX_H_NUM_AGENTS] < 1 || P[MGX_H_TOKEN_BASE] < 2 || P[MGX_H_TOKEN_BASE] > 256 ||
      P[MGX_H_NUM_TOKENS] < 1 || P[MGX_H_OBS_HEIGHT] > 15 || P[MGX_H_OBS_WIDTH] > 15)
    return fail(MGX_ERR_PROGRAM, "mgx_create: program exceeds engine limits (resources<=13, agents<255, map<=255x255)");
  {
    const size_t hw = (size_t)P[MGX_H_HEIGHT] * P[MGX_H_WIDTH];
    for (size_t i = 0; i < (size_t)num_envs * hw; i++)
      if (class_maps[i] > P[MGX_H_NUM_CLASSES])
        return fail(MGX_ERR_BAD_ARG, "mgx_create: class map holds id " + std::to_string(class_maps[i]) + " but the program has " +
                                         std::to_string(P[MGX_H_NUM_CLASSES]) + " classes");
  }
  HIP_TRY(hipSetDevice(device));
  mgx_engine* e = new mgx_engine();
  e->device = device;
  e->prog.assign(program, program + program_words);
  hipError_t se = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
  if (se != hipSuccess) { delete e; return fail(MGX_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(se)); }

  MgxDev& d = e->d;
  d.E = num_envs; d.H = P[MGX_H_HEIGHT]; d.W = P[MGX_H_WIDTH]; d.A = P[MGX_H_NUM_AGENTS]; d.S = P[MGX_H_MAX_OBJECTS];
  d.R = P[MGX_H_NUM_RESOURCES] > 0 ? P[MGX_H_NUM_RESOURCES] : 1;
  d.T = P[MGX_H_NUM_TOKENS];
  d.NS = P[MGX_H_NUM_AGENT_STATS]; d.NG = P[MGX_H_NUM_GAME_STATS];
  d.NSW = (d.NS + 31) / 32; d.NGW = (d.NG + 31) / 32;
  d.NSP = d.NSW * 32;
  d.SEENW = (d.H * d.W + 31) / 32;
  d.NOFF = P[MGX_H_NUM_OBS_OFFSETS];
  d.base = P[MGX_H_TOKEN_BASE];
  d.max_steps = P[MGX_H_MAX_STEPS]; d.truncates = P[MGX_H_EPISODE_TRUNCATES]; d.max_priority = P[MGX_H_MAX_PRIORITY];
  d.nact = P[MGX_H_NUM_ACTIONS]; d.flags = P[MGX_H_GLOBAL_FLAGS]; d.hp_res = P[MGX_H_HP_RESOURCE];
  if (d.nact > 32000) { mgx_destroy(e); return fail(MGX_ERR_PROGRAM, "mgx_create: more than 32000 actions"); }
  d.n_obs_values = P[MGX_H_NUM_OBS_VALUES]; d.n_move_handlers = P[MGX_H_NUM_MOVE_HANDLERS];
  for (int i = 0; i < 14; i++) d.feat[i] = P[MGX_H_FEAT_BASE + i];
  d.feat[14] = P[MGX_H_OBS_HEIGHT] >> 1;
  d.feat[15] = P[MGX_H_OBS_WIDTH] >> 1;
  for (int i = 0; i < 32; i++) d.wk[i] = P[MGX_H_STAT_BASE + i];
  for (int s = 0; s < MGX_SEC_COUNT; s++) d.sec[s] = mgx_sec_off(P, s);
  int nrw = 1;
  for (int c = 0; c < P[MGX_H_NUM_CLASSES]; c++)
    nrw = std::max(nrw, (int)P[d.sec[MGX_SEC_CLASSES] + c * MGX_C_WORDS + MGX_C_REWARD_COUNT]);
  d.NRW = nrw;
  d.any_on_tick = 0;
  for (int c = 0; c < P[MGX_H_NUM_CLASSES]; c++)
    if (P[d.sec[MGX_SEC_CLASSES] + c * MGX_C_WORDS + MGX_C_ON_TICK] >= 0) d.any_on_tick = 1;

  // ---- extended (rung 4) features: capacities come from a host scan of the class maps ----
  d.n_events = P[MGX_H_NUM_EVENTS]; d.n_schedule = P[MGX_H_NUM_SCHEDULE]; d.n_matq = P[MGX_H_NUM_MATQ];
  d.NT = P[MGX_H_NUM_TERRITORIES]; d.game_on_tick = P[MGX_H_GAME_ON_TICK]; d.NL = P[MGX_H_NUM_INDEXED_TAGS];
  d.aoe_mask_feat = P[MGX_H_FEAT_BASE + MGX_F_AOE_MASK];
  e->num_tags = P[MGX_H_NUM_TAGS];
  d.QD = std::max(1, (int)P[MGX_H_QUERY_DEPTH]) + 1;
  d.QB = 3 + 2 * (d.QD + 1);
  d.AW = (d.A + 31) / 32;
  d.SW = (d.S + 31) / 32;
  {
    const int nc = P[MGX_H_NUM_CLASSES];
    std::vector<int> cf(nc), cm(nc), ct(nc);
    bool any_aoe = false;
    for (int c = 0; c < nc; c++) {
      const int32_t* C = P + d.sec[MGX_SEC_CLASSES] + c * MGX_C_WORDS;
      for (int i = 0; i < C[MGX_C_AOE_COUNT]; i++) {
        bool st = P[d.sec[MGX_SEC_AOES] + (C[MGX_C_AOE_START] + i) * MGX_AO_WORDS + MGX_AO_STATIC] != 0;
        (st ? cf[c] : cm[c])++;
        any_aoe = true;
      }
      ct[c] = C[MGX_C_TERR_COUNT];
    }
    int nf = 0, nm = 0, nts = 0;
    const size_t hw = (size_t)d.H * d.W;
    if (any_aoe || d.NT > 0)
      for (int env = 0; env < num_envs; env++) {
        int f = 0, m = 0, t = 0;
        const uint16_t* cmap = class_maps + (size_t)env * hw;
        for (size_t i = 0; i < hw; i++) {
          int k = cmap[i];
          if (k > 0 && k <= nc) { f += cf[k - 1]; m += cm[k - 1]; t += ct[k - 1]; }
        }
        nf = std::max(nf, f); nm = std::max(nm, m); nts = std::max(nts, t);
      }
    if (P[MGX_H_SPAWNS]) {  // spawned objects may bring AoEs: leave room for one set per object slot
      int per_f = 0, per_m = 0;
      for (int c = 0; c < nc; c++) { per_f = std::max(per_f, cf[c]); per_m = std::max(per_m, cm[c]); }
      nf += per_f ? std::min<int>(d.S * per_f, 4096) : 0;
      nm += per_m ? std::min<int>(d.S * per_m, 4096) : 0;
    }
    d.NF = nf; d.NM = nm; d.NTS = nts;
    d.FW = std::max(1, (nf + 31) / 32); d.MW = std::max(1, (nm + 31) / 32);
    e->aoe_local = mgx_aoe_is_target_local(P) && !getenv("MGX_AOE_SERIAL");
    d.X = (any_aoe || d.NT > 0 || d.n_schedule > 0 || d.n_matq > 0 || d.game_on_tick >= 0 || P[MGX_H_DYNAMIC_TAGS] ||
           mgx_sec_cnt(P, MGX_SEC_QUERIES) > 0) ? 1 : 0;
  }
  const size_t E = d.E, HW = (size_t)d.H * d.W, S = d.S, A = d.A, rows = E * A;
  int rc = MGX_OK;
  int32_t* dprog = nullptr;
  uint16_t* dmaps = nullptr;
  uint32_t* dseeds = nullptr;
#define A_(call) if (rc == MGX_OK) rc = (call)
  A_(e->alloc(&dprog, program_words));
  A_(e->alloc_env(&d.grid, HW));
  A_(e->alloc_env(&d.obj_cls, S, 0xFF));
  A_(e->alloc_env(&d.obj_rc, S));
  A_(e->alloc_env(&d.obj_vibe, S));
  A_(e->alloc_env(&d.obj_agent, S, 0xFF));
  A_(e->alloc_env(&d.obj_visited, S));
  A_(e->alloc_env(&d.obj_inv, S * MGX_INV_PITCH));
  A_(e->alloc_env(&d.obj_order, S, 0xFF));
  A_(e->alloc_env(&d.num_objs, 1));
  A_(e->alloc_env(&d.ag_obj, A));
  A_(e->alloc_env(&d.ag_prev, A));
  A_(e->alloc_env(&d.ag_spawn, A));
  A_(e->alloc_env(&d.ag_rc, A));
  A_(e->alloc_env(&d.ag_rwinfo, A));
  A_(e->alloc_env(&d.ag_cls, A));
  A_(e->alloc_env(&d.ag_stepprev, A));
  A_(e->alloc_env(&d.ag_covrc, A, 0xFF));
  A_(e->alloc_env(&d.ag_invk, A * MGX_INVALID_EXTRA));
  A_(e->alloc_env(&d.ag_invn, A * MGX_INVALID_EXTRA));
  A_(e->alloc_env(&d.ag_swm, A));
  A_(e->alloc_env(&d.ag_cnt, A * 8));
  A_(e->alloc_env(&d.ag_maxdist, A));
  A_(e->alloc_env(&d.ag_unique, A));
  A_(e->alloc_env(&d.ag_seen, A * d.SEENW));
  A_(e->alloc_env(&d.ag_rprev, A * d.NRW));
  A_(e->alloc_env(&d.ag_stats, A * d.NSP));
  A_(e->alloc_env(&d.ag_touched, A * d.NSW));
  A_(e->alloc_env(&d.game_stats, (size_t)d.NG));
  A_(e->alloc_env(&d.game_touched, (size_t)d.NGW));
  A_(e->alloc(&d.step, E));
  A_(e->alloc(&d.err, E));
  A_(e->alloc_env(&d.executed, A));
  A_(e->alloc_env(&d.success, A));
  A_(e->alloc_env(&d.episode_rewards, A));
  A_(e->alloc(&d.mt, 624 * E));
  A_(e->alloc(&d.mt_idx, E));
  A_(e->alloc(&e->own_obs, rows * d.T * 3, 0xFF));
  A_(e->alloc(&e->own_term, rows));
  A_(e->alloc(&e->own_trunc, rows));
  A_(e->alloc(&e->own_rew, rows));
  A_(e->alloc(&e->own_act, rows));
  A_(e->alloc(&e->own_vact, rows));
  if (d.X) {
    if (P[MGX_H_DYNAMIC_TAGS]) A_(e->alloc_env(&d.obj_tags, S * MGX_TAG_WORDS));
    if (d.NL) { A_(e->alloc_env(&d.tl_items, d.NL * S)); A_(e->alloc_env(&d.tl_count, (size_t)d.NL)); }
    if (d.NF) { A_(e->alloc_env(&d.fx_obj, (size_t)d.NF)); A_(e->alloc_env(&d.fx_aoe, (size_t)d.NF)); A_(e->alloc_env(&d.fx_rc, (size_t)d.NF));
                A_(e->alloc_env(&d.fx_inside, A * (size_t)d.FW)); A_(e->alloc_env(&d.fx_count, 1));
                if (e->aoe_local) A_(e->alloc(&d.fx_pack, E * (size_t)d.NF)); }
    if (d.NM) { A_(e->alloc_env(&d.mb_obj, (size_t)d.NM)); A_(e->alloc_env(&d.mb_aoe, (size_t)d.NM));
                A_(e->alloc_env(&d.mb_inside, A * (size_t)d.MW)); A_(e->alloc_env(&d.mb_count, 1));
                if (e->aoe_local) A_(e->alloc(&d.mb_pack, E * (size_t)d.NM)); }
    if (d.NTS) { A_(e->alloc_env(&d.ts_obj, (size_t)d.NTS)); A_(e->alloc_env(&d.ts_ctrl, (size_t)d.NTS)); A_(e->alloc_env(&d.ts_rc, (size_t)d.NTS));
                 A_(e->alloc_env(&d.ts_count, 1)); }
    A_(e->alloc_env(&d.terr_prev, A * std::max(1, d.NT)));
    if (d.NT > 0 && d.NTS > 0) { A_(e->alloc(&d.terr_owner, E * (size_t)d.NT * HW, 0xFF)); A_(e->alloc_env(&d.terr_dirty, 1, 1)); }
    A_(e->alloc_env(&d.next_event, 1));
    A_(e->alloc_env(&d.obj_flags, S));
    if (P[MGX_H_SPAWNS]) { A_(e->alloc_env(&d.def_aoe, S)); A_(e->alloc_env(&d.def_count, 1)); }
    A_(e->alloc(&d.qws, E * d.QB * S));
    A_(e->alloc(&d.qvis, E * (d.QD + 1) * d.SW));
  }
  A_(e->alloc(&e->d_dev, 1));
  A_(e->alloc(&e->d_hot, 1));
  A_(e->alloc(&dmaps, E * HW));
  A_(e->alloc(&dseeds, E));
  A_(e->alloc(&e->dmask, E));
#undef A_
  if (rc != MGX_OK) { mgx_destroy(e); return rc; }
  d.P = dprog;
  d.obs = e->own_obs; d.terminals = e->own_term; d.truncations = e->own_trunc; d.rewards = e->own_rew;
  d.actions = e->own_act; d.vibe_actions = e->own_vact;

  e->verbose = getenv("MGX_VERBOSE") != nullptr;
  {
    const bool split = d.X && e->aoe_local && (d.NF > 0 || d.NM > 0 || d.NT > 0);
    d.tick_in_aoe = (split && d.any_on_tick && mgx_aoe_on_tick_local(P) && !getenv("MGX_TICK_SERIAL")) ? 1 : 0;
    d.cov_in_aoe = (split && d.game_on_tick < 0 && !getenv("MGX_TICK_SERIAL")) ? 1 : 0;
  }
  d.x_aoe_lds = (d.X && !(e->aoe_local && (d.NF > 0 || d.NM > 0 || d.NT > 0))) ? 1 : 0;
  if (d.X && e->aoe_local && (d.NF > 0 || d.NM > 0 || d.NT > 0)) {
    // the agent stats the lane-per-agent area-effect kernel keeps in LDS (mgx_aoe_local.h)
    std::vector<int16_t> ids;
    mgx_aoe_collect_stats(P, d.tick_in_aoe != 0, d.cov_in_aoe != 0, ids);
    if (getenv("MGX_AOE_NO_STAT_CELLS")) ids.clear();   // (debug: every stat access of the kernel goes to HBM)
    std::vector<uint8_t> map(256, 0xFF);
    for (size_t k = 0; k < ids.size(); k++) map[(size_t)ids[k]] = (uint8_t)k;
    int16_t* dids = nullptr;
    uint8_t* dmap = nullptr;
    int arc = e->alloc(&dids, std::max<size_t>(1, ids.size()));
    if (arc == MGX_OK) arc = e->alloc(&dmap, 256);
    if (arc != MGX_OK) { mgx_destroy(e); return arc; }
    hipError_t ce = hipSuccess;
    if (!ids.empty()) ce = hipMemcpyAsync(dids, ids.data(), ids.size() * 2, hipMemcpyHostToDevice, e->stream);
    if (ce == hipSuccess) ce = hipMemcpyAsync(dmap, map.data(), 256, hipMemcpyHostToDevice, e->stream);
    if (ce == hipSuccess) ce = hipStreamSynchronize(e->stream);   // `ids` / `map` are locals
    if (ce != hipSuccess) { mgx_destroy(e); return fail(MGX_ERR_HIP, std::string("aoe stat table upload: ") + hipGetErrorString(ce)); }
    d.aoe_nstat = (int)ids.size();
    d.aoe_stat_ids = dids;
    d.aoe_stat_map = dmap;
    // hot program range in the kernel's LDS when it leaves room for three workgroups per CU
    e->hot_words = ((d.sec[MGX_SEC_TAG_LISTS] - (d.sec[MGX_SEC_LIMITS] & ~3)) + 3) & ~3;
    e->aoe_prog_lds = (size_t)mgx_aoe_lds_bytes(d.aoe_nstat) + (size_t)e->hot_words * 4 <= 52 * 1024 && !getenv("MGX_AOE_PROG_HBM");
#ifdef MGX_CPU_EMU
    e->aoe_prog_lds = false;  // the LDS copy needs a workgroup barrier; the sanitizer build runs work-items one by one
#endif
    if (!mgx_aoe_set_lds(d.aoe_nstat, e->aoe_prog_lds ? e->hot_words : 0)) { mgx_destroy(e); return fail(MGX_ERR_HIP, "mgx_create: cannot raise the area-effect kernel's dynamic LDS limit"); }
    if (getenv("MGX_VERBOSE")) fprintf(stderr, "[mgx] aoe kernel: %d agent stats staged per lane, %d B of LDS per workgroup\n", d.aoe_nstat, mgx_aoe_lds_bytes(d.aoe_nstat));
  }
  e->lds_world = d.X ? mgx_world_x_lds_bytes(d.A, d.x_aoe_lds != 0) : mgx_world_fast_lds_bytes(d.A);
  // The world kernels copy the program — everything in front of the schedule, the last and only section that grows with
  // the episode length — into LDS when it leaves room for 4 (lean) / 3 (extended) workgroups per CU (160 KB LDS).
  // (Rung 4, measured: the copy at 2 workgroups per CU is slower than the program in HBM at 3: 9.8 against 8.4 ms.)
  e->prog_lds_words = (int)((d.sec[MGX_SEC_SCHEDULE] + 3) & ~3);
  if (d.X) {  // extended kernel: only the sections the handler VM walks (LIMITS .. TERR_CONTROLS); the class table — one
              // record per agent — and the tag-list index are read from HBM / L2 (mgx_world.h MGX_HOT_PROG)
    e->hot_lo = d.sec[MGX_SEC_LIMITS] & ~3;
    e->hot_hi = d.sec[MGX_SEC_TAG_LISTS];
    e->prog_lds_words = ((e->hot_hi - e->hot_lo) + 3) & ~3;
  }
  e->prog_in_lds = (size_t)e->prog_lds_words * 4 + e->lds_world <= (size_t)(d.X ? 53 : 40) * 1024;
  if (const char* o = getenv("MGX_PROG_LDS")) e->prog_in_lds = e->prog_in_lds && atoi(o) != 0;
#ifdef MGX_CPU_EMU
  e->prog_in_lds = false;  // the LDS copy needs a workgroup barrier; the sanitizer build runs work-items one by one
#endif
  if (e->prog_in_lds) e->lds_world += (size_t)e->prog_lds_words * 4;
  if (getenv("MGX_VERBOSE"))
    fprintf(stderr, "[mgx] E=%d A=%d S=%d program=%zu B world: X=%d prog_in_lds=%d lds=%zu B\n", d.E, d.A, d.S,
            program_words * 4, d.X, (int)e->prog_in_lds, e->lds_world);
  {  // per-class static tag tokens (ascending tag id, core/grid_object.cpp:181-186)
    std::vector<uint32_t> info(P[MGX_H_NUM_CLASSES]);
    std::vector<uint16_t> toks;
    for (int c = 0; c < P[MGX_H_NUM_CLASSES]; c++) {
      const int32_t* C = P + d.sec[MGX_SEC_CLASSES] + c * MGX_C_WORDS;
      uint32_t start = (uint32_t)toks.size();
      int nt = 0;
      for (int t = 0; t < 256; t++)
        if (((uint32_t)C[MGX_C_TAGS + (t >> 5)] >> (t & 31)) & 1u) { toks.push_back((uint16_t)(d.feat[MGX_F_TAG] | (t << 8))); nt++; }
      if (nt > 63 || start > 0xFFFF) { mgx_destroy(e); return fail(MGX_ERR_PROGRAM, "mgx_create: more than 63 tags on one class"); }
      info[c] = start | ((uint32_t)(C[MGX_C_GROUP] & 0xFF) << 16) | ((uint32_t)nt << 24) |
                (C[MGX_C_KIND] == MGX_KIND_AGENT ? 0x40000000u : 0u) | (C[MGX_C_STATIC] ? 0x80000000u : 0u);
    }
    if (toks.empty()) toks.push_back(0);
    uint32_t* dinfo = nullptr;
    uint16_t* dtoks = nullptr;
    rc = e->alloc(&dinfo, info.size());
    if (rc == MGX_OK) rc = e->alloc(&dtoks, toks.size());
    if (rc != MGX_OK) { mgx_destroy(e); return rc; }
    hipError_t ce = hipMemcpyAsync(dinfo, info.data(), info.size() * 4, hipMemcpyHostToDevice, e->stream);
    if (ce == hipSuccess) ce = hipMemcpyAsync(dtoks, toks.data(), toks.size() * 2, hipMemcpyHostToDevice, e->stream);
    if (ce == hipSuccess) ce = hipStreamSynchronize(e->stream);
    if (ce != hipSuccess) { mgx_destroy(e); return fail(MGX_ERR_HIP, std::string("class token upload: ") + hipGetErrorString(ce)); }
    d.cls_tokinfo = dinfo;
    d.cls_tok = dtoks;
    e->pool_prefix = (int)toks.size();
  }
  {  // LDS token pool = class-tag prefix + room for the per-step lists (objects of non-static classes).  Worst case
     // per class from the program: tags + vibe + R * digits + 2.  Without run-time object creation the class maps
     // bound it exactly (sum over the env's objects, max over envs); otherwise one worst-case list per slot.
    int digits = 1;
    for (unsigned v = 65535u / (unsigned)d.base; v > 0; v /= (unsigned)d.base) digits++;
    const int nc = P[MGX_H_NUM_CLASSES];
    e->class_list_tokens.assign(nc, 0);
    int max_per_obj = 1;
    for (int c = 0; c < nc; c++) {
      const int32_t* C = P + d.sec[MGX_SEC_CLASSES] + c * MGX_C_WORDS;
      int n = 0;
      for (int w = 0; w < MGX_TAG_WORDS; w++) n += __builtin_popcount((unsigned)C[MGX_C_TAGS + w]);
      if (!C[MGX_C_STATIC]) n += 1 + P[MGX_H_NUM_RESOURCES] * digits + (C[MGX_C_KIND] == MGX_KIND_AGENT ? 2 : 0) +
                                 P[MGX_H_NUM_MATQ_TAGS];  // + the tags materialized queries may add
      max_per_obj = std::max(max_per_obj, n);
      e->class_list_tokens[c] = C[MGX_C_STATIC] ? 0 : n;
    }
    // without run-time object creation and without tag mutations the class maps bound the lists exactly
    e->pool_from_maps = !P[MGX_H_SPAWNS] && !P[MGX_H_TAG_MUTATIONS];
    long long bound = e->pool_from_maps ? e->list_tokens_bound(class_maps, 0, (size_t)E, nullptr) : (long long)S * max_per_obj;
    e->pool_tokens = (e->pool_prefix + (int)std::min<long long>(bound, 16384) + 7) & ~7;
  }
  {  // sections INV_FEATURES..OBS_VALUES are contiguous (sections are laid out in id order); small block -> LDS copy
    const int b0 = d.sec[MGX_SEC_INV_FEATURES], b1 = d.sec[MGX_SEC_WORDLIST];
    const bool ordered = b0 <= d.sec[MGX_SEC_GV_CODE] && d.sec[MGX_SEC_GV_CODE] <= d.sec[MGX_SEC_REWARDS] &&
                         d.sec[MGX_SEC_REWARDS] <= d.sec[MGX_SEC_OBS_VALUES] && d.sec[MGX_SEC_OBS_VALUES] <= b1;
    e->obs_blk_start = b0;
    e->obs_blk_words = b1 - b0;
    e->obs_blk_lds = !d.X && ordered && (b0 & 3) == 0 && (e->obs_blk_words & 3) == 0 && e->obs_blk_words * 4 <= 8 * 1024;
  }
  {  // Per-action bookkeeping stats can be applied at the end of the tick unless some game value reads agent stats
     // (a filter or SetStat evaluated mid-tick could then see them early or late).
    bool reads_agent_stats = false;
    const int n_code = P[MGX_H_SECTION_BASE + 2 * MGX_SEC_GV_CODE + 1];
    const int32_t* code = P + d.sec[MGX_SEC_GV_CODE];
    std::vector<char> reward_only(n_code, 0);  // reward expressions run in the observation kernel, after the flush
    const int32_t* rwr = P + d.sec[MGX_SEC_REWARDS];
    for (int k = 0; k < P[MGX_H_SECTION_BASE + 2 * MGX_SEC_REWARDS + 1]; k++)
      for (int i = 0; i < rwr[k * MGX_RW_WORDS + MGX_RW_GV_COUNT]; i++) {
        const int at = rwr[k * MGX_RW_WORDS + MGX_RW_GV_START] + i;
        if (at >= 0 && at < n_code) reward_only[at] = 1;
      }
    // ... and then only when it reads one of the counters the deferred pass writes: the per-kind success / failed
    // counters, action.failed and max_steps_without_motion (a game that counts zone entries in an agent stat and adds
    // to it from a territory handler does not care when `action.move.success` is brought up to date)
    auto booked = [&](int id) {
      for (int k : {MGX_S_NOOP_SUCCESS, MGX_S_MOVE_SUCCESS, MGX_S_VIBE_SUCCESS})
        if (id == d.wk[k] || id == d.wk[k + 1]) return true;
      return id == d.wk[MGX_S_ACTION_FAILED] || id == d.wk[MGX_S_MAX_STEPS_WITHOUT_MOTION];
    };
    for (int i = 0; i < n_code; i++)
      if (!reward_only[i] && code[i * MGX_GV_WORDS + MGX_GV_OP] == MGX_GOP_STAT && code[i * MGX_GV_WORDS + MGX_GV_A0] != 1 &&
          booked(code[i * MGX_GV_WORDS + MGX_GV_A1]))
        reads_agent_stats = true;
    // ... or a handler WRITES one of them mid-tick (SetStat / a game-value mutation on a StatValue): set-then-deferred-add
    // would end on a different value than the reference's add-then-set
    bool writes_booked = false;
    const int n_mut = mgx_sec_cnt(P, MGX_SEC_MUTS);
    for (int i = 0; i < n_mut; i++) {
      const int32_t* m = P + d.sec[MGX_SEC_MUTS] + i * MGX_MU_WORDS;
      if (m[MGX_MU_OP] == MGX_MOP_STATS && m[MGX_MU_A0] != 0 && booked(m[MGX_MU_A2])) writes_booked = true;
      if (m[MGX_MU_OP] == MGX_MOP_GAME_VALUE && m[MGX_MU_A1] >= 0) {
        const int32_t* V = P + d.sec[MGX_SEC_OBS_VALUES] + m[MGX_MU_A1] * MGX_OV_WORDS;
        const int32_t* c0 = P + d.sec[MGX_SEC_GV_CODE] + V[MGX_OV_GV_START] * MGX_GV_WORDS;
        if (V[MGX_OV_GV_COUNT] > 0 && c0[MGX_GV_OP] == MGX_GOP_STAT && c0[MGX_GV_A0] != 1 && booked(c0[MGX_GV_A1])) writes_booked = true;
      }
    }
    d.defer_book = (reads_agent_stats || writes_booked) ? 0 : 1;  // (flushed at the end of the launch that runs the action phase)
    // ... and can be kept as integers beside the stat rows (mgx_world.h tail_shadow) when nothing on the device reads the
    // cells between two flushes: no game value at all — rewards and observation values included — reads one of them or a
    // coverage stat, and no mutation writes a coverage stat.  Lean lane-per-env dispatch only.
    bool counters = d.defer_book && !getenv("MGX_NO_SHADOW"), coverage = true;
#ifdef MGX_CPU_EMU
    counters = false;   // (the sanitizer build has no flush kernel)
#endif
    auto covers = [&](int id) { return id == d.wk[MGX_S_CELL_UNIQUE] || id == d.wk[MGX_S_CELL_MAXDIST]; };
    for (int i = 0; i < n_code; i++)
      if (code[i * MGX_GV_WORDS + MGX_GV_OP] == MGX_GOP_STAT && code[i * MGX_GV_WORDS + MGX_GV_A0] != 1) {
        if (booked(code[i * MGX_GV_WORDS + MGX_GV_A1])) counters = false;
        if (covers(code[i * MGX_GV_WORDS + MGX_GV_A1])) coverage = false;
      }
    for (int i = 0; i < n_mut; i++) {
      const int32_t* m = P + d.sec[MGX_SEC_MUTS] + i * MGX_MU_WORDS;
      if (m[MGX_MU_OP] == MGX_MOP_STATS && m[MGX_MU_A0] != 0 && covers(m[MGX_MU_A2])) coverage = false;
      if (m[MGX_MU_OP] == MGX_MOP_GAME_VALUE && m[MGX_MU_A1] >= 0) {
        const int32_t* V = P + d.sec[MGX_SEC_OBS_VALUES] + m[MGX_MU_A1] * MGX_OV_WORDS;
        const int32_t* c0 = P + d.sec[MGX_SEC_GV_CODE] + V[MGX_OV_GV_START] * MGX_GV_WORDS;
        if (V[MGX_OV_GV_COUNT] > 0 && c0[MGX_GV_OP] == MGX_GOP_STAT && c0[MGX_GV_A0] != 1 && covers(c0[MGX_GV_A1])) coverage = false;
      }
    }
    d.shadow = !counters ? 0 : coverage ? 3 : 1;   // (settled below, once the dispatch is known)
  }
  if (d.X) {  // can the action phase's top-level handlers run on the register VM?  (mgx_world.h apply_top)
    bool flat = !getenv("MGX_NO_FLAT_TOP");
    const int n_atoms = mgx_sec_cnt(P, MGX_SEC_ATOMS);
    std::vector<char> hseen(mgx_sec_cnt(P, MGX_SEC_HANDLERS), 0);
    std::vector<int> todo;
    auto push = [&](int h) { if (h >= 0 && h < (int)hseen.size() && !hseen[h]) { hseen[h] = 1; todo.push_back(h); } };
    const int32_t* mh = P + d.sec[MGX_SEC_MOVE_HANDLERS];
    for (int k = 0; k < d.n_move_handlers; k++) push(mh[k * MGX_MH_WORDS + MGX_MH_HANDLER]);
    for (int c = 0; c < P[MGX_H_NUM_CLASSES]; c++) {
      const int32_t* C = P + d.sec[MGX_SEC_CLASSES] + c * MGX_C_WORDS;
      push(C[MGX_C_ON_USE]); push(C[MGX_C_ON_AFTER_USE]); push(C[MGX_C_ON_TICK]);
    }
    push(d.game_on_tick);
    while (!todo.empty() && flat) {
      const int32_t* hd = P + d.sec[MGX_SEC_HANDLERS] + todo.back() * MGX_HD_WORDS;
      todo.pop_back();
      if (hd[MGX_HD_KIND] != MGX_HK_LEAF) {
        const int32_t* kids = P + d.sec[MGX_SEC_CHILDREN] + hd[MGX_HD_CHILD_START];
        for (int i = 0; i < hd[MGX_HD_CHILD_COUNT]; i++) push(kids[i]);
        continue;
      }
      for (int i = 0; i < hd[MGX_HD_MUT_COUNT] && flat; i++) {
        switch (P[d.sec[MGX_SEC_MUTS] + (hd[MGX_HD_MUT_START] + i) * MGX_MU_WORDS + MGX_MU_OP]) {
          case MGX_MOP_RESOURCE_DELTA: case MGX_MOP_RESOURCE_TRANSFER: case MGX_MOP_CLEAR_INVENTORY: case MGX_MOP_ATTACK: case MGX_MOP_STATS:
          case MGX_MOP_CHANGE_VIBE: case MGX_MOP_RELOCATE: case MGX_MOP_SWAP: case MGX_MOP_USE_TARGET: case MGX_MOP_GAME_VALUE: break;
          default: flat = false;   // tag mutations (lifecycle handlers), query recomputation, push / spawn / query inventory
        }
      }
      std::vector<int> pcs{hd[MGX_HD_FILTER_PC]};   // filters: no atom that evaluates a query (check_filters<0>)
      std::vector<char> aseen(n_atoms, 0);
      while (!pcs.empty() && flat) {
        const int pc = pcs.back();
        pcs.pop_back();
        if (pc < 0 || pc >= n_atoms || aseen[pc]) continue;
        aseen[pc] = 1;
        const int32_t* a = P + d.sec[MGX_SEC_ATOMS] + pc * MGX_AT_WORDS;
        if (a[MGX_AT_OP] == MGX_FOP_QUERY_RESOURCE || (a[MGX_AT_OP] == MGX_FOP_MAX_DISTANCE && a[MGX_AT_A2] >= 0)) flat = false;
        if (a[MGX_AT_OP] == MGX_FOP_GAME_VALUE)
          for (int rec : {a[MGX_AT_A1], a[MGX_AT_A2]})
            if (rec >= 0) {
              const int32_t* V = P + d.sec[MGX_SEC_OBS_VALUES] + rec * MGX_OV_WORDS;
              for (int q = 0; q < V[MGX_OV_GV_COUNT]; q++) {
                const int op = P[d.sec[MGX_SEC_GV_CODE] + (V[MGX_OV_GV_START] + q) * MGX_GV_WORDS + MGX_GV_OP];
                if (op == MGX_GOP_QUERY_INVENTORY || op == MGX_GOP_QUERY_COUNT) flat = false;
              }
            }
        pcs.push_back(a[MGX_AT_ON_TRUE]);
        pcs.push_back(a[MGX_AT_ON_FALSE]);
      }
    }
    d.flat_top = flat ? 1 : 0;
    if (getenv("MGX_VERBOSE")) fprintf(stderr, "[mgx] action-phase handlers on the %s VM\n", flat ? "register" : "LDS");
  }
  {  // Do the handler tables equal a preset the build generated straight-line handler code for (mgx_handlers_gen.h)?
    unsigned long long h = 0xCBF29CE484222325ull;   // FNV-1a over 32-bit words: mettagrid_amd/gen_handlers.py fingerprint()
    auto mix = [&](int32_t v) { h = (h ^ (unsigned long long)(uint32_t)v) * 0x100000001B3ull; };
    const int secs[5] = {MGX_SEC_HANDLERS, MGX_SEC_CHILDREN, MGX_SEC_ATOMS, MGX_SEC_MUTS, MGX_SEC_MOVE_HANDLERS};
    const int words[5] = {MGX_HD_WORDS, 1, MGX_AT_WORDS, MGX_MU_WORDS, MGX_MH_WORDS};
    for (int k = 0; k < 5; k++) {
      const int n = mgx_sec_cnt(P, secs[k]);
      mix(n);
      for (int i = 0; i < n * words[k]; i++) mix(P[d.sec[secs[k]] + i]);
    }
    mix(P[MGX_H_NUM_CLASSES]);
    for (int c = 0; c < P[MGX_H_NUM_CLASSES]; c++) {
      const int32_t* C = P + d.sec[MGX_SEC_CLASSES] + c * MGX_C_WORDS;
      mix(C[MGX_C_ON_USE]); mix(C[MGX_C_ON_AFTER_USE]); mix(C[MGX_C_ON_TICK]);
    }
    mix(P[MGX_H_GAME_ON_TICK]);
    e->handler_fp = h;
    d.gen_prog = getenv("MGX_NO_GEN") ? 0 : (!d.X && h == MGX_GEN_R3_FP) ? 3 : (d.X && h == MGX_GEN_R4_FP) ? 4 : 0;
    if (getenv("MGX_VERBOSE")) fprintf(stderr, "[mgx] handler code: %s\n", d.gen_prog ? "generated for this program at build()" : "interpreter");
  }
  {  // Can the action dispatch run with one lane per AGENT (mgx_act.h)?  Every handler an action reaches must stay with
     // its actor and target, look at no game-wide state, and the order of game-stat SETs must be recoverable.
    bool par = !getenv("MGX_ACT_SERIAL");
    int ap = 1;
    while (ap < d.A) ap <<= 1;
    if (d.A > 64) par = false;
    if (!d.X && 16 * ap > 256) par = false;   // mgx_act_fast.hip: 16 envs per workgroup of at most 256 lanes
    // Measured (MI355X, 65 536 envs): 64 agents per env 4.92 -> 2.0 ms; 16 agents per env 0.445 -> 0.52 ms — four envs share
    // a wavefront there and a round costs what 2.4 serial steps cost, so lean games stay lane per env unless asked.
    if (!d.X && !getenv("MGX_ACT_LEAN")) par = false;
    if (d.X && !d.flat_top) par = false;      // tag mutations, query recomputation / filters: lane-per-env VM
    if (d.X && !e->prog_in_lds) par = false;  // mgx_act_x.hip is built for the hot program range in LDS only
    const int32_t* mh = P + d.sec[MGX_SEC_MOVE_HANDLERS];
    for (int k = 0; k < d.n_move_handlers; k++)
      if (mh[k * MGX_MH_WORDS + MGX_MH_MAX_RANGE] != 1) par = false;   // footprint = own cell + the cell ahead
    for (int c = 0; c < P[MGX_H_NUM_CLASSES]; c++) {
      const int32_t* C = P + d.sec[MGX_SEC_CLASSES] + c * MGX_C_WORDS;
      if (d.X && C[MGX_C_KIND] == MGX_KIND_AGENT && C[MGX_C_TERR_COUNT] > 0) par = false;   // a moving territory source re-registers
    }
    std::vector<int> gset;
    const int n_code = mgx_sec_cnt(P, MGX_SEC_GV_CODE), n_vals = mgx_sec_cnt(P, MGX_SEC_OBS_VALUES);
    auto pure_value = [&](int rec) {   // reads inventories, agent-scope stats and constants only
      if (rec < 0) return true;
      if (rec >= n_vals) return false;
      const int32_t* V = P + d.sec[MGX_SEC_OBS_VALUES] + rec * MGX_OV_WORDS;
      for (int q = 0; q < V[MGX_OV_GV_COUNT]; q++) {
        const int at = V[MGX_OV_GV_START] + q;
        if (at < 0 || at >= n_code) return false;
        const int32_t* g = P + d.sec[MGX_SEC_GV_CODE] + at * MGX_GV_WORDS;
        if (g[MGX_GV_OP] == MGX_GOP_QUERY_INVENTORY || g[MGX_GV_OP] == MGX_GOP_QUERY_COUNT) return false;
        if (g[MGX_GV_OP] == MGX_GOP_STAT && g[MGX_GV_A0] == 1) return false;
      }
      return true;
    };
    const int n_atoms = mgx_sec_cnt(P, MGX_SEC_ATOMS), n_hd = mgx_sec_cnt(P, MGX_SEC_HANDLERS);
    bool saw_use_target = false;
    auto local = [&](std::vector<int> todo, bool may_move) {   // every handler reachable from `todo`
      std::vector<char> hseen(n_hd, 0);
      std::vector<int> stack;
      auto push = [&](int h) { if (h >= 0 && h < n_hd && !hseen[h]) { hseen[h] = 1; stack.push_back(h); } };
      for (int h : todo) push(h);
      bool use_target = false;
      while (!stack.empty()) {
        const int32_t* hd = P + d.sec[MGX_SEC_HANDLERS] + stack.back() * MGX_HD_WORDS;
        stack.pop_back();
        if (hd[MGX_HD_KIND] != MGX_HK_LEAF) {
          const int32_t* kids = P + d.sec[MGX_SEC_CHILDREN] + hd[MGX_HD_CHILD_START];
          for (int i = 0; i < hd[MGX_HD_CHILD_COUNT]; i++) push(kids[i]);
          continue;
        }
        for (int i = 0; i < hd[MGX_HD_MUT_COUNT]; i++) {
          const int32_t* m = P + d.sec[MGX_SEC_MUTS] + (hd[MGX_HD_MUT_START] + i) * MGX_MU_WORDS;
          switch (m[MGX_MU_OP]) {
            case MGX_MOP_RESOURCE_DELTA: case MGX_MOP_CLEAR_INVENTORY: case MGX_MOP_ATTACK: case MGX_MOP_CHANGE_VIBE: break;
            case MGX_MOP_RESOURCE_TRANSFER: if (d.X && m[MGX_MU_A4]) return false; break;   // may remove the emptied object
            case MGX_MOP_RELOCATE: case MGX_MOP_SWAP: if (!may_move) return false; break;
            case MGX_MOP_USE_TARGET: use_target = true; break;
            case MGX_MOP_STATS:
              if (!pure_value(m[MGX_MU_A3])) return false;
              if (m[MGX_MU_A0] == 0 && std::find(gset.begin(), gset.end(), m[MGX_MU_A2]) == gset.end()) gset.push_back(m[MGX_MU_A2]);
              break;
            case MGX_MOP_GAME_VALUE: {
              if (!pure_value(m[MGX_MU_A1]) || !pure_value(m[MGX_MU_A2])) return false;   // (a game-scope target is impure too)
              break;
            }
            default: return false;
          }
        }
        std::vector<int> pcs{hd[MGX_HD_FILTER_PC]};
        std::vector<char> aseen(n_atoms, 0);
        while (!pcs.empty()) {
          const int pc = pcs.back();
          pcs.pop_back();
          if (pc < 0 || pc >= n_atoms || aseen[pc]) continue;
          aseen[pc] = 1;
          const int32_t* a = P + d.sec[MGX_SEC_ATOMS] + pc * MGX_AT_WORDS;
          switch (a[MGX_AT_OP]) {
            case MGX_FOP_VIBE: case MGX_FOP_RESOURCE: case MGX_FOP_SHARED_TAG: case MGX_FOP_TAG: case MGX_FOP_TARGET_LOC_EMPTY:
            case MGX_FOP_TARGET_IS_USABLE: case MGX_FOP_PERIODIC: case MGX_FOP_TRUE: break;
            case MGX_FOP_GAME_VALUE: if (!pure_value(a[MGX_AT_A1]) || !pure_value(a[MGX_AT_A2])) return false; break;
            case MGX_FOP_MAX_DISTANCE: if (a[MGX_AT_A2] >= 0) return false; break;
            default: return false;
          }
          pcs.push_back(a[MGX_AT_ON_TRUE]);
          pcs.push_back(a[MGX_AT_ON_FALSE]);
        }
      }
      saw_use_target = use_target;   // UseTarget applies the target's on_use / the actor's on_after_use: roots of the action phase
      return true;
    };
    std::vector<int> act_roots, tick_roots;
    for (int k = 0; k < d.n_move_handlers; k++) act_roots.push_back(mh[k * MGX_MH_WORDS + MGX_MH_HANDLER]);
    for (int c = 0; c < P[MGX_H_NUM_CLASSES]; c++) {
      const int32_t* C = P + d.sec[MGX_SEC_CLASSES] + c * MGX_C_WORDS;
      act_roots.push_back(C[MGX_C_ON_USE]);
      act_roots.push_back(C[MGX_C_ON_AFTER_USE]);
      tick_roots.push_back(C[MGX_C_ON_TICK]);
    }
    // Two agents of an env side by side in the lean lane-per-env kernel (MgxDev::duo, mgx_world.h): the same conditions as the
    // lane-per-agent dispatch — every handler an action reaches stays with actor and target, move handlers look one cell
    // ahead, at most four game-scope stats are SET — without that kernel's shape limits.
    bool duo = !d.X && !getenv("MGX_NO_DUO") && d.A >= 2;
    std::vector<int> duo_gset;
    {
      for (int k = 0; k < d.n_move_handlers; k++)
        if (mh[k * MGX_MH_WORDS + MGX_MH_MAX_RANGE] != 1) duo = false;
      const std::vector<int> gset_saved = gset;
      gset.clear();
      if (duo && !local(act_roots, true)) duo = false;
      duo_gset = gset;
      gset = gset_saved;
      saw_use_target = false;
      if (duo_gset.size() > 4) duo = false;
    }
    {   // on_tick handlers that stay with their agent (no game-scope stat, no UseTarget, nothing that moves): the helper lanes
        // of the lean kernel may run them for half of the agents (MgxDev::tick_split)
      const std::vector<int> gset_saved = gset;
      const bool ok = d.any_on_tick && !d.X && local(tick_roots, false) && !saw_use_target && gset.size() == gset_saved.size();
      gset = gset_saved;
      saw_use_target = false;
      d.tick_split = (ok && !getenv("MGX_NO_TICK_SPLIT")) ? 1 : 0;
    }
    if (par && !local(act_roots, true)) par = false;
    bool tick = false;
    if (par && !d.X) {   // the lean kernel is the whole world update: its on_tick handlers must be lane-local as well
      if (d.any_on_tick && (!local(tick_roots, false) || saw_use_target)) par = false;   // (an on_tick handler that uses itself: lane per env)
      tick = par;
    }
    if (gset.size() > 4) par = false;
#ifdef MGX_CPU_EMU
    par = false;   // wavefront-cooperative (ballot / readlane): not part of the sanitizer build
#endif
    d.act_par = par ? 1 : 0;
    // integer bookkeeping: counters and coverage stats in the lean lane-per-env kernel (all or nothing: one fused pass),
    // counters only in the lane-per-agent dispatch kernels (bookkeeping_flush_one), none in the extended lane-per-env kernel
    if (d.act_par) d.shadow &= 1;
    else if (d.X || d.shadow != 3) d.shadow = 0;
    d.act_replay = getenv("MGX_ACT_SHUFFLE_REPLAY") ? 1 : 0;
    d.act_tick = (par && tick) ? 1 : 0;
    d.act_ngset = par ? (int)gset.size() : 0;
    for (int k = 0; k < 4; k++) d.act_gset_ids[k] = (par && k < (int)gset.size()) ? gset[k] : -1;
    d.duo = (duo && !par) ? 1 : 0;
    if (d.duo) {   // the paired dispatch orders game-stat SETs through the same per-env cells
      d.act_ngset = (int)duo_gset.size();
      for (int k = 0; k < 4; k++) d.act_gset_ids[k] = k < (int)duo_gset.size() ? duo_gset[k] : -1;
    }
    if (getenv("MGX_VERBOSE")) fprintf(stderr, "[mgx] lean dispatch: %s\n", d.duo ? "two agents of an env at a time (disjoint footprints)" : "one agent at a time");
    if (getenv("MGX_VERBOSE")) fprintf(stderr, "[mgx] action dispatch: one lane per %s\n", par ? "agent (conflict-ordered rounds)" : "env");
  }
  {  // reward code made only of inventory / constant arithmetic reads nothing the observation kernel writes
    bool pure = !d.X;
    const int32_t* rw = P + d.sec[MGX_SEC_REWARDS];
    const int n_rw = P[MGX_H_SECTION_BASE + 2 * MGX_SEC_REWARDS + 1];
    for (int k = 0; k < n_rw && pure; k++) {
      const int32_t* code = P + d.sec[MGX_SEC_GV_CODE] + rw[k * MGX_RW_WORDS + MGX_RW_GV_START] * MGX_GV_WORDS;
      for (int i = 0; i < rw[k * MGX_RW_WORDS + MGX_RW_GV_COUNT]; i++) {
        const int op = code[i * MGX_GV_WORDS + MGX_GV_OP];
        if (op == MGX_GOP_STAT || op == MGX_GOP_QUERY_INVENTORY || op == MGX_GOP_QUERY_COUNT) pure = false;
      }
    }
    e->rewards_early = pure;
    // Extended games: the reward expressions may still be evaluated before the kernel's end — by a wavefront that has no
    // share of the encode — when no operand is something this kernel writes (the cell.visited agent stat, the token game
    // stats) and none is a query (those go through mgx_values_kernel).
    {
      bool safe = d.X != 0;
      for (int k = 0; k < n_rw && safe; k++) {
        const int32_t* code = P + d.sec[MGX_SEC_GV_CODE] + rw[k * MGX_RW_WORDS + MGX_RW_GV_START] * MGX_GV_WORDS;
        for (int i = 0; i < rw[k * MGX_RW_WORDS + MGX_RW_GV_COUNT]; i++) {
          const int32_t* ins = code + i * MGX_GV_WORDS;
          const int op = ins[MGX_GV_OP];
          if (op == MGX_GOP_QUERY_INVENTORY || op == MGX_GOP_QUERY_COUNT) safe = false;
          if (op == MGX_GOP_STAT) {
            const int id = ins[MGX_GV_A1];
            if (ins[MGX_GV_A0] != 1 && id == d.wk[MGX_S_CELL_VISITED]) safe = false;
            if (ins[MGX_GV_A0] == 1 && (id == d.wk[MGX_S_GAME_TOKENS_WRITTEN] || id == d.wk[MGX_S_GAME_TOKENS_FREE] ||
                                        id == d.wk[MGX_S_GAME_TOKENS_DROPPED])) safe = false;
          }
        }
      }
      e->rewards_mid = safe && !getenv("MGX_REWARDS_LATE");
    }
    auto has_query = [&](int start, int count) {
      for (int i = 0; i < count; i++) {
        const int op = P[d.sec[MGX_SEC_GV_CODE] + (start + i) * MGX_GV_WORDS + MGX_GV_OP];
        if (op == MGX_GOP_QUERY_INVENTORY || op == MGX_GOP_QUERY_COUNT) return true;
      }
      return false;
    };
    for (int k = 0; k < n_rw; k++)
      if (has_query(rw[k * MGX_RW_WORDS + MGX_RW_GV_START], rw[k * MGX_RW_WORDS + MGX_RW_GV_COUNT])) e->rewards_ext = true;
    bool obsval_ext = false;
    for (int i = 0; i < d.n_obs_values; i++) {
      const int32_t* V = P + d.sec[MGX_SEC_OBS_VALUES] + i * MGX_OV_WORDS;
      if (has_query(V[MGX_OV_GV_START], V[MGX_OV_GV_COUNT])) obsval_ext = true;
    }
    if (obsval_ext) {
      int arc = e->alloc(&d.obsval, (size_t)d.E * d.A * d.n_obs_values);
      if (arc != MGX_OK) { mgx_destroy(e); return arc; }
    }
  }
  // The world kernel stages 17 (+ extended: 176) bytes per agent and env in LDS, 64 envs per workgroup: up to 160 KB,
  // i.e. about 147 agents per env in the lean variant.  Past 64 KB the kernels need the opt-in attribute.
  if (e->lds_world > 160 * 1024) {
    mgx_destroy(e);
    return fail(MGX_ERR_PROGRAM, "mgx_create: too many agents per env for the world kernel's LDS staging (160 KB per 64 envs)");
  }
  {
    static std::mutex mu;
    static int next_slot = 0;
    std::lock_guard<std::mutex> lock(mu);
    e->slot = next_slot;
    if (!d.X) next_slot = (next_slot + 1) % MGX_FAST_SLOTS;
  }
  if (!(d.X ? mgx_world_x_set_lds(e->lds_world)
            : (e->slot == 0 ? mgx_world_fast_set_lds_s0(e->lds_world) : mgx_world_fast_set_lds_s1(e->lds_world)))) {
    mgx_destroy(e);
    return fail(MGX_ERR_HIP, "mgx_create: cannot raise the world kernel's dynamic LDS limit");
  }
  if (d.act_par) {
    {  // footprint table + (maps up to 64 x 64) the cell map of the conflict lookup, per workgroup
      int ap = 1;
      while (ap < d.A) ap <<= 1;
      const int epg = d.X ? mgx_act_x_epg() : mgx_act_fast_epg();
      d.act_map = (d.H * d.W <= 4096 && !getenv("MGX_ACT_NO_MAP")) ? 1 : 0;
      d.act_lds_extra = ((4 * ap * epg + 15) & ~15) + (d.act_map ? epg * ((d.H * d.W + 15) & ~15) : 0);
    }
    e->lds_act = (d.X ? mgx_act_x_lds_bytes(d.A, d.x_aoe_lds != 0, d.act_lds_extra) : mgx_act_fast_lds_bytes(d.A, d.act_lds_extra)) +
                 (e->prog_in_lds ? (size_t)e->prog_lds_words * 4 : 0);
    if (!(d.X ? mgx_act_x_set_lds(e->lds_act) : mgx_act_fast_set_lds_s0(e->lds_act))) {
      mgx_destroy(e);
      return fail(MGX_ERR_HIP, "mgx_create: cannot raise the action kernel's dynamic LDS limit");
    }
  }
  if (e->verbose && d.X)
    fprintf(stderr, "[mgx] extended world kernel: %zu bytes of private memory per lane\n", mgx_world_x_private_bytes());
  rc = size_obs_lds(e);
  if (rc != MGX_OK) { mgx_destroy(e); return rc; }
  hipError_t he = hipSuccess;
  if (he == hipSuccess) he = hipMemcpyAsync(dprog, program, program_words * 4, hipMemcpyHostToDevice, e->stream);
  if (he == hipSuccess) he = hipMemcpyAsync(dmaps, class_maps, E * HW * 2, hipMemcpyHostToDevice, e->stream);
  if (he == hipSuccess) he = hipMemcpyAsync(dseeds, seeds, E * 4, hipMemcpyHostToDevice, e->stream);
  if (he != hipSuccess) { mgx_destroy(e); return fail(MGX_ERR_HIP, std::string("mgx_create upload: ") + hipGetErrorString(he)); }
  e->dmaps = dmaps;
  e->dseeds = dseeds;
  MGX_LAUNCH_INIT(e->stream, dev_copy(e), (const uint16_t*)dmaps,
                     (const int32_t*)nullptr, (const uint32_t*)dseeds, (const uint8_t*)nullptr, MgxList());
  he = hipGetLastError();
  if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
  if (he != hipSuccess) { mgx_destroy(e); return fail(MGX_ERR_HIP, std::string("mgx_init_kernel: ") + hipGetErrorString(he)); }
  MGX_TRACE_POINT(e, "init kernel");
  for (int i = 0; i <= MGX_T_COUNT; i++) (void)hipEventCreate(&e->ev[i]);
  (void)hipEventCreateWithFlags(&e->world_done, hipEventDisableTiming);
  rc = init_buffers(e);  // ctor -> _make_buffers -> set_buffers -> _init_buffers (mettagrid_c.cpp:190, 271-292)
  if (rc != MGX_OK) { mgx_destroy(e); return rc; }
  { std::lock_guard<std::mutex> g(g_live_mu); g_live.push_back(e); }
  *out = e;
  return MGX_OK;
}

void mgx_destroy(mgx_engine* e) {
  if (!e) return;
  {  // engines chained behind this one (mgx_chain_world) lose the dependency
    std::lock_guard<std::mutex> g(g_live_mu);
    g_live.erase(std::remove(g_live.begin(), g_live.end(), e), g_live.end());
    for (mgx_engine* o : g_live) if (o->world_after == e) o->world_after = nullptr;
  }
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  for (void* p : e->allocs) (void)hipFree(p);
  if (e->d_pool) (void)hipFree(e->d_pool);
  if (e->d_stage) (void)hipFree(e->d_stage);
  if (e->d_objs) (void)hipFree(e->d_objs);
  if (e->h_flags) (void)hipHostFree(e->h_flags);
  if (e->h_act_err) (void)hipHostFree(e->h_act_err);
  if (e->jit_world.mod) (void)hipModuleUnload(e->jit_world.mod);
  if (e->jit_obs.mod) (void)hipModuleUnload(e->jit_obs.mod);
  if (e->jit_actx.mod) (void)hipModuleUnload(e->jit_actx.mod);
  free_episode_stats(e);
  for (int i = 0; i <= MGX_T_COUNT; i++) if (e->ev[i]) (void)hipEventDestroy(e->ev[i]);
  if (e->world_done) (void)hipEventDestroy(e->world_done);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

int mgx_set_buffers(mgx_engine* e, uint8_t* observations, uint8_t* terminals, uint8_t* truncations, float* rewards,
                    int32_t* actions, int32_t* vibe_actions, int64_t n_rows, int64_t n_tokens, int32_t mem_kind) {
  if (!e) return fail(MGX_ERR_BAD_ARG, "mgx_set_buffers: null engine");
  HIP_TRY(hipSetDevice(e->device));
  MgxDev& d = e->d;
  bool all_null = !observations && !terminals && !truncations && !rewards && !actions && !vibe_actions;
  if (!all_null) {
    if (!observations || !terminals || !truncations || !rewards || !actions || !vibe_actions)
      return fail(MGX_ERR_BAD_ARG, "mgx_set_buffers: all six buffers are required");
    if (n_rows != (int64_t)d.E * d.A)  // validate_buffers (mettagrid_c.cpp:1104-1150)
      return fail(MGX_ERR_BAD_ARG, "observations has shape [" + std::to_string(n_rows) + ", " + std::to_string(n_tokens) +
                                       ", 3] but expected [" + std::to_string((int64_t)d.E * d.A) + ", [something], 3]");
    if (n_tokens != d.T) return fail(MGX_ERR_BAD_ARG, "observations token dimension does not match num_observation_tokens");
  }
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (all_null || mem_kind == MGX_MEM_HOST) {
    d.obs = e->own_obs; d.terminals = e->own_term; d.truncations = e->own_trunc; d.rewards = e->own_rew;
    d.actions = e->own_act; d.vibe_actions = e->own_vact;
  }
  if (all_null) {
    e->mem_kind = MGX_MEM_DEVICE;
    e->external = false;
  } else if (mem_kind == MGX_MEM_HOST) {
    e->mem_kind = MGX_MEM_HOST;
    e->external = true;
    e->h_obs = observations; e->h_term = terminals; e->h_trunc = truncations; e->h_rew = rewards;
    e->h_act = actions; e->h_vact = vibe_actions;
  } else if (mem_kind == MGX_MEM_DEVICE) {
    e->mem_kind = MGX_MEM_DEVICE;
    e->external = true;
    d.obs = observations; d.terminals = terminals; d.truncations = truncations; d.rewards = rewards;
    d.actions = actions; d.vibe_actions = vibe_actions;
  } else {
    return fail(MGX_ERR_BAD_ARG, "mgx_set_buffers: unknown mem_kind");
  }
  return init_buffers(e);
}

// Table of the arrays a restart clears, on the device (rebuilt when the bound caller buffers change).
static int upload_rows(mgx_engine* e) {
  const MgxDev& d = e->d;
  std::vector<MgxRow> rows;
  for (const auto& r : e->rows_state) rows.push_back({(uint8_t*)r.base, (unsigned long long)r.row_bytes, r.fill, 0});
  // caller-visible rows of the restarted envs: terminals / truncations / rewards cleared (_init_buffers :294-319)
  rows.push_back({(uint8_t*)d.terminals, (unsigned long long)d.A, 0, 0});
  rows.push_back({(uint8_t*)d.truncations, (unsigned long long)d.A, 0, 0});
  rows.push_back({(uint8_t*)d.rewards, (unsigned long long)d.A * 4, 0, 0});
  if (rows.size() == e->rows_host.size() && memcmp(rows.data(), e->rows_host.data(), rows.size() * sizeof(MgxRow)) == 0) return MGX_OK;
  if (!e->d_rows) { int rc = e->alloc(&e->d_rows, rows.size()); if (rc) return rc; }
  HIP_TRY(hipMemcpyAsync(e->d_rows, rows.data(), rows.size() * sizeof(MgxRow), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));  // `rows` is a local
  e->rows_host = rows;
  e->n_rows = (int)rows.size();
  return MGX_OK;
}
static int stage(mgx_engine* e, size_t bytes) {
  if (bytes <= e->stage_bytes) return MGX_OK;
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (e->d_stage) (void)hipFree(e->d_stage);
  e->d_stage = nullptr;
  e->stage_bytes = 0;
  HIP_TRY(hipMalloc(&e->d_stage, bytes));
  e->stage_bytes = bytes;
  return MGX_OK;
}
// Restart the envs of a DEVICE mask: clear their state rows, rebuild them (construction kernel), initial observations.
// from_pool: maps come from the pool through d_map_index; bump: auto-reset bookkeeping (episode counter, next pool map).
// l: the same envs as an ascending device list (the kernels then walk the list with a small grid instead of testing E mask bytes).
static int restart_masked(mgx_engine* e, const uint8_t* dmask, bool from_pool, bool bump, const MgxList& l = MgxList()) {
  const MgxDev& d = e->d;
  int rc = consume_out_fence(e);   // the cleared rows include terminals / truncations / rewards
  if (rc) return rc;
  rc = upload_rows(e);
  if (rc) return rc;
#ifdef MGX_CPU_EMU
  MgxList ll;   // the sanitizer build runs the mask forms
#else
  MgxList ll = l;
#endif
  ll.passes = 2;   // new MettaGrid + set_buffers = two initial observation passes in the reference (token statistics)
  hipLaunchKernelGGL(mgx_clear_rows_kernel, dim3(ll.list ? list_grid(e, ll, 128, 1024) : (unsigned)d.E), dim3(256), 0, e->stream, (const MgxRow*)e->d_rows,
                     e->n_rows, dmask, d.E, bump ? e->d_episodes : (uint32_t*)nullptr, bump ? e->d_map_index : (int32_t*)nullptr, e->n_pool,
                     e->pool_stride, ll.list, ll.n);
  HIP_TRY(hipGetLastError());
  MGX_LAUNCH_INIT(e->stream, dev_copy(e),
                     (const uint16_t*)(from_pool ? e->d_pool : e->dmaps), (const int32_t*)(from_pool ? e->d_map_index : nullptr),
                     (const uint32_t*)e->dseeds, dmask, ll);
  HIP_TRY(hipGetLastError());
  return launch_obs(e, false, dmask, ll);
}
// Class ids of maps handed over by the caller: 0 = empty, else class index + 1.
static int validate_maps(const mgx_engine* e, const uint16_t* maps, size_t n_maps, const uint8_t* mask, const char* who) {
  const size_t hw = (size_t)e->d.H * e->d.W;
  const int nc = e->prog[MGX_H_NUM_CLASSES];
  for (size_t m = 0; m < n_maps; m++) {
    if (mask && !mask[m]) continue;
    for (size_t i = 0; i < hw; i++)
      if (maps[m * hw + i] > nc)
        return fail(MGX_ERR_BAD_ARG, std::string(who) + ": class map holds id " + std::to_string(maps[m * hw + i]) +
                                         " but the program has " + std::to_string(nc) + " classes");
  }
  return MGX_OK;
}
// AoE / territory source counts of a set of maps against the capacities sized at mgx_create, and the observation kernel's
// token pool against their object lists.
static int fit_maps(mgx_engine* e, const uint16_t* maps, size_t first, size_t count, const uint8_t* mask, const char* who) {
  const MgxDev& d = e->d;
  const int32_t* P = e->prog.data();
  const int nc = P[MGX_H_NUM_CLASSES];
  const size_t hw = (size_t)d.H * d.W;
  if (d.X && (d.NF || d.NM || d.NTS || mgx_sec_cnt(P, MGX_SEC_AOES) > 0 || d.NT > 0)) {
    std::vector<int> cf(nc, 0), cm(nc, 0), ct(nc, 0);
    for (int c = 0; c < nc; c++) {
      const int32_t* C = P + d.sec[MGX_SEC_CLASSES] + c * MGX_C_WORDS;
      for (int i = 0; i < C[MGX_C_AOE_COUNT]; i++)
        (P[d.sec[MGX_SEC_AOES] + (C[MGX_C_AOE_START] + i) * MGX_AO_WORDS + MGX_AO_STATIC] ? cf[c] : cm[c])++;
      ct[c] = C[MGX_C_TERR_COUNT];
    }
    const int spare_f = P[MGX_H_SPAWNS] ? 0 : 0;
    (void)spare_f;
    for (size_t m = first; m < first + count; m++) {
      if (mask && !mask[m]) continue;
      int f = 0, mo = 0, t = 0;
      for (size_t i = 0; i < hw; i++) {
        const int k = maps[m * hw + i];
        if (k > 0 && k <= nc) { f += cf[k - 1]; mo += cm[k - 1]; t += ct[k - 1]; }
      }
      if (f > d.NF || mo > d.NM || t > d.NTS)
        return fail(MGX_ERR_PROGRAM, std::string(who) + ": a map holds more AoE / territory sources (" + std::to_string(f) + " fixed, " +
                                         std::to_string(mo) + " mobile, " + std::to_string(t) + " territory) than any map given to "
                                         "mgx_create (capacity " + std::to_string(d.NF) + " / " + std::to_string(d.NM) + " / " +
                                         std::to_string(d.NTS) + "): create the engine with a map that has the maximum");
    }
  }
  if (e->pool_from_maps) {  // new maps may hold more non-static objects than any map seen so far
    const long long bound = e->list_tokens_bound(maps, first, count, mask);
    const int need = (e->pool_prefix + (int)std::min<long long>(bound, 16384) + 7) & ~7;
    if (need > e->pool_tokens) {
      e->pool_tokens = need;
      int rcl = size_obs_lds(e);
      if (rcl) return rcl;
    }
  }
  return MGX_OK;
}

int mgx_reset_envs(mgx_engine* e, const uint8_t* env_mask, const uint16_t* class_maps, const uint32_t* seeds) {
  if (!e || !env_mask) return fail(MGX_ERR_BAD_ARG, "mgx_reset_envs: null argument");
  HIP_TRY(hipSetDevice(e->device));
  const MgxDev& d = e->d;
  const size_t E = d.E, HW = (size_t)d.H * d.W, A = d.A;
  std::vector<int32_t> idx;
  for (size_t i = 0; i < E; i++) if (env_mask[i]) idx.push_back((int32_t)i);
  if (idx.empty()) return MGX_OK;
  const size_t n = idx.size();
  if (class_maps) {
    int rc = validate_maps(e, class_maps, E, env_mask, "mgx_reset_envs");
    if (!rc) rc = fit_maps(e, class_maps, 0, E, env_mask, "mgx_reset_envs");
    if (rc) return rc;
  }
  // ONE contiguous upload: [indices | packed seeds | packed maps] of the masked envs, scattered on the device
  const size_t off_seeds = n * 4, off_maps = (off_seeds + (seeds ? n * 4 : 0) + 15) & ~(size_t)15;
  const size_t off_n = (off_maps + (class_maps ? n * HW * 2 : 0) + 15) & ~(size_t)15;   // the list length, for the list-walking kernels
  const size_t total = off_n + 16;
  std::vector<uint8_t> host(total);
  memcpy(host.data(), idx.data(), n * 4);
  { const uint32_t n32 = (uint32_t)n; memcpy(host.data() + off_n, &n32, 4); }
  for (size_t k = 0; k < n; k++) {
    if (seeds) memcpy(host.data() + off_seeds + k * 4, seeds + idx[k], 4);
    if (class_maps) memcpy(host.data() + off_maps + k * HW * 2, class_maps + (size_t)idx[k] * HW, HW * 2);
  }
  int rc = stage(e, total);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(e->dmask, env_mask, E, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->d_stage, host.data(), total, hipMemcpyHostToDevice, e->stream));
  const uint8_t* st = (const uint8_t*)e->d_stage;
  if (seeds) hipLaunchKernelGGL(mgx_scatter_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, e->dseeds,
                                (const uint32_t*)(st + off_seeds), (const int32_t*)st, (int)n);
  if (class_maps) hipLaunchKernelGGL(mgx_scatter_maps_kernel, dim3(4, (unsigned)n), dim3(256), 0, e->stream, e->dmaps,
                                     (const uint16_t*)(st + off_maps), (const int32_t*)st, (int)n, (int)HW);
  HIP_TRY(hipGetLastError());
  MgxList l;
  l.list = (const int32_t*)st; l.n = (const uint32_t*)(st + off_n); l.n_host = (int)n;
  rc = restart_masked(e, e->dmask, false, false, l);
  if (rc) return rc;
  if (e->mem_kind == MGX_MEM_HOST) {  // host buffers: the restarted rows, one copy per contiguous run of envs
    for (size_t k = 0; k < n;) {
      size_t j = k;
      while (j + 1 < n && idx[j + 1] == idx[j] + 1) j++;
      const size_t r0 = (size_t)idx[k] * A, rows = (size_t)(idx[j] - idx[k] + 1) * A;
      HIP_TRY(hipMemcpyAsync(e->h_obs + r0 * d.T * 3, d.obs + r0 * d.T * 3, rows * d.T * 3, hipMemcpyDeviceToHost, e->stream));
      HIP_TRY(hipMemcpyAsync(e->h_term + r0, d.terminals + r0, rows, hipMemcpyDeviceToHost, e->stream));
      HIP_TRY(hipMemcpyAsync(e->h_trunc + r0, d.truncations + r0, rows, hipMemcpyDeviceToHost, e->stream));
      HIP_TRY(hipMemcpyAsync(e->h_rew + r0, d.rewards + r0, rows * 4, hipMemcpyDeviceToHost, e->stream));
      k = j + 1;
    }
  }
  HIP_TRY(hipStreamSynchronize(e->stream));  // `host` is a local
  return MGX_OK;
}

int mgx_set_map_pool(mgx_engine* e, const uint16_t* class_maps, int32_t n_maps) {
  if (!e || !class_maps || n_maps <= 0) return fail(MGX_ERR_BAD_ARG, "mgx_set_map_pool: null/empty argument");
  HIP_TRY(hipSetDevice(e->device));
  const MgxDev& d = e->d;
  const size_t HW = (size_t)d.H * d.W;
  int rc = validate_maps(e, class_maps, (size_t)n_maps, nullptr, "mgx_set_map_pool");
  if (!rc) rc = fit_maps(e, class_maps, 0, (size_t)n_maps, nullptr, "mgx_set_map_pool");
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (e->d_pool) (void)hipFree(e->d_pool);
  e->d_pool = nullptr;
  HIP_TRY(hipMalloc((void**)&e->d_pool, (size_t)n_maps * HW * 2));
  HIP_TRY(hipMemcpyAsync(e->d_pool, class_maps, (size_t)n_maps * HW * 2, hipMemcpyHostToDevice, e->stream));
  if (!e->d_map_index) {
    rc = e->alloc(&e->d_map_index, (size_t)d.E);
    if (!rc) rc = e->alloc(&e->d_episodes, (size_t)d.E);
    if (rc) return rc;
  }
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->n_pool = n_maps;
  return MGX_OK;
}

int mgx_reset_envs_from_pool(mgx_engine* e, const uint8_t* env_mask, const int32_t* pool_index, const uint32_t* seeds) {
  if (!e || !env_mask || !pool_index) return fail(MGX_ERR_BAD_ARG, "mgx_reset_envs_from_pool: null argument");
  if (e->n_pool <= 0) return fail(MGX_ERR_BAD_ARG, "mgx_reset_envs_from_pool: no map pool (mgx_set_map_pool)");
  HIP_TRY(hipSetDevice(e->device));
  const MgxDev& d = e->d;
  const size_t E = d.E;
  std::vector<int32_t> idx;
  std::vector<uint32_t> vals;
  for (size_t i = 0; i < E; i++) if (env_mask[i]) {
    if (pool_index[i] < 0 || pool_index[i] >= e->n_pool) return fail(MGX_ERR_BAD_ARG, "mgx_reset_envs_from_pool: pool index out of range");
    idx.push_back((int32_t)i);
  }
  if (idx.empty()) return MGX_OK;
  const size_t n = idx.size();
  std::vector<uint8_t> host(n * 12 + 16);
  memcpy(host.data(), idx.data(), n * 4);
  { const uint32_t n32 = (uint32_t)n; memcpy(host.data() + n * 12, &n32, 4); }
  for (size_t k = 0; k < n; k++) {
    memcpy(host.data() + n * 4 + k * 4, pool_index + idx[k], 4);
    if (seeds) memcpy(host.data() + n * 8 + k * 4, seeds + idx[k], 4);
  }
  int rc = stage(e, host.size());
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(e->dmask, env_mask, E, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->d_stage, host.data(), host.size(), hipMemcpyHostToDevice, e->stream));
  const uint8_t* st = (const uint8_t*)e->d_stage;
  const dim3 g((unsigned)((n + 255) / 256)), b(256);
  hipLaunchKernelGGL(mgx_scatter_words_kernel, g, b, 0, e->stream, (uint32_t*)e->d_map_index, (const uint32_t*)(st + n * 4), (const int32_t*)st, (int)n);
  if (seeds) hipLaunchKernelGGL(mgx_scatter_words_kernel, g, b, 0, e->stream, e->dseeds, (const uint32_t*)(st + n * 8), (const int32_t*)st, (int)n);
  HIP_TRY(hipGetLastError());
  MgxList l;
  l.list = (const int32_t*)st; l.n = (const uint32_t*)(st + n * 12); l.n_host = (int)n;
  rc = restart_masked(e, e->dmask, true, false, l);
  if (rc) return rc;
  if (e->mem_kind == MGX_MEM_HOST) {
    const size_t rows = E * d.A;
    HIP_TRY(hipMemcpyAsync(e->h_obs, d.obs, rows * d.T * 3, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(e->h_term, d.terminals, rows, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(e->h_trunc, d.truncations, rows, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(e->h_rew, d.rewards, rows * 4, hipMemcpyDeviceToHost, e->stream));
  }
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MGX_OK;
}

int mgx_set_auto_reset(mgx_engine* e, int32_t enabled, int32_t pool_stride, const uint32_t* early_end_steps) {
  if (!e) return fail(MGX_ERR_BAD_ARG, "mgx_set_auto_reset: null engine");
  HIP_TRY(hipSetDevice(e->device));
  if (!enabled) { e->auto_reset = false; return MGX_OK; }
  if (e->n_pool <= 0) return fail(MGX_ERR_BAD_ARG, "mgx_set_auto_reset: no map pool (mgx_set_map_pool)");
  const MgxDev& d = e->d;
  int rc = MGX_OK;
  if (!e->d_next_mask) {
    rc = e->alloc(&e->d_next_mask, ((size_t)d.E + 3) & ~(size_t)3);
    if (!rc) rc = e->alloc(&e->d_counters, 2);
    if (!rc && !e->d_done_list) rc = e->alloc(&e->d_done_list, (size_t)d.E);
    if (!rc && !e->d_done_n) rc = e->alloc(&e->d_done_n, 4);
    if (rc) return rc;
    HIP_TRY(hipHostMalloc((void**)&e->h_flags, 8, hipHostMallocMapped));
    e->h_flags[0] = 0xFFFFFFFFu; e->h_flags[1] = 0;
    HIP_TRY(hipHostGetDevicePointer((void**)&e->h_flags_dev, e->h_flags, 0));
  }
  if (early_end_steps) {
    if (!e->d_early) { rc = e->alloc(&e->d_early, (size_t)d.E); if (rc) return rc; }
    HIP_TRY(hipMemcpyAsync(e->d_early, early_end_steps, (size_t)d.E * 4, hipMemcpyHostToDevice, e->stream));
  } else if (e->d_early) {
    HIP_TRY(hipMemsetAsync(e->d_early, 0, (size_t)d.E * 4, e->stream));
  }
  HIP_TRY(hipMemsetAsync(e->d_next_mask, 0, (size_t)d.E, e->stream));
  HIP_TRY(hipMemsetAsync(e->d_done_n, 0, 4, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->pool_stride = pool_stride > 0 ? pool_stride : 1;
  e->auto_reset = true;
  e->h_flags[0] = e->step_seq; e->h_flags[1] = 0;  // nothing is done before the first step
  return MGX_OK;
}

int mgx_get_episodes(mgx_engine* e, uint32_t* episodes, int32_t* map_index) {
  if (!e || (!episodes && !map_index)) return fail(MGX_ERR_BAD_ARG, "mgx_get_episodes: null argument");
  if (!e->d_episodes) return fail(MGX_ERR_BAD_ARG, "mgx_get_episodes: no map pool (mgx_set_map_pool)");
  HIP_TRY(hipSetDevice(e->device));
  if (episodes) HIP_TRY(hipMemcpyAsync(episodes, e->d_episodes, (size_t)e->d.E * 4, hipMemcpyDeviceToHost, e->stream));
  if (map_index) HIP_TRY(hipMemcpyAsync(map_index, e->d_map_index, (size_t)e->d.E * 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MGX_OK;
}

int mgx_attach_code(mgx_engine* e, int32_t kind, const char* path) {
  if (!e || !path) return fail(MGX_ERR_BAD_ARG, "mgx_attach_code: null argument");
#ifdef MGX_CPU_EMU
  return fail(MGX_ERR_BAD_ARG, "mgx_attach_code: not part of the sanitizer build");
#else
  HIP_TRY(hipSetDevice(e->device));
  const MgxDev& d = e->d;
  if (kind != MGX_CODE_WORLD && kind != MGX_CODE_OBS && kind != MGX_CODE_ACT_X) return fail(MGX_ERR_BAD_ARG, "mgx_attach_code: unknown kind");
  if ((d.X != 0) != (kind == MGX_CODE_ACT_X))
    return fail(MGX_ERR_PROGRAM, "mgx_attach_code: world / observation code objects are for lean programs, the dispatch one for extended programs");
  mgx_engine::Jit& j = kind == MGX_CODE_WORLD ? e->jit_world : kind == MGX_CODE_OBS ? e->jit_obs : e->jit_actx;
  if (j.mod) return fail(MGX_ERR_BAD_ARG, "mgx_attach_code: a code object of this kind is attached already");
  hipModule_t mod = nullptr;
  hipError_t he = hipModuleLoad(&mod, path);
  if (he != hipSuccess) return fail(MGX_ERR_HIP, std::string("mgx_attach_code: hipModuleLoad(") + path + "): " + hipGetErrorString(he));
  auto refuse = [&](const std::string& why) { (void)hipModuleUnload(mod); return fail(MGX_ERR_PROGRAM, "mgx_attach_code: " + why); };
  auto read_sym = [&](const char* name, void* dst, size_t bytes) -> bool {
    hipDeviceptr_t p = nullptr;
    size_t n = 0;
    if (hipModuleGetGlobal(&p, &n, mod, name) != hipSuccess || n < bytes) return false;
    return hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost) == hipSuccess;
  };
  if (kind == MGX_CODE_ACT_X) {
    if (!d.act_par) return refuse("the engine does not run the lane-per-agent dispatch for this program");
    unsigned long long info[8] = {};
    if (!read_sym("mgx_jit_info", info, sizeof info) || info[0] != 0x4D47584A49544131ull) return refuse("not a dispatch code object");
    if (info[1] != sizeof(MgxDev) || info[6] != (unsigned long long)MGX_VERSION) return refuse("built against other headers than this libmgx");
    if (info[4] != e->handler_fp) return refuse("generated for another program (handler tables differ)");
    if ((int)info[2] != mgx_act_x_epg()) return refuse("built for another workgroup shape");
    if (e->lds_act > 64 * 1024) return refuse("the kernel needs more than 64 KiB of LDS");
    if (hipModuleGetFunction(&j.f0, mod, "mgx_jit_act_x") != hipSuccess) return refuse("no mgx_jit_act_x kernel");
    j.epg = (int)info[2];
    j.mod = mod;
    e->d.gen_prog = (int)info[5];
  } else if (kind == MGX_CODE_WORLD) {
    if (d.act_par) return refuse("the engine runs the lane-per-agent dispatch (MGX_ACT_LEAN), not the kernel this code object replaces");
    unsigned long long info[8] = {};
    if (!read_sym("mgx_jit_info", info, sizeof info) || info[0] != 0x4D47584A49545731ull) return refuse("not a world code object");
    if (info[1] != sizeof(MgxDev) || info[6] != (unsigned long long)MGX_VERSION) return refuse("built against other headers than this libmgx");
    if (info[4] != e->handler_fp) {
      char buf[96];
      snprintf(buf, sizeof buf, " (code object %016llx, engine %016llx)", info[4], e->handler_fp);
      return refuse(std::string("generated for another program (handler tables differ)") + buf);
    }
    if (e->lds_world > 64 * 1024) return refuse("the kernel needs more than 64 KiB of LDS");
    hipDeviceptr_t sym = nullptr;
    size_t n = 0;
    if (hipModuleGetGlobal(&sym, &n, mod, "g_mgx_dev") != hipSuccess || n != sizeof(MgxDev)) return refuse("no g_mgx_dev symbol of the right size");
    if (hipModuleGetFunction(&j.f0, mod, e->prog_in_lds ? "mgx_jit_world_pl" : "mgx_jit_world_pg") != hipSuccess)
      return refuse(std::string("no ") + (e->prog_in_lds ? "mgx_jit_world_pl" : "mgx_jit_world_pg") + " kernel");
    j.dev_sym = sym; j.epg = (int)info[2]; j.threads = (int)info[3]; j.dev_valid = false;
    j.mod = mod;
    e->d.gen_prog = (int)info[5];
  } else {
    long long info[24] = {};
    if (!read_sym("mgx_jit_obs_info", info, sizeof info) || info[0] != 0x4D47584A49544F31ll) return refuse("not an observation code object");
    if (info[1] != (long long)sizeof(MgxDev) || info[4] != MGX_VERSION) return refuse("built against other headers than this libmgx");
    if (!e->obs_blk_lds || e->box_dtype != MGX_BOX_OFF) return refuse("the engine does not run the lean token-row kernel with the program block in LDS");
    const long long* K = info + 5;   // H, W, A, S, T, NOFF, BASE, NOV, NT, NRW, FLAGS, MAX_STEPS, BLKW, MASK_FEAT, REWARDS_EARLY
    const int rmode = e->rewards_early ? 1 : e->rewards_mid ? 2 : 0;
    const bool same = d.H == K[0] && d.W == K[1] && d.A == K[2] && d.S == K[3] && d.T == K[4] && d.NOFF == K[5] && d.base == K[6] &&
                      d.n_obs_values == K[7] && d.NT == K[8] && d.NRW == K[9] && d.flags == K[10] && (K[11] < 0 || d.max_steps == K[11]) &&
                      e->obs_blk_words == K[12] && d.aoe_mask_feat == K[13] && rmode == K[14];
    if (!same) return refuse("compiled for another observation shape");
    if (e->lds_obs > 64 * 1024) return refuse("the kernel needs more than 64 KiB of LDS");
    if (hipModuleGetFunction(&j.f0, mod, "mgx_jit_obs_r") != hipSuccess || hipModuleGetFunction(&j.f1, mod, "mgx_jit_obs_n") != hipSuccess)
      return refuse("kernels missing");
    j.threads = (int)info[2];
    j.mod = mod;
    e->obs_variant = 9;
  }
  return MGX_OK;
#endif
}

int mgx_set_episode_stats(mgx_engine* e, int32_t enabled, int32_t log_capacity, int32_t log_per_agent) {
  if (!e) return fail(MGX_ERR_BAD_ARG, "mgx_set_episode_stats: null engine");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  free_episode_stats(e);
  if (!enabled) return MGX_OK;
#ifdef MGX_CPU_EMU
  return fail(MGX_ERR_BAD_ARG, "mgx_set_episode_stats: not part of the sanitizer build");
#else
  static_assert(MGX_EPT_HEADER == MGX_EP_TOT_HDR, "totals header");
  if (log_capacity < 0) return fail(MGX_ERR_BAD_ARG, "mgx_set_episode_stats: negative log capacity");
  const MgxDev& d = e->d;
  e->ep = mgx_ep_layout(d.NG, d.NS, d.A, d.NSP, log_per_agent ? 1 : 0);
  const size_t TW = (size_t)e->ep_tw(), nch = ((size_t)d.E + MGX_EP_CHUNK - 1) / MGX_EP_CHUNK;
  HIP_TRY(hipMalloc((void**)&e->d_ep_rec, (size_t)d.E * e->ep.rec_words * 4));
  HIP_TRY(hipMalloc((void**)&e->d_ep_partial, nch * TW * 8));
  HIP_TRY(hipMalloc((void**)&e->d_ep_totals, TW * 8));
  HIP_TRY(hipMalloc((void**)&e->d_ep_snap, TW * 8));
  HIP_TRY(hipMalloc((void**)&e->d_ep_ticket, 4));
  HIP_TRY(hipMalloc((void**)&e->d_ep_log_state, 8));
  HIP_TRY(hipHostMalloc((void**)&e->h_ep_snap, TW * 8, hipHostMallocDefault));
  HIP_TRY(hipEventCreateWithFlags(&e->ep_ev, hipEventDisableTiming));
  HIP_TRY(hipMemsetAsync(e->d_ep_ticket, 0, 4, e->stream));
  HIP_TRY(hipMemsetAsync(e->d_ep_log_state, 0, 8, e->stream));
  if (log_capacity > 0) {
    HIP_TRY(hipMalloc((void**)&e->d_ep_log, (size_t)log_capacity * e->ep.log_words * 4));
    e->ep_log_cap = log_capacity;
  }
  if (!e->d_done_list) { int rc = e->alloc(&e->d_done_list, (size_t)d.E); if (rc) return rc; }
  if (!e->d_done_n) { int rc = e->alloc(&e->d_done_n, 4); if (rc) return rc; }
  // totals start as a cleared snapshot would leave them
  hipLaunchKernelGGL(mgx_episode_snapshot_kernel, dim3(1), dim3(256), 0, e->stream, e->d_ep_totals, e->d_ep_snap, (int)TW);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->ep_stats = true;
  return MGX_OK;
#endif
}

int mgx_episode_stats_layout(mgx_engine* e, int32_t* out) {
  if (!e || !out) return fail(MGX_ERR_BAD_ARG, "mgx_episode_stats_layout: null argument");
  if (!e->ep_stats) return fail(MGX_ERR_BAD_ARG, "mgx_episode_stats_layout: episode statistics are off (mgx_set_episode_stats)");
  const MgxEpLayout& L = e->ep;
  const int32_t v[MGX_EPL_COUNT] = {L.NG, L.NS, L.A, e->ep_tw(), L.rec_words, L.log_words, L.off_game, L.off_gbits, L.off_agent, L.off_abits,
                                    L.off_rew, L.off_pa, L.off_pabits, L.off_invk, L.off_invn, L.per_agent, e->ep_log_cap};
  memcpy(out, v, sizeof(v));
  return MGX_OK;
}

int mgx_record_episodes(mgx_engine* e, const uint8_t* env_mask) {
  if (!e || !env_mask) return fail(MGX_ERR_BAD_ARG, "mgx_record_episodes: null argument");
  if (!e->ep_stats) return fail(MGX_ERR_BAD_ARG, "mgx_record_episodes: episode statistics are off (mgx_set_episode_stats)");
  if (e->auto_reset) return fail(MGX_ERR_BAD_ARG, "mgx_record_episodes: the engine is in auto-reset mode and records finished envs itself");
  HIP_TRY(hipSetDevice(e->device));
  std::vector<int32_t> idx;
  for (int i = 0; i < e->d.E; i++) if (env_mask[i]) idx.push_back(i);
  if (idx.empty()) return MGX_OK;
  const uint32_t n = (uint32_t)idx.size();
  HIP_TRY(hipMemcpyAsync(e->d_done_list, idx.data(), (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->d_done_n, &n, 4, hipMemcpyHostToDevice, e->stream));
  int rc = launch_episode_stats(e);
  if (rc) return rc;
  HIP_TRY(hipMemsetAsync(e->d_done_n, 0, 4, e->stream));   // the list belongs to this call only
  HIP_TRY(hipStreamSynchronize(e->stream));  // `idx` and `n` are locals
  return MGX_OK;
}

int mgx_request_episode_stats(mgx_engine* e) {
  if (!e) return fail(MGX_ERR_BAD_ARG, "mgx_request_episode_stats: null engine");
  if (!e->ep_stats) return fail(MGX_ERR_BAD_ARG, "mgx_request_episode_stats: episode statistics are off (mgx_set_episode_stats)");
  if (e->ep_pending) return MGX_OK;   // the snapshot already under way is fetched first; the totals keep accumulating
#ifndef MGX_CPU_EMU
  HIP_TRY(hipSetDevice(e->device));
  const int TW = e->ep_tw();
  hipLaunchKernelGGL(mgx_episode_snapshot_kernel, dim3(1), dim3(256), 0, e->stream, e->d_ep_totals, e->d_ep_snap, TW);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(e->h_ep_snap, e->d_ep_snap, (size_t)TW * 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipEventRecord(e->ep_ev, e->stream));
  e->ep_pending = true;
#endif
  return MGX_OK;
}

int mgx_fetch_episode_stats(mgx_engine* e, int32_t wait, double* totals, int32_t* ready) {
  if (!e || !totals || !ready) return fail(MGX_ERR_BAD_ARG, "mgx_fetch_episode_stats: null argument");
  *ready = 0;
  if (!e->ep_stats) return fail(MGX_ERR_BAD_ARG, "mgx_fetch_episode_stats: episode statistics are off (mgx_set_episode_stats)");
  if (!e->ep_pending) return MGX_OK;
  HIP_TRY(hipSetDevice(e->device));
  if (wait) {
    HIP_TRY(hipEventSynchronize(e->ep_ev));
  } else {
    const hipError_t q = hipEventQuery(e->ep_ev);
    if (q == hipErrorNotReady) return MGX_OK;
    if (q != hipSuccess) return fail(MGX_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(q));
  }
  memcpy(totals, e->h_ep_snap, (size_t)e->ep_tw() * 8);
  e->ep_pending = false;
  *ready = 1;
  return MGX_OK;
}

int mgx_drain_episode_log(mgx_engine* e, uint32_t* records, int32_t max_records, int32_t* n_records, int32_t* n_dropped) {
  if (!e || !records || !n_records) return fail(MGX_ERR_BAD_ARG, "mgx_drain_episode_log: null argument");
  if (!e->ep_stats || !e->d_ep_log) return fail(MGX_ERR_BAD_ARG, "mgx_drain_episode_log: no episode log (mgx_set_episode_stats)");
  if (max_records < e->ep_log_cap) return fail(MGX_ERR_BAD_ARG, "mgx_drain_episode_log: the buffer must hold the whole log (log_capacity records)");
  HIP_TRY(hipSetDevice(e->device));
  uint32_t st[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(st, e->d_ep_log_state, 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (st[0]) HIP_TRY(hipMemcpyAsync(records, e->d_ep_log, (size_t)st[0] * e->ep.log_words * 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemsetAsync(e->d_ep_log_state, 0, 8, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  *n_records = (int32_t)st[0];
  if (n_dropped) *n_dropped = (int32_t)st[1];
  return MGX_OK;
}

int mgx_step(mgx_engine* e) {
  if (!e) return fail(MGX_ERR_BAD_ARG, "mgx_step: null engine");
  HIP_TRY(hipSetDevice(e->device));
  const MgxDev& d = e->d;
  const size_t rows = (size_t)d.E * d.A;
  e->terr_fresh = false;
  if (e->mem_kind == MGX_MEM_HOST) {
    HIP_TRY(hipMemcpyAsync(e->own_act, e->h_act, rows * 4, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->own_vact, e->h_vact, rows * 4, hipMemcpyHostToDevice, e->stream));
  }
  if (e->auto_reset) {
    // Lazy auto-reset (mettagrid_puffer_env.py:299-302): envs found done at the end of the previous step are restarted
    // first, then stepped.  The previous step published its done count in host-visible memory; when the host has
    // already synchronised with it and nothing was done, the restart launches are skipped altogether.
    const bool known_none = e->h_flags[0] == e->step_seq && e->h_flags[1] == 0;
    if (!known_none) {
      MgxList l;
      l.list = e->d_done_list; l.n = e->d_done_n;
      int rrc = restart_masked(e, e->d_next_mask, true, true, l);
      if (rrc) return rrc;
    }
  }
  // timing segments (profiling only): event k closes segment k - 1 of include/mgx.h MGX_T_*
#define MGX_MARK(k) do { if (e->profiling) HIP_TRY(hipEventRecord(e->ev[k], e->stream)); } while (0)
  MGX_MARK(0);
  {
    const int pw = e->prog_lds_words;
    if (e->world_after) HIP_TRY(hipStreamWaitEvent(e->stream, e->world_after->world_done, 0));
    MGX_TRACE_POINT(e, "before world");
    const bool x_events = d.n_schedule > 0 || (d.any_on_tick && !d.tick_in_aoe);   // what is left of the action launch behind mgx_act_kernel
    if (!d.X) {
      if (d.act_par) {
        mgx_launch_act_fast_s0(e->prog_in_lds, e->lds_act, e->stream, e->d, pw);   // (one constant-memory copy: an opt-in path)
      } else {
        if (e->jit_world.mod) { int jrc = launch_world_jit(e, pw); if (jrc) return jrc; }
        else if (e->slot == 0) mgx_launch_world_fast_s0(e->prog_in_lds, e->lds_world, e->stream, e->d, pw);
        else mgx_launch_world_fast_s1(e->prog_in_lds, e->lds_world, e->stream, e->d, pw);
      }
      MGX_MARK(1); MGX_MARK(2); MGX_MARK(3);
    } else if (e->aoe_local && (d.NF > 0 || d.NM > 0 || d.NT > 0)) {
      if (d.act_par) {
        if (e->jit_actx.mod) { int jrc = launch_act_x_jit(e, dev_copy_world_x(e), pw); if (jrc) return jrc; }
        else mgx_launch_act_x(e->prog_in_lds, e->lds_act, e->stream, e->d, dev_copy_world_x(e), pw);
        if (x_events) mgx_launch_world_x(e->prog_in_lds, e->lds_world, e->stream, e->d, dev_copy_world_x(e), pw, MGX_PH_EVENTS);
      } else {
        mgx_launch_world_x(e->prog_in_lds, e->lds_world, e->stream, e->d, dev_copy_world_x(e), pw, MGX_PH_ACTIONS | MGX_PH_EVENTS);
      }
      MGX_TRACE_POINT(e, "world kernel (actions)");
      MGX_MARK(1);
      int trc = launch_terr(e);  // the per-agent territory effects read the ownership map
      if (trc) return trc;
      mgx_launch_aoe(e->stream, e->d, dev_copy(e), e->aoe_prog_lds ? dev_copy_hot(e) : nullptr, e->aoe_prog_lds ? e->hot_words : 0);
      MGX_TRACE_POINT(e, "aoe kernel");
      MGX_MARK(2);
      if (!d.cov_in_aoe) mgx_launch_world_x(e->prog_in_lds, e->lds_world, e->stream, e->d, dev_copy_world_x(e), pw, MGX_PH_TAIL);
      else e->terr_fresh = true;   // target-local area effects move no source and change no tag: the ownership maps stay current
      MGX_MARK(3);
    } else {
      if (d.act_par) {
        if (e->jit_actx.mod) { int jrc = launch_act_x_jit(e, dev_copy_world_x(e), pw); if (jrc) return jrc; }
        else mgx_launch_act_x(e->prog_in_lds, e->lds_act, e->stream, e->d, dev_copy_world_x(e), pw);
        mgx_launch_world_x(e->prog_in_lds, e->lds_world, e->stream, e->d, dev_copy_world_x(e), pw, MGX_PH_ALL & ~MGX_PH_ACTIONS);
      } else {
        mgx_launch_world_x(e->prog_in_lds, e->lds_world, e->stream, e->d, dev_copy_world_x(e), pw, MGX_PH_ALL);
      }
      MGX_MARK(1); MGX_MARK(2); MGX_MARK(3);
    }
    MGX_TRACE_POINT(e, "world kernel");
    if (e->world_chained) HIP_TRY(hipEventRecord(e->world_done, e->stream));
  }
  HIP_TRY(hipGetLastError());
  int rc = launch_obs(e, true);   // (+ ownership map refresh and query-backed obs values in front of it)
  if (rc) return rc;
  MGX_TRACE_POINT(e, "obs kernel");
  MGX_MARK(4);
  if (e->rewards_ext) { mgx_launch_values(e->stream, e->d, dev_copy(e), 1, nullptr); HIP_TRY(hipGetLastError()); }
  MGX_MARK(5);
#undef MGX_MARK
  if (e->profiling) e->timing_valid = true;
  if (e->auto_reset) {
    e->step_seq++;
#ifdef MGX_CPU_EMU
    const unsigned bt = 1;  // the sanitizer build runs work-items one after another: one env per workgroup
#else
    const unsigned bt = MGX_EPEND_THREADS;
#endif
    hipLaunchKernelGGL(mgx_episode_end_kernel, dim3((d.E + bt - 1) / bt), dim3(bt), 0, e->stream, dev_copy(e), (const uint32_t*)e->d_early,
                       (const uint32_t*)e->d_episodes, e->d_next_mask, e->d_counters, (volatile uint32_t*)e->h_flags_dev, e->step_seq,
                       e->d_counters + 1, e->d_done_list, e->d_done_n);
    HIP_TRY(hipGetLastError());
    if (e->ep_stats) { int erc = launch_episode_stats(e); if (erc) return erc; }
  }
  if (e->mem_kind == MGX_MEM_HOST) {
    HIP_TRY(hipMemcpyAsync(e->h_obs, d.obs, rows * d.T * 3, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(e->h_term, d.terminals, rows, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(e->h_trunc, d.truncations, rows, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(e->h_rew, d.rewards, rows * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  return MGX_OK;
}

int mgx_sync(mgx_engine* e) {
  if (!e) return fail(MGX_ERR_BAD_ARG, "mgx_sync: null engine");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MGX_OK;
}

void* mgx_stream(mgx_engine* e) { return e ? (void*)e->stream : nullptr; }

int mgx_chain_world(mgx_engine* e, mgx_engine* after) {
  if (!e || e == after) return fail(MGX_ERR_BAD_ARG, "mgx_chain_world: needs two different engines");
  if (after && after->device != e->device) return fail(MGX_ERR_BAD_ARG, "mgx_chain_world: engines live on different devices");
  e->world_after = after;
  if (after) after->world_chained = true;
  return MGX_OK;
}

int mgx_wait_before_outputs(mgx_engine* e, void* hip_event) {
  if (!e) return fail(MGX_ERR_BAD_ARG, "mgx_wait_before_outputs: null engine");
  if (e->out_fence && hip_event && e->out_fence != (hipEvent_t)hip_event) {
    // a second consumer before the next writer: the first one's event is not dropped — the engine's stream waits for it now
    // (coarser than needed: the world update of the next step waits too)
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamWaitEvent(e->stream, e->out_fence, 0));
  }
  e->out_fence = (hipEvent_t)hip_event;
  return MGX_OK;
}

int mgx_get_buffers(mgx_engine* e, uint8_t** observations, uint8_t** terminals, uint8_t** truncations,
                    float** rewards, int32_t** actions, int32_t** vibe_actions, int32_t* mem_kind) {
  if (!e) return fail(MGX_ERR_BAD_ARG, "mgx_get_buffers: null engine");
  bool host = e->mem_kind == MGX_MEM_HOST;
  if (observations) *observations = host ? e->h_obs : e->d.obs;
  if (terminals) *terminals = host ? e->h_term : e->d.terminals;
  if (truncations) *truncations = host ? e->h_trunc : e->d.truncations;
  if (rewards) *rewards = host ? e->h_rew : e->d.rewards;
  if (actions) *actions = host ? e->h_act : (int32_t*)e->d.actions;
  if (vibe_actions) *vibe_actions = host ? e->h_vact : (int32_t*)e->d.vibe_actions;
  if (mem_kind) *mem_kind = e->mem_kind;
  return MGX_OK;
}

static int d2h(mgx_engine* e, void* dst, const void* src, size_t bytes) {
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MGX_OK;
}

int mgx_get_episode_rewards(mgx_engine* e, float* out) {
  if (!e || !out) return fail(MGX_ERR_BAD_ARG, "mgx_get_episode_rewards: null argument");
  return d2h(e, out, e->d.episode_rewards, (size_t)e->d.E * e->d.A * 4);
}
int mgx_get_action_success(mgx_engine* e, uint8_t* out) {
  if (!e || !out) return fail(MGX_ERR_BAD_ARG, "mgx_get_action_success: null argument");
  return d2h(e, out, e->d.success, (size_t)e->d.E * e->d.A);
}
int mgx_get_current_steps(mgx_engine* e, uint32_t* out) {
  if (!e || !out) return fail(MGX_ERR_BAD_ARG, "mgx_get_current_steps: null argument");
  return d2h(e, out, e->d.step, (size_t)e->d.E * 4);
}

int mgx_get_stats(mgx_engine* e, int32_t env, float* game_values, uint8_t* game_touched, float* agent_values,
                  uint8_t* agent_touched) {
  if (!e || env < 0 || env >= e->d.E) return fail(MGX_ERR_BAD_ARG, "mgx_get_stats: bad env");
  const MgxDev& d = e->d;
  std::vector<uint32_t> gt(d.NGW), at((size_t)d.A * d.NSW);
  int rc = flush_shadow_all(e);
  if (rc) return rc;
  rc = d2h(e, game_values, d.game_stats + (size_t)env * d.NG, (size_t)d.NG * 4);
  if (rc) return rc;
  rc = d2h(e, gt.data(), d.game_touched + (size_t)env * d.NGW, (size_t)d.NGW * 4);
  if (rc) return rc;
  {  // device rows have pitch NSP; the caller's array is [A][NS]
    std::vector<float> rows((size_t)d.A * d.NSP);
    rc = d2h(e, rows.data(), d.ag_stats + (size_t)env * d.A * d.NSP, rows.size() * 4);
    if (rc) return rc;
    for (int a = 0; a < d.A; a++) memcpy(agent_values + (size_t)a * d.NS, rows.data() + (size_t)a * d.NSP, (size_t)d.NS * 4);
  }
  rc = d2h(e, at.data(), d.ag_touched + (size_t)env * d.A * d.NSW, (size_t)d.A * d.NSW * 4);
  if (rc) return rc;
  for (int i = 0; i < d.NG; i++) game_touched[i] = ((gt[i >> 5] >> (i & 31)) & 1u) | (game_values[i] != 0.f);
  for (int a = 0; a < d.A; a++)
    for (int i = 0; i < d.NS; i++) agent_touched[(size_t)a * d.NS + i] =
          ((at[(size_t)a * d.NSW + (i >> 5)] >> (i & 31)) & 1u) | (agent_values[(size_t)a * d.NS + i] != 0.f);
  return MGX_OK;
}

int mgx_get_invalid_index_extra(mgx_engine* e, int32_t env, int32_t* k_out, float* n_out) {
  if (!e || !k_out || !n_out || env < 0 || env >= e->d.E) return fail(MGX_ERR_BAD_ARG, "mgx_get_invalid_index_extra: bad argument");
  const MgxDev& d = e->d;
  const size_t n = (size_t)d.A * MGX_INVALID_EXTRA;
  int rc = d2h(e, k_out, d.ag_invk + (size_t)env * n, n * 4);
  if (rc) return rc;
  return d2h(e, n_out, d.ag_invn + (size_t)env * n, n * 4);
}

int mgx_get_objects(mgx_engine* e, int32_t env, int32_t* out, int32_t* n_objects) {
  if (!e || !out || !n_objects || env < 0 || env >= e->d.E) return fail(MGX_ERR_BAD_ARG, "mgx_get_objects: bad argument");
  const MgxDev& d = e->d;
  const size_t S = d.S;
  std::vector<uint16_t> cls(S), rc(S), inv(S * MGX_INV_PITCH);
  std::vector<uint8_t> vibe(S), agent(S), oflags(S, 0);
  std::vector<unsigned long long> ord(S);
  std::vector<uint32_t> tags(d.obj_tags ? S * MGX_TAG_WORDS : 0);
  uint32_t n = 0;
  int r = d2h(e, &n, d.num_objs + env, 4);
  if (!r) r = d2h(e, cls.data(), d.obj_cls + env * S, S * 2);
  if (!r) r = d2h(e, rc.data(), d.obj_rc + env * S, S * 2);
  if (!r) r = d2h(e, vibe.data(), d.obj_vibe + env * S, S);
  if (!r) r = d2h(e, agent.data(), d.obj_agent + env * S, S);
  if (!r) r = d2h(e, inv.data(), d.obj_inv + env * S * MGX_INV_PITCH, S * MGX_INV_PITCH * 2);
  if (!r) r = d2h(e, ord.data(), d.obj_order + env * S, S * 8);
  if (!r && d.obj_flags) r = d2h(e, oflags.data(), d.obj_flags + env * S, S);
  if (!r && d.obj_tags) r = d2h(e, tags.data(), d.obj_tags + env * S * MGX_TAG_WORDS, S * MGX_TAG_WORDS * 4);
  if (r) return r;
  for (uint32_t s = 0; s < n; s++) {
    int32_t* w = out + (size_t)s * MGX_OBJ_RECORD_WORDS;
    w[0] = (int)s + 1; w[1] = cls[s]; w[2] = rc[s] >> 8; w[3] = rc[s] & 0xFF; w[4] = vibe[s];
    w[5] = cls[s] != MGX_DEAD_CLASS && !(oflags[s] & 1); w[6] = agent[s] == MGX_NO_AGENT ? -1 : agent[s];
    int cnt = 0;
    for (int k = 0; k < MGX_MAX_RESOURCES; k++) {
      int item = (int)((ord[s] >> (4 * k)) & 0xF);
      if (item == 0xF) { for (int q = k; q < MGX_MAX_RESOURCES; q++) w[8 + q] = -1; break; }
      w[8 + k] = item;
      cnt++;
    }
    w[7] = cnt;
    for (int k = 0; k < MGX_MAX_RESOURCES; k++) w[8 + MGX_MAX_RESOURCES + k] = k < d.R ? inv[s * MGX_INV_PITCH + k] : 0;
    for (int k = 0; k < MGX_TAG_WORDS; k++)
      w[8 + 2 * MGX_MAX_RESOURCES + k] = d.obj_tags ? (int32_t)tags[s * MGX_TAG_WORDS + k]
                                                    : (cls[s] != MGX_DEAD_CLASS ? e->prog[d.sec[MGX_SEC_CLASSES] + cls[s] * MGX_C_WORDS + MGX_C_TAGS + k] : 0);
  }
  *n_objects = (int32_t)n;
  return MGX_OK;
}

int mgx_get_objects_batch(mgx_engine* e, const int32_t* envs, int32_t n_envs, int32_t* out, int32_t* n_objects) {
  if (!e || !envs || !out || !n_objects || n_envs <= 0) return fail(MGX_ERR_BAD_ARG, "mgx_get_objects_batch: bad argument");
  const MgxDev& d = e->d;
  for (int i = 0; i < n_envs; i++)
    if (envs[i] < 0 || envs[i] >= d.E) return fail(MGX_ERR_BAD_ARG, "mgx_get_objects_batch: env index out of range");
  HIP_TRY(hipSetDevice(e->device));
  const size_t rec = (size_t)d.S * MGX_OBJ_RECORD_WORDS, n = (size_t)n_envs;
  if (n > e->objs_cap) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->d_objs) (void)hipFree(e->d_objs);
    e->d_objs = nullptr;
    e->objs_cap = 0;
    HIP_TRY(hipMalloc((void**)&e->d_objs, n * (2 + rec) * 4));
    e->objs_cap = n;
  }
  int32_t* d_envs = e->d_objs;
  int32_t* d_counts = e->d_objs + e->objs_cap;
  int32_t* d_recs = e->d_objs + 2 * e->objs_cap;
  HIP_TRY(hipMemcpyAsync(d_envs, envs, n * 4, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(mgx_objects_kernel, dim3((unsigned)n), dim3(256), 0, e->stream, dev_copy(e), (const int32_t*)d_envs, d_recs, d_counts);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(n_objects, d_counts, n * 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(out, d_recs, n * rec * 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MGX_OK;
}

int mgx_get_reward_state(mgx_engine* e, int32_t env, float* out) {
  if (!e || !out || env < 0 || env >= e->d.E) return fail(MGX_ERR_BAD_ARG, "mgx_get_reward_state: bad argument");
  const MgxDev& d = e->d;
  std::vector<float> prev((size_t)d.A * d.NRW);
  std::vector<uint16_t> ag_obj(d.A), cls(d.S);
  int r = d2h(e, prev.data(), d.ag_rprev + (size_t)env * d.A * d.NRW, prev.size() * 4);
  if (!r) r = d2h(e, ag_obj.data(), d.ag_obj + (size_t)env * d.A, (size_t)d.A * 2);
  if (!r) r = d2h(e, cls.data(), d.obj_cls + (size_t)env * d.S, (size_t)d.S * 2);
  if (r) return r;
  for (int a = 0; a < d.A; a++) {
    const uint16_t c = ag_obj[a] < d.S ? cls[ag_obj[a]] : (uint16_t)MGX_DEAD_CLASS;
    int n = c == MGX_DEAD_CLASS ? 0 : e->prog[d.sec[MGX_SEC_CLASSES] + c * MGX_C_WORDS + MGX_C_REWARD_COUNT];
    float t = 0.f;
    for (int k = 0; k < n; k++) t += prev[(size_t)a * d.NRW + k];
    out[a] = t;
  }
  return MGX_OK;
}

int mgx_set_inventory(mgx_engine* e, int32_t env, int32_t agent_id, const int32_t* items, const int32_t* amounts, int32_t n) {
  if (!e || env < 0 || env >= e->d.E || n < 0 || (n > 0 && (!items || !amounts)))
    return fail(MGX_ERR_BAD_ARG, "mgx_set_inventory: bad argument");
  if (agent_id < 0 || agent_id >= e->d.A) return MGX_OK;  // the reference ignores unknown agent ids (mettagrid_py.cpp:205)
  for (int i = 0; i < n; i++)
    if (items[i] < 0 || items[i] >= e->d.R || amounts[i] < 0 || amounts[i] > 65535)
      return fail(MGX_ERR_BAD_ARG, "mgx_set_inventory: resource id or amount out of range");
  HIP_TRY(hipSetDevice(e->device));
  int32_t* dbuf = nullptr;
  if (n > 0) {
    HIP_TRY(hipMalloc((void**)&dbuf, (size_t)n * 8));
    HIP_TRY(hipMemcpyAsync(dbuf, items, (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(dbuf + n, amounts, (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
  }
  hipLaunchKernelGGL(mgx_set_inventory_kernel, dim3(1), dim3(1), 0, e->stream, dev_copy(e), env, agent_id,
                     (const int32_t*)dbuf, (const int32_t*)(dbuf + n), n);
  hipError_t le = hipGetLastError();
  hipError_t se = hipStreamSynchronize(e->stream);
  if (dbuf) (void)hipFree(dbuf);
  if (le != hipSuccess || se != hipSuccess) return fail(MGX_ERR_HIP, "mgx_set_inventory: kernel failed");
  return MGX_OK;
}

int mgx_count_objects_with_tag(mgx_engine* e, int32_t env, int32_t tag_id, int32_t* out) {
  if (!e || !out || env < 0 || env >= e->d.E) return fail(MGX_ERR_BAD_ARG, "mgx_count_objects_with_tag: bad argument");
  *out = 0;
  if (tag_id < 0 || tag_id >= 256) return MGX_OK;
  const MgxDev& d = e->d;
  const size_t S = d.S;
  uint32_t n = 0;
  std::vector<uint16_t> cls(S);
  std::vector<uint8_t> oflags(S, 0);
  std::vector<uint32_t> tags(d.obj_tags ? S * MGX_TAG_WORDS : 0);
  int r = d2h(e, &n, d.num_objs + env, 4);
  if (!r) r = d2h(e, cls.data(), d.obj_cls + env * S, S * 2);
  if (!r && d.obj_flags) r = d2h(e, oflags.data(), d.obj_flags + env * S, S);
  if (!r && d.obj_tags) r = d2h(e, tags.data(), d.obj_tags + env * S * MGX_TAG_WORDS, S * MGX_TAG_WORDS * 4);
  if (r) return r;
  int count = 0;
  for (uint32_t s = 0; s < n; s++) {
    if (cls[s] == MGX_DEAD_CLASS || (oflags[s] & 1)) continue;  // removed objects are unregistered (tag_index.cpp:21-31)
    const uint32_t w = d.obj_tags ? tags[s * MGX_TAG_WORDS + (tag_id >> 5)]
                                  : (uint32_t)e->prog[d.sec[MGX_SEC_CLASSES] + cls[s] * MGX_C_WORDS + MGX_C_TAGS + (tag_id >> 5)];
    count += (w >> (tag_id & 31)) & 1u;
  }
  *out = count;
  return MGX_OK;
}

int mgx_state_digests(mgx_engine* e, uint64_t* out) {
  if (!e || !out) return fail(MGX_ERR_BAD_ARG, "mgx_state_digests: null argument");
  HIP_TRY(hipSetDevice(e->device));
  if (!e->d_digest) { int rc = e->alloc(&e->d_digest, (size_t)e->d.E); if (rc) return rc; }
  { int rc = flush_shadow_all(e); if (rc) return rc; }
  hipLaunchKernelGGL(mgx_digest_kernel, dim3((unsigned)e->d.E), dim3(MGX_WAVE), 0, e->stream, dev_copy(e), e->d_digest);
  HIP_TRY(hipGetLastError());
  return d2h(e, out, e->d_digest, (size_t)e->d.E * 8);
}

// joint action id -> (primary, vibe) action indices (mettagrid_puffer_env.py:331-381), one thread per agent row
// ... and the range checks MettaGridPufferEnv.step makes on the host (:336-360: negative ids, ids >= num_primary * (num_vibe + 1)):
// a bad id sets a flag word in host-visible memory (bit 0 negative, bit 1 too large) — read by mgx_poll_action_errors without
// waiting for the device — and keeps the lowest offending (row, value) in device memory.
__global__ void __launch_bounds__(256) mgx_joint_actions_kernel(const int32_t* __restrict__ joint, int32_t* __restrict__ actions,
                                                                int32_t* __restrict__ vibe_actions, const int32_t* __restrict__ vibe_ids,
                                                                long long rows, int num_primary, int num_vibe,
                                                                unsigned long long* __restrict__ first_bad, uint32_t* __restrict__ host_err) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows) return;
  const int a = joint[i];
  const int max_valid = num_vibe > 0 ? num_primary + num_primary * num_vibe : num_primary;
  if (a < 0 || a >= max_valid) {
    const unsigned long long key = ((unsigned long long)i << 32) | (uint32_t)a;
    atomicMin(first_bad, key);   // the lowest bad row (fetched by the host once it has seen the flag)
    atomicOr(host_err, a < 0 ? 1u : 2u);
  }
  int core = a, vibe = 0;
  if (num_vibe > 0 && a >= num_primary) {
    const int off = a - num_primary;
    core = off / num_vibe;
    vibe = vibe_ids[off % num_vibe];
  }
  actions[i] = core;
  vibe_actions[i] = vibe;
}

int mgx_set_joint_actions(mgx_engine* e, const int32_t* joint, int32_t num_primary, const int32_t* vibe_ids, int32_t num_vibe) {
  if (!e || !joint || num_primary <= 0 || num_vibe < 0 || num_vibe > 256 || (num_vibe > 0 && !vibe_ids))
    return fail(MGX_ERR_BAD_ARG, "mgx_set_joint_actions: bad argument");
  if (e->mem_kind != MGX_MEM_DEVICE) return fail(MGX_ERR_BAD_ARG, "mgx_set_joint_actions: needs device buffers");
  HIP_TRY(hipSetDevice(e->device));
  const MgxDev& d = e->d;
  if (!e->d_vibe_ids) { int rc = e->alloc(&e->d_vibe_ids, 256); if (rc) return rc; }
  if (num_vibe > 0 && (e->n_vibe_ids != num_vibe || memcmp(e->vibe_ids_host, vibe_ids, (size_t)num_vibe * 4) != 0)) {
    memcpy(e->vibe_ids_host, vibe_ids, (size_t)num_vibe * 4);
    e->n_vibe_ids = num_vibe;
    HIP_TRY(hipMemcpyAsync(e->d_vibe_ids, e->vibe_ids_host, (size_t)num_vibe * 4, hipMemcpyHostToDevice, e->stream));
  }
  const long long rows = (long long)d.E * d.A;
  if (!e->h_act_err) {
    HIP_TRY(hipHostMalloc((void**)&e->h_act_err, 16, hipHostMallocMapped));
    memset(e->h_act_err, 0, 16);
    HIP_TRY(hipHostGetDevicePointer((void**)&e->h_act_err_dev, e->h_act_err, 0));
    int rc = e->alloc(&e->d_first_bad, 1, 0xFF);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(mgx_joint_actions_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, e->stream, joint, (int32_t*)d.actions, (int32_t*)d.vibe_actions,
                     (const int32_t*)e->d_vibe_ids, rows, num_primary, num_vibe, e->d_first_bad, e->h_act_err_dev);
  HIP_TRY(hipGetLastError());
  return MGX_OK;
}

int mgx_poll_action_errors(mgx_engine* e, int32_t wait, uint32_t* flags, int64_t* row, int32_t* value) {
  if (!e || !flags) return fail(MGX_ERR_BAD_ARG, "mgx_poll_action_errors: null argument");
  *flags = 0;
  if (!e->h_act_err) return MGX_OK;   // mgx_set_joint_actions has not run
  if (wait) { HIP_TRY(hipSetDevice(e->device)); HIP_TRY(hipStreamSynchronize(e->stream)); }
  const uint32_t f = ((volatile uint32_t*)e->h_act_err)[0];
  if (!f) return MGX_OK;
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));   // (an error is the slow path: report the lowest bad row of everything enqueued)
  *flags = ((volatile uint32_t*)e->h_act_err)[0];
  unsigned long long key = 0;
  HIP_TRY(hipMemcpy(&key, e->d_first_bad, 8, hipMemcpyDeviceToHost));
  if (row) *row = (int64_t)(key >> 32);
  if (value) *value = (int32_t)(uint32_t)key;
  memset(e->h_act_err, 0, 16);
  HIP_TRY(hipMemsetAsync(e->d_first_bad, 0xFF, 8, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MGX_OK;
}

int mgx_decode_obs(mgx_engine* e, const uint8_t* tokens, int64_t n_rows, float* box, int32_t num_features, const float* scale) {
  if (!e || !box || !scale || num_features < 1 || num_features > 256) return fail(MGX_ERR_BAD_ARG, "mgx_decode_obs: bad argument");
  HIP_TRY(hipSetDevice(e->device));
  const MgxDev& d = e->d;
  if (!tokens) { tokens = d.obs; n_rows = (int64_t)d.E * d.A; }
  if (n_rows <= 0) return fail(MGX_ERR_BAD_ARG, "mgx_decode_obs: no rows");
  if (!e->d_scale) { int rc = e->alloc(&e->d_scale, 256); if (rc) return rc; }
  if (!e->scale_valid || memcmp(e->scale_host, scale, sizeof e->scale_host) != 0) {
    memcpy(e->scale_host, scale, sizeof e->scale_host);
    HIP_TRY(hipMemcpyAsync(e->d_scale, e->scale_host, sizeof e->scale_host, hipMemcpyHostToDevice, e->stream));
    e->scale_valid = true;
  }
  const int rc = mgx_launch_decode(e->stream, tokens, box, e->d_scale, (long long)n_rows, d.T, num_features, e->prog[MGX_H_OBS_HEIGHT],
                                   e->prog[MGX_H_OBS_WIDTH]);
  if (rc == -1) return fail(MGX_ERR_PROGRAM, "mgx_decode_obs: the box of one agent does not fit the decode kernel's LDS");
  if (rc) return fail(MGX_ERR_HIP, "mgx_decode_obs: launch failed");
  return MGX_OK;
}

int mgx_set_box_output(mgx_engine* e, void* box, int32_t dtype, int32_t num_features, const float* scale) {
  if (!e) return fail(MGX_ERR_BAD_ARG, "mgx_set_box_output: null engine");
  if (dtype == MGX_BOX_OFF || !box) { e->box_dtype = MGX_BOX_OFF; e->box_out = nullptr; return MGX_OK; }
  if ((dtype != MGX_BOX_F32 && dtype != MGX_BOX_BF16) || !scale || num_features < 1 || num_features > 256)
    return fail(MGX_ERR_BAD_ARG, "mgx_set_box_output: bad argument");
#ifdef MGX_CPU_EMU
  return fail(MGX_ERR_PROGRAM, "mgx_set_box_output: the observation kernel is not part of the sanitizer build");
#endif
  HIP_TRY(hipSetDevice(e->device));
  if (!e->d_scale) { int rc = e->alloc(&e->d_scale, 256); if (rc) return rc; }
  if (!e->scale_valid || memcmp(e->scale_host, scale, sizeof e->scale_host) != 0) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    memcpy(e->scale_host, scale, sizeof e->scale_host);
    HIP_TRY(hipMemcpyAsync(e->d_scale, e->scale_host, sizeof e->scale_host, hipMemcpyHostToDevice, e->stream));
    e->scale_valid = true;
  }
  e->box_out = box; e->box_dtype = dtype; e->box_C = num_features;
  return MGX_OK;   // (from the next observation pass on; the box of the observations already made: mgx_decode_obs)
}

int mgx_poll_errors(mgx_engine* e, uint32_t* bits, int32_t* first_env) {
  if (!e || !bits) return fail(MGX_ERR_BAD_ARG, "mgx_poll_errors: null argument");
  std::vector<uint32_t> err(e->d.E);
  int r = d2h(e, err.data(), e->d.err, (size_t)e->d.E * 4);
  if (r) return r;
  uint32_t all = 0;
  int first = -1;
  for (int i = 0; i < e->d.E; i++) {
    if (err[i] && first < 0) first = i;
    all |= err[i];
  }
  *bits = all;
  if (first_env) *first_env = first;
  return MGX_OK;
}

int mgx_set_profiling(mgx_engine* e, int32_t enabled) {
  if (!e) return fail(MGX_ERR_BAD_ARG, "mgx_set_profiling: null engine");
  e->profiling = enabled != 0;
  e->timing_valid = false;
  return MGX_OK;
}
int mgx_get_step_timing(mgx_engine* e, float* ms_out) {
  if (!e || !ms_out) return fail(MGX_ERR_BAD_ARG, "mgx_get_step_timing: null argument");
  if (!e->profiling) return fail(MGX_ERR_BAD_ARG, "mgx_get_step_timing: profiling is off");
  if (!e->timing_valid) {  // no step since profiling was switched on: nothing recorded yet
    for (int k = 0; k < MGX_T_COUNT; k++) ms_out[k] = 0.f;
    return MGX_OK;
  }
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipEventSynchronize(e->ev[MGX_T_COUNT]));
  for (int k = 0; k < MGX_T_COUNT; k++) HIP_TRY(hipEventElapsedTime(&ms_out[k], e->ev[k], e->ev[k + 1]));
  return MGX_OK;
}

int32_t mgx_obs_variant(const mgx_engine* e) { return e ? e->obs_variant : 0; }
int32_t mgx_act_variant(const mgx_engine* e) { return e ? e->d.act_par : 0; }
int32_t mgx_handler_variant(const mgx_engine* e) { return e ? e->d.gen_prog : 0; }
int32_t mgx_world_prog_in_lds(const mgx_engine* e) { return e && e->prog_in_lds ? 1 : 0; }
int32_t mgx_is_extended(const mgx_engine* e) { return e && e->d.X ? 1 : 0; }
int32_t mgx_dispatch_pairs(const mgx_engine* e) { return e && e->d.duo ? 1 : 0; }
int32_t mgx_integer_bookkeeping(const mgx_engine* e) { return e ? e->d.shadow : 0; }
int32_t mgx_num_envs(const mgx_engine* e) { return e ? e->d.E : 0; }
int32_t mgx_num_agents(const mgx_engine* e) { return e ? e->d.A : 0; }
int32_t mgx_num_tokens(const mgx_engine* e) { return e ? e->d.T : 0; }
int64_t mgx_state_bytes(const mgx_engine* e) { return e ? e->state_bytes : 0; }

}  // extern "C"

#ifdef MGX_WORLD_TIMING  // instrumented developer build only (scripts/world_timing.py); not part of the ABI
extern "C" void mgx_debug_obs_cycles(unsigned long long* out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(mgx_dbg_cycles), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(mgx_dbg_cycles), z, sizeof z); }
}
#endif
