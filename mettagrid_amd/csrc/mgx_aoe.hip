// mgx_aoe.hip — area effects with one lane per AGENT: fixed AoE, territory effects and mobile AoE of _step
// (/root/reference/cpp/bindings/mettagrid_c.cpp:1032-1042) for games whose AoE / territory handlers only touch their
// target (MgxLocalAgent, mgx_aoe_local.h; the host proves that from the program, mgx_aoe_is_target_local below).
// One wavefront per env, lane = agent index: 65 536 envs x 64 agents are 65 536 independent wavefronts instead of 1 024
// wavefronts each walking 64 agents x (16 fixed + 8 territory + 64 mobile sources) serially.  One flat kernel: the
// filters / mutations these records may use need neither the handler VM nor queries.
#define MGX_BIG __forceinline__
#define MGX_OUTLINE __forceinline__
#define MGX_WORLD_FAST_TU 1
#define MGX_TU_NS mgx_tu_aoe
#define MGX_HOT_PROG 1   // an LDS program copy holds the hot sections only; class records come from HBM / L2 (mgx_world.h)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "mgx_device.h"
#include "mgx_world.h"
#include "mgx_aoe_local.h"

#ifdef MGX_CPU_EMU
#define MGX_AOE_OCCUPANCY
#else
#ifndef MGX_AOE_WPE
#define MGX_AOE_WPE 3, 4
#endif
#define MGX_AOE_OCCUPANCY __attribute__((amdgpu_waves_per_eu(MGX_AOE_WPE)))
#endif
// One wavefront per env, lane = agent (MgxLocalAgent, mgx_aoe_local.h).  Between the pieces of the phase the lanes of the
// wavefront stage the env's source records in LDS, 64 at a time: s_pack (location | radius | live), s_info (object | AoE
// record << 16) for the fixed and for the mobile sources.
// PROG_LDS: the hot program range (limits .. territory controls, a few KB) is first copied into LDS: the interpreter's
// record reads — handler, filter atoms, tag masks, mutations, limits — are chains of dependent loads, 60 cycles each from
// LDS instead of an L2 round trip.
template <bool PROG_LDS>
__global__ void __launch_bounds__(MGX_AOE_THREADS) MGX_AOE_OCCUPANCY mgx_aoe_kernel(const MgxDev* __restrict__ dp, int prog_words) {
  const MgxDev& d = *dp;  // per-engine copy in device memory (see mgx_world_x.hip)
  extern __shared__ __align__(16) uint8_t aoe_lds[];
  const int tid = (int)threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid / MGX_WAVE), lane = tid & (MGX_WAVE - 1);
  const int env = (int)blockIdx.x * MGX_AOE_WAVES + wave;
  uint8_t* s_map = aoe_lds;
  int16_t* s_ids = (int16_t*)(aoe_lds + 256);
  uint32_t* s_stage = (uint32_t*)(aoe_lds + MGX_AOE_HDR) + wave * 4 * MGX_WAVE;
  float* s_val = (float*)(aoe_lds + MGX_AOE_HDR + MGX_AOE_WAVES * 4 * MGX_WAVE * 4) + tid;
  int* s_def = (int*)(s_val - tid + d.aoe_nstat * MGX_AOE_THREADS) + tid;
  int32_t* s_prog = (int32_t*)(aoe_lds + mgx_aoe_lds_bytes(d.aoe_nstat));
#ifdef MGX_CPU_EMU
  for (int i = 0; i < 256; i++) s_map[i] = d.aoe_stat_map[i];   // (work-items run one after another in the sanitizer build)
  for (int i = 0; i < d.aoe_nstat; i++) s_ids[i] = d.aoe_stat_ids[i];
#else
  for (int i = tid; i < 256; i += MGX_AOE_THREADS) s_map[i] = d.aoe_stat_map[i];
  if (tid < d.aoe_nstat) s_ids[tid] = d.aoe_stat_ids[tid];
  if constexpr (PROG_LDS) {
    const int4* src = (const int4*)(d.P + d.hot_lo);
    for (int i = tid; i < prog_words / 4; i += MGX_AOE_THREADS) ((int4*)s_prog)[i] = src[i];
  }
  __syncthreads();
#endif
  if (env >= d.E) return;
  MGX_TICK0();
  typedef typename std::conditional<PROG_LDS, MgxLdsProg, MgxGlobalProg>::type PP;
  typedef MgxEnvT<PP, true> Env;
  PP prog;
  if constexpr (PROG_LDS) prog = (MgxLdsProg)s_prog;
  else prog = d.P;
  Env e(d, prog, env);
  e.step = d.step[env];
  const int nf = d.NF ? (int)d.fx_count[env] : 0, nm = d.NM ? (int)d.mb_count[env] : 0;
  const size_t fb = (size_t)env * d.NF, mb = (size_t)env * d.NM;
  uint32_t* s_pack = s_stage;
  uint32_t* s_info = s_stage + MGX_WAVE;
  // stage sources [q0, q0 + 64) of one list: lane q loads record q0 + q.  The packed location / radius record stays in the
  // loading lane's register (returned): the range test of every agent against source q reads it with v_readlane, so the
  // record is a scalar and its unpacking scalar arithmetic; the (object, AoE index) pairs, read by lane-dependent q later,
  // go to LDS.
  auto stage = [&](const uint32_t* pack, const uint16_t* obj, const uint16_t* aoe, size_t base, int q0, int n) -> uint32_t {
#ifdef MGX_CPU_EMU
    for (int q = 0; q < MGX_WAVE; q++)   // work-items run one after another here: each fills the whole staging itself
      if (q0 + q < n) { s_pack[q] = pack[base + q0 + q]; s_info[q] = (uint32_t)obj[base + q0 + q] | ((uint32_t)aoe[base + q0 + q] << 16); }
    return 0u;
#else
    __builtin_amdgcn_wave_barrier();     // the previous chunk has been read by every lane
    uint32_t mine = 0u;
    if (q0 + lane < n) {
      mine = pack[base + q0 + lane];
      s_info[lane] = (uint32_t)obj[base + q0 + lane] | ((uint32_t)aoe[base + q0 + lane] << 16);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return mine;
#endif
  };
  for (int a0 = 0; a0 < d.A; a0 += MGX_WAVE) {   // (more than 64 agents per env: one pass per 64)
    const int ai = a0 + lane;
    const bool valid = ai < d.A;
    MgxLocalAgent<Env> ag(e);
    if (valid) {
      ag.load(ai, s_val, s_map, s_ids, s_def);
      MGX_TICK(0);
      if (d.tick_in_aoe) ag.on_tick();
    }
    MGX_TICK(1);
    if (nf > 0) {
      for (int pass = 0; pass < 2; pass++)
        for (int f0 = 0; f0 < nf; f0 += MGX_WAVE) {
          const uint32_t mine = stage(d.fx_pack, d.fx_obj, d.fx_aoe, fb, f0, nf);
          ag.fixed_chunk(valid, pass, f0, min(MGX_WAVE, nf - f0), mine, s_pack, s_info);
        }
      if (valid) ag.fixed_finish();
    }
    MGX_TICK(2);
    if (valid && d.NT > 0) ag.territory();
    MGX_TICK(3);
    for (int m0 = 0; m0 < nm; m0 += MGX_WAVE) {
      const uint32_t mine = stage(d.mb_pack, d.mb_obj, d.mb_aoe, mb, m0, nm);
      ag.mobile_chunk(valid, m0, min(MGX_WAVE, nm - m0), mine, s_pack, s_info);
    }
    MGX_TICK(4);
    if (valid) {
      if (d.cov_in_aoe) ag.coverage();
      MGX_TICK(5);
      ag.store();
    }
    MGX_TICK(6);
  }
}

// One thread per (env, registered AoE source): the packed (location, radius, live) record the lanes scan.
__global__ void __launch_bounds__(256) mgx_aoe_prep_kernel(const MgxDev* __restrict__ dp) {
  const MgxDev& d = *dp;
  const int per = d.NF + d.NM;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)d.E * per) return;
  MgxEnvT<MgxGlobalProg, true> e(d, d.P, (int)(idx / per));
  mgx_aoe_pack_source(e, (int)(idx % per));
}

void mgx_launch_aoe(hipStream_t stream, const MgxDev& d, const MgxDev* dp, const MgxDev* hot, int prog_words) {
  const long long sources = (long long)d.E * (d.NF + d.NM);
  if (sources > 0) hipLaunchKernelGGL(mgx_aoe_prep_kernel, dim3((unsigned)((sources + 255) / 256)), dim3(256), 0, stream, dp);
  const dim3 grid((d.E + MGX_AOE_WAVES - 1) / MGX_AOE_WAVES), block(MGX_AOE_THREADS);
  const size_t lds = (size_t)mgx_aoe_lds_bytes(d.aoe_nstat) + (hot ? (size_t)prog_words * 4 : 0);
  if (hot) hipLaunchKernelGGL((mgx_aoe_kernel<true>), grid, block, lds, stream, hot, prog_words);
  else hipLaunchKernelGGL((mgx_aoe_kernel<false>), grid, block, lds, stream, dp, 0);
}
bool mgx_aoe_set_lds(int nstat, int prog_words) {   // dynamic LDS past 64 KB needs the opt-in attribute
#ifdef MGX_CPU_EMU
  (void)nstat; (void)prog_words;
  return true;
#else
  const int lds = mgx_aoe_lds_bytes(nstat) + prog_words * 4;
  return hipFuncSetAttribute((const void*)mgx_aoe_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess &&
         hipFuncSetAttribute((const void*)mgx_aoe_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
#endif
}

// Host analysis: true when every AoE record and every territory handler of the program is "target-local" — its filters
// read only the target's mutable state (and tags / positions / the period, which nothing changes during the phase) and
// its mutations write only the target's inventory, vibe or agent stats.  Then agents are independent in the AoE phase.
static bool code_is_local(const int32_t* P, int start, int count) {
  for (int i = 0; i < count; i++) {
    const int32_t* ins = P + mgx_sec_off(P, MGX_SEC_GV_CODE) + (start + i) * MGX_GV_WORDS;
    switch (ins[MGX_GV_OP]) {
      case MGX_GOP_INVENTORY: case MGX_GOP_CONST: case MGX_GOP_ADD_TERM: case MGX_GOP_RATIO: case MGX_GOP_MAX2: case MGX_GOP_MIN2: break;
      case MGX_GOP_STAT: if (ins[MGX_GV_A0] != 0) return false; break;  // game-scope stats are shared by the env's agents
      default: return false;
    }
  }
  return true;
}
static bool value_is_local(const int32_t* P, int rec) {
  const int32_t* V = P + mgx_sec_off(P, MGX_SEC_OBS_VALUES) + rec * MGX_OV_WORDS;
  return code_is_local(P, V[MGX_OV_GV_START], V[MGX_OV_GV_COUNT]);
}
static bool filters_are_local(const int32_t* P, int pc0, bool& reads_actor_vibe) {
  if (pc0 < 0) return true;
  const int n = mgx_sec_cnt(P, MGX_SEC_ATOMS);
  std::vector<char> seen(n, 0);
  std::vector<int> todo{pc0};
  while (!todo.empty()) {
    const int pc = todo.back();
    todo.pop_back();
    if (pc < 0) continue;
    if (pc >= n) return false;
    if (seen[pc]) continue;
    seen[pc] = 1;
    const int32_t* a = P + mgx_sec_off(P, MGX_SEC_ATOMS) + pc * MGX_AT_WORDS;
    switch (a[MGX_AT_OP]) {
      case MGX_FOP_VIBE: if (a[MGX_AT_A0] != MGX_ENT_TARGET) reads_actor_vibe = true; break;
      case MGX_FOP_RESOURCE: if (a[MGX_AT_A0] != MGX_ENT_TARGET) return false; break;
      case MGX_FOP_SHARED_TAG: case MGX_FOP_TAG: case MGX_FOP_TARGET_LOC_EMPTY: case MGX_FOP_TARGET_IS_USABLE:
      case MGX_FOP_PERIODIC: case MGX_FOP_TRUE: case MGX_FOP_FALSE: break;
      case MGX_FOP_GAME_VALUE:
        if (a[MGX_AT_A0] != MGX_ENT_TARGET || !value_is_local(P, a[MGX_AT_A1]) || !value_is_local(P, a[MGX_AT_A2])) return false;
        break;
      case MGX_FOP_MAX_DISTANCE: if (a[MGX_AT_A2] >= 0) return false; break;  // the binary form only (no query)
      default: return false;
    }
    todo.push_back(a[MGX_AT_ON_TRUE]);
    todo.push_back(a[MGX_AT_ON_FALSE]);
  }
  return true;
}
static bool mutations_are_local(const int32_t* P, int start, int count, bool& changes_vibe) {
  for (int i = 0; i < count; i++) {
    const int32_t* m = P + mgx_sec_off(P, MGX_SEC_MUTS) + (start + i) * MGX_MU_WORDS;
    switch (m[MGX_MU_OP]) {
      case MGX_MOP_RESOURCE_DELTA: case MGX_MOP_CLEAR_INVENTORY: if (m[MGX_MU_A0] != MGX_ENT_TARGET) return false; break;
      case MGX_MOP_CHANGE_VIBE: if (m[MGX_MU_A0] != MGX_ENT_TARGET) return false; changes_vibe = true; break;
      case MGX_MOP_STATS:
        if (m[MGX_MU_A0] != 1 || m[MGX_MU_A1] != MGX_ENT_TARGET || !value_is_local(P, m[MGX_MU_A3])) return false;
        break;
      case MGX_MOP_GAME_VALUE: {
        if (m[MGX_MU_A0] != MGX_ENT_TARGET || !value_is_local(P, m[MGX_MU_A1]) || !value_is_local(P, m[MGX_MU_A2])) return false;
        break;
      }
      default: return false;
    }
  }
  return true;
}
static bool aoe_scan(const int32_t* P, bool with_on_tick);
bool mgx_aoe_is_target_local(const int32_t* P) { return aoe_scan(P, false); }
bool mgx_aoe_on_tick_local(const int32_t* P) { return aoe_scan(P, true); }
static bool aoe_scan(const int32_t* P, bool with_on_tick) {
  if (P[MGX_H_SPAWNS]) return false;  // deferred AoE registration of spawned objects stays with the serial form
  bool reads_actor_vibe = false, changes_vibe = false;
  const int na = mgx_sec_cnt(P, MGX_SEC_AOES);
  for (int i = 0; i < na; i++) {
    const int32_t* a = P + mgx_sec_off(P, MGX_SEC_AOES) + i * MGX_AO_WORDS;
    if (a[MGX_AO_RADIUS] < 0 || a[MGX_AO_RADIUS] > 255) return false;  // packed source records hold 8 bits
    if (!filters_are_local(P, a[MGX_AO_FILTER_PC], reads_actor_vibe)) return false;
    if (!mutations_are_local(P, a[MGX_AO_MUT_START], a[MGX_AO_MUT_COUNT], changes_vibe)) return false;
  }
  const int nt = mgx_sec_cnt(P, MGX_SEC_TERRITORIES);
  for (int t = 0; t < nt; t++) {
    const int32_t* TE = P + mgx_sec_off(P, MGX_SEC_TERRITORIES) + t * MGX_TE_WORDS;
    if (TE[MGX_TE_TAGS_COUNT] > 8) return false;
    const int lists[3][2] = {{TE[MGX_TE_ENTER_START], TE[MGX_TE_ENTER_COUNT]}, {TE[MGX_TE_EXIT_START], TE[MGX_TE_EXIT_COUNT]},
                             {TE[MGX_TE_PRES_START], TE[MGX_TE_PRES_COUNT]}};
    for (const auto& l : lists)
      for (int i = 0; i < l[1]; i++) {
        const int32_t* hd = P + mgx_sec_off(P, MGX_SEC_HANDLERS) + (l[0] + i) * MGX_HD_WORDS;
        if (!filters_are_local(P, hd[MGX_HD_FILTER_PC], reads_actor_vibe)) return false;
        if (!mutations_are_local(P, hd[MGX_HD_MUT_START], hd[MGX_HD_MUT_COUNT], changes_vibe)) return false;
      }
  }
  if (with_on_tick) {  // + the per-agent on_tick handlers, run by the same lanes in front of the area effects
    const int nc = P[MGX_H_NUM_CLASSES];
    for (int c = 0; c < nc; c++) {
      const int h = P[mgx_sec_off(P, MGX_SEC_CLASSES) + c * MGX_C_WORDS + MGX_C_ON_TICK];
      if (h < 0) continue;
      const int32_t* hd = P + mgx_sec_off(P, MGX_SEC_HANDLERS) + h * MGX_HD_WORDS;
      if (hd[MGX_HD_KIND] != MGX_HK_LEAF) return false;
      if (!filters_are_local(P, hd[MGX_HD_FILTER_PC], reads_actor_vibe)) return false;
      if (!mutations_are_local(P, hd[MGX_HD_MUT_START], hd[MGX_HD_MUT_COUNT], changes_vibe)) return false;
    }
  }
  return !(reads_actor_vibe && changes_vibe);  // a source's vibe may be another lane's target's vibe
}

// The agent stats the records of the lane-per-agent phase can read or write, in the order they get LDS cells (most
// likely first; at most MGX_AOE_MAX_STATS, ids below 256 — anything else stays in HBM, see MgxLocalAgent::sget / sset):
// gained / lost / amount of every resource a local mutation or presence delta names (Agent::on_inventory_change,
// objects/agent.cpp:106-121), the death counter, the targets of SetStat / game-value mutations, the agent stats local value
// code reads, and the two coverage stats when the kernel also tracks coverage.
void mgx_aoe_collect_stats(const int32_t* P, bool with_on_tick, bool with_coverage, std::vector<int16_t>& out) {
  const int R = P[MGX_H_NUM_RESOURCES];
  auto wk = [&](int k) { return P[MGX_H_STAT_BASE + k]; };
  uint32_t items = 0;
  std::vector<int> ids;
  auto add_id = [&](int id) { if (id >= 0 && id < 256 && std::find(ids.begin(), ids.end(), id) == ids.end()) ids.push_back(id); };
  auto code_reads = [&](int rec) {
    if (rec < 0) return;
    const int32_t* V = P + mgx_sec_off(P, MGX_SEC_OBS_VALUES) + rec * MGX_OV_WORDS;
    for (int i = 0; i < V[MGX_OV_GV_COUNT]; i++) {
      const int32_t* ins = P + mgx_sec_off(P, MGX_SEC_GV_CODE) + (V[MGX_OV_GV_START] + i) * MGX_GV_WORDS;
      if (ins[MGX_GV_OP] == MGX_GOP_STAT && ins[MGX_GV_A0] != 1) add_id(ins[MGX_GV_A1]);
    }
  };
  auto filters = [&](int pc0) {
    const int n = mgx_sec_cnt(P, MGX_SEC_ATOMS);
    std::vector<char> seen(n, 0);
    std::vector<int> todo{pc0};
    while (!todo.empty()) {
      const int pc = todo.back();
      todo.pop_back();
      if (pc < 0 || pc >= n || seen[pc]) continue;
      seen[pc] = 1;
      const int32_t* a = P + mgx_sec_off(P, MGX_SEC_ATOMS) + pc * MGX_AT_WORDS;
      if (a[MGX_AT_OP] == MGX_FOP_GAME_VALUE) { code_reads(a[MGX_AT_A1]); code_reads(a[MGX_AT_A2]); }
      todo.push_back(a[MGX_AT_ON_TRUE]);
      todo.push_back(a[MGX_AT_ON_FALSE]);
    }
  };
  auto mutations = [&](int start, int count) {
    for (int i = 0; i < count; i++) {
      const int32_t* m = P + mgx_sec_off(P, MGX_SEC_MUTS) + (start + i) * MGX_MU_WORDS;
      switch (m[MGX_MU_OP]) {
        case MGX_MOP_RESOURCE_DELTA: if (m[MGX_MU_A1] >= 0 && m[MGX_MU_A1] < 32) items |= 1u << m[MGX_MU_A1]; break;
        case MGX_MOP_CLEAR_INVENTORY:
          if (m[MGX_MU_A2] == 0) items |= (R >= 32 ? ~0u : (1u << R) - 1u);
          else for (int k = 0; k < m[MGX_MU_A2]; k++) items |= 1u << (P[mgx_sec_off(P, MGX_SEC_WORDLIST) + m[MGX_MU_A1] + k] & 31);
          break;
        case MGX_MOP_STATS: add_id(m[MGX_MU_A2]); code_reads(m[MGX_MU_A3]); break;
        case MGX_MOP_GAME_VALUE: {
          code_reads(m[MGX_MU_A2]);
          if (m[MGX_MU_A1] >= 0) {
            const int32_t* V = P + mgx_sec_off(P, MGX_SEC_OBS_VALUES) + m[MGX_MU_A1] * MGX_OV_WORDS;
            const int32_t* c0 = P + mgx_sec_off(P, MGX_SEC_GV_CODE) + V[MGX_OV_GV_START] * MGX_GV_WORDS;
            if (V[MGX_OV_GV_COUNT] > 0 && c0[MGX_GV_OP] == MGX_GOP_INVENTORY) items |= 1u << (c0[MGX_GV_A0] & 31);
            if (V[MGX_OV_GV_COUNT] > 0 && c0[MGX_GV_OP] == MGX_GOP_STAT && c0[MGX_GV_A0] != 1) add_id(c0[MGX_GV_A1]);
          }
          break;
        }
        default: break;
      }
    }
  };
  const int na = mgx_sec_cnt(P, MGX_SEC_AOES);
  for (int i = 0; i < na; i++) {
    const int32_t* a = P + mgx_sec_off(P, MGX_SEC_AOES) + i * MGX_AO_WORDS;
    filters(a[MGX_AO_FILTER_PC]);
    mutations(a[MGX_AO_MUT_START], a[MGX_AO_MUT_COUNT]);
    for (int k = 0; k < a[MGX_AO_PRES_COUNT]; k++)
      items |= 1u << (P[mgx_sec_off(P, MGX_SEC_PRESENCE) + (a[MGX_AO_PRES_START] + k) * MGX_PR_WORDS + MGX_PR_RESOURCE] & 31);
  }
  const int nt = mgx_sec_cnt(P, MGX_SEC_TERRITORIES);
  for (int t = 0; t < nt; t++) {
    const int32_t* TE = P + mgx_sec_off(P, MGX_SEC_TERRITORIES) + t * MGX_TE_WORDS;
    const int lists[3][2] = {{TE[MGX_TE_ENTER_START], TE[MGX_TE_ENTER_COUNT]}, {TE[MGX_TE_EXIT_START], TE[MGX_TE_EXIT_COUNT]},
                             {TE[MGX_TE_PRES_START], TE[MGX_TE_PRES_COUNT]}};
    for (const auto& l : lists)
      for (int i = 0; i < l[1]; i++) {
        const int32_t* hd = P + mgx_sec_off(P, MGX_SEC_HANDLERS) + (l[0] + i) * MGX_HD_WORDS;
        filters(hd[MGX_HD_FILTER_PC]);
        mutations(hd[MGX_HD_MUT_START], hd[MGX_HD_MUT_COUNT]);
      }
  }
  if (with_on_tick)
    for (int c = 0; c < P[MGX_H_NUM_CLASSES]; c++) {
      const int h = P[mgx_sec_off(P, MGX_SEC_CLASSES) + c * MGX_C_WORDS + MGX_C_ON_TICK];
      if (h < 0) continue;
      const int32_t* hd = P + mgx_sec_off(P, MGX_SEC_HANDLERS) + h * MGX_HD_WORDS;
      filters(hd[MGX_HD_FILTER_PC]);
      mutations(hd[MGX_HD_MUT_START], hd[MGX_HD_MUT_COUNT]);
    }
  // limit enforcement may drop other resources of a group when a modifier resource shrinks (inventory.cpp:141-173)
  for (int pass = 0; pass < 2; pass++) {
    const int nl = mgx_sec_cnt(P, MGX_SEC_LIMITS);
    for (int l = 0; l < nl; l++) {
      const int32_t* L = P + mgx_sec_off(P, MGX_SEC_LIMITS) + l * MGX_L_WORDS;
      bool hit = false;
      for (int k = 0; k < L[MGX_L_MOD_COUNT]; k++)
        if (items & (1u << (P[mgx_sec_off(P, MGX_SEC_MODS) + (L[MGX_L_MOD_START] + k) * MGX_MOD_WORDS + MGX_MOD_ITEM] & 31))) hit = true;
      if (hit)
        for (int k = 0; k < L[MGX_L_DROP_COUNT]; k++) items |= 1u << (P[mgx_sec_off(P, MGX_SEC_DROP_ORDER) + L[MGX_L_DROP_START] + k] & 31);
    }
  }
  std::vector<int> front;
  for (int item = 0; item < R && item < 32; item++)
    if (items & (1u << item)) {
      front.push_back(wk(MGX_S_RES_AMOUNT_BASE) + item);
      front.push_back(wk(MGX_S_RES_GAINED_BASE) + item);
      front.push_back(wk(MGX_S_RES_LOST_BASE) + item);
    }
  if (items) front.push_back(wk(MGX_S_DEATH));
  if (with_coverage) { front.push_back(wk(MGX_S_CELL_UNIQUE)); front.push_back(wk(MGX_S_CELL_MAXDIST)); }
  std::vector<int> all;
  for (int id : ids) all.push_back(id);          // named by SetStat / value code: few, and touched by every agent they apply to
  for (int id : front) if (id >= 0 && id < 256 && std::find(all.begin(), all.end(), id) == all.end()) all.push_back(id);
  out.clear();
  for (int id : all) if ((int)out.size() < MGX_AOE_MAX_STATS) out.push_back((int16_t)id);
}

#ifdef MGX_WORLD_TIMING  // instrumented developer build only (scripts/aoe_timing.py); not part of the ABI
extern "C" int mgx_debug_aoe_cycles(unsigned long long* out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(mgx_tu_aoe::mgx_dbg_cycles), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(mgx_tu_aoe::mgx_dbg_cycles), z, sizeof z); }
  return 0;
}
#endif
