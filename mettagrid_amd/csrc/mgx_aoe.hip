// mgx_aoe.hip — area effects with one lane per AGENT: fixed AoE, territory effects and mobile AoE of _step
// (/root/reference/cpp/bindings/mettagrid_c.cpp:1032-1042) for games whose AoE / territory handlers only touch their
// target (MgxEnvT::aoe_local_agent; the host proves that from the program, mgx_aoe_is_target_local below).
// One wavefront per env, lane = agent index: 65 536 envs x 64 agents are 65 536 independent wavefronts instead of 1 024
// wavefronts each walking 64 agents x (16 fixed + 8 territory + 64 mobile sources) serially.  One flat kernel: the
// filters / mutations these records may use need neither the handler VM nor queries.
#define MGX_BIG __forceinline__
#define MGX_OUTLINE __forceinline__
#define MGX_WORLD_FAST_TU 1
#define MGX_TU_NS mgx_tu_aoe
#include <hip/hip_runtime.h>

#include <vector>

#include "mgx_device.h"
#include "mgx_world.h"

#define MGX_AOE_THREADS 256
#ifdef MGX_CPU_EMU
#define MGX_AOE_OCCUPANCY
#else
#define MGX_AOE_OCCUPANCY __attribute__((amdgpu_waves_per_eu(4, 4)))  // 128 VGPRs: the kernel waits on memory 80 % of the time
#endif
__global__ void __launch_bounds__(MGX_AOE_THREADS) MGX_AOE_OCCUPANCY mgx_aoe_kernel(const MgxDev* __restrict__ dp) {
  const MgxDev& d = *dp;  // per-engine copy in device memory (see mgx_world_x.hip)
  extern __shared__ __align__(16) uint8_t aoe_lds[];
  const int wave = (int)threadIdx.x / MGX_WAVE, lane = (int)threadIdx.x & (MGX_WAVE - 1);
  const int env = (int)blockIdx.x * (MGX_AOE_THREADS / MGX_WAVE) + wave;
  if (env >= d.E) return;
  MgxEnvT<MgxGlobalProg, true> e(d, d.P, env);
  e.step = d.step[env];
  e.xl.def_delta = (int*)aoe_lds;  // deferred target deltas of apply_fixed: [28][threads]
  e.xl.lane = (int)threadIdx.x;
  e.xl.stride = MGX_AOE_THREADS;
  for (int a = lane; a < d.A; a += MGX_WAVE) e.aoe_local_agent(a);
}

// One thread per (env, registered AoE source): the packed (location, radius, live) record aoe_local_agent scans.
__global__ void __launch_bounds__(256) mgx_aoe_prep_kernel(const MgxDev* __restrict__ dp) {
  const MgxDev& d = *dp;
  const int per = d.NF + d.NM;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)d.E * per) return;
  MgxEnvT<MgxGlobalProg, true> e(d, d.P, (int)(idx / per));
  e.aoe_pack_source((int)(idx % per));
}

void mgx_launch_aoe(hipStream_t stream, const MgxDev& d, const MgxDev* dp) {
  const long long sources = (long long)d.E * (d.NF + d.NM);
  if (sources > 0) hipLaunchKernelGGL(mgx_aoe_prep_kernel, dim3((unsigned)((sources + 255) / 256)), dim3(256), 0, stream, dp);
  const int epb = MGX_AOE_THREADS / MGX_WAVE;
  hipLaunchKernelGGL(mgx_aoe_kernel, dim3((d.E + epb - 1) / epb), dim3(MGX_AOE_THREADS), 28 * MGX_AOE_THREADS * 4, stream, dp);
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
