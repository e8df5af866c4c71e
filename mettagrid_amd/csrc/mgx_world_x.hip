// mgx_world_x.hip — the extended world-update kernel (rung 4: dynamic tags, tag index, queries, events, AoE,
// territory, run-time object creation).  One env per lane like the lean kernel, but the handler VM keeps its frames and
// handler contexts in LDS (MgxEnvT::vm_run) so that tag lifecycle handlers, materialized-query recomputation and
// UseTarget nest without recursion.  These kernels take the engine's MgxDev through a POINTER to a per-engine copy in
// device memory, not by value: with out-of-line functions holding a reference to it, a by-value argument is copied into
// every lane's private memory (992 B of scratch) and each field access becomes a scratch load followed by a flat access.
// A few large functions stay real calls (MGX_OUTLINE: vm_run, check_filters<Q>,
// eval_code<Q>, eval_query<Q> — one instance each, bounded static call depth); everything else is inlined.
#define MGX_BIG __forceinline__
#define MGX_OUTLINE __noinline__
#define MGX_WORLD_FAST_TU 1
#define MGX_TU_NS mgx_tu_x
#define MGX_WORLD_IDS 1
#define MGX_HOT_PROG 1      // the LDS program copy is the hot range only; class records from HBM / L2 (mgx_world.h)
#define MGX_NO_CLS_STAGE 1  // 14 instead of 16 bytes of LDS per agent and env
// One 32-env wavefront per workgroup: with many agents per env the LDS staging (17 B per agent and env + 320 B per env)
// is what limits how many envs a CU holds; small workgroups pack it better (rung 4: 45 KB per workgroup).
#ifndef MGX_WORLD_LPW
#define MGX_WORLD_LPW 32
#endif
#ifndef MGX_WORLD_EPG
#define MGX_WORLD_EPG 32
#endif
#include <hip/hip_runtime.h>

#include <mutex>

#include "mgx_device.h"
#include "mgx_world.h"

template <bool PROG_LDS>
__global__ void __launch_bounds__(MGX_WORLD_THREADS) mgx_world_kernel_x(const MgxDev* __restrict__ dp, int prog_words, int phases) {
  mgx_world_entry<PROG_LDS, true>(*dp, prog_words, phases);
}

// Game values with query operands (QueryInventoryValue / QueryCountValue) outside the world update: the global
// observation values (phase 0, before the observation kernel) and the reward entries + truncation / termination
// (phase 1, after it — RewardHelper::compute_entries, systems/reward.hpp:56-77, mettagrid_c.cpp:1070-1096).  Queries
// share one workspace per env, so an env is walked by one lane, agents in index order.  Games without such values
// never launch this kernel: the observation kernel evaluates plain values itself, one agent per lane.
__global__ void __launch_bounds__(MGX_WORLD_THREADS) mgx_values_kernel(const MgxDev* __restrict__ dp, int phase, const uint8_t* env_mask) {
  const MgxDev& d = *dp;
  const int lane = mgx_world_lane();
  const bool active = (threadIdx.x & (MGX_WAVE - 1)) < MGX_WORLD_LPW;
  const int env = blockIdx.x * MGX_WORLD_EPG + lane;
  if (!active || env >= d.E) return;
  if (env_mask && !env_mask[env]) return;
  MgxEnvT<MgxGlobalProg, true> e(d, d.P, env);
  e.step = d.step[env];
  const uint32_t step = e.step;
  for (int a = 0; a < d.A; a++) {
    const int slot = d.ag_obj[e.ao(a)];
    MgxCtx vc = mgx_ctx(slot, slot);
    if (phase == 0) {
      for (int i = 0; i < d.n_obs_values; i++) {  // _emit_obs_value_tokens mettagrid_c.cpp:1207-1238 (value -> u32)
        const int32_t* V = d.P + d.sec[MGX_SEC_OBS_VALUES] + i * MGX_OV_WORDS;
        d.obsval[((size_t)env * d.A + a) * d.n_obs_values + i] =
            (uint32_t)e.eval_code<3>(V[MGX_OV_GV_START], V[MGX_OV_GV_COUNT], slot, vc, 0);
      }
    } else {
      const int32_t* C = mgx_cls(d, d.obj_cls[e.so(slot)]);
      const int32_t* rw = d.P + d.sec[MGX_SEC_REWARDS] + C[MGX_C_REWARD_START] * MGX_RW_WORDS;
      const float ep = d.episode_rewards[e.ao(a)];
      float total = 0.f;
      for (int k = 0; k < C[MGX_C_REWARD_COUNT]; k++, rw += MGX_RW_WORDS) {
        float* prev = &d.ag_rprev[e.ao(a) * d.NRW + k];
        const float pv = *prev;
        const float val = e.eval_code<3>(rw[MGX_RW_GV_START], rw[MGX_RW_GV_COUNT], slot, vc, 0);
        total = rw[MGX_RW_ACCUMULATE] ? __fadd_rn(total, val) : __fadd_rn(total, __fsub_rn(val, pv));
        *prev = val;
      }
      const float reward = total != 0.f ? total : 0.f;  // rewards were zeroed at the top of the step; += total
      d.rewards[e.ao(a)] = reward;
      d.episode_rewards[e.ao(a)] = __fadd_rn(ep, reward);
      if (d.max_steps > 0 && step >= (uint32_t)d.max_steps) {
        if (d.truncates) d.truncations[e.ao(a)] = 1;
        else d.terminals[e.ao(a)] = 1;
      }
    }
  }
}
void mgx_launch_values(hipStream_t stream, const MgxDev& d, const MgxDev* dp, int phase, const uint8_t* env_mask) {
  dim3 grid((d.E + MGX_WORLD_EPG - 1) / MGX_WORLD_EPG), block(MGX_WORLD_THREADS);
  hipLaunchKernelGGL(mgx_values_kernel, grid, block, 0, stream, dp, phase, env_mask);
}

static std::mutex g_lds_mutex;
static size_t g_lds_max_dev[64] = {0};   // the attribute is per kernel AND per device: one maximum for each
bool mgx_world_x_set_lds(size_t lds) {  // process-wide maximum, only ever raised (the attribute is per kernel)
  std::lock_guard<std::mutex> lock(g_lds_mutex);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  size_t& g_lds_max = g_lds_max_dev[dev];
  if (lds <= g_lds_max) return true;
  if (hipFuncSetAttribute((const void*)mgx_world_kernel_x<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
      hipFuncSetAttribute((const void*)mgx_world_kernel_x<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return false;
  g_lds_max = lds;
  return true;
}

size_t mgx_world_x_lds_bytes(int A, bool aoe_lds) { return (size_t)mgx_world_lds_fixed(A, true, aoe_lds); }

size_t mgx_world_x_private_bytes() {  // per-lane private segment of the build (reported by MGX_VERBOSE, checked by tests)
  size_t m = 0;
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, (const void*)mgx_world_kernel_x<true>) == hipSuccess) m = fa.localSizeBytes;
  if (hipFuncGetAttributes(&fa, (const void*)mgx_world_kernel_x<false>) == hipSuccess && fa.localSizeBytes > m) m = fa.localSizeBytes;
  return m;
}

void mgx_launch_world_x(bool prog_lds, size_t lds, hipStream_t stream, const MgxDev& d, const MgxDev* dp, int prog_words, int phases) {
  dim3 grid((d.E + MGX_WORLD_EPG - 1) / MGX_WORLD_EPG), block(MGX_WORLD_THREADS);
  if (prog_lds) hipLaunchKernelGGL((mgx_world_kernel_x<true>), grid, block, lds, stream, dp, prog_words, phases);
  else hipLaunchKernelGGL((mgx_world_kernel_x<false>), grid, block, lds, stream, dp, prog_words, phases);
}

#ifdef MGX_WORLD_TIMING  // instrumented developer build only (scripts/world_timing_x.py); not part of the ABI
extern "C" int mgx_debug_world_x_cycles(unsigned long long* out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(mgx_tu_x::mgx_dbg_cycles), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(mgx_tu_x::mgx_dbg_cycles), z, sizeof z); }
  return 0;
}
#endif
