// mgx_act_x.hip — the extended variant's action dispatch with one lane per AGENT (mgx_act.h): staging, shuffle,
// conflict-ordered action rounds, vibe stream and bookkeeping; timestep events, area effects and the tail stay with
// mgx_world_kernel_x / mgx_aoe_kernel.  Same switches as mgx_world_x.hip (MgxDev through a pointer, hot program range in
// LDS); only games whose action-phase handlers run on the register VM come here (MgxDev::flat_top), so nothing is
// out of line.
#define MGX_BIG __forceinline__
#ifndef MGX_OUTLINE
#define MGX_OUTLINE __forceinline__
#endif
#define MGX_WORLD_FAST_TU 1
#define MGX_ACT_TU 1
#define MGX_TU_NS mgx_tu_actx
#define MGX_WORLD_IDS 1
#ifndef MGX_NO_GEN_HANDLERS
#define MGX_GEN_HANDLERS MgxGenR4   // straight-line handler code of the preset (mgx_handlers_gen.h), used when MgxDev::gen_prog says so
#define MGX_GEN_ID 4
#endif
#define MGX_HOT_PROG 1
#define MGX_NO_CLS_STAGE 1
// 4 envs per workgroup: 256 lanes at 64 agents per env (one env per wavefront)
#define MGX_WORLD_EPG 4
#define MGX_WORLD_LPW 64
// measured (rung 4): 3 wavefronts per SIMD (168 VGPRs, 10 spilled) 2.0 ms, 4 (128, 83 spilled) 2.3 ms
#ifndef MGX_ACT_WPE
#define MGX_ACT_WPE 3
#endif
#include <hip/hip_runtime.h>

#include <mutex>

#include "mgx_device.h"
#include "mgx_act.h"

#ifdef MGX_CPU_EMU
#define MGX_WPE_ATTR
#else
#define MGX_WPE_ATTR __attribute__((amdgpu_waves_per_eu(MGX_ACT_WPE, MGX_ACT_WPE)))
#endif
template <bool PROG_LDS>
__global__ void __launch_bounds__(256) MGX_WPE_ATTR mgx_act_kernel_x(const MgxDev* __restrict__ dp, int prog_words) {
  mgx_act_entry<PROG_LDS, true>(*dp, prog_words);
}

static std::mutex g_lds_mutex;
static size_t g_lds_max_dev[64] = {0};   // the attribute is per kernel AND per device: one maximum for each
bool mgx_act_x_set_lds(size_t lds) {
  std::lock_guard<std::mutex> lock(g_lds_mutex);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  size_t& g_lds_max = g_lds_max_dev[dev];
  if (lds <= g_lds_max) return true;
  if (hipFuncSetAttribute((const void*)mgx_act_kernel_x<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return false;
  g_lds_max = lds;
  return true;
}
size_t mgx_act_x_lds_bytes(int A, bool aoe_lds, int extra) { return (size_t)mgx_world_lds_fixed(A, true, aoe_lds, extra); }
int mgx_act_x_epg() { return MGX_WORLD_EPG; }

void mgx_launch_act_x(bool prog_lds, size_t lds, hipStream_t stream, const MgxDev& d, const MgxDev* dp, int prog_words) {
  int ap = 1;
  while (ap < d.A) ap <<= 1;
  dim3 grid((d.E + MGX_WORLD_EPG - 1) / MGX_WORLD_EPG), block(MGX_WORLD_EPG * ap);
  // (only the variant with the hot program range in LDS is built — the fully inlined kernel takes minutes to compile;
  // programs whose hot range does not fit keep the lane-per-env kernel: mgx_create)
  (void)prog_lds;
  hipLaunchKernelGGL((mgx_act_kernel_x<true>), grid, block, lds, stream, dp, prog_words);
}

#ifdef MGX_ACT_DEBUG
extern "C" int mgx_debug_act_env(int env) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(mgx_tu_actx::mgx_act_dbg_env), &env, sizeof(int)); }
extern "C" int mgx_debug_act_read(uint32_t* out) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mgx_tu_actx::mgx_act_dbg), sizeof(uint32_t) * 192);
}
#endif

#ifdef MGX_WORLD_TIMING  // instrumented developer build only (scripts/act_timing.py); not part of the ABI
extern "C" int mgx_debug_act_x_cycles(unsigned long long* out, int reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(mgx_tu_actx::mgx_dbg_cycles), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(mgx_tu_actx::mgx_dbg_cycles), z, sizeof z); }
  return 0;
}
#endif
