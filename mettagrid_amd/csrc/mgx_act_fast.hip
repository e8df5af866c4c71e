// mgx_act_fast.hip — the lean world update with one lane per AGENT (mgx_act.h): staging, shuffle, conflict-ordered
// action rounds, vibe stream, per-agent on_tick, bookkeeping and coverage tracking in one kernel.  Own translation unit
// per constant-memory slot, like mgx_world_fast.hip (same switches: constant-memory MgxDev, id-derived accessors, the
// register handler VM, everything inlined).
#define MGX_BIG __forceinline__
#define MGX_OUTLINE __forceinline__
#define MGX_WORLD_FAST_TU 1
#define MGX_ACT_TU 1
#ifndef MGX_SLOT
#define MGX_SLOT 0
#endif
#define MGX_CAT2(a, b) a##b
#define MGX_CAT(a, b) MGX_CAT2(a, b)
#define MGX_TU_NS MGX_CAT(mgx_tu_actf, MGX_SLOT)
#define MGX_CONST_DEV 1
#define MGX_WORLD_IDS 1
#ifndef MGX_NO_GEN_HANDLERS
#define MGX_GEN_HANDLERS MgxGenR3   // straight-line handler code of the preset (mgx_handlers_gen.h), used when MgxDev::gen_prog says so
#define MGX_GEN_ID 3
#endif
// 16 envs per workgroup: 256 lanes at 16 agents per env (four envs per wavefront); games with more agents stay lane per env.
#define MGX_WORLD_EPG 16
#define MGX_WORLD_LPW 64
#ifndef MGX_ACT_WPE
#define MGX_ACT_WPE 4
#endif
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "mgx_device.h"
#include "mgx_act.h"

#ifdef MGX_CPU_EMU
#define MGX_WPE_ATTR
#else
#define MGX_WPE_ATTR __attribute__((amdgpu_waves_per_eu(MGX_ACT_WPE, MGX_ACT_WPE)))
#endif
namespace MGX_TU_NS {
template <bool PROG_LDS>
__global__ void __launch_bounds__(256) MGX_WPE_ATTR mgx_act_kernel_fast(int prog_words) {
  mgx_act_entry<PROG_LDS, false>(g_mgx_dev, prog_words);
}
}  // namespace

static std::mutex g_lds_mutex;
static size_t g_lds_max_dev[64] = {0};   // the attribute is per kernel AND per device: one maximum for each
bool MGX_CAT(mgx_act_fast_set_lds_s, MGX_SLOT)(size_t lds) {
  std::lock_guard<std::mutex> lock(g_lds_mutex);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  size_t& g_lds_max = g_lds_max_dev[dev];
  if (lds <= g_lds_max) return true;
  if (hipFuncSetAttribute((const void*)mgx_act_kernel_fast<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
      hipFuncSetAttribute((const void*)mgx_act_kernel_fast<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return false;
  g_lds_max = lds;
  return true;
}

#if MGX_SLOT == 0
size_t mgx_act_fast_lds_bytes(int A, int extra) { return (size_t)mgx_world_lds_fixed(A, false, true, extra); }
int mgx_act_fast_epg() { return MGX_WORLD_EPG; }
#endif

static std::mutex g_dev_mutex;
static MgxDev g_dev_host_dev[16];     // the __constant__ symbol exists once per DEVICE: one host image for each
static bool g_dev_valid_dev[16] = {false};
static bool g_thrash_warned = false;
void MGX_CAT(mgx_launch_act_fast_s, MGX_SLOT)(bool prog_lds, size_t lds, hipStream_t stream, const MgxDev& d, int prog_words) {
  std::lock_guard<std::mutex> lock(g_dev_mutex);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  MgxDev& g_dev_host = g_dev_host_dev[dev];
  bool& g_dev_valid = g_dev_valid_dev[dev];
  if (!g_dev_valid || memcmp(&g_dev_host, &d, sizeof(MgxDev)) != 0) {
    if (g_dev_valid) {
      (void)hipDeviceSynchronize();
      if (!g_thrash_warned && getenv("MGX_VERBOSE")) {   // two engines alternating on one slot pay a device-wide wait per step
        fprintf(stderr, "[mgx] %s: the constant-memory engine table of device %d changed (another engine on the same slot, or re-bound "
                        "buffers): device synchronised before the upload\n", __func__, dev);
        g_thrash_warned = true;
      }
    }
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_mgx_dev), &d, sizeof(MgxDev), 0, hipMemcpyHostToDevice, stream);
    memcpy(&g_dev_host, &d, sizeof(MgxDev));
    g_dev_valid = true;
  }
  int ap = 1;
  while (ap < d.A) ap <<= 1;
  dim3 grid((d.E + MGX_WORLD_EPG - 1) / MGX_WORLD_EPG), block(MGX_WORLD_EPG * ap);
  if (prog_lds) hipLaunchKernelGGL((mgx_act_kernel_fast<true>), grid, block, lds, stream, prog_words);
  else hipLaunchKernelGGL((mgx_act_kernel_fast<false>), grid, block, lds, stream, prog_words);
}

#if defined(MGX_WORLD_TIMING) && MGX_SLOT == 0  // instrumented developer build only (scripts/act_timing.py); not part of the ABI
extern "C" int mgx_debug_act_fast_cycles(unsigned long long* out, int reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(MGX_TU_NS::mgx_dbg_cycles), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(MGX_TU_NS::mgx_dbg_cycles), z, sizeof z); }
  return 0;
}
#endif
