// mgx_world_fast.hip — the lean world-update kernels (games without rung-4 features: no dynamic tags, queries,
// events, AoE, territory, run-time object creation).  Own translation unit: MgxDev in constant memory (one copy per
// MGX_SLOT), the register handler VM, and MGX_BIG=__forceinline__ — one flat kernel, no calls but mgx_logf.
#define MGX_BIG __forceinline__  // safe here: the lean variant's handler VM is iterative (MgxEnvT::run_handler)
#define MGX_OUTLINE __forceinline__
#define MGX_WORLD_FAST_TU 1
#ifndef MGX_SLOT
#define MGX_SLOT 0  // which constant-memory copy of MgxDev this object file owns (mgx_device.h MGX_CONST_DEV)
#endif
#define MGX_CAT2(a, b) a##b
#define MGX_CAT(a, b) MGX_CAT2(a, b)
#define MGX_TU_NS MGX_CAT(mgx_tu_fast, MGX_SLOT)
#define MGX_CONST_DEV 1
#define MGX_WORLD_IDS 1
#ifndef MGX_NO_WORLD_HELPERS
#define MGX_WORLD_HELPERS 1   // idle upper half-wavefronts take half of the batched per-agent passes (mgx_world.h)
#endif
#ifndef MGX_NO_GEN_HANDLERS
#define MGX_GEN_HANDLERS MgxGenR3   // straight-line handler code of the preset (mgx_handlers_gen.h), used when MgxDev::gen_prog says so
#define MGX_GEN_ID 3
#endif
// Measured on MI355X (rung 3, 65 536 envs): 32 envs per wavefront, two wavefronts per SIMD is the best split —
// 64 lanes serialise more distinct handler paths per wavefront, 16 or 8 multiply the instruction stream.
#ifndef MGX_WORLD_LPW
#define MGX_WORLD_LPW 32
#endif
#ifndef MGX_WORLD_WPE
#define MGX_WORLD_WPE 2
#endif
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "mgx_device.h"
#include "mgx_world.h"

#ifdef MGX_WORLD_WPE
#define MGX_WPE_ATTR __attribute__((amdgpu_waves_per_eu(MGX_WORLD_WPE, MGX_WORLD_WPE)))
#else
#define MGX_WPE_ATTR
#endif
#ifdef MGX_CPU_EMU
#undef MGX_WPE_ATTR
#define MGX_WPE_ATTR
#endif
namespace MGX_TU_NS {
template <bool PROG_LDS>
__global__ void __launch_bounds__(MGX_WORLD_THREADS) MGX_WPE_ATTR mgx_world_kernel_fast(int prog_words) {
  mgx_world_entry<PROG_LDS, false>(g_mgx_dev, prog_words);
}
}  // namespace

// One process-wide maximum per kernel: the attribute is per kernel, not per engine, and must never be lowered under a
// live engine that needs more.
static std::mutex g_lds_mutex;
static size_t g_lds_max_dev[64] = {0};   // the attribute is per kernel AND per device: one maximum for each
bool MGX_CAT(mgx_world_fast_set_lds_s, MGX_SLOT)(size_t lds) {
  std::lock_guard<std::mutex> lock(g_lds_mutex);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  size_t& g_lds_max = g_lds_max_dev[dev];
  if (lds <= g_lds_max) return true;
  if (hipFuncSetAttribute((const void*)mgx_world_kernel_fast<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
      hipFuncSetAttribute((const void*)mgx_world_kernel_fast<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return false;
  g_lds_max = lds;
  return true;
}

#if MGX_SLOT == 0
size_t mgx_world_fast_lds_bytes(int A) { return (size_t)mgx_world_lds_fixed(A, false); }
#endif

// Keeps this slot's g_mgx_dev equal to the launching engine's MgxDev.  A change of content first waits for the kernels
// that still read the old content (another engine on the same slot, or re-bound buffers).
static std::mutex g_dev_mutex;
static MgxDev g_dev_host_dev[16];     // the __constant__ symbol exists once per DEVICE: one host image for each
static bool g_dev_valid_dev[16] = {false};
static bool g_thrash_warned = false;
void MGX_CAT(mgx_launch_world_fast_s, MGX_SLOT)(bool prog_lds, size_t lds, hipStream_t stream, const MgxDev& d, int prog_words) {
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
  dim3 grid((d.E + MGX_WORLD_EPG - 1) / MGX_WORLD_EPG), block(MGX_WORLD_THREADS);
  if (prog_lds) hipLaunchKernelGGL((mgx_world_kernel_fast<true>), grid, block, lds, stream, prog_words);
  else hipLaunchKernelGGL((mgx_world_kernel_fast<false>), grid, block, lds, stream, prog_words);
}

#if defined(MGX_WORLD_TIMING) && MGX_SLOT == 0  // instrumented developer build only (scripts/world_timing.py); not part of the ABI
extern "C" int mgx_debug_world_cycles(unsigned long long* out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(mgx_dbg_cycles), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(mgx_dbg_cycles), z, sizeof z); }
  return 0;
}
#endif
