// mgx_jit_world.hip — run-time specialisation unit, NOT part of libmgx.so: the lean world-update kernel with straight-line
// handler code generated for ONE program (mettagrid_amd/jit.py compiles this file with
//   hipcc --offload-arch=gfx950 --genco -DMGX_GEN_HEADER='"<generated header>"' -DMGX_JIT_FP=0x...ull
// into a code object that mgx_attach_code loads with hipModuleLoad).  Same switches as mgx_world_fast.hip: MgxDev in
// constant memory (here an external symbol the host writes through hipModuleGetGlobal), the register handler VM, one flat
// kernel.  The generated header defines `template <class Env> struct MgxGenJ` (mettagrid_amd/gen_handlers.py render_one).
#define MGX_BIG __forceinline__
#define MGX_OUTLINE __forceinline__
#define MGX_WORLD_FAST_TU 1
#define MGX_JIT_UNIT 1
#define MGX_TU_NS mgx_tu_jit
#define MGX_CONST_DEV 1
#define MGX_WORLD_IDS 1
#ifndef MGX_NO_WORLD_HELPERS
#define MGX_WORLD_HELPERS 1   // idle upper half-wavefronts take half of the batched per-agent passes (mgx_world.h)
#endif
#define MGX_GEN_HANDLERS MgxGenJ
#define MGX_GEN_ID MGX_JIT_GEN_ID
#define MGX_JIT_GEN_ID 9   // MgxDev::gen_prog of an engine running this code object
#ifndef MGX_WORLD_LPW
#define MGX_WORLD_LPW 32
#endif
#ifndef MGX_WORLD_WPE
#define MGX_WORLD_WPE 2
#endif
#include <hip/hip_runtime.h>

#include "mgx_device.h"
#include "mgx_world.h"

#ifndef MGX_JIT_FP
#error "MGX_JIT_FP: fingerprint of the handler tables the generated header was made from"
#endif
// What the host checks before it trusts the code object (mgx_attach_code): layout of MgxDev, launch geometry, program.
extern "C" __constant__ unsigned long long mgx_jit_info[8] = {0x4D47584A49545731ull /* "MGXJITW1" */, sizeof(MgxDev), MGX_WORLD_EPG, MGX_WORLD_THREADS,
                                                             MGX_JIT_FP, MGX_JIT_GEN_ID, MGX_VERSION, 0};

// MGX_JIT_ONLY_PL = 1 / 0: build only the variant the engine will launch (each takes over a minute to compile); unset: both.
#if !defined(MGX_JIT_ONLY_PL) || MGX_JIT_ONLY_PL == 1
extern "C" __global__ void __launch_bounds__(MGX_WORLD_THREADS) __attribute__((amdgpu_waves_per_eu(MGX_WORLD_WPE, MGX_WORLD_WPE)))
mgx_jit_world_pl(int prog_words) {   // program copied into LDS
  mgx_tu_jit::mgx_world_entry<true, false>(g_mgx_dev, prog_words);
}
#endif
#if !defined(MGX_JIT_ONLY_PL) || MGX_JIT_ONLY_PL == 0
extern "C" __global__ void __launch_bounds__(MGX_WORLD_THREADS) __attribute__((amdgpu_waves_per_eu(MGX_WORLD_WPE, MGX_WORLD_WPE)))
mgx_jit_world_pg(int prog_words) {   // program read from global memory
  mgx_tu_jit::mgx_world_entry<false, false>(g_mgx_dev, prog_words);
}
#endif
