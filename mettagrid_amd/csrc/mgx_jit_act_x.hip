// mgx_jit_act_x.hip — run-time specialisation unit, NOT part of libmgx.so: the extended games' action dispatch with one lane
// per agent (mgx_act_x.hip / mgx_act.h) carrying straight-line handler code generated for ONE program (mettagrid_amd/jit.py:
//   hipcc --offload-arch=gfx950 --genco -DMGX_GEN_HEADER='"<generated header>"' -DMGX_JIT_FP=0x...ull
// -> a code object that mgx_attach_code(MGX_CODE_ACT_X) loads).  Same switches as mgx_act_x.hip.
#define MGX_BIG __forceinline__
#define MGX_OUTLINE __forceinline__
#define MGX_WORLD_FAST_TU 1
#define MGX_ACT_TU 1
#define MGX_JIT_UNIT 1
#define MGX_TU_NS mgx_tu_jit_actx
#define MGX_WORLD_IDS 1
#define MGX_GEN_HANDLERS MgxGenJ
#define MGX_GEN_ID 9
#define MGX_HOT_PROG 1
#define MGX_NO_CLS_STAGE 1
#define MGX_WORLD_EPG 4
#define MGX_WORLD_LPW 64
#ifndef MGX_ACT_WPE
#define MGX_ACT_WPE 3
#endif
#include <hip/hip_runtime.h>

#include "mgx_device.h"
#include "mgx_act.h"

#ifndef MGX_JIT_FP
#error "MGX_JIT_FP: fingerprint of the handler tables the generated header was made from"
#endif
extern "C" __constant__ unsigned long long mgx_jit_info[8] = {0x4D47584A49544131ull /* "MGXJITA1" */, sizeof(MgxDev), MGX_WORLD_EPG, 0,
                                                             MGX_JIT_FP, 9, MGX_VERSION, 0};

extern "C" __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MGX_ACT_WPE, MGX_ACT_WPE)))
mgx_jit_act_x(const MgxDev* __restrict__ dp, int prog_words) {   // (hot program range in LDS, like the build's only variant)
  mgx_tu_jit_actx::mgx_act_entry<true, true>(*dp, prog_words);
}
