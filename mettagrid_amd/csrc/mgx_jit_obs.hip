// mgx_jit_obs.hip — run-time specialisation unit, NOT part of libmgx.so: the observation kernel instantiated for the shape of
// ONE program (map, agents, slots, token budget, window, value base, interpreted block, flags as compile-time constants —
// what gen_presets.py does for the benchmark presets at build()).  mettagrid_amd/jit.py compiles it with
//   hipcc --offload-arch=gfx950 --genco -DMGX_JIT_SHAPE=H,W,A,S,T,NOFF,BASE,NOV,NT,NRW,FLAGS,MAX_STEPS,BLKW,MASK_FEAT,REWARDS_EARLY
// and mgx_attach_code loads the code object.  Lean games with the interpreted program block in LDS only (PL).
#define MGX_WORLD_FAST_TU 1
#define MGX_JIT_UNIT 1
#define MGX_TU_NS mgx_tu_jit_obs
#include <hip/hip_runtime.h>

#include "mgx_device.h"
#include "mgx_obs.h"

#ifndef MGX_JIT_SHAPE
#error "MGX_JIT_SHAPE: the fifteen fields of MgxObsShape"
#endif
typedef MgxObsShape<MGX_JIT_SHAPE> MgxObsShapeJ;
extern "C" __constant__ long long mgx_jit_obs_info[24] = {0x4D47584A49544F31ll /* "MGXJITO1" */, sizeof(MgxDev), MGX_OBS_THREADS, MGX_OBS_THREADS / MGX_WAVE, MGX_VERSION,
                                                          MGX_JIT_SHAPE, 0, 0, 0, 0};

#define MGX_JIT_OBS_ARGS                                                                                                                    \
  MgxDev d, int pool_tokens, int pool_prefix, const uint8_t *env_mask, int blk_start, int blk_words_arg, int rewards_early_arg, void *box_out, \
      const float *box_scale, int box_C, int box_dtype, const int32_t *env_list, const uint32_t *env_list_n, int stat_passes
extern "C" __global__ void __launch_bounds__(MGX_OBS_THREADS) mgx_jit_obs_r(MGX_JIT_OBS_ARGS) {   // the step: observations + rewards
  mgx_obs_run<true, false, true, MGX_OBS_THREADS, MGX_OBS_THREADS / MGX_WAVE, MgxObsShapeJ, false>(
      d, pool_tokens, pool_prefix, env_mask, blk_start, blk_words_arg, rewards_early_arg, box_out, box_scale, box_C, box_dtype, env_list, env_list_n,
      stat_passes);
}
extern "C" __global__ void __launch_bounds__(MGX_OBS_THREADS) mgx_jit_obs_n(MGX_JIT_OBS_ARGS) {   // initial observations (construction, restarts)
  mgx_obs_run<false, false, true, MGX_OBS_THREADS, MGX_OBS_THREADS / MGX_WAVE, MgxObsShapeJ, false>(
      d, pool_tokens, pool_prefix, env_mask, blk_start, blk_words_arg, rewards_early_arg, box_out, box_scale, box_C, box_dtype, env_list, env_list_n,
      stat_passes);
}
