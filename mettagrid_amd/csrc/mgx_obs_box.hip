// mgx_obs_box.hip — the dense-output instances of the observation kernel (fused form): mgx_obs_kernel<...,
// BOX = true> writes the policy's [C][H][W] box of every agent straight from its LDS staging row instead of the token rows
// (mgx_obs.h).  Own translation unit so that the token-path build does not wait for these instantiations.
#define MGX_WORLD_FAST_TU 1   // (no construction / restart kernels in this unit)
#define MGX_TU_NS mgx_tu_box
#include <hip/hip_runtime.h>

#include <mutex>

#include "mgx_device.h"
#include "mgx_obs.h"

template <bool X, bool PL, int NTH, int EW>
static void launch_box(hipStream_t stream, const MgxDev& dd, size_t lds, int pool_tokens, int pool_prefix, const uint8_t* mask, int blk_start,
                       int blk_words, int rewards_early, bool with_rewards, void* box, const float* scale, int C, int dtype, const int32_t* env_list,
                       const uint32_t* env_list_n, int list_grid, int stat_passes) {
  dim3 grid(env_list && !with_rewards ? list_grid : dd.E), block(NTH);
  if (with_rewards)
    hipLaunchKernelGGL((mgx_obs_kernel<true, X, PL, NTH, EW, MgxObsShapeDyn, true>), grid, block, lds, stream, dd, pool_tokens, pool_prefix, mask,
                       blk_start, blk_words, rewards_early, box, scale, C, dtype, env_list, env_list_n, stat_passes);
  else
    hipLaunchKernelGGL((mgx_obs_kernel<false, X, PL, NTH, EW, MgxObsShapeDyn, true>), grid, block, lds, stream, dd, pool_tokens, pool_prefix, mask,
                       blk_start, blk_words, rewards_early, box, scale, C, dtype, env_list, env_list_n, stat_passes);
}

// The variant the token path would launch for the same engine (mgx_engine.hip launch_obs): lean with / without the program
// block in LDS, extended with 256 threads or 512 threads and 4 / 3 / 2 encode wavefronts.
bool mgx_launch_obs_box(hipStream_t stream, const MgxDev& dd, size_t lds, int pool_tokens, int pool_prefix, const uint8_t* mask, int blk_start,
                        int blk_words, int rewards_early, bool with_rewards, bool X, bool PL, int threads, int ew, void* box, const float* scale,
                        int C, int dtype, const int32_t* env_list, const uint32_t* env_list_n, int list_grid, int stat_passes) {
#define MGX_BOX_ARGS stream, dd, lds, pool_tokens, pool_prefix, mask, blk_start, blk_words, rewards_early, with_rewards, box, scale, C, dtype, env_list, env_list_n, list_grid, stat_passes
  if (!X && PL) launch_box<false, true, MGX_OBS_THREADS, MGX_OBS_THREADS / MGX_WAVE>(MGX_BOX_ARGS);
  else if (!X) launch_box<false, false, MGX_OBS_THREADS, MGX_OBS_THREADS / MGX_WAVE>(MGX_BOX_ARGS);
  else if (threads == 512 && ew == 4) launch_box<true, false, 512, 4>(MGX_BOX_ARGS);
  else if (threads == 512 && ew == 3) launch_box<true, false, 512, 3>(MGX_BOX_ARGS);
  else if (threads == 512 && ew == 2) launch_box<true, false, 512, 2>(MGX_BOX_ARGS);
  else if (threads == MGX_OBS_THREADS) launch_box<true, false, MGX_OBS_THREADS, MGX_OBS_THREADS / MGX_WAVE>(MGX_BOX_ARGS);
  else return false;
#undef MGX_BOX_ARGS
  return true;
}

bool mgx_obs_box_set_lds(size_t lds) {   // per-kernel attribute, only ever raised
  static std::mutex mu;
  static size_t cur_max_dev[64] = {0};   // ... per device
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  size_t& cur_max = cur_max_dev[dev];
  if (lds <= cur_max) return true;
#define MGX_BOX_KERNELS(X, PL, NTH, EW) (const void*)mgx_obs_kernel<true, X, PL, NTH, EW, MgxObsShapeDyn, true>, (const void*)mgx_obs_kernel<false, X, PL, NTH, EW, MgxObsShapeDyn, true>
  const void* fns[] = {MGX_BOX_KERNELS(false, true, MGX_OBS_THREADS, MGX_OBS_THREADS / MGX_WAVE), MGX_BOX_KERNELS(false, false, MGX_OBS_THREADS, MGX_OBS_THREADS / MGX_WAVE),
                       MGX_BOX_KERNELS(true, false, 512, 4), MGX_BOX_KERNELS(true, false, 512, 3), MGX_BOX_KERNELS(true, false, 512, 2),
                       MGX_BOX_KERNELS(true, false, MGX_OBS_THREADS, MGX_OBS_THREADS / MGX_WAVE)};
#undef MGX_BOX_KERNELS
  for (const void* f : fns)
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
  cur_max = lds;
  return true;
}
