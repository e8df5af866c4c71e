// TEST INFRASTRUCTURE ONLY — storage for the names the HIP runtime provides on a GPU (see hip/hip_runtime.h here).
#include <hip/hip_runtime.h>
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
alignas(16) uint8_t mgx_dyn_lds[160 * 1024];  // dynamic LDS of the world kernels
alignas(16) uint8_t smem[160 * 1024];         // dynamic LDS of the (not emulated) observation kernel
alignas(16) uint8_t aoe_lds[160 * 1024];     // dynamic LDS of mgx_aoe_kernel
// the token-decode kernel is wavefront-cooperative (ballot / shuffles): not part of the sanitizer build
int mgx_launch_decode(hipStream_t, const uint8_t*, float*, const float*, long long, int, int, int, int) { return -2; }
// ... and so are the dense-output instances of the observation kernel
struct MgxDev;
bool mgx_launch_obs_box(hipStream_t, const MgxDev&, size_t, int, int, const uint8_t*, int, int, int, bool, bool, bool, int, int, void*, const float*, int, int) { return false; }
bool mgx_obs_box_set_lds(size_t) { return true; }
// ... and the lane-per-agent action kernels (ballot / readlane): the sanitizer build never selects them (MgxDev::act_par = 0)
size_t mgx_act_fast_lds_bytes(int, int) { return 0; }
int mgx_act_fast_epg() { return 1; }
int mgx_act_x_epg() { return 1; }
void mgx_launch_act_fast_s0(bool, size_t, hipStream_t, const MgxDev&, int) {}
bool mgx_act_fast_set_lds_s0(size_t) { return true; }
void mgx_launch_act_x(bool, size_t, hipStream_t, const MgxDev&, const MgxDev*, int) {}
bool mgx_act_x_set_lds(size_t) { return true; }
size_t mgx_act_x_lds_bytes(int, bool, int) { return 0; }
