// mgx_pack.hip — packed form of observation rows for the multi-GPU gather (include/mgx.h mgx_pack_rows).
//
// A row of the observation buffer is a PREFIX of used tokens followed by 0xFF padding (the reference clears the buffer to
// 0xFF and appends tokens in order: cpp/bindings/mettagrid_c.cpp:319-375), and at BASELINE's shapes a row is about one
// third full.  The gather to the trainer rank is bound by the root's xGMI links (7 x ~153 GB/s), so what crosses them is the
// used prefix only: per row a u16 token count, and all used tokens back to back.  Three kernels on the caller's stream:
//   count   one thread per row: binary search for the first padding token (location byte 0xFF)
//   scan    exclusive prefix sum of the counts (4 096 rows per workgroup, then one workgroup over the workgroup sums)
//   copy    one wavefront per row: its used bytes to packed + 3 * offset[row], 64 contiguous bytes per instruction
// and the inverse (mgx_unpack_rows: offsets again from the counts, rows rebuilt with their 0xFF padding).  Byte-exact.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgx.h"

#define MGX_PK_WAVE 64
#define MGX_PK_SCAN_THREADS 1024
#define MGX_PK_SCAN_ITEMS 4   // rows per thread of the scan

__global__ void __launch_bounds__(256) mgx_pack_count_kernel(const uint8_t* __restrict__ rows, long long n_rows, int T, uint16_t* __restrict__ counts) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const uint8_t* row = rows + (size_t)r * T * 3;
  int lo = 0, hi = T;   // first index whose location byte is 0xFF lies in [lo, hi]
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (row[mid * 3] != 0xFF) lo = mid + 1; else hi = mid;
  }
  counts[r] = (uint16_t)lo;
}

// offsets[r] = sum of counts[0 .. r); offsets[n_rows] = total.  Level 1: per workgroup; block_sums[b] = the workgroup's total.
__global__ void __launch_bounds__(MGX_PK_SCAN_THREADS) mgx_pack_scan1_kernel(const uint16_t* __restrict__ counts, long long n_rows,
                                                                             uint32_t* __restrict__ offsets, uint32_t* __restrict__ block_sums) {
  __shared__ uint32_t s_wave[MGX_PK_SCAN_THREADS / MGX_PK_WAVE];
  const int tid = threadIdx.x, lane = tid & (MGX_PK_WAVE - 1), wave = tid / MGX_PK_WAVE;
  const long long r0 = ((long long)blockIdx.x * MGX_PK_SCAN_THREADS + tid) * MGX_PK_SCAN_ITEMS;
  uint32_t v[MGX_PK_SCAN_ITEMS], sum = 0;
  for (int i = 0; i < MGX_PK_SCAN_ITEMS; i++) { v[i] = r0 + i < n_rows ? counts[r0 + i] : 0u; sum += v[i]; }
  uint32_t incl = sum;
  for (int o = 1; o < MGX_PK_WAVE; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (lane >= o) incl += t; }
  if (lane == MGX_PK_WAVE - 1) s_wave[wave] = incl;
  __syncthreads();
  uint32_t base = incl - sum;
  for (int w = 0; w < wave; w++) base += s_wave[w];
  for (int i = 0; i < MGX_PK_SCAN_ITEMS; i++) { if (r0 + i < n_rows) offsets[r0 + i] = base; base += v[i]; }
  if (tid == MGX_PK_SCAN_THREADS - 1) block_sums[blockIdx.x] = base;
}
// Level 2: one workgroup turns block_sums into exclusive prefixes (in place) and stores the grand total at offsets[n_rows].
__global__ void __launch_bounds__(MGX_PK_SCAN_THREADS) mgx_pack_scan2_kernel(uint32_t* __restrict__ block_sums, int n_blocks, uint32_t* __restrict__ offsets,
                                                                             long long n_rows) {
  __shared__ uint32_t s_wave[MGX_PK_SCAN_THREADS / MGX_PK_WAVE];
  __shared__ uint32_t s_carry;
  const int tid = threadIdx.x, lane = tid & (MGX_PK_WAVE - 1), wave = tid / MGX_PK_WAVE;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < n_blocks; b0 += MGX_PK_SCAN_THREADS) {
    const int b = b0 + tid;
    const uint32_t v = b < n_blocks ? block_sums[b] : 0u;
    uint32_t incl = v;
    for (int o = 1; o < MGX_PK_WAVE; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == MGX_PK_WAVE - 1) s_wave[wave] = incl;
    __syncthreads();
    uint32_t base = s_carry + incl - v;
    for (int w = 0; w < wave; w++) base += s_wave[w];
    if (b < n_blocks) block_sums[b] = base;
    __syncthreads();
    if (tid == MGX_PK_SCAN_THREADS - 1) s_carry = base + v;
    __syncthreads();
  }
  if (tid == 0) offsets[n_rows] = s_carry;
}
__global__ void __launch_bounds__(256) mgx_pack_scan3_kernel(uint32_t* __restrict__ offsets, const uint32_t* __restrict__ block_sums, long long n_rows) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n_rows) offsets[r] += block_sums[r / (MGX_PK_SCAN_THREADS * MGX_PK_SCAN_ITEMS)];
}

// one wavefront per row; capacity in tokens (rows that would run past it are dropped and counted in *overflow)
__global__ void __launch_bounds__(256) mgx_pack_copy_kernel(const uint8_t* __restrict__ rows, long long n_rows, int T, const uint16_t* __restrict__ counts,
                                                            const uint32_t* __restrict__ offsets, uint8_t* __restrict__ packed, long long capacity,
                                                            uint32_t* __restrict__ overflow) {
  const int lane = threadIdx.x & (MGX_PK_WAVE - 1);
  const long long r = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / MGX_PK_WAVE;
  if (r >= n_rows) return;
  const int n = counts[r] * 3;
  const long long off = offsets[r];
  if (off + counts[r] > capacity) { if (lane == 0 && counts[r]) atomicAdd(overflow, 1u); return; }
  const uint8_t* src = rows + (size_t)r * T * 3;
  uint8_t* dst = packed + (size_t)off * 3;
  for (int b = lane; b < n; b += MGX_PK_WAVE) dst[b] = src[b];
}
__global__ void __launch_bounds__(256) mgx_unpack_copy_kernel(const uint8_t* __restrict__ packed, const uint16_t* __restrict__ counts,
                                                              const uint32_t* __restrict__ offsets, long long n_rows, int T, uint8_t* __restrict__ rows) {
  const int lane = threadIdx.x & (MGX_PK_WAVE - 1);
  const long long r = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / MGX_PK_WAVE;
  if (r >= n_rows) return;
  const int n = counts[r] * 3;
  const uint8_t* src = packed + (size_t)offsets[r] * 3;
  uint8_t* dst = rows + (size_t)r * T * 3;
  for (int b = lane; b < T * 3; b += MGX_PK_WAVE) dst[b] = b < n ? src[b] : (uint8_t)0xFF;
}

static int pk_scan(const uint16_t* counts, long long n_rows, uint32_t* scratch, hipStream_t st) {
  const long long per = (long long)MGX_PK_SCAN_THREADS * MGX_PK_SCAN_ITEMS;
  const int nb = (int)((n_rows + per - 1) / per);
  uint32_t* offsets = scratch;                 // [n_rows + 1]
  uint32_t* block_sums = scratch + n_rows + 1; // [nb]
  hipLaunchKernelGGL(mgx_pack_scan1_kernel, dim3((unsigned)nb), dim3(MGX_PK_SCAN_THREADS), 0, st, counts, n_rows, offsets, block_sums);
  hipLaunchKernelGGL(mgx_pack_scan2_kernel, dim3(1), dim3(MGX_PK_SCAN_THREADS), 0, st, block_sums, nb, offsets, n_rows);
  hipLaunchKernelGGL(mgx_pack_scan3_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, offsets, (const uint32_t*)block_sums, n_rows);
  return hipGetLastError() == hipSuccess ? MGX_OK : MGX_ERR_HIP;
}

extern "C" {

int64_t mgx_pack_scratch_bytes(int64_t n_rows) {
  if (n_rows < 0) return 0;
  const int64_t per = (int64_t)MGX_PK_SCAN_THREADS * MGX_PK_SCAN_ITEMS;
  return (n_rows + 1 + (n_rows + per - 1) / per + 1 /* overflow counter */) * 4;
}

int mgx_pack_rows(const uint8_t* rows, int64_t n_rows, int32_t n_tokens, uint16_t* counts, uint8_t* packed, int64_t capacity_tokens,
                  uint32_t* scratch, void* hip_stream) {
  if (!rows || !counts || !packed || !scratch || n_rows <= 0 || n_tokens <= 0 || n_tokens > 65535 || capacity_tokens < 0) return MGX_ERR_BAD_ARG;
  if (n_rows * (int64_t)n_tokens >= (1ll << 32)) return MGX_ERR_BAD_ARG;   // offsets are 32-bit token indices
  hipStream_t st = (hipStream_t)hip_stream;
  const int64_t per = (int64_t)MGX_PK_SCAN_THREADS * MGX_PK_SCAN_ITEMS;
  uint32_t* overflow = scratch + n_rows + 1 + (n_rows + per - 1) / per;
  if (hipMemsetAsync(overflow, 0, 4, st) != hipSuccess) return MGX_ERR_HIP;
  hipLaunchKernelGGL(mgx_pack_count_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, rows, (long long)n_rows, (int)n_tokens, counts);
  if (pk_scan(counts, n_rows, scratch, st) != MGX_OK) return MGX_ERR_HIP;
  hipLaunchKernelGGL(mgx_pack_copy_kernel, dim3((unsigned)((n_rows * MGX_PK_WAVE + 255) / 256)), dim3(256), 0, st, rows, (long long)n_rows, (int)n_tokens,
                     (const uint16_t*)counts, (const uint32_t*)scratch, packed, (long long)capacity_tokens, overflow);
  return hipGetLastError() == hipSuccess ? MGX_OK : MGX_ERR_HIP;
}

int mgx_pack_result(const uint32_t* scratch, int64_t n_rows, int64_t* total_tokens, int32_t* overflow_rows, void* hip_stream) {
  if (!scratch || !total_tokens || n_rows <= 0) return MGX_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)hip_stream;
  const int64_t per = (int64_t)MGX_PK_SCAN_THREADS * MGX_PK_SCAN_ITEMS;
  uint32_t tot = 0, ovf = 0;
  if (hipMemcpyAsync(&tot, scratch + n_rows, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return MGX_ERR_HIP;
  if (hipMemcpyAsync(&ovf, scratch + n_rows + 1 + (n_rows + per - 1) / per, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return MGX_ERR_HIP;
  if (hipStreamSynchronize(st) != hipSuccess) return MGX_ERR_HIP;
  *total_tokens = (int64_t)tot;
  if (overflow_rows) *overflow_rows = (int32_t)ovf;
  return MGX_OK;
}

int mgx_unpack_rows(const uint8_t* packed, const uint16_t* counts, int64_t n_rows, int32_t n_tokens, uint8_t* rows, uint32_t* scratch, void* hip_stream) {
  if (!packed || !counts || !rows || !scratch || n_rows <= 0 || n_tokens <= 0 || n_tokens > 65535) return MGX_ERR_BAD_ARG;
  if (n_rows * (int64_t)n_tokens >= (1ll << 32)) return MGX_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)hip_stream;
  if (pk_scan(counts, n_rows, scratch, st) != MGX_OK) return MGX_ERR_HIP;
  hipLaunchKernelGGL(mgx_unpack_copy_kernel, dim3((unsigned)((n_rows * MGX_PK_WAVE + 255) / 256)), dim3(256), 0, st, packed, counts, (const uint32_t*)scratch,
                     (long long)n_rows, (int)n_tokens, rows);
  return hipGetLastError() == hipSuccess ? MGX_OK : MGX_ERR_HIP;
}

}  // extern "C"
