// mgx_decode.hip — policy-side token decode on the GPU: sparse observation tokens u8[rows][T][3] ->
// dense float32 box [rows][C][H][W], what the reference's GridObsWrapper._convert computes on the host
// (/root/reference/python/src/mettagrid/envs/grid_obs_wrapper.py:57-95): value / scale[feature] ADDED into cell
// (feature, y, x), tokens in row order, global tokens (location 0xFE) on the centre cell, padding (0xFF) skipped.
//
// One wavefront per agent row, four rows per workgroup.  The kernel is an HBM write stream (4 * C * H * W bytes per row,
// 26x the token bytes), so nothing of the box is staged: the row is zero-filled with 16-byte stores straight to HBM and the
// <= T token values are stored over it (same wavefront, program order).  LDS holds only the row's tokens and two bitmaps
// over the cells ("seen", "hit more than once"): 1.6 KB per row instead of the 15.5 KB box, so the CU is full of
// wavefronts that do nothing but store.
// Bit-exact sums: a cell that a single token targets gets that token's value (0 + v == v); the few cells several tokens
// share (tag tokens of one object) are summed by the first of their tokens, in token order, like np.add.at.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "mgx.h"

#define MGX_DEC_WAVES 4
#define MGX_DEC_WAVE 64

__global__ void __launch_bounds__(MGX_DEC_WAVES* MGX_DEC_WAVE) mgx_decode_kernel(const uint8_t* __restrict__ tokens, float* __restrict__ box,
                                                                                 const float* __restrict__ scale, long long rows, int T,
                                                                                 int C, int H, int W) {
  extern __shared__ __align__(16) uint8_t dec_lds[];
  const int wave = (int)threadIdx.x / MGX_DEC_WAVE, lane = (int)threadIdx.x & (MGX_DEC_WAVE - 1);
  const long long row = (long long)blockIdx.x * MGX_DEC_WAVES + wave;
  if (row >= rows) return;
  const int cells = C * H * W, bmw = (cells + 31) / 32;
  const int tok_words = (3 * T + 3) / 4, tok_pad = (tok_words + 3) & ~3;
  uint8_t* mine = dec_lds + (size_t)wave * ((size_t)tok_pad * 4 + (size_t)bmw * 8);
  uint8_t* stok = mine;
  uint32_t* seen = (uint32_t*)(mine + (size_t)tok_pad * 4);
  uint32_t* dup = seen + bmw;
  // ---- stage the token row, clear the bitmaps, zero-fill the row of the box in HBM ----
  {
    const uint8_t* src = tokens + row * (long long)T * 3;
    if ((((uintptr_t)src) & 3) == 0) {
      const uint32_t* s32 = (const uint32_t*)src;
      for (int i = lane; i < tok_words; i += MGX_DEC_WAVE) {
        // the last word of a row whose 3T is not a multiple of 4 reaches into the next row (or past the buffer for the
        // last row): read it bytewise
        if (i * 4 + 4 <= 3 * T) ((uint32_t*)stok)[i] = s32[i];
        else for (int b = i * 4; b < 3 * T; b++) stok[b] = src[b];
      }
    } else {
      for (int b = lane; b < 3 * T; b += MGX_DEC_WAVE) stok[b] = src[b];
    }
    for (int i = lane; i < 2 * bmw; i += MGX_DEC_WAVE) seen[i] = 0u;
  }
  float* dst = box + row * (long long)cells;
  if ((cells & 3) == 0 && (((uintptr_t)dst) & 15) == 0) {
    for (int i = lane; i < cells / 4; i += MGX_DEC_WAVE) ((uint4*)dst)[i] = make_uint4(0u, 0u, 0u, 0u);
  } else {
    for (int i = lane; i < cells; i += MGX_DEC_WAVE) dst[i] = 0.f;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // one wavefront per row: LDS operations complete in issue order; no need to wait for the zero-fill stores
  const int cy = H / 2, cx = W / 2;
  auto key_of = [&](int t) -> int {   // cell index of token t, or -1 (padding / outside the box)
    if (t >= T) return -1;
    const uint32_t coord = stok[3 * t], fid = stok[3 * t + 1];
    if (coord == 0xFFu) return -1;
    const int y = coord == 0xFEu ? cy : (int)(coord >> 4), x = coord == 0xFEu ? cx : (int)(coord & 0xF);
    if (y >= H || x >= W || (int)fid >= C) return -1;
    return ((int)fid * H + y) * W + x;
  };
  auto value_of = [&](int t) -> float { return __fdiv_rn((float)stok[3 * t + 2], scale[stok[3 * t + 1]]); };
  // ---- pass 1: which cells are targeted, which of them more than once ----
  for (int t = lane; t < T; t += MGX_DEC_WAVE) {
    const int k = key_of(t);
    if (k >= 0) {
      const uint32_t bit = 1u << (k & 31);
      if (atomicOr(&seen[k >> 5], bit) & bit) atomicOr(&dup[k >> 5], bit);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // one wavefront per row: LDS operations complete in issue order
  // The value stores below go to addresses OTHER lanes of this wavefront have just zero-filled: wait until every fill
  // store of the wavefront has been acknowledged, so that no value can be overtaken by a zero (pass 1 ran meanwhile).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // ---- pass 2: values over the zeros.  A cell with one token: that token's lane.  A shared cell: the lane of its first
  // token adds the later ones in token order. ----
  for (int t = lane; t < T; t += MGX_DEC_WAVE) {
    const int k = key_of(t);
    if (k < 0) continue;
    float v = value_of(t);
    if ((dup[k >> 5] >> (k & 31)) & 1u) {
      bool first = true;
      for (int u = 0; u < t && first; u++) first = key_of(u) != k;
      if (!first) continue;
      for (int u = t + 1; u < T; u++)
        if (key_of(u) == k) v = __fadd_rn(v, value_of(u));
    }
    dst[k] = v;
  }
}

size_t mgx_decode_lds_bytes(int T, int C, int H, int W) {
  const int bmw = (C * H * W + 31) / 32;
  const int tok_pad = (((3 * T + 3) / 4) + 3) & ~3;
  return (size_t)MGX_DEC_WAVES * ((size_t)tok_pad * 4 + (size_t)bmw * 8);
}

int mgx_launch_decode(hipStream_t stream, const uint8_t* tokens, float* box, const float* scale_dev, long long rows, int T, int C, int H,
                      int W) {
  const size_t lds = mgx_decode_lds_bytes(T, C, H, W);
  if (lds > 160 * 1024) return -1;
  static size_t cur_max[64] = {};   // the attribute is per device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -2;
  if (lds > cur_max[dev]) {
    if (hipFuncSetAttribute((const void*)mgx_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
    cur_max[dev] = lds;
  }
  const long long blocks = (rows + MGX_DEC_WAVES - 1) / MGX_DEC_WAVES;
  hipLaunchKernelGGL(mgx_decode_kernel, dim3((unsigned)blocks), dim3(MGX_DEC_WAVES * MGX_DEC_WAVE), lds, stream, tokens, box, scale_dev, rows, T, C,
                     H, W);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
