// mgx_decode.hip — policy-side token decode on the GPU: sparse observation tokens u8[rows][T][3] ->
// dense float32 box [rows][C][H][W], what the reference's GridObsWrapper._convert computes on the host
// (/root/reference/python/src/mettagrid/envs/grid_obs_wrapper.py:57-95): value / scale[feature] ADDED into cell
// (feature, y, x), tokens in row order, global tokens (location 0xFE) on the centre cell, padding (0xFF) skipped.
//
// One wavefront per agent row, four rows per workgroup.  The row's box is assembled in LDS and streamed out with
// 16-byte stores (the box is 26x the token bytes: the kernel is a pure HBM-write stream, 4 * C * H * W bytes per row).
// Bit-exact sums: tokens whose cell nobody else targets (almost all) are written by their own lane; the few cells several
// tokens share (tag tokens of one object) are accumulated by one lane in token order, like np.add.at.
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
  const int cells = C * H * W, cells4 = (cells + 3) & ~3;
  const int tok_words = (3 * T + 3) / 4, tok_pad = (tok_words + 3) & ~3;
  uint8_t* mine = dec_lds + (size_t)wave * ((size_t)cells4 * 4 + (size_t)tok_pad * 4);
  float* sbox = (float*)mine;
  uint32_t* cnt = (uint32_t*)mine;                  // the same cells as integer token counts during the first pass
  uint8_t* stok = mine + (size_t)cells4 * 4;
  // ---- stage the token row, zero the box ----
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
    for (int i = lane; i < cells4 / 4; i += MGX_DEC_WAVE) ((uint4*)sbox)[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  const int cy = H / 2, cx = W / 2;
  auto key_of = [&](int t) -> int {   // cell index of token t, or -1 (padding / outside the box)
    if (t >= T) return -1;
    const uint32_t coord = stok[3 * t], fid = stok[3 * t + 1];
    if (coord == 0xFFu) return -1;
    const int y = coord == 0xFEu ? cy : (int)(coord >> 4), x = coord == 0xFEu ? cx : (int)(coord & 0xF);
    if (y >= H || x >= W || (int)fid >= C) return -1;
    return ((int)fid * H + y) * W + x;
  };
  const int passes = (T + MGX_DEC_WAVE - 1) / MGX_DEC_WAVE;
  const bool fits = passes <= 8;   // per-lane token multiplicities live in registers: up to 512 tokens per row
  uint32_t mult[8];
#pragma unroll
  for (int p = 0; p < 8; p++) mult[p] = 0;
  if (fits) {
    // ---- pass 1: how many tokens target each cell (the box doubles as the counters) ----
    for (int t = lane; t < T; t += MGX_DEC_WAVE) {
      const int k = key_of(t);
      if (k >= 0) atomicAdd(&cnt[k], 1u);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    // ---- pass 2: every lane notes the multiplicity of its own tokens' cells, then the box is zeroed again ----
#pragma unroll
    for (int p = 0; p < 8; p++) {
      const int k = key_of(p * MGX_DEC_WAVE + lane);
      mult[p] = k >= 0 ? cnt[k] : 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    for (int i = lane; i < cells4 / 4; i += MGX_DEC_WAVE) ((uint4*)sbox)[i] = make_uint4(0u, 0u, 0u, 0u);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  }
  // ---- pass 3: values.  Unshared cells by their own lane; shared cells by lane 0, tokens in row order. ----
  if (fits) {
#pragma unroll
    for (int p = 0; p < 8; p++) {
      if (p >= passes) break;
      const int t = p * MGX_DEC_WAVE + lane;
      const int k = key_of(t);
      float v = 0.f;
      if (k >= 0) v = __fdiv_rn((float)stok[3 * t + 2], scale[stok[3 * t + 1]]);
      if (k >= 0 && mult[p] == 1u) sbox[k] = v;   // 0 + v == v
      unsigned long long shared = __ballot(k >= 0 && mult[p] > 1u);
      while (shared) {
        const int src = __ffsll((long long)shared) - 1;
        shared &= shared - 1;
        const int sk = __shfl(k, src);
        const float sv = __shfl(v, src);
        if (lane == 0) sbox[sk] = __fadd_rn(sbox[sk], sv);
      }
    }
  } else if (lane == 0) {
    for (int t = 0; t < T; t++) {
      const int k = key_of(t);
      if (k >= 0) sbox[k] = __fadd_rn(sbox[k], __fdiv_rn((float)stok[3 * t + 2], scale[stok[3 * t + 1]]));
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  // ---- stream the box out ----
  float* dst = box + row * (long long)cells;
  if ((cells & 3) == 0 && (((uintptr_t)dst) & 15) == 0) {
    for (int i = lane; i < cells / 4; i += MGX_DEC_WAVE) ((uint4*)dst)[i] = ((const uint4*)sbox)[i];
  } else {
    for (int i = lane; i < cells; i += MGX_DEC_WAVE) dst[i] = sbox[i];
  }
}

size_t mgx_decode_lds_bytes(int T, int C, int H, int W) {
  const int cells4 = (C * H * W + 3) & ~3;
  const int tok_pad = (((3 * T + 3) / 4) + 3) & ~3;
  return (size_t)MGX_DEC_WAVES * ((size_t)cells4 * 4 + (size_t)tok_pad * 4);
}

int mgx_launch_decode(hipStream_t stream, const uint8_t* tokens, float* box, const float* scale_dev, long long rows, int T, int C, int H,
                      int W) {
  const size_t lds = mgx_decode_lds_bytes(T, C, H, W);
  if (lds > 160 * 1024) return -1;
  static size_t cur_max = 0;
  if (lds > cur_max) {
    if (hipFuncSetAttribute((const void*)mgx_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
    cur_max = lds;
  }
  const long long blocks = (rows + MGX_DEC_WAVES - 1) / MGX_DEC_WAVES;
  hipLaunchKernelGGL(mgx_decode_kernel, dim3((unsigned)blocks), dim3(MGX_DEC_WAVES * MGX_DEC_WAVE), lds, stream, tokens, box, scale_dev, rows, T, C,
                     H, W);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
