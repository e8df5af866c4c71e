// mgx_obs.h — observation / reward kernel: one workgroup (4 wavefronts) per env.
//
// Covers phases 12-14 of MettaGrid::_step (/root/reference/cpp/bindings/mettagrid_c.cpp:1062-1096):
// the tokenised local-window observation of every agent (_compute_observation_optimized :665-824), the per-agent
// reward entries (systems/reward.hpp:56-77) and truncation/termination.  ~91 % of the reference's CPU step time is
// spent here, so this is the dominant kernel and the one the HBM roofline is quoted for.
//
// Mapping: the env's H*W u16 occupancy grid is streamed once into LDS with coalesced 16-byte loads; each wavefront
// then encodes agents w, w+4, ... : lanes map to window cells in the reference's Manhattan-shell order, per-cell
// token counts are turned into output slots with a wavefront exclusive scan (ordered compaction), tokens are
// assembled in an LDS staging row that starts out as 0xFF (the reference's separate memset of the whole buffer is
// fused away) and the finished 3*T-byte row is written to HBM with coalesced dword stores.
// The reference's "first observer of an object this step gets the cell.visited staleness" rule (:789-796, serial
// agent order) becomes an LDS atomicMin over observer indices followed by an in-order wavefront reduction.
//
// The kernel is instruction-issue bound (profiles/), so the hot loops are written branch-free: LDS reads use clamped
// indices and selects instead of nested divergent ifs, predicated LDS stores go to a per-wave trash word instead of
// toggling EXEC, and everything that is the same for all observers is computed once per env (token lists, global
// tokens, the window -> slot map of every agent).
#ifndef MGX_OBS_H_
#define MGX_OBS_H_

#include <type_traits>

#include "mgx_world.h"

#define MGX_OBS_THREADS 256
enum { MGX_BOX_OFF = 0, MGX_BOX_F32 = 1, MGX_BOX_BF16 = 2 };   // mgx_set_box_output dtypes
#ifdef MGX_OBS_STOP  // profiling builds only: leave the kernel after phase k to attribute its time (scripts/obs_phases.sh)
#define MGX_PHASE_END(k) do { if (MGX_OBS_STOP == (k)) return; } while (0)
#else
#define MGX_PHASE_END(k)
#endif
#define MGX_OBS_WAVES (MGX_OBS_THREADS / MGX_WAVE)

// Wavefront inclusive scan with DPP row shifts / row broadcasts (no LDS traffic): Hillis-Steele inside each 16-lane
// row, then row 0->1, 2->3 (row_bcast:15) and rows {0,1}->{2,3} (row_bcast:31).
__device__ __forceinline__ int mgx_wave_incl_scan(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
  return x;
}

// Inclusive scan inside each 16-lane DPP row (Hillis-Steele, four row shifts, zero fill at the row start).
__device__ __forceinline__ int mgx_row_incl_scan(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
  return x;
}

__device__ __forceinline__ int mgx_digits(uint32_t v, uint32_t base) {  // encoding_utils.hpp:39-62
  int n = 1;
  v /= base;
  while (v > 0) { n++; v /= base; }
  return n;
}

// ---------------------------------------------------------------------------------------------------------------
// Per-object token cache.  GridObject::write_obs_features / Agent::write_obs_features (core/grid_object.cpp:178-203,
// objects/agent.cpp:142-154) produce the same (feature, value) list for every observer of an object — only the
// location byte differs.  One thread per object slot builds that list ONCE per env and step into an LDS pool
// (u16 = feature | value << 8); the per-agent loop then only copies from LDS.  The tag part of the list is static per
// class and comes from a table the host builds at mgx_create (MgxDev::cls_tok*).
// ---------------------------------------------------------------------------------------------------------------
struct MgxObjTok {
  uint16_t* pool;
  int pos;
  __device__ __forceinline__ void put(uint32_t f, uint32_t v) { pool[pos++] = (uint16_t)((f & 0xFF) | ((v & 0xFF) << 8)); }
};

// Digit arithmetic of the token encoder (encoding_utils.hpp:39-62).  token_value_base is a power of two in every
// shipped config (default 256): that case is shift/mask; any other base takes the division path.
struct MgxBase {
  uint32_t base;
  int shift;  // log2(base) if base is a power of two, else -1
  __device__ __forceinline__ explicit MgxBase(uint32_t b) : base(b), shift((b & (b - 1)) == 0 ? __ffs(b) - 1 : -1) {}
  __device__ __forceinline__ uint32_t lo(uint32_t v) const { return shift >= 0 ? (v & (base - 1)) : v % base; }
  __device__ __forceinline__ uint32_t hi(uint32_t v) const { return shift >= 0 ? (v >> shift) : v / base; }
  __device__ __forceinline__ int digits(uint32_t v) const {
    int n = 1;
    for (v = hi(v); v > 0; v = hi(v)) n++;
    return n;
  }
};

#define MGX_MAX_ITEMS 13  // resources per inventory (mgx_create enforces R <= 13: 4-bit ids, 0xF terminator)

// Dynamic LDS layout, all regions 16-byte aligned (the host calls the same function):
//   grid u16[HW] | offsets i8x2[NOFF] | loc u8[CP] (packed window coordinate of offset j) | minobs u32[S+1] |
//   visited u32[S] | tokinfo u32[S] (start | count << 16) | dyn u16[S] (slots whose token list is built per step) |
//   agents u32[A] (slot | rc << 16) | aginfo u32[A] | spawn u16[A] | vstat f32[A] | written i32[A] |
//   rwinfo u32[A] (reward start | count << 16) | misc u32[4] | trash u32[64] | rows u32[WAVES*4][Tpad] |
//   cell u16[A][CP], vj u8[A][CP], vcount u32[A]: per agent the occupied window cells in window order (slot + 1 and
//   window index j of each), compacted | X: vmask u8[A][CP] territory token of each listed cell, obs values |
//   blk i32[blk_words]: program sections INV_FEATURES..OBS_VALUES (PL variants only) | gtok u32[A][GT] global tokens |
//   pool u16[POOL+8] per-object token lists (lean games: last; extended games: behind misc)
// CP = NOFF rounded up to 16: stride of the per-agent lists.
__host__ __device__ inline int mgx_align16(int x) { return (x + 15) & ~15; }
struct MgxObsLds {
  int grid, offs, loc, minobs, visited, tokinfo, dyn, agents, aginfo, spawn, vstat, written, rwinfo, misc, pool, rows, row_words,
      row_pitch, cell, vj, vcount, cp, cs, blk, gtok, trash, total;
  int owner, obsval, tscore, vmask, agtags;  // X only: per-cell territory owner u16[HW] (overlaid), obs values u32[A][NOV], mask list
};
// Upper bound of the global (location 0xFE) tokens of one agent: completion, last action, last action move, last
// reward, two local-position tokens and every digit of every obs value (a u32 has at most 32 digits in base 2).
__host__ __device__ inline int mgx_obs_gt(int NOV, int base, int flags) {
  int digits = 1;
  for (unsigned long long v = 0xFFFFFFFFull / (unsigned)base; v > 0; v /= (unsigned)base) digits++;
  const int fixed = ((flags & MGX_G_COMPLETION) ? 1 : 0) + ((flags & MGX_G_LAST_ACTION) ? 1 : 0) + ((flags & MGX_G_LAST_ACTION_MOVE) ? 1 : 0) +
                    ((flags & MGX_G_LAST_REWARD) ? 1 : 0) + ((flags & MGX_G_LOCAL_POSITION) ? 2 : 0);
  return (fixed > 0 ? fixed : 1) + NOV * digits;
}
// Lean games whose slot numbers and window indices fit 9 + 7 bits keep the window index of a listed cell in the cell entry
// itself (no separate index list) and pack the lists at the window size rounded up to even instead of to 16.
__host__ __device__ inline bool mgx_obs_pack_j(bool X, int S, int NOFF) { return !X && S + 1 <= 512 && NOFF <= 128; }
// xmode of the extended variant: what the territory observability mask needs in LDS (0 = lean kernel)
enum { MGX_OX_ON = 1, MGX_OX_MASK = 2, MGX_OX_PACK = 4, MGX_OX_OWNER8 = 8 };
__host__ __device__ inline int mgx_obs_xmode(bool X, bool want_mask, int S, int num_tags) {
  if (!X) return 0;
  int m = MGX_OX_ON;
  if (want_mask) m |= MGX_OX_MASK | (S + 1 < (1 << 14) ? MGX_OX_PACK : 0) | (num_tags <= 255 ? MGX_OX_OWNER8 : 0);
  // bits 4-7: words of an agent's tag bitset that can hold a tag of this program (owner tags are tag ids < num_tags)
  int tw = (num_tags + 31) / 32;
  tw = tw < 1 ? 1 : tw > MGX_TAG_WORDS ? MGX_TAG_WORDS : tw;
  m |= tw << 4;
  return m;
}
__host__ __device__ inline MgxObsLds mgx_obs_lds_layout(int HW, int NOFF, int S, int A, int T, int pool_tokens,
                                                        int xmode = 0, int NOV = 0, int blk_words = 0, int GT = 6,
                                                        bool blk_early = false, int waves = MGX_OBS_WAVES) {
  const bool X = xmode != 0;
  const bool pack_j = mgx_obs_pack_j(X, S, NOFF);
  MgxObsLds l;
  int o = 0;
  l.cp = (NOFF + 15) & ~15;
  l.cs = pack_j ? ((NOFF + 1) & ~1) : l.cp;
  // The grid, the window offsets, the list-builder queue, the territory owner map (X) and (when the rewards are
  // evaluated early, blk_early) the program block are dead once the visible-cell lists and the global tokens exist
  // (barrier before the encode phase); the staging rows are first written after it, so the rows overlay them.
  l.loc = o; o += mgx_align16(l.cp);
  l.minobs = o; o += mgx_align16((S + 1) * 4);
  l.visited = o; o += mgx_align16(S * 4);
  l.tokinfo = o; o += mgx_align16(S * 4);
  l.agents = o; o += mgx_align16(A * 4);
  l.aginfo = o; o += mgx_align16(A * 4);
  l.spawn = o; o += mgx_align16(A * 2);
  l.vstat = o; o += mgx_align16(A * 4);
  l.written = o; o += mgx_align16(A * 4);
  l.rwinfo = o; o += mgx_align16(A * 4);
  l.misc = o; o += 16;
  l.trash = o; o += MGX_WAVE * 4;
  // The token pool's size follows the maps (run-time); every other region only the program's shape.  Lean games put it
  // LAST, so that a kernel specialised for a shape (MgxObsShape) has every other offset as a compile-time constant; the
  // extended variant keeps it here (measured at rung 4: 4.76 ms here, 5.01 ms at the end of the layout).
  const bool pool_last = !X;
  if (!pool_last) { l.pool = o; o += mgx_align16((pool_tokens + 8) * 2); }
  l.row_words = (T + 3) & ~3;
  // Masked-off stores of the encode land in a trash word of the storing lane: one block of 64 words for the workgroup (its
  // wavefronts never collide inside one instruction) instead of 16 words behind every row — 768 B less, which is what takes
  // the rung-3 layout from 23 088 to 22 320 B and under the 23 040 B that seven workgroups per CU allow (LDS is handed out
  // in 1 280-byte granules; measured: the 23 088-byte layout ran as fast with 400 B of padding as without, i.e. at six).
  l.row_pitch = l.row_words;
#ifndef MGX_OBS_PITCH_OLD
  // ... + padding to 8 words mod 16: the two DPP rows of a 32-lane LDS group write to staging rows two pitches apart
  // (row permutation in the kernel), and 2 * pitch = 16 mod 32 keeps their 16-word spans on disjoint banks for every T
  // (T = 200: 216, no padding; T = 256: 272 -> 280, where both rows of a half started on the same bank).
  l.row_pitch += (8 - (l.row_pitch & 15)) & 15;
#endif
  l.rows = o;
  l.owner = 0;
  {
    const int rows_bytes = waves * 4 * l.row_pitch * 4;  // one staging row per DPP row of every wavefront
    int eo = o;
    l.grid = eo; eo += mgx_align16(HW * 2);
    l.offs = eo; eo += mgx_align16(NOFF * 2);
    l.dyn = eo; eo += mgx_align16(S * 2);
    if (xmode & MGX_OX_MASK) { l.owner = eo; eo += mgx_align16((xmode & MGX_OX_OWNER8) ? HW : HW * 2); }
    if (blk_early) { l.blk = eo; eo += mgx_align16(blk_words * 4); }
    const int early_bytes = eo - o;
    o += rows_bytes > early_bytes ? rows_bytes : early_bytes;
  }
  l.cell = o; o += mgx_align16(A * l.cs * 2);
  l.vj = o; if (!pack_j) o += mgx_align16(A * l.cp);
  l.vcount = o; o += mgx_align16(A * 4);
  l.vmask = l.obsval = l.tscore = l.agtags = 0;
  if (X) {
    // per visible cell: 0 no territory token, 1 friendly, 2 foreign owner — in bits 14-15 of the cell entry when slot
    // numbers leave them free (MGX_OX_PACK), else a byte list of its own
    if ((xmode & MGX_OX_MASK) && !(xmode & MGX_OX_PACK)) { l.vmask = o; o += mgx_align16(A * l.cp); }
    if (xmode & MGX_OX_MASK) { l.agtags = o; o += mgx_align16(A * ((xmode >> 4) & 15) * 4); }  // the agents' own tag bitsets (friend / foe of a cell's owner tag)
    l.obsval = o; o += mgx_align16((A * NOV + 1) * 4);
  }
  if (!blk_early) { l.blk = o; o += mgx_align16(blk_words * 4); }
  l.gtok = o; o += mgx_align16(A * GT * 4);
  if (pool_last) { l.pool = o; o += mgx_align16((pool_tokens + 8) * 2); }
  l.total = o;
  return l;
}

// Wavefront sum (every lane gets the total).
__device__ __forceinline__ uint32_t mgx_wave_sum(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_readlane(mgx_wave_incl_scan((int)x), MGX_WAVE - 1);
}

// Shape policy of the kernel.  MgxObsShapeDyn: every size comes from the engine's table at run time (any program).
// MgxObsShape<...>: the sizes of ONE compiled program as compile-time constants — map, agents, slots, token budget,
// window, value base, interpreted block, global-token flags — generated for the benchmark presets at build() time
// (mgx_presets_gen.h); the engine launches that instance when a program's shape equals it (mgx_create) and the generic
// one otherwise.  With the shape fixed the LDS layout offsets, the loop bounds over agents / window cells and the index
// arithmetic are immediates: the generic kernel spent as many scalar as vector instructions, most of them on exactly
// that arithmetic, and ~290 v_readlane reloads of spilled layout offsets.
struct MgxObsShapeDyn {
  static constexpr bool fixed = false;
  static constexpr int H = 0, W = 0, A = 0, S = 0, T = 0, NOFF = 0, BASE = 0, NOV = 0, NT = 0, NRW = 0, FLAGS = 0, MAX_STEPS = 0, BLKW = 0,
                       MASK_FEAT = 0, REWARDS_EARLY = 0;
};
template <int H_, int W_, int A_, int S_, int T_, int NOFF_, int BASE_, int NOV_, int NT_, int NRW_, int FLAGS_, int MAX_STEPS_, int BLKW_,
          int MASK_FEAT_, int REWARDS_EARLY_>
struct MgxObsShape {
  static constexpr bool fixed = true;
  static constexpr int H = H_, W = W_, A = A_, S = S_, T = T_, NOFF = NOFF_, BASE = BASE_, NOV = NOV_, NT = NT_, NRW = NRW_, FLAGS = FLAGS_,
                       MAX_STEPS = MAX_STEPS_, BLKW = BLKW_, MASK_FEAT = MASK_FEAT_, REWARDS_EARLY = REWARDS_EARLY_;
};
// does an engine table + launch parameters have this shape?  (host)
template <class K>
inline bool mgx_obs_shape_matches(const MgxDev& d, int blk_words, int rewards_early) {
  return K::fixed && d.H == K::H && d.W == K::W && d.A == K::A && d.S == K::S && d.T == K::T && d.NOFF == K::NOFF && d.base == K::BASE &&
         d.n_obs_values == K::NOV && d.NT == K::NT && d.NRW == K::NRW && d.flags == K::FLAGS && (K::MAX_STEPS < 0 || d.max_steps == K::MAX_STEPS) &&   // (MAX_STEPS = -1: any episode length, read at run time)
         blk_words == K::BLKW && d.aoe_mask_feat == K::MASK_FEAT && rewards_early == K::REWARDS_EARLY;
}

// PL: the program sections this kernel interprets (inventory feature ids, game-value code, reward records, obs
// values: [blk_start, blk_start + blk_words) of the blob, contiguous because the compiler lays sections out in id
// order) are copied into LDS so the reward / obs-value interpreter and the token builder never wait on global loads
// for program words.  The host picks PL when the block is small (mgx_create) and passes a MgxDev whose section
// offsets for exactly those sections are relative to the block start, so the LDS view is a plain base pointer (no
// out-of-object pointer arithmetic on an LDS address).
// pool_prefix: the first pool_prefix pool entries are the per-class static tag tokens (a copy of MgxDev::cls_tok), which
// is all a static object (walls ...) ever shows; only the other objects get a per-step list behind them.
// NTH: threads per workgroup — 256, or 512 for envs with many agents (more wavefronts over the same per-env LDS).
// EW: wavefronts that take part in the encode (each owns four staging rows in LDS); the others wait at the barrier behind it.
// BOX (fused form): false = token rows u8[T][3] to MgxDev::obs (the reference's buffer); true = the policy's
// dense input instead — the box [C][H][W] of GridObsWrapper._convert (python/src/mettagrid/envs/grid_obs_wrapper.py:57-95:
// value / scale[feature] added into cell (feature, y, x) in token order, global tokens on the centre cell) as float32
// (box_dtype MGX_BOX_F32) or bfloat16 (MGX_BOX_BF16: the float32 sum rounded to nearest even once) — written straight from
// the LDS staging row; the token buffer is neither written nor read back.
// Which envs: all of them (one workgroup per env, grid = E) — the step; the masked ones (env_mask) or, WITHOUT rewards only,
// the entries of a device list of unknown length walked by a small fixed grid (env_list / env_list_n) — the initial
// observations of restarted envs.
template <bool WITH_REWARDS, bool X, bool PL, int NTH, int EW, class K, bool BOX>
static __device__ __forceinline__ void mgx_obs_env(const MgxDev& d, const int env, uint8_t* smem, int pool_tokens, int pool_prefix,
                                                   const uint8_t* env_mask, int blk_start, int blk_words_arg, int rewards_early_arg,
                                                   void* box_out, const float* box_scale, int box_C, int box_dtype, int stat_passes);
// The kernel's body (also what the run-time specialised code objects wrap: mgx_jit_obs.hip).
template <bool WITH_REWARDS, bool X, bool PL, int NTH, int EW, class K, bool BOX>
static __device__ __forceinline__ void mgx_obs_run(const MgxDev& d, int pool_tokens, int pool_prefix, const uint8_t* env_mask, int blk_start,
                                                   int blk_words_arg, int rewards_early_arg, void* box_out, const float* box_scale, int box_C,
                                                   int box_dtype, const int32_t* env_list, const uint32_t* env_list_n, int stat_passes) {
  extern __shared__ __align__(16) uint8_t smem[];
  const bool listed = !WITH_REWARDS && env_list != nullptr;   // (constant false in the instances the step launches)
  const int n_listed = listed ? (int)*env_list_n : 0;
  for (int k = blockIdx.x;; k += gridDim.x) {
    if (listed && k >= n_listed) break;
    mgx_obs_env<WITH_REWARDS, X, PL, NTH, EW, K, BOX>(d, listed ? env_list[k] : (int)blockIdx.x, smem, pool_tokens, pool_prefix, env_mask, blk_start,
                                                      blk_words_arg, rewards_early_arg, box_out, box_scale, box_C, box_dtype, stat_passes);
    if (!listed) break;
    __syncthreads();   // the next env's staging overwrites this one's LDS
  }
}
template <bool WITH_REWARDS, bool X, bool PL, int NTH = MGX_OBS_THREADS, int EW = NTH / MGX_WAVE, class K = MgxObsShapeDyn, bool BOX = false>
__global__ void __launch_bounds__(NTH) mgx_obs_kernel(MgxDev d, int pool_tokens, int pool_prefix, const uint8_t* env_mask,
                                                                  int blk_start, int blk_words_arg, int rewards_early_arg,
                                                                  void* box_out, const float* box_scale, int box_C, int box_dtype,
                                                                  const int32_t* env_list, const uint32_t* env_list_n, int stat_passes) {
  MGX_KERNARG_ENTRY(d);
  mgx_obs_run<WITH_REWARDS, X, PL, NTH, EW, K, BOX>(d, pool_tokens, pool_prefix, env_mask, blk_start, blk_words_arg, rewards_early_arg, box_out,
                                                    box_scale, box_C, box_dtype, env_list, env_list_n, stat_passes);
}
template <bool WITH_REWARDS, bool X, bool PL, int NTH, int EW, class K, bool BOX>
static __device__ __forceinline__ void mgx_obs_env(const MgxDev& d, const int env, uint8_t* smem, int pool_tokens, int pool_prefix,
                                                   const uint8_t* env_mask, int blk_start, int blk_words_arg, int rewards_early_arg,
                                                   void* box_out, const float* box_scale, int box_C, int box_dtype, int stat_passes) {
  if (env_mask && !env_mask[env]) return;  // episode restart: only the restarted envs get initial observations
  const int tid = threadIdx.x, lane = tid & (MGX_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / MGX_WAVE);  // wave-uniform: per-agent values and branches go scalar
  constexpr bool FX = K::fixed;   // the shape is a compile-time constant (every FX ? K::x : d.x below folds)
  const int dH = FX ? K::H : d.H, dW = FX ? K::W : d.W, A = FX ? K::A : d.A, S = FX ? K::S : d.S, T = FX ? K::T : d.T,
            NOFF = FX ? K::NOFF : d.NOFF, HW = dH * dW;
  const int dBase = FX ? K::BASE : d.base, NOV = FX ? K::NOV : d.n_obs_values, dNT = FX ? K::NT : d.NT, dNRW = FX ? K::NRW : d.NRW,
            dFlags = FX ? K::FLAGS : d.flags, dMaxSteps = (FX && K::MAX_STEPS >= 0) ? K::MAX_STEPS : d.max_steps, dMaskFeat = FX ? K::MASK_FEAT : d.aoe_mask_feat;
  const int blk_words = FX ? K::BLKW : blk_words_arg, rewards_mode = FX ? K::REWARDS_EARLY : rewards_early_arg;
  // when the reward expressions read nothing this kernel writes (host: mgx_create), they need not wait for its end:
  //   1 = evaluated early, beside the token-list phase (lean games);
  //   2 = evaluated by a wavefront that takes no part in the encode, during the encode (EW < wavefronts: extended games
  //       with many agents), instead of by 64 threads behind the final barrier while everybody else idles
  const int rewards_early = rewards_mode == 1 ? 1 : 0;
  const bool rewards_mid = rewards_mode == 2 && EW < NTH / MGX_WAVE;
  const int GT = mgx_obs_gt(NOV, dBase, dFlags);
  const bool want_mask = X && dMaskFeat != 0 && dNT > 0;
  const int xmode = mgx_obs_xmode(X, want_mask, S, d.P[MGX_H_NUM_TAGS]);
  const bool pack_mask = (xmode & MGX_OX_PACK) != 0, owner8 = (xmode & MGX_OX_OWNER8) != 0;
  const int TW = (xmode >> 4) & 15;  // tag words kept per agent for the territory mask
  const MgxObsLds L = mgx_obs_lds_layout(HW, NOFF, S, A, T, pool_tokens, xmode, NOV, PL ? blk_words : 0, GT,
                                         rewards_early != 0, EW);
  const int CP = L.cp, CS = L.cs;
  const bool pack_j = mgx_obs_pack_j(X, S, NOFF);
  uint16_t* s_grid = (uint16_t*)(smem + L.grid);
  char2* s_offs = (char2*)(smem + L.offs);
  uint8_t* s_loc = smem + L.loc;
  uint32_t* s_minobs = (uint32_t*)(smem + L.minobs);
  uint32_t* s_visited = (uint32_t*)(smem + L.visited);
  uint32_t* s_tokinfo = (uint32_t*)(smem + L.tokinfo);
  uint16_t* s_dyn = (uint16_t*)(smem + L.dyn);
  uint32_t* s_agents = (uint32_t*)(smem + L.agents);
  uint32_t* s_aginfo = (uint32_t*)(smem + L.aginfo);
  uint16_t* s_spawn = (uint16_t*)(smem + L.spawn);
  float* s_vstat = (float*)(smem + L.vstat);
  int* s_written = (int*)(smem + L.written);
  uint32_t* s_rwinfo = (uint32_t*)(smem + L.rwinfo);
  uint32_t* s_misc = (uint32_t*)(smem + L.misc);
  uint16_t* s_pool = (uint16_t*)(smem + L.pool);
  const int row = lane >> 4, rl = lane & 15;  // encode: one agent per 16-lane DPP row
  // One u32 per token: loc | f << 8 | v << 16.  The pitch is 8 words mod 16 (layout), so the rows of a wavefront start at
  // banks 0, 8 | 24, 16, 24 | 8: DPP rows 0 and 1 (one 32-lane LDS group) take the 1st and 3rd of them, rows 2 and 3 the 2nd
  // and 4th — the two 16-word spans of a group then never share a bank.
  uint32_t* s_row = (uint32_t*)(smem + L.rows) + (wave * 4 + (((row & 1) << 1) | (row >> 1))) * L.row_pitch;
  // s_row[TRASH]: this lane's word of the workgroup's trash block, as an index relative to its staging row
  const int TRASH = (int)(((uint32_t*)(smem + L.trash) + lane) - s_row);
  uint16_t* s_cell = (uint16_t*)(smem + L.cell);
  uint8_t* s_vj = smem + L.vj;
  uint32_t* s_vcount = (uint32_t*)(smem + L.vcount);
  uint32_t* s_gtok = (uint32_t*)(smem + L.gtok);

  typedef MgxEnvT<MgxGlobalProg, X> Env;
  Env e(d, d.P, env);
  e.step = d.step[env];
  const uint32_t step = e.step;
  const int sid_visited = d.wk[MGX_S_CELL_VISITED];
  uint16_t* s_owner = (uint16_t*)(smem + L.owner);
  uint8_t* s_vmask = smem + L.vmask;
  uint32_t* s_agtags = (uint32_t*)(smem + L.agtags);
  uint32_t* s_obsval = (uint32_t*)(smem + L.obsval);
  uint8_t* s_owner8 = smem + L.owner;
  // Program view of the interpreted sections: LDS copy (PL) or the blob itself.
  typedef typename std::conditional<PL, MgxLdsProg, MgxGlobalProg>::type VP;
  VP vp;
  if constexpr (PL) vp = (MgxLdsProg)(int32_t*)(smem + L.blk);
  else vp = d.P;
  MgxEnvT<VP, X> ev(d, vp, env);
  ev.step = step;
  ev.xl = e.xl;
  const MgxBase B((uint32_t)dBase);
  const int hr = d.feat[14], wr = d.feat[15];  // obs_height >> 1, obs_width >> 1 (stored by the host)

  // RewardHelper::compute_entries (reward.hpp:56-77) + truncation/termination (mettagrid_c.cpp:1086-1096) of one agent
  auto agent_rewards = [&](int a) {
    const int slot = s_agents[a] & 0xFFFF;
    const uint32_t rwi = s_rwinfo[a];
    const int nrw = (int)(rwi >> 16);
    VP rw = vp + d.sec[MGX_SEC_REWARDS] + (int)(rwi & 0xFFFF) * MGX_RW_WORDS;
    const float ep = d.episode_rewards[e.ao(a)];
    float total = 0.f;
    for (int k = 0; k < nrw; k++, rw += MGX_RW_WORDS) {
      float* prev = &d.ag_rprev[e.ao(a) * dNRW + k];
      const float pv = *prev;
      MgxCtx vc = mgx_ctx(slot, slot);
      float val = ev.template eval_code<0>(rw[MGX_RW_GV_START], rw[MGX_RW_GV_COUNT], slot, vc, 0);
      if (rw[MGX_RW_ACCUMULATE]) total = __fadd_rn(total, val);
      else total = __fadd_rn(total, __fsub_rn(val, pv));
      *prev = val;
    }
    float reward = total != 0.f ? total : 0.f;  // rewards were zeroed at the top of the step; += total
    d.rewards[e.ao(a)] = reward;
    d.episode_rewards[e.ao(a)] = __fadd_rn(ep, reward);
    if (dMaxSteps > 0 && step >= (uint32_t)dMaxSteps) {
      if (d.truncates) d.truncations[e.ao(a)] = 1;
      else d.terminals[e.ao(a)] = 1;
    }
  };

  MGX_TICK0();
  if constexpr (BOX) {
    // The boxes of the env's agents are 99 % zeros: all of them are streamed out NOW with 16-byte stores by every thread
    // of the workgroup, under the staging and classification phases; the encode phase only stores the few hundred values.
    const size_t env_bytes = (size_t)A * (size_t)(box_C * (2 * hr + 1) * (2 * wr + 1)) * (box_dtype == MGX_BOX_F32 ? 4 : 2);
    uint8_t* eb = (uint8_t*)box_out + (size_t)env * env_bytes;
    if ((env_bytes & 15) == 0 && (((uintptr_t)eb) & 15) == 0) {
      for (size_t i = tid; i < env_bytes / 16; i += NTH) ((uint4*)eb)[i] = make_uint4(0u, 0u, 0u, 0u);
    } else {
      for (size_t i = tid; i < env_bytes / 2; i += NTH) ((uint16_t*)eb)[i] = 0;
    }
  }
  // The classification of object slot `tid` (phase 0b) needs only global data: its loads are issued first, so they
  // are in flight together with the staging loads below instead of one barrier later.
  uint16_t pre_cls = MGX_DEAD_CLASS;
  uint32_t pre_vis = 0, pre_cinfo = 0;
  // the two game stats the tail adds to (nothing else in this kernel writes them): loaded now, so the last thing a
  // workgroup does is not an HBM round trip that keeps its LDS occupied
  float pre_tw = 0.f, pre_tf = 0.f;
  if (tid == 0) {
    const float* gs0 = d.game_stats + (size_t)env * d.NG;
    pre_tw = gs0[d.wk[MGX_S_GAME_TOKENS_WRITTEN]];
    pre_tf = gs0[d.wk[MGX_S_GAME_TOKENS_FREE]];
  }
  if (tid < S) {
    pre_cls = d.obj_cls[e.so(tid)];
    pre_vis = d.obj_visited[e.so(tid)];
    if (pre_cls != MGX_DEAD_CLASS) pre_cinfo = d.cls_tokinfo[pre_cls];
  }
  // ---- phase 0a: stage the env (coalesced) ----
  {
    const uint4* src = (const uint4*)(d.grid + (size_t)env * HW);
    uint4* dst = (uint4*)s_grid;
    const int n16 = (HW * 2) / 16;
    for (int i = tid; i < n16; i += NTH) dst[i] = src[i];
    for (int i = n16 * 8 + tid; i < HW; i += NTH) s_grid[i] = d.grid[(size_t)env * HW + i];
    if constexpr (PL) {
      const int4* bsrc = (const int4*)(d.P + blk_start);
      int4* bdst = (int4*)(smem + L.blk);
      for (int i = tid; i < blk_words / 4; i += NTH) bdst[i] = bsrc[i];
    }
    for (int i = tid; i < pool_prefix; i += NTH) s_pool[i] = d.cls_tok[i];
    const int32_t* offs = d.P + d.sec[MGX_SEC_OBS_OFFSETS];
    for (int i = tid; i < CP; i += NTH) {
      const int ii = min(i, NOFF - 1);
      const int orow = offs[ii * 2], ocol = offs[ii * 2 + 1];
      if (i < NOFF) s_offs[i] = make_char2((char)orow, (char)ocol);
      s_loc[i] = (uint8_t)(((orow + hr) << 4) | (ocol + wr));
    }
    for (int i = tid; i < A; i += NTH) {
      const uint32_t slot = d.ag_obj[e.ao(i)];
      const int32_t ex = d.executed[e.ao(i)];
      const uint16_t sprev = d.ag_stepprev[e.ao(i)];
      const uint16_t spawn = d.ag_spawn[e.ao(i)];
      const float vs = step > 0 ? e.astat_get(i, sid_visited) : 0.f;
      // the agent's cell and its class's reward records from the per-agent mirrors (MgxDev::ag_rc, ag_rwinfo): every load of
      // this loop is independent — one memory round trip in front of the first barrier instead of three (slot -> object row ->
      // class record)
      const uint16_t rc = d.ag_rc[e.ao(i)];
      const uint32_t rwi = d.ag_rwinfo[e.ao(i)];
      s_agents[i] = slot | ((uint32_t)rc << 16);
      s_aginfo[i] = ((uint32_t)ex & 0xFF) | (rc != sprev ? 0x100u : 0u);
      s_spawn[i] = spawn;
      s_vstat[i] = vs;
      s_rwinfo[i] = rwi;
      if constexpr (X) {
        if (want_mask) {   // the observer's own tag bitset (friend / foe of a cell's owner tag), once per env
          const int32_t* C = d.obj_tags ? nullptr : mgx_cls(d, d.obj_cls[e.so(slot)]);
          for (int w = 0; w < TW; w++)
            s_agtags[i * TW + w] = d.obj_tags ? d.obj_tags[e.so(slot) * MGX_TAG_WORDS + w] : (uint32_t)C[MGX_C_TAGS + w];
        }
      }
    }
    if constexpr (X) {
      // The territory owner of every cell (TerritoryTracker::compute_observability_at :254-273: the first territory type
      // with an owner decides; mgx_terr_kernel keeps the per-type maps current) and the query-backed obs values are plain
      // per-env data: staged here with the grid, one round trip, instead of behind a barrier of their own in the middle
      // of the kernel.
      if (want_mask)
        for (int cellidx = tid; cellidx < HW; cellidx += NTH) {
          uint16_t owner = 0xFFFF;
          for (int ti = 0; ti < dNT && owner == 0xFFFF; ti++) owner = d.terr_owner[((size_t)env * dNT + ti) * (size_t)HW + cellidx];
          if (owner8) s_owner8[cellidx] = (uint8_t)owner; else s_owner[cellidx] = owner;  // (0xFFFF -> 0xFF: tag ids stop at 254 then)
        }
      if (d.obsval)  // evaluated by mgx_values_kernel (one env per lane) right before this kernel
        for (int i = tid; i < A * NOV; i += NTH) s_obsval[i] = d.obsval[(size_t)env * A * NOV + i];
    }
    if (tid == 0) { s_misc[0] = (uint32_t)pool_prefix; if (S > NTH) s_misc[1] = 0; }  // pool top, number of per-step lists (S <= NTH: a count byte per wavefront, below)
  }
  const bool dyn_tags = X && d.obj_tags != nullptr;
  // ---- phase 0b: classify the object slots.  Static classes show their class tag list (pool prefix); everything
  // else is queued for the list builder.  When every slot has a thread of its own (S <= NTH) the classification needs
  // nothing another thread staged — its inputs are this thread's loads from the top of the kernel — so it runs IN FRONT of
  // the staging barrier and the barrier behind it is gone: a wavefront queues its slots in its own segment of the queue
  // (ballot rank, no shared counter) and leaves its count in a byte of s_misc[1..2]. ----
  const bool early_cls = S <= NTH;
  auto classify = [&](int s, uint16_t cls, uint32_t vis, uint32_t cinfo, bool& queue) -> uint32_t {
    s_minobs[s] = 0xFFFFFFFFu;
    uint32_t info = 0;
    queue = false;
    if (cls != MGX_DEAD_CLASS) {  // cinfo: start(16) | group(8) | ntags(6) | agent(1) | static(1)
      s_visited[s] = vis;
      if ((cinfo >> 31) != 0) info = (cinfo & 0xFFFFu) | (((cinfo >> 24) & 0x3Fu) << 16);  // static class: nothing about it changes, tags included (MGX_C_STATIC, compiler.py)
      else { queue = true; info = cinfo; }  // parked here for the builder, which replaces it with start | count << 16
    }
    return info;
  };
  if (early_cls) {
    bool queue = false;
    if (tid < S) s_tokinfo[tid] = classify(tid, pre_cls, pre_vis, pre_cinfo, queue);
    const unsigned long long qm = __ballot(queue);
    if (queue) s_dyn[wave * MGX_WAVE + __popcll(qm & ((1ull << lane) - 1ull))] = (uint16_t)tid;
    if (lane == 0) ((uint8_t*)&s_misc[1])[wave] = (uint8_t)__popcll(qm);
  }
  if constexpr (BOX) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's zero stores are out: no value stored
                                                                         // behind a barrier can be overtaken by one of them
  __syncthreads();
  MGX_TICK(8);
  MGX_PHASE_END(1);
  if (!early_cls) {
    for (int s = tid; s < S; s += NTH) {
      const size_t o = e.so(s);
      const bool pre = s == tid;  // first pass: loaded above
      const uint16_t cls = pre ? pre_cls : d.obj_cls[o];
      const uint32_t vis = pre ? pre_vis : d.obj_visited[o];
      const uint32_t cinfo = cls == MGX_DEAD_CLASS ? 0u : pre ? pre_cinfo : d.cls_tokinfo[cls];
      bool queue = false;
      const uint32_t info = classify(s, cls, vis, cinfo, queue);
      if (queue) s_dyn[atomicAdd(&s_misc[1], 1u)] = (uint16_t)s;
      s_tokinfo[s] = info;
    }
    __syncthreads();
  }

  // ---- phase 0c: per-step token lists, one thread per queued object (a fraction of one wavefront in most envs).
  // The wavefronts that get no objects ("free") take the whole window-map phase and the global tokens meanwhile, so
  // the workgroup's critical path is the longer of the two jobs, not their sum. ----
  int ndyn = (int)s_misc[1];
  uint32_t seg_lo = 0, seg_hi = 0;   // early_cls: the wavefronts' queue lengths, one byte each
  if (early_cls) {
    seg_lo = s_misc[1]; seg_hi = NTH > 4 * MGX_WAVE ? s_misc[2] : 0u;
    ndyn = 0;
    for (int w = 0; w < NTH / MGX_WAVE; w++) ndyn += (int)(((w < 4 ? seg_lo : seg_hi) >> (8 * (w & 3))) & 0xFFu);
  }
  const int nbw = min((ndyn + MGX_WAVE - 1) / MGX_WAVE, (NTH / MGX_WAVE));  // wavefronts with list-building work
  const int nfree = (NTH / MGX_WAVE) - nbw;
  {
    const int f_vibe = d.feat[MGX_F_VIBE], f_group = d.feat[MGX_F_GROUP], f_agent = d.feat[MGX_F_AGENT_ID], f_tag = d.feat[MGX_F_TAG];
    VP feat = vp + d.sec[MGX_SEC_INV_FEATURES];
    for (int i = tid; i < ndyn; i += NTH) {
      int qi = i;
      if (early_cls) {   // i-th queued slot -> (wavefront segment, rank)
        int w = 0, r = i;
        for (; w < NTH / MGX_WAVE - 1; w++) {
          const int c = (int)(((w < 4 ? seg_lo : seg_hi) >> (8 * (w & 3))) & 0xFFu);
          if (r < c) break;
          r -= c;
        }
        qi = w * MGX_WAVE + r;
      }
      const int s = s_dyn[qi];
      const uint32_t cinfo = s_tokinfo[s];
      // every field of the slot at once: independent loads, one memory round trip
      const size_t o = e.so(s);
      const uint32_t vibe = d.obj_vibe[o];
      const uint32_t agent_id = d.obj_agent[o];
      unsigned long long ord = d.obj_order[o];
      if constexpr (X) {
        if (d.obj_flags && (d.obj_flags[o] & 2)) ord = ~0ull;  // created at run time without an ObservationEncoder
      }
      const bool is_static = (cinfo >> 31) != 0;
      const bool is_agent = (cinfo & 0x40000000u) != 0;
      // Inventory amounts in iteration order.  Extended kernels: the whole 32-byte row is loaded with the fields above (the
      // same round trip; picking the items out by their order word first made it a second one: rung 4 3.55 -> 3.35 ms) and an
      // item's amount is picked out of the eight row words when its tokens are written.  Lean kernels keep the second round
      // trip of per-item loads: the picks cost 15 VGPRs and 27 spilled SGPRs there (rung 3: 0.52 -> 0.59 ms).
      static_assert(MGX_INV_PITCH == 16, "inventory rows are two 16-byte words");
      uint4 r0 = make_uint4(0u, 0u, 0u, 0u), r1 = r0;
      if constexpr (X) { r0 = ((const uint4*)(d.obj_inv + o * MGX_INV_PITCH))[0]; r1 = ((const uint4*)(d.obj_inv + o * MGX_INV_PITCH))[1]; }
      uint32_t amt[MGX_MAX_ITEMS];
      uint32_t live_mask = 0;
      {
        bool live = !is_static;
#pragma unroll
        for (int k = 0; k < MGX_MAX_ITEMS; k++) {
          const int item = (int)((ord >> (4 * k)) & 0xF);
          live = live && item != 0xF;
          if constexpr (!X) amt[k] = live ? (uint32_t)d.obj_inv[o * MGX_INV_PITCH + item] : 0u;
          if (live) live_mask |= 1u << k;
        }
      }
      auto amount_of = [&](int k, int item) -> uint32_t {
        if constexpr (!X) return amt[k];
        const uint32_t lo = (item & 4) ? ((item & 2) ? r0.w : r0.z) : ((item & 2) ? r0.y : r0.x);
        const uint32_t hi = (item & 4) ? ((item & 2) ? r1.w : r1.z) : ((item & 2) ? r1.y : r1.x);
        const uint32_t w = (item & 8) ? hi : lo;
        return (item & 1) ? (w >> 16) : (w & 0xFFFFu);
      };
      int ntags = (cinfo >> 24) & 0x3F;
      uint32_t tagw[MGX_TAG_WORDS];
      if (dyn_tags && !is_static) {
        ntags = 0;
#pragma unroll
        for (int w = 0; w < MGX_TAG_WORDS; w++) { tagw[w] = d.obj_tags[o * MGX_TAG_WORDS + w]; ntags += __popc(tagw[w]); }
      }
      // worst-case length reserves the pool space (the host sized the pool for exactly this bound); the real length
      // is known once the list is written
      int n = ntags;
      if (!is_static) n += 1 + __popc(live_mask) * B.digits(65535u) + (is_agent ? 2 : 0);
      uint32_t info = 0;
      const uint32_t start = atomicAdd(&s_misc[0], (uint32_t)n);
      if ((int)(start + n) <= pool_tokens) {
        MgxObjTok w{s_pool, (int)start};
        if (dyn_tags && !is_static) {  // ascending tag id (core/grid_object.cpp:181-186)
#pragma unroll
          for (int wd = 0; wd < MGX_TAG_WORDS; wd++) {
            uint32_t m = tagw[wd];
            while (m) { int b = __ffs(m) - 1; m &= m - 1; w.put(f_tag, wd * 32 + b); }
          }
        } else {
          const uint16_t* src = s_pool + (cinfo & 0xFFFF);  // class tag list from the pool prefix
          for (int k = 0; k < ntags; k++) s_pool[w.pos++] = src[k];
        }
        if (!is_static) {
          if (vibe != 0) w.put(f_vibe, vibe);
#pragma unroll
          for (int k = 0; k < MGX_MAX_ITEMS; k++) {  // observation_encoder.hpp:198-225: base digit, then :pK digits
            if (live_mask & (1u << k)) {
              const int item = (int)((ord >> (4 * k)) & 0xF);
              VP F = feat + item * MGX_IF_WORDS;
              uint32_t rem = amount_of(k, item);
              w.put((uint32_t)F[0], B.lo(rem));
              rem = B.hi(rem);
              for (int pdig = 1; rem > 0; pdig++) { w.put((uint32_t)F[pdig], B.lo(rem)); rem = B.hi(rem); }
            }
          }
          if (is_agent) {
            w.put(f_group, (cinfo >> 16) & 0xFF);
            w.put(f_agent, agent_id);
          }
        }
        info = start | ((uint32_t)(w.pos - (int)start) << 16);
      } else {
        d.err[env] |= 16u;  // token pool exhausted (sized by the host from the class maps)
      }
      s_tokinfo[s] = info;
    }
  }
  // Rewards that read nothing this kernel writes (no stat operands; host flag) are evaluated here by the last
  // wavefront, which has little or no list-building work, instead of serially at the end of the workgroup.
  if (WITH_REWARDS && rewards_early && wave == (NTH / MGX_WAVE) - 1) {
    // One lane per (agent, reward entry): the entries of an agent are evaluated side by side (their loads in flight
    // together) and then summed in entry order by the agent's first lane, exactly like the serial loop of
    // agent_rewards (reward.hpp:56-77: total += val or val - prev, entry by entry).
    const int NRW = dNRW;
    if (NRW <= MGX_WAVE) {
      const int per = MGX_WAVE / NRW;  // agents per pass
      for (int a0 = 0; a0 < A; a0 += per) {
        const int a = a0 + lane / NRW, k = lane % NRW;
        const bool have = lane < per * NRW && a < A;
        float contrib = 0.f, ep = 0.f;
        int nrw = 0;
        if (have) {
          if (k == 0) ep = d.episode_rewards[e.ao(a)];
          const uint32_t rwi = s_rwinfo[a];
          nrw = (int)(rwi >> 16);
          if (k < nrw) {
            const int slot = s_agents[a] & 0xFFFF;
            VP rw = vp + d.sec[MGX_SEC_REWARDS] + ((int)(rwi & 0xFFFF) + k) * MGX_RW_WORDS;
            float* prev = &d.ag_rprev[e.ao(a) * NRW + k];
            const float pv = *prev;
            MgxCtx vc = mgx_ctx(slot, slot);
            const float val = ev.template eval_code<0>(rw[MGX_RW_GV_START], rw[MGX_RW_GV_COUNT], slot, vc, 0);
            contrib = rw[MGX_RW_ACCUMULATE] ? val : __fsub_rn(val, pv);
            *prev = val;
          }
        }
        float total = 0.f;
        for (int j = 0; j < NRW; j++) {  // uniform trip count; lanes past their agent's entry count add nothing
          const float cj = __shfl(contrib, lane + j);
          if (j < nrw) total = __fadd_rn(total, cj);
        }
        if (have && k == 0) {
          const float reward = total != 0.f ? total : 0.f;  // rewards were zeroed at the top of the step; += total
          d.rewards[e.ao(a)] = reward;
          d.episode_rewards[e.ao(a)] = __fadd_rn(ep, reward);
          if (dMaxSteps > 0 && step >= (uint32_t)dMaxSteps) {
            if (d.truncates) d.truncations[e.ao(a)] = 1;
            else d.terminals[e.ao(a)] = 1;
          }
        }
      }
    } else {
      for (int a = lane; a < A; a += MGX_WAVE) agent_rewards(a);
    }
  }
  // ---- phase 1: per agent the list of window cells that will emit tokens, in window order (ballot compaction), and
  // the first observer (lowest agent index) of every object.  The encode loop then only walks real entries. ----
  // (the wavefront that evaluates the early rewards is left out when another free one exists: its job is as long as
  // half of this phase)
  const int np1 = nfree - ((WITH_REWARDS && rewards_early && nfree >= 2) ? 1 : 0);  // free wavefronts taking agents
  // a lane's window offsets of the first two passes are the same for every agent: read once, not once per agent
  const char2 offs_p0 = s_offs[min(lane, NOFF - 1)], offs_p1 = s_offs[min(lane + MGX_WAVE, NOFF - 1)];
  const int a_first = nfree > 0 ? (wave >= nbw && wave < nbw + np1 ? wave - nbw : A) : wave;
  const int a_step = nfree > 0 ? np1 : (NTH / MGX_WAVE);
  // A window of 65..96 cells (11x11: 89) leaves the second pass of an agent at most half full: two agents share it,
  // one per 32-lane half, so a pair costs three passes instead of four.
  bool pair_tail = false;
  if constexpr (!X) pair_tail = NOFF > MGX_WAVE && NOFF <= MGX_WAVE + 32;
  if (pair_tail) {
    if constexpr (!X) {
      const int hl = lane & 31, half = lane >> 5;
      const int jt = MGX_WAVE + hl;
      const char2 offs_t = s_offs[min(jt, NOFF - 1)];
      // one window cell of agent x: its grid entry (0 = nothing to emit) + first-observer bookkeeping
      auto probe = [&](int x, char2 o, bool ok) -> uint32_t {
        const uint32_t ag = s_agents[x];
        const int r = (int)((ag >> 24) & 0xFF) + o.x, c = (int)((ag >> 16) & 0xFF) + o.y;
        const bool inb = ok && (unsigned)r < (unsigned)dH && (unsigned)c < (unsigned)dW;
        const uint32_t cs = inb ? (uint32_t)s_grid[inb ? __mul24(r, dW) + c : 0] : 0u;
        if (step > 0 && cs) atomicMin(&s_minobs[cs - 1], (uint32_t)x);
        return cs;
      };
      auto full_pass = [&](int x) -> int {  // cells 0..63 of agent x
        const uint32_t cs = probe(x, offs_p0, true);
        const unsigned long long m = __ballot(cs != 0);
        if (cs) {
          const int k = __popcll(m & ((1ull << lane) - 1ull));
          s_cell[x * CS + k] = (uint16_t)(pack_j ? cs | ((uint32_t)lane << 9) : cs);
          if (!pack_j) s_vj[x * CP + k] = (uint8_t)lane;
        }
        return __popcll(m);
      };
      for (int a = a_first; a < A; a += 2 * a_step) {
        const int b = a + a_step;
        const bool hb = b < A;
        int ca = full_pass(a);
        int cb = hb ? full_pass(b) : 0;
        const int x = (half && hb) ? b : a;
        const uint32_t cs = probe(x, offs_t, (half == 0 || hb) && jt < NOFF);
        const unsigned long long m = __ballot(cs != 0);
        const uint32_t mh = half ? (uint32_t)(m >> 32) : (uint32_t)m;
        if (cs) {
          const int k = (half ? cb : ca) + __popc(mh & ((1u << hl) - 1u));
          s_cell[x * CS + k] = (uint16_t)(pack_j ? cs | ((uint32_t)jt << 9) : cs);
          if (!pack_j) s_vj[x * CP + k] = (uint8_t)jt;
        }
        ca += __popc((uint32_t)m);
        cb += __popc((uint32_t)(m >> 32));
        if (lane == 0) {
          s_vcount[a] = (uint32_t)ca;
          if (hb) s_vcount[b] = (uint32_t)cb;
        }
      }
    }
  } else {
    for (int a = a_first; a < A; a += a_step) {
      const uint32_t ag = s_agents[a];
      const int r0 = (ag >> 24) & 0xFF, c0 = (ag >> 16) & 0xFF;
      int count = 0;
      for (int j = lane; j < ((NOFF + MGX_WAVE - 1) & ~(MGX_WAVE - 1)); j += MGX_WAVE) {  // whole wavefront in every pass (ballots)
        const char2 o = j == lane ? offs_p0 : j == lane + MGX_WAVE ? offs_p1 : s_offs[min(j, NOFF - 1)];
        const int r = r0 + o.x, c = c0 + o.y;
        const bool inb = j < NOFF && (unsigned)r < (unsigned)dH && (unsigned)c < (unsigned)dW;
        const uint32_t cs = inb ? (uint32_t)s_grid[inb ? __mul24(r, dW) + c : 0] : 0u;  // map <= 255 x 255
        bool keep = cs != 0;
        uint32_t mv = 0;  // _emit_tile_observability_tokens (:337-362): 1 = the cell's owner tag is one of mine, 2 = not
        if constexpr (X) {
          if (want_mask && inb) {
            const uint32_t ow = owner8 ? (uint32_t)s_owner8[r * dW + c] : (uint32_t)s_owner[r * dW + c];
            if (ow != (owner8 ? 0xFFu : 0xFFFFu)) { mv = ((s_agtags[a * TW + (ow >> 5)] >> (ow & 31)) & 1u) ? 1u : 2u; keep = true; }  // mask-only cells still emit one token
          }
        }
        if (step > 0 && cs) atomicMin(&s_minobs[cs - 1], (uint32_t)a);
        const unsigned long long m = __ballot(keep);
        if (keep) {
          const int k = count + __popcll(m & ((1ull << lane) - 1ull));
          s_cell[a * CS + k] = (uint16_t)(pack_mask ? cs | (mv << 14) : pack_j ? cs | ((uint32_t)j << 9) : cs);
          if (!pack_j) s_vj[a * CP + k] = (uint8_t)j;
          if constexpr (X) { if (want_mask && !pack_mask) s_vmask[a * CP + k] = (uint8_t)mv; }
        }
        count += __popcll(m);
      }
      if (lane == 0) s_vcount[a] = (uint32_t)count;
    }
  }
  // global tokens (location 0xFE), mettagrid_c.cpp:700-753: one thread per agent, all agents at once
  const int gwave = np1 < nfree ? (NTH / MGX_WAVE) - 1 : (nfree > 0 ? nbw : 0);  // after the early rewards when that wavefront skipped the lists
  for (int a = (wave == gwave) ? lane : A; a < A; a += MGX_WAVE) {
    const uint32_t ag = s_agents[a];
    const int my_slot = ag & 0xFFFF;
    const int r0 = (ag >> 24) & 0xFF, c0 = (ag >> 16) & 0xFF;
    uint32_t* g = s_gtok + a * GT;
    int pos = 0;
    auto put = [&](int f, uint32_t v) { g[pos++] = 0xFEu | ((uint32_t)(f & 0xFF) << 8) | ((v & 0xFF) << 16); };
    const uint32_t info = s_aginfo[a];
    if (dFlags & MGX_G_COMPLETION) {
      uint32_t pct = 0;
      if (dMaxSteps > 0) pct = step >= (uint32_t)dMaxSteps ? 255u : (256u * step / (uint32_t)dMaxSteps);
      put(d.feat[MGX_F_COMPLETION], pct);
    }
    if (dFlags & MGX_G_LAST_ACTION) put(d.feat[MGX_F_LAST_ACTION], info & 0xFF);
    if ((dFlags & MGX_G_LAST_ACTION_MOVE) && d.feat[MGX_F_LAST_ACTION_MOVE] != 0)
      put(d.feat[MGX_F_LAST_ACTION_MOVE], (info >> 8) & 1);
    // last_reward: the reference reads the reward buffer it zeroed at the top of the step (:937-938,722-726),
    // so the token is always round(0 * 100) = 0.
    if (dFlags & MGX_G_LAST_REWARD) put(d.feat[MGX_F_LAST_REWARD], 0);
    if (dFlags & MGX_G_LOCAL_POSITION) {
      uint16_t sp = s_spawn[a];
      int dc = c0 - (int)(sp & 0xFF), dr = (int)(sp >> 8) - r0;
      if (dc > 0) put(d.feat[MGX_F_LP_EAST], (uint32_t)min(dc, 255));
      else if (dc < 0) put(d.feat[MGX_F_LP_WEST], (uint32_t)min(-dc, 255));
      if (dr > 0) put(d.feat[MGX_F_LP_NORTH], (uint32_t)min(dr, 255));
      else if (dr < 0) put(d.feat[MGX_F_LP_SOUTH], (uint32_t)min(-dr, 255));
    }
    for (int i = 0; i < NOV; i++) {  // _emit_obs_value_tokens :1207-1238
      VP V = vp + d.sec[MGX_SEC_OBS_VALUES] + i * MGX_OV_WORDS;
      uint32_t rem;
      bool ext = false;
      if constexpr (X) ext = d.obsval != nullptr;
      if (ext) {
        rem = s_obsval[a * NOV + i];
      } else {
        MgxCtx vc = mgx_ctx(my_slot, my_slot);
        rem = (uint32_t)ev.template eval_code<0>(V[MGX_OV_GV_START], V[MGX_OV_GV_COUNT], my_slot, vc, 0);
      }
      int f = V[MGX_OV_FEATURE];
      put(f, B.lo(rem));
      rem = B.hi(rem);
      f++;
      while (rem > 0) { put(f, B.lo(rem)); rem = B.hi(rem); f++; }
    }
    s_aginfo[a] = (uint32_t)pos;  // from here on: the agent's global token count
  }
  __syncthreads();
  MGX_TICK(10);
  MGX_PHASE_END(3);

  // ---- phase 2: encode.  One agent per 16-lane DPP row, four agents per wavefront at a time: an agent sees ~15-20
  // occupied cells, so a whole wavefront per agent left three quarters of the lanes idle.  Scans are row scans (four
  // DPP steps), row-wide values travel by ds_bpermute, long token lists are copied by the 16 lanes of their row. ----
  if (WITH_REWARDS && rewards_mid && wave == (NTH / MGX_WAVE) - 1)
    for (int a = lane; a < A; a += MGX_WAVE) agent_rewards(a);
  // GridObject::visited stamps (:789-796): first observers are final since the window lists, and the encode only reads the
  // LDS copies — with wavefronts to spare (EW < NTH / 64) they write the stamps now instead of everybody behind the encode
  constexpr bool stamps_mid = EW < NTH / MGX_WAVE;
  if (stamps_mid && step > 0 && wave >= EW)
    for (int s = (wave - EW) * MGX_WAVE + lane; s < S; s += (NTH / MGX_WAVE - EW) * MGX_WAVE)
      if (s_minobs[s] != 0xFFFFFFFFu && s_visited[s] < step) d.obj_visited[e.so(s)] = step;
  for (int a0 = wave < EW ? wave * 4 : A; a0 < A; a0 += EW * 4) {
    const int a = a0 + row;
    const bool av = a < A;          // this row has an agent
    const int ac = av ? a : A - 1;  // clamped for the unconditional reads
    const uint32_t ag = s_agents[ac];
    const int my_slot = ag & 0xFFFF;
    for (int i = rl; i < L.row_words / 4; i += 16) ((uint4*)s_row)[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
    // global tokens were assembled once per env above; copy this agent's
    const int n_global = av ? (int)s_aginfo[ac] : 0;
    for (int k = rl; k < n_global && k < T; k += 16) s_row[k] = s_gtok[ac * GT + k];
    int base_pos = n_global;
    float visited_acc = s_vstat[ac];
    bool visited_any = false;
    const int nvis = av ? (int)s_vcount[ac] : 0;
    const int nmax = max(max(__builtin_amdgcn_readlane(nvis, 0), __builtin_amdgcn_readlane(nvis, 16)),
                         max(__builtin_amdgcn_readlane(nvis, 32), __builtin_amdgcn_readlane(nvis, 48)));
    // the agent's visible cells in reference (window) order, 16 per pass; everything is LDS and every read is
    // unconditional (lanes past the list read entry 0 and are masked by a select)
    for (int k0 = 0; k0 < nmax; k0 += 16) {
      const int kk = k0 + rl;
      const bool valid = kk < nvis;
      uint32_t cs = valid ? (uint32_t)s_cell[ac * CS + kk] : 0u;
      uint32_t pmask = 0;
      if constexpr (X) { if (pack_mask) { pmask = cs >> 14; cs &= 0x3FFFu; } }
      int j;
      if (pack_j) { j = (int)(cs >> 9); cs &= 0x1FFu; }
      else j = s_vj[ac * CP + (valid ? kk : 0)];
      const bool has = cs != 0;
      const int slot = has ? (int)cs - 1 : 0;
      const uint32_t info = s_tokinfo[slot];
      const uint32_t loc = s_loc[j];
      const int start = info & 0xFFFF;
      int n = has ? (int)(info >> 16) : 0;
      uint32_t stale = 0, mask = 0;
      if (step > 0) {  // uniform
        const uint32_t mo = s_minobs[slot], pv = s_visited[slot];
        stale = (has && mo == (uint32_t)a && pv < step) ? step - pv : 0u;
      }
      if constexpr (X) {
        if (want_mask) {  // the territory token comes before the cell's object tokens (:337-362); decided in phase 1
          mask = pack_mask ? pmask : valid ? (uint32_t)s_vmask[ac * CP + kk] : 0u;
          if (mask) n += 1;
        }
      }
      const int incl = mgx_row_incl_scan(n);
      const int tot = __shfl(incl, lane | 15);
      int pos = base_pos + incl - n;
      if constexpr (X) {
        if (mask) { if (pos < T) s_row[pos] = loc | ((uint32_t)dMaskFeat << 8) | (mask << 16); pos++; n--; }
      }
      // Token lists of up to MGX_SMALL_LIST entries (walls, plain objects) are copied by the cell's own lane; longer
      // ones (agents carrying an inventory: the observer itself is always one) by the 16 lanes of the row, one list
      // at a time, so the copy loop's trip count is not set by the longest list in the window.
#define MGX_SMALL_LIST 3
      {
        const bool big = n > MGX_SMALL_LIST;
        const int ns = big ? 0 : n;
#pragma unroll
        for (int k = 0; k < MGX_SMALL_LIST; k++) {  // pool has 8 spare entries: the read is always in range
          const uint32_t tok = loc | ((uint32_t)s_pool[start + k] << 8);
          s_row[(k < ns && pos + k < T) ? pos + k : TRASH] = tok;
        }
        uint32_t rm = (uint32_t)(__ballot(big) >> (row * 16)) & 0xFFFFu;  // this row's long lists
        while (__ballot(rm != 0)) {
          const bool mine = rm != 0;
          const int src = row * 16 + (mine ? __ffs(rm) - 1 : 0);
          rm &= rm - 1;
          const int bp = __shfl(pos, src), bs = __shfl(start, src), bn = mine ? __shfl(n, src) : 0;
          const uint32_t bl = (uint32_t)__shfl((int)loc, src);
          for (int k = rl; k < bn; k += 16)
            if (bp + k < T) s_row[bp + k] = bl | ((uint32_t)s_pool[bs + k] << 8);
        }
      }
      base_pos += tot;
      // cell.visited staleness (:789-796): the reference adds the values one by one in cell order.  All of them are
      // integers, so while the running sum is an integer below 2^24 every partial sum is exact and the order does not
      // matter: one row sum.  Otherwise replay the serial order.
      uint32_t fm = (uint32_t)(__ballot(stale != 0) >> (row * 16)) & 0xFFFFu;
      if (__ballot(fm != 0)) {
        const uint32_t ssum = (uint32_t)__shfl(mgx_row_incl_scan((int)stale), lane | 15);
        if (fm != 0) {
          visited_any = true;
          if (visited_acc == truncf(visited_acc) && visited_acc >= 0.f && visited_acc + (float)ssum <= 16777216.f && ssum < 16777216u) {
            visited_acc += (float)ssum;
            fm = 0;
          }
        }
        while (__ballot(fm != 0)) {  // serial replay for the rows that could not take the exact shortcut
          const bool mine = fm != 0;
          const int src = row * 16 + (mine ? __ffs(fm) - 1 : 0);
          fm &= fm - 1;
          const uint32_t sv = (uint32_t)__shfl((int)stale, src);
          if (mine) visited_acc = __fadd_rn(visited_acc, (float)sv);
        }
      }
    }
    if (rl == 0 && av) {
      // every addend is >= 1, so the value is non-zero and "key exists" follows from it (see astat_add): a plain
      // store, not a store plus a read-modify-write of the touched word that stalls the row for a round trip
      if (visited_any && sid_visited >= 0) d.ag_stats[e.ao(a) * d.NSP + sid_visited] = visited_acc;
      s_written[a] = base_pos;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // LDS row complete before it is read back
    if constexpr (BOX) {
      // ---- the agent's dense box from its staging row: the cell values are stored over the zeros the whole workgroup
      // streamed out at the start of the kernel (waited for in front of the first barrier).  Tokens that share a cell are
      // adjacent in a row this kernel built (the tag tokens of one object) — the first lane of such a run adds the rest in
      // token order — except a global token and a same-feature token of the centre object (below). ----
      const bool f32 = box_dtype == MGX_BOX_F32;
      const int Hh = 2 * hr + 1, Ww = 2 * wr + 1, cells = box_C * Hh * Ww;
      const int bytes = cells * (f32 ? 4 : 2);
      uint8_t* bx = (uint8_t*)box_out + ((size_t)env * A + ac) * (size_t)bytes;
      auto key_of = [&](uint32_t tok) -> uint32_t {        // (cell, feature) of a token; 0xFFFFFFFF = padding
        const uint32_t lc = tok & 0xFFu;
        return lc == 0xFFu ? 0xFFFFFFFFu : ((lc == 0xFEu ? (uint32_t)((hr << 4) | wr) : lc) | (tok & 0xFF00u));
      };
      for (int k = rl; k < T && av; k += 16) {
        const uint32_t tok = s_row[k];
        const uint32_t key = key_of(tok);
        if (key == 0xFFFFFFFFu) continue;
        if (k > 0 && key_of(s_row[k - 1]) == key) continue;   // a later token of a run: its first lane adds it
        const int f = (int)((key >> 8) & 0xFFu), y = (int)((key >> 4) & 0xFu), x = (int)(key & 0xFu);
        const bool global_tok = (tok & 0xFFu) == 0xFEu;
        // The one way two tokens of a key are NOT neighbours: a global token (location 0xFE, counted on the centre cell; the
        // globals lead the row) and a token of the object ON the centre cell with the same feature id.  np.add.at
        // (grid_obs_wrapper.py:57-95) adds them in token order, so the global token's lane adds every later match and
        // a centre-cell token that finds its key among the leading globals leaves the cell to that lane.
        if (!global_tok && (key & 0xFFu) == (uint32_t)((hr << 4) | wr)) {
          bool taken = false;
          for (int j = 0; j < k && (s_row[j] & 0xFFu) == 0xFEu; j++) taken = taken || key_of(s_row[j]) == key;
          if (taken) continue;
        }
        const float sc = box_scale[f];
        float sum = __fdiv_rn((float)((tok >> 16) & 0xFFu), sc);
        if (global_tok) {
          for (int j = k + 1; j < T; j++) if (key_of(s_row[j]) == key) sum = __fadd_rn(sum, __fdiv_rn((float)((s_row[j] >> 16) & 0xFFu), sc));
        } else {
          for (int j = k + 1; j < T && key_of(s_row[j]) == key; j++) sum = __fadd_rn(sum, __fdiv_rn((float)((s_row[j] >> 16) & 0xFFu), sc));
        }
        if (y < Hh && x < Ww && f < box_C) {
          const int cellidx = (f * Hh + y) * Ww + x;
          if (f32) {
            ((float*)bx)[cellidx] = sum;
          } else {   // bfloat16, round to nearest even (the sums are finite and non-negative)
            const uint32_t u = __float_as_uint(sum);
            ((uint16_t*)bx)[cellidx] = (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
          }
        }
      }
    } else {
    // ---- pack 4 tokens (4 x u32, low 3 bytes valid) into 12 bytes and store them: the row's 16 lanes sweep it ----
    uint8_t* out = d.obs + ((size_t)env * A + ac) * (size_t)T * 3;
    for (int q = rl; q * 4 < T && av; q += 16) {
      uint4 t = ((const uint4*)s_row)[q];
      uint32_t w0 = (t.x & 0xFFFFFFu) | (t.y << 24);
      uint32_t w1 = ((t.y >> 8) & 0xFFFFu) | (t.z << 16);
      uint32_t w2 = ((t.z >> 16) & 0xFFu) | (t.w << 8);
      if (q * 4 + 4 <= T && ((3 * T) & 3) == 0) {
        uint32_t* o32 = (uint32_t*)(out + q * 12);
        o32[0] = w0; o32[1] = w1; o32[2] = w2;
      } else {  // ragged tail or unaligned row pitch
        uint32_t w[3] = {w0, w1, w2};
        for (int b = 0; b < 12 && q * 12 + b < 3 * T; b++) out[q * 12 + b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
      }
    }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  }
  MGX_TICK(11);
  __syncthreads();
  MGX_TICK(12);
  MGX_PHASE_END(4);

  // ---- visited stamps, token statistics, rewards, termination ----
  if (!stamps_mid && step > 0)
    for (int s = tid; s < S; s += NTH)
      if (s_minobs[s] != 0xFFFFFFFFu && s_visited[s] < step) d.obj_visited[e.so(s)] = step;
  if (wave == 0) {  // tokens_written / tokens_free_space (:659-661, 821-823): the reference adds agent by agent in f32
    float* gs = d.game_stats + (size_t)env * d.NG;
    float tw = pre_tw, tf = pre_tf;  // lane 0 (= thread 0) holds the values
    // All addends are integers: while both running sums stay integers below 2^24 every partial sum is exact and the
    // order does not matter -> one wavefront sum.  Otherwise (or on a token overflow) replay the serial order.
    const int nw = lane < A ? s_written[lane] : 0;
    const bool over = __ballot(lane < A && nw > T) != 0;
    const uint32_t sum_w = mgx_wave_sum((uint32_t)nw);
    if (lane == 0) {
      // passes: an episode restart stands for the reference's new MettaGrid + set_buffers, which computes the initial
      // observations TWICE (ctor -> _make_buffers -> set_buffers, then the caller's set_buffers: mettagrid_c.cpp:190, 271-292,
      // 1165-1184) and so counts their tokens twice; a step counts once.
      const uint32_t passes = WITH_REWARDS ? 1u : (uint32_t)stat_passes;
      const uint32_t sum_f = (uint32_t)A * (uint32_t)T - sum_w;
      const bool exact = A <= MGX_WAVE && !over && tw == truncf(tw) && tf == truncf(tf) && tw >= 0.f && tf >= 0.f &&
                         tw + (float)(sum_w * passes) <= 16777216.f && tf + (float)(sum_f * passes) <= 16777216.f;
      bool overflow = false;
      if (exact) {
        tw += (float)(sum_w * passes);
        tf += (float)(sum_f * passes);
      } else {
        for (uint32_t p = 0; p < passes && !overflow; p++)
          for (int a = 0; a < A && !overflow; a++) {
            int n = s_written[a];
            if (n > T) { overflow = true; break; }  // reference: std::runtime_error (:813-819)
            tw = __fadd_rn(tw, (float)n);
            tf = __fadd_rn(tf, (float)(T - n));
          }
      }
      gs[d.wk[MGX_S_GAME_TOKENS_WRITTEN]] = tw;
      gs[d.wk[MGX_S_GAME_TOKENS_FREE]] = tf;
      if (overflow) d.err[env] |= 1u;
    }
  }
  MGX_TICK(13);
  if (WITH_REWARDS && !rewards_early && !rewards_mid) {
    // (reward expressions with query operands never reach this kernel: mgx_values_kernel evaluates them, one env per
    // lane, after it — the host launches this kernel without rewards then)
    for (int a = tid; a < A; a += NTH) agent_rewards(a);
  }
  MGX_TICK(14);
}

#endif  // MGX_OBS_H_
