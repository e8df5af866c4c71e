// mgx_act.h — the action dispatch with one lane per AGENT (mettagrid_c.cpp:921-999, and for lean games the rest of the
// world update: per-agent on_tick :1019-1024, coverage tracking :1054-1056).
//
// The reference walks the agents of an env one after another in the step's shuffled order, and that order is
// observable wherever two agents touch the same thing: the cell both want to step into, the extractor both draw from,
// the agent one of them attacks.  Everywhere else it is not.  An action (actions/action_handler.hpp:78-105) reads and
// writes the acting agent, the cell it stands on, the cell in front of it and the object there — its FOOTPRINT is
// {own cell, target cell} — plus counters of the two agents involved.  Two actions with disjoint footprints commute.
//
// So: a run of A' lanes (A' = agents rounded up to a power of two, at most one wavefront) owns one env, lane p holds
// the agent at place p of the shuffled order, and the dispatch proceeds in ROUNDS.  In every round each pending lane
// publishes its footprint, looks at the pending lanes BEFORE it in the order, and runs its action now if none of them
// shares a cell with it; otherwise it waits for the next round.  An agent standing on the cell an earlier pending agent
// acts on may be swapped elsewhere before its turn — its footprint is unknown until then, so everything behind it in
// the order waits too and footprints are recomputed every round.  The lowest pending lane always runs, two lanes that
// run in the same round never share a cell, and a lane never overtakes an earlier lane it could interact with: the
// result is the reference's, bit for bit.  Typical games need two or three rounds instead of A serial steps, and the wavefronts are full.
//
// What makes this legal is decided by the host (mgx_engine.hip, `act_par`): every handler an action can reach only
// touches actor and target (no tag mutation, query, object creation / removal, no game-scope stat read); move handlers
// look one cell ahead; agents are no territory sources.  Game-scope stats that handlers SET are collected per env and
// applied by position (MgxEnvT::act_gstat_set).  Everything else runs the lane-per-env kernels (mgx_world.h).
// The per-agent code is the very same MgxEnvT the lane-per-env kernels run.
#ifndef MGX_ACT_H_
#define MGX_ACT_H_
#ifndef MGX_ACT_TU
#error "mgx_act.h is for the agent-parallel translation units (MGX_ACT_TU)"
#endif
#include "mgx_world.h"

namespace MGX_TU_NS {

#ifdef MGX_ACT_DEBUG  // (developer build) trace of one env: per order position the agent, the round it ran in, its footprint
__device__ int mgx_act_dbg_env = -1;
__device__ uint32_t mgx_act_dbg[192];
#endif
// Same-wavefront visibility: the lanes of an env live in one wavefront, whose LDS and vector-memory operations are
// issued in program order; what is needed is that the compiler keeps that order (and nothing cached in registers).
__device__ __forceinline__ void mgx_act_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

// Applies and clears the env's collected game-stat sets (lane 0 of the env).
template <class ENV>
__device__ __forceinline__ void mgx_act_apply_gsets(const ENV& e, const MgxDev& d, int envl) {
  unsigned long long* cells = e.act_gset_cells();
  for (int k = 0; k < d.act_ngset; k++) {
    const unsigned long long w = cells[k * MGX_WORLD_EPG + envl];
    if (w) {
      e.gstat_set(d.act_gset_ids[k], __uint_as_float((uint32_t)w));
      cells[k * MGX_WORLD_EPG + envl] = 0ull;
    }
  }
}

// std::shuffle of the env's agent order (bits/stl_algo.h:3729-3795; same draws as mgx_shuffle_order), spread over the lanes
// of the env's run.  The serial form costs a 64-agent env 32 draws one after another — each an integer multiply and three
// integer divisions, ≈ 80 instructions — which every lane of a one-env wavefront would execute redundantly: it was 38 % of
// the kernel.  Here lane q < A / 2 produces generator output q itself (incremental twist of state word i0 + q: it needs
// the old words i0 + q + 1 and i0 + q + 397, none of which another lane of the block rewrites) and turns it into draw q's
// pair of swaps; the swaps — the only serial part — are then applied in draw order, the order held one element per lane
// (lane p holds order[p]; a swap is one cross-lane read).  A Lemire rejection (probability ≈ range / 2^32 per draw) shifts
// every later draw by one output: an env that sees one replays the draws serially from the outputs already made, pulling
// the extra output(s) with rng_next.  Every lane of the wavefront executes the same instruction sequence.
template <class ENV>
__device__ __forceinline__ int mgx_act_shuffle(const ENV& e, const MgxDev& d, int A, int p, int segbase, unsigned long long seg, bool env_valid) {
  int ord = p;
  const uint32_t nd = (env_valid && A >= 2) ? (uint32_t)A / 2 : 0u;  // A even: 1 + (A - 2) / 2 draws, A odd: (A - 1) / 2
  auto swap = [&](int a, int b) {   // a == 0xFF: nothing
    const int src = a == 0xFF ? p : p == a ? b : p == b ? a : p;
    ord = __shfl(ord, segbase + src);
  };
  // the two swaps of draw j from generator output r: a1 | b1 << 8 | a2 << 16 | b2 << 24; accept = no Lemire rejection
  auto draw = [&](uint32_t j, uint32_t r, bool& accept) -> uint32_t {
    const bool first_even = (A & 1) == 0 && j == 0;
    const uint32_t i = (A & 1) == 0 ? 2 * j : 2 * j + 1;  // index of the pair's first element (paired draws)
    const uint32_t sft = i + 1;
    const uint32_t range = first_even ? 2u : sft * (sft + 1);
    const unsigned long long pr = (unsigned long long)r * range;
    const uint32_t low = (uint32_t)pr;
    accept = low >= range || low >= (0u - range) % range;
    const uint32_t x = (uint32_t)(pr >> 32);
    if (first_even) return 1u | (x << 8) | (0xFFu << 16);
    return i | ((x / (sft + 1)) << 8) | ((i + 1) << 16) | ((x % (sft + 1)) << 24);
  };
  const int ev = e.envi();
  uint32_t rq = 0, packed = 0xFFu | (0xFFu << 16);
  bool rejected = false;
  if ((uint32_t)p < nd) {
    const size_t E = (size_t)d.E;
    const uint32_t i0 = d.mt_idx[ev];
    uint32_t i = i0 + (uint32_t)p; i = i >= 624 ? i - 624 : i;
    uint32_t i1 = i + 1; i1 = i1 >= 624 ? i1 - 624 : i1;
    uint32_t im = i + 397; im = im >= 624 ? im - 624 : im;
    const uint32_t w0 = d.mt[i * E + ev], w1 = d.mt[i1 * E + ev], wm = d.mt[im * E + ev];
    const uint32_t y = (w0 & 0x80000000u) | (w1 & 0x7fffffffu);
    uint32_t x = wm ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    d.mt[i * E + ev] = x;   // (data dependence on the loads of every lane: all old words are read before any is rewritten)
    x ^= x >> 11;
    x ^= (x << 7) & 0x9d2c5680u;
    x ^= (x << 15) & 0xefc60000u;
    x ^= x >> 18;
    rq = x;
    if (p == 0) { const uint32_t n1 = i0 + nd; d.mt_idx[ev] = n1 >= 624 ? n1 - 624 : n1; }
    bool accept;
    packed = draw((uint32_t)p, rq, accept);
    rejected = !accept;
  }
  if (d.act_replay) rejected = true;   // (MGX_ACT_SHUFFLE_REPLAY: every step takes the replay path below — how the tests reach it)
  const bool env_rejected = (__ballot(rejected) & seg) != 0ull;
  if (!__any(env_rejected)) {
    const uint32_t ndmax = (uint32_t)A / 2;   // (the same for every env of the launch)
    for (uint32_t j = 0; j < ndmax; j++) {
      const uint32_t pk = (uint32_t)__shfl((int)packed, segbase + (int)j);
      swap((int)(pk & 0xFF), (int)((pk >> 8) & 0xFF));
      swap((int)((pk >> 16) & 0xFF), (int)(pk >> 24));
    }
    return ord;
  }
  // a rejection somewhere in the wavefront: every env replays its draws serially over the outputs (same result for the
  // envs without one)
  mgx_act_fence();   // the parallel twist's stores are visible to rng_next below
  uint32_t j = 0, k = 0;   // draws done, outputs consumed (the same in every lane of the run)
  while (__any(j < nd)) {
    uint32_t extra = 0;
    if (j < nd && k >= nd && p == 0) extra = e.rng_next();
    const uint32_t from_block = (uint32_t)__shfl((int)rq, segbase + (int)min(k, nd > 0 ? nd - 1 : 0u));
    const uint32_t from_extra = (uint32_t)__shfl((int)extra, segbase);
    const uint32_t r = k < nd ? from_block : from_extra;
    bool accept = false;
    uint32_t pk = 0xFFu | (0xFFu << 16);
    if (j < nd) {
      const uint32_t v = draw(j, r, accept);
      if (accept) { pk = v; j++; }
      k++;
    }
    swap((int)(pk & 0xFF), (int)((pk >> 8) & 0xFF));
    swap((int)((pk >> 16) & 0xFF), (int)(pk >> 24));
  }
  return ord;
}

// envl: env inside the workgroup, p: lane inside the env's run, seg: the run's lanes inside the wavefront
template <class PP, bool X>
__device__ __forceinline__ void mgx_act_body(const MgxDev& d, PP P, uint8_t* order, MgxALds al, int envl, int p, int env, bool valid,
                                             unsigned long long seg) {
  MgxEnvT<PP, X> e(d, P, env);
  e.al_ = al;
  const int A = d.A;
  const int wl = (int)(threadIdx.x & (MGX_WAVE - 1));
  MGX_TICK0();
  uint32_t step = 0;
  if (valid) {  // ++current_step (:933); executed_actions / _action_success cleared (:944,962-964)
    step = d.step[env] + 1;
    if (p == 0) {
      d.step[env] = step;
      unsigned long long* cells = e.act_gset_cells();
      for (int k = 0; k < MGX_ACT_GSET; k++) cells[k * MGX_WORLD_EPG + envl] = 0ull;
    }
  }
  e.step = step;
  // ---- every lane stages ITS agent (index p): own state + both action ids ----
  if (valid) {
    const int i = p;
    const uint16_t slot = d.ag_obj[e.ao(i)], prev = d.ag_prev[e.ao(i)];
    const uint32_t swm = d.ag_swm[e.ao(i)];
    const int32_t a = d.actions[e.ao(i)], v = d.vibe_actions[e.ao(i)];
    const uint16_t rc = d.ag_rc[e.ao(i)], cls = d.ag_cls[e.ao(i)];   // (per-agent mirrors of the object row: one round of loads)
    const int li = i * MGX_WORLD_EPG + envl;
    al.slot[li] = slot; al.rc[li] = rc; al.prev[li] = prev; al.swm[li] = swm;
    if (al.cls) al.cls[li] = cls;
    al.act[li] = mgx_sat16(a); al.act[A * MGX_WORLD_EPG + li] = mgx_sat16(v);
    if (d.flags & MGX_G_LAST_ACTION_MOVE) d.ag_stepprev[e.ao(i)] = rc;  // mettagrid_c.cpp:929-931
    d.executed[e.ao(i)] = 0;
    d.success[e.ao(i)] = 0;
  }
  mgx_act_fence();
  MGX_TICK(0);
  const int segbase = wl - p;
  const int ai = mgx_act_shuffle(e, d, A, p, segbase, seg, env < d.E);
  MGX_TICK(1);
  // footprints u32[A'][EPG] (by order position) and, when the map is small enough, cell -> order position of the pending
  // agent standing there u8[EPG][H * W] (entries are validated against the footprints, so stale ones are harmless)
  uint32_t* fps = (uint32_t*)(mgx_dyn_lds + mgx_world_lds_fixed(A, X, d.x_aoe_lds != 0));
  const int apad = (int)(blockDim.x / MGX_WORLD_EPG);
  uint8_t* cmap = (uint8_t*)(fps + apad * MGX_WORLD_EPG) + (size_t)envl * ((d.H * d.W + 15) & ~15);
  e.act_pos = p;
  PP acts = P + d.sec[MGX_SEC_ACTIONS];
  const int repeats = d.max_priority + 1;

  // ---- primary stream: conflict-ordered rounds; then the vibe stream in one round (change_vibe only writes the acting
  // agent, actions/change_vibe.hpp:48-57).  One loop, so the handler interpreter is instantiated once. ----
  bool pending = valid;
  int stream = 0;
#ifdef MGX_ACT_DEBUG
  int dbg_round = 0;
#endif
  for (;;) {
    const unsigned long long pend = __ballot(pending);
    if (!pend) {
      MGX_TICK(2 + stream);
      if (stream == 1) break;
      stream = 1;
      pending = valid;
      continue;
    }
#ifdef MGX_WORLD_TIMING
    if (threadIdx.x == 0 && stream == 0) atomicAdd(&mgx_dbg_cycles[8], 1ull);
#endif
    bool clear = pending;
#ifdef MGX_WORLD_TIMING
    const unsigned long long tq0 = clock64();
#endif
    if (stream == 0) {
      uint32_t own = 0, tgt = 0;
      if (pending) {
        own = al.rc[ai * MGX_WORLD_EPG + envl];
        tgt = own;
        const int a = al.act[ai * MGX_WORLD_EPG + envl];
        if (a >= 0 && a < d.nact && acts[a * MGX_AC_WORDS + MGX_AC_KIND] == MGX_AK_MOVE && d.n_move_handlers > 0) {
          const int orient = acts[a * MGX_AC_WORDS + MGX_AC_ARG];  // orientation.hpp:28-48
          const int dx = (orient == 2 || orient == 4 || orient == 6) ? -1 : (orient == 3 || orient == 5 || orient == 7) ? 1 : 0;
          const int dy = (orient == 0 || orient == 4 || orient == 5) ? -1 : (orient == 1 || orient == 6 || orient == 7) ? 1 : 0;
          const int r = (int)(own >> 8) + dy, c = (int)(own & 0xFF) + dx;
          if (r >= 0 && c >= 0 && r < d.H && c < d.W) tgt = (uint32_t)((r << 8) | c);
        }
      }
      const uint32_t F = (own << 16) | tgt;
#ifdef MGX_ACT_STRICT  // (debug) one agent per round, in order: the dispatch of the lane-per-env kernels
      if ((pend & seg & ((1ull << wl) - 1ull)) != 0ull) clear = false;
#endif
      bool hit = false;  // an earlier pending agent acts on THIS agent's cell
      if (d.act_map) {
        // Every conflict is local: the other agent stands on my target cell, or next to my cell / my target cell and acts
        // on it.  Pending agents publish footprint and cell; a lane then looks at the 17 cells that matter.
        if (pending) {
          fps[p * MGX_WORLD_EPG + envl] = F;
          cmap[(own >> 8) * d.W + (own & 0xFF)] = (uint8_t)p;
        }
        mgx_act_fence();
        if (pending) {
          // 17 cells: [0] my target, [1..8] the neighbours of my cell, [9..16] the neighbours of my target.  All map reads
          // first, then all footprint reads, then the decisions: two LDS latencies instead of thirty-four in a row.
          const int orow = (int)(own >> 8), ocol = (int)(own & 0xFF), trow = (int)(tgt >> 8), tcol = (int)(tgt & 0xFF);
          const bool has_tgt = tgt != own;
          int jj[17];
          uint32_t rc17[17];
          bool ok[17];
#pragma unroll
          for (int k = 0; k < 17; k++) {
            const int n = k == 0 ? 4 : (k - 1) & 7;          // neighbour number 0..7 -> (dr, dc) without the centre
            const int m = n < 4 ? n : n + 1;
            const int dr = k == 0 ? 0 : m / 3 - 1, dc = k == 0 ? 0 : m % 3 - 1;
            const int r = (k >= 1 && k <= 8 ? orow : trow) + dr, c = (k >= 1 && k <= 8 ? ocol : tcol) + dc;
            ok[k] = r >= 0 && c >= 0 && r < d.H && c < d.W && (k >= 1 && k <= 8 ? true : has_tgt);
            rc17[k] = (uint32_t)((r << 8) | c);
            jj[k] = cmap[ok[k] ? r * d.W + c : 0];
          }
          uint32_t fj[17];
#pragma unroll
          for (int k = 0; k < 17; k++) {
            ok[k] = ok[k] && jj[k] < p && ((pend >> (segbase + (jj[k] & 63))) & 1ull);
            fj[k] = fps[(ok[k] ? jj[k] : 0) * MGX_WORLD_EPG + envl];
          }
#pragma unroll
          for (int k = 0; k < 17; k++) {
            const bool there = ok[k] && (fj[k] >> 16) == rc17[k];   // an earlier pending agent stands on that cell
            if (k == 0) { if (there) clear = false; }                                            // tgt == oj
            else if (k <= 8) { if (there && (fj[k] & 0xFFFFu) == own) { clear = false; hit = true; } }   // own == tj
            else { if (there && (fj[k] & 0xFFFFu) == tgt) clear = false; }                      // tgt == tj
          }
        }
      } else
      for (unsigned long long m = pend; m;) {  // wave-uniform walk over the pending lanes
        const int j = __builtin_ctzll(m);
        m &= m - 1;
        const uint32_t Fj = (uint32_t)__builtin_amdgcn_readlane((int)F, j);
        const uint32_t oj = Fj >> 16, tj = Fj & 0xFFFFu;
        if (j < wl && ((seg >> j) & 1ull)) {
          if (own == tj) hit = true;
          if (own == tj || tgt == oj || tgt == tj) clear = false;
        }
      }
      // An agent that an earlier one acts on may be swapped to another cell (swap_mutation.hpp:15-21) before its turn: its
      // footprint is not known yet, so nothing behind it in the order may run this round.
      const unsigned long long unstable = __ballot(hit);
      if ((unstable & seg & ((1ull << wl) - 1ull)) != 0ull) clear = false;
#ifdef MGX_ACT_DEBUG
      if (clear && env == mgx_act_dbg_env) { mgx_act_dbg[p] = (uint32_t)ai | ((uint32_t)dbg_round << 8); mgx_act_dbg[64 + p] = F; mgx_act_dbg[128 + p] = (uint32_t)al.act[ai * MGX_WORLD_EPG + envl]; }
      dbg_round++;
#endif
    }
#ifdef MGX_WORLD_TIMING
    const unsigned long long tq1 = clock64();
#endif
    if (clear) {
      mgx_dispatch_one(e, d, acts, al, ai, stream, envl, repeats);
      pending = false;
    }
    mgx_act_fence();
#ifdef MGX_WORLD_TIMING
    if (threadIdx.x == 0) { atomicAdd(&mgx_dbg_cycles[9], tq1 - tq0); atomicAdd(&mgx_dbg_cycles[10], clock64() - tq1); }
#endif
  }
  if (valid && p == 0 && d.act_ngset > 0) mgx_act_apply_gsets(e, d, envl);
  // ---- per-agent on_tick (mettagrid_c.cpp:1019-1024): agent index order, every handler confined to its own agent ----
  if (d.act_tick && d.any_on_tick) {
    if (valid) {
      const int li = p * MGX_WORLD_EPG + envl;
      const int slot = al.slot[li];
      const int h = (al.cls ? e.cls(al.cls[li]) : e.cls_of(slot))[MGX_C_ON_TICK];
      if (h >= 0) {
        MgxCtx c = mgx_ctx(slot, slot);
        e.apply_top(h, c);
      }
    }
    mgx_act_fence();
    if (valid && p == 0 && d.act_ngset > 0) mgx_act_apply_gsets(e, d, envl);
  }
  MGX_TICK(4);
  // ---- deferred bookkeeping and (lean games) coverage tracking: own agent only ----
  if (valid) {
    if (d.defer_book) e.bookkeeping_flush_one(p);
    if constexpr (!X) e.track_coverage(p);
  }
  MGX_TICK(5);
}

template <bool PROG_LDS, bool X>
__device__ __forceinline__ void mgx_act_entry(const MgxDev& d, int prog_words) {
  uint8_t* order = mgx_dyn_lds;
  const int sh = mgx_act_shift();
  const int envl = (int)(threadIdx.x >> sh);
  const int p = (int)(threadIdx.x & ((1u << sh) - 1u));
  const int env = blockIdx.x * MGX_WORLD_EPG + envl;
  const bool valid = env < d.E && p < d.A;
  const int wl = (int)(threadIdx.x & (MGX_WAVE - 1));
  const unsigned long long seg = sh >= 6 ? ~0ull : (((1ull << (1 << sh)) - 1ull) << (wl & ~((1 << sh) - 1)));
  const MgxALds al = mgx_world_alds(mgx_dyn_lds, d.A, envl);
  if (PROG_LDS) {
    int32_t* lprog = (int32_t*)(mgx_dyn_lds + mgx_world_lds_fixed(d.A, X, d.x_aoe_lds != 0, d.act_lds_extra));
    const int4* src = (const int4*)(d.P + d.hot_lo);
    int4* dst = (int4*)lprog;
    for (int i = threadIdx.x; i < prog_words / 4; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
    mgx_act_body<MgxLdsProg, X>(d, (MgxLdsProg)lprog, order, al, envl, p, env, valid, seg);
  } else {
    mgx_act_body<MgxGlobalProg, X>(d, d.P, order, al, envl, p, env, valid, seg);
  }
}

}  // namespace MGX_TU_NS
#endif
