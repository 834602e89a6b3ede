// step_state.hpp -- the device-resident Floe struct array of one context and everything a timestep keeps between
// its stages (shared by step.hip: the step itself, and sync.hip: upload / download / partial synchronisation).
#pragma once
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include "context.hpp"
#include "step_core.hpp"

using namespace szs;

// what the back half of the step -- contact rows, calc_trajectory -- is to do (step.hip k_publish)
enum { SP_HDR_WORDS = 8 };   // (= SP_HDR of spatial_core.hpp: int64s of a message header)
struct GoRule { int mode; long long spec_np, spec_nvt, rows_cap; int team_pending; long long next_cap; };   // next_cap > 0: entries of the extended list the queued back half covers   // team_pending: pairs on team_list have not had their pass   // mode 0: leave C_GO, 1: go, 2: go if the narrow phase is complete and fits
enum { GO_LEAVE = 0, GO_SET = 1, GO_IF_CLEAN = 2 };

struct StepState {
  Floes F;            // device pointers
  Work W;             // device pointers
  std::vector<DevBuf> bufs;
  int n = 0, nb = 0, n_stress = 0, n_mc = 0, nverts = 0;
  int pot_cap = 32;               // candidates / mirrored pairs per floe the lists hold; grows when a step needs more (sticky)
  DevBuf pot_list, inc_list;      // [cap_ext * pot_cap] each
  DevBuf stage_d, stage_i, stage_ids; // packed transfers of szFloesUpdate / szFloesFetch
  std::vector<double> h_rmax;     // host copies of rmax (sizes the grids) and of the ring layout (checked by szFloesUpdate)
  std::vector<int> h_vert_off;
  std::vector<int> mc_empty;      // floes (1-based) the last szStep asked new Monte-Carlo points for (SZ_STEP_NEEDS_MC)
  // ocean grids on the device
  DevBuf oXo, oYo, oU, oV, oWu, oWv;
  int nxo = 0, nyo = 0;
  double xo_max = 0, xo_min = 0, yo_max = 0, yo_min = 0; // extent of the uploaded grid axes
  bool have_ocean = false;
  // grow-only per-step buffers
  DevBuf pair_i, pair_j, pair_nvoff, pair_scr, pair_mult, pair_nrows, pair_status, pair_flag, retry_list, retry_copy, big_list, team_list, wide_list, live_list, cold_list, wall_list;
  DevBuf scratch, rows;
  DevBuf cell_cnt, cell_off, cell_fill, cell_item;
  DevBuf pair_rowoff, x_off, touch_list, pair_nreg, pair_task, filter_spill;
  DevBuf post[64];                // device buffers of the post-step entries (corners.hip, eulerian.hip, fracture.hip): grow-only, kept between calls
  std::vector<long long> hdr;     // segment headers of all ranks (szStepShardFinish)
  long long *scan_agg = nullptr;  // up to 4 x 256 workgroup aggregates of the scan in flight (launch_scan)
  int fast_path = 0;              // two-stage LDS path for small rings (opt-in: SUBZERO_FAST_PATH=1)
  int fast_lanes = 64;            // pairs per workgroup on the fast path (SUBZERO_FAST_LANES)
  double rmax_max = 0, rmax_p90 = 0; // over the floes that can be candidates (i >= Nb)
  double rmax_max_all = 0;           // over all floes (the reach of the spatially sharded step)
  // spatially sharded step (spatial.hip)
  int sp_pack_periodic = 0;        // (szSpatialPackQueued's argument, for szSpatialPackFinish)
  bool sp_unpack_queued = false;   // szSpatialUnpackQueued since the last szStep: that step honours C_VETO
  long long steps_vetoed = 0;
  int sp_rank = 0, sp_world = 1, sp_axis = 0;   // (axis the strips are cut along: 0 x, 1 y; szSpatialSetAxis)
  double sp_halo = 0;
  std::vector<double> sp_edges;
  DevBuf sp_masks, sp_edges_dev, sp_counts;   // present[n] then owned[n]; strip edges; two counters
  // pinned, 5 x SP_HDR_WORDS: the headers of the two messages a Pack produced, of the two a queued Unpack took in, the veto
  long long *sp_hdr = nullptr;
  Params P;                       // parameters of the step in flight
  unsigned long long *c = nullptr; // host copy of the device counters (pinned; k_publish writes it)
  unsigned long long *c_dev = nullptr; // the same buffer as the device sees it
  i64 pair_rows_total = 0;        // contact rows of all pairs of the step (bounds the rows of the CSR)
  unsigned long long bump = 0, zero = 0;
  i64 np = 0, nvt = 0, retried = 0, pair_rows_local = 0, seg_lo = 0, seg_hi = 0;
  int N = 0, rank = 0, world = 1;
  i64 nr = 0;
  int force_exchange = 0;
  int team = 1;                   // the team pass runs first (teamcontact.hpp; SUBZERO_TEAM=0: lane-per-pair fast path for all)
  i64 last_hot = -1, last_cold = -1, last_wide = -1;   // what the team pass's lists held in the last step (sizes its grid)
  bool classified = false;        // this step's pairs went through pair_classify (live / wide / team lists are filled)
  int convex_fast = 1;            // first pass with the single-minimum fast path only (convex.hpp); SUBZERO_CONVEX_FAST=0: general sweep for all
  int convex_lanes = 0;           // pairs per wave of that pass (SUBZERO_CONVEX_LANES; 0: chosen per launch)
  int broad_wave = 1;   // the wave-per-floe broad phase for real floes (SUBZERO_BROAD_WAVE=0: thread-per-floe kernels only)
  bool wall_tasks = false;        // the same for the floes at the walls (SUBZERO_WALL_TASKS=1; measured: 0.424 -> 0.414 ms/step between walls, left off)
  int wave_tasks = 3;             // waves per deferred pair in the wave-per-pair general pass (SUBZERO_WAVE_TASKS, 1-4): a pair with several overlap regions is shared out over up to that many (step_core.hpp narrow_t)
  int big_lanes = 0;              // pairs per wave of the second fast pass (32-vertex blocks, SUBZERO_BIG_LANES; 0: no such pass: its pairs take
                                  // the general kernel -- the default until the round-1 fault of that kernel is settled)
  int retry_lanes = 0;            // pairs per wave when the general kernel redoes what the fast pass deferred; 0: as few as keep
                                  // the chip at about two waves per SIMD (SUBZERO_RETRY_LANES)
  int retry_lds_max = 65536;      // ... and what it may grow to while the launch still fits the chip at once (SUBZERO_RETRY_LDS_MAX)
  int retry_lds = 16384;          // bytes of LDS per pair for that pass when its list is short (SUBZERO_RETRY_LDS, 0: HBM workspace)
  int narrow_lanes = 64;          // pairs per wave in k_narrow (tuning knob, SUBZERO_NARROW_LANES)
  int narrow_lds = 0;             // bytes of LDS per lane for the sweep workspace (SUBZERO_NARROW_LDS)
  i64 last_npairs = 0, last_nrows_real = 0;
  int last_next = 0;
  bool stepped = false;
  bool speculating = false, no_speculation = false;   // szStep queues front + narrow phase without the read-back in between
  i64 spec_np = 0, spec_nvt = 0, spec_misses = 0;     // ... sized by these capacities (0: no previous step to size from)
  i64 spec_rows = 0, back_misses = 0;                 // the back half of the step queued behind the narrow phase as well: contact-row room it is sized for
  bool back_queued = false;                           // this step's back half was queued behind the narrow phase (set while it is in flight)
  bool back_spec_ok = false, no_back_speculation = false;   // ... when the previous step deferred no pair (SUBZERO_SPECULATE_BACK=0: never)
  bool counters_clean = false;    // the last step ended by resetting the device counters (k_publish with reset)
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // The stage times of a step (szStepStats) are read from its events one step LATE: the runtime sees an event's signal some
  // microseconds after the step's own completion flag reaches the host, and waiting for it cost every step 8 us (measured).
  // Two sets of events take turns; the set of the step before is read while the host waits for the current step anyway.
  hipEvent_t ev_prev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool ev_recorded = false, prev_pending = false;
  float prev_ms[6] = {0, 0, 0, 0, 0, 0};      // step, broad, pairs, narrow, rows, integrate of step `prev_step`
  long long prev_step = -1;
  // Queue-ahead (szStepParams.flags & SZ_STEP_QUEUE_AHEAD): the first two kernels of the NEXT step (kf_front_a / kf_front_b:
  // they read the floe array and a handful of parameters, nothing of the pair buffers) are queued behind this step's last
  // kernel before the host waits for it, so that the device has work while the host returns to its caller and comes back
  // (24 us between two steps otherwise).  They run only if this step ran to its end (C_AHEAD); the next szStep uses them
  // if nothing they read has changed -- parameters, grid, buffers -- and launches them afresh otherwise.
  struct FrontSig { const void *cell_item, *cell_cnt, *gh_mask, *gh_agg, *row_agg, *prow_part; int cell_cap, cap_ext, nc, ntiles, ntiles_ext, pad; };
  bool ahead_valid = false, ahead_pending = false;
  bool ghosts_overwritten = false; // the last step queued the next one's front kernels: they have rebuilt the ghost list (szGhostsDownload refuses)
  FrontSig ahead_sig = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0}, front_sig = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0};
  Params ahead_P;
  int step_flags = 0;              // szStepParams.flags of the step under way
  long long steps_ahead = 0;       // szStepDiagnostics: steps whose front kernels had been queued ahead and were used
  // the floes at the walls are swept on a stream of their own beside the team pass (step.hip step_narrow_first)
  hipEvent_t wall_fork = nullptr, wall_join = nullptr;
  long long last_team = 0;        // pairs the team pass handed on to the lane-per-pair fast pass in the step before
  GoRule go_rule = {0, 0, 0, 0, 0, 0};
  i64 next_cap = 0;               // fused chain: entries of the extended list the queued back half's launches cover
  bool publish_with_incoming = false;   // the counters after the narrow phase are published by the back half's first kernel (k_incoming_go)
  bool convex_skipped = false;    // ... whose launch this step left out (step_narrow_first)
  long long last_walls = 0;       // pairs on the wall list in the step before (sizes the launch)
  long long last_deferred = 0;    // pairs the step before handed to the general pass after its read-back, the wall pairs not counted
  int walls_early = 1;            // SUBZERO_WALLS_EARLY=0: the wall pairs wait for the general pass after the read-back
  // the fused launch chain of the steady step (step_fused.hpp)
  bool fused_on = true;           // SUBZERO_FUSED=0: the scan-based chain always
  bool fused = false;             // this step's front ran fused (its back half then does too)
  int cell_cap = 32;              // slots per grid cell (SUBZERO_CELL_CAP); doubles when a step finds a cell full
  DevBuf gh_mask, gh_agg, pair_agg, row_agg, rel_mask, pot_nvpre, force_list, hint;
  bool hint_valid = false;        // the step before ran fused: W.hint holds its contacts
  long long steps_done = 0, steps_fused = 0, last_team_vertices = 0;   // szStepDiagnostics
  bool spin_wait = true;          // the host waits for the step's last kernel by spinning on a flag in the pinned counters (SUBZERO_SPIN_WAIT=0: hipStreamSynchronize)
  unsigned long long publish_seq = 0;
  bool no_hint = false, debug_lists = false;   // SUBZERO_NO_HINT, SUBZERO_DEBUG_LISTS (read once)
};


// narrow_team.hip: the team pass of the narrow phase
void sz_launch_pair_classify(hipStream_t st, i64 count, const unsigned long long *dev_count, int prev_n, const Params &P, const Work &W);
void sz_launch_narrow_team(hipStream_t st, i64 count, i64 hot, i64 cold, i64 wide, const Floes &F, const Params &P, const Work &W);
void sz_launch_narrow_team_range(hipStream_t st, i64 first, i64 count, const Floes &F, const Params &P, const Work &W);

inline dim3 grid_for(i64 n, int block) { return dim3((unsigned)((n + block - 1) / block > 0 ? (n + block - 1) / block : 1)); }

template <class T>
inline int dev_alloc(szContext *ctx, StepState *S, T *&ptr, size_t count) {
  S->bufs.emplace_back();
  DevBuf &b = S->bufs.back();
  int rc = sz_reserve(ctx, b, count * sizeof(T) + 64);
  if (rc) return rc;
  ptr = (T *)b.p;
  return SZ_OK;
}
template <class T>
inline int dev_put(szContext *ctx, StepState *S, T *&ptr, const T *src, size_t count) {
  int rc = dev_alloc(ctx, S, ptr, count);
  if (rc) return rc;
  if (count) SZ_HIP(ctx, hipMemcpy(ptr, src, count * sizeof(T), hipMemcpyHostToDevice));
  return SZ_OK;
}


// largest and 90th-percentile bounding radius of the floes that can be candidates (i >= Nb): they size the grids
inline void rmax_stats(StepState *S, const double *rmax) {
  std::vector<double> r(rmax + S->nb, rmax + S->n);
  S->rmax_max = 0; S->rmax_p90 = 0; S->rmax_max_all = 0;
  for (int i = 0; i < S->n; ++i) if (rmax[i] > S->rmax_max_all) S->rmax_max_all = rmax[i];
  if (r.empty()) return;
  for (double v : r) if (v > S->rmax_max) S->rmax_max = v;
  const size_t k = (size_t)(0.9 * (double)(r.size() - 1));
  std::nth_element(r.begin(), r.begin() + k, r.end());
  S->rmax_p90 = r[k];
}
// candidate and mirror lists at the current capacity
inline int bind_lists(szContext *ctx, StepState *S) {
  Work &W = S->W;
  int rc;
  const size_t bytes = (size_t)W.cap_ext * S->pot_cap * 4;
  if ((rc = sz_reserve(ctx, S->pot_list, bytes)) || (rc = sz_reserve(ctx, S->inc_list, bytes))) return rc;
  W.pot_list = (int *)S->pot_list.p; W.inc_list = (int *)S->inc_list.p; W.pot_cap = S->pot_cap;
  return SZ_OK;
}
