// step_core.hpp -- the per-timestep pipeline of floe_interactions_all.m:16-285 as a sequence of kernel
// bodies.  Each body is the work of ONE thread of its kernel; step.hip wraps every body in a __global__
// launcher, and tests/hostsim runs the same bodies in plain loops on the CPU, so the whole GPU step can be
// checked against the oracle without a GPU.
//
// Stages (reference lines) and their parallel shape:
//   ext_init / ghost_flag_* / ghost_emit_*   :16-66    one thread per floe, stable compaction by prefix sum
//   broad                                     :76-120   one thread per (extended) floe i, all j > i
//   pair_count / pair_expand                  :123-133  per-floe pair counts -> prefix sum -> flat pair list
//   narrow                                    :133,151  one thread per candidate pair (contact.hpp)
//   incoming                                  :187-214  pairs register with their partner floe
//   row_count / row_fill                      :136-145,167,198,221-263  per-floe contact rows (CSR), in the
//                                             reference's row order; torques, sums, kill/transfer marks
//   integrate                                 :242-245,249-285 + calc_trajectory.m
// Summation order inside every floe follows the reference (own rows ascending partner, wall, mirrored rows
// ascending partner), so forces do not depend on scheduling.
#pragma once
#include "contact.hpp"
#include "teamcontact.hpp"
#include "detmath.hpp"

namespace szs {

using szv::i64;

enum { WALL = -2 }; // partner id of the floe-vs-boundary candidate
enum { AGG_STRIDE = 16 }; // words between two sums that atomics bump: a 128-byte line each (the L2 serves the atomics of one line one by one)

// device image of the Floe struct array (SoA).  Field names follow initialize_floe_values.m:12-52.
struct Floes {
  int n, nb, n_stress, n_mc;
  double *Xi, *Yi, *alpha_i, *Ui, *Vi, *ksi_ice;
  double *h, *area, *mass, *inertia_moment, *rmax;
  double *dXi_p, *dYi_p, *dUi_p, *dVi_p, *dalpha_i_p, *dksi_ice_p;
  double *FxOA, *FyOA, *torqueOA;
  int *alive;
  int *vert_off;
  double *c0x, *c0y, *cax, *cay;
  double *StressH;      // [floe][slot][4]
  int *StressCount;
  double *Stress, *strain;
  double *mcX, *mcY;    // point q of floe i at [i * mc_si + q * mc_sq]
  unsigned char *mcA;   // ([floe][q] everywhere today: mc_si = n_mc, mc_sq = 1)
  long long mc_si, mc_sq;
  double *collision_force, *collision_torque, *OverlapArea;
  int *kill, *transfer; // capacity 4 * n (extended list)
  double *StressSum;    // [floe][8] sum of the StressH slots as an unevaluated pair hi[4], lo[4] (device only; stress_sum_*)
};

struct Params {
  double dt, HFo, Modulus, mu, nu;
  int doInt, COLLISION, PERIODIC, reserved;
  double c2_boundary[10];
  double fb_h, fb_area, fb_Xi, fb_Yi, fb_Ui, fb_Vi, fb_ksi_ice;
  int nxo, nyo;
  const double *Xo, *Yo, *Uocn, *Vocn, *Uwind, *Vwind;
  double fCoriolis, turn_angle;
  double Lx, Ly; // max of c2_boundary rows (:9-10), filled by the host wrapper
  double Xo_max, Xo_min, Yo_max, Yo_min; // extent of the ocean grid axes (calc_trajectory.m:116), filled by the host wrapper (axis_extent)
  // what the contact model derives from c2_boundary alone, once per step on the host (boundary_summary) -- a kernel that
  // walks the array itself reads it through a pointer, which puts the by-value Params into scratch memory:
  double bnd_xmax, bnd_xmin, bnd_ymax, bnd_ymin, bnd_area;   // floe_interactions.m:54-60: extent of the 5 closed corners, area of the box
  int bnd_rect, bnd_pad;                                     // 1: the 4 corners form an axis-aligned rectangle (contact.hpp well_inside_box)
};
// (the expressions of contact.hpp pair_finish / well_inside_box on c2_boundary, evaluated once)
SZ_HD void boundary_summary(Params &P) {
  const double *c = P.c2_boundary;
  double bxmax = c[0], bxmin = c[0], bymax = c[1], bymin = c[1];
  for (int k = 1; k < 5; ++k) {
    bxmax = szc::maxd(bxmax, c[2 * k]); bxmin = szc::mind(bxmin, c[2 * k]);
    bymax = szc::maxd(bymax, c[2 * k + 1]); bymin = szc::mind(bymin, c[2 * k + 1]);
  }
  double sum = 0;
  for (int i = 0; i < 4; ++i) {
    const int k = (i + 1 == 4) ? 0 : i + 1;
    const double ax = c[2 * i] - c[0], ay = c[2 * i + 1] - c[1];
    const double bx = c[2 * k] - c[0], by = c[2 * k + 1] - c[1];
    sum += ax * by - bx * ay;
  }
  P.bnd_xmax = bxmax; P.bnd_xmin = bxmin; P.bnd_ymax = bymax; P.bnd_ymin = bymin; P.bnd_area = szc::absd(0.5 * sum);
  // well_inside_box: extent of the first 4 corners (the 5th closes the ring: a copy of the first whenever the box is a
  // rectangle the test accepts -- else the flag is off and the pair is left to the lane-per-pair path)
  double x0 = c[0], x1 = c[0], y0 = c[1], y1 = c[1];
  for (int k = 1; k < 4; ++k) { x0 = szc::mind(x0, c[2 * k]); x1 = szc::maxd(x1, c[2 * k]); y0 = szc::mind(y0, c[2 * k + 1]); y1 = szc::maxd(y1, c[2 * k + 1]); }
  bool rect = (c[8] == c[0] && c[9] == c[1]);
  for (int k = 0; k < 4; ++k) {
    const double x = c[2 * k], y = c[2 * k + 1];
    if (!((x == x0 || x == x1) && (y == y0 || y == y1))) rect = false;
    const int k1 = (k + 1) & 3;
    if (!((x == c[2 * k1]) != (y == c[2 * k1 + 1]))) rect = false;
  }
  P.bnd_rect = rect ? 1 : 0; P.bnd_pad = 0;
}

// one entry of ext_rec
enum { REC_X = 0, REC_Y, REC_UI, REC_VI, REC_KSI, REC_H, REC_AREA, REC_BX0, REC_BX1, REC_BY0, REC_BY1,
       REC_NSTAGE,     // vertices to stage: the stored length, one less when the ring is stored closed and one too long for the block (contact.hpp pair_contact_fast); -1: too long
       REC_NSTORED,    // the stored length
       REC_INRANGE,    // 1: every staged coordinate is within the fast path's fixed-point range
       REC_WORDS = 16, RING_SLOTS = 16 };
static_assert((int)RING_SLOTS == (int)sztm::T_NV, "the staged rings are what the team's block holds");

// counters living in device memory.  Each on a 128-byte line of its own (C_STRIDE words apart): the L2 serves atomics on
// one line one after the other (~12 ns each, measured: 1e4 waves on one address take 120 us) and the list counters are
// bumped by hundreds of waves per launch -- side by side in one array they queued behind each other.  The enumerators
// are the word offsets; slot k (k < C_SLOTS) lives at word k * C_STRIDE, on the device and in the host's pinned copy.
enum {
  C_STRIDE = 16,
  C_GX = 0 * C_STRIDE,
  C_GY = 1 * C_STRIDE,
  C_NEXT = 2 * C_STRIDE,
  C_NPAIRS = 3 * C_STRIDE,
  C_NVTOT = 4 * C_STRIDE,
  C_NROWS = 5 * C_STRIDE,
  C_ERR_POT = 6 * C_STRIDE,
  C_ERR_INC = 7 * C_STRIDE,
  C_RETRY = 8 * C_STRIDE,
  C_ERR_CLIP = 9 * C_STRIDE,
  C_ERR_ARENA = 10 * C_STRIDE,
  C_ERR_MC = 11 * C_STRIDE,
  C_PAIRROWS = 12 * C_STRIDE,
  C_TOUCH = 13 * C_STRIDE,
  C_SCR_BUMP = 14 * C_STRIDE,
  C_NEED_POT = 15 * C_STRIDE,
  C_NEED_MC = 16 * C_STRIDE,
  C_BIG = 17 * C_STRIDE,
  C_GO = 18 * C_STRIDE,
  C_TEAM = 19 * C_STRIDE,
  C_WIDE = 20 * C_STRIDE,
  C_LIVE = 21 * C_STRIDE,
  C_COLD = 22 * C_STRIDE,
  C_WALLS = 23 * C_STRIDE,
  C_ERR_CELL = 24 * C_STRIDE,   // fused front: entries that found their grid cell full (Work::cell_cap slots) -- the step is then done again by the scan-based chain
  C_NEED_CELL = 25 * C_STRIDE,   // ... and the fullest cell's demand
  C_TEAMV = 26 * C_STRIDE,   // ring vertices (stored lengths, both rings) of the pairs on the team pass's three lists: what that kernel's roofline is priced on
  C_FORCE = 27 * C_STRIDE,   // fused back half: floes on force_list (forcing due on a step that does not refresh it for everybody: h < 0.1)
  C_AHEAD = 28 * C_STRIDE,   // the step's last publish: 1 when the step ran to its end -- the front kernels queued ahead for the NEXT step (szStepParams.flags) run only then
  C_VETO = 29 * C_STRIDE,    // spatial partition, queued exchange (szSpatialUnpackQueued): a neighbour's message did not fit what was agreed to travel -- the step must not run its back half (GoRule); kept by k_counters_clear, cleared by the pack of the next step
  C_SLOTS = 30, C_COUNT = C_SLOTS * C_STRIDE
};

// one floe in the uniform grid: position, bounding radius and index side by side, so that a candidate test
// reads one 32-byte record
struct CellItem { double x, y, r; int e, pad; };

struct Work {
  int cap_ext;   // 4 * n
  int pot_cap;   // candidates per floe
  i64 cap_pairs; // pairs the scratch arena is sized for
  // extended list: real floes then ghosts
  int *ext_src, *ext_num, *ext_parent, *xg_of, *yg_of;
  int *ext_nv;                  // stored ring length of the entry's floe (vert_off difference), so that pair sizing needs one lookup
  double *ext_x, *ext_y;
  // what the team pass of the narrow phase reads of an entry, laid out for one coalesced load each (cell_count):
  double *ext_rec;              // [REC_WORDS * cap_ext] position, velocities, thickness, area, bounding box, ring length
  double *ext_ring;             // [2 * RING_SLOTS * cap_ext] the first RING_SLOTS vertices of the ring at the entry's position: x then y
  int *flag, *scan;             // cap_ext
  unsigned long long *counters; // C_COUNT
  // broad-phase
  int *pot_cnt, *pot_list;      // [cap_ext], [cap_ext * pot_cap] partner indices, 1-based, ascending
  // flat pair list
  int *pair_cnt;                // per ext floe (incl. wall candidate)
  i64 *pair_nv;                 // per ext floe: sum of ring sizes of its pairs
  i64 *pair_base, *nv_base;     // exclusive prefix sums of the two
  int *pair_i, *pair_j;         // [cap_pairs]
  i64 *pair_nvoff;              // [cap_pairs] vertices in pairs before p
  // narrow-phase results
  char *scratch; i64 scratch_bytes;
  i64 *pair_scr;                // [cap_pairs] offset of the pair's scratch slice
  int *pair_mult;               // [cap_pairs] capacity tier used
  int *pair_nrows, *pair_status, *pair_flag; // flag: 1 sentinel +Inf, 2 sentinel -Inf, 4 contributes rows
  int *retry_list;
  int *wall_list;               // floes at the walls (PC_GENERAL), when the step runs them beside the team pass (null: they go on retry_list)
  int *big_list;                // pairs the first pass hands to the fast path on the 32-vertex block (C_BIG of them)
  int *team_list;               // pairs the team pass hands to the lane-per-pair fast path (C_TEAM of them)
  int *wide_list;               // pairs with more vertices than a 16-lane team has lanes: 32 lanes each (C_WIDE of them)
  int *live_list;               // pairs whose bounding boxes meet and that fit a 16-lane team, in contact in the step before (C_LIVE of them; pair_classify)
  int *cold_list;               // ... not in contact in the step before (C_COLD of them)
  int *touch_list;              // pairs whose first clip found overlap regions (two-stage fast path)
  int *pair_nreg;               // regions of the first clip
  int *pair_task;               // [4 * cap_pairs] the overlap regions of a general pair shared out over waves (narrow_t, k_narrow_wave): contact rows of the pair in the step before (kf_pairs, from the hint), tasks done, their merged status, spare
  // mirrored rows
  int *inc_cnt, *inc_list;      // [cap_ext], [cap_ext * pot_cap] pair ids
  // contact rows, CSR over the extended list
  int *row_cnt; i64 *row_off;   // [cap_ext], [cap_ext + 1]
  double *rows; i64 cap_rows;   // 7 doubles per row
  double *cf_ext, *ct_ext, *ov_ext; // per ext floe: sum Fx,Fy (2), sum torque, OverlapArea
  int *tmax;                    // cap_ext: transfer(kill(i)) = i resolution
  // Hierarchy of uniform grids over the extended list for the broad-phase.  Level k holds the floes with
  // lvl_r[k-1] < rmax <= lvl_r[k] in cells of edge lvl_cs[k] = 2 * lvl_r[k], so that any partner at level b >= the
  // level of floe i lies in the 3 x 3 cells of level b around i (rmax_i + rmax_j <= 2 * lvl_r[b]).  A field of
  // similar floes has one level; one floe 20 times the others gets a coarse level of its own instead of setting the
  // cell size for everybody.  The cells of all levels share one CSR (cell index = lvl_base[k] + cy * lvl_cx[k] + cx).
  enum { MAX_LEVELS = 8 };
  int n_levels;
  double lvl_r[MAX_LEVELS], lvl_cs[MAX_LEVELS], lvl_x0[MAX_LEVELS], lvl_y0[MAX_LEVELS];
  int lvl_cx[MAX_LEVELS], lvl_cy[MAX_LEVELS], lvl_base[MAX_LEVELS + 1];
  int *cell_cnt, *cell_off, *cell_fill;              // [lvl_base[n_levels] (+1)]
  struct CellItem *cell_item;                        // the floes of all cells, in cell order (cap_ext)
  // Spatially sharded step (multi-GPU, spatial.hip): every rank keeps the arrays at their global size and index, but
  // only the floes of its strip (owned) and the halo it received this step are current (present).  null = all floes.
  const unsigned char *present, *owned;
  int *inc_pre;                                      // [cap_ext] candidate pairs that name the floe as partner (bounds inc_cnt)
  int *mc_cnt;                                       // [n] points inside the floe's Monte-Carlo mask (sum(A), calc_trajectory.m:100)
  int *mc_list; int mc_list_cap;                     // floes whose forcing is due with an empty mask (1-based), C_NEED_MC of them
  // calc_trajectory values handed from integrate_pre to forcing / integrate_post (per real floe)
  double *st_ext; // [3 n] external force x, y and torque after the limiter (:42-46)
  int *st_flag;   // ST_GO | ST_FORCE
  // sharded narrow-phase (multi-GPU): contact rows of all pairs gathered from the ranks that computed them
  double *g_rows;    // null on one GPU: rows are read from the pair's scratch slice
  i64 *pair_rowoff;  // [cap_pairs] offset (in doubles) of pair p's rows inside g_rows (sharded step) or inside scratch
  // ---- the fused launch chain of the steady step (step_fused.hpp): tiles of 64 consecutive floes / entries ----
  int cell_cap;                          // > 0: cell c owns the slots [c * cell_cap, (c + 1) * cell_cap) of cell_item, filled by atomic append (no cell scan); 0: CSR by cell_off
  unsigned long long *gh_mask, *gh_agg;  // [2 * tiles] ballots of the x / y ghost flags of a tile's floes; [tiles] their counts packed x | y << 21 | both << 42
  unsigned long long *pair_agg;          // [groups of 4 entries] pairs (low 24 bits) and pair vertices (<< 24) of the group's entries (kf_broad: a workgroup each)
  unsigned long long *row_agg;           // [AGG_STRIDE * tiles of 64 entries] contact rows of a tile's entries (a 128-byte line each: bumped by atomics)
  unsigned long long *prow_part;         // [AGG_STRIDE * 64] C_PAIRROWS in 64 parts (pair p bumps part p & 63), a line each
  unsigned long long *rel_mask;          // [cap_ext] bit k: candidate k of the entry is a pair this rank evaluates (pair_relevant)
  int *pot_nvpre;                        // [cap_ext * pot_cap] ring vertices of the entry's pairs before candidate k
  int *hint;                             // [8 * cap_ext] the partners (j + 1, 0: none) of the entry's first eight own pairs in contact in the step before (kf_rows -> kf_pairs)
  int reg_rows;                          // 1: a pair registers its rows with both floes the moment it completes (pair_done) -- no k_incoming / k_row_count
  int *force_list;                       // [n] floes whose forcing is due though the step does not refresh it for all (C_FORCE of them)
};

// per-pair scratch slice: linear bound so that offsets follow from prefix sums (see carve())
enum { SCR_PER_V = 704, SCR_PER_PAIR = 4096 };

SZ_HD bool is_present(int s, const Work &W) { return !W.present || W.present[s] != 0; }
SZ_HD bool is_owned(int s, const Work &W) { return !W.owned || W.owned[s] != 0; }
SZ_HD double sgn(double v) { return (v > 0) ? 1.0 : (v < 0 ? -1.0 : 0.0); }
SZ_HD double absd(double v) { return v < 0 ? -v : v; }
SZ_HD int iabs(int v) { return v < 0 ? -v : v; }

#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline unsigned long long atomic_add(unsigned long long *p, unsigned long long v) { return atomicAdd(p, v); }
__device__ inline int atomic_add(int *p, int v) { return atomicAdd(p, v); }
__device__ inline int atomic_max(int *p, int v) { return atomicMax(p, v); }
__device__ inline unsigned long long atomic_max(unsigned long long *p, unsigned long long v) { return atomicMax(p, v); }
__device__ inline void atomic_inc_noret(int *p) { (void)__hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void atomic_add_noret(unsigned long long *p, unsigned long long v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
// (host build, and the host pass of hipcc, where __device__ functions that name these are parsed but never run)
SZ_HD unsigned long long atomic_add(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }
SZ_HD int atomic_add(int *p, int v) { int o = *p; *p += v; return o; }
SZ_HD int atomic_max(int *p, int v) { int o = *p; if (v > o) *p = v; return o; }
SZ_HD unsigned long long atomic_max(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; if (v > o) *p = v; return o; }
SZ_HD void atomic_inc_noret(int *p) { ++*p; }
SZ_HD void atomic_add_noret(unsigned long long *p, unsigned long long v) { *p += v; }
#endif

// ---------------------------------------------------------------------------------------------
// ghosts (:16-66)
// ---------------------------------------------------------------------------------------------
SZ_HD void ext_init(int i, const Floes &F, Work &W) { // i < n
  W.ext_src[i] = i; W.ext_x[i] = F.Xi[i]; W.ext_y[i] = F.Yi[i]; W.ext_num[i] = i + 1;
  W.ext_nv[i] = F.vert_off[i + 1] - F.vert_off[i];
}
// calc_trajectory.m:94-110: a floe whose forcing is due this step (doInt.flag, or thinner than 0.1 m after the
// thinning of :76-80) and whose Monte-Carlo mask is empty gets new points from MATLAB's rand.  The step must not
// guess them: such floes are listed BEFORE anything of the floe array is changed, szStep returns
// SZ_STEP_NEEDS_MC, the caller regenerates (szFloesUpdate) and steps again.  (alive is the value at the start of
// the step: a floe that leaves the box in this very step, :152-155, is listed although the reference would skip it.)
SZ_HD void mc_precheck(int i, const Floes &F, const Params &P, Work &W) { // nb <= i < n
  if (!F.alive[i] || F.Xi[i] != F.Xi[i] || W.mc_cnt[i] != 0 || !is_owned(i, W)) return;
  double hh = F.h[i];
  if (hh > 10) hh = 10;
  const double h_new = hh - P.HFo * P.dt / hh;
  if (!(P.doInt || h_new < 0.1)) return;
  unsigned long long k = atomic_add(&W.counters[C_NEED_MC], 1ULL);
  if (k < (unsigned long long)W.mc_list_cap) W.mc_list[k] = i + 1;
}
SZ_HD void mc_count(int i, const Floes &F, Work &W) { // i < n
  int c = 0;
  const unsigned char *A = F.mcA + i * F.mc_si;
  for (int q = 0; q < F.n_mc; ++q) c += A[q * F.mc_sq] ? 1 : 0;
  W.mc_cnt[i] = c;
}
SZ_HD void ext_clear(int e, Work &W) { // e < cap_ext
  W.xg_of[e] = -1; W.yg_of[e] = -1; W.pot_cnt[e] = 0; W.inc_cnt[e] = 0; W.tmax[e] = 0;
  W.pair_cnt[e] = 0; W.pair_nv[e] = 0; W.row_cnt[e] = 0; W.inc_pre[e] = 0;
  W.cf_ext[2 * e] = 0; W.cf_ext[2 * e + 1] = 0; W.ct_ext[e] = 0; W.ov_ext[e] = 0;
}
SZ_HD void marks_clear(int e, const Floes &F) { F.kill[e] = 0; F.transfer[e] = 0; } // e < 4 n (:123)
// alive and the translated ring reaches beyond the box in dimension dim (:31,:52); s = the entry's floe, c = its
// coordinate in that dimension
SZ_HD int ghost_flag_of(int s, double c, int dim, const Floes &F, const Params &P, const Work &W) {
  const int v0 = F.vert_off[s], nv = F.vert_off[s + 1] - v0;
  if (!is_present(s, W)) return 0;   // (sharded step: not a floe this rank knows this step)
  // no vertex is farther from the floe position than rmax (initialize_floe_values.m:26 defines it so, and the
  // rotation of c0 into c_alpha keeps the distances): a floe well inside the box needs no look at its ring
  // (the 1e-9 relative margin is far above the rounding of the rotated ring and of the sums below)
  const double L = (dim == 0) ? P.Lx : P.Ly;
  if (absd(c) + F.rmax[s] < L * (1 - 1e-9)) return 0;
  double m = 0;
  const double *ca = (dim == 0) ? F.cax : F.cay;
  for (int k = 0; k < nv; ++k) m = szc::maxd(m, absd(ca[v0 + k] + c));
  return (F.alive[s] && m > L) ? 1 : 0;
}
// flag[e] = that, for entry e
SZ_HD void ghost_flag(int e, int dim, const Floes &F, const Params &P, Work &W) {
  W.flag[e] = ghost_flag_of(W.ext_src[e], (dim == 0) ? W.ext_x[e] : W.ext_y[e], dim, F, P, W);
}
// e's ghost goes to slot base + scan[e] (:33-37, :54-58)
SZ_HD void ghost_emit(int e, int dim, int base, const Floes &F, const Params &P, Work &W) {
  if (!W.flag[e]) return;
  int g = base + W.scan[e];
  if (g >= W.cap_ext) return; // cannot happen: cap_ext = 4 n
  W.ext_src[g] = W.ext_src[e]; W.ext_nv[g] = W.ext_nv[e];
  W.ext_x[g] = (dim == 0) ? W.ext_x[e] - 2 * P.Lx * sgn(W.ext_x[e]) : W.ext_x[e];
  W.ext_y[g] = (dim == 1) ? W.ext_y[e] - 2 * P.Ly * sgn(W.ext_y[e]) : W.ext_y[e];
  W.ext_num[g] = -iabs(W.ext_num[e]);
  W.ext_parent[g] = e;
  if (dim == 0) W.xg_of[e] = g; else W.yg_of[e] = g;
}

// ---------------------------------------------------------------------------------------------
// broad-phase (:76-120): candidates of extended floe i among j > i, in ascending j
// ---------------------------------------------------------------------------------------------
SZ_HD void broad(int i, int N, const Floes &F, const Params &P, Work &W) {
  int cnt = 0;
  int *mine = W.pot_list + (size_t)i * W.pot_cap;
  const int si = W.ext_src[i];
  // mems starts as the candidate list of the floe a ghost was copied from (:94-99)
  const int *init = nullptr; int ninit = 0;
  if (W.ext_num[i] < 0) {
    int root = iabs(W.ext_num[i]) - 1;
    init = W.pot_list + (size_t)root * W.pot_cap;
    ninit = W.pot_cnt[root];
  }
  const double xi = W.ext_x[i], yi = W.ext_y[i], ri = F.rmax[si];
  const double lim = szc::mind(2 * P.Lx, 2 * P.Ly);
  if (F.alive[si] && !(xi != xi) && P.COLLISION) {
    for (int j = i + 1; j < N; ++j) {
      const int sj = W.ext_src[j];
      if (!F.alive[sj]) continue;
      double dx = xi - W.ext_x[j], dy = yi - W.ext_y[j];
      double R = ri + F.rmax[sj];
      double d2 = dx * dx + dy * dy;
      if (d2 > R * R * 1.000001) continue;      // cheap reject; the exact test follows
      if (!(::sqrt(d2) < R)) continue;          // :103
      int key = iabs(W.ext_num[j]);
      bool member = false;
      for (int q = 0; q < ninit && !member; ++q) member = (init[q] == key);
      for (int q = 0; q < cnt && !member; ++q) member = (W.ext_num[mine[q] - 1] == key); // mems grows by FloeNums(j) (:113)
      if (!(!member || 2 * R > lim)) continue;
      if (cnt >= W.pot_cap) { atomic_add(&W.counters[C_ERR_POT], 1ULL); break; }
      mine[cnt++] = j + 1;
    }
  }
  W.pot_cnt[i] = cnt;
}

// Same candidates through a uniform grid: every partner of i lies in the 3 x 3 cells around i because the
// cell edge is at least 2 * max(rmax) >= rmax(i) + rmax(j).  Out-of-range positions clamp to the border
// cells, which only adds tests.  Candidates are gathered unordered, sorted by j, then filtered in order.
SZ_HD int cell_coord(double v, double v0, double cs, int ncell) {
  double t = (v - v0) / cs;
  if (!(t > 0)) return 0;            // also NaN
  if (t >= (double)(ncell - 1)) return ncell - 1;
  return (int)t;
}
SZ_HD int level_of(double r, const Work &W) {
  int k = 0;
  while (k + 1 < W.n_levels && r > W.lvl_r[k]) ++k;
  return k;
}
SZ_HD int cell_at(int level, double x, double y, const Work &W) {
  return W.lvl_base[level] + cell_coord(y, W.lvl_y0[level], W.lvl_cs[level], W.lvl_cy[level]) * W.lvl_cx[level] +
         cell_coord(x, W.lvl_x0[level], W.lvl_cs[level], W.lvl_cx[level]);
}
// Levels for a field whose candidate floes have radii up to r_max, 90 % of them up to r_p90: the base level takes
// everything up to min(r_max, 2 r_p90) (one level when the sizes are alike: a Voronoi pack has r_max = 1.6 r_p90), every further level doubles the radius,
// the top level is open-ended.  Grids span [-L - cs, L + cs] (ghosts lie up to one radius outside the box).
SZ_HD void grid_levels(double r_max, double r_p90, double Lx, double Ly, Work &W) {
  if (!(r_max > 0)) r_max = 0.5;
  double r0 = 2.0 * r_p90;
  if (!(r0 > 0) || r0 > r_max) r0 = r_max;
  const int top = (int)Work::MAX_LEVELS - 1;
  double rt = r0;
  for (int k = 0; k < top; ++k) rt *= 2;
  if (rt < r_max) r0 = r0 * (r_max / rt);          // at most MAX_LEVELS levels: stretch the base instead
  int nl = 0, base = 0;
  double r = r0;
  for (;;) {
    const bool last = (r >= r_max) || nl == top;
    const double R = last ? (r > r_max ? r : r_max) : r;
    const double cs = 2 * R;
    W.lvl_r[nl] = R; W.lvl_cs[nl] = cs; W.lvl_x0[nl] = -Lx - cs; W.lvl_y0[nl] = -Ly - cs;
    const double nx = (2 * Lx + 2 * cs) / cs + 1, ny = (2 * Ly + 2 * cs) / cs + 1;
    W.lvl_cx[nl] = (nx > 2048 || !(nx > 0)) ? 2048 : (int)nx;
    W.lvl_cy[nl] = (ny > 2048 || !(ny > 0)) ? 2048 : (int)ny;
    W.lvl_base[nl] = base;
    base += W.lvl_cx[nl] * W.lvl_cy[nl];
    ++nl;
    if (last) break;
    r *= 2;
  }
  W.n_levels = nl; W.lvl_base[nl] = base;
  for (int k = nl; k < (int)Work::MAX_LEVELS; ++k) { W.lvl_r[k] = W.lvl_r[nl - 1]; W.lvl_cs[k] = 0; W.lvl_cx[k] = W.lvl_cy[k] = 0; W.lvl_base[k + 1] = base; W.lvl_x0[k] = W.lvl_y0[k] = 0; }
}
SZ_HD int cell_of(int e, const Floes &F, const Work &W) {
  return cell_at(level_of(F.rmax[W.ext_src[e]], W), W.ext_x[e], W.ext_y[e], W);
}
// only live floes that can be a candidate enter the grid (a dead floe never is, :101,:103; entries 1..Nb of the list
// neither, since j > i >= 1 + Nb -- but the GHOST of a boundary floe is an entry beyond N0 and is one, :76-103); each
// item carries its position and radius so that a cell is one contiguous stream of independent loads
SZ_HD bool in_grid(int e, const Floes &F, const Work &W) { const int s = W.ext_src[e]; return F.alive[s] && e >= F.nb && is_present(s, W); }
SZ_HD void cell_count(int e, const Floes &F, Work &W) {
  if (in_grid(e, F, W)) atomic_add(&W.cell_cnt[cell_of(e, F, W)], 1);
}
// (one thread per entry: the host build and small fields; the GPU uses k_ext_records, 16 lanes per entry, same values)
SZ_HD void ext_record(int e, const Floes &F, Work &W) {
  // what the narrow phase's team pass needs of the entry, once per step instead of once per pair: the ring where the
  // entry lies (ghosts: their own position) -- c_alpha + position, the sums every pair would otherwise form again --
  // its bounding box (contact.hpp ring_box), the scalars of floe_interactions.m:10-21,170-183
  const int s = W.ext_src[e];
  const int v0 = F.vert_off[s], n = F.vert_off[s + 1] - v0;
  const szc::RingD ring = {F.cax + v0, F.cay + v0, W.ext_x[e], W.ext_y[e], n, 0};
  szc::Box b;
  szc::ring_box(ring, b);
  int nstage = n;
  if (n == (int)RING_SLOTS + 1 && ring.X(n - 1) == ring.X(0) && ring.Y(n - 1) == ring.Y(0) &&
      szc::hyp(ring.X(0) - ring.X(n - 2), ring.Y(0) - ring.Y(n - 2)) > 1) --nstage;
  if (nstage > (int)RING_SLOTS || nstage < 1) nstage = -1;
  const double safe = SZX_FIXED_SAFE_METRES - 2;
  bool in_range = true;
  double *rx = W.ext_ring + 2 * (size_t)RING_SLOTS * e, *ry = rx + RING_SLOTS;
  for (int k = 0; k < nstage; ++k) {
    const double X = ring.X(k), Y = ring.Y(k);
    in_range = in_range && szc::absd(X) < safe && szc::absd(Y) < safe;
    rx[k] = X; ry[k] = Y;
  }
  double *o = W.ext_rec + (size_t)REC_WORDS * e;
  o[REC_X] = W.ext_x[e]; o[REC_Y] = W.ext_y[e]; o[REC_UI] = F.Ui[s]; o[REC_VI] = F.Vi[s]; o[REC_KSI] = F.ksi_ice[s];
  o[REC_H] = F.h[s]; o[REC_AREA] = F.area[s];
  o[REC_BX0] = b.x0; o[REC_BX1] = b.x1; o[REC_BY0] = b.y0; o[REC_BY1] = b.y1;
  o[REC_NSTAGE] = (double)nstage; o[REC_NSTORED] = (double)n; o[REC_INRANGE] = in_range ? 1.0 : 0.0;
}
SZ_HD void cell_fill(int e, const Floes &F, Work &W) {
  const int s = W.ext_src[e];
  if (!in_grid(e, F, W)) return;
  int c = cell_of(e, F, W);
  int k = W.cell_off[c] + atomic_add(&W.cell_fill[c], 1);
  CellItem it; it.x = W.ext_x[e]; it.y = W.ext_y[e]; it.r = F.rmax[s]; it.e = e; it.pad = 0;
  W.cell_item[k] = it;
}
// one more candidate of extended floe `owner`; the count keeps running past the capacity so that the host learns how
// much room the lists need (step_front grows them and runs the front stages again)
SZ_HD void pot_append(int owner, int partner, Work &W) {
  int k = atomic_add(&W.pot_cnt[owner], 1);
  if (k < W.pot_cap) W.pot_list[(size_t)owner * W.pot_cap + k] = partner + 1;
}
// broad_scan: thread (i, c9) tests floe i against the items of ONE of its 3 x 3 cells -- at its own level and at
// every coarser one -- and appends the hits (unordered, one atomic per hit; nine short loops run side by side -- a
// thread per row of three cells was measured slower at 1e4 floes, where this stage is latency-bound).  A pair of
// floes of one level is found from both sides and kept by the lower index; a pair across levels is found from the
// finer side only, which appends to whichever of the two has the lower index.  broad_filter: thread i sorts its list
// by j and applies the `mems` filter in order (:93-99, :103, :113); real floes first, ghosts afterwards (they read
// their root's list).
SZ_HD void broad_scan(int i, int c9, int N, const Floes &F, const Params &P, Work &W) {
  (void)N;
  const int si = W.ext_src[i];
  const double xi = W.ext_x[i], yi = W.ext_y[i], ri = F.rmax[si];
  if (!(F.alive[si] && !(xi != xi) && P.COLLISION && is_present(si, W))) return;
  const int own = level_of(ri, W);
  for (int b = own; b < W.n_levels; ++b) {
    const int gx = cell_coord(xi, W.lvl_x0[b], W.lvl_cs[b], W.lvl_cx[b]) + (c9 % 3) - 1;
    const int gy = cell_coord(yi, W.lvl_y0[b], W.lvl_cs[b], W.lvl_cy[b]) + (c9 / 3) - 1;
    if (gx < 0 || gx >= W.lvl_cx[b] || gy < 0 || gy >= W.lvl_cy[b]) continue;
    const int c = W.lvl_base[b] + gy * W.lvl_cx[b] + gx;
    for (int q = W.cell_off[c]; q < W.cell_off[c + 1]; ++q) {
      const CellItem it = W.cell_item[q];
      const int j = it.e;
      if (b == own ? j <= i : j == i) continue;
      double dx = xi - it.x, dy = yi - it.y;
      double R = ri + it.r;
      double d2 = dx * dx + dy * dy;
      if (d2 > R * R * 1.000001) continue;
      if (!(::sqrt(d2) < R)) continue;      // :103 (the expression is symmetric in i and j up to the sign of dx, dy)
      if (j > i) pot_append(i, j, W); else pot_append(j, i, W);
    }
  }
}
// ascending order in place (Shell sort, Ciura's gaps: n^1.3 for the few long lists, the plain insertion sort for the
// short ones the fields are full of)
SZ_HD void sort_ascending(int *v, int n) {
  const int gaps[8] = {701, 301, 132, 57, 23, 10, 4, 1};
  for (int g = 0; g < 8; ++g) {
    const int gap = gaps[g];
    if (gap >= n && gap > 1) continue;
    for (int a = gap; a < n; ++a) {
      int x = v[a], b = a;
      while (b >= gap && v[b - gap] > x) { v[b] = v[b - gap]; b -= gap; }
      v[b] = x;
    }
  }
}
// tmp: 2 * pot_cap ints of fast private memory (a slice of LDS on the GPU): the list is sorted and filtered
// there, so that the dependent accesses of the insertion sort do not go to HBM
SZ_HD void broad_filter(int i, const Floes &F, const Params &P, Work &W, int *tmp) {
  int *mine = W.pot_list + (size_t)i * W.pot_cap;
  int cnt = W.pot_cnt[i];
  if (cnt > W.pot_cap) {                   // the list is incomplete: the host grows the lists and runs the front again
    atomic_add(&W.counters[C_ERR_POT], 1ULL);
    atomic_max(&W.counters[C_NEED_POT], (unsigned long long)cnt);
    cnt = W.pot_cap;
  }
  int *num = tmp + W.pot_cap;              // ext_num of the entries of tmp
  for (int a = 0; a < cnt; ++a) tmp[a] = mine[a];
  sort_ascending(tmp, cnt);                // ascending j
  for (int a = 0; a < cnt; ++a) num[a] = W.ext_num[tmp[a] - 1];
  const int *init = nullptr; int ninit = 0;
  if (W.ext_num[i] < 0) {
    int root = iabs(W.ext_num[i]) - 1;
    init = W.pot_list + (size_t)root * W.pot_cap;
    ninit = W.pot_cnt[root];
  }
  const double ri = F.rmax[W.ext_src[i]];
  const double lim = szc::mind(2 * P.Lx, 2 * P.Ly);
  int kept = 0;
  if (!P.PERIODIC) kept = cnt;             // no ghosts: every FloeNum is its own, nothing is ever in `mems` twice (:113)
  for (int a = kept; a < cnt; ++a) {
    const int j = tmp[a] - 1;
    const int key = iabs(num[a]);
    bool member = false;
    for (int q = 0; q < ninit && !member; ++q) member = (init[q] == key);
    for (int q = 0; q < kept && !member; ++q) member = (num[q] == key);
    if (member) {   // (:113) a floe already listed through another image stays only if the pair spans the domain
      const double R = ri + F.rmax[W.ext_src[j]];
      if (!(2 * R > lim)) continue;
    }
    tmp[kept] = j + 1; num[kept] = num[a]; ++kept;
  }
  for (int a = 0; a < kept; ++a) mine[a] = tmp[a];
  W.pot_cnt[i] = kept;
}

// ---------------------------------------------------------------------------------------------
// flat pair list
// ---------------------------------------------------------------------------------------------
SZ_HD bool has_wall_pair(int i, int N, const Floes &F, const Params &P, const Work &W) {
  return !P.PERIODIC && i >= F.nb && i < N && is_present(W.ext_src[i], W) && is_owned(W.ext_src[i], W);
}
// (sharded step) a candidate pair is evaluated by the ranks that integrate one of its two floes -- both of them when
// the floes belong to different ranks: same inputs, same arithmetic, same rows on both sides, nothing to send back
SZ_HD bool pair_relevant(int i, int j, const Work &W) { return !W.owned || W.owned[W.ext_src[i]] || W.owned[W.ext_src[j]]; }

SZ_HD void pair_count(int i, int N, const Floes &F, const Params &P, Work &W) { // i < N
  int cnt = 0; i64 nv = 0;
  if (i >= F.nb) {
    const int ni = W.ext_nv[i];
    const int all = W.pot_cnt[i];
    const int *mine = W.pot_list + (size_t)i * W.pot_cap;
    for (int q = 0; q < all; ++q) {
      if (!pair_relevant(i, mine[q] - 1, W)) continue;
      ++cnt;
      nv += ni + W.ext_nv[mine[q] - 1];
      atomic_inc_noret(&W.inc_pre[mine[q] - 1]);   // an upper bound of the partner's mirror list (:187-214), see pair_room
    }
    if (has_wall_pair(i, N, F, P, W)) { ++cnt; nv += ni + 4; }
  }
  W.pair_cnt[i] = cnt; W.pair_nv[i] = nv;
}
// (after every pair_count) the mirror list of floe e cannot outgrow its capacity later in the step -- after
// calc_trajectory has begun to change the floes -- if the candidate pairs naming e fit it now
SZ_HD void pair_room(int e, Work &W) {
  if (W.inc_pre[e] > W.pot_cap) {
    atomic_add(&W.counters[C_ERR_POT], 1ULL);
    atomic_max(&W.counters[C_NEED_POT], (unsigned long long)W.inc_pre[e]);
  }
}
SZ_HD void pair_expand(int i, int N, const Floes &F, const Params &P, Work &W) { // after scans
  if (i < F.nb) return;
  i64 p = W.pair_base[i], nvo = W.nv_base[i];
  const int ni = W.ext_nv[i];
  const int *mine = W.pot_list + (size_t)i * W.pot_cap;
  for (int q = 0; q < W.pot_cnt[i]; ++q) {
    int j = mine[q] - 1;
    if (!pair_relevant(i, j, W)) continue;
    if (p >= W.cap_pairs) return;
    W.pair_i[p] = i; W.pair_j[p] = j; W.pair_nvoff[p] = nvo;
    nvo += ni + W.ext_nv[j];
    ++p;
  }
  if (has_wall_pair(i, N, F, P, W) && p < W.cap_pairs) {
    W.pair_i[p] = i; W.pair_j[p] = WALL; W.pair_nvoff[p] = nvo;
  }
}

// ---------------------------------------------------------------------------------------------
// narrow-phase: one candidate pair
// ---------------------------------------------------------------------------------------------
SZ_HD size_t up16(size_t v) { return (v + 15) & ~(size_t)15; }

// Everything of a PairScratch but the sweep workspace and the contact rows: the rings in fixed point, the two
// region slots, the crossing points (offsets from a 16-byte aligned base; `end` = bytes used).
struct SmallLayout { size_t f1x, f1y, f2x, f2y, fsx, fsy, r1x, r1y, r2x, r2y, r1off, r2off, px, py, end; };
SZ_HD SmallLayout small_layout(int n1, int n2, int cap_v, int cap_r, int cap_p) {
  SmallLayout L;
  size_t p = 0;
  L.f1x = p; p += up16(8 * (size_t)(n1 + 1));
  L.f1y = p; p += up16(8 * (size_t)(n1 + 1));
  L.f2x = p; p += up16(8 * (size_t)(n2 + 1));
  L.f2y = p; p += up16(8 * (size_t)(n2 + 1));
  L.fsx = p; p += up16(8 * (size_t)(n1 + 1));
  L.fsy = p; p += up16(8 * (size_t)(n1 + 1));
  L.r1x = p; p += up16(8 * (size_t)cap_v);
  L.r1y = p; p += up16(8 * (size_t)cap_v);
  L.r2x = p; p += up16(8 * (size_t)cap_v);
  L.r2y = p; p += up16(8 * (size_t)cap_v);
  L.r1off = p; p += up16(4 * (size_t)(cap_r + 1));
  L.r2off = p; p += up16(4 * (size_t)(cap_r + 1));
  L.px = p; p += up16(8 * (size_t)cap_p);
  L.py = p; p += up16(8 * (size_t)cap_p);
  L.end = p;
  return L;
}
SZ_HD void point_small(char *base, const SmallLayout &L, szc::PairScratch &S) {
  S.f1x = (i64 *)(base + L.f1x); S.f1y = (i64 *)(base + L.f1y); S.f2x = (i64 *)(base + L.f2x); S.f2y = (i64 *)(base + L.f2y);
  S.fsx = (i64 *)(base + L.fsx); S.fsy = (i64 *)(base + L.fsy);
  S.r1x = (i64 *)(base + L.r1x); S.r1y = (i64 *)(base + L.r1y); S.r2x = (i64 *)(base + L.r2x); S.r2y = (i64 *)(base + L.r2y);
  S.r1off = (int *)(base + L.r1off); S.r2off = (int *)(base + L.r2off);
  S.px = (double *)(base + L.px); S.py = (double *)(base + L.py);
}
// Lays the PairScratch out inside [base, base + bytes).  Returns false when it does not fit.
SZ_HD bool carve(char *base, i64 bytes, int n1, int n2, int mult, szc::PairScratch &S) {
  const int cap_nv = mult * (n1 + n2 + 2);
  char *p = base, *end = base + bytes;
  S.cap_nv = cap_nv;
  S.ws = p; p += up16((size_t)szv::WS_PER_VERTEX * cap_nv + szv::WS_PER_CLIP);
  S.cap_v = 2 * cap_nv + 8; S.cap_r = cap_nv / 2 + 4;
  S.cap_p = cap_nv + 8;
  const SmallLayout L = small_layout(n1, n2, S.cap_v, S.cap_r, S.cap_p);
  point_small(p, L, S); p += L.end;
  S.cap_rows = S.cap_r;
  S.rows = (double *)p; p += up16(40 * (size_t)S.cap_rows);
  S.wave = 0;
  S.task_slot = 0; S.task_n = 1;
  return p <= end;
}
// where carve() puts the contact rows of a slice (for a task that works in another slice: narrow_t)
SZ_HD double *slice_rows(char *base, int n1, int n2, int mult) {
  const int cap_nv = mult * (n1 + n2 + 2);
  char *p = base + up16((size_t)szv::WS_PER_VERTEX * cap_nv + szv::WS_PER_CLIP);
  const SmallLayout L = small_layout(n1, n2, 2 * cap_nv + 8, cap_nv / 2 + 4, cap_nv + 8);
  return (double *)(p + L.end);
}
SZ_HD i64 scratch_bytes_for(i64 nv, int mult) { return (i64)SCR_PER_V * mult * (nv + 2) + SCR_PER_PAIR; }

// The pair p = (i, j) has completed with nrows > 0 contact rows -- by the lane that stores the pair's result, whichever
// kernel that is.  Scan-based chain: the step's row count.  Fused chain: the count in 64 parts, both floes' row counts
// (per entry and per tile of 64 entries: kf_rows places the rows from them) and p on j's mirror list (:187-214, incoming).
SZ_HD void pair_done(i64 p, int i, int j, int nrows, Work &W) {
  if (!W.reg_rows) { atomic_add_noret(&W.counters[C_PAIRROWS], (unsigned long long)nrows); return; }
  atomic_add_noret(&W.prow_part[(size_t)AGG_STRIDE * (size_t)(p & 63)], (unsigned long long)nrows);
  (void)atomic_add(&W.row_cnt[i], nrows);
  atomic_add_noret(&W.row_agg[(size_t)AGG_STRIDE * (i >> 6)], (unsigned long long)nrows);
  if (j < 0) return;                                   // the wall: no partner floe
  const int k = atomic_add(&W.inc_cnt[j], 1);
  if (k >= W.pot_cap) { atomic_add(&W.counters[C_ERR_INC], 1ULL); return; }
  W.inc_list[(size_t)j * W.pot_cap + k] = (int)p;
  (void)atomic_add(&W.row_cnt[j], nrows);
  atomic_add_noret(&W.row_agg[(size_t)AGG_STRIDE * (j >> 6)], (unsigned long long)nrows);
}

// t-th entry of the work list (all pairs when list == nullptr, else the retry list at tier `mult`)
// (WAVE: all 64 lanes of the wave are here with the same arguments -- k_narrow_wave, mult 1 -- and run the pair alike:
// same reads, same plain stores; one lane speaks for the wave where a counter or a list is touched, and the loops that
// split over vertices -- staging, conversion to fixed point, trimming -- spread over the lanes)
// (slot / tasks: k_narrow_wave launches `tasks` waves per deferred pair; a pair that had h >= 2 contact rows in the step
// before -- as many overlap regions, each with a second clip of its own, the long part of such a pair -- is run by
// min(h, tasks) of them, each taking the regions PairScratch::task_slot names, every one in a scratch slice of its own
// but with its rows written to the pair's; the task that finishes last puts the pair's result together.  The other
// waves of a pair leave at once.)
template <bool CONVEX_ONLY, class Blk, bool LDS_WS = false, bool WAVE = false>
SZ_HD void narrow_t(i64 t, const int *list, int mult, const Floes &F, const Params &P, Work &W,
                    void *fast_ws, int fast_ws_bytes, Blk *fast_blk, int slot = 0, int tasks = 1) {
#if defined(__HIP_DEVICE_COMPILE__)
  const bool speaker = !WAVE || (threadIdx.x & 63) == 0;
  const int lane0 = WAVE ? (int)(threadIdx.x & 63) : 0, lanes = WAVE ? 64 : 1;
#else
  const bool speaker = true;
  const int lane0 = 0, lanes = 1;
#endif
  i64 p = list ? (i64)list[t] : t;
  int ntasks = 1;
  if (WAVE && tasks > 1 && W.pair_task) {
    const int h = W.pair_task[4 * p];
    ntasks = h < 1 ? 1 : (h > tasks ? tasks : h);
  }
  if (slot >= ntasks) return;
  const int i = W.pair_i[p], j = W.pair_j[p];
  const int si = W.ext_src[i];
  const int v1 = F.vert_off[si], n1 = F.vert_off[si + 1] - v1;
  szc::RingD ring1 = {F.cax + v1, F.cay + v1, W.ext_x[i], W.ext_y[i], n1, 0};
  szc::Body b1 = {W.ext_x[i], W.ext_y[i], F.Ui[si], F.Vi[si], F.ksi_ice[si], F.h[si], F.area[si]};
  szc::RingD ring2;
  szc::Body b2;
  double bx[4], by[4];
  bool boundary = (j == WALL);
  int n2;
  if (boundary) {
    for (int k = 0; k < 4; ++k) { bx[k] = P.c2_boundary[2 * k]; by[k] = P.c2_boundary[2 * k + 1]; }
    n2 = 4;
    szc::RingD r = {bx, by, 0.0, 0.0, 4, 0};
    ring2 = r;
    szc::Body b = {P.fb_Xi, P.fb_Yi, P.fb_Ui, P.fb_Vi, P.fb_ksi_ice, P.fb_h, P.fb_area};
    b2 = b;
  } else {
    const int sj = W.ext_src[j];
    const int v2 = F.vert_off[sj];
    n2 = F.vert_off[sj + 1] - v2;
    szc::RingD r = {F.cax + v2, F.cay + v2, W.ext_x[j], W.ext_y[j], n2, 0};
    ring2 = r;
    szc::Body b = {W.ext_x[j], W.ext_y[j], F.Ui[sj], F.Vi[sj], F.ksi_ice[sj], F.h[sj], F.area[sj]};
    b2 = b;
  }
  // scratch slice: closed form in the first pass, bump-allocated behind it for retries
  i64 off, bytes = scratch_bytes_for((i64)n1 + n2, mult);
  if (!list || mult == 1) off = (i64)SCR_PER_V * (W.pair_nvoff[p] + 2 * p) + (i64)SCR_PER_PAIR * p;  // (a deferred pair redone at 1x keeps its slice)
  else off = (i64)atomic_add(&W.counters[C_SCR_BUMP], (unsigned long long)bytes);
  const i64 off0 = off;                      // the pair's own slice: where its rows go, whichever task makes them
  bool arena_ok = true;
#if defined(__HIP_DEVICE_COMPILE__)
  if (WAVE && slot > 0) {                    // a task beyond the first works in a slice of its own, from behind the closed-form part
    unsigned long long o = 0;
    if (speaker) o = atomic_add(&W.counters[C_SCR_BUMP], (unsigned long long)bytes);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(o & 0xffffffffULL)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(o >> 32));
    off = (i64)(((unsigned long long)hi << 32) | lo);
  }
#endif
  szc::PairScratch S;
  if (ntasks == 1) {
    W.pair_scr[p] = off; W.pair_mult[p] = mult;
    W.pair_nrows[p] = 0; W.pair_flag[p] = 0;
    if (off + bytes > W.scratch_bytes || !carve(W.scratch + off, bytes, n1, n2, mult, S)) {
      W.pair_status[p] = szv::ST_WORKSPACE;
      if (speaker) atomic_add(&W.counters[C_ERR_ARENA], 1ULL);
      return;
    }
    W.pair_rowoff[p] = (i64)(S.rows - (const double *)W.scratch);   // where row_fill / export_pair find the rows
  } else {
    // (the pair's words are written once, by the task that finishes last: below)
    arena_ok = off + bytes <= W.scratch_bytes && off0 + bytes <= W.scratch_bytes && carve(W.scratch + off, bytes, n1, n2, mult, S);
    if (arena_ok) { S.rows = slice_rows(W.scratch + off0, n1, n2, mult); S.task_slot = slot; S.task_n = ntasks; }
  }
  // the sweep workspace may live in faster memory (LDS) when the caller provides a large enough block
  const bool ws_fits_fast = fast_ws && (i64)szv::WS_PER_VERTEX * S.cap_nv + szv::WS_PER_CLIP <= fast_ws_bytes;
  if (ws_fits_fast && !LDS_WS) S.ws = fast_ws;
  // k_narrow_lds: both rings (position added) move to the end of the lane's slice of LDS, where every loop of the
  // contact model over their vertices reads them at LDS latency; the sweep keeps the rest of the slice.
  // (x + -0.0 == x for every x, signed zeros included: the views below return the staged values unchanged)
  int lds_ws_bytes = fast_ws_bytes;
  if (LDS_WS && fast_ws) {
    const int ring_bytes = (int)up16(16 * (size_t)(n1 + n2));
    if (ring_bytes + (int)szv::WS_PER_CLIP + 8 * (int)szv::WS_PER_VERTEX <= fast_ws_bytes) {
      lds_ws_bytes = fast_ws_bytes - ring_bytes;
      double *lx1 = (double *)((char *)fast_ws + lds_ws_bytes), *ly1 = lx1 + n1, *lx2 = ly1 + n1, *ly2 = lx2 + n2;
#pragma unroll 4
      for (int k = lane0; k < n1; k += lanes) { lx1[k] = ring1.X(k); ly1[k] = ring1.Y(k); }
#pragma unroll 4
      for (int k = lane0; k < n2; k += lanes) { lx2[k] = ring2.X(k); ly2[k] = ring2.Y(k); }
#if defined(__HIP_DEVICE_COMPILE__)
      if (WAVE) __builtin_amdgcn_wave_barrier();
#endif
      szc::RingD s1 = {lx1, ly1, -0.0, -0.0, n1, 0}, s2 = {lx2, ly2, -0.0, -0.0, n2, 0};
      ring1 = s1; ring2 = s2;
      // ... and, while the sweep still keeps room for 24 vertices, the rings in fixed point, the region slots and
      // the crossing points (the trim, the conversion and every loop over a region then run at LDS latency too);
      // the contact rows stay in the pair's HBM slice, where the kernels after this one read them
      const SmallLayout L = small_layout(n1, n2, S.cap_v, S.cap_r, S.cap_p);
      if ((i64)L.end + (i64)szv::WS_PER_CLIP + 24 * (i64)szv::WS_PER_VERTEX <= (i64)lds_ws_bytes) {
        lds_ws_bytes -= (int)L.end;
        point_small((char *)fast_ws + lds_ws_bytes, L, S);
        S.wave = WAVE ? 1 : 0;
      }
    }
  }
  szc::PairOut out;
  out.nrows = 0; out.sentinel = 0; out.has_sentinel = 0;
  int st;
  SZC_PHASE(8);
  if (!arena_ok) st = szv::ST_WORKSPACE;
  else if (CONVEX_ONLY)
    st = szc::pair_contact_fast(ring1, b1, ring2, b2, boundary, P.c2_boundary, P.PERIODIC != 0, P.Modulus, P.dt,
                                P.mu, P.nu, S, out, fast_blk);
  else if (WAVE)
    st = szc::pair_contact_t<szc::ClipWideLdsWave>(ring1, b1, ring2, b2, boundary, P.c2_boundary, P.PERIODIC != 0, P.Modulus, P.dt,
                                                   P.mu, P.nu, S, out, (LDS_WS && fast_ws) ? fast_ws : nullptr, lds_ws_bytes);
  else
    st = szc::pair_contact(ring1, b1, ring2, b2, boundary, P.c2_boundary, P.PERIODIC != 0, P.Modulus, P.dt,
                           P.mu, P.nu, S, out, (LDS_WS && fast_ws) ? fast_ws : nullptr, lds_ws_bytes);
  SZC_PHASE(0);
#if defined(__HIP_DEVICE_COMPILE__)
  if (WAVE && ntasks > 1) {
    // The tasks of a pair meet here: each leaves its status (the one of the lowest task wins: the serial loop over the
    // regions would have met that failure first) and counts itself done behind a fence -- its rows are out; the one that
    // counts last speaks for the pair from here on, with every task's rows in the pair's slice.
    int *task = W.pair_task + 4 * p;
    int last = 0;
    if (speaker) {
      // (a task without room for a slice of its own reports ST_WORKSPACE: the pair is then redone whole at the next tier,
      // which reports an arena that is really full)
      if (st != szv::ST_OK) (void)atomicMax(&task[2], ((15 - slot) << 8) | (st & 0xff));
      __threadfence();
      last = atomicAdd(&task[1], 1) == ntasks - 1 ? 1 : 0;
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (!last) return;
    __threadfence();
    const int merged = __hip_atomic_load(&task[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    st = merged ? (merged & 0xff) : (int)szv::ST_OK;
    // (the words as kf_pairs left them: a pair that failed at this tier may be run by tasks again in the same step -- the
    // floes at the walls are swept beside the team pass and, when one of them outgrows its slice, once more with the
    // deferred pairs; found by tests/campaign_gpu.py: a second run that starts from a full counter never ends)
    if (speaker) { task[1] = 0; task[2] = 0; }
    W.pair_scr[p] = off0; W.pair_mult[p] = mult;
    W.pair_nrows[p] = 0; W.pair_flag[p] = 0;
    W.pair_rowoff[p] = (i64)(S.rows - (const double *)W.scratch);
  }
#endif
  W.pair_status[p] = st;
  if (st == szc::ST_DEFER_BIG) {
    if (speaker) { unsigned long long k = atomic_add(&W.counters[C_BIG], 1ULL); W.big_list[k] = (int)p; }
    return;
  }
  if (st == szv::ST_WORKSPACE || st == szv::ST_OUT_CAPACITY || st == szc::ST_DEFER) {
    if (speaker) { unsigned long long k = atomic_add(&W.counters[C_RETRY], 1ULL); W.retry_list[k] = (int)p; }
    return;
  }
  if (st != szv::ST_OK) { if (speaker) atomic_add(&W.counters[C_ERR_CLIP], 1ULL); return; }
  int flag = 0;
  double sabs = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  if (WAVE && ntasks > 1) {     // (rows of other waves: read past this CU's cache)
    for (int r = 0; r < out.nrows; ++r) {
      const double a = __hip_atomic_load(&S.rows[5 * r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const double b = __hip_atomic_load(&S.rows[5 * r + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sabs += absd(a) + absd(b);
    }
  } else
#endif
  for (int r = 0; r < out.nrows; ++r) sabs += absd(S.rows[5 * r]) + absd(S.rows[5 * r + 1]);
  if (sabs != 0) flag |= 4;                                       // :135 / :157
  else if (out.has_sentinel && szc::is_inf(out.sentinel)) flag |= (out.sentinel > 0) ? 1 : 2;
  W.pair_nrows[p] = (flag & 4) ? out.nrows : 0;
  W.pair_flag[p] = flag;
  if ((flag & 4) && speaker) pair_done(p, i, j, out.nrows, W);
}
// the general sweep for every clip of the pair
SZ_HD void narrow(i64 t, const int *list, int mult, const Floes &F, const Params &P, Work &W,
                  void *fast_ws = nullptr, int fast_ws_bytes = 0) {
  narrow_t<false, szx::FastBlock>(t, list, mult, F, P, W, fast_ws, fast_ws_bytes, nullptr);
}
// the general sweep with its workspace in the lane's slice of LDS when that is large enough (k_narrow_lds)
SZ_HD void narrow_lds(i64 t, const int *list, int mult, const Floes &F, const Params &P, Work &W, void *lds_ws, int lds_stride) {
  narrow_t<false, szx::FastBlock, true>(t, list, mult, F, P, W, lds_ws, lds_stride, nullptr);
}
// ... by a whole wave per pair (k_narrow_wave; list at the 1x tier only: no allocation from the bump counter)
SZ_HD void narrow_wave(i64 t, const int *list, const Floes &F, const Params &P, Work &W, void *lds_ws, int lds_bytes, int slot = 0, int tasks = 1) {
  narrow_t<false, szx::FastBlock, true, true>(t, list, 1, F, P, W, lds_ws, lds_bytes, nullptr, slot, tasks);
}
// the single-minimum fast path only (convex.hpp); pairs it does not cover go to the retry list for narrow()
template <class Blk>
SZ_HD void narrow_convex(i64 p, const int *list, const Floes &F, const Params &P, Work &W, Blk *blk) {
  narrow_t<true, Blk>(p, list, 1, F, P, W, nullptr, 0, blk);
}
// Sorting the candidate pairs before the narrow phase, a thread per pair (the first questions of contact.hpp
// pair_contact_fast, answered from the entries' records): a pair whose bounding boxes are apart -- a quarter of the
// candidates of a packed field -- or whose floe lies well inside the domain box is DONE here with the no-contact
// result; the others are appended to the list of the kernel that takes them, so that the team pass runs on dense teams:
//   live_list  16 lanes per pair          wide_list  32 lanes per pair (more than 16 scanlines)
//   cold_list  16 lanes per pair, the pairs that were NOT in contact in the step before: a wave of four pairs takes as
//              long as its longest, and a pair in contact runs the whole model while one that is not stops after the
//              first clip -- floes move metres per step, so last step's contact rows (still in W.rows / W.row_off when
//              this runs; prev_n = entries of last step's extended list, 0: none) say which is which nearly always,
//              and waves of like pairs waste no lanes.  A hint only: a wrong guess costs time, never a result.
//   team_list  the lane-per-pair kernels (walls, rings too long for the block, coordinates out of the fast range)
// The order inside a list is arbitrary (every pair writes its own words).  Returns the class; the caller appends.
enum { PC_DONE = 0, PC_LIVE = 1, PC_WIDE = 2, PC_OTHER = 3, PC_COLD = 4, PC_GENERAL = 5 };   // PC_GENERAL: straight to the general sweep (a floe at a wall: the box has horizontal edges, no fast path takes it)
SZ_HD int pair_classify_ij(i64 p, int i, int j, i64 nvoff, int prev_n, const Params &P, Work &W, int live_hint = -1);
SZ_HD int pair_classify(i64 p, int prev_n, const Params &P, Work &W) {
  return pair_classify_ij(p, W.pair_i[p], W.pair_j[p], W.pair_nvoff[p], prev_n, P, W);
}
// (the pair's words passed in: the fused chain classifies a pair in the thread that has just written them; live_hint
// 0 / 1: was the pair in contact in the step before, looked up by the caller -- -1: look it up in that step's rows)
SZ_HD int pair_classify_ij(i64 p, int i, int j, i64 nvoff, int prev_n, const Params &P, Work &W, int live_hint) {
  const bool boundary = (j == WALL);
  const double *ri = W.ext_rec + (size_t)REC_WORDS * i, *rj = boundary ? ri : W.ext_rec + (size_t)REC_WORDS * j;
  const int n1 = (int)ri[REC_NSTORED], n2 = boundary ? 4 : (int)rj[REC_NSTORED];
  const i64 off = (i64)SCR_PER_V * (nvoff + 2 * p) + (i64)SCR_PER_PAIR * p;
  W.pair_scr[p] = off; W.pair_mult[p] = 1; W.pair_nrows[p] = 0; W.pair_flag[p] = 0; W.pair_rowoff[p] = off / 8;
  if (off + scratch_bytes_for((i64)n1 + n2, 1) > W.scratch_bytes) { W.pair_status[p] = szv::ST_WORKSPACE; atomic_add(&W.counters[C_ERR_ARENA], 1ULL); return PC_DONE; }
  W.pair_status[p] = szv::ST_OK;
  const szc::Box b1 = {ri[REC_BX0], ri[REC_BX1], ri[REC_BY0], ri[REC_BY1]};
  if (boundary) {
    const double gap = 1e-6;
    if (n1 > 0 && P.bnd_rect && b1.x0 > P.bnd_xmin + gap && b1.x1 < P.bnd_xmax - gap && b1.y0 > P.bnd_ymin + gap && b1.y1 < P.bnd_ymax - gap) return PC_DONE;
    W.pair_status[p] = szc::ST_DEFER;
    return PC_GENERAL;
  }
  const szc::Box b2 = {rj[REC_BX0], rj[REC_BX1], rj[REC_BY0], rj[REC_BY1]};
  const bool apart = szc::boxes_apart(b1, b2);
  if (ri[REC_NSTAGE] < 1 || rj[REC_NSTAGE] < 1) return (n1 > 0 && n2 > 0 && apart) ? PC_DONE : PC_OTHER;
  if (ri[REC_INRANGE] == 0 || rj[REC_INRANGE] == 0) return PC_OTHER;
  if (apart) return PC_DONE;
  if (n1 + n2 - 2 > 16) return PC_WIDE;
  if (live_hint >= 0) return live_hint ? PC_LIVE : PC_COLD;
  if (i < prev_n && W.rows != nullptr) {
    const i64 r0 = W.row_off[i];
    i64 r1 = W.row_off[i + 1];
    if (r1 > r0 + 64) r1 = r0 + 64;
    if (r1 > W.cap_rows) r1 = W.cap_rows;
    const double want = (double)(j + 1);
    bool live = false;                      // (no early exit: the loads of the rows then overlap)
#pragma unroll 4
    for (i64 r = r0; r < r1; ++r) live |= (W.rows[7 * r] == want);
    if (live) return PC_LIVE;
  }
  return PC_COLD;
}
SZ_HD void pair_classified(i64 p, int cls, Work &W) {   // (the kernel appends a wave at a time instead)
  if (cls == PC_LIVE) W.live_list[atomic_add(&W.counters[C_LIVE], 1ULL)] = (int)p;
  else if (cls == PC_COLD) W.cold_list[atomic_add(&W.counters[C_COLD], 1ULL)] = (int)p;
  else if (cls == PC_WIDE) W.wide_list[atomic_add(&W.counters[C_WIDE], 1ULL)] = (int)p;
  else if (cls == PC_OTHER) W.team_list[atomic_add(&W.counters[C_TEAM], 1ULL)] = (int)p;
  else if (cls == PC_GENERAL) W.retry_list[atomic_add(&W.counters[C_RETRY], 1ULL)] = (int)p;
}

// The pair by a TEAM of lanes (teamcontact.hpp): the first pass of the default path.  Two dependent rounds of loads:
// the pair (i, j), then -- one coalesced load each -- the two entries' records and staged rings (cell_count).  Only the
// contact row and the per-pair result words go to HBM, written by one lane; the row sits at the start of the pair's
// slice of the scratch arena (closed form, as in narrow_t; a pair handed on is laid out afresh by the kernel that redoes
// it).  What the team form does not cover goes to `team_list` and is redone by the lane-per-pair fast path
// (narrow_convex), whose own leftovers take the general sweep as before; a pair with more vertices than the team has
// lanes goes to `wide_list` (the same pass with 32 lanes per pair).
template <int G>
SZ_TEAM void narrow_team(i64 p, const sztm::Team<G> &T, sztm::TeamBlock *B, const Floes &F, const Params &P, Work &W, bool cold = false, bool prepared = false) {
  SZTM_PHASE(0);
  const int i = W.pair_i[p], j = W.pair_j[p];
  const bool boundary = (j == WALL);
  // lane l: word l of the records, vertex l of the rings
  {
    const int l = T.lane & 15, half = T.lane >> 4;               // (32-lane teams: the upper half loads the partner's)
    const double *ri = W.ext_rec + (size_t)REC_WORDS * i, *rj = boundary ? ri : W.ext_rec + (size_t)REC_WORDS * j;
    const double *gi = W.ext_ring + 2 * (size_t)RING_SLOTS * i, *gj = boundary ? gi : W.ext_ring + 2 * (size_t)RING_SLOTS * j;
    if (G == 16) {
      const double a = ri[l], b = rj[l], xi = gi[l], yi = gi[RING_SLOTS + l], xj = gj[l], yj = gj[RING_SLOTS + l];
      B->u.rec[0][l] = a; B->u.rec[1][l] = b; B->mx[0][l] = xi; B->my[0][l] = yi; B->mx[1][l] = xj; B->my[1][l] = yj;
    } else {
      const double a = (half ? rj : ri)[l], x = (half ? gj : gi)[l], y = (half ? gj : gi)[RING_SLOTS + l];
      B->u.rec[half][l] = a; B->mx[half][l] = x; B->my[half][l] = y;
    }
  }
  T.sync(TEAM_SITE);
  const int n1 = (int)B->u.rec[0][REC_NSTORED], n2 = boundary ? 4 : (int)B->u.rec[1][REC_NSTORED];
  // (prepared: pair_classify has written the pair's words -- slice, no rows, status OK -- and checked the arena)
  if (!prepared) {
    const i64 nvoff = W.pair_nvoff[p];
    const i64 bytes = scratch_bytes_for((i64)n1 + n2, 1);
    const i64 off = (i64)SCR_PER_V * (nvoff + 2 * p) + (i64)SCR_PER_PAIR * p;
    if (T.lane == 0) {
      W.pair_scr[p] = off; W.pair_mult[p] = 1;
      W.pair_nrows[p] = 0; W.pair_flag[p] = 0;
      W.pair_rowoff[p] = off / 8;
    }
    if (!(off + bytes <= W.scratch_bytes)) {
      if (T.lane == 0) { W.pair_status[p] = szv::ST_WORKSPACE; atomic_add(&W.counters[C_ERR_ARENA], 1ULL); }
      return;
    }
  }
  const szc::Box bx1 = {B->u.rec[0][REC_BX0], B->u.rec[0][REC_BX1], B->u.rec[0][REC_BY0], B->u.rec[0][REC_BY1]};
  const szc::Box bx2 = {B->u.rec[1][REC_BX0], B->u.rec[1][REC_BX1], B->u.rec[1][REC_BY0], B->u.rec[1][REC_BY1]};
  sztm::Staged sg;
  sg.n1 = (int)B->u.rec[0][REC_NSTAGE]; sg.n2 = (int)B->u.rec[1][REC_NSTAGE];
  sg.stored1 = n1; sg.stored2 = n2;
  sg.in_range = B->u.rec[0][REC_INRANGE] != 0 && B->u.rec[1][REC_INRANGE] != 0;
  sg.cold = cold;
  // the bodies (contact.hpp Body) stay in the block: words REC_X .. REC_AREA of the records, in that order
  static_assert((int)REC_X == (int)sztm::BODY_X && (int)REC_AREA == (int)sztm::BODY_AREA && (int)REC_KSI == (int)sztm::BODY_KSI, "record and body word order");
  static_assert((int)sztm::BODY_WORDS == (int)sztm::BODY_AREA + 2, "the spare word of a body holds the entry's index");
  if (T.lane < (int)sztm::BODY_WORDS) {
    const int w = T.lane;
    // (a pair with the domain box never gets as far as its second body here: it is answered from the floe's bounding
    // box or handed to the general sweep.  The spare word: the entries i, j -- whatever every lane of the team holds
    // alike costs each of them a register for the whole model, so the tail below reads them back from the block)
    const bool spare = w == (int)sztm::BODY_WORDS - 1;
    B->body[0][w] = spare ? (double)i : B->u.rec[0][w]; B->body[1][w] = spare ? (double)j : B->u.rec[1][w];
  }
  T.sync(TEAM_SITE);
  szc::PairOut out;
  double row[5];                         // (written when out.nrows == 1)
  sztm::BoundaryBox bb = {P.bnd_xmax, P.bnd_xmin, P.bnd_ymax, P.bnd_ymin, P.bnd_area, P.bnd_rect != 0};
  const int st = sztm::pair_contact_team(T, B, sg, bx1, bx2, boundary, bb, P.PERIODIC != 0, P.Modulus, P.dt, P.mu, P.nu, row, out);
  if (T.lane != 0) return;
  const int i2 = (int)B->body[0][sztm::BODY_WORDS - 1], j2 = (int)B->body[1][sztm::BODY_WORDS - 1];
  if (!prepared || st != szv::ST_OK) W.pair_status[p] = st;
  if (st == szc::ST_DEFER_BIG) { const unsigned long long k = atomic_add(&W.counters[C_BIG], 1ULL); W.big_list[k] = (int)p; return; }
  if (st == szc::ST_DEFER || st == sztm::ST_DEFER_WIDE || st == szv::ST_WORKSPACE || st == szv::ST_OUT_CAPACITY) {
    const unsigned long long k = atomic_add(&W.counters[C_TEAM], 1ULL);
    W.team_list[k] = (int)p;
    return;
  }
  if (st != szv::ST_OK) { atomic_add(&W.counters[C_ERR_CLIP], 1ULL); return; }
  int flag = 0;
  double sabs = 0;
  if (out.nrows > 0) sabs += absd(row[0]) + absd(row[1]);
  if (sabs != 0) flag |= 4;                                       // :135 / :157
  else if (out.has_sentinel && szc::is_inf(out.sentinel)) flag |= (out.sentinel > 0) ? 1 : 2;
  if (flag & 4) {
    const i64 off2 = (i64)SCR_PER_V * (W.pair_nvoff[p] + 2 * p) + (i64)SCR_PER_PAIR * p;   // (`off` again, from the pair's own word)
    double *rows = (double *)(W.scratch + off2);
    rows[0] = row[0]; rows[1] = row[1]; rows[2] = row[2]; rows[3] = row[3]; rows[4] = row[4];
    W.pair_nrows[p] = out.nrows;
    pair_done(p, i2, j2, out.nrows, W);
  }
  if (flag) W.pair_flag[p] = flag;
}
// ---- two-stage fast path: compact sweep records in a block of LDS (`fast_ws`), first clip for every pair
// (narrow_a), the rest only for the pairs that overlap (narrow_b).  Pairs that do not fit the compact tier,
// or whose lists overflow it, go to the retry list and take the general path (narrow) from scratch.
struct PairSetup {
  szc::RingD ring1, ring2;
  szc::Body b1, b2;
  double bx[4], by[4];
  bool boundary;
  int n1, n2;
};
SZ_HD void pair_setup(i64 p, const Floes &F, const Params &P, const Work &W, PairSetup &u) {
  const int i = W.pair_i[p], j = W.pair_j[p];
  const int si = W.ext_src[i];
  const int v1 = F.vert_off[si];
  u.n1 = F.vert_off[si + 1] - v1;
  szc::RingD r1 = {F.cax + v1, F.cay + v1, W.ext_x[i], W.ext_y[i], u.n1, 0};
  u.ring1 = r1;
  szc::Body b1 = {W.ext_x[i], W.ext_y[i], F.Ui[si], F.Vi[si], F.ksi_ice[si], F.h[si], F.area[si]};
  u.b1 = b1;
  u.boundary = (j == WALL);
  if (u.boundary) {
    for (int k = 0; k < 4; ++k) { u.bx[k] = P.c2_boundary[2 * k]; u.by[k] = P.c2_boundary[2 * k + 1]; }
    u.n2 = 4;
    szc::RingD r = {u.bx, u.by, 0.0, 0.0, 4, 0};
    u.ring2 = r;
    szc::Body b = {P.fb_Xi, P.fb_Yi, P.fb_Ui, P.fb_Vi, P.fb_ksi_ice, P.fb_h, P.fb_area};
    u.b2 = b;
  } else {
    const int sj = W.ext_src[j];
    const int v2 = F.vert_off[sj];
    u.n2 = F.vert_off[sj + 1] - v2;
    szc::RingD r = {F.cax + v2, F.cay + v2, W.ext_x[j], W.ext_y[j], u.n2, 0};
    u.ring2 = r;
    szc::Body b = {W.ext_x[j], W.ext_y[j], F.Ui[sj], F.Vi[sj], F.ksi_ice[sj], F.h[sj], F.area[sj]};
    u.b2 = b;
  }
}
SZ_HD void to_general_path(i64 p, Work &W) {
  unsigned long long k = atomic_add(&W.counters[C_RETRY], 1ULL);
  W.retry_list[k] = (int)p;
  W.pair_status[p] = szv::ST_WORKSPACE;
}
SZ_HD bool pair_slice(i64 p, const PairSetup &u, Work &W, szc::PairScratch &S) {
  i64 bytes = scratch_bytes_for((i64)u.n1 + u.n2, 1);
  i64 off = (i64)SCR_PER_V * (W.pair_nvoff[p] + 2 * p) + (i64)SCR_PER_PAIR * p;
  W.pair_scr[p] = off; W.pair_mult[p] = 1;
  if (off + bytes > W.scratch_bytes || !carve(W.scratch + off, bytes, u.n1, u.n2, 1, S)) return false;
  W.pair_rowoff[p] = (i64)(S.rows - (const double *)W.scratch);
  return true;
}
SZ_HD void narrow_a(i64 p, const Floes &F, const Params &P, Work &W, void *fast_ws) {
  PairSetup u;
  pair_setup(p, F, P, W, u);
  W.pair_nrows[p] = 0; W.pair_flag[p] = 0; W.pair_nreg[p] = 0;
  szc::PairScratch S;
  if (!pair_slice(p, u, W, S)) { W.pair_status[p] = szv::ST_WORKSPACE; atomic_add(&W.counters[C_ERR_ARENA], 1ULL); return; }
  if (u.n1 + u.n2 > szv::COMPACT_MAX_NV) { to_general_path(p, W); return; }
  szc::ViaArrays<szc::ClipCompact> clip = {{fast_ws}};
  int nreg = 0;
  int st = szc::pair_clip1(u.ring1, u.ring2, u.boundary, S, clip, nreg);
  if (st == szv::ST_WORKSPACE || st == szv::ST_OUT_CAPACITY) { to_general_path(p, W); return; }
  W.pair_status[p] = st;
  if (st != szv::ST_OK) { atomic_add(&W.counters[C_ERR_CLIP], 1ULL); return; }
  W.pair_nreg[p] = nreg;
  if (nreg == 0) return; // no overlap region: floe_interactions.m:71-74 returns the zero sentinel
  unsigned long long k = atomic_add(&W.counters[C_TOUCH], 1ULL);
  W.touch_list[k] = (int)p;
}
SZ_HD void narrow_b(i64 t, const Floes &F, const Params &P, Work &W, void *fast_ws) {
  const i64 p = W.touch_list[t];
  PairSetup u;
  pair_setup(p, F, P, W, u);
  szc::PairScratch S;
  pair_slice(p, u, W, S); // same slice as in narrow_a: regions of the first clip are in S.r1*
  szc::ViaArrays<szc::ClipCompact> clip = {{fast_ws}};
  szc::PairOut out;
  int st = szc::pair_finish(u.ring1, u.b1, u.ring2, u.b2, u.boundary, P.c2_boundary, P.PERIODIC != 0, P.Modulus, P.dt,
                            P.mu, P.nu, S, W.pair_nreg[p], clip, out);
  if (st == szv::ST_WORKSPACE || st == szv::ST_OUT_CAPACITY) { to_general_path(p, W); return; }
  W.pair_status[p] = st;
  if (st != szv::ST_OK) { atomic_add(&W.counters[C_ERR_CLIP], 1ULL); return; }
  int flag = 0;
  double sabs = 0;
  for (int r = 0; r < out.nrows; ++r) sabs += absd(S.rows[5 * r]) + absd(S.rows[5 * r + 1]);
  if (sabs != 0) flag |= 4;
  else if (out.has_sentinel && szc::is_inf(out.sentinel)) flag |= (out.sentinel > 0) ? 1 : 2;
  W.pair_nrows[p] = (flag & 4) ? out.nrows : 0;
  W.pair_flag[p] = flag;
  if (flag & 4) pair_done(p, W.pair_i[p], W.pair_j[p], out.nrows, W);
}

// contact rows of pair p: in the pair's scratch slice, or in the gathered exchange buffer (sharded step)
SZ_HD const double *pair_rows(i64 p, const Floes &, const Work &W) {
  return (W.g_rows ? W.g_rows : (const double *)W.scratch) + W.pair_rowoff[p];
}

// ---------------------------------------------------------------------------------------------
// mirrored rows: pair p = (i, j) registers with j (:187-214)
// ---------------------------------------------------------------------------------------------
SZ_HD void incoming(i64 p, Work &W) {
  if (!(W.pair_flag[p] & 4)) return;
  int j = W.pair_j[p];
  if (j == WALL) return;
  int k = atomic_add(&W.inc_cnt[j], 1);
  if (k >= W.pot_cap) { atomic_add(&W.counters[C_ERR_INC], 1ULL); return; }
  W.inc_list[(size_t)j * W.pot_cap + k] = (int)p;
}

// tmp: 2 * pot_cap ints of fast private memory, as for broad_filter
SZ_HD void row_count(int i, Work &W, int *tmp) { // i < N
  int cnt = 0;
  for (i64 p = W.pair_base[i]; p < W.pair_base[i] + W.pair_cnt[i]; ++p) cnt += W.pair_nrows[p];
  int ninc = W.inc_cnt[i] < W.pot_cap ? W.inc_cnt[i] : W.pot_cap;
  int *inc = W.inc_list + (size_t)i * W.pot_cap;
  int *src = tmp + W.pot_cap;              // source floe of the entries of tmp
  for (int a = 0; a < ninc; ++a) tmp[a] = inc[a];
  for (int a = 0; a < ninc; ++a) src[a] = W.pair_i[tmp[a]];
  // mirrored rows come in ascending order of the source floe (:187)
  for (int a = 1; a < ninc; ++a) {
    int v = tmp[a], sv = src[a], b = a;
    while (b > 0 && src[b - 1] > sv) { tmp[b] = tmp[b - 1]; src[b] = src[b - 1]; --b; }
    tmp[b] = v; src[b] = sv;
  }
  for (int a = 0; a < ninc; ++a) { inc[a] = tmp[a]; cnt += W.pair_nrows[tmp[a]]; }
  W.row_cnt[i] = cnt;
}

// floe centre outside the box (:152-155): the floe is dead from here on
SZ_HD void centre_check(int i, double x, double y, const Floes &F, const Params &P, Work &W) {
  const int N0 = F.n;
  if (!P.PERIODIC && i >= F.nb && i < N0 && is_present(i, W) && is_owned(i, W)) {
    const double *c = P.c2_boundary;
    // inpolygon of one point against the 5-point closed box, inpolygon.m:150-224
    double mnx = c[0], mxx = c[0], mny = c[1], mxy = c[1];
    for (int k = 1; k < 5; ++k) { mnx = szc::mind(mnx, c[2 * k]); mxx = szc::maxd(mxx, c[2 * k]); mny = szc::mind(mny, c[2 * k + 1]); mxy = szc::maxd(mxy, c[2 * k + 1]); }
    bool in = false;
    if (x >= mnx && x <= mxx && y >= mny && y <= mxy) {
      double sumdq = 0; bool on = false;
      for (int m = 0; m < 4; ++m) {
        double xa = c[2 * m], ya = c[2 * m + 1], xb = c[2 * m + 2], yb = c[2 * m + 3];
        double avx = absd(0.5 * (xa + xb)), avy = absd(0.5 * (ya + yb));
        double sf = szc::maxd(avx, avy); sf = szc::maxd(sf, avx * avy);
        double ax = xa - x, ay = ya - y, bx = xb - x, by = yb - y;
        int qa, qb;
        { bool px = ax > 0, py = ay > 0; qa = (!px && py) ? 1 : (!px && !py) ? 2 : (px && !py) ? 3 : 0; }
        { bool px = bx > 0, py = by > 0; qb = (!px && py) ? 1 : (!px && !py) ? 2 : (px && !py) ? 3 : 0; }
        double cp = ax * by - bx * ay;
        double scp = (cp > 0) ? 1 : (cp < 0 ? -1 : 0);
        if (absd(cp) < sf * SZC_EPS * 3) scp = 0;
        double dp = ax * bx + ay * by;
        double dq = qb - qa;
        if (absd(dq) == 3) dq = -dq / 3;
        if (absd(dq) == 2) dq = 2 * scp;
        sumdq += dq;
        if (scp == 0 && dp <= 0) on = true;
      }
      in = (sumdq != 0) || on;
    }
    if (!in) F.alive[i] = 0;
  }
}

// rows of extended floe i in the reference's order, torques and sums; kill/transfer marks of its pairs
SZ_HD void row_fill(int i, int N, const Floes &F, const Params &P, Work &W) {
  const int N0 = F.n;
  i64 r = W.row_off[i];
  double *rows = W.rows;
  double ov = 0;
  // torque about the floe position of this (possibly ghost) entry and the sums, as the rows are written: the same
  // values in the same (row) order as a pass over the finished rows (:221-239, :251-263)
  const double x = W.ext_x[i], y = W.ext_y[i];
  double sfx = 0, sfy = 0, stq = 0;
  // own pairs, ascending partner, wall last (:129-172)
  for (i64 p = W.pair_base[i]; p < W.pair_base[i] + W.pair_cnt[i]; ++p) {
    const int j = W.pair_j[p];
    const int fl = W.pair_flag[p];
    if (fl & 4) {
      const double *pr = pair_rows(p, F, W);
      double so = 0;
      for (int k = 0; k < W.pair_nrows[p]; ++k, ++r) {
        if (r >= W.cap_rows) return;
        double *w = rows + 7 * r;
        double fx = pr[5 * k], fy = pr[5 * k + 1];
        if (j == WALL) {
          if (absd(pr[5 * k + 3]) == P.Ly) fx = 0;            // :160-162
          if (absd(pr[5 * k + 2]) == P.Lx) fy = 0;            // :163-165
          w[0] = SZC_INF;
        } else w[0] = (double)(j + 1);
        const double px = pr[5 * k + 2], py = pr[5 * k + 3];
        const double tq = (px - x) * fy - (py - y) * fx;       // :221-239 / :251-263, in row order
        w[1] = fx; w[2] = fy; w[3] = px; w[4] = py; w[5] = tq; w[6] = pr[5 * k + 4];
        sfx += fx; sfy += fy; stq += tq;
        so += pr[5 * k + 4];
      }
      ov = so + ov;                                           // :137 / :168
    } else if ((fl & 3) && j != WALL && i + 1 > F.nb) {       // :138-145
      if (i + 1 <= N0 && (fl & 1)) { F.kill[i] = i + 1; F.transfer[i] = j + 1; }
      else if (j + 1 <= N0) F.kill[i] = j + 1;
    }
  }
  // mirrored rows, ascending source floe (:187-214)
  int ninc = W.inc_cnt[i] < W.pot_cap ? W.inc_cnt[i] : W.pot_cap;
  const int *inc = W.inc_list + (size_t)i * W.pot_cap;
  for (int a = 0; a < ninc; ++a) {
    const i64 p = inc[a];
    const double *pr = pair_rows(p, F, W);
    const int src = W.pair_i[p];
    for (int k = 0; k < W.pair_nrows[p]; ++k, ++r) {
      if (r >= W.cap_rows) return;
      double *w = rows + 7 * r;
      const double fx = -pr[5 * k], fy = -pr[5 * k + 1], px = pr[5 * k + 2], py = pr[5 * k + 3];
      const double tq = (px - x) * fy - (py - y) * fx;
      w[0] = (double)(src + 1); w[1] = fx; w[2] = fy;
      w[3] = px; w[4] = py; w[5] = tq; w[6] = pr[5 * k + 4];
      sfx += fx; sfy += fy; stq += tq;
      ov = ov + pr[5 * k + 4];                                // :199
    }
  }
  W.ov_ext[i] = ov;
  W.cf_ext[2 * i] = sfx; W.cf_ext[2 * i + 1] = sfy; W.ct_ext[i] = stq;
  centre_check(i, x, y, F, P, W);
}

// ---------------------------------------------------------------------------------------------
// sharded narrow-phase: the pairs [p_lo, p_hi) a rank computed leave as (nrows, flag) per pair plus their
// rows packed back to back; every rank imports every segment, after which all ranks hold identical results
// ---------------------------------------------------------------------------------------------
// One exported segment: [header 32 B: pairs, contact rows, status, rows_max as int64][rows-per-pair int32 x seg_max]
// [flags int32 x seg_max][rows-before-pair int32 x seg_max][pad to 8][contact rows: 5 doubles each, rows_max of them]
enum { SEG_HDR = 32 };
SZ_HD size_t seg_rows_at(i64 seg_max) { return SEG_HDR + (size_t)((12 * seg_max + 7) / 8) * 8; }
SZ_HD size_t seg_bytes_for(i64 seg_max, i64 rows_max) { return seg_rows_at(seg_max) + 40 * (size_t)rows_max; }
struct Xfer {
  int *nrows, *flag, *before; // per pair of the segment
  double *rows;               // 5 doubles per row, packed
  i64 rows_max;               // rows the buffer holds
};
SZ_HD Xfer seg_view(char *base, i64 seg_max, i64 rows_max) {
  Xfer X;
  X.nrows = (int *)(base + SEG_HDR); X.flag = X.nrows + seg_max; X.before = X.flag + seg_max;
  X.rows = (double *)(base + seg_rows_at(seg_max));
  X.rows_max = rows_max;
  return X;
}
SZ_HD void seg_header(char *base, i64 ns, i64 rows, i64 status, i64 rows_max) {
  i64 *h = (i64 *)base;
  h[0] = ns; h[1] = rows; h[2] = status; h[3] = rows_max;
}
// off = exclusive prefix sum of pair_nrows over my segment (rows before pair k)
SZ_HD void export_pair(i64 k, i64 p_lo, const i64 *off, const Floes &F, const Work &W, const Xfer &X) {
  const i64 p = p_lo + k;
  const int nr = W.pair_nrows[p];
  X.nrows[k] = nr; X.flag[k] = W.pair_flag[p]; X.before[k] = (int)off[k];
  if (nr > 0 && off[k] + nr <= X.rows_max) {       // (a segment that does not fit is exchanged again, larger)
    const double *src = pair_rows(p, F, W);
    double *dst = X.rows + 5 * off[k];
    for (int q = 0; q < 5 * nr; ++q) dst[q] = src[q];
  }
}
// segment of the global pair list computed by rank r of w (all ranks derive the same split)
SZ_HD void shard_range(i64 np, int r, int w, i64 &lo, i64 &hi) { lo = np * r / w; hi = np * (r + 1) / w; }
SZ_HD i64 shard_max(i64 np, int w) { return (np + w - 1) / w; }
// pair p of the global list out of the gathered buffer (`world` segments of seg_bytes each, rank order)
SZ_HD void import_pair(i64 p, i64 np, int world, i64 seg_max, i64 rows_max, char *recv, Work &W) {
  int r = (int)(p * world / np);
  i64 lo, hi;
  shard_range(np, r, world, lo, hi);
  while (p < lo) { --r; shard_range(np, r, world, lo, hi); }
  while (p >= hi) { ++r; shard_range(np, r, world, lo, hi); }
  char *base = recv + (size_t)r * seg_bytes_for(seg_max, rows_max);
  const Xfer X = seg_view(base, seg_max, rows_max);
  const i64 k = p - lo;
  W.pair_nrows[p] = X.nrows[k]; W.pair_flag[p] = X.flag[k];
  W.pair_rowoff[p] = (i64)((X.rows - (const double *)recv) + 5 * (i64)X.before[k]);  // in doubles from the start of recv
}

// transfer(kill(i)) = i for the largest such i (:175-179): first collect, then apply
SZ_HD void transfer_collect(int i, const Floes &F, Work &W) { // i < N
  int k = F.kill[i];
  if (iabs(k - (i + 1)) > 0 && k > 0) atomic_max(&W.tmax[k - 1], i + 1);
}
SZ_HD void transfer_apply(int i, const Floes &F, Work &W) {
  if (W.tmax[i] > 0) F.transfer[i] = W.tmax[i];
}

// ---------------------------------------------------------------------------------------------
// calc_trajectory.m for real floe i (plus :242-245 ghost sums, :262-277 own sums and periodic wrap)
// ---------------------------------------------------------------------------------------------
// interp2(Xo, Yo, V, xq, yq) (linear) in two parts, so that several fields sampled at one point share the cell search
struct Interp2 { int ix, iy; double tx, ty; bool inside; };
SZ_HD Interp2 interp2_locate(const double *Xo, int nx, const double *Yo, int ny, double xq, double yq) {
  Interp2 w; w.ix = 0; w.iy = 0; w.tx = 0; w.ty = 0;
  w.inside = (xq >= Xo[0] && xq <= Xo[nx - 1] && yq >= Yo[0] && yq <= Yo[ny - 1]);
  if (!w.inside) return w;
  int ix = (int)::floor((xq - Xo[0]) / (Xo[1] - Xo[0]));
  int iy = (int)::floor((yq - Yo[0]) / (Yo[1] - Yo[0]));
  if (ix > nx - 2) ix = nx - 2;
  if (iy > ny - 2) iy = ny - 2;
  if (ix < 0) ix = 0;
  if (iy < 0) iy = 0;
  while (ix > 0 && xq < Xo[ix]) --ix;
  while (ix < nx - 2 && xq > Xo[ix + 1]) ++ix;
  while (iy > 0 && yq < Yo[iy]) --iy;
  while (iy < ny - 2 && yq > Yo[iy + 1]) ++iy;
  w.ix = ix; w.iy = iy;
  w.tx = (xq - Xo[ix]) / (Xo[ix + 1] - Xo[ix]);
  w.ty = (yq - Yo[iy]) / (Yo[iy + 1] - Yo[iy]);
  return w;
}
SZ_HD double interp2_at(const Interp2 &w, int ny, const double *V) {
  if (!w.inside) return __builtin_nan("");
  const int ix = w.ix, iy = w.iy;
  double v00 = V[iy + (size_t)ny * ix], v01 = V[iy + (size_t)ny * (ix + 1)];
  double v10 = V[iy + 1 + (size_t)ny * ix], v11 = V[iy + 1 + (size_t)ny * (ix + 1)];
  double a = v00 * (1 - w.tx) + v01 * w.tx;
  double b = v10 * (1 - w.tx) + v11 * w.tx;
  return a * (1 - w.ty) + b * w.ty;
}
SZ_HD double interp2_linear(const double *Xo, int nx, const double *Yo, int ny, const double *V, double xq, double yq) {
  return interp2_at(interp2_locate(Xo, nx, Yo, ny, xq, yq), ny, V);
}

// calc_trajectory is split in three kernels so that the Monte-Carlo forcing (1000 points per floe, every
// 10th step) can run with one wave per floe:  integrate_pre  ->  forcing  ->  integrate_post.
// Values handed from one to the next live in Work::st_* (per real floe).
enum { ST_GO = 1, ST_FORCE = 2 };

// max(Xo), min(Xo) of a grid axis, the same for every floe: evaluated once by the host wrapper
SZ_HD void axis_extent(const double *a, int n, double &mx, double &mn) {
  mx = a[0]; mn = a[0];
  for (int k = 1; k < n; ++k) { mx = szc::maxd(mx, a[k]); mn = szc::mind(mn, a[k]); }
}
// mean(StressH,3) (calc_trajectory.m:20) without reading the 32 KB history of a floe every step: the sum of the
// slots is carried as an unevaluated pair hi + lo (|lo| <= ulp(hi)/2) and every replacement of a slot is applied
// with error-free transformations (Knuth's TwoSum), so hi + lo stays equal to the exact sum of the current slots to
// ~1e-32 relative for any number of steps -- no drift, and nothing left behind when a large entry rotates out.
// hi is then that exact sum correctly rounded: closer to the true mean than the left-to-right sum of 1000 terms
// the reference evaluates (whose own rounding is what separates the two, <= 1.2e-16 * sum|slots|).
// (needs IEEE arithmetic as written: the library is built with -ffp-contract=off and without fast-math)
SZ_HD void two_sum(double a, double b, double &s, double &e) {
  s = a + b;
  const double bb = s - a;
  e = (a - (s - bb)) + (b - bb);
}
SZ_HD void stress_sum_add(double &hi, double &lo, double v) {
  double s, e;
  two_sum(hi, v, s, e);
  lo += e;
  two_sum(s, lo, hi, lo);
}
SZ_HD void stress_sum_replace(double &hi, double &lo, double old, double now) {
  double a, e1, b, e2;
  two_sum(hi, -old, a, e1);
  two_sum(a, now, b, e2);
  lo += e1 + e2;
  two_sum(b, lo, hi, lo);
}
// the pair for the history as uploaded: slots added in order (stress_sum_add keeps the sum exact)
SZ_HD void stress_sum_init(int i, const Floes &F) {
  for (int q = 0; q < 4; ++q) {
    double hi = 0, lo = 0;
    for (int z = 0; z < F.n_stress; ++z) stress_sum_add(hi, lo, F.StressH[((size_t)i * F.n_stress + z) * 4 + q]);
    F.StressSum[8 * i + q] = hi; F.StressSum[8 * i + 4 + q] = lo;
  }
}

SZ_HD void integrate_pre(int i, const Floes &F, const Params &P, Work &W) { // nb <= i < n
  W.st_flag[i] = 0;
  if (!is_owned(i, W)) return;   // (sharded step: another rank integrates this floe)
  // ghost sums first (:242-245): the X ghost, then the Y ghost, of this floe; then the floe's own (:262-263)
  double cfx = 0, cfy = 0, ctq = 0;
  if (P.PERIODIC) {
    int g = W.xg_of[i];
    if (g >= 0) { cfx = cfx + W.cf_ext[2 * g]; cfy = cfy + W.cf_ext[2 * g + 1]; ctq = ctq + W.ct_ext[g]; }
    g = W.yg_of[i];
    if (g >= 0) { cfx = cfx + W.cf_ext[2 * g]; cfy = cfy + W.cf_ext[2 * g + 1]; ctq = ctq + W.ct_ext[g]; }
  }
  const i64 r0 = W.row_off[i], r1 = W.row_off[i + 1];
  const int nrows = (int)(r1 - r0);
  if (nrows > 0) { cfx = W.cf_ext[2 * i] + cfx; cfy = W.cf_ext[2 * i + 1] + cfy; ctq = W.ct_ext[i] + ctq; }
  F.collision_force[2 * i] = cfx; F.collision_force[2 * i + 1] = cfy;
  F.collision_torque[i] = ctq;
  F.OverlapArea[i] = W.ov_ext[i];
  const double x_before = W.ext_x[i];
  if (P.PERIODIC) {                                              // :267-277
    if (absd(F.Xi[i]) > P.Lx) F.Xi[i] = F.Xi[i] - 2 * P.Lx * sgn(F.Xi[i]);
    if (absd(F.Yi[i]) > P.Ly) F.Yi[i] = F.Yi[i] - 2 * P.Ly * sgn(F.Yi[i]);
  }
  if (!F.alive[i]) return;                                       // :280

  // ---- calc_trajectory.m ----
  const double dt = P.dt;
  double ext_fx = cfx, ext_fy = cfy, ext_torque = ctq;
  const int Nz = F.n_stress;
  double *SH = F.StressH + (size_t)i * Nz * 4;
  double S[4] = {0, 0, 0, 0};
  if (nrows > 0) {                                               // :9-13
    double rx = F.Xi[i], ry = F.Yi[i];
    double s11 = 0, s12 = 0, s21 = 0, s22 = 0, t11 = 0, t12 = 0, t21 = 0, t22 = 0;
    for (i64 q = r0; q < r1; ++q) {
      const double *a = W.rows + 7 * q;
      s11 += (a[3] - rx) * a[1]; s12 += (a[4] - ry) * a[1];
      s21 += (a[3] - rx) * a[2]; s22 += (a[4] - ry) * a[2];
      t11 += a[1] * (a[3] - rx); t12 += a[2] * (a[3] - rx);
      t21 += a[1] * (a[4] - ry); t22 += a[2] * (a[4] - ry);
    }
    double c = 1 / (2 * F.area[i] * F.h[i]);
    S[0] = c * (s11 + t11); S[1] = c * (s21 + t21); S[2] = c * (s12 + t12); S[3] = c * (s22 + t22);
  }
  // everything calc_trajectory changes before it may return [] is staged here, so that a removed floe
  // keeps its pre-call fields (floe_interactions_all.m:282)
  int scount = F.StressCount[i];
  if (scount > Nz) scount = 1;                                   // :15-17
  double hh = F.h[i], mass = F.mass[i], inertia = F.inertia_moment[i];
  int alive = F.alive[i];
  if (hh > 10) hh = 10;                                          // :36-41
  else if (mass < 100) { mass = 1e3; alive = 0; }
  while (szc::maxd(absd(ext_fx), absd(ext_fy)) > mass / (5 * dt)) { // :42-46
    ext_fx = ext_fx / 10; ext_fy = ext_fy / 10; ext_torque = ext_torque / 10;
  }
  double dh = P.HFo * dt / hh;                                   // :76-80
  double floe_mass = (hh - dh) / hh * mass;
  double floe_inertia = (hh - dh) / hh * inertia;
  double h_new = hh - dh;
  bool removed = false;
  if (F.Xi[i] != F.Xi[i]) removed = true;                        // :89
  const int v0 = F.vert_off[i], nvr = F.vert_off[i + 1] - v0;
  const bool doforce = P.doInt || h_new < 0.1;                   // :94,:121
  if (!removed) {                                                // :116
    double mxx = F.cax[v0], mnx = F.cax[v0], mxy = F.cay[v0], mny = F.cay[v0];
    for (int k = 1; k < nvr; ++k) {
      mxx = szc::maxd(mxx, F.cax[v0 + k]); mnx = szc::mind(mnx, F.cax[v0 + k]);
      mxy = szc::maxd(mxy, F.cay[v0 + k]); mny = szc::mind(mny, F.cay[v0 + k]);
    }
    const double Xomax = P.Xo_max, Xomin = P.Xo_min, Yomax = P.Yo_max, Yomin = P.Yo_min;
    if (mxx + F.Xi[i] > Xomax || mnx + F.Xi[i] < Xomin || mxy + F.Yi[i] > Yomax || mny + F.Yi[i] < Yomin)
      removed = true;
  }
  if (removed || x_before != x_before) { F.kill[i] = i + 1; }   // :282
  if (removed) return;
  // commit the staged fields
  for (int q = 0; q < 4; ++q) {
    double old = SH[(size_t)(scount - 1) * 4 + q];
    SH[(size_t)(scount - 1) * 4 + q] = S[q];                     // :18 / :26
    double hi = F.StressSum[8 * i + q], lo = F.StressSum[8 * i + 4 + q];
    stress_sum_replace(hi, lo, old, S[q]);
    F.StressSum[8 * i + q] = hi; F.StressSum[8 * i + 4 + q] = lo;
    F.Stress[4 * i + q] = hi / Nz;                               // :20 mean(StressH,3)
  }
  F.StressCount[i] = scount + 1;
  F.h[i] = h_new; F.mass[i] = floe_mass; F.inertia_moment[i] = floe_inertia; F.alive[i] = alive;
  if (alive != 1) return;                                        // :118
  W.st_ext[3 * i] = ext_fx; W.st_ext[3 * i + 1] = ext_fy; W.st_ext[3 * i + 2] = ext_torque;
  W.st_flag[i] = ST_GO | (doforce ? ST_FORCE : 0);
}

// ---- forcing (:95-98, :121-166): per-point pieces, summed left to right over the points inside the floe ----
struct ForceCtx {
  double sa, ca, sta, cta, Xi, Yi, Ui, Vi, ksi, mass_per_area, fc, Fx_atm, Fy_atm;
};
SZ_HD void force_ctx(int i, const Floes &F, const Params &P, ForceCtx &c) {
  szm::sincos(F.alpha_i[i], c.sa, c.ca);
  szm::sincos(P.turn_angle, c.sta, c.cta);
  c.Xi = F.Xi[i]; c.Yi = F.Yi[i]; c.Ui = F.Ui[i]; c.Vi = F.Vi[i]; c.ksi = F.ksi_ice[i];
  c.mass_per_area = F.mass[i] / F.area[i]; c.fc = P.fCoriolis;
  c.Fx_atm = 0; c.Fy_atm = 0;
}
// Everything of one Monte-Carlo point that depends only on its position (:95-98, :122-137): winds for the first
// mean, ocean velocities and the polar angle for the second.  McPoint keeps the latter between the two passes.
struct McPoint { double Uo, Vo, sth, cth, rho; };
SZ_HD void point_sample(const ForceCtx &c, const Params &P, double mx, double my, bool want_wind, bool want_ocean,
                        double &uw, double &vw, McPoint &m) {
  double xr1 = c.ca * mx + (-c.sa) * my, xr2 = c.sa * mx + c.ca * my;
  double Xg = xr1 + c.Xi, Yg = xr2 + c.Yi;
  const Interp2 w = interp2_locate(P.Xo, P.nxo, P.Yo, P.nyo, Xg, Yg);
  if (want_wind) { uw = interp2_at(w, P.nyo, P.Uwind); vw = interp2_at(w, P.nyo, P.Vwind); }
  if (want_ocean) {
    double th = szm::atan2(xr2, xr1);
    m.rho = szc::hyp(xr1, xr2);
    m.Uo = interp2_at(w, P.nyo, P.Uocn);
    m.Vo = interp2_at(w, P.nyo, P.Vocn);
    szm::sincos(th, m.sth, m.cth);
  }
}
// stresses and torque of one point from its sample (:127-160); needs the wind stress of the floe (wind_stress)
SZ_HD void point_force(const ForceCtx &c, const McPoint &m, double &Fx, double &Fy, double &tq) {
  const double rho0 = 1027, Cd = 3e-3;
  double Uice = c.Ui - m.rho * c.ksi * m.sth;
  double Vice = c.Vi + m.rho * c.ksi * m.cth;
  double Fx_p = -(c.mass_per_area) * c.fc * m.Vo;
  double Fy_p = +(c.mass_per_area) * c.fc * m.Uo;
  double du = m.Uo - Uice, dv = m.Vo - Vice;
  double sp = ::sqrt(du * du + dv * dv);
  double tauX = rho0 * Cd * sp * (c.cta * du - c.sta * dv);
  double tauY = rho0 * Cd * sp * (c.sta * du + c.cta * dv);
  Fx = tauX + c.Fx_atm + Fx_p;
  Fy = tauY + c.Fy_atm + Fy_p;
  tq = (-Fx * m.sth + Fy * m.cth) * m.rho;
  Fx = Fx + (c.mass_per_area) * c.fc * c.Vi;
  Fy = Fy - (c.mass_per_area) * c.fc * c.Ui;
}
SZ_HD void wind_point(const ForceCtx &c, const Params &P, double mx, double my, double &uw, double &vw) {
  McPoint unused;
  point_sample(c, P, mx, my, true, false, uw, vw, unused);
}
SZ_HD void wind_stress(ForceCtx &c, double su, double sv, int cnt) {            // :139-141
  const double rho_air = 1.2, Cd_atm = 1e-3;
  double U10 = su / cnt, V10 = sv / cnt;
  double w10 = ::sqrt(U10 * U10 + V10 * V10);
  c.Fx_atm = rho_air * Cd_atm * w10 * U10;
  c.Fy_atm = rho_air * Cd_atm * w10 * V10;
}
SZ_HD void force_point(const ForceCtx &c, const Params &P, double mx, double my, double &Fx, double &Fy, double &tq) {
  McPoint m;
  double uw, vw;
  point_sample(c, P, mx, my, false, true, uw, vw, m);
  point_force(c, m, Fx, Fy, tq);
}
// The means over the Monte-Carlo points (calc_trajectory.m:139,163-165) are summed in a fixed order that 256
// threads can follow: point q goes into partial q mod 256 (increasing q), then the partials are folded pairwise,
// p[t] += p[t + s] for s = 128, 64, ..., 1.  MC_PARTS is part of the numerical definition (the oracle has the same).
enum { MC_PARTS = 256 };
SZ_HD double mc_fold(double *p) {
  for (int s = MC_PARTS / 2; s >= 1; s >>= 1)
    for (int t = 0; t < s; ++t) p[t] += p[t + s];
  return p[0];
}
// whole forcing of one floe by one thread (host emulation of k_forcing in step.hip: the same per-point functions,
// the same summation order)
SZ_HD void forcing_serial(int i, const Floes &F, const Params &P, Work &W) {
  if ((W.st_flag[i] & (ST_GO | ST_FORCE)) != (ST_GO | ST_FORCE)) return;
  const int nmc = F.n_mc;
  const double *mx = F.mcX + i * F.mc_si, *my = F.mcY + i * F.mc_si;
  const unsigned char *A = F.mcA + i * F.mc_si;
  const long long sq = F.mc_sq;
  int cnt = 0;
  for (int q = 0; q < nmc; ++q) cnt += A[q * sq] ? 1 : 0;
  if (cnt == 0) { atomic_add(&W.counters[C_ERR_MC], 1ULL); W.st_flag[i] = 0; return; } // :100-110 needs rand()
  ForceCtx c;
  force_ctx(i, F, P, c);
  double p0[MC_PARTS], p1[MC_PARTS], p2[MC_PARTS];
  for (int t = 0; t < MC_PARTS; ++t) { p0[t] = 0.0; p1[t] = 0.0; }
  for (int q = 0; q < nmc; ++q) {
    if (!A[q * sq]) continue;
    double uw, vw;
    wind_point(c, P, mx[q * sq], my[q * sq], uw, vw);
    p0[q % MC_PARTS] += uw; p1[q % MC_PARTS] += vw;
  }
  const double su = mc_fold(p0), sv = mc_fold(p1);
  wind_stress(c, su, sv, cnt);
  for (int t = 0; t < MC_PARTS; ++t) { p0[t] = 0.0; p1[t] = 0.0; p2[t] = 0.0; }
  for (int q = 0; q < nmc; ++q) {
    if (!A[q * sq]) continue; // only the points inside the floe enter the means (:163-165)
    double Fx, Fy, tq;
    force_point(c, P, mx[q * sq], my[q * sq], Fx, Fy, tq);
    p0[q % MC_PARTS] += Fx; p1[q % MC_PARTS] += Fy; p2[q % MC_PARTS] += tq;
  }
  const double sFx = mc_fold(p0), sFy = mc_fold(p1), sT = mc_fold(p2);
  F.FxOA[i] = sFx / cnt; F.FyOA[i] = sFy / cnt; F.torqueOA[i] = sT / cnt;
}

SZ_HD void integrate_post(int i, const Floes &F, const Params &P, Work &W) { // nb <= i < n
  if (!(W.st_flag[i] & ST_GO)) return;
  const double dt = P.dt;
  const double ext_fx = W.st_ext[3 * i], ext_fy = W.st_ext[3 * i + 1], ext_torque = W.st_ext[3 * i + 2];
  const double floe_area = F.area[i], floe_mass = F.mass[i], floe_inertia = F.inertia_moment[i], h_new = F.h[i];
  const int v0 = F.vert_off[i], nvr = F.vert_off[i + 1] - v0;
  // :174-177
  double dx = 1.5 * dt * F.Ui[i] - 0.5 * dt * F.dXi_p[i];
  double dy = 1.5 * dt * F.Vi[i] - 0.5 * dt * F.dYi_p[i];
  F.Xi[i] = F.Xi[i] + dx; F.dXi_p[i] = F.Ui[i];
  F.Yi[i] = F.Yi[i] + dy; F.dYi_p[i] = F.Vi[i];
  F.alpha_i[i] = F.alpha_i[i] + 1.5 * dt * F.ksi_ice[i] - 0.5 * dt * F.dalpha_i_p[i];
  F.dalpha_i_p[i] = F.ksi_ice[i];
  // :181-208
  double numU = F.FxOA[i] * floe_area + ext_fx, numV = F.FyOA[i] * floe_area + ext_fy;
  double dUi_dt = numU / floe_mass, dVi_dt = numV / floe_mass;
  bool have_frac = false;
  double frac = 0;
  bool bigU = absd(dt * dUi_dt) > 0.5 * h_new, bigV = absd(dt * dVi_dt) > 0.5 * h_new;
  bool smallU = absd(dt * dUi_dt) < 0.5 * h_new, smallV = absd(dt * dVi_dt) < 0.5 * h_new;
  if (bigU && bigV) {
    double a = sgn(dUi_dt) * 0.5 * h_new / dt, b = sgn(dVi_dt) * 0.5 * h_new / dt;
    double frac1 = a / numU * floe_mass, frac2 = b / numV * floe_mass;
    frac = szc::mind(frac1, frac2); have_frac = true;
  } else if (bigU && smallV) {
    double a = sgn(dUi_dt) * 0.5 * h_new / dt;
    frac = a / numU * floe_mass; have_frac = true;
  } else if (smallU && bigV) {
    double b = sgn(dVi_dt) * 0.5 * h_new / dt;
    frac = b / numV * floe_mass; have_frac = true;
  }
  if (have_frac) { dUi_dt = frac * (numU / floe_mass); dVi_dt = frac * (numV / floe_mass); }
  F.Ui[i] = F.Ui[i] + 1.5 * dt * dUi_dt - 0.5 * dt * F.dUi_p[i];
  F.Vi[i] = F.Vi[i] + 1.5 * dt * dVi_dt - 0.5 * dt * F.dVi_p[i];
  F.dUi_p[i] = dUi_dt; F.dVi_p[i] = dVi_dt;
  // :210-219
  double dksi = (F.torqueOA[i] * floe_area + ext_torque) / floe_inertia;
  if (have_frac) dksi = frac * dksi;
  double ksi = F.ksi_ice[i] + 1.5 * dt * dksi - 0.5 * dt * F.dksi_ice_p[i];
  if (absd(ksi) > 1e-5) ksi = sgn(ksi) * 1e-5;
  F.ksi_ice[i] = ksi;
  F.dksi_ice_p[i] = dksi;
  // :221-222
  double sa, ca;
  szm::sincos(F.alpha_i[i], sa, ca);
  for (int k = 0; k < nvr; ++k) {
    F.cax[v0 + k] = ca * F.c0x[v0 + k] + (-sa) * F.c0y[v0 + k];
    F.cay[v0 + k] = sa * F.c0x[v0 + k] + ca * F.c0y[v0 + k];
  }
  if (P.doInt) {                                                 // :224-234
    double sudy = 0, sudx = 0, svdy = 0, svdx = 0;
    double U0 = 0, V0 = 0, Uprev = 0, Vprev = 0;
    for (int k = 0; k <= nvr; ++k) {
      int kk = (k == nvr) ? 0 : k;
      double cxk = F.cax[v0 + kk], cyk = F.cay[v0 + kk];
      double Uk, Vk;
      if (k == nvr) { Uk = U0; Vk = V0; }
      else {
        double th = szm::atan2(cyk, cxk), rh = szc::hyp(cxk, cyk);
        double sth, cth;
        szm::sincos(th, sth, cth);
        Uk = F.Ui[i] - rh * F.ksi_ice[i] * sth;
        Vk = F.Vi[i] + rh * F.ksi_ice[i] * cth;
        if (k == 0) { U0 = Uk; V0 = Vk; }
      }
      if (k > 0) {
        double dU = Uk - Uprev, dV = Vk - Vprev;
        double dcy = cyk - F.cay[v0 + k - 1], dcx = cxk - F.cax[v0 + k - 1];
        sudy += dU * dcy; sudx += dU * dcx; svdy += dV * dcy; svdx += dV * dcx;
      }
      Uprev = Uk; Vprev = Vk;
    }
    double du_dx = 0.5 * sudy / floe_area, du_dy = 0.5 * sudx / floe_area;
    double dv_dx = 0.5 * svdy / floe_area, dv_dy = 0.5 * svdx / floe_area;
    F.strain[4 * i + 0] = 0.5 * (du_dx + du_dx);
    F.strain[4 * i + 1] = 0.5 * (dv_dx + du_dy);
    F.strain[4 * i + 2] = 0.5 * (du_dy + dv_dx);
    F.strain[4 * i + 3] = 0.5 * (dv_dy + dv_dy);
  }
}

} // namespace szs
