// TEST-ONLY: runs the kernel bodies of subzero_amd/csrc/step_core.hpp in plain loops on the CPU -- the
// same source the GPU kernels execute -- so the CPU test-suite can compare the device pipeline with the
// oracle without a GPU, and exercise the sharded (multi-GPU) form of the step under a real process group
// (gloo).  Not part of the product; nothing in subzero_amd/ loads this.
static long long g_fast_hits = 0, g_fast_bails = 0, g_bail_sites[64];
#define SZX_BAIL_PROBE(site) (++g_bail_sites[(site) & 63])
#include <stdio.h>
static int g_dump_third = 0;
static void dump_third(const long long *tx, const long long *ty, int nt, const long long *rx, const long long *ry, int nr) {
  if (!g_dump_third) return;
  printf("THIRD T");
  for (int i = 0; i < nt; ++i) printf(" %.9f,%.9f", (double)tx[i] / 4294967296.0, (double)ty[i] / 4294967296.0);
  printf(" R");
  for (int i = 0; i < nr; ++i) printf(" %.9f,%.9f", (double)rx[i] / 4294967296.0, (double)ry[i] / 4294967296.0);
  printf("\n"); fflush(stdout);
}
#define SZC_THIRD_PROBE(tx, ty, nt, rx, ry, nr) dump_third((const long long *)(tx), (const long long *)(ty), nt, (const long long *)(rx), (const long long *)(ry), nr)
#define SZC_FAST_PROBE(st) ((st) == 100 ? ++g_fast_bails : ++g_fast_hits)
#include <x86intrin.h>
static unsigned long long g_phase_t[16], g_phase_last = 0; static int g_phase_cur = 0;
static long long g_trim_before = 0, g_trim_after = 0;
static int g_last_trim[4], g_last_k = 0, g_worst_trim[4];
#define SZC_TRIM_PROBE(before, after) (g_trim_before += (before), g_trim_after += (after), g_last_trim[g_last_k & 3] = (before), g_last_trim[(g_last_k + 1) & 3] = (after), g_last_k += 2)
static unsigned long long g_pair_t[16], g_worst_t[16], g_worst_total = 0;
static inline void phase_mark(int k) {
  unsigned long long now = __rdtsc();
  g_phase_t[g_phase_cur & 15] += now - g_phase_last; g_pair_t[g_phase_cur & 15] += now - g_phase_last;
  g_phase_last = now; g_phase_cur = k;
  if (k == 8) for (int i = 0; i < 16; ++i) g_pair_t[i] = 0;            // a pair starts
  if (k == 0) {                                                          // a pair ends
    unsigned long long tot = 0; for (int i = 1; i < 16; ++i) tot += g_pair_t[i];
    if (tot > g_worst_total) { g_worst_total = tot; for (int i = 0; i < 16; ++i) g_worst_t[i] = g_pair_t[i]; for (int i = 0; i < 4; ++i) g_worst_trim[i] = g_last_k > i ? g_last_trim[i] : -1; }
    g_last_k = 0;
  }
}
#define SZC_PHASE(k) phase_mark(k)
static bool g_trim_off = false, g_prune_off = false;
#define SZC_PRUNE_OFF g_prune_off
static long long g_prune_runs[2];
#define SZC_PRUNE_PROBE(skipped) (++g_prune_runs[(skipped)])
#define SZC_TRIM_OFF g_trim_off
static long long g_team_bails[128];
#define SZTM_BAIL_PROBE(site) (++g_team_bails[(site) & 127])
#include "team_host.hpp"
#include "../../subzero_amd/csrc/step_core.hpp"
#include "../../subzero_amd/csrc/spatial_core.hpp"
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

using namespace szs;

// spatially sharded step (spatial_core.hpp): the masks and the strips live across the calls of one rank's process
static bool g_sp_active = false;
static std::vector<unsigned char> g_sp_masks;     // present[n] then owned[n]
static std::vector<double> g_sp_edges;
static int g_sp_rank = 0, g_sp_world = 1, g_sp_axis = 0;
static double g_sp_halo = 0, g_sp_rmax_max = 0;
static int g_narrow_mode = 3;   // (the GPU default: team pass first)
static long long g_big_pairs = 0;
extern "C" long long host_big_pairs(void) { return g_big_pairs; }
static int g_keep_sums = 0;
static std::vector<double> g_sums;
static const void *g_sums_for = nullptr;
extern "C" void host_keep_stress_sums(int on) { g_keep_sums = on; if (!on) g_sums_for = nullptr; }
extern "C" void host_set_narrow_mode(int m) { g_narrow_mode = m; }
static long long g_team_pairs = 0, g_team_deferred = 0, g_team_wide = 0, g_cold_toggle = 0;
extern "C" void host_team_counts(long long *pairs, long long *deferred, long long *sites) { *pairs = g_team_pairs; *deferred = g_team_deferred; for (int i = 0; i < 128; ++i) sites[i] = g_team_bails[i]; }
extern "C" void host_set_dump_third(int on) { g_dump_third = on; }
extern "C" void host_set_trim(int on) { g_trim_off = !on; }
extern "C" void host_set_prune(int on) { g_prune_off = !on; }
extern "C" void host_prune_counts(long long *kept, long long *skipped) { *kept = g_prune_runs[0]; *skipped = g_prune_runs[1]; }
// floe_interactions.m:127's test of p_poly_dist against 1e-8 on a closed ring: the full distance / the near-elements form
extern "C" int host_closer(double xp, double yp, const double *x, const double *y, int n, int pruned) {
  szc::RingD c = {x, y, 0.0, 0.0, n, 0};
  return pruned ? (int)szc::closer_than_1e8(xp, yp, c) : (int)(szc::dist_to_ring(xp, yp, c) < 1e-8);
}
// InterX on two closed rings in metres: the plain dense form (pruned = 0) or the one for long rings
extern "C" int host_crossings(const double *x1, const double *y1, int n1, const double *x2, const double *y2, int n2,
                              int pruned, double *px, double *py, int cap) {
  szc::RingD c1 = {x1, y1, 0.0, 0.0, n1, 0}, c2 = {x2, y2, 0.0, 0.0, n2, 0};
  if (!pruned) return szc::crossings(c1, c2, px, py, cap);
  std::vector<double> sx((size_t)n2 + 4 * ((size_t)n2 / szc::CROSS_CHUNK + 1) + 8), sy((size_t)n2 + 8);
  return szc::crossings_long(c1, c2, px, py, cap, sx.data(), sy.data());
}
extern "C" void host_worst_trim(int *out) { for (int i = 0; i < 4; ++i) out[i] = g_worst_trim[i]; }
extern "C" void host_worst_pair(unsigned long long *out) { for (int i = 0; i < 16; ++i) { out[i] = g_worst_t[i]; g_worst_t[i] = 0; } g_worst_total = 0; }
extern "C" void host_phase_times(unsigned long long *out) { phase_mark(0); for (int i = 0; i < 16; ++i) { out[i] = g_phase_t[i]; g_phase_t[i] = 0; } }
extern "C" void host_trim_counts(long long *before, long long *after) { *before = g_trim_before; *after = g_trim_after; }
extern "C" void host_bail_sites(long long *out) { for (int i = 0; i < 64; ++i) out[i] = g_bail_sites[i]; }
extern "C" void host_fast_counts(long long *hits, long long *bails) { *hits = g_fast_hits; *bails = g_fast_bails; }

// the stress-sum pair (step_core.hpp) on a stream of values: hi after the init and after every replacement
extern "C" void host_stress_sum_trace(const double *vals, int nslots, int nsteps, double *hi_out) {
  std::vector<double> slot(vals, vals + nslots);
  double hi = 0, lo = 0;
  for (int z = 0; z < nslots; ++z) stress_sum_add(hi, lo, slot[z]);
  hi_out[0] = hi;
  for (int k = 0; k < nsteps; ++k) {
    const int z = k % nslots;
    stress_sum_replace(hi, lo, slot[z], vals[nslots + k]);
    slot[z] = vals[nslots + k];
    hi_out[k + 1] = hi;
  }
}

namespace {

template <class T> T *alloc(std::vector<std::vector<char> > &pool, size_t n) {
  pool.emplace_back(n * sizeof(T) + 64);
  memset(pool.back().data(), 0, pool.back().size());
  return (T *)(((uintptr_t)pool.back().data() + 15) & ~(uintptr_t)15);
}

// same field order as OrFloes / szFloeState
struct HFloes {
  int32_t n, nb, n_stress, n_mc;
  double *Xi, *Yi, *alpha_i, *Ui, *Vi, *ksi_ice;
  double *h, *area, *mass, *inertia_moment, *rmax;
  double *dXi_p, *dYi_p, *dUi_p, *dVi_p, *dalpha_i_p, *dksi_ice_p;
  double *FxOA, *FyOA, *torqueOA;
  int32_t *alive;
  int32_t *vert_off;
  double *c0x, *c0y, *cax, *cay;
  double *StressH;
  int32_t *StressCount;
  double *Stress, *strain;
  double *mcX, *mcY;
  uint8_t *mcA;
  double *collision_force, *collision_torque, *OverlapArea;
  int32_t *kill, *transfer;
};
struct HParams {
  double dt, HFo, Modulus, mu, nu;
  int32_t doInt, COLLISION, PERIODIC, reserved;
  double c2_boundary[10];
  double fb_h, fb_area, fb_Xi, fb_Yi, fb_Ui, fb_Vi, fb_ksi_ice;
  int32_t nxo, nyo;
  const double *Xo, *Yo, *Uocn, *Vocn, *Uwind, *Vwind;
  double fCoriolis, turn_angle;
};

// one step in flight (front -> narrow -> back), as step.hip keeps it in StepState
struct Sim {
  std::vector<std::vector<char> > pool;
  Floes F;
  Params P;
  Work W;
  int n = 0, N = 0, cap = 0;
  i64 np = 0, nvt = 0, retried = 0;
  int rank = 0, world = 1;
  i64 seg_lo = 0, seg_hi = 0, seg_rows = 0;
  std::vector<i64> x_off;

  void bind(HFloes *hf, const HParams *hp, int pot_cap) {
    pool.clear();
    bind_floes(hf);
    bind_rest(hp, pot_cap);
  }
  void bind_floes(HFloes *hf) {
    F.n = hf->n; F.nb = hf->nb; F.n_stress = hf->n_stress; F.n_mc = hf->n_mc;
    F.Xi = hf->Xi; F.Yi = hf->Yi; F.alpha_i = hf->alpha_i; F.Ui = hf->Ui; F.Vi = hf->Vi; F.ksi_ice = hf->ksi_ice;
    F.h = hf->h; F.area = hf->area; F.mass = hf->mass; F.inertia_moment = hf->inertia_moment; F.rmax = hf->rmax;
    F.dXi_p = hf->dXi_p; F.dYi_p = hf->dYi_p; F.dUi_p = hf->dUi_p; F.dVi_p = hf->dVi_p;
    F.dalpha_i_p = hf->dalpha_i_p; F.dksi_ice_p = hf->dksi_ice_p;
    F.FxOA = hf->FxOA; F.FyOA = hf->FyOA; F.torqueOA = hf->torqueOA;
    F.alive = hf->alive; F.vert_off = hf->vert_off;
    F.c0x = hf->c0x; F.c0y = hf->c0y; F.cax = hf->cax; F.cay = hf->cay;
    F.StressH = hf->StressH; F.StressCount = hf->StressCount; F.Stress = hf->Stress; F.strain = hf->strain;
    F.mcX = hf->mcX; F.mcY = hf->mcY; F.mcA = hf->mcA; F.mc_si = hf->n_mc; F.mc_sq = 1;
    F.collision_force = hf->collision_force; F.collision_torque = hf->collision_torque; F.OverlapArea = hf->OverlapArea;
    F.kill = hf->kill; F.transfer = hf->transfer;
    n = F.n;
    // The sums of the stress history are device-only state that lives across steps on the GPU; here they are kept
    // between calls while the caller steps the same arrays (g_keep_sums), else rebuilt from the history as at upload.
    if (!(g_keep_sums && g_sums_for == (const void *)hf->StressH && (int)(g_sums.size() / 8) == n)) {
      g_sums.assign((size_t)n * 8, 0.0);
      F.StressSum = g_sums.data();
      for (int i = 0; i < n; ++i) stress_sum_init(i, F);
      g_sums_for = (const void *)hf->StressH;
    }
    F.StressSum = g_sums.data();
  }
  void bind_rest(const HParams *hp, int pot_cap) {
    P.dt = hp->dt; P.HFo = hp->HFo; P.Modulus = hp->Modulus; P.mu = hp->mu; P.nu = hp->nu;
    P.doInt = hp->doInt; P.COLLISION = hp->COLLISION; P.PERIODIC = hp->PERIODIC; P.reserved = 0;
    memcpy(P.c2_boundary, hp->c2_boundary, sizeof(P.c2_boundary));
    P.fb_h = hp->fb_h; P.fb_area = hp->fb_area; P.fb_Xi = hp->fb_Xi; P.fb_Yi = hp->fb_Yi; P.fb_Ui = hp->fb_Ui;
    P.fb_Vi = hp->fb_Vi; P.fb_ksi_ice = hp->fb_ksi_ice;
    P.nxo = hp->nxo; P.nyo = hp->nyo; P.Xo = hp->Xo; P.Yo = hp->Yo; P.Uocn = hp->Uocn; P.Vocn = hp->Vocn;
    P.Uwind = hp->Uwind; P.Vwind = hp->Vwind; P.fCoriolis = hp->fCoriolis; P.turn_angle = hp->turn_angle;
    axis_extent(P.Xo, P.nxo, P.Xo_max, P.Xo_min); axis_extent(P.Yo, P.nyo, P.Yo_max, P.Yo_min);
    P.Lx = P.c2_boundary[0]; P.Ly = P.c2_boundary[1];
    for (int k = 1; k < 5; ++k) { P.Lx = szc::maxd(P.Lx, P.c2_boundary[2 * k]); P.Ly = szc::maxd(P.Ly, P.c2_boundary[2 * k + 1]); }
    boundary_summary(P);
    memset(&W, 0, sizeof(W));
    cap = 4 * n;
    W.cap_ext = cap; W.pot_cap = pot_cap < 0 ? -pot_cap : pot_cap;
    W.ext_src = alloc<int>(pool, cap); W.ext_num = alloc<int>(pool, cap); W.ext_parent = alloc<int>(pool, cap); W.ext_nv = alloc<int>(pool, cap);
    W.xg_of = alloc<int>(pool, cap); W.yg_of = alloc<int>(pool, cap);
    W.ext_x = alloc<double>(pool, cap); W.ext_y = alloc<double>(pool, cap); W.ext_rec = alloc<double>(pool, (size_t)REC_WORDS * cap); W.ext_ring = alloc<double>(pool, 2 * (size_t)RING_SLOTS * cap);
    W.flag = alloc<int>(pool, cap); W.scan = alloc<int>(pool, cap);
    W.counters = alloc<unsigned long long>(pool, C_COUNT);
    W.pot_cnt = alloc<int>(pool, cap); W.pot_list = alloc<int>(pool, (size_t)cap * W.pot_cap);
    W.inc_pre = alloc<int>(pool, cap); W.mc_cnt = alloc<int>(pool, n); W.mc_list = alloc<int>(pool, n); W.mc_list_cap = n;
    for (int i = 0; i < n; ++i) mc_count(i, F, W);
    W.pair_cnt = alloc<int>(pool, cap); W.pair_nv = alloc<i64>(pool, cap);
    W.pair_base = alloc<i64>(pool, cap + 1); W.nv_base = alloc<i64>(pool, cap + 1);
    W.inc_cnt = alloc<int>(pool, cap); W.inc_list = alloc<int>(pool, (size_t)cap * W.pot_cap);
    W.row_cnt = alloc<int>(pool, cap); W.row_off = alloc<i64>(pool, cap + 1);
    W.cf_ext = alloc<double>(pool, 2 * (size_t)cap); W.ct_ext = alloc<double>(pool, cap); W.ov_ext = alloc<double>(pool, cap);
    W.tmax = alloc<int>(pool, cap);
    W.st_ext = alloc<double>(pool, 3 * (size_t)n); W.st_flag = alloc<int>(pool, n);
    if (g_sp_active) { W.present = g_sp_masks.data(); W.owned = g_sp_masks.data() + n; }
  }

  std::vector<int> mc_empty;
  int front(bool all_pairs_broad) {
    for (int attempt = 0;; ++attempt) {
      int rc = front_once(all_pairs_broad);
      if (rc != -7 || all_pairs_broad || attempt >= 6) return rc;
      // as step_front does: grow the candidate / mirror lists to what this step asks for and run the front again
      int want = W.pot_cap * 2;
      while ((unsigned long long)want < W.counters[C_NEED_POT]) want *= 2;
      W.pot_cap = want;
      W.pot_list = alloc<int>(pool, (size_t)cap * W.pot_cap); W.inc_list = alloc<int>(pool, (size_t)cap * W.pot_cap);
      ++grown;
    }
  }
  int grown = 0;
  int front_once(bool all_pairs_broad) {
    for (int k = 0; k < C_COUNT; ++k) if (k != C_SCR_BUMP) W.counters[k] = 0;
    for (int e = 0; e < cap; ++e) { ext_clear(e, W); marks_clear(e, F); }
    for (int i = 0; i < n; ++i) ext_init(i, F, W);
    for (int i = F.nb; i < n; ++i) mc_precheck(i, F, P, W);
    if (W.counters[C_NEED_MC]) {
      mc_empty.assign(W.mc_list, W.mc_list + W.counters[C_NEED_MC]);
      return 2;
    }
    N = n;
    if (P.PERIODIC) {
      for (int dim = 0; dim < 2; ++dim) {
        int n1 = N, g = 0;
        for (int e = 0; e < n1; ++e) ghost_flag(e, dim, F, P, W);
        for (int e = 0; e < n1; ++e) { W.scan[e] = g; g += W.flag[e]; }
        for (int e = 0; e < n1; ++e) ghost_emit(e, dim, n1, F, P, W);
        N = n1 + g;
      }
    }
    if (all_pairs_broad) {
      for (int i = F.nb; i < n; ++i) broad(i, N, F, P, W);     // reals first: ghosts read their lists
      for (int i = n; i < N; ++i) broad(i, N, F, P, W);
    } else {
      std::vector<double> r(F.rmax + F.nb, F.rmax + n);
      double rm = 0, p90 = 0;
      if (!r.empty()) {
        for (double v : r) rm = szc::maxd(rm, v);
        const size_t k = (size_t)(0.9 * (double)(r.size() - 1));
        std::nth_element(r.begin(), r.begin() + k, r.end());
        p90 = r[k];
      }
      grid_levels(rm, p90, P.Lx, P.Ly, W);
      const int nc = W.lvl_base[W.n_levels];
      W.cell_cnt = alloc<int>(pool, nc + 1); W.cell_off = alloc<int>(pool, nc + 1); W.cell_fill = alloc<int>(pool, nc + 1);
      W.cell_item = alloc<CellItem>(pool, cap);
      for (int e = 0; e < N; ++e) { cell_count(e, F, W); ext_record(e, F, W); }
      int acc = 0;
      for (int c = 0; c < nc; ++c) { W.cell_off[c] = acc; acc += W.cell_cnt[c]; }
      W.cell_off[nc] = acc;
      for (int e = N - 1; e >= 0; --e) cell_fill(e, F, W);        // reversed on purpose: order must not matter
      for (int i = N - 1; i >= F.nb; --i)
        for (int c9 = 8; c9 >= 0; --c9) broad_scan(i, c9, N, F, P, W);   // any order
      std::vector<int> ftmp(2 * (size_t)W.pot_cap + 2);
      for (int i = F.nb; i < n; ++i) broad_filter(i, F, P, W, ftmp.data());   // reals first: ghosts read their lists
      for (int i = n; i < N; ++i) broad_filter(i, F, P, W, ftmp.data());
    }
    if (W.counters[C_ERR_POT]) return -7;
    for (int i = 0; i < N; ++i) pair_count(i, N, F, P, W);
    for (int e = 0; e < N; ++e) pair_room(e, W);
    if (W.counters[C_ERR_POT]) return -7;
    np = 0; nvt = 0;
    for (int i = 0; i < N; ++i) { W.pair_base[i] = np; W.nv_base[i] = nvt; np += W.pair_cnt[i]; nvt += W.pair_nv[i]; }
    W.pair_base[N] = np; W.nv_base[N] = nvt;
    W.cap_pairs = np + 1;
    W.pair_i = alloc<int>(pool, np + 1); W.pair_j = alloc<int>(pool, np + 1); W.pair_nvoff = alloc<i64>(pool, np + 1);
    W.pair_scr = alloc<i64>(pool, np + 1); W.pair_mult = alloc<int>(pool, np + 1);
    W.pair_nrows = alloc<int>(pool, np + 1); W.pair_status = alloc<int>(pool, np + 1); W.pair_flag = alloc<int>(pool, np + 1);
    W.retry_list = alloc<int>(pool, np + 1); W.pair_rowoff = alloc<i64>(pool, np + 1); W.big_list = alloc<int>(pool, np + 1); W.team_list = alloc<int>(pool, np + 1); W.wide_list = alloc<int>(pool, np + 1); W.live_list = alloc<int>(pool, np + 1); W.cold_list = alloc<int>(pool, np + 1);
    W.touch_list = alloc<int>(pool, np + 1); W.pair_nreg = alloc<int>(pool, np + 1);
    for (int i = 0; i < N; ++i) pair_expand(i, N, F, P, W);
    i64 main_bytes = (i64)SCR_PER_V * (nvt + 2 * np) + (i64)SCR_PER_PAIR * np;
    i64 reserve = main_bytes / 2 + (8 << 20);
    W.scratch_bytes = main_bytes + reserve;
    W.scratch = alloc<char>(pool, (size_t)W.scratch_bytes);
    W.counters[C_SCR_BUMP] = (unsigned long long)main_bytes;
    return 0;
  }

  // 0: general sweep only; 1: two-stage compact-record path first (the GPU's SUBZERO_FAST_PATH=1);
  // 2: single-minimum fast path, a lane per pair; 3 / 4: the team form first (16 / 32 lanes per pair, lanes as fibers:
  // the GPU default), its leftovers through the lane-per-pair fast path
  template <int G>
  struct TeamJob {
    Sim *self; i64 p; sztm::TeamBlock B; teamhost::Runtime *rt; bool cold, prepared;
    static void lane(void *a, int l) {
      TeamJob *J = (TeamJob *)a;
      sztm::Team<G> T; T.lane = l; T.rt = J->rt;
      narrow_team(J->p, T, &J->B, J->self->F, J->self->P, J->self->W, J->cold, J->prepared);
    }
  };
  template <int G>
  void team_pass(i64 lo, i64 hi, const int *list, bool cold = false) {
    static thread_local teamhost::Runtime rt;
    static thread_local TeamJob<G> J;
    J.self = this; J.rt = &rt; J.cold = cold; J.prepared = list != nullptr;   // (a list: the pairs pair_classify has prepared)
    for (i64 p = lo; p < hi; ++p) {
      J.p = list ? (i64)list[p] : p;
      rt.run(G, &TeamJob<G>::lane, &J);
      if (!teamhost::exited_together(rt)) { fprintf(stderr, "narrow_team: the lanes of a team did not return together (pair %lld)\n", (long long)p); abort(); }
    }
    if (!list) g_team_pairs += hi - lo;
  }
  int narrow_range(i64 lo, i64 hi) {
    const bool two_stage = (g_narrow_mode == 1);
    std::vector<char> fast(szv::workspace_bytes_compact() + 64);
    void *fw = (void *)(((uintptr_t)fast.data() + 15) & ~(uintptr_t)15);
    W.counters[C_TOUCH] = 0;
    if (two_stage) {
      for (i64 p = hi - 1; p >= lo; --p) narrow_a(p, F, P, W, fw);           // any order
      const i64 nt = (i64)W.counters[C_TOUCH];
      for (i64 t = 0; t < nt; ++t) narrow_b(t, F, P, W, fw);
    } else {
      szx::FastBlock blk;
      if (g_narrow_mode == 3 || g_narrow_mode == 4) {
        // mode 3: as the GPU's default path -- the pairs sorted by pair_classify, 16 lanes for the live list, 32 for the
        // wide one; mode 4: 32 lanes for every pair of the range, unsorted (the sharded step)
        W.counters[C_TEAM] = 0; W.counters[C_WIDE] = 0; W.counters[C_LIVE] = 0;
        if (g_narrow_mode == 3 && lo == 0 && hi == np) {
          W.counters[C_COLD] = 0;
          for (i64 p = hi - 1; p >= lo; --p) pair_classified(p, pair_classify(p, 0, P, W), W);   // any order (no contact rows of a step before here: all cold)
          std::vector<int> ll(W.cold_list, W.cold_list + W.counters[C_COLD]), wl(W.wide_list, W.wide_list + W.counters[C_WIDE]);
          g_team_wide += (long long)wl.size();
          team_pass<32>(0, (i64)wl.size(), wl.data());
          team_pass<16>(0, (i64)ll.size(), ll.data(), (g_cold_toggle++ & 1) != 0);   // (the certificate of the cold list on every other step: with and without it, same results)
          g_team_pairs += hi - lo;
        } else team_pass<32>(lo, hi, nullptr);
        g_team_deferred += (long long)W.counters[C_TEAM];
        std::vector<int> tl(W.team_list, W.team_list + W.counters[C_TEAM]);
        for (size_t t = 0; t < tl.size(); ++t) narrow_convex((i64)t, tl.data(), F, P, W, &blk);
      } else
      for (i64 p = lo; p < hi; ++p) {
        if (g_narrow_mode == 2) narrow_convex(p, nullptr, F, P, W, &blk);
        else narrow(p, nullptr, 1, F, P, W);
      }
      // second fast pass on the 32-vertex block, as step_narrow does
      szx::FastBlockBig big;
      std::vector<int> bl(W.big_list, W.big_list + W.counters[C_BIG]);
      g_big_pairs += (long long)bl.size();
      for (size_t t = 0; t < bl.size(); ++t) narrow_convex((i64)t, bl.data(), F, P, W, &big);
    }
    retried = (i64)W.counters[C_RETRY];
    int mult = (g_narrow_mode != 0) ? 0 : 1;   // the fast path's leftovers start at the 1x general tier
    std::vector<int> list;
    while (W.counters[C_RETRY] > 0) {
      mult = mult ? mult * 8 : 1;
      if (mult > 512) return -5;
      list.assign(W.retry_list, W.retry_list + W.counters[C_RETRY]);
      W.counters[C_RETRY] = 0;
      // as step_narrow does: the fast path's leftovers at 1x go through k_narrow_lds (16 KB of LDS per lane)
      const bool lds = g_narrow_mode >= 2 && mult == 1;
      std::vector<char> slice(16384 + 64);              // stands in for the lane's slice of LDS
      void *sl = (void *)(((uintptr_t)slice.data() + 15) & ~(uintptr_t)15);
      for (size_t t = 0; t < list.size(); ++t) {
        if (lds) narrow_lds((i64)t, list.data(), mult, F, P, W, sl, 16384);
        else narrow((i64)t, list.data(), mult, F, P, W);
      }
    }
    if (W.counters[C_ERR_ARENA]) return -5;
    if (W.counters[C_ERR_CLIP]) return -8;
    return 0;
  }

  int back_a() {
    for (i64 p = np - 1; p >= 0; --p) incoming(p, W);        // reversed on purpose: order must not matter
    if (W.counters[C_ERR_INC]) return -7;
    { std::vector<int> rtmp(2 * (size_t)W.pot_cap + 2); for (int i = 0; i < N; ++i) row_count(i, W, rtmp.data()); }
    i64 nr = 0;
    for (int i = 0; i < N; ++i) { W.row_off[i] = nr; nr += W.row_cnt[i]; }
    W.row_off[N] = nr;
    W.cap_rows = nr + 1;
    W.rows = alloc<double>(pool, 7 * (size_t)(nr + 1));
    for (int i = 0; i < N; ++i) row_fill(i, N, F, P, W);
    for (int i = 0; i < N; ++i) transfer_collect(i, F, W);
    for (int i = 0; i < cap; ++i) transfer_apply(i, F, W);
    for (int i = F.nb; i < n; ++i) integrate_pre(i, F, P, W);
    return 0;
  }
  void forcing(int first, int last) { for (int i = first; i < last; ++i) forcing_serial(i, F, P, W); }
  int back_b() {
    for (int i = F.nb; i < n; ++i) integrate_post(i, F, P, W);
    if (W.counters[C_ERR_MC]) return -9;
    for (int i = 0; i < F.nb; ++i) { F.collision_force[2 * i] = 0; F.collision_force[2 * i + 1] = 0; F.collision_torque[i] = 0; F.OverlapArea[i] = 0; }
    return 0;
  }
  int back() {
    int rc = back_a();
    if (rc) return rc;
    forcing(F.nb, n);
    return back_b();
  }
  void force_range(int r, int w, int &first, int &last) const {
    const long long m = n - F.nb;
    first = F.nb + (int)(m * r / w); last = F.nb + (int)(m * (r + 1) / w);
  }
  int force_cnt_max(int w) const { return (n - F.nb + w - 1) / w; }
  int force_exchange = 0;

  int outputs(int32_t *inter_off, double *inter, int64_t inter_cap, int32_t *pair_i, int32_t *pair_j,
              int64_t pair_cap, int64_t *npairs_out, int32_t *n_ext) {
    *n_ext = N;
    int64_t k = 0;
    for (i64 p = 0; p < np; ++p) {
      if (W.pair_j[p] == WALL) continue;
      if (k < pair_cap) { pair_i[k] = W.pair_i[p] + 1; pair_j[k] = W.pair_j[p] + 1; }
      ++k;
    }
    *npairs_out = k;
    if (k > pair_cap) return -1;
    inter_off[0] = 0;
    if (W.row_off[n] > inter_cap) return -1;
    for (int i = 0; i < n; ++i) inter_off[i + 1] = (int32_t)W.row_off[i + 1];
    memcpy(inter, W.rows, sizeof(double) * 7 * (size_t)W.row_off[n]);
    return 0;
  }
};

int g_last_grown = 0, g_last_pot_cap = 0, g_last_levels = 0;
std::vector<int> g_last_mc_empty, g_last_ghost_src, g_last_ghost_parent;
std::vector<double> g_last_ghost_x, g_last_ghost_y;
Sim g_sim; // the sharded entry points keep their state between calls, as the GPU context does

} // namespace

extern "C" {

// returns 0, or: -1 output capacity, -5 workspace, -7 candidate-list overflow, -8 clip error; 2 = floes need new
// Monte-Carlo points (nothing changed; host_last_mc_empty lists them), as szStep's SZ_STEP_NEEDS_MC.
// pot_cap < 0 selects the all-pairs broad-phase with capacity -pot_cap.
int host_step(HFloes *hf, const HParams *hp, int32_t *inter_off, double *inter, int64_t inter_cap,
              int32_t *pair_i, int32_t *pair_j, int64_t pair_cap, int64_t *npairs_out, int32_t *n_ext,
              int32_t pot_cap, int64_t *retried) {
  Sim s;
  s.bind(hf, hp, pot_cap);
  int rc;
  rc = s.front(pot_cap < 0);
  g_last_grown = s.grown; g_last_pot_cap = s.W.pot_cap; g_last_levels = s.W.n_levels; g_last_mc_empty = s.mc_empty;
  if (rc) return rc;
  g_last_ghost_src.assign(s.W.ext_src + s.n, s.W.ext_src + s.N); g_last_ghost_parent.assign(s.W.ext_parent + s.n, s.W.ext_parent + s.N);
  g_last_ghost_x.assign(s.W.ext_x + s.n, s.W.ext_x + s.N); g_last_ghost_y.assign(s.W.ext_y + s.n, s.W.ext_y + s.N);
  if ((rc = s.narrow_range(0, s.np))) return rc;
  *retried = s.retried;
  if ((rc = s.back())) return rc;
  return s.outputs(inter_off, inter, inter_cap, pair_i, pair_j, pair_cap, npairs_out, n_ext);
}

// what the last host_step did about capacities: [times the lists grew, final capacity, grid levels, floes needing new MC points]
void host_last_front(int *out4) { out4[0] = g_last_grown; out4[1] = g_last_pot_cap; out4[2] = g_last_levels; out4[3] = (int)g_last_mc_empty.size(); }
int host_last_mc_empty(int *ids, int cap) {
  int n = (int)g_last_mc_empty.size();
  std::sort(g_last_mc_empty.begin(), g_last_mc_empty.end());
  for (int k = 0; k < n && k < cap; ++k) ids[k] = g_last_mc_empty[k];
  return n;
}

// the ghost list of the last host_step (as szGhostsDownload): 1-based src / parent, positions
int host_last_ghosts(int *src, int *parent, double *x, double *y, int cap) {
  const int g = (int)g_last_ghost_src.size();
  for (int k = 0; k < g && k < cap; ++k) { src[k] = g_last_ghost_src[k] + 1; parent[k] = g_last_ghost_parent[k] + 1; x[k] = g_last_ghost_x[k]; y[k] = g_last_ghost_y[k]; }
  return g;
}

// ---- spatially sharded form, mirroring szSpatialInit / Pack / Unpack (then host_step as usual) ----
void host_spatial_axis(int axis) { g_sp_axis = axis ? 1 : 0; }
int host_spatial_init(HFloes *hf, int rank, int world, const double *edges, double halo) {
  const int n = hf->n;
  double rm = 0;
  for (int i = 0; i < n; ++i) rm = szc::maxd(rm, hf->rmax[i]);
  if (!(halo > 0)) halo = 2 * rm;          // reach = 2 max rmax
  if (halo < 2 * rm) return -2;
  if (world > 1) for (int r = 0; r < world; ++r) if (edges[r + 1] - edges[r] < halo) return -2;
  g_sp_rank = rank; g_sp_world = world; g_sp_halo = halo; g_sp_rmax_max = rm;
  g_sp_edges.assign(edges, edges + world + 1);
  g_sp_masks.assign(2 * (size_t)n, 0);
  for (int i = 0; i < n; ++i) {
    const double x = g_sp_axis ? hf->Yi[i] : hf->Xi[i];
    const bool nan = x != x || hf->Xi[i] != hf->Xi[i];
    const bool mine = nan ? rank == 0 : sp_strip_of(x, edges, world) == rank;
    g_sp_masks[i] = mine; g_sp_masks[(size_t)n + i] = mine;
  }
  g_sp_active = true;
  return 0;
}
void host_spatial_off(void) { g_sp_active = false; }
void host_spatial_pause(int paused) { g_sp_active = !paused && !g_sp_masks.empty(); }   // (a test's single-process reference step runs unmasked)
int64_t host_spatial_message_bytes(HFloes *hf, int64_t cap_rec, int64_t cap_dbl, int64_t cap_migr) {
  return (int64_t)sp_msg_bytes(cap_rec, cap_dbl, cap_migr, sp_bulk_row(hf->n_stress, hf->n_mc));
}
static SpatialGeom host_geom(int periodic) {
  SpatialGeom g;
  g.rank = g_sp_rank; g.world = g_sp_world; g.periodic = periodic;
  g.lo = g_sp_edges[g_sp_rank]; g.hi = g_sp_edges[g_sp_rank + 1]; g.halo = g_sp_halo; g.rmax_max = g_sp_rmax_max;
  g.has_left = g_sp_world > 1 && (periodic || g_sp_rank > 0);
  g.has_right = g_sp_world > 1 && (periodic || g_sp_rank < g_sp_world - 1);
  g.axis = g_sp_axis;
  return g;
}
// info: [left rec, dbl, migr, right rec, dbl, migr, need rec, dbl, migr, overflow, has_left, has_right, n_owned, n_present]
int host_spatial_pack(HFloes *hf, int periodic, void *left, void *right, int64_t cap_rec, int64_t cap_dbl, int64_t cap_migr, int64_t *info) {
  Sim &s = g_sim;
  s.bind_floes(hf);
  std::vector<int> mc((size_t)hf->n);
  s.W.mc_cnt = mc.data();
  for (int i = 0; i < hf->n; ++i) mc_count(i, s.F, s.W);
  const size_t row = sp_bulk_row(hf->n_stress, hf->n_mc);
  SpatialMsg l = sp_msg_view((char *)left, cap_rec, cap_dbl, cap_migr, row), r = sp_msg_view((char *)right, cap_rec, cap_dbl, cap_migr, row);
  for (int k = 0; k < SP_HDR; ++k) { l.hdr[k] = 0; r.hdr[k] = 0; }
  const SpatialGeom g = host_geom(periodic);
  unsigned char *present = g_sp_masks.data(), *owned = present + hf->n;
  const std::vector<unsigned char> owned_before(owned, owned + hf->n);
  for (int i = hf->n - 1; i >= 0; --i) sp_classify(i, s.F, s.W, g, g_sp_edges.data(), present, owned, l, r);   // any order
  sp_seal(l); sp_seal(r);
  if (l.hdr[3] != SP_OK || r.hdr[3] != SP_OK) std::copy(owned_before.begin(), owned_before.end(), owned);     // (as szSpatialPack: nothing was handed over)
  SpatialMsg *ms[2] = {&l, &r};
  for (int side = 0; side < 2; ++side) {
    SpatialMsg &m = *ms[side];
    for (long long k = 0; k < m.hdr[0]; ++k) {
      const double *rec = m.dbl + m.rec_off[k];
      if (rec[1] == 0.0) continue;
      char *rowp = m.bulk + (long long)rec[3] * m.bulk_row;
      for (size_t w = 0; w < (size_t)m.bulk_row / 8; ++w) sp_bulk_word((int)rec[0], w, true, s.F, rowp, s.W);
    }
  }
  info[0] = l.hdr[0]; info[1] = l.hdr[1]; info[2] = l.hdr[2]; info[3] = r.hdr[0]; info[4] = r.hdr[1]; info[5] = r.hdr[2];
  info[6] = std::max(l.hdr[4], r.hdr[4]); info[7] = std::max(l.hdr[5], r.hdr[5]); info[8] = std::max(l.hdr[6], r.hdr[6]);
  info[9] = (l.hdr[3] != SP_OK || r.hdr[3] != SP_OK) ? 1 : 0; info[10] = g.has_left; info[11] = g.has_right;
  s.W.mc_cnt = nullptr;
  return 0;
}
int host_spatial_unpack(HFloes *hf, const void *left, const void *right, int64_t cap_rec, int64_t cap_dbl, int64_t cap_migr, int64_t *info) {
  Sim &s = g_sim;
  s.bind_floes(hf);
  std::vector<int> mc((size_t)hf->n);
  s.W.mc_cnt = mc.data();
  const size_t row = sp_bulk_row(hf->n_stress, hf->n_mc);
  unsigned char *present = g_sp_masks.data(), *owned = present + hf->n;
  const void *bufs[2] = {left, right};
  for (int k = 0; k < 14; ++k) info[k] = 0;
  for (int side = 0; side < 2; ++side) {
    if (!bufs[side]) continue;
    SpatialMsg m = sp_msg_view((char *)bufs[side], cap_rec, cap_dbl, cap_migr, row);
    info[6] = std::max<int64_t>(info[6], m.hdr[4]); info[7] = std::max<int64_t>(info[7], m.hdr[5]); info[8] = std::max<int64_t>(info[8], m.hdr[6]);
    if (m.hdr[3] != SP_OK) { info[9] = 1; continue; }
    info[3 * side] = m.hdr[0]; info[3 * side + 1] = m.hdr[1]; info[3 * side + 2] = m.hdr[2];
    for (long long k = m.hdr[0] - 1; k >= 0; --k) {                     // any order
      const double *rec = m.dbl + m.rec_off[k];
      const int i = (int)rec[0];
      const bool was_owned = owned[i] != 0;
      sp_take(k, s.F, s.W, m, present, owned);
      if (rec[1] != 0.0 && !was_owned) {
        char *rowp = m.bulk + (long long)rec[3] * m.bulk_row;
        for (size_t w = 0; w < (size_t)m.bulk_row / 8; ++w) sp_bulk_word(i, w, false, s.F, rowp, s.W);
      }
    }
  }
  for (int i = 0; i < hf->n; ++i) { info[12] += owned[i]; info[13] += present[i]; }
  s.W.mc_cnt = nullptr;
  return 0;
}
void host_spatial_owned(unsigned char *out, int n) { for (int i = 0; i < n; ++i) out[i] = g_sp_masks[(size_t)n + i]; }

// ---- sharded form, mirroring szStepShardBegin / Export / Finish ----
int64_t host_shard_segment_bytes(int64_t seg_max, int64_t rows_max) { return (int64_t)seg_bytes_for(seg_max, rows_max); }

// info: [n_pairs, seg_lo, seg_hi, seg_max, seg_rows (-1: in the segment header), status, replicated_failure]
int host_shard_begin(HFloes *hf, const HParams *hp, int rank, int world, int64_t *info) {
  Sim &s = g_sim;
  s.bind(hf, hp, 32);
  s.rank = rank; s.world = world;
  int rc = s.front(false);
  info[6] = rc ? 1 : 0;
  if (!rc) {
    shard_range(s.np, rank, world, s.seg_lo, s.seg_hi);
    rc = s.narrow_range(s.seg_lo, s.seg_hi);
  }
  s.x_off.assign((size_t)(s.seg_hi - s.seg_lo) + 1, 0);
  i64 acc = 0;
  if (!rc) for (i64 k = 0; k < s.seg_hi - s.seg_lo; ++k) { s.x_off[k] = acc; acc += s.W.pair_nrows[s.seg_lo + k]; }
  s.seg_rows = acc;
  info[0] = s.np; info[1] = s.seg_lo; info[2] = s.seg_hi; info[3] = shard_max(s.np, world); info[4] = -1; info[5] = rc;
  return rc;
}
int host_shard_export(void *send, int64_t seg_max, int64_t rows_max, int32_t status) {
  Sim &s = g_sim;
  const i64 ns = status ? 0 : s.seg_hi - s.seg_lo;
  if (ns > seg_max) return -1;
  char *base = (char *)send;
  seg_header(base, ns, status ? 0 : s.seg_rows, status, rows_max);
  Xfer X = seg_view(base, seg_max, rows_max);
  for (i64 k = 0; k < ns; ++k) export_pair(k, s.seg_lo, s.x_off.data(), s.F, s.W, X);
  return 0;
}
// Finish: headers, import + contact rows + sums; on a forcing step with world > 1 only this rank's floe range is
// forced and *fx = 1 (then host_shard_forcing / host_shard_end complete the step).  Return codes as szStepShardFinish.
int host_shard_finish(const void *recv, int64_t seg_max, int64_t rows_max, int32_t *fx, int64_t *rows_needed) {
  Sim &s = g_sim;
  const size_t seg_bytes = seg_bytes_for(seg_max, rows_max);
  i64 need = 0;
  for (int r = 0; r < s.world; ++r) {
    const i64 *h = (const i64 *)((const char *)recv + (size_t)r * seg_bytes);
    if (h[2] != 0) return -7;
    i64 lo, hi;
    shard_range(s.np, r, s.world, lo, hi);
    if (h[0] != hi - lo || h[3] != rows_max) return -6;
    if (h[1] > need) need = h[1];
  }
  *rows_needed = need;
  if (need > rows_max) return 1;
  s.W.g_rows = (double *)recv;
  for (i64 p = s.np - 1; p >= 0; --p) import_pair(p, s.np, s.world, seg_max, rows_max, (char *)recv, s.W);   // any order
  int rc = s.back_a();
  s.W.g_rows = nullptr;
  if (rc) return rc;
  s.force_exchange = (s.P.doInt && s.world > 1) ? 1 : 0;
  *fx = s.force_exchange;
  if (s.force_exchange) {
    int f0, f1;
    s.force_range(s.rank, s.world, f0, f1);
    s.forcing(f0, f1);
    return 0;
  }
  s.forcing(s.F.nb, s.n);
  return s.back_b();
}
int64_t host_shard_forcing_bytes(int world) { return 24 * (int64_t)g_sim.force_cnt_max(world); }
int host_shard_forcing(void *send) {
  Sim &s = g_sim;
  int f0, f1;
  s.force_range(s.rank, s.world, f0, f1);
  const int cm = s.force_cnt_max(s.world);
  double *d = (double *)send;
  for (int k = 0; k < f1 - f0; ++k) { d[k] = s.F.FxOA[f0 + k]; d[cm + k] = s.F.FyOA[f0 + k]; d[2 * (size_t)cm + k] = s.F.torqueOA[f0 + k]; }
  return 0;
}
int host_shard_end(const void *recv) {
  Sim &s = g_sim;
  if (!s.force_exchange) return 0;
  const int cm = s.force_cnt_max(s.world);
  for (int r = 0; r < s.world; ++r) {
    if (r == s.rank) continue;
    int f0, f1;
    s.force_range(r, s.world, f0, f1);
    const double *d = (const double *)recv + 3 * (size_t)cm * r;
    for (int k = 0; k < f1 - f0; ++k) { s.F.FxOA[f0 + k] = d[k]; s.F.FyOA[f0 + k] = d[cm + k]; s.F.torqueOA[f0 + k] = d[2 * (size_t)cm + k]; }
  }
  s.force_exchange = 0;
  return s.back_b();
}
int host_shard_outputs(int32_t *inter_off, double *inter, int64_t inter_cap, int32_t *pair_i, int32_t *pair_j,
                       int64_t pair_cap, int64_t *npairs_out, int32_t *n_ext) {
  return g_sim.outputs(inter_off, inter, inter_cap, pair_i, pair_j, pair_cap, npairs_out, n_ext);
}

} // extern "C"
