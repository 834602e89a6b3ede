// spatial_core.hpp -- the spatially sharded step (BASELINE north_star: "floes are spatially binned and
// sharded across the GPUs of one node with RCCL halo exchange of ghost-floe state over xGMI") as kernel bodies.
//
// Partition.  The domain is cut along one axis -- x, or y when the box is longer that way (SpatialGeom::axis: the Nares
// channel of BASELINE configs[4] is 2e5 x 1e6 m) -- into `world` strips whose edges lie on multiples of the broad-phase bin
// (>= 2 max rmax) and balance the floe count; a floe belongs to the rank whose strip holds its centroid.  Every rank
// keeps the floe arrays at their GLOBAL size and index -- a floe's index, and with it every index comparison of
// floe_interactions_all.m (j > i, ascending partner order, the mems filter), means the same on every rank -- but only
// the rows of the floes it owns, and of the halo it received this step, are current.  The 0.5 GB of stress history
// and Monte-Carlo points exist once: in the owner's rows; they travel only when a floe changes strip.
//
// One exchange per step, before the pair phase, with the left and the right neighbour (the strips form a ring when the
// domain is periodic): the floes that can be a partner of a floe beyond the shared edge -- floe i within rmax_i + max rmax
// of it (the candidate test of floe_interactions_all.m:101-103 is |c_i - c_j| < rmax_i + rmax_j, and c_j lies beyond the
// edge) -- 26 scalars + ring, 208 + 16 nv bytes each -- and the few floes whose centroid crossed the edge, with their
// whole state.  (Rounds 2-3 sent everything within 4 max rmax: "a rank must evaluate the candidate list of every
// lower-indexed partner of its floes".  It need not evaluate that list WHOLE: of a halo floe k's pairs only those with
// an owned floe are relevant (pair_relevant), and whether (k, i) is a candidate depends on k, i and the images of i
// alone -- the `mems` rule of :101-113 compares FloeNums, and entries of other floes never carry i's number.  So k's
// other partners may be absent.)  With the halo in place each rank runs the unchanged pipeline on what it knows: ghosts (the reference's rule, from the rings it has),
// grid, candidate lists, the pairs that involve a floe it owns (a pair between two ranks' floes is evaluated by both:
// same inputs, same arithmetic, same rows -- nothing is sent back), contact rows, and calc_trajectory for its own floes.
// Results for owned floes are bit-identical to the single-GPU step (tests/test_spatial_*.py).
#pragma once
#include "step_core.hpp"

namespace szs {

enum { SP_REC_BASE = 26, SP_EXTRA = 21, SP_HDR = 8 };   // doubles per record before the ring / migrant extras; header int64s
enum { SP_OK = 0, SP_OVERFLOW = 1 };

struct SpatialGeom {
  int rank, world, periodic;
  double lo, hi;        // my strip [lo, hi)
  double halo;          // width of the halo region along the strip axis for a floe of the largest rmax (>= 2 max rmax)
  double rmax_max;      // the largest rmax of the field: a floe of radius r is halo material within halo - (rmax_max - r) of an edge
  int has_left, has_right;
  int axis;             // 0: strips along x (edges are x values), 1: along y
};

// one message (to one neighbour) as the device sees it
struct SpatialMsg {
  long long *hdr;       // [SP_HDR]: records, doubles used, migrants, status, records / doubles / migrants needed
  int *rec_off;         // [cap_rec] start of each record in `dbl`
  double *dbl;          // [cap_dbl]
  char *bulk;           // [cap_migr][bulk_row] stress history + Monte-Carlo points of the migrants
  long long cap_rec, cap_dbl, cap_migr, bulk_row;
};
SZ_HD size_t sp_bulk_row(int n_stress, int n_mc) { return ((size_t)n_stress * 32 + (size_t)n_mc * 17 + 7) & ~(size_t)7; }
SZ_HD size_t sp_msg_bytes(long long cap_rec, long long cap_dbl, long long cap_migr, size_t bulk_row) {
  return 8 * (size_t)SP_HDR + (((size_t)cap_rec * 4 + 7) & ~(size_t)7) + 8 * (size_t)cap_dbl + (size_t)cap_migr * bulk_row;
}
SZ_HD SpatialMsg sp_msg_view(char *base, long long cap_rec, long long cap_dbl, long long cap_migr, size_t bulk_row) {
  SpatialMsg m;
  m.hdr = (long long *)base;
  m.rec_off = (int *)(base + 8 * SP_HDR);
  m.dbl = (double *)(base + 8 * SP_HDR + (((size_t)cap_rec * 4 + 7) & ~(size_t)7));
  m.bulk = (char *)(m.dbl + cap_dbl);
  m.cap_rec = cap_rec; m.cap_dbl = cap_dbl; m.cap_migr = cap_migr; m.bulk_row = (long long)bulk_row;
  return m;
}

// strip index of a centroid (all ranks evaluate the same expression on the same edges)
SZ_HD int sp_strip_of(double x, const double *edges, int world) {
  int s = 0;
  while (s + 1 < world && x >= edges[s + 1]) ++s;
  return s;
}

// the 20 scalar fields in the order of a record
#define SZ_SP_SCALARS(F) {F.Xi, F.Yi, F.alpha_i, F.Ui, F.Vi, F.ksi_ice, F.h, F.area, F.mass, F.inertia_moment, F.rmax, \
                          F.dXi_p, F.dYi_p, F.dUi_p, F.dVi_p, F.dalpha_i_p, F.dksi_ice_p, F.FxOA, F.FyOA, F.torqueOA}

// Floe i into message m (halo record, or full record when `migrant`).  Counts keep running past the capacities so
// that the header tells how much room the edge needs.
SZ_HD bool sp_put(int i, bool migrant, const Floes &F, const Work &W, SpatialMsg &m) {
  const int v0 = F.vert_off[i], nv = F.vert_off[i + 1] - v0;
  const int len = SP_REC_BASE + 2 * nv + (migrant ? (int)SP_EXTRA : 0);
  const long long k = (long long)atomic_add((unsigned long long *)&m.hdr[4], 1ULL);
  const long long off = (long long)atomic_add((unsigned long long *)&m.hdr[5], (unsigned long long)len);
  long long slot = -1;
  if (migrant) slot = (long long)atomic_add((unsigned long long *)&m.hdr[6], 1ULL);
  if (k >= m.cap_rec || off + len > m.cap_dbl || slot >= m.cap_migr) { m.hdr[3] = SP_OVERFLOW; return false; }
  m.rec_off[k] = (int)off;
  double *r = m.dbl + off;
  r[0] = (double)i; r[1] = migrant ? 1.0 : 0.0; r[2] = (double)nv; r[3] = (double)slot;
  double *sc[20] = SZ_SP_SCALARS(F);
  for (int q = 0; q < 20; ++q) r[4 + q] = sc[q][i];
  r[24] = (double)F.alive[i]; r[25] = (double)F.StressCount[i];
  for (int q = 0; q < nv; ++q) { r[SP_REC_BASE + q] = F.cax[v0 + q]; r[SP_REC_BASE + nv + q] = F.cay[v0 + q]; }
  if (migrant) {
    double *x = r + SP_REC_BASE + 2 * nv;
    for (int q = 0; q < 4; ++q) { x[q] = F.Stress[4 * i + q]; x[4 + q] = F.strain[4 * i + q]; }
    x[8] = F.collision_force[2 * i]; x[9] = F.collision_force[2 * i + 1]; x[10] = F.collision_torque[i]; x[11] = F.OverlapArea[i];
    for (int q = 0; q < 8; ++q) x[12 + q] = F.StressSum[8 * i + q];
    x[20] = (double)W.mc_cnt[i];
  }
  return true;
}
// (after every sp_put of the message) what fitted
SZ_HD void sp_seal(SpatialMsg &m) {
  const bool ok = m.hdr[3] == SP_OK;
  m.hdr[0] = ok ? m.hdr[4] : 0; m.hdr[1] = ok ? m.hdr[5] : 0; m.hdr[2] = ok ? m.hdr[6] : 0;
}

// Where owned floe i goes this step: into the halo messages of the edges it is near, and -- when its centroid has
// left the strip -- to the neighbour that now owns it (this rank keeps it `present` for the step: it has just
// integrated it, the state is current).
SZ_HD void sp_classify(int i, const Floes &F, const Work &W, const SpatialGeom &g, const double *edges, unsigned char *present,
                       unsigned char *owned, SpatialMsg &left, SpatialMsg &right) {
  if (!owned[i]) { present[i] = 0; return; }
  present[i] = 1;
  const double x = g.axis ? F.Yi[i] : F.Xi[i];       // (the coordinate the strips are cut along)
  if (!F.alive[i] || x != x || F.Xi[i] != F.Xi[i]) return;   // dead or sacked floes stay where they are, nobody needs them
  const int s = sp_strip_of(x, edges, g.world);
  // how near an edge this floe is halo material: a neighbour's floe j can be its partner only within rmax_i + rmax_j <=
  // rmax_i + rmax_max of it (floe_interactions_all.m:101-103), and j's centroid lies beyond the edge
  const double need = g.halo - (g.rmax_max - F.rmax[i]);
  const int lrank = (g.rank + g.world - 1) % g.world, rrank = (g.rank + 1) % g.world;
  if (s != g.rank && g.world > 1) {
    const bool to_right = (s == rrank) && g.has_right && !(s == lrank && g.has_left && x < g.lo);
    const bool to_left = !to_right && (s == lrank) && g.has_left;
    if (to_right || to_left) {
      if (!sp_put(i, true, F, W, to_right ? right : left)) return;   // (no room: it stays mine, the caller is told)
      owned[i] = 0;
      // (and it may be needed as halo on the other side as well)
      if (to_right && g.has_left && x - g.lo < need) sp_put(i, false, F, W, left);
      if (to_left && g.has_right && g.hi - x < need) sp_put(i, false, F, W, right);
      return;
    }
  }
  if (g.has_left && x - g.lo < need) sp_put(i, false, F, W, left);
  if (g.has_right && g.hi - x < need) sp_put(i, false, F, W, right);
}

// record k of a received message into the arrays (the bulk rows of migrants are copied by sp_take_bulk)
SZ_HD void sp_take(long long k, const Floes &F, Work &W, const SpatialMsg &m, unsigned char *present, unsigned char *owned) {
  const double *r = m.dbl + m.rec_off[k];
  const int i = (int)r[0], nv = (int)r[2];
  const bool migrant = r[1] != 0.0;
  const int v0 = F.vert_off[i];
  if (i < 0 || i >= F.n || nv != F.vert_off[i + 1] - v0) return;   // (cannot happen between ranks that uploaded the same array)
  if (owned[i]) return;                                            // my own state is the current one
  double *sc[20] = SZ_SP_SCALARS(F);
  for (int q = 0; q < 20; ++q) sc[q][i] = r[4 + q];
  F.alive[i] = (int)r[24]; F.StressCount[i] = (int)r[25];
  for (int q = 0; q < nv; ++q) { F.cax[v0 + q] = r[SP_REC_BASE + q]; F.cay[v0 + q] = r[SP_REC_BASE + nv + q]; }
  present[i] = 1;
  if (migrant) {
    const double *x = r + SP_REC_BASE + 2 * nv;
    for (int q = 0; q < 4; ++q) { F.Stress[4 * i + q] = x[q]; F.strain[4 * i + q] = x[4 + q]; }
    F.collision_force[2 * i] = x[8]; F.collision_force[2 * i + 1] = x[9]; F.collision_torque[i] = x[10]; F.OverlapArea[i] = x[11];
    for (int q = 0; q < 8; ++q) F.StressSum[8 * i + q] = x[12 + q];
    W.mc_cnt[i] = (int)x[20];
    owned[i] = 1;
  }
}
// byte `b` of the bulk row of migrant record k: [StressH][mcX][mcY][mcA]
SZ_HD void sp_bulk_byte_range(const Floes &F, size_t &sh, size_t &mx, size_t &my, size_t &ma) {
  sh = (size_t)F.n_stress * 32; mx = sh + (size_t)F.n_mc * 8; my = mx + (size_t)F.n_mc * 8; ma = my + (size_t)F.n_mc;
}
// 8-byte word w of a migrant's bulk row, out of (pack) or into (take) the floe's rows
SZ_HD void sp_bulk_word(int i, size_t w, bool pack, const Floes &F, char *row, Work &W) {
  size_t sh, mx, my, ma;
  sp_bulk_byte_range(F, sh, mx, my, ma);
  const size_t b = 8 * w;
  char *dev;
  size_t n = 8;
  if (b < sh) dev = (char *)(F.StressH + (size_t)i * F.n_stress * 4) + b;
  else if (b < mx) dev = (char *)(F.mcX + (size_t)i * F.n_mc) + (b - sh);
  else if (b < my) dev = (char *)(F.mcY + (size_t)i * F.n_mc) + (b - mx);
  else if (b < ma) { dev = (char *)(F.mcA + (size_t)i * F.n_mc) + (b - my); if (ma - b < 8) n = ma - b; }
  else return;
  for (size_t q = 0; q < n; ++q) { if (pack) row[b + q] = dev[q]; else dev[q] = row[b + q]; }
  (void)W;
}

} // namespace szs
