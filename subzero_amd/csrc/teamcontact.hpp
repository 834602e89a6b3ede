// teamcontact.hpp -- one candidate pair of the narrow phase by a TEAM of lanes (team.hpp): the contact model of the
// reference's collisions/floe_interactions.m:1-199 with the work of every loop over vertices, edges and edge pairs
// spread over the lanes, on top of the team form of the single-minimum intersection (teamclip.hpp).
//
// It is contact.hpp's pair_contact_fast / pair_finish (one lane per pair) restated, expression for expression:
//   * both rings sit in the team's block of LDS in metres for the whole model; the two overlap regions too, converted
//     to metres once (every use of a region coordinate in the model is from_fixed of it);
//   * InterX (InterX.m:54-81): lane j tests segment j of ring 2 against every segment of ring 1, the few segment pairs
//     that pass the first sign test are finished one per lane and made a sorted set by ranking;
//   * sums whose order the result depends on (region area, centroid, polyarea, the sums of :117-137) are formed in the
//     reference's order from terms the lanes computed side by side; sums of small integers and minima are order-free;
//   * inpolygon and the 1e-6 m clearance certificates run an edge per lane.
// What the fast path does not cover comes back as ST_DEFER (ST_DEFER_BIG) exactly as from pair_contact_fast: the pair
// is redone by the lane-per-pair kernels, so every pair still goes through arithmetic that is checked against the oracle.
#pragma once
#include "contact.hpp"
#include "teamclip.hpp"

namespace sztm {

using szc::absd;
using szc::maxd;
using szc::mind;
using szc::hyp;

enum { T_CAPP = 8, T_CAPR = 16 };   // crossings kept (sorted set) / segment pairs that reach the second test
enum { ST_DEFER_WIDE = 103 };        // "redo this pair with a wider team": the two rings have more vertices than the team has lanes
enum { BODY_X = 0, BODY_Y, BODY_UI, BODY_VI, BODY_KSI, BODY_H, BODY_AREA, BODY_WORDS = 8 };   // (the order of step_core.hpp's records)

struct TeamBlock {
  double mx[2][T_NV], my[2][T_NV];           // both rings in metres (c_alpha + position)
  union Coord { i64 f; double m; };
  Coord qx[2][T_CAPV], qy[2][T_CAPV];        // the overlap regions of clip 1 / clip 2: fixed point out of the clip, metres afterwards
  // InterX crossings (sorted, unique) -- and, while no crossings are held, the pair's team-uniform values that must
  // survive a clip: a value every lane holds alike costs a vector register in each of them, and the clip needs the
  // registers (the kernel is built for four waves per SIMD: 128 registers).  keep[0..3]: the first floe's box until the
  // merge flags (:54-60), before InterX; keep[0..8]: area, centroid, direction, dl, contact point, stiffness across the
  // second clip, after the crossings' last use.
  union { struct { double px[T_CAPP], py[T_CAPP]; }; double keep[2 * T_CAPP]; };
  double body[2][BODY_WORDS];                // what floe_interactions.m reads of floe1 / floe2 (contact.hpp Body)
  union {
    double rec[2][16];                       // the two entries' records as loaded (step_core.hpp narrow_team), until the first clip
    ClipScratch clip;
    double terms[4][T_CAPV];
    struct { double x[T_CAPR], y[T_CAPR]; int key[T_CAPR]; unsigned char dup[T_CAPR]; } raw;
  } u;
};

template <int G>
SZ_TEAM int min_all_int(const Team<G> &T, int v, int site) {
#pragma unroll
  for (int d = 1; d < G; d <<= 1) { const int u = T.shfl(v, T.lane ^ d, site + d); v = (u < v) ? u : v; }
  return v;
}

// one overlap region in metres, as the model reads it
struct RegionM {
  const TeamBlock::Coord *x, *y;
  int n;
  SZ_HD double X(int k) const { return x[k].m; }
  SZ_HD double Y(int k) const { return y[k].m; }
};

// the clip of the two staged rings (ring 0 moved by (sx, sy) when `shifted`) into region slot `slot`, converted to
// metres; *nr = its vertex count (0: no overlap)
template <int G>
SZ_TEAM int clip_staged(const Team<G> &T, TeamBlock *B, int n1, int n2, bool shifted, double sx, double sy, int slot, int *nr) {
  ClipScratch *S = &B->u.clip;
  for (int k = T.lane; k < n1; k += G) {
    double X = B->mx[0][k], Y = B->my[0][k];
    if (shifted) { X = X + sx; Y = Y + sy; }
    S->fx[0][k] = szx::to_fixed_in_range(X); S->fy[0][k] = szx::to_fixed_in_range(Y);
  }
  for (int k = T.lane; k < n2; k += G) { S->fx[1][k] = szx::to_fixed_in_range(B->mx[1][k]); S->fy[1][k] = szx::to_fixed_in_range(B->my[1][k]); }
  T.sync(TEAM_SITE);
  const int st = team_clip(T, S, n1, n2, &B->qx[slot][0].f, &B->qy[slot][0].f, nr);
  if (st != szv::ST_OK) return st;
  // (team_clip ends with a sync: the ring is visible)  polyclip.m:67 once per vertex
  i64 vx[(T_CAPV + G - 1) / G], vy[(T_CAPV + G - 1) / G];
#pragma unroll
  for (int q = 0; q < (T_CAPV + G - 1) / G; ++q) { const int k = T.lane + q * G; vx[q] = vy[q] = 0; if (k < *nr) { vx[q] = B->qx[slot][k].f; vy[q] = B->qy[slot][k].f; } }
#pragma unroll
  for (int q = 0; q < (T_CAPV + G - 1) / G; ++q) { const int k = T.lane + q * G; if (k < *nr) { B->qx[slot][k].m = szc::from_fixed(vx[q]); B->qy[slot][k].m = szc::from_fixed(vy[q]); } }
  T.sync(TEAM_SITE);
  return szv::ST_OK;
}

// sum of terms[0 .. n) in index order
SZ_HD double sum_in_order(const double *t, int n) { double s = 0; for (int i = 0; i < n; ++i) s += t[i]; return s; }

// area(polyshape) of a region: |shoelace| relative to the first vertex (contact.hpp region_area)
template <int G>
SZ_TEAM double region_area_team(const Team<G> &T, TeamBlock *B, const RegionM &R) {
  if (R.n < 3) return 0;
  const double x0 = R.X(0), y0 = R.Y(0);
  for (int i = T.lane; i < R.n; i += G) {
    const int k = (i + 1 == R.n) ? 0 : i + 1;
    const double ax = R.X(i) - x0, ay = R.Y(i) - y0, bx = R.X(k) - x0, by = R.Y(k) - y0;
    B->u.terms[0][i] = ax * by - bx * ay;
  }
  T.sync(TEAM_SITE);
  const double s = sum_in_order(B->u.terms[0], R.n);
  T.sync(TEAM_SITE);
  return absd(0.5 * s);
}
template <int G>
SZ_TEAM void region_centroid_team(const Team<G> &T, TeamBlock *B, const RegionM &R, double &cx, double &cy) {
  const double x0 = R.X(0), y0 = R.Y(0);
  for (int i = T.lane; i < R.n; i += G) {
    const int k = (i + 1 == R.n) ? 0 : i + 1;
    const double ax = R.X(i) - x0, ay = R.Y(i) - y0, bx = R.X(k) - x0, by = R.Y(k) - y0;
    const double w = ax * by - bx * ay;
    B->u.terms[0][i] = w; B->u.terms[1][i] = (ax + bx) * w; B->u.terms[2][i] = (ay + by) * w;
  }
  T.sync(TEAM_SITE);
  const double s = sum_in_order(B->u.terms[0], R.n), sx = sum_in_order(B->u.terms[1], R.n), sy = sum_in_order(B->u.terms[2], R.n);
  T.sync(TEAM_SITE);
  cx = x0 + sx / (3.0 * s);
  cy = y0 + sy / (3.0 * s);
}
// polyarea.m (contact.hpp region_polyarea)
template <int G>
SZ_TEAM double region_polyarea_team(const Team<G> &T, TeamBlock *B, const RegionM &R) {
  for (int i = T.lane; i < R.n; i += G) {
    const int k = (i + 1 == R.n) ? 0 : i + 1;
    B->u.terms[0][i] = (R.X(k) - R.X(i)) * (R.Y(k) + R.Y(i));
  }
  T.sync(TEAM_SITE);
  const double s = sum_in_order(B->u.terms[0], R.n);
  T.sync(TEAM_SITE);
  return absd(s / 2);
}

// what edge m of the region contributes to inpolygon.m:150-224 for the point (x, y): the quadrant step (a small
// integer) and the on-the-edge flag
SZ_HD void inpoly_edge(double x, double y, double xa, double ya, double xb, double yb, int &dq_out, bool &on) {
  const double avx = absd(0.5 * (xa + xb)), avy = absd(0.5 * (ya + yb));
  double sf = maxd(avx, avy);
  sf = maxd(sf, avx * avy);
  const double ax = xa - x, ay = ya - y, bx = xb - x, by = yb - y;
  int qa, qb;
  { const bool px = ax > 0, py = ay > 0; qa = (!px && py) ? 1 : (!px && !py) ? 2 : (px && !py) ? 3 : 0; }
  { const bool px = bx > 0, py = by > 0; qb = (!px && py) ? 1 : (!px && !py) ? 2 : (px && !py) ? 3 : 0; }
  const double cp = ax * by - bx * ay;
  int scp = (cp > 0) ? 1 : (cp < 0 ? -1 : 0);
  if (absd(cp) < sf * SZC_EPS * 3) scp = 0;
  const double dp = ax * bx + ay * by;
  int dq = qb - qa;
  if (dq == 3 || dq == -3) dq = -dq / 3;
  if (dq == 2 || dq == -2) dq = 2 * scp;
  dq_out = dq;
  on = (scp == 0 && dp <= 0);
}
// inpolygon of one point against a region, every lane for itself (contact.hpp in_region on metres)
SZ_HD bool in_region_m(double x, double y, const RegionM &R) {
  double mnx = R.X(0), mxx = mnx, mny = R.Y(0), mxy = mny;
  for (int k = 1; k < R.n; ++k) { const double vx = R.X(k), vy = R.Y(k); mnx = mind(mnx, vx); mxx = maxd(mxx, vx); mny = mind(mny, vy); mxy = maxd(mxy, vy); }
  if (!(x >= mnx && x <= mxx && y >= mny && y <= mxy)) return false;
  int sumdq = 0; bool on = false;
  for (int m = 0; m < R.n; ++m) {
    const int m1 = (m + 1 == R.n) ? 0 : m + 1;
    int dq; bool o;
    inpoly_edge(x, y, R.X(m), R.Y(m), R.X(m1), R.Y(m1), dq, o);
    sumdq += dq; on = on || o;
  }
  return (sumdq != 0) || on;
}
// contact.hpp clearly_inside for one point known to the whole team: an edge per lane
template <int G>
SZ_TEAM bool clearly_inside_team(const Team<G> &T, double x, double y, const RegionM &R, double margin) {
  if (R.n < 3) return false;
  double mnx = SZC_INF, mxx = -SZC_INF, mny = SZC_INF, mxy = -SZC_INF;
  int sumdq = 0; bool on = false, clear = true;
  for (int m = T.lane; m < R.n; m += G) {
    const int m1 = (m + 1 == R.n) ? 0 : m + 1;
    const double xa = R.X(m), ya = R.Y(m), xb = R.X(m1), yb = R.Y(m1);
    mnx = mind(mnx, xa); mxx = maxd(mxx, xa); mny = mind(mny, ya); mxy = maxd(mxy, ya);
    int dq; bool o;
    inpoly_edge(x, y, xa, ya, xb, yb, dq, o);
    sumdq += dq; on = on || o;
    const double ex = xb - xa, ey = yb - ya, px = x - xa, py = y - ya;
    const double ee = ex * ex + ey * ey;
    double t = (ee > 0) ? (px * ex + py * ey) / ee : 0;
    if (t < 0) t = 0;
    if (t > 1) t = 1;
    const double dx = px - t * ex, dy = py - t * ey;
    if (!(dx * dx + dy * dy > margin * margin)) clear = false;
  }
  mnx = T.min_all(mnx, TEAM_SITE); mxx = T.max_all(mxx, TEAM_SITE); mny = T.min_all(mny, TEAM_SITE); mxy = T.max_all(mxy, TEAM_SITE);
  sumdq = T.sum_all(sumdq, TEAM_SITE);
  const unsigned onm = T.ballot(on, TEAM_SITE), notclear = T.ballot(!clear, TEAM_SITE);
  if (!(x >= mnx && x <= mxx && y >= mny && y <= mxy)) return false;
  if (!((sumdq != 0) || onm != 0)) return false;
  return notclear == 0;
}

// InterX.m:54-81 on the two staged rings (closed by one more vertex when extra is set): the sorted set of crossings in
// B->px / py.  -> their number, or -1 when the lists do not hold them / a coordinate is not finite (the caller defers)
template <int G>
SZ_TEAM int crossings_team(const Team<G> &T, TeamBlock *B, int n1, int extra1, int n2, int extra2) {
  enum { JT = (T_NV + 1 + G - 1) / G };
  const int m1 = n1 + extra1 - 1, m2 = n2 + extra2 - 1;
  if (m1 <= 0 || m2 <= 0) return 0;
  // lane j: segment j of ring 2 (and j + G ...)
  double x2a[JT], y2a[JT], x2b[JT], y2b[JT];
  unsigned hit[JT];
#pragma unroll
  for (int q = 0; q < JT; ++q) {
    const int j = T.lane + q * G;
    hit[q] = 0; x2a[q] = y2a[q] = x2b[q] = y2b[q] = 0;
    if (j < m2) {
      const int ja = (j >= n2) ? 0 : j, jb = (j + 1 >= n2) ? 0 : j + 1;
      x2a[q] = B->mx[1][ja]; y2a[q] = B->my[1][ja]; x2b[q] = B->mx[1][jb]; y2b[q] = B->my[1][jb];
    }
  }
  for (int i = 0; i < m1; ++i) {
    const int ia = (i >= n1) ? 0 : i, ib = (i + 1 >= n1) ? 0 : i + 1;
    const double x1a = B->mx[0][ia], y1a = B->my[0][ia], x1b = B->mx[0][ib], y1b = B->my[0][ib];
    const double dx1 = x1b - x1a, dy1 = y1b - y1a;
    const double S1 = dx1 * y1a - dy1 * x1a;
#pragma unroll
    for (int q = 0; q < JT; ++q) {
      const double a = dx1 * y2a[q] - dy1 * x2a[q], b = dx1 * y2b[q] - dy1 * x2b[q];
      if (T.lane + q * G < m2 && ((a - S1) * (b - S1)) <= 0) hit[q] |= 1u << i;
    }
  }
  // the segment pairs that passed: second sign test, the point; appended in rounds of one per lane
  int nraw = 0;
  bool overflow = false, notfinite = false;
  for (;;) {
    bool have = false;
#pragma unroll
    for (int q = 0; q < JT; ++q) have = have || hit[q] != 0;
    if (!T.any(have, TEAM_SITE)) break;
    bool valid = false;
    double X = 0, Y = 0;
    int key = 0;
    if (have) {
      int q = 0;
#pragma unroll
      for (int qq = JT - 1; qq >= 0; --qq) if (hit[qq] != 0) q = qq;
      unsigned hq = 0; double ax2 = 0, ay2 = 0, bx2 = 0, by2 = 0;
#pragma unroll
      for (int qq = 0; qq < JT; ++qq) if (qq == q) { hq = hit[qq]; ax2 = x2a[qq]; ay2 = y2a[qq]; bx2 = x2b[qq]; by2 = y2b[qq]; }
      const int i = __builtin_ctz(hq);
#pragma unroll
      for (int qq = 0; qq < JT; ++qq) if (qq == q) hit[qq] &= hit[qq] - 1;
      const int ia = (i >= n1) ? 0 : i, ib = (i + 1 >= n1) ? 0 : i + 1;
      const double x1a = B->mx[0][ia], y1a = B->my[0][ia], x1b = B->mx[0][ib], y1b = B->my[0][ib];
      const double dx1 = x1b - x1a, dy1 = y1b - y1a;
      const double S1 = dx1 * y1a - dy1 * x1a;
      const double dx2 = bx2 - ax2, dy2 = by2 - ay2;
      const double S2 = dx2 * ay2 - dy2 * ax2;
      const double c = y1a * dx2 - x1a * dy2, d = y1b * dx2 - x1b * dy2;
      if (((c - S2) * (d - S2)) <= 0) {
        const double L = dy2 * dx1 - dy1 * dx2;
        if (L != 0) {
          X = (dx2 * S1 - dx1 * S2) / L;
          Y = (dy2 * S1 - dy1 * S2) / L;
          valid = true;
          key = i * 64 + (T.lane + q * G);
          if (!(absd(X) < 1.7976931348623157e308) || !(absd(Y) < 1.7976931348623157e308)) notfinite = true;
        }
      }
    }
    const unsigned vm = T.ballot(valid, TEAM_SITE);
    if (valid) {
      const int at = nraw + __builtin_popcount(vm & ((1u << T.lane) - 1u));
      if (at < (int)T_CAPR) { B->u.raw.x[at] = X; B->u.raw.y[at] = Y; B->u.raw.key[at] = key; } else overflow = true;
    }
    nraw += __builtin_popcount(vm);
  }
  if (T.any(overflow || notfinite, TEAM_SITE)) return -1;
  if (nraw == 0) return 0;
  T.sync(TEAM_SITE);
  // unique(...,'rows'): of equal points the one found first (segment-of-ring-1 major) stays; ascending x, then y
  for (int q = T.lane; q < nraw; q += G) {
    const double X = B->u.raw.x[q], Y = B->u.raw.y[q];
    const int key = B->u.raw.key[q];
    bool dup = false;
    for (int m = 0; m < nraw; ++m) if (B->u.raw.key[m] < key && B->u.raw.x[m] == X && B->u.raw.y[m] == Y) dup = true;
    B->u.raw.dup[q] = dup ? 1 : 0;
  }
  T.sync(TEAM_SITE);
  int np = 0;
  for (int m = 0; m < nraw; ++m) np += B->u.raw.dup[m] ? 0 : 1;
  if (np > (int)T_CAPP) return -1;
  for (int q = T.lane; q < nraw; q += G) {
    if (B->u.raw.dup[q]) continue;
    const double X = B->u.raw.x[q], Y = B->u.raw.y[q];
    int rank = 0;
    for (int m = 0; m < nraw; ++m)
      if (!B->u.raw.dup[m] && (B->u.raw.x[m] < X || (B->u.raw.x[m] == X && B->u.raw.y[m] < Y))) ++rank;
    B->px[rank] = X; B->py[rank] = Y;
  }
  T.sync(TEAM_SITE);
  return np;
}

// |p_poly_dist| < 1e-8 of one point against staged ring 0 closed as the model closes it (contact.hpp dist_to_ring on RingL)
SZ_HD bool closer_than_1e8_staged(double xp, double yp, const TeamBlock *B, int n1, int extra1) {
  szc::RingL c = {B->mx[0], B->my[0], n1, extra1};
  return szc::dist_to_ring(xp, yp, c) < 1e-8;
}

// A certificate that the model returns no contact row, from the projections of the two staged rings onto the normal n
// and the direction t of an edge of either ring (an edge per lane, every lane over all vertices):
//   * the projections onto n lie more than 1e-6 m apart: the convex hulls -- and the rings -- are more than 1e-6 m (4295
//     units of the fixed-point grid) apart, the clip of the rounded rings is empty and InterX finds no crossing (its
//     sign tests differ from exact arithmetic by ~1e-10 m at these coordinates): floe_interactions.m returns its
//     no-contact form ([0 0], [0 0], 0; :43-74) -- the argument of contact.hpp boxes_apart with an edge normal in place
//     of a coordinate axis;
//   * or the overlap of the two hulls fits a rectangle d x w (d: overlap of the projections onto n, w: onto t; both
//     grown by the rounding of the projections and of the clip) of less area than `amin`, a lower bound of the model's
//     threshold (:78-83: regions below min(N1, N2) * 100 / 1.75 m^2 are skipped): whatever regions the clip returns
//     lie inside both rings, hence inside that rectangle -- each is skipped, there are no rows; and with `amin` below
//     0.55 of either floe's area (the caller checks) the merge flags of :54-60 stay 0 as well.
// Either way every output of the pair is the no-contact result, without running the clip.  A `false` only means
// "not certified": the caller runs the model.  Rounding: a projection is two products and a sum of coordinates below
// `cmax`, off by less than 4e-16 (|nx| + |ny|) cmax; |nx| + |ny| bounds the length of n from above.  NaN compares false.
template <int G>
SZ_TEAM bool certainly_no_rows(const Team<G> &T, const TeamBlock *B, int n1, int n2, double cmax, double amin) {
  bool found = false;
  for (int k = T.lane; k < n1 + n2; k += G) {
    const int r = (k >= n1) ? 1 : 0, l = k - (r ? n1 : 0), n = r ? n2 : n1, l1 = (l + 1 == n) ? 0 : l + 1;
    const double nx = B->my[r][l1] - B->my[r][l], ny = B->mx[r][l] - B->mx[r][l1];
    const double len = absd(nx) + absd(ny), len2 = nx * nx + ny * ny;
    if (!(len2 > 0)) continue;                                 // (a repeated vertex is no axis)
    double nlo0 = SZC_INF, nhi0 = -SZC_INF, nlo1 = SZC_INF, nhi1 = -SZC_INF, tlo0 = SZC_INF, thi0 = -SZC_INF, tlo1 = SZC_INF, thi1 = -SZC_INF;
    for (int v = 0; v < n1; ++v) {
      const double x = B->mx[0][v], y = B->my[0][v], pn = nx * x + ny * y, pt = nx * y - ny * x;
      nlo0 = mind(nlo0, pn); nhi0 = maxd(nhi0, pn); tlo0 = mind(tlo0, pt); thi0 = maxd(thi0, pt);
    }
    for (int v = 0; v < n2; ++v) {
      const double x = B->mx[1][v], y = B->my[1][v], pn = nx * x + ny * y, pt = nx * y - ny * x;
      nlo1 = mind(nlo1, pn); nhi1 = maxd(nhi1, pn); tlo1 = mind(tlo1, pt); thi1 = maxd(thi1, pt);
    }
    const double slack = len * (1e-6 + 1e-15 * cmax);          // 1e-6 m and the rounding, in units of the projections
    if (nlo1 - nhi0 > slack || nlo0 - nhi1 > slack) { found = true; continue; }
    const double d = mind(nhi0, nhi1) - maxd(nlo0, nlo1) + 2 * slack, w = mind(thi0, thi1) - maxd(tlo0, tlo1) + 2 * slack;
    if (d > 0 && w > 0 && (d * w) / len2 * (1 + 1e-9) < amin) found = true;   // (projections are lengths times |n|: the area is d w / |n|^2)
  }
  return T.any(found, TEAM_SITE);
}

// what narrow_team has put into the block: both rings in mx / my (sg.n1, sg.n2 vertices: the stored length, one less
// for a ring stored closed that is one vertex too long for the block -- the closure rule of the model, :62-67, puts the
// copy back; -1: too long), the bodies in `body`
struct Staged { int n1, n2, stored1, stored2; bool in_range; bool cold; };   // cold: not in contact last step (a hint: try the cheap no-contact certificate first)
// the domain box as the model uses it (step_core.hpp boundary_summary)
struct BoundaryBox { double xmax, xmin, ymax, ymin, area; bool rect; };

// The pair.  Same contract as contact.hpp pair_contact_fast, on rings already staged (b1, b2: the bounding boxes of
// everything the rings hold, formed once per floe and step -- step_core.hpp cell_count); row: the contact row
// [Fx, Fy, Px, Py, overlap] when out.nrows == 1.  Every lane returns the same status, `out` and row.
template <int G>
SZ_TEAM int pair_contact_team(const Team<G> &T, TeamBlock *B, const Staged &sg, const szc::Box &b1_in, const szc::Box &b2, bool boundary,
                              const BoundaryBox &bb, bool periodic, double Modulus, double dt, double mu, double nu, double *row, szc::PairOut &out) {
  SZTM_PHASE(1);
  const szc::Box &b1 = b1_in;          // (valid up to the first clip; after it the box is read back from B->keep)
  out.nrows = 0; out.sentinel = 0; out.has_sentinel = 0;
  if (boundary) {   // only floes at the walls need the clip against the box (general sweep: the box has horizontal edges)
    // (contact.hpp well_inside_box: the floe's box more than 1e-6 m inside an axis-aligned rectangular domain box;
    // bb.rect also asks the ring to be closed by a copy of its first corner, so that the 4- and 5-corner extents agree)
    const double gap = 1e-6;
    if (sg.stored1 > 0 && bb.rect && b1.x0 > bb.xmin + gap && b1.x1 < bb.xmax - gap && b1.y0 > bb.ymin + gap && b1.y1 < bb.ymax - gap) { szc::no_contact(out); return szv::ST_OK; }
    return szc::ST_DEFER;
  }
  const int n1 = sg.n1, n2 = sg.n2;
  if (n1 < 1 || n2 < 1) {   // too long for the block: the lane-per-pair kernels' business -- unless the boxes are apart
    if (sg.stored1 > 0 && sg.stored2 > 0 && szc::boxes_apart(b1, b2)) { szc::no_contact(out); return szv::ST_OK; }
    return (sg.stored1 <= (int)szx::FastBlockBig::NV && sg.stored2 <= (int)szx::FastBlockBig::NV && sg.stored1 >= 1 && sg.stored2 >= 1) ? (int)szc::ST_DEFER_BIG : (int)szc::ST_DEFER;
  }
  if (!sg.in_range) return szc::ST_DEFER;
  if (szc::boxes_apart(b1, b2)) { szc::no_contact(out); return szv::ST_OK; }
  if (n1 + n2 - 2 > G) return ST_DEFER_WIDE;   // (even with a closing copy in each ring: more scanlines than lanes)
  if (sg.cold) {
    const double cmax = maxd(maxd(maxd(absd(b1.x0), absd(b1.x1)), maxd(absd(b1.y0), absd(b1.y1))), maxd(maxd(absd(b2.x0), absd(b2.x1)), maxd(absd(b2.y0), absd(b2.y1))));
    // (N1, N2 of :78-79 are at least the staged lengths less one: the rings may be closed by one more vertex, :62-67)
    double amin = (double)((n1 < n2 ? n1 : n2) - 1) * 100 / 1.75 * (1 - 1e-9);
    if (!(amin < 0.55 * mind(B->body[0][BODY_AREA], B->body[1][BODY_AREA]))) amin = 0;
    if (certainly_no_rows(T, B, n1, n2, cmax, amin)) { szc::no_contact(out); return szv::ST_OK; }
  }
  if (T.lane == 0) { B->keep[0] = b1.x0; B->keep[1] = b1.x1; B->keep[2] = b1.y0; B->keep[3] = b1.y1; }
  T.sync(TEAM_SITE);
  SZTM_PHASE(2);
  // ---- the first clip (:29), the model up to the force direction, the second clip (:140-155) ----
  int nr = 0, nt = 0, tslot = 0;
  double sx = 0, sy = 0;                 // the shift of the second clip (the force direction; dead once its rings are staged)
  // (SZ_TEAM_ONE_CLIP: one instance of the clip in the code, passed through twice -- smaller code, but what the clip
  // holds then stays live around the loop: 24 registers more than the kernel's 128)
#if defined(SZ_TEAM_ONE_CLIP)
#pragma nounroll
#else
#pragma unroll
#endif
  for (int pass = 0; pass < 2; ++pass) {
  int nout = 0;
  Team<G> Tc = T;
  opaque(Tc.lane);
  const int st = clip_staged(Tc, B, n1, n2, pass == 1, sx, sy, pass, &nout);
  if (st == ST_WIDE) return ST_DEFER_WIDE;
  if (st == szx::ST_BAIL) return szc::ST_DEFER;
  if (st != szv::ST_OK) return st;
  if (pass == 1) { nt = nout; tslot = 1; break; }
  nr = nout;
  const int nreg = nr > 0 ? 1 : 0;
  const RegionM R = {B->qx[0], B->qy[0], nr};
  SZTM_PHASE(3);
  // ---- pair_finish ----
  double Ar = 0, cx = 0, cy = 0, fdx = 0, fdy = 0, dl = 0, pcx = 0, pcy = 0, Force_factor = 0;
  const double area1 = B->body[0][BODY_AREA], area2 = B->body[1][BODY_AREA];
  double h1 = B->body[0][BODY_H], h2 = B->body[1][BODY_H];                                // :10-21
  double r1 = ::sqrt(area1), r2 = ::sqrt(area2);
  Force_factor = Modulus * (h1 * h2) / (h1 * r2 + h2 * r1);
  double overlap0 = 0;
  if (r1 > 1e5 || r2 > 1e5) {
    r1 = mind(r1, r2);
    h1 = mind(h1, h2);
    Force_factor = Modulus * h1 / r1;
  }
  Ar = nreg ? region_area_team(T, B, R) : 0.0;
  SZTM_PHASE(4);
  {                                                                                       // :54-60
    const double bxmax = bb.xmax, bxmin = bb.xmin, bymax = bb.ymax, bymin = bb.ymin, barea = bb.area;
    // (the box of the staged ring: the closing copy left out is a copy of vertex 0; kept in the block across the clip)
    const double b1x0 = B->keep[0], b1x1 = B->keep[1], b1y0 = B->keep[2], b1y1 = B->keep[3];
    const bool inside = (b1x1 < bxmax && b1x0 > bxmin && b1y1 < bymax && b1y0 > bymin);
    if (inside || area2 < 0.95 * barea || periodic) {
      double sumAr = 0;
      if (nreg) sumAr += Ar;
      if (sumAr / area1 > 0.55) overlap0 = SZC_INF;
      else if (sumAr / area2 > 0.55) overlap0 = -SZC_INF;
    }
  }
  // :62-67 close the rings
  const int extra1 = (hyp(B->mx[0][0] - B->mx[0][n1 - 1], B->my[0][0] - B->my[0][n1 - 1]) > 1) ? 1 : 0;
  const int extra2 = (hyp(B->mx[1][0] - B->mx[1][n2 - 1], B->my[1][0] - B->my[1][n2 - 1]) > 1) ? 1 : 0;
  const int np = crossings_team(T, B, n1, extra1, n2, extra2);                             // :70
  SZTM_PHASE(5);
  if (np < 0) return szc::ST_DEFER;
  if (np < 2 || szc::is_inf(overlap0) || nreg == 0) {                                      // :71-74
    out.has_sentinel = 1;
    out.sentinel = overlap0;
    return szv::ST_OK;
  }
  const int N1 = n1 + extra1 - 1, N2 = n2 + extra2 - 1;                                    // :78-79
  const double Amin = (double)(N1 < N2 ? N1 : N2) * 100 / 1.75;
  if (Ar < Amin) return szv::ST_OK;                                                        // :83
  region_centroid_team(T, B, R, cx, cy);                                                   // :96-97
  // :98-100 nearest region vertex of each crossing; keep those closer than 1 m (every lane the same loop: the
  // region and the crossings are a handful of values in LDS)
  int m = 0;
  double p0x = 0, p0y = 0, p1x = 0, p1y = 0;
  for (int q = 0; q < np; ++q) {
    double best = SZC_INF; int bi = 0;
    const double qx = B->px[q], qy = B->py[q];
    for (int v = 0; v < nr; ++v) {
      const double ddx = R.X(v) - qx, ddy = R.Y(v) - qy;
      const double d2 = ddx * ddx + ddy * ddy;
      if (d2 < best) { best = d2; bi = v; }
    }
    if (::sqrt(best) < 1) {
      if (m == 0) { p0x = R.X(bi); p0y = R.Y(bi); }
      else if (m == 1) { p1x = R.X(bi); p1y = R.Y(bi); }
      ++m;
    }
  }
  fdx = 0; fdy = 0; dl = 0; pcx = cx; pcy = cy;
  if (Ar == 0) { fdx = fdy = 0; pcx = pcy = 0; dl = 0; }                                   // :103-106
  else if (m == 2) {                                                                       // :107-112
    const double xgh = p1x - p0x, ygh = p1y - p0y;
    const double b = ::sqrt(xgh * xgh + ygh * ygh);
    fdx = -ygh / b; fdy = xgh / b;
    dl = b;
  } else if (m == 0) { fdx = fdy = 0; dl = 0; }                                            // :113-116
  else {                                                                                   // :117-137, an edge of the region per lane
    for (int e = T.lane; e < nr; e += G) {
      const int e1 = (e + 1 == nr) ? 0 : e + 1;
      const double xa = R.X(e), ya = R.Y(e), xb = R.X(e1), yb = R.Y(e1);
      const double xgh = xb - xa, ygh = yb - ya;
      const double xm = (xb + xa) / 2, ym = (yb + ya) / 2;
      const double b = ::sqrt(xgh * xgh + ygh * ygh);
      double nx = -ygh / b, ny = xgh / b;
      const double xt = xm + nx / 100, yt = ym + ny / 100;
      if (!in_region_m(xt, yt, R)) { nx = -nx; ny = -ny; }
      const bool near = closer_than_1e8_staged(xm, ym, B, n1, extra1);                     // p_poly_dist(...) < 1e-8, :127
      B->u.terms[0][e] = near ? 1.0 : 0.0;
      B->u.terms[1][e] = (-Force_factor * b) * nx;
      B->u.terms[2][e] = (-Force_factor * b) * ny;
      B->u.terms[3][e] = b;
    }
    T.sync(TEAM_SITE);
    int non = 0;
    double sfx = 0, sfy = 0, sb = 0;
    for (int e = 0; e < nr; ++e)
      if (B->u.terms[0][e] != 0) { ++non; sfx += B->u.terms[1][e]; sfy += B->u.terms[2][e]; sb += B->u.terms[3][e]; }
    T.sync(TEAM_SITE);
    if (non < nr && non > 0) {
      const double nrm = ::sqrt(sfx * sfx + sfy * sfy);
      fdx = sfx / nrm; fdy = sfy / nrm;
      dl = sb / non;
    } else { fdx = fdy = 0; dl = 0; }
  }
  SZTM_PHASE(6);
  // :140-155 shift ring 1 by the unit direction and clip again (not shifted: the first clip again, same rings, same result)
  const bool again = !(dl < 0.1);
  if (!again) { fdx = 0; fdy = 0; }
  // (the crossings have had their last use: their room keeps what the rest of the model needs across the second clip)
  T.sync(TEAM_SITE);
  if (T.lane == 0) {
    B->keep[0] = Ar; B->keep[1] = cx; B->keep[2] = cy; B->keep[3] = fdx; B->keep[4] = fdy; B->keep[5] = dl;
    B->keep[6] = pcx; B->keep[7] = pcy; B->keep[8] = Force_factor;
  }
  sx = fdx; sy = fdy;
  if (!again) { nt = nr; tslot = 0; break; }
  }
  T.sync(TEAM_SITE);
  const double Ar = B->keep[0], cx = B->keep[1], cy = B->keep[2], dl = B->keep[5], pcx = B->keep[6], pcy = B->keep[7], Force_factor = B->keep[8];
  double fdx = B->keep[3], fdy = B->keep[4];
  const double Gm = Modulus / (2 * (1 + nu));
  const RegionM R = {B->qx[0], B->qy[0], nr};
  SZTM_PHASE(7);
  if (nt > 0) {                                                                            // :156-165 (one region at most)
    const RegionM Tn = {B->qx[tslot], B->qy[tslot], nt};
    const double Anew = region_polyarea_team(T, B, Tn);
    if (Anew / Ar - 1 > 0) {
      // a point with a 1e-6 m disc inside both regions certifies that the third clip (:158-159) is not empty
      // (the centroid of the original overlap first -- the larger, shifted one normally contains it -- then the
      // centroid of the shifted overlap; one instance of the test in the code)
      bool overlaps = false;
#pragma nounroll
      for (int attempt = 0; attempt < 2 && !overlaps; ++attempt) {
        double x = cx, y = cy;
        if (attempt == 1) { if (nt < 3) break; region_centroid_team(T, B, Tn, x, y); }
        bool ok = true;
#pragma nounroll
        for (int w = 0; w < 2 && ok; ++w) ok = clearly_inside_team(T, x, y, w ? Tn : R, 1e-6);
        overlaps = ok;
      }
      if (!overlaps) return szc::ST_DEFER;                                                 // the third clip decides: general sweep
      fdx = -fdx; fdy = -fdy;
    }
  }
  SZTM_PHASE(8);
  const double Fx = fdx * Ar * Force_factor, Fy = fdy * Ar * Force_factor;                 // :167
  // :170-183
  const double *f1 = B->body[0], *f2 = B->body[1];
  const double v1x = f1[BODY_UI] + f1[BODY_KSI] * (pcx - f1[BODY_X]), v1y = f1[BODY_VI] + f1[BODY_KSI] * (pcy - f1[BODY_Y]);
  const double v2x = f2[BODY_UI] + f2[BODY_KSI] * (pcx - f2[BODY_X]), v2y = f2[BODY_VI] + f2[BODY_KSI] * (pcy - f2[BODY_Y]);
  const double vtx = v1x - v2x, vty = v1y - v2y;
  double dtx, dty;
  if (maxd(absd(vtx), absd(vty)) == 0) { dtx = 0; dty = 0; }
  else { const double nv = ::sqrt(vtx * vtx + vty * vty); dtx = vtx / nv; dty = vty / nv; }
  const double vn = ::sqrt(vtx * vtx + vty * vty);
  const double coef = -(dtx * vtx + dty * vty) * dl * Gm * vn;
  double ftx = coef * dtx * dt, fty = coef * dty * dt;
  const double nF = ::sqrt(Fx * Fx + Fy * Fy);
  if (::sqrt(ftx * ftx + fty * fty) > mu * nF) { ftx = -mu * nF * dtx; fty = -mu * nF * dty; }
  row[0] = Fx + ftx; row[1] = Fy + fty; row[2] = pcx; row[3] = pcy; row[4] = Ar;
  SZTM_PHASE(9);
  out.nrows = 1;
  return szv::ST_OK;
}

} // namespace sztm
