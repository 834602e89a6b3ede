// corners.hpp -- the per-vertex part of corner grinding: corners.m:55-92 up to the `grind` decision.
//
// For a floe that touched something this step the reference computes, per vertex of its outline,
//   angles  the interior angle (polygon_operations/polyangles.m:1-159, clockwise rings), which with one uniform draw
//           per vertex decides `break1 = rand > angles / Anorm` (corners.m:69-71), and
//   da      1 when the vertex lies inside a partner's outline / outside the domain boundary (inpolygon, :76-87) or is
//           the vertex nearest to one of the floe's contact points (dsearchn, :73,88);
// grind = break1 & da (:91).  Everything but the draw is a function of the device-resident contact rows and rings, so
// the library returns `angles` and `da` and the caller applies its own random stream (exact parity with a given
// stream).  frac_corner.m -- polyshape subtraction, new floes, fresh Monte-Carlo points -- reshapes the floe array and
// stays on the MATLAB side.
//
// polyshape(...).Vertices is taken to be the stored outline without its closing copy, in the stored order (floe
// outlines are written by polyshape in the first place: clockwise, no repeated vertices).
#pragma once
#include "step_core.hpp"
#include "detmath.hpp"

namespace szk {

using szc::absd; using szc::maxd; using szc::mind;

// polyangles.m:136-159 polyedge: coefficients of the line through p1, p2 (w3 in w[2])
SZ_HD void polyedge(double x1, double y1, double x2, double y2, double *w) {
  const double eps = 2.220446049250313e-16;
  if (x1 == x2) { w[1] = 0; w[0] = -1 / x1; w[2] = 1; }
  else if (y1 == y2) { w[0] = 0; w[1] = -1 / y1; w[2] = 1; }
  else if (x1 == y1 && x2 == y2) { w[0] = 1; w[1] = 1; w[2] = 0; }
  else {
    w[0] = (y1 - y2) / (x1 * (y2 - y1) - y1 * (x2 - x1) + eps);
    w[1] = -w[0] * (x2 - x1) / (y2 - y1);
    w[2] = 1;
  }
}
SZ_HD double sign_of(double v) { return v > 0 ? 1.0 : (v < 0 ? -1.0 : v); }   // sign(0) = 0, sign(NaN) = NaN

// polyangles.m:57-133 convex_angle_test for vertex j (0-based) of the open ring (x, y)[n]: +1 convex, -1 concave
SZ_HD int convex_sign(int j, const double *x, const double *y, int n) {
  if (j == 0) return 1;                                    // :91 "the first vertex is known to be convex"
  const int jn = (j + 1 == n) ? 0 : j + 1;
  const double fx = x[j - 1], fy = y[j - 1], sx = x[j], sy = y[j], tx = x[jn], ty = y[jn];
  double w[3];
  polyedge(fx, fy, sx, sy, w);
  const double L = 1e25;
  const double dx = sx - fx, dy = sy - fy;
  double ex, ey;                                           // the extreme point that should lie on the + side (:114-126)
  if (dx <= 0 && dy >= 0) { ex = L; ey = L; }              // bottom_right
  else if (dx <= 0 && dy <= 0) { ex = -L; ey = L; }        // top_right
  else if (dx >= 0 && dy <= 0) { ex = -L; ey = -L; }       // top_left
  else { ex = L; ey = -L; }                                // bottom_left
  const double vp = (w[0] * ex + w[1] * ey) + w[2] * 1.0;  // dot(w, [ex ey 1])
  const double s = sign_of(vp);
  w[0] = s * w[0]; w[1] = s * w[1]; w[2] = s * w[2];
  return ((w[0] * tx + w[1] * ty + w[2]) < 0) ? -1 : 1;    // :129-132
}

// polyangles.m:26-54: interior angle, degrees, at vertex j of the open clockwise ring.  ACOS is the caller's arccosine.
template <class Acos>
SZ_HD double interior_angle(int j, const double *x, const double *y, int n, Acos acos_fn) {
  const double eps = 2.220446049250313e-16, pi = 3.141592653589793;
  const int jp = (j == 0) ? n - 1 : j - 1, jn = (j + 1 == n) ? 0 : j + 1;
  const double v1x = -(x[j] - x[jp]), v1y = -(y[j] - y[jp]);   // v1 = -d(k), d = diff(xy)
  const double v2x = x[jn] - x[j], v2y = y[jn] - y[j];
  const double dot = v1x * v2x + v1y * v2y;
  double m1 = ::sqrt(v1x * v1x + v1y * v1y), m2 = ::sqrt(v2x * v2x + v2y * v2y);
  if (m1 == 0) m1 = eps;
  if (m2 == 0) m2 = eps;
  double a = acos_fn(dot / m1 / m2) * 180 / pi;
  if (convex_sign(j, x, y, n) == -1) a = 360 - a;
  return a;
}

// inpolygon.m:64-224 for one point against the ring (rx + ox, ry + oy)[nr] (closed or not: close_loops, :93-148)
SZ_HD bool inpolygon_ring(double x, double y, const double *rx, const double *ry, int nr, double ox, double oy) {
  if (nr <= 0) return false;
  double mnx = rx[0] + ox, mxx = mnx, mny = ry[0] + oy, mxy = mny;
  for (int k = 1; k < nr; ++k) {
    const double vx = rx[k] + ox, vy = ry[k] + oy;
    mnx = mind(mnx, vx); mxx = maxd(mxx, vx); mny = mind(mny, vy); mxy = maxd(mxy, vy);
  }
  if (!(x >= mnx && x <= mxx && y >= mny && y <= mxy)) return false;
  const bool closed = !(nr >= 3 && (rx[0] != rx[nr - 1] || ry[0] != ry[nr - 1]));
  const int edges = closed ? nr - 1 : nr;
  double sumdq = 0;
  bool on = false;
  for (int m = 0; m < edges; ++m) {
    const int m1 = (m + 1 == nr) ? 0 : m + 1;
    const double xa = rx[m] + ox, ya = ry[m] + oy, xb = rx[m1] + ox, yb = ry[m1] + oy;
    const double avx = absd(0.5 * (xa + xb)), avy = absd(0.5 * (ya + yb));
    double sf = maxd(avx, avy);
    sf = maxd(sf, avx * avy);
    const double ax = xa - x, ay = ya - y, bx = xb - x, by = yb - y;
    int qa, qb;
    { bool px = ax > 0, py = ay > 0; qa = (!px && py) ? 1 : (!px && !py) ? 2 : (px && !py) ? 3 : 0; }
    { bool px = bx > 0, py = by > 0; qb = (!px && py) ? 1 : (!px && !py) ? 2 : (px && !py) ? 3 : 0; }
    const double cp = ax * by - bx * ay;
    double scp = (cp > 0) ? 1 : (cp < 0 ? -1 : 0);
    if (absd(cp) < sf * SZC_EPS * 3) scp = 0;
    const double dp = ax * bx + ay * by;
    double dq = qb - qa;
    if (absd(dq) == 3) dq = -dq / 3;
    if (absd(dq) == 2) dq = 2 * scp;
    sumdq += dq;
    if (scp == 0 && dp <= 0) on = true;
  }
  return (sumdq != 0) || on;
}

// dsearchn (:73): index of the vertex of (x + ox, y + oy)[n] nearest to (px, py); the first of equals
SZ_HD int nearest_vertex(double px, double py, const double *x, const double *y, int n, double ox, double oy) {
  int best = 0;
  double bd = 0;
  for (int k = 0; k < n; ++k) {
    const double dx = (x[k] + ox) - px, dy = (y[k] + oy) - py;
    const double d = dx * dx + dy * dy;
    if (k == 0 || d < bd) { bd = d; best = k; }
  }
  return best;
}

// ---------------------------------------------------------------------------------------------------------------------
// frac_corner.m:1-188 for the PLAIN case: every corner to break off is convex, its two cut points lie
// inside the two adjacent edges without meeting the cut of a neighbouring corner, and the cut neither crosses another
// edge nor leaves an empty piece.  Then the reference's polyshape operations have closed forms: poly1 of :53-80 is the
// outline with the one corner replaced by its two cut points, poly2 the triangle, and subtract(poly, union(triangles))
// of :136-137 the outline with every broken corner replaced -- no boolean operation is needed.  Anything else returns 0
// (nothing written): the caller leaves that floe to frac_corner.m itself.
// polyshape conventions assumed (the reference holds no output of these functions to pin them):
// area / centroid = the shoelace forms relative to the first vertex (contact.hpp region_area / region_centroid), outlines
// listed clockwise from the first surviving stored vertex, a new triangle from its first cut point.
// Monte-Carlo points of the new floes (:108-110,:161-163) are rand: the caller draws them (the step reports floes whose
// mask is empty: SZ_STEP_NEEDS_MC).
// ---------------------------------------------------------------------------------------------------------------------
enum { PIECE_KIND = 0, PIECE_XI, PIECE_YI, PIECE_AREA, PIECE_MASS, PIECE_H, PIECE_INERTIA, PIECE_RMAX, PIECE_ALIVE, PIECE_KSI,
       PIECE_STRESS_FACTOR, PIECE_NV, PIECE_WORDS = 12 };

SZ_HD double ring_area_abs(const double *x, const double *y, int n) {
  if (n < 3) return 0;
  double s = 0;
  for (int i = 0; i < n; ++i) {
    const int k = (i + 1 == n) ? 0 : i + 1;
    const double ax = x[i] - x[0], ay = y[i] - y[0], bx = x[k] - x[0], by = y[k] - y[0];
    s += ax * by - bx * ay;
  }
  return absd(0.5 * s);
}
SZ_HD void ring_centroid_of(const double *x, const double *y, int n, double &cx, double &cy) {
  double s = 0, sx = 0, sy = 0;
  for (int i = 0; i < n; ++i) {
    const int k = (i + 1 == n) ? 0 : i + 1;
    const double ax = x[i] - x[0], ay = y[i] - y[0], bx = x[k] - x[0], by = y[k] - y[0];
    const double w = ax * by - bx * ay;
    s += w; sx += (ax + bx) * w; sy += (ay + by) * w;
  }
  cx = x[0] + sx / (3.0 * s); cy = y[0] + sy / (3.0 * s);
}
// PolygonMoments.m:14-32 on the closed ring (rows v1..vn, v1; the function appends v1 once more)
SZ_HD double polygon_moment(const double *cx, const double *cy, int nclosed, double h) {
  double ixx = 0, iyy = 0;
  for (int i = 0; i < nclosed; ++i) {
    const int k = (i + 1 == nclosed) ? 0 : i + 1;          // (the appended row is v1 again: row nclosed + 1 = row 1)
    const double x0 = cx[i], y0 = cy[i], x1 = (i + 1 == nclosed) ? cx[0] : cx[k], y1 = (i + 1 == nclosed) ? cy[0] : cy[k];
    const double wi = x0 * y1 - x1 * y0;
    ixx += wi * ((y0 + y1) * (y0 + y1) - y0 * y1);
    iyy += wi * ((x0 + x1) * (x0 + x1) - x0 * x1);
  }
  return absd(1.0 / 12 * ixx + 1.0 / 12 * iyy) * h * 920.0;
}
// do the closed segments a-b and c-d meet (touching counts)?
SZ_HD bool segments_meet(double ax, double ay, double bx, double by, double cx, double cy, double dx, double dy) {
  const double d1 = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax), d2 = (bx - ax) * (dy - ay) - (by - ay) * (dx - ax);
  const double d3 = (dx - cx) * (ay - cy) - (dy - cy) * (ax - cx), d4 = (dx - cx) * (by - cy) - (dy - cy) * (bx - cx);
  if (((d1 > 0 && d2 < 0) || (d1 < 0 && d2 > 0)) && ((d3 > 0 && d4 < 0) || (d3 < 0 && d4 > 0))) return true;
  auto on = [](double px, double py, double qx, double qy, double rx, double ry) {   // r on segment p-q, given collinear
    return mind(px, qx) <= rx && rx <= maxd(px, qx) && mind(py, qy) <= ry && ry <= maxd(py, qy);
  };
  if (d1 == 0 && on(ax, ay, bx, by, cx, cy)) return true;
  if (d2 == 0 && on(ax, ay, bx, by, dx, dy)) return true;
  if (d3 == 0 && on(cx, cy, dx, dy, ax, ay)) return true;
  if (d4 == 0 && on(cx, cy, dx, dy, bx, by)) return true;
  return false;
}
// one piece into the output: scalars of frac_corner.m:91-125 / :144-178 and the closed c_alpha ring
SZ_HD void put_piece(double kind, const double *px, const double *py, int m, double mass, double floe_area_field, double floe_ksi,
                     double stress_factor, double fx, double fy, double *rec, double *rx, double *ry) {
  double cx, cy;
  ring_centroid_of(px, py, m, cx, cy);
  const double area = ring_area_abs(px, py, m);
  double r2 = 0;
  for (int k = 0; k < m; ++k) { const double dx = px[k] - cx, dy = py[k] - cy; rx[k] = dx; ry[k] = dy; r2 = maxd(r2, dx * dx + dy * dy); }
  rx[m] = px[0] - cx; ry[m] = py[0] - cy;
  const double h = mass / (920.0 * area);
  rec[PIECE_KIND] = kind; rec[PIECE_XI] = fx + cx; rec[PIECE_YI] = fy + cy; rec[PIECE_AREA] = area; rec[PIECE_MASS] = mass; rec[PIECE_H] = h;
  rec[PIECE_INERTIA] = polygon_moment(rx, ry, m + 1, h); rec[PIECE_RMAX] = ::sqrt(r2); rec[PIECE_ALIVE] = (area < 1e4) ? 0.0 : 1.0;
  rec[PIECE_KSI] = area / floe_area_field * floe_ksi; rec[PIECE_STRESS_FACTOR] = stress_factor; rec[PIECE_NV] = (double)(m + 1);
}

// x, y[n]: the floe's outline without its closing copy (corners.m:63-68), grind[n] (modified: the quirk of :12-15).
// tmpx / tmpy: n + 2 * n doubles of scratch.  Output: rec[(1 + ng) * PIECE_WORDS], rings at rx / ry (capacity
// n + 1 + 5 ng), ring_off[2 + ng].  -> number of pieces (main first, then the triangles in vertex order), 0: not plain.
template <class Acos>
SZ_HD int frac_corner_plain(const double *x, const double *y, int n, unsigned char *grind, double mass, double area_field, double ksi,
                            double fx, double fy, double *tmpx, double *tmpy, double *rec, double *rx, double *ry, int *ring_off, Acos acos_fn) {
  if (n < 3) return 0;
  // :10-16
  double alph = 1e300;
  int ng = 0;
  bool equal_angles = false;
  for (int k = 0; k < n; ++k) {
    const double a = interior_angle(k, x, y, n, acos_fn);
    tmpx[k] = a;
    alph = mind(alph, a);
  }
  for (int k = 0; k < n; ++k) if (grind[k]) { ++ng; for (int m = 0; m < k; ++m) if (grind[m] && tmpx[m] == tmpx[k]) equal_angles = true; }
  if (equal_angles && grind[0] && grind[n - 1]) { grind[n - 1] = 0; --ng; }
  if (ng == 0) return 0;
  const double Anorm = 180.0 - 360.0 / n;
  const double poly_area = ring_area_abs(x, y, n);
  if (!(poly_area > 0)) return 0;
  // cut points of every ground corner (:19-52)
  double *c1x = tmpx, *c1y = tmpy, *c2x = tmpx + n, *c2y = tmpy + n;
  for (int k = 0; k < n; ++k) {
    if (!grind[k]) continue;
    if (interior_angle(k, x, y, n, acos_fn) >= 180) return 0;                   // (a reflex corner: the cut would add ice)
    const int kp = (k == 0) ? n - 1 : k - 1, kn = (k + 1 == n) ? 0 : k + 1;
    const double X1 = x[kp], Y1 = y[kp], X2 = x[k], Y2 = y[k], X3 = x[kn], Y3 = y[kn];
    const double d1 = ::sqrt((X1 - X2) * (X1 - X2) + (Y1 - Y2) * (Y1 - Y2)), d2 = ::sqrt((X3 - X2) * (X3 - X2) + (Y3 - Y2) * (Y3 - Y2));
    const double d = mind(d1, d2);
    double x1, y1, x2, y2;
    // :36-37 / :43-44 a short edge (under 120 m in x and in y) goes whole: the cut ends AT the neighbouring vertex.  Taken
    // here (round 4) when that vertex itself stays (its own cut would overlap this one: union / subtract of the
    // reference, :136-137, not this closed form) and the outline has a vertex to spare.
    const bool s1 = absd(X1 - X2) < 120 && absd(Y1 - Y2) < 120, s2 = absd(X3 - X2) < 120 && absd(Y3 - Y2) < 120;
    if ((s1 || s2) && n < 5) return 0;
    if ((s1 && grind[kp]) || (s2 && grind[kn])) return 0;
    if (s1) { x1 = X1; y1 = Y1; }
    else if (absd(d * alph / Anorm / 5) < 120) { x1 = X2 + 120 / d1 * (X1 - X2); y1 = Y2 + 120 / d1 * (Y1 - Y2); }
    else { x1 = X2 + d / d1 * alph / Anorm * (X1 - X2) / 5; y1 = Y2 + d / d1 * alph / Anorm * (Y1 - Y2) / 5; }
    if (s2) { x2 = X3; y2 = Y3; }
    else if (absd(d * alph / Anorm / 5) < 120) { x2 = X2 + 120 / d2 * (X3 - X2); y2 = Y2 + 120 / d2 * (Y3 - Y2); }
    else { x2 = X2 + d / d2 * alph / Anorm * (X3 - X2) / 5; y2 = Y2 + d / d2 * alph / Anorm * (Y3 - Y2) / 5; }
    if (absd(x1 - x2) < 0.1) { if (s1 || s2) return 0; x1 = x1 + 1; }          // :50-52 (with a cut at a vertex: not this form)
    c1x[k] = x1; c1y[k] = y1; c2x[k] = x2; c2y[k] = y2;
    // the cut must stay inside its two edges (or end at their far vertex: a short edge) and meet no other edge of the outline
    const double t1 = ::sqrt((x1 - X2) * (x1 - X2) + (y1 - Y2) * (y1 - Y2)), t2 = ::sqrt((x2 - X2) * (x2 - X2) + (y2 - Y2) * (y2 - Y2));
    if (!(t1 > 0 && (s1 || t1 < d1) && t2 > 0 && (s2 || t2 < d2))) return 0;
    const int kpp = (kp == 0) ? n - 1 : kp - 1, knn = (kn + 1 == n) ? 0 : kn + 1;
    for (int e = 0; e < n; ++e) {
      const int e1 = (e + 1 == n) ? 0 : e + 1;
      if (e == kp || e == k) continue;                                          // the two edges at the corner
      if (s1 && e == kpp) continue;                                             // (ends where the cut starts: below)
      if (s2 && e == kn) continue;                                              // (starts where the cut ends)
      if (segments_meet(x1, y1, x2, y2, x[e], y[e], x[e1], y[e1])) return 0;
    }
    // a cut from a vertex must leave the outline's edge there at an angle (in line with it, polyshape drops the vertex)
    if (s1 && (x[kp] - x[kpp]) * (y2 - y[kpp]) - (y[kp] - y[kpp]) * (x2 - x[kpp]) == 0) return 0;
    if (s2 && (x[kn] - x1) * (y[knn] - y1) - (y[kn] - y1) * (x[knn] - x1) == 0) return 0;
  }
  // neighbouring corners: their cuts on the shared edge must not meet
  for (int k = 0; k < n; ++k) {
    const int kn = (k + 1 == n) ? 0 : k + 1;
    if (!(grind[k] && grind[kn])) continue;
    const double L = ::sqrt((x[kn] - x[k]) * (x[kn] - x[k]) + (y[kn] - y[k]) * (y[kn] - y[k]));
    const double ta = ::sqrt((c2x[k] - x[k]) * (c2x[k] - x[k]) + (c2y[k] - y[k]) * (c2y[k] - y[k]));
    const double tb = ::sqrt((c1x[kn] - x[kn]) * (c1x[kn] - x[kn]) + (c1y[kn] - y[kn]) * (c1y[kn] - y[kn]));
    if (!(ta + tb < L)) return 0;
  }
  // the main piece: every ground corner replaced by its two cut points (:133-180)
  double *mx = rx, *my = ry;                       // built in place at the head of the ring output, then re-centred
  int m = 0;
  double *bx = tmpx + 2 * n, *by = tmpy + 2 * n;   // (scratch beyond the cut points: n + ng values)
  for (int k = 0; k < n; ++k) {
    if (grind[k]) {
      // (a cut point that IS the neighbouring vertex -- a short edge -- is in the outline already)
      const int kp = (k == 0) ? n - 1 : k - 1, kn = (k + 1 == n) ? 0 : k + 1;
      if (!(c1x[k] == x[kp] && c1y[k] == y[kp])) { bx[m] = c1x[k]; by[m] = c1y[k]; ++m; }
      if (!(c2x[k] == x[kn] && c2y[k] == y[kn])) { bx[m] = c2x[k]; by[m] = c2y[k]; ++m; }
    }
    else { bx[m] = x[k]; by[m] = y[k]; ++m; }
  }
  if (m < 3) return 0;
  const double new_area = ring_area_abs(bx, by, m);
  if (!(new_area > 0)) return 0;
  int np = 0, at = 0;
  ring_off[0] = 0;
  put_piece(0.0, bx, by, m, mass * new_area / poly_area, area_field, ksi, new_area / poly_area, fx, fy, rec, mx, my);
  at += m + 1; ++np; ring_off[np] = at;
  // the triangles (:53-129), each with the Atot of ITS single cut
  for (int k = 0; k < n; ++k) {
    if (!grind[k]) continue;
    double tx[3] = {c1x[k], c2x[k], x[k]}, ty[3] = {c1y[k], c2y[k], y[k]};
    const double ta = ring_area_abs(tx, ty, 3);
    if (!(ta > 0)) return 0;
    // area of poly1: the outline with this one corner replaced = the outline less the triangle (the corner is convex)
    double *px = bx + m, *py = by + m;              // scratch for poly1 (n + 1 vertices)
    int q = 0;
    for (int v = 0; v < n; ++v) {
      if (v == k) { px[q] = c1x[k]; py[q] = c1y[k]; ++q; px[q] = c2x[k]; py[q] = c2y[k]; ++q; }
      else { px[q] = x[v]; py[q] = y[v]; ++q; }
    }
    const double a1 = ring_area_abs(px, py, q);
    double Atot = 0;
    if (a1 > 10) Atot += a1;
    if (ta > 10) Atot += ta;
    if (!(Atot > 0)) return 0;
    const double Mtot = mass * Atot / poly_area;
    // clockwise, from the first cut point
    const double sgn_area = (tx[1] - tx[0]) * (ty[2] - ty[0]) - (tx[2] - tx[0]) * (ty[1] - ty[0]);
    if (sgn_area > 0) { const double sx = tx[1], sy = ty[1]; tx[1] = tx[2]; ty[1] = ty[2]; tx[2] = sx; ty[2] = sy; }
    put_piece(1.0, tx, ty, 3, Mtot * ta / Atot, area_field, ksi, ta / Atot, fx, fy, rec + (size_t)np * PIECE_WORDS, rx + at, ry + at);
    at += 4; ++np; ring_off[np] = at;
  }
  return np;
}

} // namespace szk
