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

} // namespace szk
