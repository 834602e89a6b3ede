// TEST INFRASTRUCTURE ONLY (oracle).  Never linked, imported or called by the product path
// (subzero_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
// checker / the timed CPU baseline.
//
// CPU restatement of SubZero's per-timestep floe-interaction hot path, following the reference's
// MATLAB text function by function and line by line (quirks included):
//
//   or_step                 floe_interactions_all.m:9-285
//   floe_interactions       collisions/floe_interactions.m:1-199
//   polyclip                polyclip.m:63-80   (clip itself = the reference's own Clipper, oracle/_ref)
//   interx                  collisions/InterX.m:54-81
//   p_poly_dist             polygon_operations/p_poly_dist.m:111-288
//   inpolygon               polygon_operations/inpolygon.m:64-224
//   calc_trajectory         calc_trajectory.m:1-241
//
// MathWorks builtins that are not in the reference repo are replaced by their documented definitions
// (SURVEY.md section 8c): polyshape area/centroid = shoelace (pinned against MATLAB's own cached values
// in the FloeShapes.mat fixture), polyarea = its published one-liner, dsearchn = brute-force nearest
// vertex, interp2 = bilinear, cart2pol = atan2/hypot, unique(...,'rows') = lexicographic sort + dedupe,
// sum/mean = left-to-right, except the means over the Monte-Carlo points of calc_trajectory.m:139,163-165, which
// use a fixed order that 256 parallel workers can follow: point q is added to partial q mod 256 (in increasing
// q), then the partials are folded pairwise, p[t] += p[t + s] for s = 128, 64, ..., 1 (struct McSum).  MATLAB
// documents no order for sum; this one is also more accurate than left-to-right.
// sin/cos/atan2 come from or_math.h (self-contained, <= 1-2 ulp).
//
// PARITY STATUS: the C++ clip is pinned by the reference's own known answer (private/clipper_test.m)
// and by being the reference's own source.  The MATLAB layers are "parity unpinned" by the reference
// (it ships no golden output and no MATLAB/Octave exists here); they are pinned only through the
// FloeShapes.mat area/centroid values and by review against the .m text cited at each function.
#include "subzero_oracle.h"
#include "or_math.h"

#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <limits>
#include <utility>
#include <vector>

namespace {

typedef std::vector<double> Vec;
const double INF = std::numeric_limits<double>::infinity();
const double NaN = std::numeric_limits<double>::quiet_NaN();
const double EPS = 2.220446049250313e-16;

// Which definitions the run uses where the reference leaves a MathWorks builtin unspecified (or_set_mode):
//   bit 0 (OR_MODE_LIBM)    sin / cos / atan2 from the host's libm instead of or_math.h
//   bit 1 (OR_MODE_SEQSUM)  the means over the Monte-Carlo points (calc_trajectory.m:139,163-165) summed left to
//                           right, as the .m text reads, instead of the 256-partial fold
//   bits 8-10               start vertex (0-3) and direction (bit 10) of the domain-box ring handed to the 'dif' clip
//                           (floe_interactions.m:31-32: holes(floe2.poly) fixes neither)
// Mode 0 is the definition the device path reproduces bit for bit; mode 3 is independent of the kernels' own trig code
// and summation order, and the device must stay within the north-star tolerance (1e-10 relative) of it.
int g_mode = 0;
inline void t_sincos(double x, double *s, double *c) {
  if (g_mode & 1) { *s = sin(x); *c = cos(x); } else orm_sincos(x, s, c);
}
inline double t_atan2(double y, double x) { return (g_mode & 1) ? atan2(y, x) : orm_atan2(y, x); }

struct Ring {
  Vec x, y;
};

// ---------------------------------------------------------------------------------------------
// clip function (dlopen'ed so the oracle itself has no link-time dependency on either checker)
// ---------------------------------------------------------------------------------------------
typedef int (*ref_clip_fn)(const int64_t *, const int64_t *, const int32_t *, int, const int64_t *,
                           const int64_t *, const int32_t *, int, int, int, int, int64_t *, int64_t *, int,
                           int32_t *, int, int32_t *);
typedef int (*host_clip_fn)(int, const int64_t *, const int64_t *, const int32_t *, const int64_t *,
                            const int64_t *, const int32_t *, int, int, int, int64_t *, int64_t *, int64_t,
                            int32_t *, int64_t, int32_t *, int32_t *);
ref_clip_fn g_ref_clip = nullptr;
host_clip_fn g_host_clip = nullptr;

// MATLAB int64(v * 2^32): round half away from zero, saturate (polyclip.m:63,66)
int64_t to_fixed(double v) {
  double s = v * 4294967296.0;
  if (s != s) return 0;
  if (s >= 9223372036854775807.0) return INT64_MAX;
  if (s <= -9223372036854775808.0) return INT64_MIN;
  double t = trunc(s);
  double f = s - t;
  if (f >= 0.5) t += 1.0;
  else if (f <= -0.5) t -= 1.0;
  return (int64_t)t;
}

// polyclip.m:63-80.  method: 0 'dif', 1 'int', 2 'xor', 3 'union'.  ok=false -> the mex would have raised.
std::vector<Ring> polyclip(const Vec &x1, const Vec &y1, const Vec &x2, const Vec &y2, int method, bool &ok) {
  std::vector<Ring> out;
  ok = true;
  int n1 = (int)x1.size(), n2 = (int)x2.size();
  std::vector<int64_t> ax(n1), ay(n1), bx(n2), by(n2);
  for (int i = 0; i < n1; ++i) { ax[i] = to_fixed(x1[i]); ay[i] = to_fixed(y1[i]); }
  for (int i = 0; i < n2; ++i) { bx[i] = to_fixed(x2[i]); by[i] = to_fixed(y2[i]); }
  int cap = 4 * (n1 + n2) + 64;
  for (int attempt = 0; attempt < 6; ++attempt, cap *= 8) {
    std::vector<int64_t> ox(cap), oy(cap);
    std::vector<int32_t> roff(cap + 1);
    int32_t aoff[2] = {0, n1}, boff[2] = {0, n2};
    int32_t nr = 0;
    int rc, st = 0;
    if (g_ref_clip) {
      rc = g_ref_clip(ax.data(), ay.data(), aoff, 1, bx.data(), by.data(), boff, 1, method, 0, 0, ox.data(),
                      oy.data(), cap, roff.data(), cap, &nr);
      if (rc == 3) continue;
      st = rc;
    } else {
      int32_t proff[2] = {0, 0};
      int32_t stat = 0;
      rc = g_host_clip(1, ax.data(), ay.data(), aoff, bx.data(), by.data(), boff, method, 0, 0, ox.data(),
                       oy.data(), cap, roff.data(), cap, proff, &stat);
      if (rc == 3) continue;
      st = stat;
      nr = proff[1];
    }
    if (st != 0) { ok = false; return out; }
    out.resize(nr);
    for (int r = 0; r < nr; ++r) {
      for (int v = roff[r]; v < roff[r + 1]; ++v) {
        out[r].x.push_back((double)ox[v] / 4294967296.0); // polyclip.m:67
        out[r].y.push_back((double)oy[v] / 4294967296.0);
      }
    }
    return out;
  }
  ok = false;
  return out;
}

// ---------------------------------------------------------------------------------------------
// MathWorks builtins, by definition
// ---------------------------------------------------------------------------------------------
// area(polyshape(X,Y)) of one ring: |shoelace|, evaluated relative to the first vertex
double ring_area_signed(const double *x, const double *y, int n) {
  if (n < 3) return 0;
  double s = 0;
  for (int i = 0; i < n; ++i) {
    int k = (i + 1 == n) ? 0 : i + 1;
    double ax = x[i] - x[0], ay = y[i] - y[0], bx = x[k] - x[0], by = y[k] - y[0];
    s += ax * by - bx * ay;
  }
  return 0.5 * s;
}
double ring_area(const double *x, const double *y, int n) { return fabs(ring_area_signed(x, y, n)); }

// centroid(polyshape(X,Y)) of one ring, relative to the first vertex
void ring_centroid(const double *x, const double *y, int n, double &cx, double &cy) {
  double s = 0, sx = 0, sy = 0;
  for (int i = 0; i < n; ++i) {
    int k = (i + 1 == n) ? 0 : i + 1;
    double ax = x[i] - x[0], ay = y[i] - y[0], bx = x[k] - x[0], by = y[k] - y[0];
    double w = ax * by - bx * ay;
    s += w;
    sx += (ax + bx) * w;
    sy += (ay + by) * w;
  }
  cx = x[0] + sx / (3.0 * s);
  cy = y[0] + sy / (3.0 * s);
}

// polyarea.m: abs(sum((x([2:n 1]) - x) .* (y([2:n 1]) + y)) / 2)
double polyarea(const double *x, const double *y, int n) {
  double s = 0;
  for (int i = 0; i < n; ++i) {
    int k = (i + 1 == n) ? 0 : i + 1;
    s += (x[k] - x[i]) * (y[k] + y[i]);
  }
  return fabs(s / 2);
}

double hyp(double a, double b) { return sqrt(a * a + b * b); }

// clipper(Subj, Clip, method, subjFill, clipFill) with several contours per operand (mexclipper.cpp:13-62,204-305), rings
// in metres converted as polyclip.m:63-67 does; needs the reference build (the host build's batch entry is single-ring)
std::vector<Ring> polyclip_paths(const std::vector<Ring> &A, const std::vector<Ring> &B, int method, int fs, int fc, bool &ok) {
  std::vector<Ring> out;
  ok = true;
  if (!g_ref_clip) { ok = false; return out; }
  std::vector<int64_t> ax, ay, bx, by;
  std::vector<int32_t> aoff(1, 0), boff(1, 0);
  for (size_t r = 0; r < A.size(); ++r) { for (size_t k = 0; k < A[r].x.size(); ++k) { ax.push_back(to_fixed(A[r].x[k])); ay.push_back(to_fixed(A[r].y[k])); } aoff.push_back((int32_t)ax.size()); }
  for (size_t r = 0; r < B.size(); ++r) { for (size_t k = 0; k < B[r].x.size(); ++k) { bx.push_back(to_fixed(B[r].x[k])); by.push_back(to_fixed(B[r].y[k])); } boff.push_back((int32_t)bx.size()); }
  ax.push_back(0); ay.push_back(0); bx.push_back(0); by.push_back(0);
  int cap = 4 * (int)(ax.size() + bx.size()) + 64;
  for (int attempt = 0; attempt < 6; ++attempt, cap *= 8) {
    std::vector<int64_t> ox(cap), oy(cap);
    std::vector<int32_t> roff(cap + 1);
    int32_t nr = 0;
    int rc = g_ref_clip(ax.data(), ay.data(), aoff.data(), (int)A.size(), bx.data(), by.data(), boff.data(), (int)B.size(), method, fs, fc,
                        ox.data(), oy.data(), cap, roff.data(), cap, &nr);
    if (rc == 3) continue;
    if (rc != 0) { ok = false; return out; }
    out.resize(nr);
    for (int r = 0; r < nr; ++r)
      for (int v = roff[r]; v < roff[r + 1]; ++v) { out[r].x.push_back((double)ox[v] / 4294967296.0); out[r].y.push_back((double)oy[v] / 4294967296.0); }
    return out;
  }
  ok = false;
  return out;
}
// area(polyshape) of a region given by its rings (outer rings and holes come out of the clip in opposite senses): the
// signed shoelace sums, each relative to its ring's first vertex, added up
double rings_area(const std::vector<Ring> &R) {
  double s = 0;
  for (size_t r = 0; r < R.size(); ++r) s += ring_area_signed(R[r].x.data(), R[r].y.data(), (int)R[r].x.size());
  return fabs(s);
}


// ---------------------------------------------------------------------------------------------
// InterX.m:54-81 -- all crossings of two polylines (here: closed rings), sorted unique rows
// ---------------------------------------------------------------------------------------------
void interx(const double *x1, const double *y1, int n1, const double *x2, const double *y2, int n2,
            Vec &px, Vec &py) {
  px.clear(); py.clear();
  int m1 = n1 - 1, m2 = n2 - 1;
  if (m1 < 1 || m2 < 1) return;
  Vec dx1(m1), dy1(m1), S1(m1), dx2(m2), dy2(m2), S2(m2);
  for (int i = 0; i < m1; ++i) {
    dx1[i] = x1[i + 1] - x1[i]; dy1[i] = y1[i + 1] - y1[i];
    S1[i] = dx1[i] * y1[i] - dy1[i] * x1[i];           // :60
  }
  for (int j = 0; j < m2; ++j) {
    dx2[j] = x2[j + 1] - x2[j]; dy2[j] = y2[j + 1] - y2[j];
    S2[j] = dx2[j] * y2[j] - dy2[j] * x2[j];           // :61
  }
  std::vector<std::pair<double, double> > pts;
  // [i,j] = find(C1 & C2): column-major, j outer
  for (int j = 0; j < m2; ++j) {
    for (int i = 0; i < m1; ++i) {
      double a = dx1[i] * y2[j] - dy1[i] * x2[j];          // :63
      double b = dx1[i] * y2[j + 1] - dy1[i] * x2[j + 1];
      bool C1 = ((a - S1[i]) * (b - S1[i])) <= 0;
      double c = y1[i] * dx2[j] - x1[i] * dy2[j];          // :64
      double d = y1[i + 1] * dx2[j] - x1[i + 1] * dy2[j];
      bool C2 = ((c - S2[j]) * (d - S2[j])) <= 0;
      if (!(C1 && C2)) continue;
      double L = dy2[j] * dx1[i] - dy1[i] * dx2[j];        // :72
      if (L == 0) continue;                                 // :73
      double X = (dx2[j] * S1[i] - dx1[i] * S2[j]) / L;    // :76-77
      double Y = (dy2[j] * S1[i] - dy1[i] * S2[j]) / L;
      pts.push_back(std::make_pair(X, Y));
    }
  }
  std::sort(pts.begin(), pts.end());
  for (size_t k = 0; k < pts.size(); ++k) {
    if (k > 0 && pts[k].first == pts[k - 1].first && pts[k].second == pts[k - 1].second) continue;
    px.push_back(pts[k].first); py.push_back(pts[k].second);
  }
}

// ---------------------------------------------------------------------------------------------
// inpolygon.m:64-224 (vendored copy of the MathWorks routine), single loop, one point at a time
// ---------------------------------------------------------------------------------------------
bool inpolygon1(double x, double y, const Vec &xv0, const Vec &yv0) {
  int n = (int)xv0.size();
  if (n == 0) return false;
  double mnx = xv0[0], mxx = xv0[0], mny = yv0[0], mxy = yv0[0];
  for (int k = 1; k < n; ++k) {
    mnx = std::min(mnx, xv0[k]); mxx = std::max(mxx, xv0[k]);
    mny = std::min(mny, yv0[k]); mxy = std::max(mxy, yv0[k]);
  }
  if (!(x >= mnx && x <= mxx && y >= mny && y <= mxy)) return false; // :76-81
  Vec xv = xv0, yv = yv0;
  if (n >= 3 && (xv[0] != xv[n - 1] || yv[0] != yv[n - 1])) { xv.push_back(xv[0]); yv.push_back(yv[0]); } // close_loops
  int Nv = (int)xv.size();
  double sumdq = 0;
  bool on = false;
  for (int m = 0; m + 1 < Nv; ++m) {
    double avx = fabs(0.5 * (xv[m] + xv[m + 1]));          // :165-168
    double avy = fabs(0.5 * (yv[m] + yv[m + 1]));
    double sf = std::max(avx, avy);
    sf = std::max(sf, avx * avy);
    double ax = xv[m] - x, ay = yv[m] - y, bx = xv[m + 1] - x, by = yv[m + 1] - y; // :171-172
    int qa, qb;
    { bool px = ax > 0, py = ay > 0; qa = (!px && py) ? 1 : (!px && !py) ? 2 : (px && !py) ? 3 : 0; }
    { bool px = bx > 0, py = by > 0; qb = (!px && py) ? 1 : (!px && !py) ? 2 : (px && !py) ? 3 : 0; }
    double cp = ax * by - bx * ay;                           // :188
    double scp = (cp > 0) ? 1 : (cp < 0 ? -1 : 0);
    if (fabs(cp) < sf * EPS * 3) scp = 0;                    // :196-198
    double dp = ax * bx + ay * by;                           // :200
    double dq = qb - qa;                                     // :203
    if (fabs(dq) == 3) dq = -dq / 3;                         // :208-209
    if (fabs(dq) == 2) dq = 2 * scp;                         // :210-211
    sumdq += dq;
    if (scp == 0 && dp <= 0) on = true;                      // :222
  }
  return (sumdq != 0) || on;
}

// ---------------------------------------------------------------------------------------------
// p_poly_dist.m:111-288 with find_in_out = true (4 inputs, 1 output -> :115-116); only |d_min| is
// consumed by the caller (floe_interactions.m:127), so the sign (inpolygon, :285-288) is applied too
// but is immaterial.
// ---------------------------------------------------------------------------------------------
void p_poly_dist(const Vec &xp, const Vec &yp, const Vec &xv0, const Vec &yv0, Vec &dmin) {
  Vec xv = xv0, yv = yv0;
  int nv = (int)xv.size();
  if (xv[0] != xv[nv - 1] || yv[0] != yv[nv - 1]) { xv.push_back(xv[0]); yv.push_back(yv[0]); nv++; } // :133-141
  int np = (int)xp.size();
  dmin.assign(np, 0.0);
  int ns = nv - 1;
  Vec vds(ns), ct(ns), st(ns), p1r1(ns), p1r2(ns);
  for (int k = 0; k < ns; ++k) {
    double dvx = xv[k + 1] - xv[k], dvy = yv[k + 1] - yv[k];
    vds[k] = hyp(dvx, dvy);                                  // :163
    ct[k] = dvx / vds[k]; st[k] = dvy / vds[k];              // :185-186
    p1r1[k] = ct[k] * xv[k] + st[k] * yv[k];                 // :195-196
    p1r2[k] = -st[k] * xv[k] + ct[k] * yv[k];
  }
  for (int j = 0; j < np; ++j) {
    double dpv_min = INF; int I_dpv = 0;
    for (int k = 0; k < nv; ++k) {                           // :151-155
      double d = hyp(xv[k] - xp[j], yv[k] - yp[j]);
      if (d < dpv_min) { dpv_min = d; I_dpv = k; }
    }
    double cr_min = NaN; int I_cr = 0;
    for (int k = 0; k < ns; ++k) {
      double r = (xp[j] * ct[k] + yp[j] * st[k]) - p1r1[k];  // :210,214
      double cr = (xp[j] * (-st[k]) + yp[j] * ct[k]) - p1r2[k]; // :211,215
      if (r > 0 && r < vds[k]) {                             // :256
        double a = fabs(cr);
        if (cr_min != cr_min || a < cr_min) { cr_min = a; I_cr = k; } // :264
      }
    }
    bool cond1 = (cr_min != cr_min);                         // :271
    bool cond2 = (I_cr != I_dpv) && ((cr_min - dpv_min) > 0);// :275
    double d = (cond1 || cond2) ? dpv_min : cr_min;          // :280-281
    if (inpolygon1(xp[j], yp[j], xv, yv)) d = -d;            // :285-288
    dmin[j] = d;
  }
}

// ---------------------------------------------------------------------------------------------
// collisions/floe_interactions.m
// ---------------------------------------------------------------------------------------------
struct Body { // the fields of floe1 / potentialInteractions(k) / floebound that the function reads
  double Xi, Yi, Ui, Vi, ksi_ice, h, area;
};
struct Contact {
  double fx, fy, px, py, overlap;
};

// returns false if the clipper failed (MATLAB would have raised an error)
bool floe_interactions(const Vec &c1x_in, const Vec &c1y_in, const Body &f1, const Vec &c2x_in,
                       const Vec &c2y_in, const Body &f2, bool boundary, const double *c2b, bool PERIODIC,
                       double Modulus, double dt, double mu, double nu, std::vector<Contact> &rows) {
  rows.clear();
  // :10-21
  double h1 = f1.h, h2 = f2.h;
  double r1 = sqrt(f1.area), r2 = sqrt(f2.area);
  double Force_factor = Modulus * (h1 * h2) / (h1 * r2 + h2 * r1);
  double overlap0 = 0;
  if (boundary) Force_factor = Modulus * h1 / r1;              // :13-14 (floebound has field 'alive')
  else if (r1 > 1e5 || r2 > 1e5) {                             // :15-19
    r1 = std::min(r1, r2);
    h1 = std::min(h1, h2);
    Force_factor = Modulus * h1 / r1;
  }
  double G = Modulus / (2 * (1 + nu));                         // :21

  Vec c1x = c1x_in, c1y = c1y_in, c2x = c2x_in, c2y = c2y_in;  // :25-27
  bool ok;
  std::vector<Ring> reg = polyclip(c1x, c1y, c2x, c2y, boundary ? 0 : 1, ok); // :29 / :34
  if (!ok) return false;
  if (boundary && !reg.empty()) {                              // :35-40
    if (polyarea(reg[0].x.data(), reg[0].y.data(), (int)reg[0].x.size()) / f1.area > 0.75) overlap0 = INF;
  }
  Vec Ar;
  if (reg.empty()) Ar.push_back(0);                            // :43-44
  else for (size_t k = 0; k < reg.size(); ++k) Ar.push_back(ring_area(reg[k].x.data(), reg[k].y.data(), (int)reg[k].x.size())); // :46-50

  // :54-60 merge flags
  {
    double mx1 = c1x[0], mn1 = c1x[0], my1 = c1y[0], mny1 = c1y[0];
    for (size_t k = 1; k < c1x.size(); ++k) {
      mx1 = std::max(mx1, c1x[k]); mn1 = std::min(mn1, c1x[k]);
      my1 = std::max(my1, c1y[k]); mny1 = std::min(mny1, c1y[k]);
    }
    double bxmax = c2b[0], bxmin = c2b[0], bymax = c2b[1], bymin = c2b[1];
    for (int k = 1; k < 5; ++k) {
      bxmax = std::max(bxmax, c2b[2 * k]); bxmin = std::min(bxmin, c2b[2 * k]);
      bymax = std::max(bymax, c2b[2 * k + 1]); bymin = std::min(bymin, c2b[2 * k + 1]);
    }
    double bx[4] = {c2b[0], c2b[2], c2b[4], c2b[6]}, by[4] = {c2b[1], c2b[3], c2b[5], c2b[7]};
    double barea = ring_area(bx, by, 4);
    bool inside = (mx1 < bxmax && mn1 > bxmin && my1 < bymax && mny1 > bymin);
    if (inside || f2.area < 0.95 * barea || PERIODIC) {
      double sumAr = 0;
      for (size_t k = 0; k < Ar.size(); ++k) sumAr += Ar[k];
      if (sumAr / f1.area > 0.55) overlap0 = INF;
      else if (sumAr / f2.area > 0.55) overlap0 = -INF;
    }
  }
  // :62-67 close the rings
  if (hyp(c1x[0] - c1x.back(), c1y[0] - c1y.back()) > 1) { c1x.push_back(c1x[0]); c1y.push_back(c1y[0]); }
  if (hyp(c2x[0] - c2x.back(), c2y[0] - c2y.back()) > 1) { c2x.push_back(c2x[0]); c2y.push_back(c2y[0]); }

  Vec Px, Py;
  interx(c1x.data(), c1y.data(), (int)c1x.size(), c2x.data(), c2y.data(), (int)c2x.size(), Px, Py); // :70
  if (Px.size() < 2 || std::isinf(overlap0) || reg.empty()) {  // :71-74
    Contact c = {0, 0, 0, 0, overlap0};
    rows.push_back(c);
    return true;
  }
  int N1 = (int)c1x.size() - 1, N2 = (int)c2x.size() - 1;      // :78-79
  double Amin = std::min(N1, N2) * 100 / 1.75;
  {                                                             // :83
    std::vector<Ring> r2v; Vec a2;
    for (size_t k = 0; k < reg.size(); ++k) if (!(Ar[k] < Amin)) { r2v.push_back(reg[k]); a2.push_back(Ar[k]); }
    reg.swap(r2v); Ar.swap(a2);
  }
  int N_contact = (int)reg.size();
  for (int k = 0; k < N_contact; ++k) {
    const Vec &X = reg[k].x, &Y = reg[k].y;
    int nr = (int)X.size();
    double cx, cy;
    ring_centroid(X.data(), Y.data(), nr, cx, cy);              // :96-97
    // :98-100 dsearchn: nearest region vertex of every crossing point
    Vec ppx, ppy;
    for (size_t q = 0; q < Px.size(); ++q) {
      double best = INF; int bi = 0;
      for (int v = 0; v < nr; ++v) {
        double ddx = X[v] - Px[q], ddy = Y[v] - Py[q];
        double d2 = ddx * ddx + ddy * ddy;
        if (d2 < best) { best = d2; bi = v; }
      }
      if (sqrt(best) < 1) { ppx.push_back(X[bi]); ppy.push_back(Y[bi]); }
    }
    int m = (int)ppx.size();
    double fdx = 0, fdy = 0, dl = 0, pcx = cx, pcy = cy;
    if (Ar[k] == 0) { fdx = fdy = 0; pcx = pcy = 0; dl = 0; }   // :103-106
    else if (m == 2) {                                          // :107-112
      double xgh = ppx[1] - ppx[0], ygh = ppy[1] - ppy[0];
      double b = sqrt(xgh * xgh + ygh * ygh);
      fdx = -ygh / b; fdy = xgh / b;
      dl = b;
    } else if (m == 0) { fdx = fdy = 0; dl = 0; }               // :113-116
    else {                                                      // :117-137
      Vec xv = X, yv = Y; xv.push_back(X[0]); yv.push_back(Y[0]);
      Vec xm(nr), ym(nr), b(nr), nx(nr), ny(nr);
      for (int e = 0; e < nr; ++e) {
        double xgh = xv[e + 1] - xv[e], ygh = yv[e + 1] - yv[e];
        xm[e] = (xv[e + 1] + xv[e]) / 2; ym[e] = (yv[e + 1] + yv[e]) / 2;
        b[e] = sqrt(xgh * xgh + ygh * ygh);
        nx[e] = -ygh / b[e]; ny[e] = xgh / b[e];
        double xt = xm[e] + nx[e] / 100, yt = ym[e] + ny[e] / 100;
        if (!inpolygon1(xt, yt, xv, yv)) { nx[e] = -nx[e]; ny[e] = -ny[e]; }   // :123-124
      }
      Vec d;
      p_poly_dist(xm, ym, c1x, c1y, d);                         // :126
      int non = 0;
      double sfx = 0, sfy = 0, sb = 0;
      for (int e = 0; e < nr; ++e) {
        if (fabs(d[e]) < 1e-8) {                                // :127
          ++non;
          sfx += (-Force_factor * b[e]) * nx[e];                // :125,129  Fn = -Force_factor*(b*[1 1]).*n
          sfy += (-Force_factor * b[e]) * ny[e];
          sb += b[e];
        }
      }
      if (non < nr && non > 0) {                                // :128-131
        double nrm = sqrt(sfx * sfx + sfy * sfy);
        fdx = sfx / nrm; fdy = sfy / nrm;
        dl = sb / non;
      } else { fdx = fdy = 0; dl = 0; }
    }
    // :140-155 direction check by re-clipping the ring shifted 1 m along the force direction
    Vec X1n = c1x, Y1n = c1y;
    if (dl < 0.1) { fdx = 0; fdy = 0; }
    else for (size_t q = 0; q < X1n.size(); ++q) { X1n[q] = c1x[q] + fdx; Y1n[q] = c1y[q] + fdy; }
    std::vector<Ring> regn = polyclip(X1n, Y1n, c2x, c2y, boundary ? 0 : 1, ok);
    if (!ok) return false;
    for (size_t ii = 0; ii < regn.size(); ++ii) {               // :156-165
      std::vector<Ring> rn = polyclip(regn[ii].x, regn[ii].y, X, Y, 1, ok);
      if (!ok) return false;
      if (!rn.empty()) {
        double Anew = polyarea(regn[ii].x.data(), regn[ii].y.data(), (int)regn[ii].x.size());
        if (Anew / Ar[k] - 1 > 0) { fdx = -fdx; fdy = -fdy; }
      }
    }
    double Fx = fdx * Ar[k] * Force_factor, Fy = fdy * Ar[k] * Force_factor; // :167
    // :170-183 tangential force
    double v1x = f1.Ui + f1.ksi_ice * (pcx - f1.Xi), v1y = f1.Vi + f1.ksi_ice * (pcy - f1.Yi);
    double v2x = f2.Ui + f2.ksi_ice * (pcx - f2.Xi), v2y = f2.Vi + f2.ksi_ice * (pcy - f2.Yi);
    double vtx = v1x - v2x, vty = v1y - v2y;
    double dtx, dty;
    if (std::max(fabs(vtx), fabs(vty)) == 0) { dtx = 0; dty = 0; }
    else { double nv = sqrt(vtx * vtx + vty * vty); dtx = vtx / nv; dty = vty / nv; }
    double vn = sqrt(vtx * vtx + vty * vty);
    double coef = -(dtx * vtx + dty * vty) * dl * G * vn;       // :178, left to right
    double ftx = coef * dtx * dt, fty = coef * dty * dt;
    double nF = sqrt(Fx * Fx + Fy * Fy);
    if (sqrt(ftx * ftx + fty * fty) > mu * nF) {                // :180-181
      ftx = -mu * nF * dtx; fty = -mu * nF * dty;
    }
    Contact c = {Fx + ftx, Fy + fty, pcx, pcy, Ar[k]};          // :186-187
    rows.push_back(c);
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// calc_trajectory.m
// ---------------------------------------------------------------------------------------------
// the fixed summation order of the Monte-Carlo means (see the header): 256 interleaved partials, then a binary fold
struct McSum {
  double p[256];
  McSum() { for (int t = 0; t < 256; ++t) p[t] = 0.0; }
  void add(int q, double v) { if (g_mode & 2) p[0] += v; else p[q & 255] += v; }
  double total() {
    if (g_mode & 2) return p[0];
    for (int s = 128; s >= 1; s >>= 1)
      for (int t = 0; t < s; ++t) p[t] += p[t + s];
    return p[0];
  }
};
double interp2_linear(const double *Xo, int nx, const double *Yo, int ny, const double *V, double xq, double yq) {
  // V is ny x nx column-major.  NaN outside the grid, as interp2 does.
  if (!(xq >= Xo[0] && xq <= Xo[nx - 1] && yq >= Yo[0] && yq <= Yo[ny - 1])) return NaN;
  int ix = (int)floor((xq - Xo[0]) / (Xo[1] - Xo[0]));
  int iy = (int)floor((yq - Yo[0]) / (Yo[1] - Yo[0]));
  if (ix > nx - 2) ix = nx - 2;
  if (iy > ny - 2) iy = ny - 2;
  if (ix < 0) ix = 0;
  if (iy < 0) iy = 0;
  while (ix > 0 && xq < Xo[ix]) --ix;
  while (ix < nx - 2 && xq > Xo[ix + 1]) ++ix;
  while (iy > 0 && yq < Yo[iy]) --iy;
  while (iy < ny - 2 && yq > Yo[iy + 1]) ++iy;
  double tx = (xq - Xo[ix]) / (Xo[ix + 1] - Xo[ix]);
  double ty = (yq - Yo[iy]) / (Yo[iy + 1] - Yo[iy]);
  double v00 = V[iy + (size_t)ny * ix], v01 = V[iy + (size_t)ny * (ix + 1)];
  double v10 = V[iy + 1 + (size_t)ny * ix], v11 = V[iy + 1 + (size_t)ny * (ix + 1)];
  double a = v00 * (1 - tx) + v01 * tx;
  double b = v10 * (1 - tx) + v11 * tx;
  return a * (1 - ty) + b * ty;
}

struct FloeRef {
  OrFloes *f;
  int i;
};

// returns: 0 updated, 1 floe removed (tmp = []), -3 unsupported (Monte-Carlo regeneration needs rand)
int calc_trajectory(OrFloes *F, int i, const OrParams *P, const double *rows, int nrows) {
  const double dt = P->dt;
  double ext_fx = F->collision_force[2 * i], ext_fy = F->collision_force[2 * i + 1]; // :3
  double ext_torque = F->collision_torque[i];                                         // :4
  const double HFo = P->HFo;                                                          // :5
  const int Nz = F->n_stress;
  double *SH = F->StressH + (size_t)i * Nz * 4;
  // :9-29 stress
  double S[4] = {0, 0, 0, 0}; // column-major 2x2: [xx, yx, xy, yy]
  if (nrows > 0) {
    double r1 = F->Xi[i], r2 = F->Yi[i];
    double s11 = 0, s12 = 0, s21 = 0, s22 = 0, t11 = 0, t12 = 0, t21 = 0, t22 = 0;
    for (int k = 0; k < nrows; ++k) {
      const double *a = rows + 7 * k;
      s11 += (a[3] - r1) * a[1]; s12 += (a[4] - r2) * a[1];
      s21 += (a[3] - r1) * a[2]; s22 += (a[4] - r2) * a[2];
      t11 += a[1] * (a[3] - r1); t12 += a[2] * (a[3] - r1);
      t21 += a[1] * (a[4] - r2); t22 += a[2] * (a[4] - r2);
    }
    double c = 1 / (2 * F->area[i] * F->h[i]);
    S[0] = c * (s11 + t11); S[1] = c * (s21 + t21); S[2] = c * (s12 + t12); S[3] = c * (s22 + t22);
  }
  if (F->StressCount[i] > Nz) F->StressCount[i] = 1;
  for (int q = 0; q < 4; ++q) SH[(size_t)(F->StressCount[i] - 1) * 4 + q] = S[q];
  F->StressCount[i] += 1;
  for (int q = 0; q < 4; ++q) {                               // :20 mean(StressH,3)
    double s = 0;
    for (int z = 0; z < Nz; ++z) s += SH[(size_t)z * 4 + q];
    F->Stress[4 * i + q] = s / Nz;
  }
  // :36-41
  if (F->h[i] > 10) F->h[i] = 10;
  else if (F->mass[i] < 100) { F->mass[i] = 1e3; F->alive[i] = 0; }
  // :42-46
  while (std::max(fabs(ext_fx), fabs(ext_fy)) > F->mass[i] / (5 * dt)) {
    ext_fx = ext_fx / 10; ext_fy = ext_fy / 10; ext_torque = ext_torque / 10;
  }
  const double *Xo = P->Xo, *Yo = P->Yo;
  const int nxo = P->nxo, nyo = P->nyo;
  const double rho0 = 1027, Cd = 3e-3, rho_air = 1.2, Cd_atm = 1e-3; // :58-64
  const double fc = P->fCoriolis;
  const double floe_area = F->area[i];
  double floe_mass = F->mass[i];
  double h = F->h[i];
  double floe_inertia = F->inertia_moment[i];
  // :76-80
  double dh = HFo * dt / h;
  floe_mass = (h - dh) / h * floe_mass; F->mass[i] = floe_mass;
  floe_inertia = (h - dh) / h * floe_inertia; F->inertia_moment[i] = floe_inertia;
  F->h[i] = h - dh;
  if (F->Xi[i] != F->Xi[i]) return 1;                          // :89
  const int v0 = F->vert_off[i], nvr = F->vert_off[i + 1] - v0;
  const int nmc = F->n_mc;
  const bool doforce = P->doInt || F->h[i] < 0.1;              // :94,121 (FxOA is never empty here)
  Vec xr1, xr2;
  if (doforce) {
    double sa, ca;
    t_sincos(F->alpha_i[i], &sa, &ca);                        // :97
    const double *mx = F->mcX + (size_t)i * nmc, *my = F->mcY + (size_t)i * nmc;
    xr1.resize(nmc); xr2.resize(nmc);
    for (int q = 0; q < nmc; ++q) { xr1[q] = ca * mx[q] + (-sa) * my[q]; xr2[q] = sa * mx[q] + ca * my[q]; } // :98
    int cnt = 0;
    for (int q = 0; q < nmc; ++q) cnt += F->mcA[(size_t)i * nmc + q] ? 1 : 0;
    if (cnt == 0) return -3;                                    // :100-110 needs rand()
  }
  // :116 out of the ocean grid
  {
    double mxx = F->cax[v0], mnx = F->cax[v0], mxy = F->cay[v0], mny = F->cay[v0];
    for (int k = 1; k < nvr; ++k) {
      mxx = std::max(mxx, F->cax[v0 + k]); mnx = std::min(mnx, F->cax[v0 + k]);
      mxy = std::max(mxy, F->cay[v0 + k]); mny = std::min(mny, F->cay[v0 + k]);
    }
    double Xomax = Xo[0], Xomin = Xo[0], Yomax = Yo[0], Yomin = Yo[0];
    for (int k = 1; k < nxo; ++k) { Xomax = std::max(Xomax, Xo[k]); Xomin = std::min(Xomin, Xo[k]); }
    for (int k = 1; k < nyo; ++k) { Yomax = std::max(Yomax, Yo[k]); Yomin = std::min(Yomin, Yo[k]); }
    if (mxx + F->Xi[i] > Xomax || mnx + F->Xi[i] < Xomin || mxy + F->Yi[i] > Yomax || mny + F->Yi[i] < Yomin)
      return 1;
  }
  if (F->alive[i] != 1) return 0;                              // :118
  if (doforce) {                                               // :121-166
    const double Xi = F->Xi[i], Yi = F->Yi[i];
    const double ta = P->turn_angle;
    double sta, cta;
    t_sincos(ta, &sta, &cta);
    const uint8_t *A = F->mcA + (size_t)i * nmc;
    // U10, V10: mean of the interpolated winds over the points inside the floe (:139)
    McSum su, sv; int cnt = 0;
    Vec Uo(nmc), Vo(nmc), th(nmc), rho(nmc);
    for (int q = 0; q < nmc; ++q) {
      double Xg = xr1[q] + Xi, Yg = xr2[q] + Yi;                // :122
      th[q] = t_atan2(xr2[q], xr1[q]); rho[q] = hyp(xr1[q], xr2[q]); // :124
      Uo[q] = interp2_linear(Xo, nxo, Yo, nyo, P->Uocn, Xg, Yg);
      Vo[q] = interp2_linear(Xo, nxo, Yo, nyo, P->Vocn, Xg, Yg);
      if (A[q]) {
        su.add(q, interp2_linear(Xo, nxo, Yo, nyo, P->Uwind, Xg, Yg));
        sv.add(q, interp2_linear(Xo, nxo, Yo, nyo, P->Vwind, Xg, Yg));
        ++cnt;
      }
    }
    double U10 = su.total() / cnt, V10 = sv.total() / cnt;
    double w10 = sqrt(U10 * U10 + V10 * V10);
    double Fx_atm = rho_air * Cd_atm * w10 * U10;               // :140-141
    double Fy_atm = rho_air * Cd_atm * w10 * V10;
    McSum sFx, sFy, sT;
    for (int q = 0; q < nmc; ++q) {
      double sth, cth;
      t_sincos(th[q], &sth, &cth);
      double Uice = F->Ui[i] - rho[q] * F->ksi_ice[i] * sth;    // :127-128
      double Vice = F->Vi[i] + rho[q] * F->ksi_ice[i] * cth;
      double Fx_p = -(floe_mass / floe_area) * fc * Vo[q];      // :143-144
      double Fy_p = +(floe_mass / floe_area) * fc * Uo[q];
      double du = Uo[q] - Uice, dv = Vo[q] - Vice;              // :146
      double sp = sqrt(du * du + dv * dv);
      double tauX = rho0 * Cd * sp * (cta * du - sta * dv);     // :148-149
      double tauY = rho0 * Cd * sp * (sta * du + cta * dv);
      double Fx = tauX + Fx_atm + Fx_p;                         // :151-152
      double Fy = tauY + Fy_atm + Fy_p;
      double tq = (-Fx * sth + Fy * cth) * rho[q];              // :156
      Fx = Fx + (floe_mass / floe_area) * fc * F->Vi[i];        // :159-160
      Fy = Fy - (floe_mass / floe_area) * fc * F->Ui[i];
      if (A[q]) { sFx.add(q, Fx); sFy.add(q, Fy); sT.add(q, tq); }
    }
    F->FxOA[i] = sFx.total() / cnt; F->FyOA[i] = sFy.total() / cnt; F->torqueOA[i] = sT.total() / cnt; // :163-165
  }
  // :174-177 positions, Adams-Bashforth 2
  double dx = 1.5 * dt * F->Ui[i] - 0.5 * dt * F->dXi_p[i];
  double dy = 1.5 * dt * F->Vi[i] - 0.5 * dt * F->dYi_p[i];
  F->Xi[i] = F->Xi[i] + dx; F->dXi_p[i] = F->Ui[i];
  F->Yi[i] = F->Yi[i] + dy; F->dYi_p[i] = F->Vi[i];
  F->alpha_i[i] = F->alpha_i[i] + 1.5 * dt * F->ksi_ice[i] - 0.5 * dt * F->dalpha_i_p[i];
  F->dalpha_i_p[i] = F->ksi_ice[i];
  // :181-208 velocities
  const double hh = F->h[i];
  double numU = F->FxOA[i] * floe_area + ext_fx, numV = F->FyOA[i] * floe_area + ext_fy;
  double dUi_dt = numU / floe_mass, dVi_dt = numV / floe_mass;
  bool have_frac = false;
  double frac = 0;
  bool bigU = fabs(dt * dUi_dt) > 0.5 * hh, bigV = fabs(dt * dVi_dt) > 0.5 * hh;
  bool smallU = fabs(dt * dUi_dt) < 0.5 * hh, smallV = fabs(dt * dVi_dt) < 0.5 * hh;
  auto sgn = [](double v) { return (v > 0) ? 1.0 : (v < 0 ? -1.0 : 0.0); };
  if (bigU && bigV) {
    double a = sgn(dUi_dt) * 0.5 * hh / dt, b = sgn(dVi_dt) * 0.5 * hh / dt;
    double frac1 = a / numU * floe_mass, frac2 = b / numV * floe_mass;
    frac = std::min(frac1, frac2); have_frac = true;
  } else if (bigU && smallV) {
    double a = sgn(dUi_dt) * 0.5 * hh / dt;
    frac = a / numU * floe_mass; have_frac = true;
  } else if (smallU && bigV) {
    double b = sgn(dVi_dt) * 0.5 * hh / dt;
    frac = b / numV * floe_mass; have_frac = true;
  }
  if (have_frac) { dUi_dt = frac * (numU / floe_mass); dVi_dt = frac * (numV / floe_mass); }
  F->Ui[i] = F->Ui[i] + 1.5 * dt * dUi_dt - 0.5 * dt * F->dUi_p[i];
  F->Vi[i] = F->Vi[i] + 1.5 * dt * dVi_dt - 0.5 * dt * F->dVi_p[i];
  F->dUi_p[i] = dUi_dt; F->dVi_p[i] = dVi_dt;
  // :210-219 angular velocity
  double dksi = (F->torqueOA[i] * floe_area + ext_torque) / floe_inertia;
  if (have_frac) dksi = frac * dksi;
  double ksi = F->ksi_ice[i] + 1.5 * dt * dksi - 0.5 * dt * F->dksi_ice_p[i];
  if (fabs(ksi) > 1e-5) ksi = sgn(ksi) * 1e-5;
  F->ksi_ice[i] = ksi;
  F->dksi_ice_p[i] = dksi;
  // :221-222 rotate the ring
  double sa, ca;
  t_sincos(F->alpha_i[i], &sa, &ca);
  for (int k = 0; k < nvr; ++k) {
    F->cax[v0 + k] = ca * F->c0x[v0 + k] + (-sa) * F->c0y[v0 + k];
    F->cay[v0 + k] = sa * F->c0x[v0 + k] + ca * F->c0y[v0 + k];
  }
  if (P->doInt) {                                              // :224-234 strain
    Vec Uice(nvr), Vice(nvr);
    for (int k = 0; k < nvr; ++k) {
      double th = t_atan2(F->cay[v0 + k], F->cax[v0 + k]), rh = hyp(F->cax[v0 + k], F->cay[v0 + k]);
      double sth, cth;
      t_sincos(th, &sth, &cth);
      Uice[k] = F->Ui[i] - rh * F->ksi_ice[i] * sth;
      Vice[k] = F->Vi[i] + rh * F->ksi_ice[i] * cth;
    }
    double sudy = 0, sudx = 0, svdy = 0, svdx = 0;
    for (int k = 0; k < nvr; ++k) {
      int k2 = (k + 1 == nvr) ? 0 : k + 1;
      double dU = Uice[k2] - Uice[k], dV = Vice[k2] - Vice[k];
      double dcy = F->cay[v0 + k2] - F->cay[v0 + k], dcx = F->cax[v0 + k2] - F->cax[v0 + k];
      sudy += dU * dcy; sudx += dU * dcx; svdy += dV * dcy; svdx += dV * dcx;
    }
    double du_dx = 0.5 * sudy / floe_area, du_dy = 0.5 * sudx / floe_area;
    double dv_dx = 0.5 * svdy / floe_area, dv_dy = 0.5 * svdx / floe_area;
    // strain = 1/2*([du_dx du_dy; dv_dx dv_dy] + [du_dx dv_dx; du_dy dv_dy]), column-major
    F->strain[4 * i + 0] = 0.5 * (du_dx + du_dx);
    F->strain[4 * i + 1] = 0.5 * (dv_dx + du_dy);
    F->strain[4 * i + 2] = 0.5 * (du_dy + dv_dx);
    F->strain[4 * i + 3] = 0.5 * (dv_dy + dv_dy);
  }
  return 0;
}

} // namespace

// =================================================================================================
extern "C" {

int or_set_mode(int mode) { int old = g_mode; g_mode = mode; return old; }

int or_set_clipper(const char *path, int kind) {
  void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) return -2;
  g_ref_clip = nullptr; g_host_clip = nullptr;
  if (kind == 0) g_ref_clip = (ref_clip_fn)dlsym(h, "ref_clip");
  else g_host_clip = (host_clip_fn)dlsym(h, "host_clip_batch");
  return (g_ref_clip || g_host_clip) ? 0 : -2;
}

double or_ring_area(const double *x, const double *y, int n) { return ring_area(x, y, n); }
void or_ring_centroid(const double *x, const double *y, int n, double *cx, double *cy) { ring_centroid(x, y, n, *cx, *cy); }
double or_polyarea(const double *x, const double *y, int n) { return polyarea(x, y, n); }
int or_interx(const double *x1, const double *y1, int n1, const double *x2, const double *y2, int n2,
              double *px, double *py, int cap) {
  Vec a, b;
  interx(x1, y1, n1, x2, y2, n2, a, b);
  int n = (int)a.size();
  for (int k = 0; k < n && k < cap; ++k) { px[k] = a[k]; py[k] = b[k]; }
  return n;
}
void or_p_poly_dist(const double *xp, const double *yp, int np, const double *xv, const double *yv, int nv, double *dmin) {
  Vec a(xp, xp + np), b(yp, yp + np), c(xv, xv + nv), d(yv, yv + nv), out;
  p_poly_dist(a, b, c, d, out);
  for (int k = 0; k < np; ++k) dmin[k] = out[k];
}
void or_inpolygon(const double *x, const double *y, int np, const double *xv, const double *yv, int nv, uint8_t *in) {
  Vec c(xv, xv + nv), d(yv, yv + nv);
  for (int k = 0; k < np; ++k) in[k] = inpolygon1(x[k], y[k], c, d) ? 1 : 0;
}
int or_floe_interactions(const double *c1x, const double *c1y, int n1, const double *s1, const double *c2x,
                         const double *c2y, int n2, const double *s2, int floe2_is_boundary,
                         const double *c2_boundary, int periodic, double Modulus, double dt, double mu,
                         double nu, double *force, double *pcontact, double *overlap, int cap) {
  if (!g_ref_clip && !g_host_clip) return -2;
  Body b1 = {s1[0], s1[1], s1[2], s1[3], s1[4], s1[5], s1[6]}, b2 = {s2[0], s2[1], s2[2], s2[3], s2[4], s2[5], s2[6]};
  std::vector<Contact> rows;
  Vec a(c1x, c1x + n1), b(c1y, c1y + n1), c(c2x, c2x + n2), d(c2y, c2y + n2);
  if (!floe_interactions(a, b, b1, c, d, b2, floe2_is_boundary != 0, c2_boundary, periodic != 0, Modulus, dt, mu, nu, rows)) return -4;
  int n = (int)rows.size();
  for (int k = 0; k < n && k < cap; ++k) {
    force[2 * k] = rows[k].fx; force[2 * k + 1] = rows[k].fy;
    pcontact[2 * k] = rows[k].px; pcontact[2 * k + 1] = rows[k].py;
    overlap[k] = rows[k].overlap;
  }
  return n;
}
void or_sincos(double x, double *s, double *c) { orm_sincos(x, s, c); }
double or_atan2(double y, double x) { return orm_atan2(y, x); }
int64_t or_matlab_int64_scaled(double v) { return to_fixed(v); }

// floe_interactions_all.m:9-285
int or_step(OrFloes *F, const OrParams *P, int32_t *inter_off, double *inter, int64_t inter_cap,
            int32_t *pair_i, int32_t *pair_j, int64_t pair_cap, int64_t *npairs_out, int32_t *n_ext,
            int32_t threads) {
  if (!g_ref_clip && !g_host_clip) return -2;
  const int N0 = F->n, Nb = F->nb;
  const double *c2b = P->c2_boundary;
  double Lx = c2b[0], Ly = c2b[1];                              // :9-10
  for (int k = 1; k < 5; ++k) { Lx = std::max(Lx, c2b[2 * k]); Ly = std::max(Ly, c2b[2 * k + 1]); }
  const bool PERIODIC = P->PERIODIC != 0, COLLISION = P->COLLISION != 0;

  // extended list: real floes then ghosts.  Ghosts copy their source's state at creation (:33-37,:54-58)
  struct Ext { int src; double x, y; int floenum; }; // src = real floe whose ring/state it carries
  std::vector<Ext> ext(N0);
  for (int i = 0; i < N0; ++i) { ext[i].src = i; ext[i].x = F->Xi[i]; ext[i].y = F->Yi[i]; ext[i].floenum = i + 1; }
  std::vector<int> parent; // 0-based index into ext of each ghost's parent (:36,:57)
  auto max_abs_coord = [&](int e, int dim) {
    int s = ext[e].src, v0 = F->vert_off[s], nv = F->vert_off[s + 1] - v0;
    double m = 0;
    for (int k = 0; k < nv; ++k) {
      double v = (dim == 0) ? F->cax[v0 + k] + ext[e].x : F->cay[v0 + k] + ext[e].y;
      m = std::max(m, fabs(v));
    }
    return m;
  };
  auto sgn = [](double v) { return (v > 0) ? 1.0 : (v < 0 ? -1.0 : 0.0); };
  if (PERIODIC) {
    int n1 = (int)ext.size();
    for (int i = 0; i < n1; ++i) {                              // :29-42
      if (F->alive[ext[i].src] && max_abs_coord(i, 0) > Lx) {
        Ext g = ext[i];
        g.x = ext[i].x - 2 * Lx * sgn(ext[i].x);
        g.floenum = -abs(ext[i].floenum);
        ext.push_back(g); parent.push_back(i);
      }
    }
    int n2 = (int)ext.size();
    for (int i = 0; i < n2; ++i) {                              // :50-62
      if (F->alive[ext[i].src] && max_abs_coord(i, 1) > Ly) {
        Ext g = ext[i];
        g.y = ext[i].y - 2 * Ly * sgn(ext[i].y);
        g.floenum = -abs(ext[i].floenum);
        ext.push_back(g); parent.push_back(i);
      }
    }
  }
  const int N = (int)ext.size();
  *n_ext = N;
  auto alive = [&](int e) { return F->alive[ext[e].src] != 0; };
  auto rmax = [&](int e) { return F->rmax[ext[e].src]; };

  // ---- :76-120 broad-phase ----
  std::vector<std::vector<int> > pot(N); // accepted partner indices (1-based), ascending
  for (int i = Nb; i < N; ++i) {
    std::vector<double> mems;
    if (ext[i].floenum < 0) {                                  // :94-99
      int num = abs(ext[i].floenum) - 1;
      for (size_t q = 0; q < pot[num].size(); ++q) mems.push_back((double)pot[num][q]);
    }
    if (alive(i) && !(ext[i].x != ext[i].x) && COLLISION) {   // :101
      for (int j = i + 1; j < N; ++j) {                        // :102-103 (j > i)
        if (!alive(j)) continue;
        double ddx = ext[i].x - ext[j].x, ddy = ext[i].y - ext[j].y;
        if (!(sqrt(ddx * ddx + ddy * ddy) < (rmax(i) + rmax(j)))) continue;
        double key = (double)abs(ext[j].floenum);
        bool member = std::find(mems.begin(), mems.end(), key) != mems.end();
        if (!(!member || 2 * (rmax(i) + rmax(j)) > std::min(2 * Lx, 2 * Ly))) continue;
        pot[i].push_back(j + 1);                               // :104
        mems.push_back((double)ext[j].floenum);                // :113
      }
    }
  }
  int64_t np = 0;
  for (int i = 0; i < N; ++i)
    for (size_t q = 0; q < pot[i].size(); ++q) {
      if (np < pair_cap) { pair_i[np] = i + 1; pair_j[np] = pot[i][q]; }
      ++np;
    }
  *npairs_out = np;
  if (np > pair_cap) return -1;

  // ---- :123-174 narrow-phase ----
  typedef std::vector<double> Rows; // 7 doubles per row
  std::vector<Rows> inter_rows(N);
  std::vector<double> overlapArea(N, 0.0);
  std::vector<int> kill(N, 0), transfer(N, 0);
  std::vector<int> alive_new(N0);
  for (int i = 0; i < N0; ++i) alive_new[i] = F->alive[i];
  int err = 0;
  auto ring_of = [&](int e, Vec &x, Vec &y) {
    int s = ext[e].src, v0 = F->vert_off[s], nv = F->vert_off[s + 1] - v0;
    x.resize(nv); y.resize(nv);
    for (int k = 0; k < nv; ++k) { x[k] = F->cax[v0 + k] + ext[e].x; y[k] = F->cay[v0 + k] + ext[e].y; } // :105 / floe_interactions.m:25
  };
  auto body_of = [&](int e) {
    int s = ext[e].src;
    Body b = {ext[e].x, ext[e].y, F->Ui[s], F->Vi[s], F->ksi_ice[s], F->h[s], F->area[s]};
    return b;
  };
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads > 0 ? threads : 1)
  for (int i = Nb; i < N; ++i) {
    Vec c1x, c1y, c2x, c2y;
    std::vector<Contact> rows;
    ring_of(i, c1x, c1y);
    Body b1 = body_of(i);
    for (size_t q = 0; q < pot[i].size(); ++q) {               // :129-147
      int j = pot[i][q] - 1;
      ring_of(j, c2x, c2y);
      Body b2 = body_of(j);
      if (!floe_interactions(c1x, c1y, b1, c2x, c2y, b2, false, c2b, PERIODIC, P->Modulus, P->dt, P->mu, P->nu, rows)) {
#pragma omp atomic write
        err = -4;
        continue;
      }
      double sabs = 0;
      for (size_t r = 0; r < rows.size(); ++r) sabs += fabs(rows[r].fx) + fabs(rows[r].fy);
      if (sabs != 0) {                                          // :135-137
        double so = 0;
        for (size_t r = 0; r < rows.size(); ++r) {
          double row[7] = {(double)(j + 1), rows[r].fx, rows[r].fy, rows[r].px, rows[r].py, 0, rows[r].overlap};
          inter_rows[i].insert(inter_rows[i].end(), row, row + 7);
          so += rows[r].overlap;
        }
        overlapArea[i] = so + overlapArea[i];
      } else if (!rows.empty() && std::isinf(rows[0].overlap) && i + 1 > Nb) {   // :138-145
        // kill/transfer start as 1 x N0 (:123) and grow when a ghost row (i > N0) assigns into them
        if (i + 1 <= N0 && rows[0].overlap > 0) { kill[i] = i + 1; transfer[i] = j + 1; }
        else if (j + 1 <= N0) kill[i] = j + 1;
      }
    }
    if (!PERIODIC) {                                            // :150-172
      Vec bx(4), by(4);
      {
        const int start = (g_mode >> 8) & 3, rev = (g_mode >> 10) & 1;
        for (int k = 0; k < 4; ++k) {
          const int q = (start + (rev ? 4 - k : k)) & 3;
          bx[k] = c2b[2 * q]; by[k] = c2b[2 * q + 1];
        }
      }
      Body fb = {P->fb_Xi, P->fb_Yi, P->fb_Ui, P->fb_Vi, P->fb_ksi_ice, P->fb_h, P->fb_area};
      if (!floe_interactions(c1x, c1y, b1, bx, by, fb, true, c2b, PERIODIC, P->Modulus, P->dt, P->mu, P->nu, rows)) {
#pragma omp atomic write
        err = -4;
        continue;
      }
      Vec b5x(5), b5y(5);
      for (int k = 0; k < 5; ++k) { b5x[k] = c2b[2 * k]; b5y[k] = c2b[2 * k + 1]; }
      if (!inpolygon1(ext[i].x, ext[i].y, b5x, b5y)) alive_new[i] = 0;     // :152-155 (non-periodic: no ghosts)
      double sabs = 0;
      for (size_t r = 0; r < rows.size(); ++r) sabs += fabs(rows[r].fx) + fabs(rows[r].fy);
      if (sabs != 0) {
        double so = 0;
        for (size_t r = 0; r < rows.size(); ++r) {
          double fx = rows[r].fx, fy = rows[r].fy;
          if (fabs(rows[r].py) == Ly) fx = 0;                   // :160-162
          if (fabs(rows[r].px) == Lx) fy = 0;                   // :163-165
          double row[7] = {INF, fx, fy, rows[r].px, rows[r].py, 0, rows[r].overlap};
          inter_rows[i].insert(inter_rows[i].end(), row, row + 7);
          so += rows[r].overlap;
        }
        overlapArea[i] = so + overlapArea[i];
      }
    }
  }
  if (err) return err;
  for (int i = 0; i < N0; ++i) F->alive[i] = alive_new[i];
  for (int i = 0; i < N; ++i)                                   // :175-179
    if (abs(kill[i] - (i + 1)) > 0 && kill[i] > 0) transfer[kill[i] - 1] = i + 1;

  // ---- :187-214 mirror, sequential ----
  for (int i = 0; i < N; ++i) {
    size_t nrow = inter_rows[i].size() / 7; // rows present when floe i is visited (own + mirrored from lower i)
    for (size_t r = 0; r < nrow; ++r) {
      double a[7];
      memcpy(a, &inter_rows[i][7 * r], sizeof(a));
      if (a[0] <= N && a[0] > i + 1) {
        int j = (int)a[0] - 1;
        double row[7] = {(double)(i + 1), -a[1], -a[2], a[3], a[4], 0, a[6]};
        inter_rows[j].insert(inter_rows[j].end(), row, row + 7);
        overlapArea[j] = overlapArea[j] + a[6];
      }
    }
  }

  // ---- :218-246 torques and sums of ghosts, then added to parents ----
  std::vector<double> cfx(N, 0.0), cfy(N, 0.0), ctq(N, 0.0);
  auto torque_and_sums = [&](int i, double &sfx, double &sfy, double &stq) {
    Rows &a = inter_rows[i];
    size_t nrow = a.size() / 7;
    sfx = sfy = stq = 0;
    for (size_t r = 0; r < nrow; ++r) {
      double *w = &a[7 * r];
      // cross([P - r, 0], [F, 0]) (3) = (Px - x) * Fy - (Py - y) * Fx   (:230,:258)
      w[5] = (w[3] - ext[i].x) * w[2] - (w[4] - ext[i].y) * w[1];
    }
    for (size_t r = 0; r < nrow; ++r) { sfx += a[7 * r + 1]; sfy += a[7 * r + 2]; stq += a[7 * r + 5]; }
  };
  if (PERIODIC) {
    for (int i = N0; i < N; ++i) {
      if (!inter_rows[i].empty()) { double a, b, c; torque_and_sums(i, a, b, c); cfx[i] = a; cfy[i] = b; ctq[i] = c; }
    }
    for (size_t g = 0; g < parent.size(); ++g) {                // :242-245
      int p = parent[g], gi = N0 + (int)g;
      cfx[p] = cfx[p] + cfx[gi]; cfy[p] = cfy[p] + cfy[gi]; ctq[p] = ctq[p] + ctq[gi];
    }
  }

  // ---- :249-285 real floes: torque, sums, wrap, integrate ----
  int rc_all = 0;
  // (a parfor in the reference, floe_interactions_all.m:249: every floe touches only its own fields)
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads > 0 ? threads : 1)
  for (int i = Nb; i < N0; ++i) {
    if (!inter_rows[i].empty()) {
      double a, b, c;
      torque_and_sums(i, a, b, c);
      cfx[i] = a + cfx[i]; cfy[i] = b + cfy[i]; ctq[i] = c + ctq[i];   // :262-263
    }
    F->collision_force[2 * i] = cfx[i]; F->collision_force[2 * i + 1] = cfy[i];
    F->collision_torque[i] = ctq[i];
    F->OverlapArea[i] = overlapArea[i];
    if (PERIODIC) {                                             // :267-277
      if (fabs(F->Xi[i]) > Lx) F->Xi[i] = F->Xi[i] - 2 * Lx * sgn(F->Xi[i]);
      if (fabs(F->Yi[i]) > Ly) F->Yi[i] = F->Yi[i] - 2 * Ly * sgn(F->Yi[i]);
    }
    if (F->alive[i]) {                                          // :280-283
      double x_before = ext[i].x;
      // tmp = calc_trajectory(...): when tmp is [] the stored Floe(i) keeps its pre-call fields (:282)
      const int Nz = F->n_stress;
      int sc0 = F->StressCount[i] > Nz ? 1 : F->StressCount[i];
      double slot0[4], st0[4];
      for (int q = 0; q < 4; ++q) { slot0[q] = F->StressH[((size_t)i * Nz + (sc0 - 1)) * 4 + q]; st0[q] = F->Stress[4 * i + q]; }
      int cnt0 = F->StressCount[i], alive0 = F->alive[i];
      double h0 = F->h[i], m0 = F->mass[i], in0 = F->inertia_moment[i];
      int rc = calc_trajectory(F, i, P, inter_rows[i].empty() ? nullptr : &inter_rows[i][0], (int)(inter_rows[i].size() / 7));
      if (rc < 0) {
#pragma omp atomic write
        rc_all = rc;
      }
      if (rc == 1) {
        for (int q = 0; q < 4; ++q) { F->StressH[((size_t)i * Nz + (sc0 - 1)) * 4 + q] = slot0[q]; F->Stress[4 * i + q] = st0[q]; }
        F->StressCount[i] = cnt0; F->alive[i] = alive0; F->h[i] = h0; F->mass[i] = m0; F->inertia_moment[i] = in0;
      }
      if (rc == 1 || x_before != x_before) kill[i] = i + 1;
    }
  }
  for (int i = 0; i < Nb; ++i) { F->collision_force[2 * i] = 0; F->collision_force[2 * i + 1] = 0; F->collision_torque[i] = 0; F->OverlapArea[i] = 0; }
  for (int i = 0; i < N; ++i) { F->kill[i] = kill[i]; F->transfer[i] = transfer[i]; } // caller sizes both 4*n

  // contact rows of the real floes, CSR
  int64_t tot = 0;
  inter_off[0] = 0;
  for (int i = 0; i < N0; ++i) {
    int64_t nrow = (int64_t)(inter_rows[i].size() / 7);
    if (tot + nrow > inter_cap) return -1;
    if (nrow) memcpy(inter + 7 * tot, &inter_rows[i][0], sizeof(double) * 7 * nrow);
    tot += nrow;
    inter_off[i + 1] = (int32_t)tot;
  }
  return rc_all;
}

} // extern "C"

// =================================================================================================
// calc_eulerian_data.m:1-192.  area(intersect(box, poly)) is a MathWorks polyshape operation; by its
// definition: the ring cut down to the rectangle, half-plane by half-plane (x >= xl, x <= xh, y >= yl, y <= yh), and
// the |shoelace| of the result relative to its first vertex.  Written here with explicit vertex lists per stage (the
// device streams the same stages without buffers); tests pin it against the reference Clipper's 'int' of the same
// ring and rectangle.  Nb > 0 (boundaries subtracted from the box, :10-24,:139-141) is not restated: returns -3.
namespace {
void clip_half_plane(const Vec &xi, const Vec &yi, int k, double bound, Vec &xo, Vec &yo) {
  xo.clear(); yo.clear();
  const size_t n = xi.size();
  if (n == 0) return;
  auto inside = [&](double x, double y) { return k == 0 ? x >= bound : k == 1 ? x <= bound : k == 2 ? y >= bound : y <= bound; };
  // edges in the order v0->v1, v1->v2, ..., v(n-1)->v0: the first vertex is emitted last
  for (size_t q = 0; q < n; ++q) {
    const double ax = xi[q], ay = yi[q], bx = xi[(q + 1) % n], by = yi[(q + 1) % n];
    const bool ia = inside(ax, ay), ib = inside(bx, by);
    if (ia != ib) {
      double x, y;
      if (k < 2) { double t = (bound - ax) / (bx - ax); x = bound; y = ay + t * (by - ay); }
      else { double t = (bound - ay) / (by - ay); y = bound; x = ax + t * (bx - ax); }
      xo.push_back(x); yo.push_back(y);
    }
    if (ib) { xo.push_back(bx); yo.push_back(by); }
  }
}
double ring_box_area(const Vec &x, const Vec &y, double xl, double xh, double yl, double yh) {
  Vec ax = x, ay = y, bx, by;
  const double bounds[4] = {xl, xh, yl, yh};
  for (int k = 0; k < 4; ++k) { clip_half_plane(ax, ay, k, bounds[k], bx, by); ax.swap(bx); ay.swap(by); }
  const int n = (int)ax.size();
  if (n == 0) return 0.0;
  double s = 0;
  for (int i = 0; i < n; ++i) {   // every vertex after the first against its predecessor, then the closing edge
    const int p = i, c = (i + 1) % n;
    const double px = ax[p] - ax[0], py = ay[p] - ay[0], cx = ax[c] - ax[0], cy = ay[c] - ay[0];
    s += px * cy - cx * py;
  }
  return fabs(0.5 * s);
}
} // namespace

extern "C" double or_ring_box_area(const double *x, const double *y, int n, double xl, double xh, double yl, double yh) {
  return ring_box_area(Vec(x, x + n), Vec(y, y + n), xl, xh, yl, yh);
}

// out: 18 fields of Ny x Nx (MATLAB column-major: element (jj, ii) at jj + Ny * ii), in the order
// u v du dv stress stressxx stressyx stressxy stressyy strainux strainvx strainuy strainvy c Over Mtot area h
extern "C" int or_eulerian(const OrFloes *F, int Nx, int Ny, int Nb, const double *c2b, int PERIODIC, double *out) {
  if (Nb > 0 && !g_ref_clip) return -3;      // the topography path clips operands of several contours: reference build only
  const int N0 = F->n;
  double Lx = c2b[0], Ly = c2b[1], xmin = c2b[0], ymin = c2b[1];
  for (int k = 1; k < 5; ++k) { Lx = std::max(Lx, c2b[2 * k]); Ly = std::max(Ly, c2b[2 * k + 1]); xmin = std::min(xmin, c2b[2 * k]); ymin = std::min(ymin, c2b[2 * k + 1]); }
  struct Ent { int src; double x, y; };
  std::vector<Ent> E;
  // :7-8 live floes; :10-24 the first Nb of them are the topography: their union is cut out of the cells, they do not
  // count towards the averages
  std::vector<Ring> topo;
  {
    int seen = 0;
    for (int i = 0; i < N0; ++i) {
      if (!F->alive[i]) continue;
      if (seen < Nb) {
        ++seen;
        int v0 = F->vert_off[i], nv = F->vert_off[i + 1] - v0;
        Ring r;
        for (int k = 0; k < nv; ++k) { r.x.push_back(F->cax[v0 + k] + F->Xi[i]); r.y.push_back(F->cay[v0 + k] + F->Yi[i]); }
        if (ring_area_signed(r.x.data(), r.y.data(), nv) < 0) { std::reverse(r.x.begin(), r.x.end()); std::reverse(r.y.begin(), r.y.end()); }
        topo.push_back(r);
        continue;
      }
      Ent e = {i, F->Xi[i], F->Yi[i]}; E.push_back(e);
    }
  }
  auto sgn = [](double v) { return (v > 0) ? 1.0 : (v < 0 ? -1.0 : 0.0); };
  auto extent = [&](const Ent &e, int dim) {
    int v0 = F->vert_off[e.src], nv = F->vert_off[e.src + 1] - v0;
    double m = 0;
    for (int k = 0; k < nv; ++k) m = std::max(m, fabs(dim == 0 ? F->cax[v0 + k] + e.x : F->cay[v0 + k] + e.y));
    return m;
  };
  if (PERIODIC && !E.empty()) {
    const int n1 = (int)E.size();
    for (int i = 0; i < n1; ++i)                                             // :41-50
      if (extent(E[i], 0) > Lx) { Ent g = E[i]; g.x = E[i].x - 2 * Lx * sgn(E[i].x); E.push_back(g); }
    // :58-67 -- as shipped the test reads `poly`, which still holds the LAST floe of the first loop: every live entry
    // (or none) gets a Y image
    const bool all = extent(E[n1 - 1], 1) > Ly;
    const int n2 = (int)E.size();
    if (all) for (int i = 0; i < n2; ++i) { Ent g = E[i]; g.y = E[i].y - 2 * Ly * sgn(E[i].y); E.push_back(g); }
  }
  const int M = (int)E.size();
  Vec xg(Nx + 1), yg(Ny + 1);
  const double stepx = (Lx - xmin) / Nx, stepy = (Ly - ymin) / Ny;
  for (int k = 0; k <= Nx; ++k) xg[k] = xmin + k * stepx;                  // :73 (colon operator: start + k * step)
  for (int k = 0; k <= Ny; ++k) yg[Ny - k] = ymin + k * stepy;             // :74-75 fliplr
  xg[Nx] = Lx; yg[0] = Ly;                                                 // (the colon operator ends on the upper limit)
  const double dx = fabs(xg[1] - xg[0]), dy = fabs(yg[1] - yg[0]);
  const double r_max = sqrt((dx / 2) * (dx / 2) + (dy / 2) * (dy / 2));
  const size_t cells = (size_t)Nx * Ny;
  for (size_t k = 0; k < 18 * cells; ++k) out[k] = 0.0;
  auto nz = [](double v) { return v != v ? 0.0 : v; };
  for (int ii = 0; ii < Nx; ++ii) {
    for (int jj = 0; jj < Ny; ++jj) {
      const double xc = 0.5 * (xg[ii] + xg[ii + 1]), yc = 0.5 * (yg[jj] + yg[jj + 1]);
      std::vector<int> live;
      double Mpre = 0;
      for (int e = 0; e < M; ++e) {                                        // :122-128
        double ddx = xc - E[e].x, ddy = yc - E[e].y;
        double pint = sqrt(ddx * ddx + ddy * ddy) - (F->rmax[E[e].src] + r_max);
        if (pint < 0) { live.push_back(e); Mpre += F->mass[E[e].src]; }
      }
      if (!(Mpre > 0)) continue;                                           // :136-137
      const double xl = xg[ii], xh = xg[ii + 1], yh = yg[jj], yl = yg[jj + 1];
      double bxs[4] = {xl, xl, xh, xh}, bys[4] = {yh, yl, yl, yh};         // :140 bound
      double abox = ring_area(bxs, bys, 4);
      // :141-143 box = subtract(box, boundaries).  Only a cell whose rectangle meets the bounding box of a topography
      // floe can lose anything; such a cell goes through the clipper (the polyshape operations restated by the
      // reference's own fixed-point clip, as polyclip.m uses it), every other cell keeps the plain rectangle.
      std::vector<Ring> near_topo;
      for (size_t t = 0; t < topo.size(); ++t) {
        double x0 = topo[t].x[0], x1 = x0, y0 = topo[t].y[0], y1 = y0;
        for (size_t k = 1; k < topo[t].x.size(); ++k) { x0 = std::min(x0, topo[t].x[k]); x1 = std::max(x1, topo[t].x[k]); y0 = std::min(y0, topo[t].y[k]); y1 = std::max(y1, topo[t].y[k]); }
        if (!(x1 < xl || x0 > xh || y1 < yl || y0 > yh)) near_topo.push_back(topo[t]);
      }
      std::vector<Ring> cut_box;
      if (!near_topo.empty()) {
        Ring b; b.x.assign(bxs, bxs + 4); b.y.assign(bys, bys + 4);
        bool ok = true;
        cut_box = polyclip_paths(std::vector<Ring>(1, b), near_topo, 0, 0, 1, ok);    // 'dif', clip fill non-zero = the union of :16-20
        if (!ok) return -4;
        abox = rings_area(cut_box);
      }
      std::vector<int> fn; Vec Aover;
      for (size_t q = 0; q < live.size(); ++q) {
        const Ent &en = E[live[q]];
        int v0 = F->vert_off[en.src], nv = F->vert_off[en.src + 1] - v0;
        if (nv > 1 && F->cax[v0] == F->cax[v0 + nv - 1] && F->cay[v0] == F->cay[v0 + nv - 1]) --nv;   // polyshape drops the closing copy
        Vec rx(nv), ry(nv);
        for (int k = 0; k < nv; ++k) { rx[k] = F->cax[v0 + k] + en.x; ry[k] = F->cay[v0 + k] + en.y; }
        double a;
        if (near_topo.empty()) a = ring_box_area(rx, ry, xl, xh, yl, yh);
        else if (cut_box.empty()) a = 0;
        else {
          Ring f; f.x = rx; f.y = ry;
          bool ok = true;
          a = rings_area(polyclip_paths(cut_box, std::vector<Ring>(1, f), 1, 0, 0, ok));   // :154 intersect(box, poly)
          if (!ok) return -4;
        }
        if (a != 0) { fn.push_back(live[q]); Aover.push_back(a); }          // :150-151
      }
      double Mtot = 0, Atot = 0;
      for (size_t q = 0; q < fn.size(); ++q) {
        const int s = E[fn[q]].src;
        Mtot += nz(F->mass[s]) * Aover[q] / nz(F->area[s]);                 // :153
        Atot += Aover[q];
      }
      const size_t at = (size_t)jj + (size_t)Ny * ii;
      out[13 * cells + at] = Atot / abox;                                   // c
      if (!(Mtot > 0)) continue;
      double sOver = 0;
      for (size_t q = 0; q < fn.size(); ++q) sOver += F->OverlapArea[E[fn[q]].src];
      out[14 * cells + at] = sOver / (double)fn.size();
      out[15 * cells + at] = Mtot;
      out[16 * cells + at] = Atot;
      auto wavg = [&](const double *field, int stride, int comp) {
        double s = 0;
        for (size_t q = 0; q < fn.size(); ++q) {
          const int src = E[fn[q]].src;
          s += nz(field[(size_t)stride * src + comp]) * nz(F->mass[src]) * Aover[q] / nz(F->area[src]);
        }
        return s / Mtot;
      };
      out[17 * cells + at] = wavg(F->h, 1, 0);
      out[0 * cells + at] = wavg(F->Ui, 1, 0); out[1 * cells + at] = wavg(F->Vi, 1, 0);
      out[2 * cells + at] = wavg(F->dUi_p, 1, 0); out[3 * cells + at] = wavg(F->dVi_p, 1, 0);
      // Stress(1,1) Stress(1,2) Stress(2,1) Stress(2,2) of the column-major 2 x 2: elements 0, 2, 1, 3
      const double sxx = wavg(F->Stress, 4, 0), syx = wavg(F->Stress, 4, 2), sxy = wavg(F->Stress, 4, 1), syy = wavg(F->Stress, 4, 3);
      out[5 * cells + at] = sxx; out[6 * cells + at] = syx; out[7 * cells + at] = sxy; out[8 * cells + at] = syy;
      out[9 * cells + at] = wavg(F->strain, 4, 0); out[10 * cells + at] = wavg(F->strain, 4, 2);
      out[11 * cells + at] = wavg(F->strain, 4, 1); out[12 * cells + at] = wavg(F->strain, 4, 3);
      // max(eig([sxx syx; sxy syy])): the larger root of the characteristic polynomial
      const double tr = 0.5 * (sxx + syy), df = 0.5 * (sxx - syy);
      double ev = tr + sqrt(df * df + syx * sxy);
      if (fabs(ev) > 1e8) ev = 0;
      out[4 * cells + at] = ev;
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Corner grinding up to the `grind` decision: Physical_Processes/corners.m:13-92 with
// polygon_operations/polyangles.m:1-173.  The uniform draws of :71 belong to the caller (the selection of Subzero.m:
// 339-348 and `break1` are applied in the tests with numpy's generator on both sides).  polyshape(...).Vertices is taken
// as the stored outline without its closing copy (floe outlines come out of polyshape: clockwise, no repeats; dupgone
// of polyangles.m:162-173 then removes nothing).
// ---------------------------------------------------------------------------------------------
namespace {

double t_acos_real(double a) {
  if (g_mode & 1) { if (a != a) return a; return a >= 1 ? 0.0 : (a <= -1 ? M_PI : acos(a)); }
  return orm_acos_real(a);
}

// polyangles.m:136-159
void pa_polyedge(double x1, double y1, double x2, double y2, double w[3]) {
  if (x1 == x2) { w[1] = 0; w[0] = -1 / x1; w[2] = 1; }
  else if (y1 == y2) { w[0] = 0; w[1] = -1 / y1; w[2] = 1; }
  else if (x1 == y1 && x2 == y2) { w[0] = 1; w[1] = 1; w[2] = 0; }
  else {
    w[0] = (y1 - y2) / (x1 * (y2 - y1) - y1 * (x2 - x1) + EPS);
    w[1] = -w[0] * (x2 - x1) / (y2 - y1);
    w[2] = 1;
  }
}

// polyangles.m:13-54 for a ring given without its closing copy
Vec polyangles(const Vec &x, const Vec &y) {
  const int n = (int)x.size();
  Vec ang(n);
  if (n == 0) return ang;
  Vec cx = x, cy = y;                                  // :21-24 close the polygon
  cx.push_back(x[0]); cy.push_back(y[0]);
  Vec dx(n), dy(n);                                    // :27 d = diff(xy)
  for (int k = 0; k < n; ++k) { dx[k] = cx[k + 1] - cx[k]; dy[k] = cy[k + 1] - cy[k]; }
  Vec raw(n);
  for (int k = 0; k < n; ++k) {
    const int k2 = (k + 1) % n;                        // :29 v2 = [d(2:end,:); d(1,:)]
    const double v1x = -dx[k], v1y = -dy[k], v2x = dx[k2], v2y = dy[k2];
    const double dot = v1x * v2x + v1y * v2y;          // :30
    double m1 = sqrt(v1x * v1x + v1y * v1y), m2 = sqrt(v2x * v2x + v2y * v2y);
    if (m1 == 0) m1 = EPS;                             // :37-38
    if (m2 == 0) m2 = EPS;
    raw[k] = t_acos_real(dot / m1 / m2) * 180 / M_PI;  // :39
  }
  for (int k = 0; k < n; ++k) ang[(k + 1) % n] = raw[k];   // :44 circshift
  // :57-133 convex_angle_test; sgn(1) = 1
  const double L = 1e25;
  for (int k = 1; k < n; ++k) {                        // k = 2 : length(xy) - 1, 1-based
    const double fx = cx[k - 1], fy = cy[k - 1], sx = cx[k], sy = cy[k], tx = cx[k + 1], ty = cy[k + 1];
    double w[3];
    pa_polyedge(fx, fy, sx, sy, w);
    const double ddx = sx - fx, ddy = sy - fy;
    double e[3] = {0, 0, 1};
    if (ddx <= 0 && ddy >= 0) { e[0] = L; e[1] = L; }
    else if (ddx <= 0 && ddy <= 0) { e[0] = -L; e[1] = L; }
    else if (ddx >= 0 && ddy <= 0) { e[0] = -L; e[1] = -L; }
    else { e[0] = L; e[1] = -L; }
    const double vp = w[0] * e[0] + w[1] * e[1] + w[2] * e[2];
    const double s = vp > 0 ? 1.0 : (vp < 0 ? -1.0 : vp);
    for (int q = 0; q < 3; ++q) w[q] = s * w[q];
    if (w[0] * tx + w[1] * ty + w[2] < 0) ang[k] = 360 - ang[k];   // :53-54,130-132
  }
  return ang;
}

} // namespace

extern "C" void or_polyangles(const double *x, const double *y, int n, double *out) {
  Vec a = polyangles(Vec(x, x + n), Vec(y, y + n));
  for (int k = 0; k < n; ++k) out[k] = a[k];
}

// ---------------------------------------------------------------------------------------------
// Physical_Processes/frac_corner.m:1-188 for ONE floe, restated for the case in which the reference's polyshape
// operations have closed forms: every corner to break is convex and its cut (:19-52) stays inside the two adjacent
// edges, clear of the neighbouring cuts and of every other edge of the outline.  poly1 (:53-80) is then the outline
// with that corner replaced by the two cut points, poly2 the triangle, subtract(poly, union(...)) (:136-137) the
// outline with every such corner replaced.  Returns the pieces in the order Floes = [FloeNEW fracturedFloes] (:185),
// or nothing when the floe is not that case (the product then leaves it to frac_corner.m as well).  Round 4: a short edge
// taken whole (:36-37, :43-44 -- the cut ends at the neighbouring vertex) is that case too when the vertex stays.
// polyshape conventions assumed, unpinned by anything the reference holds: area / centroid = ring_area / ring_centroid
// above; outlines clockwise from the first surviving vertex, a triangle from its first cut point.
// x, y: poly.Vertices (the stored c_alpha without its closing copy, corners.m:63-68); out: 12 doubles per piece
// [kind, Xi, Yi, area, mass, h, inertia_moment, rmax, alive, ksi_ice, factor of Stress / StressH, ring length], rings
// closed (c_alpha = c0, :96-97 / :149-150).
// ---------------------------------------------------------------------------------------------
namespace {
struct CornerPiece { double rec[12]; Vec cx, cy; };

double polygon_moments(const Vec &cx, const Vec &cy, double h) {       // PolygonMoments.m:14-32, xy = c0' (closed ring)
  const int N = (int)cx.size();
  Vec x = cx, y = cy;
  x.push_back(cx[0]); y.push_back(cy[0]);                             // :19-20
  double Ixx = 0, Iyy = 0;
  for (int i = 0; i < N; ++i) {
    const double wi = x[i] * y[i + 1] - x[i + 1] * y[i];              // :22
    Ixx += wi * ((y[i] + y[i + 1]) * (y[i] + y[i + 1]) - y[i] * y[i + 1]);   // :29
    Iyy += wi * ((x[i] + x[i + 1]) * (x[i] + x[i + 1]) - x[i] * x[i + 1]);   // :30
  }
  return fabs(1.0 / 12 * Ixx + 1.0 / 12 * Iyy) * h * 920.0;          // :32
}
bool seg_touch(double ax, double ay, double bx, double by, double cx, double cy, double dx, double dy) {
  auto orient = [](double px, double py, double qx, double qy, double rx, double ry) { return (qx - px) * (ry - py) - (qy - py) * (rx - px); };
  auto within = [](double px, double py, double qx, double qy, double rx, double ry) {
    return std::min(px, qx) <= rx && rx <= std::max(px, qx) && std::min(py, qy) <= ry && ry <= std::max(py, qy);
  };
  const double o1 = orient(ax, ay, bx, by, cx, cy), o2 = orient(ax, ay, bx, by, dx, dy), o3 = orient(cx, cy, dx, dy, ax, ay), o4 = orient(cx, cy, dx, dy, bx, by);
  if (o1 * o2 < 0 && o3 * o4 < 0) return true;
  return (o1 == 0 && within(ax, ay, bx, by, cx, cy)) || (o2 == 0 && within(ax, ay, bx, by, dx, dy)) ||
         (o3 == 0 && within(cx, cy, dx, dy, ax, ay)) || (o4 == 0 && within(cx, cy, dx, dy, bx, by));
}
CornerPiece make_piece(double kind, const Vec &vx, const Vec &vy, double mass, const OrFloes *F, int i, double factor) {
  CornerPiece P;
  double Xi, Yi;
  ring_centroid(vx.data(), vy.data(), (int)vx.size(), Xi, Yi);        // :92 / :145
  const double A = ring_area(vx.data(), vy.data(), (int)vx.size());   // :93 / :146
  double r2 = 0;
  for (size_t k = 0; k < vx.size(); ++k) { P.cx.push_back(vx[k] - Xi); P.cy.push_back(vy[k] - Yi); r2 = std::max(r2, (vx[k] - Xi) * (vx[k] - Xi) + (vy[k] - Yi) * (vy[k] - Yi)); }
  P.cx.push_back(vx[0] - Xi); P.cy.push_back(vy[0] - Yi);             // :96 / :149
  const double h = mass / (920.0 * A);                                // :95 / :148
  P.rec[0] = kind; P.rec[1] = F->Xi[i] + Xi; P.rec[2] = F->Yi[i] + Yi; P.rec[3] = A; P.rec[4] = mass; P.rec[5] = h;
  P.rec[6] = polygon_moments(P.cx, P.cy, h); P.rec[7] = sqrt(r2); P.rec[8] = (A < 1e4) ? 0.0 : 1.0;   // :98,:100,:113-115 / :151,:153,:166-168
  P.rec[9] = A / F->area[i] * F->ksi_ice[i]; P.rec[10] = factor; P.rec[11] = (double)P.cx.size();   // :119 / :172, :103-104 / :156-157
  return P;
}
std::vector<CornerPiece> frac_corner_plain(const OrFloes *F, int i, const uint8_t *grind_in) {
  std::vector<CornerPiece> none, out;
  const int v0 = F->vert_off[i];
  int n = F->vert_off[i + 1] - v0;
  if (n > 1 && F->cax[v0] == F->cax[v0 + n - 1] && F->cay[v0] == F->cay[v0 + n - 1]) --n;
  if (n < 3) return none;
  Vec x(F->cax + v0, F->cax + v0 + n), y(F->cay + v0, F->cay + v0 + n);
  std::vector<uint8_t> grind(grind_in, grind_in + n);
  const Vec angles = polyangles(x, y);                                // :10
  const double Anorm = 180.0 - 360.0 / n;                             // :11
  {                                                                   // :12-15 unique(angles(grind))
    Vec ga;
    for (int k = 0; k < n; ++k) if (grind[k]) ga.push_back(angles[k]);
    Vec u = ga;
    std::sort(u.begin(), u.end());
    u.erase(std::unique(u.begin(), u.end()), u.end());
    if (u.size() < ga.size() && grind[0] && grind[n - 1]) grind[n - 1] = 0;
  }
  double alph = angles[0];
  for (int k = 1; k < n; ++k) alph = std::min(alph, angles[k]);       // :16
  const double Apoly = ring_area(x.data(), y.data(), n);
  if (!(Apoly > 0)) return none;
  struct Cut { double x1, y1, x2, y2; };
  std::vector<Cut> cut(n);
  int ng = 0;
  for (int jj = 0; jj < n; ++jj) {
    if (!grind[jj]) continue;
    ++ng;
    const int p = (jj + n - 1) % n, q = (jj + 1) % n;                 // :25-33
    const double X1 = x[p], Y1 = y[p], X2 = x[jj], Y2 = y[jj], X3 = x[q], Y3 = y[q];
    const double d1 = sqrt((X1 - X2) * (X1 - X2) + (Y1 - Y2) * (Y1 - Y2)), d2 = sqrt((X3 - X2) * (X3 - X2) + (Y3 - Y2) * (Y3 - Y2));
    const double d = std::min(d1, d2);                                // :34-35
    Cut c;
    // :36-37 / :43-44 an edge shorter than 120 m in x and in y goes whole: the cut ends AT the neighbouring vertex.  The
    // closed form covers it when that vertex stays (were it ground too, the two triangles would overlap: the union /
    // subtract of :136-137) and the outline keeps at least four vertices.
    const bool s1 = fabs(X1 - X2) < 120 && fabs(Y1 - Y2) < 120, s2 = fabs(X3 - X2) < 120 && fabs(Y3 - Y2) < 120;
    if ((s1 || s2) && n < 5) return none;
    if ((s1 && grind[p]) || (s2 && grind[q])) return none;
    if (s1) { c.x1 = X1; c.y1 = Y1; }                                 // :36-37
    else if (fabs(d * alph / Anorm / 5) < 120) { c.x1 = X2 + 120 / d1 * (X1 - X2); c.y1 = Y2 + 120 / d1 * (Y1 - Y2); }   // :38-39
    else { c.x1 = X2 + d / d1 * alph / Anorm * (X1 - X2) / 5; c.y1 = Y2 + d / d1 * alph / Anorm * (Y1 - Y2) / 5; }     // :41
    if (s2) { c.x2 = X3; c.y2 = Y3; }                                 // :43-44
    else if (fabs(d * alph / Anorm / 5) < 120) { c.x2 = X2 + 120 / d2 * (X3 - X2); c.y2 = Y2 + 120 / d2 * (Y3 - Y2); }   // :45-46
    else { c.x2 = X2 + d / d2 * alph / Anorm * (X3 - X2) / 5; c.y2 = Y2 + d / d2 * alph / Anorm * (Y3 - Y2) / 5; }     // :48
    if (fabs(c.x1 - c.x2) < 0.1) { if (s1 || s2) return none; c.x1 = c.x1 + 1; }   // :50-52 (with a cut at a vertex: not this form)
    cut[jj] = c;
    // the closed form: a convex corner, cut points inside the two edges (or at their far vertex: a short edge), the cut
    // clear of every other edge
    if (angles[jj] >= 180) return none;
    const double t1 = sqrt((c.x1 - X2) * (c.x1 - X2) + (c.y1 - Y2) * (c.y1 - Y2)), t2 = sqrt((c.x2 - X2) * (c.x2 - X2) + (c.y2 - Y2) * (c.y2 - Y2));
    if (!(t1 > 0 && (s1 || t1 < d1) && t2 > 0 && (s2 || t2 < d2))) return none;
    const int pp = (p + n - 1) % n, qq = (q + 1) % n;
    for (int e = 0; e < n; ++e) {
      if (e == p || e == jj) continue;
      if (s1 && e == pp) continue;                                    // (ends where the cut starts: below)
      if (s2 && e == q) continue;                                     // (starts where the cut ends)
      if (seg_touch(c.x1, c.y1, c.x2, c.y2, x[e], y[e], x[(e + 1) % n], y[(e + 1) % n])) return none;
    }
    // a cut from a vertex must leave the outline's edge there at an angle (in line with it, polyshape drops the vertex)
    if (s1 && (x[p] - x[pp]) * (c.y2 - y[pp]) - (y[p] - y[pp]) * (c.x2 - x[pp]) == 0) return none;
    if (s2 && (x[q] - c.x1) * (y[qq] - c.y1) - (y[q] - c.y1) * (x[qq] - c.x1) == 0) return none;
  }
  if (ng == 0) return none;
  for (int k = 0; k < n; ++k) {                                       // neighbouring cuts on one edge must not meet
    const int q = (k + 1) % n;
    if (!(grind[k] && grind[q])) continue;
    const double L = sqrt((x[q] - x[k]) * (x[q] - x[k]) + (y[q] - y[k]) * (y[q] - y[k]));
    const double ta = sqrt((cut[k].x2 - x[k]) * (cut[k].x2 - x[k]) + (cut[k].y2 - y[k]) * (cut[k].y2 - y[k]));
    const double tb = sqrt((cut[q].x1 - x[q]) * (cut[q].x1 - x[q]) + (cut[q].y1 - y[q]) * (cut[q].y1 - y[q]));
    if (!(ta + tb < L)) return none;
  }
  // :133-180 the floe less its corners
  Vec nx, ny;
  for (int k = 0; k < n; ++k) {
    if (grind[k]) {
      // (a cut point that IS the neighbouring vertex -- a short edge -- is in the outline already)
      const int kp = (k + n - 1) % n, kn = (k + 1) % n;
      if (!(cut[k].x1 == x[kp] && cut[k].y1 == y[kp])) { nx.push_back(cut[k].x1); ny.push_back(cut[k].y1); }
      if (!(cut[k].x2 == x[kn] && cut[k].y2 == y[kn])) { nx.push_back(cut[k].x2); ny.push_back(cut[k].y2); }
    }
    else { nx.push_back(x[k]); ny.push_back(y[k]); }
  }
  if (nx.size() < 3) return none;
  const double Anew = ring_area(nx.data(), ny.data(), (int)nx.size());
  if (!(Anew > 0)) return none;
  out.push_back(make_piece(0.0, nx, ny, F->mass[i] * Anew / Apoly, F, i, Anew / Apoly));   // :147, :156-157
  // :53-129 one triangle per corner, with the Atot of poly1 and poly2 of ITS cut
  for (int jj = 0; jj < n; ++jj) {
    if (!grind[jj]) continue;
    Vec tx = {cut[jj].x1, cut[jj].x2, x[jj]}, ty = {cut[jj].y1, cut[jj].y2, y[jj]};   // Pc2, :55 / :64 / :73
    const double At = ring_area(tx.data(), ty.data(), 3);
    if (!(At > 0)) return none;
    Vec px, py;                                                       // Pc1
    for (int k = 0; k < n; ++k) {
      if (k == jj) { px.push_back(cut[jj].x1); py.push_back(cut[jj].y1); px.push_back(cut[jj].x2); py.push_back(cut[jj].y2); }
      else { px.push_back(x[k]); py.push_back(y[k]); }
    }
    const double A1 = ring_area(px.data(), py.data(), (int)px.size());
    double Atot = 0;                                                  // :82-84 a = R1(area(R1) > 10)
    if (A1 > 10) Atot += A1;
    if (At > 10) Atot += At;
    if (!(Atot > 0)) return none;
    const double Mtot = F->mass[i] * Atot / Apoly;                    // :85
    if ((tx[1] - tx[0]) * (ty[2] - ty[0]) - (tx[2] - tx[0]) * (ty[1] - ty[0]) > 0) { std::swap(tx[1], tx[2]); std::swap(ty[1], ty[2]); }   // clockwise
    out.push_back(make_piece(1.0, tx, ty, Mtot * At / Atot, F, i, At / Atot));   // :94, :103-104
  }
  return out;
}
} // namespace

extern "C" int or_frac_corner(const OrFloes *F, const int *ids, int nsel, const int *offsets, const uint8_t *grind, int *npieces, int *piece_off,
                              double *pieces, long long cap_pieces, int *ring_off, double *rx, double *ry, long long cap_ring) {
  long long np = 0, nr = 0;
  piece_off[0] = 0; ring_off[0] = 0;
  for (int s = 0; s < nsel; ++s) {
    const std::vector<CornerPiece> P = frac_corner_plain(F, ids[s] - 1, grind + offsets[s]);
    npieces[s] = (int)P.size();
    for (const CornerPiece &q : P) {
      if (np + 1 > cap_pieces || nr + (long long)q.cx.size() > cap_ring) return -3;
      for (int k = 0; k < 12; ++k) pieces[np * 12 + k] = q.rec[k];
      for (size_t k = 0; k < q.cx.size(); ++k) { rx[nr + (long long)k] = q.cx[k]; ry[nr + (long long)k] = q.cy[k]; }
      ++np; nr += (long long)q.cx.size(); ring_off[np] = (int)nr;
    }
    piece_off[s + 1] = (int)np;
  }
  return 0;
}

// ids[q]: 1-based floe numbers of Floe(~keep); inter_off / inter: the floes' contact rows (7 columns, CSR), as the step
// left them.  offsets[nsub + 1], count[nsub], angles / da at the offsets (stored outline lengths back to back).
extern "C" int or_corner_candidates(const OrFloes *F, const int *inter_off, const double *inter, const int *ids, int nsub, int Nb,
                                    const double *c2b, int *offsets, int *count, double *angles, uint8_t *da, int *n_ext) {
  const int n = F->n;
  double Lx = c2b[0], Ly = c2b[1];
  for (int k = 1; k < 5; ++k) { Lx = std::max(Lx, c2b[2 * k]); Ly = std::max(Ly, c2b[2 * k + 1]); }
  struct Ent { int src; double x, y; int alive; };
  std::vector<Ent> E;
  for (int i = 0; i < n; ++i) { Ent e = {i, F->Xi[i], F->Yi[i], F->alive[i]}; E.push_back(e); }
  auto sgn = [](double v) { return (v > 0) ? 1.0 : (v < 0 ? -1.0 : 0.0); };
  auto extent = [&](const Ent &e, int dim) {
    int v0 = F->vert_off[e.src], nv = F->vert_off[e.src + 1] - v0;
    double m = 0;
    for (int k = 0; k < nv; ++k) m = std::max(m, fabs(dim == 0 ? F->cax[v0 + k] + e.x : F->cay[v0 + k] + e.y));
    return m;
  };
  for (int i = 0; i < n; ++i)                                                   // :21-30
    if (E[i].alive && extent(E[i], 0) > Lx) { Ent g = E[i]; g.x = E[i].x - 2 * Lx * sgn(E[i].x); E.push_back(g); }
  const int n2 = (int)E.size();
  for (int i = 0; i < n2; ++i)                                                  // :38-47
    if (E[i].alive && extent(E[i], 1) > Ly) { Ent g = E[i]; g.y = E[i].y - 2 * Ly * sgn(E[i].y); E.push_back(g); }
  const int N0 = (int)E.size();                                                 // :52
  if (n_ext) *n_ext = N0;
  Vec bx(5), by(5);
  for (int k = 0; k < 5; ++k) { bx[k] = c2b[2 * k]; by[k] = c2b[2 * k + 1]; }
  int at = 0;
  for (int q = 0; q < nsub; ++q) {
    const int i = ids[q] - 1;
    const int v0 = F->vert_off[i], nvs = F->vert_off[i + 1] - v0;
    offsets[q] = at; at += nvs;
    count[q] = 0;
    const int r0 = inter_off[i], r1 = inter_off[i + 1];
    if (q < Nb || r1 == r0) continue;                                           // :55-56
    int nv = nvs;
    if (nv > 1 && F->cax[v0] == F->cax[v0 + nv - 1] && F->cay[v0] == F->cay[v0 + nv - 1]) --nv;   // :66-68
    Vec px(F->cax + v0, F->cax + v0 + nv), py(F->cay + v0, F->cay + v0 + nv);
    Vec ang = polyangles(px, py);                                               // :69
    Vec tx(nv), ty(nv);                                                         // :64 polytrue
    for (int k = 0; k < nv; ++k) { tx[k] = px[k] + F->Xi[i]; ty[k] = py[k] + F->Yi[i]; }
    std::vector<int> in(nv, 0), flag(nv, 0);
    bool bnd = false;
    for (int r = r0; r < r1; ++r) {
      const double partner = inter[7 * r];
      if (std::isinf(partner)) { bnd = true; continue; }                        // :75-76
      if (!(partner <= N0)) continue;                                           // :77
      const Ent &e = E[(int)partner - 1];
      const int p0 = F->vert_off[e.src], pn = F->vert_off[e.src + 1] - p0;
      Vec cx(pn), cy(pn);
      for (int k = 0; k < pn; ++k) { cx[k] = F->cax[p0 + k] + e.x; cy[k] = F->cay[p0 + k] + e.y; }   // :79
      for (int k = 0; k < nv; ++k) in[k] += inpolygon1(tx[k], ty[k], cx, cy) ? 1 : 0;               // :80-81
      // :60-61,73,88 the vertex nearest to the contact point
      int best = 0; double bd = 0;
      for (int k = 0; k < nv; ++k) {
        const double ddx = tx[k] - inter[7 * r + 3], ddy = ty[k] - inter[7 * r + 4];
        const double d = ddx * ddx + ddy * ddy;
        if (k == 0 || d < bd) { bd = d; best = k; }
      }
      flag[best] = 1;
    }
    if (bnd) for (int k = 0; k < nv; ++k) in[k] += inpolygon1(tx[k], ty[k], bx, by) ? 0 : 1;        // :83-86
    for (int k = 0; k < nv; ++k) if (in[k] > 0) flag[k] = 1;                   // :87
    count[q] = nv;
    for (int k = 0; k < nv; ++k) { angles[offsets[q] + k] = ang[k]; da[offsets[q] + k] = (uint8_t)flag[k]; }
  }
  offsets[nsub] = at;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// fracture_floe.m:14-52 -- the deformation a floe receives from the partner of its largest overlap before it is
// tessellated.  ids: 1-based floes (fracture.m:51-55); inter_off / inter: the contact rows of the step.
// out: changed[s]; Xi / Yi / area [s]; new c_alpha rings back to back (ring_off[nsel + 1]).
// ---------------------------------------------------------------------------------------------
extern "C" int or_fracture_deform(const OrFloes *F, const int *inter_off, const double *inter, const int *ids, int nsel, int *changed,
                                  double *Xi, double *Yi, double *area, int *ring_off, double *cx, double *cy, int cap) {
  int at = 0;
  ring_off[0] = 0;
  for (int s = 0; s < nsel; ++s) {
    const int i = ids[s] - 1;
    changed[s] = 0; Xi[s] = Yi[s] = area[s] = 0;
    ring_off[s + 1] = at;
    // :16-22 rows with a finite partner; [~,k] = max(a(:,7))
    int k = -1;
    for (int r = inter_off[i]; r < inter_off[i + 1]; ++r) {
      if (std::isinf(inter[7 * r])) continue;
      if (k < 0 || inter[7 * r + 6] > inter[7 * k + 6]) k = r;
    }
    if (k < 0) continue;
    const int v0 = F->vert_off[i], nv = F->vert_off[i + 1] - v0;
    Vec X1(nv), Y1(nv);
    for (int q = 0; q < nv; ++q) { X1[q] = F->cax[v0 + q] + F->Xi[i]; Y1[q] = F->cay[v0 + q] + F->Yi[i]; }   // :23-24
    const double clip = inter[7 * k];
    if (!(clip < F->n + 1)) continue;                                       // :26
    const int j = (int)clip - 1;
    const int w0 = F->vert_off[j], nw = F->vert_off[j + 1] - w0;
    Vec X2(nw), Y2(nw);
    for (int q = 0; q < nw; ++q) { X2[q] = F->cax[w0 + q] + F->Xi[j]; Y2[q] = F->cay[w0 + q] + F->Yi[j]; }   // :27-28
    bool ok = true;
    std::vector<Ring> reg = polyclip(X1, Y1, X2, Y2, 1, ok);                // :29 'int'
    if (!ok) return -4;
    if (reg.empty()) continue;                                              // :31
    const Ring &t = reg[0];                                                 // :32
    double xm, ym;
    ring_centroid(t.x.data(), t.y.data(), (int)t.x.size(), xm, ym);         // :33
    Vec d;
    p_poly_dist(Vec(1, xm), Vec(1, ym), t.x, t.y, d);                       // :34
    const double fx = inter[7 * k + 1], fy = inter[7 * k + 2];
    const double Fn = sqrt(fx * fx + fy * fy);                              // :35 vecnorm
    const double xs = fx * fabs(d[0]) / 2 / Fn, ys = fy * fabs(d[0]) / 2 / Fn;   // :36-37
    for (int q = 0; q < nw; ++q) { X2[q] = X2[q] + xs; Y2[q] = Y2[q] + ys; }      // :38
    reg = polyclip(X1, Y1, X2, Y2, 0, ok);                                  // :39 'dif'
    if (!ok) return -4;
    if (reg.empty()) continue;                                              // :40
    const Ring &u = reg[0];                                                 // :41
    const int m = (int)u.x.size();
    const double Anew = polyarea(u.x.data(), u.y.data(), m);                // :42
    if (!(Anew / F->area[i] > 0.9)) continue;                               // :43
    ring_centroid(u.x.data(), u.y.data(), m, xm, ym);                       // :44
    if (at + m > cap) return -2;
    changed[s] = 1; Xi[s] = xm; Yi[s] = ym; area[s] = Anew;                 // :45-46
    for (int q = 0; q < m; ++q) { cx[at + q] = u.x[q] - xm; cy[at + q] = u.y[q] - ym; }   // :47
    at += m;
    ring_off[s + 1] = at;
  }
  return 0;
}
