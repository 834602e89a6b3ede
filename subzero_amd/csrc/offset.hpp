// offset.hpp -- the first half of ClipperOffset (clipper.cpp:3786-4218) as the mex's offset mode uses it
// (mexclipper.cpp:146-203: closed polygons only): AddPath's clean-up of the contours, the orientation fix, and DoOffset
// -- one pass over the vertices of each contour that builds the raw offset polygon from unit normals and a mitred,
// squared or rounded join per vertex.  Plain host C++: sin, cos, acos, atan2, tan of the host's libm decide integer
// roundings here exactly as they do in the reference library.  The second half, the union that cleans the raw
// polygons up, is szv::offset_cleanup (vatti.hpp) -- on the GPU in the product (clip.hip szOffsetPaths), on the host
// in the CPU test build (tests/hostsim).
#pragma once
#include <stdint.h>
#include <algorithm>
#include <cmath>
#include <vector>

namespace szo {

struct IPt { int64_t X, Y; };
struct DPt { double X, Y; };
typedef std::vector<IPt> OPath;

inline int64_t oround(double v) { return v < 0 ? (int64_t)(v - 0.5) : (int64_t)(v + 0.5); }     // clipper.cpp:136-140
inline bool orientation_of(const OPath &poly) {                                                  // :385-403
  const int size = (int)poly.size();
  if (size < 3) return 0.0 >= 0;
  double a = 0;
  for (int i = 0, j = size - 1; i < size; ++i) { a += ((double)poly[j].X + poly[i].X) * ((double)poly[j].Y - poly[i].Y); j = i; }
  return -a * 0.5 >= 0;
}
inline DPt unit_normal(const IPt &p1, const IPt &p2) {                                          // :3769-3780
  if (p2.X == p1.X && p2.Y == p1.Y) { DPt z = {0, 0}; return z; }
  double Dx = (double)(p2.X - p1.X), dy = (double)(p2.Y - p1.Y);
  const double f = 1 * 1.0 / std::sqrt(Dx * Dx + dy * dy);
  Dx *= f; dy *= f;
  DPt n = {dy, -Dx};
  return n;
}

struct Offsetter {
  double MiterLimit = 2.0, ArcTolerance = 0.25;
  std::vector<OPath> contours, dest;
  int low_path = -1, low_k = 0;
  // state of DoOffset
  OPath src, out;
  std::vector<DPt> normals;
  double delta = 0, sinA = 0, msin = 0, mcos = 0, miterLim = 0, stepsPerRad = 0;

  void add_path(const int64_t *x, const int64_t *y, int n) {                                    // :3809-3851, etClosedPolygon
    int highI = n - 1;
    if (highI < 0) return;
    while (highI > 0 && x[0] == x[highI] && y[0] == y[highI]) highI--;
    OPath c;
    IPt p0 = {x[0], y[0]};
    c.push_back(p0);
    int j = 0, k = 0;
    for (int i = 1; i <= highI; i++)
      if (c[j].X != x[i] || c[j].Y != y[i]) {
        j++;
        IPt p = {x[i], y[i]};
        c.push_back(p);
        if (y[i] > c[k].Y || (y[i] == c[k].Y && x[i] < c[k].X)) k = j;
      }
    if (j < 2) return;
    contours.push_back(c);
    if (low_path < 0) { low_path = (int)contours.size() - 1; low_k = k; }
    else {
      const IPt ip = contours[low_path][low_k];
      if (c[k].Y > ip.Y || (c[k].Y == ip.Y && c[k].X < ip.X)) { low_path = (int)contours.size() - 1; low_k = k; }
    }
  }
  void fix_orientations() {                                                                    // :3861-3884
    if (low_path >= 0 && !orientation_of(contours[low_path]))
      for (size_t i = 0; i < contours.size(); ++i) std::reverse(contours[i].begin(), contours[i].end());
  }
  void push(double X, double Y) { IPt p = {oround(X), oround(Y)}; out.push_back(p); }
  void do_square(int j, int k) {                                                               // :4175-4185
    const double dx = std::tan(std::atan2(sinA, normals[k].X * normals[j].X + normals[k].Y * normals[j].Y) / 4);
    push(src[j].X + delta * (normals[k].X - normals[k].Y * dx), src[j].Y + delta * (normals[k].Y + normals[k].X * dx));
    push(src[j].X + delta * (normals[j].X + normals[j].Y * dx), src[j].Y + delta * (normals[j].Y - normals[j].X * dx));
  }
  void do_miter(int j, int k, double r) {                                                      // :4188-4193
    const double q = delta / r;
    push(src[j].X + (normals[k].X + normals[j].X) * q, src[j].Y + (normals[k].Y + normals[j].Y) * q);
  }
  void do_round(int j, int k) {                                                                // :4196-4214
    const double a = std::atan2(sinA, normals[k].X * normals[j].X + normals[k].Y * normals[j].Y);
    const int steps = std::max((int)oround(stepsPerRad * std::fabs(a)), 1);
    double X = normals[k].X, Y = normals[k].Y, X2;
    for (int i = 0; i < steps; ++i) {
      push(src[j].X + X * delta, src[j].Y + Y * delta);
      X2 = X;
      X = X * mcos - msin * Y;
      Y = X2 * msin + Y * mcos;
    }
    push(src[j].X + normals[j].X * delta, src[j].Y + normals[j].Y * delta);
  }
  void offset_point(int j, int &k, int join) {                                                 // :4131-4172
    sinA = (normals[k].X * normals[j].Y - normals[j].X * normals[k].Y);
    if (std::fabs(sinA * delta) < 1.0) {
      const double cosA = (normals[k].X * normals[j].X + normals[j].Y * normals[k].Y);
      if (cosA > 0) { push(src[j].X + normals[k].X * delta, src[j].Y + normals[k].Y * delta); return; }
    } else if (sinA > 1.0) sinA = 1.0;
    else if (sinA < -1.0) sinA = -1.0;
    if (sinA * delta < 0) {
      push(src[j].X + normals[k].X * delta, src[j].Y + normals[k].Y * delta);
      out.push_back(src[j]);
      push(src[j].X + normals[j].X * delta, src[j].Y + normals[j].Y * delta);
    } else if (join == 2) {
      const double r = 1 + (normals[j].X * normals[k].X + normals[j].Y * normals[k].Y);
      if (r >= miterLim) do_miter(j, k, r); else do_square(j, k);
    } else if (join == 0) do_square(j, k);
    else do_round(j, k);
    k = j;
  }
  void do_offset(double d, int join) {                                                         // :3958-4128, closed polygons
    const double pi = 3.141592653589793238, two_pi = pi * 2, def_arc_tolerance = 0.25;
    dest.clear();
    delta = d;
    if (d > -1.0e-20 && d < 1.0e-20) { dest = contours; return; }
    miterLim = MiterLimit > 2 ? 2 / (MiterLimit * MiterLimit) : 0.5;
    double y;
    if (ArcTolerance <= 0.0) y = def_arc_tolerance;
    else if (ArcTolerance > std::fabs(d) * def_arc_tolerance) y = std::fabs(d) * def_arc_tolerance;
    else y = ArcTolerance;
    double steps = pi / std::acos(1 - y / std::fabs(d));
    if (steps > std::fabs(d) * pi) steps = std::fabs(d) * pi;
    msin = std::sin(two_pi / steps);
    mcos = std::cos(two_pi / steps);
    stepsPerRad = steps / two_pi;
    if (d < 0.0) msin = -msin;
    for (size_t i = 0; i < contours.size(); ++i) {
      src = contours[i];
      const int len = (int)src.size();
      if (len == 0 || (d <= 0 && len < 3)) continue;
      out.clear();
      normals.clear();
      for (int j = 0; j < len - 1; ++j) normals.push_back(unit_normal(src[j], src[j + 1]));
      normals.push_back(unit_normal(src[len - 1], src[0]));
      int k = len - 1;
      for (int j = 0; j < len; ++j) offset_point(j, k, join);
      dest.push_back(out);
    }
  }
};

} // namespace szo
