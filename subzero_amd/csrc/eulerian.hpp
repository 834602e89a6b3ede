// eulerian.hpp -- calc_eulerian_data.m:1-192: mass-weighted averages of the floe fields over a coarse
// Nx x Ny grid; the overlap of a floe with a grid cell is the batched-clip primitive of this path in its simplest form
// (ring against axis-aligned rectangle).
//
// area(intersect(box, poly)) (calc_eulerian_data.m:148-149) is a MathWorks polyshape operation whose source is not in
// the reference; it is restated by its definition: the ring clipped to the rectangle by four half-planes
// (Sutherland-Hodgman, streamed: each stage keeps its first and its previous vertex only, so a ring of any length
// needs no buffer) and the shoelace area of what comes out, taken relative to the first emitted vertex.  For a ring
// that leaves and re-enters the rectangle the clipped outline runs along the rectangle's edges there and back: the
// area is still exact.  The oracle (oracle/subzero_oracle.cpp or_eulerian) writes the same definition independently;
// tests pin both against the fixed-point 'int' clip of the reference's own Clipper.
#pragma once
#include "step_core.hpp"

namespace sze {

struct BoxArea {
  // half-planes in the order x >= xl, x <= xh, y >= yl, y <= yh
  double bound[4];
  double fx[4], fy[4], px[4], py[4];   // first / previous vertex seen by each stage
  int seen[4];
  double ox, oy, qx, qy, sum;          // shoelace: origin = first emitted vertex, previous emitted vertex, running sum
  int emitted;
  SZ_HD void start(double xl, double xh, double yl, double yh) {
    bound[0] = xl; bound[1] = xh; bound[2] = yl; bound[3] = yh;
    for (int k = 0; k < 4; ++k) { seen[k] = 0; fx[k] = fy[k] = px[k] = py[k] = 0; }
    ox = oy = qx = qy = sum = 0; emitted = 0;
  }
  SZ_HD bool inside(int k, double x, double y) const {
    return k == 0 ? x >= bound[0] : k == 1 ? x <= bound[1] : k == 2 ? y >= bound[2] : y <= bound[3];
  }
  SZ_HD void out(double x, double y) {
    if (!emitted) { ox = x; oy = y; emitted = 1; }
    else {
      const double ax = qx - ox, ay = qy - oy, bx = x - ox, by = y - oy;
      sum += ax * by - bx * ay;
    }
    qx = x; qy = y;
  }
  // the point where the edge a -> b meets the boundary of half-plane k
  SZ_HD void cut(int k, double ax, double ay, double bx, double by, double &x, double &y) const {
    if (k < 2) { const double t = (bound[k] - ax) / (bx - ax); x = bound[k]; y = ay + t * (by - ay); }
    else { const double t = (bound[k] - ay) / (by - ay); y = bound[k]; x = ax + t * (bx - ax); }
  }
  template <int K> SZ_HD void edge(double ax, double ay, double bx, double by) {   // edge a -> b arrives at stage K
    const bool ia = inside(K, ax, ay), ib = inside(K, bx, by);
    if (ia != ib) { double x, y; cut(K, ax, ay, bx, by, x, y); feed<K + 1>(x, y); }
    if (ib) feed<K + 1>(bx, by);
  }
  template <int K> SZ_HD void feed(double x, double y) {                           // vertex (x, y) arrives at stage K
    if constexpr (K == 4) { out(x, y); }
    else {
      if (!seen[K]) { seen[K] = 1; fx[K] = x; fy[K] = y; px[K] = x; py[K] = y; return; }
      const double ax = px[K], ay = py[K];
      px[K] = x; py[K] = y;
      edge<K>(ax, ay, x, y);
    }
  }
  // closes the ring of every stage in turn (the edge from its last vertex back to its first), then the shoelace sum
  SZ_HD double finish() {
    if (seen[0]) edge<0>(px[0], py[0], fx[0], fy[0]);
    if (seen[1]) edge<1>(px[1], py[1], fx[1], fy[1]);
    if (seen[2]) edge<2>(px[2], py[2], fx[2], fy[2]);
    if (seen[3]) edge<3>(px[3], py[3], fx[3], fy[3]);
    if (emitted) { const double ax = qx - ox, ay = qy - oy, bx = ox - ox, by = oy - oy; sum += ax * by - bx * ay; }
    const double a = 0.5 * sum;
    return a < 0 ? -a : a;
  }
};
// NOTE: a stage's first vertex is only remembered, never emitted by itself: the closing edge of finish() (last -> first)
// emits it, so every vertex inside the half-plane reaches the next stage exactly once.

// area of ring (x[k] + ox, y[k] + oy), k < n (open: no closing duplicate), inside the rectangle
SZ_HD double ring_box_area(const double *x, const double *y, int n, double ox, double oy, double xl, double xh, double yl, double yh) {
  BoxArea b;
  b.start(xl, xh, yl, yh);
  for (int k = 0; k < n; ++k) b.feed<0>(x[k] + ox, y[k] + oy);
  return b.finish();
}

} // namespace sze
