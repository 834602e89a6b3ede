// contact.hpp -- one candidate pair of the narrow-phase: the contact model of the reference's
// collisions/floe_interactions.m:1-199 for one lane, on top of the sweep in vatti.hpp.
//
// Everything a pair needs lives in one scratch slice (PairScratch): the sweep workspace, the two rings
// in 2^32 fixed point, the shifted ring of the direction check, two region slots and the crossing
// points.  Arithmetic follows the .m text operation by operation (IEEE double, no FMA) so that contact
// rows agree with the CPU oracle to the last bit:
//   spring constants                       floe_interactions.m:10-21
//   overlap regions, 'int' / 'dif'         :25-41      (polyclip.m:63-80 fixed-point round trip)
//   region areas, merge flags              :43-60
//   ring closure, crossings P              :62-70      (InterX.m:54-81)
//   small-region filter                    :78-83
//   per region: centroid, contact points   :95-100
//               force direction            :103-137    (p_poly_dist.m:111-288, inpolygon.m:150-224)
//               direction check by re-clip :140-165
//               normal + tangential force  :167-187
#pragma once
#include <math.h>
#include "vatti.hpp"
#include "convex.hpp"
#include "trim.hpp"

#if defined(__HIPCC__)
#define SZ_HD_NOINLINE static __host__ __device__ __attribute__((noinline))
#else
#define SZ_HD_NOINLINE static __attribute__((noinline))
#endif

namespace szc {

using szv::i64;
using szv::Pt;

SZ_HD double absd(double v) { return v < 0 ? -v : v; }
SZ_HD double mind(double a, double b) { return b < a ? b : a; }   // std::min(a, b)
SZ_HD double maxd(double a, double b) { return a < b ? b : a; }   // std::max(a, b)
SZ_HD double hyp(double a, double b) { return ::sqrt(a * a + b * b); }
SZ_HD bool is_inf(double v) { return v > 1.7976931348623157e308 || v < -1.7976931348623157e308; }
#define SZC_INF (__builtin_huge_val())
#define SZC_EPS 2.220446049250313e-16

using szx::to_fixed; // MATLAB int64(v * 2^32) (polyclip.m:63-66), convex.hpp
SZ_HD double from_fixed(i64 v) { return (double)v / 4294967296.0; } // polyclip.m:67

// ---- one shared, non-inlined instance of the sweep for all three clips of a pair ----
struct FixedRing {
  const i64 *x, *y;
  SZ_HD Pt operator()(int i) const { Pt p; p.x = x[i]; p.y = y[i]; return p; }
};
struct RegionSink { // stores rings, or only counts them when ox == nullptr
  i64 *ox, *oy;
  int *roff;
  int cap_v, cap_r, nv, nr;
  SZ_HD bool ring_begin() { return ox == nullptr || nr < cap_r; }
  SZ_HD bool vertex(const Pt &p) {
    if (ox == nullptr) return true;
    if (nv >= cap_v) return false;
    ox[nv] = p.x; oy[nv] = p.y; ++nv;
    return true;
  }
  SZ_HD bool ring_end() { ++nr; if (ox != nullptr) roff[nr] = nv; return true; }
};
SZ_HD_NOINLINE int clip_fixed(void *ws, int cap_nv, const i64 *ax, const i64 *ay, int na, const i64 *bx,
                           const i64 *by, int nb, int op, RegionSink *sink) {
  FixedRing a = {ax, ay}, b = {bx, by};
  sink->nv = 0; sink->nr = 0;
  if (sink->ox != nullptr) sink->roff[0] = 0;
  return szv::clip_rings(ws, cap_nv, a, na, b, nb, op, szv::FILL_EVENODD, szv::FILL_EVENODD, *sink);
}

// A pointer the caller vouches to be LDS.  The kernels hand their lanes' blocks of the dynamic LDS down as ordinary
// pointers; telling the compiler what they are lets it emit ds_ instructions (LDS latency, no flat address check)
// without a non-kernel function ever naming an `extern __shared__` array itself.  That matters: a function that does
// is lowered through a per-kernel offset table indexed by an implicit kernel-id register as soon as two kernels of
// the module can reach such functions, and the 32-vertex second pass of round 1 -- the second kernel to do so -- was
// the one that faulted ("the k_narrow_convex_big fault").  No callee re-derives a block from the
// launch geometry any more, so there is no unchecked launch contract either.
#if defined(__HIP_DEVICE_COMPILE__)
#define SZ_ASSUME_LDS(ptr) __builtin_assume(__builtin_amdgcn_is_shared((const __attribute__((address_space(0))) void *)(ptr)))
#else
#define SZ_ASSUME_LDS(ptr) ((void)(ptr))
#endif

// The same sweep with its Wide records in the lane's slice of LDS (`lds_ws`, handed down by k_narrow_lds; on the host
// build a plain buffer stands in for it).
SZ_HD_NOINLINE int clip_fixed_lds(void *lds_ws, int cap_nv, const i64 *ax, const i64 *ay, int na, const i64 *bx,
                                  const i64 *by, int nb, int op, RegionSink *sink) {
  FixedRing a = {ax, ay}, b = {bx, by};
  sink->nv = 0; sink->nr = 0;
  if (sink->ox != nullptr) sink->roff[0] = 0;
  SZ_ASSUME_LDS(lds_ws);
  return szv::clip_rings_lds(lds_ws, cap_nv, a, na, b, nb, op, szv::FILL_EVENODD, szv::FILL_EVENODD, *sink);
}

// ... for a wave whose 64 lanes all run the pair alike (k_narrow_wave): the sweep's walk over the active edges spreads
// over them (vatti.hpp WideLdsWave)
SZ_HD_NOINLINE int clip_fixed_lds_wave(void *lds_ws, int cap_nv, const i64 *ax, const i64 *ay, int na, const i64 *bx,
                                       const i64 *by, int nb, int op, RegionSink *sink) {
  FixedRing a = {ax, ay}, b = {bx, by};
  sink->nv = 0; sink->nr = 0;
  if (sink->ox != nullptr) sink->roff[0] = 0;
  SZ_ASSUME_LDS(lds_ws);
  if (cap_nv < na + nb) return szv::ST_WORKSPACE;
  return szv::clip_rings_t<szv::WideLdsWave>(lds_ws, szv::caps_for(cap_nv), a, na, b, nb, op, szv::FILL_EVENODD, szv::FILL_EVENODD, *sink);
}
// compact-record form (inlined: the workspace pointer then stays provably in LDS for the GPU compiler)
SZ_HD int clip_fixed_compact(void *ws, const i64 *ax, const i64 *ay, int na, const i64 *bx, const i64 *by, int nb,
                             int op, RegionSink *sink) {
  FixedRing a = {ax, ay}, b = {bx, by};
  sink->nv = 0; sink->nr = 0;
  if (sink->ox != nullptr) sink->roff[0] = 0;
  return szv::clip_rings_compact(ws, a, na, b, nb, op, *sink);
}
#ifndef SZC_THIRD_PROBE
#define SZC_THIRD_PROBE(tx, ty, nt, rx, ry, nr) ((void)0)
#endif
#ifndef SZC_PRUNE_OFF
#define SZC_PRUNE_OFF false           // (tests switch the pruning of crossings_long off to compare)
#endif
#ifndef SZC_PRUNE_PROBE
#define SZC_PRUNE_PROBE(skipped) ((void)0)
#endif
#ifndef SZC_PHASE
#define SZC_PHASE(k) ((void)0)        // the host test build attributes time to the parts of the contact model here
#endif
#ifndef SZC_FAST_PROBE
#define SZC_FAST_PROBE(st) ((void)0)   // the host test build counts fast-path hits and bail-outs here
#endif
// single-minimum fast path (convex.hpp) on the two rings held in the block in metres; ring 0 optionally shifted
// (floe_interactions.m:148-149).  ST_BAIL = not the plain case, nothing written.
template <class Blk>
SZ_HD_NOINLINE int clip_in_block(Blk *blk, void *deque_mem, int shifted, double sx, double sy, int na, int nb,
                                 RegionSink *sink) {
  sink->nv = 0; sink->nr = 0;
  if (sink->ox != nullptr) sink->roff[0] = 0;
  SZ_ASSUME_LDS(blk);   // the lane's block of the kernel's dynamic LDS (k_narrow_convex / k_narrow_convex_big)
  szx::FastWsT<Blk> fw = {blk, (Pt *)deque_mem, 24};
  szx::VertsMetres verts = {shifted, sx, sy};
  return szx::clip_monotone(fw, verts, na, nb, *sink);
}
// which sweep a pair uses: Wide records in the pair's HBM slice, or Compact records in a block of LDS
struct ClipWide {
  void *ws; int cap_nv;
  SZ_HD double *idle(int &bytes) const { bytes = 0; return nullptr; }
  SZ_HD int operator()(const i64 *ax, const i64 *ay, int na, const i64 *bx, const i64 *by, int nb, int op, RegionSink *sink) const {
    return clip_fixed(ws, cap_nv, ax, ay, na, bx, by, nb, op, sink);
  }
  SZ_HD bool fits(int nv) const { return nv <= cap_nv; }
};
// k_narrow_lds: each clip takes the lane's slice of LDS when its (trimmed) rings fit it, the pair's HBM slice
// otherwise -- or when the records of the clip outgrow the slice after all
template <bool WAVE>
struct ClipWideLdsT {
  void *host_ws; void *lds_ws; int lds_ws_bytes, cap_nv;    // HBM slice / the lane's slice of LDS and what of it the sweep may use; cap_nv: what the HBM slice holds
  // the sweep's part of the slice, free between two clips (device only)
  SZ_HD double *idle(int &bytes) const {
#if defined(__HIP_DEVICE_COMPILE__)
    bytes = lds_ws_bytes;
    return (double *)lds_ws;
#else
    bytes = 0; return nullptr;
#endif
  }
  SZ_HD int lds_cap() const {
    int c = (lds_ws_bytes - (int)szv::WS_PER_CLIP) / (int)szv::WS_PER_VERTEX;
#if !defined(__HIP_DEVICE_COMPILE__)
    if (c > cap_nv) c = cap_nv;             // (the host build has no LDS: the slice stands in for it)
#endif
    return c;
  }
  SZ_HD int operator()(const i64 *ax, const i64 *ay, int na, const i64 *bx, const i64 *by, int nb, int op, RegionSink *sink) const {
    const int lc = lds_cap();
    if (na + nb <= lc) {
      int st = WAVE ? clip_fixed_lds_wave(lds_ws, lc, ax, ay, na, bx, by, nb, op, sink)
                    : clip_fixed_lds(lds_ws, lc, ax, ay, na, bx, by, nb, op, sink);
      if (st != szv::ST_WORKSPACE || lc >= cap_nv) return st;
    }
    return clip_fixed(host_ws, cap_nv, ax, ay, na, bx, by, nb, op, sink);
  }
  SZ_HD bool fits(int nv) const { return nv <= cap_nv; }
};
typedef ClipWideLdsT<false> ClipWideLds;
typedef ClipWideLdsT<true> ClipWideLdsWave;
struct ClipCompact {
  void *ws;
  SZ_HD double *idle(int &bytes) const { bytes = 0; return nullptr; }
  SZ_HD int operator()(const i64 *ax, const i64 *ay, int na, const i64 *bx, const i64 *by, int nb, int op, RegionSink *sink) const {
    return clip_fixed_compact(ws, ax, ay, na, bx, by, nb, op, sink);
  }
  SZ_HD bool fits(int nv) const { return nv <= szv::COMPACT_MAX_NV; }
};
enum { ST_DEFER = 101,      // "redo this pair with the general sweep" (the fast path does not cover it)
       ST_DEFER_BIG = 102 }; // "... with the fast path on the larger block": the rings are only too long for this one

// ---- rings in metres: c_alpha + (Xi, Yi), optionally closed by one more vertex (:62-67) ----
struct RingD {
  const double *x, *y; // c_alpha of the floe (or the box)
  double ox, oy;       // floe position (0 for the box)
  int n;               // stored vertices
  int extra;           // 1 when the ring had to be closed (index n maps to vertex 0)
  SZ_HD int len() const { return n + extra; }
  SZ_HD double X(int k) const { return x[k >= n ? 0 : k] + ox; }
  SZ_HD double Y(int k) const { return y[k >= n ? 0 : k] + oy; }
};

// the same ring already summed (c_alpha + position) and held in a lane's block of LDS (the fast path)
struct RingL {
  const double *x, *y;
  int n, extra;
  SZ_HD int len() const { return n + extra; }
  SZ_HD double X(int k) const { return x[k >= n ? 0 : k]; }
  SZ_HD double Y(int k) const { return y[k >= n ? 0 : k]; }
};

struct Body { // floe_interactions.m reads these of floe1 / floe2
  double Xi, Yi, Ui, Vi, ksi_ice, h, area;
};

struct PairScratch {
  void *ws; int cap_nv;           // sweep workspace
  i64 *f1x, *f1y, *f2x, *f2y;     // rings in fixed point (cap n1 + 1, n2 + 1)
  i64 *fsx, *fsy;                 // ring 1 shifted by the unit force direction
  i64 *r1x, *r1y; int *r1off;     // regions of clip 1
  i64 *r2x, *r2y; int *r2off;     // regions of clip 2
  int cap_v, cap_r;               // capacity of each region slot
  double *px, *py; int cap_p;     // crossings P
  double *rows; int cap_rows;     // output: 5 doubles per contact row [Fx, Fy, Px, Py, overlap]
  int wave;                       // 1: all 64 lanes of the wave run this pair alike and the fixed-point rings are in LDS --
                                  // the conversion and the trimming then spread over the lanes (step_core.hpp narrow_t)
  // The overlap regions of a pair shared out over several waves (step_core.hpp narrow_t, k_narrow_wave): task `task_slot`
  // of `task_n` runs the pair like any other up to the loop over the regions, and there takes the region whose contact
  // row is number task_slot (the last task: that one and every later one) -- the second and third clip of a region
  // depend on the first clip alone, not on the regions before it.  task_n = 1: every region (the plain case).
  int task_slot, task_n;
};

struct PairOut {
  int nrows;        // contact rows written
  double sentinel;  // the `overlap` of the no-contact return: 0, +Inf or -Inf (:71-74); only when nrows == 0
  int has_sentinel; // 1 when the function returned the [0 0],[0 0],overlap form
};

// area(polyshape(X,Y)): |shoelace| relative to the first vertex
SZ_HD double region_area(const i64 *rx, const i64 *ry, int n) {
  if (n < 3) return 0;
  double x0 = from_fixed(rx[0]), y0 = from_fixed(ry[0]);
  double s = 0;
  for (int i = 0; i < n; ++i) {
    int k = (i + 1 == n) ? 0 : i + 1;
    double ax = from_fixed(rx[i]) - x0, ay = from_fixed(ry[i]) - y0;
    double bx = from_fixed(rx[k]) - x0, by = from_fixed(ry[k]) - y0;
    s += ax * by - bx * ay;
  }
  return absd(0.5 * s);
}
SZ_HD void region_centroid(const i64 *rx, const i64 *ry, int n, double &cx, double &cy) {
  double x0 = from_fixed(rx[0]), y0 = from_fixed(ry[0]);
  double s = 0, sx = 0, sy = 0;
  for (int i = 0; i < n; ++i) {
    int k = (i + 1 == n) ? 0 : i + 1;
    double ax = from_fixed(rx[i]) - x0, ay = from_fixed(ry[i]) - y0;
    double bx = from_fixed(rx[k]) - x0, by = from_fixed(ry[k]) - y0;
    double w = ax * by - bx * ay;
    s += w;
    sx += (ax + bx) * w;
    sy += (ay + by) * w;
  }
  cx = x0 + sx / (3.0 * s);
  cy = y0 + sy / (3.0 * s);
}
// polyarea.m
SZ_HD double region_polyarea(const i64 *rx, const i64 *ry, int n) {
  double s = 0;
  for (int i = 0; i < n; ++i) {
    int k = (i + 1 == n) ? 0 : i + 1;
    s += (from_fixed(rx[k]) - from_fixed(rx[i])) * (from_fixed(ry[k]) + from_fixed(ry[i]));
  }
  return absd(s / 2);
}

// inpolygon.m:150-224 for one point against the closed region ring (nr vertices + closing one)
SZ_HD bool in_region(double x, double y, const i64 *rx, const i64 *ry, int nr) {
  double mnx = from_fixed(rx[0]), mxx = mnx, mny = from_fixed(ry[0]), mxy = mny;
  for (int k = 1; k < nr; ++k) {
    double vx = from_fixed(rx[k]), vy = from_fixed(ry[k]);
    mnx = mind(mnx, vx); mxx = maxd(mxx, vx); mny = mind(mny, vy); mxy = maxd(mxy, vy);
  }
  if (!(x >= mnx && x <= mxx && y >= mny && y <= mxy)) return false;
  double sumdq = 0;
  bool on = false;
  for (int m = 0; m < nr; ++m) {
    int m1 = (m + 1 == nr) ? 0 : m + 1;
    double xa = from_fixed(rx[m]), ya = from_fixed(ry[m]), xb = from_fixed(rx[m1]), yb = from_fixed(ry[m1]);
    double avx = absd(0.5 * (xa + xb)), avy = absd(0.5 * (ya + yb));
    double sf = maxd(avx, avy);
    sf = maxd(sf, avx * avy);
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
  return (sumdq != 0) || on;
}

// Is (x, y) inside the region with a clearance of more than `margin` metres from its boundary?  Used to
// certify, without running the sweep, that two regions overlap: if one point keeps a 1e-6 m clearance inside
// both, their intersection contains a disc of ~4300 grid units radius, which no rounding of the sweep (<= 1
// unit per vertex) can make vanish -- the reference's third clip (floe_interactions.m:158-159) is then
// certainly non-empty.  A `false` only means "not certified": the caller runs the clip.
SZ_HD bool clearly_inside(double x, double y, const i64 *rx, const i64 *ry, int nr, double margin) {
  if (nr < 3 || !in_region(x, y, rx, ry, nr)) return false;
  for (int m = 0; m < nr; ++m) {
    int m1 = (m + 1 == nr) ? 0 : m + 1;
    double xa = from_fixed(rx[m]), ya = from_fixed(ry[m]), xb = from_fixed(rx[m1]), yb = from_fixed(ry[m1]);
    double ex = xb - xa, ey = yb - ya;
    double px = x - xa, py = y - ya;
    double ee = ex * ex + ey * ey;
    double t = (ee > 0) ? (px * ex + py * ey) / ee : 0;
    if (t < 0) t = 0;
    if (t > 1) t = 1;
    double dx = px - t * ex, dy = py - t * ey;
    if (!(dx * dx + dy * dy > margin * margin)) return false;
  }
  return true;
}

// |p_poly_dist| of one point against the closed ring c (p_poly_dist.m:143-281)
template <class Ring>
SZ_HD double dist_to_ring(double xp, double yp, const Ring &c) {
  int nv = c.len();
  double dpv_min = SZC_INF; int I_dpv = 0;
  for (int k = 0; k < nv; ++k) {
    double d = hyp(c.X(k) - xp, c.Y(k) - yp);
    if (d < dpv_min) { dpv_min = d; I_dpv = k; }
  }
  double cr_min = 0; int I_cr = 0; bool have = false;
  for (int k = 0; k + 1 < nv; ++k) {
    double x0 = c.X(k), y0 = c.Y(k);
    double dvx = c.X(k + 1) - x0, dvy = c.Y(k + 1) - y0;
    double vds = hyp(dvx, dvy);
    double ct = dvx / vds, st = dvy / vds;
    double p1r1 = ct * x0 + st * y0;
    double p1r2 = -st * x0 + ct * y0;
    double r = (xp * ct + yp * st) - p1r1;
    double cr = (xp * (-st) + yp * ct) - p1r2;
    if (r > 0 && r < vds) {
      double a = absd(cr);
      if (!have || a < cr_min) { cr_min = a; I_cr = k; have = true; }
    }
  }
  bool cond1 = !have;
  bool cond2 = have && (I_cr != I_dpv) && ((cr_min - dpv_min) > 0);
  return (cond1 || cond2) ? dpv_min : cr_min;
}

// dist_to_ring(xp, yp, c) < 1e-8 -- all the contact model asks of that distance (floe_interactions.m:127) -- for
// long rings, from the vertices and edges within 1e-3 m of the point alone (a vertex is "near" when both coordinate
// differences are within 1e-3 m, an edge when the point lies in its bounding box grown by 1e-3 m).  The elements
// left out have |distance| > 5e-4 m by a wide margin over the rounding of these expressions (1e-10 m at 1e5 m
// coordinates, 1e-6 m at the 1e9 m limit of the fixed-point range), so:
//   * a minimum below 5e-4 m -- of the vertex distances, of the edge distances -- is attained by near elements
//     only, at the same (first) index, and is found with the same arithmetic;
//   * a minimum that is not below 5e-4 m cannot make the result smaller than 1e-8 whichever branch of
//     p_poly_dist.m:266-281 selects it, and the one branch that compares indices (I_cr ~= I_dpv) cannot be
//     flipped by a far edge: the edge that starts at a vertex within 5e-4 m is near itself.
// Hence the same truth value as dist_to_ring(...) < 1e-8, without the square root and the two divisions per far
// edge.  NaN coordinates compare false everywhere, in both forms (nothing near: not closer).
SZ_HD bool closer_than_1e8(double xp, double yp, const RingD &c) {
  const int nv = c.len();
  if (SZC_PRUNE_OFF || nv < 24) return dist_to_ring(xp, yp, c) < 1e-8;
  const double near = 1e-3;
  double dpv_min = SZC_INF; int I_dpv = 0;
  double cr_min = 0; int I_cr = 0; bool have = false;
  double x1 = c.X(0), y1 = c.Y(0);
  for (int k = 0; k < nv; ++k) {
    const double x0 = x1, y0 = y1;
    if (absd(x0 - xp) <= near && absd(y0 - yp) <= near) {
      double d = hyp(x0 - xp, y0 - yp);
      if (d < dpv_min) { dpv_min = d; I_dpv = k; }
    }
    if (k + 1 >= nv) break;
    x1 = c.X(k + 1); y1 = c.Y(k + 1);
    if (!(xp >= mind(x0, x1) - near && xp <= maxd(x0, x1) + near && yp >= mind(y0, y1) - near && yp <= maxd(y0, y1) + near)) continue;
    double dvx = x1 - x0, dvy = y1 - y0;
    double vds = hyp(dvx, dvy);
    double ct = dvx / vds, st = dvy / vds;
    double p1r1 = ct * x0 + st * y0;
    double p1r2 = -st * x0 + ct * y0;
    double r = (xp * ct + yp * st) - p1r1;
    double cr = (xp * (-st) + yp * ct) - p1r2;
    if (r > 0 && r < vds) {
      double a = absd(cr);
      if (!have || a < cr_min) { cr_min = a; I_cr = k; have = true; }
    }
  }
  bool cond1 = !have;
  bool cond2 = have && (I_cr != I_dpv) && ((cr_min - dpv_min) > 0);
  return ((cond1 || cond2) ? dpv_min : cr_min) < 1e-8;
}
SZ_HD bool closer_than_1e8(double xp, double yp, const RingL &c) { return dist_to_ring(xp, yp, c) < 1e-8; }

// InterX.m:54-81.  Returns the number of unique crossings (sorted by x, then y), or -1 on overflow.
// The segment pairs are visited ring-1-major with the cross products of the ring-2 vertices carried from one
// segment to the next (every product is the expression of InterX, evaluated once instead of twice); the result
// is a sorted set, so the order of discovery does not matter.
template <class Ring>
SZ_HD int crossings(const Ring &c1, const Ring &c2, double *px, double *py, int cap) {
  int m1 = c1.len() - 1, m2 = c2.len() - 1;
  int np = 0;
  if (m1 <= 0 || m2 <= 0) return 0;
  double x1b = c1.X(0), y1b = c1.Y(0);
  for (int i = 0; i < m1; ++i) {
    const double x1a = x1b, y1a = y1b;
    x1b = c1.X(i + 1); y1b = c1.Y(i + 1);
    const double dx1 = x1b - x1a, dy1 = y1b - y1a;
    const double S1 = dx1 * y1a - dy1 * x1a;
    double x2b = c2.X(0), y2b = c2.Y(0);
    double b = dx1 * y2b - dy1 * x2b;
    for (int j = 0; j < m2; ++j) {
      const double x2a = x2b, y2a = y2b, a = b;
      x2b = c2.X(j + 1); y2b = c2.Y(j + 1);
      b = dx1 * y2b - dy1 * x2b;
      if (!(((a - S1) * (b - S1)) <= 0)) continue;
      const double dx2 = x2b - x2a, dy2 = y2b - y2a;
      const double S2 = dx2 * y2a - dy2 * x2a;
      double c = y1a * dx2 - x1a * dy2;
      double d = y1b * dx2 - x1b * dy2;
      if (!(((c - S2) * (d - S2)) <= 0)) continue;
      double L = dy2 * dx1 - dy1 * dx2;
      if (L == 0) continue;
      double X = (dx2 * S1 - dx1 * S2) / L;
      double Y = (dy2 * S1 - dy1 * S2) / L;
      // insertion into the sorted unique list (unique(...,'rows'))
      int k = np;
      while (k > 0 && (px[k - 1] > X || (px[k - 1] == X && py[k - 1] > Y))) --k;
      if (k > 0 && px[k - 1] == X && py[k - 1] == Y) continue;
      if (np >= cap) return -1;
      for (int q = np; q > k; --q) { px[q] = px[q - 1]; py[q] = py[q - 1]; }
      px[k] = X; py[k] = Y;
      ++np;
    }
  }
  return np;
}

// insertion into the sorted unique list (unique(...,'rows'), InterX.m:76); -1 when the list is full
SZ_HD_NOINLINE int crossing_insert(double *px, double *py, int np, int cap, double X, double Y) {
  int k = np;
  while (k > 0 && (px[k - 1] > X || (px[k - 1] == X && py[k - 1] > Y))) --k;
  if (k > 0 && px[k - 1] == X && py[k - 1] == Y) return np;
  if (np >= cap) return -1;
  for (int q = np; q > k; --q) { px[q] = px[q - 1]; py[q] = py[q - 1]; }
  px[k] = X; py[k] = Y;
  return np + 1;
}
// The same crossings for long rings (the general path: evolved floe outlines of 50-600 vertices), where the dense
// n1 x n2 sign test of InterX is most of a pair's time.  Ring 2 is laid out once (closed, position added) together
// with the bounding box of every run of CHUNK edges; for an edge of ring 1 a run is skipped when the first sign
// test of InterX.m:60-64 is certain to fail for each of its edges: the value a = dx1*y - dy1*x that the test
// compares with S1 is linear in the vertex, so over the run it lies between its values at two corners of the
// run's box, and the floating-point a of every vertex within 3u(|dx1||y| + |dy1||x|) of the exact one (u = 2^-53;
// two products, one difference).  With S1 outside that range by more than the margin every (a - S1) of the run
// has the same sign and is not zero, every product is positive, and the reference's test fails: the edges
// skipped are edges InterX rejects, the rest go through the unchanged arithmetic in the unchanged order.
// (margin > 1e-100 keeps the products away from underflow, where a positive product could round to zero.)
// x2 must hold len + 4 * (len / CROSS_CHUNK + 1) doubles, y2 len doubles (len = closed length of ring 2).
enum { CROSS_CHUNK = 8, CROSS_PRUNE_MIN = 256 };
template <class Ring>
SZ_HD int crossings_long(const Ring &c1, const Ring &c2, double *px, double *py, int cap, double *x2, double *y2) {
  const int m1 = c1.len() - 1, m2 = c2.len() - 1;
  int np = 0;
  if (m1 <= 0 || m2 <= 0) return 0;
  for (int k = 0; k <= m2; ++k) { x2[k] = c2.X(k); y2[k] = c2.Y(k); }
  double *box = x2 + (m2 + 1);
  const int nch = (m2 + CROSS_CHUNK - 1) / CROSS_CHUNK;
  for (int c = 0; c < nch; ++c) {
    const int j0 = c * CROSS_CHUNK, j1 = (j0 + CROSS_CHUNK < m2) ? j0 + CROSS_CHUNK : m2;
    double xl = x2[j0], xh = xl, yl = y2[j0], yh = yl;
    for (int k = j0 + 1; k <= j1; ++k) { xl = mind(xl, x2[k]); xh = maxd(xh, x2[k]); yl = mind(yl, y2[k]); yh = maxd(yh, y2[k]); }
    box[4 * c] = xl; box[4 * c + 1] = xh; box[4 * c + 2] = yl; box[4 * c + 3] = yh;
  }
  double x1b = c1.X(0), y1b = c1.Y(0);
  for (int i = 0; i < m1; ++i) {
    const double x1a = x1b, y1a = y1b;
    x1b = c1.X(i + 1); y1b = c1.Y(i + 1);
    const double dx1 = x1b - x1a, dy1 = y1b - y1a;
    const double S1 = dx1 * y1a - dy1 * x1a;
    const double adx = absd(dx1), ady = absd(dy1);
    // 64 runs at a time: first which of them survive (no branches: the loads of the boxes overlap), then those
    for (int cb = 0; cb < nch; cb += 64) {
      const int ce = (cb + 64 < nch) ? cb + 64 : nch;
      unsigned long long keep = 0;
#pragma unroll 4
      for (int c = cb; c < ce; ++c) {
        const double xl = box[4 * c], xh = box[4 * c + 1], yl = box[4 * c + 2], yh = box[4 * c + 3];
        const double lo = dx1 * (dx1 >= 0 ? yl : yh) - dy1 * (dy1 >= 0 ? xh : xl);
        const double hi = dx1 * (dx1 >= 0 ? yh : yl) - dy1 * (dy1 >= 0 ? xl : xh);
        const double margin = 8.9e-16 * (adx * maxd(absd(yl), absd(yh)) + ady * maxd(absd(xl), absd(xh)));
        const bool skip = margin > 1e-100 && (lo - margin > S1 || hi + margin < S1);
        SZC_PRUNE_PROBE(skip ? 1 : 0);
        keep |= (unsigned long long)(skip ? 0 : 1) << (c - cb);
      }
      while (keep) {
        const int c = cb + __builtin_ctzll(keep);
        keep &= keep - 1;
        const int j0 = c * CROSS_CHUNK, left = m2 - j0;     // edges j0 .. j0 + min(CHUNK, left) - 1
        double vx[CROSS_CHUNK + 1], vy[CROSS_CHUNK + 1];      // their vertices, all loads issued at once
#pragma unroll
        for (int k = 0; k <= CROSS_CHUNK; ++k) { const int q = j0 + (k < left ? k : left); vx[k] = x2[q]; vy[k] = y2[q]; }
        double b = dx1 * vy[0] - dy1 * vx[0];
#pragma unroll
        for (int k = 0; k < CROSS_CHUNK; ++k) {
          if (k >= left) break;
          const double x2a = vx[k], y2a = vy[k], a = b;
          const double x2b = vx[k + 1], y2b = vy[k + 1];
          b = dx1 * y2b - dy1 * x2b;
          if (!(((a - S1) * (b - S1)) <= 0)) continue;
          const double dx2 = x2b - x2a, dy2 = y2b - y2a;
          const double S2 = dx2 * y2a - dy2 * x2a;
          const double cc = y1a * dx2 - x1a * dy2;
          const double d = y1b * dx2 - x1b * dy2;
          if (!(((cc - S2) * (d - S2)) <= 0)) continue;
          const double L = dy2 * dx1 - dy1 * dx2;
          if (L == 0) continue;
          const double X = (dx2 * S1 - dx1 * S2) / L;
          const double Y = (dy2 * S1 - dy1 * S2) / L;
          np = crossing_insert(px, py, np, cap, X, Y);
          if (np < 0) return -1;
        }
      }
    }
  }
  return np;
}
#if defined(__HIPCC__)
// The same set found by the 64 lanes of a wave that all hold the same pair (step_core.hpp narrow_t, WAVE; the rings and
// px / py in LDS): lane l takes the edges l, l + 64, ... of ring 1 against every edge of ring 2 -- the arithmetic of
// each (i, j) is InterX's, unchanged -- and the lanes that found a point take turns to insert it, every lane
// performing the same insertion.  The result is a sorted set: the order of discovery does not matter.
template <class Ring>
__device__ inline int crossings_wave(const Ring &c1, const Ring &c2, double *px, double *py, int cap) {
  const int m1 = c1.len() - 1, m2 = c2.len() - 1;
  int np = 0;
  if (m1 <= 0 || m2 <= 0) return 0;
  const int lane = (int)(threadIdx.x & 63);
  for (int ib = 0; ib < m1; ib += 64) {
    const int i = ib + lane;
    const bool live = i < m1;
    const int ii = live ? i : 0;
    const double x1a = c1.X(ii), y1a = c1.Y(ii), x1b = c1.X(ii + 1), y1b = c1.Y(ii + 1);
    const double dx1 = x1b - x1a, dy1 = y1b - y1a;
    const double S1 = dx1 * y1a - dy1 * x1a;
    double x2b = c2.X(0), y2b = c2.Y(0);
    double b = dx1 * y2b - dy1 * x2b;
    for (int j = 0; j < m2; ++j) {
      const double x2a = x2b, y2a = y2b, a = b;
      x2b = c2.X(j + 1); y2b = c2.Y(j + 1);
      b = dx1 * y2b - dy1 * x2b;
      bool hit = live && (((a - S1) * (b - S1)) <= 0);
      double X = 0, Y = 0;
      if (hit) {
        const double dx2 = x2b - x2a, dy2 = y2b - y2a;
        const double S2 = dx2 * y2a - dy2 * x2a;
        const double c = y1a * dx2 - x1a * dy2;
        const double d = y1b * dx2 - x1b * dy2;
        const double L = dy2 * dx1 - dy1 * dx2;
        hit = (((c - S2) * (d - S2)) <= 0) && L != 0;
        if (hit) { X = (dx2 * S1 - dx1 * S2) / L; Y = (dy2 * S1 - dy1 * S2) / L; }
      }
      unsigned long long m = __ballot(hit);
      while (m) {
        const int l = __builtin_ctzll(m);
        m &= m - 1;
        const double Xl = __shfl(X, l), Yl = __shfl(Y, l);
        np = crossing_insert(px, py, np, cap, Xl, Yl);
        if (np < 0) return -1;
      }
    }
  }
  return np;
}
#endif
// (rings held in LDS by the fast path are short: nothing to prune)
SZ_HD int crossings_any(const RingL &c1, const RingL &c2, double *px, double *py, int cap, double *, double *) {
  return crossings(c1, c2, px, py, cap);
}
SZ_HD int crossings_any(const RingD &c1, const RingD &c2, double *px, double *py, int cap, double *x2, double *y2) {
  if (SZC_PRUNE_OFF || (c1.len() - 1) * (c2.len() - 1) < (int)CROSS_PRUNE_MIN) return crossings(c1, c2, px, py, cap);
  return crossings_long(c1, c2, px, py, cap, x2, y2);
}

// How the three clips of the contact model are run.
#ifndef SZC_TRIM_OFF
#define SZC_TRIM_OFF false
#endif

#ifndef SZC_TRIM_PROBE
#define SZC_TRIM_PROBE(before, after) ((void)0)
#endif
// ViaArrays: the rings go through int64 arrays of the pair's scratch slice into a sweep (Wide or Compact records).
template <class Sweep>
struct ViaArrays {
  Sweep sweep;
  SZ_HD double *idle(int &bytes) const { return sweep.idle(bytes); }
  template <class Ring>
  SZ_HD int first(const Ring &ring1, const Ring &ring2, bool boundary, PairScratch &S, RegionSink *sink) const {   // :29 / :34
    int n1 = ring1.n, n2 = ring2.n;
    SZC_PHASE(10);
    fill(ring1, n1, false, 0.0, 0.0, S.f1x, S.f1y, S.wave);
    fill(ring2, n2, false, 0.0, 0.0, S.f2x, S.f2y, S.wave);
    SZC_PHASE(11);
    // an intersection sees only what lies in the box the two rings share (trim.hpp); on a tie the rings go in whole
    if (!boundary && !SZC_TRIM_OFF && !trim(S.f1x, S.f1y, n1, S.f2x, S.f2y, n2, S.wave)) {
      n1 = ring1.n; n2 = ring2.n;
      fill(ring1, n1, false, 0.0, 0.0, S.f1x, S.f1y, S.wave);
      fill(ring2, n2, false, 0.0, 0.0, S.f2x, S.f2y, S.wave);
    }
    SZC_TRIM_PROBE(ring1.n + ring2.n, n1 + n2);
    SZC_PHASE(1);
    return sweep(S.f1x, S.f1y, n1, S.f2x, S.f2y, n2, boundary ? szv::OP_DIFFERENCE : szv::OP_INTERSECTION, sink);
  }
  SZ_HD static bool trim(i64 *ax, i64 *ay, int &na, i64 *bx, i64 *by, int &nb, int wave) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (wave) return szt::trim_pair_wave(ax, ay, na, bx, by, nb);
#endif
    return szt::trim_pair(ax, ay, na, bx, by, nb);
  }
  template <class Ring>
  SZ_HD static void fill(const Ring &r, int n, bool shifted, double dx, double dy, i64 *fx, i64 *fy, int wave = 0) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (wave) {   // a vertex per lane (the conversion is ~100 instructions of FP64 / int64 emulation a coordinate)
      for (int k = (int)(threadIdx.x & 63); k < n; k += 64) {
        if (!shifted) { fx[k] = to_fixed(r.X(k)); fy[k] = to_fixed(r.Y(k)); }
        else { fx[k] = to_fixed(r.X(k) + dx); fy[k] = to_fixed(r.Y(k) + dy); }
      }
      __builtin_amdgcn_wave_barrier();
      return;
    }
#endif
#pragma unroll 4
    for (int k = 0; k < n; ++k) {
      if (!shifted) { fx[k] = to_fixed(r.X(k)); fy[k] = to_fixed(r.Y(k)); }
      else { fx[k] = to_fixed(r.X(k) + dx); fy[k] = to_fixed(r.Y(k) + dy); }
    }
  }
  // :140-155 ring 1 (closed) moved by the unit force direction, against ring 2 (closed)
  template <class Ring>
  SZ_HD int second(const Ring &c1, bool shifted, double fdx, double fdy, const Ring &c2, int op, PairScratch &S, RegionSink *sink) const {
    int nc1 = c1.len(), nc2 = c2.len();
    // (both rings closed here, :62-70; ring 2 is converted again: clip 1 may have trimmed its copy)
    SZC_PHASE(10);
    fill(c1, nc1, shifted, fdx, fdy, S.fsx, S.fsy, S.wave);
    fill(c2, nc2, false, 0.0, 0.0, S.f2x, S.f2y, S.wave);
    SZC_PHASE(11);
    if (op == szv::OP_INTERSECTION && !SZC_TRIM_OFF && !trim(S.fsx, S.fsy, nc1, S.f2x, S.f2y, nc2, S.wave)) {
      nc1 = c1.len(); nc2 = c2.len();
      fill(c1, nc1, shifted, fdx, fdy, S.fsx, S.fsy, S.wave);
      fill(c2, nc2, false, 0.0, 0.0, S.f2x, S.f2y, S.wave);
    }
    SZC_TRIM_PROBE(c1.len() + c2.len(), nc1 + nc2);
    SZC_PHASE(6);
    return sweep(S.fsx, S.fsy, nc1, S.f2x, S.f2y, nc2, op, sink);
  }
  SZ_HD int third(const i64 *tx, const i64 *ty, int nt, const i64 *rx, const i64 *ry, int nr, RegionSink *sink) const {   // :158
    if (!sweep.fits(nt + nr)) return szv::ST_WORKSPACE;
    return sweep(tx, ty, nt, rx, ry, nr, szv::OP_INTERSECTION, sink);
  }
};
// InBlock: the fast path alone, on the rings the caller has put into the lane's block (RingL views of it).
// Anything it does not cover -- a bail-out, the difference with the domain box, a third clip whose outcome the
// overlap certificate does not settle -- comes back as ST_DEFER and the whole pair is redone by the general
// kernel, so a wave never mixes the short fast path with the long general sweep.  The 768-byte output deque
// borrows the start of the (idle) Wide workspace of the pair's slice.
template <class Blk>
struct InBlock {
  Blk *blk;
  void *deque_mem;
  SZ_HD double *idle(int &bytes) const { bytes = 0; return nullptr; }
  SZ_HD int run(int shifted, double sx, double sy, int na, int nb, RegionSink *sink) const {
    int st = clip_in_block(blk, deque_mem, shifted, sx, sy, na, nb, sink);
    SZC_FAST_PROBE(st);
    return st == szx::ST_BAIL ? (int)ST_DEFER : st;
  }
  SZ_HD int first(const RingL &ring1, const RingL &ring2, bool boundary, PairScratch &, RegionSink *sink) const {
    if (boundary) return ST_DEFER;
    return run(0, 0.0, 0.0, ring1.n, ring2.n, sink);
  }
  // (the closing vertices the reference appends are duplicates, which the sweep strips: the stored rings do)
  SZ_HD int second(const RingL &c1, bool shifted, double fdx, double fdy, const RingL &c2, int op, PairScratch &, RegionSink *sink) const {
    if (op != szv::OP_INTERSECTION) return ST_DEFER;
    return run(shifted ? 1 : 0, fdx, fdy, c1.n, c2.n, sink);
  }
  SZ_HD int third(const i64 *tx, const i64 *ty, int nt, const i64 *rx, const i64 *ry, int nr, RegionSink *) const {
    SZC_THIRD_PROBE(tx, ty, nt, rx, ry, nr);
    return ST_DEFER;
  }
};

// The contact model for one pair, in two parts so that candidates whose rings do not overlap stop after the
// first clip (and free their lane): pair_clip1 -> regions in S.r1*, pair_finish -> contact rows.
// Both return a szv::Status (ST_WORKSPACE / ST_OUT_CAPACITY ask for a larger tier; ST_CLIP_ERROR is the
// reference's "Clipper Error.").
template <class Ring, class Clip>
SZ_HD int pair_clip1(const Ring &ring1, const Ring &ring2, bool boundary, PairScratch &S, const Clip &clip, int &nreg) {
  RegionSink s1 = {S.r1x, S.r1y, S.r1off, S.cap_v, S.cap_r, 0, 0};
  SZC_PHASE(1);
  int st = clip.first(ring1, ring2, boundary, S, &s1);                                 // :29 / :34
  SZC_PHASE(2);
  nreg = (st == szv::ST_OK) ? s1.nr : 0;
  return st;
}

template <class Ring, class Clip>
SZ_HD int pair_finish(const Ring &ring1, const Body &f1, const Ring &ring2, const Body &f2, bool boundary,
                      const double *c2b, bool periodic, double Modulus, double dt, double mu, double nu,
                      PairScratch &S, int nreg, const Clip &clip, PairOut &out) {
  out.nrows = 0; out.sentinel = 0; out.has_sentinel = 0;
  // :10-21
  double h1 = f1.h, h2 = f2.h;
  double r1 = ::sqrt(f1.area), r2 = ::sqrt(f2.area);
  double Force_factor = Modulus * (h1 * h2) / (h1 * r2 + h2 * r1);
  double overlap0 = 0;
  if (boundary) Force_factor = Modulus * h1 / r1;
  else if (r1 > 1e5 || r2 > 1e5) {
    r1 = mind(r1, r2);
    h1 = mind(h1, h2);
    Force_factor = Modulus * h1 / r1;
  }
  double G = Modulus / (2 * (1 + nu));
  const int n1 = ring1.n, n2 = ring2.n;
  const int op = boundary ? szv::OP_DIFFERENCE : szv::OP_INTERSECTION;
  int st;
  if (boundary && nreg > 0) {                                                          // :35-40
    if (region_polyarea(S.r1x, S.r1y, S.r1off[1]) / f1.area > 0.75) overlap0 = SZC_INF;
  }
  // :54-60 merge flags (sum of the region areas, :43-51)
  {
    double mx1 = ring1.X(0), mn1 = mx1, my1 = ring1.Y(0), mny1 = my1;
    for (int k = 1; k < n1; ++k) {
      double vx = ring1.X(k), vy = ring1.Y(k);
      mx1 = maxd(mx1, vx); mn1 = mind(mn1, vx); my1 = maxd(my1, vy); mny1 = mind(mny1, vy);
    }
    double bxmax = c2b[0], bxmin = c2b[0], bymax = c2b[1], bymin = c2b[1];
    for (int k = 1; k < 5; ++k) {
      bxmax = maxd(bxmax, c2b[2 * k]); bxmin = mind(bxmin, c2b[2 * k]);
      bymax = maxd(bymax, c2b[2 * k + 1]); bymin = mind(bymin, c2b[2 * k + 1]);
    }
    // area(polyshape(c2_boundary')): shoelace of the 4 corners relative to the first
    double barea;
    {
      double s = 0;
      for (int i = 0; i < 4; ++i) {
        int k = (i + 1 == 4) ? 0 : i + 1;
        double ax = c2b[2 * i] - c2b[0], ay = c2b[2 * i + 1] - c2b[1];
        double bx = c2b[2 * k] - c2b[0], by = c2b[2 * k + 1] - c2b[1];
        s += ax * by - bx * ay;
      }
      barea = absd(0.5 * s);
    }
    bool inside = (mx1 < bxmax && mn1 > bxmin && my1 < bymax && mny1 > bymin);
    if (inside || f2.area < 0.95 * barea || periodic) {
      double sumAr = 0;
      if (nreg == 0) sumAr += 0;
      for (int k = 0; k < nreg; ++k)
        sumAr += region_area(S.r1x + S.r1off[k], S.r1y + S.r1off[k], S.r1off[k + 1] - S.r1off[k]);
      if (sumAr / f1.area > 0.55) overlap0 = SZC_INF;
      else if (sumAr / f2.area > 0.55) overlap0 = -SZC_INF;
    }
  }
  // :62-67 close the rings
  Ring c1 = ring1, c2 = ring2;
  if (hyp(c1.X(0) - c1.X(n1 - 1), c1.Y(0) - c1.Y(n1 - 1)) > 1) c1.extra = 1;
  if (hyp(c2.X(0) - c2.X(n2 - 1), c2.Y(0) - c2.Y(n2 - 1)) > 1) c2.extra = 1;

  SZC_PHASE(3);
  // (ring 2 and its chunk boxes are laid out in the sweep's idle LDS when the kernel has it, else in the region
  // slots of clip 2, idle until :140)
  double *lay_x = (double *)S.r2x, *lay_y = (double *)S.r2y;
  {
    int room = 0;
    double *blk = clip.idle(room);
    const int len2 = c2.len(), nx = len2 + 4 * (len2 / (int)CROSS_CHUNK + 1);
    if (blk && 8 * (nx + len2) <= room) { lay_x = blk; lay_y = blk + nx; }
  }
#if defined(__HIP_DEVICE_COMPILE__)
  int np = S.wave ? crossings_wave(c1, c2, S.px, S.py, S.cap_p) : crossings_any(c1, c2, S.px, S.py, S.cap_p, lay_x, lay_y);   // :70
#else
  int np = crossings_any(c1, c2, S.px, S.py, S.cap_p, lay_x, lay_y);                   // :70
#endif
  SZC_PHASE(4);
  if (np < 0) return szv::ST_WORKSPACE;
#if defined(SZ_ABLATE) && SZ_ABLATE == 3
  out.has_sentinel = 1; out.sentinel = (double)np * 0; return szv::ST_OK;   // diagnostic build: stop after InterX
#endif
  if (np < 2 || is_inf(overlap0) || nreg == 0) {                                       // :71-74
    out.has_sentinel = 1;
    out.sentinel = overlap0;
    return szv::ST_OK;
  }
  int N1 = c1.len() - 1, N2 = c2.len() - 1;                                            // :78-79
  double Amin = (double)(N1 < N2 ? N1 : N2) * 100 / 1.75;
  for (int k = 0; k < nreg; ++k) {
    const i64 *RX = S.r1x + S.r1off[k], *RY = S.r1y + S.r1off[k];
    const int nr = S.r1off[k + 1] - S.r1off[k];
    double Ar = region_area(RX, RY, nr);
    if (Ar < Amin) continue;                                                           // :83
    if (S.task_n > 1) {       // (another wave's region: its row is counted, not made)
      const int r = out.nrows;
      const bool mine = r == S.task_slot || (S.task_slot == S.task_n - 1 && r >= S.task_n - 1);
      if (!mine) {
        if (out.nrows >= S.cap_rows) return szv::ST_OUT_CAPACITY;
        ++out.nrows;
        continue;
      }
    }
    double cx, cy;
    region_centroid(RX, RY, nr, cx, cy);                                               // :96-97
    // :98-100 nearest region vertex of each crossing; keep those closer than 1 m
    int m = 0;
    double p0x = 0, p0y = 0, p1x = 0, p1y = 0;
    for (int q = 0; q < np; ++q) {
      double best = SZC_INF; int bi = 0;
      for (int v = 0; v < nr; ++v) {
        double ddx = from_fixed(RX[v]) - S.px[q], ddy = from_fixed(RY[v]) - S.py[q];
        double d2 = ddx * ddx + ddy * ddy;
        if (d2 < best) { best = d2; bi = v; }
      }
      if (::sqrt(best) < 1) {
        if (m == 0) { p0x = from_fixed(RX[bi]); p0y = from_fixed(RY[bi]); }
        else if (m == 1) { p1x = from_fixed(RX[bi]); p1y = from_fixed(RY[bi]); }
        ++m;
      }
    }
    double fdx = 0, fdy = 0, dl = 0, pcx = cx, pcy = cy;
    if (Ar == 0) { fdx = fdy = 0; pcx = pcy = 0; dl = 0; }                             // :103-106
    else if (m == 2) {                                                                 // :107-112
      double xgh = p1x - p0x, ygh = p1y - p0y;
      double b = ::sqrt(xgh * xgh + ygh * ygh);
      fdx = -ygh / b; fdy = xgh / b;
      dl = b;
    } else if (m == 0) { fdx = fdy = 0; dl = 0; }                                      // :113-116
    else {                                                                             // :117-137
      SZC_PHASE(5);
      int non = 0;
      double sfx = 0, sfy = 0, sb = 0;
#if defined(__HIP_DEVICE_COMPILE__)
      if (S.wave) {
        // (a wave on the pair: an edge of the region per lane for what costs -- the inside test against the region, the
        // distance to ring 1, O(nr) and O(n1) each -- then the three sums edge by edge in the order of the loop below,
        // by every lane alike from the lanes' values: the same additions in the same order)
        const int lane = (int)(threadIdx.x & 63);
        for (int e0 = 0; e0 < nr; e0 += 64) {
          const int e = e0 + lane;
          double b = 0, nx = 0, ny = 0;
          bool close = false;
          if (e < nr) {
            const int e1 = (e + 1 == nr) ? 0 : e + 1;
            const double xa = from_fixed(RX[e]), ya = from_fixed(RY[e]), xb = from_fixed(RX[e1]), yb = from_fixed(RY[e1]);
            const double xgh = xb - xa, ygh = yb - ya;
            const double xm = (xb + xa) / 2, ym = (yb + ya) / 2;
            b = ::sqrt(xgh * xgh + ygh * ygh);
            nx = -ygh / b; ny = xgh / b;
            const double xt = xm + nx / 100, yt = ym + ny / 100;
            if (!in_region(xt, yt, RX, RY, nr)) { nx = -nx; ny = -ny; }
            close = closer_than_1e8(xm, ym, c1);
          }
          const unsigned long long mc = __ballot(close);
          const int cnt = nr - e0 < 64 ? nr - e0 : 64;
          for (int k = 0; k < cnt; ++k) {
            if (!((mc >> k) & 1ULL)) continue;
            const double bk = __shfl(b, k), nxk = __shfl(nx, k), nyk = __shfl(ny, k);
            ++non;
            sfx += (-Force_factor * bk) * nxk;
            sfy += (-Force_factor * bk) * nyk;
            sb += bk;
          }
        }
      } else
#endif
      for (int e = 0; e < nr; ++e) {
        int e1 = (e + 1 == nr) ? 0 : e + 1;
        double xa = from_fixed(RX[e]), ya = from_fixed(RY[e]), xb = from_fixed(RX[e1]), yb = from_fixed(RY[e1]);
        double xgh = xb - xa, ygh = yb - ya;
        double xm = (xb + xa) / 2, ym = (yb + ya) / 2;
        double b = ::sqrt(xgh * xgh + ygh * ygh);
        double nx = -ygh / b, ny = xgh / b;
        double xt = xm + nx / 100, yt = ym + ny / 100;
        if (!in_region(xt, yt, RX, RY, nr)) { nx = -nx; ny = -ny; }
        if (closer_than_1e8(xm, ym, c1)) {                                            // p_poly_dist(...) < 1e-8, :127
          ++non;
          sfx += (-Force_factor * b) * nx;
          sfy += (-Force_factor * b) * ny;
          sb += b;
        }
      }
      if (non < nr && non > 0) {
        double nrm = ::sqrt(sfx * sfx + sfy * sfy);
        fdx = sfx / nrm; fdy = sfy / nrm;
        dl = sb / non;
      } else { fdx = fdy = 0; dl = 0; }
    }
    // :140-155 shift ring 1 by the unit direction and clip again
    const bool shifted = !(dl < 0.1);
    if (!shifted) { fdx = 0; fdy = 0; }
#if defined(SZ_ABLATE) && SZ_ABLATE == 4
    { if (out.nrows >= S.cap_rows) return szv::ST_OUT_CAPACITY; double *row = S.rows + 5 * out.nrows; row[0] = fdx; row[1] = fdy; row[2] = pcx; row[3] = pcy; row[4] = Ar; ++out.nrows; continue; }   // diagnostic build: no second clip
#endif
    RegionSink s2 = {S.r2x, S.r2y, S.r2off, S.cap_v, S.cap_r, 0, 0};
    SZC_PHASE(6);
    st = clip.second(c1, shifted, fdx, fdy, c2, op, S, &s2);
    SZC_PHASE(7);
    if (st != szv::ST_OK) return st;
    for (int ii = 0; ii < s2.nr; ++ii) {                                               // :156-165
      const i64 *TX = S.r2x + S.r2off[ii], *TY = S.r2y + S.r2off[ii];
      const int nt = S.r2off[ii + 1] - S.r2off[ii];
      // The third clip only decides whether the direction is reversed, and only when the shifted overlap is
      // the larger one (:160-163): otherwise its outcome cannot matter and it is not run.
      const double Anew = region_polyarea(TX, TY, nt);
      if (!(Anew / Ar - 1 > 0)) continue;
      // A point with a 1e-6 m disc inside both regions certifies that the third clip is not empty (a disc of
      // 4295 fixed-point units cannot vanish in a sweep that rounds by one unit).  Try the centroid of the
      // original overlap (the larger, shifted one normally contains it) and the centroid of the shifted
      // overlap; only if neither certifies run the clip.
      bool overlaps = clearly_inside(cx, cy, RX, RY, nr, 1e-6) && clearly_inside(cx, cy, TX, TY, nt, 1e-6);
      if (!overlaps && nt >= 3) {
        double tx, ty;
        region_centroid(TX, TY, nt, tx, ty);
        overlaps = clearly_inside(tx, ty, RX, RY, nr, 1e-6) && clearly_inside(tx, ty, TX, TY, nt, 1e-6);
      }
      if (!overlaps) {                                                                 // :158-159
        RegionSink s3 = {nullptr, nullptr, nullptr, 0, 0, 0, 0};
        st = clip.third(TX, TY, nt, RX, RY, nr, &s3);
        if (st != szv::ST_OK) return st;
        overlaps = s3.nr > 0;
      }
      if (overlaps) { fdx = -fdx; fdy = -fdy; }
    }
    double Fx = fdx * Ar * Force_factor, Fy = fdy * Ar * Force_factor;                 // :167
    // :170-183
    double v1x = f1.Ui + f1.ksi_ice * (pcx - f1.Xi), v1y = f1.Vi + f1.ksi_ice * (pcy - f1.Yi);
    double v2x = f2.Ui + f2.ksi_ice * (pcx - f2.Xi), v2y = f2.Vi + f2.ksi_ice * (pcy - f2.Yi);
    double vtx = v1x - v2x, vty = v1y - v2y;
    double dtx, dty;
    if (maxd(absd(vtx), absd(vty)) == 0) { dtx = 0; dty = 0; }
    else { double nv = ::sqrt(vtx * vtx + vty * vty); dtx = vtx / nv; dty = vty / nv; }
    double vn = ::sqrt(vtx * vtx + vty * vty);
    double coef = -(dtx * vtx + dty * vty) * dl * G * vn;
    double ftx = coef * dtx * dt, fty = coef * dty * dt;
    double nF = ::sqrt(Fx * Fx + Fy * Fy);
    if (::sqrt(ftx * ftx + fty * fty) > mu * nF) {
      ftx = -mu * nF * dtx; fty = -mu * nF * dty;
    }
    if (out.nrows >= S.cap_rows) return szv::ST_OUT_CAPACITY;
    double *row = S.rows + 5 * out.nrows;
    row[0] = Fx + ftx; row[1] = Fy + fty; row[2] = pcx; row[3] = pcy; row[4] = Ar;
    ++out.nrows;
  }
  return szv::ST_OK;
}

// Axis-aligned boxes of two rings further apart than 1e-6 m: no edge of one can meet the other and neither lies
// inside the other, so the clip is empty and floe_interactions.m returns its no-contact form ([0 0], [0 0], 0;
// :43-74 with no regions and no crossings) -- exactly what the full path would produce, without running it.
// (1e-6 m is 4295 units of the fixed-point grid: the boxes stay apart after the conversion.  NaN compares false.)
struct Box { double x0, x1, y0, y1; };
SZ_HD void box_start(Box &b) { b.x0 = SZC_INF; b.x1 = -SZC_INF; b.y0 = SZC_INF; b.y1 = -SZC_INF; }
SZ_HD void box_add(Box &b, double x, double y) {
  b.x0 = mind(b.x0, x); b.x1 = maxd(b.x1, x); b.y0 = mind(b.y0, y); b.y1 = maxd(b.y1, y);
}
template <class Ring>
SZ_HD void ring_box(const Ring &r, Box &b) {
  box_start(b);
#pragma unroll 4
  for (int k = 0; k < r.n; ++k) box_add(b, r.X(k), r.Y(k));
}
SZ_HD bool boxes_apart(const Box &a, const Box &b) {
  const double gap = 1e-6;
  return a.x1 + gap < b.x0 || b.x1 + gap < a.x0 || a.y1 + gap < b.y0 || b.y1 + gap < a.y0;
}
SZ_HD void no_contact(PairOut &out) { out.nrows = 0; out.sentinel = 0; out.has_sentinel = 1; }
// Floe against the domain box (floe_interactions.m:31-40: the part of the floe outside the box): a floe whose
// bounding box lies more than 1e-6 m inside an axis-aligned rectangular box has nothing outside and crosses no
// wall, so the result is the no-contact form as well.  c2b = the 5 closed corners (x, y interleaved).
SZ_HD bool well_inside_box(const Box &b, const double *c2b) {
  double x0 = c2b[0], x1 = c2b[0], y0 = c2b[1], y1 = c2b[1];
  for (int k = 1; k < 4; ++k) { x0 = mind(x0, c2b[2 * k]); x1 = maxd(x1, c2b[2 * k]); y0 = mind(y0, c2b[2 * k + 1]); y1 = maxd(y1, c2b[2 * k + 1]); }
  for (int k = 0; k < 4; ++k) {   // every corner must be a corner of that rectangle, each one used
    const double x = c2b[2 * k], y = c2b[2 * k + 1];
    if (!((x == x0 || x == x1) && (y == y0 || y == y1))) return false;
    const int k1 = (k + 1) & 3;
    if (!((x == c2b[2 * k1]) != (y == c2b[2 * k1 + 1]))) return false;   // consecutive corners share exactly one coordinate
  }
  const double gap = 1e-6;
  return b.x0 > x0 + gap && b.x1 < x1 - gap && b.y0 > y0 + gap && b.y1 < y1 - gap;
}

// Both parts with Wide records in the pair's HBM slice (the general path).
template <class LdsClip>
SZ_HD int pair_contact_t(const RingD &ring1, const Body &f1, const RingD &ring2, const Body &f2, bool boundary,
                         const double *c2b, bool periodic, double Modulus, double dt, double mu, double nu,
                         PairScratch &S, PairOut &out, void *lds_ws = nullptr, int lds_ws_bytes = 0) {
  out.nrows = 0; out.sentinel = 0; out.has_sentinel = 0;
  {
    Box b1, b2;
    ring_box(ring1, b1);
    if (boundary) {
      if (ring1.n > 0 && well_inside_box(b1, c2b)) { no_contact(out); return szv::ST_OK; }
    } else {
      ring_box(ring2, b2);
      if (ring1.n > 0 && ring2.n > 0 && boxes_apart(b1, b2)) { no_contact(out); return szv::ST_OK; }
    }
  }
  int nreg = 0;
  if (lds_ws != nullptr) {   // the sweep workspace of this lane is its slice of LDS (k_narrow_lds)
#if defined(__HIP_DEVICE_COMPILE__)
    ViaArrays<LdsClip> clip = {{S.ws, lds_ws, lds_ws_bytes, S.cap_nv}};
#else
    ViaArrays<LdsClip> clip = {{S.ws, S.ws, lds_ws_bytes, S.cap_nv}};   // (no LDS on the host: the pair's own workspace stands in)
#endif
    int st = pair_clip1(ring1, ring2, boundary, S, clip, nreg);
    if (st != szv::ST_OK) return st;
    return pair_finish(ring1, f1, ring2, f2, boundary, c2b, periodic, Modulus, dt, mu, nu, S, nreg, clip, out);
  }
  ViaArrays<ClipWide> clip = {{S.ws, S.cap_nv}};
  int st = pair_clip1(ring1, ring2, boundary, S, clip, nreg);
  if (st != szv::ST_OK) return st;
  return pair_finish(ring1, f1, ring2, f2, boundary, c2b, periodic, Modulus, dt, mu, nu, S, nreg, clip, out);
}
SZ_HD int pair_contact(const RingD &ring1, const Body &f1, const RingD &ring2, const Body &f2, bool boundary,
                       const double *c2b, bool periodic, double Modulus, double dt, double mu, double nu,
                       PairScratch &S, PairOut &out, void *lds_ws = nullptr, int lds_ws_bytes = 0) {
  return pair_contact_t<ClipWideLds>(ring1, f1, ring2, f2, boundary, c2b, periodic, Modulus, dt, mu, nu, S, out, lds_ws, lds_ws_bytes);
}
// Both parts through the single-minimum fast path only, with both rings staged in the lane's block (so the
// contact model reads them from LDS as well); ST_DEFER = redo this pair with pair_contact.
template <class Blk>
SZ_HD int pair_contact_fast(const RingD &ring1, const Body &f1, const RingD &ring2, const Body &f2, bool boundary,
                            const double *c2b, bool periodic, double Modulus, double dt, double mu, double nu,
                            PairScratch &S, PairOut &out, Blk *blk) {
  out.nrows = 0; out.sentinel = 0; out.has_sentinel = 0;
  if (boundary) {   // only floes at the walls need the clip against the box (general sweep: the box has horizontal edges)
    Box b1;
    box_start(b1);
    for (int k = 0; k < ring1.n; ++k) box_add(b1, ring1.X(k), ring1.Y(k));
    if (ring1.n > 0 && well_inside_box(b1, c2b)) { no_contact(out); return szv::ST_OK; }
    return ST_DEFER;
  }
  // A ring stored closed (last vertex an exact copy of the first) that is one vertex too long for the block is
  // staged without the copy when the closure rule of the contact model (:62-67, more than 1 m between the ends)
  // puts it back: every function below then sees the same closed ring, and the sweep strips the copy anyway.
  int n1 = ring1.n, n2 = ring2.n;
  if (n1 == (int)Blk::NV + 1 && ring1.X(n1 - 1) == ring1.X(0) && ring1.Y(n1 - 1) == ring1.Y(0) &&
      hyp(ring1.X(0) - ring1.X(n1 - 2), ring1.Y(0) - ring1.Y(n1 - 2)) > 1) --n1;
  if (n2 == (int)Blk::NV + 1 && ring2.X(n2 - 1) == ring2.X(0) && ring2.Y(n2 - 1) == ring2.Y(0) &&
      hyp(ring2.X(0) - ring2.X(n2 - 2), ring2.Y(0) - ring2.Y(n2 - 2)) > 1) --n2;
  if (n1 > (int)Blk::NV || n2 > (int)Blk::NV || n1 < 1 || n2 < 1) {
    // too long for the block: the general kernel's business -- unless the boxes are apart, which is its first
    // question too (pair_contact) and costs one pass over the rings here instead of a place in its work list
    if (ring1.n > 0 && ring2.n > 0) {
      Box b1, b2;
      ring_box(ring1, b1); ring_box(ring2, b2);
      if (boxes_apart(b1, b2)) { no_contact(out); return szv::ST_OK; }
    }
    return ((int)Blk::NV < (int)szx::FastBlockBig::NV && n1 <= (int)szx::FastBlockBig::NV && n2 <= (int)szx::FastBlockBig::NV && n1 >= 1 && n2 >= 1)
               ? (int)ST_DEFER_BIG : (int)ST_DEFER;
  }
  // (the sweep converts with to_fixed_in_range: coordinates must be finite and, with the <= 1 m shift of the
  // second clip, within its range -- anything else, e.g. a NaN floe, goes to the general sweep)
  const double safe = SZX_FIXED_SAFE_METRES - 2;
  bool in_range = true;
  Box b1, b2;
  box_start(b1); box_start(b2);
  for (int k = 0; k < n1; ++k) {
    const double X = ring1.X(k), Y = ring1.Y(k);
    in_range = in_range && absd(X) < safe && absd(Y) < safe;
    box_add(b1, X, Y);
    blk->v.m.x[0][k] = X; blk->v.m.y[0][k] = Y;
  }
  for (int k = 0; k < n2; ++k) {
    const double X = ring2.X(k), Y = ring2.Y(k);
    in_range = in_range && absd(X) < safe && absd(Y) < safe;
    box_add(b2, X, Y);
    blk->v.m.x[1][k] = X; blk->v.m.y[1][k] = Y;
  }
  if (!in_range) return ST_DEFER;
  if (boxes_apart(b1, b2)) { no_contact(out); return szv::ST_OK; }
#if defined(SZ_ABLATE) && SZ_ABLATE == 1
  no_contact(out); return szv::ST_OK;      // diagnostic build: stop after staging
#endif
  RingL l1 = {blk->v.m.x[0], blk->v.m.y[0], n1, 0}, l2 = {blk->v.m.x[1], blk->v.m.y[1], n2, 0};
  InBlock<Blk> clip = {blk, S.ws};
  int nreg = 0;
  int st = pair_clip1(l1, l2, false, S, clip, nreg);
  if (st != szv::ST_OK) return st;
#if defined(SZ_ABLATE) && SZ_ABLATE == 2
  no_contact(out); return szv::ST_OK;      // diagnostic build: stop after the first clip
#endif
  return pair_finish(l1, f1, l2, f2, false, c2b, periodic, Modulus, dt, mu, nu, S, nreg, clip, out);
}

} // namespace szc
