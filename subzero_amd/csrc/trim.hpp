// trim.hpp -- two rings cut down to what an INTERSECTION clip of them can see.
//
// Whatever two rings have in common lies in the box Q that their bounding boxes have in common, and every
// crossing of an edge of one with an edge of the other lies there too.  A run of consecutive vertices of a ring
// that all lie strictly beyond the SAME side of Q (all left of it, all right, all below or all above) stays in
// that open half-plane, and so does the chord from its start to its end: replacing the run's inner vertices by
// that chord changes the ring only beyond that side of Q (the closed curve run + chord has winding number 0 about
// every point of Q), never an edge that reaches Q, never the even-odd parity left of anything in Q.  The sweep of
// the reference (clipper.cpp:1560-1621) then meets the same contributing edges in the same order and emits the
// same rings, vertex for vertex -- with a fraction of the edges: evolved floe outlines (the reference's
// FloeShapes.mat, ~50-90 vertices) that touch along one side come down to a third or less.
//
// What keeps it exact, not just equal as a point set:
//   * two vertices are kept at each end of a run, so every vertex the ring keeps next to an edge that can reach Q
//     has its original neighbours on both sides: AddPath's collinear-vertex removal (clipper.cpp:1100-1114) sees
//     the triples it saw before, and the kept edges keep their end points and slopes (Dx, :591-596) to the bit;
//   * a run is left alone when the chord would be collinear with the kept vertices next to it (the removal would
//     then cascade into them);
//   * the reference orders local minima of equal Y, and crossings of equal Y, by an unstable std::sort
//     (clipper.cpp:1234, :2896): with fewer elements equal keys may land in another order, and with them rings
//     that share a vertex or an edge (floes at rest against each other, fracture pieces) come out split, joined
//     or rotated differently.  So both rings are handed over untrimmed whenever two kept vertices share a Y
//     (this covers horizontal edges and repeated vertices too).
// A crossing that the reference's double-precision IntersectPoint (:604-664) puts outside its scanbeam is
// clamped to the beam, and trimming beyond the left / right side of Q makes beams taller; this needs a crossing
// within rounding error (~1e-8 m for near-parallel floe edges) of the height of an unrelated, removed vertex
// and moves one output vertex by that error: not seen in 1.2e6 fuzzed pairs against the reference Clipper
// (tests/test_clip_cpu.py), far inside the 1e-10 relative bar on forces if it ever is.
#pragma once
#include "vatti.hpp"

namespace szt {

using szv::i64;
using szv::Pt;

struct Box { i64 xmin, xmax, ymin, ymax; };

SZ_HD void box_of(const i64 *x, const i64 *y, int n, Box &b) {
  b.xmin = b.xmax = x[0]; b.ymin = b.ymax = y[0];
#pragma unroll 4
  for (int k = 1; k < n; ++k) {
    i64 vx = x[k], vy = y[k];
    b.xmin = vx < b.xmin ? vx : b.xmin; b.xmax = vx > b.xmax ? vx : b.xmax;
    b.ymin = vy < b.ymin ? vy : b.ymin; b.ymax = vy > b.ymax ? vy : b.ymax;
  }
}
SZ_HD int beyond(i64 x, i64 y, const Box &q) {
  return (int)(x < q.xmin) | ((int)(x > q.xmax) << 1) | ((int)(y < q.ymin) << 2) | ((int)(y > q.ymax) << 3);
}

enum { MIN_RING = 7 };   // shorter rings have no run worth cutting (a run needs 5 vertices to lose one)

// Compacts the ring in place; returns the new vertex count (== n when nothing was removed; a closing duplicate,
// which AddPath strips anyway (clipper.cpp:1055), is dropped only together with a removal).
SZ_HD int trim_ring(i64 *x, i64 *y, int n, const Box &q) {
  int m = n;
  if (m > 1 && x[m - 1] == x[0] && y[m - 1] == y[0]) --m;
  if (m < MIN_RING) return n;
  int w = 0, i = 0;
  while (i < m) {
    const int s = i;
    int side = beyond(x[i], y[i], q);
    if (side)
      while (i + 1 < m) {
        int both = side & beyond(x[i + 1], y[i + 1], q);
        if (!both) break;
        side = both; ++i;
      }
    const int e = i;
    ++i;
    bool cut = side && e - s >= 4;
    if (cut) {
      Pt a, b, c, d;
      a.x = x[s]; a.y = y[s]; b.x = x[s + 1]; b.y = y[s + 1];
      c.x = x[e - 1]; c.y = y[e - 1]; d.x = x[e]; d.y = y[e];
      cut = !szv::collinear3(a, b, c) && !szv::collinear3(b, c, d);
    }
    for (int k = s; k <= e; ++k) {
      if (cut && k >= s + 2 && k <= e - 2) continue;
      x[w] = x[k]; y[w] = y[k]; ++w;
    }
  }
  return w == m ? n : w;
}

SZ_HD bool heights_distinct(const i64 *y1, int n1, const i64 *y2, int n2) {
  // (no early exit inside a row: the loads of a row are independent and overlap)
  for (int a = 0; a < n1 + n2; ++a) {
    const i64 va = a < n1 ? y1[a] : y2[a - n1];
    bool same = false;
#pragma unroll 4
    for (int b = a + 1; b < n1; ++b) same |= (va == y1[b]);
#pragma unroll 4
    for (int b = (a + 1 > n1 ? a + 1 - n1 : 0); b < n2; ++b) same |= (va == y2[b]);
    if (same) return false;
  }
  return true;
}

// Both rings of an INTERSECTION, in place.  Returns false when the rings must be handed over as they were
// (the caller restores them: the arrays may have been compacted); true with the new counts otherwise
// (which equal the old ones when there was nothing to cut).
SZ_HD void common_box(const Box &A, const Box &B, Box &q) {
  q.xmin = A.xmin > B.xmin ? A.xmin : B.xmin; q.xmax = A.xmax < B.xmax ? A.xmax : B.xmax;
  q.ymin = A.ymin > B.ymin ? A.ymin : B.ymin; q.ymax = A.ymax < B.ymax ? A.ymax : B.ymax;
}
SZ_HD bool box_empty(const Box &q) { return q.xmin > q.xmax || q.ymin > q.ymax; }
// Two rings whose boxes have nothing in common (strictly apart along x or along y) have an empty INTERSECTION under
// every fill rule: at no scanline does an edge of one have an odd number -- or any non-zero winding sum -- of the
// other's edges to its left (they are never active together, or all of them are on one side), so IsContributing
// (clipper.cpp:1740-1790) never holds and Execute returns no paths.  The caller still owes the reference's range
// error (clipper.cpp:883-896), which `in_range` of both boxes rules out, and its "Clipper Error." for a ring that is
// degenerate on its own (a lone lattice polygon with coincident vertices and edges can fail DoMaxima or the crossing
// order, clipper.cpp:2841,3005): clip.hip answers without a sweep only when no two vertices of the pair share a
// height, the condition under which this file trims at all -- and when at least one of the rings is a ring AddPath keeps
// (three consecutive vertices not on one line): with BOTH rings degenerate Execute finds no local minimum and fails.
SZ_HD bool box_in_range(const Box &b) {
  const i64 hi = 0x3FFFFFFFFFFFFFFFLL;
  return !(b.xmax > hi || b.ymax > hi || -b.xmin > hi || -b.ymin > hi);
}

SZ_HD bool trim_pair_in(i64 *ax, i64 *ay, int &na, i64 *bx, i64 *by, int &nb, const Box &q);
SZ_HD bool trim_pair(i64 *ax, i64 *ay, int &na, i64 *bx, i64 *by, int &nb) {
  if (na < 1 || nb < 1 || (na < MIN_RING && nb < MIN_RING)) return true;
  Box A, B, q;
  box_of(ax, ay, na, A); box_of(bx, by, nb, B);
  common_box(A, B, q);
  return trim_pair_in(ax, ay, na, bx, by, nb, q);
}
// (the same with the common box already known)
SZ_HD bool trim_pair_in(i64 *ax, i64 *ay, int &na, i64 *bx, i64 *by, int &nb, const Box &q) {
  if (na < 1 || nb < 1 || (na < MIN_RING && nb < MIN_RING)) return true;
  if (box_empty(q)) return true;                              // nothing in common: left to the sweep as it is
  const int ta = trim_ring(ax, ay, na, q), tb = trim_ring(bx, by, nb, q);
  if (ta == na && tb == nb) return true;
  // (an untouched ring may still carry its closing duplicate: not a tie)
  const int ea = (ta == na && na > 1 && ax[na - 1] == ax[0] && ay[na - 1] == ay[0]) ? na - 1 : ta;
  const int eb = (tb == nb && nb > 1 && bx[nb - 1] == bx[0] && by[nb - 1] == by[0]) ? nb - 1 : tb;
  if (!heights_distinct(ay, ea, by, eb)) return false;
  na = ta; nb = tb;
  return true;
}

#if defined(__HIPCC__)
// ---- the same by the 64 lanes of a wave that all hold the same pair (clip.hip k_clip_front, step_core.hpp narrow_t in
// its wave form): the reductions and the O(n^2) height check spread over the lanes, the serial scan of trim_ring run
// by every lane alike (same reads, same writes).  The arrays must be in LDS (or otherwise coherent inside a wave).
__device__ inline i64 wave_min(i64 v) {
  for (int d = 32; d >= 1; d >>= 1) { const i64 o = __shfl_xor(v, d); v = o < v ? o : v; }
  return v;
}
__device__ inline i64 wave_max(i64 v) {
  for (int d = 32; d >= 1; d >>= 1) { const i64 o = __shfl_xor(v, d); v = o > v ? o : v; }
  return v;
}
__device__ inline void wave_box(const i64 *x, const i64 *y, int n, Box &b) {
  const i64 big = 0x7FFFFFFFFFFFFFFFLL;
  i64 x0 = big, x1 = -big - 1, y0 = big, y1 = -big - 1;
  for (int k = (int)(threadIdx.x & 63); k < n; k += 64) {
    const i64 vx = x[k], vy = y[k];
    x0 = vx < x0 ? vx : x0; x1 = vx > x1 ? vx : x1; y0 = vy < y0 ? vy : y0; y1 = vy > y1 ? vy : y1;
  }
  b.xmin = wave_min(x0); b.xmax = wave_max(x1); b.ymin = wave_min(y0); b.ymax = wave_max(y1);
}
__device__ inline bool wave_heights_distinct(const i64 *y1, int n1, const i64 *y2, int n2) {
  const int total = n1 + n2;
  bool same = false;
  for (int r = (int)(threadIdx.x & 63); r < total; r += 64) {
    const i64 v = r < n1 ? y1[r] : y2[r - n1];
    for (int b = r + 1; b < n1; ++b) same |= (v == y1[b]);
    for (int b = (r + 1 > n1 ? r + 1 - n1 : 0); b < n2; ++b) same |= (v == y2[b]);
  }
  return __ballot(same) == 0ULL;
}
// trim_ring by a whole wave, for rings of up to 128 vertices (two per lane): the side codes by the lanes, four ballots
// per chunk; the serial scan of trim_ring then goes from RUN to run on those masks (a run from s ends at the last e
// for which the codes of s..e still have a side in common -- the greedy narrowing of `side` there stops at exactly
// that e -- so its length is the longest stretch of ones from bit s among the masks of the sides s lies beyond; a
// stretch of vertices inside the box is kept whole), and the kept vertices are moved by the lanes that hold them.
// Every lane returns the same; x / y must be in LDS (or otherwise coherent inside a wave).
__device__ inline int ones_from(unsigned long long lo, unsigned long long hi, int s) {   // ones in a row from bit s of hi:lo
  unsigned long long a, b;
  if (s == 0) { a = lo; b = hi; }
  else if (s < 64) { a = (lo >> s) | (hi << (64 - s)); b = hi >> s; }
  else { a = hi >> (s - 64); b = 0; }
  if (~a) return __builtin_ctzll(~a);
  return 64 + (~b ? __builtin_ctzll(~b) : 64);
}
__device__ inline int trim_ring_wave(i64 *x, i64 *y, int n, const Box &q) {
  int m = n;
  if (m > 1 && x[m - 1] == x[0] && y[m - 1] == y[0]) --m;
  if (m < MIN_RING) return n;
  if (m > 128) return trim_ring(x, y, n, q);
  const int lane = (int)(threadIdx.x & 63);
  i64 vx0 = 0, vy0 = 0, vx1 = 0, vy1 = 0;
  int c0 = 0, c1 = 0;
  if (lane < m) { vx0 = x[lane]; vy0 = y[lane]; c0 = beyond(vx0, vy0, q); }
  if (lane + 64 < m) { vx1 = x[lane + 64]; vy1 = y[lane + 64]; c1 = beyond(vx1, vy1, q); }
  unsigned long long lo[4], hi[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) { lo[b] = __ballot((c0 >> b) & 1); hi[b] = __ballot((c1 >> b) & 1); }
  const unsigned long long any_lo = lo[0] | lo[1] | lo[2] | lo[3], any_hi = hi[0] | hi[1] | hi[2] | hi[3];
  if ((any_lo | any_hi) == 0ULL) return n;                  // every vertex inside the box: nothing to cut
  unsigned long long keep_lo = ~0ULL, keep_hi = ~0ULL;      // (bits beyond m - 1 are never looked at)
  int s = 0;
  while (s < m) {
    // a stretch of vertices inside the box (code 0): kept, each a run of its own
    const int inside = ones_from(~any_lo, ~any_hi, s);
    s += inside;
    if (s >= m) break;
    int len = 1;
#pragma unroll
    for (int b = 0; b < 4; ++b) { const int l = ones_from(lo[b], hi[b], s); len = l > len ? l : len; }
    int e = s + len - 1;
    if (e > m - 1) e = m - 1;
    if (e - s >= 4) {
      Pt a, b, c, d;
      a.x = x[s]; a.y = y[s]; b.x = x[s + 1]; b.y = y[s + 1];
      c.x = x[e - 1]; c.y = y[e - 1]; d.x = x[e]; d.y = y[e];
      if (!szv::collinear3(a, b, c) && !szv::collinear3(b, c, d))
        for (int k = s + 2; k <= e - 2; ++k) { if (k < 64) keep_lo &= ~(1ULL << k); else keep_hi &= ~(1ULL << (k - 64)); }
    }
    s = e + 1;
  }
  const unsigned long long in_lo = m >= 64 ? ~0ULL : ((1ULL << m) - 1ULL);
  const unsigned long long in_hi = m >= 128 ? ~0ULL : (m > 64 ? ((1ULL << (m - 64)) - 1ULL) : 0ULL);
  keep_lo &= in_lo; keep_hi &= in_hi;
  const int w = __popcll(keep_lo) + __popcll(keep_hi);
  if (w == m) return n;
  __builtin_amdgcn_wave_barrier();                          // (every read of the scan above before the first write below)
  const unsigned long long below = (1ULL << lane) - 1ULL;
  if ((keep_lo >> lane) & 1ULL) { const int at = __popcll(keep_lo & below); x[at] = vx0; y[at] = vy0; }
  if ((keep_hi >> lane) & 1ULL) { const int at = __popcll(keep_lo) + __popcll(keep_hi & below); x[at] = vx1; y[at] = vy1; }
  __builtin_amdgcn_wave_barrier();
  return w;
}
// trim_pair by a whole wave (every lane returns the same)
__device__ inline bool trim_pair_wave(i64 *ax, i64 *ay, int &na, i64 *bx, i64 *by, int &nb) {
  if (na < 1 || nb < 1 || (na < MIN_RING && nb < MIN_RING)) return true;
  Box A, B, q;
  wave_box(ax, ay, na, A); wave_box(bx, by, nb, B);
  common_box(A, B, q);
  if (box_empty(q)) return true;
  const int ta = trim_ring_wave(ax, ay, na, q), tb = trim_ring_wave(bx, by, nb, q);
  if (ta == na && tb == nb) return true;
  __builtin_amdgcn_wave_barrier();
  const int ea = (ta == na && na > 1 && ax[na - 1] == ax[0] && ay[na - 1] == ay[0]) ? na - 1 : ta;
  const int eb = (tb == nb && nb > 1 && bx[nb - 1] == bx[0] && by[nb - 1] == by[0]) ? nb - 1 : tb;
  if (!wave_heights_distinct(ay, ea, by, eb)) return false;
  na = ta; nb = tb;
  return true;
}
#endif

} // namespace szt
