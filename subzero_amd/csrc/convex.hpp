// convex.hpp -- the sweep of vatti.hpp specialised for the case that dominates a sea-ice step: the INTERSECTION
// (even-odd) of two rings that each have ONE local minimum and ONE local maximum in y (every convex ring,
// e.g. a Voronoi floe or an overlap region, is such a ring) and that are in general position.
//
// For such input the general sweep (clipper.cpp:1560-3113) never holds more than four active edges -- the left
// and the right bound of each ring -- and produces at most one output ring.  This file walks exactly the same
// events in the same order and emits exactly the same points (same cross_point arithmetic, same front/back
// insertion, same orientation fix-up, same first vertex), but its whole state is four edge records, the <= 12
// vertices of each ring and six crossing slots (824 B, LDS-sized), and the output ring is a write-only deque.
// There are no linked lists to chase.
//
// It NEVER guesses.  Anything outside the plain case makes it return ST_BAIL before anything is written to the
// sink, and the caller runs the general sweep instead:
//   more vertices than the block holds (12 in the first pass, 32 in the second), duplicate or collinear input vertices (the general code would drop them),
//   more than one local minimum, a horizontal edge, two vertices on one scanline, parallel crossing edges,
//   any tie in x between edges of the two rings (a vertex on an edge, coincident vertices, shared edges),
//   a second output ring, an edge between the two top edges of a ring, collinear or duplicate output points,
//   a crossing order the general code would have to repair.
// tests/ check it bit for bit against the reference's Clipper (oracle/_ref) on millions of pairs.
#pragma once
#include "vatti.hpp"

namespace szx {

using szv::i64;
using szv::Pt;
using szv::same;

enum { ST_BAIL = 100, MAX_CROSS = 6 };
#ifndef SZX_BAIL_PROBE
#define SZX_BAIL_PROBE(site) ((void)(site))  // the host test build histograms the bail-out sites
#endif
#define SZX_BAIL(site) do { SZX_BAIL_PROBE(site); return ST_BAIL; } while (0)

struct CEdge {      // one active edge = the current edge of one bound of one ring
  Pt bot, top;
  i64 curx;
  double slope;
  short vtop;       // ring index of the top vertex
  signed char dir;  // ring index step towards the ring's top vertex along this bound
  signed char ring; // 0 = carries the output ring, -1 = not contributing
  signed char side; // szv::SIDE_LEFT / SIDE_RIGHT of the output ring
  signed char wind2;// inside (1) / outside (0) the other ring
  signed char poly; // 0 subject, 1 clip
  signed char last; // 1 when top is the ring's top vertex (no further edge in this bound)
};
struct CCross {
  Pt pt;
  signed char e1, e2; // edge ids (poly * 2 + bound)
  short pad0; int pad1;
};
struct Mono {         // one cleaned ring: vertex count, bottom / top vertex, direction of the left bound, bottom y
  short n, bot, top, dir_left;
  i64 bot_y;
};

// per-lane state block: 440 + 32 NV bytes (824 for the 12-vertex block of the first pass)
template <int NV_>
struct FastBlockT {
  enum { NV = NV_ };   // most vertices a ring may have
  // The vertices of the two rings, either as doubles in metres (VertsMetres: the step's fast path keeps the
  // rings of the pair here for the whole contact model, and the sweep converts on access) or as fixed-point
  // coordinates (VertsFixed: callers that start from int64 rings).
  union {
    struct { double x[2][NV_], y[2][NV_]; } m;
    Pt f[2][NV_];
  } v;
  double shift_x, shift_y;  // VertsMetres: added to ring 0 before the conversion when `shifted`
  int shifted, pad;
  CEdge ed[4];
  CCross X[MAX_CROSS];
  Mono mono[2];
  signed char ord[4];  // edge ids in active-edge-list order
  signed char sel[4];
  char tail[8];        // makes the size 8 x odd bytes: the same field of neighbouring lanes falls into different LDS banks
};
typedef FastBlockT<12> FastBlock;      // first pass over all pairs: 64 lanes x 824 B of LDS per wave
typedef FastBlockT<32> FastBlockBig;   // second pass over the few pairs with longer rings: 1464 B per lane
static_assert(sizeof(FastBlock) == 824 && sizeof(FastBlockBig) == 1464, "FastBlock layout changed: keep the size 8 x odd");

// MATLAB int64(v * 2^32): round half away from zero, saturating, NaN -> 0 (polyclip.m:63-66)
SZ_HD i64 to_fixed(double v) {
  double s = v * 4294967296.0;
  if (s != s) return 0;
  if (s >= 9223372036854775807.0) return 0x7FFFFFFFFFFFFFFFLL;
  if (s <= -9223372036854775808.0) return (i64)0x8000000000000000ULL;
  double t = (double)(i64)s; // trunc
  double f = s - t;
  if (f >= 0.5) t += 1.0;
  else if (f <= -0.5) t -= 1.0;
  return (i64)t;
}
// how the sweep reads vertex i of ring `poly` in fixed-point coordinates
struct VertsMetres {
  template <class Blk> SZ_HD Pt operator()(const Blk &K, int poly, int i) const {
    double X = K.v.m.x[poly][i], Y = K.v.m.y[poly][i];
    if (poly == 0 && K.shifted) { X = X + K.shift_x; Y = Y + K.shift_y; }   // floe_interactions.m:148-149
    Pt p; p.x = to_fixed(X); p.y = to_fixed(Y);
    return p;
  }
};
struct VertsFixed {
  template <class Blk> SZ_HD Pt operator()(const Blk &K, int poly, int i) const { return K.v.f[poly][i]; }
};

template <class Blk>
struct FastWsT {
  Blk *blk;       // LDS on the GPU
  Pt *deq;        // output deque, capacity 2 * cap_half, middle at cap_half (HBM: written, read once at the end)
  int cap_half;
};

SZ_HD i64 cx_at(const CEdge &g, i64 y) {
  return (y == g.top.y) ? g.top.x : g.bot.x + szv::rnd(g.slope * (double)(y - g.bot.y));
}
SZ_HD void cset_slope(CEdge &e) {
  i64 dy = e.top.y - e.bot.y;
  e.slope = (dy == 0) ? SZV_HORIZONTAL : (double)(e.top.x - e.bot.x) / (double)dy;
}
// clipper.cpp:622-689 with Curr.Y = cur_y for both edges (every active edge sits on the current scanline)
SZ_HD Pt ccross_point(const CEdge &a, const CEdge &b, i64 cur_y) {
  Pt ip;
  const double fa = szv::dabs(a.slope), fb = szv::dabs(b.slope);
  if (a.slope == 0 || b.slope == 0) {
    const CEdge &vert = (a.slope == 0) ? a : b, &other = (a.slope == 0) ? b : a;
    ip.x = vert.bot.x;
    double k = (double)other.bot.y - ((double)other.bot.x / other.slope);
    ip.y = szv::rnd((double)ip.x / other.slope + k);
  } else {
    double ka = (double)a.bot.x - (double)a.bot.y * a.slope;
    double kb = (double)b.bot.x - (double)b.bot.y * b.slope;
    double q = (kb - ka) / (a.slope - b.slope);
    ip.y = szv::rnd(q);
    ip.x = (fa < fb) ? szv::rnd(a.slope * q + ka) : szv::rnd(b.slope * q + kb);
  }
  if (ip.y < a.top.y || ip.y < b.top.y) {
    ip.y = (a.top.y > b.top.y) ? a.top.y : b.top.y;
    ip.x = cx_at((fa < fb) ? a : b, ip.y);
  }
  if (ip.y > cur_y) {
    ip.y = cur_y;
    ip.x = cx_at((fa > fb) ? b : a, ip.y);
  }
  return ip;
}

// Checks that the general code would keep every vertex of both rings (closing duplicates stripped as
// clipper.cpp:1055-1058); finds each ring's single bottom / top vertex and which way its left bound runs.
// 0, or the reason for the bail-out.
template <class Blk, class Verts>
SZ_HD int scan_rings(Blk &K, const Verts &vert, int na, int nb) {
  for (int poly = 0; poly < 2; ++poly) {
    int hi = (poly ? nb : na) - 1;
    if (hi + 1 > (int)Blk::NV) return 2;
    if (hi < 2) return 1;
    const Pt first = vert(K, poly, 0);
    Pt last = vert(K, poly, hi);
    while (hi > 0 && same(last, first)) { --hi; last = vert(K, poly, hi); }
    while (hi > 0) { const Pt before = vert(K, poly, hi - 1); if (!same(last, before)) break; --hi; last = before; }
    if (hi < 2) return 1;
    const int n = hi + 1;
    int nmin = 0, nmax = 0, bot = 0, top = 0;
    Pt p = last, c = first, at_bot = first, bot_next = first, bot_prev = first;
    for (int i = 0; i < n; ++i) {
      const Pt q = (i + 1 == n) ? first : vert(K, poly, i + 1);
      const i64 lim = 0x3FFFFFFFFFFFFFFFLL;
      if (c.x > lim || c.y > lim || -c.x > lim || -c.y > lim) return 3;
      if (same(c, q) || szv::collinear3(p, c, q)) return 4;   // the general code would drop this vertex
      if (c.y == q.y) return 5;                               // horizontal edge
      if (c.y > p.y && c.y > q.y) { ++nmin; bot = i; at_bot = c; bot_next = q; bot_prev = p; }
      if (c.y < p.y && c.y < q.y) { ++nmax; top = i; }
      p = c; c = q;
    }
    if (nmin != 1 || nmax != 1) return 6;
    // the left bound starts with the edge of larger slope (clipper.cpp:1194-1204)
    double s_next = (double)(bot_next.x - at_bot.x) / (double)(bot_next.y - at_bot.y);
    double s_prev = (double)(bot_prev.x - at_bot.x) / (double)(bot_prev.y - at_bot.y);
    if (s_next == s_prev) return 7;
    Mono m; m.n = (short)n; m.bot = (short)bot; m.top = (short)top; m.dir_left = (short)((s_next < s_prev) ? -1 : +1);
    m.bot_y = at_bot.y;
    K.mono[poly] = m;
  }
  return 0;
}

struct Deque {
  Pt *d; int cap_half, lo, hi; // elements d[lo .. hi)
  Pt front, back;              // copies of d[lo], d[hi - 1]
  bool open, used, overflow;
};

SZ_HD void dq_start(Deque &q, const Pt &pt) {
  q.used = true; q.open = true;
  q.lo = q.cap_half; q.hi = q.cap_half + 1;
  q.d[q.cap_half] = pt; q.front = q.back = pt;
}
SZ_HD void dq_emit(Deque &q, int side, const Pt &pt) {
  // Clipper::AddOutPt for the single output ring: front insertion for the left side, back for the right
  if (!q.open) return;
  const bool to_front = (side == szv::SIDE_LEFT);
  if (same(pt, to_front ? q.front : q.back)) return;
  if (to_front ? (q.lo == 0) : (q.hi == 2 * q.cap_half)) { q.overflow = true; return; }
  const int at = to_front ? --q.lo : q.hi++;
  q.d[at] = pt;
  if (to_front) q.front = pt; else q.back = pt;
}

// The rings are in w.blk->v (see FastBlockT); na, nb = vertex counts as stored (a closing duplicate may follow).
template <class Blk, class Verts, class Sink>
SZ_HD int clip_monotone(const FastWsT<Blk> &w, const Verts &vert, int na, int nb, Sink &sink) {
  Blk &K = *w.blk;
  if (int why = scan_rings(K, vert, na, nb)) SZX_BAIL(30 + why);
  CEdge *ed = K.ed;
  signed char *ord = K.ord;
  int k = 0;                   // active edges
  int pending = 3;             // bit p: ring p not yet inserted
  Deque q;
  q.d = w.deq; q.cap_half = w.cap_half; q.lo = q.hi = w.cap_half; q.open = false; q.used = false; q.overflow = false;
  q.front.x = q.front.y = q.back.x = q.back.y = 0;
  i64 y = 0;                   // current scanline (unused until the first ring is inserted)

  bool done_early = false;
  for (int guard = 0; guard < 64; ++guard) {
    // Scanbeams with the edges of only one ring cannot cross or emit anything (its edges are outside the
    // other ring, winding 0), in the general sweep as here: skip them.  (1) Only the lower ring is in and the
    // upper one is still pending: walk the two bounds up to the scanline of the pending ring's bottom; if the
    // ring ends first the y-ranges are disjoint and the result is empty.  (2) One ring has ended and no output
    // ring is open: nothing more can happen.
    if (k == 2 && pending != 0) {
      const int pp = (pending & 1) ? 0 : 1;
      const i64 by = K.mono[pp].bot_y;
      bool ended = false;
      for (int i = 0; i < 2; ++i) {
        CEdge &e = ed[ord[i]];
        const Mono m = K.mono[e.poly];
        while (e.top.y > by) {
          if (e.last) { ended = true; break; }
          int vt = e.vtop + e.dir;
          if (vt < 0) vt += m.n; else if (vt >= m.n) vt -= m.n;
          e.bot = e.top; e.top = vert(K, e.poly, vt); e.vtop = (short)vt;
          e.last = (vt == m.top);
          cset_slope(e);
        }
        if (ended) break;
      }
      if (ended) return szv::ST_OK;
      y = by + 1;                 // any scanline below by: cross points are clamped to it only from above
    }
    if (k == 2 && pending == 0 && !q.open) { done_early = true; break; }
    // next scanline: the highest y still below the current one
    bool have = false;
    i64 top_y = 0;
    int sources = 0, src_poly = -1;
    for (int i = 0; i < k; ++i) {
      const CEdge &e = ed[ord[i]];
      if (!have || e.top.y > top_y) { have = true; top_y = e.top.y; sources = 1; src_poly = e.poly; }
      else if (e.top.y == top_y) { if (e.poly != src_poly) SZX_BAIL(4); ++sources; }
    }
    int insert = -1;            // ring whose local minimum lies on the next scanline
    for (int p = 0; p < 2; ++p) {
      if (!(pending >> p & 1)) continue;
      i64 by = K.mono[p].bot_y;
      if (!have || by > top_y) { have = true; top_y = by; sources = 1; insert = p; }
      else if (by == top_y) SZX_BAIL(5);
    }
    if (!have) break;
    // two edges of one ring may only end together at that ring's top vertex
    if (sources > 1) {
      int ends = 0, lasts = 0;
      for (int i = 0; i < k; ++i) { const CEdge &e = ed[ord[i]]; if (e.top.y == top_y) { ++ends; lasts += e.last; } }
      if (ends != 2 || lasts != 2) SZX_BAIL(6);
    }

    // ---- crossings inside the scanbeam (clipper.cpp:2856-2955) ----
    for (int i = 0; i < k; ++i) { CEdge &e = ed[ord[i]]; e.curx = cx_at(e, top_y); }
    int nX = 0;
    {
      signed char *sl = K.sel;
      int ns = k;
      for (int i = 0; i < k; ++i) sl[i] = ord[i];
      bool modified;
      do {
        modified = false;
        int i = 0;
        while (i + 1 < ns) {
          CEdge &e = ed[sl[i]], &en = ed[sl[i + 1]];
          if (e.curx > en.curx) {
            if (e.poly == en.poly) SZX_BAIL(7);             // bounds of one ring do not cross
            if (e.slope == en.slope) SZX_BAIL(8);
            if (nX >= MAX_CROSS) SZX_BAIL(9);
            Pt pt = ccross_point(e, en, y);
            if (pt.y < top_y) { pt.x = e.curx; pt.y = top_y; }
            K.X[nX].e1 = sl[i]; K.X[nX].e2 = sl[i + 1]; K.X[nX].pt = pt; ++nX;
            signed char t = sl[i]; sl[i] = sl[i + 1]; sl[i + 1] = t;
            modified = true;
          } else if (e.curx == en.curx && e.poly != en.poly) SZX_BAIL(10); // tie between the two rings
          ++i;                                               // (after a swap the same edge keeps moving right)
        }
        if (ns > 1) --ns; else break;
      } while (modified);
    }
    if (nX > 1) {
      // stable sort by y, bottom-most first, then every crossing must be between neighbours (clipper.cpp:2934)
      for (int a = 1; a < nX; ++a) {
        CCross v = K.X[a];
        int b = a;
        while (b > 0 && v.pt.y > K.X[b - 1].pt.y) { K.X[b] = K.X[b - 1]; --b; }
        K.X[b] = v;
      }
      signed char *sl = K.sel;
      for (int i = 0; i < k; ++i) sl[i] = ord[i];
      for (int a = 0; a < nX; ++a) {
        int p1 = -1, p2 = -1;
        for (int i = 0; i < k; ++i) { if (sl[i] == K.X[a].e1) p1 = i; if (sl[i] == K.X[a].e2) p2 = i; }
        if (p1 - p2 != 1 && p2 - p1 != 1) SZX_BAIL(11);      // the general code would reorder: leave it to it
        signed char t = sl[p1]; sl[p1] = sl[p2]; sl[p2] = t;
      }
    }
    for (int a = 0; a < nX; ++a) {
      // Clipper::IntersectEdges for two closed edges of different rings, even-odd, intersection
      CEdge &e1 = ed[K.X[a].e1], &e2 = ed[K.X[a].e2];
      const Pt pt = K.X[a].pt;
      e1.wind2 = (signed char)(e1.wind2 == 0 ? 1 : 0);
      e2.wind2 = (signed char)(e2.wind2 == 0 ? 1 : 0);
      const bool c1 = e1.ring >= 0, c2 = e2.ring >= 0;
      if (c1 || c2) {
        // both: AddLocalMaxPoly, the ring closes here; one: that edge emits and hands ring and side over
        dq_emit(q, c1 ? e1.side : e2.side, pt);
        if (c1 && c2) { e1.ring = e2.ring = -1; q.open = false; }
        else {
          signed char s = e1.side; e1.side = e2.side; e2.side = s;
          signed char r = e1.ring; e1.ring = e2.ring; e2.ring = r;
        }
      } else {                              // AddLocalMinPoly(e1, e2): a ring starts at a crossing
        if (q.used) SZX_BAIL(12);
        if (!(e1.slope < e2.slope)) SZX_BAIL(13);   // e1 crosses from left to right: its slope is the smaller
        dq_start(q, pt);
        e1.ring = e2.ring = 0;
        e1.side = szv::SIDE_RIGHT; e2.side = szv::SIDE_LEFT;
      }
      int p1 = -1, p2 = -1;
      for (int i = 0; i < k; ++i) { if (ord[i] == K.X[a].e1) p1 = i; if (ord[i] == K.X[a].e2) p2 = i; }
      signed char t = ord[p1]; ord[p1] = ord[p2]; ord[p2] = t;
    }

    // ---- top of the scanbeam (clipper.cpp:3009-3113); curx of every edge is already x(top_y) ----
    for (int i = 0; i < k; ++i) {
      CEdge &e = ed[ord[i]];
      if (e.top.y != top_y) continue;
      if (e.last) {
        // DoMaxima: the other top edge of the ring must be the next active edge
        if (i + 1 >= k) SZX_BAIL(14);
        CEdge &mp = ed[ord[i + 1]];
        if (mp.poly != e.poly || !(mp.top.y == top_y && mp.last)) SZX_BAIL(15);
        if (e.ring >= 0 && mp.ring >= 0) { dq_emit(q, e.side, e.top); e.ring = mp.ring = -1; q.open = false; }
        else if (e.ring >= 0 || mp.ring >= 0) SZX_BAIL(16);   // the general code reports an error here
        for (int j = i + 2; j < k; ++j) ord[j - 2] = ord[j];
        k -= 2;
        --i;                                 // the edge that slid into slot i is looked at next
        continue;
      }
      // an intermediate vertex of a bound: emit it and step to the next edge of the bound.  (The general
      // code does this in a second pass over the list; with one vertex per scanline the order is the same.)
      if (e.ring >= 0) dq_emit(q, e.side, e.top);
      const Mono m = K.mono[e.poly];
      int vt = e.vtop + e.dir;
      if (vt < 0) vt += m.n; else if (vt >= m.n) vt -= m.n;
      e.bot = e.top; e.top = vert(K, e.poly, vt); e.vtop = (short)vt;
      e.last = (vt == m.top);
      cset_slope(e);
      e.curx = e.bot.x;
      // a neighbour of the other ring passing through this vertex is the join case of the general code
      if (i > 0 && ed[ord[i - 1]].curx == e.curx) SZX_BAIL(17);
      if (i + 1 < k && ed[ord[i + 1]].curx == e.curx) SZX_BAIL(18);
    }
    y = top_y;

    // ---- insertion of a ring's local minimum on this scanline (clipper.cpp:1978-2077) ----
    if (insert >= 0) {
      const Mono m = K.mono[insert];
      const Pt bot = vert(K, insert, m.bot);
      int pos = 0;
      while (pos < k) {
        const i64 ox = ed[ord[pos]].curx;
        if (ox == bot.x) SZX_BAIL(19);        // a tie: vertex on an edge or coincident vertices
        if (bot.x < ox) break;
        ++pos;
      }
      for (int i = k - 1; i >= pos; --i) ord[i + 2] = ord[i];
      // even-odd winding against the other ring = parity of its edges to the left (clipper.cpp:1624-1729);
      // inside the other ring -> contributing -> AddLocalMinPoly (clipper.cpp:1841-1882)
      const int w2 = pos & 1;
      if (w2) {
        if (q.used) SZX_BAIL(3);              // a second output ring: not the plain case
        dq_start(q, bot);
      }
      for (int bound = 0; bound < 2; ++bound) {
        const int id = insert * 2 + bound;
        ord[pos + bound] = (signed char)id;
        CEdge e;
        e.poly = (signed char)insert;
        e.dir = (signed char)(bound == 0 ? m.dir_left : -m.dir_left);
        int vt = m.bot + e.dir;
        if (vt < 0) vt += m.n; else if (vt >= m.n) vt -= m.n;
        e.bot = bot; e.top = vert(K, insert, vt); e.vtop = (short)vt;
        e.last = (vt == m.top);
        cset_slope(e);
        e.curx = bot.x;
        e.wind2 = (signed char)w2;
        e.ring = (signed char)(w2 ? 0 : -1);
        e.side = (signed char)(w2 ? (bound == 0 ? szv::SIDE_LEFT : szv::SIDE_RIGHT) : szv::SIDE_NONE);
        ed[id] = e;
      }
      k += 2;
      pending &= ~(1 << insert);
    }
    if (q.overflow) SZX_BAIL(20);
  }
  if ((k != 0 && !done_early) || pending != 0 || q.open || q.overflow) SZX_BAIL(21);
  if (!q.used) return szv::ST_OK;            // no overlap: empty result
  const int cnt = q.hi - q.lo;
  if (cnt < 3) SZX_BAIL(22);               // the general code disposes of such rings in FixupOutPolygon
  const Pt *d = q.d + q.lo;
  // Area(outRec) from the head, following next (clipper.cpp:405-416); an outer ring must come out positive.
  // FixupOutPolygon would drop duplicate / collinear points and move the start: leave those to the general code.
  double a = 0;
  {
    Pt pv = d[cnt - 1], cu = d[0];
    for (int i = 0; i < cnt; ++i) {
      const Pt nx = d[i + 1 == cnt ? 0 : i + 1];
      a += (double)(pv.x + cu.x) * (double)(pv.y - cu.y);
      if (same(cu, nx) || szv::collinear3(pv, cu, nx)) SZX_BAIL(23);
      pv = cu; cu = nx;
    }
  }
  const bool reverse = !(a * 0.5 > 0);
  // BuildResult: start at pts.prev and walk backwards (clipper.cpp:3199-3217)
  if (!sink.ring_begin()) return szv::ST_OUT_CAPACITY;
  for (int i = 0; i < cnt; ++i) {
    const int at = reverse ? (i + 1 == cnt ? 0 : i + 1) : cnt - 1 - i;
    if (!sink.vertex(d[at])) return szv::ST_OUT_CAPACITY;
  }
  if (!sink.ring_end()) return szv::ST_OUT_CAPACITY;
  return szv::ST_OK;
}

} // namespace szx
