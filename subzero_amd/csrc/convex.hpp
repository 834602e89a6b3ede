// convex.hpp -- the sweep of vatti.hpp specialised for the case that dominates a sea-ice step: the INTERSECTION
// (even-odd) of two rings that each have ONE local minimum and ONE local maximum in y (every convex ring,
// e.g. a Voronoi floe or an overlap region, is such a ring) and that are in general position.
//
// For such input the general sweep (clipper.cpp:1560-3113) never holds more than four active edges -- the left
// and the right bound of each ring -- and produces at most one output ring.  This file walks exactly the same
// events in the same order and emits exactly the same points (same cross_point arithmetic, same front/back
// insertion, same orientation fix-up, same first vertex), but its whole state is four edge records (registers), the
// <= 16 vertices of each ring and four crossing slots (616 B of LDS), and the output ring is a write-only deque.
// There are no linked lists to chase.
//
// It NEVER guesses.  Anything outside the plain case makes it return ST_BAIL before anything is written to the
// sink, and the caller runs the general sweep instead:
//   more vertices than the block holds (16), duplicate or collinear input vertices (the general code would drop them),
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

enum { ST_BAIL = 100, MAX_CROSS = 4 };   // crossings in one scanbeam: each bound of one ring with each bound of the other
#ifndef SZX_BAIL_PROBE
#define SZX_BAIL_PROBE(site) ((void)(site))  // the host test build histograms the bail-out sites
#endif
#define SZX_BAIL(site) do { SZX_BAIL_PROBE(site); return ST_BAIL; } while (0)

// four copies of a statement with s = 0, 1, 2, 3 as literal constants (no loop: the slot arrays must be indexed
// by constants from the first compiler pass on, or they are demoted from registers to memory)
#define SZX_EACH(s, ...) { { const int s = 0; __VA_ARGS__ } { const int s = 1; __VA_ARGS__ } { const int s = 2; __VA_ARGS__ } { const int s = 3; __VA_ARGS__ } }
// the sweep's lambdas must be inlined: a call would take the address of the slot arrays
#define SZX_INLINE __attribute__((always_inline))
// value of slot `id`: all four are read (constant indices), then values -- never addresses -- are selected; a
// conditional read or write of a slot would be merged by the optimizer into one access through a selected address
#define SZX_PICK(arr, id, F) szx::pick4((id), arr[0] F, arr[1] F, arr[2] F, arr[3] F)
#define SZX_SET(lhs, cond, v) lhs = (cond) ? (v) : lhs

// One active edge = the current edge of one bound of one ring.  The four of them live in REGISTERS: slot id =
// ring * 2 + bound (0 left, 1 right), every access to the slot array uses a compile-time index (SZX_EACH or a
// select chain), so the sweep has no dependent memory round trips between its events.
template <class T>
SZ_HD T pick4(int id, T a0, T a1, T a2, T a3) { return id == 1 ? a1 : id == 2 ? a2 : id == 3 ? a3 : a0; }

struct CEdge {
  Pt bot, top;
  i64 curx;
  double slope;
  int vtop;   // ring index of the top vertex
  int dir;    // ring index step towards the ring's top vertex along this bound
  int ring;   // 0 = carries the output ring, -1 = not contributing
  int side;   // szv::SIDE_LEFT / SIDE_RIGHT of the output ring
  int wind2;  // inside (1) / outside (0) the other ring
  int last;   // 1 when top is the ring's top vertex (no further edge in this bound)
};
struct CCross {
  Pt pt;
  int e1, e2; // slot ids
};
struct Mono {         // one cleaned ring: vertex count, bottom / top vertex, direction of the left bound, bottom y
  int n, bot, top, dir_left;
  i64 bot_y;
};

// per-lane block of fast memory (LDS on the GPU): the vertices of both rings and the crossings of one scanbeam
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
  CCross X[MAX_CROSS];
  char tail[8];        // makes the size 8 x odd bytes: the same field of neighbouring lanes falls into different LDS banks
};
typedef FastBlockT<16> FastBlock;      // the step's fast pass: 64 lanes x 616 B of LDS per wave (four waves per CU); longer rings take the general sweep
static_assert(sizeof(FastBlock) == 616 && (sizeof(FastBlock) / 8) % 2 == 1, "FastBlock layout changed: keep the size 8 x odd");
typedef FastBlockT<32> FastBlockBig;   // second pass over the pairs whose rings have 17-32 vertices: 1128 B per lane, a few lanes per wave
static_assert(sizeof(FastBlockBig) == 1128 && (sizeof(FastBlockBig) / 8) % 2 == 1, "FastBlockBig layout changed: keep the size 8 x odd");

// MATLAB int64(v * 2^32): round half away from zero, saturating, NaN -> 0 (polyclip.m:63-66)
SZ_HD i64 to_fixed(double v) {
  double s = v * 4294967296.0;
  if (s != s) return 0;
  if (s >= 9223372036854775807.0) return 0x7FFFFFFFFFFFFFFFLL;
  if (s <= -9223372036854775808.0) return (i64)0x8000000000000000ULL;
  double t = __builtin_trunc(s);   // == (double)(int64)s in this range, without the two conversions
  double f = s - t;
  if (f >= 0.5) t += 1.0;
  else if (f <= -0.5) t -= 1.0;
  return (i64)t;
}
// the same for |v| < 2^30 m (FIXED_SAFE_METRES): no NaN, no saturation, so those branches are dropped
#define SZX_FIXED_SAFE_METRES 1.0e9
SZ_HD i64 to_fixed_in_range(double v) {
  double s = v * 4294967296.0;
  double t = __builtin_trunc(s);
  double f = s - t;
  if (f >= 0.5) t += 1.0;
  else if (f <= -0.5) t -= 1.0;
  return (i64)t;
}
// how the sweep reads vertex i of ring `poly` in fixed-point coordinates
struct VertsMetres { // the caller has checked that every coordinate, shifted or not, is within +-FIXED_SAFE_METRES
  int shifted; double shift_x, shift_y;   // ring 0 moved by this before the conversion (floe_interactions.m:148-149)
  template <class Blk> SZ_HD Pt operator()(const Blk &K, int poly, int i) const {
    double X = K.v.m.x[poly][i], Y = K.v.m.y[poly][i];
    if (poly == 0 && shifted) { X = X + shift_x; Y = Y + shift_y; }
    Pt p; p.x = to_fixed_in_range(X); p.y = to_fixed_in_range(Y);
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
  // (values, not references, are selected below: a reference chosen at run time would force both edges into memory)
  if (a.slope == 0 || b.slope == 0) {
    const bool va = (a.slope == 0);
    const i64 vx = va ? a.bot.x : b.bot.x;                       // the vertical edge
    const i64 ox = va ? b.bot.x : a.bot.x, oy = va ? b.bot.y : a.bot.y;
    const double os = va ? b.slope : a.slope;                    // the other one
    ip.x = vx;
    double k = (double)oy - ((double)ox / os);
    ip.y = szv::rnd((double)ip.x / os + k);
  } else {
    double ka = (double)a.bot.x - (double)a.bot.y * a.slope;
    double kb = (double)b.bot.x - (double)b.bot.y * b.slope;
    double q = (kb - ka) / (a.slope - b.slope);
    ip.y = szv::rnd(q);
    ip.x = (fa < fb) ? szv::rnd(a.slope * q + ka) : szv::rnd(b.slope * q + kb);
  }
  bool clamp = false, use_a = false;
  if (ip.y < a.top.y || ip.y < b.top.y) {
    ip.y = (a.top.y > b.top.y) ? a.top.y : b.top.y;
    clamp = true; use_a = (fa < fb);
  }
  if (ip.y > cur_y) {
    ip.y = cur_y;
    clamp = true; use_a = !(fa > fb);
  }
  if (clamp) {   // x of the flatter edge at the clamped y (TopX)
    const i64 tx = use_a ? a.top.x : b.top.x, ty = use_a ? a.top.y : b.top.y;
    const i64 bx = use_a ? a.bot.x : b.bot.x, by = use_a ? a.bot.y : b.bot.y;
    const double sl = use_a ? a.slope : b.slope;
    ip.x = (ip.y == ty) ? tx : bx + szv::rnd(sl * (double)(ip.y - by));
  }
  return ip;
}

// Checks that the general code would keep every vertex of both rings (closing duplicates stripped as
// clipper.cpp:1055-1058); finds each ring's single bottom / top vertex and which way its left bound runs.
// 0, or the reason for the bail-out.
template <class Blk, class Verts>
SZ_HD int scan_rings(const Blk &K, const Verts &vert, int na, int nb, Mono &m0, Mono &m1) {
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
    Mono m; m.n = n; m.bot = bot; m.top = top; m.dir_left = (s_next < s_prev) ? -1 : +1;
    m.bot_y = at_bot.y;
    if (poly == 0) m0 = m; else m1 = m;
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
  Mono m0, m1;
  if (int why = scan_rings(K, vert, na, nb, m0, m1)) SZX_BAIL(30 + why);
  CEdge ed[4];                 // slot = ring * 2 + bound; valid while its bit in `act` is set
  SZX_EACH(s, ed[s].bot.x = ed[s].bot.y = ed[s].top.x = ed[s].top.y = ed[s].curx = 0; ed[s].slope = 0; ed[s].vtop = ed[s].dir = ed[s].wind2 = ed[s].last = 0; ed[s].ring = -1; ed[s].side = szv::SIDE_NONE;)
  int ord[4] = {0, 0, 0, 0};   // slot ids in active-edge-list order, k of them
  int k = 0, act = 0;
  int pending = 3;             // bit p: ring p not yet inserted
  Deque q;
  q.d = w.deq; q.cap_half = w.cap_half; q.lo = q.hi = w.cap_half; q.open = false; q.used = false; q.overflow = false;
  q.front.x = q.front.y = q.back.x = q.back.y = 0;
  i64 y = 0;                   // current scanline (unused until the first ring is inserted)

  // ---- the register file by run-time index: field-by-field select chains, no memory ----
  auto get = [&](int id) SZX_INLINE {
    CEdge r;
    r.bot.x = SZX_PICK(ed, id, .bot.x); r.bot.y = SZX_PICK(ed, id, .bot.y);
    r.top.x = SZX_PICK(ed, id, .top.x); r.top.y = SZX_PICK(ed, id, .top.y);
    r.curx = SZX_PICK(ed, id, .curx); r.slope = SZX_PICK(ed, id, .slope);
    r.vtop = SZX_PICK(ed, id, .vtop); r.dir = SZX_PICK(ed, id, .dir); r.ring = SZX_PICK(ed, id, .ring);
    r.side = SZX_PICK(ed, id, .side); r.wind2 = SZX_PICK(ed, id, .wind2); r.last = SZX_PICK(ed, id, .last);
    return r;
  };
  auto put_state = [&](int id, const CEdge &r) SZX_INLINE {
    SZX_EACH(s, SZX_SET(ed[s].ring, s == id, r.ring); SZX_SET(ed[s].side, s == id, r.side); SZX_SET(ed[s].wind2, s == id, r.wind2);)
  };
  auto put = [&](int id, const CEdge &r) SZX_INLINE {
    SZX_EACH(s,
      SZX_SET(ed[s].bot.x, s == id, r.bot.x); SZX_SET(ed[s].bot.y, s == id, r.bot.y);
      SZX_SET(ed[s].top.x, s == id, r.top.x); SZX_SET(ed[s].top.y, s == id, r.top.y);
      SZX_SET(ed[s].curx, s == id, r.curx); SZX_SET(ed[s].slope, s == id, r.slope);
      SZX_SET(ed[s].vtop, s == id, r.vtop); SZX_SET(ed[s].dir, s == id, r.dir); SZX_SET(ed[s].ring, s == id, r.ring);
      SZX_SET(ed[s].side, s == id, r.side); SZX_SET(ed[s].wind2, s == id, r.wind2); SZX_SET(ed[s].last, s == id, r.last);)
  };
  auto put_geometry = [&](int id, const CEdge &r) SZX_INLINE {
    SZX_EACH(s,
      SZX_SET(ed[s].bot.x, s == id, r.bot.x); SZX_SET(ed[s].bot.y, s == id, r.bot.y);
      SZX_SET(ed[s].top.x, s == id, r.top.x); SZX_SET(ed[s].top.y, s == id, r.top.y);
      SZX_SET(ed[s].curx, s == id, r.curx); SZX_SET(ed[s].slope, s == id, r.slope);
      SZX_SET(ed[s].vtop, s == id, r.vtop); SZX_SET(ed[s].last, s == id, r.last);)
  };
  auto curx_of = [&](int id) SZX_INLINE { return SZX_PICK(ed, id, .curx); };
  auto ord_at = [&](int pos) SZX_INLINE { return SZX_PICK(ord, pos, ); };
  auto pos_of = [&](int id) SZX_INLINE { int v = -1; SZX_EACH(s, if (s < k && ord[s] == id) v = s;) return v; };
  auto mono_of = [&](int poly) SZX_INLINE { return poly ? m1 : m0; };
  // steps an edge to the next edge of its bound (the general code's UpdateEdgeIntoAEL)
  auto advance = [&](CEdge &e, int poly) SZX_INLINE {
    const Mono m = mono_of(poly);
    int vt = e.vtop + e.dir;
    if (vt < 0) vt += m.n; else if (vt >= m.n) vt -= m.n;
    e.bot = e.top; e.top = vert(K, poly, vt); e.vtop = vt;
    e.last = (vt == m.top);
    cset_slope(e);
  };

  bool done_early = false;
  for (int guard = 0; guard < 64; ++guard) {
    // Scanbeams with the edges of only one ring cannot cross or emit anything (its edges are outside the
    // other ring, winding 0), in the general sweep as here: skip them.  (1) Only the lower ring is in and the
    // upper one is still pending: walk the two bounds up to the scanline of the pending ring's bottom; if the
    // ring ends first the y-ranges are disjoint and the result is empty.  (2) One ring has ended and no output
    // ring is open: nothing more can happen.
    if (k == 2 && pending != 0) {
      const int pp = (pending & 1) ? 0 : 1, pin = 1 - pp;
      const i64 by = mono_of(pp).bot_y;
      bool ended = false;
      for (int b = 0; b < 2 && !ended; ++b) {
        CEdge e = get(pin * 2 + b);
        while (e.top.y > by) {
          if (e.last) { ended = true; break; }
          advance(e, pin);
        }
        put(pin * 2 + b, e);
      }
      if (ended) return szv::ST_OK;
      y = by + 1;                 // any scanline below by: cross points are clamped to it only from above
    }
    if (k == 2 && pending == 0 && !q.open) { done_early = true; break; }

    // next scanline: the highest y still below the current one
    bool have = false;
    i64 top_y = 0;
    int src_poly = -1;
    SZX_EACH(s, if (act >> s & 1) {
      const i64 ty = ed[s].top.y;
      if (!have || ty > top_y) { have = true; top_y = ty; src_poly = s >> 1; }
      else if (ty == top_y && (s >> 1) != src_poly) SZX_BAIL(4);
    })
    int insert = -1;            // ring whose local minimum lies on the next scanline
    for (int p = 0; p < 2; ++p) {
      if (!(pending >> p & 1)) continue;
      const i64 by = mono_of(p).bot_y;
      if (!have || by > top_y) { have = true; top_y = by; insert = p; src_poly = -1; }
      else if (by == top_y) SZX_BAIL(5);
    }
    if (!have) break;
    // two edges of one ring may only end together at that ring's top vertex
    {
      int ends = 0, lasts = 0;
      SZX_EACH(s, if ((act >> s & 1) && ed[s].top.y == top_y) { ++ends; lasts += ed[s].last; })
      if (ends > 1 && (ends != 2 || lasts != 2)) SZX_BAIL(6);
    }

    // ---- crossings inside the scanbeam (clipper.cpp:2856-2955) ----
    SZX_EACH(s, { const i64 cx = cx_at(ed[s], top_y); SZX_SET(ed[s].curx, (act >> s & 1) != 0, cx); })
    int nX = 0;
    int sl[4];
    SZX_EACH(s, sl[s] = ord[s];)
    {
      int ns = k;
      bool modified;
      do {
        modified = false;
        SZX_EACH(i, if (i < 3 && i + 1 < ns) {
            const int ia = sl[i], ib = sl[(i + 1) & 3];
            const i64 xa = curx_of(ia), xb = curx_of(ib);
            if (xa > xb) {
              if ((ia >> 1) == (ib >> 1)) SZX_BAIL(7);        // bounds of one ring do not cross
              const CEdge e = get(ia), en = get(ib);
              if (e.slope == en.slope) SZX_BAIL(8);
              if (nX >= MAX_CROSS) SZX_BAIL(9);
              Pt pt = ccross_point(e, en, y);
              if (pt.y < top_y) { pt.x = e.curx; pt.y = top_y; }
              K.X[nX].e1 = ia; K.X[nX].e2 = ib; K.X[nX].pt = pt; ++nX;
              sl[i] = ib; sl[(i + 1) & 3] = ia;   // (constant indices: this block is instantiated per i)
              modified = true;
            } else if (xa == xb && (ia >> 1) != (ib >> 1)) SZX_BAIL(10); // tie between the two rings
          })                                                 // (after a swap the same edge keeps moving right)
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
      SZX_EACH(s, sl[s] = ord[s];)
      for (int a = 0; a < nX; ++a) {
        const int i1 = K.X[a].e1, i2 = K.X[a].e2;
        int p1 = -1, p2 = -1;
        SZX_EACH(s, if (s < k) { if (sl[s] == i1) p1 = s; if (sl[s] == i2) p2 = s; })
        if (p1 - p2 != 1 && p2 - p1 != 1) SZX_BAIL(11);      // the general code would reorder: leave it to it
        SZX_EACH(s, sl[s] = (s == p1) ? i2 : (s == p2) ? i1 : sl[s];)
      }
    }
    for (int a = 0; a < nX; ++a) {
      // Clipper::IntersectEdges for two closed edges of different rings, even-odd, intersection
      const int i1 = K.X[a].e1, i2 = K.X[a].e2;
      const Pt pt = K.X[a].pt;
      CEdge e1 = get(i1), e2 = get(i2);
      e1.wind2 = (e1.wind2 == 0) ? 1 : 0;
      e2.wind2 = (e2.wind2 == 0) ? 1 : 0;
      const bool c1 = e1.ring >= 0, c2 = e2.ring >= 0;
      if (c1 || c2) {
        // both: AddLocalMaxPoly, the ring closes here; one: that edge emits and hands ring and side over
        dq_emit(q, c1 ? e1.side : e2.side, pt);
        if (c1 && c2) { e1.ring = e2.ring = -1; q.open = false; }
        else {
          int sd = e1.side; e1.side = e2.side; e2.side = sd;
          int rg = e1.ring; e1.ring = e2.ring; e2.ring = rg;
        }
      } else {                              // AddLocalMinPoly(e1, e2): a ring starts at a crossing
        if (q.used) SZX_BAIL(12);
        if (!(e1.slope < e2.slope)) SZX_BAIL(13);   // e1 crosses from left to right: its slope is the smaller
        dq_start(q, pt);
        e1.ring = e2.ring = 0;
        e1.side = szv::SIDE_RIGHT; e2.side = szv::SIDE_LEFT;
      }
      put_state(i1, e1); put_state(i2, e2);
      const int p1 = pos_of(i1), p2 = pos_of(i2);
      SZX_EACH(s, ord[s] = (s == p1) ? i2 : (s == p2) ? i1 : ord[s];)
    }

    // ---- top of the scanbeam (clipper.cpp:3009-3113); curx of every edge is already x(top_y) ----
    // One vertex per scanline: either the top vertex of a ring (its two edges leave together, DoMaxima) or an
    // intermediate vertex of one bound.
    {
      int ending = -1, n_end = 0, n_last = 0;
      SZX_EACH(s, if ((act >> s & 1) && ed[s].top.y == top_y) { ending = s; ++n_end; n_last += ed[s].last; })
      if (n_end > 0 && n_last > 0) {
        // DoMaxima: the two top edges of the ring must be neighbours in the active list
        if (n_end != 2 || n_last != 2) SZX_BAIL(15);
        const int pm = ending >> 1;
        const int pa = pos_of(pm * 2), pb = pos_of(pm * 2 + 1);
        if (pa < 0 || pb < 0 || (pa - pb != 1 && pb - pa != 1)) SZX_BAIL(14);
        const int pf = pa < pb ? pa : pb;
        const CEdge e = get(ord_at(pf)), mp = get(ord_at(pf + 1));
        if (e.ring >= 0 && mp.ring >= 0) { dq_emit(q, e.side, e.top); q.open = false; }
        else if (e.ring >= 0 || mp.ring >= 0) SZX_BAIL(16);   // the general code reports an error here
        SZX_EACH(s, SZX_SET(ed[s].ring, (s >> 1) == pm, -1);)
        SZX_EACH(s, SZX_SET(ord[s], s >= pf && s + 2 < 4, ord[s + 2 < 4 ? s + 2 : 3]);)
        k -= 2;
        act &= ~(3 << (pm * 2));
      } else if (n_end == 1) {
        // an intermediate vertex of a bound: emit it and step to the next edge of the bound
        CEdge e = get(ending);
        if (e.ring >= 0) dq_emit(q, e.side, e.top);
        advance(e, ending >> 1);
        e.curx = e.bot.x;
        put_geometry(ending, e);          // ring, side, winding and direction stay
        // a neighbour of the other ring passing through this vertex is the join case of the general code
        const int pe = pos_of(ending);
        if (pe > 0 && curx_of(ord_at(pe - 1)) == e.curx) SZX_BAIL(17);
        if (pe + 1 < k && curx_of(ord_at(pe + 1)) == e.curx) SZX_BAIL(18);
      }
    }
    y = top_y;

    // ---- insertion of a ring's local minimum on this scanline (clipper.cpp:1978-2077) ----
    if (insert >= 0) {
      const Mono m = mono_of(insert);
      const Pt bot = vert(K, insert, m.bot);
      int pos = 0;
      bool tie = false, found = false;
      SZX_EACH(s, if (s < k && !found) {
        const i64 ox = curx_of(ord[s]);
        if (ox == bot.x) tie = true;
        if (bot.x < ox) found = true; else pos = s + 1;
      })
      if (tie) SZX_BAIL(19);                  // a tie: vertex on an edge or coincident vertices
      SZX_SET(ord[3], 1 >= pos, ord[1]);
      SZX_SET(ord[2], 0 >= pos, ord[0]);
      // even-odd winding against the other ring = parity of its edges to the left (clipper.cpp:1624-1729);
      // inside the other ring -> contributing -> AddLocalMinPoly (clipper.cpp:1841-1882)
      const int w2 = pos & 1;
      if (w2) {
        if (q.used) SZX_BAIL(3);              // a second output ring: not the plain case
        dq_start(q, bot);
      }
      for (int bound = 0; bound < 2; ++bound) {
        const int id = insert * 2 + bound;
        SZX_EACH(s, SZX_SET(ord[s], s == pos + bound, id);)
        CEdge e;
        e.dir = (bound == 0) ? m.dir_left : -m.dir_left;
        int vt = m.bot + e.dir;
        if (vt < 0) vt += m.n; else if (vt >= m.n) vt -= m.n;
        e.bot = bot; e.top = vert(K, insert, vt); e.vtop = vt;
        e.last = (vt == m.top);
        cset_slope(e);
        e.curx = bot.x;
        e.wind2 = w2;
        e.ring = w2 ? 0 : -1;
        e.side = w2 ? (bound == 0 ? szv::SIDE_LEFT : szv::SIDE_RIGHT) : szv::SIDE_NONE;
        put(id, e);
      }
      k += 2;
      act |= 3 << (insert * 2);
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
