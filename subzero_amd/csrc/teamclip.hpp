// teamclip.hpp -- the INTERSECTION (even-odd) of two single-minimum rings by a TEAM of lanes (team.hpp), with the
// output of the reference's Clipper 6.4.2 (private/clipper.cpp:1560-3217): same ring, same first vertex and direction,
// same rounded crossing points.
//
// convex.hpp walks the scanbeams of such a pair one after the other in one lane.  Nothing in that walk depends on
// history: with one vertex per scanline and no ties in x (anything else bails out), the state the sweep holds at a
// scanline is a function of the scanline alone --
//   * which edge of each of the four bounds (left / right bound of ring 0 / 1) is active: the number of that bound's
//     vertices above the scanline (a prefix count over the vertices of both rings sorted by y);
//   * the order of the active edges: the order of their x at the scanline (clipper.cpp:2856-2902 sorts them so);
//   * which edges contribute and to which side of the output ring: an edge of one ring contributes iff an odd number
//     of the other ring's edges lie to its left (even-odd winding, clipper.cpp:1624-1729), and a contributing left
//     bound is the left side of the output, a contributing right bound its right side;
//   * the crossings inside a scanbeam: the pairs of edges whose order differs between its two scanlines, each point
//     by clipper.cpp:622-689 on the two whole edges.
// So every lane takes one scanline (= one vertex of one of the rings), derives the state just below and at it,
// finds the crossings of its beam and what it emits -- crossing points in the order of their y, then the vertex --
// and two prefix sums over the lanes put the emitted points where the sweep's front / back insertions would have put
// them (clipper.cpp:1890-1930 AddOutPt, :3199-3217 BuildResult).  The crossing points themselves (two divisions each)
// are computed by as many lanes as there are crossings.
//
// Like convex.hpp it NEVER guesses: ST_BAIL (nothing written) for everything outside the plain case -- what
// convex.hpp bails out for, and a little more where telling the cases apart would cost more than it saves (any two
// vertices on one scanline, any tie in x between active edges at a scanline, two crossings of one beam at the same y,
// duplicate or collinear output points).  The caller hands such a pair to the
// lane-per-pair paths.  tests/ check it bit for bit against the reference's Clipper build (oracle/_ref).
#pragma once
#include "team.hpp"
#include "convex.hpp"

namespace sztm {

using szv::Pt;
using szv::same;

enum { T_NV = 16, T_EV = 2 * T_NV, T_CAPX = 8, T_CAPV = 16 };   // vertices per ring, scanlines, crossings of all beams, vertices of the output ring

#ifndef SZTM_BAIL_PROBE
#define SZTM_BAIL_PROBE(site) ((void)(site))   // the host test build histograms the bail-out sites here
#endif
#ifndef SZTM_PHASE
#define SZTM_PHASE(k) ((void)0)   // a diagnostic build attributes the kernel's time to its parts here (narrow_team.hip)
#endif
enum { ST_WIDE = 104 };   // more vertices in the two rings than the team has lanes
#define SZTM_BAIL(site) do { if (T.lane == 0) { SZTM_BAIL_PROBE(site); } return (int)szx::ST_BAIL; } while (0)

// the team's scratch for one clip (LDS): both rings in fixed point, the slope of every ring edge, the vertices of
// both rings in sweep order with the packed counts of bound vertices above each, the crossings of all beams
struct ClipScratch {
  i64 fx[2][T_NV], fy[2][T_NV];
  double slope[2][T_NV];          // of the edge from vertex l to l + 1 (cleaned ring)
  Pt xpt[T_CAPX];
  int xinfo[T_CAPX];              // event | a << 8 | b << 9 | (ring 0's edge is the left one below) << 10 | (slope e1 < slope e2) << 11 | parallel << 12
  int pc[T_EV];                   // event s: four 8-bit counts (slot = ring * 2 + bound) of bound vertices above it
  unsigned char ev[T_EV];         // event s: ring << 4 | vertex
  int rank_bot[2], rank_top[2];
};

struct Bounds { int n, bot, top, dirl, ptop; };   // cleaned ring: vertex count, bottom / top vertex, direction of the left bound, position of the top along it

// a*b == c*d (operands: coordinate differences) -- the exact 128-bit test only when doubles cannot tell
SZ_HD bool collinear3_filtered(const Pt &p1, const Pt &p2, const Pt &p3) {
  const double A = (double)(p1.y - p2.y) * (double)(p2.x - p3.x), C = (double)(p1.x - p2.x) * (double)(p2.y - p3.y);
  const double diff = szv::dabs(A - C), mag = szv::dabs(A) > szv::dabs(C) ? szv::dabs(A) : szv::dabs(C);
  if (diff > mag * 1e-14) return false;   // each product carries <= 4 roundings of 2^-53: equal exact products differ by < 1e-15 mag
  return szv::collinear3(p1, p2, p3);
}

// the edge of bound `b` of a ring that is active just below a scanline with `cnt` of the bound's vertices above it
SZ_HD bool active_edge(const ClipScratch *S, const Bounds &g, int r, int b, int cnt, szx::CEdge &e) {
  const int len = (b == 0) ? g.ptop + 1 : g.n - g.ptop + 1;
  if (cnt < 1 || cnt > len - 1) return false;
  const int dirb = (b == 0) ? g.dirl : -g.dirl;
  int vb = g.bot + dirb * (cnt - 1), vt = g.bot + dirb * cnt;
  if (vb < 0) vb += g.n; else if (vb >= g.n) vb -= g.n;
  if (vt < 0) vt += g.n; else if (vt >= g.n) vt -= g.n;
  e.bot.x = S->fx[r][vb]; e.bot.y = S->fy[r][vb];
  e.top.x = S->fx[r][vt]; e.top.y = S->fy[r][vt];
  e.slope = S->slope[r][dirb > 0 ? vb : vt];
  e.last = (cnt == len - 1);
  return true;
}
SZ_HD int pc_field(int pc, int slot) { return (pc >> (8 * slot)) & 0xff; }
// the record of ring r by value (a run-time index into the pair of records would put them into scratch memory)
SZ_HD Bounds ring_of(const Bounds *g, int r) {
  Bounds b;
  b.n = r ? g[1].n : g[0].n; b.bot = r ? g[1].bot : g[0].bot; b.top = r ? g[1].top : g[0].top;
  b.dirl = r ? g[1].dirl : g[0].dirl; b.ptop = r ? g[1].ptop : g[0].ptop;
  return b;
}

// S->fx / fy [r][0 .. n_r) hold the rings (a closing duplicate may follow the stored vertices), visible to the team.
// ST_OK: *out_n vertices in ox / oy (0 = no overlap); ST_BAIL: not the plain case -- or more vertices in the two rings
// together than the team has lanes (one scanline per lane).  Every lane returns the same.
template <int G>
SZ_TEAM int team_clip(const Team<G> &T, ClipScratch *S, int na, int nb, i64 *ox, i64 *oy, int *out_n) {
  SZTM_PHASE(10);
  *out_n = 0;
  Bounds g[2];
  // ---- closing duplicates (clipper.cpp:1055-1058), uniform over the team ----
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    int hi = (r ? nb : na) - 1;
    if (hi + 1 > (int)T_NV) SZTM_BAIL(2);
    if (hi < 2) SZTM_BAIL(1);
    const i64 x0 = S->fx[r][0], y0 = S->fy[r][0];
    while (hi > 0 && S->fx[r][hi] == x0 && S->fy[r][hi] == y0) --hi;
    while (hi > 0 && S->fx[r][hi] == S->fx[r][hi - 1] && S->fy[r][hi] == S->fy[r][hi - 1]) --hi;
    if (hi < 2) SZTM_BAIL(1);
    g[r].n = hi + 1;
  }
  const int n0 = g[0].n, M = g[0].n + g[1].n;
  if (M > G) { if (T.lane == 0) { SZTM_BAIL_PROBE(43); } return (int)ST_WIDE; }   // (the caller sends such a pair to a wider team)
  SZTM_PHASE(11);
  // ---- every vertex (lanes 0 .. n0-1: ring 0, the next n1: ring 1): kept by the general code? local minimum /
  // maximum? slope of its edge ----
  const int v_r = (T.lane >= n0) ? 1 : 0, v_l = T.lane - (v_r ? n0 : 0);
  const bool v_valid = T.lane < M;
  i64 v_y = 0;
  {
    int why = 0;
    bool is_min = false, is_max = false;
    if (v_valid) {
      const int n = v_r ? g[1].n : g[0].n, l = v_l;
      const int lp = (l == 0) ? n - 1 : l - 1, ln = (l + 1 == n) ? 0 : l + 1;
      Pt c, p, q;
      c.x = S->fx[v_r][l]; c.y = S->fy[v_r][l]; p.x = S->fx[v_r][lp]; p.y = S->fy[v_r][lp]; q.x = S->fx[v_r][ln]; q.y = S->fy[v_r][ln];
      v_y = c.y;
      const i64 lim = 0x3FFFFFFFFFFFFFFFLL;
      if (c.x > lim || c.y > lim || -c.x > lim || -c.y > lim) why = 3;
      else if (same(c, q) || collinear3_filtered(p, c, q)) why = 4;
      else if (c.y == q.y) why = 5;
      else {
        S->slope[v_r][l] = (double)(q.x - c.x) / (double)(q.y - c.y);
        is_min = c.y > p.y && c.y > q.y;
        is_max = c.y < p.y && c.y < q.y;
      }
    }
    const unsigned bad = T.ballot(why != 0, TEAM_SITE), mn = T.ballot(is_min, TEAM_SITE), mx = T.ballot(is_max, TEAM_SITE);
    if (bad) SZTM_BAIL(34);
    const unsigned lo = (1u << n0) - 1u;
    const unsigned mins[2] = {mn & lo, mn >> n0}, maxs[2] = {mx & lo, mx >> n0};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (__builtin_popcount(mins[r]) != 1 || __builtin_popcount(maxs[r]) != 1) SZTM_BAIL(36);
      g[r].bot = __builtin_ctz(mins[r]); g[r].top = __builtin_ctz(maxs[r]);
    }
  }
  T.sync(TEAM_SITE);
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    // the left bound starts with the edge of larger slope (clipper.cpp:1194-1204)
    const int n = g[r].n, bot = g[r].bot;
    const double s_next = S->slope[r][bot], s_prev = S->slope[r][bot == 0 ? n - 1 : bot - 1];
    if (s_next == s_prev) SZTM_BAIL(37);
    g[r].dirl = (s_next < s_prev) ? -1 : +1;
    int pt = (g[r].dirl > 0) ? g[r].top - bot : bot - g[r].top;
    if (pt < 0) pt += n;
    g[r].ptop = pt;
  }
  SZTM_PHASE(12);
  // ---- sweep order: rank of every vertex by descending y; which bounds it lies on ----
  {
    int rank = 0, ties = 0;
#pragma unroll
    for (int r = 0; r < 2; ++r)
      for (int k = 0; k < g[r].n; ++k) {
        const i64 y = S->fy[r][k];
        rank += (y > v_y) ? 1 : 0; ties += (y == v_y) ? 1 : 0;
      }
    if (T.any(v_valid && ties != 1, TEAM_SITE)) SZTM_BAIL(40);   // two vertices on one scanline
    if (v_valid) {
      const Bounds gr = ring_of(g, v_r);
      int pl = (gr.dirl > 0) ? v_l - gr.bot : gr.bot - v_l;      // position along the left bound
      if (pl < 0) pl += gr.n;
      const bool onL = pl <= gr.ptop, onR = pl >= gr.ptop || pl == 0;
      S->ev[rank] = (unsigned char)(v_r << 4 | v_l);
      S->pc[rank] = (onL ? 1 << (8 * (v_r * 2)) : 0) | (onR ? 1 << (8 * (v_r * 2 + 1)) : 0);
      if (v_l == gr.bot) S->rank_bot[v_r] = rank;
      if (v_l == gr.top) S->rank_top[v_r] = rank;
    }
  }
  T.sync(TEAM_SITE);
  SZTM_PHASE(13);
  // ---- lane s takes scanline s.  Bound vertices above it: exclusive prefix sum of the packed flags in sweep order ----
  const int s = T.lane;
  int pcs;
  {
    int tot;
    pcs = T.scan_excl((s < M) ? S->pc[s] : 0, &tot, TEAM_SITE);
    if (s < M) S->pc[s] = pcs;                                // (the lanes that compute the crossing points read it)
  }
  T.sync(TEAM_SITE);
  const int s_ins = S->rank_bot[0] > S->rank_bot[1] ? S->rank_bot[0] : S->rank_bot[1];   // the upper ring's local minimum
  const int s_end = S->rank_top[0] < S->rank_top[1] ? S->rank_top[0] : S->rank_top[1];   // the first local maximum
  if (s_end < s_ins) return szv::ST_OK;                       // the y-ranges are disjoint: no overlap
  SZTM_PHASE(14);
  // ---- every scanline in the common range: active edges below it, their order below and at it, the crossings ----
  // st: bits 0-7 order below (slot at position p in bits 2p, 2p+1), 8-11 crossed (combo = a * 2 + b), 12-15 ring 0's
  // edge is the left one below, 16 the ring inserted here starts the output ring, 17 the scanline is in range
  int st = 0;
  const bool in_range = s < M && s >= s_ins && s <= s_end;
  const int evs = S->ev[s & (T_EV - 1)], e_r = evs >> 4, e_l = evs & 15;
  {
    int why = 0;
    if (in_range) {
      st = 1 << 17;
      const i64 ys = S->fy[e_r][e_l];
      if (s == s_ins) {
        // insertion of ring r's local minimum between the two edges of the other ring (clipper.cpp:1978-2077)
        const int ro = 1 - e_r;
        const Bounds go = ring_of(g, ro);
        szx::CEdge e0, e1;
        if (!active_edge(S, go, ro, 0, pc_field(pcs, ro * 2), e0) || !active_edge(S, go, ro, 1, pc_field(pcs, ro * 2 + 1), e1)) why = 50;
        else {
          const i64 x0 = szx::cx_at(e0, ys), x1 = szx::cx_at(e1, ys), bx = S->fx[e_r][e_l];
          if (x0 > x1) why = 51;
          else if (bx == x0 || bx == x1) why = 19;            // a tie: vertex on an edge or coincident vertices
          else if (x0 < bx && bx < x1) st |= 1 << 16;         // inside the other ring -> contributing -> AddLocalMinPoly
        }
      } else {
        const int evp = S->ev[(s - 1) & (T_EV - 1)];
        const i64 yp = S->fy[evp >> 4][evp & 15];
        i64 xb[4], xt[4];
        bool startsb[4], endst[4];
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          szx::CEdge e;
          if (!active_edge(S, g[k >> 1], k >> 1, k & 1, pc_field(pcs, k), e)) { ok = false; xb[k] = xt[k] = 0; startsb[k] = endst[k] = false; continue; }
          xb[k] = e.bot.x + szv::rnd(e.slope * (double)(yp - e.bot.y));
          xt[k] = szx::cx_at(e, ys);
          startsb[k] = (e.bot.y == yp); endst[k] = (e.top.y == ys);
        }
        if (!ok) why = 52;
        // the two bounds of one ring do not cross; they meet only at the ring's own bottom / top vertex
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int L = 2 * q, R = 2 * q + 1;
          if (xb[L] > xb[R] || xt[L] > xt[R]) why = 7;
          if (xb[L] == xb[R] && !(startsb[L] && startsb[R])) why = 53;
          if (xt[L] == xt[R] && !(endst[L] && endst[R])) why = 53;
        }
        int crossed = 0, left0 = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int a = c >> 1, b = 2 + (c & 1);
          if (xb[a] == xb[b] || xt[a] == xt[b]) why = 10;      // a tie between the two rings
          const bool lb = xb[a] < xb[b], lt = xt[a] < xt[b];
          if (lb != lt) crossed |= 1 << c;
          if (lb) left0 |= 1 << c;
        }
        int order = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          int pos = 0;
#pragma unroll
          for (int m = 0; m < 4; ++m) pos += (xb[m] < xb[k] || (xb[m] == xb[k] && m < k)) ? 1 : 0;
          order |= k << (2 * pos);
        }
        st |= order | crossed << 8 | left0 << 12;
      }
    }
    const unsigned bad = T.ballot(why != 0, TEAM_SITE);
    if (bad) { const int w = T.shfl(why, __builtin_ctz(bad), TEAM_SITE); SZTM_BAIL(w); }
  }
  SZTM_PHASE(15);
  // ---- the crossings of all beams, one lane each (clipper.cpp:622-689, :2868-2876) ----
  int nX, xbase;
  {
    const int crossed = (st >> 8) & 15;
    xbase = T.scan_excl(__builtin_popcount(crossed), &nX, TEAM_SITE);
    if (nX > (int)T_CAPX) SZTM_BAIL(9);
    int k = xbase;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (crossed >> c & 1) { S->xinfo[k & (T_CAPX - 1)] = s | c << 8 | ((st >> (12 + c)) & 1) << 10; ++k; }
  }
  if (nX > 0) {
    T.sync(TEAM_SITE);
    if (T.lane < nX) {
      const int k = T.lane, info = S->xinfo[k], xs = info & 0xff, c = (info >> 8) & 3;
      const int ev1 = S->ev[xs], ev0 = S->ev[xs - 1], pcx = S->pc[xs];
      const i64 ys = S->fy[ev1 >> 4][ev1 & 15], yp = S->fy[ev0 >> 4][ev0 & 15];
      szx::CEdge ea, eb;
      (void)active_edge(S, g[0], 0, c >> 1, pc_field(pcx, c >> 1), ea);
      (void)active_edge(S, g[1], 1, c & 1, pc_field(pcx, 2 + (c & 1)), eb);
      const bool a_left = (info >> 10) & 1;
      const szx::CEdge e1 = a_left ? ea : eb, e2 = a_left ? eb : ea;
      int extra = 0;
      Pt pt; pt.x = 0; pt.y = 0;
      if (e1.slope == e2.slope) extra |= 1 << 12;
      else {
        pt = szx::ccross_point(e1, e2, yp);
        if (pt.y < ys) { pt.x = szx::cx_at(e1, ys); pt.y = ys; }
      }
      if (e1.slope < e2.slope) extra |= 1 << 11;
      S->xpt[k] = pt;
      S->xinfo[k] = info | extra;
    }
    T.sync(TEAM_SITE);
  }
  SZTM_PHASE(16);
  // ---- what every scanline emits: the crossings of its beam in the order of their y, then its vertex ----
  // One byte per emission, in order: bits 0-3 the crossing (index into xpt), 4-5 the side (0: front of the output ring =
  // its left side, 1: back, 2: the point the ring starts with), 6 it closes the ring, 7 it is the scanline's vertex.
  unsigned long long em = 0;
  int em_n = 0;
  {
    int why = 0;
    if (in_range && s == s_ins) {
      if (st >> 16 & 1) { em = 0x80 | 2 << 4; em_n = 1; }
    } else if (in_range) {
      int order = st & 0xff;
      const int crossed = (st >> 8) & 15, nc = __builtin_popcount(crossed);
      // contributing: an odd number of the other ring's edges to the left (position p in `order`)
      auto contributes = [&](int p) SZX_INLINE {
        const int me = (order >> (2 * p)) & 3;
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) if (q < p && ((((order >> (2 * q)) & 3) >> 1) != (me >> 1))) ++cnt;
        return (cnt & 1) != 0;
      };
      auto pos_of = [&](int slot) SZX_INLINE {
        int p = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) if (((order >> (2 * q)) & 3) == slot) p = q;
        return p;
      };
      if (nc > 0) {
        // bottom-most first (clipper.cpp:2918-2922); two crossings of one beam at the same y are ordered by the
        // general code's sort: left to it
        int ks[4]; i64 ky[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ks[i] = (i < nc) ? xbase + i : -1;
          ky[i] = (i < nc) ? S->xpt[(xbase + i) & (T_CAPX - 1)].y : (i64)0x8000000000000000ULL;
        }
#define SZTM_CSWAP(a, b) { if (ky[b] > ky[a]) { const i64 ty = ky[a]; ky[a] = ky[b]; ky[b] = ty; const int tk = ks[a]; ks[a] = ks[b]; ks[b] = tk; } }
        if (nc > 1) { SZTM_CSWAP(0, 1) SZTM_CSWAP(2, 3) SZTM_CSWAP(0, 2) SZTM_CSWAP(1, 3) SZTM_CSWAP(1, 2) }
#undef SZTM_CSWAP
#pragma unroll
        for (int i = 0; i < 3; ++i) if (i + 1 < nc && ky[i] == ky[i + 1]) why = 41;
        for (int i = 0; i < nc && !why; ++i) {
          int k = ks[0];
#pragma unroll
          for (int q = 1; q < 4; ++q) if (q == i) k = ks[q];
          const int info = S->xinfo[k & (T_CAPX - 1)], c = (info >> 8) & 3;
          if (info >> 12 & 1) { why = 8; break; }
          const int sa = c >> 1, sb = 2 + (c & 1);
          const bool a_left = (info >> 10) & 1;
          const int e1 = a_left ? sa : sb, e2 = a_left ? sb : sa;
          const int p1 = pos_of(e1), p2 = pos_of(e2);
          if (p2 != p1 + 1) { why = 11; break; }               // the general code would reorder: leave it to it
          const bool c1 = contributes(p1), c2 = contributes(p2);
          int rec = k;
          if (c1 && c2) {                                      // AddLocalMaxPoly: the ring closes here
            if ((e1 & 1) != 0 || (e2 & 1) != 1) { why = 54; break; }
            rec |= (e1 & 1) << 4 | 1 << 6;
          } else if (c1 || c2) {                               // the contributing edge emits and hands its side over
            if ((e1 & 1) != (e2 & 1)) { why = 55; break; }
            rec |= (e1 & 1) << 4;
          } else {                                             // AddLocalMinPoly: the ring starts at a crossing
            if (!(info >> 11 & 1)) { why = 13; break; }        // e1 crosses from left to right: its slope is the smaller
            if ((e1 & 1) != 1 || (e2 & 1) != 0) { why = 56; break; }
            rec |= 2 << 4;
          }
          em |= (unsigned long long)rec << (8 * em_n);
          ++em_n;
          order = (order & ~(15 << (2 * p1))) | e2 << (2 * p1) | e1 << (2 * p2);
        }
      }
      if (!why) {
        // the vertex on this scanline (clipper.cpp:3009-3113): the top of its ring, or an intermediate vertex of a bound
        const Bounds gr = ring_of(g, e_r);
        if (e_l == gr.top) {
          const int pa = pos_of(e_r * 2), pb = pos_of(e_r * 2 + 1);
          if (pb != pa + 1) why = 14;
          else if (contributes(pa)) { em |= (unsigned long long)(0x80 | 1 << 6) << (8 * em_n); ++em_n; }
        } else {
          int pl = (gr.dirl > 0) ? e_l - gr.bot : gr.bot - e_l;
          if (pl < 0) pl += gr.n;
          const int bnd = (pl < gr.ptop) ? 0 : 1;
          if (contributes(pos_of(e_r * 2 + bnd))) { em |= (unsigned long long)(0x80 | bnd << 4) << (8 * em_n); ++em_n; }
        }
      }
    }
    const unsigned bad = T.ballot(why != 0, TEAM_SITE);
    if (bad) { const int w = T.shfl(why, __builtin_ctz(bad), TEAM_SITE); SZTM_BAIL(w); }
  }
  SZTM_PHASE(17);
  // ---- where the points go: front insertions before the start point in reverse order, back insertions after it ----
  int NF, NB;
  {
    // packed per lane: front emissions, back emissions, starts, closes in a byte each
    int cnts = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if (i >= em_n) continue;
      const int rec = (int)(em >> (8 * i)) & 0xff, sd = (rec >> 4) & 3;
      cnts += (sd == 0) ? 1 : (sd == 1) ? 1 << 8 : 1 << 16;
      if (rec >> 6 & 1) cnts += 1 << 24;
    }
    int tot;
    const int ex = T.scan_excl(cnts, &tot, TEAM_SITE);
    NF = tot & 0xff; NB = (tot >> 8) & 0xff;
    const int nstart = (tot >> 16) & 0xff, nclose = (tot >> 24) & 0xff;
    if (NF + NB + nstart == 0) return szv::ST_OK;              // nothing contributes: no overlap
    if (nstart != 1) SZTM_BAIL(12);                            // no start point, or a second output ring
    if (nclose != 1) SZTM_BAIL(21);
    const int cnt = NF + 1 + NB;
    if (cnt > (int)T_CAPV) SZTM_BAIL(20);
    // the ring is open from its start point to the point that closes it: nothing is emitted outside (AddOutPt on a
    // closed ring would be an error of the general code, clipper.cpp:1890).  In sweep order: the start point comes
    // before every emission (none before it in the prefix), the closing point after every other one.
    const int f0 = ex & 0xff, b0 = (ex >> 8) & 0xff, starts_before = (ex >> 16) & 0xff, closes_before = (ex >> 24) & 0xff;
    bool outside = false;
    int nf = 0, nb2 = 0, seen_start = starts_before, seen_close = closes_before;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if (i >= em_n) continue;
      const int rec = (int)(em >> (8 * i)) & 0xff, sd = (rec >> 4) & 3;
      int at;
      if (sd == 2) { at = NF; if (f0 + nf + b0 + nb2 != 0 || seen_close) outside = true; seen_start = 1; }
      else {
        if (!seen_start || seen_close) outside = true;
        if (sd == 0) { at = NF - 1 - (f0 + nf); ++nf; } else { at = NF + 1 + b0 + nb2; ++nb2; }
      }
      if (rec >> 6 & 1) seen_close = 1;
      Pt pt;
      if (rec & 0x80) { pt.x = S->fx[e_r][e_l]; pt.y = S->fy[e_r][e_l]; }
      else pt = S->xpt[rec & (T_CAPX - 1)];
      if (at >= 0 && at < (int)T_CAPV) { ox[at] = pt.x; oy[at] = pt.y; }
    }
    if (T.any(outside, TEAM_SITE)) SZTM_BAIL(57);
  }
  T.sync(TEAM_SITE);
  SZTM_PHASE(18);
  // ---- the finished ring: FixupOutPolygon would drop duplicate / collinear points (left to the general code), the
  // orientation from Area(outRec) in the order of the list (clipper.cpp:405-416), BuildResult from pts.prev backwards ----
  const int cnt = NF + 1 + NB;
  if (cnt < 3) SZTM_BAIL(22);
  double *terms = (double *)&S->fx[0][0];    // (the rings' vertices are in the output ring by now)
  static_assert(sizeof(((ClipScratch *)0)->fx) >= 8 * T_CAPV, "the area terms borrow the rings' room");
  {
    bool bad = false;
    for (int i = T.lane; i < cnt; i += G) {
      const int ip = (i == 0) ? cnt - 1 : i - 1, in = (i + 1 == cnt) ? 0 : i + 1;
      Pt pv, cu, nx;
      pv.x = ox[ip]; pv.y = oy[ip]; cu.x = ox[i]; cu.y = oy[i]; nx.x = ox[in]; nx.y = oy[in];
      if (same(cu, nx) || collinear3_filtered(pv, cu, nx)) bad = true;
      terms[i] = (double)(pv.x + cu.x) * (double)(pv.y - cu.y);
    }
    if (T.any(bad, TEAM_SITE)) SZTM_BAIL(23);
  }
  T.sync(TEAM_SITE);
  double a = 0;
  for (int i = 0; i < cnt; ++i) a += terms[i];
  const bool reverse = !(a * 0.5 > 0);
  {
    enum { Q = (T_CAPV + G - 1) / G };
    Pt mv[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int i = T.lane + q * G;
      mv[q].x = 0; mv[q].y = 0;
      if (i < cnt) { const int at = reverse ? (i + 1 == cnt ? 0 : i + 1) : cnt - 1 - i; mv[q].x = ox[at]; mv[q].y = oy[at]; }
    }
    T.sync(TEAM_SITE);
#pragma unroll
    for (int q = 0; q < Q; ++q) { const int i = T.lane + q * G; if (i < cnt) { ox[i] = mv[q].x; oy[i] = mv[q].y; } }
  }
  T.sync(TEAM_SITE);
  SZTM_PHASE(19);
  *out_n = cnt;
  return szv::ST_OK;
}

} // namespace sztm
