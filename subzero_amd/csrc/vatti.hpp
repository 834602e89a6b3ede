// vatti.hpp -- polygon boolean operations by a Vatti scanbeam sweep, for one thread.
//
// This is the narrow-phase primitive of the floe-interaction path: it produces, for a subject ring
// and a clip ring on int64 coordinates (metres * 2^32, reference polyclip.m:63-66), exactly the rings
// the reference's native component returns (Clipper 6.4.2 behind private/mexclipper.cpp:291-302):
// same rings, same order of rings, same first vertex and vertex order, same rounded intersection
// coordinates.  It is written from scratch for a GPU lane: no pointers, no heap, no exceptions, no
// recursion -- every list of the sweep (edge ring, active-edge list, sorted-edge list, output rings,
// joins, intersections) is an index-linked array carved out of one caller-provided workspace, and
// every failure is a status code.
//
// What must agree with the reference bit for bit, and where the reference defines it:
//   rounding of doubles to int64               clipper.cpp:136-140   -> rnd()
//   edge slope dx/dy in double, horizontals    clipper.cpp:591-596   -> set_slope()
//   x of an edge at a scanline                 clipper.cpp:615-619   -> x_at()
//   intersection point of two edges            clipper.cpp:622-689   -> cross_point()
//   exact slope comparison (128-bit products)  clipper.cpp:541-575   -> mul_eq()
//   ring area in double (orientation fix-up)   clipper.cpp:405-416   -> ring_area()
//   order of equal keys after std::sort        libstdc++ introsort   -> gnu_sort()
//   vertex clean-up of input rings             clipper.cpp:1045-1114 -> add_ring()
//   bounds / local minima                      clipper.cpp:883-1042,1179-1219
//   the sweep proper                           clipper.cpp:1560-3113
//   joins of touching output rings             clipper.cpp:3371-3786
//   output clean-up and emission               clipper.cpp:3143-3219
// Closed rings only; the options the mex never sets (reverse solution, strictly simple, preserve
// collinear, PolyTree output, open paths) are not implemented.
//
// Build note: must be compiled with -ffp-contract=off (the reference binary has no FMA).
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define SZ_HD __host__ __device__ inline
#else
#define SZ_HD inline
#endif

#ifndef SZV_PHASE
#define SZV_PHASE(k) ((void)0)   // a diagnostic build attributes time to the parts of the sweep here
#endif
namespace szv {

typedef long long i64;
typedef unsigned long long u64;

struct Pt {
  i64 x, y;
};
SZ_HD bool same(const Pt &a, const Pt &b) { return a.x == b.x && a.y == b.y; }

enum { NIL = -1 };
// operation codes use the mex's method numbering (mexclipper.cpp:215-230)
enum Op { OP_DIFFERENCE = 0, OP_INTERSECTION = 1, OP_XOR = 2, OP_UNION = 3 };
enum Fill { FILL_EVENODD = 0, FILL_NONZERO = 1, FILL_POSITIVE = 2, FILL_NEGATIVE = 3 };
enum Status {
  ST_OK = 0,
  ST_CLIP_ERROR = 1,   // the reference would raise "Clipper Error." (mexclipper.cpp:304)
  ST_BAD_ARG = 2,
  ST_OUT_CAPACITY = 3, // caller's output buffers too small
  ST_RANGE = 4,        // coordinate outside +-0x3FFFFFFFFFFFFFFF (clipper.cpp:883-896)
  ST_WORKSPACE = 5     // an internal list outgrew the workspace (never silent)
};

#define SZV_HORIZONTAL (-1.0E+40)
enum { SIDE_NONE = 0, SIDE_LEFT = 1, SIDE_RIGHT = 2 };
enum { KIND_SUBJECT = 0, KIND_CLIP = 1 };

// The records of the sweep's lists, parameterised by the integer type of list links (Ix) and of winding
// counts (Wn): `Wide` (int) for rings of any size in an HBM workspace, `Compact` (int8) for the small
// rings of Voronoi-like floes, so that a lane's whole workspace is ~2.3 KB and fits in LDS
// (64 lanes x 2.3 KB < 160 KB per CU).  The algorithm is written once against the field names.
template <class Ix, class Wn>
struct Records {
  static const bool kWave = false;   // (see WideLdsWave)
  static const bool kMirror = false; // (see WideLdsWaveMirror)
  struct Edge {
    Pt bot, cur, top;
    double slope; // dx/dy, SZV_HORIZONTAL for dy == 0
    Ix nxt, prv;  // neighbours in the input ring
    Ix up;        // next edge of the same bound (towards the local maximum)
    Ix anext, aprev; // active-edge list
    Ix snext, sprev; // sorted-edge list (also used as the pending-horizontals stack)
    Wn wind, wind2;  // winding count of own kind / of the other kind
    Ix ring;         // output ring index, or -1
    signed char kind, side, wdelta, pad;
  };
  struct LocMin {
    i64 y;
    Ix left, right;
  };
  struct OutRing {
    Ix idx;       // may be redirected to the ring this one was merged into
    Ix first_left;
    Ix pts;       // "left-most" point; pts.prev is the "right-most"
    Ix bottom;
    Ix hole;
  };
  struct OutPt {
    Pt pt;
    Ix idx, next, prev, pad;
  };
  struct Join {
    Ix a, b;
    Pt off;
  };
  struct Cross {
    Ix e1, e2;
    Pt pt;
  };
};
typedef Records<int, int> Wide;
typedef Records<signed char, signed char> Compact; // at most 127 entries per list, even-odd fill only
// Wide records again under a name of their own: the sweep instantiated on it is a separate set of functions, so a
// caller that keeps the workspace in LDS gets its own copy, specialised for ds_ addressing (contact.hpp clip_fixed_lds)
struct WideLds : Records<int, int> {};
// ... and for a sweep that ALL 64 lanes of a wave run alike (clip.hip k_clip_sweep): the one walk over the active edges
// per beam (build_crossings) then spreads over the lanes -- lane k takes the k-th edge
struct WideLdsWave : Records<int, int> { static const bool kWave = true; };
// ... with the active-edge list mirrored in the lanes (SweepT::am): the clip batch's sweep pass, whose long outlines hold
// many active edges.  The step's wave-per-pair kernel sweeps rings trimmed to a few dozen vertices with a handful of
// active edges, where the mirror's upkeep costs what the short walks did and its registers more (measured: 2500 evolved
// outlines 0.94 -> 0.99 ms/step with it, the clip batch 4.3 -> 3.8 ms)
struct WideLdsWaveMirror : Records<int, int> { static const bool kWave = true; static const bool kMirror = true; };

// ---- numeric kernels ----------------------------------------------------------------------

SZ_HD i64 rnd(double v) { return (v < 0) ? (i64)(v - 0.5) : (i64)(v + 0.5); }
SZ_HD i64 iabs(i64 v) { return v < 0 ? -v : v; }
SZ_HD double dabs(double v) { return v < 0 ? -v : v; }

// a*b == c*d exactly.  All operands are coordinate differences, |v| < 2^63, so the products need
// 127 bits; compared as sign + 128-bit magnitude.
SZ_HD void umul128(u64 a, u64 b, u64 &hi, u64 &lo) {
#if defined(__HIP_DEVICE_COMPILE__)
  lo = a * b;
  hi = __umul64hi(a, b);
#else
  unsigned __int128 p = (unsigned __int128)a * b;
  lo = (u64)p;
  hi = (u64)(p >> 64);
#endif
}
SZ_HD bool mul_eq(i64 a, i64 b, i64 c, i64 d) {
  bool neg1 = (a < 0) != (b < 0), neg2 = (c < 0) != (d < 0);
  u64 h1, l1, h2, l2;
  umul128((u64)iabs(a), (u64)iabs(b), h1, l1);
  umul128((u64)iabs(c), (u64)iabs(d), h2, l2);
  if (h1 != h2 || l1 != l2) return false;
  if ((h1 | l1) == 0) return true; // both zero
  return neg1 == neg2;
}
// collinearity of p1-p2 and p2-p3 (clipper.cpp:554-563)
SZ_HD bool collinear3(const Pt &p1, const Pt &p2, const Pt &p3) {
  return mul_eq(p1.y - p2.y, p2.x - p3.x, p1.x - p2.x, p2.y - p3.y);
}
// parallelism of p1-p2 and p3-p4 (clipper.cpp:566-575)
SZ_HD bool parallel4(const Pt &p1, const Pt &p2, const Pt &p3, const Pt &p4) {
  return mul_eq(p1.y - p2.y, p3.x - p4.x, p1.x - p2.x, p3.y - p4.y);
}

SZ_HD double slope_between(const Pt &p1, const Pt &p2) {
  return (p1.y == p2.y) ? SZV_HORIZONTAL : (double)(p2.x - p1.x) / (double)(p2.y - p1.y);
}

// libstdc++'s std::sort, restated so that elements with equal keys end up where the reference
// binary (g++/libstdc++) leaves them: introsort with median-of-three to the front, unguarded
// Hoare partition, heap sort below the depth limit, threshold 16, then one insertion pass.
// Less(a, b) must be a strict weak order.  No recursion: the right part is pushed on a small stack.
template <class T, class Less>
SZ_HD void gnu_sift(T *a, int hole, int len, T v, Less less) {
  // __adjust_heap
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (less(a[child], a[child - 1])) child--;
    a[hole] = a[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    a[hole] = a[child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && less(a[parent], v)) {
    a[hole] = a[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  a[hole] = v;
}
template <class T, class Less>
SZ_HD void gnu_heapsort(T *a, int n, Less less) {
  // partial_sort(first, last, last): make_heap + sort_heap
  if (n < 2) return;
  for (int parent = (n - 2) / 2;; --parent) {
    T v = a[parent];
    gnu_sift(a, parent, n, v, less);
    if (parent == 0) break;
  }
  for (int last = n - 1; last > 0; --last) {
    T v = a[last];
    a[last] = a[0];
    gnu_sift(a, 0, last, v, less);
  }
}
template <class T, class Less>
SZ_HD void gnu_sort(T *a, int n, Less less) {
  if (n < 2) return;
  struct Span { int lo, hi, depth; };
  Span stack[24];
  int sp = 0;
  int lg = 0;
  for (int m = n; m > 1; m >>= 1) ++lg;
  int lo = 0, hi = n, depth = 2 * lg;
  for (;;) {
    while (hi - lo > 16) {
      if (depth == 0) {
        gnu_heapsort(a + lo, hi - lo, less);
        break;
      }
      --depth;
      // median of a[lo+1], a[mid], a[hi-1] moved to a[lo]
      int mid = lo + (hi - lo) / 2;
      int ia = lo + 1, ib = mid, ic = hi - 1, pick;
      if (less(a[ia], a[ib])) {
        if (less(a[ib], a[ic])) pick = ib;
        else if (less(a[ia], a[ic])) pick = ic;
        else pick = ia;
      } else if (less(a[ia], a[ic])) pick = ia;
      else if (less(a[ib], a[ic])) pick = ic;
      else pick = ib;
      { T t = a[lo]; a[lo] = a[pick]; a[pick] = t; }
      int f = lo + 1, l = hi;
      for (;;) {
        while (less(a[f], a[lo])) ++f;
        --l;
        while (less(a[lo], a[l])) --l;
        if (!(f < l)) break;
        T t = a[f]; a[f] = a[l]; a[l] = t;
        ++f;
      }
      // recurse on [f, hi), loop on [lo, f)
      if (sp < 24) { stack[sp].lo = f; stack[sp].hi = hi; stack[sp].depth = depth; ++sp; }
      hi = f;
    }
    if (sp == 0) break;
    --sp;
    lo = stack[sp].lo; hi = stack[sp].hi; depth = stack[sp].depth;
  }
  // final insertion sort: guarded on the first 16, unguarded on the rest
  int guarded = n > 16 ? 16 : n;
  for (int i = 1; i < guarded; ++i) {
    T v = a[i];
    if (less(v, a[0])) {
      for (int k = i; k > 0; --k) a[k] = a[k - 1];
      a[0] = v;
    } else {
      int k = i;
      while (less(v, a[k - 1])) { a[k] = a[k - 1]; --k; }
      a[k] = v;
    }
  }
  for (int i = guarded; i < n; ++i) {
    T v = a[i];
    int k = i;
    while (less(v, a[k - 1])) { a[k] = a[k - 1]; --k; }
    a[k] = v;
  }
}

template <class LocMin> struct LocMinLess {
  SZ_HD bool operator()(const LocMin &a, const LocMin &b) const { return b.y < a.y; }
};
template <class Cross> struct CrossLess {
  SZ_HD bool operator()(const Cross &a, const Cross &b) const { return b.pt.y < a.pt.y; }
};

// ---- workspace ------------------------------------------------------------------------------

struct Caps {
  int edges, minima, beams, rings, pts, joins, ghosts, crosses;
};
// Capacities for a clip whose inputs total nv vertices.  Enough for simple polygons (floe outlines);
// heavily self-intersecting input can outgrow them: that is ST_WORKSPACE, and the caller retries with
// a larger nv (never a silent truncation).
SZ_HD Caps caps_for(int nv) {
  Caps c;
  c.edges = nv + 2;
  c.minima = nv / 2 + 2;
  c.beams = nv + 2;
  c.rings = 2 * nv + 4;
  c.pts = 6 * nv + 8;
  c.joins = 2 * nv + 4;
  c.ghosts = nv + 4;
  c.crosses = 4 * nv + 8;
  return c;
}
// Fixed capacities of the compact tier: two closed rings of at most 18 points in total, i.e. 16 edges (what two Voronoi-like
// floes, closed rings of 6-9 points, need).  Anything that does not fit reports ST_WORKSPACE and is re-run
// with Wide records in HBM.
enum { COMPACT_MAX_NV = 18 }; // closed rings: the closing duplicates are stripped, leaving <= 16 edges
SZ_HD Caps caps_compact() {
  Caps c;
  c.edges = 16; c.minima = 4; c.beams = 16; c.rings = 6; c.pts = 24; c.joins = 4; c.ghosts = 4; c.crosses = 8;
  return c;
}
SZ_HD size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }
template <class Rec>
SZ_HD size_t workspace_bytes_caps(const Caps &c) {
  return align16(sizeof(typename Rec::Edge) * c.edges) + align16(sizeof(typename Rec::LocMin) * c.minima) +
         align16(sizeof(i64) * c.beams) + align16(sizeof(typename Rec::OutRing) * c.rings) +
         align16(sizeof(typename Rec::OutPt) * c.pts) + align16(sizeof(typename Rec::Join) * c.joins) +
         align16(sizeof(typename Rec::Join) * c.ghosts) + align16(sizeof(typename Rec::Cross) * c.crosses);
}
SZ_HD size_t workspace_bytes(int nv) { return workspace_bytes_caps<Wide>(caps_for(nv)); }
// Capacities for nv input vertices inside a budget of `bytes` (a slice of LDS): the lists whose length the input fixes
// (edges, minima, scanlines) in full, the lists of the output side (rings, points, joins, crossings) in caps_for's
// proportions from what is left.  Simple outlines use a fraction of caps_for's worst case; a sweep that outgrows the
// slice reports ST_WORKSPACE like any other and the caller repeats it in the HBM workspace.  False: nv does not fit.
template <class Rec>
SZ_HD bool caps_budget(int nv, size_t bytes, Caps &c) {
  Caps in;
  in.edges = nv + 2; in.minima = nv / 2 + 2; in.beams = nv + 2;
  in.rings = in.pts = in.joins = in.ghosts = in.crosses = 0;
  const size_t fixed = workspace_bytes_caps<Rec>(in) + 5 * 16;
  Caps one = caps_for(1), two = caps_for(2);                  // (every list is linear in nv)
  one.edges = one.minima = one.beams = two.edges = two.minima = two.beams = 0;
  const size_t b1 = workspace_bytes_caps<Rec>(one), b2 = workspace_bytes_caps<Rec>(two);
  const size_t per = b2 > b1 ? b2 - b1 : 1;
  if (bytes < fixed + b1 + 4 * per) return false;
  size_t u = (bytes - fixed - b1) / per;
  while (u > 4) {                                             // (alignment padding: settle on a u that really fits)
    c = caps_for((int)u);
    c.edges = in.edges; c.minima = in.minima; c.beams = in.beams;
    if (workspace_bytes_caps<Rec>(c) <= bytes) return true;
    --u;
  }
  return false;
}
SZ_HD size_t workspace_bytes_compact() { return workspace_bytes_caps<Compact>(caps_compact()); }
// Linear upper bound of workspace_bytes(nv), so that per-pair workspace offsets follow in closed form
// from the CSR offsets of the inputs: offset(p) = WS_PER_VERTEX * (verts before p) + WS_PER_CLIP * p.
enum { WS_PER_VERTEX = 528, WS_PER_CLIP = 1024 };

template <class Rec>
struct SweepT {
  typedef typename Rec::Edge Edge;
  typedef typename Rec::LocMin LocMin;
  typedef typename Rec::OutRing OutRing;
  typedef typename Rec::OutPt OutPt;
  typedef typename Rec::Join Join;
  typedef typename Rec::Cross Cross;
  // storage
  Edge *E;
  LocMin *M;
  i64 *B; // pending scanline ys, ascending, unique
  OutRing *R;
  OutPt *P;
  Join *J;
  Join *G;
  Cross *X;
  Caps cap;
  int nE, nM, nB, nR, nP, nJ, nG, nX;
  // state
  int ael;    // head of the active-edge list
  int sel;    // head of the sorted-edge list / horizontals stack
  int cur_lm; // next local minimum to insert
  int op, fill_s, fill_c;
  int err;
  int reverse_out;   // Clipper::ReverseSolution (clipper.cpp:1478,1598,3721,3734); only ClipperOffset sets it
  // Curr.Y of the reference's active edges is one value at any moment: the scanline the sweep stands on (every edge
  // gets it at the top of each beam, clipper.cpp:3020-3038, and an edge that comes in starts there).  It is kept here,
  // once, instead of in every edge -- the walk that stored it per edge and beam is gone; cur.x stays per edge.
  i64 y_now;
  // build_crossings notes the edges that end at the top of the beam, in list order, for do_top_of_beam -- in the
  // memory of the crossing list, which is empty exactly when the note is valid (a beam without crossings: the active
  // list is then unchanged in between); note[0] their number, note[1..4] the first four
  bool beam_in_order;
  SZ_HD int *end_note() const { return (int *)X; }
  // WideLdsWave: every ring so far went in by add_ring_wave (AddPath's plain case: all its vertices are edges, none
  // horizontal) / the pending scanlines were laid out at once from the vertices' heights (prebuild_beams)
  bool all_plain, beams_prebuilt;
  // WideLdsWave: the active-edge list again, dense and in list order, an edge per lane -- lane k holds the k-th active
  // edge in `am` (NIL beyond the `an` edges of the list; an < 0: no mirror -- more than 64 active edges, or not a wave).
  // The linked list in the records stays what the sweep's serial parts walk; the mirror is kept in step by the four
  // functions that change the list (ael_insert / ael_remove / ael_swap / ael_advance), and what asks "the k-th edge" or
  // "where in the list" -- the lane-per-edge walk of build_crossings, the place of a new edge, the winding counts of a
  // new bound -- reads it instead of following the links one dependent load after the other.
  int am, an;

  SZ_HD void bind(void *buf, int nv) { bind_caps(buf, caps_for(nv)); }
  SZ_HD void bind_caps(void *buf, const Caps &c) {
    cap = c;
    char *p = (char *)buf;
    E = (Edge *)p; p += align16(sizeof(Edge) * cap.edges);
    M = (LocMin *)p; p += align16(sizeof(LocMin) * cap.minima);
    B = (i64 *)p; p += align16(sizeof(i64) * cap.beams);
    R = (OutRing *)p; p += align16(sizeof(OutRing) * cap.rings);
    P = (OutPt *)p; p += align16(sizeof(OutPt) * cap.pts);
    J = (Join *)p; p += align16(sizeof(Join) * cap.joins);
    G = (Join *)p; p += align16(sizeof(Join) * cap.ghosts);
    X = (Cross *)p;
    nE = nM = nB = nR = nP = nJ = nG = nX = 0;
    ael = sel = NIL;
    cur_lm = 0;
    err = ST_OK;
    reverse_out = 0;
    y_now = 0; beam_in_order = false;
    all_plain = Rec::kWave; beams_prebuilt = false;
    am = NIL; an = -1;
  }
  SZ_HD void fail(int code) { if (err == ST_OK) err = code; }

  // ---- edge helpers ----
  SZ_HD bool horiz(int e) const { return E[e].slope == SZV_HORIZONTAL; }
  SZ_HD void set_slope(Edge &e) {
    i64 dy = e.top.y - e.bot.y;
    if (dy == 0) e.slope = SZV_HORIZONTAL;
    else e.slope = (double)(e.top.x - e.bot.x) / (double)dy;
  }
  SZ_HD i64 x_at(int e, i64 y) const {
    const Edge &g = E[e];
    return (y == g.top.y) ? g.top.x : g.bot.x + rnd(g.slope * (double)(y - g.bot.y));
  }
  SZ_HD Pt cur_of(int e) const { Pt p; p.x = E[e].cur.x; p.y = y_now; return p; }   // Curr of an active edge
  SZ_HD void flip_horizontal(int e) {
    i64 t = E[e].top.x; E[e].top.x = E[e].bot.x; E[e].bot.x = t;
  }

  SZ_HD Pt cross_point(int i1, int i2) const {
    const Edge &a = E[i1], &b = E[i2];
    Pt ip;
    if (a.slope == b.slope) {
      ip.y = y_now;
      ip.x = x_at(i1, ip.y);
      return ip;
    } else if (a.slope == 0) {
      ip.x = a.bot.x;
      if (b.slope == SZV_HORIZONTAL) ip.y = b.bot.y;
      else {
        double k = (double)b.bot.y - ((double)b.bot.x / b.slope);
        ip.y = rnd((double)ip.x / b.slope + k);
      }
    } else if (b.slope == 0) {
      ip.x = b.bot.x;
      if (a.slope == SZV_HORIZONTAL) ip.y = a.bot.y;
      else {
        double k = (double)a.bot.y - ((double)a.bot.x / a.slope);
        ip.y = rnd((double)ip.x / a.slope + k);
      }
    } else {
      double ka = (double)a.bot.x - (double)a.bot.y * a.slope;
      double kb = (double)b.bot.x - (double)b.bot.y * b.slope;
      double q = (kb - ka) / (a.slope - b.slope);
      ip.y = rnd(q);
      if (dabs(a.slope) < dabs(b.slope)) ip.x = rnd(a.slope * q + ka);
      else ip.x = rnd(b.slope * q + kb);
    }
    if (ip.y < a.top.y || ip.y < b.top.y) {
      ip.y = (a.top.y > b.top.y) ? a.top.y : b.top.y;
      if (dabs(a.slope) < dabs(b.slope)) ip.x = x_at(i1, ip.y);
      else ip.x = x_at(i2, ip.y);
    }
    // never below the bottom of the scanbeam
    if (ip.y > y_now) {
      ip.y = y_now;
      if (dabs(a.slope) > dabs(b.slope)) ip.x = x_at(i2, ip.y);
      else ip.x = x_at(i1, ip.y);
    }
    return ip;
  }

  // ---- pending scanlines: ascending unique array, largest y popped first ----
  // (the place by bisection: a new edge's top lies below every vertex between it and the scanline, tens to hundreds of
  // pending heights for a long outline, and walking down to it is a chain of dependent loads -- the reference keeps a
  // binary heap and drops duplicates when it pops, clipper.cpp:1332-1349; the heights come out the same)
  SZ_HD void push_beam(i64 y) {
    if (beams_prebuilt) return;             // (every height the sweep will push is there already)
    int k = nB;
    if (k > 0 && !(B[k - 1] < y)) {
      int lo = 0;
      --k;                                  // B[k] >= y: the first such index lies in [lo, k]
      while (lo < k) {
        const int mid = (lo + k) >> 1;
        if (B[mid] < y) lo = mid + 1; else k = mid;
      }
      if (B[k] == y) return;
    }
    if (nB >= cap.beams) { fail(ST_WORKSPACE); return; }
    for (int j = nB; j > k; --j) B[j] = B[j - 1];
    B[k] = y;
    ++nB;
  }
#if defined(__HIP_DEVICE_COMPILE__)
  // The pending scanlines at once, by the wave (WideLdsWave, every ring plain): the heights the sweep pushes one by one --
  // a local minimum's when run() starts, an edge's top when the edge becomes active (clipper.cpp:1332-1349 InsertScanbeam
  // from :1229-1243, :2036-2060, :2975-2997) -- are, over a whole sweep, the heights of all the vertices: every edge of
  // a ring becomes active (its bound starts at a local minimum and each edge hands on to the next at its top), a
  // height is pushed before the sweep reaches it, and the sweep pops the largest pending one whenever it was pushed.  So
  // the array is filled here with every vertex's height at its rank, and push_beam has nothing left to do -- when no
  // two vertices share a height (then the rank is the place; otherwise the heights are pushed as before).
  __device__ void prebuild_beams() {
    const int lane = (int)(threadIdx.x & 63), n = nE;
    if (n > cap.beams || n < 1) return;
    bool dup = false;
    for (int i0 = 0; i0 < n; i0 += 64) {
      const int i = i0 + lane;
      if (i < n) {
        const i64 yi = E[i].cur.y;                        // (cur: still the vertex add_ring_wave put there)
        int less = 0;
        for (int j = 0; j < n; ++j) {
          const i64 yj = E[j].cur.y;
          less += (int)(yj < yi);
          dup |= (yj == yi) && j < i;
        }
        if (less < n) B[less] = yi;
      }
    }
    if (__ballot(dup) != 0ULL) return;
    nB = n;
    beams_prebuilt = true;
    __builtin_amdgcn_wave_barrier();
  }
#endif
  SZ_HD bool pop_beam(i64 &y) {
    if (nB == 0) return false;
    y = B[--nB];
    return true;
  }

  // ---- input: ring clean-up, bounds, local minima (clipper.cpp:1045-1221) ----
  SZ_HD int unlink_edge(int e) {
    int n = E[e].nxt, p = E[e].prv;
    E[p].nxt = n;
    E[n].prv = p;
    E[e].prv = NIL;
    return n;
  }

  SZ_HD int next_local_min(int e) {
    for (;;) {
      while (!same(E[e].bot, E[E[e].prv].bot) || same(E[e].cur, E[e].top)) e = E[e].nxt;
      if (!horiz(e) && !horiz(E[e].prv)) break;
      while (horiz(E[e].prv)) e = E[e].prv;
      int e2 = e;
      while (horiz(e)) e = E[e].nxt;
      if (E[e].top.y == E[E[e].prv].bot.y) continue; // an intermediate horizontal
      if (E[E[e2].prv].bot.x < E[e].bot.x) e = e2;
      break;
    }
    return e;
  }

  // Walk one bound from its bottom edge, chaining the edges with .up; returns the edge just past it.
  SZ_HD int chain_bound(int e, bool forward) {
    int res = e;
    if (horiz(e)) {
      // the bottom horizontal must start where the other bound's bottom edge starts
      int s = forward ? E[e].prv : E[e].nxt;
      if (horiz(s)) {
        if (E[s].bot.x != E[e].bot.x && E[s].top.x != E[e].bot.x) flip_horizontal(e);
      } else if (E[s].bot.x != E[e].bot.x) flip_horizontal(e);
    }
    int start = e;
    if (forward) {
      while (E[res].top.y == E[E[res].nxt].bot.y) res = E[res].nxt;
      if (horiz(res)) {
        // top horizontals belong to this bound only when the preceding edge joins their left end
        int h = res;
        while (horiz(E[h].prv)) h = E[h].prv;
        if (E[E[h].prv].top.x > E[E[res].nxt].top.x) res = E[h].prv;
      }
      while (e != res) {
        E[e].up = E[e].nxt;
        if (horiz(e) && e != start && E[e].bot.x != E[E[e].prv].top.x) flip_horizontal(e);
        e = E[e].nxt;
      }
      if (horiz(e) && e != start && E[e].bot.x != E[E[e].prv].top.x) flip_horizontal(e);
      res = E[res].nxt;
    } else {
      while (E[res].top.y == E[E[res].prv].bot.y) res = E[res].prv;
      if (horiz(res)) {
        int h = res;
        while (horiz(E[h].nxt)) h = E[h].nxt;
        if (E[E[h].nxt].top.x == E[E[res].prv].top.x || E[E[h].nxt].top.x > E[E[res].prv].top.x)
          res = E[h].nxt;
      }
      while (e != res) {
        E[e].up = E[e].prv;
        if (horiz(e) && e != start && E[e].bot.x != E[E[e].nxt].top.x) flip_horizontal(e);
        e = E[e].prv;
      }
      if (horiz(e) && e != start && E[e].bot.x != E[E[e].nxt].top.x) flip_horizontal(e);
      res = E[res].prv;
    }
    return res;
  }

  SZ_HD bool in_range(const Pt &p) const {
    const i64 hi = 0x3FFFFFFFFFFFFFFFLL;
    return !(p.x > hi || p.y > hi || -p.x > hi || -p.y > hi);
  }

#if defined(__HIP_DEVICE_COMPILE__)
  // add_ring by the 64 lanes of a wave that all run the sweep alike (WideLdsWave), for the ring AddPath leaves as it is:
  // no repeated vertex, no collinear triple (clipper.cpp:1100-1114 removes nothing) and no horizontal edge (none of
  // the special cases of clipper.cpp:960-1043, 1146-1221).  Then every edge's record is a function of four consecutive
  // vertices: lane i writes edge i -- orientation and slope (:1118-1131), the link to the next edge of its bound
  // (ProcessBound's walk to the local maximum without horizontals: the bound goes on while the next edge ascends the
  // same way), WindDelta of the two bottom edges of a local minimum (:1189-1198) -- and the local minima are listed
  // in ring order from the first vertex, the order the serial walk below finds them in (it starts at `first`, takes
  // a minimum, continues past the top of its forward bound: ascending vertex numbers).  Returns 1 / 0 = add_ring's
  // true / false, -1: not this case, nothing written or counted (the serial code below does the ring).
  template <class Ring>
  __device__ int add_ring_wave(const Ring &ring, int hi, int kind, int base) {
    const int lane = (int)(threadIdx.x & 63);
    bool plain = true;
    for (int i0 = 0; i0 <= hi; i0 += 64) {               // first round: is it the plain case?
      const int i = i0 + lane;
      bool bad = false;
      if (i <= hi) {
        const Pt pp = ring(i == 0 ? hi : i - 1), p = ring(i), pn = ring(i == hi ? 0 : i + 1);
        bad = !in_range(p) || p.y == pn.y || collinear3(pp, p, pn);     // (the serial loop reports a vertex out of range)
      }
      if (__ballot(bad) != 0ULL) plain = false;
    }
    if (!plain) return -1;
    int nmin = 0;                                        // local minima before this chunk of 64 vertices
    for (int i0 = 0; i0 <= hi; i0 += 64) {
      const int i = i0 + lane;
      bool is_min = false;
      LocMin lm;
      lm.y = 0; lm.left = lm.right = NIL;
      if (i <= hi) {
        const int ip = i == 0 ? hi : i - 1, in = i == hi ? 0 : i + 1, inn = in == hi ? 0 : in + 1;
        const Pt pp = ring(ip), p = ring(i), pn = ring(in), pnn = ring(inn);
        // edge k "ascends forward" when its bottom is vertex k (y grows downwards)
        const bool asc_prev = pp.y > p.y, asc = p.y > pn.y, asc_next = pn.y > pnn.y;
        Edge &e = E[base + i];
        e.cur = p;
        if (asc) { e.bot = p; e.top = pn; } else { e.top = p; e.bot = pn; }
        set_slope(e);
        e.nxt = base + in;
        e.prv = base + ip;
        e.up = asc ? (asc_next ? base + in : NIL) : (!asc_prev ? base + ip : NIL);
        e.anext = e.aprev = e.snext = e.sprev = NIL;
        e.wind = e.wind2 = 0;
        e.ring = NIL;
        e.kind = (signed char)kind;
        e.side = SIDE_NONE;
        // a local minimum at vertex k: edge k ascends forward from it, edge k - 1 backward; WindDelta +1 / -1
        is_min = asc && !asc_prev;
        e.wdelta = is_min ? (signed char)1 : ((asc_next && !asc) ? (signed char)-1 : (signed char)0);
        e.pad = 0;
        if (is_min) {
          // (the slope of edge i - 1 as its own lane computes it: bottom at vertex i, top at vertex i - 1)
          const double slope_prev = (double)(pp.x - p.x) / (double)(pp.y - p.y);
          lm.y = p.y;
          if (e.slope < slope_prev) { lm.left = base + ip; lm.right = base + i; }
          else { lm.left = base + i; lm.right = base + ip; }
        }
      }
      const unsigned long long mm = __ballot(is_min);
      const int total = nmin + __popcll(mm);
      if (nM + total > cap.minima) { fail(ST_WORKSPACE); return 0; }
      if (is_min) M[nM + nmin + __popcll(mm & ((1ULL << lane) - 1ULL))] = lm;
      nmin = total;
    }
    nE += hi + 1;
    nM += nmin;
    __builtin_amdgcn_wave_barrier();
    return 1;
  }
#endif

  // Adds one closed ring.  Returns false when the ring is degenerate (as AddPath does).
  template <class Ring>
  SZ_HD bool add_ring(const Ring &ring, int n, int kind) {
    int hi = n - 1;
    while (hi > 0 && same(ring(hi), ring(0))) --hi;
    while (hi > 0 && same(ring(hi), ring(hi - 1))) --hi;
    if (hi < 2) return false;
    if (nE + hi + 1 > cap.edges) { fail(ST_WORKSPACE); return false; }
    const int base = nE;
#if defined(__HIP_DEVICE_COMPILE__)
    if (Rec::kWave) { const int r = add_ring_wave(ring, hi, kind, base); if (r >= 0) return r != 0; }
#endif
    all_plain = false;
    for (int i = 0; i <= hi; ++i) {
      Edge &e = E[base + i];
      Pt p = ring(i);
      if (!in_range(p)) { fail(ST_RANGE); return false; }
      e.bot.x = e.bot.y = e.top.x = e.top.y = 0;
      e.cur = p;
      e.slope = 0;
      e.nxt = base + (i == hi ? 0 : i + 1);
      e.prv = base + (i == 0 ? hi : i - 1);
      e.up = e.anext = e.aprev = e.snext = e.sprev = NIL;
      e.wind = e.wind2 = 0;
      e.ring = NIL;
      e.kind = (signed char)kind;
      e.side = SIDE_NONE;
      e.wdelta = 0;
      e.pad = 0;
    }
    nE += hi + 1;

    // drop duplicate vertices and the middle vertex of collinear triples
    int first = base, e = base, stop = base;
    for (;;) {
      if (same(E[e].cur, E[E[e].nxt].cur)) {
        if (e == E[e].nxt) break;
        if (e == first) first = E[e].nxt;
        e = unlink_edge(e);
        stop = e;
        continue;
      }
      if (E[e].prv == E[e].nxt) break;
      if (collinear3(E[E[e].prv].cur, E[e].cur, E[E[e].nxt].cur)) {
        if (e == first) first = E[e].nxt;
        e = unlink_edge(e);
        e = E[e].prv;
        stop = e;
        continue;
      }
      e = E[e].nxt;
      if (e == stop) break;
    }
    if (E[e].prv == E[e].nxt) { nE = base; return false; }

    // orient every edge bottom -> top and compute its slope
    bool flat = true;
    e = first;
    do {
      Edge &g = E[e];
      const Pt &q = E[g.nxt].cur;
      if (g.cur.y >= q.y) { g.bot = g.cur; g.top = q; }
      else { g.top = g.cur; g.bot = q; }
      set_slope(g);
      e = g.nxt;
      if (flat && E[e].cur.y != E[first].cur.y) flat = false;
    } while (e != first);
    if (flat) { nE = base; return false; }

    if (same(E[E[e].prv].bot, E[E[e].prv].top)) e = E[e].nxt;
    int first_min = NIL;
    for (;;) {
      e = next_local_min(e);
      if (e == first_min) break;
      if (first_min == NIL) first_min = e;
      if (nM >= cap.minima) { fail(ST_WORKSPACE); return false; }
      LocMin lm;
      lm.y = E[e].bot.y;
      bool left_forward;
      if (E[e].slope < E[E[e].prv].slope) { lm.left = E[e].prv; lm.right = e; left_forward = false; }
      else { lm.left = e; lm.right = E[e].prv; left_forward = true; }
      E[lm.left].wdelta = (E[lm.left].nxt == lm.right) ? -1 : 1;
      E[lm.right].wdelta = (signed char)(-E[lm.left].wdelta);
      e = chain_bound(lm.left, left_forward);
      int e2 = chain_bound(lm.right, !left_forward);
      M[nM++] = lm;
      if (!left_forward) e = e2;
    }
    return true;
  }

  // ---- active-edge list ----
  SZ_HD bool goes_before(int e1, int e2) const {
    // does e2 insert before e1?
    if (E[e2].cur.x == E[e1].cur.x) {
      if (E[e2].top.y > E[e1].top.y) return E[e2].top.x < x_at(e1, E[e2].top.y);
      else return E[e1].top.x > x_at(e2, E[e1].top.y);
    }
    return E[e2].cur.x < E[e1].cur.x;
  }
#if defined(__HIP_DEVICE_COMPILE__)
  __device__ int ael_index(int e) const {                      // the place of e in the mirror, -1: not there
    const unsigned long long m = __ballot(am == e);
    return m ? (int)__builtin_ctzll(m) : -1;
  }
  // ael_insert by the wave: every lane asks goes_before of its own edge, the first one (behind `from`) that answers yes is
  // the place -- what the walk below finds (it stops in front of the first edge the new one goes before)
  __device__ bool ael_insert_wave(int e, int from) {
    if (an >= 64) { an = -1; return false; }
    int start = 0;
    if (from != NIL) { const int f = ael_index(from); if (f < 0) { an = -1; return false; } start = f + 1; }
    const int lane = (int)(threadIdx.x & 63);
    bool gb = false;
    if (am != NIL) gb = goes_before(am, e);
    unsigned long long m = __ballot(gb);
    if (start > 0) m &= ~((1ULL << start) - 1ULL);
    const int pos = m ? (int)__builtin_ctzll(m) : an;
    const int p = pos > 0 ? __shfl(am, pos - 1) : NIL;
    const int n = pos < an ? __shfl(am, pos) : NIL;
    E[e].aprev = p; E[e].anext = n;
    if (p != NIL) E[p].anext = e; else ael = e;
    if (n != NIL) E[n].aprev = e;
    const int up1 = __shfl_up(am, 1);
    if (lane > pos) am = up1; else if (lane == pos) am = e;
    ++an;
    return true;
  }
#endif
  SZ_HD void ael_insert(int e, int from) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (Rec::kMirror && an >= 0 && ael_insert_wave(e, from)) return;
#endif
    if (ael == NIL) {
      E[e].aprev = E[e].anext = NIL;
      ael = e;
    } else if (from == NIL && goes_before(ael, e)) {
      E[e].aprev = NIL;
      E[e].anext = ael;
      E[ael].aprev = e;
      ael = e;
    } else {
      if (from == NIL) from = ael;
      while (E[from].anext != NIL && !goes_before(E[from].anext, e)) from = E[from].anext;
      E[e].anext = E[from].anext;
      if (E[from].anext != NIL) E[E[from].anext].aprev = e;
      E[e].aprev = from;
      E[from].anext = e;
    }
  }
  SZ_HD void ael_remove(int e) {
    int p = E[e].aprev, n = E[e].anext;
    if (p == NIL && n == NIL && e != ael) return;
    if (p != NIL) E[p].anext = n; else ael = n;
    if (n != NIL) E[n].aprev = p;
    E[e].anext = E[e].aprev = NIL;
#if defined(__HIP_DEVICE_COMPILE__)
    if (Rec::kMirror && an >= 0) {
      const int idx = ael_index(e);
      if (idx < 0) an = -1;
      else {
        const int lane = (int)(threadIdx.x & 63);
        const int dn = __shfl_down(am, 1);
        if (lane >= idx) am = (lane >= an - 1) ? NIL : dn;
        --an;
      }
    }
#endif
  }
  SZ_HD void ael_swap(int a, int b) {
    if (E[a].anext == E[a].aprev || E[b].anext == E[b].aprev) return;
#if defined(__HIP_DEVICE_COMPILE__)
    if (Rec::kMirror && an >= 0) { if (am == a) am = b; else if (am == b) am = a; }
#endif
    if (E[a].anext == b) {
      int n = E[b].anext;
      if (n != NIL) E[n].aprev = a;
      int p = E[a].aprev;
      if (p != NIL) E[p].anext = b;
      E[b].aprev = p; E[b].anext = a; E[a].aprev = b; E[a].anext = n;
    } else if (E[b].anext == a) {
      int n = E[a].anext;
      if (n != NIL) E[n].aprev = b;
      int p = E[b].aprev;
      if (p != NIL) E[p].anext = a;
      E[a].aprev = p; E[a].anext = b; E[b].aprev = a; E[b].anext = n;
    } else {
      int n = E[a].anext, p = E[a].aprev;
      E[a].anext = E[b].anext;
      if (E[a].anext != NIL) E[E[a].anext].aprev = a;
      E[a].aprev = E[b].aprev;
      if (E[a].aprev != NIL) E[E[a].aprev].anext = a;
      E[b].anext = n;
      if (n != NIL) E[n].aprev = b;
      E[b].aprev = p;
      if (p != NIL) E[p].anext = b;
    }
    if (E[a].aprev == NIL) ael = a;
    else if (E[b].aprev == NIL) ael = b;
  }
  // replace e in the AEL by the next edge of its bound; returns that edge
  SZ_HD int ael_advance(int e) {
    int u = E[e].up;
    if (u == NIL) { fail(ST_CLIP_ERROR); return e; }
#if defined(__HIP_DEVICE_COMPILE__)
    if (Rec::kMirror && an >= 0 && am == e) am = u;
#endif
    E[u].ring = E[e].ring;
    int p = E[e].aprev, n = E[e].anext;
    if (p != NIL) E[p].anext = u; else ael = u;
    if (n != NIL) E[n].aprev = u;
    E[u].side = E[e].side;
    E[u].wdelta = E[e].wdelta;
    E[u].wind = E[e].wind;
    E[u].wind2 = E[e].wind2;
    E[u].cur = E[u].bot;
    E[u].aprev = p;
    E[u].anext = n;
    if (!horiz(u)) push_beam(E[u].top.y);
    return u;
  }

  // ---- sorted-edge list ----
  SZ_HD void sel_push(int e) {
    if (sel == NIL) {
      sel = e;
      E[e].sprev = E[e].snext = NIL;
    } else {
      E[e].snext = sel;
      E[e].sprev = NIL;
      E[sel].sprev = e;
      sel = e;
    }
  }
  SZ_HD void sel_remove(int e) {
    int p = E[e].sprev, n = E[e].snext;
    if (p == NIL && n == NIL && e != sel) return;
    if (p != NIL) E[p].snext = n; else sel = n;
    if (n != NIL) E[n].sprev = p;
    E[e].snext = E[e].sprev = NIL;
  }
  SZ_HD void sel_from_ael() {
    int e = ael;
    sel = e;
    while (e != NIL) {
      E[e].sprev = E[e].aprev;
      E[e].snext = E[e].anext;
      e = E[e].anext;
    }
  }
  SZ_HD void sel_swap(int a, int b) {
    if (E[a].snext == NIL && E[a].sprev == NIL) return;
    if (E[b].snext == NIL && E[b].sprev == NIL) return;
    if (E[a].snext == b) {
      int n = E[b].snext;
      if (n != NIL) E[n].sprev = a;
      int p = E[a].sprev;
      if (p != NIL) E[p].snext = b;
      E[b].sprev = p; E[b].snext = a; E[a].sprev = b; E[a].snext = n;
    } else if (E[b].snext == a) {
      int n = E[a].snext;
      if (n != NIL) E[n].sprev = b;
      int p = E[b].sprev;
      if (p != NIL) E[p].snext = a;
      E[a].sprev = p; E[a].snext = b; E[b].sprev = a; E[b].snext = n;
    } else {
      int n = E[a].snext, p = E[a].sprev;
      E[a].snext = E[b].snext;
      if (E[a].snext != NIL) E[E[a].snext].sprev = a;
      E[a].sprev = E[b].sprev;
      if (E[a].sprev != NIL) E[E[a].sprev].snext = a;
      E[b].snext = n;
      if (n != NIL) E[n].sprev = b;
      E[b].sprev = p;
      if (p != NIL) E[p].snext = b;
    }
    if (E[a].sprev == NIL) sel = a;
    else if (E[b].sprev == NIL) sel = b;
  }

  // ---- winding ----
  SZ_HD int fill_of(int e) const { return E[e].kind == KIND_SUBJECT ? fill_s : fill_c; }
  SZ_HD int alt_fill_of(int e) const { return E[e].kind == KIND_SUBJECT ? fill_c : fill_s; }

#if defined(__HIP_DEVICE_COMPILE__)
  // set_winding for even-odd fill on both sides, by the wave: the nearest edge of the same kind below ei in the list
  // and the number of edges between the two are a ballot each over the mirror (clipper.cpp:1660-1737: the two walks
  // below)
  __device__ bool set_winding_wave(int ei) {
    const int pos = ael_index(ei);
    if (pos < 0) { an = -1; return false; }
    Edge &g = E[ei];
    int kd = -1, wd = 0;
    if (am != NIL) { kd = E[am].kind; wd = E[am].wdelta; }
    const unsigned long long below = pos > 0 ? ((1ULL << pos) - 1ULL) : 0ULL;
    const unsigned long long m_nz = __ballot(wd != 0) & below;
    const unsigned long long m_same = __ballot(kd == (int)g.kind && wd != 0) & below;
    int start = 0;
    g.wind = g.wdelta;
    if (m_same == 0ULL) g.wind2 = 0;
    else {
      const int j = 63 - (int)__builtin_clzll(m_same);
      g.wind2 = E[__shfl(am, j)].wind2;
      start = j + 1;
    }
    const int cnt = (int)__popcll(start > 0 ? (m_nz & ~((1ULL << start) - 1ULL)) : m_nz);
    if (cnt > 0) g.wind2 = (cnt & 1) ? (g.wind2 == 0 ? 1 : 0) : (g.wind2 == 0 ? 0 : 1);
    return true;
  }
#endif
  SZ_HD void set_winding(int ei) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (Rec::kMirror && an >= 0 && fill_s == FILL_EVENODD && fill_c == FILL_EVENODD && set_winding_wave(ei)) return;
#endif
    Edge &g = E[ei];
    int e = g.aprev;
    while (e != NIL && (E[e].kind != g.kind || E[e].wdelta == 0)) e = E[e].aprev;
    if (e == NIL) {
      g.wind = g.wdelta;
      g.wind2 = 0;
      e = ael;
    } else if (fill_of(ei) == FILL_EVENODD) {
      g.wind = g.wdelta;
      g.wind2 = E[e].wind2;
      e = E[e].anext;
    } else {
      const Edge &q = E[e];
      if (q.wind * q.wdelta < 0) {
        // the previous edge of this kind is closing its polygon
        if (q.wind > 1 || q.wind < -1) {
          if (q.wdelta * g.wdelta < 0) g.wind = q.wind;
          else g.wind = q.wind + g.wdelta;
        } else g.wind = g.wdelta;
      } else {
        if (q.wdelta * g.wdelta < 0) g.wind = q.wind;
        else g.wind = q.wind + g.wdelta;
      }
      g.wind2 = q.wind2;
      e = q.anext;
    }
    if (alt_fill_of(ei) == FILL_EVENODD) {
      while (e != ei) {
        if (E[e].wdelta != 0) g.wind2 = (g.wind2 == 0 ? 1 : 0);
        e = E[e].anext;
      }
    } else {
      while (e != ei) {
        g.wind2 += E[e].wdelta;
        e = E[e].anext;
      }
    }
  }

  SZ_HD bool contributing(int ei) const {
    const Edge &g = E[ei];
    int f = fill_of(ei), f2 = alt_fill_of(ei);
    switch (f) {
    case FILL_EVENODD: break;
    case FILL_NONZERO: if (g.wind != 1 && g.wind != -1) return false; break;
    case FILL_POSITIVE: if (g.wind != 1) return false; break;
    default: if (g.wind != -1) return false;
    }
    // inside-ness of the other polygon set at this edge
    bool in2, out2;
    switch (f2) {
    case FILL_EVENODD:
    case FILL_NONZERO: in2 = (g.wind2 != 0); out2 = (g.wind2 == 0); break;
    case FILL_POSITIVE: in2 = (g.wind2 > 0); out2 = (g.wind2 <= 0); break;
    default: in2 = (g.wind2 < 0); out2 = (g.wind2 >= 0);
    }
    switch (op) {
    case OP_INTERSECTION: return in2;
    case OP_UNION: return out2;
    case OP_DIFFERENCE: return (g.kind == KIND_SUBJECT) ? out2 : in2;
    default: return true; // xor
    }
  }

  // ---- output rings ----
  SZ_HD int new_ring() {
    if (nR >= cap.rings) { fail(ST_WORKSPACE); return 0; }
    OutRing &r = R[nR];
    r.hole = 0;
    r.first_left = NIL;
    r.pts = NIL;
    r.bottom = NIL;
    r.idx = nR;
    return nR++;
  }
  SZ_HD int new_pt() {
    if (nP >= cap.pts) { fail(ST_WORKSPACE); return 0; }
    return nP++;
  }

  SZ_HD void set_hole_state(int e, int ring) {
    int e2 = E[e].aprev, tmp = NIL;
    while (e2 != NIL) {
      if (E[e2].ring >= 0 && E[e2].wdelta != 0) {
        if (tmp == NIL) tmp = e2;
        else if (E[tmp].ring == E[e2].ring) tmp = NIL;
      }
      e2 = E[e2].aprev;
    }
    if (tmp == NIL) {
      R[ring].first_left = NIL;
      R[ring].hole = 0;
    } else {
      R[ring].first_left = E[tmp].ring;
      R[ring].hole = !R[E[tmp].ring].hole;
    }
  }

  SZ_HD int emit(int e, const Pt &pt) {
    if (E[e].ring < 0) {
      int r = new_ring();
      int p = new_pt();
      if (err) return 0;
      R[r].pts = p;
      P[p].idx = R[r].idx;
      P[p].pt = pt;
      P[p].next = P[p].prev = p;
      set_hole_state(e, r);
      E[e].ring = R[r].idx;
      return p;
    }
    OutRing &r = R[E[e].ring];
    int head = r.pts;
    bool front = (E[e].side == SIDE_LEFT);
    if (front && same(pt, P[head].pt)) return head;
    if (!front && same(pt, P[P[head].prev].pt)) return P[head].prev;
    int p = new_pt();
    if (err) return 0;
    P[p].idx = r.idx;
    P[p].pt = pt;
    P[p].next = head;
    P[p].prev = P[head].prev;
    P[P[p].prev].next = p;
    P[head].prev = p;
    if (front) r.pts = p;
    return p;
  }
  SZ_HD int last_pt(int e) const {
    const OutRing &r = R[E[e].ring];
    return (E[e].side == SIDE_LEFT) ? r.pts : P[r.pts].prev;
  }

  SZ_HD void add_join(int a, int b, const Pt &off) {
    if (nJ >= cap.joins) { fail(ST_WORKSPACE); return; }
    J[nJ].a = a; J[nJ].b = b; J[nJ].off = off;
    ++nJ;
  }
  SZ_HD void add_ghost(int a, const Pt &off) {
    if (nG >= cap.ghosts) { fail(ST_WORKSPACE); return; }
    G[nG].a = a; G[nG].b = NIL; G[nG].off = off;
    ++nG;
  }

  SZ_HD double ring_area(int start) const {
    if (start == NIL) return 0;
    double a = 0;
    int p = start;
    do {
      const OutPt &q = P[p];
      a += (double)(P[q.prev].pt.x + q.pt.x) * (double)(P[q.prev].pt.y - q.pt.y);
      p = q.next;
    } while (p != start);
    return a * 0.5;
  }
  SZ_HD void reverse_ring(int start) {
    if (start == NIL) return;
    int p = start;
    do {
      int n = P[p].next;
      P[p].next = P[p].prev;
      P[p].prev = n;
      p = n;
    } while (p != start);
  }

  SZ_HD int local_min_poly(int e1, int e2, const Pt &pt) {
    int res, e, prev_e;
    if (horiz(e2) || E[e1].slope > E[e2].slope) {
      res = emit(e1, pt);
      E[e2].ring = E[e1].ring;
      E[e1].side = SIDE_LEFT;
      E[e2].side = SIDE_RIGHT;
      e = e1;
      prev_e = (E[e].aprev == e2) ? E[e2].aprev : E[e].aprev;
    } else {
      res = emit(e2, pt);
      E[e1].ring = E[e2].ring;
      E[e1].side = SIDE_RIGHT;
      E[e2].side = SIDE_LEFT;
      e = e2;
      prev_e = (E[e].aprev == e1) ? E[e1].aprev : E[e].aprev;
    }
    if (prev_e != NIL && E[prev_e].ring >= 0 && E[prev_e].top.y < pt.y && E[e].top.y < pt.y) {
      i64 xp = x_at(prev_e, pt.y);
      i64 xe = x_at(e, pt.y);
      if (xp == xe && E[e].wdelta != 0 && E[prev_e].wdelta != 0) {
        Pt a; a.x = xp; a.y = pt.y;
        Pt b; b.x = xe; b.y = pt.y;
        if (parallel4(a, E[prev_e].top, b, E[e].top)) {
          int o = emit(prev_e, pt);
          add_join(res, o, E[e].top);
        }
      }
    }
    return res;
  }

  // bottom-most point of a ring, resolving duplicates by the flatter adjoining edge
  SZ_HD bool first_is_bottom(int b1, int b2) const {
    int p = P[b1].prev;
    while (same(P[p].pt, P[b1].pt) && p != b1) p = P[p].prev;
    double dx1p = dabs(slope_between(P[b1].pt, P[p].pt));
    p = P[b1].next;
    while (same(P[p].pt, P[b1].pt) && p != b1) p = P[p].next;
    double dx1n = dabs(slope_between(P[b1].pt, P[p].pt));
    p = P[b2].prev;
    while (same(P[p].pt, P[b2].pt) && p != b2) p = P[p].prev;
    double dx2p = dabs(slope_between(P[b2].pt, P[p].pt));
    p = P[b2].next;
    while (same(P[p].pt, P[b2].pt) && p != b2) p = P[p].next;
    double dx2n = dabs(slope_between(P[b2].pt, P[p].pt));
    double max1 = dx1p < dx1n ? dx1n : dx1p, min1 = dx1n < dx1p ? dx1n : dx1p;
    double max2 = dx2p < dx2n ? dx2n : dx2p, min2 = dx2n < dx2p ? dx2n : dx2p;
    if (max1 == max2 && min1 == min2) return ring_area(b1) > 0;
    return (dx1p >= dx2p && dx1p >= dx2n) || (dx1n >= dx2p && dx1n >= dx2n);
  }
  SZ_HD int bottom_pt(int pp) const {
    int dups = NIL;
    int p = P[pp].next;
    while (p != pp) {
      if (P[p].pt.y > P[pp].pt.y) {
        pp = p;
        dups = NIL;
      } else if (P[p].pt.y == P[pp].pt.y && P[p].pt.x <= P[pp].pt.x) {
        if (P[p].pt.x < P[pp].pt.x) {
          dups = NIL;
          pp = p;
        } else if (P[p].next != pp && P[p].prev != pp) dups = p;
      }
      p = P[p].next;
    }
    if (dups != NIL) {
      while (dups != p) {
        if (!first_is_bottom(p, dups)) pp = dups;
        dups = P[dups].next;
        while (!same(P[dups].pt, P[pp].pt)) dups = P[dups].next;
      }
    }
    return pp;
  }
  SZ_HD int lowermost_ring(int r1, int r2) {
    if (R[r1].bottom == NIL) R[r1].bottom = bottom_pt(R[r1].pts);
    if (R[r2].bottom == NIL) R[r2].bottom = bottom_pt(R[r2].pts);
    int p1 = R[r1].bottom, p2 = R[r2].bottom;
    if (P[p1].pt.y > P[p2].pt.y) return r1;
    if (P[p1].pt.y < P[p2].pt.y) return r2;
    if (P[p1].pt.x < P[p2].pt.x) return r1;
    if (P[p1].pt.x > P[p2].pt.x) return r2;
    if (P[p1].next == p1) return r2;
    if (P[p2].next == p2) return r1;
    return first_is_bottom(p1, p2) ? r1 : r2;
  }
  SZ_HD bool right_of(int r1, int r2) const {
    do {
      r1 = R[r1].first_left;
      if (r1 == r2) return true;
    } while (r1 != NIL);
    return false;
  }
  SZ_HD int resolve_ring(int idx) const {
    int r = idx;
    while (r != R[r].idx) r = R[r].idx;
    return r;
  }

  SZ_HD void append_ring(int e1, int e2) {
    int r1 = E[e1].ring, r2 = E[e2].ring;
    int keep_state;
    if (right_of(r1, r2)) keep_state = r2;
    else if (right_of(r2, r1)) keep_state = r1;
    else keep_state = lowermost_ring(r1, r2);

    int p1l = R[r1].pts, p1r = P[p1l].prev;
    int p2l = R[r2].pts, p2r = P[p2l].prev;
    if (E[e1].side == SIDE_LEFT) {
      if (E[e2].side == SIDE_LEFT) {
        reverse_ring(p2l);
        P[p2l].next = p1l; P[p1l].prev = p2l;
        P[p1r].next = p2r; P[p2r].prev = p1r;
        R[r1].pts = p2r;
      } else {
        P[p2r].next = p1l; P[p1l].prev = p2r;
        P[p2l].prev = p1r; P[p1r].next = p2l;
        R[r1].pts = p2l;
      }
    } else {
      if (E[e2].side == SIDE_RIGHT) {
        reverse_ring(p2l);
        P[p1r].next = p2r; P[p2r].prev = p1r;
        P[p2l].next = p1l; P[p1l].prev = p2l;
      } else {
        P[p1r].next = p2l; P[p2l].prev = p1r;
        P[p1l].prev = p2r; P[p2r].next = p1l;
      }
    }
    R[r1].bottom = NIL;
    if (keep_state == r2) {
      if (R[r2].first_left != r1) R[r1].first_left = R[r2].first_left;
      R[r1].hole = R[r2].hole;
    }
    R[r2].pts = NIL;
    R[r2].bottom = NIL;
    R[r2].first_left = r1;

    int ok = E[e1].ring, gone = E[e2].ring;
    E[e1].ring = NIL;
    E[e2].ring = NIL;
    for (int e = ael; e != NIL; e = E[e].anext) {
      if (E[e].ring == gone) {
        E[e].ring = ok;
        E[e].side = E[e1].side;
        break;
      }
    }
    R[r2].idx = R[r1].idx;
  }

  SZ_HD void local_max_poly(int e1, int e2, const Pt &pt) {
    emit(e1, pt);
    if (err) return;
    if (E[e1].ring == E[e2].ring) {
      E[e1].ring = NIL;
      E[e2].ring = NIL;
    } else if (E[e1].ring < E[e2].ring) append_ring(e1, e2);
    else append_ring(e2, e1);
  }

  SZ_HD void swap_sides(int a, int b) { signed char s = E[a].side; E[a].side = E[b].side; E[b].side = s; }
  SZ_HD void swap_rings(int a, int b) { int r = E[a].ring; E[a].ring = E[b].ring; E[b].ring = r; }

  // winding count of an edge as its own fill rule sees it
  SZ_HD i64 seen_wind(int fill, int w) const {
    switch (fill) {
    case FILL_POSITIVE: return w;
    case FILL_NEGATIVE: return -w;
    default: return w < 0 ? -w : w;
    }
  }

  // e1 is to the right of e2 above the crossing
  SZ_HD void cross_edges(int e1, int e2, const Pt &pt) {
    bool c1 = E[e1].ring >= 0, c2 = E[e2].ring >= 0;
    Edge &a = E[e1], &b = E[e2];
    if (a.kind == b.kind) {
      if (fill_of(e1) == FILL_EVENODD) {
        int t = a.wind; a.wind = b.wind; b.wind = t;
      } else {
        if (a.wind + b.wdelta == 0) a.wind = -a.wind; else a.wind += b.wdelta;
        if (b.wind - a.wdelta == 0) b.wind = -b.wind; else b.wind -= a.wdelta;
      }
    } else {
      if (fill_of(e2) != FILL_EVENODD) a.wind2 += b.wdelta; else a.wind2 = (a.wind2 == 0) ? 1 : 0;
      if (fill_of(e1) != FILL_EVENODD) b.wind2 -= a.wdelta; else b.wind2 = (b.wind2 == 0) ? 1 : 0;
    }
    i64 w1 = seen_wind(fill_of(e1), a.wind), w2 = seen_wind(fill_of(e2), b.wind);
    bool u1 = (w1 == 0 || w1 == 1), u2 = (w2 == 0 || w2 == 1);

    if (c1 && c2) {
      if (!u1 || !u2 || (a.kind != b.kind && op != OP_XOR)) local_max_poly(e1, e2, pt);
      else {
        emit(e1, pt);
        emit(e2, pt);
        swap_sides(e1, e2);
        swap_rings(e1, e2);
      }
    } else if (c1) {
      if (u2) { emit(e1, pt); swap_sides(e1, e2); swap_rings(e1, e2); }
    } else if (c2) {
      if (u1) { emit(e2, pt); swap_sides(e1, e2); swap_rings(e1, e2); }
    } else if (u1 && u2) {
      i64 v1 = seen_wind(alt_fill_of(e1), a.wind2), v2 = seen_wind(alt_fill_of(e2), b.wind2);
      if (a.kind != b.kind) local_min_poly(e1, e2, pt);
      else if (w1 == 1 && w2 == 1) {
        switch (op) {
        case OP_INTERSECTION: if (v1 > 0 && v2 > 0) local_min_poly(e1, e2, pt); break;
        case OP_UNION: if (v1 <= 0 && v2 <= 0) local_min_poly(e1, e2, pt); break;
        case OP_DIFFERENCE:
          if ((a.kind == KIND_CLIP && v1 > 0 && v2 > 0) || (a.kind == KIND_SUBJECT && v1 <= 0 && v2 <= 0))
            local_min_poly(e1, e2, pt);
          break;
        default: local_min_poly(e1, e2, pt);
        }
      } else swap_sides(e1, e2);
    }
  }

  // ---- sweep steps ----
  SZ_HD static bool spans_overlap(i64 a1, i64 a2, i64 b1, i64 b2) {
    if (a1 > a2) { i64 t = a1; a1 = a2; a2 = t; }
    if (b1 > b2) { i64 t = b1; b1 = b2; b2 = t; }
    return (a1 < b2) && (b1 < a2);
  }

  SZ_HD void insert_minima(i64 y) {
    while (cur_lm < nM && M[cur_lm].y == y) {
      if (err) return;
      int lb = M[cur_lm].left, rb = M[cur_lm].right;
      ++cur_lm;
      int op1 = NIL;
      ael_insert(lb, NIL);
      ael_insert(rb, lb);
      set_winding(lb);
      E[rb].wind = E[lb].wind;
      E[rb].wind2 = E[lb].wind2;
      if (contributing(lb)) op1 = local_min_poly(lb, rb, E[lb].bot);
      push_beam(E[lb].top.y);
      if (horiz(rb)) {
        sel_push(rb);
        if (E[rb].up != NIL) push_beam(E[E[rb].up].top.y);
      } else push_beam(E[rb].top.y);

      // a horizontal right bound lying on an earlier horizontal of this scanline
      if (op1 != NIL && horiz(rb) && nG > 0 && E[rb].wdelta != 0) {
        for (int i = 0; i < nG; ++i)
          if (spans_overlap(P[G[i].a].pt.x, G[i].off.x, E[rb].bot.x, E[rb].top.x))
            add_join(G[i].a, op1, G[i].off);
      }
      int lp = E[lb].aprev;
      if (E[lb].ring >= 0 && lp != NIL && E[lp].cur.x == E[lb].bot.x && E[lp].ring >= 0 &&
          parallel4(E[lp].bot, E[lp].top, cur_of(lb), E[lb].top) && E[lb].wdelta != 0 && E[lp].wdelta != 0) {
        int op2 = emit(lp, E[lb].bot);
        add_join(op1, op2, E[lb].top);
      }
      if (E[lb].anext != rb) {
        int rp = E[rb].aprev;
        if (E[rb].ring >= 0 && E[rp].ring >= 0 &&
            parallel4(cur_of(rp), E[rp].top, cur_of(rb), E[rb].top) && E[rb].wdelta != 0 && E[rp].wdelta != 0) {
          int op2 = emit(rp, E[rb].bot);
          add_join(op1, op2, E[rb].top);
        }
        int e = E[lb].anext;
        if (e != NIL) {
          while (e != rb) {
            if (err) return;
            cross_edges(rb, e, cur_of(lb));
            e = E[e].anext;
          }
        }
      }
    }
  }

  SZ_HD int maxima_pair(int e) const {
    int n = E[e].nxt, p = E[e].prv;
    if (same(E[n].top, E[e].top) && E[n].up == NIL) return n;
    if (same(E[p].top, E[e].top) && E[p].up == NIL) return p;
    return NIL;
  }
  SZ_HD int maxima_pair_in_ael(int e) const {
    int r = maxima_pair(e);
    if (r != NIL && (E[r].anext == E[r].aprev && !horiz(r))) return NIL;
    return r;
  }

  SZ_HD void join_with_pending_horizontals(int h, int op1) {
    for (int o = sel; o != NIL; o = E[o].snext) {
      if (E[o].ring >= 0 && spans_overlap(E[h].bot.x, E[h].top.x, E[o].bot.x, E[o].top.x)) {
        int op2 = last_pt(o);
        add_join(op2, op1, E[o].top);
      }
    }
  }

  SZ_HD void do_horizontal(int h) {
    bool ltr;
    i64 left, right;
    if (E[h].bot.x < E[h].top.x) { left = E[h].bot.x; right = E[h].top.x; ltr = true; }
    else { left = E[h].top.x; right = E[h].bot.x; ltr = false; }

    int last = h, max_pair = NIL;
    while (E[last].up != NIL && horiz(E[last].up)) last = E[last].up;
    if (E[last].up == NIL) max_pair = maxima_pair(last);

    int op1 = NIL;
    for (;;) {
      bool is_last = (h == last);
      int e = ltr ? E[h].anext : E[h].aprev;
      while (e != NIL) {
        if (err) return;
        if ((ltr && E[e].cur.x > right) || (!ltr && E[e].cur.x < left)) break;
        // the end of an intermediate horizontal: smaller slopes lie to the right above it
        if (E[e].cur.x == E[h].top.x && E[h].up != NIL && E[e].slope < E[E[h].up].slope) break;

        if (E[h].ring >= 0) {
          op1 = emit(h, cur_of(e));
          join_with_pending_horizontals(h, op1);
          add_ghost(op1, E[h].bot);
        }
        if (e == max_pair && is_last) {
          if (E[h].ring >= 0) local_max_poly(h, max_pair, E[h].top);
          ael_remove(h);
          ael_remove(max_pair);
          return;
        }
        Pt pt; pt.x = E[e].cur.x; pt.y = E[h].cur.y;
        if (ltr) cross_edges(h, e, pt);
        else cross_edges(e, h, pt);
        int en = ltr ? E[e].anext : E[e].aprev;
        ael_swap(h, e);
        e = en;
      }
      if (E[h].up == NIL || !horiz(E[h].up)) break;
      h = ael_advance(h);
      if (E[h].ring >= 0) emit(h, E[h].bot);
      if (E[h].bot.x < E[h].top.x) { left = E[h].bot.x; right = E[h].top.x; ltr = true; }
      else { left = E[h].top.x; right = E[h].bot.x; ltr = false; }
    }

    if (E[h].ring >= 0 && op1 == NIL) {
      op1 = last_pt(h);
      join_with_pending_horizontals(h, op1);
      add_ghost(op1, E[h].top);
    }

    if (E[h].up != NIL) {
      if (E[h].ring >= 0) {
        op1 = emit(h, E[h].top);
        h = ael_advance(h);
        // h is no longer horizontal
        int ep = E[h].aprev, en = E[h].anext;
        if (ep != NIL && E[ep].cur.x == E[h].bot.x && y_now == E[h].bot.y && E[ep].wdelta != 0 &&
            E[ep].ring >= 0 && y_now > E[ep].top.y &&
            mul_eq(E[h].top.y - E[h].bot.y, E[ep].top.x - E[ep].bot.x, E[h].top.x - E[h].bot.x, E[ep].top.y - E[ep].bot.y)) {
          int op2 = emit(ep, E[h].bot);
          add_join(op1, op2, E[h].top);
        } else if (en != NIL && E[en].cur.x == E[h].bot.x && y_now == E[h].bot.y && E[en].wdelta != 0 &&
                   E[en].ring >= 0 && y_now > E[en].top.y &&
                   mul_eq(E[h].top.y - E[h].bot.y, E[en].top.x - E[en].bot.x, E[h].top.x - E[h].bot.x, E[en].top.y - E[en].bot.y)) {
          int op2 = emit(en, E[h].bot);
          add_join(op1, op2, E[h].top);
        }
      } else ael_advance(h);
    } else {
      if (E[h].ring >= 0) emit(h, E[h].top);
      ael_remove(h);
    }
  }

  SZ_HD void do_horizontals() {
    while (sel != NIL) {
      if (err) return;
      int h = sel;
      sel_remove(h);
      do_horizontal(h);
    }
  }

  // all crossings inside the scanbeam ending at top_y, found by bubble-sorting the AEL by x at top_y
  SZ_HD void build_crossings(i64 top_y) {
    if (ael == NIL) return;
    int e = ael;
    sel = e;
    bool walked = false;
#if defined(__HIP_DEVICE_COMPILE__)
    if (Rec::kWave) {
      // lane k <- the k-th active edge (a chain of one load per edge), then every lane its edge: the links of the sorted
      // list, x at the top of the beam, whether it ends there; neighbours compare by shuffle
      const int lane = (int)(threadIdx.x & 63);
      int mine = NIL, n = 0;
      if (Rec::kMirror && an >= 0) { mine = am; n = an; }
      else for (int q = ael; q != NIL && n <= 64; q = E[q].anext, ++n) if (n == lane) mine = q;
      if (n <= 64) {
        i64 x = 0;
        bool ends = false;
        if (mine != NIL) {
          E[mine].sprev = E[mine].aprev;
          E[mine].snext = E[mine].anext;
          x = x_at(mine, top_y);
          E[mine].cur.x = x;
          ends = E[mine].top.y == top_y;
        }
        const i64 xb = __shfl_up(x, 1);
        const bool out_of_order = lane > 0 && lane < n && xb > x;
        const unsigned long long m_end = __ballot(ends);
        int *note = end_note();
        const int rank = __popcll(m_end & ((1ULL << lane) - 1ULL));
        if (ends && rank < 4) note[1 + rank] = mine;
        if (__ballot(out_of_order) == 0ULL) {
          if (n >= 2 && lane == n - 2) E[mine].snext = NIL;       // (what the bubble pass leaves behind, see below)
          if (lane == 0) note[0] = __popcll(m_end);
          __builtin_amdgcn_wave_barrier();
          sel = NIL;
          beam_in_order = true;
          return;
        }
        __builtin_amdgcn_wave_barrier();
        walked = true;                                              // links and cur.x are set: the bubble passes as they are
      }
    }
#endif
    // (most beams hold no crossing: the edges come out in order at top_y, which this first walk already sees -- then
    // the bubble pass below would compare every neighbour pair once more, find nothing, cut the last link and stop)
    bool in_order = true;
    int before = NIL;
    i64 x_before = 0;
    int end_n = 0;
    int *note = end_note();
    while (e != NIL && !walked) {
      E[e].sprev = E[e].aprev;
      const int en = E[e].anext;
      E[e].snext = en;
      const i64 x = x_at(e, top_y);
      E[e].cur.x = x;
      if (E[e].top.y == top_y) { if (end_n < 4) note[1 + end_n] = e; ++end_n; }   // (for do_top_of_beam, see there)
      if (before != NIL && x_before > x) in_order = false;
      if (en == NIL && in_order) {
        if (before != NIL) E[before].snext = NIL;
        sel = NIL;
        note[0] = end_n;
        beam_in_order = true;
        return;
      }
      before = e; x_before = x;
      e = en;
    }
    bool modified;
    do {
      modified = false;
      e = sel;
      while (E[e].snext != NIL) {
        int en = E[e].snext;
        if (E[e].cur.x > E[en].cur.x) {
          Pt pt = cross_point(e, en);
          if (pt.y < top_y) { pt.x = x_at(e, top_y); pt.y = top_y; }
          if (nX >= cap.crosses) { fail(ST_WORKSPACE); return; }
          X[nX].e1 = e; X[nX].e2 = en; X[nX].pt = pt;
          ++nX;
          sel_swap(e, en);
          modified = true;
        } else e = en;
      }
      if (E[e].sprev != NIL) E[E[e].sprev].snext = NIL;
      else break;
    } while (modified);
    sel = NIL;
  }
  SZ_HD bool adjacent_in_sel(const Cross &c) const {
    return E[c.e1].snext == c.e2 || E[c.e1].sprev == c.e2;
  }
  SZ_HD bool order_crossings() {
    // crossings must be applied between edges that are adjacent at that moment
    sel_from_ael();
    gnu_sort(X, nX, CrossLess<Cross>());
    for (int i = 0; i < nX; ++i) {
      if (!adjacent_in_sel(X[i])) {
        int j = i + 1;
        while (j < nX && !adjacent_in_sel(X[j])) ++j;
        if (j == nX) return false;
        Cross t = X[i]; X[i] = X[j]; X[j] = t;
      }
      sel_swap(X[i].e1, X[i].e2);
    }
    return true;
  }
  SZ_HD bool do_crossings(i64 top_y) {
    beam_in_order = false;
    if (ael == NIL) return true;
    build_crossings(top_y);
    if (err) return false;
    if (nX == 0) return true;
    SZV_PHASE(23);
    if (nX == 1 || order_crossings()) {
      for (int i = 0; i < nX; ++i) {
        if (err) break;
        cross_edges(X[i].e1, X[i].e2, X[i].pt);
        ael_swap(X[i].e1, X[i].e2);
      }
      nX = 0;
    } else {
      nX = 0;
      return false;
    }
    sel = NIL;
    return true;
  }

  SZ_HD void do_maxima(int e) {
    int mp = maxima_pair_in_ael(e);
    if (mp == NIL) {
      if (E[e].ring >= 0) emit(e, E[e].top);
      ael_remove(e);
      return;
    }
    int en = E[e].anext;
    while (en != NIL && en != mp) {
      if (err) return;
      cross_edges(e, en, E[e].top);
      ael_swap(e, en);
      en = E[e].anext;
    }
    if (E[e].ring == NIL && E[mp].ring == NIL) {
      ael_remove(e);
      ael_remove(mp);
    } else if (E[e].ring >= 0 && E[mp].ring >= 0) {
      local_max_poly(e, mp, E[e].top);
      ael_remove(e);
      ael_remove(mp);
    } else fail(ST_CLIP_ERROR);
  }

  // an edge that ends at the scanline with a successor: the vertex goes out, the successor takes its place (and a join is
  // recorded where it starts on a neighbour's edge that runs parallel); returns the successor
  SZ_HD int promote_vertex(int e) {
    int op = NIL;
    if (E[e].ring >= 0) op = emit(e, E[e].top);
    e = ael_advance(e);
    int ep = E[e].aprev, en = E[e].anext;
    if (ep != NIL && E[ep].cur.x == E[e].bot.x && y_now == E[e].bot.y && op != NIL &&
        E[ep].ring >= 0 && y_now > E[ep].top.y &&
        parallel4(cur_of(e), E[e].top, cur_of(ep), E[ep].top) && E[e].wdelta != 0 && E[ep].wdelta != 0) {
      int op2 = emit(ep, E[e].bot);
      add_join(op, op2, E[e].top);
    } else if (en != NIL && E[en].cur.x == E[e].bot.x && y_now == E[e].bot.y && op != NIL &&
               E[en].ring >= 0 && y_now > E[en].top.y &&
               parallel4(cur_of(e), E[e].top, cur_of(en), E[en].top) && E[e].wdelta != 0 && E[en].wdelta != 0) {
      int op2 = emit(en, E[e].bot);
      add_join(op, op2, E[e].top);
    }
    return e;
  }
  SZ_HD void do_top_of_beam(i64 top_y) {
    y_now = top_y;
    // The reference walks the whole list here (clipper.cpp:3006-3040): Curr of every edge, the maxima, the edges whose
    // successor is horizontal.  Curr needs no walk (cur.x is x_at(e, top_y) since build_crossings, the height is
    // y_now); the edges that END at top_y are the ones with anything to do, and when the beam held no crossing
    // build_crossings has met them in list order on its own walk -- they are then taken from its note, in that order
    // (one loop for both ways: the noted edges, or the whole list).
    // The second walk below looks for the ending edges with a successor; when there is a single one and no horizontal
    // -- the common scanline, one vertex of one ring -- that edge is promoted without walking again.
    int ending = NIL, n_ending = 0;
    const int *note = end_note();
    const int end_n = beam_in_order ? note[0] : 5;
    const bool noted = end_n <= 4;
    beam_in_order = false;
    int k = 0;
    int e = noted ? (end_n > 0 ? note[1] : NIL) : ael;
    while (e != NIL) {
      if (err) return;
      int next;
      // (a noted edge may have left the list with its maxima partner)
      const bool here = (!noted || !(E[e].aprev == NIL && E[e].anext == NIL && e != ael)) && E[e].top.y == top_y;
      bool is_max = here && E[e].up == NIL;
      if (is_max) {
        int mp = maxima_pair_in_ael(e);
        is_max = (mp == NIL || !horiz(mp));
      }
      if (is_max) {
        int ep = E[e].aprev;
        SZV_PHASE(20);
        do_maxima(e);
        SZV_PHASE(17);
        next = (ep == NIL) ? ael : E[ep].anext;
      } else {
        if (here && E[e].up != NIL && horiz(E[e].up)) {
          e = ael_advance(e);
          if (E[e].ring >= 0) emit(e, E[e].bot);
          sel_push(e);
        } else if (here && E[e].up != NIL) { ending = e; ++n_ending; }
        next = E[e].anext;
      }
      if (noted) { ++k; next = k < end_n ? note[1 + k] : NIL; }
      e = next;
    }
    const bool had_horizontals = sel != NIL;      // (horizontals bring edges of their own into the list)
    SZV_PHASE(21);
    do_horizontals();
    SZV_PHASE(22);
    // promote intermediate vertices
    if (!had_horizontals) {                         // (maxima remove edges and move others past them, they make none)
      if (n_ending == 0) return;
      if (n_ending == 1) { (void)promote_vertex(ending); return; }
    }
    e = ael;
    while (e != NIL) {
      if (err) return;
      if (E[e].top.y == top_y && E[e].up != NIL) e = promote_vertex(e);
      e = E[e].anext;
    }
  }

  // ---- joins (clipper.cpp:3371-3786) ----
  SZ_HD int dup_pt(int p, bool after) {
    int r = new_pt();
    if (err) return p;
    P[r].pt = P[p].pt;
    P[r].idx = P[p].idx;
    if (after) {
      P[r].next = P[p].next;
      P[r].prev = p;
      P[P[p].next].prev = r;
      P[p].next = r;
    } else {
      P[r].prev = P[p].prev;
      P[r].next = p;
      P[P[p].prev].next = r;
      P[p].prev = r;
    }
    return r;
  }

  // walk one side of a horizontal join to the splice vertex and duplicate it
  SZ_HD void horz_splice(int &op, int &opb, bool ltr, const Pt &pt, bool discard_left) {
    if (ltr) {
      while (P[P[op].next].pt.x <= pt.x && P[P[op].next].pt.x >= P[op].pt.x && P[P[op].next].pt.y == pt.y)
        op = P[op].next;
      if (discard_left && P[op].pt.x != pt.x) op = P[op].next;
      opb = dup_pt(op, !discard_left);
      if (!same(P[opb].pt, pt)) {
        op = opb;
        P[op].pt = pt;
        opb = dup_pt(op, !discard_left);
      }
    } else {
      while (P[P[op].next].pt.x >= pt.x && P[P[op].next].pt.x <= P[op].pt.x && P[P[op].next].pt.y == pt.y)
        op = P[op].next;
      if (!discard_left && P[op].pt.x != pt.x) op = P[op].next;
      opb = dup_pt(op, discard_left);
      if (!same(P[opb].pt, pt)) {
        op = opb;
        P[op].pt = pt;
        opb = dup_pt(op, discard_left);
      }
    }
  }
  SZ_HD bool join_horz(int op1, int op1b, int op2, int op2b, const Pt &pt, bool discard_left) {
    bool ltr1 = !(P[op1].pt.x > P[op1b].pt.x);
    bool ltr2 = !(P[op2].pt.x > P[op2b].pt.x);
    if (ltr1 == ltr2) return false;
    horz_splice(op1, op1b, ltr1, pt, discard_left);
    horz_splice(op2, op2b, ltr2, pt, discard_left);
    if (ltr1 == discard_left) {
      P[op1].prev = op2; P[op2].next = op1;
      P[op1b].next = op2b; P[op2b].prev = op1b;
    } else {
      P[op1].next = op2; P[op2].prev = op1;
      P[op1b].prev = op2b; P[op2b].next = op1b;
    }
    return true;
  }
  SZ_HD static bool overlap_range(i64 a1, i64 a2, i64 b1, i64 b2, i64 &l, i64 &r) {
    i64 alo = a1 < a2 ? a1 : a2, ahi = a1 < a2 ? a2 : a1;
    i64 blo = b1 < b2 ? b1 : b2, bhi = b1 < b2 ? b2 : b1;
    l = alo > blo ? alo : blo;
    r = ahi < bhi ? ahi : bhi;
    return l < r;
  }

  SZ_HD bool join_points(Join &j, int r1, int r2) {
    int op1 = j.a, op2 = j.b, op1b, op2b;
    bool is_h = (P[j.a].pt.y == j.off.y);
    if (is_h && same(j.off, P[j.a].pt) && same(j.off, P[j.b].pt)) {
      // touching-vertex join (only produced by the strictly-simple option; kept for completeness)
      if (r1 != r2) return false;
      op1b = P[j.a].next;
      while (op1b != op1 && same(P[op1b].pt, j.off)) op1b = P[op1b].next;
      bool rev1 = (P[op1b].pt.y > j.off.y);
      op2b = P[j.b].next;
      while (op2b != op2 && same(P[op2b].pt, j.off)) op2b = P[op2b].next;
      bool rev2 = (P[op2b].pt.y > j.off.y);
      if (rev1 == rev2) return false;
      if (rev1) {
        op1b = dup_pt(op1, false); op2b = dup_pt(op2, true);
        P[op1].prev = op2; P[op2].next = op1; P[op1b].next = op2b; P[op2b].prev = op1b;
      } else {
        op1b = dup_pt(op1, true); op2b = dup_pt(op2, false);
        P[op1].next = op2; P[op2].prev = op1; P[op1b].prev = op2b; P[op2b].next = op1b;
      }
      j.a = op1; j.b = op1b;
      return true;
    } else if (is_h) {
      // the two points lie somewhere on collinear horizontal edges; find the extents first
      op1b = op1;
      while (P[P[op1].prev].pt.y == P[op1].pt.y && P[op1].prev != op1b && P[op1].prev != op2) op1 = P[op1].prev;
      while (P[P[op1b].next].pt.y == P[op1b].pt.y && P[op1b].next != op1 && P[op1b].next != op2) op1b = P[op1b].next;
      if (P[op1b].next == op1 || P[op1b].next == op2) return false;
      op2b = op2;
      while (P[P[op2].prev].pt.y == P[op2].pt.y && P[op2].prev != op2b && P[op2].prev != op1b) op2 = P[op2].prev;
      while (P[P[op2b].next].pt.y == P[op2b].pt.y && P[op2b].next != op2 && P[op2b].next != op1) op2b = P[op2b].next;
      if (P[op2b].next == op2 || P[op2b].next == op1) return false;
      i64 l, r;
      if (!overlap_range(P[op1].pt.x, P[op1b].pt.x, P[op2].pt.x, P[op2b].pt.x, l, r)) return false;
      Pt pt;
      bool discard_left;
      if (P[op1].pt.x >= l && P[op1].pt.x <= r) { pt = P[op1].pt; discard_left = (P[op1].pt.x > P[op1b].pt.x); }
      else if (P[op2].pt.x >= l && P[op2].pt.x <= r) { pt = P[op2].pt; discard_left = (P[op2].pt.x > P[op2b].pt.x); }
      else if (P[op1b].pt.x >= l && P[op1b].pt.x <= r) { pt = P[op1b].pt; discard_left = (P[op1b].pt.x > P[op1].pt.x); }
      else { pt = P[op2b].pt; discard_left = (P[op2b].pt.x > P[op2].pt.x); }
      j.a = op1; j.b = op2;
      return join_horz(op1, op1b, op2, op2b, pt, discard_left);
    } else {
      // both points sit at the bottom of a shared sloping segment, off is above them
      op1b = P[op1].next;
      while (same(P[op1b].pt, P[op1].pt) && op1b != op1) op1b = P[op1b].next;
      bool rev1 = (P[op1b].pt.y > P[op1].pt.y) || !collinear3(P[op1].pt, P[op1b].pt, j.off);
      if (rev1) {
        op1b = P[op1].prev;
        while (same(P[op1b].pt, P[op1].pt) && op1b != op1) op1b = P[op1b].prev;
        if ((P[op1b].pt.y > P[op1].pt.y) || !collinear3(P[op1].pt, P[op1b].pt, j.off)) return false;
      }
      op2b = P[op2].next;
      while (same(P[op2b].pt, P[op2].pt) && op2b != op2) op2b = P[op2b].next;
      bool rev2 = (P[op2b].pt.y > P[op2].pt.y) || !collinear3(P[op2].pt, P[op2b].pt, j.off);
      if (rev2) {
        op2b = P[op2].prev;
        while (same(P[op2b].pt, P[op2].pt) && op2b != op2) op2b = P[op2b].prev;
        if ((P[op2b].pt.y > P[op2].pt.y) || !collinear3(P[op2].pt, P[op2b].pt, j.off)) return false;
      }
      if (op1b == op1 || op2b == op2 || op1b == op2b || (r1 == r2 && rev1 == rev2)) return false;
      if (rev1) {
        op1b = dup_pt(op1, false); op2b = dup_pt(op2, true);
        P[op1].prev = op2; P[op2].next = op1; P[op1b].next = op2b; P[op2b].prev = op1b;
      } else {
        op1b = dup_pt(op1, true); op2b = dup_pt(op2, false);
        P[op1].next = op2; P[op2].prev = op1; P[op1b].prev = op2b; P[op2b].next = op1b;
      }
      j.a = op1; j.b = op1b;
      return true;
    }
  }

  // 0 outside, +1 inside, -1 on the boundary of the ring starting at op
  SZ_HD int point_in_ring(const Pt &pt, int op) const {
    int res = 0;
    const int start = op;
    for (;;) {
      const Pt &c = P[op].pt, &n = P[P[op].next].pt;
      if (n.y == pt.y) {
        if (n.x == pt.x || (c.y == pt.y && ((n.x > pt.x) == (c.x < pt.x)))) return -1;
      }
      if ((c.y < pt.y) != (n.y < pt.y)) {
        if (c.x >= pt.x) {
          if (n.x > pt.x) res = 1 - res;
          else {
            double d = (double)(c.x - pt.x) * (double)(n.y - pt.y) - (double)(n.x - pt.x) * (double)(c.y - pt.y);
            if (!d) return -1;
            if ((d > 0) == (n.y > c.y)) res = 1 - res;
          }
        } else if (n.x > pt.x) {
          double d = (double)(c.x - pt.x) * (double)(n.y - pt.y) - (double)(n.x - pt.x) * (double)(c.y - pt.y);
          if (!d) return -1;
          if ((d > 0) == (n.y > c.y)) res = 1 - res;
        }
      }
      op = P[op].next;
      if (op == start) break;
    }
    return res;
  }
  SZ_HD bool ring_inside(int inner, int outer) const {
    int op = inner;
    do {
      int r = point_in_ring(P[op].pt, outer);
      if (r >= 0) return r > 0;
      op = P[op].next;
    } while (op != inner);
    return true;
  }

  SZ_HD void do_joins() {
    for (int i = 0; i < nJ; ++i) {
      if (err) return;
      Join &j = J[i];
      int r1 = resolve_ring(P[j.a].idx);
      int r2 = resolve_ring(P[j.b].idx);
      if (R[r1].pts == NIL || R[r2].pts == NIL) continue;
      int keep_state;
      if (r1 == r2) keep_state = r1;
      else if (right_of(r1, r2)) keep_state = r2;
      else if (right_of(r2, r1)) keep_state = r1;
      else keep_state = lowermost_ring(r1, r2);
      if (!join_points(j, r1, r2)) continue;
      if (err) return;
      if (r1 == r2) {
        // the ring was split in two
        R[r1].pts = j.a;
        R[r1].bottom = NIL;
        r2 = new_ring();
        if (err) return;
        R[r2].pts = j.b;
        { int p = R[r2].pts; do { P[p].idx = R[r2].idx; p = P[p].prev; } while (p != R[r2].pts); }
        if (ring_inside(R[r2].pts, R[r1].pts)) {
          R[r2].hole = !R[r1].hole;
          R[r2].first_left = r1;
          if (((R[r2].hole != 0) != (reverse_out != 0)) == (ring_area(R[r2].pts) > 0)) reverse_ring(R[r2].pts);
        } else if (ring_inside(R[r1].pts, R[r2].pts)) {
          R[r2].hole = R[r1].hole;
          R[r1].hole = !R[r2].hole;
          R[r2].first_left = R[r1].first_left;
          R[r1].first_left = r2;
          if (((R[r1].hole != 0) != (reverse_out != 0)) == (ring_area(R[r1].pts) > 0)) reverse_ring(R[r1].pts);
        } else {
          R[r2].hole = R[r1].hole;
          R[r2].first_left = R[r1].first_left;
        }
      } else {
        // two rings became one
        R[r2].pts = NIL;
        R[r2].bottom = NIL;
        R[r2].idx = R[r1].idx;
        R[r1].hole = R[keep_state].hole;
        if (keep_state == r2) R[r1].first_left = R[r2].first_left;
        R[r2].first_left = r1;
      }
    }
  }

  // remove duplicate points and the middle vertex of collinear triples from an output ring
  SZ_HD void tidy_ring(int r) {
    int last_ok = NIL;
    R[r].bottom = NIL;
    int pp = R[r].pts;
    for (;;) {
      if (P[pp].prev == pp || P[pp].prev == P[pp].next) {
        R[r].pts = NIL;
        return;
      }
      if (same(P[pp].pt, P[P[pp].next].pt) || same(P[pp].pt, P[P[pp].prev].pt) ||
          collinear3(P[P[pp].prev].pt, P[pp].pt, P[P[pp].next].pt)) {
        last_ok = NIL;
        int pr = P[pp].prev, nx = P[pp].next;
        P[pr].next = nx;
        P[nx].prev = pr;
        pp = pr;
      } else if (pp == last_ok) break;
      else {
        if (last_ok == NIL) last_ok = pp;
        pp = P[pp].next;
      }
    }
    R[r].pts = pp;
  }

  // ---- driver ----
  SZ_HD bool run() {
    if (nM == 0) return false;
    SZV_PHASE(13);
    gnu_sort(M, nM, LocMinLess<LocMin>());
    nB = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    if (Rec::kWave && all_plain) prebuild_beams();
#endif
    for (int i = 0; i < nM; ++i) {
      push_beam(M[i].y);
      Edge &l = E[M[i].left];
      l.cur = l.bot; l.side = SIDE_LEFT; l.ring = NIL;
      Edge &r = E[M[i].right];
      r.cur = r.bot; r.side = SIDE_RIGHT; r.ring = NIL;
    }
    ael = NIL;
    sel = NIL;
    cur_lm = 0;
    nR = nP = nJ = nG = nX = 0;
    am = NIL;
#if defined(__HIP_DEVICE_COMPILE__)
    an = Rec::kMirror ? 0 : -1;
#else
    an = -1;
#endif

    bool ok = true;
    i64 bot_y, top_y = 0;
    if (!pop_beam(bot_y)) return false;
    y_now = bot_y;
    SZV_PHASE(14);
    insert_minima(bot_y);
    for (;;) {
      if (err) break;
      bool have = pop_beam(top_y);
      if (!have && !(cur_lm < nM)) break;
      SZV_PHASE(15);
      do_horizontals();
      nG = 0;
      if (err) break;
      SZV_PHASE(16);
      if (!do_crossings(top_y)) { ok = false; break; }
      SZV_PHASE(17);
      do_top_of_beam(top_y);
      bot_y = top_y;
      SZV_PHASE(14);
      insert_minima(bot_y);
    }
    SZV_PHASE(18);
    if (err) {
      // list overflow and range errors are reported as such; an internal inconsistency is the
      // reference's "Clipper Error."
      return false;
    }
    if (!ok) return false;

    for (int i = 0; i < nR; ++i) {
      if (R[i].pts == NIL) continue;
      if (((R[i].hole != 0) != (reverse_out != 0)) == (ring_area(R[i].pts) > 0)) reverse_ring(R[i].pts);
    }
    if (nJ > 0) do_joins();
    if (err) return false;
    for (int i = 0; i < nR; ++i)
      if (R[i].pts != NIL) tidy_ring(i);
    return true;
  }

  // Emit the result rings: ring k of the solution starts at pts.prev and walks backwards
  // (clipper.cpp:3199-3217).  sink.ring_begin(), sink.vertex(Pt), sink.ring_end(); each returns
  // false when the caller's buffer is full.
  // skip_first: the first ring of the solution is left out (ClipperOffset::Execute, clipper.cpp:3913)
  template <class Sink>
  SZ_HD bool write(Sink &sink, bool skip_first = false) const {
    for (int i = 0; i < nR; ++i) {
      if (R[i].pts == NIL) continue;
      int p = P[R[i].pts].prev;
      int cnt = 0;
      { int q = p; do { ++cnt; q = P[q].next; } while (q != p); }
      if (cnt < 2) continue;
      if (skip_first) { skip_first = false; continue; }
      if (!sink.ring_begin()) return false;
      for (int k = 0; k < cnt; ++k) {
        if (!sink.vertex(P[p].pt)) return false;
        p = P[p].prev;
      }
      if (!sink.ring_end()) return false;
    }
    return true;
  }
};

typedef SweepT<Wide> Sweep;

// One boolean operation between a single subject ring and a single clip ring, evaluated as the mex
// does (mexclipper.cpp:291-304).  RingA/RingB: callable (int) -> Pt.  Returns a Status.
// workspace must hold workspace_bytes(cap_nv) bytes, cap_nv >= na + nb; simple polygons never need
// more than cap_nv = na + nb, heavily self-intersecting input may (ST_WORKSPACE -> retry larger).
template <class Rec, class RingA, class RingB, class Sink>
SZ_HD int clip_rings_t(void *workspace, const Caps &caps, const RingA &a, int na, const RingB &b, int nb,
                       int op, int fill_subject, int fill_clip, Sink &sink) {
  if (op < 0 || op > 3 || fill_subject < 0 || fill_subject > 3 || fill_clip < 0 || fill_clip > 3)
    return ST_BAD_ARG;
  SweepT<Rec> s;
  s.bind_caps(workspace, caps);
  s.op = op;
  s.fill_s = fill_subject;
  s.fill_c = fill_clip;
  SZV_PHASE(12);
  s.add_ring(a, na, KIND_SUBJECT);
  if (s.err) return s.err;
  s.add_ring(b, nb, KIND_CLIP);
  if (s.err) return s.err;
  if (!s.run()) return s.err ? s.err : ST_CLIP_ERROR;
  SZV_PHASE(19);
  if (!s.write(sink)) return ST_OUT_CAPACITY;
  return ST_OK;
}
// The same with several rings per operand (mexclipper.cpp:291-297: AddPaths(subj), AddPaths(clip), every contour of a
// struct array).  PathsA / PathsB: .count() rings, .len(k) vertices of ring k, .ring(k) -> callable (int) -> Pt.
template <class PathsA, class PathsB, class Sink>
SZ_HD int clip_paths(void *workspace, int cap_nv, const PathsA &A, const PathsB &B, int op, int fill_subject, int fill_clip, Sink &sink) {
  if (op < 0 || op > 3 || fill_subject < 0 || fill_subject > 3 || fill_clip < 0 || fill_clip > 3) return ST_BAD_ARG;
  int total = 0;
  for (int k = 0; k < A.count(); ++k) total += A.len(k);
  for (int k = 0; k < B.count(); ++k) total += B.len(k);
  if (cap_nv < total) return ST_WORKSPACE;
  Sweep s;
  s.bind_caps(workspace, caps_for(cap_nv));
  s.op = op; s.fill_s = fill_subject; s.fill_c = fill_clip;
  for (int k = 0; k < A.count(); ++k) { s.add_ring(A.ring(k), A.len(k), KIND_SUBJECT); if (s.err) return s.err; }
  for (int k = 0; k < B.count(); ++k) { s.add_ring(B.ring(k), B.len(k), KIND_CLIP); if (s.err) return s.err; }
  if (!s.run()) return s.err ? s.err : ST_CLIP_ERROR;
  if (!s.write(sink)) return ST_OUT_CAPACITY;
  return ST_OK;
}
// The clean-up of ClipperOffset::Execute (clipper.cpp:3887-3915) for the raw offset polygons A: their union with
// positive fill (delta > 0), or -- delta < 0 -- with a rectangle 10 units outside their bounds added, negative fill,
// reversed output and the first ring of the solution (the rectangle's) dropped.  cap_nv >= total vertices + 4.
struct BoxRing {
  Pt p[4];
  SZ_HD Pt operator()(int i) const { return p[i]; }
};
template <class PathsA, class Sink>
SZ_HD int offset_cleanup(void *workspace, int cap_nv, const PathsA &A, bool outset, Sink &sink) {
  int total = outset ? 0 : 4;
  for (int k = 0; k < A.count(); ++k) total += A.len(k);
  if (cap_nv < total) return ST_WORKSPACE;
  Sweep s;
  s.bind_caps(workspace, caps_for(cap_nv));
  s.op = OP_UNION;
  s.fill_s = s.fill_c = outset ? FILL_POSITIVE : FILL_NEGATIVE;
  for (int k = 0; k < A.count(); ++k) { s.add_ring(A.ring(k), A.len(k), KIND_SUBJECT); if (s.err) return s.err; }
  if (!outset) {
    // ClipperBase::GetBounds (clipper.cpp:1295-1332) over the bounds of every local minimum
    i64 left = 0, top = 0, right = 0, bottom = 0;
    for (int m = 0; m < s.nM; ++m) {
      const int lb = s.M[m].left;
      if (m == 0) { left = right = s.E[lb].bot.x; top = bottom = s.E[lb].bot.y; }
      if (s.E[lb].bot.y > bottom) bottom = s.E[lb].bot.y;
      for (int side = 0; side < 2; ++side) {
        int e = side ? s.M[m].right : lb;
        while (s.E[e].up != NIL) {
          if (s.E[e].bot.x < left) left = s.E[e].bot.x;
          if (s.E[e].bot.x > right) right = s.E[e].bot.x;
          e = s.E[e].up;
        }
        if (s.E[e].bot.x < left) left = s.E[e].bot.x;
        if (s.E[e].bot.x > right) right = s.E[e].bot.x;
        if (s.E[e].top.x < left) left = s.E[e].top.x;
        if (s.E[e].top.x > right) right = s.E[e].top.x;
        if (s.E[e].top.y < top) top = s.E[e].top.y;
      }
    }
    BoxRing box;
    box.p[0].x = left - 10; box.p[0].y = bottom + 10;
    box.p[1].x = right + 10; box.p[1].y = bottom + 10;
    box.p[2].x = right + 10; box.p[2].y = top - 10;
    box.p[3].x = left - 10; box.p[3].y = top - 10;
    s.add_ring(box, 4, KIND_SUBJECT);
    if (s.err) return s.err;
    s.reverse_out = 1;
  }
  if (s.nM == 0) return ST_OK;            // Execute returns false on no input, the solution stays empty: no error in this mode
  if (!s.run()) return s.err ? s.err : ST_OK;   // (ClipperOffset::Execute ignores the return value of Clipper::Execute)
  if (!s.write(sink, !outset)) return ST_OUT_CAPACITY;
  return ST_OK;
}
template <class RingA, class RingB, class Sink>
SZ_HD int clip_rings(void *workspace, int cap_nv, const RingA &a, int na, const RingB &b, int nb, int op,
                     int fill_subject, int fill_clip, Sink &sink) {
  if (cap_nv < na + nb) return ST_WORKSPACE;
  return clip_rings_t<Wide>(workspace, caps_for(cap_nv), a, na, b, nb, op, fill_subject, fill_clip, sink);
}
template <class RingA, class RingB, class Sink>
SZ_HD int clip_rings_lds(void *workspace, int cap_nv, const RingA &a, int na, const RingB &b, int nb, int op,
                         int fill_subject, int fill_clip, Sink &sink) {
  if (cap_nv < na + nb) return ST_WORKSPACE;
  return clip_rings_t<WideLds>(workspace, caps_for(cap_nv), a, na, b, nb, op, fill_subject, fill_clip, sink);
}
// The compact tier: even-odd fill, at most COMPACT_MAX_NV input vertices, workspace_bytes_compact() bytes.
template <class RingA, class RingB, class Sink>
SZ_HD int clip_rings_compact(void *workspace, const RingA &a, int na, const RingB &b, int nb, int op, Sink &sink) {
  if (na + nb > COMPACT_MAX_NV) return ST_WORKSPACE;
  return clip_rings_t<Compact>(workspace, caps_compact(), a, na, b, nb, op, FILL_EVENODD, FILL_EVENODD, sink);
}

} // namespace szv
