// TEST INFRASTRUCTURE ONLY (oracle) -- never linked, imported or called by the product path.
//
// C shim around the REFERENCE's own Clipper 6.4.2 sources, compiled where they lie
// (/root/reference/private/clipper.cpp + clipper.hpp; see oracle/Makefile, output
// oracle/_ref/libclipper_ref.so).  Nothing of the reference is copied into this repo.
//
// The shim restates the 12 lines of glue of the reference's mex gateway, boolean-clip mode
// (private/mexclipper.cpp:204-305): read int64 rings -> Paths, AddPaths(subj, ptSubject, true),
// AddPaths(clip, ptClip, true), Execute(CT, solution, SFT, CFT), write the rings out.
// Method numbering is the mex's (mexclipper.cpp:215-230): 0 Difference, 1 Intersection,
// 2 Xor, 3 Union; fill types 0 EvenOdd, 1 NonZero, 2 Positive, 3 Negative (:243-287).
//
// and of its offset mode (:146-203, ref_offset below).
// mexclipper.cpp itself cannot be compiled here (needs MATLAB's mex.h).
#include "clipper.hpp"
#include <stdint.h>
#include <vector>

using namespace ClipperLib;

namespace {

Paths to_paths(const int64_t *x, const int64_t *y, const int32_t *off, int nrings) {
  Paths p((size_t)nrings);
  for (int r = 0; r < nrings; ++r) {
    int a = off[r], b = off[r + 1];
    p[r].resize((size_t)(b - a));
    for (int j = a; j < b; ++j) {
      p[r][j - a].X = x[j];
      p[r][j - a].Y = y[j];
    }
  }
  return p;
}

bool map_method(int method, ClipType &ct) {
  switch (method) {
  case 0: ct = ctDifference; return true;
  case 1: ct = ctIntersection; return true;
  case 2: ct = ctXor; return true;
  case 3: ct = ctUnion; return true;
  }
  return false;
}

bool map_fill(int f, PolyFillType &pft) {
  switch (f) {
  case 0: pft = pftEvenOdd; return true;
  case 1: pft = pftNonZero; return true;
  case 2: pft = pftPositive; return true;
  case 3: pft = pftNegative; return true;
  }
  return false;
}

} // namespace

extern "C" {

// Status: 0 ok; 1 = "Clipper Error." (Execute returned false, mexclipper.cpp:304);
// 2 = bad method/fill; 3 = output capacity too small; 4 = exception from AddPath (range).
//
// subject / clip: CSR ring sets (x[], y[], off[nrings+1]).
// out: rings in solution order, vertex order exactly as Clipper returns them.
int ref_clip(const int64_t *sx, const int64_t *sy, const int32_t *soff, int snr,
             const int64_t *cx, const int64_t *cy, const int32_t *coff, int cnr,
             int method, int subj_fill, int clip_fill,
             int64_t *outx, int64_t *outy, int out_cap,
             int32_t *out_off, int ring_cap, int32_t *out_nrings) {
  ClipType ct;
  PolyFillType sft, cft;
  if (!map_method(method, ct) || !map_fill(subj_fill, sft) || !map_fill(clip_fill, cft)) return 2;
  *out_nrings = 0;
  out_off[0] = 0;
  try {
    Paths subj = to_paths(sx, sy, soff, snr);
    Paths clip = to_paths(cx, cy, coff, cnr);
    Paths solution;
    Clipper c;
    c.AddPaths(subj, ptSubject, true);
    c.AddPaths(clip, ptClip, true);
    if (!c.Execute(ct, solution, sft, cft)) return 1;
    if ((int)solution.size() > ring_cap) return 3;
    int n = 0;
    for (size_t i = 0; i < solution.size(); ++i) {
      if (n + (int)solution[i].size() > out_cap) return 3;
      for (size_t j = 0; j < solution[i].size(); ++j) {
        outx[n] = solution[i][j].X;
        outy[n] = solution[i][j].Y;
        ++n;
      }
      out_off[i + 1] = n;
    }
    *out_nrings = (int)solution.size();
  } catch (...) {
    return 4;
  }
  return 0;
}

// Orientation mode of the mex (mexclipper.cpp:134-145): Orientation(path) = Area >= 0.
int ref_orientation(const int64_t *x, const int64_t *y, int n) {
  Path p((size_t)n);
  for (int i = 0; i < n; ++i) { p[i].X = x[i]; p[i].Y = y[i]; }
  return Orientation(p) ? 1 : 0;
}

// Offset mode of the mex (mexclipper.cpp:146-203): ClipperOffset co; [ArcTolerance / MiterLimit = info when given];
// co.AddPaths(subj, jt, etClosedPolygon); co.Execute(solution, delta).  join: 0 square, 1 round, 2 miter (the mex's
// 's' / 'r' / 'm'); info <= 0: not given (the class defaults: MiterLimit 2, ArcTolerance 0.25).
// Status as ref_clip (3 = output capacity).
int ref_offset(const int64_t *sx, const int64_t *sy, const int32_t *soff, int snr, double delta, int join, double info,
               int64_t *outx, int64_t *outy, int out_cap, int32_t *out_off, int ring_cap, int32_t *out_nrings) {
  *out_nrings = 0;
  out_off[0] = 0;
  if (join < 0 || join > 2) return 2;
  try {
    Paths subj = to_paths(sx, sy, soff, snr);
    Paths solution;
    ClipperOffset co;
    if (info > 0 && join == 1) co.ArcTolerance = info;
    if (info > 0 && join == 2) co.MiterLimit = info;
    co.AddPaths(subj, join == 0 ? jtSquare : join == 1 ? jtRound : jtMiter, etClosedPolygon);
    co.Execute(solution, delta);
    if ((int)solution.size() > ring_cap) return 3;
    int n = 0;
    for (size_t i = 0; i < solution.size(); ++i) {
      if (n + (int)solution[i].size() > out_cap) return 3;
      for (size_t j = 0; j < solution[i].size(); ++j) { outx[n] = solution[i][j].X; outy[n] = solution[i][j].Y; ++n; }
      out_off[i + 1] = n;
    }
    *out_nrings = (int)solution.size();
  } catch (...) {
    return 4;
  }
  return 0;
}

// Batch form for tests and for the cpu_baseline leg of bench.py: single-ring subject and
// clip per pair (the only form polyclip.m:72 allows), pairs given as CSR over one vertex pool.
// Results are written as a CSR of rings per pair.  status[p] as ref_clip.
int ref_clip_batch(int npairs,
                   const int64_t *ax, const int64_t *ay, const int32_t *aoff,
                   const int64_t *bx, const int64_t *by, const int32_t *boff,
                   int method,
                   int64_t *outx, int64_t *outy, int64_t out_cap,
                   int32_t *ring_off, int64_t ring_cap,  // ring_off[0..total_rings]
                   int32_t *pair_ring_off,               // [npairs+1]
                   int32_t *status) {
  ClipType ct;
  if (!map_method(method, ct)) return 2;
  int64_t nv = 0, nr = 0;
  pair_ring_off[0] = 0;
  ring_off[0] = 0;
  for (int p = 0; p < npairs; ++p) {
    status[p] = 0;
    try {
      Path s((size_t)(aoff[p + 1] - aoff[p])), c((size_t)(boff[p + 1] - boff[p]));
      for (int j = aoff[p]; j < aoff[p + 1]; ++j) { s[j - aoff[p]].X = ax[j]; s[j - aoff[p]].Y = ay[j]; }
      for (int j = boff[p]; j < boff[p + 1]; ++j) { c[j - boff[p]].X = bx[j]; c[j - boff[p]].Y = by[j]; }
      Paths subj(1, s), clip(1, c), solution;
      Clipper cl;
      cl.AddPaths(subj, ptSubject, true);
      cl.AddPaths(clip, ptClip, true);
      if (!cl.Execute(ct, solution, pftEvenOdd, pftEvenOdd)) {
        status[p] = 1;
      } else {
        for (size_t i = 0; i < solution.size(); ++i) {
          if (nr + 1 > ring_cap || nv + (int64_t)solution[i].size() > out_cap) return 3;
          for (size_t j = 0; j < solution[i].size(); ++j) {
            outx[nv] = solution[i][j].X;
            outy[nv] = solution[i][j].Y;
            ++nv;
          }
          ++nr;
          ring_off[nr] = (int32_t)nv;
        }
      }
    } catch (...) {
      status[p] = 4;
    }
    pair_ring_off[p + 1] = (int32_t)nr;
  }
  return 0;
}

} // extern "C"
