// clip.hip -- batched polygon boolean operations on gfx950: the B1 surface of include/subzero_hip.h
// (replaces /root/reference/private/mexclipper.cpp:204-305 as called from polyclip.m:73).
//
// Mapping to the hardware (irregular integer/FP64 geometry, no dense contraction, so no MFMA):
//  * single rings in fixed point (szClipBatch): k_clip_front, a wave per pair (staging, boxes, boxes-apart answers,
//    trimming, slice class), then k_clip_sweep per slice class side by side: one sweep (szv::Sweep, vatti.hpp) per
//    wave with its records in 20-160 KB of LDS -- see the comment in front of k_clip_front;
//  * everything else (several contours, double input, retries): k_clip, one lane per ring pair, a few pairs per wave;
//  * every pair owns a workspace slice of one HBM arena; slice and result-slot offsets
//    follow in closed form from the CSR offsets of the input rings (no scan, no allocator);
//  * pairs whose lists outgrow their slice (only heavily self-intersecting input does) are appended to
//    a retry list with one wave-aggregated atomic and re-run at 8x / 64x capacity -- on the GPU.
//    There is no CPU fallback anywhere in this file.
#include <stdlib.h>
#include <algorithm>
#include <cmath>
#include <vector>
#if defined(SZ_CLIP_CLOCKS)
// diagnostic build: time per part of the sweep (vatti.hpp SZV_PHASE), one pair per workgroup
#include <hip/hip_runtime.h>
__device__ unsigned long long sz_ph_t[1 << 16], sz_ph_acc[(1 << 16) * 16];
__device__ int sz_ph_k[1 << 16];
#endif
#if defined(SZ_CLIP_CLOCKS) && defined(__HIP_DEVICE_COMPILE__)
#define SZV_PHASE(k) do { const unsigned b_ = blockIdx.x; if (b_ < 65536u) { const unsigned long long now_ = wall_clock64(); \
  sz_ph_acc[b_ * 16 + sz_ph_k[b_]] += now_ - sz_ph_t[b_]; sz_ph_t[b_] = now_; sz_ph_k[b_] = (k) - 12; } } while (0)
#endif
#include "clip_batch.hpp"
#include "offset.hpp"
#include "trim.hpp"
#include "teamclip.hpp"

using szb::ClipArgs;

namespace {

using szv::i64;
using szv::Pt;

// result slot capacities per pair, nv = (na + nb) * mult
__host__ __device__ inline i64 slot_verts(i64 nv) { return 2 * nv + 8; }
__host__ __device__ inline i64 slot_rings(i64 nv) { return nv + 4; } // + 1 entry for the closing offset

struct RingI64 {
  const i64 *x, *y;
  __device__ Pt operator()(int i) const { Pt p; p.x = x[i]; p.y = y[i]; return p; }
};

// MATLAB int64(v * 2^32): round half away from zero, saturating, NaN -> 0 (polyclip.m:63-66)
__host__ __device__ inline i64 to_fixed(double v) {
  double s = v * 4294967296.0;
  if (s != s) return 0;
  if (s >= 9223372036854775807.0) return 0x7FFFFFFFFFFFFFFFLL;
  if (s <= -9223372036854775808.0) return (i64)0x8000000000000000ULL;
  double t = (s < 0) ? -floor(-s) : floor(s);
  double f = s - t;
  if (f >= 0.5) t += 1.0;
  else if (f <= -0.5) t -= 1.0;
  return (i64)t;
}
struct RingF64 {
  const double *x, *y;
  __device__ Pt operator()(int i) const { Pt p; p.x = to_fixed(x[i]); p.y = to_fixed(y[i]); return p; }
};

// several rings per operand (szClipPathsBatch): rings ring0 .. ring0 + n of a ring CSR over the vertex pool
struct PathsI64 {
  const i64 *x, *y; const int *ring_off; int ring0, n;
  __device__ int count() const { return n; }
  __device__ int len(int k) const { return ring_off[ring0 + k + 1] - ring_off[ring0 + k]; }
  __device__ RingI64 ring(int k) const { const int v = ring_off[ring0 + k]; RingI64 r = {x + v, y + v}; return r; }
};

enum { CLIP_OFFSET_OUTSET = 4, CLIP_OFFSET_INSET = 5 };   // ClipArgs::method beyond the four boolean operations

struct SlotSink {
  i64 *ox, *oy;
  int *roff;
  int cap_v, cap_r, nv, nr;
  __device__ bool ring_begin() { return nr < cap_r; }
  __device__ bool vertex(const Pt &p) {
    if (nv >= cap_v) return false;
    ox[nv] = p.x; oy[nv] = p.y; ++nv;
    return true;
  }
  __device__ bool ring_end() { ++nr; roff[nr] = nv; return true; }
};


// the even-odd sweep with its records in a slice of LDS (a function of its own, so that the compiler specialises this
// copy of the sweep for ds_ addressing)
__device__ __attribute__((noinline, flatten)) int clip_in_slice(void *slice, const szv::Caps &caps, const RingI64 &ra, int na,
                                                       const RingI64 &rb, int nb, int method, int fill_s, int fill_c,
                                                       SlotSink &sink) {
  __builtin_assume(__builtin_amdgcn_is_shared((const __attribute__((address_space(0))) void *)slice));
  return szv::clip_rings_t<szv::WideLds>(slice, caps, ra, na, rb, nb, method, fill_s, fill_c, sink);
}
// (for the sweep pass, where all 64 lanes of the wave run the sweep alike: vatti.hpp WideLdsWaveMirror)
__device__ __attribute__((noinline, flatten)) int clip_in_slice_wave(void *slice, const szv::Caps &caps, const RingI64 &ra, int na,
                                                            const RingI64 &rb, int nb, int method, int fill_s, int fill_c,
                                                            SlotSink &sink) {
  __builtin_assume(__builtin_amdgcn_is_shared((const __attribute__((address_space(0))) void *)slice));
  return szv::clip_rings_t<szv::WideLdsWaveMirror>(slice, caps, ra, na, rb, nb, method, fill_s, fill_c, sink);
}
// (the same inlined into its caller: k_clip_sweep_small, a kernel compiled for four waves per SIMD -- a function of its
// own is compiled for all 256 registers whatever its caller asks for)
__device__ __attribute__((always_inline)) int clip_in_slice_inlined(void *slice, const szv::Caps &caps, const RingI64 &ra, int na,
                                                                    const RingI64 &rb, int nb, int method, int fill_s, int fill_c,
                                                                    SlotSink &sink) {
  __builtin_assume(__builtin_amdgcn_is_shared((const __attribute__((address_space(0))) void *)slice));
  return szv::clip_rings_t<szv::WideLdsWaveMirror>(slice, caps, ra, na, rb, nb, method, fill_s, fill_c, sink);
}

#if defined(SZ_CLIP_CLOCKS)
// diagnostic build (make diag_clip): per pair the 100 MHz clock at start, after trimming and at the end, the trimmed
// size and where the records were
__device__ unsigned long long sz_clip_clocks[5 * (1 << 16)];
#define SZ_CLIP_CLOCK(k, v) do { if (t < (1 << 16)) sz_clip_clocks[5 * t + (k)] = (v); } while (0)
#else
#define SZ_CLIP_CLOCK(k, v) ((void)0)
#endif
__global__ void __launch_bounds__(64) k_clip(ClipArgs a) {
  extern __shared__ __attribute__((aligned(16))) char clip_lds[];
  if ((int)threadIdx.x >= a.lanes) return;
  int t = blockIdx.x * a.lanes + threadIdx.x;
  int n = a.list ? a.nlist : a.npairs;
  if (t >= n) return;
  int p = a.list ? a.list[t] : t;
  SZ_CLIP_CLOCK(0, wall_clock64());
#if defined(SZ_CLIP_CLOCKS) && defined(__HIP_DEVICE_COMPILE__)
  if (blockIdx.x < 65536u) { sz_ph_t[blockIdx.x] = wall_clock64(); sz_ph_k[blockIdx.x] = 0; for (int k = 0; k < 16; ++k) sz_ph_acc[blockIdx.x * 16 + k] = 0; }
#endif
  int a0 = a.aoff[p], na = a.aoff[p + 1] - a0;
  int b0 = a.boff[p], nb = a.boff[p + 1] - b0;
  const int ra0 = a0, nra = na, rb0 = b0, nrb = nb;   // (ring ranges when the operands have several contours)
  if (a.aring) {
    a0 = a.aring[ra0]; na = a.aring[ra0 + nra] - a0;
    b0 = a.bring[rb0]; nb = a.bring[rb0 + nrb] - b0;
  }
  i64 nv = (i64)(na + nb) * a.mult;
  i64 wbytes = (i64)szv::WS_PER_VERTEX * nv + szv::WS_PER_CLIP;
  i64 cv = slot_verts(nv), cr = slot_rings(nv);
  i64 woff, voff, roff;
  if (!a.list) {
    i64 before = (i64)a0 + b0;
    woff = (i64)szv::WS_PER_VERTEX * before + (i64)szv::WS_PER_CLIP * p;
    voff = 2 * before + 8 * (i64)p;
    roff = before + 5 * (i64)p;
  } else {
    woff = (i64)atomicAdd(&a.counters[1], (unsigned long long)wbytes);
    voff = (i64)atomicAdd(&a.counters[2], (unsigned long long)cv);
    roff = (i64)atomicAdd(&a.counters[3], (unsigned long long)(cr + 1));
  }
  int st;
  if (woff + wbytes > a.ws_cap || voff + cv > a.ov_cap || roff + cr + 1 > a.or_cap || nv > 0x3FFFFFFF) {
    st = szv::ST_WORKSPACE; // arena exhausted: reported, never silent
    a.nrings[p] = 0;
    a.status[p] = st;
    atomicAdd(&a.counters[7], 1ULL);
    return;
  }
  SlotSink sink;
  sink.ox = a.ovx + voff; sink.oy = a.ovy + voff; sink.roff = a.oroff + roff;
  sink.cap_v = (int)cv; sink.cap_r = (int)cr; sink.nv = 0; sink.nr = 0;
  sink.roff[0] = 0;
  if (a.aring && a.method >= CLIP_OFFSET_OUTSET) {
    // clean-up of the raw offset polygons (ClipperOffset::Execute); the clip operand only reserves room for the box
    PathsI64 pa = {a.ax, a.ay, a.aring, ra0, nra};
    st = szv::offset_cleanup(a.ws + woff, (int)nv, pa, a.method == CLIP_OFFSET_OUTSET, sink);
  } else if (a.aring) {
    PathsI64 pa = {a.ax, a.ay, a.aring, ra0, nra}, pb = {a.bx, a.by, a.bring, rb0, nrb};
    st = szv::clip_paths(a.ws + woff, (int)nv, pa, pb, a.method, a.fill_s, a.fill_c, sink);
  } else if (a.fax) {
    RingF64 ra = {a.fax + a0, a.fay + a0}, rb = {a.fbx + b0, a.fby + b0};
    st = szv::clip_rings(a.ws + woff, (int)nv, ra, na, rb, nb, a.method, a.fill_s, a.fill_c, sink);
  } else {
    RingI64 ra = {a.ax + a0, a.ay + a0}, rb = {a.bx + b0, a.by + b0};
    int ta = na, tb = nb, cap = (int)nv;
    // An intersection sees only what lies in the box the two rings share (trim.hpp: the reference's Clipper returns the
    // same rings vertex for vertex on the trimmed input; the step's general path has used it since round 1): real floe
    // outlines that overlap in part lose most of their vertices before the sweep.  The trimmed copies sit at the end of
    // the pair's workspace, which the sweep of fewer vertices leaves idle; when the cut is too small to make that room
    // (or a tie forbids it) the rings go in whole.
    if (a.method == szv::OP_INTERSECTION && a.fill_s == szv::FILL_EVENODD && a.fill_c == szv::FILL_EVENODD && na + nb >= 24) {
      i64 *tx = (i64 *)(a.ws + woff + wbytes - 16 * (i64)(na + nb)), *ty = tx + (na + nb);
      for (int k = 0; k < na; ++k) { tx[k] = ra.x[k]; ty[k] = ra.y[k]; }
      for (int k = 0; k < nb; ++k) { tx[na + k] = rb.x[k]; ty[na + k] = rb.y[k]; }
      if (szt::trim_pair(tx, ty, ta, tx + na, ty + na, tb) &&
          (i64)szv::WS_PER_VERTEX * (i64)(ta + tb) * a.mult + szv::WS_PER_CLIP + 16 * (i64)(na + nb) <= wbytes) {
        RingI64 sa = {tx, ty}, sb = {tx + na, ty + na};
        ra = sa; rb = sb; cap = (ta + tb) * a.mult;
      } else { ta = na; tb = nb; }
    }
    // The sweep is a chain of dependent reads and writes of its records: with the records in the lane's slice of LDS
    // (ds_ addressing, a fraction of the latency of the HBM workspace) when the trimmed rings fit it at capacities cut
    // to what simple outlines use (vatti.hpp caps_budget); a sweep that outgrows the slice is repeated in HBM.
    st = szv::ST_WORKSPACE;
    SZ_CLIP_CLOCK(1, wall_clock64());
    SZ_CLIP_CLOCK(3, (unsigned long long)(ta + tb));
    SZ_CLIP_CLOCK(4, 0ULL);
    szv::Caps lc;
    if (a.lds_per_lane > 0 && a.mult == 1 && szv::caps_budget<szv::WideLds>(ta + tb, (size_t)a.lds_per_lane, lc) &&
        lc.pts >= ta + tb + 8) {
      st = clip_in_slice((void *)(clip_lds + (size_t)threadIdx.x * a.lds_per_lane), lc, ra, ta, rb, tb, a.method, a.fill_s, a.fill_c, sink);
      SZ_CLIP_CLOCK(4, st == szv::ST_WORKSPACE ? 2ULL : 1ULL);
      if (st == szv::ST_WORKSPACE) { sink.nv = 0; sink.nr = 0; sink.roff[0] = 0; atomicAdd(&a.counters[8], 1ULL); }
    }
    if (st == szv::ST_WORKSPACE)
      st = szv::clip_rings(a.ws + woff, cap, ra, ta, rb, tb, a.method, a.fill_s, a.fill_c, sink);
  }
  SZ_CLIP_CLOCK(2, wall_clock64());
#if defined(SZ_CLIP_CLOCKS) && defined(__HIP_DEVICE_COMPILE__)
  SZV_PHASE(12);
#endif
  a.ov_base[p] = voff;
  a.or_base[p] = roff;
  a.status[p] = st;
  if (st == szv::ST_OK) {
    a.nrings[p] = sink.nr;
    atomicAdd(&a.counters[4], (unsigned long long)sink.nr);
    atomicAdd(&a.counters[5], (unsigned long long)sink.nv);
  } else {
    a.nrings[p] = 0;
    if (st == szv::ST_WORKSPACE || st == szv::ST_OUT_CAPACITY) {
      unsigned long long k = atomicAdd(&a.counters[0], 1ULL);
      if (k < (unsigned long long)a.retry_cap) a.retry[k] = p;
    } else {
      atomicAdd(&a.counters[6], 1ULL);
    }
  }
}


// ---- the main pass over single rings in fixed point: a front pass and a sweep pass per slice class -------------------
// The sweep itself is one lane's chain of dependent accesses; what comes before it is not.  In the front pass the 64
// lanes of a wave stage both rings of a pair in LDS (coalesced), reduce their boxes, and -- for an intersection --
//   * find rings whose boxes are apart (trim.hpp: nothing in common, no sweep), or
//   * cut the rings down to the common box: lane 0 runs trim_ring on the staged copy (a serial scan, now of LDS), the
//     wave checks the heights of what is kept (the O(n^2) part) and writes the short rings to the end of the pair's HBM
//     workspace, where the serial path puts them too.
// The sweep pass then runs one launch per slice class, concurrently: most pairs sweep in 20 KB of LDS, eight to a CU;
// the few long ones get 40, 80 or all 160 KB (a batch lasts as long as its longest sweep: those start at once and
// run in LDS too); what fits no slice keeps its records in HBM.
using szt::wave_box;
using szt::wave_heights_distinct;

// slice classes of the sweep pass: bytes of LDS per pair; class CLASS_HBM keeps its records in the HBM workspace
enum { CLIP_CLASSES = 5, CLASS_HBM = 5, META_TAIL = 1 << 30 };
__device__ __constant__ int clip_class_bytes[CLIP_CLASSES] = {10 * 1024, 20 * 1024, 40 * 1024, 80 * 1024, 160 * 1024};
__device__ int slice_class(int nv) {
  szv::Caps c;
  for (int k = 0; k < CLIP_CLASSES; ++k)
    if (szv::caps_budget<szv::WideLds>(nv, (size_t)clip_class_bytes[k], c) && c.pts >= nv + 8) return k;
  return CLASS_HBM;
}
struct PairSlots { i64 nv, wbytes, cv, cr, woff, voff, roff; };
__device__ bool pair_slots(const ClipArgs &a, int p, int a0, int na, int b0, int nb, PairSlots &s) {
  s.nv = (i64)(na + nb);
  s.wbytes = (i64)szv::WS_PER_VERTEX * s.nv + szv::WS_PER_CLIP;
  s.cv = slot_verts(s.nv); s.cr = slot_rings(s.nv);
  const i64 before = (i64)a0 + b0;
  s.woff = (i64)szv::WS_PER_VERTEX * before + (i64)szv::WS_PER_CLIP * p;
  s.voff = 2 * before + 8 * (i64)p; s.roff = before + 5 * (i64)p;
  return !(s.woff + s.wbytes > a.ws_cap || s.voff + s.cv > a.ov_cap || s.roff + s.cr + 1 > a.or_cap || s.nv > 0x3FFFFFFF);
}

// does the ring (open, n vertices, no two of a height) hold three consecutive vertices that are not on one line?  (by the wave)
__device__ inline bool wave_ring_proper(const i64 *x, const i64 *y, int n) {
  if (n < 3) return false;
  bool ok = false;
  for (int i = (int)(threadIdx.x & 63); i < n; i += 64) {
    const int p = i == 0 ? n - 1 : i - 1, q = i + 1 == n ? 0 : i + 1;
    szv::Pt a, b, c;
    a.x = x[p]; a.y = y[p]; b.x = x[i]; b.y = y[i]; c.x = x[q]; c.y = y[q];
    ok |= !szv::collinear3(a, b, c);
  }
  return __ballot(ok) != 0ULL;
}

// Front pass, one pair per wave: staging, boxes, the pairs that are apart (finished here), trimming; every other pair is
// appended to the list of the smallest slice class that holds the records of its (trimmed) rings.
__global__ void __launch_bounds__(64) k_clip_front(ClipArgs a) {
  extern __shared__ __attribute__((aligned(16))) char clip_lds[];
  const int t = blockIdx.x, p = t, lane = (int)threadIdx.x;
  if (t >= a.npairs) return;
  if (lane == 0) SZ_CLIP_CLOCK(0, wall_clock64());
  const int a0 = a.aoff[p], na = a.aoff[p + 1] - a0;
  const int b0 = a.boff[p], nb = a.boff[p + 1] - b0;
  PairSlots s;
  if (!pair_slots(a, p, a0, na, b0, nb, s)) {
    if (lane == 0) { a.nrings[p] = 0; a.status[p] = szv::ST_WORKSPACE; atomicAdd(&a.counters[7], 1ULL); }
    return;
  }
  const i64 nv = s.nv;
  RingI64 ra = {a.ax + a0, a.ay + a0}, rb = {a.bx + b0, a.by + b0};
  const bool evenodd = a.fill_s == szv::FILL_EVENODD && a.fill_c == szv::FILL_EVENODD;
  // Two short rings (what a Voronoi field hands polyclip 'int', floe_interactions.m:29): the team form of the clip
  // first -- teamclip.hpp, the clip inside the step's pair kernel: one scanline per lane of a 32-lane team, a few
  // microseconds where the sweep of one lane takes 70+.  It answers the plain case (two convex rings in general
  // position, bit-identical to the reference's Execute: tests/test_team_clip_gpu.py) and hands everything else on.
  if (a.method == szv::OP_INTERSECTION && evenodd && !a.no_team && na >= 3 && nb >= 3 && na <= (int)sztm::T_NV + 1 && nb <= (int)sztm::T_NV + 1) {
    __shared__ sztm::ClipScratch team_scr;
    __shared__ i64 team_ox[sztm::T_CAPV], team_oy[sztm::T_CAPV];
    __shared__ int team_res[2];
    int ea = na, eb = nb;
    if (ra.x[ea - 1] == ra.x[0] && ra.y[ea - 1] == ra.y[0]) --ea;
    if (rb.x[eb - 1] == rb.x[0] && rb.y[eb - 1] == rb.y[0]) --eb;
    if (ea <= (int)sztm::T_NV && eb <= (int)sztm::T_NV) {
      if (lane < 32) {
        const sztm::Team<32> T = sztm::Team<32>::here();
        for (int i = T.lane; i < ea; i += 32) { team_scr.fx[0][i] = ra.x[i]; team_scr.fy[0][i] = ra.y[i]; }
        for (int i = T.lane; i < eb; i += 32) { team_scr.fx[1][i] = rb.x[i]; team_scr.fy[1][i] = rb.y[i]; }
        T.sync(TEAM_SITE);
        int n = 0;
        const int st = sztm::team_clip(T, &team_scr, ea, eb, team_ox, team_oy, &n);
        T.sync(TEAM_SITE);
        if (T.lane == 0) { team_res[0] = st; team_res[1] = n; }
      }
      __syncthreads();
      const int st = team_res[0], n = team_res[1];
      if (st == szv::ST_OK && (i64)n <= s.cv && s.cr >= 2) {
        for (int i = lane; i < n; i += 64) { a.ovx[s.voff + i] = team_ox[i]; a.ovy[s.voff + i] = team_oy[i]; }
        if (lane == 0) {
          a.oroff[s.roff] = 0;
          if (n) a.oroff[s.roff + 1] = n;
          a.ov_base[p] = s.voff; a.or_base[p] = s.roff;
          a.status[p] = szv::ST_OK; a.nrings[p] = n ? 1 : 0;
        }
        return;
      }
      __syncthreads();
    }
  }
  i64 *tx = (i64 *)(a.ws + s.woff + s.wbytes - 16 * nv), *ty = tx + (na + nb);
  int ta = na, tb = nb;
  bool apart = false, tail = false;
  const bool staged = a.method == szv::OP_INTERSECTION && na > 0 && nb > 0 && 16 * nv <= (i64)a.lds_per_lane;
  if (staged) {
    i64 *sx = (i64 *)clip_lds, *sy = sx + (na + nb);
    for (int k = lane; k < na; k += 64) { sx[k] = ra.x[k]; sy[k] = ra.y[k]; }
    for (int k = lane; k < nb; k += 64) { sx[na + k] = rb.x[k]; sy[na + k] = rb.y[k]; }
    __syncthreads();
    szt::Box A, B, q;
    wave_box(sx, sy, na, A); wave_box(sx + na, sy + na, nb, B);
    szt::common_box(A, B, q);
    if (szt::box_empty(q)) {
      // apart: no sweep when neither ring can raise the reference's errors on its own -- in range, and in general
      // position as far as the trimming asks it of the vertices it keeps (no two vertices of a height)
      if (szt::box_in_range(A) && szt::box_in_range(B)) {
        const int ea = (na > 1 && sx[na - 1] == sx[0] && sy[na - 1] == sy[0]) ? na - 1 : na;
        const int eb = (nb > 1 && sx[na + nb - 1] == sx[na] && sy[na + nb - 1] == sy[na]) ? nb - 1 : nb;
        // (... and at least one ring that AddPath keeps: with both rings degenerate -- two distinct points and a closing
        // copy, or every vertex on one line -- Execute has no local minimum and fails, clipper.cpp:1560-1570 -> "Clipper
        // Error."; a ring with distinct heights and a consecutive triple that is not collinear keeps three vertices)
        apart = wave_heights_distinct(sy, ea, sy + na, eb) && (wave_ring_proper(sx, sy, ea) || wave_ring_proper(sx + na, sy + na, eb));
      }
    } else if (evenodd && na + nb >= 24) {
      ta = szt::trim_ring_wave(sx, sy, na, q);       // (by the wave; rings of more than 128 vertices: every lane the serial scan alike)
      tb = szt::trim_ring_wave(sx + na, sy + na, nb, q);
      __syncthreads();
      if (ta != na || tb != nb) {
        // (an untouched ring may still carry its closing duplicate: not a tie -- trim_pair)
        const int ea = (ta == na && na > 1 && sx[na - 1] == sx[0] && sy[na - 1] == sy[0]) ? na - 1 : ta;
        const int eb = (tb == nb && nb > 1 && sx[na + nb - 1] == sx[na] && sy[na + nb - 1] == sy[na]) ? nb - 1 : tb;
        const bool room = (i64)szv::WS_PER_VERTEX * (i64)(ta + tb) + szv::WS_PER_CLIP + 16 * nv <= s.wbytes;
        if (room && wave_heights_distinct(sy, ea, sy + na, eb)) {
          for (int k = lane; k < ta; k += 64) { tx[k] = sx[k]; ty[k] = sy[k]; }
          for (int k = lane; k < tb; k += 64) { tx[na + k] = sx[na + k]; ty[na + k] = sy[na + k]; }
          tail = true;
        } else { ta = na; tb = nb; }
      }
    }
  }
  if (lane != 0) return;
  if (!staged && a.method == szv::OP_INTERSECTION && evenodd && na + nb >= 24) {
    // rings too long to stage in the slice: the serial trimming of k_clip
    for (int k = 0; k < na; ++k) { tx[k] = ra.x[k]; ty[k] = ra.y[k]; }
    for (int k = 0; k < nb; ++k) { tx[na + k] = rb.x[k]; ty[na + k] = rb.y[k]; }
    if (szt::trim_pair(tx, ty, ta, tx + na, ty + na, tb) &&
        (i64)szv::WS_PER_VERTEX * (i64)(ta + tb) + szv::WS_PER_CLIP + 16 * nv <= s.wbytes) tail = true;
    else { ta = na; tb = nb; }
  }
  SZ_CLIP_CLOCK(1, wall_clock64());
  SZ_CLIP_CLOCK(3, (unsigned long long)(apart ? 0 : ta + tb));
  SZ_CLIP_CLOCK(4, apart ? 3ULL : 0ULL);
  if (apart) {
    a.oroff[s.roff] = 0;
    a.ov_base[p] = s.voff; a.or_base[p] = s.roff;
    a.status[p] = szv::ST_OK; a.nrings[p] = 0;
    SZ_CLIP_CLOCK(2, wall_clock64());
    return;
  }
  a.meta[2 * p] = ta | (tail ? META_TAIL : 0);
  a.meta[2 * p + 1] = tb;
  const int c = slice_class(ta + tb);
  const unsigned long long k = atomicAdd(&a.counters[9 + c], 1ULL);
  a.class_list[(size_t)c * a.npairs + k] = p;
}

// Sweep pass over one class list: lane 0 of each wave sweeps one pair with its records in `slice` bytes of LDS
// (clip_in_slice), or in the pair's HBM workspace (slice 0, or a sweep that outgrew the slice after all).
template <bool SMALL>
__device__ __attribute__((always_inline)) void clip_sweep_body(const ClipArgs &a, const int *list, int count, int slice) {
  extern __shared__ __attribute__((aligned(16))) char clip_lds[];
  // (all 64 lanes run the pair alike -- same reads, same plain stores -- so that the sweep in LDS can hand its walk
  // over the active edges to them; lane 0 alone touches counters and lists)
  const int t = blockIdx.x;
  if (t >= count) return;
  const bool speaker = threadIdx.x == 0;
  const int p = list[t];
  const int a0 = a.aoff[p], na = a.aoff[p + 1] - a0;
  const int b0 = a.boff[p], nb = a.boff[p + 1] - b0;
  PairSlots s;
  pair_slots(a, p, a0, na, b0, nb, s);
  const int m0 = a.meta[2 * p], ta = m0 & ~META_TAIL, tb = a.meta[2 * p + 1];
  RingI64 ra = {a.ax + a0, a.ay + a0}, rb = {a.bx + b0, a.by + b0};
  if (m0 & META_TAIL) {
    i64 *tx = (i64 *)(a.ws + s.woff + s.wbytes - 16 * s.nv), *ty = tx + (na + nb);
    RingI64 sa = {tx, ty}, sb = {tx + na, ty + na};
    ra = sa; rb = sb;
  }
  SlotSink sink;
  sink.ox = a.ovx + s.voff; sink.oy = a.ovy + s.voff; sink.roff = a.oroff + s.roff;
  sink.cap_v = (int)s.cv; sink.cap_r = (int)s.cr; sink.nv = 0; sink.nr = 0;
  sink.roff[0] = 0;
  int st = szv::ST_WORKSPACE;
  szv::Caps lc;
#if defined(SZ_CLIP_CLOCKS)
  const unsigned long long t_in = wall_clock64();
#endif
  if (slice > 0 && szv::caps_budget<szv::WideLds>(ta + tb, (size_t)slice, lc) && lc.pts >= ta + tb + 8) {
    st = SMALL ? clip_in_slice_inlined((void *)clip_lds, lc, ra, ta, rb, tb, a.method, a.fill_s, a.fill_c, sink)
               : clip_in_slice_wave((void *)clip_lds, lc, ra, ta, rb, tb, a.method, a.fill_s, a.fill_c, sink);
#if defined(SZ_CLIP_CLOCKS)
    if (p < (1 << 16)) sz_clip_clocks[5 * p + 4] = st == szv::ST_WORKSPACE ? 2ULL : 1ULL;
#endif
    if (st == szv::ST_WORKSPACE) { sink.nv = 0; sink.nr = 0; sink.roff[0] = 0; if (speaker) atomicAdd(&a.counters[8], 1ULL); }
  }
  // (a sweep that outgrew its slice: again in the pair's HBM workspace -- in the small-class kernel it goes to the retry
  // tier instead, which keeps the second copy of the sweep, and its registers, out of that kernel)
  if (st == szv::ST_WORKSPACE && !SMALL)
    st = szv::clip_rings(a.ws + s.woff, ta + tb, ra, ta, rb, tb, a.method, a.fill_s, a.fill_c, sink);
#if defined(SZ_CLIP_CLOCKS)
  if (p < (1 << 16)) { sz_clip_clocks[5 * p + 1] = t_in; sz_clip_clocks[5 * p + 2] = wall_clock64(); }
#endif
  a.ov_base[p] = s.voff;
  a.or_base[p] = s.roff;
  a.status[p] = st;
  if (st == szv::ST_OK) {
    a.nrings[p] = sink.nr;
    if (speaker) {
      atomicAdd(&a.counters[4], (unsigned long long)sink.nr);
      atomicAdd(&a.counters[5], (unsigned long long)sink.nv);
    }
  } else {
    a.nrings[p] = 0;
    if (st == szv::ST_WORKSPACE || st == szv::ST_OUT_CAPACITY) {
      if (speaker) {
        unsigned long long k = atomicAdd(&a.counters[0], 1ULL);
        if (k < (unsigned long long)a.retry_cap) a.retry[k] = p;
      }
    } else {
      if (speaker) atomicAdd(&a.counters[6], 1ULL);
    }
  }
}

__global__ void __launch_bounds__(64) k_clip_sweep(ClipArgs a, const int *list, int count, int slice) {
  clip_sweep_body<false>(a, list, count, slice);
}
// the 10 KB class: 128 registers, four waves per SIMD, sixteen sweeps per CU (the sweep is a chain of LDS round trips:
// what a SIMD gets done grows with the waves it can switch between)
__global__ void __launch_bounds__(64) __attribute__((flatten, amdgpu_waves_per_eu(4, 4))) k_clip_sweep_small(ClipArgs a, const int *list, int count, int slice) {
  clip_sweep_body<true>(a, list, count, slice);
}

} // namespace

// Sizes and (re)allocates the context's clip buffers for a batch; fills the common ClipArgs fields.
int szb::prepare(szContext *ctx, int npairs, i64 total, ClipArgs &a) {
  i64 ws_main = (i64)szv::WS_PER_VERTEX * total + (i64)szv::WS_PER_CLIP * npairs;
  i64 ov_main = 2 * total + 8 * (i64)npairs;
  i64 or_main = total + 5 * (i64)npairs;
  // reserve for retries: room for 1/16 of the batch at 8x, at least 4 MiB of workspace
  i64 ws_res = ws_main / 2 + (4 << 20), ov_res = ov_main / 2 + 65536, or_res = or_main / 2 + 65536;
  int rc;
  if ((rc = sz_reserve(ctx, ctx->ws, (size_t)(ws_main + ws_res)))) return rc;
  if ((rc = sz_reserve(ctx, ctx->ovx, (size_t)(ov_main + ov_res) * 8))) return rc;
  if ((rc = sz_reserve(ctx, ctx->ovy, (size_t)(ov_main + ov_res) * 8))) return rc;
  if ((rc = sz_reserve(ctx, ctx->oroff, (size_t)(or_main + or_res) * 4))) return rc;
  if ((rc = sz_reserve(ctx, ctx->ov_base, (size_t)npairs * 8 + 8))) return rc;
  if ((rc = sz_reserve(ctx, ctx->or_base, (size_t)npairs * 8 + 8))) return rc;
  if ((rc = sz_reserve(ctx, ctx->nrings, (size_t)npairs * 4 + 4))) return rc;
  if ((rc = sz_reserve(ctx, ctx->status, (size_t)npairs * 4 + 4))) return rc;
  if ((rc = sz_reserve(ctx, ctx->retry_list, (size_t)npairs * 4 + 4))) return rc;
  if ((rc = sz_reserve(ctx, ctx->counters, 128))) return rc;
  a.npairs = npairs;
  a.list = nullptr; a.nlist = 0;
  a.mult = 1;
  a.lds_per_lane = 0;
  a.ws = (char *)ctx->ws.p; a.ws_cap = (i64)ctx->ws.cap;
  a.ovx = (i64 *)ctx->ovx.p; a.ovy = (i64 *)ctx->ovy.p;
  a.ov_cap = (i64)(ctx->ovx.cap < ctx->ovy.cap ? ctx->ovx.cap : ctx->ovy.cap) / 8;
  a.oroff = (int *)ctx->oroff.p; a.or_cap = (i64)ctx->oroff.cap / 4;
  a.ov_base = (i64 *)ctx->ov_base.p; a.or_base = (i64 *)ctx->or_base.p;
  a.nrings = (int *)ctx->nrings.p; a.status = (int *)ctx->status.p;
  a.retry = (int *)ctx->retry_list.p; a.retry_cap = npairs;
  a.counters = (unsigned long long *)ctx->counters.p;
  // bump pointers of the retry region start behind the closed-form region
  // (through pinned words, in stream order: a blocking copy from the stack is staged by the runtime -- the counters'
  // slots: 0 this start, 1 the front pass's read-back, 2 the read-back of a pass; a caller's two batches are a wait apart)
  { const int rs = sz_stage_reserve(ctx, ctx->h_cnt, 3 * 16 * 8); if (rs) return rs; }
  unsigned long long *init = (unsigned long long *)ctx->h_cnt.p;
  for (int k = 0; k < 16; ++k) init[k] = 0;
  init[1] = (unsigned long long)ws_main; init[2] = (unsigned long long)ov_main; init[3] = (unsigned long long)or_main;
  SZ_HIP(ctx, hipMemcpyAsync(ctx->counters.p, init, 16 * 8, hipMemcpyHostToDevice, ctx->stream));
  return SZ_OK;
}

static int clip_lanes(int npairs) {
  if (const char *v = getenv("SUBZERO_CLIP_LANES")) { const int l = atoi(v); if (l >= 1 && l <= 64) return l; }
  const int want = (npairs + 2047) / 2048;
  return want < 1 ? 1 : want > 64 ? 64 : want;
}
// bytes of LDS per lane for the sweep records (0: HBM only)
static int clip_lds_per_lane() {
  if (const char *v = getenv("SUBZERO_CLIP_LDS")) { const int b = atoi(v); if (b >= 0 && b <= 160 * 1024) return b & ~15; }
  return 20 * 1024;
}
static void clip_launch(szContext *ctx, int n, const ClipArgs &a) {
  hipLaunchKernelGGL(k_clip, dim3((n + a.lanes - 1) / a.lanes), dim3(64), (size_t)a.lds_per_lane * a.lanes, ctx->stream, a);
}
// The main pass as a front pass and one sweep pass per slice class (k_clip_front / k_clip_sweep): the sweep launches
// run side by side on streams of their own, forked from and joined to the context's stream.
static int clip_main_in_classes(szContext *ctx, ClipArgs &a) {
  static const int class_bytes[CLIP_CLASSES] = {10 * 1024, 20 * 1024, 40 * 1024, 80 * 1024, 160 * 1024};
  int rc;
  if ((rc = sz_reserve(ctx, ctx->clip_meta, (size_t)a.npairs * 8 + 8))) return rc;
  if ((rc = sz_reserve(ctx, ctx->clip_lists, (size_t)a.npairs * 4 * (CLIP_CLASSES + 1) + 8))) return rc;
  a.meta = (int *)ctx->clip_meta.p;
  a.class_list = (int *)ctx->clip_lists.p;
  if ((rc = sz_side_streams(ctx))) return rc;
  a.lds_per_lane = class_bytes[1];
  hipLaunchKernelGGL(k_clip_front, dim3(a.npairs), dim3(64), (size_t)class_bytes[1], ctx->stream, a);
  SZ_HIP(ctx, hipGetLastError());
  const unsigned long long *c = (const unsigned long long *)ctx->h_cnt.p + 16;
  SZ_HIP(ctx, hipMemcpyAsync((void *)c, a.counters, 16 * 8, hipMemcpyDeviceToHost, ctx->stream));
  SZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  SZ_HIP(ctx, hipEventRecord(ctx->side_fork, ctx->stream));
  // Six launches on four streams (context.hpp: no more than the hardware queues), the longest sweeps first:
  //   side 0: the 160 KB class, then what stays in HBM      side 1: the 80 KB class
  //   side 2: the 40 KB class          the context's stream: the 20 KB class, then the 10 KB class (the many short ones)
  auto launch = [&](hipStream_t st, int k) {
    const int count = (int)c[9 + k];
    if (count == 0) return;
    const int bytes = k < CLIP_CLASSES ? class_bytes[k] : 0;
    const int *list = (const int *)a.class_list + (size_t)k * a.npairs;
    if (k == 0) hipLaunchKernelGGL(k_clip_sweep_small, dim3(count), dim3(64), (size_t)bytes, st, a, list, count, bytes);
    else hipLaunchKernelGGL(k_clip_sweep, dim3(count), dim3(64), (size_t)bytes, st, a, list, count, bytes);
  };
  static const int plan[3][2] = {{4, CLASS_HBM}, {3, -1}, {2, -1}};
  for (int q = 0; q < 3; ++q) {
    SZ_HIP(ctx, hipStreamWaitEvent(ctx->side[q], ctx->side_fork, 0));
    for (int j = 0; j < 2; ++j) if (plan[q][j] >= 0) launch(ctx->side[q], plan[q][j]);
    SZ_HIP(ctx, hipGetLastError());
    SZ_HIP(ctx, hipEventRecord(ctx->side_join[q], ctx->side[q]));
  }
  launch(ctx->stream, 1);
  launch(ctx->stream, 0);
  SZ_HIP(ctx, hipGetLastError());
  for (int q = 0; q < 3; ++q) SZ_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_join[q], 0));
  return SZ_OK;
}

// Runs the main pass and the retry tiers.  On return the per-pair arrays on the device are final.
int szb::run(szContext *ctx, ClipArgs &a, szClipStats *stats) {
  if (a.npairs == 0) {
    if (stats) memset(stats, 0, sizeof(*stats));
    return SZ_OK;
  }
  if (!ctx->clip_lds_set) {   // per context, i.e. per device: hipFuncSetAttribute applies to the current device only
    SZ_HIP(ctx, hipFuncSetAttribute((const void *)k_clip, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SZ_HIP(ctx, hipFuncSetAttribute((const void *)k_clip_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SZ_HIP(ctx, hipFuncSetAttribute((const void *)k_clip_sweep_small, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ctx->clip_lds_set = true;
  }
  // Single rings in fixed point (szClipBatch and what the fracture entries build on it): front pass + sweep classes,
  // one pair per wave.  Everything else (several contours, double input, SUBZERO_CLIP_LANES set): k_clip, one sweep
  // per lane, as few pairs per wave as still give each of the 1024 SIMDs its two waves (252 VGPRs).
  a.lanes = clip_lanes(a.npairs);
  a.lds_per_lane = clip_lds_per_lane();
  static const bool no_team = getenv("SUBZERO_CLIP_NO_TEAM") != nullptr;
  a.no_team = no_team ? 1 : 0;
  const bool classes = a.lds_per_lane > 0 && !getenv("SUBZERO_CLIP_LANES") && !a.aring && !a.fax;
  if (!classes && !getenv("SUBZERO_CLIP_LDS")) a.lds_per_lane = 0;
  if ((i64)a.lds_per_lane * a.lanes > 160 * 1024) a.lds_per_lane = (160 * 1024 / a.lanes) & ~15;
  SZ_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  if (classes) {
    const int rc = clip_main_in_classes(ctx, a);
    if (rc) return rc;
  } else {
    clip_launch(ctx, a.npairs, a);
    SZ_HIP(ctx, hipGetLastError());
  }
  SZ_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  unsigned long long *c = (unsigned long long *)ctx->h_cnt.p + 32;
  SZ_HIP(ctx, hipMemcpyAsync(c, a.counters, 8 * 8, hipMemcpyDeviceToHost, ctx->stream));
  SZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  float ms = 0;
  SZ_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  long long retried = 0;
  DevBuf list2; // second list for ping-pong
  int mult = 1;
  while (c[0] > 0) {
    mult *= 8;
    if (mult > 512) {
      sz_release(list2);
      return sz_fail(ctx, SZ_ERR_WORKSPACE, "%llu pair(s) outgrew the 512x workspace tier", c[0]);
    }
    int nlist = (int)c[0];
    retried += (mult == 8) ? nlist : 0;
    // the list being consumed must not be the one being appended to
    int rc = sz_reserve(ctx, list2, (size_t)nlist * 4 + 4);
    if (rc) return rc;
    SZ_HIP(ctx, hipMemcpyAsync(list2.p, a.retry, (size_t)nlist * 4, hipMemcpyDeviceToDevice, ctx->stream));
    unsigned long long zero = 0;
    SZ_HIP(ctx, hipMemcpy(a.counters, &zero, 8, hipMemcpyHostToDevice));
    ClipArgs b = a;
    b.list = (const int *)list2.p;
    b.nlist = nlist;
    b.mult = mult;
    b.lanes = clip_lanes(nlist);
    b.lds_per_lane = 0;
    clip_launch(ctx, nlist, b);
    SZ_HIP(ctx, hipGetLastError());
    SZ_HIP(ctx, hipMemcpyAsync(c, a.counters, 8 * 8, hipMemcpyDeviceToHost, ctx->stream));
    SZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  sz_release(list2);
  if (c[7] > 0)
    return sz_fail(ctx, SZ_ERR_WORKSPACE, "%llu pair(s) did not fit the retry arena", c[7]);
  if (stats) {
    stats->kernel_ms = ms;
    stats->pairs = a.npairs;
    stats->rings_out = (int64_t)c[4];
    stats->vertices_out = (int64_t)c[5];
    stats->retried = retried;
    stats->failed = (int64_t)c[6];
  }
  return SZ_OK;
}

namespace {

using szb::prepare;
using szb::run;

template <class T>
int upload(szContext *ctx, DevBuf &b, const T *src, size_t n) {
  int rc = sz_reserve(ctx, b, n * sizeof(T) + 16);
  if (rc) return rc;
  if (n) SZ_HIP(ctx, hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  return SZ_OK;
}

// polyclip.m:63-66 for a whole batch: the rings in metres -> Clipper's fixed point, so that a batch of doubles takes the
// front pass and the sweeps of szClipBatch (the same to_fixed the per-access conversion of RingF64 applies)
__global__ void k_rings_to_fixed(const double *fax, const double *fay, const double *fbx, const double *fby, i64 ta, i64 tb,
                                 i64 *ax, i64 *ay, i64 *bx, i64 *by) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ta) { ax[i] = to_fixed(fax[i]); ay[i] = to_fixed(fay[i]); }
  if (i < tb) { bx[i] = to_fixed(fbx[i]); by[i] = to_fixed(fby[i]); }
}

// A small batch (a pair or two per call is how polyclip.m calls the reference's mex): the six input arrays through the
// pinned staging buffer into ONE device block with one copy, instead of six copies the runtime stages one by one.
static int upload_packed(szContext *ctx, const void *const src[6], const size_t bytes[6], const char *dev[6], bool &done) {
  done = false;
  size_t tot = 0;
  for (int k = 0; k < 6; ++k) tot += bytes[k];
  if (tot > ((size_t)1 << 20)) return SZ_OK;
  Packer pk;
  size_t off[6];
  for (int k = 0; k < 6; ++k) off[k] = pk.add(bytes[k] + 16);
  int rc;
  if ((rc = sz_stage_reserve(ctx, ctx->h_in, pk.at)) || (rc = sz_reserve(ctx, ctx->d_in, pk.at))) return rc;
  for (int k = 0; k < 6; ++k) if (bytes[k]) memcpy(ctx->h_in.p + off[k], src[k], bytes[k]);
  SZ_HIP(ctx, hipMemcpyAsync(ctx->d_in.p, ctx->h_in.p, pk.at, hipMemcpyHostToDevice, ctx->stream));
  for (int k = 0; k < 6; ++k) dev[k] = (const char *)ctx->d_in.p + off[k];
  done = true;
  return SZ_OK;
}

inline void from_fixed(i64 v, int64_t &o) { o = v; }
inline void from_fixed(i64 v, double &o) { o = (double)v / 4294967296.0; } // polyclip.m:67

// Gathers the per-pair result slots into the caller's CSR.  T = int64_t (raw) or double (/2^32).
template <class T>
int gather(szContext *ctx, const ClipArgs &a, T *outx, T *outy, int64_t out_cap, int32_t *ring_off,
           int64_t ring_cap, int32_t *pair_ring_off, int32_t *status) {
  const int n = a.npairs;
  // run() has left the counters of its last pass in the context's pinned words (slot 2): how much of the output arrays
  // is in use is known without asking again.  Everything comes back in ONE block of async copies and one wait -- into
  // the pinned staging buffer unless the batch is a large one (there were eight blocking copies to pageable vectors:
  // most of a call with a pair or two, which is how polyclip.m calls the reference's mex)
  const unsigned long long *c = (const unsigned long long *)ctx->h_cnt.p + 32;
  size_t used_v = (size_t)c[2], used_r = (size_t)c[3];
  if (used_v > (size_t)a.ov_cap) used_v = (size_t)a.ov_cap;
  if (used_r > (size_t)a.or_cap) used_r = (size_t)a.or_cap;
  Packer pk;
  const size_t o_vb = pk.add((size_t)n * 8), o_rb = pk.add((size_t)n * 8), o_nr = pk.add((size_t)n * 4), o_st = pk.add((size_t)n * 4),
               o_hx = pk.add(used_v * 8), o_hy = pk.add(used_v * 8), o_hr = pk.add(used_r * 4);
  std::vector<char> pageable;
  char *base;
  if (pk.at <= ((size_t)32 << 20)) { const int rs = sz_stage_reserve(ctx, ctx->h_out, pk.at + 16); if (rs) return rs; base = ctx->h_out.p; }
  else { pageable.resize(pk.at + 16); base = pageable.data(); }
  hipStream_t stq = ctx->stream;
  if (n) {
    SZ_HIP(ctx, hipMemcpyAsync(base + o_vb, a.ov_base, (size_t)n * 8, hipMemcpyDeviceToHost, stq));
    SZ_HIP(ctx, hipMemcpyAsync(base + o_rb, a.or_base, (size_t)n * 8, hipMemcpyDeviceToHost, stq));
    SZ_HIP(ctx, hipMemcpyAsync(base + o_nr, a.nrings, (size_t)n * 4, hipMemcpyDeviceToHost, stq));
    SZ_HIP(ctx, hipMemcpyAsync(base + o_st, a.status, (size_t)n * 4, hipMemcpyDeviceToHost, stq));
  }
  if (used_v) {
    SZ_HIP(ctx, hipMemcpyAsync(base + o_hx, a.ovx, used_v * 8, hipMemcpyDeviceToHost, stq));
    SZ_HIP(ctx, hipMemcpyAsync(base + o_hy, a.ovy, used_v * 8, hipMemcpyDeviceToHost, stq));
  }
  if (used_r) SZ_HIP(ctx, hipMemcpyAsync(base + o_hr, a.oroff, used_r * 4, hipMemcpyDeviceToHost, stq));
  SZ_HIP(ctx, hipStreamSynchronize(stq));
  const i64 *vb = (const i64 *)(base + o_vb), *rb = (const i64 *)(base + o_rb), *hx = (const i64 *)(base + o_hx), *hy = (const i64 *)(base + o_hy);
  const int *nr = (const int *)(base + o_nr), *st = (const int *)(base + o_st), *hr = (const int *)(base + o_hr);
  int64_t nv = 0, nrt = 0;
  pair_ring_off[0] = 0;
  ring_off[0] = 0;
  for (int p = 0; p < n; ++p) {
    status[p] = st[p];
    if (st[p] == SZ_CLIP_OK) {
      const int *ro = hr + rb[p];
      for (int r = 0; r < nr[p]; ++r) {
        int v0 = ro[r], v1 = ro[r + 1];
        if (nrt + 1 > ring_cap || nv + (v1 - v0) > out_cap)
          return sz_fail(ctx, SZ_ERR_CAPACITY, "output buffers too small (need > %lld vertices / %lld rings)",
                         (long long)out_cap, (long long)ring_cap);
        for (int v = v0; v < v1; ++v) {
          from_fixed(hx[vb[p] + v], outx[nv]);
          from_fixed(hy[vb[p] + v], outy[nv]);
          ++nv;
        }
        ++nrt;
        ring_off[nrt] = (int32_t)nv;
      }
    }
    pair_ring_off[p + 1] = (int32_t)nrt;
  }
  return SZ_OK;
}

bool valid_op(int method, int fs, int fc) {
  return method >= 0 && method <= 3 && fs >= 0 && fs <= 3 && fc >= 0 && fc <= 3;
}

} // namespace

extern "C" {

int szClipBatch(szContext *ctx, int npairs, const int64_t *ax, const int64_t *ay, const int32_t *aoff,
                const int64_t *bx, const int64_t *by, const int32_t *boff, int method, int subj_fill,
                int clip_fill, int64_t *outx, int64_t *outy, int64_t out_cap, int32_t *ring_off,
                int64_t ring_cap, int32_t *pair_ring_off, int32_t *status) {
  if (!ctx) return SZ_ERR_BAD_ARG;
  if (npairs < 0 || !aoff || !boff || !ring_off || !pair_ring_off || (npairs && !status))
    return sz_fail(ctx, SZ_ERR_BAD_ARG, "szClipBatch: null argument");
  if (!valid_op(method, subj_fill, clip_fill)) {
    for (int p = 0; p < npairs; ++p) status[p] = SZ_CLIP_BAD_ARG;
    for (int p = 0; p <= npairs; ++p) pair_ring_off[p] = 0;
    ring_off[0] = 0;
    return SZ_OK;
  }
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  i64 ta = aoff[npairs], tb = boff[npairs];
  int rc;
  const void *const src[6] = {ax, ay, aoff, bx, by, boff};
  const size_t bytes[6] = {(size_t)ta * 8, (size_t)ta * 8, ((size_t)npairs + 1) * 4, (size_t)tb * 8, (size_t)tb * 8, ((size_t)npairs + 1) * 4};
  const char *dev[6];
  bool packed = false;
  if ((rc = upload_packed(ctx, src, bytes, dev, packed))) return rc;
  if (!packed) {
    if ((rc = upload(ctx, ctx->in_ax, ax, (size_t)ta))) return rc;
    if ((rc = upload(ctx, ctx->in_ay, ay, (size_t)ta))) return rc;
    if ((rc = upload(ctx, ctx->in_aoff, aoff, (size_t)npairs + 1))) return rc;
    if ((rc = upload(ctx, ctx->in_bx, bx, (size_t)tb))) return rc;
    if ((rc = upload(ctx, ctx->in_by, by, (size_t)tb))) return rc;
    if ((rc = upload(ctx, ctx->in_boff, boff, (size_t)npairs + 1))) return rc;
    dev[0] = (const char *)ctx->in_ax.p; dev[1] = (const char *)ctx->in_ay.p; dev[2] = (const char *)ctx->in_aoff.p;
    dev[3] = (const char *)ctx->in_bx.p; dev[4] = (const char *)ctx->in_by.p; dev[5] = (const char *)ctx->in_boff.p;
  }
  ClipArgs a;
  memset(&a, 0, sizeof(a));
  if ((rc = prepare(ctx, npairs, ta + tb, a))) return rc;
  a.ax = (const i64 *)dev[0]; a.ay = (const i64 *)dev[1]; a.aoff = (const int *)dev[2];
  a.bx = (const i64 *)dev[3]; a.by = (const i64 *)dev[4]; a.boff = (const int *)dev[5];
  a.method = method; a.fill_s = subj_fill; a.fill_c = clip_fill;
  if ((rc = run(ctx, a, nullptr))) return rc;
  return gather<int64_t>(ctx, a, outx, outy, out_cap, ring_off, ring_cap, pair_ring_off, status);
}

// host arrays -> device -> k_clip in its several-contours form -> gather (method may be one of the CLIP_OFFSET_* modes)
static int clip_paths_host(szContext *ctx, int npairs, const int64_t *ax, const int64_t *ay, const int32_t *a_ring_off,
                           const int32_t *a_pair_off, const int64_t *bx, const int64_t *by, const int32_t *b_ring_off,
                           const int32_t *b_pair_off, int method, int subj_fill, int clip_fill, int64_t *outx, int64_t *outy,
                           int64_t out_cap, int32_t *ring_off, int64_t ring_cap, int32_t *pair_ring_off, int32_t *status) {
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  const int nra = a_pair_off[npairs], nrb = b_pair_off[npairs];
  const i64 ta = a_ring_off[nra], tb = b_ring_off[nrb];
  int rc;
  DevBuf d_aring, d_bring;   // (rare call: its two extra index arrays are not kept in the context)
  if ((rc = upload(ctx, ctx->in_ax, ax, (size_t)ta)) || (rc = upload(ctx, ctx->in_ay, ay, (size_t)ta)) ||
      (rc = upload(ctx, ctx->in_aoff, a_pair_off, (size_t)npairs + 1)) || (rc = upload(ctx, ctx->in_bx, bx, (size_t)tb)) ||
      (rc = upload(ctx, ctx->in_by, by, (size_t)tb)) || (rc = upload(ctx, ctx->in_boff, b_pair_off, (size_t)npairs + 1)) ||
      (rc = upload(ctx, d_aring, a_ring_off, (size_t)nra + 1)) || (rc = upload(ctx, d_bring, b_ring_off, (size_t)nrb + 1))) {
    sz_release(d_aring); sz_release(d_bring);
    return rc;
  }
  ClipArgs a;
  memset(&a, 0, sizeof(a));
  if (!(rc = prepare(ctx, npairs, ta + tb, a))) {
    a.ax = (const i64 *)ctx->in_ax.p; a.ay = (const i64 *)ctx->in_ay.p; a.aoff = (const int *)ctx->in_aoff.p;
    a.bx = (const i64 *)ctx->in_bx.p; a.by = (const i64 *)ctx->in_by.p; a.boff = (const int *)ctx->in_boff.p;
    a.aring = (const int *)d_aring.p; a.bring = (const int *)d_bring.p;
    a.method = method; a.fill_s = subj_fill; a.fill_c = clip_fill;
    if (!(rc = run(ctx, a, nullptr)))
      rc = gather<int64_t>(ctx, a, outx, outy, out_cap, ring_off, ring_cap, pair_ring_off, status);
  }
  (void)hipStreamSynchronize(ctx->stream);
  sz_release(d_aring); sz_release(d_bring);
  return rc;
}

int szClipPathsBatch(szContext *ctx, int npairs, const int64_t *ax, const int64_t *ay, const int32_t *a_ring_off,
                     const int32_t *a_pair_off, const int64_t *bx, const int64_t *by, const int32_t *b_ring_off,
                     const int32_t *b_pair_off, int method, int subj_fill, int clip_fill, int64_t *outx, int64_t *outy,
                     int64_t out_cap, int32_t *ring_off, int64_t ring_cap, int32_t *pair_ring_off, int32_t *status) {
  if (!ctx) return SZ_ERR_BAD_ARG;
  if (npairs < 0 || !a_ring_off || !a_pair_off || !b_ring_off || !b_pair_off || !ring_off || !pair_ring_off || (npairs && !status))
    return sz_fail(ctx, SZ_ERR_BAD_ARG, "szClipPathsBatch: null argument");
  if (!valid_op(method, subj_fill, clip_fill)) {
    for (int p = 0; p < npairs; ++p) status[p] = SZ_CLIP_BAD_ARG;
    for (int p = 0; p <= npairs; ++p) pair_ring_off[p] = 0;
    ring_off[0] = 0;
    return SZ_OK;
  }
  // One contour per operand in every pair -- floe against floe, the way floe_interactions.m:29 calls polyclip and
  // polyclip.m:73 the mex -- is szClipBatch's case (the ring CSR is then the pair CSR): front pass with the team form of
  // the clip for short rings, trimming, sweeps with their records in LDS.
  bool single = npairs > 0 && a_pair_off[0] == 0 && b_pair_off[0] == 0 && a_ring_off[0] == 0 && b_ring_off[0] == 0;
  for (int p = 0; single && p < npairs; ++p) single = a_pair_off[p + 1] == p + 1 && b_pair_off[p + 1] == p + 1;
  if (single)
    return szClipBatch(ctx, npairs, ax, ay, a_ring_off, bx, by, b_ring_off, method, subj_fill, clip_fill, outx, outy, out_cap, ring_off, ring_cap,
                       pair_ring_off, status);
  return clip_paths_host(ctx, npairs, ax, ay, a_ring_off, a_pair_off, bx, by, b_ring_off, b_pair_off, method, subj_fill, clip_fill,
                         outx, outy, out_cap, ring_off, ring_cap, pair_ring_off, status);
}

// ---- offset mode of the mex: clipper(P, delta, type [, info]) (mexclipper.cpp:146-203) ---------------------------
// ClipperOffset (clipper.cpp:3786-4218) in two parts.  The raw offset polygons -- one pass over the vertices of each
// contour: unit normals, and per vertex a mitred / squared / rounded join -- are built here on the host with the
// host's libm, as the reference library does (sin, cos, acos, atan2, tan decide integer roundings: the same libm is
// what makes the result identical).  Their clean-up -- a union of heavily self-intersecting polygons, which is where
// the time goes -- is the sweep on the GPU (szv::offset_cleanup).
int szOffsetPaths(szContext *ctx, const int64_t *x, const int64_t *y, const int32_t *ring_off, int nrings, double delta, int join,
                  double info, int64_t *outx, int64_t *outy, int64_t out_cap, int32_t *out_ring_off, int64_t ring_cap,
                  int32_t *nrings_out, int32_t *status) {
  if (!ctx) return SZ_ERR_BAD_ARG;
  if (nrings < 0 || !ring_off || !out_ring_off || !nrings_out || !status || join < 0 || join > 2)
    return sz_fail(ctx, SZ_ERR_BAD_ARG, "szOffsetPaths: bad argument");
  szo::Offsetter co;
  if (info > 0 && join == 1) co.ArcTolerance = info;
  if (info > 0 && join == 2) co.MiterLimit = info;
  for (int r = 0; r < nrings; ++r) co.add_path(x + ring_off[r], y + ring_off[r], ring_off[r + 1] - ring_off[r]);
  co.fix_orientations();
  co.do_offset(delta, join);
  // the clean-up as one "pair" of the batched clip: subject = the raw offset polygons; for an inset the clip operand
  // is a placeholder of four vertices that reserves the workspace of the bounding rectangle the sweep adds
  const bool outset = delta > 0;
  std::vector<int64_t> sx, sy, cx, cy;
  std::vector<int32_t> sro(1, 0), cro(1, 0);
  for (size_t i = 0; i < co.dest.size(); ++i) {
    for (size_t j = 0; j < co.dest[i].size(); ++j) { sx.push_back(co.dest[i][j].X); sy.push_back(co.dest[i][j].Y); }
    sro.push_back((int32_t)sx.size());
  }
  if (!outset) { cx.assign(4, 0); cy.assign(4, 0); cro.push_back(4); }
  int32_t spo[2] = {0, (int32_t)sro.size() - 1}, cpo[2] = {0, (int32_t)cro.size() - 1}, proff[2] = {0, 0};
  sx.push_back(0); sy.push_back(0); cx.push_back(0); cy.push_back(0);   // (never empty pointers)
  const int rc = clip_paths_host(ctx, 1, sx.data(), sy.data(), sro.data(), spo, cx.data(), cy.data(), cro.data(), cpo,
                                 outset ? CLIP_OFFSET_OUTSET : CLIP_OFFSET_INSET, 0, 0, outx, outy, out_cap, out_ring_off, ring_cap, proff, status);
  *nrings_out = proff[1];
  return rc;
}

int szPolyclipBatch(szContext *ctx, int npairs, const double *ax, const double *ay, const int32_t *aoff,
                    const double *bx, const double *by, const int32_t *boff, int method, double *outx,
                    double *outy, int64_t out_cap, int32_t *ring_off, int64_t ring_cap,
                    int32_t *pair_ring_off, int32_t *status) {
  if (!ctx) return SZ_ERR_BAD_ARG;
  if (npairs < 0 || !aoff || !boff || !ring_off || !pair_ring_off || (npairs && !status))
    return sz_fail(ctx, SZ_ERR_BAD_ARG, "szPolyclipBatch: null argument");
  if (!valid_op(method, 0, 0)) {
    for (int p = 0; p < npairs; ++p) status[p] = SZ_CLIP_BAD_ARG;
    for (int p = 0; p <= npairs; ++p) pair_ring_off[p] = 0;
    ring_off[0] = 0;
    return SZ_OK;
  }
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  i64 ta = aoff[npairs], tb = boff[npairs];
  int rc;
  const void *const src[6] = {ax, ay, aoff, bx, by, boff};
  const size_t bytes[6] = {(size_t)ta * 8, (size_t)ta * 8, ((size_t)npairs + 1) * 4, (size_t)tb * 8, (size_t)tb * 8, ((size_t)npairs + 1) * 4};
  const char *dev[6];
  bool packed = false;
  if ((rc = upload_packed(ctx, src, bytes, dev, packed))) return rc;
  if (!packed) {
    if ((rc = upload(ctx, ctx->in_fax, ax, (size_t)ta))) return rc;
    if ((rc = upload(ctx, ctx->in_fay, ay, (size_t)ta))) return rc;
    if ((rc = upload(ctx, ctx->in_aoff, aoff, (size_t)npairs + 1))) return rc;
    if ((rc = upload(ctx, ctx->in_fbx, bx, (size_t)tb))) return rc;
    if ((rc = upload(ctx, ctx->in_fby, by, (size_t)tb))) return rc;
    if ((rc = upload(ctx, ctx->in_boff, boff, (size_t)npairs + 1))) return rc;
    dev[0] = (const char *)ctx->in_fax.p; dev[1] = (const char *)ctx->in_fay.p; dev[2] = (const char *)ctx->in_aoff.p;
    dev[3] = (const char *)ctx->in_fbx.p; dev[4] = (const char *)ctx->in_fby.p; dev[5] = (const char *)ctx->in_boff.p;
  }
  ClipArgs a;
  memset(&a, 0, sizeof(a));
  if ((rc = prepare(ctx, npairs, ta + tb, a))) return rc;
  if ((rc = sz_reserve(ctx, ctx->in_ax, (size_t)ta * 8 + 16)) || (rc = sz_reserve(ctx, ctx->in_ay, (size_t)ta * 8 + 16)) ||
      (rc = sz_reserve(ctx, ctx->in_bx, (size_t)tb * 8 + 16)) || (rc = sz_reserve(ctx, ctx->in_by, (size_t)tb * 8 + 16))) return rc;
  if (ta > 0 || tb > 0) {
    const i64 m = ta > tb ? ta : tb;
    hipLaunchKernelGGL(k_rings_to_fixed, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, (const double *)dev[0], (const double *)dev[1],
                       (const double *)dev[3], (const double *)dev[4], ta, tb, (i64 *)ctx->in_ax.p, (i64 *)ctx->in_ay.p, (i64 *)ctx->in_bx.p, (i64 *)ctx->in_by.p);
    SZ_HIP(ctx, hipGetLastError());
  }
  a.ax = (const i64 *)ctx->in_ax.p; a.ay = (const i64 *)ctx->in_ay.p; a.bx = (const i64 *)ctx->in_bx.p; a.by = (const i64 *)ctx->in_by.p;
  a.aoff = (const int *)dev[2]; a.boff = (const int *)dev[5];
  a.method = method; a.fill_s = SZ_FILL_EVENODD; a.fill_c = SZ_FILL_EVENODD;
  if ((rc = run(ctx, a, nullptr))) return rc;
  return gather<double>(ctx, a, outx, outy, out_cap, ring_off, ring_cap, pair_ring_off, status);
}

int szClipBatchDevice(szContext *ctx, int npairs, const int64_t *d_ax, const int64_t *d_ay,
                      const int32_t *d_aoff, const int64_t *d_bx, const int64_t *d_by,
                      const int32_t *d_boff, int total_a, int total_b, int method, szClipStats *stats) {
  if (!ctx) return SZ_ERR_BAD_ARG;
  if (npairs < 0 || !d_aoff || !d_boff || !valid_op(method, 0, 0))
    return sz_fail(ctx, SZ_ERR_BAD_ARG, "szClipBatchDevice: bad argument");
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  ClipArgs a;
  memset(&a, 0, sizeof(a));
  int rc;
  if ((rc = prepare(ctx, npairs, (i64)total_a + total_b, a))) return rc;
  a.ax = (const i64 *)d_ax; a.ay = (const i64 *)d_ay; a.aoff = (const int *)d_aoff;
  a.bx = (const i64 *)d_bx; a.by = (const i64 *)d_by; a.boff = (const int *)d_boff;
  a.method = method; a.fill_s = SZ_FILL_EVENODD; a.fill_c = SZ_FILL_EVENODD;
  if ((rc = run(ctx, a, stats))) return rc;
  if (stats) stats->input_vertices = (int64_t)total_a + total_b;
  return SZ_OK;
}

int szOrientation(const int64_t *x, const int64_t *y, int n, int *orientation) {
  if (!orientation || (n > 0 && (!x || !y))) return SZ_ERR_BAD_ARG;
  // Area(path) >= 0, clipper.cpp:383-402: a += (X[j] + X[i]) * (Y[j] - Y[i]) in double, area = -a/2
  double a = 0;
  if (n >= 3) {
    for (int i = 0, j = n - 1; i < n; ++i) {
      a += ((double)x[j] + (double)x[i]) * ((double)y[j] - (double)y[i]);
      j = i;
    }
  }
  *orientation = (-a * 0.5 >= 0) ? 1 : 0;
  return SZ_OK;
}

} // extern "C"

#if defined(SZ_CLIP_CLOCKS)
extern "C" int szDebugClipPhases(unsigned long long *out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sz_ph_acc), sizeof(unsigned long long) * 16 * (size_t)(n < (1 << 16) ? n : (1 << 16)));
}
extern "C" int szDebugClipClocks(unsigned long long *out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sz_clip_clocks), sizeof(unsigned long long) * 5 * (size_t)(n < (1 << 16) ? n : (1 << 16)));
}
#endif
