// clip.hip -- batched polygon boolean operations on gfx950: the B1 surface of include/subzero_hip.h
// (replaces /root/reference/private/mexclipper.cpp:204-305 as called from polyclip.m:73).
//
// Mapping to the hardware (irregular integer/FP64 geometry, no dense contraction, so no MFMA):
//  * one lane per ring pair; a wave's 64 lanes run 64 independent sweeps (szv::Sweep, vatti.hpp);
//  * every sweep works in its own workspace slice of one HBM arena; slice and result-slot offsets
//    follow in closed form from the CSR offsets of the input rings (no scan, no allocator);
//  * pairs whose lists outgrow their slice (only heavily self-intersecting input does) are appended to
//    a retry list with one wave-aggregated atomic and re-run at 8x / 64x capacity -- on the GPU.
//    There is no CPU fallback anywhere in this file.
#include <stdlib.h>
#include <algorithm>
#include <cmath>
#include <vector>
#include "clip_batch.hpp"
#include "offset.hpp"
#include "trim.hpp"

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


__global__ void __launch_bounds__(64) k_clip(ClipArgs a) {
  if ((int)threadIdx.x >= a.lanes) return;
  int t = blockIdx.x * a.lanes + threadIdx.x;
  int n = a.list ? a.nlist : a.npairs;
  if (t >= n) return;
  int p = a.list ? a.list[t] : t;
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
    st = szv::clip_rings(a.ws + woff, cap, ra, ta, rb, tb, a.method, a.fill_s, a.fill_c, sink);
  }
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
  if ((rc = sz_reserve(ctx, ctx->counters, 64))) return rc;
  a.npairs = npairs;
  a.list = nullptr; a.nlist = 0;
  a.mult = 1;
  a.ws = (char *)ctx->ws.p; a.ws_cap = (i64)ctx->ws.cap;
  a.ovx = (i64 *)ctx->ovx.p; a.ovy = (i64 *)ctx->ovy.p;
  a.ov_cap = (i64)(ctx->ovx.cap < ctx->ovy.cap ? ctx->ovx.cap : ctx->ovy.cap) / 8;
  a.oroff = (int *)ctx->oroff.p; a.or_cap = (i64)ctx->oroff.cap / 4;
  a.ov_base = (i64 *)ctx->ov_base.p; a.or_base = (i64 *)ctx->or_base.p;
  a.nrings = (int *)ctx->nrings.p; a.status = (int *)ctx->status.p;
  a.retry = (int *)ctx->retry_list.p; a.retry_cap = npairs;
  a.counters = (unsigned long long *)ctx->counters.p;
  // bump pointers of the retry region start behind the closed-form region
  unsigned long long init[8] = {0, (unsigned long long)ws_main, (unsigned long long)ov_main,
                                (unsigned long long)or_main, 0, 0, 0, 0};
  SZ_HIP(ctx, hipMemcpy(ctx->counters.p, init, sizeof(init), hipMemcpyHostToDevice));
  return SZ_OK;
}

static int clip_lanes(int npairs) {
  if (const char *v = getenv("SUBZERO_CLIP_LANES")) { const int l = atoi(v); if (l >= 1 && l <= 64) return l; }
  const int want = (npairs + 2047) / 2048;
  return want < 1 ? 1 : want > 64 ? 64 : want;
}

// Runs the main pass and the retry tiers.  On return the per-pair arrays on the device are final.
int szb::run(szContext *ctx, ClipArgs &a, szClipStats *stats) {
  if (a.npairs == 0) {
    if (stats) memset(stats, 0, sizeof(*stats));
    return SZ_OK;
  }
  // One sweep per lane, and a wave lasts as long as the sum of its lanes' divergent paths: as few pairs per wave as
  // still give each of the 1024 SIMDs its two waves (252 VGPRs) -- a batch of 2e4 pairs runs 10 per wave.
  a.lanes = clip_lanes(a.npairs);
  SZ_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  hipLaunchKernelGGL(k_clip, dim3((a.npairs + a.lanes - 1) / a.lanes), dim3(64), 0, ctx->stream, a);
  SZ_HIP(ctx, hipGetLastError());
  SZ_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  unsigned long long c[8];
  SZ_HIP(ctx, hipMemcpyAsync(c, a.counters, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
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
    hipLaunchKernelGGL(k_clip, dim3((nlist + b.lanes - 1) / b.lanes), dim3(64), 0, ctx->stream, b);
    SZ_HIP(ctx, hipGetLastError());
    SZ_HIP(ctx, hipMemcpyAsync(c, a.counters, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
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

inline void from_fixed(i64 v, int64_t &o) { o = v; }
inline void from_fixed(i64 v, double &o) { o = (double)v / 4294967296.0; } // polyclip.m:67

// Gathers the per-pair result slots into the caller's CSR.  T = int64_t (raw) or double (/2^32).
template <class T>
int gather(szContext *ctx, const ClipArgs &a, T *outx, T *outy, int64_t out_cap, int32_t *ring_off,
           int64_t ring_cap, int32_t *pair_ring_off, int32_t *status) {
  int n = a.npairs;
  std::vector<i64> vb(n), rb(n);
  std::vector<int> nr(n), st(n);
  SZ_HIP(ctx, hipMemcpy(vb.data(), a.ov_base, (size_t)n * 8, hipMemcpyDeviceToHost));
  SZ_HIP(ctx, hipMemcpy(rb.data(), a.or_base, (size_t)n * 8, hipMemcpyDeviceToHost));
  SZ_HIP(ctx, hipMemcpy(nr.data(), a.nrings, (size_t)n * 4, hipMemcpyDeviceToHost));
  SZ_HIP(ctx, hipMemcpy(st.data(), a.status, (size_t)n * 4, hipMemcpyDeviceToHost));
  unsigned long long c[8];
  SZ_HIP(ctx, hipMemcpy(c, a.counters, sizeof(c), hipMemcpyDeviceToHost));
  size_t used_v = (size_t)c[2], used_r = (size_t)c[3];
  if (used_v > (size_t)a.ov_cap) used_v = (size_t)a.ov_cap;
  if (used_r > (size_t)a.or_cap) used_r = (size_t)a.or_cap;
  std::vector<i64> hx(used_v), hy(used_v);
  std::vector<int> hr(used_r);
  if (used_v) {
    SZ_HIP(ctx, hipMemcpy(hx.data(), a.ovx, used_v * 8, hipMemcpyDeviceToHost));
    SZ_HIP(ctx, hipMemcpy(hy.data(), a.ovy, used_v * 8, hipMemcpyDeviceToHost));
  }
  if (used_r) SZ_HIP(ctx, hipMemcpy(hr.data(), a.oroff, used_r * 4, hipMemcpyDeviceToHost));
  int64_t nv = 0, nrt = 0;
  pair_ring_off[0] = 0;
  ring_off[0] = 0;
  for (int p = 0; p < n; ++p) {
    status[p] = st[p];
    if (st[p] == SZ_CLIP_OK) {
      const int *ro = hr.data() + rb[p];
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
  if ((rc = upload(ctx, ctx->in_ax, ax, (size_t)ta))) return rc;
  if ((rc = upload(ctx, ctx->in_ay, ay, (size_t)ta))) return rc;
  if ((rc = upload(ctx, ctx->in_aoff, aoff, (size_t)npairs + 1))) return rc;
  if ((rc = upload(ctx, ctx->in_bx, bx, (size_t)tb))) return rc;
  if ((rc = upload(ctx, ctx->in_by, by, (size_t)tb))) return rc;
  if ((rc = upload(ctx, ctx->in_boff, boff, (size_t)npairs + 1))) return rc;
  ClipArgs a;
  memset(&a, 0, sizeof(a));
  if ((rc = prepare(ctx, npairs, ta + tb, a))) return rc;
  a.ax = (const i64 *)ctx->in_ax.p; a.ay = (const i64 *)ctx->in_ay.p; a.aoff = (const int *)ctx->in_aoff.p;
  a.bx = (const i64 *)ctx->in_bx.p; a.by = (const i64 *)ctx->in_by.p; a.boff = (const int *)ctx->in_boff.p;
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
  if ((rc = upload(ctx, ctx->in_fax, ax, (size_t)ta))) return rc;
  if ((rc = upload(ctx, ctx->in_fay, ay, (size_t)ta))) return rc;
  if ((rc = upload(ctx, ctx->in_aoff, aoff, (size_t)npairs + 1))) return rc;
  if ((rc = upload(ctx, ctx->in_fbx, bx, (size_t)tb))) return rc;
  if ((rc = upload(ctx, ctx->in_fby, by, (size_t)tb))) return rc;
  if ((rc = upload(ctx, ctx->in_boff, boff, (size_t)npairs + 1))) return rc;
  ClipArgs a;
  memset(&a, 0, sizeof(a));
  if ((rc = prepare(ctx, npairs, ta + tb, a))) return rc;
  a.fax = (const double *)ctx->in_fax.p; a.fay = (const double *)ctx->in_fay.p;
  a.fbx = (const double *)ctx->in_fbx.p; a.fby = (const double *)ctx->in_fby.p;
  a.aoff = (const int *)ctx->in_aoff.p; a.boff = (const int *)ctx->in_boff.p;
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
