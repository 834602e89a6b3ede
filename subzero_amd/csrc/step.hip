// step.hip -- the B2 surface of include/subzero_hip.h on gfx950: device-resident Floe struct array and one
// timestep of floe_interactions_all.m:16-285 as a chain of kernels over the bodies in step_core.hpp.
//
// Launch shapes (all 1-D, 256-thread workgroups unless noted; the work is irregular FP64/int64 geometry,
// so the rules that matter are occupancy, coalesced SoA loads and short dependent chains -- no MFMA):
//   k_ext_clear (+ init)                                      one thread per (extended) floe; ghosts are made inside their scan
//   k_scan_reduce / k_scan_write                              exclusive scans, two launches of <= 256 workgroups
//   k_cell_* / k_broad_scan / k_broad_filter                  uniform grid; thread per (floe, neighbour cell); thread per floe
//   k_pair_expand                                             one thread per extended floe (the counts come from k_broad_filter)
//   k_pair_classify / k_narrow_team (narrow_team.hip)         thread per pair: boxes apart -> answered, else onto its kernel's list;
//                                                             a team of 16 (32) lanes per pair, four (two) pairs per wave
//   k_narrow_convex / k_narrow / k_narrow_lds                 one lane per pair: the single-minimum fast path for what the team pass
//                                                             hands on, the general sweep for what that defers
//   k_incoming / k_row_count / k_row_fill                     per pair / per extended floe (the transfer marks ride along)
//   k_integrate_pre / k_integrate_post                        one thread per real floe
//   k_forcing                                                 one 256-thread workgroup per real floe (Monte-Carlo forcing)
// Host read-backs per step: one, of the counters at its end (k_publish into a pinned host buffer), in a steady step --
// the front stages and the narrow phase are sized from the step before, the back half runs behind a device-side flag
// (C_GO); a step whose capacities do not fit, or whose narrow phase defers pairs to the general sweep, falls back to
// the read-backs in the middle (pair count, deferred pairs).  Buffers are grow-only, owned by the context.  No CPU
// fallback exists: every failure is an error code.
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#if defined(SZ_HOST_CLOCKS)
// diagnostic build (make diag_host): where the HOST's time of a step goes (one context per process in that build)
static struct { std::chrono::steady_clock::time_point entry, ret; bool have = false; int n = 0; double first = 0, queue = 0, wait = 0, tail = 0, between = 0; } g_hc;
#endif
#include "context.hpp"
#include "step_core.hpp"
#include "step_state.hpp"

#include "narrow_general.hpp"

namespace {

// ---------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------
__device__ inline int dev_next(const Work &W) { return (int)W.counters[C_NEXT]; }

__global__ void k_counters_clear(Floes F, Work W) {
  const int e = threadIdx.x * C_STRIDE;
  if (e < C_COUNT && e != C_SCR_BUMP && e != C_VETO) W.counters[e] = (e == C_NEXT) ? (unsigned long long)F.n : 0ULL;
}
__global__ void k_ext_clear(Floes F, Params P, Work W) {   // (and the cells of the broad-phase grid: one launch fewer)
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  for (int c = e; c <= W.lvl_base[W.n_levels]; c += gridDim.x * blockDim.x) { W.cell_cnt[c] = 0; W.cell_fill[c] = 0; }
  if (e < W.cap_ext) { ext_clear(e, W); marks_clear(e, F); }
  if (e < F.n) ext_init(e, F, W);
  if (e >= F.nb && e < F.n) mc_precheck(e, F, P, W);
}
// ---- exclusive scans: two launches of <= 256 workgroups (one per CU), no spinning --------------------------
// (one launch with a look-back wait between the halves was measured in round 3: 2 us SLOWER per scan than the kernel
// boundary it removes -- a back-to-back launch costs about 2 us here, a cross-XCD hand-off of a flag more)
// k_scan_reduce: workgroup b sums its tile of the input -> agg[b]; k_scan_write: workgroup b adds up the
// aggregates before it, scans its tile (coalesced 1024-element slabs, wave shuffles + 16 LDS words) and writes;
// the last workgroup stores the total.  A Job names the arrays: n(), first(ch, k) (the value, read or produced in
// the first launch), load(ch, k) (read again in the second), store(ch, k, v) (may act on the element right there),
// finish(totals), CH channels side by side.
__device__ inline long long wave_scan_incl(long long v) {
  const int lane = threadIdx.x & 63;
  for (int d = 1; d < 64; d <<= 1) {
    long long u = __shfl_up(v, d);
    if (lane >= d) v += u;
  }
  return v;
}
// inclusive scan over the 1024 threads of the workgroup; *total = sum over the workgroup
__device__ inline long long block_scan_incl(long long v, long long *lds16, long long *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long inc = wave_scan_incl(v);
  __syncthreads();                       // lds16 may still be read from the previous slab
  if (lane == 63) lds16[wave] = inc;
  __syncthreads();
  long long before = 0, all = 0;
  for (int w = 0; w < 16; ++w) { long long x = lds16[w]; if (w < wave) before += x; all += x; }
  *total = all;
  return inc + before;
}
template <class Job>
__global__ void __launch_bounds__(1024) k_scan_reduce(Job job, int tile, long long *agg) {
  __shared__ long long lds16[16];
  const int n = job.n();
  const long long lo = (long long)blockIdx.x * tile;
  for (int ch = 0; ch < Job::CH; ++ch) {
    long long s = 0;
    for (long long k = lo + threadIdx.x; k < lo + tile && k < n; k += 1024) s += job.first(ch, (int)k);
    long long tot;
    (void)block_scan_incl(s, lds16, &tot);
    if (threadIdx.x == 0) agg[ch * 256 + blockIdx.x] = tot;
  }
}
template <class Job>
__global__ void __launch_bounds__(1024) k_scan_write(Job job, int tile, const long long *agg) {
  __shared__ long long lds16[16];
  const int n = job.n();
  const long long lo = (long long)blockIdx.x * tile;
  long long totals[Job::CH];
  long long all[Job::CH];          // the sums over ALL tiles, for a job whose store needs them (ScanGhosts)
  if (Job::NEEDS_TOTALS)
    for (int ch = 0; ch < Job::CH; ++ch) {
      const long long one = ((int)threadIdx.x < (int)gridDim.x) ? agg[ch * 256 + threadIdx.x] : 0;
      (void)block_scan_incl(one, lds16, &all[ch]);
    }
  for (int ch = 0; ch < Job::CH; ++ch) {
    long long mine = ((int)threadIdx.x < (int)blockIdx.x) ? agg[ch * 256 + threadIdx.x] : 0, carry;
    (void)block_scan_incl(mine, lds16, &carry);
    for (long long s0 = lo; s0 < lo + tile && s0 < n; s0 += 1024) {
      const long long k = s0 + threadIdx.x;
      const long long v = (k < n) ? job.load(ch, (int)k) : 0;
      long long tot;
      const long long inc = block_scan_incl(v, lds16, &tot);
      if (k < n) { if (Job::NEEDS_TOTALS) job.store_with_totals(ch, (int)k, carry + inc - v, all); else job.store(ch, (int)k, carry + inc - v); }
      carry += tot;
    }
    totals[ch] = carry;
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) job.finish(totals);
}
struct ScanGhost { // ghost creation of one dimension fused into its scan: flags in the first launch, ghosts in the second
  enum { CH = 1, NEEDS_TOTALS = 0 };
  __device__ void store_with_totals(int, int, long long, const long long *) const {}
  int dim; Floes F; Params P; Work W;
  __device__ int n() const { return (dim == 0) ? F.n : F.n + (int)W.counters[C_GX]; }
  __device__ long long first(int, int k) const { Work w = W; ghost_flag(k, dim, F, P, w); return W.flag[k]; }
  __device__ long long load(int, int k) const { return W.flag[k]; }
  __device__ void store(int, int k, long long v) const { Work w = W; W.scan[k] = (int)v; ghost_emit(k, dim, n(), F, P, w); }
  __device__ void finish(const long long *t) const {
    W.counters[dim == 0 ? C_GX : C_GY] = (unsigned long long)t[0];
    W.counters[C_NEXT] = (unsigned long long)(n() + t[0]);
  }
};
// Both periodic dimensions in one scan (the reference makes the x ghosts of every floe, then the y ghosts of every floe AND
// x ghost, ghost_images.m:33-37,54-58): the y test of an x ghost is its parent's (same ring, same y), so the three
// prefix sums over the real floes -- x ghosts, y ghosts, both -- give every slot: x ghost of k at N + px, y ghost of k at
// N + GX + py, y ghost of k's x ghost at N + GX + GYreal + pxy (the order the two passes produce).  Two launches instead
// of four, over n entries instead of n + 2n.
struct ScanGhosts {
  enum { CH = 3, NEEDS_TOTALS = 1 };
  Floes F; Params P; Work W;
  __device__ int n() const { return F.n; }
  __device__ long long first(int ch, int k) const {
    if (ch == 0) {
      Work w = W;
      ghost_flag(k, 0, F, P, w); const int fx = W.flag[k];
      ghost_flag(k, 1, F, P, w); const int fy = W.flag[k];
      W.flag[k] = fx | (fy << 1);
    }
    return load(ch, k);
  }
  __device__ long long load(int ch, int k) const { const int f = W.flag[k]; return ch == 0 ? (f & 1) : ch == 1 ? ((f >> 1) & 1) : ((f & 3) == 3); }
  __device__ void store(int, int, long long) const {}
  __device__ void entry(int g, int parent, int k, bool sx, bool sy) const {
    if (g >= W.cap_ext) return; // cannot happen: cap_ext = 4 n
    W.ext_src[g] = W.ext_src[k]; W.ext_nv[g] = W.ext_nv[k];
    W.ext_x[g] = sx ? W.ext_x[k] - 2 * P.Lx * sgn(W.ext_x[k]) : W.ext_x[k];
    W.ext_y[g] = sy ? W.ext_y[k] - 2 * P.Ly * sgn(W.ext_y[k]) : W.ext_y[k];
    W.ext_num[g] = -iabs(W.ext_num[k]);
    W.ext_parent[g] = parent;
  }
  __device__ void store_with_totals(int ch, int k, long long v, const long long *all) const {
    if (ch == 0) { W.scan[k] = (int)v; return; }
    if (ch == 1) { W.yg_of[k] = (int)v; return; }          // (kept there until the last channel comes by)
    const int f = W.flag[k], N = F.n, GX = (int)all[0], GYr = (int)all[1];
    const int px = W.scan[k], py = W.yg_of[k];
    W.yg_of[k] = -1;
    if (f & 1) { const int g = N + px; entry(g, k, k, true, false); W.xg_of[k] = g; }
    if (f & 2) { const int g = N + GX + py; entry(g, k, k, false, true); W.yg_of[k] = g; }
    if ((f & 3) == 3) { const int gx = N + px, g = N + GX + GYr + (int)v; entry(g, gx, k, true, true); if (gx < W.cap_ext) W.yg_of[gx] = g; }
  }
  __device__ void finish(const long long *t) const {
    W.counters[C_GX] = (unsigned long long)t[0];
    W.counters[C_GY] = (unsigned long long)(t[1] + t[2]);
    W.counters[C_NEXT] = (unsigned long long)(F.n + t[0] + t[1] + t[2]);
  }
};
struct ScanPairs { // pair counts -> pair_base, pair vertex counts -> nv_base
  enum { CH = 2, NEEDS_TOTALS = 0 };
  __device__ void store_with_totals(int, int, long long, const long long *) const {}
  Work W;
  __device__ int n() const { return dev_next(W); }
  __device__ long long first(int ch, int k) const { if (ch == 0) { Work w = W; pair_room(k, w); } return load(ch, k); }
  __device__ long long load(int ch, int k) const { return ch == 0 ? (long long)W.pair_cnt[k] : (long long)W.pair_nv[k]; }
  __device__ void store(int ch, int k, long long v) const { if (ch == 0) W.pair_base[k] = v; else W.nv_base[k] = v; }
  __device__ void finish(const long long *t) const {
    const int N = n();
    W.counters[C_NPAIRS] = (unsigned long long)t[0]; W.pair_base[N] = t[0];
    W.counters[C_NVTOT] = (unsigned long long)t[1]; W.nv_base[N] = t[1];
  }
};
struct ScanRows {
  enum { CH = 1, NEEDS_TOTALS = 0 };
  __device__ void store_with_totals(int, int, long long, const long long *) const {}
  Work W;
  __device__ int n() const { return dev_next(W); }
  __device__ long long first(int ch, int k) const { return load(ch, k); }
  __device__ long long load(int, int k) const { return W.row_cnt[k]; }
  __device__ void store(int, int k, long long v) const { W.row_off[k] = v; }
  __device__ void finish(const long long *t) const { W.counters[C_NROWS] = (unsigned long long)t[0]; W.row_off[n()] = t[0]; }
};
struct ScanCells {
  enum { CH = 1, NEEDS_TOTALS = 0 };
  __device__ void store_with_totals(int, int, long long, const long long *) const {}
  Work W;
  __device__ int n() const { return W.lvl_base[W.n_levels]; }
  __device__ long long first(int ch, int k) const { return load(ch, k); }
  __device__ long long load(int, int k) const { return W.cell_cnt[k]; }
  __device__ void store(int, int k, long long v) const { W.cell_off[k] = (int)v; }
  __device__ void finish(const long long *t) const { W.cell_off[n()] = (int)t[0]; }
};
struct ScanSegment {
  enum { CH = 1, NEEDS_TOTALS = 0 };
  __device__ void store_with_totals(int, int, long long, const long long *) const {}
  const int *in; i64 *out; int count; unsigned long long *total;
  __device__ int n() const { return count; }
  __device__ long long first(int ch, int k) const { return load(ch, k); }
  __device__ long long load(int, int k) const { return in[k]; }
  __device__ void store(int, int k, long long v) const { out[k] = v; }
  __device__ void finish(const long long *t) const { if (total) *total = (unsigned long long)t[0]; }
};

__global__ void k_broad(int lo_is_real, Floes F, Params P, Work W) {
  int N = dev_next(W);
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (lo_is_real) { i += F.nb; if (i >= F.n) return; }
  else { i += F.n; if (i >= N) return; }
  broad(i, N, F, P, W);
}
// The records and staged rings of the entries (step_core.hpp ext_record) with 16 lanes per entry: lane l forms vertex l
// of the ring where the entry lies and word l of the record, so that the 128-byte lines the team pass loads are written
// whole; the bounding box is a minimum / maximum over the lanes (exact whatever the order), the same values as the
// one-thread form.
__global__ void __launch_bounds__(256) k_ext_records(Floes F, Work W) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, e = t >> 4, l = t & 15;
  if (e >= dev_next(W)) return;                         // (whole groups of 16 lanes leave together)
  const int s = W.ext_src[e];
  const int v0 = F.vert_off[s], n = F.vert_off[s + 1] - v0;
  const double ox = W.ext_x[e], oy = W.ext_y[e];
  double x0 = SZC_INF, x1 = -SZC_INF, y0 = SZC_INF, y1 = -SZC_INF;
  for (int k = l; k < n; k += 16) {
    const double X = F.cax[v0 + k] + ox, Y = F.cay[v0 + k] + oy;
    x0 = szc::mind(x0, X); x1 = szc::maxd(x1, X); y0 = szc::mind(y0, Y); y1 = szc::maxd(y1, Y);
  }
  for (int d = 1; d < 16; d <<= 1) {
    x0 = szc::mind(x0, __shfl_xor(x0, d, 16)); x1 = szc::maxd(x1, __shfl_xor(x1, d, 16));
    y0 = szc::mind(y0, __shfl_xor(y0, d, 16)); y1 = szc::maxd(y1, __shfl_xor(y1, d, 16));
  }
  int nstage = n;
  if (n == (int)RING_SLOTS + 1) {
    const double fx = F.cax[v0] + ox, fy = F.cay[v0] + oy, lx = F.cax[v0 + n - 1] + ox, ly = F.cay[v0 + n - 1] + oy;
    const double px = F.cax[v0 + n - 2] + ox, py = F.cay[v0 + n - 2] + oy;
    if (lx == fx && ly == fy && szc::hyp(fx - px, fy - py) > 1) --nstage;
  }
  if (nstage > (int)RING_SLOTS || nstage < 1) nstage = -1;
  const double safe = SZX_FIXED_SAFE_METRES - 2;
  bool ok = true;
  double X = 0, Y = 0;
  if (l < nstage) { X = F.cax[v0 + l] + ox; Y = F.cay[v0 + l] + oy; ok = szc::absd(X) < safe && szc::absd(Y) < safe; }
  const unsigned long long bad = __ballot(!ok);
  const bool in_range = ((bad >> ((threadIdx.x & 63) & ~15)) & 0xffffULL) == 0;
  double *rx = W.ext_ring + 2 * (size_t)RING_SLOTS * e;
  if (l < nstage) { rx[l] = X; rx[RING_SLOTS + l] = Y; }
  double w = 0;
  switch (l) {
    case REC_X: w = ox; break; case REC_Y: w = oy; break; case REC_UI: w = F.Ui[s]; break; case REC_VI: w = F.Vi[s]; break;
    case REC_KSI: w = F.ksi_ice[s]; break; case REC_H: w = F.h[s]; break; case REC_AREA: w = F.area[s]; break;
    case REC_BX0: w = x0; break; case REC_BX1: w = x1; break; case REC_BY0: w = y0; break; case REC_BY1: w = y1; break;
    case REC_NSTAGE: w = (double)nstage; break; case REC_NSTORED: w = (double)n; break; case REC_INRANGE: w = in_range ? 1.0 : 0.0; break;
    default: break;
  }
  W.ext_rec[(size_t)REC_WORDS * e + l] = w;
  if (l == 0) cell_count(e, F, W);      // (the entry's cell of the broad-phase grid: one launch fewer than as a kernel of its own)
}
__global__ void k_cell_fill(Floes F, Work W) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < dev_next(W)) cell_fill(e, F, W);
}
__global__ void k_broad_scan(int skip, Floes F, Params P, Work W) { // thread (i, c9): 9 consecutive threads share floe i; starts `skip` floes after the first
  int N = dev_next(W);
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int i = (int)(t / 9) + F.nb + skip, c9 = (int)(t % 9);
  if (i < N) broad_scan(i, c9, N, F, P, W);
}
// The candidate list of a REAL floe in one go -- k_broad_scan + k_broad_filter + pair_count with a wave per floe instead
// of a thread (one-level grids and lists of up to 64 entries; the thread-per-floe kernels stay for everything else and
// for the ghosts, whose lists start from their root's finished list).  Lanes test the items of the 3 x 3 cells side by
// side, the hits are ranked by index with shuffles (ascending j), the `mems` rule of :93-113 is a duplicate test --
// the first floe with a given FloeNum is always kept, so a later one is "already listed" exactly when an earlier
// entry carries the same number -- and the pairs and their vertex total are a ballot and a wave sum.  Same lists, same
// counts as the thread-per-floe kernels (every parity test runs through this one).
__global__ void __launch_bounds__(64) k_broad_wave(Floes F, Params P, Work W) {
  __shared__ int hit[64], sorted[64], s_end[9], s_c0[9], s_cn[9];
  const int i = F.nb + blockIdx.x, lane = threadIdx.x;
  if (i >= F.n) return;
  const int N = dev_next(W);
  const int si = W.ext_src[i];
  const double xi = W.ext_x[i], yi = W.ext_y[i], ri = F.rmax[si];
  int H = 0;
  if (F.alive[si] && !(xi != xi) && P.COLLISION && is_present(si, W)) {
    const int gx0 = cell_coord(xi, W.lvl_x0[0], W.lvl_cs[0], W.lvl_cx[0]), gy0 = cell_coord(yi, W.lvl_y0[0], W.lvl_cs[0], W.lvl_cy[0]);
    // lanes 0..8: extent of one neighbour cell; exclusive prefix over the nine
    int c0 = 0, cn = 0;
    if (lane < 9) {
      const int gx = gx0 + (lane % 3) - 1, gy = gy0 + (lane / 3) - 1;
      if (gx >= 0 && gx < W.lvl_cx[0] && gy >= 0 && gy < W.lvl_cy[0]) {
        const int c = W.lvl_base[0] + gy * W.lvl_cx[0] + gx;
        c0 = W.cell_off[c]; cn = W.cell_off[c + 1] - c0;
      }
    }
    int pre = cn;                                     // inclusive scan over lanes 0..8 (zeros elsewhere)
    for (int d = 1; d < 16; d <<= 1) { const int v = __shfl_up(pre, d); if (lane >= d) pre += v; }
    if (lane < 9) { s_end[lane] = pre; s_c0[lane] = c0; s_cn[lane] = cn; }
    __syncthreads();
    const int T = s_end[8];
    for (int base = 0; base < T; base += 64) {
      const int t = base + lane;
      bool ok = false;
      int j = 0;
      if (t < T) {
        int cell = 0;                                 // the cell whose range holds item t
        for (int c = 0; c < 8; ++c) if (t >= s_end[c]) cell = c + 1;
        const int q = s_c0[cell] + (t - (s_end[cell] - s_cn[cell]));
        const CellItem it = W.cell_item[q];
        j = it.e;
        if (j > i) {
          const double dx = xi - it.x, dy = yi - it.y, R = ri + it.r, d2 = dx * dx + dy * dy;
          ok = !(d2 > R * R * 1.000001) && (::sqrt(d2) < R);       // :103
        }
      }
      const unsigned long long m = __ballot(ok);
      if (ok) { const int r = H + __popcll(m & ((1ULL << lane) - 1ULL)); if (r < 64) hit[r] = j; }
      H += __popcll(m);
    }
  }
  __syncthreads();
  if (H > W.pot_cap || H > 64) {                      // the list does not fit: the host grows the lists and runs the front again
    if (lane == 0) { atomicAdd(&W.counters[C_ERR_POT], 1ULL); atomicMax(&W.counters[C_NEED_POT], (unsigned long long)H); }
    H = 0;                                            // (nothing of this step's lists is used after such a report)
  }
  // ascending j: rank = how many hits are smaller
  const int mine = lane < H ? hit[lane] : 0x7fffffff;
  int rank = 0;
  for (int k = 0; k < H; ++k) rank += (hit[k] < mine) ? 1 : 0;
  if (lane < H) sorted[rank] = mine;
  __syncthreads();
  // `mems` (:93-99,113): an entry whose FloeNum an earlier entry carries stays only if the pair spans the domain
  const int j = lane < H ? sorted[lane] : 0;
  const int num = lane < H ? W.ext_num[j] : 0;       // FloeNum of the entry: negative for a ghost (FloeNums are >= 1)
  const int key = iabs(num);
  bool keep = lane < H;
  if (P.PERIODIC) {
    bool dup = false;                                // an earlier entry IS the floe this one is an image of
    for (int k = 0; k < H; ++k) { const int nk = __shfl(num, k); if (k < lane && lane < H && nk == key) dup = true; }
    if (keep && dup) {
      const double lim = szc::mind(2 * P.Lx, 2 * P.Ly), R = ri + F.rmax[W.ext_src[j]];
      keep = 2 * R > lim;
    }
  }
  const unsigned long long km = __ballot(keep);
  const int K = __popcll(km);
  if (keep) W.pot_list[(size_t)i * W.pot_cap + __popcll(km & ((1ULL << lane) - 1ULL))] = j + 1;
  // pairs of this floe and their vertex total (pair_count)
  const bool rel = keep && pair_relevant(i, j, W);
  const int ni = W.ext_nv[i];
  long long nv = rel ? (long long)(ni + W.ext_nv[j]) : 0;
  if (rel) (void)__hip_atomic_fetch_add(&W.inc_pre[j], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int d = 32; d > 0; d >>= 1) nv += __shfl_xor(nv, d);
  int cnt = __popcll(__ballot(rel));
  if (lane == 0) {
    if (has_wall_pair(i, N, F, P, W)) { ++cnt; nv += ni + 4; }
    W.pot_cnt[i] = K; W.pair_cnt[i] = cnt; W.pair_nv[i] = nv;
  }
}
// 64 floes per workgroup; each thread sorts and filters its candidate list in its own slice of LDS
// (2 * pot_cap + 1 ints: the odd stride spreads the threads over the banks), or of `spill` when that does not fit
// (`lanes` of the 64 threads work: as many as have room for their slice in 64 KB of LDS at the current capacity)
__global__ void __launch_bounds__(64) k_broad_filter(int lo_is_real, int lanes, Floes F, Params P, Work W, int *spill) {
  extern __shared__ int filter_lds[];
  int N = dev_next(W);
  if ((int)threadIdx.x >= lanes) return;
  int i = blockIdx.x * lanes + threadIdx.x;
  const size_t stride = 2 * (size_t)W.pot_cap + 1;
  int *tmp = spill ? spill + ((size_t)blockIdx.x * lanes + threadIdx.x) * stride : filter_lds + threadIdx.x * stride;
  if (lo_is_real) { i += F.nb; if (i >= F.n) return; }
  else { i += F.n; if (i >= N) return; }
  broad_filter(i, F, P, W, tmp);
  pair_count(i, N, F, P, W);   // the list of i is final: its pairs and their vertex total
}
__global__ void k_pair_expand(Floes F, Params P, Work W, unsigned long long scratch_bump) {
  int N = dev_next(W);
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) W.counters[C_SCR_BUMP] = scratch_bump;   // where the retry tiers of the narrow phase start to allocate
  if (i < N) pair_expand(i, N, F, P, W);
}
// The step's counters for the host, written straight into its pinned buffer (a launch back to back with the kernel
// before it; a hipMemcpyAsync of 150 bytes costs a copy-engine start of ~6 us in the middle of the step)
// (reset_next >= 0: the counters are then set up for the next step, with C_NEXT = reset_next, so that it need not
// start with a launch that clears them)
// (go: what the back half of the step -- contact rows, calc_trajectory -- is to do, see GoRule.  A back half queued
// behind the narrow phase without a host round trip runs only if the narrow phase left nothing for the host to do;
// its kernels return at once otherwise, and so does the reset of the counters.)
// what k_publish will make of C_GO, for a kernel that runs in its place (k_incoming_go): no side effect
// (pair_rows: the step's pair rows -- C_PAIRROWS, plus the 64 parts the fused chain keeps them in: pair_rows_total)
__device__ inline unsigned long long pair_rows_total(const unsigned long long *counters, const unsigned long long *parts) {   // by a whole wave
  unsigned long long v = parts ? parts[(size_t)AGG_STRIDE * (threadIdx.x & 63)] : 0ULL;
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  return v + counters[C_PAIRROWS];
}
// (by a whole wave: lane k looks at counter k -- one round of loads, where `a || b || c ...` over the counters compiles to
// a chain of scalar loads, each waiting for the one before: 5 us at the head of the back half, measured)
// (k, v: this lane's counter -- word offset and value; GO_LEAVE is the caller's: it needs C_GO itself)
__device__ inline bool go_by_values(int k, unsigned long long v, const GoRule &rule, unsigned long long pair_rows) {
  if (rule.mode == GO_SET) return true;
  bool bad = false;
  switch (k) {
    case C_NEED_MC: case C_ERR_POT: case C_ERR_ARENA: case C_ERR_CLIP: case C_RETRY: case C_BIG: case C_ERR_CELL: case C_VETO: bad = v != 0; break;
    case C_NPAIRS: bad = (long long)v > rule.spec_np; break;
    case C_NVTOT: bad = (long long)v > rule.spec_nvt; break;
    case C_TEAM: bad = rule.team_pending && v != 0; break;
    case C_NEXT: bad = rule.next_cap > 0 && (long long)v > rule.next_cap; break;
    default: break;
  }
  return __ballot(bad) == 0ULL && !(2 * (long long)pair_rows + 1 > rule.rows_cap);
}
__device__ inline bool go_by_rule(const unsigned long long *counters, const GoRule &rule, unsigned long long pair_rows) {
  if (rule.mode == GO_SET) return true;
  if (rule.mode != GO_IF_CLEAN) return counters[C_GO] != 0;
  const int k = (int)(threadIdx.x & 63) * C_STRIDE;
  const unsigned long long v = k < (int)C_COUNT ? counters[k] : 0ULL;
  return go_by_values(k, v, rule, pair_rows);      // (one statement of the rule: ADVICE r3)
}
// (seq != 0: the last word of the host's copy -- host_copy[C_COUNT] -- is set to seq once the others are out: the host
// spins on it instead of sleeping in hipStreamSynchronize, whose wake-up costs some tens of microseconds per step)
__global__ void k_publish(unsigned long long *counters, unsigned long long *host_copy, long long reset_next, GoRule rule, int reset_only_if_go, const unsigned long long *parts, unsigned long long seq = 0) {
  const unsigned long long prows = pair_rows_total(counters, parts);
  const unsigned long long go = go_by_rule(counters, rule, prows) ? 1ULL : 0ULL;
  const unsigned e = threadIdx.x * C_STRIDE;
  if (e < (unsigned)C_COUNT) {
    const unsigned long long v = (e == C_GO) ? go : (e == C_PAIRROWS) ? prows : counters[e];
    host_copy[e] = v;
    const bool reset = reset_next >= 0 && (!reset_only_if_go || go);
    if (e == C_AHEAD) counters[e] = go;                 // (the step's last publish: did it run to its end? -- kf_front_a / b queued ahead)
    else if (reset) { if (e != C_SCR_BUMP && e != C_VETO) counters[e] = (e == C_NEXT) ? (unsigned long long)reset_next : 0ULL; }
    else if (e == C_GO) counters[e] = go;
  }
  if (seq) {
    __threadfence_system();                             // (every lane's words are out before the flag)
    if (threadIdx.x == 0) __hip_atomic_store(&host_copy[C_COUNT], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
static_assert(C_SLOTS <= 64, "k_publish is launched with one wave");
__global__ void k_export_header(char *base, i64 ns, const unsigned long long *rows, i64 status, i64 rows_max) {
  seg_header(base, ns, status ? 0 : (i64)*rows, status, rows_max);
}
__global__ void k_export(i64 ns, i64 p_lo, const i64 *off, Floes F, Work W, Xfer X) {
  i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < ns) export_pair(k, p_lo, off, F, W, X);
}
__global__ void k_import(i64 np, int world, i64 seg_max, i64 rows_max, char *recv, Work W) {
  i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < np) import_pair(p, np, world, seg_max, rows_max, recv, W);
}
// k_publish (the counters after the narrow phase, C_GO by the rule) and k_incoming in one launch, for the step whose
// back half is queued behind its narrow phase: every workgroup works the rule out for itself from the counters -- which
// nothing changes meanwhile but the first workgroup's C_GO, and that one they do not read -- and the first one publishes
__global__ void __launch_bounds__(256) k_incoming_go(i64 np, Work W, unsigned long long *host_copy, GoRule rule) {
  const bool go = go_by_rule(W.counters, rule, W.counters[C_PAIRROWS]);
  const i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (go && p < np && p < (i64)W.counters[C_NPAIRS]) incoming(p, W);
  if (blockIdx.x == 0 && threadIdx.x < (unsigned)C_SLOTS) {
    const unsigned e = threadIdx.x * C_STRIDE;
    host_copy[e] = (e == C_GO) ? (go ? 1ULL : 0ULL) : W.counters[e];
    if (e == C_GO) W.counters[C_GO] = go ? 1ULL : 0ULL;
  }
}
__global__ void k_incoming(i64 np, Work W) {   // (np may be a capacity: the pairs are those of C_NPAIRS)
  if (!W.counters[C_GO]) return;
  i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < np && p < (i64)W.counters[C_NPAIRS]) incoming(p, W);
}
// (agg: the fused chain -- the count also goes into its tile's sum, from which kf_row_fill places the rows)
__global__ void __launch_bounds__(64) k_row_count(int lanes, Work W, int *spill, int agg) { // slices of LDS (or of `spill`) as in k_broad_filter
  extern __shared__ int filter_lds[];
  if (!W.counters[C_GO]) return;
  int N = dev_next(W);
  if ((int)threadIdx.x >= lanes) return;
  int i = blockIdx.x * lanes + threadIdx.x;
  const size_t stride = 2 * (size_t)W.pot_cap + 1;
  int *tmp = spill ? spill + ((size_t)blockIdx.x * lanes + threadIdx.x) * stride : filter_lds + threadIdx.x * stride;
  if (i < N) {
    row_count(i, W, tmp);
    if (agg) { const int c = W.row_cnt[i]; if (c) atomicAdd(&W.row_agg[i >> 6], (unsigned long long)c); }
  }
}
__global__ void k_row_fill(Floes F, Params P, Work W) {
  if (!W.counters[C_GO]) return;
  int N = dev_next(W);
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  // (the kill mark of floe i is its own row_fill's: the first half of the transfer resolution rides along; the second
  // half, which needs every floe's first half, rides with k_integrate_pre -- two launches fewer)
  if (i < N) { row_fill(i, N, F, P, W); transfer_collect(i, F, W); }
}
__global__ void k_integrate_pre(Floes F, Params P, Work W) {   // grid: max(n, cap_ext) threads
  if (!W.counters[C_GO]) return;
  int z = blockIdx.x * blockDim.x + threadIdx.x;
  if (z < W.cap_ext) transfer_apply(z, F, W);
  int i = z + F.nb;
  if (i < F.n) integrate_pre(i, F, P, W);
  if (z < F.nb) { F.collision_force[2 * z] = 0; F.collision_force[2 * z + 1] = 0; F.collision_torque[z] = 0; F.OverlapArea[z] = 0; }
}
__global__ void k_integrate_post(Floes F, Params P, Work W) {
  if (!W.counters[C_GO]) return;
  int i = blockIdx.x * blockDim.x + threadIdx.x + F.nb;
  if (i < F.n) integrate_post(i, F, P, W);
}
// Ocean / wind forcing (calc_trajectory.m:95-98,121-166): one 256-thread workgroup per floe.  Thread t evaluates
// the Monte-Carlo points t, t + 256, ... (contiguous reads of the floe's point rows) and keeps partial t of
// the fixed summation order (step_core.hpp MC_PARTS); the workgroup then folds the 256 partials pairwise:
// across the four waves through LDS, inside the first wave with shuffles.
__device__ inline double mc_fold_block(double v, double *lds256) {
  const int t = threadIdx.x;
  lds256[t] = v;
  __syncthreads();
  if (t < 128) lds256[t] += lds256[t + 128];
  __syncthreads();
  double r = 0;
  if (t < 64) {
    r = lds256[t] + lds256[t + 64];
    for (int s = 32; s >= 1; s >>= 1) r += __shfl_down(r, s);   // lane t: p[t] += p[t + s]; lane 0 ends with the total
  }
  __syncthreads();                                              // lds256 is reused by the next fold
  return r;                                                     // valid in thread 0
}
static_assert(MC_PARTS == 256, "k_forcing folds exactly 256 partials");
// (five waves per SIMD, 96 registers: measured 0.279 ms for 1e4 floes on a forcing step against 0.308 at the four the
// compiler settles for by itself, 0.290 at six, 0.377 at eight -- the kernel waits on its gathers from the ocean grid)
#ifndef SZ_FORCING_WAVES
#define SZ_FORCING_WAVES 5
#endif
#define SZ_FORCING_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(SZ_FORCING_WAVES, SZ_FORCING_WAVES)))
// the forcing of floe i by the 256 threads of the workgroup (every thread is here with the same i); false: the floe's
// Monte-Carlo mask is empty (reported, st_flag cleared)
__device__ inline bool forcing_block(int i, const Floes &F, const Params &P, Work &W, double *fold, double *red, int *scnt) {
  const int t = threadIdx.x;
  const int nmc = F.n_mc;
  const double *mx = F.mcX + i * F.mc_si, *my = F.mcY + i * F.mc_si;
  const unsigned char *A = F.mcA + i * F.mc_si;
  const long long sq = F.mc_sq;
  ForceCtx c;
  force_ctx(i, F, P, c);
  // pass 1: every point inside the floe is sampled once -- winds for the first mean, ocean velocities and polar
  // angle kept in registers for the second (up to KEEP points per thread: n_mc <= 1024; beyond that they are
  // sampled again in pass 2)
  enum { KEEP = 4 };
  McPoint kept[KEEP];
  __syncthreads();                                              // (scnt / red of the floe before are done with)
  if (t == 0) *scnt = 0;
  __syncthreads();
  const bool keep = nmc <= 256 * KEEP;
  double a0 = 0.0, a1 = 0.0;
  int mine = 0;
#pragma unroll
  for (int s = 0; s < KEEP; ++s) {
    const int q = t + 256 * s;
    if (q < nmc && A[q * sq]) {
      double uw, vw;
      point_sample(c, P, mx[q * sq], my[q * sq], true, keep, uw, vw, kept[s]);
      a0 += uw; a1 += vw; ++mine;
    }
  }
  for (int q = t + 256 * KEEP; q < nmc; q += 256) {
    if (!A[q * sq]) continue;
    double uw, vw;
    wind_point(c, P, mx[q * sq], my[q * sq], uw, vw);
    a0 += uw; a1 += vw; ++mine;
  }
  if (mine) atomicAdd(scnt, mine);
  const double su = mc_fold_block(a0, fold), sv = mc_fold_block(a1, fold);
  if (t == 0) { red[0] = su; red[1] = sv; }
  __syncthreads();
  const int cnt = *scnt;
  if (cnt == 0) { // calc_trajectory.m:100-110 would regenerate the points with rand()
    if (t == 0) { atomicAdd(&W.counters[C_ERR_MC], 1ULL); W.st_flag[i] = 0; }
    return false;
  }
  wind_stress(c, red[0], red[1], cnt);
  // pass 2: stresses and torque
  a0 = 0.0; a1 = 0.0;
  double a2 = 0.0;
#pragma unroll
  for (int s = 0; s < KEEP; ++s) {
    const int q = t + 256 * s;
    if (q < nmc && A[q * sq]) {
      double Fx, Fy, tq;
      if (keep) point_force(c, kept[s], Fx, Fy, tq);
      else force_point(c, P, mx[q * sq], my[q * sq], Fx, Fy, tq);
      a0 += Fx; a1 += Fy; a2 += tq;
    }
  }
  for (int q = t + 256 * KEEP; q < nmc; q += 256) {
    if (!A[q * sq]) continue;
    double Fx, Fy, tq;
    force_point(c, P, mx[q * sq], my[q * sq], Fx, Fy, tq);
    a0 += Fx; a1 += Fy; a2 += tq;
  }
  const double sFx = mc_fold_block(a0, fold), sFy = mc_fold_block(a1, fold), sT = mc_fold_block(a2, fold);
  if (t == 0) { F.FxOA[i] = sFx / cnt; F.FyOA[i] = sFy / cnt; F.torqueOA[i] = sT / cnt; }
  return true;
}
// The same with the lanes kept busy: only the Monte-Carlo points INSIDE the floe are evaluated (~55 % of a Voronoi cell's
// 1000), and with point q on thread q mod 256 the other lanes of a wave sat idle through both passes.  The points inside
// are listed (ballots + a prefix over the workgroup), lane c mod 256 evaluates the c-th of them and leaves its values at
// the point's index in LDS, and thread t then adds up ITS points t, t + 256, ... from there -- the same values into the
// same partial in the same order as before (MC_PARTS), so every sum is bit-identical.  n_mc <= 1024.
struct ForcingLds { double v[3][1024]; double fold[256]; double red[3]; unsigned short lst[1024]; int wcnt[4][4]; int scnt; };
__device__ inline bool forcing_block_dense(int i, const Floes &F, const Params &P, Work &W, ForcingLds &L) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int nmc = F.n_mc;
  const double *mx = F.mcX + i * F.mc_si, *my = F.mcY + i * F.mc_si;
  const unsigned char *A = F.mcA + i * F.mc_si;
  const long long sq = F.mc_sq;
  ForceCtx c;
  force_ctx(i, F, P, c);
  __syncthreads();                                              // (the block of the floe before is done with)
  bool in[4];
  unsigned long long bal[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int q = t + 256 * s;
    in[s] = q < nmc && A[q * sq] != 0;
    bal[s] = __ballot(in[s]);
    if (lane == 0) L.wcnt[s][wv] = __popcll(bal[s]);
  }
  __syncthreads();
  int cnt = 0;                                                  // the list in order of (round s, wave, lane); cnt: its length
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w == wv && in[s]) L.lst[cnt + __popcll(bal[s] & ((1ULL << lane) - 1ULL))] = (unsigned short)(t + 256 * s);
      cnt += L.wcnt[s][w];
    }
  }
  __syncthreads();
  if (cnt == 0) { // calc_trajectory.m:100-110 would regenerate the points with rand()
    if (t == 0) { atomicAdd(&W.counters[C_ERR_MC], 1ULL); W.st_flag[i] = 0; }
    return false;
  }
  // pass 1: every point inside the floe is sampled once -- winds for the first mean, ocean velocities and polar angle
  // kept in registers for the second
  McPoint kept[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int cc = t + 256 * k;
    if (cc < cnt) {
      const int q = L.lst[cc];
      double uw, vw;
      point_sample(c, P, mx[q * sq], my[q * sq], true, true, uw, vw, kept[k]);
      L.v[0][q] = uw; L.v[1][q] = vw;
    }
  }
  __syncthreads();
  double a0 = 0.0, a1 = 0.0;
#pragma unroll
  for (int s = 0; s < 4; ++s) if (in[s]) { a0 += L.v[0][t + 256 * s]; a1 += L.v[1][t + 256 * s]; }
  const double su = mc_fold_block(a0, L.fold), sv = mc_fold_block(a1, L.fold);
  if (t == 0) { L.red[0] = su; L.red[1] = sv; }
  __syncthreads();
  wind_stress(c, L.red[0], L.red[1], cnt);
  // pass 2: stresses and torque
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int cc = t + 256 * k;
    if (cc < cnt) {
      const int q = L.lst[cc];
      double Fx, Fy, tq;
      point_force(c, kept[k], Fx, Fy, tq);
      L.v[0][q] = Fx; L.v[1][q] = Fy; L.v[2][q] = tq;
    }
  }
  __syncthreads();
  a0 = 0.0; a1 = 0.0;
  double a2 = 0.0;
#pragma unroll
  for (int s = 0; s < 4; ++s) if (in[s]) { a0 += L.v[0][t + 256 * s]; a1 += L.v[1][t + 256 * s]; a2 += L.v[2][t + 256 * s]; }
  const double sFx = mc_fold_block(a0, L.fold), sFy = mc_fold_block(a1, L.fold), sT = mc_fold_block(a2, L.fold);
  if (t == 0) { F.FxOA[i] = sFx / cnt; F.FyOA[i] = sFy / cnt; F.torqueOA[i] = sT / cnt; }
  return true;
}
__global__ void __launch_bounds__(256) SZ_FORCING_WAVES_ATTR k_forcing(int first, int last, Floes F, Params P, Work W) {
  __shared__ ForcingLds L;
  const int i = blockIdx.x + first;
  if (i >= last || !W.counters[C_GO]) return;
  if ((W.st_flag[i] & (ST_GO | ST_FORCE)) != (ST_GO | ST_FORCE)) return; // uniform over the workgroup
  if (F.n_mc <= 1024) (void)forcing_block_dense(i, F, P, W, L);
  else (void)forcing_block(i, F, P, W, L.fold, L.red, &L.scnt);
}
// The floes of force_list (kf_integrate; a step that does not refresh the forcing for everybody): forcing, then
// integrate_post, a few workgroups striding the list.  (No register cap here, unlike k_forcing: with integrate_post
// inlined under 96 registers the kernel spills and every launch -- nearly always over an empty list -- sets up scratch.)
__global__ void __launch_bounds__(256) kf_forcing_post_list(int first, int last, Floes F, Params P, Work W) {
  __shared__ double fold[256];
  __shared__ double red[3];
  __shared__ int scnt;
  if (!W.counters[C_GO]) return;
  const int n = (int)W.counters[C_FORCE];
  for (int k = blockIdx.x; k < n; k += gridDim.x) {
    const int i = W.force_list[k];
    if (i < first || i >= last) continue;
    (void)forcing_block(i, F, P, W, fold, red, &scnt);
    if (threadIdx.x == 0) integrate_post(i, F, P, W);
  }
}
#include "step_fused.hpp"
// sharded forcing: FxOA, FyOA, torqueOA of a floe range leave / arrive as three packed arrays of `cnt_max`
__global__ void k_force_export(int first, int count, int cnt_max, Floes F, double *dst) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  dst[k] = F.FxOA[first + k]; dst[cnt_max + k] = F.FyOA[first + k]; dst[2 * (size_t)cnt_max + k] = F.torqueOA[first + k];
}
__global__ void k_force_import(int first, int count, int cnt_max, Floes F, const double *src) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  F.FxOA[first + k] = src[k]; F.FyOA[first + k] = src[cnt_max + k]; F.torqueOA[first + k] = src[2 * (size_t)cnt_max + k];
}
// both launches of a scan over at most `bound` elements (the device-side count may be smaller)
template <class Job>
static void launch_scan(hipStream_t st, long long *agg, i64 bound, const Job &job) {
  i64 per = (bound + 255) / 256;
  int tile = (int)((per + 1023) / 1024) * 1024;
  if (tile < 1024) tile = 1024;
  int blocks = (int)((bound + tile - 1) / tile);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_scan_reduce<Job>, dim3(blocks), dim3(1024), 0, st, job, tile, agg);
  hipLaunchKernelGGL(k_scan_write<Job>, dim3(blocks), dim3(1024), 0, st, job, tile, (const long long *)agg);
}
} // namespace

// k_broad_filter / k_row_count give each working thread 2 * pot_cap + 1 ints of LDS: as many threads per workgroup
// as fit 64 KB, or slices of HBM (`spill`, `slices` of them) when not even one fits
static int filter_launch_shape(szContext *ctx, StepState *S, i64 slices, int &lanes, size_t &lds, int *&spill) {
  const size_t per_thread = (2 * (size_t)S->W.pot_cap + 1) * 4;
  spill = nullptr;
  lanes = (int)(((size_t)64 << 10) / per_thread);
  if (lanes > 64) lanes = 64;
  if (lanes < 1) {
    int rc = sz_reserve(ctx, S->filter_spill, (size_t)slices * per_thread);
    if (rc) return rc;
    spill = (int *)S->filter_spill.p; lanes = 64; lds = 0;
    return SZ_OK;
  }
  lds = (size_t)lanes * per_thread;
  return SZ_OK;
}

// tuning knobs (environment) and kernel attributes of a fresh StepState
int sz_step_configure(szContext *ctx, StepState *S) {
  if (const char *v = getenv("SUBZERO_NARROW_LANES")) { int l = atoi(v); if (l >= 1 && l <= 64) S->narrow_lanes = l; }
  if (const char *v = getenv("SUBZERO_NARROW_LDS")) { int b = atoi(v); if (b >= 0 && (size_t)b * S->narrow_lanes <= 160 * 1024) S->narrow_lds = b; }
  if (const char *v = getenv("SUBZERO_CONVEX_FAST")) S->convex_fast = atoi(v) != 0;
  if (const char *v = getenv("SUBZERO_TEAM")) S->team = atoi(v) != 0;
  if (const char *v = getenv("SUBZERO_WALLS_EARLY")) S->walls_early = atoi(v) != 0;
  if (const char *v = getenv("SUBZERO_CONVEX_LANES")) { int l = atoi(v); if (l >= 0 && l <= 64) S->convex_lanes = l; }
  if (const char *v = getenv("SUBZERO_RETRY_LANES")) { int l = atoi(v); if (l >= 0 && l <= 64) S->retry_lanes = l; }
  if (const char *v = getenv("SUBZERO_RETRY_LDS")) { int b = atoi(v); if (b >= 0 && b <= 40 * 1024) S->retry_lds = b; }
  if (const char *v = getenv("SUBZERO_RETRY_LDS_MAX")) { int b = atoi(v); if (b >= 0 && b <= 64 * 1024) S->retry_lds_max = b; }
  if (const char *v = getenv("SUBZERO_FAST_PATH")) S->fast_path = atoi(v) != 0;
  if (const char *v = getenv("SUBZERO_FAST_LANES")) { int l = atoi(v); if (l >= 1 && l <= 64) S->fast_lanes = l; }
  if (const char *v = getenv("SUBZERO_SPECULATE")) S->no_speculation = atoi(v) == 0;
  if (const char *v = getenv("SUBZERO_SPECULATE_BACK")) S->no_back_speculation = atoi(v) == 0;
  if (const char *v = getenv("SUBZERO_BROAD_WAVE")) S->broad_wave = atoi(v) != 0;
  if (const char *v = getenv("SUBZERO_SPIN_WAIT")) S->spin_wait = atoi(v) != 0;
  if (const char *v = getenv("SUBZERO_FUSED")) S->fused_on = atoi(v) != 0;
  if (const char *v = getenv("SUBZERO_CELL_CAP")) { int l = atoi(v); if (l >= 1 && l <= 4096) S->cell_cap = l; }
  S->no_hint = getenv("SUBZERO_NO_HINT") != nullptr;
  S->debug_lists = getenv("SUBZERO_DEBUG_LISTS") != nullptr;
  if (const char *v = getenv("SUBZERO_POT_CAP")) { int l = atoi(v); if (l >= 1 && l <= 4096) S->pot_cap = l; }
  S->wall_tasks = getenv("SUBZERO_WALL_TASKS") != nullptr && atoi(getenv("SUBZERO_WALL_TASKS")) != 0;
  if (const char *v = getenv("SUBZERO_WAVE_TASKS")) { int l = atoi(v); if (l >= 1 && l <= 4) S->wave_tasks = l; }
  if (const char *v = getenv("SUBZERO_BIG_LANES")) { int l = atoi(v); if (l >= 0 && l <= 64) S->big_lanes = l; }
  { const int rcg = sz_narrow_general_configure(ctx, S->narrow_lds != 0); if (rcg) return rcg; }
  (void)hipFuncSetAttribute((const void *)k_broad_filter, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  (void)hipFuncSetAttribute((const void *)k_row_count, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  return SZ_OK;
}

extern "C" {

// ---- the step in three parts, so that the narrow-phase can be sharded over GPUs ----------------------
// front : ghosts, grid, broad-phase, flat pair list, per-pair buffers      (every rank, identical results)
// narrow: pair_contact on the pairs [lo, hi) + capacity-tier retries       (one segment per rank)
// back  : mirror lists, contact rows, sums, calc_trajectory                (every rank, identical results)
enum { STEP_RETRY_EXACT = 1000, STEP_BACK_NOGO = 1001 };   // internal: the speculative step did not fit its capacities; run it again the exact way
// the step's parameters on the device side: ocean / wind grids, Params
static int step_params(szContext *ctx, StepState *S, const szStepParams *hp) {
  hipStream_t st = ctx->stream;
  int rc;
  // ocean / wind grids: uploaded on the first step and whenever the forcing is refreshed
  if (!S->have_ocean || hp->doInt) {
    if (hp->nxo < 2 || hp->nyo < 2 || !hp->Xo || !hp->Yo || !hp->Uocn || !hp->Vocn || !hp->Uwind || !hp->Vwind)
      return sz_fail(ctx, SZ_ERR_BAD_ARG, "szStep: ocean / wind grids missing");
    size_t g = (size_t)hp->nxo * hp->nyo * 8;
    if ((rc = sz_reserve(ctx, S->oXo, (size_t)hp->nxo * 8))) return rc;
    if ((rc = sz_reserve(ctx, S->oYo, (size_t)hp->nyo * 8))) return rc;
    if ((rc = sz_reserve(ctx, S->oU, g)) || (rc = sz_reserve(ctx, S->oV, g)) || (rc = sz_reserve(ctx, S->oWu, g)) ||
        (rc = sz_reserve(ctx, S->oWv, g))) return rc;
    SZ_HIP(ctx, hipMemcpyAsync(S->oXo.p, hp->Xo, (size_t)hp->nxo * 8, hipMemcpyHostToDevice, st));
    SZ_HIP(ctx, hipMemcpyAsync(S->oYo.p, hp->Yo, (size_t)hp->nyo * 8, hipMemcpyHostToDevice, st));
    SZ_HIP(ctx, hipMemcpyAsync(S->oU.p, hp->Uocn, g, hipMemcpyHostToDevice, st));
    SZ_HIP(ctx, hipMemcpyAsync(S->oV.p, hp->Vocn, g, hipMemcpyHostToDevice, st));
    SZ_HIP(ctx, hipMemcpyAsync(S->oWu.p, hp->Uwind, g, hipMemcpyHostToDevice, st));
    SZ_HIP(ctx, hipMemcpyAsync(S->oWv.p, hp->Vwind, g, hipMemcpyHostToDevice, st));
    S->nxo = hp->nxo; S->nyo = hp->nyo; S->have_ocean = true;
    axis_extent(hp->Xo, hp->nxo, S->xo_max, S->xo_min); axis_extent(hp->Yo, hp->nyo, S->yo_max, S->yo_min);
  }
  Params &P = S->P;
  P.dt = hp->dt; P.HFo = hp->HFo; P.Modulus = hp->Modulus; P.mu = hp->mu; P.nu = hp->nu;
  P.doInt = hp->doInt; P.COLLISION = hp->COLLISION; P.PERIODIC = hp->PERIODIC; P.reserved = 0;   // (hp->flags: StepState::step_flags)
  memcpy(P.c2_boundary, hp->c2_boundary, sizeof(P.c2_boundary));
  P.fb_h = hp->fb_h; P.fb_area = hp->fb_area; P.fb_Xi = hp->fb_Xi; P.fb_Yi = hp->fb_Yi; P.fb_Ui = hp->fb_Ui;
  P.fb_Vi = hp->fb_Vi; P.fb_ksi_ice = hp->fb_ksi_ice;
  P.nxo = S->nxo; P.nyo = S->nyo;
  P.Xo_max = S->xo_max; P.Xo_min = S->xo_min; P.Yo_max = S->yo_max; P.Yo_min = S->yo_min;
  P.Xo = (const double *)S->oXo.p; P.Yo = (const double *)S->oYo.p;
  P.Uocn = (const double *)S->oU.p; P.Vocn = (const double *)S->oV.p;
  P.Uwind = (const double *)S->oWu.p; P.Vwind = (const double *)S->oWv.p;
  P.fCoriolis = hp->fCoriolis; P.turn_angle = hp->turn_angle;
  P.Lx = P.c2_boundary[0]; P.Ly = P.c2_boundary[1];
  for (int k = 1; k < 5; ++k) {
    if (P.c2_boundary[2 * k] > P.Lx) P.Lx = P.c2_boundary[2 * k];
    if (P.c2_boundary[2 * k + 1] > P.Ly) P.Ly = P.c2_boundary[2 * k + 1];
  }
  boundary_summary(P);
  return SZ_OK;
}
// per-pair buffers and scratch for np pairs with nvt ring vertices in all (grow-only); -> bytes of the closed-form
// part of the scratch arena (the retry tiers allocate behind it)
static int bind_pairs(szContext *ctx, StepState *S, i64 np, i64 nvt, i64 &main_bytes) {
  Work &W = S->W;
  int rc;
  {
    const size_t npa = (size_t)np + 1;
    if ((rc = sz_reserve(ctx, S->pair_i, npa * 4)) || (rc = sz_reserve(ctx, S->pair_j, npa * 4)) ||
        (rc = sz_reserve(ctx, S->pair_nvoff, npa * 8)) || (rc = sz_reserve(ctx, S->pair_scr, npa * 8)) ||
        (rc = sz_reserve(ctx, S->pair_mult, npa * 4)) || (rc = sz_reserve(ctx, S->pair_nrows, npa * 4)) ||
        (rc = sz_reserve(ctx, S->pair_status, npa * 4)) || (rc = sz_reserve(ctx, S->pair_flag, npa * 4)) ||
        (rc = sz_reserve(ctx, S->retry_list, npa * 4)) || (rc = sz_reserve(ctx, S->retry_copy, npa * 4)) ||
        (rc = sz_reserve(ctx, S->wall_list, npa * 4)) ||
        (rc = sz_reserve(ctx, S->big_list, npa * 4)) || (rc = sz_reserve(ctx, S->team_list, npa * 4)) || (rc = sz_reserve(ctx, S->wide_list, npa * 4)) || (rc = sz_reserve(ctx, S->live_list, npa * 4)) || (rc = sz_reserve(ctx, S->cold_list, npa * 4)) ||
        (rc = sz_reserve(ctx, S->pair_rowoff, npa * 8)) || (rc = sz_reserve(ctx, S->touch_list, npa * 4)) ||
        (rc = sz_reserve(ctx, S->pair_nreg, npa * 4)) || (rc = sz_reserve(ctx, S->pair_task, npa * 16))) return rc;
    main_bytes = (i64)SCR_PER_V * (nvt + 2 * np) + (i64)SCR_PER_PAIR * np;
    const i64 reserve = main_bytes / 2 + (8 << 20);
    if ((rc = sz_reserve(ctx, S->scratch, (size_t)(main_bytes + reserve)))) return rc;
    W.cap_pairs = (i64)npa;
    W.pair_i = (int *)S->pair_i.p; W.pair_j = (int *)S->pair_j.p; W.pair_nvoff = (i64 *)S->pair_nvoff.p;
    W.pair_scr = (i64 *)S->pair_scr.p; W.pair_mult = (int *)S->pair_mult.p; W.pair_nrows = (int *)S->pair_nrows.p;
    W.pair_status = (int *)S->pair_status.p; W.pair_flag = (int *)S->pair_flag.p; W.retry_list = (int *)S->retry_list.p;
    // (a list of their own for the floes at the walls when the step before had any: step_narrow_first)
    // (... and deferred no other pair: a general pass after the read-back then sweeps the wall pairs in the time it
    // takes anyway -- uniaxial200, whose six many-sided pairs a step go that way)
    W.wall_list = (S->walls_early && S->classified && S->retry_lds > 0 && S->stepped && S->last_walls > 0 && S->last_deferred == 0) ? (int *)S->wall_list.p : nullptr;
    W.big_list = (int *)S->big_list.p; W.team_list = (int *)S->team_list.p; W.wide_list = (int *)S->wide_list.p; W.live_list = (int *)S->live_list.p; W.cold_list = (int *)S->cold_list.p;
    W.pair_rowoff = (i64 *)S->pair_rowoff.p;
    W.touch_list = (int *)S->touch_list.p; W.pair_nreg = (int *)S->pair_nreg.p; W.pair_task = (int *)S->pair_task.p;
    W.scratch = (char *)S->scratch.p; W.scratch_bytes = (i64)S->scratch.cap;
    return SZ_OK;
  }
}
static int step_front(szContext *ctx, StepState *S, const szStepParams *hp, bool speculative = false, bool classify = true) {
  hipStream_t st = ctx->stream;
  int rc;
  if ((rc = step_params(ctx, S, hp))) return rc;
  Params &P = S->P;
  Floes &F = S->F;
  Work &W = S->W;
  const int n = S->n;
  const int cap = W.cap_ext;
  W.g_rows = nullptr;
  W.cell_cap = 0; W.reg_rows = 0;
  S->fused = false;

  SZ_HIP(ctx, hipEventRecord(S->ev[0], st));
  unsigned long long *c = S->c;
  S->mc_empty.clear();
  i64 main_bytes = 0;
  for (int attempt = 0;; ++attempt) {
    if (attempt > 0 || !S->counters_clean) hipLaunchKernelGGL(k_counters_clear, dim3(1), dim3(64), 0, st, F, W);
    S->counters_clean = false;
    // hierarchy of uniform grids (step_core.hpp grid_levels); positions outside clamp to the border cells
    grid_levels(S->rmax_max, S->rmax_p90, P.Lx, P.Ly, W);
    const size_t nc = (size_t)W.lvl_base[W.n_levels] + 1;
    if ((rc = sz_reserve(ctx, S->cell_cnt, nc * 4)) || (rc = sz_reserve(ctx, S->cell_off, nc * 4)) ||
        (rc = sz_reserve(ctx, S->cell_fill, nc * 4)) || (rc = sz_reserve(ctx, S->cell_item, (size_t)cap * sizeof(CellItem)))) return rc;
    W.cell_item = (CellItem *)S->cell_item.p;
    W.cell_cnt = (int *)S->cell_cnt.p; W.cell_off = (int *)S->cell_off.p; W.cell_fill = (int *)S->cell_fill.p;
    hipLaunchKernelGGL(k_ext_clear, grid_for(cap, 256), dim3(256), 0, st, F, P, W);   // (clears the cells as well)
    if (P.PERIODIC) {
      static const bool two_passes = getenv("SUBZERO_GHOST_TWO_PASSES") != nullptr;
      if (two_passes) {
        for (int dim = 0; dim < 2; ++dim) {
          int bound = (dim == 0) ? n : 2 * n;
          launch_scan(st, S->scan_agg, bound, ScanGhost{dim, F, P, W});   // flags, slots and the ghosts themselves
        }
      } else launch_scan(st, S->scan_agg, n, ScanGhosts{F, P, W});      // both dimensions: flags, slots and the ghosts themselves
    }
    {
      hipLaunchKernelGGL(k_ext_records, grid_for(16 * (i64)cap, 256), dim3(256), 0, st, F, W);   // (counts the cells as well)
      launch_scan(st, S->scan_agg, (i64)nc - 1, ScanCells{W});
      hipLaunchKernelGGL(k_cell_fill, grid_for(cap, 256), dim3(256), 0, st, F, W);
    }
    // real floes: a wave per floe does scan, sort, filter and pair count in one launch when the grid has one level and
    // the lists are short (the common case); ghosts -- whose lists start from their root's -- and everything else go
    // through the thread-per-floe kernels
    const bool wave = S->broad_wave && W.n_levels == 1 && W.pot_cap <= 64 && n > F.nb;
    if (wave) {
      hipLaunchKernelGGL(k_broad_wave, dim3((unsigned)(n - F.nb)), dim3(64), 0, st, F, P, W);
      if (P.PERIODIC) hipLaunchKernelGGL(k_broad_scan, grid_for(9 * (i64)(cap - n), 256), dim3(256), 0, st, n - F.nb, F, P, W);
    } else {
      hipLaunchKernelGGL(k_broad_scan, grid_for(9 * (i64)(P.PERIODIC ? cap : n), 256), dim3(256), 0, st, 0, F, P, W);
    }
    {
      int lanes; size_t lds; int *spill = nullptr;
      if ((rc = filter_launch_shape(ctx, S, 3 * (i64)n + 64, lanes, lds, spill))) return rc;
      if (!wave) hipLaunchKernelGGL(k_broad_filter, grid_for(n, lanes), dim3(64), lds, st, 1, lanes, F, P, W, spill);
      if (P.PERIODIC) hipLaunchKernelGGL(k_broad_filter, grid_for(3 * (i64)n, lanes), dim3(64), lds, st, 0, lanes, F, P, W, spill);
    }
    launch_scan(st, S->scan_agg, cap, ScanPairs{W});   // (writing the pair records from inside this scan was measured: +19 us, the per-floe loops serialise in its 1024-thread workgroups)
    SZ_HIP(ctx, hipGetLastError());
    SZ_HIP(ctx, hipEventRecord(S->ev[1], st));
    if (speculative) break;   // no read-back here: the counts are checked with those of the narrow phase (step_narrow)
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, W.counters, S->c_dev, -1LL, GoRule{GO_LEAVE, 0, 0, 0, 0, 0}, 0, (const unsigned long long *)(S->fused ? S->W.prow_part : nullptr));
    SZ_HIP(ctx, hipGetLastError());
    SZ_HIP(ctx, hipStreamSynchronize(st));
    if (c[C_NEED_MC]) {   // nothing of the floe array has been touched yet: the caller regenerates the points and steps again
      const size_t k = (size_t)c[C_NEED_MC];
      S->mc_empty.resize(k);
      SZ_HIP(ctx, hipMemcpy(S->mc_empty.data(), W.mc_list, k * 4, hipMemcpyDeviceToHost));
      std::sort(S->mc_empty.begin(), S->mc_empty.end());
      sz_fail(ctx, SZ_STEP_NEEDS_MC, "szStep: %zu floe(s) need new Monte-Carlo points (calc_trajectory.m:100-110); nothing was changed", k);
      return SZ_STEP_NEEDS_MC;
    }
    if (!c[C_ERR_POT]) break;
    // a candidate list or a mirror list would not fit: grow both to the size this step asks for and run the front again
    // (the capacity is kept for the following steps)
    int want = S->pot_cap * 2;
    while ((unsigned long long)want < c[C_NEED_POT]) want *= 2;
    if (attempt >= 6 || (size_t)cap * want * 8 > ((size_t)16 << 30))
      return sz_fail(ctx, SZ_ERR_WORKSPACE, "broad-phase: a floe has %llu candidates; lists of that size are not supported", c[C_NEED_POT]);
    S->pot_cap = want;
    if ((rc = bind_lists(ctx, S))) return rc;
  }
  S->speculating = speculative;
  const i64 np = speculative ? S->spec_np : (i64)c[C_NPAIRS], nvt = speculative ? S->spec_nvt : (i64)c[C_NVTOT];
  S->np = np; S->nvt = nvt; S->N = (int)c[C_NEXT]; S->retried = 0;

  // the pairs sorted for the team pass (narrow_team.hip): boxes apart -> answered, the others onto their kernel's list
  S->classified = classify && S->team != 0 && S->convex_fast != 0 && S->fast_path == 0;
  if ((rc = bind_pairs(ctx, S, np, nvt, main_bytes))) return rc;
  hipLaunchKernelGGL(k_pair_expand, grid_for(cap, 256), dim3(256), 0, st, F, P, W, (unsigned long long)main_bytes);
  if (S->classified)
    sz_launch_pair_classify(st, np, speculative ? (const unsigned long long *)&W.counters[C_NPAIRS] : (const unsigned long long *)nullptr,
                            (S->stepped && !S->no_hint) ? S->last_next : 0, P, W);
  SZ_HIP(ctx, hipGetLastError());
  return SZ_OK;
}

// Can this step run the fused launch chain (step_fused.hpp)?  One grid level, lists of at most 64 candidates, the team
// pass first, counts that fit the packed tile sums, a cell array of bearable size.
static bool fused_eligible(StepState *S, const szStepParams *hp) {
  if (!S->fused_on || !(S->team != 0 && S->convex_fast != 0 && S->fast_path == 0) || S->pot_cap > 64 || !S->broad_wave) return false;
  if (S->n >= (1 << 21) || S->spec_np >= (1 << 24) || S->world > 1) return false;
  double Lx = hp->c2_boundary[0], Ly = hp->c2_boundary[1];
  for (int k = 1; k < 5; ++k) { if (hp->c2_boundary[2 * k] > Lx) Lx = hp->c2_boundary[2 * k]; if (hp->c2_boundary[2 * k + 1] > Ly) Ly = hp->c2_boundary[2 * k + 1]; }
  Work tmp = S->W;
  grid_levels(S->rmax_max, S->rmax_p90, Lx, Ly, tmp);
  if (tmp.n_levels != 1) return false;
  return (size_t)tmp.lvl_base[1] * (size_t)S->cell_cap * sizeof(CellItem) <= ((size_t)512 << 20);
}
// what kf_front_a / kf_front_b read of the parameters (ghost rule, grid, the Monte-Carlo pre-check)
static bool front_params_same(const Params &a, const Params &b) {
  if (!(a.dt == b.dt && a.HFo == b.HFo && a.doInt == b.doInt && a.COLLISION == b.COLLISION && a.PERIODIC == b.PERIODIC && a.Lx == b.Lx && a.Ly == b.Ly)) return false;
  for (int k = 0; k < 10; ++k) if (!(a.c2_boundary[k] == b.c2_boundary[k])) return false;
  return true;
}
// The front stages and the pair records of a speculative step as four launches (step_fused.hpp); capacities from the step
// before, no read-back.
static int step_front_fused(szContext *ctx, StepState *S, const szStepParams *hp) {
  hipStream_t st = ctx->stream;
  int rc;
  if ((rc = step_params(ctx, S, hp))) return rc;
  Params &P = S->P;
  Floes &F = S->F;
  Work &W = S->W;
  const int n = S->n, cap = W.cap_ext;
  W.g_rows = nullptr;
  S->mc_empty.clear();
  const bool was_clean = S->counters_clean;
  S->counters_clean = false;
  grid_levels(S->rmax_max, S->rmax_p90, P.Lx, P.Ly, W);
  const size_t nc = (size_t)W.lvl_base[W.n_levels] + 1;
  const int ntiles = (n + 63) / 64, ntiles_ext = (cap + 63) / 64;
  if ((rc = sz_reserve(ctx, S->cell_cnt, nc * 4 * CELL_STRIDE)) || (rc = sz_reserve(ctx, S->cell_item, nc * (size_t)S->cell_cap * sizeof(CellItem))) ||
      (rc = sz_reserve(ctx, S->gh_mask, 16 * (size_t)ntiles)) || (rc = sz_reserve(ctx, S->gh_agg, 8 * (size_t)ntiles)) ||
      (rc = sz_reserve(ctx, S->pair_agg, 8 * (64 / BROAD_WAVES) * (size_t)ntiles_ext)) || (rc = sz_reserve(ctx, S->row_agg, 8 * (size_t)AGG_STRIDE * (ntiles_ext + 64))) ||
      (rc = sz_reserve(ctx, S->rel_mask, 8 * (size_t)cap)) || (rc = sz_reserve(ctx, S->pot_nvpre, 4 * (size_t)cap * S->pot_cap)) ||
      (rc = sz_reserve(ctx, S->force_list, 4 * (size_t)n)) || (rc = sz_reserve(ctx, S->hint, 32 * (size_t)cap))) return rc;
  W.cell_item = (CellItem *)S->cell_item.p; W.cell_cnt = (int *)S->cell_cnt.p; W.cell_cap = S->cell_cap;
  W.gh_mask = (unsigned long long *)S->gh_mask.p; W.gh_agg = (unsigned long long *)S->gh_agg.p;
  W.pair_agg = (unsigned long long *)S->pair_agg.p; W.row_agg = (unsigned long long *)S->row_agg.p;
  W.prow_part = W.row_agg + (size_t)AGG_STRIDE * ntiles_ext;     // (64 more lines of the same buffer)
  W.hint = (int *)S->hint.p;
  W.rel_mask = (unsigned long long *)S->rel_mask.p; W.pot_nvpre = (int *)S->pot_nvpre.p; W.force_list = (int *)S->force_list.p;
  S->speculating = true; S->fused = true;
  W.reg_rows = 1;
  S->next_cap = S->stepped ? std::min<i64>(cap, (i64)S->last_next + (S->last_next - n) / 4 + 256) : (i64)cap;   // entries the back half's grid covers (GoRule)
  const i64 np = S->spec_np, nvt = S->spec_nvt;
  S->np = np; S->nvt = nvt; S->N = (int)S->c[C_NEXT]; S->retried = 0;
  S->classified = true;
  i64 main_bytes = 0;
  if ((rc = bind_pairs(ctx, S, np, nvt, main_bytes))) return rc;
  // the first two kernels: queued ahead behind the step before (step_back_b) -- used if nothing they read has changed
  // since -- or launched now
  const StepState::FrontSig sig = {W.cell_item, W.cell_cnt, W.gh_mask, W.gh_agg, W.row_agg, W.prow_part, W.cell_cap, cap, (int)nc, ntiles, ntiles_ext, 0};
  S->front_sig = sig;
  const bool ahead = S->ahead_pending && was_clean && front_params_same(S->ahead_P, P) && memcmp(&S->ahead_sig, &sig, sizeof(sig)) == 0;
  S->ahead_pending = false;
  if (ahead) ++S->steps_ahead;
  else {
    SZ_HIP(ctx, hipEventRecord(S->ev[0], st));
    if (!was_clean) hipLaunchKernelGGL(k_counters_clear, dim3(1), dim3(64), 0, st, F, W);
    hipLaunchKernelGGL(kf_front_a, grid_for(cap, 256), dim3(256), 0, st, (int)nc - 1, ntiles_ext, 0, F, P, W);
  }
#if defined(SZ_HOST_CLOCKS)
  g_hc.first += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - g_hc.entry).count();
#endif
  if (!ahead) hipLaunchKernelGGL(kf_front_b, dim3((unsigned)ntiles), dim3(1024), 0, st, ntiles, 0, F, P, W);
  const unsigned tiles = grid_for(S->next_cap, 64).x;      // tiles of 64 entries the per-entry launches cover
  hipLaunchKernelGGL(kf_broad, dim3((64 / BROAD_WAVES) * tiles), dim3(64 * BROAD_WAVES), 0, st, F, P, W);
  SZ_HIP(ctx, hipEventRecord(S->ev[1], st));
  hipLaunchKernelGGL(kf_pairs, dim3(tiles), dim3(512), 0, st, (S->hint_valid && !S->no_hint) ? S->last_next : 0, (unsigned long long)main_bytes, F, P, W);
  SZ_HIP(ctx, hipGetLastError());
  return SZ_OK;
}

// first pass of the narrow phase over the pairs [lo, hi) and the counters after it; no read-back
static int step_narrow_first(szContext *ctx, StepState *S, i64 lo, i64 hi, bool queue_back) {
  hipStream_t st = ctx->stream;
  Floes &F = S->F; Work &W = S->W; Params &P = S->P;
  SZ_HIP(ctx, hipEventRecord(S->ev[2], st));
  bool team_timed = false;
  const bool convex = S->convex_fast != 0 && S->fast_path == 0;
  const bool fast = S->fast_path != 0 || convex;   // a first pass that defers its leftovers to the 1x general tier
  if (hi > lo) {
    if (convex && S->team) {
      // The wall pairs first, on this stream, and the team pass beside them on a stream of its own: a wall pair's
      // sweep is one wave of 368 registers for a quarter of a millisecond, which finds no room once the team pass has
      // filled every SIMD -- started first it takes one SIMD each and the team pass the rest.
      hipStream_t ts = st;
      if (W.wall_list) {
        { const int rcs = sz_side_streams(ctx); if (rcs) return rcs; }
        if (!S->wall_fork) {
          SZ_HIP(ctx, hipEventCreateWithFlags(&S->wall_fork, hipEventDisableTiming));
          SZ_HIP(ctx, hipEventCreateWithFlags(&S->wall_join, hipEventDisableTiming));
        }
        // (a fork and a join between two streams cost some tens of microseconds: a team pass shorter than that follows
        // the wall pairs on this stream instead -- which still spares such a field the host round trips of the deferred
        // pairs: uniaxial200)
        if (S->last_hot + S->last_cold >= 4096) {
          ts = ctx->side[0];
          SZ_HIP(ctx, hipEventRecord(S->wall_fork, st));
          SZ_HIP(ctx, hipStreamWaitEvent(ts, S->wall_fork, 0));
        }
        // (sized from the last step's list; a 16-64 KB slice of LDS per wave while every workgroup still finds a place
        // at once; the waves stride over the list)
        i64 wg = S->last_walls + S->last_walls / 4 + 16;
        if (wg > 1024) wg = 1024;
        const i64 per_cu = (wg + 255) / 256;
        i64 room = ((i64)160 * 1024 / per_cu) & ~(i64)4095;      // (whole 4 KB: the CU hands LDS out in coarser units than 1 KB -- measured below, step_narrow_rest)
        if (room > S->retry_lds_max) room = S->retry_lds_max;
        if (room < S->retry_lds) room = S->retry_lds;
        static const bool walls_lane0 = getenv("SUBZERO_WALLS_LANE0") != nullptr;     // (a tuning comparison)
        szl_k_narrow_walls(dim3((unsigned)wg), (size_t)room, st, walls_lane0 ? -(int)room : (int)room, (S->fused && !walls_lane0 && S->wall_tasks) ? S->wave_tasks : 1, F, P, W);
        SZ_HIP(ctx, hipGetLastError());
        SZ_HIP(ctx, hipEventRecord(S->ev[2], ts));
      }
      // the team pass over every pair, then the lane-per-pair fast path over what it handed on (a device-side list: the
      // launch covers the pair count and its workgroups beyond the list return at once)
      if (S->classified) sz_launch_narrow_team(ts, hi - lo, S->stepped ? S->last_hot : -1, S->last_cold, S->last_wide, F, P, W);
      else sz_launch_narrow_team_range(ts, lo, hi - lo, F, P, W);
      SZ_HIP(ctx, hipEventRecord(S->ev[3], ts));   // narrow_ms = the team pass alone (the kernel bench.py prices)
      team_timed = true;
      // (the pass over what the team pass handed on: in a steady field that list is empty step after step -- when the
      // last one was and the back half is queued behind this launch, the pass is not launched; a pair that does turn up
      // on the list keeps the back half from running (GoRule) and the pass is made up for after the read-back)
      S->convex_skipped = queue_back && S->stepped && S->last_team == 0 && S->classified;
      if (!S->convex_skipped) {
        const int L = 16;
        szl_k_narrow_convex(grid_for(hi - lo, L), (size_t)L * CONVEX_STRIDE, ts, (i64)0, hi - lo,
                           (const unsigned long long *)&W.counters[C_TEAM], L, F, P, W, (const int *)W.team_list);
      }
      if (ts != st) {
        SZ_HIP(ctx, hipEventRecord(S->wall_join, ts));
        SZ_HIP(ctx, hipStreamWaitEvent(st, S->wall_join, 0));
      }
    } else if (convex) {
      // one wave per SIMD is all this kernel can use (it is bound by the path of its longest lane): as few pairs
      // per wave as still give every one of the 1024 SIMDs at most one wave -- fewer lanes, less divergence
      int L = S->convex_lanes;
      const i64 expect = S->speculating ? S->last_npairs : hi - lo;   // (speculative step: the launch covers a capacity, the pairs are about last step's)
      if (L == 0) { const i64 want = (expect + 1023) / 1024; L = (int)(want < 16 ? 16 : want > 64 ? 64 : want); }
      const int *order = nullptr;
#if defined(SZ_SORT_EXPERIMENT)
      // EXPERIMENT (diagnostic build): the pairs handed to the waves in descending order of their vertex total, sorted
      // on the host -- does a wave of like pairs finish sooner?
      static DevBuf d_order;
      if (!S->speculating && lo == 0 && getenv("SUBZERO_SORT_PAIRS")) {
        const i64 np = hi;
        std::vector<int> pi((size_t)np), pj((size_t)np), nvx((size_t)W.cap_ext), ord((size_t)np);
        SZ_HIP(ctx, hipStreamSynchronize(st));
        SZ_HIP(ctx, hipMemcpy(pi.data(), W.pair_i, (size_t)np * 4, hipMemcpyDeviceToHost));
        SZ_HIP(ctx, hipMemcpy(pj.data(), W.pair_j, (size_t)np * 4, hipMemcpyDeviceToHost));
        SZ_HIP(ctx, hipMemcpy(nvx.data(), W.ext_nv, (size_t)W.cap_ext * 4, hipMemcpyDeviceToHost));
        for (i64 p = 0; p < np; ++p) ord[(size_t)p] = (int)p;
        const int mode = atoi(getenv("SUBZERO_SORT_PAIRS"));
        auto key = [&](int p) { const int a = nvx[(size_t)pi[(size_t)p]], b = pj[(size_t)p] >= 0 ? nvx[(size_t)pj[(size_t)p]] : 4; return (mode == 2 || mode == 4) ? (a > b ? a : b) : a + b; };
        std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return key(a) > key(b); });
        if (mode >= 3) {   // dealt out: wave w gets the pairs w, w + nw, w + 2 nw ... of the sorted order (a like mix in every wave)
          const i64 nw = (np + L - 1) / L;
          std::vector<int> dealt((size_t)np);
          i64 at = 0;
          for (i64 w = 0; w < nw; ++w) for (i64 k = w; k < np; k += nw) dealt[(size_t)at++] = ord[(size_t)k];
          ord.swap(dealt);
        }
        int rc2 = sz_reserve(ctx, d_order, (size_t)np * 4 + 16);
        if (rc2) return rc2;
        SZ_HIP(ctx, hipMemcpy(d_order.p, ord.data(), (size_t)np * 4, hipMemcpyHostToDevice));
        order = (const int *)d_order.p;
        SZ_HIP(ctx, hipEventRecord(S->ev[2], st));
      }
#endif
      szl_k_narrow_convex(grid_for(hi - lo, L), (size_t)L * CONVEX_STRIDE, st, lo, hi - lo,
                         S->speculating ? (const unsigned long long *)&W.counters[C_NPAIRS] : (const unsigned long long *)nullptr, L, F, P, W, order);
    } else if (fast) {
      const int L = S->fast_lanes;
      const size_t lds = (size_t)L * FAST_WS_STRIDE;
      szl_k_narrow_a(grid_for(hi - lo, L), lds, st, lo, hi - lo, L, F, P, W);
      szl_k_narrow_b(grid_for(hi - lo, L), lds, st, L, F, P, W);
    } else {
      szl_k_narrow(grid_for(hi - lo, S->narrow_lanes), (size_t)S->narrow_lds * S->narrow_lanes, st, lo, hi - lo, (const int *)nullptr, 1, S->narrow_lanes, S->narrow_lds, F, P, W);
    }
    SZ_HIP(ctx, hipGetLastError());
  }
  if (!team_timed) SZ_HIP(ctx, hipEventRecord(S->ev[3], st));
  // the counters for the host -- and for the back half of the step, which runs only once the narrow phase is complete:
  // queued right behind this launch (queue_back) it goes ahead if nothing was deferred and everything fitted
  const GoRule rule = queue_back ? GoRule{GO_IF_CLEAN, (long long)S->spec_np, (long long)S->spec_nvt, (long long)S->spec_rows, S->convex_skipped ? 1 : 0, S->fused ? (long long)S->next_cap : 0LL} : GoRule{GO_SET, 0, 0, 0, 0, 0};
  S->go_rule = rule;
  S->publish_with_incoming = queue_back && S->spec_np > 0;      // (step_back_a's first kernel then does both)
  if (!S->publish_with_incoming) hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, W.counters, S->c_dev, -1LL, rule, 0, (const unsigned long long *)(S->fused ? S->W.prow_part : nullptr));
  SZ_HIP(ctx, hipGetLastError());
  return SZ_OK;
}

// after the read-back of the first pass: capacities of a speculative step, then the pairs the first pass deferred
static int step_narrow_rest(szContext *ctx, StepState *S) {
  hipStream_t st = ctx->stream;
  Floes &F = S->F; Work &W = S->W; Params &P = S->P;
  unsigned long long *c = S->c;
  const bool convex = S->convex_fast != 0 && S->fast_path == 0;
  const bool fast = S->fast_path != 0 || convex;
  if (S->speculating) {
    // The front stages ran without a read-back, for capacities taken from the last step.  Nothing of the floe array
    // has been changed so far (the narrow phase writes its own buffers only), so whatever does not fit is simply done
    // again the exact way.
    S->speculating = false;
    if (c[C_ERR_CELL]) {   // fused front: a grid cell was full -- more slots from the next step on, this one the scan-based way
      while ((unsigned long long)S->cell_cap < c[C_NEED_CELL] && S->cell_cap < (1 << 20)) S->cell_cap *= 2;
      return STEP_RETRY_EXACT;
    }
    if (c[C_NEED_MC] || c[C_ERR_POT] || (i64)c[C_NPAIRS] > S->spec_np || (i64)c[C_NVTOT] > S->spec_nvt || c[C_ERR_ARENA] ||
        (S->fused && (i64)c[C_NEXT] > S->next_cap))      // (fused front: its per-entry launches covered next_cap entries)
      return STEP_RETRY_EXACT;
    S->np = (i64)c[C_NPAIRS]; S->nvt = (i64)c[C_NVTOT]; S->N = (int)c[C_NEXT];
  }
  if (S->convex_skipped) {
    S->convex_skipped = false;
    if (c[C_TEAM] > 0) {    // pairs the team pass handed on after all: their pass now, then the counters again
      const i64 nt = (i64)c[C_TEAM];
      const int L = 16;
      szl_k_narrow_convex(grid_for(nt, L), (size_t)L * CONVEX_STRIDE, st, (i64)0, nt,
                         (const unsigned long long *)nullptr, L, F, P, W, (const int *)W.team_list);
      SZ_HIP(ctx, hipGetLastError());
      hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, W.counters, S->c_dev, -1LL, GoRule{GO_LEAVE, 0, 0, 0, 0, 0}, 0, (const unsigned long long *)(S->fused ? S->W.prow_part : nullptr));
      SZ_HIP(ctx, hipGetLastError());
      SZ_HIP(ctx, hipStreamSynchronize(st));
    }
  }
  if (convex && c[C_BIG] > 0) {
    // rings of 17-32 vertices: the fast path again, on the larger block; its leftovers join the general kernel's list
    const i64 nb = (i64)c[C_BIG];
    if (S->big_lanes > 0) {
      const int L = S->big_lanes;
      szl_k_narrow_convex_big(grid_for(nb, L), (size_t)L * CONVEX_BIG_STRIDE, st, nb, (const int *)S->big_list.p, L, F, P, W);
    } else {   // (pass switched off: straight to the general kernel, through the same list mechanics)
      SZ_HIP(ctx, hipMemcpyAsync((int *)S->retry_list.p + c[C_RETRY], S->big_list.p, (size_t)nb * 4, hipMemcpyDeviceToDevice, st));
      S->bump = c[C_RETRY] + (unsigned long long)nb;
      SZ_HIP(ctx, hipMemcpyAsync(&W.counters[C_RETRY], &S->bump, 8, hipMemcpyHostToDevice, st));
    }
    SZ_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, W.counters, S->c_dev, -1LL, GoRule{GO_LEAVE, 0, 0, 0, 0, 0}, 0, (const unsigned long long *)(S->fused ? S->W.prow_part : nullptr));
    SZ_HIP(ctx, hipGetLastError());
    SZ_HIP(ctx, hipStreamSynchronize(st));
  }
  int mult = fast ? 0 : 1; // what the fast path could not hold starts at the 1x tier of the general path
  while (c[C_RETRY] > 0) {
    mult = mult ? mult * 8 : 1;
    if (mult > 512) return sz_fail(ctx, SZ_ERR_WORKSPACE, "narrow-phase: %llu pair(s) outgrew the 512x scratch tier", c[C_RETRY]);
    i64 nl = (i64)c[C_RETRY];
    if (mult == (fast ? 1 : 8)) S->retried = nl;
    SZ_HIP(ctx, hipMemcpyAsync(S->retry_copy.p, S->retry_list.p, (size_t)nl * 4, hipMemcpyDeviceToDevice, st));
    S->zero = 0;
    SZ_HIP(ctx, hipMemcpyAsync(&W.counters[C_RETRY], &S->zero, 8, hipMemcpyHostToDevice, st));
    // the general sweep of one pair is a chain of dependent accesses that takes milliseconds for long rings: few
    // pairs per wave keep the wave as short as its longest pair, as long as the waves still fit the chip
    int RL = S->narrow_lanes;
    if (fast) {
      RL = S->retry_lanes;
      if (RL == 0) { i64 want = (nl + 2047) / 2048; RL = (int)(want < 1 ? 1 : want > 64 ? 64 : want); }
    }
    // deferred pairs at the 1x tier: a short list of short rings (ties, floes at the walls) runs with the sweep
    // workspace in LDS, one to four pairs per wave; a pair that needs more than the slice keeps its HBM workspace
    // (the slice grows from SUBZERO_RETRY_LDS, 16 KB, up to 64 KB while every workgroup of the launch still finds
    // a place on the 256 CUs at once: a 64 KB slice holds the sweep of 119 trimmed vertices)
    int retry_lds = (fast && mult == 1 && RL <= 4) ? S->retry_lds : 0;
    if (retry_lds > 0) {
      const i64 wgs = (nl + RL - 1) / RL, per_cu = (wgs + 255) / 256;
      i64 room = (i64)160 * 1024 / (per_cu * RL);
      // (whole 4 KB: three workgroups of 53 KB do not fit a CU's 160 KB, three of 52 KB do -- the LDS is handed out in coarser
      // units than 1 KB; measured on the evolved field: 0.84 ms/step with two workgroups per CU, 0.73 with three)
      room = (room > S->retry_lds_max ? S->retry_lds_max : room) & ~(i64)4095;
      if (room > retry_lds) retry_lds = (int)room;
    }
    static const bool wave_off = getenv("SUBZERO_WAVE_GENERAL") != nullptr && atoi(getenv("SUBZERO_WAVE_GENERAL")) == 0;
    if (retry_lds > 0 && RL == 1 && mult == 1 && !wave_off)
    {
      // (`wave_tasks` waves per pair: those a pair has no use for leave at once -- narrow_t; only the fused chain's
      // classification knows how many rows a pair had: elsewhere every pair is one task)
      const int tasks = S->fused ? S->wave_tasks : 1;
      szl_k_narrow_wave(dim3((unsigned)(nl * tasks)), (size_t)retry_lds, st, nl, (const int *)S->retry_copy.p, retry_lds, tasks, F, P, W);
    }
    else if (retry_lds > 0)
      szl_k_narrow_lds(grid_for(nl, RL), (size_t)retry_lds * RL, st, nl, (const int *)S->retry_copy.p, mult, RL, retry_lds, F, P, W);
    else
      szl_k_narrow(grid_for(nl, RL), 0, st, (i64)0, nl, (const int *)S->retry_copy.p, mult, RL, 0, F, P, W);
    SZ_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, W.counters, S->c_dev, -1LL, GoRule{GO_LEAVE, 0, 0, 0, 0, 0}, 0, (const unsigned long long *)(S->fused ? S->W.prow_part : nullptr));
    SZ_HIP(ctx, hipGetLastError());
    SZ_HIP(ctx, hipStreamSynchronize(st));
  }
  if (c[C_ERR_ARENA]) return sz_fail(ctx, SZ_ERR_WORKSPACE, "narrow-phase: %llu pair(s) did not fit the scratch arena", c[C_ERR_ARENA]);
  if (c[C_ERR_CLIP]) return sz_fail(ctx, SZ_ERR_HIP, "narrow-phase: %llu pair(s) hit the reference's \"Clipper Error.\" condition", c[C_ERR_CLIP]);
  S->pair_rows_local = (i64)c[C_PAIRROWS];
  S->pair_rows_total = S->pair_rows_local;   // (the sharded step replaces this by the sum over the ranks)
  return SZ_OK;
}

static int step_narrow(szContext *ctx, StepState *S, i64 lo, i64 hi) {
  int rc;
  if ((rc = step_narrow_first(ctx, S, lo, hi, false))) return rc;
  SZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return step_narrow_rest(ctx, S);
}

static int step_forcing(szContext *ctx, StepState *S, int first, int last) {
  if (last <= first) return SZ_OK;
  if (S->fused) {   // forcing and integrate_post of the floes kf_integrate left for it
    // (a step that refreshes the forcing: every live floe, a workgroup each, then integrate_post -- the scan-based chain's two kernels)
    if (!S->P.doInt) {
      hipLaunchKernelGGL(kf_forcing_post_list, dim3(32), dim3(256), 0, ctx->stream, first, last, S->F, S->P, S->W);
      SZ_HIP(ctx, hipGetLastError());
      return SZ_OK;
    }
  }
  hipLaunchKernelGGL(k_forcing, dim3(last - first), dim3(256), 0, ctx->stream, first, last, S->F, S->P, S->W);
  SZ_HIP(ctx, hipGetLastError());
  return SZ_OK;
}
// floes [first, last) whose forcing rank r of w evaluates on a forcing step
static void force_range(const StepState *S, int r, int w, int &first, int &last) {
  const long long m = S->n - S->nb;
  first = S->nb + (int)(m * r / w); last = S->nb + (int)(m * (r + 1) / w);
}
static int force_cnt_max(const StepState *S, int w) { return (S->n - S->nb + w - 1) / w; }

static int step_back_a(szContext *ctx, StepState *S) {
  hipStream_t st = ctx->stream;
  Floes &F = S->F; Work &W = S->W; Params &P = S->P;
  const i64 np = S->back_queued ? S->spec_np : S->np;     // (queued behind the narrow phase: a capacity, the device knows the count)
  const int n = S->n, cap = W.cap_ext;
  int rc;
  // No host round trip here: every contact row of a pair appears at most twice in the CSR (once for each floe
  // of the pair), so the pair rows counted by the narrow phase bound it; the exact count and the overflow flag
  // of the mirror lists are read with the counters at the end of the step.
  const i64 nr_cap = S->back_queued ? S->spec_rows : 2 * S->pair_rows_total + 1;
  if ((rc = sz_reserve(ctx, S->rows, 56 * (size_t)nr_cap))) return rc;
  W.rows = (double *)S->rows.p; W.cap_rows = nr_cap;
  if (S->fused) {
    // the pairs have registered their rows as they completed (pair_register): one launch places and writes them
    const int publish = S->publish_with_incoming ? 1 : 0;
    S->publish_with_incoming = false;
    const i64 entries = S->back_queued ? S->next_cap : (i64)S->N;
    hipLaunchKernelGGL(kf_rows, grid_for(entries > 0 ? entries : 1, 64), dim3(1024), 0, st, publish, S->c_dev, S->go_rule, F, P, W);
    SZ_HIP(ctx, hipEventRecord(S->ev[4], st));
    hipLaunchKernelGGL(kf_integrate, grid_for(std::max<i64>(16 * (i64)(n - S->nb), cap), 256), dim3(256), 0, st, F, P, W);
    SZ_HIP(ctx, hipGetLastError());
    return SZ_OK;
  }
  if (S->publish_with_incoming) {
    S->publish_with_incoming = false;
    hipLaunchKernelGGL(k_incoming_go, grid_for(np > 0 ? np : 1, 256), dim3(256), 0, st, np, W, S->c_dev, S->go_rule);
  } else if (np > 0) hipLaunchKernelGGL(k_incoming, grid_for(np, 256), dim3(256), 0, st, np, W);
  {
    int lanes; size_t lds; int *spill = nullptr;
    if ((rc = filter_launch_shape(ctx, S, (i64)cap + 64, lanes, lds, spill))) return rc;
    hipLaunchKernelGGL(k_row_count, grid_for(cap, lanes), dim3(64), lds, st, lanes, W, spill, 0);
  }
  launch_scan(st, S->scan_agg, cap, ScanRows{W});
  hipLaunchKernelGGL(k_row_fill, grid_for(cap, 64), dim3(64), 0, st, F, P, W);
  SZ_HIP(ctx, hipEventRecord(S->ev[4], st));
  hipLaunchKernelGGL(k_integrate_pre, grid_for(n > cap ? n : cap, 64), dim3(64), 0, st, F, P, W);
  SZ_HIP(ctx, hipGetLastError());
  return SZ_OK;
}

// step / broad / pairs / narrow / rows / integrate milliseconds from a set of events whose last one has completed
static void stage_times(hipEvent_t *ev, float *out) {
  const int a[6] = {0, 0, 1, 2, 3, 4}, b[6] = {5, 1, 2, 3, 4, 5};
  for (int k = 0; k < 6; ++k) { float ms = 0; if (hipEventElapsedTime(&ms, ev[a[k]], ev[b[k]]) != hipSuccess) ms = 0; out[k] = ms; }
}
// at the start of a step: the events of the step before become the set that is read late
static void begin_step_events(StepState *S) {
  for (int k = 0; k < 6; ++k) std::swap(S->ev[k], S->ev_prev[k]);
  S->prev_pending = S->ev_recorded; S->ev_recorded = false;
  if (S->prev_pending) S->prev_step = S->steps_done - 1;
}
static int step_back_b(szContext *ctx, StepState *S, szStepStats *stats) {
  hipStream_t st = ctx->stream;
  Floes &F = S->F; Work &W = S->W; Params &P = S->P;
  unsigned long long *c = S->c;
  const int n = S->n;
  if (!S->fused || P.doInt) hipLaunchKernelGGL(k_integrate_post, grid_for(n, 64), dim3(64), 0, st, F, P, W);
  SZ_HIP(ctx, hipGetLastError());
  SZ_HIP(ctx, hipEventRecord(S->ev[5], st));
  const unsigned long long seq = S->spin_wait ? ++S->publish_seq : 0ULL;
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, W.counters, S->c_dev, (long long)n, GoRule{GO_LEAVE, 0, 0, 0, 0, 0}, S->back_queued ? 1 : 0, (const unsigned long long *)(S->fused ? S->W.prow_part : nullptr), seq);
  SZ_HIP(ctx, hipGetLastError());
#if defined(SZ_HOST_CLOCKS)
  const auto hc_q = std::chrono::steady_clock::now();
  g_hc.queue += std::chrono::duration<double, std::micro>(hc_q - g_hc.entry).count();
#endif
  if (S->prev_pending) { stage_times(S->ev_prev, S->prev_ms); S->prev_pending = false; }   // (the step before: complete long ago, nothing waits)
  // queue-ahead: the next step's first two kernels behind this step's last one (they run only if this step ran to its
  // end: C_AHEAD, written by the publish above).  Not for a sharded step (the halo arrives between two steps).
  bool queued_ahead = false;
  if ((S->step_flags & SZ_STEP_QUEUE_AHEAD) && S->fused && !W.present && S->front_sig.cell_item) {
    Params Pn = P;
    Pn.doInt = (S->step_flags & SZ_STEP_NEXT_DOINT) ? 1 : 0;
    const StepState::FrontSig &g = S->front_sig;
    SZ_HIP(ctx, hipEventRecord(S->ev_prev[0], st));      // (that set's times have just been read: it is the next step's now)
    hipLaunchKernelGGL(kf_front_a, grid_for(g.cap_ext, 256), dim3(256), 0, st, g.nc - 1, g.ntiles_ext, 1, F, Pn, W);
    hipLaunchKernelGGL(kf_front_b, dim3((unsigned)g.ntiles), dim3(1024), 0, st, g.ntiles, 1, F, Pn, W);
    SZ_HIP(ctx, hipGetLastError());
    S->ahead_P = Pn; S->ahead_sig = g;
    queued_ahead = true;
  }
  if (seq) {
    // the step's last kernel raises a flag in the pinned copy of the counters: spin on it (with a look at the stream now and
    // then, so that a failed launch cannot leave the host spinning)
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned long long spins = 0;; ++spins) {
      if (__atomic_load_n(&c[C_COUNT], __ATOMIC_ACQUIRE) == seq) break;
      if ((spins & 0xfff) == 0xfff) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) { if (__atomic_load_n(&c[C_COUNT], __ATOMIC_ACQUIRE) == seq) break; SZ_HIP(ctx, hipStreamSynchronize(st)); break; }
        if (q != hipErrorNotReady) SZ_HIP(ctx, q);
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) { SZ_HIP(ctx, hipStreamSynchronize(st)); break; }
      }
      __builtin_ia32_pause();
    }
  } else SZ_HIP(ctx, hipStreamSynchronize(st));
#if defined(SZ_HOST_CLOCKS)
  const auto hc_w = std::chrono::steady_clock::now();
  g_hc.wait += std::chrono::duration<double, std::micro>(hc_w - hc_q).count();
#endif
  if (S->back_queued) {
    S->back_queued = false;
    if (!c[C_GO]) return STEP_BACK_NOGO;     // the narrow phase left work for the host: nothing after it has run, the counters are as it left them
    S->speculating = false;
    S->np = (i64)c[C_NPAIRS]; S->nvt = (i64)c[C_NVTOT]; S->N = (int)c[C_NEXT];
    S->pair_rows_local = S->pair_rows_total = (i64)c[C_PAIRROWS];
    S->retried = 0;
  }
  S->counters_clean = true;
  S->nr = (i64)c[C_NROWS];
  if (c[C_ERR_INC]) return sz_fail(ctx, SZ_ERR_WORKSPACE, "contact mirror: a floe has more than %d incoming pairs", W.pot_cap);
  if (c[C_ERR_MC])
    return sz_fail(ctx, SZ_ERR_STATE, "integrate: %llu floe(s) have an empty Monte-Carlo mask; regenerating it needs MATLAB's rand (calc_trajectory.m:100-110)", c[C_ERR_MC]);
  const i64 np = S->np;
  S->last_npairs = np; S->last_next = S->N; S->stepped = true;
  ++S->steps_done; if (S->fused) ++S->steps_fused;
  S->ahead_valid = queued_ahead;     // (the step ran to its end: what was queued ahead runs too)
  S->ghosts_overwritten = queued_ahead;
  S->hint_valid = S->fused;      // (kf_rows has left next step's contact hints)
  S->last_team_vertices = (i64)c[C_TEAMV];
  S->last_hot = (i64)c[C_LIVE]; S->last_cold = (i64)c[C_COLD]; S->last_wide = (i64)c[C_WIDE]; S->last_walls = (i64)c[C_WALLS]; S->last_team = (i64)c[C_TEAM];
  S->last_deferred = S->retried - (W.wall_list ? 0 : S->last_walls) + (i64)c[C_BIG];   // pairs of this step that waited for the host besides the floes at the walls
  if (S->debug_lists) fprintf(stderr, "lists: pairs %lld hot %lld cold %lld wide %lld other %lld pair_rows %lld | walls %lld (%s) deferred besides %lld big %lld\n", (long long)np, (long long)c[C_LIVE], (long long)c[C_COLD], (long long)c[C_WIDE], (long long)c[C_TEAM], (long long)c[C_PAIRROWS], (long long)c[C_WALLS], W.wall_list ? "beside the team pass" : "on the retry list", (long long)S->last_deferred, (long long)c[C_BIG]);
  S->ev_recorded = true;
  if (stats) {
    // (the stage times of the step BEFORE: this step's events are read while the host waits for the next one -- or by
    // szStepTimes, which waits for them)
    stats->step_ms = S->prev_ms[0]; stats->broad_ms = S->prev_ms[1]; stats->pairs_ms = S->prev_ms[2]; stats->narrow_ms = S->prev_ms[3];
    stats->rows_ms = S->prev_ms[4]; stats->integrate_ms = S->prev_ms[5]; stats->times_step = S->prev_step;
    stats->n_ext = S->N; stats->n_pairs = np; stats->n_rows = S->nr; stats->pair_vertices = S->nvt;
    stats->retried = S->retried + (W.wall_list ? S->last_walls : 0);   // (pairs through the general sweep: deferred ones and the floes at the walls)
    stats->pair_rows = S->pair_rows_local;
    stats->team_pairs = S->last_hot + S->last_cold + S->last_wide; stats->team_vertices = S->last_team_vertices;
  }
#if defined(SZ_HOST_CLOCKS)
  g_hc.ret = std::chrono::steady_clock::now(); g_hc.have = true;
  g_hc.tail += std::chrono::duration<double, std::micro>(g_hc.ret - hc_w).count();
  if (++g_hc.n == 200) {
    fprintf(stderr, "host clocks per step (us): entry -> first launch %.1f, entry -> everything queued %.1f, waiting for the step %.1f, flag -> return %.1f, between two calls (the caller) %.1f\n",
            g_hc.first / g_hc.n, g_hc.queue / g_hc.n, g_hc.wait / g_hc.n, g_hc.tail / g_hc.n, g_hc.between / g_hc.n);
    g_hc.n = 0; g_hc.first = g_hc.queue = g_hc.wait = g_hc.tail = g_hc.between = 0;
  }
#endif
  return SZ_OK;
}

// capacities of the next speculative step: this step's counts and a margin
static void spec_caps(StepState *S) {
  S->spec_np = S->np + S->np / 4 + 1024;
  S->spec_nvt = S->nvt + S->nvt / 4 + 16384;
  const i64 rows = 2 * S->pair_rows_total + 1;
  S->spec_rows = rows + rows / 4 + 4096;
  // the back half is queued behind the narrow phase of the next step when this one deferred nothing (a field whose
  // floes lean on the walls defers pairs every step: its steps keep the read-back in the middle)
  S->back_spec_ok = !S->no_back_speculation && S->retried == 0 && S->c[C_BIG] == 0 && S->world <= 1;
}
static int step_back(szContext *ctx, StepState *S, szStepStats *stats) {
  int rc;
  if ((rc = step_back_a(ctx, S))) return rc;
  if ((rc = step_forcing(ctx, S, S->nb, S->n))) return rc;
  return step_back_b(ctx, S, stats);
}

// (a vetoed step has run its front half and narrow phase at most: the floe array is as it was, the next szStep starts over)
static int step_vetoed(szContext *ctx, StepState *S) {
  S->counters_clean = false;
  S->back_queued = false; S->publish_with_incoming = false;
  ++S->steps_vetoed;
  sz_fail(ctx, SZ_STEP_VETOED, "szStep: a neighbour's message did not fit what was agreed to travel (szSpatialUnpackQueued); nothing was changed");
  return SZ_STEP_VETOED;
}

int szStep(szContext *ctx, const szStepParams *hp, szStepStats *stats) {
  if (!ctx || !hp) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S) return sz_fail(ctx, SZ_ERR_STATE, "szStep: no floes uploaded");
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  begin_step_events(S);
  S->step_flags = hp->flags;
  S->ahead_pending = S->ahead_valid; S->ahead_valid = false;     // (what was queued ahead serves this call or none)
#if defined(SZ_HOST_CLOCKS)
  g_hc.entry = std::chrono::steady_clock::now();
  if (g_hc.have) g_hc.between += std::chrono::duration<double, std::micro>(g_hc.entry - g_hc.ret).count();
#endif
  int rc;
  const bool speculative = S->spec_np > 0 && !S->no_speculation && S->convex_fast && !S->fast_path;
  // Spatial partition, queued exchange: the neighbours' messages were unpacked without a look at their headers
  // (szSpatialUnpackQueued).  A message that outgrew what was agreed to travel has raised C_VETO on the device: the
  // queued back half then does not run (GoRule) and this call reports SZ_STEP_VETOED with nothing changed.  A step
  // that would read the counters back half-way anyway (not in steady state) asks before it starts.
  const bool veto_armed = S->sp_unpack_queued;
  S->sp_unpack_queued = false;
  if (veto_armed && !(speculative && S->back_spec_ok)) {
    SZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (S->sp_hdr && S->sp_hdr[4 * SP_HDR_WORDS]) return step_vetoed(ctx, S);
  }
  // Steady state: the front stages and the narrow phase are queued back to back, sized by the last step's counts plus
  // a margin, and validated at the one read-back after the narrow phase (one host round trip fewer per step).
  if (speculative) {
    if ((rc = fused_eligible(S, hp) ? step_front_fused(ctx, S, hp) : step_front(ctx, S, hp, true))) return rc;
    const bool queue_back = S->back_spec_ok;
    if ((rc = step_narrow_first(ctx, S, 0, S->spec_np, queue_back))) return rc;
    if (queue_back) {
      // the whole step in one go: the back half runs if the narrow phase completed every pair within the capacities,
      // and does nothing at all otherwise (the floe array is untouched until then)
      S->back_queued = true;
      rc = step_back(ctx, S, stats);
      if (rc != STEP_BACK_NOGO) {
        S->back_queued = false;
        if (!rc) spec_caps(S);
        return rc;
      }
      if (veto_armed && S->c[C_VETO]) return step_vetoed(ctx, S);
      ++S->back_misses;
      hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, ctx->stream, S->W.counters, S->c_dev, -1LL, GoRule{GO_SET, 0, 0, 0, 0, 0}, 0, (const unsigned long long *)(S->fused ? S->W.prow_part : nullptr));   // from here on the host decides
      SZ_HIP(ctx, hipGetLastError());
    }
    SZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rc = step_narrow_rest(ctx, S);
    if (rc != STEP_RETRY_EXACT) {
      if (rc) return rc;
      rc = step_back(ctx, S, stats);
      if (!rc) spec_caps(S);
      return rc;
    }
    S->counters_clean = false;
    ++S->spec_misses;
  }
  if ((rc = step_front(ctx, S, hp))) return rc;
  if ((rc = step_narrow(ctx, S, 0, S->np))) return rc;
  rc = step_back(ctx, S, stats);
  if (!rc) spec_caps(S);
  return rc;
}

// ---- sharded form: Begin (front + my segment of the narrow-phase) / Export / Finish (import all + back) ----
int64_t szShardSegmentBytes(int64_t seg_max, int64_t rows_max) { return (int64_t)seg_bytes_for(seg_max, rows_max); }

int szStepShardBegin(szContext *ctx, const szStepParams *hp, int rank, int world, szShardInfo *info) {
  if (!ctx || !hp || !info || world < 1 || rank < 0 || rank >= world) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S) return sz_fail(ctx, SZ_ERR_STATE, "szStepShardBegin: no floes uploaded");
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  begin_step_events(S);
  memset(info, 0, sizeof(*info));
  info->seg_rows = -1;
  int rc;
  if ((rc = step_front(ctx, S, hp, false, false))) { info->status = rc; info->replicated_failure = 1; return rc; }
  S->rank = rank; S->world = world;
  shard_range(S->np, rank, world, S->seg_lo, S->seg_hi);
  rc = step_narrow(ctx, S, S->seg_lo, S->seg_hi);
  info->status = rc;
  info->n_pairs = S->np; info->seg_lo = S->seg_lo; info->seg_hi = S->seg_hi;
  info->seg_max = shard_max(S->np, world);
  if (rc) return rc;
  // rows before each pair of my segment (exclusive scan of pair_nrows over the segment); the total stays on
  // the device (counters[C_PAIRROWS]) and goes into the segment header
  const i64 ns = S->seg_hi - S->seg_lo;
  if ((rc = sz_reserve(ctx, S->x_off, (size_t)(info->seg_max + 1) * 8))) { info->status = rc; return rc; }
  launch_scan(ctx->stream, S->scan_agg, ns, ScanSegment{(const int *)S->W.pair_nrows + S->seg_lo, (i64 *)S->x_off.p, (int)ns,
                                                       &S->W.counters[C_PAIRROWS]});
  SZ_HIP(ctx, hipGetLastError());
  return SZ_OK;
}

int szStepShardExport(szContext *ctx, void *d_send, int64_t seg_max, int64_t rows_max, int32_t status) {
  if (!ctx || !d_send) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S) return sz_fail(ctx, SZ_ERR_STATE, "szStepShardExport: no floes uploaded");
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  const i64 ns = status ? 0 : S->seg_hi - S->seg_lo;   // a failed rank sends its header only
  if (ns > seg_max) return sz_fail(ctx, SZ_ERR_CAPACITY, "szStepShardExport: segment does not fit");
  char *base = (char *)d_send;
  Xfer X = seg_view(base, seg_max, rows_max);
  hipLaunchKernelGGL(k_export_header, dim3(1), dim3(1), 0, ctx->stream, base, ns, (const unsigned long long *)&S->W.counters[C_PAIRROWS],
                     (i64)status, (i64)rows_max);
  if (ns > 0) hipLaunchKernelGGL(k_export, grid_for(ns, 256), dim3(256), 0, ctx->stream, ns, S->seg_lo, (const i64 *)S->x_off.p, S->F, S->W, X);
  SZ_HIP(ctx, hipGetLastError());
  if (ctx->stream == ctx->own_stream) SZ_HIP(ctx, hipStreamSynchronize(ctx->stream)); // the collective runs on another stream
  return SZ_OK;
}

int szStepShardFinish(szContext *ctx, const void *d_recv, int64_t seg_max, int64_t rows_max, szStepStats *stats,
                      int32_t *forcing_exchange, int64_t *rows_needed) {
  if (!ctx || !d_recv) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S) return sz_fail(ctx, SZ_ERR_STATE, "szStepShardFinish: no floes uploaded");
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  const size_t seg_bytes = seg_bytes_for(seg_max, rows_max);
  int rc;
  // the headers first: did every rank succeed, does every segment fit?
  S->hdr.resize(4 * (size_t)S->world);
  SZ_HIP(ctx, hipMemcpy2DAsync(S->hdr.data(), SEG_HDR, d_recv, seg_bytes, SEG_HDR, (size_t)S->world, hipMemcpyDeviceToHost, ctx->stream));
  SZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  i64 need = 0;
  for (int r = 0; r < S->world; ++r) {
    const long long *h = &S->hdr[4 * (size_t)r];
    if (h[2] != 0) return sz_fail(ctx, SZ_ERR_PEER, "sharded step: rank %d reported error %lld", r, h[2]);
    i64 lo, hi;
    shard_range(S->np, r, S->world, lo, hi);
    if (h[0] != hi - lo || h[3] != rows_max)
      return sz_fail(ctx, SZ_ERR_STATE, "sharded step: rank %d sent %lld pairs (rows_max %lld), expected %lld (%lld): the ranks are out of step",
                     r, h[0], h[3], (long long)(hi - lo), (long long)rows_max);
    if (h[1] > need) need = h[1];
  }
  if (rows_needed) *rows_needed = need;
  if (need > rows_max) return SZ_SHARD_RESIZE;
  S->pair_rows_local = S->hdr[4 * (size_t)S->rank + 1];
  S->pair_rows_total = 0;
  for (int r = 0; r < S->world; ++r) S->pair_rows_total += S->hdr[4 * (size_t)r + 1];
  S->W.g_rows = (double *)d_recv; // pair_rowoff is in doubles from the start of the gathered buffer
  if (S->np > 0) hipLaunchKernelGGL(k_import, grid_for(S->np, 256), dim3(256), 0, ctx->stream, S->np, S->world, (i64)seg_max, (i64)rows_max, (char *)d_recv, S->W);
  SZ_HIP(ctx, hipGetLastError());
  rc = step_back_a(ctx, S);
  S->W.g_rows = nullptr; // the rows have been copied into the CSR; d_recv may be reused by the caller
  if (rc) return rc;
  // forcing: on a forcing step every rank evaluates its own floe range and the results are exchanged
  // (szStepShardForcing / szStepShardEnd); otherwise the few thin floes (h < 0.1) are done redundantly
  S->force_exchange = (S->P.doInt && S->world > 1) ? 1 : 0;
  if (forcing_exchange) *forcing_exchange = S->force_exchange;
  if (S->force_exchange) {
    int f0, f1;
    force_range(S, S->rank, S->world, f0, f1);
    return step_forcing(ctx, S, f0, f1);
  }
  if ((rc = step_forcing(ctx, S, S->nb, S->n))) return rc;
  return step_back_b(ctx, S, stats);
}

#if defined(SZ_FUSED_CLOCKS)
int szDebugFusedClocks(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(sz_fused_clock), sizeof(unsigned long long) * SZ_FC_KERNELS * SZ_FC_BLOCKS * SZ_FC_SLOTS) == hipSuccess ? 0 : -1;
}
#endif
int szStepTimes(szContext *ctx, szStepStats *stats) {
  if (!ctx || !stats) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S || !S->ev_recorded) return sz_fail(ctx, SZ_ERR_STATE, "szStepTimes: no completed step");
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  SZ_HIP(ctx, hipEventSynchronize(S->ev[5]));   // (the runtime sees the event's signal a little after the step's own flag reached the host)
  float ms[6];
  stage_times(S->ev, ms);
  stats->step_ms = ms[0]; stats->broad_ms = ms[1]; stats->pairs_ms = ms[2]; stats->narrow_ms = ms[3]; stats->rows_ms = ms[4]; stats->integrate_ms = ms[5];
  stats->times_step = S->steps_done - 1;
  return SZ_OK;
}
int szStepDiagnostics(szContext *ctx, int64_t *out, int32_t n) {
  if (!ctx || !out || n < 0) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S) return sz_fail(ctx, SZ_ERR_STATE, "szStepDiagnostics: no floes uploaded");
  const int64_t v[SZ_DIAG_COUNT] = {S->steps_done, S->steps_fused, S->spec_misses, S->back_misses, S->last_hot, S->last_cold, S->last_wide,
                                    S->last_team, S->last_walls, S->last_team_vertices, S->cell_cap, S->pot_cap, S->steps_ahead};
  for (int k = 0; k < n && k < (int)SZ_DIAG_COUNT; ++k) out[k] = v[k];
  return SZ_OK;
}

int64_t szShardForcingBytes(szContext *ctx, int world) {
  if (!ctx || !ctx->step || world < 1) return 0;
  return 24 * (int64_t)force_cnt_max(ctx->step, world);
}

int szStepShardForcing(szContext *ctx, void *d_send) {
  if (!ctx || !d_send) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S || !S->force_exchange) return sz_fail(ctx, SZ_ERR_STATE, "szStepShardForcing: no forcing exchange pending");
  int f0, f1;
  force_range(S, S->rank, S->world, f0, f1);
  const int cm = force_cnt_max(S, S->world);
  if (f1 > f0) hipLaunchKernelGGL(k_force_export, grid_for(f1 - f0, 256), dim3(256), 0, ctx->stream, f0, f1 - f0, cm, S->F, (double *)d_send);
  SZ_HIP(ctx, hipGetLastError());
  if (ctx->stream == ctx->own_stream) SZ_HIP(ctx, hipStreamSynchronize(ctx->stream)); // the collective runs on another stream
  return SZ_OK;
}

int szStepShardEnd(szContext *ctx, const void *d_recv, szStepStats *stats) {
  if (!ctx) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S) return sz_fail(ctx, SZ_ERR_STATE, "szStepShardEnd: no floes uploaded");
  if (!S->force_exchange) return SZ_OK; // Finish already completed the step
  if (!d_recv) return SZ_ERR_BAD_ARG;
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  const int cm = force_cnt_max(S, S->world);
  for (int r = 0; r < S->world; ++r) {
    if (r == S->rank) continue;
    int f0, f1;
    force_range(S, r, S->world, f0, f1);
    if (f1 > f0) hipLaunchKernelGGL(k_force_import, grid_for(f1 - f0, 256), dim3(256), 0, ctx->stream, f0, f1 - f0, cm, S->F,
                                    (const double *)d_recv + 3 * (size_t)cm * r);
  }
  SZ_HIP(ctx, hipGetLastError());
  S->force_exchange = 0;
  return step_back_b(ctx, S, stats);
}

} // extern "C"
