// step_fused.hpp -- the steady step's launch chain with its global scans folded into the kernels on either side
// (included by step.hip inside its anonymous namespace; the bodies are those of step_core.hpp).
//
// Round 3 ended with ~22 launches of 5-20 us around the team kernel: every prefix sum of the step was two launches of
// its own between two per-floe kernels, and 1e4 floes are 157 waves on a chip of 1024 SIMDs -- each of those launches
// is latency.  A kernel boundary is the cheapest grid-wide hand-off on this chip (a look-back or a ticket between
// workgroups on different XCDs costs more than the ~2 us of a back-to-back launch: measured in round 3), so the chain
// below keeps one boundary per TRUE dependency and nothing else:
//
//   kf_front_a   per entry    clears, ext_init, mc_precheck, ghost flags -> a ballot and a packed count per TILE of 64 floes
//   kf_front_b   16 lanes / floe  sums the tile counts before its tile (<= a few hundred words), places the floe's ghosts
//                             (:33-37,:54-58 in the order of the reference's two passes), writes the records and staged
//                             rings of the floe and of its ghosts, appends each to its grid cell (fixed slots per cell:
//                             no count -> scan -> fill)
//   kf_broad     wave / floe  candidate list of the floe (k_broad_wave) and then of its ghosts, whose `mems` start from the
//                             floe's finished list (:94-99) -- which this wave has just made; pair counts per entry and, by
//                             one atomic, per tile of 64 entries
//   kf_pairs     tile of 64 entries   sums the tile counts before its tile, scans its own, writes the pair records and
//                             classifies each pair in the thread that wrote it (pair_classify)
//   [team pass]
//   k_incoming_go, k_row_count (+ tile counts)
//   kf_row_fill  tile of 64 entries   row offsets from the tile counts, then row_fill
//   kf_integrate per floe     transfer marks, integrate_pre and -- where no forcing is due -- integrate_post
//   kf_forcing_post  workgroup / floe   forcing + integrate_post of the floes whose forcing is due
//   k_publish
// Same arrays, same values as the scan-based chain (which stays: first steps, grids of several levels, lists beyond 64
// entries, the exact re-run of a step that outgrew its capacities); tests/test_fused_gpu.py compares the two word by word.
#pragma once

enum { CELL_STRIDE = 8 };   // ints between the fill counts of two grid cells (four cells to a 128-byte line: the appends are atomics)
// ---- kf_front_a ----------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) kf_front_a(int ncells, int ntiles_ext, Floes F, Params P, Work W) {
  const int e = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
  for (int c = e; c < ncells; c += stride) W.cell_cnt[CELL_STRIDE * c] = 0;
  for (int t = e; t < ntiles_ext; t += stride) W.row_agg[(size_t)AGG_STRIDE * t] = 0;
  if (e < 64) W.prow_part[(size_t)AGG_STRIDE * e] = 0;
  if (e < W.cap_ext) { ext_clear(e, W); marks_clear(e, F); }
  int fx = 0, fy = 0;
  if (e < F.n) {
    ext_init(e, F, W);
    if (P.PERIODIC) {
      fx = ghost_flag_of(e, F.Xi[e], 0, F, P, W);
      fy = ghost_flag_of(e, F.Yi[e], 1, F, P, W);
      W.flag[e] = fx | (fy << 1);
    }
  }
  if (e >= F.nb && e < F.n) mc_precheck(e, F, P, W);
  if (P.PERIODIC && (e & ~63) < F.n) {                // (whole waves: a tile is a wave)
    const unsigned long long mx = __ballot(fx != 0), my = __ballot(fy != 0);
    if ((threadIdx.x & 63) == 0) {
      const int t = e >> 6;
      W.gh_mask[2 * t] = mx; W.gh_mask[2 * t + 1] = my;
      W.gh_agg[t] = (unsigned long long)__popcll(mx) | ((unsigned long long)__popcll(my) << 21) | ((unsigned long long)__popcll(mx & my) << 42);
    }
  }
}

// ---- kf_front_b ----------------------------------------------------------------------------------------------------
// entry e into its cell of the one-level grid (fixed slots per cell)
__device__ inline int item_aux(int num, int nv);
__device__ inline void cell_insert(int e, int s, double x, double y, const Floes &F, Work &W) {
  if (!(F.alive[s] && e >= F.nb && is_present(s, W))) return;       // in_grid
  const int c = cell_at(0, x, y, W);
  const int k = atomicAdd(&W.cell_cnt[CELL_STRIDE * c], 1);
  if (k < W.cell_cap) {
    CellItem it; it.x = x; it.y = y; it.r = F.rmax[s]; it.e = e;
    it.pad = item_aux(e == s ? s + 1 : -(s + 1), F.vert_off[s + 1] - F.vert_off[s]);   // (FloeNum: negative for a ghost; ring length)
    W.cell_item[(size_t)c * W.cell_cap + k] = it;
  } else {
    atomicAdd(&W.counters[C_ERR_CELL], 1ULL); atomicMax(&W.counters[C_NEED_CELL], (unsigned long long)(k + 1));
  }
}
// record and staged ring of entry e (floe s at (ox, oy)) by the 16 lanes of a group, lane l: k_ext_records' body
__device__ inline void ext_record_lanes(int e, int s, double ox, double oy, int l, const Floes &F, Work &W) {
  const int v0 = F.vert_off[s], n = F.vert_off[s + 1] - v0;
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
}
// One workgroup per tile of 64 floes, 16 lanes per floe.  Ghost slots as ScanGhosts places them: x ghost of k at N + px,
// y ghost of k at N + GX + py, y ghost of k's x ghost at N + GX + GYreal + pxy, the p's = exclusive counts over the
// floes before k.
__global__ void __launch_bounds__(1024) kf_front_b(int ntiles, Floes F, Params P, Work W) {
  __shared__ unsigned long long red[2][16];
  const int w = blockIdx.x, t = threadIdx.x;
  unsigned long long before = 0, all = 0;
  if (P.PERIODIC) {
    for (int u = t; u < ntiles; u += 1024) { const unsigned long long a = W.gh_agg[u]; all += a; if (u < w) before += a; }
    for (int d = 32; d > 0; d >>= 1) { before += __shfl_xor(before, d); all += __shfl_xor(all, d); }
    if ((t & 63) == 0) { red[0][t >> 6] = before; red[1][t >> 6] = all; }
    __syncthreads();
    before = 0; all = 0;
    for (int q = 0; q < 16; ++q) { before += red[0][q]; all += red[1][q]; }
  }
  const int N = F.n;
  const unsigned FLD = (1u << 21) - 1;
  const int GX = (int)(all & FLD), GYr = (int)((all >> 21) & FLD), GXY = (int)((all >> 42) & FLD);
  if (P.PERIODIC && w == ntiles - 1 && t == 0) {
    W.counters[C_GX] = (unsigned long long)GX; W.counters[C_GY] = (unsigned long long)(GYr + GXY);
    W.counters[C_NEXT] = (unsigned long long)(N + GX + GYr + GXY);
  }
  const int f = 64 * w + (t >> 4), l = t & 15;
  if (f >= N) return;
  const double ox = F.Xi[f], oy = F.Yi[f];
  int gx = -1, gy = -1, gxy = -1;
  double gxx = ox, gyy = oy;
  if (P.PERIODIC) {
    const unsigned long long mx = W.gh_mask[2 * w], my = W.gh_mask[2 * w + 1];
    const int bit = f & 63;
    const unsigned long long below = (1ULL << bit) - 1ULL;
    const bool fx = (mx >> bit) & 1ULL, fy = (my >> bit) & 1ULL;
    if (fx) { gx = N + (int)(before & FLD) + __popcll(mx & below); gxx = ox - 2 * P.Lx * sgn(ox); }
    if (fy) { gy = N + GX + (int)((before >> 21) & FLD) + __popcll(my & below); gyy = oy - 2 * P.Ly * sgn(oy); }
    if (fx && fy) gxy = N + GX + GYr + (int)((before >> 42) & FLD) + __popcll(mx & my & below);
    if (gx >= W.cap_ext) gx = -1;       // (cannot happen: cap_ext = 4 n)
    if (gy >= W.cap_ext) gy = -1;
    if (gxy >= W.cap_ext || gx < 0) gxy = -1;
    if (l == 0) {
      const int nvf = F.vert_off[f + 1] - F.vert_off[f], num = -(f + 1);
      if (gx >= 0) { W.ext_src[gx] = f; W.ext_nv[gx] = nvf; W.ext_x[gx] = gxx; W.ext_y[gx] = oy; W.ext_num[gx] = num; W.ext_parent[gx] = f; W.xg_of[f] = gx; }
      if (gy >= 0) { W.ext_src[gy] = f; W.ext_nv[gy] = nvf; W.ext_x[gy] = ox; W.ext_y[gy] = gyy; W.ext_num[gy] = num; W.ext_parent[gy] = f; W.yg_of[f] = gy; }
      if (gxy >= 0) { W.ext_src[gxy] = f; W.ext_nv[gxy] = nvf; W.ext_x[gxy] = gxx; W.ext_y[gxy] = gyy; W.ext_num[gxy] = num; W.ext_parent[gxy] = gx; W.yg_of[gx] = gxy; }
    }
  }
  ext_record_lanes(f, f, ox, oy, l, F, W);
  if (gx >= 0) ext_record_lanes(gx, f, gxx, oy, l, F, W);
  if (gy >= 0) ext_record_lanes(gy, f, ox, gyy, l, F, W);
  if (gxy >= 0) ext_record_lanes(gxy, f, gxx, gyy, l, F, W);
  // (one lane each: the four appends go out side by side)
  if (l == 0) cell_insert(f, f, ox, oy, F, W);
  if (l == 1 && gx >= 0) cell_insert(gx, f, gxx, oy, F, W);
  if (l == 2 && gy >= 0) cell_insert(gy, f, ox, gyy, F, W);
  if (l == 3 && gxy >= 0) cell_insert(gxy, f, gxx, gyy, F, W);
}

// ---- kf_broad ------------------------------------------------------------------------------------------------------
// A wave per entry of the extended list (real floes from Nb on, then the ghosts): k_broad_wave's body on cells with fixed
// slots, every item carrying its FloeNum and ring length (CellItem::pad, item_aux), so that nothing about a partner is
// loaded after its item.  A ghost's `mems` starts from the finished list of the floe it is an image of (:94-99), and an
// entry is already listed there exactly when its real floe k is a candidate of that root -- k > root, both in the grid,
// sqrt(dx^2 + dy^2) < r_root + r_k (:103; the root's own filter never drops a real floe: nothing earlier in its list
// carries that number) -- which this wave works out from the two floes directly: the ghosts do not wait for their roots.
__device__ inline int item_aux(int num, int nv) { return (int)(((unsigned)(nv < 1023 ? nv : 1023) << 22) | ((unsigned)num & 0x3fffffu)); }
__device__ inline int aux_num(int aux) { return (aux << 10) >> 10; }
__device__ inline int aux_nv(int aux) { return (int)((unsigned)aux >> 22); }
struct BroadWave { int hit[64], haux[64], sj[64], saux[64], relf[64], s_end[9], s_c0[9], s_cn[9]; double hr[64], sr[64]; };
__device__ inline void wave_sync() {   // LDS writes of the wave's lanes before it are visible to its reads after it (team.hpp)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// one entry by one wave -> its pair count | pair vertices << 24 (0 for an entry without a list)
__device__ inline unsigned long long broad_entry(int i, BroadWave &L, const Floes &F, const Params &P, Work &W) {
  int *hit = L.hit, *haux = L.haux, *sj = L.sj, *saux = L.saux, *relf = L.relf, *s_end = L.s_end, *s_c0 = L.s_c0, *s_cn = L.s_cn;
  double *hr = L.hr, *sr = L.sr;
  const int lane = threadIdx.x & 63;
  if (i < F.nb) return 0ULL;
  const bool ghost = i >= F.n;
  // (an entry beyond this step's list reads stale words of the arrays and leaves)
  const int N = dev_next(W);
  const int si = ghost ? W.ext_src[i] : i;
  const double xi = ghost ? W.ext_x[i] : F.Xi[i], yi = ghost ? W.ext_y[i] : F.Yi[i];
  const int ni = W.ext_nv[i];
  if (i >= N) return 0ULL;
  const double ri = F.rmax[si];
  const int alive_i = F.alive[si];
  const bool present_i = is_present(si, W);
  const double rx = F.Xi[si], ry = F.Yi[si];          // (a ghost's root: the floe it copies)
  int H = 0;
  if (alive_i && !(xi != xi) && P.COLLISION && present_i) {
    const int gx0 = cell_coord(xi, W.lvl_x0[0], W.lvl_cs[0], W.lvl_cx[0]), gy0 = cell_coord(yi, W.lvl_y0[0], W.lvl_cs[0], W.lvl_cy[0]);
    int c0 = 0, cn = 0;
    if (lane < 9) {
      const int gx = gx0 + (lane % 3) - 1, gy = gy0 + (lane / 3) - 1;
      if (gx >= 0 && gx < W.lvl_cx[0] && gy >= 0 && gy < W.lvl_cy[0]) {
        const int c = W.lvl_base[0] + gy * W.lvl_cx[0] + gx;
        cn = W.cell_cnt[CELL_STRIDE * c]; if (cn > W.cell_cap) cn = W.cell_cap;
        c0 = c * W.cell_cap;
      }
    }
    int pre = cn;
    for (int d = 1; d < 16; d <<= 1) { const int v = __shfl_up(pre, d); if (lane >= d) pre += v; }
    if (lane < 9) { s_end[lane] = pre; s_c0[lane] = c0; s_cn[lane] = cn; }
    wave_sync();
    const int T = s_end[8];
    for (int base = 0; base < T; base += 64) {
      const int t = base + lane;
      bool ok = false;
      CellItem it; it.e = 0; it.pad = 0; it.r = 0;
      if (t < T) {
        int cell = 0;
        for (int c = 0; c < 8; ++c) if (t >= s_end[c]) cell = c + 1;
        it = W.cell_item[s_c0[cell] + (t - (s_end[cell] - s_cn[cell]))];
        if (it.e > i) {
          const double dx = xi - it.x, dy = yi - it.y, R = ri + it.r, d2 = dx * dx + dy * dy;
          ok = !(d2 > R * R * 1.000001) && (::sqrt(d2) < R);       // :103
        }
      }
      const unsigned long long m = __ballot(ok);
      if (ok) { const int r = H + __popcll(m & ((1ULL << lane) - 1ULL)); if (r < 64) { hit[r] = it.e; haux[r] = it.pad; hr[r] = it.r; } }
      H += __popcll(m);
    }
  }
  wave_sync();
  if (H > W.pot_cap || H > 64) {
    if (lane == 0) { atomicAdd(&W.counters[C_ERR_POT], 1ULL); atomicMax(&W.counters[C_NEED_POT], (unsigned long long)H); }
    H = 0;
  }
  const int mine = lane < H ? hit[lane] : 0x7fffffff;
  int rank = 0;
  for (int k = 0; k < H; ++k) rank += (hit[k] < mine) ? 1 : 0;
  if (lane < H) { sj[rank] = mine; saux[rank] = haux[lane]; sr[rank] = hr[lane]; }
  wave_sync();
  const int j = lane < H ? sj[lane] : 0;
  const int aux = lane < H ? saux[lane] : 0;
  const int num = lane < H ? aux_num(aux) : 0;        // FloeNum of the entry: negative for a ghost
  const int key = iabs(num);
  bool keep = lane < H;
  if (P.PERIODIC) {
    bool dup = false;                                // an earlier entry IS the floe this one is an image of
    for (int k = 0; k < H; ++k) { const int nk = __shfl(num, k); if (k < lane && lane < H && nk == key) dup = true; }
    if (ghost && lane < H && key >= 1) {             // ... or that floe is on the root's list
      const int k = key - 1;
      if (k > si && k >= F.nb && si >= F.nb && F.alive[k] && is_present(k, W) && alive_i && !(rx != rx) && present_i) {
        const double dx = rx - F.Xi[k], dy = ry - F.Yi[k], R = ri + F.rmax[k], d2 = dx * dx + dy * dy;
        if (!(d2 > R * R * 1.000001) && (::sqrt(d2) < R)) dup = true;
      }
    }
    if (keep && dup) {
      const double lim = szc::mind(2 * P.Lx, 2 * P.Ly), R = ri + sr[lane];
      keep = 2 * R > lim;
    }
  }
  const unsigned long long km = __ballot(keep);
  const int K = __popcll(km);
  const int idx = __popcll(km & ((1ULL << lane) - 1ULL));
  if (keep) W.pot_list[(size_t)i * W.pot_cap + idx] = j + 1;
  // the pairs of this entry (pair_count), which of its candidates they are, the ring vertices before each
  const bool rel = keep && pair_relevant(i, j, W);
  int nvj = 0;
  if (rel) { const int c = aux_nv(aux); nvj = ni + (c < 1023 ? c : W.ext_nv[j]); }
  if (rel) (void)__hip_atomic_fetch_add(&W.inc_pre[j], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int inc = nvj;
  for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(inc, d); if (lane >= d) inc += v; }
  if (keep) W.pot_nvpre[(size_t)i * W.pot_cap + idx] = inc - nvj;
  long long nv = __shfl(inc, 63);
  unsigned long long relm;
  if (W.owned) {
    if (keep) relf[idx] = rel ? 1 : 0;
    wave_sync();
    relm = __ballot(lane < K && relf[lane] != 0);
  } else relm = (K >= 64) ? ~0ULL : ((1ULL << K) - 1ULL);
  int cnt = __popcll(__ballot(rel));
  if (has_wall_pair(i, N, F, P, W)) { ++cnt; nv += ni + 4; }
  if (lane == 0) { W.pot_cnt[i] = K; W.pair_cnt[i] = cnt; W.pair_nv[i] = nv; W.rel_mask[i] = relm; }
  return (unsigned long long)cnt | ((unsigned long long)nv << 24);
}
// sixteen waves, sixteen consecutive entries per workgroup; the group's pair totals are one plain store (no atomics: the
// L2 serves the atomics of a cache line one by one, and 1e4 of them on the few lines of such sums took 12 us)
__global__ void __launch_bounds__(1024) kf_broad(Floes F, Params P, Work W) {
  __shared__ BroadWave L[16];
  __shared__ unsigned long long tot[16];
  const int w = threadIdx.x >> 6;
  const unsigned long long mine = broad_entry(16 * blockIdx.x + w, L[w], F, P, W);
  if ((threadIdx.x & 63) == 0) tot[w] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long sum = 0;
    for (int q = 0; q < 16; ++q) sum += tot[q];
    W.pair_agg[blockIdx.x] = sum;
  }
}


// ---- kf_pairs ------------------------------------------------------------------------------------------------------
__device__ inline int nth_set_bit(unsigned long long m, int q) {   // position of the q-th (0-based) set bit of m
  int pos = 0;
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const int c = __popcll(m & ((1ULL << s) - 1ULL));
    if (q >= c) { q -= c; m >>= s; pos += s; }
  }
  return pos;
}
// One workgroup per tile of 64 entries: ScanPairs + pair_expand + pair_classify.  The pairs of entry i are its relevant
// candidates in list order, then the wall (has_wall_pair) -- pair p = pair_base[i] + rank.
__global__ void __launch_bounds__(512) kf_pairs(int prev_n, unsigned long long scratch_bump, Floes F, Params P, Work W) {
  __shared__ unsigned long long red[8];
  __shared__ long long s_base[65], s_nvb[64];
  __shared__ int s_cnt[64];
  const int T = blockIdx.x, t = threadIdx.x, lane = t & 63;
  const int N = dev_next(W);
  const int ntiles = (N + 63) >> 6;
  if (T == 0 && t == 0) W.counters[C_SCR_BUMP] = scratch_bump;
  if (T >= ntiles) return;
  unsigned long long before = 0;
  for (int u = t; u < 4 * T; u += 512) before += W.pair_agg[u];      // (kf_broad's sums: four groups of 16 entries to a tile)
  for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d);
  if (lane == 0) red[t >> 6] = before;
  __syncthreads();
  before = 0;
  for (int q = 0; q < 8; ++q) before += red[q];
  const long long base_p = (long long)(before & 0xffffffULL), base_nv = (long long)(before >> 24);
  if (t < 64) {
    const int e = 64 * T + t;
    long long cnt = 0, nv = 0;
    if (e < N) { cnt = W.pair_cnt[e]; nv = W.pair_nv[e]; pair_room(e, W); }
    long long ic = cnt, iv = nv;
    for (int d = 1; d < 64; d <<= 1) { const long long a = __shfl_up(ic, d), b = __shfl_up(iv, d); if (lane >= d) { ic += a; iv += b; } }
    s_base[t] = ic - cnt; s_nvb[t] = iv - nv; s_cnt[t] = (int)cnt;
    if (t == 63) s_base[64] = ic;
    if (e < N) { W.pair_base[e] = base_p + ic - cnt; W.nv_base[e] = base_nv + iv - nv; }
    if (T == ntiles - 1 && t == 63) {
      W.counters[C_NPAIRS] = (unsigned long long)(base_p + ic); W.pair_base[N] = base_p + ic;
      W.counters[C_NVTOT] = (unsigned long long)(base_nv + iv); W.nv_base[N] = base_nv + iv;
    }
  }
  __syncthreads();
  const int tp = (int)s_base[64];
  const bool walls = W.wall_list != nullptr;
  long long teamv = 0;                                  // ring vertices of the pairs the team pass will take (its algorithmic bytes: bench.py)
  for (int pl0 = 0; pl0 < tp; pl0 += 512) {
    const int pl = pl0 + t;
    int cls = PC_DONE;
    i64 p = 0;
    int team_nv = 0;
    if (pl < tp) {
      int lo = 0;                                     // the entry whose range holds pl: the last one with s_base <= pl
#pragma unroll
      for (int s = 32; s >= 1; s >>= 1) if (lo + s < 64 && s_base[lo + s] <= pl) lo += s;
      const int i = 64 * T + lo, q = pl - (int)s_base[lo];
      const unsigned long long rm = W.rel_mask[i];
      int j; i64 nvo;
      if (q < __popcll(rm)) {
        const int k = nth_set_bit(rm, q);
        j = W.pot_list[(size_t)i * W.pot_cap + k] - 1;
        nvo = base_nv + s_nvb[lo] + W.pot_nvpre[(size_t)i * W.pot_cap + k];
      } else {                                        // the wall, after the floe's last candidate
        j = WALL;
        nvo = base_nv + s_nvb[lo] + (W.pair_nv[i] - (W.ext_nv[i] + 4));
      }
      p = base_p + pl;
      if (p < W.cap_pairs) {
        W.pair_i[p] = i; W.pair_j[p] = j; W.pair_nvoff[p] = nvo;
        cls = pair_classify_ij(p, i, j, nvo, prev_n, P, W);
        team_nv = j >= 0 ? W.ext_nv[i] + W.ext_nv[j] : 0;
      }
    }
    if (cls == PC_LIVE || cls == PC_COLD || cls == PC_WIDE) teamv += team_nv;
    // the lists are appended to a wave at a time (k_pair_classify)
    unsigned long long mine = 0, m_own = 0;
#pragma unroll
    for (int c = PC_LIVE; c <= PC_GENERAL; ++c) {
      const unsigned long long m = __ballot(cls == c);
      if (lane == c) mine = m;
      if (cls == c) m_own = m;
    }
    unsigned long long base = 0;
    if (lane >= PC_LIVE && lane <= PC_GENERAL && mine) {
      const int ctr = lane == PC_LIVE ? C_LIVE : lane == PC_WIDE ? C_WIDE : lane == PC_COLD ? C_COLD : lane == PC_GENERAL ? (walls ? C_WALLS : C_RETRY) : C_TEAM;
      base = atomicAdd(&W.counters[ctr], (unsigned long long)__popcll(mine));
      if (lane == PC_GENERAL && !walls) atomicAdd(&W.counters[C_WALLS], (unsigned long long)__popcll(mine));
    }
    const int src = (cls >= PC_LIVE && cls <= PC_GENERAL) ? cls : 0;
    base = __shfl(base, src);
    if (cls >= PC_LIVE && cls <= PC_GENERAL)
      (cls == PC_LIVE ? W.live_list : cls == PC_WIDE ? W.wide_list : cls == PC_COLD ? W.cold_list : cls == PC_GENERAL ? (walls ? W.wall_list : W.retry_list) : W.team_list)[base + __popcll(m_own & ((1ULL << lane) - 1ULL))] = (int)p;
  }
  for (int d = 32; d > 0; d >>= 1) teamv += __shfl_xor(teamv, d);
  __syncthreads();
  if (lane == 0) red[t >> 6] = (unsigned long long)teamv;
  __syncthreads();
  if (t == 0) { unsigned long long v = 0; for (int q = 0; q < 8; ++q) v += red[q]; if (v) atomicAdd(&W.counters[C_TEAMV], v); }
}

// ---- kf_rows ---------------------------------------------------------------------------------------------------------
// The contact rows of the step in one launch: k_incoming + k_row_count + ScanRows + k_row_fill.  The pairs have
// registered their rows with both floes as they completed (pair_register: counts per entry and per tile, mirror
// lists), so a tile of 64 entries knows where its rows start from the tile sums before it.  16 lanes per entry: lane l
// takes the entry's l-th ITEM -- its own pairs in list order, then the pairs that name it as partner in ascending order
// of their source floe (:187) -- loads the pair's words and rows side by side with the other lanes, and the sums of
// :221-263 are then formed in item order from registers (shuffles), the same additions in the same order as the
// one-thread loop of row_fill.  More than 16 items or more than two rows in a pair: one lane runs row_fill itself.
// publish: the workgroups work the go rule out for themselves and the first one publishes the counters (k_incoming_go).
__global__ void __launch_bounds__(1024) kf_rows(int publish, unsigned long long *host_copy, GoRule rule, Floes F, Params P, Work W) {
  __shared__ unsigned long long red[16];
  __shared__ long long s_off[64];
  const unsigned long long prows = publish ? pair_rows_total(W.counters, W.prow_part) : 0ULL;    // (every wave for itself)
  const bool go = publish ? go_by_rule(W.counters, rule, prows) : (W.counters[C_GO] != 0);
  const int T = blockIdx.x, t = threadIdx.x;
  if (publish && T == 0 && t < (int)C_SLOTS) {
    const int c = t * C_STRIDE;
    host_copy[c] = (c == C_GO) ? (go ? 1ULL : 0ULL) : (c == C_PAIRROWS) ? prows : W.counters[c];
    if (c == C_GO) W.counters[C_GO] = go ? 1ULL : 0ULL;
  }
  if (!go) return;
  const int N = dev_next(W);
  const int ntiles = (N + 63) >> 6;
  if (T >= ntiles) return;
  unsigned long long before = 0;
  for (int u = t; u < T; u += 1024) before += W.row_agg[(size_t)AGG_STRIDE * u];
  for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d);
  if ((t & 63) == 0) red[t >> 6] = before;
  __syncthreads();
  before = 0;
  for (int q = 0; q < 16; ++q) before += red[q];
  if (t < 64) {
    const int e = 64 * T + t;
    const long long cnt = e < N ? W.row_cnt[e] : 0;
    long long inc = cnt;
    for (int d = 1; d < 64; d <<= 1) { const long long a = __shfl_up(inc, d); if (t >= d) inc += a; }
    s_off[t] = (long long)before + inc - cnt;
    if (e < N) W.row_off[e] = (i64)before + inc - cnt;
    if (T == ntiles - 1 && t == 63) { W.counters[C_NROWS] = before + (unsigned long long)inc; W.row_off[N] = (i64)before + inc; }
  }
  __syncthreads();
  const int e = 64 * T + (t >> 4), l = t & 15;
  if (e >= N) return;
  const int N0 = F.n;
  const int K = W.pair_cnt[e];
  const i64 pb = W.pair_base[e];
  int ninc = W.inc_cnt[e]; if (ninc > W.pot_cap) ninc = W.pot_cap;
  const int items = K + ninc;
  const double x = W.ext_x[e], y = W.ext_y[e];
  const unsigned gsh = (threadIdx.x & 63) & ~15u;            // this group's lanes within the wave's ballots
  // lane l: item l as it lies in memory (own pairs in order, mirrored pairs unsorted)
  i64 p = -1; int j = 0, fl = 0, nr = 0, src = 0x7fffffff; i64 ro = 0;
  const bool own = l < K, mir = l >= K && l < items;
  if (items <= 16) {
    if (own) { p = pb + l; j = W.pair_j[p]; fl = W.pair_flag[p]; nr = (fl & 4) ? W.pair_nrows[p] : 0; ro = W.pair_rowoff[p]; }
    else if (mir) { p = W.inc_list[(size_t)e * W.pot_cap + (l - K)]; src = W.pair_i[p]; nr = W.pair_nrows[p]; ro = W.pair_rowoff[p]; fl = 4; }
  }
  const bool serial = items > 16 || ((__ballot(nr > 2) >> gsh) & 0xffffULL) != 0;
  if (serial) {
    if (l == 0) {
      int *inc = W.inc_list + (size_t)e * W.pot_cap;          // ascending source floe, in place (row_count's insertion sort)
      for (int a = 1; a < ninc; ++a) {
        const int v = inc[a], sv = W.pair_i[v]; int b = a;
        while (b > 0 && W.pair_i[inc[b - 1]] > sv) { inc[b] = inc[b - 1]; --b; }
        inc[b] = v;
      }
      row_fill(e, N, F, P, W);
      transfer_collect(e, F, W);
    }
    return;
  }
  // position of the item in the reference's row order: own pairs keep theirs, mirrored ones are ranked by source floe
  int pos = l;
  if (mir) {
    int rank = 0;
    for (int b = 0; b < 16; ++b) { const int sb = __shfl(src, b, 16); if (b >= K && b < items && sb < src) ++rank; }
    pos = K + rank;
  }
  int inv = 0;                                               // the lane that holds the item of position l
  for (int b = 0; b < 16; ++b) { const int pq = __shfl(pos, b, 16); if (pq == l) inv = b; }
  // ... brought to lane l: from here on lane order is row order
  {
    const int p_lo = __shfl((int)(p & 0xffffffffLL), inv, 16), p_hi = __shfl((int)(p >> 32), inv, 16);
    p = ((i64)p_hi << 32) | (unsigned)p_lo;
    const int r_lo = __shfl((int)(ro & 0xffffffffLL), inv, 16), r_hi = __shfl((int)(ro >> 32), inv, 16);
    ro = ((i64)r_hi << 32) | (unsigned)r_lo;
    j = __shfl(j, inv, 16); fl = __shfl(fl, inv, 16); nr = __shfl(nr, inv, 16); src = __shfl(src, inv, 16);
  }
  const bool have = l < items;
  if (!have) nr = 0;
  // the rows of my item: up to two, read side by side
  double fx0 = 0, fy0 = 0, px0 = 0, py0 = 0, o0 = 0, fx1 = 0, fy1 = 0, px1 = 0, py1 = 0, o1 = 0;
  if (nr > 0) {
    const double *pr = (const double *)W.scratch + ro;
    fx0 = pr[0]; fy0 = pr[1]; px0 = pr[2]; py0 = pr[3]; o0 = pr[4];
    if (nr > 1) { fx1 = pr[5]; fy1 = pr[6]; px1 = pr[7]; py1 = pr[8]; o1 = pr[9]; }
  }
  double w0;
  if (own) {
    if (j == WALL) {
      if (absd(py0) == P.Ly) fx0 = 0;                          // :160-162
      if (absd(px0) == P.Lx) fy0 = 0;                          // :163-165
      if (absd(py1) == P.Ly) fx1 = 0;
      if (absd(px1) == P.Lx) fy1 = 0;
      w0 = SZC_INF;
    } else w0 = (double)(j + 1);
  } else { fx0 = -fx0; fy0 = -fy0; fx1 = -fx1; fy1 = -fy1; w0 = (double)(src + 1); }
  const double tq0 = (px0 - x) * fy0 - (py0 - y) * fx0, tq1 = (px1 - x) * fy1 - (py1 - y) * fx1;
  int rb = nr;                                               // rows before my item (exclusive scan over the lanes)
  for (int d = 1; d < 16; d <<= 1) { const int v = __shfl_up(rb, d, 16); if (l >= d) rb += v; }
  rb -= nr;
  {
    const i64 r = s_off[t >> 4] + rb;
    if (nr > 0 && r < W.cap_rows) { double *w = W.rows + 7 * r; w[0] = w0; w[1] = fx0; w[2] = fy0; w[3] = px0; w[4] = py0; w[5] = tq0; w[6] = o0; }
    if (nr > 1 && r + 1 < W.cap_rows) { double *w = W.rows + 7 * (r + 1); w[0] = w0; w[1] = fx1; w[2] = fy1; w[3] = px1; w[4] = py1; w[5] = tq1; w[6] = o1; }
  }
  // the sums, item by item (:221-239, :251-263; OverlapArea :137,:168,:199)
  double so = 0;
  if (own && nr > 0) { so += o0; if (nr > 1) so += o1; }
  double sfx = 0, sfy = 0, stq = 0, ov = 0;
  for (int a = 0; a < 16; ++a) {
    if (a >= items) break;                                   // (uniform over the group)
    const int na = __shfl(nr, a, 16);
    const double ax0 = __shfl(fx0, a, 16), ay0 = __shfl(fy0, a, 16), at0 = __shfl(tq0, a, 16);
    const double ax1 = __shfl(fx1, a, 16), ay1 = __shfl(fy1, a, 16), at1 = __shfl(tq1, a, 16);
    const double aso = __shfl(so, a, 16), ao0 = __shfl(o0, a, 16), ao1 = __shfl(o1, a, 16);
    if (na > 0) { sfx += ax0; sfy += ay0; stq += at0; }
    if (na > 1) { sfx += ax1; sfy += ay1; stq += at1; }
    if (a < K) { if (na > 0) ov = aso + ov; }
    else { if (na > 0) ov = ov + ao0; if (na > 1) ov = ov + ao1; }
  }
  // kill / transfer marks of the own pairs, the last one in list order that sets each (:138-145)
  const bool mark = own && !(fl & 4) && (fl & 3) && j != WALL && e + 1 > F.nb;
  const bool markA = mark && e + 1 <= N0 && (fl & 1), markB = mark && !markA && j + 1 <= N0;
  const unsigned mk = (unsigned)((__ballot(markA || markB) >> gsh) & 0xffffULL), mt = (unsigned)((__ballot(markA) >> gsh) & 0xffffULL);
  int kill = 0;
  if (mk) { const int la = 31 - __clz(mk); const int jj = __shfl(j, la, 16), isA = __shfl((int)markA, la, 16); kill = isA ? e + 1 : jj + 1; }
  int transfer_to = 0;
  if (mt) { const int la = 31 - __clz(mt); transfer_to = __shfl(j, la, 16) + 1; }
  if (l == 0) {
    W.ov_ext[e] = ov; W.cf_ext[2 * e] = sfx; W.cf_ext[2 * e + 1] = sfy; W.ct_ext[e] = stq;
    if (mk) F.kill[e] = kill;
    if (mt) F.transfer[e] = transfer_to;
    centre_check(e, x, y, F, P, W);
    if (iabs(kill - (e + 1)) > 0 && kill > 0) atomic_max(&W.tmax[kill - 1], e + 1);     // transfer_collect
  }
}
