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

// Diagnostic build only (make -C subzero_amd/csrc diag_fused): wall-clock ticks (100 MHz) at marked points of the fused
// kernels, first thread of every workgroup; read with szDebugFusedClocks.  Not in the product build.
#if defined(SZ_FUSED_CLOCKS)
#define SZ_FC_KERNELS 8
#define SZ_FC_BLOCKS 1024
#define SZ_FC_SLOTS 12
__device__ unsigned long long sz_fused_clock[SZ_FC_KERNELS * SZ_FC_BLOCKS * SZ_FC_SLOTS];
#define FCLK(kern, slot) do { if (threadIdx.x == 0 && blockIdx.x < SZ_FC_BLOCKS) sz_fused_clock[((kern) * SZ_FC_BLOCKS + blockIdx.x) * SZ_FC_SLOTS + (slot)] = wall_clock64(); } while (0)
#else
#define FCLK(kern, slot) do { } while (0)
#endif
enum { HINT_ROWS_SHIFT = 28, HINT_PARTNER_MASK = (1 << 28) - 1 };   // W.hint: partner entry + 1 in the low bits (entries < 2^23; all ones: the wall), the pair's contact rows (up to 7) above
enum { CELL_STRIDE = 8 };   // ints between the fill counts of two grid cells (four cells to a 128-byte line: the appends are atomics)
// ---- kf_front_a ----------------------------------------------------------------------------------------------------
// (ahead: queued behind the step BEFORE the one it belongs to, szStepParams.flags -- it runs only if that step ran to its
// end: a step that left work for the host still needs the arrays this kernel clears)
__global__ void __launch_bounds__(256) kf_front_a(int ncells, int ntiles_ext, int ahead, Floes F, Params P, Work W) {
  FCLK(5, 0);
  if (ahead && !W.counters[C_AHEAD]) return;
  const int e = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
  for (int c = e; c < ncells; c += stride) W.cell_cnt[CELL_STRIDE * c] = 0;
  for (int t = e; t < ntiles_ext; t += stride) W.row_agg[(size_t)AGG_STRIDE * t] = 0;
  if (e < 64) W.prow_part[(size_t)AGG_STRIDE * e] = 0;
  if (e < W.cap_ext) ext_clear(e, W);           // (the kill / transfer marks of the step before: kf_pairs -- the caller may still read them)
  FCLK(5, 1);
  int fx = 0, fy = 0;
  if (e < F.n) {
    ext_init(e, F, W);
    if (P.PERIODIC) {
      // ghost_flag_of for both dimensions in one pass over the ring, eight vertices' loads in flight at a time (a floe at
      // the box edge used to walk its ring twice, a memory round trip per vertex: the launch waited 4 us for those floes)
      const double xe = F.Xi[e], ye = F.Yi[e], re = F.rmax[e];
      const bool nearx = !(absd(xe) + re < P.Lx * (1 - 1e-9)), neary = !(absd(ye) + re < P.Ly * (1 - 1e-9));
      if (is_present(e, W) && (nearx || neary)) {
        const int v0 = F.vert_off[e], nv = F.vert_off[e + 1] - v0;
        double mx = 0, my = 0;
        for (int k0 = 0; k0 < nv; k0 += 8) {
          double vx[8], vy[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) { const int k = k0 + q < nv ? k0 + q : nv - 1; vx[q] = F.cax[v0 + k]; vy[q] = F.cay[v0 + k]; }
#pragma unroll
          for (int q = 0; q < 8; ++q) { mx = szc::maxd(mx, absd(vx[q] + xe)); my = szc::maxd(my, absd(vy[q] + ye)); }
        }
        const int al = F.alive[e];
        fx = (nearx && al && mx > P.Lx) ? 1 : 0;
        fy = (neary && al && my > P.Ly) ? 1 : 0;
      }
      W.flag[e] = fx | (fy << 1);
    }
  }
  FCLK(5, 2);
  if (e >= F.nb && e < F.n) mc_precheck(e, F, P, W);
  FCLK(5, 3);
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
// entry e (floe s: alive, radius r, FloeNum / ring length word aux) into its cell of the one-level grid (fixed slots per cell)
__device__ inline int item_aux(int num, int nv);
__device__ inline void cell_insert(int e, int s, double x, double y, int alive, double r, int nv, const Floes &F, Work &W) {
  if (!(alive && e >= F.nb && is_present(s, W))) return;            // in_grid
  const int c = cell_at(0, x, y, W);
  const int k = atomicAdd(&W.cell_cnt[CELL_STRIDE * c], 1);
  if (k < W.cell_cap) {
    CellItem it; it.x = x; it.y = y; it.r = r; it.e = e;
    it.pad = item_aux(e == s ? s + 1 : -(s + 1), nv);               // (FloeNum: negative for a ghost; ring length)
    W.cell_item[(size_t)c * W.cell_cap + k] = it;
  } else {
    atomicAdd(&W.counters[C_ERR_CELL], 1ULL); atomicMax(&W.counters[C_NEED_CELL], (unsigned long long)(k + 1));
  }
}
// record and staged ring of entry e (a floe of n vertices from v0, at (ox, oy)) by the 16 lanes of a group, lane l:
// k_ext_records' body.  (cxl, cyl): c_alpha of vertex l (l < n), scal: the floe's scalar that is word l of the record --
// loaded by the caller, once for the floe and its ghosts.
__device__ inline void ext_record_lanes(int e, int n, int v0, double ox, double oy, int l, double cxl, double cyl, double scal, const Floes &F, Work &W) {
  double x0 = SZC_INF, x1 = -SZC_INF, y0 = SZC_INF, y1 = -SZC_INF;
  if (l < n) { const double X = cxl + ox, Y = cyl + oy; x0 = x1 = X; y0 = y1 = Y; }
  for (int k = l + 16; k < n; k += 16) {
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
  if (l < nstage) { X = cxl + ox; Y = cyl + oy; ok = szc::absd(X) < safe && szc::absd(Y) < safe; }
  const unsigned long long bad = __ballot(!ok);
  const bool in_range = ((bad >> ((threadIdx.x & 63) & ~15)) & 0xffffULL) == 0;
  double *rx = W.ext_ring + 2 * (size_t)RING_SLOTS * e;
  if (l < nstage) { rx[l] = X; rx[RING_SLOTS + l] = Y; }
  double w = scal;
  switch (l) {
    case REC_X: w = ox; break; case REC_Y: w = oy; break;
    case REC_BX0: w = x0; break; case REC_BX1: w = x1; break; case REC_BY0: w = y0; break; case REC_BY1: w = y1; break;
    case REC_NSTAGE: w = (double)nstage; break; case REC_NSTORED: w = (double)n; break; case REC_INRANGE: w = in_range ? 1.0 : 0.0; break;
    default: break;
  }
  W.ext_rec[(size_t)REC_WORDS * e + l] = w;
}
// One workgroup per tile of 64 floes, 16 lanes per floe.  Ghost slots as ScanGhosts places them: x ghost of k at N + px,
// y ghost of k at N + GX + py, y ghost of k's x ghost at N + GX + GYreal + pxy, the p's = exclusive counts over the
// floes before k.  Two rounds of loads: what the lane's floe and tile own at known addresses, then the ring.
__global__ void __launch_bounds__(1024) kf_front_b(int ntiles, int ahead, Floes F, Params P, Work W) {
  __shared__ unsigned long long red[2][16];
  FCLK(4, 0);
  if (ahead && !W.counters[C_AHEAD]) return;
  const int w = blockIdx.x, t = threadIdx.x;
  const int N = F.n;
  const int f = 64 * w + (t >> 4), l = t & 15;
  const bool have = f < N;
  // (sharded step: a tile none of whose floes this rank holds has no ghosts, no records and no grid items to write; the
  // last tile stays for the totals)
  if (W.present && !__syncthreads_or(have && W.present[f] != 0) && w != ntiles - 1) return;
  const double ox = have ? F.Xi[f] : 0, oy = have ? F.Yi[f] : 0;
  const int v0 = have ? F.vert_off[f] : 0, nvf = have ? F.vert_off[f + 1] - v0 : 0;
  const int alive = have ? F.alive[f] : 0;
  const double rmax = have ? F.rmax[f] : 0;
  double scal = 0;                                     // the scalar that is word l of the floe's record
  if (have) switch (l) {
    case REC_UI: scal = F.Ui[f]; break; case REC_VI: scal = F.Vi[f]; break; case REC_KSI: scal = F.ksi_ice[f]; break;
    case REC_H: scal = F.h[f]; break; case REC_AREA: scal = F.area[f]; break; default: break;
  }
  const unsigned long long mx = P.PERIODIC ? W.gh_mask[2 * w] : 0ULL, my = P.PERIODIC ? W.gh_mask[2 * w + 1] : 0ULL;
  unsigned long long before = 0, all = 0;
  if (P.PERIODIC) for (int u = t; u < ntiles; u += 1024) { const unsigned long long a = W.gh_agg[u]; all += a; if (u < w) before += a; }
  double cxl = 0, cyl = 0;
  if (l < nvf) { cxl = F.cax[v0 + l]; cyl = F.cay[v0 + l]; }       // (second round, in flight during the reduction below)
  if (P.PERIODIC) {
    for (int d = 32; d > 0; d >>= 1) { before += __shfl_xor(before, d); all += __shfl_xor(all, d); }
    if ((t & 63) == 0) { red[0][t >> 6] = before; red[1][t >> 6] = all; }
    __syncthreads();
    before = 0; all = 0;
    for (int q = 0; q < 16; ++q) { before += red[0][q]; all += red[1][q]; }
  }
  const unsigned FLD = (1u << 21) - 1;
  const int GX = (int)(all & FLD), GYr = (int)((all >> 21) & FLD), GXY = (int)((all >> 42) & FLD);
  if (P.PERIODIC && w == ntiles - 1 && t == 0) {
    W.counters[C_GX] = (unsigned long long)GX; W.counters[C_GY] = (unsigned long long)(GYr + GXY);
    W.counters[C_NEXT] = (unsigned long long)(N + GX + GYr + GXY);
  }
  FCLK(4, 1);
  if (!have) return;
  int gx = -1, gy = -1, gxy = -1;
  double gxx = ox, gyy = oy;
  if (P.PERIODIC) {
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
      const int num = -(f + 1);
      if (gx >= 0) { W.ext_src[gx] = f; W.ext_nv[gx] = nvf; W.ext_x[gx] = gxx; W.ext_y[gx] = oy; W.ext_num[gx] = num; W.ext_parent[gx] = f; W.xg_of[f] = gx; }
      if (gy >= 0) { W.ext_src[gy] = f; W.ext_nv[gy] = nvf; W.ext_x[gy] = ox; W.ext_y[gy] = gyy; W.ext_num[gy] = num; W.ext_parent[gy] = f; W.yg_of[f] = gy; }
      if (gxy >= 0) { W.ext_src[gxy] = f; W.ext_nv[gxy] = nvf; W.ext_x[gxy] = gxx; W.ext_y[gxy] = gyy; W.ext_num[gxy] = num; W.ext_parent[gxy] = gx; W.yg_of[gx] = gxy; }
    }
  }
  FCLK(4, 2);
  // (one lane each: the four appends go out side by side, and before the records so that their round trip overlaps)
  if (l == 0) cell_insert(f, f, ox, oy, alive, rmax, nvf, F, W);
  if (l == 1 && gx >= 0) cell_insert(gx, f, gxx, oy, alive, rmax, nvf, F, W);
  if (l == 2 && gy >= 0) cell_insert(gy, f, ox, gyy, alive, rmax, nvf, F, W);
  if (l == 3 && gxy >= 0) cell_insert(gxy, f, gxx, gyy, alive, rmax, nvf, F, W);
  FCLK(4, 3);
  ext_record_lanes(f, nvf, v0, ox, oy, l, cxl, cyl, scal, F, W);
  if (gx >= 0) ext_record_lanes(gx, nvf, v0, gxx, oy, l, cxl, cyl, scal, F, W);
  if (gy >= 0) ext_record_lanes(gy, nvf, v0, ox, gyy, l, cxl, cyl, scal, F, W);
  if (gxy >= 0) ext_record_lanes(gxy, nvf, v0, gxx, gyy, l, cxl, cyl, scal, F, W);
  FCLK(4, 5);
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
  if (i < F.nb || i >= W.cap_ext) return 0ULL;
  const bool ghost = i >= F.n;
  // (spatially sharded step: a floe this rank does not hold this step has no list -- ext_clear has left its counts at 0 --
  // and its wave leaves at once: on a rank of eight, seven waves in eight)
  if (!ghost && W.present && !W.present[i]) return 0ULL;
  // (an entry beyond this step's list reads stale words of the arrays and leaves)
  const int N = dev_next(W);
  const int si = ghost ? W.ext_src[i] : i;
  const double xi = ghost ? W.ext_x[i] : F.Xi[i], yi = ghost ? W.ext_y[i] : F.Yi[i];
  const int ni = W.ext_nv[i];
  if (i >= N) return 0ULL;
  FCLK(3, 1);
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
    FCLK(3, 2);
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
  FCLK(3, 3);
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
  FCLK(3, 4);
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
  FCLK(3, 5);
  return (unsigned long long)cnt | ((unsigned long long)nv << 24);
}
// four waves, four consecutive entries per workgroup (sixteen waves of 1024 threads left the chip half empty: two such
// workgroups fill a CU's wave slots); the group's pair totals are one plain store (no atomics: the L2 serves the atomics
// of a cache line one by one, and 1e4 of them on the few lines of such sums took 12 us)
enum { BROAD_WAVES = 4 };
__global__ void __launch_bounds__(64 * BROAD_WAVES) kf_broad(Floes F, Params P, Work W) {
  __shared__ BroadWave L[BROAD_WAVES];
  __shared__ unsigned long long tot[BROAD_WAVES];
  FCLK(3, 0);
  const int w = threadIdx.x >> 6;
  const unsigned long long mine = broad_entry(BROAD_WAVES * blockIdx.x + w, L[w], F, P, W);
  if ((threadIdx.x & 63) == 0) tot[w] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long sum = 0;
    for (int q = 0; q < BROAD_WAVES; ++q) sum += tot[q];
    W.pair_agg[blockIdx.x] = sum;
  }
  FCLK(3, 6);
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
// candidates in list order, then the wall (has_wall_pair) -- pair p = pair_base[i] + rank.  The grid covers the entries
// the step may have (next_cap, GoRule); an entry beyond this step's list has no pairs (ext_clear) and costs nothing, so
// no workgroup needs the list's length -- a count another kernel has only just written, a memory round trip of its own.
// Loads are issued as early as their addresses are known: the kernel is as long as its chain of dependent round trips
// (tile sums and counts | candidates | the two entries' records and the hint | one atomic per list).
__global__ void __launch_bounds__(512) kf_pairs(int prev_n, unsigned long long scratch_bump, Floes F, Params P, Work W) {
  __shared__ unsigned long long red[8];
  __shared__ long long s_base[65], s_nvb[64];
  __shared__ int wcnt[8][8];
  __shared__ unsigned long long wbase[8];
  FCLK(1, 0);
  const int T = blockIdx.x, t = threadIdx.x, lane = t & 63;
  if (T == 0 && t == 0) W.counters[C_SCR_BUMP] = scratch_bump;
  for (int e = T * 512 + t; e < W.cap_ext; e += (int)gridDim.x * 512) marks_clear(e, F);    // :123 (the back half of the step sets them)
  // first round: the sums of the groups before this tile, and (first wave) the counts of its entries
  long long cnt = 0, nv = 0;
  const int e0 = 64 * T + t;
  if (t < 64 && e0 < W.cap_ext) { cnt = W.pair_cnt[e0]; nv = W.pair_nv[e0]; pair_room(e0, W); }
  // (a tile whose entries have no pairs -- the floes of other ranks in a sharded step, the slack of the grid -- has nothing
  // to place: pair_base is read for entries with pairs only; the last tile stays for the totals)
  if (!__syncthreads_or(cnt != 0) && T != (int)gridDim.x - 1) return;
  unsigned long long before = 0;
  for (int u = t; u < (64 / BROAD_WAVES) * T; u += 512) before += W.pair_agg[u];      // (kf_broad's sums, a workgroup of BROAD_WAVES entries each)
  for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d);
  if (lane == 0) red[t >> 6] = before;
  __syncthreads();
  before = 0;
  for (int q = 0; q < 8; ++q) before += red[q];
  FCLK(1, 1);
  const long long base_p = (long long)(before & 0xffffffULL), base_nv = (long long)(before >> 24);
  if (t < 64) {
    long long ic = cnt, iv = nv;
    for (int d = 1; d < 64; d <<= 1) { const long long a = __shfl_up(ic, d), b = __shfl_up(iv, d); if (lane >= d) { ic += a; iv += b; } }
    s_base[t] = ic - cnt; s_nvb[t] = iv - nv;
    if (t == 63) s_base[64] = ic;
    if (e0 <= W.cap_ext) { W.pair_base[e0] = base_p + ic - cnt; W.nv_base[e0] = base_nv + iv - nv; }
    if (T == (int)gridDim.x - 1 && t == 63) {      // the last tile: the totals (entries beyond the list have no pairs: pair_base[N] is the total too)
      W.counters[C_NPAIRS] = (unsigned long long)(base_p + ic); W.counters[C_NVTOT] = (unsigned long long)(base_nv + iv);
      if (e0 + 1 <= W.cap_ext) { W.pair_base[e0 + 1] = base_p + ic; W.nv_base[e0 + 1] = base_nv + iv; }
    }
  }
  __syncthreads();
  FCLK(1, 2);
  const int tp = (int)s_base[64];
  const bool walls = W.wall_list != nullptr;
  long long teamv = 0;                                  // ring vertices of the pairs the team pass will take (its algorithmic bytes: bench.py)
  for (int pl0 = 0; pl0 < tp; pl0 += 512) {
    const int pl = pl0 + t;
    int cls = PC_DONE;
    i64 p = 0;
    if (pl < tp) {
      int lo = 0;                                     // the entry whose range holds pl: the last one with s_base <= pl
#pragma unroll
      for (int s = 32; s >= 1; s >>= 1) if (lo + s < 64 && s_base[lo + s] <= pl) lo += s;
      const int i = 64 * T + lo, q = pl - (int)s_base[lo];
      const int npairs_i = (int)(s_base[lo + 1] - s_base[lo]);
      int k = q;                                      // the q-th relevant candidate (all of them unless the step is sharded)
      const bool wall = !P.PERIODIC && q == npairs_i - 1 && i >= F.nb && i < F.n && is_present(i, W) && is_owned(i, W);   // has_wall_pair
      if (W.owned && !wall) { const unsigned long long rm = W.rel_mask[i]; k = nth_set_bit(rm, q); }
      int j = WALL; i64 nvo;
      const int ni = W.ext_nv[i];
      if (!wall) {
        j = W.pot_list[(size_t)i * W.pot_cap + k] - 1;
        nvo = base_nv + s_nvb[lo] + W.pot_nvpre[(size_t)i * W.pot_cap + k];
      } else nvo = base_nv + s_nvb[lo] + (W.pair_nv[i] - (ni + 4));      // the wall, after the floe's last candidate
      p = base_p + pl;
      if (p < W.cap_pairs) {
        // was the pair in contact in the step before? (kf_rows left the partners of i's own pairs in contact; a hint: a
        // wrong one costs time, never a result)
        int live = 0, rows_before = 0;      // (the hint's top bits: the pair's contact rows in that step, HINT_ROWS_SHIFT)
        if (i < prev_n && (j >= 0 || wall)) {
          const int4 h0 = *(const int4 *)(W.hint + 8 * (size_t)i), h1 = *(const int4 *)(W.hint + 8 * (size_t)i + 4);
          const int want = wall ? (int)HINT_PARTNER_MASK : j + 1;       // (the wall: a partner code of its own)
          const int hv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
          for (int q8 = 0; q8 < 8; ++q8) if ((hv[q8] & HINT_PARTNER_MASK) == want) { live = 1; rows_before = (hv[q8] >> HINT_ROWS_SHIFT) & 7; }
        }
        if (W.pair_task) *(int4 *)(W.pair_task + 4 * p) = make_int4(rows_before, 0, 0, 0);
        FCLK(1, 3);
        W.pair_i[p] = i; W.pair_j[p] = j; W.pair_nvoff[p] = nvo;
        cls = pair_classify_ij(p, i, j, nvo, prev_n, P, W, live);
        FCLK(1, 4);
        if (cls == PC_LIVE || cls == PC_COLD || cls == PC_WIDE) teamv += ni + W.ext_nv[j];
      }
    }
    // The lists are appended to a WORKGROUP at a time: the counts of the eight waves meet in LDS and one thread per list
    // issues one atomic (the L2 serves the atomics of a line one by one, ~12 ns each: a wave at a time, the ~700 waves of
    // this launch queued for up to 6 us behind the two counters nearly all of them bump).
    unsigned long long mine = 0, m_own = 0;
#pragma unroll
    for (int c = PC_LIVE; c <= PC_GENERAL; ++c) {
      const unsigned long long m = __ballot(cls == c);
      if (lane == c) mine = m;
      if (cls == c) m_own = m;
    }
    if (lane >= PC_LIVE && lane <= PC_GENERAL) wcnt[t >> 6][lane] = __popcll(mine);
    __syncthreads();
    if (t >= PC_LIVE && t <= PC_GENERAL) {
      int tot = 0;
      for (int w = 0; w < 8; ++w) tot += wcnt[w][t];
      unsigned long long base = 0;
      if (tot) {
        const int ctr = t == PC_LIVE ? C_LIVE : t == PC_WIDE ? C_WIDE : t == PC_COLD ? C_COLD : t == PC_GENERAL ? (walls ? C_WALLS : C_RETRY) : C_TEAM;
        base = atomicAdd(&W.counters[ctr], (unsigned long long)tot);
        if (t == PC_GENERAL && !walls) atomicAdd(&W.counters[C_WALLS], (unsigned long long)tot);   // (counted either way: the next step sizes its wall launch from it)
      }
      wbase[t] = base;
    }
    __syncthreads();
    FCLK(1, 5);
    if (cls >= PC_LIVE && cls <= PC_GENERAL) {
      unsigned long long at = wbase[cls] + (unsigned long long)__popcll(m_own & ((1ULL << lane) - 1ULL));
      for (int w = 0; w < (t >> 6); ++w) at += (unsigned long long)wcnt[w][cls];
      (cls == PC_LIVE ? W.live_list : cls == PC_WIDE ? W.wide_list : cls == PC_COLD ? W.cold_list : cls == PC_GENERAL ? (walls ? W.wall_list : W.retry_list) : W.team_list)[at] = (int)p;
    }
    __syncthreads();                                   // (wcnt / wbase are written again in the next round)
  }
  for (int d = 32; d > 0; d >>= 1) teamv += __shfl_xor(teamv, d);
  __syncthreads();
  if (lane == 0) red[t >> 6] = (unsigned long long)teamv;
  __syncthreads();
  if (t == 0) { unsigned long long v = 0; for (int q = 0; q < 8; ++q) v += red[q]; if (v) atomicAdd(&W.counters[C_TEAMV], v); }
  FCLK(1, 6);
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
  FCLK(0, 0);
  const int T = blockIdx.x, t = threadIdx.x;
  const int e = 64 * T + (t >> 4), l = t & 15;
  const bool valid = e < W.cap_ext;
  // First round of loads: everything whose address is known on entry -- the counters of the go rule, the tile sums before
  // this tile, the row counts of its entries (first wave), and what the 16 lanes need of their entry.  (The grid covers
  // the entries the step may have -- next_cap, GoRule; an entry beyond this step's list has no pairs and no rows.)
  const int e0 = 64 * T + t;
  const long long cnt = (t < 64 && e0 < W.cap_ext) ? W.row_cnt[e0] : 0;
  const int K = valid ? W.pair_cnt[e] : 0;
  const i64 pb = valid ? W.pair_base[e] : 0;
  int ninc = valid ? W.inc_cnt[e] : 0;
  const double x = valid ? W.ext_x[e] : 0, y = valid ? W.ext_y[e] : 0;
  unsigned long long before = 0;
  for (int u = t; u < T; u += 1024) before += W.row_agg[(size_t)AGG_STRIDE * u];
  // (the go rule's words with the rest of the first round: lane k counter k and part k of the pair rows, every wave for itself)
  const int ck = (int)(threadIdx.x & 63) * C_STRIDE;
  const unsigned long long cv = (publish && ck < (int)C_COUNT) ? W.counters[ck] : 0ULL;
  unsigned long long prows = publish ? W.prow_part[(size_t)AGG_STRIDE * (threadIdx.x & 63)] : 0ULL;
  const unsigned long long go_word = W.counters[C_GO];
  for (int d = 32; d > 0; d >>= 1) prows += __shfl_xor(prows, d);
  prows += __shfl(cv, (int)C_PAIRROWS / (int)C_STRIDE);
  const bool go = publish ? go_by_values(ck, cv, rule, prows) : (go_word != 0);
  if (publish && T == 0 && t < (int)C_SLOTS) {
    const int c = t * C_STRIDE;
    host_copy[c] = (c == C_GO) ? (go ? 1ULL : 0ULL) : (c == C_PAIRROWS) ? prows : cv;
    if (c == C_GO) W.counters[C_GO] = go ? 1ULL : 0ULL;
  }
  asm volatile("" :: "v"(K), "v"(ninc), "v"(cnt), "v"(pb), "v"(x), "v"(y));    // (the loads above stay above: one round trip, not two)
  FCLK(0, 1);
  if (!go) return;
  for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d);
  if ((t & 63) == 0) red[t >> 6] = before;
  __syncthreads();
  before = 0;
  for (int q = 0; q < 16; ++q) before += red[q];
  if (t < 64) {
    long long inc = cnt;
    for (int d = 1; d < 64; d <<= 1) { const long long a = __shfl_up(inc, d); if (t >= d) inc += a; }
    s_off[t] = (long long)before + inc - cnt;
    if (e0 <= W.cap_ext) W.row_off[e0] = (i64)before + inc - cnt;
    if (T == (int)gridDim.x - 1 && t == 63) {      // the last tile: the total (an entry beyond the list has no rows: row_off[N] is the total too)
      W.counters[C_NROWS] = before + (unsigned long long)inc;
      if (e0 + 1 <= W.cap_ext) W.row_off[e0 + 1] = (i64)before + inc;
    }
  }
  __syncthreads();
  FCLK(0, 3);
  if (!valid) return;
  const int N0 = F.n;
  if (ninc > W.pot_cap) ninc = W.pot_cap;
  const int items = K + ninc;
  if (items == 0) {                                            // (most entries beyond the list, and floes out of contact)
    if (l == 0) { W.ov_ext[e] = 0; W.cf_ext[2 * e] = 0; W.cf_ext[2 * e + 1] = 0; W.ct_ext[e] = 0; centre_check(e, x, y, F, P, W); }
    if (l < 8) W.hint[8 * (size_t)e + l] = 0;
    return;
  }
  const unsigned gsh = (threadIdx.x & 63) & ~15u;            // this group's lanes within the wave's ballots
  // lane l: item l as it lies in memory (own pairs in order, mirrored pairs unsorted)
  i64 p = -1; int j = 0, fl = 0, nr = 0, src = 0x7fffffff; i64 ro = 0;
  const bool own = l < K, mir = l >= K && l < items;
  if (items <= 16) {
    if (own) { p = pb + l; j = W.pair_j[p]; fl = W.pair_flag[p]; nr = (fl & 4) ? W.pair_nrows[p] : 0; ro = W.pair_rowoff[p]; }
    else if (mir) { p = W.inc_list[(size_t)e * W.pot_cap + (l - K)]; src = W.pair_i[p]; nr = W.pair_nrows[p]; ro = W.pair_rowoff[p]; fl = 4; }
  }
  const bool serial = items > 16 || ((__ballot(nr > 2) >> gsh) & 0xffffULL) != 0;
  FCLK(0, 4);
  if (serial) {
    if (l == 0) {
      int *inc = W.inc_list + (size_t)e * W.pot_cap;          // ascending source floe, in place (row_count's insertion sort)
      for (int a = 1; a < ninc; ++a) {
        const int v = inc[a], sv = W.pair_i[v]; int b = a;
        while (b > 0 && W.pair_i[inc[b - 1]] > sv) { inc[b] = inc[b - 1]; --b; }
        inc[b] = v;
      }
      row_fill(e, W.cap_ext, F, P, W);
      transfer_collect(e, F, W);
    }
    // (next step's hint all the same when the items are in the lanes: a pair of three rows or more is where the general
    // pass shares the regions out over waves, and it asks the hint how many there were)
    if (items <= 16) {
      const unsigned mh = (unsigned)((__ballot(own && nr > 0) >> gsh) & 0xffffULL);
      if (own && nr > 0) { const int rk = __popc(mh & ((1u << l) - 1u)); if (rk < 8) W.hint[8 * (size_t)e + rk] = (j == WALL ? (int)HINT_PARTNER_MASK : j + 1) | ((nr > 7 ? 7 : nr) << HINT_ROWS_SHIFT); }
      if (l < 8 && l >= __popc(mh)) W.hint[8 * (size_t)e + l] = 0;
    } else if (l < 8) W.hint[8 * (size_t)e + l] = 0;
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
  FCLK(0, 5);
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
  // next step's hint: the partners of the own pairs in contact (kf_pairs sorts the team pass's pairs by it)
  {
    const unsigned mh = (unsigned)((__ballot(own && nr > 0) >> gsh) & 0xffffULL);
    if (own && nr > 0) { const int rk = __popc(mh & ((1u << l) - 1u)); if (rk < 8) W.hint[8 * (size_t)e + rk] = (j == WALL ? (int)HINT_PARTNER_MASK : j + 1) | ((nr > 7 ? 7 : nr) << HINT_ROWS_SHIFT); }
    if (l < 8 && l >= __popc(mh)) W.hint[8 * (size_t)e + l] = 0;
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
  FCLK(0, 6);
}

// ---- kf_integrate ----------------------------------------------------------------------------------------------------
// The transfer marks, integrate_pre and -- where no forcing is due -- integrate_post with 16 lanes per real floe: the
// one-thread forms walk the floe's contact rows and its ring vertex by vertex, a memory round trip per iteration (12 + 11
// us for 1e4 floes); here lane q loads row q and vertex q, the ring's extent is a min / max over the lanes, the ordered
// sums of calc_trajectory.m:9-13 are formed from registers in row order (the same additions in the same order), the four
// stress components are committed by four lanes.  Every lane carries the floe's scalars (same loads, same arithmetic).
// On a step that refreshes the forcing every live floe waits for k_forcing and k_integrate_post; on the others the few
// floes thinner than 0.1 m (calc_trajectory.m:94,121) go on a list that kf_forcing_post_list strides.
__global__ void __launch_bounds__(256) kf_integrate(Floes F, Params P, Work W) {   // grid: 16 lanes per entry of max(n - nb, cap_ext / 16 ...)
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (!W.counters[C_GO]) return;
  if (t < W.cap_ext) transfer_apply(t, F, W);
  if (t < F.nb) { F.collision_force[2 * t] = 0; F.collision_force[2 * t + 1] = 0; F.collision_torque[t] = 0; F.OverlapArea[t] = 0; }
  const int i = F.nb + (t >> 4), l = t & 15;
  if (i >= F.n) return;
  // (sharded step: another rank integrates this floe -- asked before the round of loads below, so that the waves of
  // floes that are not this rank's cost one byte each)
  if (W.owned && !W.owned[i]) { if (l == 0) W.st_flag[i] = 0; return; }
  // ---- first round of loads: the floe's scalars, its ghosts, its row range, its ring range
  const int gxe = P.PERIODIC ? W.xg_of[i] : -1, gye = P.PERIODIC ? W.yg_of[i] : -1;
  const i64 r0 = W.row_off[i], r1 = W.row_off[i + 1];
  const int v0 = F.vert_off[i], nvr = F.vert_off[i + 1] - v0;
  const double ov = W.ov_ext[i], x_before = W.ext_x[i];
  const double own_fx = W.cf_ext[2 * i], own_fy = W.cf_ext[2 * i + 1], own_tq = W.ct_ext[i];
  double Xi = F.Xi[i], Yi = F.Yi[i];
  const int alive0 = F.alive[i];
  const double area = F.area[i], h0 = F.h[i], mass0 = F.mass[i], inertia0 = F.inertia_moment[i];
  int scount = F.StressCount[i];
  const bool owned = is_owned(i, W);
  const double Ui = F.Ui[i], Vi = F.Vi[i], ksi0 = F.ksi_ice[i], alpha0 = F.alpha_i[i];
  const double dXi_p = F.dXi_p[i], dYi_p = F.dYi_p[i], dUi_p = F.dUi_p[i], dVi_p = F.dVi_p[i], dalpha_p = F.dalpha_i_p[i], dksi_p = F.dksi_ice_p[i];
  const double FxOA = F.FxOA[i], FyOA = F.FyOA[i], tqOA = F.torqueOA[i];
  asm volatile("" :: "v"(gxe), "v"(gye), "v"(r0), "v"(r1), "v"(v0), "v"(nvr), "v"(scount));
  if (l == 0) W.st_flag[i] = 0;
  if (!owned) return;   // (sharded step: another rank integrates this floe)
  // ---- second round: the ghosts' sums, my row, my vertex, the stress slot
  const int nrows = (int)(r1 - r0);
  const int Nz = F.n_stress;
  if (scount > Nz) scount = 1;                                   // :15-17
  double *SH = F.StressH + (size_t)i * Nz * 4;
  double gxf = 0, gxg = 0, gxt = 0, gyf = 0, gyg = 0, gyt = 0;
  if (gxe >= 0) { gxf = W.cf_ext[2 * gxe]; gxg = W.cf_ext[2 * gxe + 1]; gxt = W.ct_ext[gxe]; }
  if (gye >= 0) { gyf = W.cf_ext[2 * gye]; gyg = W.cf_ext[2 * gye + 1]; gyt = W.ct_ext[gye]; }
  double a1 = 0, a2 = 0, a3 = 0, a4 = 0;
  if (l < nrows) { const double *a = W.rows + 7 * (r0 + l); a1 = a[1]; a2 = a[2]; a3 = a[3]; a4 = a[4]; }
  double cx = 0, cy = 0, c0x = 0, c0y = 0;
  if (l < nvr) { cx = F.cax[v0 + l]; cy = F.cay[v0 + l]; c0x = F.c0x[v0 + l]; c0y = F.c0y[v0 + l]; }
  double sh_old = 0, ss_hi = 0, ss_lo = 0;
  if (l < 4) { sh_old = SH[(size_t)(scount - 1) * 4 + l]; ss_hi = F.StressSum[8 * i + l]; ss_lo = F.StressSum[8 * i + 4 + l]; }
  // ghost sums first (:242-245): the X ghost, then the Y ghost, of this floe; then the floe's own (:262-263)
  double cfx = 0, cfy = 0, ctq = 0;
  if (gxe >= 0) { cfx = cfx + gxf; cfy = cfy + gxg; ctq = ctq + gxt; }
  if (gye >= 0) { cfx = cfx + gyf; cfy = cfy + gyg; ctq = ctq + gyt; }
  if (nrows > 0) { cfx = own_fx + cfx; cfy = own_fy + cfy; ctq = own_tq + ctq; }
  if (l == 0) { F.collision_force[2 * i] = cfx; F.collision_force[2 * i + 1] = cfy; F.collision_torque[i] = ctq; F.OverlapArea[i] = ov; }
  if (P.PERIODIC) {                                              // :267-277
    bool moved = false;
    if (absd(Xi) > P.Lx) { Xi = Xi - 2 * P.Lx * sgn(Xi); moved = true; }
    if (absd(Yi) > P.Ly) { Yi = Yi - 2 * P.Ly * sgn(Yi); moved = true; }
    if (moved && l == 0) { F.Xi[i] = Xi; F.Yi[i] = Yi; }
  }
  if (!alive0) return;                                           // :280
  // ---- calc_trajectory.m ----
  const double dt = P.dt;
  double ext_fx = cfx, ext_fy = cfy, ext_torque = ctq;
  double S[4] = {0, 0, 0, 0};
  if (nrows > 0) {                                               // :9-13, the sums in row order
    double s11 = 0, s12 = 0, s21 = 0, s22 = 0, t11 = 0, t12 = 0, t21 = 0, t22 = 0;
    for (int base = 0; base < nrows; base += 16) {
      if (base > 0) { a1 = a2 = a3 = a4 = 0; if (base + l < nrows) { const double *a = W.rows + 7 * (r0 + base + l); a1 = a[1]; a2 = a[2]; a3 = a[3]; a4 = a[4]; } }
      const double dx = a3 - Xi, dy = a4 - Yi;
      const double p11 = dx * a1, p12 = dy * a1, p21 = dx * a2, p22 = dy * a2, q11 = a1 * dx, q12 = a2 * dx, q21 = a1 * dy, q22 = a2 * dy;
      const int m = nrows - base < 16 ? nrows - base : 16;
      for (int a = 0; a < m; ++a) {
        s11 += __shfl(p11, a, 16); s12 += __shfl(p12, a, 16); s21 += __shfl(p21, a, 16); s22 += __shfl(p22, a, 16);
        t11 += __shfl(q11, a, 16); t12 += __shfl(q12, a, 16); t21 += __shfl(q21, a, 16); t22 += __shfl(q22, a, 16);
      }
    }
    const double c = 1 / (2 * area * h0);
    S[0] = c * (s11 + t11); S[1] = c * (s21 + t21); S[2] = c * (s12 + t12); S[3] = c * (s22 + t22);
  }
  double hh = h0, mass = mass0, inertia = inertia0;
  int alive = alive0;
  if (hh > 10) hh = 10;                                          // :36-41
  else if (mass < 100) { mass = 1e3; alive = 0; }
  while (szc::maxd(absd(ext_fx), absd(ext_fy)) > mass / (5 * dt)) { // :42-46
    ext_fx = ext_fx / 10; ext_fy = ext_fy / 10; ext_torque = ext_torque / 10;
  }
  const double dh = P.HFo * dt / hh;                             // :76-80
  const double floe_mass = (hh - dh) / hh * mass;
  const double floe_inertia = (hh - dh) / hh * inertia;
  const double h_new = hh - dh;
  bool removed = false;
  if (Xi != Xi) removed = true;                                  // :89
  const bool doforce = P.doInt || h_new < 0.1;                   // :94,:121
  if (!removed) {                                                // :116, the extent of the ring over the lanes
    double mxx = -SZC_INF, mnx = SZC_INF, mxy = -SZC_INF, mny = SZC_INF;
    if (l < nvr) { mxx = mnx = cx; mxy = mny = cy; }
    for (int k = l + 16; k < nvr; k += 16) {
      const double X = F.cax[v0 + k], Y = F.cay[v0 + k];
      mxx = szc::maxd(mxx, X); mnx = szc::mind(mnx, X); mxy = szc::maxd(mxy, Y); mny = szc::mind(mny, Y);
    }
    for (int d = 1; d < 16; d <<= 1) {
      mxx = szc::maxd(mxx, __shfl_xor(mxx, d, 16)); mnx = szc::mind(mnx, __shfl_xor(mnx, d, 16));
      mxy = szc::maxd(mxy, __shfl_xor(mxy, d, 16)); mny = szc::mind(mny, __shfl_xor(mny, d, 16));
    }
    if (mxx + Xi > P.Xo_max || mnx + Xi < P.Xo_min || mxy + Yi > P.Yo_max || mny + Yi < P.Yo_min) removed = true;
  }
  if ((removed || x_before != x_before) && l == 0) F.kill[i] = i + 1;   // :282
  if (removed) return;
  // commit the staged fields: lane q the stress component q
  if (l < 4) {
    SH[(size_t)(scount - 1) * 4 + l] = S[l];                     // :18 / :26
    double hi = ss_hi, lo = ss_lo;
    stress_sum_replace(hi, lo, sh_old, S[l]);
    F.StressSum[8 * i + l] = hi; F.StressSum[8 * i + 4 + l] = lo;
    F.Stress[4 * i + l] = hi / Nz;                               // :20 mean(StressH,3)
  }
  if (l == 0) { F.StressCount[i] = scount + 1; F.h[i] = h_new; F.mass[i] = floe_mass; F.inertia_moment[i] = floe_inertia; F.alive[i] = alive; }
  if (alive != 1) return;                                        // :118
  if (l == 0) { W.st_ext[3 * i] = ext_fx; W.st_ext[3 * i + 1] = ext_fy; W.st_ext[3 * i + 2] = ext_torque; W.st_flag[i] = ST_GO | (doforce ? ST_FORCE : 0); }
  if (doforce) {
    if (!P.doInt && l == 0) W.force_list[atomicAdd(&W.counters[C_FORCE], 1ULL)] = i;
    return;
  }
  // ---- integrate_post (no forcing due: FxOA, FyOA, torqueOA are those of the last refresh); P.doInt is 0 here
  const double floe_area = area;
  // :174-177
  const double ddx = 1.5 * dt * Ui - 0.5 * dt * dXi_p, ddy = 1.5 * dt * Vi - 0.5 * dt * dYi_p;
  const double Xn = Xi + ddx, Yn = Yi + ddy;
  const double alpha = alpha0 + 1.5 * dt * ksi0 - 0.5 * dt * dalpha_p;
  // :181-208
  const double numU = FxOA * floe_area + ext_fx, numV = FyOA * floe_area + ext_fy;
  double dUi_dt = numU / floe_mass, dVi_dt = numV / floe_mass;
  bool have_frac = false;
  double frac = 0;
  const bool bigU = absd(dt * dUi_dt) > 0.5 * h_new, bigV = absd(dt * dVi_dt) > 0.5 * h_new;
  const bool smallU = absd(dt * dUi_dt) < 0.5 * h_new, smallV = absd(dt * dVi_dt) < 0.5 * h_new;
  if (bigU && bigV) {
    const double a = sgn(dUi_dt) * 0.5 * h_new / dt, b = sgn(dVi_dt) * 0.5 * h_new / dt;
    const double frac1 = a / numU * floe_mass, frac2 = b / numV * floe_mass;
    frac = szc::mind(frac1, frac2); have_frac = true;
  } else if (bigU && smallV) {
    const double a = sgn(dUi_dt) * 0.5 * h_new / dt;
    frac = a / numU * floe_mass; have_frac = true;
  } else if (smallU && bigV) {
    const double b = sgn(dVi_dt) * 0.5 * h_new / dt;
    frac = b / numV * floe_mass; have_frac = true;
  }
  if (have_frac) { dUi_dt = frac * (numU / floe_mass); dVi_dt = frac * (numV / floe_mass); }
  const double Un = Ui + 1.5 * dt * dUi_dt - 0.5 * dt * dUi_p, Vn = Vi + 1.5 * dt * dVi_dt - 0.5 * dt * dVi_p;
  // :210-219
  double dksi = (tqOA * floe_area + ext_torque) / floe_inertia;
  if (have_frac) dksi = frac * dksi;
  double ksi = ksi0 + 1.5 * dt * dksi - 0.5 * dt * dksi_p;
  if (absd(ksi) > 1e-5) ksi = sgn(ksi) * 1e-5;
  if (l == 0) {
    F.Xi[i] = Xn; F.dXi_p[i] = Ui; F.Yi[i] = Yn; F.dYi_p[i] = Vi; F.alpha_i[i] = alpha; F.dalpha_i_p[i] = ksi0;
    F.Ui[i] = Un; F.Vi[i] = Vn; F.dUi_p[i] = dUi_dt; F.dVi_p[i] = dVi_dt; F.ksi_ice[i] = ksi; F.dksi_ice_p[i] = dksi;
  }
  // :221-222, a vertex per lane
  double sa, ca;
  szm::sincos(alpha, sa, ca);
  if (l < nvr) { F.cax[v0 + l] = ca * c0x + (-sa) * c0y; F.cay[v0 + l] = sa * c0x + ca * c0y; }
  for (int k = l + 16; k < nvr; k += 16) {
    const double ux = F.c0x[v0 + k], uy = F.c0y[v0 + k];
    F.cax[v0 + k] = ca * ux + (-sa) * uy; F.cay[v0 + k] = sa * ux + ca * uy;
  }
}
