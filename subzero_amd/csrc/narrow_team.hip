// narrow_team.hip -- the first pass of the narrow phase: a TEAM of lanes per candidate pair (team.hpp, teamclip.hpp,
// teamcontact.hpp; body: step_core.hpp narrow_team).  A translation unit of its own: it is the kernel the step spends
// most of its time in, and it is tuned on its own.
//
// Launch shape: one wave per workgroup, 64 / G pairs per wave, each team's whole working set in its TeamBlock of LDS
// (2.5 KB); only the contact rows and the per-pair result words go to HBM.  No MFMA: irregular int64 / FP64 geometry.
#if defined(SZ_TEAM_PHASES)
// Diagnostic build only (make -C subzero_amd/csrc diag_team): the time lane 0 of every wave spends in each part of
// teamcontact.hpp / teamclip.hpp (SZTM_PHASE(k) = "entering part k"; 100 MHz wall clock), kept per wave in LDS and
// written out when the wave's first team is done; the last launch's records are read with szDebugTeamPhases.
#include <hip/hip_runtime.h>
#define SZ_TP_SLOTS 50    // 0-47: time per part, 48: start of the wave, 49: its end (100 MHz wall clock)
#define SZ_TP_WAVES (1 << 15)
__device__ unsigned long long sz_team_phase[(size_t)SZ_TP_WAVES * SZ_TP_SLOTS];
__shared__ unsigned long long sz_tp_acc[SZ_TP_SLOTS];
__shared__ unsigned long long sz_tp_last;
__shared__ int sz_tp_cur, sz_tp_base;
#define SZTM_PHASE(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); sz_tp_acc[sz_tp_cur] += now_ - sz_tp_last; sz_tp_last = now_; const int k_ = (k); sz_tp_cur = (k_ >= 10 && k_ < 20) ? k_ + sz_tp_base : k_; if (k_ == 6) sz_tp_base = 20; } } while (0)
extern "C" int szDebugTeamPhases(unsigned long long *out, int nwaves) {
  if (nwaves > SZ_TP_WAVES) nwaves = SZ_TP_WAVES;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(sz_team_phase), sizeof(unsigned long long) * (size_t)nwaves * SZ_TP_SLOTS) == hipSuccess ? 0 : -1;
}
#endif
#include <stdlib.h>
#include "step_state.hpp"

namespace {
#ifndef SZ_TEAM_WAVES
#define SZ_TEAM_WAVES 4   // waves per SIMD the kernel is built for: 128 registers (round 4: team-uniform values kept in the block across the clips)
#endif
// One launch for the three lists of pair_classify (counts on the device); a wave takes four pairs of 16 lanes (or two
// of 32) at a time.  The grid is laid out from what the lists held in the step before -- any size is correct, the
// waves stride over their list:
//   blocks [0, wide_blocks)            wide_list, 32 lanes per pair         (few, and the longest: first)
//   then, in the order sz_launch_narrow_team picks (cold first by default: measured there),
//   hot_blocks                         live_list (in contact last step: the whole model, ~45 us a wave) -- a wave each
//   cold_blocks                        cold_list (most stop at the no-row certificate, ~9 us a wave)
#ifndef SZ_TEAM_WPB
#define SZ_TEAM_WPB 1     // waves per workgroup (each wave on its own: no barrier between them)
#endif
__global__ void __launch_bounds__(64 * SZ_TEAM_WPB, SZ_TEAM_WAVES) k_narrow_team(int wide_blocks, int hot_blocks, int cold_blocks, int ratio, Floes F, Params P, Work W) {
#if SZ_TEAM_WPB == 1
  __shared__ sztm::TeamBlock blocks[4];
  const int wave = 0;
#else
  __shared__ sztm::TeamBlock blocks_all[4 * SZ_TEAM_WPB];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (a scalar: the same in every lane of the wave)
  sztm::TeamBlock *blocks = blocks_all + 4 * wave;
#endif
#if defined(SZ_TEAM_LDS_PAD)
  // tuning builds only (make team_variant): idle LDS that caps the waves per SIMD below what the registers allow
  __shared__ volatile int lds_pad[SZ_TEAM_LDS_PAD / 4];
  if (wide_blocks < 0) lds_pad[threadIdx.x] = hot_blocks;
#endif
#if defined(SZ_TEAM_PHASES)
  if (threadIdx.x == 0) { for (int k = 0; k < SZ_TP_SLOTS; ++k) sz_tp_acc[k] = 0; sz_tp_cur = 0; sz_tp_base = 0; sz_tp_last = wall_clock64(); sz_tp_acc[48] = sz_tp_last; }
#endif
  const int b = (int)blockIdx.x * SZ_TEAM_WPB + wave;   // the wave's place in the order of dispatch
  if (b < wide_blocks) {
    const sztm::Team<32> T = sztm::Team<32>::here();
    const int tw = sztm::Team<32>::index_in_wave();
    const i64 n = (i64)W.counters[C_WIDE];
    for (i64 t = (i64)b * 2 + tw; t < n; t += (i64)wide_blocks * 2) {
      sztm::Team<32> Tp = T; int twp = tw;
      sztm::opaque(Tp.lane); sztm::opaque(twp);               // (team.hpp opaque: nothing of the lane hoisted out of this loop)
      narrow_team((i64)W.wide_list[t], Tp, &blocks[twp], F, P, W, false, true);
    }
  } else {
    const sztm::Team<16> T = sztm::Team<16>::here();
    const int tw = sztm::Team<16>::index_in_wave();
    // hot and cold workgroups take turns in the order of dispatch (`ratio` hot ones, then a cold one): both kinds run
    // for tens of microseconds, and a kind left for the end would leave the chip part empty for its whole duration
    // (one instance of the model for both kinds: `cold` only says whether the no-row certificate is tried first)
    const int k = b - wide_blocks;
    bool hot; int blk;
    if (ratio > 0) {
      // `ratio` hot workgroups, then a cold one, until the hot list is covered; the rest of the cold ones behind
      const int n1 = (hot_blocks + ratio - 1) / ratio, p1 = n1 * (ratio + 1);
      if (k < p1) { const int g = k / (ratio + 1), r = k % (ratio + 1); hot = r < ratio; blk = hot ? g * ratio + r : g; }
      else { hot = false; blk = n1 + (k - p1); }
    }
    else if (ratio == 0) { hot = k < hot_blocks; blk = hot ? k : k - hot_blocks; }       // every hot workgroup first
    else {                                                                               // -ratio cold workgroups first, then every hot one, then the other cold ones
      const int c1 = -ratio < cold_blocks ? -ratio : cold_blocks;
      hot = k >= c1 && k < c1 + hot_blocks; blk = hot ? k - c1 : (k < c1 ? k : k - hot_blocks);
    }
    const int nblk = hot ? hot_blocks : cold_blocks;
    const int *list = hot ? W.live_list : W.cold_list;
    const i64 n = (i64)W.counters[hot ? C_LIVE : C_COLD];
    if (blk < nblk) for (i64 t = (i64)blk * 4 + tw; t < n; t += (i64)nblk * 4) {
      sztm::Team<16> Tp = T; int twp = tw;
      sztm::opaque(Tp.lane); sztm::opaque(twp);
      narrow_team((i64)list[t], Tp, &blocks[twp], F, P, W, !hot, true);
    }
  }
#if defined(SZ_TEAM_PHASES)
  SZTM_PHASE(47);
  if (threadIdx.x == 0) { sz_tp_acc[49] = wall_clock64(); }
  if (threadIdx.x == 0 && blockIdx.x < SZ_TP_WAVES) for (int k = 0; k < SZ_TP_SLOTS; ++k) sz_team_phase[(size_t)blockIdx.x * SZ_TP_SLOTS + k] = sz_tp_acc[k];
#endif
}
// the pairs [first, first + count) with G lanes each (the sharded step: a segment of the pair list; no classification)
template <int G>
__global__ void __launch_bounds__(64, SZ_TEAM_WAVES) k_narrow_team_range(i64 first, i64 count, const int *list, const unsigned long long *list_count, Floes F, Params P, Work W) {
  __shared__ sztm::TeamBlock blocks[64 / G];
  const sztm::Team<G> T = sztm::Team<G>::here();
  const int tw = sztm::Team<G>::index_in_wave();
  if (list) {
    const i64 n = (i64)*list_count;
    for (i64 t = (i64)blockIdx.x * (64 / G) + tw; t < n; t += (i64)gridDim.x * (64 / G)) {
      sztm::Team<G> Tp = T; int twp = tw;
      sztm::opaque(Tp.lane); sztm::opaque(twp);               // (team.hpp opaque: nothing of the lane hoisted out of this loop)
      narrow_team((i64)list[t], Tp, &blocks[twp], F, P, W);
    }
  } else {
    const i64 t = (i64)blockIdx.x * (64 / G) + tw;
    if (t < count) narrow_team(first + t, T, &blocks[tw], F, P, W);
  }
}
// pair_classify for the pairs [0, count) (*dev_count of them when the host does not know the count yet): the lists are
// appended to a wave at a time (one atomic per wave and list)
__global__ void __launch_bounds__(256) k_pair_classify(i64 count, const unsigned long long *dev_count, int prev_n, Params P, Work W) {
  const i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const i64 n = dev_count ? (i64)*dev_count : count;
  const int cls = (p < n && p < count) ? pair_classify(p, prev_n, P, W) : (int)PC_DONE;
  const int lane = threadIdx.x & 63;
  {   // ring vertices of the pairs the team pass will take (its algorithmic bytes: bench.py)
    long long tv = 0;
    if (cls == PC_LIVE || cls == PC_COLD || cls == PC_WIDE) tv = W.ext_nv[W.pair_i[p]] + W.ext_nv[W.pair_j[p]];
    for (int d = 32; d > 0; d >>= 1) tv += __shfl_xor(tv, d);
    if (lane == 0 && tv) atomicAdd(&W.counters[C_TEAMV], (unsigned long long)tv);
  }
  // the five lists are appended to a wave at a time: the ballots first, then ONE atomic per list, all of them issued by
  // one instruction (lane c takes list c), so that a wave waits for one round trip to the counters, not five
  const bool walls = W.wall_list != nullptr;      // (a list of their own when the step sweeps them beside the team pass)
  unsigned long long mine = 0;                    // lane c: the ballot of class c
  unsigned long long m_own = 0;                   // every lane: the ballot of its own class
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
    if (lane == PC_GENERAL && !walls) atomicAdd(&W.counters[C_WALLS], (unsigned long long)__popcll(mine));   // (counted either way: the next step sizes its wall launch from it)
  }
  const int src = (cls >= PC_LIVE && cls <= PC_GENERAL) ? cls : 0;
  base = __shfl(base, src);
  if (cls >= PC_LIVE && cls <= PC_GENERAL)
    (cls == PC_LIVE ? W.live_list : cls == PC_WIDE ? W.wide_list : cls == PC_COLD ? W.cold_list : cls == PC_GENERAL ? (walls ? W.wall_list : W.retry_list) : W.team_list)[base + __popcll(m_own & ((1ULL << lane) - 1ULL))] = (int)p;
}
}

// ---- szDebugTeamClipBatch: the clip of the team pass ALONE (tests) -------------------------------------------------------
// One team per pair, 64 / G pairs per wave, rings in 2^32 fixed point as the B1 surface takes them; the closing copy of a
// ring is dropped as the step's staging does.  status 100 (ST_BAIL): not this form's case (the step hands such a pair on).
namespace {
template <int G>
__global__ void __launch_bounds__(64) k_debug_team_clip(int npairs, const i64 *ax, const i64 *ay, const int *aoff, const i64 *bx, const i64 *by,
                                                        const int *boff, i64 *outx, i64 *outy, int *out_n, int *status) {
  __shared__ sztm::ClipScratch scr[64 / G];
  __shared__ i64 ox[64 / G][sztm::T_CAPV], oy[64 / G][sztm::T_CAPV];
  const sztm::Team<G> T = sztm::Team<G>::here();
  const int tw = sztm::Team<G>::index_in_wave();
  const int p = (int)blockIdx.x * (64 / G) + tw;
  if (p >= npairs) return;                                  // (team-uniform: the teams of a wave may diverge from each other)
  int na = aoff[p + 1] - aoff[p], nb = boff[p + 1] - boff[p];
  const i64 *pax = ax + aoff[p], *pay = ay + aoff[p], *pbx = bx + boff[p], *pby = by + boff[p];
  if (na > 1 && pax[na - 1] == pax[0] && pay[na - 1] == pay[0]) --na;
  if (nb > 1 && pbx[nb - 1] == pbx[0] && pby[nb - 1] == pby[0]) --nb;
  int st = szx::ST_BAIL, n = 0;
  if (na <= (int)sztm::T_NV && nb <= (int)sztm::T_NV && na >= 1 && nb >= 1) {
    sztm::ClipScratch *S = &scr[tw];
    for (int i = T.lane; i < na; i += G) { S->fx[0][i] = pax[i]; S->fy[0][i] = pay[i]; }
    for (int i = T.lane; i < nb; i += G) { S->fx[1][i] = pbx[i]; S->fy[1][i] = pby[i]; }
    T.sync(TEAM_SITE);
    st = sztm::team_clip(T, S, na, nb, ox[tw], oy[tw], &n);
    T.sync(TEAM_SITE);
    if (st == szv::ST_OK) for (int i = T.lane; i < n; i += G) { outx[(size_t)p * sztm::T_CAPV + i] = ox[tw][i]; outy[(size_t)p * sztm::T_CAPV + i] = oy[tw][i]; }
  }
  if (T.lane == 0) { status[p] = st; out_n[p] = st == szv::ST_OK ? n : 0; }
}
}
extern "C" int szDebugTeamClipBatch(szContext *ctx, int lanes, int npairs, const int64_t *ax, const int64_t *ay, const int32_t *aoff,
                                    const int64_t *bx, const int64_t *by, const int32_t *boff, int64_t *outx, int64_t *outy,
                                    int32_t *out_n, int32_t *status) {
  if (!ctx || npairs < 0 || (lanes != 16 && lanes != 32) || !ax || !ay || !aoff || !bx || !by || !boff || !outx || !outy || !out_n || !status) return SZ_ERR_BAD_ARG;
  if (npairs == 0) return SZ_OK;
  for (int p = 0; p < npairs; ++p) if (aoff[p + 1] < aoff[p] || boff[p + 1] < boff[p]) return sz_fail(ctx, SZ_ERR_BAD_ARG, "szDebugTeamClipBatch: offsets of pair %d descend", p);
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  const size_t na = (size_t)aoff[npairs], nb = (size_t)boff[npairs], no = (size_t)npairs * sztm::T_CAPV;
  DevBuf d[11];
  const size_t bytes[11] = {8 * na, 8 * na, 4 * ((size_t)npairs + 1), 8 * nb, 8 * nb, 4 * ((size_t)npairs + 1), 8 * no, 8 * no, 4 * (size_t)npairs, 4 * (size_t)npairs, 0};
  const void *src[6] = {ax, ay, aoff, bx, by, boff};
  int rc = SZ_OK;
  hipError_t e = hipSuccess;
  for (int k = 0; k < 10 && !rc; ++k) rc = sz_reserve(ctx, d[k], bytes[k] + 8);
  for (int k = 0; k < 6 && !rc && e == hipSuccess; ++k) e = hipMemcpyAsync(d[k].p, src[k], bytes[k], hipMemcpyHostToDevice, ctx->stream);
  if (!rc && e == hipSuccess) {
    const unsigned grid = (unsigned)((npairs + 64 / lanes - 1) / (64 / lanes));
    if (lanes == 16) hipLaunchKernelGGL(k_debug_team_clip<16>, dim3(grid), dim3(64), 0, ctx->stream, npairs, (const i64 *)d[0].p, (const i64 *)d[1].p, (const int *)d[2].p, (const i64 *)d[3].p, (const i64 *)d[4].p, (const int *)d[5].p, (i64 *)d[6].p, (i64 *)d[7].p, (int *)d[8].p, (int *)d[9].p);
    else hipLaunchKernelGGL(k_debug_team_clip<32>, dim3(grid), dim3(64), 0, ctx->stream, npairs, (const i64 *)d[0].p, (const i64 *)d[1].p, (const int *)d[2].p, (const i64 *)d[3].p, (const i64 *)d[4].p, (const int *)d[5].p, (i64 *)d[6].p, (i64 *)d[7].p, (int *)d[8].p, (int *)d[9].p);
    e = hipGetLastError();
  }
  void *dst[4] = {outx, outy, out_n, status};
  for (int k = 0; k < 4 && !rc && e == hipSuccess; ++k) e = hipMemcpyAsync(dst[k], d[6 + k].p, bytes[6 + k], hipMemcpyDeviceToHost, ctx->stream);
  const hipError_t e2 = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) e = e2;
  for (int k = 0; k < 10; ++k) sz_release(d[k]);
  if (rc) return rc;
  if (e != hipSuccess) return sz_fail(ctx, SZ_ERR_HIP, "szDebugTeamClipBatch: %s", hipGetErrorString(e));
  return SZ_OK;
}

void sz_launch_pair_classify(hipStream_t st, i64 count, const unsigned long long *dev_count, int prev_n, const Params &P, const Work &W) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_pair_classify, grid_for(count, 256), dim3(256), 0, st, count, dev_count, prev_n, P, W);
}
// the team pass over the classified pairs: `count` candidates; hot / cold / wide: what the three lists held in the step
// before (negative: unknown -- every pair may be on any list)
void sz_launch_narrow_team(hipStream_t st, i64 count, i64 hot, i64 cold, i64 wide, const Floes &F, const Params &P, const Work &W) {
  if (count <= 0) return;
  if (hot < 0) { hot = count; cold = count; wide = count / 16; }
  const i64 hb = (hot + hot / 8) / 4 + 8;              // a wave of four per workgroup, with room for the list to grow
  static int chunk = 0;                                // waves' worth of cold pairs per workgroup (SUBZERO_COLD_CHUNK)
  if (!chunk) { const char *v = getenv("SUBZERO_COLD_CHUNK"); chunk = v ? atoi(v) : 1; if (chunk < 1 || chunk > 16) chunk = 1; }   // (measured at 1e4 floes: 1 -> 0.133 ms, 2 -> 0.139, 4 -> 0.172)
  const i64 cb = (cold + cold / 8) / (4 * chunk) + 8;
  i64 wb = (wide + 1) / 2 + 4;
  if (wb > 512) wb = 512;
  i64 ratio = (hb + cb / 2) / cb;                       // hot workgroups per cold one in the order of dispatch
  if (ratio < 1) ratio = 1;
  // Order of dispatch (SUBZERO_TEAM_ORDER: 0 interleaved, 1 hot first, 2 cold first -- the default since round 4).  With
  // four waves per SIMD the chip holds 4096 waves.  Measured at 1e4 floes (profiles/team_order.sh; timelines by
  // profiles/team_phases.py): hot first 0.103 ms -- the 3.4 k hot waves fill the chip for 38 us during which nothing else
  // starts, then the cold ones, whose long items (a pair newly in contact runs the whole model) make a 40 us tail; `ratio`
  // hot per cold, any ratio, 0.102-0.106; cold first 0.095: the cold waves are through in ~10 us each, the hot ones start
  // as they leave and keep the chip full until 50 us; the share of the cold workgroups put first, 15 % ... 100 %: 0.103 ...
  // 0.095, monotone.
  static int order = -1;
  if (order < 0) { const char *v = getenv("SUBZERO_TEAM_ORDER"); order = v ? atoi(v) : 2; if (order < 0 || order > 2) order = 2; }
  static int fixed_ratio = -1;                          // SUBZERO_TEAM_RATIO: hot workgroups per cold one (0: chosen here)
  if (fixed_ratio < 0) { const char *v = getenv("SUBZERO_TEAM_RATIO"); fixed_ratio = v ? atoi(v) : 0; if (fixed_ratio < 0 || fixed_ratio > 64) fixed_ratio = 0; }
  if (fixed_ratio) ratio = fixed_ratio;
  const i64 n1 = (hb + ratio - 1) / ratio;
  i64 blocks = wb + n1 * (ratio + 1) + std::max<i64>(0, cb - n1);
  static double split = -1;                             // SUBZERO_TEAM_SPLIT: share of the cold workgroups dispatched before the hot ones (order 2)
  if (split < 0) { const char *v = getenv("SUBZERO_TEAM_SPLIT"); split = v ? atof(v) : 1.0; if (!(split >= 0 && split <= 1)) split = 1.0; }
  if (order) { ratio = order == 1 ? 0 : -std::max<i64>(1, (i64)(split * (double)cb)); blocks = wb + hb + cb; }
  hipLaunchKernelGGL(k_narrow_team, dim3((unsigned)((blocks + SZ_TEAM_WPB - 1) / SZ_TEAM_WPB)), dim3(64 * SZ_TEAM_WPB), 0, st, (int)wb, (int)hb, (int)cb, (int)ratio, F, P, W);
}
// ... over a range of the pair list, unclassified, 32 lanes per pair (the sharded step: a segment of the pair list)
void sz_launch_narrow_team_range(hipStream_t st, i64 first, i64 count, const Floes &F, const Params &P, const Work &W) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_narrow_team_range<32>, grid_for(count, 2), dim3(64), 0, st, first, count, (const int *)nullptr, (const unsigned long long *)nullptr, F, P, W);
}
