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
#define SZ_TP_SLOTS 48
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
#include "step_state.hpp"

namespace {
#ifndef SZ_TEAM_WAVES
#define SZ_TEAM_WAVES 3   // waves per SIMD the kernel is built for: 168 registers (at 128, for 4 waves, it spills 76 of them and is 8 % slower)
#endif
// One launch for both team widths: the first WIDE_BLOCKS workgroups take the pairs of W.wide_list with 32 lanes each
// (striding over the list: few pairs, and they start first), the others the pairs of W.live_list with 16 lanes each,
// four per wave (lists by pair_classify; counts on the device).  `count` bounds the live pairs the grid was sized for.
enum { WIDE_BLOCKS = 128 };
__global__ void __launch_bounds__(64, SZ_TEAM_WAVES) k_narrow_team(i64 count, Floes F, Params P, Work W) {
  __shared__ sztm::TeamBlock blocks[4];
#if defined(SZ_TEAM_PHASES)
  if (threadIdx.x == 0) { for (int k = 0; k < SZ_TP_SLOTS; ++k) sz_tp_acc[k] = 0; sz_tp_cur = 0; sz_tp_base = 0; sz_tp_last = wall_clock64(); }
#endif
  if (blockIdx.x < WIDE_BLOCKS) {
    const sztm::Team<32> T = sztm::Team<32>::here();
    const int tw = sztm::Team<32>::index_in_wave();
    const i64 n = (i64)W.counters[C_WIDE];
    for (i64 t = (i64)blockIdx.x * 2 + tw; t < n; t += (i64)WIDE_BLOCKS * 2) narrow_team((i64)W.wide_list[t], T, &blocks[tw], F, P, W);
  } else {
    const sztm::Team<16> T = sztm::Team<16>::here();
    const int tw = sztm::Team<16>::index_in_wave();
    const i64 n = (i64)W.counters[C_LIVE];
    const i64 t = (i64)(blockIdx.x - WIDE_BLOCKS) * 4 + tw;
    if (t < n && t < count) narrow_team((i64)W.live_list[t], T, &blocks[tw], F, P, W);
  }
#if defined(SZ_TEAM_PHASES)
  SZTM_PHASE(47);
  if (threadIdx.x == 0 && blockIdx.x >= WIDE_BLOCKS && blockIdx.x - WIDE_BLOCKS < SZ_TP_WAVES) for (int k = 0; k < SZ_TP_SLOTS; ++k) sz_team_phase[(size_t)(blockIdx.x - WIDE_BLOCKS) * SZ_TP_SLOTS + k] = sz_tp_acc[k];
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
    for (i64 t = (i64)blockIdx.x * (64 / G) + tw; t < n; t += (i64)gridDim.x * (64 / G)) narrow_team((i64)list[t], T, &blocks[tw], F, P, W);
  } else {
    const i64 t = (i64)blockIdx.x * (64 / G) + tw;
    if (t < count) narrow_team(first + t, T, &blocks[tw], F, P, W);
  }
}
// pair_classify for the pairs [0, count) (*dev_count of them when the host does not know the count yet): the lists are
// appended to a wave at a time (one atomic per wave and list)
__global__ void __launch_bounds__(256) k_pair_classify(i64 count, const unsigned long long *dev_count, Params P, Work W) {
  const i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const i64 n = dev_count ? (i64)*dev_count : count;
  const int cls = (p < n && p < count) ? pair_classify(p, P, W) : (int)PC_DONE;
  const int lane = threadIdx.x & 63;
  for (int c = PC_LIVE; c <= PC_OTHER; ++c) {
    const unsigned long long m = __ballot(cls == c);
    if (!m) continue;
    unsigned long long base = 0;
    if (lane == __builtin_ctzll(m)) base = atomicAdd(&W.counters[c == PC_LIVE ? C_LIVE : c == PC_WIDE ? C_WIDE : C_TEAM], (unsigned long long)__popcll(m));
    base = __shfl(base, __builtin_ctzll(m));
    if (cls == c) (c == PC_LIVE ? W.live_list : c == PC_WIDE ? W.wide_list : W.team_list)[base + __popcll(m & ((1ULL << lane) - 1ULL))] = (int)p;
  }
}
}

void sz_launch_pair_classify(hipStream_t st, i64 count, const unsigned long long *dev_count, const Params &P, const Work &W) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_pair_classify, grid_for(count, 256), dim3(256), 0, st, count, dev_count, P, W);
}
// the team pass over the classified pairs (count: the live pairs the launch is sized for, at most all candidates)
void sz_launch_narrow_team(hipStream_t st, i64 count, const Floes &F, const Params &P, const Work &W) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_narrow_team, dim3((unsigned)(WIDE_BLOCKS + (count + 3) / 4)), dim3(64), 0, st, count, F, P, W);
}
// ... over a range of the pair list, unclassified, 32 lanes per pair (the sharded step: a segment of the pair list)
void sz_launch_narrow_team_range(hipStream_t st, i64 first, i64 count, const Floes &F, const Params &P, const Work &W) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_narrow_team_range<32>, grid_for(count, 2), dim3(64), 0, st, first, count, (const int *)nullptr, (const unsigned long long *)nullptr, F, P, W);
}
