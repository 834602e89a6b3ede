// narrow_general.hip -- the narrow-phase kernels with one lane (or one wave) per candidate pair: the single-minimum fast
// path of convex.hpp over what the team pass hands on, the general Vatti sweep of vatti.hpp for what that defers (ties,
// concave and long rings, the capacity tiers), the floes at the walls.  Bodies: step_core.hpp narrow_t.
#include <stdlib.h>
#include <algorithm>
#include "context.hpp"
#if defined(SZ_PHASE_PROFILE)
// Diagnostic build only (make EXTRA=-DSZ_PHASE_PROFILE): wall-clock time of the parts of the contact model per
// pair of the general kernels, summed and for the slowest pair; read with szDebugPhases.  Not in the product build.
#include <hip/hip_runtime.h>
struct SzPhaseState { unsigned long long last, t[24]; int cur, trim[4], ntrim; };
__device__ unsigned long long sz_phase_acc[24];
__device__ unsigned long long sz_phase_worst[32];
__device__ SzPhaseState sz_phase_state[1 << 17];
static __host__ __device__ inline void sz_phase_mark(int k) {
#if defined(__HIP_DEVICE_COMPILE__)
  SzPhaseState &s = sz_phase_state[(blockIdx.x * blockDim.x + threadIdx.x) & 0x1ffff];
  const unsigned long long now = wall_clock64();
  if (k == 8) { for (int i = 0; i < 24; ++i) s.t[i] = 0; s.ntrim = 0; }
  else s.t[s.cur < 24 ? s.cur : 9] += now - s.last;
  s.last = now; s.cur = k;
  if (k == 0) {
    unsigned long long tot = 0;
    for (int i = 1; i < 24; ++i) { tot += s.t[i]; atomicAdd(&sz_phase_acc[i], s.t[i]); }
    if (atomicMax(&sz_phase_worst[0], tot) < tot) {
      for (int i = 1; i < 24; ++i) sz_phase_worst[i] = s.t[i];
      for (int i = 0; i < 4; ++i) sz_phase_worst[24 + i] = i < s.ntrim ? (unsigned long long)s.trim[i] : 0ull;
      sz_phase_worst[28] = (unsigned long long)s.ntrim;
    }
  }
#endif
}
static __host__ __device__ inline void sz_trim_mark(int before, int after) {
#if defined(__HIP_DEVICE_COMPILE__)
  SzPhaseState &s = sz_phase_state[(blockIdx.x * blockDim.x + threadIdx.x) & 0x1ffff];
  if (s.ntrim < 4) { s.trim[s.ntrim] = before; s.trim[s.ntrim + 1] = after; } s.ntrim += 2;
#endif
}
#define SZC_PHASE(k) sz_phase_mark(k)
#define SZV_PHASE(k) sz_phase_mark(k)
#define SZC_TRIM_PROBE(before, after) sz_trim_mark(before, after)
#endif
#include "step_core.hpp"
#if defined(SZ_PHASE_PROFILE)
extern "C" int szDebugPhases(unsigned long long *out56, int reset) {
  unsigned long long z[32] = {0};
  if (hipMemcpyFromSymbol(out56, HIP_SYMBOL(sz_phase_acc), sizeof(unsigned long long) * 24) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out56 + 24, HIP_SYMBOL(sz_phase_worst), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
  if (reset) { (void)hipMemcpyToSymbol(HIP_SYMBOL(sz_phase_acc), z, 192); (void)hipMemcpyToSymbol(HIP_SYMBOL(sz_phase_worst), z, 256); }
  return 0;
}
#endif

#include "narrow_general.hpp"

#if defined(SZ_WAVE_CYCLES)
// Diagnostic build only (make EXTRA=-DSZ_WAVE_CYCLES): wall-clock ticks (100 MHz) of every wave of the last fast-pass
// launch, read with szDebugWaveCycles.  Not in the product build.
__device__ unsigned long long sz_wave_cycles[1 << 17];
extern "C" int szDebugWaveCycles(unsigned long long *out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(sz_wave_cycles), sizeof(unsigned long long) * (size_t)(n < (1 << 17) ? n : (1 << 17))) == hipSuccess ? 0 : -1;
}
#endif

namespace {
__device__ inline int dev_next(const Work &W) { return (int)W.counters[C_NEXT]; }
// `lanes` of the 64 lanes of each wave carry a pair (the rest stay idle): with only ~4e4 pairs per step the
// chip is short of waves, not of lanes; fewer pairs per wave means more waves in flight to overlap the
// latency of the workspace accesses, and fewer divergent paths per wave.
__global__ void __launch_bounds__(64) k_narrow(i64 first, i64 count, const int *list, int mult, int lanes, int lds_per_lane, Floes F, Params P, Work W) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)threadIdx.x >= lanes) return;
  i64 t = (i64)blockIdx.x * lanes + threadIdx.x;
  void *fast = lds_per_lane ? (void *)(smem + (size_t)threadIdx.x * lds_per_lane) : nullptr;
  if (t < count) narrow(list ? t : first + t, list, mult, F, P, W, fast, lds_per_lane);
}
// k_narrow with the sweep workspace of every lane in LDS (ds_ addressing): for deferred pairs of short rings (ties,
// floes at the walls), whose general sweeps are chains of dependent accesses to a few KB of records
__global__ void __launch_bounds__(64) k_narrow_lds(i64 count, const int *list, int mult, int lanes, int lds_per_lane, Floes F, Params P, Work W) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)threadIdx.x >= lanes) return;
  i64 t = (i64)blockIdx.x * lanes + threadIdx.x;
  if (t < count) narrow_lds(t, list, mult, F, P, W, (void *)(smem + (size_t)threadIdx.x * lds_per_lane), lds_per_lane);
}
// k_narrow_lds with one pair per wave and ALL 64 lanes on it (step_core.hpp narrow_t, WAVE): the lanes run the pair's
// serial parts alike and share the loops over vertices in front of every clip -- the conversion to fixed point and the
// trimming, a quarter of a long pair's time when one lane does them.  Deferred lists at the 1x tier.
// (`tasks` waves per pair, the first of each pair in front: workgroup b is task b / count of pair b % count -- a pair whose
// overlap regions are shared out uses up to `tasks` of them, the others leave at once: step_core.hpp narrow_t)
__global__ void __launch_bounds__(64) k_narrow_wave(i64 count, const int *list, int lds_per_wave, int tasks, Floes F, Params P, Work W) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const i64 b = blockIdx.x;
  const i64 t = b % count;
  const int slot = (int)(b / count);
  if (slot < tasks) narrow_wave(t, list, F, P, W, (void *)smem, lds_per_wave, slot, tasks);
}
// The floes at the walls (k_pair_classify's list of their own): the general sweep of each against the domain box, one
// pair per wave with the sweep workspace in LDS, launched beside the team pass on a stream of its own -- the box has
// horizontal edges, so no fast path takes these pairs, and queued behind the team pass (as deferred pairs are) their
// quarter of a millisecond came on top of the step.  The waves stride over the list: any grid size is correct.
__global__ void __launch_bounds__(64) k_narrow_walls(int lds_per_wave, int tasks, Floes F, Params P, Work W) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // (all 64 lanes on the pair, as k_narrow_wave: the conversion to fixed point, InterX and the ring set-up of the sweep
  // spread over them -- SUBZERO_WALLS_LANE0: lane 0 alone, as until round 4)
  const i64 n = (i64)W.counters[C_WALLS];
  if (lds_per_wave < 0) {
    if (threadIdx.x != 0) return;
    for (i64 t = blockIdx.x; t < n; t += gridDim.x) narrow_lds(t, W.wall_list, 1, F, P, W, (void *)smem, -lds_per_wave);
    return;
  }
  // (`tasks` waves per pair as in k_narrow_wave: a floe in a corner of the box, or lying along a wall with several pieces
  // outside, has as many overlap regions)
  for (i64 b = blockIdx.x; b < n * tasks; b += gridDim.x) narrow_wave(b % n, W.wall_list, F, P, W, (void *)smem, lds_per_wave, (int)(b / n), tasks);
}
// First pass of the default path: every pair through the single-minimum fast path of convex.hpp, whose whole
// state is one FastBlock of LDS per lane (stride 8 x odd bytes: neighbouring lanes hit different banks).
// Pairs it does not cover are appended to the retry list and redone by k_narrow.
static_assert(CONVEX_STRIDE % 16 == 8 && CONVEX_STRIDE == sizeof(szx::FastBlock), "lane stride must be 8 x odd and hold a FastBlock");
// (dev_count: the pair count where the host does not know it yet -- the speculative step launches for a capacity)
#ifndef SZ_CONVEX_WAVES_ATTR
#define SZ_CONVEX_WAVES_ATTR
#endif
__global__ void __launch_bounds__(64) SZ_CONVEX_WAVES_ATTR k_narrow_convex(i64 first, i64 count, const unsigned long long *dev_count, int lanes, Floes F, Params P, Work W, const int *order) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)threadIdx.x >= lanes) return;
  i64 t = (i64)blockIdx.x * lanes + threadIdx.x;
  if (dev_count && t >= (i64)*dev_count) return;
#if defined(SZ_WAVE_CYCLES)
  const unsigned long long t0 = wall_clock64();
#endif
  if (t < count) narrow_convex(order ? t : first + t, order, F, P, W, (szx::FastBlock *)(smem + (size_t)threadIdx.x * CONVEX_STRIDE));
#if defined(SZ_WAVE_CYCLES)
  if (threadIdx.x == 0 && blockIdx.x < (1u << 17)) sz_wave_cycles[blockIdx.x] = wall_clock64() - t0;
#endif
}

// Second fast pass: the pairs the first one found too long for its 16-vertex blocks but not for 32-vertex ones (many-
// sided and mildly evolved floes), a few lanes per wave (1128 B of LDS each).  What it does not cover joins the list of
// the general kernel.
static_assert(CONVEX_BIG_STRIDE % 16 == 8 && CONVEX_BIG_STRIDE == sizeof(szx::FastBlockBig), "lane stride must be 8 x odd and hold a FastBlockBig");
__global__ void __launch_bounds__(64) k_narrow_convex_big(i64 count, const int *list, int lanes, Floes F, Params P, Work W) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)threadIdx.x >= lanes) return;
  i64 t = (i64)blockIdx.x * lanes + threadIdx.x;
  if (t < count) narrow_convex(t, list, F, P, W, (szx::FastBlockBig *)(smem + (size_t)threadIdx.x * CONVEX_BIG_STRIDE));
}

// Two-stage fast path: the sweep workspace of every lane (compact records, 2.3 KB) lives in LDS -- one
// 64-lane workgroup per CU (64 x 2344 B = 146.5 KB of the 160 KB); the lane stride is 8 x odd bytes so that
// the same field of neighbouring lanes falls into different banks.  k_narrow_a: first clip of every pair;
// k_narrow_b: contact model of the pairs that overlap (compacted by one atomic append per pair).
static_assert(FAST_WS_STRIDE % 16 == 8, "lane stride must be 8 x odd");
__global__ void __launch_bounds__(64) k_narrow_a(i64 first, i64 count, int lanes, Floes F, Params P, Work W) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)threadIdx.x >= lanes) return;
  i64 t = (i64)blockIdx.x * lanes + threadIdx.x;
  if (t < count) narrow_a(first + t, F, P, W, smem + (size_t)threadIdx.x * FAST_WS_STRIDE);
}
__global__ void __launch_bounds__(64) k_narrow_b(int lanes, Floes F, Params P, Work W) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)threadIdx.x >= lanes) return;
  const i64 count = (i64)W.counters[C_TOUCH];
  i64 t = (i64)blockIdx.x * lanes + threadIdx.x;
  if (t < count) narrow_b(t, F, P, W, smem + (size_t)threadIdx.x * FAST_WS_STRIDE);
}
} // namespace

int sz_narrow_general_configure(szContext *ctx, bool narrow_in_lds) {
  SZ_HIP(ctx, hipFuncSetAttribute((const void *)k_narrow_a, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * FAST_WS_STRIDE));
  SZ_HIP(ctx, hipFuncSetAttribute((const void *)k_narrow_b, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * FAST_WS_STRIDE));
  (void)hipFuncSetAttribute((const void *)k_narrow_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void *)k_narrow_walls, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void *)k_narrow_wave, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void *)k_narrow_convex_big, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * CONVEX_BIG_STRIDE);
  if (narrow_in_lds) (void)hipFuncSetAttribute((const void *)k_narrow, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return SZ_OK;
}
void szl_k_narrow(dim3 grid, size_t lds, hipStream_t st, i64 first, i64 count, const int *list, int mult, int lanes, int lds_per_lane, const Floes &F, const Params &P, const Work &W) {
  hipLaunchKernelGGL(k_narrow, grid, dim3(64), lds, st, first, count, list, mult, lanes, lds_per_lane, F, P, W);
}
void szl_k_narrow_lds(dim3 grid, size_t lds, hipStream_t st, i64 count, const int *list, int mult, int lanes, int lds_per_lane, const Floes &F, const Params &P, const Work &W) {
  hipLaunchKernelGGL(k_narrow_lds, grid, dim3(64), lds, st, count, list, mult, lanes, lds_per_lane, F, P, W);
}
void szl_k_narrow_wave(dim3 grid, size_t lds, hipStream_t st, i64 count, const int *list, int lds_per_wave, int tasks, const Floes &F, const Params &P, const Work &W) {
  hipLaunchKernelGGL(k_narrow_wave, grid, dim3(64), lds, st, count, list, lds_per_wave, tasks, F, P, W);
}
void szl_k_narrow_walls(dim3 grid, size_t lds, hipStream_t st, int lds_per_wave, int tasks, const Floes &F, const Params &P, const Work &W) {
  hipLaunchKernelGGL(k_narrow_walls, grid, dim3(64), lds, st, lds_per_wave, tasks, F, P, W);
}
void szl_k_narrow_convex(dim3 grid, size_t lds, hipStream_t st, i64 first, i64 count, const unsigned long long *dev_count, int lanes, const Floes &F, const Params &P, const Work &W, const int *order) {
  hipLaunchKernelGGL(k_narrow_convex, grid, dim3(64), lds, st, first, count, dev_count, lanes, F, P, W, order);
}
void szl_k_narrow_convex_big(dim3 grid, size_t lds, hipStream_t st, i64 count, const int *list, int lanes, const Floes &F, const Params &P, const Work &W) {
  hipLaunchKernelGGL(k_narrow_convex_big, grid, dim3(64), lds, st, count, list, lanes, F, P, W);
}
void szl_k_narrow_a(dim3 grid, size_t lds, hipStream_t st, i64 first, i64 count, int lanes, const Floes &F, const Params &P, const Work &W) {
  hipLaunchKernelGGL(k_narrow_a, grid, dim3(64), lds, st, first, count, lanes, F, P, W);
}
void szl_k_narrow_b(dim3 grid, size_t lds, hipStream_t st, int lanes, const Floes &F, const Params &P, const Work &W) {
  hipLaunchKernelGGL(k_narrow_b, grid, dim3(64), lds, st, lanes, F, P, W);
}
