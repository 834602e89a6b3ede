// team.hpp -- a TEAM of G lanes of one wavefront (G = 16 or 32) working on one candidate pair.
//
// The narrow phase of a sea-ice step is tens of thousands of small, irregular problems (two rings of 4-16 vertices
// each).  One lane per pair leaves a wave as slow as its slowest pair's serial chain of ~3e4 dependent int64 / FP64
// instructions; here the vertex x edge work of one pair is spread over the G lanes of a team,
// 64 / G teams per wave, and the lanes talk through the team's block of LDS and a handful of cross-lane primitives:
//
//   shfl(v, src)   value of `v` in lane `src` of the team          (ds_bpermute within the team's lanes)
//   ballot(p)      bit l = p of lane l of the team                 (one v_cmp + s_and of the wave's exec mask)
//   sync()         LDS writes of the team's lanes before it are visible to its reads after it
//   scans / reductions built from those
//
// RULE: every primitive is called in TEAM-UNIFORM control flow only (all lanes of the team reach the same call; teams of
// one wave may diverge from each other).  The host build checks exactly that: tests/hostsim runs the G lanes of a team
// as G fibers that meet at every primitive, and aborts when they arrive from different call sites or when one lane has
// returned while the others still wait (TEAM_SITE).  The same source runs on the GPU and, lane by lane, on the CPU.
#pragma once
#include "vatti.hpp"
#if !defined(__HIPCC__)
#include <stdio.h>
#include <stdlib.h>
#endif

#if defined(__HIPCC__)
#define SZ_TEAM __device__ inline
#define SZ_TEAM_NOINLINE __device__ __attribute__((noinline))
#else
#define SZ_TEAM inline
#define SZ_TEAM_NOINLINE inline __attribute__((noinline))
#endif

namespace sztm {

using szv::i64;
using szv::u64;

#if !defined(__HIPCC__)
// ---- host runtime interface (implemented by tests/hostsim/team_host.hpp) -------------------------------------------
struct HostTeamRt {
  // all live lanes meet here; returns the array of the G payloads of this meeting (index = lane); dead[l] != 0 for
  // lanes that have returned
  virtual const u64 *meet(int lane, int site, u64 payload, const unsigned char **dead) = 0;
  virtual ~HostTeamRt() {}
};
#endif

template <int G_>
struct Team {
  enum { G = G_ };
  static_assert(G_ == 16 || G_ == 32, "a team is 16 or 32 lanes of one wavefront");
  int lane;   // 0 .. G-1
#if defined(__HIPCC__)
  // ---- device -------------------------------------------------------------------------------------------------
  __device__ static Team here() { Team t; t.lane = (int)(threadIdx.x & (G - 1)); return t; }
  __device__ static int index_in_wave() { return (int)((threadIdx.x & 63) / G); }
  __device__ int shfl(int v, int src, int) const { return __shfl(v, src, G); }
  __device__ unsigned ballot(bool p, int) const {
    const unsigned long long m = __ballot(p);
    const int sh = (int)((threadIdx.x & 63) & ~(G - 1));
    return (unsigned)(m >> sh) & (G == 32 ? 0xffffffffu : ((1u << (G & 31)) - 1u));
  }
  // the lanes of a team are lanes of one wave and the LDS serves a wave's accesses in order: all that is needed is
  // that the compiler keeps this lane's LDS writes before, and its reads after, this point (no s_barrier: teams of one
  // wave may stand at different syncs)
  __device__ void sync(int) const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
#else
  // ---- host: fibers --------------------------------------------------------------------------------------------
  HostTeamRt *rt;
  int shfl(int v, int src, int site) const {
    const unsigned char *dead;
    const u64 *s = rt->meet(lane, site, (u64)(unsigned)v, &dead);
    if (src < 0 || src >= G || dead[src]) { fprintf(stderr, "team: lane %d reads lane %d (dead or out of range) at line %d\n", lane, src, site / 256); abort(); }   // a value of a lane that has left: undefined on the GPU
    return (int)(unsigned)s[src];
  }
  unsigned ballot(bool p, int site) const {
    const unsigned char *dead;
    const u64 *s = rt->meet(lane, site, p ? 1 : 0, &dead);
    unsigned m = 0;
    for (int l = 0; l < G; ++l) if (!dead[l] && s[l]) m |= 1u << l;
    return m;
  }
  void sync(int site) const { const unsigned char *dead; (void)rt->meet(lane, site, 0, &dead); }
#endif
  // ---- built from the three ------------------------------------------------------------------------------------
  SZ_TEAM i64 shfl64(i64 v, int src, int site) const {
    const unsigned lo = (unsigned)shfl((int)(unsigned)((u64)v & 0xffffffffu), src, site);
    const unsigned hi = (unsigned)shfl((int)(unsigned)((u64)v >> 32), src, site + 100000);
    return (i64)(((u64)hi << 32) | lo);
  }
  SZ_TEAM double shfld(double v, int src, int site) const {
    union { double d; i64 i; } a, b;
    a.d = v; b.i = shfl64(a.i, src, site);
    return b.d;
  }
  SZ_TEAM bool any(bool p, int site) const { return ballot(p, site) != 0; }
  // exclusive prefix sum over the lanes; *total = sum over the team (packed small counters add without carries
  // between their fields as long as every field's total stays below its width)
  SZ_TEAM int scan_excl(int v, int *total, int site) const {
    int inc = v;
#pragma unroll
    for (int d = 1; d < G; d <<= 1) {
      const int u = shfl(inc, lane >= d ? lane - d : lane, site + d);
      if (lane >= d) inc += u;
    }
    *total = shfl(inc, G - 1, site + 64);
    return inc - v;
  }
  // minimum / maximum over the team, in every lane (exact and order-free)
  SZ_TEAM double min_all(double v, int site) const {
#pragma unroll
    for (int d = 1; d < G; d <<= 1) { const double u = shfld(v, lane ^ d, site + d); v = (u < v) ? u : v; }
    return v;
  }
  SZ_TEAM double max_all(double v, int site) const {
#pragma unroll
    for (int d = 1; d < G; d <<= 1) { const double u = shfld(v, lane ^ d, site + d); v = (u > v) ? u : v; }
    return v;
  }
  SZ_TEAM int sum_all(int v, int site) const {
#pragma unroll
    for (int d = 1; d < G; d <<= 1) v += shfl(v, lane ^ d, site + d);
    return v;
  }
};

// `v` as a value the compiler cannot see through.  Inside a loop over pairs (or over the two clips of a pair) everything
// that depends only on the lane number -- ring and vertex of the lane, its LDS addresses, masks -- is loop-invariant: the
// compiler hoists it out of the loop and holds it in registers across the whole body (40 of them, measured), which costs
// the kernel a wave per SIMD.  A lane number laundered at the top of the body keeps those values where they are used.
SZ_TEAM void opaque(int &v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#else
  (void)v;
#endif
}

// call sites: the line number tells the host runtime where a lane stands
#define TEAM_SITE (__LINE__ * 256)

} // namespace sztm
