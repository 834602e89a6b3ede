// TEST-ONLY: compiles the device sweep engine (subzero_amd/csrc/vatti.hpp) for the host with g++,
// so the CPU test-suite can check the algorithm the GPU kernels run against the reference Clipper
// (oracle/_ref) without a GPU.  Not part of the product; nothing in subzero_amd/ loads this.
#include "../../subzero_amd/csrc/vatti.hpp"
#include "../../subzero_amd/csrc/convex.hpp"
#include <stdint.h>
#include <stdlib.h>
#include <vector>

namespace {
struct RingView {
  const int64_t *x, *y;
  szv::Pt operator()(int i) const { szv::Pt p; p.x = x[i]; p.y = y[i]; return p; }
};
struct Sink {
  int64_t *ox, *oy; int64_t cap_v; int32_t *ring_off; int64_t cap_r; int64_t nv, nr;
  bool ring_begin() { return nr < cap_r; }
  bool vertex(const szv::Pt &p) { if (nv >= cap_v) return false; ox[nv] = p.x; oy[nv] = p.y; ++nv; return true; }
  bool ring_end() { ++nr; ring_off[nr] = (int32_t)nv; return true; }
};
}

static int g_try_compact = 0;
static long long g_compact_done = 0;
extern "C" void host_clip_set_compact(int on) { g_try_compact = on; g_compact_done = 0; }
extern "C" long long host_clip_compact_count(void) { return g_compact_done; }
static int g_try_fast = 0;
static long long g_fast_done = 0;
extern "C" void host_clip_set_fast(int on) { g_try_fast = on; g_fast_done = 0; }
extern "C" long long host_clip_fast_count(void) { return g_fast_done; }

extern "C" int host_clip_batch(int npairs,
                   const int64_t *ax, const int64_t *ay, const int32_t *aoff,
                   const int64_t *bx, const int64_t *by, const int32_t *boff,
                   int method, int fill_s, int fill_c,
                   int64_t *outx, int64_t *outy, int64_t out_cap,
                   int32_t *ring_off, int64_t ring_cap,
                   int32_t *pair_ring_off, int32_t *status) {
  Sink s; s.ox = outx; s.oy = outy; s.cap_v = out_cap; s.ring_off = ring_off; s.cap_r = ring_cap; s.nv = 0; s.nr = 0;
  ring_off[0] = 0; pair_ring_off[0] = 0;
  std::vector<char> ws;
  for (int p = 0; p < npairs; ++p) {
    int na = aoff[p + 1] - aoff[p], nb = boff[p + 1] - boff[p];
    RingView a = {ax + aoff[p], ay + aoff[p]}, b = {bx + boff[p], by + boff[p]};
    int64_t nv0 = s.nv, nr0 = s.nr;
    int st = szv::ST_WORKSPACE;
    // single-minimum fast path first (intersection, even-odd), as k_narrow does; ST_BAIL falls through
    if (g_try_fast && method == szv::OP_INTERSECTION && fill_s == 0 && fill_c == 0) {
      szx::FastBlock blk; szv::Pt deq[64];
      szx::FastWsT<szx::FastBlock> fw = {&blk, deq, 32};
      szx::RingPtr pa = {(const szv::i64 *)a.x, (const szv::i64 *)a.y}, pb = {(const szv::i64 *)b.x, (const szv::i64 *)b.y};
      int fs = szx::clip_monotone(fw, pa, na, pb, nb, s);
      if (fs == szx::ST_BAIL) { s.nv = nv0; s.nr = nr0; }
      else { st = fs; ++g_fast_done; }
    }
    // compact tier first (even-odd only), as the GPU's LDS path does; anything it cannot hold falls through
    if (st == szv::ST_WORKSPACE && g_try_compact && fill_s == 0 && fill_c == 0) {
      ws.resize(szv::workspace_bytes_compact() + 16);
      void *w = (void *)(((uintptr_t)ws.data() + 15) & ~(uintptr_t)15);
      st = szv::clip_rings_compact(w, a, na, b, nb, method, s);
      if (st != szv::ST_WORKSPACE && st != szv::ST_OUT_CAPACITY) ++g_compact_done;
    }
    // capacity tiers, as the GPU path retries overflowing pairs with a larger workspace
    for (int cap = na + nb, tier = 0; (st == szv::ST_WORKSPACE) && tier < 4; ++tier, cap *= 8) {
      ws.resize(szv::workspace_bytes(cap) + 16);
      void *w = (void *)(((uintptr_t)ws.data() + 15) & ~(uintptr_t)15);
      s.nv = nv0; s.nr = nr0;
      st = szv::clip_rings(w, cap, a, na, b, nb, method, fill_s, fill_c, s);
      if (st != szv::ST_WORKSPACE) break;
    }
    if (st != 0) { s.nv = nv0; s.nr = nr0; }
    if (st == szv::ST_OUT_CAPACITY) return 3;
    status[p] = st;
    pair_ring_off[p + 1] = (int32_t)s.nr;
  }
  return 0;
}
