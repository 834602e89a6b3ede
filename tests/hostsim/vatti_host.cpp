// TEST-ONLY: compiles the device sweep engine (subzero_amd/csrc/vatti.hpp) for the host with g++,
// so the CPU test-suite can check the algorithm the GPU kernels run against the reference Clipper
// (oracle/_ref) without a GPU.  Not part of the product; nothing in subzero_amd/ loads this.
#include "../../subzero_amd/csrc/vatti.hpp"
#include "../../subzero_amd/csrc/convex.hpp"
#include "../../subzero_amd/csrc/trim.hpp"
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

// rings trimmed to their common box first (trim.hpp), as the step's general path does for intersections
static int g_try_trim = 0;
static long long g_trim_before = 0, g_trim_after = 0, g_trim_declined = 0;
extern "C" void host_clip_set_trim(int on) { g_try_trim = on; g_trim_before = g_trim_after = g_trim_declined = 0; }
extern "C" void host_clip_trim_counts(long long *before, long long *after, long long *declined) {
  *before = g_trim_before; *after = g_trim_after; *declined = g_trim_declined;
}

// the records of the sweep inside a byte budget first (vatti.hpp caps_budget: the LDS slice classes of the clip batch's
// sweep pass), in a buffer of exactly that size with a canary behind it
static int g_budget = 0;
static long long g_budget_done = 0, g_budget_outgrown = 0;
extern "C" void host_clip_set_budget(int bytes) { g_budget = bytes; g_budget_done = g_budget_outgrown = 0; }
extern "C" void host_clip_budget_counts(long long *done, long long *outgrown) { *done = g_budget_done; *outgrown = g_budget_outgrown; }

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
    std::vector<int64_t> tx1, ty1, tx2, ty2;
    if (g_try_trim && method == szv::OP_INTERSECTION && fill_s == 0 && fill_c == 0) {
      tx1.assign(a.x, a.x + na); ty1.assign(a.y, a.y + na); tx2.assign(b.x, b.x + nb); ty2.assign(b.y, b.y + nb);
      int ta = na, tb = nb;
      g_trim_before += na + nb;
      if (szt::trim_pair((szv::i64 *)tx1.data(), (szv::i64 *)ty1.data(), ta, (szv::i64 *)tx2.data(), (szv::i64 *)ty2.data(), tb)) {
        a.x = tx1.data(); a.y = ty1.data(); b.x = tx2.data(); b.y = ty2.data(); na = ta; nb = tb;
      } else ++g_trim_declined;
      g_trim_after += na + nb;
    }
    int64_t nv0 = s.nv, nr0 = s.nr;
    int st = szv::ST_WORKSPACE;
    // single-minimum fast path first (intersection, even-odd), as k_narrow does; ST_BAIL falls through
    if (g_try_fast && method == szv::OP_INTERSECTION && fill_s == 0 && fill_c == 0) {
      // (the step keeps the rings as doubles in the block; this entry starts from int64 rings: VertsFixed)
      szx::FastBlock blk; szv::Pt deq[64];
      szx::FastWsT<szx::FastBlock> fw = {&blk, deq, 32};
      int fs = szx::ST_BAIL;
      if (na <= (int)szx::FastBlock::NV + 1 && nb <= (int)szx::FastBlock::NV + 1 && na >= 1 && nb >= 1) {
        int ma = na, mb = nb;     // a closing duplicate need not fit the block
        if (ma > 1 && a.x[ma - 1] == a.x[0] && a.y[ma - 1] == a.y[0]) --ma;
        if (mb > 1 && b.x[mb - 1] == b.x[0] && b.y[mb - 1] == b.y[0]) --mb;
        if (ma <= (int)szx::FastBlock::NV && mb <= (int)szx::FastBlock::NV) {
          bool exact = true;      // mode 2: hand the rings over in metres, as the step does (needs |v| < 2^53)
          if (g_try_fast == 2) {
            const int64_t lim = (int64_t)1 << 53;   // exact in a double, and 2^21 m is well inside the range VertsMetres needs
            for (int i = 0; i < ma; ++i) exact = exact && a.x[i] < lim && a.x[i] > -lim && a.y[i] < lim && a.y[i] > -lim;
            for (int i = 0; i < mb; ++i) exact = exact && b.x[i] < lim && b.x[i] > -lim && b.y[i] < lim && b.y[i] > -lim;
          }
          if (g_try_fast == 2 && exact) {
            for (int i = 0; i < ma; ++i) { blk.v.m.x[0][i] = (double)a.x[i] / 4294967296.0; blk.v.m.y[0][i] = (double)a.y[i] / 4294967296.0; }
            for (int i = 0; i < mb; ++i) { blk.v.m.x[1][i] = (double)b.x[i] / 4294967296.0; blk.v.m.y[1][i] = (double)b.y[i] / 4294967296.0; }
            szx::VertsMetres vm = {0, 0.0, 0.0};
            fs = szx::clip_monotone(fw, vm, ma, mb, s);
          } else {
            for (int i = 0; i < ma; ++i) blk.v.f[0][i] = a(i);
            for (int i = 0; i < mb; ++i) blk.v.f[1][i] = b(i);
            fs = szx::clip_monotone(fw, szx::VertsFixed(), ma, mb, s);
          }
        }
      }
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
    if (st == szv::ST_WORKSPACE && g_budget > 0) {
      szv::Caps c;
      if (szv::caps_budget<szv::WideLdsWave>(na + nb, (size_t)g_budget, c) && c.pts >= na + nb + 8) {
        std::vector<char> slice((size_t)g_budget + 16 + 64);
        char *w = (char *)(((uintptr_t)slice.data() + 15) & ~(uintptr_t)15);
        for (int k = 0; k < 64; ++k) w[g_budget + k] = (char)0x5A;
        if (szv::workspace_bytes_caps<szv::WideLdsWave>(c) > (size_t)g_budget) return 7;      // the capacities must fit the slice
        s.nv = nv0; s.nr = nr0;
        st = szv::clip_rings_t<szv::WideLdsWave>((void *)w, c, a, na, b, nb, method, fill_s, fill_c, s);
        for (int k = 0; k < 64; ++k) if (w[g_budget + k] != (char)0x5A) return 8;             // ... and the sweep stay inside it
        if (st == szv::ST_WORKSPACE) { ++g_budget_outgrown; s.nv = nv0; s.nr = nr0; }
        else ++g_budget_done;
      }
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

// ---- the offset mode as the library runs it (offset.hpp on the host + szv::offset_cleanup), with the clean-up on the
// host instead of the GPU: the CPU suite checks it against the reference's ClipperOffset (oracle/_ref ref_offset)
#include "../../subzero_amd/csrc/offset.hpp"
namespace {
struct HostPaths {
  const std::vector<szo::OPath> *p;
  struct Ring { const szo::OPath *r; szv::Pt operator()(int i) const { szv::Pt q; q.x = (*r)[(size_t)i].X; q.y = (*r)[(size_t)i].Y; return q; } };
  int count() const { return (int)p->size(); }
  int len(int k) const { return (int)(*p)[(size_t)k].size(); }
  Ring ring(int k) const { Ring g = {&(*p)[(size_t)k]}; return g; }
};
}
extern "C" int host_offset(const int64_t *x, const int64_t *y, const int32_t *ring_off, int nrings, double delta, int join, double info,
                           int64_t *outx, int64_t *outy, int64_t out_cap, int32_t *out_ring_off, int64_t ring_cap, int32_t *nrings_out) {
  szo::Offsetter co;
  if (info > 0 && join == 1) co.ArcTolerance = info;
  if (info > 0 && join == 2) co.MiterLimit = info;
  for (int r = 0; r < nrings; ++r) co.add_path(x + ring_off[r], y + ring_off[r], ring_off[r + 1] - ring_off[r]);
  co.fix_orientations();
  co.do_offset(delta, join);
  HostPaths hp = {&co.dest};
  int total = 4;
  for (size_t i = 0; i < co.dest.size(); ++i) total += (int)co.dest[i].size();
  int st = szv::ST_WORKSPACE;
  std::vector<char> ws;
  Sink s;
  for (int cap = total, tier = 0; st == szv::ST_WORKSPACE && tier < 5; ++tier, cap *= 8) {
    ws.resize(szv::workspace_bytes(cap) + 16);
    void *w = (void *)(((uintptr_t)ws.data() + 15) & ~(uintptr_t)15);
    s.ox = outx; s.oy = outy; s.cap_v = out_cap; s.ring_off = out_ring_off; s.cap_r = ring_cap; s.nv = 0; s.nr = 0;
    out_ring_off[0] = 0;
    st = szv::offset_cleanup(w, cap, hp, delta > 0, s);
  }
  *nrings_out = (int32_t)s.nr;
  return st;
}
