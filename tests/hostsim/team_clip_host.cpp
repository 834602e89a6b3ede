// TEST-ONLY: the team form of the single-minimum intersection (subzero_amd/csrc/teamclip.hpp) compiled for the host:
// the G lanes of a team run as fibers (team_host.hpp).  The CPU test-suite checks it bit for bit against the
// reference's Clipper build (oracle/_ref).  Not part of the product; nothing in subzero_amd/ loads this.
static long long g_team_bails[128];
#define SZTM_BAIL_PROBE(site) (++g_team_bails[(site) & 127])
#include "team_host.hpp"
#include "../../subzero_amd/csrc/teamclip.hpp"
#include <stdint.h>

extern "C" void host_team_bail_sites(long long *out) { for (int i = 0; i < 128; ++i) out[i] = g_team_bails[i]; }

namespace {
template <int G>
struct ClipJob {
  sztm::ClipScratch S;
  szv::i64 ox[sztm::T_CAPV], oy[sztm::T_CAPV];
  int na, nb, st[G], out_n[G];
  teamhost::Runtime *rt;
  static void lane(void *self, int l) {
    ClipJob *J = (ClipJob *)self;
    sztm::Team<G> T; T.lane = l; T.rt = J->rt;
    J->st[l] = sztm::team_clip(T, &J->S, J->na, J->nb, J->ox, J->oy, &J->out_n[l]);
  }
};

template <int G>
int clip_batch(int npairs, const int64_t *ax, const int64_t *ay, const int32_t *aoff, const int64_t *bx, const int64_t *by,
               const int32_t *boff, int64_t *outx, int64_t *outy, int64_t out_cap, int32_t *ring_off, int64_t ring_cap,
               int32_t *pair_ring_off, int32_t *status) {
  static thread_local teamhost::Runtime rt;
  static thread_local ClipJob<G> J;
  J.rt = &rt;
  int64_t nv = 0, nr = 0;
  ring_off[0] = 0; pair_ring_off[0] = 0;
  for (int p = 0; p < npairs; ++p) {
    int na = aoff[p + 1] - aoff[p], nb = boff[p + 1] - boff[p];
    const int64_t *pax = ax + aoff[p], *pay = ay + aoff[p], *pbx = bx + boff[p], *pby = by + boff[p];
    // a closing duplicate need not fit the block (the step stages rings without it, contact.hpp pair_contact_fast)
    if (na > 1 && pax[na - 1] == pax[0] && pay[na - 1] == pay[0]) --na;
    if (nb > 1 && pbx[nb - 1] == pbx[0] && pby[nb - 1] == pby[0]) --nb;
    int st = szx::ST_BAIL;
    if (na <= (int)sztm::T_NV && nb <= (int)sztm::T_NV && na >= 1 && nb >= 1) {
      for (int i = 0; i < na; ++i) { J.S.fx[0][i] = pax[i]; J.S.fy[0][i] = pay[i]; }
      for (int i = 0; i < nb; ++i) { J.S.fx[1][i] = pbx[i]; J.S.fy[1][i] = pby[i]; }
      J.na = na; J.nb = nb;
      rt.run(G, &ClipJob<G>::lane, &J);
      if (!teamhost::exited_together(rt)) { fprintf(stderr, "team_clip: the lanes of a team did not return together (pair %d)\n", p); abort(); }
      for (int l = 1; l < G; ++l)
        if (J.st[l] != J.st[0] || J.out_n[l] != J.out_n[0]) { fprintf(stderr, "team_clip: lanes disagree on the result (pair %d)\n", p); abort(); }
      st = J.st[0];
      if (st == szv::ST_OK && J.out_n[0] > 0) {
        if (nv + J.out_n[0] > out_cap || nr + 1 > ring_cap) return 3;
        for (int i = 0; i < J.out_n[0]; ++i) { outx[nv] = J.ox[i]; outy[nv] = J.oy[i]; ++nv; }
        ++nr; ring_off[nr] = (int32_t)nv;
      }
    }
    status[p] = st;
    pair_ring_off[p + 1] = (int32_t)nr;
  }
  return 0;
}
}

extern "C" int host_team_clip_batch(int G, int npairs, const int64_t *ax, const int64_t *ay, const int32_t *aoff, const int64_t *bx,
                                    const int64_t *by, const int32_t *boff, int64_t *outx, int64_t *outy, int64_t out_cap,
                                    int32_t *ring_off, int64_t ring_cap, int32_t *pair_ring_off, int32_t *status) {
  if (G == 16) return clip_batch<16>(npairs, ax, ay, aoff, bx, by, boff, outx, outy, out_cap, ring_off, ring_cap, pair_ring_off, status);
  if (G == 32) return clip_batch<32>(npairs, ax, ay, aoff, bx, by, boff, outx, outy, out_cap, ring_off, ring_cap, pair_ring_off, status);
  return -1;
}
