// fracture.hip -- szFractureDeform: the "slight permanent deformation" a floe receives when it fractures
// (Physical_Processes/fracture_floe.m:14-52), on the device-resident floes and contact rows.
//
// For every listed floe: the contact row of largest overlap among its floe partners (:17-23), the overlap region
// polyclip(floe, partner, 'int') (:30), the partner pushed into the floe by half the distance from the region's
// centroid to its outline, along the contact force (:33-39), and the floe cut back by polyclip(floe, partner, 'dif')
// (:40); when more than 90 % of the floe is left, its first piece becomes the floe's new outline, centroid and area
// (:43-50).  Both clips are launches of the batched clip kernel (clip.hip) on operands put together on the device;
// the few scalars in between are one thread per floe.  The Voronoi tessellation that follows (:54-124: rand,
// polybnd_voronoi, polyshape intersections, new floes) reshapes the floe array and stays on the MATLAB side.
//
// centroid(polyshape(X,Y)) and polyarea are restated as in contact.hpp (shoelace relative to the first vertex;
// polyarea.m's own expression); p_poly_dist is dist_to_ring.
#include "step_state.hpp"
#include "clip_batch.hpp"

namespace {

struct Pick { int partner; int pad; double fx, fy; };   // partner: 0-based floe, -1 none

// :14-27  row of largest overlap among the rows with a finite partner; partner must be a real floe (:26)
__global__ void k_frac_pick(Floes F, Work W, const int *ids, int nsel, Pick *pick) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nsel) return;
  const int i = ids[s] - 1;
  Pick p; p.partner = -1; p.pad = 0; p.fx = p.fy = 0;
  bool have = false;
  double best = 0, clip = 0;
  for (i64 r = W.row_off[i]; r < W.row_off[i + 1]; ++r) {
    const double *row = W.rows + 7 * r;
    if (szc::is_inf(row[0])) continue;                        // :19-21
    if (!have || row[6] > best) { have = true; best = row[6]; clip = row[0]; p.fx = row[1]; p.fy = row[2]; }   // [~,k] = max(a(:,7)): the first of equals
  }
  if (have && clip < (double)(F.n + 1) && clip >= 1) p.partner = (int)clip - 1;   // :26 (ghost partners: no deformation)
  pick[s] = p;
}

// operands of the clips: ring of the floe (c_alpha + position, as stored: closed) and of its partner, in metres
__global__ void k_frac_rings(Floes F, const int *ids, int nsel, const Pick *pick, const int *aoff, const int *boff,
                             double *ax, double *ay, double *bx, double *by) {
  int s = blockIdx.x;
  if (s >= nsel) return;
  const int i = ids[s] - 1, j = pick[s].partner;
  const int v0 = F.vert_off[i], nv = F.vert_off[i + 1] - v0;
  for (int k = threadIdx.x; k < nv; k += blockDim.x) { ax[aoff[s] + k] = F.cax[v0 + k] + F.Xi[i]; ay[aoff[s] + k] = F.cay[v0 + k] + F.Yi[i]; }
  if (j < 0) return;
  const int w0 = F.vert_off[j], nw = F.vert_off[j + 1] - w0;
  for (int k = threadIdx.x; k < nw; k += blockDim.x) { bx[boff[s] + k] = F.cax[w0 + k] + F.Xi[j]; by[boff[s] + k] = F.cay[w0 + k] + F.Yi[j]; }
}

// a result ring of the clip kernel as dist_to_ring reads it (closed by its first vertex, p_poly_dist.m:133-141)
struct OutRing {
  const i64 *x, *y; int n;
  __device__ int len() const { return n + 1; }
  __device__ double X(int k) const { return szc::from_fixed(x[k >= n ? 0 : k]); }
  __device__ double Y(int k) const { return szc::from_fixed(y[k >= n ? 0 : k]); }
};

// :31-39  first region of the overlap -> centroid, distance to its outline, the partner's ring moved along the force
__global__ void k_frac_shift(szb::ClipArgs a, int nsel, const Pick *pick, const int *boff, double *bx, double *by, int *stage) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nsel) return;
  stage[s] = 0;
  if (pick[s].partner < 0) return;
  if (a.status[s] != szv::ST_OK) { stage[s] = -1; return; }                          // the mex would have raised "Clipper Error."
  if (a.nrings[s] < 1) return;                                                       // :31 XInew empty
  const int *ro = a.oroff + a.or_base[s];
  const i64 *rx = a.ovx + a.ov_base[s] + ro[0], *ry = a.ovy + a.ov_base[s] + ro[0];
  const int n = ro[1] - ro[0];
  double xm, ym;
  szc::region_centroid(rx, ry, n, xm, ym);                                // :32
  OutRing ring = {rx, ry, n};
  const double d = szc::dist_to_ring(xm, ym, ring);                        // :33 (|d_min1|)
  const double fx = pick[s].fx, fy = pick[s].fy;
  const double Fn = ::sqrt(fx * fx + fy * fy);                              // :34 vecnorm
  const double xs = fx * d / 2 / Fn, ys = fy * d / 2 / Fn;                  // :35-36
  const int nb = boff[s + 1] - boff[s];
  for (int k = 0; k < nb; ++k) { bx[boff[s] + k] = bx[boff[s] + k] + xs; by[boff[s] + k] = by[boff[s] + k] + ys; }   // :37
  stage[s] = 1;
}

// :41-50  first piece of the difference: kept when it holds more than 90 % of the floe's area
__global__ void k_frac_finish(szb::ClipArgs a, Floes F, const int *ids, int nsel, const int *stage, const int *ooff, int *changed,
                              double *nXi, double *nYi, double *narea, int *nlen, double *ox, double *oy) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nsel) return;
  changed[s] = 0; nlen[s] = 0;
  if (stage[s] < 0 || (stage[s] > 0 && a.status[s] != szv::ST_OK)) { changed[s] = -2; return; }
  if (!stage[s] || a.nrings[s] < 1) return;
  const int *ro = a.oroff + a.or_base[s];
  const i64 *rx = a.ovx + a.ov_base[s] + ro[0], *ry = a.ovy + a.ov_base[s] + ro[0];
  const int n = ro[1] - ro[0];
  const double Anew = szc::region_polyarea(rx, ry, n);                     // :43
  if (!(Anew / F.area[ids[s] - 1] > 0.9)) return;                          // :44
  if (n > ooff[s + 1] - ooff[s]) { changed[s] = -1; return; }              // (cannot happen: the slot holds 2 (na + nb) + 8 vertices)
  double xm, ym;
  szc::region_centroid(rx, ry, n, xm, ym);                                // :45
  nXi[s] = xm; nYi[s] = ym; narea[s] = Anew;                               // :46-47
  for (int k = 0; k < n; ++k) { ox[ooff[s] + k] = szc::from_fixed(rx[k]) - xm; oy[ooff[s] + k] = szc::from_fixed(ry[k]) - ym; }   // :48
  nlen[s] = n; changed[s] = 1;
}

template <class T>
int fetch(szContext *ctx, std::vector<T> &h, const void *d, size_t n) {
  h.resize(n ? n : 1);
  if (n) SZ_HIP(ctx, hipMemcpyAsync(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
  return SZ_OK;
}

} // namespace

extern "C" int szFractureDeform(szContext *ctx, const int32_t *ids, int nsel, int32_t *changed, double *Xi, double *Yi, double *area,
                                int32_t *ring_off, double *cx, double *cy, int64_t cap) {
  if (!ctx || nsel < 0 || !ring_off || (nsel > 0 && (!ids || !changed || !Xi || !Yi || !area))) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S || !S->stepped) return sz_fail(ctx, SZ_ERR_STATE, "szFractureDeform: no step taken (the contact rows of a step are the input)");
  ring_off[0] = 0;
  if (nsel == 0) return SZ_OK;
  const int n = S->n;
  for (int s = 0; s < nsel; ++s)
    if (ids[s] < 1 || ids[s] > n) return sz_fail(ctx, SZ_ERR_BAD_ARG, "szFractureDeform: ids[%d] = %d outside 1..%d", s, (int)ids[s], n);
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  int rc = SZ_OK;
  DevBuf &d_ids = S->post[40]; DevBuf &d_pick = S->post[41]; DevBuf &d_aoff = S->post[42]; DevBuf &d_boff = S->post[43]; DevBuf &d_stage = S->post[44]; DevBuf &d_ooff = S->post[45]; DevBuf &d_chg = S->post[46]; DevBuf &d_sc = S->post[47]; DevBuf &d_len = S->post[48]; DevBuf &d_ox = S->post[49]; DevBuf &d_oy = S->post[50];
  auto cleanup = [&]() { (void)d_ids; (void)d_pick; (void)d_aoff; (void)d_boff; (void)d_stage; (void)d_ooff;
                         (void)d_chg; (void)d_sc; (void)d_len; (void)d_ox; (void)d_oy; };
#define FR_TRY(expr) do { if ((rc = (expr))) { cleanup(); return rc; } } while (0)
#define FR_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return sz_fail(ctx, SZ_ERR_HIP, "szFractureDeform: %s", hipGetErrorString(e_)); } } while (0)
  // (every transfer through the context's pinned staging buffers, one block per direction and phase: a copy from or to
  // pageable memory is staged by the runtime, tens of microseconds each -- there were ten of them)
  FR_TRY(sz_reserve(ctx, d_ids, (size_t)nsel * 4)); FR_TRY(sz_reserve(ctx, d_pick, (size_t)nsel * sizeof(Pick)));
  FR_TRY(sz_stage_reserve(ctx, ctx->h_in, (size_t)3 * (nsel + 4) * 4 + 64)); FR_TRY(sz_stage_reserve(ctx, ctx->h_out, (size_t)nsel * sizeof(Pick)));
  memcpy(ctx->h_in.p, ids, (size_t)nsel * 4);
  FR_HIP(hipMemcpyAsync(d_ids.p, ctx->h_in.p, (size_t)nsel * 4, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_frac_pick, grid_for(nsel, 64), dim3(64), 0, st, S->F, S->W, (const int *)d_ids.p, nsel, (Pick *)d_pick.p);
  FR_HIP(hipMemcpyAsync(ctx->h_out.p, d_pick.p, (size_t)nsel * sizeof(Pick), hipMemcpyDeviceToHost, st));
  FR_HIP(hipStreamSynchronize(st));
  std::vector<Pick> pick((const Pick *)ctx->h_out.p, (const Pick *)ctx->h_out.p + nsel);
  // operand layout: the floe's stored ring, its partner's stored ring (none: empty)
  Packer offs;
  const size_t o_a = offs.add((size_t)(nsel + 1) * 4), o_b = offs.add((size_t)(nsel + 1) * 4), o_o = offs.add((size_t)(nsel + 1) * 4);
  int *aoff = (int *)(ctx->h_in.p + o_a), *boff = (int *)(ctx->h_in.p + o_b), *ooff = (int *)(ctx->h_in.p + o_o);   // (nothing of the first phase is pending: the stream is idle)
  aoff[0] = boff[0] = ooff[0] = 0;
  for (int s = 0; s < nsel; ++s) {
    const int na = S->h_vert_off[ids[s]] - S->h_vert_off[ids[s] - 1];
    const int nb = pick[s].partner >= 0 ? S->h_vert_off[pick[s].partner + 1] - S->h_vert_off[pick[s].partner] : 0;
    aoff[s + 1] = aoff[s] + na; boff[s + 1] = boff[s] + nb;
    ooff[s + 1] = ooff[s] + (nb ? 2 * (na + nb) + 8 : 0);
  }
  const i64 ta = aoff[nsel], tb = boff[nsel], to = ooff[nsel];
  // the results in one device block: ox | oy | Xi, Yi, area | changed | length
  Packer res;
  const size_t r_ox = res.add((size_t)to * 8 + 16), r_oy = res.add((size_t)to * 8 + 16), r_sc = res.add((size_t)nsel * 24), r_chg = res.add((size_t)nsel * 4), r_len = res.add((size_t)nsel * 4);
  FR_TRY(sz_reserve(ctx, d_aoff, offs.at));
  FR_TRY(sz_reserve(ctx, ctx->in_fax, (size_t)ta * 8 + 16)); FR_TRY(sz_reserve(ctx, ctx->in_fay, (size_t)ta * 8 + 16));
  FR_TRY(sz_reserve(ctx, ctx->in_fbx, (size_t)tb * 8 + 16)); FR_TRY(sz_reserve(ctx, ctx->in_fby, (size_t)tb * 8 + 16));
  FR_TRY(sz_reserve(ctx, d_stage, (size_t)nsel * 4)); FR_TRY(sz_reserve(ctx, d_ox, res.at));
  FR_TRY(sz_stage_reserve(ctx, ctx->h_out, res.at));
  FR_HIP(hipMemcpyAsync(d_aoff.p, ctx->h_in.p, offs.at, hipMemcpyHostToDevice, st));
  const int *const dv_aoff = (const int *)((char *)d_aoff.p + o_a), *const dv_boff = (const int *)((char *)d_aoff.p + o_b), *const dv_ooff = (const int *)((char *)d_aoff.p + o_o);
  double *ax = (double *)ctx->in_fax.p, *ay = (double *)ctx->in_fay.p, *bx = (double *)ctx->in_fbx.p, *by = (double *)ctx->in_fby.p;
  hipLaunchKernelGGL(k_frac_rings, dim3((unsigned)nsel), dim3(64), 0, st, S->F, (const int *)d_ids.p, nsel, (const Pick *)d_pick.p,
                     dv_aoff, dv_boff, ax, ay, bx, by);
  FR_HIP(hipGetLastError());
  // the two clips; a floe without a partner is an empty second operand: the clip of it is skipped by its stage flag
  szb::ClipArgs a;
  memset(&a, 0, sizeof(a));
  FR_TRY(szb::prepare(ctx, nsel, ta + tb, a));
  a.fax = ax; a.fay = ay; a.fbx = bx; a.fby = by; a.aoff = dv_aoff; a.boff = dv_boff;
  a.method = SZ_CLIP_INTERSECTION; a.fill_s = SZ_FILL_EVENODD; a.fill_c = SZ_FILL_EVENODD;
  FR_TRY(szb::run(ctx, a, nullptr));
  hipLaunchKernelGGL(k_frac_shift, grid_for(nsel, 64), dim3(64), 0, st, a, nsel, (const Pick *)d_pick.p, dv_boff, bx, by, (int *)d_stage.p);
  FR_HIP(hipGetLastError());
  szb::ClipArgs b;
  memset(&b, 0, sizeof(b));
  FR_TRY(szb::prepare(ctx, nsel, ta + tb, b));
  b.fax = ax; b.fay = ay; b.fbx = bx; b.fby = by; b.aoff = dv_aoff; b.boff = dv_boff;
  b.method = SZ_CLIP_DIFFERENCE; b.fill_s = SZ_FILL_EVENODD; b.fill_c = SZ_FILL_EVENODD;
  FR_TRY(szb::run(ctx, b, nullptr));
  double *sc = (double *)((char *)d_ox.p + r_sc);
  hipLaunchKernelGGL(k_frac_finish, grid_for(nsel, 64), dim3(64), 0, st, b, S->F, (const int *)d_ids.p, nsel, (const int *)d_stage.p, dv_ooff,
                     (int *)((char *)d_ox.p + r_chg), sc, sc + nsel, sc + 2 * (size_t)nsel, (int *)((char *)d_ox.p + r_len), (double *)((char *)d_ox.p + r_ox), (double *)((char *)d_ox.p + r_oy));
  FR_HIP(hipGetLastError());
  FR_HIP(hipMemcpyAsync(ctx->h_out.p, d_ox.p, res.at, hipMemcpyDeviceToHost, st));
  FR_HIP(hipStreamSynchronize(st));
  cleanup();
  const int *chg = (const int *)(ctx->h_out.p + r_chg), *len = (const int *)(ctx->h_out.p + r_len);
  const double *scal = (const double *)(ctx->h_out.p + r_sc), *hx = (const double *)(ctx->h_out.p + r_ox), *hy = (const double *)(ctx->h_out.p + r_oy);
  int64_t at = 0;
  for (int s = 0; s < nsel; ++s) {
    if (chg[s] == -2) return sz_fail(ctx, SZ_ERR_HIP, "szFractureDeform: floe %d: the reference's \"Clipper Error.\" condition", (int)ids[s]);
    if (chg[s] < 0) return sz_fail(ctx, SZ_ERR_WORKSPACE, "szFractureDeform: the piece of floe %d outgrew its slot", (int)ids[s]);
    changed[s] = chg[s];
    Xi[s] = chg[s] ? scal[s] : 0; Yi[s] = chg[s] ? scal[(size_t)nsel + s] : 0; area[s] = chg[s] ? scal[2 * (size_t)nsel + s] : 0;
    if (chg[s]) {
      if (at + len[s] > cap || !cx || !cy) return sz_fail(ctx, SZ_ERR_CAPACITY, "szFractureDeform: %lld vertices or more, capacity %lld", (long long)(at + len[s]), (long long)cap);
      for (int k = 0; k < len[s]; ++k) { cx[at + k] = hx[(size_t)ooff[s] + k]; cy[at + k] = hy[(size_t)ooff[s] + k]; }
      at += len[s];
    }
    ring_off[s + 1] = (int32_t)at;
  }
  return SZ_OK;
}

// ---- FloeSimplify.m:14-34: the floes a floe may touch, by bounding circles, in ascending floe number ---------------
namespace {
__global__ void __launch_bounds__(64) k_potential(Floes F, const int *ids, int nsel, int cap_each, int *cnt, int *list) {
  const int s = blockIdx.x;
  if (s >= nsel) return;
  const int i = ids[s] - 1, lane = threadIdx.x;
  const double xi = F.Xi[i], yi = F.Yi[i], ri = F.rmax[i];
  int kept = 0;
  for (int base = 0; base < F.n; base += 64) {
    const int j = base + lane;
    bool hit = false;
    if (j < F.n && F.alive[j]) {
      const double dx = xi - F.Xi[j], dy = yi - F.Yi[j];
      const double d = ::sqrt(dx * dx + dy * dy);                 // :21
      hit = d > 1 && d < (ri + F.rmax[j]);
    }
    const unsigned long long m = __ballot(hit);
    if (hit) {
      const int at = kept + __popcll(m & ((1ULL << lane) - 1ULL));
      if (at < cap_each) list[(size_t)s * cap_each + at] = j + 1;
    }
    kept += __popcll(m);
  }
  if (lane == 0) cnt[s] = kept;
}
} // namespace

extern "C" int szPotentialInteractions(szContext *ctx, const int32_t *ids, int nsel, int32_t *off, int32_t *list, int64_t cap) {
  if (!ctx || nsel < 0 || !off || (nsel > 0 && !ids)) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S) return sz_fail(ctx, SZ_ERR_STATE, "szPotentialInteractions: no floes uploaded");
  off[0] = 0;
  if (nsel == 0) return SZ_OK;
  for (int s = 0; s < nsel; ++s)
    if (ids[s] < 1 || ids[s] > S->n) return sz_fail(ctx, SZ_ERR_BAD_ARG, "szPotentialInteractions: ids[%d] = %d outside 1..%d", s, (int)ids[s], S->n);
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  int cap_each = 64, rc;
  DevBuf &d_ids = S->post[51]; DevBuf &d_cnt = S->post[52]; DevBuf &d_list = S->post[53];
  auto cleanup = [&]() { (void)d_ids; (void)d_cnt; (void)d_list; };
  if ((rc = sz_reserve(ctx, d_ids, (size_t)nsel * 4)) || (rc = sz_reserve(ctx, d_cnt, (size_t)nsel * 4))) { cleanup(); return rc; }
  std::vector<int> cnt((size_t)nsel), h;
  for (;;) {      // (a floe with more neighbours than the room per floe: once more with what it needs)
    if ((rc = sz_reserve(ctx, d_list, (size_t)nsel * cap_each * 4))) { cleanup(); return rc; }
    hipError_t e = hipMemcpyAsync(d_ids.p, ids, (size_t)nsel * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) { hipLaunchKernelGGL(k_potential, dim3((unsigned)nsel), dim3(64), 0, st, S->F, (const int *)d_ids.p, nsel, cap_each, (int *)d_cnt.p, (int *)d_list.p); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(cnt.data(), d_cnt.p, (size_t)nsel * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { cleanup(); return sz_fail(ctx, SZ_ERR_HIP, "szPotentialInteractions: %s", hipGetErrorString(e)); }
    int most = 0;
    for (int s = 0; s < nsel; ++s) most = cnt[s] > most ? cnt[s] : most;
    if (most <= cap_each) break;
    cap_each = most;
  }
  h.resize((size_t)nsel * cap_each);
  hipError_t e = hipMemcpy(h.data(), d_list.p, h.size() * 4, hipMemcpyDeviceToHost);
  cleanup();
  if (e != hipSuccess) return sz_fail(ctx, SZ_ERR_HIP, "szPotentialInteractions: %s", hipGetErrorString(e));
  int64_t at = 0;
  for (int s = 0; s < nsel; ++s) {
    if (at + cnt[s] > cap || (cnt[s] && !list)) return sz_fail(ctx, SZ_ERR_CAPACITY, "szPotentialInteractions: %lld entries or more, capacity %lld", (long long)(at + cnt[s]), (long long)cap);
    for (int k = 0; k < cnt[s]; ++k) list[at + k] = h[(size_t)s * cap_each + k];
    at += cnt[s];
    off[s + 1] = (int32_t)at;
  }
  return SZ_OK;
}
