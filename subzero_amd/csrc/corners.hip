// corners.hip -- szCornerCandidates: corners.m:13-92 on the device-resident floes and contact rows.
//   k_corner_extent   per floe: max |x|, max |y| of its translated outline (the ghost rule of :21-47 reads them)
//   k_corner_flags    one wavefront per listed floe: a lane per vertex for the interior angle and the inpolygon
//                     tests against every partner of the floe's contact rows, then a lane per contact row for the
//                     nearest vertex (corners.hpp)
// The extended floe list (every floe, then the X images, then the Y images of floes and X images: :21-49) is put
// together on the host from the downloaded extents; partner numbers in the contact rows index it as they do in the
// reference (rows written by the step, list rebuilt from the positions after it).
#include "step_state.hpp"
#include "corners.hpp"

namespace {

struct CornerEnt { double x, y; int src, pad; };
struct Box5 { double x[5], y[5]; };

__global__ void k_corner_extent(Floes F, double *ext2) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F.n) return;
  const int v0 = F.vert_off[i], nv = F.vert_off[i + 1] - v0;
  double mx = 0, my = 0;
  for (int k = 0; k < nv; ++k) { mx = szc::maxd(mx, szs::absd(F.cax[v0 + k] + F.Xi[i])); my = szc::maxd(my, szs::absd(F.cay[v0 + k] + F.Yi[i])); }
  ext2[2 * i] = mx; ext2[2 * i + 1] = my;
}

struct DetAcos { __device__ double operator()(double a) const { return szm::acos_real(a); } };

__global__ void __launch_bounds__(64) k_corner_flags(Floes F, Work W, const int *ids, int nsub, int Nb, const CornerEnt *E, int N0, Box5 box,
                                                    const int *out_off, int *count, double *angles, unsigned char *da) {
  const int q = blockIdx.x;
  if (q >= nsub) return;
  const int i = ids[q] - 1;
  const i64 r0 = W.row_off[i], r1 = W.row_off[i + 1];
  if (q < Nb || r1 == r0) { if (threadIdx.x == 0) count[q] = 0; return; }     // :55-56
  const int v0 = F.vert_off[i];
  int n = F.vert_off[i + 1] - v0;
  const double *x = F.cax + v0, *y = F.cay + v0;
  if (n > 1 && x[0] == x[n - 1] && y[0] == y[n - 1]) --n;                      // :66-68
  if (threadIdx.x == 0) count[q] = n;
  const double X = F.Xi[i], Y = F.Yi[i];
  double *ang = angles + out_off[q];
  unsigned char *flag = da + out_off[q];
  for (int v = threadIdx.x; v < n; v += blockDim.x) {
    ang[v] = szk::interior_angle(v, x, y, n, DetAcos());
    const double px = x[v] + X, py = y[v] + Y;
    int in = 0;
    bool bnd = false;
    for (i64 r = r0; r < r1; ++r) {
      const double partner = W.rows[7 * r];
      if (partner > 1.7e308) { bnd = true; continue; }                         // :75 the boundary's rows carry Inf
      if (!(partner <= (double)N0) || partner < 1) continue;                   // :77
      const CornerEnt e = E[(int)partner - 1];
      const int p0 = F.vert_off[e.src], pn = F.vert_off[e.src + 1] - p0;
      in += szk::inpolygon_ring(px, py, F.cax + p0, F.cay + p0, pn, e.x, e.y) ? 1 : 0;   // :79-81
    }
    if (bnd) in += szk::inpolygon_ring(px, py, box.x, box.y, 5, 0.0, 0.0) ? 0 : 1;      // :83-86
    flag[v] = in > 0;
  }
  __syncthreads();
  for (i64 r = r0 + threadIdx.x; r < r1; r += blockDim.x) {                    // :60-61,73,88
    const double partner = W.rows[7 * r];
    if (!(partner <= (double)N0)) continue;
    flag[szk::nearest_vertex(W.rows[7 * r + 3], W.rows[7 * r + 4], x, y, n, X, Y)] = 1;
  }
}

// frac_corner.m for the plain case (corners.hpp frac_corner_plain): a thread per selected floe; every floe writes into the
// slots the host laid out from its vertex and corner counts
__global__ void k_corner_cuts(Floes F, const int *ids, int nsel, const int *voff, const unsigned char *grind_in, const int *slot_piece, const int *slot_ring,
                              const int *slot_tmp, unsigned char *grind_tmp, double *tmpx, double *tmpy, int *npieces, double *rec, int *ring_off, double *rx, double *ry) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nsel) return;
  const int i = ids[s] - 1;
  const int v0 = F.vert_off[i];
  int n = F.vert_off[i + 1] - v0;
  const double *x = F.cax + v0, *y = F.cay + v0;
  if (n > 1 && x[0] == x[n - 1] && y[0] == y[n - 1]) --n;                      // corners.m:66-68
  unsigned char *g = grind_tmp + voff[s];
  for (int k = 0; k < n; ++k) g[k] = grind_in[voff[s] + k];
  npieces[s] = szk::frac_corner_plain(x, y, n, g, F.mass[i], F.area[i], F.ksi_ice[i], F.Xi[i], F.Yi[i], tmpx + slot_tmp[s], tmpy + slot_tmp[s],
                                      rec + (size_t)slot_piece[s] * szk::PIECE_WORDS, rx + slot_ring[s], ry + slot_ring[s], ring_off + slot_piece[s] + s, DetAcos());
}

} // namespace

extern "C" int szCornerCuts(szContext *ctx, const int32_t *ids, int nsel, const int32_t *offsets, const uint8_t *grind, int32_t *npieces, int32_t *piece_off,
                            double *pieces, int64_t cap_pieces, int32_t *ring_off, double *rx, double *ry, int64_t cap_ring, double *kernel_ms) {
  if (!ctx || nsel < 0 || (nsel > 0 && (!ids || !offsets || !grind || !npieces)) || !piece_off || !ring_off) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S) return sz_fail(ctx, SZ_ERR_STATE, "szCornerCuts: no floes uploaded");
  const int n = S->n;
  piece_off[0] = 0; ring_off[0] = 0;
  if (kernel_ms) *kernel_ms = 0;
  if (nsel == 0) return SZ_OK;
  // slots: 1 + ng pieces, n + 1 + 5 ng ring vertices, 5 n + 4 scratch values per floe
  std::vector<int> sp((size_t)nsel + 1), sr((size_t)nsel + 1), stmp((size_t)nsel + 1);
  sp[0] = sr[0] = stmp[0] = 0;
  for (int s = 0; s < nsel; ++s) {
    if (ids[s] < 1 || ids[s] > n) return sz_fail(ctx, SZ_ERR_BAD_ARG, "szCornerCuts: ids[%d] = %d outside 1..%d", s, (int)ids[s], n);
    const int nv = S->h_vert_off[ids[s]] - S->h_vert_off[ids[s] - 1];
    if (offsets[s + 1] - offsets[s] < nv - 1) return sz_fail(ctx, SZ_ERR_BAD_ARG, "szCornerCuts: offsets[%d]: room for %d vertices, the outline has %d", s, (int)(offsets[s + 1] - offsets[s]), nv);
    int ng = 0;
    for (int k = offsets[s]; k < offsets[s + 1]; ++k) ng += grind[k] ? 1 : 0;
    sp[s + 1] = sp[s] + 1 + ng; sr[s + 1] = sr[s] + nv + 1 + 5 * ng; stmp[s + 1] = stmp[s] + 5 * nv + 4;
  }
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t nvt = (size_t)offsets[nsel];
  DevBuf &d_ids = S->post[0]; DevBuf &d_voff = S->post[1]; DevBuf &d_g = S->post[2]; DevBuf &d_gt = S->post[3]; DevBuf &d_sp = S->post[4]; DevBuf &d_sr = S->post[5]; DevBuf &d_st = S->post[6]; DevBuf &d_tx = S->post[7]; DevBuf &d_ty = S->post[8]; DevBuf &d_np = S->post[9]; DevBuf &d_rec = S->post[10]; DevBuf &d_ro = S->post[11]; DevBuf &d_rx = S->post[12]; DevBuf &d_ry = S->post[13];
  DevBuf *all[] = {&d_ids, &d_voff, &d_g, &d_gt, &d_sp, &d_sr, &d_st, &d_tx, &d_ty, &d_np, &d_rec, &d_ro, &d_rx, &d_ry};
  auto cleanup = [&]() { (void)all; };
  int rc;
  if ((rc = sz_reserve(ctx, d_ids, (size_t)nsel * 4)) || (rc = sz_reserve(ctx, d_voff, ((size_t)nsel + 1) * 4)) || (rc = sz_reserve(ctx, d_g, nvt + 8)) ||
      (rc = sz_reserve(ctx, d_gt, nvt + 8)) || (rc = sz_reserve(ctx, d_sp, ((size_t)nsel + 1) * 4)) || (rc = sz_reserve(ctx, d_sr, ((size_t)nsel + 1) * 4)) ||
      (rc = sz_reserve(ctx, d_st, ((size_t)nsel + 1) * 4)) || (rc = sz_reserve(ctx, d_tx, (size_t)stmp[nsel] * 8 + 8)) || (rc = sz_reserve(ctx, d_ty, (size_t)stmp[nsel] * 8 + 8)) ||
      (rc = sz_reserve(ctx, d_np, (size_t)nsel * 4)) || (rc = sz_reserve(ctx, d_rec, (size_t)sp[nsel] * szk::PIECE_WORDS * 8 + 8)) ||
      (rc = sz_reserve(ctx, d_ro, ((size_t)sp[nsel] + nsel + 1) * 4)) || (rc = sz_reserve(ctx, d_rx, (size_t)sr[nsel] * 8 + 8)) || (rc = sz_reserve(ctx, d_ry, (size_t)sr[nsel] * 8 + 8))) { cleanup(); return rc; }
  hipError_t he = hipMemcpyAsync(d_ids.p, ids, (size_t)nsel * 4, hipMemcpyHostToDevice, st);
  if (he == hipSuccess) he = hipMemcpyAsync(d_voff.p, offsets, ((size_t)nsel + 1) * 4, hipMemcpyHostToDevice, st);
  if (he == hipSuccess) he = hipMemcpyAsync(d_g.p, grind, nvt, hipMemcpyHostToDevice, st);
  if (he == hipSuccess) he = hipMemcpyAsync(d_sp.p, sp.data(), ((size_t)nsel + 1) * 4, hipMemcpyHostToDevice, st);
  if (he == hipSuccess) he = hipMemcpyAsync(d_sr.p, sr.data(), ((size_t)nsel + 1) * 4, hipMemcpyHostToDevice, st);
  if (he == hipSuccess) he = hipMemcpyAsync(d_st.p, stmp.data(), ((size_t)nsel + 1) * 4, hipMemcpyHostToDevice, st);
  if (he == hipSuccess) he = hipEventRecord(ctx->ev0, st);
  if (he == hipSuccess)
    hipLaunchKernelGGL(k_corner_cuts, grid_for(nsel, 64), dim3(64), 0, st, S->F, (const int *)d_ids.p, nsel, (const int *)d_voff.p, (const unsigned char *)d_g.p,
                       (const int *)d_sp.p, (const int *)d_sr.p, (const int *)d_st.p, (unsigned char *)d_gt.p, (double *)d_tx.p, (double *)d_ty.p, (int *)d_np.p,
                       (double *)d_rec.p, (int *)d_ro.p, (double *)d_rx.p, (double *)d_ry.p);
  if (he == hipSuccess) he = hipEventRecord(ctx->ev1, st);
  std::vector<int> h_np((size_t)nsel), h_ro((size_t)sp[nsel] + nsel + 1);
  std::vector<double> h_rec((size_t)sp[nsel] * szk::PIECE_WORDS + 1), h_rx((size_t)sr[nsel] + 1), h_ry((size_t)sr[nsel] + 1);
  if (he == hipSuccess) he = hipMemcpyAsync(h_np.data(), d_np.p, (size_t)nsel * 4, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipMemcpyAsync(h_ro.data(), d_ro.p, h_ro.size() * 4, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipMemcpyAsync(h_rec.data(), d_rec.p, (size_t)sp[nsel] * szk::PIECE_WORDS * 8, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipMemcpyAsync(h_rx.data(), d_rx.p, (size_t)sr[nsel] * 8, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipMemcpyAsync(h_ry.data(), d_ry.p, (size_t)sr[nsel] * 8, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipStreamSynchronize(st);
  float ms = 0;
  if (he == hipSuccess) he = hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  cleanup();
  if (he != hipSuccess) return sz_fail(ctx, SZ_ERR_HIP, "szCornerCuts: %s", hipGetErrorString(he));
  if (kernel_ms) *kernel_ms = ms;
  // pieces of the floes that were cut, back to back
  int64_t np = 0, nr = 0;
  for (int s = 0; s < nsel; ++s) {
    npieces[s] = h_np[s];
    const int *ro = h_ro.data() + sp[s] + s;
    for (int q = 0; q < h_np[s]; ++q) {
      const int m = ro[q + 1] - ro[q];
      if (np + 1 > cap_pieces || nr + m > cap_ring) return sz_fail(ctx, SZ_ERR_CAPACITY, "szCornerCuts: output capacity (%lld pieces, %lld vertices) is too small", (long long)cap_pieces, (long long)cap_ring);
      memcpy(pieces + np * szk::PIECE_WORDS, h_rec.data() + ((size_t)sp[s] + q) * szk::PIECE_WORDS, szk::PIECE_WORDS * 8);
      memcpy(rx + nr, h_rx.data() + sr[s] + ro[q], (size_t)m * 8); memcpy(ry + nr, h_ry.data() + sr[s] + ro[q], (size_t)m * 8);
      ++np; nr += m; ring_off[np] = (int32_t)nr;
    }
    piece_off[s + 1] = (int32_t)np;
  }
  return SZ_OK;
}

extern "C" int szCornerCandidates(szContext *ctx, const int32_t *ids, int nsub, int Nb, const double *c2b, int32_t *offsets, int32_t *count,
                                  double *angles, uint8_t *da, int64_t cap, int32_t *n_ext, double *kernel_ms) {
  if (!ctx || !c2b || !offsets || !count || nsub < 0 || (nsub > 0 && !ids)) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S || !S->stepped) return sz_fail(ctx, SZ_ERR_STATE, "szCornerCandidates: no step taken (the contact rows of a step are the input)");
  const int n = S->n;
  // where each listed floe's values go: the stored outline lengths, back to back
  int64_t need = 0;
  for (int q = 0; q < nsub; ++q) {
    if (ids[q] < 1 || ids[q] > n) return sz_fail(ctx, SZ_ERR_BAD_ARG, "szCornerCandidates: ids[%d] = %d outside 1..%d", q, (int)ids[q], n);
    offsets[q] = (int32_t)need;
    need += S->h_vert_off[ids[q]] - S->h_vert_off[ids[q] - 1];
  }
  offsets[nsub] = (int32_t)need;
  if (need > cap) return sz_fail(ctx, SZ_ERR_CAPACITY, "szCornerCandidates: %lld vertices, capacity %lld", (long long)need, (long long)cap);
  if (need > 0 && (!angles || !da)) return SZ_ERR_BAD_ARG;
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  int rc;
  DevBuf &d_ext = S->post[14]; DevBuf &d_ent = S->post[15]; DevBuf &d_ids = S->post[16]; DevBuf &d_off = S->post[17]; DevBuf &d_cnt = S->post[18]; DevBuf &d_ang = S->post[19]; DevBuf &d_da = S->post[20];
  auto cleanup = [&]() { (void)d_ext; (void)d_ent; (void)d_ids; (void)d_off; (void)d_cnt; (void)d_ang; (void)d_da; };
  if ((rc = sz_reserve(ctx, d_ext, (size_t)n * 16 + 16))) return rc;
  hipLaunchKernelGGL(k_corner_extent, grid_for(n, 256), dim3(256), 0, st, S->F, (double *)d_ext.p);
  std::vector<double> ext((size_t)2 * n), x((size_t)n), y((size_t)n);
  std::vector<int> alive((size_t)n);
  hipError_t he = hipMemcpyAsync(ext.data(), d_ext.p, ext.size() * 8, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipMemcpyAsync(x.data(), S->F.Xi, (size_t)n * 8, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipMemcpyAsync(y.data(), S->F.Yi, (size_t)n * 8, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipMemcpyAsync(alive.data(), S->F.alive, (size_t)n * 4, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipStreamSynchronize(st);
  if (he != hipSuccess) { cleanup(); return sz_fail(ctx, SZ_ERR_HIP, "szCornerCandidates: %s", hipGetErrorString(he)); }
  double Lx = c2b[0], Ly = c2b[1];
  for (int k = 1; k < 5; ++k) { Lx = szc::maxd(Lx, c2b[2 * k]); Ly = szc::maxd(Ly, c2b[2 * k + 1]); }
  // Floe0 with its images (:13-49)
  std::vector<CornerEnt> E((size_t)n);
  std::vector<double> ey((size_t)n);
  std::vector<int> live((size_t)n);
  for (int i = 0; i < n; ++i) { CornerEnt e = {x[i], y[i], i, 0}; E[i] = e; ey[i] = ext[2 * i + 1]; live[i] = alive[i]; }
  for (int i = 0; i < n; ++i)
    if (alive[i] && ext[2 * i] > Lx) { CornerEnt g = E[i]; g.x = x[i] - 2 * Lx * sgn(x[i]); E.push_back(g); ey.push_back(ey[i]); live.push_back(1); }
  const size_t n2 = E.size();
  for (size_t i = 0; i < n2; ++i)
    if (live[i] && ey[i] > Ly) { CornerEnt g = E[i]; g.y = E[i].y - 2 * Ly * sgn(E[i].y); E.push_back(g); }
  const int N0 = (int)E.size();
  if (n_ext) *n_ext = N0;
  Box5 box;
  for (int k = 0; k < 5; ++k) { box.x[k] = c2b[2 * k]; box.y[k] = c2b[2 * k + 1]; }
  if (nsub == 0) { cleanup(); if (kernel_ms) *kernel_ms = 0; return SZ_OK; }
  if ((rc = sz_reserve(ctx, d_ent, (size_t)N0 * sizeof(CornerEnt))) || (rc = sz_reserve(ctx, d_ids, (size_t)nsub * 4)) ||
      (rc = sz_reserve(ctx, d_off, (size_t)(nsub + 1) * 4)) || (rc = sz_reserve(ctx, d_cnt, (size_t)nsub * 4)) ||
      (rc = sz_reserve(ctx, d_ang, (size_t)need * 8 + 8)) || (rc = sz_reserve(ctx, d_da, (size_t)need + 8))) { cleanup(); return rc; }
  he = hipMemcpyAsync(d_ent.p, E.data(), (size_t)N0 * sizeof(CornerEnt), hipMemcpyHostToDevice, st);
  if (he == hipSuccess) he = hipMemcpyAsync(d_ids.p, ids, (size_t)nsub * 4, hipMemcpyHostToDevice, st);
  if (he == hipSuccess) he = hipMemcpyAsync(d_off.p, offsets, (size_t)(nsub + 1) * 4, hipMemcpyHostToDevice, st);
  if (he == hipSuccess) he = hipMemsetAsync(d_ang.p, 0, (size_t)need * 8 + 8, st);
  if (he == hipSuccess) he = hipMemsetAsync(d_da.p, 0, (size_t)need + 8, st);
  if (he == hipSuccess) he = hipEventRecord(ctx->ev0, st);
  if (he == hipSuccess)
    hipLaunchKernelGGL(k_corner_flags, dim3((unsigned)nsub), dim3(64), 0, st, S->F, S->W, (const int *)d_ids.p, nsub, Nb, (const CornerEnt *)d_ent.p, N0, box,
                       (const int *)d_off.p, (int *)d_cnt.p, (double *)d_ang.p, (unsigned char *)d_da.p);
  if (he == hipSuccess) he = hipEventRecord(ctx->ev1, st);
  if (he == hipSuccess) he = hipMemcpyAsync(count, d_cnt.p, (size_t)nsub * 4, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess && need) he = hipMemcpyAsync(angles, d_ang.p, (size_t)need * 8, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess && need) he = hipMemcpyAsync(da, d_da.p, (size_t)need, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipStreamSynchronize(st);
  float ms = 0;
  if (he == hipSuccess) he = hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  cleanup();
  if (he != hipSuccess) return sz_fail(ctx, SZ_ERR_HIP, "szCornerCandidates: %s", hipGetErrorString(he));
  if (kernel_ms) *kernel_ms = ms;
  return SZ_OK;
}
