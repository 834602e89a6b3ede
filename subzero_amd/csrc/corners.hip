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

// Floe0 with its images (corners.m:13-49) in the reference's order -- every floe, then the X images of the floes whose
// outline reaches past Lx, then the Y images of floes and X images alike (an X image inherits its floe's y extent) --
// by ONE workgroup: a chunk of floes per thread, three counts per chunk (X images, Y images of floes, Y images of X
// images), one scan over the 1024 chunks, and every image lands where the reference's two loops append it.
// hdr[0] = N0, the length of the list.
__global__ void __launch_bounds__(1024) k_corner_entities(Floes F, const double *ext2, double Lx, double Ly, CornerEnt *E, int *hdr) {
  __shared__ int cx[1024], cy[1024], cxy[1024];
  const int n = F.n, t = threadIdx.x, chunk = (n + 1023) / 1024;
  const int a = (t * chunk < n) ? t * chunk : n, b = (a + chunk < n) ? a + chunk : n;
  int kx = 0, ky = 0, kxy = 0;
  for (int i = a; i < b; ++i) {
    const bool live = F.alive[i] != 0, fx = live && ext2[2 * i] > Lx, fy = live && ext2[2 * i + 1] > Ly;
    kx += fx; ky += fy; kxy += (fx && ext2[2 * i + 1] > Ly);
    CornerEnt e = {F.Xi[i], F.Yi[i], i, 0};
    E[i] = e;
  }
  cx[t] = kx; cy[t] = ky; cxy[t] = kxy;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int ux = t >= d ? cx[t - d] : 0, uy = t >= d ? cy[t - d] : 0, uxy = t >= d ? cxy[t - d] : 0;
    __syncthreads();
    cx[t] += ux; cy[t] += uy; cxy[t] += uxy;
    __syncthreads();
  }
  const int nX = cx[1023], nYf = cy[1023], nXY = cxy[1023];
  int ex = cx[t] - kx, ey = cy[t] - ky, exy = cxy[t] - kxy;
  for (int i = a; i < b; ++i) {
    const bool live = F.alive[i] != 0, fx = live && ext2[2 * i] > Lx, fy = live && ext2[2 * i + 1] > Ly;
    const double x = F.Xi[i], y = F.Yi[i];
    const double xg = x - 2 * Lx * sgn(x), yg = y - 2 * Ly * sgn(y);
    if (fx) { CornerEnt g = {xg, y, i, 0}; E[n + ex++] = g; }
    if (fy) { CornerEnt g = {x, yg, i, 0}; E[n + nX + ey++] = g; }
    if (fx && ext2[2 * i + 1] > Ly) { CornerEnt g = {xg, yg, i, 0}; E[n + nX + nYf + exy++] = g; }
  }
  if (t == 0) hdr[0] = n + nX + nYf + nXY;
}

struct DetAcos { __device__ double operator()(double a) const { return szm::acos_real(a); } };

__global__ void __launch_bounds__(64) k_corner_flags(Floes F, Work W, const int *ids, int nsub, int Nb, const CornerEnt *E, const int *n0p, Box5 box,
                                                    const int *out_off, int *count, double *angles, unsigned char *da) {
  const int q = blockIdx.x;
  if (q >= nsub) return;
  const int N0 = *n0p;
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
  // (the caller's slot may be one short of the stored length -- a ring stored closed; an UNCLOSED ring in such a slot would
  // read the next floe's first flag as its last corner: not cut here)
  if (n > voff[s + 1] - voff[s]) { npieces[s] = 0; return; }
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
  // ONE packed transfer each way through the context's pinned staging buffers (round 3 made eleven copies from and to
  // pageable memory, each staged by the runtime: 1.2 ms a call for 0.14 ms of kernel)
  Packer in, out;
  const size_t i_ids = in.add((size_t)nsel * 4), i_voff = in.add(((size_t)nsel + 1) * 4), i_g = in.add(nvt + 8), i_sp = in.add(((size_t)nsel + 1) * 4),
               i_sr = in.add(((size_t)nsel + 1) * 4), i_st = in.add(((size_t)nsel + 1) * 4);
  const size_t o_np = out.add((size_t)nsel * 4), o_ro = out.add(((size_t)sp[nsel] + nsel + 1) * 4), o_rec = out.add((size_t)sp[nsel] * szk::PIECE_WORDS * 8 + 8),
               o_rx = out.add((size_t)sr[nsel] * 8 + 8), o_ry = out.add((size_t)sr[nsel] * 8 + 8);
  DevBuf &d_gt = S->post[3]; DevBuf &d_tx = S->post[7]; DevBuf &d_ty = S->post[8];
  int rc;
  if ((rc = sz_stage_reserve(ctx, ctx->h_in, in.at)) || (rc = sz_stage_reserve(ctx, ctx->h_out, out.at)) || (rc = sz_reserve(ctx, ctx->d_in, in.at)) ||
      (rc = sz_reserve(ctx, ctx->d_out, out.at)) || (rc = sz_reserve(ctx, d_gt, nvt + 8)) || (rc = sz_reserve(ctx, d_tx, (size_t)stmp[nsel] * 8 + 8)) ||
      (rc = sz_reserve(ctx, d_ty, (size_t)stmp[nsel] * 8 + 8))) return rc;
  char *hi = ctx->h_in.p, *di = (char *)ctx->d_in.p, *dout = (char *)ctx->d_out.p;
  memcpy(hi + i_ids, ids, (size_t)nsel * 4); memcpy(hi + i_voff, offsets, ((size_t)nsel + 1) * 4); memcpy(hi + i_g, grind, nvt);
  memcpy(hi + i_sp, sp.data(), ((size_t)nsel + 1) * 4); memcpy(hi + i_sr, sr.data(), ((size_t)nsel + 1) * 4); memcpy(hi + i_st, stmp.data(), ((size_t)nsel + 1) * 4);
  hipError_t he = hipMemcpyAsync(di, hi, in.at, hipMemcpyHostToDevice, st);
  if (he == hipSuccess) he = hipEventRecord(ctx->ev0, st);
  if (he == hipSuccess)
    hipLaunchKernelGGL(k_corner_cuts, grid_for(nsel, 64), dim3(64), 0, st, S->F, (const int *)(di + i_ids), nsel, (const int *)(di + i_voff), (const unsigned char *)(di + i_g),
                       (const int *)(di + i_sp), (const int *)(di + i_sr), (const int *)(di + i_st), (unsigned char *)d_gt.p, (double *)d_tx.p, (double *)d_ty.p, (int *)(dout + o_np),
                       (double *)(dout + o_rec), (int *)(dout + o_ro), (double *)(dout + o_rx), (double *)(dout + o_ry));
  if (he == hipSuccess) he = hipEventRecord(ctx->ev1, st);
  if (he == hipSuccess) he = hipMemcpyAsync(ctx->h_out.p, dout, out.at, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipStreamSynchronize(st);
  float ms = 0;
  if (he == hipSuccess) he = hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  if (he != hipSuccess) return sz_fail(ctx, SZ_ERR_HIP, "szCornerCuts: %s", hipGetErrorString(he));
  if (kernel_ms) *kernel_ms = ms;
  const char *ho = ctx->h_out.p;
  const int *h_np = (const int *)(ho + o_np), *h_ro = (const int *)(ho + o_ro);
  const double *h_rec = (const double *)(ho + o_rec), *h_rx = (const double *)(ho + o_rx), *h_ry = (const double *)(ho + o_ry);
  // pieces of the floes that were cut, back to back
  int64_t np = 0, nr = 0;
  for (int s = 0; s < nsel; ++s) {
    npieces[s] = h_np[s];
    const int *ro = h_ro + sp[s] + s;
    for (int q = 0; q < h_np[s]; ++q) {
      const int m = ro[q + 1] - ro[q];
      if (np + 1 > cap_pieces || nr + m > cap_ring) return sz_fail(ctx, SZ_ERR_CAPACITY, "szCornerCuts: output capacity (%lld pieces, %lld vertices) is too small", (long long)cap_pieces, (long long)cap_ring);
      memcpy(pieces + np * szk::PIECE_WORDS, h_rec + ((size_t)sp[s] + q) * szk::PIECE_WORDS, szk::PIECE_WORDS * 8);
      memcpy(rx + nr, h_rx + sr[s] + ro[q], (size_t)m * 8); memcpy(ry + nr, h_ry + sr[s] + ro[q], (size_t)m * 8);
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
  double Lx = c2b[0], Ly = c2b[1];
  for (int k = 1; k < 5; ++k) { Lx = szc::maxd(Lx, c2b[2 * k]); Ly = szc::maxd(Ly, c2b[2 * k + 1]); }
  Box5 box;
  for (int k = 0; k < 5; ++k) { box.x[k] = c2b[2 * k]; box.y[k] = c2b[2 * k + 1]; }
  // Everything on the device, one packed transfer each way (round 3 read the extents back, built the list of images on
  // the host and made ten copies from / to pageable memory: 1.6 ms a call for 0.08 ms of kernel)
  DevBuf &d_ext = S->post[14]; DevBuf &d_ent = S->post[15];
  Packer in, out;
  const size_t i_ids = in.add((size_t)nsub * 4 + 4), i_off = in.add(((size_t)nsub + 1) * 4);
  const size_t o_hdr = out.add(16), o_cnt = out.add((size_t)nsub * 4 + 4), o_ang = out.add((size_t)need * 8 + 8), o_da = out.add((size_t)need + 8);
  if ((rc = sz_reserve(ctx, d_ext, (size_t)n * 16 + 16)) || (rc = sz_reserve(ctx, d_ent, (size_t)4 * n * sizeof(CornerEnt) + 64)) ||
      (rc = sz_stage_reserve(ctx, ctx->h_in, in.at)) || (rc = sz_stage_reserve(ctx, ctx->h_out, out.at)) || (rc = sz_reserve(ctx, ctx->d_in, in.at)) ||
      (rc = sz_reserve(ctx, ctx->d_out, out.at))) return rc;
  char *hi = ctx->h_in.p, *di = (char *)ctx->d_in.p, *dout = (char *)ctx->d_out.p;
  if (nsub) memcpy(hi + i_ids, ids, (size_t)nsub * 4);
  memcpy(hi + i_off, offsets, ((size_t)nsub + 1) * 4);
  hipError_t he = hipMemcpyAsync(di, hi, in.at, hipMemcpyHostToDevice, st);
  if (he == hipSuccess) he = hipMemsetAsync(dout, 0, out.at, st);      // (a slot is the stored outline's length: the closing copy's entry stays 0)
  if (he == hipSuccess) he = hipEventRecord(ctx->ev0, st);
  if (he == hipSuccess) {
    hipLaunchKernelGGL(k_corner_extent, grid_for(n, 256), dim3(256), 0, st, S->F, (double *)d_ext.p);
    hipLaunchKernelGGL(k_corner_entities, dim3(1), dim3(1024), 0, st, S->F, (const double *)d_ext.p, Lx, Ly, (CornerEnt *)d_ent.p, (int *)(dout + o_hdr));
    if (nsub)
      hipLaunchKernelGGL(k_corner_flags, dim3((unsigned)nsub), dim3(64), 0, st, S->F, S->W, (const int *)(di + i_ids), nsub, Nb, (const CornerEnt *)d_ent.p, (const int *)(dout + o_hdr), box,
                         (const int *)(di + i_off), (int *)(dout + o_cnt), (double *)(dout + o_ang), (unsigned char *)(dout + o_da));
    he = hipGetLastError();
  }
  if (he == hipSuccess) he = hipEventRecord(ctx->ev1, st);
  if (he == hipSuccess) he = hipMemcpyAsync(ctx->h_out.p, dout, out.at, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipStreamSynchronize(st);
  float ms = 0;
  if (he == hipSuccess) he = hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  if (he != hipSuccess) return sz_fail(ctx, SZ_ERR_HIP, "szCornerCandidates: %s", hipGetErrorString(he));
  const char *ho = ctx->h_out.p;
  if (n_ext) *n_ext = *(const int *)(ho + o_hdr);
  if (nsub) memcpy(count, ho + o_cnt, (size_t)nsub * 4);
  if (need) { memcpy(angles, ho + o_ang, (size_t)need * 8); memcpy(da, ho + o_da, (size_t)need); }
  if (kernel_ms) *kernel_ms = ms;
  return SZ_OK;
}
