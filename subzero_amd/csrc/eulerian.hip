// eulerian.hip -- szEulerianData: calc_eulerian_data.m:1-192 on the device-resident floes.
//   k_eul_extent  per floe: max |x|, max |y| of its translated ring (the ghost rule of :41-67 reads them)
//   k_eul_cells   one workgroup per grid cell: the 256 threads share the (extended) floe list, test the bounding-circle
//                 predicate of :122-128 and clip the hits to the cell (eulerian.hpp, no buffers); thread 0 then forms the
//                 cell's sums over the hits in ascending floe order, operation for operation as :152-178 read
// The list of live floes and their periodic images (a few hundred ids and positions) is put together on the host from
// the downloaded per-floe extents -- including the reference's own quirk in the Y images (:58-67).
#include <algorithm>
#include "step_state.hpp"
#include "eulerian.hpp"

namespace {

struct EulEnt { double x, y; int src, pad; };
enum { EUL_FIELDS = 18, EUL_CAP = 2048, EUL_PAR = 192 };   // fields; potential floes of a cell held in LDS; ... summed by a thread per output

// (the extents, positions and live flags of all floes in ONE block -- one copy to the host instead of four -- and the
// result array cleared on the way: `zero` doubles, the overflow flag behind them included)
__global__ void k_eul_extent(Floes F, double *ext2, double *xs, double *ys, int *live, double *zero, long long nzero) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long long)gridDim.x * blockDim.x;
  for (long long k = t; k < nzero; k += nt) zero[k] = 0.0;
  const int i = (int)t;
  if (i >= F.n) return;
  const int v0 = F.vert_off[i], nv = F.vert_off[i + 1] - v0;
  const double X = F.Xi[i], Y = F.Yi[i];
  double mx = 0, my = 0;
  for (int k = 0; k < nv; ++k) { mx = szc::maxd(mx, szs::absd(F.cax[v0 + k] + X)); my = szc::maxd(my, szs::absd(F.cay[v0 + k] + Y)); }
  ext2[2 * i] = mx; ext2[2 * i + 1] = my;
  xs[i] = X; ys[i] = Y; live[i] = F.alive[i];
}

__device__ inline double nz(double v) { return v != v ? 0.0 : v; }

// cells whose rectangle loses a piece to the topography (Nb > 0) come with their potential floes and the overlaps
// already clipped (ov_*: szEulerianData ran them through the clip kernel); every other cell is done here
struct EulOverride { const int *cell, *off, *ent; const double *area, *abox; };

__global__ void __launch_bounds__(256) k_eul_cells(Floes F, const EulEnt *E, int M, int Nx, int Ny, const double *xg, const double *yg,
                                                  double r_max, double *out, int *overflow, EulOverride ov, int *spill_e, double *spill_a, int spill_len) {
  // the cell's potential floes and their overlaps: EUL_CAP of them in LDS -- or, in the second launch that a cell with
  // more provokes (a coarse grid over a large field: the reference has no such limit), in the cell's slice of HBM
  __shared__ int lds_e[EUL_CAP];
  __shared__ double lds_a[EUL_CAP];
  __shared__ int npot;
  int *hit_e = spill_e ? spill_e + (size_t)blockIdx.x * spill_len : lds_e;
  double *hit_a = spill_a ? spill_a + (size_t)blockIdx.x * spill_len : lds_a;
  const int cap = spill_e ? spill_len : (int)EUL_CAP;
  const int ii = blockIdx.x / Ny, jj = blockIdx.x % Ny;
  if (threadIdx.x == 0) npot = 0;
  __syncthreads();
  const double xl = xg[ii], xh = xg[ii + 1], yh = yg[jj], yl = yg[jj + 1];
  const double xc = 0.5 * (xg[ii] + xg[ii + 1]), yc = 0.5 * (yg[jj] + yg[jj + 1]);
  const int ovc = ov.cell ? ov.cell[blockIdx.x] : -1;
  if (ovc >= 0) {
    const int o0 = ov.off[ovc], cnt = ov.off[ovc + 1] - o0;
    if (cnt > cap) { if (threadIdx.x == 0) atomicAdd(overflow, 1); return; }
    for (int q = threadIdx.x; q < cnt; q += blockDim.x) { hit_e[q] = ov.ent[o0 + q]; hit_a[q] = ov.area[o0 + q]; }
    if (threadIdx.x == 0) npot = cnt;
  }
  for (int e = threadIdx.x; e < M && ovc < 0; e += blockDim.x) {
    const EulEnt en = E[e];
    const double ddx = xc - en.x, ddy = yc - en.y;
    const double pint = ::sqrt(ddx * ddx + ddy * ddy) - (F.rmax[en.src] + r_max);   // :123
    if (!(pint < 0)) continue;
    int k = atomicAdd(&npot, 1);
    if (k >= cap) continue;
    hit_e[k] = e;
  }
  __syncthreads();
  if (npot > cap) { if (threadIdx.x == 0) atomicAdd(overflow, 1); return; }
  const int np = npot;
  // Mtot of :136 (any positive mass among the potential floes lets the cell through); then the overlaps
  for (int q = threadIdx.x; q < np && ovc < 0; q += blockDim.x) {
    const EulEnt en = E[hit_e[q]];
    const int v0 = F.vert_off[en.src];
    int nv = F.vert_off[en.src + 1] - v0;
    if (nv > 1 && F.cax[v0] == F.cax[v0 + nv - 1] && F.cay[v0] == F.cay[v0 + nv - 1]) --nv;   // polyshape drops the closing copy
    hit_a[q] = sze::ring_box_area(F.cax + v0, F.cay + v0, nv, en.x, en.y, xl, xh, yl, yh);
  }
  __syncthreads();
  if (spill_e && np > (int)EUL_CAP) {
    // long lists (HBM): rank every entry by all threads (entries are distinct), scatter into the second half of the slice
    int *se = hit_e + spill_len / 2; double *sa = hit_a + spill_len / 2;
    for (int q = threadIdx.x; q < np; q += blockDim.x) {
      const int e = hit_e[q];
      int rank = 0;
      for (int m = 0; m < np; ++m) rank += (hit_e[m] < e) ? 1 : 0;
      se[rank] = e; sa[rank] = hit_a[q];
    }
    __syncthreads();
    for (int q = threadIdx.x; q < np; q += blockDim.x) { hit_e[q] = se[q]; hit_a[q] = sa[q]; }
    __syncthreads();
  }
  // ---- short lists (every cell of an ordinary grid): the hits' fields are gathered by a thread each and every output's
  // ordered sum is formed by a thread of its own, term for term as the one-thread form below forms it (round 4: that
  // form walked 18 sums of ~20 dependent gathers each by one thread, 0.2 of the kernel's 0.25 ms)
  if (np <= (int)EUL_PAR && !(spill_e && np > (int)EUL_CAP)) {
    __shared__ double s_val[16][EUL_PAR];
    __shared__ double s_sum[20];
    __shared__ int s_idx[EUL_PAR];
    __shared__ int s_nf;
    const int t = threadIdx.x;
    if (t == 0) {   // ascending floe order (the potential list was filled by atomics)
      for (int a = 1; a < np; ++a) {
        int e = hit_e[a]; double ar = hit_a[a]; int b = a;
        while (b > 0 && hit_e[b - 1] > e) { hit_e[b] = hit_e[b - 1]; hit_a[b] = hit_a[b - 1]; --b; }
        hit_e[b] = e; hit_a[b] = ar;
      }
    }
    __syncthreads();
    if (t < np) {
      const int src = E[hit_e[t]].src;
      s_val[0][t] = F.mass[src]; s_val[1][t] = F.area[src]; s_val[2][t] = F.OverlapArea[src]; s_val[3][t] = F.h[src];
      s_val[4][t] = F.Ui[src]; s_val[5][t] = F.Vi[src]; s_val[6][t] = F.dUi_p[src]; s_val[7][t] = F.dVi_p[src];
      for (int c = 0; c < 4; ++c) { s_val[8 + c][t] = F.Stress[4 * (size_t)src + c]; s_val[12 + c][t] = F.strain[4 * (size_t)src + c]; }
    }
    __syncthreads();
    if (t == 0) {
      double Mpre = 0;
      for (int q = 0; q < np; ++q) Mpre += s_val[0][q];
      int nf = 0;
      if (Mpre > 0) for (int q = 0; q < np; ++q) if (hit_a[q] != 0) s_idx[nf++] = q;      // drop the floes without a piece in the cell (:150-151)
      s_nf = (Mpre > 0) ? nf : -1;
    }
    __syncthreads();
    const int nf = s_nf;
    if (nf < 0) return;
    // sum t: 0 Mtot, 1 Atot, 2 the OverlapArea sum, 3.. the mass-weighted sums of h, Ui, Vi, dUi_p, dVi_p, Stress 0..3, strain 0..3
    if (t < 16) {
      double acc = 0;
      for (int k = 0; k < nf; ++k) {
        const int q = s_idx[k];
        const double a = hit_a[q], m = nz(s_val[0][q]), ar = nz(s_val[1][q]);
        if (t == 0) acc += m * a / ar;
        else if (t == 1) acc += a;
        else if (t == 2) acc += s_val[2][q];
        else acc += nz(s_val[t][q]) * m * a / ar;
      }
      s_sum[t] = acc;
    }
    __syncthreads();
    if (t != 0) return;
    const size_t cells = (size_t)Nx * Ny, at = (size_t)jj + (size_t)Ny * ii;
    double bxs[4] = {xl, xl, xh, xh}, bys[4] = {yh, yl, yl, yh};
    double abox;
    { double sa = 0; for (int i = 0; i < 4; ++i) { int k = (i + 1) & 3; double ax = bxs[i] - bxs[0], ay = bys[i] - bys[0], bx = bxs[k] - bxs[0], by = bys[k] - bys[0]; sa += ax * by - bx * ay; } abox = szs::absd(0.5 * sa); }
    if (ovc >= 0) abox = ov.abox[ovc];
    const double Mtot = s_sum[0], Atot = s_sum[1];
    out[13 * cells + at] = Atot / abox;
    if (!(Mtot > 0)) return;
    out[14 * cells + at] = s_sum[2] / (double)nf;
    out[15 * cells + at] = Mtot;
    out[16 * cells + at] = Atot;
    out[17 * cells + at] = s_sum[3] / Mtot;
    out[0 * cells + at] = s_sum[4] / Mtot; out[1 * cells + at] = s_sum[5] / Mtot;
    out[2 * cells + at] = s_sum[6] / Mtot; out[3 * cells + at] = s_sum[7] / Mtot;
    const double sxx = s_sum[8] / Mtot, syx = s_sum[10] / Mtot, sxy = s_sum[9] / Mtot, syy = s_sum[11] / Mtot;
    out[5 * cells + at] = sxx; out[6 * cells + at] = syx; out[7 * cells + at] = sxy; out[8 * cells + at] = syy;
    out[9 * cells + at] = s_sum[12] / Mtot; out[10 * cells + at] = s_sum[14] / Mtot;
    out[11 * cells + at] = s_sum[13] / Mtot; out[12 * cells + at] = s_sum[15] / Mtot;
    const double tr = 0.5 * (sxx + syy), df = 0.5 * (sxx - syy);
    double ev = tr + ::sqrt(df * df + syx * sxy);      // max(eig([sxx syx; sxy syy])), :176
    if (szs::absd(ev) > 1e8) ev = 0;
    out[4 * cells + at] = ev;
    return;
  }
  if (threadIdx.x != 0) return;
  // ascending floe order (the potential list was filled by atomics)
  for (int a = 1; a < np && !(spill_e && np > (int)EUL_CAP); ++a) {
    int e = hit_e[a]; double ar = hit_a[a]; int b = a;
    while (b > 0 && hit_e[b - 1] > e) { hit_e[b] = hit_e[b - 1]; hit_a[b] = hit_a[b - 1]; --b; }
    hit_e[b] = e; hit_a[b] = ar;
  }
  double Mpre = 0;
  for (int q = 0; q < np; ++q) Mpre += F.mass[E[hit_e[q]].src];
  if (!(Mpre > 0)) return;
  const size_t cells = (size_t)Nx * Ny, at = (size_t)jj + (size_t)Ny * ii;
  double bxs[4] = {xl, xl, xh, xh}, bys[4] = {yh, yl, yl, yh};
  double abox;
  { double s = 0; for (int i = 0; i < 4; ++i) { int k = (i + 1) & 3; double ax = bxs[i] - bxs[0], ay = bys[i] - bys[0], bx = bxs[k] - bxs[0], by = bys[k] - bys[0]; s += ax * by - bx * ay; } abox = szs::absd(0.5 * s); }
  if (ovc >= 0) abox = ov.abox[ovc];
  // drop the floes without a piece in the cell (:150-151)
  int nf = 0;
  for (int q = 0; q < np; ++q) if (hit_a[q] != 0) { hit_e[nf] = hit_e[q]; hit_a[nf] = hit_a[q]; ++nf; }
  double Mtot = 0, Atot = 0;
  for (int q = 0; q < nf; ++q) { const int s = E[hit_e[q]].src; Mtot += nz(F.mass[s]) * hit_a[q] / nz(F.area[s]); Atot += hit_a[q]; }
  out[13 * cells + at] = Atot / abox;
  if (!(Mtot > 0)) return;
  double sOver = 0;
  for (int q = 0; q < nf; ++q) sOver += F.OverlapArea[E[hit_e[q]].src];
  out[14 * cells + at] = sOver / (double)nf;
  out[15 * cells + at] = Mtot;
  out[16 * cells + at] = Atot;
  auto wavg = [&](const double *field, int stride, int comp) {
    double s = 0;
    for (int q = 0; q < nf; ++q) { const int src = E[hit_e[q]].src; s += nz(field[(size_t)stride * src + comp]) * nz(F.mass[src]) * hit_a[q] / nz(F.area[src]); }
    return s / Mtot;
  };
  out[17 * cells + at] = wavg(F.h, 1, 0);
  out[0 * cells + at] = wavg(F.Ui, 1, 0); out[1 * cells + at] = wavg(F.Vi, 1, 0);
  out[2 * cells + at] = wavg(F.dUi_p, 1, 0); out[3 * cells + at] = wavg(F.dVi_p, 1, 0);
  const double sxx = wavg(F.Stress, 4, 0), syx = wavg(F.Stress, 4, 2), sxy = wavg(F.Stress, 4, 1), syy = wavg(F.Stress, 4, 3);
  out[5 * cells + at] = sxx; out[6 * cells + at] = syx; out[7 * cells + at] = sxy; out[8 * cells + at] = syy;
  out[9 * cells + at] = wavg(F.strain, 4, 0); out[10 * cells + at] = wavg(F.strain, 4, 2);
  out[11 * cells + at] = wavg(F.strain, 4, 1); out[12 * cells + at] = wavg(F.strain, 4, 3);
  const double tr = 0.5 * (sxx + syy), df = 0.5 * (sxx - syy);
  double ev = tr + ::sqrt(df * df + syx * sxy);      // max(eig([sxx syx; sxy syy])), :176
  if (szs::absd(ev) > 1e8) ev = 0;
  out[4 * cells + at] = ev;
}

} // namespace

extern "C" int szEulerianData(szContext *ctx, int Nx, int Ny, int Nb, const double *c2b, int periodic, double *out, double *kernel_ms) {
  if (!ctx || !c2b || !out || Nx < 1 || Ny < 1) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S) return sz_fail(ctx, SZ_ERR_STATE, "szEulerianData: no floes uploaded");
  if (Nb < 0) return SZ_ERR_BAD_ARG;
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int n = S->n;
  int rc;
  DevBuf &d_ext = S->post[24]; DevBuf &d_ent = S->post[25]; DevBuf &d_grid = S->post[26]; DevBuf &d_out = S->post[27];
  auto cleanup = [&]() { (void)d_ext; (void)d_ent; (void)d_grid; (void)d_out; };
  // (read back into the context's pinned staging buffer: a copy to pageable memory is staged by the runtime, tens of
  // microseconds each; the device block has the layout of the staging block)
  Packer back;
  const size_t b_ext = back.add((size_t)2 * n * 8), b_x = back.add((size_t)n * 8), b_y = back.add((size_t)n * 8), b_al = back.add((size_t)n * 4);
  const size_t cells = (size_t)Nx * Ny;
  const size_t out_bytes = EUL_FIELDS * cells * 8;        // (+ 16: the overflow flag travels behind the result)
  if ((rc = sz_reserve(ctx, d_ext, back.at + 16)) || (rc = sz_reserve(ctx, d_out, out_bytes + 16))) return rc;
  if ((rc = sz_stage_reserve(ctx, ctx->h_out, back.at))) return rc;
  int *const d_flag = (int *)((char *)d_out.p + out_bytes);
  const i64 nzero = (i64)(out_bytes / 8 + 2), threads = std::max<i64>(n, std::min<i64>(nzero, 1 << 16));     // (a thread per floe; the clearing strides)
  hipLaunchKernelGGL(k_eul_extent, grid_for(threads, 256), dim3(256), 0, st, S->F, (double *)((char *)d_ext.p + b_ext),
                     (double *)((char *)d_ext.p + b_x), (double *)((char *)d_ext.p + b_y), (int *)((char *)d_ext.p + b_al), (double *)d_out.p, (long long)nzero);
  const double *ext = (const double *)(ctx->h_out.p + b_ext), *x = (const double *)(ctx->h_out.p + b_x), *y = (const double *)(ctx->h_out.p + b_y);
  const int *alive = (const int *)(ctx->h_out.p + b_al);
  hipError_t he = hipGetLastError();
  if (he == hipSuccess) he = hipMemcpyAsync(ctx->h_out.p, d_ext.p, back.at, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipStreamSynchronize(st);
  if (he != hipSuccess) { cleanup(); return sz_fail(ctx, SZ_ERR_HIP, "szEulerianData: %s", hipGetErrorString(he)); }
  double Lx = c2b[0], Ly = c2b[1], xmin = c2b[0], ymin = c2b[1];
  for (int k = 1; k < 5; ++k) { Lx = szc::maxd(Lx, c2b[2 * k]); Ly = szc::maxd(Ly, c2b[2 * k + 1]); xmin = szc::mind(xmin, c2b[2 * k]); ymin = szc::mind(ymin, c2b[2 * k + 1]); }
  // the live floes and their periodic images (:7-8, :29-69), with their extents translated along
  std::vector<EulEnt> E;
  std::vector<double> ex, ey;     // max |x|, max |y| of each entry's ring at its own position
  std::vector<int> topo;          // :10-24 the first Nb live floes: their union is cut out of the cells, they are not averaged
  for (int i = 0; i < n; ++i) {
    if (!alive[i]) continue;
    if ((int)topo.size() < Nb) { topo.push_back(i); continue; }
    EulEnt e = {x[i], y[i], i, 0}; E.push_back(e); ex.push_back(ext[2 * i]); ey.push_back(ext[2 * i + 1]);
  }
  if (periodic && !E.empty()) {
    const size_t n1 = E.size();
    for (size_t i = 0; i < n1; ++i)
      if (ex[i] > Lx) { EulEnt g = E[i]; g.x = E[i].x - 2 * Lx * sgn(E[i].x); E.push_back(g); ex.push_back(0); ey.push_back(ey[i]); }
    const bool all = ey[n1 - 1] > Ly;      // (:58-67 reads the polyshape left over from the first loop: the last floe's)
    const size_t n2 = E.size();
    if (all) for (size_t i = 0; i < n2; ++i) { EulEnt g = E[i]; g.y = E[i].y - 2 * Ly * sgn(E[i].y); E.push_back(g); }
  }
  const int M = (int)E.size();
  std::vector<double> grid((size_t)Nx + Ny + 2);
  double *xg = grid.data(), *yg = grid.data() + Nx + 1;
  const double stepx = (Lx - xmin) / Nx, stepy = (Ly - ymin) / Ny;
  for (int k = 0; k <= Nx; ++k) xg[k] = xmin + k * stepx;
  for (int k = 0; k <= Ny; ++k) yg[Ny - k] = ymin + k * stepy;
  xg[Nx] = Lx; yg[0] = Ly;
  const double dx = absd(xg[1] - xg[0]), dy = absd(yg[1] - yg[0]);
  const double r_max = ::sqrt((dx / 2) * (dx / 2) + (dy / 2) * (dy / 2));
  // ---- cells that meet the topography (Nb > 0): box = subtract(box, boundaries), overlap = intersect(box, poly) (:141-154)
  // through the clip kernel (szClipPathsBatch: the polyshape operations restated by the fixed-point clip, as polyclip.m
  // uses it).  Only a cell whose rectangle meets the bounding box of a topography floe can lose anything.
  DevBuf &d_ovc = S->post[29]; DevBuf &d_ovo = S->post[30]; DevBuf &d_ove = S->post[31]; DevBuf &d_ova = S->post[32]; DevBuf &d_ovb = S->post[33];
  auto cleanup2 = [&]() { (void)d_ovc; (void)d_ovo; (void)d_ove; (void)d_ova; (void)d_ovb; };
  EulOverride ov = {nullptr, nullptr, nullptr, nullptr, nullptr};
  if (!topo.empty() && M > 0) {
    const size_t nvt = (size_t)S->nverts;
    std::vector<double> cax(nvt + 1), cay(nvt + 1), rmax((size_t)n);
    he = hipMemcpyAsync(cax.data(), S->F.cax, nvt * 8, hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipMemcpyAsync(cay.data(), S->F.cay, nvt * 8, hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipMemcpyAsync(rmax.data(), S->F.rmax, (size_t)n * 8, hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    if (he != hipSuccess) { cleanup(); return sz_fail(ctx, SZ_ERR_HIP, "szEulerianData: %s", hipGetErrorString(he)); }
    const int *vo = S->h_vert_off.data();
    auto signed_area = [](const double *px, const double *py, int m) {       // shoelace relative to the first vertex
      if (m < 3) return 0.0;
      double a = 0;
      for (int i = 0; i < m; ++i) { const int k = (i + 1 == m) ? 0 : i + 1; const double ax = px[i] - px[0], ay = py[i] - py[0], bx = px[k] - px[0], by = py[k] - py[0]; a += ax * by - bx * ay; }
      return 0.5 * a;
    };
    // topography rings in metres (positive sense), their boxes, and in fixed point
    struct TRing { std::vector<double> x, y; double x0, x1, y0, y1; std::vector<int64_t> fx, fy; };
    std::vector<TRing> T(topo.size());
    for (size_t t = 0; t < topo.size(); ++t) {
      const int i = topo[t], v0 = vo[i], nv = vo[i + 1] - v0;
      TRing &r = T[t];
      for (int k = 0; k < nv; ++k) { r.x.push_back(cax[v0 + k] + x[i]); r.y.push_back(cay[v0 + k] + y[i]); }
      if (signed_area(r.x.data(), r.y.data(), nv) < 0) { std::reverse(r.x.begin(), r.x.end()); std::reverse(r.y.begin(), r.y.end()); }
      r.x0 = r.x1 = r.x[0]; r.y0 = r.y1 = r.y[0];
      for (int k = 1; k < nv; ++k) { r.x0 = szc::mind(r.x0, r.x[k]); r.x1 = szc::maxd(r.x1, r.x[k]); r.y0 = szc::mind(r.y0, r.y[k]); r.y1 = szc::maxd(r.y1, r.y[k]); }
      for (int k = 0; k < nv; ++k) { r.fx.push_back(szc::to_fixed(r.x[k])); r.fy.push_back(szc::to_fixed(r.y[k])); }
    }
    std::vector<int> ovc(cells, -1), cut_cells;           // cell -> index among the cut cells
    std::vector<int64_t> ax, ay, bx, by;
    std::vector<int32_t> aro(1, 0), apo(1, 0), bro(1, 0), bpo(1, 0);
    for (int ii = 0; ii < Nx; ++ii)
      for (int jj = 0; jj < Ny; ++jj) {
        const double xl = xg[ii], xh = xg[ii + 1], yh = yg[jj], yl = yg[jj + 1];
        bool any = false;
        for (size_t t = 0; t < T.size(); ++t) {
          if (T[t].x1 < xl || T[t].x0 > xh || T[t].y1 < yl || T[t].y0 > yh) continue;
          any = true;
          bx.insert(bx.end(), T[t].fx.begin(), T[t].fx.end()); by.insert(by.end(), T[t].fy.begin(), T[t].fy.end());
          bro.push_back((int32_t)bx.size());
        }
        if (!any) continue;
        const double bxs[4] = {xl, xl, xh, xh}, bys[4] = {yh, yl, yl, yh};
        for (int k = 0; k < 4; ++k) { ax.push_back(szc::to_fixed(bxs[k])); ay.push_back(szc::to_fixed(bys[k])); }
        aro.push_back((int32_t)ax.size()); apo.push_back((int32_t)aro.size() - 1); bpo.push_back((int32_t)bro.size() - 1);
        ovc[(size_t)ii * Ny + jj] = (int)cut_cells.size();
        cut_cells.push_back(ii * Ny + jj);
      }
    const int nc = (int)cut_cells.size();
    if (nc > 0) {
      // batch 1: rectangle minus the union of the topography floes near it ('dif', clip fill non-zero)
      auto run_batch = [&](int np, std::vector<int64_t> &sx, std::vector<int64_t> &sy, std::vector<int32_t> &sro, std::vector<int32_t> &spo,
                           std::vector<int64_t> &cx, std::vector<int64_t> &cy, std::vector<int32_t> &cro, std::vector<int32_t> &cpo, int method, int fc,
                           std::vector<int64_t> &ox, std::vector<int64_t> &oy, std::vector<int32_t> &oro, std::vector<int32_t> &opo) -> int {
        sx.push_back(0); sy.push_back(0); cx.push_back(0); cy.push_back(0);
        int64_t cap = 4 * (int64_t)(sx.size() + cx.size()) + 64 * (int64_t)np + 64;
        std::vector<int32_t> stt((size_t)np + 1);
        for (;;) {
          ox.assign((size_t)cap, 0); oy.assign((size_t)cap, 0); oro.assign((size_t)cap + 1, 0); opo.assign((size_t)np + 1, 0);
          const int r = szClipPathsBatch(ctx, np, sx.data(), sy.data(), sro.data(), spo.data(), cx.data(), cy.data(), cro.data(), cpo.data(), method, 0, fc,
                                         ox.data(), oy.data(), cap, oro.data(), cap, opo.data(), stt.data());
          if (r == SZ_ERR_CAPACITY) { cap *= 4; continue; }
          if (r != SZ_OK) return r;
          for (int p = 0; p < np; ++p) if (stt[p] != SZ_CLIP_OK) return sz_fail(ctx, SZ_ERR_HIP, "szEulerianData: a cell hit the reference's \"Clipper Error.\" condition");
          return SZ_OK;
        }
      };
      auto rings_area = [&](const std::vector<int64_t> &ox, const std::vector<int64_t> &oy, const std::vector<int32_t> &oro, int r0, int r1) {
        double sum = 0;
        std::vector<double> px, py;
        for (int r = r0; r < r1; ++r) {
          const int m = oro[r + 1] - oro[r];
          px.resize((size_t)m); py.resize((size_t)m);
          for (int k = 0; k < m; ++k) { px[k] = (double)ox[oro[r] + k] / 4294967296.0; py[k] = (double)oy[oro[r] + k] / 4294967296.0; }
          sum += signed_area(px.data(), py.data(), m);
        }
        return absd(sum);
      };
      std::vector<int64_t> c1x, c1y; std::vector<int32_t> c1ro, c1po;
      if ((rc = run_batch(nc, ax, ay, aro, apo, bx, by, bro, bpo, SZ_CLIP_DIFFERENCE, SZ_FILL_NONZERO, c1x, c1y, c1ro, c1po))) { cleanup(); return rc; }
      std::vector<double> ovb((size_t)nc);
      for (int c = 0; c < nc; ++c) ovb[c] = rings_area(c1x, c1y, c1ro, c1po[c], c1po[c + 1]);
      // the potential floes of each cut cell (:122-128) and batch 2: what is left of the rectangle against each of them
      std::vector<int32_t> ovo(1, 0), ove;
      std::vector<int> pair_cell;
      std::vector<int64_t> sx, sy, fx, fy;
      std::vector<int32_t> sro(1, 0), spo(1, 0), fro(1, 0), fpo(1, 0);
      for (int c = 0; c < nc; ++c) {
        const int ii = cut_cells[c] / Ny, jj = cut_cells[c] % Ny;
        const double xc = 0.5 * (xg[ii] + xg[ii + 1]), yc = 0.5 * (yg[jj] + yg[jj + 1]);
        for (int e = 0; e < M; ++e) {
          const double ddx = xc - E[e].x, ddy = yc - E[e].y;
          const double pint = ::sqrt(ddx * ddx + ddy * ddy) - (rmax[E[e].src] + r_max);
          if (!(pint < 0)) continue;
          ove.push_back(e);
          if (c1po[c + 1] == c1po[c]) continue;            // nothing of the rectangle is left: no overlap to compute
          for (int r = c1po[c]; r < c1po[c + 1]; ++r) {
            sx.insert(sx.end(), c1x.begin() + c1ro[r], c1x.begin() + c1ro[r + 1]); sy.insert(sy.end(), c1y.begin() + c1ro[r], c1y.begin() + c1ro[r + 1]);
            sro.push_back((int32_t)sx.size());
          }
          spo.push_back((int32_t)sro.size() - 1);
          const int src = E[e].src, v0 = vo[src];
          int nv = vo[src + 1] - v0;
          if (nv > 1 && cax[v0] == cax[v0 + nv - 1] && cay[v0] == cay[v0 + nv - 1]) --nv;
          for (int k = 0; k < nv; ++k) { fx.push_back(szc::to_fixed(cax[v0 + k] + E[e].x)); fy.push_back(szc::to_fixed(cay[v0 + k] + E[e].y)); }
          fro.push_back((int32_t)fx.size()); fpo.push_back((int32_t)fro.size() - 1);
          pair_cell.push_back((int)ove.size() - 1);
        }
        ovo.push_back((int32_t)ove.size());
      }
      std::vector<double> ova(ove.size() + 1, 0.0);
      const int np2 = (int)pair_cell.size();
      if (np2 > 0) {
        std::vector<int64_t> c2x, c2y; std::vector<int32_t> c2ro, c2po;
        if ((rc = run_batch(np2, sx, sy, sro, spo, fx, fy, fro, fpo, SZ_CLIP_INTERSECTION, SZ_FILL_EVENODD, c2x, c2y, c2ro, c2po))) { cleanup(); return rc; }
        for (int p = 0; p < np2; ++p) ova[(size_t)pair_cell[p]] = rings_area(c2x, c2y, c2ro, c2po[p], c2po[p + 1]);
      }
      ove.push_back(0);
      if ((rc = sz_reserve(ctx, d_ovc, cells * 4)) || (rc = sz_reserve(ctx, d_ovo, ovo.size() * 4)) || (rc = sz_reserve(ctx, d_ove, ove.size() * 4)) ||
          (rc = sz_reserve(ctx, d_ova, ova.size() * 8)) || (rc = sz_reserve(ctx, d_ovb, ovb.size() * 8))) { cleanup(); cleanup2(); return rc; }
      he = hipMemcpyAsync(d_ovc.p, ovc.data(), cells * 4, hipMemcpyHostToDevice, st);
      if (he == hipSuccess) he = hipMemcpyAsync(d_ovo.p, ovo.data(), ovo.size() * 4, hipMemcpyHostToDevice, st);
      if (he == hipSuccess) he = hipMemcpyAsync(d_ove.p, ove.data(), ove.size() * 4, hipMemcpyHostToDevice, st);
      if (he == hipSuccess) he = hipMemcpyAsync(d_ova.p, ova.data(), ova.size() * 8, hipMemcpyHostToDevice, st);
      if (he == hipSuccess) he = hipMemcpyAsync(d_ovb.p, ovb.data(), ovb.size() * 8, hipMemcpyHostToDevice, st);
      if (he == hipSuccess) he = hipStreamSynchronize(st);      // (the host vectors go out of scope below)
      if (he != hipSuccess) { cleanup(); cleanup2(); return sz_fail(ctx, SZ_ERR_HIP, "szEulerianData: %s", hipGetErrorString(he)); }
      ov.cell = (const int *)d_ovc.p; ov.off = (const int *)d_ovo.p; ov.ent = (const int *)d_ove.p; ov.area = (const double *)d_ova.p; ov.abox = (const double *)d_ovb.p;
    }
  }
  if ((rc = sz_reserve(ctx, d_ent, (size_t)(M + 1) * sizeof(EulEnt))) || (rc = sz_reserve(ctx, d_grid, grid.size() * 8))) { cleanup(); cleanup2(); return rc; }
  // (entities, grid lines and the result through the pinned staging buffer, as the read-back above)
  Packer up;
  const size_t u_ent = up.add((size_t)(M + 1) * sizeof(EulEnt)), u_grid = up.add(grid.size() * 8), u_out = up.add(out_bytes + 16);
  if ((rc = sz_stage_reserve(ctx, ctx->h_in, up.at))) { cleanup(); cleanup2(); return rc; }
  if (M) memcpy(ctx->h_in.p + u_ent, E.data(), (size_t)M * sizeof(EulEnt));
  memcpy(ctx->h_in.p + u_grid, grid.data(), grid.size() * 8);
  he = hipMemcpyAsync(d_ent.p, ctx->h_in.p + u_ent, (size_t)M * sizeof(EulEnt), hipMemcpyHostToDevice, st);
  if (he == hipSuccess) he = hipMemcpyAsync(d_grid.p, ctx->h_in.p + u_grid, grid.size() * 8, hipMemcpyHostToDevice, st);
  // (the result and the flag were cleared by k_eul_extent; both come back in one copy behind the kernel: one wait)
  if (he == hipSuccess) he = hipEventRecord(ctx->ev0, st);
  if (he == hipSuccess && M > 0)
    hipLaunchKernelGGL(k_eul_cells, dim3((unsigned)cells), dim3(256), 0, st, S->F, (const EulEnt *)d_ent.p, M, Nx, Ny,
                       (const double *)d_grid.p, (const double *)d_grid.p + Nx + 1, r_max, (double *)d_out.p, d_flag, ov,
                       (int *)nullptr, (double *)nullptr, 0);
  if (he == hipSuccess) he = hipEventRecord(ctx->ev1, st);
  const int *const h_flag = (const int *)(ctx->h_in.p + u_out + out_bytes);
  if (he == hipSuccess) he = hipMemcpyAsync(ctx->h_in.p + u_out, d_out.p, out_bytes + 16, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipStreamSynchronize(st);
  int over = (he == hipSuccess) ? h_flag[0] : 0;
  if (he == hipSuccess && over) {
    // a cell holds more potential floes than the LDS lists: once more with the lists in HBM (the reference
    // has no limit; a 1 x 1 grid over 1e4 floes is a legitimate domain average).  Twice M entries per cell: list + sort room.
    const size_t per = 2 * (size_t)(M + 8);
    if (cells * per * 12 > ((size_t)2 << 30)) { cleanup(); cleanup2(); return sz_fail(ctx, SZ_ERR_WORKSPACE, "szEulerianData: a grid cell has more than %d potential floes and the lists of all %zu cells would not fit 2 GB", (int)EUL_CAP, cells); }
    DevBuf &d_se = S->post[34]; DevBuf &d_sa = S->post[35];
    if ((rc = sz_reserve(ctx, d_se, cells * per * 4)) || (rc = sz_reserve(ctx, d_sa, cells * per * 8))) { cleanup(); cleanup2(); return rc; }
    he = hipMemsetAsync(d_out.p, 0, out_bytes + 16, st);
    if (he == hipSuccess) he = hipEventRecord(ctx->ev0, st);
    if (he == hipSuccess)
      hipLaunchKernelGGL(k_eul_cells, dim3((unsigned)cells), dim3(256), 0, st, S->F, (const EulEnt *)d_ent.p, M, Nx, Ny,
                         (const double *)d_grid.p, (const double *)d_grid.p + Nx + 1, r_max, (double *)d_out.p, d_flag, ov,
                         (int *)d_se.p, (double *)d_sa.p, (int)per);
    if (he == hipSuccess) he = hipEventRecord(ctx->ev1, st);
    if (he == hipSuccess) he = hipMemcpyAsync(ctx->h_in.p + u_out, d_out.p, out_bytes + 16, hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    if (he == hipSuccess) over = h_flag[0];
  }
  if (he == hipSuccess) memcpy(out, ctx->h_in.p + u_out, out_bytes);
  float ms = 0;
  if (he == hipSuccess) he = hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  cleanup(); cleanup2();
  if (he != hipSuccess) return sz_fail(ctx, SZ_ERR_HIP, "szEulerianData: %s", hipGetErrorString(he));
  if (over) return sz_fail(ctx, SZ_ERR_WORKSPACE, "szEulerianData: a grid cell has more than %d potential floes (use a finer grid)", (int)EUL_CAP);
  if (kernel_ms) *kernel_ms = ms;
  return SZ_OK;
}
