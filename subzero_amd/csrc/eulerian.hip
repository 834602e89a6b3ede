// eulerian.hip -- szEulerianData: calc_eulerian_data.m:1-192 on the device-resident floes.
//   k_eul_extent  per floe: max |x|, max |y| of its translated ring (the ghost rule of :41-67 reads them)
//   k_eul_cells   one workgroup per grid cell: the 256 threads share the (extended) floe list, test the bounding-circle
//                 predicate of :122-128 and clip the hits to the cell (eulerian.hpp, no buffers); thread 0 then forms the
//                 cell's sums over the hits in ascending floe order, operation for operation as :152-178 read
// The list of live floes and their periodic images (a few hundred ids and positions) is put together on the host from
// the downloaded per-floe extents -- including the reference's own quirk in the Y images (:58-67).
#include "step_state.hpp"
#include "eulerian.hpp"

namespace {

struct EulEnt { double x, y; int src, pad; };
enum { EUL_FIELDS = 18, EUL_CAP = 2048 };

__global__ void k_eul_extent(Floes F, double *ext2) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F.n) return;
  const int v0 = F.vert_off[i], nv = F.vert_off[i + 1] - v0;
  double mx = 0, my = 0;
  for (int k = 0; k < nv; ++k) { mx = szc::maxd(mx, szs::absd(F.cax[v0 + k] + F.Xi[i])); my = szc::maxd(my, szs::absd(F.cay[v0 + k] + F.Yi[i])); }
  ext2[2 * i] = mx; ext2[2 * i + 1] = my;
}

__device__ inline double nz(double v) { return v != v ? 0.0 : v; }

__global__ void __launch_bounds__(256) k_eul_cells(Floes F, const EulEnt *E, int M, int Nx, int Ny, const double *xg, const double *yg,
                                                  double r_max, double *out, int *overflow) {
  __shared__ int hit_e[EUL_CAP];
  __shared__ double hit_a[EUL_CAP];
  __shared__ int npot;
  const int ii = blockIdx.x / Ny, jj = blockIdx.x % Ny;
  if (threadIdx.x == 0) npot = 0;
  __syncthreads();
  const double xl = xg[ii], xh = xg[ii + 1], yh = yg[jj], yl = yg[jj + 1];
  const double xc = 0.5 * (xg[ii] + xg[ii + 1]), yc = 0.5 * (yg[jj] + yg[jj + 1]);
  for (int e = threadIdx.x; e < M; e += blockDim.x) {
    const EulEnt en = E[e];
    const double ddx = xc - en.x, ddy = yc - en.y;
    const double pint = ::sqrt(ddx * ddx + ddy * ddy) - (F.rmax[en.src] + r_max);   // :123
    if (!(pint < 0)) continue;
    int k = atomicAdd(&npot, 1);
    if (k >= EUL_CAP) continue;
    hit_e[k] = e;
  }
  __syncthreads();
  if (npot > EUL_CAP) { if (threadIdx.x == 0) atomicAdd(overflow, 1); return; }
  const int np = npot;
  // Mtot of :136 (any positive mass among the potential floes lets the cell through); then the overlaps
  for (int q = threadIdx.x; q < np; q += blockDim.x) {
    const EulEnt en = E[hit_e[q]];
    const int v0 = F.vert_off[en.src];
    int nv = F.vert_off[en.src + 1] - v0;
    if (nv > 1 && F.cax[v0] == F.cax[v0 + nv - 1] && F.cay[v0] == F.cay[v0 + nv - 1]) --nv;   // polyshape drops the closing copy
    hit_a[q] = sze::ring_box_area(F.cax + v0, F.cay + v0, nv, en.x, en.y, xl, xh, yl, yh);
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  // ascending floe order (the potential list was filled by atomics)
  for (int a = 1; a < np; ++a) {
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
  if (Nb > 0) return sz_fail(ctx, SZ_ERR_BAD_ARG, "szEulerianData: Nb > 0 (topography subtracted from the cells, calc_eulerian_data.m:10-24,139-141) is not provided");
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int n = S->n;
  int rc;
  DevBuf d_ext, d_ent, d_grid, d_out, d_flag;
  auto cleanup = [&]() { sz_release(d_ext); sz_release(d_ent); sz_release(d_grid); sz_release(d_out); sz_release(d_flag); };
  if ((rc = sz_reserve(ctx, d_ext, (size_t)n * 16 + 16))) return rc;
  hipLaunchKernelGGL(k_eul_extent, grid_for(n, 256), dim3(256), 0, st, S->F, (double *)d_ext.p);
  std::vector<double> ext((size_t)2 * n), x((size_t)n), y((size_t)n);
  std::vector<int> alive((size_t)n);
  hipError_t he = hipMemcpyAsync(ext.data(), d_ext.p, ext.size() * 8, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipMemcpyAsync(x.data(), S->F.Xi, (size_t)n * 8, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipMemcpyAsync(y.data(), S->F.Yi, (size_t)n * 8, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipMemcpyAsync(alive.data(), S->F.alive, (size_t)n * 4, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipStreamSynchronize(st);
  if (he != hipSuccess) { cleanup(); return sz_fail(ctx, SZ_ERR_HIP, "szEulerianData: %s", hipGetErrorString(he)); }
  double Lx = c2b[0], Ly = c2b[1], xmin = c2b[0], ymin = c2b[1];
  for (int k = 1; k < 5; ++k) { Lx = szc::maxd(Lx, c2b[2 * k]); Ly = szc::maxd(Ly, c2b[2 * k + 1]); xmin = szc::mind(xmin, c2b[2 * k]); ymin = szc::mind(ymin, c2b[2 * k + 1]); }
  // the live floes and their periodic images (:7-8, :29-69), with their extents translated along
  std::vector<EulEnt> E;
  std::vector<double> ex, ey;     // max |x|, max |y| of each entry's ring at its own position
  for (int i = 0; i < n; ++i) if (alive[i]) { EulEnt e = {x[i], y[i], i, 0}; E.push_back(e); ex.push_back(ext[2 * i]); ey.push_back(ext[2 * i + 1]); }
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
  const size_t cells = (size_t)Nx * Ny;
  if ((rc = sz_reserve(ctx, d_ent, (size_t)(M + 1) * sizeof(EulEnt))) || (rc = sz_reserve(ctx, d_grid, grid.size() * 8)) ||
      (rc = sz_reserve(ctx, d_out, EUL_FIELDS * cells * 8)) || (rc = sz_reserve(ctx, d_flag, 16))) { cleanup(); return rc; }
  he = hipMemcpyAsync(d_ent.p, E.data(), (size_t)M * sizeof(EulEnt), hipMemcpyHostToDevice, st);
  if (he == hipSuccess) he = hipMemcpyAsync(d_grid.p, grid.data(), grid.size() * 8, hipMemcpyHostToDevice, st);
  if (he == hipSuccess) he = hipMemsetAsync(d_out.p, 0, EUL_FIELDS * cells * 8, st);
  if (he == hipSuccess) he = hipMemsetAsync(d_flag.p, 0, 16, st);
  if (he == hipSuccess) he = hipEventRecord(ctx->ev0, st);
  if (he == hipSuccess && M > 0)
    hipLaunchKernelGGL(k_eul_cells, dim3((unsigned)cells), dim3(256), 0, st, S->F, (const EulEnt *)d_ent.p, M, Nx, Ny,
                       (const double *)d_grid.p, (const double *)d_grid.p + Nx + 1, r_max, (double *)d_out.p, (int *)d_flag.p);
  if (he == hipSuccess) he = hipEventRecord(ctx->ev1, st);
  int over = 0;
  if (he == hipSuccess) he = hipMemcpyAsync(out, d_out.p, EUL_FIELDS * cells * 8, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipMemcpyAsync(&over, d_flag.p, 4, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipStreamSynchronize(st);
  float ms = 0;
  if (he == hipSuccess) he = hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  cleanup();
  if (he != hipSuccess) return sz_fail(ctx, SZ_ERR_HIP, "szEulerianData: %s", hipGetErrorString(he));
  if (over) return sz_fail(ctx, SZ_ERR_WORKSPACE, "szEulerianData: a grid cell has more than %d potential floes (use a finer grid)", (int)EUL_CAP);
  if (kernel_ms) *kernel_ms = ms;
  return SZ_OK;
}
