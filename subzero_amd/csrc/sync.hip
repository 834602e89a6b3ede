// sync.hip -- the Floe struct array between the caller's memory and the GPU (B2 surface of include/subzero_hip.h):
//   szFloesUpload / szFloesResync     the whole array (Resync: bulk histories of unchanged floes stay on the device)
//   szFloesUpdate / szFloesFetch      field groups of a few floes, packed into one transfer and scattered / gathered
//   szFloesDownload                   field groups of all floes
//   szInteractionsDownload / szPairsDownload / szGhostsDownload / szMcEmptyDownload   results of the last step
// the state lives in HBM across steps (0.5 GB at 1e4 floes, 97 % of it the stress history and the
// Monte-Carlo points that MATLAB-side code only touches for the floes it reshapes), so what crosses PCIe after a
// MATLAB-side process is the scalars and rings (a few MB) plus the bulk rows of the floes that process changed.
#include "step_state.hpp"

int sz_step_configure(szContext *ctx, StepState *S);   // step.hip: tuning knobs and kernel attributes

namespace {

__global__ void k_counters_init(Floes F, Work W) {
  const int e = threadIdx.x * C_STRIDE;
  if (e < C_COUNT) W.counters[e] = (e == C_NEXT) ? (unsigned long long)F.n : 0ULL;
}
// sums of the stress history / points inside the Monte-Carlo mask of the floes ids[0..cnt) (all floes when ids == nullptr)
__global__ void k_stress_sum(Floes F, const int *ids, int cnt) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < cnt) stress_sum_init(ids ? ids[k] - 1 : k, F);
}
__global__ void k_mc_count(Floes F, Work W, const int *ids, int cnt) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < cnt) mc_count(ids ? ids[k] - 1 : k, F, W);
}

// ---- field groups of listed floes through one packed transfer ----------------------------------------------
// STATE block of one floe: NSD doubles then NSI ints (as doubles' neighbours in a second array)
enum { NSD = 20 + 4 + 4 + 2 + 1 + 1, NSI = 2 };
struct StatePtrs { double *d[20]; double *Stress, *strain, *cf, *ct, *ov; int *alive, *scount; };
template <class FloesLike>
__host__ __device__ inline StatePtrs state_ptrs(const FloesLike &F) {
  StatePtrs p;
  double *d[20] = {F.Xi, F.Yi, F.alpha_i, F.Ui, F.Vi, F.ksi_ice, F.h, F.area, F.mass, F.inertia_moment, F.rmax,
                   F.dXi_p, F.dYi_p, F.dUi_p, F.dVi_p, F.dalpha_i_p, F.dksi_ice_p, F.FxOA, F.FyOA, F.torqueOA};
  for (int q = 0; q < 20; ++q) p.d[q] = d[q];
  p.Stress = F.Stress; p.strain = F.strain; p.cf = F.collision_force; p.ct = F.collision_torque; p.ov = F.OverlapArea;
  p.alive = F.alive; p.scount = F.StressCount;
  return p;
}
// to_device: packed -> arrays (scatter); else arrays -> packed (gather).  A null array pointer is skipped.
__host__ __device__ inline void state_move(const StatePtrs &p, int i, double *pd, int *pi, bool to_device) {
#define MV(ptr, idx, slot) if (ptr) { if (to_device) (ptr)[idx] = pd[slot]; else pd[slot] = (ptr)[idx]; }
  for (int q = 0; q < 20; ++q) MV(p.d[q], i, q)
  for (int q = 0; q < 4; ++q) { MV(p.Stress, 4 * i + q, 20 + q) MV(p.strain, 4 * i + q, 24 + q) }
  MV(p.cf, 2 * i, 28) MV(p.cf, 2 * i + 1, 29) MV(p.ct, i, 30) MV(p.ov, i, 31)
#undef MV
  if (p.alive) { if (to_device) p.alive[i] = pi[0]; else pi[0] = p.alive[i]; }
  if (p.scount) { if (to_device) p.scount[i] = pi[1]; else pi[1] = p.scount[i]; }
}
static_assert(NSD == 32, "state block layout");
__global__ void k_state_move(Floes F, const int *ids, int cnt, double *pd, int *pi, int to_device) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < cnt) state_move(state_ptrs(F), ids[k] - 1, pd + (size_t)k * NSD, pi + (size_t)k * NSI, to_device != 0);
}
// rows of a [floe][row_len] array <-> packed rows, one workgroup-stride loop per listed floe
template <class T>
__global__ void k_rows_move(T *arr, size_t row_len, const int *ids, int cnt, T *packed, int to_device) {
  for (int k = blockIdx.y; k < cnt; k += gridDim.y) {     // (gridDim.y stops at 65535: the rows are strided over)
    T *row = arr + (size_t)(ids[k] - 1) * row_len, *pk = packed + (size_t)k * row_len;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < row_len; q += (size_t)gridDim.x * blockDim.x) {
      if (to_device) row[q] = pk[q]; else pk[q] = row[q];
    }
  }
}
// ring vertices of the listed floes: packed back to back in list order, off[k] = vertices before floe k of the list
__global__ void k_ring_move(Floes F, const int *ids, const int *off, int cnt, double *packed, int total, int to_device) {
  double *arrs[4] = {F.c0x, F.c0y, F.cax, F.cay};
  for (int k = blockIdx.y; k < cnt; k += gridDim.y) {
    const int i = ids[k] - 1, v0 = F.vert_off[i], nv = F.vert_off[i + 1] - v0;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nv; q += gridDim.x * blockDim.x)
      for (int a = 0; a < 4; ++a) {
        double *pk = packed + (size_t)a * total + off[k] + q;
        if (to_device) arrs[a][v0 + q] = *pk; else *pk = arrs[a][v0 + q];
      }
  }
}
// bulk histories of the new array taken from the previous one on the device (szFloesResync): new floe k <- old floe src[k]
template <class T>
__global__ void k_rows_inherit(T *dst, const T *old, size_t row_len, const int *src, int n_new) {
  for (int k = blockIdx.y; k < n_new; k += gridDim.y) {
    if (src[k] <= 0) continue;
    const T *from = old + (size_t)(src[k] - 1) * row_len;
    T *to = dst + (size_t)k * row_len;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < row_len; q += (size_t)gridDim.x * blockDim.x) to[q] = from[q];
  }
}

void destroy_state(StepState *S) {
  if (!S) return;
  for (DevBuf &b : S->bufs) sz_release(b);
  DevBuf *more[] = {&S->oXo, &S->oYo, &S->oU, &S->oV, &S->oWu, &S->oWv, &S->pair_i, &S->pair_j, &S->pair_nvoff,
                    &S->pair_scr, &S->pair_mult, &S->pair_nrows, &S->pair_status, &S->pair_flag, &S->retry_list,
                    &S->retry_copy, &S->big_list, &S->team_list, &S->wide_list, &S->live_list, &S->cold_list, &S->wall_list, &S->scratch, &S->rows, &S->cell_cnt, &S->cell_off, &S->cell_fill, &S->cell_item,
                    &S->pair_rowoff, &S->x_off, &S->touch_list, &S->pair_nreg, &S->pair_task, &S->filter_spill, &S->pot_list, &S->inc_list,
                    &S->stage_d, &S->stage_i, &S->stage_ids, &S->sp_masks, &S->sp_edges_dev, &S->sp_counts,
                    &S->gh_mask, &S->gh_agg, &S->pair_agg, &S->row_agg, &S->rel_mask, &S->pot_nvpre, &S->force_list, &S->hint};
  for (DevBuf *b : more) sz_release(*b);
  for (DevBuf &b : S->post) sz_release(b);
  for (hipEvent_t &e : S->ev) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t &e : S->ev_prev) if (e) (void)hipEventDestroy(e);
  if (S->wall_fork) (void)hipEventDestroy(S->wall_fork);
  if (S->wall_join) (void)hipEventDestroy(S->wall_join);
  if (S->c) (void)hipHostFree(S->c);
  if (S->sp_hdr) (void)hipHostFree(S->sp_hdr);
  delete S;
}

bool valid_ids(const StepState *S, const int32_t *ids, int nids) {
  for (int k = 0; k < nids; ++k) if (ids[k] < 1 || ids[k] > S->n) return false;
  return true;
}

// Builds the device image of `h`.  With `old` and `src`, the stress history and the Monte-Carlo points of floe k are
// copied on the device from floe src[k] of `old` when src[k] > 0 and read from the host arrays only for src[k] == 0.
int build_state(szContext *ctx, const szFloeState *h, StepState *old, const int32_t *src, StepState **out) {
  StepState *S = new StepState();
  *out = S;
  for (hipEvent_t &e : S->ev) SZ_HIP(ctx, hipEventCreate(&e));
  for (hipEvent_t &e : S->ev_prev) SZ_HIP(ctx, hipEventCreate(&e));
  { int rc0 = dev_alloc(ctx, S, S->scan_agg, 1024); if (rc0) return rc0; }
  SZ_HIP(ctx, hipHostMalloc((void **)&S->c, (C_COUNT + C_STRIDE) * sizeof(unsigned long long), hipHostMallocDefault));   // (+ the flag of k_publish)
  memset(S->c, 0, (C_COUNT + C_STRIDE) * sizeof(unsigned long long));
  SZ_HIP(ctx, hipHostGetDevicePointer((void **)&S->c_dev, S->c, 0));
  int rc;
  if ((rc = sz_step_configure(ctx, S))) return rc;
  if (old) S->pot_cap = old->pot_cap;          // list capacities a previous step had to grow to stay
  const int n = h->n;
  const int nv = h->vert_off[n];
  S->n = n; S->nb = h->nb; S->n_stress = h->n_stress; S->n_mc = h->n_mc; S->nverts = nv;
  S->bufs.reserve(128);
  Floes &F = S->F;
  F.n = n; F.nb = h->nb; F.n_stress = h->n_stress; F.n_mc = h->n_mc;
#define PUT(field, count) if ((rc = dev_put(ctx, S, F.field, h->field, (size_t)(count)))) return rc;
  PUT(Xi, n) PUT(Yi, n) PUT(alpha_i, n) PUT(Ui, n) PUT(Vi, n) PUT(ksi_ice, n)
  PUT(h, n) PUT(area, n) PUT(mass, n) PUT(inertia_moment, n) PUT(rmax, n)
  PUT(dXi_p, n) PUT(dYi_p, n) PUT(dUi_p, n) PUT(dVi_p, n) PUT(dalpha_i_p, n) PUT(dksi_ice_p, n)
  PUT(FxOA, n) PUT(FyOA, n) PUT(torqueOA, n)
  PUT(alive, n) PUT(vert_off, n + 1)
  PUT(c0x, nv) PUT(c0y, nv) PUT(cax, nv) PUT(cay, nv)
  PUT(StressCount, n) PUT(Stress, 4 * (size_t)n) PUT(strain, 4 * (size_t)n)
  PUT(collision_force, 2 * (size_t)n) PUT(collision_torque, n) PUT(OverlapArea, n)
  F.mc_si = h->n_mc; F.mc_sq = 1; // [floe][point]: the forcing kernel reads one floe's points with one wave
  const size_t sh_row = (size_t)h->n_stress * 4, mc_row = (size_t)h->n_mc;
  if (!old) {
    PUT(StressH, (size_t)n * sh_row) PUT(mcX, (size_t)n * mc_row) PUT(mcY, (size_t)n * mc_row) PUT(mcA, (size_t)n * mc_row)
  } else {
    if ((rc = dev_alloc(ctx, S, F.StressH, (size_t)n * sh_row)) || (rc = dev_alloc(ctx, S, F.mcX, (size_t)n * mc_row)) ||
        (rc = dev_alloc(ctx, S, F.mcY, (size_t)n * mc_row)) || (rc = dev_alloc(ctx, S, F.mcA, (size_t)n * mc_row))) return rc;
    int *d_src = nullptr;
    if ((rc = dev_put(ctx, S, d_src, src, (size_t)n))) return rc;
    const dim3 g(8, (unsigned)(n < 65535 ? n : 65535));
    hipLaunchKernelGGL(k_rows_inherit<double>, g, dim3(256), 0, ctx->stream, F.StressH, (const double *)old->F.StressH, sh_row, (const int *)d_src, n);
    hipLaunchKernelGGL(k_rows_inherit<double>, g, dim3(256), 0, ctx->stream, F.mcX, (const double *)old->F.mcX, mc_row, (const int *)d_src, n);
    hipLaunchKernelGGL(k_rows_inherit<double>, g, dim3(256), 0, ctx->stream, F.mcY, (const double *)old->F.mcY, mc_row, (const int *)d_src, n);
    hipLaunchKernelGGL(k_rows_inherit<unsigned char>, g, dim3(256), 0, ctx->stream, F.mcA, (const unsigned char *)old->F.mcA, mc_row, (const int *)d_src, n);
    SZ_HIP(ctx, hipGetLastError());
    // the floes without a predecessor: their rows come from the host, run by run of consecutive floes
    for (int k = 0; k < n;) {
      if (src[k] > 0) { ++k; continue; }
      int e = k;
      while (e < n && src[e] <= 0) ++e;
      const size_t cnt = (size_t)(e - k);
      SZ_HIP(ctx, hipMemcpyAsync(F.StressH + (size_t)k * sh_row, h->StressH + (size_t)k * sh_row, cnt * sh_row * 8, hipMemcpyHostToDevice, ctx->stream));
      SZ_HIP(ctx, hipMemcpyAsync(F.mcX + (size_t)k * mc_row, h->mcX + (size_t)k * mc_row, cnt * mc_row * 8, hipMemcpyHostToDevice, ctx->stream));
      SZ_HIP(ctx, hipMemcpyAsync(F.mcY + (size_t)k * mc_row, h->mcY + (size_t)k * mc_row, cnt * mc_row * 8, hipMemcpyHostToDevice, ctx->stream));
      SZ_HIP(ctx, hipMemcpyAsync(F.mcA + (size_t)k * mc_row, h->mcA + (size_t)k * mc_row, cnt * mc_row, hipMemcpyHostToDevice, ctx->stream));
      k = e;
    }
  }
#undef PUT
  if ((rc = dev_alloc(ctx, S, F.kill, 4 * (size_t)n))) return rc;
  if ((rc = dev_alloc(ctx, S, F.transfer, 4 * (size_t)n))) return rc;
  if ((rc = dev_alloc(ctx, S, F.StressSum, 8 * (size_t)n))) return rc;
  Work &W = S->W;
  memset(&W, 0, sizeof(W));
  const size_t cap = 4 * (size_t)n;
  W.cap_ext = (int)cap; W.pot_cap = S->pot_cap;
#define NEW(field, count) if ((rc = dev_alloc(ctx, S, W.field, (size_t)(count)))) return rc;
  NEW(ext_src, cap) NEW(ext_num, cap) NEW(ext_parent, cap) NEW(xg_of, cap) NEW(yg_of, cap) NEW(ext_nv, cap)
  NEW(ext_x, cap) NEW(ext_y, cap) NEW(ext_rec, (size_t)REC_WORDS * cap) NEW(ext_ring, 2 * (size_t)RING_SLOTS * cap) NEW(flag, cap) NEW(scan, cap) NEW(counters, C_COUNT)
  NEW(pot_cnt, cap)
  NEW(pair_cnt, cap) NEW(pair_nv, cap) NEW(pair_base, cap + 1) NEW(nv_base, cap + 1)
  NEW(inc_cnt, cap) NEW(inc_pre, cap) NEW(mc_cnt, n) NEW(mc_list, n)
  NEW(row_cnt, cap) NEW(row_off, cap + 1) NEW(cf_ext, 2 * cap) NEW(ct_ext, cap) NEW(ov_ext, cap) NEW(tmax, cap)
  NEW(st_ext, 3 * (size_t)n) NEW(st_flag, n)
#undef NEW
  W.mc_list_cap = n;
  if ((rc = bind_lists(ctx, S))) return rc;
  S->h_rmax.assign(h->rmax, h->rmax + n);
  S->h_vert_off.assign(h->vert_off, h->vert_off + n + 1);
  rmax_stats(S, S->h_rmax.data());
  hipLaunchKernelGGL(k_counters_init, dim3(1), dim3(64), 0, ctx->stream, F, W);
  hipLaunchKernelGGL(k_mc_count, grid_for(n, 256), dim3(256), 0, ctx->stream, F, W, (const int *)nullptr, n);
  hipLaunchKernelGGL(k_stress_sum, grid_for(n, 256), dim3(256), 0, ctx->stream, F, (const int *)nullptr, n);
  SZ_HIP(ctx, hipGetLastError());
  SZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  S->counters_clean = true;
  return SZ_OK;
}

int check_image(szContext *ctx, const szFloeState *h, const char *who) {
  if (!ctx || !h) return SZ_ERR_BAD_ARG;
  if (h->n <= 0 || h->nb < 0 || h->nb > h->n || h->n_stress <= 0 || h->n_mc <= 0 || !h->vert_off)
    return sz_fail(ctx, SZ_ERR_BAD_ARG, "%s: bad sizes", who);
  return SZ_OK;
}

// listed floes' field groups between `h` (the caller's full image) and the device, through packed staging buffers
int move_fields(szContext *ctx, StepState *S, szFloeState *h, const int32_t *ids, int nids, int what, bool to_device) {
  hipStream_t st = ctx->stream;
  int rc;
  if ((rc = sz_reserve(ctx, S->stage_ids, (size_t)nids * 8 + 16))) return rc;
  int *d_ids = (int *)S->stage_ids.p, *d_off = d_ids + nids;
  SZ_HIP(ctx, hipMemcpyAsync(d_ids, ids, (size_t)nids * 4, hipMemcpyHostToDevice, st));
  const Floes &F = S->F;
  const hipMemcpyKind dir = to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
  if (what & SZ_F_STATE) {
    std::vector<double> pd((size_t)nids * NSD);
    std::vector<int> pi((size_t)nids * NSI);
    const StatePtrs hp = state_ptrs(*h);
    if ((rc = sz_reserve(ctx, S->stage_d, pd.size() * 8)) || (rc = sz_reserve(ctx, S->stage_i, pi.size() * 4))) return rc;
    if (to_device) {
      for (int k = 0; k < nids; ++k) state_move(hp, ids[k] - 1, &pd[(size_t)k * NSD], &pi[(size_t)k * NSI], false);
      // (a field the caller left null keeps its device value: read it back first so that the scatter rewrites the same)
      bool any_null = !h->alive || !h->StressCount || !h->Stress || !h->strain || !h->collision_force || !h->collision_torque || !h->OverlapArea;
      for (int q = 0; q < 20; ++q) any_null = any_null || !hp.d[q];
      if (any_null) {
        std::vector<double> cd(pd.size()); std::vector<int> ci(pi.size());
        hipLaunchKernelGGL(k_state_move, grid_for(nids, 256), dim3(256), 0, st, F, (const int *)d_ids, nids, (double *)S->stage_d.p, (int *)S->stage_i.p, 0);
        SZ_HIP(ctx, hipMemcpyAsync(cd.data(), S->stage_d.p, cd.size() * 8, hipMemcpyDeviceToHost, st));
        SZ_HIP(ctx, hipMemcpyAsync(ci.data(), S->stage_i.p, ci.size() * 4, hipMemcpyDeviceToHost, st));
        SZ_HIP(ctx, hipStreamSynchronize(st));
        for (int k = 0; k < nids; ++k) {
          double *a = &pd[(size_t)k * NSD]; const double *b = &cd[(size_t)k * NSD];
          for (int q = 0; q < 20; ++q) if (!hp.d[q]) a[q] = b[q];
          for (int q = 0; q < 4; ++q) { if (!hp.Stress) a[20 + q] = b[20 + q]; if (!hp.strain) a[24 + q] = b[24 + q]; }
          if (!hp.cf) { a[28] = b[28]; a[29] = b[29]; }
          if (!hp.ct) a[30] = b[30];
          if (!hp.ov) a[31] = b[31];
          if (!hp.alive) pi[(size_t)k * NSI] = ci[(size_t)k * NSI];
          if (!hp.scount) pi[(size_t)k * NSI + 1] = ci[(size_t)k * NSI + 1];
        }
      }
      SZ_HIP(ctx, hipMemcpyAsync(S->stage_d.p, pd.data(), pd.size() * 8, hipMemcpyHostToDevice, st));
      SZ_HIP(ctx, hipMemcpyAsync(S->stage_i.p, pi.data(), pi.size() * 4, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_state_move, grid_for(nids, 256), dim3(256), 0, st, F, (const int *)d_ids, nids, (double *)S->stage_d.p, (int *)S->stage_i.p, 1);
      SZ_HIP(ctx, hipStreamSynchronize(st));
      if (h->rmax) {
        for (int k = 0; k < nids; ++k) S->h_rmax[ids[k] - 1] = h->rmax[ids[k] - 1];
        rmax_stats(S, S->h_rmax.data());
      }
    } else {
      hipLaunchKernelGGL(k_state_move, grid_for(nids, 256), dim3(256), 0, st, F, (const int *)d_ids, nids, (double *)S->stage_d.p, (int *)S->stage_i.p, 0);
      SZ_HIP(ctx, hipMemcpyAsync(pd.data(), S->stage_d.p, pd.size() * 8, hipMemcpyDeviceToHost, st));
      SZ_HIP(ctx, hipMemcpyAsync(pi.data(), S->stage_i.p, pi.size() * 4, hipMemcpyDeviceToHost, st));
      SZ_HIP(ctx, hipStreamSynchronize(st));
      for (int k = 0; k < nids; ++k) state_move(hp, ids[k] - 1, &pd[(size_t)k * NSD], &pi[(size_t)k * NSI], true);
    }
  }
  if (what & SZ_F_RING) {
    if (!h->vert_off || !h->c0x || !h->c0y || !h->cax || !h->cay) return sz_fail(ctx, SZ_ERR_BAD_ARG, "ring fields missing");
    std::vector<int> off((size_t)nids + 1, 0);
    for (int k = 0; k < nids; ++k) {
      const int i = ids[k] - 1, nv = S->h_vert_off[i + 1] - S->h_vert_off[i];
      if (h->vert_off[i + 1] - h->vert_off[i] != nv || h->vert_off[i] != S->h_vert_off[i])
        return sz_fail(ctx, SZ_ERR_BAD_ARG, "floe %d: the ring layout differs from the uploaded one (a ring that changes length needs szFloesResync)", ids[k]);
      off[k + 1] = off[k] + nv;
    }
    const int total = off[nids];
    std::vector<double> pk((size_t)4 * total);
    const double *arrs[4] = {h->c0x, h->c0y, h->cax, h->cay};
    double *warrs[4] = {h->c0x, h->c0y, h->cax, h->cay};
    if ((rc = sz_reserve(ctx, S->stage_d, pk.size() * 8 + 16))) return rc;
    SZ_HIP(ctx, hipMemcpyAsync(d_off, off.data(), (size_t)nids * 4, hipMemcpyHostToDevice, st));
    if (to_device) {
      for (int a = 0; a < 4; ++a)
        for (int k = 0; k < nids; ++k)
          memcpy(&pk[(size_t)a * total + off[k]], arrs[a] + S->h_vert_off[ids[k] - 1], (size_t)(off[k + 1] - off[k]) * 8);
      SZ_HIP(ctx, hipMemcpyAsync(S->stage_d.p, pk.data(), pk.size() * 8, hipMemcpyHostToDevice, st));
    }
    hipLaunchKernelGGL(k_ring_move, dim3(2, (unsigned)(nids < 65535 ? nids : 65535)), dim3(64), 0, st, F, (const int *)d_ids, (const int *)d_off, nids, (double *)S->stage_d.p, total, to_device ? 1 : 0);
    if (!to_device) {
      SZ_HIP(ctx, hipMemcpyAsync(pk.data(), S->stage_d.p, pk.size() * 8, hipMemcpyDeviceToHost, st));
      SZ_HIP(ctx, hipStreamSynchronize(st));
      for (int a = 0; a < 4; ++a)
        for (int k = 0; k < nids; ++k)
          memcpy(warrs[a] + S->h_vert_off[ids[k] - 1], &pk[(size_t)a * total + off[k]], (size_t)(off[k + 1] - off[k]) * 8);
    } else SZ_HIP(ctx, hipStreamSynchronize(st));
  }
  // bulk rows: packed on the host, one transfer per array, moved by a workgroup-stride kernel
  struct Bulk { int flag; void *dev; void *host; size_t row_len, elem; };
  const Bulk bulk[4] = {{SZ_F_STRESSH, F.StressH, h->StressH, (size_t)S->n_stress * 4, 8},
                        {SZ_F_MC, F.mcX, h->mcX, (size_t)S->n_mc, 8}, {SZ_F_MC, F.mcY, h->mcY, (size_t)S->n_mc, 8},
                        {SZ_F_MC, F.mcA, h->mcA, (size_t)S->n_mc, 1}};
  for (const Bulk &b : bulk) {
    if (!(what & b.flag)) continue;
    if (!b.host) return sz_fail(ctx, SZ_ERR_BAD_ARG, "bulk field missing in the host image");
    const size_t row_bytes = b.row_len * b.elem;
    std::vector<char> pk((size_t)nids * row_bytes);
    if ((rc = sz_reserve(ctx, S->stage_d, pk.size() + 16))) return rc;
    if (to_device) {
      for (int k = 0; k < nids; ++k) memcpy(&pk[(size_t)k * row_bytes], (const char *)b.host + (size_t)(ids[k] - 1) * row_bytes, row_bytes);
      SZ_HIP(ctx, hipMemcpyAsync(S->stage_d.p, pk.data(), pk.size(), dir, st));
    }
    const dim3 g(8, (unsigned)(nids < 65535 ? nids : 65535));
    if (b.elem == 8) hipLaunchKernelGGL(k_rows_move<double>, g, dim3(256), 0, st, (double *)b.dev, b.row_len, (const int *)d_ids, nids, (double *)S->stage_d.p, to_device ? 1 : 0);
    else hipLaunchKernelGGL(k_rows_move<unsigned char>, g, dim3(256), 0, st, (unsigned char *)b.dev, b.row_len, (const int *)d_ids, nids, (unsigned char *)S->stage_d.p, to_device ? 1 : 0);
    if (!to_device) {
      SZ_HIP(ctx, hipMemcpyAsync(pk.data(), S->stage_d.p, pk.size(), dir, st));
      SZ_HIP(ctx, hipStreamSynchronize(st));
      for (int k = 0; k < nids; ++k) memcpy((char *)b.host + (size_t)(ids[k] - 1) * row_bytes, &pk[(size_t)k * row_bytes], row_bytes);
    } else SZ_HIP(ctx, hipStreamSynchronize(st));
  }
  if (to_device && (what & SZ_F_STRESSH)) hipLaunchKernelGGL(k_stress_sum, grid_for(nids, 256), dim3(256), 0, st, F, (const int *)d_ids, nids);
  if (to_device && (what & SZ_F_MC)) hipLaunchKernelGGL(k_mc_count, grid_for(nids, 256), dim3(256), 0, st, F, S->W, (const int *)d_ids, nids);
  SZ_HIP(ctx, hipGetLastError());
  SZ_HIP(ctx, hipStreamSynchronize(st));
  return SZ_OK;
}

} // namespace

void sz_step_destroy(szContext *ctx) {
  destroy_state(ctx->step);
  ctx->step = nullptr;
}

extern "C" {

int szFloesUpload(szContext *ctx, const szFloeState *h) {
  int rc = check_image(ctx, h, "szFloesUpload");
  if (rc) return rc;
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  SZ_HIP(ctx, sz_sync_all(ctx));
  sz_step_destroy(ctx);
  StepState *S = nullptr;
  rc = build_state(ctx, h, nullptr, nullptr, &S);
  if (rc) { destroy_state(S); return rc; }
  ctx->step = S;
  return SZ_OK;
}

int szFloesResync(szContext *ctx, const szFloeState *h, const int32_t *src) {
  int rc = check_image(ctx, h, "szFloesResync");
  if (rc) return rc;
  StepState *old = ctx->step;
  if (!old || !src) return szFloesUpload(ctx, h);
  if (h->n_stress != old->n_stress || h->n_mc != old->n_mc)
    return sz_fail(ctx, SZ_ERR_BAD_ARG, "szFloesResync: history / Monte-Carlo sizes differ from the resident array");
  for (int k = 0; k < h->n; ++k)
    if (src[k] < 0 || src[k] > old->n) return sz_fail(ctx, SZ_ERR_BAD_ARG, "szFloesResync: src[%d] = %d is not a floe of the resident array", k, src[k]);
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  SZ_HIP(ctx, sz_sync_all(ctx));
  StepState *S = nullptr;
  rc = build_state(ctx, h, old, src, &S);
  if (rc) { destroy_state(S); return rc; }   // the resident array is still the old one
  destroy_state(old);
  ctx->step = S;
  return SZ_OK;
}

int szFloesUpdate(szContext *ctx, const szFloeState *h, const int32_t *ids, int32_t nids, int what) {
  if (!ctx || !h || (nids > 0 && !ids) || nids < 0) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S) return sz_fail(ctx, SZ_ERR_STATE, "szFloesUpdate: no floes uploaded");
  if (h->n != S->n) return sz_fail(ctx, SZ_ERR_BAD_ARG, "szFloesUpdate: n differs from the resident array (szFloesResync handles that)");
  if (!valid_ids(S, ids, nids)) return sz_fail(ctx, SZ_ERR_BAD_ARG, "szFloesUpdate: floe number out of range");
  if (nids == 0) return SZ_OK;
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  S->ahead_valid = false;      // (what the step before queued ahead for the next one has read the floes as they were)
  return move_fields(ctx, S, const_cast<szFloeState *>(h), ids, nids, what, true);
}

int szFloesFetch(szContext *ctx, szFloeState *h, const int32_t *ids, int32_t nids, int what) {
  if (!ctx || !h || (nids > 0 && !ids) || nids < 0) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S) return sz_fail(ctx, SZ_ERR_STATE, "szFloesFetch: no floes uploaded");
  if (h->n != S->n) return sz_fail(ctx, SZ_ERR_BAD_ARG, "szFloesFetch: n mismatch");
  if (!valid_ids(S, ids, nids)) return sz_fail(ctx, SZ_ERR_BAD_ARG, "szFloesFetch: floe number out of range");
  if (nids == 0) return SZ_OK;
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  return move_fields(ctx, S, h, ids, nids, what, false);
}

int szFloesDownload(szContext *ctx, szFloeState *h, int what) {
  if (!ctx || !h) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S) return sz_fail(ctx, SZ_ERR_STATE, "szFloesDownload: no floes uploaded");
  if (h->n != S->n) return sz_fail(ctx, SZ_ERR_BAD_ARG, "szFloesDownload: n mismatch");
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  SZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const Floes &F = S->F;
  const size_t n = (size_t)S->n, nv = (size_t)S->nverts;
#define GET(field, count) if (h->field) SZ_HIP(ctx, hipMemcpy(h->field, F.field, (size_t)(count) * sizeof(*h->field), hipMemcpyDeviceToHost));
  if (what & SZ_SYNC_DYNAMIC) {
    GET(Xi, n) GET(Yi, n) GET(alpha_i, n) GET(Ui, n) GET(Vi, n) GET(ksi_ice, n)
    GET(h, n) GET(area, n) GET(mass, n) GET(inertia_moment, n) GET(rmax, n)
    GET(dXi_p, n) GET(dYi_p, n) GET(dUi_p, n) GET(dVi_p, n) GET(dalpha_i_p, n) GET(dksi_ice_p, n)
    GET(FxOA, n) GET(FyOA, n) GET(torqueOA, n) GET(alive, n)
    GET(cax, nv) GET(cay, nv)
    GET(StressCount, n) GET(Stress, 4 * n) GET(strain, 4 * n)
    GET(collision_force, 2 * n) GET(collision_torque, n) GET(OverlapArea, n)
    GET(kill, 4 * n) GET(transfer, 4 * n)
  }
  if (what & SZ_SYNC_STRESSH) { GET(StressH, n * S->n_stress * 4) }
#undef GET
  return SZ_OK;
}

int szInteractionsDownload(szContext *ctx, int32_t *inter_off, double *inter, int64_t cap_rows) {
  if (!ctx || !inter_off) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S || !S->stepped) return sz_fail(ctx, SZ_ERR_STATE, "szInteractionsDownload: no step taken");
  SZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int n = S->n;
  std::vector<i64> off((size_t)n + 1);
  SZ_HIP(ctx, hipMemcpy(off.data(), S->W.row_off, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost));
  for (int i = 0; i <= n; ++i) inter_off[i] = (int32_t)off[i];
  if (off[n] > cap_rows) return sz_fail(ctx, SZ_ERR_CAPACITY, "szInteractionsDownload: %lld rows, capacity %lld", (long long)off[n], (long long)cap_rows);
  if (off[n] > 0) {
    if (!inter) return SZ_ERR_BAD_ARG;
    SZ_HIP(ctx, hipMemcpy(inter, S->W.rows, (size_t)off[n] * 56, hipMemcpyDeviceToHost));
  }
  return SZ_OK;
}

int szPairsDownload(szContext *ctx, int32_t *pair_i, int32_t *pair_j, int64_t cap, int64_t *npairs, int32_t *n_ext) {
  if (!ctx || !npairs || !n_ext) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S || !S->stepped) return sz_fail(ctx, SZ_ERR_STATE, "szPairsDownload: no step taken");
  SZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const i64 np = S->last_npairs;
  std::vector<int> pi((size_t)np + 1), pj((size_t)np + 1);
  if (np) {
    SZ_HIP(ctx, hipMemcpy(pi.data(), S->W.pair_i, (size_t)np * 4, hipMemcpyDeviceToHost));
    SZ_HIP(ctx, hipMemcpy(pj.data(), S->W.pair_j, (size_t)np * 4, hipMemcpyDeviceToHost));
  }
  int64_t k = 0;
  for (i64 p = 0; p < np; ++p) {
    if (pj[p] == WALL) continue;
    if (k < cap) { pair_i[k] = pi[p] + 1; pair_j[k] = pj[p] + 1; }
    ++k;
  }
  *npairs = k;
  *n_ext = S->last_next;
  if (k > cap) return sz_fail(ctx, SZ_ERR_CAPACITY, "szPairsDownload: %lld pairs, capacity %lld", (long long)k, (long long)cap);
  return SZ_OK;
}

int szGhostsDownload(szContext *ctx, int32_t *src, int32_t *parent, double *Xi, double *Yi, int64_t cap, int64_t *nghost) {
  if (!ctx || !nghost) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S || !S->stepped) return sz_fail(ctx, SZ_ERR_STATE, "szGhostsDownload: no step taken");
  if (S->ghosts_overwritten)
    return sz_fail(ctx, SZ_ERR_STATE, "szGhostsDownload: the last step was asked to queue the next one ahead (SZ_STEP_QUEUE_AHEAD): the ghost list it made has been overwritten");
  SZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int n = S->n;
  const i64 g = (i64)S->last_next - n;
  *nghost = g;
  if (g > cap) return sz_fail(ctx, SZ_ERR_CAPACITY, "szGhostsDownload: %lld ghost floes, capacity %lld", (long long)g, (long long)cap);
  if (g <= 0) return SZ_OK;
  std::vector<int> s((size_t)g), p((size_t)g);
  SZ_HIP(ctx, hipMemcpy(s.data(), S->W.ext_src + n, (size_t)g * 4, hipMemcpyDeviceToHost));
  SZ_HIP(ctx, hipMemcpy(p.data(), S->W.ext_parent + n, (size_t)g * 4, hipMemcpyDeviceToHost));
  for (i64 k = 0; k < g; ++k) { if (src) src[k] = s[k] + 1; if (parent) parent[k] = p[k] + 1; }
  if (Xi) SZ_HIP(ctx, hipMemcpy(Xi, S->W.ext_x + n, (size_t)g * 8, hipMemcpyDeviceToHost));
  if (Yi) SZ_HIP(ctx, hipMemcpy(Yi, S->W.ext_y + n, (size_t)g * 8, hipMemcpyDeviceToHost));
  return SZ_OK;
}

int szMcEmptyDownload(szContext *ctx, int32_t *ids, int64_t cap, int64_t *count) {
  if (!ctx || !count) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S) return sz_fail(ctx, SZ_ERR_STATE, "szMcEmptyDownload: no floes uploaded");
  *count = (int64_t)S->mc_empty.size();
  if (*count > cap) return sz_fail(ctx, SZ_ERR_CAPACITY, "szMcEmptyDownload: %lld floes, capacity %lld", (long long)*count, (long long)cap);
  for (size_t k = 0; k < S->mc_empty.size(); ++k) ids[k] = S->mc_empty[k];
  return SZ_OK;
}

} // extern "C"
