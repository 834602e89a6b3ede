// spatial.hip -- C ABI of the spatially sharded step (spatial_core.hpp has the design and the kernel bodies):
//   szSpatialInit      strips, ownership of the uploaded floes, masks
//   szSpatialPack      this step's two messages (halo + floes that changed strip) into device buffers
//   szSpatialUnpack    the neighbours' messages into the arrays
// and then the ordinary szStep.  The exchange between Pack and Unpack is the caller's (RCCL send / recv over xGMI:
// subzero_amd/multigpu.py SpatialInteractions; gloo in the CPU tests).
#include "step_state.hpp"
#include "spatial_core.hpp"

namespace {

static_assert((int)SP_HDR == (int)SP_HDR_WORDS, "step_state.hpp sizes the pinned header block");
__global__ void k_sp_header_clear(SpatialMsg l, SpatialMsg r, unsigned long long *veto, long long *veto_host) {
  if (threadIdx.x < SP_HDR) { l.hdr[threadIdx.x] = 0; r.hdr[threadIdx.x] = 0; }
  if (threadIdx.x == 0 && veto) { *veto = 0ULL; *veto_host = 0; }
}
// szSpatialUnpackQueued: the records of a received message into the arrays when the message is what was agreed to travel
// (the header within `plan`, no floe changing strip: its rows would travel in a second round, no overflow on the sender's
// side) -- otherwise nothing is taken and the veto is raised: the step queued behind does not run its back half, and the
// host, which has not looked at the header, finds out at the step's one read-back.  Launched for the planned number of
// records; the header goes to the pinned block for the host to read then.
__global__ void k_sp_take_queued(Floes F, Work W, SpatialMsg m, unsigned char *present, unsigned char *owned, long long plan_rec,
                                 long long plan_dbl, long long plan_migr, unsigned long long *veto, long long *hdr_host, long long *veto_host) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long nrec = m.hdr[0];
  const bool fits = m.hdr[3] == SP_OK && nrec <= plan_rec && m.hdr[1] <= plan_dbl && m.hdr[2] <= plan_migr && nrec <= m.cap_rec;
  if (k < SP_HDR) hdr_host[k] = m.hdr[k];
  if (!fits) { if (k == 0) { *veto = 1ULL; *veto_host = 1; } return; }
  if (k < nrec) sp_take(k, F, W, m, present, owned);
}
__global__ void k_sp_classify(Floes F, Work W, SpatialGeom g, const double *edges, unsigned char *present, unsigned char *owned,
                              SpatialMsg l, SpatialMsg r) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < F.n) sp_classify(i, F, W, g, edges, present, owned, l, r);
}
// (veto: szSpatialPackQueued -- a message of mine that overflowed its capacities must keep my own step from running too:
// the ownership is given back afterwards, szSpatialPackFinish, and the state must still be the one that was packed)
__global__ void k_sp_seal(SpatialMsg l, SpatialMsg r, long long *host_copy, unsigned long long *veto, long long *veto_host) {
  if (threadIdx.x == 0) {
    sp_seal(l); sp_seal(r);
    if (veto && (l.hdr[3] != SP_OK || r.hdr[3] != SP_OK)) { *veto = 1ULL; *veto_host = 1; }
  }
  __syncthreads();
  if (threadIdx.x < SP_HDR) { host_copy[threadIdx.x] = l.hdr[threadIdx.x]; host_copy[SP_HDR + threadIdx.x] = r.hdr[threadIdx.x]; }
}
// bulk rows of the migrants of a message: one workgroup per record, out of (pack) or into (take) the floe's rows
__global__ void k_sp_bulk(Floes F, Work W, SpatialMsg m, int pack) {
  const long long k = blockIdx.x;
  if (k >= m.hdr[0]) return;
  const double *r = m.dbl + m.rec_off[k];
  if (r[1] == 0.0) return;
  const int i = (int)r[0];
  const long long slot = (long long)r[3];
  if (slot < 0 || slot >= m.cap_migr) return;
  char *row = m.bulk + slot * m.bulk_row;
  for (size_t w = threadIdx.x; w < (size_t)m.bulk_row / 8; w += blockDim.x) sp_bulk_word(i, w, pack != 0, F, row, W);
}
// ... into the floe's rows by a queued unpack: only of a message that k_sp_take_queued has taken (the same test)
__global__ void k_sp_bulk_take_queued(Floes F, Work W, SpatialMsg m, long long plan_rec, long long plan_dbl, long long plan_migr) {
  const long long k = blockIdx.x;
  const long long nrec = m.hdr[0];
  const bool fits = m.hdr[3] == SP_OK && nrec <= plan_rec && m.hdr[1] <= plan_dbl && m.hdr[2] <= plan_migr && nrec <= m.cap_rec;
  if (!fits || k >= nrec || m.hdr[2] == 0) return;
  const double *r = m.dbl + m.rec_off[k];
  if (r[1] == 0.0) return;
  const int i = (int)r[0];
  const long long slot = (long long)r[3];
  if (slot < 0 || slot >= m.cap_migr) return;
  char *row = m.bulk + slot * m.bulk_row;
  for (size_t w = threadIdx.x; w < (size_t)m.bulk_row / 8; w += blockDim.x) sp_bulk_word(i, w, false, F, row, W);
}
__global__ void k_sp_take(Floes F, Work W, SpatialMsg m, unsigned char *present, unsigned char *owned) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < m.hdr[0] && k < m.cap_rec) sp_take(k, F, W, m, present, owned);
}
__global__ void k_sp_count(Floes F, const unsigned char *present, const unsigned char *owned, unsigned long long *out2) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F.n) return;
  if (owned[i]) atomicAdd(&out2[0], 1ULL);
  if (present[i]) atomicAdd(&out2[1], 1ULL);
}

} // namespace

extern "C" {

int szSpatialSetAxis(szContext *ctx, int axis) {
  if (!ctx || (axis != 0 && axis != 1)) return SZ_ERR_BAD_ARG;
  if (!ctx->step) return sz_fail(ctx, SZ_ERR_STATE, "szSpatialSetAxis: no floes uploaded");
  ctx->step->sp_axis = axis;
  return SZ_OK;
}

int szSpatialInit(szContext *ctx, int rank, int world, const double *edges, double halo) {
  if (!ctx || !edges || world < 1 || rank < 0 || rank >= world) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S) return sz_fail(ctx, SZ_ERR_STATE, "szSpatialInit: no floes uploaded");
  for (int r = 0; r < world; ++r) if (!(edges[r] < edges[r + 1])) return sz_fail(ctx, SZ_ERR_BAD_ARG, "szSpatialInit: strip edges must ascend");
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  const double reach = 2 * S->rmax_max_all;
  if (!(halo > 0)) halo = reach;
  if (halo < reach) return sz_fail(ctx, SZ_ERR_BAD_ARG, "szSpatialInit: the halo must be at least the reach of a pair, 2 max rmax = %g m, wide", reach);
  if (world > 1)
    for (int r = 0; r < world; ++r)
      if (edges[r + 1] - edges[r] < halo)
        return sz_fail(ctx, SZ_ERR_BAD_ARG, "szSpatialInit: strip %d is %g m wide, narrower than the halo (%g m): use fewer ranks for this field",
                       r, edges[r + 1] - edges[r], halo);
  S->sp_rank = rank; S->sp_world = world; S->sp_halo = halo;
  S->sp_edges.assign(edges, edges + world + 1);
  const int n = S->n;
  int rc;
  if ((rc = sz_reserve(ctx, S->sp_masks, 3 * (size_t)n + 64)) || (rc = sz_reserve(ctx, S->sp_edges_dev, ((size_t)world + 1) * 8)) ||
      (rc = sz_reserve(ctx, S->sp_counts, 64))) return rc;
  // ownership of the uploaded array: the strip of each centroid (every rank was handed the same array)
  std::vector<unsigned char> m(2 * (size_t)n, 0);
  std::vector<double> x((size_t)n), xs((size_t)n);
  SZ_HIP(ctx, hipMemcpy(x.data(), S->sp_axis ? S->F.Yi : S->F.Xi, (size_t)n * 8, hipMemcpyDeviceToHost));
  SZ_HIP(ctx, hipMemcpy(xs.data(), S->F.Xi, (size_t)n * 8, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; ++i) {
    const bool nan = x[i] != x[i] || xs[i] != xs[i];
    const bool mine = nan ? rank == 0 : sp_strip_of(x[i], edges, world) == rank;   // (a sacked floe, NaN, stays with rank 0)
    m[i] = mine; m[(size_t)n + i] = mine;
  }
  SZ_HIP(ctx, hipMemcpy(S->sp_masks.p, m.data(), m.size(), hipMemcpyHostToDevice));
  SZ_HIP(ctx, hipMemcpy(S->sp_edges_dev.p, edges, ((size_t)world + 1) * 8, hipMemcpyHostToDevice));
  S->W.present = (const unsigned char *)S->sp_masks.p;
  S->W.owned = (const unsigned char *)S->sp_masks.p + n;
  if (!S->sp_hdr) SZ_HIP(ctx, hipHostMalloc((void **)&S->sp_hdr, 5 * SP_HDR * sizeof(long long), hipHostMallocDefault));
  memset(S->sp_hdr, 0, 5 * SP_HDR * sizeof(long long));
  return SZ_OK;
}

int64_t szSpatialMessageBytes(szContext *ctx, int64_t cap_rec, int64_t cap_dbl, int64_t cap_migr) {
  if (!ctx || !ctx->step) return 0;
  return (int64_t)sp_msg_bytes(cap_rec, cap_dbl, cap_migr, sp_bulk_row(ctx->step->n_stress, ctx->step->n_mc));
}

static SpatialGeom geom_of(const StepState *S, int periodic) {
  SpatialGeom g;
  g.rank = S->sp_rank; g.world = S->sp_world; g.periodic = periodic;
  g.lo = S->sp_edges[S->sp_rank]; g.hi = S->sp_edges[S->sp_rank + 1]; g.halo = S->sp_halo; g.rmax_max = S->rmax_max_all;
  g.has_left = S->sp_world > 1 && (periodic || S->sp_rank > 0);
  g.has_right = S->sp_world > 1 && (periodic || S->sp_rank < S->sp_world - 1);
  g.axis = S->sp_axis;
  return g;
}

int szSpatialPack(szContext *ctx, int periodic, void *d_left, void *d_right, int64_t cap_rec, int64_t cap_dbl, int64_t cap_migr,
                  szSpatialInfo *info) {
  if (!ctx || !d_left || !d_right || !info) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S || !S->W.owned) return sz_fail(ctx, SZ_ERR_STATE, "szSpatialPack: szSpatialInit first");
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t row = sp_bulk_row(S->n_stress, S->n_mc);
  SpatialMsg l = sp_msg_view((char *)d_left, cap_rec, cap_dbl, cap_migr, row), r = sp_msg_view((char *)d_right, cap_rec, cap_dbl, cap_migr, row);
  unsigned char *present = (unsigned char *)S->sp_masks.p, *owned = present + S->n;
  const SpatialGeom g = geom_of(S, periodic);
  // (ownership as it is now: a message that overflows gives it back, so that Pack can be called again with more room)
  SZ_HIP(ctx, hipMemcpyAsync(owned + S->n, owned, (size_t)S->n, hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(k_sp_header_clear, dim3(1), dim3(64), 0, st, l, r, (unsigned long long *)nullptr, (long long *)nullptr);
  hipLaunchKernelGGL(k_sp_classify, grid_for(S->n, 256), dim3(256), 0, st, S->F, S->W, g, (const double *)S->sp_edges_dev.p, present, owned, l, r);
  long long *hdev = nullptr;
  SZ_HIP(ctx, hipHostGetDevicePointer((void **)&hdev, S->sp_hdr, 0));
  hipLaunchKernelGGL(k_sp_seal, dim3(1), dim3(64), 0, st, l, r, hdev, (unsigned long long *)nullptr, (long long *)nullptr);
  SZ_HIP(ctx, hipGetLastError());
  SZ_HIP(ctx, hipStreamSynchronize(st));
  const long long *h = S->sp_hdr;
  const bool overflow = h[3] != SP_OK || h[SP_HDR + 3] != SP_OK;
  if (overflow) SZ_HIP(ctx, hipMemcpyAsync(owned, owned + S->n, (size_t)S->n, hipMemcpyDeviceToDevice, st));   // nothing was handed over
  if (!overflow && h[2] > 0) hipLaunchKernelGGL(k_sp_bulk, dim3((unsigned)h[0]), dim3(256), 0, st, S->F, S->W, l, 1);
  if (!overflow && h[SP_HDR + 2] > 0) hipLaunchKernelGGL(k_sp_bulk, dim3((unsigned)h[SP_HDR]), dim3(256), 0, st, S->F, S->W, r, 1);
  SZ_HIP(ctx, hipGetLastError());
  if (ctx->stream == ctx->own_stream) SZ_HIP(ctx, hipStreamSynchronize(st));   // the exchange runs on another stream
  memset(info, 0, sizeof(*info));
  info->left_records = h[0]; info->left_doubles = h[1]; info->left_migrants = h[2];
  info->right_records = h[SP_HDR]; info->right_doubles = h[SP_HDR + 1]; info->right_migrants = h[SP_HDR + 2];
  info->need_records = h[4] > h[SP_HDR + 4] ? h[4] : h[SP_HDR + 4];
  info->need_doubles = h[5] > h[SP_HDR + 5] ? h[5] : h[SP_HDR + 5];
  info->need_migrants = h[6] > h[SP_HDR + 6] ? h[6] : h[SP_HDR + 6];
  info->overflow = overflow ? 1 : 0;
  info->has_left = g.has_left; info->has_right = g.has_right;
  return SZ_OK;
}

static int unpack_impl(szContext *ctx, const void *d_left, const void *d_right, int64_t cap_rec, int64_t cap_dbl, int64_t cap_migr,
                       const int64_t *known_left, const int64_t *known_right, szSpatialInfo *info) {
  if (!ctx) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S || !S->W.owned) return sz_fail(ctx, SZ_ERR_STATE, "szSpatialUnpack: szSpatialInit first");
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t row = sp_bulk_row(S->n_stress, S->n_mc);
  unsigned char *present = (unsigned char *)S->sp_masks.p, *owned = present + S->n;
  const void *bufs[2] = {d_left, d_right};
  long long seen[2][SP_HDR] = {{0}, {0}};
  SZ_HIP(ctx, hipMemsetAsync(&S->W.counters[C_VETO], 0, sizeof(unsigned long long), st));   // (the caller has looked at the headers: whatever a queued unpack vetoed is settled)
  S->sp_unpack_queued = false;
  for (int side = 0; side < 2; ++side) {
    if (!bufs[side]) continue;
    SpatialMsg m = sp_msg_view((char *)bufs[side], cap_rec, cap_dbl, cap_migr, row);
    const int64_t *known = side ? known_right : known_left;
    if (known) { for (int k = 0; k < 3; ++k) seen[side][k] = known[k]; seen[side][3] = SP_OK; }   // (the caller has read the header already)
    else {
      SZ_HIP(ctx, hipMemcpyAsync(seen[side], m.hdr, sizeof(seen[side]), hipMemcpyDeviceToHost, st));
      SZ_HIP(ctx, hipStreamSynchronize(st));
    }
    if (seen[side][3] != SP_OK) continue;            // the sender could not fit its floes: the caller repeats this edge, larger
    const long long nrec = seen[side][0];
    if (nrec <= 0) continue;
    hipLaunchKernelGGL(k_sp_take, grid_for(nrec, 256), dim3(256), 0, st, S->F, S->W, m, present, owned);
    if (seen[side][2] > 0) hipLaunchKernelGGL(k_sp_bulk, dim3((unsigned)nrec), dim3(256), 0, st, S->F, S->W, m, 0);
  }
  SZ_HIP(ctx, hipGetLastError());
  if (info) {
    memset(info, 0, sizeof(*info));
    info->left_records = seen[0][0]; info->left_doubles = seen[0][1]; info->left_migrants = seen[0][2];
    info->right_records = seen[1][0]; info->right_doubles = seen[1][1]; info->right_migrants = seen[1][2];
    for (int side = 0; side < 2; ++side) {
      if (seen[side][4] > info->need_records) info->need_records = seen[side][4];
      if (seen[side][5] > info->need_doubles) info->need_doubles = seen[side][5];
      if (seen[side][6] > info->need_migrants) info->need_migrants = seen[side][6];
      if (seen[side][3] != SP_OK) info->overflow = 1;
    }
    SZ_HIP(ctx, hipMemsetAsync(S->sp_counts.p, 0, 16, st));
    hipLaunchKernelGGL(k_sp_count, grid_for(S->n, 256), dim3(256), 0, st, S->F, (const unsigned char *)present, (const unsigned char *)owned, (unsigned long long *)S->sp_counts.p);
    unsigned long long c[2];
    SZ_HIP(ctx, hipMemcpyAsync(c, S->sp_counts.p, 16, hipMemcpyDeviceToHost, st));
    SZ_HIP(ctx, hipStreamSynchronize(st));
    info->n_owned = (int64_t)c[0]; info->n_present = (int64_t)c[1];
  }
  return SZ_OK;
}

int szSpatialUnpack(szContext *ctx, const void *d_left, const void *d_right, int64_t cap_rec, int64_t cap_dbl, int64_t cap_migr,
                    szSpatialInfo *info) {
  return unpack_impl(ctx, d_left, d_right, cap_rec, cap_dbl, cap_migr, nullptr, nullptr, info);
}
int szSpatialUnpackKnown(szContext *ctx, const void *d_left, const void *d_right, int64_t cap_rec, int64_t cap_dbl, int64_t cap_migr,
                         const int64_t *left_counts, const int64_t *right_counts, szSpatialInfo *info) {
  return unpack_impl(ctx, d_left, d_right, cap_rec, cap_dbl, cap_migr, left_counts, right_counts, info);
}

// ---- the exchange without host round trips of its own (one per step: szStep's) ---------------------------------------
int szSpatialPackQueued(szContext *ctx, int periodic, void *d_left, void *d_right, int64_t cap_rec, int64_t cap_dbl, int64_t cap_migr,
                        const int64_t *plan_left, const int64_t *plan_right) {
  if (!ctx || !d_left || !d_right) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S || !S->W.owned) return sz_fail(ctx, SZ_ERR_STATE, "szSpatialPackQueued: szSpatialInit first");
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t row = sp_bulk_row(S->n_stress, S->n_mc);
  SpatialMsg l = sp_msg_view((char *)d_left, cap_rec, cap_dbl, cap_migr, row), r = sp_msg_view((char *)d_right, cap_rec, cap_dbl, cap_migr, row);
  unsigned char *present = (unsigned char *)S->sp_masks.p, *owned = present + S->n;
  const SpatialGeom g = geom_of(S, periodic);
  long long *hdev = nullptr;
  SZ_HIP(ctx, hipHostGetDevicePointer((void **)&hdev, S->sp_hdr, 0));
  SZ_HIP(ctx, hipMemcpyAsync(owned + S->n, owned, (size_t)S->n, hipMemcpyDeviceToDevice, st));   // (as szSpatialPack: given back by szSpatialPackFinish on overflow)
  hipLaunchKernelGGL(k_sp_header_clear, dim3(1), dim3(64), 0, st, l, r, (unsigned long long *)&S->W.counters[C_VETO], hdev + 4 * SP_HDR);
  hipLaunchKernelGGL(k_sp_classify, grid_for(S->n, 256), dim3(256), 0, st, S->F, S->W, g, (const double *)S->sp_edges_dev.p, present, owned, l, r);
  hipLaunchKernelGGL(k_sp_seal, dim3(1), dim3(64), 0, st, l, r, hdev, (unsigned long long *)&S->W.counters[C_VETO], hdev + 4 * SP_HDR);
  // the rows of the floes that change strip, when room for some is agreed to travel in the first round (a workgroup per
  // planned record: the count is the device's; a message beyond its plan is finished by szSpatialPackFinish)
  if (plan_left && plan_left[2] > 0 && plan_left[0] > 0) hipLaunchKernelGGL(k_sp_bulk, dim3((unsigned)plan_left[0]), dim3(256), 0, st, S->F, S->W, l, 1);
  if (plan_right && plan_right[2] > 0 && plan_right[0] > 0) hipLaunchKernelGGL(k_sp_bulk, dim3((unsigned)plan_right[0]), dim3(256), 0, st, S->F, S->W, r, 1);
  SZ_HIP(ctx, hipGetLastError());
  S->sp_pack_periodic = periodic;
  S->sp_unpack_queued = true;     // (the veto is live from here: the next szStep honours it)
  return SZ_OK;
}

int szSpatialPackFinish(szContext *ctx, void *d_left, void *d_right, int64_t cap_rec, int64_t cap_dbl, int64_t cap_migr, szSpatialInfo *info) {
  if (!ctx || !d_left || !d_right || !info) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S || !S->W.owned) return sz_fail(ctx, SZ_ERR_STATE, "szSpatialPackFinish: szSpatialInit first");
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  SZ_HIP(ctx, hipStreamSynchronize(st));           // (normally complete long ago: szStep has waited for the stream since)
  const size_t row = sp_bulk_row(S->n_stress, S->n_mc);
  SpatialMsg l = sp_msg_view((char *)d_left, cap_rec, cap_dbl, cap_migr, row), r = sp_msg_view((char *)d_right, cap_rec, cap_dbl, cap_migr, row);
  unsigned char *owned = (unsigned char *)S->sp_masks.p + S->n;
  const SpatialGeom g = geom_of(S, S->sp_pack_periodic);
  const long long *h = S->sp_hdr;
  const bool overflow = h[3] != SP_OK || h[SP_HDR + 3] != SP_OK;
  if (overflow) SZ_HIP(ctx, hipMemcpyAsync(owned, owned + S->n, (size_t)S->n, hipMemcpyDeviceToDevice, st));   // nothing was handed over
  // the rows of the floes that changed strip (their owner's step leaves them alone: kf_integrate asks ownership first)
  if (!overflow && h[2] > 0) hipLaunchKernelGGL(k_sp_bulk, dim3((unsigned)h[0]), dim3(256), 0, st, S->F, S->W, l, 1);
  if (!overflow && h[SP_HDR + 2] > 0) hipLaunchKernelGGL(k_sp_bulk, dim3((unsigned)h[SP_HDR]), dim3(256), 0, st, S->F, S->W, r, 1);
  SZ_HIP(ctx, hipGetLastError());
  if (ctx->stream == ctx->own_stream) SZ_HIP(ctx, hipStreamSynchronize(st));
  memset(info, 0, sizeof(*info));
  info->left_records = h[0]; info->left_doubles = h[1]; info->left_migrants = h[2];
  info->right_records = h[SP_HDR]; info->right_doubles = h[SP_HDR + 1]; info->right_migrants = h[SP_HDR + 2];
  info->need_records = h[4] > h[SP_HDR + 4] ? h[4] : h[SP_HDR + 4];
  info->need_doubles = h[5] > h[SP_HDR + 5] ? h[5] : h[SP_HDR + 5];
  info->need_migrants = h[6] > h[SP_HDR + 6] ? h[6] : h[SP_HDR + 6];
  info->overflow = overflow ? 1 : 0;
  info->has_left = g.has_left; info->has_right = g.has_right;
  return SZ_OK;
}

int szSpatialUnpackQueued(szContext *ctx, const void *d_left, const void *d_right, int64_t cap_rec, int64_t cap_dbl, int64_t cap_migr,
                          const int64_t *plan_left, const int64_t *plan_right) {
  if (!ctx) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S || !S->W.owned) return sz_fail(ctx, SZ_ERR_STATE, "szSpatialUnpackQueued: szSpatialInit first");
  if ((d_left && !plan_left) || (d_right && !plan_right)) return SZ_ERR_BAD_ARG;
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t row = sp_bulk_row(S->n_stress, S->n_mc);
  unsigned char *present = (unsigned char *)S->sp_masks.p, *owned = present + S->n;
  long long *hdev = nullptr;
  SZ_HIP(ctx, hipHostGetDevicePointer((void **)&hdev, S->sp_hdr, 0));
  const void *bufs[2] = {d_left, d_right};
  for (int side = 0; side < 2; ++side) {
    if (!bufs[side]) continue;
    const int64_t *plan = side ? plan_right : plan_left;
    SpatialMsg m = sp_msg_view((char *)bufs[side], cap_rec, cap_dbl, cap_migr, row);
    const long long nrec = plan[0] > 0 ? plan[0] : 1;
    hipLaunchKernelGGL(k_sp_take_queued, grid_for(nrec, 256), dim3(256), 0, st, S->F, S->W, m, present, owned, (long long)plan[0], (long long)plan[1],
                       (long long)plan[2], (unsigned long long *)&S->W.counters[C_VETO], hdev + (2 + side) * SP_HDR, hdev + 4 * SP_HDR);
    if (plan[2] > 0 && plan[0] > 0)
      hipLaunchKernelGGL(k_sp_bulk_take_queued, dim3((unsigned)plan[0]), dim3(256), 0, st, S->F, S->W, m, (long long)plan[0], (long long)plan[1], (long long)plan[2]);
  }
  SZ_HIP(ctx, hipGetLastError());
  S->sp_unpack_queued = true;
  return SZ_OK;
}

// after the szStep that followed szSpatialPackQueued / szSpatialUnpackQueued (it has waited for the stream): the four
// headers -- my messages to the left and to the right, the messages from the left and from the right -- 8 int64 each
// (records, doubles, migrants, status, records / doubles / migrants needed, 0), and whether the unpack vetoed the step
int szSpatialHeaders(szContext *ctx, int64_t *out32, int32_t *vetoed) {
  if (!ctx || !out32) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S || !S->sp_hdr) return sz_fail(ctx, SZ_ERR_STATE, "szSpatialHeaders: szSpatialInit first");
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  SZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < 4 * SP_HDR; ++k) out32[k] = S->sp_hdr[k];
  if (vetoed) *vetoed = S->sp_hdr[4 * SP_HDR] ? 1 : 0;
  return SZ_OK;
}

int szSpatialOwned(szContext *ctx, uint8_t *owned) {
  if (!ctx || !owned) return SZ_ERR_BAD_ARG;
  StepState *S = ctx->step;
  if (!S || !S->W.owned) return sz_fail(ctx, SZ_ERR_STATE, "szSpatialOwned: szSpatialInit first");
  SZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  SZ_HIP(ctx, hipMemcpy(owned, (const unsigned char *)S->sp_masks.p + S->n, (size_t)S->n, hipMemcpyDeviceToHost));
  return SZ_OK;
}

} // extern "C"
