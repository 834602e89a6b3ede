// capi.hip -- context management and raw device-memory helpers of include/subzero_hip.h.
#include "context.hpp"

void sz_step_destroy(szContext *ctx); // step.hip

extern "C" {

int szAbiVersion(void) { return SZ_ABI_VERSION; }

int szCreate(szContext **out, int device) {
  if (!out) return SZ_ERR_BAD_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return SZ_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SZ_ERR_NO_DEVICE;
  // the code objects in this library are gfx950 only
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return SZ_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return SZ_ERR_NO_DEVICE;
  szContext *ctx = new szContext();
  ctx->device = device;
  if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
    delete ctx;
    return SZ_ERR_HIP;
  }
  ctx->stream = ctx->own_stream;
  *out = ctx;
  return SZ_OK;
}

int szDestroy(szContext *ctx) {
  if (!ctx) return SZ_ERR_BAD_ARG;
  (void)hipSetDevice(ctx->device);
  (void)sz_sync_all(ctx);
  sz_step_destroy(ctx);
  DevBuf *bufs[] = {&ctx->ws, &ctx->ovx, &ctx->ovy, &ctx->oroff, &ctx->ov_base, &ctx->or_base,
                    &ctx->nrings, &ctx->status, &ctx->retry_list, &ctx->counters, &ctx->clip_meta, &ctx->clip_lists,
                    &ctx->in_ax, &ctx->in_ay, &ctx->in_aoff, &ctx->in_bx, &ctx->in_by, &ctx->in_boff,
                    &ctx->in_fax, &ctx->in_fay, &ctx->in_fbx, &ctx->in_fby};
  for (DevBuf *b : bufs) sz_release(*b);
  sz_release(ctx->d_in); sz_release(ctx->d_out);
  if (ctx->h_in.p) (void)hipHostFree(ctx->h_in.p);
  if (ctx->h_out.p) (void)hipHostFree(ctx->h_out.p);
  if (ctx->h_cnt.p) (void)hipHostFree(ctx->h_cnt.p);
  for (int k = 0; k < 3; ++k) {
    if (ctx->side[k]) (void)hipStreamDestroy(ctx->side[k]);
    if (ctx->side_join[k]) (void)hipEventDestroy(ctx->side_join[k]);
  }
  if (ctx->side_fork) (void)hipEventDestroy(ctx->side_fork);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return SZ_OK;
}

int szSetStream(szContext *ctx, void *hip_stream) {
  if (!ctx) return SZ_ERR_BAD_ARG;
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  SZ_HIP(ctx, hipStreamSynchronize(ctx->stream));   // nothing of ours is left in flight on the old stream
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return SZ_OK;
}

const char *szLastError(const szContext *ctx) { return ctx ? ctx->err : "null context"; }

int szDeviceAlloc(szContext *ctx, void **dptr, int64_t bytes) {
  if (!ctx || !dptr || bytes < 0) return SZ_ERR_BAD_ARG;
  SZ_HIP(ctx, hipSetDevice(ctx->device));
  SZ_HIP(ctx, hipMalloc(dptr, (size_t)(bytes > 0 ? bytes : 16)));
  return SZ_OK;
}
int szDeviceFree(szContext *ctx, void *dptr) {
  if (!ctx) return SZ_ERR_BAD_ARG;
  if (dptr) SZ_HIP(ctx, hipFree(dptr));
  return SZ_OK;
}
int szCopyToDevice(szContext *ctx, void *dst, const void *src, int64_t bytes) {
  if (!ctx || bytes < 0) return SZ_ERR_BAD_ARG;
  if (bytes) SZ_HIP(ctx, hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice));
  return SZ_OK;
}
int szCopyToHost(szContext *ctx, void *dst, const void *src, int64_t bytes) {
  if (!ctx || bytes < 0) return SZ_ERR_BAD_ARG;
  SZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (bytes) SZ_HIP(ctx, hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost));
  return SZ_OK;
}
int szSynchronize(szContext *ctx) {
  if (!ctx) return SZ_ERR_BAD_ARG;
  SZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SZ_OK;
}

} // extern "C"
