// context.hpp -- the szContext behind the C ABI (include/subzero_hip.h): one HIP device, one stream,
// a pair of timing events and growable device buffers owned by the context.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../../include/subzero_hip.h"

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
};

// pinned host memory (hipHostMalloc), grow-only: what an entry's small arrays travel through, packed into ONE transfer
// each way (a hipMemcpyAsync from or to pageable memory is staged by the runtime and costs tens of microseconds each)
struct HostStage {
  char *p = nullptr;
  size_t cap = 0;
};
// layout of several arrays in one buffer, 16-byte aligned
struct Packer {
  size_t at = 0;
  size_t add(size_t bytes) { const size_t o = at; at = (at + bytes + 15) & ~(size_t)15; return o; }
};

struct szContext {
  int device = 0;
  hipStream_t stream = nullptr;      // the stream everything runs on: own_stream, or the caller's (szSetStream)
  hipStream_t own_stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  char err[512] = {0};
  std::vector<DevBuf *> owned;
  struct StepState *step = nullptr; // floe state for the B2 surface (step.hip)

  // clip path buffers
  DevBuf ws, ovx, ovy, oroff, ov_base, or_base, nrings, status, retry_list, counters;
  DevBuf clip_meta, clip_lists;      // front pass -> sweep pass (clip.hip)
  HostStage h_in, h_out;             // staging of the post-step entries (corners.hip ...)
  HostStage h_cnt;                   // the clip batch's counters on their way in and out (clip.hip: three slots of 16 words)
  DevBuf d_in, d_out;
  bool clip_lds_set = false;         // the clip kernels' dynamic-LDS limit has been raised on this context's device
  // Three side streams for everything in the library that runs kernels side by side (the sweep passes of the clip batch,
  // the wall pairs of a step): with the context's own stream that makes four, the number of hardware queues a process
  // gets by default (GPU_MAX_HW_QUEUES) -- a fifth stream shares a queue with another and silently runs behind it.
  hipStream_t side[3] = {nullptr, nullptr, nullptr};
  hipEvent_t side_fork = nullptr, side_join[3] = {nullptr, nullptr, nullptr};
  DevBuf in_ax, in_ay, in_aoff, in_bx, in_by, in_boff;
  DevBuf in_fax, in_fay, in_fbx, in_fby; // double inputs of szPolyclipBatch
};

inline int sz_fail(szContext *ctx, int code, const char *fmt, ...) {
  if (ctx) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

#define SZ_HIP(ctx, call)                                                                        \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      return sz_fail(ctx, SZ_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),     \
                     __FILE__, __LINE__);                                                        \
  } while (0)

// grow-only device buffer; contents are NOT preserved on growth
inline int sz_reserve(szContext *ctx, DevBuf &b, size_t bytes) {
  if (bytes <= b.cap && b.p) return SZ_OK;
  if (b.p) {
    SZ_HIP(ctx, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  size_t want = bytes + bytes / 4 + 256;
  SZ_HIP(ctx, hipMalloc(&b.p, want));
  b.cap = want;
  return SZ_OK;
}
inline int sz_stage_reserve(szContext *ctx, HostStage &h, size_t bytes) {
  if (bytes <= h.cap && h.p) return SZ_OK;
  if (h.p) { SZ_HIP(ctx, hipHostFree(h.p)); h.p = nullptr; h.cap = 0; }
  const size_t want = bytes + bytes / 4 + 4096;
  SZ_HIP(ctx, hipHostMalloc((void **)&h.p, want, hipHostMallocDefault));
  h.cap = want;
  return SZ_OK;
}
inline int sz_side_streams(szContext *ctx) {
  if (ctx->side_fork) return SZ_OK;
  SZ_HIP(ctx, hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming));
  for (int k = 0; k < 3; ++k) {
    SZ_HIP(ctx, hipStreamCreateWithFlags(&ctx->side[k], hipStreamNonBlocking));
    SZ_HIP(ctx, hipEventCreateWithFlags(&ctx->side_join[k], hipEventDisableTiming));
  }
  return SZ_OK;
}
// everything the context has in flight, side streams included (an error return between a fork and its join leaves work
// on a side stream that the main stream never waited for): before buffers or events are freed
inline hipError_t sz_sync_all(szContext *ctx) {
  hipError_t e = hipStreamSynchronize(ctx->stream);
  for (int k = 0; k < 3; ++k) if (ctx->side[k]) { const hipError_t e2 = hipStreamSynchronize(ctx->side[k]); if (e == hipSuccess) e = e2; }
  return e;
}
inline void sz_release(DevBuf &b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}
