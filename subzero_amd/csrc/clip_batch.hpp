// clip_batch.hpp -- the batched clip of clip.hip as other translation units of the library drive it (fracture.hip):
// arguments of one launch of the clip kernel and the two host steps around it.  The results stay on the device
// (per pair: ov_base / or_base / nrings / status; rings in ovx / ovy / oroff) until the caller gathers them.
#pragma once
#include "context.hpp"
#include "vatti.hpp"

namespace szb {
using szv::i64;

struct ClipArgs {
  int npairs;
  const int *list; // null: all pairs, closed-form slots; else: pairs to re-run, bump-allocated slots
  int nlist;
  const i64 *ax, *ay; const int *aoff;
  const i64 *bx, *by; const int *boff;
  const int *aring, *bring;     // when non-null: aoff / boff give RING ranges per pair and these the vertex range of each ring
  const double *fax, *fay, *fbx, *fby; // when non-null the inputs are doubles (polyclip form)
  int method, fill_s, fill_c;
  int mult;
  int lanes;                    // pairs per 64-lane wave (the rest of the lanes idle: fewer divergent sweeps per wave)
  int lds_per_lane;             // bytes of LDS per lane for the sweep's records (0: the HBM workspace only)
  int no_team;                  // k_clip_front does not try the team form of the clip on short rings (SUBZERO_CLIP_NO_TEAM: comparisons)
  char *ws; i64 ws_cap;         // workspace arena
  i64 *ovx, *ovy; i64 ov_cap;   // result vertices
  int *oroff; i64 or_cap;       // result ring offsets (local to the pair's vertex slot)
  i64 *ov_base, *or_base;       // per pair: where its slots start
  int *nrings, *status;
  int *retry; int retry_cap;    // pairs to re-run with a larger workspace
  int *meta;                    // front pass -> sweep pass, per pair: trimmed sizes (bit 30 of the first: rings at the workspace tail)
  int *class_list;              // 5 lists of npairs: the pairs of each slice class
  unsigned long long *counters; // [0] retry count, [1] ws bump, [2] ov bump, [3] or bump,
                                // [4] rings out, [5] verts out, [6] failed, [7] arena exhausted, [8] outgrew the LDS slice, [9..14] class list lengths
};

// sizes and (re)allocates the context's clip buffers for a batch of npairs pairs with `total` input vertices; fills the
// common fields of a
int prepare(szContext *ctx, int npairs, i64 total, ClipArgs &a);
// the main pass and the retry tiers; on return the per-pair arrays on the device are final
int run(szContext *ctx, ClipArgs &a, szClipStats *stats);

} // namespace szb
