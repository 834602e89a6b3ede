/* subzero_hip.h -- C ABI of libsubzero_hip.so, the MI355X (gfx950) implementation of SubZero's
 * per-timestep floe-interaction hot path.
 *
 * Two reference surfaces are replaced:
 *
 *  B1  the `clipper` mex                 /root/reference/private/mexclipper.cpp:83-305
 *      called only from                  /root/reference/polyclip.m:73
 *      -> szClipBatch / szPolyclipBatch / szOrientation
 *
 *  B2  the Floe struct-array step        /root/reference/floe_interactions_all.m:1-285
 *      called from                       /root/reference/Subzero.m:133,301
 *      -> szFloesUpload / szStep / szFloesDownload / szInteractionsDownload
 *
 * Conventions: plain C, plain pointers and sizes, caller owns every host buffer, the context owns all
 * device memory.  Every function returns an int status (SZ_OK = 0) and never throws, exits or calls
 * into MATLAB.  One context per process and GPU; a context is not thread-safe (the mex it replaces
 * is called on MATLAB's interpreter thread only).
 *
 * There is no CPU fallback: without a usable gfx950 device szCreate fails with SZ_ERR_NO_DEVICE.
 */
#ifndef SUBZERO_HIP_H
#define SUBZERO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SZ_ABI_VERSION 1

/* library status codes */
enum {
  SZ_OK = 0,
  SZ_ERR_NO_DEVICE = -1,
  SZ_ERR_BAD_ARG = -2,
  SZ_ERR_HIP = -3,       /* a HIP runtime call failed; see szLastError */
  SZ_ERR_CAPACITY = -4,  /* caller-provided output buffer too small */
  SZ_ERR_WORKSPACE = -5, /* a device-side list outgrew every workspace tier (never silent) */
  SZ_ERR_STATE = -6,     /* call out of order (e.g. szStep before szFloesUpload) */
  SZ_ERR_PEER = -7,      /* sharded step: another rank reported an error in its segment header */
  SZ_SHARD_RESIZE = 1,   /* sharded step, not an error: a segment has more contact rows than rows_max; exchange again */
  SZ_STEP_NEEDS_MC = 2   /* szStep, not an error: floes whose ocean/wind forcing is due have an empty Monte-Carlo mask
                            (calc_trajectory.m:100-110 draws new points with rand); NOTHING of the floe array was
                            changed; fetch them with szMcEmptyDownload, regenerate on the host, szFloesUpdate, step again */
  , SZ_STEP_VETOED = 3   /* szStep after szSpatialUnpackQueued, not an error: a neighbour's message did not fit what was agreed
                            to travel, so its records were not taken; NOTHING of the floe array was changed; exchange the rest
                            (szSpatialHeaders tells what), szSpatialUnpackKnown, step again */
};

/* per-clip status, one per pair (what the mex would have done for that call) */
enum {
  SZ_CLIP_OK = 0,
  SZ_CLIP_ERROR = 1,        /* mexErrMsgTxt("Clipper Error."), mexclipper.cpp:304 */
  SZ_CLIP_BAD_ARG = 2,      /* bad method / fill type, mexclipper.cpp:229,259,282 */
  SZ_CLIP_OUT_CAPACITY = 3, /* internal: result larger than its slot (retried, never returned) */
  SZ_CLIP_RANGE = 4,        /* coordinate outside Clipper's range, clipper.cpp:883-896 */
  SZ_CLIP_WORKSPACE = 5     /* internal: retried with a larger workspace, never returned */
};

/* boolean operation, numbered as the mex's third argument (mexclipper.cpp:215-230; polyclip.m:52-58) */
enum { SZ_CLIP_DIFFERENCE = 0, SZ_CLIP_INTERSECTION = 1, SZ_CLIP_XOR = 2, SZ_CLIP_UNION = 3 };
/* fill rule, numbered as the mex's 4th/5th argument (mexclipper.cpp:243-287); default even-odd */
enum { SZ_FILL_EVENODD = 0, SZ_FILL_NONZERO = 1, SZ_FILL_POSITIVE = 2, SZ_FILL_NEGATIVE = 3 };

typedef struct szContext szContext;

int szAbiVersion(void);
/* device = HIP device ordinal.  Fails with SZ_ERR_NO_DEVICE when no gfx950 GPU is usable. */
int szCreate(szContext **ctx, int device);
int szDestroy(szContext *ctx);
/* Run everything on `hip_stream` (a hipStream_t of the context's device, owned by the caller, e.g. the stream a
 * framework orders its collectives on) instead of the context's own stream; NULL switches back. */
int szSetStream(szContext *ctx, void *hip_stream);
/* text of the last failure on this context (static storage inside the context) */
const char *szLastError(const szContext *ctx);

/* ---- B1: boolean clipping, batched ---------------------------------------------------------
 * npairs independent calls of  clipper(Subj, Clip, method, subj_fill, clip_fill)  with single-ring
 * Subj and Clip (polyclip.m:72 asserts exactly that).  Rings are given as CSR over one vertex pool:
 * pair p's subject is (ax,ay)[aoff[p] .. aoff[p+1]), its clip (bx,by)[boff[p] .. boff[p+1]).
 * int64 coordinates as produced by polyclip.m:66.
 *
 * Results, in the order the mex returns them (ring order, first vertex, direction all identical):
 *   pair p owns rings   pair_ring_off[p] .. pair_ring_off[p+1]
 *   ring r owns vertices ring_off[r] .. ring_off[r+1]   of (outx,outy)
 * status[p] is SZ_CLIP_OK / SZ_CLIP_ERROR / SZ_CLIP_BAD_ARG / SZ_CLIP_RANGE.
 * Returns SZ_ERR_CAPACITY if out_cap / ring_cap are too small (nothing partial is promised).
 * All pointers are HOST pointers. */
int szClipBatch(szContext *ctx, int npairs,
                const int64_t *ax, const int64_t *ay, const int32_t *aoff,
                const int64_t *bx, const int64_t *by, const int32_t *boff,
                int method, int subj_fill, int clip_fill,
                int64_t *outx, int64_t *outy, int64_t out_cap,
                int32_t *ring_off, int64_t ring_cap,
                int32_t *pair_ring_off, int32_t *status);

/* The same with several contours per operand, as the mex accepts them (a struct ARRAY Subj / Clip: every element
 * is one contour, mexclipper.cpp:44-62, all handed to AddPaths :291-297).  Pair p's subject is the rings
 * a_pair_off[p] .. a_pair_off[p+1], ring r the vertices a_ring_off[r] .. a_ring_off[r+1] of (ax, ay); the clip likewise.
 * (polyclip.m:72 never passes more than one contour; the form exists because the reference's gateway does.) */
int szClipPathsBatch(szContext *ctx, int npairs,
                     const int64_t *ax, const int64_t *ay, const int32_t *a_ring_off, const int32_t *a_pair_off,
                     const int64_t *bx, const int64_t *by, const int32_t *b_ring_off, const int32_t *b_pair_off,
                     int method, int subj_fill, int clip_fill,
                     int64_t *outx, int64_t *outy, int64_t out_cap,
                     int32_t *ring_off, int64_t ring_cap,
                     int32_t *pair_ring_off, int32_t *status);

/* polyclip.m:63-80 in one call: doubles in (metres), x2^32 fixed-point conversion with MATLAB int64()
 * semantics (round half away from zero, saturating), clip, back to doubles (/2^32). Same layout. */
int szPolyclipBatch(szContext *ctx, int npairs,
                    const double *ax, const double *ay, const int32_t *aoff,
                    const double *bx, const double *by, const int32_t *boff,
                    int method,
                    double *outx, double *outy, int64_t out_cap,
                    int32_t *ring_off, int64_t ring_cap,
                    int32_t *pair_ring_off, int32_t *status);

/* Device-resident form used by the step path and by bench.py: inputs already in HBM (device
 * pointers), results stay in HBM in per-pair slots; only counters come back.  See DESIGN.md. */
typedef struct {
  double kernel_ms;      /* HIP-event time of the clip kernel(s) on the context's stream */
  int64_t pairs;         /* pairs processed */
  int64_t input_vertices;/* sum of na+nb */
  int64_t rings_out;     /* rings produced */
  int64_t vertices_out;  /* vertices produced */
  int64_t retried;       /* pairs that needed a larger workspace tier */
  int64_t failed;        /* pairs with status != SZ_CLIP_OK */
} szClipStats;

int szClipBatchDevice(szContext *ctx, int npairs,
                      const int64_t *d_ax, const int64_t *d_ay, const int32_t *d_aoff,
                      const int64_t *d_bx, const int64_t *d_by, const int32_t *d_boff,
                      int total_a, int total_b, int method, szClipStats *stats);


/* ---- B2: the Floe struct-array step ---------------------------------------------------------
 * Replaces floe_interactions_all.m:16-285 (ghost floes, broad-phase, floe_interactions per candidate,
 * contact mirror, torques and sums, periodic wrap, calc_trajectory).  Lines 288-507 (ridging, rafting,
 * fuse / kill bookkeeping) stay with the caller and consume `kill` / `transfer`.
 *
 * szFloeState is the struct-of-arrays image of the Floe struct array; field names are those of
 * Initialize_Model/initialize_floe_values.m:12-52.  All pointers are HOST pointers owned by the caller.
 * Rings are CSR: floe i owns vertices vert_off[i]..vert_off[i+1] of c0 (unrotated) and c_alpha, closed
 * (first vertex repeated, initialize_floe_values.m:17).  2x2 matrices are MATLAB column-major
 * [xx, yx, xy, yy].  StressH is [floe][slot][4]; mcX/mcY/mcA are the Monte-Carlo fields X, Y, A as
 * [floe][point].  kill / transfer have capacity 4*n: they are indexed over the extended list (real floes
 * then ghost floes), exactly as the reference's arrays grow when a ghost row assigns into them. */
typedef struct szFloeState {
  int32_t n;        /* floes in the array (all alive ones the caller kept, floe_interactions_all.m:12-13) */
  int32_t nb;       /* Nb: leading topography floes */
  int32_t n_stress; /* slots of StressH (1000) */
  int32_t n_mc;     /* Monte-Carlo points per floe (1000) */
  double *Xi, *Yi, *alpha_i, *Ui, *Vi, *ksi_ice;
  double *h, *area, *mass, *inertia_moment, *rmax;
  double *dXi_p, *dYi_p, *dUi_p, *dVi_p, *dalpha_i_p, *dksi_ice_p;
  double *FxOA, *FyOA, *torqueOA;
  int32_t *alive;
  int32_t *vert_off;
  double *c0x, *c0y, *cax, *cay;
  double *StressH;
  int32_t *StressCount;
  double *Stress, *strain;
  double *mcX, *mcY;
  uint8_t *mcA;
  double *collision_force, *collision_torque, *OverlapArea;
  int32_t *kill, *transfer;
} szFloeState;

/* the arguments of floe_interactions_all.m:1 the accelerated part reads (+ the Modulus global, :7) */
typedef struct szStepParams {
  double dt, HFo, Modulus;
  double mu, nu;            /* floe_interactions.m:20-21 (0.2, 0.3 as shipped) */
  int32_t doInt, COLLISION, PERIODIC;
  int32_t flags;       /* SZ_STEP_* below; 0: none */
  double c2_boundary[10];   /* 2x5 column-major: x0,y0,x1,y1,... */
  double fb_h, fb_area, fb_Xi, fb_Yi, fb_Ui, fb_Vi, fb_ksi_ice; /* floebound (Subzero.m:67-69) */
  int32_t nxo, nyo;
  const double *Xo, *Yo;       /* ocean.Xo, ocean.Yo */
  const double *Uocn, *Vocn;   /* nyo x nxo column-major */
  const double *Uwind, *Vwind; /* winds.u, winds.v, same shape */
  double fCoriolis, turn_angle;
} szStepParams;

/* szStepParams.flags.  SZ_STEP_QUEUE_AHEAD: the caller will call szStep again with the same parameters -- but for doInt, which
 * it announces by SZ_STEP_NEXT_DOINT -- before it changes the floes (szFloesUpdate) or asks for the ghost list
 * (szGhostsDownload): the library then queues the first kernels of that next step behind this one's, and the device does
 * not idle while the host returns to its caller and comes back.  A promise not kept costs nothing but the benefit (the
 * next step finds its parameters changed and launches those kernels afresh; szGhostsDownload reports SZ_ERR_STATE). */
enum { SZ_STEP_QUEUE_AHEAD = 1, SZ_STEP_NEXT_DOINT = 2 };

typedef struct szStepStats {
  /* HIP-event stage times.  As szStep returns them they are those of step `times_step` -- normally THE STEP BEFORE this
   * one: the runtime sees an event's signal some microseconds after the step's own completion reaches the host, and
   * waiting for it would cost every step that much; the events are read one step late, while the host waits for the next
   * step anyway.  szStepTimes waits for the events of the last step and returns its times. */
  double step_ms;      /* the whole step on the context's stream */
  double broad_ms, narrow_ms, rows_ms, integrate_ms;   /* narrow_ms: the pair-interaction kernel (k_narrow_team) alone */
  int64_t n_ext;       /* real + ghost floes */
  int64_t n_pairs;     /* candidate pairs incl. floe-vs-wall candidates */
  int64_t n_rows;      /* contact rows over the extended list */
  int64_t pair_vertices; /* sum of ring sizes over the candidate pairs */
  int64_t retried;     /* pairs re-run at a larger scratch tier */
  int64_t pair_rows;   /* contact rows produced by the candidate pairs (before mirroring) */
  int64_t team_pairs;  /* pairs on the three lists of the pair-interaction kernel (k_narrow_team: hot + cold + wide); the other
                          candidates were answered by the classification (boxes apart) or handed to the general sweep */
  int64_t team_vertices; /* stored ring vertices (both rings) summed over those pairs: what that kernel's roofline is priced on */
  double pairs_ms;     /* HIP-event time between the broad phase and the pair-interaction kernel: pair records + classification */
  int64_t times_step;  /* the step (0-based count of completed steps of this array) the *_ms fields belong to; -1: none yet */
} szStepStats;

/* which fields szFloesDownload copies back */
enum {
  SZ_SYNC_DYNAMIC = 1, /* everything calc_trajectory / the step writes except the two bulk histories */
  SZ_SYNC_STRESSH = 2, /* StressH */
  SZ_SYNC_ALL = 3
};

/* field groups of szFloesUpdate / szFloesFetch */
enum {
  SZ_F_STATE = 1,   /* every per-floe scalar field, alive, StressCount, Stress, strain, collision_force/torque, OverlapArea */
  SZ_F_RING = 2,    /* c0 and c_alpha (same number of vertices as uploaded; a ring that changes length: szFloesResync) */
  SZ_F_STRESSH = 4, /* StressH */
  SZ_F_MC = 8,      /* the Monte-Carlo fields X, Y, A */
  SZ_F_ALL = 15
};

/* Copies the whole struct array to the GPU (replaces any previous one).  The state then lives on the
 * device across szStep calls (dirty-field download only when MATLAB needs it). */
int szFloesUpload(szContext *ctx, const szFloeState *host);
/* SURVEY.md 8(f)2 -- synchronisation after MATLAB-side edits without moving the 0.5 GB of histories again.
 *
 * szFloesUpdate / szFloesFetch move the field groups `what` (SZ_F_*) of the listed floes (floe numbers, 1-based, as in
 * interactions(:,1)) from / into `host`, the caller's image of the WHOLE array (same n and ring layout as the resident
 * one; only the rows of the listed floes are read / written; a null pointer skips that field).  One packed transfer
 * per group, whatever the number of floes.  Typical use: regenerated Monte-Carlo points (SZ_F_MC, after
 * SZ_STEP_NEEDS_MC), velocities or thicknesses edited by a MATLAB-side process (SZ_F_STATE), the stress history of
 * the floes a process is about to reshape (Fetch SZ_F_STRESSH: frac_corner.m:104,157 and Fuse_Floes.m:60-66 scale it).
 *
 * szFloesResync replaces the resident array by `host` (any n, any ring layout: floes added, removed, reshaped by
 * FloeSimplify / fracture / corners / ridging, Subzero.m:169-217,317-372) like szFloesUpload, except that the stress
 * history and the Monte-Carlo points of new floe k are copied ON THE DEVICE from floe src[k] (1-based) of the previous
 * array when src[k] > 0; host->StressH / mcX / mcY / mcA are read only at the rows with src[k] == 0 (floes the
 * MATLAB-side code created or changed).  src == NULL is szFloesUpload.  List capacities grown by earlier steps stay. */
int szFloesUpdate(szContext *ctx, const szFloeState *host, const int32_t *ids, int32_t nids, int what);
int szFloesFetch(szContext *ctx, szFloeState *host, const int32_t *ids, int32_t nids, int what);
int szFloesResync(szContext *ctx, const szFloeState *host, const int32_t *src);
/* One timestep on the device-resident floes.  SZ_OK, an error, or SZ_STEP_NEEDS_MC (nothing changed; see above). */
int szStep(szContext *ctx, const szStepParams *params, szStepStats *stats);
/* How the steps of this floe array have run so far (bench.py, tests): out[k] for k < n, slots SZ_DIAG_*.  The three lists
 * are those of the pair-interaction kernel (k_narrow_team): what its roofline is priced on. */
enum {
  SZ_DIAG_STEPS = 0,        /* completed steps */
  SZ_DIAG_STEPS_FUSED,      /* ... whose front and back stages ran as the fused launch chain */
  SZ_DIAG_SPEC_MISSES,      /* steps redone because a capacity taken from the step before did not hold */
  SZ_DIAG_BACK_MISSES,      /* steps whose queued back half waited for the host after all */
  SZ_DIAG_LIST_HOT, SZ_DIAG_LIST_COLD, SZ_DIAG_LIST_WIDE,   /* last step: pairs on the team pass's three lists */
  SZ_DIAG_LIST_HANDED_ON,   /* last step: pairs the team pass or the classification left to the lane-per-pair kernels */
  SZ_DIAG_LIST_WALLS,       /* last step: floes against the domain box (general sweep) */
  SZ_DIAG_TEAM_VERTICES,    /* last step: stored ring vertices (both rings) summed over the pairs of the three lists */
  SZ_DIAG_CELL_CAP, SZ_DIAG_POT_CAP,   /* slots per grid cell / candidates per floe the lists hold now */
  SZ_DIAG_STEPS_AHEAD,      /* steps that used front kernels queued ahead by the step before (SZ_STEP_QUEUE_AHEAD) */
  SZ_DIAG_COUNT
};
int szStepDiagnostics(szContext *ctx, int64_t *out, int32_t n);
/* The *_ms fields and times_step of `stats` for the LAST completed step (waits for its events; the other fields are left). */
int szStepTimes(szContext *ctx, szStepStats *stats);
/* Diagnostic (tests/test_team_clip_gpu.py): the clip INSIDE the pair-interaction kernel alone -- the team form of the
 * intersection of two single-minimum rings (teamclip.hpp; Clipper's Execute(ctIntersection), private/clipper.cpp:1560-1621,
 * for that case), `lanes` = 16 or 32 lanes of a wave per pair.  Rings as szClipBatch takes them (2^32 fixed point, CSR).
 * Pair p: status[p] = 0 and its one output ring (out_n[p] vertices, 0 = empty) at outx / outy[16 p ...], or status[p] = 100:
 * not this form's case (more than 16 vertices, ties, several minima ...) -- the step hands such a pair to the general sweep. */
int szDebugTeamClipBatch(szContext *ctx, int32_t lanes, int32_t npairs, const int64_t *ax, const int64_t *ay, const int32_t *aoff,
                         const int64_t *bx, const int64_t *by, const int32_t *boff, int64_t *outx, int64_t *outy,
                         int32_t *out_n, int32_t *status);
/* floes (1-based, ascending) for which the last szStep returned SZ_STEP_NEEDS_MC */
int szMcEmptyDownload(szContext *ctx, int32_t *ids, int64_t cap, int64_t *count);
int szFloesDownload(szContext *ctx, szFloeState *host, int what);
/* Floe(i).interactions of the last step for the n real floes, CSR: 7 doubles per row
 * [partner (1-based; Inf = boundary), Fx, Fy, Px, Py, torque, overlap], row order of the reference. */
int szInteractionsDownload(szContext *ctx, int32_t *inter_off, double *inter, int64_t cap_rows);
/* candidate list of the last step's broad-phase (floe_interactions_all.m:103), 1-based (i, j) over the
 * extended list, in the reference's order; wall candidates excluded. */
int szPairsDownload(szContext *ctx, int32_t *pair_i, int32_t *pair_j, int64_t cap, int64_t *npairs,
                    int32_t *n_ext);

/* The ghost floes of the last step (floe_interactions_all.m:16-66) in the order of the extended list: ghost g is
 * Floe(n + g) there (g = 1..nghost).  src = the real floe whose ring and state it copies, parent = the entry of the
 * extended list it was made from (a real floe or an X ghost, :36,:57), Xi / Yi = its translated position -- what the
 * ridging / rafting tail needs to index Floe(a(jj,1)) over the extended list (:305-330,:385-458).  Any pointer may be
 * NULL. */
int szGhostsDownload(szContext *ctx, int32_t *src, int32_t *parent, double *Xi, double *Yi, int64_t cap, int64_t *nghost);

/* ---- multi-GPU: the step with its dominant stage (the per-pair narrow-phase) sharded over ranks ----------
 * One process per GPU, every rank holds the full floe array (szFloesUpload of the same data on each) and
 * runs the cheap stages redundantly with identical results; the candidate-pair list is cut in `world`
 * contiguous segments and rank r computes segment r.  Between Begin and Finish the caller all-gathers the
 * exported segments (RCCL over xGMI: torch.distributed all_gather_into_tensor on device buffers; see
 * subzero_amd/multigpu.py).  After Finish every rank holds the same state a single-GPU szStep produces.
 *
 *   szStepShardBegin (params, rank, world) -> info        front stages + narrow-phase on my segment
 *   szStepShardExport(d_send, info.seg_max, rows_max, info.status)
 *                                                         my results into a device buffer of
 *                                                         szShardSegmentBytes(seg_max, rows_max) bytes; rows_max
 *                                                         = the caller's guess of the most contact rows a segment
 *                                                         has (the same on every rank, e.g. from the last step)
 *   [caller: all-gather d_send of every rank into d_recv (world x segment bytes, rank order)]
 *   szStepShardFinish(d_recv, seg_max, rows_max, stats, &fx, &rows_needed)
 *                                                         import all segments, contact rows, sums
 * Each segment starts with a 32-byte header (pairs, contact rows, status, rows_max) written on the device, so no
 * host round trip is needed between Begin and the all-gather.  Finish reads the headers first: if a rank reports
 * an error every rank returns SZ_ERR_PEER; if a segment has more rows than rows_max every rank returns
 * SZ_SHARD_RESIZE with the size that fits in *rows_needed, and the caller repeats Export / all-gather / Finish.
 * A failure of the replicated front stages (info.replicated_failure = 1) is the same on every rank: nobody may
 * enter the exchange.  With szSetStream(ctx, s) all of this runs on the caller's stream (the one its collectives
 * are ordered on), otherwise Export and Forcing synchronise the context's own stream before they return.
 * d_recv must stay valid until Finish returns.  On steps that refresh the ocean/wind forcing (doInt) that
 * stage is sharded too: Finish evaluates the forcing of this rank's floe range only and sets fx = 1; then
 *   szStepShardForcing(d_send)    my FxOA / FyOA / torqueOA (szShardForcingBytes(ctx, world) bytes)
 *   [caller: all-gather]
 *   szStepShardEnd(d_recv, stats) import the other ranks' values, finish the step (calc_trajectory.m:174-234)
 * When fx = 0 the step is complete after Finish (szStepShardEnd is then a no-op). */
/* Segment bytes: header 32 + 12 per pair slot (rows, flags, rows-before as int32) + 40 per contact row. */
typedef struct szShardInfo {
  int64_t n_pairs;          /* candidate pairs of the step (same on every rank) */
  int64_t seg_lo, seg_hi;   /* my segment of the pair list */
  int64_t seg_max;          /* largest segment over all ranks */
  int64_t seg_rows;         /* -1: known on the device only (it travels in the segment header) */
  int32_t status;           /* SZ_OK or the error of this rank; hand it to szStepShardExport so that all ranks stop together */
  int32_t replicated_failure; /* 1: the error is in the replicated stages, identical on every rank */
} szShardInfo;
int64_t szShardSegmentBytes(int64_t seg_max, int64_t rows_max);
int szStepShardBegin(szContext *ctx, const szStepParams *params, int rank, int world, szShardInfo *info);
int szStepShardExport(szContext *ctx, void *d_send, int64_t seg_max, int64_t rows_max, int32_t status);
int szStepShardFinish(szContext *ctx, const void *d_recv, int64_t seg_max, int64_t rows_max, szStepStats *stats,
                      int32_t *forcing_exchange, int64_t *rows_needed);
int64_t szShardForcingBytes(szContext *ctx, int world);
int szStepShardForcing(szContext *ctx, void *d_send);
int szStepShardEnd(szContext *ctx, const void *d_recv, szStepStats *stats);

/* ---- calc_eulerian_data.m:1-192 on the device-resident floes -----------------------------------
 * Mass-weighted averages of the floe fields over the Nx x Ny grid of the domain box; floe-cell overlaps are clipped on
 * the GPU (ring against rectangle).  out: 18 fields of Ny x Nx doubles each, MATLAB column-major (element (jj, ii) at
 * jj + Ny * ii, jj counted from the top row as calc_eulerian_data.m:75 flips y), in the order
 *   u v du dv stress stressxx stressyx stressxy stressyy strainux strainvx strainuy strainvy c Over Mtot area h
 * (the fields of the struct the reference returns).  Nb > 0: the first Nb live floes are the topography -- they are
 * not averaged and their union is cut out of the cells (:10-24,:141-143); the cells whose rectangle meets the bounding
 * box of such a floe are computed with the clip kernel (rectangle 'dif' topography, then 'int' each floe: the polyshape
 * operations restated by the fixed-point clip, as polyclip.m uses it), every other cell by the plain rectangle clip.
 * kernel_ms (may be NULL): HIP-event time of the cell kernel. */
int szEulerianData(szContext *ctx, int Nx, int Ny, int Nb, const double *c2_boundary, int periodic, double *out, double *kernel_ms);

/* ---- corner grinding, the per-vertex decision: Physical_Processes/corners.m:13-92 ---------------
 * Call after a step, for the floes the driver selected (Subzero.m:339-348: Floe(~keep)); ids[q] is the 1-based number
 * of the q-th of them in the floe array, in ascending order as Floe(~keep) lists them.  For each, from the contact
 * rows of that step and the positions after it:
 *   count[q]   number of vertices considered (0: not eligible -- q < Nb, as the loop of :55 starts at 1+Nb of the
 *              SUBSET, or no contact row, :56)
 *   angles     interior angles in degrees (polygon_operations/polyangles.m), count[q] values from angles[offsets[q]]
 *   da         1 where the vertex lies in a partner's outline / outside c2_boundary or is nearest to a contact point
 *              (:72-88), at the same offsets
 * The caller completes :70-71,91 with its own random stream, in the order of q over the eligible floes:
 *   Anorm = 180 - 360/count;  grind = (rand(count,1) > angles/Anorm) & da;  sum(grind) > 1 -> frac_corner (MATLAB side).
 * offsets[nsub+1] is filled from the stored outline lengths; cap = room of angles / da in vertices (SZ_ERR_CAPACITY when
 * short).  n_ext (may be NULL): length of the extended list Floe0 the partner numbers were resolved in. */
int szCornerCandidates(szContext *ctx, const int32_t *ids, int nsub, int Nb, const double *c2_boundary, int32_t *offsets, int32_t *count,
                       double *angles, uint8_t *da, int64_t cap, int32_t *n_ext, double *kernel_ms);

/* ---- corner grinding, the cut itself for the plain case: Physical_Processes/frac_corner.m:1-188 ------------------------
 * For the floes ids[s] (1-based) and the `grind` flags the caller formed from szCornerCandidates' angles / da and its own
 * draws (corners.m:91; at offsets[s] .. offsets[s + 1], the layout szCornerCandidates returned): the pieces frac_corner.m
 * returns -- the floe with every broken corner replaced by its two cut points (:20-52, :159-185), then one triangle per
 * corner (:54-136) -- when every such corner is convex, its cut stays inside the two adjacent edges, clear of the
 * neighbouring cuts and of every other edge: then the polyshape operations of the reference have closed forms and no
 * boolean operation is needed.  npieces[s] = 0: not that case, nothing returned for the floe -- leave it to frac_corner.m.
 * Pieces of floe s: piece_off[s] .. piece_off[s + 1]; each 12 doubles
 *   kind (0 floe, 1 triangle), Xi, Yi, area, mass, h, inertia_moment, rmax, alive, ksi_ice, factor, vertices
 * (factor: what frac_corner.m multiplies Stress and StressH by, :121-122 / :171-172; dXi_p .. dksi_ice_p, Ui, Vi, strain,
 * StressCount carry over from the floe, :128-137; alpha_i, FxOA, FyOA, torqueOA, the collision fields = 0) and its closed
 * c_alpha = c0 ring at ring_off[piece] .. ring_off[piece + 1] of rx / ry.  The Monte-Carlo points of the new floes
 * (:124-126) are rand: the caller draws them.  Conventions assumed of polyshape: DESIGN.md section 2. */
int szCornerCuts(szContext *ctx, const int32_t *ids, int nsel, const int32_t *offsets, const uint8_t *grind, int32_t *npieces, int32_t *piece_off,
                 double *pieces, int64_t cap_pieces, int32_t *ring_off, double *rx, double *ry, int64_t cap_ring, double *kernel_ms);

/* ---- fracture, the deformation step: Physical_Processes/fracture_floe.m:14-52 ------------------
 * Call after a step, for the floes about to be fractured (fracture.m:51-55; ids 1-based).  Each is pushed by the
 * partner of its largest contact overlap: polyclip 'int' -> centroid and p_poly_dist of the overlap region -> partner
 * moved along the contact force -> polyclip 'dif' (both clips on the GPU).  changed[s] = 1 when fracture_floe.m:44-50
 * replaces the floe's outline: then Xi / Yi / area [s] hold the new centroid and area and cx / cy [ring_off[s] ..
 * ring_off[s+1]) the new c_alpha (open ring, as polyclip returns it); otherwise the floe is left as it is.  The device
 * copy of the floes is NOT modified: the Voronoi tessellation that follows (:54-124, rand / polybnd_voronoi / new
 * floes) stays on the MATLAB side and the array comes back through szFloesResync.  cap: room of cx / cy. */
int szFractureDeform(szContext *ctx, const int32_t *ids, int nsel, int32_t *changed, double *Xi, double *Yi, double *area,
                     int32_t *ring_off, double *cx, double *cy, int64_t cap);

/* polygon_operations/FloeSimplify.m:14-34: for each listed floe the floes it may touch -- alive, centres more than 1 m
 * apart and closer than the sum of the rmax -- as 1-based floe numbers in ascending order (CSR: off[nsel + 1], list).
 * Reads the resident positions; the caller takes c_alpha, Ui ... of those floes from its own array. */
int szPotentialInteractions(szContext *ctx, const int32_t *ids, int nsel, int32_t *off, int32_t *list, int64_t cap);

/* ---- multi-GPU, spatial partition (BASELINE north_star): floes owned by x-strips of the domain, one
 * neighbour exchange of halo floes per step --------------------------------------------------------------------
 * Every rank uploads the same array (szFloesUpload) and keeps it at its global size and index; after szSpatialInit
 * only the floes whose centroid lies in the rank's strip [edges[rank], edges[rank+1]) are its own: it alone
 * integrates them, and their stress history and Monte-Carlo points are current nowhere else.  Per step:
 *
 *   szSpatialPack   (periodic, d_left, d_right, caps, &info)   my floes within `halo` of each edge (26 scalars + ring:
 *                                                               208 + 16 nv bytes each) and the floes whose centroid
 *                                                               left the strip (whole state), into two device buffers
 *                                                               of szSpatialMessageBytes(caps) bytes
 *   [caller: send d_left to the left neighbour and d_right to the right one, receive theirs: RCCL send/recv over
 *    xGMI (torch.distributed batch_isend_irecv), strips form a ring when the domain is periodic]
 *   szSpatialUnpack (d_from_left, d_from_right, caps, &info)    (NULL for a side without neighbour)
 *   szStep          (params, stats)                             the unchanged step on what the rank now knows
 *
 * Results for owned floes (state, contact rows, kill / transfer) are bit-identical to the single-GPU step; a pair
 * between floes of two ranks is evaluated by both.  Partner numbers of GHOST floes in the contact rows are local to
 * the rank (its ghost list: szGhostsDownload); real partners carry their global number.  halo >= 2 x (2 max rmax)
 * and every strip at least as wide as the halo (szSpatialInit checks both).  info.overflow = 1: a message did not fit
 * its capacities (info.need_* tell what the edge needs): repeat Pack / exchange / Unpack with larger buffers before
 * szStep -- nothing has been lost, Pack can be called again. */
typedef struct szSpatialInfo {
  int64_t left_records, left_doubles, left_migrants;
  int64_t right_records, right_doubles, right_migrants;
  int64_t need_records, need_doubles, need_migrants;
  int64_t n_owned, n_present;    /* after Unpack: floes this rank integrates / knows this step */
  int32_t overflow, has_left, has_right, reserved;
} szSpatialInfo;
int szSpatialInit(szContext *ctx, int rank, int world, const double *edges, double halo);
/* The axis the strips are cut along: 0 = x (the default), 1 = y -- for a box that is longer in y, e.g. the 2e5 x 1e6 m
 * channel of /root/reference/Initialize_Model/initialize_boundaries.m:14-16.  Call before szSpatialInit; `edges` are then
 * values of that coordinate. */
int szSpatialSetAxis(szContext *ctx, int axis);
int64_t szSpatialMessageBytes(szContext *ctx, int64_t cap_records, int64_t cap_doubles, int64_t cap_migrants);
int szSpatialPack(szContext *ctx, int periodic, void *d_left, void *d_right, int64_t cap_records, int64_t cap_doubles,
                  int64_t cap_migrants, szSpatialInfo *info);
int szSpatialUnpack(szContext *ctx, const void *d_from_left, const void *d_from_right, int64_t cap_records,
                    int64_t cap_doubles, int64_t cap_migrants, szSpatialInfo *info);
/* szSpatialUnpack for a caller that has read the message headers itself (it needs them to size the second exchange
 * round): left_counts / right_counts = {records, doubles, migrants} of the received messages; nothing is read back
 * from the device, and with info == NULL nothing is counted either: no host round trip at all. */
int szSpatialUnpackKnown(szContext *ctx, const void *d_from_left, const void *d_from_right, int64_t cap_records,
                         int64_t cap_doubles, int64_t cap_migrants, const int64_t *left_counts, const int64_t *right_counts,
                         szSpatialInfo *info);
int szSpatialOwned(szContext *ctx, uint8_t *owned);   /* owned[i] = 1: this rank integrates floe i (n bytes) */

/* The same exchange with ONE host round trip per step, the step's own.  Both ends of an edge agree beforehand, from what the
 * edge carried the step before (+ a margin), how much of a message travels: plan = {records, doubles, migrants} -- room
 * for the rows (stress history, Monte-Carlo points: 49 KB in a model run) of a couple of floes changing strip included, so
 * that the odd floe that crosses an edge needs no second round.
 *
 *   szSpatialPackQueued   (periodic, d_left, d_right, caps, plan_left, plan_right)
 *                                                                queued on the context's stream; nothing is read back
 *   [caller: header + the planned leading part of the two record arrays, each way, one grouped round]
 *   szSpatialUnpackQueued (d_from_left, d_from_right, caps, plan_left, plan_right)
 *                                                                queued: the DEVICE reads each header; a message within its
 *                                                                plan is taken; one that outgrew it (more records, doubles
 *                                                                or floes changing strip than agreed, or an overflow on the
 *                                                                sender's side) is left alone and vetoes the step
 *   szStep                                                       SZ_OK, or SZ_STEP_VETOED with nothing changed
 *   szSpatialHeaders      (headers[32], &vetoed)                 after that szStep, no device work: the four headers (my
 *                                                                message to the left / right, the message from the left /
 *                                                                right; 8 int64 each: records, doubles, migrants, status,
 *                                                                then what the edge needs) -- both ends of an edge see the
 *                                                                same numbers and so agree on whether a second round is due
 *   second round only (a message beyond its plan, floes changing strip):
 *   szSpatialPackFinish   (d_left, d_right, caps, &info)         the rows of the floes that changed strip into the messages
 *                                                                (info as szSpatialPack's; on overflow the ownership is given
 *                                                                back: szSpatialPack again with more room)
 *   [caller: the rest of each message], szSpatialUnpackKnown, and szStep again if the first one was vetoed.
 * In steady state (no floe crossing an edge this step) the host touches the device once per step instead of three times. */
int szSpatialPackQueued(szContext *ctx, int periodic, void *d_left, void *d_right, int64_t cap_records, int64_t cap_doubles,
                        int64_t cap_migrants, const int64_t *plan_left, const int64_t *plan_right);
int szSpatialPackFinish(szContext *ctx, void *d_left, void *d_right, int64_t cap_records, int64_t cap_doubles, int64_t cap_migrants,
                        szSpatialInfo *info);
int szSpatialUnpackQueued(szContext *ctx, const void *d_from_left, const void *d_from_right, int64_t cap_records,
                          int64_t cap_doubles, int64_t cap_migrants, const int64_t *plan_left, const int64_t *plan_right);
int szSpatialHeaders(szContext *ctx, int64_t *headers32, int32_t *vetoed);

/* clipper(P, delta, type [, info]) offset mode (mexclipper.cpp:146-203): ClipperOffset with AddPaths(P, jt,
 * etClosedPolygon), Execute(solution, delta).  x / y / ring_off: the nrings contours of P (int64, as the mex reads
 * them); delta in the same units (+ outset, - inset); join: 0 square ('s'), 1 round ('r'), 2 miter ('m'); info: the
 * optional fourth argument (ArcTolerance for round, MiterLimit for miter), <= 0 when not given (defaults 0.25, 2).
 * The raw offset polygons are built on the host (one pass over the vertices, with the host's libm as in the
 * reference), their clean-up -- the union of self-intersecting polygons -- runs on the GPU.  status: SZ_CLIP_*. */
int szOffsetPaths(szContext *ctx, const int64_t *x, const int64_t *y, const int32_t *ring_off, int nrings, double delta,
                  int join, double info, int64_t *outx, int64_t *outy, int64_t out_cap, int32_t *out_ring_off,
                  int64_t ring_cap, int32_t *nrings_out, int32_t *status);

/* clipper(P) orientation mode (mexclipper.cpp:134-145): 1 if Area(P) >= 0 else 0.  Host arithmetic
 * on n vertices (an O(n) sum; there is nothing to accelerate). */
int szOrientation(const int64_t *x, const int64_t *y, int n, int *orientation);

/* raw device memory helpers for callers without a HIP binding (bench.py, tests) */
int szDeviceAlloc(szContext *ctx, void **dptr, int64_t bytes);
int szDeviceFree(szContext *ctx, void *dptr);
int szCopyToDevice(szContext *ctx, void *dst, const void *src, int64_t bytes);
int szCopyToHost(szContext *ctx, void *dst, const void *src, int64_t bytes);
int szSynchronize(szContext *ctx);

#ifdef __cplusplus
}
#endif
#endif /* SUBZERO_HIP_H */
