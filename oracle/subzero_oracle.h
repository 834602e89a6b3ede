/* TEST INFRASTRUCTURE ONLY (oracle) -- only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this; nothing under subzero_amd/ does.
 *
 * C interface of the CPU restatement of SubZero's per-timestep hot path
 * (floe_interactions_all.m:9-285 and what it calls).  Plain structs of pointers so Python (ctypes)
 * can hand numpy arrays to it.  The field order of OrFloes / OrParams is the same as szFloeState /
 * szStepParams in include/subzero_hip.h so one set of arrays feeds both sides of a parity test.
 */
#ifndef SUBZERO_ORACLE_H
#define SUBZERO_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int32_t n, nb, n_stress, n_mc;
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
  int32_t *kill, *transfer; /* capacity 4*n: indexed over the extended (real + ghost) list */
} OrFloes;

typedef struct {
  double dt, HFo, Modulus, mu, nu;
  int32_t doInt, COLLISION, PERIODIC, reserved;
  double c2_boundary[10]; /* 2x5, MATLAB column-major: x0,y0,x1,y1,... */
  double fb_h, fb_area, fb_Xi, fb_Yi, fb_Ui, fb_Vi, fb_ksi_ice;
  int32_t nxo, nyo;
  const double *Xo, *Yo;       /* ocean grid axes */
  const double *Uocn, *Vocn;   /* nyo x nxo, MATLAB column-major (index y + nyo*x) */
  const double *Uwind, *Vwind; /* same shape */
  double fCoriolis, turn_angle;
} OrParams;

/* where the clip function comes from: path of a shared library exporting either
 *   ref_clip        (oracle/_ref/libclipper_ref.so, the reference's own Clipper)   kind = 0
 *   host_clip_batch (tests/hostsim/libvatti_host.so)                                kind = 1 */
int or_set_clipper(const char *path, int kind);

/* Definitions used where the reference leaves a MathWorks builtin unspecified; returns the previous mode.
 *   bit 0: libm sin/cos/atan2 instead of or_math.h; bit 1: Monte-Carlo means summed left to right instead of
 *   the 256-partial fold; bits 8-9 / bit 10: start vertex / reversed direction of the domain-box ring of the
 *   'dif' clip (floe_interactions.m:31-32).  0 = what the device path reproduces bit for bit. */
int or_set_mode(int mode);

/* One timestep.  Updates the floes in place.  Contact rows (floe_interactions_all.m:136,167,198) are
 * returned as CSR: floe i owns rows inter_off[i]..inter_off[i+1] of inter (7 doubles each,
 * [partner(1-based; Inf = wall), Fx, Fy, Px, Py, torque, overlap]).  Candidate pairs of the
 * broad-phase (:103) come back as (i,j) 1-based indices into the extended (real+ghost) list.
 * Returns 0, or <0 on error (-1 capacity, -2 clipper missing, -3 unsupported: Monte-Carlo regeneration). */
int or_step(OrFloes *f, const OrParams *p,
            int32_t *inter_off, double *inter, int64_t inter_cap,
            int32_t *pair_i, int32_t *pair_j, int64_t pair_cap, int64_t *npairs,
            int32_t *n_ext /* N: real + ghost floes */, int32_t threads);

/* pieces, exposed for unit tests against fixtures */
double or_ring_area(const double *x, const double *y, int n);
void or_ring_centroid(const double *x, const double *y, int n, double *cx, double *cy);
double or_polyarea(const double *x, const double *y, int n);
int or_interx(const double *x1, const double *y1, int n1, const double *x2, const double *y2, int n2,
              double *px, double *py, int cap);
void or_p_poly_dist(const double *xp, const double *yp, int np, const double *xv, const double *yv, int nv,
                    double *dmin);
void or_inpolygon(const double *x, const double *y, int np, const double *xv, const double *yv, int nv,
                  uint8_t *in);
/* floe_interactions.m for one pair; floe2_is_boundary selects the 'alive'-field branch (:13,:31-40).
 * s1/s2 = [Xi, Yi, Ui, Vi, ksi_ice, h, area].  Outputs up to cap rows: force(2), pcontact(2), overlap.
 * Returns number of rows (>=1; the [0 0],[0 0],0/+-Inf sentinel is one row). */
int or_floe_interactions(const double *c1x, const double *c1y, int n1, const double *s1,
                         const double *c2x, const double *c2y, int n2, const double *s2,
                         int floe2_is_boundary, const double *c2_boundary, int periodic,
                         double Modulus, double dt, double mu, double nu,
                         double *force, double *pcontact, double *overlap, int cap);
void or_sincos(double x, double *s, double *c);
double or_atan2(double y, double x);
int64_t or_matlab_int64_scaled(double v);

#ifdef __cplusplus
}
#endif
#endif
