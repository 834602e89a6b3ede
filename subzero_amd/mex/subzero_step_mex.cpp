// subzero_step_mex.cpp -- MATLAB gateway of the B2 surface: floe_interactions_all.m:16-285 in one call, on the
// device-resident Floe struct array of libsubzero_hip.so (include/subzero_hip.h).  INTEGRATION.md shows the edit of
// floe_interactions_all.m that calls it.  Build on a MATLAB host with
//   mex -R2017b subzero_step_mex.cpp -output subzero_step -I<repo>/include -L<repo>/subzero_amd/csrc -lsubzero_hip
// (this repository has no MATLAB: tests/mexstub/ provides a working test double of the mx* / mex* calls used here, and
// tests/test_mex_*.py compile this file against it and drive it end to end.)
//
//   Floe = subzero_step('upload', Floe, Nb)
//       the whole struct array (initialize_floe_values.m:12-52) to the GPU.  Refused while the device holds a newer
//       stress history than the Floe passed in (see 'step'): use 'resync', or 'fetch' the history first.
//   Floe = subzero_step('resync', Floe, Nb)
//       after MATLAB-side changes (dead floes removed, floe_interactions_all.m:12-13; FloeSimplify, fracture, corners,
//       ridging, welding, new ice: Subzero.m:169-217,317-372): Floe(k).gpuIndex names the floe of the resident array
//       that Floe(k) still is; its 32 KB stress history and 17 KB of Monte-Carlo points then stay on the device.  0, a
//       missing field, or a floe whose ring / area / mass no longer match what the gateway last saw = new or changed:
//       everything of it is read from Floe(k).  Returns Floe with gpuIndex = 1..n.
//   [Floe, kill, transfer, ghosts, needMC] = subzero_step('step', Floe, floebound, ocean, winds, c2_boundary, dt, HFo,
//                                                         Modulus, doIntFlag, COLLISION, PERIODIC [, mu, nu])
//       one timestep.  Written back into Floe: every dynamic scalar, c_alpha, Stress, strain, StressCount, alive,
//       interactions, collision_force, collision_torque, OverlapArea.  NOT StressH: the history stays on the device
//       (it is 97 % of the bytes and MATLAB-side code reads it only for the few floes it reshapes) -- 'fetch' it when
//       needed.  kill / transfer: as floe_interactions_all.m:123,138-145,175-179 leaves them (length N = real + ghost
//       floes).  ghosts: struct with row vectors src, parent, Xi, Yi -- ghost g is entry N0 + g of the reference's
//       extended Floe array (:16-66), a copy of Floe(src(g)) moved to (Xi(g), Yi(g)); the ridging / rafting tail
//       (:305-330,:385-458) indexes them.  needMC: floes whose ocean / wind forcing is due with an empty Monte-Carlo
//       mask (calc_trajectory.m:100-110 draws new points with rand).  When it is not empty NOTHING was changed: run the
//       reference's lines :102-109 for those floes, subzero_step('update', Floe, needMC, 'm') and call 'step' again.
//   [kill, transfer, needMC, dead, ghosts] = subzero_step('advance', floebound, ocean, winds, c2_boundary, dt, HFo, Modulus,
//                                                          doIntFlag, COLLISION, PERIODIC [, mu, nu])
//       the same timestep with nothing written back: the MATLAB array goes stale and is brought up to
//       date with 'fetch' (..., 'sri') at output steps (Subzero.m:220,285) and before MATLAB-side processes.  What the
//       tail of floe_interactions_all.m acts on every step comes back: kill / transfer (:467-501), the floes that died.
//       'upload' / 'resync' / 'update' of state refuse a stale array.
//   Floe = subzero_step('fetch', Floe, ids, what)     what: any of 's' state, 'r' c0 / c_alpha, 'h' StressH, 'm' X, Y, A,
//                                                     'i' interactions of the last step
//   subzero_step('update', Floe, ids, what)           the same groups from Floe(ids) to the device (same ring lengths)
//   C = subzero_step('corners', ids, Nb, c2_boundary)
//       after a 'step': the per-vertex part of corners.m:13-92 for the floes Floe(ids) (ids = find(~keep), Subzero.m:
//       341-348).  C(q).angles: interior angles (polyangles.m), C(q).da: the contact / inside flags of :72-88; both []
//       for a floe the loop of :55-56 skips.  The caller draws `rand(n,1) > C(q).angles/Anorm` and calls frac_corner as
//       before.
//   eul = subzero_step('eulerian', Nx, Ny, Nb, c2_boundary, PERIODIC)
//       calc_eulerian_data.m:1-192 on the resident floes: the struct of Ny x Nx fields the reference returns.
//   D = subzero_step('deform', ids)
//       after a 'step': fracture_floe.m:14-52 for Floe(ids).  D(q).changed; when 1, D(q).Xi / Yi / area / c_alpha are
//       the values :46-48 assign (c_alpha 2 x m, open).  The resident floes are not modified.
//   Pt = subzero_step('potential', ids)
//       FloeSimplify.m:14-34: Pt(q).floeNum = the floes Floe(ids(q)) may touch (row vector, ascending).
//   subzero_step('close')
// The struct <-> struct-of-arrays marshalling below is the only work done on the MATLAB thread.
#include "mex.h"
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "subzero_hip.h"

namespace {

szContext *g_ctx = nullptr;
void at_exit(void) { if (g_ctx) { szDestroy(g_ctx); g_ctx = nullptr; } }

// host image of the resident array (what the gateway last sent or received)
struct Soa {
  int n = 0, nb = 0, n_stress = 0, n_mc = 0;
  std::vector<double> Xi, Yi, alpha_i, Ui, Vi, ksi_ice, h, area, mass, inertia_moment, rmax, dXi_p, dYi_p, dUi_p,
      dVi_p, dalpha_i_p, dksi_ice_p, FxOA, FyOA, torqueOA, c0x, c0y, cax, cay, StressH, Stress, strain, mcX, mcY,
      collision_force, collision_torque, OverlapArea;
  std::vector<int32_t> alive, vert_off, StressCount, kill, transfer;
  std::vector<uint8_t> mcA;
  szFloeState view() {
    szFloeState s; memset(&s, 0, sizeof(s));
    s.n = n; s.nb = nb; s.n_stress = n_stress; s.n_mc = n_mc;
#define P(f) s.f = f.data();
    P(Xi) P(Yi) P(alpha_i) P(Ui) P(Vi) P(ksi_ice) P(h) P(area) P(mass) P(inertia_moment) P(rmax) P(dXi_p) P(dYi_p)
    P(dUi_p) P(dVi_p) P(dalpha_i_p) P(dksi_ice_p) P(FxOA) P(FyOA) P(torqueOA) P(alive) P(vert_off) P(c0x) P(c0y)
    P(cax) P(cay) P(StressH) P(StressCount) P(Stress) P(strain) P(mcX) P(mcY) P(mcA) P(collision_force)
    P(collision_torque) P(OverlapArea) P(kill) P(transfer)
#undef P
    return s;
  }
  // rows `ids` (1-based) of the field groups in `what` taken over from `from` (same n, same ring layout): what an
  // 'update' has sent to the device -- and nothing else, so that a floe changed in MATLAB but not listed is still seen
  // as changed by the next 'resync'
  void take_rows(const Soa &from, const int32_t *ids, int nids, int what) {
    for (int k = 0; k < nids; ++k) {
      const size_t i = (size_t)ids[k] - 1;
      if (what & SZ_F_STATE) {
#define R1(f) f[i] = from.f[i];
        R1(Xi) R1(Yi) R1(alpha_i) R1(Ui) R1(Vi) R1(ksi_ice) R1(h) R1(area) R1(mass) R1(inertia_moment) R1(rmax) R1(dXi_p) R1(dYi_p)
        R1(dUi_p) R1(dVi_p) R1(dalpha_i_p) R1(dksi_ice_p) R1(FxOA) R1(FyOA) R1(torqueOA) R1(alive) R1(StressCount) R1(collision_torque) R1(OverlapArea)
#undef R1
        for (int q = 0; q < 4; ++q) Stress[4 * i + q] = from.Stress[4 * i + q];
        for (int q = 0; q < 4; ++q) strain[4 * i + q] = from.strain[4 * i + q];
        for (int q = 0; q < 2; ++q) collision_force[2 * i + q] = from.collision_force[2 * i + q];
      }
      if (what & SZ_F_RING)
        for (int v = vert_off[i]; v < vert_off[i + 1]; ++v) { c0x[v] = from.c0x[v]; c0y[v] = from.c0y[v]; cax[v] = from.cax[v]; cay[v] = from.cay[v]; }
      if (what & SZ_F_STRESSH)
        for (size_t q = 0; q < 4 * (size_t)n_stress; ++q) StressH[4 * (size_t)n_stress * i + q] = from.StressH[4 * (size_t)n_stress * i + q];
      if (what & SZ_F_MC)
        for (size_t q = 0; q < (size_t)n_mc; ++q) { mcX[(size_t)n_mc * i + q] = from.mcX[(size_t)n_mc * i + q]; mcY[(size_t)n_mc * i + q] = from.mcY[(size_t)n_mc * i + q]; mcA[(size_t)n_mc * i + q] = from.mcA[(size_t)n_mc * i + q]; }
    }
  }
};
Soa g_soa;              // the resident array as the gateway knows it
bool g_resident = false;
bool g_history_ahead = false;   // the device has stepped since the host last saw StressH
bool g_state_ahead = false;     // 'advance' has stepped since the host last saw the dynamic fields and c_alpha

void fail(const char *fmt, ...) {
  char buf[768];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  mexErrMsgTxt(buf);
}
void check(int rc) { if (rc != SZ_OK) mexErrMsgTxt(szLastError(g_ctx)); }

const mxArray *need_field(const mxArray *F, size_t i, const char *name) {
  const mxArray *a = mxGetField(F, i, name);
  if (!a) fail("subzero_step: Floe(%d) has no field '%s'", (int)i + 1, name);
  return a;
}
double scalar_field(const mxArray *F, size_t i, const char *name) {
  const mxArray *a = mxGetField(F, i, name);
  return (a && mxGetNumberOfElements(a)) ? mxGetScalar(a) : 0.0;   // ([] reads as 0, e.g. FxOA before the first forcing)
}
const double *doubles(const mxArray *a, size_t count, const char *what, int floe) {
  if (!mxIsDouble(a) || mxGetNumberOfElements(a) != count) fail("subzero_step: Floe(%d).%s must hold %d doubles", floe + 1, what, (int)count);
  return mxGetPr(a);
}

// Floe struct array -> struct of arrays.  n_stress / n_mc follow the arrays (size(StressH,3), numel(X)).
void pack(const mxArray *F, int nb, Soa &S) {
  if (!mxIsStruct(F) || mxGetNumberOfElements(F) == 0) fail("subzero_step: Floe must be a non-empty struct array");
  const int n = (int)mxGetNumberOfElements(F);
  S.n = n; S.nb = nb;
  if (nb < 0 || nb > n) fail("subzero_step: Nb = %d with %d floes", nb, n);
  {
    const mxArray *sh = need_field(F, 0, "StressH"), *x = need_field(F, 0, "X");
    const mwSize nd = mxGetNumberOfDimensions(sh);
    const mwSize *d = mxGetDimensions(sh);
    if (nd < 2 || d[0] != 2 || d[1] != 2) fail("subzero_step: StressH must be 2 x 2 x Nz");
    S.n_stress = nd >= 3 ? (int)d[2] : 1;
    S.n_mc = (int)mxGetNumberOfElements(x);
    if (S.n_stress < 1 || S.n_mc < 1) fail("subzero_step: empty StressH or X");
  }
#define SC(f) S.f.resize(n); for (int i = 0; i < n; ++i) S.f[i] = scalar_field(F, i, #f);
  SC(Xi) SC(Yi) SC(alpha_i) SC(Ui) SC(Vi) SC(ksi_ice) SC(h) SC(area) SC(mass) SC(inertia_moment) SC(rmax) SC(dXi_p)
  SC(dYi_p) SC(dUi_p) SC(dVi_p) SC(dalpha_i_p) SC(dksi_ice_p) SC(FxOA) SC(FyOA) SC(torqueOA) SC(collision_torque) SC(OverlapArea)
#undef SC
  S.alive.resize(n); S.StressCount.resize(n); S.vert_off.assign((size_t)n + 1, 0);
  for (int i = 0; i < n; ++i) {
    S.alive[i] = (int)scalar_field(F, i, "alive");
    S.StressCount[i] = (int)scalar_field(F, i, "StressCount");
    const mxArray *c0 = need_field(F, i, "c0"), *ca = need_field(F, i, "c_alpha");
    if (mxGetM(c0) != 2 || mxGetM(ca) != 2 || mxGetN(c0) != mxGetN(ca) || mxGetN(c0) < 3) fail("subzero_step: Floe(%d).c0 / c_alpha must be 2 x (n+1)", i + 1);
    S.vert_off[i + 1] = S.vert_off[i] + (int)mxGetN(c0);
  }
  const size_t nv = (size_t)S.vert_off[n], shr = (size_t)S.n_stress * 4, mcr = (size_t)S.n_mc;
  S.c0x.resize(nv); S.c0y.resize(nv); S.cax.resize(nv); S.cay.resize(nv);
  S.StressH.resize((size_t)n * shr); S.Stress.assign(4 * (size_t)n, 0); S.strain.assign(4 * (size_t)n, 0);
  S.mcX.resize((size_t)n * mcr); S.mcY.resize((size_t)n * mcr); S.mcA.resize((size_t)n * mcr);
  S.collision_force.assign(2 * (size_t)n, 0);
  for (int i = 0; i < n; ++i) {
    const int nvi = S.vert_off[i + 1] - S.vert_off[i];
    const double *c0 = doubles(need_field(F, i, "c0"), 2 * (size_t)nvi, "c0", i), *ca = doubles(need_field(F, i, "c_alpha"), 2 * (size_t)nvi, "c_alpha", i);
    for (int q = 0; q < nvi; ++q) {
      const size_t k = (size_t)S.vert_off[i] + q;
      S.c0x[k] = c0[2 * q]; S.c0y[k] = c0[2 * q + 1]; S.cax[k] = ca[2 * q]; S.cay[k] = ca[2 * q + 1];
    }
    memcpy(&S.StressH[(size_t)i * shr], doubles(need_field(F, i, "StressH"), shr, "StressH", i), shr * 8);   // 2 x 2 x Nz, column-major
    memcpy(&S.mcX[(size_t)i * mcr], doubles(need_field(F, i, "X"), mcr, "X", i), mcr * 8);
    memcpy(&S.mcY[(size_t)i * mcr], doubles(need_field(F, i, "Y"), mcr, "Y", i), mcr * 8);
    const mxArray *A = need_field(F, i, "A");
    if (mxGetNumberOfElements(A) != mcr) fail("subzero_step: Floe(%d).A must have %d elements", i + 1, (int)mcr);
    if (mxIsLogical(A)) { const mxLogical *a = mxGetLogicals(A); for (size_t q = 0; q < mcr; ++q) S.mcA[(size_t)i * mcr + q] = a[q] ? 1 : 0; }
    else { const double *a = doubles(A, mcr, "A", i); for (size_t q = 0; q < mcr; ++q) S.mcA[(size_t)i * mcr + q] = a[q] != 0; }
    const mxArray *st = mxGetField(F, i, "Stress"), *sn = mxGetField(F, i, "strain"), *cf = mxGetField(F, i, "collision_force");
    if (st && mxIsDouble(st) && mxGetNumberOfElements(st) == 4) memcpy(&S.Stress[4 * (size_t)i], mxGetPr(st), 32);
    if (sn && mxIsDouble(sn) && mxGetNumberOfElements(sn) == 4) memcpy(&S.strain[4 * (size_t)i], mxGetPr(sn), 32);
    if (cf && mxIsDouble(cf)) for (size_t q = 0; q < 2 && q < mxGetNumberOfElements(cf); ++q) S.collision_force[2 * (size_t)i + q] = mxGetPr(cf)[q];
  }
  S.kill.assign(4 * (size_t)n, 0); S.transfer.assign(4 * (size_t)n, 0);
}

// replaces a field value (mxSetField does not free the old one)
void put_field(mxArray *F, size_t i, const char *name, mxArray *v) {
  int k = mxGetFieldNumber(F, name);
  if (k < 0) k = mxAddField(F, name);
  mxArray *old = mxGetFieldByNumber(F, i, k);
  if (old) mxDestroyArray(old);
  mxSetFieldByNumber(F, i, k, v);
}
void put_scalar(mxArray *F, size_t i, const char *name, double v) {
  mxArray *a = mxGetField(F, i, name);
  if (a && mxIsDouble(a) && mxGetNumberOfElements(a) == 1) { *mxGetPr(a) = v; return; }
  put_field(F, i, name, mxCreateDoubleScalar(v));
}
mxArray *matrix_from(size_t m, size_t n, const double *src) {
  mxArray *a = mxCreateDoubleMatrix(m, n, mxREAL);
  if (m * n) memcpy(mxGetPr(a), src, m * n * 8);
  return a;
}
void put_gpu_index(mxArray *F, int n) { for (int i = 0; i < n; ++i) put_scalar(F, (size_t)i, "gpuIndex", (double)(i + 1)); }

// the fields a step changes, from S into the struct array (`ids` == nullptr: every floe)
void unpack_state(mxArray *F, const Soa &S, const int32_t *ids, int nids) {
  const int cnt = ids ? nids : S.n;
  for (int t = 0; t < cnt; ++t) {
    const size_t i = ids ? (size_t)(ids[t] - 1) : (size_t)t;
#define W(f) put_scalar(F, i, #f, S.f[i]);
    W(Xi) W(Yi) W(alpha_i) W(Ui) W(Vi) W(ksi_ice) W(h) W(area) W(mass) W(inertia_moment) W(rmax) W(dXi_p) W(dYi_p) W(dUi_p) W(dVi_p)
    W(dalpha_i_p) W(dksi_ice_p) W(FxOA) W(FyOA) W(torqueOA) W(collision_torque) W(OverlapArea)
#undef W
    put_scalar(F, i, "alive", S.alive[i]);
    put_scalar(F, i, "StressCount", S.StressCount[i]);
    put_field(F, i, "collision_force", matrix_from(1, 2, &S.collision_force[2 * i]));
    put_field(F, i, "Stress", matrix_from(2, 2, &S.Stress[4 * i]));
    put_field(F, i, "strain", matrix_from(2, 2, &S.strain[4 * i]));
  }
}
void unpack_rings(mxArray *F, const Soa &S, const int32_t *ids, int nids, bool c0_too) {
  const int cnt = ids ? nids : S.n;
  for (int t = 0; t < cnt; ++t) {
    const size_t i = ids ? (size_t)(ids[t] - 1) : (size_t)t;
    const int v0 = S.vert_off[i], nv = S.vert_off[i + 1] - v0;
    mxArray *ca = mxCreateDoubleMatrix(2, nv, mxREAL);
    for (int k = 0; k < nv; ++k) { mxGetPr(ca)[2 * k] = S.cax[v0 + k]; mxGetPr(ca)[2 * k + 1] = S.cay[v0 + k]; }
    put_field(F, i, "c_alpha", ca);
    if (c0_too) {
      mxArray *c0 = mxCreateDoubleMatrix(2, nv, mxREAL);
      for (int k = 0; k < nv; ++k) { mxGetPr(c0)[2 * k] = S.c0x[v0 + k]; mxGetPr(c0)[2 * k + 1] = S.c0y[v0 + k]; }
      put_field(F, i, "c0", c0);
    }
  }
}
void unpack_bulk(mxArray *F, const Soa &S, const int32_t *ids, int nids, bool hist, bool mc) {
  const size_t shr = (size_t)S.n_stress * 4, mcr = (size_t)S.n_mc;
  for (int t = 0; t < nids; ++t) {
    const size_t i = (size_t)(ids[t] - 1);
    if (hist) {
      const mwSize d[3] = {2, 2, (mwSize)S.n_stress};
      mxArray *a = mxCreateNumericArray(3, d, mxDOUBLE_CLASS, mxREAL);
      memcpy(mxGetPr(a), &S.StressH[i * shr], shr * 8);
      put_field(F, i, "StressH", a);
    }
    if (mc) {
      put_field(F, i, "X", matrix_from(mcr, 1, &S.mcX[i * mcr]));
      put_field(F, i, "Y", matrix_from(mcr, 1, &S.mcY[i * mcr]));
      mxArray *A = mxCreateLogicalMatrix(mcr, 1);
      for (size_t q = 0; q < mcr; ++q) mxGetLogicals(A)[q] = S.mcA[i * mcr + q] != 0;
      put_field(F, i, "A", A);
    }
  }
}

enum { F_INTERACTIONS = 1 << 8 };   // 'i': Floe(ids).interactions of the last step (gateway-level group, not a szFloesFetch one)
int parse_what(const mxArray *a) {
  char w[16];
  if (!a || !mxIsChar(a) || mxGetString(a, w, sizeof(w))) fail("subzero_step: `what` must be a short char array of s, r, h, m, i");
  int what = 0;
  for (const char *p = w; *p; ++p) {
    if (*p == 's') what |= SZ_F_STATE; else if (*p == 'r') what |= SZ_F_RING; else if (*p == 'h') what |= SZ_F_STRESSH;
    else if (*p == 'm') what |= SZ_F_MC; else if (*p == 'i') what |= F_INTERACTIONS; else fail("subzero_step: unknown field group '%c'", *p);
  }
  return what;
}
std::vector<int32_t> parse_ids(const mxArray *a, int n) {
  if (!a || !mxIsDouble(a)) fail("subzero_step: ids must be a double vector of floe numbers");
  std::vector<int32_t> ids(mxGetNumberOfElements(a));
  for (size_t k = 0; k < ids.size(); ++k) {
    const double v = mxGetPr(a)[k];
    if (!(v >= 1 && v <= n) || v != floor(v)) fail("subzero_step: floe number %g out of range 1..%d", v, n);
    ids[k] = (int32_t)v;
  }
  return ids;
}

// does floe k of `now` still carry what floe g of the resident image did?  (outline and size: every MATLAB-side
// process that rewrites a floe's history or Monte-Carlo points also changes its outline)
bool same_floe(const Soa &now, int k, const Soa &was, int g) {
  const int nv = now.vert_off[k + 1] - now.vert_off[k];
  if (nv != was.vert_off[g + 1] - was.vert_off[g]) return false;
  if (now.area[k] != was.area[g] || now.rmax[k] != was.rmax[g]) return false;
  return !memcmp(&now.c0x[now.vert_off[k]], &was.c0x[was.vert_off[g]], (size_t)nv * 8) &&
         !memcmp(&now.c0y[now.vert_off[k]], &was.c0y[was.vert_off[g]], (size_t)nv * 8);
}


// Floe(ids).interactions (all floes when ids == nullptr) of the last step: K x 7, column-major, [] when the floe touched nothing
void put_interactions(mxArray *F, int n, const int32_t *ids, int nids) {
  std::vector<int32_t> off((size_t)n + 1);
  int rc = szInteractionsDownload(g_ctx, off.data(), nullptr, 0);     // offsets first: they say how many rows there are
  if (rc != SZ_OK && rc != SZ_ERR_CAPACITY && rc != SZ_ERR_BAD_ARG) check(rc);
  std::vector<double> rows((size_t)(off[n] > 0 ? off[n] : 1) * 7);
  check(szInteractionsDownload(g_ctx, off.data(), rows.data(), (int64_t)(rows.size() / 7)));
  const int cnt = ids ? nids : n;
  for (int q = 0; q < cnt; ++q) {
    const int i = ids ? ids[q] - 1 : q;
    const int k0 = off[i], K = off[i + 1] - k0;
    mxArray *in = mxCreateDoubleMatrix(K, K ? 7 : 0, mxREAL);
    for (int r = 0; r < K; ++r) for (int c = 0; c < 7; ++c) mxGetPr(in)[(size_t)c * K + r] = rows[(size_t)7 * (k0 + r) + c];
    put_field(F, (size_t)i, "interactions", in);
  }
}

// floebound, ocean, winds, c2_boundary, dt, HFo, Modulus, doInt, COLLISION, PERIODIC [, mu, nu] from a[0..]
void step_params(int na, const mxArray **a, szStepParams &p) {
  if (na < 10) mexErrMsgTxt("subzero_step: floebound, ocean, winds, c2_boundary, dt, HFo, Modulus, doInt, COLLISION, PERIODIC [, mu, nu] expected");
  const mxArray *fb = a[0], *oc = a[1], *wi = a[2];
  memset(&p, 0, sizeof(p));
  memcpy(p.c2_boundary, doubles(a[3], 10, "c2_boundary", -1), 10 * sizeof(double));
  p.dt = mxGetScalar(a[4]);
  { const mxArray *H = a[5]; const size_t m = mxGetNumberOfElements(H); double s = 0; for (size_t k = 0; k < m; ++k) s += mxGetPr(H)[k]; p.HFo = m ? s / (double)m : 0; } // mean(HFo(:)), calc_trajectory.m:5
  p.Modulus = mxGetScalar(a[6]);
  p.mu = na > 10 ? mxGetScalar(a[10]) : 0.2; p.nu = na > 11 ? mxGetScalar(a[11]) : 0.3;   // floe_interactions.m:20-21
  p.doInt = mxGetScalar(a[7]) != 0; p.COLLISION = mxGetScalar(a[8]) != 0; p.PERIODIC = mxGetScalar(a[9]) != 0;
  p.fb_h = scalar_field(fb, 0, "h"); p.fb_area = scalar_field(fb, 0, "area"); p.fb_Xi = scalar_field(fb, 0, "Xi");
  p.fb_Yi = scalar_field(fb, 0, "Yi"); p.fb_Ui = scalar_field(fb, 0, "Ui"); p.fb_Vi = scalar_field(fb, 0, "Vi");
  p.fb_ksi_ice = scalar_field(fb, 0, "ksi_ice");
  const mxArray *Xo = need_field(oc, 0, "Xo"), *Yo = need_field(oc, 0, "Yo");
  p.nxo = (int)mxGetNumberOfElements(Xo); p.nyo = (int)mxGetNumberOfElements(Yo);
  const size_t grid = (size_t)p.nxo * p.nyo;
  p.Xo = mxGetPr(Xo); p.Yo = mxGetPr(Yo);
  p.Uocn = doubles(need_field(oc, 0, "Uocn"), grid, "Uocn", -1); p.Vocn = doubles(need_field(oc, 0, "Vocn"), grid, "Vocn", -1);
  p.Uwind = doubles(need_field(wi, 0, "u"), grid, "u", -1); p.Vwind = doubles(need_field(wi, 0, "v"), grid, "v", -1);
  p.fCoriolis = scalar_field(oc, 0, "fCoriolis"); p.turn_angle = scalar_field(oc, 0, "turn_angle");
}
mxArray *row_of(const std::vector<int32_t> &v, size_t count) {
  mxArray *a = mxCreateDoubleMatrix(1, count, mxREAL);
  for (size_t k = 0; k < count; ++k) mxGetPr(a)[k] = v[k];
  return a;
}
mxArray *ghost_struct(int G) {
  std::vector<int32_t> gs((size_t)G + 1), gp((size_t)G + 1);
  std::vector<double> gx((size_t)G + 1), gy((size_t)G + 1);
  int64_t cnt = 0;
  if (G >= 0) check(szGhostsDownload(g_ctx, gs.data(), gp.data(), gx.data(), gy.data(), G, &cnt));
  const char *names[] = {"src", "parent", "Xi", "Yi"};
  mxArray *out = mxCreateStructMatrix(1, 1, 4, names);
  mxArray *a[4];
  for (int k = 0; k < 4; ++k) a[k] = mxCreateDoubleMatrix(1, (size_t)cnt, mxREAL);
  for (int64_t g = 0; g < cnt; ++g) { mxGetPr(a[0])[g] = gs[(size_t)g]; mxGetPr(a[1])[g] = gp[(size_t)g]; mxGetPr(a[2])[g] = gx[(size_t)g]; mxGetPr(a[3])[g] = gy[(size_t)g]; }
  for (int k = 0; k < 4; ++k) mxSetFieldByNumber(out, 0, k, a[k]);
  return out;
}

} // namespace

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgTxt("subzero_step(command, ...)");
  char cmd[16];
  mxGetString(prhs[0], cmd, sizeof(cmd));
  if (!strcmp(cmd, "close")) { at_exit(); g_resident = false; g_history_ahead = false; g_state_ahead = false; g_soa = Soa(); return; }
  const bool upload = !strcmp(cmd, "upload"), resync = !strcmp(cmd, "resync");
  if (upload || resync) {
    if (nrhs < 3) fail("Floe = subzero_step('%s', Floe, Nb)", cmd);
    if (g_resident && g_state_ahead)
      fail("subzero_step('%s'): the device has stepped ('advance') since this Floe array was current; "
           "Floe = subzero_step('fetch', Floe, 1:numel(Floe), 'sr') first", cmd);
    Soa S;
    pack(prhs[1], (int)mxGetScalar(prhs[2]), S);     // argument errors first: they do not need a GPU
    if (!g_ctx) {
      if (szCreate(&g_ctx, 0) != SZ_OK) mexErrMsgTxt("subzero_hip: no usable gfx950 GPU (there is no CPU fallback).");
      mexAtExit(at_exit);
    }
    szFloeState s = S.view();
    bool kept_any = false;
    if (upload || !g_resident) {
      if (g_resident && g_history_ahead)
        mexErrMsgTxt("subzero_step('upload'): the device holds a newer stress history than this Floe array (StressH is not "
                     "written back by 'step'); use 'resync', or 'fetch' the histories ('h') first");
      check(szFloesUpload(g_ctx, &s));
    } else {
      if (S.n_stress != g_soa.n_stress || S.n_mc != g_soa.n_mc) mexErrMsgTxt("subzero_step('resync'): the size of StressH or of X changed");
      std::vector<int32_t> src((size_t)S.n, 0);
      for (int k = 0; k < S.n; ++k) {
        const int g = (int)scalar_field(prhs[1], (size_t)k, "gpuIndex");
        if (g >= 1 && g <= g_soa.n && same_floe(S, k, g_soa, g - 1)) { src[k] = g; kept_any = true; }
      }
      check(szFloesResync(g_ctx, &s, src.data()));
      // the image of the floes that stayed keeps the history / points the gateway last saw for them (not what the stale
      // Floe array carried); nobody reads those rows before a 'fetch' refreshes them
    }
    // (floes that stayed keep the device's history: what Floe carries for them is as old as it was before)
    g_history_ahead = kept_any && g_history_ahead;
    g_soa = S; g_resident = true;
    if (nlhs > 0) { plhs[0] = mxDuplicateArray(prhs[1]); put_gpu_index(plhs[0], S.n); }
    return;
  }
  if (!g_resident) mexErrMsgTxt("subzero_step: no floes on the device ('upload' first)");
  const int n = g_soa.n;
  if (!strcmp(cmd, "fetch") || !strcmp(cmd, "update")) {
    if (nrhs < 4) fail("subzero_step('%s', Floe, ids, what)", cmd);
    if ((int)mxGetNumberOfElements(prhs[1]) != n) mexErrMsgTxt("subzero_step: Floe has a different length than the resident array ('resync' first)");
    const std::vector<int32_t> ids = parse_ids(prhs[2], n);
    const int what = parse_what(prhs[3]);
    if (!strcmp(cmd, "update")) {
      Soa S;
      pack(prhs[1], g_soa.nb, S);
      szFloeState s = S.view();
      if (what & F_INTERACTIONS) mexErrMsgTxt("subzero_step('update'): interactions are a result of 'step', not an input");
      if (g_state_ahead && (what & (SZ_F_STATE | SZ_F_RING)))
        mexErrMsgTxt("subzero_step('update'): the device has stepped ('advance') since this Floe array was current; 'fetch' the floes ('sr') before changing and updating them");
      check(szFloesUpdate(g_ctx, &s, ids.data(), (int32_t)ids.size(), what));
      g_soa.take_rows(S, ids.data(), (int)ids.size(), what);   // (szFloesUpdate has checked that the ring layout is the resident one)
      return;
    }
    szFloeState s = g_soa.view();
    if (what & SZ_F_ALL) check(szFloesFetch(g_ctx, &s, ids.data(), (int32_t)ids.size(), what & SZ_F_ALL));
    plhs[0] = mxDuplicateArray(prhs[1]);
    if (what & F_INTERACTIONS) put_interactions(plhs[0], n, ids.data(), (int)ids.size());
    if (what & SZ_F_STATE) unpack_state(plhs[0], g_soa, ids.data(), (int)ids.size());
    if (what & SZ_F_RING) unpack_rings(plhs[0], g_soa, ids.data(), (int)ids.size(), true);
    unpack_bulk(plhs[0], g_soa, ids.data(), (int)ids.size(), (what & SZ_F_STRESSH) != 0, (what & SZ_F_MC) != 0);
    if ((what & SZ_F_STRESSH) && (int)ids.size() == n) g_history_ahead = false;
    if ((what & SZ_F_STATE) && (what & SZ_F_RING) && (int)ids.size() == n) g_state_ahead = false;
    return;
  }
  if (!strcmp(cmd, "corners")) {
    if (nrhs < 4) mexErrMsgTxt("C = subzero_step('corners', ids, Nb, c2_boundary)");
    const std::vector<int32_t> ids = parse_ids(prhs[1], n);
    const int nsub = (int)ids.size();
    const double *cb = doubles(prhs[3], 10, "c2_boundary", -1);
    int64_t cap = 0;
    for (int q = 0; q < nsub; ++q) cap += g_soa.vert_off[ids[q]] - g_soa.vert_off[ids[q] - 1];
    std::vector<int32_t> off((size_t)nsub + 1), cnt((size_t)nsub + 1);
    std::vector<double> ang((size_t)cap + 1);
    std::vector<uint8_t> da((size_t)cap + 1);
    check(szCornerCandidates(g_ctx, ids.data(), nsub, (int)mxGetScalar(prhs[2]), cb, off.data(), cnt.data(), ang.data(), da.data(), cap, nullptr, nullptr));
    const char *names[] = {"angles", "da"};
    plhs[0] = mxCreateStructMatrix(1, (size_t)nsub, 2, names);
    for (int q = 0; q < nsub; ++q) {
      const int m = cnt[q];
      mxArray *a = mxCreateDoubleMatrix((size_t)m, m ? 1 : 0, mxREAL), *d = mxCreateDoubleMatrix((size_t)m, m ? 1 : 0, mxREAL);
      for (int k = 0; k < m; ++k) { mxGetPr(a)[k] = ang[(size_t)off[q] + k]; mxGetPr(d)[k] = da[(size_t)off[q] + k]; }
      mxSetFieldByNumber(plhs[0], (size_t)q, 0, a);
      mxSetFieldByNumber(plhs[0], (size_t)q, 1, d);
    }
    return;
  }
  if (!strcmp(cmd, "cuts")) {
    // K = subzero_step('cuts', ids, grind): frac_corner.m for the plain case (szCornerCuts).  grind: the `grind` vectors
    // of corners.m:91 of the listed floes one after the other (each as long as the floe's outline without its closing
    // copy, or with it: a trailing entry is ignored).  K(q).pieces: P x 12 (0 rows: leave the floe to frac_corner.m),
    // K(q).rings: 2 x M, the closed c_alpha rings of the P pieces side by side, K(q).ring_off: 1 x (P + 1) (0-based)
    if (nrhs < 4) mexErrMsgTxt("K = subzero_step('cuts', ids, grind, lengths)   (grind: the floes' grind vectors one after the other, lengths: of each)");
    const std::vector<int32_t> ids = parse_ids(prhs[1], n);
    const int nsel = (int)ids.size();
    if ((int)mxGetNumberOfElements(prhs[3]) != nsel || !mxIsDouble(prhs[3])) mexErrMsgTxt("subzero_step('cuts'): lengths must hold one number per listed floe");
    if (!(mxIsDouble(prhs[2]) || mxIsLogical(prhs[2]))) mexErrMsgTxt("subzero_step('cuts'): grind must be logical or double");
    std::vector<int32_t> off((size_t)nsel + 1, 0);
    for (int q = 0; q < nsel; ++q) off[(size_t)q + 1] = off[q] + (g_soa.vert_off[ids[q]] - g_soa.vert_off[ids[q] - 1]);
    std::vector<uint8_t> grind((size_t)off[nsel] + 1, 0);
    int64_t ng = 0, at = 0;
    const int64_t total = (int64_t)mxGetNumberOfElements(prhs[2]);
    for (int q = 0; q < nsel; ++q) {
      const int m = (int)mxGetPr(prhs[3])[q];
      if (m < 0 || m > off[(size_t)q + 1] - off[q] || at + m > total) fail("subzero_step('cuts'): lengths(%d) = %d does not fit the floe (%d vertices) or the grind vector", q + 1, m, off[(size_t)q + 1] - off[q]);
      for (int k = 0; k < m; ++k) {
        const bool on = mxIsLogical(prhs[2]) ? ((const unsigned char *)mxGetData(prhs[2]))[at + k] != 0 : mxGetPr(prhs[2])[at + k] != 0;
        grind[(size_t)off[q] + k] = on ? 1 : 0; ng += on;
      }
      at += m;
    }
    const int64_t cap_p = nsel + ng + 1, cap_r = off[nsel] + nsel + 5 * ng + 8;
    std::vector<int32_t> npieces((size_t)nsel + 1), poff((size_t)nsel + 1), roff((size_t)cap_p + 1);
    std::vector<double> pieces((size_t)cap_p * 12), rx((size_t)cap_r), ry((size_t)cap_r);
    check(szCornerCuts(g_ctx, ids.data(), nsel, off.data(), grind.data(), npieces.data(), poff.data(), pieces.data(), cap_p, roff.data(), rx.data(), ry.data(), cap_r, nullptr));
    const char *names[] = {"pieces", "rings", "ring_off"};
    plhs[0] = mxCreateStructMatrix(1, (size_t)nsel, 3, names);
    for (int q = 0; q < nsel; ++q) {
      const int P = poff[(size_t)q + 1] - poff[q];
      mxArray *pm = mxCreateDoubleMatrix((size_t)P, P ? 12 : 0, mxREAL);
      const int r0 = roff[(size_t)poff[q]], M = roff[(size_t)poff[q] + P] - r0;
      mxArray *rings = mxCreateDoubleMatrix(M ? 2 : 0, (size_t)M, mxREAL), *ro = mxCreateDoubleMatrix(1, (size_t)P + 1, mxREAL);
      for (int k = 0; k < P; ++k) for (int w = 0; w < 12; ++w) mxGetPr(pm)[(size_t)w * P + k] = pieces[((size_t)poff[q] + k) * 12 + w];
      for (int v = 0; v < M; ++v) { mxGetPr(rings)[2 * v] = rx[(size_t)r0 + v]; mxGetPr(rings)[2 * v + 1] = ry[(size_t)r0 + v]; }
      for (int k = 0; k <= P; ++k) mxGetPr(ro)[k] = roff[(size_t)poff[q] + k] - r0;
      mxSetFieldByNumber(plhs[0], (size_t)q, 0, pm);
      mxSetFieldByNumber(plhs[0], (size_t)q, 1, rings);
      mxSetFieldByNumber(plhs[0], (size_t)q, 2, ro);
    }
    return;
  }
  if (!strcmp(cmd, "deform")) {
    if (nrhs < 2) mexErrMsgTxt("D = subzero_step('deform', ids)");
    const std::vector<int32_t> ids = parse_ids(prhs[1], n);
    const int nsel = (int)ids.size();
    std::vector<int32_t> chg((size_t)nsel + 1), roff((size_t)nsel + 1);
    std::vector<double> X((size_t)nsel + 1), Y((size_t)nsel + 1), A((size_t)nsel + 1);
    int64_t cap = 4 * (int64_t)g_soa.vert_off[n] + 64 * (int64_t)nsel + 64;
    std::vector<double> cx((size_t)cap), cy((size_t)cap);
    check(szFractureDeform(g_ctx, ids.data(), nsel, chg.data(), X.data(), Y.data(), A.data(), roff.data(), cx.data(), cy.data(), cap));
    const char *names[] = {"changed", "Xi", "Yi", "area", "c_alpha"};
    plhs[0] = mxCreateStructMatrix(1, (size_t)nsel, 5, names);
    for (int q = 0; q < nsel; ++q) {
      const int m = roff[q + 1] - roff[q];
      mxSetFieldByNumber(plhs[0], (size_t)q, 0, mxCreateDoubleScalar(chg[q]));
      mxSetFieldByNumber(plhs[0], (size_t)q, 1, mxCreateDoubleScalar(X[q]));
      mxSetFieldByNumber(plhs[0], (size_t)q, 2, mxCreateDoubleScalar(Y[q]));
      mxSetFieldByNumber(plhs[0], (size_t)q, 3, mxCreateDoubleScalar(A[q]));
      mxArray *c = mxCreateDoubleMatrix(m ? 2 : 0, (size_t)m, mxREAL);
      for (int k = 0; k < m; ++k) { mxGetPr(c)[2 * k] = cx[(size_t)roff[q] + k]; mxGetPr(c)[2 * k + 1] = cy[(size_t)roff[q] + k]; }
      mxSetFieldByNumber(plhs[0], (size_t)q, 4, c);
    }
    return;
  }
  if (!strcmp(cmd, "potential")) {
    if (nrhs < 2) mexErrMsgTxt("Pt = subzero_step('potential', ids)");
    const std::vector<int32_t> ids = parse_ids(prhs[1], n);
    const int nsel = (int)ids.size();
    std::vector<int32_t> off((size_t)nsel + 1), list;
    int64_t cap = 64 * (int64_t)nsel + 64;
    for (;;) {
      list.resize((size_t)cap);
      const int rc = szPotentialInteractions(g_ctx, ids.data(), nsel, off.data(), list.data(), cap);
      if (rc == SZ_ERR_CAPACITY) { cap *= 4; continue; }
      check(rc);
      break;
    }
    const char *names[] = {"floeNum"};
    plhs[0] = mxCreateStructMatrix(1, (size_t)nsel, 1, names);
    for (int q = 0; q < nsel; ++q) {
      const int m = off[q + 1] - off[q];
      mxArray *v = mxCreateDoubleMatrix(1, (size_t)m, mxREAL);
      for (int k = 0; k < m; ++k) mxGetPr(v)[k] = list[(size_t)off[q] + k];
      mxSetFieldByNumber(plhs[0], (size_t)q, 0, v);
    }
    return;
  }
  if (!strcmp(cmd, "eulerian")) {
    if (nrhs < 6) mexErrMsgTxt("eul = subzero_step('eulerian', Nx, Ny, Nb, c2_boundary, PERIODIC)");
    const int Nx = (int)mxGetScalar(prhs[1]), Ny = (int)mxGetScalar(prhs[2]);
    if (Nx < 1 || Ny < 1) mexErrMsgTxt("subzero_step('eulerian'): Nx, Ny must be positive");
    const double *cb = doubles(prhs[4], 10, "c2_boundary", -1);
    const size_t cells = (size_t)Nx * Ny;
    std::vector<double> out(18 * cells);
    check(szEulerianData(g_ctx, Nx, Ny, (int)mxGetScalar(prhs[3]), cb, mxGetScalar(prhs[5]) != 0, out.data(), nullptr));
    const char *names[] = {"u", "v", "du", "dv", "stress", "stressxx", "stressyx", "stressxy", "stressyy", "strainux", "strainvx",
                           "strainuy", "strainvy", "c", "Over", "Mtot", "area", "h"};
    plhs[0] = mxCreateStructMatrix(1, 1, 18, names);
    for (int f = 0; f < 18; ++f) {
      mxArray *a = mxCreateDoubleMatrix((size_t)Ny, (size_t)Nx, mxREAL);
      memcpy(mxGetPr(a), out.data() + (size_t)f * cells, cells * 8);
      mxSetFieldByNumber(plhs[0], 0, f, a);
    }
    return;
  }
  if (!strcmp(cmd, "advance")) {
    // one timestep with NOTHING written back: the state stays on the device; what the MATLAB tail
    // of floe_interactions_all.m acts on comes back -- kill / transfer (:467-501) and the floes that died
    szStepParams p;
    step_params(nrhs - 1, prhs + 1, p);
    szStepStats stats; memset(&stats, 0, sizeof(stats));
    const int rc = szStep(g_ctx, &p, &stats);
    if (rc == SZ_STEP_NEEDS_MC) {
      std::vector<int32_t> ids((size_t)n);
      int64_t cnt = 0;
      check(szMcEmptyDownload(g_ctx, ids.data(), n, &cnt));
      if (nlhs < 3) mexErrMsgTxt("subzero_step('advance'): floes need new Monte-Carlo points (calc_trajectory.m:100-110); call with three outputs to receive the list");
      plhs[0] = mxCreateDoubleMatrix(1, 0, mxREAL); plhs[1] = mxCreateDoubleMatrix(1, 0, mxREAL);
      plhs[2] = row_of(ids, (size_t)cnt);
      if (nlhs > 3) plhs[3] = mxCreateDoubleMatrix(1, 0, mxREAL);
      if (nlhs > 4) plhs[4] = ghost_struct(-1);
      return;
    }
    check(rc);
    g_history_ahead = true; g_state_ahead = true;
    // only alive / kill / transfer cross PCIe
    szFloeState s; memset(&s, 0, sizeof(s));
    s.n = g_soa.n; s.nb = g_soa.nb; s.n_stress = g_soa.n_stress; s.n_mc = g_soa.n_mc;
    s.alive = g_soa.alive.data(); s.kill = g_soa.kill.data(); s.transfer = g_soa.transfer.data();
    check(szFloesDownload(g_ctx, &s, SZ_SYNC_DYNAMIC));
    const int N = (int)stats.n_ext;
    if (nlhs > 0) plhs[0] = row_of(g_soa.kill, (size_t)N);
    if (nlhs > 1) plhs[1] = row_of(g_soa.transfer, (size_t)N);
    if (nlhs > 2) plhs[2] = mxCreateDoubleMatrix(1, 0, mxREAL);
    if (nlhs > 3) { std::vector<int32_t> dead; for (int i = 0; i < n; ++i) if (!g_soa.alive[i]) dead.push_back(i + 1); plhs[3] = row_of(dead, dead.size()); }
    if (nlhs > 4) plhs[4] = ghost_struct(N - n);
    return;
  }
  if (strcmp(cmd, "step")) mexErrMsgTxt("subzero_step: unknown command");
  if (nrhs < 12) mexErrMsgTxt("subzero_step('step', Floe, floebound, ocean, winds, c2_boundary, dt, HFo, Modulus, doInt, COLLISION, PERIODIC [, mu, nu])");
  if (nlhs < 1) mexErrMsgTxt("subzero_step('step'): at least one output (Floe) required");
  if ((int)mxGetNumberOfElements(prhs[1]) != n) mexErrMsgTxt("subzero_step('step'): Floe has a different length than the resident array ('resync' first)");
  szStepParams p;
  step_params(nrhs - 2, prhs + 2, p);
  szStepStats stats; memset(&stats, 0, sizeof(stats));
  const int rc = szStep(g_ctx, &p, &stats);
  if (rc == SZ_STEP_NEEDS_MC) {
    std::vector<int32_t> ids((size_t)n);
    int64_t cnt = 0;
    check(szMcEmptyDownload(g_ctx, ids.data(), n, &cnt));
    plhs[0] = mxDuplicateArray(prhs[1]);
    if (nlhs > 1) plhs[1] = mxCreateDoubleMatrix(1, 0, mxREAL);
    if (nlhs > 2) plhs[2] = mxCreateDoubleMatrix(1, 0, mxREAL);
    if (nlhs > 3) { const char *names[] = {"src", "parent", "Xi", "Yi"}; plhs[3] = mxCreateStructMatrix(1, 1, 4, names); for (int k = 0; k < 4; ++k) mxSetFieldByNumber(plhs[3], 0, k, mxCreateDoubleMatrix(1, 0, mxREAL)); }
    if (nlhs > 4) { plhs[4] = mxCreateDoubleMatrix(1, (size_t)cnt, mxREAL); for (int64_t k = 0; k < cnt; ++k) mxGetPr(plhs[4])[k] = ids[(size_t)k]; }
    else mexErrMsgTxt("subzero_step('step'): floes need new Monte-Carlo points (calc_trajectory.m:100-110); call with five outputs to receive the list");
    return;
  }
  check(rc);
  g_history_ahead = true; g_state_ahead = false;   // (every dynamic field is written back below)
  szFloeState s = g_soa.view();
  check(szFloesDownload(g_ctx, &s, SZ_SYNC_DYNAMIC));
  std::vector<int32_t> off((size_t)n + 1);
  std::vector<double> rows((size_t)(stats.n_rows > 0 ? stats.n_rows : 1) * 7);
  check(szInteractionsDownload(g_ctx, off.data(), rows.data(), (int64_t)(rows.size() / 7)));
  plhs[0] = mxDuplicateArray(prhs[1]);
  mxArray *F = plhs[0];
  unpack_state(F, g_soa, nullptr, 0);
  unpack_rings(F, g_soa, nullptr, 0, false);
  for (int i = 0; i < n; ++i) {
    const int k0 = off[i], k1 = off[i + 1], K = k1 - k0;
    mxArray *in = mxCreateDoubleMatrix(K, K ? 7 : 0, mxREAL);   // K x 7, column-major ([] when the floe touched nothing)
    for (int r = 0; r < K; ++r) for (int c = 0; c < 7; ++c) mxGetPr(in)[(size_t)c * K + r] = rows[(size_t)7 * (k0 + r) + c];
    put_field(F, (size_t)i, "interactions", in);
  }
  put_gpu_index(F, n);
  const int N = (int)stats.n_ext;
  if (nlhs > 1) { plhs[1] = mxCreateDoubleMatrix(1, N, mxREAL); for (int i = 0; i < N; ++i) mxGetPr(plhs[1])[i] = g_soa.kill[i]; }
  if (nlhs > 2) { plhs[2] = mxCreateDoubleMatrix(1, N, mxREAL); for (int i = 0; i < N; ++i) mxGetPr(plhs[2])[i] = g_soa.transfer[i]; }
  if (nlhs > 3) {
    const int G = N - n;
    std::vector<int32_t> gs((size_t)G + 1), gp((size_t)G + 1);
    std::vector<double> gx((size_t)G + 1), gy((size_t)G + 1);
    int64_t cnt = 0;
    check(szGhostsDownload(g_ctx, gs.data(), gp.data(), gx.data(), gy.data(), G, &cnt));
    const char *names[] = {"src", "parent", "Xi", "Yi"};
    plhs[3] = mxCreateStructMatrix(1, 1, 4, names);
    mxArray *a[4];
    for (int k = 0; k < 4; ++k) a[k] = mxCreateDoubleMatrix(1, (size_t)cnt, mxREAL);
    for (int64_t g = 0; g < cnt; ++g) { mxGetPr(a[0])[g] = gs[(size_t)g]; mxGetPr(a[1])[g] = gp[(size_t)g]; mxGetPr(a[2])[g] = gx[(size_t)g]; mxGetPr(a[3])[g] = gy[(size_t)g]; }
    for (int k = 0; k < 4; ++k) mxSetFieldByNumber(plhs[3], 0, k, a[k]);
  }
  if (nlhs > 4) plhs[4] = mxCreateDoubleMatrix(1, 0, mxREAL);
}
