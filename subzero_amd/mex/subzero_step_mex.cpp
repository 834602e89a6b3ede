// subzero_step_mex.cpp -- MATLAB gateway of the B2 surface (floe_interactions_all.m:16-285 in one call).
// NOT compiled in this repository's CI (needs mex.h).  Build on a MATLAB host with
//   mex -R2017b subzero_step_mex.cpp -output subzero_step -I<repo>/include -L<repo>/subzero_amd/csrc -lsubzero_hip
//
//   subzero_step('upload', Floe, Nb)        Floe struct array (initialize_floe_values.m:12-52) -> GPU
//   [Floe, kill, transfer] = subzero_step('step', Floe, floebound, ocean, winds, c2_boundary, dt, HFo, Modulus,
//                                         doIntFlag, COLLISION, PERIODIC)
//       advances the device-resident floes by one step and writes the dynamic fields, `interactions`,
//       `collision_force`, `collision_torque`, `OverlapArea`, `Stress`, `strain`, `alive` back into Floe.
// The struct <-> SoA marshalling below is the only work done on the MATLAB thread.
#include "mex.h"
#include <stdint.h>
#include <string.h>
#include <vector>
#include "subzero_hip.h"

static szContext *g_ctx = nullptr;
static void at_exit(void) { if (g_ctx) { szDestroy(g_ctx); g_ctx = nullptr; } }

struct Soa {
  int n = 0;
  std::vector<double> Xi, Yi, alpha_i, Ui, Vi, ksi_ice, h, area, mass, inertia_moment, rmax, dXi_p, dYi_p, dUi_p,
      dVi_p, dalpha_i_p, dksi_ice_p, FxOA, FyOA, torqueOA, c0x, c0y, cax, cay, StressH, Stress, strain, mcX, mcY,
      collision_force, collision_torque, OverlapArea;
  std::vector<int32_t> alive, vert_off, StressCount, kill, transfer;
  std::vector<uint8_t> mcA;
  szFloeState view(int nb) {
    szFloeState s; memset(&s, 0, sizeof(s));
    s.n = n; s.nb = nb; s.n_stress = 1000; s.n_mc = 1000;
#define P(f) s.f = f.data();
    P(Xi) P(Yi) P(alpha_i) P(Ui) P(Vi) P(ksi_ice) P(h) P(area) P(mass) P(inertia_moment) P(rmax) P(dXi_p) P(dYi_p)
    P(dUi_p) P(dVi_p) P(dalpha_i_p) P(dksi_ice_p) P(FxOA) P(FyOA) P(torqueOA) P(alive) P(vert_off) P(c0x) P(c0y)
    P(cax) P(cay) P(StressH) P(StressCount) P(Stress) P(strain) P(mcX) P(mcY) P(mcA) P(collision_force)
    P(collision_torque) P(OverlapArea) P(kill) P(transfer)
#undef P
    return s;
  }
};
static Soa g_soa;
static int g_nb = 0;

static double scalar_field(const mxArray *F, int i, const char *name) {
  const mxArray *a = mxGetField(F, i, name);
  return (a && mxGetNumberOfElements(a)) ? mxGetScalar(a) : 0.0;
}

static void pack(const mxArray *F, Soa &S) {
  const int n = (int)mxGetNumberOfElements(F);
  S.n = n;
#define SC(f) S.f.resize(n); for (int i = 0; i < n; ++i) S.f[i] = scalar_field(F, i, #f);
  SC(Xi) SC(Yi) SC(alpha_i) SC(Ui) SC(Vi) SC(ksi_ice) SC(h) SC(area) SC(mass) SC(inertia_moment) SC(rmax) SC(dXi_p)
  SC(dYi_p) SC(dUi_p) SC(dVi_p) SC(dalpha_i_p) SC(dksi_ice_p) SC(FxOA) SC(FyOA) SC(torqueOA)
#undef SC
  S.alive.resize(n); S.StressCount.resize(n); S.vert_off.assign(n + 1, 0);
  for (int i = 0; i < n; ++i) {
    S.alive[i] = (int)scalar_field(F, i, "alive");
    S.StressCount[i] = (int)scalar_field(F, i, "StressCount");
    S.vert_off[i + 1] = S.vert_off[i] + (int)mxGetN(mxGetField(F, i, "c0")); // c0 is 2 x (n+1)
  }
  const int nv = S.vert_off[n];
  S.c0x.resize(nv); S.c0y.resize(nv); S.cax.resize(nv); S.cay.resize(nv);
  S.StressH.assign((size_t)n * 4000, 0); S.Stress.assign(4 * (size_t)n, 0); S.strain.assign(4 * (size_t)n, 0);
  S.mcX.resize((size_t)n * 1000); S.mcY.resize((size_t)n * 1000); S.mcA.resize((size_t)n * 1000);
  for (int i = 0; i < n; ++i) {
    const double *c0 = mxGetPr(mxGetField(F, i, "c0")), *ca = mxGetPr(mxGetField(F, i, "c_alpha"));
    for (int k = S.vert_off[i]; k < S.vert_off[i + 1]; ++k) {
      int q = k - S.vert_off[i];
      S.c0x[k] = c0[2 * q]; S.c0y[k] = c0[2 * q + 1]; S.cax[k] = ca[2 * q]; S.cay[k] = ca[2 * q + 1];
    }
    memcpy(&S.StressH[(size_t)i * 4000], mxGetPr(mxGetField(F, i, "StressH")), 4000 * sizeof(double)); // 2x2x1000
    memcpy(&S.mcX[(size_t)i * 1000], mxGetPr(mxGetField(F, i, "X")), 1000 * sizeof(double));
    memcpy(&S.mcY[(size_t)i * 1000], mxGetPr(mxGetField(F, i, "Y")), 1000 * sizeof(double));
    const mxLogical *A = mxGetLogicals(mxGetField(F, i, "A"));
    for (int q = 0; q < 1000; ++q) S.mcA[(size_t)i * 1000 + q] = A[q] ? 1 : 0;
  }
  S.collision_force.assign(2 * (size_t)n, 0); S.collision_torque.assign(n, 0); S.OverlapArea.assign(n, 0);
  S.kill.assign(4 * (size_t)n, 0); S.transfer.assign(4 * (size_t)n, 0);
}

static void set_scalar(mxArray *F, int i, const char *name, double v) {
  mxArray *a = mxGetField(F, i, name);
  if (a && mxIsDouble(a) && mxGetNumberOfElements(a) == 1) { *mxGetPr(a) = v; return; }
  mxSetField(F, i, name, mxCreateDoubleScalar(v));
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgTxt("subzero_step(command, ...)");
  char cmd[16]; mxGetString(prhs[0], cmd, sizeof(cmd));
  if (!g_ctx) {
    if (szCreate(&g_ctx, 0) != SZ_OK) mexErrMsgTxt("subzero_hip: no usable gfx950 GPU (there is no CPU fallback).");
    mexAtExit(at_exit);
  }
  if (!strcmp(cmd, "upload")) {
    if (nrhs < 3) mexErrMsgTxt("subzero_step('upload', Floe, Nb)");
    pack(prhs[1], g_soa);
    g_nb = (int)mxGetScalar(prhs[2]);
    szFloeState s = g_soa.view(g_nb);
    if (szFloesUpload(g_ctx, &s) != SZ_OK) mexErrMsgTxt(szLastError(g_ctx));
    return;
  }
  if (strcmp(cmd, "step")) mexErrMsgTxt("subzero_step: unknown command");
  if (nrhs < 12) mexErrMsgTxt("subzero_step('step', Floe, floebound, ocean, winds, c2_boundary, dt, HFo, Modulus, doInt, COLLISION, PERIODIC)");
  const mxArray *fb = prhs[2], *oc = prhs[3], *wi = prhs[4];
  szStepParams p; memset(&p, 0, sizeof(p));
  memcpy(p.c2_boundary, mxGetPr(prhs[5]), 10 * sizeof(double));
  p.dt = mxGetScalar(prhs[6]);
  { const mxArray *H = prhs[7]; size_t m = mxGetNumberOfElements(H); double s = 0; for (size_t k = 0; k < m; ++k) s += mxGetPr(H)[k]; p.HFo = s / (double)m; } // mean(HFo(:))
  p.Modulus = mxGetScalar(prhs[8]); p.mu = 0.2; p.nu = 0.3;
  p.doInt = (int)mxGetScalar(prhs[9]); p.COLLISION = (int)mxGetScalar(prhs[10]); p.PERIODIC = (int)mxGetScalar(prhs[11]);
  p.fb_h = scalar_field(fb, 0, "h"); p.fb_area = scalar_field(fb, 0, "area"); p.fb_Xi = scalar_field(fb, 0, "Xi");
  p.fb_Yi = scalar_field(fb, 0, "Yi"); p.fb_Ui = scalar_field(fb, 0, "Ui"); p.fb_Vi = scalar_field(fb, 0, "Vi");
  p.fb_ksi_ice = scalar_field(fb, 0, "ksi_ice");
  const mxArray *Xo = mxGetField(oc, 0, "Xo"), *Yo = mxGetField(oc, 0, "Yo");
  p.nxo = (int)mxGetNumberOfElements(Xo); p.nyo = (int)mxGetNumberOfElements(Yo);
  p.Xo = mxGetPr(Xo); p.Yo = mxGetPr(Yo);
  p.Uocn = mxGetPr(mxGetField(oc, 0, "Uocn")); p.Vocn = mxGetPr(mxGetField(oc, 0, "Vocn"));
  p.Uwind = mxGetPr(mxGetField(wi, 0, "u")); p.Vwind = mxGetPr(mxGetField(wi, 0, "v"));
  p.fCoriolis = scalar_field(oc, 0, "fCoriolis"); p.turn_angle = scalar_field(oc, 0, "turn_angle");
  if (szStep(g_ctx, &p, nullptr) != SZ_OK) mexErrMsgTxt(szLastError(g_ctx));
  szFloeState s = g_soa.view(g_nb);
  if (szFloesDownload(g_ctx, &s, SZ_SYNC_DYNAMIC) != SZ_OK) mexErrMsgTxt(szLastError(g_ctx));
  const int n = g_soa.n;
  std::vector<int32_t> off(n + 1);
  std::vector<double> rows((size_t)64 * n * 7 + 7168);
  if (szInteractionsDownload(g_ctx, off.data(), rows.data(), (int64_t)(rows.size() / 7)) != SZ_OK) mexErrMsgTxt(szLastError(g_ctx));
  plhs[0] = mxDuplicateArray(prhs[1]);
  mxArray *F = plhs[0];
  for (int i = 0; i < n; ++i) {
#define W(f) set_scalar(F, i, #f, g_soa.f[i]);
    W(Xi) W(Yi) W(alpha_i) W(Ui) W(Vi) W(ksi_ice) W(h) W(mass) W(inertia_moment) W(dXi_p) W(dYi_p) W(dUi_p) W(dVi_p)
    W(dalpha_i_p) W(dksi_ice_p) W(FxOA) W(FyOA) W(torqueOA) W(collision_torque) W(OverlapArea)
#undef W
    set_scalar(F, i, "alive", g_soa.alive[i]);
    set_scalar(F, i, "StressCount", g_soa.StressCount[i]);
    int nv = g_soa.vert_off[i + 1] - g_soa.vert_off[i];
    mxArray *ca = mxCreateDoubleMatrix(2, nv, mxREAL);
    for (int k = 0; k < nv; ++k) { mxGetPr(ca)[2 * k] = g_soa.cax[g_soa.vert_off[i] + k]; mxGetPr(ca)[2 * k + 1] = g_soa.cay[g_soa.vert_off[i] + k]; }
    mxSetField(F, i, "c_alpha", ca);
    mxArray *cf = mxCreateDoubleMatrix(1, 2, mxREAL);
    mxGetPr(cf)[0] = g_soa.collision_force[2 * i]; mxGetPr(cf)[1] = g_soa.collision_force[2 * i + 1];
    mxSetField(F, i, "collision_force", cf);
    mxArray *st = mxCreateDoubleMatrix(2, 2, mxREAL), *sn = mxCreateDoubleMatrix(2, 2, mxREAL);
    memcpy(mxGetPr(st), &g_soa.Stress[4 * i], 32); memcpy(mxGetPr(sn), &g_soa.strain[4 * i], 32);
    mxSetField(F, i, "Stress", st); mxSetField(F, i, "strain", sn);
    int k0 = off[i], k1 = off[i + 1];
    mxArray *in = mxCreateDoubleMatrix(k1 - k0, 7, mxREAL); // K x 7, column-major
    for (int r = k0; r < k1; ++r) for (int c = 0; c < 7; ++c) mxGetPr(in)[(size_t)c * (k1 - k0) + (r - k0)] = rows[(size_t)7 * r + c];
    mxSetField(F, i, "interactions", in);
  }
  if (nlhs > 1) { plhs[1] = mxCreateDoubleMatrix(1, 4 * n, mxREAL); for (int i = 0; i < 4 * n; ++i) mxGetPr(plhs[1])[i] = g_soa.kill[i]; }
  if (nlhs > 2) { plhs[2] = mxCreateDoubleMatrix(1, 4 * n, mxREAL); for (int i = 0; i < 4 * n; ++i) mxGetPr(plhs[2])[i] = g_soa.transfer[i]; }
}
