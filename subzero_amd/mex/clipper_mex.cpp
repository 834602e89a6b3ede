// clipper_mex.cpp -- drop-in replacement of the reference's `clipper` mex (private/mexclipper.cpp:83-305)
// on top of libsubzero_hip.so.  Build on a MATLAB host with
//   mex -R2017b clipper_mex.cpp -output clipper -I<repo>/include -L<repo>/subzero_amd/csrc -lsubzero_hip
// and put the result on the path before the reference's private/clipper.mexa64.  (This repository has no MATLAB:
// tests/mexstub/ is a working test double of the mx* / mex* calls used here; tests/test_mex_*.py compile this file
// against it and drive it.)
//
// The reference gateway's three call forms and error texts:
//   out = clipper(Subj, Clip, method [, subjFill, clipFill])   -> szClipPathsBatch (mexclipper.cpp:204-305; Subj / Clip
//                                                                 may be struct ARRAYS: one contour per element, :44-62)
//   cw  = clipper(P)                                           -> szOrientation (mexclipper.cpp:134-145)
//   out = clipper(Subj, delta [, 's'|'r'|'m' [, info]])         -> szOffsetPaths (mexclipper.cpp:146-203)
// Subj / Clip: struct arrays with int64 column fields x, y (mexclipper.cpp:28-41); out: 1xK struct array
// with double fields x, y holding integer values (mexclipper.cpp:65-81).
#include "mex.h"
#include <stdint.h>
#include <vector>
#include "subzero_hip.h"

static szContext *g_ctx = nullptr;
static void at_exit(void) { if (g_ctx) { szDestroy(g_ctx); g_ctx = nullptr; } }

// every element of the struct array is one contour (mexclipper.cpp:13-62); off = vertex range of each
static void read_rings(const mxArray *p, std::vector<int64_t> &x, std::vector<int64_t> &y, std::vector<int32_t> &off) {
  if (!mxIsStruct(p)) mexErrMsgTxt("Input needs to be structure.");
  if (!mxGetM(p) || !mxGetN(p)) mexErrMsgTxt("Empty structure.");
  int ix = mxGetFieldNumber(p, "x"), iy = mxGetFieldNumber(p, "y");
  if (ix == -1) mexErrMsgTxt("Input structure must contain a field 'x'.");
  if (iy == -1) mexErrMsgTxt("Input structure must contain a field 'y'.");
  const size_t nrings = mxGetNumberOfElements(p);
  x.clear(); y.clear(); off.assign(1, 0);
  for (size_t r = 0; r < nrings; ++r) {
    const mxArray *fx = mxGetFieldByNumber(p, r, ix), *fy = mxGetFieldByNumber(p, r, iy);
    if (!fx || !mxIsInt64(fx)) mexErrMsgTxt("Structure field 'x' must be of type INT64.");
    if (!fy || !mxIsInt64(fy)) mexErrMsgTxt("Structure field 'y' must be of type INT64.");
    size_t n = mxGetNumberOfElements(fx);
    if (n != mxGetNumberOfElements(fy)) mexErrMsgTxt("Structure fields x and y must be the same length.");
    const int64_t *px = (const int64_t *)mxGetData(fx), *py = (const int64_t *)mxGetData(fy);
    x.insert(x.end(), px, px + n); y.insert(y.end(), py, py + n);
    off.push_back((int32_t)x.size());
  }
}
static void need_context() {
  if (g_ctx) return;
  if (szCreate(&g_ctx, 0) != SZ_OK) mexErrMsgTxt("subzero_hip: no usable gfx950 GPU (there is no CPU fallback).");
  mexAtExit(at_exit);
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs == 0) { mexPrintf("OutPol = clipper(RefPol, ClipPol, Method, [RefF], [ClipF]);  (MI355X build)\n"); return; }
  if (nlhs != 1) mexErrMsgTxt("One output required.");
  std::vector<int64_t> sx, sy, cx, cy;
  std::vector<int32_t> soff, coff;
  if (nrhs == 1) { // orientation of every contour (mexclipper.cpp:134-145): host arithmetic, no GPU needed
    read_rings(prhs[0], sx, sy, soff);
    const size_t nr = soff.size() - 1;
    plhs[0] = mxCreateDoubleMatrix(nr, 1, mxREAL);      // (a double column of 0 / 1, as the reference returns)
    for (size_t r = 0; r < nr; ++r) {
      int o = 0;
      szOrientation(sx.data() + soff[r], sy.data() + soff[r], soff[r + 1] - soff[r], &o);
      mxGetPr(plhs[0])[r] = (double)(o != 0);
    }
    return;
  }
  // offset mode (mexclipper.cpp:146-203: clipper(P, delta [, 's'|'r'|'m' [, info]]))
  if (nrhs == 2 || (nrhs >= 3 && nrhs <= 4 && mxIsChar(prhs[2]))) {
    if (!mxIsStruct(prhs[0]) || mxGetM(prhs[0]) != 1 || mxGetN(prhs[0]) != 1)
      mexErrMsgTxt("First input must be a scalar polygon structure (with fields .x and .y).");
    if (!mxIsDouble(prhs[1]) || mxGetM(prhs[1]) != 1 || mxGetN(prhs[1]) != 1)
      mexErrMsgTxt("Second input must be either a structure (a second polygon) or a scalar double (outset Delta).");
    const double delta = mxGetScalar(prhs[1]);
    double info = 0.0;
    char type = 'm';
    if (nrhs >= 3) {
      char *t = mxArrayToString(prhs[2]);
      if (!t) mexErrMsgTxt("Unable to access the MiterType string.");
      type = t[0] >= 'A' && t[0] <= 'Z' ? (char)(t[0] - 'A' + 'a') : t[0];
      mxFree(t);
      if (nrhs >= 4) {
        if (!mxIsDouble(prhs[3]) || mxGetM(prhs[3]) != 1 || mxGetN(prhs[3]) != 1) mexErrMsgTxt("The MITERINFO (fourth argument) must be a scalar double.");
        info = mxGetScalar(prhs[3]);
        if (info <= 0.0) mexErrMsgTxt("The MITERINFO must be a positive double.");
      }
    }
    int join;
    if (type == 'r') { join = 1; if (nrhs < 4) mexErrMsgTxt("The MITERINFO (fourth argument) must specify the arc tolerance for round joints."); }
    else if (type == 's') { join = 0; info = 0.0; }
    else if (type == 'm') { join = 2; if (nrhs >= 4 && info < 2.0) mexWarnMsgTxt("MiterLimit below 2.0 is ignored"); }
    else { join = 2; mexErrMsgTxt("Invalid MiterType string."); }
    read_rings(prhs[0], sx, sy, soff);
    need_context();
    int64_t cap = 64 * (int64_t)(sx.size() + 16);
    for (;;) {
      std::vector<int64_t> ox((size_t)cap), oy((size_t)cap);
      std::vector<int32_t> roff((size_t)cap + 1);
      int32_t nr = 0, status = 0;
      const int rc = szOffsetPaths(g_ctx, sx.data(), sy.data(), soff.data(), (int)soff.size() - 1, delta, join, info, ox.data(), oy.data(), cap,
                                   roff.data(), cap, &nr, &status);
      if (rc == SZ_ERR_CAPACITY) { cap *= 4; continue; }
      if (rc != SZ_OK) mexErrMsgTxt(szLastError(g_ctx));
      if (status != SZ_CLIP_OK) mexErrMsgTxt("Clipper Error.");
      const char *names[] = {"x", "y"};
      mwSize dims[2] = {1, (mwSize)nr};
      plhs[0] = mxCreateStructArray(2, dims, 2, names);
      for (int r = 0; r < nr; ++r) {
        const int n = roff[r + 1] - roff[r];
        mxArray *ax = mxCreateDoubleMatrix(n, 1, mxREAL), *ay = mxCreateDoubleMatrix(n, 1, mxREAL);
        for (int k = 0; k < n; ++k) { mxGetPr(ax)[k] = (double)ox[roff[r] + k]; mxGetPr(ay)[k] = (double)oy[roff[r] + k]; }
        mxSetFieldByNumber(plhs[0], r, 0, ax);
        mxSetFieldByNumber(plhs[0], r, 1, ay);
      }
      return;
    }
  }
  if (!mxIsDouble(prhs[2]) || mxGetM(prhs[2]) != 1 || mxGetN(prhs[2]) != 1) mexErrMsgTxt("Third input must be scalar.");
  if (!mxIsStruct(prhs[1])) mexErrMsgTxt("second input must be a polygon structure.");
  int method = (int)mxGetScalar(prhs[2]);
  if (method < 0 || method > 3) mexErrMsgTxt("Third input must be 0, 1, 2, or 3.");
  int sf = 0, cf = 0;
  if (nrhs >= 4) {
    if (!mxIsDouble(prhs[3]) || mxGetM(prhs[3]) != 1 || mxGetN(prhs[3]) != 1) mexErrMsgTxt("Fourth input must be scalar if specified.");
    sf = (int)mxGetScalar(prhs[3]); if (sf < 0 || sf > 3) mexErrMsgTxt("Fourth input must be 0, 1, 2, or 3.");
  }
  if (nrhs >= 5) {
    if (!mxIsDouble(prhs[4]) || mxGetM(prhs[4]) != 1 || mxGetN(prhs[4]) != 1) mexErrMsgTxt("Fifth input must be scalar if specified.");
    cf = (int)mxGetScalar(prhs[4]); if (cf < 0 || cf > 3) mexErrMsgTxt("Fifth input must be 0, 1, 2, or 3.");
  }
  read_rings(prhs[0], sx, sy, soff);
  read_rings(prhs[1], cx, cy, coff);
  need_context();                     // (every argument error above is raised without touching the GPU)
  int32_t apair[2] = {0, (int32_t)soff.size() - 1}, bpair[2] = {0, (int32_t)coff.size() - 1};
  int64_t cap = 4 * (int64_t)(sx.size() + cx.size()) + 64;
  for (;;) {
    std::vector<int64_t> ox(cap), oy(cap);
    std::vector<int32_t> roff(cap + 1);
    int32_t proff[2], status = 0;
    int rc = szClipPathsBatch(g_ctx, 1, sx.data(), sy.data(), soff.data(), apair, cx.data(), cy.data(), coff.data(), bpair,
                              method, sf, cf, ox.data(), oy.data(), cap, roff.data(), cap, proff, &status);
    if (rc == SZ_ERR_CAPACITY) { cap *= 4; continue; }
    if (rc != SZ_OK) mexErrMsgTxt(szLastError(g_ctx));
    if (status != SZ_CLIP_OK) mexErrMsgTxt("Clipper Error.");
    const char *names[] = {"x", "y"};
    mwSize dims[2] = {1, (mwSize)proff[1]};
    plhs[0] = mxCreateStructArray(2, dims, 2, names);
    for (int r = 0; r < proff[1]; ++r) {
      int n = roff[r + 1] - roff[r];
      mxArray *ax = mxCreateDoubleMatrix(n, 1, mxREAL), *ay = mxCreateDoubleMatrix(n, 1, mxREAL);
      for (int k = 0; k < n; ++k) { mxGetPr(ax)[k] = (double)ox[roff[r] + k]; mxGetPr(ay)[k] = (double)oy[roff[r] + k]; }
      mxSetFieldByNumber(plhs[0], r, 0, ax);
      mxSetFieldByNumber(plhs[0], r, 1, ay);
    }
    return;
  }
}
