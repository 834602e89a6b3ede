// TEST INFRASTRUCTURE ONLY -- see mex.h in this directory.  The runtime behind the test mex.h plus the entry points
// tests/mexharness.py uses to call a gateway's mexFunction and to look at the arrays it returns.
#include "mex.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdexcept>
#include <string>
#include <vector>

struct mxArray_tag {
  mxClassID cls = mxDOUBLE_CLASS;
  std::vector<mwSize> dims;
  std::vector<char> data;                       // numeric / logical / char payload, column-major
  std::vector<std::string> fields;              // struct arrays
  std::vector<mxArray *> values;                // [element][field]
};

namespace {
size_t elem_size(mxClassID c) {
  switch (c) {
    case mxDOUBLE_CLASS: case mxINT64_CLASS: case mxUINT64_CLASS: return 8;
    case mxSINGLE_CLASS: case mxINT32_CLASS: case mxUINT32_CLASS: return 4;
    case mxINT16_CLASS: case mxUINT16_CLASS: case mxCHAR_CLASS: return 2;
    case mxLOGICAL_CLASS: case mxINT8_CLASS: case mxUINT8_CLASS: return 1;
    default: return 0;
  }
}
size_t numel(const mxArray *a) { size_t n = 1; for (mwSize d : a->dims) n *= d; return a->dims.empty() ? 0 : n; }
mxArray *make(mxClassID cls, mwSize ndim, const mwSize *dims) {
  mxArray *a = new mxArray_tag();
  a->cls = cls;
  a->dims.assign(dims, dims + ndim);
  if (a->dims.size() < 2) a->dims.resize(2, 1);
  a->data.assign(numel(a) * elem_size(cls), 0);
  return a;
}
std::vector<void (*)(void)> g_atexit;
long g_live = 0;   // arrays alive (a leak check for the gateways: see stub_live_arrays)
}

extern "C" {

mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID cls, mxComplexity) { ++g_live; return make(cls, ndim, dims); }
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity c) { mwSize d[2] = {m, n}; return mxCreateNumericArray(2, d, cls, c); }
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity c) { return mxCreateNumericMatrix(m, n, mxDOUBLE_CLASS, c); }
mxArray *mxCreateDoubleScalar(double v) { mxArray *a = mxCreateDoubleMatrix(1, 1, mxREAL); *(double *)a->data.data() = v; return a; }
mxArray *mxCreateLogicalMatrix(mwSize m, mwSize n) { return mxCreateNumericMatrix(m, n, mxLOGICAL_CLASS, mxREAL); }
mxArray *mxCreateLogicalScalar(bool v) { mxArray *a = mxCreateLogicalMatrix(1, 1); a->data[0] = v ? 1 : 0; return a; }
mxArray *mxCreateString(const char *s) {
  mwSize d[2] = {1, strlen(s)};
  ++g_live;
  mxArray *a = make(mxCHAR_CLASS, 2, d);
  for (size_t k = 0; k < d[1]; ++k) ((mxChar *)a->data.data())[k] = (mxChar)(unsigned char)s[k];
  return a;
}
mxArray *mxCreateStructArray(mwSize ndim, const mwSize *dims, int nfields, const char **names) {
  ++g_live;
  mxArray *a = make(mxSTRUCT_CLASS, ndim, dims);
  for (int k = 0; k < nfields; ++k) a->fields.push_back(names[k]);
  a->values.assign(numel(a) * (size_t)nfields, nullptr);
  return a;
}
mxArray *mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char **names) { mwSize d[2] = {m, n}; return mxCreateStructArray(2, d, nfields, names); }
mxArray *mxDuplicateArray(const mxArray *a) {
  if (!a) return nullptr;
  ++g_live;
  mxArray *b = new mxArray_tag(*a);
  for (mxArray *&v : b->values) if (v) v = mxDuplicateArray(v);
  return b;
}
void mxDestroyArray(mxArray *a) {
  if (!a) return;
  for (mxArray *v : a->values) mxDestroyArray(v);
  --g_live;
  delete a;
}
void *mxMalloc(size_t n) { return malloc(n ? n : 1); }
void *mxCalloc(size_t n, size_t sz) { return calloc(n ? n : 1, sz ? sz : 1); }
void mxFree(void *p) { free(p); }

size_t mxGetNumberOfElements(const mxArray *a) { return numel(a); }
size_t mxGetM(const mxArray *a) { return a->dims[0]; }
size_t mxGetN(const mxArray *a) { size_t n = 1; for (size_t k = 1; k < a->dims.size(); ++k) n *= a->dims[k]; return n; }
mwSize mxGetNumberOfDimensions(const mxArray *a) { return a->dims.size(); }
const mwSize *mxGetDimensions(const mxArray *a) { return a->dims.data(); }
mxClassID mxGetClassID(const mxArray *a) { return a->cls; }
bool mxIsStruct(const mxArray *a) { return a->cls == mxSTRUCT_CLASS; }
bool mxIsDouble(const mxArray *a) { return a->cls == mxDOUBLE_CLASS; }
bool mxIsInt64(const mxArray *a) { return a->cls == mxINT64_CLASS; }
bool mxIsInt32(const mxArray *a) { return a->cls == mxINT32_CLASS; }
bool mxIsChar(const mxArray *a) { return a->cls == mxCHAR_CLASS; }
bool mxIsLogical(const mxArray *a) { return a->cls == mxLOGICAL_CLASS; }
bool mxIsNumeric(const mxArray *a) { return a->cls >= mxDOUBLE_CLASS; }
bool mxIsEmpty(const mxArray *a) { return numel(a) == 0; }
bool mxIsComplex(const mxArray *) { return false; }

double *mxGetPr(const mxArray *a) { return (a->cls == mxDOUBLE_CLASS && !a->data.empty()) ? (double *)a->data.data() : nullptr; }
void *mxGetData(const mxArray *a) { return a->data.empty() ? nullptr : (void *)a->data.data(); }
mxLogical *mxGetLogicals(const mxArray *a) { return (a->cls == mxLOGICAL_CLASS && !a->data.empty()) ? (mxLogical *)a->data.data() : nullptr; }
double mxGetScalar(const mxArray *a) {
  if (!a || a->data.empty()) return 0.0;
  const void *p = a->data.data();
  switch (a->cls) {
    case mxDOUBLE_CLASS: return *(const double *)p;
    case mxSINGLE_CLASS: return *(const float *)p;
    case mxINT64_CLASS: return (double)*(const int64_t *)p;
    case mxUINT64_CLASS: return (double)*(const uint64_t *)p;
    case mxINT32_CLASS: return *(const int32_t *)p;
    case mxUINT32_CLASS: return *(const uint32_t *)p;
    case mxINT16_CLASS: return *(const int16_t *)p;
    case mxUINT16_CLASS: case mxCHAR_CLASS: return *(const uint16_t *)p;
    case mxINT8_CLASS: return *(const int8_t *)p;
    case mxUINT8_CLASS: case mxLOGICAL_CLASS: return *(const uint8_t *)p;
    default: return 0.0;
  }
}
int mxGetString(const mxArray *a, char *buf, mwSize buflen) {
  if (!a || a->cls != mxCHAR_CLASS || buflen == 0) return 1;
  const size_t n = numel(a);
  const size_t m = n < buflen - 1 ? n : buflen - 1;
  for (size_t k = 0; k < m; ++k) buf[k] = (char)((const mxChar *)a->data.data())[k];
  buf[m] = 0;
  return n > buflen - 1 ? 1 : 0;
}
char *mxArrayToString(const mxArray *a) {
  if (!a || a->cls != mxCHAR_CLASS) return nullptr;
  const size_t n = numel(a);
  char *s = (char *)mxMalloc(n + 1);
  mxGetString(a, s, n + 1);
  return s;
}

int mxGetNumberOfFields(const mxArray *a) { return (int)a->fields.size(); }
int mxGetFieldNumber(const mxArray *a, const char *name) {
  for (size_t k = 0; k < a->fields.size(); ++k) if (a->fields[k] == name) return (int)k;
  return -1;
}
const char *mxGetFieldNameByNumber(const mxArray *a, int k) { return (k >= 0 && (size_t)k < a->fields.size()) ? a->fields[k].c_str() : nullptr; }
mxArray *mxGetFieldByNumber(const mxArray *a, mwIndex i, int k) {
  if (!a || a->cls != mxSTRUCT_CLASS || k < 0 || (size_t)k >= a->fields.size() || i >= numel(a)) return nullptr;
  return a->values[i * a->fields.size() + (size_t)k];
}
mxArray *mxGetField(const mxArray *a, mwIndex i, const char *name) { return a ? mxGetFieldByNumber(a, i, mxGetFieldNumber(a, name)) : nullptr; }
void mxSetFieldByNumber(mxArray *a, mwIndex i, int k, mxArray *v) {
  if (!a || a->cls != mxSTRUCT_CLASS || k < 0 || (size_t)k >= a->fields.size() || i >= numel(a)) return;
  a->values[i * a->fields.size() + (size_t)k] = v;
}
void mxSetField(mxArray *a, mwIndex i, const char *name, mxArray *v) { mxSetFieldByNumber(a, i, mxGetFieldNumber(a, name), v); }
int mxAddField(mxArray *a, const char *name) {
  if (!a || a->cls != mxSTRUCT_CLASS) return -1;
  const size_t nf = a->fields.size(), n = numel(a);
  std::vector<mxArray *> nv(n * (nf + 1), nullptr);
  for (size_t i = 0; i < n; ++i) for (size_t k = 0; k < nf; ++k) nv[i * (nf + 1) + k] = a->values[i * nf + k];
  a->values.swap(nv);
  a->fields.push_back(name);
  return (int)nf;
}

void mexErrMsgTxt(const char *msg) { throw std::runtime_error(msg ? msg : ""); }
void mexErrMsgIdAndTxt(const char *, const char *fmt, ...) {
  char buf[1024];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  throw std::runtime_error(buf);
}
void mexWarnMsgTxt(const char *msg) { fprintf(stderr, "Warning: %s\n", msg); }
int mexPrintf(const char *fmt, ...) { va_list ap; va_start(ap, fmt); int r = vprintf(fmt, ap); va_end(ap); return r; }
int mexAtExit(void (*fn)(void)) { g_atexit.push_back(fn); return 0; }
void mexLock(void) {}

// ---- the harness side (tests/mexharness.py) ----
typedef void (*mexfn_t)(int, mxArray **, int, const mxArray **);
// calls a gateway; 0, or 1 with the text of its mexErrMsgTxt in errbuf (MATLAB would raise it as an error)
int stub_call(mexfn_t fn, int nlhs, mxArray **plhs, int nrhs, const mxArray **prhs, char *errbuf, int errcap) {
  try { fn(nlhs, plhs, nrhs, prhs); }
  catch (const std::exception &e) { snprintf(errbuf, (size_t)errcap, "%s", e.what()); return 1; }
  return 0;
}
void stub_run_atexit(void) { for (auto fn : g_atexit) fn(); g_atexit.clear(); }
long stub_live_arrays(void) { return g_live; }
int stub_class(const mxArray *a) { return (int)a->cls; }
void stub_set_dims(mxArray *a, int ndim, const mwSize *dims) { a->dims.assign(dims, dims + ndim); }

} // extern "C"
