/* TEST INFRASTRUCTURE ONLY -- NOT MathWorks' mex.h.
 *
 * A small, working implementation of the part of MATLAB's C Matrix / MEX API that this repository's two gateways
 * (subzero_amd/mex/clipper_mex.cpp, subzero_step_mex.cpp) call, written from the public documentation of those
 * functions, so that the gateways can be COMPILED and DRIVEN without MATLAB (which this image does not have):
 * tests/mexharness.py builds mxArrays through it, calls mexFunction and reads the results back.  Column-major
 * storage, struct arrays with per-element field values, mexErrMsgTxt that does not return (it throws; the harness
 * catches) -- the behaviours the gateways rely on.  On a MATLAB host the gateways are built with the real mex.h.
 */
#ifndef SUBZERO_TEST_MEX_H
#define SUBZERO_TEST_MEX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef bool mxLogical;
typedef uint16_t mxChar;
typedef enum { mxUNKNOWN_CLASS = 0, mxCELL_CLASS, mxSTRUCT_CLASS, mxLOGICAL_CLASS, mxCHAR_CLASS, mxVOID_CLASS,
               mxDOUBLE_CLASS, mxSINGLE_CLASS, mxINT8_CLASS, mxUINT8_CLASS, mxINT16_CLASS, mxUINT16_CLASS,
               mxINT32_CLASS, mxUINT32_CLASS, mxINT64_CLASS, mxUINT64_CLASS } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX } mxComplexity;

/* creation / destruction */
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity c);
mxArray *mxCreateDoubleScalar(double v);
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity c);
mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID cls, mxComplexity c);
mxArray *mxCreateLogicalMatrix(mwSize m, mwSize n);
mxArray *mxCreateLogicalScalar(bool v);
mxArray *mxCreateString(const char *s);
mxArray *mxCreateStructArray(mwSize ndim, const mwSize *dims, int nfields, const char **names);
mxArray *mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char **names);
mxArray *mxDuplicateArray(const mxArray *a);
void mxDestroyArray(mxArray *a);
void *mxMalloc(size_t n);
void *mxCalloc(size_t n, size_t sz);
void mxFree(void *p);

/* shape and class */
size_t mxGetNumberOfElements(const mxArray *a);
size_t mxGetM(const mxArray *a);
size_t mxGetN(const mxArray *a);   /* product of the dimensions after the first, as MATLAB defines it */
mwSize mxGetNumberOfDimensions(const mxArray *a);
const mwSize *mxGetDimensions(const mxArray *a);
mxClassID mxGetClassID(const mxArray *a);
bool mxIsStruct(const mxArray *a);
bool mxIsDouble(const mxArray *a);
bool mxIsInt64(const mxArray *a);
bool mxIsInt32(const mxArray *a);
bool mxIsChar(const mxArray *a);
bool mxIsLogical(const mxArray *a);
bool mxIsNumeric(const mxArray *a);
bool mxIsEmpty(const mxArray *a);
bool mxIsComplex(const mxArray *a);

/* data */
double *mxGetPr(const mxArray *a);
void *mxGetData(const mxArray *a);
mxLogical *mxGetLogicals(const mxArray *a);
double mxGetScalar(const mxArray *a);
int mxGetString(const mxArray *a, char *buf, mwSize buflen);
char *mxArrayToString(const mxArray *a);

/* struct arrays */
int mxGetNumberOfFields(const mxArray *a);
int mxGetFieldNumber(const mxArray *a, const char *name);
const char *mxGetFieldNameByNumber(const mxArray *a, int k);
mxArray *mxGetField(const mxArray *a, mwIndex i, const char *name);
mxArray *mxGetFieldByNumber(const mxArray *a, mwIndex i, int k);
void mxSetField(mxArray *a, mwIndex i, const char *name, mxArray *v);        /* does not free the previous value */
void mxSetFieldByNumber(mxArray *a, mwIndex i, int k, mxArray *v);
int mxAddField(mxArray *a, const char *name);

/* mex */
void mexErrMsgTxt(const char *msg);                         /* does not return */
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...);
void mexWarnMsgTxt(const char *msg);
int mexPrintf(const char *fmt, ...);
int mexAtExit(void (*fn)(void));
void mexLock(void);
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);

#ifdef __cplusplus
}
#endif
#endif
