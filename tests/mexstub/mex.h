/* Minimal declarations of the MATLAB MEX C API used by the gateways under mex/, for a syntax/type check of the gateways with gcc
 * (tests/test_mex_sources.py).  TEST INFRASTRUCTURE: declarations only, written from the public API documentation
 * (MathWorks "C Matrix API" / "C MEX API" reference pages); the real mex.h ships with MATLAB. */
#ifndef EEPACC_TEST_MEX_STUB_H
#define EEPACC_TEST_MEX_STUB_H
#include <stddef.h>
#include <stdbool.h>
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
#ifdef __cplusplus
extern "C" {
#endif
mxArray* mxGetField(const mxArray* pm, mwIndex index, const char* fieldname);
double mxGetScalar(const mxArray* pm);
double* mxGetPr(const mxArray* pm);
size_t mxGetNumberOfElements(const mxArray* pm);
size_t mxGetM(const mxArray* pm);
size_t mxGetN(const mxArray* pm);
bool mxIsStruct(const mxArray* pm);
bool mxIsDouble(const mxArray* pm);
bool mxIsLogical(const mxArray* pm);
bool mxIsComplex(const mxArray* pm);
bool mxIsEmpty(const mxArray* pm);
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray* mxCreateDoubleScalar(double value);
mxArray* mxCreateString(const char* str);
mxArray* mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char** fieldnames);
int mxAddField(mxArray* pm, const char* fieldname);
int mxGetFieldNumber(const mxArray* pm, const char* fieldname);
mxArray* mxGetFieldByNumber(const mxArray* pm, mwIndex index, int fieldnumber);
void mxRemoveField(mxArray* pm, int fieldnumber);
void mxSetField(mxArray* pm, mwIndex index, const char* fieldname, mxArray* pvalue);
void mxDestroyArray(mxArray* pm);
void* mxMalloc(size_t n);
void* mxCalloc(size_t n, size_t size);
void mxFree(void* ptr);
int mexCallMATLAB(int nlhs, mxArray* plhs[], int nrhs, mxArray* prhs[], const char* name);
void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...);
int mexAtExit(void (*fn)(void));
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);
#ifdef __cplusplus
}
#endif
#endif
