/* A small functional stand-in for the MATLAB MEX runtime -- TEST INFRASTRUCTURE (tests/test_mex_run.py).
 *
 * Implements the part of the C Matrix / MEX API the gateways under mex/ use (tests/mexstub/mex.h) on a minimal mxArray
 * (real double matrices, 1 x 1 structs, strings), plus a driver: main() reads an OPTsettings description from a text
 * file, calls the gateway's mexFunction exactly as MATLAB would (one struct in, one struct out), and writes every field
 * of the returned struct to a text file.  mexCallMATLAB("SetVehicleParameters") is answered from the "V." entries of
 * the same file.  Written from the public API documentation; no MathWorks code.
 *
 * Input format, one entry per line:   <name> <rows> <cols> v1 v2 ...   (column-major, as MATLAB stores matrices);
 * names starting with "V." belong to the vehicle struct.   Output format: the same, strings as  <name> str <text>.
 */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mex.h"

enum { K_DOUBLE = 0, K_STRUCT = 1, K_STRING = 2 };
struct mxArray_tag {
    int kind;
    size_t m, n;
    double* pr;
    char* str;
    int nf;
    char** names;
    mxArray** vals;
};

static mxArray* new_array(int kind) {
    mxArray* a = (mxArray*)calloc(1, sizeof *a);
    if (!a) { fprintf(stderr, "mex_mock: out of memory\n"); exit(3); }
    a->kind = kind;
    return a;
}
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag) {
    mxArray* a = new_array(K_DOUBLE);
    (void)flag;
    a->m = m; a->n = n;
    a->pr = (double*)calloc((m * n) != 0 ? m * n : 1, sizeof(double));
    return a;
}
mxArray* mxCreateDoubleScalar(double value) { mxArray* a = mxCreateDoubleMatrix(1, 1, mxREAL); a->pr[0] = value; return a; }
mxArray* mxCreateString(const char* s) {
    mxArray* a = new_array(K_STRING);
    a->m = 1; a->n = strlen(s);
    a->str = (char*)malloc(a->n + 1);
    memcpy(a->str, s, a->n + 1);
    return a;
}
mxArray* mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char** fieldnames) {
    mxArray* a = new_array(K_STRUCT);
    int i;
    a->m = m; a->n = n;
    for (i = 0; i < nfields; ++i) mxAddField(a, fieldnames[i]);
    return a;
}
int mxGetFieldNumber(const mxArray* pm, const char* name) {
    int i;
    if (!pm || pm->kind != K_STRUCT) return -1;
    for (i = 0; i < pm->nf; ++i) if (strcmp(pm->names[i], name) == 0) return i;
    return -1;
}
int mxAddField(mxArray* pm, const char* name) {
    if (!pm || pm->kind != K_STRUCT) return -1;
    pm->names = (char**)realloc(pm->names, sizeof(char*) * (size_t)(pm->nf + 1));
    pm->vals = (mxArray**)realloc(pm->vals, sizeof(mxArray*) * (size_t)(pm->nf + 1));
    pm->names[pm->nf] = (char*)malloc(strlen(name) + 1);
    strcpy(pm->names[pm->nf], name);
    pm->vals[pm->nf] = NULL;
    return pm->nf++;
}
mxArray* mxGetFieldByNumber(const mxArray* pm, mwIndex index, int f) { (void)index; return (pm && f >= 0 && f < pm->nf) ? pm->vals[f] : NULL; }
mxArray* mxGetField(const mxArray* pm, mwIndex index, const char* name) { return mxGetFieldByNumber(pm, index, mxGetFieldNumber(pm, name)); }
void mxSetField(mxArray* pm, mwIndex index, const char* name, mxArray* v) {
    int f = mxGetFieldNumber(pm, name);
    (void)index;
    if (f < 0) f = mxAddField(pm, name);
    pm->vals[f] = v;
}
void mxRemoveField(mxArray* pm, int f) {
    int i;
    if (!pm || f < 0 || f >= pm->nf) return;
    free(pm->names[f]);
    for (i = f; i + 1 < pm->nf; ++i) { pm->names[i] = pm->names[i + 1]; pm->vals[i] = pm->vals[i + 1]; }
    --pm->nf;
}
double mxGetScalar(const mxArray* pm) { return (pm && pm->kind == K_DOUBLE && pm->m * pm->n > 0) ? pm->pr[0] : 0.0; }
double* mxGetPr(const mxArray* pm) { return pm ? pm->pr : NULL; }
size_t mxGetNumberOfElements(const mxArray* pm) { return pm ? pm->m * pm->n : 0; }
size_t mxGetM(const mxArray* pm) { return pm ? pm->m : 0; }
size_t mxGetN(const mxArray* pm) { return pm ? pm->n : 0; }
bool mxIsStruct(const mxArray* pm) { return pm && pm->kind == K_STRUCT; }
bool mxIsDouble(const mxArray* pm) { return pm && pm->kind == K_DOUBLE; }
bool mxIsLogical(const mxArray* pm) { (void)pm; return false; }
bool mxIsComplex(const mxArray* pm) { (void)pm; return false; }
bool mxIsEmpty(const mxArray* pm) { return !pm || pm->m * pm->n == 0; }
void mxDestroyArray(mxArray* pm) {
    int i;
    if (!pm) return;
    for (i = 0; i < pm->nf; ++i) { free(pm->names[i]); mxDestroyArray(pm->vals[i]); }
    free(pm->names); free(pm->vals); free(pm->pr); free(pm->str); free(pm);
}
void* mxMalloc(size_t n) { return malloc(n ? n : 1); }
void* mxCalloc(size_t n, size_t size) { return calloc(n ? n : 1, size ? size : 1); }
void mxFree(void* p) { free(p); }

static void (*g_at_exit)(void) = NULL;
int mexAtExit(void (*fn)(void)) { g_at_exit = fn; return 0; }
void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...) {
    va_list ap;
    fprintf(stderr, "MEX ERROR %s: ", id);
    va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap);
    fprintf(stderr, "\n");
    if (g_at_exit) g_at_exit();
    exit(2);                                   /* MATLAB would unwind to the prompt */
}

static mxArray* g_vehicle = NULL;
int mexCallMATLAB(int nlhs, mxArray* plhs[], int nrhs, mxArray* prhs[], const char* name) {
    (void)nrhs; (void)prhs;
    if (strcmp(name, "SetVehicleParameters") != 0 || nlhs != 1 || !g_vehicle) return 1;
    {   /* the gateway destroys what it gets: hand out a copy */
        mxArray* c = mxCreateStructMatrix(1, 1, 0, NULL);
        int i;
        for (i = 0; i < g_vehicle->nf; ++i) {
            const mxArray* v = g_vehicle->vals[i];
            mxArray* d = mxCreateDoubleMatrix(v->m, v->n, mxREAL);
            memcpy(d->pr, v->pr, sizeof(double) * v->m * v->n);
            mxSetField(c, 0, g_vehicle->names[i], d);
        }
        plhs[0] = c;
    }
    return 0;
}

static mxArray* read_struct(const char* path) {
    FILE* f = fopen(path, "r");
    mxArray* S = mxCreateStructMatrix(1, 1, 0, NULL);
    char name[256];
    long m, n;
    if (!f) { fprintf(stderr, "mex_mock: cannot read %s\n", path); exit(3); }
    g_vehicle = mxCreateStructMatrix(1, 1, 0, NULL);
    while (fscanf(f, "%255s %ld %ld", name, &m, &n) == 3) {
        mxArray* a = mxCreateDoubleMatrix((mwSize)m, (mwSize)n, mxREAL);
        long i;
        for (i = 0; i < m * n; ++i) {
            char tok[64];
            if (fscanf(f, "%63s", tok) != 1) { fprintf(stderr, "mex_mock: short entry %s\n", name); exit(3); }
            a->pr[i] = (strcmp(tok, "inf") == 0 || strcmp(tok, "Inf") == 0) ? 1e308 * 10 : ((strcmp(tok, "-inf") == 0) ? -1e308 * 10 : strtod(tok, NULL));
        }
        if (strncmp(name, "V.", 2) == 0) mxSetField(g_vehicle, 0, name + 2, a);
        else mxSetField(S, 0, name, a);
    }
    fclose(f);
    return S;
}

static void write_struct(const char* path, const mxArray* S) {
    FILE* f = fopen(path, "w");
    int i;
    if (!f) { fprintf(stderr, "mex_mock: cannot write %s\n", path); exit(3); }
    for (i = 0; i < S->nf; ++i) {
        const mxArray* v = S->vals[i];
        size_t k;
        if (!v) continue;
        if (v->kind == K_STRING) { fprintf(f, "%s str %s\n", S->names[i], v->str); continue; }
        if (v->kind != K_DOUBLE) continue;
        fprintf(f, "%s %zu %zu", S->names[i], v->m, v->n);
        for (k = 0; k < v->m * v->n; ++k) fprintf(f, " %.17g", v->pr[k]);
        fprintf(f, "\n");
    }
    fclose(f);
}

int main(int argc, char** argv) {
    mxArray* in;
    mxArray* out[1] = {NULL};
    const mxArray* prhs[1];
    if (argc != 3) { fprintf(stderr, "usage: %s <OPTsettings.txt> <optSol.txt>\n", argv[0]); return 3; }
    in = read_struct(argv[1]);
    prhs[0] = in;
    mexFunction(1, out, 1, prhs);
    if (!out[0] || !mxIsStruct(out[0])) { fprintf(stderr, "mex_mock: the gateway returned no struct\n"); return 3; }
    write_struct(argv[2], out[0]);
    if (g_at_exit) g_at_exit();
    return 0;
}
