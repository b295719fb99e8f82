/* mex_decl.h -- DECLARATIONS ONLY of the MATLAB MEX API used by mex/prost_mex.cpp (mex.h / matrix.h are not in the build
 * image).  TEST INFRASTRUCTURE: tests/test_frontend.py compiles the gateway with `g++ -fsyntax-only -include tests/mex_decl.h`
 * so that it cannot rot; nothing here is linked or shipped, and no function has a body.  Signatures as documented in the
 * MATLAB C Matrix / MEX API reference. */
#ifndef PROST_TESTS_MEX_DECL_H_
#define PROST_TESTS_MEX_DECL_H_
#define PROST_MEX_DECLARATIONS_PROVIDED 1
#include <stddef.h>
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef bool mxLogical;
typedef enum { mxREAL, mxCOMPLEX } mxComplexity;
extern "C" {
void mexErrMsgTxt(const char* msg);
int mexPrintf(const char* fmt, ...);
int mexEvalString(const char* command);
int mexCallMATLAB(int nlhs, mxArray* plhs[], int nrhs, mxArray* prhs[], const char* name);
mxArray* mxCreateDoubleScalar(double v);
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray* mxCreateString(const char* s);
mxArray* mxCreateCellMatrix(mwSize m, mwSize n);
mxArray* mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char** names);
void mxDestroyArray(mxArray* a);
void mxFree(void* p);
double* mxGetPr(const mxArray* a);
void* mxGetData(const mxArray* a);
double mxGetScalar(const mxArray* a);
size_t mxGetM(const mxArray* a);
size_t mxGetN(const mxArray* a);
size_t mxGetNumberOfElements(const mxArray* a);
int mxGetNumberOfFields(const mxArray* a);
const char* mxGetFieldNameByNumber(const mxArray* a, int n);
mxArray* mxGetFieldByNumber(const mxArray* a, mwIndex i, int field);
void mxSetFieldByNumber(mxArray* a, mwIndex i, int field, mxArray* v);
mxArray* mxGetCell(const mxArray* a, mwIndex i);
void mxSetCell(mxArray* a, mwIndex i, mxArray* v);
mwIndex* mxGetIr(const mxArray* a);
mwIndex* mxGetJc(const mxArray* a);
char* mxArrayToString(const mxArray* a);
bool mxIsCell(const mxArray* a);
bool mxIsStruct(const mxArray* a);
bool mxIsChar(const mxArray* a);
bool mxIsSparse(const mxArray* a);
bool mxIsEmpty(const mxArray* a);
bool mxIsLogical(const mxArray* a);
bool mxIsDouble(const mxArray* a);
bool mxIsSingle(const mxArray* a);
bool mxIsInt8(const mxArray* a);
bool mxIsUint8(const mxArray* a);
bool mxIsInt16(const mxArray* a);
bool mxIsUint16(const mxArray* a);
bool mxIsInt32(const mxArray* a);
bool mxIsUint32(const mxArray* a);
bool mxIsInt64(const mxArray* a);
bool mxIsUint64(const mxArray* a);
bool mxIsClass(const mxArray* a, const char* classname);
}
#endif
