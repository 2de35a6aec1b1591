/* mex.h — DECLARATIONS ONLY, test infrastructure.
 *
 * MATLAB is not installed in this image, so matlab/aps_mex.cpp can never be linked here.  This header declares the
 * subset of the documented MEX C API that the gateway uses (names, argument and return types as in MathWorks' public
 * C Matrix API reference) so that tests/test_mex_syntax.py can run `g++ -fsyntax-only` over the gateway: every call is
 * type-checked against the API, every aps.h entry point against its use.  Nothing here has a definition and nothing is
 * ever linked or executed; it is not an oracle and not a stand-in for a reference build. */
#ifndef APS_TEST_MEX_DECL_H_
#define APS_TEST_MEX_DECL_H_
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef bool mxLogical;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef enum {
    mxUNKNOWN_CLASS = 0, mxCELL_CLASS, mxSTRUCT_CLASS, mxLOGICAL_CLASS, mxCHAR_CLASS, mxVOID_CLASS, mxDOUBLE_CLASS,
    mxSINGLE_CLASS, mxINT8_CLASS, mxUINT8_CLASS, mxINT16_CLASS, mxUINT16_CLASS, mxINT32_CLASS, mxUINT32_CLASS,
    mxINT64_CLASS, mxUINT64_CLASS, mxFUNCTION_CLASS
} mxClassID;

bool mxIsChar(const mxArray*);
bool mxIsCell(const mxArray*);
bool mxIsStruct(const mxArray*);
bool mxIsDouble(const mxArray*);
bool mxIsSingle(const mxArray*);
bool mxIsUint8(const mxArray*);
bool mxIsUint32(const mxArray*);
bool mxIsLogical(const mxArray*);
bool mxIsEmpty(const mxArray*);
mxClassID mxGetClassID(const mxArray*);
size_t mxGetM(const mxArray*);
size_t mxGetN(const mxArray*);
size_t mxGetNumberOfElements(const mxArray*);
mwSize mxGetNumberOfDimensions(const mxArray*);
const mwSize* mxGetDimensions(const mxArray*);
void* mxGetData(const mxArray*);
double* mxGetPr(const mxArray*);
double mxGetScalar(const mxArray*);
char* mxArrayToString(const mxArray*);
mxArray* mxGetCell(const mxArray*, mwIndex);
void mxSetCell(mxArray*, mwIndex, mxArray*);
mxArray* mxGetField(const mxArray*, mwIndex, const char*);
mxArray* mxCreateNumericMatrix(mwSize, mwSize, mxClassID, mxComplexity);
mxArray* mxCreateNumericArray(mwSize, const mwSize*, mxClassID, mxComplexity);
mxArray* mxCreateDoubleMatrix(mwSize, mwSize, mxComplexity);
mxArray* mxCreateDoubleScalar(double);
mxArray* mxCreateLogicalScalar(bool);
mxArray* mxCreateLogicalMatrix(mwSize, mwSize);
mxLogical* mxGetLogicals(const mxArray*);
mxArray* mxCreateCellMatrix(mwSize, mwSize);
void mxDestroyArray(mxArray*);
void mxFree(void*);
double mxGetNaN(void);
double mxGetInf(void);
void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...);
void mexWarnMsgIdAndTxt(const char* id, const char* fmt, ...);
int mexPrintf(const char* fmt, ...);

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);

#ifdef __cplusplus
}
#endif
#endif
