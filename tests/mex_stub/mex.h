/* TEST-ONLY declaration stub: the handful of MEX API names mex/hjbdp_mex.c uses, declared (never defined) so that the
 * gateway can be SYNTAX- and TYPE-checked with `gcc -fsyntax-only` in an image without MATLAB.  Not MathWorks' header,
 * not shipped, not linked against; signatures follow the documented C Matrix API. */
#ifndef MEX_STUB_H
#define MEX_STUB_H
#include <stddef.h>
#include <stdbool.h>
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef enum { mxUNKNOWN_CLASS = 0, mxLOGICAL_CLASS = 3, mxDOUBLE_CLASS = 6, mxSINGLE_CLASS = 7, mxINT32_CLASS = 12 } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
bool mxIsStruct(const mxArray *);
bool mxIsCell(const mxArray *);
bool mxIsDouble(const mxArray *);
bool mxIsSingle(const mxArray *);
bool mxIsLogicalScalarTrue(const mxArray *);
mxArray *mxGetField(const mxArray *, mwIndex, const char *);
mxArray *mxGetCell(const mxArray *, mwIndex);
size_t mxGetNumberOfElements(const mxArray *);
double mxGetScalar(const mxArray *);
double *mxGetPr(const mxArray *);
void *mxGetData(const mxArray *);
void *mxMalloc(size_t);
void mxFree(void *);
mxArray *mxCreateNumericArray(mwSize, const mwSize *, mxClassID, mxComplexity);
mxArray *mxCreateNumericMatrix(mwSize, mwSize, mxClassID, mxComplexity);
mxArray *mxCreateStructMatrix(mwSize, mwSize, int, const char **);
mxArray *mxCreateDoubleScalar(double);
mxArray *mxCreateLogicalScalar(bool);
void mxSetField(mxArray *, mwIndex, const char *, mxArray *);
void mexErrMsgIdAndTxt(const char *, const char *, ...);
#endif
