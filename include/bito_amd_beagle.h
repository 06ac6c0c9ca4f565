/*
 * bito_amd_beagle.h -- the BEAGLE C API subset that bito's FatBeagle calls
 * (seam 1), implemented on the MI355X by libbito_amd.so.
 *
 * This is a compatibility shim: one tree at a time, synchronous, one kernel per
 * BEAGLE operation list -- the op-by-op formulation whose HBM traffic SURVEY 8d
 * prices at 40.5 MB/tree.  It exists so that the reference's src/fat_beagle.cpp
 * can link against this library unchanged; throughput work goes through the
 * batched engine in bito_amd.h.
 *
 * BEAGLE's own header (libhmsbeagle/beagle.h) is not vendored in the reference;
 * the signatures are those of BEAGLE's public C API as used at the call sites
 * cited below (all in reference src/fat_beagle.cpp).  Every function returns 0
 * on success and a negative BEAGLE_ERROR_* value otherwise, except
 * beagleCreateInstance which returns the instance number (>= 0).
 * 4 states, double precision, manual scaling with log scalers.
 */
#ifndef BITO_AMD_BEAGLE_H
#define BITO_AMD_BEAGLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define BEAGLE_SUCCESS 0
#define BEAGLE_ERROR_GENERAL (-1)
#define BEAGLE_ERROR_OUT_OF_MEMORY (-2)
#define BEAGLE_ERROR_UNIDENTIFIED_EXCEPTION (-3)
#define BEAGLE_ERROR_UNINITIALIZED_INSTANCE (-4)
#define BEAGLE_ERROR_OUT_OF_RANGE (-5)
#define BEAGLE_ERROR_NO_RESOURCE (-6)
#define BEAGLE_ERROR_NO_IMPLEMENTATION (-7)

#define BEAGLE_OP_NONE (-1)

/* bit positions as listed in reference src/beagle_flag_names.hpp:21-53 */
#define BEAGLE_FLAG_PRECISION_SINGLE (1L << 0)
#define BEAGLE_FLAG_PRECISION_DOUBLE (1L << 1)
#define BEAGLE_FLAG_COMPUTATION_SYNCH (1L << 2)
#define BEAGLE_FLAG_EIGEN_REAL (1L << 4)
#define BEAGLE_FLAG_SCALING_MANUAL (1L << 6)
#define BEAGLE_FLAG_SCALERS_LOG (1L << 10)
#define BEAGLE_FLAG_VECTOR_SSE (1L << 11)
#define BEAGLE_FLAG_VECTOR_NONE (1L << 12)
#define BEAGLE_FLAG_THREADING_NONE (1L << 14)
#define BEAGLE_FLAG_PROCESSOR_CPU (1L << 15)
#define BEAGLE_FLAG_PROCESSOR_GPU (1L << 16)
#define BEAGLE_FLAG_INVEVEC_STANDARD (1L << 20)

typedef struct {
  int resourceNumber;
  char *resourceName;
  char *implName;
  char *implDescription;
  long flags;
} BeagleInstanceDetails;

/* field order as filled at fat_beagle.cpp:345-352 */
typedef struct {
  int destinationPartials;
  int destinationScaleWrite;
  int destinationScaleRead;
  int child1Partials;
  int child1TransitionMatrix;
  int child2Partials;
  int child2TransitionMatrix;
} BeagleOperation;

/* fat_beagle.cpp:258-262 */
int beagleCreateInstance(int tipCount, int partialsBufferCount, int compactBufferCount, int stateCount,
                         int patternCount, int eigenBufferCount, int matrixBufferCount, int categoryCount,
                         int scaleBufferCount, int *resourceList, int resourceCount, long preferenceFlags,
                         long requirementFlags, BeagleInstanceDetails *returnInfo);
int beagleFinalizeInstance(int instance);                                               /* :31 */
int beagleSetTipStates(int instance, int tipIndex, const int *inStates);                /* :272 */
int beagleSetTipPartials(int instance, int tipIndex, const double *inPartials);         /* :279 */
int beagleSetPartials(int instance, int bufferIndex, const double *inPartials);         /* :334 */
int beagleSetPatternWeights(int instance, const double *inPatternWeights);              /* :274,281 */
int beagleSetCategoryWeights(int instance, int categoryWeightsIndex, const double *inCategoryWeights); /* :288 */
int beagleSetCategoryRates(int instance, const double *inCategoryRates);                /* :289 */
int beagleSetStateFrequencies(int instance, int stateFrequenciesIndex, const double *inStateFrequencies); /* :300 */
int beagleSetEigenDecomposition(int instance, int eigenIndex, const double *inEigenVectors,
                                const double *inInverseEigenVectors, const double *inEigenValues); /* :301-304 */
int beagleUpdateTransitionMatrices(int instance, int eigenIndex, const int *probabilityIndices,
                                   const int *firstDerivativeIndices, const int *secondDerivativeIndices,
                                   const double *edgeLengths, int count);               /* :318-324 */
int beagleResetScaleFactors(int instance, int cumulativeScaleIndex);                    /* :53,116 */
int beagleUpdatePartials(const int instance, const BeagleOperation *operations, int operationCount,
                         int cumulativeScaleIndex);                                     /* :59-62,133-135 */
int beagleUpdatePrePartials(const int instance, const BeagleOperation *operations, int operationCount,
                            int cumulativeScaleIndex);                                  /* :143-145 */
int beagleSetDifferentialMatrix(int instance, int matrixIndex, const double *inMatrix); /* :123 */
int beagleCalculateEdgeDerivatives(int instance, const int *postBufferIndices, const int *preBufferIndices,
                                   const int *derivativeMatrixIndices, const int *categoryWeightsIndices,
                                   int count, double *outDerivatives, double *outSumDerivatives,
                                   double *outSumSquaredDerivatives);                   /* :151-160 */
int beagleCalculateRootLogLikelihoods(int instance, const int *bufferIndices, const int *categoryWeightsIndices,
                                      const int *stateFrequenciesIndices, const int *cumulativeScaleIndices,
                                      int count, double *outSumLogLikelihood);          /* :64-67,164-167 */

#ifdef __cplusplus
}
#endif
#endif
