// beagle_shim.hip -- the 17 BEAGLE entry points bito's FatBeagle uses, on the GPU.
//
// One instance = one set of device buffers laid out as FatBeagle::CreateInstance asks
// (reference src/fat_beagle.cpp:218-267): partials [buffer][category][pattern][state],
// compact tip states, transition/differential matrices [index][category][4][4], one
// eigensystem, log scale factors [buffer][pattern].  Each BEAGLE call maps to one small
// kernel and is synchronous where it returns data.  This is the op-by-op formulation; the
// batched engine (engine.cpp) is the fast path.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/bito_amd_beagle.h"

namespace {

constexpr int S = 4;

struct Instance {
  int tips = 0, partials = 0, compact = 0, patterns = 0, eigens = 0, matrices = 0, categories = 0, scales = 0;
  double* d_partials = nullptr;  // [partials][C][P][S]
  int* d_states = nullptr;       // [tips][P]
  std::vector<char> tip_is_compact;
  double* d_mats = nullptr;      // [matrices][C][16]
  double* d_scale = nullptr;     // [scales][P] log scalers
  double* d_weights = nullptr;   // [P]
  double* d_catw = nullptr;      // [C]
  double* d_catr = nullptr;      // [C]
  double* d_freq = nullptr;      // [S]
  double* d_eigen = nullptr;     // V[16] Vinv[16] lambda[4]
  double* d_out = nullptr;       // scratch for reductions
  int* d_idx = nullptr;          // scratch for index lists / ops
  size_t idx_cap = 0, out_cap = 0;
  int device = 0;                // HIP device ordinal = this shim's BEAGLE resource number
  size_t plv() const { return (size_t)categories * patterns * S; }
  ~Instance() {
    (void)hipSetDevice(device);
    for (void* p : {(void*)d_partials, (void*)d_states, (void*)d_mats, (void*)d_scale, (void*)d_weights,
                    (void*)d_catw, (void*)d_catr, (void*)d_freq, (void*)d_eigen, (void*)d_out, (void*)d_idx})
      if (p) (void)hipFree(p);
  }
};

std::mutex g_mu;
std::vector<std::unique_ptr<Instance>> g_instances;

// The instance, with its device made current for the calling thread (instances may live on
// different GPUs; BEAGLE's contract is one thread per instance at a time).
Instance* Get(int id) {
  Instance* in = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    if (id < 0 || id >= (int)g_instances.size()) return nullptr;
    in = g_instances[id].get();
  }
  if (in && hipSetDevice(in->device) != hipSuccess) return nullptr;
  return in;
}

bool Ok(hipError_t rc) { return rc == hipSuccess; }

// out[i] = sum_j M[i][j] x[j], or the tip-state column (gap = all ones) for a compact child
__device__ inline void ChildMessage(const double* __restrict__ M, const double* __restrict__ part,
                                    const int* __restrict__ states, int p, size_t cp, double out[S]) {
  if (states != nullptr) {
    const int s = states[p];
    for (int i = 0; i < S; i++) out[i] = s < S ? M[i * 4 + s] : 1.0;
  } else {
    const double* x = part + cp * S;
    for (int i = 0; i < S; i++) out[i] = M[i * 4] * x[0] + M[i * 4 + 1] * x[1] + M[i * 4 + 2] * x[2] + M[i * 4 + 3] * x[3];
  }
}

__global__ void matrices_kernel(const double* __restrict__ eigen, const double* __restrict__ catr,
                                const int* __restrict__ idx, const double* __restrict__ t, int count, int C,
                                double* __restrict__ mats) {
#pragma clang fp contract(off)
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= count * C * 16) return;
  const int e = u & 15, c = (u >> 4) % C, b = (u >> 4) / C;
  const int i = e >> 2, j = e & 3;
  const double* V = eigen;
  const double* Vi = eigen + 16;
  const double* lam = eigen + 32;
  double s = 0;
  for (int k = 0; k < 4; k++) s += V[i * 4 + k] * exp(lam[k] * (t[b] * catr[c])) * Vi[k * 4 + j];
  mats[((size_t)idx[b] * C + c) * 16 + e] = s;
}

// one thread per pattern, loops over categories (the rescale max spans categories)
__global__ void partials_kernel(double* __restrict__ partials, const int* __restrict__ states_base,
                                const double* __restrict__ mats, double* __restrict__ scale, int P, int C,
                                size_t plv, int dest, int c1, int m1, int c1_compact, int c2, int m2, int c2_compact,
                                int scale_write, int cumulative, int pre_order) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  double* out = partials + (size_t)dest * plv;
  const int* s1 = c1_compact ? states_base + (size_t)c1 * P : nullptr;
  const int* s2 = c2_compact ? states_base + (size_t)c2 * P : nullptr;
  double mx = 0;
  for (int c = 0; c < C; c++) {
    const double* M1 = mats + ((size_t)m1 * C + c) * 16;
    const double* M2 = mats + ((size_t)m2 * C + c) * 16;
    const size_t cp = (size_t)c * P + p;
    double b[S], d[S];
    ChildMessage(M2, partials + (size_t)c2 * plv, s2, p, cp, b);
    if (!pre_order) {
      double a[S];
      ChildMessage(M1, partials + (size_t)c1 * plv, s1, p, cp, a);
      for (int i = 0; i < S; i++) d[i] = a[i] * b[i];
    } else {
      // child1 = pre-order partial of the parent, matrix1 = this node's matrix (applied
      // transposed), child2 = sibling's post-order partial (fat_beagle.cpp:355-373)
      const double* par = partials + (size_t)c1 * plv + cp * S;
      double u[S];
      for (int i = 0; i < S; i++) u[i] = par[i] * b[i];
      for (int j = 0; j < S; j++) d[j] = M1[j] * u[0] + M1[4 + j] * u[1] + M1[8 + j] * u[2] + M1[12 + j] * u[3];
    }
    for (int i = 0; i < S; i++) {
      out[cp * S + i] = d[i];
      mx = fmax(mx, d[i]);
    }
  }
  if (scale_write >= 0) {
    if (mx == 0) mx = 1.0;
    const double inv = 1.0 / mx;
    for (int c = 0; c < C; c++)
      for (int i = 0; i < S; i++) out[((size_t)c * P + p) * S + i] *= inv;
    const double lg = log(mx);
    scale[(size_t)scale_write * P + p] = lg;
    if (cumulative >= 0) scale[(size_t)cumulative * P + p] += lg;
  }
}

__device__ inline double BlockSum(double v, double* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double s = 0;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) s += sh[w];
  return s;
}

// block per edge: sum_p w_p [sum_c wc pre^T dQ_c post] / [sum_c wc pre^T post]
__global__ void edge_derivative_kernel(const double* __restrict__ partials, const int* __restrict__ states_base,
                                       const double* __restrict__ mats, const double* __restrict__ weights,
                                       const double* __restrict__ catw, const int* __restrict__ lists, int count,
                                       int P, int C, size_t plv, int tips, const char* __restrict__ compact,
                                       double* __restrict__ out) {
  __shared__ double sh[4];
  const int e = blockIdx.x;
  const int post = lists[e], pre = lists[count + e], dm = lists[2 * count + e];
  const bool is_compact = post < tips && compact[post];
  double acc = 0;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    double num = 0, den = 0;
    for (int c = 0; c < C; c++) {
      const double* D = mats + ((size_t)dm * C + c) * 16;
      const double* u = partials + (size_t)pre * plv + ((size_t)c * P + p) * S;
      double x[S];
      if (is_compact) {
        const int s = states_base[(size_t)post * P + p];
        for (int i = 0; i < S; i++) x[i] = (s >= S || s == i) ? 1.0 : 0.0;
      } else {
        const double* xp = partials + (size_t)post * plv + ((size_t)c * P + p) * S;
        for (int i = 0; i < S; i++) x[i] = xp[i];
      }
      double nc = 0, dc = 0;
      for (int i = 0; i < S; i++) {
        nc += u[i] * (D[i * 4] * x[0] + D[i * 4 + 1] * x[1] + D[i * 4 + 2] * x[2] + D[i * 4 + 3] * x[3]);
        dc += u[i] * x[i];
      }
      num += catw[c] * nc;
      den += catw[c] * dc;
    }
    acc += weights[p] * (num / den);
  }
  const double s = BlockSum(acc, sh);
  if (threadIdx.x == 0) out[e] = s;
}

__global__ void root_kernel(const double* __restrict__ root, const double* __restrict__ weights,
                            const double* __restrict__ catw, const double* __restrict__ freq,
                            const double* __restrict__ cum, int P, int C, double* __restrict__ out) {
  __shared__ double sh[4];
  double acc = 0;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    double site = 0;
    for (int c = 0; c < C; c++) {
      const double* x = root + ((size_t)c * P + p) * S;
      site += catw[c] * (freq[0] * x[0] + freq[1] * x[1] + freq[2] * x[2] + freq[3] * x[3]);
    }
    double lp = log(site);
    if (cum) lp += cum[p];
    acc += weights[p] * lp;
  }
  const double s = BlockSum(acc, sh);
  if (threadIdx.x == 0) out[0] = s;
}

template <typename T>
bool Upload(T* dst, const T* src, size_t count) {
  return Ok(hipMemcpy(dst, src, count * sizeof(T), hipMemcpyHostToDevice));
}

bool ReserveIdx(Instance* in, size_t count) {
  if (count <= in->idx_cap) return true;
  if (in->d_idx) (void)hipFree(in->d_idx);
  in->idx_cap = 0;
  if (!Ok(hipMalloc((void**)&in->d_idx, count * sizeof(int)))) return false;
  in->idx_cap = count;
  return true;
}

bool ReserveOut(Instance* in, size_t count) {
  if (count <= in->out_cap) return true;
  if (in->d_out) (void)hipFree(in->d_out);
  in->out_cap = 0;
  if (!Ok(hipMalloc((void**)&in->d_out, count * sizeof(double)))) return false;
  in->out_cap = count;
  return true;
}

int RunOps(int instance, const BeagleOperation* ops, int count, int cumulative, int pre_order) {
  Instance* in = Get(instance);
  if (!in) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
  const int blocks = (in->patterns + 255) / 256;
  // every index of every operation is checked before the first launch: they go straight into device
  // pointer arithmetic (and tip_is_compact on the host)
  auto partial_ok = [&](int i) { return i >= 0 && i < in->partials; };
  auto matrix_ok = [&](int i) { return i >= 0 && i < in->matrices; };
  auto scale_ok = [&](int i) { return i == BEAGLE_OP_NONE || (i >= 0 && i < in->scales); };
  if (count > 0 && !ops) return BEAGLE_ERROR_OUT_OF_RANGE;
  if (!scale_ok(cumulative)) return BEAGLE_ERROR_OUT_OF_RANGE;
  for (int o = 0; o < count; o++) {
    const BeagleOperation& op = ops[o];
    if (!partial_ok(op.destinationPartials) || !partial_ok(op.child1Partials) || !partial_ok(op.child2Partials) ||
        !matrix_ok(op.child1TransitionMatrix) || !matrix_ok(op.child2TransitionMatrix) ||
        !scale_ok(op.destinationScaleWrite) || !scale_ok(op.destinationScaleRead))
      return BEAGLE_ERROR_OUT_OF_RANGE;
  }
  for (int o = 0; o < count; o++) {
    const BeagleOperation& op = ops[o];
    const int c1c = !pre_order && op.child1Partials < in->tips && in->tip_is_compact[op.child1Partials];
    const int c2c = op.child2Partials < in->tips && in->tip_is_compact[op.child2Partials];
    hipLaunchKernelGGL(partials_kernel, dim3(blocks), dim3(256), 0, 0, in->d_partials, in->d_states, in->d_mats,
                       in->d_scale, in->patterns, in->categories, in->plv(), op.destinationPartials,
                       op.child1Partials, op.child1TransitionMatrix, c1c, op.child2Partials,
                       op.child2TransitionMatrix, c2c, op.destinationScaleWrite, cumulative, pre_order);
  }
  return Ok(hipGetLastError()) ? BEAGLE_SUCCESS : BEAGLE_ERROR_GENERAL;
}

}  // namespace

extern "C" {

int beagleCreateInstance(int tipCount, int partialsBufferCount, int compactBufferCount, int stateCount,
                         int patternCount, int eigenBufferCount, int matrixBufferCount, int categoryCount,
                         int scaleBufferCount, int* resourceList, int resourceCount, long, long requirementFlags,
                         BeagleInstanceDetails* returnInfo) {
  if (stateCount != S) return BEAGLE_ERROR_NO_IMPLEMENTATION;
  if (requirementFlags & BEAGLE_FLAG_PRECISION_SINGLE) return BEAGLE_ERROR_NO_IMPLEMENTATION;
  if (tipCount < 0 || partialsBufferCount < 0 || compactBufferCount < 0 || patternCount <= 0 || categoryCount <= 0 ||
      matrixBufferCount < 0 || scaleBufferCount < 0)
    return BEAGLE_ERROR_OUT_OF_RANGE;
  int devices = 0;
  if (!Ok(hipGetDeviceCount(&devices)) || devices <= 0) return BEAGLE_ERROR_NO_RESOURCE;
  // Resource numbers of this library are HIP device ordinals.  With a resource list the instance goes to the
  // first listed device that exists; without one (bito passes NULL, fat_beagle.cpp:247-252) to the calling
  // thread's current device.
  int device = 0;
  if (resourceList != nullptr && resourceCount > 0) {
    device = -1;
    for (int i = 0; i < resourceCount && device < 0; i++)
      if (resourceList[i] >= 0 && resourceList[i] < devices) device = resourceList[i];
    if (device < 0) return BEAGLE_ERROR_NO_RESOURCE;
  } else if (!Ok(hipGetDevice(&device))) {
    return BEAGLE_ERROR_NO_RESOURCE;
  }
  if (!Ok(hipSetDevice(device))) return BEAGLE_ERROR_NO_RESOURCE;
  auto in = std::make_unique<Instance>();
  in->device = device;
  // BEAGLE's buffer index space covers partials AND compact buffers (tips first): with tip
  // states FatBeagle asks for 3n-2 partials + n compact and addresses buffers up to 4n-3.
  in->tips = tipCount; in->partials = partialsBufferCount + compactBufferCount; in->compact = compactBufferCount;
  in->patterns = patternCount; in->eigens = eigenBufferCount; in->matrices = matrixBufferCount;
  in->categories = categoryCount; in->scales = scaleBufferCount;
  in->tip_is_compact.assign(tipCount, 0);
  const size_t P = patternCount;
  bool ok = Ok(hipMalloc((void**)&in->d_partials, (size_t)in->partials * in->plv() * sizeof(double))) &&
            Ok(hipMalloc((void**)&in->d_states, (size_t)(tipCount > 0 ? tipCount : 1) * P * sizeof(int))) &&
            Ok(hipMalloc((void**)&in->d_mats, (size_t)matrixBufferCount * categoryCount * 16 * sizeof(double))) &&
            Ok(hipMalloc((void**)&in->d_scale, (size_t)(scaleBufferCount > 0 ? scaleBufferCount : 1) * P * sizeof(double))) &&
            Ok(hipMalloc((void**)&in->d_weights, P * sizeof(double))) &&
            Ok(hipMalloc((void**)&in->d_catw, categoryCount * sizeof(double))) &&
            Ok(hipMalloc((void**)&in->d_catr, categoryCount * sizeof(double))) &&
            Ok(hipMalloc((void**)&in->d_freq, S * sizeof(double))) &&
            Ok(hipMalloc((void**)&in->d_eigen, 36 * sizeof(double)));
  if (!ok) return BEAGLE_ERROR_OUT_OF_MEMORY;
  (void)hipMemset(in->d_scale, 0, (size_t)(scaleBufferCount > 0 ? scaleBufferCount : 1) * P * sizeof(double));
  (void)hipMemset(in->d_mats, 0, (size_t)matrixBufferCount * categoryCount * 16 * sizeof(double));
  if (returnInfo) {
    static char name[] = "AMD Instinct MI355X (gfx950)";
    static char impl[] = "bito_amd-HIP-double";
    static char desc[] = "bito_amd BEAGLE-compatible shim, 4-state FP64, manual log scaling";
    returnInfo->resourceNumber = device;
    returnInfo->resourceName = name;
    returnInfo->implName = impl;
    returnInfo->implDescription = desc;
    returnInfo->flags = BEAGLE_FLAG_PRECISION_DOUBLE | BEAGLE_FLAG_COMPUTATION_SYNCH | BEAGLE_FLAG_EIGEN_REAL |
                        BEAGLE_FLAG_SCALING_MANUAL | BEAGLE_FLAG_SCALERS_LOG | BEAGLE_FLAG_VECTOR_NONE |
                        BEAGLE_FLAG_THREADING_NONE | BEAGLE_FLAG_PROCESSOR_GPU | BEAGLE_FLAG_INVEVEC_STANDARD;
  }
  std::lock_guard<std::mutex> lock(g_mu);
  for (size_t i = 0; i < g_instances.size(); i++)
    if (!g_instances[i]) {
      g_instances[i] = std::move(in);
      return (int)i;
    }
  g_instances.push_back(std::move(in));
  return (int)g_instances.size() - 1;
}

int beagleFinalizeInstance(int instance) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (instance < 0 || instance >= (int)g_instances.size() || !g_instances[instance])
    return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
  g_instances[instance].reset();
  return BEAGLE_SUCCESS;
}

int beagleSetTipStates(int instance, int tipIndex, const int* inStates) {
  Instance* in = Get(instance);
  if (!in) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
  if (tipIndex < 0 || tipIndex >= in->tips) return BEAGLE_ERROR_OUT_OF_RANGE;
  if (!Upload(in->d_states + (size_t)tipIndex * in->patterns, inStates, in->patterns)) return BEAGLE_ERROR_GENERAL;
  in->tip_is_compact[tipIndex] = 1;
  return BEAGLE_SUCCESS;
}

int beagleSetTipPartials(int instance, int tipIndex, const double* inPartials) {
  Instance* in = Get(instance);
  if (!in) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
  if (tipIndex < 0 || tipIndex >= in->tips || tipIndex >= in->partials) return BEAGLE_ERROR_OUT_OF_RANGE;
  // one [pattern][state] block, replicated over categories like BEAGLE does
  for (int c = 0; c < in->categories; c++)
    if (!Upload(in->d_partials + (size_t)tipIndex * in->plv() + (size_t)c * in->patterns * S, inPartials,
                (size_t)in->patterns * S))
      return BEAGLE_ERROR_GENERAL;
  in->tip_is_compact[tipIndex] = 0;
  return BEAGLE_SUCCESS;
}

int beagleSetPartials(int instance, int bufferIndex, const double* inPartials) {
  Instance* in = Get(instance);
  if (!in) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
  if (bufferIndex < 0 || bufferIndex >= in->partials) return BEAGLE_ERROR_OUT_OF_RANGE;
  if (!Upload(in->d_partials + (size_t)bufferIndex * in->plv(), inPartials, in->plv())) return BEAGLE_ERROR_GENERAL;
  if (bufferIndex < in->tips) in->tip_is_compact[bufferIndex] = 0;
  return BEAGLE_SUCCESS;
}

int beagleSetPatternWeights(int instance, const double* w) {
  Instance* in = Get(instance);
  if (!in) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
  return Upload(in->d_weights, w, in->patterns) ? BEAGLE_SUCCESS : BEAGLE_ERROR_GENERAL;
}

int beagleSetCategoryWeights(int instance, int index, const double* w) {
  Instance* in = Get(instance);
  if (!in) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
  if (index != 0) return BEAGLE_ERROR_OUT_OF_RANGE;
  return Upload(in->d_catw, w, in->categories) ? BEAGLE_SUCCESS : BEAGLE_ERROR_GENERAL;
}

int beagleSetCategoryRates(int instance, const double* r) {
  Instance* in = Get(instance);
  if (!in) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
  return Upload(in->d_catr, r, in->categories) ? BEAGLE_SUCCESS : BEAGLE_ERROR_GENERAL;
}

int beagleSetStateFrequencies(int instance, int index, const double* f) {
  Instance* in = Get(instance);
  if (!in) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
  if (index != 0) return BEAGLE_ERROR_OUT_OF_RANGE;
  return Upload(in->d_freq, f, S) ? BEAGLE_SUCCESS : BEAGLE_ERROR_GENERAL;
}

int beagleSetEigenDecomposition(int instance, int eigenIndex, const double* V, const double* Vinv,
                                const double* lambda) {
  Instance* in = Get(instance);
  if (!in) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
  if (eigenIndex != 0) return BEAGLE_ERROR_OUT_OF_RANGE;
  return (Upload(in->d_eigen, V, 16) && Upload(in->d_eigen + 16, Vinv, 16) && Upload(in->d_eigen + 32, lambda, 4))
             ? BEAGLE_SUCCESS
             : BEAGLE_ERROR_GENERAL;
}

int beagleUpdateTransitionMatrices(int instance, int eigenIndex, const int* probabilityIndices, const int* d1,
                                   const int* d2, const double* edgeLengths, int count) {
  Instance* in = Get(instance);
  if (!in) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
  if (eigenIndex != 0) return BEAGLE_ERROR_OUT_OF_RANGE;
  if (d1 != nullptr || d2 != nullptr) return BEAGLE_ERROR_NO_IMPLEMENTATION;  // bito passes nullptr (:321-322)
  for (int i = 0; i < count; i++)
    if (probabilityIndices[i] < 0 || probabilityIndices[i] >= in->matrices) return BEAGLE_ERROR_OUT_OF_RANGE;
  if (!ReserveIdx(in, count) || !ReserveOut(in, count)) return BEAGLE_ERROR_OUT_OF_MEMORY;
  if (!Upload(in->d_idx, probabilityIndices, count) || !Upload(in->d_out, edgeLengths, count)) return BEAGLE_ERROR_GENERAL;
  const int total = count * in->categories * 16;
  hipLaunchKernelGGL(matrices_kernel, dim3((total + 255) / 256), dim3(256), 0, 0, in->d_eigen, in->d_catr, in->d_idx,
                     in->d_out, count, in->categories, in->d_mats);
  return Ok(hipDeviceSynchronize()) ? BEAGLE_SUCCESS : BEAGLE_ERROR_GENERAL;
}

int beagleResetScaleFactors(int instance, int cumulativeScaleIndex) {
  Instance* in = Get(instance);
  if (!in) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
  if (cumulativeScaleIndex < 0 || cumulativeScaleIndex >= in->scales) return BEAGLE_ERROR_OUT_OF_RANGE;
  return Ok(hipMemset(in->d_scale + (size_t)cumulativeScaleIndex * in->patterns, 0, in->patterns * sizeof(double)))
             ? BEAGLE_SUCCESS
             : BEAGLE_ERROR_GENERAL;
}

int beagleUpdatePartials(const int instance, const BeagleOperation* ops, int count, int cumulativeScaleIndex) {
  return RunOps(instance, ops, count, cumulativeScaleIndex, 0);
}

int beagleUpdatePrePartials(const int instance, const BeagleOperation* ops, int count, int cumulativeScaleIndex) {
  return RunOps(instance, ops, count, cumulativeScaleIndex, 1);
}

int beagleSetDifferentialMatrix(int instance, int matrixIndex, const double* inMatrix) {
  Instance* in = Get(instance);
  if (!in) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
  if (matrixIndex < 0 || matrixIndex >= in->matrices) return BEAGLE_ERROR_OUT_OF_RANGE;
  return Upload(in->d_mats + (size_t)matrixIndex * in->categories * 16, inMatrix, (size_t)in->categories * 16)
             ? BEAGLE_SUCCESS
             : BEAGLE_ERROR_GENERAL;
}

int beagleCalculateEdgeDerivatives(int instance, const int* postIdx, const int* preIdx, const int* dmatIdx,
                                   const int* catWeightsIdx, int count, double* outPerSite, double* outSum,
                                   double* outSumSq) {
  Instance* in = Get(instance);
  if (!in) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
  if (outPerSite != nullptr || outSumSq != nullptr) return BEAGLE_ERROR_NO_IMPLEMENTATION;  // bito passes nullptr (:158-160)
  if (catWeightsIdx && catWeightsIdx[0] != 0) return BEAGLE_ERROR_OUT_OF_RANGE;
  if (count <= 0 || !outSum) return BEAGLE_SUCCESS;
  for (int i = 0; i < count; i++)
    if (postIdx[i] < 0 || postIdx[i] >= in->partials || preIdx[i] < 0 || preIdx[i] >= in->partials ||
        dmatIdx[i] < 0 || dmatIdx[i] >= in->matrices)
      return BEAGLE_ERROR_OUT_OF_RANGE;
  std::vector<int> lists(3 * (size_t)count);
  std::memcpy(lists.data(), postIdx, count * sizeof(int));
  std::memcpy(lists.data() + count, preIdx, count * sizeof(int));
  std::memcpy(lists.data() + 2 * count, dmatIdx, count * sizeof(int));
  char* d_compact = nullptr;
  if (!ReserveIdx(in, lists.size() + (in->tips + 3) / 4 + 1) || !ReserveOut(in, count)) return BEAGLE_ERROR_OUT_OF_MEMORY;
  if (!Upload(in->d_idx, lists.data(), lists.size())) return BEAGLE_ERROR_GENERAL;
  d_compact = reinterpret_cast<char*>(in->d_idx + lists.size());
  if (!Ok(hipMemcpy(d_compact, in->tip_is_compact.data(), in->tips, hipMemcpyHostToDevice))) return BEAGLE_ERROR_GENERAL;
  hipLaunchKernelGGL(edge_derivative_kernel, dim3(count), dim3(256), 0, 0, in->d_partials, in->d_states, in->d_mats,
                     in->d_weights, in->d_catw, in->d_idx, count, in->patterns, in->categories, in->plv(), in->tips,
                     d_compact, in->d_out);
  return Ok(hipMemcpy(outSum, in->d_out, count * sizeof(double), hipMemcpyDeviceToHost)) ? BEAGLE_SUCCESS
                                                                                         : BEAGLE_ERROR_GENERAL;
}

int beagleCalculateRootLogLikelihoods(int instance, const int* bufferIndices, const int* catWeightsIdx,
                                      const int* freqIdx, const int* cumulativeScaleIndices, int count,
                                      double* outSumLogLikelihood) {
  Instance* in = Get(instance);
  if (!in) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
  if (count != 1) return BEAGLE_ERROR_NO_IMPLEMENTATION;  // bito always integrates one buffer (beagle_accessories.hpp:28)
  if ((catWeightsIdx && catWeightsIdx[0] != 0) || (freqIdx && freqIdx[0] != 0)) return BEAGLE_ERROR_OUT_OF_RANGE;
  const int root = bufferIndices[0];
  if (root < 0 || root >= in->partials) return BEAGLE_ERROR_OUT_OF_RANGE;
  const int cum = cumulativeScaleIndices ? cumulativeScaleIndices[0] : BEAGLE_OP_NONE;
  if (cum != BEAGLE_OP_NONE && (cum < 0 || cum >= in->scales)) return BEAGLE_ERROR_OUT_OF_RANGE;
  if (!ReserveOut(in, 1)) return BEAGLE_ERROR_OUT_OF_MEMORY;
  hipLaunchKernelGGL(root_kernel, dim3(1), dim3(256), 0, 0, in->d_partials + (size_t)root * in->plv(), in->d_weights,
                     in->d_catw, in->d_freq, cum >= 0 ? in->d_scale + (size_t)cum * in->patterns : nullptr,
                     in->patterns, in->categories, in->d_out);
  return Ok(hipMemcpy(outSumLogLikelihood, in->d_out, sizeof(double), hipMemcpyDeviceToHost)) ? BEAGLE_SUCCESS
                                                                                             : BEAGLE_ERROR_GENERAL;
}

}  // extern "C"
