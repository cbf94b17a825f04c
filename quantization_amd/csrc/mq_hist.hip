// Minimum / maximum and np.histogram counts of a device array on gfx950 (MI355X): the scans behind the
// reference's plots -- the 200-bin weight histogram with its two-pass global range
// (/root/reference/plot_matmul_weight_rev0.py:51-101, :221-238), the 200-bin |error| histogram
// (elwise_error/onnx_error_cli.py:208-215, MSE&P99/quant_error_single_weight.py:81) and the 50-bin per-vector
// MSE histograms (MSE&P99/plot.py:81-90, quant_error_single_weight.py:120-124).
//
// np.histogram decides every count by comparing the value with the bin edges (its uniform-bin shortcut ends with
// the same comparisons), so the counts are exact once the edges are numpy's: the host computes the edges with
// numpy (a few hundred values), the kernels do the two scans.  Both stream the array once with 16-byte loads
// (HBM-bound: 2 / 4 / 8 bytes per element, no output to speak of); counts go through per-wave LDS histograms.
#include "mq_common.h"

template <typename T>
struct Elem;
template <>
struct Elem<uint16_t> {  // float16 bits
  static constexpr int kVec = 8;
  static __device__ __forceinline__ double get(uint16_t b) { return (double)(float)h_from_bits(b); }
};
template <>
struct Elem<float> {
  static constexpr int kVec = 4;
  static __device__ __forceinline__ double get(float v) { return (double)v; }
};
template <>
struct Elem<double> {
  static constexpr int kVec = 2;
  static __device__ __forceinline__ double get(double v) { return v; }
};

// Calls f(value) for every element of x[0..n): 16-byte loads over the aligned body, scalar loads at the ends.
template <typename T, typename F>
__device__ __forceinline__ void for_each_elem(const T* x, int64_t n, F f) {
  constexpr int V = Elem<T>::kVec;
  const int64_t gt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gn = (int64_t)gridDim.x * blockDim.x;
  // elements before the first 16-byte boundary
  int64_t head = (int64_t)((16 - (reinterpret_cast<uintptr_t>(x) & 15)) & 15) / (int64_t)sizeof(T);
  if (head > n) head = n;
  const int64_t nvec = (n - head) / V;
  for (int64_t i = gt; i < head; i += gn) f(Elem<T>::get(x[i]));
  const u32x4* xv = reinterpret_cast<const u32x4*>(x + head);
  for (int64_t v = gt; v < nvec; v += gn) {
    u32x4 r = xv[v];
    T e[V];
    __builtin_memcpy(e, &r, 16);
#pragma unroll
    for (int k = 0; k < V; ++k) f(Elem<T>::get(e[k]));
  }
  for (int64_t i = head + nvec * V + gt; i < n; i += gn) f(Elem<T>::get(x[i]));
}

// ---- min / max --------------------------------------------------------------------------------------------
// partials[block] = {min, max, nan flag}; np.min / np.max return NaN when the array holds one.
template <typename T>
__global__ __launch_bounds__(256) void k_minmax(const T* x, int64_t n, double* partials) {
  double lo = INFINITY, hi = -INFINITY;
  int nan = 0;
  for_each_elem(x, n, [&](double v) {
    nan |= (v != v);
    lo = v < lo ? v : lo;
    hi = v > hi ? v : hi;
  });
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double a = __shfl_xor(lo, off), b = __shfl_xor(hi, off);
    nan |= __shfl_xor(nan, off);
    lo = a < lo ? a : lo;
    hi = b > hi ? b : hi;
  }
  __shared__ double s_lo[4], s_hi[4];
  __shared__ int s_nan[4];
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_lo[wv] = lo; s_hi[wv] = hi; s_nan[wv] = nan; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i) {
      lo = s_lo[i] < lo ? s_lo[i] : lo;
      hi = s_hi[i] > hi ? s_hi[i] : hi;
      nan |= s_nan[i];
    }
    partials[3 * blockIdx.x] = lo;
    partials[3 * blockIdx.x + 1] = hi;
    partials[3 * blockIdx.x + 2] = nan ? 1.0 : 0.0;
  }
}

// one workgroup: partials -> out[0] = min, out[1] = max; `merge` folds the previous out in (running range
// over several arrays)
__global__ __launch_bounds__(256) void k_minmax_final(const double* partials, int nblocks, double* out, int merge) {
  double lo = INFINITY, hi = -INFINITY, nan = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x) {
    double a = partials[3 * i], b = partials[3 * i + 1];
    lo = a < lo ? a : lo;
    hi = b > hi ? b : hi;
    nan += partials[3 * i + 2];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double a = __shfl_xor(lo, off), b = __shfl_xor(hi, off);
    nan += __shfl_xor(nan, off);
    lo = a < lo ? a : lo;
    hi = b > hi ? b : hi;
  }
  __shared__ double s_lo[4], s_hi[4], s_nan[4];
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_lo[wv] = lo; s_hi[wv] = hi; s_nan[wv] = nan; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i) {
      lo = s_lo[i] < lo ? s_lo[i] : lo;
      hi = s_hi[i] > hi ? s_hi[i] : hi;
      nan += s_nan[i];
    }
    if (merge) {
      const double plo = out[0], phi = out[1];
      if (plo != plo || phi != phi) nan = 1.0;
      lo = plo < lo ? plo : lo;
      hi = phi > hi ? phi : hi;
    }
    out[0] = nan != 0.0 ? NAN : lo;
    out[1] = nan != 0.0 ? NAN : hi;
  }
}

// ---- histogram --------------------------------------------------------------------------------------------
// counts[i] += #{ edges[i] <= v < edges[i+1] }, the last bin closed on the right (np.histogram); values outside
// [edges[0], edges[nbins]] and NaN are not counted.  Dynamic LDS: (nbins + 1) doubles, then `copies` arrays of
// nbins counters (one per wave when they fit, to spread the atomics of peaked distributions).
template <typename T>
__global__ __launch_bounds__(256) void k_histogram(const T* x, int64_t n, const double* edges, int nbins, int copies,
                                                   unsigned long long* counts) {
  extern __shared__ double hist_lds[];
  __shared__ uint32_t s_dev;
  double* s_e = hist_lds;
  uint32_t* s_c = reinterpret_cast<uint32_t*>(hist_lds + nbins + 1);
  if (threadIdx.x == 0) s_dev = 0u;
  for (int i = threadIdx.x; i <= nbins; i += blockDim.x) s_e[i] = edges[i];
  for (int i = threadIdx.x; i < nbins * copies; i += blockDim.x) s_c[i] = 0u;
  __syncthreads();
  const double e0 = s_e[0], eN = s_e[nbins];
  const double width = (eN - e0) / (double)nbins;
  const double inv = (double)nbins / (eN - e0);  // first guess of the bin; inf / NaN guesses are clamped below
  // How far, in bin widths, the edges are from the uniform grid e0 + i * width (np.linspace edges: a few 1e-6
  // for float32 edges).  A value whose position g = (v - e0) / width is farther than that (plus the rounding of
  // g itself) from an integer lies in bin floor(g) for certain; only the others are compared with the edges.
  // Non-uniform edges give a margin >= 0.5: every value is compared.
  {
    float dev = 0.f;
    for (int i = threadIdx.x; i <= nbins; i += blockDim.x) {
      double d = __builtin_fabs(s_e[i] - (e0 + (double)i * width)) * inv;
      float f = d < 1.0 ? (float)d : 1.0f;  // (NaN -> 1)
      f = f == f ? f : 1.0f;
      dev = f > dev ? f : dev;
    }
    atomicMax(&s_dev, f_bits(dev));  // non-negative floats order like their bits
  }
  __syncthreads();
  const double margin = (double)f_from_bits(s_dev) * 1.0001 + 1e-9;
  uint32_t* mine = s_c + (copies > 1 ? (int)(threadIdx.x >> 6) * nbins : 0);
  for_each_elem(x, n, [&](double v) {
    if (!(v >= e0 && v <= eN)) return;
    const double g = (v - e0) * inv;
    int idx = g >= 0.0 && g < (double)nbins ? (int)g : (g >= (double)nbins ? nbins - 1 : 0);
    const double frac = g - (double)idx;
    if (!(frac > margin && frac < 1.0 - margin)) {
      while (idx > 0 && v < s_e[idx]) --idx;
      while (idx < nbins - 1 && v >= s_e[idx + 1]) ++idx;
    }
    atomicAdd(&mine[idx], 1u);
  });
  __syncthreads();
  for (int i = threadIdx.x; i < nbins; i += blockDim.x) {
    unsigned long long c = 0;
    for (int k = 0; k < copies; ++k) c += s_c[k * nbins + i];
    if (c) atomicAdd(&counts[i], c);
  }
}

// ================================================================================================
static const int kScanMaxGrid = 2048;
extern "C" size_t mq_minmax_workspace_bytes(void) { return (size_t)kScanMaxGrid * 3 * sizeof(double); }

static int scan_grid(int64_t n, int per_thread) {
  int64_t b = (n + 256ll * per_thread - 1) / (256ll * per_thread);
  return (int)(b < 1 ? 1 : (b > kScanMaxGrid ? kScanMaxGrid : b));
}
static bool scan_dtype_ok(int32_t dtype) { return dtype == MQ_F16 || dtype == MQ_F32 || dtype == MQ_F64; }
static size_t scan_elem_size(int32_t dtype) { return dtype == MQ_F16 ? 2 : (dtype == MQ_F32 ? 4 : 8); }

extern "C" int mq_minmax(const void* x, int32_t dtype, int64_t n, int32_t merge, double* out, void* workspace,
                         size_t workspace_bytes, void* hip_stream) {
  if (!scan_dtype_ok(dtype) || n < 0 || !out) return MQ_ERR_ARG;
  if (n == 0) return merge ? MQ_OK : MQ_ERR_ARG;  // no minimum of an empty array
  if (!x) return MQ_ERR_ARG;
  if (!workspace || workspace_bytes < mq_minmax_workspace_bytes()) return MQ_ERR_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(x) % scan_elem_size(dtype) || reinterpret_cast<uintptr_t>(out) % 8 ||
      reinterpret_cast<uintptr_t>(workspace) % 8)
    return MQ_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  double* partials = static_cast<double*>(workspace);
  const int grid = scan_grid(n, 64);
  if (dtype == MQ_F16) hipLaunchKernelGGL(k_minmax<uint16_t>, dim3(grid), dim3(256), 0, st, static_cast<const uint16_t*>(x), n, partials);
  else if (dtype == MQ_F32) hipLaunchKernelGGL(k_minmax<float>, dim3(grid), dim3(256), 0, st, static_cast<const float*>(x), n, partials);
  else hipLaunchKernelGGL(k_minmax<double>, dim3(grid), dim3(256), 0, st, static_cast<const double*>(x), n, partials);
  hipLaunchKernelGGL(k_minmax_final, dim3(1), dim3(256), 0, st, partials, grid, out, merge ? 1 : 0);
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}

extern "C" int mq_histogram(const void* x, int32_t dtype, int64_t n, const double* edges, int32_t nbins,
                            int64_t* counts, void* hip_stream) {
  if (!scan_dtype_ok(dtype) || n < 0 || !edges || !counts || nbins < 1 || nbins > 4096) return MQ_ERR_ARG;
  if (n == 0) return MQ_OK;
  if (!x) return MQ_ERR_ARG;
  if (reinterpret_cast<uintptr_t>(x) % scan_elem_size(dtype) || reinterpret_cast<uintptr_t>(edges) % 8 ||
      reinterpret_cast<uintptr_t>(counts) % 8)
    return MQ_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const int copies = nbins <= 1024 ? 4 : 1;
  const size_t lds = (size_t)(nbins + 1) * sizeof(double) + (size_t)copies * nbins * sizeof(uint32_t);
  const int grid = scan_grid(n, 64);
  unsigned long long* c = reinterpret_cast<unsigned long long*>(counts);
  if (dtype == MQ_F16) hipLaunchKernelGGL(k_histogram<uint16_t>, dim3(grid), dim3(256), lds, st, static_cast<const uint16_t*>(x), n, edges, nbins, copies, c);
  else if (dtype == MQ_F32) hipLaunchKernelGGL(k_histogram<float>, dim3(grid), dim3(256), lds, st, static_cast<const float*>(x), n, edges, nbins, copies, c);
  else hipLaunchKernelGGL(k_histogram<double>, dim3(grid), dim3(256), lds, st, static_cast<const double*>(x), n, edges, nbins, copies, c);
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}
