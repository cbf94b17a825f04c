// Exact percentiles of non-negative float arrays on gfx950 (MI355X): radix select on the float bit
// patterns instead of a sort.
//
// Replaces np.percentile(x, p) (method 'linear') as the reference's analysis scripts call it:
//   elwise_error/onnx_error_cli.py:113-114        P99 of the float32 |error| array (K*C values)
//   MSE&P99/quant_error_multi_weight.py:85         P99 of the float64 per-vector MSE list
//   MSE&P99/quant_error_single_weight.py:79        the same, row form
//
// The bits of a non-negative IEEE float order like unsigned integers.  Keys are widened to 63 bits
// (float32 bits << 32) and resolved most-significant digit first:
//   level 0: 15 bits, histogram of EVERY element in LDS (32 768 x u32 = 128 KiB, one 1024-thread
//            workgroup per CU), merged into a global histogram with one atomic per non-empty bin;
//   level 1..3: 16 bits each, only elements whose higher bits match the prefix found so far take part
//            (a fraction of a percent); each workgroup aggregates them in a small LDS hash table before
//            touching the global histogram, so heavy duplicates cost no global atomic traffic.
// A float32 array needs levels 0 and 1 -- two passes over the data; float64 needs four.
// Both order statistics numpy interpolates between (ranks k and k + 1) are tracked, each with its own
// prefix once they part.  The final interpolation repeats numpy 2.2's arithmetic in the array's dtype.
#include "mq_common.h"

#define SEL_TOP_BINS 32768
#define SEL_DIGIT_BINS 65536
#define SEL_HASH 2048

struct SelState {
  uint64_t prefix[2];  // resolved high bits of the two keys, left-aligned in 63 bits
  uint64_t rank[2];    // remaining rank inside the current prefix
  int32_t same;        // 1 while both ranks share one prefix
  int32_t pad;
};

struct SelWs {
  SelState st;
  uint32_t pad[48];
  uint32_t top[SEL_TOP_BINS];
  uint32_t digit[3][2][SEL_DIGIT_BINS];  // [level - 1][rank][bin]; cleared once per selection
};

__device__ __forceinline__ uint64_t key_of(float v) { return (uint64_t)(f_bits(v) & 0x7fffffffu) << 32; }
__device__ __forceinline__ uint64_t key_of(double v) { return __builtin_bit_cast(uint64_t, v) & 0x7fffffffffffffffull; }

// ---- level 0 -------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void k_sel_hist_top(const T* x, int64_t n, uint32_t* ghist) {
  extern __shared__ uint32_t lh[];
  for (int i = threadIdx.x; i < SEL_TOP_BINS; i += blockDim.x) lh[i] = 0;
  __syncthreads();
  constexpr int V = 16 / sizeof(T);  // elements per 16-byte load
  const int64_t nvec = n / V;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t v = t0; v < nvec; v += step) {
    u32x4 raw = reinterpret_cast<const u32x4*>(x)[v];
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) atomicAdd(&lh[(raw[i] >> 16) & 0x7fffu], 1u);
    } else {
      atomicAdd(&lh[(raw[1] >> 16) & 0x7fffu], 1u);
      atomicAdd(&lh[(raw[3] >> 16) & 0x7fffu], 1u);
    }
  }
  for (int64_t i = nvec * V + t0; i < n; i += step) atomicAdd(&lh[(uint32_t)(key_of(x[i]) >> 48)], 1u);
  __syncthreads();
  for (int i = threadIdx.x; i < SEL_TOP_BINS; i += blockDim.x) {
    uint32_t c = lh[i];
    if (c) atomicAdd(&ghist[i], c);
  }
}

// ---- scan of a histogram: which bin holds each remaining rank --------------------------------------
// One 1024-thread workgroup.  Thread t owns PER consecutive bins and sums them with 16-byte loads; the
// prefix over threads is a wave scan plus one LDS exchange; the one thread whose range holds a rank then
// walks its bins again.  With finish_dtype != 0 the last level also performs numpy's interpolation
// (lib/_function_base_impl.py::_lerp) in that dtype:
//   r = a + (b - a) * t;  where t >= 0.5:  r = b - (b - a) * (1 - t)
struct FinishArgs {
  int32_t dtype;  // -1: no finish here; MQ_F32 / MQ_F64
  double gamma;
  void* out;
  void* out_pair;
};

template <int PER>
__device__ __forceinline__ void scan_hist(const uint32_t* h, int shift, const SelState& st, int r_lo, int r_hi,
                                          SelState* res, uint64_t* wsum /*[16]*/) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const u32x4* hv = reinterpret_cast<const u32x4*>(h) + threadIdx.x * (PER / 4);
  uint64_t sum = 0;
#pragma unroll
  for (int i = 0; i < PER / 4; ++i) {
    u32x4 v = hv[i];
    sum += (uint64_t)v[0] + v[1] + v[2] + v[3];
  }
  uint64_t inc = sum;  // inclusive scan inside the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint64_t o = __shfl_up(inc, off);
    if (lane >= off) inc += o;
  }
  __syncthreads();
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  uint64_t before = inc - sum;
  for (int w = 0; w < wv; ++w) before += wsum[w];
  for (int r = r_lo; r <= r_hi; ++r) {
    const uint64_t rk = st.rank[r];
    if (rk >= before && rk < before + sum) {  // exactly one thread owns the rank: it re-reads its bins
      uint64_t acc = before;
      int found = -1;
      for (int i = 0; i < PER / 4 && found < 0; ++i) {
        u32x4 v = hv[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (found < 0 && rk < acc + v[e]) { found = 4 * i + e; res->rank[r] = rk - acc; }
          acc += v[e];
        }
      }
      res->prefix[r] = st.prefix[r] | ((uint64_t)(threadIdx.x * PER + found) << shift);
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void k_sel_scan(SelWs* ws, int level, uint64_t k0, uint64_t k1, FinishArgs fin) {
  __shared__ uint64_t wsum[16];
  __shared__ SelState st, res;
  if (threadIdx.x == 0) {
    if (level == 0) {  // the ranks arrive with the first scan
      SelState s0 = {};
      s0.rank[0] = k0;
      s0.rank[1] = k1;
      s0.same = 1;
      st = s0;
    } else {
      st = ws->st;
    }
    res = st;
  }
  __syncthreads();
  const int shift = 48 - 16 * level;
  if (level == 0) {
    scan_hist<SEL_TOP_BINS / 1024>(ws->top, shift, st, 0, 1, &res, wsum);
  } else if (st.same) {
    scan_hist<SEL_DIGIT_BINS / 1024>(ws->digit[level - 1][0], shift, st, 0, 1, &res, wsum);
  } else {
    scan_hist<SEL_DIGIT_BINS / 1024>(ws->digit[level - 1][0], shift, st, 0, 0, &res, wsum);
    scan_hist<SEL_DIGIT_BINS / 1024>(ws->digit[level - 1][1], shift, st, 1, 1, &res, wsum);
  }
  if (threadIdx.x == 0) {
    res.same = res.prefix[0] == res.prefix[1] ? 1 : 0;
    ws->st = res;
    if (fin.dtype == MQ_F32) {
      float a = f_from_bits((uint32_t)(res.prefix[0] >> 32)), b = f_from_bits((uint32_t)(res.prefix[1] >> 32));
      float g = (float)fin.gamma, diff = b - a;
      float r = a + diff * g;
      if (g >= 0.5f) r = b - diff * (1.0f - g);
      if (fin.out) *static_cast<float*>(fin.out) = r;
      if (fin.out_pair) { static_cast<float*>(fin.out_pair)[0] = a; static_cast<float*>(fin.out_pair)[1] = b; }
    } else if (fin.dtype == MQ_F64) {
      double a = __builtin_bit_cast(double, res.prefix[0]), b = __builtin_bit_cast(double, res.prefix[1]);
      double g = fin.gamma, diff = b - a;
      double r = a + diff * g;
      if (g >= 0.5) r = b - diff * (1.0 - g);
      if (fin.out) *static_cast<double*>(fin.out) = r;
      if (fin.out_pair) { static_cast<double*>(fin.out_pair)[0] = a; static_cast<double*>(fin.out_pair)[1] = b; }
    }
  }
}

// ---- levels 1..3 -----------------------------------------------------------------------------------
// Elements whose bits above the digit equal a tracked prefix are counted per digit value; a workgroup
// first aggregates them in an LDS hash table (linear probing), then merges the table into the global
// histogram.  A full table (more distinct values than slots) falls back to global atomics.
template <typename T>
__global__ __launch_bounds__(256) void k_sel_hist_digit(const T* x, int64_t n, SelWs* ws, int level) {
  __shared__ uint32_t hkey[2][SEL_HASH];  // digit + 1, 0 = empty
  __shared__ uint32_t hcnt[2][SEL_HASH];
  for (int i = threadIdx.x; i < 2 * SEL_HASH; i += blockDim.x) { (&hkey[0][0])[i] = 0; (&hcnt[0][0])[i] = 0; }
  __syncthreads();
  const int shift = 48 - 16 * level;
  const uint64_t himask = ~((1ull << (shift + 16)) - 1ull) & 0x7fffffffffffffffull;
  const uint64_t p0 = ws->st.prefix[0], p1 = ws->st.prefix[1];
  const bool same = ws->st.same != 0;
  auto put = [&](uint64_t key) {
    const uint64_t hi = key & himask;
    const bool m0 = hi == (p0 & himask), m1 = !same && hi == (p1 & himask);
    if (!(m0 || m1)) return;
    const int s = m0 ? 0 : 1;
    const uint32_t d = (uint32_t)(key >> shift) & 0xffffu;
    uint32_t h = (d * 2654435761u) >> 21;  // 11 bits
    for (int probe = 0; probe < 64; ++probe) {
      uint32_t old = atomicCAS(&hkey[s][h], 0u, d + 1u);
      if (old == 0u || old == d + 1u) { atomicAdd(&hcnt[s][h], 1u); return; }
      h = (h + 1) & (SEL_HASH - 1);
    }
    atomicAdd(&ws->digit[level - 1][s][d], 1u);
  };
  constexpr int V = 16 / sizeof(T);
  const int64_t nvec = n / V;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t v = t0; v < nvec; v += step) {
    u32x4 raw = reinterpret_cast<const u32x4*>(x)[v];
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) put((uint64_t)(raw[i] & 0x7fffffffu) << 32);
    } else {
      put((((uint64_t)raw[1] << 32) | raw[0]) & 0x7fffffffffffffffull);
      put((((uint64_t)raw[3] << 32) | raw[2]) & 0x7fffffffffffffffull);
    }
  }
  for (int64_t i = nvec * V + t0; i < n; i += step) put(key_of(x[i]));
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * SEL_HASH; i += blockDim.x) {
    const int s = i / SEL_HASH, h = i % SEL_HASH;
    if (hkey[s][h]) atomicAdd(&ws->digit[level - 1][s][hkey[s][h] - 1u], hcnt[s][h]);
  }
}

// ================================================================================================
extern "C" size_t mq_percentile_workspace_bytes(void) { return sizeof(SelWs); }

// Byte offset / element count (uint32) of the histogram that `level` accumulates into, for callers that
// merge histograms across devices (all-reduce SUM) between accumulate and scan.
extern "C" void mq_select_layout(int32_t level, size_t* offset, size_t* count) {
  if (level <= 0) {
    if (offset) *offset = offsetof(SelWs, top);
    if (count) *count = SEL_TOP_BINS;
  } else {
    if (offset) *offset = offsetof(SelWs, digit) + (size_t)(level - 1) * 2 * SEL_DIGIT_BINS * sizeof(uint32_t);
    if (count) *count = 2 * SEL_DIGIT_BINS;
  }
}

// numpy 2.2's index arithmetic for method 'linear', in the array's dtype:
//   q = pct / dtype(100);  vi = (n - 1) * q;  k0 = floor(vi), k1 = k0 + 1, gamma = vi - k0
//   (vi >= n - 1: both ranks are the last element)
extern "C" int mq_percentile_index(int32_t dtype, int64_t n, double pct, int64_t* k0, int64_t* k1, double* gamma) {
  if (n <= 0 || !(pct >= 0.0 && pct <= 100.0) || !k0 || !k1 || !gamma) return MQ_ERR_ARG;
  if (dtype == MQ_F32) {
    float q = (float)pct / 100.0f;
    float vi = (float)(n - 1) * q;
    float prev = __builtin_floorf(vi);
    *gamma = (double)(vi - prev);
    *k0 = (int64_t)prev;
    if (vi >= (float)(n - 1)) *k0 = n - 1;
  } else if (dtype == MQ_F64) {
    double q = pct / 100.0;
    double vi = (double)(n - 1) * q;
    double prev = __builtin_floor(vi);
    *gamma = vi - prev;
    *k0 = (int64_t)prev;
    if (vi >= (double)(n - 1)) *k0 = n - 1;
  } else {
    return MQ_ERR_ARG;
  }
  if (*k0 > n - 1) *k0 = n - 1;
  *k1 = *k0 + 1 > n - 1 ? n - 1 : *k0 + 1;
  return MQ_OK;
}

static int check_ws(const void* x, int32_t dtype, int64_t n, void* workspace, size_t workspace_bytes) {
  if (n < 0 || (n > 0 && !x)) return MQ_ERR_ARG;
  if (dtype != MQ_F32 && dtype != MQ_F64) return MQ_ERR_ARG;
  if (!workspace || workspace_bytes < sizeof(SelWs)) return MQ_ERR_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(workspace) % 16) return MQ_ERR_ALIGN;
  return MQ_OK;
}

// Step 1: clear every histogram of the workspace (one fill).
extern "C" int mq_select_begin(void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!workspace || workspace_bytes < sizeof(SelWs)) return MQ_ERR_WORKSPACE;
  MQ_HIP(hipMemsetAsync(workspace, 0, sizeof(SelWs), static_cast<hipStream_t>(hip_stream)));
  return MQ_OK;
}

// Step 2 (level 0) / step 4 (level >= 1): add one array's contribution to that level's histogram.
// May be called for several arrays before the scan.
extern "C" int mq_select_accumulate(const void* x, int32_t dtype, int64_t n, int32_t level, void* workspace,
                                    size_t workspace_bytes, void* hip_stream) {
  int rc = check_ws(x, dtype, n, workspace, workspace_bytes);
  if (rc) return rc;
  if (level < 0 || level > 3) return MQ_ERR_ARG;
  if (n == 0) return MQ_OK;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  SelWs* ws = static_cast<SelWs*>(workspace);
  const int V = dtype == MQ_F32 ? 4 : 2;
  if (level == 0) {
    int64_t nb = (n / V + 1024 * 4 - 1) / (1024 * 4);
    unsigned grid = (unsigned)(nb < 1 ? 1 : (nb > 256 ? 256 : nb));
    if (dtype == MQ_F32) {
      MQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sel_hist_top<float>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, SEL_TOP_BINS * 4));
      hipLaunchKernelGGL((k_sel_hist_top<float>), dim3(grid), dim3(1024), SEL_TOP_BINS * 4, st,
                         static_cast<const float*>(x), n, ws->top);
    } else {
      MQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sel_hist_top<double>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, SEL_TOP_BINS * 4));
      hipLaunchKernelGGL((k_sel_hist_top<double>), dim3(grid), dim3(1024), SEL_TOP_BINS * 4, st,
                         static_cast<const double*>(x), n, ws->top);
    }
  } else {
    int64_t nb = (n / V + 256 * 4 - 1) / (256 * 4);
    unsigned grid = (unsigned)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
    if (dtype == MQ_F32)
      hipLaunchKernelGGL((k_sel_hist_digit<float>), dim3(grid), dim3(256), 0, st, static_cast<const float*>(x), n, ws, level);
    else
      hipLaunchKernelGGL((k_sel_hist_digit<double>), dim3(grid), dim3(256), 0, st, static_cast<const double*>(x), n, ws, level);
  }
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}

// Step 3 / 5: locate the ranks in the (merged) histogram of `level` and extend the prefixes.  The ranks
// k0 <= k1 are read at level 0.  With finish != 0 (pass it at the last level: 1 for float32, 3 for
// float64) the two order statistics and numpy's interpolation between them are written, in `dtype`, to
// out (one value) and out_pair (two values); either may be NULL.
extern "C" int mq_select_scan(int32_t level, int64_t k0, int64_t k1, int32_t finish, int32_t dtype, double gamma,
                              void* out, void* out_pair, void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (level < 0 || level > 3 || (level == 0 && (k0 < 0 || k1 < k0))) return MQ_ERR_ARG;
  if (finish && dtype != MQ_F32 && dtype != MQ_F64) return MQ_ERR_ARG;
  if (!workspace || workspace_bytes < sizeof(SelWs)) return MQ_ERR_WORKSPACE;
  FinishArgs fin = {finish ? dtype : -1, gamma, out, out_pair};
  hipLaunchKernelGGL(k_sel_scan, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(hip_stream),
                     static_cast<SelWs*>(workspace), (int)level, (uint64_t)k0, (uint64_t)k1, fin);
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}

// np.percentile(x, pct) for a non-negative float32 / float64 array in device memory (the sign bit is
// ignored, i.e. the percentile of |x| is returned): the steps above on one array -- 5 launches for
// float32 (fill, histogram, scan, matched-digit histogram, scan + interpolation), 9 for float64.
extern "C" int mq_percentile(const void* x, int32_t dtype, int64_t n, double pct, void* out, void* out_pair,
                             void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!x || n <= 0 || (!out && !out_pair)) return MQ_ERR_ARG;
  int rc = check_ws(x, dtype, n, workspace, workspace_bytes);
  if (rc) return rc;
  int64_t k0, k1;
  double gamma;
  if ((rc = mq_percentile_index(dtype, n, pct, &k0, &k1, &gamma))) return rc;
  if ((rc = mq_select_begin(workspace, workspace_bytes, hip_stream))) return rc;
  const int levels = dtype == MQ_F32 ? 1 : 3;
  for (int level = 0; level <= levels; ++level) {
    if ((rc = mq_select_accumulate(x, dtype, n, level, workspace, workspace_bytes, hip_stream))) return rc;
    if ((rc = mq_select_scan(level, k0, k1, level == levels, dtype, gamma, out, out_pair, workspace, workspace_bytes,
                             hip_stream)))
      return rc;
  }
  return MQ_OK;
}
