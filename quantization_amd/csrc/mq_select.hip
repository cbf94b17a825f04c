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
//            workgroup per CU), merged into a global histogram with atomics on the non-empty bins;
//   level 1..3: 16 bits each, only elements whose higher bits match the prefix found so far take part
//            (a fraction of a percent); each workgroup aggregates them in a small LDS hash table before
//            touching the global histogram, so heavy duplicates cost no global atomic traffic.
// Every histogram has a coarse companion (one counter per 256 bins, updated by the same flush), so the
// scan that locates a rank reads 2 KiB instead of the whole histogram.
// A float32 array needs levels 0 and 1 -- two passes over the data; float64 needs four.
// Both order statistics numpy interpolates between (ranks k and k + 1) are tracked, each with its own
// prefix once they part.  The final interpolation repeats numpy 2.2's arithmetic in the array's dtype.
// All kernels take a batch of arrays (blockIdx.y = array), each with its own workspace slice.
#include "mq_common.h"

#define SEL_TOP_BINS 32768
#define SEL_DIGIT_BINS 65536
#define SEL_HASH 2048
#define SEL_MAX_BATCH 32

struct SelState {
  uint64_t prefix[2];  // resolved high bits of the two keys, left-aligned in 63 bits
  uint64_t rank[2];    // remaining rank inside the current prefix
  int32_t same;        // 1 while both ranks share one prefix
  int32_t pad;
};

struct SelLevel {  // histograms of one level >= 1, for the two tracked ranks
  uint32_t coarse[2][SEL_DIGIT_BINS / 256];
  uint32_t fine[2][SEL_DIGIT_BINS];
};

struct SelWs {
  SelState st;
  uint32_t pad[52];
  uint32_t top_coarse[SEL_TOP_BINS / 256];
  uint32_t top[SEL_TOP_BINS];
  SelLevel lvl[3];  // cleared once per selection, only as many as the dtype needs
};

struct SelBatch {
  const void* x[SEL_MAX_BATCH];
  char* ws;          // SelWs of array 0
  int64_t ws_stride; // bytes between workspace slices (0: all arrays accumulate into one selection)
  int64_t n;
};
__device__ __forceinline__ SelWs* batch_ws(const SelBatch& b, int y) {
  return reinterpret_cast<SelWs*>(b.ws + (int64_t)y * b.ws_stride);
}

__device__ __forceinline__ uint64_t key_of(float v) { return (uint64_t)(f_bits(v) & 0x7fffffffu) << 32; }
__device__ __forceinline__ uint64_t key_of(double v) { return __builtin_bit_cast(uint64_t, v) & 0x7fffffffffffffffull; }

// ---- level 0 -------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void k_sel_hist_top(const SelBatch b) {
  extern __shared__ uint32_t lh[];
  const T* x = static_cast<const T*>(b.x[blockIdx.y]);
  SelWs* ws = batch_ws(b, blockIdx.y);
  const int64_t n = b.n;
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
    if (c) {
      atomicAdd(&ws->top[i], c);
      atomicAdd(&ws->top_coarse[i >> 8], c);
    }
  }
}

// ---- scan: which bin holds each remaining rank -----------------------------------------------------
// One 256-thread workgroup per array: locate the rank among the coarse counters, then among the 256
// fine bins under that counter.  With fin.dtype >= 0 the last level also performs numpy's interpolation
// (lib/_function_base_impl.py::_lerp) in that dtype:
//   r = a + (b - a) * t;  where t >= 0.5:  r = b - (b - a) * (1 - t)
struct FinishArgs {
  int32_t dtype;  // -1: no finish here; MQ_F32 / MQ_F64
  double gamma;
  char* out;       // [batch] values of dtype
  char* out_pair;  // [batch][2]
};

// Inclusive block scan of one value per thread (256 threads); returns the exclusive prefix of this thread.
__device__ __forceinline__ uint64_t block_excl_scan(uint64_t v, uint64_t* wsum /*[4]*/) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint64_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint64_t o = __shfl_up(inc, off);
    if (lane >= off) inc += o;
  }
  __syncthreads();
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  uint64_t before = inc - v;
  for (int w = 0; w < wv; ++w) before += wsum[w];
  return before;
}

// Finds bin and residual rank of `rk` in fine[ncoarse * 256] with its coarse companion.
__device__ __forceinline__ void locate(const uint32_t* coarse, const uint32_t* fine, int ncoarse, uint64_t rk,
                                       uint64_t* wsum, uint32_t* sh_bin, uint64_t* sh_rank) {
  uint64_t c = (int)threadIdx.x < ncoarse ? coarse[threadIdx.x] : 0;
  uint64_t before = block_excl_scan(c, wsum);
  if (rk >= before && rk < before + c) { *sh_bin = threadIdx.x; *sh_rank = rk - before; }
  __syncthreads();
  const uint32_t cb = *sh_bin;
  const uint64_t r2 = *sh_rank;
  __syncthreads();
  uint64_t f = fine[cb * 256 + threadIdx.x];
  before = block_excl_scan(f, wsum);
  if (r2 >= before && r2 < before + f) { *sh_bin = cb * 256 + threadIdx.x; *sh_rank = r2 - before; }
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_sel_scan(const SelBatch b, int level, uint64_t k0, uint64_t k1, FinishArgs fin) {
  __shared__ uint64_t wsum[4];
  __shared__ uint32_t sh_bin;
  __shared__ uint64_t sh_rank;
  __shared__ SelState st;
  SelWs* ws = batch_ws(b, blockIdx.x);
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
    sh_bin = 0;
    sh_rank = 0;
  }
  __syncthreads();
  const int shift = 48 - 16 * level;
  SelState res = st;
  for (int r = 0; r < 2; ++r) {
    const int s = (level == 0 || st.same) ? 0 : r;  // which histogram this rank lives in
    const uint32_t* coarse = level == 0 ? ws->top_coarse : ws->lvl[level - 1].coarse[s];
    const uint32_t* fine = level == 0 ? ws->top : ws->lvl[level - 1].fine[s];
    locate(coarse, fine, level == 0 ? SEL_TOP_BINS / 256 : SEL_DIGIT_BINS / 256, st.rank[r], wsum, &sh_bin, &sh_rank);
    res.prefix[r] = st.prefix[r] | ((uint64_t)sh_bin << shift);
    res.rank[r] = sh_rank;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    res.same = res.prefix[0] == res.prefix[1] ? 1 : 0;
    ws->st = res;
    if (fin.dtype == MQ_F32) {
      float a = f_from_bits((uint32_t)(res.prefix[0] >> 32)), c = f_from_bits((uint32_t)(res.prefix[1] >> 32));
      float g = (float)fin.gamma, diff = c - a;
      float v = a + diff * g;
      if (g >= 0.5f) v = c - diff * (1.0f - g);
      if (fin.out) reinterpret_cast<float*>(fin.out)[blockIdx.x] = v;
      if (fin.out_pair) { reinterpret_cast<float*>(fin.out_pair)[2 * blockIdx.x] = a; reinterpret_cast<float*>(fin.out_pair)[2 * blockIdx.x + 1] = c; }
    } else if (fin.dtype == MQ_F64) {
      double a = __builtin_bit_cast(double, res.prefix[0]), c = __builtin_bit_cast(double, res.prefix[1]);
      double g = fin.gamma, diff = c - a;
      double v = a + diff * g;
      if (g >= 0.5) v = c - diff * (1.0 - g);
      if (fin.out) reinterpret_cast<double*>(fin.out)[blockIdx.x] = v;
      if (fin.out_pair) { reinterpret_cast<double*>(fin.out_pair)[2 * blockIdx.x] = a; reinterpret_cast<double*>(fin.out_pair)[2 * blockIdx.x + 1] = c; }
    }
  }
}

// ---- levels 1..3 -----------------------------------------------------------------------------------
// Elements whose bits above the digit equal a tracked prefix are counted per digit value; a workgroup
// first aggregates them in an LDS hash table (linear probing), then merges the table into the global
// histogram.  A full table (more distinct values than slots) falls back to global atomics.
template <typename T>
__global__ __launch_bounds__(256) void k_sel_hist_digit(const SelBatch b, int level) {
  __shared__ uint32_t hkey[2][SEL_HASH];  // digit + 1, 0 = empty
  __shared__ uint32_t hcnt[2][SEL_HASH];
  const T* x = static_cast<const T*>(b.x[blockIdx.y]);
  SelWs* ws = batch_ws(b, blockIdx.y);
  SelLevel* lv = &ws->lvl[level - 1];
  const int64_t n = b.n;
  for (int i = threadIdx.x; i < 2 * SEL_HASH; i += blockDim.x) { (&hkey[0][0])[i] = 0; (&hcnt[0][0])[i] = 0; }
  __syncthreads();
  const int shift = 48 - 16 * level;
  const uint64_t himask = ~((1ull << (shift + 16)) - 1ull) & 0x7fffffffffffffffull;
  const uint64_t p0 = ws->st.prefix[0] & himask, p1 = ws->st.prefix[1] & himask;
  const bool same = ws->st.same != 0;
  auto put = [&](uint64_t key) {
    const uint64_t hi = key & himask;
    const bool m0 = hi == p0, m1 = !same && hi == p1;
    if (!(m0 || m1)) return;
    const int s = m0 ? 0 : 1;
    const uint32_t d = (uint32_t)(key >> shift) & 0xffffu;
    uint32_t h = (d * 2654435761u) >> 21;  // 11 bits
    for (int probe = 0; probe < 64; ++probe) {
      uint32_t old = atomicCAS(&hkey[s][h], 0u, d + 1u);
      if (old == 0u || old == d + 1u) { atomicAdd(&hcnt[s][h], 1u); return; }
      h = (h + 1) & (SEL_HASH - 1);
    }
    atomicAdd(&lv->fine[s][d], 1u);
    atomicAdd(&lv->coarse[s][d >> 8], 1u);
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
    if (hkey[s][h]) {
      const uint32_t d = hkey[s][h] - 1u;
      atomicAdd(&lv->fine[s][d], hcnt[s][h]);
      atomicAdd(&lv->coarse[s][d >> 8], hcnt[s][h]);
    }
  }
}

// ================================================================================================
static size_t ws_used_bytes(int32_t dtype) {  // what one selection of this dtype touches
  return offsetof(SelWs, lvl) + (size_t)(dtype == MQ_F32 ? 1 : 3) * sizeof(SelLevel);
}
extern "C" size_t mq_percentile_workspace_bytes(void) { return sizeof(SelWs); }
extern "C" size_t mq_percentile_batch_workspace_bytes(int32_t count) {
  return (size_t)(count < 1 ? 1 : count) * sizeof(SelWs);
}

// Byte offset / element count (uint32) of the histogram area that `level` accumulates into (coarse
// counters followed by the fine bins; level 0: 128 + 32 768, level >= 1: 2 x 256 + 2 x 65 536), for
// callers that merge histograms across devices (all-reduce SUM) between accumulate and scan.
extern "C" void mq_select_layout(int32_t level, size_t* offset, size_t* count) {
  if (level <= 0) {
    if (offset) *offset = offsetof(SelWs, top_coarse);
    if (count) *count = SEL_TOP_BINS / 256 + SEL_TOP_BINS;
  } else {
    if (offset) *offset = offsetof(SelWs, lvl) + (size_t)(level - 1) * sizeof(SelLevel);
    if (count) *count = sizeof(SelLevel) / sizeof(uint32_t);
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

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0; }

static int launch_accumulate(const SelBatch& b, int count, int32_t dtype, int level, hipStream_t st) {
  const int V = dtype == MQ_F32 ? 4 : 2;
  if (level == 0) {
    int64_t nb = (b.n / V + 1024 * 4 - 1) / (1024 * 4);
    int64_t cap = 256 / count < 4 ? 4 : 256 / count;  // one 128 KiB-LDS workgroup fits per CU
    dim3 grid((unsigned)(nb < 1 ? 1 : (nb > cap ? cap : nb)), (unsigned)count);
    if (dtype == MQ_F32) {
      MQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sel_hist_top<float>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, SEL_TOP_BINS * 4));
      hipLaunchKernelGGL((k_sel_hist_top<float>), grid, dim3(1024), SEL_TOP_BINS * 4, st, b);
    } else {
      MQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sel_hist_top<double>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, SEL_TOP_BINS * 4));
      hipLaunchKernelGGL((k_sel_hist_top<double>), grid, dim3(1024), SEL_TOP_BINS * 4, st, b);
    }
  } else {
    int64_t nb = (b.n / V + 256 * 4 - 1) / (256 * 4);
    int64_t cap = 1024 / count < 16 ? 16 : 1024 / count;
    dim3 grid((unsigned)(nb < 1 ? 1 : (nb > cap ? cap : nb)), (unsigned)count);
    if (dtype == MQ_F32) hipLaunchKernelGGL((k_sel_hist_digit<float>), grid, dim3(256), 0, st, b, level);
    else hipLaunchKernelGGL((k_sel_hist_digit<double>), grid, dim3(256), 0, st, b, level);
  }
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}

// Step 1: clear the histograms of one selection workspace (one fill).
extern "C" int mq_select_begin(void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!workspace || workspace_bytes < sizeof(SelWs)) return MQ_ERR_WORKSPACE;
  MQ_HIP(hipMemsetAsync(workspace, 0, sizeof(SelWs), static_cast<hipStream_t>(hip_stream)));
  return MQ_OK;
}

// Step 2 (level 0) / step 4 (level >= 1): add one array's contribution to that level's histogram.
// May be called for several arrays before the scan.
extern "C" int mq_select_accumulate(const void* x, int32_t dtype, int64_t n, int32_t level, void* workspace,
                                    size_t workspace_bytes, void* hip_stream) {
  if (n < 0 || (n > 0 && !x) || (dtype != MQ_F32 && dtype != MQ_F64) || level < 0 || level > 3) return MQ_ERR_ARG;
  if (!workspace || workspace_bytes < sizeof(SelWs)) return MQ_ERR_WORKSPACE;
  if (!al16(x) || !al16(workspace)) return MQ_ERR_ALIGN;
  if (n == 0) return MQ_OK;
  SelBatch b = {};
  b.x[0] = x;
  b.ws = static_cast<char*>(workspace);
  b.n = n;
  return launch_accumulate(b, 1, dtype, level, static_cast<hipStream_t>(hip_stream));
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
  SelBatch b = {};
  b.ws = static_cast<char*>(workspace);
  FinishArgs fin = {finish ? dtype : -1, gamma, static_cast<char*>(out), static_cast<char*>(out_pair)};
  hipLaunchKernelGGL(k_sel_scan, dim3(1), dim3(256), 0, static_cast<hipStream_t>(hip_stream), b, (int)level,
                     (uint64_t)k0, (uint64_t)k1, fin);
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}

// np.percentile(xs[i], pct) for `count` non-negative arrays of n values each (same dtype), one result per
// array in out[i] (and out_pair[2 i], out_pair[2 i + 1]): 5 launches per 32 arrays for float32 (fill,
// histogram, scan, matched-digit histogram, scan + interpolation), 9 for float64.  xs is a HOST array.
extern "C" int mq_percentile_batch(const void* const* xs, int32_t count, int32_t dtype, int64_t n, double pct,
                                   void* out, void* out_pair, void* workspace, size_t workspace_bytes,
                                   void* hip_stream) {
  if (count < 0 || (count > 0 && !xs) || n <= 0 || (!out && !out_pair)) return MQ_ERR_ARG;
  if (dtype != MQ_F32 && dtype != MQ_F64) return MQ_ERR_ARG;
  if (count == 0) return MQ_OK;
  if (!workspace || workspace_bytes < mq_percentile_batch_workspace_bytes(count > SEL_MAX_BATCH ? SEL_MAX_BATCH : count))
    return MQ_ERR_WORKSPACE;
  if (!al16(workspace)) return MQ_ERR_ALIGN;
  for (int i = 0; i < count; ++i) {
    if (!xs[i]) return MQ_ERR_ARG;
    if (!al16(xs[i])) return MQ_ERR_ALIGN;
  }
  int64_t k0, k1;
  double gamma;
  int rc = mq_percentile_index(dtype, n, pct, &k0, &k1, &gamma);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const size_t esz = dtype == MQ_F32 ? 4 : 8;
  const int levels = dtype == MQ_F32 ? 1 : 3;
  for (int done = 0; done < count; done += SEL_MAX_BATCH) {
    const int c = count - done < SEL_MAX_BATCH ? count - done : SEL_MAX_BATCH;
    SelBatch b = {};
    for (int i = 0; i < c; ++i) b.x[i] = xs[done + i];
    b.ws = static_cast<char*>(workspace);
    b.ws_stride = (int64_t)sizeof(SelWs);
    b.n = n;
    // one fill per slice would be c launches: the slices are contiguous, clear them together up to the
    // last level this dtype uses
    if (ws_used_bytes(dtype) == sizeof(SelWs) || c == 1) {
      MQ_HIP(hipMemsetAsync(workspace, 0, (size_t)(c - 1) * sizeof(SelWs) + ws_used_bytes(dtype), st));
    } else {
      MQ_HIP(hipMemset2DAsync(workspace, sizeof(SelWs), 0, ws_used_bytes(dtype), (size_t)c, st));
    }
    for (int level = 0; level <= levels; ++level) {
      if ((rc = launch_accumulate(b, c, dtype, level, st))) return rc;
      FinishArgs fin = {level == levels ? dtype : -1, gamma,
                        out ? static_cast<char*>(out) + (size_t)done * esz : nullptr,
                        out_pair ? static_cast<char*>(out_pair) + (size_t)done * 2 * esz : nullptr};
      hipLaunchKernelGGL(k_sel_scan, dim3((unsigned)c), dim3(256), 0, st, b, level, (uint64_t)k0, (uint64_t)k1, fin);
      MQ_HIP(hipGetLastError());
    }
  }
  return MQ_OK;
}

// np.percentile(x, pct) for one array: a batch of one.
extern "C" int mq_percentile(const void* x, int32_t dtype, int64_t n, double pct, void* out, void* out_pair,
                             void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!x) return MQ_ERR_ARG;
  const void* xs[1] = {x};
  return mq_percentile_batch(xs, 1, dtype, n, pct, out, out_pair, workspace, workspace_bytes, hip_stream);
}
