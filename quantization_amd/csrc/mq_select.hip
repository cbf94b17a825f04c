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
//
// Large float32 arrays (the |error| dumps) take a cheaper route first, the exact one only if it fails:
//   sample   1/32 of the array (one pseudo-randomly placed 256-byte line of every 8 KiB) goes into a
//            15-bit histogram; the sample ranks 8 standard deviations either side of the wanted
//            quantile give a bracket of at most 16 top-level bins;
//   bracket  ONE pass over the whole array counts the elements below the bracket (registers, no
//            atomics) and builds the full-resolution histogram of the elements inside it (a fraction
//            of a percent, aggregated in an LDS hash table);
//   scan     if both wanted ranks fall inside the bracket their values are exact and final;
//            otherwise the slice's `ok` flag stays 0 and the exact level-0/level-1 kernels, which are
//            always enqueued behind and return at once when `ok` is set, produce the result.
// The bracket is only ever a guess about where to look: the counts that decide are exact.
#include <mutex>

#include "mq_sel.h"

struct SelBatch {
  const void* x[SEL_MAX_BATCH];
  int64_t n[SEL_MAX_BATCH];  // values per array
  char* ws;          // SelWs of array 0
  int64_t ws_stride; // bytes between workspace slices (0: all arrays accumulate into one selection)
};
struct SelRanks {  // per array: the two wanted ranks and numpy's interpolation weight
  uint64_t k0[SEL_MAX_BATCH], k1[SEL_MAX_BATCH];
  double gamma[SEL_MAX_BATCH];
};
struct SelSampleRanks {  // per array: sample ranks that bracket the wanted quantile
  uint64_t lo[SEL_MAX_BATCH], hi[SEL_MAX_BATCH];
};
__device__ __forceinline__ SelWs* batch_ws(const SelBatch& b, int y) {
  return reinterpret_cast<SelWs*>(b.ws + (int64_t)y * b.ws_stride);
}

__device__ __forceinline__ uint64_t key_of(float v) { return (uint64_t)(f_bits(v) & 0x7fffffffu) << 32; }
__device__ __forceinline__ uint64_t key_of(double v) { return __builtin_bit_cast(uint64_t, v) & 0x7fffffffffffffffull; }

// ---- level 0 -------------------------------------------------------------------------------------
// Merges a workgroup's LDS histogram into the global one.  Same-address global atomics serialise in L2, and
// every workgroup of an array meets in the same few coarse counters: a wave covers 64 consecutive bins that
// share one coarse counter, so it adds their sum with ONE atomic.
__device__ __forceinline__ void flush_top(const uint32_t* lh, uint32_t* fine, uint32_t* coarse) {
  for (int i = threadIdx.x; i < SEL_TOP_BINS; i += blockDim.x) {
    const uint32_t c = lh[i];
    if (__builtin_amdgcn_ballot_w64(c != 0u) == 0ull) continue;  // (most waves' 64 bins are empty: data spans few binades)
    if (c) atomicAdd(&fine[i], c);
    uint32_t sum = c;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if ((threadIdx.x & 63) == 0 && sum) atomicAdd(&coarse[i >> 8], sum);
  }
}

template <typename T>
__global__ __launch_bounds__(1024) void k_sel_hist_top(const SelBatch b, int guard) {
  extern __shared__ uint32_t lh[];
  const T* x = static_cast<const T*>(b.x[blockIdx.y]);
  SelWs* ws = batch_ws(b, blockIdx.y);
  const int64_t n = b.n[blockIdx.y];
  if (guard && ws->ok) return;
  for (int i = threadIdx.x; i < SEL_TOP_BINS; i += blockDim.x) lh[i] = 0;
  __syncthreads();
  constexpr int V = 16 / sizeof(T);  // elements per 16-byte load
  const int64_t nvec = n / V;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
  constexpr int U = 4;  // loads in flight per lane
  for (int64_t v0 = t0; v0 < nvec; v0 += U * step) {
    u32x4 raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t v = v0 + u * step;
      raw[u] = u32x4{0, 0, 0, 0};
      if (v < nvec) raw[u] = reinterpret_cast<const u32x4*>(x)[v];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (v0 + u * step >= nvec) break;
      if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(&lh[(raw[u][i] >> 16) & 0x7fffu], 1u);
      } else {
        atomicAdd(&lh[(raw[u][1] >> 16) & 0x7fffu], 1u);
        atomicAdd(&lh[(raw[u][3] >> 16) & 0x7fffu], 1u);
      }
    }
  }
  for (int64_t i = nvec * V + t0; i < n; i += step) atomicAdd(&lh[(uint32_t)(key_of(x[i]) >> 48)], 1u);
  __syncthreads();
  flush_top(lh, ws->top, ws->top_coarse);
}

// ---- level 0 in 9 KiB of LDS: for launches that run beside a kernel which leaves no room for 128 KiB --------------------
// A window of SEL_WINDOW consecutive level-0 counters, placed so that the largest key among the workgroup's first
// vectors lies at least 256 counters under its top, plus the counters 0 .. 255 (exact zeros are common in error
// data, and one global counter taking every workgroup's zeros serialises them in one L2 channel).  Keys outside both
// go to the global counters directly: rare for the data this library selects from (squared-error means and absolute
// errors of one tensor span a few binades, 128 counters each), slow but exact for anything else.  256 threads.
#define SEL_WINDOW 2048
// clears lh[SEL_WINDOW + 256] and returns the window's first counter; mx: this thread's largest level-0 bin so far
__device__ __forceinline__ uint32_t win_setup(uint32_t* lh, uint32_t* wmax /*[4]*/, uint32_t mx) {
  for (int i = threadIdx.x; i < SEL_WINDOW + 256; i += blockDim.x) lh[i] = 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(mx, off); mx = mx > o ? mx : o; }
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = wmax[0];
#pragma unroll
  for (int w = 1; w < 4; ++w) mx = mx > wmax[w] ? mx : wmax[w];
  // a multiple of 256 (a wave's 64 consecutive counters then share one coarse counter in the flush)
  uint32_t top = ((mx + 256u) | 255u) + 1u;
  top = top > (uint32_t)SEL_TOP_BINS ? (uint32_t)SEL_TOP_BINS : top;
  return top > (uint32_t)SEL_WINDOW ? top - (uint32_t)SEL_WINDOW : 0u;
}
__device__ __forceinline__ void win_put(uint32_t* lh, uint32_t base, uint32_t* fine, uint32_t* coarse, uint32_t bin) {
  const uint32_t rel = bin - base;
  if (rel < (uint32_t)SEL_WINDOW) {
    atomicAdd(&lh[rel], 1u);
  } else if (bin < 256u) {
    atomicAdd(&lh[SEL_WINDOW + bin], 1u);
  } else {
    atomicAdd(&fine[bin], 1u);
    atomicAdd(&coarse[bin >> 8], 1u);
  }
}
__device__ __forceinline__ void win_flush(const uint32_t* lh, uint32_t base, uint32_t* fine, uint32_t* coarse) {
  __syncthreads();
  for (int i = threadIdx.x; i < SEL_WINDOW; i += blockDim.x) {
    const uint32_t c = lh[i];
    if (__builtin_amdgcn_ballot_w64(c != 0u) == 0ull) continue;
    if (c) atomicAdd(&fine[base + i], c);
    uint32_t sum = c;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if ((threadIdx.x & 63) == 0 && sum) atomicAdd(&coarse[(base + i) >> 8], sum);
  }
  if (base >= 256u) {  // (otherwise the window holds those counters itself)
    const uint32_t c = lh[SEL_WINDOW + threadIdx.x];
    if (c) atomicAdd(&fine[threadIdx.x], c);
    uint32_t sum = c;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if ((threadIdx.x & 63) == 0 && sum) atomicAdd(&coarse[0], sum);
  }
}
// level-0 bins of the keys in one 16-byte vector
template <typename T>
__device__ __forceinline__ void top_bins(const u32x4& raw, uint32_t* bins) {
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) bins[i] = (raw[i] >> 16) & 0x7fffu;
  } else {
    bins[0] = (raw[1] >> 16) & 0x7fffu;
    bins[1] = (raw[3] >> 16) & 0x7fffu;
  }
}
// k_sel_hist_top with the windowed histogram
template <typename T>
__global__ __launch_bounds__(256) void k_sel_hist_top_window(const SelBatch b, int guard) {
  __shared__ uint32_t lh[SEL_WINDOW + 256];
  __shared__ uint32_t wmax[4];
  const T* x = static_cast<const T*>(b.x[blockIdx.y]);
  SelWs* ws = batch_ws(b, blockIdx.y);
  const int64_t n = b.n[blockIdx.y];
  if (guard && ws->ok) return;
  constexpr int V = 16 / sizeof(T);
  const int64_t nvec = n / V;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
  uint32_t mx = 0;
  if (t0 < nvec) {
    uint32_t bins[V];
    top_bins<T>(reinterpret_cast<const u32x4*>(x)[t0], bins);
#pragma unroll
    for (int i = 0; i < V; ++i) mx = mx > bins[i] ? mx : bins[i];
  }
  const uint32_t base = win_setup(lh, wmax, mx);
  __syncthreads();
  constexpr int U = 4;  // loads in flight per lane
  for (int64_t v0 = t0; v0 < nvec; v0 += U * step) {
    u32x4 raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t v = v0 + u * step;
      raw[u] = u32x4{0, 0, 0, 0};
      if (v < nvec) raw[u] = reinterpret_cast<const u32x4*>(x)[v];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (v0 + u * step >= nvec) break;
      uint32_t bins[V];
      top_bins<T>(raw[u], bins);
#pragma unroll
      for (int i = 0; i < V; ++i) win_put(lh, base, ws->top, ws->top_coarse, bins[i]);
    }
  }
  for (int64_t i = nvec * V + t0; i < n; i += step) win_put(lh, base, ws->top, ws->top_coarse, (uint32_t)(key_of(x[i]) >> 48));
  win_flush(lh, base, ws->top, ws->top_coarse);
}

// ---- scan: which bin holds each remaining rank -----------------------------------------------------
// One 256-thread workgroup per array: locate the rank among the coarse counters, then among the 256
// fine bins under that counter.  With fin.dtype >= 0 the last level also performs numpy's interpolation
// (lib/_function_base_impl.py::_lerp) in that dtype:
//   r = a + (b - a) * t;  where t >= 0.5:  r = b - (b - a) * (1 - t)
struct FinishArgs {
  int32_t dtype;  // -1: no finish here; MQ_F32 / MQ_F64 / MQ_F32_AS_F64
  char* out;       // [batch] values of dtype
  char* out_pair;  // [batch][2]
};

// numpy's interpolation between the two order statistics (lib/_function_base_impl.py::_lerp) and the stores.
// p0 / p1: the located keys (float32 bits in the high word, or float64 bits).  MQ_F32_AS_F64: float32 keys, but
// the array they stand for is float64: widen first, interpolate and store in float64.
__device__ __forceinline__ void finish_store(const FinishArgs& fin, int idx, uint64_t p0, uint64_t p1, double gamma) {
  if (fin.dtype == MQ_F32) {
    float a = f_from_bits((uint32_t)(p0 >> 32)), c = f_from_bits((uint32_t)(p1 >> 32));
    float g = (float)gamma, diff = c - a;
    float v = a + diff * g;
    if (g >= 0.5f) v = c - diff * (1.0f - g);
    if (fin.out) reinterpret_cast<float*>(fin.out)[idx] = v;
    if (fin.out_pair) { reinterpret_cast<float*>(fin.out_pair)[2 * idx] = a; reinterpret_cast<float*>(fin.out_pair)[2 * idx + 1] = c; }
  } else if (fin.dtype == MQ_F64 || fin.dtype == MQ_F32_AS_F64) {
    const bool wide = fin.dtype == MQ_F32_AS_F64;
    double a = wide ? (double)f_from_bits((uint32_t)(p0 >> 32)) : __builtin_bit_cast(double, p0);
    double c = wide ? (double)f_from_bits((uint32_t)(p1 >> 32)) : __builtin_bit_cast(double, p1);
    double g = gamma, diff = c - a;
    double v = a + diff * g;
    if (g >= 0.5) v = c - diff * (1.0 - g);
    if (fin.out) reinterpret_cast<double*>(fin.out)[idx] = v;
    if (fin.out_pair) { reinterpret_cast<double*>(fin.out_pair)[2 * idx] = a; reinterpret_cast<double*>(fin.out_pair)[2 * idx + 1] = c; }
  }
}

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

__global__ __launch_bounds__(256) void k_sel_scan(const SelBatch b, int level, const SelRanks rk, FinishArgs fin,
                                                  int guard) {
  __shared__ uint64_t wsum[4];
  __shared__ uint32_t sh_bin;
  __shared__ uint64_t sh_rank;
  __shared__ SelState st;
  SelWs* ws = batch_ws(b, blockIdx.x);
  if (guard) {
    const uint32_t ok = ws->ok;
    if (ok == 1) return;
    if (ok == 2) {  // a refined selection (float32): the last level's scan walks the <= 16 patterns of each
      if (fin.dtype < 0) return;  // located entry, after the matched-digit kernel has counted them
      if (threadIdx.x == 0) {
        uint32_t bits[2];
        for (int r = 0; r < 2; ++r) {
          uint64_t rest = ws->ref_rank[r];
          uint32_t d = 0;
          const uint32_t width = 1u << ws->br_shift;
          while (d + 1 < width && rest >= ws->refine[r][d]) { rest -= ws->refine[r][d]; ++d; }
          bits[r] = ws->ref_base[r] + d;
        }
        finish_store(fin, blockIdx.x, (uint64_t)bits[0] << 32, (uint64_t)bits[1] << 32, rk.gamma[blockIdx.x]);
        ws->st.prefix[0] = (uint64_t)bits[0] << 32;
        ws->st.prefix[1] = (uint64_t)bits[1] << 32;
        ws->ok = 1;
      }
      return;
    }
  }
  if (threadIdx.x == 0) {
    if (level == 0) {  // the ranks arrive with the first scan
      SelState s0 = {};
      s0.rank[0] = rk.k0[blockIdx.x];
      s0.rank[1] = rk.k1[blockIdx.x];
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
    finish_store(fin, blockIdx.x, res.prefix[0], res.prefix[1], rk.gamma[blockIdx.x]);
  }
}

// ---- short arrays: the whole selection in one launch ------------------------------------------------------
// One 1024-thread workgroup per array of up to SEL_SMALL_N values (per-channel MSE lists: a few thousand values per
// tensor, for which the chain of clear / histogram / scan launches above is pure launch latency).  The keys are
// resolved 11 bits at a time -- 3 sweeps for float32, 6 for float64, the array stays in L2 -- with one LDS
// histogram per tracked rank, then interpolated as k_sel_scan does.
#define SEL_SMALL_N 32768
#define SEL_SMALL_BITS 11
// the keys of ranks r0 and r1 among x[0 .. n): every thread of the workgroup (THREADS threads, a power of two
// between 64 and 1024) calls this
template <typename T, int THREADS = 1024>
__device__ __forceinline__ void sel_small_core(const T* x, int64_t n, uint64_t r0, uint64_t r1, uint64_t* key0, uint64_t* key1) {
  constexpr int NB = 1 << SEL_SMALL_BITS, BPT = NB / THREADS;  // bins, bins per thread
  static_assert(BPT >= 2 && BPT * THREADS == NB, "every thread owns a whole run of bins");
  __shared__ uint32_t hist[2][NB];
  __shared__ uint32_t wtot[16];
  __shared__ uint64_t s_prefix[2], s_rank[2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();  // (a previous call's shared state has been read)
  if (threadIdx.x == 0) {
    s_prefix[0] = s_prefix[1] = 0;
    s_rank[0] = r0;
    s_rank[1] = r1;
  }
  int remaining = sizeof(T) == 4 ? 31 : 63;  // key bits still unresolved; float32 keys sit in the high word
  const int low = sizeof(T) == 4 ? 32 : 0;    // bits below the key
  while (remaining > 0) {
    const int width = remaining < SEL_SMALL_BITS ? remaining : SEL_SMALL_BITS;
    const int shift = low + remaining - width;
    for (int i = threadIdx.x; i < 2 * NB; i += THREADS) (&hist[0][0])[i] = 0;
    __syncthreads();
    const uint64_t p0 = s_prefix[0], p1 = s_prefix[1];
    const bool same = p0 == p1;
    const uint64_t himask = remaining == (sizeof(T) == 4 ? 31 : 63) ? 0ull : ~((1ull << (shift + width)) - 1ull);
    const uint32_t dmask = (1u << width) - 1u;
    auto tally = [&](uint64_t key) __attribute__((always_inline)) {
      const uint32_t d = (uint32_t)(key >> shift) & dmask;
      if ((key & himask) == p0) atomicAdd(&hist[0][d], 1u);
      if (!same && (key & himask) == p1) atomicAdd(&hist[1][d], 1u);
    };
    // 16-byte loads, four in flight per thread (a 256-thread workgroup walks 32768 values in 32 rounds of single
    // loads otherwise, each a trip to L2): elements before the first 16-byte boundary and after the last whole
    // vector one by one
    constexpr int V = 16 / sizeof(T), U = 4;
    int64_t head = (int64_t)(((16u - (uint32_t)(reinterpret_cast<uintptr_t>(x) & 15u)) & 15u) / sizeof(T));
    head = head < n ? head : n;
    for (int64_t i = threadIdx.x; i < head; i += THREADS) tally(key_of(x[i]));
    const u32x4* xv = reinterpret_cast<const u32x4*>(x + head);
    const int64_t nvec = (n - head) / V;
    for (int64_t v0 = threadIdx.x; v0 < nvec; v0 += U * THREADS) {
      u32x4 raw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        raw[u] = u32x4{0, 0, 0, 0};
        if (v0 + u * THREADS < nvec) raw[u] = xv[v0 + u * THREADS];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (v0 + u * THREADS >= nvec) break;
        if constexpr (sizeof(T) == 4) {
#pragma unroll
          for (int k = 0; k < 4; ++k) tally((uint64_t)(raw[u][k] & 0x7fffffffu) << 32);
        } else {
          tally((((uint64_t)raw[u][1] << 32) | raw[u][0]) & 0x7fffffffffffffffull);
          tally((((uint64_t)raw[u][3] << 32) | raw[u][2]) & 0x7fffffffffffffffull);
        }
      }
    }
    for (int64_t i = head + nvec * V + threadIdx.x; i < n; i += THREADS) tally(key_of(x[i]));
    __syncthreads();
    // locate each rank: every thread owns BPT consecutive bins
    for (int r = 0; r < 2; ++r) {
      const uint32_t* h = hist[(same || r == 0) ? 0 : 1];
      const uint64_t want = s_rank[r];
      uint32_t c[BPT], mine = 0;
#pragma unroll
      for (int j = 0; j < BPT; ++j) { c[j] = h[BPT * threadIdx.x + j]; mine += c[j]; }
      uint32_t inc = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
      }
      __syncthreads();  // (s_rank / wtot of the previous round have been read)
      if (lane == 63) wtot[wv] = inc;
      __syncthreads();
      uint64_t before = inc - mine;
      for (int w = 0; w < wv; ++w) before += wtot[w];
      if (want >= before && want < before + mine) {
        uint64_t at = before;
#pragma unroll
        for (int j = 0; j < BPT; ++j) {
          if (want >= at && want < at + c[j]) {
            s_prefix[r] = (r == 0 ? p0 : p1) | ((uint64_t)(BPT * threadIdx.x + j) << shift);
            s_rank[r] = want - at;
          }
          at += c[j];
        }
      }
      __syncthreads();
    }
    remaining -= width;
  }
  *key0 = s_prefix[0];
  *key1 = s_prefix[1];
}
template <typename T, int THREADS>
__global__ __launch_bounds__(THREADS) void k_sel_small(const SelBatch b, const SelRanks rk, FinishArgs fin) {
  uint64_t key0, key1;
  sel_small_core<T, THREADS>(static_cast<const T*>(b.x[blockIdx.x]), b.n[blockIdx.x], rk.k0[blockIdx.x], rk.k1[blockIdx.x],
                             &key0, &key1);
  if (threadIdx.x == 0) finish_store(fin, blockIdx.x, key0, key1, rk.gamma[blockIdx.x]);
}

// ---- percentile of a list selected inside the statistics pass (see mq_sel.h) ----------------------------------------
// (The list kernels run beside the statistics pass of another group size: 256-thread workgroups find room on a busy
//  chip where 1024-thread ones wait for a whole compute unit to drain.)
#define LIST_THREADS 256
__global__ __launch_bounds__(LIST_THREADS) void k_list_bracket(const ListJob* jobs) {
  const ListJob j = jobs[blockIdx.x];
  uint64_t key0, key1;
  sel_small_core<float, LIST_THREADS>(j.samples, (int64_t)j.m, j.s_lo, j.s_hi, &key0, &key1);
  if (threadIdx.x == 0) {
    j.brk[0] = (uint32_t)(key0 >> 32);
    j.brk[1] = (uint32_t)(key1 >> 32);
  }
}
__global__ __launch_bounds__(LIST_THREADS) void k_list_cand(const ListJob* jobs, ListFeedState* state, double* out) {
  const ListJob j = jobs[blockIdx.x];
  __shared__ uint32_t wsum[16][3];
  __shared__ uint32_t s_below, s_total, s_over, s_carry;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // the workgroups' counts
  uint32_t below = 0, total = 0, over = 0;
  for (uint32_t w = threadIdx.x; w < j.nblk; w += LIST_THREADS) {
    const uint32_t c = j.wg_cnt[2 * w + 1];
    below += j.wg_cnt[2 * w];
    total += c < (uint32_t)LIST_SLOT ? c : (uint32_t)LIST_SLOT;
    over |= c > (uint32_t)LIST_SLOT ? 1u : 0u;
  }
  for (int off = 32; off > 0; off >>= 1) {
    below += __shfl_xor(below, off); total += __shfl_xor(total, off); over |= __shfl_xor(over, off);
  }
  if (lane == 0) { wsum[wv][0] = below; wsum[wv][1] = total; wsum[wv][2] = over; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t a = 0, c = 0, o = 0;
    for (int i = 0; i < LIST_THREADS / 64; ++i) { a += wsum[i][0]; c += wsum[i][1]; o |= wsum[i][2]; }
    s_below = a; s_total = c; s_over = o | state[j.slot].ovf;
    s_carry = 0;
  }
  __syncthreads();
  below = s_below;
  total = s_total;
  // both ranks inside the kept values, and every kept value at hand?
  const bool ok = s_over == 0 && total <= j.ccap && j.k0 >= below && j.k1 < (uint64_t)below + total;
  if (!ok) {
    if (threadIdx.x == 0) state[j.slot].miss = 1;
    return;
  }
  // the slots packed into one list: a scan over the workgroups' counts, LIST_THREADS at a time
  uint32_t* cand = reinterpret_cast<uint32_t*>(j.cand);
  for (uint32_t w0 = 0; w0 < j.nblk; w0 += LIST_THREADS) {
    const uint32_t w = w0 + threadIdx.x;
    const uint32_t c = w < j.nblk ? j.wg_cnt[2 * w + 1] : 0u;
    uint32_t inc = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_up(inc, off);
      if (lane >= off) inc += o;
    }
    __syncthreads();  // (the previous round's wsum / s_carry have been read)
    if (lane == 63) wsum[wv][0] = inc;
    __syncthreads();
    uint32_t before = s_carry + inc - c;
    for (int i = 0; i < wv; ++i) before += wsum[i][0];
    for (uint32_t i = 0; i < c; ++i) cand[before + i] = j.slots[(size_t)w * LIST_SLOT + i];
    __syncthreads();
    if (threadIdx.x == LIST_THREADS - 1) s_carry = before + c;
  }
  __syncthreads();
  uint64_t key0, key1;
  sel_small_core<float, LIST_THREADS>(j.cand, (int64_t)total, j.k0 - below, j.k1 - below, &key0, &key1);
  if (threadIdx.x == 0) {
    const FinishArgs fin = {MQ_F32_AS_F64, reinterpret_cast<char*>(out), nullptr};
    finish_store(fin, (int)j.out, key0, key1, j.gamma);
  }
}
__global__ __launch_bounds__(LIST_THREADS) void k_list_repair(const ListJob* jobs, const ListFeedState* state, double* out) {
  const ListJob j = jobs[blockIdx.x];
  if (state[j.slot].miss == 0) return;
  uint64_t key0, key1;
  sel_small_core<float, LIST_THREADS>(j.list, (int64_t)j.n, j.k0, j.k1, &key0, &key1);
  if (threadIdx.x == 0) {
    const FinishArgs fin = {MQ_F32_AS_F64, reinterpret_cast<char*>(out), nullptr};
    finish_store(fin, (int)j.out, key0, key1, j.gamma);
  }
}

// ---- levels 1..3 -----------------------------------------------------------------------------------
// Elements whose bits above the digit equal a tracked prefix are counted per digit value; a workgroup
// first aggregates them in an LDS hash table (linear probing), then merges the table into the global
// histogram.  A full table (more distinct values than slots) falls back to global atomics.
template <typename T>
__global__ __launch_bounds__(256) void k_sel_hist_digit(const SelBatch b, int level, int guard) {
  __shared__ uint32_t hkey[2][SEL_HASH];  // digit + 1, 0 = empty
  __shared__ uint32_t hcnt[2][SEL_HASH];
  const T* x = static_cast<const T*>(b.x[blockIdx.y]);
  SelWs* ws = batch_ws(b, blockIdx.y);
  const int64_t n = b.n[blockIdx.y];
  if (guard) {
    const uint32_t ok = ws->ok;
    if (ok == 1) return;
    if (ok == 2) {  // refining pass: count the bit patterns inside the two located histogram entries
      if constexpr (sizeof(T) == 4) {
        const uint32_t base0 = ws->ref_base[0], base1 = ws->ref_base[1], width = 1u << ws->br_shift;
        const uint32_t* xs = static_cast<const uint32_t*>(b.x[blockIdx.y]);
        const int64_t nvec = n / 4;
        const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
        auto put = [&](uint32_t bits) {
          bits &= 0x7fffffffu;
          if (bits - base0 < width) atomicAdd(&ws->refine[0][bits - base0], 1u);
          if (bits - base1 < width) atomicAdd(&ws->refine[1][bits - base1], 1u);
        };
        for (int64_t v0 = t0; v0 < nvec; v0 += 4 * step) {  // 4 loads in flight per lane
          u32x4 raw[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int64_t v = v0 + u * step;
            raw[u] = v < nvec ? reinterpret_cast<const u32x4*>(xs)[v] : u32x4{~0u, ~0u, ~0u, ~0u};
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) put(raw[u][i]);
        }
        for (int64_t i = nvec * 4 + t0; i < n; i += step) put(xs[i]);
      }
      return;
    }
  }
  SelLevel* lv = &ws->lvl[level - 1];
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
      uint32_t old = hkey[s][h];  // plain read first: repeated values skip the compare-and-swap
      if (old == 0u) old = atomicCAS(&hkey[s][h], 0u, d + 1u);
      if (old == 0u || old == d + 1u) { atomicAdd(&hcnt[s][h], 1u); return; }
      h = (h + 1) & (SEL_HASH - 1);
    }
    atomicAdd(&lv->fine[s][d], 1u);
    atomicAdd(&lv->coarse[s][d >> 8], 1u);
  };
  constexpr int V = 16 / sizeof(T);
  const int64_t nvec = n / V;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
  constexpr int U = 4;  // loads in flight per lane
  for (int64_t v0 = t0; v0 < nvec; v0 += U * step) {
    u32x4 raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t v = v0 + u * step;
      raw[u] = u32x4{0, 0, 0, 0};
      if (v < nvec) raw[u] = reinterpret_cast<const u32x4*>(x)[v];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (v0 + u * step >= nvec) break;
      if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) put((uint64_t)(raw[u][i] & 0x7fffffffu) << 32);
      } else {
        put((((uint64_t)raw[u][1] << 32) | raw[u][0]) & 0x7fffffffffffffffull);
        put((((uint64_t)raw[u][3] << 32) | raw[u][2]) & 0x7fffffffffffffffull);
      }
    }
  }
  for (int64_t i = nvec * V + t0; i < n; i += step) put(key_of(x[i]));
  __syncthreads();
  // coarse counters meet in LDS first: one global atomic per counter and workgroup
  __shared__ uint32_t lcoarse[2][SEL_DIGIT_BINS / 256];
  for (int i = threadIdx.x; i < 2 * (SEL_DIGIT_BINS / 256); i += blockDim.x) (&lcoarse[0][0])[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * SEL_HASH; i += blockDim.x) {
    const int s = i / SEL_HASH, h = i % SEL_HASH;
    if (hkey[s][h]) {
      const uint32_t d = hkey[s][h] - 1u;
      atomicAdd(&lv->fine[s][d], hcnt[s][h]);
      atomicAdd(&lcoarse[s][d >> 8], hcnt[s][h]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * (SEL_DIGIT_BINS / 256); i += blockDim.x) {
    const uint32_t c = (&lcoarse[0][0])[i];
    if (c) atomicAdd(&(&lv->coarse[0][0])[i], c);
  }
}

// ---- sample / bracket route (float32) ----------------------------------------------------------------
// Clears two byte ranges of every workspace slice (16-byte stores).
__global__ __launch_bounds__(256) void k_sel_clear(const SelBatch b, uint32_t a_end, uint32_t b_begin, uint32_t b_end) {
  u32x4* base = reinterpret_cast<u32x4*>(batch_ws(b, blockIdx.y));
  const u32x4 z = {0, 0, 0, 0};
  const uint32_t na = a_end / 16, nb0 = b_begin / 16, nb1 = b_end / 16;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < na; i += gridDim.x * blockDim.x) base[i] = z;
  for (uint32_t i = nb0 + blockIdx.x * blockDim.x + threadIdx.x; i < nb1; i += gridDim.x * blockDim.x) base[i] = z;
}

__global__ __launch_bounds__(1024) void k_sel_sample(const SelBatch b) {
  extern __shared__ uint32_t lh[];
  const u32x4* x = static_cast<const u32x4*>(b.x[blockIdx.y]);
  SelWs* ws = batch_ws(b, blockIdx.y);
  for (int i = threadIdx.x; i < SEL_TOP_BINS; i += blockDim.x) lh[i] = 0;
  __syncthreads();
  const int64_t nchunks = b.n[blockIdx.y] / (SEL_CHUNK_LINES * 64);  // 64 float32 per line
  const int64_t total = nchunks * 16;                    // 16 lanes x 16 bytes read one line
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
  constexpr int U = 8;  // scattered lines: keep 8 loads in flight per lane
  for (int64_t i0 = t0; i0 < total; i0 += U * step) {
    u32x4 raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * step;
      const int64_t c = i >> 4;
      raw[u] = u32x4{0, 0, 0, 0};
      if (i < total) raw[u] = x[c * (SEL_CHUNK_LINES * 16) + sample_line((uint32_t)c) * 16 + ((uint32_t)i & 15u)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i0 + u * step < total) {
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(&lh[(raw[u][k] >> 16) & 0x7fffu], 1u);
      }
    }
  }
  __syncthreads();
  flush_top(lh, ws->samp, ws->samp_coarse);
}

// k_sel_sample with the windowed histogram (256 threads)
__global__ __launch_bounds__(256) void k_sel_sample_window(const SelBatch b) {
  __shared__ uint32_t lh[SEL_WINDOW + 256];
  __shared__ uint32_t wmax[4];
  const u32x4* x = static_cast<const u32x4*>(b.x[blockIdx.y]);
  SelWs* ws = batch_ws(b, blockIdx.y);
  const int64_t nchunks = b.n[blockIdx.y] / (SEL_CHUNK_LINES * 64);  // 64 float32 per line
  const int64_t total = nchunks * 16;                    // 16 lanes x 16 bytes read one line
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
  auto load = [&](int64_t i) {
    const int64_t c = i >> 4;
    return x[c * (SEL_CHUNK_LINES * 16) + sample_line((uint32_t)c) * 16 + ((uint32_t)i & 15u)];
  };
  uint32_t mx = 0;
  if (t0 < total) {
    uint32_t bins[4];
    top_bins<float>(load(t0), bins);
#pragma unroll
    for (int i = 0; i < 4; ++i) mx = mx > bins[i] ? mx : bins[i];
  }
  const uint32_t base = win_setup(lh, wmax, mx);
  __syncthreads();
  constexpr int U = 8;  // scattered lines: keep 8 loads in flight per lane
  for (int64_t i0 = t0; i0 < total; i0 += U * step) {
    u32x4 raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * step;
      raw[u] = u32x4{0, 0, 0, 0};
      if (i < total) raw[u] = load(i);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i0 + u * step < total) {
        uint32_t bins[4];
        top_bins<float>(raw[u], bins);
#pragma unroll
        for (int k = 0; k < 4; ++k) win_put(lh, base, ws->samp, ws->samp_coarse, bins[k]);
      }
    }
  }
  win_flush(lh, base, ws->samp, ws->samp_coarse);
}

// Level-0 histogram of the samples a feeding kernel stored (top 16 bits of each value, m per array), 256 threads and
// 9 KiB of LDS per workgroup: a window of 2048 consecutive counters, placed so that the largest of the workgroup's
// first samples sits at least 256 counters under its top (the samples of an error array thin out by half per 128
// counters below the largest ones), plus the counters 0 .. 255 (an error that is exactly zero is common, and one
// global counter taking every workgroup's zeros serialises them in one L2 channel: 0.3 ms).  A sample outside both
// goes to the global counters directly: rare, and spread over many addresses.  (A workgroup-wide histogram of all
// 32768 counters -- 128 KiB, one workgroup per CU, every one of them clearing and flushing the lot -- took 55 us for
// 48 arrays of 512 Ki samples.)
#define SEL_SAMPLE_WINDOW 2048
__global__ __launch_bounds__(256) void k_sel_sample_u16_window(const SelBatch b, const uint16_t* samples, int64_t stride,
                                                                int64_t m) {
  __shared__ uint32_t lh[SEL_SAMPLE_WINDOW + 256];  // the window, then the counters 0 .. 255
  __shared__ uint32_t wmax[4];
  const uint16_t* x = samples + (int64_t)blockIdx.y * stride;
  SelWs* ws = batch_ws(b, blockIdx.y);
  uint32_t* fine = ws->samp;
  uint32_t* coarse = ws->samp_coarse;
  const int64_t nvec = m / 8;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
  for (int i = threadIdx.x; i < SEL_SAMPLE_WINDOW + 256; i += blockDim.x) lh[i] = 0;
  // the window, from the workgroup's first vectors
  uint32_t mx = 0;
  if (t0 < nvec) {
    const u32x4 raw = reinterpret_cast<const u32x4*>(x)[t0];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t lo = raw[k] & 0x7fffu, hi = (raw[k] >> 16) & 0x7fffu;
      mx = mx > lo ? mx : lo;
      mx = mx > hi ? mx : hi;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(mx, off); mx = mx > o ? mx : o; }
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = wmax[0];
#pragma unroll
  for (int w = 1; w < 4; ++w) mx = mx > wmax[w] ? mx : wmax[w];
  // base: a multiple of 256 (a wave's 64 consecutive counters then share one coarse counter in the flush)
  uint32_t top = ((mx + 256u) | 255u) + 1u;
  top = top > (uint32_t)SEL_TOP_BINS ? (uint32_t)SEL_TOP_BINS : top;
  const uint32_t base = top > (uint32_t)SEL_SAMPLE_WINDOW ? top - (uint32_t)SEL_SAMPLE_WINDOW : 0u;
  auto put = [&](uint32_t bin) {
    const uint32_t rel = bin - base;
    if (rel < (uint32_t)SEL_SAMPLE_WINDOW) {
      atomicAdd(&lh[rel], 1u);
    } else if (bin < 256u) {
      atomicAdd(&lh[SEL_SAMPLE_WINDOW + bin], 1u);
    } else {
      atomicAdd(&fine[bin], 1u);
      atomicAdd(&coarse[bin >> 8], 1u);
    }
  };
  constexpr int U = 4;  // loads in flight per thread
  for (int64_t v0 = t0; v0 < nvec; v0 += U * step) {
    u32x4 raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t v = v0 + u * step;
      raw[u] = v < nvec ? reinterpret_cast<const u32x4*>(x)[v] : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (v0 + u * step >= nvec) break;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        put(raw[u][k] & 0x7fffu);
        put((raw[u][k] >> 16) & 0x7fffu);
      }
    }
  }
  for (int64_t i = nvec * 8 + t0; i < m; i += step) put(x[i] & 0x7fffu);
  __syncthreads();
  for (int i = threadIdx.x; i < SEL_SAMPLE_WINDOW; i += blockDim.x) {
    const uint32_t c = lh[i];
    if (__builtin_amdgcn_ballot_w64(c != 0u) == 0ull) continue;
    if (c) atomicAdd(&fine[base + i], c);
    uint32_t sum = c;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if ((threadIdx.x & 63) == 0 && sum) atomicAdd(&coarse[(base + i) >> 8], sum);
  }
  if (base >= 256u) {  // (otherwise the window holds those counters itself)
    const uint32_t c = lh[SEL_SAMPLE_WINDOW + threadIdx.x];
    if (c) atomicAdd(&fine[threadIdx.x], c);
    uint32_t sum = c;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if ((threadIdx.x & 63) == 0 && sum) atomicAdd(&coarse[0], sum);
  }
}

// Bins of the two sample ranks -> bracket.
__global__ __launch_bounds__(256) void k_sel_scan_sample(const SelBatch b, const SelSampleRanks sr) {
  const uint64_t r_lo = sr.lo[blockIdx.x], r_hi = sr.hi[blockIdx.x];
  __shared__ uint64_t wsum[4];
  __shared__ uint32_t sh_bin;
  __shared__ uint64_t sh_rank;
  SelWs* ws = batch_ws(b, blockIdx.x);
  if (threadIdx.x == 0) { sh_bin = 0; sh_rank = 0; }
  __syncthreads();
  locate(ws->samp_coarse, ws->samp, SEL_TOP_BINS / 256, r_lo, wsum, &sh_bin, &sh_rank);
  const uint32_t lo = sh_bin;
  __syncthreads();
  locate(ws->samp_coarse, ws->samp, SEL_TOP_BINS / 256, r_hi, wsum, &sh_bin, &sh_rank);
  const uint32_t hi = sh_bin;
  if (threadIdx.x == 0) {
    // up to 16 bins: one histogram entry per bit pattern; up to 256 bins: 2^shift patterns per entry, and a
    // refining pass over the stored array finishes the job
    const bool usable = hi >= lo && hi - lo < SEL_BR_MAX_BINS;
    const uint32_t nb = usable ? hi - lo + 1 : 0;
    uint32_t shift = 0;
    while (((nb << 16) >> shift) > (uint32_t)SEL_BR_BINS * 65536u) ++shift;
    ws->br_lo = lo;
    ws->br_nb = nb;
    ws->br_shift = shift;
  }
}

// Clears the part of the bracket histogram that the bracket found by the sample scan will use.
__global__ __launch_bounds__(256) void k_sel_clear_bracket(const SelBatch b) {
  SelWs* ws = batch_ws(b, blockIdx.y);
  const uint32_t entries = (ws->br_nb << 16) >> ws->br_shift;
  const uint32_t nbins = (entries + 65535u) >> 16;  // whole chunks of 65536 entries, as the scan reads them
  const u32x4 z = {0, 0, 0, 0};
  u32x4* coarse = reinterpret_cast<u32x4*>(ws->br_coarse);
  u32x4* fine = reinterpret_cast<u32x4*>(ws->br_fine);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nbins * 64; i += gridDim.x * blockDim.x) coarse[i] = z;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nbins * 16384; i += gridDim.x * blockDim.x) fine[i] = z;
}

__global__ __launch_bounds__(256) void k_sel_bracket(const SelBatch b) {
  __shared__ uint32_t hkey[SEL_BR_HASH];  // key + 1, 0 = empty
  __shared__ uint32_t hcnt[SEL_BR_HASH];
  __shared__ uint32_t wcnt[4];
  const u32x4* x = static_cast<const u32x4*>(b.x[blockIdx.y]);
  SelWs* ws = batch_ws(b, blockIdx.y);
  const uint32_t lo = ws->br_lo, nbins = ws->br_nb, shift = ws->br_shift;
  if (nbins == 0) return;
  for (int i = threadIdx.x; i < SEL_BR_HASH; i += blockDim.x) { hkey[i] = 0; hcnt[i] = 0; }
  __syncthreads();
  uint32_t below = 0;  // per lane: at most n / (grid * 256) + 4 < 2^32
  auto put = [&](uint32_t bits) {
    bits &= 0x7fffffffu;
    const uint32_t t = (bits >> 16) - lo;  // wraps for bins under the bracket
    below += (bits >> 16) < lo ? 1u : 0u;
    if (t < nbins) {
      const uint32_t key = ((t << 16) | (bits & 0xffffu)) >> shift;
      uint32_t h = (key * 2654435761u) >> 20;  // 12 bits
      for (int probe = 0; probe < 64; ++probe) {
        uint32_t old = hkey[h];  // plain read first: repeated values skip the compare-and-swap
        if (old == 0u) old = atomicCAS(&hkey[h], 0u, key + 1u);
        if (old == 0u || old == key + 1u) { atomicAdd(&hcnt[h], 1u); return; }
        h = (h + 1) & (SEL_BR_HASH - 1);
      }
      atomicAdd(&ws->br_fine[key], 1u);
      atomicAdd(&ws->br_coarse[key >> 8], 1u);
    }
  };
  const int64_t n = b.n[blockIdx.y], nvec = n / 4;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t v0 = t0; v0 < nvec; v0 += 4 * step) {  // four loads in flight per lane
    u32x4 raw[4];
    bool have[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t v = v0 + u * step;
      have[u] = v < nvec;
      raw[u] = have[u] ? x[v] : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!have[u]) break;
#pragma unroll
      for (int i = 0; i < 4; ++i) put(raw[u][i]);
    }
  }
  const uint32_t* xs = static_cast<const uint32_t*>(b.x[blockIdx.y]);
  for (int64_t i = nvec * 4 + t0; i < n; i += step) put(xs[i]);
  // block total of `below`
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) below += __shfl_xor(below, off);
  if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = below;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&ws->below, (unsigned long long)wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3]);
  // coarse counters meet in LDS first: one global atomic per counter and workgroup
  __shared__ uint32_t lcoarse[SEL_BR_BINS * 256];
  for (int i = threadIdx.x; i < SEL_BR_BINS * 256; i += blockDim.x) lcoarse[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < SEL_BR_HASH; i += blockDim.x) {
    if (hkey[i]) {
      const uint32_t key = hkey[i] - 1u;
      atomicAdd(&ws->br_fine[key], hcnt[i]);
      atomicAdd(&lcoarse[key >> 8], hcnt[i]);
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < ((nbins << 16) >> shift) >> 8; i += blockDim.x) {
    const uint32_t c = lcoarse[i];
    if (c) atomicAdd(&ws->br_coarse[i], c);
  }
}

// Both ranks inside the bracket -> exact values, interpolation, ok = 1; with a coarse bracket histogram
// (br_shift > 0) -> the entries holding them, ok = 2 (the refining pass finishes).
__global__ __launch_bounds__(256) void k_sel_scan_bracket(const SelBatch b, const SelRanks rk, FinishArgs fin,
                                                          int cand_mode) {
  const uint64_t k0 = rk.k0[blockIdx.x], k1 = rk.k1[blockIdx.x];
  __shared__ uint64_t wsum[4];
  __shared__ uint32_t sh_bin;
  __shared__ uint64_t sh_rank;
  __shared__ uint64_t tot[SEL_BR_BINS + 1];
  SelWs* ws = batch_ws(b, blockIdx.x);
  if (ws->br_nb == 0) return;
  const uint32_t shift = ws->br_shift;
  if (cand_mode && shift) return;  // a feeding kernel kept this bracket's values as a list: k_sel_cand
  const uint32_t entries = (ws->br_nb << 16) >> shift;
  const uint32_t nchunks = (entries + 65535u) >> 16;  // chunks of 65536 histogram entries
  if (threadIdx.x == 0) { sh_bin = 0; sh_rank = 0; tot[0] = 0; }
  for (uint32_t j = 0; j < nchunks; ++j) {  // elements per chunk
    uint64_t c = ws->br_coarse[j * 256 + threadIdx.x];
    uint64_t before = block_excl_scan(c, wsum);
    if (threadIdx.x == 255) tot[j + 1] = before + c;
    __syncthreads();
  }
  if (threadIdx.x == 0)
    for (uint32_t j = 0; j < nchunks; ++j) tot[j + 1] += tot[j];
  __syncthreads();
  const uint64_t below = ws->below, inside = tot[nchunks];
  if (k0 < below || k1 >= below + inside) return;  // not resolved here: the exact route runs
  uint32_t bits[2];
  uint64_t rest[2];
  for (int r = 0; r < 2; ++r) {
    const uint64_t rk = (r ? k1 : k0) - below;
    uint32_t j = 0;
    while (j + 1 < nchunks && rk >= tot[j + 1]) ++j;
    locate(ws->br_coarse + j * 256, ws->br_fine + j * 65536, 256, rk - tot[j], wsum, &sh_bin, &sh_rank);
    bits[r] = (ws->br_lo << 16) + (((j << 16) | sh_bin) << shift);  // first bit pattern of the entry
    rest[r] = sh_rank;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (shift) {
      ws->ref_base[0] = bits[0];
      ws->ref_base[1] = bits[1];
      ws->ref_rank[0] = rest[0];
      ws->ref_rank[1] = rest[1];
      ws->ok = 2;
      return;
    }
    finish_store(fin, blockIdx.x, (uint64_t)bits[0] << 32, (uint64_t)bits[1] << 32, rk.gamma[blockIdx.x]);
    ws->st.prefix[0] = (uint64_t)bits[0] << 32;
    ws->st.prefix[1] = (uint64_t)bits[1] << 32;
    ws->ok = 1;
  }
}

// ---- wide brackets of a fed selection: select among the kept values -----------------------------------------
// One 1024-thread workgroup per array whose feeding kernel left the bit patterns of the values inside the bracket
// as a list (SelWs::cand_n entries in br_fine, see sel_feed_cand_add): the wanted ranks, less the count below
// the bracket, are resolved among the list in two sweeps of 12 bits over the patterns' offsets from the bracket's
// first pattern (the list is a few hundred KiB and stays in L2).  Slices with a narrow bracket, a missed bracket or an overflowed list are left alone (the
// exact route over the stored array, enqueued behind, handles whatever has ok == 0).
// (256 threads: beside a full quantize pass of another configuration, which keeps every CU's wave slots taken, a
//  1024-thread workgroup waits for sixteen free slots on one CU until that pass drains -- measured 370 us against 5)
#define SEL_CAND_THREADS 256
__global__ __launch_bounds__(SEL_CAND_THREADS) void k_sel_cand(const SelBatch b, const SelRanks rk, FinishArgs fin) {
  constexpr int BITS = 12, NB = 1 << BITS;  // keys relative to the bracket's first pattern: < 2^24, two sweeps
  constexpr int BPT = NB / SEL_CAND_THREADS;  // consecutive bins a thread owns in the scan
  __shared__ uint32_t hist[2][NB];
  __shared__ uint32_t wtot[16];
  __shared__ uint32_t s_prefix[2];
  __shared__ uint64_t s_rank[2];
  SelWs* ws = batch_ws(b, blockIdx.x);
  if (ws->br_nb == 0 || ws->br_shift == 0 || ws->ok) return;
  const uint64_t n = ws->cand_n, below = ws->below;
  const uint64_t k0 = rk.k0[blockIdx.x], k1 = rk.k1[blockIdx.x];
  if (n > (uint64_t)SEL_CAND_MAX || k0 < below || k1 >= below + n) return;
  const uint32_t* x = ws->br_fine;
  const uint32_t lo_bits = ws->br_lo << 16;  // (br_nb <= SEL_BR_MAX_BINS = 256 bins of 2^16 patterns)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) {
    s_prefix[0] = s_prefix[1] = 0;
    s_rank[0] = k0 - below;
    s_rank[1] = k1 - below;
  }
  for (int sweep = 0; sweep < 2; ++sweep) {
    const int shift = sweep == 0 ? BITS : 0;
    for (int i = threadIdx.x; i < 2 * NB; i += blockDim.x) (&hist[0][0])[i] = 0;
    __syncthreads();
    const uint32_t p0 = s_prefix[0], p1 = s_prefix[1];
    const bool same = p0 == p1;
    const uint32_t himask = sweep == 0 ? 0u : ~((uint32_t)NB - 1u);
    auto put = [&](uint32_t bits) {
      const uint32_t key = bits - lo_bits;
      const uint32_t d = (key >> shift) & (NB - 1);
      if ((key & himask) == p0) atomicAdd(&hist[0][d], 1u);
      if (!same && (key & himask) == p1) atomicAdd(&hist[1][d], 1u);
    };
    // 16-byte loads, eight in flight per thread (one workgroup walks the list: the loop is bound by load latency)
    const uint64_t nvec = n / 4;
    constexpr int U = 8;  // loads in flight per thread
    for (uint64_t v0 = threadIdx.x; v0 < nvec; v0 += U * blockDim.x) {
      u32x4 raw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t v = v0 + (uint64_t)u * blockDim.x;
        raw[u] = v < nvec ? reinterpret_cast<const u32x4*>(x)[v] : u32x4{0, 0, 0, 0};
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (v0 + (uint64_t)u * blockDim.x >= nvec) break;
#pragma unroll
        for (int j = 0; j < 4; ++j) put(raw[u][j]);
      }
    }
    for (uint64_t i = nvec * 4 + threadIdx.x; i < n; i += blockDim.x) put(x[i]);
    __syncthreads();
    for (int r = 0; r < 2; ++r) {  // every thread owns BPT consecutive bins
      const uint32_t* h = hist[(same || r == 0) ? 0 : 1];
      const uint64_t want = s_rank[r];
      uint32_t c[BPT];
      uint32_t mine = 0;
#pragma unroll
      for (int j = 0; j < BPT; ++j) { c[j] = h[BPT * threadIdx.x + j]; mine += c[j]; }
      uint32_t inc = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
      }
      __syncthreads();
      if (lane == 63) wtot[wv] = inc;
      __syncthreads();
      uint64_t before = inc - mine;
      for (int w = 0; w < wv; ++w) before += wtot[w];
      if (want >= before && want < before + mine) {
        int j = 0;
        while (want >= before + c[j]) { before += c[j]; ++j; }
        s_prefix[r] = (r == 0 ? p0 : p1) | ((uint32_t)(BPT * threadIdx.x + j) << shift);
        s_rank[r] = want - before;
      }
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) {
    const uint64_t a = (uint64_t)(s_prefix[0] + lo_bits) << 32, c = (uint64_t)(s_prefix[1] + lo_bits) << 32;
    finish_store(fin, blockIdx.x, a, c, rk.gamma[blockIdx.x]);
    ws->st.prefix[0] = a;
    ws->st.prefix[1] = c;
    ws->ok = 1;
  }
}

// ================================================================================================
// MQ_F32_AS_F64 selects among float32 keys; only the ranks, the interpolation and the results are float64.
static bool f32_keys(int32_t dtype) { return dtype == MQ_F32 || dtype == MQ_F32_AS_F64; }
static bool sel_dtype_ok(int32_t dtype) { return dtype == MQ_F32 || dtype == MQ_F64 || dtype == MQ_F32_AS_F64; }
static size_t ws_used_bytes(int32_t dtype) {  // what one selection of this dtype touches
  return offsetof(SelWs, lvl) + (size_t)(f32_keys(dtype) ? 1 : 3) * sizeof(SelLevel);
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
  } else if (dtype == MQ_F64 || dtype == MQ_F32_AS_F64) {
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

#define SEL_FEED_SIGMAS 16.0
static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0; }

// Sample ranks `sigmas` binomial standard deviations either side of the wanted quantile among the m sampled
// elements of an array of n; false when the array is too small or the quantile too near an end for the bracket
// route.  The samples are clusters of consecutive elements, not independent draws, so the sample quantile
// scatters more than the binomial figure: by a factor 2 .. 2.6 for the 64-element lines of a stored |error|
// array of normal weights (8 sigmas are used there), by 3.5 .. 5 (11 at the worst of 20 trials) for the
// 512-element runs a feeding quantize kernel samples from Student-t weights, whose |error| level is a property
// of the whole group (16 sigmas there; a bracket that misses is only a lost shortcut, the exact route runs).
static bool bracket_ranks_m(int64_t n, int64_t m, int64_t k0, int64_t k1, double* r_lo, double* r_hi,
                            double sigmas = 8.0) {
  if (n < SEL_BR_MIN_N || m < 1) return false;
  const double frac = (double)m / (double)n, p = ((double)k0 + 0.5) / (double)n;
  const double margin = sigmas * __builtin_sqrt((double)m * p * (1.0 - p)) + 8.0;
  *r_lo = __builtin_floor((double)k0 * frac) - margin;
  *r_hi = __builtin_ceil((double)k1 * frac) + margin;
  return *r_lo >= 0.0 && *r_hi <= (double)(m - 1);
}
// ... with the sample k_sel_sample takes from a stored array
static bool bracket_ranks(int64_t n, int64_t k0, int64_t k1, int64_t* m_out, double* r_lo, double* r_hi) {
  *m_out = n / (SEL_CHUNK_LINES * 64) * 64;
  return bracket_ranks_m(n, *m_out, k0, k1, r_lo, r_hi);
}

// Workgroup size of the kernels that would otherwise run 1024 threads (the 128 KiB histograms, the one-launch
// selection of short arrays).  A 1024-thread workgroup wants four wave slots and their registers on every SIMD of a
// CU at once; beside a kernel that fills the chip it gets them only when a whole CU has drained, i.e. near that
// kernel's end (measured: 10 .. 20 times its time alone), while a 256-thread workgroup -- one wave per SIMD -- is
// placed as single waves retire.  Alone on the chip the large workgroups are faster (more loads in flight per
// histogram): mq_percentile_multi_beside selects the small ones for its call, everything else keeps the large.
static thread_local int tl_sel_wg_threads = 1024;

static int launch_accumulate(const SelBatch& b, int count, int64_t nmax, int32_t dtype, int level, hipStream_t st,
                             int guard = 0) {
  const int V = f32_keys(dtype) ? 4 : 2;
  if (level == 0) {
    const int threads = tl_sel_wg_threads;
    static const int capk = getenv("MQ_SEL_WINCAP") ? atoi(getenv("MQ_SEL_WINCAP")) : 512;
    if (threads == 256 && capk > 0) {  // beside a busy chip: the 9 KiB windowed histogram, several workgroups per CU
      int64_t nb = (nmax / V + 256 * 16 - 1) / (256 * 16);
      int64_t cap = capk / count < 4 ? 4 : capk / count;
      dim3 grid((unsigned)(nb < 1 ? 1 : (nb > cap ? cap : nb)), (unsigned)count);
      if (f32_keys(dtype)) hipLaunchKernelGGL((k_sel_hist_top_window<float>), grid, dim3(256), 0, st, b, guard);
      else hipLaunchKernelGGL((k_sel_hist_top_window<double>), grid, dim3(256), 0, st, b, guard);
      MQ_HIP(hipGetLastError());
      return MQ_OK;
    }
    int64_t nb = (nmax / V + threads * 4 - 1) / (threads * 4);
    int64_t cap = 256 / count < 4 ? 4 : 256 / count;  // one 128 KiB-LDS workgroup fits per CU
    dim3 grid((unsigned)(nb < 1 ? 1 : (nb > cap ? cap : nb)), (unsigned)count);
    if (f32_keys(dtype)) {
      MQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sel_hist_top<float>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, SEL_TOP_BINS * 4));
      hipLaunchKernelGGL((k_sel_hist_top<float>), grid, dim3(threads), SEL_TOP_BINS * 4, st, b, guard);
    } else {
      MQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sel_hist_top<double>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, SEL_TOP_BINS * 4));
      hipLaunchKernelGGL((k_sel_hist_top<double>), grid, dim3(threads), SEL_TOP_BINS * 4, st, b, guard);
    }
  } else {
    int64_t nb = (nmax / V + 256 * 4 - 1) / (256 * 4);
    int64_t cap = 1024 / count < 16 ? 16 : 1024 / count;
    dim3 grid((unsigned)(nb < 1 ? 1 : (nb > cap ? cap : nb)), (unsigned)count);
    if (f32_keys(dtype)) hipLaunchKernelGGL((k_sel_hist_digit<float>), grid, dim3(256), 0, st, b, level, guard);
    else hipLaunchKernelGGL((k_sel_hist_digit<double>), grid, dim3(256), 0, st, b, level, guard);
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
  if (n < 0 || (n > 0 && !x) || !sel_dtype_ok(dtype) || level < 0 || level > 3) return MQ_ERR_ARG;
  if (!workspace || workspace_bytes < sizeof(SelWs)) return MQ_ERR_WORKSPACE;
  if (!al16(x) || !al16(workspace)) return MQ_ERR_ALIGN;
  if (n == 0) return MQ_OK;
  SelBatch b = {};
  b.x[0] = x;
  b.n[0] = n;
  b.ws = static_cast<char*>(workspace);
  return launch_accumulate(b, 1, n, dtype, level, static_cast<hipStream_t>(hip_stream));
}

// The same for `count` arrays at once (xs, ns: HOST arrays): one launch per 64 arrays, all adding into the one
// selection of `workspace`.  A model-wide percentile over hundreds of short per-layer lists is otherwise a chain
// of launches that each clear and flush a whole histogram for a few thousand values.
extern "C" int mq_select_accumulate_multi(const void* const* xs, const int64_t* ns, int32_t count, int32_t dtype,
                                          int32_t level, void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (count < 0 || (count > 0 && (!xs || !ns)) || !sel_dtype_ok(dtype) || level < 0 || level > 3) return MQ_ERR_ARG;
  if (!workspace || workspace_bytes < sizeof(SelWs)) return MQ_ERR_WORKSPACE;
  if (!al16(workspace)) return MQ_ERR_ALIGN;
  for (int i = 0; i < count; ++i) {
    if (ns[i] < 0 || (ns[i] > 0 && !xs[i])) return MQ_ERR_ARG;
    if (!al16(xs[i])) return MQ_ERR_ALIGN;
  }
  int done = 0;
  while (done < count) {
    SelBatch b = {};
    int c = 0;
    int64_t nmax = 0;
    for (; done < count && c < SEL_MAX_BATCH; ++done) {
      if (ns[done] == 0) continue;
      b.x[c] = xs[done];
      b.n[c] = ns[done];
      nmax = ns[done] > nmax ? ns[done] : nmax;
      ++c;
    }
    if (c == 0) break;
    b.ws = static_cast<char*>(workspace);
    b.ws_stride = 0;  // every array of the launch adds into the same histograms
    int rc = launch_accumulate(b, c, nmax, dtype, level, static_cast<hipStream_t>(hip_stream));
    if (rc) return rc;
  }
  return MQ_OK;
}

// Step 3 / 5: locate the ranks in the (merged) histogram of `level` and extend the prefixes.  The ranks
// k0 <= k1 are read at level 0.  With finish != 0 (pass it at the last level: 1 for float32, 3 for
// float64) the two order statistics and numpy's interpolation between them are written, in `dtype`, to
// out (one value) and out_pair (two values); either may be NULL.
extern "C" int mq_select_scan(int32_t level, int64_t k0, int64_t k1, int32_t finish, int32_t dtype, double gamma,
                              void* out, void* out_pair, void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (level < 0 || level > 3 || (level == 0 && (k0 < 0 || k1 < k0))) return MQ_ERR_ARG;
  if (finish && !sel_dtype_ok(dtype)) return MQ_ERR_ARG;
  if (!workspace || workspace_bytes < sizeof(SelWs)) return MQ_ERR_WORKSPACE;
  SelBatch b = {};
  b.ws = static_cast<char*>(workspace);
  SelRanks rk = {};
  rk.k0[0] = (uint64_t)k0;
  rk.k1[0] = (uint64_t)k1;
  rk.gamma[0] = gamma;
  FinishArgs fin = {finish ? dtype : -1, static_cast<char*>(out), static_cast<char*>(out_pair)};
  hipLaunchKernelGGL(k_sel_scan, dim3(1), dim3(256), 0, static_cast<hipStream_t>(hip_stream), b, (int)level, rk, fin, 0);
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}

// One group of <= 64 arrays that all take the same route.
static int select_group(const void* const* xs, const int64_t* ns, int c, int32_t dtype, double pct, bool bracket,
                        char* out, char* out_pair, void* workspace, hipStream_t st) {
  SelBatch b = {};
  SelRanks rk = {};
  SelSampleRanks sr = {};
  int64_t nmax = 0, mmax = 0;
  for (int i = 0; i < c; ++i) {
    b.x[i] = xs[i];
    b.n[i] = ns[i];
    nmax = ns[i] > nmax ? ns[i] : nmax;
    int64_t k0, k1;
    int rc = mq_percentile_index(dtype, ns[i], pct, &k0, &k1, &rk.gamma[i]);
    if (rc) return rc;
    rk.k0[i] = (uint64_t)k0;
    rk.k1[i] = (uint64_t)k1;
    if (bracket) {
      double lo, hi;
      int64_t m;
      if (!bracket_ranks(ns[i], k0, k1, &m, &lo, &hi)) return MQ_ERR_ARG;  // grouping bug
      sr.lo[i] = (uint64_t)lo;
      sr.hi[i] = (uint64_t)hi;
      mmax = m > mmax ? m : mmax;
    }
  }
  b.ws = static_cast<char*>(workspace);
  b.ws_stride = (int64_t)sizeof(SelWs);
  const int levels = f32_keys(dtype) ? 1 : 3;
  // clear what this selection accumulates into: header + level histograms, and the sample area
  const uint32_t a_end = (uint32_t)ws_used_bytes(dtype);
  const uint32_t b_begin = bracket ? (uint32_t)offsetof(SelWs, samp_coarse) : 0u;
  const uint32_t b_end = bracket ? (uint32_t)offsetof(SelWs, br_coarse) : 0u;
  hipLaunchKernelGGL(k_sel_clear, dim3((unsigned)(1024 / c < 8 ? 8 : 1024 / c), (unsigned)c), dim3(256), 0, st, b, a_end,
                     b_begin, b_end);
  MQ_HIP(hipGetLastError());
  FinishArgs fin_last = {dtype, out, out_pair};
  if (bracket) {
    if (tl_sel_wg_threads == 256) {
      int64_t nb = (mmax / 4 + 256 * 8 - 1) / (256 * 8);
      int64_t cap = 1024 / c < 8 ? 8 : 1024 / c;
      hipLaunchKernelGGL(k_sel_sample_window, dim3((unsigned)(nb < 1 ? 1 : (nb > cap ? cap : nb)), (unsigned)c), dim3(256), 0,
                         st, b);
    } else {
      int64_t nb = (mmax / 4 + 1024 * 8 - 1) / (1024 * 8);
      int64_t cap = 256 / c < 2 ? 2 : 256 / c;
      MQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sel_sample), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 SEL_TOP_BINS * 4));
      hipLaunchKernelGGL(k_sel_sample, dim3((unsigned)(nb < 1 ? 1 : (nb > cap ? cap : nb)), (unsigned)c), dim3(1024),
                         SEL_TOP_BINS * 4, st, b);
    }
    hipLaunchKernelGGL(k_sel_scan_sample, dim3((unsigned)c), dim3(256), 0, st, b, sr);
    hipLaunchKernelGGL(k_sel_clear_bracket, dim3((unsigned)(512 / c < 8 ? 8 : 512 / c), (unsigned)c), dim3(256), 0, st, b);
    int64_t nb = (nmax / 4 + 256 * 8 - 1) / (256 * 8);
    int64_t cap = 2048 / c < 16 ? 16 : 2048 / c;
    hipLaunchKernelGGL(k_sel_bracket, dim3((unsigned)(nb > cap ? cap : nb), (unsigned)c), dim3(256), 0, st, b);
    hipLaunchKernelGGL(k_sel_scan_bracket, dim3((unsigned)c), dim3(256), 0, st, b, rk, fin_last, 0);
    MQ_HIP(hipGetLastError());
  }
  const int guard = bracket ? 1 : 0;  // the exact route returns at once for slices the bracket resolved
  for (int level = 0; level <= levels; ++level) {
    int rc = launch_accumulate(b, c, nmax, dtype, level, st, guard);
    if (rc) return rc;
    FinishArgs fin = fin_last;
    if (level != levels) fin.dtype = -1;
    hipLaunchKernelGGL(k_sel_scan, dim3((unsigned)c), dim3(256), 0, st, b, level, rk, fin, guard);
    MQ_HIP(hipGetLastError());
  }
  return MQ_OK;
}

// Arrays of up to SEL_SMALL_N values: one launch, one workgroup per array, no workspace.
static int select_small(const void* const* xs, const int64_t* ns, int c, int32_t dtype, double pct, char* out,
                        char* out_pair, hipStream_t st) {
  SelBatch b = {};
  SelRanks rk = {};
  for (int i = 0; i < c; ++i) {
    b.x[i] = xs[i];
    b.n[i] = ns[i];
    int64_t k0, k1;
    int rc = mq_percentile_index(dtype, ns[i], pct, &k0, &k1, &rk.gamma[i]);
    if (rc) return rc;
    rk.k0[i] = (uint64_t)k0;
    rk.k1[i] = (uint64_t)k1;
  }
  FinishArgs fin = {dtype, out, out_pair};
  if (tl_sel_wg_threads == 256) {
    if (f32_keys(dtype)) hipLaunchKernelGGL((k_sel_small<float, 256>), dim3((unsigned)c), dim3(256), 0, st, b, rk, fin);
    else hipLaunchKernelGGL((k_sel_small<double, 256>), dim3((unsigned)c), dim3(256), 0, st, b, rk, fin);
  } else {
    if (f32_keys(dtype)) hipLaunchKernelGGL((k_sel_small<float, 1024>), dim3((unsigned)c), dim3(1024), 0, st, b, rk, fin);
    else hipLaunchKernelGGL((k_sel_small<double, 1024>), dim3((unsigned)c), dim3(1024), 0, st, b, rk, fin);
  }
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}

// np.percentile(xs[i], pct) for `count` non-negative arrays of ns[i] values (one dtype), one result per
// array in out[i] (and out_pair[2 i], out_pair[2 i + 1]).  Arrays travel in groups of <= 64 per set of
// launches: 5 launches for float32 (clear, histogram, scan, matched-digit histogram, scan + interpolation),
// 9 for float64; float32 arrays of 2^21 values and more go first through sample + bracket (5 launches in
// front).  xs and ns are HOST arrays.
extern "C" int mq_percentile_multi(const void* const* xs, const int64_t* ns, int32_t count, int32_t dtype, double pct,
                                   void* out, void* out_pair, void* workspace, size_t workspace_bytes,
                                   void* hip_stream) {
  if (count < 0 || (count > 0 && (!xs || !ns)) || (!out && !out_pair)) return MQ_ERR_ARG;
  if (!sel_dtype_ok(dtype)) return MQ_ERR_ARG;
  if (count == 0) return MQ_OK;
  if (!workspace || workspace_bytes < mq_percentile_batch_workspace_bytes(count > SEL_MAX_BATCH ? SEL_MAX_BATCH : count))
    return MQ_ERR_WORKSPACE;
  if (!al16(workspace)) return MQ_ERR_ALIGN;
  for (int i = 0; i < count; ++i) {
    if (!xs[i] || ns[i] <= 0) return MQ_ERR_ARG;
    if (!al16(xs[i])) return MQ_ERR_ALIGN;
  }
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const size_t esz = dtype == MQ_F32 ? 4 : 8;  // of the results
  // consecutive arrays that take the same route form a group
  // 0: exact multi-launch route, 1: sample + bracket in front of it, 2: short array, one launch
  auto route = [&](int i) {
    int64_t k0, k1, m;
    double g, lo, hi;
    if (ns[i] <= SEL_SMALL_N) return 2;
    if (mq_percentile_index(dtype, ns[i], pct, &k0, &k1, &g)) return 0;
    return f32_keys(dtype) && bracket_ranks(ns[i], k0, k1, &m, &lo, &hi) ? 1 : 0;
  };
  int done = 0;
  while (done < count) {
    const int rt = route(done);
    int c = 1;
    while (done + c < count && c < SEL_MAX_BATCH && route(done + c) == rt) ++c;
    char* o = out ? static_cast<char*>(out) + (size_t)done * esz : nullptr;
    char* op = out_pair ? static_cast<char*>(out_pair) + (size_t)done * 2 * esz : nullptr;
    int rc = rt == 2 ? select_small(xs + done, ns + done, c, dtype, pct, o, op, st)
                     : select_group(xs + done, ns + done, c, dtype, pct, rt == 1, o, op, workspace, st);
    if (rc) return rc;
    done += c;
  }
  return MQ_OK;
}

// The same for arrays of one length.
// mq_percentile_multi for a call that runs BESIDE a kernel that fills the chip (on another stream): the same
// selections with 256-thread workgroups where that one launches 1024-thread ones (tl_sel_wg_threads).
extern "C" int mq_percentile_multi_beside(const void* const* xs, const int64_t* ns, int32_t count, int32_t dtype,
                                          double pct, void* out, void* out_pair, void* workspace, size_t workspace_bytes,
                                          void* hip_stream) {
  tl_sel_wg_threads = 256;
  const int rc = mq_percentile_multi(xs, ns, count, dtype, pct, out, out_pair, workspace, workspace_bytes, hip_stream);
  tl_sel_wg_threads = 1024;
  return rc;
}

extern "C" int mq_percentile_batch(const void* const* xs, int32_t count, int32_t dtype, int64_t n, double pct,
                                   void* out, void* out_pair, void* workspace, size_t workspace_bytes,
                                   void* hip_stream) {
  if (count < 0 || n <= 0) return MQ_ERR_ARG;
  int64_t ns[SEL_MAX_BATCH];
  for (int i = 0; i < SEL_MAX_BATCH; ++i) ns[i] = n;
  const size_t esz = dtype == MQ_F32 ? 4 : 8;
  for (int done = 0; done < count; done += SEL_MAX_BATCH) {
    const int c = count - done < SEL_MAX_BATCH ? count - done : SEL_MAX_BATCH;
    int rc = mq_percentile_multi(xs ? xs + done : nullptr, ns, c, dtype, pct,
                                 out ? static_cast<char*>(out) + (size_t)done * esz : nullptr,
                                 out_pair ? static_cast<char*>(out_pair) + (size_t)done * 2 * esz : nullptr, workspace,
                                 workspace_bytes, hip_stream);
    if (rc) return rc;
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

// ---- list percentile fed by the statistics pass (see mq_sel.h) -------------------------------------------------
#define SEL_LIST_SIGMAS 12.0
// The wanted ranks of a list of n entries and the sample ranks of its bracket for m sample entries; false: not
// worth feeding (short lists, quantiles too near an end for the margin).
bool sel_list_ranks(int64_t n, int64_t m, double pct, int64_t* k0, int64_t* k1, double* gamma, int64_t* s_lo, int64_t* s_hi) {
  if (mq_percentile_index(MQ_F32_AS_F64, n, pct, k0, k1, gamma)) return false;
  if (m < 1 || n < 1 || n >= (1ll << 32)) return false;
  const double frac = (double)m / (double)n, p = ((double)*k0 + 0.5) / (double)n;
  const double margin = SEL_LIST_SIGMAS * __builtin_sqrt((double)m * p * (1.0 - p)) + 8.0;
  const double lo = __builtin_floor((double)*k0 * frac) - margin, hi = __builtin_ceil((double)*k1 * frac) + margin;
  if (!(lo >= 0.0 && hi <= (double)(m - 1))) return false;
  *s_lo = (int64_t)lo;
  *s_hi = (int64_t)hi;
  return true;
}
int sel_list_bracket(const ListJob* jobs, int count, hipStream_t st) {
  if (count < 1) return MQ_OK;
  hipLaunchKernelGGL(k_list_bracket, dim3((unsigned)count), dim3(LIST_THREADS), 0, st, jobs);
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}
int sel_list_cand(const ListJob* jobs, int count, ListFeedState* state, double* out, hipStream_t st) {
  if (count < 1) return MQ_OK;
  hipLaunchKernelGGL(k_list_cand, dim3((unsigned)count), dim3(LIST_THREADS), 0, st, jobs, state, out);
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}
int sel_list_repair(const ListJob* jobs, int count, const ListFeedState* state, double* out, hipStream_t st) {
  if (count < 1) return MQ_OK;
  hipLaunchKernelGGL(k_list_repair, dim3((unsigned)count), dim3(LIST_THREADS), 0, st, jobs, state, out);
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}

// ---- selection fed by another kernel (see mq_sel.h) ------------------------------------------------------
size_t sel_ws_bytes() { return sizeof(SelWs); }

bool sel_feed_eligible(int64_t n, int64_t m, double pct) {
  int64_t k0, k1;
  double g, lo, hi;
  if (mq_percentile_index(MQ_F32, n, pct, &k0, &k1, &g)) return false;
  return bracket_ranks_m(n, m, k0, k1, &lo, &hi, SEL_FEED_SIGMAS);
}

static void feed_batch(void* ws, int count, const void* const* xs, int64_t n, SelBatch* b) {
  *b = SelBatch{};
  for (int i = 0; i < count; ++i) {
    b->x[i] = xs ? xs[i] : nullptr;
    b->n[i] = n;
  }
  b->ws = static_cast<char*>(ws);
  b->ws_stride = (int64_t)sizeof(SelWs);
}

int sel_feed_begin(void* ws, int count, hipStream_t st) {
  SelBatch b;
  feed_batch(ws, count, nullptr, 0, &b);
  hipLaunchKernelGGL(k_sel_clear, dim3((unsigned)(1024 / count < 8 ? 8 : 1024 / count), (unsigned)count), dim3(256), 0, st,
                     b, (uint32_t)ws_used_bytes(MQ_F32), (uint32_t)offsetof(SelWs, samp_coarse),
                     (uint32_t)offsetof(SelWs, br_coarse));
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}

int sel_feed_bracket(void* ws, int count, const uint16_t* samples, int64_t sample_stride, int64_t n, int64_t m,
                     double pct, hipStream_t st) {
  int64_t k0, k1;
  double g, lo, hi;
  int rc = mq_percentile_index(MQ_F32, n, pct, &k0, &k1, &g);
  if (rc) return rc;
  if (!bracket_ranks_m(n, m, k0, k1, &lo, &hi, SEL_FEED_SIGMAS)) return MQ_ERR_ARG;
  SelBatch b;
  feed_batch(ws, count, nullptr, n, &b);
  SelSampleRanks sr = {};
  for (int i = 0; i < count; ++i) {
    sr.lo[i] = (uint64_t)lo;
    sr.hi[i] = (uint64_t)hi;
  }
  // (each workgroup's flush is ~1500 global atomics into the array's counters, and the chip takes about 120 of them
  //  per microsecond: 48 arrays x 512 Ki samples took 26 / 22 / 28 / 38 us with 10 / 16 / 24 / 42 workgroups per
  //  array, 11 us of it the loads -- scratch/sample_hist_probe.hip)
  int64_t nb = (m / 8 + 256 * 16 - 1) / (256 * 16);
  const int64_t cap = 768 / count < 4 ? 4 : 768 / count;
  hipLaunchKernelGGL(k_sel_sample_u16_window, dim3((unsigned)(nb < 1 ? 1 : (nb > cap ? cap : nb)), (unsigned)count), dim3(256),
                     0, st, b, samples, sample_stride, m);
  hipLaunchKernelGGL(k_sel_scan_sample, dim3((unsigned)count), dim3(256), 0, st, b, sr);
  hipLaunchKernelGGL(k_sel_clear_bracket, dim3((unsigned)(512 / count < 8 ? 8 : 512 / count), (unsigned)count), dim3(256),
                     0, st, b);
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}

int sel_feed_finish(void* ws, int count, const void* const* xs, int64_t n, double pct, void* out, hipStream_t st,
                    int part) {
  SelBatch b;
  feed_batch(ws, count, xs, n, &b);
  SelRanks rk = {};
  for (int i = 0; i < count; ++i) {
    int64_t k0, k1;
    int rc = mq_percentile_index(MQ_F32, n, pct, &k0, &k1, &rk.gamma[i]);
    if (rc) return rc;
    rk.k0[i] = (uint64_t)k0;
    rk.k1[i] = (uint64_t)k1;
  }
  FinishArgs fin_last = {MQ_F32, static_cast<char*>(out), nullptr};
  if (part != SEL_FINISH_EXACT) {
    hipLaunchKernelGGL(k_sel_scan_bracket, dim3((unsigned)count), dim3(256), 0, st, b, rk, fin_last, 1);
    hipLaunchKernelGGL(k_sel_cand, dim3((unsigned)count), dim3(SEL_CAND_THREADS), 0, st, b, rk, fin_last);
    MQ_HIP(hipGetLastError());
  }
  if (part == SEL_FINISH_BRACKET) return MQ_OK;
  for (int level = 0; level <= 1; ++level) {  // the exact route, skipped per slice when the bracket held
    int rc = launch_accumulate(b, count, n, MQ_F32, level, st, 1);
    if (rc) return rc;
    FinishArgs fin = fin_last;
    if (level != 1) fin.dtype = -1;
    hipLaunchKernelGGL(k_sel_scan, dim3((unsigned)count), dim3(256), 0, st, b, level, rk, fin, 1);
    MQ_HIP(hipGetLastError());
  }
  return MQ_OK;
}
