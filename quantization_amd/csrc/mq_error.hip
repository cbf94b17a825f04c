// Error statistics between two stored tensors (baseline a, quantized b) on gfx950 (MI355X) -- the
// arithmetic of the reference's analysis scripts, which only ever see the two arrays:
//   MSE&P99/quant_error_multi_weight.py:55-87    tensor MSE; per-COLUMN / per-column-block MSE vector
//   MSE&P99/quant_error_single_weight.py:46-79   per-ROW / per-(row, group) MSE vector
//   elwise_error/onnx_error_cli.py:90-127        float32 error, |error|, its sum (and P99 via mq_percentile)
//   elwise_error/single_weight_quant_error.py:39-52  float16 error, bucket index by np.digitize(right=True)
//   elwise_error/compare.py:99-125               per-format |error|, element-wise minimum, outlier / best masks
// All kernels stream both inputs once with 16-byte loads; they are HBM-bandwidth bound.
// Squares are rounded to float32 per element (as numpy's float32 arithmetic does); a lane's 8 values are
// summed pairwise in float32 and everything above that in float64, in a fixed order.
#include "mq_common.h"

struct ErrParams {
  const void* a;
  const void* b;
  int32_t dtype;  // MQ_F16 or MQ_F32 (both inputs)
  int64_t rows, cols, n;
  int64_t seg_len;  // row-form segment length (elements), 0 = none
  void* abs_err;
  int32_t abs_err_mode;  // 0 none, 1 float32 |a32 - b32|, 2 float16 |h(a - b)| (float16 inputs only)
  double* seg_sse;
  double* col_part;  // [row_chunks][cols] partial column sums, or null
  int32_t row_chunks;
  int32_t cols_totals;  // k_err_cols also leaves the workgroup's sum of squares / |error| in partials[]
  double* partials;  // [grid][2]
};

// 8 consecutive values widened to float32
__device__ __forceinline__ void load8(const void* p, int dtype, int64_t e, float* x) {
  if (dtype == MQ_F32) {
    const f32x4* q = reinterpret_cast<const f32x4*>(static_cast<const float*>(p) + e);
    f32x4 u = __builtin_nontemporal_load(&q[0]), v = __builtin_nontemporal_load(&q[1]);  // (touched once)
#pragma unroll
    for (int k = 0; k < 4; ++k) { x[k] = u[k]; x[4 + k] = v[k]; }
  } else {
    u32x4 r = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(p) + e));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      x[2 * k] = (float)h_from_bits(r[k] & 0xffffu);
      x[2 * k + 1] = (float)h_from_bits(r[k] >> 16);
    }
  }
}
__device__ __forceinline__ float load1(const void* p, int dtype, int64_t e) {
  return dtype == MQ_F32 ? static_cast<const float*>(p)[e] : (float)h_from_bits(static_cast<const uint16_t*>(p)[e]);
}
__device__ __forceinline__ float tree8(const float* q) {
  return ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
}

// error of one 8-element vector: stores |err| if asked, returns the lane's float32 partial sums
__device__ __forceinline__ void err8(const ErrParams& p, int64_t e0, float& psq, float& pab) {
  float x[8], y[8], e[8], q[8], ab[8];
  load8(p.a, p.dtype, e0, x);
  load8(p.b, p.dtype, e0, y);
#pragma unroll
  for (int i = 0; i < 8; ++i) { e[i] = x[i] - y[i]; q[i] = e[i] * e[i]; ab[i] = __builtin_fabsf(e[i]); }
  if (p.abs_err_mode == 1) {
    f32x4* o = reinterpret_cast<f32x4*>(static_cast<float*>(p.abs_err) + e0);
    f32x4 o0 = {ab[0], ab[1], ab[2], ab[3]}, o1 = {ab[4], ab[5], ab[6], ab[7]};
    __builtin_nontemporal_store(o0, &o[0]);
    __builtin_nontemporal_store(o1, &o[1]);
  } else if (p.abs_err_mode == 2) {
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (f2h_bits(e[2 * k]) | (f2h_bits(e[2 * k + 1]) << 16)) & 0x7fff7fffu;
    __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(static_cast<uint16_t*>(p.abs_err) + e0));
  }
  psq = tree8(q);
  pab = tree8(ab);
}

__device__ __forceinline__ void err_block_partials(const ErrParams& p, double sse, double sabs) {
  __shared__ double lds[32];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  sse = wave_sum(sse);
  sabs = wave_sum(sabs);
  if (lane == 0) { lds[wv] = sse; lds[16 + wv] = sabs; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double u = 0, v = 0;
    for (int i = 0; i < nw; ++i) { u += lds[i]; v += lds[16 + i]; }
    p.partials[2 * (int64_t)blockIdx.x] = u;
    p.partials[2 * (int64_t)blockIdx.x + 1] = v;
  }
}

// Row form: segments of seg_len elements (seg_len % 8 == 0).  A wave owns whole segments; segments
// shorter than a wave are packed several to a wave (team of the next power of two lanes).
__global__ __launch_bounds__(256) void k_err_rows(ErrParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int64_t L = p.seg_len ? p.seg_len : p.n;  // no segments: the flat tensor is one run
  const int64_t vps = L / 8, nseg = p.n / L;
  double acc_sq = 0.0, acc_ab = 0.0;
  if (vps >= 64 || !p.seg_sse) {
    if (!p.seg_sse) {  // totals / dump only: flat grid-stride sweep
      const int64_t nvec = p.n / 8;
      for (int64_t v = gw * 64 + lane; v < nvec; v += nwaves * 64) {
        float a, b;
        err8(p, v * 8, a, b);
        acc_sq += (double)a;
        acc_ab += (double)b;
      }
    } else {
      for (int64_t seg = gw; seg < nseg; seg += nwaves) {
        double s = 0.0;
        for (int64_t v = lane; v < vps; v += 64) {
          float a, b;
          err8(p, seg * L + v * 8, a, b);
          s += (double)a;
          acc_ab += (double)b;
        }
        acc_sq += s;
        s = wave_sum(s);
        if (lane == 0) p.seg_sse[seg] = s;
      }
    }
  } else {
    int tlog = 0;
    while ((1 << tlog) < vps) ++tlog;
    const int T = 1 << tlog, spw = 64 >> tlog;
    const int tl = lane & (T - 1), team = lane >> tlog;
    const int64_t tiles = (nseg + spw - 1) / spw;
    for (int64_t tile = gw; tile < tiles; tile += nwaves) {
      const int64_t seg = tile * spw + team;
      const bool valid = tl < vps && seg < nseg;
      float a = 0.f, b = 0.f;
      if (valid) err8(p, seg * L + (int64_t)tl * 8, a, b);
      double s = (double)a;
      acc_sq += s;
      acc_ab += (double)b;
      for (int off = 1; off < T; off <<= 1) s += __shfl_xor(s, off);
      if (valid && tl == 0) p.seg_sse[seg] = s;
    }
  }
  err_block_partials(p, acc_sq, acc_ab);
}

// Column form: a workgroup owns 64 column vectors (512 columns) of one row chunk rc; its four waves take every
// fourth row of the chunk's rows (rows rc, rc + row_chunks, ...), four rows in flight per lane -- the loop is bound
// by load latency, so the chip wants many waves with several loads each -- and their partial sums meet in LDS in a
// fixed order.  Partial column sums go to col_part[rc][col] and are added up in a fixed order by k_err_cols_finish.
__global__ __launch_bounds__(256) void k_err_cols(ErrParams p) {
  __shared__ double part[3][64][8];
  const int64_t cvecs = p.cols / 8, cblocks = (cvecs + 63) / 64;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t cv = (int64_t)(blockIdx.x % cblocks) * 64 + lane, rc = blockIdx.x / cblocks;
  const bool in = cv < cvecs;
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double sab = 0.0;  // (cols_totals: this lane's sum of |error|, so that one pass over a and b serves both forms)
  constexpr int U = 4;
  const int64_t stride = (int64_t)p.row_chunks * 4;  // rows between two rows of one wave
  int64_t r = rc + (int64_t)wv * p.row_chunks;
  if (in) {
    for (; r + (U - 1) * stride < p.rows; r += U * stride) {
      float x[U][8], y[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        load8(p.a, p.dtype, (r + u * stride) * p.cols + cv * 8, x[u]);
        load8(p.b, p.dtype, (r + u * stride) * p.cols + cv * 8, y[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float ab[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float e = x[u][i] - y[u][i];
          acc[i] += (double)(e * e);
          ab[i] = __builtin_fabsf(e);
        }
        if (p.cols_totals) sab += (double)tree8(ab);
      }
    }
    for (; r < p.rows; r += stride) {
      float x[8], y[8], ab[8];
      load8(p.a, p.dtype, r * p.cols + cv * 8, x);
      load8(p.b, p.dtype, r * p.cols + cv * 8, y);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float e = x[i] - y[i];
        acc[i] += (double)(e * e);
        ab[i] = __builtin_fabsf(e);
      }
      if (p.cols_totals) sab += (double)tree8(ab);
    }
  }
  if (p.cols_totals) {  // (uniform; every thread of the workgroup takes part)
    const double ssq = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    err_block_partials(p, ssq, sab);
  }
  if (wv > 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) part[wv - 1][lane][i] = acc[i];
  }
  __syncthreads();
  if (wv == 0 && in) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      p.col_part[rc * p.cols + cv * 8 + i] = ((acc[i] + part[0][lane][i]) + part[1][lane][i]) + part[2][lane][i];
  }
}
// The same for any column count / alignment: thread (c, rc) owns one column (consecutive lanes read consecutive
// elements of a row).
__global__ __launch_bounds__(256) void k_err_cols_scalar(ErrParams p) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= p.cols * p.row_chunks) return;
  const int64_t c = t % p.cols, rc = t / p.cols;
  double acc = 0.0;
  for (int64_t r = rc; r < p.rows; r += p.row_chunks) {
    const float e = load1(p.a, p.dtype, r * p.cols + c) - load1(p.b, p.dtype, r * p.cols + c);
    acc += (double)(e * e);
  }
  p.col_part[rc * p.cols + c] = acc;
}
// col_sse[c] = sum over the row chunks, in a fixed order: a workgroup owns 64 columns, wave w adds chunks w, w + 4,
// ... (four loads in flight), the four partial sums meet in LDS.
__global__ __launch_bounds__(256) void k_err_cols_finish(const double* part, int row_chunks, int64_t cols,
                                                         double* col_sse) {
  __shared__ double ex[3][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + lane;
  double s = 0.0;
  if (c < cols) {
    int r = wv;
    for (; r + 12 < row_chunks; r += 16) {
      const double a0 = part[(int64_t)r * cols + c], a1 = part[(int64_t)(r + 4) * cols + c];
      const double a2 = part[(int64_t)(r + 8) * cols + c], a3 = part[(int64_t)(r + 12) * cols + c];
      s = (((s + a0) + a1) + a2) + a3;
    }
    for (; r < row_chunks; r += 4) s += part[(int64_t)r * cols + c];
  }
  if (wv > 0) ex[wv - 1][lane] = s;
  __syncthreads();
  if (wv == 0 && c < cols) col_sse[c] = ((s + ex[0][lane]) + ex[1][lane]) + ex[2][lane];
}

// scalar fallback for shapes that are not whole, aligned 8-element vectors: totals and row segments
__global__ __launch_bounds__(256) void k_err_generic(ErrParams p) {
  __shared__ double ex[16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int64_t L = p.seg_len ? p.seg_len : p.n, nseg = p.n / L;
  double acc_sq = 0.0, acc_ab = 0.0;
  if (!p.seg_sse) {  // totals / dump only: every workgroup takes its share of the flat tensor
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * blockDim.x) {
      float x = load1(p.a, p.dtype, i), y = load1(p.b, p.dtype, i);
      float e = x - y, ab = __builtin_fabsf(e);
      if (p.abs_err_mode == 1) static_cast<float*>(p.abs_err)[i] = ab;
      else if (p.abs_err_mode == 2) static_cast<uint16_t*>(p.abs_err)[i] = (uint16_t)(f2h_bits(e) & 0x7fffu);
      acc_sq += (double)(e * e);
      acc_ab += (double)ab;
    }
    err_block_partials(p, acc_sq, acc_ab);
    return;
  }
  for (int64_t seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < L; i += blockDim.x) {
      float x = load1(p.a, p.dtype, seg * L + i), y = load1(p.b, p.dtype, seg * L + i);
      float e = x - y, ab = __builtin_fabsf(e);
      if (p.abs_err_mode == 1) static_cast<float*>(p.abs_err)[seg * L + i] = ab;
      else if (p.abs_err_mode == 2) static_cast<uint16_t*>(p.abs_err)[seg * L + i] = (uint16_t)(f2h_bits(e) & 0x7fffu);
      s += (double)(e * e);
      acc_ab += (double)ab;
    }
    acc_sq += s;
    if (p.seg_sse) {
      s = wave_sum(s);
      __syncthreads();
      if (lane == 0) ex[wv] = s;
      __syncthreads();
      if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < nw; ++i) t += ex[i];
        p.seg_sse[seg] = t;
      }
    }
  }
  err_block_partials(p, acc_sq, acc_ab);
}

__global__ __launch_bounds__(256) void k_err_finalize(const double* partials, int nblocks, double* totals) {
  __shared__ double lds[32];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x) { a += partials[2 * i]; b += partials[2 * i + 1]; }
  a = wave_sum(a);
  b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) { lds[threadIdx.x >> 6] = a; lds[16 + (threadIdx.x >> 6)] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double x = 0.0, y = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { x += lds[i]; y += lds[16 + i]; }
    totals[0] = x;
    totals[1] = y;
  }
}

// np.digitize(abs_err, bins, right=True) clipped to nbins - 1, for a float16 |error| array:
// idx = #{bins[j] < x} (single_weight_quant_error.py:43-51).  bins: ascending float64, nbins <= 64.
struct BucketParams {
  const uint16_t* abs_err;
  int64_t n;
  int32_t nbins;
  double bins[64];
  uint8_t* idx;
};
__global__ __launch_bounds__(256) void k_bucketize(BucketParams p) {
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = t0; i < p.n; i += step) {
    double x = (double)(float)h_from_bits(p.abs_err[i]);
    int c = 0;
    for (int j = 0; j < p.nbins; ++j) c += p.bins[j] < x ? 1 : 0;
    p.idx[i] = (uint8_t)(c < p.nbins - 1 ? c : p.nbins - 1);
  }
}

// compare.py:99-125 for up to 7 formats: per element e_f = |h(w - q_f)| (float16 arithmetic),
// m = min_f e_f, outlier = m > thr (thr as float16), best_f = (e_f == m) & !outlier.
// flags[i]: bit f = best_f, bit 7 = outlier;  min_err[i] = m (float16 bits), optional.
struct CompareParams {
  const uint16_t* w;
  const uint16_t* q[7];
  int32_t nfmt;
  int64_t n;
  uint32_t thr_bits;
  uint8_t* flags;
  uint16_t* min_err;
};
__global__ __launch_bounds__(256) void k_compare(CompareParams p) {
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
  const half_t thr = h_from_bits(p.thr_bits);
  for (int64_t i = t0; i < p.n; i += step) {
    const half_t w = h_from_bits(p.w[i]);
    half_t e[7], m = h_from_bits(0x7c00u);
    bool nan = false;
    for (int f = 0; f < p.nfmt; ++f) {
      half_t d = w - h_from_bits(p.q[f][i]);
      e[f] = h_from_bits(h_bits(d) & 0x7fffu);
      nan |= e[f] != e[f];
      m = e[f] < m ? e[f] : m;
    }
    if (nan) m = h_from_bits(0x7e00u);  // numpy's min propagates NaN
    uint32_t fl = (m > thr) ? 0x80u : 0u;
    if (!fl)
      for (int f = 0; f < p.nfmt; ++f) fl |= (e[f] == m) ? (1u << f) : 0u;
    p.flags[i] = (uint8_t)fl;
    if (p.min_err) p.min_err[i] = (uint16_t)h_bits(m);
  }
}

// The same for 8 consecutive elements per lane (16-byte loads, packed float16 subtraction, 16-bit integer
// min / max / compares on the sign-cleared bits: for non-negative float16 values, and NaN above 0x7c00, the
// integer order is the float order).
__device__ __forceinline__ uint32_t err_pk_min_u16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t err_pk_max_u16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <int NF>
__global__ __launch_bounds__(256) void k_compare_vec(CompareParams p) {
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
  const int64_t nvec = p.n / 8;
  const uint32_t thr = p.thr_bits & 0xffffu;
  const bool thr_nan = (thr & 0x7fffu) > 0x7c00u, thr_neg = (thr & 0x8000u) != 0;
  for (int64_t v = t0; v < nvec; v += step) {
    const u32x4 w = __builtin_nontemporal_load(&reinterpret_cast<const u32x4*>(p.w)[v]);  // every byte is touched once
    // (plain arrays: with ext_vector variables as the asm operands hipcc 7.2 used word 0 for every k)
    uint32_t e[NF][4], mn[4], mx[4], mo[4];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const u32x4 q = __builtin_nontemporal_load(&reinterpret_cast<const u32x4*>(p.q[f])[v]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        h16x2 d = __builtin_bit_cast(h16x2, (uint32_t)w[k]) - __builtin_bit_cast(h16x2, (uint32_t)q[k]);
        e[f][k] = __builtin_bit_cast(uint32_t, d) & 0x7fff7fffu;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mn[k] = e[0][k];
      mx[k] = e[0][k];
#pragma unroll
      for (int f = 1; f < NF; ++f) {
        mn[k] = err_pk_min_u16(mn[k], e[f][k]);
        mx[k] = err_pk_max_u16(mx[k], e[f][k]);
      }
    }
    uint32_t fl[2] = {0u, 0u};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = i >> 1, sh = (i & 1) * 16;
      uint32_t m = (mn[k] >> sh) & 0xffffu;
      const bool nan = ((mx[k] >> sh) & 0xffffu) > 0x7c00u;
      // m > thr as float16 values (m >= +0): false for a NaN threshold, true for any negative one except -0 vs 0
      const bool out = thr_nan ? false : (thr_neg ? ((thr & 0x7fffu) != 0u || m != 0u) : m > thr);
      uint32_t f8 = out ? 0x80u : 0u;
      if (!out) {
#pragma unroll
        for (int f = 0; f < NF; ++f) f8 |= (((e[f][k] >> sh) & 0xffffu) == m) ? (1u << f) : 0u;
      }
      if (nan) { f8 = 0u; m = 0x7e00u; }  // numpy's min propagates NaN; every comparison with it is false
      fl[i >> 2] |= f8 << ((i & 3) * 8);
      if (i & 1) mo[k] |= m << 16; else mo[k] = m;
    }
    __builtin_nontemporal_store(u32x2{fl[0], fl[1]}, &reinterpret_cast<u32x2*>(p.flags)[v]);
    if (p.min_err) __builtin_nontemporal_store(u32x4{mo[0], mo[1], mo[2], mo[3]}, &reinterpret_cast<u32x4*>(p.min_err)[v]);
  }
}

template <int NF>
static void launch_compare_vec(const CompareParams& p, hipStream_t st) {
  int64_t blocks = (p.n / 8 + 255) / 256;
  hipLaunchKernelGGL(k_compare_vec<NF>, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, st, p);
}

// ================================================================================================
static const int kErrMaxGrid = MQ_MAX_GRID;

extern "C" size_t mq_error_workspace_bytes(int64_t rows, int64_t cols, int32_t want_cols) {
  size_t b = (size_t)kErrMaxGrid * 2 * sizeof(double);
  if (want_cols) b += (size_t)64 * (size_t)cols * sizeof(double);
  (void)rows;
  return b;
}

static int error_stats_impl(const void* a, const void* b, int32_t dtype, int64_t rows, int64_t cols, int64_t seg_len,
                            void* abs_err, int32_t abs_err_dtype, double* seg_sse, double* col_sse, double* totals,
                            void* workspace, size_t workspace_bytes, hipStream_t st) {
  // every argument is checked before anything is enqueued
  if (rows < 0 || cols < 0 || seg_len < 0) return MQ_ERR_ARG;
  if (dtype != MQ_F16 && dtype != MQ_F32) return MQ_ERR_ARG;
  if (abs_err && abs_err_dtype != MQ_F32 && abs_err_dtype != MQ_F16) return MQ_ERR_ARG;
  if (abs_err && abs_err_dtype == MQ_F16 && dtype != MQ_F16) return MQ_ERR_UNSUPPORTED;
  const int64_t n = rows * cols;
  if (seg_sse && (seg_len == 0 || (n && n % seg_len))) return MQ_ERR_GROUP;
  if (n == 0) return MQ_OK;
  if (!a || !b) return MQ_ERR_ARG;
  if (!workspace || workspace_bytes < mq_error_workspace_bytes(rows, cols, col_sse != nullptr)) return MQ_ERR_WORKSPACE;
  const size_t esz = dtype == MQ_F32 ? 4 : 2;
  auto al = [](const void* q, size_t k) { return (reinterpret_cast<uintptr_t>(q) % k) == 0; };
  if (!al(a, esz) || !al(b, esz) || !al(abs_err, abs_err_dtype == MQ_F32 ? 4 : 2) || !al(seg_sse, 8) || !al(col_sse, 8) ||
      !al(totals, 8) || !al(workspace, 8))
    return MQ_ERR_ALIGN;
  ErrParams p = {};
  p.a = a; p.b = b; p.dtype = dtype; p.rows = rows; p.cols = cols; p.n = n;
  p.seg_len = seg_sse ? seg_len : 0;
  p.abs_err = abs_err;
  p.abs_err_mode = !abs_err ? 0 : (abs_err_dtype == MQ_F32 ? 1 : 2);
  p.seg_sse = seg_sse;
  p.partials = static_cast<double*>(workspace);
  const int64_t L = p.seg_len ? p.seg_len : n;
  const bool vec_ok = L % 8 == 0 && n % 8 == 0 && al(a, 16) && al(b, 16) && al(abs_err, 16);
  unsigned grid = 0;
  // column sums over whole, aligned 8-element vectors: one pass gives them AND the totals (the default form of
  // quant_error_multi_weight.py: tensor MSE + per-column list)
  const bool cols_vec = col_sse && cols % 8 == 0 && al(a, 16) && al(b, 16);
  const bool fused_cols = cols_vec && !seg_sse && !abs_err;
  const bool need_main = (totals || seg_sse || abs_err) && !fused_cols;
  if (!need_main) {
    // nothing for the row pass to do
  } else if (vec_ok) {
    int64_t waves;
    if (!p.seg_sse) {
      waves = (n / 8 + 63) / 64;
    } else if (L / 8 >= 64) {
      waves = n / L;
    } else {
      int tlog = 0;
      while ((1 << tlog) < L / 8) ++tlog;
      waves = (n / L + (64 >> tlog) - 1) / (64 >> tlog);
    }
    int64_t blocks = (waves + 3) / 4;
    grid = (unsigned)(blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks));
    hipLaunchKernelGGL(k_err_rows, dim3(grid), dim3(256), 0, st, p);
  } else {
    // one workgroup per segment, or -- without segments -- a flat sweep by enough workgroups to fill the chip
    int64_t blocks = p.seg_sse ? n / L : (n + 255) / 256;
    const int64_t cap = p.seg_sse ? kErrMaxGrid : 2048;
    grid = (unsigned)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
    hipLaunchKernelGGL(k_err_generic, dim3(grid), dim3(256), 0, st, p);
  }
  MQ_HIP(hipGetLastError());
  if (totals && need_main) hipLaunchKernelGGL(k_err_finalize, dim3(1), dim3(256), 0, st, p.partials, (int)grid, totals);
  if (col_sse) {
    p.col_part = static_cast<double*>(workspace) + (size_t)kErrMaxGrid * 2;
    if (cols_vec) {
      // enough (column-vector, row-chunk) threads to fill the chip, at most 64 chunks
      int64_t cvecs = cols / 8;
      int64_t chunks = (256 * 8 * 256 + cvecs - 1) / cvecs;
      chunks = chunks < 1 ? 1 : (chunks > 64 ? 64 : chunks);
      if (chunks > rows) chunks = rows;
      p.row_chunks = (int32_t)chunks;
      const unsigned cgrid = (unsigned)(((cvecs + 63) / 64) * chunks);
      p.cols_totals = (fused_cols && totals && cgrid <= (unsigned)kErrMaxGrid) ? 1 : 0;
      hipLaunchKernelGGL(k_err_cols, dim3(cgrid), dim3(256), 0, st, p);
      if (fused_cols && totals) {
        if (p.cols_totals) {
          hipLaunchKernelGGL(k_err_finalize, dim3(1), dim3(256), 0, st, p.partials, (int)cgrid, totals);
        } else {  // (more workgroups than partial slots: the row pass after all)
          ErrParams q = p;
          q.col_part = nullptr;
          const int64_t blocks = ((n / 8 + 63) / 64 + 3) / 4;
          const unsigned g2 = (unsigned)(blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks));
          hipLaunchKernelGGL(k_err_rows, dim3(g2), dim3(256), 0, st, q);
          hipLaunchKernelGGL(k_err_finalize, dim3(1), dim3(256), 0, st, q.partials, (int)g2, totals);
        }
      }
    } else {  // any column count or alignment: one column per thread
      int64_t chunks = (256 * 8 * 256 + cols - 1) / cols;
      chunks = chunks < 1 ? 1 : (chunks > 64 ? 64 : chunks);
      if (chunks > rows) chunks = rows;
      p.row_chunks = (int32_t)chunks;
      int64_t threads = cols * chunks;
      hipLaunchKernelGGL(k_err_cols_scalar, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, p);
    }
    hipLaunchKernelGGL(k_err_cols_finish, dim3((unsigned)((cols + 63) / 64)), dim3(256), 0, st, p.col_part,
                       (int)p.row_chunks, cols, col_sse);
  }
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}

extern "C" int mq_error_stats(const void* a, const void* b, int32_t dtype, int64_t rows, int64_t cols,
                              int64_t seg_len, void* abs_err, int32_t abs_err_dtype, double* seg_sse,
                              double* col_sse, double* totals, void* workspace, size_t workspace_bytes,
                              void* hip_stream) {
  return error_stats_impl(a, b, dtype, rows, cols, seg_len, abs_err, abs_err_dtype, seg_sse, col_sse, totals, workspace,
                          workspace_bytes, static_cast<hipStream_t>(hip_stream));
}

// ---- the same without a caller workspace and without any allocation ---------------------------------------------
// For callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid hipMalloc): the
// per-workgroup partial sums and the partial column sums live in one device array of this code object, sized for
// at most MQ_ERROR_STATS_STATIC_MAX_COLS columns.  Every GPU holds its own copy of the array for the life of the
// process; its address is looked up once per device (with that device current) and used from then on -- the
// array's name itself is only a host-side handle.  Calls that overlap on one GPU share the array and race.
static const size_t kStaticScratchBytes =
    (size_t)kErrMaxGrid * 2 * sizeof(double) + (size_t)64 * MQ_ERROR_STATS_STATIC_MAX_COLS * sizeof(double);
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
#include <atomic>
#include <mutex>
static const int kMaxDevices = 64;
static std::atomic<void*> scratch_ptr[kMaxDevices];  // per device, null until looked up
static std::mutex scratch_mu;

// This device's copy of the array.  The launches go to the CURRENT device, so the tensors must live there: a
// tensor of another GPU (with its stream) would be paired with this GPU's scratch address.
static int static_scratch(const void* tensor, void** out) {
  int dev = -1;
  MQ_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return MQ_ERR_UNSUPPORTED;
  if (tensor) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, tensor) != hipSuccess) {
      (void)hipGetLastError();  // (not a pointer HIP knows: the launch itself will say so)
    } else if (at.type == hipMemoryTypeDevice && at.device != dev) {
      return MQ_ERR_ARG;
    }
  }
  void* q = scratch_ptr[dev].load(std::memory_order_acquire);
  if (!q) {
    std::lock_guard<std::mutex> lock(scratch_mu);
    q = scratch_ptr[dev].load(std::memory_order_relaxed);
    if (!q) {
      MQ_HIP(hipGetSymbolAddress(&q, HIP_SYMBOL(scratch)));  // the current device's copy of the array
      scratch_ptr[dev].store(q, std::memory_order_release);
    }
  }
  *out = q;
  return MQ_OK;
}

extern "C" int mq_error_stats_static(const void* a, const void* b, int32_t dtype, int64_t rows, int64_t cols,
                                     int64_t seg_len, void* abs_err, int32_t abs_err_dtype, double* seg_sse,
                                     double* col_sse, double* totals, void* hip_stream) {
  if (rows < 0 || cols < 0) return MQ_ERR_ARG;
  if (col_sse && cols > MQ_ERROR_STATS_STATIC_MAX_COLS) return MQ_ERR_TOO_LARGE;
  void* ws = nullptr;
  if (int rc = static_scratch(a, &ws)) return rc;
  // (nothing in the array has to start at zero: every slot the finishing kernels read is written by this call)
  return error_stats_impl(a, b, dtype, rows, cols, seg_len, abs_err, abs_err_dtype, seg_sse, col_sse, totals, ws,
                          kStaticScratchBytes, static_cast<hipStream_t>(hip_stream));
}

extern "C" int mq_bucketize_f16(const void* abs_err_f16, int64_t n, const double* bins_host, int32_t nbins,
                                uint8_t* idx, void* hip_stream) {
  if (n < 0 || nbins < 1 || nbins > 64 || !bins_host) return MQ_ERR_ARG;
  if (n == 0) return MQ_OK;
  if (!abs_err_f16 || !idx) return MQ_ERR_ARG;
  BucketParams p = {};
  p.abs_err = static_cast<const uint16_t*>(abs_err_f16);
  p.n = n;
  p.nbins = nbins;
  for (int i = 0; i < nbins; ++i) p.bins[i] = bins_host[i];
  p.idx = idx;
  int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_bucketize, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0,
                     static_cast<hipStream_t>(hip_stream), p);
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}

extern "C" int mq_compare_formats_f16(const void* w_f16, const void* const* q_f16, int32_t nfmt, int64_t n,
                                      uint16_t threshold_f16_bits, uint8_t* flags, void* min_err_f16,
                                      void* hip_stream) {
  if (n < 0 || nfmt < 1 || nfmt > 7 || !q_f16) return MQ_ERR_ARG;
  if (n == 0) return MQ_OK;
  if (!w_f16 || !flags) return MQ_ERR_ARG;
  CompareParams p = {};
  p.w = static_cast<const uint16_t*>(w_f16);
  for (int i = 0; i < nfmt; ++i) {
    if (!q_f16[i]) return MQ_ERR_ARG;
    p.q[i] = static_cast<const uint16_t*>(q_f16[i]);
  }
  p.nfmt = nfmt;
  p.n = n;
  p.thr_bits = threshold_f16_bits;
  p.flags = flags;
  p.min_err = static_cast<uint16_t*>(min_err_f16);
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  auto al = [](const void* q, size_t a) { return (reinterpret_cast<uintptr_t>(q) % a) == 0; };
  bool vec_ok = n >= 8 && al(w_f16, 16) && al(flags, 8) && al(min_err_f16, 16);
  for (int i = 0; i < nfmt; ++i) vec_ok = vec_ok && al(q_f16[i], 16);
  int64_t done = 0;
  if (vec_ok) {
    switch (nfmt) {
      case 1: launch_compare_vec<1>(p, st); break;
      case 2: launch_compare_vec<2>(p, st); break;
      case 3: launch_compare_vec<3>(p, st); break;
      case 4: launch_compare_vec<4>(p, st); break;
      case 5: launch_compare_vec<5>(p, st); break;
      case 6: launch_compare_vec<6>(p, st); break;
      default: launch_compare_vec<7>(p, st); break;
    }
    MQ_HIP(hipGetLastError());
    done = n / 8 * 8;
  }
  if (done < n) {  // the ragged end, or everything when a pointer is not 16-byte aligned
    p.w += done;
    for (int i = 0; i < nfmt; ++i) p.q[i] += done;
    p.flags += done;
    if (p.min_err) p.min_err += done;
    p.n = n - done;
    int64_t blocks = (p.n + 255) / 256;
    hipLaunchKernelGGL(k_compare, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, st, p);
    MQ_HIP(hipGetLastError());
  }
  return MQ_OK;
}
