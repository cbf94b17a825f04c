// Quantize -> dequantize -> error for value tables given as explicit lists (gfx950).
//
// Covers the reference's formats that the closed-form (E, M, bias) kernels of mq_quant.hip cannot express
// (/root/reference/quant_utils/quant_weight.py): the irregular lists FP3 :11, FLINT4 :20, FLINT5 :36 and the
// *_er_* / *_ea_* variants :12-15, :24-31 in float16 arithmetic (quant_datatype :577-601, _fp_quant_meta
// :829-872), the float32 paths (quant_datatype's per-tensor branch :562-575 and FP8_E5M2 :505-540, :796-826)
// and MX (quant_mx :337-380).  These formats are used for 3- / 4-bit studies on single layers, not for the
// bandwidth-critical sweeps: the layout is the simple one -- a wave (segments up to 2048 elements) or a
// workgroup per segment, one element per lane and step, the table searched in LDS.
//
// Arithmetic (one rounding per reference op; h() = round to float16, everything else float32):
//   F16: v = h(w); s = clamp(h(max|v| / qmax)); x = h(v / s); q = table(x); d = h(q * s)
//   F32: s = clamp(max|w32| / qmax, 1e-5f, 1e4f); x = w32 / s; q = table(x); t = q * s; d = t or h(t)
//   MX : e = floor(log2(max|w32|)); x = (w32 / 2^e) / sc, sc = float32(1 / (qmax / 2)); d = h((q * sc) * 2^e)
//   table(x): entry i with mid[i-1] < x <= mid[i]; NaN -> 0
// Compile with -ffp-contract=off, no fast-math, denormals kept (Makefile).
#include <math.h>

#include "mq_common.h"

struct TableParams {
  const void* w;
  int32_t in_dtype, math, flags, deq_f32, count, need_err, abs_err_f16, pad;
  int64_t n, L, nseg;
  float qmax, mx_scale;
  void* deq;
  void* codes;
  void* scale;
  void* abs_err;
  double* seg_sse;
  double seg_mse_scale;  // 0: float64 sums; 1 / L: seg_sse points to float32 MSEs (mq_quant_out.seg_mse_f32)
  double* partials;  // [gridDim.x][2]
  uint32_t* tred;    // per-tensor scope: max over the |value| bits
  float vals[MQ_TABLE_MAX_VALUES];
  float mids[MQ_TABLE_MAX_VALUES];  // count - 1 midpoints, rounded to the working type
};

struct SegT {
  float s;    // F16: the float16 scale widened; F32: the scale; MX: 2^e
  float aux;  // MX: float32(1 / (qmax / 2))
};

__device__ __forceinline__ float tbl_pick(const float* vals, const float* mids, int n, float x, int flags) {
  if (x != x) return 0.0f;
  if ((flags & MQ_TABLE_NEG_INF_ZERO) && x == -INFINITY) return 0.0f;
  int lo = 0, hi = n - 1;  // first i with x <= mid[i]; n - 1 if none
  while (lo < hi) {
    int m = (lo + hi) >> 1;
    if (x <= mids[m]) hi = m; else lo = m + 1;
  }
  return vals[lo];
}

__device__ __forceinline__ uint32_t tbl_load_f16(const void* w, int dtype, int64_t e) {
  if (dtype == MQ_F16) return static_cast<const uint16_t*>(w)[e];
  if (dtype == MQ_BF16) return f2h_bits(f_from_bits((uint32_t) static_cast<const uint16_t*>(w)[e] << 16));
  return f2h_bits(static_cast<const float*>(w)[e]);
}
__device__ __forceinline__ float tbl_load_f32(const void* w, int dtype, int64_t e) {
  if (dtype == MQ_F16) return (float)h_from_bits(static_cast<const uint16_t*>(w)[e]);
  if (dtype == MQ_BF16) return f_from_bits((uint32_t) static_cast<const uint16_t*>(w)[e] << 16);
  return static_cast<const float*>(w)[e];
}

// |value| bits of element e in the working type: an integer max over them is the float max, and a NaN
// (above the infinity pattern) wins, which is torch.amax's NaN propagation
template <int MATH>
__device__ __forceinline__ uint32_t tbl_abs_bits(const TableParams& p, int64_t e) {
  if (MATH == MQ_TABLE_F16) return tbl_load_f16(p.w, p.in_dtype, e) & 0x7fffu;
  return f_bits(tbl_load_f32(p.w, p.in_dtype, e)) & 0x7fffffffu;
}

template <int MATH>
__device__ __forceinline__ SegT tbl_seg_params(const TableParams& p, uint32_t amax_bits) {
  SegT sp;
  sp.aux = p.mx_scale;
  if (MATH == MQ_TABLE_F16) {
    half_t s = (half_t)((float)h_from_bits(amax_bits) / p.qmax);
    const half_t lo = h_from_bits(0x00A8u), hi = h_from_bits(0x70E2u);  // float16(1e-5), float16(1e4)
    s = s < lo ? lo : s;
    s = s > hi ? hi : s;
    sp.s = (float)s;
  } else if (MATH == MQ_TABLE_F32) {
    float s = f_from_bits(amax_bits) / p.qmax;
    s = s < 1e-5f ? 1e-5f : s;
    s = s > 1e4f ? 1e4f : s;
    sp.s = s;
  } else {
    // 2^floor(log2(amax)) with log2 in float32, as torch computes it (quant_weight.py:361-362): for a float32
    // maximum a step or two under a power of two the float32 logarithm rounds UP to the integer (7.9999995 -> 3.0),
    // so the exponent field of amax is not enough.  float32(log2 in float64) is the correctly rounded float32
    // logarithm, which is what torch's log2 returns next to powers of two (pinned by the pow2edge golden case).
    const float a = f_from_bits(amax_bits);
    if (a == 0.0f) sp.s = 0.0f;                       // log2(0) = -inf, 2^-inf = 0
    else if (amax_bits >= 0x7f800000u) sp.s = a;      // inf stays inf, NaN stays NaN
    else sp.s = ldexpf(1.0f, (int)floorf((float)log2((double)a)));
  }
  return sp;
}

// One element: returns the float32 error float32(h(w)) - float32(deq) and stores the requested outputs.
template <int MATH>
__device__ __forceinline__ float tbl_elem(const TableParams& p, const float* vals, const float* mids, const SegT& sp,
                                          int64_t e) {
  const uint32_t vb = tbl_load_f16(p.w, p.in_dtype, e);
  const float v16 = (float)h_from_bits(vb);
  float d;
  if (MATH == MQ_TABLE_F16) {
    const float x = (float)(half_t)(v16 / sp.s);
    const float q = tbl_pick(vals, mids, p.count, x, p.flags);
    const uint32_t db = f2h_bits(q * sp.s);
    d = (float)h_from_bits(db);
    if (p.deq) static_cast<uint16_t*>(p.deq)[e] = (uint16_t)db;
    if (p.codes) static_cast<uint16_t*>(p.codes)[e] = (uint16_t)f2h_bits(q);
  } else {
    const float w32 = tbl_load_f32(p.w, p.in_dtype, e);
    float q, t;
    if (MATH == MQ_TABLE_F32) {
      q = tbl_pick(vals, mids, p.count, w32 / sp.s, p.flags);
      t = q * sp.s;
      if (p.codes) static_cast<float*>(p.codes)[e] = q;
    } else {
      const float wn = w32 / sp.s;
      q = tbl_pick(vals, mids, p.count, wn / sp.aux, p.flags);
      t = (q * sp.aux) * sp.s;
    }
    if (p.deq_f32) {
      d = t;
      if (p.deq) static_cast<float*>(p.deq)[e] = t;
    } else {
      const uint32_t db = f2h_bits(t);
      d = (float)h_from_bits(db);
      if (p.deq) static_cast<uint16_t*>(p.deq)[e] = (uint16_t)db;
    }
  }
  const float err = v16 - d;
  if (p.abs_err) {
    const float ab = __builtin_fabsf(err);
    if (p.abs_err_f16) static_cast<uint16_t*>(p.abs_err)[e] = (uint16_t)f2h_bits(ab);
    else static_cast<float*>(p.abs_err)[e] = ab;
  }
  return err;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    uint32_t o = __shfl_xor(v, off);
    v = o > v ? o : v;
  }
  return v;
}

__device__ __forceinline__ void tbl_store_scale(const TableParams& p, const SegT& sp, int64_t seg) {
  if (!p.scale) return;
  if (p.math == MQ_TABLE_F16) static_cast<uint16_t*>(p.scale)[seg] = (uint16_t)f2h_bits(sp.s);
  else static_cast<float*>(p.scale)[seg] = sp.s;
}

// per-workgroup partial sums (sum err^2, sum |err|) in a fixed order
__device__ __forceinline__ void tbl_block_partials(const TableParams& p, double sse, double sabs, double* lds /*[32]*/) {
  if (!p.partials) return;
  sse = wave_sum(sse);
  sabs = wave_sum(sabs);
  const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { lds[wv] = sse; lds[16 + wv] = sabs; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int i = 0; i < nw; ++i) { a += lds[i]; b += lds[16 + i]; }
    p.partials[2 * blockIdx.x] = a;
    p.partials[2 * blockIdx.x + 1] = b;
  }
}

// TEAM = 64: one wave per segment (4 per workgroup); TEAM = 256: one workgroup per segment.
template <int MATH, int TEAM>
__global__ __launch_bounds__(256) void k_table_seg(const TableParams p) {
  __shared__ float s_vals[MQ_TABLE_MAX_VALUES], s_mids[MQ_TABLE_MAX_VALUES];
  __shared__ double lds[32];
  __shared__ uint32_t red_u[4];
  __shared__ double red_d[4];
  for (int i = threadIdx.x; i < p.count; i += blockDim.x) { s_vals[i] = p.vals[i]; s_mids[i] = p.mids[i]; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t team = TEAM == 64 ? (int64_t)blockIdx.x * 4 + wv : blockIdx.x;
  const int64_t nteams = TEAM == 64 ? (int64_t)gridDim.x * 4 : gridDim.x;
  const int tid = TEAM == 64 ? lane : (int)threadIdx.x;
  double acc_sse = 0.0, acc_abs = 0.0;
  // (workgroup teams: the trip count depends on blockIdx only, so the barriers below are uniform)
  for (int64_t seg = team; seg < p.nseg; seg += nteams) {
    const int64_t base = seg * p.L;
    uint32_t m = 0;
    for (int64_t i = tid; i < p.L; i += TEAM) {
      uint32_t a = tbl_abs_bits<MATH>(p, base + i);
      m = a > m ? a : m;
    }
    m = wave_max_u32(m);
    if (TEAM == 256) {
      __syncthreads();
      if (lane == 0) red_u[wv] = m;
      __syncthreads();
      for (int i = 0; i < 4; ++i) m = red_u[i] > m ? red_u[i] : m;
    }
    const SegT sp = tbl_seg_params<MATH>(p, m);
    double s = 0.0;
    for (int64_t i = tid; i < p.L; i += TEAM) {
      const float err = tbl_elem<MATH>(p, s_vals, s_mids, sp, base + i);
      if (p.need_err) {
        s += (double)(err * err);
        acc_abs += (double)__builtin_fabsf(err);
      }
    }
    acc_sse += s;
    if (p.seg_sse) {
      s = wave_sum(s);
      if (TEAM == 256) {
        __syncthreads();
        if (lane == 0) red_d[wv] = s;
        __syncthreads();
        s = (red_d[0] + red_d[1]) + (red_d[2] + red_d[3]);
      }
      if (tid == 0) {
        if (p.seg_mse_scale != 0.0) reinterpret_cast<float*>(p.seg_sse)[seg] = (float)(s * p.seg_mse_scale);
        else p.seg_sse[seg] = s;
      }
    }
    if (tid == 0) tbl_store_scale(p, sp, seg);
  }
  tbl_block_partials(p, acc_sse, acc_abs, lds);
}

// per-tensor scope: maximum first (zero-initialised word), then every workgroup quantizes a share
template <int MATH>
__global__ __launch_bounds__(256) void k_table_reduce(const TableParams p) {
  uint32_t m = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t a = tbl_abs_bits<MATH>(p, i);
    m = a > m ? a : m;
  }
  m = wave_max_u32(m);
  if ((threadIdx.x & 63) == 0) atomicMax(p.tred, m);
}

template <int MATH>
__global__ __launch_bounds__(256) void k_table_flat(const TableParams p) {
  __shared__ float s_vals[MQ_TABLE_MAX_VALUES], s_mids[MQ_TABLE_MAX_VALUES];
  __shared__ double lds[32];
  for (int i = threadIdx.x; i < p.count; i += blockDim.x) { s_vals[i] = p.vals[i]; s_mids[i] = p.mids[i]; }
  __syncthreads();
  const SegT sp = tbl_seg_params<MATH>(p, p.tred[0]);
  double acc_sse = 0.0, acc_abs = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * blockDim.x) {
    const float err = tbl_elem<MATH>(p, s_vals, s_mids, sp, i);
    if (p.need_err) {
      acc_sse += (double)(err * err);
      acc_abs += (double)__builtin_fabsf(err);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) tbl_store_scale(p, sp, 0);
  tbl_block_partials(p, acc_sse, acc_abs, lds);
}

__global__ __launch_bounds__(256) void k_table_finalize(const double* partials, int nblocks, double* totals,
                                                        double* seg_sse0, double seg_mse_scale) {
  __shared__ double lds[32];
  double a = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x) { a += partials[2 * i]; c += partials[2 * i + 1]; }
  a = wave_sum(a);
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) { lds[threadIdx.x >> 6] = a; lds[16 + (threadIdx.x >> 6)] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double x = 0.0, y = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { x += lds[i]; y += lds[16 + i]; }
    if (totals) { totals[0] = x; totals[1] = y; }
    if (seg_sse0) {
      if (seg_mse_scale != 0.0) reinterpret_cast<float*>(seg_sse0)[0] = (float)(x * seg_mse_scale);
      else seg_sse0[0] = x;
    }
  }
}

// ================================================================================================
// host side
// ================================================================================================
static const size_t kTblHeader = 256;
static const int64_t kTblMaxGrid = 8192;

extern "C" size_t mq_quantize_table_workspace_bytes(int64_t, int64_t, const mq_table_cfg*) {
  return kTblHeader + (size_t)kTblMaxGrid * 2 * sizeof(double);
}

// float -> float16 -> float on the host (round to nearest even), for the midpoints of the float16 path
static float host_round_f16(float f) { return (float)(_Float16)f; }

template <int MATH>
static int table_launch(const TableParams& p, bool per_tensor, const mq_quant_out* out, hipStream_t st) {
  TableParams q = p;
  int64_t blocks;
  if (per_tensor) {
    blocks = (p.n + 255) / 256;
    if (blocks > kTblMaxGrid) blocks = kTblMaxGrid;
    MQ_HIP(hipMemsetAsync(p.tred, 0, 16, st));
    hipLaunchKernelGGL((k_table_reduce<MATH>), dim3((unsigned)blocks), dim3(256), 0, st, q);
    hipLaunchKernelGGL((k_table_flat<MATH>), dim3((unsigned)blocks), dim3(256), 0, st, q);
  } else if (p.L <= 2048) {
    blocks = (p.nseg + 3) / 4;
    if (blocks > kTblMaxGrid) blocks = kTblMaxGrid;
    hipLaunchKernelGGL((k_table_seg<MATH, 64>), dim3((unsigned)blocks), dim3(256), 0, st, q);
  } else {
    blocks = p.nseg < kTblMaxGrid ? p.nseg : kTblMaxGrid;
    hipLaunchKernelGGL((k_table_seg<MATH, 256>), dim3((unsigned)blocks), dim3(256), 0, st, q);
  }
  MQ_HIP(hipGetLastError());
  if (out->totals || (per_tensor && out->seg_sse)) {
    hipLaunchKernelGGL(k_table_finalize, dim3(1), dim3(256), 0, st, p.partials, (int)blocks, out->totals,
                       per_tensor ? out->seg_sse : nullptr, p.seg_mse_scale);
    MQ_HIP(hipGetLastError());
  }
  return MQ_OK;
}

static bool tbl_aligned(const void* q, size_t a) { return (reinterpret_cast<uintptr_t>(q) % a) == 0; }

extern "C" int mq_quantize_table(const void* w, int32_t w_dtype, int64_t rows, int64_t cols, const mq_table_cfg* cfg,
                                 const mq_quant_out* out, void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!cfg || !out || rows < 0 || cols < 0) return MQ_ERR_ARG;
  if (w_dtype != MQ_F16 && w_dtype != MQ_BF16 && w_dtype != MQ_F32) return MQ_ERR_ARG;
  if (cfg->math != MQ_TABLE_F16 && cfg->math != MQ_TABLE_F32 && cfg->math != MQ_TABLE_MX) return MQ_ERR_ARG;
  if (cfg->count < 2 || cfg->count > MQ_TABLE_MAX_VALUES) return MQ_ERR_ARG;
  if (cfg->deq_dtype != MQ_F16 && !(cfg->deq_dtype == MQ_F32 && cfg->math == MQ_TABLE_F32)) return MQ_ERR_UNSUPPORTED;
  float qmax = 0.f;
  for (int i = 0; i < cfg->count; ++i) {
    const float v = cfg->values[i];
    if (!(v == v) || isinf(v) || (i > 0 && !(v > cfg->values[i - 1]))) return MQ_ERR_ARG;
    // float16 arithmetic needs float16 table values
    if (cfg->math == MQ_TABLE_F16 && host_round_f16(v) != v) return MQ_ERR_UNSUPPORTED;
    qmax = fabsf(v) > qmax ? fabsf(v) : qmax;
  }
  if (!(qmax > 0.f)) return MQ_ERR_ARG;
  if (cfg->math == MQ_TABLE_MX) {
    if (cfg->group_size <= 0) return MQ_ERR_UNSUPPORTED;
    if (out->codes || out->scale) return MQ_ERR_UNSUPPORTED;
  }
  if (out->zero_point) return MQ_ERR_UNSUPPORTED;
  if (cfg->group_size > 0 && cols % cfg->group_size != 0) return MQ_ERR_GROUP;
  const int64_t n = rows * cols;
  if (n == 0) return MQ_OK;
  if (!w) return MQ_ERR_ARG;
  if (out->abs_err && out->abs_err_dtype != MQ_F32 && out->abs_err_dtype != MQ_F16) return MQ_ERR_ARG;
  if (out->seg_mse_f32 == 2) return MQ_ERR_UNSUPPORTED;  // numpy-order entries: mq_quantize's group kernels only
  if (out->seg_mse_f32 != 0 && out->seg_mse_f32 != 1) return MQ_ERR_ARG;
  if (!workspace || workspace_bytes < mq_quantize_table_workspace_bytes(rows, cols, cfg)) return MQ_ERR_WORKSPACE;
  const size_t wide = cfg->math == MQ_TABLE_F16 ? 2 : 4;  // element size of codes / scale
  if (!tbl_aligned(workspace, 16) || !tbl_aligned(w, w_dtype == MQ_F32 ? 4 : 2) ||
      !tbl_aligned(out->deq, cfg->deq_dtype == MQ_F32 ? 4 : 2) || !tbl_aligned(out->codes, wide) ||
      !tbl_aligned(out->scale, wide) || !tbl_aligned(out->abs_err, out->abs_err_dtype == MQ_F16 ? 2 : 4) ||
      !tbl_aligned(out->seg_sse, out->seg_mse_f32 ? 4 : 8) || !tbl_aligned(out->totals, 8))
    return MQ_ERR_ALIGN;

  TableParams p = {};
  p.w = w;
  p.in_dtype = w_dtype;
  p.math = cfg->math;
  p.flags = cfg->flags;
  p.deq_f32 = cfg->deq_dtype == MQ_F32;
  p.count = cfg->count;
  p.need_err = (out->abs_err || out->seg_sse || out->totals) ? 1 : 0;
  p.abs_err_f16 = out->abs_err_dtype == MQ_F16;
  p.n = n;
  p.L = cfg->group_size == -1 ? n : (cfg->group_size <= 0 ? cols : cfg->group_size);
  p.nseg = n / p.L;
  p.qmax = qmax;
  p.mx_scale = (float)(1.0 / ((double)qmax / 2.0));
  p.deq = out->deq;
  p.codes = out->codes;
  p.scale = out->scale;
  p.abs_err = out->abs_err;
  p.seg_sse = out->seg_sse;
  p.seg_mse_scale = (out->seg_sse && out->seg_mse_f32) ? 1.0 / (double)p.L : 0.0;
  p.tred = static_cast<uint32_t*>(workspace);
  p.partials = (out->totals || (cfg->group_size == -1 && out->seg_sse))
                   ? reinterpret_cast<double*>(static_cast<char*>(workspace) + kTblHeader)
                   : nullptr;
  for (int i = 0; i < cfg->count; ++i) p.vals[i] = cfg->values[i];
  for (int i = 0; i + 1 < cfg->count; ++i) {
    // Python computes (a + b) / 2 in float64; the comparison then runs in the tensor's dtype
    const float mid = (float)(((double)cfg->values[i] + (double)cfg->values[i + 1]) / 2.0);
    p.mids[i] = cfg->math == MQ_TABLE_F16 ? host_round_f16(mid) : mid;
  }
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const bool per_tensor = cfg->group_size == -1;
  switch (cfg->math) {
    case MQ_TABLE_F16: return table_launch<MQ_TABLE_F16>(p, per_tensor, out, st);
    case MQ_TABLE_F32: return table_launch<MQ_TABLE_F32>(p, per_tensor, out, st);
    default: return table_launch<MQ_TABLE_MX>(p, per_tensor, out, st);
  }
}

// ================================================================================================
// search_datatype (quant_weight.py:604-649): per segment, the first of several lists with the smallest float16
// mean squared error.  One wave (segments up to 2048 elements) or one workgroup per segment; the segment is
// swept once per list and once more to write the winner (the re-reads hit the cache).
// ================================================================================================
#define MQ_SEARCH_MAX_TABLES 8
#define MQ_SEARCH_MAX_VALUES 32
struct SearchParams {
  const uint16_t* w;
  uint16_t* deq;
  uint8_t* choice;
  int64_t L, nseg;
  int32_t ntables, pad;
  int32_t count[MQ_SEARCH_MAX_TABLES];
  float qmax[MQ_SEARCH_MAX_TABLES];
  float vals[MQ_SEARCH_MAX_TABLES][MQ_SEARCH_MAX_VALUES];
  float mids[MQ_SEARCH_MAX_TABLES][MQ_SEARCH_MAX_VALUES];
};

__device__ __forceinline__ float search_scale(uint32_t amax_bits, float qmax) {
  half_t s = (half_t)((float)h_from_bits(amax_bits) / qmax);
  const half_t lo = h_from_bits(0x00A8u), hi = h_from_bits(0x70E2u);
  s = s < lo ? lo : s;
  s = s > hi ? hi : s;
  return (float)s;
}
// dequantized float16 bits of one element under one list
__device__ __forceinline__ uint32_t search_deq(const float* vals, const float* mids, int n, float v, float s) {
  const float x = (float)(half_t)(v / s);
  return f2h_bits(tbl_pick(vals, mids, n, x, 0) * s);
}

template <int TEAM>
__global__ __launch_bounds__(256) void k_table_search(const SearchParams p) {
  __shared__ float s_vals[MQ_SEARCH_MAX_TABLES][MQ_SEARCH_MAX_VALUES], s_mids[MQ_SEARCH_MAX_TABLES][MQ_SEARCH_MAX_VALUES];
  __shared__ uint32_t red_u[4];
  __shared__ double red_d[4];
  for (int i = threadIdx.x; i < p.ntables * MQ_SEARCH_MAX_VALUES; i += blockDim.x) {
    (&s_vals[0][0])[i] = (&p.vals[0][0])[i];
    (&s_mids[0][0])[i] = (&p.mids[0][0])[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t team = TEAM == 64 ? (int64_t)blockIdx.x * 4 + wv : blockIdx.x;
  const int64_t nteams = TEAM == 64 ? (int64_t)gridDim.x * 4 : gridDim.x;
  const int tid = TEAM == 64 ? lane : (int)threadIdx.x;
  for (int64_t seg = team; seg < p.nseg; seg += nteams) {
    const uint16_t* w = p.w + seg * p.L;
    uint32_t m = 0;
    for (int64_t i = tid; i < p.L; i += TEAM) {
      uint32_t a = w[i] & 0x7fffu;
      m = a > m ? a : m;
    }
    m = wave_max_u32(m);
    if (TEAM == 256) {
      __syncthreads();
      if (lane == 0) red_u[wv] = m;
      __syncthreads();
      for (int i = 0; i < 4; ++i) m = red_u[i] > m ? red_u[i] : m;
    }
    half_t best = (half_t)1000.0f;  // torch.full(..., 1e3, dtype=float16)
    int pick = -1;
    for (int t = 0; t < p.ntables; ++t) {
      const float s = search_scale(m, p.qmax[t]);
      // the float16 squares are multiples of 2^-24 (or inf / NaN): their float64 sum is exact
      double acc = 0.0;
      for (int64_t i = tid; i < p.L; i += TEAM) {
        const half_t v = h_from_bits(w[i]);
        const half_t d = h_from_bits(search_deq(s_vals[t], s_mids[t], p.count[t], (float)v, s));
        const half_t e = d - v;
        acc += (double)(float)(half_t)(e * e);
      }
      acc = wave_sum(acc);
      if (TEAM == 256) {
        __syncthreads();
        if (lane == 0) red_d[wv] = acc;
        __syncthreads();
        acc = (red_d[0] + red_d[1]) + (red_d[2] + red_d[3]);
      }
      // mean of a float16 tensor: float32 sum, divided in float32, rounded to float16
      const half_t mean = (half_t)((float)acc / (float)p.L);
      if (mean < best) { best = mean; pick = t; }  // a NaN mean never wins
    }
    const float s = pick >= 0 ? search_scale(m, p.qmax[pick]) : 0.0f;
    for (int64_t i = tid; i < p.L; i += TEAM) {
      uint32_t db = 0;
      if (pick >= 0) db = search_deq(s_vals[pick], s_mids[pick], p.count[pick], (float)h_from_bits(w[i]), s);
      p.deq[seg * p.L + i] = (uint16_t)db;
    }
    if (p.choice && tid == 0) p.choice[seg] = pick >= 0 ? (uint8_t)pick : (uint8_t)255;
  }
}

extern "C" int mq_search_tables(const void* w_f16, int64_t rows, int64_t cols, int64_t group_size,
                                const mq_table_cfg* tables, int32_t ntables, void* deq, uint8_t* choice,
                                void* hip_stream) {
  if (!tables || ntables < 1 || ntables > MQ_SEARCH_MAX_TABLES || rows < 0 || cols < 0) return MQ_ERR_ARG;
  if (group_size > 0 && cols % group_size != 0) return MQ_ERR_GROUP;
  SearchParams p = {};
  for (int t = 0; t < ntables; ++t) {
    const mq_table_cfg& c = tables[t];
    if (c.count < 2) return MQ_ERR_ARG;
    if (c.count > MQ_SEARCH_MAX_VALUES) return MQ_ERR_UNSUPPORTED;
    float qmax = 0.f;
    for (int i = 0; i < c.count; ++i) {
      const float v = c.values[i];
      if (!(v == v) || isinf(v) || (i > 0 && !(v > c.values[i - 1]))) return MQ_ERR_ARG;
      if (host_round_f16(v) != v) return MQ_ERR_UNSUPPORTED;
      qmax = fabsf(v) > qmax ? fabsf(v) : qmax;
      p.vals[t][i] = v;
    }
    if (!(qmax > 0.f)) return MQ_ERR_ARG;
    for (int i = 0; i + 1 < c.count; ++i)
      p.mids[t][i] = host_round_f16((float)(((double)c.values[i] + (double)c.values[i + 1]) / 2.0));
    p.count[t] = c.count;
    p.qmax[t] = qmax;
  }
  const int64_t n = rows * cols;
  if (n == 0) return MQ_OK;
  if (!w_f16 || !deq) return MQ_ERR_ARG;
  if (!tbl_aligned(w_f16, 2) || !tbl_aligned(deq, 2)) return MQ_ERR_ALIGN;
  p.w = static_cast<const uint16_t*>(w_f16);
  p.deq = static_cast<uint16_t*>(deq);
  p.choice = choice;
  p.L = group_size > 0 ? group_size : cols;
  p.nseg = n / p.L;
  p.ntables = ntables;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  if (p.L <= 2048) {
    int64_t blocks = (p.nseg + 3) / 4;
    if (blocks > kTblMaxGrid) blocks = kTblMaxGrid;
    hipLaunchKernelGGL((k_table_search<64>), dim3((unsigned)blocks), dim3(256), 0, st, p);
  } else {
    int64_t blocks = p.nseg < kTblMaxGrid ? p.nseg : kTblMaxGrid;
    hipLaunchKernelGGL((k_table_search<256>), dim3((unsigned)blocks), dim3(256), 0, st, p);
  }
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}
