// Fused quantize -> dequantize -> error kernels for gfx950 (MI355X).
//
// One pass per weight tensor: each lane loads 8 consecutive elements (16 B of float16), the lanes that
// share a scale ("team": part of a wave, a wave, or a workgroup) reduce absmax or min/max with
// cross-lane shuffles (workgroup teams add one LDS exchange), every lane derives the segment's
// scale / zero-point, quantizes and dequantizes its 8 elements from registers, stores the requested
// outputs with 8/16-byte vector stores and accumulates the float32 error in float64.  The input is
// read from HBM exactly once; the kernel is HBM-bandwidth bound (no contraction, no MFMA).
//
// Arithmetic contract (reference: /root/reference/quant_utils/quant_weight.py :229-335, :503-601,
// :717-872): the reference runs every op as its own float16 torch kernel, i.e. widen to float32,
// compute, round once to float16.  Each h(...) below is one such rounding:
//   v  = h(w)
//   sym : s = clamp(h(max|v| / qmax));  q = clamp(rint(h(v / s)), +-qmax);          d = h(q * s)
//   asym: s = clamp(h(h(rmax - rmin) / qmax));  zp = clamp(rint(h(-rmin / s)), 0, qmax)
//         q = clamp(h(rint(h(v / s)) + zp), 0, qmax);                               d = h(h(q - zp) * s)
//   lut : s = clamp(h(max|v| / LUTmax));  q = nearest table value to h(v / s), ties toward -inf;  d = h(q * s)
// This file must be compiled with -ffp-contract=off and without fast-math.
#include "mq_common.h"

enum { SC_SUBWAVE = 0, SC_WAVE = 1, SC_BLOCK = 2, SC_TENSOR = 3 };

struct LutConst {
  uint32_t shift, half, maxb, minpos, mid;  // rounding constants in the float16 (or float32) bit domain
};

struct QuantParams {
  const void* w;
  int32_t in_dtype;
  int32_t f32_math;
  int64_t n;     // elements
  int64_t L;     // segment length in elements
  int64_t nseg;  // segments
  int32_t team_log2;     // SC_SUBWAVE: log2(lanes per segment)
  int32_t vecs_per_seg;  // L / 8
  float qmax;            // sym: 2^(b-1)-1, asym: 2^b-1, lut: largest table value
  LutConst lut;
  uint16_t* deq;
  void* codes;
  uint16_t* scale;
  uint16_t* zp;
  void* abs_err;
  int32_t abs_err_f16;
  double* seg_sse;
  double* partials;      // [gridDim.x][2]
  const uint32_t* tred;  // SC_TENSOR: reduction words written by k_tensor_reduce
};

struct SegP {
  half_t s;   // scale
  float sf;   // (float)s
  float r;    // RN(1 / sf)
  half_t zp;  // zero-point (asym)
};

// ------------------------------------------------------------------------------------------------
// loads: 8 consecutive elements of the caller's tensor as packed float16 (w.to(torch.float16))
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32x4 load8_f16(const void* w, int dtype, int64_t e) {
  if (dtype == MQ_F16) return *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(w) + e);
  u32x4 out;
  if (dtype == MQ_BF16) {
    u32x4 b = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(w) + e);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      out[k] = f2h_bits(f_from_bits(b[k] << 16)) | (f2h_bits(f_from_bits(b[k] & 0xffff0000u)) << 16);
    return out;
  }
  const f32x4* p = reinterpret_cast<const f32x4*>(static_cast<const float*>(w) + e);
  f32x4 a = p[0], c = p[1];
  out[0] = f2h_bits(a[0]) | (f2h_bits(a[1]) << 16);
  out[1] = f2h_bits(a[2]) | (f2h_bits(a[3]) << 16);
  out[2] = f2h_bits(c[0]) | (f2h_bits(c[1]) << 16);
  out[3] = f2h_bits(c[2]) | (f2h_bits(c[3]) << 16);
  return out;
}

// the same 8 elements widened to float32 without passing through float16 (w.to(torch.float32))
__device__ __forceinline__ void load8_f32(const void* w, int dtype, int64_t e, float* x) {
  if (dtype == MQ_F32) {
    const f32x4* p = reinterpret_cast<const f32x4*>(static_cast<const float*>(w) + e);
    f32x4 a = p[0], c = p[1];
#pragma unroll
    for (int k = 0; k < 4; ++k) { x[k] = a[k]; x[4 + k] = c[k]; }
    return;
  }
  u32x4 b = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(w) + e);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (dtype == MQ_BF16) {
      x[2 * k] = f_from_bits(b[k] << 16);
      x[2 * k + 1] = f_from_bits(b[k] & 0xffff0000u);
    } else {
      x[2 * k] = (float)h_from_bits(b[k] & 0xffffu);
      x[2 * k + 1] = (float)h_from_bits(b[k] >> 16);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// per-lane reductions over one 8-element vector
// ------------------------------------------------------------------------------------------------
// max |v| as an integer max over the sign-cleared float16 bits: order-preserving, and a NaN (> 0x7c00)
// wins, which is torch.amax's NaN propagation.
__device__ __forceinline__ uint32_t absmax8(u32x4 v) {
  u16x2 m = __builtin_bit_cast(u16x2, v[0] & 0x7fff7fffu);
#pragma unroll
  for (int k = 1; k < 4; ++k) m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x2, v[k] & 0x7fff7fffu));
  return m[0] > m[1] ? m[0] : m[1];
}
// Packed float16 min / max.  Written as instructions: hipcc (ROCm 7.2) drops all but the first
// __builtin_elementwise_min/max of a chain over _Float16 vectors.
__device__ __forceinline__ uint32_t pk_min_f16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_min_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t pk_max_f16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void minmax8(u32x4 v, half_t& mn, half_t& mx) {
  uint32_t lo = pk_min_f16(pk_min_f16(v[0], v[1]), pk_min_f16(v[2], v[3]));
  uint32_t hi = pk_max_f16(pk_max_f16(v[0], v[1]), pk_max_f16(v[2], v[3]));
  half_t l0 = h_from_bits(lo & 0xffffu), l1 = h_from_bits(lo >> 16);
  half_t h0 = h_from_bits(hi & 0xffffu), h1 = h_from_bits(hi >> 16);
  mn = l0 < l1 ? l0 : l1;
  mx = h0 > h1 ? h0 : h1;
}

template <int T>
__device__ __forceinline__ uint32_t team_max_u32(uint32_t m) {
#pragma unroll
  for (int off = 1; off < T; off <<= 1) {
    uint32_t o = __shfl_xor(m, off);
    m = o > m ? o : m;
  }
  return m;
}
// (min, max) packed as float16 bits in one word so that each step costs one shuffle
template <int T>
__device__ __forceinline__ void team_minmax(half_t& mn, half_t& mx) {
#pragma unroll
  for (int off = 1; off < T; off <<= 1) {
    uint32_t o = __shfl_xor(h_bits(mn) | (h_bits(mx) << 16), off);
    half_t omn = h_from_bits(o & 0xffffu), omx = h_from_bits(o >> 16);
    mn = omn < mn ? omn : mn;
    mx = omx > mx ? omx : mx;
  }
}
template <int T>
__device__ __forceinline__ double team_sum(double v) {
#pragma unroll
  for (int off = 1; off < T; off <<= 1) v += __shfl_xor(v, off);
  return v;
}

// ------------------------------------------------------------------------------------------------
// segment parameters
// ------------------------------------------------------------------------------------------------
// torch.clamp(min=1e-5, max=1e4) on a float16 tensor: the bounds become float16 0x00A8 / 0x70E2.
__device__ __forceinline__ half_t clamp_scale(half_t s) {
  const half_t lo = h_from_bits(0x00A8u), hi = h_from_bits(0x70E2u);
  s = s < lo ? lo : s;
  s = s > hi ? hi : s;
  return s;  // NaN stays NaN
}
__device__ __forceinline__ SegP finish_params(half_t s) {
  SegP sp;
  sp.s = s;
  sp.sf = (float)s;
  sp.r = 1.0f / sp.sf;  // correctly rounded (IEEE) reciprocal: required by div_h()
  sp.zp = (half_t)0;
  return sp;
}
__device__ __forceinline__ SegP params_absmax(uint32_t amax_bits, float qmax) {
  return finish_params(clamp_scale((half_t)((float)h_from_bits(amax_bits) / qmax)));
}
__device__ __forceinline__ SegP params_minmax(half_t rmin, half_t rmax, float qmax) {
  half_t range = rmax - rmin;
  SegP sp = finish_params(clamp_scale((half_t)((float)range / qmax)));
  half_t z = __builtin_rintf16((half_t)((float)(-rmin) / sp.sf));
  z = z < (half_t)0 ? (half_t)0 : z;
  z = z > (half_t)qmax ? (half_t)qmax : z;
  sp.zp = z;
  return sp;
}

// h(v / s).  q1 below equals the IEEE float32 quotient v/s for every finite float16 v and every float16
// s in [0x00A8, 0x70E2] (the clamped scale range) -- checked exhaustively on the CPU, see
// tests/test_division_shortcut.py -- so rounding it to float16 is h(v / s).
__device__ __forceinline__ half_t div_h(float vf, const SegP& sp) {
  float q0 = vf * sp.r;
  float q1 = __builtin_fmaf(__builtin_fmaf(-q0, sp.sf, vf), sp.r, q0);
  // +-0 and +-inf quotients are already exact in q0; the correction step would turn -0 into +0 and
  // inf into NaN (class mask: -inf | -0 | +0 | +inf)
  q1 = __builtin_amdgcn_classf(q0, 0x264) ? q0 : q1;
  return (half_t)q1;
}

// Nearest value of an (E, M, bias) table for x given as float16 bits: round the 10-bit mantissa to M
// bits (ties toward -inf: down in magnitude for x > 0, up for x < 0), saturate at the largest entry,
// and below the smallest positive entry choose between it and 0 at their midpoint.  NaN -> +0.
__device__ __forceinline__ uint32_t lut_round_bits(uint32_t xb, uint32_t signbit, uint32_t nanlimit, const LutConst& c) {
  uint32_t a = xb & (signbit - 1u);
  bool neg = (xb & signbit) != 0;
  uint32_t r = ((a + (neg ? c.half : c.half - 1u)) >> c.shift) << c.shift;
  r = r > c.maxb ? c.maxb : r;
  if (r < c.minpos) r = (neg ? a >= c.mid : a > c.mid) ? c.minpos : 0u;
  if (a > nanlimit) r = 0u;
  return r ? (r | (neg ? signbit : 0u)) : 0u;
}

// ------------------------------------------------------------------------------------------------
// one lane's 8 elements: quantize, dequantize, store, accumulate the error
// ------------------------------------------------------------------------------------------------
template <int FMT>
__device__ __forceinline__ double process8(const QuantParams& p, u32x4 raw, const SegP& sp, int64_t e0,
                                           double& acc_abs) {
  const half_t qmax_h = (half_t)p.qmax;
  u32x4 dq;          // 8 dequantized float16
  u32x4 c16;         // LUT codes (float16 values)
  uint32_t c8[2];    // INT codes (bytes)
  float ae[8];
  c8[0] = c8[1] = 0;
  double sse = 0.0, sabs = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint32_t vb = (i & 1) ? (raw[i >> 1] >> 16) : (raw[i >> 1] & 0xffffu);
    float vf = (float)h_from_bits(vb);
    half_t x = div_h(vf, sp);
    half_t d;
    uint32_t code;
    if constexpr (FMT == MQ_INT_SYM) {
      half_t q = __builtin_rintf16(x);
      q = q < -qmax_h ? -qmax_h : q;
      q = q > qmax_h ? qmax_h : q;
      d = q * sp.s;
      code = (uint32_t)(int)q & 0xffu;
    } else if constexpr (FMT == MQ_INT_ASYM) {
      half_t q = __builtin_rintf16(x) + sp.zp;
      q = q < (half_t)0 ? (half_t)0 : q;
      q = q > qmax_h ? qmax_h : q;
      d = (q - sp.zp) * sp.s;
      code = (uint32_t)(int)q & 0xffu;
    } else {
      code = lut_round_bits(h_bits(x), 0x8000u, 0x7c00u, p.lut);
      d = h_from_bits(code) * sp.s;
    }
    uint32_t db = h_bits(d);
    if (i & 1) dq[i >> 1] |= db << 16; else dq[i >> 1] = db;
    if constexpr (FMT == MQ_LUT_FP) {
      if (i & 1) c16[i >> 1] |= code << 16; else c16[i >> 1] = code;
    } else {
      c8[i >> 2] |= code << (8 * (i & 3));
    }
    float err = vf - (float)d;
    float sq = err * err;
    ae[i] = __builtin_fabsf(err);
    sse += (double)sq;
    sabs += (double)ae[i];
  }
  if (p.deq) *reinterpret_cast<u32x4*>(p.deq + e0) = dq;
  if (p.codes) {
    if constexpr (FMT == MQ_LUT_FP) {
      *reinterpret_cast<u32x4*>(static_cast<uint16_t*>(p.codes) + e0) = c16;
    } else {
      u32x2 c = {c8[0], c8[1]};
      *reinterpret_cast<u32x2*>(static_cast<uint8_t*>(p.codes) + e0) = c;
    }
  }
  if (p.abs_err) {
    if (p.abs_err_f16) {
      u32x4 a;
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] = f2h_bits(ae[2 * k]) | (f2h_bits(ae[2 * k + 1]) << 16);
      *reinterpret_cast<u32x4*>(static_cast<uint16_t*>(p.abs_err) + e0) = a;
    } else {
      f32x4* o = reinterpret_cast<f32x4*>(static_cast<float*>(p.abs_err) + e0);
      f32x4 a0 = {ae[0], ae[1], ae[2], ae[3]}, a1 = {ae[4], ae[5], ae[6], ae[7]};
      o[0] = a0;
      o[1] = a1;
    }
  }
  acc_abs += sabs;
  return sse;
}

// quant_datatype's per-tensor float32 branch (quant_weight.py:562-575): x = w32 / s in float32, table
// lookup in float32, (q * s) rounded to float32 and then to float16.
__device__ __forceinline__ double process8_lut_f32(const QuantParams& p, const float* w32, u32x4 raw, float s,
                                                   int64_t e0, double& acc_abs) {
  u32x4 dq, c16;
  float ae[8];
  double sse = 0.0, sabs = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float x = w32[i] / s;
    float q = f_from_bits(lut_round_bits(f_bits(x), 0x80000000u, 0x7f800000u, p.lut));
    float t = q * s;
    uint32_t db = f2h_bits(t), cb = f2h_bits(q);
    if (i & 1) { dq[i >> 1] |= db << 16; c16[i >> 1] |= cb << 16; } else { dq[i >> 1] = db; c16[i >> 1] = cb; }
    uint32_t vb = (i & 1) ? (raw[i >> 1] >> 16) : (raw[i >> 1] & 0xffffu);
    float err = (float)h_from_bits(vb) - (float)h_from_bits(db);
    float sq = err * err;
    ae[i] = __builtin_fabsf(err);
    sse += (double)sq;
    sabs += (double)ae[i];
  }
  if (p.deq) *reinterpret_cast<u32x4*>(p.deq + e0) = dq;
  if (p.codes) *reinterpret_cast<u32x4*>(static_cast<uint16_t*>(p.codes) + e0) = c16;
  if (p.abs_err) {
    if (p.abs_err_f16) {
      u32x4 a;
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] = f2h_bits(ae[2 * k]) | (f2h_bits(ae[2 * k + 1]) << 16);
      *reinterpret_cast<u32x4*>(static_cast<uint16_t*>(p.abs_err) + e0) = a;
    } else {
      f32x4* o = reinterpret_cast<f32x4*>(static_cast<float*>(p.abs_err) + e0);
      f32x4 a0 = {ae[0], ae[1], ae[2], ae[3]}, a1 = {ae[4], ae[5], ae[6], ae[7]};
      o[0] = a0;
      o[1] = a1;
    }
  }
  acc_abs += sabs;
  return sse;
}

// Workgroup partial sums -> partials[blockIdx][0..1], in a fixed order (bitwise reproducible).
__device__ __forceinline__ void block_partials(const QuantParams& p, double sse, double sabs, double* lds /*[32]*/) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  sse = wave_sum(sse);
  sabs = wave_sum(sabs);
  __syncthreads();
  if (lane == 0) { lds[wv] = sse; lds[16 + wv] = sabs; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, b = 0;
    for (int i = 0; i < nw; ++i) { a += lds[i]; b += lds[16 + i]; }
    p.partials[2 * (int64_t)blockIdx.x] = a;
    p.partials[2 * (int64_t)blockIdx.x + 1] = b;
  }
}

// ------------------------------------------------------------------------------------------------
// SC_SUBWAVE: a segment is T <= 64 lanes of one wave; NV independent segments per lane in flight.
// ------------------------------------------------------------------------------------------------
template <int FMT, int T, int NV>
__device__ __forceinline__ void subwave_body(const QuantParams& p, double& acc_sse, double& acc_abs) {
  const int lane = threadIdx.x & 63;
  const int tl = lane & (T - 1), team = lane / T;
  constexpr int SPW = 64 / T;  // segments per wave per vector slot
  const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int64_t tiles = (p.nseg + (int64_t)SPW * NV - 1) / ((int64_t)SPW * NV);
  const bool lane_in_seg = tl < p.vecs_per_seg;
  for (int64_t tile = gw; tile < tiles; tile += nwaves) {
    u32x4 raw[NV];
    bool valid[NV];
    int64_t seg[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      seg[j] = (tile * NV + j) * SPW + team;
      valid[j] = lane_in_seg && seg[j] < p.nseg;
      raw[j] = valid[j] ? load8_f16(p.w, p.in_dtype, seg[j] * p.L + (int64_t)tl * 8) : u32x4{0, 0, 0, 0};
    }
    SegP sp[NV];
    if constexpr (FMT == MQ_INT_ASYM) {
      half_t mn[NV], mx[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        minmax8(raw[j], mn[j], mx[j]);
        if (!valid[j]) { mn[j] = h_from_bits(0x7c00u); mx[j] = h_from_bits(0xfc00u); }
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) team_minmax<T>(mn[j], mx[j]);
#pragma unroll
      for (int j = 0; j < NV; ++j) sp[j] = params_minmax(mn[j], mx[j], p.qmax);
    } else {
      uint32_t m[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) m[j] = absmax8(raw[j]);
#pragma unroll
      for (int j = 0; j < NV; ++j) m[j] = team_max_u32<T>(m[j]);
#pragma unroll
      for (int j = 0; j < NV; ++j) sp[j] = params_absmax(m[j], p.qmax);
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      double s = 0.0;
      if (valid[j]) s = process8<FMT>(p, raw[j], sp[j], seg[j] * p.L + (int64_t)tl * 8, acc_abs);
      acc_sse += s;
      if (p.seg_sse) {
        s = team_sum<T>(s);
        if (valid[j] && tl == 0) p.seg_sse[seg[j]] = s;
      }
      if (valid[j] && tl == 0) {
        if (p.scale) p.scale[seg[j]] = (uint16_t)h_bits(sp[j].s);
        if (FMT == MQ_INT_ASYM && p.zp) p.zp[seg[j]] = (uint16_t)h_bits(sp[j].zp);
      }
    }
  }
}

template <int FMT, int NV>
__global__ __launch_bounds__(256) void k_quant_subwave(QuantParams p) {
  __shared__ double lds[32];
  double acc_sse = 0.0, acc_abs = 0.0;
  switch (p.team_log2) {
    case 0: subwave_body<FMT, 1, NV>(p, acc_sse, acc_abs); break;
    case 1: subwave_body<FMT, 2, NV>(p, acc_sse, acc_abs); break;
    case 2: subwave_body<FMT, 4, NV>(p, acc_sse, acc_abs); break;
    case 3: subwave_body<FMT, 8, NV>(p, acc_sse, acc_abs); break;
    case 4: subwave_body<FMT, 16, NV>(p, acc_sse, acc_abs); break;
    case 5: subwave_body<FMT, 32, NV>(p, acc_sse, acc_abs); break;
    default: subwave_body<FMT, 64, NV>(p, acc_sse, acc_abs); break;
  }
  block_partials(p, acc_sse, acc_abs, lds);
}

// ------------------------------------------------------------------------------------------------
// SC_WAVE / SC_BLOCK: a segment (a row, or a long group) is held in the registers of one wave or one
// workgroup, NV vectors per lane, interleaved so that every load instruction is 1 KiB contiguous.
// ------------------------------------------------------------------------------------------------
template <int FMT, int NV, bool BLOCK>
__device__ __forceinline__ void row_body(const QuantParams& p, double* lds, double& acc_sse, double& acc_abs) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int64_t first = BLOCK ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * nw + wv;
  const int64_t stride = BLOCK ? (int64_t)gridDim.x : (int64_t)gridDim.x * nw;
  const int tid = BLOCK ? (int)threadIdx.x : lane;
  const int tcount = BLOCK ? (int)blockDim.x : 64;
  uint32_t* lds_u = reinterpret_cast<uint32_t*>(lds + 32);  // [2][16] reduction exchange (BLOCK only)
  int parity = 0;
  for (int64_t seg = first; seg < p.nseg; seg += stride) {
    const int64_t base = seg * p.L;
    u32x4 raw[NV];
    bool valid[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      int v = j * tcount + tid;
      valid[j] = v < p.vecs_per_seg;
      raw[j] = valid[j] ? load8_f16(p.w, p.in_dtype, base + (int64_t)v * 8) : u32x4{0, 0, 0, 0};
    }
    SegP sp;
    if constexpr (FMT == MQ_INT_ASYM) {
      half_t mn = h_from_bits(0x7c00u), mx = h_from_bits(0xfc00u);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        half_t a, b;
        minmax8(raw[j], a, b);
        if (valid[j]) { mn = a < mn ? a : mn; mx = b > mx ? b : mx; }
      }
      team_minmax<64>(mn, mx);
      if constexpr (BLOCK) {
        if (lane == 0) lds_u[parity * 16 + wv] = h_bits(mn) | (h_bits(mx) << 16);
        __syncthreads();
        for (int i = 0; i < nw; ++i) {
          uint32_t o = lds_u[parity * 16 + i];
          half_t omn = h_from_bits(o & 0xffffu), omx = h_from_bits(o >> 16);
          mn = omn < mn ? omn : mn;
          mx = omx > mx ? omx : mx;
        }
      }
      sp = params_minmax(mn, mx, p.qmax);
    } else {
      uint32_t m = 0;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        uint32_t a = absmax8(raw[j]);
        m = a > m ? a : m;
      }
      m = team_max_u32<64>(m);
      if constexpr (BLOCK) {
        if (lane == 0) lds_u[parity * 16 + wv] = m;
        __syncthreads();
        for (int i = 0; i < nw; ++i) {
          uint32_t o = lds_u[parity * 16 + i];
          m = o > m ? o : m;
        }
      }
      sp = params_absmax(m, p.qmax);
    }
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < NV; ++j)
      if (valid[j]) s += process8<FMT>(p, raw[j], sp, base + (int64_t)(j * tcount + tid) * 8, acc_abs);
    acc_sse += s;
    if (p.seg_sse) {
      s = wave_sum(s);
      if constexpr (BLOCK) {
        double* ex = lds + 64 + parity * 16;  // [2][16]
        if (lane == 0) ex[wv] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
          double t = 0.0;
          for (int i = 0; i < nw; ++i) t += ex[i];
          p.seg_sse[seg] = t;
        }
      } else if (lane == 0) {
        p.seg_sse[seg] = s;
      }
    }
    if (tid == 0) {
      if (p.scale) p.scale[seg] = (uint16_t)h_bits(sp.s);
      if (FMT == MQ_INT_ASYM && p.zp) p.zp[seg] = (uint16_t)h_bits(sp.zp);
    }
    parity ^= 1;
  }
}

template <int FMT, int NV>
__global__ __launch_bounds__(256) void k_quant_wave(QuantParams p) {
  __shared__ double lds[32];
  double acc_sse = 0.0, acc_abs = 0.0;
  row_body<FMT, NV, false>(p, lds, acc_sse, acc_abs);
  block_partials(p, acc_sse, acc_abs, lds);
}

template <int FMT, int NV, int THREADS>
__global__ __launch_bounds__(THREADS) void k_quant_block(QuantParams p) {
  __shared__ double lds[96];
  double acc_sse = 0.0, acc_abs = 0.0;
  row_body<FMT, NV, true>(p, lds, acc_sse, acc_abs);
  block_partials(p, acc_sse, acc_abs, lds);
}

// ------------------------------------------------------------------------------------------------
// SC_TENSOR: group_size == -1.  k_tensor_reduce leaves the tensor's absmax / min / max in tred[];
// k_quant_tensor then treats the flat tensor as one segment whose parameters every lane derives itself.
//   tred[0] max over |float16 bits|  (or |float32 bits| when f32_math)
//   tred[1] max over key(v),  tred[2] max over ~key(v)   with key() the order-preserving map of
//           signed float16 bits onto unsigned integers, so that a zero-initialised word works for both.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t h_key(uint32_t b) { return (b & 0x8000u) ? (~b & 0xffffu) : (b | 0x8000u); }
__device__ __forceinline__ uint32_t h_unkey(uint32_t k) { return (k & 0x8000u) ? (k & 0x7fffu) : (~k & 0xffffu); }

__global__ __launch_bounds__(256) void k_tensor_reduce(QuantParams p, uint32_t* tred) {
  const int64_t nvec = p.n / 8;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
  uint32_t amax = 0, kmax = 0, kmin = 0xffffu;
  for (int64_t v = t0; v < nvec; v += step) {
    if (p.f32_math) {
      float x[8];
      load8_f32(p.w, p.in_dtype, v * 8, x);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        uint32_t a = f_bits(x[i]) & 0x7fffffffu;
        amax = a > amax ? a : amax;
      }
    } else {
      u32x4 raw = load8_f16(p.w, p.in_dtype, v * 8);
      uint32_t a = absmax8(raw);
      amax = a > amax ? a : amax;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        uint32_t k = h_key((i & 1) ? (raw[i >> 1] >> 16) : (raw[i >> 1] & 0xffffu));
        kmax = k > kmax ? k : kmax;
        kmin = k < kmin ? k : kmin;
      }
    }
  }
  amax = team_max_u32<64>(amax);
  kmax = team_max_u32<64>(kmax);
  uint32_t nkmin = team_max_u32<64>(0xffffu - kmin);
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&tred[0], amax);
    atomicMax(&tred[1], kmax);
    atomicMax(&tred[2], nkmin);
  }
}

template <int FMT, int NV>
__global__ __launch_bounds__(256) void k_quant_tensor(QuantParams p) {
  __shared__ double lds[32];
  double acc_sse = 0.0, acc_abs = 0.0;
  const int lane = threadIdx.x & 63;
  const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int64_t nvec = p.n / 8;
  const int64_t tiles = (nvec + 64 * NV - 1) / (64 * NV);
  SegP sp;
  float s32 = 0.f;
  if (FMT == MQ_LUT_FP && p.f32_math) {
    // scale = (rmax / qmax).clamp(1e-5, 1e4), all float32
    float s = f_from_bits(p.tred[0]) / p.qmax;
    s = s < 1e-5f ? 1e-5f : s;
    s = s > 1e4f ? 1e4f : s;
    s32 = s;
    sp = finish_params((half_t)s);
  } else if constexpr (FMT == MQ_INT_ASYM) {
    sp = params_minmax(h_from_bits(h_unkey(0xffffu - p.tred[2])), h_from_bits(h_unkey(p.tred[1])), p.qmax);
  } else {
    sp = params_absmax(p.tred[0], p.qmax);
  }
  for (int64_t tile = gw; tile < tiles; tile += nwaves) {
    u32x4 raw[NV];
    bool valid[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      int64_t v = (tile * NV + j) * 64 + lane;
      valid[j] = v < nvec;
      raw[j] = valid[j] ? load8_f16(p.w, p.in_dtype, v * 8) : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      if (!valid[j]) continue;
      int64_t e0 = ((tile * NV + j) * 64 + lane) * 8;
      if (FMT == MQ_LUT_FP && p.f32_math) {
        float w32[8];
        load8_f32(p.w, p.in_dtype, e0, w32);
        acc_sse += process8_lut_f32(p, w32, raw[j], s32, e0, acc_abs);
      } else {
        acc_sse += process8<FMT>(p, raw[j], sp, e0, acc_abs);
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (p.scale) p.scale[0] = (uint16_t)h_bits(sp.s);
    if (FMT == MQ_INT_ASYM && p.zp) p.zp[0] = (uint16_t)h_bits(sp.zp);
  }
  block_partials(p, acc_sse, acc_abs, lds);
}

// ------------------------------------------------------------------------------------------------
// Generic fallback: any segment length / alignment, one workgroup per segment, scalar accesses, two
// sweeps (the second one hits L2).  Correct for every shape the reference accepts; not a fast path.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t load1_f16(const void* w, int dtype, int64_t e) {
  if (dtype == MQ_F16) return static_cast<const uint16_t*>(w)[e];
  if (dtype == MQ_BF16) return f2h_bits(f_from_bits((uint32_t) static_cast<const uint16_t*>(w)[e] << 16));
  return f2h_bits(static_cast<const float*>(w)[e]);
}

template <int FMT>
__global__ __launch_bounds__(256) void k_quant_generic(QuantParams p) {
  __shared__ double lds[96];
  uint32_t* lds_u = reinterpret_cast<uint32_t*>(lds + 32);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const half_t qmax_h = (half_t)p.qmax;
  double acc_sse = 0.0, acc_abs = 0.0;
  int parity = 0;
  for (int64_t seg = blockIdx.x; seg < p.nseg; seg += gridDim.x) {
    const int64_t base = seg * p.L;
    uint32_t m = 0;
    half_t mn = h_from_bits(0x7c00u), mx = h_from_bits(0xfc00u);
    for (int64_t i = threadIdx.x; i < p.L; i += blockDim.x) {
      uint32_t vb = load1_f16(p.w, p.in_dtype, base + i);
      uint32_t a = vb & 0x7fffu;
      m = a > m ? a : m;
      half_t v = h_from_bits(vb);
      mn = v < mn ? v : mn;
      mx = v > mx ? v : mx;
    }
    m = team_max_u32<64>(m);
    team_minmax<64>(mn, mx);
    if (lane == 0) { lds_u[parity * 32 + wv] = m; lds_u[parity * 32 + 16 + wv] = h_bits(mn) | (h_bits(mx) << 16); }
    __syncthreads();
    for (int i = 0; i < nw; ++i) {
      uint32_t o = lds_u[parity * 32 + i], o2 = lds_u[parity * 32 + 16 + i];
      m = o > m ? o : m;
      half_t omn = h_from_bits(o2 & 0xffffu), omx = h_from_bits(o2 >> 16);
      mn = omn < mn ? omn : mn;
      mx = omx > mx ? omx : mx;
    }
    SegP sp = (FMT == MQ_INT_ASYM) ? params_minmax(mn, mx, p.qmax) : params_absmax(m, p.qmax);
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < p.L; i += blockDim.x) {
      uint32_t vb = load1_f16(p.w, p.in_dtype, base + i);
      float vf = (float)h_from_bits(vb);
      half_t x = div_h(vf, sp), d;
      uint32_t code;
      if constexpr (FMT == MQ_INT_SYM) {
        half_t q = __builtin_rintf16(x);
        q = q < -qmax_h ? -qmax_h : q;
        q = q > qmax_h ? qmax_h : q;
        d = q * sp.s;
        code = (uint32_t)(int)q & 0xffu;
      } else if constexpr (FMT == MQ_INT_ASYM) {
        half_t q = __builtin_rintf16(x) + sp.zp;
        q = q < (half_t)0 ? (half_t)0 : q;
        q = q > qmax_h ? qmax_h : q;
        d = (q - sp.zp) * sp.s;
        code = (uint32_t)(int)q & 0xffu;
      } else {
        code = lut_round_bits(h_bits(x), 0x8000u, 0x7c00u, p.lut);
        d = h_from_bits(code) * sp.s;
      }
      if (p.deq) p.deq[base + i] = (uint16_t)h_bits(d);
      if (p.codes) {
        if (FMT == MQ_LUT_FP) static_cast<uint16_t*>(p.codes)[base + i] = (uint16_t)code;
        else static_cast<uint8_t*>(p.codes)[base + i] = (uint8_t)code;
      }
      float err = vf - (float)d;
      float sq = err * err, ab = __builtin_fabsf(err);
      if (p.abs_err) {
        if (p.abs_err_f16) static_cast<uint16_t*>(p.abs_err)[base + i] = (uint16_t)f2h_bits(ab);
        else static_cast<float*>(p.abs_err)[base + i] = ab;
      }
      s += (double)sq;
      acc_abs += (double)ab;
    }
    acc_sse += s;
    if (p.seg_sse) {
      s = wave_sum(s);
      double* ex = lds + 64 + parity * 16;
      if (lane == 0) ex[wv] = s;
      __syncthreads();
      if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < nw; ++i) t += ex[i];
        p.seg_sse[seg] = t;
      }
    }
    if (threadIdx.x == 0) {
      if (p.scale) p.scale[seg] = (uint16_t)h_bits(sp.s);
      if (FMT == MQ_INT_ASYM && p.zp) p.zp[seg] = (uint16_t)h_bits(sp.zp);
    }
    parity ^= 1;
  }
  block_partials(p, acc_sse, acc_abs, lds);
}

// Sum of the per-workgroup partials in a fixed order -> totals[0..1] (and seg_sse[0] for per-tensor).
__global__ __launch_bounds__(256) void k_finalize_totals(const double* partials, int nblocks, double* totals,
                                                         double* seg_sse0) {
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
    if (totals) { totals[0] = x; totals[1] = y; }
    if (seg_sse0) seg_sse0[0] = x;
  }
}

// ================================================================================================
// host side
// ================================================================================================
static int ilog2_ceil(int64_t x) {
  int l = 0;
  while (((int64_t)1 << l) < x) ++l;
  return l;
}

static bool lut_consts(const mq_quant_cfg* cfg, bool f32, LutConst* c, float* lutmax) {
  const int E = cfg->exp_bits, M = cfg->man_bits, B = cfg->exp_bias;
  if (E < 1 || E > 5 || M < 1 || M > 9) return false;
  const int emax = (1 << E) - 1 - B;  // largest exponent, smallest is -B
  const int mant = f32 ? 23 : 10, ebias = f32 ? 127 : 15;
  // every table value must be a normal number of the working float type
  if (-B - 1 + ebias < 1 || emax + ebias > (f32 ? 254 : 30)) return false;
  c->shift = (uint32_t)(mant - M);
  c->half = 1u << (c->shift - 1);
  c->maxb = ((uint32_t)(emax + ebias) << mant) | (((1u << M) - 1u) << c->shift);
  c->minpos = ((uint32_t)(-B + ebias) << mant) | (1u << c->shift);
  c->mid = ((uint32_t)(-B - 1 + ebias) << mant) | (1u << c->shift);  // minpos / 2
  *lutmax = ldexpf(2.0f - ldexpf(1.0f, -M), emax);
  return true;
}

template <int FMT>
static void launch_fmt(int scope, int nv, int threads, dim3 grid, hipStream_t st, const QuantParams& p) {
  if (scope == SC_SUBWAVE) {
    hipLaunchKernelGGL((k_quant_subwave<FMT, 4>), grid, dim3(256), 0, st, p);
  } else if (scope == SC_TENSOR) {
    hipLaunchKernelGGL((k_quant_tensor<FMT, 4>), grid, dim3(256), 0, st, p);
  } else if (scope == SC_WAVE) {
    switch (nv) {
      case 2: hipLaunchKernelGGL((k_quant_wave<FMT, 2>), grid, dim3(256), 0, st, p); break;
      case 4: hipLaunchKernelGGL((k_quant_wave<FMT, 4>), grid, dim3(256), 0, st, p); break;
      case 8: hipLaunchKernelGGL((k_quant_wave<FMT, 8>), grid, dim3(256), 0, st, p); break;
      default: hipLaunchKernelGGL((k_quant_wave<FMT, 16>), grid, dim3(256), 0, st, p); break;
    }
  } else if (scope == SC_BLOCK) {
    if (threads == 256) {
      if (nv == 8) hipLaunchKernelGGL((k_quant_block<FMT, 8, 256>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((k_quant_block<FMT, 16, 256>), grid, dim3(256), 0, st, p);
    } else {
      if (nv == 8) hipLaunchKernelGGL((k_quant_block<FMT, 8, 1024>), grid, dim3(1024), 0, st, p);
      else hipLaunchKernelGGL((k_quant_block<FMT, 16, 1024>), grid, dim3(1024), 0, st, p);
    }
  } else {
    hipLaunchKernelGGL((k_quant_generic<FMT>), grid, dim3(256), 0, st, p);
  }
}

static const size_t kWsHeader = 256;  // reduction words, kept apart from the partials

extern "C" size_t mq_quantize_workspace_bytes(int64_t, int64_t, const mq_quant_cfg*) {
  return kWsHeader + (size_t)MQ_MAX_GRID * 2 * sizeof(double);
}

extern "C" int mq_quantize(const void* w, int32_t w_dtype, int64_t rows, int64_t cols, const mq_quant_cfg* cfg,
                           const mq_quant_out* out, void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!cfg || !out || rows < 0 || cols < 0) return MQ_ERR_ARG;
  if (w_dtype != MQ_F16 && w_dtype != MQ_BF16 && w_dtype != MQ_F32) return MQ_ERR_ARG;
  if (cfg->format != MQ_INT_SYM && cfg->format != MQ_INT_ASYM && cfg->format != MQ_LUT_FP) return MQ_ERR_ARG;
  if (out->abs_err && out->abs_err_dtype != MQ_F32 && out->abs_err_dtype != MQ_F16) return MQ_ERR_ARG;
  if (cfg->format != MQ_LUT_FP && (cfg->bits < 2 || cfg->bits > 8)) return MQ_ERR_UNSUPPORTED;
  if (cfg->f32_math && !(cfg->format == MQ_LUT_FP && cfg->group_size == -1)) return MQ_ERR_UNSUPPORTED;
  const int64_t n = rows * cols;
  if (cfg->group_size > 0 && cols % cfg->group_size != 0) return MQ_ERR_GROUP;
  if (n == 0) return MQ_OK;
  if (!w) return MQ_ERR_ARG;
  if (!workspace || workspace_bytes < mq_quantize_workspace_bytes(rows, cols, cfg)) return MQ_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);

  QuantParams p = {};
  p.w = w;
  p.in_dtype = w_dtype;
  p.f32_math = cfg->f32_math;
  p.n = n;
  p.L = cfg->group_size == -1 ? n : (cfg->group_size <= 0 ? cols : cfg->group_size);
  p.nseg = n / p.L;
  if (cfg->format == MQ_INT_SYM) p.qmax = (float)((1 << (cfg->bits - 1)) - 1);
  else if (cfg->format == MQ_INT_ASYM) p.qmax = (float)((1 << cfg->bits) - 1);
  else if (!lut_consts(cfg, cfg->f32_math != 0, &p.lut, &p.qmax)) return MQ_ERR_UNSUPPORTED;
  p.deq = static_cast<uint16_t*>(out->deq);
  p.codes = out->codes;
  p.scale = static_cast<uint16_t*>(out->scale);
  p.zp = static_cast<uint16_t*>(out->zero_point);
  p.abs_err = out->abs_err;
  p.abs_err_f16 = out->abs_err_dtype == MQ_F16;
  p.seg_sse = out->seg_sse;
  uint32_t* tred = static_cast<uint32_t*>(workspace);
  p.tred = tred;
  p.partials = reinterpret_cast<double*>(static_cast<char*>(workspace) + kWsHeader);

  // vector (16-byte) accesses need aligned bases and segments that are whole vectors
  const size_t esz = w_dtype == MQ_F32 ? 4 : 2;
  auto al = [](const void* q, size_t a) { return (reinterpret_cast<uintptr_t>(q) % a) == 0; };
  if (!al(w, esz) || !al(out->deq, 2) || !al(out->scale, 2) || !al(out->zero_point, 2) ||
      !al(out->seg_sse, 8) || !al(out->totals, 8) || !al(workspace, 16) ||
      !al(out->abs_err, out->abs_err_dtype == MQ_F16 ? 2 : 4) || (cfg->format == MQ_LUT_FP && !al(out->codes, 2)))
    return MQ_ERR_ALIGN;
  const bool vec_ok = (p.L % 8 == 0) && al(w, 16) && al(out->deq, 16) && al(out->codes, 16) && al(out->abs_err, 16);

  int scope, nv = 4, threads = 256;
  int64_t work_blocks;
  p.vecs_per_seg = (int32_t)(p.L / 8 > INT32_MAX ? INT32_MAX : p.L / 8);
  if (cfg->group_size == -1 && vec_ok) {
    scope = SC_TENSOR;
    work_blocks = (n / 8 + 64 * 4 * 4 - 1) / (64 * 4 * 4);
  } else if (!vec_ok || p.L / 8 > 16384) {
    if (cfg->f32_math) return MQ_ERR_UNSUPPORTED;  // the float32 branch needs whole, aligned 8-element vectors
    scope = -1;
    work_blocks = p.nseg;
  } else if (p.L / 8 <= 64) {
    scope = SC_SUBWAVE;
    p.team_log2 = ilog2_ceil(p.L / 8);
    int64_t spw = 64 >> p.team_log2;
    int64_t tiles = (p.nseg + spw * 4 - 1) / (spw * 4);
    work_blocks = (tiles + 3) / 4;
  } else if (p.L / 8 <= 1024) {
    scope = SC_WAVE;
    nv = 1 << ilog2_ceil((p.L / 8 + 63) / 64);
    work_blocks = (p.nseg + 3) / 4;
  } else {
    scope = SC_BLOCK;
    threads = p.L / 8 <= 4096 ? 256 : 1024;
    nv = (p.L / 8 + threads - 1) / threads <= 8 ? 8 : 16;
    work_blocks = p.nseg;
  }
  int64_t max_grid = threads == 1024 ? MQ_MAX_GRID / 4 : MQ_MAX_GRID;
  dim3 grid((unsigned)(work_blocks < max_grid ? work_blocks : max_grid));

  if (scope == SC_TENSOR) {
    MQ_HIP(hipMemsetAsync(tred, 0, 16, st));
    int64_t rb = (n / 8 + 255) / 256;
    hipLaunchKernelGGL(k_tensor_reduce, dim3((unsigned)(rb < MQ_MAX_GRID ? rb : MQ_MAX_GRID)), dim3(256), 0, st, p, tred);
  }
  switch (cfg->format) {
    case MQ_INT_SYM: launch_fmt<MQ_INT_SYM>(scope, nv, threads, grid, st, p); break;
    case MQ_INT_ASYM: launch_fmt<MQ_INT_ASYM>(scope, nv, threads, grid, st, p); break;
    default: launch_fmt<MQ_LUT_FP>(scope, nv, threads, grid, st, p); break;
  }
  MQ_HIP(hipGetLastError());
  if (out->totals || (scope == SC_TENSOR && out->seg_sse)) {
    hipLaunchKernelGGL(k_finalize_totals, dim3(1), dim3(256), 0, st, p.partials, (int)grid.x, out->totals,
                       scope == SC_TENSOR ? out->seg_sse : nullptr);
    MQ_HIP(hipGetLastError());
  }
  return MQ_OK;
}
