// Fused quantize -> dequantize -> error kernels for gfx950 (MI355X).
//
// One pass per weight tensor: each lane loads 8 consecutive elements (16 B of float16), the lanes that
// share a scale ("team": part of a wave, a wave, or a workgroup) reduce absmax or min/max across lanes
// (DPP inside a 16-lane row, one LDS exchange for workgroup teams), every lane derives the segment's
// scale / zero-point, quantizes and dequantizes its 8 elements from registers with packed float16
// arithmetic, stores the requested outputs with 8/16-byte vector stores and accumulates the float32
// error.  The input is read from HBM exactly once; there is no contraction, so no MFMA: the kernel is
// bounded by HBM bandwidth once its VALU work per element is small enough, which is what the packed
// formulation below is for.
//
// Arithmetic contract (reference: /root/reference/quant_utils/quant_weight.py :229-335, :503-601,
// :717-872): the reference runs every op as its own float16 torch kernel, i.e. widen to float32,
// compute, round once to float16.  Each h(...) below is one such rounding:
//   v  = h(w)
//   sym : s = clamp(h(max|v| / qmax));  q = clamp(rint(h(v / s)), +-qmax);          d = h(q * s)
//   asym: s = clamp(h(h(rmax - rmin) / qmax));  zp = clamp(rint(h(-rmin / s)), 0, qmax)
//         q = clamp(h(rint(h(v / s)) + zp), 0, qmax);                               d = h(h(q - zp) * s)
//   lut : s = clamp(h(max|v| / LUTmax));  q = nearest table value to h(v / s), ties toward -inf;  d = h(q * s)
// Every shortcut taken on the fast path is exact on its stated domain (proofs: exhaustive CPU sweeps in
// tests/test_division_shortcut.py); segments outside that domain (inf / NaN, or |v/s| >= 500) take the
// per-element slow path, which follows the contract literally.
// This file must be compiled with -ffp-contract=off and without fast-math.
//
// The file is compiled five times (Makefile): MQ_PART 0 holds the host side and the format-independent
// kernels; MQ_PART 1 / 2 / 3 / 4 hold the kernel instantiations of one format each (symmetric INT, asymmetric INT,
// value tables without / with subnormals) behind mq_launch_fmt<N>(), so that a clean build runs in parallel.
#include <stdlib.h>
#include <type_traits>

#include "mq_common.h"
#include "mq_sel.h"

#ifndef MQ_PART
#error "compile with -DMQ_PART=0..4 (see Makefile)"
#endif
#define MQ_IS_LUT(F) ((F) == MQ_LUT_FP || (F) == MQ_LUT_FP_SUB)

enum { SC_SUBWAVE = 0, SC_WAVE = 1, SC_BLOCK = 2, SC_TENSOR = 3,
       // the statistics-only kernels (k_stats_*): groups of 32 * 2^team_log2 elements; rows held by one wave / by
       // a 256-thread workgroup with nv vectors per lane
       SC_STATS_GROUP = 4, SC_STATS_ROW = 5, SC_STATS_ROWBLK = 6, SC_STATS_ROWNP = 7, SC_TENSOR_NP = 8 };
// The kernels' SEL template parameter: one of the SEL_FEED_* values, or SEL_STATS_ONLY = no selection feed and no
// element-sized output (deq / codes / abs_err all absent: the sweeps that only want the error statistics), which
// drops the per-vector tests of the output pointers and everything that only served the stores.
enum { SEL_STATS_ONLY = 3 };
enum { LIST_OFF = 0, LIST_SAMPLE = 1, LIST_FEED = 2, LIST_REPAIR = 3 };
#define SEL_FEEDS(S) ((S) == SEL_FEED_SAMPLE || (S) == SEL_FEED_COUNT)
#define SEL_STORES(S) ((S) == SEL_FEED_OFF || (S) == SEL_FEED_COUNT)  // may have element-sized outputs

struct LutConst {
  uint32_t shift, half, maxb, minpos, mid;  // rounding constants in the float16 (or float32) bit domain
  // tables WITH subnormals (MQ_LUT_FP_SUB): values m * 2^(1-bias-M) below sub_lim = 2^(1-bias), then
  // (1 + m / 2^M) * 2^(e-bias) up to sub_max.  Fast path: lut_round_pk_sub on the float16 bits (shift / half /
  // maxb / minpos / mid as above, with minpos = the grid step u and mid = u / 2; elim = the float16 exponent
  // field of sub_lim); per-element path and the float32 branch: lut_round_sub in float64.
  int32_t sub, man;
  uint32_t elim;
  double sub_lim, sub_max;
};

struct QuantParams {
  const void* w;
  int32_t in_dtype;
  int32_t f32_math;
  int64_t n;     // elements
  int64_t L;     // segment length in elements
  int64_t nseg;  // segments
  int32_t vecs_per_seg;  // L / 8
  int32_t need_err;      // any of abs_err / seg_sse / totals requested
  float qmax;            // sym: 2^(b-1)-1, asym: 2^b-1, lut: largest table value
  float rqmax;           // RN(1 / qmax)
  int32_t qmax_recip_ok; // a * rqmax with one correction step equals a / qmax for every float16 a
  LutConst lut;
  uint16_t* deq;
  void* codes;
  uint16_t* scale;
  uint16_t* zp;
  void* abs_err;
  int32_t abs_err_f16;
  double* seg_sse;
  double seg_mse_scale;  // 0: seg_sse receives float64 sums; 1 / L: seg_sse points to float32 words that receive
                         // the segment's mean squared error float32(sum * (1 / L))
  int32_t seg_exact;     // the float32 entry is summed in numpy's pairwise order instead (k_stats_group<.., true>)
  double* partials;      // [gx][2]
  float* leaf;           // numpy-order sweeps: the sums of the tensor's 128-element leaves (flattened), or null
  // sweeps that select a list's percentile inside the pass (see mq_sel.h, ListFeedState):
  int32_t list_mode;     // LIST_OFF / LIST_SAMPLE (the sampling launch) / LIST_FEED (the pass) / LIST_REPAIR
  ListFeedState* feed;   // this tensor's feed state, or null: its list is written as always
  uint32_t blo, bhi;     // ... its bracket (float32 bit patterns, inclusive)
  uint32_t* wg_cnt;      // ... per workgroup of the tensor: entries under the bracket, entries kept
  uint32_t* slots;       // ... per workgroup: LIST_SLOT words for the kept bit patterns
  float* sample;         // sampling launch: where this tensor's sample entries go (null: not sampled)
  uint32_t sunit, scount;  // one wave step of every sunit is sampled, scount of them
  uint32_t bx, gx;       // this workgroup's index among the gx workgroups that share the tensor (blockIdx.x and
                         // gridDim.x, or the tensor's block range of a sweep launch)
  const uint32_t* tred;  // SC_TENSOR: reduction words written by k_tensor_reduce
  int32_t sel_mode;      // SEL_FEED_*: feed the float32 |error| values to a percentile selection
  SelWs* sel_ws;         // this tensor's selection workspace
  uint16_t* sel_samples; // sampling pass: this tensor's dense sample array
};

// Per-lane accumulators threaded through the element routines.
struct Acc {
  double abs;   // sum of |error|
  SelFeed sel;  // selection feed (off unless sel_mode)
  float* xp;    // this wave's 512-float LDS exchange area for the float32 |error| stores, or null
  float* sq = nullptr;  // per-element path of the statistics kernels: where the 8 float32 squared errors go, or null
};
extern __shared__ uint32_t sel_lds[];  // 2 * SEL_FEED_HASH words when sel_mode is on, else none

// Up to MQ_MAX_BATCH same-shaped tensors per launch (blockIdx.y picks the tensor); the per-tensor
// pointers travel in the kernel arguments, so a batch needs no descriptor upload.
#define MQ_MAX_BATCH 32
struct BatchItem {
  const void* w;
  uint16_t* deq;
  void* codes;
  uint16_t* scale;
  uint16_t* zp;
  void* abs_err;
  double* seg_sse;
  double* totals;
};
// One pass over tensors of DIFFERENT sizes in one launch (statistics kernels): a one-dimensional grid, tensor i
// owns the workgroups [blk0, blk0 + nblk) and has its own segment count; the table lives in device memory (any
// number of tensors; the kernel arguments could not hold it).
struct SweepItem {  // 120 bytes
  const void* w;
  uint16_t* scale;
  uint16_t* zp;
  double* seg_sse;
  double* totals;
  int64_t nseg;
  uint32_t blk0, nblk;
  float* leaf;       // numpy-order sweeps: this tensor's 128-element leaf sums (flattened tensor)
  uint32_t lchunk0;  // ... and the index of its first whole 8192-entry chunk among those of all the pass's lists
  uint32_t pad;
  ListFeedState* feed;  // fed lists: this tensor's state (null: the list is written)
  uint32_t blo, bhi;    // ... its bracket, written into the DEVICE copy of the table by k_list_bracket
  uint32_t* wg_cnt;     // ... the workgroups' counts and slots
  uint32_t* slots;
  float* sample;        // sampling launch: the sample entries
  uint32_t sunit, scount;
};
static_assert(sizeof(SweepItem) == 120, "table entry");
static_assert(offsetof(SweepItem, bhi) == offsetof(SweepItem, blo) + 4, "the bracket is stored as two words");
// numpy's summation of a contiguous run of n float32 values (n a multiple of 8, n <= 32768), unrolled by the host:
// chunks of 8192 elements (the reduction's inner loop size) added up in order, each chunk summed pairwise -- runs
// longer than 128 elements are split in two, the left part n / 2 rounded down to a multiple of 8, until the parts
// hold at most 128 elements (numpy/_core/src/umath/loops_utils.h.src).  len[i]: vectors (8 elements)
// of leaf i, in memory order; ops: the post-order walk of the split tree, one bit per step -- 0: take the next
// leaf's sum, 1: add the two newest partial sums (the older one on the left).
#define MQ_NP_MAX_LEAVES 512
struct NumpyProg {
  uint16_t nleaves, nops;
  uint8_t len[MQ_NP_MAX_LEAVES];
  uint32_t ops[2 * MQ_NP_MAX_LEAVES / 32];
};
struct QuantBatch {
  QuantParams common;  // shape, format and flags; pointer fields are filled per tensor
  NumpyProg np;        // SC_STATS_ROWNP only
  const SweepItem* sweep;  // non-null: a sweep launch (sweep_count entries sorted by blk0); item[] is unused
  const struct NpRowProg* rowprog;  // numpy-order row kernels: the row's summation program (device memory)
  const SweepItem* np_tabs[3];      // k_np_group3: the launch tables of the passes with groups of 32 / 64 / 128 (or null)
  const struct MultiItem* multi;    // k_stats_multi: the launch's tensors (device memory, sorted by blk0) ...
  int32_t multi_count;              // ... how many, and which groupings are computed (bit 0 rows, 1 .. 3: groups of
  uint32_t multi_mask;              //     32 / 64 / 128); sweep_partials holds 8 doubles per workgroup then
  int32_t sweep_count;
  double* sweep_partials;  // [total workgroups][2]
  int64_t ws_stride;   // bytes between the tensors' workspace slices
  char* sel_ws;        // selection workspace of tensor 0 (slices of sizeof(SelWs)), when common.sel_mode
  int64_t sample_stride;  // uint16 elements between the tensors' sample arrays (common.sel_samples: tensor 0)
  BatchItem item[MQ_MAX_BATCH];
};
static_assert(sizeof(QuantBatch) <= 4096, "kernel arguments");
static const size_t kWsHeader = 256;  // reduction words at the start of a tensor's workspace slice

__device__ __forceinline__ QuantParams batch_params(const QuantBatch& b) {
  QuantParams p = b.common;
  if (b.sweep) {
    // the last entry whose first workgroup is not past this one (uniform over the workgroup: scalar loads)
    int lo = 0, hi = b.sweep_count - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (b.sweep[mid].blk0 <= blockIdx.x) lo = mid;
      else hi = mid - 1;
    }
    const SweepItem& it = b.sweep[lo];
    p.w = it.w; p.scale = it.scale; p.zp = it.zp; p.seg_sse = it.seg_sse;
    p.deq = nullptr; p.codes = nullptr; p.abs_err = nullptr;
    p.nseg = it.nseg;
    p.n = it.nseg * p.L;
    p.bx = blockIdx.x - it.blk0;
    p.gx = it.nblk;
    p.partials = b.sweep_partials + 2 * (int64_t)it.blk0;
    p.leaf = it.leaf;
    p.feed = it.feed; p.blo = it.blo; p.bhi = it.bhi; p.wg_cnt = it.wg_cnt; p.slots = it.slots;
    p.sample = it.sample; p.sunit = it.sunit; p.scount = it.scount;
    p.tred = nullptr; p.sel_ws = nullptr; p.sel_samples = nullptr;
    return p;
  }
  p.bx = blockIdx.x;
  p.gx = gridDim.x;
  const BatchItem& it = b.item[blockIdx.y];
  p.w = it.w; p.deq = it.deq; p.codes = it.codes; p.scale = it.scale; p.zp = it.zp; p.abs_err = it.abs_err;
  p.seg_sse = it.seg_sse;
  char* ws = reinterpret_cast<char*>(const_cast<uint32_t*>(b.common.tred)) + (int64_t)blockIdx.y * b.ws_stride;
  p.tred = reinterpret_cast<const uint32_t*>(ws);
  p.partials = reinterpret_cast<double*>(ws + kWsHeader);
  p.sel_ws = p.sel_mode ? reinterpret_cast<SelWs*>(b.sel_ws + (int64_t)blockIdx.y * sizeof(SelWs)) : nullptr;
  p.sel_samples = p.sel_mode ? b.common.sel_samples + (int64_t)blockIdx.y * b.sample_stride : nullptr;
  return p;
}

__device__ __forceinline__ void store_seg(const QuantParams& p, int64_t seg, double sum) {
  if (p.seg_mse_scale != 0.0) reinterpret_cast<float*>(p.seg_sse)[seg] = (float)(sum * p.seg_mse_scale);
  else p.seg_sse[seg] = sum;
}

struct SegP {
  half_t s;    // scale
  half_t zp;   // zero-point (asym)
  float sf;    // (float)s
  float r;     // ~1 / sf (v_rcp_f32, at most 1 ulp off: enough for div_h, see there)
  bool slow;   // this segment must take the per-element path
  // statistics-only integer formats (div_pk16): P = 2^k with s * P in [1, 2), s * P and RN16(1 / (s * P)), each
  // as a pair of equal float16 halves
  uint32_t pk_p, pk_s, pk_r;
  int pk_k;        // log2 of P
  bool pk_shared;  // P is the power of two the caller offered (a larger segment's), not the segment's own
};

// ------------------------------------------------------------------------------------------------
// small instruction wrappers
// ------------------------------------------------------------------------------------------------
// Packed 16-bit min / max / saturating subtract.  Written as instructions: hipcc (ROCm 7.2) drops all
// but the first __builtin_elementwise_min/max of a chain over _Float16 vectors.
#define MQ_PK_OP(name, insn)                                            \
  __device__ __forceinline__ uint32_t name(uint32_t a, uint32_t b) {    \
    uint32_t r;                                                         \
    asm(insn " %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));                 \
    return r;                                                           \
  }
#define MQ_PK_OP3(name, insn)                                                      \
  __device__ __forceinline__ uint32_t name(uint32_t a, uint32_t b, uint32_t c) {  \
    uint32_t r;                                                                   \
    asm(insn " %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));               \
    return r;                                                                     \
  }
MQ_PK_OP(pk_min_f16, "v_pk_min_f16")
MQ_PK_OP(pk_max_f16, "v_pk_max_f16")
MQ_PK_OP(pk_min_u16, "v_pk_min_u16")
MQ_PK_OP(pk_max_u16, "v_pk_max_u16")
MQ_PK_OP(pk_add_u16, "v_pk_add_u16")
MQ_PK_OP(pk_sub_u16, "v_pk_sub_u16")
// NaN-propagating packed float16 max / min of three (gfx950): a NaN operand gives a NaN half, -0 < +0
MQ_PK_OP3(pk_maximum3_f16, "v_pk_maximum3_f16")
MQ_PK_OP3(pk_minimum3_f16, "v_pk_minimum3_f16")
__device__ __forceinline__ uint32_t pk_sub_u16_sat(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t pk_shr_u16(uint32_t sh, uint32_t v);
__device__ __forceinline__ h16x2 as_h2(uint32_t w) { return __builtin_bit_cast(h16x2, w); }
__device__ __forceinline__ uint32_t as_u32(h16x2 h) { return __builtin_bit_cast(uint32_t, h); }
__device__ __forceinline__ uint32_t dup16(uint32_t b) { return (b & 0xffffu) | (b << 16); }

// Cross-lane moves inside a 16-lane row (DPP): partner lane^1, lane^2, and the mirrored partner in the
// row's 8-lane halves / in the row, which serve as the lane^4 / lane^8 steps of a butterfly.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}
enum { DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140 };

template <int T, typename F>
__device__ __forceinline__ uint32_t team_reduce_u32(uint32_t m, F op) {
  if constexpr (T >= 2) m = op(m, dpp<DPP_XOR1>(m));
  if constexpr (T >= 4) m = op(m, dpp<DPP_XOR2>(m));
  if constexpr (T >= 8) m = op(m, dpp<DPP_HALF_MIRROR>(m));
  if constexpr (T >= 16) m = op(m, dpp<DPP_MIRROR>(m));
  if constexpr (T >= 32) m = op(m, (uint32_t)__shfl_xor((int)m, 16));
  if constexpr (T >= 64) m = op(m, (uint32_t)__shfl_xor((int)m, 32));
  return m;
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  uint64_t u = __builtin_bit_cast(uint64_t, v);
  uint32_t lo = dpp<CTRL>((uint32_t)u), hi = dpp<CTRL>((uint32_t)(u >> 32));
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
// fixed butterfly order -> bitwise reproducible
template <int T>
__device__ __forceinline__ double team_sum(double v) {
  if constexpr (T >= 2) v += dpp_f64<DPP_XOR1>(v);
  if constexpr (T >= 4) v += dpp_f64<DPP_XOR2>(v);
  if constexpr (T >= 8) v += dpp_f64<DPP_HALF_MIRROR>(v);
  if constexpr (T >= 16) v += dpp_f64<DPP_MIRROR>(v);
  if constexpr (T >= 32) v += __shfl_xor(v, 16);
  if constexpr (T >= 64) v += __shfl_xor(v, 32);
  return v;
}

// ------------------------------------------------------------------------------------------------
// loads: 8 consecutive elements of the caller's tensor as packed float16 (w.to(torch.float16))
// ------------------------------------------------------------------------------------------------
// bfloat16 -> float16 (round to nearest even, as w.to(torch.float16) does) for 8 packed elements
__device__ __forceinline__ u32x4 bf16x8_to_f16(u32x4 b) {
  u32x4 out;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    h16x2 h;
    h[0] = (half_t)f_from_bits(b[k] << 16);
    h[1] = (half_t)f_from_bits(b[k] & 0xffff0000u);
    out[k] = as_u32(h);
  }
  return out;
}

// Two-step load for the software-pipelined kernels: load8_raw only moves bytes (float16 / bfloat16: one
// 16-byte load that can stay in flight; float32: converted at once, it needs two loads anyway), raw_to_f16
// converts bfloat16 when the values are consumed.
__device__ __forceinline__ u32x4 load8_raw(const void* w, int dtype, int64_t e) {
  if (dtype != MQ_F32) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(w) + e));
  u32x4 out;
  const f32x4* p = reinterpret_cast<const f32x4*>(static_cast<const float*>(w) + e);
  f32x4 a = p[0], c = p[1];
  h16x2 h0 = {(half_t)a[0], (half_t)a[1]}, h1 = {(half_t)a[2], (half_t)a[3]};
  h16x2 h2 = {(half_t)c[0], (half_t)c[1]}, h3 = {(half_t)c[2], (half_t)c[3]};
  out[0] = as_u32(h0); out[1] = as_u32(h1); out[2] = as_u32(h2); out[3] = as_u32(h3);
  return out;
}
__device__ __forceinline__ u32x4 raw_to_f16(u32x4 raw, int dtype) { return dtype == MQ_BF16 ? bf16x8_to_f16(raw) : raw; }

__device__ __forceinline__ u32x4 load8_f16(const void* w, int dtype, int64_t e) {
  return raw_to_f16(load8_raw(w, dtype, e), dtype);
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
// wins, which is torch.amax's NaN propagation.  Returns the packed maximum of the even / odd elements.
__device__ __forceinline__ uint32_t absmax8_pk(u32x4 v) {
  const uint32_t m = 0x7fff7fffu;
  return pk_max_u16(pk_max_u16(v[0] & m, v[1] & m), pk_max_u16(v[2] & m, v[3] & m));
}
__device__ __forceinline__ uint32_t fold_max_u16(uint32_t pk) {
  uint32_t a = pk & 0xffffu, b = pk >> 16;
  return a > b ? a : b;
}
// packed (min over even, min over odd) / (max ...) of the signed float16 values.  v_pk_minimum3 / maximum3_f16
// (gfx950) return a NaN when an operand is one -- torch.amin / amax propagate a NaN in the segment into scale and
// zero-point, and v_pk_min / max_f16 would drop it.
__device__ __forceinline__ void minmax8_pk(u32x4 v, uint32_t& lo, uint32_t& hi) {
  lo = pk_minimum3_f16(pk_minimum3_f16(v[0], v[1], v[2]), v[3], v[3]);
  hi = pk_maximum3_f16(pk_maximum3_f16(v[0], v[1], v[2]), v[3], v[3]);
}
// -> one word: min in the low half, max in the high half
__device__ __forceinline__ uint32_t fold_minmax(uint32_t lo, uint32_t hi) {
  // (lo.l, hi.l) vs (lo.h, hi.h): min of the low halves, max of the high halves
  uint32_t a = (lo & 0xffffu) | (hi << 16), b = (lo >> 16) | (hi & 0xffff0000u);
  uint32_t mn = pk_minimum3_f16(a, b, b), mx = pk_maximum3_f16(a, b, b);
  return (mn & 0xffffu) | (mx & 0xffff0000u);
}
__device__ __forceinline__ uint32_t op_minmax(uint32_t a, uint32_t b) {
  uint32_t mn = pk_minimum3_f16(a, b, b), mx = pk_maximum3_f16(a, b, b);
  return (mn & 0xffffu) | (mx & 0xffff0000u);
}
__device__ __forceinline__ uint32_t op_max_u32(uint32_t a, uint32_t b) { return a > b ? a : b; }

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
// h(a / qmax) for a float16 a.  With RN(1/qmax) and one correction step the float32 quotient is exact
// for every finite a when the host has verified it for this qmax (qmax_recip_ok); otherwise divide.
__device__ __forceinline__ half_t div_qmax(half_t a, const QuantParams& p) {
  float af = (float)a;
  if (p.qmax_recip_ok) {
    float q0 = af * p.rqmax;
    return (half_t)__builtin_fmaf(__builtin_fmaf(-q0, p.qmax, af), p.rqmax, q0);
  }
  return (half_t)(af / p.qmax);
}
// v_rcp_f32 is a transcendental instruction: a vector instruction that reads its result needs one wait state in
// between.  The compiler puts it in front of the instructions it emits itself, but not in front of an inline-asm
// statement -- and several of the statistics kernels' instructions are inline asm.  The copy below carries the wait
// state, so whatever reads the reciprocal afterwards reads a settled register.  (Found the hard way: a packed
// conversion written as asm directly behind a reciprocal gave wrong scales in ONE instantiation of the row kernels,
// where the scheduler had left nothing between the two.)
__device__ __forceinline__ float rcp_settled(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  float out;
  asm("s_nop 0\n\tv_mov_b32 %0, %1" : "=v"(out) : "v"(r));
  return out;
}
__device__ __forceinline__ void finish_params(SegP& sp, half_t s) {
  sp.s = s;
  sp.sf = (float)s;
  sp.r = rcp_settled(sp.sf);
}

// h(v / s) for finite v.  q1 rounds to the same float16 as the IEEE quotient v / s for every finite
// float16 v, every float16 s in [0x00A8, 0x70E2] (the clamped scale range) and every r within 2 ulp of
// 1/s -- checked exhaustively on the CPU (tests/test_division_shortcut.py); v_rcp_f32 is within 1 ulp.
// The sign of a zero quotient is lost by the correction step: callers restore it from v.
__device__ __forceinline__ float div_f(float vf, const SegP& sp) {
  float q0 = vf * sp.r;
  return __builtin_fmaf(__builtin_fmaf(-q0, sp.sf, vf), sp.r, q0);
}
// the same with +-0 and +-inf quotients handled (class mask: -inf | -0 | +0 | +inf)
__device__ __forceinline__ half_t div_h_full(float vf, const SegP& sp) {
  float q0 = vf * sp.r;
  float q1 = __builtin_fmaf(__builtin_fmaf(-q0, sp.sf, vf), sp.r, q0);
  q1 = __builtin_amdgcn_classf(q0, 0x264) ? q0 : q1;
  if (sp.sf != sp.sf) q1 = sp.sf;
  return (half_t)q1;
}

// amax: sign-cleared float16 bits
__device__ __forceinline__ SegP params_absmax(uint32_t amax_bits, const QuantParams& p) {
  SegP sp;
  const half_t s0 = div_qmax(h_from_bits(amax_bits), p);
  finish_params(sp, clamp_scale(s0));
  sp.zp = (half_t)0;
  // inf or NaN in the segment; or a scale cut down to the upper clamp (bit widths of 2 and 3 with values above
  // 1e4 * qmax): then |v / s| exceeds qmax and the clamp of the codes, which the fast path leaves out, acts
  sp.slow = amax_bits >= 0x7c00u || s0 > h_from_bits(0x70E2u);
  if (sp.slow) finish_params(sp, clamp_scale((half_t)((float)h_from_bits(amax_bits) / p.qmax)));
  return sp;
}
// mm: min in the low half, max in the high half
__device__ __forceinline__ SegP params_minmax(uint32_t mm, const QuantParams& p) {
  SegP sp;
  half_t rmin = h_from_bits(mm & 0xffffu), rmax = h_from_bits(mm >> 16);
  half_t range = rmax - rmin;
  finish_params(sp, clamp_scale(div_qmax(range, p)));
  const half_t qmax_h = (half_t)p.qmax;
  float nm = -(float)rmin;
  half_t z = __builtin_rintf16(div_h_full(nm, sp));  // keeps the sign of a zero -rmin, as the reference does
  z = z < (half_t)0 ? (half_t)0 : z;
  z = z > qmax_h ? qmax_h : z;
  sp.zp = z;
  float big = __builtin_fmaxf(__builtin_fabsf((float)rmin), __builtin_fabsf((float)rmax));
  // fast path needs finite data and |v / s| < 512 (exact integer rounding by the 1536 trick)
  sp.slow = !(big * sp.r < 500.0f) || !((float)range < 65504.0f);
  if (sp.slow) {
    finish_params(sp, clamp_scale((half_t)((float)range / p.qmax)));
    sp.r = 1.0f / sp.sf;
    half_t zz = __builtin_rintf16(div_h_full(nm, sp));
    zz = zz < (half_t)0 ? (half_t)0 : zz;
    zz = zz > qmax_h ? qmax_h : zz;
    sp.zp = zz;
  }
  return sp;
}

// ------------------------------------------------------------------------------------------------
// The sign of a zero zero-point.  zero-point = clamp(round(-rmin / scale)) is a zero when the segment's minimum is
// one, and it carries the sign of -rmin.  When a segment holds zeros of both signs torch.amin (quant_weight.py:757)
// returns the one its reduction meets first: the first one in memory for segments of fewer than 64 elements, which
// it walks in order.  (From 64 elements on the reduction runs through the vector lanes of the CPU and the winner
// depends on their width; there -0 orders below +0 here, as in the packed minimum.)  The packed minimum always
// prefers -0, so the only case to put right is "rmin = -0 although a +0 comes before the first -0": the zero-point
// is -0 then.  No other output depends on it.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t op_min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }
// index of the first +0 / first -0 among the 8 halves of a vector whose first element is element e0 of its segment
__device__ __forceinline__ void first_zeros8(u32x4 v, uint32_t e0, uint32_t& fp, uint32_t& fm) {
#pragma unroll
  for (int i = 7; i >= 0; --i) {
    const uint32_t h = (v[i >> 1] >> (16 * (i & 1))) & 0xffffu;
    fp = h == 0u ? e0 + (uint32_t)i : fp;
    fm = h == 0x8000u ? e0 + (uint32_t)i : fm;
  }
}
// -> the bits of the zero-point to store.  mm: the segment's (min, max) word; the segment is held by the T lanes of
// a team, this lane's raw[j] being vector vec0 + j * vstep of it (holds: false for a lane without data).  Every
// lane of the wave calls this together.
template <int T, int NVV>
__device__ __forceinline__ uint32_t zp_bits_first_zero(const SegP& sp, uint32_t mm, const u32x4* raw, int vec0, int vstep,
                                                        bool holds, int64_t L) {
  const uint32_t zb = h_bits(sp.zp);
  const bool need = zb == 0u && (mm & 0xffffu) == 0x8000u && L < 64;
  if (__builtin_amdgcn_ballot_w64(need) == 0ull) return zb;
  uint32_t fp = 0xffffu, fm = 0xffffu;
  if (holds) {
#pragma unroll
    for (int j = NVV - 1; j >= 0; --j) first_zeros8(raw[j], (uint32_t)(vec0 + j * vstep) * 8u, fp, fm);
  }
  fp = team_reduce_u32<T>(fp, op_min_u32);
  fm = team_reduce_u32<T>(fm, op_min_u32);
  return (need && fp < fm) ? 0x8000u : zb;
}

// Nearest value of an (E, M, bias) table for x given as float16 (or float32) bits: round the mantissa
// to M bits (ties toward -inf: down in magnitude for x > 0, up for x < 0), saturate at the largest
// entry, and below the smallest positive entry choose between it and 0 at their midpoint.  NaN -> +0.
__device__ __forceinline__ uint32_t lut_round_bits(uint32_t xb, uint32_t signbit, uint32_t nanlimit, const LutConst& c) {
  uint32_t a = xb & (signbit - 1u);
  bool neg = (xb & signbit) != 0;
  uint32_t r = ((a + (neg ? c.half : c.half - 1u)) >> c.shift) << c.shift;
  r = r > c.maxb ? c.maxb : r;
  if (r < c.minpos) r = (neg ? a >= c.mid : a > c.mid) ? c.minpos : 0u;
  if (a > nanlimit) r = 0u;
  return r ? (r | (neg ? signbit : 0u)) : 0u;
}
// The same for two float16 values in one word, branch-free (no NaN handling: NaN segments are slow-path).
__device__ __forceinline__ uint32_t lut_round_pk(uint32_t xw, const LutConst& c) {
  uint32_t a = xw & 0x7fff7fffu;
  uint32_t an = pk_add_u16(a, (xw >> 15) & 0x00010001u);  // a + 1 for negative x: turns '>=' into '>'
  uint32_t r = pk_add_u16(an, dup16(c.half - 1u)) & dup16(0xffffu << c.shift);
  r = pk_max_u16(pk_min_u16(r, dup16(c.maxb)), dup16(c.minpos));
  uint32_t z = pk_sub_u16_sat(an, dup16(c.mid));  // nonzero iff |x| is past the 0 / minpos midpoint
  uint32_t m = pk_sub_u16(0u, pk_min_u16(z, 0x00010001u));  // 0xffff or 0 per half
  return ((xw & 0x80008000u) | r) & m;  // zero is +0
}

// The packed rounding for a table WITH subnormals.  Below sub_lim = 2^(elim-15) the table is the uniform grid
// m * u, u = sub_lim / 2^M: a value whose exponent field is elim - k (k = 1 .. M) keeps M - k mantissa bits,
// i.e. the mantissa rounding above with a shift that grows by k; below u the choice is between u and 0 at
// their midpoint (minpos = u, mid = u / 2: the same rule as for the smallest entry of a table without
// subnormals; u is a normal float16, so every float16 subnormal input lands here).
__device__ __forceinline__ uint32_t pk_shl_u16(uint32_t sh, uint32_t v) {
  uint32_t r;
  asm("v_pk_lshlrev_b16 %0, %1, %2" : "=v"(r) : "v"(sh), "v"(v));
  return r;
}
__device__ __forceinline__ uint32_t pk_shr_u16(uint32_t sh, uint32_t v) {
  uint32_t r;
  asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(r) : "v"(sh), "v"(v));
  return r;
}
__device__ __forceinline__ uint32_t lut_round_pk_sub(uint32_t xw, const LutConst& c) {
  uint32_t a = xw & 0x7fff7fffu;
  uint32_t an = pk_add_u16(a, (xw >> 15) & 0x00010001u);  // a + 1 for negative x: turns '>=' into '>'
  uint32_t k = pk_min_u16(pk_sub_u16_sat(dup16(c.elim), pk_shr_u16(dup16(10u), a)), dup16((uint32_t)c.man));
  uint32_t one = pk_shl_u16(pk_add_u16(k, dup16(c.shift)), 0x00010001u);  // 1 << shift, per half
  uint32_t r = pk_add_u16(an, pk_sub_u16(pk_shr_u16(0x00010001u, one), 0x00010001u)) & pk_sub_u16(0u, one);
  r = pk_max_u16(pk_min_u16(r, dup16(c.maxb)), dup16(c.minpos));
  uint32_t z = pk_sub_u16_sat(an, dup16(c.mid));  // nonzero iff |x| is past the 0 / u midpoint
  uint32_t m = pk_sub_u16(0u, pk_min_u16(z, 0x00010001u));
  return ((xw & 0x80008000u) | r) & m;  // zero is +0
}

// Nearest value of a table with subnormals, ties toward -inf, NaN -> +0, zero is +0.  Below sub_lim the
// table is the uniform grid m * 2^(1-bias-M): adding sub_lim moves |x| into the binade [sub_lim, 2 sub_lim)
// whose M-bit mantissa grid is exactly that grid (the sum is exact in float64 for float16 and float32 x
// near every tie), so one mantissa rounding serves both ranges.  Ties: a positive tie goes to the smaller
// magnitude, a negative one to the larger (the reference picks entry i iff mid[i-1] < x <= mid[i]).
__device__ __forceinline__ float lut_round_sub(float x, const LutConst& c) {
  if (x != x) return 0.0f;
  const bool neg = x < 0.0f;
  const double a = __builtin_fabs((double)x);
  double r;
  if (a >= c.sub_max) {
    r = c.sub_max;
  } else {
    const bool low = a < c.sub_lim;
    const double y = low ? a + c.sub_lim : a;
    const int sh = 52 - c.man;
    const uint64_t half = 1ull << (sh - 1);
    uint64_t b = __builtin_bit_cast(uint64_t, y);
    b = (b + (neg ? half : half - 1ull)) & ~((1ull << sh) - 1ull);
    r = __builtin_bit_cast(double, b);
    if (low) r -= c.sub_lim;
    if (r > c.sub_max) r = c.sub_max;
  }
  if (r == 0.0) return 0.0f;
  return neg ? -(float)r : (float)r;
}

// ------------------------------------------------------------------------------------------------
// one lane's 8 elements
// ------------------------------------------------------------------------------------------------
// A lane holds 8 consecutive elements, i.e. 32 consecutive bytes of the float32 |error| array: stored straight
// from the lane, each of the two 16-byte store instructions of a wave would fill only half of every 64 bytes it
// touches.  When all 64 lanes are active (their 512 elements are then contiguous) the wave exchanges the
// values through LDS so that each store instruction writes 1 KiB of consecutive bytes (measured: 8 % off the
// whole pass).
__device__ __forceinline__ void store_abs_err(const QuantParams& p, const float* ae, int64_t e0, float* xp) {
  if (p.abs_err_f16) {
    u32x4 a;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      h16x2 h = {(half_t)ae[2 * k], (half_t)ae[2 * k + 1]};
      a[k] = as_u32(h) & 0x7fff7fffu;
    }
    *reinterpret_cast<u32x4*>(static_cast<uint16_t*>(p.abs_err) + e0) = a;
  } else {
    f32x4 a0 = {__builtin_fabsf(ae[0]), __builtin_fabsf(ae[1]), __builtin_fabsf(ae[2]), __builtin_fabsf(ae[3])};
    f32x4 a1 = {__builtin_fabsf(ae[4]), __builtin_fabsf(ae[5]), __builtin_fabsf(ae[6]), __builtin_fabsf(ae[7])};
    if (xp && __builtin_amdgcn_ballot_w64(true) == ~0ull) {
      const int lane = threadIdx.x & 63;
      f32x4* x4 = reinterpret_cast<f32x4*>(xp);
      x4[2 * lane] = a0;
      x4[2 * lane + 1] = a1;
      __builtin_amdgcn_wave_barrier();  // LDS executes a wave's accesses in order; this keeps the compiler from
      f32x4 b0 = x4[lane], b1 = x4[64 + lane];  // moving the reads of other lanes' values above the writes
      __builtin_amdgcn_wave_barrier();
      f32x4* ob = reinterpret_cast<f32x4*>(static_cast<float*>(p.abs_err) + (e0 - (int64_t)lane * 8));
      __builtin_nontemporal_store(b0, &ob[lane]);
      __builtin_nontemporal_store(b1, &ob[64 + lane]);
    } else {
      f32x4* o = reinterpret_cast<f32x4*>(static_cast<float*>(p.abs_err) + e0);
      o[0] = a0;
      o[1] = a1;
    }
  }
}

// Squares are rounded to float32 per element as numpy does (quant_error_multi_weight.py:57-58); a lane's
// 8 squares (and 8 absolute errors) are summed pairwise in float32 -- at most 3 roundings, 1.8e-7
// relative -- and every sum above that level is carried in float64.
__device__ __forceinline__ void error_sums(const float* e, double& sse, double& sabs) {
  float q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) q[i] = e[i] * e[i];
  float s = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
  float a = ((__builtin_fabsf(e[0]) + __builtin_fabsf(e[1])) + (__builtin_fabsf(e[2]) + __builtin_fabsf(e[3]))) +
            ((__builtin_fabsf(e[4]) + __builtin_fabsf(e[5])) + (__builtin_fabsf(e[6]) + __builtin_fabsf(e[7])));
  sse = (double)s;
  sabs += (double)a;
}

// ---- statistics-only fast path ---------------------------------------------------------------------------------
// Measured issue cost on gfx950 (cycles per wave instruction per SIMD, 4 .. 8 waves per SIMD): plain float32
// add / mul / fma and 32-bit and: ~2.4; every packed, conversion, mixed-precision, dot, v_bfi / v_perm, DPP and
// float64 instruction: ~4.2; v_rcp_f32, v_med3_f16, v_max3_f16: ~8.  The sweeps that only want error statistics
// are bound by that issue rate, so this path is written for the fewest issue cycles per pair of elements:
//  * the float16 operands of the exact division go into v_fma_mix_f32 as they are (no float16 -> float32
//    conversions): q0 = v * r, rem = fma(-q0, s, v), q1 = fma(rem, r, q0), the same three roundings as div_f;
//  * e = v - d is taken in packed float16.  It is EXACT there: on this path d is 0 or within a factor 2 of v
//    (Sterbenz), checked for every (scale, value) pair the fast path admits by tests/tools/sterbenz_exhaustive.c
//    (tables without subnormals are the exception and take the float32 difference, see quant8_stats);
//  * sum e^2 and sum |e| go through v_dot2_f32_f16 (float16 products are exact in float32; float16 subnormals are
//    honoured; the two-term sum is not single-rounded, about one float32 ulp per step, far inside the 1e-6 bar).
__device__ __forceinline__ float mix_mul_lo(uint32_t vw, float r) {
  float d;
  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(vw), "v"(r));
  return d;
}
__device__ __forceinline__ float mix_mul_hi(uint32_t vw, float r) {
  float d;
  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(vw), "v"(r));
  return d;
}
// fma(-q0, s, v) with v the low / high float16 half of vw
__device__ __forceinline__ float mix_rem_lo(float q0, float s, uint32_t vw) {
  float d;
  asm("v_fma_mix_f32 %0, -%1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(d) : "v"(q0), "v"(s), "v"(vw));
  return d;
}
__device__ __forceinline__ float mix_rem_hi(float q0, float s, uint32_t vw) {
  float d;
  asm("v_fma_mix_f32 %0, -%1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(d) : "v"(q0), "v"(s), "v"(vw));
  return d;
}
// h(v / s) for both halves of vw (finite v; the sign of a zero quotient is not kept, see div_f)
__device__ __forceinline__ h16x2 div_pk(uint32_t vw, const SegP& sp) {
  const float q0a = mix_mul_lo(vw, sp.r), q0b = mix_mul_hi(vw, sp.r);
  const float ra = mix_rem_lo(q0a, sp.sf, vw), rb = mix_rem_hi(q0b, sp.sf, vw);
  // (written as an instruction: from a plain conversion of the two fma results the compiler forms
  //  v_fma_mixlo_f16 + v_fma_mixhi_f16, which issue at 8 cycles each against 2.6 + 2.6 + 4.2 for this form)
  uint32_t x;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(x) : "v"(__builtin_fmaf(ra, sp.r, q0a)), "v"(__builtin_fmaf(rb, sp.r, q0b)));
  return as_h2(x);
}
// h(|v| / s) for both halves of vw: div_pk with the absolute value taken by the source modifier of the mixed
// instructions (free), for the table formats, which round magnitudes
__device__ __forceinline__ uint32_t div_pk_abs(uint32_t vw, const SegP& sp) {
  float q0a, q0b, ra, rb;
  asm("v_fma_mix_f32 %0, |%1|, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(q0a) : "v"(vw), "v"(sp.r));
  asm("v_fma_mix_f32 %0, |%1|, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(q0b) : "v"(vw), "v"(sp.r));
  asm("v_fma_mix_f32 %0, -%1, %2, |%3| op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(ra) : "v"(q0a), "v"(sp.sf), "v"(vw));
  asm("v_fma_mix_f32 %0, -%1, %2, |%3| op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "v"(q0b), "v"(sp.sf), "v"(vw));
  uint32_t x;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(x) : "v"(__builtin_fmaf(ra, sp.r, q0a)), "v"(__builtin_fmaf(rb, sp.r, q0b)));
  return x;
}
// h(v / s) for both halves of vw in packed float16 arithmetic, four instructions per pair:
//   v' = v P,  q0 = v' r,  rem = fma(-q0, s', v'),  q1 = fma(rem, r, q0)      s' = s P in [1, 2), r = RN16(1 / s')
// P is a power of two, so v' / s' = v / s; it lifts the remainder out of the float16 subnormal range, where the
// fma could not hold it.  For EVERY finite float16 v and every float16 s in [2^-15, 0x70E2] with |v / s| <= 600:
// q1 has the bits of the correctly rounded quotient whenever |v / s| >= 2^-4, and rint(q1) = rint(h(v / s))
// always (below 2^-4 the remainder may be inexact and q1 a few ulp off: both round to 0) -- enumerated on the
// CPU by tests/tools/div16_exhaustive.c and on the device by mq_selfcheck_div16.  That is all the integer formats
// need from the quotient.  The sign of a zero quotient follows v.
//
// A SHARED P.  Segments that nest (a group inside a row) can divide the same v' = v P_b, P_b = 2^k_b the power of two
// of the LARGER segment's scale: with j = k - k_b in {0, 1, 2} every operand above is the own-P operand times 2^-j
// (v', s', the remainder) or 2^j (r = RN16(1 / s'): rounding commutes with a power of two), so q0 is unchanged and q1
// has the same bits wherever the smaller remainder is still exact -- |v / s| >= 2^(j - 4) <= 1/4 -- and below that
// both quotients lie under 1/2 and round to 0: rint(q1) = rint(h(v / s)) as before.  mq_selfcheck_div16_shared
// enumerates it on the device for j = 1, 2 like mq_selfcheck_div16 does for j = 0.  k_base: the offered exponent
// (MQ_PK_OWN: none); taken when 0 <= k - k_base <= 2 (sp.pk_shared), the own one otherwise.
#define MQ_PK_OWN (-100000)
__device__ __forceinline__ void pk_div_params(SegP& sp, int k_base = MQ_PK_OWN) {
  const int k_own = 1 - __builtin_amdgcn_frexp_expf(sp.sf);  // sf = m 2^e, m in [0.5, 1): s 2^k in [1, 2)
  sp.pk_shared = (unsigned)(k_own - k_base) <= 2u;
  const int k = sp.pk_shared ? k_base : k_own;
  sp.pk_k = k;
  const float s1 = __builtin_ldexpf(sp.sf, k);
  // (the packed conversion with one source in both places rounds like the scalar one, to nearest even, and leaves
  //  the value in both halves: one instruction where a conversion and a duplication were two; the reciprocal's
  //  carries the wait state a transcendental result needs, see rcp_settled)
  auto both = [](float x) __attribute__((always_inline)) {
    uint32_t r;
    asm("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(r) : "v"(x));
    return r;
  };
  sp.pk_p = both(__builtin_ldexpf(1.0f, k));  // (k <= 15 is the caller's test: s >= 2^-15)
  sp.pk_s = both(s1);
  // v_rcp_f32 is within 1 ulp of 1 / s'; for each of the 1024 values of s' every float32 within 2 ulp of 1 / s'
  // rounds to the same float16 (same test program)
  {
    const float rf = __builtin_amdgcn_rcpf(s1);
    asm("s_nop 0\n\tv_cvt_pk_f16_f32 %0, %1, %1" : "=v"(sp.pk_r) : "v"(rf));
  }
}
__device__ __forceinline__ h16x2 div_pk16_pre(uint32_t v1w, const SegP& sp) {  // v1w: v P, both halves
  const h16x2 v1 = as_h2(v1w);
  const h16x2 q0 = v1 * as_h2(sp.pk_r);
  const h16x2 rem = __builtin_elementwise_fma(-q0, as_h2(sp.pk_s), v1);
  return __builtin_elementwise_fma(rem, as_h2(sp.pk_r), q0);
}
__device__ __forceinline__ h16x2 div_pk16(uint32_t vw, const SegP& sp) {
  return div_pk16_pre(as_u32(as_h2(vw) * as_h2(sp.pk_p)), sp);
}
// |nearest table value| for the two float16 magnitudes |x| in xw (sign bits clear), for a table WITHOUT
// subnormals; vw: the weights (x has their sign: a tie goes to the smaller VALUE, i.e. down for positive and up in
// magnitude for negative x).  lut_round_pk without the sign handling and without the upper clamp: the caller
// knows |x| <= (1 + 2^-9) * largest entry, which rounds to the largest entry for up to 6 mantissa bits (the host
// sends tables with more to the general kernels).  Seven instructions per pair.
__device__ __forceinline__ uint32_t lut_round_mag_pk(uint32_t xw, uint32_t vw, const LutConst& c) {
  // t = |x| bits + (1 for negative x) + half - 1: one v_add3_u32 (no carry leaves a half: t < 0x8000)
  const uint32_t t = xw + pk_shr_u16(dup16(15u), vw) + dup16(c.half - 1u);
  const uint32_t r = pk_max_u16(t & dup16(0xffffu << c.shift), dup16(c.minpos));
  uint32_t d;  // (mid + half - 1) - t per half: negative iff |x| (+ 1) is past the 0 / minpos midpoint
  asm("v_pk_sub_i16 %0, %1, %2" : "=v"(d) : "v"(dup16(c.mid + c.half - 1u)), "v"(t));
  uint32_t m;  // (the shift count from a register: an inline constant would reach the low half only)
  asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(m) : "v"(dup16(15u)), "v"(d));
  return r & m;
}
// One lane's 8 elements, statistics only.  sse / sabs: float32 sums over these 8 elements are ADDED to the caller's
// (the caller carries everything above a few vectors in float64).
// PK16: the integer formats divide in packed float16 (div_pk16; sp must carry pk_p / pk_s / pk_r: stats_params).
// SQ: the 8 float32 squared errors (exact: a float16 error squared has 22 significant bits) also go to sq[0..7], in
// element order, for the list entries summed in numpy's order (numpy_entry).
// PRE (integer formats with PK16): pre holds the words of raw times sp's P already (a shared P, see pk_div_params).
template <int FMT, bool PK16 = false, bool SQ = false, bool PRE = false>
__device__ __forceinline__ void quant8_stats(const QuantParams& p, const u32x4 raw, const SegP& sp, float& sse,
                                             float& sabs, float* sq = nullptr, const u32x4 pre = u32x4{0, 0, 0, 0}) {
  const h16x2 s2 = as_h2(dup16(h_bits(sp.s)));
  const h16x2 magic = as_h2(0x66006600u);  // 1536
  float q = 0.f, q1 = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t vw = raw[k];
    h16x2 x, e;
    if constexpr (PRE && PK16 && !MQ_IS_LUT(FMT)) x = div_pk16_pre(pre[k], sp);
    else if constexpr (FMT != MQ_LUT_FP) x = (PK16 && !MQ_IS_LUT(FMT)) ? div_pk16(vw, sp) : div_pk(vw, sp);
    if constexpr (FMT == MQ_INT_SYM) {
      e = as_h2(vw) - ((x + magic) - magic) * s2;
    } else if constexpr (FMT == MQ_INT_ASYM) {
      // clamp(rint(x) + zp, 0, qmax) - zp = clamp(rint(x), -zp, qmax - zp): every value is an integer below 2048,
      // so the float16 additions are exact; the clamp is applied to 1536 + rint(x)
      const h16x2 zp2 = as_h2(dup16(h_bits(sp.zp)));
      const h16x2 lo = magic - zp2, hi = lo + as_h2(dup16(h_bits((half_t)p.qmax)));
      uint32_t t = pk_min_f16(pk_max_f16(as_u32(x + magic), as_u32(lo)), as_u32(hi));
      e = as_h2(vw) - (as_h2(t) - magic) * s2;
    } else if constexpr (FMT == MQ_LUT_FP) {
      // magnitudes: |e| = | |v| - |q| s |, the table value has the sign of v or is zero.  Here the error is
      // taken in float32: a negative x exactly on the midpoint between 0 and the smallest entry goes to that
      // entry (ties toward -inf), d is then 2 |v| (1 + 2^-11) and v - d needs 12 bits (13 float16 patterns per
      // scale binade for fp8_e4m3: tests/tools/sterbenz_exhaustive.c).  Plain float32 instructions issue at
      // 2.4 .. 2.6 cycles, so this costs ~3 cycles per pair more than the packed form.
      const uint32_t dm = as_u32(as_h2(lut_round_mag_pk(div_pk_abs(vw, sp), vw, p.lut)) * s2);
      float e0, e1;
      asm("v_fma_mix_f32 %0, -%1, 1.0, |%2| op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(e0) : "v"(dm), "v"(vw));
      asm("v_fma_mix_f32 %0, -%1, 1.0, |%2| op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(e1) : "v"(dm), "v"(vw));
      q = __builtin_fmaf(e0, e0, q);    // two chains of 4 steps per vector, as the dot form has
      q1 = __builtin_fmaf(e1, e1, q1);
      sabs += __builtin_fabsf(e0) + __builtin_fabsf(e1);
      if constexpr (SQ) { sq[2 * k] = e0 * e0; sq[2 * k + 1] = e1 * e1; }
      continue;
    } else {
      e = as_h2(vw) - as_h2(lut_round_pk_sub(as_u32(x), p.lut)) * s2;
    }
    q = __builtin_amdgcn_fdot2(e, e, q, false);
    sabs = __builtin_amdgcn_fdot2(as_h2(as_u32(e) & 0x7fff7fffu), as_h2(0x3c003c00u), sabs, false);
    if constexpr (SQ) {
      const float e0 = (float)e[0], e1 = (float)e[1];
      sq[2 * k] = e0 * e0;
      sq[2 * k + 1] = e1 * e1;
    }
  }
  if constexpr (FMT == MQ_LUT_FP) q += q1;
  sse += q;
}

// Fast path: packed float16 arithmetic, exact for finite data with |v / s| < 512.
//  * rint(x) for |x| < 512 is (x + 1536) - 1536 in float16 (ulp 1 in [1024, 2048), round to nearest even);
//    the low byte of the float16 bits of 1536 + q (0x6600 + q) is q's two's-complement byte.
//  * symmetric: |v / s| <= qmax * (1 + 2^-11) < qmax + 0.5, so the clamp to +-qmax never acts.
template <int FMT, int SEL>
__device__ __forceinline__ double quant8_fast(const QuantParams& p, const u32x4 raw, const SegP& sp, int64_t e0,
                                              Acc& acc) {
  if constexpr (SEL == SEL_STATS_ONLY) {
    float sse = 0.f, sabs = 0.f;
    quant8_stats<FMT>(p, raw, sp, sse, sabs);
    acc.abs += (double)sabs;
    return (double)sse;
  }
  const uint32_t s2 = dup16(h_bits(sp.s)), zp2 = dup16(h_bits(sp.zp));
  const h16x2 magic = as_h2(0x66006600u);  // 1536
  u32x4 dq, cw;
  float e[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t vw = raw[k];
    const h16x2 vp = as_h2(vw);
    const float v0 = (float)vp[0], v1 = (float)vp[1];
    // (a zero quotient may come out as +0 here whatever the sign of v: see div_f)
    const h16x2 x = {(half_t)div_f(v0, sp), (half_t)div_f(v1, sp)};
    h16x2 d;
    if constexpr (FMT == MQ_INT_SYM) {
      h16x2 t = x + magic;
      d = (t - magic) * as_h2(s2);
      // q * s carries the sign of v, also when q is zero (rint(-0.3) is -0 in the reference)
      d = as_h2((as_u32(d) & 0x7fff7fffu) | (vw & 0x80008000u));
      cw[k] = as_u32(t);
    } else if constexpr (FMT == MQ_INT_ASYM) {
      // (q - zp) * s is +0 whenever it is zero, so no sign repair is needed here
      h16x2 t = ((x + magic) - magic) + as_h2(zp2);
      uint32_t tw = pk_min_f16(pk_max_f16(as_u32(t), 0u), dup16(h_bits((half_t)p.qmax)));
      d = (as_h2(tw) - as_h2(zp2)) * as_h2(s2);
      cw[k] = tw;
    } else {
      // the table's zero is +0 and a nonzero quotient has the right sign already
      uint32_t qw = FMT == MQ_LUT_FP_SUB ? lut_round_pk_sub(as_u32(x), p.lut) : lut_round_pk(as_u32(x), p.lut);
      d = as_h2(qw) * as_h2(s2);
      cw[k] = qw;
    }
    dq[k] = as_u32(d);
    e[2 * k] = v0 - (float)d[0];
    e[2 * k + 1] = v1 - (float)d[1];
  }
  if constexpr (SEL_STORES(SEL)) {
    if (p.deq) __builtin_nontemporal_store(dq, reinterpret_cast<u32x4*>(p.deq + e0));
    if (p.codes) {
      if constexpr (MQ_IS_LUT(FMT)) {
        __builtin_nontemporal_store(cw, reinterpret_cast<u32x4*>(static_cast<uint16_t*>(p.codes) + e0));
      } else {
        if constexpr (FMT == MQ_INT_ASYM) {
          // integers 0..255 as float16: the low byte of the bits of 1024 + q is q
          const h16x2 k1024 = as_h2(0x64006400u);
#pragma unroll
          for (int k = 0; k < 4; ++k) cw[k] = as_u32(as_h2(cw[k]) + k1024);
        }
        u32x2 c = {__builtin_amdgcn_perm(cw[1], cw[0], 0x06040200u), __builtin_amdgcn_perm(cw[3], cw[2], 0x06040200u)};
        __builtin_nontemporal_store(c, reinterpret_cast<u32x2*>(static_cast<uint8_t*>(p.codes) + e0));
      }
    }
  }
  double sse = 0.0;
  if (!SEL_STORES(SEL) || p.need_err) {
    if constexpr (SEL_STORES(SEL)) {
      if (p.abs_err) store_abs_err(p, e, e0, acc.xp);
    }
    error_sums(e, sse, acc.abs);
    if constexpr (SEL_FEEDS(SEL)) sel_feed8<SEL>(acc.sel, e, e0);
  }
  return sse;
}

// Slow path: the contract element by element (any float16 input, including inf / NaN / huge v / s).
template <int FMT, int SEL>
__device__ __forceinline__ double quant8_slow(const QuantParams& p, const u32x4 raw, const SegP& sp, int64_t e0,
                                           Acc& acc) {
  const half_t qmax_h = (half_t)p.qmax;
  u32x4 dq, c16;
  uint32_t c8[2] = {0u, 0u};
  float e[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint32_t vb = (i & 1) ? (raw[i >> 1] >> 16) : (raw[i >> 1] & 0xffffu);
    float vf = (float)h_from_bits(vb);
    half_t x = div_h_full(vf, sp), d;
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
      code = p.lut.sub ? h_bits((half_t)lut_round_sub((float)x, p.lut))
                       : lut_round_bits(h_bits(x), 0x8000u, 0x7c00u, p.lut);
      d = h_from_bits(code) * sp.s;
    }
    uint32_t db = h_bits(d);
    if (i & 1) dq[i >> 1] |= db << 16; else dq[i >> 1] = db;
    if constexpr (MQ_IS_LUT(FMT)) {
      if (i & 1) c16[i >> 1] |= code << 16; else c16[i >> 1] = code;
    } else {
      c8[i >> 2] |= code << (8 * (i & 3));
    }
    e[i] = vf - (float)d;
  }
  if constexpr (SEL_STORES(SEL)) {
    if (p.deq) *reinterpret_cast<u32x4*>(p.deq + e0) = dq;
    if (p.codes) {
      if constexpr (MQ_IS_LUT(FMT)) {
        *reinterpret_cast<u32x4*>(static_cast<uint16_t*>(p.codes) + e0) = c16;
      } else {
        u32x2 c = {c8[0], c8[1]};
        *reinterpret_cast<u32x2*>(static_cast<uint8_t*>(p.codes) + e0) = c;
      }
    }
  }
  double sse = 0.0;
  if (!SEL_STORES(SEL) || p.need_err) {
    if constexpr (SEL_STORES(SEL)) {
      if (p.abs_err) store_abs_err(p, e, e0, acc.xp);
    }
    error_sums(e, sse, acc.abs);
    if constexpr (SEL_FEEDS(SEL)) sel_feed8<SEL>(acc.sel, e, e0);
    if constexpr (SEL == SEL_STATS_ONLY) {
      if (acc.sq) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc.sq[i] = e[i] * e[i];
      }
    }
  }
  return sse;
}

template <int FMT, int SEL>
__device__ __forceinline__ double quant8(const QuantParams& p, const u32x4 raw, const SegP& sp, int64_t e0,
                                         Acc& acc) {
  if (__builtin_expect(sp.slow, 0)) return quant8_slow<FMT, SEL>(p, raw, sp, e0, acc);
  return quant8_fast<FMT, SEL>(p, raw, sp, e0, acc);
}

// quant_datatype's per-tensor float32 branch (quant_weight.py:562-575): x = w32 / s in float32, table
// lookup in float32, (q * s) rounded to float32 and then to float16.
__device__ __forceinline__ double quant8_lut_f32(const QuantParams& p, const float* w32, u32x4 raw, float s,
                                                 int64_t e0, Acc& acc) {
  u32x4 dq, c16;
  float e[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float x = w32[i] / s;
    float q = p.lut.sub ? lut_round_sub(x, p.lut)
                        : f_from_bits(lut_round_bits(f_bits(x), 0x80000000u, 0x7f800000u, p.lut));
    float t = q * s;
    uint32_t db = f2h_bits(t), cb = f2h_bits(q);
    if (i & 1) { dq[i >> 1] |= db << 16; c16[i >> 1] |= cb << 16; } else { dq[i >> 1] = db; c16[i >> 1] = cb; }
    uint32_t vb = (i & 1) ? (raw[i >> 1] >> 16) : (raw[i >> 1] & 0xffffu);
    e[i] = (float)h_from_bits(vb) - (float)h_from_bits(db);
  }
  if (p.deq) *reinterpret_cast<u32x4*>(p.deq + e0) = dq;
  if (p.codes) *reinterpret_cast<u32x4*>(static_cast<uint16_t*>(p.codes) + e0) = c16;
  double sse = 0.0;
  if (p.need_err) {
    if (p.abs_err) store_abs_err(p, e, e0, acc.xp);
    error_sums(e, sse, acc.abs);
  }
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
    p.partials[2 * (int64_t)p.bx] = a;
    p.partials[2 * (int64_t)p.bx + 1] = b;
  }
}

// ------------------------------------------------------------------------------------------------
// SC_SUBWAVE: a segment is T <= 64 lanes of one wave.  A wave streams NV x 1 KiB per iteration; the NV
// vectors a lane holds belong to NV different segments.
// ------------------------------------------------------------------------------------------------
template <int FMT, int T, int NV, int SEL>
__global__ __launch_bounds__(256) void k_quant_team(const QuantBatch b) {
  const QuantParams p = batch_params(b);
  __shared__ double lds[32];
  double acc_sse = 0.0;
  __shared__ __attribute__((aligned(16))) float xpose[4 * 512];
  Acc acc;
  acc.abs = 0.0;
  acc.xp = (p.abs_err && !p.abs_err_f16) ? xpose + (threadIdx.x >> 6) * 512 : nullptr;
  if constexpr (SEL_FEEDS(SEL)) sel_feed_init<SEL>(acc.sel, p.sel_ws, sel_lds, p.sel_samples);
  const int lane = threadIdx.x & 63;
  const int tl = lane & (T - 1), team = lane / T;
  constexpr int SPW = 64 / T;  // segments per wave per vector slot
  const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int64_t tiles = (p.nseg + (int64_t)SPW * NV - 1) / ((int64_t)SPW * NV);
  // sampling pass of a selection: ONE of the NV vector slots (SPW whole segments, 512 consecutive elements
  // when the teams are full) of one pseudo-randomly chosen tile out of every 8 FULL tiles, so that the
  // samples are dense and come from 4 x more places than whole sampled tiles would
  constexpr bool samp = SEL == SEL_FEED_SAMPLE;
  static_assert(NV == 4, "the sampling pattern assumes 4 vector slots per tile");
  const int64_t steps = samp ? (p.nseg / ((int64_t)SPW * NV)) / 8 : tiles;
  auto tile_of = [&](int64_t t) { return samp ? t * 8 + (int64_t)(sample_line((uint32_t)t) & 7u) : t; };
  auto slot_of = [&](int64_t t) { return (int)(sample_line((uint32_t)t) >> 3); };
  const bool lane_in_seg = tl < p.vecs_per_seg;
  // Software pipeline: the next tile's loads are issued before the current tile is computed, so a wave
  // keeps 2 x NV KiB in flight.
  // (the sampling pass holds only its one slot: NJ = 1, slot index js)
  constexpr int NJ = samp ? 1 : NV;
  u32x4 nxt[NJ];
  auto issue = [&](int64_t t, u32x4* dst) {
    const int64_t tile = tile_of(t);
    const int js = samp ? slot_of(t) : 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      int64_t sg = (tile * NV + (samp ? js : j)) * SPW + team;
      dst[j] = (lane_in_seg && sg < p.nseg) ? load8_raw(p.w, p.in_dtype, sg * p.L + (int64_t)tl * 8)
                                            : u32x4{0, 0, 0, 0};
    }
  };
  if (gw < steps) issue(gw, nxt);
  for (int64_t t = gw; t < steps; t += nwaves) {
    const int64_t tile = tile_of(t);
    const int js = samp ? slot_of(t) : 0;
    // dense sample index of an element = its offset inside the sampled slot + t * (slot size)
    if constexpr (samp) acc.sel.sample_shift = (t - (tile * NV + js)) * (int64_t)SPW * p.L;
    u32x4 raw[NJ];
    bool valid[NJ];
    int64_t seg[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      raw[j] = raw_to_f16(nxt[j], p.in_dtype);
      seg[j] = (tile * NV + (samp ? js : j)) * SPW + team;
      valid[j] = lane_in_seg && seg[j] < p.nseg;
    }
    if (t + nwaves < steps) issue(t + nwaves, nxt);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      SegP sp;
      uint32_t zpb = 0;
      if constexpr (FMT == MQ_INT_ASYM) {
        uint32_t lo, hi;
        minmax8_pk(raw[j], lo, hi);
        uint32_t mm = valid[j] ? fold_minmax(lo, hi) : 0xfc007c00u;  // (+inf, -inf): neutral
        mm = team_reduce_u32<T>(mm, op_minmax);
        sp = params_minmax(mm, p);
        zpb = h_bits(sp.zp);
        if constexpr (T <= 8) {
          if (p.zp) zpb = zp_bits_first_zero<T, 1>(sp, mm, &raw[j], tl, 0, valid[j], p.L);
        }
      } else {
        uint32_t m = fold_max_u16(absmax8_pk(raw[j]));
        sp = params_absmax(team_reduce_u32<T>(m, op_max_u32), p);
      }
      double s = 0.0;
      const int64_t e0 = seg[j] * p.L + (int64_t)tl * 8;
      if (valid[j]) s = quant8<FMT, SEL>(p, raw[j], sp, e0, acc);
      acc_sse += s;
      if (p.seg_sse) {
        s = team_sum<T>(s);
        if (valid[j] && tl == 0) store_seg(p, seg[j], s);
      }
      if (valid[j] && tl == 0) {
        if (p.scale) p.scale[seg[j]] = (uint16_t)h_bits(sp.s);
        if (FMT == MQ_INT_ASYM && p.zp) p.zp[seg[j]] = (uint16_t)zpb;
      }
    }
  }
  block_partials(p, acc_sse, acc.abs, lds);
  if constexpr (SEL_FEEDS(SEL)) sel_feed_flush<SEL>(acc.sel);
}

// ------------------------------------------------------------------------------------------------
// SC_SUBWAVE, statistics only, L = 32 * T (T = 1 .. 16 lanes): the 4 vectors a lane holds belong to ONE segment
// (vector j * T + tl of it), so the reduction, the segment's parameters and the per-segment sum are paid once per
// 4 vectors instead of once per vector -- the group sweeps are bound by instruction issue, not by memory.  A load
// instruction reads runs of 16 * T bytes; the other three cover the gaps right after it.  (Not used when
// element-sized outputs are stored: their stores would be split the same way.)
// ------------------------------------------------------------------------------------------------
template <int FMT, int T>
__global__ __launch_bounds__(256) void k_quant_team4(const QuantBatch b) {
  constexpr int NV = 4, SEL = SEL_STATS_ONLY;
  const QuantParams p = batch_params(b);
  __shared__ double lds[32];
  double acc_sse = 0.0;
  Acc acc;
  acc.abs = 0.0;
  acc.xp = nullptr;
  const int lane = threadIdx.x & 63;
  const int tl = lane & (T - 1), team = lane / T;
  constexpr int SPW = 64 / T;  // segments per wave and step
  const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int64_t steps = (p.nseg + SPW - 1) / SPW;
  u32x4 nxt[NV];
  auto issue = [&](int64_t t, u32x4* dst) {
    const int64_t sg = t * SPW + team;
#pragma unroll
    for (int j = 0; j < NV; ++j)
      dst[j] = sg < p.nseg ? load8_raw(p.w, p.in_dtype, sg * p.L + (int64_t)(j * T + tl) * 8) : u32x4{0, 0, 0, 0};
  };
  if (gw < steps) issue(gw, nxt);
  for (int64_t t = gw; t < steps; t += nwaves) {
    const int64_t seg = t * SPW + team;
    const bool valid = seg < p.nseg;
    u32x4 raw[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) raw[j] = raw_to_f16(nxt[j], p.in_dtype);
    if (t + nwaves < steps) issue(t + nwaves, nxt);
    SegP sp;
    uint32_t zpb = 0;
    if constexpr (FMT == MQ_INT_ASYM) {
      uint32_t mm = 0xfc007c00u;  // (+inf, -inf): neutral
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        uint32_t lo, hi;
        minmax8_pk(raw[j], lo, hi);
        if (valid) mm = op_minmax(mm, fold_minmax(lo, hi));
      }
      mm = team_reduce_u32<T>(mm, op_minmax);
      sp = params_minmax(mm, p);
      zpb = h_bits(sp.zp);
      if constexpr (T == 1) {  // (32 elements: the lane's four vectors in memory order)
        if (p.zp) zpb = zp_bits_first_zero<1, NV>(sp, mm, raw, 0, 1, valid, p.L);
      }
    } else {
      uint32_t m2 = 0;
#pragma unroll
      for (int j = 0; j < NV; ++j) m2 = pk_max_u16(m2, absmax8_pk(raw[j]));
      sp = params_absmax(team_reduce_u32<T>(fold_max_u16(m2), op_max_u32), p);
    }
    double s = 0.0;
    if (valid) {
#pragma unroll
      for (int j = 0; j < NV; ++j) s += quant8<FMT, SEL>(p, raw[j], sp, seg * p.L + (int64_t)(j * T + tl) * 8, acc);
    }
    acc_sse += s;
    if (p.seg_sse) {
      s = team_sum<T>(s);
      if (valid && tl == 0) store_seg(p, seg, s);
    }
    if (valid && tl == 0) {
      if (p.scale) p.scale[seg] = (uint16_t)h_bits(sp.s);
      if (FMT == MQ_INT_ASYM && p.zp) p.zp[seg] = (uint16_t)zpb;
    }
  }
  block_partials(p, acc_sse, acc.abs, lds);
}

// ------------------------------------------------------------------------------------------------
// SC_WAVE / SC_BLOCK: a segment (a row, or a long group) is held in the registers of one wave or one
// workgroup, NV vectors per lane, interleaved so that every load instruction is 1 KiB contiguous.
// ------------------------------------------------------------------------------------------------
template <int FMT, int NV, bool BLOCK, int SEL>
__device__ __forceinline__ void row_body(const QuantParams& p, double* lds, double& acc_sse, Acc& acc) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int64_t first = BLOCK ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * nw + wv;
  const int64_t stride = BLOCK ? (int64_t)gridDim.x : (int64_t)gridDim.x * nw;
  const int tid = BLOCK ? (int)threadIdx.x : lane;
  const int tcount = BLOCK ? (int)blockDim.x : 64;
  uint32_t* lds_u = reinterpret_cast<uint32_t*>(lds + 32);  // [2][16] reduction exchange (BLOCK only)
  int parity = 0;
  // Software pipeline (NV <= 8): the next segment's loads are issued before this one is computed.
  constexpr bool PREFETCH = NV <= 8;
  u32x4 nxt[PREFETCH ? NV : 1];
  auto issue = [&](int64_t sg, u32x4* dst) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      int v = j * tcount + tid;
      dst[j] = v < p.vecs_per_seg ? load8_raw(p.w, p.in_dtype, sg * p.L + (int64_t)v * 8) : u32x4{0, 0, 0, 0};
    }
  };
  // sampling pass of a selection: one pseudo-randomly chosen segment of every 32
  constexpr bool samp = SEL == SEL_FEED_SAMPLE;
  const int64_t steps = samp ? p.nseg / 32 : p.nseg;
  auto seg_of = [&](int64_t t) { return samp ? t * 32 + (int64_t)sample_line((uint32_t)t) : t; };
  if constexpr (PREFETCH) {
    if (first < steps) issue(seg_of(first), nxt);
  }
  for (int64_t t = first; t < steps; t += stride) {
    const int64_t seg = seg_of(t);
    const int64_t base = seg * p.L;
    if constexpr (SEL == SEL_FEED_SAMPLE) acc.sel.sample_shift = (t - seg) * p.L;
    u32x4 raw[NV];
    bool valid[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) valid[j] = j * tcount + tid < p.vecs_per_seg;
    if constexpr (PREFETCH) {
#pragma unroll
      for (int j = 0; j < NV; ++j) raw[j] = raw_to_f16(nxt[j], p.in_dtype);
      if (t + stride < steps) issue(seg_of(t + stride), nxt);
    } else {
      issue(seg, raw);
#pragma unroll
      for (int j = 0; j < NV; ++j) raw[j] = raw_to_f16(raw[j], p.in_dtype);
    }
    SegP sp;
    if constexpr (FMT == MQ_INT_ASYM) {
      uint32_t mm = 0xfc007c00u;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        uint32_t lo, hi;
        minmax8_pk(raw[j], lo, hi);
        if (valid[j]) mm = op_minmax(mm, fold_minmax(lo, hi));
      }
      mm = team_reduce_u32<64>(mm, op_minmax);
      if constexpr (BLOCK) {
        if (lane == 0) lds_u[parity * 16 + wv] = mm;
        __syncthreads();
        for (int i = 0; i < nw; ++i) mm = op_minmax(mm, lds_u[parity * 16 + i]);
      }
      sp = params_minmax(mm, p);
    } else {
      uint32_t m2 = 0;
#pragma unroll
      for (int j = 0; j < NV; ++j) m2 = pk_max_u16(m2, absmax8_pk(raw[j]));
      uint32_t m = team_reduce_u32<64>(fold_max_u16(m2), op_max_u32);
      if constexpr (BLOCK) {
        if (lane == 0) lds_u[parity * 16 + wv] = m;
        __syncthreads();
        for (int i = 0; i < nw; ++i) m = op_max_u32(m, lds_u[parity * 16 + i]);
      }
      sp = params_absmax(m, p);
    }
    double s = 0.0;
    if (__builtin_expect(sp.slow, 0)) {
      // The per-element path is the same for the whole segment.  It reloads its vectors (cache hits) in a rolled
      // loop: indexing raw[] with a run-time j would move the register array to scratch memory for both paths.
#pragma unroll 1
      for (int j = 0; j < NV; ++j) {
        const int v = j * tcount + tid;
        if (v < p.vecs_per_seg) {
          const int64_t e0 = base + (int64_t)v * 8;
          s += quant8_slow<FMT, SEL>(p, load8_f16(p.w, p.in_dtype, e0), sp, e0, acc);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j)
        if (valid[j]) s += quant8_fast<FMT, SEL>(p, raw[j], sp, base + (int64_t)(j * tcount + tid) * 8, acc);
    }
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
          store_seg(p, seg, t);
        }
      } else if (lane == 0) {
        store_seg(p, seg, s);
      }
    }
    if (tid == 0) {
      if (p.scale) p.scale[seg] = (uint16_t)h_bits(sp.s);
      if (FMT == MQ_INT_ASYM && p.zp) p.zp[seg] = (uint16_t)h_bits(sp.zp);
    }
    parity ^= 1;
  }
}

template <int FMT, int NV, int SEL>
__global__ __launch_bounds__(256) void k_quant_wave(const QuantBatch b) {
  const QuantParams p = batch_params(b);
  __shared__ double lds[32];
  __shared__ __attribute__((aligned(16))) float xpose[4 * 512];
  double acc_sse = 0.0;
  Acc acc;
  acc.abs = 0.0;
  acc.xp = (p.abs_err && !p.abs_err_f16) ? xpose + (threadIdx.x >> 6) * 512 : nullptr;
  if constexpr (SEL_FEEDS(SEL)) sel_feed_init<SEL>(acc.sel, p.sel_ws, sel_lds, p.sel_samples);
  row_body<FMT, NV, false, SEL>(p, lds, acc_sse, acc);
  block_partials(p, acc_sse, acc.abs, lds);
  if constexpr (SEL_FEEDS(SEL)) sel_feed_flush<SEL>(acc.sel);
}

template <int FMT, int NV, int THREADS, int SEL>
__global__ __launch_bounds__(THREADS) void k_quant_block(const QuantBatch b) {
  const QuantParams p = batch_params(b);
  __shared__ double lds[96];
  __shared__ __attribute__((aligned(16))) float xpose[(THREADS / 64) * 512];
  double acc_sse = 0.0;
  Acc acc;
  acc.abs = 0.0;
  acc.xp = (p.abs_err && !p.abs_err_f16) ? xpose + (threadIdx.x >> 6) * 512 : nullptr;
  if constexpr (SEL_FEEDS(SEL)) sel_feed_init<SEL>(acc.sel, p.sel_ws, sel_lds, p.sel_samples);
  row_body<FMT, NV, true, SEL>(p, lds, acc_sse, acc);
  block_partials(p, acc_sse, acc.abs, lds);
  if constexpr (SEL_FEEDS(SEL)) sel_feed_flush<SEL>(acc.sel);
}

// ------------------------------------------------------------------------------------------------
// Statistics-only kernels of the model sweeps: float16 / bfloat16 weights, 16-byte aligned, no element-sized
// output.  Same arithmetic as the kernels above (quant8_stats, quant8_slow for segments off the fast path), with
// everything that is not arithmetic taken out of the loop, because these passes are bound by instruction issue:
// a wave's position is wave-uniform (scalar base address + one per-lane byte offset + immediate offsets), the
// input type is a template parameter, and whole steps carry no per-lane predicates.
//   k_stats_group<FMT, BF16, T>     groups of 32 * T elements (T = 1 .. 16 lanes per group, 4 vectors per lane)
//   k_stats_row<FMT, BF16, NV, 1>   rows of up to 512 * NV elements, one wave per row
//   k_stats_row<FMT, BF16, NV, 4>   rows of up to 2048 * NV elements, one 256-thread workgroup per row
// ------------------------------------------------------------------------------------------------
template <bool BF16>
__device__ __forceinline__ u32x4 stats_cvt(u32x4 raw) {
  if constexpr (BF16) return bf16x8_to_f16(raw);
  else return raw;
}
// NT: non-temporal load.  The weights are read once; for whole 1 KiB runs per wave instruction (rows) a
// non-temporal read measured 7.0 against 6.3 TB/s.  The group kernels' instructions read runs of 16 * T bytes and
// four of them share every cache line: there the line must stay in the vector cache (plain loads; non-temporal
// ones took g = 32 from 123 to 159 us per 16 x 4096^2).
template <bool NT>
__device__ __forceinline__ u32x4 stats_load(const char* ptr) {
  if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ptr));
  else return *reinterpret_cast<const u32x4*>(ptr);
}
// The statistics kernels read the weights through a raw buffer descriptor (base, size in bytes): a load whose
// offset lies past the size returns zeros and touches no memory.  A lane past the end of a ragged row or of the
// last partial step, and a whole prefetch past the wave's last step, are such loads -- the load instructions
// stay unconditional.  (A branch around a load makes the compiler wait for EVERY outstanding memory operation
// where the paths merge, s_waitcnt vmcnt(0), the prefetch just issued included; with straight-line loads it
// counts: vmcnt(7), (6), ... as the vectors of the older buffer are consumed.)
#define MQ_STATS_OOB 0xffff0000u   // an offset past every admitted tensor (the host keeps them below 0xf0000000)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t stats_rsrc(const QuantParams& p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)(uint32_t)(p.n * 2), 0x00020000);
}
template <bool NT>
__device__ __forceinline__ u32x4 stats_bload(__amdgpu_buffer_rsrc_t rs, uint32_t off) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, NT ? 2 : 0));
}
// One lane's reduction word over NV vectors: absmax bits, or (min | max << 16).  v_pk_maximum3 / minimum3_f16 take
// two new words per instruction and, unlike v_pk_max / min_f16, return a NaN when an operand is one (a NaN in the
// segment must reach the parameters: torch.amax / amin propagate it).
template <int FMT, int NV>
__device__ __forceinline__ uint32_t stats_lane_reduce(const u32x4* raw) {
  uint32_t mx = raw[0][0], mn = raw[0][0];
#pragma unroll
  for (int i = 1; i + 1 < 4 * NV; i += 2) {
    mx = pk_maximum3_f16(mx, raw[i >> 2][i & 3], raw[(i + 1) >> 2][(i + 1) & 3]);
    mn = pk_minimum3_f16(mn, raw[i >> 2][i & 3], raw[(i + 1) >> 2][(i + 1) & 3]);
  }
  const uint32_t last = raw[NV - 1][3];  // 4 NV - 1 words follow the first: an odd count
  mx = pk_maximum3_f16(mx, last, last);
  mn = pk_minimum3_f16(mn, last, last);
  if constexpr (FMT == MQ_INT_ASYM) {
    // (mn.l, mx.l) vs (mn.h, mx.h): min of the low halves, max of the high halves
    const uint32_t a = (mn & 0xffffu) | (mx << 16), c = (mn >> 16) | (mx & 0xffff0000u);
    return (pk_minimum3_f16(a, c, c) & 0xffffu) | (pk_maximum3_f16(a, c, c) & 0xffff0000u);
  } else {
    return fold_max_u16(pk_max_u16(mx & 0x7fff7fffu, mn & 0x7fff7fffu));  // integer max: a NaN wins
  }
}
template <int FMT>
__device__ __forceinline__ uint32_t stats_op(uint32_t a, uint32_t b) {
  if constexpr (FMT == MQ_INT_ASYM) return op_minmax(a, b);
  else return op_max_u32(a, b);
}
// params_absmax / params_minmax with the common case in few instructions: h(a / qmax) by the reciprocal multiply
// in mixed precision when the host has verified it for this qmax, the clamp as v_max / v_min_f16.
// k_base: a shared power of two offered to the packed division (pk_div_params).
template <int FMT>
__device__ __forceinline__ SegP stats_params(uint32_t red, const QuantParams& p, int k_base = MQ_PK_OWN) {
  if (!p.qmax_recip_ok) {
    SegP sp;
    if constexpr (FMT == MQ_INT_ASYM) sp = params_minmax(red, p);
    else sp = params_absmax(red, p);
    if constexpr (!MQ_IS_LUT(FMT)) {
      sp.slow = sp.slow || sp.sf < 0x1p-15f;
      pk_div_params(sp, k_base);
    }
    return sp;
  }
  const half_t lo = h_from_bits(0x00A8u), hi = h_from_bits(0x70E2u);
  SegP sp;
  if constexpr (FMT == MQ_INT_ASYM) {
    const half_t rmin = h_from_bits(red & 0xffffu), rmax = h_from_bits(red >> 16);
    const uint32_t range = h_bits(rmax - rmin);
    const float q0 = mix_mul_lo(range, p.rqmax);
    const half_t s0 = (half_t)__builtin_fmaf(mix_rem_lo(q0, p.qmax, range), p.rqmax, q0);
    finish_params(sp, __builtin_fminf16(__builtin_fmaxf16(s0, lo), hi));
    const half_t qmax_h = (half_t)p.qmax;
    half_t z = __builtin_rintf16(div_h_full(-(float)rmin, sp));  // (a -0 stays -0, as torch.clamp leaves it)
    z = z < (half_t)0 ? (half_t)0 : z;
    sp.zp = z > qmax_h ? qmax_h : z;
    const float big = __builtin_fmaxf(__builtin_fabsf((float)rmin), __builtin_fabsf((float)rmax));
    // the fast path needs finite data, |v / s| < 512, a scale below the upper clamp and -- for the exact float16
    // error of quant8_stats -- a row that does not lie on one side of zero (then the clamp to [0, qmax] removes
    // at most one step from a code, see tests/tools/sterbenz_exhaustive.c)
    sp.slow = !(big * sp.r < 500.0f) || !((float)h_from_bits(range) < 65504.0f) || s0 > hi || !(s0 == s0) ||
              (float)rmin > 0.f || (float)rmax < 0.f || sp.sf < 0x1p-15f;
    if (sp.slow) { sp = params_minmax(red, p); sp.slow = true; }
    else pk_div_params(sp, k_base);
  } else {
    const float q0 = mix_mul_lo(red, p.rqmax);
    const half_t s0 = (half_t)__builtin_fmaf(mix_rem_lo(q0, p.qmax, red), p.rqmax, q0);
    finish_params(sp, __builtin_fminf16(__builtin_fmaxf16(s0, lo), hi));
    sp.zp = (half_t)0;
    // inf / NaN in the segment, a scale cut by the upper clamp (then |v / s| may pass qmax), or one below the
    // range of the packed division
    sp.slow = red >= 0x7c00u || s0 > hi || sp.sf < 0x1p-15f;
    if (sp.slow) { sp = params_absmax(red, p); sp.slow = true; }
    else if constexpr (!MQ_IS_LUT(FMT)) pk_div_params(sp, k_base);
  }
  return sp;
}
// NV vectors of one segment held by this lane (fast path) -> float64 sum of squared errors, sum of |error|
template <int FMT, int NV, bool SQ = false, bool PRE = false>
__device__ __forceinline__ void stats_vectors(const QuantParams& p, const u32x4* raw, const SegP& sp, double& sse,
                                              double& sabs, float (*sq)[8] = nullptr, const u32x4* pre = nullptr) {
  // float32 within two vectors (squares: 8 dot steps + 1 add) resp. four vectors (|error|), float64 above
  sse = 0.0;
  sabs = 0.0;
  float q = 0.f, ab = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    if constexpr (SQ) quant8_stats<FMT, true, true>(p, raw[j], sp, q, ab, sq[j]);
    else if constexpr (PRE) quant8_stats<FMT, true, false, true>(p, raw[j], sp, q, ab, nullptr, pre[j]);
    else quant8_stats<FMT, true>(p, raw[j], sp, q, ab);
    if ((j & 1) == 1 || j == NV - 1) { sse += (double)q; q = 0.f; }
    if ((j & 3) == 3 || j == NV - 1) { sabs += (double)ab; ab = 0.f; }
    // one vector's arithmetic at a time: interleaving all NV of them costs 100+ registers of temporaries
    __builtin_amdgcn_sched_barrier(0);
  }
}
// The float32 sum of a segment's squared errors in the order numpy's pairwise summation takes for a contiguous
// run of n <= 128 elements (n a multiple of 8) -- what np.mean(diff * diff) of compute_mse adds up
// (quant_error_single_weight.py:46-48; numpy/_core/src/umath/loops_utils.h.src, pairwise sum): eight accumulators
// r[i] = a[i], then r[i] += a[8 k + i] for k = 1, 2, ... in order, then ((r0 + r1) + (r2 + r3)) + ((r4 + r5) +
// (r6 + r7)).  sq: this lane's four vectors of the segment, in memory order; the T lanes of a team hold
// consecutive quarters of it, so the chain runs through the lanes one after the other: in round q lane q of the
// team continues from what lane q - 1 handed over (a quad-permute), and after T rounds lane T - 1 holds the
// accumulators of the whole segment (the other lanes' values are not used).  0 + a = a exactly, so the first
// lane may start from zeros.  Valid in the team's LAST lane.
template <int T>
__device__ __forceinline__ float numpy_entry(const float (*sq)[8]) {
  static_assert(T == 1 || T == 2 || T == 4, "segments of 32, 64 or 128 elements");
  float in[8], r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) in[i] = 0.f;
#pragma unroll
  for (int q = 0; q < T; ++q) {
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (((in[i] + sq[0][i]) + sq[1][i]) + sq[2][i]) + sq[3][i];
    if (q + 1 < T) {
#pragma unroll
      for (int i = 0; i < 8; ++i)  // lane l takes r of lane l - 1 (quad_perm [0, 0, 1, 2])
        in[i] = f_from_bits((uint32_t)__builtin_amdgcn_update_dpp(0, (int)f_bits(r[i]), 0x90, 0xf, 0xf, false));
    }
  }
  return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
}

// Segments off the fast path (inf / NaN, a clamped scale, asymmetric rows on one side of zero ...) are rare and
// their per-element code is register-hungry.  The main loops skip them (no stores, nothing added to the sums) and
// remember that one was seen; a second loop after the main one -- whose registers the prefetch buffers no longer
// occupy -- walks the same segments again and handles only those.  count vectors, stride bytes apart, starting
// at first.
template <int FMT>
__device__ __forceinline__ void stats_slow_segment(const QuantParams& p, const SegP& sp, const char* first, int count,
                                                   int stride, double& sse, double& sabs, float (*sq)[8] = nullptr) {
  Acc acc;
  acc.abs = 0.0;
  acc.xp = nullptr;
  sse = 0.0;
#pragma unroll 1
  for (int j = 0; j < count; ++j) {
    float tmp[8];
    acc.sq = sq ? tmp : nullptr;
    sse += quant8_slow<FMT, SEL_STATS_ONLY>(p, load8_f16(first + (int64_t)j * stride, p.in_dtype, 0), sp, 0, acc);
    if (sq) {  // (j is not a compile-time index: spread the copy over the four rows)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        if (jj == j) {
#pragma unroll
          for (int i = 0; i < 8; ++i) sq[jj][i] = tmp[i];
        }
    }
  }
  sabs = acc.abs;
}

// EXACT: the per-segment output is the float32 mean squared error summed in numpy's order (numpy_entry), written
// by the team's last lane; T <= 4 (segments of up to 128 elements).
template <int FMT, bool BF16, int T, bool EXACT = false>
__global__ __launch_bounds__(256) void k_stats_group(const QuantBatch b) {
  constexpr int NV = 4, SPW = 64 / T;  // segments per wave and step: 4 KiB of weights
  const QuantParams p = batch_params(b);
  __shared__ double lds[32];
  const int lane = threadIdx.x & 63, tl = lane & (T - 1), team = lane / T;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t gw = (int64_t)p.bx * 4 + wv, nwaves = (int64_t)p.gx * 4;
  // The percentile of a large tensor's list is selected inside the pass (mq_sel.h, ListFeedState): the sampling
  // launch walks one pseudo-randomly chosen whole step of every p.sunit and stores those entries densely; the pass
  // itself counts the entries under the tensor's bracket and keeps the bit patterns inside it instead of writing
  // the list; the repair launch writes the list of a tensor whose bracket missed.  (Uniform over the workgroup.)
  const int list_mode = EXACT ? (int)LIST_OFF : p.list_mode;
  if (list_mode == LIST_SAMPLE && !p.sample) return;
  if (list_mode == LIST_REPAIR && !(p.feed && p.feed->miss)) return;
  const bool sampling = list_mode == LIST_SAMPLE;
  const bool feeding = list_mode == LIST_FEED && p.feed != nullptr;
  const uint32_t blo = p.blo, bhi = p.bhi;
  // every WAVE keeps its own slot and counts (wave-uniform registers: no shared counter, nothing to wait for)
  uint32_t* const slot = feeding ? p.slots + ((size_t)p.bx * 4 + wv) * LIST_SLOT : nullptr;
  uint32_t nbelow = 0, nkept = 0;
  const int64_t full = sampling ? (int64_t)p.scount : p.nseg / SPW;
  const int64_t steps = sampling ? (int64_t)p.scount : (p.nseg + SPW - 1) / SPW;
  // the wave step behind loop index t: itself, or -- sampling -- the chosen step of unit t
  auto step_of = [&](int64_t t) __attribute__((always_inline)) {
    return sampling ? t * (int64_t)p.sunit + (int64_t)(((uint32_t)t * 0x9E3779B1u >> 8) % p.sunit) : t;
  };
  const char* wb = static_cast<const char*>(p.w);
  // A lane ends up holding the 64-byte run `lane` of the step's 4 KiB (four vectors of one segment; a team is T
  // consecutive runs).  Loaded that way, each load instruction would touch all 32 cache lines of the 4 KiB and
  // the four of them share every line, which rules out non-temporal loads (measured: 123 -> 159 us).  Instead
  // load instruction j reads the consecutive KiB j, lane a + 16 i taking vector 4 a + i of it -- vector i of run
  // 16 j + a -- and a 4 x 4 transpose between (lane / 16) and j, two v_permlane32_swap + two v_permlane16_swap per
  // register column, hands every lane its run (whole-KiB non-temporal reads: 7.1 against 6.3 TB/s).
  // (The table formats are bound by instruction issue, not by the loads: for them the 16 extra instructions cost
  //  more than the better read pattern returns -- fp8_e4m3 g = 128: 117 -> 130 us -- so they keep the direct layout:
  //  lane tl of a team holds vectors tl, tl + T, tl + 2 T, tl + 3 T of its segment, plain loads.)
  constexpr bool TR = !MQ_IS_LUT(FMT) || EXACT;  // (numpy's order needs a lane's vectors in memory order)
  const uint32_t loff = TR ? (uint32_t)((4 * (lane & 15) + (lane >> 4)) * 16) : (uint32_t)(team * (64 * T) + tl * 16);
  double acc_sse = 0.0, acc_abs = 0.0;
  bool seen_slow = false;
  const __amdgpu_buffer_rsrc_t rs = stats_rsrc(p);
  // the wave's 4 KiB of step t, untransposed; zeros for runs past the last segment and for a step past the last one
  auto issue = [&](int64_t t, u32x4* dst) __attribute__((always_inline)) {
    const uint32_t base = (uint32_t)(step_of(t) * 4096) + loff;
    if constexpr (!TR) {
      const bool in = t < full || t * SPW + team < p.nseg;
#pragma unroll
      for (int j = 0; j < NV; ++j) dst[j] = stats_bload<false>(rs, in ? base + j * (16 * T) : MQ_STATS_OOB);
    } else if (t < full) {
#pragma unroll
      for (int j = 0; j < NV; ++j) dst[j] = stats_bload<true>(rs, base + j * 1024);
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const bool in = t * SPW + (16 * j + (lane & 15)) / T < p.nseg;  // the segment of run 16 j + a
        dst[j] = stats_bload<true>(rs, in ? base + j * 1024 : MQ_STATS_OOB);
      }
    }
  };
  auto transpose = [&](u32x4* r) __attribute__((always_inline)) {
    if constexpr (!TR) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      auto a = __builtin_amdgcn_permlane32_swap(r[0][k], r[2][k], false, false);
      auto c = __builtin_amdgcn_permlane32_swap(r[1][k], r[3][k], false, false);
      auto e = __builtin_amdgcn_permlane16_swap(a[0], c[0], false, false);
      auto f = __builtin_amdgcn_permlane16_swap(a[1], c[1], false, false);
      r[0][k] = e[0]; r[1][k] = e[1]; r[2][k] = f[0]; r[3][k] = f[1];
    }
  };
  auto finish = [&](int64_t t, const SegP& sp, uint32_t zpb, double s, double ab, bool take, const float (*sq)[8]) __attribute__((always_inline)) {
    const int64_t seg = t * SPW + team;
    const bool valid = (t < full || seg < p.nseg) && take;
    s = take ? s : 0.0;
    acc_sse += s;
    acc_abs += take ? ab : 0.0;
    if constexpr (EXACT) {
      if constexpr (T <= 4) {
        // np.mean: the pairwise float32 sum divided by the count, a power of two here (exact)
        const float entry = numpy_entry<(T <= 4 ? T : 1)>(sq) * (float)p.seg_mse_scale;
        if (valid && tl == T - 1) reinterpret_cast<float*>(p.seg_sse)[seg] = entry;
      }
    } else if (sampling) {
      s = team_sum<T>(s);
      if (valid && tl == 0) p.sample[seg] = (float)(s * p.seg_mse_scale);  // (seg: the dense sample index here)
      return;
    } else if (feeding) {
      s = team_sum<T>(s);
      const uint32_t bits = f_bits((float)(s * p.seg_mse_scale));  // the entry store_seg would write
      const bool mine = valid && tl == 0;
      nbelow += (mine && bits < blo) ? 1u : 0u;
      const bool inb = mine && bits >= blo && bits <= bhi;        // (a NaN entry lies above every bracket)
      const unsigned long long mask = __builtin_amdgcn_ballot_w64(inb);
      if (mask) {
        const uint32_t at = nkept + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
        if (inb && at < (uint32_t)LIST_SLOT) slot[at] = bits;
        nkept += (uint32_t)__builtin_popcountll(mask);
      }
    } else if (p.seg_sse) {
      s = team_sum<T>(s);
      if (valid && tl == 0) store_seg(p, seg, s);
    }
    if (valid && tl == 0) {
      if (p.scale) p.scale[seg] = (uint16_t)h_bits(sp.s);
      if (FMT == MQ_INT_ASYM && p.zp) p.zp[seg] = (uint16_t)zpb;
    }
  };
  // the zero-point to store (32-element segments: the lane's four vectors are the segment, in memory order)
  auto zp_of = [&](const SegP& sp, uint32_t red, const u32x4* raw) __attribute__((always_inline)) {
    uint32_t zpb = h_bits(sp.zp);
    if constexpr (FMT == MQ_INT_ASYM && T == 1 && TR) {
      if (p.zp) zpb = zp_bits_first_zero<1, NV>(sp, red, raw, 0, 1, true, p.L);
    }
    return zpb;
  };
  auto step = [&](int64_t t, const u32x4* buf) __attribute__((always_inline)) {
    u32x4 raw[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) raw[j] = buf[j];
    transpose(raw);
#pragma unroll
    for (int j = 0; j < NV; ++j) raw[j] = stats_cvt<BF16>(raw[j]);
    const uint32_t red = team_reduce_u32<T>(stats_lane_reduce<FMT, NV>(raw), stats_op<FMT>);
    const SegP sp = stats_params<FMT>(red, p);
    const uint32_t zpb = zp_of(sp, red, raw);
    seen_slow = seen_slow || sp.slow;
    double s, ab;
    if constexpr (EXACT) {
      float sq[NV][8];
      stats_vectors<FMT, NV, true>(p, raw, sp, s, ab, sq);
      finish(t, sp, zpb, s, ab, !sp.slow, sq);
    } else {
      stats_vectors<FMT, NV>(p, raw, sp, s, ab);  // (a lane past the end holds zeros: no error)
      finish(t, sp, zpb, s, ab, !sp.slow, nullptr);
    }
  };
  // two register buffers take turns, so that the next step's loads are in flight while this one is computed and
  // no vector is copied between registers (past the wave's last step the prefetch loads nothing, see stats_bload)
  {
    u32x4 bufa[NV], bufb[NV];
    int64_t t = gw;
    issue(t, bufa);
    while (t < steps) {
      const int64_t t1 = t + nwaves, t2 = t1 + nwaves;
      issue(t1, bufb);
      step(t, bufa);
      if (t1 >= steps) break;
      issue(t2, bufa);
      step(t1, bufb);
      t = t2;
    }
  }
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(seen_slow) != 0, 0)) {
#pragma unroll 1
    for (int64_t t = gw; t < steps; t += nwaves) {
      u32x4 raw[NV];
      issue(t, raw);
      transpose(raw);
#pragma unroll
      for (int j = 0; j < NV; ++j) raw[j] = stats_cvt<BF16>(raw[j]);
      const uint32_t red = team_reduce_u32<T>(stats_lane_reduce<FMT, NV>(raw), stats_op<FMT>);
      const SegP sp = stats_params<FMT>(red, p);
      if (__builtin_amdgcn_ballot_w64(sp.slow) == 0) continue;
      const uint32_t zpb = zp_of(sp, red, raw);
      double s = 0.0, ab = 0.0;
      float sq[NV][8];
      if constexpr (EXACT) {
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
          for (int i = 0; i < 8; ++i) sq[j][i] = 0.f;
      }
      if (sp.slow)
        stats_slow_segment<FMT>(p, sp, wb + step_of(t) * 4096 + (TR ? lane * 64 : (int)loff),
                                (t < full || t * SPW + team < p.nseg) ? NV : 0, TR ? 16 : 16 * T, s, ab,
                                EXACT ? sq : nullptr);
      finish(t, sp, zpb, s, ab, sp.slow, sq);
    }
  }
  if constexpr (!EXACT) {
    if (feeding) {  // the wave's counts (plain stores: k_list_cand adds them up)
      for (int off = 32; off > 0; off >>= 1) nbelow += __shfl_xor(nbelow, off);
      if (lane == 0) {
        p.wg_cnt[2 * ((size_t)p.bx * 4 + wv)] = nbelow;
        p.wg_cnt[2 * ((size_t)p.bx * 4 + wv) + 1] = nkept;
        if (nkept > (uint32_t)LIST_SLOT) p.feed->ovf = 1u;
      }
    }
    // (the sampling and repair launches run beside another pass: its partial sums are not theirs to write)
    if (list_mode == LIST_SAMPLE || list_mode == LIST_REPAIR) return;
  }
  block_partials(p, acc_sse, acc_abs, lds);
}

// Rows.  WPR waves share a row (1: a wave per row, 4: the workgroup per row); lane l of wave w holds vectors
// (j * WPR + w) * 64 + l, j < NV, so every load instruction of a wave reads 1 KiB of consecutive bytes.  Vectors
// past the end of a row (rows that are not a multiple of 512 * NV * WPR elements) are loaded as zeros: they
// change neither the absolute maximum nor the error sums (min / max skip them explicitly).
// (The second launch bound -- waves per SIMD -- and the scheduling barriers of stats_vectors keep the compiler from
//  interleaving all NV vectors' arithmetic, which took 160 .. 220 registers.)
template <int FMT, bool BF16, int NV, int WPR>
__global__ __launch_bounds__(256, (MQ_IS_LUT(FMT) ? 3 : NV <= 4 ? 6 : 4)) void k_stats_row(const QuantBatch b) {
  const QuantParams p = batch_params(b);
  __shared__ double lds[32];
  __shared__ uint32_t xr[2][4];
  __shared__ double xs[2][4];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t first = WPR == 1 ? (int64_t)p.bx * 4 + wv : (int64_t)p.bx;
  const int64_t stride = WPR == 1 ? (int64_t)p.gx * 4 : (int64_t)p.gx;
  const int sub = WPR == 1 ? 0 : wv;  // which part of the row this wave holds
  const char* wb = static_cast<const char*>(p.w);
  const int64_t row_bytes = p.L * 2;
  const uint32_t loff = (uint32_t)(sub * 1024 + lane * 16);
  const bool whole = p.vecs_per_seg == 64 * NV * WPR;  // no vector of the grid lies past the row's end
  double acc_sse = 0.0, acc_abs = 0.0;
  int parity = 0;
  bool seen_slow = false;
  const __amdgpu_buffer_rsrc_t rs = stats_rsrc(p);
  // this lane's NV vectors of a row; zeros for vectors past the row's end and for a row past the last one
  auto issue = [&](int64_t row, u32x4* dst) __attribute__((always_inline)) {
    const uint32_t base = row < p.nseg ? (uint32_t)(row * row_bytes) + loff : MQ_STATS_OOB;
    if (whole) {
#pragma unroll
      for (int j = 0; j < NV; ++j) dst[j] = stats_bload<true>(rs, base + j * (1024 * WPR));
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j)
        dst[j] = stats_bload<true>(rs, (j * WPR + sub) * 64 + lane < p.vecs_per_seg ? base + j * (1024 * WPR) : MQ_STATS_OOB);
    }
  };
  // this row's parameters from the lane's vectors (every wave of the row gets the same)
  auto row_params = [&](const u32x4* raw) __attribute__((always_inline)) {
    uint32_t red = stats_lane_reduce<FMT, NV>(raw);
    if constexpr (FMT == MQ_INT_ASYM) {
      if (!whole) {  // the zeros past the row's end must not enter min / max
        red = 0xfc007c00u;  // (+inf, -inf): neutral
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          u32x4 one[1] = {raw[j]};
          if ((j * WPR + sub) * 64 + lane < p.vecs_per_seg) red = op_minmax(red, stats_lane_reduce<FMT, 1>(one));
        }
      }
    }
    red = team_reduce_u32<64>(red, stats_op<FMT>);
    if constexpr (WPR > 1) {
      if (lane == 0) xr[parity][wv] = red;
      __syncthreads();
#pragma unroll
      for (int i = 0; i < WPR; ++i) red = stats_op<FMT>(red, xr[parity][i]);
    }
    return stats_params<FMT>(red, p);
  };
  auto finish = [&](int64_t row, const SegP& sp, double s, double ab) __attribute__((always_inline)) {
    acc_sse += s;
    acc_abs += ab;
    if (p.seg_sse) {
      s = team_sum<64>(s);
      if constexpr (WPR > 1) {
        if (lane == 0) xs[parity][wv] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
          double t = 0.0;
#pragma unroll
          for (int i = 0; i < WPR; ++i) t += xs[parity][i];
          store_seg(p, row, t);
        }
      } else if (lane == 0) {
        store_seg(p, row, s);
      }
    }
    if (lane == 0 && sub == 0) {
      if (p.scale) p.scale[row] = (uint16_t)h_bits(sp.s);
      if (FMT == MQ_INT_ASYM && p.zp) p.zp[row] = (uint16_t)h_bits(sp.zp);
    }
  };
  auto step = [&](int64_t row, const u32x4* buf) __attribute__((always_inline)) {
    u32x4 raw[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) raw[j] = stats_cvt<BF16>(buf[j]);
    const SegP sp = row_params(raw);
    if (__builtin_expect(sp.slow, 0)) {  // (uniform over the row's waves: no barrier is skipped by some of them)
      seen_slow = true;
    } else {
      double s, ab;
      stats_vectors<FMT, NV>(p, raw, sp, s, ab);
      finish(row, sp, s, ab);
    }
    parity ^= 1;
  };
  if constexpr (NV <= 8) {
    // two register buffers take turns: the next row's loads are in flight while this one is computed (past the
    // last row the prefetch loads nothing, see stats_bload)
    u32x4 bufa[NV], bufb[NV];
    int64_t row = first;
    issue(row, bufa);
    while (row < p.nseg) {
      const int64_t r1 = row + stride, r2 = r1 + stride;
      issue(r1, bufb);
      step(row, bufa);
      if (r1 >= p.nseg) break;
      issue(r2, bufa);
      step(r1, bufb);
      row = r2;
    }
  } else {
    for (int64_t row = first; row < p.nseg; row += stride) {
      u32x4 buf[NV];
      issue(row, buf);
      step(row, buf);
    }
  }
  if (__builtin_expect(seen_slow, 0)) {
#pragma unroll 1
    for (int64_t row = first; row < p.nseg; row += stride) {
      SegP sp;
      {
        u32x4 raw[NV];
        issue(row, raw);
#pragma unroll
        for (int j = 0; j < NV; ++j) raw[j] = stats_cvt<BF16>(raw[j]);
        sp = row_params(raw);
      }
      if (sp.slow) {
        int count = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j) count += (j * WPR + sub) * 64 + lane < p.vecs_per_seg;
        double s, ab;
        stats_slow_segment<FMT>(p, sp, wb + row * row_bytes + loff, count, 1024 * WPR, s, ab);
        finish(row, sp, s, ab);
      }
      parity ^= 1;
    }
  }
  block_partials(p, acc_sse, acc_abs, lds);
}

#include "mq_np.h"
#include "mq_multi.h"

// Runs a post-order program of numpy's split tree (see NumpyProg) over the leaves' sums: bit 0 takes the next leaf's
// sum, bit 1 adds the two newest partial sums (the older one on the left).  One lane; the stack lives in registers
// (selected by compare chains, not indexed: an indexed register array would go to scratch memory).
template <typename T, typename LeafPtr>
__device__ __forceinline__ T numpy_run_prog(const uint32_t* ops, int nops, LeafPtr leaves) {
  T st[12];  // (at most ten levels for 512 leaves)
#pragma unroll
  for (int j = 0; j < 12; ++j) st[j] = (T)0;
  int top = 0, next = 0;
#pragma unroll 1
  for (int i = 0; i < nops; ++i) {
    if ((ops[i >> 5] >> (i & 31)) & 1u) {
      T x = (T)0, y = (T)0;
#pragma unroll
      for (int j = 0; j < 12; ++j) { x = j == top - 2 ? st[j] : x; y = j == top - 1 ? st[j] : y; }
      const T z = x + y;
#pragma unroll
      for (int j = 0; j < 12; ++j) st[j] = j == top - 2 ? z : st[j];
      --top;
    } else {
      const T z = leaves[next++];
#pragma unroll
      for (int j = 0; j < 12; ++j) st[j] = j == top ? z : st[j];
      ++top;
    }
  }
  return st[0];
}

// Rows (segments of more than 128 elements) with the per-segment output summed in numpy's order: the float32 mean
// squared error of a row has the BITS of np.mean(diff * diff) over that row (compute_mse on a row slice,
// quant_error_single_weight.py:46-48, :56-60).  A wave per row; the row is read twice (reduction, then the errors:
// the second read comes from L2).  Lane l adds up whole leaves of numpy's split tree (b.np) -- eight accumulators
// r[i] += a[8 k + i] over the leaf's vectors in memory order, then ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7))
// -- and lane 0 combines the leaf sums in the tree's order.  np.mean divides the float32 sum by the count in
// float64 and rounds to float32: one correctly rounded float32 division.  A correctness mode, not a bandwidth
// path: the leaves' strided reads and the one-lane combine leave most of the machine idle.
template <int FMT, bool BF16>
__global__ __launch_bounds__(256) void k_rows_numpy(const QuantBatch b) {
  const QuantParams p = batch_params(b);
  __shared__ double lds[32];
  __shared__ float leaf_sum[4][MQ_NP_MAX_LEAVES];
  __shared__ uint16_t leaf_off[MQ_NP_MAX_LEAVES];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) {
    uint32_t o = 0;
    for (int i = 0; i < b.np.nleaves; ++i) { leaf_off[i] = (uint16_t)o; o += b.np.len[i]; }
  }
  __syncthreads();
  const char* wb = static_cast<const char*>(p.w);
  const int64_t row_bytes = p.L * 2;
  double acc_sse = 0.0, acc_abs = 0.0;
  for (int64_t row = (int64_t)p.bx * 4 + wv; row < p.nseg; row += (int64_t)p.gx * 4) {
    const char* rp = wb + row * row_bytes;
    uint32_t red = FMT == MQ_INT_ASYM ? 0xfc007c00u : 0u;  // (+inf, -inf) resp. 0: neutral
    for (int v = lane; v < p.vecs_per_seg; v += 64) {
      u32x4 one[1] = {stats_cvt<BF16>(*reinterpret_cast<const u32x4*>(rp + (int64_t)v * 16))};
      red = stats_op<FMT>(red, stats_lane_reduce<FMT, 1>(one));
    }
    const SegP sp = stats_params<FMT>(team_reduce_u32<64>(red, stats_op<FMT>), p);
    for (int leaf = lane; leaf < b.np.nleaves; leaf += 64) {
      float r[8];
      const int nv = b.np.len[leaf];
      const char* lp = rp + (int64_t)leaf_off[leaf] * 16;
#pragma unroll 1
      for (int k = 0; k < nv; ++k) {
        const u32x4 raw = stats_cvt<BF16>(*reinterpret_cast<const u32x4*>(lp + k * 16));
        float sq[8];
        if (!sp.slow) {
          float q = 0.f, ab = 0.f;
          quant8_stats<FMT, true, true>(p, raw, sp, q, ab, sq);
          acc_sse += (double)q;
          acc_abs += (double)ab;
        } else {
          Acc acc;
          acc.abs = 0.0;
          acc.xp = nullptr;
          acc.sq = sq;
          acc_sse += quant8_slow<FMT, SEL_STATS_ONLY>(p, raw, sp, 0, acc);
          acc_abs += acc.abs;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = k == 0 ? sq[i] : r[i] + sq[i];
      }
      leaf_sum[wv][leaf] = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane == 0) {
      const float total = numpy_run_prog<float>(b.np.ops, b.np.nops, &leaf_sum[wv][0]);
      if (p.seg_sse) reinterpret_cast<float*>(p.seg_sse)[row] = total / (float)p.L;
      if (p.scale) p.scale[row] = (uint16_t)h_bits(sp.s);
      if (FMT == MQ_INT_ASYM && p.zp) p.zp[row] = (uint16_t)h_bits(sp.zp);
    }
    __builtin_amdgcn_wave_barrier();  // (lane 0 has read the leaf sums before the next row's are written)
  }
  block_partials(p, acc_sse, acc_abs, lds);
}

// The float32 sum of squared errors over the whole flattened tensor in numpy's order -- what np.mean((a32 - b32) ** 2)
// of compute_mse adds up for a 2-D array (quant_error_multi_weight.py:55-58): chunks of 8192 elements, each summed
// pairwise (64 leaves of 128 elements in a whole chunk; b.np: the split of the last, shorter chunk).  The chunks
// cross rows and groups, so the segments' parameters come from a pass that ran before: p.scale / p.zp are INPUTS
// here, and every element goes through the per-element routine (quant8_slow: the contract literally, equal to the
// fast paths wherever those apply).  A wave per chunk, a lane per leaf; chunk sums go to the workspace as float32
// [chunks], k_tensor_mse_finish adds them in order.  A correctness mode, not a bandwidth path.
template <int FMT, bool BF16>
__global__ __launch_bounds__(256) void k_tensor_mse_numpy(const QuantBatch b) {
  const QuantParams p = batch_params(b);
  __shared__ float leaf_sum[4][128];
  __shared__ uint16_t leaf_off[128];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) {
    uint32_t o = 0;
    for (int i = 0; i < b.np.nleaves; ++i) { leaf_off[i] = (uint16_t)o; o += b.np.len[i]; }
  }
  __syncthreads();
  const int64_t nfull = p.n / 8192, nchunks = nfull + (p.n % 8192 ? 1 : 0);
  float* cs = reinterpret_cast<float*>(p.partials);
  const char* wb = static_cast<const char*>(p.w);
  auto leaf = [&](int64_t e0, int nv) __attribute__((always_inline)) {
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = 0.f;
#pragma unroll 1
    for (int k = 0; k < nv; ++k) {
      const int64_t e = e0 + (int64_t)k * 8, seg = e / p.L;
      SegP sp;
      sp.s = h_from_bits(p.scale[seg]);
      sp.sf = (float)sp.s;
      sp.r = 1.0f / sp.sf;
      sp.zp = FMT == MQ_INT_ASYM ? h_from_bits(p.zp[seg]) : (half_t)0;
      sp.slow = true;
      const u32x4 raw = stats_cvt<BF16>(*reinterpret_cast<const u32x4*>(wb + e * 2));
      float sq[8];
      Acc acc;
      acc.abs = 0.0;
      acc.xp = nullptr;
      acc.sq = sq;
      quant8_slow<FMT, SEL_STATS_ONLY>(p, raw, sp, 0, acc);
#pragma unroll
      for (int i = 0; i < 8; ++i) r[i] = k == 0 ? sq[i] : r[i] + sq[i];
    }
    return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  };
  auto wave_sync = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  for (int64_t c = (int64_t)p.bx * 4 + wv; c < nchunks; c += (int64_t)p.gx * 4) {
    float total = 0.f;
    if (c < nfull) {
      leaf_sum[wv][lane] = leaf(c * 8192 + (int64_t)lane * 128, 16);
      wave_sync();
      // the perfect tree over the 64 leaves: neighbours first, the left operand on the left
      for (int w2 = 1; w2 < 64; w2 <<= 1) {
        if ((lane & (2 * w2 - 1)) == 0) leaf_sum[wv][lane] = leaf_sum[wv][lane] + leaf_sum[wv][lane + w2];
        wave_sync();
      }
      total = leaf_sum[wv][0];
    } else {
      for (int l = lane; l < b.np.nleaves; l += 64)
        leaf_sum[wv][l] = leaf(c * 8192 + (int64_t)leaf_off[l] * 8, b.np.len[l]);
      wave_sync();
      if (lane == 0) total = numpy_run_prog<float>(b.np.ops, b.np.nops, &leaf_sum[wv][0]);
    }
    if (lane == 0) cs[c] = total;
    wave_sync();  // (the leaf sums are consumed before the next chunk's are written)
  }
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

#if MQ_PART == 0
__global__ __launch_bounds__(256) void k_tensor_reduce(const QuantBatch b) {
  const QuantParams p = batch_params(b);
  uint32_t* tred = const_cast<uint32_t*>(p.tred);
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
      uint32_t a = fold_max_u16(absmax8_pk(raw));
      amax = a > amax ? a : amax;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        uint32_t k = h_key((i & 1) ? (raw[i >> 1] >> 16) : (raw[i >> 1] & 0xffffu));
        kmax = k > kmax ? k : kmax;
        kmin = k < kmin ? k : kmin;
      }
    }
  }
  amax = team_reduce_u32<64>(amax, op_max_u32);
  kmax = team_reduce_u32<64>(kmax, op_max_u32);
  uint32_t nkmin = team_reduce_u32<64>(0xffffu - kmin, op_max_u32);
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&tred[0], amax);
    atomicMax(&tred[1], kmax);
    atomicMax(&tred[2], nkmin);
  }
}

#endif  // MQ_PART == 0

template <int FMT, int NV>
__global__ __launch_bounds__(256) void k_quant_tensor(const QuantBatch b) {
  const QuantParams p = batch_params(b);
  __shared__ double lds[32];
  double acc_sse = 0.0;
  __shared__ __attribute__((aligned(16))) float xpose[4 * 512];
  Acc acc;  // (this scope never feeds a selection)
  acc.abs = 0.0;
  acc.xp = (p.abs_err && !p.abs_err_f16) ? xpose + (threadIdx.x >> 6) * 512 : nullptr;
  const int lane = threadIdx.x & 63;
  const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int64_t nvec = p.n / 8;
  const int64_t tiles = (nvec + 64 * NV - 1) / (64 * NV);
  SegP sp;
  float s32 = 0.f;
  if (MQ_IS_LUT(FMT) && p.f32_math) {
    // scale = (rmax / qmax).clamp(1e-5, 1e4), all float32
    float s = f_from_bits(p.tred[0]) / p.qmax;
    s = s < 1e-5f ? 1e-5f : s;
    s = s > 1e4f ? 1e4f : s;
    s32 = s;
    finish_params(sp, (half_t)s);
    sp.zp = (half_t)0;
    sp.slow = false;
  } else if constexpr (FMT == MQ_INT_ASYM) {
    sp = params_minmax(h_unkey(0xffffu - p.tred[2]) | (h_unkey(p.tred[1]) << 16), p);
  } else {
    sp = params_absmax(p.tred[0], p);
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
      if (MQ_IS_LUT(FMT) && p.f32_math) {
        float w32[8];
        load8_f32(p.w, p.in_dtype, e0, w32);
        acc_sse += quant8_lut_f32(p, w32, raw[j], s32, e0, acc);
      } else {
        acc_sse += quant8<FMT, SEL_FEED_OFF>(p, raw[j], sp, e0, acc);
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (p.scale) p.scale[0] = (uint16_t)h_bits(sp.s);
    if (FMT == MQ_INT_ASYM && p.zp) p.zp[0] = (uint16_t)h_bits(sp.zp);
  }
  block_partials(p, acc_sse, acc.abs, lds);
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
__global__ __launch_bounds__(256) void k_quant_generic(const QuantBatch b) {
  const QuantParams p = batch_params(b);
  __shared__ double lds[96];
  uint32_t* lds_u = reinterpret_cast<uint32_t*>(lds + 32);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const half_t qmax_h = (half_t)p.qmax;
  double acc_sse = 0.0, acc_abs = 0.0;
  int parity = 0;
  for (int64_t seg = blockIdx.x; seg < p.nseg; seg += gridDim.x) {
    const int64_t base = seg * p.L;
    uint32_t m = 0, mm = 0xfc007c00u;
    for (int64_t i = threadIdx.x; i < p.L; i += blockDim.x) {
      uint32_t vb = load1_f16(p.w, p.in_dtype, base + i);
      m = op_max_u32(m, vb & 0x7fffu);
      mm = op_minmax(mm, dup16(vb));
    }
    m = team_reduce_u32<64>(m, op_max_u32);
    mm = team_reduce_u32<64>(mm, op_minmax);
    if (lane == 0) { lds_u[parity * 32 + wv] = m; lds_u[parity * 32 + 16 + wv] = mm; }
    __syncthreads();
    for (int i = 0; i < nw; ++i) {
      m = op_max_u32(m, lds_u[parity * 32 + i]);
      mm = op_minmax(mm, lds_u[parity * 32 + 16 + i]);
    }
    SegP sp = (FMT == MQ_INT_ASYM) ? params_minmax(mm, p) : params_absmax(m, p);
    sp.r = 1.0f / sp.sf;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < p.L; i += blockDim.x) {
      uint32_t vb = load1_f16(p.w, p.in_dtype, base + i);
      float vf = (float)h_from_bits(vb);
      half_t x = div_h_full(vf, sp), d;
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
        code = p.lut.sub ? h_bits((half_t)lut_round_sub((float)x, p.lut))
                         : lut_round_bits(h_bits(x), 0x8000u, 0x7c00u, p.lut);
        d = h_from_bits(code) * sp.s;
      }
      if (p.deq) p.deq[base + i] = (uint16_t)h_bits(d);
      if (p.codes) {
        if (MQ_IS_LUT(FMT)) static_cast<uint16_t*>(p.codes)[base + i] = (uint16_t)code;
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
        store_seg(p, seg, t);
      }
    }
    if (threadIdx.x == 0) {
      if (p.scale) p.scale[seg] = (uint16_t)h_bits(sp.s);
      if (FMT == MQ_INT_ASYM && p.zp) {
        uint32_t zpb = h_bits(sp.zp);
        if (zpb == 0u && (mm & 0xffffu) == 0x8000u && p.L < 64) {  // (see zp_bits_first_zero)
          for (int64_t i = 0; i < p.L; ++i) {
            const uint32_t vb = load1_f16(p.w, p.in_dtype, base + i);
            if ((vb & 0x7fffu) == 0u) { zpb = vb == 0u ? 0x8000u : 0u; break; }
          }
        }
        p.zp[seg] = (uint16_t)zpb;
      }
    }
    parity ^= 1;
  }
  block_partials(p, acc_sse, acc_abs, lds);
}

// Sum of a tensor's per-workgroup partials in a fixed order -> totals[0..1] (and seg_sse[0] for
// per-tensor quantization).  One workgroup per tensor of the batch.
#if MQ_PART == 0
__global__ __launch_bounds__(256) void k_finalize_totals(const QuantBatch b, int nblocks, int tensor_scope) {
  __shared__ double lds[32];
  const BatchItem& it = b.item[blockIdx.x];
  const double* partials = reinterpret_cast<const double*>(
      reinterpret_cast<const char*>(b.common.tred) + (int64_t)blockIdx.x * b.ws_stride + kWsHeader);
  double a = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x) { a += partials[2 * i]; c += partials[2 * i + 1]; }
  a = wave_sum(a);
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) { lds[threadIdx.x >> 6] = a; lds[16 + (threadIdx.x >> 6)] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double x = 0.0, y = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { x += lds[i]; y += lds[16 + i]; }
    if (it.totals) { it.totals[0] = x; it.totals[1] = y; }
    if (tensor_scope && it.seg_sse) {
      if (b.common.seg_mse_scale != 0.0) reinterpret_cast<float*>(it.seg_sse)[0] = (float)(x * b.common.seg_mse_scale);
      else it.seg_sse[0] = x;
    }
  }
}

// The same for a sweep launch: one workgroup per table entry sums that tensor's block range.
__global__ __launch_bounds__(256) void k_finalize_sweep(const QuantBatch b) {
  __shared__ double lds[32];
  const SweepItem& it = b.sweep[blockIdx.x];
  const double* partials = b.sweep_partials + 2 * (int64_t)it.blk0;
  double a = 0.0, c = 0.0;
  for (uint32_t i = threadIdx.x; i < it.nblk; i += blockDim.x) { a += partials[2 * i]; c += partials[2 * i + 1]; }
  a = wave_sum(a);
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) { lds[threadIdx.x >> 6] = a; lds[16 + (threadIdx.x >> 6)] = c; }
  __syncthreads();
  if (threadIdx.x == 0 && it.totals) {
    double x = 0.0, y = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { x += lds[i]; y += lds[16 + i]; }
    it.totals[0] = x;
    it.totals[1] = y;
  }
}

// np.mean of the float64 array that holds a list of float32 values (vec = np.array(per_vec); vec.mean():
// quant_error_single_weight.py:74-78), in numpy's summation order: chunks of 8192 elements added in order, each
// chunk pairwise with leaves of at most 128 elements (see NumpyProg), the sum divided by the count in float64.
// full / tail: the split programs of a whole chunk and of the last, shorter one (leaf lengths in ELEMENTS here).
struct ListProg {
  uint16_t nleaves, nops;
  uint8_t len[128];
  uint32_t ops[8];
};
#define MQ_LIST_BATCH 64
struct ListBatch {
  const float* x[MQ_LIST_BATCH];
  void* out[MQ_LIST_BATCH];
  int64_t n;       // every list of a launch has this length
  int32_t sum_only;  // 1: the sum itself (np.sum), no division
  int64_t run, S;  // element t of a list is x[(t / run) * S + t % run]: runs of `run` contiguous values, S elements
                   // apart (a contiguous list: run = n; a column of a row-major matrix: run = 1, S = columns; a block
                   // of gs columns in numpy's C order: run = gs, S = columns)
  ListProg full, tail;
};
// one leaf (n <= 128 values, the first at list position t0) in accumulator type Acc
template <typename Acc>
__device__ __forceinline__ Acc numpy_leaf(const ListBatch& b, const float* x, int64_t t0, int n) {
  auto at = [&](int i) __attribute__((always_inline)) {
    const int64_t t = t0 + i;
    return (Acc)(b.run == b.n ? x[t] : x[(t / b.run) * b.S + t % b.run]);
  };
  if (n < 8) {
    Acc res = (Acc)0;
    for (int i = 0; i < n; ++i) res += at(i);
    return res;
  }
  Acc r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = at(j);
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] += at(i + j);
  }
  Acc res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += at(i);
  return res;
}
// Acc = double: the float64 mean of float32 values (vec.mean() of the float64 array holding them), out: double.
// Acc = float: np.mean of a float32 array -- the float32 sum divided by the count in float64, rounded to float32
// (compute_mse on a slice) -- out: float.
template <typename Acc>
__global__ __launch_bounds__(128) void k_list_mean_numpy(const ListBatch b) {
  __shared__ Acc leaf[128];
  __shared__ uint16_t off_full[128], off_tail[128];
  const float* x = b.x[blockIdx.x];
  if (threadIdx.x == 0) {
    uint32_t o = 0;
    for (int i = 0; i < b.full.nleaves; ++i) { off_full[i] = (uint16_t)o; o += b.full.len[i]; }
    o = 0;
    for (int i = 0; i < b.tail.nleaves; ++i) { off_tail[i] = (uint16_t)o; o += b.tail.len[i]; }
  }
  __syncthreads();
  Acc total = (Acc)0;  // (thread 0's; the first chunk's sum as it is)
  for (int64_t c0 = 0; c0 < b.n; c0 += 8192) {
    const bool is_tail = b.n - c0 < 8192;
    const ListProg& pr = is_tail ? b.tail : b.full;
    const uint16_t* off = is_tail ? off_tail : off_full;
    if ((int)threadIdx.x < pr.nleaves) leaf[threadIdx.x] = numpy_leaf<Acc>(b, x, c0 + off[threadIdx.x], pr.len[threadIdx.x]);
    __syncthreads();
    if (!is_tail) {
      // a whole chunk: 64 leaves of 128 values, a perfect tree -- neighbours first, the left operand on the left
      for (int w2 = 1; w2 < 64; w2 <<= 1) {
        if (threadIdx.x < 64 && (threadIdx.x & (2 * w2 - 1)) == 0) leaf[threadIdx.x] = leaf[threadIdx.x] + leaf[threadIdx.x + w2];
        __syncthreads();
      }
      if (threadIdx.x == 0) total = c0 == 0 ? leaf[0] : total + leaf[0];
    } else if (threadIdx.x == 0) {
      const Acc c = numpy_run_prog<Acc>(pr.ops, pr.nops, &leaf[0]);
      total = c0 == 0 ? c : total + c;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if constexpr (sizeof(Acc) == 8) *static_cast<double*>(b.out[blockIdx.x]) = b.sum_only ? total : total / (double)b.n;
    else *static_cast<float*>(b.out[blockIdx.x]) = b.sum_only ? total : (float)((double)total / (double)b.n);
  }
}

// The chunk sums of k_tensor_mse_numpy added in order in float32 (numpy's reduction adds each chunk's pairwise sum to
// its running result), then np.mean's division: float32(float64(sum) / n).  One workgroup: the sums are staged
// through LDS 1024 at a time, thread 0 adds.
__global__ __launch_bounds__(256) void k_tensor_mse_finish(const float* cs, int64_t nchunks, int64_t n, float* out) {
  __shared__ float buf[1024];
  float total = 0.f;
  for (int64_t base = 0; base < nchunks; base += 1024) {
    for (int j = threadIdx.x; j < 1024; j += 256) buf[j] = base + j < nchunks ? cs[base + j] : 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int m = nchunks - base < 1024 ? (int)(nchunks - base) : 1024;
      for (int j = 0; j < m; ++j) total = (base == 0 && j == 0) ? buf[0] : total + buf[j];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = (float)((double)total / (double)n);
}

// The record rows of one pass of a sweep (numel, sse, sum |err|, mean squared error, P99 of the per-vector list, 0),
// scattered into the order of the caller's weights: row perm[i] from the i-th tensor's totals and percentile.
__global__ __launch_bounds__(256) void k_sweep_records(const double* totals, const double* p99, const double* numel,
                                                       const int64_t* perm, int n, double* records) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double sse = totals[2 * i], mse = sse / numel[i];
  double* r = records + 6 * perm[i];
  r[0] = numel[i];
  r[1] = sse;
  r[2] = totals[2 * i + 1];
  r[3] = mse;
  r[4] = mse != mse ? mse : p99[i];  // a list that holds a NaN has a NaN mean; np.percentile of it is NaN too
  r[5] = 0.0;
}

// The same rows gathered: row i of the table comes from source src[i] (none: zeros), whose percentile lies at
// p99[p99_index[source]].  Every row of the table is written.
__global__ __launch_bounds__(256) void k_sweep_table(const double* totals, const double* p99, const int64_t* p99_index,
                                                     const double* numel, const int64_t* src, int n, double* table) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t s = src ? src[i] : i;
  double* r = table + 6 * (int64_t)i;
  if (s < 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) r[k] = 0.0;
    return;
  }
  const double sse = totals[2 * s], mse = sse / numel[s];
  r[0] = numel[s];
  r[1] = sse;
  r[2] = totals[2 * s + 1];
  r[3] = mse;
  r[4] = mse != mse ? mse : p99[p99_index ? p99_index[s] : s];
  r[5] = 0.0;
}

#endif  // MQ_PART == 0

// ================================================================================================
// host side
// ================================================================================================
#if MQ_PART == 0
static int ilog2_ceil(int64_t x) {
  int l = 0;
  while (((int64_t)1 << l) < x) ++l;
  return l;
}

// float16 bits of a non-negative value that float16 represents exactly
static uint32_t host_f2h_exact(double v) {
  if (v < ldexp(1.0, -14)) return (uint32_t)ldexp(v, 24);
  int e = ilogb(v);
  return ((uint32_t)(e + 15) << 10) | (uint32_t)ldexp(v / ldexp(1.0, e) - 1.0, 10);
}

static bool lut_consts(const mq_quant_cfg* cfg, bool f32, LutConst* c, float* lutmax) {
  const int E = cfg->exp_bits, M = cfg->man_bits, B = cfg->exp_bias;
  *c = LutConst{};
  if (cfg->format == MQ_LUT_FP_SUB) {
    // E == 0: the uniform grid alone (the INTn value lists used as tables); values must be float16 numbers
    if (E < 0 || E > 5 || M < 1 || M > 9) return false;
    const double lim = ldexp(1.0, 1 - B), u = ldexp(1.0, 1 - B - M);
    const double mx = E == 0 ? lim - u : ldexp(2.0 - ldexp(1.0, -M), (1 << E) - 1 - B);
    if (!(mx <= 65504.0) || !(u >= ldexp(1.0, -14))) return false;
    c->sub = 1;
    c->man = M;
    c->sub_lim = lim;
    c->sub_max = mx;
    if (!f32) {  // constants of lut_round_pk_sub (float16 bit domain)
      if (1 - B + 15 < 1 || 1 - B + 15 > 30) return false;
      c->elim = (uint32_t)(1 - B + 15);
      c->shift = (uint32_t)(10 - M);
      c->half = 1u << (c->shift - 1);
      c->maxb = host_f2h_exact(mx);
      c->minpos = host_f2h_exact(u);
      c->mid = host_f2h_exact(u / 2);
    }
    *lutmax = (float)mx;
    return true;
  }
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

// float16 value of a bit pattern, on the host
static float host_h2f(uint32_t h) {
  uint32_t e = (h >> 10) & 31u, m = h & 1023u;
  float v = e == 0 ? ldexpf((float)m, -24) : ldexpf((float)(m | 1024u), (int)e - 25);
  return (h & 0x8000u) ? -v : v;
}
// Does a * RN(1/c), corrected once by the fma residual, equal a / c for every finite float16 a >= 0?
// (31 744 cases; the answer is cached per constant by the caller.)
static bool recip_division_exact(float c) {
  const float r = 1.0f / c;
  for (uint32_t b = 0; b < 0x7c00u; ++b) {
    float a = host_h2f(b), q0 = a * r;
    float q1 = __builtin_fmaf(__builtin_fmaf(-q0, c, a), r, q0);
    if (q1 != a / c) return false;
  }
  return true;
}
static bool recip_ok_cached(float c) {
  static thread_local float known_c[8];
  static thread_local int known_ok[8], nknown = 0;
  for (int i = 0; i < nknown; ++i)
    if (known_c[i] == c) return known_ok[i] != 0;
  bool ok = recip_division_exact(c);
  if (nknown < 8) { known_c[nknown] = c; known_ok[nknown] = ok; ++nknown; }
  return ok;
}

#endif  // MQ_PART == 0

#if MQ_PART >= 1 && !defined(MQ_NP_UNIT) && !defined(MQ_MULTI_UNIT)
template <int FMT, int NV, int SEL>
static void launch_team(int team_log2, dim3 grid, size_t lds, hipStream_t st, const QuantBatch& p) {
  static_assert(SEL_FEED_THREADS == 256, "the feeding kernels' candidate stages assume four waves per workgroup");
  switch (team_log2) {
    case 0: hipLaunchKernelGGL((k_quant_team<FMT, 1, NV, SEL>), grid, dim3(256), lds, st, p); break;
    case 1: hipLaunchKernelGGL((k_quant_team<FMT, 2, NV, SEL>), grid, dim3(256), lds, st, p); break;
    case 2: hipLaunchKernelGGL((k_quant_team<FMT, 4, NV, SEL>), grid, dim3(256), lds, st, p); break;
    case 3: hipLaunchKernelGGL((k_quant_team<FMT, 8, NV, SEL>), grid, dim3(256), lds, st, p); break;
    case 4: hipLaunchKernelGGL((k_quant_team<FMT, 16, NV, SEL>), grid, dim3(256), lds, st, p); break;
    case 5: hipLaunchKernelGGL((k_quant_team<FMT, 32, NV, SEL>), grid, dim3(256), lds, st, p); break;
    default: hipLaunchKernelGGL((k_quant_team<FMT, 64, NV, SEL>), grid, dim3(256), lds, st, p); break;
  }
}

template <int FMT, bool BF16>
static void launch_stats(int scope, int nv, int team_log2, dim3 grid, hipStream_t st, const QuantBatch& p) {
#define MQ_STATS_ROW(NV, WPR) hipLaunchKernelGGL((k_stats_row<FMT, BF16, NV, WPR>), grid, dim3(256), 0, st, p)
  if (scope == SC_TENSOR_NP) {
    hipLaunchKernelGGL((k_tensor_mse_numpy<FMT, BF16>), grid, dim3(256), 0, st, p);
  } else if (scope == SC_STATS_ROWNP) {
    hipLaunchKernelGGL((k_rows_numpy<FMT, BF16>), grid, dim3(256), 0, st, p);
  } else if (scope == SC_STATS_GROUP && p.common.seg_exact) {  // (the host admits team_log2 <= 2 only)
    switch (team_log2) {
      case 0: hipLaunchKernelGGL((k_stats_group<FMT, BF16, 1, true>), grid, dim3(256), 0, st, p); break;
      case 1: hipLaunchKernelGGL((k_stats_group<FMT, BF16, 2, true>), grid, dim3(256), 0, st, p); break;
      default: hipLaunchKernelGGL((k_stats_group<FMT, BF16, 4, true>), grid, dim3(256), 0, st, p); break;
    }
  } else if (scope == SC_STATS_GROUP) {
    switch (team_log2) {
      case 0: hipLaunchKernelGGL((k_stats_group<FMT, BF16, 1>), grid, dim3(256), 0, st, p); break;
      case 1: hipLaunchKernelGGL((k_stats_group<FMT, BF16, 2>), grid, dim3(256), 0, st, p); break;
      case 2: hipLaunchKernelGGL((k_stats_group<FMT, BF16, 4>), grid, dim3(256), 0, st, p); break;
      case 3: hipLaunchKernelGGL((k_stats_group<FMT, BF16, 8>), grid, dim3(256), 0, st, p); break;
      default: hipLaunchKernelGGL((k_stats_group<FMT, BF16, 16>), grid, dim3(256), 0, st, p); break;
    }
  } else if (scope == SC_STATS_ROW) {
    switch (nv) {
      case 2: MQ_STATS_ROW(2, 1); break;
      case 4: MQ_STATS_ROW(4, 1); break;
      case 8: MQ_STATS_ROW(8, 1); break;
      case 12: MQ_STATS_ROW(12, 1); break;
      default: MQ_STATS_ROW(16, 1); break;
    }
  } else {
    switch (nv) {
      case 6: MQ_STATS_ROW(6, 4); break;  // (rows of up to 12288 elements: 11008 fills 22 of 24 wave-vectors, 28 with NV = 7)
      case 7: MQ_STATS_ROW(7, 4); break;
      case 8: MQ_STATS_ROW(8, 4); break;
      case 14: MQ_STATS_ROW(14, 4); break;
      default: MQ_STATS_ROW(16, 4); break;
    }
  }
#undef MQ_STATS_ROW
}

// SEL: SEL_FEED_OFF / SEL_STATS_ONLY, or the two passes that feed a percentile selection (team scope only: rows
// are not fed).
template <int FMT, int SEL>
static void launch_fmt(int scope, int nv, int threads, int team_log2, dim3 grid, size_t lds, hipStream_t st,
                       const QuantBatch& p) {
  if constexpr (SEL == SEL_STATS_ONLY) {
    if (scope >= SC_STATS_GROUP) {
      if (p.common.in_dtype == MQ_BF16) launch_stats<FMT, true>(scope, nv, team_log2, grid, st, p);
      else launch_stats<FMT, false>(scope, nv, team_log2, grid, st, p);
      return;
    }
  }
  if (scope == SC_SUBWAVE) {
    if constexpr (SEL == SEL_STATS_ONLY) {
      if (nv == -4) {  // L = 32 * 2^team_log2: four vectors of one segment per lane
        switch (team_log2) {
          case 0: hipLaunchKernelGGL((k_quant_team4<FMT, 1>), grid, dim3(256), 0, st, p); break;
          case 1: hipLaunchKernelGGL((k_quant_team4<FMT, 2>), grid, dim3(256), 0, st, p); break;
          case 2: hipLaunchKernelGGL((k_quant_team4<FMT, 4>), grid, dim3(256), 0, st, p); break;
          case 3: hipLaunchKernelGGL((k_quant_team4<FMT, 8>), grid, dim3(256), 0, st, p); break;
          default: hipLaunchKernelGGL((k_quant_team4<FMT, 16>), grid, dim3(256), 0, st, p); break;
        }
        return;
      }
    }
    launch_team<FMT, 4, SEL>(team_log2, grid, lds, st, p);
    return;
  }
  if constexpr (!SEL_FEEDS(SEL)) {
    if (scope == SC_WAVE) {
      switch (nv) {
        case 2: hipLaunchKernelGGL((k_quant_wave<FMT, 2, SEL>), grid, dim3(256), lds, st, p); break;
        case 4: hipLaunchKernelGGL((k_quant_wave<FMT, 4, SEL>), grid, dim3(256), lds, st, p); break;
        case 8: hipLaunchKernelGGL((k_quant_wave<FMT, 8, SEL>), grid, dim3(256), lds, st, p); break;
        default: hipLaunchKernelGGL((k_quant_wave<FMT, 16, SEL>), grid, dim3(256), lds, st, p); break;
      }
    } else if (scope == SC_BLOCK) {
      if (threads == 256) {
        if (nv == 8) hipLaunchKernelGGL((k_quant_block<FMT, 8, 256, SEL>), grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL((k_quant_block<FMT, 16, 256, SEL>), grid, dim3(256), lds, st, p);
      } else {
        if (nv == 8) hipLaunchKernelGGL((k_quant_block<FMT, 8, 1024, SEL>), grid, dim3(1024), lds, st, p);
        else hipLaunchKernelGGL((k_quant_block<FMT, 16, 1024, SEL>), grid, dim3(1024), lds, st, p);
      }
    } else if (scope == SC_TENSOR) {
      hipLaunchKernelGGL((k_quant_tensor<FMT, 4>), grid, dim3(256), 0, st, p);
    } else {
      hipLaunchKernelGGL((k_quant_generic<FMT>), grid, dim3(256), 0, st, p);
    }
  }
}

#endif  // MQ_PART >= 1

// The numpy-order kernels of one format (their own translation units: MQ_NP_UNIT).
#if MQ_PART >= 1 && defined(MQ_NP_UNIT)
#define MQ_CAT2(a, b) a##b
#define MQ_CAT(a, b) MQ_CAT2(a, b)
int MQ_CAT(mq_launch_np, MQ_PART)(int bf16, int scope, int nv, int team_log2, dim3 grid, size_t lds_bytes, hipStream_t st,
                                  const QuantBatch& b) {
  if (bf16) return launch_np<MQ_PART - 1, true>(scope, nv, team_log2, grid, lds_bytes, st, b);
  return launch_np<MQ_PART - 1, false>(scope, nv, team_log2, grid, lds_bytes, st, b);
}
#endif

// The several-groupings kernels of one format (their own translation units: MQ_MULTI_UNIT).
#if MQ_PART >= 1 && defined(MQ_MULTI_UNIT)
#define MQ_CAT2(a, b) a##b
#define MQ_CAT(a, b) MQ_CAT2(a, b)
template <int FMT, bool BF16>
static int launch_multi(int nr, int wpr, dim3 grid, hipStream_t st, const QuantBatch& b) {
#define MQ_MULTI(NR, WPR) hipLaunchKernelGGL((k_stats_multi<FMT, BF16, NR, WPR>), grid, dim3(256), 0, st, b)
  if (wpr == 1) {
    switch (nr) {
      case 1: MQ_MULTI(1, 1); break;
      case 2: MQ_MULTI(2, 1); break;
      case 3: MQ_MULTI(3, 1); break;
      case 4: MQ_MULTI(4, 1); break;
      default: return MQ_ERR_UNSUPPORTED;
    }
  } else {
    switch (nr) {
      case 1: MQ_MULTI(1, 4); break;
      case 2: MQ_MULTI(2, 4); break;
      case 3: MQ_MULTI(3, 4); break;
      case 4: MQ_MULTI(4, 4); break;
      default: return MQ_ERR_UNSUPPORTED;
    }
  }
#undef MQ_MULTI
  return MQ_OK;
}
int MQ_CAT(mq_launch_multi, MQ_PART)(int bf16, int nv, int wpr, dim3 grid, hipStream_t st, const QuantBatch& b) {
  if (bf16) return launch_multi<MQ_PART - 1, true>(nv, wpr, grid, st, b);
  return launch_multi<MQ_PART - 1, false>(nv, wpr, grid, st, b);
}
#endif

// One format's kernels per translation unit.
#if MQ_PART >= 1 && !defined(MQ_NP_UNIT) && !defined(MQ_MULTI_UNIT)
#define MQ_PART_FMT (MQ_PART - 1)  // MQ_INT_SYM, MQ_INT_ASYM, MQ_LUT_FP, MQ_LUT_FP_SUB
#define MQ_CAT2(a, b) a##b
#define MQ_CAT(a, b) MQ_CAT2(a, b)
void MQ_CAT(mq_launch_fmt, MQ_PART)(int sel_mode, int scope, int nv, int threads, int team_log2, dim3 grid, size_t lds,
                                    hipStream_t st, const QuantBatch& b) {
  if (sel_mode == SEL_FEED_SAMPLE) launch_fmt<MQ_PART_FMT, SEL_FEED_SAMPLE>(scope, nv, threads, team_log2, grid, lds, st, b);
  else if (sel_mode == SEL_FEED_COUNT) launch_fmt<MQ_PART_FMT, SEL_FEED_COUNT>(scope, nv, threads, team_log2, grid, lds, st, b);
  else if (sel_mode == SEL_STATS_ONLY) launch_fmt<MQ_PART_FMT, SEL_STATS_ONLY>(scope, nv, threads, team_log2, grid, lds, st, b);
  else launch_fmt<MQ_PART_FMT, SEL_FEED_OFF>(scope, nv, threads, team_log2, grid, lds, st, b);
}
#endif

#if MQ_PART == 0
void mq_launch_fmt1(int, int, int, int, int, dim3, size_t, hipStream_t, const QuantBatch&);
void mq_launch_fmt2(int, int, int, int, int, dim3, size_t, hipStream_t, const QuantBatch&);
void mq_launch_fmt3(int, int, int, int, int, dim3, size_t, hipStream_t, const QuantBatch&);
void mq_launch_fmt4(int, int, int, int, int, dim3, size_t, hipStream_t, const QuantBatch&);
int mq_launch_np1(int, int, int, int, dim3, size_t, hipStream_t, const QuantBatch&);
int mq_launch_np2(int, int, int, int, dim3, size_t, hipStream_t, const QuantBatch&);
int mq_launch_np3(int, int, int, int, dim3, size_t, hipStream_t, const QuantBatch&);
int mq_launch_np4(int, int, int, int, dim3, size_t, hipStream_t, const QuantBatch&);
int mq_launch_multi1(int, int, int, dim3, hipStream_t, const QuantBatch&);
int mq_launch_multi2(int, int, int, dim3, hipStream_t, const QuantBatch&);
int mq_launch_multi3(int, int, int, dim3, hipStream_t, const QuantBatch&);
int mq_launch_multi4(int, int, int, dim3, hipStream_t, const QuantBatch&);

// ---- optional timing of the full quantize launches (HIP events on the launch stream) ----------------------
// State of the calling thread: a pool of events that is created outside the timed region (mq_profile_reserve, or
// on first use) and reused by every later mq_profile_begin; prof_used counts the events of the current run.
#include <mutex>
#include <vector>
static thread_local bool prof_on = false;
static thread_local std::vector<hipEvent_t> prof_pool;  // start, stop, start, stop, ...
static thread_local size_t prof_used = 0;

extern "C" int mq_profile_reserve(int64_t launches) {
  if (launches < 0) return MQ_ERR_ARG;
  while (prof_pool.size() < (size_t)launches * 2) {
    hipEvent_t e;
    MQ_HIP(hipEventCreate(&e));
    prof_pool.push_back(e);
  }
  return MQ_OK;
}

extern "C" int mq_profile_begin(void) {
  prof_used = 0;
  prof_on = true;
  return MQ_OK;
}

extern "C" int mq_profile_end(double* total_ms, int64_t* launches) {
  prof_on = false;
  double sum = 0.0;
  int64_t cnt = 0;
  int rc = MQ_OK;
  for (size_t i = 0; i + 1 < prof_used && rc == MQ_OK; i += 2) {
    float ms = 0.f;
    if (hipEventSynchronize(prof_pool[i + 1]) != hipSuccess || hipEventElapsedTime(&ms, prof_pool[i], prof_pool[i + 1]) != hipSuccess) {
      rc = MQ_ERR_HIP;
      break;
    }
    sum += ms;
    ++cnt;
  }
  prof_used = 0;  // (the events stay in the pool for the next run)
  if (total_ms) *total_ms = sum;
  if (launches) *launches = cnt;
  return rc;
}
// the next pair of pooled events, created here only if the pool was not reserved far enough
static int prof_next_pair(hipEvent_t* e0, hipEvent_t* e1) {
  if (int rc = mq_profile_reserve((int64_t)(prof_used / 2 + 1))) return rc;
  *e0 = prof_pool[prof_used];
  *e1 = prof_pool[prof_used + 1];
  prof_used += 2;
  return MQ_OK;
}

// Upper bound on the grid of the streaming kernels, in 256-thread workgroups.  Measured on the 16 x 4096^2
// batch: 1024 .. 3072 workgroups (every wave resident for the whole launch, looping over many tiles) run the
// full pass in ~518 us, 8192 in ~490 us, 32768 and more lose again (per-workgroup epilogues).  MQ_MAX_BLOCKS
// overrides it for tuning runs.
// numpy's split of a run of n elements (see NumpyProg); false if it does not fit the program.
static bool numpy_prog(int64_t n, NumpyProg* pr) {
  *pr = NumpyProg{};
  if (n < 8 || n % 8 || n > 32768) return false;
  bool ok = true;
  auto op = [&](uint32_t bit) {
    if (pr->nops >= 2 * MQ_NP_MAX_LEAVES) { ok = false; return; }
    pr->ops[pr->nops >> 5] |= bit << (pr->nops & 31);
    ++pr->nops;
  };
  auto rec = [&](auto&& self, int64_t m) -> void {
    if (!ok) return;
    if (m <= 128) {
      if (pr->nleaves >= MQ_NP_MAX_LEAVES) { ok = false; return; }
      pr->len[pr->nleaves++] = (uint8_t)(m / 8);
      op(0u);
    } else {
      int64_t n2 = m / 2;
      n2 -= n2 % 8;
      self(self, n2);
      self(self, m - n2);
      op(1u);
    }
  };
  // np.add.reduce hands its inner loop at most 8192 elements at a time (numpy's default buffer size,
  // np.getbufsize()) and adds the chunks' pairwise sums in order
  for (int64_t off = 0; off < n; off += 8192) {
    rec(rec, n - off < 8192 ? n - off : 8192);
    if (off) op(1u);
  }
  return ok;
}

// the split program of one chunk of m <= 8192 elements, leaf lengths in elements (any m >= 1)
static bool list_prog(int64_t m, ListProg* pr) {
  *pr = ListProg{};
  if (m < 1 || m > 8192) return m == 0;
  bool ok = true;
  auto op = [&](uint32_t bit) {
    if (pr->nops >= 256) { ok = false; return; }
    pr->ops[pr->nops >> 5] |= bit << (pr->nops & 31);
    ++pr->nops;
  };
  auto rec = [&](auto&& self, int64_t k) -> void {
    if (!ok) return;
    if (k <= 128) {
      if (pr->nleaves >= 128) { ok = false; return; }
      pr->len[pr->nleaves++] = (uint8_t)k;
      op(0u);
    } else {
      int64_t n2 = k / 2;
      n2 -= n2 % 8;
      self(self, n2);
      self(self, k - n2);
      op(1u);
    }
  };
  rec(rec, m);
  return ok;
}

static int64_t grid_cap() {
  static const int64_t cap = [] {
    const char* e = getenv("MQ_MAX_BLOCKS");
    int64_t v = e ? atoll(e) : 8192;
    return v < 1 ? 1 : (v > MQ_MAX_GRID ? MQ_MAX_GRID : v);
  }();
  return cap;
}

static size_t ws_slice_bytes() { return kWsHeader + (size_t)MQ_MAX_GRID * 2 * sizeof(double); }

extern "C" size_t mq_quantize_workspace_bytes(int64_t, int64_t, const mq_quant_cfg*) { return ws_slice_bytes(); }
extern "C" size_t mq_quantize_batch_workspace_bytes(int32_t count, int64_t, int64_t, const mq_quant_cfg*) {
  return (size_t)(count < 1 ? 1 : (count > MQ_MAX_BATCH ? MQ_MAX_BATCH : count)) * ws_slice_bytes();
}

// ---- device self-check of the packed float16 division -------------------------------------------------------
// One workgroup per float16 scale in [2^-15, 0x70E2]; every finite float16 v of either sign with |v / s| <= 600:
// div_pk16 (with this device's v_rcp_f32 in pk_div_params) against the float32 division sequence of the general
// kernels (div_f: equal to the IEEE quotient on this domain, tests/tools/div_exhaustive.c).
// counts[0] pairs checked, [1] pairs whose rint differs, [2] pairs with |v / s| >= 2^-4 whose bits differ.
// j > 0: with the shared power of two 2^(k - j) in place of the scale's own 2^k (pk_div_params); the bits are then
// compared for |v / s| >= 2^(j - 4).
__global__ __launch_bounds__(256) void k_selfcheck_div16(unsigned long long* counts, int j) {
  const uint32_t sb = 0x0200u + blockIdx.x;
  SegP sp;
  finish_params(sp, h_from_bits(sb));
  sp.zp = (half_t)0;
  sp.slow = false;
  pk_div_params(sp);
  if (j > 0) pk_div_params(sp, sp.pk_k - j);
  const float exact_from = __builtin_ldexpf(0.0625f, j);
  unsigned long long n = 0, bad_rint = 0, bad_bits = 0;
  for (uint32_t vb = threadIdx.x; vb < 0x7c00u; vb += blockDim.x) {
    const half_t v = h_from_bits(vb);
    const half_t want = (half_t)div_f((float)v, sp);
    if (!((float)want <= 600.0f)) continue;
    const h16x2 got = div_pk16(vb | ((vb | 0x8000u) << 16), sp);  // (+v, -v)
    n += 2;
    const float w = (float)want;
    bad_rint += (__builtin_rintf((float)got[0]) != __builtin_rintf(w)) + (__builtin_rintf((float)got[1]) != __builtin_rintf(-w));
    if (w >= exact_from) bad_bits += (h_bits(got[0]) != h_bits(want)) + (h_bits(got[1]) != (h_bits(want) | 0x8000u));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    n += __shfl_xor(n, off); bad_rint += __shfl_xor(bad_rint, off); bad_bits += __shfl_xor(bad_bits, off);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&counts[0], n); atomicAdd(&counts[1], bad_rint); atomicAdd(&counts[2], bad_bits);
  }
}
extern "C" int mq_selfcheck_div16(uint64_t* counts_device, void* hip_stream) {
  if (!counts_device) return MQ_ERR_ARG;
  if (reinterpret_cast<uintptr_t>(counts_device) % 8) return MQ_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  MQ_HIP(hipMemsetAsync(counts_device, 0, 3 * sizeof(uint64_t), st));
  hipLaunchKernelGGL(k_selfcheck_div16, dim3(0x70E2u - 0x0200u + 1u), dim3(256), 0, st,
                     reinterpret_cast<unsigned long long*>(counts_device), 0);
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}
extern "C" int mq_selfcheck_div16_shared(uint64_t* counts_device, int32_t j, void* hip_stream) {
  if (!counts_device || j < 0 || j > 2) return MQ_ERR_ARG;
  if (reinterpret_cast<uintptr_t>(counts_device) % 8) return MQ_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  MQ_HIP(hipMemsetAsync(counts_device, 0, 3 * sizeof(uint64_t), st));
  hipLaunchKernelGGL(k_selfcheck_div16, dim3(0x70E2u - 0x0200u + 1u), dim3(256), 0, st,
                     reinterpret_cast<unsigned long long*>(counts_device), (int)j);
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}

// Does the hardware's packed float32 -> fp8 (e4m3) conversion round like the reference's FP8_E4M3 table (nearest
// table value, ties toward -inf, largest value 480, no inf / NaN: quant_weight.py:114-138, :577-601)?  Every finite
// float16 x with |x| <= 480 (the scaled weights' range) goes through v_cvt_pk_fp8_f32 and back (v_cvt_f32_fp8) and
// through lut_round_bits.  counts[0]: values checked; [1]: values where the two differ; [2]: of those, exact ties
// between two table values; [3]: of those, |x| above 448 (the hardware format's largest finite value); [4]: of
// those, |x| below 2^-6, the hardware format's smallest normal value (it continues with subnormals of step 2^-9;
// the table keeps three mantissa bits down to 1.125 * 2^-7 and has nothing between that and 0).
__global__ __launch_bounds__(256) void k_selfcheck_fp8_cvt(LutConst lut, unsigned long long* counts) {
  const uint32_t xb = blockIdx.x * 256u + threadIdx.x;  // all 65536 float16 bit patterns
  const uint32_t mag = xb & 0x7fffu;
  unsigned long long n = 0, bad = 0, tie = 0, top = 0, low = 0;
  if (mag <= 0x5f80u) {  // |x| <= 480
    const float xf = (float)h_from_bits(xb);
    const int packed = __builtin_amdgcn_cvt_pk_fp8_f32(xf, xf, 0, false);
    const float hw = __builtin_amdgcn_cvt_f32_fp8(packed, 0);
    const float ref = (float)h_from_bits(lut_round_bits(xb, 0x8000u, 0x7c00u, lut));
    n = 1;
    if (!(hw == ref)) {  // (a NaN from the hardware counts as different; +0 == -0)
      bad = 1;
      // a tie: |x| has exactly the bit below the kept mantissa bits set
      const uint32_t below = mag & ((1u << lut.shift) - 1u);
      tie = below == (1u << (lut.shift - 1)) ? 1 : 0;
      top = mag > 0x5f00u ? 1 : 0;          // > 448
      low = mag < 0x2400u ? 1 : 0;           // < 2^-6
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    n += __shfl_xor(n, off); bad += __shfl_xor(bad, off); tie += __shfl_xor(tie, off); top += __shfl_xor(top, off);
    low += __shfl_xor(low, off);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&counts[0], n); atomicAdd(&counts[1], bad); atomicAdd(&counts[2], tie); atomicAdd(&counts[3], top);
    atomicAdd(&counts[4], low);
  }
}
extern "C" int mq_selfcheck_fp8_cvt(uint64_t* counts_device, void* hip_stream) {
  if (!counts_device) return MQ_ERR_ARG;
  if (reinterpret_cast<uintptr_t>(counts_device) % 8) return MQ_ERR_ALIGN;
  mq_quant_cfg cfg = {};
  cfg.format = MQ_LUT_FP;
  cfg.bits = 8;
  cfg.exp_bits = 4; cfg.man_bits = 3; cfg.exp_bias = 7;
  LutConst lut;
  float lutmax;
  if (!lut_consts(&cfg, false, &lut, &lutmax)) return MQ_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  MQ_HIP(hipMemsetAsync(counts_device, 0, 5 * sizeof(uint64_t), st));
  hipLaunchKernelGGL(k_selfcheck_fp8_cvt, dim3(256), dim3(256), 0, st, lut,
                     reinterpret_cast<unsigned long long*>(counts_device));
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}

static bool aligned(const void* q, size_t a) { return (reinterpret_cast<uintptr_t>(q) % a) == 0; }

// One launch (plus one finalize launch when totals are requested) for up to MQ_MAX_BATCH tensors.
// sel_mode: SEL_FEED_OFF, or one of the two passes of a fed selection (SEL_FEED_SAMPLE launches the kernels
// over the sampled tiles / rows only, with every output suppressed).  With `sampled` set nothing is launched:
// it receives the number of elements per tensor the sampling pass would feed (0: this shape's kernels cannot
// feed a selection).
// Set around a call: the kernel that adds up the workgroups' partial sums goes to this stream, behind this event
// recorded on the launch stream after the quantize kernel.
static thread_local hipStream_t tl_totals_stream = nullptr;
static thread_local hipEvent_t tl_totals_event = nullptr;
static int quantize_launch(const mq_quant_item* items, int32_t count, int32_t w_dtype, int64_t rows, int64_t cols,
                           const mq_quant_cfg* cfg, void* workspace, hipStream_t st, int sel_mode = SEL_FEED_OFF,
                           char* sel_ws = nullptr, int64_t* sampled = nullptr, uint16_t* samples = nullptr,
                           int64_t sample_stride = 0) {
  const int64_t n = rows * cols;
  QuantBatch b = {};
  QuantParams& p = b.common;
  p.in_dtype = w_dtype;
  p.f32_math = cfg->f32_math;
  p.n = n;
  p.L = cfg->group_size == -1 ? n : (cfg->group_size <= 0 ? cols : cfg->group_size);
  p.nseg = n / p.L;
  const int fmt = cfg->format;
  if (cfg->format == MQ_INT_SYM) p.qmax = (float)((1 << (cfg->bits - 1)) - 1);
  else if (cfg->format == MQ_INT_ASYM) p.qmax = (float)((1 << cfg->bits) - 1);
  else if (!lut_consts(cfg, cfg->f32_math != 0, &p.lut, &p.qmax)) return MQ_ERR_UNSUPPORTED;
  p.rqmax = 1.0f / p.qmax;
  p.qmax_recip_ok = recip_ok_cached(p.qmax);
  // which outputs exist is a property of the batch (taken from the first tensor, checked for the rest)
  const mq_quant_out& o0 = items[0].out;
  if (o0.abs_err && o0.abs_err_dtype != MQ_F32 && o0.abs_err_dtype != MQ_F16) return MQ_ERR_ARG;
  p.abs_err_f16 = o0.abs_err_dtype == MQ_F16;
  p.need_err = (o0.abs_err || o0.seg_sse || o0.totals) ? 1 : 0;
  if (o0.seg_mse_f32 < 0 || o0.seg_mse_f32 > 2) return MQ_ERR_ARG;
  p.seg_mse_scale = (o0.seg_sse && o0.seg_mse_f32) ? 1.0 / (double)p.L : 0.0;
  p.seg_exact = (o0.seg_sse && o0.seg_mse_f32 == 2) ? 1 : 0;
  p.tred = static_cast<const uint32_t*>(workspace);
  b.ws_stride = (int64_t)ws_slice_bytes();
  const size_t esz = w_dtype == MQ_F32 ? 4 : 2;
  bool vec_ok = p.L % 8 == 0;
  for (int i = 0; i < count; ++i) {
    const mq_quant_out& o = items[i].out;
    if (!items[i].w) return MQ_ERR_ARG;
    if (!o.deq != !o0.deq || !o.codes != !o0.codes || !o.scale != !o0.scale || !o.zero_point != !o0.zero_point ||
        !o.abs_err != !o0.abs_err || !o.seg_sse != !o0.seg_sse || !o.totals != !o0.totals ||
        (o.abs_err && o.abs_err_dtype != o0.abs_err_dtype) || o.seg_mse_f32 != o0.seg_mse_f32)
      return MQ_ERR_ARG;
    if (!aligned(items[i].w, esz) || !aligned(o.deq, 2) || !aligned(o.scale, 2) || !aligned(o.zero_point, 2) ||
        !aligned(o.seg_sse, o.seg_mse_f32 ? 4 : 8) || !aligned(o.totals, 8) ||
        !aligned(o.abs_err, p.abs_err_f16 ? 2 : 4) ||
        (MQ_IS_LUT(fmt) && !aligned(o.codes, 2)))
      return MQ_ERR_ALIGN;
    vec_ok = vec_ok && aligned(items[i].w, 16) && aligned(o.deq, 16) && aligned(o.codes, 16) && aligned(o.abs_err, 16);
    BatchItem& it = b.item[i];
    it.w = items[i].w;
    it.deq = static_cast<uint16_t*>(o.deq);
    it.codes = o.codes;
    it.scale = static_cast<uint16_t*>(o.scale);
    it.zp = static_cast<uint16_t*>(o.zero_point);
    it.abs_err = o.abs_err;
    it.seg_sse = o.seg_sse;
    it.totals = o.totals;
  }
  p.sel_mode = sel_mode;
  p.sel_samples = samples;
  b.sel_ws = sel_ws;
  b.sample_stride = sample_stride;
  if (sel_mode == SEL_FEED_SAMPLE) {  // errors only
    for (int i = 0; i < count; ++i) {
      BatchItem& it = b.item[i];
      it.deq = nullptr; it.codes = nullptr; it.scale = nullptr; it.zp = nullptr; it.abs_err = nullptr;
      it.seg_sse = nullptr; it.totals = nullptr;
    }
    p.need_err = 1;
  }
  // flags the kernels test through the (per-tensor) pointers: keep the common copy non-null where set
  p.deq = b.item[0].deq; p.codes = b.item[0].codes; p.scale = b.item[0].scale; p.zp = b.item[0].zp;
  p.abs_err = b.item[0].abs_err; p.seg_sse = b.item[0].seg_sse;

  int scope, nv = 4, threads = 256, team_log2 = 0;
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
    team_log2 = ilog2_ceil(p.L / 8);
    int64_t spw = 64 >> team_log2;
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
  int64_t max_grid = scope == SC_BLOCK || scope == -1 ? (threads == 1024 ? MQ_MAX_GRID / 4 : MQ_MAX_GRID) : grid_cap();
  // a batch shares the chip: keep the total number of workgroups near the single-tensor cap
  if (count > 1 && max_grid > 64) max_grid = max_grid / count < 64 ? 64 : max_grid / count;
  if (sel_mode == SEL_FEED_SAMPLE && scope == SC_SUBWAVE) {
    // one slot of every 8 tiles: few enough waves to be resident at once, 4 prefetched steps each
    work_blocks = (work_blocks * 4 / 8 + 15) / 16;
    if (work_blocks < 1) work_blocks = 1;
  }
  dim3 grid((unsigned)(work_blocks < max_grid ? work_blocks : max_grid), (unsigned)count);

  if (sampled) {
    *sampled = 0;
    if (scope == SC_SUBWAVE) {  // tiles of (64 / team) * 4 segments, one of every 32 tiles
      const int64_t per_tile = (int64_t)(64 >> team_log2) * 4;
      *sampled = ((p.nseg / per_tile) / 8) * (per_tile / 4) * p.L;  // one slot of every 8 full tiles
    }
    // Rows (SC_WAVE / SC_BLOCK) are not sampled: whole sampled rows are too few, too correlated clusters for
    // the bracket to hold, and a row cannot be quantized in part before its maximum is known.  Those shapes
    // select from the stored error array, whose samples are single 256-byte lines.
    return MQ_OK;
  }
  if (sel_mode != SEL_FEED_OFF && scope != SC_SUBWAVE) return MQ_ERR_UNSUPPORTED;
  const size_t lds = sel_mode == SEL_FEED_COUNT ? (size_t)2 * SEL_FEED_HASH * sizeof(uint32_t) : 0;

  if (scope == SC_TENSOR) {
    for (int i = 0; i < count; ++i)
      MQ_HIP(hipMemsetAsync(static_cast<char*>(workspace) + (size_t)i * ws_slice_bytes(), 0, 16, st));
    int64_t rb = (n / 8 + 255) / 256;
    int64_t cap = MQ_MAX_GRID / count < 64 ? 64 : MQ_MAX_GRID / count;
    hipLaunchKernelGGL(k_tensor_reduce, dim3((unsigned)(rb < cap ? rb : cap), (unsigned)count), dim3(256), 0, st, b);
  }
  const bool timed = prof_on && sel_mode != SEL_FEED_SAMPLE;
  hipEvent_t prof_e0 = nullptr, prof_e1 = nullptr;
  if (timed) {
    if (int rc = prof_next_pair(&prof_e0, &prof_e1)) return rc;
    MQ_HIP(hipEventRecord(prof_e0, st));
  }
  // statistics only: the kernels compiled without the output stores
  const int kmode = (sel_mode == SEL_FEED_OFF && p.need_err && !p.deq && !p.codes && !p.abs_err) ? SEL_STATS_ONLY : sel_mode;
  // (tables without subnormals and more than 6 mantissa bits need the upper clamp lut_round_mag_pk leaves out)
  // (... and a tensor inside the 32-bit offsets of the buffer loads)
  const bool stats_kernels = kmode == SEL_STATS_ONLY && vec_ok && (w_dtype == MQ_F16 || w_dtype == MQ_BF16) &&
                             !(fmt == MQ_LUT_FP && cfg->man_bits > 6) && n * 2 <= 0xf0000000ll;
  const bool pow2_group = scope == SC_SUBWAVE && p.L % 32 == 0 && ((p.L / 32) & (p.L / 32 - 1)) == 0;
  if (kmode == SEL_STATS_ONLY && pow2_group && !sampled && !(p.seg_exact && p.L > 128)) {
    // groups of 32 * 2^k elements: the kernels that hold four vectors of one segment per lane
    team_log2 = ilog2_ceil(p.L / 32);
    nv = -4;
    if (stats_kernels) scope = SC_STATS_GROUP;
    if (p.seg_exact && !stats_kernels) return MQ_ERR_UNSUPPORTED;
    const int64_t spw = 64 >> team_log2;
    const int64_t wb = ((p.nseg + spw - 1) / spw + 3) / 4;
    grid.x = (unsigned)(wb < max_grid ? wb : max_grid);
  } else if (p.seg_exact) {
    // numpy-order entries of every other segment length (rows, groups that are not 32 / 64 / 128 elements): the
    // wave-per-segment kernel that walks numpy's split tree
    if (!(stats_kernels && (scope == SC_SUBWAVE || scope == SC_WAVE || scope == SC_BLOCK) && p.L <= 32768 &&
          numpy_prog(p.L, &b.np)))
      return MQ_ERR_UNSUPPORTED;
    scope = SC_STATS_ROWNP;
    const int64_t wb = (p.nseg + 3) / 4;
    int64_t cap = grid_cap();
    if (count > 1) cap = cap / count < 64 ? 64 : cap / count;
    grid.x = (unsigned)(wb < cap ? wb : cap);
  } else if (stats_kernels && (scope == SC_WAVE || scope == SC_BLOCK) && p.L / 8 <= 4096) {
    const int64_t vecs = p.L / 8;
    int64_t wb;
    if (vecs <= 1024) {  // one wave per row
      scope = SC_STATS_ROW;
      nv = vecs <= 128 ? 2 : vecs <= 256 ? 4 : vecs <= 512 ? 8 : vecs <= 768 ? 12 : 16;
      wb = (p.nseg + 3) / 4;
    } else {             // one 256-thread workgroup per row
      scope = SC_STATS_ROWBLK;
      nv = vecs <= 1536 ? 6 : vecs <= 1792 ? 7 : vecs <= 2048 ? 8 : vecs <= 3584 ? 14 : 16;
      wb = p.nseg;
    }
    int64_t cap = grid_cap();
    if (count > 1) cap = cap / count < 64 ? 64 : cap / count;
    grid.x = (unsigned)(wb < cap ? wb : cap);
  }
  switch (fmt) {
    case MQ_INT_SYM: mq_launch_fmt1(kmode, scope, nv, threads, team_log2, grid, lds, st, b); break;
    case MQ_INT_ASYM: mq_launch_fmt2(kmode, scope, nv, threads, team_log2, grid, lds, st, b); break;
    case MQ_LUT_FP: mq_launch_fmt3(kmode, scope, nv, threads, team_log2, grid, lds, st, b); break;
    default: mq_launch_fmt4(kmode, scope, nv, threads, team_log2, grid, lds, st, b); break;
  }
  MQ_HIP(hipGetLastError());
  if (timed) MQ_HIP(hipEventRecord(prof_e1, st));
  hipStream_t fst = st;
  if (tl_totals_stream && sel_mode != SEL_FEED_SAMPLE) {
    // (see mq_quantize_p99_passes: whatever follows the quantize kernel goes to the second stream, behind this event)
    MQ_HIP(hipEventRecord(tl_totals_event, st));
    MQ_HIP(hipStreamWaitEvent(tl_totals_stream, tl_totals_event, 0));
    fst = tl_totals_stream;
  }
  if (sel_mode != SEL_FEED_SAMPLE && (o0.totals || (scope == SC_TENSOR && o0.seg_sse))) {
    hipLaunchKernelGGL(k_finalize_totals, dim3((unsigned)count), dim3(256), 0, fst, b, (int)grid.x,
                       scope == SC_TENSOR ? 1 : 0);
    MQ_HIP(hipGetLastError());
  }
  return MQ_OK;
}

extern "C" int mq_quantize_batch(const mq_quant_item* items, int32_t count, int32_t w_dtype, int64_t rows,
                                 int64_t cols, const mq_quant_cfg* cfg, void* workspace, size_t workspace_bytes,
                                 void* hip_stream) {
  if (!cfg || count < 0 || rows < 0 || cols < 0 || (count > 0 && !items)) return MQ_ERR_ARG;
  if (w_dtype != MQ_F16 && w_dtype != MQ_BF16 && w_dtype != MQ_F32) return MQ_ERR_ARG;
  const bool is_lut = cfg->format == MQ_LUT_FP || cfg->format == MQ_LUT_FP_SUB;
  if (cfg->format != MQ_INT_SYM && cfg->format != MQ_INT_ASYM && !is_lut) return MQ_ERR_ARG;
  if (!is_lut && (cfg->bits < 2 || cfg->bits > 8)) return MQ_ERR_UNSUPPORTED;
  if (cfg->f32_math && !(is_lut && cfg->group_size == -1)) return MQ_ERR_UNSUPPORTED;
  if (cfg->group_size > 0 && cols % cfg->group_size != 0) return MQ_ERR_GROUP;
  if (count == 0 || rows * cols == 0) return MQ_OK;
  for (int32_t i = 0; i < count; ++i)
    if (!items[i].w) return MQ_ERR_ARG;
  if (!workspace || workspace_bytes < mq_quantize_batch_workspace_bytes(count, rows, cols, cfg)) return MQ_ERR_WORKSPACE;
  if (!aligned(workspace, 16)) return MQ_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  for (int32_t done = 0; done < count; done += MQ_MAX_BATCH) {
    int32_t c = count - done < MQ_MAX_BATCH ? count - done : MQ_MAX_BATCH;
    int rc = quantize_launch(items + done, c, w_dtype, rows, cols, cfg, workspace, st);
    if (rc) return rc;
  }
  return MQ_OK;
}

// ---- one pass of a model sweep as ONE launch (tensors of different sizes) -------------------------------------
// Which statistics kernel serves segments of L elements (the choice of quantize_launch for statistics-only calls on
// float16 / bfloat16 weights), or false when none does.
static bool sweep_scope(int64_t L, int* scope, int* nv, int* team_log2) {
  if (L <= 0 || L % 8) return false;
  const int64_t vecs = L / 8;
  *team_log2 = 0;
  if (vecs <= 64) {  // groups of 32 * 2^k elements
    if (L % 32 || ((L / 32) & (L / 32 - 1))) return false;
    *scope = SC_STATS_GROUP;
    *team_log2 = ilog2_ceil(L / 32);
    *nv = -4;
  } else if (vecs <= 1024) {
    *scope = SC_STATS_ROW;
    *nv = vecs <= 128 ? 2 : vecs <= 256 ? 4 : vecs <= 512 ? 8 : vecs <= 768 ? 12 : 16;
  } else if (vecs <= 4096) {
    *scope = SC_STATS_ROWBLK;
    *nv = vecs <= 1536 ? 6 : vecs <= 1792 ? 7 : vecs <= 2048 ? 8 : vecs <= 3584 ? 14 : 16;
  } else {
    return false;
  }
  return true;
}
// Weight bytes one workgroup of a sweep launch works through.  Measured on the Llama-3-8B sweep (15 GB per pass):
// 64 KiB 11.65 ms per four passes, 128 KiB 11.22, 256 KiB 11.23, 512 KiB 12.2 (the batched launches' grid cap
// gives 235 KiB there: 11.42).  MQ_SWEEP_WG_KB overrides it for tuning runs.
static int64_t sweep_wg_bytes() {
  static const int64_t v = [] {
    const char* e = getenv("MQ_SWEEP_WG_KB");
    int64_t kb = e ? atoll(e) : 128;
    return (kb < 4 ? 4 : (kb > 65536 ? 65536 : kb)) * 1024;
  }();
  return v;
}
static int sweep_check_cfg(int32_t w_dtype, const mq_quant_cfg* cfg) {
  if (!cfg) return MQ_ERR_ARG;
  if (w_dtype != MQ_F16 && w_dtype != MQ_BF16) return w_dtype == MQ_F32 ? MQ_ERR_UNSUPPORTED : MQ_ERR_ARG;
  const bool is_lut = cfg->format == MQ_LUT_FP || cfg->format == MQ_LUT_FP_SUB;
  if (cfg->format != MQ_INT_SYM && cfg->format != MQ_INT_ASYM && !is_lut) return MQ_ERR_ARG;
  if (!is_lut && (cfg->bits < 2 || cfg->bits > 8)) return MQ_ERR_UNSUPPORTED;
  if (cfg->f32_math || cfg->group_size == -1) return MQ_ERR_UNSUPPORTED;
  if (cfg->format == MQ_LUT_FP && cfg->man_bits > 6) return MQ_ERR_UNSUPPORTED;
  return MQ_OK;
}

extern "C" size_t mq_sweep_table_bytes(int32_t count) { return (size_t)(count < 1 ? 1 : count) * sizeof(SweepItem); }
extern "C" size_t mq_sweep_workspace_bytes(int64_t total_blocks) {
  return (size_t)(total_blocks < 1 ? 1 : total_blocks) * 2 * sizeof(double);
}

extern "C" int mq_sweep_prepare(const mq_quant_item* items, const int64_t* rows, const int64_t* cols, int32_t count,
                                int32_t w_dtype, const mq_quant_cfg* cfg, void* table_host, size_t table_bytes,
                                int64_t* seg_len, int64_t* total_blocks) {
  if (!items || !rows || !cols || count < 1 || !table_host || !seg_len || !total_blocks) return MQ_ERR_ARG;
  if (int rc = sweep_check_cfg(w_dtype, cfg)) return rc;
  if (table_bytes < mq_sweep_table_bytes(count)) return MQ_ERR_WORKSPACE;
  const int64_t L = cfg->group_size > 0 ? cfg->group_size : cols[0];
  int scope, nv, team_log2;
  if (!sweep_scope(L, &scope, &nv, &team_log2)) return MQ_ERR_UNSUPPORTED;
  const mq_quant_out& o0 = items[0].out;
  if (o0.seg_mse_f32 < 0 || o0.seg_mse_f32 > 2) return MQ_ERR_ARG;
  if (o0.seg_mse_f32 == 2 && !(o0.seg_sse && scope == SC_STATS_GROUP && team_log2 <= 2)) return MQ_ERR_UNSUPPORTED;
  if (!o0.seg_sse && !o0.totals) return MQ_ERR_ARG;  // nothing to compute
  // workgroups per tensor: one per sweep_wg_bytes() of weights, in whole steps of the kernel
  int64_t unit_bytes, units_per_block;  // a unit = what one wave (one workgroup: SC_STATS_ROWBLK) takes per step
  if (scope == SC_STATS_GROUP) unit_bytes = 4096;
  else unit_bytes = L * 2;
  units_per_block = sweep_wg_bytes() / unit_bytes / (scope == SC_STATS_ROWBLK ? 1 : 4);
  if (units_per_block < 1) units_per_block = 1;
  if (scope != SC_STATS_ROWBLK) units_per_block *= 4;
  SweepItem* t = static_cast<SweepItem*>(table_host);
  int64_t blocks = 0;
  for (int32_t i = 0; i < count; ++i) {
    const mq_quant_out& o = items[i].out;
    if (rows[i] < 1 || cols[i] < 1 || !items[i].w) return MQ_ERR_ARG;
    if (cfg->group_size > 0 ? cols[i] % L != 0 : cols[i] != L) return cfg->group_size > 0 ? MQ_ERR_GROUP : MQ_ERR_ARG;
    if (o.deq || o.codes || o.abs_err) return MQ_ERR_UNSUPPORTED;  // statistics only
    if (!o.seg_sse != !o0.seg_sse || !o.totals != !o0.totals || !o.scale != !o0.scale ||
        !o.zero_point != !o0.zero_point || o.seg_mse_f32 != o0.seg_mse_f32)
      return MQ_ERR_ARG;
    if (!aligned(items[i].w, 16)) return MQ_ERR_UNSUPPORTED;  // (the batched entry points take unaligned weights)
    if (!aligned(o.scale, 2) || !aligned(o.zero_point, 2) || !aligned(o.seg_sse, o.seg_mse_f32 ? 4 : 8) ||
        !aligned(o.totals, 8))
      return MQ_ERR_ALIGN;
    const int64_t n = rows[i] * cols[i];
    if (n * 2 > 0xf0000000ll) return MQ_ERR_UNSUPPORTED;  // the 32-bit offsets of the buffer loads
    const int64_t nseg = n / L;
    const int64_t units = scope == SC_STATS_GROUP ? (nseg + (64 >> team_log2) - 1) / (64 >> team_log2) : nseg;
    int64_t nb = (units + units_per_block - 1) / units_per_block;
    if (nb < 1) nb = 1;
    if (blocks + nb > 0x7fffffffll) return MQ_ERR_TOO_LARGE;
    t[i] = SweepItem{items[i].w, static_cast<uint16_t*>(o.scale), static_cast<uint16_t*>(o.zero_point), o.seg_sse,
                     o.totals, nseg, (uint32_t)blocks, (uint32_t)nb, nullptr, 0u, 0u, nullptr, 0u, 0u, nullptr, nullptr, nullptr,
                     0u, 0u};
    blocks += nb;
  }
  *seg_len = L;
  *total_blocks = blocks;
  return MQ_OK;
}

extern "C" int mq_sweep_run(const void* table_dev, int32_t count, int64_t seg_len, int64_t total_blocks, int32_t w_dtype,
                            const mq_quant_cfg* cfg, int32_t seg_mse_f32, void* workspace, size_t workspace_bytes,
                            void* hip_stream) {
  if (!table_dev || count < 1 || total_blocks < 1 || total_blocks > 0x7fffffffll) return MQ_ERR_ARG;
  if (int rc = sweep_check_cfg(w_dtype, cfg)) return rc;
  if (seg_mse_f32 < 0 || seg_mse_f32 > 2) return MQ_ERR_ARG;
  int scope, nv, team_log2;
  if (!sweep_scope(seg_len, &scope, &nv, &team_log2)) return MQ_ERR_UNSUPPORTED;
  if (seg_mse_f32 == 2 && !(scope == SC_STATS_GROUP && team_log2 <= 2)) return MQ_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < mq_sweep_workspace_bytes(total_blocks)) return MQ_ERR_WORKSPACE;
  if (!aligned(workspace, 16) || !aligned(table_dev, 16)) return MQ_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  QuantBatch b = {};
  QuantParams& p = b.common;
  p.in_dtype = w_dtype;
  p.L = seg_len;
  p.vecs_per_seg = (int32_t)(seg_len / 8);
  const int fmt = cfg->format;
  if (fmt == MQ_INT_SYM) p.qmax = (float)((1 << (cfg->bits - 1)) - 1);
  else if (fmt == MQ_INT_ASYM) p.qmax = (float)((1 << cfg->bits) - 1);
  else if (!lut_consts(cfg, false, &p.lut, &p.qmax)) return MQ_ERR_UNSUPPORTED;
  p.rqmax = 1.0f / p.qmax;
  p.qmax_recip_ok = recip_ok_cached(p.qmax);
  p.need_err = 1;
  p.seg_mse_scale = seg_mse_f32 ? 1.0 / (double)seg_len : 0.0;
  p.seg_exact = seg_mse_f32 == 2 ? 1 : 0;
  p.sel_mode = SEL_FEED_OFF;
  b.sweep = static_cast<const SweepItem*>(table_dev);
  b.sweep_count = count;
  b.sweep_partials = static_cast<double*>(workspace);
  hipEvent_t prof_e0 = nullptr, prof_e1 = nullptr;
  if (prof_on) {
    if (int rc = prof_next_pair(&prof_e0, &prof_e1)) return rc;
    MQ_HIP(hipEventRecord(prof_e0, st));
  }
  const dim3 grid((unsigned)total_blocks);
  switch (fmt) {
    case MQ_INT_SYM: mq_launch_fmt1(SEL_STATS_ONLY, scope, nv, 256, team_log2, grid, 0, st, b); break;
    case MQ_INT_ASYM: mq_launch_fmt2(SEL_STATS_ONLY, scope, nv, 256, team_log2, grid, 0, st, b); break;
    case MQ_LUT_FP: mq_launch_fmt3(SEL_STATS_ONLY, scope, nv, 256, team_log2, grid, 0, st, b); break;
    default: mq_launch_fmt4(SEL_STATS_ONLY, scope, nv, 256, team_log2, grid, 0, st, b); break;
  }
  MQ_HIP(hipGetLastError());
  if (prof_on) MQ_HIP(hipEventRecord(prof_e1, st));
  hipLaunchKernelGGL(k_finalize_sweep, dim3((unsigned)count), dim3(256), 0, st, b);
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}

#include "mq_feed_host.h"
#include "mq_np_host.h"
#include "mq_multi_host.h"

extern "C" size_t mq_tensor_mse_numpy_workspace_bytes(int64_t rows, int64_t cols) {
  const int64_t n = rows * cols;
  return kWsHeader + (size_t)((n + 8191) / 8192 < 1 ? 1 : (n + 8191) / 8192) * sizeof(float);
}
extern "C" int mq_tensor_mse_numpy(const void* w, int32_t w_dtype, int64_t rows, int64_t cols, const mq_quant_cfg* cfg,
                                   const void* scale, const void* zero_point, float* mse_out, void* workspace,
                                   size_t workspace_bytes, void* hip_stream) {
  if (!w || !cfg || !scale || !mse_out || rows < 1 || cols < 1) return MQ_ERR_ARG;
  if (w_dtype != MQ_F16 && w_dtype != MQ_BF16) return w_dtype == MQ_F32 ? MQ_ERR_UNSUPPORTED : MQ_ERR_ARG;
  const bool is_lut = cfg->format == MQ_LUT_FP || cfg->format == MQ_LUT_FP_SUB;
  if (cfg->format != MQ_INT_SYM && cfg->format != MQ_INT_ASYM && !is_lut) return MQ_ERR_ARG;
  if (!is_lut && (cfg->bits < 2 || cfg->bits > 8)) return MQ_ERR_UNSUPPORTED;
  if (cfg->f32_math) return MQ_ERR_UNSUPPORTED;
  if (cfg->format == MQ_INT_ASYM && !zero_point) return MQ_ERR_ARG;
  if (cfg->group_size > 0 && cols % cfg->group_size != 0) return MQ_ERR_GROUP;
  const int64_t n = rows * cols;
  const int64_t L = cfg->group_size == -1 ? n : (cfg->group_size <= 0 ? cols : cfg->group_size);
  if (L % 8 || !aligned(w, 16)) return MQ_ERR_UNSUPPORTED;  // a 16-byte vector must lie inside one segment
  if (!aligned(scale, 2) || !aligned(zero_point, 2) || !aligned(mse_out, 4) || !aligned(workspace, 16)) return MQ_ERR_ALIGN;
  if (!workspace || workspace_bytes < mq_tensor_mse_numpy_workspace_bytes(rows, cols)) return MQ_ERR_WORKSPACE;
  QuantBatch b = {};
  QuantParams& p = b.common;
  p.in_dtype = w_dtype;
  p.n = n;
  p.L = L;
  p.nseg = n / L;
  p.vecs_per_seg = (int32_t)(L / 8 > INT32_MAX ? INT32_MAX : L / 8);
  const int fmt = cfg->format;
  if (fmt == MQ_INT_SYM) p.qmax = (float)((1 << (cfg->bits - 1)) - 1);
  else if (fmt == MQ_INT_ASYM) p.qmax = (float)((1 << cfg->bits) - 1);
  else if (!lut_consts(cfg, false, &p.lut, &p.qmax)) return MQ_ERR_UNSUPPORTED;
  p.rqmax = 1.0f / p.qmax;
  p.qmax_recip_ok = recip_ok_cached(p.qmax);
  p.need_err = 1;
  p.sel_mode = SEL_FEED_OFF;
  p.tred = static_cast<const uint32_t*>(workspace);
  b.ws_stride = 0;
  b.item[0].w = w;
  b.item[0].scale = static_cast<uint16_t*>(const_cast<void*>(scale));
  b.item[0].zp = static_cast<uint16_t*>(const_cast<void*>(zero_point));
  if (n % 8192 && !numpy_prog(n % 8192, &b.np)) return MQ_ERR_UNSUPPORTED;
  const int64_t nchunks = (n + 8191) / 8192, wb = (nchunks + 3) / 4;
  const dim3 grid((unsigned)(wb < grid_cap() ? wb : grid_cap()));
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  switch (fmt) {
    case MQ_INT_SYM: mq_launch_fmt1(SEL_STATS_ONLY, SC_TENSOR_NP, 4, 256, 0, grid, 0, st, b); break;
    case MQ_INT_ASYM: mq_launch_fmt2(SEL_STATS_ONLY, SC_TENSOR_NP, 4, 256, 0, grid, 0, st, b); break;
    case MQ_LUT_FP: mq_launch_fmt3(SEL_STATS_ONLY, SC_TENSOR_NP, 4, 256, 0, grid, 0, st, b); break;
    default: mq_launch_fmt4(SEL_STATS_ONLY, SC_TENSOR_NP, 4, 256, 0, grid, 0, st, b); break;
  }
  MQ_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_tensor_mse_finish, dim3(1), dim3(256), 0, st,
                     reinterpret_cast<const float*>(static_cast<const char*>(workspace) + kWsHeader), nchunks, n, mse_out);
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}

static int list_mean_launch(const void* const* lists, int32_t count, int64_t n, int64_t run, int64_t stride,
                            void* const* out, bool f64, hipStream_t st, bool sum_only = false) {
  if (count < 0 || n < 1 || run < 1 || (count > 0 && (!lists || !out))) return MQ_ERR_ARG;
  if (run > n || n % run || (run != n && stride < run)) return MQ_ERR_ARG;
  ListBatch b = {};
  b.n = n;
  b.sum_only = sum_only ? 1 : 0;
  b.run = run;
  b.S = stride;
  if (!list_prog(n < 8192 ? 0 : 8192, &b.full) || !list_prog(n % 8192, &b.tail)) return MQ_ERR_UNSUPPORTED;
  for (int32_t done = 0; done < count; done += MQ_LIST_BATCH) {
    const int32_t c = count - done < MQ_LIST_BATCH ? count - done : MQ_LIST_BATCH;
    for (int32_t i = 0; i < c; ++i) {
      if (!lists[done + i] || !out[done + i]) return MQ_ERR_ARG;
      if (!aligned(lists[done + i], 4) || !aligned(out[done + i], f64 ? 8 : 4)) return MQ_ERR_ALIGN;
      b.x[i] = static_cast<const float*>(lists[done + i]);
      b.out[i] = out[done + i];
    }
    if (f64) hipLaunchKernelGGL(k_list_mean_numpy<double>, dim3((unsigned)c), dim3(128), 0, st, b);
    else hipLaunchKernelGGL(k_list_mean_numpy<float>, dim3((unsigned)c), dim3(128), 0, st, b);
    MQ_HIP(hipGetLastError());
  }
  return MQ_OK;
}
extern "C" int mq_list_mean_numpy(const void* const* lists, int32_t count, int64_t n, double* const* out,
                                  void* hip_stream) {
  return list_mean_launch(lists, count, n, n, n, reinterpret_cast<void* const*>(out), true,
                          static_cast<hipStream_t>(hip_stream));
}
extern "C" int mq_slice_sum_numpy_f32(const void* const* firsts, int32_t count, int64_t n, int64_t run, int64_t stride,
                                      float* const* out, void* hip_stream) {
  return list_mean_launch(firsts, count, n, run, stride, reinterpret_cast<void* const*>(out), false,
                          static_cast<hipStream_t>(hip_stream), true);
}
extern "C" int mq_slice_mean_numpy_f32(const void* const* firsts, int32_t count, int64_t n, int64_t run, int64_t stride,
                                       float* const* out, void* hip_stream) {
  return list_mean_launch(firsts, count, n, run, stride, reinterpret_cast<void* const*>(out), false,
                          static_cast<hipStream_t>(hip_stream));
}

extern "C" int mq_sweep_records(const double* totals, const double* p99, const double* numel, const int64_t* perm,
                                int32_t count, double* records, void* hip_stream) {
  if (count < 0 || (count > 0 && (!totals || !p99 || !numel || !perm || !records))) return MQ_ERR_ARG;
  if (count == 0) return MQ_OK;
  if (!aligned(totals, 8) || !aligned(p99, 8) || !aligned(numel, 8) || !aligned(perm, 8) || !aligned(records, 8))
    return MQ_ERR_ALIGN;
  hipLaunchKernelGGL(k_sweep_records, dim3((unsigned)((count + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(hip_stream), totals, p99, numel, perm, (int)count, records);
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}

extern "C" int mq_sweep_table(const double* totals, const double* p99, const int64_t* p99_index, const double* numel,
                              const int64_t* src, int32_t rows, double* table, void* hip_stream) {
  if (rows < 0 || (rows > 0 && (!totals || !p99 || !numel || !table))) return MQ_ERR_ARG;
  if (rows == 0) return MQ_OK;
  if (!aligned(totals, 8) || !aligned(p99, 8) || !aligned(p99_index, 8) || !aligned(numel, 8) || !aligned(src, 8) ||
      !aligned(table, 8))
    return MQ_ERR_ALIGN;
  hipLaunchKernelGGL(k_sweep_table, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(hip_stream), totals, p99, p99_index, numel, src, (int)rows, table);
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}

extern "C" int mq_quantize(const void* w, int32_t w_dtype, int64_t rows, int64_t cols, const mq_quant_cfg* cfg,
                           const mq_quant_out* out, void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!out) return MQ_ERR_ARG;
  mq_quant_item item = {w, *out};
  return mq_quantize_batch(&item, 1, w_dtype, rows, cols, cfg, workspace, workspace_bytes, hip_stream);
}

// ---- quantize + exact percentile of the float32 |error| in the same pass ---------------------------------
static size_t sel_slices_offset(int32_t count) {
  return ((size_t)(count > MQ_MAX_BATCH ? MQ_MAX_BATCH : (count < 1 ? 1 : count)) * ws_slice_bytes() + 255) / 256 * 256;
}
// uint16 elements per tensor of the sample area: 1/32 of the tensor, rounded up to 128 bytes
static int64_t sample_stride_elems(int64_t n) { return (n / 32 + 63) / 64 * 64; }
extern "C" size_t mq_quantize_p99_batch_workspace_bytes(int32_t count, int64_t rows, int64_t cols, const mq_quant_cfg*) {
  const int32_t c = count > MQ_MAX_BATCH ? MQ_MAX_BATCH : (count < 1 ? 1 : count);
  return sel_slices_offset(count) + (size_t)c * sel_ws_bytes() + (size_t)c * sample_stride_elems(rows * cols) * 2;
}

extern "C" int mq_quantize_p99_batch(const mq_quant_item* items, int32_t count, int32_t w_dtype, int64_t rows,
                                     int64_t cols, const mq_quant_cfg* cfg, double pct, float* p99,
                                     void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!cfg || count < 0 || rows < 0 || cols < 0 || (count > 0 && !items) || !p99) return MQ_ERR_ARG;
  if (!(pct >= 0.0 && pct <= 100.0)) return MQ_ERR_ARG;
  if (w_dtype != MQ_F16 && w_dtype != MQ_BF16 && w_dtype != MQ_F32) return MQ_ERR_ARG;
  const bool is_lut = cfg->format == MQ_LUT_FP || cfg->format == MQ_LUT_FP_SUB;
  if (cfg->format != MQ_INT_SYM && cfg->format != MQ_INT_ASYM && !is_lut) return MQ_ERR_ARG;
  if (!is_lut && (cfg->bits < 2 || cfg->bits > 8)) return MQ_ERR_UNSUPPORTED;
  if (cfg->f32_math && !(is_lut && cfg->group_size == -1)) return MQ_ERR_UNSUPPORTED;
  if (cfg->group_size > 0 && cols % cfg->group_size != 0) return MQ_ERR_GROUP;
  const int64_t n = rows * cols;
  if (count == 0) return MQ_OK;
  if (n == 0) return MQ_ERR_ARG;  // no percentile of an empty array
  for (int32_t i = 0; i < count; ++i) {
    // the stored float32 |error| array is what the exact two-pass route reads when the bracket misses
    if (!items[i].w || !items[i].out.abs_err || items[i].out.abs_err_dtype != MQ_F32) return MQ_ERR_ARG;
    if (!aligned(items[i].out.abs_err, 16)) return MQ_ERR_ALIGN;
  }
  if (!workspace || workspace_bytes < mq_quantize_p99_batch_workspace_bytes(count, rows, cols, cfg)) return MQ_ERR_WORKSPACE;
  if (!aligned(workspace, 16)) return MQ_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  char* sel = static_cast<char*>(workspace) + sel_slices_offset(count);
  const int32_t cmax = count > MQ_MAX_BATCH ? MQ_MAX_BATCH : count;
  uint16_t* samples = reinterpret_cast<uint16_t*>(sel + (size_t)cmax * sel_ws_bytes());
  const int64_t sstride = sample_stride_elems(n);
  for (int32_t done = 0; done < count; done += MQ_MAX_BATCH) {
    const int32_t c = count - done < MQ_MAX_BATCH ? count - done : MQ_MAX_BATCH;
    const mq_quant_item* it = items + done;
    int64_t m = 0;
    int rc = quantize_launch(it, c, w_dtype, rows, cols, cfg, workspace, st, SEL_FEED_OFF, nullptr, &m);
    if (rc) return rc;
    const void* errs[MQ_MAX_BATCH];
    for (int i = 0; i < c; ++i) errs[i] = it[i].out.abs_err;
    if (m > 0 && sel_feed_eligible(n, m, pct)) {
      if ((rc = sel_feed_begin(sel, c, st))) return rc;
      if ((rc = quantize_launch(it, c, w_dtype, rows, cols, cfg, workspace, st, SEL_FEED_SAMPLE, sel, nullptr, samples,
                                sstride)))
        return rc;
      if ((rc = sel_feed_bracket(sel, c, samples, sstride, n, m, pct, st))) return rc;
      if ((rc = quantize_launch(it, c, w_dtype, rows, cols, cfg, workspace, st, SEL_FEED_COUNT, sel))) return rc;
      if ((rc = sel_feed_finish(sel, c, errs, n, pct, p99 + done, st))) return rc;
    } else {  // small tensors, quantiles at the very ends, per-tensor or unaligned shapes: select from the dump
      if ((rc = quantize_launch(it, c, w_dtype, rows, cols, cfg, workspace, st))) return rc;
      int64_t ns[MQ_MAX_BATCH];
      for (int i = 0; i < c; ++i) ns[i] = n;
      if ((rc = mq_percentile_multi(errs, ns, c, MQ_F32, pct, p99 + done, nullptr, sel, (size_t)c * sel_ws_bytes(),
                                    hip_stream)))
        return rc;
    }
  }
  return MQ_OK;
}
// ---- several passes over same-shaped stacks enqueued together ---------------------------------------------
// What a pass's selection needs in front of its full quantize pass (clear, sampling pass, sample histogram, bracket)
// and behind it (bracket scan, candidates, guarded exact route, list percentile) is a chain of short dependent
// kernels.  With the passes of a sweep known together:
//  * the front chains of all passes run first, and the kernels that are per selection slice (clear, sample
//    histogram, sample scan, bracket clear) take the slices of ALL passes in one launch each;
//  * behind a full pass, the kernels with small workgroups (partial sums, bracket scan, candidates) go to a second,
//    library-owned stream and run beside the next full pass, so the caller's stream carries the full passes back to
//    back;
//  * the kernels that need a CU to themselves (128 KiB histograms: the guarded exact route, the list percentiles)
//    wait for the last full pass and again take all passes' slices per launch.
// (Measured on 16 x 4096 x 4096 float16, three group sizes: front chains beside a full pass slowed it by as much as
//  they took alone, and a 1024-thread / 128 KiB workgroup beside a full pass is not scheduled until that pass drains.)
#define MQ_P99_MAX_UNITS 24
namespace {
struct PassStreams {
  hipStream_t post = nullptr;
  hipEvent_t done = nullptr, main_ev[MQ_P99_MAX_UNITS] = {};
};
std::mutex g_pass_mu;  // held while a call enqueues: the stream and events of a device are shared by its callers
PassStreams* pass_streams(int* rc) {
  static PassStreams per_dev[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { *rc = MQ_ERR_HIP; return nullptr; }
  PassStreams& ps = per_dev[dev];
  if (!ps.post) {
    hipStream_t a = nullptr;
    if (hipStreamCreateWithFlags(&a, hipStreamNonBlocking) != hipSuccess) { *rc = MQ_ERR_HIP; return nullptr; }
    bool ok = hipEventCreateWithFlags(&ps.done, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < MQ_P99_MAX_UNITS; ++i)
      ok = hipEventCreateWithFlags(&ps.main_ev[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) { *rc = MQ_ERR_HIP; return nullptr; }
    ps.post = a;
  }
  return &ps;
}
size_t pass_unit_bytes(int32_t c, int64_t rows, int64_t cols) {
  return sel_slices_offset(c) + (size_t)c * sel_ws_bytes() + ((size_t)c * sample_stride_elems(rows * cols) * 2 + 255) / 256 * 256;
}
}  // namespace

extern "C" size_t mq_quantize_p99_passes_workspace_bytes(int32_t npass, int32_t count, int64_t rows, int64_t cols) {
  if (npass < 1 || count < 1) return 0;
  const int64_t chunks = (count + MQ_MAX_BATCH - 1) / MQ_MAX_BATCH;
  return (size_t)npass * (size_t)chunks * pass_unit_bytes(count < MQ_MAX_BATCH ? count : MQ_MAX_BATCH, rows, cols);
}

extern "C" int mq_quantize_p99_passes(const mq_p99_pass* passes, int32_t npass, int32_t count, int32_t w_dtype,
                                      int64_t rows, int64_t cols, double pct, double list_pct, void* workspace,
                                      size_t workspace_bytes, void* hip_stream) {
  if (npass < 0 || count < 0 || rows < 0 || cols < 0 || (npass > 0 && !passes)) return MQ_ERR_ARG;
  if (!(pct >= 0.0 && pct <= 100.0) || !(list_pct >= 0.0 && list_pct <= 100.0)) return MQ_ERR_ARG;
  if (w_dtype != MQ_F16 && w_dtype != MQ_BF16 && w_dtype != MQ_F32) return MQ_ERR_ARG;
  if (npass == 0 || count == 0) return MQ_OK;
  const int64_t n = rows * cols;
  if (n == 0) return MQ_ERR_ARG;
  const int32_t chunks = (count + MQ_MAX_BATCH - 1) / MQ_MAX_BATCH;
  if ((int64_t)npass * chunks > MQ_P99_MAX_UNITS) return MQ_ERR_UNSUPPORTED;
  for (int32_t k = 0; k < npass; ++k) {
    const mq_p99_pass& ps = passes[k];
    if (!ps.items || !ps.cfg || !ps.p99) return MQ_ERR_ARG;
    const mq_quant_cfg* cfg = ps.cfg;
    const bool is_lut = cfg->format == MQ_LUT_FP || cfg->format == MQ_LUT_FP_SUB;
    if (cfg->format != MQ_INT_SYM && cfg->format != MQ_INT_ASYM && !is_lut) return MQ_ERR_ARG;
    if (!is_lut && (cfg->bits < 2 || cfg->bits > 8)) return MQ_ERR_UNSUPPORTED;
    if (cfg->f32_math && !(is_lut && cfg->group_size == -1)) return MQ_ERR_UNSUPPORTED;
    if (cfg->group_size > 0 && cols % cfg->group_size != 0) return MQ_ERR_GROUP;
    for (int32_t i = 0; i < count; ++i) {
      const mq_quant_out& o = ps.items[i].out;
      if (!ps.items[i].w || !o.abs_err || o.abs_err_dtype != MQ_F32) return MQ_ERR_ARG;
      if (!aligned(o.abs_err, 16)) return MQ_ERR_ALIGN;
      // the list percentile reads the float32 per-segment mean squared errors the pass writes
      if (ps.list_p99 && !(o.seg_sse && o.seg_mse_f32 == 1 && aligned(o.seg_sse, 16))) return MQ_ERR_ARG;
    }
  }
  if (!workspace || workspace_bytes < mq_quantize_p99_passes_workspace_bytes(npass, count, rows, cols)) return MQ_ERR_WORKSPACE;
  if (!aligned(workspace, 256)) return MQ_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  std::lock_guard<std::mutex> lock(g_pass_mu);
  int rc = MQ_OK;
  PassStreams* S = pass_streams(&rc);
  if (!S) return rc;
  const int32_t cmax = count < MQ_MAX_BATCH ? count : MQ_MAX_BATCH;
  const int total_units = npass * chunks;
  const int64_t sstride = sample_stride_elems(n);
  // workspace: per unit the launch's reduction slices | per unit the selection slices | per unit the samples
  const size_t red_bytes = sel_slices_offset(cmax), slice_bytes = (size_t)cmax * sel_ws_bytes();
  char* const red0 = static_cast<char*>(workspace);
  char* const sel0 = red0 + (size_t)total_units * red_bytes;
  char* const smp0 = sel0 + (size_t)total_units * slice_bytes;
  struct Unit { const mq_quant_item* it; const mq_quant_cfg* cfg; int32_t c; float* p99; double* lp; char* ws; char* sel;
                uint16_t* samples; bool fed; int64_t m; };
  Unit units[MQ_P99_MAX_UNITS];
  int nu = 0;
  for (int32_t k = 0; k < npass; ++k)
    for (int32_t done = 0; done < count; done += MQ_MAX_BATCH) {
      Unit& u = units[nu];
      u.c = count - done < MQ_MAX_BATCH ? count - done : MQ_MAX_BATCH;
      u.it = passes[k].items + done;
      u.cfg = passes[k].cfg;
      u.p99 = passes[k].p99 + done;
      u.lp = passes[k].list_p99 ? passes[k].list_p99 + done : nullptr;
      u.ws = red0 + (size_t)nu * red_bytes;
      u.sel = sel0 + (size_t)nu * slice_bytes;
      u.samples = reinterpret_cast<uint16_t*>(smp0 + (size_t)nu * (size_t)cmax * (size_t)sstride * 2);
      u.m = 0;
      if ((rc = quantize_launch(u.it, u.c, w_dtype, rows, cols, u.cfg, u.ws, st, SEL_FEED_OFF, nullptr, &u.m))) return rc;
      u.fed = u.m > 0 && sel_feed_eligible(n, u.m, pct);
      ++nu;
    }
  // Units [i0, i1) share the launches that are per selection slice when their slices (and samples) follow each other
  // in the workspace -- one chunk per pass -- and, where results are written, so do the caller's result arrays; at
  // most SEL_MAX_BATCH slices per launch.
  auto group_end = [&](int i0, bool same_m, bool outputs) {
    int i1 = i0 + 1, tot = units[i0].c;
    while (i1 < nu && chunks == 1 && tot + units[i1].c <= SEL_MAX_BATCH) {
      const Unit &a = units[i1 - 1], &c = units[i1];
      if (a.fed != c.fed || (same_m && a.m != c.m)) break;
      if (outputs && !(c.p99 == a.p99 + a.c && (a.lp != nullptr) == (c.lp != nullptr) && (!a.lp || c.lp == a.lp + a.c))) break;
      tot += c.c;
      ++i1;
    }
    return i1;
  };
  // From the first launch on the second stream a failure leaves work enqueued there; the caller's stream is joined
  // to it before returning either way.
  auto join = [&]() {
    (void)hipEventRecord(S->done, S->post);
    (void)hipStreamWaitEvent(st, S->done, 0);
  };
  // 1. the front chains, on the caller's stream
  for (int i0 = 0; i0 < nu;) {
    const int i1 = group_end(i0, true, false);
    if (units[i0].fed) {
      int tot = 0;
      for (int i = i0; i < i1; ++i) tot += units[i].c;
      if ((rc = sel_feed_begin(units[i0].sel, tot, st))) return rc;
      for (int i = i0; i < i1; ++i) {
        const Unit& u = units[i];
        if ((rc = quantize_launch(u.it, u.c, w_dtype, rows, cols, u.cfg, u.ws, st, SEL_FEED_SAMPLE, u.sel, nullptr, u.samples,
                                  sstride)))
          return rc;
      }
      if ((rc = sel_feed_bracket(units[i0].sel, tot, units[i0].samples, sstride, n, units[i0].m, pct, st))) return rc;
    }
    i0 = i1;
  }
  // 2. the full passes back to back; behind each, on the second stream: its partial sums, bracket scan, candidates
  for (int i = 0; i < nu && !rc; ++i) {
    const Unit& u = units[i];
    tl_totals_stream = S->post;
    tl_totals_event = S->main_ev[i];  // (recorded behind the quantize kernel; the second stream waits for it)
    rc = u.fed ? quantize_launch(u.it, u.c, w_dtype, rows, cols, u.cfg, u.ws, st, SEL_FEED_COUNT, u.sel)
               : quantize_launch(u.it, u.c, w_dtype, rows, cols, u.cfg, u.ws, st);
    tl_totals_stream = nullptr;
    tl_totals_event = nullptr;
    if (!rc && u.fed) {
      const void* errs[MQ_MAX_BATCH];
      for (int j = 0; j < u.c; ++j) errs[j] = u.it[j].out.abs_err;
      rc = sel_feed_finish(u.sel, u.c, errs, n, pct, u.p99, S->post, SEL_FINISH_BRACKET);
    }
  }
  // 3. behind the last full pass: the guarded exact route of the fed units, the selections from the stored arrays,
  //    the list percentiles
  for (int i0 = 0; i0 < nu && !rc;) {
    const int i1 = group_end(i0, false, true);
    const void* xs[SEL_MAX_BATCH];
    int64_t ns[SEL_MAX_BATCH];
    int tot = 0;
    for (int i = i0; i < i1; ++i)
      for (int j = 0; j < units[i].c; ++j) { xs[tot] = units[i].it[j].out.abs_err; ns[tot] = n; ++tot; }
    const Unit& u = units[i0];
    rc = u.fed ? sel_feed_finish(u.sel, tot, xs, n, pct, u.p99, S->post, SEL_FINISH_EXACT)
               : mq_percentile_multi(xs, ns, tot, MQ_F32, pct, u.p99, nullptr, u.sel, (size_t)tot * sel_ws_bytes(), S->post);
    if (!rc && u.lp) {
      tot = 0;
      for (int i = i0; i < i1; ++i) {
        const int64_t gs = units[i].cfg->group_size;
        const int64_t L = gs == -1 ? n : (gs <= 0 ? cols : gs);
        for (int j = 0; j < units[i].c; ++j) { xs[tot] = units[i].it[j].out.seg_sse; ns[tot] = n / L; ++tot; }
      }
      rc = mq_percentile_multi(xs, ns, tot, MQ_F32_AS_F64, list_pct, u.lp, nullptr, u.sel, (size_t)tot * sel_ws_bytes(), S->post);
    }
    i0 = i1;
  }
  join();
  if (rc) return rc;
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}
#endif  // MQ_PART == 0
