// Selection workspace layout and the device-side feed that lets another kernel (the fused quantize pass)
// contribute to a selection while the values are still in its registers.  Shared by mq_select.hip (which
// owns the kernels that scan the histograms) and mq_quant.hip.
#pragma once
#include <stddef.h>

#include "mq_common.h"

#define SEL_TOP_BINS 32768
#define SEL_DIGIT_BINS 65536
#define SEL_HASH 2048
#define SEL_MAX_BATCH 64
#define SEL_BR_BINS 16       // bracket histogram: 16 x 65536 entries
#define SEL_BR_MAX_BINS 256  // widest bracket in level-0 bins (then 16 bit patterns share a histogram entry)
#define SEL_BR_HASH 4096
#define SEL_CHUNK_LINES 32   // one 256-byte line of every 32 is sampled
#define SEL_BR_MIN_N (1 << 21)
#define SEL_FEED_HASH 2048   // LDS table of a feeding kernel: keys[2048] + counts[2048] = 16 KiB
#define SEL_CAND_MAX (SEL_BR_BINS * 65536)  // capacity of the candidate list that overlays br_fine
#define SEL_FEED_THREADS 256 // workgroup size of a feeding kernel (four candidate stages share the LDS table)
static_assert((2 * SEL_FEED_HASH) / (SEL_FEED_THREADS / 64) >= 1 + 511 + 8 * 64, "candidate stage of a wave");

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
  uint32_t br_lo, br_nb;     // bracket: first level-0 bin, number of bins (0: none)
  uint32_t ok;               // 1: the result is final; 2: the bracket located both ranks up to 2^br_shift
                             //    bit patterns and the refining pass over the stored array must finish them
  uint32_t br_shift;         // bit patterns per bracket-histogram entry = 2^br_shift (0 for <= 16 bins)
  unsigned long long below;  // exact number of elements under the bracket
  uint32_t ref_base[2];      // ok == 2: first bit pattern of the entry holding rank r ...
  uint32_t cand_n;           // fed selection with a wide bracket (br_shift > 0): br_fine holds the bit patterns of
                             // the values inside the bracket, as a list of cand_n entries (may pass SEL_CAND_MAX:
                             // then the list overflowed and is not used)
  uint32_t pad1;
  unsigned long long ref_rank[2];  // ... and the rank inside that entry
  uint32_t refine[2][16];    // counts per bit pattern inside those entries (filled by the refining pass)
  uint32_t pad[8];
  uint32_t top_coarse[SEL_TOP_BINS / 256];
  uint32_t top[SEL_TOP_BINS];
  SelLevel lvl[3];  // cleared once per selection, only as many as the dtype needs
  uint32_t samp_coarse[SEL_TOP_BINS / 256];
  uint32_t samp[SEL_TOP_BINS];
  uint32_t br_coarse[SEL_BR_BINS * 256];
  uint32_t br_fine[SEL_BR_BINS * 65536];
};
static_assert(offsetof(SelWs, top_coarse) == 256, "header is 256 bytes");

// Which of 32 consecutive units (256-byte lines of an array, tiles or rows of a weight) is sampled.
__host__ __device__ __forceinline__ uint32_t sample_line(uint32_t c) { return (c * 0x9E3779B1u) >> 27; }

// ---- host side (mq_select.hip): the selection steps around a feeding kernel ------------------------------
// float32 |values|, `count` <= 32 arrays of n values each, m of them sampled by the feeding kernel's first,
// sampled pass.  Order on one stream:
//   sel_feed_eligible()  -> false: select from the stored array instead (mq_percentile_multi)
//   sel_feed_begin()       clears the slices
//   [feeding kernel, mode SEL_FEED_SAMPLE: stores the top 16 bits of its m sampled values, densely, at
//    samples + i * sample_stride (uint16 units) for array i]
//   sel_feed_bracket()     histogram of the samples, sample ranks -> bracket, clears the bracket histograms
//   [feeding kernel, mode SEL_FEED_COUNT over ALL values]
//   sel_feed_finish()      ranks inside the bracket -> results; otherwise the exact two-pass route over the
//                          stored arrays xs[i] (enqueued always, returns at once when not needed)
enum { SEL_FEED_OFF = 0, SEL_FEED_SAMPLE = 1, SEL_FEED_COUNT = 2 };
size_t sel_ws_bytes();
bool sel_feed_eligible(int64_t n, int64_t m, double pct);
int sel_feed_begin(void* ws, int count, hipStream_t st);
int sel_feed_bracket(void* ws, int count, const uint16_t* samples, int64_t sample_stride, int64_t n, int64_t m,
                     double pct, hipStream_t st);
// part: everything, only the bracket's own resolution (scan + kept candidates), or only the guarded exact route --
// the two halves may be enqueued at different places (the exact route's kernels are large workgroups that do not fit
// beside a running full pass; each returns at once for a slice the bracket resolved)
enum { SEL_FINISH_ALL = 0, SEL_FINISH_BRACKET = 1, SEL_FINISH_EXACT = 2 };
int sel_feed_finish(void* ws, int count, const void* const* xs, int64_t n, double pct, void* out, hipStream_t st,
                    int part = SEL_FINISH_ALL);

// ---- percentile of a per-vector MSE list selected inside the statistics pass (sweep launches) -----------------
// The list of a large tensor is not written: a sampling launch evaluates one wave step of every `unit` steps and
// stores those entries, k_list_bracket takes two sample ranks a safe margin either side of the wanted quantile as
// the bracket [blo, bhi] (float32 bit patterns, inclusive), the pass counts the entries under the bracket and keeps
// the bit patterns inside it (ListFeedState, candidate list), k_list_cand selects the wanted ranks among those.
// The counts are exact: the sample only decides where to look.  A bracket that does not hold both ranks (or a
// candidate list that overflowed) sets `miss`; the repair launches then write that tensor's list after all and
// select from it.
#define LIST_SLOT 64    // kept entries a wave of the pass has room for (it sees ~500 entries, ~2 % are kept)
struct ListFeedState {  // one per fed tensor, device memory; cleared before the sampling launch
  uint32_t ovf;         // a wave of the pass kept more than LIST_SLOT entries
  uint32_t miss;        // written by k_list_cand: the kept entries do not hold both ranks
};
// The pass writes no shared counter: wave w of a tensor (four per workgroup) puts the bit patterns it keeps into
// its own slot (slots[w * LIST_SLOT ..]) and its two counts into wg_cnt[2 w] (entries under the bracket) and
// wg_cnt[2 w + 1] (entries kept) -- wave-uniform registers and plain stores, nothing to wait for under a saturated
// memory system.  k_list_cand adds the counts, packs the slots into one list and selects.
struct ListJob {        // one per fed tensor, device memory, written by the host when the sweep is planned
  const float* samples; // sample entries
  float* cand;          // room for the packed kept entries (ccap)
  const float* list;    // the tensor's full list (written by the repair launch only)
  const uint32_t* wg_cnt;  // [nblk][2]   (nblk: WAVES of the tensor in the pass)
  const uint32_t* slots;   // [nblk][LIST_SLOT]
  uint32_t* brk;        // where the bracket goes: the blo / bhi fields of the tensor's entry in the pass's table
  double gamma;         // numpy's interpolation weight for the full list
  uint64_t k0, k1;      // wanted ranks of the full list
  uint32_t m, s_lo, s_hi;  // sample entries, the sample ranks of the bracket
  uint32_t ccap;        // capacity of cand
  uint32_t n;           // entries of the full list
  uint32_t nblk;        // waves of the tensor in the pass (four per workgroup)
  uint32_t out;         // index of the result (float64)
  uint32_t slot;        // index of the ListFeedState
};
int sel_list_bracket(const ListJob* jobs, int count, hipStream_t st);
int sel_list_cand(const ListJob* jobs, int count, ListFeedState* state, double* out, hipStream_t st);
int sel_list_repair(const ListJob* jobs, int count, const ListFeedState* state, double* out, hipStream_t st);
bool sel_list_ranks(int64_t n, int64_t m, double pct, int64_t* k0, int64_t* k1, double* gamma, int64_t* s_lo, int64_t* s_hi);

// ---- device side: per-lane state of a feeding kernel --------------------------------------------------
// LDS pointers keep their address space in the type: through a generic pointer the compiler merges the table
// update with the global overflow update into ONE flat atomic, which is several times slower than ds_add.
typedef __attribute__((address_space(3))) uint32_t lds_u32;

struct SelFeed {
  lds_u32* hkey;  // LDS [SEL_FEED_HASH]: key + 1, 0 = empty
  lds_u32* hcnt;  // LDS [SEL_FEED_HASH]
  SelWs* ws;
  uint32_t lo, nb, shift;  // bracket (count mode)
  uint32_t cand_words;     // wide bracket: LDS words per wave of the candidate staging area
  uint32_t below;   // this lane's count of values under the bracket
  int mode;
  uint16_t* samples;     // sample mode: where this array's samples go
  int64_t sample_shift;  // sample mode: element offset -> dense sample index (set by the kernel per tile)
};

// lds: SEL_FEED_HASH * 2 words of (dynamic) shared memory; every thread of the workgroup calls this.
// MODE (SEL_FEED_SAMPLE / SEL_FEED_COUNT) is a compile-time parameter of the feeding kernel: the two passes
// are different kernels (and show up under different names in a profile).
template <int MODE>
__device__ __forceinline__ void sel_feed_init(SelFeed& f, SelWs* ws, uint32_t* lds, uint16_t* samples) {
  const int mode = MODE;
  f.mode = mode;
  f.samples = samples;
  f.sample_shift = 0;
  f.ws = ws;
  f.hkey = (lds_u32*)lds;
  f.hcnt = (lds_u32*)lds + SEL_FEED_HASH;
  f.lo = 0;
  f.nb = 0;
  f.shift = 0;
  f.cand_words = 0;
  f.below = 0;
  if (mode == SEL_FEED_COUNT) {
    for (int i = threadIdx.x; i < 2 * SEL_FEED_HASH; i += blockDim.x) lds[i] = 0;
    f.lo = ws->br_lo;
    f.nb = ws->br_nb;
    f.shift = ws->br_shift;
    // (a stage must hold its count word, up to 511 entries left from before and the 8 x 64 of one vector: feeding
    //  kernels are launched with SEL_FEED_THREADS threads, which the host side checks)
    f.cand_words = (2 * SEL_FEED_HASH) / (blockDim.x >> 6);
    __syncthreads();
  }
}

__device__ __forceinline__ void sel_feed_add(SelFeed& f, uint32_t key, uint32_t* gfine, uint32_t* gcoarse) {
  uint32_t h = (key * 2654435761u) >> 21;  // 11 bits
  bool placed = false;
#pragma unroll 1
  for (int probe = 0; probe < 32 && !placed; ++probe) {
    // a plain read first: once a key is in the table (the common case, values repeat heavily) the insert
    // is one fire-and-forget LDS add, no compare-and-swap round trip
    uint32_t cur = f.hkey[h];
    if (cur == 0u) {
      uint32_t expect = 0u;
      __hip_atomic_compare_exchange_strong(&f.hkey[h], &expect, key + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
      cur = expect;  // the previous content
    }
    if (cur == 0u || cur == key + 1u) {
      __hip_atomic_fetch_add(&f.hcnt[h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      placed = true;
    }
    h = (h + 1) & (SEL_FEED_HASH - 1);
  }
  if (!placed) {  // table full: straight to the global histogram
    atomicAdd(&gfine[key], 1u);
    atomicAdd(&gcoarse[key >> 8], 1u);
  }
}

// Wide bracket (more than SEL_BR_BINS level-0 bins: heavy-tailed data): no histogram could hold every bit pattern
// of the bracket, but the values inside it are still only the fraction of a percent that the sample ranks span.
// Their exact bit patterns are kept: each wave stages them in its slice of the LDS area (word 0: count) and
// appends full stages to the list in ws->br_fine with one global atomic; a small kernel then selects among the
// list (k_sel_cand), so the stored array is not read again.
// Called by whichever lanes of the wave are active (the per-element path of a kernel runs under a per-segment
// branch, so lane 0 may be off): the first active lane reserves the list range and resets the stage, the active
// lanes share the copy.  The lanes that are off do not touch the stage meanwhile (they are not executing).
__device__ __forceinline__ void sel_feed_cand_flush(SelFeed& f) {
  lds_u32* reg = f.hkey + (threadIdx.x >> 6) * f.cand_words;
  const uint32_t cnt = reg[0];
  if (cnt == 0u) return;
  const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
  const int lane = threadIdx.x & 63, leader = __builtin_ctzll(act);
  const uint32_t nact = (uint32_t)__builtin_popcountll(act);
  const uint32_t mine = (uint32_t)__builtin_popcountll(act & ((1ull << lane) - 1ull));  // index among the active lanes
  uint32_t base = 0u;
  if (lane == leader) base = atomicAdd(&f.ws->cand_n, cnt);
  base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
  for (uint32_t i = mine; i < cnt; i += nact)
    if (base + i < (uint32_t)SEL_CAND_MAX) f.ws->br_fine[base + i] = reg[1 + i];
  __builtin_amdgcn_wave_barrier();  // (every copy above reads the stage before the count is reset)
  if (lane == leader) reg[0] = 0u;
}
__device__ __forceinline__ void sel_feed_cand_add(SelFeed& f, uint32_t bits) {
  lds_u32* reg = f.hkey + (threadIdx.x >> 6) * f.cand_words;
  const uint32_t pos = __hip_atomic_fetch_add(&reg[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  reg[1 + pos] = bits;  // (room for a whole vector of every lane was made before the vector: sel_feed8)
}

// Eight float32 values (their sign is ignored) of one lane, elements e0 .. e0 + 7 of the array.
// Sample mode: the top 16 bits of each go to the dense sample array with one 16-byte store (a small kernel
// histograms them afterwards: a direct LDS histogram is far cheaper than table inserts for EVERY value).
// Count mode: the compare part is straight-line code; the table inserts of the few values inside the bracket
// run in ONE loop (a wave iterates as often as its busiest lane has values to insert).
template <int MODE>
__device__ __forceinline__ void sel_feed8(SelFeed& f, const float* e, int64_t e0) {
  uint32_t a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = f_bits(e[i]) & 0x7fffffffu;
  if constexpr (MODE == SEL_FEED_SAMPLE) {
    u32x4 pk;
#pragma unroll
    for (int k = 0; k < 4; ++k) pk[k] = (a[2 * k] >> 16) | (a[2 * k + 1] & 0xffff0000u);
    *reinterpret_cast<u32x4*>(f.samples + e0 + f.sample_shift) = pk;
    return;
  }
  const uint32_t lo_bits = f.lo << 16, span = f.nb << 16;
  uint32_t mask = 0u;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    f.below += a[i] < lo_bits ? 1u : 0u;
    mask |= (a[i] - lo_bits < span ? 1u : 0u) << i;  // the subtraction wraps for values under the bracket
  }
  const bool wide = f.shift != 0u;  // (uniform over the workgroup: a property of the tensor's bracket)
  if (wide && __builtin_amdgcn_ballot_w64(mask != 0u) != 0ull) {
    // the wave is about to add up to 8 x 64 entries: empty the stage if they might not fit
    lds_u32* reg = f.hkey + (threadIdx.x >> 6) * f.cand_words;
    if (reg[0] + 512u > f.cand_words - 1u) sel_feed_cand_flush(f);
  }
#pragma unroll 1
  while (mask) {
    const int i = __builtin_ctz(mask);
    mask &= mask - 1u;
    // a[i] without indexing the register array
    const uint32_t p0 = (i & 1) ? a[1] : a[0], p1 = (i & 1) ? a[3] : a[2], p2 = (i & 1) ? a[5] : a[4],
                   p3 = (i & 1) ? a[7] : a[6];
    const uint32_t q0 = (i & 2) ? p1 : p0, q1 = (i & 2) ? p3 : p2;
    const uint32_t bits = (i & 4) ? q1 : q0;
    if (wide) sel_feed_cand_add(f, bits);
    else sel_feed_add(f, bits - lo_bits, f.ws->br_fine, f.ws->br_coarse);
  }
}

// End of the feeding kernel: every thread of the workgroup calls this once.
template <int MODE>
__device__ __forceinline__ void sel_feed_flush(SelFeed& f) {
  if constexpr (MODE != SEL_FEED_COUNT) return;
  __shared__ uint32_t wcnt[16];
  uint32_t b = f.below;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) b += __shfl_xor(b, off);
  if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = b;
  __syncthreads();  // all table inserts of the workgroup are done
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += wcnt[i];
    if (t) atomicAdd(&f.ws->below, t);
  }
  if (f.shift != 0u) {  // wide bracket: the LDS area holds the waves' candidate stages, not a table
    sel_feed_cand_flush(f);
    return;
  }
  // table entries -> registers, then the table's memory becomes the coarse counters of this workgroup
  uint32_t k[8], c[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int i = threadIdx.x + j * blockDim.x;
    k[j] = i < SEL_FEED_HASH ? f.hkey[i] : 0u;
    c[j] = i < SEL_FEED_HASH ? f.hcnt[i] : 0u;
  }
  __syncthreads();
  lds_u32* lco = f.hkey;  // [2 * SEL_FEED_HASH] >= SEL_BR_BINS * 256 counters
  for (int i = threadIdx.x; i < 2 * SEL_FEED_HASH; i += blockDim.x) lco[i] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (k[j]) {
      const uint32_t key = k[j] - 1u;
      atomicAdd(&f.ws->br_fine[key], c[j]);
      __hip_atomic_fetch_add(&lco[key >> 8], c[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  const uint32_t ncoarse = (f.nb << 16) >> 8;  // <= SEL_BR_BINS * 256
  for (uint32_t i = threadIdx.x; i < ncoarse; i += blockDim.x)
    if (lco[i]) atomicAdd(&f.ws->br_coarse[i], lco[i]);
}
static_assert(2 * SEL_FEED_HASH >= SEL_BR_BINS * 256, "coarse overlay fits the table");
