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
#define SEL_BR_BINS 16       // widest bracket, in level-0 bins
#define SEL_BR_HASH 4096
#define SEL_CHUNK_LINES 32   // one 256-byte line of every 32 is sampled
#define SEL_BR_MIN_N (1 << 21)
#define SEL_FEED_HASH 2048   // LDS table of a feeding kernel: keys[2048] + counts[2048] = 16 KiB

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
  uint32_t ok, pad0;         // 1: the bracket pass produced the result
  unsigned long long below;  // exact number of elements under the bracket
  uint32_t pad[48];
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
//   [feeding kernel, mode SEL_FEED_SAMPLE]
//   sel_feed_bracket()     sample ranks -> bracket, clears the bracket histograms
//   [feeding kernel, mode SEL_FEED_COUNT over ALL values]
//   sel_feed_finish()      ranks inside the bracket -> results; otherwise the exact two-pass route over the
//                          stored arrays xs[i] (enqueued always, returns at once when not needed)
enum { SEL_FEED_OFF = 0, SEL_FEED_SAMPLE = 1, SEL_FEED_COUNT = 2 };
size_t sel_ws_bytes();
bool sel_feed_eligible(int64_t n, int64_t m, double pct);
int sel_feed_begin(void* ws, int count, hipStream_t st);
int sel_feed_bracket(void* ws, int count, int64_t n, int64_t m, double pct, hipStream_t st);
int sel_feed_finish(void* ws, int count, const void* const* xs, int64_t n, double pct, void* out, hipStream_t st);

// ---- device side: per-lane state of a feeding kernel --------------------------------------------------
struct SelFeed {
  uint32_t* hkey;  // LDS [SEL_FEED_HASH]: key + 1, 0 = empty
  uint32_t* hcnt;  // LDS [SEL_FEED_HASH]
  SelWs* ws;
  uint32_t lo, nb;  // bracket (count mode)
  uint32_t below;   // this lane's count of values under the bracket
  int mode;
};

// lds: SEL_FEED_HASH * 2 words of (dynamic) shared memory; every thread of the workgroup calls this.
__device__ __forceinline__ void sel_feed_init(SelFeed& f, int mode, SelWs* ws, uint32_t* lds) {
  f.mode = mode;
  f.ws = ws;
  f.hkey = lds;
  f.hcnt = lds + SEL_FEED_HASH;
  f.lo = 0;
  f.nb = 0;
  f.below = 0;
  if (mode != SEL_FEED_OFF) {
    for (int i = threadIdx.x; i < 2 * SEL_FEED_HASH; i += blockDim.x) lds[i] = 0;
    if (mode == SEL_FEED_COUNT) {
      f.lo = ws->br_lo;
      f.nb = ws->br_nb;
    }
    __syncthreads();
  }
}

__device__ __forceinline__ void sel_feed_add(SelFeed& f, uint32_t key, uint32_t* gfine, uint32_t* gcoarse) {
  uint32_t h = (key * 2654435761u) >> 21;  // 11 bits
  for (int probe = 0; probe < 32; ++probe) {
    uint32_t old = atomicCAS(&f.hkey[h], 0u, key + 1u);
    if (old == 0u || old == key + 1u) {
      atomicAdd(&f.hcnt[h], 1u);
      return;
    }
    h = (h + 1) & (SEL_FEED_HASH - 1);
  }
  atomicAdd(&gfine[key], 1u);  // table full: straight to the global histogram
  atomicAdd(&gcoarse[key >> 8], 1u);
}

// Eight float32 values (their sign is ignored) of one lane.
__device__ __forceinline__ void sel_feed8(SelFeed& f, const float* e) {
  if (f.mode == SEL_FEED_SAMPLE) {
#pragma unroll
    for (int i = 0; i < 8; ++i) sel_feed_add(f, (f_bits(e[i]) >> 16) & 0x7fffu, f.ws->samp, f.ws->samp_coarse);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t bits = f_bits(e[i]) & 0x7fffffffu, top = bits >> 16;
      f.below += top < f.lo ? 1u : 0u;
      const uint32_t t = top - f.lo;  // wraps for bins under the bracket
      if (t < f.nb) sel_feed_add(f, (t << 16) | (bits & 0xffffu), f.ws->br_fine, f.ws->br_coarse);
    }
  }
}

// End of the feeding kernel: every thread of the workgroup calls this once.
__device__ __forceinline__ void sel_feed_flush(SelFeed& f) {
  if (f.mode == SEL_FEED_OFF) return;
  __shared__ uint32_t wcnt[16];
  if (f.mode == SEL_FEED_COUNT) {
    uint32_t b = f.below;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) b += __shfl_xor(b, off);
    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = b;
  }
  __syncthreads();  // all table inserts of the workgroup are done
  if (f.mode == SEL_FEED_COUNT && threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += wcnt[i];
    if (t) atomicAdd(&f.ws->below, t);
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
  uint32_t* lco = f.hkey;  // [2 * SEL_FEED_HASH] >= SEL_BR_BINS * 256 counters
  for (int i = threadIdx.x; i < 2 * SEL_FEED_HASH; i += blockDim.x) lco[i] = 0;
  __syncthreads();
  uint32_t* gfine = f.mode == SEL_FEED_SAMPLE ? f.ws->samp : f.ws->br_fine;
  uint32_t* gcoarse = f.mode == SEL_FEED_SAMPLE ? f.ws->samp_coarse : f.ws->br_coarse;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (k[j]) {
      const uint32_t key = k[j] - 1u;
      atomicAdd(&gfine[key], c[j]);
      atomicAdd(&lco[key >> 8], c[j]);
    }
  }
  __syncthreads();
  const uint32_t ncoarse = f.mode == SEL_FEED_SAMPLE ? SEL_TOP_BINS / 256 : f.nb * 256;
  for (uint32_t i = threadIdx.x; i < ncoarse; i += blockDim.x)
    if (lco[i]) atomicAdd(&gcoarse[i], lco[i]);
}
static_assert(2 * SEL_FEED_HASH >= SEL_BR_BINS * 256, "coarse overlay fits the table");
