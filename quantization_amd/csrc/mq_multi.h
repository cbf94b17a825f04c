// Several groupings of ONE format from one read of the weights (included by mq_quant.hip).
//
// A model sweep quantizes the same weights once per (format, grouping): per row, groups of 32, 64, 128 elements --
// four passes that each read the tensor and spend most of their time on instruction issue.  This kernel holds a row
// in registers and computes all four while it is there.  Layout: a lane holds RUNS of 32 consecutive elements
// (four 16-byte vectors: run (r * WPR + w) * 64 + l of the row for lane l of wave w, r < NR), so a group of 32 is a
// lane's own run, a group of 64 a pair of lanes, a group of 128 a quad, the row all lanes of the row's WPR waves:
// the groups' maxima and sums need at most two DPP steps, and the segment parameters -- two reciprocals, a dozen
// conversions: as much issue time as quantizing a vector -- are computed once per run and grouping, i.e. once per
// four vectors.  (With one vector per lane, as the row kernels hold a row, they come due once per vector: measured
// 9.1 ms for the four groupings of the Llama-3-8B sweep against 4 x 2.9 ms for four passes.)
// Per grouping the arithmetic is that of the one-grouping kernels (stats_params, stats_vectors; quant8_slow for
// segments off the fast path), and a group's sum is taken in the same order as k_stats_group takes it: the lists of
// the group configurations have the same bits.  Rows and the tensors' totals are summed in another order (another
// assignment of vectors to lanes), inside the 1e-6 bar of the float64-sum mode like theirs.
//
// Outputs per tensor: for each wanted grouping the float32 list of per-segment mean squared errors and the tensor's
// (sum err^2, sum |err|).  Slots: 0 rows, 1 groups of 32, 2 groups of 64, 3 groups of 128.
#pragma once

struct MultiItem {  // one tensor of a launch; the table lives in device memory, sorted by blk0
  const void* w;
  float* list[4];     // per slot: the list, or null
  double* totals[4];  // per slot: two doubles, or null
  int64_t nrow;
  uint32_t blk0, nblk;
};
static_assert(sizeof(MultiItem) == 88, "table entry");

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), in order
template <int N, int I = 0, class F>
__device__ __forceinline__ void multi_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    multi_static_for<N, I + 1>(f);
  }
}

// (second launch bound: the table formats' widest shape sits at the edge of three waves per SIMD -- 168 registers)
template <int FMT, bool BF16, int NR, int WPR>
__global__ __launch_bounds__(256, (MQ_IS_LUT(FMT) && NR == 4 && WPR == 4) ? 3 : 1) void k_stats_multi(const QuantBatch b) {
  QuantParams p = b.common;  // L: the row length
  __shared__ double lds[32];
  __shared__ uint32_t xr[2][4];
  __shared__ double xs[2][4];
  const MultiItem* it;
  {
    int lo = 0, hi = b.multi_count - 1;
    while (lo < hi) {  // the last entry whose first workgroup is not past this one (uniform: scalar loads)
      const int mid = (lo + hi + 1) >> 1;
      if (b.multi[mid].blk0 <= blockIdx.x) lo = mid;
      else hi = mid - 1;
    }
    it = &b.multi[lo];
  }
  p.w = it->w;
  p.nseg = it->nrow;
  p.n = it->nrow * p.L;
  p.bx = blockIdx.x - it->blk0;
  p.gx = it->nblk;
  float* const list0 = it->list[0];
  float* const list1 = it->list[1];
  float* const list2 = it->list[2];
  float* const list3 = it->list[3];
  // which groupings the launch computes (uniform over the launch)
  const bool on0 = (b.multi_mask & 1u) != 0, on1 = (b.multi_mask & 2u) != 0, on2 = (b.multi_mask & 4u) != 0,
             on3 = (b.multi_mask & 8u) != 0;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t first = WPR == 1 ? (int64_t)p.bx * 4 + wv : (int64_t)p.bx;
  const int64_t stride = WPR == 1 ? (int64_t)p.gx * 4 : (int64_t)p.gx;
  const int sub = WPR == 1 ? 0 : wv;
  const char* wb = static_cast<const char*>(p.w);
  const int64_t row_bytes = p.L * 2;
  const int runs = (int)(p.L / 32);                       // runs per row
  const uint32_t loff = (uint32_t)(sub * 4096 + lane * 64);
  const bool whole = runs == 64 * NR * WPR;               // no run of the grid lies past the row's end
  // Rows that do not fill the grid leave whole wave-runs empty (14336 elements: wave 3's second run; 28672: the
  // fourth run of waves 2 and 3; 11008: the second run of waves 2 and 3): a wave skips the arithmetic of an empty
  // run (uniform branch; with one wave per row the grid's last run is never empty).  Measured on the fused int8
  // launch: 14336 columns -5.6 %, 28672 -4.8 %, 11008 -14 %.
  const bool skip_empty = !(b.multi_mask & 0x40000000u);  // (bit 30: compute them all the same, for measurements)
  auto empty = [&](int r) __attribute__((always_inline)) {
    return WPR > 1 && !whole && skip_empty && (r * WPR + sub) * 64 >= runs;
  };
  const double row_scale = 1.0 / (double)p.L;
  double sse0 = 0.0, sse1 = 0.0, sse2 = 0.0, sse3 = 0.0, abs0 = 0.0, abs1 = 0.0, abs2 = 0.0, abs3 = 0.0;
  int parity = 0;
  bool seen_slow = false;
  const __amdgpu_buffer_rsrc_t rs = stats_rsrc(p);

  // this lane's runs of a row; zeros for runs past the row's end
  auto issue = [&](int64_t row, u32x4* dst) __attribute__((always_inline)) {
    const uint32_t base = (uint32_t)(row * row_bytes) + loff;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const uint32_t off = (whole || (r * WPR + sub) * 64 + lane < runs) ? base + r * (4096 * WPR) : MQ_STATS_OOB;
#pragma unroll
      for (int k = 0; k < 4; ++k) dst[4 * r + k] = stats_bload<false>(rs, off == MQ_STATS_OOB ? off : off + 16 * k);
    }
  };
  // reduction word of one run; a run past the row's end is neutral (its zeros must not enter min / max)
  auto run_red = [&](const u32x4* v, bool valid) __attribute__((always_inline)) {
    const uint32_t r = stats_lane_reduce<FMT, 4>(v);
    if constexpr (FMT == MQ_INT_ASYM) return valid ? r : 0xfc007c00u;
    else return r;  // (absolute maximum: zeros change nothing)
  };
  auto row_params = [&](uint32_t red) __attribute__((always_inline)) {
    red = team_reduce_u32<64>(red, stats_op<FMT>);
    if constexpr (WPR > 1) {
      if (lane == 0) xr[parity][wv] = red;
      __syncthreads();
#pragma unroll
      for (int i = 0; i < WPR; ++i) red = stats_op<FMT>(red, xr[parity][i]);
    }
    return stats_params<FMT>(red, p);
  };
  // the row's sum over the waves' lanes -> its list entry
  auto row_finish = [&](int64_t row, double s) __attribute__((always_inline)) {
    if (!list0) return;
    s = team_sum<64>(s);
    if constexpr (WPR > 1) {
      if (lane == 0) xs[parity][wv] = s;
      __syncthreads();
      if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < WPR; ++i) t += xs[parity][i];
        list0[row] = (float)(t * row_scale);
      }
    } else if (lane == 0) {
      list0[row] = (float)(s * row_scale);
    }
  };

  auto step = [&](int64_t row, const u32x4* buf) __attribute__((always_inline)) {
    u32x4 raw[4 * NR];
    uint32_t red[NR];
    uint32_t rowred = FMT == MQ_INT_ASYM ? 0xfc007c00u : 0u;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if (empty(r)) {
        red[r] = FMT == MQ_INT_ASYM ? 0xfc007c00u : 0u;
        continue;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) raw[4 * r + k] = stats_cvt<BF16>(buf[4 * r + k]);
      const bool valid = whole || (r * WPR + sub) * 64 + lane < runs;
      red[r] = run_red(&raw[4 * r], valid);
      rowred = stats_op<FMT>(rowred, red[r]);
    }
    SegP sp0;
    // integer formats: the row's power of two serves every grouping of the row (pk_div_params): v P once per word
    // (rows of up to two runs per lane: with more, the 16 registers of the scaled copy cost a wave of occupancy --
    //  170 VGPRs for NR = 4, WPR = 4 -- and the launch was up to 27 % slower)
    constexpr bool SHARE = !MQ_IS_LUT(FMT) && NR <= 2;
    int kb = MQ_PK_OWN;
    if (on0) {
      sp0 = row_params(rowred);
      seen_slow = seen_slow || sp0.slow;
      if constexpr (SHARE) kb = (sp0.slow || (b.multi_mask & 0x80000000u)) ? MQ_PK_OWN : sp0.pk_k;
    }
    const bool share = kb != MQ_PK_OWN;  // (uniform over the row)
    double rs0 = 0.0;
    // the lists' entries of this row: list + row * entries per row (64-bit once per row, 32-bit offsets below)
    float* const l1 = list1 ? list1 + row * runs : nullptr;
    float* const l2 = list2 ? list2 + row * (runs >> 1) : nullptr;
    float* const l3 = list3 ? list3 + row * (runs >> 2) : nullptr;
    // (unrolled by construction: raw[] must stay in registers, and the table formats' body is past the size up to
    //  which the compiler follows an unroll pragma)
    multi_static_for<NR>([&](auto rc) __attribute__((always_inline)) {
      constexpr int r = decltype(rc)::value;
      if (empty(r)) return;
      const int run = (r * WPR + sub) * 64 + lane;
      const bool valid = whole || run < runs;
      const u32x4* v = &raw[4 * r];
      u32x4 pre[4];
      if (SHARE && share) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int i = 0; i < 4; ++i) pre[k][i] = as_u32(as_h2(v[k][i]) * as_h2(sp0.pk_p));
      }
      // one grouping of the run: on the shared words when every lane's segment took the row's power of two
      auto vectors = [&](const SegP& sp, bool lane_ok, double& s, double& ab) __attribute__((always_inline)) {
        if constexpr (SHARE) {
          if (share && __builtin_amdgcn_ballot_w64(lane_ok && !sp.pk_shared) == 0ull) {
            stats_vectors<FMT, 4, false, true>(p, v, sp, s, ab, nullptr, pre);
            return;
          }
        }
        stats_vectors<FMT, 4>(p, v, sp, s, ab);
      };
      if (on0 && !sp0.slow) {
        double s, ab;
        vectors(sp0, true, s, ab);
        rs0 += s;
        abs0 += ab;
      }
      const uint32_t r1 = red[r];
      const uint32_t r2 = stats_op<FMT>(r1, dpp<DPP_XOR1>(r1));
      const uint32_t r3 = stats_op<FMT>(r2, dpp<DPP_XOR2>(r2));
      // Formats whose parameters come from the absolute maximum alone: the half of a group that holds the group's
      // maximum has the parameters -- and so the errors -- it had in the grouping below, bit for bit; only the other
      // half is quantized again.  Its run (g = 64; its two runs for g = 128) is shared out: the lane that holds it
      // takes its first two vectors, the lane one (two) over in the half that has nothing to compute its last two.
      // A run's sum is double(float32 sum of vectors 0, 1) + double(float32 sum of vectors 2, 3) either way, so the
      // list entries keep their bits; the |error| sums are added in another order (tensor totals, to 1e-6 as rows).
      // own: this lane's half holds the maximum; prev_*: the run's sums in the grouping below
      auto half_again = [&](auto ctrl, bool own, const SegP& sp, double prev_s, double prev_ab, double& s, double& ab)
                            __attribute__((always_inline)) {
        constexpr int CTRL = decltype(ctrl)::value;
        u32x4 x[2];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const uint32_t theirs = dpp<CTRL>(v[2 + k][i]);
            x[k][i] = own ? theirs : v[k][i];
          }
        double sx, abx;
        stats_vectors<FMT, 2>(p, x, sp, sx, abx);
        const double ox = dpp_f64<CTRL>(sx), oab = dpp_f64<CTRL>(abx);
        s = own ? prev_s : sx + ox;
        ab = own ? prev_ab : abx + oab;
      };
      constexpr bool NEST = FMT != MQ_INT_ASYM;
      double run_s = 0.0, run_ab = 0.0;  // this lane's run in the grouping below (before the slow / valid mask)
      if (on1) {
        const SegP sp = stats_params<FMT>(r1, p, kb);
        const bool take = valid && !sp.slow;
        seen_slow = seen_slow || (valid && sp.slow);
        double s, ab;
        vectors(sp, take, s, ab);
        run_s = s;
        run_ab = ab;
        s = take ? s : 0.0;
        sse1 += s;
        abs1 += take ? ab : 0.0;
        if (l1 && take) l1[run] = (float)(s * (1.0 / 32.0));
      }
      if (on2) {
        const SegP sp = stats_params<FMT>(r2, p, kb);
        const bool take = valid && !sp.slow;
        seen_slow = seen_slow || (valid && sp.slow);
        double s, ab;
        if (NEST && on1) half_again(std::integral_constant<int, DPP_XOR1>{}, r1 == r2, sp, run_s, run_ab, s, ab);
        else vectors(sp, take, s, ab);
        run_s = s;
        run_ab = ab;
        s = take ? s : 0.0;
        sse2 += s;
        abs2 += take ? ab : 0.0;
        s = team_sum<2>(s);
        if (l2 && take && (lane & 1) == 0) l2[run >> 1] = (float)(s * (1.0 / 64.0));
      }
      if (on3) {
        const SegP sp = stats_params<FMT>(r3, p, kb);
        const bool take = valid && !sp.slow;
        seen_slow = seen_slow || (valid && sp.slow);
        double s, ab;
        if (NEST && on2) half_again(std::integral_constant<int, DPP_XOR2>{}, r2 == r3, sp, run_s, run_ab, s, ab);
        else vectors(sp, take, s, ab);
        s = take ? s : 0.0;
        sse3 += s;
        abs3 += take ? ab : 0.0;
        s = team_sum<4>(s);
        if (l3 && take && (lane & 3) == 0) l3[run >> 2] = (float)(s * (1.0 / 128.0));
      }
    });
    if (on0 && !sp0.slow) {
      sse0 += rs0;
      row_finish(row, rs0);
    }
    parity ^= 1;
  };

  for (int64_t row = first; row < p.nseg; row += stride) {
    u32x4 buf[4 * NR];
    issue(row, buf);
    step(row, buf);
  }

  // Segments off the fast path (inf / NaN, clamped scales, asymmetric segments on one side of zero): the loop above
  // left them out; here every row is walked again with rolled loops and only those are computed, per element.
  bool any_slow;
  if constexpr (WPR > 1) any_slow = __syncthreads_or(seen_slow) != 0;  // (the row's waves take the barriers below together)
  else any_slow = __builtin_amdgcn_ballot_w64(seen_slow) != 0ull;
  if (__builtin_expect(any_slow, 0)) {
#pragma unroll 1
    for (int64_t row = first; row < p.nseg; row += stride) {
      const char* rowp = wb + row * row_bytes + loff;
      const uint32_t base = (uint32_t)(row * row_bytes) + loff;
      auto load_run = [&](int r, u32x4* v) __attribute__((always_inline)) {
        const bool valid = (r * WPR + sub) * 64 + lane < runs;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          v[k] = stats_cvt<BF16>(stats_bload<false>(rs, valid ? base + r * (4096 * WPR) + 16 * k : MQ_STATS_OOB));
        return valid;
      };
      uint32_t rowred = FMT == MQ_INT_ASYM ? 0xfc007c00u : 0u;
      int count = 0;
#pragma unroll 1
      for (int r = 0; r < NR; ++r) {
        u32x4 v[4];
        const bool valid = load_run(r, v);
        rowred = stats_op<FMT>(rowred, run_red(v, valid));
        count += valid ? 1 : 0;
      }
      if (on0) {
        const SegP sp0 = row_params(rowred);
        if (sp0.slow) {  // (uniform over the row's waves)
          double s = 0.0, ab = 0.0;
#pragma unroll 1
          for (int r = 0; r < count; ++r) {  // (a lane's valid runs are its first ones)
            double s1, ab1;
            stats_slow_segment<FMT>(p, sp0, rowp + (int64_t)r * (4096 * WPR), 4, 16, s1, ab1);
            s += s1;
            ab += ab1;
          }
          sse0 += s;
          abs0 += ab;
          row_finish(row, s);
        }
      }
#pragma unroll 1
      for (int r = 0; r < NR; ++r) {
        u32x4 v[4];
        const bool valid = load_run(r, v);
        const int run = (r * WPR + sub) * 64 + lane;
        const uint32_t r1 = run_red(v, valid);
        const uint32_t r2 = stats_op<FMT>(r1, dpp<DPP_XOR1>(r1));
        const uint32_t r3 = stats_op<FMT>(r2, dpp<DPP_XOR2>(r2));
#pragma unroll 1
        for (int c = 1; c <= 3; ++c) {
          if (!((b.multi_mask >> c) & 1u)) continue;
          const SegP sp = stats_params<FMT>(c == 1 ? r1 : c == 2 ? r2 : r3, p);
          const bool slow = sp.slow && valid;
          if (__builtin_amdgcn_ballot_w64(slow) == 0ull) continue;
          double s = 0.0, ab = 0.0;
          if (slow) stats_slow_segment<FMT>(p, sp, rowp + (int64_t)r * (4096 * WPR), 4, 16, s, ab);
          // the group's sum over its lanes (every lane of a group is slow together)
          double t = s;
          if (c >= 2) t += dpp_f64<DPP_XOR1>(t);
          if (c >= 3) t += dpp_f64<DPP_XOR2>(t);
          float* list = c == 1 ? list1 : c == 2 ? list2 : list3;
          const int T = 1 << (c - 1);  // lanes per group
          if (list && slow && (lane & (T - 1)) == 0)
            list[row * (runs >> (c - 1)) + (run >> (c - 1))] = (float)(t * (1.0 / (32.0 * T)));
          if (c == 1) { sse1 += s; abs1 += ab; }
          else if (c == 2) { sse2 += s; abs2 += ab; }
          else { sse3 += s; abs3 += ab; }
        }
      }
      parity ^= 1;
    }
  }
  // per workgroup and slot: (sum err^2, sum |err|)
  double* const base = b.sweep_partials + 8 * (int64_t)blockIdx.x;
  p.bx = 0;
  p.partials = base;
  block_partials(p, sse0, abs0, lds);
  p.partials = base + 2;
  block_partials(p, sse1, abs1, lds);
  p.partials = base + 4;
  block_partials(p, sse2, abs2, lds);
  p.partials = base + 6;
  block_partials(p, sse3, abs3, lds);
}

#if MQ_PART == 0
__global__ __launch_bounds__(256) void k_finalize_multi(const QuantBatch b) {
  __shared__ double lds[32];
  const MultiItem& it = b.multi[blockIdx.x];
  const int slot = blockIdx.y;
  double* out = it.totals[slot];
  if (!out) return;
  const double* partials = b.sweep_partials + 8 * (int64_t)it.blk0 + 2 * slot;
  double a = 0.0, c = 0.0;
  for (uint32_t i = threadIdx.x; i < it.nblk; i += blockDim.x) { a += partials[8 * (int64_t)i]; c += partials[8 * (int64_t)i + 1]; }
  a = wave_sum(a);
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) { lds[threadIdx.x >> 6] = a; lds[16 + (threadIdx.x >> 6)] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double x = 0.0, y = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { x += lds[i]; y += lds[16 + i]; }
    out[0] = x;
    out[1] = y;
  }
}
#endif
