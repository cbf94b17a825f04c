// Error statistics in numpy's summation order at bandwidth speed.  Included by mq_quant.hip (after its device
// helpers): the format kernels below are compiled in the MQ_NP_UNIT translation units (one per format), the
// finishing kernels and the host side in part 0.
//
// What the analysis scripts print per weight (MSE&P99/quant_error_single_weight.py:46-79, quant_error_multi_weight.py
// :55-58) are float32 sums taken by numpy's pairwise summation:
//   * a contiguous run of float32 values is cut into chunks of 8192 elements whose sums are added in order;
//   * a chunk is split in two -- the left part n / 2 rounded down to a multiple of 8 -- until the parts hold at
//     most 128 elements ("leaves");
//   * a leaf is summed by eight accumulators r[i] += a[8 k + i], k = 0, 1, ..., combined as
//     ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)).
// The unit of work here is the leaf.  A lane holds 8 consecutive elements as four packed pairs; pair m of every
// vector of a leaf feeds accumulators 2 m and 2 m + 1, so lane m of a QUAD takes pair m of all (up to 16) vectors
// of one leaf and runs those two chains in order, with no waiting on other lanes: the chains of a wave are
// independent.  The weights are loaded as always (every load instruction of a wave reads 1 KiB of consecutive
// bytes) and handed to the quads through LDS: 16-byte writes in memory order, 4-byte reads at a stride of four
// words, each leaf padded by four words so that the 64 reads of an instruction hit 64 different banks.
// The squared error of the fast paths is exact in float32 (a float16 error has 11 significant bits, the table
// formats' float32 error at most 12), so r = fma(e, e, r) rounds once, exactly as numpy's r += e * e does.
//
// Outputs per tensor: the per-segment list entries (float32, np.mean's bits), the sum of every 128-element leaf of
// the FLATTENED tensor (from which k_np_chunk_sums / k_np_tensor_finish form np.mean((a32 - b32) ** 2) over the
// whole weight), scale / zero-point when asked for, and the float64 totals of the other statistics kernels.
#pragma once

#define MQ_NP_LEAF_WORDS 68     // LDS words per 128-element leaf: 64 + 4 of padding
#define MQ_NPL_MAX 800          // leaves of a row program: <= 512 row leaves + <= 256 flat leaves + spare
struct NpLeaf {
  uint16_t sv;   // first vector (8 elements) of the leaf inside the row
  uint8_t nv;    // vectors: 8 .. 16
  uint8_t kind;  // bit 0: a leaf of the row's own summation tree (slot = its index); bit 1: a 128-element leaf of
                 // the flattened tensor
};
// The summation of one row (per-channel passes), built on the host: the leaves of the row's own tree first, in
// memory order, then the flat leaves that are not among them.  Every chunk's tree must be complete (all leaves at
// one depth: chunk_leaves[] are powers of two) -- true for every row length that is 128 * 2^k * odd-free ... in
// practice for the multiples of 128 that models use; the host checks and refuses the others.
struct NpRowProg {
  uint32_t nleaves, nrow, nchunks;
  uint32_t regular;  // every chunk holds a multiple of 16 leaves: 16 consecutive leaves are a subtree
  uint32_t chunk_leaves[4];
  NpLeaf leaf[MQ_NPL_MAX];
};

#if MQ_PART >= 1
// ---- one packed word (two elements) ---------------------------------------------------------------------------
struct NpSeg {  // what a lane needs of a segment's parameters to quantize a word of it
  uint32_t szp;                 // scale | zero-point << 16 (float16 bits)
  uint32_t pk_p, pk_s, pk_r;    // integer formats: the packed division's constants (pk_div_params)
  float sf, r;                  // (float)scale, ~1 / scale
};
template <int FMT>
__device__ __forceinline__ NpSeg np_seg(const SegP& sp) {
  NpSeg g;
  g.szp = h_bits(sp.s) | (h_bits(sp.zp) << 16);
  g.pk_p = sp.pk_p; g.pk_s = sp.pk_s; g.pk_r = sp.pk_r;
  g.sf = sp.sf; g.r = sp.r;
  return g;
}
// the parameters lane SRC of every quad holds (quad_perm broadcast); the fast paths' fields only
template <int FMT, int SRC>
__device__ __forceinline__ NpSeg np_seg_from(const NpSeg& own) {
  constexpr int CTRL = SRC * 0x55;
  NpSeg g;
  g.szp = dpp<CTRL>(own.szp);
  if constexpr (!MQ_IS_LUT(FMT)) {
    g.pk_p = dpp<CTRL>(own.pk_p); g.pk_s = dpp<CTRL>(own.pk_s); g.pk_r = dpp<CTRL>(own.pk_r);
    g.sf = 0.f; g.r = 0.f;
  } else {
    g.pk_p = g.pk_s = g.pk_r = 0u;
    g.r = f_from_bits(dpp<CTRL>(f_bits(own.r)));
    g.sf = (float)h_from_bits(g.szp & 0xffffu);
  }
  return g;
}
__device__ __forceinline__ uint32_t bf16x2_to_f16(uint32_t b) {
  h16x2 h;
  h[0] = (half_t)f_from_bits(b << 16);
  h[1] = (half_t)f_from_bits(b & 0xffff0000u);
  return as_u32(h);
}
// Error of the two elements of vw on the fast path (the arithmetic of quant8_stats): packed float16, exact, for the
// integer formats and the tables with subnormals; two float32 values (|v| - |d|, exact) for the tables without.
template <int FMT>
__device__ __forceinline__ void np_word_fast(const QuantParams& p, uint32_t vw, const NpSeg& g, uint32_t& epk, float& e0,
                                             float& e1) {
  const h16x2 s2 = as_h2(dup16(g.szp & 0xffffu));
  const h16x2 magic = as_h2(0x66006600u);  // 1536
  SegP sp;
  sp.pk_p = g.pk_p; sp.pk_s = g.pk_s; sp.pk_r = g.pk_r; sp.sf = g.sf; sp.r = g.r;
  epk = 0u; e0 = 0.f; e1 = 0.f;
  if constexpr (FMT == MQ_INT_SYM) {
    const h16x2 x = div_pk16(vw, sp);
    epk = as_u32(as_h2(vw) - ((x + magic) - magic) * s2);
  } else if constexpr (FMT == MQ_INT_ASYM) {
    const h16x2 x = div_pk16(vw, sp);
    const h16x2 zp2 = as_h2(dup16(g.szp >> 16));
    const h16x2 lo = magic - zp2, hi = lo + as_h2(dup16(h_bits((half_t)p.qmax)));
    const uint32_t t = pk_min_f16(pk_max_f16(as_u32(x + magic), as_u32(lo)), as_u32(hi));
    epk = as_u32(as_h2(vw) - (as_h2(t) - magic) * s2);
  } else if constexpr (FMT == MQ_LUT_FP) {
    const uint32_t dm = as_u32(as_h2(lut_round_mag_pk(div_pk_abs(vw, sp), vw, p.lut)) * s2);
    asm("v_fma_mix_f32 %0, -%1, 1.0, |%2| op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(e0) : "v"(dm), "v"(vw));
    asm("v_fma_mix_f32 %0, -%1, 1.0, |%2| op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(e1) : "v"(dm), "v"(vw));
  } else {
    const h16x2 x = div_pk(vw, sp);
    epk = as_u32(as_h2(vw) - as_h2(lut_round_pk_sub(as_u32(x), p.lut)) * s2);
  }
}
// r += e * e for the two chains of a lane (accumulators 2 m and 2 m + 1 of the leaf): one rounding per step
template <int FMT>
__device__ __forceinline__ void np_chain_fast(float& lo, float& hi, uint32_t epk, float e0, float e1) {
  if constexpr (FMT == MQ_LUT_FP) {
    lo = __builtin_fmaf(e0, e0, lo);
    hi = __builtin_fmaf(e1, e1, hi);
  } else {
    float a, b;
    asm("v_fma_mix_f32 %0, %1, %1, %2 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "=v"(a) : "v"(epk), "v"(lo));
    asm("v_fma_mix_f32 %0, %1, %1, %2 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "=v"(b) : "v"(epk), "v"(hi));
    lo = a;
    hi = b;
  }
}
// sum |e| of the word, added to ab
template <int FMT>
__device__ __forceinline__ void np_abs_fast(float& ab, uint32_t epk, float e0, float e1) {
  if constexpr (FMT == MQ_LUT_FP) ab += __builtin_fabsf(e0) + __builtin_fabsf(e1);
  else ab = __builtin_amdgcn_fdot2(as_h2(epk & 0x7fff7fffu), as_h2(0x3c003c00u), ab, false);
}
// The contract element by element (quant8_slow for one word): float32 errors of any float16 input.
template <int FMT>
__device__ __forceinline__ void np_word_slow(const QuantParams& p, uint32_t vw, const NpSeg& g, float& e0, float& e1) {
  SegP sp;
  sp.s = h_from_bits(g.szp & 0xffffu);
  sp.zp = h_from_bits(g.szp >> 16);
  sp.sf = (float)sp.s;
  sp.r = __builtin_amdgcn_rcpf(sp.sf);
  const half_t qmax_h = (half_t)p.qmax;
  float e[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const uint32_t vb = i ? (vw >> 16) : (vw & 0xffffu);
    const float vf = (float)h_from_bits(vb);
    const half_t x = div_h_full(vf, sp);
    half_t d;
    if constexpr (FMT == MQ_INT_SYM) {
      half_t q = __builtin_rintf16(x);
      q = q < -qmax_h ? -qmax_h : q;
      q = q > qmax_h ? qmax_h : q;
      d = q * sp.s;
    } else if constexpr (FMT == MQ_INT_ASYM) {
      half_t q = __builtin_rintf16(x) + sp.zp;
      q = q < (half_t)0 ? (half_t)0 : q;
      q = q > qmax_h ? qmax_h : q;
      d = (q - sp.zp) * sp.s;
    } else {
      const uint32_t code = p.lut.sub ? h_bits((half_t)lut_round_sub((float)x, p.lut))
                                      : lut_round_bits(h_bits(x), 0x8000u, 0x7c00u, p.lut);
      d = h_from_bits(code) * sp.s;
    }
    e[i] = vf - (float)d;
  }
  e0 = e[0];
  e1 = e[1];
}
// ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)) from the four lanes' r[2 m] + r[2 m + 1]; every lane of the quad
// gets the sum
__device__ __forceinline__ float np_quad_tree(float s) {
  s += f_from_bits(dpp<DPP_XOR1>(f_bits(s)));
  s += f_from_bits(dpp<DPP_XOR2>(f_bits(s)));
  return s;
}
// the reduction word (absmax bits, or min | max << 16) over NW packed words of one segment
template <int FMT, int NW>
__device__ __forceinline__ uint32_t np_words_reduce(const uint32_t* w) {
  uint32_t mx = w[0], mn = w[0];
#pragma unroll
  for (int i = 1; i + 1 < NW; i += 2) {
    mx = pk_maximum3_f16(mx, w[i], w[i + 1]);
    mn = pk_minimum3_f16(mn, w[i], w[i + 1]);
  }
  if constexpr ((NW & 1) == 0) {  // NW - 1 words follow the first: an odd count leaves one
    mx = pk_maximum3_f16(mx, w[NW - 1], w[NW - 1]);
    mn = pk_minimum3_f16(mn, w[NW - 1], w[NW - 1]);
  }
  if constexpr (FMT == MQ_INT_ASYM) {
    const uint32_t a = (mn & 0xffffu) | (mx << 16), c = (mn >> 16) | (mx & 0xffff0000u);
    return (pk_minimum3_f16(a, c, c) & 0xffffu) | (pk_maximum3_f16(a, c, c) & 0xffff0000u);
  } else {
    return fold_max_u16(pk_max_u16(mx & 0x7fff7fffu, mn & 0x7fff7fffu));  // integer max: a NaN wins
  }
}
__device__ __forceinline__ void np_lds_order() { __builtin_amdgcn_wave_barrier(); }
// float32 values of lanes 0 .. 31 added as a perfect tree, neighbours first; valid in lane 0
__device__ __forceinline__ float np_wave_tree32(float v) {
  v += f_from_bits(dpp<DPP_XOR1>(f_bits(v)));
  v += f_from_bits(dpp<DPP_XOR2>(f_bits(v)));
  v += f_from_bits(dpp<DPP_HALF_MIRROR>(f_bits(v)));
  v += f_from_bits(dpp<DPP_MIRROR>(f_bits(v)));
  v += __shfl_xor(v, 16);
  return v;
}

// ---- groups of 32 * T elements, T = 1, 2, 4 -------------------------------------------------------------------
// A wave step is 4 KiB = 16 leaves of the flattened tensor (the host admits tensors of a multiple of 8192 elements,
// so there are no partial steps); quad q takes leaf q, which holds 4 / T groups.  The group's entry is its own
// leaf-style sum (4 T vectors), the first group's chain is the start of the flat leaf's.
template <int FMT, bool BF16, int T>
__global__ __launch_bounds__(256) void k_np_group(const QuantBatch b) {
  constexpr int NG = 4 / T, WPG = 4 * T;  // groups per leaf, words of a lane per group
  const QuantParams p = batch_params(b);
  __shared__ double lds[32];
  __shared__ __attribute__((aligned(16))) uint32_t xch[4][16 * MQ_NP_LEAF_WORDS];
  const int lane = threadIdx.x & 63, q = lane >> 2, m = lane & 3;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t gw = (int64_t)p.bx * 4 + wv, nwaves = (int64_t)p.gx * 4;
  const int64_t steps = p.n >> 11;
  uint32_t* reg = xch[wv];
  const __amdgpu_buffer_rsrc_t rs = stats_rsrc(p);
  const uint32_t loff = (uint32_t)lane * 16u;
  double acc_sse = 0.0, acc_abs = 0.0;
  bool seen_slow = false;
  float* ent_out = reinterpret_cast<float*>(p.seg_sse);
  const float inv_len = (float)p.seg_mse_scale;  // 1 / L, a power of two: np.mean's division is exact

  auto issue = [&](int64_t t, u32x4* dst) __attribute__((always_inline)) {
    const uint32_t base = (uint32_t)(t * 4096) + loff;  // (past the tensor: zeros, no access -- prefetches only)
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j] = stats_bload<true>(rs, base + j * 1024);
  };
  // this lane's pair of the 16 vectors of its quad's leaf, as float16
  auto exchange = [&](const u32x4* buf, uint32_t* W) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j)  // vector 64 j + lane of the step: leaf 4 j + lane / 16, position lane % 16
      *reinterpret_cast<u32x4*>(&reg[(4 * j + (lane >> 4)) * MQ_NP_LEAF_WORDS + (lane & 15) * 4]) = buf[j];
    np_lds_order();
#pragma unroll
    for (int k = 0; k < 16; ++k) W[k] = reg[q * MQ_NP_LEAF_WORDS + 4 * k + m];
    np_lds_order();
    if constexpr (BF16) {
#pragma unroll
      for (int k = 0; k < 16; ++k) W[k] = bf16x2_to_f16(W[k]);
    }
  };
  // the parameters of the group this lane computes (group m / T of the leaf); false: some group of the wave's step
  // is off the fast path
  auto params = [&](const uint32_t* W, SegP& sp, uint32_t& zpb) __attribute__((always_inline)) {
    uint32_t red[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      uint32_t r = np_words_reduce<FMT, WPG>(W + g * WPG);
      r = stats_op<FMT>(r, dpp<DPP_XOR1>(r));
      red[g] = stats_op<FMT>(r, dpp<DPP_XOR2>(r));
    }
    uint32_t mine = red[0];
#pragma unroll
    for (int g = 1; g < NG; ++g) mine = (m / T) == g ? red[g] : mine;
    sp = stats_params<FMT>(mine, p);
    zpb = h_bits(sp.zp);
    if constexpr (FMT == MQ_INT_ASYM && T == 1) {
      // the sign of a zero zero-point of a 32-element group (see zp_bits_first_zero): lane m holds word m of every
      // vector, elements 8 (k % 4) + 2 m and + 1 of group k / 4
      const bool need = zpb == 0u && (mine & 0xffffu) == 0x8000u;
      if (p.zp && __builtin_amdgcn_ballot_w64(need) != 0ull) {
        bool plus_first = false;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          uint32_t fp = 0xffffu, fm = 0xffffu;
#pragma unroll
          for (int k = WPG - 1; k >= 0; --k) {
            const uint32_t w = W[g * WPG + k], e0 = (uint32_t)(8 * k + 2 * m);
            fp = (w >> 16) == 0u ? e0 + 1u : fp;
            fm = (w >> 16) == 0x8000u ? e0 + 1u : fm;
            fp = (w & 0xffffu) == 0u ? e0 : fp;
            fm = (w & 0xffffu) == 0x8000u ? e0 : fm;
          }
          fp = op_min_u32(fp, dpp<DPP_XOR1>(fp)); fp = op_min_u32(fp, dpp<DPP_XOR2>(fp));
          fm = op_min_u32(fm, dpp<DPP_XOR1>(fm)); fm = op_min_u32(fm, dpp<DPP_XOR2>(fm));
          plus_first = m == g ? fp < fm : plus_first;
        }
        zpb = (need && plus_first) ? 0x8000u : zpb;
      }
    }
    return __builtin_amdgcn_ballot_w64(sp.slow) == 0ull;
  };
  // entries, leaf sum, parameters -> memory
  auto store = [&](int64_t t, const SegP& sp, uint32_t zpb, const float* ent, float leaf, float ab) __attribute__((always_inline)) {
    const int64_t lf = t * 16 + q;
    float mine = ent[0];
#pragma unroll
    for (int g = 1; g < NG; ++g) mine = m == g ? ent[g] : mine;
    if (m < NG) {
      if (ent_out) ent_out[lf * NG + m] = mine * inv_len;
    }
    if ((m % T) == 0) {  // the lane that computed group m / T's parameters
      if (p.scale) p.scale[lf * NG + m / T] = (uint16_t)h_bits(sp.s);
      if (FMT == MQ_INT_ASYM && p.zp) p.zp[lf * NG + m / T] = (uint16_t)zpb;
    }
    if (m == 0) {
      if (p.leaf) p.leaf[lf] = leaf;
      acc_sse += (double)leaf;  // (the tensor's float64 sum from the leaves' float32 sums: 1e-7, inside the bar)
    }
    acc_abs += (double)ab;
  };
  auto step = [&](int64_t t, const u32x4* buf) __attribute__((always_inline)) {
    uint32_t W[16];
    exchange(buf, W);
    SegP sp;
    uint32_t zpb;
    if (!params(W, sp, zpb)) { seen_slow = true; return; }
    const NpSeg own = np_seg<FMT>(sp);
    NpSeg G[NG];
    if constexpr (NG == 1) { G[0] = own; }
    else if constexpr (NG == 2) { G[0] = np_seg_from<FMT, 0>(own); G[1] = np_seg_from<FMT, 2>(own); }
    else { G[0] = np_seg_from<FMT, 0>(own); G[1] = np_seg_from<FMT, 1>(own); G[2] = np_seg_from<FMT, 2>(own);
           G[3] = np_seg_from<FMT, 3>(own); }
    float flo = 0.f, fhi = 0.f, rlo = 0.f, rhi = 0.f, ab = 0.f;
    float ent[NG];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int g = k / WPG;
      uint32_t epk;
      float e0, e1;
      np_word_fast<FMT>(p, W[k], G[g], epk, e0, e1);
      np_chain_fast<FMT>(flo, fhi, epk, e0, e1);
      if (g > 0) np_chain_fast<FMT>(rlo, rhi, epk, e0, e1);
      np_abs_fast<FMT>(ab, epk, e0, e1);
      if ((k + 1) % WPG == 0) {
        ent[g] = np_quad_tree(g == 0 ? flo + fhi : rlo + rhi);
        rlo = 0.f;
        rhi = 0.f;
      }
      if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    const float leaf = NG == 1 ? ent[0] : np_quad_tree(flo + fhi);
    store(t, sp, zpb, ent, leaf, ab);
  };
  {
    u32x4 bufa[4], bufb[4];
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
    // steps with a segment off the fast path: every word by the per-element routine (equal to the fast path where
    // that applies), squares rounded to float32 as numpy's diff * diff, then added
#pragma unroll 1
    for (int64_t t = gw; t < steps; t += nwaves) {
      u32x4 buf[4];
      issue(t, buf);
      uint32_t W[16];
      exchange(buf, W);
      SegP sp;
      uint32_t zpb;
      if (params(W, sp, zpb)) continue;
      const NpSeg own = np_seg<FMT>(sp);
      float flo = 0.f, fhi = 0.f, ab = 0.f;
      float ent[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        NpSeg sg;  // group g's scale and zero-point, from the lane that computed them
        sg.szp = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane & ~3) + g * T) * 4, (int)own.szp);
        float rlo = 0.f, rhi = 0.f;
#pragma unroll 1
        for (int k = g * WPG; k < (g + 1) * WPG; ++k) {
          uint32_t vw = W[0];  // W[k] without indexing the register array by a run-time k
#pragma unroll
          for (int kk = 1; kk < 16; ++kk) vw = kk == k ? W[kk] : vw;
          float e0, e1;
          np_word_slow<FMT>(p, vw, sg, e0, e1);
          const float s0 = e0 * e0, s1 = e1 * e1;
          flo += s0; fhi += s1;
          rlo += s0; rhi += s1;
          ab += __builtin_fabsf(e0) + __builtin_fabsf(e1);
        }
        ent[g] = np_quad_tree(rlo + rhi);
      }
      const float leaf = NG == 1 ? ent[0] : np_quad_tree(flo + fhi);
      store(t, sp, zpb, ent, leaf, ab);
    }
  }
  block_partials(p, acc_sse, acc_abs, lds);
}

// ---- groups of 32, 64 AND 128 elements from one read --------------------------------------------------------------
// The passes of a sweep that differ only in the group size read the same weights, exchange them through LDS the same
// way and differ from the words W[] on.  This kernel does the load and the exchange once and then runs k_np_group's
// step for every group size the launch asks for (b.np_tabs[c]: the pass's own launch table for groups of 32 << c, null
// when the size is not wanted; the tables list the same tensors with the same workgroup ranges -- a step is 4 KiB
// whatever the group size).  Every list entry, leaf sum, scale, zero-point and total is what k_np_group<FMT, BF16,
// 1 << c> writes.  Partial sums of size c: b.sweep_partials + 2 * (c * gridDim.x + blockIdx.x).
template <int FMT, bool BF16>
__global__ __launch_bounds__(256) void k_np_group3(const QuantBatch b) {
  QuantParams p = b.common;
  __shared__ double lds[32];
  __shared__ __attribute__((aligned(16))) uint32_t xch[4][16 * MQ_NP_LEAF_WORDS];
  const int first_c = b.np_tabs[0] ? 0 : b.np_tabs[1] ? 1 : 2;
  int ti;
  {
    const SweepItem* tab = b.np_tabs[first_c];
    int lo = 0, hi = b.sweep_count - 1;
    while (lo < hi) {  // the last entry whose first workgroup is not past this one (uniform: scalar loads)
      const int mid = (lo + hi + 1) >> 1;
      if (tab[mid].blk0 <= blockIdx.x) lo = mid;
      else hi = mid - 1;
    }
    ti = lo;
    const SweepItem& it = tab[lo];
    p.w = it.w;
    p.n = it.nseg * (32 << first_c);
    p.bx = blockIdx.x - it.blk0;
    p.gx = it.nblk;
  }
  const int lane = threadIdx.x & 63, q = lane >> 2, m = lane & 3;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t gw = (int64_t)p.bx * 4 + wv, nwaves = (int64_t)p.gx * 4;
  const int64_t steps = p.n >> 11;
  uint32_t* reg = xch[wv];
  const __amdgpu_buffer_rsrc_t rs = stats_rsrc(p);
  const uint32_t loff = (uint32_t)lane * 16u;
  double acc_sse[3] = {0.0, 0.0, 0.0}, acc_abs[3] = {0.0, 0.0, 0.0};
  uint32_t seen_slow = 0u;  // bit c: some step of this wave holds a group of size c off the fast path

  auto issue = [&](int64_t t, u32x4* dst) __attribute__((always_inline)) {
    const uint32_t base = (uint32_t)(t * 4096) + loff;  // (past the tensor: zeros, no access -- prefetches only)
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j] = stats_bload<true>(rs, base + j * 1024);
  };
  auto exchange = [&](const u32x4* buf, uint32_t* W) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<u32x4*>(&reg[(4 * j + (lane >> 4)) * MQ_NP_LEAF_WORDS + (lane & 15) * 4]) = buf[j];
    np_lds_order();
#pragma unroll
    for (int k = 0; k < 16; ++k) W[k] = reg[q * MQ_NP_LEAF_WORDS + 4 * k + m];
    np_lds_order();
    if constexpr (BF16) {
#pragma unroll
      for (int k = 0; k < 16; ++k) W[k] = bf16x2_to_f16(W[k]);
    }
  };
  // One group size over the words of a step.  SLOW = false: the fast path, nothing is done (and false returned) when a
  // group of the wave's step is off it; SLOW = true: only such steps, element by element.
  auto run = [&](auto tc, auto slowc, int64_t t, const uint32_t* W) __attribute__((always_inline)) -> bool {
    constexpr int T = decltype(tc)::value, C = T == 1 ? 0 : T == 2 ? 1 : 2;
    constexpr bool SLOW = decltype(slowc)::value;
    constexpr int NG = 4 / T, WPG = 4 * T;
    const SweepItem& it = b.np_tabs[C][ti];
    float* const ent_out = reinterpret_cast<float*>(it.seg_sse);
    const float inv_len = 1.0f / (float)(32 * T);
    // the parameters of the group this lane computes (group m / T of the leaf)
    uint32_t red[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      uint32_t r = np_words_reduce<FMT, WPG>(W + g * WPG);
      r = stats_op<FMT>(r, dpp<DPP_XOR1>(r));
      red[g] = stats_op<FMT>(r, dpp<DPP_XOR2>(r));
    }
    uint32_t mine = red[0];
#pragma unroll
    for (int g = 1; g < NG; ++g) mine = (m / T) == g ? red[g] : mine;
    const SegP sp = stats_params<FMT>(mine, p);
    uint32_t zpb = h_bits(sp.zp);
    const bool all_fast = __builtin_amdgcn_ballot_w64(sp.slow) == 0ull;
    if (SLOW == all_fast) return all_fast;
    if constexpr (FMT == MQ_INT_ASYM && T == 1) {
      // the sign of a zero zero-point of a 32-element group (see zp_bits_first_zero): lane m holds word m of every
      // vector, elements 8 (k % 4) + 2 m and + 1 of group k / 4
      const bool need = zpb == 0u && (mine & 0xffffu) == 0x8000u;
      if (it.zp && __builtin_amdgcn_ballot_w64(need) != 0ull) {
        bool plus_first = false;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          uint32_t fp = 0xffffu, fm = 0xffffu;
#pragma unroll
          for (int k = WPG - 1; k >= 0; --k) {
            const uint32_t w = W[g * WPG + k], e0 = (uint32_t)(8 * k + 2 * m);
            fp = (w >> 16) == 0u ? e0 + 1u : fp;
            fm = (w >> 16) == 0x8000u ? e0 + 1u : fm;
            fp = (w & 0xffffu) == 0u ? e0 : fp;
            fm = (w & 0xffffu) == 0x8000u ? e0 : fm;
          }
          fp = op_min_u32(fp, dpp<DPP_XOR1>(fp)); fp = op_min_u32(fp, dpp<DPP_XOR2>(fp));
          fm = op_min_u32(fm, dpp<DPP_XOR1>(fm)); fm = op_min_u32(fm, dpp<DPP_XOR2>(fm));
          plus_first = m == g ? fp < fm : plus_first;
        }
        zpb = (need && plus_first) ? 0x8000u : zpb;
      }
    }
    const NpSeg own = np_seg<FMT>(sp);
    float flo = 0.f, fhi = 0.f, ab = 0.f;
    float ent[NG];
    if constexpr (!SLOW) {
      NpSeg G[NG];
      if constexpr (NG == 1) { G[0] = own; }
      else if constexpr (NG == 2) { G[0] = np_seg_from<FMT, 0>(own); G[1] = np_seg_from<FMT, 2>(own); }
      else { G[0] = np_seg_from<FMT, 0>(own); G[1] = np_seg_from<FMT, 1>(own); G[2] = np_seg_from<FMT, 2>(own);
             G[3] = np_seg_from<FMT, 3>(own); }
      float rlo = 0.f, rhi = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int g = k / WPG;
        uint32_t epk;
        float e0, e1;
        np_word_fast<FMT>(p, W[k], G[g], epk, e0, e1);
        np_chain_fast<FMT>(flo, fhi, epk, e0, e1);
        if (g > 0) np_chain_fast<FMT>(rlo, rhi, epk, e0, e1);
        np_abs_fast<FMT>(ab, epk, e0, e1);
        if ((k + 1) % WPG == 0) {
          ent[g] = np_quad_tree(g == 0 ? flo + fhi : rlo + rhi);
          rlo = 0.f;
          rhi = 0.f;
        }
        if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // every word by the per-element routine (equal to the fast path where that applies), squares rounded to float32
      // as numpy's diff * diff, then added
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        NpSeg sg;  // group g's scale and zero-point, from the lane that computed them
        sg.szp = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane & ~3) + g * T) * 4, (int)own.szp);
        float rlo = 0.f, rhi = 0.f;
#pragma unroll 1
        for (int k = g * WPG; k < (g + 1) * WPG; ++k) {
          uint32_t vw = W[0];  // W[k] without indexing the register array by a run-time k
#pragma unroll
          for (int kk = 1; kk < 16; ++kk) vw = kk == k ? W[kk] : vw;
          float e0, e1;
          np_word_slow<FMT>(p, vw, sg, e0, e1);
          const float s0 = e0 * e0, s1 = e1 * e1;
          flo += s0; fhi += s1;
          rlo += s0; rhi += s1;
          ab += __builtin_fabsf(e0) + __builtin_fabsf(e1);
        }
        ent[g] = np_quad_tree(rlo + rhi);
      }
    }
    const float leaf = NG == 1 ? ent[0] : np_quad_tree(flo + fhi);
    // entries, leaf sum, parameters -> memory
    const int64_t lf = t * 16 + q;
    float mine_ent = ent[0];
#pragma unroll
    for (int g = 1; g < NG; ++g) mine_ent = m == g ? ent[g] : mine_ent;
    if (m < NG) {
      if (ent_out) ent_out[lf * NG + m] = mine_ent * inv_len;
    }
    if ((m % T) == 0) {  // the lane that computed group m / T's parameters
      if (it.scale) it.scale[lf * NG + m / T] = (uint16_t)h_bits(sp.s);
      if (FMT == MQ_INT_ASYM && it.zp) it.zp[lf * NG + m / T] = (uint16_t)zpb;
    }
    if (m == 0) {
      if (it.leaf) it.leaf[lf] = leaf;
      acc_sse[C] += (double)leaf;
    }
    acc_abs[C] += (double)ab;
    return all_fast;
  };
  using std::integral_constant;
  auto step = [&](int64_t t, const u32x4* buf) __attribute__((always_inline)) {
    uint32_t W[16];
    exchange(buf, W);
    if (b.np_tabs[0] && !run(integral_constant<int, 1>{}, integral_constant<bool, false>{}, t, W)) seen_slow |= 1u;
    if (b.np_tabs[1] && !run(integral_constant<int, 2>{}, integral_constant<bool, false>{}, t, W)) seen_slow |= 2u;
    if (b.np_tabs[2] && !run(integral_constant<int, 4>{}, integral_constant<bool, false>{}, t, W)) seen_slow |= 4u;
  };
  {
    u32x4 bufa[4], bufb[4];
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
  seen_slow = (uint32_t)__builtin_amdgcn_readfirstlane((int)seen_slow);  // (every flag came from a ballot: uniform)
  if (__builtin_expect(seen_slow != 0u, 0)) {
#pragma unroll 1
    for (int64_t t = gw; t < steps; t += nwaves) {
      u32x4 buf[4];
      issue(t, buf);
      uint32_t W[16];
      exchange(buf, W);
      if (seen_slow & 1u) run(integral_constant<int, 1>{}, integral_constant<bool, true>{}, t, W);
      if (seen_slow & 2u) run(integral_constant<int, 2>{}, integral_constant<bool, true>{}, t, W);
      if (seen_slow & 4u) run(integral_constant<int, 4>{}, integral_constant<bool, true>{}, t, W);
    }
  }
  p.bx = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (!b.np_tabs[c]) continue;
    p.partials = b.sweep_partials + 2 * ((int64_t)c * gridDim.x + blockIdx.x);
    block_partials(p, acc_sse[c], acc_abs[c], lds);
  }
}

// ---- rows of 2048 * S elements, S = 1, 2, 4 ---------------------------------------------------------------------
// A wave per row: the row is S wave steps of the group kernel (16 leaves each) and at most one chunk of numpy's
// summation, a perfect tree -- everything after the one LDS exchange stays in registers: the words of all S steps
// (the row's parameters need them all), the leaf sums, the 16 leaves of a step added by lane shuffles, the S step
// sums by a small tree.  These rows' leaves ARE the leaves of the flattened tensor.
template <int FMT, bool BF16, int S>
__global__ __launch_bounds__(256) void k_np_rowx(const QuantBatch b) {
  const QuantParams p = batch_params(b);
  __shared__ double lds[32];
  __shared__ __attribute__((aligned(16))) uint32_t xch[4][16 * MQ_NP_LEAF_WORDS];
  const int lane = threadIdx.x & 63, q = lane >> 2, m = lane & 3;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t first = (int64_t)p.bx * 4 + wv, stride = (int64_t)p.gx * 4;
  uint32_t* reg = xch[wv];
  const __amdgpu_buffer_rsrc_t rs = stats_rsrc(p);
  const uint32_t loff = (uint32_t)lane * 16u;
  double acc_sse = 0.0, acc_abs = 0.0;
  float* ent_out = reinterpret_cast<float*>(p.seg_sse);
  const float row_len = (float)p.L;

  auto issue = [&](int64_t row, u32x4* dst) __attribute__((always_inline)) {
    const uint32_t base = row < p.nseg ? (uint32_t)(row * (4096 * S)) + loff : MQ_STATS_OOB;
#pragma unroll
    for (int j = 0; j < 4 * S; ++j) dst[j] = stats_bload<true>(rs, base + j * 1024);
  };
  auto exchange = [&](const u32x4* buf, uint32_t* W) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<u32x4*>(&reg[(4 * j + (lane >> 4)) * MQ_NP_LEAF_WORDS + (lane & 15) * 4]) = buf[j];
    np_lds_order();
#pragma unroll
    for (int k = 0; k < 16; ++k) W[k] = reg[q * MQ_NP_LEAF_WORDS + 4 * k + m];
    np_lds_order();
    if constexpr (BF16) {
#pragma unroll
      for (int k = 0; k < 16; ++k) W[k] = bf16x2_to_f16(W[k]);
    }
  };
  u32x4 buf[4 * S];
  issue(first, buf);
  for (int64_t row = first; row < p.nseg; row += stride) {
    uint32_t W[S][16];
#pragma unroll
    for (int s = 0; s < S; ++s) exchange(buf + 4 * s, W[s]);
    issue(row + stride, buf);  // the next row's loads fly while this one is summed (past the last row: nothing)
    uint32_t red = np_words_reduce<FMT, 16>(W[0]);
#pragma unroll
    for (int s = 1; s < S; ++s) red = stats_op<FMT>(red, np_words_reduce<FMT, 16>(W[s]));
    const SegP sp = stats_params<FMT>(team_reduce_u32<64>(red, stats_op<FMT>), p);
    const NpSeg g = np_seg<FMT>(sp);
    float B[S];
    float ab = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      float lo = 0.f, hi = 0.f;
      if (__builtin_expect(!sp.slow, 1)) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          uint32_t epk;
          float e0, e1;
          np_word_fast<FMT>(p, W[s][k], g, epk, e0, e1);
          np_chain_fast<FMT>(lo, hi, epk, e0, e1);
          np_abs_fast<FMT>(ab, epk, e0, e1);
          if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
      } else {  // the whole row by the per-element routine
#pragma unroll 1
        for (int k = 0; k < 16; ++k) {
          uint32_t vw = W[s][0];
#pragma unroll
          for (int kk = 1; kk < 16; ++kk) vw = kk == k ? W[s][kk] : vw;
          float e0, e1;
          np_word_slow<FMT>(p, vw, g, e0, e1);
          lo += e0 * e0;
          hi += e1 * e1;
          ab += __builtin_fabsf(e0) + __builtin_fabsf(e1);
        }
      }
      const float leaf = np_quad_tree(lo + hi);
      if (m == 0) {
        if (p.leaf) p.leaf[(row * S + s) * 16 + q] = leaf;
        acc_sse += (double)leaf;
      }
      float v = leaf;  // the 16 leaves of the step (a quad's four lanes hold the same value)
      v += f_from_bits(dpp<DPP_HALF_MIRROR>(f_bits(v)));
      v += f_from_bits(dpp<DPP_MIRROR>(f_bits(v)));
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      B[s] = v;
    }
    acc_abs += (double)ab;
    float total = B[0];
    if constexpr (S == 2) total = B[0] + B[1];
    if constexpr (S == 4) total = (B[0] + B[1]) + (B[2] + B[3]);
    if (lane == 0) {
      if (ent_out) ent_out[row] = total / row_len;  // np.mean: the float32 sum over the count, rounded once
      if (p.scale) p.scale[row] = (uint16_t)h_bits(sp.s);
      if (FMT == MQ_INT_ASYM && p.zp) p.zp[row] = (uint16_t)h_bits(sp.zp);
    }
  }
  block_partials(p, acc_sse, acc_abs, lds);
}

// ---- rows ------------------------------------------------------------------------------------------------------
// WPR waves share a row (1: a wave per row, rows of up to 512 * NV elements; 4: the workgroup per row, up to 2048 *
// NV).  The row is loaded as in k_stats_row, its parameters come from the lanes' own vectors, then the float16 row
// goes to LDS (the registers are free for the next row's loads at once) and the quads walk the leaves of the row
// program: first the leaves of the row's own tree, then the 128-element leaves of the flattened tensor that are not
// among them.  Those read the ERRORS the row leaves left in place of the weights (packed float16, exact), so no
// element is quantized twice; the tables without subnormals, whose error needs float32, quantize again.
// The row's tree: 16 consecutive leaves are a subtree whenever every chunk holds a multiple of 16 leaves
// (prog.regular) -- the 16 quads of a wave add them with lane shuffles, the few block sums meet in LDS; otherwise the
// leaf sums are combined level by level in LDS.  Then the chunks' sums in order, divided as np.mean divides.
template <int FMT, bool BF16, int NV, int WPR>
__global__ __launch_bounds__(256) void k_np_row(const QuantBatch b) {
  const QuantParams p = batch_params(b);
  extern __shared__ __attribute__((aligned(16))) uint32_t np_dyn[];
  __shared__ double lds[32];
  __shared__ uint32_t xr[2][4];
  __shared__ NpLeaf leaves[MQ_NPL_MAX];
  const NpRowProg& prog = *b.rowprog;
  const int nleaves = (int)prog.nleaves, nrow = (int)prog.nrow;
  const bool regular = prog.regular != 0;
  for (int i = threadIdx.x; i < nleaves; i += 256) leaves[i] = prog.leaf[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, m = lane & 3;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t first = WPR == 1 ? (int64_t)p.bx * 4 + wv : (int64_t)p.bx;
  const int64_t stride = WPR == 1 ? (int64_t)p.gx * 4 : (int64_t)p.gx;
  const int sub = WPR == 1 ? 0 : wv;
  const int nleaf128 = p.vecs_per_seg >> 4;                       // 128-element runs of a row
  const int region_words = nleaf128 * MQ_NP_LEAF_WORDS;
  const int nrow_pad = (nrow + 3) & ~3;
  uint32_t* region = np_dyn + (WPR == 1 ? wv * region_words : 0);
  float* lsum = reinterpret_cast<float*>(np_dyn + (WPR == 1 ? 4 : 1) * region_words) + (WPR == 1 ? wv * nrow_pad : 0);
  const int64_t row_bytes = p.L * 2;
  const uint32_t loff = (uint32_t)(sub * 1024 + lane * 16);
  const bool whole = p.vecs_per_seg == 64 * NV * WPR;
  const int quad = (WPR == 1 ? 0 : wv * 16) + (lane >> 2), nquads = 16 * WPR;
  double acc_sse = 0.0, acc_abs = 0.0;
  int parity = 0;
  const __amdgpu_buffer_rsrc_t rs = stats_rsrc(p);
  float* ent_out = reinterpret_cast<float*>(p.seg_sse);
  const float row_len = (float)p.L;
  constexpr bool KEEP_ERR = FMT != MQ_LUT_FP;  // the row leaves leave packed float16 errors for the flat leaves

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
  auto group_sync = [&]() __attribute__((always_inline)) {
    if constexpr (WPR > 1) __syncthreads();
    else np_lds_order();
  };
  auto row_params = [&](const u32x4* raw) __attribute__((always_inline)) {
    uint32_t red = stats_lane_reduce<FMT, NV>(raw);
    if constexpr (FMT == MQ_INT_ASYM) {
      if (!whole) {  // the zeros past the row's end must not enter min / max
        red = 0xfc007c00u;
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
  // One leaf: this lane's two chains over its (up to 16) vectors -> the leaf's sum in every lane of the quad.
  // from_err: the words are errors a row leaf left; keep: leave the errors in place of the words.
  auto leaf_sum = [&](const NpLeaf L, const NpSeg& g, bool slow, bool from_err, bool keep, float& ab) __attribute__((always_inline)) {
    float lo = 0.f, hi = 0.f;
    if (__builtin_expect(!slow, 1)) {
      // four vectors at a time; a group past the leaf's end is skipped by the whole wave when no quad needs it, and
      // vectors past the end inside a group are zeros: a zero weight has a zero error on the fast path (asymmetric
      // rows on one side of zero are not on it) and adds nothing
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        if (kb >= 2 && __builtin_amdgcn_ballot_w64(4 * kb < (int)L.nv) == 0ull) continue;
        uint32_t W[4];
        int a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int v = (int)L.sv + 4 * kb + k;
          a[k] = (v >> 4) * MQ_NP_LEAF_WORDS + (v & 15) * 4 + m;
          W[k] = 4 * kb + k < (int)L.nv ? region[a[k]] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          uint32_t epk = W[k];
          float e0 = 0.f, e1 = 0.f;
          if (!(KEEP_ERR && from_err)) np_word_fast<FMT>(p, W[k], g, epk, e0, e1);
          np_chain_fast<FMT>(lo, hi, epk, e0, e1);
          np_abs_fast<FMT>(ab, epk, e0, e1);
          if constexpr (KEEP_ERR) {
            if (keep && 4 * kb + k < (int)L.nv) region[a[k]] = epk;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll 1
      for (int k = 0; k < (int)L.nv; ++k) {
        const int v = (int)L.sv + k;
        float e0, e1;
        np_word_slow<FMT>(p, region[(v >> 4) * MQ_NP_LEAF_WORDS + (v & 15) * 4 + m], g, e0, e1);
        lo += e0 * e0;
        hi += e1 * e1;
        ab += __builtin_fabsf(e0) + __builtin_fabsf(e1);
      }
    }
    return np_quad_tree(lo + hi);
  };
  u32x4 buf[NV];
  issue(first, buf);
  for (int64_t row = first; row < p.nseg; row += stride) {
    SegP sp;
    {
      u32x4 raw[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) raw[j] = stats_cvt<BF16>(buf[j]);
      sp = row_params(raw);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int v = (j * WPR + sub) * 64 + lane;
        if (whole || v < p.vecs_per_seg)
          *reinterpret_cast<u32x4*>(&region[(v >> 4) * MQ_NP_LEAF_WORDS + (v & 15) * 4]) = raw[j];
      }
    }
    issue(row + stride, buf);  // the next row's loads fly while this one is summed (past the last row: nothing)
    group_sync();
    const NpSeg g = np_seg<FMT>(sp);
    const bool slow = sp.slow;  // (uniform over the row's waves)
    float ab_row = 0.f;
    // the leaves of the row's own tree
    for (int base = 0; base < nrow; base += nquads) {
      const int lf = base + quad;
      float leaf = 0.f;
      if (lf < nrow) {
        const NpLeaf L = leaves[lf];
        float ab = 0.f;
        leaf = leaf_sum(L, g, slow, false, L.kind == 1 && !slow, ab);
        if (L.kind & 2) {
          ab_row += ab;
          if (m == 0) {
            if (p.leaf) p.leaf[row * nleaf128 + (L.sv >> 4)] = leaf;
            acc_sse += (double)leaf;
          }
        }
      }
      if (regular) {
        // (whole blocks of 16 leaves: a wave either holds one or none)
        if (base + (WPR == 1 ? 0 : wv * 16) < nrow) {
          float v = leaf;
          v += f_from_bits(dpp<DPP_HALF_MIRROR>(f_bits(v)));  // the quads' values are the same in their four lanes
          v += f_from_bits(dpp<DPP_MIRROR>(f_bits(v)));
          v += __shfl_xor(v, 16);
          v += __shfl_xor(v, 32);
          if (lane == 0) lsum[(base >> 4) + (WPR == 1 ? 0 : wv)] = v;
        }
      } else if (lf < nrow && m == 0) {
        lsum[lf] = leaf;
      }
    }
    group_sync();
    // the flat leaves that are not row leaves
    for (int lf = nrow + quad; lf < nleaves; lf += nquads) {
      const NpLeaf L = leaves[lf];
      float ab = 0.f;
      const float leaf = leaf_sum(L, g, slow, !slow, false, ab);
      ab_row += ab;
      if (m == 0) {
        if (p.leaf) p.leaf[row * nleaf128 + (L.sv >> 4)] = leaf;
        acc_sse += (double)leaf;
      }
    }
    acc_abs += (double)ab_row;
    if (WPR == 1 || wv == 0) {
      float total = 0.f;
      int o = 0;
      for (int c = 0; c < (int)prog.nchunks; ++c) {
        const int nl = (int)prog.chunk_leaves[c];
        float cs;
        if (regular) {  // the chunk's blocks: a perfect tree of at most 32 values (zeros beside them add nothing)
          const int nb = nl >> 4;
          cs = np_wave_tree32(lane < nb ? lsum[(o >> 4) + lane] : 0.f);
        } else {
          // level by level in place (a wave's LDS accesses execute in order: the reads of a strip come before its
          // writes, and a strip only writes slots no later strip reads)
          for (int n = nl; n > 1; n >>= 1) {
            for (int i = lane; i < (n >> 1); i += 64) {
              const float a = lsum[o + 2 * i] + lsum[o + 2 * i + 1];
              np_lds_order();
              lsum[o + i] = a;
            }
            np_lds_order();
          }
          cs = lsum[o];
        }
        total = c == 0 ? cs : total + cs;
        o += nl;
      }
      if (lane == 0) {
        if (ent_out) ent_out[row] = total / row_len;  // np.mean: the float32 sum over the count, rounded once
        if (p.scale) p.scale[row] = (uint16_t)h_bits(sp.s);
        if (FMT == MQ_INT_ASYM && p.zp) p.zp[row] = (uint16_t)h_bits(sp.zp);
      }
    }
    group_sync();
    parity ^= 1;
  }
  block_partials(p, acc_sse, acc_abs, lds);
}

template <int FMT, bool BF16>
static int launch_np(int scope, int nv, int team_log2, dim3 grid, size_t lds_bytes, hipStream_t st, const QuantBatch& p) {
#define MQ_NP_ROW(NV, WPR)                                                                                         \
  do {                                                                                                             \
    static bool attr_set = false; /* (more than 64 KiB of dynamic LDS must be asked for once per kernel) */        \
    if (!attr_set) {                                                                                               \
      MQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_np_row<FMT, BF16, NV, WPR>),                     \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));                         \
      attr_set = true;                                                                                             \
    }                                                                                                              \
    hipLaunchKernelGGL((k_np_row<FMT, BF16, NV, WPR>), grid, dim3(256), lds_bytes, st, p);                         \
  } while (0)
  if (scope == SC_STATS_GROUP) {
    switch (team_log2) {
      case -1: hipLaunchKernelGGL((k_np_group3<FMT, BF16>), grid, dim3(256), 0, st, p); break;  // several sizes per read
      case 0: hipLaunchKernelGGL((k_np_group<FMT, BF16, 1>), grid, dim3(256), 0, st, p); break;
      case 1: hipLaunchKernelGGL((k_np_group<FMT, BF16, 2>), grid, dim3(256), 0, st, p); break;
      default: hipLaunchKernelGGL((k_np_group<FMT, BF16, 4>), grid, dim3(256), 0, st, p); break;
    }
  } else if (scope == SC_STATS_ROWNP) {  // rows of 2048 * nv elements
    switch (nv) {
      case 1: hipLaunchKernelGGL((k_np_rowx<FMT, BF16, 1>), grid, dim3(256), 0, st, p); break;
      case 2: hipLaunchKernelGGL((k_np_rowx<FMT, BF16, 2>), grid, dim3(256), 0, st, p); break;
      default: hipLaunchKernelGGL((k_np_rowx<FMT, BF16, 4>), grid, dim3(256), 0, st, p); break;
    }
  } else if (scope == SC_STATS_ROW) {
    switch (nv) {
      case 2: MQ_NP_ROW(2, 1); break;
      case 4: MQ_NP_ROW(4, 1); break;
      default: MQ_NP_ROW(8, 1); break;
    }
  } else {
    switch (nv) {
      case 4: MQ_NP_ROW(4, 4); break;
      case 7: MQ_NP_ROW(7, 4); break;
      case 8: MQ_NP_ROW(8, 4); break;
      case 14: MQ_NP_ROW(14, 4); break;
      default: MQ_NP_ROW(16, 4); break;
    }
  }
#undef MQ_NP_ROW
  return MQ_OK;
}
#endif  // MQ_PART >= 1
