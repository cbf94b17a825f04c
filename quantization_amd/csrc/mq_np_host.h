// Host side and finishing kernels of the numpy-order sweeps (see mq_np.h).  Included by mq_quant.hip, part 0, after
// the sweep-launch helpers.
#pragma once

// ---- finishing kernels --------------------------------------------------------------------------------------------
// float32 values of a wave added as a perfect tree, neighbours first (addition is commutative, so the butterfly's
// operand order does not matter); every lane gets the sum
__device__ __forceinline__ float np_wave_tree(float v) {
  v += f_from_bits(dpp<DPP_XOR1>(f_bits(v)));
  v += f_from_bits(dpp<DPP_XOR2>(f_bits(v)));
  v += f_from_bits(dpp<DPP_HALF_MIRROR>(f_bits(v)));
  v += f_from_bits(dpp<DPP_MIRROR>(f_bits(v)));
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
// The pairwise sum of every 8192-element chunk of the flattened tensors from its 64 leaf sums: a wave per chunk.
__global__ __launch_bounds__(256) void k_np_chunk_sums(const float* leaf, int64_t nchunks, float* cs) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  for (int64_t c = gw; c < nchunks; c += nw) {
    const float s = np_wave_tree(leaf[c * 64 + lane]);
    if (lane == 0) cs[c] = s;
  }
}
// np.mean((a32 - b32) ** 2) of each tensor of the table: its chunks' sums added in order in float32 (numpy's
// reduction adds each chunk's pairwise sum to its running result), then float32(float64(sum) / n).  One workgroup
// per tensor: the sums are staged through LDS 1024 at a time, thread 0 adds.
__global__ __launch_bounds__(256) void k_np_tensor_finish(const SweepItem* tab, int64_t seg_len, const float* leaf_base,
                                                          const float* cs, float* out) {
  __shared__ __attribute__((aligned(16))) float buf[2][2048];
  const SweepItem& it = tab[blockIdx.x];
  const int64_t n = it.nseg * seg_len, nchunks = n / 8192;
  const float* mine = cs + (it.leaf - leaf_base) / 64;
  float total = 0.f;
  // two LDS areas take turns: wave 0's first lane adds the 2048 sums of one while the other waves fetch the next
  // (the chain of additions is the whole cost: sixteen values per four LDS reads, nothing else in the loop)
  auto fetch = [&](int64_t base, float* dst) {
    for (int j = threadIdx.x; j < 2048; j += 256) dst[j] = base + j < nchunks ? mine[base + j] : 0.f;
  };
  fetch(0, buf[0]);
  __syncthreads();
  int cur = 0;
  for (int64_t base = 0; base < nchunks; base += 2048) {
    if (threadIdx.x >= 64) {
      if (base + 2048 < nchunks) {
        for (int j = threadIdx.x - 64; j < 2048; j += 192) buf[cur ^ 1][j] = base + 2048 + j < nchunks ? mine[base + 2048 + j] : 0.f;
      }
    } else if (threadIdx.x == 0) {
      const int m = nchunks - base < 2048 ? (int)(nchunks - base) : 2048;
      const f32x4* b4 = reinterpret_cast<const f32x4*>(buf[cur]);
      int j = 0;
      if (base == 0) {  // the first chunk's sum as it is (numpy starts from it, not from zero)
        const f32x4 a = b4[0];
        total = a[0];
        if (m > 1) total += a[1];
        if (m > 2) total += a[2];
        if (m > 3) total += a[3];
        j = 4;
      }
      for (; j + 16 <= m; j += 16) {
        const f32x4 a = b4[j >> 2], c = b4[(j >> 2) + 1], d = b4[(j >> 2) + 2], e = b4[(j >> 2) + 3];
        total += a[0]; total += a[1]; total += a[2]; total += a[3];
        total += c[0]; total += c[1]; total += c[2]; total += c[3];
        total += d[0]; total += d[1]; total += d[2]; total += d[3];
        total += e[0]; total += e[1]; total += e[2]; total += e[3];
      }
      for (; j < m; ++j) total += buf[cur][j];
    }
    __syncthreads();
    cur ^= 1;
  }
  if (threadIdx.x == 0) out[blockIdx.x] = (float)((double)total / (double)n);
}
// The float64 pairwise sum of every whole 8192-entry chunk of the per-vector lists (vec = np.array(per_vec);
// vec.mean(): quant_error_single_weight.py:74-78): a workgroup per chunk.  The chunk is staged in LDS (each 128-entry
// leaf padded by 8 words: the 8 accumulators of 8 leaves then read 64 different banks), thread t runs accumulator
// t % 8 of leaves t / 8 and t / 8 + 32, eight lanes combine a leaf, a wave combines the 64 leaves.
__global__ __launch_bounds__(256) void k_np_list_chunks(const SweepItem* tab, int count, int64_t total, double* lcs) {
  __shared__ __attribute__((aligned(16))) float buf[64 * 136];
  __shared__ double leafs[64];
  const int t = threadIdx.x;
  for (int64_t c = blockIdx.x; c < total; c += gridDim.x) {
    int lo = 0, hi = count - 1;  // the last list whose first chunk is not past c (lists without a whole chunk share
    while (lo < hi) {            //  their index with the next list, which this skips)
      const int mid = (lo + hi + 1) >> 1;
      if ((int64_t)tab[mid].lchunk0 <= c) lo = mid;
      else hi = mid - 1;
    }
    const SweepItem& it = tab[lo];
    const float* x = reinterpret_cast<const float*>(it.seg_sse) + (c - (int64_t)it.lchunk0) * 8192;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int idx = 4 * (t + 256 * j);
      *reinterpret_cast<f32x4*>(&buf[(idx >> 7) * 136 + (idx & 127)]) = *reinterpret_cast<const f32x4*>(x + idx);
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int lf = (t >> 3) + 32 * h, acc = t & 7;
      double r = (double)buf[lf * 136 + acc];
#pragma unroll
      for (int k = 1; k < 16; ++k) r += (double)buf[lf * 136 + 8 * k + acc];
      r = team_sum<8>(r);  // ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7))
      if (acc == 0) leafs[lf] = r;
    }
    __syncthreads();
    if (t < 64) {
      const double s = team_sum<64>(leafs[t]);
      if (t == 0) lcs[c] = s;
    }
    __syncthreads();
  }
}
// float(vec.mean()) of each list: the whole chunks' sums in order, the last, shorter chunk by its own split program
// (tails[i], built on the host), divided by the count in float64.
__global__ __launch_bounds__(128) void k_np_list_finish(const SweepItem* tab, const ListProg* tails, const double* lcs,
                                                        double* out) {
  __shared__ double leaf[128];
  __shared__ uint16_t off[128];
  const SweepItem& it = tab[blockIdx.x];
  const ListProg& pr = tails[blockIdx.x];
  const int64_t n = it.nseg, nfull = n / 8192;
  const float* x = reinterpret_cast<const float*>(it.seg_sse) + nfull * 8192;
  if (threadIdx.x == 0) {
    uint32_t o = 0;
    for (int i = 0; i < pr.nleaves; ++i) { off[i] = (uint16_t)o; o += pr.len[i]; }
  }
  __syncthreads();
  if ((int)threadIdx.x < pr.nleaves) {
    const float* a = x + off[threadIdx.x];
    const int m = pr.len[threadIdx.x];
    double res;
    if (m < 8) {
      res = 0.0;
      for (int i = 0; i < m; ++i) res += (double)a[i];
    } else {
      double r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = (double)a[j];
      int i = 8;
      for (; i < m - (m % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += (double)a[i + j];
      }
      res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
      for (; i < m; ++i) res += (double)a[i];
    }
    leaf[threadIdx.x] = res;
  }
  __syncthreads();
  double tail = 0.0;
  if (threadIdx.x == 0 && pr.nleaves) tail = numpy_run_prog<double>(pr.ops, pr.nops, &leaf[0]);
  __syncthreads();
  double total = 0.0;
  const double* mine = lcs + it.lchunk0;
  for (int64_t base = 0; base < nfull; base += 128) {
    leaf[threadIdx.x] = base + threadIdx.x < nfull ? mine[base + threadIdx.x] : 0.0;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int m = nfull - base < 128 ? (int)(nfull - base) : 128;
      for (int j = 0; j < m; ++j) total = (base == 0 && j == 0) ? leaf[0] : total + leaf[j];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (pr.nleaves) total = nfull == 0 ? tail : total + tail;
    out[blockIdx.x] = total / (double)n;
  }
}

// ---- host -------------------------------------------------------------------------------------------------------
static size_t np_align16(size_t v) { return (v + 15) / 16 * 16; }
static size_t np_tails_offset(int32_t count) { return np_align16((size_t)count * sizeof(SweepItem)); }
static size_t np_rowprog_offset(int32_t count) { return np_align16(np_tails_offset(count) + (size_t)count * sizeof(ListProg)); }

// numpy's summation of a row of L elements as a leaf table (see NpRowProg); false when it does not fit: a chunk
// whose leaves are not all at one depth, too many leaves, rows that are not whole 128-element runs.
static bool np_row_prog(int64_t L, NpRowProg* pr) {
  *pr = NpRowProg{};
  if (L < 128 || L % 128 || L > 32768) return false;
  bool ok = true;
  uint32_t sv = 0;
  int depth_of_chunk = -1;
  auto rec = [&](auto&& self, int64_t m, int depth) -> void {
    if (!ok) return;
    if (m <= 128) {
      if (depth_of_chunk < 0) depth_of_chunk = depth;
      if (depth != depth_of_chunk || pr->nleaves >= 512 || m % 8) { ok = false; return; }
      pr->leaf[pr->nleaves++] = NpLeaf{(uint16_t)sv, (uint8_t)(m / 8), 1};
      sv += (uint32_t)(m / 8);
    } else {
      int64_t n2 = m / 2;
      n2 -= n2 % 8;
      self(self, n2, depth + 1);
      self(self, m - n2, depth + 1);
    }
  };
  for (int64_t off = 0; off < L && ok; off += 8192) {
    depth_of_chunk = -1;
    const uint32_t before = pr->nleaves;
    rec(rec, L - off < 8192 ? L - off : 8192, 0);
    if (pr->nchunks >= 4) return false;
    pr->chunk_leaves[pr->nchunks++] = pr->nleaves - before;
  }
  if (!ok) return false;
  pr->nrow = pr->nleaves;
  pr->regular = 1;
  for (uint32_t c = 0; c < pr->nchunks; ++c)
    if (pr->chunk_leaves[c] % 16) pr->regular = 0;
  // the 128-element leaves of the flattened tensor (rows are whole leaves): those that are row leaves too are marked,
  // the others are appended
  for (uint32_t f = 0; f < (uint32_t)(L / 128); ++f) {
    bool found = false;
    for (uint32_t i = 0; i < pr->nrow && !found; ++i)
      if (pr->leaf[i].sv == 16 * f && pr->leaf[i].nv == 16) { pr->leaf[i].kind = 3; found = true; }
    if (!found) {
      if (pr->nleaves >= MQ_NPL_MAX) return false;
      pr->leaf[pr->nleaves++] = NpLeaf{(uint16_t)(16 * f), 16, 2};
    }
  }
  return true;
}

// which kernel serves segments of L elements in numpy's order, and its dynamic LDS
static bool np_scope_build(int64_t L, int* scope, int* nv, int* team_log2, size_t* lds_bytes, NpRowProg* pr);
// (the answer for the last segment length is kept per thread: a sweep asks again at every launch)
static bool np_scope(int64_t L, int* scope, int* nv, int* team_log2, size_t* lds_bytes, NpRowProg* pr) {
  static thread_local int64_t last_L = -1;
  static thread_local NpRowProg last_prog;
  static thread_local int last_scope, last_nv, last_team;
  static thread_local size_t last_lds;
  static thread_local bool last_ok;
  if (L != last_L) {
    last_ok = np_scope_build(L, &last_scope, &last_nv, &last_team, &last_lds, &last_prog);
    last_L = L;
  }
  *scope = last_scope; *nv = last_nv; *team_log2 = last_team; *lds_bytes = last_lds;
  if (pr) *pr = last_prog;
  return last_ok;
}
static bool np_scope_build(int64_t L, int* scope, int* nv, int* team_log2, size_t* lds_bytes, NpRowProg* pr) {
  *team_log2 = 0;
  *lds_bytes = 0;
  *nv = -4;
  if (L == 32 || L == 64 || L == 128) {
    *scope = SC_STATS_GROUP;
    *team_log2 = ilog2_ceil(L / 32);
    *pr = NpRowProg{};
    return true;
  }
  if (!np_row_prog(L, pr)) return false;
  if (L == 2048 || L == 4096 || L == 8192) {  // a wave per row, all in registers
    *scope = SC_STATS_ROWNP;
    *nv = (int)(L / 2048);
    return true;
  }
  const int64_t vecs = L / 8;
  int wpr;
  if (vecs <= 512) {
    *scope = SC_STATS_ROW;
    *nv = vecs <= 128 ? 2 : vecs <= 256 ? 4 : 8;
    wpr = 1;
  } else {
    *scope = SC_STATS_ROWBLK;
    *nv = vecs <= 1024 ? 4 : vecs <= 1792 ? 7 : vecs <= 2048 ? 8 : vecs <= 3584 ? 14 : 16;
    wpr = 4;
  }
  const size_t region = (size_t)(L / 128) * MQ_NP_LEAF_WORDS * 4, sums = (size_t)((pr->nrow + 3) & ~3u) * 4;
  *lds_bytes = (wpr == 1 ? 4 : 1) * (region + sums);
  return *lds_bytes <= 92 * 1024;
}

extern "C" size_t mq_sweep_numpy_table_bytes(int32_t count) {
  return np_rowprog_offset(count < 1 ? 1 : count) + sizeof(NpRowProg);
}

extern "C" int mq_sweep_numpy_prepare(const mq_quant_item* items, const int64_t* rows, const int64_t* cols, int32_t count,
                                      int32_t w_dtype, const mq_quant_cfg* cfg, float* leaf_sums, void* table_host,
                                      size_t table_bytes, int64_t* seg_len, int64_t* total_blocks, int64_t* total_leaves,
                                      int64_t* list_chunks) {
  if (!items || !rows || !cols || count < 1 || !table_host || !seg_len || !total_blocks || !total_leaves || !list_chunks ||
      !leaf_sums)
    return MQ_ERR_ARG;
  if (int rc = sweep_check_cfg(w_dtype, cfg)) return rc;
  if (table_bytes < mq_sweep_numpy_table_bytes(count)) return MQ_ERR_WORKSPACE;
  if (!aligned(leaf_sums, 256)) return MQ_ERR_ALIGN;
  const int64_t L = cfg->group_size > 0 ? cfg->group_size : cols[0];
  int scope, nv, team_log2;
  size_t lds_bytes;
  char* base = static_cast<char*>(table_host);
  NpRowProg* rowprog = reinterpret_cast<NpRowProg*>(base + np_rowprog_offset(count));
  if (!np_scope(L, &scope, &nv, &team_log2, &lds_bytes, rowprog)) return MQ_ERR_UNSUPPORTED;
  const mq_quant_out& o0 = items[0].out;
  int64_t unit_bytes = scope == SC_STATS_GROUP ? 4096 : L * 2;
  int64_t units_per_block = sweep_wg_bytes() / unit_bytes / (scope == SC_STATS_ROWBLK ? 1 : 4);
  if (units_per_block < 1) units_per_block = 1;
  if (scope != SC_STATS_ROWBLK) units_per_block *= 4;
  SweepItem* t = reinterpret_cast<SweepItem*>(base);
  ListProg* tails = reinterpret_cast<ListProg*>(base + np_tails_offset(count));
  int64_t blocks = 0, leaves = 0, lchunks = 0;
  for (int32_t i = 0; i < count; ++i) {
    const mq_quant_out& o = items[i].out;
    if (rows[i] < 1 || cols[i] < 1 || !items[i].w) return MQ_ERR_ARG;
    if (cfg->group_size > 0 ? cols[i] % L != 0 : cols[i] != L) return cfg->group_size > 0 ? MQ_ERR_GROUP : MQ_ERR_ARG;
    if (o.deq || o.codes || o.abs_err) return MQ_ERR_UNSUPPORTED;  // statistics only
    if (!o.seg_sse || !o.totals) return MQ_ERR_ARG;                 // the list and the totals are what this form is for
    if (!o.scale != !o0.scale || !o.zero_point != !o0.zero_point) return MQ_ERR_ARG;
    if (!aligned(items[i].w, 16)) return MQ_ERR_UNSUPPORTED;
    if (!aligned(o.scale, 2) || !aligned(o.zero_point, 2) || !aligned(o.seg_sse, 16) || !aligned(o.totals, 8))
      return MQ_ERR_ALIGN;
    const int64_t n = rows[i] * cols[i];
    // whole 8192-element chunks of the flattened tensor (then its leaves are whole 128-element runs and the group
    // kernel has no partial step); inside the 32-bit offsets of the buffer loads
    if (n % 8192 || n * 2 > 0xf0000000ll) return MQ_ERR_UNSUPPORTED;
    const int64_t nseg = n / L;
    const int64_t units = scope == SC_STATS_GROUP ? n / 2048 : nseg;
    int64_t nb = (units + units_per_block - 1) / units_per_block;
    if (nb < 1) nb = 1;
    if (blocks + nb > 0x7fffffffll || lchunks + nseg / 8192 > 0x7fffffffll) return MQ_ERR_TOO_LARGE;
    t[i] = SweepItem{items[i].w, static_cast<uint16_t*>(o.scale), static_cast<uint16_t*>(o.zero_point), o.seg_sse,
                     o.totals, nseg, (uint32_t)blocks, (uint32_t)nb, leaf_sums + leaves, (uint32_t)lchunks, 0u, nullptr, 0u, 0u,
                     nullptr, nullptr, nullptr, 0u, 0u};
    if (!list_prog(nseg % 8192, &tails[i])) return MQ_ERR_UNSUPPORTED;
    blocks += nb;
    leaves += n / 128;
    lchunks += nseg / 8192;
  }
  *seg_len = L;
  *total_blocks = blocks;
  *total_leaves = leaves;
  *list_chunks = lchunks;
  return MQ_OK;
}

extern "C" int mq_sweep_numpy_run(const void* table_dev, int32_t count, int64_t seg_len, int64_t total_blocks,
                                  int32_t w_dtype, const mq_quant_cfg* cfg, void* workspace, size_t workspace_bytes,
                                  void* hip_stream) {
  if (!table_dev || count < 1 || total_blocks < 1 || total_blocks > 0x7fffffffll) return MQ_ERR_ARG;
  if (int rc = sweep_check_cfg(w_dtype, cfg)) return rc;
  int scope, nv, team_log2;
  size_t lds_bytes;
  if (!np_scope(seg_len, &scope, &nv, &team_log2, &lds_bytes, nullptr)) return MQ_ERR_UNSUPPORTED;
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
  p.seg_mse_scale = 1.0 / (double)seg_len;
  p.seg_exact = 1;
  p.sel_mode = SEL_FEED_OFF;
  b.sweep = static_cast<const SweepItem*>(table_dev);
  b.sweep_count = count;
  b.sweep_partials = static_cast<double*>(workspace);
  b.rowprog = reinterpret_cast<const NpRowProg*>(static_cast<const char*>(table_dev) + np_rowprog_offset(count));
  hipEvent_t prof_e0 = nullptr, prof_e1 = nullptr;
  if (prof_on) {
    if (int rc = prof_next_pair(&prof_e0, &prof_e1)) return rc;
    MQ_HIP(hipEventRecord(prof_e0, st));
  }
  const dim3 grid((unsigned)total_blocks);
  const int bf16 = w_dtype == MQ_BF16 ? 1 : 0;
  int rc;
  switch (fmt) {
    case MQ_INT_SYM: rc = mq_launch_np1(bf16, scope, nv, team_log2, grid, lds_bytes, st, b); break;
    case MQ_INT_ASYM: rc = mq_launch_np2(bf16, scope, nv, team_log2, grid, lds_bytes, st, b); break;
    case MQ_LUT_FP: rc = mq_launch_np3(bf16, scope, nv, team_log2, grid, lds_bytes, st, b); break;
    default: rc = mq_launch_np4(bf16, scope, nv, team_log2, grid, lds_bytes, st, b); break;
  }
  if (rc) return rc;
  MQ_HIP(hipGetLastError());
  if (prof_on) MQ_HIP(hipEventRecord(prof_e1, st));
  hipLaunchKernelGGL(k_finalize_sweep, dim3((unsigned)count), dim3(256), 0, st, b);
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}

// The passes of one format with groups of 32, 64 and 128 elements from ONE read of the weights (k_np_group3): tables
// of mq_sweep_numpy_prepare calls over the same tensors with those group sizes (null: size not wanted; at least two).
extern "C" size_t mq_sweep_numpy_groups_workspace_bytes(int64_t total_blocks) { return 3 * mq_sweep_workspace_bytes(total_blocks); }
extern "C" int mq_sweep_numpy_run_groups(const void* table32_dev, const void* table64_dev, const void* table128_dev,
                                         int32_t count, int64_t total_blocks, int32_t w_dtype, const mq_quant_cfg* cfg,
                                         void* workspace, size_t workspace_bytes, void* hip_stream) {
  const void* tabs[3] = {table32_dev, table64_dev, table128_dev};
  const int wanted = (tabs[0] ? 1 : 0) + (tabs[1] ? 1 : 0) + (tabs[2] ? 1 : 0);
  if (wanted < 2 || count < 1 || total_blocks < 1 || total_blocks > 0x7fffffffll / 3) return MQ_ERR_ARG;
  if (int rc = sweep_check_cfg(w_dtype, cfg)) return rc;
  if (!workspace || workspace_bytes < mq_sweep_numpy_groups_workspace_bytes(total_blocks)) return MQ_ERR_WORKSPACE;
  if (!aligned(workspace, 16) || !aligned(tabs[0], 16) || !aligned(tabs[1], 16) || !aligned(tabs[2], 16)) return MQ_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  QuantBatch b = {};
  QuantParams& p = b.common;
  p.in_dtype = w_dtype;
  p.L = 32;
  p.vecs_per_seg = 4;
  const int fmt = cfg->format;
  if (fmt == MQ_INT_SYM) p.qmax = (float)((1 << (cfg->bits - 1)) - 1);
  else if (fmt == MQ_INT_ASYM) p.qmax = (float)((1 << cfg->bits) - 1);
  else if (!lut_consts(cfg, false, &p.lut, &p.qmax)) return MQ_ERR_UNSUPPORTED;
  p.rqmax = 1.0f / p.qmax;
  p.qmax_recip_ok = recip_ok_cached(p.qmax);
  p.need_err = 1;
  p.seg_exact = 1;
  p.sel_mode = SEL_FEED_OFF;
  for (int c = 0; c < 3; ++c) b.np_tabs[c] = static_cast<const SweepItem*>(tabs[c]);
  b.sweep_count = count;
  b.sweep_partials = static_cast<double*>(workspace);
  hipEvent_t prof_e0 = nullptr, prof_e1 = nullptr;
  if (prof_on) {
    if (int rc = prof_next_pair(&prof_e0, &prof_e1)) return rc;
    MQ_HIP(hipEventRecord(prof_e0, st));
  }
  const dim3 grid((unsigned)total_blocks);
  const int bf16 = w_dtype == MQ_BF16 ? 1 : 0;
  int rc;
  switch (fmt) {
    case MQ_INT_SYM: rc = mq_launch_np1(bf16, SC_STATS_GROUP, -4, -1, grid, 0, st, b); break;
    case MQ_INT_ASYM: rc = mq_launch_np2(bf16, SC_STATS_GROUP, -4, -1, grid, 0, st, b); break;
    case MQ_LUT_FP: rc = mq_launch_np3(bf16, SC_STATS_GROUP, -4, -1, grid, 0, st, b); break;
    default: rc = mq_launch_np4(bf16, SC_STATS_GROUP, -4, -1, grid, 0, st, b); break;
  }
  if (rc) return rc;
  MQ_HIP(hipGetLastError());
  if (prof_on) MQ_HIP(hipEventRecord(prof_e1, st));
  for (int c = 0; c < 3; ++c) {  // each size's totals from its own partial sums
    if (!tabs[c]) continue;
    QuantBatch f = {};
    f.sweep = static_cast<const SweepItem*>(tabs[c]);
    f.sweep_count = count;
    f.sweep_partials = static_cast<double*>(workspace) + 2 * (int64_t)c * total_blocks;
    hipLaunchKernelGGL(k_finalize_sweep, dim3((unsigned)count), dim3(256), 0, st, f);
    MQ_HIP(hipGetLastError());
  }
  return MQ_OK;
}

static size_t np_finish_lcs_offset(int64_t total_leaves) { return np_align16((size_t)(total_leaves / 64 + 1) * sizeof(float)); }
extern "C" size_t mq_sweep_numpy_finish_workspace_bytes(int64_t total_leaves, int64_t list_chunks) {
  return np_finish_lcs_offset(total_leaves < 0 ? 0 : total_leaves) + (size_t)((list_chunks < 0 ? 0 : list_chunks) + 1) * sizeof(double);
}

extern "C" int mq_sweep_numpy_finish(const void* table_dev, int32_t count, int64_t seg_len, const float* leaf_sums,
                                     int64_t total_leaves, int64_t list_chunks, float* mse_tensor, double* mse_rep,
                                     void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!table_dev || count < 1 || seg_len < 8 || !leaf_sums || total_leaves < 64 || total_leaves % 64 || list_chunks < 0 ||
      (!mse_tensor && !mse_rep))
    return MQ_ERR_ARG;
  if (!workspace || workspace_bytes < mq_sweep_numpy_finish_workspace_bytes(total_leaves, list_chunks)) return MQ_ERR_WORKSPACE;
  if (!aligned(workspace, 16) || !aligned(table_dev, 16) || !aligned(mse_tensor, 4) || !aligned(mse_rep, 8)) return MQ_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const SweepItem* tab = static_cast<const SweepItem*>(table_dev);
  const ListProg* tails = reinterpret_cast<const ListProg*>(static_cast<const char*>(table_dev) + np_tails_offset(count));
  float* cs = static_cast<float*>(workspace);
  double* lcs = reinterpret_cast<double*>(static_cast<char*>(workspace) + np_finish_lcs_offset(total_leaves));
  if (mse_tensor) {
    const int64_t nchunks = total_leaves / 64, wb = (nchunks + 3) / 4;
    hipLaunchKernelGGL(k_np_chunk_sums, dim3((unsigned)(wb < 4096 ? wb : 4096)), dim3(256), 0, st, leaf_sums, nchunks, cs);
    MQ_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_np_tensor_finish, dim3((unsigned)count), dim3(256), 0, st, tab, seg_len, leaf_sums, cs, mse_tensor);
    MQ_HIP(hipGetLastError());
  }
  if (mse_rep) {
    if (list_chunks > 0) {
      hipLaunchKernelGGL(k_np_list_chunks, dim3((unsigned)(list_chunks < 4096 ? list_chunks : 4096)), dim3(256), 0, st, tab,
                         (int)count, list_chunks, lcs);
      MQ_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_np_list_finish, dim3((unsigned)count), dim3(128), 0, st, tab, tails, lcs, mse_rep);
    MQ_HIP(hipGetLastError());
  }
  return MQ_OK;
}
