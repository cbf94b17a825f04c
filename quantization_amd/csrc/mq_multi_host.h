// Host side of the several-groupings sweep launch (k_stats_multi, mq_multi.h); included by mq_quant.hip (part 0).
#pragma once

// the kernel's shape for rows of `cols` elements: runs of 32 elements per lane (NR) and waves per row (WPR), or false
static bool multi_shape_any(int64_t cols, int* nr, int* wpr) {
  if (cols < 128 || cols % 128 || cols > 32768) return false;
  const int64_t runs = cols / 32;
  *wpr = runs <= 256 ? 1 : 4;
  *nr = (int)((runs + 64 * *wpr - 1) / (64 * *wpr));
  return true;
}

extern "C" size_t mq_sweep_multi_table_bytes(int32_t count) { return (size_t)(count < 1 ? 1 : count) * sizeof(MultiItem); }
extern "C" size_t mq_sweep_multi_workspace_bytes(int64_t total_blocks) {
  return (size_t)(total_blocks < 1 ? 1 : total_blocks) * 8 * sizeof(double);
}

extern "C" int mq_sweep_multi_prepare(const mq_multi_item* items, const int64_t* rows, int64_t cols, int32_t count,
                                      int32_t w_dtype, const mq_quant_cfg* cfg, uint32_t mask, void* table_host,
                                      size_t table_bytes, int64_t* total_blocks) {
  if (!items || !rows || count < 1 || !table_host || !total_blocks || !(mask & 15u) || (mask & ~15u)) return MQ_ERR_ARG;
  if (int rc = sweep_check_cfg(w_dtype, cfg)) return rc;
  if (table_bytes < mq_sweep_multi_table_bytes(count)) return MQ_ERR_WORKSPACE;
  int nr, wpr;
  if (!multi_shape_any(cols, &nr, &wpr)) return MQ_ERR_UNSUPPORTED;
  // workgroups per tensor: one per sweep_wg_bytes() of weights, in whole rows (a wave per row: four rows at a time)
  int64_t rows_per_block = sweep_wg_bytes() / (cols * 2) / (wpr == 4 ? 1 : 4);
  if (rows_per_block < 1) rows_per_block = 1;
  if (wpr == 1) rows_per_block *= 4;
  MultiItem* t = static_cast<MultiItem*>(table_host);
  int64_t blocks = 0;
  for (int32_t i = 0; i < count; ++i) {
    if (rows[i] < 1 || !items[i].w) return MQ_ERR_ARG;
    if (!aligned(items[i].w, 16)) return MQ_ERR_UNSUPPORTED;
    if (rows[i] * cols * 2 > 0xf0000000ll) return MQ_ERR_UNSUPPORTED;  // the 32-bit offsets of the buffer loads
    MultiItem m = {};
    m.w = items[i].w;
    for (int c = 0; c < 4; ++c) {
      if (!((mask >> c) & 1u)) continue;
      if (!aligned(items[i].list[c], 4) || !aligned(items[i].totals[c], 8)) return MQ_ERR_ALIGN;
      m.list[c] = static_cast<float*>(items[i].list[c]);
      m.totals[c] = items[i].totals[c];
    }
    int64_t nb = (rows[i] + rows_per_block - 1) / rows_per_block;
    if (blocks + nb > 0x7fffffffll) return MQ_ERR_TOO_LARGE;
    m.nrow = rows[i];
    m.blk0 = (uint32_t)blocks;
    m.nblk = (uint32_t)nb;
    t[i] = m;
    blocks += nb;
  }
  *total_blocks = blocks;
  return MQ_OK;
}

static int sweep_multi_launch(const void* table_dev, int32_t count, int64_t cols, int64_t total_blocks, int32_t w_dtype,
                              const mq_quant_cfg* cfg, uint32_t mask, void* workspace, size_t workspace_bytes,
                              void* hip_stream, bool finalize);
extern "C" int mq_sweep_multi_run(const void* table_dev, int32_t count, int64_t cols, int64_t total_blocks, int32_t w_dtype,
                                  const mq_quant_cfg* cfg, uint32_t mask, void* workspace, size_t workspace_bytes,
                                  void* hip_stream) {
  return sweep_multi_launch(table_dev, count, cols, total_blocks, w_dtype, cfg, mask, workspace, workspace_bytes, hip_stream,
                            true);
}
// The two halves of mq_sweep_multi_run for callers that keep the small totals launch off the stream of the big ones
// (a workspace of its own per launch in flight): _launch enqueues the statistics kernel only, _finalize adds the
// workgroups' partial sums into the tensors' totals -- on any stream that is ordered behind the launch.
extern "C" int mq_sweep_multi_launch(const void* table_dev, int32_t count, int64_t cols, int64_t total_blocks,
                                     int32_t w_dtype, const mq_quant_cfg* cfg, uint32_t mask, void* workspace,
                                     size_t workspace_bytes, void* hip_stream) {
  return sweep_multi_launch(table_dev, count, cols, total_blocks, w_dtype, cfg, mask, workspace, workspace_bytes, hip_stream,
                            false);
}
extern "C" int mq_sweep_multi_finalize(const void* table_dev, int32_t count, int64_t total_blocks, void* workspace,
                                       size_t workspace_bytes, void* hip_stream) {
  if (!table_dev || count < 1 || total_blocks < 1) return MQ_ERR_ARG;
  if (!workspace || workspace_bytes < mq_sweep_multi_workspace_bytes(total_blocks)) return MQ_ERR_WORKSPACE;
  if (!aligned(workspace, 16) || !aligned(table_dev, 8)) return MQ_ERR_ALIGN;
  QuantBatch b = {};
  b.multi = static_cast<const MultiItem*>(table_dev);
  b.multi_count = count;
  b.sweep_partials = static_cast<double*>(workspace);
  hipLaunchKernelGGL(k_finalize_multi, dim3((unsigned)count, 4), dim3(256), 0, static_cast<hipStream_t>(hip_stream), b);
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}
static int sweep_multi_launch(const void* table_dev, int32_t count, int64_t cols, int64_t total_blocks, int32_t w_dtype,
                              const mq_quant_cfg* cfg, uint32_t mask, void* workspace, size_t workspace_bytes,
                              void* hip_stream, bool finalize) {
  if (!table_dev || count < 1 || total_blocks < 1 || total_blocks > 0x7fffffffll || !(mask & 15u) || (mask & ~15u))
    return MQ_ERR_ARG;
  if (int rc = sweep_check_cfg(w_dtype, cfg)) return rc;
  int nr, wpr;
  if (!multi_shape_any(cols, &nr, &wpr)) return MQ_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < mq_sweep_multi_workspace_bytes(total_blocks)) return MQ_ERR_WORKSPACE;
  if (!aligned(workspace, 16) || !aligned(table_dev, 8)) return MQ_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  QuantBatch b = {};
  QuantParams& p = b.common;
  p.in_dtype = w_dtype;
  p.L = cols;
  p.vecs_per_seg = (int32_t)(cols / 8);
  const int fmt = cfg->format;
  if (fmt == MQ_INT_SYM) p.qmax = (float)((1 << (cfg->bits - 1)) - 1);
  else if (fmt == MQ_INT_ASYM) p.qmax = (float)((1 << cfg->bits) - 1);
  else if (!lut_consts(cfg, false, &p.lut, &p.qmax)) return MQ_ERR_UNSUPPORTED;
  p.rqmax = 1.0f / p.qmax;
  p.qmax_recip_ok = recip_ok_cached(p.qmax);
  p.need_err = 1;
  p.sel_mode = SEL_FEED_OFF;
  b.multi = static_cast<const MultiItem*>(table_dev);
  b.multi_count = count;
  // (measurements: MQ_MULTI_NOSHARE=1 keeps every grouping on its own power of two, see pk_div_params)
  static const bool noshare = getenv("MQ_MULTI_NOSHARE") && atoi(getenv("MQ_MULTI_NOSHARE")) != 0;
  // (MQ_MULTI_NOSKIP=1: waves compute the empty runs of rows that do not fill the kernel's grid)
  static const bool noskip = getenv("MQ_MULTI_NOSKIP") && atoi(getenv("MQ_MULTI_NOSKIP")) != 0;
  b.multi_mask = mask | (noshare ? 0x80000000u : 0u) | (noskip ? 0x40000000u : 0u);
  b.sweep_partials = static_cast<double*>(workspace);
  hipEvent_t prof_e0 = nullptr, prof_e1 = nullptr;
  if (prof_on) {
    if (int rc = prof_next_pair(&prof_e0, &prof_e1)) return rc;
    MQ_HIP(hipEventRecord(prof_e0, st));
  }
  const dim3 grid((unsigned)total_blocks);
  int rc;
  const int bf16 = w_dtype == MQ_BF16 ? 1 : 0;
  switch (fmt) {
    case MQ_INT_SYM: rc = mq_launch_multi1(bf16, nr, wpr, grid, st, b); break;
    case MQ_INT_ASYM: rc = mq_launch_multi2(bf16, nr, wpr, grid, st, b); break;
    case MQ_LUT_FP: rc = mq_launch_multi3(bf16, nr, wpr, grid, st, b); break;
    default: rc = mq_launch_multi4(bf16, nr, wpr, grid, st, b); break;
  }
  if (rc) return rc;
  MQ_HIP(hipGetLastError());
  if (prof_on) MQ_HIP(hipEventRecord(prof_e1, st));
  if (finalize) {
    hipLaunchKernelGGL(k_finalize_multi, dim3((unsigned)count, 4), dim3(256), 0, st, b);
    MQ_HIP(hipGetLastError());
  }
  return MQ_OK;
}
