// Host side of the sweeps that select the lists' percentiles inside the pass (mq_sel.h: ListFeedState, ListJob).
// Included by mq_quant.hip, part 0, after the sweep-launch helpers.
#pragma once

struct FeedChoice {  // how one tensor's list is sampled and bracketed
  uint32_t sunit, scount, ccap;
  int64_t m, k0, k1, s_lo, s_hi;
  double gamma;
};
// Is the list of nseg entries (groups of 32 << team_log2 elements) worth feeding?  The sample: one whole wave step
// of every sunit >= 16 steps (at most 1/16 of the weights is read twice), up to 32768 entries.
static bool feed_choice(int64_t nseg, int team_log2, double pct, FeedChoice* c) {
  const int64_t spw = 64 >> team_log2, nsteps = nseg / spw, want = 32768 / spw;
  if (nsteps < 16) return false;
  int64_t unit = nsteps / want;
  if (unit < 16) unit = 16;
  const int64_t cnt = nsteps / unit;
  if (cnt < 1 || unit > 0x7fffffffll) return false;
  c->sunit = (uint32_t)unit;
  c->scount = (uint32_t)cnt;
  c->m = cnt * spw;
  if (!sel_list_ranks(nseg, c->m, pct, &c->k0, &c->k1, &c->gamma, &c->s_lo, &c->s_hi)) return false;
  // the bracket spans (s_hi - s_lo) of m sample ranks: about that share of the entries, twice over, plus slack
  const double share = (double)(c->s_hi - c->s_lo + 2) / (double)c->m;
  const double cap = 2.0 * share * (double)nseg + 4096.0;
  c->ccap = cap > 4.0e9 ? 0xffffffffu : (uint32_t)cap;
  return true;
}
static size_t feed_align(size_t v) { return (v + 255) / 256 * 256; }
// workgroups mq_sweep_prepare gives a tensor of nseg groups (its formula for the group kernels)
static int64_t feed_tensor_blocks(int64_t nseg, int team_log2) {
  int64_t units_per_block = sweep_wg_bytes() / 4096 / 4;
  if (units_per_block < 1) units_per_block = 1;
  units_per_block *= 4;
  const int64_t spw = 64 >> team_log2, units = (nseg + spw - 1) / spw;
  const int64_t nb = (units + units_per_block - 1) / units_per_block;
  return nb < 1 ? 1 : nb;
}
// 32-bit words of feed_ws one fed tensor takes: samples, packed kept entries, the waves' counts and slots
static size_t feed_tensor_words(const FeedChoice& c, int64_t nblk) {
  auto r = [](size_t v) { return (v + 63) / 64 * 64; };
  return r((size_t)c.m) + r((size_t)c.ccap) + r((size_t)nblk * 4 * 2) + (size_t)nblk * 4 * LIST_SLOT;
}

// feed_ws (device, one per pass): [ListFeedState x nfed], then per fed tensor its samples, room for the packed kept
// entries, its workgroups' counts and slots
extern "C" int mq_sweep_feed_sizes(const int64_t* rows, const int64_t* cols, int32_t count, const mq_quant_cfg* cfg,
                                   double pct, int32_t* nfed, size_t* feed_ws_bytes, size_t* aux_bytes) {
  if (!rows || !cols || count < 1 || !cfg || !nfed || !feed_ws_bytes || !aux_bytes) return MQ_ERR_ARG;
  if (!(pct >= 0.0 && pct <= 100.0)) return MQ_ERR_ARG;
  *nfed = 0;
  *feed_ws_bytes = 0;
  *aux_bytes = 0;
  int scope, nv, team_log2;
  if (cfg->group_size <= 0 || !sweep_scope(cfg->group_size, &scope, &nv, &team_log2) || scope != SC_STATS_GROUP)
    return MQ_OK;  // rows: their lists are short
  size_t floats = 0;
  int n = 0;
  for (int32_t i = 0; i < count; ++i) {
    if (rows[i] < 1 || cols[i] < 1) return MQ_ERR_ARG;
    FeedChoice c;
    const int64_t nseg = rows[i] * cols[i] / cfg->group_size;
    if (!feed_choice(nseg, team_log2, pct, &c)) continue;
    ++n;
    floats += feed_tensor_words(c, feed_tensor_blocks(nseg, team_log2));
  }
  *nfed = n;
  if (n) {
    *feed_ws_bytes = feed_align((size_t)n * sizeof(ListFeedState)) + floats * 4;
    *aux_bytes = feed_align((size_t)n * sizeof(SweepItem)) + (size_t)n * sizeof(ListJob);
  }
  return MQ_OK;
}

// mq_sweep_prepare plus the feed: the main table's entries of the fed tensors point at their state and candidate
// list inside feed_ws; aux_host receives the sampling launch's table and the selection jobs (the caller copies it to
// the device).  fed_rows[f] (room for count entries) receives the table row of the f-th fed tensor; the percentiles
// of the fed tensors go to p99_out[row] (mq_sweep_feed_finish).
extern "C" int mq_sweep_feed_prepare(const mq_quant_item* items, const int64_t* rows, const int64_t* cols, int32_t count,
                                     int32_t w_dtype, const mq_quant_cfg* cfg, double pct, void* feed_ws_dev,
                                     size_t feed_ws_bytes, void* table_dev, void* table_host, size_t table_bytes, void* aux_host,
                                     size_t aux_bytes, int64_t* seg_len, int64_t* total_blocks, int64_t* sample_blocks,
                                     int32_t* nfed_out, int32_t* fed_rows) {
  if (!sample_blocks || !nfed_out || !feed_ws_dev || !aux_host || !fed_rows || !table_dev) return MQ_ERR_ARG;
  int32_t nfed = 0;
  size_t need_ws = 0, need_aux = 0;
  if (int rc = mq_sweep_feed_sizes(rows, cols, count, cfg, pct, &nfed, &need_ws, &need_aux)) return rc;
  if (nfed < 1) return MQ_ERR_UNSUPPORTED;
  if (feed_ws_bytes < need_ws || aux_bytes < need_aux) return MQ_ERR_WORKSPACE;
  if (!aligned(feed_ws_dev, 256)) return MQ_ERR_ALIGN;
  if (int rc = mq_sweep_prepare(items, rows, cols, count, w_dtype, cfg, table_host, table_bytes, seg_len, total_blocks)) return rc;
  if (items[0].out.seg_mse_f32 != 1 || !items[0].out.seg_sse) return MQ_ERR_UNSUPPORTED;  // float32 lists only
  int scope, nv, team_log2;
  sweep_scope(*seg_len, &scope, &nv, &team_log2);
  SweepItem* t = static_cast<SweepItem*>(table_host);
  SweepItem* st = static_cast<SweepItem*>(aux_host);
  ListJob* jobs = reinterpret_cast<ListJob*>(static_cast<char*>(aux_host) + feed_align((size_t)nfed * sizeof(SweepItem)));
  ListFeedState* state = static_cast<ListFeedState*>(feed_ws_dev);
  float* pool = reinterpret_cast<float*>(static_cast<char*>(feed_ws_dev) + feed_align((size_t)nfed * sizeof(ListFeedState)));
  int f = 0;
  int64_t sblocks = 0;
  for (int32_t i = 0; i < count; ++i) {
    FeedChoice c;
    if (!feed_choice(t[i].nseg, team_log2, pct, &c)) continue;
    if ((int64_t)t[i].nblk != feed_tensor_blocks(t[i].nseg, team_log2)) return MQ_ERR_UNSUPPORTED;  // (sizes assumed it)
    auto r64 = [](size_t v) { return (v + 63) / 64 * 64; };
    float* samples = pool;
    pool += r64((size_t)c.m);
    float* cand = pool;
    pool += r64((size_t)c.ccap);
    uint32_t* wg_cnt = reinterpret_cast<uint32_t*>(pool);
    pool += r64((size_t)t[i].nblk * 4 * 2);
    uint32_t* slots = reinterpret_cast<uint32_t*>(pool);
    pool += (size_t)t[i].nblk * 4 * LIST_SLOT;
    t[i].feed = state + f;
    t[i].wg_cnt = wg_cnt;
    t[i].slots = slots;
    // the sampling launch: a workgroup per 16 sampled steps (four per wave)
    const int64_t nb = ((int64_t)c.scount + 15) / 16;
    st[f] = t[i];  // (the repair launch walks this table too: it keeps the list pointer)
    st[f].scale = nullptr; st[f].zp = nullptr; st[f].totals = nullptr;
    st[f].blk0 = (uint32_t)sblocks;
    st[f].nblk = (uint32_t)nb;
    st[f].sample = samples;
    st[f].sunit = c.sunit;
    st[f].scount = c.scount;
    sblocks += nb;
    ListJob j = {};
    j.samples = samples;
    j.cand = cand;
    j.list = reinterpret_cast<const float*>(t[i].seg_sse);
    j.wg_cnt = wg_cnt;
    j.slots = slots;
    j.brk = &static_cast<SweepItem*>(table_dev)[i].blo;
    j.nblk = t[i].nblk * 4u;
    j.gamma = c.gamma;
    j.k0 = (uint64_t)c.k0;
    j.k1 = (uint64_t)c.k1;
    j.m = (uint32_t)c.m;
    j.s_lo = (uint32_t)c.s_lo;
    j.s_hi = (uint32_t)c.s_hi;
    j.ccap = c.ccap;
    j.n = (uint32_t)t[i].nseg;
    j.out = (uint32_t)i;
    j.slot = (uint32_t)f;
    jobs[f] = j;
    fed_rows[f] = i;
    ++f;
  }
  *sample_blocks = sblocks;
  *nfed_out = nfed;
  return MQ_OK;
}

static int feed_batch_params(QuantBatch* b, int64_t seg_len, int32_t w_dtype, const mq_quant_cfg* cfg, int* scope, int* nv,
                             int* team_log2) {
  if (int rc = sweep_check_cfg(w_dtype, cfg)) return rc;
  if (!sweep_scope(seg_len, scope, nv, team_log2) || *scope != SC_STATS_GROUP) return MQ_ERR_UNSUPPORTED;
  QuantParams& p = b->common;
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
  p.seg_exact = 0;
  p.sel_mode = SEL_FEED_OFF;
  return MQ_OK;
}
static void feed_launch(int fmt, int scope, int nv, int team_log2, dim3 grid, hipStream_t st, const QuantBatch& b) {
  switch (fmt) {
    case MQ_INT_SYM: mq_launch_fmt1(SEL_STATS_ONLY, scope, nv, 256, team_log2, grid, 0, st, b); break;
    case MQ_INT_ASYM: mq_launch_fmt2(SEL_STATS_ONLY, scope, nv, 256, team_log2, grid, 0, st, b); break;
    case MQ_LUT_FP: mq_launch_fmt3(SEL_STATS_ONLY, scope, nv, 256, team_log2, grid, 0, st, b); break;
    default: mq_launch_fmt4(SEL_STATS_ONLY, scope, nv, 256, team_log2, grid, 0, st, b); break;
  }
}

// Before the pass (any stream the pass is ordered behind): clears the feed states, evaluates the sampled steps,
// turns each tensor's samples into its bracket.
extern "C" int mq_sweep_feed_sample(const void* aux_dev, int32_t nfed, int64_t seg_len, int64_t sample_blocks,
                                    int32_t w_dtype, const mq_quant_cfg* cfg, void* feed_ws_dev, void* hip_stream) {
  if (!aux_dev || nfed < 1 || sample_blocks < 1 || sample_blocks > 0x7fffffffll || !feed_ws_dev) return MQ_ERR_ARG;
  if (!aligned(aux_dev, 16) || !aligned(feed_ws_dev, 256)) return MQ_ERR_ALIGN;
  QuantBatch b = {};
  int scope, nv, team_log2;
  if (int rc = feed_batch_params(&b, seg_len, w_dtype, cfg, &scope, &nv, &team_log2)) return rc;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  MQ_HIP(hipMemsetAsync(feed_ws_dev, 0, (size_t)nfed * sizeof(ListFeedState), st));
  b.common.list_mode = LIST_SAMPLE;
  b.sweep = static_cast<const SweepItem*>(aux_dev);
  b.sweep_count = nfed;
  b.sweep_partials = nullptr;  // (the sampling launch writes no partial sums)
  feed_launch(cfg->format, scope, nv, team_log2, dim3((unsigned)sample_blocks), st, b);
  MQ_HIP(hipGetLastError());
  const ListJob* jobs = reinterpret_cast<const ListJob*>(static_cast<const char*>(aux_dev) + feed_align((size_t)nfed * sizeof(SweepItem)));
  return sel_list_bracket(jobs, nfed, st);
}

// The pass: mq_sweep_run with the fed tensors' lists counted and kept instead of written.
extern "C" int mq_sweep_feed_run(const void* table_dev, int32_t count, int64_t seg_len, int64_t total_blocks,
                                 int32_t w_dtype, const mq_quant_cfg* cfg, void* workspace, size_t workspace_bytes,
                                 void* hip_stream) {
  if (!table_dev || count < 1 || total_blocks < 1 || total_blocks > 0x7fffffffll) return MQ_ERR_ARG;
  if (!workspace || workspace_bytes < mq_sweep_workspace_bytes(total_blocks)) return MQ_ERR_WORKSPACE;
  if (!aligned(workspace, 16) || !aligned(table_dev, 16)) return MQ_ERR_ALIGN;
  QuantBatch b = {};
  int scope, nv, team_log2;
  if (int rc = feed_batch_params(&b, seg_len, w_dtype, cfg, &scope, &nv, &team_log2)) return rc;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  b.common.list_mode = LIST_FEED;
  b.sweep = static_cast<const SweepItem*>(table_dev);
  b.sweep_count = count;
  b.sweep_partials = static_cast<double*>(workspace);
  hipEvent_t prof_e0 = nullptr, prof_e1 = nullptr;
  if (prof_on) {
    if (int rc = prof_next_pair(&prof_e0, &prof_e1)) return rc;
    MQ_HIP(hipEventRecord(prof_e0, st));
  }
  feed_launch(cfg->format, scope, nv, team_log2, dim3((unsigned)total_blocks), st, b);
  MQ_HIP(hipGetLastError());
  if (prof_on) MQ_HIP(hipEventRecord(prof_e1, st));
  hipLaunchKernelGGL(k_finalize_sweep, dim3((unsigned)count), dim3(256), 0, st, b);
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}

// After the pass: the wanted ranks among the kept entries -> p99_out[tensor]; for a tensor whose bracket missed the
// repair launches (they return at once otherwise) write its list after all and select from it.
extern "C" int mq_sweep_feed_finish(const void* aux_dev, int32_t nfed, int64_t seg_len, int64_t sample_blocks,
                                    int32_t w_dtype, const mq_quant_cfg* cfg, void* feed_ws_dev, double* p99_out,
                                    void* hip_stream) {
  if (!aux_dev || nfed < 1 || !feed_ws_dev || !p99_out || sample_blocks < 1 || sample_blocks > 0x7fffffffll)
    return MQ_ERR_ARG;
  if (!aligned(p99_out, 8)) return MQ_ERR_ALIGN;
  QuantBatch b = {};
  int scope, nv, team_log2;
  if (int rc = feed_batch_params(&b, seg_len, w_dtype, cfg, &scope, &nv, &team_log2)) return rc;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const ListJob* jobs = reinterpret_cast<const ListJob*>(static_cast<const char*>(aux_dev) + feed_align((size_t)nfed * sizeof(SweepItem)));
  ListFeedState* state = static_cast<ListFeedState*>(feed_ws_dev);
  if (int rc = sel_list_cand(jobs, nfed, state, p99_out, st)) return rc;
  // the repair launch walks the fed tensors with the sampling launch's (small) grid: nearly always it has nothing
  // to do, and then it costs a few microseconds
  b.common.list_mode = LIST_REPAIR;
  b.sweep = static_cast<const SweepItem*>(aux_dev);
  b.sweep_count = nfed;
  b.sweep_partials = nullptr;  // (it writes no partial sums)
  feed_launch(cfg->format, scope, nv, team_log2, dim3((unsigned)sample_blocks), st, b);
  MQ_HIP(hipGetLastError());
  return sel_list_repair(jobs, nfed, state, p99_out, st);
}
