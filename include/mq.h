/*
 * mq.h -- C ABI of libmq.so: MI355X (gfx950) kernels for the weight quantize -> dequantize -> error path.
 *
 * Plain C: raw device pointers, sizes, a HIP stream handle passed as void*.  No C++ or torch types.
 * Every function enqueues on the caller's stream and returns without synchronising unless it says
 * otherwise.  All buffers are allocated by the caller and the library keeps no pointer to them past the call.
 * State the library does keep: per calling thread, the last HIP error text, the event pool of the
 * mq_profile_* hook and a small cache of verified reciprocal constants; per process, the grid cap and the
 * workgroup grain of the sweep launches, read once from the environment variables MQ_MAX_BLOCKS and
 * MQ_SWEEP_WG_KB (tuning runs) and, per device, the address of the static scratch
 * array of mq_error_stats_static.  Functions return MQ_OK (0) or a negative mq_status and never throw.
 *
 * "Reference" citations are into /root/reference (weirdmaverick/quantization).
 */
#ifndef MQ_H_
#define MQ_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  MQ_F16 = 0, MQ_BF16 = 1, MQ_F32 = 2, MQ_F64 = 3,
  MQ_F32_AS_F64 = 4 /* percentile functions only: a float32 array standing for the float64 array that holds the same
                       values -- vec = np.array(per_vec) of float(np.float32) means in the analysis scripts
                       (quant_error_single_weight.py:46-48, :74-79): the selection runs on the float32 bits (two
                       passes), ranks, interpolation and results are float64, bit for bit what np.percentile
                       returns for the float64 array */
} mq_dtype;

/* Quantization formats of the hot path. */
typedef enum {
  MQ_INT_SYM = 0,  /* quant_int / _int_quant_meta(asym=False)      quant_weight.py:229-266, :768-775 */
  MQ_INT_ASYM = 1, /* quant_int_asym / _int_quant_meta(asym=True)  quant_weight.py:298-335, :756-766 */
  MQ_LUT_FP = 2,   /* quant_datatype / _fp_quant_meta value tables quant_weight.py:503-601, :778-872:
                      tables without subnormals (FP8_E4M3 :114, FP8_E3M4 :91, FP8_E2M5 :65)              */
  MQ_LUT_FP_SUB = 3 /* the same for tables WITH subnormals: +-m*2^(1-bias-M) below 2^(1-bias), then
                      +-(1+m/2^M)*2^(e-bias), e = 1 .. 2^E-1: FP4_E2M1 (2,1,0) :22, FP5_E2M2 (2,2,-1) :38,
                      FP5_E3M1 (3,1,0) :40, FP6_E2M3 (2,3,-2) :50, FP6_E3M2 (3,2,-1) :56; exp_bits = 0 is the
                      uniform grid alone: INTn lists as tables (n-1 mantissa bits, bias 2-n) :10,:18,:34,:44,:64 */
} mq_format;

typedef enum {
  MQ_OK = 0,
  MQ_ERR_ARG = -1,         /* null pointer, negative size, unknown enum            -> ValueError   */
  MQ_ERR_GROUP = -2,       /* cols % group_size != 0                               -> RuntimeError */
  MQ_ERR_UNSUPPORTED = -3, /* bit width / format outside the supported set         -> ValueError   */
  MQ_ERR_WORKSPACE = -4,   /* workspace missing or smaller than mq_*_workspace_bytes               */
  MQ_ERR_HIP = -5,         /* a HIP call failed (see mq_last_hip_error)                            */
  MQ_ERR_ALIGN = -6,       /* a pointer is not aligned to its element size                         */
  MQ_ERR_TOO_LARGE = -7    /* a size beyond the fixed capacity of an allocation-free entry point  -> ValueError */
} mq_status;

/* How to quantize.  group_size follows the reference: -1 per-tensor, any other value <= 0 per-channel
 * (one segment per row), > 0 per-group (quant_weight.py:241-255). */
typedef struct {
  int32_t format;     /* mq_format */
  int32_t bits;       /* INT formats: 2..8 */
  int32_t exp_bits;   /* LUT formats: the value table is +-(1+m/2^M)*2^(e-bias) over all exponent codes, */
  int32_t man_bits;   /*   no subnormals/inf/NaN, code (e=0,m=0) is the single zero:                      */
  int32_t exp_bias;   /*   FP8_E4M3 = (4,3,7) quant_weight.py:114-138; FP8_E3M4 = (3,4,3) :91-113        */
  int32_t f32_math;   /* 1 = quant_datatype's per-tensor float32 branch (quant_weight.py:562-575)        */
  int64_t group_size;
} mq_quant_cfg;

/* Where to put results; any pointer may be NULL to skip that output.  nseg = number of segments
 * (rows*cols/group, rows for per-channel, 1 for per-tensor). */
typedef struct {
  void* deq;             /* float16 [rows*cols]   dequantized weight (w_deq of _int_quant_meta/_fp_quant_meta) */
  void* codes;           /* INT: 1 byte/element, low byte of the integer code (int8 or uint8 view,
                            quant_weight.py:738,745,766,775); LUT: float16 table value (:870-872)  */
  void* scale;           /* float16 [nseg] */
  void* zero_point;      /* float16 [nseg], INT_ASYM only */
  void* abs_err;         /* |float16(w) - deq| per element, float32 or float16 [rows*cols]          */
  int32_t abs_err_dtype; /* MQ_F32: onnx_error_cli.py:90-91;  MQ_F16: single_weight_quant_error.py:39-40 */
  int32_t seg_mse_f32;   /* 0: seg_sse receives float64 sums; 1: seg_sse points to a float32 [nseg] array that
                            receives each segment's mean squared error, float32(sum * (1 / length)): the list
                            entry compute_mse returns, a float32 np.mean (quant_error_single_weight.py:46-48),
                            to 1e-6; 2: the same array, each entry summed in float32 in the order of numpy's
                            summation (pairwise, in chunks of 8192 elements) and divided as np.mean divides, so
                            that it has the BITS of np.mean(diff * diff) -- statistics-only calls (no deq / codes /
                            abs_err) on 16-byte-aligned float16 / bfloat16 weights with segments (groups or rows)
                            of a multiple of 8 and at most 32768 elements: groups of 32 / 64 / 128 elements in the
                            bandwidth kernels, every other length in a wave-per-segment kernel that is several
                            times slower; anything else returns MQ_ERR_UNSUPPORTED */
  double* seg_sse;       /* float64 [nseg] sum of squared float32 errors per segment (row form,
                            quant_error_single_weight.py:56-71 before the division by the length)  */
  double* totals;        /* float64 [2]: sum of squared errors, sum of absolute errors (float32 errors,
                            quant_error_multi_weight.py:55-58, onnx_error_cli.py:116-117)           */
} mq_quant_out;

int mq_version(void);
/*
 * Up to four small device buffers (multiples of 4 bytes) stored into page-locked host memory by ONE kernel on the
 * caller's stream: how the statistics of a sweep (a record table of tens of KiB, a percentile) reach the host
 * inside a step -- the place of the scripts' float(...) conversions (quant_error_multi_weight.py:57-58, :84-86).
 * dst_host[i]: host memory the device can address (hipHostMalloc, torch pinned memory), readable by the host once
 * the stream has been synchronised.  No counterpart in the reference (its arrays never leave the host).
 */
int mq_store_host(void* const* dst_host, const void* const* src_device, const size_t* bytes, int32_t count,
                  void* hip_stream);
const char* mq_strerror(int status);
/* Text of the last HIP error seen by this thread ("" if none). */
const char* mq_last_hip_error(void);

/*
 * Measurement hook (no counterpart in the reference): between mq_profile_begin and mq_profile_end every full
 * quantize launch made from the calling thread (the dominant kernel of mq_quantize / mq_quantize_batch /
 * mq_quantize_p99_batch; not the 1/32 sampling pass) is bracketed by a pair of HIP events on its stream.
 * mq_profile_end waits for them and returns the summed kernel time and the number of launches.  The events come
 * from a per-thread pool that outlives the run; mq_profile_reserve(launches) creates them ahead of time so that
 * no event is created between begin and end.
 */
int mq_profile_reserve(int64_t launches);
int mq_profile_begin(void);
int mq_profile_end(double* total_ms, int64_t* launches);

/*
 * Device self-check (no counterpart in the reference): the statistics-only kernels of the integer formats take the
 * float16 quotient v / scale in packed float16 arithmetic (a scaled reciprocal multiply and two fused corrections).
 * This enumerates every float16 scale in [2^-15, 0x70E2] against every finite float16 value with |v / scale| <= 600
 * on the device and compares with the float32 division of the other kernels (= torch's w / scale,
 * quant_weight.py:230-266).  counts (DEVICE, 3 x uint64, 8-byte aligned) receives: pairs checked, pairs whose
 * rounded integer differs, pairs with |v / scale| >= 2^-4 whose float16 bits differ; the last two must be 0.
 */
int mq_selfcheck_div16(uint64_t* counts, void* hip_stream);
/* The same for the division with a SHARED power of two: 2^(k - j), j = 0, 1 or 2, in place of the scale's own 2^k --
 * what the several-groupings kernel uses for a group inside a row whose scale is up to 4 x larger (one scaled copy of
 * the weights serves every grouping).  counts[2] then counts differing bits for |v / s| >= 2^(j - 4); counts[1] must
 * be 0 as above.  No counterpart in the reference. */
int mq_selfcheck_div16_shared(uint64_t* counts, int32_t j, void* hip_stream);

/*
 * Device self-check (no counterpart in the reference): does the hardware's packed float32 -> fp8 e4m3 conversion
 * (v_cvt_pk_fp8_f32, decoded again by v_cvt_f32_fp8) round like the reference's FP8_E4M3 value table -- nearest table
 * value, ties toward -inf, largest value 480, no subnormals (quant_weight.py:114-138, :577-601)?  Every finite
 * float16 x with |x| <= 480 is tried.  counts (DEVICE, 5 x uint64, 8-byte aligned) receives: values checked; values
 * where hardware and table differ; of those, exact ties between two table values; of those, |x| above 448 (the
 * hardware format's largest finite value); of those, |x| below 2^-6, the hardware format's smallest normal value
 * (it continues with subnormals of step 2^-9; the table keeps three mantissa bits down to 1.125 * 2^-7).  A non-zero second count is the reason the kernels round e4m3 in integer
 * bit arithmetic (lut_round_*), as they do e3m4, which has no hardware format.
 */
int mq_selfcheck_fp8_cvt(uint64_t* counts, void* hip_stream);

/* Bytes of device workspace mq_quantize needs for this shape (reduction slots + per-block partials). */
size_t mq_quantize_workspace_bytes(int64_t rows, int64_t cols, const mq_quant_cfg* cfg);

/*
 * One fused pass over a row-major [rows, cols] weight: cast to float16, per-segment absmax or
 * min/max, scale (and zero-point), quantize (round half to even), dequantize, error accumulation.
 * Replaces quant_int :230, quant_int_asym :299, quant_datatype :504, _int_quant_meta :718 and
 * _fp_quant_meta :779 of quant_weight.py, fused with the error arithmetic of
 * MSE&P99/quant_error_multi_weight.py:55-58 and elwise_error/onnx_error_cli.py:90-91,:116-117.
 */
int mq_quantize(const void* w, int32_t w_dtype, int64_t rows, int64_t cols, const mq_quant_cfg* cfg,
                const mq_quant_out* out, void* workspace, size_t workspace_bytes, void* hip_stream);

/* One tensor of a batch: its weight and where its results go. */
typedef struct {
  const void* w;
  mq_quant_out out;
} mq_quant_item;

size_t mq_quantize_batch_workspace_bytes(int32_t count, int64_t rows, int64_t cols, const mq_quant_cfg* cfg);

/*
 * mq_quantize for `count` tensors of one shape, dtype and configuration in one launch per 32 tensors
 * (the tensors' pointers travel in the kernel arguments; blockIdx.y selects the tensor).  The counterpart
 * of the reference's per-Linear Python loop (quant_weight.py:660-704: every layer of a model has the same
 * few shapes).  items is a HOST array; every item must request the same set of outputs.
 */
int mq_quantize_batch(const mq_quant_item* items, int32_t count, int32_t w_dtype, int64_t rows, int64_t cols,
                      const mq_quant_cfg* cfg, void* workspace, size_t workspace_bytes, void* hip_stream);

/*
 * One pass of a model sweep as ONE launch over tensors of DIFFERENT shapes (statistics only): the reference's loop
 * over every nn.Linear of a model (quant_weight.py:660-704) with the error arithmetic of
 * MSE&P99/quant_error_single_weight.py:46-79 fused in, for float16 / bfloat16 weights whose segments have one length:
 * groups of 32 .. 512 elements (a power of two; any shapes whose column counts the group divides) or segments of 520
 * .. 32768 elements, a multiple of 8 (per-channel: tensors with the same column count, any row counts; or longer
 * groups).  Every item requests seg_sse and
 * / or totals (and scale / zero_point if wanted) and none of deq / codes / abs_err.
 *
 * mq_sweep_prepare validates and writes the launch table -- mq_sweep_table_bytes(count) bytes, per tensor its
 * pointers, segment count and range of workgroups (one workgroup per 128 KiB of weights) -- into HOST memory; the
 * caller copies it to the device once (the weights and result buffers stay where they are between runs) and
 * mq_sweep_run enqueues the pass: a one-dimensional grid of total_blocks workgroups, each finding its tensor in the
 * table, then one launch that sums every tensor's per-workgroup partial sums in a fixed order.  workspace:
 * mq_sweep_workspace_bytes(total_blocks) bytes.  Results are those of mq_quantize_batch on each tensor (per-segment
 * values bit for bit; the float64 totals are summed over a different number of workgroups).
 * MQ_ERR_UNSUPPORTED: a shape, dtype, alignment or configuration outside the above -- use mq_quantize_batch.
 */
size_t mq_sweep_table_bytes(int32_t count);
size_t mq_sweep_workspace_bytes(int64_t total_blocks);
int mq_sweep_prepare(const mq_quant_item* items, const int64_t* rows, const int64_t* cols, int32_t count,
                     int32_t w_dtype, const mq_quant_cfg* cfg, void* table_host, size_t table_bytes,
                     int64_t* seg_len, int64_t* total_blocks);
int mq_sweep_run(const void* table_dev, int32_t count, int64_t seg_len, int64_t total_blocks, int32_t w_dtype,
                 const mq_quant_cfg* cfg, int32_t seg_mse_f32, void* workspace, size_t workspace_bytes,
                 void* hip_stream);
/*
 * mq_sweep_run for per-group passes WITHOUT writing the lists of large tensors: the percentile of such a list
 * (mse_p99 = np.percentile(vec, pct), quant_error_single_weight.py:79) is selected inside the pass.
 *   mq_sweep_feed_sample  evaluates one wave step of every `unit` steps of each fed tensor (at most 1 / 16 of its
 *                         weights, up to 32768 list entries) and takes two sample ranks, 12 binomial standard
 *                         deviations either side of the wanted quantile, as the tensor's bracket;
 *   mq_sweep_feed_run     is the pass: every wave counts its entries under the bracket and keeps the bit patterns of
 *                         those inside it (plain stores into its own slot; no shared counter); the other tensors'
 *                         lists are written as always;
 *   mq_sweep_feed_finish  adds the counts, packs the kept entries and selects the two wanted ranks among them ->
 *                         p99_out[table row] (float64, numpy's interpolation).  The counts are exact, so the result
 *                         is the percentile of the list the pass would have written, bit for bit.  A bracket that does
 *                         not hold both ranks, or a wave that kept more than it has room for, marks the tensor: the
 *                         repair launch (a few microseconds when nothing is marked) writes its list after all and
 *                         the percentile is selected from it.
 * Fed: tensors whose list has a quarter of a million entries or so (mq_sweep_feed_sizes tells how many of `count`,
 * and the bytes of feed_ws -- device, 256-byte aligned, one per pass -- and of aux).  mq_sweep_feed_prepare is
 * mq_sweep_prepare plus the feed: table_dev is the DEVICE address the caller will copy table_host to (the brackets
 * are written into it), aux_host receives the sampling launch's table and the selection jobs (copy it to the
 * device: aux_dev), fed_rows[f] the table row of the f-th fed tensor.  MQ_ERR_UNSUPPORTED when no tensor is fed.
 * Order: sample (any stream) -> run -> finish (any stream), each behind the one before.  No counterpart in the
 * reference, which holds every list in host memory.
 */
int mq_sweep_feed_sizes(const int64_t* rows, const int64_t* cols, int32_t count, const mq_quant_cfg* cfg, double pct,
                        int32_t* nfed, size_t* feed_ws_bytes, size_t* aux_bytes);
int mq_sweep_feed_prepare(const mq_quant_item* items, const int64_t* rows, const int64_t* cols, int32_t count,
                          int32_t w_dtype, const mq_quant_cfg* cfg, double pct, void* feed_ws_dev, size_t feed_ws_bytes,
                          void* table_dev, void* table_host, size_t table_bytes, void* aux_host, size_t aux_bytes,
                          int64_t* seg_len, int64_t* total_blocks, int64_t* sample_blocks, int32_t* nfed,
                          int32_t* fed_rows);
int mq_sweep_feed_sample(const void* aux_dev, int32_t nfed, int64_t seg_len, int64_t sample_blocks, int32_t w_dtype,
                         const mq_quant_cfg* cfg, void* feed_ws_dev, void* hip_stream);
int mq_sweep_feed_run(const void* table_dev, int32_t count, int64_t seg_len, int64_t total_blocks, int32_t w_dtype,
                      const mq_quant_cfg* cfg, void* workspace, size_t workspace_bytes, void* hip_stream);
int mq_sweep_feed_finish(const void* aux_dev, int32_t nfed, int64_t seg_len, int64_t sample_blocks, int32_t w_dtype,
                         const mq_quant_cfg* cfg, void* feed_ws_dev, double* p99_out, void* hip_stream);

/*
 * The same pass with every number of a report row in numpy's bits, at the speed of the pass above: per tensor
 *   the float32 per-vector list   per_vec[i] = compute_mse(slice)       quant_error_single_weight.py:56-71
 *   mse_tensor = float(np.mean((a32 - b32) ** 2)), float32              quant_error_multi_weight.py:55-58
 *   mse_rep    = float(vec.mean()), vec the float64 array of the list    quant_error_single_weight.py:74-78
 * (mse_p99 is mq_percentile_multi over the lists, dtype MQ_F32_AS_F64.)  numpy sums pairwise: chunks of 8192
 * elements added in order, each chunk split down to leaves of at most 128 elements, a leaf by eight strided
 * accumulators.  The launch runs the leaves' accumulator chains in that order while it streams the weights and writes
 * the list entries and the sums of the 128-element leaves of every FLATTENED tensor (leaf_sums, one float per 128
 * weights: the caller provides sum(rows[i] * cols[i]) / 128 floats, 256-byte aligned; tensor i's leaves follow tensor
 * i - 1's); mq_sweep_numpy_finish turns leaf sums and lists into mse_tensor[count] and mse_rep[count] (table order)
 * with a few small launches that may run on another stream once the pass has finished.
 * Taken: float16 / bfloat16 weights of a multiple of 8192 elements, groups of 32 / 64 / 128 elements, or segments
 * (rows, longer groups) of a multiple of 128 and at most 32768 elements whose chunks split into leaves of one depth
 * (every multiple of 128 that model shapes use).  Every item requests seg_sse (the float32 list, 16-byte aligned) and
 * totals, optionally scale / zero_point, none of deq / codes / abs_err.  MQ_ERR_UNSUPPORTED otherwise: use
 * mq_sweep_run / mq_quantize_batch with seg_mse_f32 = 2, mq_tensor_mse_numpy and mq_list_mean_numpy.
 * prepare writes mq_sweep_numpy_table_bytes(count) bytes of HOST memory (copy them to the device once) and reports
 * the launch's workgroups, the number of leaf sums and the number of whole 8192-entry chunks of the lists;
 * run: workspace of mq_sweep_workspace_bytes(total_blocks) bytes; finish: mq_sweep_numpy_finish_workspace_bytes.
 */
size_t mq_sweep_numpy_table_bytes(int32_t count);
int mq_sweep_numpy_prepare(const mq_quant_item* items, const int64_t* rows, const int64_t* cols, int32_t count,
                           int32_t w_dtype, const mq_quant_cfg* cfg, float* leaf_sums, void* table_host,
                           size_t table_bytes, int64_t* seg_len, int64_t* total_blocks, int64_t* total_leaves,
                           int64_t* list_chunks);
int mq_sweep_numpy_run(const void* table_dev, int32_t count, int64_t seg_len, int64_t total_blocks, int32_t w_dtype,
                       const mq_quant_cfg* cfg, void* workspace, size_t workspace_bytes, void* hip_stream);
/*
 * The launches of up to three numpy-order passes of ONE format whose group sizes are 32, 64 and 128, from one read of
 * the weights: table32 / 64 / 128_dev are the device copies of the tables mq_sweep_numpy_prepare wrote for those group
 * sizes over the SAME tensors in the same order (NULL: size not wanted; at least two), total_blocks what any of those
 * calls returned (it does not depend on the group size).  Every output -- list entries, leaf sums, scales,
 * zero-points, totals -- has the bits mq_sweep_numpy_run writes for that table; each pass is finished with its own
 * mq_sweep_numpy_finish as before.  (sweep_quant_error_to_csv.py's --group-sizes loop, one weight read instead of
 * one per size.)  Workspace: mq_sweep_numpy_groups_workspace_bytes(total_blocks).
 */
size_t mq_sweep_numpy_groups_workspace_bytes(int64_t total_blocks);
int mq_sweep_numpy_run_groups(const void* table32_dev, const void* table64_dev, const void* table128_dev, int32_t count,
                              int64_t total_blocks, int32_t w_dtype, const mq_quant_cfg* cfg, void* workspace,
                              size_t workspace_bytes, void* hip_stream);
size_t mq_sweep_numpy_finish_workspace_bytes(int64_t total_leaves, int64_t list_chunks);
int mq_sweep_numpy_finish(const void* table_dev, int32_t count, int64_t seg_len, const float* leaf_sums,
                          int64_t total_leaves, int64_t list_chunks, float* mse_tensor, double* mse_rep,
                          void* workspace, size_t workspace_bytes, void* hip_stream);
/*
 * mse_tensor in numpy's bits: mse_out[0] = float32 np.mean((a32 - b32) ** 2) over the whole [rows, cols] tensor
 * (compute_mse, quant_error_multi_weight.py:55-58), a32 the float16-rounded weight and b32 its dequantized value:
 * the float32 squares summed in numpy's order over the FLATTENED tensor -- chunks of 8192 elements added in order,
 * each chunk pairwise down to leaves of at most 128 elements -- and divided by the count in float64.  The chunks
 * cross rows and groups, so the segments' parameters are inputs: scale (and zero_point for MQ_INT_ASYM), float16
 * [nseg], as a statistics pass of mq_quantize / mq_sweep_run wrote them for this weight and configuration.
 * float16 / bfloat16 weights, 16-byte aligned, segment length a multiple of 8 (MQ_ERR_UNSUPPORTED otherwise).  workspace: mq_tensor_mse_numpy_workspace_bytes(rows, cols).  Two launches; a
 * correctness mode for report.*(numpy_order=True) -- every element takes the per-element routine -- not a bandwidth
 * path.
 */
size_t mq_tensor_mse_numpy_workspace_bytes(int64_t rows, int64_t cols);
int mq_tensor_mse_numpy(const void* w, int32_t w_dtype, int64_t rows, int64_t cols, const mq_quant_cfg* cfg,
                        const void* scale, const void* zero_point, float* mse_out, void* workspace,
                        size_t workspace_bytes, void* hip_stream);

/*
 * mse_rep in numpy's bits: out[i][0] = np.array(list_i, dtype=float64).mean() for `count` float32 lists of n values
 * each -- vec = np.array(per_vec); float(vec.mean()) of quant_error_single_weight.py:74-78, whose entries are
 * float32 means held in a float64 array -- summed in float64 in numpy's order (chunks of 8192 elements added in
 * order, each chunk pairwise down to leaves of at most 128 elements with eight strided accumulators) and divided by
 * n.  lists / out: HOST arrays of device pointers (float32 [n], 4-byte aligned; one float64 each).  One workgroup
 * per list, 64 lists per launch: a correctness mode for report.*(numpy_order=True), not a bandwidth path.
 */
int mq_list_mean_numpy(const void* const* lists, int32_t count, int64_t n, double* const* out, void* hip_stream);

/*
 * np.mean of slices of a float32 array in numpy's bits (float32 summation in numpy's order, the sum divided by the
 * count in float64 and rounded to float32): compute_mse on the slices the analysis scripts take from the squared-
 * error array -- a column a[:, j] or a block of gs columns a[:, g*gs:(g+1)*gs] (quant_error_multi_weight.py:67-78),
 * a row or a run of a row (quant_error_single_weight.py:56-71), the whole array (:55-58).  Slice i holds n values:
 * value t is firsts[i][(t / run) * stride + t % run] -- runs of `run` contiguous values, `stride` elements apart, the
 * order in which numpy walks the slice (a column: run 1, stride = columns; a block: run gs; contiguous: run = n).
 * firsts / out: HOST arrays of device pointers (float32; one float32 result each).  One workgroup per slice, 64
 * slices per launch: a correctness mode for error_metrics.compute_errors(numpy_order=True).
 */
int mq_slice_mean_numpy_f32(const void* const* firsts, int32_t count, int64_t n, int64_t run, int64_t stride,
                            float* const* out, void* hip_stream);
/* The same slices' float32 SUMS in numpy's order, no division: np.sum(abs_err) of onnx_error_cli.py:116-117,
 * :125-127. */
int mq_slice_sum_numpy_f32(const void* const* firsts, int32_t count, int64_t n, int64_t run, int64_t stride,
                           float* const* out, void* hip_stream);

/*
 * The report rows of one pass (the CSV rows of quant_error_multi_weight.py:142-151 before formatting): for tensor i
 * of `count`, records[perm[i]][0..5] = numel[i], totals[i][0] (sum of squared errors), totals[i][1] (sum of |error|),
 * totals[i][0] / numel[i] (mse_tensor = mse_rep, IEEE float64 division), p99[i] (mse_p99; NaN when the mean is NaN,
 * as np.percentile of a list holding a NaN), 0.  All arrays are device arrays; one small launch.
 */
int mq_sweep_records(const double* totals, const double* p99, const double* numel, const int64_t* perm, int32_t count,
                     double* records, void* hip_stream);
/*
 * The same rows for a whole report table in one launch, gathered instead of scattered: row i of `table` (rows x 6
 * doubles, EVERY row is written) is the record of source s = src[i] -- numel[s], totals[s][0], totals[s][1],
 * totals[s][0] / numel[s], p99[p99_index[s]] (NaN when the mean is NaN), 0 -- or six zeros where src[i] < 0: the rows
 * of layers another GPU holds, which the all-reduce over the ranks' tables fills in (quant_error_multi_weight.py:142-151
 * run over a model whose layers are sharded).  src NULL: s = i; p99_index NULL: p99[s].  All arrays are device arrays.
 */
int mq_sweep_table(const double* totals, const double* p99, const int64_t* p99_index, const double* numel,
                   const int64_t* src, int32_t rows, double* table, void* hip_stream);

/*
 * Several groupings of ONE format from one read of the weights: the reference's sweeps quantize the same tensor once
 * per (format, grouping) -- per row, groups of 32 / 64 / 128 (MSE&P99/quant_error_single_weight.py:56-71 per
 * configuration file; the configurations of onnx_error_cli.py) -- and this launch computes up to four of them while a
 * row is in registers.  Per tensor and grouping slot c (0: rows, 1: groups of 32, 2: 64, 3: 128; `mask` bit c set):
 * list[c], the float32 per-segment mean squared errors (rows x cols / L entries, as seg_sse with seg_mse_f32 = 1),
 * and totals[c] = (sum err^2, sum |err|) in float64 -- the outputs of mq_sweep_run for that grouping, to its
 * tolerance (the sums are taken in another order; scales, codes and errors are the same arithmetic).  Statistics
 * only.  All tensors of a launch have `cols` columns (a multiple of 128, at most 32768), float16 / bfloat16, 16-byte
 * aligned; cfg->group_size is ignored.  prepare fills the HOST table (mq_sweep_multi_table_bytes(count) bytes) that
 * the caller copies to the device; run launches over the device copy (workspace:
 * mq_sweep_multi_workspace_bytes(total_blocks), 16-byte aligned).
 */
typedef struct mq_multi_item {
  const void* w;
  void* list[4];     /* float32 lists per slot (NULL where the mask bit is clear) */
  double* totals[4]; /* two doubles per slot */
} mq_multi_item;
size_t mq_sweep_multi_table_bytes(int32_t count);
size_t mq_sweep_multi_workspace_bytes(int64_t total_blocks);
int mq_sweep_multi_prepare(const mq_multi_item* items, const int64_t* rows, int64_t cols, int32_t count, int32_t w_dtype,
                           const mq_quant_cfg* cfg, uint32_t mask, void* table_host, size_t table_bytes,
                           int64_t* total_blocks);
int mq_sweep_multi_run(const void* table_dev, int32_t count, int64_t cols, int64_t total_blocks, int32_t w_dtype,
                       const mq_quant_cfg* cfg, uint32_t mask, void* workspace, size_t workspace_bytes, void* hip_stream);
/* The two halves of mq_sweep_multi_run, for callers that keep the small totals launch off the stream of the statistics
 * launches (then every launch in flight needs a workspace of its own): _launch enqueues the statistics kernel only,
 * _finalize adds its workgroups' partial sums into the tensors' totals, on any stream ordered behind that launch. */
int mq_sweep_multi_launch(const void* table_dev, int32_t count, int64_t cols, int64_t total_blocks, int32_t w_dtype,
                          const mq_quant_cfg* cfg, uint32_t mask, void* workspace, size_t workspace_bytes,
                          void* hip_stream);
int mq_sweep_multi_finalize(const void* table_dev, int32_t count, int64_t total_blocks, void* workspace,
                            size_t workspace_bytes, void* hip_stream);

/*
 * mq_quantize_batch plus p99[i] = np.percentile(abs_err_i, pct) (float32, method 'linear': the P99 of
 * elwise_error/onnx_error_cli.py:113-114) for every tensor, with the selection fed from the quantize pass
 * itself: a sampling pass over 1/32 of the weight brackets the wanted ranks, the full pass counts the errors
 * below the bracket and histograms those inside it while they are in registers, so the error array is not
 * read back (the two-pass selection over it runs only when the bracket misses).  This holds for segments of
 * up to 512 elements (groups); per-channel / per-tensor shapes, small tensors and quantiles at the very ends
 * quantize first and select from the stored error array.  Every item must request
 * the float32 abs_err array (16-byte aligned).  p99: device array of `count` floats.
 */
size_t mq_quantize_p99_batch_workspace_bytes(int32_t count, int64_t rows, int64_t cols, const mq_quant_cfg* cfg);
int mq_quantize_p99_batch(const mq_quant_item* items, int32_t count, int32_t w_dtype, int64_t rows, int64_t cols,
                          const mq_quant_cfg* cfg, double pct, float* p99, void* workspace, size_t workspace_bytes,
                          void* hip_stream);

/*
 * Several mq_quantize_p99_batch passes over stacks of one shape -- the configurations of a sweep, e.g. the group
 * sizes the reference's scripts loop over (elwise_error/onnx_error_cli.py:96-120 per configuration) -- enqueued
 * together: pass k has its own items (outputs), configuration and results.  The results are those of npass separate
 * mq_quantize_p99_batch calls; the difference is where the short kernels run: the chain in front of a full pass
 * (sampling pass, sample histogram, bracket) and the one behind it (bracket scan, candidates, the guarded exact
 * route) go to two streams the library owns, beside the neighbouring full passes, and the caller's stream carries the
 * full passes back to back.  list_p99 (optional, device array of `count` doubles per pass): np.percentile(list_i,
 * list_pct) of every item's float32 per-segment MSE list (items must then request seg_sse with seg_mse_f32 = 1), in
 * float64 as mq_percentile_multi with MQ_F32_AS_F64, selected behind the pass as well.  The caller's stream waits for
 * both library streams before the call returns, so everything enqueued on it afterwards sees every output.
 * The kernels that need a CU to themselves (the guarded exact route, the list percentiles) run after the last full
 * pass; when the p99 (and list_p99) arrays of consecutive passes follow each other in memory and count <= 32 they
 * go through one set of launches for up to 64 arrays instead of one per pass.
 * At most 24 (pass, 32-item chunk) pairs per call (MQ_ERR_UNSUPPORTED beyond).  workspace: 256-byte aligned.
 * Calls on one device are serialised while they enqueue (the library streams and their events are shared).
 * No counterpart in the reference (it runs one configuration after the other on the CPU).
 */
typedef struct mq_p99_pass {
  const mq_quant_item* items; /* [count] */
  const mq_quant_cfg* cfg;
  float* p99;                 /* device [count] */
  double* list_p99;           /* device [count], or NULL */
} mq_p99_pass;
size_t mq_quantize_p99_passes_workspace_bytes(int32_t npass, int32_t count, int64_t rows, int64_t cols);
int mq_quantize_p99_passes(const mq_p99_pass* passes, int32_t npass, int32_t count, int32_t w_dtype, int64_t rows,
                           int64_t cols, double pct, double list_pct, void* workspace, size_t workspace_bytes,
                           void* hip_stream);

/*
 * Value tables given as explicit lists: the reference's formats whose values have no (E, M, bias) closed form
 * (FP3 quant_weight.py:11, FLINT4 :20, FLINT5 :36, the FP3/FP4 *_er_* / *_ea_* lists :12-15, :24-31), FP8_E5M2
 * :139-169, whose path runs in float32 (:505-540, :796-826), and the MX path (quant_mx :337-380).
 * The chosen entry is the one with mid[i-1] < x <= mid[i], mid[i] = (values[i] + values[i+1]) / 2 (a NaN x
 * gives 0, :590-595); per-segment results are as for mq_quantize.
 */
typedef enum {
  MQ_TABLE_F16 = 0, /* float16 op by op, as MQ_LUT_FP: quant_datatype :577-601, _fp_quant_meta :829-872;
                       deq / codes / scale are float16 */
  MQ_TABLE_F32 = 1, /* float32: scale = clamp(max|w32| / qmax, 1e-5, 1e4), x = w32 / scale, deq = q * scale.
                       quant_datatype's per-tensor branch :562-575 (deq_dtype MQ_F16: the product is rounded to
                       float16) and the FP8_E5M2 path :505-540, :796-826 (deq_dtype MQ_F32); codes and scale
                       are float32 */
  MQ_TABLE_MX = 2   /* quant_mx :337-380: per group e = floor(log2(max|w32|)), x = (w32 / 2^e) / (2 / qmax),
                       deq = float16((q * (2 / qmax)) * 2^e); group_size must be > 0; only deq and the error
                       outputs are produced (codes, scale and zero_point must be NULL) */
} mq_table_math;

#define MQ_TABLE_MAX_VALUES 256
#define MQ_TABLE_NEG_INF_ZERO 1 /* flags: x = -inf selects 0, not values[0]: the (edge[i] < x) test of the
                                   FP8_E5M2 loop (quant_weight.py:535-538) has no entry for it */
typedef struct {
  int32_t math;       /* mq_table_math */
  int32_t count;      /* number of values, 2 .. MQ_TABLE_MAX_VALUES, strictly ascending */
  int32_t deq_dtype;  /* MQ_F16, or MQ_F32 (MQ_TABLE_F32 only) */
  int32_t flags;      /* MQ_TABLE_NEG_INF_ZERO or 0 */
  int64_t group_size; /* as mq_quant_cfg */
  float values[MQ_TABLE_MAX_VALUES];
} mq_table_cfg;

size_t mq_quantize_table_workspace_bytes(int64_t rows, int64_t cols, const mq_table_cfg* cfg);
/* out->abs_err / seg_sse / totals: errors float32(float16(w)) - float32(deq), as for mq_quantize.  These formats
 * are off the bandwidth-critical path: one wave (or workgroup) per segment, scalar element accesses. */
int mq_quantize_table(const void* w, int32_t w_dtype, int64_t rows, int64_t cols, const mq_table_cfg* cfg,
                      const mq_quant_out* out, void* workspace, size_t workspace_bytes, void* hip_stream);

/*
 * search_datatype (quant_weight.py:604-649): every segment (row, or group of group_size elements; -1 means rows
 * here too, :631-632) of a float16 weight is quantized with each of ntables <= 8 value lists of at most 32
 * entries (MQ_TABLE_F16 arithmetic; tables[i].group_size / math / deq_dtype are ignored) and keeps the first
 * list whose error h(h(sum_f32 h(h(d - v)^2)) ... / L) -- the float16 (tmp - w).pow(2).mean(-1) -- is strictly
 * below the best so far, starting from 1000; a segment no list improves stays 0.  deq: float16 [rows*cols];
 * choice (optional): uint8 [nseg], the index of the chosen list or 255.  The float16 squares are multiples of
 * 2^-24, so while a segment's sum stays below 1 every float32 summation order gives the same, exact sum: there
 * the result equals torch's bit for bit; beyond it torch's own result depends on the host's vector width.
 * No workspace is needed.
 */
int mq_search_tables(const void* w_f16, int64_t rows, int64_t cols, int64_t group_size, const mq_table_cfg* tables,
                     int32_t ntables, void* deq, uint8_t* choice, void* hip_stream);

/* Bytes of device workspace mq_percentile needs (histograms + selection state). */
size_t mq_percentile_workspace_bytes(void);

/*
 * np.percentile(x, pct) with the default 'linear' method for a NON-NEGATIVE float32 (MQ_F32) or float64
 * (MQ_F64) array of n values in device memory (the sign bit is ignored), computed exactly by radix
 * select on the float bit patterns: 2 passes over a float32 array, 4 over a float64 one; float32 arrays of
 * 2^21 values and more are first tried with a 1/32 sample + ONE counting pass over a bracket around the
 * wanted ranks (exact whenever both ranks fall inside, otherwise the two-pass route runs).  Replaces
 *   np.percentile(abs_err, 99)  elwise_error/onnx_error_cli.py:113-114   (float32, K*C values)
 *   np.percentile(vec, 99)      MSE&P99/quant_error_multi_weight.py:85, quant_error_single_weight.py:79
 *                               (float64 per-vector MSE list)
 * The virtual index, its floor and the interpolation follow numpy 2.2 in the array's dtype, so the result
 * has the same bits as numpy's.  out (one value) and out_pair (the two order statistics interpolated
 * between) are device pointers of the array's dtype (float64 for MQ_F32_AS_F64); either may be NULL.  x must
 * be 16-byte aligned.  Every percentile / select function below takes the same three dtype codes.
 */
int mq_percentile(const void* x, int32_t dtype, int64_t n, double pct, void* out, void* out_pair,
                  void* workspace, size_t workspace_bytes, void* hip_stream);

/*
 * mq_percentile for `count` arrays of n values each (one dtype): out[i] (and out_pair[2 i], [2 i + 1])
 * receive array i's result.  One set of launches per 64 arrays (blockIdx.y = array), each array with its
 * own slice of the workspace.  xs is a HOST array of device pointers, each 16-byte aligned.
 */
size_t mq_percentile_batch_workspace_bytes(int32_t count);
/* The same for arrays of different lengths ns[i] (HOST array); arrays travel 64 per set of launches, the
 * workspace needs mq_percentile_batch_workspace_bytes(min(count, 64)). */
int mq_percentile_multi(const void* const* xs, const int64_t* ns, int32_t count, int32_t dtype, double pct, void* out,
                        void* out_pair, void* workspace, size_t workspace_bytes, void* hip_stream);
/* mq_percentile_multi for a call enqueued on a stream of its own BESIDE a kernel that fills the chip (the
 * selections of one chunk of a model sweep beside the statistics launch of the next): identical results, from
 * 256-thread workgroups where mq_percentile_multi launches 1024-thread ones.  A 1024-thread workgroup needs four
 * free wave slots on every SIMD of a compute unit at once and waits until the neighbour drains a whole unit; single
 * waves find room as the neighbour's waves retire.  Alone on the chip mq_percentile_multi is the faster one.  No
 * counterpart in the reference (its np.percentile calls run one after the other on the host). */
int mq_percentile_multi_beside(const void* const* xs, const int64_t* ns, int32_t count, int32_t dtype, double pct,
                               void* out, void* out_pair, void* workspace, size_t workspace_bytes, void* hip_stream);
int mq_percentile_batch(const void* const* xs, int32_t count, int32_t dtype, int64_t n, double pct, void* out,
                        void* out_pair, void* workspace, size_t workspace_bytes, void* hip_stream);

/*
 * The steps mq_percentile is made of, for selecting jointly from several arrays (a model-wide P99 over all
 * layers) or from arrays on several GPUs (merge that level's histogram with an all-reduce SUM of uint32
 * between accumulate and scan; every rank then runs the same scan).  No counterpart in the reference, which
 * holds one array in host memory.  Order: begin -> for level = 0 .. last (1 for float32, 3 for float64):
 * accumulate(level, each array) -> [merge the area mq_select_layout(level) names] -> scan(level), the last
 * scan with finish = 1.  The ranks k0 <= k1 and gamma come from mq_percentile_index on the TOTAL count.
 */
void mq_select_layout(int32_t level, size_t* offset, size_t* count);
int mq_percentile_index(int32_t dtype, int64_t n, double pct, int64_t* k0, int64_t* k1, double* gamma);
int mq_select_begin(void* workspace, size_t workspace_bytes, void* hip_stream);
int mq_select_accumulate(const void* x, int32_t dtype, int64_t n, int32_t level, void* workspace,
                         size_t workspace_bytes, void* hip_stream);
/* mq_select_accumulate for `count` arrays (xs, ns: HOST arrays of device pointers / lengths), 64 per launch. */
int mq_select_accumulate_multi(const void* const* xs, const int64_t* ns, int32_t count, int32_t dtype, int32_t level,
                               void* workspace, size_t workspace_bytes, void* hip_stream);
int mq_select_scan(int32_t level, int64_t k0, int64_t k1, int32_t finish, int32_t dtype, double gamma, void* out,
                   void* out_pair, void* workspace, size_t workspace_bytes, void* hip_stream);

/* Bytes of device workspace mq_error_stats needs (want_cols: column sums requested). */
size_t mq_error_workspace_bytes(int64_t rows, int64_t cols, int32_t want_cols);

/*
 * Error statistics between two stored row-major [rows, cols] tensors a (baseline) and b (quantized), both
 * float16 or both float32, as the analysis scripts compute them from the two ONNX initializers:
 *   totals[0] = sum (a32-b32)^2, totals[1] = sum |a32-b32|   compute_mse, quant_error_multi_weight.py:55-58;
 *                                                           get_error_sum, onnx_error_cli.py:116-117
 *   seg_sse[s]  row-form sums over consecutive runs of seg_len elements (rows, or (row, group) pairs in
 *               row-major order)                            quant_error_single_weight.py:56-71
 *   col_sse[c]  per-column sums over all rows (column blocks are sums of these)
 *                                                           quant_error_multi_weight.py:67-78
 *   abs_err     |a32 - b32| as float32 (onnx_error_cli.py:90-91) or, for float16 inputs, the float16
 *               |h(a - b)| of single_weight_quant_error.py:39-40 and compare.py:104
 * Divide by the vector length on the host side to get the MSEs.  Any output may be NULL.
 * Sums are added in a fixed order: a call form (the set of outputs requested on one shape) gives the same bits run
 * after run.  Different call forms may add in different orders -- with col_sse the totals come from the column
 * pass, without it from the row pass -- so totals agree ACROSS forms to the 1e-6 relative bar, not bit for bit.
 */
int mq_error_stats(const void* a, const void* b, int32_t dtype, int64_t rows, int64_t cols, int64_t seg_len,
                   void* abs_err, int32_t abs_err_dtype, double* seg_sse, double* col_sse, double* totals,
                   void* workspace, size_t workspace_bytes, void* hip_stream);

/*
 * mq_error_stats without a caller workspace and without any allocation.  It has no counterpart in the reference;
 * it is for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  The partial sums mq_error_stats keeps in its workspace live in a device
 * array that is part of the library's code object instead: 4.25 MiB, held by every GPU the library runs on until
 * the process ends, enough for col_sse over at most MQ_ERROR_STATS_STATIC_MAX_COLS columns (any number of rows;
 * without col_sse any shape).  More columns return MQ_ERR_TOO_LARGE before anything is enqueued.  The call makes
 * no hipMalloc / hipFree and calls no allocating routine; the array's address is looked up once per device.  All
 * calls on one GPU share the array: calls that overlap there (different streams, different threads) race; calls
 * on one stream are ordered as usual.  The GPU that holds the tensors (and owns the stream) must be the calling
 * thread's current device -- the array used is the current device's copy; a tensor of another GPU returns
 * MQ_ERR_ARG.  Arguments and results are those of mq_error_stats.
 */
#define MQ_ERROR_STATS_STATIC_MAX_COLS 8192
int mq_error_stats_static(const void* a, const void* b, int32_t dtype, int64_t rows, int64_t cols, int64_t seg_len,
                          void* abs_err, int32_t abs_err_dtype, double* seg_sse, double* col_sse, double* totals,
                          void* hip_stream);

/*
 * idx[i] = min(np.digitize(abs_err[i], bins, right=True), nbins - 1) for a float16 |error| array
 * (elwise_error/single_weight_quant_error.py:43-51).  bins: host array of nbins <= 64 ascending float64.
 */
int mq_bucketize_f16(const void* abs_err_f16, int64_t n, const double* bins_host, int32_t nbins, uint8_t* idx,
                     void* hip_stream);

/*
 * elwise_error/compare.py:99-125 for nfmt <= 7 quantized float16 tensors q_f16[f] of one float16 baseline:
 * e_f = |h(w - q_f)|, m = min_f e_f, flags[i] bit 7 = (m > threshold), bit f = (e_f == m) and not bit 7.
 * threshold_f16_bits: the threshold rounded to float16 (numpy compares a float16 array with a Python
 * float in float16).  min_err_f16 (optional) receives m.  q_f16 is a HOST array of device pointers.
 */
int mq_compare_formats_f16(const void* w_f16, const void* const* q_f16, int32_t nfmt, int64_t n,
                           uint16_t threshold_f16_bits, uint8_t* flags, void* min_err_f16, void* hip_stream);

/*
 * The scans behind the reference's histograms (np.histogram / plt.hist): the 200-bin weight histogram over a
 * global range found in a first pass (plot_matmul_weight_rev0.py:51-101, :221-238), the 200-bin |error|
 * histograms (elwise_error/onnx_error_cli.py:208-215, MSE&P99/quant_error_single_weight.py:81) and the 50-bin
 * per-vector MSE histograms (MSE&P99/plot.py:81-90).  x: device array of n float16 / float32 / float64 values.
 *
 * mq_minmax: out[0] = min, out[1] = max as float64 (device; NaN when x holds a NaN, like np.min / np.max);
 * with merge != 0 the previous contents of out are folded in (a running range over several arrays).
 * mq_histogram: counts[i] += #{ edges[i] <= v < edges[i+1] }, the last bin closed on the right, values outside
 * [edges[0], edges[nbins]] dropped -- np.histogram's rule; with numpy's own edges (np.histogram_bin_edges /
 * np.linspace, computed by the caller) the counts equal numpy's.  edges: DEVICE array of nbins + 1 ascending
 * float64 values, nbins <= 4096; counts: device int64 [nbins], accumulated into (zero it first; several arrays
 * may be added into one histogram).
 */
size_t mq_minmax_workspace_bytes(void);
int mq_minmax(const void* x, int32_t dtype, int64_t n, int32_t merge, double* out, void* workspace,
              size_t workspace_bytes, void* hip_stream);
int mq_histogram(const void* x, int32_t dtype, int64_t n, const double* edges, int32_t nbins, int64_t* counts,
                 void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* MQ_H_ */
