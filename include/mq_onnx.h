/*
 * mq_onnx.h -- reading weight initializers out of ONNX files without the `onnx` package (host code, part of
 * libmq.so).  The reference's analysis scripts get their two input tensors this way:
 *   onnx.load + {init.name: init}                 MSE&P99/quant_error_multi_weight.py:30-32
 *   np.frombuffer(init.raw_data, dtype).reshape   MSE&P99/quant_error_multi_weight.py:35-38 (proto_to_array)
 *   MatMul / Gemm weights, through one Transpose  MSE&P99/quant_error_multi_weight.py:90-116 (find_compute_weights)
 * and the files are written by export_and_quant.py:107-126 (one external-data file next to model.onnx for
 * tensors of 1024 bytes and more).
 *
 * The model file (and every external-data file it names) is memory-mapped; tensor data pointers point into
 * those mappings and stay valid until mq_onnx_close.  Nothing is copied, nothing touches the GPU.
 */
#ifndef MQ_ONNX_H_
#define MQ_ONNX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mq_onnx_model mq_onnx_model;

/* onnx.TensorProto.DataType values of the types the scripts map (quant_error_multi_weight.py:16-27) */
enum {
  MQ_ONNX_FLOAT = 1, MQ_ONNX_UINT8 = 2, MQ_ONNX_INT8 = 3, MQ_ONNX_INT32 = 6, MQ_ONNX_INT64 = 7,
  MQ_ONNX_FLOAT16 = 10, MQ_ONNX_DOUBLE = 11, MQ_ONNX_BFLOAT16 = 16
};

typedef struct {
  const char* name;       /* NUL-terminated, owned by the model */
  int32_t data_type;      /* TensorProto.data_type */
  int32_t ndims;
  int64_t dims[8];
  const void* data;       /* raw_data bytes (in the model file or in its external-data file); NULL when the
                             tensor keeps its values in typed fields (float_data ...), which the scripts'
                             proto_to_array cannot read either */
  uint64_t nbytes;
  int32_t external;       /* 1: data lives in an external-data file */
  int32_t compute_weight; /* 1: second input of a MatMul / Gemm node, directly or through one Transpose */
} mq_onnx_tensor;

/* 0 on success; on failure *out is NULL and mq_onnx_error() describes the problem. */
int mq_onnx_open(const char* model_path, mq_onnx_model** out);
void mq_onnx_close(mq_onnx_model* model);
int64_t mq_onnx_num_initializers(const mq_onnx_model* model);
/* Initializers in file order (the order of model.graph.initializer). */
int mq_onnx_initializer(const mq_onnx_model* model, int64_t index, mq_onnx_tensor* out);
/* Index of the initializer called `name`, or -1. */
int64_t mq_onnx_find(const mq_onnx_model* model, const char* name);
const char* mq_onnx_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MQ_ONNX_H_ */
