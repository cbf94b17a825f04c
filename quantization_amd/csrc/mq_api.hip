// Version, status strings and per-thread HIP error text of libmq.
#include <stdio.h>
#include <string.h>

#include "mq_common.h"

static thread_local char g_hip_err[256] = "";

void mq_set_hip_error(hipError_t e, const char* what) {
  snprintf(g_hip_err, sizeof g_hip_err, "%s: %s", what, hipGetErrorString(e));
}

extern "C" int mq_version(void) { return 100; }

extern "C" const char* mq_last_hip_error(void) { return g_hip_err; }

extern "C" const char* mq_strerror(int status) {
  switch (status) {
    case MQ_OK: return "ok";
    case MQ_ERR_ARG: return "invalid argument";
    case MQ_ERR_GROUP: return "cols is not a multiple of group_size";
    case MQ_ERR_UNSUPPORTED: return "unsupported bit width, format or shape";
    case MQ_ERR_WORKSPACE: return "workspace missing or too small";
    case MQ_ERR_HIP: return "HIP runtime error";
    case MQ_ERR_ALIGN: return "pointer not aligned to its element size";
    case MQ_ERR_TOO_LARGE: return "size beyond the fixed capacity of this allocation-free entry point";
    default: return "unknown status";
  }
}

// The statistics of a sweep are a few KiB: a kernel stores them straight into page-locked host memory (mapped into
// the device's address space) on the caller's stream -- no hand-over to a copy engine, whose packets cost tens of
// microseconds each at the end of a step.  The values have arrived when the stream has been synchronised.
struct StoreParts {
  void* dst[4];
  const void* src[4];
  uint64_t bytes[4];
  int count;
};
__global__ __launch_bounds__(256) void k_store_host(const StoreParts p) {
  for (int i = 0; i < p.count; ++i) {
    const uint32_t* s = static_cast<const uint32_t*>(p.src[i]);
    uint32_t* d = static_cast<uint32_t*>(p.dst[i]);
    const uint64_t words = p.bytes[i] / 4;
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < words; j += (uint64_t)gridDim.x * 256) d[j] = s[j];
  }
  __threadfence_system();
}
extern "C" int mq_store_host(void* const* dst_host, const void* const* src_device, const size_t* bytes, int32_t count,
                             void* hip_stream) {
  if (count < 0 || count > 4 || (count > 0 && (!dst_host || !src_device || !bytes))) return MQ_ERR_ARG;
  StoreParts p = {};
  size_t most = 0;
  for (int i = 0; i < count; ++i) {
    if (!dst_host[i] || !src_device[i]) return MQ_ERR_ARG;
    if (bytes[i] % 4 || reinterpret_cast<uintptr_t>(dst_host[i]) % 4 || reinterpret_cast<uintptr_t>(src_device[i]) % 4)
      return MQ_ERR_ALIGN;
    p.dst[i] = dst_host[i]; p.src[i] = src_device[i]; p.bytes[i] = bytes[i];
    most = bytes[i] > most ? bytes[i] : most;
  }
  p.count = count;
  if (most == 0) return MQ_OK;
  const size_t blocks = (most / 4 + 255) / 256;
  hipLaunchKernelGGL(k_store_host, dim3((unsigned)(blocks < 64 ? blocks : 64)), dim3(256), 0,
                     static_cast<hipStream_t>(hip_stream), p);
  MQ_HIP(hipGetLastError());
  return MQ_OK;
}
