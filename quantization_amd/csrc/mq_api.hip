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
