// Opens every file named on the command line with the ONNX reader and walks all initializers, touching the
// first and last byte of each tensor.  Built with -fsanitize=address,undefined by tests/test_onnx_io.py and
// fed truncated / bit-flipped copies of a valid model: the reader must fail cleanly or succeed, never fault.
#include <cstdio>

#include "mq_onnx.h"

int main(int argc, char** argv) {
  unsigned long sink = 0;
  int opened = 0;
  for (int a = 1; a < argc; ++a) {
    mq_onnx_model* m = nullptr;
    if (mq_onnx_open(argv[a], &m) != 0) continue;
    ++opened;
    for (long i = 0; i < mq_onnx_num_initializers(m); ++i) {
      mq_onnx_tensor t;
      if (mq_onnx_initializer(m, i, &t) != 0) continue;
      if (t.data && t.nbytes) sink += ((const unsigned char*)t.data)[0] + ((const unsigned char*)t.data)[t.nbytes - 1];
      sink += (unsigned long)mq_onnx_find(m, t.name);
    }
    mq_onnx_close(m);
  }
  printf("opened %d of %d (%lu)\n", opened, argc - 1, sink);
  return 0;
}
