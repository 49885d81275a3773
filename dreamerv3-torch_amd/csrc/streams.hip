// Compute-unit partitioned HIP streams ("lanes") for libdv3hip.
//
// The reverse observe scan is a chain of ~320 dependent few-row launches that runs at full speed on half of the chip's
// CUs; the weight gradients of the decoder / heads / prior head are chip-filling launches nothing waits for until the
// optimizer.  Two streams with COMPLEMENTARY compute-unit masks run the two side by side without touching each other
// (tools/cumask_probe.py: the chain 7.7 us per launch alone, 7.7 us beside 4096^3 GEMMs on the other mask; beside the
// same GEMMs on an unmasked second queue it makes no progress at all).  The mask is a property of the hardware queue;
// a hipGraph launched on such a stream inherits it.
#include "dv3_common.h"
#include "dv3hip.h"

extern "C" int dv3_device_cu_count(int* count_out) {
  if (!count_out) return DV3_ERR_ARG;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) return (int)e;
  *count_out = prop.multiProcessorCount;
  return DV3_OK;
}

extern "C" int dv3_stream_create_cu_masked(int n_words, const unsigned int* mask_words, unsigned long long* stream_out) {
  if (n_words <= 0 || !mask_words || !stream_out) return DV3_ERR_ARG;
  bool any = false;
  for (int i = 0; i < n_words; ++i) any = any || mask_words[i] != 0u;
  if (!any) return DV3_ERR_ARG;  // a queue without compute units never finishes a kernel
  hipStream_t s = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask_words);
  if (e != hipSuccess) return (int)e;
  *stream_out = (unsigned long long)(uintptr_t)s;
  return DV3_OK;
}

extern "C" int dv3_stream_destroy(void* stream) {
  if (!stream) return DV3_ERR_ARG;
  return (int)hipStreamDestroy((hipStream_t)stream);
}
