// common.hip - error plumbing, deterministic second-stage loss reduction and
// the small non-kernel entry points of the C ABI.
#include <stdarg.h>
#include <stdio.h>

#include "apg_device.h"

namespace apg {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return APG_OK;
  set_error("%s: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
  return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? APG_ERR_NO_DEVICE
                                                               : APG_ERR_HIP;
}

// One workgroup; thread t accumulates partials[t], [t+256], ... in double,
// then a fixed-shape LDS tree: the result does not depend on scheduling.
__global__ __launch_bounds__(256) void reduce_partials_kernel(
    const float *__restrict__ partials, int n, float *__restrict__ out) {
  __shared__ double sm[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += (double)partials[i];
  sm[threadIdx.x] = acc;
  __syncthreads();
#pragma unroll
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)sm[0];
}

int launch_reduce_partials(const float *partials, int n, float *loss,
                           hipStream_t stream) {
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, stream,
                     partials, n, loss);
  return check_launch("reduce_partials");
}

}  // namespace apg

extern "C" {

int apg_loss_partials_count(int B) {
  return B <= 0 ? 1 : (B + apg::kWave - 1) / apg::kWave;
}

int apg_version(void) { return APG_VERSION_MAJOR * 1000 + APG_VERSION_MINOR; }

const char *apg_last_error_string(void) { return apg::g_err; }

}  // extern "C"
