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

// One workgroup.  Thread t sums partials[t], [t+256], ... (independent loads,
// double accumulation), then a wave butterfly and a 4-entry LDS combine: the
// shape is fixed, so the result does not depend on scheduling.
__global__ __launch_bounds__(256) void reduce_partials_kernel(
    const float *__restrict__ partials, int n, float *__restrict__ out) {
  __shared__ double sm[4];
  double acc = 0.0;
#pragma unroll 4
  for (int i = threadIdx.x; i < n; i += 256) acc += (double)partials[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)((sm[0] + sm[1]) + (sm[2] + sm[3]));
}

int launch_reduce_partials(const float *partials, int n, float *loss,
                           hipStream_t stream) {
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, stream,
                     partials, n, loss);
  return check_launch("reduce_partials");
}

// [B][R] row-major (the reference's batch-major tensors) -> [R][B] planes:
// 64 x 64 tiles through LDS, both sides coalesced.
// With `index` the rows are gathered: dst[r][b] = src[index[b]][r] - the
// minibatch selection of the training loop folded into the layout change.
__global__ __launch_bounds__(256) void to_soa_kernel(const float *__restrict__ src,
                                                     const long long *__restrict__ index,
                                                     int B, int R, int ld,
                                                     float *__restrict__ dst) {
  __shared__ float tile[64][65];
  const int b0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
  for (int i = y; i < 64; i += 4) {  // row b0 + i, columns r0 + x
    const int b = b0 + i, r = r0 + x;
    if (b < B && r < R) {
      const size_t row = index ? (size_t)index[b] : (size_t)b;
      tile[i][x] = src[row * ld + r];
    }
  }
  __syncthreads();
  for (int i = y; i < 64; i += 4) {  // plane r0 + i, trajectories b0 + x
    const int r = r0 + i, b = b0 + x;
    if (r < R && b < B) dst[(size_t)r * B + b] = tile[x][i];
  }
}

}  // namespace apg

extern "C" {

int apg_to_soa(const float *src, const long long *index, int B, int R, int ld,
               float *dst, apg_stream_t stream) {
  if (B < 0 || R < 1 || ld < R) {
    apg::set_error("apg_to_soa: need B >= 0, R >= 1, ld >= R");
    return APG_ERR_ARG;
  }
  if (B == 0) return APG_OK;
  if (!src || !dst) {
    apg::set_error("apg_to_soa: NULL pointer");
    return APG_ERR_ARG;
  }
  hipLaunchKernelGGL(apg::to_soa_kernel, dim3((B + 63) / 64, (R + 63) / 64),
                     dim3(256), 0, (hipStream_t)stream, src, index, B, R, ld, dst);
  return apg::check_launch("to_soa");
}

int apg_reduce_loss_partials(const float *partials, int n, float *loss,
                             apg_stream_t stream) {
  if (!partials || !loss || n < 0) {
    apg::set_error("apg_reduce_loss_partials: bad arguments");
    return APG_ERR_ARG;
  }
  return apg::launch_reduce_partials(partials, n, loss, (hipStream_t)stream);
}

int apg_loss_partials_count(int B) {
  return B <= 0 ? 1 : (B + apg::kWave - 1) / apg::kWave;
}

int apg_version(void) { return APG_VERSION_MAJOR * 1000 + APG_VERSION_MINOR; }

const char *apg_last_error_string(void) { return apg::g_err; }

}  // extern "C"
