// Probe (round 6): does a batch's row gather run faster when the rows' cache lines were
// read a moment ago (Infinity Cache / L2)?  touch_rows reads one dword of every
// 128-byte line of the first `bytes` of each indexed row and drops it.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/exp/libtouch_probe.so tools/touch_probe.hip
#include <hip/hip_runtime.h>
__global__ void touch_rows_kernel(const char *base, long long pitch, int lines,
                                  const long long *index, int B, long long total) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long row = t / lines;
  const int line = (int)(t - row * lines);
  if (row >= B) return;
  // (the row's first line starts wherever the row starts: line k = the 128-byte
  // aligned line holding byte 128 k of the row)
  const long long off = index[row] * pitch + 128ll * line;
  if (off + 4 > total) return;            // (past the tensor's last row)
  const char *p = base + off;
  unsigned v;
  asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  asm volatile("" ::"v"(v));
}
extern "C" int touch_rows(const void *base, long long pitch, int bytes, const long long *index,
                          int B, long long total, void *stream) {
  const int lines = (bytes + 127) / 128 + 1;   // (+1: a row that starts mid-line)
  const long long n = (long long)B * lines;
  hipLaunchKernelGGL(touch_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const char *)base, pitch, lines, index, B, total);
  return (int)hipGetLastError();
}
