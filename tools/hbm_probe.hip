// hbm_probe.hip - calibration micro-benchmarks for the roofline in DESIGN.md
// (not part of the product library).  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_probe.hip -o gpurun_out/hbm_probe && gpurun_out/hbm_probe
// 1. float4 copy of 1 GiB            -> achievable HBM copy bandwidth
// 2. "rollout-shaped" streams at B trajectories, one trajectory per lane,
//    one wave per workgroup: R dword planes read, W dword planes written,
//    no arithmetic -> the floor the fused rollout kernel can approach at
//    that batch size (launch ramp + HBM latency + one round of waves).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) b[i] = a[i];
}

template <int R, int W, int LAUX = 0, int SAUX = 0>
__global__ __launch_bounds__(64) void stream_planes(const float* __restrict__ in, float* __restrict__ out, int B) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, R * B * 4, 0x00020000);
  __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, W * B * 4, 0x00020000);
  float v[R];
#pragma unroll
  for (int i = 0; i < R; ++i) v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, b * 4, i * B * 4, LAUX));
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < R; ++i) acc += v[i];
#pragma unroll
  for (int i = 0; i < W; ++i) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc + (float)i), ro, b * 4, i * B * 4, SAUX);
}

// the same byte counts moved with 16-byte accesses per lane (rows of four
// floats, [row][B][4]): R4 row loads and W4 row stores per wave
template <int R4, int W4, int SAUX = 0>
__global__ __launch_bounds__(64) void stream_rows(const float* __restrict__ in, float* __restrict__ out, int B) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const int b = blockIdx.x * 64 + threadIdx.x;
  __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, R4 * B * 16, 0x00020000);
  __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, W4 * B * 16, 0x00020000);
  u4 v[R4];
#pragma unroll
  for (int i = 0; i < R4; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(ri, b * 16, i * B * 16, 0);
  u4 acc = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < R4; ++i) acc += v[i];
#pragma unroll
  for (int i = 0; i < W4; ++i) { acc.x += i; __builtin_amdgcn_raw_buffer_store_b128(acc, ro, b * 16, i * B * 16, SAUX); }
}

template <int R4, int W4, int SAUX = 0>
void run_rows(int B, int nsets) {
  std::vector<float*> ins(nsets), outs(nsets);
  for (int i = 0; i < nsets; ++i) {
    CK(hipMalloc(&ins[i], (size_t)R4 * B * 16)); CK(hipMalloc(&outs[i], (size_t)W4 * B * 16));
    CK(hipMemset(ins[i], 0, (size_t)R4 * B * 16));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((stream_rows<R4, W4, SAUX>), dim3(B / 64), dim3(64), 0, 0, ins[i % nsets], outs[i % nsets], B);
  CK(hipDeviceSynchronize());
  const int K = 400;
  CK(hipEventRecord(e0));
  for (int i = 0; i < K; ++i) hipLaunchKernelGGL((stream_rows<R4, W4, SAUX>), dim3(B / 64), dim3(64), 0, 0, ins[i % nsets], outs[i % nsets], B);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double us = ms * 1e3 / K, bytes = (double)(R4 + W4) * B * 16;
  printf("rows(16B) R4=%d W4=%d B=%d sets=%d saux=%d: %.2f us/launch  %.1f GB/s\n", R4, W4, B, nsets, SAUX, us, bytes / us / 1e3);
  for (int i = 0; i < nsets; ++i) { CK(hipFree(ins[i])); CK(hipFree(outs[i])); }
}

template <int R, int W, int LAUX = 0, int SAUX = 0>
void run_stream(int B, int nsets) {
  std::vector<float*> ins(nsets), outs(nsets);
  for (int i = 0; i < nsets; ++i) {
    CK(hipMalloc(&ins[i], (size_t)R * B * 4)); CK(hipMalloc(&outs[i], (size_t)W * B * 4));
    CK(hipMemset(ins[i], 0, (size_t)R * B * 4));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((stream_planes<R, W, LAUX, SAUX>), dim3(B / 64), dim3(64), 0, 0, ins[i % nsets], outs[i % nsets], B);
  CK(hipDeviceSynchronize());
  const int K = 400;
  CK(hipEventRecord(e0));
  for (int i = 0; i < K; ++i) hipLaunchKernelGGL((stream_planes<R, W, LAUX, SAUX>), dim3(B / 64), dim3(64), 0, 0, ins[i % nsets], outs[i % nsets], B);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double us = ms * 1e3 / K, bytes = (double)(R + W) * B * 4;
  printf("stream R=%d W=%d B=%d sets=%d laux=%d saux=%d: %.2f us/launch  %.1f GB/s\n", R, W, B, nsets, LAUX, SAUX, us, bytes / us / 1e3);
  for (int i = 0; i < nsets; ++i) { CK(hipFree(ins[i])); CK(hipFree(outs[i])); }
}

// launch floor: a kernel that does (almost) nothing, back to back
__global__ void tiny(float* out) { if (threadIdx.x == 0) out[blockIdx.x] = 1.f; }
void run_floor(int grid, int block) {
  float* out; CK(hipMalloc(&out, 65536 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(block), 0, 0, out);
  CK(hipDeviceSynchronize());
  const int K = 1000;
  CK(hipEventRecord(e0));
  for (int i = 0; i < K; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(block), 0, 0, out);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("launch floor grid=%d block=%d: %.2f us/launch\n", grid, block, ms * 1e3 / K);
  CK(hipFree(out));
}

int main(int argc, char **argv) {
  if (argc > 1) {  // "calib": only the stream the PMC passes are calibrated on
    run_rows<28, 10, 2>(65536, 20);
    return 0;
  }
  run_floor(1024, 64); run_floor(512, 128); run_floor(256, 256); run_floor(256, 64);
  run_floor(2048, 64); run_floor(64, 64); run_floor(1, 64); run_floor(1024, 256);
  {
    size_t bytes = (size_t)1 << 30, n = bytes / 16;
    float4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(copy4, dim3(2048), dim3(256), 0, 0, a, b, n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(copy4, dim3(2048), dim3(256), 0, 0, a, b, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("float4 copy 1 GiB: %.1f GB/s (read+write)\n", 2.0 * bytes * 10 / (ms * 1e-3) / 1e9);
    CK(hipFree(a)); CK(hipFree(b));
  }
  run_rows<28, 0, 2>(65536, 8);   // reads only
  run_rows<1, 10, 2>(65536, 8);   // (almost) writes only, nt
  run_rows<1, 10, 0>(65536, 8);   // plain
  run_rows<1, 10, 16>(65536, 8);  // sc1
  run_rows<1, 10, 18>(65536, 8);  // sc1 nt
  run_rows<1, 10, 3>(65536, 8);   // sc0 nt
  run_rows<13, 0, 2>(65536, 8);   // state0 + actions
  run_rows<28, 10, 0>(65536, 8);
  run_rows<28, 10, 2>(65536, 8);
  run_rows<28, 10, 2>(65536, 16);
  run_rows<28, 10, 2>(65536, 20);  // bench.py's round-3 protocol (inputs 588 MB)
  run_rows<28, 0, 2>(65536, 20);   // reads only, from HBM
  run_rows<1, 10, 2>(65536, 20);
  run_rows<28, 10, 2>(65536, 1);
  run_stream<112, 40, 0, 2>(65536, 16);
  run_stream<112, 40>(65536, 8);
  run_stream<112, 40>(65536, 1);
  run_stream<112, 40, 0, 2>(65536, 8);
  run_stream<64, 40>(65536, 8);
  run_stream<16, 8>(65536, 8);
  run_stream<1, 1>(65536, 8);
  return 0;
}
