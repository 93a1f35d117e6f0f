// issue_probe3.hip - round 6: what the fp16-split matrix kernels are made of.
// Do v_mfma_f32_32x32x16_f16 and the split arithmetic (v_cvt_pk_f16_f32,
// v_fma_mix_f32, v_max_i32, v_perm_b32, v_ldexp_f32) overlap - inside one wave
// (VALU in the shadow of a matrix instruction) and between the two waves of a
// SIMD?  (lstm_gate_wgrad_kernel: matrix pipe 27 % busy, VALU 41 %, and the
// kernel takes their SUM)
//   hipcc --offload-arch=gfx950 -O2 tools/issue_probe3.hip -o tools/exp/issue_probe3
// One JSON line per (probe, waves per SIMD): s_memtime ticks per block (median
// over the waves), instructions per block, kernel time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);   \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

#define R2(x) x x
#define R4(x) R2(R2(x))
#define R16(x) R4(R4(x))

#define VCLOB                                                                  \
  "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20",  \
      "v21", "v22", "v23", "v24", "v25", "v40", "v41", "v42", "v43", "v44",     \
      "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54",     \
      "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v100",    \
      "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109",  \
      "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118",  \
      "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127",  \
      "v128", "v129", "v130", "v131", "memory"

#define MFMA_A "v_mfma_f32_32x32x16_f16 v[100:115], v[40:43], v[44:47], v[100:115]\n"
#define MFMA_B "v_mfma_f32_32x32x16_f16 v[116:131], v[48:51], v[52:55], v[116:131]\n"
// four independent plain ops / four ops of the split (two values -> h, l)
#define FMA4                                                                   \
  "v_fma_f32 v10, v56, v57, v58\nv_fma_f32 v11, v57, v58, v59\n"               \
  "v_fma_f32 v12, v58, v59, v60\nv_fma_f32 v13, v59, v60, v61\n"
#define SPLIT4                                                                 \
  "v_cvt_pk_f16_f32 v14, v56, v57\n"                                           \
  "v_fma_mix_f32 v15, v14, -1.0, v56 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n"       \
  "v_fma_mix_f32 v16, v14, -1.0, v57 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"       \
  "v_cvt_pk_f16_f32 v17, v15, v16\n"
#define SPLIT4B                                                                \
  "v_cvt_pk_f16_f32 v18, v58, v59\n"                                           \
  "v_fma_mix_f32 v19, v18, -1.0, v58 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n"       \
  "v_fma_mix_f32 v20, v18, -1.0, v59 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"       \
  "v_cvt_pk_f16_f32 v21, v19, v20\n"
#define CVT4                                                                   \
  "v_cvt_pk_f16_f32 v14, v56, v57\nv_cvt_pk_f16_f32 v15, v57, v58\n"           \
  "v_cvt_pk_f16_f32 v16, v58, v59\nv_cvt_pk_f16_f32 v17, v59, v60\n"
#define MIX4                                                                   \
  "v_fma_mix_f32 v14, v56, -1.0, v57 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n"       \
  "v_fma_mix_f32 v15, v57, -1.0, v58 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"       \
  "v_fma_mix_f32 v16, v58, -1.0, v59 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n"       \
  "v_fma_mix_f32 v17, v59, -1.0, v60 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define MAXI4                                                                  \
  "v_max_i32_e32 v14, 0, v56\nv_max_i32_e32 v15, 0, v57\n"                     \
  "v_max_i32_e32 v16, 0, v58\nv_max_i32_e32 v17, 0, v59\n"
#define PERM4                                                                  \
  "v_perm_b32 v14, v56, v57, v62\nv_perm_b32 v15, v57, v58, v62\n"             \
  "v_perm_b32 v16, v58, v59, v62\nv_perm_b32 v17, v59, v60, v62\n"
#define LDEXP4                                                                 \
  "v_ldexp_f32 v14, v56, v63\nv_ldexp_f32 v15, v57, v63\n"                     \
  "v_ldexp_f32 v16, v58, v63\nv_ldexp_f32 v17, v59, v63\n"

enum { P_MFMA_CHAIN, P_MFMA_TWO, P_FMA, P_SPLIT, P_CVT, P_MIX, P_MAXI, P_PERM, P_LDEXP,
       P_MFMA_FMA8, P_MFMA_SPLIT8, P_MFMA_FMA4, P_MFMA_SPLIT16, P_COUNT };
static const char *kNames[P_COUNT] = {
    "mfma_chain_1acc", "mfma_two_accs", "fma_x8", "split_x8", "cvt_pk_x8", "fma_mix_x8",
    "max_i32_x8", "perm_x8", "ldexp_x8", "mfma_then_8_fma", "mfma_then_8_split_ops",
    "mfma_then_4_fma", "mfma_then_16_split_ops"};
static const int kInstr[P_COUNT] = {2, 2, 8, 8, 8, 8, 8, 8, 8, 9, 9, 5, 17};

template <int P>
__global__ __launch_bounds__(64) void probe(unsigned long long *out, int reps) {
  unsigned long long t0, t1;
  asm volatile(
      "v_mov_b32 v40, 0x3c003c00\nv_mov_b32 v41, 0x3c003c00\nv_mov_b32 v42, 0x3c003c00\n"
      "v_mov_b32 v43, 0x3c003c00\nv_mov_b32 v44, 0x1c001c00\nv_mov_b32 v45, 0x1c001c00\n"
      "v_mov_b32 v46, 0x1c001c00\nv_mov_b32 v47, 0x1c001c00\nv_mov_b32 v48, 0x3c003c00\n"
      "v_mov_b32 v49, 0x3c003c00\nv_mov_b32 v50, 0x3c003c00\nv_mov_b32 v51, 0x3c003c00\n"
      "v_mov_b32 v52, 0x1c001c00\nv_mov_b32 v53, 0x1c001c00\nv_mov_b32 v54, 0x1c001c00\n"
      "v_mov_b32 v55, 0x1c001c00\nv_mov_b32 v56, 0x3f7ff000\nv_mov_b32 v57, 0x3f7fe000\n"
      "v_mov_b32 v58, 0x3f7fd000\nv_mov_b32 v59, 0x3f7fc000\nv_mov_b32 v60, 0x3f7fb000\n"
      "v_mov_b32 v61, 0x3f7fa000\nv_mov_b32 v62, 0x05040100\nv_mov_b32 v63, 3\n"
      "v_mov_b32 v100, 0\nv_mov_b32 v101, 0\nv_mov_b32 v102, 0\nv_mov_b32 v103, 0\n"
      "v_mov_b32 v104, 0\nv_mov_b32 v105, 0\nv_mov_b32 v106, 0\nv_mov_b32 v107, 0\n"
      "v_mov_b32 v108, 0\nv_mov_b32 v109, 0\nv_mov_b32 v110, 0\nv_mov_b32 v111, 0\n"
      "v_mov_b32 v112, 0\nv_mov_b32 v113, 0\nv_mov_b32 v114, 0\nv_mov_b32 v115, 0\n"
      "v_mov_b32 v116, 0\nv_mov_b32 v117, 0\nv_mov_b32 v118, 0\nv_mov_b32 v119, 0\n"
      "v_mov_b32 v120, 0\nv_mov_b32 v121, 0\nv_mov_b32 v122, 0\nv_mov_b32 v123, 0\n"
      "v_mov_b32 v124, 0\nv_mov_b32 v125, 0\nv_mov_b32 v126, 0\nv_mov_b32 v127, 0\n"
      "v_mov_b32 v128, 0\nv_mov_b32 v129, 0\nv_mov_b32 v130, 0\nv_mov_b32 v131, 0\n" ::
          : VCLOB);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\ns_memtime %0\ns_waitcnt lgkmcnt(0)"
               : "=s"(t0)::"memory");
  for (int r = 0; r < reps; ++r) {
    if constexpr (P == P_MFMA_CHAIN) asm volatile(R16(MFMA_A MFMA_A) ::: VCLOB);
    if constexpr (P == P_MFMA_TWO) asm volatile(R16(MFMA_A MFMA_B) ::: VCLOB);
    if constexpr (P == P_FMA) asm volatile(R16(FMA4 FMA4) ::: VCLOB);
    if constexpr (P == P_SPLIT) asm volatile(R16(SPLIT4 SPLIT4B) ::: VCLOB);
    if constexpr (P == P_CVT) asm volatile(R16(CVT4 CVT4) ::: VCLOB);
    if constexpr (P == P_MIX) asm volatile(R16(MIX4 MIX4) ::: VCLOB);
    if constexpr (P == P_MAXI) asm volatile(R16(MAXI4 MAXI4) ::: VCLOB);
    if constexpr (P == P_PERM) asm volatile(R16(PERM4 PERM4) ::: VCLOB);
    if constexpr (P == P_LDEXP) asm volatile(R16(LDEXP4 LDEXP4) ::: VCLOB);
    if constexpr (P == P_MFMA_FMA8) asm volatile(R16(MFMA_A FMA4 FMA4) ::: VCLOB);
    if constexpr (P == P_MFMA_SPLIT8) asm volatile(R16(MFMA_A SPLIT4 SPLIT4B) ::: VCLOB);
    if constexpr (P == P_MFMA_FMA4) asm volatile(R16(MFMA_A FMA4) ::: VCLOB);
    if constexpr (P == P_MFMA_SPLIT16)
      asm volatile(R16(MFMA_A SPLIT4 SPLIT4B SPLIT4 SPLIT4B) ::: VCLOB);
  }
  asm volatile("s_nop 7\ns_nop 7\ns_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float sink;
  asm volatile("s_nop 7\ns_nop 7\nv_add_f32 %0, v100, v116\n" : "=v"(sink)::"memory");
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (sink == 123.456f) out[blockIdx.x] = 0;
}

template <int P>
void run(unsigned long long *d_out, int waves_per_simd, int reps = 32) {
  const int grid = 1024 * waves_per_simd;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(probe<P>, dim3(grid), dim3(64), 0, 0, d_out, reps);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(probe<P>, dim3(grid), dim3(64), 0, 0, d_out, reps);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(grid);
  CK(hipMemcpy(h.data(), d_out, grid * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double blocks = 16.0 * reps;
  printf("{\"probe\": \"%s\", \"waves_per_simd\": %d, \"instr_per_block\": %d, "
         "\"ticks_per_block\": %.2f, \"kernel_us\": %.2f, \"us_per_block_x1e3\": %.3f}\n",
         kNames[P], waves_per_simd, kInstr[P], (double)h[grid / 2] / blocks, ms * 1e3,
         ms * 1e6 / blocks);
  fflush(stdout);
}

int main() {
  unsigned long long *d_out;
  CK(hipMalloc(&d_out, 8192 * 8));
  for (int w = 1; w <= 2; ++w) {
    run<P_MFMA_CHAIN>(d_out, w);
    run<P_MFMA_TWO>(d_out, w);
    run<P_FMA>(d_out, w);
    run<P_SPLIT>(d_out, w);
    run<P_CVT>(d_out, w);
    run<P_MIX>(d_out, w);
    run<P_MAXI>(d_out, w);
    run<P_PERM>(d_out, w);
    run<P_LDEXP>(d_out, w);
    run<P_MFMA_FMA4>(d_out, w);
    run<P_MFMA_FMA8>(d_out, w);
    run<P_MFMA_SPLIT8>(d_out, w);
    run<P_MFMA_SPLIT16>(d_out, w);
  }
  return 0;
}
