// issue_probe2.hip - follow-up to issue_probe.hip (round 3): does the second
// wave of a SIMD still issue in the first wave's gaps when every VALU op reads
// DISTINCT registers (issue_probe.hip's blocks re-use v30 / v31 as the shared
// operands of every instruction)?  And what do packed fp32 ops cost then?
//   hipcc --offload-arch=gfx950 -O2 tools/issue_probe2.hip -o tools/exp/issue_probe2
// Prints one JSON line per (probe, waves per SIMD): s_memtime ticks per
// instruction, median over the waves.
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
#define R8(x) R2(R4(x))
#define R32(x) R4(R8(x))

#define VCLOB                                                                  \
  "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v40",  \
      "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50",     \
      "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60",     \
      "v61", "v62", "v63", "v64", "v65", "v66", "v67", "a0", "a1", "a2", "a3",    \
      "s40", "s41", "s42", "s43", "memory"

// 8 instructions per block, sources in three different register banks
#define FMA_DIST3                                                              \
  "v_fma_f32 v10, v40, v49, v58\nv_fma_f32 v11, v41, v50, v59\n"               \
  "v_fma_f32 v12, v42, v51, v60\nv_fma_f32 v13, v43, v52, v61\n"               \
  "v_fma_f32 v14, v44, v53, v62\nv_fma_f32 v15, v45, v54, v63\n"               \
  "v_fma_f32 v16, v46, v55, v64\nv_fma_f32 v17, v47, v56, v65\n"
// ... all in one bank (register index mod 4 equal)
#define FMA_SAMEBANK                                                           \
  "v_fma_f32 v10, v40, v44, v48\nv_fma_f32 v11, v41, v45, v49\n"               \
  "v_fma_f32 v12, v42, v46, v50\nv_fma_f32 v13, v43, v47, v51\n"               \
  "v_fma_f32 v14, v52, v56, v60\nv_fma_f32 v15, v53, v57, v61\n"               \
  "v_fma_f32 v16, v54, v58, v62\nv_fma_f32 v17, v55, v59, v63\n"
#define MUL_DIST2                                                              \
  "v_mul_f32_e32 v10, v40, v49\nv_mul_f32_e32 v11, v41, v50\n"                 \
  "v_mul_f32_e32 v12, v42, v51\nv_mul_f32_e32 v13, v43, v52\n"                 \
  "v_mul_f32_e32 v14, v44, v53\nv_mul_f32_e32 v15, v45, v54\n"                 \
  "v_mul_f32_e32 v16, v46, v55\nv_mul_f32_e32 v17, v47, v56\n"
#define FMAC_DIST                                                              \
  "v_fmac_f32_e32 v10, v40, v49\nv_fmac_f32_e32 v11, v41, v50\n"               \
  "v_fmac_f32_e32 v12, v42, v51\nv_fmac_f32_e32 v13, v43, v52\n"               \
  "v_fmac_f32_e32 v14, v44, v53\nv_fmac_f32_e32 v15, v45, v54\n"               \
  "v_fmac_f32_e32 v16, v46, v55\nv_fmac_f32_e32 v17, v47, v56\n"
// dependent chain, fresh second / third operands every instruction
#define FMA_CHAIN_DIST                                                         \
  "v_fma_f32 v10, v10, v40, v49\nv_fma_f32 v10, v10, v41, v50\n"               \
  "v_fma_f32 v10, v10, v42, v51\nv_fma_f32 v10, v10, v43, v52\n"               \
  "v_fma_f32 v10, v10, v44, v53\nv_fma_f32 v10, v10, v45, v54\n"               \
  "v_fma_f32 v10, v10, v46, v55\nv_fma_f32 v10, v10, v47, v56\n"
#define PKFMA_DIST                                                             \
  "v_pk_fma_f32 v[10:11], v[40:41], v[50:51], v[60:61]\n"                      \
  "v_pk_fma_f32 v[12:13], v[42:43], v[52:53], v[62:63]\n"                      \
  "v_pk_fma_f32 v[14:15], v[44:45], v[54:55], v[64:65]\n"                      \
  "v_pk_fma_f32 v[16:17], v[46:47], v[56:57], v[66:67]\n"                      \
  "v_pk_fma_f32 v[10:11], v[48:49], v[58:59], v[60:61]\n"                      \
  "v_pk_fma_f32 v[12:13], v[40:41], v[52:53], v[64:65]\n"                      \
  "v_pk_fma_f32 v[14:15], v[42:43], v[54:55], v[66:67]\n"                      \
  "v_pk_fma_f32 v[16:17], v[44:45], v[56:57], v[60:61]\n"
#define PKMUL_DIST                                                             \
  "v_pk_mul_f32 v[10:11], v[40:41], v[50:51]\n"                                \
  "v_pk_mul_f32 v[12:13], v[42:43], v[52:53]\n"                                \
  "v_pk_mul_f32 v[14:15], v[44:45], v[54:55]\n"                                \
  "v_pk_mul_f32 v[16:17], v[46:47], v[56:57]\n"                                \
  "v_pk_mul_f32 v[10:11], v[48:49], v[58:59]\n"                                \
  "v_pk_mul_f32 v[12:13], v[60:61], v[52:53]\n"                                \
  "v_pk_mul_f32 v[14:15], v[62:63], v[54:55]\n"                                \
  "v_pk_mul_f32 v[16:17], v[64:65], v[56:57]\n"
// packed op with the constant in an SGPR pair (how a coefficient would enter)
#define PKFMA_SGPR                                                             \
  "v_pk_fma_f32 v[10:11], v[40:41], s[40:41], v[60:61]\n"                      \
  "v_pk_fma_f32 v[12:13], v[42:43], s[42:43], v[62:63]\n"                      \
  "v_pk_fma_f32 v[14:15], v[44:45], s[40:41], v[64:65]\n"                      \
  "v_pk_fma_f32 v[16:17], v[46:47], s[42:43], v[66:67]\n"                      \
  "v_pk_fma_f32 v[10:11], v[48:49], s[40:41], v[60:61]\n"                      \
  "v_pk_fma_f32 v[12:13], v[50:51], s[42:43], v[64:65]\n"                      \
  "v_pk_fma_f32 v[14:15], v[52:53], s[40:41], v[66:67]\n"                      \
  "v_pk_fma_f32 v[16:17], v[54:55], s[42:43], v[60:61]\n"
// the fixed-wing mix: 15 plain distinct-operand ops, one transcendental
#define WING_MIX                                                               \
  FMA_DIST3 "v_mul_f32_e32 v18, v40, v49\nv_fmac_f32_e32 v18, v41, v50\n"      \
  "v_fmac_f32_e32 v18, v42, v51\nv_mul_f32_e32 v19, v43, v52\n"                \
  "v_fmac_f32_e32 v19, v44, v53\nv_fma_f32 v18, v45, v54, v19\n"               \
  "v_add_f32_e32 v19, v46, v55\nv_rcp_f32_e32 v18, v47\n"

// the same 15 : 1 ratio with the transcendentals CLUSTERED (60 plain, 4 rcp):
// does a transcendental cost the second wave's overlap only where it sits?
#define PLAIN15                                                                \
  FMA_DIST3 "v_mul_f32_e32 v18, v40, v49\nv_fmac_f32_e32 v18, v41, v50\n"      \
  "v_fmac_f32_e32 v18, v42, v51\nv_mul_f32_e32 v19, v43, v52\n"                \
  "v_fmac_f32_e32 v19, v44, v53\nv_fma_f32 v18, v45, v54, v19\n"               \
  "v_add_f32_e32 v19, v46, v55\n"
#define WING_MIX_CLUSTER                                                       \
  PLAIN15 PLAIN15 PLAIN15 PLAIN15                                              \
  "v_rcp_f32_e32 v18, v47\nv_rcp_f32_e32 v19, v46\n"                           \
  "v_rcp_f32_e32 v16, v45\nv_rcp_f32_e32 v17, v44\n"
#define PLAIN16 PLAIN15 "v_add_f32_e32 v17, v47, v56\n"

// ---- round 3b: which FORM of a packed op is slow?  (the rollout kernel with
// packed arithmetic ran 2.4 x slower than the scalar one although it issues
// 10 % fewer instructions)
#define PK8(x) x x x x x x x x
#define PK_PLAIN PK8("v_pk_fma_f32 v[10:11], v[40:41], v[50:51], v[60:61]\n")
#define PK_BCAST PK8("v_pk_fma_f32 v[10:11], v[40:41], v[50:51], v[60:61] op_sel_hi:[0,1,1]\n")
#define PK_OPSEL PK8("v_pk_fma_f32 v[10:11], v[40:41], v[50:51], v[60:61] op_sel:[0,1,0] op_sel_hi:[0,0,1]\n")
#define PK_LIT PK8("v_pk_mul_f32 v[10:11], v[40:41], 0.15915494 op_sel_hi:[1,0]\n")
#define PK_INLINE PK8("v_pk_add_f32 v[10:11], v[40:41], -0.5 op_sel_hi:[1,0]\n")
#define PK_NEG PK8("v_pk_add_f32 v[10:11], v[40:41], v[50:51] neg_lo:[0,1] neg_hi:[0,1]\n")
#define PK_SGPR_BCAST PK8("v_pk_fma_f32 v[10:11], v[40:41], v[50:51], s[40:41] op_sel_hi:[0,1,1]\n")
// a packed op reading a pair whose LOW half a scalar op has just written
#define PK_AFTER_SCALAR                                                        \
  "v_mul_f32_e32 v40, v42, v43\nv_pk_fma_f32 v[10:11], v[40:41], v[50:51], v[60:61]\n"  \
  "v_mul_f32_e32 v44, v46, v47\nv_pk_fma_f32 v[12:13], v[44:45], v[52:53], v[62:63]\n"  \
  "v_mul_f32_e32 v48, v42, v43\nv_pk_fma_f32 v[14:15], v[48:49], v[54:55], v[64:65]\n"  \
  "v_mul_f32_e32 v56, v46, v47\nv_pk_fma_f32 v[16:17], v[56:57], v[58:59], v[66:67]\n"
// a scalar op reading one half of a pair a packed op has just written
#define SCALAR_AFTER_PK                                                        \
  "v_pk_fma_f32 v[10:11], v[40:41], v[50:51], v[60:61]\nv_mul_f32_e32 v18, v10, v43\n"  \
  "v_pk_fma_f32 v[12:13], v[44:45], v[52:53], v[62:63]\nv_mul_f32_e32 v19, v13, v43\n"  \
  "v_pk_fma_f32 v[14:15], v[48:49], v[54:55], v[64:65]\nv_mul_f32_e32 v18, v14, v43\n"  \
  "v_pk_fma_f32 v[16:17], v[56:57], v[58:59], v[66:67]\nv_mul_f32_e32 v19, v17, v43\n"
// packed op right after a transcendental that wrote one of its halves
#define PK_AFTER_TRANS                                                         \
  "v_sin_f32_e32 v40, v42\nv_cos_f32_e32 v41, v42\nv_pk_mul_f32 v[10:11], v[40:41], v[50:51]\n"  \
  "v_sin_f32_e32 v44, v46\nv_cos_f32_e32 v45, v46\nv_pk_mul_f32 v[12:13], v[44:45], v[52:53]\n"  \
  "v_mov_b32_e32 v18, v10\nv_mov_b32_e32 v19, v12\n"
// AGPR round trips around packed ops
#define PK_ACC                                                                 \
  "v_accvgpr_write_b32 a0, v10\nv_accvgpr_write_b32 a1, v11\n"                 \
  "v_pk_fma_f32 v[10:11], v[40:41], v[50:51], v[60:61]\n"                       \
  "v_accvgpr_read_b32 v40, a0\nv_accvgpr_read_b32 v41, a1\n"                   \
  "v_pk_fma_f32 v[12:13], v[40:41], v[52:53], v[62:63]\n"                       \
  "v_accvgpr_write_b32 a2, v12\nv_accvgpr_read_b32 v44, a2\n"

enum { P_FMA_DIST3, P_FMA_SAMEBANK, P_MUL_DIST2, P_FMAC_DIST, P_FMA_CHAIN_DIST,
       P_PKFMA_DIST, P_PKMUL_DIST, P_PKFMA_SGPR, P_WING_MIX, P_WING_CLUSTER,
       P_PLAIN16, P_PK_PLAIN, P_PK_BCAST, P_PK_OPSEL, P_PK_LIT, P_PK_INLINE, P_PK_NEG,
       P_PK_SGPR_BCAST, P_PK_AFTER_SCALAR, P_SCALAR_AFTER_PK, P_PK_AFTER_TRANS,
       P_PK_ACC, P_COUNT };
static const char *kNames[P_COUNT] = {
    "fma_dist3", "fma_samebank", "mul_dist2", "fmac_dist", "fma_chain_dist",
    "pk_fma_dist", "pk_mul_dist", "pk_fma_sgpr", "wing_mix_15plain_1rcp",
    "wing_mix_60plain_4rcp_clustered", "plain16_no_transcendental",
    "pk_plain_dep", "pk_bcast_src0", "pk_op_sel_mixed", "pk_mul_literal",
    "pk_add_inline_const", "pk_add_neg", "pk_fma_sgpr_src2_bcast",
    "pk_after_scalar_halfwrite", "scalar_after_pk_halfread", "pk_after_sincos",
    "pk_with_agpr_roundtrips"};
static const int kInstr[P_COUNT] = {8, 8, 8, 8, 8, 8, 8, 8, 16, 64, 16,
                                    8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8};

template <int P>
__global__ __launch_bounds__(64) void probe(unsigned long long *out, int reps) {
  unsigned long long t0, t1;
  asm volatile(
      "v_mov_b32 v40, 0x3f7ff000\nv_mov_b32 v41, 0x3f7fe000\nv_mov_b32 v42, 0x3f7fd000\n"
      "v_mov_b32 v43, 0x3f7fc000\nv_mov_b32 v44, 0x3f7fb000\nv_mov_b32 v45, 0x3f7fa000\n"
      "v_mov_b32 v46, 0x3f7f9000\nv_mov_b32 v47, 0x3f7f8000\nv_mov_b32 v48, 0x3f7f7000\n"
      "v_mov_b32 v49, 0x3f7f6000\nv_mov_b32 v50, 0x3f7f5000\nv_mov_b32 v51, 0x3f7f4000\n"
      "v_mov_b32 v52, 0x3f7f3000\nv_mov_b32 v53, 0x3f7f2000\nv_mov_b32 v54, 0x3f7f1000\n"
      "v_mov_b32 v55, 0x3f7f0000\nv_mov_b32 v56, 0x3f7ef000\nv_mov_b32 v57, 0x3f7ee000\n"
      "v_mov_b32 v58, 0x33000000\nv_mov_b32 v59, 0x33000000\nv_mov_b32 v60, 0x33000000\n"
      "v_mov_b32 v61, 0x33000000\nv_mov_b32 v62, 0x33000000\nv_mov_b32 v63, 0x33000000\n"
      "v_mov_b32 v64, 0x33000000\nv_mov_b32 v65, 0x33000000\nv_mov_b32 v66, 0x33000000\n"
      "v_mov_b32 v67, 0x33000000\nv_mov_b32 v10, 1.0\nv_mov_b32 v11, 1.0\n"
      "v_mov_b32 v12, 1.0\nv_mov_b32 v13, 1.0\nv_mov_b32 v14, 1.0\nv_mov_b32 v15, 1.0\n"
      "v_mov_b32 v16, 1.0\nv_mov_b32 v17, 1.0\nv_mov_b32 v18, 1.0\nv_mov_b32 v19, 1.0\n"
      "s_mov_b32 s40, 0x3f7ff000\ns_mov_b32 s41, 0x3f7ff000\n"
      "s_mov_b32 s42, 0x3f7fe000\ns_mov_b32 s43, 0x3f7fe000\n" ::
          : VCLOB);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\ns_memtime %0\ns_waitcnt lgkmcnt(0)"
               : "=s"(t0)::"memory");
  for (int r = 0; r < reps; ++r) {
    if constexpr (P == P_FMA_DIST3) asm volatile(R32(FMA_DIST3) ::: VCLOB);
    if constexpr (P == P_FMA_SAMEBANK) asm volatile(R32(FMA_SAMEBANK) ::: VCLOB);
    if constexpr (P == P_MUL_DIST2) asm volatile(R32(MUL_DIST2) ::: VCLOB);
    if constexpr (P == P_FMAC_DIST) asm volatile(R32(FMAC_DIST) ::: VCLOB);
    if constexpr (P == P_FMA_CHAIN_DIST) asm volatile(R32(FMA_CHAIN_DIST) ::: VCLOB);
    if constexpr (P == P_PKFMA_DIST) asm volatile(R32(PKFMA_DIST) ::: VCLOB);
    if constexpr (P == P_PKMUL_DIST) asm volatile(R32(PKMUL_DIST) ::: VCLOB);
    if constexpr (P == P_PKFMA_SGPR) asm volatile(R32(PKFMA_SGPR) ::: VCLOB);
    if constexpr (P == P_WING_MIX) asm volatile(R32(WING_MIX) ::: VCLOB);
    if constexpr (P == P_WING_CLUSTER) asm volatile(R32(WING_MIX_CLUSTER) ::: VCLOB);
    if constexpr (P == P_PLAIN16) asm volatile(R32(PLAIN16) ::: VCLOB);
    if constexpr (P == P_PK_PLAIN) asm volatile(R32(PK_PLAIN) ::: VCLOB);
    if constexpr (P == P_PK_BCAST) asm volatile(R32(PK_BCAST) ::: VCLOB);
    if constexpr (P == P_PK_OPSEL) asm volatile(R32(PK_OPSEL) ::: VCLOB);
    if constexpr (P == P_PK_LIT) asm volatile(R32(PK_LIT) ::: VCLOB);
    if constexpr (P == P_PK_INLINE) asm volatile(R32(PK_INLINE) ::: VCLOB);
    if constexpr (P == P_PK_NEG) asm volatile(R32(PK_NEG) ::: VCLOB);
    if constexpr (P == P_PK_SGPR_BCAST) asm volatile(R32(PK_SGPR_BCAST) ::: VCLOB);
    if constexpr (P == P_PK_AFTER_SCALAR) asm volatile(R32(PK_AFTER_SCALAR) ::: VCLOB);
    if constexpr (P == P_SCALAR_AFTER_PK) asm volatile(R32(SCALAR_AFTER_PK) ::: VCLOB);
    if constexpr (P == P_PK_AFTER_TRANS) asm volatile(R32(PK_AFTER_TRANS) ::: VCLOB);
    if constexpr (P == P_PK_ACC) asm volatile(R32(PK_ACC) ::: VCLOB);
  }
  asm volatile("s_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float sink;
  asm volatile("v_add_f32 %0, v10, v11\n" : "=v"(sink)::"memory");
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (sink == 123.456f) out[blockIdx.x] = 0;
}

template <int P>
void run(unsigned long long *d_out, int waves_per_simd, int reps = 64) {
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
  const double cyc = (double)h[grid / 2], n = 32.0 * kInstr[P] * reps;
  printf("{\"probe\": \"%s\", \"waves_per_simd\": %d, \"instr\": %.0f, "
         "\"ticks_per_instr\": %.3f, \"kernel_us\": %.2f}\n",
         kNames[P], waves_per_simd, n, cyc / n, ms * 1e3);
  fflush(stdout);
}

int main() {
  unsigned long long *d_out;
  CK(hipMalloc(&d_out, 8192 * 8));
  if (true) {   // the packed-form probes: one wave per SIMD
    run<P_PK_PLAIN>(d_out, 1);
    run<P_PK_BCAST>(d_out, 1);
    run<P_PK_OPSEL>(d_out, 1);
    run<P_PK_LIT>(d_out, 1);
    run<P_PK_INLINE>(d_out, 1);
    run<P_PK_NEG>(d_out, 1);
    run<P_PK_SGPR_BCAST>(d_out, 1);
    run<P_PK_AFTER_SCALAR>(d_out, 1);
    run<P_SCALAR_AFTER_PK>(d_out, 1);
    run<P_PK_AFTER_TRANS>(d_out, 1);
    run<P_PK_ACC>(d_out, 1);
  }
  for (int w = 1; w <= 2; ++w) {
    run<P_FMA_DIST3>(d_out, w);
    run<P_FMA_SAMEBANK>(d_out, w);
    run<P_MUL_DIST2>(d_out, w);
    run<P_FMAC_DIST>(d_out, w);
    run<P_FMA_CHAIN_DIST>(d_out, w);
    run<P_PKFMA_DIST>(d_out, w);
    run<P_PKMUL_DIST>(d_out, w);
    run<P_PKFMA_SGPR>(d_out, w);
    run<P_WING_MIX>(d_out, w);
    run<P_WING_CLUSTER>(d_out, w, 16);
    run<P_PLAIN16>(d_out, w);
  }
  return 0;
}
