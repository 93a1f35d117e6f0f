// issue_probe.hip - instruction-issue micro-benchmarks for gfx950 (MI355X).
// Not part of the product library; answers the questions DESIGN.md §5/§9 left
// open about the register-resident rollout kernel, which runs ONE wave per
// SIMD (1 024 waves at B = 65 536):
//   * how fast can one wave issue dependent / independent fp32 VALU ops,
//     and what does a second / fourth wave on the SIMD buy;
//   * what v_pk_*_f32, v_sin/v_cos, v_accvgpr moves, SALU ops and LDS ops cost
//     in that regime;
//   * the issue cost of buffer loads / stores (plain, direct-to-LDS, wide).
// Every pattern is a hand-written instruction block timed with s_memtime
// inside the wave (median over all waves of the launch).
//   hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o gpurun_out/issue_probe
//   gpurun_out/issue_probe > gpurun_out/issue_probe.jsonl
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(x)                                                           \
  do {                                                                  \
    hipError_t e_ = (x);                                                \
    if (e_ != hipSuccess) {                                             \
      printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
      exit(1);                                                          \
    }                                                                   \
  } while (0)

#define R2(x) x x
#define R4(x) R2(R2(x))
#define R8(x) R2(R4(x))
#define R16(x) R4(R4(x))
#define R32(x) R2(R16(x))
#define R64(x) R4(R16(x))
#define R256(x) R4(R64(x))

#define VCLOB                                                                  \
  "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20",  \
      "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30",     \
      "v31", "v32", "v33", "v34", "v35", "a0", "a1", "a2", "a3", "a4", "a5",    \
      "a6", "a7", "s40", "s41", "s42", "s43", "memory"

// block bodies: NI instructions per BLOCK expansion
#define FMA_DEP R256("v_fma_f32 v10, v10, v30, v31\n")
#define FMA_IND2 R64(R2("v_fma_f32 v10, v10, v30, v31\nv_fma_f32 v11, v11, v30, v31\n"))
#define FMA_IND4                                                                \
  R64("v_fma_f32 v10, v10, v30, v31\nv_fma_f32 v11, v11, v30, v31\n"            \
      "v_fma_f32 v12, v12, v30, v31\nv_fma_f32 v13, v13, v30, v31\n")
#define FMA_IND8                                                                \
  R32("v_fma_f32 v10, v10, v30, v31\nv_fma_f32 v11, v11, v30, v31\n"            \
      "v_fma_f32 v12, v12, v30, v31\nv_fma_f32 v13, v13, v30, v31\n"            \
      "v_fma_f32 v14, v14, v30, v31\nv_fma_f32 v15, v15, v30, v31\n"            \
      "v_fma_f32 v16, v16, v30, v31\nv_fma_f32 v17, v17, v30, v31\n")
#define MUL_DEP R256("v_mul_f32_e32 v10, v30, v10\n")
#define FMAC_IND4                                                               \
  R64("v_fmac_f32_e32 v10, v30, v31\nv_fmac_f32_e32 v11, v30, v31\n"            \
      "v_fmac_f32_e32 v12, v30, v31\nv_fmac_f32_e32 v13, v30, v31\n")
#define FMAAK_IND4                                                              \
  R64("v_fmaak_f32 v10, v30, v10, 0x3e2aaaab\nv_fmaak_f32 v11, v30, v11, 0x3e2aaaab\n" \
      "v_fmaak_f32 v12, v30, v12, 0x3e2aaaab\nv_fmaak_f32 v13, v30, v13, 0x3e2aaaab\n")
#define FMA_SGPR_IND4                                                           \
  R64("v_fma_f32 v10, v10, s40, v31\nv_fma_f32 v11, v11, s40, v31\n"            \
      "v_fma_f32 v12, v12, s40, v31\nv_fma_f32 v13, v13, s40, v31\n")
#define PKFMA_DEP R256("v_pk_fma_f32 v[10:11], v[10:11], v[30:31], v[32:33]\n")
#define PKFMA_IND4                                                              \
  R64("v_pk_fma_f32 v[10:11], v[10:11], v[30:31], v[32:33]\n"                   \
      "v_pk_fma_f32 v[12:13], v[12:13], v[30:31], v[32:33]\n"                   \
      "v_pk_fma_f32 v[14:15], v[14:15], v[30:31], v[32:33]\n"                   \
      "v_pk_fma_f32 v[16:17], v[16:17], v[30:31], v[32:33]\n")
#define PKMUL_IND4                                                              \
  R64("v_pk_mul_f32 v[10:11], v[10:11], v[30:31]\n"                             \
      "v_pk_mul_f32 v[12:13], v[12:13], v[30:31]\n"                             \
      "v_pk_mul_f32 v[14:15], v[14:15], v[30:31]\n"                             \
      "v_pk_mul_f32 v[16:17], v[16:17], v[30:31]\n")
#define PKADD_IND4                                                              \
  R64("v_pk_add_f32 v[10:11], v[10:11], v[30:31]\n"                             \
      "v_pk_add_f32 v[12:13], v[12:13], v[30:31]\n"                             \
      "v_pk_add_f32 v[14:15], v[14:15], v[30:31]\n"                             \
      "v_pk_add_f32 v[16:17], v[16:17], v[30:31]\n")
#define SIN_DEP R256("v_sin_f32_e32 v10, v10\n")
#define SIN_IND4                                                                \
  R64("v_sin_f32_e32 v10, v10\nv_sin_f32_e32 v11, v11\n"                        \
      "v_sin_f32_e32 v12, v12\nv_sin_f32_e32 v13, v13\n")
#define RCP_IND4                                                                \
  R64("v_rcp_f32_e32 v10, v10\nv_rcp_f32_e32 v11, v11\n"                        \
      "v_rcp_f32_e32 v12, v12\nv_rcp_f32_e32 v13, v13\n")
// one transcendental per three plain ops (does the trans pipe overlap?)
#define SIN_MIX                                                                 \
  R64("v_sin_f32_e32 v10, v20\nv_fma_f32 v11, v11, v30, v31\n"                  \
      "v_fma_f32 v12, v12, v30, v31\nv_fma_f32 v13, v13, v30, v31\n")
// sin + cos of one argument followed by four dependent-free plain ops
#define SINCOS_MIX                                                              \
  R32("v_sin_f32_e32 v10, v20\nv_cos_f32_e32 v14, v20\n"                        \
      "v_fma_f32 v11, v11, v30, v31\nv_fma_f32 v12, v12, v30, v31\n"            \
      "v_fma_f32 v13, v13, v30, v31\nv_fma_f32 v15, v15, v30, v31\n"            \
      "v_fma_f32 v16, v16, v30, v31\nv_fma_f32 v17, v17, v30, v31\n")
#define ACCW_IND R32("v_accvgpr_write_b32 a0, v10\nv_accvgpr_write_b32 a1, v11\n" \
      "v_accvgpr_write_b32 a2, v12\nv_accvgpr_write_b32 a3, v13\n"              \
      "v_accvgpr_write_b32 a4, v14\nv_accvgpr_write_b32 a5, v15\n"              \
      "v_accvgpr_write_b32 a6, v16\nv_accvgpr_write_b32 a7, v17\n")
#define ACCR_IND R32("v_accvgpr_read_b32 v10, a0\nv_accvgpr_read_b32 v11, a1\n" \
      "v_accvgpr_read_b32 v12, a2\nv_accvgpr_read_b32 v13, a3\n"                \
      "v_accvgpr_read_b32 v14, a4\nv_accvgpr_read_b32 v15, a5\n"                \
      "v_accvgpr_read_b32 v16, a6\nv_accvgpr_read_b32 v17, a7\n")
// AGPR round trip feeding a plain op (spill-reload shape)
#define ACC_MIX                                                                 \
  R64("v_accvgpr_read_b32 v14, a0\nv_fma_f32 v10, v10, v14, v31\n"              \
      "v_accvgpr_write_b32 a1, v11\nv_fma_f32 v11, v11, v30, v31\n")
#define MOV_IND4                                                                \
  R64("v_mov_b32_e32 v10, v30\nv_mov_b32_e32 v11, v30\n"                        \
      "v_mov_b32_e32 v12, v30\nv_mov_b32_e32 v13, v30\n")
#define FMA_SALU                                                                \
  R64("v_fma_f32 v10, v10, v30, v31\ns_add_u32 s41, s41, 1\n"                   \
      "v_fma_f32 v11, v11, v30, v31\ns_add_u32 s42, s42, 1\n")
#define FMA_SNOP                                                                \
  R64("v_fma_f32 v10, v10, v30, v31\ns_nop 0\n"                                 \
      "v_fma_f32 v11, v11, v30, v31\ns_nop 0\n")
#define CNDMASK_IND4                                                            \
  R64("v_cndmask_b32_e32 v10, v30, v31, vcc\nv_cndmask_b32_e32 v11, v30, v31, vcc\n" \
      "v_cndmask_b32_e32 v12, v30, v31, vcc\nv_cndmask_b32_e32 v13, v30, v31, vcc\n")
#define CNDMASK64_IND4                                                          \
  R64("v_cndmask_b32_e64 v10, v30, v31, s[42:43]\nv_cndmask_b32_e64 v11, v30, v31, s[42:43]\n" \
      "v_cndmask_b32_e64 v12, v30, v31, s[42:43]\nv_cndmask_b32_e64 v13, v30, v31, s[42:43]\n")
// LDS: v20 = lane*4 (set by the kernel)
#define DSW_IND R64("ds_write_b32 v20, v30 offset:0\nds_write_b32 v20, v31 offset:256\n" \
      "ds_write_b32 v20, v30 offset:512\nds_write_b32 v20, v31 offset:768\n")
#define DSR_IND R64("ds_read_b32 v10, v20 offset:0\nds_read_b32 v11, v20 offset:256\n" \
      "ds_read_b32 v12, v20 offset:512\nds_read_b32 v13, v20 offset:768\n")
#define DSR2_IND R64("ds_read2st64_b32 v[10:11], v20 offset0:0 offset1:1\n"     \
      "ds_read2st64_b32 v[12:13], v20 offset0:2 offset1:3\n"                    \
      "ds_read2st64_b32 v[14:15], v20 offset0:4 offset1:5\n"                    \
      "ds_read2st64_b32 v[16:17], v20 offset0:6 offset1:7\n")
#define DSW2_IND R64("ds_write2st64_b32 v20, v30, v31 offset0:0 offset1:1\n"    \
      "ds_write2st64_b32 v20, v30, v31 offset0:2 offset1:3\n"                   \
      "ds_write2st64_b32 v20, v30, v31 offset0:4 offset1:5\n"                   \
      "ds_write2st64_b32 v20, v30, v31 offset0:6 offset1:7\n")
// one LDS read per three plain ops, results never waited for inside the block
#define DSR_MIX                                                                 \
  R64("ds_read_b32 v14, v20 offset:0\nv_fma_f32 v11, v11, v30, v31\n"           \
      "v_fma_f32 v12, v12, v30, v31\nv_fma_f32 v13, v13, v30, v31\n")
#define DSW_MIX                                                                 \
  R64("ds_write_b32 v20, v30 offset:0\nv_fma_f32 v11, v11, v30, v31\n"          \
      "v_fma_f32 v12, v12, v30, v31\nv_fma_f32 v13, v13, v30, v31\n")

enum {
  P_FMA_DEP, P_FMA_IND2, P_FMA_IND4, P_FMA_IND8, P_MUL_DEP, P_FMAC_IND4,
  P_FMAAK_IND4, P_FMA_SGPR_IND4, P_PKFMA_DEP, P_PKFMA_IND4, P_PKMUL_IND4,
  P_PKADD_IND4, P_SIN_DEP, P_SIN_IND4, P_RCP_IND4, P_SIN_MIX, P_SINCOS_MIX,
  P_ACCW, P_ACCR, P_ACC_MIX, P_MOV_IND4, P_FMA_SALU, P_FMA_SNOP, P_CNDMASK,
  P_DSW, P_DSR, P_DSR2, P_DSW2, P_DSR_MIX, P_DSW_MIX, P_EMPTY, P_CNDMASK64, P_COUNT
};
static const char *kNames[P_COUNT] = {
    "fma_dep", "fma_ind2", "fma_ind4", "fma_ind8", "mul_e32_dep", "fmac_ind4",
    "fmaak_literal_ind4", "fma_sgpr_ind4", "pk_fma_dep", "pk_fma_ind4",
    "pk_mul_ind4", "pk_add_ind4", "sin_dep", "sin_ind4", "rcp_ind4",
    "sin_1_per_3_fma", "sincos_2_per_6_fma", "accvgpr_write_ind8",
    "accvgpr_read_ind8", "accvgpr_rw_between_fma", "mov_ind4",
    "fma_salu_alternating", "fma_snop_alternating", "cndmask_ind4",
    "ds_write_b32", "ds_read_b32", "ds_read2st64_b32", "ds_write2st64_b32",
    "ds_read_1_per_3_fma", "ds_write_1_per_3_fma", "empty",
    "cndmask_e64_sgprpair_ind4"};

template <int P>
__global__ __launch_bounds__(64) void probe(unsigned long long *out, int reps,
                                            int half_exec) {
  __shared__ float lds[2048];
  lds[threadIdx.x] = 1.f;
  unsigned long long t0, t1;
  const int lane4 = threadIdx.x * 4;
  asm volatile("v_mov_b32 v20, %0\n"
               "v_mov_b32 v30, 0x3f7ff000\nv_mov_b32 v31, 0x33000000\n"
               "v_mov_b32 v32, 0x3f7ff000\nv_mov_b32 v33, 0x33000000\n"
               "v_mov_b32 v10, 1.0\nv_mov_b32 v11, 1.0\nv_mov_b32 v12, 1.0\n"
               "v_mov_b32 v13, 1.0\nv_mov_b32 v14, 1.0\nv_mov_b32 v15, 1.0\n"
               "v_mov_b32 v16, 1.0\nv_mov_b32 v17, 1.0\ns_mov_b32 s40, 0x3f7ff000\n"
               "s_mov_b32 s41, 0\ns_mov_b32 s42, 0\n" ::"v"(lane4)
               : VCLOB);
  if (half_exec) asm volatile("s_mov_b64 exec, 0xffffffff" ::: "memory");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\ns_memtime %0\ns_waitcnt lgkmcnt(0)"
               : "=s"(t0)::"memory");
  for (int r = 0; r < reps; ++r) {
    if constexpr (P == P_FMA_DEP) asm volatile(FMA_DEP ::: VCLOB);
    if constexpr (P == P_FMA_IND2) asm volatile(FMA_IND2 ::: VCLOB);
    if constexpr (P == P_FMA_IND4) asm volatile(FMA_IND4 ::: VCLOB);
    if constexpr (P == P_FMA_IND8) asm volatile(FMA_IND8 ::: VCLOB);
    if constexpr (P == P_MUL_DEP) asm volatile(MUL_DEP ::: VCLOB);
    if constexpr (P == P_FMAC_IND4) asm volatile(FMAC_IND4 ::: VCLOB);
    if constexpr (P == P_FMAAK_IND4) asm volatile(FMAAK_IND4 ::: VCLOB);
    if constexpr (P == P_FMA_SGPR_IND4) asm volatile(FMA_SGPR_IND4 ::: VCLOB);
    if constexpr (P == P_PKFMA_DEP) asm volatile(PKFMA_DEP ::: VCLOB);
    if constexpr (P == P_PKFMA_IND4) asm volatile(PKFMA_IND4 ::: VCLOB);
    if constexpr (P == P_PKMUL_IND4) asm volatile(PKMUL_IND4 ::: VCLOB);
    if constexpr (P == P_PKADD_IND4) asm volatile(PKADD_IND4 ::: VCLOB);
    if constexpr (P == P_SIN_DEP) asm volatile(SIN_DEP ::: VCLOB);
    if constexpr (P == P_SIN_IND4) asm volatile(SIN_IND4 ::: VCLOB);
    if constexpr (P == P_RCP_IND4) asm volatile(RCP_IND4 ::: VCLOB);
    if constexpr (P == P_SIN_MIX) asm volatile(SIN_MIX ::: VCLOB);
    if constexpr (P == P_SINCOS_MIX) asm volatile(SINCOS_MIX ::: VCLOB);
    if constexpr (P == P_ACCW) asm volatile(ACCW_IND ::: VCLOB);
    if constexpr (P == P_ACCR) asm volatile(ACCR_IND ::: VCLOB);
    if constexpr (P == P_ACC_MIX) asm volatile(ACC_MIX ::: VCLOB);
    if constexpr (P == P_MOV_IND4) asm volatile(MOV_IND4 ::: VCLOB);
    if constexpr (P == P_FMA_SALU) asm volatile(FMA_SALU ::: VCLOB);
    if constexpr (P == P_FMA_SNOP) asm volatile(FMA_SNOP ::: VCLOB);
    if constexpr (P == P_CNDMASK) asm volatile(CNDMASK_IND4 ::: VCLOB);
    if constexpr (P == P_DSW) asm volatile(DSW_IND "s_waitcnt lgkmcnt(0)\n" ::: VCLOB);
    if constexpr (P == P_DSR) asm volatile(DSR_IND "s_waitcnt lgkmcnt(0)\n" ::: VCLOB);
    if constexpr (P == P_DSR2) asm volatile(DSR2_IND "s_waitcnt lgkmcnt(0)\n" ::: VCLOB);
    if constexpr (P == P_DSW2) asm volatile(DSW2_IND "s_waitcnt lgkmcnt(0)\n" ::: VCLOB);
    if constexpr (P == P_DSR_MIX) asm volatile(DSR_MIX "s_waitcnt lgkmcnt(0)\n" ::: VCLOB);
    if constexpr (P == P_DSW_MIX) asm volatile(DSW_MIX "s_waitcnt lgkmcnt(0)\n" ::: VCLOB);
    if constexpr (P == P_EMPTY) asm volatile("" ::: VCLOB);
    if constexpr (P == P_CNDMASK64) asm volatile(CNDMASK64_IND4 ::: VCLOB);
  }
  asm volatile("s_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (half_exec) asm volatile("s_mov_b64 exec, -1" ::: "memory");
  float sink;
  asm volatile("v_add_f32 %0, v10, v11\n" : "=v"(sink)::"memory");
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (sink == 123.456f) out[blockIdx.x] = 0;  // keep the block alive
  if (lds[(threadIdx.x + 1) & 63] == 77.f) out[blockIdx.x] = 1;
}

// ---- VMEM issue cost: N back-to-back operations, stamps after issue and
// after completion -------------------------------------------------------------
enum { M_LOAD, M_LOAD_LDS, M_LOAD_LDS_X4, M_STORE, M_STORE_NT, M_STORE_X4_NT,
       M_LOAD_X2, M_LOAD_X3, M_LOAD_X4, M_LOAD_X4_S24, M_LOAD_X2_S24, M_COUNT };
static const char *kMemNames[M_COUNT] = {
    "buffer_load_dword", "buffer_load_dword_lds", "buffer_load_dwordx4_lds",
    "buffer_store_dword", "buffer_store_dword_nt", "buffer_store_dwordx4_nt",
    "buffer_load_dwordx2", "buffer_load_dwordx3", "buffer_load_dwordx4",
    "buffer_load_dwordx4_stride24", "buffer_load_dwordx2_stride24"};

template <int M, int N>
__global__ __launch_bounds__(64) void mem_probe(const float *in, float *outbuf,
                                                int planes, int B,
                                                unsigned long long *stamps) {
  __shared__ __attribute__((aligned(16))) float lds[N * 256];
  typedef __attribute__((address_space(3))) void *lds_ptr;
  const int b = blockIdx.x * 64 + threadIdx.x;
  const auto ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0,
                                                    planes * B * 4, 0x00020000);
  const auto ro = __builtin_amdgcn_make_buffer_rsrc(outbuf, 0, planes * B * 4,
                                                    0x00020000);
  unsigned long long t0, t1, t2;
  float v[N];
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = (float)i;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\ns_memtime %0\ns_waitcnt lgkmcnt(0)"
               : "=s"(t0)::"memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if constexpr (M == M_LOAD)
      v[i] = __builtin_bit_cast(
          float, __builtin_amdgcn_raw_buffer_load_b32(ri, b * 4, i * B * 4, 0));
    if constexpr (M == M_LOAD_LDS)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ri, (lds_ptr)(lds + i * 64), 4,
                                               b * 4, i * B * 4, 0, 0);
    if constexpr (M == M_LOAD_LDS_X4)  // 1 KiB per instruction: 256 trajectories
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          ri, (lds_ptr)(lds + i * 256), 16, threadIdx.x * 16,
          (i * B + (blockIdx.x & ~3) * 64) * 4, 0, 0);
    // wide per-lane loads into VGPRs: lane b reads W dwords at b * stride
    if constexpr (M == M_LOAD_X2) {
      typedef unsigned u2 __attribute__((ext_vector_type(2)));
      u2 r = __builtin_amdgcn_raw_buffer_load_b64(ri, b * 8, i * B * 8, 0);
      v[i] = __builtin_bit_cast(float, r.x) + __builtin_bit_cast(float, r.y);
    }
    if constexpr (M == M_LOAD_X3) {
      typedef unsigned u3 __attribute__((ext_vector_type(3)));
      u3 r = __builtin_amdgcn_raw_buffer_load_b96(ri, b * 12, i * B * 12, 0);
      v[i] = __builtin_bit_cast(float, r.x) + __builtin_bit_cast(float, r.z);
    }
    if constexpr (M == M_LOAD_X4) {
      u4 r = __builtin_amdgcn_raw_buffer_load_b128(ri, b * 16, i * B * 16, 0);
      v[i] = __builtin_bit_cast(float, r.x) + __builtin_bit_cast(float, r.w);
    }
    if constexpr (M == M_LOAD_X4_S24) {
      u4 r = __builtin_amdgcn_raw_buffer_load_b128(ri, b * 24, i * B * 24, 0);
      v[i] = __builtin_bit_cast(float, r.x) + __builtin_bit_cast(float, r.w);
    }
    if constexpr (M == M_LOAD_X2_S24) {
      typedef unsigned u2 __attribute__((ext_vector_type(2)));
      u2 r = __builtin_amdgcn_raw_buffer_load_b64(ri, b * 24 + 16, i * B * 24, 0);
      v[i] = __builtin_bit_cast(float, r.x) + __builtin_bit_cast(float, r.y);
    }
    if constexpr (M == M_STORE)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[i]), ro,
                                            b * 4, i * B * 4, 0);
    if constexpr (M == M_STORE_NT)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[i]), ro,
                                            b * 4, i * B * 4, 2);
    if constexpr (M == M_STORE_X4_NT)
      __builtin_amdgcn_raw_buffer_store_b128(
          (u4){__builtin_bit_cast(unsigned, v[i]), 1u, 2u, 3u}, ro,
          threadIdx.x * 16, (i * B + blockIdx.x * 256) * 4 % (planes * B * 4 - 1024), 2);
  }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  asm volatile("s_waitcnt vmcnt(0)\ns_memtime %0\ns_waitcnt lgkmcnt(0)"
               : "=s"(t2)::"memory");
  float acc = 0.f;
  if constexpr (M == M_LOAD || M >= M_LOAD_X2) {
#pragma unroll
    for (int i = 0; i < N; ++i) acc += v[i];
  }
  if constexpr (M == M_LOAD_LDS || M == M_LOAD_LDS_X4) acc = lds[threadIdx.x];
  if (acc == 123.456f) outbuf[b] = acc;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t1 - t0;
    stamps[2 * blockIdx.x + 1] = t2 - t0;
  }
}

static unsigned long long median(std::vector<unsigned long long> v) {
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

template <int P>
void run(unsigned long long *d_out, int waves_per_simd, int half_exec, int ninstr,
         int reps = 4) {
  const int grid = 1024 * waves_per_simd;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(probe<P>, dim3(grid), dim3(64), 0, 0, d_out, reps, half_exec);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(probe<P>, dim3(grid), dim3(64), 0, 0, d_out, reps, half_exec);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(grid);
  CK(hipMemcpy(h.data(), d_out, grid * 8, hipMemcpyDeviceToHost));
  const double cyc = (double)median(h);
  printf("{\"probe\": \"%s\", \"waves_per_simd\": %d, \"half_exec\": %d, "
         "\"instr\": %d, \"ticks_per_wave\": %.0f, \"ticks_per_instr\": %.3f, "
         "\"kernel_us\": %.2f}\n",
         kNames[P], waves_per_simd, half_exec, ninstr * reps, cyc,
         cyc / (ninstr * reps), ms * 1e3);
  fflush(stdout);
}

template <int M, int N>
void run_mem(const float *in, float *out, int planes, int B,
             unsigned long long *d_st) {
  const int grid = B / 64;
  for (int it = 0; it < 2; ++it) {
    hipLaunchKernelGGL((mem_probe<M, N>), dim3(grid), dim3(64), 0, 0, in, out,
                       planes, B, d_st);
    CK(hipDeviceSynchronize());
  }
  std::vector<unsigned long long> h(2 * grid);
  CK(hipMemcpy(h.data(), d_st, 2 * grid * 8, hipMemcpyDeviceToHost));
  std::vector<unsigned long long> a(grid), b(grid);
  for (int i = 0; i < grid; ++i) a[i] = h[2 * i], b[i] = h[2 * i + 1];
  printf("{\"mem_probe\": \"%s\", \"n\": %d, \"issue_ticks\": %llu, "
         "\"issue_ticks_per_op\": %.2f, \"complete_ticks\": %llu}\n",
         kMemNames[M], N, median(a), (double)median(a) / N, median(b));
  fflush(stdout);
}

int main() {
  unsigned long long *d_out;
  CK(hipMalloc(&d_out, 8192 * 8 * 2));
  // clock reference: s_memtime ticks against wall time on a long dependent chain
#define ALL(W, HX)                                                             \
  run<P_EMPTY>(d_out, W, HX, 1);                                               \
  run<P_FMA_DEP>(d_out, W, HX, 256);                                           \
  run<P_FMA_IND2>(d_out, W, HX, 256);                                          \
  run<P_FMA_IND4>(d_out, W, HX, 256);                                          \
  run<P_FMA_IND8>(d_out, W, HX, 256);                                          \
  run<P_MUL_DEP>(d_out, W, HX, 256);                                           \
  run<P_FMAC_IND4>(d_out, W, HX, 256);                                         \
  run<P_FMAAK_IND4>(d_out, W, HX, 256);                                        \
  run<P_FMA_SGPR_IND4>(d_out, W, HX, 256);                                     \
  run<P_PKFMA_DEP>(d_out, W, HX, 256);                                         \
  run<P_PKFMA_IND4>(d_out, W, HX, 256);                                        \
  run<P_PKMUL_IND4>(d_out, W, HX, 256);                                        \
  run<P_PKADD_IND4>(d_out, W, HX, 256);                                        \
  run<P_SIN_DEP>(d_out, W, HX, 256);                                           \
  run<P_SIN_IND4>(d_out, W, HX, 256);                                          \
  run<P_RCP_IND4>(d_out, W, HX, 256);                                          \
  run<P_SIN_MIX>(d_out, W, HX, 256);                                           \
  run<P_SINCOS_MIX>(d_out, W, HX, 256);                                        \
  run<P_ACCW>(d_out, W, HX, 256);                                              \
  run<P_ACCR>(d_out, W, HX, 256);                                              \
  run<P_ACC_MIX>(d_out, W, HX, 256);                                           \
  run<P_MOV_IND4>(d_out, W, HX, 256);                                          \
  run<P_FMA_SALU>(d_out, W, HX, 256);                                          \
  run<P_FMA_SNOP>(d_out, W, HX, 256);                                          \
  run<P_CNDMASK>(d_out, W, HX, 256);                                           \
  run<P_DSW>(d_out, W, HX, 256);                                               \
  run<P_DSR>(d_out, W, HX, 256);                                               \
  run<P_DSR2>(d_out, W, HX, 256);                                              \
  run<P_DSW2>(d_out, W, HX, 256);                                              \
  run<P_DSR_MIX>(d_out, W, HX, 256);                                           \
  run<P_DSW_MIX>(d_out, W, HX, 256);
  ALL(1, 0)
  // long runs (reps = 128: ~0.1 ms) so that the dispatch ramp of 2 048 / 4 096
  // workgroups no longer decides how many waves really share a SIMD
  for (int w = 1; w <= 4; w *= 2) {
    run<P_FMA_DEP>(d_out, w, 0, 256, 128);
    run<P_FMA_IND4>(d_out, w, 0, 256, 128);
    run<P_PKFMA_IND4>(d_out, w, 0, 256, 128);
    run<P_SIN_IND4>(d_out, w, 0, 256, 128);
    run<P_SINCOS_MIX>(d_out, w, 0, 256, 128);
    run<P_FMA_SALU>(d_out, w, 0, 256, 128);
    run<P_CNDMASK>(d_out, w, 0, 256, 128);
    run<P_CNDMASK64>(d_out, w, 0, 256, 128);
    run<P_DSR_MIX>(d_out, w, 0, 256, 128);
    // which instruction classes of a second wave issue in the first wave's gaps
    run<P_FMA_SGPR_IND4>(d_out, w, 0, 256, 128);
    run<P_FMAAK_IND4>(d_out, w, 0, 256, 128);
    run<P_FMAC_IND4>(d_out, w, 0, 256, 128);
    run<P_MUL_DEP>(d_out, w, 0, 256, 128);
    run<P_MOV_IND4>(d_out, w, 0, 256, 128);
    run<P_SIN_MIX>(d_out, w, 0, 256, 128);
    run<P_ACC_MIX>(d_out, w, 0, 256, 128);
  }
  run<P_FMA_DEP>(d_out, 1, 1, 256);
  run<P_FMA_IND4>(d_out, 1, 1, 256);
  run<P_FMA_IND4>(d_out, 2, 1, 256);

  const int B = 65536, planes = 64;
  float *in, *out;
  CK(hipMalloc(&in, (size_t)planes * B * 4));
  CK(hipMalloc(&out, (size_t)planes * B * 4));
  CK(hipMemset(in, 0, (size_t)planes * B * 4));
  run_mem<M_LOAD, 8>(in, out, planes, B, d_out);
  run_mem<M_LOAD, 24>(in, out, planes, B, d_out);
  run_mem<M_LOAD, 48>(in, out, planes, B, d_out);
  run_mem<M_LOAD_LDS, 8>(in, out, planes, B, d_out);
  run_mem<M_LOAD_LDS, 24>(in, out, planes, B, d_out);
  run_mem<M_LOAD_LDS, 48>(in, out, planes, B, d_out);
  run_mem<M_LOAD_LDS_X4, 6>(in, out, planes, B, d_out);
  run_mem<M_LOAD_LDS_X4, 12>(in, out, planes, B, d_out);
  run_mem<M_STORE, 8>(in, out, planes, B, d_out);
  run_mem<M_STORE, 40>(in, out, planes, B, d_out);
  run_mem<M_STORE_NT, 8>(in, out, planes, B, d_out);
  run_mem<M_STORE_NT, 40>(in, out, planes, B, d_out);
  run_mem<M_STORE_X4_NT, 10>(in, out, planes, B, d_out);
  run_mem<M_LOAD_X2, 8>(in, out, planes, B, d_out);
  run_mem<M_LOAD_X2, 24>(in, out, planes, B, d_out);
  run_mem<M_LOAD_X3, 8>(in, out, planes, B, d_out);
  run_mem<M_LOAD_X3, 16>(in, out, planes, B, d_out);
  run_mem<M_LOAD_X4, 4>(in, out, planes, B, d_out);
  run_mem<M_LOAD_X4, 8>(in, out, planes, B, d_out);
  run_mem<M_LOAD_X4, 12>(in, out, planes, B, d_out);
  run_mem<M_LOAD_X4_S24, 8>(in, out, planes, B, d_out);
  run_mem<M_LOAD_X2_S24, 8>(in, out, planes, B, d_out);
  return 0;
}
