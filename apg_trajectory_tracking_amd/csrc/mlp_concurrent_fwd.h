// mlp_concurrent_fwd.h - the forward kernel of the concurrent training step
// (policy once per trajectory, rollout + quad_mpc_loss + adjoint); shared by
// mlp_concurrent.hip (the product's step) and mlp_planes.hip (the test library's
// plane sequence, apg_quad_mlp_concurrent_fwd_bwd).
#pragma once
#include "mlp_common.h"

namespace apg {
namespace {
// ---------------------------------------------- concurrent mode, policy fused
// The concurrent training step (BASELINE config 2, the headline workload) with
// the policy inside: TrainBase.run_epoch's concurrent branch
// (scripts/train_base.py:198-204: actions = sigmoid(net(in_state, in_ref)),
// reshape [B, H, 4]) + TrainDrone.train_controller_model
// (scripts/train_drone.py:175-203: H x dynamics, quad_mpc_loss, backward).
// The network runs ONCE per trajectory (40 outputs = H x 4 actions), then the
// register-resident rollout and its adjoint (as quad.hip), then - second
// kernel - the reverse pass of the network from dL/d(head pre-activations).
// Planes are [feature][B]; the weight gradients come from apg_planes_gemm.


struct ConcArgs {
  const float *feat, *in_ref, *state0, *ref;  // [15][B], [H][9][B], [12][B], [H][C][B]
  float *x1, *h;        // [224][B], [192][B]
  unsigned *mask;       // [5][B]
  float *d_zout;        // [40][B]
  float *d_pre, *d_conv;  // [256][B], [160][B] (second kernel)
  float *states;        // [H][12][B] or NULL
  float *loss_partials;
  const float *tables;
  // [waves][4] or NULL: max |conv output|, |feature|, |in_ref| of each wave's 32
  // trajectories (the trajectory-major reverse kernel scales by them)
  float *xmax;
  QuadConst c;
  ApgQuadLossWeights w;
  int B, ref_cols, vel_col;
  // ROWS (below): feat / in_ref / state0 / ref are the DATA SET's tensors [N][ld_*]
  // and `index` [B] names this batch's rows; the feature and window planes the
  // reverse kernel reads are written to o_feat [15][B] / o_in_ref [90][B]
  const long long *index;
  float *o_feat, *o_in_ref;
  int ld_feat, ld_in_ref, ld_state0, ld_ref;
  int win16;   // the window rows start on 16-byte boundaries (gather_windows16_issue)
  unsigned bytes_feat, bytes_in_ref, bytes_state0, bytes_ref;
};

// ROWS: the minibatch gather folded into this kernel (VERDICT r4 next #4;
// TrainBase.run_epoch's batch selection, scripts/train_base.py:191-194).  The
// workgroup's rows are brought into LDS through the index before the operand
// tables are complete:
//   windows  [256][92]  at the END of the table region (zWin): 23 chunks of 16
//            bytes per row, lane-linear over (row, chunk) - 3 rows per
//            instruction, i.e. whole cache lines (`win16`: row pitch and base
//            are multiples of 16 bytes; else the same layout by dwords)
//   features [256][15], start states [256][13], row numbers: behind the tables
//   the first 32 KB of the tables (zWin floats: the small tables, states_in, conv,
//            ten fc1 blocks) in the same batch of requests; the rest once every
//            wave holds its rows in registers.
// Round 6 (profiles/r06_rows_noshuffle.txt: 8 of the 13 us this kernel took over
// the plane-reading one were its prologue, not DRAM): before, the windows were
// staged under the WHOLE table region by 364 dword requests per workgroup and
// read back with 4-way bank conflicts (row stride 91 was odd, but the half-waves'
// columns 4 apart were not), and all 127 KB of tables followed the gather.
// The reference rows [256][91] land over the tables once the policy is done with
// them, while the rollout runs.
constexpr int kRowPadW = kH * kRD + 1, kRowPadF = kNF, kRowPadS = 13;   // odd strides
constexpr int kWinChunks = (kH * kRD + 3) / 4, kWinRow = 4 * kWinChunks;  // 23, 92
constexpr int zRef = 0, zWin = kCfLds - kTrajPerBlock * kWinRow,        // floats
              zS0 = kCfLds, zRows = zS0 + kTrajPerBlock * kRowPadS,
              zFeat = zRows + kTrajPerBlock,
              kCfRowsLds = zFeat + kTrajPerBlock * kRowPadF;
static_assert(zWin >= hA / 4 + 4 * kBlock16 / 4 && zWin % 256 == 0,
              "states_in and conv blocks ahead of the window staging");
static_assert(kTrajPerBlock * kRowPadW <= kCfLds, "reference rows over the tables");
static_assert(kCfRowsLds * 4 <= 160 * 1024, "LDS");

// the windows of the workgroup's 256 rows, 16 bytes per lane: element e = 64 n +
// lane of [256][23] is chunk e % 23 of row e / 23
__device__ __forceinline__ void gather_windows16_issue(float *dst, const int *rows,
                                                       const float *base, unsigned bytes,
                                                       int ld) {
  typedef __attribute__((address_space(3))) void *lds_ptr_t;
  const auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)bytes,
                                                   0x00020000);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int waves = blockDim.x >> 6;
  for (int n = wave; n < kTrajPerBlock * kWinChunks / 64; n += waves) {
    const int e = n * 64 + lane, t = e / kWinChunks, c = e - t * kWinChunks;
    // (the last chunk of a row ends 2 floats past the window: the row's own next
    // columns, or - last row of a data set whose rows are exactly the window -
    // out of range: zeros)
    const unsigned voff = ((unsigned)rows[t] * (unsigned)ld + 4u * (unsigned)c) * 4u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(dst + n * 256), 16, (int)voff, 0,
                                             0, 0);
  }
}

// APG_CF_KNOCKOUT (experiment builds, tools/build_policy_variant.sh; results are
// WRONG on purpose): 1 row numbers = batch positions (no index round trip), 2 no
// wait for the second part of the tables, 4 no reference-row gather, 8 no row
// gathers at all, 16 every workgroup reads rows 0..7 (same requests, eight rows'
// cache lines)
#ifndef APG_CF_KNOCKOUT
#define APG_CF_KNOCKOUT 0
#endif
template <bool ROWS>
__global__ __launch_bounds__(kThreads) void mlp_concurrent_fwd_kernel(ConcArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (ROWS) {
    const int t = threadIdx.x, b_ = blockIdx.x * kTrajPerBlock + t;
    // (a dead trajectory reads the batch's last row: finite data, never stored)
    if (t < kTrajPerBlock)
      reinterpret_cast<int *>(lds + zRows)[t] =
          (APG_CF_KNOCKOUT & 16) ? (t & 7)
          : (APG_CF_KNOCKOUT & 1) ? (b_ < A.B ? b_ : A.B - 1) : (int)A.index[b_ < A.B ? b_ : A.B - 1];
    __syncthreads();
    const int *rows = reinterpret_cast<const int *>(lds + zRows);
    if (!(APG_CF_KNOCKOUT & 8)) {
    if (A.win16)
      gather_windows16_issue(lds + zWin, rows, A.in_ref, A.bytes_in_ref, A.ld_in_ref);
    else
      gather_rows_issue<kWinRow>(lds + zWin, rows, A.in_ref, A.bytes_in_ref, A.ld_in_ref,
                                 kH * kRD);
    gather_rows_issue<kRowPadF>(lds + zFeat, rows, A.feat, A.bytes_feat, A.ld_feat, kNF);
    gather_rows_issue<kRowPadS>(lds + zS0, rows, A.state0, A.bytes_state0, A.ld_state0, 12);
    }
    fill_lds_issue(lds, A.tables, zWin);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  } else {
    fill_lds(lds, A.tables, kCfLds);
  }
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const LdsView L(lds, lane);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = (blockIdx.x * (kThreads / 64) + wave) * 32 + (lane & 31);
  const int B = A.B;
  const bool live = b < B;
  const bool st_lo = live && hi == 0;
  const unsigned pN = (unsigned)B * 4u;  // every plane here is [..][B]
  const QuadConst c = A.c;
  const Planes Pfe(A.feat, kNF, pN), Pin(A.in_ref, kH * kRD, pN);
  const Planes Ps0(A.state0, 12, pN), Prf(A.ref, kH * A.ref_cols, pN);
  const Planes Px1(A.x1, kN1, pN), Ph(A.h, 3 * kW, pN), Pmk(A.mask, 5, pN);
  const Planes Pdz(A.d_zout, kNA, pN);
  const Planes Pst(A.states, A.states ? kH * 12 : 0, pN);
  const unsigned vb = live ? (unsigned)b * 4u : kDead;
  const unsigned vb_lo = st_lo ? vb : kDead;
  const unsigned vr = live ? vb + (hi ? 4u * pN : 0u) : kDead;   // + row 4 hi
  const unsigned vc = live ? vb + (hi ? 32u * pN : 0u) : kDead;  // + channel 4 hi
  const unsigned vm = live ? vb + (hi ? pN : 0u) : kDead;        // + mask word hi

  float feat[kNF];
  float w[kH][5];  // policy reference input, columns 0..4 / 4..8 per half
  const int tl = wave * 32 + (lane & 31);   // this lane's trajectory of the workgroup
  if (ROWS) {
    const float *pf = lds + zFeat + tl * kRowPadF;
#pragma unroll
    for (int j = 0; j < kNF; ++j) feat[j] = pf[j];
    // the lane's whole window row by 16-byte reads (row stride 23 chunks: 16
    // consecutive rows start in 16 different bank quads), its half's columns
    // picked in registers
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 *pw = reinterpret_cast<const f32x4 *>(lds + zWin + tl * kWinRow);
    float wr[kWinRow];
#pragma unroll
    for (int cq = 0; cq < kWinChunks; ++cq) {
      const f32x4 q = pw[cq];
#pragma unroll
      for (int i = 0; i < 4; ++i) wr[4 * cq + i] = q[i];
    }
#pragma unroll
    for (int r = 0; r < kH; ++r)
#pragma unroll
      for (int j = 0; j < 5; ++j) w[r][j] = hi ? wr[r * kRD + 4 + j] : wr[r * kRD + j];
    __syncthreads();                      // every wave has its rows: the tables may land
    fill_lds_issue(lds + zWin, A.tables + zWin, kCfLds - zWin);
  } else {
#pragma unroll
    for (int j = 0; j < kNF; ++j) feat[j] = Pfe.ld(vb, j * pN);
#pragma unroll
    for (int r = 0; r < kH; ++r)
#pragma unroll
      for (int j = 0; j < 5; ++j) w[r][j] = Pin.ld(vr, (r * kRD + j) * pN);
  }
  // (maxima of the |v| BIT PATTERNS, unsigned: inf / NaN lie above every finite
  // value - see TmMeta)
  unsigned xm_conv = 0u;
  const auto umax = [](unsigned m, float v) {
    const unsigned b_ = __builtin_bit_cast(unsigned, v) & 0x7fffffffu;
    return b_ > m ? b_ : m;
  };
  if (A.xmax) {   // (wave-uniform)
    unsigned mf = 0u, mi = 0u;
#pragma unroll
    for (int j = 0; j < kNF; ++j) mf = umax(mf, feat[j]);
#pragma unroll
    for (int r = 0; r < kH; ++r)
#pragma unroll
      for (int j = 0; j < 5; ++j) mi = umax(mi, w[r][j]);
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
      const unsigned of = (unsigned)__shfl_xor((int)mf, sft, 64),
                     oi = (unsigned)__shfl_xor((int)mi, sft, 64);
      mf = of > mf ? of : mf, mi = oi > mi ? oi : mi;
    }
    if (lane == 0) {
      unsigned *q = reinterpret_cast<unsigned *>(A.xmax) +
                    (size_t)(blockIdx.x * (kThreads / 64) + wave) * 4;
      q[1] = mf, q[2] = mi;
    }
  }

  if (ROWS && !(APG_CF_KNOCKOUT & 2)) {   // the table DMA issued above
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // (the reverse kernel reads the feature / window blocks of x^T from the data
    // set's rows itself: no planes of them are written)
  }
  // ---- policy forward on the 16-bit matrix pipe (policy_mfma16.h): every
  // operand as two fp16 terms, three products per k-block
  const LdsView16 L16(lds, lane);
  f32x16 u[2], a[2];
  init_bias(u, L, hTbs);
  {  // states_in: one k-block, features 8 hi .. 8 hi + 7
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = hi ? (8 + j < kNF ? feat[8 + j < kNF ? 8 + j : 0] : 0.f) : feat[j];
    const Op16 x = split8(v);
    u[0] = mma3(L16.A(hA, nS + 0), x, u[0]);
    u[1] = mma3(L16.A(hA, nS + 1), x, u[1]);
  }
  init_bias(a, L, hTb1);
  unsigned mbits[3] = {0u, 0u, 0u};
#pragma unroll
  for (int pp = 0; pp < kNP / 2; ++pp) {
    float rv[24];  // relu(conv) of positions 2 pp, 2 pp + 1: registers 0..11 each
    // window rows 2 pp .. 2 pp + 3, split: high term in the low half-word, low
    // term in the high half-word of one register per value
    unsigned ws[4][5];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const float xv = w[2 * pp + r][j];
        const _Float16 vh = (_Float16)xv, vl = (_Float16)(xv - (float)vh);
        const h16x2 pr = {vh, vl};
        ws[r][j] = __builtin_bit_cast(unsigned, pr);
      }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int pos = 2 * pp + e;
      f32x16 cv;
#pragma unroll
      for (int i = 0; i < 16; ++i) cv[i] = L.T(hTbc + i * 2);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        Op16 x;
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // slots 2 q, 2 q + 1 of this k-block
          const int s0 = kb * 8 + 2 * q, s1 = s0 + 1;
          const unsigned r0 = ws[e + s0 % 3][s0 / 3];
          const unsigned r1 = s1 < 15 ? ws[e + s1 % 3][s1 < 15 ? s1 / 3 : 0] : 0u;
          x.h[q] = __builtin_amdgcn_perm(r1, r0, 0x05040100u);  // low half-words
          x.l[q] = __builtin_amdgcn_perm(r1, r0, 0x07060302u);  // high half-words
        }
        cv = mma3(L16.A(hA, nC + kb), x, cv);
      }
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        float v = cv[i];
        mbits[i >> 2] |= (v > 0.f ? 1u : 0u) << ((i & 3) * 8 + pos);
        xm_conv = umax(xm_conv, v);   // (before the relu: it would drop a NaN)
        v = relu1(v);
        Px1.st(i < 8 ? vc : vb_lo, (kW + rrow(i) * kNP + pos) * pN, v);
        rv[e * 12 + i] = v;
      }
    }
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = rv[kb * 8 + j];
      const Op16 x = split8(v);
      a[0] = mma3(L16.A(hA, n1c + (0 * 4 + pp) * 3 + kb), x, a[0]);
      a[1] = mma3(L16.A(hA, n1c + (1 * 4 + pp) * 3 + kb), x, a[1]);
    }
  }
#pragma unroll
  for (int g = 0; g < 3; ++g) Pmk.stu(g < 2 ? vm : vb_lo, 2 * g * pN, mbits[g]);
  if (A.xmax) {
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
      const unsigned o = (unsigned)__shfl_xor((int)xm_conv, sft, 64);
      xm_conv = o > xm_conv ? o : xm_conv;
    }
    if (lane == 0)
      reinterpret_cast<unsigned *>(A.xmax)[(size_t)(blockIdx.x * (kThreads / 64) + wave) * 4] =
          xm_conv;
  }
  // fc1 state part on s1 = tanh(states_in), stored as the reverse pass needs it
  dense64_16(a, u, L16, hA, n1s, [&](int rb, int i, float v) {
    const float tv = tanh_fast(v);
    Px1.st(vr, (rb * 32 + rrow(i)) * pN, tv);
    return tv;
  });
  init_bias(u, L, hTb2);
  dense64_16(u, a, L16, hA, n2, [&](int rb, int i, float v) {
    const float tv = tanh_fast(v);
    Ph.st(vr, (rb * 32 + rrow(i)) * pN, tv);
    return tv;
  });
  init_bias(a, L, hTb3);
  dense64_16(a, u, L16, hA, n3, [&](int rb, int i, float v) {
    const float tv = tanh_fast(v);
    Ph.st(vr, (kW + rb * 32 + rrow(i)) * pN, tv);
    return tv;
  });
  // head: 40 outputs = row block 0 + rows 0..7 of row block 1
  init_bias(u, L, hTbo);
  dense64_16(u, a, L16, hA, nO, [&](int rb, int i, float v) {
    const float tv = tanh_fast(v);
    Ph.st(vr, (2 * kW + rb * 32 + rrow(i)) * pN, tv);
    return tv;
  });
  // every lane needs all 40 actions (both halves run the same rollout)
  float act[kH][4];
#pragma unroll
  for (int cc = 0; cc < 20; ++cc) {
    const float own = cc < 16 ? u[0][cc] : u[1][cc - 16], oth = other_half(own);
    const int row = khead(cc, 0);
    act[row >> 2][row & 3] = sigmoidf_(hi ? oth : own);
    act[(row + 4) >> 2][(row + 4) & 3] = sigmoidf_(hi ? own : oth);
  }

  // ---- rollout + adjoint in registers (quad_rollout_reg_kernel's structure)
  float s[12];
  if (ROWS) {
    // the tables are dead: the reference rows land over them while the rollout runs
    __syncthreads();
    if (!(APG_CF_KNOCKOUT & (4 | 8)))
    gather_rows_issue<kRowPadW>(lds + zRef, reinterpret_cast<const int *>(lds + zRows), A.ref,
                                A.bytes_ref, A.ld_ref, kH * A.ref_cols);
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = lds[zS0 + tl * kRowPadS + i];
  } else {
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = Ps0.ld(vb, i * pN);
  }
  Trig st_trig[kH];
  float st_w[kH + 1][3], st_pv[kH][6];
#pragma unroll
  for (int k = 0; k < kH; ++k) {
#pragma unroll
    for (int i = 0; i < 3; ++i) st_w[k][i] = s[9 + i];
    st_trig[k] = make_trig(&s[3]);
    quad_step(s, act[k], c, st_trig[k]);
#pragma unroll
    for (int i = 0; i < 3; ++i) st_pv[k][i] = s[i], st_pv[k][3 + i] = s[6 + i];
#pragma unroll
    for (int i = 0; i < 12; ++i) Pst.st(vb_lo, (k * 12 + i) * pN, s[i]);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) st_w[kH][i] = s[9 + i];
  float loss = 0.f, lam[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) lam[i] = 0.f;
  float rp[3], rv[3];
  const float *pr = lds + zRef + tl * kRowPadW;
  if (ROWS) {   // the reference rows (and every store so far) have landed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    rp[i] = ROWS ? pr[(kH - 1) * A.ref_cols + i]
                 : Prf.ld(vb, ((kH - 1) * A.ref_cols + i) * pN);
    rv[i] = ROWS ? pr[(kH - 1) * A.ref_cols + A.vel_col + i]
                 : Prf.ld(vb, ((kH - 1) * A.ref_cols + A.vel_col + i) * pN);
  }
#pragma unroll
  for (int k = kH - 1; k >= 0; --k) {
    float np_[3], nv_[3];  // next iteration's reference row, one step ahead
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      np_[i] = k == 0 ? 0.f
               : ROWS ? pr[(k - 1) * A.ref_cols + i]
                      : Prf.ld(vb, ((k - 1) * A.ref_cols + i) * pN);
      nv_[i] = k == 0 ? 0.f
               : ROWS ? pr[(k - 1) * A.ref_cols + A.vel_col + i]
                      : Prf.ld(vb, ((k - 1) * A.ref_cols + A.vel_col + i) * pN);
    }
    float lp = 0.f, lv = 0.f, lw = 0.f, lr = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float dp = st_pv[k][i] - rp[i], dv = st_pv[k][3 + i] - rv[i];
      const float wn = st_w[k + 1][i];
      lp += dp * dp, lv += dv * dv, lw += wn * wn;
      lam[i] += 2.f * A.w.pos * dp;
      lam[6 + i] += 2.f * A.w.vel * dv;
      lam[9 + i] += 2.f * A.w.av * wn;
    }
    const float a0 = act[k][0], da0 = a0 - 0.5f;
    float ga[4];
    ga[0] = 2.f * A.w.thrust * da0;
#pragma unroll
    for (int i = 1; i < 4; ++i) {
      const float d = act[k][i] - 0.5f;
      lr += d * d;
      ga[i] = 2.f * A.w.rates * d;
    }
    loss += A.w.pos * lp + A.w.vel * lv + A.w.av * lw + A.w.rates * lr +
            A.w.thrust * da0 * da0;
    quad_step_adjoint(lam, ga, a0, st_w[k], c, st_trig[k]);
#pragma unroll
    for (int i = 0; i < 4; ++i)  // dL/d(head pre-activation), in place
      act[k][i] = ga[i] * act[k][i] * (1.f - act[k][i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) rp[i] = np_[i], rv[i] = nv_[i];
  }
  // own rows of dL/dz in accumulator layout: rows khead(c, hi)
#pragma unroll
  for (int cc = 0; cc < 20; ++cc) {
    const int row = khead(cc, 0);
    const float v = hi ? act[(row + 4) >> 2][(row + 4) & 3] : act[row >> 2][row & 3];
    Pdz.st(vr, row * pN, v);
  }
  write_wave_partial(A.loss_partials, st_lo ? loss : 0.f);
}

}  // namespace
}  // namespace apg
