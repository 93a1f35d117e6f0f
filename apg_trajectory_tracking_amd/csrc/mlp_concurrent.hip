// mlp_concurrent.hip - the concurrent training step (BASELINE configs[1] as a
// full step) with the policy inside the kernels: forward kernel
// (mlp_concurrent_fwd.h), the reverse pass of the network WITH every weight and
// bias gradient (mlp_concurrent_bwd_tm_kernel), the second stage (mlp_common.h).
// Replaces TrainBase.run_epoch's concurrent branch (scripts/train_base.py:198-204)
// + TrainDrone.train_controller_model (scripts/train_drone.py:175-203).
#include "mlp_concurrent_fwd.h"

namespace apg {
namespace {
template <bool ROWS>
__global__ __launch_bounds__(kThreads) void mlp_concurrent_bwd_tm_kernel(WgArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  char *lds = reinterpret_cast<char *>(lds_f);
  const int lane = threadIdx.x & 63, hi = lane >> 5, row = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b0 = blockIdx.x * kTrajPerBlock;
  const int b = b0 + wave * 32 + row;
  const int B = A.B;
  const bool live = b < B;
  const unsigned pN = (unsigned)B * 4u;
  const Planes Pact(A.acts, kActPlanes, pN), Pdz(A.d_zout, kNA, pN);
  const unsigned vb = live ? (unsigned)b * 4u : kDead;
  const unsigned vr = live ? vb + (hi ? 4u * pN : 0u) : kDead;
  // trajectory-major addressing: lane = plane `row` of a 32-plane block, its
  // 16 trajectories start 4 hi into the wave's 32
  const unsigned wcol = (unsigned)(b0 + wave * 32) * 4u;
  const unsigned vt = (unsigned)row * pN + (unsigned)hi * 16u;
  const int nw = B - (b0 + wave * 32);      // live trajectories of this wave (may be <= 0)
  float *part = A.part + (size_t)blockIdx.x * kSlotsTm * 1024;
  char *lane_blk = lds + lane * 4;           // + region + block * 4096 + i * 256
  TmMeta &meta = *reinterpret_cast<TmMeta *>(lds + tMeta);
  bool bad = false;         // (workgroup-uniform) a non-finite operand was seen
  // The column 1-norms of W_1 (a bound on |W_1^T delta| per unit of max |delta|:
  // the scales of the states_in / conv cotangents, which are produced inside
  // the fc1 phase) - from the packed tables, first thing (nothing else is live
  // yet): wave w takes row block w of W_1^T (five blocks of the conv part, two
  // of the state part), a lane its row's 2 x 32 entries; the maxima are read
  // behind the layer barriers.  (Round 4: a block of the pack launch computed
  // them from the weights.  The tables' fp16 pairs carry the weights to 2^-22:
  // the same exponents.)
  if (wave < 7) {
    const int n0 = wave < 5 ? wC + 4 * wave : wS + 4 * (wave - 5);
    const u32x4 *tb = reinterpret_cast<const u32x4 *>(A.tables) + n0 * (kBlock16 / 16) + lane;
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const u32x4 th = tb[kb * (kBlock16 / 16)], tl = tb[kb * (kBlock16 / 16) + 64];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const h16x2 h_ = __builtin_bit_cast(h16x2, th[q]), l_ = __builtin_bit_cast(h16x2, tl[q]);
        sum += fabsf((float)h_[0] + (float)l_[0]) + fabsf((float)h_[1] + (float)l_[1]);
      }
    }
    sum += other_half(sum);
    sum = wave_fmax(sum);
    if (lane == 0) meta.wnorm[wave] = sum;
  }

  // ---- this wave's inputs: dL/dz feature-major (20 rows per half-wave) and
  // trajectory-major (rows 0..31 and 32..39), h3's first block; the maxima of
  // the unbounded x plane groups (this wave's 32 trajectories)
  float dzr[20];
#pragma unroll
  for (int cc = 0; cc < 20; ++cc) dzr[cc] = Pdz.ld(vr, khead(cc, 0) * pN);
  TBlock tz[2], tx;
  tz[0].load(Pdz, vt, wcol);
  tz[1].load(Pdz, row < kNA - 32 ? vt : kDead, 32u * pN + wcol);
  tx.load(Pact, vt, (unsigned)pH3 * pN + wcol);
  {
    zero_region(lds, tRA, tHeadEx - tRA);
    zero_region(lds, tConvLo, kNC * 32 * 4);
    fill_lds_issue(lds_f, A.tables, kWgTabFloats);
    unsigned amax = 0u;
#pragma unroll
    for (int cc = 0; cc < 20; ++cc) amax = umax_abs(amax, dzr[cc]);
    amax = wave_umax(amax);
    if (lane == 0) meta.dmax[0][wave] = amax;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the table DMA, tz
    // The head's 40 rows are the actions of ten different steps: their
    // cotangents differ by orders of magnitude, and ONE unit for the block
    // leaves the small rows with a few bits (round 5: 27 % of a row's own scale
    // with a x1e3 outlier in the workgroup, tests/test_gpu_round5.py).  Every
    // row gets its own exponent: the trajectory-major blocks have the row in
    // the lane, so a row's largest entry of this wave is lane-local; the
    // waves' biased exponents are exchanged as bytes behind this barrier.
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      float v[16];
      tz[mb].get(v);
      unsigned m = 0u;
#pragma unroll
      for (int i = 0; i < 16; ++i) m = rrow(i) + 4 * hi < nw ? umax_abs(m, v[i]) : m;
      const unsigned o = (unsigned)__shfl_xor((int)m, 32, 64);
      m = o > m ? o : m;
      if (hi == 0 && 32 * mb + row < kNA)
        reinterpret_cast<unsigned char *>(lds + tHeadEx)[wave * kNA + 32 * mb + row] =
            (unsigned char)(m >> 23);
    }
    __syncthreads();
  }
  // 2^ns, 2^nc: above the largest column 1-norm of W_1's state / conv part
  // (a non-finite norm: 0 - the gradients are non-finite anyway); scalars
  const auto norm_exp = [](float m) {
    return __builtin_amdgcn_readfirstlane(
        m > 0.f && m < 3.0e38f ? __builtin_amdgcn_frexp_expf(m) : 0);
  };
  int nc = norm_exp(fmaxf(fmaxf(fmaxf(meta.wnorm[0], meta.wnorm[1]),
                                fmaxf(meta.wnorm[2], meta.wnorm[3])), meta.wnorm[4]));
  int ns = norm_exp(fmaxf(meta.wnorm[5], meta.wnorm[6]));
  asm volatile("" : "+s"(nc), "+s"(ns));    // (computed HERE, kept in scalar registers)
  // exponent of this lane's head rows (32 mb + row): 2^e above the row's largest
  // cotangent of the workgroup (biased exponent field E: value < 2^(E - 126))
  int erow[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int r_ = 32 * mb + row < kNA ? 32 * mb + row : 0;
    unsigned e_ = 0u;
#pragma unroll
    for (int w8 = 0; w8 < kThreads / 64; ++w8) {
      const unsigned o = reinterpret_cast<const unsigned char *>(lds + tHeadEx)[w8 * kNA + r_];
      e_ = o > e_ ? o : e_;
    }
    erow[mb] = e_ ? (int)e_ - 126 : 0;
    if (wave == 0 && hi == 0 && 32 * mb + row < kNA)
      reinterpret_cast<int *>(lds + tHeadRow)[32 * mb + row] = erow[mb];
  }
  // The scales of the unbounded x plane groups (conv outputs, features + the
  // ones row, in_ref): the workgroup's maxima, which the forward kernel left per
  // wave (reading the planes for them here cost 9-16 us, wherever it was put)
  unsigned mc = 0u, mf = 0x3f800000u /* the ones row */, mi = 0u;
  {
    const unsigned *q = reinterpret_cast<const unsigned *>(A.xmax) +
                        (size_t)blockIdx.x * (kThreads / 64) * 4;
#pragma unroll
    for (int w8 = 0; w8 < kThreads / 64; ++w8) {
      mc = q[4 * w8] > mc ? q[4 * w8] : mc;
      mf = q[4 * w8 + 1] > mf ? q[4 * w8 + 1] : mf;
      mi = q[4 * w8 + 2] > mi ? q[4 * w8 + 2] : mi;
    }
  }
  const int fc = bits_exp(mc, bad, true), ff = bits_exp(mf, bad, true),
            fi = bits_exp(mi, bad, true);
  const LdsView16 L16(lds, lane);
  float hv[2][16];
  auto load_hv = [&](int plane) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) hv[rb][i] = Pact.ld(vr, (plane + rb * 32 + rrow(i)) * pN);
  };
  // this wave's largest next-layer cotangent -> its slot (read behind the barrier)
  auto post = [&](const f32x16 (&v)[2], int phase) {
    unsigned am = 0u;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) am = umax_abs(am, v[rb][i]);
    am = wave_umax(am);
    if (lane == 0) meta.dmax[phase][wave] = am;
  };
  // bias gradient of 32 rows: sums over the lane's 16 trajectories, both halves -
  // a float per wave and row, straight into the partial buffer (two slots of
  // [4 waves][4 layers][64]); the second stage sums the eight waves in order.
  // (Until round 4 a fixed-point LDS accumulator with the layer's unit: under a
  // x1e3 outlier in the workgroup the biases were 12-40 x noisier than the
  // plane path, tests/test_gpu_round5.py.)
  float *bias_part = part + (size_t)(uBias + (wave >> 2)) * 1024 + (wave & 3) * 256;
  auto add_bias = [&](const float (&v)[16], int layer, int mb, int rows) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    s += other_half(s);
    if (hi == 0 && row < rows)
      __builtin_nontemporal_store(bad ? __builtin_nanf("") : s,
                                  bias_part + layer * 64 + 32 * mb + row);
  };

  // ------------------------------------------------------------- head
  f32x16 d[2], e[2];
  const int e0 = wg_exp(meta.dmax[0], bad);
  (void)e0;   // (only its `bad` flag: the head's rows have their own exponents)
  {
    // The chain's operands are scaled PER TRAJECTORY (round 5; until round 4 by
    // the workgroup's exponent: a trajectory whose cotangents are 1e-4 of the
    // workgroup's largest then kept 2^-22 x 1e4 of relative accuracy - its
    // weight terms and bias sums were as noisy as that, tests/test_gpu_round5.py);
    // the transposed operands' rows - trajectories - come back with their own
    // scales, the exponents are brought into accumulator layout by texp.
    float amx = 0.f;
#pragma unroll
    for (int cc = 0; cc < 20; ++cc) amx = fmaxf(amx, fabsf(dzr[cc]));
    const int ex0 = scale_exponent(amx);
    Op16 x0[3];
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[j] = kb * 8 + j < 20 ? __builtin_amdgcn_ldexpf(dzr[kb * 8 + j < 20 ? kb * 8 + j : 0], -ex0)
                               : 0.f;
      x0[kb] = split8(v);
    }
    Op16 az[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      float v[16];
      tz[mb].get(v);
#pragma unroll
      for (int i = 0; i < 16; ++i)   // columns beyond B are somebody else's plane
        v[i] = rrow(i) + 4 * hi < nw ? v[i] : 0.f;
      add_bias(v, 0, mb, mb ? kNA - 32 : 32);
      split16(v, erow[mb] - kPreD, az[mb]);   // (per lane: the row's own exponent)
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      float xv[16];
      tx.get(xv);
      if (nb == 0) tx.load(Pact, vt, (unsigned)(pH3 + 32) * pN + wcol);
      else tx.load(Pact, vt, (unsigned)pH2 * pN + wcol);     // fc3's first x block

      Op16 bx[2];
      split16(xv, -kPreX, bx);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) acc = mma3(az[mb][kk], bx[kk], acc);
        add_block(lane_blk + tRA + (2 * nb + mb) * 4096, acc);
      }
      if (nb == 0) load_hv(pH3);
    }
    zero(d);
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) {
      d[0] = mma3(L16.A(0, wO + kb), x0[kb], d[0]);
      d[1] = mma3(L16.A(0, wO + 3 + kb), x0[kb], d[1]);
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        d[rb][i] = __builtin_amdgcn_ldexpf(d[rb][i], ex0) * (1.f - hv[rb][i] * hv[rb][i]);
    post(d, 1);
  }
  __syncthreads();
  {  // the head's four blocks [2 nb + mb]: row r(i) + 4 hi + 32 mb has its own unit
    const i32x4_ z = {0, 0, 0, 0};
    for (int idx = threadIdx.x; idx < 4 * 256; idx += kThreads) {
      i32x4_ *p = reinterpret_cast<i32x4_ *>(lds + tRA) + idx;
      const i32x4_ q = *p;
      const int el = 4 * idx, blk = el >> 10, i = (el >> 6) & 15, ln = el & 63;
      const int r_ = 32 * (blk & 1) + rrow(i) + 4 * (ln >> 5);
      const int e_ = reinterpret_cast<const int *>(lds + tHeadRow)[r_ < kNA ? r_ : 0];
      f32x4_ v;
#pragma unroll
      for (int c_ = 0; c_ < 4; ++c_)
        v[c_] = bad ? __builtin_nanf("") : __builtin_amdgcn_ldexpf((float)q[c_], e_ - kFix);
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4_ *>(part + sOut * 1024) + idx);
      *p = z;
    }
  }

  // The A operands of a layer's weight blocks from its cotangent in the chain's
  // orientation (round 6, as in mlp_rollout_bwd_tm_kernel; until round 5 the chain
  // ran a second time with swapped operands for them:
  // profiles/r06_transposition_probe.jsonl): the chain's own split x[kb] times an
  // identity B operand = trajectory r(i) + 4 hi of feature `lane & 31` in register
  // i (4 matrix instructions per 32 features, exact), the trajectories' exponents
  // in the same layout (texp), one ldexp per value to the workgroup's unit.  The
  // bias gradient: the wave's float sum per row, stored as before.
  u32x4 ident[2];
  ident_operands(lane, ident);
  auto transposed_operands = [&](const Op16 (&x)[4], int ex, int e_, int bias_id,
                                 Op16 (&ad)[2][2]) {
    int E[16];
    texp(ex, hi, E);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const Op16 pr[2] = {x[2 * mb], x[2 * mb + 1]};
      const f32x16 tz = to_feature_major(pr, ident);
      float v[16], sb = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        v[i] = __builtin_amdgcn_ldexpf(tz[i], E[i] - e_ + kPreD);
        sb += v[i];
      }
      sb += other_half(sb);
      if (hi == 0)
        __builtin_nontemporal_store(
            bad ? __builtin_nanf("") : __builtin_amdgcn_ldexpf(sb, e_ - kPreD),
            bias_part + bias_id * 64 + 32 * mb + row);
      split16(v, 0, ad[mb]);
    }
  };
  // One 64 x 64 layer: dl = its cotangent (accumulator layout), e_ = the
  // workgroup's exponent for it.  Weight blocks against the two x blocks (the
  // second one and `next_plane`'s first are requested on the way), the
  // cotangent of the layer below (tables `tab`), tanh' with the planes
  // `x_plane`; its maxima go to slot `phase + 1`.
  auto layer64 = [&](f32x16 (&dl)[2], f32x16 (&nx)[2], int e_, int tab, int x_plane,
                     int region, int bias_id, int next_plane, int phase) {
    Op16 x[4];
    const int ex = scaled_split64(dl, x);     // per trajectory
    Op16 ad[2][2];
    transposed_operands(x, ex, e_, bias_id, ad);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      float xv[16];
      tx.get(xv);
      tx.load(Pact, vt, (unsigned)(nb == 0 ? x_plane + 32 : next_plane) * pN + wcol);
      Op16 bx[2];
      split16(xv, -kPreX, bx);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) acc = mma3(ad[mb][kk], bx[kk], acc);
        add_block(lane_blk + region + (2 * nb + mb) * 4096, acc);
      }
      // (the tanh' operands of the feature-major chain below: requested here so
      // that they land under the second block's products)
      if (nb == 0) load_hv(x_plane);
    }
    zero(nx);
    dense64T_16(nx, x, L16, 0, tab);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        nx[rb][i] = __builtin_amdgcn_ldexpf(nx[rb][i], ex) * (1.f - hv[rb][i] * hv[rb][i]);
    post(nx, phase + 1);
  };
  // ---- fc3: x = h2 -> cotangent of h2's pre-activations
  const int e3 = wg_exp(meta.dmax[1], bad);
  layer64(d, e, e3, w3, pH2, tRB, 1, pH1, 1);
  __syncthreads();
  flush_region(lds, tRB, 4 * 1024, part + sFc3 * 1024, e3, true, bad);
  zero_region(lds, tF1a, tRA - tF1a);        // w3 and head tables: fc1's first blocks
  // ---- fc2: x = h1
  const int e2 = wg_exp(meta.dmax[2], bad);
  layer64(e, d, e2, w2, pH1, tRA, 2, pX1, 2);
  __syncthreads();
  flush_region(lds, tRA, 4 * 1024, part + sFc2 * 1024, e2, false, bad);

  // ---- fc1 (x = the 224 x1 planes: s1 | relu(conv)), states_in and conv
  const int e1 = wg_exp(meta.dmax[3], bad);
  const int es = e1 + ns, ec = e1 + nc;      // bounds of |d_pre_s|, |d conv|
  {
    Op16 x1s[4];
    const int ex1 = scaled_split64(d, x1s);   // per trajectory
    Op16 ad[2][2];
    transposed_operands(x1s, ex1, e1, 3, ad);
    // B operands that stay: the 15 feature planes + a row of ones (states_in's
    // bias column), the 90 in_ref planes in three blocks (conv windows)
    Op16 bfeat[2], binr[3][2];
    if (ROWS) {
      // x^T straight from the data set: lane = column `row` of the block, its 16
      // trajectories c + 8 g + 4 hi are 16 rows named by the index - one dword
      // load each, a half-wave on 32 consecutive floats of ONE row.  The wave's
      // 32 row numbers: one per lane, handed around by v_readlane.
      const int bw = b0 + wave * 32 + row;
      const unsigned r_ = (unsigned)A.index[bw < B ? bw : B - 1];
      const unsigned rf = r_ * (unsigned)A.ld_feat, ri = r_ * (unsigned)A.ld_in_ref;
      const auto sf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A.r_feat), 0,
                                                        (int)A.bytes_feat, 0x00020000);
      const auto si = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A.r_in_ref), 0,
                                                        (int)A.bytes_in_ref, 0x00020000);
      float v[16];
      const auto rows_block = [&](__amdgpu_buffer_rsrc_t rs, unsigned rbase, int col, bool on) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const unsigned a0 = (unsigned)__builtin_amdgcn_readlane((int)rbase, c + 8 * g),
                           a1 = (unsigned)__builtin_amdgcn_readlane((int)rbase, c + 8 * g + 4);
            const unsigned off = on ? ((hi ? a1 : a0) + (unsigned)col) * 4u : kDead;
            v[4 * g + c] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, APG_PLANES_LD_AUX));
          }
      };
      rows_block(sf, rf, row, row < kNF);
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = row == kNF ? 1.f : v[i];
      split16(v, ff - kPreX, bfeat);
#pragma unroll
      for (int jb = 0; jb < 3; ++jb) {
        rows_block(si, ri, 32 * jb + row, 32 * jb + row < kH * kRD);
        split16(v, fi - kPreXc, binr[jb]);
      }
    } else {
      TBlock tf;
      tf.load(Pact, row < kNF ? vt : kDead, (unsigned)pFeat * pN + wcol);
      float v[16];
      tf.get(v);
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = row == kNF ? 1.f : v[i];
      split16(v, ff - kPreX, bfeat);
#pragma unroll
      for (int jb = 0; jb < 3; ++jb) {
        tf.load(Pact, 32 * jb + row < kH * kRD ? vt : kDead,
                (unsigned)(pInr + 32 * jb) * pN + wcol);
        tf.get(v);
        split16(v, fi - kPreXc, binr[jb]);
      }
    }
    // fc1's weight blocks 2 nb, 2 nb + 1 against x block nb (scaled by 2^-fx)
    auto fc1_blocks = [&](const float (&xv)[16], int fx, int nb) {
      Op16 bx[2];
      split16(xv, fx - kPreX, bx);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) acc = mma3(ad[mb][kk], bx[kk], acc);
        const int blk = 2 * nb + mb;     // blocks 0..6 in the first piece
        add_block(lane_blk + (blk < 7 ? tF1a + blk * 4096 : tF1b + (blk - 7) * 4096), acc);
      }
    };
    // the transposed product of d_pre1 against four table blocks from `blk0`
    auto transposed = [&](int blk0) {
      const char *tb = L16.b0 + blk0 * kBlock16;   // (all below 60 KB)
      f32x16 t;
#pragma unroll
      for (int i = 0; i < 16; ++i) t[i] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        Op16 w;
        w.h = *reinterpret_cast<const u32x4 *>(tb + kb * kBlock16);
        w.l = *reinterpret_cast<const u32x4 *>(tb + kb * kBlock16 + 1024);
        t = mma3(x1s[kb], w, t);
      }
      return t;
    };
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      // s1's blocks: also states_in - cotangent of its pre-activations, block
      // nb, and the weight block against the features
      float xv[16];
      tx.get(xv);
      tx.load(Pact, vt, (unsigned)(pX1 + 32 * (nb + 1)) * pN + wcol);
      fc1_blocks(xv, 0, nb);
      const f32x16 t = transposed(wS + 4 * nb);
      int E1[16];
      texp(ex1, hi, E1);
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i)
        v[i] = __builtin_amdgcn_ldexpf(t[i], E1[i]) * (1.f - xv[i] * xv[i]);
      Op16 as[2];
      split16(v, es - kPreD, as);
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) acc = mma3(as[kk], bfeat[kk], acc);
      // 16 columns are real (15 features + the ones row): compact [reg][half][16],
      // 2 KB of high limbs per block, the low limbs 4 KB further
      if (row < 16) {
        char *q = lds + tSin + nb * 2048 + (hi * 16 + row) * 4;
#pragma unroll
        for (int i = 0; i < 16; ++i) lds_add2(q + i * 128, q + 4096 + i * 128, acc[i], kFix);
      }
    }
#pragma unroll 1
    for (int eb = 0; eb < 5; ++eb) {
      // conv blocks: x block 2 + eb = the saved conv outputs e = 32 eb + row
      // (channel 4 eb + row / 8, position row % 8); their cotangent with relu'
      // from the saved outputs, then its products against the in_ref planes
      float xv[16];
      tx.get(xv);
      if (eb < 4) tx.load(Pact, vt, (unsigned)(pX1 + 32 * (eb + 3)) * pN + wcol);
      fc1_blocks(xv, fc, eb + 2);
      const f32x16 t = transposed(wC + 4 * eb);
      int E1[16];
      texp(ex1, hi, E1);
      float v[16], sum = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        v[i] = xv[i] > 0.f ? __builtin_amdgcn_ldexpf(t[i], E1[i] - ec) : 0.f;   // / 2^ec
        sum += v[i];
      }
      // the conv block's rows of channels 4 eb .. 4 eb + 3 (accumulator layout:
      // channel ch = register (ch & 3) + 4 (ch >> 3) of half-wave (ch >> 2) & 1)
      char *cblk = lds + tConv + ((4 * (eb >> 1)) * 64 + 32 * (eb & 1)) * 4;
      char *clo = lds + tConvLo + 4 * eb * 32 * 4;        // low limbs: [channel][32]
      sum += other_half(sum);
      // bias: column 27; the block's unit carries in_ref's scale 2^fi as well
      if (hi == 0)
        lds_add2(cblk + ((row >> 3) * 64 + 27) * 4, clo + ((row >> 3) * 32 + 27) * 4,
                 __builtin_amdgcn_ldexpf(sum, kFixConv - fi));
      Op16 ac[2];
      split16(v, -kPreDc, ac);
#pragma unroll
      for (int jb = 0; jb < 3; ++jb) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) acc = mma3(ac[kk], binr[jb][kk], acc);
        // register 4 g + c of lane (hi, col): conv output row c + 8 g + 4 hi of the
        // block = channel 4 eb + g at position c + 4 hi, against in_ref plane
        // j = 32 jb + col: tap q = j - 9 position of that channel's 27
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int q = 32 * jb + row - kRD * (c + 4 * hi);
          if (q >= 0 && q < 27) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
              lds_add2(cblk + (g * 64 + q) * 4, clo + (g * 32 + q) * 4, acc[4 * g + c]);
          }
        }
      }
    }
  }
  __syncthreads();
  // fc1: blocks 0..3 against s1 (unit 2^e1), 4..13 against the conv outputs (2^(e1 + fc))
  flush_region(lds, tF1a, 4 * 1024, part + sFc1 * 1024, e1, false, bad);
  flush_region(lds, tF1a + 4 * 4096, 3 * 1024, part + (sFc1 + 4) * 1024, e1 + fc, false, bad);
  flush_region(lds, tF1b, 7 * 1024, part + (sFc1 + 7) * 1024, e1 + fc, false, bad);
  for (int idx = threadIdx.x; idx < 2 * 512; idx += kThreads) {   // states_in: both limbs
    const int nb = idx >> 9, r_ = idx & 511, at = (r_ >> 5) * 64 + 32 * ((r_ >> 4) & 1) + (r_ & 15);
    const int *q = reinterpret_cast<const int *>(lds + tSin) + nb * 512 + r_;
    const double v = (double)q[0] + (double)q[1024] * (1.0 / (double)(1 << kFix));
    part[(sSin + nb) * 1024 + at] =
        bad ? __builtin_nanf("") : __builtin_amdgcn_ldexpf((float)v, es + ff - kFix);
  }
  // conv: both limbs of the 20 x 28 used elements (accumulator layout in the slot)
  for (int idx = threadIdx.x; idx < kNC * 32; idx += kThreads) {
    const int ch = idx >> 5, q = idx & 31;
    const int at = ((ch & 3) + 4 * (ch >> 3)) * 64 + 32 * ((ch >> 2) & 1) + q;
    const double v = (double)reinterpret_cast<const int *>(lds + tConv)[at] +
                     (double)reinterpret_cast<const int *>(lds + tConvLo)[idx] *
                         (1.0 / (double)(1 << kFixConv));
    part[uConv * 1024 + at] =
        bad ? __builtin_nanf("") : __builtin_amdgcn_ldexpf((float)v, ec + fi - kFixConv);
  }
}

}  // namespace
}  // namespace apg

using namespace apg;

extern "C" {

// [forward tables | in-sweep reverse tables | 4 | map of the resident tables]
int apg_quad_mlp_step_workspace_floats(void) { return kCfLds + kWgTabFloats + 4 + kMapInts; }

long long apg_quad_mlp_step_partials_floats(int B) {
  if (B <= 0) return 0;
  const long long wgs = (B + kTrajPerBlock - 1) / kTrajPerBlock;
  // the workgroups' partials + the chunk sums of the first reduction level
  // + the forward kernel's per-wave x maxima
  const long long need = (wgs + (wgs + kRedChunk - 1) / kRedChunk) * kSlotsTm * 1024 + wgs * 32;
  // (also the scratch of the one-time table map: index parameters, their
  // tables, owners)
  const long long scratch = 2ll * kParamFloats + kCfLds + kWgTabFloats + 8;
  return need > scratch ? need : scratch;
}

namespace {
int concurrent_train_step(
    const ApgBatchRows *rows, const float *state0, const float *ref, int ref_cols, float dt,
    const ApgQuadParams *params, const ApgQuadLossWeights *weights,
    const ApgMlpPolicy *policy, int B, int H, float *acts, unsigned *relu_mask,
    float *d_zout, float *loss_partials, float *loss, const ApgMlpPolicyGrads *grads,
    float *states, float *workspace, float *partials, const ApgMlpSgdUpdate *update,
    const ApgStepEvents *events, apg_stream_t stream);
}  // namespace

int apg_quad_mlp_concurrent_step(
    const float *state0, const float *ref, int ref_cols, float dt,
    const ApgQuadParams *params, const ApgQuadLossWeights *weights,
    const ApgMlpPolicy *policy, int B, int H, float *acts, unsigned *relu_mask,
    float *d_zout, float *loss_partials, float *loss, const ApgMlpPolicyGrads *grads,
    float *states, float *workspace, float *partials, apg_event_t after_reverse,
    apg_stream_t stream) {
  ApgStepEvents ev = {nullptr, nullptr, after_reverse};
  return apg_quad_mlp_concurrent_train_step(
      state0, ref, ref_cols, dt, params, weights, policy, B, H, acts, relu_mask, d_zout,
      loss_partials, loss, grads, states, workspace, partials, nullptr, &ev, stream);
}

int apg_quad_mlp_concurrent_train_step(
    const float *state0, const float *ref, int ref_cols, float dt,
    const ApgQuadParams *params, const ApgQuadLossWeights *weights,
    const ApgMlpPolicy *policy, int B, int H, float *acts, unsigned *relu_mask,
    float *d_zout, float *loss_partials, float *loss, const ApgMlpPolicyGrads *grads,
    float *states, float *workspace, float *partials, const ApgMlpSgdUpdate *update,
    const ApgStepEvents *events, apg_stream_t stream) {
  return concurrent_train_step(nullptr, state0, ref, ref_cols, dt, params, weights, policy, B, H,
                               acts, relu_mask, d_zout, loss_partials, loss, grads, states,
                               workspace, partials, update, events, stream);
}

int apg_quad_mlp_concurrent_train_step_rows(
    const ApgBatchRows *rows, int ref_cols, float dt, const ApgQuadParams *params,
    const ApgQuadLossWeights *weights, const ApgMlpPolicy *policy, int B, int H, float *acts,
    unsigned *relu_mask, float *d_zout, float *loss_partials, float *loss,
    const ApgMlpPolicyGrads *grads, float *states, float *workspace, float *partials,
    const ApgMlpSgdUpdate *update, const ApgStepEvents *events, apg_stream_t stream) {
  if (!rows) { set_error("rows is NULL"); return APG_ERR_ARG; }
  if (B > 0 && (!rows->index || !rows->normed || !rows->state0 || !rows->in_ref || !rows->ref)) {
    set_error("rows: NULL pointer");
    return APG_ERR_ARG;
  }
  if (rows->n_rows < 1 || rows->ld_normed < kNF || rows->ld_state0 < 12 ||
      rows->ld_in_ref < kH * kRD || rows->ld_ref < kH * ref_cols) {
    set_error("rows: need n_rows >= 1 and row strides of at least 15 / 12 / 90 / H x ref_cols");
    return APG_ERR_ARG;
  }
  const long long widest = rows->ld_in_ref > rows->ld_ref ? rows->ld_in_ref : rows->ld_ref;
  if (rows->n_rows * widest * 4 >= (1ll << 32) - 64 ||
      rows->n_rows * (long long)rows->ld_normed * 4 >= (1ll << 32) - 64 ||
      rows->n_rows * (long long)rows->ld_state0 * 4 >= (1ll << 32) - 64) {
    set_error("rows: a data-set tensor of 4 GiB or more (32-bit row offsets); gather the "
              "batch with apg_to_soa_multi instead");
    return APG_ERR_ARG;
  }
  return concurrent_train_step(rows, nullptr, nullptr, ref_cols, dt, params, weights, policy, B,
                               H, acts, relu_mask, d_zout, loss_partials, loss, grads, states,
                               workspace, partials, update, events, stream);
}

}  // extern "C"

namespace {
int concurrent_train_step(
    const ApgBatchRows *rows, const float *state0, const float *ref, int ref_cols, float dt,
    const ApgQuadParams *params, const ApgQuadLossWeights *weights,
    const ApgMlpPolicy *policy, int B, int H, float *acts, unsigned *relu_mask,
    float *d_zout, float *loss_partials, float *loss, const ApgMlpPolicyGrads *grads,
    float *states, float *workspace, float *partials, const ApgMlpSgdUpdate *update,
    const ApgStepEvents *events, apg_stream_t stream) {
  if (int e = check_mlp(params, policy, B, H)) return e;
  if (update && (!all_set(update->param) || !all_set(update->momentum_buf))) {
    set_error("update: parameter / momentum pointer is NULL");
    return APG_ERR_ARG;
  }
  if (update && !(update->lr == update->lr && update->momentum == update->momentum)) {
    set_error("update: lr / momentum is NaN");
    return APG_ERR_ARG;
  }
  if (!weights) { set_error("weights is NULL"); return APG_ERR_ARG; }
  if (ref_cols != 9 && ref_cols != 6) {
    set_error("ref_cols must be 9 or 6");
    return APG_ERR_ARG;
  }
  if (!grads || !all_set(*grads)) {
    set_error("gradient pointer is NULL");
    return APG_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    // (an update with zero gradients would still move the parameters by the
    // decaying momentum: not implemented)
    if (update) {
      set_error("update with B = 0 is not supported");
      return APG_ERR_ARG;
    }
    // no trajectory: zero gradients, zero loss
    const ApgMlpPolicyGrads &g = *grads;
    float *ptrs[12] = {g.w_s, g.b_s, g.conv_w, g.conv_b, g.w_1, g.b_1,
                       g.w_2, g.b_2, g.w_3, g.b_3, g.w_out, g.b_out};
    const size_t n[12] = {kW * kNF, kW, kNC * 27, kNC, kW * kN1, kW,
                          kW * kW, kW, kW * kW, kW, kNA * kW, kNA};
    for (int i = 0; i < 12; ++i)
      if (hipMemsetAsync(ptrs[i], 0, n[i] * sizeof(float), st) != hipSuccess)
        return check_launch("memset(grads)");
    if (loss && hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess)
      return check_launch("memset(loss)");
    return APG_OK;
  }
  if ((!rows && (!state0 || !ref)) || !acts || !relu_mask || !d_zout || !loss_partials ||
      !workspace || !partials) {
    set_error("NULL buffer");
    return APG_ERR_ARG;
  }
  if ((long long)B * 4 * kActPlanes >= (1ll << 32) - 64) {
    set_error("B too large for 32-bit plane offsets; split the batch");
    return APG_ERR_ARG;
  }
  static PerDeviceOnce attr;
  if (!attr.test()) {
    if (int e = raise_lds(mlp_concurrent_fwd_kernel<false>, kCfLds)) return e;
    if (int e = raise_lds(mlp_concurrent_fwd_kernel<true>, kCfRowsLds)) return e;
    if (int e = raise_lds(mlp_concurrent_bwd_tm_kernel<false>, kLdsAll / 4)) return e;
    if (int e = raise_lds(mlp_concurrent_bwd_tm_kernel<true>, kLdsAll / 4)) return e;
    attr.set();
  }
  const size_t plane = (size_t)B;
  ConcArgs A;
  A.feat = acts + pFeat * plane, A.in_ref = acts + pInr * plane;
  A.state0 = state0, A.ref = ref;
  A.x1 = acts + pX1 * plane, A.h = acts + pH1 * plane, A.mask = relu_mask;
  A.d_zout = d_zout, A.d_pre = nullptr, A.d_conv = nullptr;
  A.states = states, A.loss_partials = loss_partials;
  A.tables = workspace;
  // (behind the workgroups' partials and the chunk sums)
  {
    const long long wgs_ = (B + kTrajPerBlock - 1) / kTrajPerBlock;
    A.xmax = partials + (size_t)((wgs_ + (wgs_ + kRedChunk - 1) / kRedChunk) * kSlotsTm * 1024);
  }
  A.c = make_const(*params, dt);
  A.w = *weights;
  A.B = B, A.ref_cols = ref_cols, A.vel_col = ref_cols == 9 ? 6 : 3;
  A.index = nullptr, A.o_feat = A.o_in_ref = nullptr, A.win16 = 0;
  if (rows) {
    A.index = rows->index;
    A.o_feat = acts + pFeat * plane, A.o_in_ref = acts + pInr * plane;
    A.feat = rows->normed, A.in_ref = rows->in_ref, A.state0 = rows->state0, A.ref = rows->ref;
    A.ld_feat = rows->ld_normed, A.ld_in_ref = rows->ld_in_ref;
    A.ld_state0 = rows->ld_state0, A.ld_ref = rows->ld_ref;
    const auto bytes = [&](int ld) { return (unsigned)(rows->n_rows * (long long)ld * 4); };
    A.bytes_feat = bytes(A.ld_feat), A.bytes_in_ref = bytes(A.ld_in_ref);
    A.win16 = A.ld_in_ref % 4 == 0 && (reinterpret_cast<uintptr_t>(rows->in_ref) & 15) == 0;
    A.bytes_state0 = bytes(A.ld_state0), A.bytes_ref = bytes(A.ld_ref);
  }
  PackArgs P;
  P.pol = *policy, P.dst = workspace, P.head_rows = kNA;
  const int blocks = (B + kTrajPerBlock - 1) / kTrajPerBlock;
  const int fwd_blocks = (kCfLds + 255) / 256, bwd_blocks = (kWgTabFloats + 255) / 256;
  // resident tables (ApgMlpSgdUpdate.resident): 2 - the workspace holds the tables
  // of exactly these parameters, left by the previous call's second stage: no
  // pack launch; 1 - pack, and build the map the second stage scatters by; 3 -
  // pack (somebody else wrote the parameters), the workspace's map is still good
  const int resident = update ? update->resident : 0;
  if (resident < 0 || resident > 3) {
    set_error("update: resident must be 0, 1, 2 or 3");
    return APG_ERR_ARG;
  }
  int *map = reinterpret_cast<int *>(workspace + kCfLds + kWgTabFloats + 4);
  if (resident == 1) {
    // index-valued parameters -> their tables -> owners -> map (scratch: partials,
    // overwritten by the step afterwards)
    float *par = partials, *tab = partials + kParamFloats;
    int *owner = reinterpret_cast<int *>(tab + kCfLds + kWgTabFloats + 4);
    // (the pack kernel leaves the gaps of the float tables alone: whatever the
    // scratch held there would be read as parameter indices)
    // (... and the owner table, so that nothing in this path indexes by garbage)
    if (hipMemsetAsync(tab, 0,
                       (size_t)(kCfLds + kWgTabFloats + 4 + kParamFloats) * sizeof(float),
                       st) != hipSuccess)
      return check_launch("memset(table map scratch)");
    hipLaunchKernelGGL(tabmap_iota_kernel, dim3((kParamFloats + 255) / 256), dim3(256), 0, st,
                       par);
    PackArgs Q;
    const ApgMlpPolicyGrads f = params_in(par);
    Q.pol = ApgMlpPolicy{f.w_s, f.b_s, f.conv_w, f.conv_b, f.w_1, f.b_1,
                         f.w_2, f.b_2, f.w_3, f.b_3, f.w_out, f.b_out};
    Q.dst = tab, Q.head_rows = kNA;
    hipLaunchKernelGGL(mlp_pack_step_kernel, dim3(fwd_blocks + bwd_blocks), dim3(256), 0,
                       st, Q, fwd_blocks);
    hipLaunchKernelGGL(tabmap_owner_kernel, dim3(kSlotsTm * 4), dim3(256), 0, st, par, owner,
                       map, kSlotsTm, uBias);
    hipLaunchKernelGGL(tabmap_invert_kernel, dim3((kCfLds + kWgTabFloats + 255) / 256),
                       dim3(256), 0, st, tab, owner, map);
  }
  if (resident != 2)
    hipLaunchKernelGGL(mlp_pack_step_kernel, dim3(fwd_blocks + bwd_blocks), dim3(256), 0,
                       st, P, fwd_blocks);
  // (the tables are packed while the caller's producer of acts / state0 / ref -
  // a gather on another stream - may still be running)
  if (events && events->inputs_ready &&
      hipStreamWaitEvent(st, (hipEvent_t)events->inputs_ready, 0) != hipSuccess)
    return check_launch("hipStreamWaitEvent(inputs_ready)");
  if (rows)
    hipLaunchKernelGGL(mlp_concurrent_fwd_kernel<true>, dim3(blocks), dim3(kThreads),
                       kCfRowsLds * sizeof(float), st, A);
  else
    hipLaunchKernelGGL(mlp_concurrent_fwd_kernel<false>, dim3(blocks), dim3(kThreads),
                       kCfLds * sizeof(float), st, A);
  if (events && events->after_forward &&
      hipEventRecord((hipEvent_t)events->after_forward, st) != hipSuccess)
    return check_launch("hipEventRecord(after_forward)");
  WgArgs W;
  W.acts = acts, W.mask = relu_mask, W.d_zout = d_zout, W.part = partials;
  W.tables = workspace + kCfLds, W.B = B, W.xmax = A.xmax;
  W.index = nullptr, W.r_feat = W.r_in_ref = nullptr;
  if (rows && B % kTrajPerBlock) {
    // A ragged last workgroup reads its dead trajectories' x^T entries past the
    // end of a plane - the head of the next plane: finite numbers, times a zero
    // cotangent.  Behind the LAST activation plane that is the window region,
    // which nobody writes in this mode: keep its head finite.
    const size_t head = (size_t)kTrajPerBlock * 4;
    const size_t region = (size_t)kH * kRD * plane * 4;
    if (hipMemsetAsync(acts + pInr * plane, 0, head < region ? head : region, st) != hipSuccess)
      return check_launch("memset(window planes' head)");
  }
  if (rows) {
    W.index = rows->index, W.r_feat = rows->normed, W.r_in_ref = rows->in_ref;
    W.ld_feat = A.ld_feat, W.ld_in_ref = A.ld_in_ref;
    W.bytes_feat = A.bytes_feat, W.bytes_in_ref = A.bytes_in_ref;
    hipLaunchKernelGGL(mlp_concurrent_bwd_tm_kernel<true>, dim3(blocks), dim3(kThreads), kLdsAll,
                       st, W);
  } else {
    hipLaunchKernelGGL(mlp_concurrent_bwd_tm_kernel<false>, dim3(blocks), dim3(kThreads), kLdsAll,
                       st, W);
  }
  // the inputs (activation planes, state0, ref) are not read past this point:
  // a caller that pipelines batches may start refilling the NEXT batch's
  // buffers behind this event while the second stage and the update run
  if (events && events->after_reverse &&
      hipEventRecord((hipEvent_t)events->after_reverse, st) != hipSuccess)
    return check_launch("hipEventRecord(after_reverse)");
  WgReduceArgs R;
  R.part = partials, R.g = *grads, R.loss_partials = loss_partials, R.loss = loss;
  R.loss_sum = rows && loss ? rows->running_loss : nullptr;
  R.ws = reinterpret_cast<char *>(workspace), R.map = resident ? map : nullptr;
  R.wgs = blocks, R.n_partials = blocks * (kThreads / kWave);
  R.n_slots = kSlotsTm, R.bias_slot = uBias, R.conv_src = 1, R.bias_src = 8;
  R.head_rows = kNA, R.conv_bias_here = false;
  const int columns = (R.n_slots * 1024 + 255) / 256;
  R.update = update != nullptr;
  R.param = update ? update->param : *grads, R.mom = update ? update->momentum_buf : *grads;
  R.lr = update ? update->lr : 0.0, R.momentum = update ? update->momentum : 0.0;
  if (blocks > kRedChunk) {
    const int chunks = (blocks + kRedChunk - 1) / kRedChunk;
    float *chunk_sums = partials + (size_t)blocks * R.n_slots * 1024;
    hipLaunchKernelGGL(mlp_wgrad_reduce1_kernel, dim3(columns, chunks), dim3(256), 0, st,
                       partials, chunk_sums, blocks, R.n_slots);
    R.part = chunk_sums, R.wgs = chunks;
  }
  hipLaunchKernelGGL(mlp_wgrad_reduce_kernel, dim3(columns), dim3(256), 0, st, R);
  return check_launch("quad_mlp_concurrent_step");
}
}  // namespace
