// mlp_planes.hip - the reverse kernels of rounds 1-4 that write COTANGENT PLANES
// for planes_gemm products (autoregressive: mlp_rollout_bwd_kernel; concurrent:
// mlp_concurrent_bwd_kernel), behind their C entry points.  Round 6: NOT part of
// libapg_hip.so - the product has one reverse kernel per mode, with the weight
// gradients inside the sweep - but of libapg_planes.so, which only tests load
// (tests/plane_path.py: an independent implementation of the same sums with exact
// float accumulation, what the row arbiter and the in-sweep tests compare with).
// Declarations: include/apg_planes.h.
#include "mlp_concurrent_fwd.h"
#include "apg_planes.h"

namespace apg {
namespace {
__global__ __launch_bounds__(256) void mlp_pack_cbwd_kernel(PackArgs A) {
  pack_cbwd(A, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// both tables in one launch (the concurrent step runs its sweeps back to
// back): blocks [0, fwd_blocks) write the forward tables at dst, the others the
// reverse tables at dst + kCfLds
__global__ __launch_bounds__(256) void mlp_pack_pair_kernel(PackArgs A, int fwd_blocks) {
  if ((int)blockIdx.x < fwd_blocks) {
    pack_cfwd(A, blockIdx.x * blockDim.x + threadIdx.x, fwd_blocks * blockDim.x);
  } else {
    A.dst += kCfLds;
    pack_cbwd(A, (blockIdx.x - fwd_blocks) * blockDim.x + threadIdx.x,
              (gridDim.x - fwd_blocks) * blockDim.x);
  }
}

struct BwdArgs {
  const float *state0, *states, *actions, *ref, *x1, *h;
  const unsigned *mask;
  float *loss_partials;
  float *d_pre;   // [256][N]: d_pre1, d_pre2, d_pre3, d_pre_s (64 each)
  float *d_zout;  // [4][N]
  float *d_conv;  // [720][B]: window-diagonal sums of the conv cotangents (kConvP)
  float *grad_state0;
  const float *tables;  // packed operand tables (mlp_pack_cbwd_kernel)
  QuadConst c;
  ApgQuadLossWeights w;
  int B, ref_cols, vel_col;
};

__global__ __launch_bounds__(kThreads) void mlp_rollout_bwd_kernel(BwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  fill_lds(lds, A.tables, kCbLds);
  const LdsView16 L16(lds, threadIdx.x & 63);
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const LdsView L(lds, lane);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = (blockIdx.x * (kThreads / 64) + wave) * 32 + (lane & 31);
  const int B = A.B;
  const bool live = b < B;  // dead lanes: out-of-range offsets, see forward
  const bool st_lo = live && hi == 0;
  const unsigned pitchB = (unsigned)B * 4u, pitchN = pitchB * kH;
  const QuadConst c = A.c;
  const Planes Ps0(A.state0, 12, pitchB), Pst(A.states, kH * 12, pitchB);
  const Planes Pac(A.actions, kH * 4, pitchB), Prf(A.ref, kH * A.ref_cols, pitchB);
  const Planes Px1(A.x1, kN1, pitchN), Ph(A.h, 3 * kW, pitchN);
  const Planes Pmk(A.mask, 5, pitchN), Pdp(A.d_pre, 4 * kW, pitchN);
  const Planes Pdz(A.d_zout, 4, pitchN), Pdc(A.d_conv, kConvPlanes, pitchB);
  const unsigned vb = live ? (unsigned)b * 4u : kDead;

  float lam[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) lam[i] = 0.f;
  float loss = 0.f;
  // sliding diagonal sums of the conv cotangents: dgn[ch][ii] = diagonal
  // tau = k + ii of this half-wave's positions (see kConvP)
  float dgn[kNC][4];
#pragma unroll
  for (int ch = 0; ch < kNC; ++ch)
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) dgn[ch][ii] = 0.f;
  const unsigned vg = live ? (unsigned)b * 4u + (hi ? kTau * pitchB : 0u) : kDead;
  const unsigned vb_lo = st_lo ? (unsigned)b * 4u : kDead;

#pragma unroll 1
  for (int k = kH - 1; k >= 0; --k) {
    const unsigned pB = opaque(pitchB), pN = opaque(pitchN);
    const unsigned col = (unsigned)b * 4u + (unsigned)k * pitchB;
    const unsigned vn = live ? col : kDead;
    const unsigned vn_lo = st_lo ? col : kDead;
    const unsigned vr = live ? col + (hi ? 4u * pitchN : 0u) : kDead;
    float sn[12], sc[12], a[4], rp[3], rv[3];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      sn[i] = Pst.ld(vb, (k * 12 + i) * pB);
      sc[i] = k > 0 ? Pst.ld(vb, ((k - 1) * 12 + i) * pB) : Ps0.ld(vb, i * pB);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = Pac.ld(vb, (k * 4 + j) * pB);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      rp[i] = Prf.ld(vb, (k * A.ref_cols + i) * pB);
      rv[i] = Prf.ld(vb, (k * A.ref_cols + A.vel_col + i) * pB);
    }
    unsigned mw[5];
#pragma unroll
    for (int eb = 0; eb < 5; ++eb) mw[eb] = Pmk.ldu(vn, eb * pN);
    float hv[2][16];
    load_acts(hv, Ph, 2 * kW, vr, pN);  // h3
    __builtin_amdgcn_sched_barrier(0);
    float lp = 0.f, lv = 0.f, lw = 0.f, lr = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float dp = sn[i] - rp[i], dv = sn[6 + i] - rv[i], wn = sn[9 + i];
      lp += dp * dp, lv += dv * dv, lw += wn * wn;
      lam[i] += 2.f * A.w.pos * dp;
      lam[6 + i] += 2.f * A.w.vel * dv;
      lam[9 + i] += 2.f * A.w.av * wn;
    }
    const float da0 = a[0] - 0.5f;
    float ga[4];
    ga[0] = 2.f * A.w.thrust * da0;
#pragma unroll
    for (int j = 1; j < 4; ++j) {
      const float d = a[j] - 0.5f;
      lr += d * d;
      ga[j] = 2.f * A.w.rates * d;
    }
    loss += A.w.pos * lp + A.w.vel * lv + A.w.av * lw + A.w.rates * lr +
            A.w.thrust * da0 * da0;
    const Trig t = make_trig(&sc[3]);
    quad_step_adjoint(lam, ga, a[0], &sc[9], c, t);

    // head (VALU): d/dh3 in accumulator layout
    float dz[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dz[j] = ga[j] * a[j] * (1.f - a[j]);
      Pdz.st(vn_lo, j * pN, dz[j]);
    }
    f32x16 d[2], e[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          v = fmaf(L.T(gTo + ((j * 2 + rb) * 16 + i) * 2), dz[j], v);
        d[rb][i] = v;
      }
    __builtin_amdgcn_sched_barrier(0);
    // the reverse layers on the 16-bit matrix pipe (policy_mfma16.h):
    // cotangents scaled per trajectory, two fp16 terms, three products
    tanh_adjoint(d, hv, Pdp, 2 * kW, vr, pN);  // d_pre3
    load_acts(hv, Ph, kW, vr, pN);             // h2, lands under the product
    __builtin_amdgcn_sched_barrier(0);
    Op16 x[4];
    zero(e);
    int ex = scaled_split64(d, x);
    dense64T_16(e, x, L16, gA, m3T);
    __builtin_amdgcn_sched_barrier(0);
    tanh_adjoint(e, hv, Pdp, kW, vr, pN, ex);  // d_pre2
    load_acts(hv, Ph, 0, vr, pN);              // h1
    __builtin_amdgcn_sched_barrier(0);
    zero(d);
    ex = scaled_split64(e, x);
    dense64T_16(d, x, L16, gA, m2T);
    __builtin_amdgcn_sched_barrier(0);
    tanh_adjoint(d, hv, Pdp, 0, vr, pN, ex);   // d_pre1
    load_acts(hv, Px1, 0, vr, pN);             // s1
    __builtin_amdgcn_sched_barrier(0);
    // fc1 inputs, state branch; d_pre1's split also feeds the conv part below
    zero(e);
    const int ex1 = scaled_split64(d, x);
    dense64T_16(e, x, L16, gA, m1sT);
    __builtin_amdgcn_sched_barrier(0);
    tanh_adjoint(e, hv, Pdp, 3 * kW, vr, pN, ex1);  // d_pre_s
    // features: one 32-row block (15 real rows), then both halves need all
    f32x16 f;
#pragma unroll
    for (int i = 0; i < 16; ++i) f[i] = 0.f;
    {
      Op16 xs[4];
      const int exs = scaled_split64(e, xs);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) f = mma3(L16.A(gA, mST + kb), xs[kb], f);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = __builtin_amdgcn_ldexpf(f[i], exs);
    }
    float dfeat[kNF], gs[12];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float own = f[i], oth = other_half(own);
      dfeat[rrow(i)] = hi ? oth : own;
      if (rrow(i) + 4 < kNF) dfeat[rrow(i) + 4 < kNF ? rrow(i) + 4 : 0] = hi ? own : oth;
    }
    quad_features_adjoint(sc, t, dfeat, gs);
#pragma unroll
    for (int i = 0; i < 12; ++i) lam[i] += gs[i];
    // fc1 inputs, conv outputs: five 32-row blocks over e = ch*8 + pos
    float dpos[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int eb = 0; eb < 5; ++eb) {
      f32x16 y;
#pragma unroll
      for (int i = 0; i < 16; ++i) y[i] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) y = mma3(L16.A(gA, m1cT + eb * 4 + kb), x[kb], y);
#pragma unroll
      for (int i = 0; i < 16; ++i) y[i] = __builtin_amdgcn_ldexpf(y[i], ex1);
      const unsigned mws = hi ? mw[eb] >> 4 : mw[eb];  // bit r(i) + 4 hi -> bit r(i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // registers 4g..4g+3: channel eb*4 + g,
        const int ch = eb * 4 + g;     // positions ii + 4 hi
        float sum = 0.f;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int i = 4 * g + ii;
          const float dcp = ((mws >> rrow(i)) & 1u) ? y[i] : 0.f;
          dgn[ch][ii] += dcp;
          sum += dcp;
        }
        // diagonal tau = k + 3 is complete; the others move up one position
        Pdc.st(vg, (unsigned)(ch * 2 * kTau + k + 3) * pB, dgn[ch][3]);
        dgn[ch][3] = dgn[ch][2], dgn[ch][2] = dgn[ch][1], dgn[ch][1] = dgn[ch][0];
        dgn[ch][0] = 0.f;
        Pdc.st(vb_lo, (unsigned)(kConvP + ch * kH + k) * pB, sum + other_half(sum));
#pragma unroll
        for (int q = 0; q < 3; ++q)
          dpos[q] = fmaf(L.U(gAq + (eb * 4 + g) * 3 + q), sum, dpos[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) lam[q] -= dpos[q] + other_half(dpos[q]);
  }
  // the diagonals tau = 0..2 (after the last shift they sit in slots 1..3)
#pragma unroll
  for (int ch = 0; ch < kNC; ++ch)
#pragma unroll
    for (int tau = 0; tau < 3; ++tau)
      Pdc.st(vg, (unsigned)(ch * 2 * kTau + tau) * pitchB, dgn[ch][tau + 1]);
  if (st_lo && A.grad_state0)
#pragma unroll
    for (int i = 0; i < 12; ++i) A.grad_state0[(size_t)i * B + b] = lam[i];
  write_wave_partial(A.loss_partials, st_lo ? loss : 0.f);
}

__global__ __launch_bounds__(kThreads) void mlp_concurrent_bwd_kernel(ConcArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  fill_lds(lds, A.tables, kCbLds);
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const LdsView L(lds, lane);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = (blockIdx.x * (kThreads / 64) + wave) * 32 + (lane & 31);
  const int B = A.B;
  const bool live = b < B;
  const unsigned pN = (unsigned)B * 4u;
  const Planes Px1(A.x1, kN1, pN), Ph(A.h, 3 * kW, pN), Pmk(A.mask, 5, pN);
  const Planes Pdz(A.d_zout, kNA, pN), Pdp(A.d_pre, 4 * kW, pN);
  const Planes Pdc(A.d_conv, kNC * kNP, pN);
  const unsigned vb = live ? (unsigned)b * 4u : kDead;
  const unsigned vr = live ? vb + (hi ? 4u * pN : 0u) : kDead;

  float dzr[20];
#pragma unroll
  for (int cc = 0; cc < 20; ++cc) dzr[cc] = Pdz.ld(vr, khead(cc, 0) * pN);
  unsigned mw[5];
#pragma unroll
  for (int eb = 0; eb < 5; ++eb) mw[eb] = Pmk.ldu(vb, eb * pN);
  float hv[2][16];
  load_acts(hv, Ph, 2 * kW, vr, pN);  // h3
  __builtin_amdgcn_sched_barrier(0);
  // the reverse layers on the 16-bit matrix pipe: cotangents scaled per
  // trajectory, split into two fp16 terms, three products per k-block
  const LdsView16 L16(lds, lane);
  f32x16 d[2], e[2];
  zero(d);
  int ex;
  {  // dL/dh3 = W_out^T dL/dz: this lane's 20 rows fill 2.5 k-blocks
    float amax = 0.f;
#pragma unroll
    for (int cc = 0; cc < 20; ++cc) amax = fmaxf(amax, fabsf(dzr[cc]));
    ex = scale_exponent(amax);
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[j] = kb * 8 + j < 20 ? __builtin_amdgcn_ldexpf(dzr[kb * 8 + j < 20 ? kb * 8 + j : 0], -ex)
                               : 0.f;
      const Op16 x = split8(v);
      d[0] = mma3(L16.A(gA, mOT + kb), x, d[0]);
      d[1] = mma3(L16.A(gA, mOT + 3 + kb), x, d[1]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  tanh_adjoint(d, hv, Pdp, 2 * kW, vr, pN, ex);  // d_pre3
  load_acts(hv, Ph, kW, vr, pN);                 // h2
  __builtin_amdgcn_sched_barrier(0);
  Op16 x[4];
  zero(e);
  ex = scaled_split64(d, x);
  dense64T_16(e, x, L16, gA, m3T);
  __builtin_amdgcn_sched_barrier(0);
  tanh_adjoint(e, hv, Pdp, kW, vr, pN, ex);      // d_pre2
  load_acts(hv, Ph, 0, vr, pN);                  // h1
  __builtin_amdgcn_sched_barrier(0);
  zero(d);
  ex = scaled_split64(e, x);
  dense64T_16(d, x, L16, gA, m2T);
  __builtin_amdgcn_sched_barrier(0);
  tanh_adjoint(d, hv, Pdp, 0, vr, pN, ex);       // d_pre1
  load_acts(hv, Px1, 0, vr, pN);                 // s1
  __builtin_amdgcn_sched_barrier(0);
  zero(e);
  ex = scaled_split64(d, x);                     // d_pre1 feeds both fc1^T parts
  dense64T_16(e, x, L16, gA, m1sT);
  __builtin_amdgcn_sched_barrier(0);
  tanh_adjoint(e, hv, Pdp, 3 * kW, vr, pN, ex);  // d_pre_s
  // conv outputs (the network inputs carry no gradient in this mode)
#pragma unroll
  for (int eb = 0; eb < 5; ++eb) {
    f32x16 y;
#pragma unroll
    for (int i = 0; i < 16; ++i) y[i] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) y = mma3(L16.A(gA, m1cT + eb * 4 + kb), x[kb], y);
    const unsigned mws = hi ? mw[eb] >> 4 : mw[eb];
#pragma unroll
    for (int i = 0; i < 16; ++i)
      Pdc.st(vr, (eb * 32 + rrow(i)) * pN,
             ((mws >> rrow(i)) & 1u) ? __builtin_amdgcn_ldexpf(y[i], ex) : 0.f);
  }
}

}  // namespace
}  // namespace apg

using namespace apg;

extern "C" {

int apg_quad_mlp_rollout_bwd(const float *state0, const float *states,
                             const float *actions, const float *ref,
                             int ref_cols, const float *x1, const float *h,
                             const unsigned *relu_mask, float dt,
                             const ApgQuadParams *params,
                             const ApgQuadLossWeights *weights,
                             const ApgMlpPolicy *policy, int B, int H,
                             float *loss_partials, float *loss, float *d_pre,
                             float *d_zout, float *d_conv, float *grad_state0,
                             float *workspace, apg_stream_t stream) {
  if (int e = check_mlp(params, policy, B, H)) return e;
  if (!weights) { set_error("weights is NULL"); return APG_ERR_ARG; }
  if (ref_cols != 9 && ref_cols != 6) {
    set_error("ref_cols must be 9 or 6");
    return APG_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    if (loss && hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess)
      return check_launch("memset(loss)");
    return APG_OK;
  }
  if (!state0 || !states || !actions || !ref || !x1 || !h || !relu_mask ||
      !loss_partials || !d_pre || !d_zout || !d_conv || !workspace) {
    set_error("NULL buffer");
    return APG_ERR_ARG;
  }
  static PerDeviceOnce attr;
  if (!attr.test()) {
    if (int e = raise_lds(mlp_rollout_bwd_kernel, kCbLds)) return e;
    attr.set();
  }
  BwdArgs A;
  A.state0 = state0, A.states = states, A.actions = actions, A.ref = ref;
  A.x1 = x1, A.h = h, A.mask = relu_mask;
  A.loss_partials = loss_partials, A.d_pre = d_pre, A.d_zout = d_zout;
  A.d_conv = d_conv, A.grad_state0 = grad_state0;
  A.tables = workspace;
  A.c = make_const(*params, dt);
  A.w = *weights;
  PackArgs P;
  P.pol = *policy, P.dst = workspace, P.head_rows = 4;
  hipLaunchKernelGGL(mlp_pack_cbwd_kernel, dim3((kCbLds + 255) / 256), dim3(256),
                     0, st, P);
  A.B = B, A.ref_cols = ref_cols, A.vel_col = ref_cols == 9 ? 6 : 3;
  const int blocks = (B + kTrajPerBlock - 1) / kTrajPerBlock;
  hipLaunchKernelGGL(mlp_rollout_bwd_kernel, dim3(blocks), dim3(kThreads),
                     kCbLds * sizeof(float), st, A);
  if (int e = check_launch("quad_mlp_rollout_bwd")) return e;
  if (loss)
    return launch_reduce_partials(loss_partials, blocks * (kThreads / kWave),
                                  loss, st);
  return APG_OK;
}

int apg_quad_mlp_concurrent_workspace_floats(void) {
  return kCfLds + kCbLds;   // forward and reverse tables, packed by one launch
}

int apg_quad_mlp_concurrent_fwd_bwd(
    const float *feat, const float *in_ref, const float *state0, const float *ref,
    int ref_cols, float dt, const ApgQuadParams *params,
    const ApgQuadLossWeights *weights, const ApgMlpPolicy *policy, int B, int H,
    float *x1, float *h, unsigned *relu_mask, float *d_zout, float *d_pre,
    float *d_conv, float *loss_partials, float *loss, float *states,
    float *workspace, apg_stream_t stream) {
  if (int e = check_mlp(params, policy, B, H)) return e;
  if (!weights) { set_error("weights is NULL"); return APG_ERR_ARG; }
  if (ref_cols != 9 && ref_cols != 6) {
    set_error("ref_cols must be 9 or 6");
    return APG_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    if (loss && hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess)
      return check_launch("memset(loss)");
    return APG_OK;
  }
  if (!feat || !in_ref || !state0 || !ref || !x1 || !h || !relu_mask || !d_zout ||
      !d_pre || !d_conv || !loss_partials || !workspace) {
    set_error("NULL buffer");
    return APG_ERR_ARG;
  }
  static PerDeviceOnce attr;
  if (!attr.test()) {
    if (int e = raise_lds(mlp_concurrent_fwd_kernel<false>, kCfLds)) return e;
    if (int e = raise_lds(mlp_concurrent_bwd_kernel, kCbLds)) return e;
    attr.set();
  }
  ConcArgs A;
  A.feat = feat, A.in_ref = in_ref, A.state0 = state0, A.ref = ref;
  A.x1 = x1, A.h = h, A.mask = relu_mask, A.d_zout = d_zout, A.d_pre = d_pre;
  A.d_conv = d_conv, A.states = states, A.loss_partials = loss_partials;
  A.tables = workspace;
  A.xmax = nullptr;
  A.c = make_const(*params, dt);
  A.w = *weights;
  A.B = B, A.ref_cols = ref_cols, A.vel_col = ref_cols == 9 ? 6 : 3;
  PackArgs P;
  P.pol = *policy, P.dst = workspace, P.head_rows = kNA;
  const int blocks = (B + kTrajPerBlock - 1) / kTrajPerBlock;
  const int fwd_blocks = (kCfLds + 255) / 256, bwd_blocks = (kCbLds + 255) / 256;
  hipLaunchKernelGGL(mlp_pack_pair_kernel, dim3(fwd_blocks + bwd_blocks), dim3(256), 0, st,
                     P, fwd_blocks);
  hipLaunchKernelGGL(mlp_concurrent_fwd_kernel<false>, dim3(blocks), dim3(kThreads),
                     kCfLds * sizeof(float), st, A);
  A.tables = workspace + kCfLds;
  hipLaunchKernelGGL(mlp_concurrent_bwd_kernel, dim3(blocks), dim3(kThreads),
                     kCbLds * sizeof(float), st, A);
  if (int e = check_launch("quad_mlp_concurrent_fwd_bwd")) return e;
  if (loss)
    return launch_reduce_partials(loss_partials, blocks * (kThreads / kWave), loss, st);
  return APG_OK;
}

}  // extern "C"
