// A/B harness without Python: loads two builds of libapg_hip.so, runs the
// bench workload (quadrotor rollout fwd+bwd, SoA, packed 6-column reference)
// through apg_quad_rollout_fwd_bwd of each, compares every output and times
// the launches with HIP events over rotating buffer sets.
//   tools/ab_quad <libA.so> <libB.so> [B=65536] [H=10] [sets=8] [iters=200]
// Build: hipcc -O2 -I include tools/ab_quad.cpp -o tools/exp/ab_quad -ldl
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "apg.h"

#define CK(x)                                                              \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      std::printf("{\"error\": \"%s at %s:%d\"}\n", hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                     \
      return 2;                                                            \
    }                                                                      \
  } while (0)

typedef int (*rollout_fn)(const float *, const float *, const float *, int, float,
                          const ApgQuadParams *, const ApgQuadLossWeights *, int,
                          int, int, float *, float *, float *, float *, float *,
                          const ApgDeferredLoss *, apg_stream_t);
typedef const char *(*err_fn)(void);

static unsigned g_seed = 12345u;
static float urand() {  // [0, 1)
  g_seed = g_seed * 1664525u + 1013904223u;
  return (g_seed >> 8) * (1.0f / 16777216.0f);
}

struct Lib {
  void *h;
  rollout_fn run;
  err_fn err;
};

static bool open_lib(const char *path, Lib *l) {
  l->h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!l->h) {
    std::printf("{\"error\": \"dlopen %s: %s\"}\n", path, dlerror());
    return false;
  }
  l->run = (rollout_fn)dlsym(l->h, "apg_quad_rollout_fwd_bwd");
  l->err = (err_fn)dlsym(l->h, "apg_last_error_string");
  return l->run != nullptr;
}

static double max_rel(const std::vector<float> &a, const std::vector<float> &b) {
  double num = 0, den = 1e-30;
  for (size_t i = 0; i < a.size(); ++i) {
    num = std::fmax(num, std::fabs((double)a[i] - b[i]));
    den = std::fmax(den, std::fabs((double)a[i]));
    if (std::isnan(a[i]) != std::isnan(b[i])) return 1e30;
  }
  return num / den;
}

int main(int argc, char **argv) {
  if (argc < 3) {
    std::printf("usage: ab_quad libA libB [B] [H] [sets] [iters]\n");
    return 1;
  }
  const int B = argc > 3 ? std::atoi(argv[3]) : 65536;
  const int H = argc > 4 ? std::atoi(argv[4]) : 10;
  const int sets = argc > 5 ? std::atoi(argv[5]) : 8;
  const int iters = argc > 6 ? std::atoi(argv[6]) : 200;
  Lib L[2];
  if (!open_lib(argv[1], &L[0]) || !open_lib(argv[2], &L[1])) return 1;

  ApgQuadParams par[2];
  for (int m = 0; m < 2; ++m) {
    const float mass = m ? 1.0f : 0.723f, scale = mass / 12.0f * 0.31f * 0.31f;
    const float fi[3] = {4.5f, 4.5f, 7.0f}, kv[3] = {16.6f, 16.6f, 5.0f};
    par[m].mass = mass;
    for (int i = 0; i < 3; ++i) {
      par[m].kinv[i] = kv[i], par[m].inertia[i] = scale * fi[i];
      par[m].gravity[i] = i == 2 ? -9.81f : 0.f;
      par[m].trans_drag[i] = m ? 0.1f * (i + 1) : 0.f;
      par[m].rot_drag[i] = m ? 0.01f * (i + 1) : 0.f;
    }
  }
  const ApgQuadLossWeights w = {10.f, 1.f, 0.1f, 0.1f, 5.f};
  const float dt = 0.1f;
  const size_t nS = (size_t)12 * B, nA = (size_t)H * 4 * B, nR = (size_t)H * 6 * B,
               nO = (size_t)H * 12 * B, nP = (size_t)(B + 63) / 64;

  std::vector<float *> s0(sets), act(sets), ref(sets), ga(sets);
  float *gs, *so, *part, *loss;
  std::vector<float> h;
  for (int s = 0; s < sets; ++s) {
    CK(hipMalloc(&s0[s], nS * 4));
    CK(hipMalloc(&act[s], nA * 4));
    CK(hipMalloc(&ref[s], nR * 4));
    CK(hipMalloc(&ga[s], nA * 4));
    h.resize(nS);
    for (size_t i = 0; i < nS; ++i) {
      const int row = (int)(i / B);
      const float u = urand() - 0.5f;
      h[i] = row < 3 ? 0.f : row < 6 ? 0.4f * u : row < 9 ? 3.f * u : 0.2f * u;
    }
    CK(hipMemcpy(s0[s], h.data(), nS * 4, hipMemcpyHostToDevice));
    h.resize(nA);
    for (size_t i = 0; i < nA; ++i) h[i] = 0.05f + 0.9f * urand();
    CK(hipMemcpy(act[s], h.data(), nA * 4, hipMemcpyHostToDevice));
    h.resize(nR);
    for (size_t i = 0; i < nR; ++i) {
      const int k = (int)(i / ((size_t)6 * B));
      h[i] = (urand() - 0.5f) * 0.3f * (k + 1);
    }
    CK(hipMemcpy(ref[s], h.data(), nR * 4, hipMemcpyHostToDevice));
  }
  CK(hipMalloc(&gs, nS * 4));
  CK(hipMalloc(&so, nO * 4));
  CK(hipMalloc(&part, nP * 4));
  CK(hipMalloc(&loss, 4));

  // ---- outputs of both builds: full batch and a ragged one, both parameter sets
  double worst = 0;
  std::printf("{\"B\": %d, \"H\": %d, \"checks\": [", B, H);
  bool first = true;
  for (int m = 0; m < 2; ++m)
    for (int rag = 0; rag < 2; ++rag) {
      const int Bq = rag ? B - 37 : B;
      std::vector<float> out[2][4];
      for (int l = 0; l < 2; ++l) {
        CK(hipMemset(ga[0], 0, nA * 4));
        CK(hipMemset(gs, 0, nS * 4));
        CK(hipMemset(so, 0, nO * 4));
        // a shorter batch re-interprets the same buffers as [rows][Bq] planes
        const int rc = L[l].run(s0[0], act[0], ref[0], 6, dt, &par[m], &w, Bq, H,
                                APG_LAYOUT_SOA, part, loss, ga[0], gs, so, nullptr,
                                nullptr);
        if (rc != 0) {
          std::printf("], \"error\": \"lib %d rc %d: %s\"}\n", l, rc,
                      L[l].err ? L[l].err() : "?");
          return 3;
        }
        CK(hipDeviceSynchronize());
        const size_t n[4] = {(size_t)H * 4 * Bq, (size_t)12 * Bq, (size_t)H * 12 * Bq, 1};
        float *src[4] = {ga[0], gs, so, loss};
        for (int t = 0; t < 4; ++t) {
          out[l][t].resize(n[t]);
          CK(hipMemcpy(out[l][t].data(), src[t], n[t] * 4, hipMemcpyDeviceToHost));
        }
      }
      const double e[4] = {max_rel(out[0][0], out[1][0]), max_rel(out[0][1], out[1][1]),
                           max_rel(out[0][2], out[1][2]), max_rel(out[0][3], out[1][3])};
      for (int t = 0; t < 4; ++t) worst = std::fmax(worst, e[t]);
      std::printf("%s{\"params\": %d, \"B\": %d, \"grad_actions\": %.3g, "
                  "\"grad_state0\": %.3g, \"states\": %.3g, \"loss\": %.3g, "
                  "\"lossA\": %.9g, \"lossB\": %.9g}",
                  first ? "" : ", ", m, Bq, e[0], e[1], e[2], e[3],
                  (double)out[0][3][0], (double)out[1][3][0]);
      first = false;
    }
  std::printf("], \"worst_rel_diff\": %.3g, \"us_per_launch\": [", worst);

  // ---- timing: interleaved rounds, rotating buffer sets, kernel only
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int round = 0; round < 3; ++round)
    for (int l = 0; l < 2; ++l) {
      for (int i = 0; i < 20; ++i)
        L[l].run(s0[i % sets], act[i % sets], ref[i % sets], 6, dt, &par[0], &w, B, H,
                 APG_LAYOUT_SOA, part, nullptr, ga[i % sets], nullptr, nullptr,
                 nullptr, nullptr);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, nullptr));
      for (int i = 0; i < iters; ++i)
        L[l].run(s0[i % sets], act[i % sets], ref[i % sets], 6, dt, &par[0], &w, B, H,
                 APG_LAYOUT_SOA, part, nullptr, ga[i % sets], nullptr, nullptr,
                 nullptr, nullptr);
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      std::printf("%s{\"lib\": \"%c\", \"us\": %.3f}", (round || l) ? ", " : "",
                  'A' + l, ms * 1e3 / iters);
    }
  std::printf("]}\n");
  return 0;
}
