// A/B harness without Python: loads several builds of libapg_hip.so, runs the
// bench workload (quadrotor rollout fwd+bwd, packed 6-column reference)
// through apg_quad_rollout_fwd_bwd of each, compares every output with the
// first library's and times the launches with HIP events over rotating buffer
// sets.
//   tools/ab_quad <lib0.so>[:p] <lib1.so>[:p] [...]    (env AB_B, AB_H, AB_SETS,
//   AB_ITERS)
// A ":p" suffix runs that library through APG_LAYOUT_PACKED (the harness
// converts inputs and outputs; values are compared in plane order), otherwise
// APG_LAYOUT_SOA.  A library built with -DAPG_STAMP (it exports
// apg_debug_set_stamps) also gets its per-wave s_memtime phase table printed:
// medians over the waves of
//   launch->arguments | ->end of forward step 0 | forward steps 1.. |
//   reverse sweep first half | second half | store drain
// Build: hipcc -O2 -I include tools/ab_quad.cpp -o tools/exp/ab_quad -ldl
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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
typedef int (*stamp_fn)(unsigned long long *);

static unsigned g_seed = 12345u;
static float urand() {  // [0, 1)
  g_seed = g_seed * 1664525u + 1013904223u;
  return (g_seed >> 8) * (1.0f / 16777216.0f);
}

struct Lib {
  void *h;
  rollout_fn run;
  err_fn err;
  stamp_fn stamps;
  std::string path;
  bool packed;
};

static bool open_lib(const char *arg, Lib *l) {
  l->path = arg;
  l->packed = false;
  if (l->path.size() > 2 && l->path.substr(l->path.size() - 2) == ":p") {
    l->packed = true;
    l->path.resize(l->path.size() - 2);
  }
  l->h = dlopen(l->path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!l->h) {
    std::printf("{\"error\": \"dlopen %s: %s\"}\n", l->path.c_str(), dlerror());
    return false;
  }
  l->run = (rollout_fn)dlsym(l->h, "apg_quad_rollout_fwd_bwd");
  l->err = (err_fn)dlsym(l->h, "apg_last_error_string");
  l->stamps = (stamp_fn)dlsym(l->h, "apg_debug_set_stamps");
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

// planes [G*W][B]  <->  rows [G][B][W]
static std::vector<float> to_rows(const std::vector<float> &pl, int G, int W, int B) {
  std::vector<float> r((size_t)G * W * B);
  for (int g = 0; g < G; ++g)
    for (int b = 0; b < B; ++b)
      for (int i = 0; i < W; ++i)
        r[((size_t)g * B + b) * W + i] = pl[((size_t)g * W + i) * B + b];
  return r;
}
static std::vector<float> to_planes(const std::vector<float> &r, int G, int W, int B) {
  std::vector<float> pl((size_t)G * W * B);
  for (int g = 0; g < G; ++g)
    for (int b = 0; b < B; ++b)
      for (int i = 0; i < W; ++i)
        pl[((size_t)g * W + i) * B + b] = r[((size_t)g * B + b) * W + i];
  return pl;
}

// AB_ARENA=1: the per-set tensors come out of ONE hipMalloc (2 MiB-aligned
// slices) instead of one hipMalloc each - page-table fragment experiment.
static char *g_arena = nullptr;
static size_t g_arena_off = 0, g_arena_size = 0;
static hipError_t arena_alloc(void **p, size_t bytes) {
  static const bool on = std::getenv("AB_ARENA") && std::atoi(std::getenv("AB_ARENA"));
  if (!on) return hipMalloc(p, bytes);
  if (!g_arena) {
    g_arena_size = (size_t)3 << 30;
    hipError_t e = hipMalloc((void **)&g_arena, g_arena_size);
    if (e != hipSuccess) return e;
  }
  const size_t al = (size_t)2 << 20;
  g_arena_off = (g_arena_off + al - 1) / al * al;
  if (g_arena_off + bytes > g_arena_size) return hipErrorOutOfMemory;
  *p = g_arena + g_arena_off;
  g_arena_off += bytes;
  return hipSuccess;
}

int main(int argc, char **argv) {
  if (argc < 3) {
    std::printf("usage: ab_quad lib0[:p] lib1[:p] [lib2 ...]\n");
    return 1;
  }
  auto env = [](const char *k, int d) {
    const char *v = std::getenv(k);
    return v ? std::atoi(v) : d;
  };
  const int B = env("AB_B", 65536), H = env("AB_H", 10), sets = env("AB_SETS", 8),
            iters = env("AB_ITERS", 200);
  // AB_NO_STATES=1: the comparison runs ask for no states_out (the two-role
  // packed kernel is only selected then); the states column compares zeros
  const bool no_states = env("AB_NO_STATES", 0) != 0;
  const int nl = argc - 1;
  std::vector<Lib> L(nl);
  bool any_packed = false;
  for (int i = 0; i < nl; ++i) {
    if (!open_lib(argv[1 + i], &L[i])) return 1;
    any_packed |= L[i].packed;
  }

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

  // device buffer sets in plane order and (if asked for) in row order
  std::vector<float *> s0[2], act[2], ref[2], ga(sets);
  std::vector<float> h_s0, h_act, h_ref;  // host copy of set 0 (plane order)
  for (int lay = 0; lay < (any_packed ? 2 : 1); ++lay)
    s0[lay].resize(sets), act[lay].resize(sets), ref[lay].resize(sets);
  float *gs, *so, *part, *loss;
  for (int s = 0; s < sets; ++s) {
    std::vector<float> hs(nS), ha(nA), hr(nR);
    for (size_t i = 0; i < nS; ++i) {
      const int row = (int)(i / B);
      const float u = urand() - 0.5f;
      hs[i] = row < 3 ? 0.f : row < 6 ? 0.4f * u : row < 9 ? 3.f * u : 0.2f * u;
    }
    for (size_t i = 0; i < nA; ++i) ha[i] = 0.05f + 0.9f * urand();
    for (size_t i = 0; i < nR; ++i) {
      const int k = (int)(i / ((size_t)6 * B));
      hr[i] = (urand() - 0.5f) * 0.3f * (k + 1);
    }
    for (int lay = 0; lay < (any_packed ? 2 : 1); ++lay) {
      CK(arena_alloc((void **)&s0[lay][s], nS * 4));
      CK(arena_alloc((void **)&act[lay][s], nA * 4));
      CK(arena_alloc((void **)&ref[lay][s], nR * 4));
      const std::vector<float> a = lay ? to_rows(hs, 3, 4, B) : hs,
                               b = lay ? to_rows(ha, H, 4, B) : ha,
                               c = lay ? to_rows(hr, H, 6, B) : hr;
      CK(hipMemcpy(s0[lay][s], a.data(), nS * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(act[lay][s], b.data(), nA * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(ref[lay][s], c.data(), nR * 4, hipMemcpyHostToDevice));
    }
    CK(arena_alloc((void **)&ga[s], nA * 4));
    if (s == 0) h_s0 = hs, h_act = ha, h_ref = hr;
  }
  CK(hipMalloc(&gs, nS * 4));
  CK(hipMalloc(&so, nO * 4));
  CK(hipMalloc(&part, nP * 4));
  CK(hipMalloc(&loss, 4));
  float *x_s0, *x_act, *x_ref;  // scratch inputs of the comparison runs
  CK(hipMalloc(&x_s0, nS * 4));
  CK(hipMalloc(&x_act, nA * 4));
  CK(hipMalloc(&x_ref, nR * 4));

  // ---- outputs of every build: full batch and a ragged one, both parameter
  // sets.  A shorter batch re-interprets set 0's numbers as [rows][Bq] planes.
  double worst = 0;
  std::printf("{\"B\": %d, \"H\": %d, \"checks\": [", B, H);
  bool first = true;
  for (int m = 0; m < 2; ++m)
    for (int rag = 0; rag < 2; ++rag) {
      const int Bq = rag ? B - 37 : B;
      std::vector<float> q_s0(h_s0.begin(), h_s0.begin() + (size_t)12 * Bq),
          q_act(h_act.begin(), h_act.begin() + (size_t)H * 4 * Bq),
          q_ref(h_ref.begin(), h_ref.begin() + (size_t)H * 6 * Bq);
      std::vector<std::vector<float>> out0(4);
      for (int l = 0; l < nl; ++l) {
        const bool pk = L[l].packed;
        const std::vector<float> a = pk ? to_rows(q_s0, 3, 4, Bq) : q_s0,
                                 b = pk ? to_rows(q_act, H, 4, Bq) : q_act,
                                 c = pk ? to_rows(q_ref, H, 6, Bq) : q_ref;
        CK(hipMemcpy(x_s0, a.data(), a.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(x_act, b.data(), b.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(x_ref, c.data(), c.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(ga[0], 0, nA * 4));
        CK(hipMemset(gs, 0, nS * 4));
        CK(hipMemset(so, 0, nO * 4));
        const int rc = L[l].run(x_s0, x_act, x_ref, 6, dt, &par[m], &w, Bq, H,
                                pk ? APG_LAYOUT_PACKED : APG_LAYOUT_SOA, part, loss,
                                ga[0], gs, no_states ? nullptr : so, nullptr, nullptr);
        if (rc != 0) {
          std::printf("], \"error\": \"lib %d rc %d: %s\"}\n", l, rc,
                      L[l].err ? L[l].err() : "?");
          return 3;
        }
        CK(hipDeviceSynchronize());
        const size_t n[4] = {(size_t)H * 4 * Bq, (size_t)12 * Bq, (size_t)H * 12 * Bq, 1};
        float *src[4] = {ga[0], gs, so, loss};
        std::vector<float> outl[4];
        for (int t = 0; t < 4; ++t) {
          outl[t].resize(n[t]);
          CK(hipMemcpy(outl[t].data(), src[t], n[t] * 4, hipMemcpyDeviceToHost));
        }
        if (pk) {
          outl[0] = to_planes(outl[0], H, 4, Bq);
          outl[1] = to_planes(outl[1], 3, 4, Bq);
          outl[2] = to_planes(outl[2], H * 3, 4, Bq);
        }
        if (l == 0) {
          for (int t = 0; t < 4; ++t) out0[t] = outl[t];
          continue;
        }
        const double e[4] = {max_rel(out0[0], outl[0]), max_rel(out0[1], outl[1]),
                             max_rel(out0[2], outl[2]), max_rel(out0[3], outl[3])};
        for (int t = 0; t < 4; ++t) worst = std::fmax(worst, e[t]);
        std::printf("%s{\"lib\": %d, \"params\": %d, \"B\": %d, "
                    "\"grad_actions\": %.3g, \"grad_state0\": %.3g, "
                    "\"states\": %.3g, \"loss\": %.3g, \"loss0\": %.9g, "
                    "\"lossL\": %.9g}",
                    first ? "" : ", ", l, m, Bq, e[0], e[1], e[2], e[3],
                    (double)out0[3][0], (double)outl[3][0]);
        first = false;
      }
    }
  std::printf("], \"worst_rel_diff\": %.3g, \"us_per_launch\": [", worst);

  // ---- timing: interleaved rounds, rotating buffer sets, kernel only
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto launch = [&](int l, int i) {
    const int p = L[l].packed ? 1 : 0, s = i % sets;
    return L[l].run(s0[p][s], act[p][s], ref[p][s], 6, dt, &par[0], &w, B, H,
                    p ? APG_LAYOUT_PACKED : APG_LAYOUT_SOA, part, nullptr, ga[s],
                    nullptr, nullptr, nullptr, nullptr);
  };
  for (int round = 0; round < 3; ++round)
    for (int l = 0; l < nl; ++l) {
      for (int i = 0; i < 20; ++i) launch(l, i);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, nullptr));
      for (int i = 0; i < iters; ++i) launch(l, i);
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      std::printf("%s{\"lib\": %d, \"us\": %.3f}", (round || l) ? ", " : "", l,
                  ms * 1e3 / iters);
    }
  std::printf("], \"libs\": [");
  for (int l = 0; l < nl; ++l)
    std::printf("%s\"%s%s\"", l ? ", " : "", L[l].path.c_str(), L[l].packed ? ":p" : "");
  std::printf("], \"phases\": [");
  // ---- per-wave phase stamps of instrumented builds (8 x u64 per wave)
  bool firstp = true;
  for (int l = 0; l < nl; ++l) {
    if (!L[l].stamps) continue;
    const int nw = (B + 63) / 64;
    unsigned long long *dst;
    CK(hipMalloc(&dst, (size_t)nw * 8 * 8));
    CK(hipMemset(dst, 0, (size_t)nw * 8 * 8));
    L[l].stamps(dst);
    for (int i = 0; i < 24; ++i) launch(l, i);
    CK(hipDeviceSynchronize());
    L[l].stamps(nullptr);
    std::vector<unsigned long long> st((size_t)nw * 8);
    CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
    CK(hipFree(dst));
    if (const char *dump = std::getenv("AB_DUMP")) {  // raw stamps for offline analysis
      const std::string fn = std::string(dump) + "_lib" + std::to_string(l) + ".bin";
      if (FILE *f = std::fopen(fn.c_str(), "wb")) {
        std::fwrite(st.data(), 8, st.size(), f);
        std::fclose(f);
      }
    }
    auto med = [&](int a, int b) {
      std::vector<double> d(nw);
      for (int i = 0; i < nw; ++i) d[i] = (double)(st[i * 8 + b] - st[i * 8 + a]);
      std::sort(d.begin(), d.end());
      return d[nw / 2];
    };
    // distribution of the wave lifetimes, and of start / end on the (per-XCD?)
    // s_memtime clock relative to the earliest start
    std::vector<double> life(nw), t_start(nw), t_end(nw);
    unsigned long long t0 = ~0ull;
    for (int i = 0; i < nw; ++i) t0 = std::min(t0, st[i * 8]);
    for (int i = 0; i < nw; ++i) {
      life[i] = (double)(st[i * 8 + 4] - st[i * 8]);
      t_start[i] = (double)(st[i * 8] - t0), t_end[i] = (double)(st[i * 8 + 4] - t0);
    }
    std::sort(life.begin(), life.end());
    std::sort(t_start.begin(), t_start.end());
    std::sort(t_end.begin(), t_end.end());
    auto pc = [&](const std::vector<double> &v, double q) { return v[(size_t)(q * (nw - 1))]; };
    std::printf("%s{\"lib\": %d, \"args\": %.0f, \"first_step\": %.0f, "
                "\"fwd_rest\": %.0f, \"reverse_a\": %.0f, \"reverse_b\": %.0f, "
                "\"drain\": %.0f, \"wave_total\": %.0f, "
                "\"life_p10_p90_max\": [%.0f, %.0f, %.0f], "
                "\"start_p50_p90_max\": [%.0f, %.0f, %.0f], "
                "\"end_p10_p50_max\": [%.0f, %.0f, %.0f]}",
                firstp ? "" : ", ", l, 0.0, med(0, 1), med(1, 2), med(2, 6),
                med(6, 3), med(3, 4), med(0, 4), pc(life, .1), pc(life, .9),
                pc(life, 1.), pc(t_start, .5), pc(t_start, .9), pc(t_start, 1.),
                pc(t_end, .1), pc(t_end, .5), pc(t_end, 1.));
    firstp = false;
  }
  std::printf("]}\n");
  return 0;
}
