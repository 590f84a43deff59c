// op_bench.cpp -- torch-free A/B harness for the non-contraction kernels of the hot path (flash attention, SAM window
// attention, GroupNorm, LayerNorm), the companion of tools/gemm_bench.cpp: starts in well under a second on the GPU box.
//
//   hipcc -O2 -std=c++17 -mf16c tools/op_bench.cpp -o tools/op_bench -ldl
//   tools/op_bench <lib.so>[,<lib2.so>...] [--cases all|attn|gn|ln|<substring>] [--iters 20] [--rounds 3] [--check] [--out f.jsonl]
//
// One JSON line per case x library: microseconds per launch (HIP graph of `iters` launches, best / median of `rounds`
// interleaved replays), TFLOP/s (attention: 4*B*H*Nq*Nk*D) or GB/s (norms: algorithmic bytes), and with --check the max
// |difference| against an fp64 host evaluation of ONE (batch, head) slice / a few rows (the full-tensor parity lives in
// tests/test_kernels.py; this is a smoke check for experiment builds).  Several libraries = builds of the same sources
// with different -D flags.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../include/editanything_hip.h"

#define HIP_CHECK(x)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                            \
    }                                                                                     \
  } while (0)

typedef int (*attn_fn)(const void*, const void*, const void*, void*, int, int, int, int, int, long long, long long, long long,
                       long long, long long, long long, long long, long long, float, const float*, const float*, int, void*);
typedef int (*win_fn)(const void*, const void*, const void*, void*, int, int, int, int, long long, long long, long long,
                      long long, long long, long long, long long, long long, float, const void*, const void*, void*);
typedef size_t (*gnws_fn)(int, int, int, int);
typedef int (*gn_fn)(const void*, int, const void*, int, const void*, const float*, const float*, void*, int, int, int, float,
                     int, void*, size_t, void*);
typedef int (*ln_fn)(const void*, int, const float*, const float*, void*, int, int, float, void*);

struct Lib {
  std::string path;
  attn_fn attn;
  win_fn win;
  gnws_fn gnws;
  gn_fn gn;
  ln_fn ln;
};

enum Kind { ATTN, WIN, GN, LN };
struct Case {
  std::string name;
  Kind kind;
  int B, H, Nq, Nk, D;      // attention (fused-qkv layout: token stride 3*H*D for self-attention, H*D for cross K/V)
  int cross;                // 1: K/V from a separate [B][Nk][2*H*D] tensor (text tokens)
  int HW, C, c2, silu;      // groupnorm: [B][HW][C (+c2)]
  int M, in_f32;            // layernorm rows
  double flops, bytes;
};

static std::vector<Case> all_cases() {
  std::vector<Case> v;
  auto at = [&](int B, int H, int Nq, int Nk, int D, int cross) {
    Case c{};
    char b[128];
    snprintf(b, sizeof b, "attn B%d H%d Nq%d Nk%d D%d%s", B, H, Nq, Nk, D, cross ? " cross" : "");
    c.name = b; c.kind = ATTN; c.B = B; c.H = H; c.Nq = Nq; c.Nk = Nk; c.D = D; c.cross = cross;
    c.flops = 4.0 * B * H * Nq * Nk * D;
    v.push_back(c);
  };
  auto wn = [&](int nWin, int H, int S, int D) {
    Case c{};
    char b[128];
    snprintf(b, sizeof b, "attn-window W%d H%d S%d D%d", nWin, H, S, D);
    c.name = b; c.kind = WIN; c.B = nWin; c.H = H; c.Nq = S; c.D = D;
    c.flops = 4.0 * nWin * H * (double)(S * S) * (S * S) * D;
    v.push_back(c);
  };
  auto gn = [&](int B, int HW, int C, int c2, int silu) {
    Case c{};
    char b[128];
    snprintf(b, sizeof b, "groupnorm B%d HW%d C%d+%d silu%d", B, HW, C, c2, silu);
    c.name = b; c.kind = GN; c.B = B; c.HW = HW; c.C = C; c.c2 = c2; c.silu = silu;
    c.bytes = 2.0 * B * HW * (C + c2) * 2;      // one read + one write (the single-pass form); the two-pass form reads twice
    v.push_back(c);
  };
  auto ln = [&](int M, int C, int in_f32) {
    Case c{};
    char b[128];
    snprintf(b, sizeof b, "layernorm M%d C%d in%s", M, C, in_f32 ? "f32" : "f16");
    c.name = b; c.kind = LN; c.M = M; c.C = C; c.in_f32 = in_f32;
    c.bytes = (double)M * C * (in_f32 ? 4 : 2) + (double)M * C * 2;
    v.push_back(c);
  };
  // one ControlNet + UNet evaluation at network batch 8 (profiles/r01_eval_breakdown_v3.json)
  at(8, 5, 4096, 4096, 64, 0);  at(8, 10, 1024, 1024, 64, 0);  at(8, 20, 256, 256, 64, 0);  at(8, 20, 64, 64, 64, 0);
  at(8, 5, 4096, 77, 64, 1);    at(8, 10, 1024, 77, 64, 1);    at(8, 20, 256, 77, 64, 1);
  // SD1.5 head dims (config 4) and the 128x128-latent stress shape (config 5)
  at(8, 8, 4096, 4096, 40, 0);  at(2, 5, 16384, 16384, 64, 0);
  // workgroup-count sweep around the level-0 shape (1280 workgroups = 5 per CU): 4 and 6 per CU
  at(8, 4, 4096, 4096, 64, 0);  at(8, 6, 4096, 4096, 64, 0);
  // the shared CFG prefix runs the first level-0 self-attention on ONE copy of the batch: 640 workgroups
  at(4, 5, 4096, 4096, 64, 0);
  // SAM ViT-H: global attention without the bias tables (4 images), fused window attention (100 windows)
  at(4, 16, 4096, 4096, 80, 0);
  wn(100, 16, 14, 80);
  gn(8, 4096, 320, 0, 1);   gn(8, 1024, 640, 0, 1);   gn(8, 256, 1280, 0, 1);   gn(8, 64, 1280, 0, 1);
  gn(8, 4096, 320, 320, 1); gn(8, 1024, 640, 640, 1); gn(8, 64, 1280, 1280, 1);
  gn(4, 262144, 128, 0, 1);                                       // VAE decoder 512^2 level
  ln(32768, 320, 0);  ln(8192, 640, 0);  ln(2048, 1280, 0);  ln(512, 1280, 0);  ln(16384, 1280, 1);   // last: SAM fp32 residual stream
  return v;
}

static uint32_t rng_state = 4242u;
static inline float urand() {
  rng_state = rng_state * 1664525u + 1013904223u;
  return ((rng_state >> 8) * (1.0f / 8388608.0f)) - 1.0f;
}
static std::vector<_Float16> host_f16(size_t n, float scale) {
  std::vector<_Float16> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (_Float16)(urand() * scale);
  return h;
}
template <typename T>
static void* to_dev(const std::vector<T>& h) {
  void* d;
  HIP_CHECK(hipMalloc(&d, h.size() * sizeof(T)));
  HIP_CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

static std::vector<std::string> split(const std::string& s, char sep) {
  std::vector<std::string> out;
  size_t p = 0;
  while (true) {
    size_t q = s.find(sep, p);
    out.push_back(s.substr(p, q == std::string::npos ? q : q - p));
    if (q == std::string::npos) break;
    p = q + 1;
  }
  return out;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s lib.so[,lib2.so] [--cases s] [--iters n] [--rounds n] [--check] [--out f]\n", argv[0]);
    return 1;
  }
  std::string cases_sel = "all", out_path;
  int iters = 20, rounds = 3, check = 0;
  for (int i = 2; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "--cases" && i + 1 < argc) cases_sel = argv[++i];
    else if (a == "--iters" && i + 1 < argc) iters = atoi(argv[++i]);
    else if (a == "--rounds" && i + 1 < argc) rounds = atoi(argv[++i]);
    else if (a == "--check") check = 1;
    else if (a == "--out" && i + 1 < argc) out_path = argv[++i];
  }
  std::vector<Lib> libs;
  for (auto& p : split(argv[1], ',')) {
    void* h = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", p.c_str(), dlerror()); return 2; }
    Lib l{p, (attn_fn)dlsym(h, "ea_attention_f16"), (win_fn)dlsym(h, "ea_sam_window_attn_f16"),
          (gnws_fn)dlsym(h, "ea_groupnorm_workspace_bytes"), (gn_fn)dlsym(h, "ea_groupnorm_f16"), (ln_fn)dlsym(h, "ea_layernorm_f16")};
    if (!l.attn || !l.win || !l.gnws || !l.gn || !l.ln) { fprintf(stderr, "%s: missing entry points\n", p.c_str()); return 2; }
    libs.push_back(l);
  }
  FILE* out = out_path.empty() ? nullptr : fopen(out_path.c_str(), "a");
  hipStream_t stream;
  HIP_CHECK(hipStreamCreate(&stream));
  hipEvent_t e0, e1;
  HIP_CHECK(hipEventCreate(&e0));
  HIP_CHECK(hipEventCreate(&e1));

  for (auto& c : all_cases()) {
    if (cases_sel != "all") {
      const bool kind_match = (cases_sel == "attn" && (c.kind == ATTN || c.kind == WIN)) || (cases_sel == "gn" && c.kind == GN) ||
                              (cases_sel == "ln" && c.kind == LN);
      if (!kind_match && c.name.find(cases_sel) == std::string::npos) continue;
    }
    std::vector<void*> frees;
    std::vector<_Float16> hq, hkv, hx;
    std::vector<float> hx32, hgamma, hbeta;
    void *q = nullptr, *kv = nullptr, *o = nullptr, *x = nullptr, *x2 = nullptr, *ws = nullptr, *gamma = nullptr, *beta = nullptr,
         *relh = nullptr, *relw = nullptr;
    size_t ws_bytes = 0, out_elems = 0;
    const float scale = c.D ? 1.0f / sqrtf((float)c.D) : 1.0f;
    if (c.kind == ATTN) {
      const int HD = c.H * c.D;
      hq = host_f16((size_t)c.B * c.Nq * (c.cross ? HD : 3 * HD), 1.0f);
      q = to_dev(hq); frees.push_back(q);
      if (c.cross) { hkv = host_f16((size_t)c.B * c.Nk * 2 * HD, 1.0f); kv = to_dev(hkv); frees.push_back(kv); }
      out_elems = (size_t)c.B * c.Nq * HD;
    } else if (c.kind == WIN) {
      const int HD = c.H * c.D, T = c.Nq * c.Nq;
      hq = host_f16((size_t)c.B * T * 3 * HD, 1.0f);
      q = to_dev(hq); frees.push_back(q);
      relh = to_dev(host_f16((size_t)(2 * c.Nq - 1) * c.D, 0.2f)); frees.push_back(relh);
      relw = to_dev(host_f16((size_t)(2 * c.Nq - 1) * c.D, 0.2f)); frees.push_back(relw);
      out_elems = (size_t)c.B * T * HD;
    } else if (c.kind == GN) {
      hx = host_f16((size_t)c.B * c.HW * c.C, 1.0f);
      x = to_dev(hx); frees.push_back(x);
      if (c.c2) { x2 = to_dev(host_f16((size_t)c.B * c.HW * c.c2, 1.0f)); frees.push_back(x2); }
      hgamma.resize(c.C + c.c2); hbeta.resize(c.C + c.c2);
      for (auto& g : hgamma) g = 1.0f + 0.1f * urand();
      for (auto& g : hbeta) g = 0.1f * urand();
      gamma = to_dev(hgamma); beta = to_dev(hbeta); frees.push_back(gamma); frees.push_back(beta);
      ws_bytes = libs[0].gnws(c.B, c.HW, c.C + c.c2, 32);
      if (ws_bytes) { HIP_CHECK(hipMalloc(&ws, ws_bytes)); frees.push_back(ws); }
      out_elems = (size_t)c.B * c.HW * (c.C + c.c2);
    } else {
      if (c.in_f32) { hx32.resize((size_t)c.M * c.C); for (auto& f : hx32) f = urand(); x = to_dev(hx32); }
      else { hx = host_f16((size_t)c.M * c.C, 1.0f); x = to_dev(hx); }
      frees.push_back(x);
      hgamma.resize(c.C); hbeta.resize(c.C);
      for (auto& g : hgamma) g = 1.0f + 0.1f * urand();
      for (auto& g : hbeta) g = 0.1f * urand();
      gamma = to_dev(hgamma); beta = to_dev(hbeta); frees.push_back(gamma); frees.push_back(beta);
      out_elems = (size_t)c.M * c.C;
    }
    HIP_CHECK(hipMalloc(&o, out_elems * 2));
    frees.push_back(o);
    auto launch = [&](Lib& l) -> int {
      if (c.kind == ATTN) {
        const long long HD = (long long)c.H * c.D;
        if (c.cross)
          return l.attn(q, kv, (const char*)kv + HD * 2, o, c.B, c.H, c.Nq, c.Nk, c.D, (long long)c.Nq * HD, HD,
                        (long long)c.Nk * 2 * HD, 2 * HD, (long long)c.Nk * 2 * HD, 2 * HD, (long long)c.Nq * HD, HD, scale,
                        nullptr, nullptr, 0, stream);
        return l.attn(q, (const char*)q + HD * 2, (const char*)q + HD * 4, o, c.B, c.H, c.Nq, c.Nk, c.D, (long long)c.Nq * 3 * HD,
                      3 * HD, (long long)c.Nk * 3 * HD, 3 * HD, (long long)c.Nk * 3 * HD, 3 * HD, (long long)c.Nq * HD, HD, scale,
                      nullptr, nullptr, 0, stream);
      }
      if (c.kind == WIN) {
        const long long HD = (long long)c.H * c.D, T = (long long)c.Nq * c.Nq;
        return l.win(q, (const char*)q + HD * 2, (const char*)q + HD * 4, o, c.B, c.H, c.Nq, c.D, T * 3 * HD, 3 * HD, T * 3 * HD,
                     3 * HD, T * 3 * HD, 3 * HD, T * HD, HD, scale, relh, relw, stream);
      }
      if (c.kind == GN)
        return l.gn(x, c.C, x2, c.c2, nullptr, (const float*)gamma, (const float*)beta, o, c.B, c.HW, 32, 1e-5f, c.silu, ws, ws_bytes, stream);
      return l.ln(x, c.in_f32, (const float*)gamma, (const float*)beta, o, c.M, c.C, 1e-5f, stream);
    };
    struct Cfg { int lib; hipGraphExec_t exec; std::vector<float> us; double maxdiff; int st; };
    std::vector<Cfg> cfgs;
    for (size_t li = 0; li < libs.size(); ++li) cfgs.push_back(Cfg{(int)li, nullptr, {}, -1.0, 0});
    for (auto& cf : cfgs) {
      cf.st = launch(libs[cf.lib]);
      HIP_CHECK(hipStreamSynchronize(stream));
      if (cf.st != 0) continue;
      if (check) {   // fp64 host evaluation of one slice
        std::vector<_Float16> ho(out_elems);
        HIP_CHECK(hipMemcpy(ho.data(), o, out_elems * 2, hipMemcpyDeviceToHost));
        double md = 0.0;
        if (c.kind == ATTN) {
          const int HD = c.H * c.D, b = c.B - 1, h = c.H - 1;
          const int qs = c.cross ? HD : 3 * HD;
          for (int i = 0; i < c.Nq; i += (c.Nq > 64 ? c.Nq / 16 : 1)) {      // 16 query rows of the last (batch, head)
            std::vector<double> sc(c.Nk);
            double mx = -1e300;
            for (int j = 0; j < c.Nk; ++j) {
              double s = 0.0;
              for (int d = 0; d < c.D; ++d) {
                const double qv = (double)(float)hq[((size_t)b * c.Nq + i) * qs + h * c.D + d];
                const double kvv = c.cross ? (double)(float)hkv[((size_t)b * c.Nk + j) * 2 * HD + h * c.D + d]
                                           : (double)(float)hq[((size_t)b * c.Nk + j) * 3 * HD + HD + h * c.D + d];
                s += qv * kvv;
              }
              sc[j] = s * scale;
              mx = std::max(mx, sc[j]);
            }
            double den = 0.0;
            for (int j = 0; j < c.Nk; ++j) { sc[j] = exp(sc[j] - mx); den += sc[j]; }
            for (int d = 0; d < c.D; ++d) {
              double acc = 0.0;
              for (int j = 0; j < c.Nk; ++j) {
                const double vv = c.cross ? (double)(float)hkv[((size_t)b * c.Nk + j) * 2 * HD + HD + h * c.D + d]
                                          : (double)(float)hq[((size_t)b * c.Nk + j) * 3 * HD + 2 * HD + h * c.D + d];
                acc += sc[j] * vv;
              }
              const double got = (double)(float)ho[((size_t)b * c.Nq + i) * HD + h * c.D + d];
              md = std::max(md, fabs(got - acc / den));
            }
          }
          cf.maxdiff = md;
        } else if (c.kind == LN && !c.in_f32) {
          for (int r = 0; r < c.M; r += std::max(1, c.M / 8)) {
            double mean = 0.0, var = 0.0;
            for (int j = 0; j < c.C; ++j) mean += (double)(float)hx[(size_t)r * c.C + j];
            mean /= c.C;
            for (int j = 0; j < c.C; ++j) { const double d = (double)(float)hx[(size_t)r * c.C + j] - mean; var += d * d; }
            const double rstd = 1.0 / sqrt(var / c.C + 1e-5);
            for (int j = 0; j < c.C; ++j) {
              const double ref = ((double)(float)hx[(size_t)r * c.C + j] - mean) * rstd * hgamma[j] + hbeta[j];
              md = std::max(md, fabs((double)(float)ho[(size_t)r * c.C + j] - ref));
            }
          }
          cf.maxdiff = md;
        }
      }
      hipGraph_t graph;
      HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < iters; ++i) launch(libs[cf.lib]);
      HIP_CHECK(hipStreamEndCapture(stream, &graph));
      HIP_CHECK(hipGraphInstantiate(&cf.exec, graph, nullptr, nullptr, 0));
      HIP_CHECK(hipGraphDestroy(graph));
    }
    for (int r = 0; r < rounds; ++r)
      for (auto& cf : cfgs) {
        if (!cf.exec) continue;
        HIP_CHECK(hipEventRecord(e0, stream));
        HIP_CHECK(hipGraphLaunch(cf.exec, stream));
        HIP_CHECK(hipEventRecord(e1, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        float ms;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        cf.us.push_back(ms * 1000.0f / iters);
      }
    for (auto& cf : cfgs) {
      char line[768];
      if (!cf.exec) {
        snprintf(line, sizeof line, "{\"case\": \"%s\", \"lib\": \"%s\", \"error\": %d}", c.name.c_str(), libs[cf.lib].path.c_str(), cf.st);
      } else {
        std::sort(cf.us.begin(), cf.us.end());
        const float best = cf.us.front(), med = cf.us[cf.us.size() / 2];
        int n = snprintf(line, sizeof line, "{\"case\": \"%s\", \"lib\": \"%s\", \"us\": %.2f, \"us_median\": %.2f", c.name.c_str(),
                         libs[cf.lib].path.c_str(), best, med);
        if (c.flops > 0) n += snprintf(line + n, sizeof line - n, ", \"tflops\": %.1f, \"mfma_frac\": %.4f", c.flops / best * 1e-6, c.flops / best * 1e-6 / 2500.0);
        if (c.bytes > 0) n += snprintf(line + n, sizeof line - n, ", \"gbs\": %.1f, \"hbm_frac\": %.4f", c.bytes / best * 1e-3, c.bytes / best * 1e-3 / 8000.0);
        if (cf.maxdiff >= 0.0) n += snprintf(line + n, sizeof line - n, ", \"max_abs_diff_vs_fp64_slice\": %.5g", cf.maxdiff);
        snprintf(line + n, sizeof line - n, "}");
        HIP_CHECK(hipGraphExecDestroy(cf.exec));
      }
      puts(line);
      fflush(stdout);
      if (out) { fputs(line, out); fputc('\n', out); fflush(out); }
    }
    for (void* p : frees) HIP_CHECK(hipFree(p));
  }
  if (out) fclose(out);
  return 0;
}
