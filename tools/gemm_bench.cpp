// gemm_bench.cpp -- torch-free A/B harness for the contraction kernel (starts in well under a second on the GPU
// box, where `import torch` costs 1-2 minutes of a 90-minute budget).
//
//   hipcc -O2 tools/gemm_bench.cpp -o tools/gemm_bench -ldl
//   tools/gemm_bench <lib.so>[,<lib2.so>...] [--cases all|gemm|conv|<substring>] [--variants auto,1,9,...]
//                    [--debug 0,1,...] [--iters 20] [--rounds 3] [--check] [--out file.jsonl]
//
// For every case x library x variant (x EA_GEMM2_DEBUG knob) it prints one JSON line: microseconds per launch (HIP
// graph of `iters` launches, best and median of `rounds` interleaved replays -- the product replays graphs too),
// TFLOP/s, and with --check the max |difference| against the generic kernel (EA_GEMM_FORCE=generic) of the FIRST
// library on the same inputs, repeated every round (race screen).  Variants/knobs are the library's own environment
// knobs (ea_gemm.hip), re-read on every call.  Several libraries = builds of the same sources with different -D
// experiment flags (tools/build_exp.sh).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
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

typedef int (*gemm_fn)(const void*, int, const void*, int, int, int, int, int, long long, long long, long long, long long,
                       const ea_epilogue*, void*, size_t, void*);
typedef int (*conv_fn)(const ea_conv_src*, const void*, int, const ea_epilogue*, void*, size_t, void*);
typedef int (*tune_fn)(const ea_tuning*);

struct Lib {
  std::string path;
  gemm_fn gemm;
  conv_fn conv;
  tune_fn tune;   // ea_set_tuning (libraries built before it existed read the EA_GEMM* environment instead: both are set)
};
// one configuration = (variant, debug knob, split factor) + the process-wide A/B switches of this run
static int g_no_tr = 0, g_bn = 0;
static void apply_tuning(std::vector<Lib>& libs, const std::string& variant, const std::string& debug, const std::string& splits, bool generic = false) {
  if (generic) setenv("EA_GEMM_FORCE", "generic", 1); else unsetenv("EA_GEMM_FORCE");
  if (variant == "auto" || variant.empty()) unsetenv("EA_GEMM2_VARIANT"); else setenv("EA_GEMM2_VARIANT", variant.c_str(), 1);
  if (debug == "0" || debug.empty()) unsetenv("EA_GEMM2_DEBUG"); else setenv("EA_GEMM2_DEBUG", debug.c_str(), 1);
  if (splits == "0" || splits.empty()) unsetenv("EA_GEMM2_SPLITS"); else setenv("EA_GEMM2_SPLITS", splits.c_str(), 1);
  ea_tuning t{};
  t.force_generic = generic ? 1 : 0;
  t.variant = (variant == "auto" || variant.empty()) ? 0 : atoi(variant.c_str());
  t.debug = atoi(debug.c_str());
  t.splits = atoi(splits.c_str());
  t.bn = g_bn;
  t.no_register_direct = g_no_tr;
  for (auto& l : libs)
    if (l.tune) l.tune(&t);
}

struct Case {
  std::string name;
  int conv;            // 0 gemm, 1 conv
  int M, N, K;         // gemm
  int B, H, c1, c2, Cout, ksize, stride, ups;   // conv (square H x H input)
  int act, residual, rowvec;
  double flops;
};

static std::vector<Case> all_cases() {
  std::vector<Case> v;
  auto g = [&](int M, int N, int K, int act, int res) {
    Case c{};
    char b[128];
    snprintf(b, sizeof b, "gemm M%d N%d K%d act%d%s", M, N, K, act, res ? " res" : "");
    c.name = b; c.conv = 0; c.M = M; c.N = N; c.K = K; c.act = act; c.residual = res;
    c.flops = 2.0 * M * N * K;
    v.push_back(c);
  };
  auto cv = [&](int B, int H, int c1, int c2, int Cout, int ks, int stride, int ups, int rowvec = 0, int res = 0) {
    Case c{};
    char b[128];
    snprintf(b, sizeof b, "conv%d B%d H%d c%d+%d->%d s%d u%d%s%s", ks, B, H, c1, c2, Cout, stride, ups, rowvec ? " emb" : "", res ? " res" : "");
    c.name = b; c.conv = 1; c.rowvec = rowvec; c.residual = res; c.B = B; c.H = H; c.c1 = c1; c.c2 = c2; c.Cout = Cout; c.ksize = ks; c.stride = stride; c.ups = ups;
    const int Ho = ups ? 2 * H : (stride == 2 ? H / 2 : H);
    c.flops = 2.0 * B * Ho * Ho * Cout * ks * ks * (c1 + c2);
    v.push_back(c);
  };
  // the launches of one ControlNet + UNet evaluation at network batch 8 (profiles/r01_eval_breakdown_v3.json), by time
  g(32768, 2560, 320, 3, 0);  g(8192, 5120, 640, 3, 0);   g(2048, 10240, 1280, 3, 0);
  g(32768, 320, 320, 0, 1);   g(32768, 320, 320, 0, 0);   g(32768, 960, 320, 0, 0);   g(32768, 320, 1280, 0, 1);
  g(8192, 640, 640, 0, 1);    g(8192, 640, 640, 0, 0);    g(8192, 1920, 640, 0, 0);   g(8192, 640, 2560, 0, 1);
  g(2048, 1280, 1280, 0, 1);  g(2048, 1280, 1280, 0, 0);  g(2048, 3840, 1280, 0, 0);  g(2048, 1280, 5120, 0, 1);
  g(512, 1280, 1280, 0, 1);   g(512, 3840, 1280, 0, 0);   g(512, 10240, 1280, 3, 0);  g(512, 1280, 5120, 0, 1);
  cv(8, 64, 320, 0, 320, 3, 1, 0);    cv(8, 32, 640, 0, 640, 3, 1, 0);   cv(8, 16, 1280, 0, 1280, 3, 1, 0);
  cv(8, 8, 1280, 0, 1280, 3, 1, 0);   cv(8, 64, 640, 0, 320, 3, 1, 0);   cv(8, 16, 2560, 0, 1280, 3, 1, 0);
  cv(8, 32, 1920, 0, 640, 3, 1, 0);   cv(8, 64, 960, 0, 320, 3, 1, 0);   cv(8, 8, 2560, 0, 1280, 3, 1, 0);
  cv(8, 32, 640, 0, 640, 3, 1, 1);    cv(8, 16, 1280, 0, 1280, 3, 1, 1); cv(8, 64, 320, 0, 320, 3, 2, 0);
  cv(8, 32, 320, 0, 640, 3, 1, 0);    cv(8, 16, 640, 0, 1280, 3, 1, 0);  cv(8, 64, 320, 320, 320, 1, 1, 0);
  cv(8, 64, 320, 0, 320, 1, 1, 0);    cv(8, 8, 1280, 0, 1280, 1, 1, 0);
  // ResBlock forms of the same convolutions: in_layers conv + the per-sample embedding row vector (openaimodel.py:259-262),
  // out_layers conv + the skip tensor as residual (openaimodel.py:263)
  cv(8, 64, 320, 0, 320, 3, 1, 0, 1, 0);  cv(8, 64, 320, 0, 320, 3, 1, 0, 0, 1);
  cv(8, 32, 640, 0, 640, 3, 1, 0, 1, 0);  cv(8, 32, 640, 0, 640, 3, 1, 0, 0, 1);
  cv(8, 16, 1280, 0, 1280, 3, 1, 0, 1, 0);
  // the shared CFG prefix runs the first level-0 ResBlock / transformer projections on ONE copy of the batch (B = 4)
  cv(4, 64, 320, 0, 320, 3, 1, 0, 1, 0);  cv(4, 64, 320, 0, 320, 3, 1, 0, 0, 1);
  g(16384, 320, 320, 0, 1);  g(16384, 960, 320, 0, 0);
  // SAM ViT-H linears (4 images: 16384 tokens / 19600 window tokens) and VAE decoder convs (batch 4)
  g(16384, 3840, 1280, 0, 0); g(16384, 1280, 1280, 0, 1); g(16384, 5120, 1280, 2, 0); g(16384, 1280, 5120, 0, 1);
  cv(4, 256, 256, 0, 256, 3, 1, 0);   cv(4, 512, 128, 0, 128, 3, 1, 0);
  g(19600, 3840, 1280, 0, 0); g(19600, 1280, 1280, 0, 0);   // windowed blocks: 25 windows of 196 tokens per image
  cv(4, 128, 512, 0, 512, 3, 1, 0);   cv(4, 64, 512, 0, 512, 3, 1, 0);   // VAE decoder, 512-channel levels
  // round 5: the remaining launch classes of a step above 0.3 % of its contraction time (PMC coverage: tools/pmc_cases.txt)
  cv(8, 32, 1280, 0, 640, 3, 1, 0);   cv(8, 32, 960, 0, 640, 3, 1, 0);    cv(8, 16, 1920, 0, 1280, 3, 1, 0);
  cv(8, 32, 640, 0, 640, 3, 2, 0);    cv(8, 16, 1280, 0, 1280, 3, 2, 0);  cv(8, 16, 1280, 1280, 1280, 1, 1, 0);
  cv(8, 8, 1280, 0, 1280, 3, 1, 1);   cv(8, 8, 1280, 1280, 1280, 1, 1, 0);
  cv(4, 256, 256, 0, 256, 3, 1, 1);   cv(4, 128, 512, 0, 512, 3, 1, 1);
  cv(8, 32, 640, 0, 640, 1, 1, 0);    cv(8, 16, 1280, 0, 1280, 1, 1, 0);   // zero-convs of the 32 x 32 / 16 x 16 levels
  // calibration cubes (the guide's ladder is quoted at 4096^3 / 8192^3): compare with tools/gemm8_probe in the same call
  g(4096, 4096, 4096, 0, 0);  g(8192, 8192, 8192, 0, 0);
  // row-stride probes (round 3): the same launches with K moved off the power-of-two row strides (2560 / 10240 bytes), to see
  // whether the L2 / fabric channel interleave penalises those strides ("--cases stride")
  g(2048, 1280, 1344, 0, 0);  g(2048, 1280, 5184, 0, 1);  g(8192, 640, 2624, 0, 1);
  for (size_t i = v.size() - 3; i < v.size(); ++i) v[i].name += " stride";
  // L2-resident operands (2 x 2 MB) with many tiles: what does the DMA stream deliver when nothing misses? ("--cases l2fit")
  g(4096, 4096, 256, 0, 0);  g(2048, 2048, 512, 0, 0);
  for (size_t i = v.size() - 2; i < v.size(); ++i) v[i].name += " l2fit";
  return v;
}

static uint32_t rng_state = 12345u;
static inline float urand() {   // uniform [-1, 1): full-range random operands (guide rule 25: not zeros / constants)
  rng_state = rng_state * 1664525u + 1013904223u;
  return ((rng_state >> 8) * (1.0f / 8388608.0f)) - 1.0f;
}

static void* dev_f16(size_t n, float scale) {
  std::vector<_Float16> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (_Float16)(urand() * scale);
  void* d;
  HIP_CHECK(hipMalloc(&d, n * 2));
  HIP_CHECK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
  return d;
}
static void* dev_f32(size_t n, float scale) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = urand() * scale;
  void* d;
  HIP_CHECK(hipMalloc(&d, n * 4));
  HIP_CHECK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
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
    fprintf(stderr, "usage: %s lib.so[,lib2.so] [--cases s] [--variants a,b] [--debug a,b] [--iters n] [--rounds n] [--check] [--out f]\n", argv[0]);
    return 1;
  }
  std::string cases_sel = "all", out_path;
  std::vector<std::string> variants = {"auto"}, debugs = {"0"}, splits_opt = {"0"};
  int iters = 20, rounds = 3, check = 0, geglu = 80, rotate_mb = 0;
  for (int i = 2; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "--cases" && i + 1 < argc) cases_sel = argv[++i];
    else if (a == "--variants" && i + 1 < argc) variants = split(argv[++i], ',');
    else if (a == "--debug" && i + 1 < argc) debugs = split(argv[++i], ',');
    else if (a == "--splits" && i + 1 < argc) splits_opt = split(argv[++i], ',');   // EA_GEMM2_SPLITS (0 = plan's own)
    else if (a == "--iters" && i + 1 < argc) iters = atoi(argv[++i]);
    else if (a == "--rounds" && i + 1 < argc) rounds = atoi(argv[++i]);
    else if (a == "--check") check = 1;
    else if (a == "--geglu" && i + 1 < argc) geglu = atoi(argv[++i]);   // EA_ACT_GEGLU weight-row packing: 80 | 32
    else if (a == "--bn" && i + 1 < argc) { g_bn = atoi(argv[++i]); setenv("EA_GEMM2_BN", argv[i], 1); }
    else if (a == "--slab-epilogue") { g_no_tr = 1; setenv("EA_GEMM2_TR", "0", 1); }   // A/B: no register-direct epilogue
    else if (a == "--out" && i + 1 < argc) out_path = argv[++i];
    else if (a == "--rotate-mb" && i + 1 < argc) rotate_mb = atoi(argv[++i]);   // weight copies worth this many MB, rotated launch by launch
  }
  std::vector<Lib> libs;
  for (auto& p : split(argv[1], ',')) {
    void* h = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", p.c_str(), dlerror()); return 2; }
    Lib l{p, (gemm_fn)dlsym(h, "ea_gemm_f16"), (conv_fn)dlsym(h, "ea_conv2d_f16"), (tune_fn)dlsym(h, "ea_set_tuning")};
    if (!l.gemm || !l.conv) { fprintf(stderr, "%s: missing ea_gemm_f16 / ea_conv2d_f16\n", p.c_str()); return 2; }
    libs.push_back(l);
  }
  FILE* out = out_path.empty() ? nullptr : fopen(out_path.c_str(), "a");
  hipStream_t stream;
  HIP_CHECK(hipStreamCreate(&stream));
  const size_t ws_bytes = (size_t)512 << 20;
  void* ws;
  HIP_CHECK(hipMalloc(&ws, ws_bytes));
  hipEvent_t e0, e1;
  HIP_CHECK(hipEventCreate(&e0));
  HIP_CHECK(hipEventCreate(&e1));

  for (auto& c : all_cases()) {
    if (cases_sel == "all" && (c.flops > 3e11 || c.name.find("stride") != std::string::npos || c.name.find("l2fit") != std::string::npos)) continue;   // cubes / probes: by name only
    if (cases_sel != "all") {
      if (cases_sel[0] == '=' ? c.name != cases_sel.substr(1)          // "=<name>": exactly that case
          : cases_sel == "gemm" ? c.conv != 0 : cases_sel == "conv" ? c.conv != 1 : c.name.find(cases_sel) == std::string::npos) continue;
    }
    // ---- operands
    int M, N, K;
    ea_conv_src src{};
    void *A = nullptr, *A2 = nullptr;
    if (c.conv) {
      const int Ho = c.ups ? 2 * c.H : (c.stride == 2 ? c.H / 2 : c.H);
      M = c.B * Ho * Ho; N = c.Cout; K = c.ksize * c.ksize * (c.c1 + c.c2);
      A = dev_f16((size_t)c.B * c.H * c.H * c.c1, 1.0f);
      if (c.c2) A2 = dev_f16((size_t)c.B * c.H * c.H * c.c2, 1.0f);
      src.x1 = A; src.c1 = c.c1; src.x2 = A2; src.c2 = c.c2; src.x2_add = nullptr;
      src.B = c.B; src.Hin = c.H; src.Win = c.H; src.ksize = c.ksize; src.stride = c.stride; src.pad = c.ksize == 3 ? 1 : 0;
      src.ups = c.ups; src.Hout = Ho; src.Wout = Ho;
    } else {
      M = c.M; N = c.N; K = c.K;
      A = dev_f16((size_t)M * K, 1.0f);
    }
    void* W = dev_f16((size_t)N * K, 1.0f / sqrtf((float)K));
    // --rotate-mb X: the graph's launches walk through copies of W worth X MB in all (one launch = one copy), so a launch never
    // finds its weights where the previous replay left them: X between the L2s' 32 MB and the 256-MB Infinity Cache = weights
    // in the Infinity Cache but in no L2; X well above 256 = weights from HBM, as inside a denoising step (2.5 GB of weights per
    // evaluation); 0 = one copy, hot in every cache (what every table of rounds 2-6 was measured with).  A stays one buffer: in
    // the step it was written by the previous launch.
    std::vector<void*> Wrot{W};
    if (rotate_mb > 0) {
      const size_t wb = (size_t)N * K * 2;
      const size_t ncopy = std::min<size_t>(400, std::max<size_t>(2, ((size_t)rotate_mb << 20) / wb + 1));   // (tiny weights: capped -- they are no traffic)
      for (size_t i = 1; i < ncopy; ++i) {
        void* d;
        HIP_CHECK(hipMalloc(&d, wb));
        HIP_CHECK(hipMemcpy(d, W, wb, hipMemcpyDeviceToDevice));
        Wrot.push_back(d);
      }
    }
    const int iters_c = rotate_mb > 0 ? (int)std::max<size_t>(iters, Wrot.size()) : iters;
    void* bias = dev_f32(N, 0.1f);
    const int Nout = c.act == EA_ACT_GEGLU ? N / 2 : N;
    void* res = c.residual ? dev_f16((size_t)M * Nout, 1.0f) : nullptr;
    void* rowvec = (c.conv && c.rowvec) ? dev_f32((size_t)c.B * N, 0.5f) : nullptr;
    void *o_test, *o_ref;
    HIP_CHECK(hipMalloc(&o_test, (size_t)M * Nout * 2));
    HIP_CHECK(hipMalloc(&o_ref, (size_t)M * Nout * 2));
    auto make_epi = [&](void* o) {
      ea_epilogue e{};
      e.bias = (const float*)bias; e.act = c.act; e.scale = 1.0f; e.rows_per_group = 1;
      if (rowvec) { e.rowvec = (const float*)rowvec; e.rowvec_ld = N; e.rows_per_group = M / c.B; }
      e.residual = res; e.ldr = Nout; e.out = o; e.ldc = Nout; e.geglu_block = c.act == EA_ACT_GEGLU ? geglu : 0;
      return e;
    };
    auto launch = [&](Lib& l, void* o, int wi = 0) {
      ea_epilogue e = make_epi(o);
      void* Wi = Wrot[(size_t)wi % Wrot.size()];
      return c.conv ? l.conv(&src, Wi, N, &e, ws, ws_bytes, stream)
                    : l.gemm(A, K, Wi, K, M, N, K, 1, 0, 0, 0, 0, &e, ws, ws_bytes, stream);
    };
    std::vector<_Float16> h_ref, h_test;
    if (check) {
      apply_tuning(libs, "auto", "0", "0", true);
      int st = launch(libs[0], o_ref);
      HIP_CHECK(hipStreamSynchronize(stream));
      if (st != 0) {   // e.g. GEGLU with 80-row packing exists only in the fast kernel: reference = its GENERAL epilogue
        apply_tuning(libs, "auto", "9", "0");
        st = launch(libs[0], o_ref);
        HIP_CHECK(hipStreamSynchronize(stream));
      }
      apply_tuning(libs, "auto", "0", "0");
      if (st != 0) { fprintf(stderr, "%s: reference launch failed (%d)\n", c.name.c_str(), st); check = 0; }
      h_ref.resize((size_t)M * Nout);
      h_test.resize((size_t)M * Nout);
      HIP_CHECK(hipMemcpy(h_ref.data(), o_ref, h_ref.size() * 2, hipMemcpyDeviceToHost));
    }
    // ---- every (lib, variant, debug) configuration gets a graph; rounds interleave the configurations
    struct Cfg { int lib; std::string variant, debug, splits; hipGraphExec_t exec; std::vector<float> us; double maxdiff; long long bad; int st; };
    std::vector<Cfg> cfgs;
    for (size_t li = 0; li < libs.size(); ++li)
      for (auto& v : variants)
        for (auto& d : debugs)
          for (auto& sp : splits_opt) {
            if (sp != "0" && c.act == EA_ACT_GEGLU && sp != "1") continue;   // GEGLU launches never split
            cfgs.push_back(Cfg{(int)li, v, d, sp, nullptr, {}, 0.0, 0, 0});
          }
    for (auto& cf : cfgs) {
      apply_tuning(libs, cf.variant, cf.debug, cf.splits);
      cf.st = launch(libs[cf.lib], o_test);   // warm-up (module load, LDS attribute)
      HIP_CHECK(hipStreamSynchronize(stream));
      if (cf.st != 0) continue;
      hipGraph_t graph;
      HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < iters_c; ++i) launch(libs[cf.lib], o_test, i);
      HIP_CHECK(hipStreamEndCapture(stream, &graph));
      HIP_CHECK(hipGraphInstantiate(&cf.exec, graph, nullptr, nullptr, 0));
      HIP_CHECK(hipGraphDestroy(graph));
    }
    apply_tuning(libs, "auto", "0", "0");
    for (int r = 0; r < rounds; ++r) {
      for (auto& cf : cfgs) {
        if (!cf.exec) continue;
        if (check && cf.debug == "0") HIP_CHECK(hipMemsetAsync(o_test, 0xff, (size_t)M * Nout * 2, stream));   // NaN canary
        HIP_CHECK(hipEventRecord(e0, stream));
        HIP_CHECK(hipGraphLaunch(cf.exec, stream));
        HIP_CHECK(hipEventRecord(e1, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        float ms;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        cf.us.push_back(ms * 1000.0f / iters_c);
        if (check && cf.debug == "0") {
          HIP_CHECK(hipMemcpy(h_test.data(), o_test, h_test.size() * 2, hipMemcpyDeviceToHost));
          for (size_t i = 0; i < h_ref.size(); ++i) {
            const double d = fabs((double)(float)h_test[i] - (double)(float)h_ref[i]);
            if (d != d) ++cf.bad;              // NaN: an output the kernel never wrote (canary) or a broken one
            else if (d > cf.maxdiff) cf.maxdiff = d;
          }
        }
      }
    }
    // --debug 3 (tools builds): the kernel's phase timestamps (ea_gemm2.h EA_STAMP: 100 MHz wall clock per workgroup at
    // 0 entry, 1 setup done, 2 K loop done, 3 past the pre-epilogue barrier, 4 end) from ONE launch, after the timed rounds
    for (auto& cf : cfgs) {
      if (!cf.exec || cf.debug != "3") continue;
      apply_tuning(libs, cf.variant, cf.debug, cf.splits);
      HIP_CHECK(hipMemsetAsync(ws, 0, 8 << 20, stream));
      launch(libs[cf.lib], o_test);
      HIP_CHECK(hipStreamSynchronize(stream));
      std::vector<unsigned long long> st((8 << 20) / 8);
      HIP_CHECK(hipMemcpy(st.data(), ws, 8 << 20, hipMemcpyDeviceToHost));
      apply_tuning(libs, "auto", "0", "0");
      unsigned long long t0 = ~0ull, t_end = 0;
      int nb = 0;
      double d[4] = {0, 0, 0, 0}, dmax[4] = {0, 0, 0, 0};
      for (size_t b = 0; b + 8 <= st.size(); b += 8) {
        if (!st[b] || !st[b + 4]) continue;
        ++nb;
        if (st[b] < t0) t0 = st[b];
        if (st[b + 4] > t_end) t_end = st[b + 4];
      }
      double spread = 0;
      for (size_t b = 0; b + 8 <= st.size(); b += 8) {
        if (!st[b] || !st[b + 4]) continue;
        if ((st[b] - t0) / 100.0 > spread) spread = (st[b] - t0) / 100.0;
        const unsigned long long t[5] = {st[b], st[b + 1], st[b + 2], st[b + 3] ? st[b + 3] : st[b + 2], st[b + 4]};
        for (int i = 0; i < 4; ++i) {
          const double x = (double)(t[i + 1] - t[i]) / 100.0;
          d[i] += x;
          if (x > dmax[i]) dmax[i] = x;
        }
      }
      char line[768];
      snprintf(line, sizeof line,
               "{\"case\": \"%s\", \"lib\": \"%s\", \"variant\": \"%s\", \"stamps\": 1, \"workgroups\": %d, \"start_spread_us\": %.2f, \"kernel_span_us\": %.2f, "
               "\"mean_us\": {\"setup\": %.2f, \"k_loop\": %.2f, \"barrier\": %.2f, \"epilogue\": %.2f}, \"max_us\": {\"setup\": %.2f, \"k_loop\": %.2f, \"barrier\": %.2f, \"epilogue\": %.2f}}",
               c.name.c_str(), libs[cf.lib].path.c_str(), cf.variant.c_str(), nb, spread, nb ? (double)(t_end - t0) / 100.0 : 0.0,
               nb ? d[0] / nb : 0, nb ? d[1] / nb : 0, nb ? d[2] / nb : 0, nb ? d[3] / nb : 0, dmax[0], dmax[1], dmax[2], dmax[3]);
      puts(line);
      fflush(stdout);
      if (out) { fputs(line, out); fputc('\n', out); fflush(out); }
    }
    for (auto& cf : cfgs) {
      char line[768];
      if (!cf.exec) {
        snprintf(line, sizeof line, "{\"case\": \"%s\", \"lib\": \"%s\", \"variant\": \"%s\", \"debug\": \"%s\", \"splits\": \"%s\", \"error\": %d}",
                 c.name.c_str(), libs[cf.lib].path.c_str(), cf.variant.c_str(), cf.debug.c_str(), cf.splits.c_str(), cf.st);
      } else {
        std::sort(cf.us.begin(), cf.us.end());
        const float best = cf.us.front(), med = cf.us[cf.us.size() / 2];
        int n = snprintf(line, sizeof line,
                         "{\"case\": \"%s\", \"lib\": \"%s\", \"variant\": \"%s\", \"debug\": \"%s\", \"splits\": \"%s\", \"us\": %.2f, \"us_median\": %.2f, "
                         "\"tflops\": %.1f, \"mfma_frac\": %.4f, \"rotate_mb\": %d, \"weight_copies\": %d",
                         c.name.c_str(), libs[cf.lib].path.c_str(), cf.variant.c_str(), cf.debug.c_str(), cf.splits.c_str(), best, med,
                         c.flops / best * 1e-6, c.flops / best * 1e-6 / 2500.0, rotate_mb, (int)Wrot.size());
        if (check && cf.debug == "0") n += snprintf(line + n, sizeof line - n, ", \"max_abs_diff_vs_generic\": %.5g, \"nan_outputs\": %lld", cf.maxdiff, cf.bad);
        snprintf(line + n, sizeof line - n, "}");
        HIP_CHECK(hipGraphExecDestroy(cf.exec));
      }
      puts(line);
      fflush(stdout);
      if (out) { fputs(line, out); fputc('\n', out); fflush(out); }
    }
    for (size_t i = 1; i < Wrot.size(); ++i) HIP_CHECK(hipFree(Wrot[i]));
    for (void* p : {A, A2, W, bias, res, rowvec, o_test, o_ref})
      if (p) HIP_CHECK(hipFree(p));
  }
  if (out) fclose(out);
  return 0;
}
