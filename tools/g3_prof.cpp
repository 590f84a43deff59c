// g3_prof.cpp -- phase-cycle breakdown of the persistent contraction kernel (ea_gemm3.h built with -DEA_G3_PROF=1).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -DEA_G3_PROF=1 -shared \
//         editanything_amd/csrc/ea_gemm.hip -o gpurun_exp/libea_g3prof.so
//   hipcc -O2 -std=c++17 tools/g3_prof.cpp -o tools/g3_prof -ldl
//   tools/g3_prof gpurun_exp/libea_g3prof.so <variant 21|22> <splits> gemm M N K | conv B H Cin Cout
// Prints, per wave of workgroups 0 and 37, the s_memtime cycle totals of the loop phases (wait, barrier, early issue,
// reads + MFMAs, late issue, item end, item open, total) and the launch time.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../include/editanything_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef int (*gemm_fn)(const void*, int, const void*, int, int, int, int, int, long long, long long, long long, long long, const ea_epilogue*, void*, size_t, void*);
typedef int (*conv_fn)(const ea_conv_src*, const void*, int, const ea_epilogue*, void*, size_t, void*);
typedef int (*tune_fn)(const ea_tuning*);
static void* dev_rand(size_t n) {
  std::vector<_Float16> h(n);
  unsigned s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (_Float16)(((s >> 8) * (1.0f / 8388608.0f) - 1.0f) * 0.5f); }
  void* d; CK(hipMalloc(&d, n * 2)); CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice)); return d;
}
int main(int argc, char** argv) {
  if (argc < 8) { printf("usage: g3_prof lib variant splits gemm M N K | conv B H Cin Cout\n"); return 1; }
  void* h = dlopen(argv[1], RTLD_NOW);
  if (!h) { printf("dlopen: %s\n", dlerror()); return 1; }
  gemm_fn gemm = (gemm_fn)dlsym(h, "ea_gemm_f16"); conv_fn conv = (conv_fn)dlsym(h, "ea_conv2d_f16"); tune_fn tune = (tune_fn)dlsym(h, "ea_set_tuning");
  ea_tuning t{}; t.variant = atoi(argv[2]); t.splits = atoi(argv[3]); tune(&t);
  const bool is_conv = !strcmp(argv[4], "conv");
  int M, N, K; ea_conv_src src{}; void* A;
  if (is_conv) {
    const int B = atoi(argv[5]), H = atoi(argv[6]), ci = atoi(argv[7]), co = atoi(argv[8]);
    M = B * H * H; N = co; K = 9 * ci; A = dev_rand((size_t)M * ci);
    src.x1 = A; src.c1 = ci; src.B = B; src.Hin = H; src.Win = H; src.ksize = 3; src.stride = 1; src.pad = 1; src.Hout = H; src.Wout = H;
  } else { M = atoi(argv[5]); N = atoi(argv[6]); K = atoi(argv[7]); A = dev_rand((size_t)M * K); }
  void* W = dev_rand((size_t)N * K);
  void* out; CK(hipMalloc(&out, (size_t)M * N * 2));
  const size_t ws_bytes = (size_t)512 << 20; void* ws; CK(hipMalloc(&ws, ws_bytes)); CK(hipMemset(ws, 0, ws_bytes));
  ea_epilogue e{}; e.out = out; e.ldc = N; e.scale = 1.0f; e.rows_per_group = 1;
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() { return is_conv ? conv(&src, W, N, &e, ws, ws_bytes, st) : gemm(A, K, W, K, M, N, K, 1, 0, 0, 0, 0, &e, ws, ws_bytes, st); };
  int rc = launch(); CK(hipStreamSynchronize(st));
  if (rc) { printf("launch rc %d\n", rc); return 1; }
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) { CK(hipEventRecord(e0, st)); for (int i = 0; i < 10; ++i) launch(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms / 10 < best) best = ms / 10; }
  printf("%s M%d N%d K%d variant %s splits %s: %.1f us, %.0f TF/s\n", argv[4], M, N, K, argv[2], argv[3], best * 1e3, 2.0 * M * N * K / best * 1e-9);
  const int splits = atoi(argv[3]) > 1 ? atoi(argv[3]) : 0;
  std::vector<unsigned long long> p(256 * 64);
  // the library places the totals behind the partials it planned; with the plan's own split choice scan for them
  for (int s = 0; s <= 16; ++s) {
    if (splits && s != splits) continue;
    CK(hipMemcpy(p.data(), (char*)ws + (size_t)s * M * N * 4, p.size() * 8, hipMemcpyDeviceToHost));
    if (p[7] > 0 && p[7] < (1ull << 40)) { printf("(totals found behind %d slices)\n", s); break; }
  }
  const char* names[8] = {"wait", "barrier", "issue_e", "rd+mfma", "issue_l", "item_end", "item_open", "TOTAL"};
  for (int wg : {0, 37}) {
    printf("workgroup %d   ", wg); for (int i = 0; i < 8; ++i) printf("%10s", names[i]); printf("\n");
    for (int w = 0; w < 8; ++w) { printf("  wave %d      ", w); for (int i = 0; i < 8; ++i) printf("%10llu", p[(wg * 8 + w) * 8 + i]); printf("\n"); }
  }
  return 0;
}
