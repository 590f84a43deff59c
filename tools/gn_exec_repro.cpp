// gn_exec_repro.cpp -- stand-alone (no torch) reproducer for the round-4 GroupNorm statistics loss:
// `ea_groupnorm_f16` of a library build (the shipped one, or a tools/build_gn_repro.sh side build with one of the round-3
// loop forms) on stream A -- alone on fixed inputs, or in CHAIN position behind a contraction that has just written its
// input -- while a second host thread streams launches of the generic register-staged contraction kernel (M = 20 GEMMs:
// the workload that perturbed 30-70 % of the evaluations) on stream B.  Every call's partial sums (the statistics pass's
// output: [B][chunk][group][sum, sum of squares]) are copied aside and compared with the undisturbed call's, bit for bit;
// a mismatch is classified by which half (sum / sum of squares) moved.
//   hipcc -O2 -std=c++17 tools/gn_exec_repro.cpp -o tools/gn_exec_repro -ldl -lpthread
//   tools/gn_exec_repro <library.so> [iters=1000] [geom=0|1]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "../include/editanything_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef _Float16 f16;

typedef int (*gemm_fn)(const void*, int, const void*, int, int, int, int, int, long long, long long, long long, long long, const ea_epilogue*, void*,
                       size_t, void*);
typedef int (*gn_fn)(const void*, int, const void*, int, const void*, const float*, const float*, void*, int, int, int, float, int, void*, size_t,
                     void*);

static f16* dev_random(size_t n, unsigned seed, float scale) {
  std::vector<f16> h(n);
  unsigned s = seed;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (f16)((((s >> 8) & 0xFFFF) / 65536.0f * 2.0f - 1.0f) * scale); }
  f16* d;
  CK(hipMalloc(&d, n * 2));
  CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: gn_exec_repro <library.so> [iters] [geom]\n"); return 2; }
  const int iters = argc > 2 ? atoi(argv[2]) : 1000;
  const int geom = argc > 3 ? atoi(argv[3]) : 0;
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { printf("dlopen: %s\n", dlerror()); return 2; }
  gemm_fn gemm = (gemm_fn)dlsym(h, "ea_gemm_f16");
  gn_fn gn = (gn_fn)dlsym(h, "ea_groupnorm_f16");
  if (!gemm || !gn) { printf("missing symbols\n"); return 2; }

  // geometry 0: the decoder norm the round-4 watch caught first (x1 [8,32,32,1280] | x2 [8,32,32,640]); 1: [8,64,64,320] | 320
  const int B = 8, HW = geom ? 4096 : 1024, c1 = geom ? 320 : 1280, c2 = geom ? 320 : 640, C = c1 + c2, groups = 32;
  const int M = B * HW, Kp = 1280;
  f16* a_prod = dev_random((size_t)M * Kp, 11, 0.5f);       // producer: x1 = a_prod [M x 1280] W_prod^T
  f16* w_prod = dev_random((size_t)c1 * Kp, 12, 0.05f);
  f16* x1 = dev_random((size_t)M * c1, 13, 0.5f);
  f16* x2 = dev_random((size_t)M * c2, 14, 0.5f);
  f16* out;
  CK(hipMalloc(&out, (size_t)M * C * 2));
  std::vector<float> ones(C, 1.0f), zeros(C, 0.0f);
  float *gamma, *beta;
  CK(hipMalloc(&gamma, C * 4));
  CK(hipMalloc(&beta, C * 4));
  CK(hipMemcpy(gamma, ones.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(beta, zeros.data(), C * 4, hipMemcpyHostToDevice));
  const size_t ws_bytes = 64u << 20;
  char *ws_a, *ws_b;
  CK(hipMalloc(&ws_a, ws_bytes));
  CK(hipMalloc(&ws_b, ws_bytes));
  // neighbour operands (M = 20: the generic kernel)
  f16* a20 = dev_random(20 * 1280, 21, 0.1f);
  f16* w12 = dev_random(1280 * 1280, 22, 0.05f);
  f16* o20;
  CK(hipMalloc(&o20, 20 * 1280 * 2));

  // gn_plan (ea_norm.hip): chunks per sample -> bytes of partial sums
  int r = 256 / (C / 8); if (r < 1) r = 1; if (r > 32) r = 32; if (r > HW) r = HW;
  int nchunk = HW / (r * 4); int target = 2048 / B; if (nchunk > target) nchunk = target; if (nchunk > 128) nchunk = 128; if (nchunk < 1) nchunk = 1;
  const int chunk_px = (HW + nchunk - 1) / nchunk; nchunk = (HW + chunk_px - 1) / chunk_px;
  const size_t part_floats = (size_t)B * nchunk * groups * 2;
  float* parts;
  CK(hipMalloc(&parts, part_floats * 4 * (size_t)(iters + 1)));

  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  ea_epilogue ep;
  memset(&ep, 0, sizeof(ep));
  ep.scale = 1.0f; ep.out = x1; ep.ldc = c1;
  ea_epilogue en;
  memset(&en, 0, sizeof(en));
  en.scale = 1.0f; en.out = o20; en.ldc = 1280; en.act = EA_ACT_SILU;

  auto victim = [&](bool chain, int slot) {
    if (chain) {
      const int st = gemm(a_prod, Kp, w_prod, Kp, M, c1, Kp, 1, 0, 0, 0, 0, &ep, ws_a + (32u << 20), ws_bytes - (32u << 20), sa);
      if (st) { printf("producer failed %d\n", st); exit(1); }
    }
    const int st = gn(x1, c1, x2, c2, nullptr, gamma, beta, out, B, HW, groups, 1e-5f, 1, ws_a, 32u << 20, sa);
    if (st) { printf("groupnorm failed %d\n", st); exit(1); }
    CK(hipMemcpyAsync(parts + (size_t)slot * part_floats, ws_a, part_floats * 4, hipMemcpyDeviceToDevice, sa));
  };
  std::vector<float> ref(part_floats), got(part_floats);
  for (int chain = 0; chain < 2; ++chain) {
    for (int busy = 0; busy < 2; ++busy) {
      victim(chain != 0, 0);                                    // undisturbed reference (the producer's output is the same every call)
      CK(hipStreamSynchronize(sa));
      CK(hipMemcpy(ref.data(), parts, part_floats * 4, hipMemcpyDeviceToHost));
      std::atomic<bool> stop{false};
      std::atomic<long> side_launches{0};
      std::thread th;
      if (busy)
        th = std::thread([&]() {
          CK(hipSetDevice(0));
          while (!stop.load()) {
            for (int i = 0; i < 64; ++i) gemm(a20, 1280, w12, 1280, 20, 1280, 1280, 1, 0, 0, 0, 0, &en, ws_b, ws_bytes, sb);
            side_launches += 64;
            CK(hipStreamSynchronize(sb));
          }
        });
      for (int it = 1; it <= iters; ++it) {
        victim(chain != 0, it);
        if ((it & 15) == 0) CK(hipStreamSynchronize(sa));
      }
      CK(hipStreamSynchronize(sa));
      stop.store(true);
      if (busy) th.join();
      int bad_calls = 0, bad_sum = 0, bad_sq = 0;
      long first[4] = {-1, -1, -1, -1};
      for (int it = 1; it <= iters; ++it) {
        CK(hipMemcpy(got.data(), parts + (size_t)it * part_floats, part_floats * 4, hipMemcpyDeviceToHost));
        if (!memcmp(got.data(), ref.data(), part_floats * 4)) continue;
        ++bad_calls;
        for (size_t i = 0; i < part_floats; i += 2) {
          const bool ds = memcmp(&got[i], &ref[i], 4) != 0, dq = memcmp(&got[i + 1], &ref[i + 1], 4) != 0;
          if (!ds && !dq) continue;
          bad_sum += ds; bad_sq += dq;
          const int g = (int)((i / 2) % groups);
          if (first[0] < 0) { first[0] = it; first[1] = (long)(i / 2 / groups / nchunk); first[2] = (long)((i / 2 / groups) % nchunk); first[3] = g; }
        }
      }
      printf("{\"library\": \"%s\", \"geometry\": \"[%d,%d,%d]|%d\", \"chain_behind_producer\": %d, \"generic_kernel_neighbour\": %d, \"calls\": %d, "
             "\"calls_with_different_partials\": %d, \"partials_sum_differs\": %d, \"partials_sum_sq_differs\": %d, "
             "\"first (call, b, chunk, group)\": [%ld, %ld, %ld, %ld], \"side_launches\": %ld}\n",
             argv[1], B, HW, c1, c2, chain, busy, iters, bad_calls, bad_sum, bad_sq, first[0], first[1], first[2], first[3], side_launches.load());
      fflush(stdout);
    }
  }
  return 0;
}
