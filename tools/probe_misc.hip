// probe_misc.hip -- round-2 hardware probes (torch-free):
//   1. v_permlane16_swap / v_permlane32_swap lane semantics (printed once)
//   2. cost of a kernel node in a HIP graph chain: empty 1-workgroup kernel, 512 x 256-thread workgroups with 72 KiB
//      of LDS each (the contraction kernel's launch shape), both doing nothing
//   3. store patterns of a [32768 x N] fp16 output written from registers by 512 four-wave workgroups (128 x 160
//      tiles): (a) 16-B stores, 64 B contiguous per row (what a transposed-accumulator epilogue can issue directly),
//      (b) 16-B stores, 160 B contiguous per row (what the LDS-slab epilogue issues), (c) 8-B stores
//   hipcc -O3 --offload-arch=gfx950 tools/probe_misc.hip -o tools/probe_misc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void k_perm(unsigned* o) {
  unsigned a = threadIdx.x, b = threadIdx.x + 100;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
  auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[128 + threadIdx.x] = s[0]; o[192 + threadIdx.x] = s[1];
}
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ __launch_bounds__(256, 2) void k_lds(int* p) {
  extern __shared__ char smem[];
  if (p && threadIdx.x == 9999) p[0] = smem[0];
}
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
// tile = blockIdx.x: tm = tile / tn_count, tn = tile % tn_count; 4 waves as 2 x 2, wave tile 64 x 80
template <int MODE> __global__ __launch_bounds__(256, 2) void k_store(_Float16* out, int M, int N, float v) {
  const int tiles_n = N / 160;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = tm * 128 + wm * 64, n0 = tn * 160 + wn * 80;
  h8 x; for (int i = 0; i < 8; ++i) x[i] = (_Float16)(v + i);
  if (MODE == 0) {   // transposed accumulators: lane (q, c): row i*16 + c; pair (j, j+1): 16 B at col (j + (q&1))*16 + 8*(q>>1)
    const int c = lane & 15, q = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long row = m0 + i * 16 + c;
#pragma unroll
      for (int jp = 0; jp < 2; ++jp)
        *reinterpret_cast<h8*>(out + row * N + n0 + (2 * jp + (q & 1)) * 16 + 8 * (q >> 1)) = x;
      h4 y; for (int k = 0; k < 4; ++k) y[k] = x[k];
      *reinterpret_cast<h4*>(out + row * N + n0 + 64 + 4 * q) = y;
    }
  } else if (MODE == 1) {   // slab pattern: 10 vectors per row, 6.4 rows per wave instruction
#pragma unroll
    for (int it = 0; it < 10; ++it) {
      const int id = it * 64 + lane;
      const int row = id / 10, cv = id % 10;
      *reinterpret_cast<h8*>(out + (long long)(m0 + row) * N + n0 + cv * 8) = x;
    }
  } else {   // 8-B stores in the accumulator layout: lane (q, c): row i*16 + c, cols j*16 + 4q
    const int c = lane & 15, q = lane >> 4;
    h4 y; for (int k = 0; k < 4; ++k) y[k] = x[k];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j)
        *reinterpret_cast<h4*>(out + (long long)(m0 + i * 16 + c) * N + n0 + j * 16 + 4 * q) = y;
  }
}

template <class F> static float graph_time(hipStream_t s, int n, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < n; ++i) launch();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return best * 1000.0f / n;
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  unsigned* d; CK(hipMalloc(&d, 256 * 4));
  hipLaunchKernelGGL(k_perm, dim3(1), dim3(64), 0, s, d);
  std::vector<unsigned> h(256); CK(hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost));
  const char* names[4] = {"permlane16_swap r[0] (a = lane, b = lane + 100)", "permlane16_swap r[1]", "permlane32_swap r[0]", "permlane32_swap r[1]"};
  for (int k = 0; k < 4; ++k) { printf("%s:", names[k]); for (int i = 0; i < 64; ++i) printf(" %u", h[k * 64 + i]); printf("\n"); }
  CK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 73728));
  printf("{\"probe\": \"graph node cost\", \"empty_1wg_us\": %.2f, \"empty_512wg_us\": %.2f, \"lds72k_512wg_us\": %.2f, \"lds72k_256wg_us\": %.2f}\n",
         graph_time(s, 200, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (int*)nullptr); }),
         graph_time(s, 200, [&] { hipLaunchKernelGGL(k_empty, dim3(512), dim3(256), 0, s, (int*)nullptr); }),
         graph_time(s, 200, [&] { hipLaunchKernelGGL(k_lds, dim3(512), dim3(256), 73728, s, (int*)nullptr); }),
         graph_time(s, 200, [&] { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 73728, s, (int*)nullptr); }));
  for (int N : {320, 1280}) {
    const int M = 32768;
    _Float16* o; CK(hipMalloc(&o, (size_t)M * N * 2));
    const int tiles = (M / 128) * (N / 160);
    float t0 = graph_time(s, 20, [&] { hipLaunchKernelGGL(k_store<0>, dim3(tiles), dim3(256), 0, s, o, M, N, 1.0f); });
    float t1 = graph_time(s, 20, [&] { hipLaunchKernelGGL(k_store<1>, dim3(tiles), dim3(256), 0, s, o, M, N, 1.0f); });
    float t2 = graph_time(s, 20, [&] { hipLaunchKernelGGL(k_store<2>, dim3(tiles), dim3(256), 0, s, o, M, N, 1.0f); });
    const double mb = (double)M * N * 2 / 1e6;
    printf("{\"probe\": \"store pattern\", \"M\": %d, \"N\": %d, \"MB\": %.1f, \"direct16_us\": %.2f, \"slab16_us\": %.2f, \"direct8_us\": %.2f, "
           "\"direct16_TBps\": %.2f, \"slab16_TBps\": %.2f, \"direct8_TBps\": %.2f}\n", M, N, mb, t0, t1, t2, mb / t0, mb / t1, mb / t2);
    CK(hipFree(o));
  }
  return 0;
}
