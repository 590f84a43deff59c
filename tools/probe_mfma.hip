// probe_mfma.hip -- hardware probe (not product code): sustained MFMA rate of ONE wave vs TWO waves per SIMD, for the two fp16
// shapes, accumulators in place, non-trivial operands.  (Round 3: the persistent kernel's multiply loop with one multiplying
// wave per SIMD tops out at 1.39 PF with every memory operation removed -- is that the single-wave issue rate?)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_mfma.hip -o tools/probe_mfma && tools/probe_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int SHAPE, int NACC>
__global__ void mfma_loop(const f16x8* src, float* out, int iters) {
  const int t = threadIdx.x + blockIdx.x * blockDim.x;
  f16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = src[(t * 8 + i) & 4095]; b[i] = src[(t * 8 + 4 + i) & 4095]; }
  float s = 0.f;
  if (SHAPE == 16) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  } else {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
  }
  out[t] = s;
}

template <int SHAPE, int NACC>
static void run(const f16x8* src, float* out, int waves_per_simd) {
  const int threads = 256 * waves_per_simd, grid = 256, iters = 2000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  mfma_loop<SHAPE, NACC><<<grid, threads>>>(src, out, 10);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  mfma_loop<SHAPE, NACC><<<grid, threads>>>(src, out, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = (double)grid * (threads / 64) * iters * NACC * (SHAPE == 16 ? 16384.0 : 32768.0);
  printf("{\"mfma\": \"%s\", \"accumulators\": %d, \"waves_per_simd\": %d, \"TFLOPs\": %.0f, \"ns_per_mfma_per_wave\": %.2f}\n",
         SHAPE == 16 ? "16x16x32_f16" : "32x32x16_f16", NACC, waves_per_simd, flop / (ms * 1e-3) / 1e12, ms * 1e6 / (iters * NACC));
}

int main() {
  f16x8* src; CK(hipMalloc(&src, 4096 * 16));
  {
    _Float16 h[4096 * 8]; unsigned s = 1; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (_Float16)(((s >> 8) & 0xffff) / 65536.0f - 0.5f); }
    CK(hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice));
  }
  float* out; CK(hipMalloc(&out, 256 * 1024 * 4));
  for (int w : {1, 2, 4}) {
    run<16, 16>(src, out, w); run<16, 32>(src, out, w);
    run<32, 4>(src, out, w); run<32, 8>(src, out, w);
  }
  return 0;
}
