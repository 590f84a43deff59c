// probe_overlap2.hip -- issue patterns of MFMA + VALU with one or two waves per SIMD (follow-up to probe_overlap.hip).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_overlap2.hip -o tools/probe_overlap2
// Every wave runs `n` groups of G MFMAs (32x32x16 f16, 32 cycles) and G*NV independent VALU operations (fma, or fma + exp2 for
// every fourth when EXP), either interleaved (1 MFMA, NV VALU) x G or clustered (G MFMAs, then G*NV VALU); WAVES = 4 or 8 per
// CU.  Output: SIMD cycles per MFMA (at the measured rate of a pure MFMA stream = 32).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) {
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  return c;
}
__device__ __forceinline__ float valu(bool e, float x, float c1) {
  if (e) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
  else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c1));
  return x;
}

template <int G, int NV, bool CLUSTER, bool EXP>
__global__ __launch_bounds__(512) void probe(float* out, int n) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.002f * j); }
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
  float c1 = 1.0001f + 1e-6f * threadIdx.x;
  constexpr int NVT = NV > 0 ? NV : 1;
  float v[NVT];
  for (int k = 0; k < NVT; ++k) v[k] = 1.0f + 0.001f * (threadIdx.x + k);
  for (int i = 0; i < n; ++i) {
    if (CLUSTER) {
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g & 3] = mfma(a, b, acc[g & 3]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = valu(EXP && (k & 3) == 3, v[k], c1);
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        acc[g & 3] = mfma(a, b, acc[g & 3]);
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = valu(EXP && (k & 3) == 3, v[k], c1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0.0f;
  for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][7];
  for (int k = 0; k < NVT; ++k) s += v[k];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int G, int NV, bool CLUSTER, bool EXP>
static int run(float* d, int waves, float ref_ns, const char* what) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int n = 40000;
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    probe<G, NV, CLUSTER, EXP><<<256, 64 * waves>>>(d, n);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  const float ns_per_mfma = best * 1e6f / (float(n) * G * (waves / 4));     // per SIMD
  printf("{\"G\": %d, \"valu_per_mfma\": %d, \"order\": \"%s\", \"exp\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"simd_cycles_per_mfma\": %.1f, \"what\": \"%s\"}\n",
         G, NV, CLUSTER ? "clustered" : "interleaved", int(EXP), waves / 4, best, ref_ns > 0 ? 32.0f * ns_per_mfma / ref_ns : 0.0f, what);
  return 0;
}

int main() {
  float* d;
  CK(hipMalloc(&d, 4096));
  for (int w = 0; w < 30; ++w) probe<8, 0, false, false><<<256, 256>>>(d, 40000);
  CK(hipDeviceSynchronize());
  // reference: pure MFMA stream, one wave per SIMD
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  probe<8, 0, false, false><<<256, 256>>>(d, 40000);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const float ref = ms * 1e6f / (40000.0f * 8);
  printf("{\"reference_ns_per_mfma\": %.2f}\n", ref);
  for (int waves = 4; waves <= 8; waves += 4) {
    run<8, 0, false, false>(d, waves, ref, "MFMA only");
    run<8, 4, false, false>(d, waves, ref, "");
    run<8, 8, false, false>(d, waves, ref, "");
    run<8, 10, false, false>(d, waves, ref, "attention-like ratio");
    run<8, 10, true, false>(d, waves, ref, "attention-like ratio, what hipcc emits today");
    run<8, 10, false, true>(d, waves, ref, "with a quarter exponentials");
    run<8, 10, true, true>(d, waves, ref, "with a quarter exponentials");
    run<8, 16, false, false>(d, waves, ref, "");
    run<8, 16, true, false>(d, waves, ref, "");
  }
  return 0;
}
