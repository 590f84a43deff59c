// probe_overlap.hip -- can one SIMD run a VALU-only wave and an MFMA-only wave at the same time?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_overlap.hip -o tools/probe_overlap
// One workgroup of 8 waves per CU (waves i and i + 4 share SIMD i).  Modes: 0 = waves 0-3 MFMA chain, 4-7 idle;
// 1 = waves 4-7 VALU chain, 0-3 idle; 2 = both at once; 3 = all eight MFMA (half the count each); 4 = all eight VALU (half
// each); 5 = one wave alternating MFMA / VALU (4 independent FMAs after every MFMA); 6 = as 1 with v_exp_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NV>
__global__ __launch_bounds__(512) void probe(float* out, int mode, int n) {
  const int wave = threadIdx.x >> 6;
  const int prio = mode >> 8;      // 1: the VALU waves run at s_setprio 3, 2: the MFMA waves do
  mode &= 255;
  if (prio == 1 && __builtin_amdgcn_readfirstlane(wave) >= 4) __builtin_amdgcn_s_setprio(3);
  if (prio == 2 && __builtin_amdgcn_readfirstlane(wave) < 4) __builtin_amdgcn_s_setprio(3);
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.002f * j); }
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
  float v[8];
  for (int k = 0; k < 8; ++k) v[k] = 1.0f + 0.001f * (threadIdx.x + k);
  const bool do_mfma = mode == 0 ? wave < 4 : mode == 1 || mode == 6 ? false : mode == 2 ? wave < 4 : mode == 3 ? true : mode == 4 ? false : wave < 4;
  const bool do_valu = mode == 0 ? false : (mode == 1 || mode == 6) ? wave >= 4 : mode == 2 ? wave >= 4 : mode == 3 ? false : mode == 4 ? true : false;
  const int cnt = (mode == 3 || mode == 4) ? n / 2 : n;
  if (mode == 5) {
    if (wave < 4)
      for (int i = 0; i < n; i += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u], 0, 0, 0);
#pragma unroll
          for (int k = 0; k < NV; ++k) v[k] = __builtin_fmaf(v[k], 1.0001f, 0.5f);
        }
      }
  } else if (do_mfma) {
    for (int i = 0; i < cnt; i += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u], 0, 0, 0);
    }
  } else if (do_valu) {
    if (mode == 6) {
      for (int i = 0; i < cnt; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __builtin_amdgcn_exp2f(v[k]) * 0.5f;
    } else {
      for (int i = 0; i < cnt; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __builtin_fmaf(v[k], 1.0001f, 0.5f);
    }
  }
  float s = 0.0f;
  for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][7];
  for (int k = 0; k < 8; ++k) s += v[k];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

int main() {
  float* d;
  CK(hipMalloc(&d, 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int n = 400000;    // MFMAs per wave (32 cycles each) / VALU groups of 8 FMAs per wave
  const char* names[] = {"MFMA waves only (4 of 8)", "VALU waves only (4 of 8)", "MFMA waves + VALU waves, one of each per SIMD", "8 MFMA waves, n/2 each",
                         "8 VALU waves, n/2 each", "one wave per SIMD: MFMA + 4 FMA interleaved", "VALU waves only, v_exp_f32 + mul",
                         "one wave per SIMD: MFMA + 6 FMA", "one wave per SIMD: MFMA + 8 FMA", "one wave per SIMD: MFMA + 12 FMA",
                         "MFMA waves + VALU waves, the VALU waves at s_setprio 3", "MFMA waves + VALU waves, the MFMA waves at s_setprio 3"};
  for (int w = 0; w < 40; ++w) probe<4><<<256, 512>>>(d, 2, n);    // clocks up
  CK(hipDeviceSynchronize());
  for (int pass = 0; pass < 2; ++pass)
  for (int mode = 0; mode < 12; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0));
      if (mode < 7) probe<4><<<256, 512>>>(d, mode, n);
      else if (mode == 7) probe<6><<<256, 512>>>(d, 5, n);
      else if (mode == 8) probe<8><<<256, 512>>>(d, 5, n);
      else if (mode == 9) probe<12><<<256, 512>>>(d, 5, n);
      else if (mode == 10) probe<4><<<256, 512>>>(d, 2 | 256, n);
      else probe<4><<<256, 512>>>(d, 2 | 512, n);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep && ms < best) best = ms;
    }
    printf("{\"mode\": %d, \"what\": \"%s\", \"ms\": %.3f, \"ns_per_iter\": %.2f}\n", mode, names[mode], best, best * 1e6f / n);
  }
  return 0;
}
