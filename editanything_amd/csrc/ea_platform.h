// ea_platform.h -- gfx950 (MI355X / CDNA4) device layer used by every kernel.
//
// The product build is `hipcc --offload-arch=gfx950`; wave = 64 lanes, MFMA
// v_mfma_f32_32x32x16_f16, LDS 160 KiB/CU.  With -DEA_EMU (tests only) the
// same kernel bodies are compiled for the host against tests/emu/hip_emu.h so
// the CPU test-suite can execute them; nothing in the shipped library uses it.
#pragma once
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <stdio.h>

#ifdef EA_EMU
#include "hip_emu.h"
typedef void* hipStream_t;
#else
#include <hip/hip_runtime.h>
#endif

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define EA_WAVE 64

// ---------------------------------------------------------------- error codes
#define EA_OK 0
#define EA_ERR_BAD_SHAPE (-1)
#define EA_ERR_BAD_ARG (-2)
#define EA_ERR_UNSUPPORTED (-3)
#define EA_ERR_WORKSPACE (-4)
#define EA_ERR_LAUNCH (-5)

// ------------------------------------------------------------------ LDS decl
#ifdef EA_EMU
#define EA_SMEM(name) char* name = ea_emu::g_smem
#else
#define EA_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#endif

// ------------------------------------------------------------------- launch
// kfn must be a plain identifier (function pointer variable for templates).
#ifdef EA_EMU
#define EA_LAUNCH(kfn, grid, block, smem, stream, ...) \
  ea_emu::launch((grid), (block), (smem), [&]() { kfn(__VA_ARGS__); })
static inline int ea_launch_status() { return EA_OK; }
#else
// hipGetLastError() is per-thread and sticky: the host runtime around us (PyTorch's caching allocators poll
// hipEventQuery, which records hipErrorNotReady) can leave a stale code behind, so it is cleared right before each
// launch and ea_launch_status() then reports only this launch's own outcome.
#define EA_LAUNCH(kfn, grid, block, smem, stream, ...) \
  do { (void)hipGetLastError(); kfn<<<(grid), (block), (smem), (hipStream_t)(stream)>>>(__VA_ARGS__); } while (0)
static inline int ea_launch_status() {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return EA_OK;
  fprintf(stderr, "editanything_hip: kernel launch failed: %s (%s)\n", hipGetErrorName(e), hipGetErrorString(e));
  return EA_ERR_LAUNCH;
}
#endif

// For kernels that need > 64 KiB of dynamic LDS the attribute must be raised.
template <typename K>
static inline void ea_allow_big_lds(K kfn, int bytes) {
#ifndef EA_EMU
  if (bytes > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
#else
  (void)kfn;
  (void)bytes;
#endif
}

// ----------------------------------------------------------- wave collectives
__device__ __forceinline__ int ea_lane() {
#ifdef EA_EMU
  return ea_emu::lane_id();
#else
  return threadIdx.x & 63;
#endif
}

__device__ __forceinline__ float ea_shfl_xor(float v, int mask) {
#ifdef EA_EMU
  return ea_emu_shfl_xor<float>(v, mask);
#else
  return __shfl_xor(v, mask, 64);
#endif
}

__device__ __forceinline__ float ea_wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += ea_shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ float ea_wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, ea_shfl_xor(v, m));
  return v;
}

// ------------------------------------------------------------------- MFMA
// v_mfma_f32_32x32x16_f16 (gfx950).  Operand layout (lane l, element j<8):
//   A[i = l & 31][k = 8*(l >> 5) + j],  B[k = 8*(l >> 5) + j][n = l & 31]
//   C/D reg r<16: col = l & 31, row = (r & 3) + 8*(r >> 2) + 4*(l >> 5)
__device__ __forceinline__ f32x16 ea_mfma_32x32x16(f16x8 a, f16x8 b, f32x16 c) {
#ifdef EA_EMU
  char* s = ea_emu::wave_scratch();
  int l = ea_emu::lane_id();
  memcpy(s + l * 64, &a, 16);
  memcpy(s + l * 64 + 16, &b, 16);
  ea_emu::wave_sync();
  int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = c[r];
    for (int k = 0; k < 16; ++k) {
      f16 av, bv;
      memcpy(&av, s + (row + 32 * (k >> 3)) * 64 + (k & 7) * 2, 2);
      memcpy(&bv, s + (col + 32 * (k >> 3)) * 64 + 16 + (k & 7) * 2, 2);
      acc += (float)av * (float)bv;
    }
    c[r] = acc;
  }
  ea_emu::wave_sync();
  return c;
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}

__device__ __forceinline__ int ea_mfma_row(int r, int lane) {
  return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

// --------------------------------------------------------------- math helpers
__device__ __forceinline__ float ea_expf(float x) {
#ifdef EA_EMU
  return expf(x);
#else
  return __expf(x);
#endif
}
__device__ __forceinline__ float ea_silu(float x) { return x / (1.0f + ea_expf(-x)); }
// Exact (erf) GELU of the reference (F.gelu, ldm/modules/attention.py:54-56; SAM's MLPBlock):
//   gelu(x) = x Phi(x) = max(x, 0) - |x| * (1 - Phi(|x|)),   1 - Phi(a) = erfc(a / sqrt 2) / 2 = 2^-(1 + a Q(a)),
// Q a degree-6 polynomial fitted on a in [0, 4 sqrt 2] (beyond it 1 - Phi < 8e-9: the argument is clamped).  ONE transcendental
// (v_exp_f32) + 11 full-rate VALU ops, no reciprocal, no sign handling: |abs error| <= 5.1e-7 over [-12, 12], i.e. the fp32
// round-off class of the values themselves (the Abramowitz-Stegun 7.1.26 form used until round 4 -- 1 rcp + 1 exp + 13 ops --
// measured 6.8e-7 on the same grid; libm erff costs ~35 instructions).  The GEGLU epilogues evaluate 10^7-10^8 of these per
// launch and are VALU-bound on them: 21 us of an 89-us [32768 x 2560 x 320] launch, 8.7 us per tile round of SAM's mlp.lin1
// on the 256 x 256 kernel (profiles/r05_launch_phase_stamps.jsonl).  Fit + error check: tests/test_abi.py.
__device__ __forceinline__ float ea_exp2_raw(float x) {
#ifdef EA_EMU
  return exp2f(x);
#else
  return __builtin_amdgcn_exp2f(x);   // bare v_exp_f32: the argument here is <= -1, never a denormal input
#endif
}
__device__ __forceinline__ float ea_gelu_erf(float x) {
  const float a = fminf(fabsf(x), 5.65685424949238f);
  float q = -4.276626896171365e-06f;
  q = q * a + 1.2775987670465838e-05f;
  q = q * a + 0.0005758762708865106f;
  q = q * a - 0.007670961786061525f;
  q = q * a + 0.05294874310493469f;
  q = q * a + 0.45904383063316345f;
  q = q * a + 1.1511269807815552f;
  const float e = ea_exp2_raw(-a * q - 1.0f);       // 1 - Phi(a)
  // (a, not |x|: +inf stays +inf; beyond the clamp the term is < 5e-8.)  NaN: fminf / fmaxf return their non-NaN operand, so the
  // clamped form alone turns a NaN into -2e-8 and hides an upstream overflow; it is re-attached explicitly (round-5 advisor;
  // the build keeps NaN semantics: -fno-finite-math-only.  tests/test_abi.py pins NaN in -> NaN out through the C ABI's SiLU /
  // GELU entry on the emulator and the GPU)
  const float r = fmaxf(x, 0.0f) - a * e;
  return (x != x) ? x : r;
}
__device__ __forceinline__ f16x8 ea_ld8(const f16* p) { return *reinterpret_cast<const f16x8*>(p); }
__device__ __forceinline__ void ea_st8(f16* p, f16x8 v) { *reinterpret_cast<f16x8*>(p) = v; }
__device__ __forceinline__ f16x8 ea_zero8() {
  f16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (f16)0.0f;
  return z;
}
