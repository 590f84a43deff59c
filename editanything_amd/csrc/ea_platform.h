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
// Exact (erf) GELU of the reference (F.gelu, ldm/modules/attention.py:54-56).  erf by Abramowitz-Stegun 7.1.26
// (|abs error| <= 1.5e-7, i.e. fp32 round-off class): 1 rcp + 1 exp + 8 FMA-class ops instead of libm erff's ~35
// instructions -- the GEGLU epilogue evaluates 10^7-10^8 of these per launch and was VALU-bound on erff.
__device__ __forceinline__ float ea_erf(float x) {
  const float ax = fabsf(x);
  const float t = 1.0f / (1.0f + 0.3275911f * ax);
  float poly = 1.061405429f;
  poly = poly * t - 1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t - 0.284496736f;
  poly = poly * t + 0.254829592f;
  const float r = 1.0f - poly * t * ea_expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float ea_gelu_erf(float x) {
  return 0.5f * x * (1.0f + ea_erf(x * 0.70710678118654752440f));
}
__device__ __forceinline__ f16x8 ea_ld8(const f16* p) { return *reinterpret_cast<const f16x8*>(p); }
__device__ __forceinline__ void ea_st8(f16* p, f16x8 v) { *reinterpret_cast<f16x8*>(p) = v; }
__device__ __forceinline__ f16x8 ea_zero8() {
  f16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (f16)0.0f;
  return z;
}
