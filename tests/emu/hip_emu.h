// hip_emu.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny single-OS-thread, fiber-based emulator of the HIP subset our gfx950
// kernels use (threadIdx/blockIdx, dynamic LDS, __syncthreads, wave64 shuffles,
// MFMA 32x32x16 f16).  It lets `pytest -m "not gpu"` execute the *same kernel
// source* that hipcc compiles for the MI355X, on the CPU, so index math,
// fragment layouts, guards and epilogues are checked without a GPU.
//
// It is never linked into the product library: editanything_amd/_lib.py loads
// only libeditanything_hip.so and fails loudly when that is missing.
#pragma once
#include <ucontext.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__

namespace ea_emu {
struct Dim3 {
  unsigned x, y, z;
  Dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern Dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
extern char* g_smem;
void yield_();
void block_sync();
void wave_sync();
int lane_id();
int wave_lanes();
// per-wave scratch: 64 lanes x 64 bytes
char* wave_scratch();
void launch(Dim3 grid, Dim3 block, size_t smem, const std::function<void()>& body);
}  // namespace ea_emu

typedef ea_emu::Dim3 dim3;
#define threadIdx ea_emu::g_threadIdx
#define blockIdx ea_emu::g_blockIdx
#define blockDim ea_emu::g_blockDim
#define gridDim ea_emu::g_gridDim

static inline void __syncthreads() { ea_emu::block_sync(); }

template <typename T>
static inline T ea_emu_shfl_xor(T v, int mask) {
  static_assert(sizeof(T) <= 64, "");
  char* s = ea_emu::wave_scratch();
  int l = ea_emu::lane_id();
  memcpy(s + l * 64, &v, sizeof(T));
  ea_emu::wave_sync();
  T r;
  int src = l ^ mask;
  if (src >= ea_emu::wave_lanes()) src = l;
  memcpy(&r, s + src * 64, sizeof(T));
  ea_emu::wave_sync();
  return r;
}
