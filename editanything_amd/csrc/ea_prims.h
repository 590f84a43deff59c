// ea_prims.h -- gfx950 primitives shared by the LDS-DMA contraction kernels (ea_gemm2.h, ea_gemm3.h): the 16x16x32 fp16
// MFMA, buffer descriptors + LDS-DMA (buffer_load ... lds), counted vmcnt waits, raw barriers, v_permlane16_swap and the
// LDS row swizzle.  Under -DEA_EMU (CPU test-suite only) the same names execute on the host emulator.
#pragma once
#include "ea_platform.h"

#define EA_OOB 0xFFFFFFF0u  // per-lane byte offset beyond any descriptor: the DMA writes zeros
#define EA_BUF_BYTES 0x80000000u

#ifdef EA_EMU
__device__ __forceinline__ f32x4 ea_mfma_16x16x32(f16x8 a, f16x8 b, f32x4 c) {
  char* s = ea_emu::wave_scratch();
  int l = ea_emu::lane_id();
  memcpy(s + l * 64, &a, 16);
  memcpy(s + l * 64 + 16, &b, 16);
  ea_emu::wave_sync();
  const int col = l & 15;
  for (int r = 0; r < 4; ++r) {
    const int row = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 32; ++k) {
      f16 av, bv;
      memcpy(&av, s + (row + 16 * (k >> 3)) * 64 + (k & 7) * 2, 2);
      memcpy(&bv, s + (col + 16 * (k >> 3)) * 64 + 16 + (k & 7) * 2, 2);
      acc += (float)av * (float)bv;
    }
    c[r] = acc;
  }
  ea_emu::wave_sync();
  return c;
}
typedef const char* ea_rsrc;
__device__ __forceinline__ ea_rsrc ea_make_rsrc(const void* p) { return (const char*)p; }
// LDS-DMA: lane l's 16 bytes land at (wave-uniform) lds_base + 16*l; out-of-range lanes get zeros
__device__ __forceinline__ void ea_dma16(ea_rsrc r, unsigned voff, unsigned soff, char* lds_base) {
  char* dst = lds_base + 16 * ea_emu::lane_id();
  if (voff >= EA_BUF_BYTES) memset(dst, 0, 16);
  else memcpy(dst, r + voff + soff, 16);
}
__device__ __forceinline__ int ea_uniform(int v) { return v; }
template <int N> __device__ __forceinline__ void ea_wait_dma() {}
__device__ __forceinline__ void ea_raw_barrier() { ea_emu::block_sync(); }
__device__ __forceinline__ void ea_wave_lds_sync() { ea_emu::wave_sync(); }
#else
// v_mfma_f32_16x16x32_f16: A[i = l & 15][k = 8*(l >> 4) + j], B[k = 8*(l >> 4) + j][n = l & 15],
// C/D reg r < 4: col = l & 15, row = 4*(l >> 4) + r.
__device__ __forceinline__ f32x4 ea_mfma_16x16x32(f16x8 a, f16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
typedef __amdgpu_buffer_rsrc_t ea_rsrc;
__device__ __forceinline__ ea_rsrc ea_make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, EA_BUF_BYTES, 0x00020000);
}
__device__ __forceinline__ void ea_dma16(ea_rsrc r, unsigned voff, unsigned soff, char* lds_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ int ea_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Counted wait: returns when at most N of this wave's LDS-DMA instructions are still in flight (they complete in
// order), so younger tiles keep streaming across the barrier (guide T3+T4; never __syncthreads() here: its fence
// would drain vmcnt to 0).
template <int N> __device__ __forceinline__ void ea_wait_dma() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void ea_raw_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// LDS hand-off between the lanes of ONE wave: the LDS pipeline is in order per wave, so retiring this wave's
// outstanding DS operations is enough; no workgroup barrier.
__device__ __forceinline__ void ea_wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
#endif

// v_permlane16_swap_b32 a, b: the ODD 16-lane rows of `a` are exchanged with the EVEN rows of `b`
// (a' = [a.row0, b.row0, a.row2, b.row2], b' = [a.row1, b.row1, a.row3, b.row3]; probed on the MI355X,
// tools/probe_misc.hip).  Used by the register-direct epilogue to turn two 4-column accumulator quads into 8
// consecutive columns per lane (guide T21 applied to the 16x16 MFMA layout).
#ifdef EA_EMU
__device__ __forceinline__ void ea_swap16(float& a, float& b) {
  char* s = ea_emu::wave_scratch();
  const int l = ea_emu::lane_id();
  memcpy(s + l * 64 + 32, &a, 4);
  memcpy(s + l * 64 + 36, &b, 4);
  ea_emu::wave_sync();
  float na = a, nb = b;
  if ((l >> 4) & 1) memcpy(&na, s + (l - 16) * 64 + 36, 4);
  else memcpy(&nb, s + (l + 16) * 64 + 32, 4);
  ea_emu::wave_sync();
  a = na;
  b = nb;
}
#else
__device__ __forceinline__ void ea_swap16(float& a, float& b) {
  // both results go through scalar `unsigned` temporaries: bit-casting the elements of the returned vector directly
  // (`bit_cast<float>(r[1])`) makes hipcc (ROCm 7.2) treat r[1] as r[0] -- seen in the ISA and on the MI355X
  const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
  const auto r = __builtin_amdgcn_permlane16_swap(ua, ub, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  a = __builtin_bit_cast(float, r0);
  b = __builtin_bit_cast(float, r1);
}
#endif

// n / d for 0 <= n < 2^22 and d >= 1 without the ~30-instruction integer division sequence: float estimate (v_rcp_f32 is within
// 1 ulp, the product rounds once: |error| < 1 for n < 2^22) + two correction steps each way.  The im2col set-up of a
// convolution launch does 2 such divisions per A piece and lane (pixel -> sample, row, column): 1.4-1.9 us of set-up per workgroup
// against 0.8 us for a dense launch (profiles/r05_launch_phase_stamps.jsonl).  Exactness: tests/test_abi.py brute-forces the
// same expression in float32 over every n < 2^22 for the workload's divisors.
__device__ __forceinline__ int ea_div_small(int n, int d, float rcp) {
  int q = (int)((float)n * rcp);
  const int r = n - q * d;
  q += (int)(r >= d) + (int)(r >= 2 * d) - (int)(r < 0) - (int)(r < -d);
  return q;
}
#define EA_DIV_SMALL_MAX (1 << 22)

// 16-B chunk swizzle of a 128-B LDS row (8 chunks): conflict-free ds_read_b128 for 16 consecutive rows.
__device__ __forceinline__ int ea_swz(int row) { return (row >> 1) & 7; }

#include <type_traits>
// ds_read_b64_tr_b16: within each 16-lane group the 16 x 8-byte pieces form a [4][16] fp16 block (row r = lanes
// 4r..4r+3, each supplying 4 consecutive columns); lane i receives column i = (M[0][i], M[1][i], M[2][i], M[3][i]).
// (Mapping measured on gfx950 with tools/probe_tr.hip.)  Lets V stay row-major [key][d] in LDS -- written with plain
// 16-byte stores -- and still be consumed as the V^T operand of O^T = V^T P^T.
template <int OFF>
__device__ __forceinline__ f16x4 ea_lds_read_tr16(const char* ptr0) {
  const char* ptr = ptr0 + OFF;
#ifdef EA_EMU
  char* sc = ea_emu::wave_scratch();
  const int l = ea_emu::lane_id();
  memcpy(sc + l * 64, ptr, 8);
  ea_emu::wave_sync();
  const int g = l & ~15, i = l & 15;
  f16x4 r;
  for (int rr = 0; rr < 4; ++rr) {
    f16 v;
    memcpy(&v, sc + (g + 4 * rr + (i >> 2)) * 64 + (i & 3) * 2, 2);
    r[rr] = v;
  }
  ea_emu::wave_sync();
  return r;
#else
  f16x4 r;
  (void)ptr;
  const unsigned addr = (unsigned)(uintptr_t)ptr0;   // one address VGPR per tile, the rest is the 16-bit immediate
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
#endif
}
// the asm read above is invisible to the compiler's lgkmcnt bookkeeping: wait for it explicitly before the first use,
// and keep the consumers behind the wait (guide section 5.4 rule 18)
template <int N, int I = 0, typename F>
__device__ __forceinline__ void ea_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    ea_static_for<N, I + 1>(f);
  }
}
__device__ __forceinline__ void ea_lds_tr_wait() {
#ifndef EA_EMU
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#endif
}
