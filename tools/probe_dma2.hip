// probe_dma2.hip -- hardware probe (not product code): LDS-DMA (buffer_load_dwordx4 ... lds) throughput per CU as a function
// of the per-lane SOURCE address pattern of one 1-KiB instruction.  Round 3 found the contraction kernels bound by the DMA
// issue path (ea_gemm3 phase totals: ~150-195 cycles of a wave per instruction, ~24 B/clk/CU) -- is it the 16-byte-granular
// XOR swizzle of the source chunks (bank-conflict-free fragment reads) that costs the address path its coalescing?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_dma2.hip -o tools/probe_dma2 && tools/probe_dma2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// pattern -> byte offset of lane l inside a [rows x pitch] source panel
__device__ __forceinline__ unsigned lane_off(int pat, int l, unsigned pitch) {
  switch (pat) {
    case 0: return l * 16u;                                                     // 1 KiB contiguous
    case 1: { const int r = l >> 3, c = l & 7; return r * pitch + c * 16u; }     // 8 rows x 128 B, in order
    case 2: { const int r = l >> 3, c = (l & 7) ^ ((r >> 1) & 7); return r * pitch + c * 16u; }   // ... 16-B XOR (rows of ONE instruction: r>>1 in 0..3)
    case 3: { const int r = l >> 3, c = (l & 7) ^ (l >> 3); return r * pitch + c * 16u; }          // ... 16-B XOR, all 8 rows different (as a tile's rows 2i)
    case 4: { const int r = l >> 3, c = (l & 7) ^ (((l >> 3) & 3) << 1); return r * pitch + c * 16u; }   // 32-B-granular XOR
    case 5: { const int r = l >> 3, c = (l & 7) ^ (((l >> 3) & 1) << 2); return r * pitch + c * 16u; }   // 64-B-granular XOR
    case 6: { const int r = l >> 2, c = l & 3; return r * pitch + c * 16u; }     // 16 rows x 64 B, in order
    case 7: { const int r = l >> 2, c = (l & 3) ^ (r & 3); return r * pitch + c * 16u; }   // 16 rows x 64 B, 16-B XOR
    case 8: { const int r = l >> 4, c = l & 15; return r * pitch + c * 16u; }    // 4 rows x 256 B, in order
    case 9: { const int r = l >> 3, c = 7 - (l & 7); return r * pitch + c * 16u; }   // 8 rows x 128 B, chunks REVERSED
  }
  return 0;
}

template <int DEPTH>
__global__ __launch_bounds__(1024) void dma_probe(const char* src, int pat, unsigned pitch, int rows_per_wg, int iters, unsigned long long* cyc, int shared) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  // each workgroup walks its own panel of rows_per_wg rows x pitch bytes (L2 resident when small), K position cycling
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (shared ? (size_t)(blockIdx.x & 7) * rows_per_wg * pitch : (size_t)blockIdx.x * rows_per_wg * pitch)), 0, 0x80000000u, 0x00020000);
  const unsigned loff = lane_off(pat, lane, pitch);
  const int rows_per_instr = pat == 0 ? 0 : (pat == 6 || pat == 7) ? 16 : pat == 8 ? 4 : 8;
  const unsigned seg = pat == 0 ? 1024u : (pat == 6 || pat == 7) ? 64u : pat == 8 ? 256u : 128u;
  char* ring = smem + wave * DEPTH * 1024;
  const unsigned long long t0 = __builtin_readcyclecounter();
  int rb = wave * rows_per_instr, kb = 0;     // row block of this wave's next instruction, K byte offset
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
      unsigned soff;
      if (pat == 0) soff = ((unsigned)(it * DEPTH + d) * nw + wave) * 1024u % (rows_per_wg * pitch - 1024u);
      else soff = (unsigned)rb * pitch + (unsigned)kb;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(ring + d * 1024), 16, loff, soff, 0, 0);
      if (pat != 0) {
        rb += nw * rows_per_instr;
        if (rb + rows_per_instr > rows_per_wg) { rb = wave * rows_per_instr; kb += seg; if (kb + seg > pitch) kb = 0; }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0 && wave == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int DEPTH>
static void run(const char* src, unsigned long long* cyc, int nwaves, int pat, unsigned pitch, int shared) {
  if (nwaves * DEPTH > 144) return;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute((const void*)dma_probe<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, nwaves * DEPTH * 1024));
  const int rows = shared ? 288 : 128, iters = 2048 / DEPTH, grid = 256;
  dma_probe<DEPTH><<<grid, nwaves * 64, nwaves * DEPTH * 1024>>>(src, pat, pitch, rows, 4, cyc, shared);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  dma_probe<DEPTH><<<grid, nwaves * 64, nwaves * DEPTH * 1024>>>(src, pat, pitch, rows, iters, cyc, shared);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(256); CK(hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
  double c = 0; for (auto v : h) c += v; c /= 256;
  const double instr = (double)iters * DEPTH * nwaves;
  printf("{\"shared\": %d, \"waves\": %d, \"outstanding_per_wave\": %d, \"pattern\": %d, \"cycles_per_instr_per_wave\": %.1f, \"bytes_per_clk_per_cu\": %.1f, \"in_flight_KB_per_cu\": %d, \"implied_latency_cycles\": %.0f, \"chip_TB_s\": %.2f}\n",
         shared, nwaves, DEPTH, pat, c / (iters * DEPTH), instr * 1024 / c, nwaves * DEPTH, (double)nwaves * DEPTH * 1024 / (instr * 1024 / c), (double)grid * instr * 1024 / (ms * 1e-3) / 1e12);
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  char* src; CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 1, bytes));
  unsigned long long* cyc; CK(hipMalloc(&cyc, 256 * 8));
  // shared = 1: the 32 CUs of an XCD read ONE panel (every L2 line is requested 32 times); shared = 0 with a small private
  // panel per workgroup (64 rows x 640 B = 40 KB, 1.3 MB per XCD: L2 resident, every line requested by ONE CU)
  for (int shared : {1, 0})
    for (int nwaves : {4, 8, 16}) {
      run<8>(src, cyc, nwaves, 2, 640, shared);
      run<16>(src, cyc, nwaves, 2, 640, shared);
    }
  return 0;
}
