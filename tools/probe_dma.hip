// probe_dma.hip -- hardware probe (not product code): per-CU global->LDS (LDS-DMA) and global->VGPR streaming rates
// from an L2/MALL-resident source, versus bytes in flight and waves per workgroup.  Sizes the staging pipeline of
// ea_gemm2.h.   hipcc --offload-arch=gfx950 -O3 tools/probe_dma.hip -o /tmp/probe_dma && /tmp/probe_dma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Each wave streams `iters` rounds of DEPTH outstanding 1-KiB DMA instructions (lane-linear 16 B per lane) into its
// private LDS ring; the source walks a per-workgroup window of `win_bytes` (L2 resident when small).
template <int DEPTH>
__global__ __launch_bounds__(512) void dma_stream(const char* src, long long win_bytes, int shared, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (shared ? 0 : (long long)blockIdx.x * win_bytes)), 0,
                                                                 0x80000000u, 0x00020000);
  char* ring = smem + wave * DEPTH * 1024;
  const unsigned per_round = (unsigned)nw * DEPTH * 1024u;
  unsigned off = (unsigned)wave * DEPTH * 1024u + lane * 16u;
  // prologue: fill the ring
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(ring + d * 1024), 16, off + d * 1024u, 0, 0, 0);
  for (int it = 1; it < iters; ++it) {
    off += per_round;
    if (off + DEPTH * 1024u > (unsigned)win_bytes) off = (unsigned)wave * DEPTH * 1024u + lane * 16u;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      // wait until the oldest of the DEPTH outstanding instructions has landed, then reuse its slot
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(ring + d * 1024), 16, off + d * 1024u, 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && iters < 0) sink[blockIdx.x] = *(float*)smem;
}

template <int DEPTH>
__global__ __launch_bounds__(512) void reg_stream(const char* src, long long win_bytes, int shared, int iters, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  const char* base = src + (shared ? 0 : (long long)blockIdx.x * win_bytes);
  const unsigned per_round = (unsigned)nw * DEPTH * 1024u;
  unsigned off = (unsigned)wave * DEPTH * 1024u + lane * 16u;
  f32x4 acc = {0, 0, 0, 0};
  f32x4 v[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) v[d] = *(const f32x4*)(base + off + d * 1024u);
  for (int it = 1; it < iters; ++it) {
    off += per_round;
    if (off + DEPTH * 1024u > (unsigned)win_bytes) off = (unsigned)wave * DEPTH * 1024u + lane * 16u;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      acc += v[d];
      v[d] = *(const f32x4*)(base + off + d * 1024u);
    }
  }
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) acc += v[d];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 1234.5f) sink[blockIdx.x] = acc[0];
}

template <typename K>
static double run(K kfn, const char* src, long long win, int shared, int threads, int depth, int blocks, size_t smem) {
  const int iters = 400;
  float* sink;
  hipMalloc(&sink, 4096 * 4);
  hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  kfn<<<blocks, threads, smem>>>(src, win, shared, iters, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kfn<<<blocks, threads, smem>>>(src, win, shared, iters, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(sink);
  const double bytes = (double)blocks * (threads / 64) * depth * 1024.0 * iters;
  return bytes / (ms * 1e-3) / 1e9;  // GB/s
}

int main() {
  const long long total = 2LL << 30;
  char* src;
  hipMalloc(&src, total);
  hipMemset(src, 1, total);
  printf("kind   waves/WG WG/CU depth(KiB in flight per CU)  window/WG   GB/s total   GB/s per CU\n");
  // shared 256 KiB window (every WG reads the same bytes: pure L2 hits, like a weight panel), private 256 KiB windows
  // (64-128 MiB footprint: Infinity Cache), private 3 MiB windows (0.75-1.5 GiB: HBM)
  const long long wins[3] = {256 << 10, 256 << 10, 3 << 20};
  const int shareds[3] = {1, 0, 0};
  for (int w = 0; w < 3; ++w)
    for (int threads : {256, 512})
      for (int wgpcu : {1, 2}) {
        const int blocks = 256 * wgpcu;
        const int nw = threads / 64;
#define RUN(D)                                                                                                   \
  {                                                                                                              \
    double g = run(dma_stream<D>, src, wins[w], shareds[w], threads, D, blocks, (size_t)nw * D * 1024);                      \
    printf("dma%s  %d        %d     %2d (%4d)   %8lld   %9.0f   %7.1f\n", shareds[w] ? "S" : "P", nw, wgpcu, D, nw * D * wgpcu, wins[w], g, g / 256); \
    double r = run(reg_stream<D>, src, wins[w], shareds[w], threads, D, blocks, 0);                                          \
    printf("reg%s  %d        %d     %2d (%4d)   %8lld   %9.0f   %7.1f\n", shareds[w] ? "S" : "P", nw, wgpcu, D, nw * D * wgpcu, wins[w], r, r / 256); \
  }
        RUN(2) RUN(4) RUN(8) RUN(16)
      }
  return 0;
}
