// Does a value written by launch W reach launch R (same stream, next launch) when ANOTHER stream is issuing many short launches?
// (round 4: GroupNorm partial sums handed from the statistics launch to the normalise launch through a small scratch buffer at a
// fixed address changed an output once in ~2000 calls beside a stream of tiny GEMMs -- tools/diag_kernel_race3.py.)
//   W: 64 workgroups, each writes 64 floats (256 B) = the iteration number
//   R: 1024 workgroups x 256 threads, every thread reads one float of the 4096 and counts a mismatch
// Build + run:  hipcc --offload-arch=gfx950 -O2 -o /tmp/probe_handoff tools/probes/probe_handoff.cpp -lpthread && /tmp/probe_handoff 200000
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>

__global__ void w_kernel(float* buf, float v) { buf[blockIdx.x * 64 + threadIdx.x] = v; }
template <int MODE>
__global__ void r_kernel(const float* buf, float v, unsigned* bad, float* sink) {
  const int i = (blockIdx.x * 256 + threadIdx.x) & 4095;
  float x;
  if (MODE == 0) x = buf[i];
  else x = __hip_atomic_load(buf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (x != v) atomicAdd(bad, 1u);
  if (x == -1.0f) sink[0] = x;
}
__global__ void tiny_kernel(float* p, int n) {
  float s = 0.0f;
  for (int i = threadIdx.x; i < n; i += 64) s += p[i];
  if (s == 12345.0f) p[0] = s;
}

// the same hand-off with a chip-wide producer: W2 writes `n` floats (tens of MiB: most of it still dirty in the eight L2s when the launch
// ends), R2 reads all of it and counts values that are not this iteration's
__global__ void w2_kernel(float4* buf, float v, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) buf[i] = float4{v, v, v, v};
}
__global__ void r2_kernel(const float4* buf, float v, size_t n4, unsigned* bad, unsigned* first) {
  unsigned mine = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 x = buf[i];
    if (x.x != v || x.y != v || x.z != v || x.w != v) { ++mine; atomicMin(first, (unsigned)(i >> 4)); }    // 256-byte block index
  }
  if (mine) atomicAdd(bad, mine);
}

static void big_handoff(int iters, size_t mib, hipStream_t s1, hipStream_t s2, float* junk) {
  float4* side_buf;                      // the second stream's own 32 MiB: its launches leave the L2s full of dirty lines
  const size_t side_n4 = (32u << 20) / 16;
  hipMalloc(&side_buf, side_n4 * 16);
  const size_t n4 = mib * (1 << 20) / 16;
  float4* buf;
  unsigned* cnt;
  hipMalloc(&buf, n4 * 16); hipMalloc(&cnt, 8);
  for (int interfere = 0; interfere < 3; ++interfere) {
    unsigned init[2] = {0u, 0xffffffffu};
    hipMemcpy(cnt, init, 8, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    std::atomic<bool> stop{false};
    std::thread th([&]() {
      if (!interfere) return;
      float sv = 1.0f;
      while (!stop.load()) {
        for (int k = 0; k < 16; ++k) {
          if (interfere == 2) { hipLaunchKernelGGL(w2_kernel, dim3(1024), dim3(256), 0, s2, side_buf, sv, side_n4); sv += 1.0f; }
          for (int t = 0; t < 4; ++t) hipLaunchKernelGGL(tiny_kernel, dim3(20), dim3(64), 0, s2, junk, 4096);
        }
        hipStreamSynchronize(s2);
      }
    });
    int bad_iters = 0;
    unsigned prev = 0;
    for (int it = 0; it < iters; ++it) {
      const float v = (float)(it % 100000 + 1);
      hipLaunchKernelGGL(w2_kernel, dim3(2048), dim3(256), 0, s1, buf, v, n4);
      hipLaunchKernelGGL(r2_kernel, dim3(2048), dim3(256), 0, s1, buf, v, n4, cnt, cnt + 1);
      if ((it & 63) == 63) {
        hipStreamSynchronize(s1);
        unsigned h[2];
        hipMemcpy(h, cnt, 8, hipMemcpyDeviceToHost);
        if (h[0] != prev) { ++bad_iters; prev = h[0]; }
      }
    }
    hipStreamSynchronize(s1);
    stop.store(true);
    th.join();
    hipDeviceSynchronize();
    unsigned h[2];
    hipMemcpy(h, cnt, 8, hipMemcpyDeviceToHost);
    printf("{\"probe\": \"chip-wide W -> R hand-off\", \"MiB\": %zu, \"iterations\": %d, \"second_stream\": \"%s\", \"stale_float4_reads\": %u, "
           "\"first_stale_256B_block\": %d, \"groups_of_64_iterations_with_a_stale_read\": %d}\n", mib, iters, interfere == 0 ? "idle" : interfere == 1 ? "tiny launches" : "32 MiB writes + tiny launches", h[0], h[0] ? (int)h[1] : -1, bad_iters);
    fflush(stdout);
  }
  hipFree(buf); hipFree(cnt); hipFree(side_buf);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 100000;
  float *buf, *sink, *junk;
  unsigned* bad;
  hipMalloc(&buf, 4096 * 4); hipMalloc(&sink, 64); hipMalloc(&junk, 1 << 20); hipMalloc(&bad, 4);
  hipMemset(junk, 0, 1 << 20);
  hipStream_t s1, s2;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  for (int interfere = 0; interfere < 2; ++interfere)
    for (int mode = 0; mode < 2; ++mode) {
      hipMemset(bad, 0, 4);
      hipDeviceSynchronize();
      std::atomic<bool> stop{false};
      long long side_launches = 0;
      std::thread th([&]() {
        if (!interfere) return;
        while (!stop.load()) {
          for (int k = 0; k < 64; ++k) { hipLaunchKernelGGL(tiny_kernel, dim3(20), dim3(64), 0, s2, junk, 4096); ++side_launches; }
          hipStreamSynchronize(s2);
        }
      });
      for (int it = 0; it < iters; ++it) {
        const float v = (float)(it % 100000);
        hipLaunchKernelGGL(w_kernel, dim3(64), dim3(64), 0, s1, buf, v);
        if (mode == 0) hipLaunchKernelGGL(r_kernel<0>, dim3(1024), dim3(256), 0, s1, buf, v, bad, sink);
        else hipLaunchKernelGGL(r_kernel<1>, dim3(1024), dim3(256), 0, s1, buf, v, bad, sink);
        if ((it & 1023) == 1023) hipStreamSynchronize(s1);
      }
      hipStreamSynchronize(s1);
      stop.store(true);
      th.join();
      hipDeviceSynchronize();
      unsigned h = 0;
      hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
      printf("{\"probe\": \"W -> R hand-off through a 16 KiB buffer\", \"iterations\": %d, \"second_stream_busy\": %d, \"reader_loads\": \"%s\", "
             "\"mismatching_reads\": %u, \"second_stream_launches\": %lld}\n", iters, interfere, mode ? "system-scope" : "plain", h, side_launches);
      fflush(stdout);
    }
  const int big_iters = argc > 2 ? atoi(argv[2]) : 20000;
  big_handoff(big_iters, 20, s1, s2, junk);
  big_handoff(big_iters, 4, s1, s2, junk);
  return 0;
}
