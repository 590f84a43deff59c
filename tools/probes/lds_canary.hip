// LDS canary (round 4 race hunt): workgroups that fill their whole LDS allocation with a known pattern, idle for a while, and then
// check it.  Anything that writes into LDS it does not own -- a late LDS-DMA of a workgroup that already left the CU, an out-of-range
// DMA destination of a co-resident workgroup -- shows up as a changed word, with its offset and value.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/probes/liblds_canary.so tools/probes/lds_canary.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

struct CanaryOut { unsigned bad_words, first_off, first_val, first_wg, wgs_hit, max_off, pad0, pad1; };

__global__ __launch_bounds__(256) void lds_canary_kernel(CanaryOut* out, int lds_words, int spin, unsigned salt) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < lds_words; i += 256) lds[i] = 0xC0DE0000u ^ (unsigned)i ^ salt;
  __syncthreads();
  for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(64);
  __syncthreads();
  unsigned mine = 0;
  for (int i = threadIdx.x; i < lds_words; i += 256) {
    const unsigned v = lds[i];
    if (v != (0xC0DE0000u ^ (unsigned)i ^ salt)) {
      ++mine;
      if (atomicAdd(&out->bad_words, 1u) == 0) { out->first_off = 4u * i; out->first_val = v; out->first_wg = blockIdx.x; }
      atomicMax(&out->max_off, 4u * i);
    }
  }
  if (mine) atomicAdd(&out->wgs_hit, 1u);    // (counts threads with a hit, an upper bound on workgroups)
}

extern "C" int lds_canary(void* out, int wgs, int lds_bytes, int spin, unsigned salt, void* stream) {
  (void)hipFuncSetAttribute((const void*)lds_canary_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  lds_canary_kernel<<<dim3(wgs), dim3(256), lds_bytes, (hipStream_t)stream>>>((CanaryOut*)out, lds_bytes / 4, spin, salt);
  return (int)hipGetLastError();
}
