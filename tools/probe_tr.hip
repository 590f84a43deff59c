// probe_tr.hip -- hardware probe (not product code): the lane/element mapping of ds_read_b64_tr_b16 on gfx950.
// LDS holds u16 value = element index; every lane reads 8 bytes at its own address; prints what each lane got.
//   hipcc --offload-arch=gfx950 tools/probe_tr.hip -o /tmp/probe_tr && /tmp/probe_tr
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const int* lane_addr, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const unsigned addr = (unsigned)(uintptr_t)lds + lane_addr[threadIdx.x];   // LDS byte address
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}

static void run(const char* name, int (*f)(int)) {
  int h_addr[64];
  for (int l = 0; l < 64; ++l) h_addr[l] = f(l);
  int* d_addr;
  uint16_t* d_out;
  uint16_t h_out[256];
  hipMalloc(&d_addr, sizeof(h_addr));
  hipMalloc(&d_out, sizeof(h_out));
  hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d_addr, d_out);
  hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  printf("== %s\n", name);
  for (int l = 0; l < 64; ++l)
    printf("lane %2d addr(elem) %4d -> %4d %4d %4d %4d\n", l, h_addr[l] / 2, h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2],
           h_out[l * 4 + 3]);
  hipFree(d_addr);
  hipFree(d_out);
}

// A: lane l reads at element 4*l (contiguous 8-byte pieces): reveals which lanes' data a lane receives
static int addr_linear(int l) { return l * 8; }
// B: a [rows][16] row-major matrix per 16-lane group: lane i of group g supplies row (i >> 2), cols 4*(i & 3) of block g
static int addr_block(int l) { const int g = l >> 4, i = l & 15; return ((g * 4 + (i >> 2)) * 16 + 4 * (i & 3)) * 2; }
// C: row stride 64 elements (a [4 x 64] tile, group g covers cols 16g..16g+15)
static int addr_wide(int l) { const int g = l >> 4, i = l & 15; return ((i >> 2) * 64 + g * 16 + 4 * (i & 3)) * 2; }

int main() {
  run("linear: lane l -> elements 4l..4l+3", addr_linear);
  run("block: group g = 4x16 block g (row-major, 16 cols)", addr_block);
  run("wide: 4 rows x 64 cols, group g = cols 16g..", addr_wide);
  return 0;
}
