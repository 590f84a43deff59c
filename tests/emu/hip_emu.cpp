// hip_emu.cpp -- TEST INFRASTRUCTURE ONLY (see hip_emu.h).
// Cooperative fibers: every logical GPU thread of one workgroup is a ucontext
// fiber; barriers and wave collectives yield to a round-robin scheduler.
#include "hip_emu.h"
#include <stdio.h>

namespace ea_emu {
Dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
char* g_smem = nullptr;

namespace {
struct Fiber {
  ucontext_t uc;
  char* stack = nullptr;
  bool done = false;
  int tid = 0;
};
constexpr size_t kStack = 256 * 1024;
constexpr size_t kSmemBytes = 160 * 1024;
ucontext_t g_main;
std::vector<Fiber> g_fibers;
int g_cur = -1;
int g_nthreads = 0;
const std::function<void()>* g_body = nullptr;

int g_bar_count = 0;
unsigned g_bar_gen = 0;
struct WaveState {
  int count = 0;
  unsigned gen = 0;
  char scratch[64 * 64];
};
std::vector<WaveState> g_waves;
char* g_smem_raw = nullptr;

void set_tid(int tid) {
  g_threadIdx.x = tid % g_blockDim.x;
  g_threadIdx.y = (tid / g_blockDim.x) % g_blockDim.y;
  g_threadIdx.z = tid / (g_blockDim.x * g_blockDim.y);
}

void trampoline() {
  (*g_body)();
  g_fibers[g_cur].done = true;
  swapcontext(&g_fibers[g_cur].uc, &g_main);
}
}  // namespace

void yield_() { swapcontext(&g_fibers[g_cur].uc, &g_main); }

int lane_id() { return g_cur & 63; }
int wave_lanes() {
  int w = g_cur >> 6;
  int rem = g_nthreads - w * 64;
  return rem > 64 ? 64 : rem;
}
char* wave_scratch() { return g_waves[g_cur >> 6].scratch; }

void block_sync() {
  unsigned gen = g_bar_gen;
  if (++g_bar_count == g_nthreads) {
    g_bar_count = 0;
    g_bar_gen++;
  } else {
    while (g_bar_gen == gen) yield_();
  }
}

void wave_sync() {
  WaveState& w = g_waves[g_cur >> 6];
  unsigned gen = w.gen;
  if (++w.count == wave_lanes()) {
    w.count = 0;
    w.gen++;
  } else {
    while (w.gen == gen) yield_();
  }
}

void launch(Dim3 grid, Dim3 block, size_t smem, const std::function<void()>& body) {
  if (!g_smem_raw) {
    g_smem_raw = (char*)aligned_alloc(256, kSmemBytes);
  }
  if (smem > kSmemBytes) {
    fprintf(stderr, "ea_emu: smem request %zu too large\n", smem);
    abort();
  }
  g_smem = g_smem_raw;
  g_blockDim = block;
  g_gridDim = grid;
  g_nthreads = block.x * block.y * block.z;
  g_body = &body;
  if ((int)g_fibers.size() < g_nthreads) {
    size_t old = g_fibers.size();
    g_fibers.resize(g_nthreads);
    for (size_t i = old; i < g_fibers.size(); ++i) g_fibers[i].stack = (char*)malloc(kStack);
  }
  g_waves.assign((g_nthreads + 63) / 64, WaveState());
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_blockIdx = Dim3(bx, by, bz);
        // poison LDS so that reads of unwritten LDS show up as NaNs/garbage
        memset(g_smem_raw, 0x7f, smem ? smem : 16);
        g_bar_count = 0;
        for (auto& w : g_waves) w.count = 0;
        for (int t = 0; t < g_nthreads; ++t) {
          Fiber& f = g_fibers[t];
          f.done = false;
          f.tid = t;
          getcontext(&f.uc);
          f.uc.uc_stack.ss_sp = f.stack;
          f.uc.uc_stack.ss_size = kStack;
          f.uc.uc_link = &g_main;
          makecontext(&f.uc, (void (*)())trampoline, 0);
        }
        int remaining = g_nthreads;
        while (remaining > 0) {
          remaining = 0;
          for (int t = 0; t < g_nthreads; ++t) {
            if (g_fibers[t].done) continue;
            g_cur = t;
            set_tid(t);
            swapcontext(&g_main, &g_fibers[t].uc);
            if (!g_fibers[t].done) remaining++;
          }
        }
      }
  g_cur = -1;
}
}  // namespace ea_emu
