// hipsim runtime: fiber scheduler + wave rendezvous (TEST-ONLY, see hipsim.h).
#include "hipsim.h"

#if !defined(__x86_64__)
#error "hipsim's context switch is written for x86-64"
#endif

namespace hipsim {

// save callee-saved registers on the current stack, store sp in *save, load sp, restore, return.
extern "C" void hipsim_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipsim_switch
.type hipsim_switch,@function
hipsim_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hipsim_switch,.-hipsim_switch
)");

static BlockState g_single;
static BlockState* g_cur = &g_single;      // the block whose fiber is running (launch_coresident switches it per fiber)
BlockState& blk() { return *g_cur; }
#define g_blk (*g_cur)

static const size_t kStack = 256 * 1024;
static std::vector<char*> g_stacks;

void yield() {
  BlockState& b = g_blk;
  hipsim_switch(&b.cur->sp, b.sched_sp);
}

static void release_barrier_if_complete(BlockState& b) {
  if (b.alive > 0 && b.bar_count == b.alive) {
    b.bar_count = 0;
    b.bar_gen++;
  }
}

void sync_threads() {
  BlockState& b = g_blk;
  unsigned my = b.bar_gen;
  b.bar_count++;
  release_barrier_if_complete(b);
  while (b.bar_gen == my) yield();
}

static void compute_mfma(WaveState& w, int slot) {
  // D[i][j] = fma(A[i][3],B[3][j], fma(A[i][2],B[2][j], fma(A[i][1],B[1][j], fma(A[i][0],B[0][j], C))))
  float A[16][4], B[4][16];
  for (int l = 0; l < 64; ++l) {
    const float* p = reinterpret_cast<const float*>(w.buf[slot][l]);
    A[l & 15][l >> 4] = p[0];
    B[l >> 4][l & 15] = p[1];
  }
  for (int l = 0; l < 64; ++l) {
    const float* p = reinterpret_cast<const float*>(w.buf[slot][l]);
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
      int row = (l >> 4) * 4 + r;
      float d = p[2 + r];
      for (int k = 0; k < 4; ++k) d = fmaf(A[row][k], B[k][col], d);
      w.res[slot][l][r] = d;
    }
  }
}

static float bf16_to_float(unsigned short h) {
  unsigned u = unsigned(h) << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

static float f16_to_float(unsigned short h) {
  _Float16 v;
  std::memcpy(&v, &h, 2);
  return (float)v;
}

template <float (*CVT)(unsigned short)>
static void compute_mfma_16bit(WaveState& w, int slot) {
  float A[16][32], B[32][16];
  for (int l = 0; l < 64; ++l) {
    const unsigned* p = reinterpret_cast<const unsigned*>(w.buf[slot][l]);
    for (int j = 0; j < 8; ++j) {
      const unsigned short ah = (p[j >> 1] >> (16 * (j & 1))) & 0xFFFFu, bh = (p[4 + (j >> 1)] >> (16 * (j & 1))) & 0xFFFFu;
      A[l & 15][(l >> 4) * 8 + j] = CVT(ah);
      B[(l >> 4) * 8 + j][l & 15] = CVT(bh);
    }
  }
  for (int l = 0; l < 64; ++l) {
    const float* cp = reinterpret_cast<const float*>(w.buf[slot][l]) + 8;
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
      const int row = (l >> 4) * 4 + r;
      float d = cp[r];
      for (int k = 0; k < 32; ++k) d = fmaf(A[row][k], B[k][col], d);   // products of bf16 / f16 values are exact in fp32 (f16 subnormal x subnormal aside: below 4e-15)
      w.res[slot][l][r] = d;
    }
  }
}

static void compute_mfma_fp8(WaveState& w, int slot) {
  float A[16][32], B[32][16];
  for (int l = 0; l < 64; ++l) {
    const unsigned* p = reinterpret_cast<const unsigned*>(w.buf[slot][l]);
    for (int j = 0; j < 8; ++j) {
      A[l & 15][(l >> 4) * 8 + j] = fp8_e4m3_value((p[j >> 2] >> (8 * (j & 3))) & 0xFFu);
      B[(l >> 4) * 8 + j][l & 15] = fp8_e4m3_value((p[4 + (j >> 2)] >> (8 * (j & 3))) & 0xFFu);
    }
  }
  for (int l = 0; l < 64; ++l) {
    const float* cp = reinterpret_cast<const float*>(w.buf[slot][l]) + 8;
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
      const int row = (l >> 4) * 4 + r;
      float d = cp[r];
      for (int k = 0; k < 32; ++k) d = fmaf(A[row][k], B[k][col], d);   // products of e4m3 values are exact in fp32
      w.res[slot][l][r] = d;
    }
  }
}

int wave_arrive(const void* payload, int nbytes, int is_mfma) {
  BlockState& b = g_blk;
  WaveState& w = b.waves[b.cur->wave];
  unsigned my = w.gen;
  int slot = my & 1;
  std::memcpy(w.buf[slot][b.cur->lane], payload, nbytes);
  w.count++;
  if (w.count == w.alive) {
    if (is_mfma) {
      if (w.alive != 64) { std::fprintf(stderr, "hipsim: MFMA issued by a partial wave\n"); std::abort(); }
      if (is_mfma == 1) compute_mfma(w, slot);
      else if (is_mfma == 2) compute_mfma_16bit<bf16_to_float>(w, slot);
      else if (is_mfma == 4) compute_mfma_16bit<f16_to_float>(w, slot);
      else compute_mfma_fp8(w, slot);
    }
    w.count = 0;
    w.gen++;
  } else {
    while (w.gen == my) yield();
  }
  return slot;
}

static void fiber_main() {
  BlockState& b = g_blk;
  b.body();
  Fiber* f = b.cur;
  f->done = true;
  b.alive--;
  WaveState& w = b.waves[f->wave];
  w.alive--;
  if (w.alive > 0 && w.count == w.alive) {   // a lane exited while its wave waits: undefined on HW too
    std::fprintf(stderr, "hipsim: lane exited while its wave was inside a wave-level op\n");
    std::abort();
  }
  release_barrier_if_complete(b);
  hipsim_switch(&f->sp, b.sched_sp);
  std::abort();  // never resumed
}

static void init_fiber(Fiber& f) {
  uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + kStack) & ~uintptr_t(15);
  void** sp = reinterpret_cast<void**>(top);
  *--sp = nullptr;                                   // fake return address of fiber_main
  *--sp = reinterpret_cast<void*>(&fiber_main);      // popped by `ret` in hipsim_switch
  for (int i = 0; i < 6; ++i) *--sp = nullptr;       // rbp rbx r12-r15
  f.sp = sp;
  f.done = false;
}

void launch(dim3 grid, dim3 block, size_t shmem, std::function<void()> body) {
  g_cur = &g_single;
  BlockState& b = g_blk;
  int n = int(block.x * block.y * block.z);
  if (n <= 0 || n > 1024) { std::fprintf(stderr, "hipsim: bad block size %d\n", n); std::abort(); }
  while (int(g_stacks.size()) < n) g_stacks.push_back(static_cast<char*>(std::malloc(kStack)));
  b.nthreads = n;
  b.bdim = block;
  b.gdim = grid;
  b.body = std::move(body);
  b.fibers.assign(n, Fiber());
  int nw = (n + 63) / 64;
  b.dyn_smem.assign(shmem + 16, 0);
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        b.bid = uint3_{bx, by, bz};
        b.alive = n;
        b.bar_count = 0;
        b.bar_gen = 0;
        b.waves.assign(nw, WaveState());
        for (int t = 0; t < n; ++t) {
          Fiber& f = b.fibers[t];
          f.stack = g_stacks[t];
          f.tid = uint3_{unsigned(t) % block.x, (unsigned(t) / block.x) % block.y, unsigned(t) / (block.x * block.y)};
          f.lane = t & 63;
          f.wave = t >> 6;
          b.waves[f.wave].alive++;
          init_fiber(f);
        }
        // alternate the sweep direction between passes so a missing barrier is more likely to
        // surface whichever side of the race the bug is on.
        bool forward = true;
        unsigned long spins = 0;
        while (b.alive > 0) {
          for (int k = 0; k < n; ++k) {
            int t = forward ? k : n - 1 - k;
            Fiber& f = b.fibers[t];
            if (f.done) continue;
            b.cur = &f;
            hipsim_switch(&b.sched_sp, f.sp);
          }
          forward = !forward;
          if (++spins > 400000000ul) { std::fprintf(stderr, "hipsim: deadlock (barrier/wave-op mismatch)\n"); std::abort(); }
        }
      }
  b.cur = nullptr;
}

// Every block of the grid alive at once (persistent kernels whose workgroups hand data to each other and spin on flags: kernels/loop_cluster.hpp).
// One fiber per work-item of EVERY block; the sweep visits the blocks in turn.  A spinning fiber must call hipsim::yield() (rt.hpp spin_pause).
// Kernels launched this way may not use static __shared__ (one host variable for all blocks): dynamic shared memory only.
void launch_coresident(dim3 grid, dim3 block, size_t shmem, std::function<void()> body) {
  const int n = int(block.x * block.y * block.z), nb = int(grid.x * grid.y * grid.z);
  if (n <= 0 || n > 1024 || nb <= 0 || nb > 64) { std::fprintf(stderr, "hipsim: bad co-resident launch (%d blocks of %d)\n", nb, n); std::abort(); }
  static std::vector<char*> stacks;
  while (stacks.size() < (size_t)n * nb) stacks.push_back(static_cast<char*>(std::malloc(kStack)));
  std::vector<BlockState> bs(nb);
  const int nw = (n + 63) / 64;
  for (int k = 0; k < nb; ++k) {
    BlockState& b = bs[k];
    b.nthreads = n; b.bdim = block; b.gdim = grid; b.body = body;
    b.fibers.assign(n, Fiber());
    b.dyn_smem.assign(shmem + 16, 0);
    b.bid = uint3_{unsigned(k) % grid.x, (unsigned(k) / grid.x) % grid.y, unsigned(k) / (grid.x * grid.y)};
    b.alive = n; b.bar_count = 0; b.bar_gen = 0;
    b.waves.assign(nw, WaveState());
    for (int t = 0; t < n; ++t) {
      Fiber& f = b.fibers[t];
      f.stack = stacks[(size_t)k * n + t];
      f.tid = uint3_{unsigned(t) % block.x, (unsigned(t) / block.x) % block.y, unsigned(t) / (block.x * block.y)};
      f.lane = t & 63;
      f.wave = t >> 6;
      b.waves[f.wave].alive++;
      init_fiber(f);
    }
  }
  unsigned long spins = 0;
  bool forward = true;
  for (;;) {
    int alive = 0;
    for (int kk = 0; kk < nb; ++kk) {
      BlockState& b = bs[forward ? kk : nb - 1 - kk];
      if (b.alive <= 0) continue;
      alive += b.alive;
      for (int q = 0; q < n; ++q) {
        Fiber& f = b.fibers[forward ? q : n - 1 - q];
        if (f.done) continue;
        g_cur = &b;
        b.cur = &f;
        hipsim_switch(&b.sched_sp, f.sp);
      }
    }
    if (!alive) break;
    forward = !forward;
    if (++spins > 400000000ul) { std::fprintf(stderr, "hipsim: deadlock in a co-resident launch\n"); std::abort(); }
  }
  g_cur = &g_single;
}

}  // namespace hipsim
