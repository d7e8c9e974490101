// TEST INFRASTRUCTURE ONLY — see sim_runtime.h.
#include "sim_runtime.h"

#include <stdio.h>

#include <vector>

extern "C" void sim_switch(void **save_sp, void *new_sp);
asm(".text\n"
    ".globl sim_switch\n"
    ".type sim_switch,@function\n"
    "sim_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n");

namespace sim {

constexpr int kMaxThreads = 1024;
constexpr size_t kStack = 256 * 1024;

struct Wave {
  int alive = 0, arrived = 0;
  uint64_t gen = 0;
  uint64_t in[2][64];
  uint8_t pred[2][64];
  uint8_t present[2][64];
  uint64_t result[2];
};
struct Fiber {
  void *sp = nullptr;
  char *stack = nullptr;
  int tid = 0;
  bool done = true;
  const uint64_t *wait_ptr = nullptr;
  uint64_t wait_val = 0;
};

Fiber *cur = nullptr;
static Fiber fibers[kMaxThreads];
static Wave waves[kMaxThreads / 64];
static void *sched_sp = nullptr;
static int g_block = 0, g_block_dim = 0, g_grid_dim = 0;
static int blk_alive = 0, blk_arrived = 0;
static uint64_t blk_gen = 0;
static const std::function<void()> *g_body = nullptr;
enum { OP_BALLOT, OP_SHFL, OP_SUM };
static int wave_op[kMaxThreads / 64][2];

int cur_tid() { return cur->tid; }
int cur_block() { return g_block; }
int cur_block_dim() { return g_block_dim; }
int cur_grid_dim() { return g_grid_dim; }

static void yield_to_sched() { sim_switch(&cur->sp, sched_sp); }

static void complete_wave(Wave &w, int widx) {
  int par = (int)(w.gen & 1);
  uint64_t r = 0;
  if (wave_op[widx][par] == OP_BALLOT) {
    for (int l = 0; l < 64; l++)
      if (w.present[par][l] && w.pred[par][l]) r |= (1ull << l);
  } else if (wave_op[widx][par] == OP_SUM) {
    for (int l = 0; l < 64; l++)
      if (w.present[par][l]) r += w.in[par][l];
  }
  w.result[par] = r;
  w.arrived = 0;
  w.gen++;
  int npar = (int)(w.gen & 1);
  memset(w.present[npar], 0, 64);
}

static int arrive(int op, uint64_t v, bool p) {
  int widx = cur->tid >> 6, l = cur->tid & 63;
  Wave &w = waves[widx];
  int par = (int)(w.gen & 1);
  if (w.arrived == 0) wave_op[widx][par] = op;
  else if (wave_op[widx][par] != op) {
    fprintf(stderr, "sim: lanes of wave %d disagree on the collective (divergent control flow)\n", widx);
    abort();
  }
  w.in[par][l] = v;
  w.pred[par][l] = p;
  w.present[par][l] = 1;
  w.arrived++;
  if (w.arrived == w.alive) {
    complete_wave(w, widx);
  } else {
    cur->wait_ptr = &w.gen;
    cur->wait_val = w.gen;
    yield_to_sched();
  }
  return par;
}

uint64_t ballot(bool p) {
  int par = arrive(OP_BALLOT, 0, p);
  return waves[cur->tid >> 6].result[par];
}
uint64_t shfl64(uint64_t v, int src) {
  int par = arrive(OP_SHFL, v, false);
  Wave &w = waves[cur->tid >> 6];
  src &= 63;
  return w.present[par][src] ? w.in[par][src] : 0;  // inactive source lane: undefined on HW, 0 here
}
bool check_uniform = false;
uint64_t first64(uint64_t v) {
  int par = arrive(OP_SHFL, v, false);
  Wave &w = waves[cur->tid >> 6];
  for (int l = 0; l < 64; l++)
    if (w.present[par][l]) return w.in[par][l];
  return v;
}
uint64_t uniform64(uint64_t v) {
  if (!check_uniform) return v;  // (the claim is trusted: no rendezvous — the checked runs are the tests that switch it on)
  int par = arrive(OP_SHFL, v, false);
  Wave &w = waves[cur->tid >> 6];
  for (int l = 0; l < 64; l++)
    if (w.present[par][l] && w.in[par][l] != v) {
      fprintf(stderr, "sim: a value declared wave-uniform differs between lanes (%d: %llu, %d: %llu)\n", cur->tid & 63, (unsigned long long)v, l,
              (unsigned long long)w.in[par][l]);
      abort();
    }
  return v;
}
void gather64(uint64_t v, uint64_t *all64) {
  int par = arrive(OP_SHFL, v, false);
  Wave &w = waves[cur->tid >> 6];
  for (int l = 0; l < 64; l++) all64[l] = w.present[par][l] ? w.in[par][l] : 0;
}
uint64_t reduce_add64(uint64_t v) {
  int par = arrive(OP_SUM, v, false);
  return waves[cur->tid >> 6].result[par];
}
void block_sync() {
  blk_arrived++;
  if (blk_arrived == blk_alive) {
    blk_arrived = 0;
    blk_gen++;
  } else {
    cur->wait_ptr = &blk_gen;
    cur->wait_val = blk_gen;
    yield_to_sched();
  }
}

static void fiber_exit() {
  Wave &w = waves[cur->tid >> 6];
  cur->done = true;
  w.alive--;
  if (w.alive > 0 && w.arrived == w.alive) complete_wave(w, cur->tid >> 6);
  blk_alive--;
  if (blk_alive > 0 && blk_arrived == blk_alive) {
    blk_arrived = 0;
    blk_gen++;
  }
  yield_to_sched();
  abort();  // never resumed
}
extern "C" void sim_fiber_entry() {
  (*g_body)();
  fiber_exit();
}

void launch(uint32_t grid, uint32_t block, const std::function<void()> &body) {
  if (block > kMaxThreads || (block & 63)) {
    fprintf(stderr, "sim: block size %u unsupported\n", block);
    abort();
  }
  g_body = &body;
  g_block_dim = (int)block;
  g_grid_dim = (int)grid;
  for (uint32_t b = 0; b < grid; b++) {
    g_block = (int)b;
    blk_alive = (int)block;
    blk_arrived = 0;
    for (uint32_t w = 0; w < block / 64; w++) {
      waves[w] = Wave();
      waves[w].alive = 64;
      memset(waves[w].present, 0, sizeof(waves[w].present));
    }
    for (uint32_t t = 0; t < block; t++) {
      Fiber &f = fibers[t];
      if (!f.stack) f.stack = (char *)aligned_alloc(64, kStack);
      f.tid = (int)t;
      f.done = false;
      f.wait_ptr = nullptr;
      uintptr_t top = ((uintptr_t)(f.stack + kStack)) & ~(uintptr_t)15;
      void **sp = (void **)(top - 64);  // 16-aligned
      sp[0] = sp[1] = sp[2] = sp[3] = sp[4] = sp[5] = nullptr;  // r15 r14 r13 r12 rbx rbp
      sp[6] = (void *)&sim_fiber_entry;
      sp[7] = nullptr;
      f.sp = (void *)sp;
    }
    int remaining = (int)block;
    while (remaining > 0) {
      bool progress = false;
      for (uint32_t t = 0; t < block; t++) {
        Fiber &f = fibers[t];
        if (f.done) continue;
        if (f.wait_ptr && *f.wait_ptr == f.wait_val) continue;
        f.wait_ptr = nullptr;
        cur = &f;
        sim_switch(&sched_sp, f.sp);
        progress = true;
        if (f.done) remaining--;
      }
      if (!progress) {
        fprintf(stderr, "sim: deadlock in block %u (a collective was not reached by every live lane)\n", b);
        abort();
      }
    }
  }
  cur = nullptr;
}

}  // namespace sim
