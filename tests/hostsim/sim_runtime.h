// TEST INFRASTRUCTURE ONLY.  A single-threaded fiber SIMT emulator: every GPU thread is a fiber,
// a workgroup's fibers are scheduled round-robin and rendezvous at wave collectives (ballot / shfl /
// reduce) and at block barriers.  It exists so the *actual kernel source* of the engine can be debugged
// on a machine without a GPU.  The shipped library (libppcsr_hip.so) never contains this code; the
// sim build is a separate .so that only tests/ load explicitly.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

#define PMA_DEV inline
#define PMA_KERNEL
#define PMA_LAUNCH_BOUNDS(threads, waves_per_simd)
#define PMA_SHARED static

struct uint4 {
  uint32_t x, y, z, w;
};
namespace sim {
struct Fiber;
extern Fiber *cur;
int cur_tid();
int cur_block();
int cur_block_dim();
int cur_grid_dim();
uint64_t ballot(bool p);
uint64_t shfl64(uint64_t v, int src);
uint64_t reduce_add64(uint64_t v);
uint64_t first64(uint64_t v);    // value of the first live lane
uint64_t uniform64(uint64_t v);  // aborts unless every live lane passes the same value
void gather64(uint64_t v, uint64_t *all64);  // every lane's value (0 for lanes that are not live)
extern bool check_uniform;
void block_sync();
void launch(uint32_t grid, uint32_t block, const std::function<void()> &body);
}  // namespace sim

namespace ppcsr {
namespace wv {
inline int lane() { return sim::cur_tid() & 63; }
inline int wave_in_block() { return sim::cur_tid() >> 6; }
inline uint64_t ballot(bool p) { return sim::ballot(p); }
inline uint32_t shfl(uint32_t v, int src) { return (uint32_t)sim::shfl64(v, src); }
inline float shfl_f32(float v, int src) {
  uint32_t b;
  memcpy(&b, &v, 4);
  b = (uint32_t)sim::shfl64(b, src);
  memcpy(&v, &b, 4);
  return v;
}
inline uint32_t first(uint32_t v) { return (uint32_t)sim::first64(v); }  // (the first LIVE lane's value)
// uni / bcast: the caller claims the value (the source lane) is the same in every live lane of the wave — checked here
inline uint32_t uni(uint32_t v) { return (uint32_t)sim::uniform64(v); }
inline int uni(int v) { return (int)(uint32_t)sim::uniform64((uint32_t)v); }
inline bool uni(bool b) { return sim::uniform64(b ? 1u : 0u) != 0; }
inline uint64_t uni(uint64_t v) { return sim::uniform64(v); }
template <int K>
inline uint32_t setlane(uint32_t v, uint32_t x) { return lane() == K ? x : v; }
template <int N>
inline void lanes(uint32_t v, uint32_t *out) {
  uint64_t all[64];
  sim::gather64(v, all);
  for (int i = 0; i < N; i++) out[i] = (uint32_t)all[i];
}
inline uint32_t bcast(uint32_t v, int src) { return (uint32_t)sim::shfl64(v, (int)(uint32_t)sim::uniform64((uint32_t)src)); }
inline uint32_t reduce_add(uint32_t v) { return (uint32_t)sim::reduce_add64(v); }
inline void fence() { (void)sim::shfl64(0, 0); }  // lanes run sequentially between collectives: a fence must be a rendezvous
inline void lds_fence() { (void)sim::shfl64(0, 0); }
inline void block_sync() { sim::block_sync(); }
inline uint32_t atomic_min_u32(uint32_t *p, uint32_t v) { uint32_t o = *p; if (v < o) *p = v; return o; }
inline unsigned long long atomic_min_u64(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; if (v < o) *p = v; return o; }
inline uint32_t atomic_add_u32(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
inline void atomic_add_nn(uint32_t *p, uint32_t v) { *p += v; }
inline unsigned long long atomic_add_u64(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
inline uint32_t atomic_max_u32(uint32_t *p, uint32_t v) { uint32_t o = *p; if (v > o) *p = v; return o; }
inline unsigned long long atomic_max_u64(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; if (v > o) *p = v; return o; }
inline uint32_t atomic_exch_u32(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = v; return o; }
inline uint32_t atomic_cas_u32(uint32_t *p, uint32_t expect, uint32_t v) { uint32_t o = *p; if (o == expect) *p = v; return o; }
inline void wait_loads() {}
inline void flag_publish(uint32_t *p, uint32_t v) { *p = v; }
inline uint32_t flag_read(const uint32_t *p) { return *(const volatile uint32_t *)p; }
inline void flag_acquire() {}
inline void store_agent_u64(unsigned long long *p, unsigned long long v) { *p = v; }
inline unsigned long long load_agent_u64(const unsigned long long *p) { return *(const volatile unsigned long long *)p; }
inline uint32_t xcc_id() { return (uint32_t)sim::cur_block() & 7u; }
inline uint32_t load_stream_u32(const uint32_t *p) { return *p; }
inline void spin_pause() {}
inline int clz64(uint64_t m) { return m ? __builtin_clzll(m) : 64; }
inline int popc64(uint64_t m) { return __builtin_popcountll(m); }
inline int ctz64(uint64_t m) { return m ? __builtin_ctzll(m) : -1; }
inline uint32_t block_idx() { return (uint32_t)sim::cur_block(); }
inline uint32_t thread_idx() { return (uint32_t)sim::cur_tid(); }
inline uint32_t block_dim() { return (uint32_t)sim::cur_block_dim(); }
inline uint32_t grid_dim() { return (uint32_t)sim::cur_grid_dim(); }
}  // namespace wv
}  // namespace ppcsr
