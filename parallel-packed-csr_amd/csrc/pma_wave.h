// Wavefront primitives used by the PMA kernels.  Product build: gfx950 intrinsics (wave64).
// When PPCSR_SIM is defined (tests/hostsim only — never in the shipped library) the same names are
// provided by a single-threaded fiber SIMT emulator so the kernel source can be debugged on CPU.
#pragma once
#include "pma_types.h"

#if defined(PPCSR_SIM)
#include "sim_runtime.h"
#else
#include <hip/hip_runtime.h>
#define PMA_DEV __device__ __forceinline__
#define PMA_KERNEL __global__
#define PMA_LAUNCH_BOUNDS(threads, waves_per_simd) __launch_bounds__(threads, waves_per_simd)
#define PMA_SHARED __shared__

namespace ppcsr {
namespace wv {
PMA_DEV int lane() { return (int)(threadIdx.x & 63u); }
PMA_DEV int wave_in_block() { return (int)(threadIdx.x >> 6); }
PMA_DEV uint64_t ballot(bool p) { return (uint64_t)__ballot(p ? 1 : 0); }
PMA_DEV uint32_t shfl(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }
PMA_DEV float shfl_f32(float v, int src) { return __shfl(v, src, 64); }
PMA_DEV uint32_t first(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// Scalar values.  One wave executes one update and every decision about it is the same in all 64 lanes — but the compiler only
// knows that of values it can prove uniform (kernel arguments, ballots): anything that descends from `threadIdx.x >> 6` or comes
// out of a shuffle looks per-lane to it, lives in a vector register and branches through EXEC-mask arithmetic.  uni() states
// that a value is the same in every lane (v_readfirstlane: the result is a scalar register); bcast() reads lane `src` (`src` itself
// uniform: v_readlane, a few cycles, where a shuffle is an LDS round trip).  The CPU emulator checks both claims on every call.
PMA_DEV uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
PMA_DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
PMA_DEV bool uni(bool b) { return __builtin_amdgcn_readfirstlane(b ? 1 : 0) != 0; }
PMA_DEV uint64_t uni(uint64_t v) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
// v with lane K replaced by the scalar `x` (v_writelane_b32; this compiler has no builtin for it)
template <int K>
PMA_DEV uint32_t setlane(uint32_t v, uint32_t x) {
  const uint32_t sx = (uint32_t)__builtin_amdgcn_readfirstlane((int)x);  // (the "s" constraint alone does not move a vector value)
  asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(sx), "n"(K));
  return v;
}
// lanes 0 .. N-1 of v as scalars (N x v_readlane_b32; the emulator does it in one rendezvous)
template <int N>
PMA_DEV void lanes(uint32_t v, uint32_t *out) {
#pragma unroll
  for (int i = 0; i < N; i++) out[i] = (uint32_t)__builtin_amdgcn_readlane((int)v, i);
}
PMA_DEV uint32_t bcast(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(src)); }
// sum over the wave (all 64 lanes active), as a scalar.  DPP adds: an inclusive scan inside each row of 16 lanes (row_shr 1, 2,
// 4, 8), the row totals passed on (row_bcast 15 into rows 1 and 3, row_bcast 31 into rows 2 and 3), the total read off lane 63 —
// six VALU instructions and a v_readlane where six shuffles were six LDS round trips.
PMA_DEV uint32_t reduce_add(uint32_t v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
  return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}
// orders this wave's LDS + global accesses among its own lanes (same CU: L1 is shared)
PMA_DEV void fence() { __threadfence_block(); }
// orders this wave's LDS accesses only (one wave's DS operations execute in order; this is a compiler barrier that
// does not wait for outstanding global loads / stores / atomics)
PMA_DEV void lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
PMA_DEV void block_sync() { __syncthreads(); }
PMA_DEV uint32_t atomic_min_u32(uint32_t *p, uint32_t v) { return atomicMin(p, v); }
PMA_DEV unsigned long long atomic_min_u64(unsigned long long *p, unsigned long long v) { return atomicMin(p, v); }
PMA_DEV uint32_t atomic_add_u32(uint32_t *p, uint32_t v) { return atomicAdd(p, v); }
// num_neighbors: the one word of nodes[] that device-wide atomics change
PMA_DEV void atomic_add_nn(uint32_t *p, uint32_t v) { (void)atomicAdd(p, v); }
PMA_DEV unsigned long long atomic_add_u64(unsigned long long *p, unsigned long long v) { return atomicAdd(p, v); }
PMA_DEV uint32_t atomic_max_u32(uint32_t *p, uint32_t v) { return atomicMax(p, v); }
PMA_DEV unsigned long long atomic_max_u64(unsigned long long *p, unsigned long long v) { return atomicMax(p, v); }
PMA_DEV uint32_t atomic_exch_u32(uint32_t *p, uint32_t v) { return atomicExch(p, v); }
PMA_DEV uint32_t atomic_cas_u32(uint32_t *p, uint32_t expect, uint32_t v) { return atomicCAS(p, expect, v); }
// hand-off flags between workgroups (k_rb_inplace): "my loads have RETURNED" — published after a wait for them, polled with
// agent-scope loads (a plain load may be served from a stale line of this XCD's L2 for ever).  No data travels with the
// flag (the poller only WRITES afterwards), so neither side needs a release / acquire: at agent scope those are an L2
// write-back and a cache invalidate per workgroup, which made the kernel 4x slower than the copy it replaces.
// every global load this wave has issued has returned (a workgroup barrier alone does not wait for vmcnt)
PMA_DEV void wait_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
PMA_DEV void flag_publish(uint32_t *p, uint32_t v) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
PMA_DEV uint32_t flag_read(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PMA_DEV void flag_acquire() { asm volatile("" ::: "memory"); }
// data that travels between workgroups INSIDE a launch (the position table of k_rb_scatter): written and read at agent scope,
// word by word — a plain store may sit in the writer's XCD L2 and a plain load may be served from a stale line of the
// reader's, and a release / acquire pair at agent scope is an L2 write-back / invalidate per use
PMA_DEV void store_agent_u64(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PMA_DEV unsigned long long load_agent_u64(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// which of the 8 XCDs this wave runs on (HW_REG_XCC_ID bits 3:0) — used for affinity only, never for correctness
PMA_DEV uint32_t xcc_id() { return (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; }
// a load of data that is read ONCE by the whole launch (the source of a rebalance pass): the non-temporal hint keeps it from
// displacing the lines the pass is still assembling in L2 (whole-array rebalance at 2^24 slots 115 -> 105 us with it on the
// source reads; the neighbour scan, which writes little, got slower with it: 60 -> 65 us — plain loads there)
PMA_DEV uint32_t load_stream_u32(const uint32_t *p) { return __builtin_nontemporal_load(p); }
PMA_DEV void spin_pause() { __builtin_amdgcn_s_sleep(2); }
PMA_DEV int clz64(uint64_t m) { return __clzll((long long)m); }
PMA_DEV int popc64(uint64_t m) { return __popcll((unsigned long long)m); }
PMA_DEV int ctz64(uint64_t m) { return __ffsll((long long)m) - 1; }
PMA_DEV uint32_t block_idx() { return blockIdx.x; }
PMA_DEV uint32_t thread_idx() { return threadIdx.x; }
PMA_DEV uint32_t block_dim() { return blockDim.x; }
PMA_DEV uint32_t grid_dim() { return gridDim.x; }
}  // namespace wv
}  // namespace ppcsr
#endif
