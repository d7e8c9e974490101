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
#if defined(PPCSR_CHAIN_INLINE)
#define PMA_DEV_CALL __device__ __forceinline__
#else
#define PMA_DEV_CALL __device__ __noinline__
#endif
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
PMA_DEV uint32_t reduce_add(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
  return v;
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
PMA_DEV void atomic_add_nn(uint32_t *p, uint32_t v) {
#if defined(PPCSR_NN_RETURNING)
  const uint32_t old = atomicAdd(p, v);
  asm volatile("" ::"v"(old));  // (the returning form of the instruction)
#else
  (void)atomicAdd(p, v);
#endif
}
PMA_DEV unsigned long long atomic_add_u64(unsigned long long *p, unsigned long long v) { return atomicAdd(p, v); }
PMA_DEV uint32_t atomic_max_u32(uint32_t *p, uint32_t v) { return atomicMax(p, v); }
PMA_DEV unsigned long long atomic_max_u64(unsigned long long *p, unsigned long long v) { return atomicMax(p, v); }
PMA_DEV uint32_t atomic_exch_u32(uint32_t *p, uint32_t v) { return atomicExch(p, v); }
PMA_DEV uint32_t atomic_cas_u32(uint32_t *p, uint32_t expect, uint32_t v) { return atomicCAS(p, expect, v); }
// hand-off flags between workgroups (k_rb_inplace): "my loads have RETURNED" — published after a wait for them, polled with
// agent-scope loads (a plain load may be served from a stale line of this XCD's L2 for ever).  No data travels with the
// flag (the poller only WRITES afterwards), so neither side needs a release / acquire: at agent scope those are an L2
// write-back and a cache invalidate per workgroup, which made the kernel 4x slower than the copy it replaces.
// everything this wave has written is visible to its own later loads whatever path they take (vector L1, scalar cache) and to
// the other XCDs' L2s; what it has cached is dropped.  For the few places where ONE kernel reads back what it wrote through
// different lanes / instructions over many steps (the in-round chains)
PMA_DEV void fence_heavy() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
  __builtin_amdgcn_s_dcache_inv();
}
PMA_DEV void fence_mode(uint32_t mode) {  // (experiment: which part of fence_heavy is the one that matters)
  __threadfence_block();
  if (mode & 1u) __builtin_amdgcn_s_dcache_inv();
  if (mode & 2u) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (mode & 4u) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  if (mode & 8u) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (mode & 16u) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (mode & 32u) for (int i = 0; i < 40; i++) __builtin_amdgcn_s_sleep(127);  // (a pure delay of a few microseconds)
  if (mode & 64u) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (mode & 128u) asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
}
// hides where a pointer came from: a pointer to LDS that is about to be biased (so that absolute slot numbers index a staged
// copy) must stay a 64-bit generic pointer — if the compiler keeps it as a 32-bit LDS offset, the biased value wraps and its
// conversion back to a generic address lands outside every aperture
template <class T>
PMA_DEV T *opaque_ptr(T *p) {
  asm volatile("" : "+v"(p));
  return p;
}
// every global load this wave has issued has returned (a workgroup barrier alone does not wait for vmcnt)
PMA_DEV void wait_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
PMA_DEV void flag_publish(uint32_t *p, uint32_t v) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a load that is coherent across the device (bypasses what this XCD's L2 may still hold)
// ... and a store that goes straight to the coherence point.  nodes[] interleaves plain words (beginning, end) with a word
// that only device-wide atomics change (num_neighbors): a plain store into a line this XCD's L2 holds valid makes the L2 write
// the WHOLE line back later — over what the atomics of the other XCDs have done to num_neighbors in the meantime
PMA_DEV void store_agent_u32(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PMA_DEV uint32_t load_agent_u32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PMA_DEV uint32_t flag_read(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PMA_DEV void flag_acquire() { asm volatile("" ::: "memory"); }
// which of the 8 XCDs this wave runs on (HW_REG_XCC_ID bits 3:0) — used for affinity only, never for correctness
PMA_DEV uint32_t xcc_id() { return (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; }
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
