// Owner bucketing for the multi-GPU exchange.
#pragma once
#include "pma_device.h"

namespace ppcsr {

// ---- owner bucketing for the multi-GPU exchange (PPPCSR routing rule, PPPCSR.cpp:46-66) ----------------------------
// Stable counting sort of a block of the update stream by owning partition, with `src` made partition-local.
// Tile = 2048 updates per workgroup; (1) per-tile histogram, (2) one small scan (partition-major, tile-minor),
// (3) scatter: rank inside the wave from ballots over the distinct owners present, inside the tile from an LDS
// prefix over (row, wave), across tiles from the scan — so every bucket keeps stream order.
constexpr uint32_t kBucketRows = 8;                          // rows of 256 updates per tile
constexpr uint32_t kBucketTile = 256 * kBucketRows;
constexpr uint32_t kMaxParts = 64;
struct PartTable {  // first global vertex of every partition (PPPCSR::distribution, PPPCSR.h:57), passed by value
  uint32_t start[kMaxParts];
};
// PPPCSR::get_partiton (PPPCSR.cpp:58-66): the last partition whose first vertex is <= src (starts are non-decreasing,
// start[0] = 0; equal starts — empty partitions — resolve to the last of them, as the reference's linear walk does)
PMA_DEV uint32_t owner_of_src(uint32_t src, const uint32_t *pstart /* LDS */, uint32_t nparts) {
  uint32_t lo = 0, hi = nparts;
  while (hi - lo > 1u) {
    const uint32_t mid = (lo + hi) >> 1;
    if (pstart[mid] <= src) lo = mid; else hi = mid;
  }
  return lo;
}
// counts[w][p] for the 4 waves of one row; returns this lane's rank among same-owner lanes of its wave
PMA_DEV uint32_t bucket_rank_in_wave(uint32_t owner, bool valid, uint32_t *wave_counts /* [kMaxParts] of this wave */) {
  const int lane = wv::lane();
  uint64_t remaining = wv::ballot(valid);
  uint32_t myrank = 0;
  while (remaining) {
    const int l0 = wv::ctz64(remaining);
    const uint32_t p0 = wv::shfl(owner, l0);
    const uint64_t m = wv::ballot(valid && owner == p0);
    if (valid && owner == p0) myrank = dev::lanemask_lt_count(m, lane);
    if (lane == 0) wave_counts[p0] = (uint32_t)wv::popc64(m);
    remaining &= ~m;
  }
  return myrank;
}
PMA_KERNEL void k_bucket_hist(const Op *ops, uint64_t n, PartTable tab, uint32_t nparts, uint32_t *hist /* [ntiles][nparts] */) {
  PMA_SHARED uint32_t cnt[kBucketRows][4][kMaxParts];
  PMA_SHARED uint32_t pstart[kMaxParts];
  const uint32_t tid = wv::thread_idx();
  const int w = wv::wave_in_block();
  const uint64_t tile = wv::block_idx();
  for (uint32_t i = tid; i < kBucketRows * 4 * kMaxParts; i += 256) (&cnt[0][0][0])[i] = 0;
  if (tid < kMaxParts) pstart[tid] = tab.start[tid];
  wv::block_sync();
  for (uint32_t r = 0; r < kBucketRows; r++) {
    const uint64_t i = tile * kBucketTile + (uint64_t)r * 256 + tid;
    const bool valid = i < n;
    const uint32_t owner = valid ? owner_of_src(ops[i].src, pstart, nparts) : 0u;
    (void)bucket_rank_in_wave(owner, valid, cnt[r][w]);
  }
  wv::block_sync();
  for (uint32_t p = tid; p < nparts; p += 256) {
    uint32_t t = 0;
    for (uint32_t r = 0; r < kBucketRows; r++)
      for (uint32_t q = 0; q < 4; q++) t += cnt[r][q][p];
    hist[tile * nparts + p] = t;
  }
}
// exclusive offsets, partition-major: off[tile][p] = sum_{p'<p} total[p'] + sum_{tile'<tile} hist[tile'][p]; counts[p] = total[p]
// ONE workgroup of kBucketScanThreads = 16 groups x 64 partitions: group g owns a contiguous run of tiles, thread (g, p)
// sums partition p over that run, the 16 x 64 partial sums are scanned in LDS, and the run is walked once more to write
// the offsets.  (One thread per partition walking all tiles — 4883 of them for a 10 M-update block — took 1.2 ms.)
constexpr uint32_t kBucketScanThreads = 1024;
PMA_KERNEL void k_bucket_scan(uint32_t *hist, uint64_t ntiles, uint32_t nparts, unsigned long long *counts) {
  constexpr uint32_t G = kBucketScanThreads / kMaxParts;
  PMA_SHARED unsigned long long part[G][kMaxParts];  // sum of partition p over group g's tiles
  PMA_SHARED unsigned long long tot[kMaxParts];
  const uint32_t tid = wv::thread_idx();
  const uint32_t g = tid / kMaxParts, p = tid % kMaxParts;
  const uint64_t per = (ntiles + G - 1) / G;
  const uint64_t t0 = (uint64_t)g * per, t1 = (t0 + per < ntiles) ? t0 + per : ntiles;
  unsigned long long mine = 0;
  if (p < nparts)
    for (uint64_t t = t0; t < t1; t++) mine += hist[t * nparts + p];
  part[g][p] = mine;
  wv::block_sync();
  if (g == 0 && p < nparts) {
    unsigned long long run = 0;
    for (uint32_t q = 0; q < G; q++) {
      const unsigned long long x = part[q][p];
      part[q][p] = run;  // partition p: what the groups in front of q hold
      run += x;
    }
    tot[p] = run;
    counts[p] = run;
  }
  wv::block_sync();
  if (p < nparts) {
    unsigned long long run = part[g][p];
    for (uint32_t q = 0; q < p; q++) run += tot[q];  // + everything of the partitions in front of p
    for (uint64_t t = t0; t < t1; t++) {
      const uint32_t c = hist[t * nparts + p];
      hist[t * nparts + p] = (uint32_t)run;
      run += c;
    }
  }
}
PMA_KERNEL void k_bucket_scatter(const Op *ops, uint64_t n, PartTable tab, uint32_t nparts, const uint32_t *off, Op *out) {
  PMA_SHARED uint32_t cnt[kBucketRows][4][kMaxParts];
  PMA_SHARED uint32_t pstart[kMaxParts];
  const uint32_t tid = wv::thread_idx();
  const int w = wv::wave_in_block();
  const uint64_t tile = wv::block_idx();
  for (uint32_t i = tid; i < kBucketRows * 4 * kMaxParts; i += 256) (&cnt[0][0][0])[i] = 0;
  if (tid < kMaxParts) pstart[tid] = tab.start[tid];
  wv::block_sync();
  Op mine[kBucketRows];
  uint32_t owner[kBucketRows], rank[kBucketRows];
  for (uint32_t r = 0; r < kBucketRows; r++) {
    const uint64_t i = tile * kBucketTile + (uint64_t)r * 256 + tid;
    const bool valid = i < n;
    mine[r] = valid ? ops[i] : Op{0u, 0u, 0u};
    owner[r] = valid ? owner_of_src(mine[r].src, pstart, nparts) : 0u;
    rank[r] = bucket_rank_in_wave(owner[r], valid, cnt[r][w]);
  }
  wv::block_sync();
  // exclusive prefix over (row, wave) per partition, in place (one thread per partition)
  for (uint32_t p = tid; p < nparts; p += 256) {
    uint32_t run = 0;
    for (uint32_t r = 0; r < kBucketRows; r++)
      for (uint32_t q = 0; q < 4; q++) {
        const uint32_t c = cnt[r][q][p];
        cnt[r][q][p] = run;
        run += c;
      }
  }
  wv::block_sync();
  for (uint32_t r = 0; r < kBucketRows; r++) {
    const uint64_t i = tile * kBucketTile + (uint64_t)r * 256 + tid;
    if (i < n) {
      const uint32_t p = owner[r];
      Op o = mine[r];
      o.src = o.src - pstart[p];  // partition-local source, global destination (PPPCSR.cpp:46-52)
      out[(uint64_t)off[tile * nparts + p] + cnt[r][w][p] + rank[r]] = o;
    }
  }
}

}  // namespace ppcsr
