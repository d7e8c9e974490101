// Read side: edge_exists / get_neighbourhood (PCSR.cpp:860-869, 901-912), the bulk neighbour scan (CSR export), the
// non-parity bulk build, BFS and PageRank over the gapped array (src/utility/bfs.h, pagerank.h).
#pragma once
#include "pma_rebalance.h"

namespace ppcsr {

// ---- read-side kernels (get_neighbourhood PCSR.cpp:901-912, edge_exists :860-869) --------------------------
PMA_KERNEL void k_edge_exists(View v, uint32_t src, uint32_t dst, ExclOut *out) {
  dev::RangeRec rr;
  rr.on = false;
  uint32_t found = 0;
  if (src < v.g.n) {
    const Node nd = v.nodes[src];
    dev::SearchHit hit_;
    const uint32_t loc = dev::pma_search(v, dst, nd.beginning + 1, nd.end, rr, &hit_);
    const Edge e = v.items[loc];
    found = (!is_null(e) && !is_sentinel(e) && e.dest == dst) ? 1u : 0u;
  }
  if (wv::lane() == 0) {
    out->found = found;
    out->result = X_DONE;
  }
}

// neighbours of one vertex: live dests in slots (beginning, end), in slot order; single workgroup of one wave
PMA_KERNEL void k_neighbourhood(View v, uint32_t src, int *outbuf, uint64_t cap, unsigned long long *count) {
  const int lane = wv::lane();
  unsigned long long run = 0;
  if (src < v.g.n) {
    const Node nd = v.nodes[src];
    for (uint64_t base = (uint64_t)nd.beginning + 1; base < (uint64_t)nd.end; base += 64) {
      const uint64_t s = base + (uint64_t)lane;
      Edge e = null_edge();
      if (s < (uint64_t)nd.end) e = v.items[s];
      const bool nn = e.value != 0;
      const uint64_t m = wv::ballot(nn);
      if (nn) {
        const unsigned long long o = run + dev::lanemask_lt_count(m, lane);
        if (outbuf && o < cap) outbuf[o] = (int)e.dest;
      }
      run += (unsigned long long)wv::popc64(m);
    }
  }
  if (lane == 0) *count = run;
}

// live edges per 64-slot chunk WITHOUT reading the edge array: leaf counts minus the sentinels that sit in the chunk
// (one atomic per vertex on a 4 B/chunk histogram), minus slot N-1 which is never part of a neighbourhood
PMA_KERNEL void k_chunk_sentinels(View v, uint32_t *chunk_sent) {
  // sentinel positions increase with the vertex id, so the sentinels of one chunk are a run of consecutive vertices:
  // the first vertex of each run counts the run and stores it (no atomics; chunks without sentinels stay 0).  Runs are
  // measured inside the wave with one ballot; the wave's last run may continue into the next 64 vertices (isolated
  // vertices sit shoulder to shoulder, up to 64 per chunk) and is finished with one more 64-wide probe.
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  const uint64_t n = v.g.n;
  const int lane = wv::lane();
  for (uint64_t base = (uint64_t)wv::block_idx() * wv::block_dim() + (wv::thread_idx() & ~63u); base < n; base += stride) {
    const uint64_t k = base + (uint64_t)lane;
    const bool valid = k < n;
    const uint32_t ch = valid ? (v.nodes[k].beginning >> 6) : kMax;
    uint32_t prev = wv::shfl(ch, lane == 0 ? 0 : lane - 1);
    if (lane == 0) prev = (k > 0) ? (v.nodes[k - 1].beginning >> 6) : kMax;
    const bool head = valid && (k == 0 || prev != ch);
    const uint64_t hm = wv::ballot(head);
    const int nvalid = wv::popc64(wv::ballot(valid));
    if (hm == 0) continue;  // the whole wave lies inside a run that an earlier wave counts
    const int lh = 63 - __builtin_clzll(hm);  // the wave's last run starts here
    if (head && lane != lh) {
      const uint64_t later = (hm >> (lane + 1)) << (lane + 1);
      chunk_sent[ch] = (uint32_t)(wv::ctz64(later) - lane);
    }
    const uint32_t chl = wv::shfl(ch, lh);
    uint32_t run = (uint32_t)(nvalid - lh);
    if (nvalid == 64) {  // a chunk holds at most 64 sentinels, so one probe of the next 64 vertices finishes the run
      const uint64_t k2 = base + 64 + (uint64_t)lane;
      const bool same = k2 < n && (v.nodes[k2].beginning >> 6) == chl;
      const uint64_t diff = wv::ballot(!same);
      run += diff ? (uint32_t)wv::ctz64(diff) : 64u;
    }
    if (lane == lh) chunk_sent[chl] = run;
  }
}
// live-edge count of every 64-slot chunk (leaf counts minus sentinels; slot N-1 is never part of a neighbourhood) and
// the sum over each tile of `tile_chunks` chunks; chunk_sent is left zeroed for the next scan
PMA_KERNEL void k_chunk_counts(View v, uint32_t *chunk_sent, uint32_t *chunkcnt, uint32_t tile_chunks, uint32_t *tilesum) {
  PMA_SHARED uint32_t red[4];
  const uint64_t N = v.g.N, nchunks = (N + 63) / 64;
  const uint32_t lpc = (v.g.logN >= 64) ? 1u : (64u >> v.g.sh);  // leaves per chunk
  const uint64_t ch = (uint64_t)wv::block_idx() * tile_chunks + wv::thread_idx();
  uint32_t c = 0;
  if (wv::thread_idx() < tile_chunks && ch < nchunks) {
    if (v.g.logN >= 64) {
      c = v.leafcnt[(ch * 64) >> v.g.sh];  // (logN = 64 only for N >= 2^32: not reachable, kept for completeness)
    } else {
      for (uint32_t q = 0; q < lpc; q++) {
        const uint64_t leaf = ch * lpc + q;
        if ((leaf << v.g.sh) < N) c += v.leafcnt[leaf];
      }
    }
    c -= chunk_sent[ch];
    chunk_sent[ch] = 0u;
    if (ch == nchunks - 1) {
      const Edge e = v.items[N - 1];
      if (e.value != 0 && !is_sentinel(e)) c -= 1u;
    }
    chunkcnt[ch] = c;
  }
  const uint32_t s = wv::reduce_add(c);
  if (wv::lane() == 0) red[wv::wave_in_block()] = s;
  wv::block_sync();
  if (wv::thread_idx() == 0) tilesum[wv::block_idx()] = red[0] + red[1] + red[2] + red[3];
}
// bulk neighbour scan (CSR export), final streaming pass: one workgroup per tile of chunks.  The tile's chunk counts are
// scanned in LDS (offset = scanned tile sum + in-tile prefix), then every wave streams its chunks — four in flight —
// writing dests in array order == CSR order and the row offsets at the sentinels.
// (contrib != nullptr: also emit, per edge, node_values[src] / num_neighbors(src) — the PageRank push of pagerank.h:21;
//  triples != nullptr: emit (src + src_base, dest, value) per edge instead of / besides dests)
PMA_KERNEL void k_scan_write(View v, const uint32_t *__restrict__ chunkcnt, uint32_t tile_chunks, const uint32_t *__restrict__ tile_excl,
                             unsigned long long *__restrict__ row_offsets, int *__restrict__ dests, uint64_t cap,
                             const float *__restrict__ node_values, float *__restrict__ contrib, Op *__restrict__ triples, uint32_t src_base) {
  PMA_SHARED uint32_t pre[256];
  PMA_SHARED uint32_t wsum[4];
  const int lane = wv::lane(), w = wv::wave_in_block();
  const uint64_t N = v.g.N;
  const uint64_t nchunks = (N + 63) / 64;
  const uint64_t tile = wv::block_idx();
  {
    const uint64_t ch = tile * tile_chunks + wv::thread_idx();
    const uint32_t x = (wv::thread_idx() < tile_chunks && ch < nchunks) ? chunkcnt[ch] : 0u;
    uint32_t incl = x;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = wv::shfl(incl, lane - o < 0 ? 0 : lane - o);
      if (lane >= o) incl += y;
    }
    if (lane == 63) wsum[w] = incl;
    wv::block_sync();
    uint32_t woff = 0;
    for (int q = 0; q < w; q++) woff += wsum[q];
    pre[wv::thread_idx()] = woff + incl - x;
  }
  wv::block_sync();
  const unsigned long long base = tile_excl[tile];
  const Edge *__restrict__ items = v.items;
  constexpr int K = 4;
  for (uint32_t c0 = (uint32_t)w * K; c0 < tile_chunks; c0 += 4 * K) {
    Edge e[K];
#pragma unroll
    for (int q = 0; q < K; q++) {
      const uint64_t s = (tile * tile_chunks + c0 + q) * 64 + (uint64_t)lane;
      e[q] = null_edge();
      if (c0 + q < tile_chunks && s < N) e[q] = items[s];
    }
#pragma unroll
    for (int q = 0; q < K; q++) {
      if (c0 + q >= tile_chunks) break;
      const uint64_t s = (tile * tile_chunks + c0 + q) * 64 + (uint64_t)lane;
      const bool nn = e[q].value != 0;
      const bool sent = nn && is_sentinel(e[q]);
      const bool live = nn && !sent && (s + 1 < N);
      const uint64_t m = wv::ballot(live);
      const unsigned long long o = base + pre[c0 + q] + dev::lanemask_lt_count(m, lane);
      if (live && o < cap && triples != nullptr)  // (pppcsr_repartition: the edge as an add of the global stream)
        triples[o] = Op{e[q].src + src_base, e[q].dest, e[q].value};
      if (live && o < cap && dests != nullptr) {
        dests[o] = (int)e[q].dest;
        if (contrib != nullptr) {
          if (e[q].dest >= v.g.n) dests[o] = (int)v.g.n;  // (the reference would write out of bounds; keeps the sort keys short)
          const uint32_t sv = e[q].src;
          contrib[o] = (sv < v.g.n) ? node_values[sv] / (float)v.nodes[sv].num_neighbors : 0.0f;
        }
      }
      if (sent && row_offsets != nullptr) {
        const uint32_t vid = (e[q].value == kMax) ? 0u : e[q].value;
        row_offsets[vid] = o;
      }
    }
  }
}

// ---- bulk build: an explicit NON-parity fast path ---------------------------------------------------
// The reference can only build a graph by single inserts, and the layout that produces is history dependent; this path
// builds a VALID packed-memory array (same invariants, same neighbourhoods, same num_neighbors) in a handful of passes:
// sort the adds by (src, dest) (stable: the last value of a duplicate wins, as it does when inserted one by one), then
// place sentinels and unique edges with the exact redistribute() positions of one whole-array window.
PMA_KERNEL void k_bb_keys(const Op *ops, uint64_t m, uint32_t n, unsigned long long *keys, uint32_t *vals) {
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t i = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); i < m; i += stride) {
    const Op o = ops[i];
    const bool ok = o.op != 0 && o.src < n;  // (add_edge ignores value 0 and src >= n, PCSR.cpp:1375-1377)
    keys[i] = ok ? (((unsigned long long)o.src << 32) | (unsigned long long)o.dst) : ((unsigned long long)n << 32);  // (sorts last)
    vals[i] = o.op;
  }
}
PMA_KERNEL void k_bb_flags(const unsigned long long *keys, uint64_t m, uint32_t n, uint32_t *flags) {
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t i = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); i < m; i += stride) {
    const unsigned long long k = keys[i];
    flags[i] = ((uint32_t)(k >> 32) < n && (i + 1 == m || keys[i + 1] != k)) ? 1u : 0u;  // last of its run = the value that survives
  }
}
PMA_DEV uint64_t bb_lower_bound(const unsigned long long *keys, uint64_t m, unsigned long long key) {
  uint64_t lo = 0, hi = m;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (keys[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// position of element k of the whole-array window, table in LDS, segment found by bisection
PMA_DEV uint64_t bb_pos(const ChainTable *tb, uint64_t k) {
  if (k == 0) return tb->index;
  const uint64_t t = tb->j - 1 - k;
  int lo = 0, hi = tb->nseg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tb->seg[mid].t0 <= t) lo = mid; else hi = mid - 1;
  }
  const ChainSeg &sg = tb->seg[lo];
  const uint64_t d = t - sg.t0;
  const uint64_t M = (d == 0) ? sg.M0 : (sg.M0 - sg.Dfirst - (d - 1) * sg.Drest);
  return M >> sg.shift;
}
PMA_DEV void bb_load_table(ChainTable *stb, const ChainTable *tb) {
  const uint32_t *g = reinterpret_cast<const uint32_t *>(tb);
  uint32_t *sp = reinterpret_cast<uint32_t *>(stb);
  const uint32_t words = (uint32_t)((sizeof(ChainTable) - sizeof(ChainSeg) * (size_t)(kMaxSeg - tb->nseg)) / 4);
  for (uint32_t i = wv::thread_idx(); i < words; i += wv::block_dim()) sp[i] = g[i];
  wv::block_sync();
}
// sentinel of every vertex + nodes[]: vertex u is element u + (unique edges of smaller sources) of the sequence
PMA_KERNEL void k_bb_vertices(View v, const unsigned long long *keys, uint64_t m, const uint32_t *rank, const unsigned long long *total,
                              const ChainTable *tb) {
  PMA_SHARED ChainTable stb;
  bb_load_table(&stb, tb);
  const uint64_t E = *total;
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t u = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); u < v.g.n; u += stride) {
    const uint64_t a = bb_lower_bound(keys, m, (unsigned long long)u << 32);
    const uint64_t b = bb_lower_bound(keys, m, (unsigned long long)(u + 1) << 32);
    const uint64_t before = (a < m) ? (uint64_t)rank[a] : E;  // unique edges whose source is smaller than u
    const uint64_t pos = bb_pos(&stb, u + before);
    Edge e;
    e.src = (uint32_t)u;
    e.dest = kMax;
    e.value = (u == 0) ? kMax : (uint32_t)u;
    v.items[pos] = e;
    v.nodes[u].beginning = (uint32_t)pos;
    v.nodes[u].num_neighbors = (uint32_t)(b - a);  // every add counts, duplicates included (PCSR.cpp:1408)
    if (u > 0) v.nodes[u - 1].end = (uint32_t)pos;
    if (u + 1 == v.g.n) v.nodes[u].end = (uint32_t)(v.g.N - 1);
  }
}
PMA_KERNEL void k_bb_edges(View v, const unsigned long long *keys, const uint32_t *vals, const uint32_t *flags, const uint32_t *rank,
                           uint64_t m, const ChainTable *tb) {
  PMA_SHARED ChainTable stb;
  bb_load_table(&stb, tb);
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t i = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); i < m; i += stride) {
    if (!flags[i]) continue;
    const unsigned long long k = keys[i];
    const uint32_t src = (uint32_t)(k >> 32);
    const uint64_t pos = bb_pos(&stb, (uint64_t)rank[i] + (uint64_t)src + 1ull);  // sentinels 0..src precede it
    Edge e;
    e.src = src;
    e.dest = (uint32_t)k;
    e.value = vals[i];
    v.items[pos] = e;
  }
}

// ---- graph-algorithm consumers over the gapped array (reference: src/utility/bfs.h, src/utility/pagerank.h) -----------
// BFS, one level per launch: one wave per frontier vertex walks its slot range (beginning, end) 64 slots at a time, skips
// nulls, claims unvisited neighbours with a compare-and-swap on their level and appends them to the next frontier (one
// atomic per wave per 64 slots).  Levels are unique, so the result equals the reference's queue-based walk exactly.
constexpr uint64_t kBfsWaveSlots = 4096;  // longest slot range one wave walks on its own
PMA_KERNEL void k_bfs_level(View v, const uint32_t *front, uint32_t nfront, uint32_t level, uint32_t *levels, uint32_t *next,
                            uint32_t *next_count) {
  const int lane = wv::lane();
  const uint64_t wstride = (uint64_t)wv::grid_dim() * (wv::block_dim() >> 6);
  for (uint64_t f = (uint64_t)wv::block_idx() * (wv::block_dim() >> 6) + wv::wave_in_block(); f < nfront; f += wstride) {
    const uint32_t u = front[f];
    const Node nd = v.nodes[u];
    if ((uint64_t)nd.end - (uint64_t)nd.beginning > kBfsWaveSlots) {  // a hub: leave it to one streaming pass (k_bfs_edges)
      if (lane == 0) next_count[1] = 1u;
      continue;
    }
    for (uint64_t base = (uint64_t)nd.beginning + 1; base < (uint64_t)nd.end; base += 64) {
      const uint64_t s = base + (uint64_t)lane;
      uint32_t val = 0, dst = 0;
      if (s < (uint64_t)nd.end) {
        val = v.items[s].value;
        dst = v.items[s].dest;
      }
      bool won = false;
      if (val != 0 && dst < v.g.n && levels[dst] == kMax) won = wv::atomic_cas_u32(&levels[dst], kMax, level + 1u) == kMax;
      const uint64_t m = wv::ballot(won);
      if (m) {
        uint32_t b = 0;
        if (lane == 0) b = wv::atomic_add_u32(next_count, (uint32_t)wv::popc64(m));
        b = wv::shfl(b, 0);
        if (won) next[b + dev::lanemask_lt_count(m, lane)] = dst;
      }
    }
  }
}
// BFS level for a LARGE frontier: one streaming pass over the gapped array instead of one wave per frontier vertex (whose
// hubs would serialise the level): every live edge whose source sits on the current level claims its destination.  All
// writers of a level store the same value, so plain stores suffice; `found` counts the claims (an upper bound is enough:
// it only steers the choice of the next level's kernel, and zero means "done").
PMA_KERNEL void k_bfs_edges(View v, uint32_t level, uint32_t *levels, uint32_t *found) {
  const int lane = wv::lane();
  const uint64_t N = v.g.N, nchunks = (N + 63) / 64;
  const uint64_t wstride = (uint64_t)wv::grid_dim() * (wv::block_dim() >> 6);
  uint32_t mine = 0;
  for (uint64_t ch = (uint64_t)wv::block_idx() * (wv::block_dim() >> 6) + wv::wave_in_block(); ch < nchunks; ch += wstride) {
    const uint64_t s = ch * 64 + (uint64_t)lane;
    Edge e = null_edge();
    if (s + 1 < N) e = v.items[s];  // (slot N-1 is never part of a neighbourhood)
    const bool live = e.value != 0 && !is_sentinel(e) && e.src < v.g.n && e.dest < v.g.n;
    if (live && levels[e.src] == level && levels[e.dest] == kMax) {
      levels[e.dest] = level + 1u;
      mine++;
    }
  }
  mine = wv::reduce_add(mine);
  if (lane == 0 && mine) wv::atomic_add_u32(found, mine);
}
// The streaming level, bitmap form.  The level's two per-edge tests — "is the source on the frontier", "is the destination
// still unvisited" — used to be two gathers from levels[] (4 MB at n = 1 M: 64-B lines fetched for 4 B, and the pass ran
// at 0.7-1.3 TB/s).  k_bfs_bits packs both answers into two bitmaps of n/8 bytes (128 KB: L2-resident on every XCD) with
// one coalesced sweep over levels[] per level; k_bfs_edges_bits then streams the array with four 64-slot chunks in flight
// per wave and touches levels[] only for edges into vertices that were unvisited when the level began.  Measured on an
// RMAT-20 / 10 M-edge graph (201 MB of slots): 35-38 us on light levels (5.5 TB/s), 63 / 40 us on the two heavy ones.
// (Claiming destinations with atomic ORs into the visited bitmap instead — exact `found`, one store per vertex — cost
// 195 / 100 us: 0.4 M atomics on 1024 cache lines are served by the memory side one line at a time.)
PMA_KERNEL void k_bfs_bits(const uint32_t *levels, uint32_t n, uint32_t level, uint32_t *front_bits, uint32_t *visited_bits) {
  const int lane = wv::lane();
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t base = (uint64_t)wv::block_idx() * wv::block_dim() + (wv::thread_idx() & ~63u); base < n; base += stride) {
    const uint64_t u = base + (uint64_t)lane;
    const uint32_t lv = u < n ? levels[u] : kMax;
    const uint64_t mf = wv::ballot(u < n && lv == level), mv = wv::ballot(u < n && lv != kMax);
    if (lane < 2) {
      front_bits[(base >> 5) + lane] = (uint32_t)(mf >> (32 * lane));
      visited_bits[(base >> 5) + lane] = (uint32_t)(mv >> (32 * lane));
    }
  }
}
constexpr uint32_t kBfsStripes = 64, kBfsStripeWords = 32;
PMA_KERNEL void k_bfs_edges_bits(View v, uint32_t level, const uint32_t *__restrict__ front_bits, const uint32_t *__restrict__ visited_bits,
                                 uint32_t *levels, uint32_t *found) {
  const int lane = wv::lane();
  const uint64_t N = v.g.N, nchunks = (N + 63) / 64;
  const uint64_t wstride = (uint64_t)wv::grid_dim() * (wv::block_dim() >> 6);
  const uint32_t n = v.g.n;
  uint32_t mine = 0;
  constexpr int kB = 4;
  for (uint64_t ch0 = ((uint64_t)wv::block_idx() * (wv::block_dim() >> 6) + wv::wave_in_block()) * kB; ch0 < nchunks; ch0 += wstride * kB) {
    Edge e[kB];
#pragma unroll
    for (int b = 0; b < kB; b++) {
      const uint64_t s = (ch0 + b) * 64 + (uint64_t)lane;
      e[b] = null_edge();
      if (s + 1 < N) e[b] = v.items[s];  // (slot N-1 is never part of a neighbourhood)
    }
    // Four phases, each over all kB chunks, so that the kB gathers of a phase are in flight TOGETHER (written one chunk after
    // the other, the levels[] load of chunk b+1 waits for the store of chunk b: they may alias).
    bool hit[kB];
    uint32_t bit[kB], old[kB];
#pragma unroll
    for (int b = 0; b < kB; b++) {
      const bool live = e[b].value != 0 && !is_sentinel(e[b]) && e[b].src < n && e[b].dest < n;
      hit[b] = live && ((front_bits[e[b].src >> 5] >> (e[b].src & 31u)) & 1u);
      bit[b] = 1u << (e[b].dest & 31u);
    }
#pragma unroll
    for (int b = 0; b < kB; b++) old[b] = hit[b] ? visited_bits[e[b].dest >> 5] : 0xFFFFFFFFu;
    // (the bitmap is the state at the start of the level; a look at levels[] itself — only for edges into NEW vertices —
    // keeps most of the repeated stores away.  It may be stale: all writers of a level store the same value.)
#pragma unroll
    for (int b = 0; b < kB; b++) {
      hit[b] = (old[b] & bit[b]) == 0u;
      old[b] = hit[b] ? levels[e[b].dest] : 0u;
    }
#pragma unroll
    for (int b = 0; b < kB; b++) {
      if (hit[b] && old[b] == kMax) {
        levels[e[b].dest] = level + 1u;
        mine++;
      }
    }
  }
  // (`found` is kBfsStripes counters on cache lines of their own, one add per workgroup: on a heavy level nearly every wave
  // has claims, and 32 K adds to ONE word are served one after the other by the memory side — that was 260-290 us of the
  // 320 / 290 us the heavy levels took, whatever the per-edge work looked like)
  PMA_SHARED uint32_t red[4];
  mine = wv::reduce_add(mine);
  if (lane == 0) red[wv::wave_in_block()] = mine;
  wv::block_sync();
  if (wv::thread_idx() == 0) {
    const uint32_t all = red[0] + red[1] + red[2] + red[3];
    if (all) wv::atomic_add_u32(found + (uint64_t)(wv::block_idx() % kBfsStripes) * kBfsStripeWords, all);
  }
}
// frontier list of one level (used when a small frontier follows an edge-centric level)
PMA_KERNEL void k_bfs_collect(const uint32_t *levels, uint32_t n, uint32_t level, uint32_t *front, uint32_t *count) {
  const int lane = wv::lane();
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t base = (uint64_t)wv::block_idx() * wv::block_dim() + (wv::thread_idx() & ~63u); base < n; base += stride) {
    const uint64_t u = base + (uint64_t)lane;
    const bool in = u < n && levels[u] == level;
    const uint64_t m = wv::ballot(in);
    if (m) {
      uint32_t b = 0;
      if (lane == 0) b = wv::atomic_add_u32(count, (uint32_t)wv::popc64(m));
      b = wv::shfl(b, 0);
      if (in) front[b + dev::lanemask_lt_count(m, lane)] = (uint32_t)u;
    }
  }
}
// PageRank push, last step: contributions sorted (stably) by destination; every destination's run is added IN ORDER —
// ascending source, the order in which the reference's loop adds them — so the fp32 sums are the reference's bit for bit.
// One thread per destination handles short runs; a run of kPrLongRun or more is queued for k_pr_longruns.
constexpr uint32_t kPrLongRun = 128;
PMA_KERNEL void k_pr_segsum(const uint32_t *__restrict__ keys, const float *__restrict__ vals, uint64_t m, uint32_t n, float *out,
                            uint32_t *long_list, uint32_t *long_count) {
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t d = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); d < n; d += stride) {
    uint64_t lo = 0, hi = m;  // first position with keys[pos] >= d
    while (lo < hi) {
      const uint64_t mid = (lo + hi) >> 1;
      if (keys[mid] < (uint32_t)d) lo = mid + 1; else hi = mid;
    }
    uint64_t lo2 = lo, hi2 = (lo + kPrLongRun < m) ? lo + kPrLongRun : m;  // first position (within reach) with keys[pos] > d
    while (lo2 < hi2) {
      const uint64_t mid = (lo2 + hi2) >> 1;
      if (keys[mid] <= (uint32_t)d) lo2 = mid + 1; else hi2 = mid;
    }
    if (lo2 - lo >= kPrLongRun) {  // long (or longer) run: a whole wave streams it
      long_list[wv::atomic_add_u32(long_count, 1u)] = (uint32_t)d;
      continue;
    }
    float acc = 0.0f;
    for (uint64_t i = lo; i < lo2; i++) acc += vals[i];
    out[d] = acc;
  }
}
// one wave per long run: 64 contributions are loaded at once, then added one after the other in order (the adds are the
// serial part by definition; the loads no longer are)
PMA_KERNEL void k_pr_longruns(const uint32_t *__restrict__ keys, const float *__restrict__ vals, uint64_t m, const uint32_t *long_list,
                              const uint32_t *long_count, float *out) {
  const int lane = wv::lane();
  const uint32_t nl = *long_count;
  const uint64_t wstride = (uint64_t)wv::grid_dim() * (wv::block_dim() >> 6);
  for (uint64_t w = (uint64_t)wv::block_idx() * (wv::block_dim() >> 6) + wv::wave_in_block(); w < nl; w += wstride) {
    const uint32_t d = long_list[w];
    uint64_t lo = 0, hi = m;
    while (lo < hi) {
      const uint64_t mid = (lo + hi) >> 1;
      if (keys[mid] < d) lo = mid + 1; else hi = mid;
    }
    float acc = 0.0f;
    for (uint64_t base = lo; base < m; base += 64) {
      const uint64_t i = base + (uint64_t)lane;
      const bool in = i < m && keys[i] == d;
      float x = 0.0f;
      if (in) x = vals[i];
      const uint64_t mm = wv::ballot(in);
      const int cnt = wv::popc64(mm);  // (the run is contiguous: lanes 0 .. cnt-1)
      for (int q = 0; q < cnt; q++) acc += wv::shfl_f32(x, q);
      if (cnt < 64) break;
    }
    if (lane == 0) out[d] = acc;
  }
}

}  // namespace ppcsr
