// Whole-array and big-window kernels of the PMA engine: ctor state, recount, the fused rebalance (redistribute, PCSR.cpp:222-249 at
// array scale: double_list :251-282, half_list :284-320, windows the rounds hand back), the in-place window rebalance,
// incremental snapshots, small maintenance kernels.
#pragma once
#include "pma_device.h"

namespace ppcsr {

// ---- whole-array kernels ---------------------------------------------------------------------------------
// 12-byte null pattern {0xFFFFFFFF,0,0} written as a dword stream (coalesced)
PMA_KERNEL void k_fill_null(Edge *items, uint64_t start, uint64_t len) {
  uint32_t *w = reinterpret_cast<uint32_t *>(items + start);
  const uint64_t total = len * 3ull;
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t i = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); i < total; i += stride)
    w[i] = (i % 3ull == 0) ? kMax : 0u;
}

// leafcnt[leaf] for leaves [leaf_lo, leaf_lo+nleaves): one wave per 64 slots
PMA_KERNEL void k_recount(View v, uint64_t slot_lo, uint64_t nslots) {
  const int lane = wv::lane();
  const uint32_t logN = (uint32_t)v.g.logN;
  const int sh = v.g.sh;
  const uint64_t nchunks = (nslots + 63) / 64;
  const uint64_t wstride = (uint64_t)wv::grid_dim() * (wv::block_dim() >> 6);
  for (uint64_t ch = (uint64_t)wv::block_idx() * (wv::block_dim() >> 6) + wv::wave_in_block(); ch < nchunks; ch += wstride) {
    const uint64_t s = slot_lo + ch * 64 + (uint64_t)lane;
    bool nn = false;
    if (s < slot_lo + nslots) nn = v.items[s].value != 0;
    const uint64_t occ = wv::ballot(nn);
    const uint32_t nleaf = (logN >= 64) ? 1u : (64u >> sh);
    if ((uint32_t)lane < nleaf) {
      const uint64_t ls = slot_lo + ch * 64 + (uint64_t)lane * logN;
      if (ls < slot_lo + nslots) {
        const uint64_t sub = (logN >= 64) ? occ : ((occ >> ((uint32_t)lane * logN)) & ((1ull << logN) - 1ull));
        v.leafcnt[ls >> sh] = (uint32_t)wv::popc64(sub);
      }
    }
  }
}

// place the initial sentinels (constructor, PCSR.cpp:815-837): sentinel k sits at nodes[k].beginning
PMA_KERNEL void k_place_sentinels(View v) {
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t k = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); k < v.g.n; k += stride) {
    Edge e;
    e.src = (uint32_t)k;
    e.dest = kMax;
    e.value = (k == 0) ? kMax : (uint32_t)k;
    v.items[v.nodes[k].beginning] = e;
  }
}

// exclusive prefix sum of leafcnt over [leaf_lo, leaf_lo + nleaves) -> rank[i]; three small kernels
constexpr uint32_t kScanTile = 1024;  // leaves per workgroup
PMA_KERNEL void k_scan_tiles(const uint32_t *cnt, uint64_t nleaves, uint32_t *tilesum) {
  PMA_SHARED uint32_t red[4];
  const uint64_t b = wv::block_idx();
  uint32_t s = 0;
  for (uint32_t i = wv::thread_idx(); i < kScanTile; i += wv::block_dim()) {
    const uint64_t l = b * kScanTile + i;
    if (l < nleaves) s += cnt[l];
  }
  s = wv::reduce_add(s);
  if (wv::lane() == 0) red[wv::wave_in_block()] = s;
  wv::block_sync();
  if (wv::thread_idx() == 0) tilesum[b] = red[0] + red[1] + red[2] + red[3];
}
// ---- partial window rebalanced IN PLACE (no scratch copy, no copy-back) ----------------------------------------------
// The reference spreads a window inside the array itself (PCSR.cpp:207-247: pack to the left, then place right to left).
// Here a tile of kIpChunks x 4 x 64 source slots is held in the registers of one workgroup: the workgroup loads its tile,
// PUBLISHES that it has done so, waits until every tile whose source slots its own destination range covers has published
// too, and only then writes elements and null runs.  Both maps (k-th live element -> source slot, -> destination slot) are
// monotone, so tile i's destination range [c_i, d_i) is contiguous, d_i = c_{i+1}, and at every tile boundary the flow
// goes one way: "R" (d_i beyond tile i's last source slot: tile i writes over sources of tiles i+1...) or "L" (tile i+1
// writes over sources of tiles ...i).  A tile waits only for tiles further along its own run of R (or L) boundaries, so
//   key(i) = max(#consecutive R boundaries starting at i|i+1, #consecutive L boundaries ending at i-1|i)
// is strictly larger than the key of every tile that tile i waits for.  rb_order_body (the tail of k_scan_tilesums) sorts
// the tiles by key; k_rb_inplace workgroups draw tickets in that order — whoever a workgroup waits for drew an earlier ticket, is resident (or done) and
// publishes without waiting for anybody: no deadlock whatever the number of resident workgroups.  (A bounded spin turns a
// broken order into an error flag instead of a hang.)
#if defined(PPCSR_SCAN_DEBUG)
#define PMA_TSTAMP(dbg, i) do { if ((dbg) && wv::thread_idx() == 0) (dbg)[i] = (unsigned long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define PMA_TSTAMP(dbg, i) do { } while (0)
#endif
constexpr uint32_t kIpMaxTiles = 8192, kIpOrderThreads = 1024;
constexpr uint32_t kIpHdrWords = 32 * 9;  // words before the order list in the engine's buffer
constexpr uint32_t kIpTicketStride = 32;  // ctl[1]: sticky error flag; ctl[kIpTicketStride * (1 + x)]: ticket counter of XCD x
PMA_DEV void rb_order_body(const uint32_t *tile_excl, const ChainTable *tb, uint32_t ntiles, uint32_t tile_slots, uint32_t *order, uint32_t *ctl,
                           unsigned long long *dbg = nullptr) {
  PMA_SHARED ChainTable stb;
  PMA_SHARED unsigned long long nr[kIpMaxTiles / 64], nl[kIpMaxTiles / 64];  // bit b: boundary b|b+1 is NOT "R" / NOT "L"
  PMA_SHARED uint32_t hist[kIpMaxTiles + 1];
  PMA_SHARED uint32_t wtot[kIpOrderThreads / 64];
  PMA_SHARED uint32_t wsmall[kIpOrderThreads / 64][4];
  {
    const uint32_t *g = reinterpret_cast<const uint32_t *>(tb);
    uint32_t *sp = reinterpret_cast<uint32_t *>(&stb);
    const uint32_t words = (uint32_t)((sizeof(ChainTable) - sizeof(ChainSeg) * (size_t)(kMaxSeg - tb->nseg)) / 4);
    for (uint32_t i = wv::thread_idx(); i < words; i += kIpOrderThreads) sp[i] = g[i];
  }
  for (uint32_t i = wv::thread_idx(); i <= ntiles; i += kIpOrderThreads) hist[i] = 0u;
  if (wv::thread_idx() < 8u) ctl[kIpTicketStride * (1u + wv::thread_idx())] = 0u;  // the ticket counters
  wv::block_sync();
  PMA_TSTAMP(dbg, 11);
  const int lane = wv::lane(), w = wv::wave_in_block();
  const uint64_t j = stb.j, wend = stb.index + stb.len;
  const uint32_t nwords = (ntiles + 63u) / 64u;
  int hint = -1;
  for (uint32_t wd = (uint32_t)w; wd < nwords; wd += kIpOrderThreads / 64) {
    const uint32_t b = wd * 64u + (uint32_t)lane;
    bool is_r = false, is_l = false;
    if (b + 1u < ntiles) {
      const uint64_t K = tile_excl[b + 1u];
      const uint64_t D = K < j ? chain_pos(&stb, K, &hint) : wend;      // where tile b's destination range ends
      const uint64_t B = stb.index + (uint64_t)(b + 1u) * tile_slots;   // where tile b's source slots end
      is_r = D > B;
      is_l = D < B;
    }
    const uint64_t mr = wv::ballot(!is_r), ml = wv::ballot(!is_l);
    if (lane == 0) {
      nr[wd] = mr;
      nl[wd] = ml;
    }
  }
  wv::block_sync();
  PMA_TSTAMP(dbg, 12);
  constexpr uint32_t kPer = kIpMaxTiles / kIpOrderThreads;
  uint32_t key[kPer];
#pragma unroll
  for (uint32_t r = 0; r < kPer; r++) {
    const uint32_t i = r * kIpOrderThreads + wv::thread_idx();
    key[r] = 0;
    if (i >= ntiles) continue;
    uint32_t wd = i >> 6;  // first boundary >= i that is not R (bit ntiles-1 is always set)
    uint64_t m = nr[wd] >> (i & 63u);
    uint32_t nb;
    if (m) nb = i + (uint32_t)wv::ctz64(m);
    else {
      do wd++; while (nr[wd] == 0ull);
      nb = wd * 64u + (uint32_t)wv::ctz64(nr[wd]);
    }
    const uint32_t d_r = nb - i;
    uint32_t d_l = 0;
    if (i > 0) {  // last boundary <= i-1 that is not L (none: every boundary down to tile 0 is L)
      const uint32_t b = i - 1u;
      int wl = (int)(b >> 6);
      m = nl[wl] << (63u - (b & 63u));
      if (m) d_l = (uint32_t)wv::clz64(m);
      else {
        do wl--; while (wl >= 0 && nl[wl] == 0ull);
        d_l = wl < 0 ? i : b - ((uint32_t)wl * 64u + 63u - (uint32_t)wv::clz64(nl[wl]));
      }
    }
    key[r] = d_r > d_l ? d_r : d_l;
  }
  PMA_TSTAMP(dbg, 13);
  // Counting sort by key.  Almost every tile has a tiny key (a window whose elements barely move: 0 or 1 everywhere), so a
  // histogram through LDS atomics is thousands of adds to two or three addresses, served one after the other (11 us of the
  // launch).  Keys below kSmall are counted and ranked with ballots — per wave, combined through a small table — and only
  // the rare larger keys go through atomics.
  constexpr uint32_t kSmall = 4;
  uint32_t wcnt[kSmall];
#pragma unroll
  for (uint32_t k = 0; k < kSmall; k++) wcnt[k] = 0;
#pragma unroll
  for (uint32_t r = 0; r < kPer; r++) {
    const bool valid = r * kIpOrderThreads + wv::thread_idx() < ntiles;
#pragma unroll
    for (uint32_t k = 0; k < kSmall; k++) wcnt[k] += (uint32_t)wv::popc64(wv::ballot(valid && key[r] == k));
    if (valid && key[r] >= kSmall) wv::atomic_add_u32(&hist[key[r]], 1u);
  }
  if ((uint32_t)lane < kSmall) {
    uint32_t x = 0;
#pragma unroll
    for (uint32_t k = 0; k < kSmall; k++) x = ((uint32_t)lane == k) ? wcnt[k] : x;
    wsmall[w][lane] = x;
  }
  wv::block_sync();
  if (wv::thread_idx() < kSmall) {  // per key: the waves' counts -> exclusive prefix over the waves, total -> hist
    uint32_t run = 0;
    for (uint32_t q = 0; q < kIpOrderThreads / 64; q++) {
      const uint32_t x = wsmall[q][wv::thread_idx()];
      wsmall[q][wv::thread_idx()] = run;
      run += x;
    }
    hist[wv::thread_idx()] = run;
  }
  wv::block_sync();
  PMA_TSTAMP(dbg, 14);
  {  // exclusive scan of hist[0 .. ntiles]
    const uint32_t total = ntiles + 1u, per = (total + kIpOrderThreads - 1u) / kIpOrderThreads;
    const uint32_t lo = wv::thread_idx() * per, hi = lo + per < total ? lo + per : total;
    uint32_t mine = 0;
    for (uint32_t i = lo; i < hi; i++) mine += hist[i];
    uint32_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = wv::shfl(incl, lane - o < 0 ? 0 : lane - o);
      if (lane >= o) incl += y;
    }
    if (lane == 63) wtot[w] = incl;
    wv::block_sync();
    uint32_t run = incl - mine;
    for (int q = 0; q < w; q++) run += wtot[q];
    for (uint32_t i = lo; i < hi; i++) {
      const uint32_t x = hist[i];
      hist[i] = run;
      run += x;
    }
  }
  wv::block_sync();
  PMA_TSTAMP(dbg, 15);
  const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (uint32_t k = 0; k < kSmall; k++) wcnt[k] = 0;  // (now: how many of this wave's tiles with key k have been placed)
#pragma unroll
  for (uint32_t r = 0; r < kPer; r++) {
    const uint32_t i = r * kIpOrderThreads + wv::thread_idx();
    const bool valid = i < ntiles;
    uint32_t pos = 0;
#pragma unroll
    for (uint32_t k = 0; k < kSmall; k++) {
      const uint64_t mk = wv::ballot(valid && key[r] == k);
      if (valid && key[r] == k) pos = hist[k] + wsmall[w][k] + wcnt[k] + (uint32_t)wv::popc64(mk & lt_mask);
      wcnt[k] += (uint32_t)wv::popc64(mk);
    }
    if (valid && key[r] >= kSmall) pos = wv::atomic_add_u32(&hist[key[r]], 1u);
    if (valid) order[pos] = i;
  }
  wv::block_sync();
  PMA_TSTAMP(dbg, 16);
}


constexpr uint32_t kTileSumThreads = 1024;
PMA_KERNEL void k_scan_tilesums(uint32_t *tilesum, uint64_t ntiles, unsigned long long *total, ChainTable *tb,
                                uint64_t tb_index, uint64_t tb_len, uint32_t *order, uint32_t *ctl, uint32_t tile_slots, uint32_t defer_table) {
  // ONE workgroup of kTileSumThreads.  Each thread owns a contiguous run of tile sums (independent loads, all in
  // flight together); waves combine through LDS; the prefix is written back while one lane builds the rebalance's exact
  // position table from the grand total (saves a launch).
  PMA_SHARED uint32_t wtot[kTileSumThreads / 64];
  PMA_TSTAMP(total, 8);
  const int lane = wv::lane(), w = wv::wave_in_block();
  const uint64_t per = (ntiles + kTileSumThreads - 1) / kTileSumThreads;
  const uint64_t lo = (uint64_t)wv::thread_idx() * per;
  const uint64_t hi = (lo + per < ntiles) ? lo + per : ntiles;
  uint32_t mine = 0;
  for (uint64_t i = lo; i < hi; i++) mine += tilesum[i];
  uint32_t incl = mine;
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = wv::shfl(incl, lane - o < 0 ? 0 : lane - o);
    if (lane >= o) incl += y;
  }
  if (lane == 63) wtot[w] = incl;
  wv::block_sync();  // also: every read of the un-scanned sums is done before anyone overwrites them
  uint32_t woff = 0, grand = 0;
  for (int q = 0; q < (int)(kTileSumThreads / 64); q++) {
    const uint32_t x = wtot[q];
    if (q < w) woff += x;
    grand += x;
  }
  if (wv::thread_idx() == kTileSumThreads - 1) {  // (this thread's own run is the shortest or empty)
    // (one serial chain — a division and a true fp64 subtraction per binade the window crosses, 23 for a window that starts
    //  at slot 0: ~9 us of this launch; run by the whole wave on wave-uniform inputs it measured the same)
    *total = grand;
    if (tb && defer_table) {  // the consumer (k_rb_scatter) builds the table itself, behind its first tiles: only the header here
      tb->index = tb_index;
      tb->len = tb_len;
      tb->j = (uint64_t)grand;
      tb->nseg = 0;
      tb->overflow = 0;
      tb->pub_nseg = grand < 2u ? kTbDone : 0u;  // (j < 2: an empty table is complete)
      tb->pub_t = 0;
    } else if (tb) {
      build_chain_table(tb_index, tb_len, (uint64_t)grand, tb);
    }
  }
  uint32_t run = woff + incl - mine;
  for (uint64_t i = lo; i < hi; i++) {
    const uint32_t x = tilesum[i];
    tilesum[i] = run;
    run += x;
  }
  PMA_TSTAMP(total, 9);
  if (order != nullptr) {  // in-place window: the order in which its tiles may be taken (needs the scanned sums and the table)
    wv::block_sync();
    PMA_TSTAMP(total, 10);
    rb_order_body(tilesum, tb, (uint32_t)ntiles, tile_slots, order, ctl, total);
  }
}
PMA_KERNEL void k_scan_apply(const uint32_t *cnt, uint64_t nleaves, const uint32_t *tilesum, uint32_t *rank) {
  PMA_SHARED uint32_t wsum[4];
  const uint64_t b = wv::block_idx();
  const int lane = wv::lane(), w = wv::wave_in_block();
  uint32_t run = tilesum[b];
  for (uint32_t it = 0; it < kScanTile / 256; it++) {
    const uint64_t l = b * kScanTile + it * 256 + wv::thread_idx();
    const uint32_t x = (l < nleaves) ? cnt[l] : 0u;
    uint32_t incl = x;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = wv::shfl(incl, lane - o < 0 ? 0 : lane - o);
      if (lane >= o) incl += y;
    }
    if (lane == 63) wsum[w] = incl;
    wv::block_sync();
    uint32_t woff = 0;
    for (int q = 0; q < w; q++) woff += wsum[q];
    if (l < nleaves) rank[l] = run + woff + incl - x;
    run += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    wv::block_sync();
  }
}

// Fused rebalance scatter: every live element of src window [src_lo, src_lo+src_len) goes to dst[pos_k - dst_bias]
// where k = its rank among the live elements (rank[] = exclusive leaf prefix) and pos_k comes from the exact chain
// table; `v` carries the NEW geometry (n, N) for the sentinel back-pointers.  Every wave also writes the null slots that follow its elements
// (element k owns output slots [pos_k, pos_{k+1})), so the destination needs no separate fill pass, every output slot
// is written exactly once, and the wave's output stretch is staged in LDS and stored as one coalesced run.
// Leaf counts of the destination are accumulated with one atomicAdd per element (dst leafcnt must be zeroed first).
constexpr uint32_t kStageSlots = 384;  // LDS staging tile per wave (4.5 KB): 64 elements at step <= 6
PMA_KERNEL void k_scatter_fill(View v, const Edge *src, uint64_t src_lo, uint64_t src_len, int src_sh, const uint32_t *rank,
                               const ChainTable *tb, Edge *dst, uint64_t dst_bias, uint32_t *dst_leafcnt, int dst_sh,
                               uint64_t dst_leaf_bias) {
  PMA_SHARED ChainTable stb;
  PMA_SHARED uint32_t stage[4][3 * kStageSlots];
  {
    const uint32_t *g = reinterpret_cast<const uint32_t *>(tb);
    uint32_t *s = reinterpret_cast<uint32_t *>(&stb);
    for (uint32_t i = wv::thread_idx(); i < sizeof(ChainTable) / 4; i += wv::block_dim()) s[i] = g[i];
  }
  wv::block_sync();
  const int lane = wv::lane();
  uint32_t *ls = stage[wv::wave_in_block()], *ld = ls + kStageSlots, *lv = ls + 2 * kStageSlots;
  const uint32_t slogN = 1u << src_sh;
  const uint64_t j = stb.j;
  const uint64_t wend = stb.index + stb.len;  // end of the destination window (absolute slot)
  const uint64_t nchunks = (src_len + 63) / 64;
  const uint64_t wstride = (uint64_t)wv::grid_dim() * (wv::block_dim() >> 6);
  if (j == 0) {  // empty window: nothing owns the output slots, null them all
    const uint64_t tstride = (uint64_t)wv::grid_dim() * wv::block_dim();
    for (uint64_t t = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); t < stb.len; t += tstride)
      dst[stb.index + t - dst_bias] = null_edge();
    return;
  }
  int hint = -1, hint2 = -1;
  for (uint64_t ch = (uint64_t)wv::block_idx() * (wv::block_dim() >> 6) + wv::wave_in_block(); ch < nchunks; ch += wstride) {
    const uint64_t off = ch * 64 + (uint64_t)lane;
    Edge e = null_edge();
    if (off < src_len) e = src[src_lo + off];
    const bool nn = e.value != 0;
    const uint64_t m = wv::ballot(nn);
    if (m == 0) continue;
    uint64_t pos = 0, nxt = 0;
    if (nn) {
      const uint64_t lleaf = off >> src_sh;
      uint64_t lmask;
      if (slogN >= 64) {
        lmask = ~0ull;
      } else {
        const uint32_t first = (uint32_t)(lane & ~(int)(slogN - 1));
        lmask = ((1ull << slogN) - 1ull) << first;
      }
      const uint64_t k = (uint64_t)rank[lleaf] + (uint64_t)wv::popc64(m & lmask & ((1ull << lane) - 1ull));
      pos = chain_pos(&stb, k, &hint);
      nxt = (k + 1 < j) ? chain_pos(&stb, k + 1, &hint2) : wend;
      dev::fix_sentinel(v, e, (uint32_t)pos);
    }
    // output stretch of this chunk: [first element's pos, last element's nxt)
    const int lfirst = wv::ctz64(m), llast = 63 - __builtin_clzll(m);
    const uint64_t o_lo = ((uint64_t)wv::shfl((uint32_t)(pos >> 32), lfirst) << 32) | wv::shfl((uint32_t)pos, lfirst);
    const uint64_t o_hi = ((uint64_t)wv::shfl((uint32_t)(nxt >> 32), llast) << 32) | wv::shfl((uint32_t)nxt, llast);
    const uint64_t p_hi = ((uint64_t)wv::shfl((uint32_t)(pos >> 32), llast) << 32) | wv::shfl((uint32_t)pos, llast);
    // destination leaf counts: the chunk's elements land in a handful of consecutive leaves -> one atomic per leaf
    {
      const uint64_t l0 = o_lo >> dst_sh, l1 = p_hi >> dst_sh;
      const uint64_t mylf = pos >> dst_sh;
      for (uint64_t L = l0; L <= l1; L++) {
        const uint64_t mm = wv::ballot(nn && mylf == L);
        if (mm && lane == 0) wv::atomic_add_u32(&dst_leafcnt[L - dst_leaf_bias], (uint32_t)wv::popc64(mm));
      }
    }
    const uint64_t olen = o_hi - o_lo;
    if (olen <= kStageSlots) {
      for (uint32_t t = (uint32_t)lane; t < (uint32_t)olen; t += 64) {
        ls[t] = kMax;
        ld[t] = 0;
        lv[t] = 0;
      }
      wv::lds_fence();
      if (nn) {
        const uint32_t t = (uint32_t)(pos - o_lo);
        ls[t] = e.src;
        ld[t] = e.dest;
        lv[t] = e.value;
      }
      wv::lds_fence();
      for (uint32_t t = (uint32_t)lane; t < (uint32_t)olen; t += 64) {
        Edge o;
        o.src = ls[t];
        o.dest = ld[t];
        o.value = lv[t];
        dst[o_lo + t - dst_bias] = o;
      }
      wv::lds_fence();
    } else if (nn) {  // very sparse destination: each lane writes its own run
      dst[pos - dst_bias] = e;
      for (uint64_t s2 = pos + 1; s2 < nxt; s2++) dst[s2 - dst_bias] = null_edge();
    }
  }
}

// Leaner variant of the fused rebalance pass: ONE position-table look-up per wave (its <= 64 live elements are
// consecutive ranks and almost always lie on one arithmetic progression of the table: pos_i = (A + i*D) >> shift),
// and every lane stores its own run — the element followed by the null slots up to the next element's position —
// straight from registers.  Consecutive lanes write consecutive runs, so a wave's stores cover one contiguous stretch.
PMA_KERNEL void k_scatter_runs(View v, const Edge *src, uint64_t src_lo, uint64_t src_len, int src_sh, const uint32_t *rank,
                               const ChainTable *tb, Edge *dst, uint64_t dst_bias, uint32_t *dst_leafcnt, int dst_sh,
                               uint64_t dst_leaf_bias) {
  PMA_SHARED ChainTable stb;
  {  // only the segments in use are copied (a table has <= ~30 of its 128 slots filled)
    const uint32_t *g = reinterpret_cast<const uint32_t *>(tb);
    uint32_t *s = reinterpret_cast<uint32_t *>(&stb);
    const uint32_t words = (uint32_t)((sizeof(ChainTable) - sizeof(ChainSeg) * (size_t)(kMaxSeg - tb->nseg)) / 4);
    for (uint32_t i = wv::thread_idx(); i < words; i += wv::block_dim()) s[i] = g[i];
  }
  wv::block_sync();
  const int lane = wv::lane();
  const uint64_t j = stb.j;
  const uint64_t wend = stb.index + stb.len;
  const uint64_t nchunks = (src_len + 63) / 64;
  const uint64_t wstride = (uint64_t)wv::grid_dim() * (wv::block_dim() >> 6);
  if (j == 0) {
    const uint64_t tstride = (uint64_t)wv::grid_dim() * wv::block_dim();
    for (uint64_t t = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); t < stb.len; t += tstride)
      dst[stb.index + t - dst_bias] = null_edge();
    return;
  }
  int hint = -1, hint2 = -1, hint3 = -1;
  for (uint64_t ch = (uint64_t)wv::block_idx() * (wv::block_dim() >> 6) + wv::wave_in_block(); ch < nchunks; ch += wstride) {
    const uint64_t off = ch * 64 + (uint64_t)lane;
    Edge e = null_edge();
    if (off < src_len) e = src[src_lo + off];
    const uint64_t k0 = rank[(ch * 64) >> src_sh];  // live elements before this (leaf-aligned) chunk
    const bool nn = e.value != 0;
    const uint64_t m = wv::ballot(nn);
    if (m == 0) continue;
    const uint32_t cnt = (uint32_t)wv::popc64(m);
    const uint32_t i = dev::lanemask_lt_count(m, lane);
    uint64_t A, D;
    int shift;
    uint64_t pos = 0, nxt = 0;
    if (chain_linear_run(&stb, k0, (k0 + cnt <= j - 1) ? cnt : cnt - 1, &hint3, &A, &D, &shift)) {
      pos = (A + (uint64_t)i * D) >> shift;
      nxt = (k0 + i + 1 < j) ? ((A + (uint64_t)(i + 1) * D) >> shift) : wend;
    } else if (nn) {
      pos = chain_pos(&stb, k0 + i, &hint);
      nxt = (k0 + i + 1 < j) ? chain_pos(&stb, k0 + i + 1, &hint2) : wend;
    }
    if (nn) {
      dst[pos - dst_bias] = e;
      for (uint64_t s2 = pos + 1; s2 < nxt; s2++) dst[s2 - dst_bias] = null_edge();
      dev::fix_sentinel(v, e, (uint32_t)pos);
    }
    // destination leaf counts: one atomic per leaf touched by this wave
    const int lfirst = wv::ctz64(m), llast = 63 - __builtin_clzll(m);
    const uint64_t l0 = (((uint64_t)wv::shfl((uint32_t)(pos >> 32), lfirst) << 32) | wv::shfl((uint32_t)pos, lfirst)) >> dst_sh;
    const uint64_t l1 = (((uint64_t)wv::shfl((uint32_t)(pos >> 32), llast) << 32) | wv::shfl((uint32_t)pos, llast)) >> dst_sh;
    const uint64_t mylf = pos >> dst_sh;
    for (uint64_t L = l0; L <= l1; L++) {
      const uint64_t mm = wv::ballot(nn && mylf == L);
      if (mm && lane == 0) wv::atomic_add_u32(&dst_leafcnt[L - dst_leaf_bias], (uint32_t)wv::popc64(mm));
    }
  }
}

// ---- rebalance with the leaf-rank scan folded in --------------------------------------------------------------------
// Tile = kRbTile source leaves per workgroup.  k_rb_tilesums: per-tile live counts (+ zeroing of the destination leaf
// counts as a side job); k_scan_tilesums: exclusive scan of the tile sums + the exact position table; k_rb_scatter: each
// workgroup scans its own tile's leaf counts in LDS (so no per-leaf rank array is ever written or read) and runs the
// register-run scatter of k_scatter_runs over the tile's chunks.  Three launches for a whole-array rebalance.
constexpr uint32_t kRbTile = 256;  // maximum tile (= workgroup size); the engine picks a power of two <= this per window
PMA_KERNEL void k_rb_tilesums(uint32_t *cnt, uint64_t nleaves, uint32_t tile_leaves, uint32_t *tilesum, uint32_t *copy_out,
                              uint32_t *zero_ptr, uint64_t zero_n, uint32_t *dirty, uint32_t serial) {
  PMA_SHARED uint32_t red[4];
  const uint64_t b = wv::block_idx();
  const uint64_t l = b * tile_leaves + wv::thread_idx();
  const bool mine = wv::thread_idx() < tile_leaves && l < nleaves;
  uint32_t s = mine ? cnt[l] : 0u;
  if (dirty != nullptr && mine) dirty[l] = serial;  // (a window of the live array is about to be rewritten: dirty tags)
  if (copy_out != nullptr && mine) {  // in-place window: park the source counts, clear them for the rebuild
    copy_out[l] = s;
    cnt[l] = 0u;
  }
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t i = b * wv::block_dim() + wv::thread_idx(); i < zero_n; i += stride) zero_ptr[i] = 0u;
  s = wv::reduce_add(s);
  if (wv::lane() == 0) red[wv::wave_in_block()] = s;
  wv::block_sync();
  if (wv::thread_idx() == 0) tilesum[b] = red[0] + red[1] + red[2] + red[3];
}
// One chunk (64 source slots, already in registers) of the rebalance scatter: rank the live slots, look their exact
// positions up, store element + trailing nulls, fix sentinels, add the destination leaf counts.
// The exact position table of a window that starts at slot 0 is a SERIAL chain — a division and a true fp64 subtraction per binade
// the window crosses, 23 for 2^24 slots: ~10 us on one lane, which the whole-array rebalance used to spend in front of its
// scatter launch.  Its first segment covers the top HALF of the positions, the second the next quarter ...: with the tiles taken
// from the top of the window down, a builder thread inside the scatter launch (workgroup 0) stays ahead of them.  Every
// finished segment is published — its six words and then the count, at agent scope, word by word: plain stores may sit in the
// builder's XCD L2 — and a workgroup copies the table to its LDS once the published part covers the chain steps its tile needs.
// A workgroup that waits too long (it never does: workgroup 0 is dispatched first) builds the table itself: no hang, no error path.
PMA_DEV void build_chain_table_publish(ChainTable *tb) {  // one thread; index / len / j are in the header already
  const uint64_t index = tb->index, len = tb->len, j = tb->j;
  if (j < 2) return;
  int nseg = 0;
  const double step = chain_step(len, j);
  double x = chain_top(index, j, step);
  const uint64_t sb = dbl_bits(step);
  const int es = (int)((sb >> 52) & 0x7FF) - 1023;
  const uint64_t S = (sb & 0xFFFFFFFFFFFFFull) | (1ull << 52);
  const uint64_t T = j - 2;
  uint64_t t = 0;
  for (;;) {
    ChainSeg sg;
    sg.t0 = t;
    uint64_t c = chain_segment(x, S, es, &sg);
    if (c > T - t) c = T - t;
    sg.count = c;
    unsigned long long *out = reinterpret_cast<unsigned long long *>(&tb->seg[nseg]);
    const unsigned long long *in = reinterpret_cast<const unsigned long long *>(&sg);
#pragma unroll
    for (int q = 0; q < (int)(sizeof(ChainSeg) / 8); q++) wv::store_agent_u64(out + q, in[q]);
    nseg++;
    t += c;
    const bool done = t >= T || nseg >= kMaxSeg;
    wv::store_agent_u64(reinterpret_cast<unsigned long long *>(&tb->pub_nseg),
                        (unsigned long long)((uint32_t)nseg | (done ? kTbDone : 0u)) | ((unsigned long long)(uint32_t)(t + 1) << 32));
    if (done) return;
    // one true fp64 subtraction across the binade boundary
    const uint64_t Mc = (c == 0) ? sg.M0 : (sg.M0 - sg.Dfirst - (c - 1) * sg.Drest);
    const int e = 52 - sg.shift;
    const double xc = bits_dbl(((uint64_t)(e + 1023) << 52) | (Mc & 0xFFFFFFFFFFFFFull));
    x = chain_sub(xc, step);
    t += 1;
  }
}
static_assert(sizeof(ChainSeg) % 8 == 0 && offsetof(ChainTable, pub_t) == offsetof(ChainTable, pub_nseg) + 4 && offsetof(ChainTable, pub_nseg) % 8 == 0,
              "the publication word is one aligned 64-bit store: count in the low half, covered steps in the high half");
// the table in LDS (stb) for a workgroup whose tile needs the chain steps up to t_need; all threads of the workgroup call
PMA_DEV void rb_table_to_lds(ChainTable *tb, ChainTable *stb, uint64_t t_need, bool builder, uint32_t *s_flag) {
  if (builder && wv::thread_idx() == 0) build_chain_table_publish(tb);
  if (wv::thread_idx() == 0) {
    uint32_t n = 0;
    bool ok = false;
    for (uint32_t spin = 0; spin < 200000u; spin++) {
      const unsigned long long w = wv::load_agent_u64(reinterpret_cast<const unsigned long long *>(&tb->pub_nseg));
      n = (uint32_t)w;
      if ((n & kTbDone) || (uint64_t)(uint32_t)(w >> 32) > t_need) {
        ok = true;
        break;
      }
      wv::spin_pause();
    }
    *s_flag = ok ? (n & ~kTbDone) : 0xFFFFFFFFu;
  }
  wv::block_sync();
  const uint32_t n = *s_flag;
  if (n == 0xFFFFFFFFu) {  // (never, see above) build it here
    if (wv::thread_idx() == 0) build_chain_table(tb->index, tb->len, tb->j, stb);
    wv::block_sync();
    return;
  }
  if (wv::thread_idx() == 0) {
    stb->index = tb->index;
    stb->len = tb->len;
    stb->j = tb->j;
    stb->nseg = (int)n;
    stb->overflow = 0;
  }
  const uint32_t words = n * (uint32_t)(sizeof(ChainSeg) / 8);
  const unsigned long long *g = reinterpret_cast<const unsigned long long *>(&tb->seg[0]);
  unsigned long long *sp = reinterpret_cast<unsigned long long *>(&stb->seg[0]);
  for (uint32_t i = wv::thread_idx(); i < words; i += wv::block_dim()) sp[i] = wv::load_agent_u64(g + i);
  wv::block_sync();
}

PMA_DEV void rb_scatter_chunk(const View &v, const Edge &e, uint64_t k0 /* wave-uniform */, const ChainTable *stb, uint64_t j, uint64_t wend,
                              Edge *__restrict__ dst, uint64_t dst_bias, uint32_t *dst_leafcnt, int dst_sh, uint64_t dst_leaf_bias,
                              int lane, uint64_t lt_mask, int *hint, int *hint2, int *hint3) {
  const bool nn = e.value != 0;
  const uint64_t m = wv::ballot(nn);
  if (m == 0) return;
  k0 = wv::uni(k0);  // (the chunk's first rank: one value per wave, as are the table look-ups that depend on it alone)
  const uint32_t cn = (uint32_t)wv::popc64(m);
  const uint64_t below = m & lt_mask;
  const uint32_t i = (uint32_t)wv::popc64(below);
  uint64_t A, D;
  int shift;
  uint64_t pos = 0, nxt = 0;
  if (chain_linear_run(stb, k0, (k0 + cn <= j - 1) ? cn : cn - 1, hint3, &A, &D, &shift)) {
    const uint64_t M = A + (uint64_t)i * D;
    pos = M >> shift;
    nxt = (k0 + i + 1 < j) ? ((M + D) >> shift) : wend;
  } else if (nn) {
    pos = chain_pos(stb, k0 + i, hint);
    nxt = (k0 + i + 1 < j) ? chain_pos(stb, k0 + i + 1, hint2) : wend;
  }
  if (nn) {
    dst[pos - dst_bias] = e;
    for (uint64_t s2 = pos + 1; s2 < nxt; s2++) dst[s2 - dst_bias] = null_edge();
    dev::fix_sentinel(v, e, (uint32_t)pos);
  }
  // destination leaf counts: the first live lane of every destination leaf adds that leaf's share of this chunk
  const uint32_t mylf = (uint32_t)(pos >> dst_sh);
  const int prevlane = below ? 63 - __builtin_clzll(below) : lane;
  const uint32_t prevlf = wv::shfl(mylf, prevlane);
  const bool head = nn && (below == 0 || prevlf != mylf);
  const uint64_t hm = wv::ballot(head);
  if (head) {
    const uint64_t later_heads = hm & ~lt_mask & ~(1ull << lane);
    const uint64_t upto = later_heads ? ((1ull << wv::ctz64(later_heads)) - 1ull) : ~0ull;
    wv::atomic_add_u32(&dst_leafcnt[(uint64_t)mylf - dst_leaf_bias], (uint32_t)wv::popc64(m & upto & ~lt_mask));
  }
}

// kRbBatch chunks are requested back to back before the first one is processed: the kernel is bound by memory latency
// per wave (load -> rank -> store -> store acknowledgement), so bytes in flight per wave are what buys bandwidth.
constexpr int kRbBatch = 4;
PMA_KERNEL void k_rb_scatter(View v, const Edge *__restrict__ src, uint64_t src_lo, uint64_t src_len, int src_sh,
                             const uint32_t *__restrict__ cnt, uint32_t tile_leaves, uint32_t batch,
                             const uint32_t *__restrict__ tile_excl, ChainTable *tb, Edge *__restrict__ dst, uint64_t dst_bias,
                             uint32_t *dst_leafcnt, int dst_sh, uint64_t dst_leaf_bias, uint32_t defer_table) {
  PMA_SHARED ChainTable stb;
  PMA_SHARED uint32_t pre[kRbTile];
  PMA_SHARED uint32_t wsum[4];
  PMA_SHARED uint32_t s_flag;
  // defer_table: the table is built inside this launch by workgroup 0 (build_chain_table_publish) and the tiles are taken from
  // the top of the window down — the high tiles need only the first segments
  const uint64_t tile = defer_table ? (uint64_t)wv::grid_dim() - 1ull - wv::block_idx() : (uint64_t)wv::block_idx();
  if (defer_table) {
    const uint64_t jj = tb->j, r0 = tile_excl[tile];
    rb_table_to_lds(tb, &stb, jj > r0 ? jj - 1ull - r0 : 0ull, wv::block_idx() == 0, &s_flag);
  } else {
    const uint32_t *g = reinterpret_cast<const uint32_t *>(tb);
    uint32_t *sp = reinterpret_cast<uint32_t *>(&stb);
    const uint32_t words = (uint32_t)((sizeof(ChainTable) - sizeof(ChainSeg) * (size_t)(kMaxSeg - tb->nseg)) / 4);
    for (uint32_t i = wv::thread_idx(); i < words; i += wv::block_dim()) sp[i] = g[i];
  }
  const int lane = wv::lane(), w = wv::wave_in_block();
  const uint64_t nleaves = src_len >> src_sh;
  {  // exclusive prefix of this tile's leaf counts (one leaf per thread)
    const uint64_t l = tile * tile_leaves + wv::thread_idx();
    const uint32_t x = (wv::thread_idx() < tile_leaves && l < nleaves) ? cnt[l] : 0u;
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
  const uint64_t j = stb.j;
  const uint64_t wend = stb.index + stb.len;
  if (j == 0) {
    const uint64_t tstride = (uint64_t)wv::grid_dim() * wv::block_dim();
    for (uint64_t t = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); t < stb.len; t += tstride)
      dst[stb.index + t - dst_bias] = null_edge();
    return;
  }
  const uint64_t base_rank = tile_excl[tile];
  const uint32_t lpc = 64u >> src_sh;               // leaves per 64-slot chunk (logN <= 32)
  const uint32_t chunks = tile_leaves / lpc;        // chunks in this tile
  const uint64_t tile_slot0 = (tile * tile_leaves) << src_sh;
  const uint64_t lt_mask = (1ull << lane) - 1ull;   // lanes below this one
  int hint = -1, hint2 = -1, hint3 = -1;
  if (batch >= (uint32_t)kRbBatch) {
    for (uint32_t c0 = (uint32_t)w * kRbBatch; c0 < chunks; c0 += 4 * kRbBatch) {
      Edge e[kRbBatch];
#pragma unroll
      for (int q = 0; q < kRbBatch; q++) {
        const uint64_t off = tile_slot0 + (uint64_t)(c0 + q) * 64 + (uint64_t)lane;
        e[q] = null_edge();
        if (c0 + q < chunks && off < src_len) e[q] = dev::load_stream(src + (src_lo + off));
      }
#pragma unroll
      for (int q = 0; q < kRbBatch; q++) {
        if (c0 + q < chunks)
          rb_scatter_chunk(v, e[q], base_rank + pre[(c0 + q) * lpc], &stb, j, wend, dst, dst_bias, dst_leafcnt, dst_sh, dst_leaf_bias,
                           lane, lt_mask, &hint, &hint2, &hint3);
      }
    }
  } else {
    for (uint32_t c = (uint32_t)w; c < chunks; c += 4) {
      const uint64_t off = tile_slot0 + (uint64_t)c * 64 + (uint64_t)lane;
      if (off - lane >= src_len) break;
      Edge e = null_edge();
      if (off < src_len) e = dev::load_stream(src + (src_lo + off));
      rb_scatter_chunk(v, e, base_rank + pre[c * lpc], &stb, j, wend, dst, dst_bias, dst_leafcnt, dst_sh, dst_leaf_bias, lane, lt_mask,
                       &hint, &hint2, &hint3);
    }
  }
}

constexpr uint32_t kIpSpinLimit = 1u << 22;
template <int CPW>  // chunks (64 slots) per wave: the tile is 4 * CPW * 64 slots
PMA_DEV void rb_inplace_body(const View &v, uint64_t wstart, uint64_t wlen, int sh, const uint32_t *__restrict__ cnt,
                             const uint32_t *__restrict__ tile_excl, const ChainTable *tb, const uint32_t *__restrict__ order, uint32_t *ctl,
                             uint32_t *flags, uint32_t epoch, uint32_t nlists) {
  PMA_SHARED ChainTable stb;
  PMA_SHARED uint32_t pre[kRbTile];
  PMA_SHARED uint32_t wsum[4];
  PMA_SHARED uint32_t s_tile;
  constexpr uint32_t kTileSlots = 4u * CPW * 64u;
  if (wv::thread_idx() == 0) {
    // Ticket: position t * L + x of the order, drawn from the counter of this workgroup's XCD x (one counter for all would
    // hand out ~one ticket per 9 ns: same-address atomics are served one after the other).  Each of the L sub-lists is
    // consumed in order, and a workgroup turns to another XCD's list only when its own is used up, so the earliest
    // unfinished tile of the order is always held by a resident workgroup or is the next ticket of an XCD with free slots.
    // L = nlists is the number of XCD ids the engine SAW workgroups run on when it was created (k_xcc_probe: 8 on an
    // MI355X in SPX mode; 1 — a single list, safe whatever the dispatcher does — if the ids were not 0..L-1 evenly).
    const uint32_t ntiles = (uint32_t)(wlen / kTileSlots), L = nlists, xcc = wv::xcc_id() % L;
    uint32_t pos = 0xFFFFFFFFu;
    for (uint32_t a = 0; a < L && pos == 0xFFFFFFFFu; a++) {
      const uint32_t x = (xcc + a) % L;
      const uint32_t have = x < ntiles ? (ntiles - x + L - 1u) / L : 0u;
      if (have == 0u) continue;
      const uint32_t t = wv::atomic_add_u32(&ctl[kIpTicketStride * (1u + x)], 1u);
      if (t < have) pos = t * L + x;
    }
    s_tile = pos == 0xFFFFFFFFu ? pos : order[pos];
  }
  wv::block_sync();
  const int lane = wv::lane(), w = wv::wave_in_block();
  const uint32_t tile = s_tile;
  if (tile == 0xFFFFFFFFu) return;  // (more workgroups than tiles: cannot happen with the engine's launch)
  const uint64_t tile_slot0 = (uint64_t)tile * kTileSlots;
  Edge e[CPW];  // requested first: everything below overlaps with these loads
#pragma unroll
  for (int q = 0; q < CPW; q++) {
    // (the window is a whole number of tiles — the engine checks — so the load needs no guard; a load under a branch makes the
    //  compiler wait for it on the spot, which turned these CPW requests into CPW / 2 round trips)
    const uint64_t off = tile_slot0 + (uint64_t)(w * CPW + q) * 64u + (uint64_t)lane;
    e[q] = dev::load_stream(v.items + (wstart + off));
  }
  {
    const uint32_t *g = reinterpret_cast<const uint32_t *>(tb);
    uint32_t *sp = reinterpret_cast<uint32_t *>(&stb);
    const uint32_t words = (uint32_t)((sizeof(ChainTable) - sizeof(ChainSeg) * (size_t)(kMaxSeg - tb->nseg)) / 4);
    for (uint32_t i = wv::thread_idx(); i < words; i += 256u /* the launch's workgroup size: blockDim.x would be a load + a wait for everything in flight */) sp[i] = g[i];
  }
  const uint32_t tile_leaves = kTileSlots >> sh;
  const uint64_t nleaves = wlen >> sh;
  uint32_t tile_cnt;
  {  // exclusive prefix of this tile's (parked) leaf counts, one leaf per thread
    const uint64_t l = (uint64_t)tile * tile_leaves + wv::thread_idx();
    const uint32_t x = (wv::thread_idx() < tile_leaves && l < nleaves) ? cnt[l] : 0u;
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
    tile_cnt = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
  const uint64_t j = stb.j, wend = stb.index + stb.len;
  wv::wait_loads();  // EVERY wave's tile loads have returned before the barrier that precedes "tile read" (a workgroup barrier
                     // does not wait for vmcnt, and flag_publish's own wait covers wave 0 only)
  wv::block_sync();
  if (wv::thread_idx() == 0) wv::flag_publish(&flags[tile], epoch);
  if (j == 0) {  // empty window: nothing is read by anybody, every tile clears its own slots
    for (uint32_t t = wv::thread_idx(); t < kTileSlots; t += 256u /* the launch's workgroup size: blockDim.x would be a load + a wait for everything in flight */)
      if (tile_slot0 + t < wlen) v.items[wstart + tile_slot0 + t] = null_edge();
    return;
  }
  if (tile_cnt == 0) return;
  const uint64_t base_rank = tile_excl[tile];
  int hint = -1, hint2 = -1, hint3 = -1;
  if (w == 0) {  // wait for the tiles whose source slots [c, d) covers
    const uint64_t c = chain_pos(&stb, base_rank, &hint);
    const uint64_t d = base_rank + tile_cnt < j ? chain_pos(&stb, base_rank + tile_cnt, &hint2) : wend;
    const uint32_t lo = (uint32_t)((c - wstart) / kTileSlots), hi = (uint32_t)((d - 1u - wstart) / kTileSlots);
    bool bad = false;
    for (uint32_t t0 = lo; t0 <= hi; t0 += 64u) {
      const uint32_t t = t0 + (uint32_t)lane;
      uint32_t spins = 0;
      while (wv::ballot(t <= hi && t != tile && wv::flag_read(&flags[t]) != epoch) != 0ull) {
        if (++spins > kIpSpinLimit) {
          bad = true;
          break;
        }
        wv::spin_pause();
      }
    }
    if (bad && lane == 0) ctl[1] = 1u;
    wv::flag_acquire();
  }
  wv::block_sync();
  const uint32_t lpc = 64u >> sh;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int q = 0; q < CPW; q++) {
    const uint32_t c = (uint32_t)(w * CPW + q);
    if (tile_slot0 + (uint64_t)c * 64u < wlen)
      rb_scatter_chunk(v, e[q], base_rank + pre[c * lpc], &stb, j, wend, v.items, 0, v.leafcnt, v.g.sh, 0, lane, lt_mask, &hint, &hint2, &hint3);
  }
}
PMA_KERNEL void k_rb_inplace8(View v, uint64_t wstart, uint64_t wlen, int sh, const uint32_t *cnt, const uint32_t *tile_excl, const ChainTable *tb,
                              const uint32_t *order, uint32_t *ctl, uint32_t *flags, uint32_t epoch, uint32_t nlists) {
  rb_inplace_body<8>(v, wstart, wlen, sh, cnt, tile_excl, tb, order, ctl, flags, epoch, nlists);
}
// which XCD ids do workgroups of this device report, and how evenly?  (one atomic per workgroup into 8 counters)
PMA_KERNEL void k_xcc_probe(uint32_t *counts) {
  if (wv::thread_idx() == 0) wv::atomic_add_u32(&counts[wv::xcc_id() & 7u], 1u);
}
PMA_KERNEL void k_rb_inplace16(View v, uint64_t wstart, uint64_t wlen, int sh, const uint32_t *cnt, const uint32_t *tile_excl, const ChainTable *tb,
                               const uint32_t *order, uint32_t *ctl, uint32_t *flags, uint32_t epoch, uint32_t nlists) {
  rb_inplace_body<16>(v, wstart, wlen, sh, cnt, tile_excl, tb, order, ctl, flags, epoch, nlists);
}

// ---- incremental snapshots (dirty tags) -----------------------------------------------------------------------------
// A snapshot (rollback point of a speculative epoch, or the user's snapshot()) is a second copy of items / leaf counts /
// node records that is kept in step with the live state by copying only what was written since it was last synchronised:
// every writer stamps the leaves / node records it modifies with the engine's serial (View::ldirty / vdirty), and an entry
// is dirty for a snapshot synchronised at serial S when its tag is > S.  to_live = 0: live -> snapshot ("commit": the
// snapshot catches up); to_live = 1: snapshot -> live ("rollback"), and the entry is re-tagged `newtag` so that the OTHER
// snapshot sees it as written.  One wave scans 64 tags per trip and copies the dirty leaves logN slots per lane group.
PMA_KERNEL void k_snap_sync_leaves(Edge *live, uint32_t *live_cnt, Edge *snap, uint32_t *snap_cnt, uint32_t *tag, uint64_t nleaves,
                                   int sh, uint32_t synced, uint32_t newtag, uint32_t to_live, unsigned long long *copied) {
  const int lane = wv::lane();
  const uint32_t logN = 1u << sh;
  const uint32_t G = logN >= 64u ? 1u : (64u >> sh);  // leaves copied per trip
  const uint64_t wstride = (uint64_t)wv::grid_dim() * (wv::block_dim() >> 6);
  unsigned long long mine = 0;
  for (uint64_t base = ((uint64_t)wv::block_idx() * (wv::block_dim() >> 6) + wv::wave_in_block()) * 64u; base < nleaves; base += wstride * 64u) {
    const uint64_t l = base + (uint64_t)lane;
    const uint32_t t = l < nleaves ? tag[l] : 0u;
    uint64_t m = wv::ballot(l < nleaves && t > synced);
    mine += (unsigned long long)wv::popc64(m);
    while (m) {
      uint64_t myleaf = ~0ull;
      for (uint32_t gI = 0; gI < G && m; gI++) {
        const int b = wv::ctz64(m);
        m &= m - 1ull;
        if (((uint32_t)lane >> sh) == gI || logN >= 64u) myleaf = base + (uint64_t)b;
      }
      if (myleaf != ~0ull) {
        const uint32_t q = (uint32_t)lane & (logN - 1u);
        for (uint32_t o = q; o < logN; o += 64u) {  // (logN <= 64: one trip)
          const uint64_t slot = (myleaf << sh) + o;
          if (to_live) live[slot] = snap[slot]; else snap[slot] = live[slot];
        }
        if (q == 0) {
          if (to_live) {
            live_cnt[myleaf] = snap_cnt[myleaf];
            tag[myleaf] = newtag;
          } else {
            snap_cnt[myleaf] = live_cnt[myleaf];
          }
        }
      }
    }
  }
  if (copied != nullptr && lane == 0 && mine) wv::atomic_add_u64(copied, mine);
}
PMA_KERNEL void k_snap_sync_nodes(Node *live, Node *snap, uint32_t *tag, uint64_t n, uint32_t synced, uint32_t newtag, uint32_t to_live) {
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t u = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); u < n; u += stride) {
    if (tag[u] > synced) {
      if (to_live) {
        live[u] = snap[u];
        tag[u] = newtag;
      } else {
        snap[u] = live[u];
      }
    }
  }
}
PMA_KERNEL void k_fill_u32(uint32_t *p, uint64_t n, uint32_t value) {
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t i = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); i < n; i += stride) p[i] = value;
}

// pppcsr_repartition: num_neighbors travels beside the edges (it is a counter of calls, not the degree: duplicate adds and
// deletes of missing edges move it, PCSR.cpp:1380/1409).  One record (vertex + base, num_neighbors, 1) per vertex out, and
// the setter for the records a partition receives (vertex partition-local again after the routing).
PMA_KERNEL void k_nn_export(View v, uint32_t base, Op *out) {
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t i = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); i < v.g.n; i += stride)
    out[i] = Op{(uint32_t)i + base, v.nodes[i].num_neighbors, 1u};
}
PMA_KERNEL void k_nn_set(View v, const Op *recs, uint64_t n) {
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t i = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); i < n; i += stride) {
    const Op r = recs[i];
    if (r.src < v.g.n) {
      v.nodes[r.src].num_neighbors = r.dst;
      v.vdirty[r.src] = v.serial;
    }
  }
}

PMA_KERNEL void k_copy_slots(const Edge *src, Edge *dst, uint64_t len) {
  const uint32_t *s = reinterpret_cast<const uint32_t *>(src);
  uint32_t *d = reinterpret_cast<uint32_t *>(dst);
  const uint64_t total = len * 3ull;
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t i = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); i < total; i += stride) d[i] = s[i];
}

// Are the vertex ranges still sorted, disjoint and consistent with nodes[]?  One wave per vertex: the node record
// (beginning / end chain, the sentinel on `beginning`) and every slot of (beginning, end): live slots carry src == vertex,
// are no sentinels and have strictly ascending dests.  Run after the one event that can break this (add_node after a
// doubling, PCSR.cpp:533-540 + 681-703): if nothing is wrong, the 64-ary search narrowing and the parallel rounds are valid again.
PMA_KERNEL void k_check_ranges(View v, unsigned long long *bad) {
  const int lane = wv::lane();
  const uint64_t wstride = (uint64_t)wv::grid_dim() * (wv::block_dim() >> 6);
  const uint32_t n = v.g.n;
  for (uint64_t u = (uint64_t)wv::block_idx() * (wv::block_dim() >> 6) + wv::wave_in_block(); u < n; u += wstride) {
    const Node nd = v.nodes[u];
    bool wrong = false;
    const uint64_t want_end = (u + 1 < n) ? (uint64_t)v.nodes[u + 1].beginning : v.g.N - 1;
    if ((uint64_t)nd.beginning >= v.g.N || (uint64_t)nd.end != want_end || nd.end <= nd.beginning) wrong = true;
    if (!wrong) {
      const Edge sn = v.items[nd.beginning];
      if (sn.src != (uint32_t)u || sn.dest != kMax || sn.value != (u == 0 ? kMax : (uint32_t)u)) wrong = true;
    }
    if (!wrong) {
      uint32_t prev = 0;
      bool have_prev = false;
      for (uint64_t base = (uint64_t)nd.beginning + 1; base < nd.end; base += 64) {
        const uint64_t s = base + (uint64_t)lane;
        Edge e = null_edge();
        if (s < nd.end) e = v.items[s];
        const bool live = s < nd.end && e.value != 0;
        const uint64_t m = wv::ballot(live);
        if (live && (e.src != (uint32_t)u || is_sentinel(e))) wrong = true;
        const uint64_t below = m & ((1ull << lane) - 1ull);
        const int pl = below ? 63 - __builtin_clzll(below) : 0;
        const uint32_t pd = wv::shfl(e.dest, pl);
        if (live && (below ? !(pd < e.dest) : (have_prev && !(prev < e.dest)))) wrong = true;
        if (m) {
          prev = wv::shfl(e.dest, 63 - __builtin_clzll(m));
          have_prev = true;
        }
      }
    }
    if (wv::ballot(wrong) != 0 && lane == 0) wv::atomic_add_u64(bad, 1ull);
  }
}

}  // namespace ppcsr
