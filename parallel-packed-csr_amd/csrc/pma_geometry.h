// Geometry, integer density thresholds and the exact rebalance-position chain.
//
// Everything here is scalar host+device code with no memory access pattern of its own; it is the
// arithmetic contract with the reference:
//   * resizeEdgeArray            /root/reference/src/pcsr/PCSR.cpp:68-73
//   * density_bound              PCSR.cpp:156-165
//   * redistribute position loop PCSR.cpp:237-247  (serial fp64 chain index_d -= step)
#pragma once
#include <math.h>
#include <string.h>

#include "pma_types.h"

namespace ppcsr {

PMA_HD inline int bsr64(uint64_t w) {  // index of highest set bit (reference bsr_word, PCSR.cpp:28-33)
#if defined(__HIP_DEVICE_COMPILE__)
  return 63 - __clzll((long long)w);
#else
  return 63 - __builtin_clzll(w);
#endif
}

// ---- host only: geometry + thresholds ----------------------------------------------------------------
inline void compute_geometry(uint64_t N, uint32_t n, int lock_search, Geometry *g) {
  memset(g, 0, sizeof(*g));
  g->N = N;
  g->n = n;
  g->lock_search = lock_search;
  g->narrow = 1u;
  g->logN = 1 << bsr64((uint64_t)(bsr64(N) * 2 + 1));
  g->sh = bsr64((uint64_t)g->logN);
  g->H = bsr64(N / (uint64_t)g->logN);
  for (int L = 0; L <= g->H && L < kMaxLevels; L++) {
    const double len = (double)((uint64_t)g->logN << (g->H - L));
    // same expressions, same evaluation order as density_bound(): x = lower, y = upper
    volatile double lower = 1.0 / 4.0 - ((0.125 * L) / g->H);
    volatile double upper = 3.0 / 4.0 + ((.25 * L) / g->H);
    // t_up = min c with (double)c/len >= upper ; never if upper is NaN (H == 0)
    uint32_t tu = kNever;
    if (upper == upper) {
      double c = ceil(upper * len);
      if (c < 0) c = 0;
      while (c > 0 && ((c - 1) / len >= upper)) c -= 1;
      while (!(c / len >= upper)) c += 1;
      tu = (c > 4294967294.0) ? kNever : (uint32_t)c;
    }
    // t_lo = min c with !((double)c/len < lower) ; 0 if lower is NaN or <= 0
    uint32_t tl = 0;
    if (lower == lower && lower > 0) {
      double c = ceil(lower * len);
      while (c > 0 && !((c - 1) / len < lower)) c -= 1;
      while ((c / len < lower)) c += 1;
      tl = (uint32_t)c;
    }
    g->t_up[L] = tu;
    g->t_lo[L] = tl;
  }
}
inline uint64_t initial_N(uint32_t init_n, uint32_t src_n) {  // PCSR.cpp:777
  uint32_t m = init_n + src_n;
  if (m < 1024u) m = 1024u;
  return (uint64_t)2 << bsr64(m);
}

// ---- exact position chain as a piecewise-linear integer table ------------------------------------------
// The reference computes, for a window (index,len) holding j live elements,
//     step = (double)len / j;  x = index + (double)(j-1)*step;
//     for k = j-1 .. 1:  pos_k = (size_t)x;  x -= step;          pos_0 = index
// The subtraction chain accumulates fp64 rounding, so floor(index + k*step) is NOT equal to pos_k
// in general.  While x stays inside one binade [2^e,2^(e+1)) its value is an integer multiple M of
// u = 2^(e-52) and RN(x - step) is an integer decrement of M (constant after the first step), so the
// whole chain is a short list of arithmetic progressions; one true fp64 subtraction is executed at each
// binade crossing.  Random access to pos_k makes the rebalance embarrassingly parallel.
constexpr int kMaxSeg = 128;
struct ChainSeg {
  uint64_t t0;      // first chain step (t = j-1-k) covered
  uint64_t count;   // covers t0 .. t0+count
  uint64_t M0;      // integer mantissa at t0
  uint64_t Dfirst;  // decrement of the first step inside the segment
  uint64_t Drest;   // decrement of every later step
  int shift;        // pos = M >> shift   (shift = 52 - e >= 0)
  int pad;
};
struct ChainTable {
  uint64_t index, len, j;
  int nseg;
  int overflow;
  // a table built INSIDE the kernel that uses it (k_rb_scatter, see there): segments published so far (bit 31: the table is
  // complete) and the first chain step they do not cover yet
  uint32_t pub_nseg, pub_t;
  ChainSeg seg[kMaxSeg];
};
constexpr uint32_t kTbDone = 0x80000000u;

PMA_HD inline uint64_t dbl_bits(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint64_t)__double_as_longlong(x);
#else
  uint64_t b;
  memcpy(&b, &x, 8);
  return b;
#endif
}
PMA_HD inline double bits_dbl(uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __longlong_as_double((long long)b);
#else
  double x;
  memcpy(&x, &b, 8);
  return x;
#endif
}

// These three must round exactly once each (no FMA contraction): the translation unit is compiled
// with -ffp-contract=off and the pragma below guards against a caller's flags.
PMA_HD inline double chain_step(uint64_t len, uint64_t j) {
#pragma clang fp contract(off)
  return (double)len / (double)j;
}
PMA_HD inline double chain_top(uint64_t index, uint64_t j, double step) {
#pragma clang fp contract(off)
  double prod = (double)(j - 1) * step;
  return (double)index + prod;
}
PMA_HD inline double chain_sub(double x, double step) {
#pragma clang fp contract(off)
  return x - step;
}

// floor(a / b) for a, b < 2^53, b > 0: one fp64 division plus an exact integer fix-up (a 64-bit integer division is a
// ~100-instruction software routine on the GPU and this sits on the table build's serial path)
PMA_HD inline uint64_t div_floor_u53(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  // (the hardware reciprocal is an estimate; the fix-up below makes the quotient exact whatever the estimate is off by — an
  //  IEEE division is ~40 dependent instructions, and one thread runs this once per binade of the chain)
  uint64_t q = (uint64_t)((double)a * __builtin_amdgcn_rcp((double)b));
#else
  uint64_t q = (uint64_t)((double)a / (double)b);
#endif
  int64_t r = (int64_t)(a - q * b);
  while (r < 0) {
    q--;
    r += (int64_t)b;
  }
  while (r >= (int64_t)b) {
    q++;
    r -= (int64_t)b;
  }
  return q;
}

// One segment of the chain starting at value x (a chain value, i.e. positive and finite): fills M0 / shift / Dfirst /
// Drest and returns how many further steps (beyond the first value) stay on the segment's arithmetic progression.
// S / es: mantissa (with the hidden bit) and unbiased exponent of `step`.
PMA_HD inline uint64_t chain_segment(double x, uint64_t S, int es, ChainSeg *sg) {
  const uint64_t xb = dbl_bits(x);
  const int e = (int)((xb >> 52) & 0x7FF) - 1023;
  const uint64_t M0 = (xb & 0xFFFFFFFFFFFFFull) | (1ull << 52);
  sg->M0 = M0;
  sg->shift = 52 - e;
  sg->pad = 0;
  sg->Dfirst = sg->Drest = 0;
  uint64_t c = 0;
  const int r = e - es;
  if (sg->shift >= 0 && r >= 0 && r <= 52) {
    uint64_t q, rem, half;
    if (r == 0) {
      q = S;
      rem = 0;
      half = 1;
    } else {
      q = S >> r;
      rem = S & ((1ull << r) - 1);
      half = 1ull << (r - 1);
    }
    uint64_t Df, Dr;
    if (rem < half) {
      Df = Dr = q;
    } else if (rem > half) {
      Df = Dr = q + 1;
    } else {  // exact tie: round to even mantissa
      Df = (((M0 - q) & 1ull) == 0) ? q : q + 1;
      Dr = ((q & 1ull) == 0) ? q : q + 1;
    }
    const uint64_t Th = (1ull << 52) + q + (rem ? 1 : 0);
    if (M0 >= Th && Dr > 0) {
      const uint64_t M1 = M0 - Df;
      c = 1;
      if (M1 >= Th) c += div_floor_u53(M1 - Th, Dr) + 1;
    }
    sg->Dfirst = Df;
    sg->Drest = Dr;
  }
  return c;
}

PMA_HD inline void build_chain_table(uint64_t index, uint64_t len, uint64_t j, ChainTable *tb) {
  tb->index = index;
  tb->len = len;
  tb->j = j;
  tb->nseg = 0;
  tb->overflow = 0;
  if (j < 2) return;
  int nseg = 0;  // kept in a register: the table usually lives in device memory and is only written, never read back
  const double step = chain_step(len, j);
  double x = chain_top(index, j, step);
  const uint64_t sb = dbl_bits(step);
  const int es = (int)((sb >> 52) & 0x7FF) - 1023;
  const uint64_t S = (sb & 0xFFFFFFFFFFFFFull) | (1ull << 52);
  const uint64_t T = j - 2;  // last chain step needed (k = 1)
  uint64_t t = 0;
  for (;;) {
    if (nseg >= kMaxSeg) {
      tb->nseg = nseg;
      tb->overflow = 1;
      return;
    }
    ChainSeg sg;
    sg.t0 = t;
    uint64_t c = chain_segment(x, S, es, &sg);
    if (c > T - t) c = T - t;
    sg.count = c;
    tb->seg[nseg++] = sg;
    t += c;
    if (t >= T) {
      tb->nseg = nseg;
      return;
    }
    // one true fp64 subtraction across the binade boundary
    const uint64_t Mc = (c == 0) ? sg.M0 : (sg.M0 - sg.Dfirst - (c - 1) * sg.Drest);
    const int e = 52 - sg.shift;
    double xc = bits_dbl(((uint64_t)(e + 1023) << 52) | (Mc & 0xFFFFFFFFFFFFFull));
    x = chain_sub(xc, step);
    t += 1;
  }
}

// Whole chain in ONE segment?  True for every window that does not straddle a power of two — i.e. all aligned PMA
// windows except those starting at slot 0 — because every chain value then lies in [index, index+len) inside one
// binade.  On success pos_k = chain_single_pos(sg, index, j, k) for 0 <= k < j, with no serial dependency.
// (chain_single_div: the same through chain_segment's step count — one division; kept as the statement the tests hold the
//  division-free form below against)
PMA_HD inline bool chain_single_div(uint64_t index, uint64_t len, uint64_t j, ChainSeg *sg) {
  if (j < 2) return false;
  const double step = chain_step(len, j);
  const double x = chain_top(index, j, step);
  const uint64_t sb = dbl_bits(step);
  const int es = (int)((sb >> 52) & 0x7FF) - 1023;
  const uint64_t S = (sb & 0xFFFFFFFFFFFFFull) | (1ull << 52);
  sg->t0 = 0;
  const uint64_t c = chain_segment(x, S, es, sg);
  sg->count = c;
  return c >= j - 2;
}
// The segment's step count c = 1 + floor((M1 - Th) / Drest) + 1 is only ever compared with j - 2 here, and
// floor(a / d) >= n  <=>  a >= n * d: a multiplication instead of the division (this runs once per in-wave rebalance, on the
// scalar unit, with the update's wave waiting).  sg->count is the number of steps the caller needs (j - 2) on success.
PMA_HD inline bool chain_single(uint64_t index, uint64_t len, uint64_t j, ChainSeg *sg) {
  if (j < 2) return false;
  const double step = chain_step(len, j);
  const double x = chain_top(index, j, step);
  const uint64_t sb = dbl_bits(step);
  const int es = (int)((sb >> 52) & 0x7FF) - 1023;
  const uint64_t S = (sb & 0xFFFFFFFFFFFFFull) | (1ull << 52);
  const uint64_t xb = dbl_bits(x);
  const int e = (int)((xb >> 52) & 0x7FF) - 1023;
  const uint64_t M0 = (xb & 0xFFFFFFFFFFFFFull) | (1ull << 52);
  sg->t0 = 0;
  sg->M0 = M0;
  sg->shift = 52 - e;
  sg->pad = 0;
  sg->Dfirst = sg->Drest = 0;
  sg->count = 0;
  const uint64_t need = j - 2;  // chain steps beyond the first value
  const int r = e - es;
  if (!(sg->shift >= 0 && r >= 0 && r <= 52)) return need == 0;
  uint64_t q, rem, half;
  if (r == 0) {
    q = S;
    rem = 0;
    half = 1;
  } else {
    q = S >> r;
    rem = S & ((1ull << r) - 1);
    half = 1ull << (r - 1);
  }
  uint64_t Df, Dr;
  if (rem < half) {
    Df = Dr = q;
  } else if (rem > half) {
    Df = Dr = q + 1;
  } else {  // exact tie: round to even mantissa
    Df = (((M0 - q) & 1ull) == 0) ? q : q + 1;
    Dr = ((q & 1ull) == 0) ? q : q + 1;
  }
  sg->Dfirst = Df;
  sg->Drest = Dr;
  if (need == 0) return true;
  const uint64_t Th = (1ull << 52) + q + (rem ? 1 : 0);
  if (!(M0 >= Th && Dr > 0)) return false;  // c = 0
  if (need == 1) {
    sg->count = 1;
    return true;
  }
  const uint64_t M1 = M0 - Df;
  if (M1 < Th) return false;  // c = 1
  // c = floor((M1 - Th) / Dr) + 2 >= need  <=>  (need - 2) * Dr <= M1 - Th.  need < 2^32 (element counts), Dr < 2^54: the
  // product is formed from two partial products that cannot wrap, and compared with M1 - Th < 2^53
  const uint64_t a = M1 - Th, n = need - 2;
  if (n >> 32) return false;
  const uint64_t Dh = Dr >> 27, Dl = Dr & ((1ull << 27) - 1);
  const uint64_t ph = n * Dh;  // < 2^32 * 2^27
  const uint64_t pl = n * Dl;  // < 2^59
  if (ph >= (1ull << 27)) return false;  // (n * Dr >= 2^54 > a)
  if ((ph << 27) + pl > a) return false;
  sg->count = need;
  return true;
}
PMA_HD inline uint64_t chain_single_pos(const ChainSeg &sg, uint64_t index, uint64_t j, uint64_t k) {
  if (k == 0) return index;
  const uint64_t d = j - 1 - k;
  const uint64_t M = (d == 0) ? sg.M0 : (sg.M0 - sg.Dfirst - (d - 1) * sg.Drest);
  return M >> sg.shift;
}

// position of element k (0 <= k < j) from the table; *hint is a segment cursor (monotone callers)
// segment that covers chain step t: the last one with t0 <= t.  hint: the segment of the caller's previous lookup (its
// lookups move slowly), or -1 = none yet: bisect instead of walking up from segment 0 (a table has tens of segments and
// every step is a dependent LDS read)
PMA_HD inline int chain_seg_find(const ChainTable *tb, uint64_t t, int hint) {
  int s = hint;
  if (s < 0 || s >= tb->nseg) {
    int lo = 0, hi = tb->nseg - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (tb->seg[mid].t0 <= t) lo = mid; else hi = mid - 1;
    }
    return lo;
  }
  while (s > 0 && tb->seg[s].t0 > t) s--;
  while (s + 1 < tb->nseg && tb->seg[s + 1].t0 <= t) s++;
  return s;
}
PMA_HD inline uint64_t chain_pos(const ChainTable *tb, uint64_t k, int *hint) {
  if (k == 0) return tb->index;
  const uint64_t t = tb->j - 1 - k;
  const int s = chain_seg_find(tb, t, *hint);
  *hint = s;
  const ChainSeg &sg = tb->seg[s];
  const uint64_t d = t - sg.t0;
  const uint64_t M = (d == 0) ? sg.M0 : (sg.M0 - sg.Dfirst - (d - 1) * sg.Drest);
  return M >> sg.shift;
}


// Linear form of a run of consecutive elements: if elements k0 .. k0+cnt (cnt >= 0, k0 >= 1, k0+cnt <= j-1) all fall
// into ONE segment of the table, on its arithmetic-progression part, then pos_{k0+i} = (A + i*D) >> shift for
// 0 <= i <= cnt.  Returns false when the run straddles segments (callers fall back to chain_pos per element).
PMA_HD inline bool chain_linear_run(const ChainTable *tb, uint64_t k0, uint64_t cnt, int *hint, uint64_t *A, uint64_t *D, int *shift) {
  if (k0 == 0 || tb->j < 2 || k0 + cnt > tb->j - 1) return false;
  const uint64_t t_hi = tb->j - 1 - k0, t_lo = tb->j - 1 - (k0 + cnt);
  const int s = chain_seg_find(tb, t_lo, *hint);
  *hint = s;
  const ChainSeg &sg = tb->seg[s];
  if (t_hi > sg.t0 + sg.count) return false;  // the run continues in an earlier segment
  const uint64_t d_lo = t_lo - sg.t0;         // smallest offset inside the segment (element k0+cnt)
  if (d_lo == 0 && sg.Dfirst != sg.Drest) return false;
  const uint64_t d_hi = t_hi - sg.t0;         // offset of element k0  (>= 1 here unless the progression is uniform)
  // M(d) = M0 - Dfirst - (d-1)*Drest for d >= 1 (and for d = 0 when Dfirst == Drest)
  *A = sg.M0 - sg.Dfirst - (d_hi - 1) * sg.Drest;
  *D = sg.Drest;
  *shift = sg.shift;
  return true;
}

}  // namespace ppcsr
