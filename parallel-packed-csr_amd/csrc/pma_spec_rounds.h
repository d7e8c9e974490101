// Speculative rounds of the update scheduler (o_plan / o_check / o_apply, o_big for the queued windows, o_settle at the end of a chunk).
#pragma once
#include "pma_rounds.h"

namespace ppcsr {

// =====================================================================================================================
// Speculative rounds ("optimistic mode"): commit more than a strict prefix per round, validated, with rollback.
//
// A round plans the M lowest pending updates (the carry list of deferred updates, then fresh ones from the stream).
// Every plan reserves its write leaves (wres) and read leaves (rres) with atomicMin(stream index).  An update PASSES
// when no earlier pending update writes anything it reads or writes and no earlier pending update reads anything it
// writes.  Passing updates commit unless an earlier update of the same REGION (aligned block of 2^regshift leaves)
// failed this round — a per-region strict prefix, which keeps later updates from overtaking a deferred update inside
// the block where its footprint can still move (rebalance windows are aligned power-of-two blocks no larger than a
// region, so a deferred update's window cannot leave its region).
//
// Soundness does not rest on that heuristic: every committed update stamps the leaves it read (rstamp) and wrote
// (wstamp) with its stream index, and an update may only commit if no LATER update has already written a leaf it
// reads or writes, nor read a leaf it writes.  With that check the executed schedule is conflict-serialisable in
// stream order (every conflicting pair ran in index order), i.e. identical to the reference's sequential result.  A
// failed check raises `violation`: the host restores the epoch snapshot and replays the epoch with the strict prefix
// rounds above.  K_EXCL updates are barriers: nothing later commits until the exclusive executor has run them.
// =====================================================================================================================
// Round r's bookkeeping — how many of round r-1's updates are still pending, the next horizon, the adapted width, is the epoch
// done, does the lowest pending update need the exclusive executor — is a pure function of words that round r-1's launches left
// behind, so o_plan(r) evaluates it itself — wave 0 of every workgroup (round_begin: scalar arithmetic on words it loads with its
// other control words; the other waves get what they need through LDS) — and wave 0 of workgroup 0 alone records the outcome
// for the later launches and the host.  That replaced a launch of its own per round (o_compact: one workgroup compacting the
// deferred updates into a sorted carry list, ~10 us of every round).  Deferred updates now append THEMSELVES to the carry list
// (o_apply: one returning atomic per workgroup that has any, on the counter of the XCD it runs on), in no particular order: stream order is carried by the reservation keys, not by the slots, and the one
// thing the sorted list guaranteed — everything not planned is later than everything planned — holds because a round always plans
// the WHOLE carry list (the adapted width only limits how many fresh updates join it).
// Words written during round r and read during round r+1 are indexed r % 3 where round r+1's wave 0 must be able to reset the
// set of round r+2 while its other waves still read that of round r; by parity where nobody writes what is being read.
constexpr uint32_t kStripes = 8;  // carry sub-lists (one per XCD id: same-address returning atomics are served one after the other)
struct OptCtl {
  // by round parity (written by wave 0 of that round's o_plan, read by its o_check / o_apply and by the next round's o_plan)
  uint32_t hor[2];    // updates planned by the round
  uint32_t used[2];   // ... of which from the carry list (slots [0, used)); fresh ones take the slots behind
  uint32_t nf[2];     // stream index of the round's first fresh update
  uint32_t cur_h[2];  // adaptive round width in force (<= max_horizon): grows while most of a round commits, shrinks otherwise
  // by round % 3 (accumulated during the round; the set of round r+2 is reset by wave 0 of o_plan(r+1))
  uint32_t kept[3][kStripes];  // deferred updates appended to the carry list, per stripe
  uint32_t minkept[3];         // smallest stream index among them (kMax: none)
  unsigned long long gbar[3];  // keyed min index of a K_EXCL update in the horizon
  unsigned long long sbar[3];  // keyed min index of a SOFT barrier (a planned window close to the exclusive threshold): a word of
                               // its own — folded into gbar as key + 1 it was indistinguishable from a real K_EXCL key of update idx + 1
  uint32_t njobs[3];           // big-window rebalances queued by the round's o_apply
  uint32_t e1;  // end of the epoch (exclusive stream index)
  uint32_t violation, excl, done, error;
  uint32_t max_horizon, excl_idx;  // max_horizon: width of the launched grid (a round's horizon never exceeds it)
  uint32_t excl_later;             // 1: an update later than excl_idx has already been committed in this epoch
  uint32_t width_cap;              // upper bound of the adaptive width (the engine's opt_horizon)
  uint32_t resident;               // waves the chip holds at once (0 = unknown): above it the width moves in whole multiples
  uint32_t viol_idx;  // smallest stream index whose commit-time validation failed
  uint32_t adaptive;  // 1: adapt the width to the share of a round that commits (round_begin)
  uint32_t book_round;  // the last round whose outcome has been added to the counters below (o_settle / a relaunch must not add it twice)
  uint32_t need_big;        // 1: the last round queued big-window rebalances and no o_big launch followed it (the engine leaves that launch
                            // out while a stream queues none): the host runs it and goes on
  uint32_t big_done_round;  // the last round whose queued rebalances an o_big launch has looked at
  unsigned long long jobs_total;  // big-window rebalances queued so far in this epoch
  unsigned long long long_checks; // updates the lane-per-update o_check left to the wave (long footprints) so far in this epoch
  uint32_t skip_idx, skip_round;  // stream index the exclusive executor has just run inside this epoch: in round skip_round its slot commits as nothing
  unsigned long long rounds, committed, planned;
  uint32_t viol_info[8];  // debug: kind, leaf, stamp, what(1=wstamp on W,2=rstamp on W,3=wstamp on R), wleaf_lo, wleaf_hi, index, round
  // diagnostics (option "diag"): why planned updates did not commit, first reason found per update
  // 0 exclusive kind, 1 behind a barrier (gbar), 2 duplicate-slot conflicts, 3 write leaf reserved by an earlier writer,
  // 4 write leaf read by an earlier update, 5 read leaf written by an earlier update, 6 sentinel located by is moved earlier,
  // 7 sentinel we move is needed earlier, 8 region prefix, 9 growth zone of a deferred reader/writer (pfail), 10 stamp violation
  unsigned long long why[12];
};
struct OptArgs {
  View v;
  const Op *ops;
  Plan *plans;
  uint32_t *status, *vdbg;
  uint32_t *carry0, *carry1;  // by round parity: kStripes sub-lists of carry_cap entries, filled by that round's o_apply
  uint32_t carry_cap;
  OptCtl *ctl;
  StatShard *stats;
  unsigned long long *regfail;
  unsigned long long *pfail;  // per-leaf: smallest deferred update whose footprint may still grow over this leaf
  uint32_t *wstamp, *rstamp;
  uint32_t *vws, *vrs;  // per vertex: 1 + latest committed update that moved / read the position of its sentinel
  uint32_t round;
  int regshift;
  uint32_t diag;
  uint32_t dbg;  // diagnostics build only: bit 0 = o_plan plans every update twice, bit 1 = searches twice, bit 2 = evaluates the round's
                 // bookkeeping twice (the SQ counters of two runs then give the cost of that part: tools/sq_run.sh)
  uint32_t defer_barrier;  // 0: off
  uint32_t soft_barrier;   // slots: see o_plan
  // windows above big_min slots are rebalanced by a workgroup of o_big (job queue + one scratch stretch per workgroup)
  uint32_t big_min;
  dev::BigJob *jobs;
  Edge *bigscratch;
  uint32_t bigscratch_stride;  // slots per workgroup
  // diag >= 2: per update of the batch {rounds it failed in, code of the last failure, stream index of what blocked it,
  // planned window}: the dependency chains of an epoch can be followed afterwards (tools/diag_chains.py)
  uint32_t *dg;
};
constexpr uint32_t kBigJobs = 256;  // capacity of the round's job queue
constexpr uint32_t kRegionPadLeaves = 2u;
constexpr uint32_t kGrowLeaves = 8u;
constexpr uint32_t OS_PASS = 1u, OS_STAMP_BAD = 2u;


PMA_DEV bool key_earlier(unsigned long long k, uint32_t tag, uint32_t idx) { return (uint32_t)(k >> 32) == tag && (uint32_t)k < idx; }

// (dev::load_plan_head: the plan record's header and this lane's read range, requested in ONE batch and before the kernel's
// early-exit tests — the per-wave arrays are padded to the launch grid, so the loads are always in bounds.  The round kernels
// are chains of dependent loads; what can be asked for together is asked for together.)
using dev::PlanHead;
using dev::kHeadRanges;
#define PMA_FOR_EACH_READ_LEAF_H(h, pl, lane, LEAFVAR, BODY)                                  \
  do {                                                                                        \
    if ((h).nr <= (uint32_t)kHeadRanges && (h).nlong == 0u) {                                 \
      if ((uint32_t)(lane) < (h).nr)                                                          \
        for (uint32_t LEAFVAR = (h).my_lo; LEAFVAR <= (h).my_hi; LEAFVAR++) { BODY; }         \
    } else {                                                                                  \
      PMA_FOR_EACH_READ_LEAF(pl, lane, LEAFVAR, BODY);                                        \
    }                                                                                         \
  } while (0)

// ---- the round's bookkeeping (see OptCtl) -------------------------------------------------------------------------------
struct RoundState {
  uint32_t hor, used, nf, ch;    // this round: updates planned, of which from the carry list, first fresh index, width in force
  uint32_t pre[kStripes + 1];    // exclusive prefix of the previous round's stripe counts (pre[kStripes] = its deferred updates)
  uint32_t done, excl, excl_idx, excl_later, error, need_big;
  uint32_t prev_hor, prev_committed, prev_jobs;
};
// everything here is wave-uniform (loads from uniform addresses, scalar arithmetic)
PMA_DEV RoundState round_begin(const OptArgs &a, const OptCtl *c) {
  const uint32_t r = a.round, pp = (r - 1u) & 1u, pt = (r - 1u) % 3u;
  const uint32_t p_hor = c->hor[pp], p_used = c->used[pp], p_nf = c->nf[pp], p_ch = c->cur_h[pp];
  const uint32_t e1 = c->e1, max_h = c->max_horizon, wcap = c->width_cap, res = c->resident, adaptive = c->adaptive;
  const unsigned long long gb = c->gbar[pt];
  const uint32_t minkept = c->minkept[pt];
  const uint32_t p_jobs = c->njobs[pt], big_done = c->big_done_round;
  RoundState s;
  uint32_t K = 0;
#pragma unroll
  for (uint32_t q = 0; q < kStripes; q++) {
    s.pre[q] = K;
    K += c->kept[pt][q];
  }
  s.pre[kStripes] = K;
  const uint32_t ncommitted = p_hor - K;
  const uint32_t new_nf = p_nf + (p_hor - p_used);
  // adaptive width: dependency chains bound the number of commits per round (a hot vertex whose range sits at its
  // density bounds yields a few hundred disjoint windows per round however many updates are planned), so planning far
  // more than can commit only makes every round slower — but a round's time grows far slower than its width (~30 us
  // + ~2.5 us per 1024 updates), so width is only given up when almost nothing of it commits.  Narrow by 1/4 when less
  // than 10 % of a full-width round committed, widen by 1/4 when more than 30 % did.
  // Above one chip-full of waves (`resident`) only whole multiples make sense (a partly filled second pass costs a
  // full pass of latency), and a multiple is only worth its re-planning when nearly all of the round commits: up at
  // > 85 %, back down at < 70 % (config #4's partitions at critical density commit 60-70 % of a chip-full: at twice the
  // width they lost 5 %; configs #2 / #3 commit 93-97 % and gain 11-12 %).
  // The width limits how many FRESH updates join a round; the carry list is always planned whole (nothing that waits may
  // be left without its reservations while later updates commit), so after a narrowing the rounds shrink as the list drains.
  uint32_t ch = p_ch ? p_ch : wcap;
  if (adaptive && p_hor >= ch && p_hor > 0u) {  // only full-width rounds carry information about the width
    if (res && ch >= res) {
      if (ncommitted * 100u > p_hor * 85u) ch += res;
      else if (ch > res && ncommitted * 100u < p_hor * 70u) ch -= res;
      else if (ch == res && ncommitted * 100u < p_hor * 10u) ch -= ch / 4u;
    } else {
      if (ncommitted * 100u > p_hor * 30u) ch += ch / 4u;
      else if (ncommitted * 100u < p_hor * 10u) ch -= ch / 4u;
      if (res && ch > res) ch = res;
    }
  }
  if (res && ch > res) ch -= ch % res;
  if (ch < 1024u) ch = 1024u;
  if (ch > wcap) ch = wcap;
  const uint32_t avail = e1 - new_nf, room = ch > K ? ch - K : 0u;
  uint32_t nh = K + (avail < room ? avail : room);
  if (nh > max_h) nh = max_h;  // (the launched grid; the host keeps it at least as wide as the carry list)
  s.error = K > max_h ? 1u : 0u;
  s.hor = nh;
  s.used = K;
  s.nf = new_nf;
  s.ch = ch;
  s.done = (K == 0u && new_nf == e1) ? 1u : 0u;
  const uint32_t lowest = K ? minkept : new_nf;
  const uint32_t tag = (uint32_t)(make_key(r - 1u, 0) >> 32);
  s.excl = (!s.done && (uint32_t)(gb >> 32) == tag && (uint32_t)gb == lowest) ? 1u : 0u;
  s.excl_idx = lowest;
  // every planned update later than the lowest pending one is either still pending (in the carry list) or committed
  s.excl_later = (K >= 1u && new_nf - lowest - 1u > K - 1u) ? 1u : 0u;
  s.prev_hor = p_hor;
  s.prev_committed = ncommitted;
  s.prev_jobs = p_jobs;
  s.need_big = (p_hor > 0u && p_jobs > 0u && big_done != r - 1u) ? 1u : 0u;  // (windows queued, and nobody has rebalanced them)
  return s;
}
// by ONE lane of the launch: the outcome for the later launches of round r and for the host; the set of round r+1 reset
PMA_DEV void round_record(const OptArgs &a, OptCtl *c, const RoundState &s) {
  const uint32_t r = a.round, par = r & 1u, nt = (r + 1u) % 3u;
  if (c->book_round != r - 1u) {
    c->book_round = r - 1u;
    c->rounds += 1ull;
    c->committed += (unsigned long long)s.prev_committed;
    c->planned += (unsigned long long)s.prev_hor;
    c->jobs_total += (unsigned long long)s.prev_jobs;
  }
  c->hor[par] = s.hor;
  c->used[par] = s.used;
  c->nf[par] = s.nf;
  c->cur_h[par] = s.ch;
#pragma unroll
  for (uint32_t q = 0; q < kStripes; q++) c->kept[nt][q] = 0u;
  c->minkept[nt] = kMax;
  c->gbar[nt] = ~0ull;
  c->sbar[nt] = ~0ull;
  c->njobs[nt] = 0u;
  if (s.error) c->error = 77u;
  if (s.need_big) {  // (first things first: whatever else the round's outcome is, it is looked at again once the windows are done)
    c->need_big = 1u;
    return;
  }
  if (s.done) c->done = 1u;
  if (s.excl) {
    c->excl_idx = s.excl_idx;
    c->excl_later = s.excl_later;
    c->excl = 1u;
  }
}
// end of a chunk of rounds: the last round's outcome recorded (done / excl / counters / the next horizon) for the host, which is
// about to read the control block; a.round = the round that WOULD come next (its o_plan repeats the evaluation with the same result)
PMA_KERNEL void o_settle(OptArgs a) {
  OptCtl *c = a.ctl;
  if (c->done || c->violation || c->excl || c->error || c->need_big) return;
  const RoundState s = round_begin(a, c);
  if (wv::thread_idx() == 0) round_record(a, c, s);
}

// words of the round's state that wave 0 of a workgroup hands to the others (LDS, lane i <-> word i)
enum RoundWord : int { RW_STOP = 0, RW_HOR, RW_USED, RW_NF, RW_PRE0, RW_SKIP_IDX = RW_PRE0 + (int)kStripes + 1, RW_SKIP_ROUND, RW_WORDS };
static_assert(RW_WORDS <= 16, "round words");
template <bool EXTRAS>
PMA_DEV void o_plan_t(const OptArgs &a) {
  // The round's bookkeeping is ~250 scalar instructions and two dozen loads — the same for every wave.  Wave 0 of each workgroup
  // evaluates it and leaves the handful of words the others need in LDS (evaluated by every wave it was 45 % of this kernel's
  // scalar instructions, and the scalar unit is what bounds it).
  PMA_SHARED uint32_t rsh[16];
  OptCtl *c = a.ctl;
  const uint32_t rt = a.round % 3u;
  const int lane = wv::lane();
  // (one wave = one update: the wave's slot and everything that follows from it is the same in all lanes — see wv::uni)
  const uint32_t wib = wv::uni((uint32_t)wv::wave_in_block());
  const uint32_t wid = wv::uni(wv::block_idx() * 4u) + wib;  // (256-thread workgroups)
  if (wib == 0u) {
    const uint32_t f_done = c->done, f_viol = c->violation, f_excl = c->excl, f_err = c->error | c->need_big;
    const uint32_t skip_idx = c->skip_idx, skip_round = c->skip_round;
    uint32_t stop = (f_done | f_viol | f_excl | f_err) ? 1u : 0u;
    uint32_t w = 0;
    if (!stop) {
      const RoundState rs = round_begin(a, c);
      if (wid == 0u && lane == 0) round_record(a, c, rs);
      stop = (rs.done | rs.excl | rs.error | rs.need_big) ? 1u : 0u;
      w = wv::setlane<RW_HOR>(w, rs.hor);
      w = wv::setlane<RW_USED>(w, rs.used);
      w = wv::setlane<RW_NF>(w, rs.nf);
      w = wv::setlane<RW_PRE0 + 0>(w, rs.pre[0]);
      w = wv::setlane<RW_PRE0 + 1>(w, rs.pre[1]);
      w = wv::setlane<RW_PRE0 + 2>(w, rs.pre[2]);
      w = wv::setlane<RW_PRE0 + 3>(w, rs.pre[3]);
      w = wv::setlane<RW_PRE0 + 4>(w, rs.pre[4]);
      w = wv::setlane<RW_PRE0 + 5>(w, rs.pre[5]);
      w = wv::setlane<RW_PRE0 + 6>(w, rs.pre[6]);
      w = wv::setlane<RW_PRE0 + 7>(w, rs.pre[7]);
      w = wv::setlane<RW_PRE0 + 8>(w, rs.pre[8]);
      w = wv::setlane<RW_SKIP_IDX>(w, skip_idx);
      w = wv::setlane<RW_SKIP_ROUND>(w, skip_round);
    }
    w = wv::setlane<RW_STOP>(w, stop);
    if (lane < 16) rsh[lane] = w;
  }
  wv::block_sync();
  const uint32_t rw = rsh[lane & 15];
  if (wv::bcast(rw, RW_STOP)) return;
  const uint32_t hor = wv::bcast(rw, RW_HOR), used = wv::bcast(rw, RW_USED);
  if (wid >= hor) return;
  uint32_t idx;
  if (wid < used) {  // the wid-th deferred update of the previous round: stripe by stripe
    const uint32_t *carry = ((a.round - 1u) & 1u) ? a.carry1 : a.carry0;
    // (lanes RW_PRE0 + k hold the exclusive prefix of the stripe counts: the stripe is the number of prefixes at or below wid)
    const uint64_t mle = wv::ballot(lane >= RW_PRE0 + 1 && lane < RW_PRE0 + (int)kStripes && rw <= wid);
    const uint32_t q = (uint32_t)wv::popc64(mle);
    const uint32_t base = wv::bcast(rw, RW_PRE0 + (int)q);
    idx = wv::uni(carry[(uint64_t)q * a.carry_cap + (wid - base)]);
  } else {
    idx = wv::bcast(rw, RW_NF) + (wid - used);
  }
  const uint32_t skip_idx = wv::bcast(rw, RW_SKIP_IDX), skip_round = wv::bcast(rw, RW_SKIP_ROUND);
  Op op = a.ops[idx];
  op.src = wv::uni(op.src);
  op.dst = wv::uni(op.dst);
  op.op = wv::uni(op.op);
  Plan *pl = &a.plans[wid];
  if (a.round == skip_round && idx == skip_idx) {  // executed by the exclusive executor in the middle of this epoch: nothing left to do, commits at once
    dev::store_plan_header(pl, K_SKIP, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, idx, op);
    return;
  }
  if (EXTRAS && (a.dbg & 1u)) (void)dev::plan_op(a.v, op, pl, idx);
  if (EXTRAS && (a.dbg & 2u) && op.src < a.v.g.n) {
    dev::RangeRec rr2;
    dev::SearchHit h2;
    const Node nd = a.v.nodes[op.src];
    const uint32_t r = dev::pma_search(a.v, op.dst, nd.beginning + 1u, nd.end, rr2, &h2);
    if (r == 0xFFFFFFF0u) a.status[wid] = rr2.nr + h2.known;  // (never: keeps the call alive)
  }
  if (EXTRAS && (a.dbg & 4u)) {
    const RoundState r2 = round_begin(a, c);
    if (r2.hor == 0xFFFFFFF0u) a.status[wid] = r2.nf;
  }
  // the plan record goes to memory for o_check / o_apply; this kernel reserves straight from the registers
  const dev::PlanRegs pr = dev::plan_op(a.v, op, pl, idx);
  const unsigned long long key = make_key(a.round, idx);
  const uint32_t kind = pr.kind;
  if (kind == K_EXCL) {
    if (lane == 0) wv::atomic_min_u64(&c->gbar[rt], key);
    return;
  }
  if (kind == K_DUP) {
    if (lane == 0) wv::atomic_min_u64(&a.v.dres[pr.wleaf_lo], key);
  } else if (kind_strong(kind)) {
    const uint32_t wl = pr.wleaf_lo, wh = pr.wleaf_hi;
    for (uint32_t base = wl; base <= wh; base += 64) {  // (scalar trip count)
      const uint32_t leaf = base + (uint32_t)lane;
      if (leaf <= wh) wv::atomic_min_u64(&a.v.wres[leaf], key);
    }
    // an update whose window is already within two levels of the exclusive threshold is likely to turn exclusive
    // once the earlier updates have landed: nothing later may overtake it (soft barrier)
    // ... and so is an update that has ALREADY been deferred at least once and whose window is big (a.defer_barrier
    // slots): its window keeps growing while it waits behind a hot range, and everything committed around it meanwhile
    // is a candidate for a rollback
    if ((pr.wlen >= a.soft_barrier || (a.defer_barrier && wid < used && pr.wlen >= a.defer_barrier)) && lane == 0) {
      wv::atomic_min_u64(&c->sbar[rt], key);
    }
    const uint32_t ml = pr.mv_lo, mh = pr.mv_hi;
    if (ml <= mh)
      for (uint64_t base = ml; base <= (uint64_t)mh; base += 64) {
        const uint64_t u = base + (uint64_t)lane;
        if (u <= (uint64_t)mh) wv::atomic_min_u64(&a.v.vw[u], key);
      }
  }
  if (pr.nlong == 0u) {  // lane r holds read range r
    if ((uint32_t)lane < pr.nr)
      for (uint32_t leaf = pr.my_lo; leaf <= pr.my_hi; leaf++) wv::atomic_min_u64(&a.v.rres[leaf], key);
  } else {
    wv::fence();  // (rare) more ranges than lanes, or long ranges: walk the record this wave has just written
    PMA_FOR_EACH_READ_LEAF(pl, lane, leaf, wv::atomic_min_u64(&a.v.rres[leaf], key));
  }
  if (kind_real(kind) && op.src < a.v.g.n) {  // readers of the positions of sentinels src and src+1 (only when the result depends on them)
    if (lane == 0 && (pr.sdep & 1u)) wv::atomic_min_u64(&a.v.vr[op.src], key);
    if (lane == 1 && (pr.sdep & 2u) && op.src + 1u < a.v.g.n) wv::atomic_min_u64(&a.v.vr[op.src + 1u], key);
  }
}
// (35 VGPRs; 8 waves per SIMD need <= 96 scalar registers: the planner is a chain of dependent loads, residency is throughput)
PMA_KERNEL void PMA_LAUNCH_BOUNDS(256, 8) o_plan(OptArgs a) { o_plan_t<false>(a); }
PMA_KERNEL void o_plan_x(OptArgs a) { o_plan_t<true>(a); }

// One update checked by one wave: `wid` its horizon slot, `h` its plan header (scalars; h.my_lo / my_hi: lane r holds read range
// r), gbar / sbar this round's barriers.  The body of the wave-per-update kernel (o_check_x, the diagnostics build) and the slow
// path of the lane-per-update kernel (o_check) for updates with long footprints.
template <bool EXTRAS>
PMA_DEV void o_check_one(const OptArgs &a, OptCtl *c, uint32_t par, uint32_t wid, const PlanHead &h, const Plan *pl, unsigned long long gbar,
                         unsigned long long sbar) {
  const int lane = wv::lane();
  const uint32_t idx = h.idx;
  const uint32_t kind = h.kind;
  const unsigned long long key = make_key(a.round, idx);
  const uint32_t tag = (uint32_t)(key >> 32);
  // (a soft barrier holds back everything AFTER the update that raised it, that update itself may commit)
  bool fail = (kind == K_EXCL) || key_earlier(gbar, tag, idx) || key_earlier(sbar, tag, idx);
  uint32_t why = (kind == K_EXCL) ? 0u : (fail ? 1u : 99u);  // diagnostics: first reason (lowest code wins below)
  if (fail) {
    // An exclusive update, or one behind this round's barrier: it does not commit now, and neither does anything after it
    // (the barrier is earlier than all of them), so there is nobody to keep out of its regions and nothing to learn from
    // its footprint — which, for a climb towards the root, is every leaf of the array (a 3.5 ms walk by one wave, while
    // the rest of the launch waits).  Its stamps are looked at in the round that does check it.
    if ((EXTRAS && a.diag) && lane == 0) wv::atomic_add_u64(&c->why[why], 1ull);
    if (lane == 0 && (EXTRAS && a.diag) && a.dg != nullptr) {
      uint32_t *r = a.dg + 4ull * idx;
      r[0] += 1u;
      r[1] = why;
      r[2] = (kind == K_EXCL) ? idx : (uint32_t)(key_earlier(gbar, tag, idx) ? gbar : sbar);
      r[3] = h.wlen;
    }
    if (lane == 0) a.status[wid] = 0u;
    return;
  }
#define PMA_WHY(code) do { if ((EXTRAS && a.diag) && (code) < why) why = (code); } while (0)
#define PMA_WHYB(code, bkey) do { if ((EXTRAS && a.diag) && (code) < why) { why = (code); blk = (uint32_t)(bkey); } } while (0)
  uint32_t blk = kMax;
  bool stamp_bad = false;
  const uint32_t me1 = idx + 1u;  // stamps hold (index + 1) of the latest committed toucher
  const bool writes = kind_writes(kind);
  const bool strong = kind_strong(kind);
  if (kind == K_DUP) {
    const uint32_t leaf = h.wleaf_lo;
    if (key_earlier(a.v.wres[leaf], tag, idx)) { fail = true; PMA_WHY(2u); }  // an earlier pending update moves slots of this leaf
    if (a.v.dres[leaf] != key) { fail = true; PMA_WHY(2u); }                   // an earlier pending duplicate on this leaf
  }
  if (strong) {
    const uint32_t wl = h.wleaf_lo, wh = h.wleaf_hi;
    // (a big window spans thousands of leaves: four leaves per lane are requested together, 20 loads per trip)
    for (uint32_t base = wl; base <= wh; base += 256u) {
      unsigned long long kw[4], kd[4], kr[4];
      uint32_t sw[4], sr[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t leaf = base + (uint32_t)q * 64u + (uint32_t)lane;
        const bool in = leaf <= wh && leaf >= base;
        kw[q] = in ? a.v.wres[leaf] : key;
        kd[q] = in ? a.v.dres[leaf] : ~0ull;
        kr[q] = in ? a.v.rres[leaf] : ~0ull;
        sw[q] = in ? a.wstamp[leaf] : 0u;
        sr[q] = in ? a.rstamp[leaf] : 0u;
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t leaf = base + (uint32_t)q * 64u + (uint32_t)lane;
        if (kw[q] != key) { fail = true; PMA_WHYB(3u, kw[q]); }                   // an earlier pending update writes it
        if (key_earlier(kd[q], tag, idx)) { fail = true; PMA_WHYB(2u, kd[q]); }   // an earlier pending duplicate overwrites a slot here
        if (key_earlier(kr[q], tag, idx)) { fail = true; PMA_WHYB(4u, kr[q]); }   // an earlier pending update reads it
        if (sw[q] > me1 || sr[q] > me1) {  // a LATER update already touched it
          stamp_bad = true;
          a.vdbg[4 * wid + 0] = leaf;
          a.vdbg[4 * wid + 1] = sw[q] > me1 ? sw[q] : sr[q];
          a.vdbg[4 * wid + 2] = sw[q] > me1 ? 1u : 2u;
        }
      }
      if (wh - base < 256u) break;  // (no wrap-around at the top of the leaf range)
    }
  }
  PMA_FOR_EACH_READ_LEAF_H(h, pl, lane, leaf, {
    if (key_earlier(a.v.wres[leaf], tag, idx)) { fail = true; PMA_WHYB(5u, a.v.wres[leaf]); }  // an earlier pending update writes what we read
    if (a.wstamp[leaf] > me1) {                               // a LATER update already wrote what we read
      stamp_bad = true;
      a.vdbg[4 * wid + 0] = leaf;
      a.vdbg[4 * wid + 1] = a.wstamp[leaf];
      a.vdbg[4 * wid + 2] = 3u;
    }
  });
  if (kind_real(kind)) {
    const uint32_t src = h.op.src;
    if (src < a.v.g.n && lane < 2 && ((h.sdep >> lane) & 1u) && src + (uint32_t)lane < a.v.g.n) {  // lane 0: sentinel src, lane 1: sentinel src+1
      const uint32_t u = src + (uint32_t)lane;
      if (key_earlier(a.v.vw[u], tag, idx)) { fail = true; PMA_WHYB(6u, a.v.vw[u]); }  // an earlier pending update moves a sentinel we located by
      if (a.vws[u] > me1) {                                // a LATER update already moved it
        stamp_bad = true;
        a.vdbg[4 * wid + 0] = u;
        a.vdbg[4 * wid + 1] = a.vws[u];
        a.vdbg[4 * wid + 2] = 4u;
      }
    }
    if (strong) {
      const uint32_t ml = h.mv_lo, mh = h.mv_hi;
      for (uint64_t u = (uint64_t)ml + (uint64_t)lane; u <= (uint64_t)mh && ml <= mh; u += 64) {
        if (key_earlier(a.v.vr[u], tag, idx)) { fail = true; PMA_WHYB(7u, a.v.vr[u]); }  // an earlier pending update still needs the old position
        if (a.vrs[u] > me1 || a.vws[u] > me1) {              // a LATER update already used / moved it
          stamp_bad = true;
          a.vdbg[4 * wid + 0] = (uint32_t)u;
          a.vdbg[4 * wid + 1] = a.vrs[u] > me1 ? a.vrs[u] : a.vws[u];
          a.vdbg[4 * wid + 2] = 5u;
        }
      }
    }
  }
  const bool anyfail = wv::ballot(fail) != 0;
  const bool anybad = wv::ballot(stamp_bad) != 0;
  if ((EXTRAS && a.diag) && anyfail) {
    uint32_t w = why, wb = blk;
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t y = wv::shfl(w, lane ^ o), yb = wv::shfl(wb, lane ^ o);
      if (y < w || (y == w && yb < wb)) {
        w = y;
        wb = yb;
      }
    }
    if (lane == 0 && w < 12u) wv::atomic_add_u64(&c->why[w], 1ull);
    if (lane == 0 && a.dg != nullptr) {
      uint32_t *r = a.dg + 4ull * idx;
      r[0] += 1u;
      r[1] = w;
      r[2] = wb;
      r[3] = h.wlen;
    }
  }
  if (anyfail && kind_real(kind)) {
    // a deferred update keeps later updates out of its region(s); its footprint may still creep over a region edge
    // by a slide, so the mark is padded by kRegionPadLeaves leaves on both sides
    const uint32_t nleaves = (uint32_t)(a.v.g.N >> a.v.g.sh);
    uint32_t ll = writes ? h.wleaf_lo : (h.index >> a.v.g.sh), lh = writes ? h.wleaf_hi : ll;
    ll = (ll > kRegionPadLeaves) ? ll - kRegionPadLeaves : 0u;
    lh = (lh + kRegionPadLeaves < nleaves) ? lh + kRegionPadLeaves : nleaves - 1u;
    const uint32_t pglo = ll >> a.regshift, pghi = lh >> a.regshift;
    for (uint32_t g = pglo + (uint32_t)lane; g <= pghi; g += 64) wv::atomic_min_u64(&a.regfail[g], key);
    // leaf-level mark for later READERS: the deferred update's window can still grow to an ancestor block; cover
    // the aligned block of 4x its tentative window (at least kGrowLeaves leaves) plus the slide pad
    uint32_t wleaves = writes && h.wlen ? (h.wlen >> a.v.g.sh) : 1u;
    if (wleaves < 1u) wleaves = 1u;
    uint32_t blk = wleaves * 4u;
    if (blk < kGrowLeaves) blk = kGrowLeaves;
    const uint32_t anchor = writes && h.wlen ? (h.wstart >> a.v.g.sh) : (h.index >> a.v.g.sh);
    uint32_t bl = anchor & ~(blk - 1u), bh = bl + blk - 1u;
    if (ll < bl) bl = ll;
    if (lh > bh) bh = lh;
    if (bh >= nleaves) bh = nleaves - 1u;
    for (uint32_t leaf = bl + (uint32_t)lane; leaf <= bh; leaf += 64) wv::atomic_min_u64(&a.pfail[leaf], key);
  }
  if (lane == 0) a.status[wid] = (anyfail ? 0u : OS_PASS) | (anybad ? OS_STAMP_BAD : 0u);
}
#undef PMA_WHY
#undef PMA_WHYB

// wave per update: the diagnostics build (o_check_x: every reason counted, the blocker traced), and the plain build for streams
// whose updates have LONG footprints (o_check_w: big windows, many-level climbs, runs of moved sentinels — a hot vertex' range) —
// the lane-per-update kernel below hands those to the wave one after the other, which is slower than a wave each from the start
template <bool EXTRAS>
PMA_DEV void o_check_wave(const OptArgs &a) {
  OptCtl *c = a.ctl;
  const uint32_t par = a.round & 1u;
  const uint32_t wid = wv::uni(wv::block_idx() * 4u + (uint32_t)wv::wave_in_block());
  const uint32_t f_done = c->done, f_viol = c->violation, f_excl = c->excl, f_err = c->error | c->need_big;
  const uint32_t hor = c->hor[par];
  const unsigned long long gbar = c->gbar[a.round % 3u], sbar = c->sbar[a.round % 3u];
  const Plan *pl = &a.plans[wid];
  const PlanHead h = dev::load_plan_head(pl);
  if (f_done || f_viol || f_excl || f_err) return;
  if (wid >= hor) return;
  o_check_one<EXTRAS>(a, c, par, wid, h, pl, gbar, sbar);
}
PMA_KERNEL void o_check_x(OptArgs a) { o_check_wave<true>(a); }
PMA_KERNEL void o_check_w(OptArgs a) { o_check_wave<false>(a); }

// LANE per update (the default).  Checking an update is a dozen independent loads and compares — reservation keys and stamps of
// the one or two leaves it writes, of the two or three it reads, of the sentinels around them — and, for the few that fail,
// a handful of atomics that mark their region.  A wave per update spent ~370 instructions on that, most of them scalar
// bookkeeping, and the scalar unit's issue rate (one instruction per SIMD every four cycles, shared by all its waves) was
// what the kernel's time followed; 64 updates per wave share one instruction stream.  Lanes diverge only in trip counts, and
// those are capped: an update with a long footprint (a big window, many read ranges, a wide growth mark) is left to the wave —
// after the lanes' own pass the wave takes such updates one by one through o_check_one, fields broadcast from the owning lane.
// What a lane checks itself has a FIXED shape — at most kCkWrite write leaves, kCkRanges read ranges of at most two leaves,
// kCkMoved moved sentinels: every load is issued unconditionally (index 0 where the update has nothing there, the result
// masked), all of them before the first compare, so the whole check is one round trip instead of one per loop.
// One wave per workgroup: a lane's loads all go to lines of its own, so a wave-wide load is 64 requests to the CU's address unit —
// 256 updates per workgroup put a whole round on 72 of the 256 CUs and the kernel took twice the time.
constexpr uint32_t kCkThreads = 64;
constexpr uint32_t kCkWrite = 2;   // write leaves
constexpr int kCkRanges = 4;       // read ranges, each of one or two leaves
constexpr uint32_t kCkMoved = 4;   // sentinels moved: checked with everything else ...
constexpr uint32_t kCkMovedMax = 68;  // ... and up to this many in further trips of 8 (a leaf of isolated vertices is all sentinels)
constexpr uint32_t kCkMark = 32;   // leaves of the growth mark (pfail) a failing lane writes itself
PMA_KERNEL void o_check(OptArgs a) {
  OptCtl *c = a.ctl;
  const uint32_t par = a.round & 1u;
  const int lane = wv::lane();
  const uint32_t slot = wv::block_idx() * kCkThreads + wv::thread_idx();  // this LANE's horizon slot
  const uint32_t wave_slot0 = wv::uni(slot - (uint32_t)lane);
  const uint32_t f_done = c->done, f_viol = c->violation, f_excl = c->excl, f_err = c->error | c->need_big;
  const uint32_t hor = c->hor[par];
  const unsigned long long gbar = c->gbar[a.round % 3u], sbar = c->sbar[a.round % 3u];
  // my record's first 112 bytes: header (20 words) + read ranges 0 .. 3 — requested WITH the control words, before the early exits
  // (the record array is padded to the launch grid)
  const uint4 *rec = reinterpret_cast<const uint4 *>(&a.plans[slot]);
  const uint4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4], q5 = rec[5], q6 = rec[6];
  if (f_done || f_viol || f_excl || f_err) return;
  if (wave_slot0 >= hor) return;
  const bool active = slot < hor;
  const uint32_t kind = active ? q0.x : (uint32_t)K_SKIP, index = q0.y, wstart = q0.w, wlen = q1.x, wl = q1.y, wh = q1.z, mv_lo = q1.w, mv_hi = q2.x,
                 nr = q3.x, nlong = q3.y, sdep = q3.z, idx = q3.w, src = q4.x;
  const uint32_t rlo[kCkRanges] = {q5.x, q5.z, q6.x, q6.z}, rhi[kCkRanges] = {q5.y, q5.w, q6.y, q6.w};
  const unsigned long long key = make_key(a.round, idx);
  const uint32_t tag = (uint32_t)(key >> 32);
  const uint32_t me1 = idx + 1u;
  const bool writes = kind_writes(kind), strong = kind_strong(kind), real = kind_real(kind);
  const int sh = a.v.g.sh;
  const uint32_t nleaves = (uint32_t)(a.v.g.N >> sh);
  // (a soft barrier holds back everything AFTER the update that raised it, that update itself may commit)
  const bool barred = (kind == K_EXCL) || key_earlier(gbar, tag, idx) || key_earlier(sbar, tag, idx);
  // what a failing update marks (the same arithmetic as o_check_one): its regions, padded, and its growth block
  uint32_t ll = writes ? wl : (index >> sh), lh = writes ? wh : ll;
  ll = (ll > kRegionPadLeaves) ? ll - kRegionPadLeaves : 0u;
  lh = (lh + kRegionPadLeaves < nleaves) ? lh + kRegionPadLeaves : nleaves - 1u;
  const uint32_t pglo = ll >> a.regshift, pghi = lh >> a.regshift;
  uint32_t wleaves = writes && wlen ? (wlen >> sh) : 1u;
  if (wleaves < 1u) wleaves = 1u;
  uint32_t gblk = wleaves * 4u;
  if (gblk < kGrowLeaves) gblk = kGrowLeaves;
  const uint32_t anchor = writes && wlen ? (wstart >> sh) : (index >> sh);
  uint32_t bl = anchor & ~(gblk - 1u), bh = bl + gblk - 1u;
  if (ll < bl) bl = ll;
  if (lh > bh) bh = lh;
  if (bh >= nleaves) bh = nleaves - 1u;
  bool longr = false;
#pragma unroll
  for (int r = 0; r < kCkRanges; r++)
    if ((uint32_t)r < nr && rhi[r] - rlo[r] >= 2u) longr = true;
  const bool moves = strong && mv_lo <= mv_hi;
  // long footprints go to the wave (a barred update's footprint is never walked: see o_check_one)
  const bool complex = active && !barred && real &&
                       (nlong != 0u || nr > (uint32_t)kCkRanges || longr || (writes && wh - wl >= kCkWrite) || (moves && mv_hi - mv_lo >= kCkMovedMax) ||
                        bh - bl >= kCkMark || pghi - pglo >= 4u);
  const bool mine = active && !barred && !complex;
  bool fail = barred, stamp_bad = false;
  uint32_t bad_where = 0, bad_stamp = 0, bad_what = 0;
  {
    // ---- every load, unconditionally ------------------------------------------------------------------------------------
    const bool wany = mine && writes;  // (K_DUP: one leaf, wl == wh)
    const uint32_t w0 = wany ? wl : 0u, w1 = (wany && wh > wl) ? wh : w0;
    const unsigned long long kw0 = a.v.wres[w0], kd0 = a.v.dres[w0], kr0 = a.v.rres[w0], kw1 = a.v.wres[w1], kd1 = a.v.dres[w1], kr1 = a.v.rres[w1];
    const uint32_t sw0 = a.wstamp[w0], sr0 = a.rstamp[w0], sw1 = a.wstamp[w1], sr1 = a.rstamp[w1];
    unsigned long long rk[kCkRanges][2];
    uint32_t rs[kCkRanges][2], rl[kCkRanges][2];
#pragma unroll
    for (int r = 0; r < kCkRanges; r++) {
      const bool on = mine && (uint32_t)r < nr;
      rl[r][0] = on ? rlo[r] : 0u;
      rl[r][1] = on ? rhi[r] : 0u;
      rk[r][0] = a.v.wres[rl[r][0]];
      rk[r][1] = a.v.wres[rl[r][1]];
      rs[r][0] = a.wstamp[rl[r][0]];
      rs[r][1] = a.wstamp[rl[r][1]];
    }
    const bool vany = mine && real && src < a.v.g.n;
    const bool von0 = vany && (sdep & 1u), von1 = vany && (sdep & 2u) && src + 1u < a.v.g.n;
    const uint32_t u0 = von0 ? src : 0u, u1 = von1 ? src + 1u : 0u;
    const unsigned long long kv0 = a.v.vw[u0], kv1 = a.v.vw[u1];
    const uint32_t sv0 = a.vws[u0], sv1 = a.vws[u1];
    const bool many = vany && moves;
    unsigned long long mk[kCkMoved];
    uint32_t m1[kCkMoved], m2[kCkMoved], mu[kCkMoved];
#pragma unroll
    for (uint32_t q = 0; q < kCkMoved; q++) {
      mu[q] = (many && mv_lo + q <= mv_hi) ? mv_lo + q : 0u;
      mk[q] = a.v.vr[mu[q]];
      m1[q] = a.vrs[mu[q]];
      m2[q] = a.vws[mu[q]];
    }
    // ---- every compare ------------------------------------------------------------------------------------------------
    if (mine && kind == K_DUP) {
      if (key_earlier(kw0, tag, idx)) fail = true;  // an earlier pending update moves slots of this leaf
      if (kd0 != key) fail = true;                   // an earlier pending duplicate on this leaf
    }
    if (mine && strong) {
      if (kw0 != key || kw1 != key) fail = true;                                  // an earlier pending update writes it
      if (key_earlier(kd0, tag, idx) || key_earlier(kd1, tag, idx)) fail = true;  // an earlier pending duplicate overwrites a slot here
      if (key_earlier(kr0, tag, idx) || key_earlier(kr1, tag, idx)) fail = true;  // an earlier pending update reads it
      if (sw1 > me1 || sr1 > me1) {                                               // a LATER update already touched it
        stamp_bad = true;
        bad_where = w1;
        bad_stamp = sw1 > me1 ? sw1 : sr1;
        bad_what = sw1 > me1 ? 1u : 2u;
      }
      if (sw0 > me1 || sr0 > me1) {
        stamp_bad = true;
        bad_where = w0;
        bad_stamp = sw0 > me1 ? sw0 : sr0;
        bad_what = sw0 > me1 ? 1u : 2u;
      }
    }
#pragma unroll
    for (int r = 0; r < kCkRanges; r++) {
      if (mine && (uint32_t)r < nr) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
          if (key_earlier(rk[r][e], tag, idx)) fail = true;  // an earlier pending update writes what we read
          if (rs[r][e] > me1) {                               // a LATER update already wrote what we read
            stamp_bad = true;
            bad_where = rl[r][e];
            bad_stamp = rs[r][e];
            bad_what = 3u;
          }
        }
      }
    }
    if (von0) {  // sentinel src — only when the search result depends on its position
      if (key_earlier(kv0, tag, idx)) fail = true;  // an earlier pending update moves a sentinel we located by
      if (sv0 > me1) {                              // a LATER update already moved it
        stamp_bad = true;
        bad_where = u0;
        bad_stamp = sv0;
        bad_what = 4u;
      }
    }
    if (von1) {  // sentinel src + 1
      if (key_earlier(kv1, tag, idx)) fail = true;
      if (sv1 > me1) {
        stamp_bad = true;
        bad_where = u1;
        bad_stamp = sv1;
        bad_what = 4u;
      }
    }
#pragma unroll
    for (uint32_t q = 0; q < kCkMoved; q++) {
      if (many && mv_lo + q <= mv_hi) {
        if (key_earlier(mk[q], tag, idx)) fail = true;  // an earlier pending update still needs the old position
        if (m1[q] > me1 || m2[q] > me1) {               // a LATER update already used / moved it
          stamp_bad = true;
          bad_where = mu[q];
          bad_stamp = m1[q] > me1 ? m1[q] : m2[q];
          bad_what = 5u;
        }
      }
    }
  }
  if (mine && real && src < a.v.g.n && moves && mv_hi - mv_lo >= kCkMoved) {  // the rest of a long run of moved sentinels, 8 per trip
    for (uint32_t base = mv_lo + kCkMoved; base <= mv_hi && base >= mv_lo; base += 8u) {
      unsigned long long mk[8];
      uint32_t m1[8], m2[8];
#pragma unroll
      for (uint32_t q = 0; q < 8u; q++) {
        const uint32_t u = (base + q <= mv_hi) ? base + q : mv_hi;
        mk[q] = a.v.vr[u];
        m1[q] = a.vrs[u];
        m2[q] = a.vws[u];
      }
#pragma unroll
      for (uint32_t q = 0; q < 8u; q++) {
        if (key_earlier(mk[q], tag, idx)) fail = true;
        if (m1[q] > me1 || m2[q] > me1) {
          stamp_bad = true;
          bad_where = (base + q <= mv_hi) ? base + q : mv_hi;
          bad_stamp = m1[q] > me1 ? m1[q] : m2[q];
          bad_what = 5u;
        }
      }
    }
  }
  {
    // a deferred update keeps later updates out of its region(s); its footprint may still creep over a region edge by a
    // slide, so the mark is padded (ll / lh above); and a leaf-level mark for later READERS over the block its window may
    // still grow into.  Written by the WAVE, one failing update at a time, lane q taking the q-th leaf / region of the mark
    // (a lane marking its own 12 leaves one atomic after the other waited for each to come back)
    uint64_t fm = wv::ballot(mine && fail && real);
    while (fm) {
      const int l = wv::ctz64(fm);
      fm &= fm - 1ull;
      const uint32_t f_bl = wv::bcast(bl, l), f_bh = wv::bcast(bh, l), f_g0 = wv::bcast(pglo, l), f_g1 = wv::bcast(pghi, l);
      const unsigned long long f_key = make_key(a.round, wv::bcast(idx, l));
      if (f_bl + (uint32_t)lane <= f_bh) wv::atomic_min_u64(&a.pfail[f_bl + (uint32_t)lane], f_key);   // (bh - bl < kCkMark <= 64)
      if (f_g0 + (uint32_t)lane <= f_g1) wv::atomic_min_u64(&a.regfail[f_g0 + (uint32_t)lane], f_key);  // (at most 4 regions)
    }
  }
  if (mine && stamp_bad) {
    a.vdbg[4 * slot + 0] = bad_where;
    a.vdbg[4 * slot + 1] = bad_stamp;
    a.vdbg[4 * slot + 2] = bad_what;
  }
  if (active && !complex) a.status[slot] = (fail ? 0u : OS_PASS) | (stamp_bad ? OS_STAMP_BAD : 0u);
  // the long ones, by the whole wave, one after the other
  uint64_t todo = wv::ballot(complex);
  if (todo && lane == 0) wv::atomic_add_u64(&c->long_checks, (unsigned long long)wv::popc64(todo));  // (the engine picks the kernel by this share)
  while (todo) {
    const int l = wv::ctz64(todo);
    todo &= todo - 1ull;
    PlanHead h;
    h.kind = wv::bcast(q0.x, l);
    h.index = wv::bcast(q0.y, l);
    h.gap = wv::bcast(q0.z, l);
    h.wstart = wv::bcast(q0.w, l);
    h.wlen = wv::bcast(q1.x, l);
    h.wleaf_lo = wv::bcast(q1.y, l);
    h.wleaf_hi = wv::bcast(q1.z, l);
    h.mv_lo = wv::bcast(q1.w, l);
    h.mv_hi = wv::bcast(q2.x, l);
    h.sleaf_b = wv::bcast(q2.y, l);
    h.sleaf_e = wv::bcast(q2.z, l);
    h.alg_calls = h.alg_slots = 0;
    h.nr = wv::bcast(q3.x, l);
    h.nlong = wv::bcast(q3.y, l);
    h.sdep = wv::bcast(q3.z, l);
    h.idx = wv::bcast(q3.w, l);
    h.op = Op{wv::bcast(q4.x, l), wv::bcast(q4.y, l), wv::bcast(q4.z, l)};
    const uint32_t wid = wave_slot0 + (uint32_t)l;
    const Plan *pl = &a.plans[wid];
    PlanRange rg{1u, 0u};
    if (lane < kHeadRanges) rg = pl->r[lane];
    h.my_lo = rg.lo;
    h.my_hi = rg.hi;
    o_check_one<false>(a, c, par, wid, h, pl, gbar, sbar);
  }
}

// EXTRAS = false: the opt-in experiments (chains, zones) and the diagnostics are compiled out — carried along as run-time
// branches they cost the calm stream 4 % (config #2: 179 vs 187 M updates/s); the engine launches the *_x kernels when one is on
// returns the stream index of an update of this round that did NOT commit (it goes to the carry list), kMax otherwise
template <bool EXTRAS>
PMA_DEV uint32_t o_apply_wave(const OptArgs &a, uint32_t *lds_wave) {
  OptCtl *c = a.ctl;
  const uint32_t par = a.round & 1u;
  const uint32_t wid = wv::uni(wv::block_idx() * 4u + (uint32_t)wv::wave_in_block());
  const int lane = wv::lane();
  const uint32_t f_done = c->done, f_viol = c->violation, f_excl = c->excl, f_err = c->error | c->need_big;
  const uint32_t hor = c->hor[par];
  const uint32_t st = wv::uni(a.status[wid]);
  const Plan *pl = &a.plans[wid];
  const PlanHead h = dev::load_plan_head(pl);
  const uint32_t idx = h.idx;
  const Op op = h.op;
  if (f_done || f_viol || f_excl || f_err) return kMax;
  if (wid >= hor) return kMax;
  if (!(st & OS_PASS)) return idx;
  const uint32_t kind = h.kind;
  const unsigned long long key = make_key(a.round, idx);
  const uint32_t tag = (uint32_t)(key >> 32);
  const bool writes = kind_writes(kind);
  if (kind_real(kind)) {
    uint32_t glo, ghi;
    if (writes) {
      glo = h.wleaf_lo >> a.regshift;
      ghi = h.wleaf_hi >> a.regshift;
    } else {
      glo = ghi = (h.index >> a.v.g.sh) >> a.regshift;
    }
    bool blocked = false, blocked_r = false;
    uint32_t rblk = kMax;
    for (uint32_t g = glo + (uint32_t)lane; g <= ghi; g += 64)
      if (key_earlier(a.regfail[g], tag, idx)) {
        blocked_r = true;
        rblk = (uint32_t)a.regfail[g];
      }
    // ... nor may we have READ a leaf an earlier deferred update may still grow over
    PMA_FOR_EACH_READ_LEAF_H(h, pl, lane, leaf, { if (key_earlier(a.pfail[leaf], tag, idx)) blocked = true; });
    // ... nor located our range by a sentinel inside the block a deferred earlier update may still grow over
    if (lane == 0 && (h.sdep & 1u) && key_earlier(a.pfail[h.sleaf_b], tag, idx)) blocked = true;
    if (lane == 1 && (h.sdep & 2u) && key_earlier(a.pfail[h.sleaf_e], tag, idx)) blocked = true;
    const bool any_r = wv::ballot(blocked_r) != 0, any_p = wv::ballot(blocked) != 0;
    if ((EXTRAS && a.diag) && (any_r || any_p) && lane == 0) wv::atomic_add_u64(&c->why[any_r ? 8 : 9], 1ull);
    if ((EXTRAS && a.diag) && a.dg != nullptr && (any_r || any_p)) {
      for (int o = 32; o > 0; o >>= 1) {
        const uint32_t y = wv::shfl(rblk, lane ^ o);
        rblk = y < rblk ? y : rblk;
      }
      if (lane == 0) {
        uint32_t *r = a.dg + 4ull * idx;
        r[0] += 1u;
        r[1] = any_r ? 8u : 9u;
        r[2] = rblk;
        r[3] = h.wlen;
      }
    }
    if (any_r || any_p) {  // an earlier update of this region was deferred: keep stream order inside it
      return idx;
    }
  }
  if (st & OS_STAMP_BAD) {
    if ((EXTRAS && a.diag) && lane == 0) wv::atomic_add_u64(&c->why[10], 1ull);
    if (lane == 0) {
      const uint32_t prev = wv::atomic_min_u32(&c->viol_idx, idx);
      wv::atomic_exch_u32(&c->violation, 1u);
      if (idx < prev) {
        c->viol_info[0] = kind;
        c->viol_info[1] = a.vdbg[4 * wid + 0];
        c->viol_info[2] = a.vdbg[4 * wid + 1];
        c->viol_info[3] = a.vdbg[4 * wid + 2];
        c->viol_info[4] = h.wleaf_lo;
        c->viol_info[5] = h.wleaf_hi;
        c->viol_info[6] = h.index;
        c->viol_info[7] = h.nr;
      }
    }
    return idx;
  }
  // a window too large for one wave goes to a workgroup of o_big: take a queue slot BEFORE touching the state (a full queue
  // leaves the update pending for the next round)
  dev::BigJob *job = nullptr;
  if (kind_strong(kind) && h.wlen > a.big_min && a.jobs) {
    uint32_t slot = 0;
    if (lane == 0) slot = wv::atomic_add_u32(&c->njobs[a.round % 3u], 1u);
    slot = wv::first(slot);
    if (slot >= kBigJobs) return idx;
    job = &a.jobs[slot];
  }
#if defined(PPCSR_SIM)
  if (lane == 0 && getenv("PPCSR_TRACE"))
    fprintf(stderr, "R%u commit idx=%u op=(%u,%u,%u) kind=%u index=%u gap=%u win=(%u,%u) wleaf=[%u,%u] nr=%u\n", a.round, idx, op.src,
            op.dst, op.op, kind, h.index, h.gap, h.wstart, h.wlen, h.wleaf_lo, h.wleaf_hi, h.nr);
#endif
  dev::apply_op(a.v, op, h, lds_wave, &a.stats[wv::block_idx() & (kStatShards - 1)], job);
  const uint32_t me1 = idx + 1u;
  if (kind_strong(kind)) {  // (a duplicate's value overwrite commutes with everything it can be reordered with: no stamp)
    const uint32_t wl = h.wleaf_lo, wh = h.wleaf_hi;
    for (uint32_t leaf = wl + (uint32_t)lane; leaf <= wh; leaf += 64) wv::atomic_max_u32(&a.wstamp[leaf], me1);
  }
  PMA_FOR_EACH_READ_LEAF_H(h, pl, lane, leaf, wv::atomic_max_u32(&a.rstamp[leaf], me1));
  if (kind_real(kind) && op.src < a.v.g.n) {
    if (lane < 2 && ((h.sdep >> lane) & 1u) && op.src + (uint32_t)lane < a.v.g.n) wv::atomic_max_u32(&a.vrs[op.src + (uint32_t)lane], me1);
    if (kind_strong(kind)) {
      const uint32_t ml = h.mv_lo, mh = h.mv_hi;
      for (uint64_t u = (uint64_t)ml + (uint64_t)lane; u <= (uint64_t)mh && ml <= mh; u += 64) wv::atomic_max_u32(&a.vws[u], me1);
    }
  }
  return kMax;
}


// The updates of this round that did not commit append themselves to the round's carry list: one returning atomic per
// WORKGROUP that has any (its waves' indices meet in LDS), on the counter of the XCD the workgroup runs on.
PMA_DEV void carry_append(const OptArgs &a, uint32_t keep, uint32_t *s_keep /* [4] */) {
  if (wv::lane() == 0) s_keep[wv::wave_in_block()] = keep;
  wv::block_sync();
  if (wv::thread_idx() != 0) return;
  OptCtl *c = a.ctl;
  uint32_t v[4], n = 0, m = kMax;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const uint32_t x = s_keep[w];
    if (x != kMax) {
      v[n++] = x;
      m = x < m ? x : m;
    }
  }
  if (n == 0) return;
  const uint32_t rt = a.round % 3u, stripe = wv::xcc_id() % kStripes;
  const uint32_t base = wv::atomic_add_u32(&c->kept[rt][stripe], n);
  if (base + n > a.carry_cap) {
    c->error = 78u;
    return;
  }
  uint32_t *out = ((a.round & 1u) ? a.carry1 : a.carry0) + (uint64_t)stripe * a.carry_cap + base;
  for (uint32_t i = 0; i < n; i++) out[i] = v[i];
  if (m < c->minkept[rt]) wv::atomic_min_u32(&c->minkept[rt], m);  // (the word only ever goes down: a stale read costs one atomic)
}

// (8 waves per SIMD: 53 VGPRs, <= 96 scalar registers, a 3 KB LDS tile per wave.  In round 2 — 78 VGPRs — forcing it meant 56 /
// 44 B of scratch per lane and 113-123 M updates/s against 141 at 6 waves per SIMD.)
PMA_KERNEL void PMA_LAUNCH_BOUNDS(256, 8) o_apply(OptArgs a) {
  PMA_SHARED uint32_t lds[4][3 * kLdsWindow];
  PMA_SHARED uint32_t s_keep[4];
  const uint32_t keep = o_apply_wave<false>(a, lds[wv::wave_in_block()]);
  carry_append(a, keep, s_keep);
}
PMA_KERNEL void o_apply_x(OptArgs a) {
  PMA_SHARED uint32_t lds[4][3 * kLdsWindow];
  PMA_SHARED uint32_t s_keep[4];
  const uint32_t keep = o_apply_wave<true>(a, lds[wv::wave_in_block()]);
  carry_append(a, keep, s_keep);
}

// (Folding the compaction into o_apply's last-finishing workgroup was measured and dropped: the device-scope fences the
// ticket needs make every workgroup write back its XCD's L2, and the round got 3x slower than with a separate launch.)
// markers: empty kernels with distinct names; `set_option("marker", i)` launches k_mark_<i> on the engine's stream so that
// tools/roofline_summary.py can cut sections (timed region, one isolated rebalance, one scan) out of a rocprofv3 kernel
// trace / counter collection of bench.py
#define PMA_MARK(i) PMA_KERNEL void k_mark_##i(uint32_t *p) { if (p && wv::thread_idx() == 0xFFFFFFFFu) *p = i; }
PMA_MARK(0) PMA_MARK(1) PMA_MARK(2) PMA_MARK(3) PMA_MARK(4) PMA_MARK(5) PMA_MARK(6) PMA_MARK(7)
#undef PMA_MARK

// test hook: one workgroup rebalances one window with the big-window routine (leaf counts must be exact)
PMA_KERNEL void k_block_rebalance(View v, uint64_t wstart, uint64_t wlen, Edge *scratch) {
  PMA_SHARED dev::BigShared sh;
  dev::redistribute_block(v, wstart, wlen, scratch, sh);
}

// The round's queued big-window rebalances, one workgroup per window (dev::redistribute_block): they finish the rebalance of
// updates that have already been committed, and must be done before the next round plans.  A launch of its own since the
// compaction it used to share one with is gone: the engine leaves it out while a stream queues no windows (see run_speculative).
PMA_KERNEL void o_big(OptArgs a) {
  PMA_SHARED dev::BigShared sh;
  OptCtl *c = a.ctl;
  const uint32_t f_viol = c->violation, f_err = c->error, f_done = c->done, f_excl = c->excl, f_need = c->need_big;
  uint32_t nj = c->njobs[a.round % 3u];
  // did round a.round run?  The launches queued behind a finished epoch / an exclusive update / a round waiting for this very
  // launch find a flag up and must not touch anything — least of all big_done_round: the round that takes their number later
  // would look served.  (Launched by the host for a waiting round, it comes after the host has lowered need_big.)
  if (f_viol || f_err || f_done || f_excl || f_need) return;
  if (wv::block_idx() == 0 && wv::thread_idx() == 0) c->big_done_round = a.round;
  if (nj == 0) return;
  if (nj > kBigJobs) nj = kBigJobs;
  const uint32_t nwg = wv::grid_dim(), me = wv::block_idx();
  for (uint32_t jb = me; jb < nj; jb += nwg) {
    const dev::BigJob job = a.jobs[jb];
    dev::redistribute_block(a.v, job.wstart, job.wlen, a.bigscratch + (uint64_t)me * a.bigscratch_stride, sh);
    wv::block_sync();  // the shared prefix / table are reused by the next job
  }
}

}  // namespace ppcsr
