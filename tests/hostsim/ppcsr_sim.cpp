// TEST INFRASTRUCTURE ONLY: the engine's kernel + host source compiled for the CPU SIMT emulator
// (sim_runtime.*).  Produces tests/hostsim/libppcsr_sim.so which only tests/test_sim_*.py load, to debug
// kernel logic without a GPU.  The product library is built from csrc/ppcsr_hip.hip by hipcc and contains
// none of this.
#define PPCSR_SIM 1
#include "engine.cc"
#include "capi.cc"

// test hook (emulator build only): every wv::uni / wv::bcast call verifies that its value / source lane really is wave-uniform
extern "C" void ppcsr_sim_check_uniform(int on) { sim::check_uniform = on != 0; }

// test hook (emulator build only): the exact position table evaluated for every element of a window, so the table
// construction and its look-ups can be checked against the oracle's serial fp64 chain without running an engine
extern "C" int ppcsr_sim_chain_positions(uint64_t index, uint64_t len, uint64_t j, uint64_t *out, int *nseg, int *linear_ok) {
  static ppcsr::ChainTable tb;
  ppcsr::build_chain_table(index, len, j, &tb);
  if (nseg) *nseg = tb.nseg;
  if (tb.overflow) return 1;
  int hint = 0;
  for (uint64_t k = 0; k < j; k++) out[k] = ppcsr::chain_pos(&tb, k, &hint);
  // every run of up to 64 consecutive elements that chain_linear_run accepts must reproduce the same positions
  int ok = 1, h3 = 0;
  for (uint64_t k0 = 1; k0 + 1 < j; k0 += 37) {
    const uint64_t cnt = (k0 + 64 <= j - 1) ? 64 : (j - 1 - k0);
    uint64_t A, D;
    int shift;
    if (ppcsr::chain_linear_run(&tb, k0, cnt, &h3, &A, &D, &shift))
      for (uint64_t i = 0; i <= cnt; i++)
        if (((A + i * D) >> shift) != out[k0 + i]) ok = 0;
  }
  // a table published segment by segment (k_rb_scatter's builder): with only the first n segments, every element whose chain step
  // the published part covers (t = j-1-k <= t0 + count of segment n-1) must already get its final position — look-ups and
  // linear runs alike
  if (j >= 2 && tb.nseg > 1) {
    static ppcsr::ChainTable part;
    for (int n = 1; n < tb.nseg; n++) {
      part = tb;
      part.nseg = n;
      const uint64_t covered = tb.seg[n - 1].t0 + tb.seg[n - 1].count;  // inclusive
      const uint64_t k_lo = (j - 1 > covered) ? j - 1 - covered : 1;     // elements k_lo .. j-1 are covered (k = 0 has no chain step)
      int h = -1, h3 = -1;
      const uint64_t stride = (j - k_lo) / 97 + 1;
      for (uint64_t k = k_lo; k < j; k += stride)
        if (ppcsr::chain_pos(&part, k, &h) != out[k]) ok = 0;
      if (ppcsr::chain_pos(&part, k_lo, &h) != out[k_lo]) ok = 0;
      const uint64_t cnt = (k_lo + 64 <= j - 1) ? 64 : (j - 1 - k_lo);
      uint64_t A, D;
      int shift;
      if (cnt && ppcsr::chain_linear_run(&part, k_lo, cnt, &h3, &A, &D, &shift))
        for (uint64_t i = 0; i <= cnt; i++)
          if (((A + i * D) >> shift) != out[k_lo + i]) ok = 0;
    }
  }
  // closed form used by the in-wave rebalance: when the chain is a single segment it must give the same positions
  ppcsr::ChainSeg sg;
  {  // the division-free closed-form test against its statement with the division: same verdict, same segment
    ppcsr::ChainSeg sd;
    const bool a = j >= 2 && ppcsr::chain_single(index, len, j, &sg), b = j >= 2 && ppcsr::chain_single_div(index, len, j, &sd);
    if (a != b) ok = 0;
    if (a && (sg.M0 != sd.M0 || sg.Dfirst != sd.Dfirst || sg.Drest != sd.Drest || sg.shift != sd.shift)) ok = 0;
  }
  if (j >= 2 && ppcsr::chain_single(index, len, j, &sg)) {
    for (uint64_t k = 0; k < j; k++)
      if (ppcsr::chain_single_pos(sg, index, j, k) != out[k]) ok = 0;
    ok |= 2;  // bit 1: the window took the closed form
  }
  if (linear_ok) *linear_ok = ok;
  return 0;
}
