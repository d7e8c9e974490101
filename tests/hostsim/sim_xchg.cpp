// TEST INFRASTRUCTURE ONLY (emulator build): the carrier under pppcsr_exchange_apply — in the product RCCL's grouped ncclSend /
// ncclRecv (engine.cc) — as a mailbox in POSIX shared memory between the emulator processes of a test, so that the exchange
// code itself (capi.cc exchange_run: pack, counts, status step, rows, apply, and what it does when one rank fails) runs past
// one rank on a machine without GPUs.  Same contract as the RCCL carrier: capi_xchg_sendrecv is collective, segment i goes
// to / comes from peer[i], empty segments are skipped on both sides, sends and receives between one pair of ranks match in
// issue order.  "Device" memory is host memory here.
#include <fcntl.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <string>

namespace {
constexpr int kMaxRanks = 16;
constexpr uint32_t kMaxSeg = 4096;
constexpr size_t kBox = 48u << 20;  // outbox bytes per rank
struct Seg {
  int peer;
  uint64_t off, bytes;
};
struct Shared {
  std::atomic<uint32_t> arrived, gen, failed;
  uint32_t nseg[kMaxRanks];
  Seg seg[kMaxRanks][kMaxSeg];
};
}  // namespace
struct ppcsr_xchg {
  Shared *sh = nullptr;
  char *boxes = nullptr;
  size_t bytes = 0;
  int nranks = 0, rank = 0, device = 0;
  std::string name, err;
};
static bool barrier(ppcsr_xchg *x) {
  Shared *s = x->sh;
  const uint32_t g = s->gen.load();
  if (s->arrived.fetch_add(1) + 1 == (uint32_t)x->nranks) {
    s->arrived.store(0);
    s->gen.fetch_add(1);
    return true;
  }
  const auto t0 = std::chrono::steady_clock::now();
  while (s->gen.load() == g) {
    sched_yield();
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;  // a peer never arrived: report, do not hang the test
  }
  return true;
}
int capi_xchg_unique_id(void *out128, std::string *) {
  static std::atomic<uint32_t> serial{0};
  char buf[128];
  memset(buf, 0, sizeof(buf));
  snprintf(buf, sizeof(buf), "/ppcsr_sim_xchg_%d_%u", (int)getpid(), serial.fetch_add(1));
  memcpy(out128, buf, sizeof(buf));
  return 0;
}
int capi_xchg_create(const void *id128, int nranks, int rank, int device, ppcsr_xchg **out, std::string *err) {
  if (nranks < 1 || nranks > kMaxRanks) { if (err) *err = "sim carrier: too many ranks"; return 1; }
  ppcsr_xchg *x = new ppcsr_xchg();
  x->name.assign((const char *)id128, strnlen((const char *)id128, 127));
  x->nranks = nranks;
  x->rank = rank;
  x->device = device;
  x->bytes = sizeof(Shared) + (size_t)nranks * kBox;
  const int fd = shm_open(x->name.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)x->bytes) != 0) { if (err) *err = "sim carrier: shm_open failed"; delete x; return 1; }
  void *m = mmap(nullptr, x->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) { if (err) *err = "sim carrier: mmap failed"; delete x; return 1; }
  x->sh = (Shared *)m;  // (a fresh segment is all zeroes: counters start at 0)
  x->boxes = (char *)m + sizeof(Shared);
  *out = x;
  return 0;
}
int capi_xchg_destroy(ppcsr_xchg *x) {
  if (!x) return 0;
  if (x->sh) munmap((void *)x->sh, x->bytes);
  if (x->rank == 0) shm_unlink(x->name.c_str());
  delete x;
  return 0;
}
int capi_xchg_ranks(ppcsr_xchg *x, int *nranks, int *rank, int *device, void **stream) {
  if (!x) return 1;
  *nranks = x->nranks;
  *rank = x->rank;
  *device = x->device;
  *stream = nullptr;
  return 0;
}
int capi_xchg_sendrecv(ppcsr_xchg *x, uint64_t nseg, const void *const *sptr, const uint64_t *sbytes, const int *speer, void *const *rptr,
                       const uint64_t *rbytes, const int *rpeer) {
  Shared *s = x->sh;
  char *mine = x->boxes + (size_t)x->rank * kBox;
  uint64_t off = 0;
  uint32_t n = 0;
  bool ok = true;
  for (uint64_t i = 0; i < nseg && ok; i++) {
    if (!sbytes[i]) continue;
    if (off + sbytes[i] > kBox || n >= kMaxSeg) { ok = false; break; }
    memcpy(mine + off, sptr[i], sbytes[i]);
    s->seg[x->rank][n++] = Seg{speer[i], off, sbytes[i]};
    off += sbytes[i];
  }
  s->nseg[x->rank] = n;
  if (!ok) s->failed.store(1);
  if (!barrier(x)) { x->err = "sim carrier: a peer did not reach the transfer"; return 1; }
  uint32_t cursor[kMaxRanks] = {0};
  for (uint64_t i = 0; i < nseg && !s->failed.load(); i++) {
    if (!rbytes[i]) continue;
    const int r = rpeer[i];
    uint32_t &c = cursor[r];
    while (c < s->nseg[r] && s->seg[r][c].peer != x->rank) c++;
    if (c >= s->nseg[r] || s->seg[r][c].bytes != rbytes[i]) { s->failed.store(1); break; }
    memcpy(rptr[i], x->boxes + (size_t)r * kBox + s->seg[r][c].off, rbytes[i]);
    c++;
  }
  if (!barrier(x)) { x->err = "sim carrier: a peer did not finish the transfer"; return 1; }
  if (s->failed.load()) { x->err = "sim carrier: segments did not match (or an outbox overflowed)"; return 1; }
  return 0;
}
const char *capi_xchg_error(ppcsr_xchg *x) { return x ? x->err.c_str() : ""; }

// failure injection for the collective-safety tests: the n-th device allocation from now fails (0: off)
int g_sim_fail_alloc = 0;
extern "C" void ppcsr_sim_fail_alloc_after(int n) { g_sim_fail_alloc = n; }
