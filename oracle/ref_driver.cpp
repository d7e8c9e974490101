// TEST INFRASTRUCTURE ONLY — not part of the product path.
//
// Thin C-ABI driver around the *unmodified* reference sources
// (/root/reference/src/pcsr/PCSR.cpp, /root/reference/src/pppcsr/PPPCSR.cpp), compiled where they
// lie by oracle/Makefile into oracle/_ref/libref_pcsr.so.  Nothing from the reference is copied
// into this repository: this file only *calls* the reference's public API
// (PCSR.h:64-124, PPPCSR.h:11-60) and reads its public `edges` member (PCSR.h:67).
//
// Used to (1) pin oracle/ppcsr_oracle.c against the real reference, (2) generate tests/golden/,
// (3) serve as the "reference" CPU baseline in bench.py.  The reference is driven strictly in
// stream order on one thread (the only deterministic mode).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <sstream>

#include "PCSR.h"
#include "PPPCSR.h"
// the reference's graph-algorithm clients (src/utility/bfs.h:15-36, src/utility/pagerank.h:15-29): header-only templates,
// instantiated below on the reference PCSR so that their results can be stored as golden vectors
#include "bfs.h"
#include "pagerank.h"

namespace {
// The reference prints on every resize / missing delete; keep stdout clean for callers.
struct Quiet {
  std::streambuf *old;
  std::ostringstream sink;
  Quiet() : old(std::cout.rdbuf(sink.rdbuf())) {}
  ~Quiet() { std::cout.rdbuf(old); }
};
}  // namespace

extern "C" {

struct ref_op {
  uint32_t src, dst, op;  // op: 1 = add (value 1), 0 = delete; values >1: add with value=op
};

void *ref_create(uint32_t init_n, uint32_t src_n, int lock_search) {
  Quiet q;
  return new PCSR(init_n, src_n, lock_search != 0, -1);
}
void ref_destroy(void *h) {
  Quiet q;
  delete static_cast<PCSR *>(h);
}
void ref_add_edge(void *h, uint32_t s, uint32_t d, uint32_t v) {
  Quiet q;
  static_cast<PCSR *>(h)->add_edge(s, d, v);
}
void ref_remove_edge(void *h, uint32_t s, uint32_t d) {
  Quiet q;
  static_cast<PCSR *>(h)->remove_edge(s, d);
}
void ref_add_node(void *h) {
  Quiet q;
  static_cast<PCSR *>(h)->add_node();
}
int ref_edge_exists(void *h, uint32_t s, uint32_t d) { return static_cast<PCSR *>(h)->edge_exists(s, d) ? 1 : 0; }
void ref_apply(void *h, const ref_op *ops, uint64_t n) {
  Quiet q;
  PCSR *p = static_cast<PCSR *>(h);
  for (uint64_t i = 0; i < n; i++) {
    if (ops[i].op)
      p->add_edge(ops[i].src, ops[i].dst, ops[i].op);
    else
      p->remove_edge(ops[i].src, ops[i].dst);
  }
}
uint64_t ref_get_n(void *h) { return static_cast<PCSR *>(h)->get_n(); }
void ref_geometry(void *h, uint64_t *N, int *logN, int *H) {
  PCSR *p = static_cast<PCSR *>(h);
  *N = p->edges.N;
  *logN = p->edges.logN;
  *H = p->edges.H;
}
// items: N*3 u32 (src,dest,value); nodes: n*3 u32 (beginning,end,num_neighbors)
void ref_export(void *h, uint32_t *items, uint32_t *nodes) {
  PCSR *p = static_cast<PCSR *>(h);
  if (items) memcpy(items, p->edges.items, p->edges.N * sizeof(edge_t));
  if (nodes) {
    uint64_t n = p->get_n();
    for (uint64_t i = 0; i < n; i++) {
      const node_t &nd = p->getNode((int)i);
      nodes[3 * i] = nd.beginning;
      nodes[3 * i + 1] = nd.end;
      nodes[3 * i + 2] = nd.num_neighbors;
    }
  }
}
uint64_t ref_get_neighbourhood(void *h, int src, int *out, uint64_t cap) {
  std::vector<int> v = static_cast<PCSR *>(h)->get_neighbourhood(src);
  uint64_t m = v.size() < cap ? v.size() : cap;
  if (out && m) memcpy(out, v.data(), m * sizeof(int));
  return v.size();
}

// bfs(graph, start): out[n] levels, UINT32_MAX = unreachable.  pagerank(graph, node_values): one push step in fp32.
void ref_bfs(void *h, uint32_t start, uint32_t *out) {
  std::vector<uint32_t> r = bfs(*static_cast<PCSR *>(h), start);
  if (!r.empty()) memcpy(out, r.data(), r.size() * sizeof(uint32_t));
}
void ref_pagerank(void *h, const float *node_values, float *out) {
  PCSR *p = static_cast<PCSR *>(h);
  std::vector<float> vals(node_values, node_values + p->get_n());
  std::vector<float> r = pagerank(*p, vals);
  if (!r.empty()) memcpy(out, r.data(), r.size() * sizeof(float));
}

// ---- PPPCSR (vertex-range partitioned) -------------------------------------------------------
void *refp_create(uint32_t init_n, uint32_t src_n, int lock_search, int num_domains, int parts_per_domain) {
  Quiet q;
  return new PPPCSR(init_n, src_n, lock_search != 0, num_domains, parts_per_domain, false);
}
void refp_destroy(void *h) {
  Quiet q;
  delete static_cast<PPPCSR *>(h);
}
void refp_apply(void *h, const ref_op *ops, uint64_t n) {
  Quiet q;
  PPPCSR *p = static_cast<PPPCSR *>(h);
  for (uint64_t i = 0; i < n; i++) {
    if (ops[i].op)
      p->add_edge(ops[i].src, ops[i].dst, ops[i].op);
    else
      p->remove_edge(ops[i].src, ops[i].dst);
  }
}
uint64_t refp_get_partition(void *h, uint64_t v) { return static_cast<PPPCSR *>(h)->get_partiton(v); }
uint64_t refp_get_n(void *h) { return static_cast<PPPCSR *>(h)->get_n(); }
int refp_edge_exists(void *h, uint32_t s, uint32_t d) { return static_cast<PPPCSR *>(h)->edge_exists(s, d) ? 1 : 0; }
void refp_get_node(void *h, int id, uint32_t *out3) {
  const node_t &nd = static_cast<PPPCSR *>(h)->getNode(id);
  out3[0] = nd.beginning;
  out3[1] = nd.end;
  out3[2] = nd.num_neighbors;
}
uint64_t refp_get_neighbourhood(void *h, int src, int *out, uint64_t cap) {
  std::vector<int> v = static_cast<PPPCSR *>(h)->get_neighbourhood(src);
  uint64_t m = v.size() < cap ? v.size() : cap;
  if (out && m) memcpy(out, v.data(), m * sizeof(int));
  return v.size();
}

}  // extern "C"
