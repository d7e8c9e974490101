/*
 * TEST INFRASTRUCTURE ONLY — never linked into, imported by, or called from the product path.
 *
 * ppcsr_oracle: a plain-C, single-threaded restatement of the reference's packed-memory-array
 * CSR update path (reference: /root/reference/src/pcsr/PCSR.cpp, src/pppcsr/PPPCSR.cpp).
 * It exists so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check the
 * HIP engine bit-for-bit on machines where the reference tree is absent (the GPU box).
 *
 * Parity pin: this restatement is validated against the UNMODIFIED reference compiled by
 * oracle/Makefile into oracle/_ref/libref_pcsr.so (tests/test_oracle_vs_ref.py, run wherever
 * /root/reference exists) and against the golden fixtures in tests/golden/ that were generated
 * from that same reference build (tests/golden/make_golden.py).
 *
 * Semantics = the reference driven in stream order on one thread (the only
 * deterministic mode of the reference).  The lock *bookkeeping* of acquire_insert_locks
 * (min_node / max_node / tries, PCSR.cpp:949-1134) is emulated because it decides which
 * code path the insert takes and therefore the final layout.
 */
#ifndef PPCSR_ORACLE_H
#define PPCSR_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint32_t src, dest, value; } po_edge;              /* PCSR.h:30-35 */
typedef struct { uint32_t beginning, end, num_neighbors; } po_node; /* PCSR.h:18-23 */
typedef struct { uint32_t src, dst, op; } po_op;                    /* op: 0 = delete, else add with value = op */

typedef struct {
  uint64_t redistribute_calls;  /* PCSR.cpp:222 invocations                       */
  uint64_t redistribute_slots;  /* sum of len over those invocations              */
  uint64_t double_calls, half_calls;
  uint64_t slide_right_calls, slide_steps, slide_left_calls;
  uint64_t global_path;         /* NEED_GLOBAL_WRITE outcomes                     */
  uint64_t duplicates, not_found, search_probes;
  uint64_t ops_add, ops_del;
} po_stats;

typedef struct po_pcsr po_pcsr;

po_pcsr *po_create(uint32_t init_n, uint32_t src_n, int lock_search);
po_pcsr *po_clone(const po_pcsr *p); /* test convenience: independent copy of the state */
void po_destroy(po_pcsr *p);
void po_add_edge(po_pcsr *p, uint32_t src, uint32_t dest, uint32_t value);
void po_remove_edge(po_pcsr *p, uint32_t src, uint32_t dest);
void po_add_node(po_pcsr *p);
int po_edge_exists(po_pcsr *p, uint32_t src, uint32_t dest);
void po_apply(po_pcsr *p, const po_op *ops, uint64_t n);
uint64_t po_get_n(const po_pcsr *p);
void po_geometry(const po_pcsr *p, uint64_t *N, int *logN, int *H);
void po_export(const po_pcsr *p, uint32_t *items3, uint32_t *nodes3);
uint64_t po_get_neighbourhood(const po_pcsr *p, int src, int *out, uint64_t cap);
void po_debug_redistribute(po_pcsr *p, uint64_t index, uint64_t len);
void po_set_num_neighbors(po_pcsr *p, uint32_t v, uint32_t nn);
po_pcsr *po_import_state(uint64_t N, const uint32_t *items3, uint32_t n, const uint32_t *nodes3, int lock_search);
void po_get_stats(const po_pcsr *p, po_stats *out);
void po_reset_stats(po_pcsr *p);
/* exact redistribute target positions (PCSR.cpp:237-247) for a window; out[k] = slot of element k */
void po_redistribute_positions(uint64_t index, uint64_t len, uint64_t j, uint64_t *out);

/* PPPCSR: vertex-range partitioning (PPPCSR.cpp:13-34, 58-66) */
typedef struct po_pppcsr po_pppcsr;
po_pppcsr *pop_create(uint32_t init_n, uint32_t src_n, int lock_search, int num_domains, int parts_per_domain);
void pop_destroy(po_pppcsr *p);
uint64_t pop_num_partitions(const po_pppcsr *p);
uint64_t pop_get_partition(const po_pppcsr *p, uint64_t v);
uint64_t pop_partition_start(const po_pppcsr *p, uint64_t part);
po_pcsr *pop_partition(po_pppcsr *p, uint64_t part);
void pop_apply(po_pppcsr *p, const po_op *ops, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif
