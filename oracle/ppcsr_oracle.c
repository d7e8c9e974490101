/*
 * TEST INFRASTRUCTURE ONLY — see ppcsr_oracle.h.  CPU restatement of the reference PMA path,
 * used exclusively as the checker.  Each function cites the reference lines it follows
 * (paths relative to /root/reference/src/).  Compile with -ffp-contract=off: the redistribute
 * position chain and the density bounds must round exactly like the reference's x86-64 build.
 */
#include "ppcsr_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PO_MAX UINT32_MAX
#define PO_NULL_SRC UINT32_MAX

struct po_pcsr {
  uint64_t N;
  int logN, H;
  po_edge *items;
  po_node *nodes;
  uint64_t n, ncap;
  int lock_search;
  po_stats st;
};

/* pcsr/PCSR.cpp:28-33 (bsr: index of the highest set bit) */
static inline int po_bsr(uint64_t w) { return 63 - __builtin_clzll(w); }

static inline int po_is_null(const po_edge *e) { return e->value == 0; }                     /* PCSR.h:57-60 */
static inline int po_is_sentinel(const po_edge *e) { return e->dest == PO_MAX || e->value == PO_MAX; } /* PCSR.cpp:64 */
static inline void po_set_null(po_edge *e) { e->src = PO_NULL_SRC; e->dest = 0; e->value = 0; }

/* pcsr/PCSR.cpp:68-73 */
static void po_resize_geometry(po_pcsr *p, uint64_t newN) {
  p->N = newN;
  p->logN = 1 << po_bsr((uint64_t)(po_bsr(newN) * 2 + 1));
  p->H = po_bsr(newN / (uint64_t)p->logN);
}

/* pcsr/PCSR.cpp:126-133 — number of non-null slots; density = count / len */
static int64_t po_count(const po_pcsr *p, int64_t index, int64_t len) {
  int64_t full = 0;
  for (int64_t i = index; i < index + len; i++) full += !po_is_null(&p->items[i]);
  return full;
}
static double po_density(const po_pcsr *p, int64_t index, int64_t len) { return (double)po_count(p, index, len) / (double)len; }

/* pcsr/PCSR.cpp:156-165 */
static double po_lower(const po_pcsr *p, int depth) { return 1.0 / 4.0 - ((0.125 * depth) / p->H); }
static double po_upper(const po_pcsr *p, int depth) { return 3.0 / 4.0 + ((.25 * depth) / p->H); }

static inline int64_t po_find_node(int64_t index, int64_t len) { return (index / len) * len; } /* PCSR.cpp:59 */
static inline int64_t po_find_leaf(const po_pcsr *p, int64_t index) { return (index / p->logN) * p->logN; } /* :393 */

/* pcsr/PCSR.cpp:168-183 */
static void po_fix_sentinel(po_pcsr *p, const po_edge *s, uint32_t in) {
  if (!po_is_sentinel(s)) return;
  uint32_t v = s->value;
  if (v == PO_MAX) {
    v = 0;
  } else {
    p->nodes[v - 1].end = in;
  }
  p->nodes[v].beginning = in;
  if (v == p->n - 1) p->nodes[v].end = (uint32_t)(p->N - 1);
}

/* pcsr/PCSR.cpp:237-247 — the serial fp64 position chain, verbatim arithmetic */
void po_redistribute_positions(uint64_t index, uint64_t len, uint64_t j, uint64_t *out) {
  if (j == 0) return;
  const double step = (double)len / (double)j;
  double index_d = (double)index + (double)(j - 1) * step;
  for (uint64_t i = j - 1; i > 0; i--) {
    out[i] = (uint64_t)index_d;
    index_d -= step;
  }
  out[0] = index;
}

/* pcsr/PCSR.cpp:222-249 */
static void po_redistribute(po_pcsr *p, int64_t index, int64_t len) {
  p->st.redistribute_calls++;
  p->st.redistribute_slots += (uint64_t)len;
  po_edge *it = p->items;
  uint64_t j = 0;
  const uint64_t end = (uint64_t)(index + len);
  for (uint64_t i = (uint64_t)index; i < end; i++) {
    it[index + j] = it[i];
    j += !po_is_null(&it[index + j]);
  }
  for (uint64_t i = index + j; i < end; i++) po_set_null(&it[i]);
  const double step = (double)len / (double)j;
  double index_d = (double)index + (double)(j - 1) * step;
  for (uint64_t i = index + j - 1; i > (uint64_t)index && j > 0; i--) {
    const uint64_t in = (uint64_t)index_d;
    po_edge t = it[in];
    it[in] = it[i];
    it[i] = t;
    po_fix_sentinel(p, &it[in], (uint32_t)in);
    index_d -= step;
  }
  po_fix_sentinel(p, &it[index], (uint32_t)index);
}

/* pcsr/PCSR.cpp:251-282 */
static void po_double_list(po_pcsr *p) {
  p->st.double_calls++;
  po_resize_geometry(p, p->N * 2);
  p->items = (po_edge *)realloc(p->items, p->N * sizeof(po_edge));
  if (!p->items) { fprintf(stderr, "po: allocation failed\n"); exit(1); }
  for (uint64_t i = p->N / 2; i < p->N; i++) { p->items[i].value = 0; p->items[i].dest = 0; }
  po_redistribute(p, 0, (int64_t)p->N);
}

/* pcsr/PCSR.cpp:284-320 */
static void po_half_list(po_pcsr *p) {
  p->st.half_calls++;
  po_resize_geometry(p, p->N / 2);
  uint64_t j = 0;
  for (uint64_t i = 0; i < p->N * 2; i++)
    if (!po_is_null(&p->items[i])) p->items[j++] = p->items[i];
  for (; j < p->N; j++) { p->items[j].value = 0; p->items[j].dest = 0; }
  p->items = (po_edge *)realloc(p->items, p->N * sizeof(po_edge));
  po_redistribute(p, 0, (int64_t)p->N);
}

static void po_slide_left(po_pcsr *p, int64_t index, uint32_t src);

/* pcsr/PCSR.cpp:326-355 */
static int po_slide_right(po_pcsr *p, int64_t index, uint32_t src) {
  p->st.slide_right_calls++;
  int rval = 0;
  po_edge el = p->items[index];
  po_set_null(&p->items[index]);
  index++;
  while (index < (int64_t)p->N && !po_is_null(&p->items[index])) {
    po_edge temp = p->items[index];
    p->items[index] = el;
    if (!po_is_null(&el)) po_fix_sentinel(p, &el, (uint32_t)index);
    el = temp;
    index++;
    p->st.slide_steps++;
  }
  if (!po_is_null(&el)) po_fix_sentinel(p, &el, (uint32_t)index);
  if (index == (int64_t)p->N) {
    index--;
    po_slide_left(p, index, src);
    rval = -1;
  }
  p->items[index] = el;
  return rval;
}

/* pcsr/PCSR.cpp:360-390 */
static void po_slide_left(po_pcsr *p, int64_t index, uint32_t src) {
  p->st.slide_left_calls++;
  po_edge el = p->items[index];
  po_set_null(&p->items[index]);
  index--;
  while (index >= 0 && !po_is_null(&p->items[index])) {
    po_edge temp = p->items[index];
    p->items[index] = el;
    if (!po_is_null(&el)) po_fix_sentinel(p, &el, (uint32_t)index);
    el = temp;
    index--;
  }
  if (index == -1) {
    po_double_list(p);
    po_slide_right(p, 0, src);
    index = 0;
  }
  if (!po_is_null(&el)) po_fix_sentinel(p, &el, (uint32_t)index);
  p->items[index] = el;
}

/* pcsr/PCSR.cpp:427-502 — gap-aware lower bound; probe order mid, mid+1, mid-1, mid+2, ... */
static uint32_t po_binary_search(po_pcsr *p, uint32_t dest, uint32_t start, uint32_t end) {
  const po_edge *it = p->items;
  while (start + 1 < end) {
    const uint32_t mid = (start + end) / 2;
    po_edge item = it[mid];
    p->st.search_probes++;
    uint32_t change = 1, check = mid;
    int flag = 1;
    while (po_is_null(&item) && flag) {
      flag = 0;
      check = mid + change;
      if (check < end) {
        flag = 1;
        item = it[check];
        p->st.search_probes++;
        if (!po_is_null(&item)) break;
      }
      check = mid - change;
      if (check >= start) {
        flag = 1;
        item = it[check];
        p->st.search_probes++;
      }
      change++;
    }
    if (po_is_null(&item) || start == check || end == check) {
      if (!po_is_null(&item) && start == check && dest <= item.dest) return check;
      return mid;
    }
    if (dest == item.dest) return check;
    if (dest < item.dest) end = check; else start = check;
  }
  if (end < start) start = end;
  if (dest <= it[start].dest && !po_is_null(&it[start])) return start;
  return end;
}

/* ---- insert planning: emulation of acquire_insert_locks, pcsr/PCSR.cpp:949-1134 ------------- */
enum { PO_PLAN_OK = 0, PO_PLAN_GLOBAL = -1 };
typedef struct { int has_info; int double_list; int64_t max_len; int64_t node_index_final; } po_plan;

static int po_plan_insert(po_pcsr *p, uint32_t index, int64_t left_bound, int tries, po_plan *out) {
  out->has_info = 0;
  if (tries > 3) return PO_PLAN_GLOBAL;                                   /* :952-955 */
  const int64_t logN = p->logN;
  int64_t node_index = po_find_leaf(p, index);
  int level = p->H;
  int64_t len = logN;
  int64_t node_id = node_index / logN;
  int64_t min_node = node_id, max_node = node_id;
  if (left_bound != -1) {
    if (left_bound < min_node) min_node = left_bound;                     /* :963-974 */
  } else if (node_id > 0 && !p->lock_search) {
    min_node = node_id - 1;                                               /* :976-979 */
  }
  if ((uint64_t)index == p->N - 1 && !po_is_null(&p->items[index])) return PO_PLAN_GLOBAL; /* :992-997 */

  /* leaf would become full: re-align to the 2-leaf parent but keep len (":1012-1023", quirk kept) */
  if (po_density(p, node_index, len) + (1.0 / (double)len) == 1) {
    int64_t new_idx = po_find_node(node_index, 2 * len);
    int64_t new_id = new_idx / logN;
    if (new_idx == node_index && new_id > max_node) {
      max_node = new_id;
    } else if (new_id < min_node) {
      return po_plan_insert(p, index, new_id, tries + 1, out);
    }
    node_index = new_idx;
  }
  double upper = po_upper(p, level);
  double density = po_density(p, node_index, len) + (1.0 / (double)len);
  while (density >= upper) {                                              /* :1028-1061 */
    len *= 2;
    if ((uint64_t)len <= p->N) {
      level--;
      int64_t new_idx = po_find_node(node_index, len);
      if (new_idx < node_index) {
        int64_t new_id = new_idx / logN;
        if (new_id < min_node) return po_plan_insert(p, index, new_id, tries + 1, out);
        node_index = new_idx;
      } else {
        int64_t endn = po_find_leaf(p, new_idx + len) / logN;
        node_index = new_idx;
        if (endn - 1 > max_node) max_node = endn - 1;
      }
      upper = po_upper(p, level);
      density = po_density(p, node_index, len) + (1.0 / (double)len);
    } else {
      out->has_info = 1;
      out->double_list = 1;
      return PO_PLAN_GLOBAL;
    }
  }
  {
    int64_t new_idx = po_find_node(node_index, len);                      /* :1062-1078 */
    if (new_idx < node_index) {
      int64_t new_id = new_idx / logN;
      if (new_id < min_node) return po_plan_insert(p, index, new_id, tries + 1, out);
    } else {
      int64_t endn = po_find_leaf(p, new_idx + len) / logN;
      if (endn - 1 > max_node) max_node = endn - 1;
    }
    node_index = new_idx;
  }
  out->has_info = 1;
  out->double_list = 0;
  out->max_len = len;
  out->node_index_final = node_index;

  /* leaves the slide will cross (:1085-1132); only the left walk can force a re-plan */
  len = logN;
  node_index = po_find_leaf(p, index);
  if (!po_is_null(&p->items[index])) {
    int64_t curr_ind = (int64_t)index + 1;
    while (curr_ind < (int64_t)p->N && !po_is_null(&p->items[curr_ind])) curr_ind++;
    if (curr_ind == (int64_t)p->N) {
      curr_ind = index;
      int64_t curr_node = node_index / logN;
      int64_t curr_node_idx = node_index;
      while (curr_ind >= 0 && !po_is_null(&p->items[curr_ind])) {
        if (--curr_ind >= 0 && curr_ind < curr_node_idx) {
          curr_node_idx = po_find_leaf(p, curr_ind);
          curr_node--;
          if (curr_node < min_node) return po_plan_insert(p, index, curr_node, tries + 1, out);
        }
      }
      if (curr_ind == -1) { out->has_info = 0; return PO_PLAN_GLOBAL; }
    }
  }
  (void)max_node;
  return PO_PLAN_OK;
}

/* pcsr/PCSR.cpp:519-595 */
static void po_insert(po_pcsr *p, uint32_t index, po_edge elem, uint32_t src, const po_plan *info) {
  int64_t node_index = po_find_leaf(p, index);
  int level = p->H;
  int64_t len = p->logN;
  if (!po_is_null(&p->items[index])) {
    if (!po_is_sentinel(&elem) && p->items[index].dest == elem.dest) {
      p->items[index].value = elem.value;
      p->st.duplicates++;
      return;
    }
    if ((uint64_t)index == p->N - 1) {
      po_double_list(p);
      po_node nd = p->nodes[src];
      uint32_t loc = po_binary_search(p, elem.dest, nd.beginning + 1, nd.end);
      po_insert(p, loc, elem, src, NULL);
      return;
    } else {
      if (po_slide_right(p, index, src) == -1) {
        index -= 1;
        po_slide_left(p, index, src);
      }
    }
  }
  p->items[index] = elem;
  double density = po_density(p, node_index, len);
  if (density == 1) {
    node_index = po_find_node(node_index, len * 2);
    po_redistribute(p, node_index, len * 2);
  } else {
    po_redistribute(p, node_index, len);
  }
  double upper = po_upper(p, level);
  density = po_density(p, node_index, len);
  if (info != NULL && info->has_info) {
    if (info->double_list) { po_double_list(p); return; }
    len = info->max_len;
    node_index = info->node_index_final;
  } else {
    while (density >= upper) {
      len *= 2;
      if ((uint64_t)len <= p->N) {
        level--;
        node_index = po_find_node(node_index, len);
        upper = po_upper(p, level);
        density = po_density(p, node_index, len);
      } else {
        po_double_list(p);
        return;
      }
    }
  }
  if (len > p->logN) po_redistribute(p, node_index, len);
}

/* pcsr/PCSR.cpp:597-630 */
static void po_remove(po_pcsr *p, uint32_t index, po_edge elem) {
  int64_t node_index = po_find_leaf(p, index);
  int level = p->H;
  int64_t len = p->logN;
  if (po_is_null(&p->items[index]) || po_is_sentinel(&elem) || p->items[index].dest != elem.dest) return;
  p->items[index].value = 0;
  p->items[index].dest = 0;
  po_redistribute(p, node_index, len);
  double lower = po_lower(p, level);
  double density = po_density(p, node_index, len);
  while (density < lower) {
    len *= 2;
    if ((uint64_t)len <= p->N) {
      level--;
      node_index = po_find_node(node_index, len);
      lower = po_lower(p, level);
      density = po_density(p, node_index, len);
    } else {
      po_half_list(p);
      return;
    }
  }
  po_redistribute(p, node_index, len);
}

/* pcsr/PCSR.cpp:706, :1374-1445 — sequential specialisation of add_edge_parallel */
void po_add_edge(po_pcsr *p, uint32_t src, uint32_t dest, uint32_t value) {
  if (value == 0 || src >= p->n) return;
  p->st.ops_add++;
  po_edge e = {src, dest, value};
  uint32_t beginning = p->nodes[src].beginning, end = p->nodes[src].end;
  p->nodes[src].num_neighbors++;
  uint32_t loc = po_binary_search(p, dest, beginning + 1, end);
  po_plan plan;
  int rc = po_plan_insert(p, loc, -1, 0, &plan);
  if (rc == PO_PLAN_GLOBAL) {
    p->st.global_path++;
    loc = po_binary_search(p, dest, p->nodes[src].beginning + 1, p->nodes[src].end); /* :1436 */
    po_insert(p, loc, e, src, plan.has_info ? &plan : NULL);
  } else {
    po_insert(p, loc, e, src, &plan);
  }
}

/* pcsr/PCSR.cpp:709-773 with acquire_remove_locks :1147-1232 reduced to its sequential outcome */
void po_remove_edge(po_pcsr *p, uint32_t src, uint32_t dest) {
  if (src >= p->n) return; /* reference: unchecked UB */
  p->st.ops_del++;
  po_edge e = {src, dest, 1};
  uint32_t loc = po_binary_search(p, dest, p->nodes[src].beginning + 1, p->nodes[src].end);
  p->nodes[src].num_neighbors--;                                                 /* :747, before the check */
  if (po_is_null(&p->items[loc]) || po_is_sentinel(&e) || p->items[loc].dest != dest) { /* :1179-1189 */
    p->st.not_found++;
    return;
  }
  po_remove(p, loc, e);
}

/* pcsr/PCSR.cpp:860-869 */
int po_edge_exists(po_pcsr *p, uint32_t src, uint32_t dest) {
  po_node nd = p->nodes[src];
  uint32_t loc = po_binary_search(p, dest, nd.beginning + 1, nd.end);
  po_edge e = p->items[loc];
  return !po_is_null(&e) && !po_is_sentinel(&e) && e.dest == dest;
}

/* pcsr/PCSR.cpp:681-703 */
void po_add_node(po_pcsr *p) {
  uint64_t len = p->n;
  po_edge s = {(uint32_t)len, PO_MAX, (uint32_t)len};
  po_node nd;
  if (len > 0) {
    nd.beginning = p->nodes[len - 1].end;
    nd.end = nd.beginning + 1;
  } else {
    nd.beginning = 0;
    nd.end = 1;
    s.value = PO_MAX;
  }
  nd.num_neighbors = 0;
  if (p->n == p->ncap) {
    p->ncap = p->ncap ? p->ncap * 2 : 16;
    p->nodes = (po_node *)realloc(p->nodes, p->ncap * sizeof(po_node));
  }
  p->nodes[p->n++] = nd;
  po_insert(p, nd.beginning, s, (uint32_t)(p->n - 1), NULL);
}

/* pcsr/PCSR.cpp:775-838 */
po_pcsr *po_create(uint32_t init_n, uint32_t src_n, int lock_search) {
  po_pcsr *p = (po_pcsr *)calloc(1, sizeof(po_pcsr));
  p->lock_search = lock_search;
  uint32_t m = init_n + src_n;
  if (m < 1024u) m = 1024u;
  po_resize_geometry(p, (uint64_t)2 << po_bsr(m));
  p->items = (po_edge *)malloc(p->N * sizeof(po_edge));
  p->n = src_n;
  p->ncap = src_n ? src_n : 1;
  p->nodes = (po_node *)calloc(p->ncap, sizeof(po_node));
  double index_d = 0.0;
  const double step = ((double)p->N) / src_n;
  int in = 0;
  for (uint32_t i = 0; i < src_n; i++) {
    p->nodes[i].beginning = (i == 0) ? 0 : p->nodes[i - 1].end;
    index_d += step;
    in = (int)index_d;
    p->nodes[i].end = (uint32_t)in;
    p->nodes[i].num_neighbors = 0;
  }
  if (src_n != 0) p->nodes[src_n - 1].end = (uint32_t)(p->N - 1);
  index_d = 0.0;
  in = 0;
  uint32_t current = 0;
  for (int64_t i = 0; i < (int64_t)p->N; i++) {
    if (i == in && current < src_n) {
      p->items[i].src = current;
      p->items[i].dest = PO_MAX;
      p->items[i].value = (i == 0) ? PO_MAX : current;
      current++;
      index_d += step;
      in = (int)index_d;
    } else {
      po_set_null(&p->items[i]);
    }
  }
  return p;
}

/* test convenience (no reference equivalent): an independent copy of the whole state, so that several update batches can
 * be replayed from one loaded core graph */
po_pcsr *po_clone(const po_pcsr *p) {
  po_pcsr *q = (po_pcsr *)malloc(sizeof(po_pcsr));
  *q = *p;
  q->items = (po_edge *)malloc(p->N * sizeof(po_edge));
  memcpy(q->items, p->items, p->N * sizeof(po_edge));
  q->nodes = (po_node *)calloc(p->ncap, sizeof(po_node));
  memcpy(q->nodes, p->nodes, p->n * sizeof(po_node));
  return q;
}

void po_destroy(po_pcsr *p) {
  if (!p) return;
  free(p->items);
  free(p->nodes);
  free(p);
}

void po_apply(po_pcsr *p, const po_op *ops, uint64_t n) {
  for (uint64_t i = 0; i < n; i++) {
    if (ops[i].op) po_add_edge(p, ops[i].src, ops[i].dst, ops[i].op);
    else po_remove_edge(p, ops[i].src, ops[i].dst);
  }
}

uint64_t po_get_n(const po_pcsr *p) { return p->n; }
void po_geometry(const po_pcsr *p, uint64_t *N, int *logN, int *H) { *N = p->N; *logN = p->logN; *H = p->H; }
void po_export(const po_pcsr *p, uint32_t *items3, uint32_t *nodes3) {
  if (items3) memcpy(items3, p->items, p->N * sizeof(po_edge));
  if (nodes3) memcpy(nodes3, p->nodes, p->n * sizeof(po_node));
}
/* pcsr/PCSR.cpp:901-912 */
uint64_t po_get_neighbourhood(const po_pcsr *p, int src, int *out, uint64_t cap) {
  uint64_t k = 0;
  if (src >= 0 && (uint64_t)src < p->n) {
    for (int64_t i = (int64_t)p->nodes[src].beginning + 1; i < (int64_t)p->nodes[src].end; i++) {
      if (p->items[i].value != 0) {
        if (out && k < cap) out[k] = (int)p->items[i].dest;
        k++;
      }
    }
  }
  return k;
}
/* test hook: run the reference's redistribute() on an arbitrary aligned window */
void po_debug_redistribute(po_pcsr *p, uint64_t index, uint64_t len) { po_redistribute(p, (int64_t)index, (int64_t)len); }
/* test hook (no reference equivalent): an oracle that starts from a given raw state — edges[N] and nodes[n] as exported —
 * so that updates applied after a non-parity step (bulk build, repartitioning) can still be checked one by one */
po_pcsr *po_import_state(uint64_t N, const uint32_t *items3, uint32_t n, const uint32_t *nodes3, int lock_search) {
  po_pcsr *p = (po_pcsr *)calloc(1, sizeof(po_pcsr));
  p->lock_search = lock_search;
  po_resize_geometry(p, N);
  p->items = (po_edge *)malloc(p->N * sizeof(po_edge));
  memcpy(p->items, items3, p->N * sizeof(po_edge));
  p->n = n;
  p->ncap = n ? n : 1;
  p->nodes = (po_node *)calloc(p->ncap, sizeof(po_node));
  if (n) memcpy(p->nodes, nodes3, (size_t)n * sizeof(po_node));
  return p;
}
/* test hook for the repartitioning tests (no reference equivalent): overwrite a vertex's call counter */
void po_set_num_neighbors(po_pcsr *p, uint32_t v, uint32_t nn) { if (v < p->n) p->nodes[v].num_neighbors = nn; }
void po_get_stats(const po_pcsr *p, po_stats *out) { *out = p->st; }
void po_reset_stats(po_pcsr *p) { memset(&p->st, 0, sizeof(p->st)); }

/* ---- PPPCSR: pppcsr/PPPCSR.cpp:13-34, 58-66 --------------------------------------------------- */
struct po_pppcsr {
  uint64_t nparts;
  po_pcsr **parts;
  uint64_t *distribution;
};

po_pppcsr *pop_create(uint32_t init_n, uint32_t src_n, int lock_search, int num_domains, int parts_per_domain) {
  (void)src_n;
  po_pppcsr *pp = (po_pppcsr *)calloc(1, sizeof(po_pppcsr));
  uint64_t P = (uint64_t)num_domains * (uint64_t)parts_per_domain;
  pp->nparts = P;
  pp->parts = (po_pcsr **)calloc(P, sizeof(po_pcsr *));
  pp->distribution = (uint64_t *)calloc(P, sizeof(uint64_t));
  uint64_t partitionSize = init_n / P; /* std::ceil of an integer division, :20 */
  for (uint64_t k = 0; k < P; k++) {
    if (k > 0) pp->distribution[k] = pp->distribution[k - 1] + partitionSize;
    uint64_t size = partitionSize;
    if (k == P - 1) size = init_n - k * partitionSize;
    pp->parts[k] = po_create((uint32_t)size, (uint32_t)size, lock_search);
  }
  return pp;
}
void pop_destroy(po_pppcsr *p) {
  if (!p) return;
  for (uint64_t k = 0; k < p->nparts; k++) po_destroy(p->parts[k]);
  free(p->parts);
  free(p->distribution);
  free(p);
}
uint64_t pop_num_partitions(const po_pppcsr *p) { return p->nparts; }
uint64_t pop_get_partition(const po_pppcsr *p, uint64_t v) {
  for (uint64_t i = 1; i < p->nparts; i++)
    if (p->distribution[i] > v) return i - 1;
  return p->nparts - 1;
}
uint64_t pop_partition_start(const po_pppcsr *p, uint64_t part) { return p->distribution[part]; }
po_pcsr *pop_partition(po_pppcsr *p, uint64_t part) { return p->parts[part]; }
void pop_apply(po_pppcsr *p, const po_op *ops, uint64_t n) {
  for (uint64_t i = 0; i < n; i++) {
    uint64_t k = pop_get_partition(p, ops[i].src);
    uint32_t ls = (uint32_t)(ops[i].src - p->distribution[k]);
    if (ops[i].op) po_add_edge(p->parts[k], ls, ops[i].dst, ops[i].op);
    else po_remove_edge(p->parts[k], ls, ops[i].dst);
  }
}
