"""ctypes bindings for the CPU checkers (TEST INFRASTRUCTURE ONLY).

* ``Oracle``  — oracle/libppcsr_oracle.so, our plain-C restatement of the reference algorithm.
* ``RefPCSR`` — oracle/_ref/libref_pcsr.so, the unmodified reference compiled where it lies
  (present only where /root/reference was available at build time, or prebuilt).

Nothing in the product package imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libppcsr_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_pcsr.so")

c_vp, c_u32, c_u64, c_int = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int


def build_oracle():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True, stdout=subprocess.DEVNULL)


def _load_oracle():
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(
        os.path.join(ORACLE_DIR, "ppcsr_oracle.c")
    ):
        build_oracle()
    L = ctypes.CDLL(ORACLE_SO)
    L.po_create.restype = c_vp
    L.po_create.argtypes = [c_u32, c_u32, c_int]
    L.po_destroy.argtypes = [c_vp]
    L.po_clone.restype = c_vp
    L.po_clone.argtypes = [c_vp]
    L.po_add_edge.argtypes = [c_vp, c_u32, c_u32, c_u32]
    L.po_remove_edge.argtypes = [c_vp, c_u32, c_u32]
    L.po_add_node.argtypes = [c_vp]
    L.po_edge_exists.argtypes = [c_vp, c_u32, c_u32]
    L.po_apply.argtypes = [c_vp, c_vp, c_u64]
    L.po_get_n.restype = c_u64
    L.po_get_n.argtypes = [c_vp]
    L.po_geometry.argtypes = [c_vp, c_vp, c_vp, c_vp]
    L.po_export.argtypes = [c_vp, c_vp, c_vp]
    L.po_get_neighbourhood.restype = c_u64
    L.po_get_neighbourhood.argtypes = [c_vp, c_int, c_vp, c_u64]
    L.po_get_stats.argtypes = [c_vp, c_vp]
    L.po_reset_stats.argtypes = [c_vp]
    L.po_debug_redistribute.argtypes = [c_vp, c_u64, c_u64]
    L.po_set_num_neighbors.argtypes = [c_vp, c_u32, c_u32]
    L.po_import_state.restype = c_vp
    L.po_import_state.argtypes = [c_u64, c_vp, c_u32, c_vp, c_int]
    L.po_redistribute_positions.argtypes = [c_u64, c_u64, c_u64, c_vp]
    L.pop_create.restype = c_vp
    L.pop_create.argtypes = [c_u32, c_u32, c_int, c_int, c_int]
    L.pop_destroy.argtypes = [c_vp]
    L.pop_num_partitions.restype = c_u64
    L.pop_num_partitions.argtypes = [c_vp]
    L.pop_get_partition.restype = c_u64
    L.pop_get_partition.argtypes = [c_vp, c_u64]
    L.pop_partition_start.restype = c_u64
    L.pop_partition_start.argtypes = [c_vp, c_u64]
    L.pop_partition.restype = c_vp
    L.pop_partition.argtypes = [c_vp, c_u64]
    L.pop_apply.argtypes = [c_vp, c_vp, c_u64]
    return L


_ORACLE = None


def oracle_lib():
    global _ORACLE
    if _ORACLE is None:
        _ORACLE = _load_oracle()
    return _ORACLE


STAT_FIELDS = [
    "redistribute_calls", "redistribute_slots", "double_calls", "half_calls", "slide_right_calls",
    "slide_steps", "slide_left_calls", "global_path", "duplicates", "not_found", "search_probes",
    "ops_add", "ops_del",
]


def as_ops(ops):
    """(n,3) uint32 C-contiguous array of (src, dst, op)."""
    a = np.ascontiguousarray(ops, dtype=np.uint32)
    assert a.ndim == 2 and a.shape[1] == 3
    return a


class _State:
    def geometry(self):
        N, lg, H = c_u64(), c_int(), c_int()
        self._geometry(ctypes.byref(N), ctypes.byref(lg), ctypes.byref(H))
        return N.value, lg.value, H.value

    def state(self):
        N, _, _ = self.geometry()
        n = self.get_n()
        items = np.empty((N, 3), np.uint32)
        nodes = np.empty((n, 3), np.uint32)
        self._export(items.ctypes.data, nodes.ctypes.data if n else None)
        return items, nodes


class Oracle(_State):
    """Our C restatement (one PCSR).  `handle` may wrap a partition owned by an OraclePPPCSR."""

    def __init__(self, init_n, src_n=None, lock_search=True, handle=None):
        self.L = oracle_lib()
        self.own = handle is None
        self.h = handle if handle is not None else self.L.po_create(init_n, init_n if src_n is None else src_n, int(lock_search))

    @classmethod
    def from_state(cls, items, nodes, lock_search=True):
        """an oracle that starts from a raw exported state (edges[N], nodes[n]): what follows a non-parity step"""
        items = np.ascontiguousarray(items, np.uint32)
        nodes = np.ascontiguousarray(nodes, np.uint32)
        L = oracle_lib()
        h = L.po_import_state(len(items), items.ctypes.data, len(nodes), nodes.ctypes.data if len(nodes) else None, int(lock_search))
        o = cls(0, handle=h)
        o.own = True
        return o

    def close(self):
        if self.own and self.h:
            self.L.po_destroy(self.h)
        self.h = None

    def clone(self):
        """independent copy of the whole state (several batches can be replayed from one loaded core graph)"""
        o = Oracle(0, handle=self.L.po_clone(self.h))
        o.own = True
        return o

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_edge(self, s, d, v=1): self.L.po_add_edge(self.h, s, d, v)
    def remove_edge(self, s, d): self.L.po_remove_edge(self.h, s, d)
    def add_node(self): self.L.po_add_node(self.h)
    def edge_exists(self, s, d): return bool(self.L.po_edge_exists(self.h, s, d))
    def get_n(self): return self.L.po_get_n(self.h)
    def apply(self, ops):
        a = as_ops(ops)
        self.L.po_apply(self.h, a.ctypes.data, len(a))
    def _geometry(self, a, b, c): self.L.po_geometry(self.h, a, b, c)
    def _export(self, a, b): self.L.po_export(self.h, a, b)
    def get_neighbourhood(self, v):
        k = self.L.po_get_neighbourhood(self.h, v, None, 0)
        out = np.empty(k, np.int32)
        self.L.po_get_neighbourhood(self.h, v, out.ctypes.data, k)
        return out
    def stats(self):
        buf = (c_u64 * len(STAT_FIELDS))()
        self.L.po_get_stats(self.h, buf)
        return dict(zip(STAT_FIELDS, list(buf)))
    def reset_stats(self): self.L.po_reset_stats(self.h)
    def debug_redistribute(self, index, length): self.L.po_debug_redistribute(self.h, index, length)
    def set_num_neighbors(self, v, nn): self.L.po_set_num_neighbors(self.h, int(v), int(nn))


class OraclePPPCSR:
    def __init__(self, init_n, lock_search=True, num_domains=1, parts_per_domain=1):
        self.L = oracle_lib()
        self.h = self.L.pop_create(init_n, init_n, int(lock_search), num_domains, parts_per_domain)
    def close(self):
        if self.h:
            self.L.pop_destroy(self.h)
        self.h = None
    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
    def num_partitions(self): return self.L.pop_num_partitions(self.h)
    def get_partition(self, v): return self.L.pop_get_partition(self.h, v)
    def partition_start(self, k): return self.L.pop_partition_start(self.h, k)
    def partition(self, k): return Oracle(0, handle=self.L.pop_partition(self.h, k))
    def apply(self, ops):
        a = as_ops(ops)
        self.L.pop_apply(self.h, a.ctypes.data, len(a))


def have_ref():
    return os.path.exists(REF_SO)


_REF = None


def ref_lib():
    global _REF
    if _REF is None:
        L = ctypes.CDLL(REF_SO)
        L.ref_create.restype = c_vp
        L.ref_create.argtypes = [c_u32, c_u32, c_int]
        L.ref_destroy.argtypes = [c_vp]
        L.ref_add_edge.argtypes = [c_vp, c_u32, c_u32, c_u32]
        L.ref_remove_edge.argtypes = [c_vp, c_u32, c_u32]
        L.ref_add_node.argtypes = [c_vp]
        L.ref_edge_exists.argtypes = [c_vp, c_u32, c_u32]
        L.ref_apply.argtypes = [c_vp, c_vp, c_u64]
        L.ref_get_n.restype = c_u64
        L.ref_get_n.argtypes = [c_vp]
        L.ref_geometry.argtypes = [c_vp, c_vp, c_vp, c_vp]
        L.ref_export.argtypes = [c_vp, c_vp, c_vp]
        L.ref_get_neighbourhood.restype = c_u64
        L.ref_get_neighbourhood.argtypes = [c_vp, c_int, c_vp, c_u64]
        if hasattr(L, "ref_bfs"):
            L.ref_bfs.argtypes = [c_vp, c_u32, c_vp]
            L.ref_pagerank.argtypes = [c_vp, c_vp, c_vp]
        L.refp_create.restype = c_vp
        L.refp_create.argtypes = [c_u32, c_u32, c_int, c_int, c_int]
        L.refp_destroy.argtypes = [c_vp]
        L.refp_apply.argtypes = [c_vp, c_vp, c_u64]
        L.refp_get_partition.restype = c_u64
        L.refp_get_partition.argtypes = [c_vp, c_u64]
        L.refp_get_n.restype = c_u64
        L.refp_get_n.argtypes = [c_vp]
        L.refp_edge_exists.argtypes = [c_vp, c_u32, c_u32]
        L.refp_get_node.argtypes = [c_vp, c_int, c_vp]
        L.refp_get_neighbourhood.restype = c_u64
        L.refp_get_neighbourhood.argtypes = [c_vp, c_int, c_vp, c_u64]
        _REF = L
    return _REF


class RefPCSR(_State):
    """The real reference PCSR, sequentially driven."""

    def __init__(self, init_n, src_n=None, lock_search=True):
        self.L = ref_lib()
        self.h = self.L.ref_create(init_n, init_n if src_n is None else src_n, int(lock_search))
    def close(self):
        if self.h:
            self.L.ref_destroy(self.h)
        self.h = None
    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
    def add_edge(self, s, d, v=1): self.L.ref_add_edge(self.h, s, d, v)
    def remove_edge(self, s, d): self.L.ref_remove_edge(self.h, s, d)
    def add_node(self): self.L.ref_add_node(self.h)
    def edge_exists(self, s, d): return bool(self.L.ref_edge_exists(self.h, s, d))
    def get_n(self): return self.L.ref_get_n(self.h)

    def bfs(self, start):
        """the reference's own template bfs(graph, start) (src/utility/bfs.h:15-36) on the reference PCSR"""
        out = np.empty(self.get_n(), np.uint32)
        self.L.ref_bfs(self.h, start, out.ctypes.data)
        return out

    def pagerank(self, node_values):
        """the reference's own template pagerank(graph, node_values) (src/utility/pagerank.h:15-29), weight_t = float"""
        vals = np.ascontiguousarray(node_values, np.float32)
        assert len(vals) == self.get_n()
        out = np.empty(len(vals), np.float32)
        self.L.ref_pagerank(self.h, vals.ctypes.data, out.ctypes.data)
        return out
    def apply(self, ops):
        a = as_ops(ops)
        self.L.ref_apply(self.h, a.ctypes.data, len(a))
    def _geometry(self, a, b, c): self.L.ref_geometry(self.h, a, b, c)
    def _export(self, a, b): self.L.ref_export(self.h, a, b)
    def get_neighbourhood(self, v):
        k = self.L.ref_get_neighbourhood(self.h, v, None, 0)
        out = np.empty(k, np.int32)
        self.L.ref_get_neighbourhood(self.h, v, out.ctypes.data, k)
        return out
