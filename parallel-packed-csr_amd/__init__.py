"""ppcsr_amd — thin ctypes binding of the MI355X packed-CSR engine (libppcsr_hip.so, C ABI in include/ppcsr.h).

The library is hand-written HIP for gfx950; there is NO CPU fallback: importing works anywhere, but creating
an engine without the built library or without a GPU raises.  Classes mirror the reference's
PCSR / PPPCSR method names (src/pcsr/PCSR.h:64-124, src/pppcsr/PPPCSR.h:11-60).

(The directory name contains a hyphen, so load this package by path — see tests/helpers.py:load_pkg.)
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PPCSR_LIB", os.path.join(_HERE, "csrc", "libppcsr_hip.so"))  # env override: debugging builds only

c_vp, c_u32, c_u64, c_int, c_i64, c_dbl = (ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int,
                                            ctypes.c_int64, ctypes.c_double)


class PpcsrError(RuntimeError):
    pass


class Stats(ctypes.Structure):
    _fields_ = [("N", c_u64), ("n", c_u64), ("logN", ctypes.c_int32), ("H", ctypes.c_int32),
                ("rounds", c_u64), ("committed", c_u64), ("planned", c_u64), ("exclusive_ops", c_u64),
                ("round_syncs", c_u64), ("redistribute_calls", c_u64), ("redistribute_slots", c_u64),
                ("double_calls", c_u64), ("half_calls", c_u64), ("big_redistributes", c_u64), ("rollbacks", c_u64),
                ("not_found", c_u64), ("duplicates", c_u64), ("noops", c_u64), ("slide_slots", c_u64),
                ("ops_applied", c_u64), ("last_batch_ms", c_dbl), ("last_batch_h2d_ms", c_dbl),
                ("prof_plan_ms", c_dbl), ("prof_check_ms", c_dbl), ("prof_apply_ms", c_dbl), ("prof_compact_ms", c_dbl), ("prof_launches", c_u64),
                ("wasted_rounds", c_u64), ("narrow_lost", c_u64), ("narrow", c_u64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


EXPORTED = [
    "ppcsr_create", "ppcsr_destroy", "ppcsr_add_edge", "ppcsr_remove_edge", "ppcsr_add_node", "ppcsr_apply_batch",
    "ppcsr_apply_batch_device", "ppcsr_edge_exists", "ppcsr_get_n", "ppcsr_get_node", "ppcsr_geometry",
    "ppcsr_get_neighbourhood", "ppcsr_read_neighbourhood", "ppcsr_scan_all", "ppcsr_bulk_build", "ppcsr_bfs", "ppcsr_pagerank", "ppcsr_export_state", "ppcsr_stats",
    "ppcsr_set_option", "ppcsr_snapshot", "ppcsr_restore", "ppcsr_check_invariants", "ppcsr_bench_scan_all", "ppcsr_bench_rebalance", "ppcsr_bench_resize", "ppcsr_strerror",
    "ppcsr_last_error", "ppcsr_device_count", "pppcsr_create", "pppcsr_destroy", "pppcsr_num_partitions",
    "pppcsr_get_partition", "pppcsr_partition_start", "pppcsr_partition", "pppcsr_add_edge", "pppcsr_remove_edge",
    "pppcsr_edge_exists", "pppcsr_get_neighbourhood", "pppcsr_get_node", "pppcsr_get_n", "pppcsr_add_node",
    "pppcsr_apply_batch", "pppcsr_bucket_ops", "pppcsr_bucket_ops_device",
    "pppcsr_create_local", "pppcsr_apply_batch_device", "pppcsr_apply_parts_device",
    "pppcsr_comm_unique_id", "pppcsr_comm_create", "pppcsr_comm_destroy", "pppcsr_exchange_apply",
    "pppcsr_xchg_create", "pppcsr_xchg_destroy", "pppcsr_xchg_pack", "pppcsr_xchg_layout", "pppcsr_xchg_apply",
    "pppcsr_repartition_export", "pppcsr_repartition", "pppcsr_balanced_starts", "pppcsr_set_num_neighbors_device",
    "pppcsr_xchg_set_num_neighbors", "pppcsr_exchange_set_num_neighbors", "pppcsr_bulk_build_device", "pppcsr_xchg_bulk_build",
    "pppcsr_exchange_bulk_build",
]

_LIBS = {}


def _share_hip_runtime():
    """A process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own libamdhip64 (SONAME libamdhip64.so.7)
    and look it up by its unversioned file name, so if the engine pulled /opt/rocm's copy in first, a later `import torch`
    would load a second runtime that finds no GPU (and the reverse order would starve the engine).  Pre-loading torch's
    copy — without importing torch — makes both resolve to the same object.  Without torch the engine uses /opt/rocm's."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load_library(path=None):
    """Load the engine library.  Default: the in-tree HIP build; raises if it has not been built."""
    path = path or LIB_PATH
    if path in _LIBS:
        return _LIBS[path]
    if not os.path.exists(path):
        raise PpcsrError(f"{path} not found: build it with `python __graft_entry__.py` "
                         "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    _share_hip_runtime()
    L = ctypes.CDLL(path)
    L.ppcsr_create.argtypes = [c_u32, c_u32, c_int, c_int, ctypes.POINTER(c_vp)]
    L.ppcsr_destroy.argtypes = [c_vp]
    L.ppcsr_add_edge.argtypes = [c_vp, c_u32, c_u32, c_u32]
    L.ppcsr_remove_edge.argtypes = [c_vp, c_u32, c_u32]
    L.ppcsr_add_node.argtypes = [c_vp]
    L.ppcsr_apply_batch.argtypes = [c_vp, c_vp, c_u64]
    L.ppcsr_apply_batch_device.argtypes = [c_vp, c_vp, c_u64]
    L.ppcsr_edge_exists.argtypes = [c_vp, c_u32, c_u32, ctypes.POINTER(c_int)]
    L.ppcsr_get_n.argtypes = [c_vp, ctypes.POINTER(c_u64)]
    L.ppcsr_get_node.argtypes = [c_vp, c_u32, c_vp]
    L.ppcsr_geometry.argtypes = [c_vp, ctypes.POINTER(c_u64), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
    L.ppcsr_get_neighbourhood.argtypes = [c_vp, c_int, c_vp, c_u64, ctypes.POINTER(c_u64)]
    L.ppcsr_read_neighbourhood.argtypes = [c_vp, c_int]
    L.ppcsr_scan_all.argtypes = [c_vp, c_vp, c_vp, c_u64, ctypes.POINTER(c_u64)]
    L.ppcsr_bulk_build.argtypes = [c_vp, c_vp, c_u64, ctypes.POINTER(ctypes.c_double)]
    L.ppcsr_bfs.argtypes = [c_vp, c_u32, c_vp, ctypes.POINTER(ctypes.c_double)]
    L.ppcsr_pagerank.argtypes = [c_vp, c_vp, c_vp, ctypes.POINTER(ctypes.c_double)]
    L.ppcsr_export_state.argtypes = [c_vp, c_vp, c_vp]
    L.ppcsr_stats.argtypes = [c_vp, ctypes.POINTER(Stats)]
    L.ppcsr_set_option.argtypes = [c_vp, ctypes.c_char_p, c_i64]
    L.ppcsr_snapshot.argtypes = [c_vp]
    L.ppcsr_restore.argtypes = [c_vp]
    L.ppcsr_check_invariants.argtypes = [c_vp, ctypes.POINTER(c_u64)]
    L.ppcsr_bench_scan_all.argtypes = [c_vp, ctypes.POINTER(c_dbl), ctypes.POINTER(c_u64)]
    L.ppcsr_bench_rebalance.argtypes = [c_vp, c_u64, c_int, ctypes.POINTER(c_dbl)]
    L.ppcsr_bench_resize.argtypes = [c_vp, c_int, ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl)]
    L.ppcsr_strerror.restype = ctypes.c_char_p
    L.ppcsr_strerror.argtypes = [c_int]
    L.ppcsr_last_error.restype = ctypes.c_char_p
    L.pppcsr_create.argtypes = [c_u32, c_u32, c_int, c_int, c_int, c_vp, c_int, ctypes.POINTER(c_vp)]
    L.pppcsr_destroy.argtypes = [c_vp]
    L.pppcsr_num_partitions.argtypes = [c_vp, ctypes.POINTER(c_u64)]
    L.pppcsr_get_partition.argtypes = [c_vp, c_u64, ctypes.POINTER(c_u64)]
    L.pppcsr_partition_start.argtypes = [c_vp, c_u64, ctypes.POINTER(c_u64)]
    L.pppcsr_partition.argtypes = [c_vp, c_u64, ctypes.POINTER(c_vp)]
    L.pppcsr_add_edge.argtypes = [c_vp, c_u32, c_u32, c_u32]
    L.pppcsr_remove_edge.argtypes = [c_vp, c_u32, c_u32]
    L.pppcsr_edge_exists.argtypes = [c_vp, c_u32, c_u32, ctypes.POINTER(c_int)]
    L.pppcsr_get_neighbourhood.argtypes = [c_vp, c_int, c_vp, c_u64, ctypes.POINTER(c_u64)]
    L.pppcsr_get_node.argtypes = [c_vp, c_u32, c_vp]
    L.pppcsr_get_n.argtypes = [c_vp, ctypes.POINTER(c_u64)]
    L.pppcsr_add_node.argtypes = [c_vp]
    L.pppcsr_apply_batch.argtypes = [c_vp, c_vp, c_u64]
    L.pppcsr_bucket_ops.argtypes = [c_u32, c_u64, c_vp, c_u64, c_vp, c_vp]
    L.pppcsr_bucket_ops_device.argtypes = [c_u32, c_u64, c_vp, c_u64, c_vp, c_vp, c_vp]
    L.pppcsr_create_local.argtypes = [c_u32, c_int, c_int, c_int, c_u64, c_u64, c_int, ctypes.POINTER(c_vp)]
    L.pppcsr_apply_batch_device.argtypes = [c_vp, c_vp, c_u64]
    L.pppcsr_apply_parts_device.argtypes = [c_vp, c_u64, c_u64, c_vp, c_vp]
    L.pppcsr_comm_unique_id.argtypes = [c_vp]
    L.pppcsr_comm_create.argtypes = [c_vp, c_int, c_int, c_int, ctypes.POINTER(c_vp)]
    L.pppcsr_comm_destroy.argtypes = [c_vp]
    L.pppcsr_exchange_apply.argtypes = [c_vp, c_vp, c_vp, c_u64]
    L.pppcsr_xchg_create.argtypes = [c_vp, c_int, c_int, ctypes.POINTER(c_vp)]
    L.pppcsr_xchg_destroy.argtypes = [c_vp]
    L.pppcsr_xchg_pack.argtypes = [c_vp, c_vp, c_u64, c_vp, ctypes.POINTER(c_vp)]
    L.pppcsr_xchg_layout.argtypes = [c_vp, c_vp, c_vp]
    L.pppcsr_xchg_apply.argtypes = [c_vp]
    L.pppcsr_repartition_export.argtypes = [c_vp, c_vp, ctypes.POINTER(c_vp), ctypes.POINTER(c_u64), ctypes.POINTER(c_vp), ctypes.POINTER(c_u64)]
    L.pppcsr_set_num_neighbors_device.argtypes = [c_vp, c_vp, c_u64]
    L.pppcsr_xchg_set_num_neighbors.argtypes = [c_vp]
    L.pppcsr_xchg_bulk_build.argtypes = [c_vp]
    L.pppcsr_bulk_build_device.argtypes = [c_vp, c_vp, c_u64]
    L.pppcsr_exchange_bulk_build.argtypes = [c_vp, c_vp, c_vp, c_u64]
    L.pppcsr_exchange_set_num_neighbors.argtypes = [c_vp, c_vp, c_vp, c_u64]
    L.pppcsr_repartition.argtypes = [c_vp, c_vp]
    L.pppcsr_balanced_starts.argtypes = [c_vp, c_vp]
    _LIBS[path] = L
    return L


def _ops(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    assert a.ndim == 2 and a.shape[1] == 3, "ops must be (n,3) uint32 rows of (src, dst, op)"
    return a


class PCSR:
    """Mirror of the reference class PCSR (PCSR.h:64-124) on one GPU."""

    def __init__(self, init_n, src_n=None, lock_search=True, device=0, lib=None, _handle=None):
        self.L = lib or load_library()
        self._own = _handle is None
        if _handle is not None:
            self.h = c_vp(_handle)
        else:
            self.h = c_vp()
            self._chk(self.L.ppcsr_create(init_n, init_n if src_n is None else src_n, int(lock_search), device,
                                          ctypes.byref(self.h)))

    def _chk(self, rc):
        if rc != 0:
            raise PpcsrError(f"ppcsr status {rc} ({self.L.ppcsr_strerror(rc).decode()}): "
                             f"{self.L.ppcsr_last_error().decode()}")

    def close(self):
        if getattr(self, "h", None) and self._own and self.h.value:
            self.L.ppcsr_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # reference API names
    def add_edge(self, src, dest, value=1): self._chk(self.L.ppcsr_add_edge(self.h, src, dest, value))
    def remove_edge(self, src, dest): self._chk(self.L.ppcsr_remove_edge(self.h, src, dest))
    def add_node(self): self._chk(self.L.ppcsr_add_node(self.h))

    def edge_exists(self, src, dest):
        out = c_int()
        self._chk(self.L.ppcsr_edge_exists(self.h, src, dest, ctypes.byref(out)))
        return bool(out.value)

    def get_n(self):
        n = c_u64()
        self._chk(self.L.ppcsr_get_n(self.h, ctypes.byref(n)))
        return n.value

    def getNode(self, v):
        out = np.zeros(3, np.uint32)
        self._chk(self.L.ppcsr_get_node(self.h, v, out.ctypes.data))
        return tuple(int(x) for x in out)

    def get_neighbourhood(self, src):
        cnt = c_u64()
        self._chk(self.L.ppcsr_get_neighbourhood(self.h, src, None, 0, ctypes.byref(cnt)))
        out = np.empty(cnt.value, np.int32)
        if cnt.value:
            self._chk(self.L.ppcsr_get_neighbourhood(self.h, src, out.ctypes.data, cnt.value, ctypes.byref(cnt)))
        return out

    def read_neighbourhood(self, src): self._chk(self.L.ppcsr_read_neighbourhood(self.h, src))

    # batch + state
    def apply(self, ops):
        a = _ops(ops)
        if len(a):
            self._chk(self.L.ppcsr_apply_batch(self.h, a.ctypes.data, len(a)))

    def apply_device(self, dev_ptr, n):
        self._chk(self.L.ppcsr_apply_batch_device(self.h, dev_ptr, n))

    def bulk_build(self, ops, with_ms=False):
        """NON-parity fast path: build an empty graph from a list of adds in a few device passes"""
        a = _ops(ops)
        ms = ctypes.c_double(0.0)
        self._chk(self.L.ppcsr_bulk_build(self.h, a.ctypes.data, len(a), ctypes.byref(ms)))
        return ms.value if with_ms else None

    # consumers on the device (reference: src/utility/bfs.h, src/utility/pagerank.h)
    def bfs(self, start, with_ms=False):
        out = np.empty(self.get_n(), np.uint32)
        ms = ctypes.c_double(0.0)
        self._chk(self.L.ppcsr_bfs(self.h, start, out.ctypes.data, ctypes.byref(ms)))
        return (out, ms.value) if with_ms else out

    def pagerank(self, node_values, with_ms=False):
        vals = np.ascontiguousarray(node_values, np.float32)
        assert len(vals) == self.get_n()
        out = np.empty(len(vals), np.float32)
        ms = ctypes.c_double(0.0)
        self._chk(self.L.ppcsr_pagerank(self.h, vals.ctypes.data, out.ctypes.data, ctypes.byref(ms)))
        return (out, ms.value) if with_ms else out

    def geometry(self):
        N, lg, H = c_u64(), c_int(), c_int()
        self._chk(self.L.ppcsr_geometry(self.h, ctypes.byref(N), ctypes.byref(lg), ctypes.byref(H)))
        return N.value, lg.value, H.value

    def state(self):
        N, _, _ = self.geometry()
        n = self.get_n()
        items = np.empty((N, 3), np.uint32)
        nodes = np.empty((n, 3), np.uint32)
        self._chk(self.L.ppcsr_export_state(self.h, items.ctypes.data, nodes.ctypes.data if n else None))
        return items, nodes

    def scan_all(self):
        tot = c_u64()
        n = self.get_n()
        rows = np.zeros(n + 1, np.uint64)
        rc = self.L.ppcsr_scan_all(self.h, rows.ctypes.data, None, 0, ctypes.byref(tot))
        if rc not in (0, 6):
            self._chk(rc)
        dests = np.empty(tot.value, np.int32)
        self._chk(self.L.ppcsr_scan_all(self.h, rows.ctypes.data, dests.ctypes.data, tot.value, ctypes.byref(tot)))
        return rows, dests

    def stats(self):
        s = Stats()
        self._chk(self.L.ppcsr_stats(self.h, ctypes.byref(s)))
        return s.as_dict()

    def set_option(self, key, value): self._chk(self.L.ppcsr_set_option(self.h, key.encode(), int(value)))

    def snapshot(self): self._chk(self.L.ppcsr_snapshot(self.h))
    def restore(self): self._chk(self.L.ppcsr_restore(self.h))

    def check_invariants(self):
        bad = c_u64()
        self._chk(self.L.ppcsr_check_invariants(self.h, ctypes.byref(bad)))
        return bad.value

    def bench_scan_all(self):
        ms, tot = c_dbl(), c_u64()
        self._chk(self.L.ppcsr_bench_scan_all(self.h, ctypes.byref(ms), ctypes.byref(tot)))
        return ms.value, tot.value

    def bench_resize(self, iters=3):
        """(double_list ms, half_list ms): device time per call, the array doubled and halved back `iters` times"""
        d, h = c_dbl(), c_dbl()
        self._chk(self.L.ppcsr_bench_resize(self.h, iters, ctypes.byref(d), ctypes.byref(h)))
        return d.value, h.value

    def bench_rebalance(self, window_slots, iters=5):
        ms = c_dbl()
        self._chk(self.L.ppcsr_bench_rebalance(self.h, window_slots, iters, ctypes.byref(ms)))
        return ms.value


class PPPCSR:
    """Mirror of the reference class PPPCSR (PPPCSR.h:11-60): vertex-range partitions, one per GPU."""

    def __init__(self, init_n, src_n=None, lock_search=True, numDomain=1, partitionsPerDomain=1, use_numa=False,
                 devices=None, lib=None, local=None):
        """local=(first_part, n_parts, device): create only that range of the layout's partitions (multi-process runs:
        one rank per GPU holds the partitions of its domain)"""
        self.L = lib or load_library()
        self.h = c_vp()
        if local is not None:
            rc = self.L.pppcsr_create_local(init_n, int(lock_search), numDomain, partitionsPerDomain, local[0], local[1],
                                            local[2], ctypes.byref(self.h))
        else:
            devs = np.array(devices if devices else [0], np.int32)
            rc = self.L.pppcsr_create(init_n, init_n if src_n is None else src_n, int(lock_search), numDomain,
                                      partitionsPerDomain, devs.ctypes.data, len(devs), ctypes.byref(self.h))
        if rc != 0:
            raise PpcsrError(f"pppcsr_create: status {rc}: {self.L.ppcsr_last_error().decode()}")

    def _chk(self, rc):
        if rc != 0:
            raise PpcsrError(f"ppcsr status {rc}: {self.L.ppcsr_last_error().decode()}")

    def close(self):
        if getattr(self, "comm", None) is not None and self.comm.value:
            self.L.pppcsr_comm_destroy(self.comm)
            self.comm = None
        if getattr(self, "xchg", None) is not None and self.xchg.value:
            self.L.pppcsr_xchg_destroy(self.xchg)
            self.xchg = None
        if getattr(self, "h", None) and self.h.value:
            self.L.pppcsr_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_partitions(self):
        out = c_u64()
        self._chk(self.L.pppcsr_num_partitions(self.h, ctypes.byref(out)))
        return out.value

    def get_partiton(self, v):  # sic: the reference's spelling (PPPCSR.h:27)
        out = c_u64()
        self._chk(self.L.pppcsr_get_partition(self.h, v, ctypes.byref(out)))
        return out.value

    def partition_start(self, k):
        out = c_u64()
        self._chk(self.L.pppcsr_partition_start(self.h, k, ctypes.byref(out)))
        return out.value

    def partition(self, k):
        out = c_vp()
        self._chk(self.L.pppcsr_partition(self.h, k, ctypes.byref(out)))
        return PCSR(0, lib=self.L, _handle=out.value)

    def add_edge(self, s, d, v=1): self._chk(self.L.pppcsr_add_edge(self.h, s, d, v))
    def remove_edge(self, s, d): self._chk(self.L.pppcsr_remove_edge(self.h, s, d))
    def add_node(self): self._chk(self.L.pppcsr_add_node(self.h))

    def edge_exists(self, s, d):
        out = c_int()
        self._chk(self.L.pppcsr_edge_exists(self.h, s, d, ctypes.byref(out)))
        return bool(out.value)

    def get_n(self):
        out = c_u64()
        self._chk(self.L.pppcsr_get_n(self.h, ctypes.byref(out)))
        return out.value

    def getNode(self, v):
        out = np.zeros(3, np.uint32)
        self._chk(self.L.pppcsr_get_node(self.h, v, out.ctypes.data))
        return tuple(int(x) for x in out)

    def get_neighbourhood(self, src):
        cnt = c_u64()
        self._chk(self.L.pppcsr_get_neighbourhood(self.h, src, None, 0, ctypes.byref(cnt)))
        out = np.empty(cnt.value, np.int32)
        if cnt.value:
            self._chk(self.L.pppcsr_get_neighbourhood(self.h, src, out.ctypes.data, cnt.value, ctypes.byref(cnt)))
        return out

    def apply(self, ops):
        a = _ops(ops)
        if len(a):
            self._chk(self.L.pppcsr_apply_batch(self.h, a.ctypes.data, len(a)))

    def apply_device(self, dev_ptr, n):
        """global stream resident in HBM (all partitions on that GPU): device bucketing + concurrent per-partition apply"""
        self._chk(self.L.pppcsr_apply_batch_device(self.h, dev_ptr, n))

    # native exchange (RCCL send/recv from the engine library; no torch in the data path)
    @staticmethod
    def comm_unique_id(lib=None):
        L = lib or load_library()
        buf = ctypes.create_string_buffer(128)
        rc = L.pppcsr_comm_unique_id(buf)
        if rc != 0:
            raise PpcsrError(f"pppcsr_comm_unique_id: status {rc}: {L.ppcsr_last_error().decode()}")
        return bytes(buf.raw)

    def comm_create(self, unique_id, n_ranks, rank, device):
        self.comm = c_vp()
        self._chk(self.L.pppcsr_comm_create(ctypes.c_char_p(unique_id), n_ranks, rank, device, ctypes.byref(self.comm)))

    def exchange_apply(self, dev_ptr, n):
        """this rank's block of the global stream (device pointer, n updates): route through RCCL + apply the local partitions"""
        self._chk(self.L.pppcsr_exchange_apply(self.h, self.comm, dev_ptr, n))

    # the same exchange with the transport left to the caller (pack -> [carrier of your choice] -> layout -> apply)
    def xchg_create(self, n_ranks, rank):
        self.xchg = c_vp()
        self._chk(self.L.pppcsr_xchg_create(self.h, n_ranks, rank, ctypes.byref(self.xchg)))
        self._xchg_shape = (n_ranks, self.num_partitions() // n_ranks)

    def xchg_pack(self, dev_ptr, n):
        """-> (send_counts[P], device address of the bucketed block); partition p's rows start at sum(send_counts[:p])"""
        counts = np.zeros(self.num_partitions(), np.uint64)
        d_send = c_vp()
        self._chk(self.L.pppcsr_xchg_pack(self.xchg, dev_ptr, n, counts.ctypes.data, ctypes.byref(d_send)))
        return counts, d_send.value or 0

    def xchg_layout(self, recv_counts):
        """recv_counts[r * ppr + q] -> device addresses the segments (source r, local partition q) must be written to"""
        rc = np.ascontiguousarray(recv_counts, np.uint64)
        assert rc.size == self._xchg_shape[0] * self._xchg_shape[1]
        dst = (c_vp * rc.size)()
        self._chk(self.L.pppcsr_xchg_layout(self.xchg, rc.ctypes.data, dst))
        return [int(x or 0) for x in dst]

    def xchg_apply(self):
        self._chk(self.L.pppcsr_xchg_apply(self.xchg))

    def xchg_set_num_neighbors(self):
        self._chk(self.L.pppcsr_xchg_set_num_neighbors(self.xchg))

    def xchg_bulk_build(self):
        self._chk(self.L.pppcsr_xchg_bulk_build(self.xchg))

    def bulk_build_device(self, dev_ptr, n):
        self._chk(self.L.pppcsr_bulk_build_device(self.h, dev_ptr, n))

    def exchange_bulk_build(self, dev_ptr, n):
        self._chk(self.L.pppcsr_exchange_bulk_build(self.h, self.comm, dev_ptr, n))

    def exchange_set_num_neighbors(self, dev_ptr, n):
        self._chk(self.L.pppcsr_exchange_set_num_neighbors(self.h, self.comm, dev_ptr, n))

    def set_num_neighbors_device(self, dev_ptr, n):
        self._chk(self.L.pppcsr_set_num_neighbors_device(self.h, dev_ptr, n))

    # repartitioning (include/ppcsr.h; no reference equivalent: PCSR.h:91-112 is a sketch)
    def repartition(self, new_starts):
        st = np.ascontiguousarray(new_starts, np.uint64)
        assert st.size == self.num_partitions()
        self._chk(self.L.pppcsr_repartition(self.h, st.ctypes.data))

    def repartition_export(self, new_starts):
        """-> ((device address, n) of the edges on the move as adds of the global stream, (device address, n) of the
        num_neighbors records); route the first into the bulk build of the recreated partitions (bulk_build_device /
        exchange_bulk_build / xchg_bulk_build), then the second through the set_num_neighbors calls"""
        st = np.ascontiguousarray(new_starts, np.uint64)
        assert st.size == self.num_partitions()
        d, n, dn, nn = c_vp(), c_u64(), c_vp(), c_u64()
        self._chk(self.L.pppcsr_repartition_export(self.h, st.ctypes.data, ctypes.byref(d), ctypes.byref(n), ctypes.byref(dn), ctypes.byref(nn)))
        return (d.value or 0, n.value), (dn.value or 0, nn.value)

    def balanced_starts(self):
        st = np.zeros(self.num_partitions(), np.uint64)
        self._chk(self.L.pppcsr_balanced_starts(self.h, st.ctypes.data))
        return st

    def apply_parts_device(self, first_part, dev_ptrs, counts):
        """already routed device-resident subsequences, one per partition of [first_part, first_part + len(counts))"""
        k = len(counts)
        ptrs = (c_vp * k)(*[int(x) for x in dev_ptrs])
        cnts = (c_u64 * k)(*[int(x) for x in counts])
        self._chk(self.L.pppcsr_apply_parts_device(self.h, first_part, k, ptrs, cnts))


def bucket_ops(init_n, n_parts, ops, lib=None):
    """stable owner bucketing (PPPCSR.cpp:46-66 routing rule); returns (bucketed ops with local src, counts)"""
    L = lib or load_library()
    a = _ops(ops)
    out = np.empty_like(a)
    counts = np.zeros(n_parts, np.uint64)
    rc = L.pppcsr_bucket_ops(init_n, n_parts, a.ctypes.data, len(a), out.ctypes.data, counts.ctypes.data)
    if rc != 0:
        raise PpcsrError(f"pppcsr_bucket_ops: status {rc}")
    return out, counts
