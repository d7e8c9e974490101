"""Multi-GPU routing of update batches: one vertex-range partition per GPU (PPPCSR.cpp:13-34, 58-66).

Each rank holds a contiguous block of the global update stream.  `exchange_ops` buckets that block by
owner (stable, so every bucket keeps stream order), makes `src` partition-local (PPPCSR.cpp:46-52 passes
`src - distribution[p]` and the GLOBAL dest) and swaps buckets with ONE all-to-all (RCCL over xGMI when the
process group is "nccl"; gloo in the CPU tests).  all_to_all_single concatenates what it receives in
source-rank order; because rank r holds block r of the stream, that order IS global stream order, so each
partition sees exactly the subsequence the reference's PPPCSR would feed it (bit-exact per partition).
The volume is tiny (12 B per update), so the exchange is latency-bound: one collective per batch.
"""
import numpy as np
import torch
import torch.distributed as dist


def partition_layout(init_n, n_parts):
    """(starts, sizes) per PPPCSR.cpp:20-29: size = floor(n / P), the last partition takes the remainder"""
    ps = init_n // n_parts
    starts = np.arange(n_parts, dtype=np.int64) * ps
    sizes = np.full(n_parts, ps, np.int64)
    sizes[-1] = init_n - (n_parts - 1) * ps
    return starts, sizes


def owners_of(src_u32, init_n, n_parts):
    """PPPCSR::get_partiton (PPPCSR.cpp:58-66) for a tensor of vertex ids (int64, values in [0, 2^32))"""
    ps = init_n // n_parts
    if ps == 0:
        return torch.full_like(src_u32, n_parts - 1)
    return torch.clamp(src_u32 // ps, max=n_parts - 1)


def bucket_ops(ops, init_n, n_parts):
    """ops: (m,3) int32 tensor (bit patterns of uint32 src,dst,op) -> (bucketed ops with local src, counts[P])"""
    src = ops[:, 0].to(torch.int64) & 0xFFFFFFFF
    own = owners_of(src, init_n, n_parts)
    order = torch.sort(own, stable=True).indices
    b = ops.index_select(0, order).clone()
    ps = init_n // n_parts
    local = src.index_select(0, order) - own.index_select(0, order) * ps
    b[:, 0] = local.to(torch.int32)
    counts = torch.bincount(own, minlength=n_parts)
    return b.contiguous(), counts


_HIP = {"lib": None, "tried": False}


def _hip_lib():
    """the engine library, for the device-side bucketing kernel (None when it is not built: CPU tests)"""
    if not _HIP["tried"]:
        _HIP["tried"] = True
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libppcsr_hip.so")
        if os.path.exists(path):
            L = ctypes.CDLL(path)
            L.pppcsr_bucket_ops_device.argtypes = [ctypes.c_uint32, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
            _HIP["lib"] = L
    return _HIP["lib"]


def bucket_ops_device(ops, init_n, n_parts):
    """same contract as bucket_ops for an (m,3) int32 CUDA tensor, done by the engine's HIP counting-sort kernels
    (k_bucket_hist / k_bucket_scan / k_bucket_scatter) on torch's current stream"""
    L = _hip_lib()
    out = torch.empty_like(ops)
    counts = torch.zeros(n_parts, dtype=torch.int64, device=ops.device)
    rc = L.pppcsr_bucket_ops_device(init_n, n_parts, ops.data_ptr(), ops.shape[0], out.data_ptr(), counts.data_ptr(),
                                    torch.cuda.current_stream(ops.device).cuda_stream)
    if rc != 0:
        raise RuntimeError(f"pppcsr_bucket_ops_device failed with status {rc}")
    return out, counts


def _bucket(ops, init_n, n_parts):
    if ops.is_cuda and n_parts <= 64 and _hip_lib() is not None:
        return bucket_ops_device(ops.contiguous(), init_n, n_parts)
    return bucket_ops(ops, init_n, n_parts)


def exchange_parts(ops, init_n, n_parts, world, group=None, cap=None):
    """Strong-scaling form: `n_parts` partitions of the global layout over `world` ranks, rank r holding the contiguous
    partitions [r * ppr, (r + 1) * ppr), ppr = n_parts / world (the reference's partitions_per_domain).  ONE collective per
    batch and no host synchronisation before it: every rank sends every peer a fixed-capacity chunk
        [header rows: the ppr bucket sizes | payload: the peer's ppr buckets, back to back | padding]
    (capacity `cap`: the largest block any rank holds — the worst case; default: this rank's block size, which is right
    whenever all ranks hold equal blocks, as bench.py's do), so the split sizes are static.  Returns (per-partition tensors in
    stream order with partition-local src, counts) for this rank's partitions; the only host sync is the read of the
    world x ppr received header words that the per-partition apply calls need anyway."""
    assert n_parts % world == 0
    ppr = n_parts // world
    m = ops.shape[0]
    cap = m if cap is None else cap
    assert m <= cap
    hrows = (ppr + 2) // 3
    dev = ops.device
    b, counts = _bucket(ops, init_n, n_parts)
    counts = counts.to(torch.int64)
    if world == 1:
        cnt = counts.cpu().tolist()
        offs = np.concatenate([[0], np.cumsum(cnt)])
        return [b[int(offs[q]):int(offs[q + 1])] for q in range(n_parts)], cnt
    rows = cap + hrows
    send = torch.zeros((world, rows, 3), dtype=ops.dtype, device=dev)
    hdr = torch.zeros((world, hrows * 3), dtype=ops.dtype, device=dev)
    hdr[:, :ppr] = counts.view(world, ppr).to(ops.dtype)
    send[:, :hrows, :] = hdr.view(world, hrows, 3)
    if m:
        offs = torch.cumsum(counts, 0) - counts                                             # first bucketed row of each partition
        part_of = torch.repeat_interleave(torch.arange(n_parts, device=dev), counts, output_size=m)
        peer = part_of // ppr
        base = offs.index_select(0, peer * ppr)                                              # first row of the peer's buckets
        row = peer * rows + hrows + (torch.arange(m, device=dev) - base)
        send.view(-1, 3).index_copy_(0, row, b)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    got = recv[:, :hrows, :].reshape(world, hrows * 3)[:, :ppr].to(torch.int64).cpu().numpy()  # [source rank][local partition]
    out, cnts = [], []
    starts = np.concatenate([np.zeros((world, 1), np.int64), np.cumsum(got, 1)], 1)
    for q in range(ppr):
        segs = [recv[r, hrows + int(starts[r, q]): hrows + int(starts[r, q]) + int(got[r, q])] for r in range(world)]
        t = torch.cat(segs) if world > 1 else segs[0]
        out.append(t.contiguous())
        cnts.append(int(got[:, q].sum()))
    return out, cnts


def exchange_ops(ops, init_n, n_parts, group=None):
    """all-to-all of owner buckets; returns this rank's partition subsequence (stream order, local src)"""
    if ops.is_cuda and n_parts <= 64 and _hip_lib() is not None:
        b, counts = bucket_ops_device(ops.contiguous(), init_n, n_parts)
    else:
        b, counts = bucket_ops(ops, init_n, n_parts)
    recv_counts = torch.empty_like(counts)
    dist.all_to_all_single(recv_counts, counts, group=group)
    in_split = [int(x) for x in counts.tolist()]
    out_split = [int(x) for x in recv_counts.tolist()]
    out = torch.empty((sum(out_split), 3), dtype=ops.dtype, device=ops.device)
    dist.all_to_all_single(out, b, output_split_sizes=out_split, input_split_sizes=in_split, group=group)
    return out
