"""Tile sharding across GPUs: tiles are independent, so tile i of a batch goes to shard i mod G and
nothing but a tile count crosses xGMI.  The work is done by the C ABI (osmt_batch_shard_*, osmt_render_batch_multi,
osmt_comm_*, osmt_allreduce_tile_count*); this module is its Python face for bench.py and the tests."""
import ctypes as C

import numpy as np

from . import abi
from .display_list import JOB_DTYPE, OP_DTYPE, RING_DTYPE, DisplayList
from .lib import check, load


def shard_indices(n_tiles, rank, world):
    """Indices of the global batch rendered by `rank`: i with i mod world == rank (http_server.rs:105-108)."""
    return np.arange(rank, n_tiles, world)


def shard_display_list(dl: DisplayList, rank, world) -> DisplayList:
    """osmt_batch_shard_create: the display list of one shard, pools re-packed by the library (host only, no GPU)."""
    L = load()
    b = dl.as_batch()
    h = C.c_void_p()
    check(L.osmt_batch_shard_create(C.byref(b), rank, world, C.byref(h)))
    try:
        sb = L.osmt_batch_shard_get(h).contents

        def arr(ptr, n, dtype):
            if n == 0 or not ptr:
                return np.zeros(0, dtype)
            size = n * np.dtype(dtype).itemsize
            return np.frombuffer((C.c_uint8 * size).from_address(C.addressof(ptr.contents)), dtype=dtype, count=n).copy()

        jobs = arr(sb.jobs, sb.n_jobs, JOB_DTYPE)
        ops = arr(sb.ops, sb.n_ops, OP_DTYPE)
        rings = arr(sb.rings, sb.n_rings, RING_DTYPE)
        dashes = arr(sb.dashes, sb.n_dashes, np.float64)
        if sb.coord_kind == abi.COORD_LATLON_F64:
            coords = arr(sb.latlon, 2 * sb.n_pts, np.float64).reshape(-1, 2)
        elif sb.coord_kind == abi.COORD_NODE_REF:
            coords = arr(sb.node_refs, sb.n_pts, np.uint32)
        else:
            coords = arr(sb.points, 2 * sb.n_pts, np.int32).reshape(-1, 2)
        return DisplayList(jobs, ops, rings, coords, dashes, sb.coord_kind, sb.scale, nodes=dl.nodes)
    finally:
        L.osmt_batch_shard_free(h)


def render_batch_multi(contexts, dl: DisplayList, out=None, labels=None, rgb=False):
    """osmt_render_batch_multi[_ex] over `contexts` (one per GPU): (RGBA8 [n, H, W, 4] — or packed RGB8 [n, H*W*3] with
    rgb=True —, all-reduced tile count).  `labels`: a LabelList for the whole batch, sliced per shard by the library."""
    shape = (dl.n_jobs, dl.dim * dl.dim * 3) if rgb else (dl.n_jobs, dl.dim, dl.dim, 4)
    if out is None:
        out = np.empty(shape, dtype=np.uint8)
    assert out.shape == shape and out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"]
    b = dl.as_batch()
    hs = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    cnt = C.c_uint64(0)
    outp = out.ctypes.data_as(C.POINTER(C.c_uint8))
    stride = dl.dim * dl.dim * (3 if rgb else 4)
    if labels is None and not rgb:
        check(load().osmt_render_batch_multi(hs, len(contexts), C.byref(b), outp, stride, C.byref(cnt)))
    else:
        lb = labels.as_batch() if labels is not None else None
        check(load().osmt_render_batch_multi_ex(hs, len(contexts), C.byref(b), C.byref(lb) if lb is not None else None,
                                                abi.MULTI_RGB8 if rgb else 0, outp, stride, C.byref(cnt)))
    return out, int(cnt.value)


def comm_unique_id():
    """osmt_comm_unique_id (rank 0): 128 bytes to hand to the other ranks."""
    buf = np.zeros(abi.COMM_ID_BYTES, dtype=np.uint8)
    check(load().osmt_comm_unique_id(buf.ctypes.data_as(C.POINTER(C.c_uint8))))
    return buf


def comm_init_rank(ctx, uid, rank, nranks):
    uid = np.ascontiguousarray(uid, dtype=np.uint8)
    assert uid.size == abi.COMM_ID_BYTES
    check(load().osmt_comm_init_rank(ctx._h, uid.ctypes.data_as(C.POINTER(C.c_uint8)), rank, nranks))


def comm_init_local(contexts):
    hs = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    check(load().osmt_comm_init_local(hs, len(contexts)))


def allreduce_tile_count(ctx, local):
    """osmt_allreduce_tile_count: RCCL sum of one uint64 over the communicator of `ctx` (collective)."""
    out = C.c_uint64(0)
    check(load().osmt_allreduce_tile_count(ctx._h, int(local), C.byref(out)))
    return int(out.value)


def allreduce_tile_count_enqueue(ctx, local, stream=None):
    """osmt_allreduce_tile_count_enqueue: the same reduction queued behind the render on `stream` (default: torch's
    current stream); no host synchronisation."""
    import torch

    s = torch.cuda.current_stream() if stream is None else stream
    check(load().osmt_allreduce_tile_count_enqueue(ctx._h, int(local), C.c_void_p(s.cuda_stream)))


def allreduce_tile_count_result(ctx, stream=None):
    """osmt_allreduce_tile_count_result: the most recent enqueued sum (waits for `stream` only)."""
    import torch

    s = torch.cuda.current_stream() if stream is None else stream
    out = C.c_uint64(0)
    check(load().osmt_allreduce_tile_count_result(ctx._h, C.c_void_p(s.cuda_stream), C.byref(out)))
    return int(out.value)


def reduce_tile_count(local_count, dist=None, device=None):
    """Sum of per-rank tile counts through torch.distributed (gloo on CPU, RCCL when the group is 'nccl'): the
    CPU-testable twin of allreduce_tile_count.  Returns the global count."""
    import torch

    t = torch.tensor([int(local_count)], dtype=torch.int64, device=device)
    if dist is not None and dist.is_initialized():
        dist.all_reduce(t)
    return int(t.item())
