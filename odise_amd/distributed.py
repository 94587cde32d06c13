"""Data-parallel sharding of the inference path: one process per GPU, images are independent units sharded
contiguously across ranks (detectron2's InferenceSampler convention, odise/data/build.py:145-151), no collective inside the model
forward, and exactly one exchange step: an all-gather of fixed-size per-image prediction records, replacing detectron2's pickled
`comm.gather` at evaluation time (odise/evaluation/evaluator.py:144).

On GPUs the exchange belongs to the library (`Exchange`: an RCCL communicator and a second HIP stream inside libodise_hip.so,
`ncclAllGather` over xGMI on device buffers the post-processing kernels wrote the records into; the launcher only broadcasts the
128-byte unique id over its CPU rendezvous).  `allgather_records` / `sum_confusion` are the same exchange on CPU tensors through
torch.distributed (gloo): what the CPU tests of the record layout and of uneven shards run, and what a CPU-side evaluator would use.

Record layout per image (all int32 so the gather is one contiguous buffer):
    panoptic_seg [H, W] | n_segments | segments [MAX_SEGMENTS, 3] = (id, isthing, category_id)
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch
import torch.distributed as dist

MAX_SEGMENTS = 100


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [begin, end) of rank (InferenceSampler: shard sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def max_shard(n_items: int, world: int) -> int:
    """Size of the largest shard of shard_range: what every rank pads its slice of an uneven exchange to."""
    return -(-n_items // world)


def valid_rows(gathered: np.ndarray) -> np.ndarray:
    """Rows of an `Exchange.allgather_records` result that hold a record (the padding rows start with -1), in rank order."""
    return gathered[gathered[:, 0] != -1]


def record_size(h: int, w: int) -> int:
    return h * w + 1 + MAX_SEGMENTS * 3


def pack_record(panoptic_seg, segments_info, out: torch.Tensor) -> None:
    """Fill one int32 record (a 1-D view of `out`) from a panoptic map (tensor/ndarray [H,W]) and its segments_info list."""
    hw = out.numel() - 1 - MAX_SEGMENTS * 3
    seg = torch.as_tensor(panoptic_seg).reshape(-1).to(out.dtype)
    assert seg.numel() == hw, (seg.numel(), hw)
    out[:hw] = seg.to(out.device)
    n = min(len(segments_info), MAX_SEGMENTS)
    table = torch.zeros(1 + MAX_SEGMENTS * 3, dtype=out.dtype)
    table[0] = n
    for i, s in enumerate(segments_info[:n]):
        table[1 + 3 * i: 4 + 3 * i] = torch.tensor([int(s["id"]), int(bool(s["isthing"])), int(s["category_id"])], dtype=out.dtype)
    out[hw:] = table.to(out.device)


def unpack_record(rec: torch.Tensor, h: int, w: int):
    rec = rec.cpu()
    seg = rec[: h * w].reshape(h, w).numpy().astype(np.int32)
    n = int(rec[h * w])
    t = rec[h * w + 1: h * w + 1 + 3 * n].reshape(n, 3).tolist()
    return seg, [{"id": a, "isthing": bool(b), "category_id": c} for a, b, c in t]


def allgather_records(local: torch.Tensor) -> torch.Tensor:
    """local [n_local, record] int32 on the rank's device -> [world * n_max, record] (ranks padded to the largest shard with -1 rows).
    One collective for the size vector (8 bytes per rank) and one all-gather of the payload."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    n_max = int(max(int(s) for s in sizes))
    padded = torch.full((n_max, local.shape[1]), -1, dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = torch.empty((world * n_max, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded)
    keep = torch.cat([torch.arange(int(s)) + r * n_max for r, s in enumerate(sizes)]).to(local.device)
    return out[keep]


def sum_confusion(conf: torch.Tensor) -> torch.Tensor:
    """Semantic evaluation's exchange step: every rank accumulates an int64 [(K+1),(K+1)] confusion matrix over its image shard
    (odise_hip_semantic_confusion); one all-reduce(SUM) of (K+1)^2 counters replaces detectron2's pickled gather of per-image
    predictions (SemSegEvaluator.evaluate, odise/evaluation/d2_evaluator.py:63).  Integer sums: the result is independent of the
    rank count and order."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return conf
    out = conf.clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM)
    return out


class Exchange:
    """The library-owned RCCL exchange (include/odise_hip.h: odise_hip_comm_*).  `broadcast(id_bytes_or_None) -> id_bytes` must deliver
    rank 0's unique id to every rank (e.g. `gloo_broadcast`); a world of one rank needs none and takes the same code path."""

    def __init__(self, ctx, rank: int = 0, world: int = 1, broadcast=None):
        import ctypes as C
        from ._lib import COMM_ID_BYTES, check
        self.ctx, self.rank, self.world = ctx, rank, world
        buf = (C.c_ubyte * COMM_ID_BYTES)()
        if rank == 0:
            check(ctx.lib.odise_hip_comm_unique_id(buf), "comm_unique_id")
        if world > 1:
            if broadcast is None:
                raise ValueError("world > 1 needs a broadcast callable for the communicator id")
            data = broadcast(bytes(buf) if rank == 0 else None)
            buf = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(data)
        check(ctx.lib.odise_hip_comm_init(ctx.h, buf, rank, world), "comm_init")
        try:   # RCCL writes a version banner through C stdio at initialisation: push it out now, not after whatever Python prints last
            C.CDLL(None).fflush(None)
        except (OSError, AttributeError):
            pass

    def allgather(self, local, out) -> None:
        """out [world * n] = every rank's local [n] (device int32 buffers: runtime.DeviceArray).  Asynchronous: runs on the library's
        exchange stream after everything queued on the compute stream so far; `wait()` joins it."""
        from ._lib import check
        n = int(np.prod(local.shape))
        assert int(np.prod(out.shape)) == n * self.world and local.dtype == np.int32 and out.dtype == np.int32
        check(self.ctx.lib.odise_hip_allgather_predictions(self.ctx.h, local.ptr, n, out.ptr), "allgather_predictions")

    def allgather_records(self, local, n_records: int, max_records: int, out) -> None:
        """Uneven shards: this rank's first `n_records` rows of local [>= n_records, record] -> out [world * max_records, record]; the library
        pads every rank's slice to `max_records` rows with -1 (`valid_rows` drops them).  max_records must be the same on every rank:
        `max_shard(n_items, world)` of the batch being exchanged."""
        from ._lib import check
        rec = int(out.shape[-1])
        assert 0 <= n_records <= max_records and tuple(out.shape) == (self.world * max_records, rec) and out.dtype == np.int32
        assert n_records == 0 or (local.dtype == np.int32 and int(local.shape[-1]) == rec and int(local.shape[0]) >= n_records)
        check(self.ctx.lib.odise_hip_allgather_records(self.ctx.h, local.ptr if n_records else None, n_records, max_records, rec, out.ptr), "allgather_records")

    def allreduce_sum_i64(self, data) -> None:
        from ._lib import check
        assert data.dtype == np.int64
        check(self.ctx.lib.odise_hip_allreduce_sum_i64(self.ctx.h, data.ptr, int(np.prod(data.shape))), "allreduce_sum_i64")

    def wait(self, host: bool = True) -> None:
        from ._lib import check
        check(self.ctx.lib.odise_hip_comm_wait(self.ctx.h, 1 if host else 0), "comm_wait")

    def close(self) -> None:
        if self.ctx is not None and self.ctx.h:
            self.ctx.lib.odise_hip_comm_destroy(self.ctx.h)
        self.ctx = None


def gloo_broadcast(data):
    """Broadcast rank 0's bytes over the (CPU) default process group of torch.distributed - the launcher's rendezvous."""
    box = [data]
    dist.broadcast_object_list(box, src=0)
    return box[0]
