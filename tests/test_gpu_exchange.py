"""GPU: the library-owned RCCL exchange (include/odise_hip.h odise_hip_comm_*) with a communicator of ONE rank - the
code path `bench.py --gpus N` runs on every rank (unique id, ncclCommInitRank, all-gather / all-reduce on the exchange stream ordered
after the compute stream, host-side and stream-side joins).  World sizes > 1 are the driver's 8-GPU run; the record layout and uneven
shards are covered on CPU (tests/test_distributed_cpu.py, gloo, world size 2)."""
import numpy as np
import pytest
import torch

from odise_amd import distributed as D

pytestmark = pytest.mark.gpu


def test_one_rank_allgather_and_allreduce(ctx):
    ex = D.Exchange(ctx, 0, 1)
    try:
        import ctypes as C
        r, w = C.c_int(-1), C.c_int(-1)
        assert ctx.lib.odise_hip_comm_info(ctx.h, C.byref(r), C.byref(w)) == 0 and (r.value, w.value) == (0, 1)
        h, wd = 33, 17
        rec = D.record_size(h, wd)
        host = torch.zeros((3, rec), dtype=torch.int32)
        for i in range(3):
            D.pack_record(np.full((h, wd), i + 1, np.int32), [{"id": 1, "isthing": bool(i % 2), "category_id": 40 + i}], host[i])
        local = ctx.to_device(host.numpy())
        out = ctx.zeros((3, rec), np.int32)
        for _ in range(3):                      # repeated collectives reuse the communicator and its stream
            ex.allgather(local, out)
        ex.wait(host=False)                     # stream-side join, then ordinary stream work sees the result
        got = out.numpy()
        np.testing.assert_array_equal(got, host.numpy())
        seg, info = D.unpack_record(torch.from_numpy(got[2]), h, wd)
        assert (seg == 3).all() and info == [{"id": 1, "isthing": False, "category_id": 42}]
        conf = ctx.to_device(np.arange(12, dtype=np.int64).reshape(3, 4))
        ex.allreduce_sum_i64(conf)
        ex.wait(host=True)
        np.testing.assert_array_equal(conf.numpy(), np.arange(12, dtype=np.int64).reshape(3, 4))
    finally:
        ex.close()
    r, w = C.c_int(-1), C.c_int(-1)
    assert ctx.lib.odise_hip_comm_info(ctx.h, C.byref(r), C.byref(w)) == 0 and w.value == 0


def test_collective_without_communicator_fails_loudly(ctx):
    a = ctx.zeros((4,), np.int32)
    rc = ctx.lib.odise_hip_allgather_predictions(ctx.h, a.ptr, 4, a.ptr)
    assert rc != 0 and b"comm_init" in ctx.lib.odise_hip_last_error()


def test_uneven_shard_is_padded_by_the_library(ctx):
    """odise_hip_allgather_records with a one-rank communicator: 2 of at most 3 records -> the gathered buffer holds them and one -1 row,
    both from a separate local buffer and in place (local aliasing this rank's slice), and an empty shard gathers padding only."""
    ex = D.Exchange(ctx, 0, 1)
    try:
        rec = D.record_size(5, 7)
        host = (np.arange(3 * rec, dtype=np.int32).reshape(3, rec) % 50)
        local = ctx.to_device(host)
        out = ctx.zeros((3, rec), np.int32)
        ex.allgather_records(local, 2, 3, out)
        ex.wait(host=True)
        got = out.numpy()
        np.testing.assert_array_equal(got[:2], host[:2])
        assert (got[2] == -1).all() and D.valid_rows(got).shape == (2, rec)
        inplace = ctx.to_device(host)                       # the records already sit in this rank's slice of the gather buffer
        ex.allgather_records(inplace, 1, 3, inplace)
        ex.wait(host=True)
        got = inplace.numpy()
        np.testing.assert_array_equal(got[0], host[0])
        assert (got[1:] == -1).all()
        ex.allgather_records(None, 0, 2, out.view((2, rec)))
        ex.wait(host=True)
        assert (out.view((2, rec)).numpy() == -1).all()
        rc = ctx.lib.odise_hip_allgather_records(ctx.h, local.ptr, 4, 3, rec, out.ptr)
        assert rc != 0 and b"allgather_records" in ctx.lib.odise_hip_last_error()
    finally:
        ex.close()
