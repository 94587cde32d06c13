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
