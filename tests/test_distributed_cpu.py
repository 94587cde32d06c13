"""CPU (gloo, world_size 2): the data-parallel sharding and the single prediction all-gather of the inference path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from odise_amd import distributed as D


def test_shard_ranges_cover_everything_contiguously():
    for n in (0, 1, 7, 8, 9, 100):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_record_roundtrip():
    h, w = 6, 5
    seg = np.arange(h * w, dtype=np.int32).reshape(h, w) % 4
    info = [{"id": 1, "isthing": True, "category_id": 17}, {"id": 2, "isthing": False, "category_id": 120}]
    rec = torch.zeros(D.record_size(h, w), dtype=torch.int32)
    D.pack_record(seg, info, rec)
    seg2, info2 = D.unpack_record(rec, h, w)
    np.testing.assert_array_equal(seg, seg2)
    assert info == info2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_images, h, w, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = D.shard_range(n_images, rank, world)
    local = torch.zeros((e - b, D.record_size(h, w)), dtype=torch.int32)
    for i, img in enumerate(range(b, e)):   # "prediction" of image `img`: a map filled with img+1 and one segment per image
        D.pack_record(np.full((h, w), img + 1, np.int32), [{"id": 1, "isthing": bool(img % 2), "category_id": img}], local[i])
    allrec = D.allgather_records(local)
    q.put((rank, allrec.numpy().copy()))   # by value: a tensor would travel as a shared-memory handle that dies with this process
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_images", [4, 5])
def test_allgather_predictions_gloo_world2(n_images):
    h, w, world = 4, 3, 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_images, h, w, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        rec = torch.from_numpy(results[r])
        assert rec.shape == (n_images, D.record_size(h, w))      # every rank holds every image, in global order
        for img in range(n_images):
            seg, info = D.unpack_record(rec[img], h, w)
            assert (seg == img + 1).all()
            assert info == [{"id": 1, "isthing": bool(img % 2), "category_id": img}]


def _conf_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.eval_ops import semantic_confusion
    K, n_images = 4, 5
    b, e = D.shard_range(n_images, rank, world)
    conf = torch.zeros((K + 1, K + 1), dtype=torch.int64)
    for img in range(b, e):
        rng = np.random.default_rng(img)
        conf += torch.from_numpy(semantic_confusion(rng.standard_normal((K, 6, 7)).astype(np.float32), rng.integers(0, K, (6, 7))))
    q.put((rank, D.sum_confusion(conf).numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_confusion_matrix_allreduce_gloo_world2():
    from oracle.eval_ops import semantic_confusion
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_conf_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = np.zeros((5, 5), np.int64)
    for img in range(5):
        rng = np.random.default_rng(img)
        ref += semantic_confusion(rng.standard_normal((4, 6, 7)).astype(np.float32), rng.integers(0, 4, (6, 7)))
    for r in range(world):
        np.testing.assert_array_equal(results[r], ref)


# ---- the launcher side of the library-owned exchange (odise_amd.distributed.Exchange) at world size 2, with the library replaced by a
# recorder: rank 0 alone asks for the communicator id, the id travels over the CPU rendezvous (gloo_broadcast, as bench.py does for
# --gpus N), every rank initialises the communicator with the SAME 128 bytes and its own rank; the collectives hand the library the
# element counts it expects.  (The RCCL half of the path runs in tests/test_gpu_exchange.py with a one-rank communicator.)
class _FakeLib:
    def __init__(self, rank, log):
        self.rank, self.log = rank, log

    def odise_hip_comm_unique_id(self, buf):
        self.log.append(("unique_id", self.rank))
        for i in range(len(buf)):
            buf[i] = (i * 7 + 3) % 256
        return 0

    def odise_hip_comm_init(self, h, buf, rank, world):
        self.log.append(("init", bytes(buf), rank, world))
        return 0

    def odise_hip_allgather_predictions(self, h, local, n, out):
        self.log.append(("allgather", int(n)))
        return 0

    def odise_hip_allgather_records(self, h, local, n_records, max_records, record_len, out):
        self.log.append(("allgather_records", local is not None, int(n_records), int(max_records), int(record_len)))
        return 0

    def odise_hip_allreduce_sum_i64(self, h, data, n):
        self.log.append(("allreduce", int(n)))
        return 0

    def odise_hip_comm_wait(self, h, host):
        self.log.append(("wait", int(host)))
        return 0

    def odise_hip_comm_destroy(self, h):
        self.log.append(("destroy",))
        return 0


class _FakeArray:
    def __init__(self, shape, dtype):
        self.shape, self.dtype, self.ptr = shape, np.dtype(dtype), 0x1000


def _exchange_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log = []

    class Ctx:
        lib = _FakeLib(rank, log)
        h = 1

    ex = D.Exchange(Ctx, rank, world, D.gloo_broadcast)
    B, rec = 3, D.record_size(4, 5)
    ex.allgather(_FakeArray((B, rec), np.int32), _FakeArray((world * B, rec), np.int32))
    # the last batch of 3 images over 2 ranks: shards of 2 and 1 (shard_range), both padded to max_shard = 2 by the library
    b, e = D.shard_range(3, rank, world)
    ex.allgather_records(_FakeArray((B, rec), np.int32), e - b, D.max_shard(3, world), _FakeArray((world * D.max_shard(3, world), rec), np.int32))
    # ... and a batch that leaves the last rank without any image
    b1, e1 = D.shard_range(1, rank, world)
    ex.allgather_records(_FakeArray((B, rec), np.int32), e1 - b1, 1, _FakeArray((world, rec), np.int32))
    ex.allreduce_sum_i64(_FakeArray((5, 5), np.int64))
    ex.wait(True)
    ex.close()
    q.put((rank, log))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_launcher_side_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    logs = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [e for e in logs[0] if e[0] == "unique_id"] == [("unique_id", 0)] and not [e for e in logs[1] if e[0] == "unique_id"]
    inits = {r: [e for e in logs[r] if e[0] == "init"][0] for r in range(world)}
    assert inits[0][1] == inits[1][1] == bytes((i * 7 + 3) % 256 for i in range(128))     # the same id everywhere
    assert (inits[0][2], inits[0][3]) == (0, 2) and (inits[1][2], inits[1][3]) == (1, 2)
    rec = D.record_size(4, 5)
    assert ("allgather_records", True, 2, 2, rec) in logs[0] and ("allgather_records", True, 1, 2, rec) in logs[1]      # 3 images: 2 + 1, padded to 2
    assert ("allgather_records", True, 1, 1, rec) in logs[0] and ("allgather_records", False, 0, 1, rec) in logs[1]     # 1 image: rank 1 sends padding only
    for r in range(world):
        assert ("allgather", 3 * D.record_size(4, 5)) in logs[r] and ("allreduce", 25) in logs[r] and ("wait", 1) in logs[r] and logs[r][-1] == ("destroy",)


def test_exchange_needs_a_broadcast_beyond_one_rank():
    class Ctx:
        lib = _FakeLib(0, [])
        h = 1
    with pytest.raises(ValueError):
        D.Exchange(Ctx, 0, 2, None)


def test_valid_rows_drops_the_library_padding():
    """What `odise_hip_allgather_records` leaves behind for uneven shards: rank r's records at row r * max_shard, -1 rows after them."""
    rec = D.record_size(2, 3)
    world, n_items = 3, 4                                    # shards of 2, 1, 1 padded to 2
    m = D.max_shard(n_items, world)
    assert m == 2 and [D.shard_range(n_items, r, world) for r in range(world)] == [(0, 2), (2, 3), (3, 4)]
    out = np.full((world * m, rec), -1, np.int32)
    item = 0
    for r in range(world):
        b, e = D.shard_range(n_items, r, world)
        for i in range(e - b):
            out[r * m + i] = item                              # a record's first element is a panoptic id >= 0
            item += 1
    rows = D.valid_rows(out)
    assert rows.shape == (n_items, rec) and rows[:, 0].tolist() == [0, 1, 2, 3]
