"""Worker of tests/test_launch_cpu.py: the launcher side of `bench.py --gpus N` without a GPU.  Like bench.py it calls
`launch.ensure_world(N)` first (outside a launcher this re-executes the script as N ranks), protects stdout, joins the gloo rendezvous,
initialises the library-owned exchange - here against the recording stand-in of tests/test_distributed_cpu.py instead of libodise_hip.so -
and rank 0 prints ONE JSON line.  `--fail-rank R` makes rank R die before the communicator is initialised."""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from odise_amd import launch  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--gpus", type=int, default=1)
p.add_argument("--fail-rank", type=int, default=-1)
p.add_argument("--log-dir", required=True)
args = p.parse_args()

launch.ensure_world(args.gpus)
out = launch.protect_stdout()
ctypes.CDLL(None).puts(b"NCCL WARN stand-in: a C-level print aimed at stdout")
rank, world, local_rank = launch.world_from_env()

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

from odise_amd import distributed as D  # noqa: E402
from test_distributed_cpu import _FakeArray, _FakeLib  # noqa: E402

if world > 1:
    dist.init_process_group("gloo")
if rank == args.fail_rank:
    raise SystemExit(f"rank {rank}: simulated failure before comm_init")
log = []


class Ctx:
    lib = _FakeLib(rank, log)
    h = 1


ex = D.Exchange(Ctx, rank, world, D.gloo_broadcast if world > 1 else None)
ex.allgather(_FakeArray((2, 7), np.int32), _FakeArray((world * 2, 7), np.int32))
ex.wait(True)
ex.close()
with open(os.path.join(args.log_dir, f"rank{rank}.json"), "w") as f:
    json.dump([[e[0]] + [x.hex() if isinstance(x, bytes) else x for x in e[1:]] for e in log], f)
if world > 1:
    dist.barrier()
if rank == 0:
    print(json.dumps({"n_gpus": world, "local_rank": local_rank}), file=out, flush=True)
if world > 1:
    dist.destroy_process_group()
