"""One process per GPU, started by the program itself.

The reference's entry points spawn their ranks themselves - `tools/train_net.py:390-399` hands `main` to detectron2's `launch`, which forks
one process per GPU - so `python tools/train_net.py --num-gpus 8 --eval-only` is the whole command.  `ensure_world(n)` gives a script of this
repository the same property: called with `n > 1` outside any launcher (no WORLD_SIZE in the environment) it re-executes the script as `n` ranks
under `torch.distributed.run` on 127.0.0.1 (one rank per GPU, LOCAL_RANK = device index) and exits with the launcher's status - non-zero as
soon as ANY rank fails, so a run that reports `n` GPUs cannot have executed on fewer.  Inside a launcher (the driver's own
`python -m torch.distributed.run ... bench.py --gpus N`) it only checks that the world size is the requested one.

`protect_stdout()` makes "stdout carries exactly one JSON line" hold against libraries that print through C stdio (RCCL's NCCL WARN /
version banner go to stdout): file descriptor 1 is pointed at stderr for the rest of the process and the returned text stream writes to
the ORIGINAL stdout."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import Optional, Sequence


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def world_from_env():
    """(rank, world, local_rank) as torch.distributed.run exports them; (0, 1, 0) outside a launcher."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def spawn_ranks(n: int, script: str, argv: Sequence[str], env: Optional[dict] = None, timeout: Optional[float] = None) -> int:
    """Run `script argv` as n ranks of one node under torch.distributed.run; stdout / stderr are inherited.  Returns the launcher's exit
    status (0 only if every rank exited 0; the launcher tears the other ranks down when one fails)."""
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL's intra-node transport needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script, *argv]
    try:
        return subprocess.run(cmd, env=e, timeout=timeout).returncode
    except subprocess.TimeoutExpired:
        return 124


def ensure_world(n: int, script: Optional[str] = None, argv: Optional[Sequence[str]] = None) -> None:
    """See the module docstring.  Returns in a process that IS one of the n ranks (or the only one for n == 1); otherwise it does not return."""
    _, world, _ = world_from_env()
    if "WORLD_SIZE" not in os.environ:
        if n <= 1:
            return
        rc = spawn_ranks(n, script if script is not None else os.path.abspath(sys.argv[0]), list(sys.argv[1:] if argv is None else argv))
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(rc if rc != 0 else 0)   # the ranks have done the work (and rank 0 printed the line); nothing to unwind here
    if world != n:
        raise SystemExit(f"--gpus {n} but the launcher started WORLD_SIZE={world} ranks")


def protect_stdout():
    """-> text stream on the original stdout; fd 1 now IS stderr (C-level prints of loaded libraries can no longer reach the line)."""
    sys.stdout.flush()
    keep = os.dup(1)
    os.dup2(2, 1)
    out = os.fdopen(keep, "w", buffering=1)
    sys.stdout = sys.stderr        # Python-level prints that forget `file=` follow the C-level ones
    return out
