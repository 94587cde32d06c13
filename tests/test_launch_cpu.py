"""`--gpus N` must be impossible to mis-launch (the reference's entry point spawns its own ranks, tools/train_net.py:390-399):
called outside a launcher the program starts its N ranks itself, stdout carries exactly one line whatever C-level libraries print, a
rank that dies takes the whole run down with a non-zero status, and under a foreign launcher a mismatching world size is refused.
CPU only (gloo); the library is the recording stand-in of tests/test_distributed_cpu.py."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "spawn_worker.py")


def _run(args, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, WORKER, *args], capture_output=True, text=True, timeout=timeout, env=e)


def test_self_spawn_runs_n_ranks_and_prints_one_line(tmp_path):
    r = _run(["--gpus", "2", "--log-dir", str(tmp_path)])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                       # the C-level print went to stderr
    assert json.loads(lines[0]) == {"n_gpus": 2, "local_rank": 0}
    assert "NCCL WARN stand-in" in r.stderr
    logs = {k: json.load(open(tmp_path / f"rank{k}.json")) for k in (0, 1)}
    inits = {k: [e for e in logs[k] if e[0] == "init"][0] for k in (0, 1)}
    assert inits[0][1] == inits[1][1] and len(bytes.fromhex(inits[0][1])) == 128      # rank 0's communicator id reached rank 1
    assert inits[0][2:] == [0, 2] and inits[1][2:] == [1, 2]
    assert [e[0] for e in logs[0]].count("unique_id") == 1 and [e[0] for e in logs[1]].count("unique_id") == 0
    for k in (0, 1):
        assert ["allgather", 14] in logs[k] and logs[k][-1] == ["destroy"]


def test_a_dying_rank_fails_the_whole_run(tmp_path):
    r = _run(["--gpus", "2", "--fail-rank", "1", "--log-dir", str(tmp_path)])
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.strip()]                       # no result line from a broken run


def test_single_rank_needs_no_launcher(tmp_path):
    r = _run(["--gpus", "1", "--log-dir", str(tmp_path)])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip()) == {"n_gpus": 1, "local_rank": 0}


def test_world_size_mismatch_under_a_foreign_launcher_is_refused(tmp_path):
    r = _run(["--gpus", "4", "--log-dir", str(tmp_path)], env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
