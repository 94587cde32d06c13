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


def test_bench_rehearsal_at_eight_ranks(tmp_path):
    """`bench.py --gpus 8` as the driver's scaling run starts it, on this host without a GPU (the 8-rank path had never run
    anywhere): eight ranks under torch.distributed.run, each with the stand-in library of tests/stub_lib.py (device memory = host memory,
    kernels do nothing, well-formed tables and records), everything on the host side real - the synthetic weights ONCE per host in /dev/shm
    (odise_amd.synthetic.synthetic_state_shared) and mapped by every rank, 1630 tensors handed to the library per rank, the calibration
    passes, rank 0's communicator id reaching every rank over the gloo rendezvous, K timed steps with a barrier on both sides, one JSON line.
    Asserts the line, one set-up report per rank with bounded time and memory, and that nothing is left in /dev/shm."""
    import glob
    import re
    sys.path.insert(0, HERE)
    import stub_lib
    stub = stub_lib.build_stub(str(tmp_path / "libodise_stub.so"))
    before = set(glob.glob("/dev/shm/odise_synth_*"))
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(ODISE_HIP_LIB=stub, OMP_NUM_THREADS="1")
    bench = os.path.join(HERE, "..", "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-inclusive"],
                       capture_output=True, text=True, timeout=900, env=e)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["rccl_ranks"] == 8 and d["config"]["units_per_step_per_gpu"] == 4 and d["scaling"] == "weak"
    assert abs(d["value"] - 8 * 4 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]                 # whole-job images/s over all ranks
    assert d["exchange"]["segments_per_image"] == [6, 6, 6, 6] and d["exchange"]["records_bytes_per_rank"] == 4 * (1024 * 1024 + 301) * 4
    setups = re.findall(r"\[bench rank (\d+)/8\] set-up ([0-9.]+) s, peak host RSS ([0-9.]+) GB", r.stderr)
    assert sorted(int(a) for a, _, _ in setups) == list(range(8)), r.stderr[-3000:]
    secs, rss = [float(b) for _, b, _ in setups], [float(c) for _, _, c in setups]
    print(f"8 ranks on {os.cpu_count()} host cores: set-up {min(secs):.0f}-{max(secs):.0f} s, peak RSS per rank {min(rss):.1f}-{max(rss):.1f} GB "
          f"(the 5.1 GB of shared weight pages count in every rank that maps them)")
    # ru_maxrss counts the shared /dev/shm pages in every rank that maps them: 8 x (5.1 shared + private) - the private part is what adds up
    assert max(secs) < 600 and max(rss) < 8.0 and sum(rss) - 7 * 5.2 < 16.0, (secs, rss)
    finals = re.findall(r"\[bench rank (\d+)/8\] device (\d+) .*rccl_ranks 8", r.stderr)
    assert sorted((int(a), int(b)) for a, b in finals) == [(k, k) for k in range(8)]            # rank k drives device k
    assert set(glob.glob("/dev/shm/odise_synth_*")) <= before, "the shared weight file was left behind in /dev/shm"
