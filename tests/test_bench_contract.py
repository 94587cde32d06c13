"""The bench line the driver parses: every committed `profiles/r04_bench_*.json` / `r05_bench_*.json` (rank 0's one JSON line of a `bench.py` run on
an MI355X, final binaries of the two rounds) carries the keys of the measurement contract, with consistent arithmetic.  CPU only: guards the
format, not the numbers."""
import glob
import json
import os

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r04_bench_*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r05_bench_*.json")) +
               glob.glob(os.path.join(ROOT, "profiles", "r06_bench_*.json")))


def test_profiles_present():
    """Its own test, not a condition inside the parametrized one: a missing or renamed profile set must fail, not generate zero cases."""
    assert len(LINES) >= 8 + 12 + 10, LINES


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_bench_line_contract(path):
    d = json.load(open(path))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f16" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    units = d["config"]["units_per_step_per_gpu"]
    assert abs(d["value"] - units * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]          # whole-job throughput = units / step time
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    if "launch_us" in r:   # the dominant kernel INSIDE the timed step: algorithmic FLOPs per launch / mean launch time of the probe
        assert r["where"].startswith("in the timed step")
        assert abs(r["achieved"] - r["algorithmic_flops_per_launch"] / (r["launch_us"] * 1e-6) / 1e12) < 1e-6 * r["achieved"]
        assert r["launch_us_min"] <= r["launch_us"] <= r["launch_us_max"] and r["launches_timed"] >= d["steps"]
        tp = r["traffic_profiled"]   # not measured in this run: a pointer to the PMC passes (the previous round's tree was run from a copy without profiles/)
        assert r["traffic"] is None and (tp["hbm_bytes_per_launch"] > 541e6 if tp is not None else "previous_round_tree" in path)
        iso = r["isolated"]
        assert iso["launch_us"] < r["launch_us"] and abs(iso["frac"] - iso["achieved"] / 2500.0) < 1e-9   # alone on the idle chip it is faster
        assert 0.0 < r["step_frac"] < r["frac"] < iso["frac"] < 1.0


def test_headline_line_has_the_cpu_baseline():
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_full_b4_1024.json")))
    assert d["metric"] == "panoptic-inference images/sec @1024x1024" and d["unit"] == "images/s"
    assert d["config"]["workload"].startswith("BASELINE configs[2]") and d["config"]["units_per_step_per_gpu"] == 4
    assert d["config"]["rccl_ranks"] == d["n_gpus"] == 1 and d["config"]["batches_in_flight"] == 1
    assert d["config"]["vae_chunk_bytes"] == 0 and d["config"]["clip_ln_fold"] == 0          # the library's defaults, nothing pinned
    assert "launch_us" in d["roofline"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    # whole images through the oracle (heads and post-processing included), not crops scaled up
    assert c["kind"] == "port" and c["unit"] == "images/s" and c["cores"] >= 1 and len(c["image_seconds_live"]) >= 2
    assert abs(c["value"] - len(c["image_seconds_live"]) / sum(c["image_seconds_live"])) < 1e-9
    assert 0 < c["value_literal"] < c["value"] < d["value"]
    # the decisions the timed kernels made are not degenerate: every picture has instances, the first one segments (bench.py check_exchange)
    e = d["exchange"]
    assert min(e["instances_per_image"]) > 0 and e["segments_image0"] > 0 and len(e["segments_per_image"]) == 4


def test_in_flight_line_says_so():
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_full_b4_1024_in_flight2.json")))
    assert d["config"]["batches_in_flight"] == 2 and d["config"]["one_batch_alone_ms"] > 0
    assert "in_flight_1" in d["exchange"]


def test_round5_headline_and_its_same_box_variants():
    """Round 5: the headline line, the two same-box A/B lines (block-wide epilogues; round 4's MFMA shape) and the encoder-prefetch line say what
    they are in `config`, and the decisions the timed kernels made cover several segments per picture (bench.py check_exchange)."""
    load = lambda n: json.load(open(os.path.join(ROOT, "profiles", n)))
    d = load("r05_bench_full_b4_1024.json")
    assert d["config"]["gemm_flags"] == 0 and d["config"]["pipeline"] is None and d["config"]["batches_in_flight"] == 1
    assert d["config"]["vae_chunk_bytes"] == 0 and d["config"]["clip_ln_fold"] == 0
    seg = d["exchange"]["segments_per_image"]
    assert len(seg) == 4 and min(seg) >= 2 and sum(seg) >= 5 * 4, seg
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and 0 < c["value"] < d["value"] / 100                      # the GPU / CPU ratio is reported, not the target
    blk, m32, pf = (load("r05_bench_full_b4_1024_" + n + ".json") for n in ("block_epilogues", "mfma32_block_epilogues", "encoder_prefetch"))
    assert blk["config"]["gemm_flags"] == 32768 and m32["config"]["gemm_flags"] == 32768
    assert m32["ms_per_step"] > blk["ms_per_step"] > d["ms_per_step"]                     # what each of the two kernel changes is worth, one box
    assert pf["config"]["pipeline"].startswith("encoder-prefetch") and pf["config"]["batches_in_flight"] == 1
    assert 0.95 * d["ms_per_step"] < pf["ms_per_step"] < d["ms_per_step"]                 # work-conserving: a percent, not the idle lane's 14 ms


def test_round6_line_carries_the_unet_share_and_the_host_fed_value():
    """Round 6: the metric's second half - "UNet MFMA %peak" - is on the line the driver parses: the UNet stage as it runs inside
    the timed step (stage marks) and alone after it, both as fractions of the 2.5 PFLOP/s peak; the evaluator-faithful rate (pictures from host
    memory every step) as a top-level key next to `value`; and the same-box line of the previous round's tree beside the headline."""
    load = lambda n: json.load(open(os.path.join(ROOT, "profiles", n)))
    d = load("r06_bench_full_b4_1024.json")
    assert d["metric"] == "panoptic-inference images/sec @1024x1024" and d["config"]["workload"].startswith("BASELINE configs[2]")
    r = d["roofline"]
    u, ui = r["unet"], r["unet_isolated"]
    assert u["crops"] == ui["crops"] == 16 and u["steps"] == d["steps"] and u["where"].startswith("in the timed step")
    for x in (u, ui):
        assert abs(x["frac"] - x["crops"] * 0.7401e12 / (x["ms"] * 1e-3) / 2.5e15) < 1e-9 and abs(x["achieved"] - x["frac"] * 2500.0) < 1e-6
    assert ui["ms"] < u["ms"] < d["ms_per_step"] and 0.1 < u["frac"] < ui["frac"] < 0.4        # alone it is faster; inside the step it shares the chip
    assert 0 < d["value_host_fed"] < d["value"] and d["value_host_fed"] == d["inclusive"]["host_u8"]["value"]
    seg = d["exchange"]["segments_per_image"]
    assert len(seg) == 4 and min(seg) >= 2 and sum(seg) >= 20
    prev = load("r06_bench_full_b4_1024_steps20_previous_round_tree.json")
    now = [load("r06_bench_full_b4_1024_steps20.json"), load("r06_bench_full_b4_1024_steps20_again.json")]
    assert all(n["ms_per_step"] < prev["ms_per_step"] for n in now) and "unet" not in prev["roofline"]          # the same box, the same hour
    pf = load("r06_bench_full_b4_1024_encoder_prefetch.json")
    assert pf["config"]["pipeline"].startswith("encoder-prefetch") and abs(pf["ms_per_step"] - now[0]["ms_per_step"]) < 0.02 * now[0]["ms_per_step"]
