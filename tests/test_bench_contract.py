"""CPU checks of the measurement contract: bench.py parses its flags without a GPU and the committed
bench line (profiles/) carries every field the driver and the judge read."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_help_runs_without_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in r.stdout


def test_committed_bench_line_has_the_contract_fields():
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    line = json.load(open(os.path.join(ROOT, "profiles", "r02_v5_bench.json")))
    assert line["metric"] == baseline["metric"]
    for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["vs_baseline"] is None and line["data"] == "synthetic" and line["dtype"] == "f64"
    assert "workload" in line["config"] and "model" not in line["config"]
    rf = line["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["traffic"] is None or rf["traffic"] > 0
    cb = line["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] == 1 and cb["value"] > 0 and cb["sample"]
    assert line["value"] > 50 * cb["value"]                 # north_star: >= 50x the CPU path
    # round 2: the honest-roofline keys, the steady-state figure, the launch count, which runtime and sources
    for k in ("algorithmic_GBps", "requested_bytes_per_launch", "frac_requested", "frac_traffic", "kernel_ms", "timing"):
        assert k in rf, k
    assert line["launches_per_step"] == 4 and line["value_steady"] > 0 and line["steady_state"]["steps"] == line["steps"]
    assert line["hip_runtime"]["path"] and len(line["source_sha16"]) == 16
    assert line["config"]["fresh_batch_every_step"] is True
    poly = line["polygon_obstacles"]["roofline"]
    assert poly["kernel"] == "edges_polygons_kernel" and poly["algorithmic_bytes_per_launch"] > 0


def test_committed_c3_line_and_traffic_file():
    c3 = json.load(open(os.path.join(ROOT, "profiles", "r02_v5_bench_c3.json")))
    assert c3["config"]["edge"] == "DubinsEdge" and c3["cpu_baseline"]["kind"] == "port" and c3["value"] > 0
    assert c3["roofline"]["bound"] == "valu_fp64"
    tr = json.load(open(os.path.join(ROOT, "profiles", "r02_traffic.json")))
    assert tr["kernel"].startswith("nn_tile_kernel") and tr["traffic_bytes_per_launch"] > 0 and len(tr["source_sha16"]) == 16
    # round-1 verdict: write amplification of the hit records (33.4 MB then; 14.8 MB in the middle of round 2; the
    # final build's register-bound fused kernel adds ~4 MB of scratch)
    assert tr["WRITE_SIZE_bytes_per_launch"] < 20e6
    assert tr["SQ_INSTS_VALU_per_launch"] > 0 and 0 < tr["wait_ratio"] < 1
