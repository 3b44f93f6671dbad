"""CPU checks of the measurement contract: bench.py parses its flags without a GPU, never reports a run it was not asked
for, and the committed bench lines (profiles/r03_*) carry every field the driver and the judge read -- with every roofline
fraction in (0, 1] and every attached counter file keyed by the kernel sources and the run it was taken on."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def _line(name):
    return json.load(open(os.path.join(P, name)))


def test_bench_help_runs_without_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--config", "--shard", "--grid", "--obstacles", "--batch"):
        assert flag in r.stdout


def test_gpus_n_without_a_launcher_never_reports_one_gpu():
    """round-2 verdict: `python bench.py --gpus 8` silently ran ONE rank and printed n_gpus = 1.  Now the N ranks are started
    (torch.distributed.run on 127.0.0.1) before anything touches the GPU; here, without a GPU, they fail -- and so does the
    call, without printing a bench line."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "n_gpus" not in r.stdout
    assert "starting 2 ranks" in r.stderr
    # a launcher that gives a different world size than --gpus is refused as well
    env["WORLD_SIZE"], env["RANK"], env["LOCAL_RANK"] = "1", "0", "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and "n_gpus" not in r.stdout


def _check_roofline(rf):
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    if rf["frac"] is not None:
        assert 0.0 < rf["frac"] <= 1.0, rf
        assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["traffic"] is None or rf["traffic"] > 0


def _check_common(line, baseline):
    assert line["metric"] == baseline["metric"]
    for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "run_key", "source_sha16", "lib_sha16"):
        assert k in line, k
    assert line["vs_baseline"] is None and line["data"] == "synthetic" and line["dtype"] == "f64" and line["n_gpus"] == 1
    assert "workload" in line["config"] and "model" not in line["config"]
    _check_roofline(line["roofline"])
    cb = line["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] == 1 and cb["value"] > 0 and cb["sample"]
    assert line["value"] > 50 * cb["value"]                 # north_star: >= 50x the CPU path


def test_committed_driver_line():
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    line = _line("r03_v9_bench.json")
    _check_common(line, baseline)
    assert line["config"]["workload"].startswith("C4") and line["config"]["obstacle_list"] == "spheres"
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["kernel"] == "nn_tile_kernel<3, true>"
    # the counters ride along only because they were taken on these sources and this run key
    tr = _line("r03_traffic.json")
    assert tr["kernel"] == rf["kernel"] and tr["source_sha16"] == line["source_sha16"] and tr["run_key"] == line["run_key"]
    assert rf["traffic"] == tr["traffic_bytes_per_launch"] and 0 < rf["frac_traffic"] <= 1
    assert 0 < rf["issue"]["frac_of_kernel_time"] <= 1 and 0 < tr["wait_ratio"] < 1
    for k in ("algorithmic_GBps", "requested_bytes_per_launch", "frac_requested", "kernel_ms", "kernel_ms_in_loop", "timing"):
        assert k in rf, k
    assert line["launches_per_step"] == 4 and line["value_steady"] > 0 and line["steady_state"]["steps"] == line["steps"]
    assert line["hip_runtime"]["path"] and len(line["source_sha16"]) == 16
    assert 0.4 <= line["roofline_bruteforce"]["frac"] <= 1.0        # north_star's kernel and its >= 40 % target
    assert line["large_batch"]["global_batch"] == 131072 and line["large_batch"]["value"] > line["value"]
    hb = line["host_buffer_path"]
    assert hb["ms_per_step"] > line["ms_per_step"] and hb["registered_arrays"]["ms_per_step"] > line["ms_per_step"]
    poly = line["polygon_obstacles"]["roofline"]
    assert poly["kernel"] == "edges_polygons_kernel" and poly["bound"] == "valu_issue" and poly["algorithmic_bytes_logical"] > 0
    _check_roofline(poly)
    assert poly["frac"] is not None          # round-2 verdict: this block said 16.2


def test_committed_polygon_c3_c5_lines():
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    poly = _line("r03_v9_bench_poly.json")
    _check_common(poly, baseline)
    assert poly["config"]["obstacle_list"] == "polygons" and "polygons" in poly["cpu_baseline"]["sample"]
    _check_roofline(poly["roofline_polygon_edges"])
    assert poly["roofline_polygon_edges"]["frac"] is not None
    c3 = _line("r03_v9_bench_c3.json")
    _check_common(c3, baseline)
    assert c3["config"]["edge"] == "DubinsEdge" and c3["roofline"]["bound"] == "valu_issue" and c3["roofline"]["frac"] is not None
    assert c3["kernel_ms"]["dubins_check"] > c3["kernel_ms"]["dubins_steer"] > 0
    c5 = _line("r03_v9_bench_c5.json")
    _check_common(c5, baseline)
    assert c5["config"]["workload"].startswith("C5") and c5["config"]["n_nodes"] == 500_000
    assert set(c5["phase_ms"]) == {"obstacle_appears", "sweep", "block", "cost_update", "extend_preamble", "append"}
    assert c5["per_cycle"]["sweep_candidates"] > 0 and c5["per_cycle"]["edges_blocked"] > 0 and c5["per_cycle"]["neighbours"] > 0
    assert c5["initial_solve"]["reachable_nodes"] > 100_000
    assert abs(sum(c5["phase_ms"].values()) - c5["ms_per_step"]) < 0.05 * c5["ms_per_step"]


def test_counter_files_say_what_they_were_taken_on():
    names = [n for n in os.listdir(P) if n.startswith("r03_traffic") and n.endswith(".json")]
    assert len(names) >= 4
    for n in names:
        t = _line(n)
        assert t["kernel"] and len(t["source_sha16"]) == 16 and isinstance(t["run_key"], dict) and "config" in t["run_key"], n
        assert t["SQ_INSTS_VALU_per_launch"] > 0 and t["SQ_INSTS_SALU_per_launch"] > 0, n
        assert 0 < t["active_lanes_per_valu_instruction"] <= 64, n


def test_rank_rehearsals_report_what_ran():
    for name, n, e, o in (("r03_bench_2rank_weak_rehearsal.json", 2, 2, 1), ("r03_bench_2rank_obstacles_rehearsal.json", 2, 1, 2),
                          ("r03_bench_4rank_grid2x2_rehearsal.json", 4, 2, 2)):
        d = _line(name)
        assert d["n_gpus"] == n and d["config"]["grid"]["edge_shards"] == e and d["config"]["grid"]["obstacle_shards"] == o
        if o > 1:
            assert "all_reduce(MAX)" in d["config"]["collective"] and d["config"]["obstacle_list"] == "polygons"
