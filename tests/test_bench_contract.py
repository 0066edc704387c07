"""The measurement contract of bench.py, checked on the committed result of the round (profiles/r05_bench.json, written by
tools/round5_suite_bench.sh on the GPU box) and on bench.py's own source - no GPU needed:
  * ONE JSON line with the driver's keys, the roofline and cpu_baseline objects and their required fields;
  * `roofline.frac` = achieved / peak, a fraction (<= 1) of the 8 TB/s HBM peak, derived from the counter traffic the profiles hold;
  * `value` is the reference-semantics number (every frame complete on return), the throughput mode sits beside it;
  * the product path of bench.py never touches oracle/ outside the cpu_baseline legs."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    txt = open(os.path.join(ROOT, "profiles", "r05_bench.json")).read().strip().splitlines()
    assert len(txt) == 1, "bench.py prints exactly one line on stdout"
    return json.loads(txt[0])


def test_driver_keys_and_objects():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - d["n_gpus"]) < 1e-6 * d["n_gpus"] + 1e-9          # value = frames of all ranks / time
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.0 < r["frac"] <= 1.0
    assert abs(r["achieved"] - r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]   # counter bytes per launch / mean launch time
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["unit"] == d["unit"]
    assert c["multi_process"]["processes"] == 8
    assert d["steps"] == 148                                                                            # the default run IS the headline run
    s = d["speedup_vs_cpu_baseline"]
    assert abs(s["value"] - d["value"] / c["value"]) < 1e-9 * s["value"]
    v, l = r["valu"], r["lds"]                                                                           # the on-chip units next in line, by counters
    assert 0.0 < v["issue_frac"] < 1.0 and 0.0 < l["busy_frac"] < 1.0 and l["bank_conflict_cycles_per_launch"] < l["active_quad_cycles_per_launch"]


def test_sharded_legs_of_the_two_rank_run():
    """with N > 1 the bench shards configs[4] (`large`) and the OMD-shaped graph, not only the control graph.  The committed line is
    `bench.py --gpus 2` with both ranks on ONE MI355X (gloo carrying the all-reduces through the host: the times are a functional record, not a
    scaling claim - the 8-GPU run is the driver's)."""
    d = json.loads(open(os.path.join(ROOT, "profiles", "r05_bench_gpus2_shared_gpu.json")).read().strip().splitlines()[-1])
    assert d["n_gpus"] == 2
    sh = d["sharded"]
    for tag, key in (("control", "ms_per_lm_iter_sharded"), ("omd", "ms_per_lm_iter_omd_sharded"), ("large", "ms_per_lm_iter_large_sharded")):
        leg = sh[tag]
        assert leg["same_trajectory_as_1gpu"] is True, tag
        assert leg["ms_per_lm_iter_sharded"] > 0 and leg["ms_per_lm_iter_1gpu"] > 0 and d[key] == leg["ms_per_lm_iter_sharded"]
        assert leg["allreduces_per_lm_iter"] > 0 and leg["allreduce_bytes_per_lm_iter"] > 0
    # the exchanged bytes grow with the number of pose vertices only: the large graph has 239 poses + 20 x 238 motions
    assert sh["large"]["allreduce_bytes_per_lm_iter"] > sh["omd"]["allreduce_bytes_per_lm_iter"] > sh["control"]["allreduce_bytes_per_lm_iter"]


def test_traffic_is_the_counter_figure_of_the_committed_profile():
    d = _line()
    txt = open(os.path.join(ROOT, "profiles", "r05_sweep_pmc_hbm_traffic.txt")).read()
    m = re.search(r"= ([0-9.]+) MB \+ ([0-9.]+) MB = ([0-9.]+) MB", txt)
    assert m, txt
    assert abs(float(m.group(3)) * 1e6 - d["roofline"]["traffic"]) <= 2e-3 * d["roofline"]["traffic"]      # (the bench ran its own two --pmc passes: same graph, same layout)
    # the kernel time of the rocprofv3 --kernel-trace pass and the live hipEvent time of the bench agree (the --pmc passes run the kernel a
    # few per cent slower: looser bound)
    kt = open(os.path.join(ROOT, "profiles", "r05_sweep_kernel_stats.txt")).read()
    avg = float(re.search(r"k_sweep_tile<true, true>[^|]*\|\s*\d+\s*\|\s*[0-9.]+\s*\|\s*([0-9.]+)", kt).group(1))
    assert abs(avg - d["roofline"]["avg_launch_ms"] * 1e3) < 0.10 * avg, (avg, d["roofline"]["avg_launch_ms"])
    us = float(re.search(r"avg_duration=([0-9.]+) us", txt).group(1))
    assert abs(us - d["roofline"]["avg_launch_ms"] * 1e3) < 0.15 * us


def test_value_is_the_reference_semantics_number():
    d = _line()
    assert d["value"] == d["value_sync"] and d["value_deferred"] >= 0.9 * d["value"]
    assert "complete when its call returns" in d["config"]["value"]
    a = d["config"]["accuracy_asserted"]["bounds"] if "accuracy_asserted" in d["config"] else None
    if a is not None:
        assert d["config"]["trajectory_drift_m"] <= a["trajectory_drift_m"] and all(e <= a["object_motion_error_m"] for e in d["config"]["object_motion_error_m_last_frame"])


def test_bench_source_keeps_the_oracle_out_of_the_product_path():
    src = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"oracle", src)]
    assert uses, "the cpu_baseline legs load the oracle"
    # every import of the oracle library / the checkers under tests/ sits inside a function whose name starts with cpu_ (the cpu_baseline legs) or parity_ (the parity
    # leg: the reference's own run of the timed sequence in child processes + the comparison - checked BEFORE anything is timed, never inside a timed region)
    for m in re.finditer(r"^(\s*)(from tests|import tests|from tests\.|.*oracle_lib|.*load_oracle).*$", src, re.M):
        head = src[:m.start()]
        fn = re.findall(r"^def (\w+)\(", head, re.M)
        assert fn and (fn[-1].startswith("cpu_") or fn[-1].startswith("parity_")), (m.group(0).strip(), fn[-1] if fn else None)
