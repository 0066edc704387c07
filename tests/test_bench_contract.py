"""The measurement contract of bench.py, checked on the committed result of the round (profiles/r03_bench.json, written by
tools/profile_round3.sh on the GPU box) and on bench.py's own source - no GPU needed:
  * ONE JSON line with the driver's keys, the roofline and cpu_baseline objects and their required fields;
  * `roofline.frac` = achieved / peak, a fraction (<= 1) of the 8 TB/s HBM peak, derived from the counter traffic the profiles hold;
  * `value` is the reference-semantics number (every frame complete on return), the throughput mode sits beside it;
  * the product path of bench.py never touches oracle/ outside the cpu_baseline legs."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    txt = open(os.path.join(ROOT, "profiles", "r03_bench.json")).read().strip().splitlines()
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


def test_traffic_is_the_counter_figure_of_the_committed_profile():
    d = _line()
    txt = open(os.path.join(ROOT, "profiles", "r03_sweep_pmc_hbm_traffic.txt")).read()
    m = re.search(r"= ([0-9.]+) MB \+ ([0-9.]+) MB = ([0-9.]+) MB", txt)
    assert m, txt
    assert abs(float(m.group(3)) * 1e6 - d["roofline"]["traffic"]) <= 0.06e6
    # the kernel time of the rocprofv3 --kernel-trace pass and the live hipEvent time of the bench agree (the --pmc passes run the kernel a
    # few per cent slower: looser bound)
    kt = open(os.path.join(ROOT, "profiles", "r03_sweep_kernel_stats.txt")).read()
    avg = float(re.search(r"k_sweep_tile<true>[^|]*\|\s*\d+\s*\|\s*[0-9.]+\s*\|\s*([0-9.]+)", kt).group(1))
    assert abs(avg - d["roofline"]["avg_launch_ms"] * 1e3) < 0.08 * avg, (avg, d["roofline"]["avg_launch_ms"])
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
    # every import of the oracle library sits inside a function whose name starts with cpu_ (the cpu_baseline legs)
    for m in re.finditer(r"^(\s*)(from tests|import tests|from tests\.|.*oracle_lib|.*load_oracle).*$", src, re.M):
        head = src[:m.start()]
        fn = re.findall(r"^def (\w+)\(", head, re.M)
        assert fn and fn[-1].startswith("cpu_"), (m.group(0).strip(), fn[-1] if fn else None)
