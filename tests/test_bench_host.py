"""Host logic of bench.py that needs no GPU: when `roofline.traffic` may quote the committed PMC file (VERDICT r03 item 6:
"live or honest"), and the committed bench line's own consistency."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _write(tmp_path, name, meta, kernels):
    os.makedirs(tmp_path / "profiles", exist_ok=True)
    d = dict(kernels)
    if meta is not None:
        d["_meta"] = meta
    (tmp_path / "profiles" / name).write_text(json.dumps(d))


KERNELS = {"void conv3x3_pp_kernel<5, false, 0, 8>": {"launches": 3, "fetch_bytes_x2": 200, "write_bytes": 100},
           "void conv_igemm_dma_kernel<5, false, 1>": {"launches": 1, "fetch_bytes_x2": 400, "write_bytes": 100},
           "void bn_apply_kernel<true, false>": {"launches": 9, "fetch_bytes_x2": 1, "write_bytes": 1}}
META = {"steps_profiled": 3.0, "igemm_launches_per_step": {"conv3x3_pp_kernel": 44.0, "conv_igemm_dma_kernel": 8.0}}


def test_traffic_is_quoted_only_from_a_file_recorded_from_the_same_launches(bench, tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    # no file at all
    got, why = bench.pmc_traffic({"conv3x3_pp_kernel": 44.0, "conv_igemm_dma_kernel": 8.0})
    assert got is None and "no PMC traffic file" in why
    # a file without the launch-count record is never quoted
    _write(tmp_path, bench.TRAFFIC_FILES[0], None, KERNELS)
    got, why = bench.pmc_traffic({"conv3x3_pp_kernel": 44.0, "conv_igemm_dma_kernel": 8.0})
    assert got is None and "_meta" in why
    # the same launches: launch-weighted bytes of the implicit-GEMM kernels only (the BatchNorm pass is not part of it)
    _write(tmp_path, bench.TRAFFIC_FILES[0], META, KERNELS)
    got, why = bench.pmc_traffic({"conv3x3_pp_kernel": 44.0, "conv_igemm_dma_kernel": 8.0})
    assert got == round((3 * 300 + 1 * 500) / 4) and why == "profiles/" + bench.TRAFFIC_FILES[0]
    # one launch re-routed to another kernel, a kernel added, a kernel gone: null with the reason
    for launches in ({"conv3x3_pp_kernel": 43.0, "conv_igemm_dma_kernel": 9.0},
                     {"conv3x3_pp_kernel": 44.0, "conv_igemm_dma_kernel": 8.0, "conv_igemm_dma_multi_kernel": 2.0},
                     {"conv3x3_pp_kernel": 44.0}):
        got, why = bench.pmc_traffic(launches)
        assert got is None and "other launches" in why, (launches, why)


def test_the_newest_file_wins_and_older_ones_are_fallbacks(bench, tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    _write(tmp_path, bench.TRAFFIC_FILES[-1], META, KERNELS)
    got, why = bench.pmc_traffic(META["igemm_launches_per_step"])
    assert got is not None and why.endswith(bench.TRAFFIC_FILES[-1])
    newer = {k: dict(v, write_bytes=v["write_bytes"] + 40) for k, v in KERNELS.items()}
    _write(tmp_path, bench.TRAFFIC_FILES[0], META, newer)
    got2, why2 = bench.pmc_traffic(META["igemm_launches_per_step"])
    assert why2.endswith(bench.TRAFFIC_FILES[0]) and got2 == got + 40


@pytest.mark.parametrize("line_file", ["r05_final_bench_line.json", "r06_final_bench_line.json"])
def test_committed_bench_line_is_consistent_with_its_committed_evidence(bench, monkeypatch, line_file):
    """profiles/r0N_final_bench_line.json (what `python bench.py` printed on the GPU box) against the files it names:
    the traffic it quotes is what pmc_traffic() derives from the committed PMC file it names for the launches that file
    records (the line of the final evidence run is printed BEFORE that run's own PMC passes, i.e. against the previous
    round's file; the next run quotes the new one), value x ms_per_step = the batch, every `frac` is achieved / peak, the
    measured ceilings are ordered achieved < attainable < mfma_stream < peak, every other configuration names an existing
    profile."""
    with open(os.path.join(ROOT, "profiles", line_file)) as f:
        line = json.load(f)
    named = line["roofline"]["traffic_source"].split("/", 1)[1]
    with open(os.path.join(ROOT, "profiles", named)) as f:
        meta = json.load(f)["_meta"]
    monkeypatch.setattr(bench, "TRAFFIC_FILES", tuple(f for f in bench.TRAFFIC_FILES if f <= named))   # newest <= the named one
    got, src = bench.pmc_traffic(meta["igemm_launches_per_step"])
    assert src == line["roofline"]["traffic_source"]
    assert abs(got - line["roofline"]["traffic"]) <= 0.02 * got
    r = line["roofline"]
    assert r["achieved"] < r["attainable"] < r["mfma_stream"] < r["peak"]
    assert abs(r["frac_of_attainable"] - r["achieved"] / r["attainable"]) < 1e-3
    assert abs(r["frac_of_mfma_stream"] - r["achieved"] / r["mfma_stream"]) < 1e-3
    assert abs(line["value"] * line["ms_per_step"] / 1e3 - 512) < 0.5
    for key in ("roofline", "roofline_wgrad"):
        r = line[key]
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    assert line["agreement"]["within_tolerance"] and line["agreement"]["tolerance"] == bench.LOGIT_TOLERANCE
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    if "roofline_rules" in line:       # round 6: the fused head's achieved GB/s follows from its bytes and microseconds
        rr = line["roofline_rules"]
        assert abs(rr["achieved"] - rr["algorithmic_bytes"] / (rr["us_per_step"] * 1e-6) / 1e9) < 0.02 * rr["achieved"] + 0.01
        assert abs(rr["frac"] - rr["achieved"] / rr["peak"]) < 1e-4
    for o in line["other_configs"]:
        assert os.path.exists(os.path.join(ROOT, o["profile"])), o["profile"]
        assert abs(o["frac"] - o["achieved"] / o["peak"]) < 1e-3
        assert abs(o["value"] * o["ms_per_step"] / 1e3 - o["batch_per_gpu"]) < 0.01 * o["batch_per_gpu"]
