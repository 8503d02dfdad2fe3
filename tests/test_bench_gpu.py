"""bench.py end to end on the MI355X with a tiny setting: the JSON contract (keys, units, roofline and cpu_baseline
objects) is what the driver parses."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--batch", "64", "--cpu-batch", "4"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "images/sec" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["scaling"] == "weak" and d["dtype"] == "bf16" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["achieved"] > 0 and (r["traffic"] is None or r["traffic"] > 0)
    # measured ceilings beside the spec peak: the dominant kernel's own steady-state K loop and a
    # register-only MFMA stream, both on this box, after the timed loop
    assert r["attainable_kernel"] == "conv3x3_pp_kernel" and 0 < r["attainable"] <= r["mfma_stream"] * 1.02 < 1.02 * r["peak"]
    assert abs(r["frac_of_attainable"] - r["achieved"] / r["attainable"]) < 1e-3
    assert 0.3 < r["attainable_frac_of_peak"] < 1 and 0.5 < r["mfma_stream_frac_of_peak"] < 1
    # the rules layer's own roofline object: wall us of the fused head launch, its algorithmic bytes, GB/s
    rr = d["roofline_rules"]
    assert rr["bound"] == "hbm" and rr["unit"] == "GB/s" and rr["peak"] == 8000.0 and rr["us_per_step"] > 0
    assert rr["algorithmic_bytes"] == 4 * (64 * 640 + rr["child_slots"] * 640 + 64 * 10) and rr["child_slots"] == 18
    assert abs(rr["achieved"] - rr["algorithmic_bytes"] / (rr["us_per_step"] * 1e-6) / 1e9) < 0.02 * rr["achieved"] + 0.01
    assert "one kernel boundary" in rr["stated_bound"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert abs(d["value"] - 64 * 2 / (d["ms_per_step"] * 2e-3)) < 0.02 * d["value"]
    # `roofline.traffic` is quoted from a committed PMC file only if this run launched the same kernels as often
    # (this 64-image run does not: other tile counts pick other kernels) -- otherwise null with the reason
    assert r["traffic"] is None and "not quoted" in r["traffic_source"]
    # "top-1 vs ref": the logit error of the trained weights against the fp32 CPU port stays inside DESIGN.md's tolerance
    a = d["agreement"]
    assert a["tolerance"] == 3e-2 and a["within_tolerance"] and a["max_abs_logit_err_over_scale"] < a["tolerance"]
    # the other BASELINE.json configurations, each with its bound and fraction
    oc = d["other_configs"]
    assert [o["config"][:2] for o in oc] == ["C1", "C3", "C4", "C4", "C4", "C5", "C5"]
    for o in oc:
        assert o["value"] > 0 and o["unit"] == "images/sec" and o["bound"] in ("mfma", "hbm") and 0 < o["frac"] < 1
        assert abs(o["value"] - o["batch_per_gpu"] / (o["ms_per_step"] * 1e-3)) < 0.02 * o["value"]
        assert abs(o["frac"] - o["achieved"] / o["peak"]) < 1e-3
        assert os.path.exists(os.path.join(ROOT, o["profile"])), o["profile"]


def test_bench_two_rank_branch_runs_on_one_gpu_over_gloo():
    """`bench.py --gpus 2` re-launches itself through torch.distributed.run (one rank per GPU over RCCL on a
    multi-GPU node).  On a 1-GPU box the same branch -- self-launch, per-rank shard, GradComm buckets reduced during
    backward, rank 0's CU-sharing decision broadcast, max-over-ranks timing, the `comm` object -- runs with both ranks
    on cuda:0 exchanging through gloo (--backend gloo --share-gpu), so the driver's scaling run is never the first
    execution of that code."""
    import torch
    multi = torch.cuda.device_count() >= 2
    extra = [] if multi else ["--backend", "gloo", "--share-gpu"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--batch", "64", "--no-cpu-baseline", "--agreement-n", "0"] + extra,
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2"
    c = d["comm"]
    assert c["ranks"] == 2 and c["backend"] == ("nccl" if multi else "gloo") and c["buckets"] == 3
    assert c["allreduce_bytes_per_rank"] > 4 * 36_000_000 and "allreduce_ms_exposed" in c
    assert len(c["cu_share_per_rank"]) == 2 and c["cu_share_per_rank"][0] == c["cu_share_per_rank"][1]
    assert abs(d["value"] - 128 * 2 / (d["ms_per_step"] * 2e-3)) < 0.02 * d["value"]
    assert "roofline" in d and "cpu_baseline" not in d


@pytest.mark.parametrize("ranks", [4, 8])
def test_bench_n_rank_branch_at_the_driver_rank_counts(ranks):
    """bucket order, `broadcast_flag`, max-over-ranks timing and the `comm` object had only ever seen
    2 ranks.  The driver's scaling run uses 4 and 8: the same branch with that many ranks, all on this box's GPU over
    gloo (one rank per GPU over RCCL when the box has them).  Every rank applies the same all-reduced gradient, so the
    replicas must be bit-identical after the steps -- bench.py gathers two checksums of the weights per rank."""
    import torch
    multi = torch.cuda.device_count() >= ranks
    extra = [] if multi else ["--backend", "gloo", "--share-gpu"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "3",
                          "--warmup", "1", "--batch", "64", "--no-cpu-baseline", "--agreement-n", "0",
                          "--no-kernel-timer", "--no-other-configs"] + extra,
                         capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["config"]["global_batch"] == 64 * ranks
    assert d["config"]["parallelism"] == f"dp{ranks}" and d["scaling"] == "weak"
    c = d["comm"]
    assert c["ranks"] == ranks and c["buckets"] == 3 and c["backend"] == ("nccl" if multi else "gloo")
    assert len(c["cu_share_per_rank"]) == ranks and len(set(c["cu_share_per_rank"])) == 1      # rank 0 decided for all
    assert len(c["replica_checksums"]) == ranks and c["replicas_identical"], c["replica_checksums"]
    assert c["replica_checksums"][0][1] > 0
    # the exchange the first real 8-GPU run will make, asserted rather than assumed -- the three
    # buckets are issued in the order backward completes them (stage 3 + head first: the highest offsets of the flat
    # buffer), they tile the whole gradient buffer exactly once, and the CU reservation follows the backend
    rng = c["bucket_ranges_last_step"]
    assert rng == c["grad_buckets_planned"] and len(rng) == 3
    assert rng[0][0] > rng[1][0] > rng[2][0] == 0 and rng[0][1] * 4 == c["allreduce_bytes_per_rank"]
    assert rng[1][1] == rng[0][0] and rng[2][1] == rng[1][0]
    assert sum(c["bucket_bytes_last_step"]) == c["allreduce_bytes_per_rank"]
    assert c["reserved_cus_while_buckets_in_flight"] == (c["rccl_max_nchannels"] if multi else 0)
    assert abs(d["value"] - 64 * ranks * 3 / (d["ms_per_step"] * 3e-3)) < 0.02 * d["value"]
